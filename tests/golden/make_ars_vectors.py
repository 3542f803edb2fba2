"""Generates tests/golden/ars_vectors.npz by RUNNING the reference's own Normalizer class and
policy() function.  ars/train.py as a module needs gym/pybullet/tensorboardX, which this image
lacks, so only those two definitions (ars/train.py:38-39 and :152-173, plain numpy) are
executed from the file's text -- nothing of it is stored.

  * 3 "env-steps" of 6 directions each: every row observed then normalised in order (the loop
    of ars/train.py:88-94); normalised rows, and n / mean / mean_diff / var after each step
  * the actions policy(state_d, W_d) of the last step

Run here:  python tests/golden/make_ars_vectors.py
"""
import os

import numpy as np

SRC = "/root/reference/ars/train.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ars_vectors.npz")
lines = open(SRC).read().split("\n")


def block(start_pat):
    i = next(k for k, l in enumerate(lines) if l.startswith(start_pat))
    j = i + 1
    while j < len(lines) and (lines[j].startswith(("\t", " ")) or lines[j].strip() == ""):
        j += 1
    return "\n".join(lines[i:j])


ns = {"np": np, "os": os}
exec(block("def policy("), ns)
exec(block("class Normalizer"), ns)
rng = np.random.default_rng(7)
N, O, A = 6, 56, 8
norm = ns["Normalizer"]([1, O])
d = {}
for step in range(3):
    X = rng.normal(size=(N, O)) * rng.uniform(0.05, 3.0, size=(1, O)) + rng.normal(size=(1, O))
    out = []
    for r in range(N):
        norm.observe(X[r])
        out.append(norm.normalize(X[r]).reshape(-1))
    d["X%d" % step] = X
    d["Y%d" % step] = np.array(out)
    for k in ("n", "mean", "mean_diff", "var"):
        d["%s%d" % (k, step)] = np.array(getattr(norm, k)).reshape(-1)
W = rng.normal(size=(N, A, O))
d["W"] = W
d["actions"] = np.array([ns["policy"](d["Y2"][r], W[r]) for r in range(N)])
np.savez_compressed(OUT, **d)
print("wrote", OUT, os.path.getsize(OUT), "bytes", d["actions"].shape)
