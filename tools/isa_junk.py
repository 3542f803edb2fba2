"""Compiler-generated instructions between the hand-written asm row steps of the solve loop.
usage: python tools/isa_junk.py   (compiles csrc/snk_api.hip with -save-temps into /tmp/snk_isa)"""
import os, subprocess, sys
from collections import Counter
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = "/tmp/snk_isa"
os.makedirs(out, exist_ok=True)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-save-temps",
                "-c", os.path.join(ROOT, "bullet-envs_amd/csrc/snk_api.hip"), "-o", os.path.join(out, "snk.o")],
               cwd=out, check=True, stderr=subprocess.DEVNULL)
s = open(os.path.join(out, "snk_api-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
a = s.index("_ZN3snk15env_step_kernelILi16EEEvPKNS_8DevModelEPfPKfS4_S4_S4_PhPiiiPKi:")
f = s[a:s.index(".Lfunc_end", a)].split("\n")
blocks, cur = [], None
for i, l in enumerate(f):
    if "ASMSTART" in l: cur = [i, None, []]
    elif "ASMEND" in l and cur: cur[1] = i; blocks.append(cur); cur = None
    elif cur is not None: cur[2].append(l.strip())
def kind(body):
    t = " ".join(body)
    if "v_rsq_f32" in t: return "cone"
    if "s_mov_b64 exec" in t: return "motor"
    if "v_permlane32_swap" in t: return "duo"
    if "v_max3" in t: return "res"
    if t.strip() == "s_nop 1": return "nop"
    return "other"
kinds = [kind(b[2]) for b in blocks]
print(Counter(kinds))
tot = Counter(); n = 0
for i in range(len(blocks) - 1):
    if kinds[i] in ("nop", "motor") and kinds[i + 1] in ("cone", "duo", "motor"):
        for l in f[blocks[i][1] + 1:blocks[i + 1][0]]:
            l = l.strip()
            if l and not l.startswith(";") and not l.startswith("."):
                tot[l.split()[0]] += 1; n += 1
print("instructions between residual-free row steps:", n, tot.most_common(12))
