"""The bits the step kernels produce, pinned: sha256 over four gait env-steps of 2000 environments (outputs + final
states) for the three kernel families -- 16 links register-resident, 16 links on the streamed-row kernels, 32 links.
Arithmetic is deterministic here (no atomics in the data path, results independent of the schedule: the scheduler tests),
so an edit that is meant to leave the arithmetic alone (a re-layout, a register split, a fence, the stash experiment of
round 5) must leave these hashes alone, and an edit that is meant to change it changes them HERE, visibly, with the
reason in the commit.  tools/state_hash.py prints the same hashes for any library (SNK_LIB).

The pins belong to ONE toolchain and flag set: another hipcc may contract FMAs or select instructions
differently without any arithmetic bug.  They are therefore keyed by what build.py recorded next to the library
(libsnk.so.cmd: the command line; libsnk.so.toolchain: hipcc's version lines); under an unknown key the test SKIPS and says
what to do -- run the parity tests, then `python tools/state_hash.py --pins` and add the block -- instead of failing or
being re-pinned blindly.  A mismatch names the parts that differ (observations, rewards, done flags, substep counts, state,
aux)."""
import hashlib
import importlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PARTS = ("obs", "rew", "done", "substeps", "state", "aux")
# key: sha256(libsnk.so.toolchain + "\n" + libsnk.so.cmd)[:12]
PINNED = {
    # round 6 (hipcc of ROCm 7.2.0: HIP 7.2.26015, AMD clang 22.0.0git; build.py's flags).  "all" = round 5's pins
    # (8e46618d..., f48c5d83..., 25eb76a4...): the arithmetic of round 5's final build -- contact_order 0 takes the same path
    "6e946590e810": {
        "16": {'obs': '20a178b133141cc1', 'rew': '7e0f39a54e618ecb', 'done': '69b49cbbddfcc133', 'substeps': 'c1e80f7aee05f6f5', 'state': '63779d129eacdd03', 'aux': '68e69346663f4505', 'all': '8e46618de5a57d8c'},
        "16s": {'obs': '5587b1038112da41', 'rew': '6f4c9cf6ee2b66da', 'done': 'a69c24752b91ccc8', 'substeps': 'c8e75749831a07d3', 'state': '89ae4711ca28174d', 'aux': 'ec9c875e540978d9', 'all': 'f48c5d83dd6c7bec'},
        "32": {'obs': 'dac6bd0cdc9b684a', 'rew': 'f175348253a73275', 'done': '915f52196924625a', 'substeps': 'e34ba6c0d736d5e6', 'state': '7603409719661c0a', 'aux': '16461c5d3e51163d', 'all': '25eb76a43fe2128d'},
    },
}


def toolchain_key():
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bullet-envs_amd")
    try:
        txt = open(os.path.join(here, "libsnk.so.toolchain")).read() + "\n" + open(os.path.join(here, "libsnk.so.cmd")).read()
    except OSError:
        return None, "libsnk.so.toolchain / .cmd are missing (a library not built by bullet-envs_amd/build.py)"
    return hashlib.sha256(txt.encode()).hexdigest()[:12], txt


def state_hashes(pkg, n, streamed):
    """{part: hash} + "all" for the pinned workload."""
    syn = importlib.import_module("bullet-envs_amd.synthetic")
    B = 2000
    st = pkg.Stepper(B, n_modules=n)
    st.reset()
    st.set_ground_friction((0.5 + np.arange(B) % 11 / 10.0).astype(np.float32))
    hs = {k: hashlib.sha256() for k in PARTS + ("all",)}
    for j in range(4):
        out = st.step(syn.gait_actions(np.arange(B), j, n // 2).astype(np.float32))
        for k, a in zip(PARTS[:4], out):
            raw = np.ascontiguousarray(a).tobytes()
            hs[k].update(raw)
            hs["all"].update(raw)
    S, X = st.get_state()
    for k, a in (("state", S), ("aux", X)):
        hs[k].update(a.tobytes())
        hs["all"].update(a.tobytes())
    st.close()
    return {k: h.hexdigest()[:16] for k, h in hs.items()}


@pytest.mark.parametrize("n,streamed", [(16, False), (16, True), (32, False)])
def test_state_hash_is_pinned(pkg, monkeypatch, n, streamed):
    key, txt = toolchain_key()
    if key not in PINNED:
        pytest.skip("no pinned hashes for this toolchain / flag set (key %s: %s).  Verify the parity tests on it, then run "
                    "`python tools/state_hash.py --pins` on the GPU box and add the block it prints to PINNED." % (key, txt))
    if streamed:
        monkeypatch.setenv("SNK_FORCE_STREAMED", "1")
    got = state_hashes(pkg, n, streamed)
    want = PINNED[key]["%d%s" % (n, "s" if streamed else "")]
    differ = [k for k in PARTS if got[k] != want[k]]
    assert got["all"] == want["all"] and not differ, "n = %d streamed = %s: %s differ from the pinned bits (got %s)" % (
        n, streamed, ", ".join(differ) or "the concatenation", got)
