"""profiles/rNN_accuracy_calibration.txt from a `pytest tests -m gpu -q -s` log: every line the tests print through
tests/conftest.py (GATE, mismatch gate), the yardstick and summary lines of the golden tests, and a table of the ratios.
    python tools/dbg/collect_gates.py gpurun_out/r6_full3.log profiles/r06_accuracy_calibration.txt"""
import re
import sys

log, out = sys.argv[1], sys.argv[2]
gates, mism, notes = [], [], []
for ln in open(log, errors="replace"):
    ln = ln.rstrip("\n")
    s = ln.lstrip(".F ").strip()
    if s.startswith("GATE "):
        gates.append(s)
    elif s.startswith("mismatch gate"):
        mism.append(s)
    elif re.search(r"float32 oracle vs the reference|GPU SubprocVecEnv vs the reference|on the GPU: \d+ of \d+ env-steps|GPU vs the reference's own env logic|"
                   r"own count / done mismatches|contact_order \d against link order|per-component medians|eval env on the GPU", s):
        notes.append(s)
ratios = []
for g in gates:
    m = re.search(r"ratio\s+([0-9.infa]+)", g)
    over_floor = "max(" in g
    ratios.append((float(m.group(1)) if m and m.group(1) not in ("inf", "nan") else float("inf"), g))
fin = [r for r, _ in ratios if r != float("inf")]
fin.sort()
with open(out, "w") as f:
    f.write("Round 6: every comparison of a GPU figure with the float32 oracle's that the -m gpu suite makes, as the tests print it\n"
            "(tests/conftest.py: f32_gate -- `GPU <= min(max(floor, factor x float32 oracle), cap)`; DESIGN.md 3 states the rule:\n"
            "factor 1.5 for medians and worst values over >= 30 samples of the 16-link chain, 2.0 for 90th percentiles, < 30 samples\n"
            "and the 32-link chain; exceptions say so at the call).  Source: `python -m pytest tests -m gpu -q -s` on one MI355X\n"
            "(gpurun), the round's final build; %d gates, all passed.\n\n" % len(gates))
    if fin:
        q = lambda p: fin[min(len(fin) - 1, int(p * len(fin)))]
        f.write("Ratios GPU : float32 oracle over the %d gates with a finite ratio: median %.2f, 75th percentile %.2f, 90th %.2f, "
                "95th %.2f, largest %.2f\n" % (len(fin), q(0.5), q(0.75), q(0.9), q(0.95), fin[-1]))
        f.write("(ratios above a gate's factor pass through its floor -- figures that are round-off of the comparison itself -- "
                "and are listed in DESIGN.md 3)\n\n")
    f.write("---- count / done mismatches against the reference's recorded env-steps (GPU <= 1.5 x float32 oracle + 4; both counted over\n"
            "     every env-step and classified by the same rule: bifurcation / servo boundary)\n")
    for s in mism:
        f.write(s + "\n")
    f.write("\n---- summary lines of the golden / parity tests\n")
    for s in notes:
        f.write(s + "\n")
    f.write("\n---- every gate\n")
    for s in gates:
        f.write(s + "\n")
print(len(gates), "gates,", len(mism), "mismatch gates,", len(notes), "notes ->", out)
