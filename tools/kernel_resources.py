#!/usr/bin/env python3
"""Registers, scratch bytes per lane and occupancy of every kernel of libsnk.so, as the compiler reports them: a device-only
compile of csrc/snk_api.hip to assembly with build.py's flags (35 s), the `; NumVgprs / ScratchSize / Occupancy` comments
of each kernel.  The numbers DESIGN.md 5 / 8 quote come from here (profiles/r05_kernel_resources.txt).
    python tools/kernel_resources.py [-DSNK_V1_RESN=48 ...]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-mllvm", "-greedy-regclass-priority-trumps-globalness=1"]


def main():
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "snk.s")
        cmd = ["/opt/rocm/bin/hipcc"] + FLAGS + sys.argv[1:] + ["--offload-device-only", "-S", "csrc/snk_api.hip", "-o", out]
        subprocess.run(cmd, cwd=os.path.join(ROOT, "bullet-envs_amd"), check=True, stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    print("# " + " ".join(cmd[:-2]))
    print("%-46s %6s %6s %9s %5s" % ("kernel", "VGPRs", "SGPRs", "scratch B", "occ"))
    name = None
    for i, l in enumerate(lines):
        m = re.match(r"^(_ZN3snk\w+):\s*; @", l)
        if m:
            name = m.group(1)
        m = re.match(r"; ScratchSize: (\d+)", l)
        if m and name and "kernel" in name:
            blk = "\n".join(lines[i - 14:i + 14])
            g = lambda k: (re.search(r"; %s: (\d+)" % k, blk) or [None, "?"])[1]      # noqa: E731
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            short = re.sub(r"\(.*", "", dem).replace("void snk::", "")
            print("%-46s %6s %6s %9s %5s" % (short, g("NumVgprs"), g("NumSGPRsForWavesPerEU"), m.group(1), g("Occupancy")))
            name = None


if __name__ == "__main__":
    main()
