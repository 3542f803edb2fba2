#!/bin/bash
# Register pressure of one step kernel, and whether anything is reloaded from scratch memory INSIDE its Gauss-Seidel loop
# (the register-resident solve pins 208 registers across that loop).
#   bash tools/dbg/loop_spills.sh -DKN=16 -DKV=true     (default: the register-resident 16-link kernel)
#   bash tools/dbg/loop_spills.sh -DKN=32 -DKV=false
# Compiles with build.py's flags into /tmp/snk_loop; extra arguments go to hipcc (compiler experiments).
here=$(cd "$(dirname "$0")" && pwd)
out=/tmp/snk_loop; mkdir -p $out
flags=$(python3 "$here/../../bullet-envs_amd/build.py" --print-flags)     # the product's own flags, from one place
/opt/rocm/bin/hipcc $flags \
    --cuda-device-only -S "$@" "$here/one_kernel.hip" -o $out/k.s -Rpass-analysis=kernel-resource-usage 2> $out/k.log
grep -E "VGPRs Spill|SGPRs Spill|ScratchSize|  VGPRs:" $out/k.log | sed 's/.*remark: *//' | tr '\n' ' '; echo
python3 "$here/loop_spills.py" $out/k.s
