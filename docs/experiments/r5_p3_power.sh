#!/bin/bash
# Where the persistent 3x3 kernel's clock goes (round 5): the stamp build under ablations, in-kernel clock and wall time of each.
#   256 = stamps only; +1 no patch traffic after the first fill; +2 no weight DMA after the first fill; +4 patch loads without
#   conversion / LDS stores; +32 no epilogue.  P3_SPLIT=1: pre-split input, patch by LDS-DMA.
out=gpurun_out/r05_p3_power.txt
: > $out
for dbg in 256 257 258 259 260 288 291; do
  LSSVC_CONV_DEBUG=$dbg timeout -k 10 120 python tools/p3_stamps.py 2>&1 | grep -v amdgpu.ids | grep -A1 "^64->64" >> $out || exit 1
done
for dbg in 256 257 258; do
  P3_SPLIT=1 LSSVC_CONV_DEBUG=$dbg timeout -k 10 120 python tools/p3_stamps.py 2>&1 | grep -v amdgpu.ids | grep -A1 "^64->64" | sed 's/^64->64/split-in 64->64/' >> $out || exit 1
done
cat $out
