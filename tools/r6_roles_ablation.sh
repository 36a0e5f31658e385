#!/bin/bash
# round 6: where do the split-roles kernels (narrow heads, stride 2) lose their time against the patch stream alone (tools/probes/patch_stream_probe.hip)?
# Diagnostic libraries with one part of the schedule switched off each (make P3_ABLATE=n: 1 no weight DMA, 2 no MFMAs, 4 no patch loads, 8 no
# conversion / LDS stores; LSSVC_CONV_DEBUG=32: no epilogue), timed on the same box as the shipped one. Results of the ablated builds are wrong by
# design. Runs on the GPU box (the objects it overwrites live in the box's scratch copy of the repo).
set -e
OUT=gpurun_out/r6
mkdir -p $OUT /tmp/abl
cp lssvc_amd/lib/liblssvc_hip.so /tmp/abl/base.so
for a in 1 2 4 8 3 6; do
  rm -f lssvc_amd/csrc/conv3_f16x3p_r2.o
  make -C lssvc_amd/csrc P3_ABLATE=$a OUT=/tmp/abl/abl$a.so > /tmp/abl/make$a.log 2>&1
done
: > $OUT/roles_ablation.txt
run() { echo "== $1" >> $OUT/roles_ablation.txt; shift; env "$@" timeout -k 10 120 python tools/r6_families_ab.py ablate 3 20 2>/dev/null | grep -v amdgpu.ids >> $OUT/roles_ablation.txt; }
run "shipped library" LSSVC_HIP_LIB=/tmp/abl/base.so
run "no epilogue (LSSVC_CONV_DEBUG=32)" LSSVC_HIP_LIB=/tmp/abl/base.so LSSVC_CONV_DEBUG=32
run "no weight DMA after phase 0 (P3_ABLATE=1)" LSSVC_HIP_LIB=/tmp/abl/abl1.so
run "no MFMAs (P3_ABLATE=2)" LSSVC_HIP_LIB=/tmp/abl/abl2.so
run "no patch loads (P3_ABLATE=4)" LSSVC_HIP_LIB=/tmp/abl/abl4.so
run "no conversion / LDS stores of the patch (P3_ABLATE=8)" LSSVC_HIP_LIB=/tmp/abl/abl8.so
run "no weight DMA, no MFMAs (P3_ABLATE=3)" LSSVC_HIP_LIB=/tmp/abl/abl3.so
run "no MFMAs, no patch loads (P3_ABLATE=6)" LSSVC_HIP_LIB=/tmp/abl/abl6.so
run "no weight DMA, no MFMAs, no epilogue" LSSVC_HIP_LIB=/tmp/abl/abl3.so LSSVC_CONV_DEBUG=32
cat $OUT/roles_ablation.txt
