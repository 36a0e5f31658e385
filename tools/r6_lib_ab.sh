#!/bin/bash
# round 6: A/B of two builds of the library on one box -- lssvc_amd/lib/liblssvc_hip_prev.so (a copy kept from before a change; not tracked)
# against the current one -- with tools/r6_families_ab.py in the modes given (default: ablate = the split-roles kernels).
OUT=gpurun_out/r6
mkdir -p $OUT
: > $OUT/lib_ab.txt
for rep in 1 2; do
  for mode in ${@:-ablate}; do
    for lib in liblssvc_hip_prev.so liblssvc_hip.so; do
      echo "== $lib $mode" >> $OUT/lib_ab.txt
      LSSVC_HIP_LIB=$PWD/lssvc_amd/lib/$lib timeout -k 10 300 python tools/r6_families_ab.py $mode 3 20 2>/dev/null | grep -v amdgpu.ids >> $OUT/lib_ab.txt || exit 1
    done
  done
done
cat $OUT/lib_ab.txt
