#!/bin/bash
# A/B of two library builds (lssvc_amd/lib/liblssvc_hip.so against liblssvc_hip_old.so) on the FFN kernels: tests, microbench, bench
set -x
mkdir -p gpurun_out/r3f
timeout -k 10 400 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "ffn or depth_conv" > gpurun_out/r3f/pytest2.log 2>&1; rc=$?; tail -3 gpurun_out/r3f/pytest2.log; [ $rc -eq 0 ] || exit 1
OLD=$PWD/lssvc_amd/lib/liblssvc_hip_old.so
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-h2d-pass --no-side-configs --no-events"
for i in 1 2 3; do
timeout -k 10 300 $B > gpurun_out/r3f/new_$i.json 2>/dev/null || exit 1
LSSVC_HIP_LIB=$OLD timeout -k 10 300 $B > gpurun_out/r3f/old_$i.json 2>/dev/null || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3f/*_?.json')):
    d=json.load(open(f)); print(f, d['value'], d['ms_per_step'], d['bpp_check'])
PY
