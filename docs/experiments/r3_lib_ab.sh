set -x
mkdir -p gpurun_out/r3f; rm -f gpurun_out/r3f/*_?.json
timeout -k 10 600 python -m pytest tests/test_gpu_bench_kernels.py -m gpu -q -x > gpurun_out/r3f/pytest5.log 2>&1; rc=$?; tail -3 gpurun_out/r3f/pytest5.log; [ $rc -eq 0 ] || exit 1
OLD=$PWD/lssvc_amd/lib/liblssvc_hip_old.so
for i in 1 2; do
timeout -k 10 300 python tools/p3_ab.py 3 10 > gpurun_out/r3f/p3_new_$i.txt 2>&1 || exit 1
LSSVC_HIP_LIB=$OLD timeout -k 10 300 python tools/p3_ab.py 3 10 > gpurun_out/r3f/p3_old_$i.txt 2>&1 || exit 1
done
paste -d'\n' gpurun_out/r3f/p3_new_2.txt gpurun_out/r3f/p3_old_2.txt | grep prodcons | cut -c1-64
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
