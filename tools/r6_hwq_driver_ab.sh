OUT=gpurun_out/r6
mkdir -p $OUT
B="--steps 8 --warmup 3 --no-cpu-baseline --no-side-configs"
: > $OUT/hwq_driver.txt
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 h2d-inclusive %.3f frames/s  resident %.3f  bits sha1 %s' % (d['value'], d['resident']['value'], d['ranks'][0]['bits_sha1'][:12]))"; }
for i in 1 2; do
  GPU_MAX_HW_QUEUES=4 timeout -k 10 300 python bench.py $B 2>/dev/null | line "GPU_MAX_HW_QUEUES=4 (runtime default)" >> $OUT/hwq_driver.txt || exit 1
  timeout -k 10 300 python bench.py $B 2>/dev/null | line "package default (8)                  " >> $OUT/hwq_driver.txt || exit 1
done
cat $OUT/hwq_driver.txt
