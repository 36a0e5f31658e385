#!/bin/bash
# round 6: GPU_MAX_HW_QUEUES (the HIP runtime's hardware queues per process, default 4) against the frame plans' streams: the bench's GOP
# (resident inputs, 6 timed GOPs) at several settings, same box, interleaved.
OUT=gpurun_out/r6
mkdir -p $OUT
B="--steps 6 --warmup 2 --no-cpu-baseline --no-side-configs --no-parity-pass --no-h2d-pass --no-events --resident-headline"
echo "# tools/r6_hwq_ab.sh: python bench.py $B under GPU_MAX_HW_QUEUES=n (unset = the runtime's default, 4); same box, interleaved" > $OUT/hwq_ab.txt
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 %.3f frames/s  %.1f ms/GOP  bits sha1 %s' % (d['value'], d['ms_per_step'], d['ranks'][0]['bits_sha1'][:12]))"; }
for i in 1 2; do
  timeout -k 10 300 python bench.py $B 2>$OUT/hwq_err.log | line "default              " >> $OUT/hwq_ab.txt || exit 1
  for n in ${QUEUES:-8 16}; do
    GPU_MAX_HW_QUEUES=$n timeout -k 10 300 python bench.py $B 2>$OUT/hwq_err_$n.log | line "GPU_MAX_HW_QUEUES=$n " >> $OUT/hwq_ab.txt || exit 1
  done
done
cat $OUT/hwq_ab.txt
