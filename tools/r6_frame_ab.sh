#!/bin/bash
# round 6: what the four kernel families' new instantiations are worth at the FRAME level -- the bench's GOP with round 5's dispatch
# (LSSVC_P3_SMALL=0 LSSVC_P3_NARROW=0 LSSVC_P3_PF2=0 LSSVC_GDN_FAST_OPT=0) against the defaults, same box, interleaved.
OUT=gpurun_out/r6
mkdir -p $OUT
B="--steps 6 --warmup 2 --no-cpu-baseline --no-side-configs --no-parity-pass --no-h2d-pass --no-events"
: > $OUT/frame_ab.txt
for i in 1 2 3; do
  LSSVC_P3_SMALL=0 LSSVC_P3_NARROW=0 LSSVC_P3_PF2=0 LSSVC_GDN_FAST_OPT=0 timeout -k 10 300 python bench.py $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round-5 dispatch   %.3f frames/s  %.1f ms/GOP  bits sha1 %s' % (d['value'], d['ms_per_step'], d['ranks'][0]['bits_sha1'][:12]))" >> $OUT/frame_ab.txt || exit 1
  timeout -k 10 300 python bench.py $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round-6 kernels    %.3f frames/s  %.1f ms/GOP  bits sha1 %s' % (d['value'], d['ms_per_step'], d['ranks'][0]['bits_sha1'][:12]))" >> $OUT/frame_ab.txt || exit 1
done
cat $OUT/frame_ab.txt
