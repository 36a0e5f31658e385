#!/bin/bash
# round 6: what the four kernel families' new instantiations are worth at the FRAME level -- the bench's GOP with round 5's dispatch
# (LSSVC_P3_SMALL=0 LSSVC_P3_NARROW=0 LSSVC_P3_PF2=0 LSSVC_GDN_FAST_OPT=0), with round 6's kernels before the split-roles schedule
# (LSSVC_P3_PF2=4 LSSVC_P3_BIG_PAIR=3: the register prefetch everywhere) and with the defaults, same box, interleaved.
OUT=gpurun_out/r6
mkdir -p $OUT
B="--steps 6 --warmup 2 --no-cpu-baseline --no-side-configs --no-parity-pass --no-h2d-pass --no-events --resident-headline"
echo "# tools/r6_frame_ab.sh: the bench's GOP (resident inputs, 6 timed GOPs) with round 5's dispatch (LSSVC_P3_SMALL=0 LSSVC_P3_NARROW=0 LSSVC_P3_PF2=0 LSSVC_GDN_FAST_OPT=0), with round 6's kernels before the split-roles schedule (LSSVC_P3_PF2=4 LSSVC_P3_BIG_PAIR=3) and with the defaults; same box, interleaved" > $OUT/frame_ab.txt
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 %.3f frames/s  %.1f ms/GOP  bits sha1 %s' % (d['value'], d['ms_per_step'], d['ranks'][0]['bits_sha1'][:12]))"; }
for i in 1 2 3; do
  LSSVC_P3_SMALL=0 LSSVC_P3_NARROW=0 LSSVC_P3_PF2=0 LSSVC_GDN_FAST_OPT=0 timeout -k 10 300 python bench.py $B 2>/dev/null | line "round-5 dispatch          " >> $OUT/frame_ab.txt || exit 1
  LSSVC_P3_PF2=4 LSSVC_P3_BIG_PAIR=3 timeout -k 10 300 python bench.py $B 2>/dev/null | line "round 6, register prefetch" >> $OUT/frame_ab.txt || exit 1
  timeout -k 10 300 python bench.py $B 2>/dev/null | line "round 6, defaults (roles) " >> $OUT/frame_ab.txt || exit 1
done
cat $OUT/frame_ab.txt
