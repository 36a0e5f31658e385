#!/bin/bash
# round 6, VERDICT r5 item 6: two independent GOPs per GPU -- as two PROCESSES on the one card (the upper bound of any in-process scheme:
# two hipGraphs launched from one process do not overlap the way two processes' do, docs/LAB_NOTEBOOK.md) -- against one. Gate: >= 6 %.
OUT=gpurun_out/r6
mkdir -p $OUT
B="--steps 8 --warmup 2 --no-cpu-baseline --no-side-configs --no-parity-pass --no-h2d-pass --no-events"
get() { python -c "import sys,json; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('%.3f' % d['value'])" $1; }
: > $OUT/two_gops_ab.txt
for i in 1 2; do
  timeout -k 10 300 python bench.py $B > $OUT/one.json 2>/dev/null || exit 1
  echo "one process           $(get $OUT/one.json) frames/s" >> $OUT/two_gops_ab.txt
  timeout -k 10 400 python bench.py $B > $OUT/twoA.json 2>/dev/null &
  PA=$!
  timeout -k 10 400 python bench.py $B > $OUT/twoB.json 2>/dev/null &
  PB=$!
  wait $PA || exit 1
  wait $PB || exit 1
  echo "two processes at once $(get $OUT/twoA.json) + $(get $OUT/twoB.json) frames/s" >> $OUT/two_gops_ab.txt
done
cat $OUT/two_gops_ab.txt
