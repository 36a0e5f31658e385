#!/bin/bash
set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3q
RAW=/tmp/r3q_raw
mkdir -p $OUT $RAW
B="--steps 1 --warmup 1 --no-cpu-baseline --no-h2d-pass --no-events --no-side-configs --no-streams"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $RAW/stats -- python3 $GRAFT_REPO_ROOT/bench.py $B > $OUT/bench_under_trace.json 2> $OUT/stats.log || exit 1
cd $GRAFT_REPO_ROOT
python tools/prof_summarize.py r03 --stats $RAW/stats --cmd "rocprofv3 --kernel-trace -- python3 bench.py $B" > $OUT/summarize.log 2>&1
python tools/overlap_report.py $(find $RAW/stats -name "*kernel_trace.csv" | head -1) > $OUT/r03_overlap_bench.txt 2>&1
cp profiles/r03_bench_kernel_stats.csv $OUT/
head -8 $OUT/r03_bench_kernel_stats.csv; cat $OUT/r03_overlap_bench.txt; cat $OUT/bench_under_trace.json | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'])"
rm -rf $RAW
