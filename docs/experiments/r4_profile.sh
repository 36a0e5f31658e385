#!/bin/bash
# round-4 profile passes of ONE bench command (single stream, so that a kernel's duration is its own): kernel trace, FETCH_SIZE,
# WRITE_SIZE, SQ counters -- separate runs, the program directly after "--" -- summarised on the box into profiles/r04_* and copied
# to gpurun_out/r4p/ (the raw CSVs are too big to merge). Then the default (multi-stream) run's kernel trace for the overlap report.
set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4p
RAW=/tmp/r4p_raw
mkdir -p $OUT $RAW
B="--steps 1 --warmup 1 --no-cpu-baseline --no-h2d-pass --no-events --no-side-configs --no-streams"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $RAW/stats -- python3 $GRAFT_REPO_ROOT/bench.py $B > $OUT/bench_under_trace.json 2> $OUT/stats.log || exit 1
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $RAW/fetch -- python3 $GRAFT_REPO_ROOT/bench.py $B --no-graph > $OUT/fetch.json 2> $OUT/fetch.log || exit 1
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $RAW/write -- python3 $GRAFT_REPO_ROOT/bench.py $B --no-graph > $OUT/write.json 2> $OUT/write.log || exit 1
timeout -k 10 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $RAW/sq -- python3 $GRAFT_REPO_ROOT/bench.py $B --no-graph > $OUT/sq.json 2> $OUT/sq.log || exit 1
cd $GRAFT_REPO_ROOT
python tools/prof_summarize.py r04 --stats $RAW/stats --fetch $RAW/fetch --write $RAW/write --sq $RAW/sq \
  --cmd "rocprofv3 [--pmc ...] --kernel-trace -- python3 bench.py $B [--no-graph for the --pmc passes]" > $OUT/summarize.log 2>&1
python tools/overlap_report.py $(find $RAW/stats -name "*kernel_trace.csv" | head -1) > $OUT/r04_overlap_bench_single_stream.txt 2>&1
cp profiles/r04_bench_kernel_stats.csv profiles/r04_pmc_traffic.json profiles/r04_mfma_busy.json $OUT/
cat $OUT/summarize.log; head -14 $OUT/r04_bench_kernel_stats.csv; cat $OUT/r04_overlap_bench_single_stream.txt
rm -rf $RAW
