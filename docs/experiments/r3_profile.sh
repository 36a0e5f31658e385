#!/bin/bash
# round-3 profile passes of ONE bench command: kernel trace, FETCH_SIZE, WRITE_SIZE, SQ counters (separate runs, program
# directly after --), summarised on the box into profiles/r03_* and copied to gpurun_out/r3p/ (raw CSVs are too big to merge)
set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3p
RAW=/tmp/r3p_raw
mkdir -p $OUT $RAW
B="--steps 1 --warmup 1 --no-cpu-baseline --no-h2d-pass --no-events --no-side-configs"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $RAW/stats -- python3 $GRAFT_REPO_ROOT/bench.py $B > $OUT/bench_under_trace.json 2> $OUT/stats.log || exit 1
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $RAW/fetch -- python3 $GRAFT_REPO_ROOT/bench.py $B --no-graph > $OUT/fetch.json 2> $OUT/fetch.log || exit 1
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $RAW/write -- python3 $GRAFT_REPO_ROOT/bench.py $B --no-graph > $OUT/write.json 2> $OUT/write.log || exit 1
timeout -k 10 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $RAW/sq -- python3 $GRAFT_REPO_ROOT/bench.py $B --no-graph > $OUT/sq.json 2> $OUT/sq.log || exit 1
cd $GRAFT_REPO_ROOT
python tools/prof_summarize.py r03 --stats $RAW/stats --fetch $RAW/fetch --write $RAW/write --sq $RAW/sq \
  --cmd "rocprofv3 [--pmc ...] --kernel-trace -- python3 bench.py $B [--no-graph for the --pmc passes]" > $OUT/summarize.log 2>&1
python tools/overlap_report.py $(find $RAW/stats -name "*kernel_trace.csv" | head -1) > $OUT/r03_overlap_bench.txt 2>&1
cp profiles/r03_bench_kernel_stats.csv profiles/r03_pmc_traffic.json profiles/r03_mfma_busy.json $OUT/
cat $OUT/summarize.log; head -12 $OUT/r03_bench_kernel_stats.csv; cat $OUT/r03_overlap_bench.txt
rm -rf $RAW
