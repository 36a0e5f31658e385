#!/bin/bash
# round-5 profile passes of ONE bench command (single stream, so that a kernel's duration is its own): kernel trace, FETCH_SIZE,
# WRITE_SIZE, SQ counters -- separate runs, the program directly after "--" -- summarised on the box into profiles/r05_* and copied
# to gpurun_out/r5p/ (the raw CSVs are too big to merge). Then the default (multi-stream) run's kernel trace for the overlap report.
set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5p
RAW=/tmp/r5p_raw
mkdir -p $OUT $RAW
B="--steps 1 --warmup 1 --no-cpu-baseline --no-h2d-pass --no-events --no-side-configs --no-streams"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $RAW/stats -- python3 $GRAFT_REPO_ROOT/bench.py $B > $OUT/bench_under_trace.json 2> $OUT/stats.log || exit 1
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $RAW/fetch -- python3 $GRAFT_REPO_ROOT/bench.py $B --no-graph > $OUT/fetch.json 2> $OUT/fetch.log || exit 1
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $RAW/write -- python3 $GRAFT_REPO_ROOT/bench.py $B --no-graph > $OUT/write.json 2> $OUT/write.log || exit 1
timeout -k 10 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $RAW/sq -- python3 $GRAFT_REPO_ROOT/bench.py $B --no-graph > $OUT/sq.json 2> $OUT/sq.log || exit 1
cd $GRAFT_REPO_ROOT
python tools/prof_summarize.py r05 --stats $RAW/stats --fetch $RAW/fetch --write $RAW/write --sq $RAW/sq \
  --cmd "rocprofv3 [--pmc ...] --kernel-trace -- python3 bench.py $B [--no-graph for the --pmc passes]" > $OUT/summarize.log 2>&1
python tools/overlap_report.py $(find $RAW/stats -name "*kernel_trace.csv" | head -1) > $OUT/r05_overlap_bench_single_stream.txt 2>&1
cp profiles/r05_bench_kernel_stats.csv profiles/r05_pmc_traffic.json profiles/r05_mfma_busy.json $OUT/
cat $OUT/summarize.log; head -14 $OUT/r05_bench_kernel_stats.csv; cat $OUT/r05_overlap_bench_single_stream.txt
rm -rf $RAW
# round 5 additions, same box, same code: the in-kernel clock of the dominant kernel (stamp build), the per-signature table of the
# 1080p bench (no side configs: they would overwrite it with the 2160p one), the plan histogram, the clock probe
cd $GRAFT_REPO_ROOT
LSSVC_CONV_DEBUG=256 timeout -k 10 200 python tools/p3_stamps.py --json profiles/r05_p3_stamps.json 2>&1 | grep -v amdgpu.ids > profiles/r05_p3_stamps.txt || exit 1
LSSVC_BENCH_SIGNATURES=profiles/r05_bench_signatures.txt timeout -k 10 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-h2d-pass --no-side-configs > $OUT/bench_signatures.json 2> $OUT/bench_signatures.log || exit 1
timeout -k 10 200 python tools/plan_histogram.py > profiles/r05_plan_histogram.txt 2>&1 || exit 1
[ -x tools/probes/clock_probe.bin ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/probes/clock_probe.hip -o tools/probes/clock_probe.bin || exit 1
timeout -k 10 120 tools/probes/clock_probe.bin 2.5 > profiles/r05_clock_probe.txt 2>&1 || exit 1
cp profiles/r05_p3_stamps.json profiles/r05_p3_stamps.txt profiles/r05_bench_signatures.txt profiles/r05_plan_histogram.txt profiles/r05_clock_probe.txt profiles/r05_overlap_bench_single_stream.txt $OUT/ 2>/dev/null
cat profiles/r05_p3_stamps.txt; head -5 profiles/r05_bench_signatures.txt; tail -5 profiles/r05_plan_histogram.txt
