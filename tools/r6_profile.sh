#!/bin/bash
# round-6 profile passes of ONE bench command (single stream, so that a kernel's duration is its own): kernel trace, FETCH_SIZE,
# WRITE_SIZE, SQ counters -- separate runs, the program directly after "--" -- summarised on the box into profiles/r06_* and copied
# to gpurun_out/r6p/ (the raw CSVs are too big to merge). Then the default (multi-stream) run's kernel trace for the overlap report.
set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6p
RAW=/tmp/r6p_raw
mkdir -p $OUT $RAW
B="--steps 1 --warmup 1 --no-cpu-baseline --no-h2d-pass --no-events --no-side-configs --no-streams --resident-headline --no-parity-pass"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $RAW/stats -- python3 $GRAFT_REPO_ROOT/bench.py $B > $OUT/bench_under_trace.json 2> $OUT/stats.log || exit 1
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $RAW/fetch -- python3 $GRAFT_REPO_ROOT/bench.py $B --no-graph > $OUT/fetch.json 2> $OUT/fetch.log || exit 1
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $RAW/write -- python3 $GRAFT_REPO_ROOT/bench.py $B --no-graph > $OUT/write.json 2> $OUT/write.log || exit 1
timeout -k 10 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $RAW/sq -- python3 $GRAFT_REPO_ROOT/bench.py $B --no-graph > $OUT/sq.json 2> $OUT/sq.log || exit 1
cd $GRAFT_REPO_ROOT
python tools/prof_summarize.py r06 --stats $RAW/stats --fetch $RAW/fetch --write $RAW/write --sq $RAW/sq \
  --cmd "rocprofv3 [--pmc ...] --kernel-trace -- python3 bench.py $B [--no-graph for the --pmc passes]" > $OUT/summarize.log 2>&1
python tools/overlap_report.py $(find $RAW/stats -name "*kernel_trace.csv" | head -1) > $OUT/r06_overlap_bench_single_stream.txt 2>&1
cp profiles/r06_bench_kernel_stats.csv profiles/r06_pmc_traffic.json profiles/r06_mfma_busy.json $OUT/
cat $OUT/summarize.log; head -14 $OUT/r06_bench_kernel_stats.csv; cat $OUT/r06_overlap_bench_single_stream.txt
rm -rf $RAW
# round 5 additions, same box, same code: the in-kernel clock of the dominant kernel (stamp build), the per-signature table of the
# 1080p bench (no side configs: they would overwrite it with the 2160p one), the plan histogram, the clock probe
cd $GRAFT_REPO_ROOT
LSSVC_CONV_DEBUG=256 timeout -k 10 200 python tools/p3_stamps.py --json profiles/r06_p3_stamps.json 2>&1 | grep -v amdgpu.ids > profiles/r06_p3_stamps.txt || exit 1
LSSVC_BENCH_SIGNATURES=profiles/r06_bench_signatures.txt timeout -k 10 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-h2d-pass --no-side-configs --resident-headline --no-parity-pass > $OUT/bench_signatures.json 2> $OUT/bench_signatures.log || exit 1
timeout -k 10 200 python tools/plan_histogram.py > profiles/r06_plan_histogram.txt 2>&1 || exit 1
[ -x tools/probes/clock_probe.bin ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/probes/clock_probe.hip -o tools/probes/clock_probe.bin || exit 1
timeout -k 10 120 tools/probes/clock_probe.bin 2.5 > profiles/r06_clock_probe.txt 2>&1 || exit 1
cp profiles/r06_p3_stamps.json profiles/r06_p3_stamps.txt profiles/r06_bench_signatures.txt profiles/r06_plan_histogram.txt profiles/r06_clock_probe.txt profiles/r06_overlap_bench_single_stream.txt $OUT/ 2>/dev/null
cat profiles/r06_p3_stamps.txt; head -5 profiles/r06_bench_signatures.txt; tail -5 profiles/r06_plan_histogram.txt
# the driver's command on the SAME box (live HIP events: roofline.avg_launch_us is held against r06_bench_kernel_stats.csv's average for the same kernel)
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.log || exit 1
tail -1 $OUT/bench_default.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('bench: %.2f frames/s; %s: %d launches, %.1f us on average (live events), frac %.4f' % (d['value'], r['kernel'], r['launches'], r['avg_launch_us'], r['frac']))"
grep -m1 "conv3_f16x3p_kernel<4, false, false, false, 1, false, 0, 0" profiles/r06_bench_kernel_stats.csv
