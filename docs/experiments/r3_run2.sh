#!/bin/bash
set -x
mkdir -p gpurun_out/r3b
timeout -k 10 900 python -m pytest tests/test_gpu_golden_full.py -q -s > gpurun_out/r3b/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3b/pytest.log
grep -E "frame [0-9]:|flipped|passed|failed|rc=" gpurun_out/r3b/pytest.log | tail -30
B="--no-cpu-baseline --no-h2d-pass --no-events --steps 1 --warmup 1"
timeout -k 10 300 python bench.py $B --no-graph --no-streams > gpurun_out/r3b/eager_single.json 2> gpurun_out/r3b/eager_single.log || exit 1
timeout -k 10 300 python bench.py $B --no-graph > gpurun_out/r3b/eager_multi.json 2> gpurun_out/r3b/eager_multi.log || exit 1
cd /tmp && export TMPDIR=/tmp
for mode in single multi; do
  extra=""; [ $mode = single ] && extra="--no-streams"
  timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3b/trace_$mode -- python3 $GRAFT_REPO_ROOT/bench.py $B --frames 8 $extra > $GRAFT_REPO_ROOT/gpurun_out/r3b/trace_$mode.json 2> $GRAFT_REPO_ROOT/gpurun_out/r3b/trace_$mode.log || exit 1
done
cd $GRAFT_REPO_ROOT
for mode in single multi; do
  f=$(find gpurun_out/r3b/trace_$mode -name "*kernel_trace.csv" | head -1)
  echo "== $mode $f"; python tools/overlap_report.py $f
done
python - <<'PY'
import json
for n in ("eager_single","eager_multi","trace_single","trace_multi"):
    d=json.load(open("gpurun_out/r3b/%s.json"%n))
    print(n, d["value"], d["ms_per_step"])
PY
