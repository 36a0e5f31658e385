#!/bin/bash
# round-3 GPU call 1: stream bit-identity + full-size goldens, then single- vs multi-stream bench
set -x
mkdir -p gpurun_out/r3a
timeout -k 10 900 python -m pytest tests/test_gpu_graph.py tests/test_gpu_golden_full.py -x -q -s > gpurun_out/r3a/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3a/pytest.log
tail -25 gpurun_out/r3a/pytest.log
timeout -k 10 300 python bench.py --no-cpu-baseline --no-h2d-pass --no-streams > gpurun_out/r3a/bench_single.json 2> gpurun_out/r3a/bench_single.log || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --no-h2d-pass > gpurun_out/r3a/bench_multi.json 2> gpurun_out/r3a/bench_multi.log || exit 1
python - <<'PY'
import json
for n in ("single","multi"):
    d=json.load(open("gpurun_out/r3a/bench_%s.json"%n))
    print(n, d["value"], d["ms_per_step"], d["config"]["launch"])
PY
