#!/bin/bash
set -x
mkdir -p gpurun_out/r3d
timeout -k 10 1200 python -m pytest tests/test_gpu_stream.py tests/test_gpu_harness.py tests/test_gpu_symbols.py tests/test_gpu_shard.py "tests/test_gpu_ops.py::test_fp16_range_audit_moves_saturating_layers_to_fp32" -q -s > gpurun_out/r3d/pytest.log 2>&1
rc=$?
echo "pytest rc=$rc" >> gpurun_out/r3d/pytest.log
grep -E "passed|failed|^E  |rc=|stream bits" gpurun_out/r3d/pytest.log | tail -40
[ $rc -eq 0 ] || exit 1
timeout -k 10 900 python bench.py > gpurun_out/r3d/bench.json 2> gpurun_out/r3d/bench.log || { tail -30 gpurun_out/r3d/bench.log; exit 1; }
tail -5 gpurun_out/r3d/bench.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3d/bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("frac_issued_fp16"), d["roofline"].get("frac_algorithmic_fp16"))
print(json.dumps(d.get("config4_stream"), indent=1))
print(json.dumps(d.get("config3_2160p"), indent=1))
print(json.dumps(d.get("cpu_baseline"), indent=1)[:600])
PY
