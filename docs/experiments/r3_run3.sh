#!/bin/bash
# round-3 GPU call 3: the whole GPU suite, then the default bench (with the new side measurements), then the CPU baseline at full size
set -x
mkdir -p gpurun_out/r3c
timeout -k 10 1500 python -m pytest tests -m gpu -q -s > gpurun_out/r3c/pytest.log 2>&1
rc=$?
echo "pytest rc=$rc" >> gpurun_out/r3c/pytest.log
grep -E "passed|failed|error|rc=|entries differ|stream bits|frame [0-9]:" gpurun_out/r3c/pytest.log | tail -60
[ $rc -eq 0 ] || exit 1
timeout -k 10 900 python bench.py > gpurun_out/r3c/bench.json 2> gpurun_out/r3c/bench.log || { tail -30 gpurun_out/r3c/bench.log; exit 1; }
tail -5 gpurun_out/r3c/bench.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3c/bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("frac_issued_fp16"), d["roofline"].get("frac_algorithmic_fp16"))
print(json.dumps(d.get("config4_stream"), indent=1))
print(json.dumps(d.get("config3_2160p"), indent=1))
print(json.dumps(d.get("cpu_baseline"), indent=1)[:600])
PY
