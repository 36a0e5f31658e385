#!/bin/bash
# two bench processes on ONE GPU at the same time vs one: how much idle time do two independent GOP chains fill?
set -x
mkdir -p gpurun_out/r3f
B="--no-cpu-baseline --no-h2d-pass --no-events --no-side-configs --steps 4 --warmup 1"
export LSSVC_RESERVE_GIB=4
timeout -k 10 300 python bench.py $B > gpurun_out/r3f/one.json 2> gpurun_out/r3f/one.log || exit 1
(timeout -k 10 400 python bench.py $B > gpurun_out/r3f/two_a.json 2> gpurun_out/r3f/two_a.log) &
(timeout -k 10 400 python bench.py $B > gpurun_out/r3f/two_b.json 2> gpurun_out/r3f/two_b.log) &
wait
python - <<'PY'
import json
for n in ("one","two_a","two_b"):
    d=json.load(open("gpurun_out/r3f/%s.json"%n)); print(n, d["value"], d["ms_per_step"])
PY
