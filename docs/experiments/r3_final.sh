#!/bin/bash
# final round-3 check: whole GPU suite, smoke, torchrun launch of the bench with one rank, default bench
set -x
mkdir -p gpurun_out/r3z
timeout -k 10 1500 python -m pytest tests -m gpu -q > gpurun_out/r3z/pytest.log 2>&1
rc=$?
echo "pytest rc=$rc" >> gpurun_out/r3z/pytest.log
tail -4 gpurun_out/r3z/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3z/smoke.log 2>&1 || { tail -20 gpurun_out/r3z/smoke.log; exit 1; }
tail -1 gpurun_out/r3z/smoke.log
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 1 --warmup 1 --no-cpu-baseline --no-h2d-pass --no-side-configs > gpurun_out/r3z/bench_torchrun.json 2> gpurun_out/r3z/bench_torchrun.log || { tail -20 gpurun_out/r3z/bench_torchrun.log; exit 1; }
python -c "import json; d=json.load(open('gpurun_out/r3z/bench_torchrun.json')); print('torchrun n=1:', d['value'], d['config']['parallelism'])"
timeout -k 10 900 python bench.py > gpurun_out/r3z/bench.json 2> gpurun_out/r3z/bench.log || { tail -30 gpurun_out/r3z/bench.log; exit 1; }
python -c "import json; d=json.load(open('gpurun_out/r3z/bench.json')); print('bench:', d['value'], d['ms_per_step'], d['roofline']['frac'], d['config4_stream']['encoding_time_s_per_p_frame'], d['config3_2160p']['value'])"
