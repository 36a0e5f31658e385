"""The multi-GPU plumbing over RCCL on the one GPU a test box has: a world of ONE rank on backend "nccl" (= RCCL on ROCm)
runs the device-side checkpoint broadcast / unpack, the gather of per-frame records, the max-reduce of the timing and the
sharded job loop of lssvc_amd/shard.py -- the code `python -m torch.distributed.run ... test.py` and `bench.py --gpus N`
execute on every rank. (world_size 2 runs of the same code are the gloo tests in tests/test_shard_gloo.py; a scaling curve
needs the 8-GPU node and is the driver's measurement.)"""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def rccl_world_of_one():
    import torch.distributed as dist
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    torch.cuda.set_device(DEV)
    dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%d" % port, world_size=1, rank=0, device_id=torch.device(DEV))
    yield dist
    dist.destroy_process_group()


def test_checkpoint_broadcast_over_rccl(rccl_world_of_one):
    from lssvc_amd import shard
    from lssvc_amd.synth import synth_state_dict
    dist = rccl_world_of_one
    assert dist.get_backend() == "nccl"
    calls = []

    def loader(name):
        calls.append(name)
        sd = dict(synth_state_dict(name, 1, 0.6))
        sd["epoch"] = 17                                   # non-tensor entries of a published checkpoint
        sd["note"] = {"arch": "lssvc", "q": 3}
        return sd

    got = shard.broadcast_state_dicts(["intra_ss", "lssvc_extend"], dist, device=DEV, loader=loader, force=True)
    assert calls == ["intra_ss", "lssvc_extend"]            # rank 0 loaded each once
    for name in ("intra_ss", "lssvc_extend"):
        want = synth_state_dict(name, 1, 0.6)
        assert set(got[name]) == set(want) | {"epoch", "note"}
        assert got[name]["epoch"] == 17 and got[name]["note"] == {"arch": "lssvc", "q": 3}
        for k, v in want.items():
            g = got[name][k]
            assert g.dtype == v.dtype and g.shape == v.shape and torch.equal(g, v), k
    # the broadcast weights drive the codec: an I-frame from them equals one from locally drawn weights, bit for bit
    from lssvc_amd import IntraSS
    x_el = torch.rand(1, 3, 128, 128, device=DEV)
    x_bl = torch.rand(1, 3, 64, 64, device=DEV)
    outs = []
    for sd in (got["intra_ss"], synth_state_dict("intra_ss", 1, 0.6)):
        sd = {k: v for k, v in sd.items() if torch.is_tensor(v)}
        net = IntraSS.from_state_dict(sd).to(DEV).eval()
        net.set_scale_information(2.0, (128, 128), (0, 0, 0, 0))
        outs.append(net.encode_decode(x_bl, x_el, None, None))
    assert outs[0]["bit_el"] == outs[1]["bit_el"] and torch.equal(outs[0]["x_hat_el"], outs[1]["x_hat_el"])


def test_gather_reduce_and_sharded_loop_over_rccl(rccl_world_of_one):
    from lssvc_amd import shard
    dist = rccl_world_of_one
    recs = {3: {"frame": 3, "bits": 1.5}, 1: {"frame": 1, "bits": 2.5}}
    assert shard.gather_frame_records(recs, dist) == [recs[1], recs[3]]
    assert shard.max_over_ranks(1.25, dist, device=DEV) == 1.25
    assert shard.run_sharded([1, 2, 3], lambda u: u * u, dist) == [1, 4, 9]

    def flaky(u):
        if u == 2:
            raise ValueError("sequence ends before frame 7")
        return u

    # world of one takes the in-process path, where the exception propagates as it is
    with pytest.raises(ValueError):
        shard.run_sharded([1, 2, 3], flaky, dist)


def test_bench_two_ranks_rehearsal():
    """`python bench.py --gpus 2` with no launcher in front, on the ONE card of a test box (LSSVC_BENCH_REHEARSAL=1: both ranks
    on cuda:0, collectives over gloo): bench.py starts its own two rank processes, rank 0's checkpoints are broadcast, each rank
    codes its own short GOP through the HIP path, the timing is max-reduced and ONE line comes back with n_gpus 2. A rehearsal of
    the flow the driver's 8-GPU run takes, not a measurement (the line says so itself)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LSSVC_BENCH_REHEARSAL="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--frames", "3",
                        "--no-cpu-baseline", "--no-side-configs", "--no-h2d-pass", "--no-events"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    line = lines[0]
    assert line["n_gpus"] == 2 and line["steps"] == 1 and line["scaling"] == "weak"
    assert line["value"] > 0 and "REHEARSAL" in line["config"]["parallelism"]
    assert abs(line["value"] - 2 * 3 / (line["ms_per_step"] * 1e-3)) < 1e-2 * line["value"]      # whole-job frames over the max-over-ranks time
