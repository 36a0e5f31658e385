"""The N>1 path on CPU: two gloo ranks shard GOPs, code them with a stand-in per-frame function, and
merge the per-frame records; the merged result must equal the single-process result."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lssvc_amd.shard import split_gops, assign, gather_frame_records, max_over_ranks


def _fake_code_gop(first, n):
    """Stand-in for the closed-loop GOP coder: each frame depends on the previous one inside the GOP only."""
    out, state = {}, 0.0
    for t in range(n):
        state = 0.5 * state + (first + t + 1)           # chain inside the GOP, reset at every I-frame
        out[first + t] = {"frame": first + t, "type": 0 if t == 0 else 1, "bits": state}
    return out


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    local = {}
    for first, n in assign(split_gops(70, 32), world, rank):
        local.update(_fake_code_gop(first, n))
    merged = gather_frame_records(local, dist)
    slow = max_over_ranks(1.0 + rank, dist)
    if rank == 0:
        q.put((merged, slow))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one():
    assert split_gops(70, 32) == [(0, 32), (32, 32), (64, 6)]
    assert assign(list(range(5)), 2, 0) == [0, 2, 4] and assign(list(range(5)), 2, 1) == [1, 3]
    single = {}
    for first, n in split_gops(70, 32):
        single.update(_fake_code_gop(first, n))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29000 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    merged, slow = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert merged == gather_frame_records(single) and len(merged) == 70
    assert slow == 2.0
