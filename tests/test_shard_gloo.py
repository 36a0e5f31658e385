"""The N>1 path on CPU: two gloo ranks shard GOPs, code them with a stand-in per-frame function, and
merge the per-frame records; the merged result must equal the single-process result."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lssvc_amd.shard import split_gops, assign, gather_frame_records, max_over_ranks, broadcast_state_dicts, run_sharded


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


# ---- the harness's torchrun mode: checkpoint broadcast from rank 0 + job sharding + end-of-run gather -----------------
def _ckpt(seed):
    g = torch.Generator().manual_seed(seed)
    return {"a.weight": torch.randn(5, 3, 3, 3, generator=g), "a.bias": torch.randn(5, generator=g), "empty": torch.zeros(0),
            "idx": torch.arange(7, dtype=torch.int32), "half": torch.randn(3, 2, generator=g).half(), "scalar": torch.tensor(2.5)}


def _harness_worker(rank, world, port, q, cfg_dir):
    import json
    import types
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lssvc_amd import harness as H

    def loader(path):                                   # only rank 0 may touch the "file system"
        assert dist.get_rank() == 0, "rank %d tried to read %s" % (dist.get_rank(), path)
        return {"i.pth": _ckpt(1), "p.pth": _ckpt(2)}[os.path.basename(path)]

    sds = broadcast_state_dicts(["i.pth", "p.pth", "i.pth"], dist, "cpu", loader)
    ok = all(torch.equal(sds[p][k], v) and sds[p][k].dtype == v.dtype for p, want in (("i.pth", _ckpt(1)), ("p.pth", _ckpt(2)))
             for k, v in want.items())
    cfg = json.load(open(os.path.join(cfg_dir, "cfg.json")))
    args = H.parse_args(["--i_frame_model_path", "i.pth", "--model_path", "p.pth", "--test_config", os.path.join(cfg_dir, "cfg.json"),
                         "--cuda", "1", "--output_path", cfg_dir])
    jobs = H.build_jobs(args, cfg)

    def fake_run(job):                                  # stand-in for run_job: closed loop inside the GOP only
        import time
        time.sleep(0.05)                                # (the work queue of round 6: a job takes time, or one rank could take them all)
        recs, state = [], 0.0
        for t in range(job["count"]):
            f = job["first"] + t
            state = 0.5 * state + f + 1
            recs.append({"frame": f, "type": 0 if t == 0 else 1, "bits_bl": state, "bits_el": 2 * state, "rgb_psnr_bl": 30.0 + f,
                         "rgb_psnr_el": 31.0 + f, "yuv_bl": (30.0, 31.0, 32.0), "yuv_el": (33.0, 34.0, 35.0), "enc_bl": 0.0,
                         "dec_bl": 0.0, "enc_el": 0.0, "dec_el": 0.0, "rank": dist.get_rank()})
        return {"key": (job["ds_name"], job["ratio"], job["seq"], job["model_idx"]), "records": recs, "seconds": 1.0,
                "pix_bl": 64 * 64, "pix_el": 128 * 128}

    results = run_sharded(jobs, fake_run, dist)
    logs = H.collect(args, cfg, results)
    ranks_used = sorted({r["rank"] for res in results for r in res["records"]})
    if rank == 0:
        q.put((ok, json.dumps(logs, sort_keys=True), ranks_used, len(jobs)))
    dist.barrier()
    dist.destroy_process_group()


def test_harness_jobs_and_checkpoints_across_two_ranks(tmp_path):
    import json
    from lssvc_amd import harness as H
    cfg = {"DS": {"test": 1, "base_path": str(tmp_path), "x1": {"width": 128, "height": 128}, "x2": {"width": 64, "height": 64},
                  "sequences": {"seqA": {"frames": 7, "gop": 2}, "seqB": {"frames": 3, "gop": 2}}}}
    with open(tmp_path / "cfg.json", "w") as f:
        json.dump(cfg, f)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31000 + os.getpid() % 2000
    procs = [ctx.Process(target=_harness_worker, args=(r, 2, port, q, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    ok, logs2, ranks_used, n_jobs = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok and ranks_used == [0, 1] and n_jobs == 6
    # the two-rank result files equal the single-process ones
    args = H.parse_args(["--i_frame_model_path", "i.pth", "--model_path", "p.pth", "--test_config", str(tmp_path / "cfg.json"),
                         "--cuda", "1", "--output_path", str(tmp_path)])

    def fake_run(job):
        recs, state = [], 0.0
        for t in range(job["count"]):
            f = job["first"] + t
            state = 0.5 * state + f + 1
            recs.append({"frame": f, "type": 0 if t == 0 else 1, "bits_bl": state, "bits_el": 2 * state, "rgb_psnr_bl": 30.0 + f,
                         "rgb_psnr_el": 31.0 + f, "yuv_bl": (30.0, 31.0, 32.0), "yuv_el": (33.0, 34.0, 35.0), "enc_bl": 0.0,
                         "dec_bl": 0.0, "enc_el": 0.0, "dec_el": 0.0, "rank": 0})
        return {"key": (job["ds_name"], job["ratio"], job["seq"], job["model_idx"]), "records": recs, "seconds": 1.0,
                "pix_bl": 64 * 64, "pix_el": 128 * 128}

    single = H.collect(args, cfg, run_sharded(H.build_jobs(args, cfg), fake_run))
    a, b = json.loads(logs2), json.loads(json.dumps(single, sort_keys=True))
    for ratio in a:                                    # test_time sums wall seconds per job: identical here (1.0 each)
        assert a[ratio] == b[ratio]


# ---- failure path and non-tensor checkpoint entries -----------------------------------------------------------
def _failing_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def loader(path):
        assert dist.get_rank() == 0
        sd = _ckpt(3)
        sd["epoch"] = 12                                 # plain Python values must not crash rank 0 before the broadcast
        sd["cfg"] = {"q": [1, 2]}
        return sd

    sd = broadcast_state_dicts(["p.pth"], dist, "cpu", loader)["p.pth"]
    meta_ok = sd["epoch"] == 12 and sd["cfg"] == {"q": [1, 2]} and torch.equal(sd["a.bias"], _ckpt(3)["a.bias"])

    def run(u):
        if u == 3:                                       # (round 6: whichever rank the work queue hands unit 3 to)
            raise ValueError("sequence ends before frame 7")
        return u * 10

    try:
        run_sharded(list(range(6)), run, dist)
        msg = None
    except RuntimeError as e:                            # BOTH ranks get here, after the gather: nobody is left waiting
        msg = str(e)
    q.put((rank, meta_ok, msg))
    dist.barrier()
    dist.destroy_process_group()


def test_a_failing_unit_is_reported_on_every_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33000 + os.getpid() % 2000
    procs = [ctx.Process(target=_failing_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, meta_ok, msg in got:
        assert meta_ok
        assert msg is not None and (", unit 3" in msg) and "sequence ends before frame 7" in msg and "1 of 6" in msg


def _queue_worker(rank, world, port, q):
    import time
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    units = [0.05] * 5 + [0.6] + [0.05] * 6             # one long clip among short ones, NOT at the front of the list
    took = []

    def run(u):
        took.append(u)
        time.sleep(u)
        return (u, dist.get_rank())

    t0 = time.time()
    res = run_sharded(units, run, dist, cost=lambda u: u)
    dyn = time.time() - t0
    t0 = time.time()
    res_static = run_sharded(units, run, dist, cost=None, dynamic=False)
    sta = time.time() - t0
    q.put((rank, res, res_static, dyn, sta))
    dist.barrier()
    dist.destroy_process_group()


def test_work_queue_balances_uneven_units():
    """Round 6 ("dynamic GOP work-queue"): run_sharded hands units out on request, most expensive first, through an atomic
    counter in the process group's store. Twelve clips of which one is twelve times as long: every unit runs exactly once, the results
    come back in unit order on both ranks, the rank that took the long clip took nothing else while the other one coded the eleven short
    ones -- and the whole thing ends when the long clip does (0.6 s), where the reference's static `idx % world` split (test.py:648-656)
    gives the long clip's rank five short ones on top (0.85 s)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35000 + os.getpid() % 2000
    procs = [ctx.Process(target=_queue_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, res0, sta0, dyn0, s0), (_, res1, sta1, dyn1, s1) = got
    assert res0 == res1 and sta0 == sta1 and len(res0) == 12
    assert [u for u, _ in res0] == [0.05] * 5 + [0.6] + [0.05] * 6          # unit order, whatever rank ran what
    long_rank = res0[5][1]
    assert all(r != long_rank for i, (_, r) in enumerate(res0) if i != 5), res0      # the long clip's rank took nothing else
    assert [r for _, r in sta0] == [0, 1] * 6                                # the static split, for comparison
    assert max(dyn0, dyn1) < 0.8 < max(s0, s1), (dyn0, dyn1, s0, s1)


def _run_bench(extra_env, *argv):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **extra_env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + list(argv), env=env, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, [json.loads(ln) for ln in lines]


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher in front (how the driver may call it, and what test.py:648-656,685-748 does
    for itself: one worker process per GPU): bench.py starts the two ranks as child processes before it touches HIP, rank 0's
    single JSON line comes back on stdout with n_gpus 2, exit code 0. LSSVC_BENCH_DRYRUN=1 leaves the codec out so that this
    runs on a host without a GPU; the full flow on a GPU is tests/test_gpu_shard.py::test_bench_two_ranks_rehearsal."""
    p, lines = _run_bench({"LSSVC_BENCH_DRYRUN": "1"}, "--gpus", "2", "--steps", "2", "--warmup", "1")
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1, p.stdout
    assert lines[0]["n_gpus"] == 2 and lines[0]["dry_run"] is True and lines[0]["steps"] == 2
    assert lines[0]["checkpoint_tensors_broadcast"] == 334
    assert "starting 2 ranks" in p.stderr
    # round 6: the line is checkable from itself -- one record per rank (its own rate, not only the max-reduce; its
    # process; the CPUs it pinned itself to BEFORE anything else, disjoint from the other rank's), the process group as the library
    # reports it, and every rank's bit counts held against rank 0's
    ranks = lines[0]["ranks"]
    assert [r["rank"] for r in ranks] == [0, 1] and len({r["pid"] for r in ranks}) == 2
    assert all(r["frames_per_s"] > 0 and r["ms_per_step"] > 0 for r in ranks)
    assert ranks[0]["ms_per_step"] < ranks[1]["ms_per_step"] <= lines[0]["ms_per_step"] + 0.5        # (the dry run's rank r sleeps (1 + r) x 10 ms per step; the line quotes the slowest)
    cpus = [_expand(r["cpu_affinity"]) for r in ranks]
    assert cpus[0] and cpus[1] and not (set(cpus[0]) & set(cpus[1])), cpus
    chk = lines[0]["rank_check"]
    assert chk["process_group"] == {"backend": "gloo", "world_size": 2} and chk["all_ranks_bits_equal_rank0"] is True
    assert chk["frames_per_s_per_rank"] == [r["frames_per_s"] for r in ranks]


def _expand(spec):
    out = []
    for part in spec.split(","):
        a, _, b = part.partition("-")
        out.extend(range(int(a), int(b or a) + 1))
    return out


def test_bench_reports_a_failed_rank():
    """A rank that dies takes the launcher's exit code with it: non-zero, no JSON line."""
    p, lines = _run_bench({"LSSVC_BENCH_DRYRUN": "1", "LSSVC_BENCH_DRYRUN_FAIL_RANK": "1"}, "--gpus", "2", "--steps", "1", "--warmup", "0")
    assert p.returncode != 0
    assert not lines
