"""GOP-level work sharding across the GPUs of one node (one process per GPU).

A GOP restarts from an I-frame with no carried state (test.py:219-227), so GOPs -- and whole
(sequence, ratio, model) jobs, which is how the reference itself spreads work over 8 GPUs
(test.py:648-656, 694-743) -- are the independent units; frames inside a GOP form a strict chain.
There is no data-path collective: ranks only exchange per-frame scalars at the end (and, in a
real deployment, the checkpoint once at start). Works over RCCL ("nccl") on GPUs and gloo on CPUs.
"""


def split_gops(n_frames, gop_size):
    """[(first_frame, n_frames_in_gop), ...] exactly as `frame_idx % gop_size == 0` restarts them."""
    return [(s, min(gop_size, n_frames - s)) for s in range(0, n_frames, gop_size)]


def assign(units, world_size, rank):
    """Static round-robin (the reference's `process_idx % gpu_num`, test.py:648-656)."""
    return [u for i, u in enumerate(units) if i % world_size == rank]


def gather_frame_records(local_records, dist=None):
    """All ranks contribute {frame_idx: record}; every rank gets the merged, frame-ordered list."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        merged = dict(local_records)
    else:
        parts = [None] * dist.get_world_size()
        dist.all_gather_object(parts, dict(local_records))
        merged = {}
        for p in parts:
            overlap = set(merged) & set(p)
            if overlap:
                raise RuntimeError("frames %s were coded by more than one rank" % sorted(overlap)[:4])
            merged.update(p)
    return [merged[k] for k in sorted(merged)]


def max_over_ranks(seconds, dist=None, device=None):
    """Wall time of the slowest rank (what bench.py reports)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.item()


def broadcast_state_dicts(paths, dist=None, device="cpu", loader=None, force=False):
    """{path: state_dict} on every rank, with only rank 0 touching the file system: rank 0 loads each checkpoint
    (`loader(path)`, default torch.load + unwrap of {"state_dict": ...}), packs all tensors of all checkpoints into ONE
    flat byte blob and broadcasts it (RCCL over xGMI when `device` is a GPU: ~245 MB once per run; gloo on CPUs) together
    with a small manifest; the other ranks cut their state dicts out of the blob. This is the "RCCL broadcast of the
    I-frame reference / weights" step of SURVEY 8(e): the only collective besides the end-of-run gather of scalars.
    Entries that are not tensors (a checkpoint may carry an epoch number or a config dict) travel inside the manifest.
    `paths` are only keys for `loader` (bench.py passes model names and a loader that draws the synthetic weights).
    force=True takes the pack / broadcast / unpack path even in a world of one rank (tests/test_gpu_shard.py runs it over
    RCCL on a single GPU)."""
    import torch

    def default_loader(path):
        sd = torch.load(path, map_location="cpu")
        return sd.get("state_dict", sd) if isinstance(sd, dict) and "state_dict" in sd else sd

    loader = loader or default_loader
    paths = list(dict.fromkeys(paths))
    solo = dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not force)
    if solo:
        return {p: loader(p) for p in paths}
    rank = dist.get_rank()
    manifest, blob = None, None
    if rank == 0:
        manifest, chunks, off = [], [], 0
        for p in paths:
            for name, t in loader(p).items():
                if not torch.is_tensor(t):
                    manifest.append((p, name, None, t, 0, 0))              # plain Python value: rides in the manifest
                    continue
                t = t.detach().contiguous().cpu()
                raw = t.reshape(-1).view(torch.uint8) if t.numel() else torch.empty(0, dtype=torch.uint8)    # (0-dim scalars too)
                manifest.append((p, name, str(t.dtype).replace("torch.", ""), tuple(t.shape), off, raw.numel()))
                chunks.append(raw)
                off += (raw.numel() + 15) // 16 * 16                      # keep every tensor 16-byte aligned inside the blob
                pad = off - (manifest[-1][4] + raw.numel())
                if pad:
                    chunks.append(torch.zeros(pad, dtype=torch.uint8))
        blob = torch.cat(chunks) if chunks else torch.empty(0, dtype=torch.uint8)
    box = [manifest, 0 if blob is None else blob.numel()]
    dist.broadcast_object_list(box, src=0)
    manifest, nbytes = box
    buf = (blob if rank == 0 else torch.empty(nbytes, dtype=torch.uint8)).to(device)
    if nbytes:
        dist.broadcast(buf, src=0)
    buf = buf.cpu()
    out = {p: {} for p in paths}
    for p, name, dtype, shape, off, n in manifest:
        if dtype is None:
            out[p][name] = shape                                          # the value itself
            continue
        t = buf[off:off + n].clone().view(getattr(torch, dtype)).reshape(shape)
        out[p][name] = t
    return out


_QUEUE_CALLS = [0]


def _shared_counter(dist):
    """An atomic counter every rank can bump: the process group's own rendezvous store (TCPStore.add is atomic), or None where
    torch keeps it out of reach."""
    try:
        from torch.distributed import distributed_c10d as c10d
        store = c10d._get_default_store()
        store.add("lssvc_probe", 0)
        return store
    except Exception:                                                     # noqa: BLE001 -- no store: the static split below
        return None


def run_sharded(units, run, dist=None, cost=None, dynamic=True):
    """`run(unit)` for every unit, the ranks sharing the work, results gathered so that EVERY rank returns the full list in unit order
    (rank 0 writes the result files, test.py:756-789).
    Round 6: a WORK QUEUE instead of the reference's static `process_idx % gpu_num` (test.py:648-656): a rank takes the next unit when
    it has finished its last one -- an atomic counter in the process group's rendezvous store (rank 0 hosts it), no data-path
    collective -- so that clips of unequal length (96-frame next to 600-frame sequences) do not leave GPUs idle at the tail.
    cost(unit) (optional): expected work; units are handed out most expensive first (longest-processing-time-first keeps the tail
    short). dynamic=False, a world of one, or no reachable store: the static round-robin."""
    solo = dist is None or not dist.is_initialized() or dist.get_world_size() == 1
    if solo:
        return [run(u) for u in units]
    world, rank = dist.get_world_size(), dist.get_rank()
    # A unit that raises (a truncated YUV file, say) must not leave the other ranks waiting in the gather until a watchdog
    # tears the job down: every rank finishes its share, failures travel with the results, and ALL ranks raise afterwards
    # with the failing unit and rank named.
    import traceback
    order = sorted(range(len(units)), key=lambda i: (-cost(units[i]), i)) if cost is not None else list(range(len(units)))
    store = _shared_counter(dist) if dynamic else None
    agree = [store is not None]
    dist.broadcast_object_list(agree, src=0)                              # every rank takes the same path as rank 0
    key = "lssvc_run_sharded_%d" % _QUEUE_CALLS[0]                        # (every rank makes the same sequence of calls)
    _QUEUE_CALLS[0] += 1

    def my_units():
        if agree[0] and store is not None:
            while True:
                k = store.add(key, 1) - 1                                 # atomic: position k of the hand-out order is mine
                if k >= len(order):
                    return
                yield order[k]
        elif agree[0]:
            raise RuntimeError("rank %d cannot reach the rendezvous store rank 0 hands work out through" % rank)
        else:
            for n, i in enumerate(order):
                if n % world == rank:
                    yield i

    mine = []
    for i in my_units():
        try:
            mine.append((i, True, run(units[i])))
        except Exception:                                                 # noqa: BLE001 -- reported below, on every rank
            mine.append((i, False, "rank %d, unit %d: %s" % (rank, i, traceback.format_exc())))
    parts = [None] * world
    dist.all_gather_object(parts, mine)
    failed = [msg for part in parts for _, ok, msg in part if not ok]
    if failed:
        raise RuntimeError("%d of %d work units failed:\n%s" % (len(failed), len(units), "\n".join(failed)))
    merged = {}
    for part in parts:
        for i, _, r in part:
            if i in merged:
                raise RuntimeError("unit %d was run by more than one rank" % i)
            merged[i] = r
    if len(merged) != len(units):
        raise RuntimeError("%d of %d units were not run" % (len(units) - len(merged), len(units)))
    return [merged[i] for i in range(len(units))]
