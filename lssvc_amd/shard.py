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
