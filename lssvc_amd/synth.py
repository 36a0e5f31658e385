"""Deterministic synthetic checkpoints and clips.

Pretrained LSSVC checkpoints are not redistributable/offline, so benchmarks, smoke tests and
golden fixtures all use weights generated here: every tensor is drawn from its own
`torch.Generator` seeded by crc32(key) ^ seed, so a tensor's values depend only on
(key, shape, seed) -- never on iteration order -- and are reproducible on any machine with the
same torch build.  The layout (keys / shapes / kinds) comes from the manifests in
`lssvc_amd/manifests/`, which mirror the reference's state-dict layout
(reference: IntraSS.py:174-214, LSSVC_net.py:141-149).
"""
import json
import math
import os
import zlib

import torch

_MANIFEST_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "manifests")


def load_manifest(name):
    """name in {'intra_ss', 'lssvc_extend'} -> list of dict(key, shape, dtype, kind[, value])."""
    with open(os.path.join(_MANIFEST_DIR, name + ".json")) as f:
        return json.load(f)["tensors"]


def _gen(key, seed):
    g = torch.Generator()
    g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


_PEDESTAL = (2.0 ** -18) ** 2


def _make(entry, seed, gain):
    key, shape, kind = entry["key"], tuple(entry["shape"]), entry["kind"]
    g = _gen(key, seed)
    randn = lambda *s: torch.randn(*s, generator=g)
    rand = lambda *s: torch.rand(*s, generator=g)
    if kind == "const":
        return torch.tensor(entry["value"], dtype=getattr(torch, entry["dtype"])).reshape(shape)
    if kind in ("conv", "dwconv"):
        fan_in = shape[1] * shape[2] * shape[3]
        return randn(*shape) * (gain * math.sqrt(2.0 / fan_in))
    if kind == "convT":       # (Cin, Cout, k, k); ~k*k/4 taps hit each output pixel at stride 2
        fan_in = shape[0] * shape[2] * shape[3] / 2.0
        return randn(*shape) * (gain * math.sqrt(2.0 / fan_in))
    if kind == "bias":
        return randn(*shape) * 0.05
    if kind == "gdn_beta":
        return torch.sqrt(1.0 + 0.5 * rand(*shape) + _PEDESTAL)
    if kind == "gdn_gamma":
        c = shape[0]
        return torch.sqrt(0.1 * torch.eye(c) + (0.2 / c) * rand(c, c) + _PEDESTAL)
    if kind == "eb_matrix":   # EntropyBottleneck._matrices.i (C, f_out, f_in)
        scale = 10.0 ** (1.0 / 5.0)
        init = math.log(math.expm1(1.0 / scale / shape[1]))
        return init + 0.3 * randn(*shape)
    if kind == "eb_bias":
        return rand(*shape) - 0.5
    if kind == "eb_factor":
        return 0.3 * randn(*shape)
    if kind == "eb_quantiles":  # (C,1,3): [-10, 0, 10] shifted by a per-channel median
        med = 0.7 * randn(shape[0], 1, 1)
        return torch.tensor([-10.0, 0.0, 10.0]).view(1, 1, 3) + med
    if kind == "bitparm":     # Bitparm h/b/a, (1,C,1,1)
        return 0.5 * randn(*shape)
    raise ValueError("unknown manifest kind %r for %s" % (kind, key))


def synth_state_dict(name, seed=0, gain=0.6):
    """Build a full state dict for `name` ('intra_ss' | 'lssvc_extend')."""
    return {e["key"]: _make(e, seed, gain) for e in load_manifest(name)}


def synth_clip(frames, height, width, seed=0, motion=(0.7, -0.4), noise=0.004):
    """A smooth-texture clip with a known global sub-pixel translation per frame plus a little
    sensor noise, quantised to 8 bits: returns uint8 (frames, 3, H, W); float frame = u8 / 255."""
    g = torch.Generator()
    g.manual_seed(1000 + seed)
    pad = 16 + int(abs(motion[0]) * frames + abs(motion[1]) * frames)
    hh, ww = height + 2 * pad, width + 2 * pad
    base = torch.zeros(1, 3, hh, ww)
    for octave, amp in ((32, 0.30), (8, 0.15), (2, 0.05)):
        coarse = torch.rand(1, 3, max(hh // octave, 2) + 2, max(ww // octave, 2) + 2, generator=g)
        base = base + amp * (torch.nn.functional.interpolate(coarse, size=(hh, ww), mode="bicubic",
                                                             align_corners=False) - 0.5)
    base = (base + 0.5).clamp(0, 1)
    ys = torch.arange(height, dtype=torch.float32)
    xs = torch.arange(width, dtype=torch.float32)
    out = []
    for t in range(frames):
        gy = (ys + pad + motion[1] * t) / (hh - 1) * 2 - 1
        gx = (xs + pad + motion[0] * t) / (ww - 1) * 2 - 1
        grid = torch.stack(torch.meshgrid(gx, gy, indexing="xy"), dim=-1).unsqueeze(0)
        fr = torch.nn.functional.grid_sample(base, grid, mode="bilinear", align_corners=True)
        fr = fr + noise * torch.randn(fr.shape, generator=g)
        out.append((fr.clamp(0, 1) * 255.0).round().to(torch.uint8))
    return torch.cat(out, 0)


def synth_clip_exact(frames, height, width, seed=0, motion_q8=(6, -3)):
    """Integer-only twin of synth_clip for the full-size golden fixtures: the same kind of picture (three octaves of
    smooth value noise, a global sub-pixel translation per frame in 1/8-pixel units, +-1 LSB sensor noise), but every
    step is exact int64 arithmetic on numpy's PCG64 integer stream, so the clip is bit-identical on any host and a
    fixture only has to store its sha1. Returns uint8 (frames, 3, H, W) as a torch tensor."""
    import numpy as np
    rng = np.random.Generator(np.random.PCG64(7000 + seed))
    pad = 4 + (abs(motion_q8[0]) * frames + 7) // 8 + (abs(motion_q8[1]) * frames + 7) // 8
    hh, ww = height + 2 * pad, width + 2 * pad
    acc = np.zeros((3, hh, ww), dtype=np.int64)                     # texture in units of 1/(256*64) of full scale
    for octave, amp in ((32, 150), (8, 75), (2, 25)):               # amplitudes sum to 250 < 256
        gh, gw = hh // octave + 2, ww // octave + 2
        coarse = rng.integers(0, 256, size=(3, gh, gw), dtype=np.int64)
        ys, xs = np.arange(hh), np.arange(ww)
        y0, fy = ys // octave, ys % octave
        x0, fx = xs // octave, xs % octave
        top = coarse[:, y0][:, :, x0] * (octave - fx) + coarse[:, y0][:, :, x0 + 1] * fx
        bot = coarse[:, y0 + 1][:, :, x0] * (octave - fx) + coarse[:, y0 + 1][:, :, x0 + 1] * fx
        v = top * (octave - fy)[None, :, None] + bot * fy[None, :, None]          # 0 .. 255 * octave^2
        acc += (v * (amp * 64)) // (octave * octave)                              # 0 .. 255 * amp * 64
    out = np.empty((frames, 3, height, width), dtype=np.uint8)
    for t in range(frames):
        ox, oy = pad * 8 + motion_q8[0] * t, pad * 8 + motion_q8[1] * t          # offsets in 1/8 pixel, >= 0
        x0, fx, y0, fy = ox // 8, ox % 8, oy // 8, oy % 8
        a = acc[:, y0:y0 + height + 1, x0:x0 + width + 1]
        top = a[:, :-1, :-1] * (8 - fx) + a[:, :-1, 1:] * fx
        bot = a[:, 1:, :-1] * (8 - fx) + a[:, 1:, 1:] * fx
        v = top * (8 - fy) + bot * fy                                             # 0 .. 255 * 250 * 64 * 64
        noise = rng.integers(-1, 2, size=v.shape, dtype=np.int64)
        px = (v + (250 * 64 * 64) // 2) // (250 * 64 * 64) + noise + 3            # small pedestal keeps it off 0
        out[t] = np.clip(px, 0, 255).astype(np.uint8)
    return torch.from_numpy(out)
