"""Host-side frame preparation around the hot path (the caller's side of the model API).

Restates what test.py does before/after each encode_decode call (test.py:185-200, 249-254):
inter-layer padding sizes (src/utils/common.py:48-86), the base-layer frame by MATLAB-style
antialiased bicubic down-sampling (src/utils/core.py:276-432, `imresize(..., kernel='cubic')`),
and RGB-PSNR (test.py:115-118). These run once per frame on torch tensors (CPU or GPU memory);
they are plumbing around the kernels, not part of the accelerated path.
"""
import math

import torch


def round_to_even(x):
    t = int(x)
    return t + 1 if t % 2 else t


def interlayer_padding(h_hr, w_hr, ratio):
    """get_interlayer_padding (common.py:48-86): EL padded to a multiple of 64 that is also a multiple
    of 64*ratio; BL = EL_padded / ratio."""
    def pad_to(v):
        i = 0
        while True:
            p = 64 + 32 * i
            t = (v + p - 1) // p * p
            if t % 64 == 0 and t % (64 * ratio) == 0:
                return t
            i += 1
    new_h, new_w = pad_to(h_hr), pad_to(w_hr)
    h_lr, w_lr = round_to_even(h_hr / ratio), round_to_even(w_hr / ratio)
    nh_lr, nw_lr = int(new_h / ratio), int(new_w / ratio)
    return {"P_LR": (0, nw_lr - w_lr, 0, nh_lr - h_lr), "P_HR": (0, new_w - w_hr, 0, new_h - h_hr),
            "LR_padded_size": (nh_lr, nw_lr), "HR_padded_size": (new_h, new_w),
            "LR_size": (h_lr, w_lr), "HR_size": (h_hr, w_hr)}


def _cubic(x, a=-0.5):
    """Keys cubic kernel (core.py:41-56)."""
    ax = x.abs()
    ax2 = ax * ax
    ax3 = ax * ax2
    c01 = ((a + 2) * ax3 - (a + 3) * ax2 + 1) * ax.le(1).to(x.dtype)
    c12 = ((a * ax3) - (5 * a * ax2) + (8 * a * ax) - (4 * a)) * torch.logical_and(ax.gt(1), ax.le(2)).to(x.dtype)
    return c01 + c12


def _resize_1d(x, dim, size, scale):
    """core.py:276-345 with kernel='cubic', antialiasing, symmetric ('reflect' with the edge sample
    repeated, core.py:96-129) padding -- expressed as a K-tap gather with precomputed weights."""
    if scale == 1:
        return x
    ksize = 4
    aa = 1.0
    if scale < 1:
        aa = scale
        ksize = math.ceil(ksize / aa)
    ksize += 2
    n_in = x.size(dim)
    pos = torch.linspace(0, size - 1, steps=size, dtype=x.dtype, device=x.device)
    pos = (pos + 0.5) / scale - 0.5
    base = pos.floor() - (ksize // 2) + 1
    dist = pos - base
    taps = torch.arange(ksize, dtype=x.dtype, device=x.device).view(-1, 1)
    weight = _cubic((dist.view(1, -1) - taps) * aa)
    weight = weight / weight.sum(dim=0, keepdim=True)
    idx = base.long().view(1, -1) + torch.arange(ksize, device=x.device).view(-1, 1)     # (K, size), may be <0 or >=n
    idx = torch.where(idx < 0, -idx - 1, idx)
    idx = torch.where(idx >= n_in, 2 * n_in - 1 - idx, idx)
    if dim in (2, -2):
        sample = x[:, :, idx, :]                                  # (B, C, K, size, W)
        return (sample * weight.view(1, 1, ksize, size, 1)).sum(dim=2)
    sample = x[:, :, :, idx]                                      # (B, C, H, K, size)
    return (sample * weight.view(1, 1, 1, ksize, size)).sum(dim=3)


def imresize_bicubic(x, sizes):
    """imresize(x, sizes=(h, w), kernel='cubic') (core.py:364-432) for a (B,C,H,W) float tensor."""
    h, w = x.shape[-2:]
    x = _resize_1d(x, -2, sizes[0], sizes[0] / h)
    return _resize_1d(x, -1, sizes[1], sizes[1] / w)


def make_layers(rgb_el, ratio):
    """test.py:191-199: zero-pad the EL frame (bottom/right), derive the padded BL frame."""
    pad = interlayer_padding(rgb_el.shape[2], rgb_el.shape[3], ratio)
    x_el = torch.nn.functional.pad(rgb_el, pad["P_HR"], mode="constant", value=0)
    x_bl = imresize_bicubic(x_el, pad["LR_padded_size"]).clamp_(0, 1)
    return x_bl, x_el, pad


def psnr(a, b):
    """test.py:115-118."""
    return (10 * torch.log10(1.0 / torch.mean((a - b) ** 2))).item()
