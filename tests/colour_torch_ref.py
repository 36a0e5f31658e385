"""Torch restatements of the reference's colour conversions (src/transforms/functional.py) -- CHECKERS for the device kernels of
lssvc_amd/csrc/prepost.hip (tests/test_harness_host.py pins them to the reference's own outputs, tests/test_gpu_prepost.py holds
the kernels to them) and clip writers for the harness tools. Test infrastructure: the product path (lssvc_amd/harness.py) does
its colour work on the device through lssvc_amd/prepost.py and never imports this."""
import numpy as np
import torch
import torch.nn.functional as F

KR, KG, KB = 0.2126, 0.7152, 0.0722                                    # ITU-R BT.709 (functional.py:10-13)


def yuv420_to_rgb(y, u, v, device):
    """`ycbcr420_to_rgb(y, uv, order=1)` (functional.py:42-58) on the device: chroma x2 by linear interpolation with
    scipy.ndimage.zoom's sample positions (output i <-> input i*(n-1)/(2n-1), i.e. align_corners=True), BT.709, clip.
    Returns (1,3,H,W) fp32 plus the normalised planes the per-plane PSNRs are taken against."""
    yt = torch.from_numpy(np.ascontiguousarray(y)).to(device).float().div_(255.0)[None, None]
    uv = torch.from_numpy(np.stack([u, v])).to(device).float().div_(255.0)[None]
    up = F.interpolate(uv, size=(yt.shape[2], yt.shape[3]), mode="bilinear", align_corners=True)
    cb, cr = up[:, 0:1], up[:, 1:2]
    r = yt + (2 - 2 * KR) * (cr - 0.5)
    b = yt + (2 - 2 * KB) * (cb - 0.5)
    g = (yt - KR * r - KB * b) / KG
    return torch.cat([r, g, b], dim=1).clamp_(0.0, 1.0), yt[0, 0], uv[0, 0], uv[0, 1]


def rgb_to_yuv420(rgb):
    """`rgb_to_ycbcr420` (functional.py:16-39) for a (1,3,H,W) device tensor -> (y, u, v) planes in [0,1]."""
    r, g, b = rgb[0, 0], rgb[0, 1], rgb[0, 2]
    y = KR * r + KG * g + KB * b
    cb = 0.5 * (b - y) / (1 - KB) + 0.5
    cr = 0.5 * (r - y) / (1 - KR) + 0.5
    h, w = y.shape
    cb = cb.reshape(h // 2, 2, w // 2, 2).mean(dim=(1, 3))
    cr = cr.reshape(h // 2, 2, w // 2, 2).mean(dim=(1, 3))
    return y.clamp(0, 1), cb.clamp(0, 1), cr.clamp(0, 1)


def _plane_psnr(a, b):
    """mse2PSNR (test.py:104-109)."""
    mse = torch.mean((a - b) ** 2).item()
    return 10 * np.log10(1.0 / mse) if mse > 1e-10 else 999.9
