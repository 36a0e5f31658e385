"""Frame pre/post-processing on the device through liblssvc_hip.so's prepost kernels (csrc/prepost.hip):
what test.py does around every encode_decode call (test.py:185-201, 249-311) -- 4:2:0 -> RGB, inter-layer zero padding, the
MATLAB-bicubic base-layer frame, and the RGB / Y,U,V PSNR sums -- without torch ops on the data path. torch is used here
for device memory and for building the (tiny) resampling tables once per size."""
import ctypes as C
import math

import numpy as np
import torch

from . import hip_ops as ops
from . import preprocess
from ._lib import lib, check
from .hip_ops import T


def _cubic_tables(n_in, n_out):
    """Tap tables of core.py:276-345 (kernel='cubic', antialiasing, symmetric padding) for one axis, built with the same
    fp32 torch expressions as preprocess._resize_1d: weights [n_out][K] fp32 and source indexes [n_out][K] int32."""
    scale = n_out / n_in
    if scale == 1:
        return torch.ones(n_out, 1), torch.arange(n_out, dtype=torch.int32).view(-1, 1)
    ksize, aa = 4, 1.0
    if scale < 1:
        aa = scale
        ksize = math.ceil(ksize / aa)
    ksize += 2
    pos = torch.linspace(0, n_out - 1, steps=n_out, dtype=torch.float32)
    pos = (pos + 0.5) / scale - 0.5
    base = pos.floor() - (ksize // 2) + 1
    dist = pos - base
    taps = torch.arange(ksize, dtype=torch.float32).view(-1, 1)
    weight = preprocess._cubic((dist.view(1, -1) - taps) * aa)
    weight = weight / weight.sum(dim=0, keepdim=True)
    idx = base.long().view(1, -1) + torch.arange(ksize).view(-1, 1)
    idx = torch.where(idx < 0, -idx - 1, idx)
    idx = torch.where(idx >= n_in, 2 * n_in - 1 - idx, idx)
    return weight.t().contiguous(), idx.t().contiguous().int()


class FramePrep:
    """Per-device frame preparation / metrics. One instance per (device); tables are cached per size."""

    def __init__(self, device):
        self.device = torch.device(device)
        self._tables = {}
        self.sums = torch.zeros(16, dtype=torch.float64, device=self.device)          # sqdiff accumulators, one D2H per frame
        self.ws = torch.empty(int(lib.lssvc_reduce_workspace_bytes()) // 8, dtype=torch.float64, device=self.device)

    # ---- inputs -------------------------------------------------------------------------------------------------------
    def frame_from_rgb8(self, rgb_u8, pad_hw):
        """(3,H,W) uint8 device tensor -> zero-padded fp32 NHWC frame T (pad_hw = padded size)."""
        assert rgb_u8.dtype == torch.uint8 and rgb_u8.dim() == 3 and rgb_u8.shape[0] == 3 and rgb_u8.is_cuda and rgb_u8.is_contiguous()
        out = T.empty(pad_hw[0], pad_hw[1], 3, self.device)
        check(lib.lssvc_rgb8_to_frame(C.c_void_p(rgb_u8.data_ptr()), rgb_u8.shape[1], rgb_u8.shape[2], out.ref, ops.stream_ptr()))
        return out

    def frame_from_yuv420(self, y, u, v, pad_hw, want_planes=True):
        """8-bit planes on the device -> (padded RGB frame T, (y, u, v) normalised fp32 planes or None)."""
        H, W = y.shape
        out = T.empty(pad_hw[0], pad_hw[1], 3, self.device)
        planes = None
        args = [None, None, None]
        if want_planes:
            planes = (torch.empty(H, W, device=self.device), torch.empty(H // 2, W // 2, device=self.device),
                      torch.empty(H // 2, W // 2, device=self.device))
            args = [C.c_void_p(t.data_ptr()) for t in planes]
        check(lib.lssvc_yuv420_to_frame(C.c_void_p(y.data_ptr()), C.c_void_p(u.data_ptr()), C.c_void_p(v.data_ptr()), H, W, out.ref,
                                        args[0], args[1], args[2], ops.stream_ptr()))
        return out, planes

    def bicubic(self, frame, out_hw, clamp=(0.0, 1.0)):
        """imresize(frame, sizes=out_hw, kernel='cubic') (+ the clamp test.py:199 applies): frame T -> T."""
        key = (frame.H, frame.W, out_hw[0], out_hw[1])
        if key not in self._tables:
            wv, iv = _cubic_tables(frame.H, out_hw[0])
            wh, ih = _cubic_tables(frame.W, out_hw[1])
            self._tables[key] = tuple(t.to(self.device) for t in (wv, iv, wh, ih))
        wv, iv, wh, ih = self._tables[key]
        out = T.empty(out_hw[0], out_hw[1], frame.C, self.device)
        check(lib.lssvc_resample2d(frame.ref, out.ref, C.c_void_p(wv.data_ptr()), C.c_void_p(iv.data_ptr()), wv.shape[1],
                                   C.c_void_p(wh.data_ptr()), C.c_void_p(ih.data_ptr()), wh.shape[1], clamp[0], clamp[1],
                                   ops.stream_ptr()))
        return out

    def make_layers_rgb8(self, rgb_u8, ratio):
        """test.py:191-199 from an 8-bit RGB frame: -> (x_bl, x_el) as (1,3,h,w) tensors for the model API, padding info."""
        pad = preprocess.interlayer_padding(rgb_u8.shape[1], rgb_u8.shape[2], ratio)
        x_el = self.frame_from_rgb8(rgb_u8, pad["HR_padded_size"])
        x_bl = self.bicubic(x_el, pad["LR_padded_size"])
        return x_bl.to_nchw(), x_el.to_nchw(), pad

    # ---- metrics ------------------------------------------------------------------------------------------------------
    def rgb_to_yuv420(self, frame, h, w, clamp01=False):
        """rgb_to_ycbcr420 of the top-left h x w crop -> (y, u, v) fp32 planes."""
        y = torch.empty(h, w, device=self.device)
        u, v = torch.empty(h // 2, w // 2, device=self.device), torch.empty(h // 2, w // 2, device=self.device)
        check(lib.lssvc_rgb_to_yuv420(frame.ref, h, w, 1 if clamp01 else 0, C.c_void_p(y.data_ptr()), C.c_void_p(u.data_ptr()),
                                      C.c_void_p(v.data_ptr()), ops.stream_ptr()))
        return y, u, v

    def _slot(self, i):
        return C.c_void_p(self.sums.data_ptr() + 8 * i)

    def sqdiff_frames(self, a, b, h, w, slot, clamp01=False):
        check(lib.lssvc_sqdiff_sum(a.ref, b.ref, h, w, 1 if clamp01 else 0, self._slot(slot), C.c_void_p(self.ws.data_ptr()),
                                   ops.stream_ptr()))

    def sqdiff_planes(self, a, b, slot):
        assert a.shape == b.shape and a.is_contiguous() and b.is_contiguous()
        check(lib.lssvc_sqdiff_sum_flat(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), a.numel(), self._slot(slot),
                                        C.c_void_p(self.ws.data_ptr()), ops.stream_ptr()))

    def fetch(self):
        return self.sums.cpu().tolist()


def psnr_from_sum(sq_sum, n):
    """mse2PSNR (test.py:104-109) / PSNR (test.py:115-118)."""
    mse = sq_sum / n
    return 10 * np.log10(1.0 / mse) if mse > 1e-10 else 999.9
