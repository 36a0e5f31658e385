"""OffsetDiversity tail (lssvc_offset_diversity) at the bench size: python tools/od_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lssvc_amd import hip_ops as ops  # noqa: E402

dev = torch.device("cuda:0")
H, W = 1152, 1920
x = ops.T(torch.randn(H * W * 48, device=dev), H, W, 48, 48)
om = ops.T(torch.randn(H * W * 96, device=dev) * 0.02, H, W, 96, 96)
fl = ops.T(torch.randn(H * W * 2, device=dev) * 3, H, W, 2, 2)
fw = torch.randn(48 * 6, device=dev)
fb = torch.randn(48, device=dev)
out = x.like()
for _ in range(3):
    ops.offset_diversity_tail(x, om, fl, fw, fb, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.offset_diversity_tail(x, om, fl, fw, fb, out=out)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print("offset_diversity 1152x1920: %.1f us (om 96 + out 48 channels: %.2f TB/s)" % (ms * 1e3, H * W * (96 + 48 + 2) * 4 / ms / 1e9))
