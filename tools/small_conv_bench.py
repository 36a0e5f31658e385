"""3x3 convs on the small maps of the prior / hyper networks: python tools/small_conv_bench.py  (env LSSVC_TILED_NARROW=0/1)"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lssvc_amd import hip_ops as ops  # noqa: E402
from lssvc_amd._lib import lib  # noqa: E402
from lssvc_amd.weights import WeightStore  # noqa: E402

dev = torch.device("cuda:0")
ops.set_conv_precision("f16x3")
g = torch.Generator().manual_seed(0)
for cin, cout, H, W, stride in ((64, 64, 72, 120, 1), (128, 128, 72, 120, 1), (192, 192, 72, 120, 1), (64, 64, 36, 60, 1), (128, 128, 36, 60, 1),
                                (64, 64, 144, 240, 1), (128, 128, 144, 240, 2), (64, 64, 144, 240, 2), (192, 192, 18, 30, 1)):
    w = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
    Wt = WeightStore({"c.weight": w, "c.bias": torch.randn(cout, generator=g)}, dev)
    x = ops.T(torch.randn(H * W * cin, device=dev), H, W, cin, cin)
    out = ops.conv(Wt, "c", x, act="lrelu", slope=0.1, stride=stride)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.conv(Wt, "c", x, act="lrelu", slope=0.1, stride=stride, out=out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print("%4d->%4d @%dx%d s%d  %6.1f us  %5.1f TF  %s" % (cin, cout, H, W, stride, us, 2.0 * out.H * out.W * cin * cout * 9 / us * 1e-6,
                                                       lib.lssvc_conv2d_last_kernel().decode()), flush=True)
