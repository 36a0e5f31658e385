"""1x1 convs with large K on small maps (the prior networks): python tools/pw_small_bench.py  (env LSSVC_PWKS_SMALL=0/1)"""
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
for cin, cout, H, W in ((1024, 384, 72, 120), (384, 1024, 72, 120), (384, 384, 72, 120), (512, 256, 36, 60), (256, 512, 144, 240), (1024, 384, 36, 60)):
    w = torch.randn(cout, cin, 1, 1, generator=g) / math.sqrt(cin)
    Wt = WeightStore({"c.weight": w, "c.bias": torch.randn(cout, generator=g)}, dev)
    x = ops.T(torch.randn(H * W * cin, device=dev), H, W, cin, cin)
    out = ops.conv(Wt, "c", x, act="lrelu", slope=0.1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.conv(Wt, "c", x, act="lrelu", slope=0.1, out=out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print("%4d->%4d @%dx%d  %6.1f us  %5.1f TF  %s" % (cin, cout, H, W, us, 2.0 * H * W * cin * cout / us * 1e-6, lib.lssvc_conv2d_last_kernel().decode()), flush=True)
