"""Micro-benchmark of lssvc_conv2d on the shapes that dominate the 1080p workload (GPU box only):
    python tools/conv_microbench.py [reps]
Prints algorithmic TFLOP/s per shape (HIP events around `reps` back-to-back launches)."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lssvc_amd import hip_ops as ops  # noqa: E402
from lssvc_amd.weights import WeightStore  # noqa: E402

SHAPES = [
    # name, cins, cout, k, stride, H, W
    ("3x3 64->64 @1152x1920", [64], 64, 3, 1, 1152, 1920),
    ("3x3 64->64 @576x960", [64], 64, 3, 1, 576, 960),
    ("3x3 48->48 @1152x1920", [48], 48, 3, 1, 1152, 1920),
    ("3x3 96->48 @1152x1920 (cat)", [48, 48], 48, 3, 1, 1152, 1920),
    ("3x3 128->64 @576x960 (cat)", [64, 64], 64, 3, 1, 576, 960),
    ("3x3 96->96 @288x480", [96], 96, 3, 1, 288, 480),
    ("7x7 32->64 @1152x1920", [32], 64, 7, 1, 1152, 1920),
    ("7x7 64->32 @1152x1920", [64], 32, 7, 1, 1152, 1920),
    ("1x1 64->256 @1152x1920", [64], 256, 1, 1, 1152, 1920),
    ("1x1 256->64 @1152x1920", [256], 64, 1, 1, 1152, 1920),
    ("1x1 48->48 @1152x1920", [48], 48, 1, 1, 1152, 1920),
    ("3x3s2 51->64 @1152x1920", [3, 48], 64, 3, 2, 1152, 1920),
    ("3x3 384->384 @72x120", [384], 384, 3, 1, 72, 120),
    ("1x1 384->384 @72x120", [384], 384, 1, 1, 72, 120),
    ("3x3 128->128 @72x120", [128], 128, 3, 1, 72, 120),
    ("1x1 1024->384 @72x120", [1024], 384, 1, 1, 72, 120),
    ("1x1 384->1024 @72x120", [384], 1024, 1, 1, 72, 120),
    ("1x1 512->128 @288x480", [512], 128, 1, 1, 288, 480),
    ("3x3 192->192 @36x60", [192], 192, 3, 1, 36, 60),
    ("big 3x3 64->64 @2304x3840", [64], 64, 3, 1, 2304, 3840),
    ("big 7x7 32->64 @2304x3840", [32], 64, 7, 1, 2304, 3840),
]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    only = sys.argv[2] if len(sys.argv) > 2 else None
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    for name, cins, cout, k, stride, H, W in SHAPES:
        if only is not None and only not in name:
            continue
        cin = sum(cins)
        w = torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)
        b = torch.randn(cout, generator=g)
        Wt = WeightStore({"c.weight": w, "c.bias": b}, dev)
        xs = [ops.T(torch.randn(H * W * c, device=dev), H, W, c, c) for c in cins]
        out = ops.conv(Wt, "c", xs, stride=stride)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.conv(Wt, "c", xs, stride=stride, out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        flops = 2.0 * out.H * out.W * cout * k * k * cin
        var = ops.lib.lssvc_conv2d_variant(out.H, out.W, (cout + 15) // 16 * 16, stride)
        print("%-34s <%d,%d> %8.1f us  %7.2f TFLOP/s" % (name, var // 16, var % 16, ms * 1e3, flops / ms * 1e-9), flush=True)


if __name__ == "__main__":
    main()
