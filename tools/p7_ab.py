"""A/B of the persistent 7x7 kernel against the tiled one on SpyNet's shapes (interleaved rounds, one process):
    python tools/p7_ab.py [rounds] [reps]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lssvc_amd import hip_ops as ops  # noqa: E402
from lssvc_amd._lib import lib, check  # noqa: E402
from lssvc_amd.weights import WeightStore  # noqa: E402

SHAPES = [("8->32 @1152x1920", 8, 32, 1152, 1920), ("32->64 @1152x1920", 32, 64, 1152, 1920), ("64->32 @1152x1920", 64, 32, 1152, 1920),
          ("32->16 @1152x1920", 32, 16, 1152, 1920), ("32->64 @576x960", 32, 64, 576, 960), ("64->32 @576x960", 64, 32, 576, 960),
          ("32->64 @288x480", 32, 64, 288, 480), ("64->32 @288x480", 64, 32, 288, 480)]


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    dev = torch.device("cuda:0")
    ops.set_conv_precision("f16x3")
    g = torch.Generator().manual_seed(0)
    for name, cin, cout, H, W in SHAPES:
        w = torch.randn(cout, cin, 7, 7, generator=g) / math.sqrt(cin * 49)
        b = torch.randn(cout, generator=g)
        Wt = WeightStore({"c.weight": w, "c.bias": b}, dev)
        x = ops.T(torch.randn(H * W * cin, device=dev), H, W, cin, cin)
        outs, times = {}, {0: [], 1: []}
        for mode in (1, 0):
            check(lib.lssvc_set_option(b"f16x3_persist7", mode))
            outs[mode] = ops.conv(Wt, "c", x, act="relu")
            torch.cuda.synchronize()
        same = torch.equal(outs[0].buf, outs[1].buf)
        for _ in range(rounds):
            for mode in (1, 0):
                check(lib.lssvc_set_option(b"f16x3_persist7", mode))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    ops.conv(Wt, "c", x, act="relu", out=outs[mode])
                e1.record()
                torch.cuda.synchronize()
                times[mode].append(e0.elapsed_time(e1) / reps)
        flops = 2.0 * H * W * cout * 49 * cin
        med = {m: sorted(t)[len(t) // 2] for m, t in times.items()}
        print("%-22s persistent %7.1f us %6.1f TF   tiled %7.1f us %6.1f TF  %s" % (
            name, med[1] * 1e3, flops / med[1] * 1e-9, med[0] * 1e3, flops / med[0] * 1e-9, "bit-identical" if same else "*** MISMATCH ***"), flush=True)
    check(lib.lssvc_set_option(b"f16x3_persist7", 1))


if __name__ == "__main__":
    main()
