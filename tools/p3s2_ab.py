"""Stride-2 3x3 convs: the persistent kernel's stride-2 form against the tiled kernel, same process, interleaved rounds, random data:
    python tools/p3s2_ab.py [rounds] [reps]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lssvc_amd import hip_ops as ops  # noqa: E402
from lssvc_amd._lib import lib, check  # noqa: E402
from lssvc_amd.weights import WeightStore  # noqa: E402

SHAPES = [([48], 64, 1152, 1920), ([56], 64, 1152, 1920), ([4], 64, 1152, 1920), ([64], 64, 576, 960), ([64], 96, 576, 960), ([128], 96, 576, 960),
          ([96], 128, 288, 480)]


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    dev = torch.device("cuda:0")
    ops.set_conv_precision("f16x3")
    g = torch.Generator().manual_seed(0)
    for cins, cout, H, W in SHAPES:
        cin = sum(cins)
        w = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
        Wt = WeightStore({"c.weight": w, "c.bias": torch.randn(cout, generator=g)}, dev)
        xs = [ops.T(torch.randn(H * W * c, device=dev), H, W, c, c) for c in cins]
        outs, times, names = {}, {0: [], 1: []}, {}
        for on in (1, 0):
            check(lib.lssvc_set_option(b"f16x3_persist_s2", on))
            outs[on] = ops.conv(Wt, "c", xs, stride=2, act="lrelu", slope=0.1)
            names[on] = lib.lssvc_conv2d_last_kernel().decode()
        torch.cuda.synchronize()
        same = torch.equal(outs[0].buf, outs[1].buf)
        for _ in range(rounds):
            for on in (1, 0):
                check(lib.lssvc_set_option(b"f16x3_persist_s2", on))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    ops.conv(Wt, "c", xs, stride=2, act="lrelu", slope=0.1, out=outs[on])
                e1.record()
                torch.cuda.synchronize()
                times[on].append(e0.elapsed_time(e1) / reps * 1e3)
        check(lib.lssvc_set_option(b"f16x3_persist_s2", 1))
        flops = 2.0 * (H // 2) * (W // 2) * cout * 9 * cin
        mb = 4e-6 * (H * W * cin + (H // 2) * (W // 2) * cout)
        t1, t0 = sorted(times[1])[len(times[1]) // 2], sorted(times[0])[len(times[0]) // 2]
        print("%4d->%3d in %4dx%-4d  persistent %7.1f us %6.1f TF %5.2f TB/s   tiled %7.1f us %6.1f TF   x%.2f  bit-identical %s   (%s | %s)" % (
            cin, cout, H, W, t1, flops / t1 * 1e-6, mb / t1, t0, flops / t0 * 1e-6, t0 / t1, same, names[1], names[0]), flush=True)


main()
