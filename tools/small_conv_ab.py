"""Small-map 3x3 convs of the bench (the launches the tiled conv_f16x3_kernel serves): time per launch and a digest of the output,
so that two builds can be compared shape by shape.   LSSVC_HIP_LIB=<other .so> python tools/small_conv_ab.py [reps]"""
import hashlib
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lssvc_amd import hip_ops as ops  # noqa: E402
from lssvc_amd.weights import WeightStore  # noqa: E402

SHAPES = [  # cin, cout, H, W (input), stride       -- from profiles/r04_bench_signatures.txt
    (64, 64, 144, 240, 1), (128, 128, 72, 120, 1), (128, 128, 144, 240, 1), (64, 128, 144, 240, 1), (128, 64, 144, 240, 1),
    (96, 96, 36, 60, 1), (96, 128, 72, 120, 1), (128, 384, 72, 120, 1), (64, 256, 72, 120, 1), (384, 320, 36, 60, 1),
    (192, 170, 72, 120, 1), (320, 256, 36, 60, 1), (256, 192, 36, 60, 1), (96, 96, 72, 120, 1), (128, 128, 18, 30, 1),
    (64, 64, 288, 480, 2), (96, 128, 144, 240, 2), (128, 64, 288, 480, 2), (128, 128, 72, 120, 2), (64, 64, 36, 60, 2),
    (64, 64, 144, 240, 2), (160, 144, 144, 240, 2), (128, 128, 144, 240, 2), (144, 192, 72, 120, 2), (192, 96, 288, 480, 2),
]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    ops.set_conv_precision("f16x3")
    dev = torch.device("cuda:0")
    total = 0.0
    only = os.environ.get("SMALL_CONV_ONLY")                 # e.g. "0,1,5": indices into SHAPES
    shapes = [SHAPES[int(i)] for i in only.split(",")] if only else SHAPES
    for cin, cout, H, W, s in shapes:
        g = torch.Generator().manual_seed(cin * 1000 + cout + H + s)
        w = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
        Wt = WeightStore({"c.weight": w, "c.bias": torch.randn(cout, generator=g)}, dev)
        x = ops.T(torch.randn(H * W * cin, generator=g).to(dev), H, W, cin, cin)
        kw = dict(stride=s, act="lrelu", slope=0.01, in_act="lrelu", in_slope=0.1)
        out = ops.conv(Wt, "c", [x], **kw)
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                ops.conv(Wt, "c", [x], out=out, **kw)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / reps * 1e3)
        t = sorted(ts)[3]
        total += t
        print("%3d->%-3d @%3dx%-3d s%d  %-36s %7.1f us   sha1 %s" % (cin, cout, H, W, s, ops.lib.lssvc_conv2d_last_kernel().decode(), t,
              hashlib.sha1(out.buf.cpu().numpy().tobytes()).hexdigest()[:12]), flush=True)
    print("sum %.1f us" % total)


main()
