"""A/B of the persistent 3x3 kernel against the tiled one on the benchmark's dominant shapes, interleaved rounds in ONE
process (cdna_hip_programming.md rule 24), random data:  python tools/p3_ab.py [rounds] [reps]
Also checks that the two kernels agree bit for bit on every shape."""
import ctypes as C
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lssvc_amd import hip_ops as ops  # noqa: E402
from lssvc_amd._lib import lib, check  # noqa: E402
from lssvc_amd.weights import WeightStore  # noqa: E402

SHAPES = [
    # name, cins, cout, H, W, in_act, act, residual, subpel
    ("64->64 @1152x1920", [64], 64, 1152, 1920, None, None, False, False),
    ("64->64 @1152x1920 lrelu/res", [64], 64, 1152, 1920, "lrelu", None, True, False),
    ("48->48 @1152x1920 lrelu/lrelu", [48], 48, 1152, 1920, "lrelu", "lrelu", False, False),
    ("48->48 @1152x1920 res", [48], 48, 1152, 1920, None, None, True, False),
    ("96->48 @1152x1920 (cat)", [48, 48], 48, 1152, 1920, None, None, False, False),
    ("96->64 @1152x1920", [96], 64, 1152, 1920, None, "lrelu", False, False),
    ("64->64 @576x960", [64], 64, 576, 960, "lrelu", "lrelu", False, False),
    ("128->64 @576x960 (cat)", [64, 64], 64, 576, 960, None, None, False, False),
    ("128->192 @576x960 subpel", [128], 192, 576, 960, None, None, False, True),
    ("128->128 @576x960 subpel", [128], 128, 576, 960, None, "lrelu", False, True),
    ("96->96 @288x480", [96], 96, 288, 480, None, None, False, False),
    ("192->256 @288x480 subpel", [192], 256, 288, 480, None, None, False, True),
]
MODES = [("prodcons", 1), ("tiled", 0)]


def setopt(name, v):
    check(lib.lssvc_set_option(name.encode(), v))


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    only = sys.argv[3] if len(sys.argv) > 3 else None
    dev = torch.device("cuda:0")
    ops.set_conv_precision("f16x3")
    g = torch.Generator().manual_seed(0)
    for name, cins, cout, H, W, in_act, act, res, subpel in SHAPES:
        if only and only not in name:
            continue
        cin = sum(cins)
        w = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
        b = torch.randn(cout, generator=g)
        key = "s.0" if subpel else "c"
        Wt = WeightStore({key + ".weight": w, key + ".bias": b}, dev)
        xs = [ops.T(torch.randn(H * W * c, device=dev), H, W, c, c) for c in cins]
        r = ops.T(torch.randn(H * W * cout, device=dev), H, W, cout, cout) if res else None
        kw = dict(in_act=in_act, in_slope=0.1, act=act, slope=0.01, residual=r)

        def run(out=None):
            return ops.subpel(Wt, "s", xs, out=out, **kw) if subpel else ops.conv(Wt, "c", xs, out=out, **kw)

        outs, times = {}, {m[0]: [] for m in MODES}
        for mname, on in MODES:
            setopt("f16x3_persist", on)
            outs[mname] = run()
            torch.cuda.synchronize()
        same = torch.equal(outs["tiled"].buf, outs["prodcons"].buf)
        for _ in range(rounds):
            for mname, on in MODES:
                setopt("f16x3_persist", on)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    run(out=outs[mname])
                e1.record()
                torch.cuda.synchronize()
                times[mname].append(e0.elapsed_time(e1) / reps)
        flops = 2.0 * H * W * cout * 9 * cin
        line = "%-32s" % name
        for mname, _ in MODES:
            t = sorted(times[mname])
            line += "  %s %7.1f us %6.1f TF" % (mname, t[len(t) // 2] * 1e3, flops / t[len(t) // 2] * 1e-9)
        print(line + ("  bit-identical" if same else "  *** MISMATCH ***"), flush=True)
    setopt("f16x3_persist", 1)


if __name__ == "__main__":
    main()
