"""A/B of the persistent 3x3 kernel with PRE-SPLIT inputs (patches by LDS-DMA) against the fp32-input
form (patches through registers: load, convert, ds_write), each with the direct and with the staged epilogue, interleaved
rounds in ONE process, random data:  python tools/p3_split_ab.py [rounds] [reps] [only]
Also checks that all four forms agree bit for bit (the split values are what the kernel makes of the fp32 input itself)."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lssvc_amd import hip_ops as ops  # noqa: E402
from lssvc_amd._lib import lib, check  # noqa: E402
from lssvc_amd.weights import WeightStore  # noqa: E402

SHAPES = [
    # name, cins, cout, H, W, in_act, act, residual, subpel, stride
    ("64->64 @1152x1920", [64], 64, 1152, 1920, None, None, False, False, 1),
    ("64->64 @1152x1920 lrelu/res", [64], 64, 1152, 1920, "lrelu", None, True, False, 1),
    ("64->64 @576x960", [64], 64, 576, 960, None, None, False, False, 1),
    ("64->64 @576x960 lrelu/res", [64], 64, 576, 960, "lrelu", None, True, False, 1),
    ("48->48 @1152x1920", [48], 48, 1152, 1920, None, None, False, False, 1),
    ("48->48 @1152x1920 lrelu/lrelu", [48], 48, 1152, 1920, "lrelu", "lrelu", False, False, 1),
    ("48->48 @1152x1920 res", [48], 48, 1152, 1920, None, None, True, False, 1),
    ("96->48 @1152x1920 (cat)", [48, 48], 48, 1152, 1920, None, None, False, False, 1),
    ("128->64 @576x960 (cat)", [64, 64], 64, 576, 960, None, None, False, False, 1),
    ("128->192 @576x960 subpel", [128], 192, 576, 960, None, None, False, True, 1),
    ("96->96 @288x480", [96], 96, 288, 480, None, None, False, False, 1),
    ("48->64 s2 @1152x1920", [48], 64, 1152, 1920, None, None, False, False, 2),
    ("32->32 @1152x1920", [32], 32, 1152, 1920, None, None, False, False, 1),
]
MODES = [("fp32-in", False, 0), ("split-in", True, 0), ("fp32-in+staged", False, 1), ("split-in+staged", True, 1)]


def setopt(name, v):
    check(lib.lssvc_set_option(name.encode(), v))


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    only = sys.argv[3] if len(sys.argv) > 3 else None
    dev = torch.device("cuda:0")
    ops.set_conv_precision("f16x3")
    g = torch.Generator().manual_seed(0)
    for name, cins, cout, H, W, in_act, act, res, subpel, stride in SHAPES:
        if only and only not in name:
            continue
        cin = sum(cins)
        w = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
        b = torch.randn(cout, generator=g)
        key = "s.0" if subpel else "c"
        Wt = WeightStore({key + ".weight": w, key + ".bias": b}, dev)
        xs = [ops.T(torch.randn(H * W * c, device=dev) * 3.0, H, W, c, c) for c in cins]
        Ho, Wo = H // stride, W // stride
        r = ops.T(torch.randn(Ho * Wo * cout, device=dev), Ho, Wo, cout, cout) if res else None
        # the pre-split copies carry the input activation; time the conversion itself as well
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        xs_split = [ops.presplit(x, in_act, 0.1) for x in xs]
        e0.record()
        for x, xsp in zip(xs, xs_split):
            ops.presplit(x, in_act, 0.1, out=xsp)
        e1.record()
        torch.cuda.synchronize()
        t_split = e0.elapsed_time(e1) * 1e3

        def run(split, out=None):
            kw = dict(act=act, slope=0.01, residual=r, stride=stride)
            if not split:
                kw.update(in_act=in_act, in_slope=0.1)
            ins = xs_split if split else xs
            return ops.subpel(Wt, "s", ins, out=out, **kw) if subpel else ops.conv(Wt, "c", ins, out=out, **kw)

        outs, times = {}, {m[0]: [] for m in MODES}
        for mname, split, stage in MODES:
            setopt("p3_stage", stage)
            outs[mname] = run(split)
            torch.cuda.synchronize()
        same = all(torch.equal(outs["fp32-in"].buf, outs[m[0]].buf) for m in MODES[1:])
        for _ in range(rounds):
            for mname, split, stage in MODES:
                setopt("p3_stage", stage)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    run(split, out=outs[mname])
                e1.record()
                torch.cuda.synchronize()
                times[mname].append(e0.elapsed_time(e1) / reps)
        flops = 2.0 * Ho * Wo * cout * 9 * cin
        line = "%-30s" % name
        base = None
        for mname, _, _ in MODES:
            t = sorted(times[mname])
            med = t[len(t) // 2]
            base = base or med
            line += "  %s %6.1f us %5.1f TF (%+5.1f%%)" % (mname, med * 1e3, flops / med * 1e-9, (base / med - 1) * 100)
        nbytes = 4 * H * W * cin * 2
        print(line + "  presplit %.0f us %.2f TB/s" % (t_split, nbytes / t_split * 1e-6) + ("  bit-identical" if same else "  *** MISMATCH ***"), flush=True)
    setopt("p3_stage", 0)


if __name__ == "__main__":
    main()
