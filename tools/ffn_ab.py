"""A/B of the fused DepthConvBlock tail's store layout (option ffn_tstore: 1 = coalescing lane layout, 0 = MFMA layout), interleaved
rounds in one process, with a bit-identity check:  python tools/ffn_ab.py [rounds] [reps]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lssvc_amd import hip_ops as ops  # noqa: E402
from lssvc_amd._lib import lib, check  # noqa: E402
from lssvc_amd.weights import WeightStore  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    dev = torch.device("cuda:0")
    ops.set_conv_precision("f16x3")
    for c, hidden, H, W, pre, skip in ((64, 256, 1152, 1920, True, False), (64, 256, 1152, 1920, True, True), (48, 192, 1152, 1920, True, False),
                                       (64, 256, 1152, 1920, False, False), (128, 512, 288, 480, True, False), (96, 384, 576, 960, True, False),
                                       (64, 256, 301, 333, True, True)):
        g = torch.Generator().manual_seed(c)
        sd = {"f.conv.0.weight": torch.randn(hidden, c, 1, 1, generator=g) / math.sqrt(c), "f.conv.0.bias": torch.randn(hidden, generator=g) * 0.1,
              "f.conv.2.weight": torch.randn(c, hidden, 1, 1, generator=g) / math.sqrt(hidden), "f.conv.2.bias": torch.randn(c, generator=g) * 0.1,
              "p.weight": torch.randn(c, c, 1, 1, generator=g) / math.sqrt(c), "p.bias": torch.randn(c, generator=g) * 0.1}
        Wt = WeightStore(sd, dev)
        t = ops.T.from_nchw(torch.randn(1, c, H, W, device=dev))
        ident = ops.T.from_nchw(torch.randn(1, c, H, W, device=dev))
        sk = ops.T.from_nchw(torch.randn(1, c, H, W, device=dev)) if skip else None
        outs = {0: ident.like(), 1: ident.like()}

        def run(mode):
            check(lib.lssvc_set_option(b"ffn_tstore", mode))
            if pre:
                ops.ffn_block(Wt, "f", pre_name="p", pre_in=t, ident=ident, out=outs[mode], skip=sk)
            else:
                ops.ffn_block(Wt, "f", x=ident, out=outs[mode], skip=sk)

        times = {0: [], 1: []}
        for m in (0, 1):
            outs[m].buf.zero_()
            run(m)
        torch.cuda.synchronize()
        same = torch.equal(outs[0].buf, outs[1].buf)
        for _ in range(rounds):
            for m in (1, 0):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    run(m)
                e1.record()
                torch.cuda.synchronize()
                times[m].append(e0.elapsed_time(e1) / reps)
        med = {m: sorted(v)[len(v) // 2] for m, v in times.items()}
        print("C=%3d hidden=%4d @%dx%d %s%s: coalescing %7.1f us   native %7.1f us   %s" % (
            c, hidden, H, W, "pre+ffn" if pre else "ffn", "+skip" if skip else "", med[1] * 1e3, med[0] * 1e3,
            "bit-identical" if same else "*** MISMATCH ***"), flush=True)
    check(lib.lssvc_set_option(b"ffn_tstore", 1))


if __name__ == "__main__":
    main()
