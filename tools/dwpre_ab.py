"""A/B of the fused 1x1 + depthwise kernel's two prefetch schedules (option dwpre_deep) on the benchmark's shapes, interleaved
rounds in ONE process, bit-identity checked:  python tools/dwpre_ab.py [rounds] [reps]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lssvc_amd import hip_ops as ops  # noqa: E402
from lssvc_amd._lib import lib, check  # noqa: E402
from lssvc_amd.weights import WeightStore  # noqa: E402

SHAPES = [("64->64 @1152x1920", 64, 64, 1152, 1920), ("48->48 @1152x1920", 48, 48, 1152, 1920), ("32->32 @1152x1920", 32, 32, 1152, 1920),
          ("64->64 @576x960", 64, 64, 576, 960), ("48->32 @576x960", 48, 32, 576, 960)]


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dev = torch.device("cuda:0")
    ops.set_conv_precision("f16x3")
    g = torch.Generator().manual_seed(0)
    for name, cin, c, H, W in SHAPES:
        sd = {"a.weight": torch.randn(c, cin, 1, 1, generator=g) / math.sqrt(cin), "a.bias": torch.randn(c, generator=g) * 0.1,
              "d.weight": torch.randn(c, 1, 3, 3, generator=g) / 3, "d.bias": torch.randn(c, generator=g) * 0.1}
        Wt = WeightStore(sd, dev)
        x = ops.T(torch.randn(H * W * cin, device=dev), H, W, cin, cin)
        outs, times = {}, {0: [], 1: []}
        for mode in (0, 1):
            check(lib.lssvc_set_option(b"dwpre_deep", mode))
            outs[mode] = ops.conv1x1_dw3x3(Wt, "a", "d", x, slope=0.01)
            assert outs[mode] is not None
        torch.cuda.synchronize()
        same = torch.equal(outs[0].buf, outs[1].buf)
        for _ in range(rounds):
            for mode in (0, 1):
                check(lib.lssvc_set_option(b"dwpre_deep", mode))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    ops.conv1x1_dw3x3(Wt, "a", "d", x, slope=0.01, out=outs[mode])
                e1.record()
                torch.cuda.synchronize()
                times[mode].append(e0.elapsed_time(e1) / reps)
        nbytes = 4.0 * H * W * (cin + c)
        line = "%-22s" % name
        for mode, tag in ((0, "phase-2 prefetch"), (1, "per-group early")):
            t = sorted(times[mode])[len(times[mode]) // 2]
            line += "  %s %7.1f us %5.2f TB/s" % (tag, t * 1e3, nbytes / t / 1e9)
        print(line + ("  bit-identical" if same else "  *** MISMATCH ***"), flush=True)
    check(lib.lssvc_set_option(b"dwpre_deep", 1))


if __name__ == "__main__":
    main()
