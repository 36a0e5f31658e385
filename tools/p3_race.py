"""Race hunt for the persistent 3x3 kernel: one tiled-kernel reference, then N launches of the persistent kernel compared bit for
bit, with the location of any difference:  python tools/p3_race.py [N]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lssvc_amd import hip_ops as ops  # noqa: E402
from lssvc_amd._lib import lib, check  # noqa: E402
from lssvc_amd.weights import WeightStore  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    dev = torch.device("cuda:0")
    ops.set_conv_precision("f16x3")
    g = torch.Generator().manual_seed(0)
    for name, cin, cout, H, W in (("64->64 @576x960", 64, 64, 576, 960), ("64->64 @1152x1920", 64, 64, 1152, 1920), ("48->48 @576x960", 48, 48, 576, 960)):
        w = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
        b = torch.randn(cout, generator=g)
        Wt = WeightStore({"c.weight": w, "c.bias": b}, dev)
        x = ops.T(torch.randn(H * W * cin, device=dev), H, W, cin, cin)
        check(lib.lssvc_set_option(b"f16x3_persist", 0))
        ref = ops.conv(Wt, "c", x, in_act="lrelu", in_slope=0.1, act="lrelu", slope=0.01).buf.clone()
        check(lib.lssvc_set_option(b"f16x3_persist", 1))
        bad = 0
        for i in range(n):
            out = ops.conv(Wt, "c", x, in_act="lrelu", in_slope=0.1, act="lrelu", slope=0.01)
            if not torch.equal(out.buf, ref):
                bad += 1
                if bad <= 3:
                    d = (out.buf != ref).view(H, W, cout)
                    idx = d.nonzero()
                    ys, xs, cs = idx[:, 0], idx[:, 1], idx[:, 2]
                    print("  launch %d: %d elements differ; rows %d..%d cols %d..%d channels %d..%d; tiles (y/24, x/16): %s" % (
                        i, idx.shape[0], ys.min(), ys.max(), xs.min(), xs.max(), cs.min(), cs.max(),
                        sorted({(int(a) // 24, int(b_) // 16) for a, b_ in zip(ys.tolist()[:2000], xs.tolist()[:2000])})[:8]), flush=True)
        print("%s: %d of %d launches differ from the tiled kernel (%s)" % (name, bad, n, lib.lssvc_conv2d_last_kernel().decode()), flush=True)


if __name__ == "__main__":
    main()
