"""Where does the persistent 3x3 kernel's time go as the map shrinks?  64->64 (and 48->48) at map sizes giving 1 .. 23 tiles
per CU: time per launch vs rounds of tiles -> per-tile time and the fixed cost of a launch; and the same launch on fewer
workgroups (option p3_blocks).   python tools/p3_scaling.py [reps]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lssvc_amd import hip_ops as ops  # noqa: E402
from lssvc_amd._lib import lib, check  # noqa: E402
from lssvc_amd.weights import WeightStore  # noqa: E402


def setopt(name, v):
    check(lib.lssvc_set_option(name.encode(), v))


def bench(c, H, W, reps, blocks=0, res=False, in_act=None):
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    w = torch.randn(c, c, 3, 3, generator=g) / math.sqrt(c * 9)
    Wt = WeightStore({"c.weight": w, "c.bias": torch.randn(c, generator=g)}, dev)
    x = ops.T(torch.randn(H * W * c, device=dev), H, W, c, c)
    r = ops.T(torch.randn(H * W * c, device=dev), H, W, c, c) if res else None
    setopt("p3_blocks", blocks)
    out = ops.conv(Wt, "c", [x], residual=r, in_act=in_act, in_slope=0.1)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.conv(Wt, "c", [x], residual=r, in_act=in_act, in_slope=0.1, out=out)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    setopt("p3_blocks", 0)
    return sorted(ts)[2]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    ops.set_conv_precision("f16x3")
    setopt("f16x3_persist_min_tiles", 1)
    for c in (64, 48):
        print("---- %d->%d 3x3, tile 24x16, 256 CUs" % (c, c))
        for H, W in ((96, 1024), (192, 1024), (384, 1024), (288, 480), (576, 960), (576, 1024), (1152, 1024), (1152, 1920), (2304, 1920)):
            tiles = ((H + 23) // 24) * ((W + 15) // 16)
            t = bench(c, H, W, reps)
            rounds = math.ceil(tiles / 256)
            print("%4dx%-4d tiles %5d = %5.2f per CU (%2d rounds): %7.1f us  %6.1f TF  %5.2f us per round" % (
                H, W, tiles, tiles / 256, rounds, t, 2.0 * H * W * c * c * 9 / t * 1e-6, t / rounds))
        print("---- the same on fewer workgroups @576x960 (1440 tiles)")
        for blocks in (256, 240, 208, 192, 180, 160, 144, 128):
            t = bench(c, 576, 960, reps, blocks)
            print("blocks %3d: %5.2f tiles per WG (max %d): %7.1f us" % (blocks, 1440 / blocks, math.ceil(1440 / blocks), t))
        print("---- @576x960 with residual / input lrelu")
        print("residual: %7.1f us   lrelu in: %7.1f us" % (bench(c, 576, 960, reps, res=True), bench(c, 576, 960, reps, in_act="lrelu")))


main()
