"""Round-6 A/B of the under-performing 3x3 families (VERDICT r5 item 1 a-c), same process, interleaved rounds, random data:
    python tools/r6_families_ab.py narrow|small|small_sweep|mid|mid_sweep|s2|gdn [rounds] [reps]

  narrow       <= 16 output channels (flow / picture heads): conv3n (16x16 tiles, 2 workgroups per CU, register prefetch) vs the tiled kernel
  small        maps with < 256 tiles of 24x16: the small-tile persistent instantiations (auto-picked) vs the tiled kernel
  small_sweep  the same shapes over every forced (MF, rows per wave) pair, with and without the register prefetch: the data the
               dispatcher's cost model (conv3_f16x3p.hip: p3_pick_small) is fitted to
  mid[_sweep]  maps with 256 ... 1500 tiles of 24x16 (where the 24x16 tiling stays: the table says why)
  s2           stride 2: the register prefetch (p3_pf2) on / off
  gdn          GDN / IGDN 1x1 kernels: the lean normalising epilogue (conv_epilogue_gdn) vs the general one

Every arm's output is compared bit for bit with the first arm's."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lssvc_amd import hip_ops as ops  # noqa: E402
from lssvc_amd._lib import lib, check  # noqa: E402
from lssvc_amd.weights import WeightStore  # noqa: E402

DEFAULTS = {"p3_small": 1, "p3_narrow": 1, "p3_pf2": 1, "p3_force": 0, "p3_big_pair": 0, "f16x3_persist": 1, "f16x3_persist_s2": 1, "gdn_fast": 1, "p7_narrow": 0}


def set_opts(**kw):
    for k, v in {**DEFAULTS, **kw}.items():
        check(lib.lssvc_set_option(k.encode(), v))


def run_arms(label, make_call, arms, rounds, reps, flops, mbytes):
    """arms: list of (name, option dict). Prints the median time of every arm and whether its output equals arm 0's."""
    outs, names, times = [], [], [[] for _ in arms]
    for _, o in arms:
        set_opts(**o)
        outs.append(make_call(None))
        names.append(lib.lssvc_conv2d_last_kernel().decode())
    torch.cuda.synchronize()
    same = [torch.equal(outs[0].buf, o.buf) for o in outs]
    for rnd in range(rounds):
        for i0 in range(len(arms)):                  # the arm that opens a round runs on a chip that has just idled: rotate it
            i = (i0 + rnd) % len(arms)
            o = arms[i][1]
            set_opts(**o)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                make_call(outs[i])
            e1.record()
            torch.cuda.synchronize()
            times[i].append(e0.elapsed_time(e1) / reps * 1e3)
    set_opts()
    med = [sorted(t)[len(t) // 2] for t in times]
    print(label)
    for i, (an, _) in enumerate(arms):
        print("    %-22s %8.1f us  %6.1f TF  %5.2f TB/s  x%.2f  %s  %s" % (an, med[i], flops / med[i] * 1e-6, mbytes / med[i], med[0] / med[i],
                                                                          "bit-identical" if same[i] else "DIFFERENT", names[i]), flush=True)
    return med, same


def conv_case(cins, cout, H, W, stride, g, dev, **kw):
    cin = sum(cins)
    w = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
    Wt = WeightStore({"c.weight": w, "c.bias": torch.randn(cout, generator=g)}, dev)
    xs = [ops.T(torch.randn(H * W * c, device=dev), H, W, c, c) for c in cins]
    res = None
    if kw.pop("residual", False):
        res = ops.T(torch.randn((H // stride) * (W // stride) * cout, device=dev), H // stride, W // stride, cout, cout)
    return lambda out: ops.conv(Wt, "c", xs, stride=stride, out=out, residual=res, **kw)


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "narrow"
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    dev = torch.device("cuda:0")
    ops.set_conv_precision("f16x3")
    g = torch.Generator().manual_seed(0)
    if what == "narrow":
        for cins, cout, H, W, kw in (([64], 2, 1152, 1920, {}), ([64], 2, 1152, 1920, {"residual": True}), ([48], 3, 1152, 1920, {}), ([64], 3, 576, 960, {}),
                                     ([64], 8, 576, 960, {}), ([64], 2, 576, 960, {}), ([64], 2, 288, 480, {}), ([32, 32], 16, 1152, 1920, {"act": "lrelu"})):
            cin = sum(cins)
            call = conv_case(cins, cout, H, W, 1, g, dev, **kw)
            run_arms("3x3 %s->%d @%dx%d %s" % (cins, cout, H, W, kw), call, [("tiled / p3 rpw 6", {"p3_narrow": 0}), ("conv3n", {"p3_pf2": 0}), ("conv3n pf2", {"p3_pf2": 4}), ("conv3n roles", {"p3_pf2": 3})], rounds, reps,
                     2.0 * H * W * cout * 9 * cin, 4e-6 * H * W * (cin + cout))
    elif what in ("small", "small_sweep", "mid", "mid_sweep"):
        mid = (([64], 64, 288, 480, {}), ([64], 64, 288, 480, {"in_act": "lrelu", "in_slope": 0.1, "residual": True}), ([96], 96, 288, 480, {}), ([128], 128, 288, 480, {}),
               ([64], 128, 288, 480, {}), ([128], 64, 288, 480, {}), ([96], 192, 288, 480, {}), ([96], 256, 288, 480, {"pixel_shuffle": True}), ([192], 96, 288, 480, {}),
               ([64], 256, 144, 240, {}), ([96], 384, 144, 240, {"pixel_shuffle": True}), ([128], 256, 144, 240, {}), ([64], 64, 576, 960, {}))
        shapes = mid if what.startswith("mid") else (([64], 64, 144, 240, {}), ([64], 64, 144, 240, {"in_act": "lrelu", "in_slope": 0.1, "residual": True}), ([128], 128, 72, 120, {}),
                  ([128], 128, 144, 240, {}), ([64], 128, 144, 240, {}), ([128], 64, 144, 240, {}), ([96], 96, 36, 60, {}), ([128], 384, 72, 120, {}),
                  ([384], 320, 36, 60, {}), ([96], 128, 72, 120, {}), ([64], 256, 72, 120, {"pixel_shuffle": True}), ([192], 256, 36, 60, {}), ([64], 64, 288, 480, {}),
                  ([96], 96, 288, 480, {}), ([64], 64, 72, 120, {}))
        for cins, cout, H, W, kw in shapes:
            cin = sum(cins)
            call = conv_case(cins, cout, H, W, 1, g, dev, **kw)
            arms = [("r5: tiled / 24x16", {"p3_small": 0}), ("auto", {}), ("auto, pf2 always", {"p3_small": 2}), ("auto, pf2 never", {"p3_small": 3})]
            if what.endswith("_sweep"):
                for mf in (4, 3, 2, 1):
                    if mf > (cout + 15) // 16 or ((cout + 15) // 16) % mf:
                        continue
                    for rpw in (4, 2, 1):
                        arms.append(("mf %d rpw %d" % (mf, rpw), {"p3_force": mf * 16 + rpw}))
                        arms.append(("mf %d rpw %d pf2" % (mf, rpw), {"p3_force": mf * 16 + rpw, "p3_small": 2}))
            run_arms("3x3 %s->%d @%dx%d %s" % (cins, cout, H, W, kw), call, arms, rounds, reps, 2.0 * H * W * cout * 9 * cin, 4e-6 * H * W * (cin + cout))
    elif what == "s2":
        for cins, cout, H, W in (([48], 64, 1152, 1920), ([56], 64, 1152, 1920), ([4], 64, 1152, 1920), ([64], 64, 576, 960), ([64], 96, 576, 960), ([128], 96, 576, 960),
                                 ([96], 128, 288, 480), ([4, 48], 64, 1152, 1920), ([128], 128, 288, 480)):
            cin = sum(cins)
            call = conv_case(cins, cout, H, W, 2, g, dev, act="lrelu", slope=0.1)
            run_arms("3x3 s2 %s->%d in %dx%d" % (cins, cout, H, W), call, [("one register set", {"p3_pf2": 0}), ("register prefetch", {"p3_pf2": 4}), ("pair loads", {"p3_pf2": 2}), ("split roles", {"p3_pf2": 3}), ("tiled", {"f16x3_persist_s2": 0})],
                     rounds, reps, 2.0 * (H // 2) * (W // 2) * cout * 9 * cin, 4e-6 * (H * W * cin + (H // 2) * (W // 2) * cout))
    elif what == "ablate":                 # tools/r6_roles_ablation.sh: the split-roles kernels only, whatever library LSSVC_HIP_LIB names
        for cins, cout, H, W, stride, kw in (([64], 2, 1152, 1920, 1, {}), ([48], 3, 1152, 1920, 1, {}), ([32, 32], 16, 1152, 1920, 1, {"act": "lrelu"}),
                                             ([48], 64, 1152, 1920, 2, {"act": "lrelu", "slope": 0.1}), ([64], 64, 576, 960, 2, {"act": "lrelu", "slope": 0.1})):
            cin = sum(cins)
            call = conv_case(cins, cout, H, W, stride, g, dev, **kw)
            run_arms("3x3 s%d %s->%d in %dx%d" % (stride, cins, cout, H, W), call, [("split roles", {"p3_pf2": 3})], rounds, reps,
                     2.0 * (H // stride) * (W // stride) * cout * 9 * cin, 4e-6 * (H * W * cin + (H // stride) * (W // stride) * cout))
    elif what == "big":
        for cins, cout, H, W, kw in (([64], 64, 1152, 1920, {}), ([64], 64, 576, 960, {}), ([64], 64, 576, 960, {"in_act": "lrelu", "in_slope": 0.1, "residual": True}), ([128], 64, 576, 960, {}),
                                     ([128], 192, 576, 960, {}), ([64], 64, 288, 480, {}), ([48], 48, 1152, 1920, {}), ([48], 48, 1152, 1920, {"in_act": "lrelu", "in_slope": 0.1, "residual": True}),
                                     ([96], 48, 1152, 1920, {}), ([96], 96, 288, 480, {}), ([64], 32, 1152, 1920, {})):
            cin = sum(cins)
            call = conv_case(cins, cout, H, W, 1, g, dev, **kw)
            arms = [("r5 schedule", {"p3_big_pair": 3}), ("split roles", {"p3_big_pair": 2}), ("late loads", {"p3_big_pair": 4}), ("default", {})]
            if cout == 64 and not kw and cin in (64, 128):
                arms.append(("pair loads", {"p3_big_pair": 1}))      # (MF = 4, no input activation: the one instantiation kept)
            run_arms("3x3 %s->%d @%dx%d %s" % (cins, cout, H, W, kw), call, arms, rounds, reps, 2.0 * H * W * cout * 9 * cin, 4e-6 * H * W * (cin + cout))
    elif what == "late":                   # which epilogue / input forms the late-loads schedule suits (MF = 4, 24x16 tiles)
        for cins, cout, H, W, kw in (([64], 64, 576, 960, {}), ([64], 64, 576, 960, {"in_act": "lrelu", "in_slope": 0.1}), ([64], 64, 576, 960, {"residual": True}),
                                     ([64], 64, 576, 960, {"act": "lrelu"}), ([64], 64, 576, 960, {"in_act": "lrelu", "in_slope": 0.1, "residual": True}),
                                     ([64], 64, 1152, 1920, {"residual": True}), ([64], 64, 1152, 1920, {"in_act": "lrelu", "in_slope": 0.1}), ([64], 64, 384, 640, {}), ([64], 64, 288, 480, {}),
                                     ([128], 128, 288, 480, {}), ([64], 128, 576, 960, {}), ([96], 64, 1152, 1920, {}), ([64], 256, 288, 480, {"pixel_shuffle": True})):
            cin = sum(cins)
            call = conv_case(cins, cout, H, W, 1, g, dev, **kw)
            run_arms("3x3 %s->%d @%dx%d %s" % (cins, cout, H, W, kw), call, [("r5 schedule", {"p3_big_pair": 3}), ("late loads", {"p3_big_pair": 4})], rounds, reps,
                     2.0 * H * W * cout * 9 * cin, 4e-6 * H * W * (cin + cout))
    elif what == "p7n":
        for cins, cout, H, W in (([32], 16, 1152, 1920), ([32], 16, 576, 960), ([32], 16, 288, 480), ([8], 16, 1152, 1920)):
            cin = sum(cins)
            w = torch.randn(cout, cin, 7, 7, generator=g) / math.sqrt(cin * 49)
            Wt = WeightStore({"c.weight": w, "c.bias": torch.randn(cout, generator=g)}, dev)
            xs = [ops.T(torch.randn(H * W * c, device=dev), H, W, c, c) for c in cins]
            call = lambda out: ops.conv(Wt, "c", xs, act="relu", out=out)
            run_arms("7x7 %s->%d @%dx%d" % (cins, cout, H, W), call, [("tiled", {}), ("persistent MF = 1", {"p7_narrow": 1})], rounds, reps, 2.0 * H * W * cout * 49 * cin, 4e-6 * H * W * (cin + cout))
    elif what == "tall":
        for cins, cout, H, W, kw in (([48], 48, 1152, 1920, {}), ([48], 48, 1152, 1920, {"in_act": "lrelu", "in_slope": 0.1, "residual": True}), ([48, 48], 48, 1152, 1920, {}),
                                     ([64, 16], 48, 1152, 1920, {}), ([96], 96, 288, 480, {}), ([192], 96, 288, 480, {"in_act": "lrelu", "in_slope": 0.1}), ([64], 48, 1152, 1920, {}),
                                     ([64], 64, 1152, 1920, {}), ([64], 64, 576, 960, {})):
            cin = sum(cins)
            mf = 3 if cout % 48 == 0 else 4
            call = conv_case(cins, cout, H, W, 1, g, dev, **kw)
            arms = [("r5: 24x16 tiles", {}), ("32x16 tiles", {"p3_force": mf * 16 + 8, "p3_small": 3})] + ([("32x16 tiles, pf2", {"p3_force": mf * 16 + 8, "p3_small": 2})] if mf == 3 else [])
            run_arms("3x3 %s->%d @%dx%d %s" % (cins, cout, H, W, kw), call, arms, rounds, reps, 2.0 * H * W * cout * 9 * cin, 4e-6 * H * W * (cin + cout))
    elif what == "gdn":
        from lssvc_amd.synth import _make
        for c, H, W, flavour, inverse, res in ((64, 576, 960, "inter", False, False), (64, 576, 960, "inter", True, False), (64, 288, 480, "inter", False, False),
                                               (64, 144, 240, "inter", False, False), (128, 288, 480, "inter", False, False), (128, 144, 240, "inter", True, False),
                                               (96, 144, 240, "intra", False, True), (64, 576, 960, "intra", True, True), (192, 72, 120, "intra", False, True)):
            sd = {"g.beta": _make({"key": "g.beta", "shape": [c], "kind": "gdn_beta"}, 3, 1.0),
                  "g.gamma": _make({"key": "g.gamma", "shape": [c, c], "kind": "gdn_gamma"}, 3, 1.0),
                  "g.beta_reparam.pedestal": torch.tensor([2.0 ** -36]), "g.gamma_reparam.pedestal": torch.tensor([2.0 ** -36]),
                  "g.beta_reparam.lower_bound.bound": torch.tensor([(1e-6 + 2.0 ** -36) ** 0.5]),
                  "g.gamma_reparam.lower_bound.bound": torch.tensor([2.0 ** -18])}
            Wt = WeightStore(sd, dev)
            x = ops.T(torch.randn(H * W * c, device=dev) * 2, H, W, c, c)
            r = ops.T(torch.randn(H * W * c, device=dev), H, W, c, c) if res else None
            call = lambda out: ops.gdn(Wt, "g", x, flavour, inverse=inverse, residual=r, out=out)
            run_arms("%s %s %d ch @%dx%d%s" % ("IGDN" if inverse else "GDN", flavour, c, H, W, " + residual" if res else ""), call,
                     [("general epilogue", {"gdn_fast": 0}), ("lean epilogue", {})], rounds, reps, 2.0 * H * W * c * c, 4e-6 * H * W * c * (3 if res else 2))
    else:
        raise SystemExit(__doc__)


main()
