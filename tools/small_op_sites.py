"""Where do the small non-conv launches of a P-frame come from?  Wraps the elementwise entry points of hip_ops during one steady-state
P-frame (eager, no plan) and prints (op, shape, caller) with counts.   python tools/small_op_sites.py"""
import collections
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lssvc_amd import IntraSS, LSSVC_extend, hip_ops  # noqa: E402
from lssvc_amd._lib import lib  # noqa: E402
from lssvc_amd.synth import synth_state_dict  # noqa: E402

LOG = collections.Counter()


def caller():
    for f in reversed(traceback.extract_stack()[:-2]):
        if "small_op_sites" not in f.filename and not f.filename.endswith("hip_ops.py"):
            return "%s:%d" % (os.path.basename(f.filename), f.lineno)
    return "?"


def wrap_lib(name):
    orig = getattr(lib, name)

    def w(*a):
        LOG[(name, caller())] += 1
        return orig(*a)
    setattr(lib, name, w)


def main():
    H, W = 256, 384
    dev = torch.device("cuda:0")
    inet = IntraSS.from_state_dict(synth_state_dict("intra_ss", 0, 0.55)).to(dev).eval()
    pnet = LSSVC_extend()
    pnet.load_dict(synth_state_dict("lssvc_extend", 0, 0.55))
    pnet.to(dev).eval()
    g = torch.Generator().manual_seed(0)
    xe = [torch.rand(1, 3, H, W, generator=g).to(dev) for _ in range(3)]
    xb = [torch.rand(1, 3, H // 2, W // 2, generator=g).to(dev) for _ in range(3)]
    for n in (inet, pnet):
        n.set_scale_information(2.0, (H, W), (0, 0, 0, 0))
    r = inet.encode_decode(xb[0], xe[0], None, None)
    dpb = {"ref_frame_bl": r["x_hat_bl"].clamp(0, 1), "ref_frame_el": r["x_hat_el"].clamp(0, 1), "ref_feature_bl": None, "ref_feature_el": r["feature_el"]}
    dpb = pnet.encode_decode(xb[1], xe[1], dpb)["dpb"]
    for name in ("lssvc_copy", "lssvc_fill_zero", "lssvc_lrelu", "lssvc_nchw_to_nhwc", "lssvc_nhwc_to_nchw", "lssvc_pool2x2", "lssvc_resize_bilinear",
                 "lssvc_flow_warp", "lssvc_add", "lssvc_softmax2_blend", "lssvc_dwconv3x3"):
        wrap_lib(name)
    pnet.encode_decode(xb[2], xe[2], dpb)
    torch.cuda.synchronize()
    for (name, site), c in sorted(LOG.items()):
        print("%3d  %-24s %s" % (c, name, site))
    print("total", sum(LOG.values()))


main()
