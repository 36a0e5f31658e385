"""Launches of one steady-state P-frame by C entry point (and, for convs, by kernel shape class): compiles the plan at a small size
and reads the recorder's launch list.   python tools/plan_histogram.py [H W]"""
import collections
import os
import sys
import tempfile

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lssvc_amd import IntraSS, LSSVC_extend, plan_compiler, hip_ops  # noqa: E402
from lssvc_amd.synth import synth_state_dict  # noqa: E402


def main():
    H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 384)
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
    dpb = {k: v.contiguous().clone() for k, v in dpb.items()}
    names = []
    orig = plan_compiler.Recorder.save

    def save(self, *a, **k):
        names.extend(n for n, _, _ in self.launches)
        return orig(self, *a, **k)
    plan_compiler.Recorder.save = save
    hip_ops.OP_LOG = log = []
    with tempfile.TemporaryDirectory() as d:
        info, _ = plan_compiler.compile_pframe(pnet, xb[2], xe[2], dpb, os.path.join(d, "p.plan"))
    hip_ops.OP_LOG = None
    print("steady-P plan at %dx%d:" % (H, W), info)
    for n, c in collections.Counter(names).most_common():
        print("%5d  %s" % (c, n))
    kinds = collections.Counter((e["kind"], e["kernel"].split("<")[0]) for e in log[len(log) // 2:])      # the recorded pass (the warm-up pass is the first half)
    print("conv-class launches of the recorded pass by (op, kernel family):")
    for (kind, fam), c in kinds.most_common():
        print("%5d  %-14s %s" % (c, kind, fam))


main()
