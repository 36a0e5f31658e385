"""For one golden case, find which coded symbols of the first P frame differ between the GPU path and the
CPU oracle, and how close to a rounding tie the oracle's pre-round value was (debug aid)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in ("tests", "", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, _p))
from helpers import load_case
from lssvc_amd import IntraSS, LSSVC_extend
from lssvc_amd.hip_ops import T
from lssvc_amd.inter import CHUNK_OF_MASK
from lssvc_amd.synth import synth_state_dict
from lssvc_oracle.intra import intra_forward
from lssvc_oracle.inter import inter_forward

case = sys.argv[1]
z, m = load_case(case)
H, W = m["H"], m["W"]
sd_i, sd_p = synth_state_dict("intra_ss", m["seed"], m["gain"]), synth_state_dict("lssvc_extend", m["seed"], m["gain"])
inet = IntraSS.from_state_dict(sd_i).to("cuda:0").eval()
pnet = LSSVC_extend(); pnet.load_dict(sd_p); pnet.to("cuda:0").eval()
inet.update(force=True); pnet.update(force=True)
inet.set_scale_information(m["scale"], (H, W), (0, 0, 0, 0)); pnet.set_scale_information(m["scale"], (H, W), (0, 0, 0, 0))
x_el = torch.from_numpy(z["x_el_u8"][0:2]).float() / 255.0
x_bl = torch.from_numpy(z["x_bl"][0:2])
with torch.no_grad():
    oi = intra_forward(sd_i, x_bl[0:1], x_el[0:1], (H, W), extras=True)
    dpo = {"ref_frame_bl": oi["x_hat_bl"].clamp(0, 1), "ref_frame_el": oi["x_hat_el"].clamp(0, 1), "ref_feature_bl": None,
           "ref_feature_el": oi["feature_el"]}
    op = inter_forward(sd_p, x_bl[1:2], x_el[1:2], dpo, (H, W), m["scale"], extras=True)
xb, xe = x_bl.cuda(), x_el.cuda()
ri = inet.encode_decode(xb[0:1], xe[0:1], None, None)
dpb = {"ref_frame_bl": ri["x_hat_bl"].clamp(0, 1), "ref_frame_el": ri["x_hat_el"].clamp(0, 1), "ref_feature_el": ri["feature_el"]}
class Rec:
    def __init__(self): self.items = []
    def push(self, s, i, t): self.items.append((s.copy(), i.copy()))
    def flush(self): return b""
nh = T.from_nchw
bl = pnet._bl_codec(nh(xb[1:2]), nh(dpb["ref_frame_bl"]), None)
rec = Rec()
pnet._el_codec(nh(xe[1:2]), bl, nh(dpb["ref_frame_el"]), nh(dpb["ref_feature_el"]), sink=rec)
pre = op["pre_round"]
def report(name, got, want_pre):
    want_pre = want_pre.reshape(-1).numpy()
    want = np.round(want_pre).astype(np.int32)   # torch.round is half-to-even like numpy
    bad = np.nonzero(got != want)[0]
    print("%-8s n=%d differ=%d" % (name, got.size, bad.size), end="")
    for i in bad[:6]:
        f = want_pre[i] - np.floor(want_pre[i])
        print("  [%d: pre=%.7f (tie dist %.2e) got %d]" % (i, want_pre[i], abs(f - 0.5), got[i]), end="")
    print()
report("mv_z", rec.items[0][0], pre["mv_z"])
report("mv_y", rec.items[1][0], pre["mv_y"])
report("z", rec.items[2][0], pre["z"])
yr = pre["y"]
for step in range(4):
    fold = torch.zeros(1, 32, yr.shape[2], yr.shape[3])
    for mi, (r, c) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
        ch = CHUNK_OF_MASK[step][mi]
        fold[:, :, r::2, c::2] = yr[:, ch * 32:(ch + 1) * 32, r::2, c::2]
    report("y_w%d" % step, rec.items[3 + step][0], fold)
