"""Replay one golden case on the GPU and print per-frame bits / max recon error (debug aid)."""
import sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from helpers import load_case, replay
from lssvc_amd import IntraSS, LSSVC_extend
from lssvc_amd.synth import synth_state_dict

case = sys.argv[1]
z, m = load_case(case)
inet = IntraSS.from_state_dict(synth_state_dict("intra_ss", m["seed"], m["gain"])).to("cuda:0")
pnet = LSSVC_extend(); pnet.load_dict(synth_state_dict("lssvc_extend", m["seed"], m["gain"])); pnet.to("cuda:0").eval()
def i_fn(xb, xe, hr):
    inet.set_scale_information(m["scale"], hr, (0, 0, 0, 0))
    return inet.encode_decode(xb, xe, None, None, m["h"], m["w"], m["H"], m["W"])
def p_fn(xb, xe, dpb, hr, s):
    pnet.set_scale_information(s, hr, (0, 0, 0, 0))
    return pnet.encode_decode(xb, xe, dpb, None, None, m["W"], m["H"], m["w"], m["h"])
for t, r, raw, dpb, p_bl, p_el in replay(case, i_fn, p_fn, device="cuda:0"):
    bits = z["f%d_bits" % t]
    want = z["f%d_x_hat_el" % t]; got = raw["x_hat_el"].cpu().numpy()
    if want.shape != got.shape: got = got[:, :, ::2, ::2]
    d = np.abs(got - want)
    print(t, "bl %.4f/%.4f el %.4f/%.4f" % (r["bit_bl"], bits[0], r["bit_el"], bits[1]), "max|dx_el| %.3e n>1e-4 %d" % (d.max(), (d > 1e-4).sum()),
          "bl max %.3e" % np.abs(raw["x_hat_bl"].cpu().numpy() - z["f%d_x_hat_bl" % t]).max())
