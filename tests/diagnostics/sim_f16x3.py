"""CPU simulation of the f16x3 conv arithmetic inside the oracle: how far do the quantiser inputs move?

Patches torch.nn.functional.conv2d (dense, stride-1 3x3/7x7/1x1 only - what the HIP path runs in f16x3) so that
both operands are split into fp16 hi + lo and the three kept products are accumulated, with optional power-of-two
prescaling of x (sx) and w (per layer, max|w| -> 2^wexp).  Reports max |pre_round - exact| on one golden case's
first P frame.  usage: sim_f16x3.py CASE [sx_log2 [wexp]]   (omit scales for the unscaled split)"""
import os, sys, numpy as np, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in ("tests", "", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, _p))
from helpers import load_case
from lssvc_amd.synth import synth_state_dict
from lssvc_oracle.intra import intra_forward
from lssvc_oracle.inter import inter_forward

case = sys.argv[1]
SX = int(sys.argv[2]) if len(sys.argv) > 2 else None
WEXP = int(sys.argv[3]) if len(sys.argv) > 3 else None
real_conv2d = F.conv2d
MODE = {"on": False}

def split(t):
    t = t.clamp(-65504, 65504)
    hi = t.half().float()
    lo = (t - hi).half().float()
    return hi, lo

def sim_conv2d(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
    k = w.shape[-1]
    if not MODE["on"] or groups != 1 or stride not in (1, (1, 1)) or k not in (1, 3, 7) or x.shape[1] % 4:
        return real_conv2d(x, w, b, stride, padding, dilation, groups)
    if SX == -99:   # baseline: exact products, fp64 accumulate (the noise floor of any fp32 re-ordering)
        out = real_conv2d(x.double(), w.double(), None, stride, padding).float()
        return out if b is None else out + b.view(1, -1, 1, 1)
    sx = 2.0 ** SX if SX is not None else 1.0
    sw = 1.0
    if WEXP is not None:
        sw = 2.0 ** (WEXP - int(np.ceil(np.log2(float(w.abs().max()) + 1e-30))))
    xh, xl = split(x * sx)
    wh, wl = split(w * sw)
    d = torch.float64
    acc = (real_conv2d(xh.to(d), wl.to(d), None, stride, padding) + real_conv2d(xl.to(d), wh.to(d), None, stride, padding)
           + real_conv2d(xh.to(d), wh.to(d), None, stride, padding))
    out = (acc / (sx * sw)).float()
    if b is not None:
        out = out + b.view(1, -1, 1, 1)
    return out

F.conv2d = sim_conv2d
torch.nn.functional.conv2d = sim_conv2d

z, m = load_case(case)
H, W = m["H"], m["W"]
sd_i, sd_p = synth_state_dict("intra_ss", m["seed"], m["gain"]), synth_state_dict("lssvc_extend", m["seed"], m["gain"])
x_el = torch.from_numpy(z["x_el_u8"][0:2]).float() / 255.0
x_bl = torch.from_numpy(z["x_bl"][0:2])

def run():
    with torch.no_grad():
        oi = intra_forward(sd_i, x_bl[0:1], x_el[0:1], (H, W), extras=True)
        dpo = {"ref_frame_bl": oi["x_hat_bl"].clamp(0, 1), "ref_frame_el": oi["x_hat_el"].clamp(0, 1), "ref_feature_bl": None,
               "ref_feature_el": oi["feature_el"]}
        return inter_forward(sd_p, x_bl[1:2], x_el[1:2], dpo, (H, W), m["scale"], extras=True)

exact = run()
MODE["on"] = True
sim = run()
for k in ("mv_z", "mv_y", "z", "y"):
    a, b = exact["pre_round"][k], sim["pre_round"][k]
    print("%-5s max|d| %.3e  rms %.3e   (max|v| %.2f)" % (k, (a - b).abs().max(), (a - b).pow(2).mean().sqrt(), a.abs().max()))
print("bits el exact %.4f sim %.4f" % (exact["bit_el"], sim["bit_el"]))
