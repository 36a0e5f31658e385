"""Shared helpers for the parity tests: load a golden case, replay it through a codec pair."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ("x2_128_ipp", "x2_128_ip_wide", "x1_5_192_ip", "x2_128x256_ip")


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    frames, H, W, h, w, seed = (int(v) for v in z["meta"])
    scale, gain = (float(v) for v in z["scale_gain"])
    return z, dict(frames=frames, H=H, W=W, h=h, w=w, seed=seed, scale=scale, gain=gain)


def psnr(a, b):
    """RGB-PSNR exactly as test.py:115-118."""
    return (10 * torch.log10(1.0 / torch.mean((a - b) ** 2))).item()


def replay(case, i_frame, p_frame, device="cpu"):
    """Replay test.py's frame loop (test.py:182-250) over a golden case.
    i_frame(x_bl, x_el, (H,W)) -> dict like IntraSS.encode_decode;
    p_frame(x_bl, x_el, dpb, (H,W), scale) -> dict like LSSVC.encode_decode.
    Yields (t, result, dpb_after_clamp, psnr_bl, psnr_el)."""
    z, m = load_case(case)
    dpb = None
    for t in range(m["frames"]):
        x_el = (torch.from_numpy(z["x_el_u8"][t:t + 1]).float() / 255.0).to(device)
        x_bl = torch.from_numpy(z["x_bl"][t:t + 1]).to(device)
        if t == 0:
            r = i_frame(x_bl, x_el, (m["H"], m["W"]))
            dpb = {"ref_frame_bl": r["x_hat_bl"], "ref_frame_el": r["x_hat_el"], "ref_feature_bl": None,
                   "ref_feature_el": r["feature_el"]}
        else:
            r = p_frame(x_bl, x_el, dpb, (m["H"], m["W"]), m["scale"])
            dpb = r["dpb"]
        raw = {"x_hat_bl": dpb["ref_frame_bl"].clone(), "x_hat_el": dpb["ref_frame_el"].clone()}
        dpb["ref_frame_bl"].clamp_(0, 1)
        dpb["ref_frame_el"].clamp_(0, 1)
        yield t, r, raw, dpb, psnr(x_bl, dpb["ref_frame_bl"]), psnr(x_el, dpb["ref_frame_el"])
