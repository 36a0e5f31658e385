"""Generate golden fixtures by running the REFERENCE (imported from /root/reference, build
container only) on seeded synthetic weights and clips:

    python tests/golden/make_golden.py

Each case replays test.py's per-frame loop (test.py:182-250): BL frame by the reference's own
bicubic imresize + clamp, I-frame through IntraSS.encode_decode, P-frames through
LSSVC_extend.encode_decode, in-place clamp of the DPB frames, RGB-PSNR as test.py:115-118.
Fixtures hold data only (inputs, expected outputs); weights are re-drawn at test time by
lssvc_amd.synth from (manifest, seed, gain).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

from ref_import import import_reference  # noqa: E402
from lssvc_amd.synth import synth_state_dict, synth_clip  # noqa: E402

CASES = {
    # name: (frames, H_el, W_el, H_bl, W_bl, scale, gain, seed)
    "x2_128_ipp": (3, 128, 128, 64, 64, 2.0, 0.6, 0),
    "x2_128_ip_wide": (2, 128, 128, 64, 64, 2.0, 0.65, 1),
    "x1_5_192_ip": (2, 192, 192, 128, 128, 1.5, 0.6, 2),
    "x2_128x256_ip": (2, 128, 256, 64, 128, 2.0, 0.6, 3),
    # a non-zero inter-layer pad_size (test.py always passes zeros; IntraSS.py:124-147, LSSVC_net.py:271-282,454-456): the BL
    # codes a 128x128 picture of which the EL uses the top-left 64x64 (negative padding = crop; /16 on the latent grid)
    "x2_128_ip_depad": (2, 128, 128, 128, 128, 2.0, 0.6, 4),
}
PAD_SIZE = {"x2_128_ip_depad": (0, -64, 0, -64)}


def psnr(a, b):
    return (10 * torch.log10(1.0 / torch.mean((a - b) ** 2))).item()


def sub(t, s):
    return t[:, :, ::s, ::s].contiguous().numpy()


def run_case(name, IntraSS, LSSVC_extend, imresize):
    frames, H, W, h, w, scale, gain, seed = CASES[name]
    sd_i = synth_state_dict("intra_ss", seed, gain)
    sd_p = synth_state_dict("lssvc_extend", seed, gain)
    inet = IntraSS.from_state_dict(dict(sd_i)).eval()
    pnet = LSSVC_extend()
    pnet.load_dict(sd_p)
    pnet.eval()
    clip = synth_clip(frames, H, W, seed=seed)
    pad_size = PAD_SIZE.get(name, (0, 0, 0, 0))
    out = {"x_el_u8": clip.numpy(), "meta": np.array([frames, H, W, h, w, seed], dtype=np.int64),
           "scale_gain": np.array([scale, gain], dtype=np.float64), "pad_size": np.array(pad_size, dtype=np.int64)}
    x_bls = []
    dpb = None
    with torch.no_grad():
        for t in range(frames):
            x_el = clip[t:t + 1].float() / 255.0
            x_bl = imresize(x_el, sizes=(h, w), kernel="cubic").clamp_(0, 1)
            x_bls.append(x_bl)
            inet.set_scale_information(scale, (H, W), pad_size)
            pnet.set_scale_information(scale, (H, W), pad_size)
            if t == 0:
                r = inet.encode_decode(x_bl, x_el, None, None, h, w, H, W)
                dpb = {"ref_frame_bl": r["x_hat_bl"], "ref_frame_el": r["x_hat_el"], "ref_feature_bl": None,
                       "ref_feature_el": r["feature_el"]}
            else:
                r = pnet.encode_decode(x_bl, x_el, dpb, None, None, W, H, w, h)
                dpb = r["dpb"]
                out["f%d_mv_hat" % t] = sub(r["mv_hat"], 2)
                out["f%d_warp_frame" % t] = sub(r["warp_frame"], 2)
                out["f%d_feature_bl" % t] = sub(dpb["ref_feature_bl"], 4)
                out["f%d_feature_bl_sum" % t] = np.array([dpb["ref_feature_bl"].double().sum().item(),
                                                          dpb["ref_feature_bl"].double().abs().sum().item()])
            out["f%d_bits" % t] = np.array([r["bit_bl"], r["bit_el"]], dtype=np.float64)
            out["f%d_x_hat_bl" % t] = dpb["ref_frame_bl"].numpy().copy()       # un-clamped, as returned
            # EL frame: full on the last frame, every 2nd pixel otherwise (keeps fixtures small)
            out["f%d_x_hat_el" % t] = (dpb["ref_frame_el"].numpy().copy() if t == frames - 1
                                       else sub(dpb["ref_frame_el"], 2))
            out["f%d_feature_el" % t] = sub(dpb["ref_feature_el"], 8)
            out["f%d_feature_el_sum" % t] = np.array([dpb["ref_feature_el"].double().sum().item(),
                                                      dpb["ref_feature_el"].double().abs().sum().item()])
            dpb["ref_frame_bl"].clamp_(0, 1)                                    # test.py:249-250
            dpb["ref_frame_el"].clamp_(0, 1)
            out["f%d_psnr" % t] = np.array([psnr(x_bl, dpb["ref_frame_bl"]), psnr(x_el, dpb["ref_frame_el"])])
            print(name, "frame", t, "bits", out["f%d_bits" % t], "psnr", out["f%d_psnr" % t])
    out["x_bl"] = torch.cat(x_bls, 0).numpy()
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    torch.set_num_threads(8)
    IntraSS, LSSVC_extend = import_reference()
    from src.utils.core import imresize  # reference's MATLAB-style bicubic (core.py:364-432)
    for case in (sys.argv[1:] or list(CASES)):
        run_case(case, IntraSS, LSSVC_extend, imresize)
