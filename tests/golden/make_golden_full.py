"""Full-size golden fixtures: the REFERENCE (imported from /root/reference, build container only) run at the
benchmark's own sizes on seeded synthetic weights and an integer-exact synthetic clip:

    python tests/golden/make_golden_full.py [case ...]

    x2_1080p_ipp    BASELINE configs[1] shape: EL 1152x1920 / BL 576x960, I + first P + steady P
    x1_5_1080p_ip   EL 1152x1920 / BL 768x1280 (the non-integer ratio at full size), I + first P
    x2_2160p_ipp    BASELINE configs[3] shape: EL 2176x3840 / BL 1088x1920, I + first P + steady P (a P-frame: ~40 GB of host memory)
    x2_1080p_gop32  BASELINE configs[1] in full: the whole 32-frame closed loop of test.py:182-250 at EL 1152x1920 / BL 576x960.
    x2_2160p_gop12  BASELINE configs[3] in full (round 5): the 12-frame closed loop (IP12) at EL 2176x3840 / BL 1088x1920.
                    Bits, PSNR, whole-tensor sums and the quantised latents of ALL 32 frames; strided samples only of
                    frames 0, 1, 2, 15, 31 (the fixture would otherwise be 50 MB)

The frame loop is test.py's (test.py:182-250), exactly as tests/golden/make_golden.py replays it. What is stored per
frame: bits, PSNR, strided samples and double-precision sums of every tensor the model hands back, and -- so that a
rounding tie flipped by a differently ordered fp32 sum can be told apart from an error -- the reference's QUANTISED
LATENTS (int16, NCHW order), recorded by wrapping torch.round (I-frames) and the static get_*_bits_probs helpers
(P-frames; LSSVC_net.py:153-167) while the reference runs. Inputs are not stored: the clip is
lssvc_amd.synth.synth_clip_exact (integer arithmetic, sha1 kept here) and the base-layer frames are recomputed by the
pinned bicubic restatement (sha1 + samples kept here; the generator asserts it equals the reference's imresize).
"""
import hashlib
import os
import resource
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

from ref_import import import_reference  # noqa: E402
from lssvc_amd.preprocess import interlayer_padding, imresize_bicubic  # noqa: E402
from lssvc_amd.synth import synth_state_dict, synth_clip_exact  # noqa: E402

CASES = {
    # name: (frames, picture H, picture W, scale, gain, seed)
    "x2_1080p_ipp": (3, 1080, 1920, 2.0, 0.55, 0),
    "x1_5_1080p_ip": (2, 1080, 1920, 1.5, 0.55, 1),
    "x2_2160p_ipp": (3, 2160, 3840, 2.0, 0.55, 2),
    "x2_1080p_gop32": (32, 1080, 1920, 2.0, 0.55, 4),
    "x2_2160p_gop12": (12, 2160, 3840, 2.0, 0.55, 6),     # round 5: BASELINE configs[3] in full (UVG 2160p enhancement layer, IP12): the whole 12-frame closed loop
    "_dev_x2_128_ipp": (3, 120, 128, 2.0, 0.55, 3),       # generator self-check only, not committed
}
# (spatial stride, channel stride) of the stored samples
DENSE_FRAMES = (0, 1, 2, 11, 15, 31)     # frames of a long case whose strided samples are stored (11: the last frame of the 12-frame 2160p case, round 5)
SAMPLE = {"x_hat_bl": (4, 1), "x_hat_el": (8, 1), "feature_el": (32, 4), "feature_bl": (16, 4), "mv_hat": (8, 1),
          "warp_frame": (8, 1), "x_bl": (8, 1)}


def psnr(a, b):
    return (10 * torch.log10(1.0 / torch.mean((a - b) ** 2))).item()


def sample(name, t):
    s, c = SAMPLE[name]
    return t[:, ::c, ::s, ::s].contiguous().numpy()


def sums(t):
    return np.array([t.double().sum().item(), t.double().abs().sum().item()])


def sha1(t):
    return hashlib.sha1(t.contiguous().numpy().tobytes()).hexdigest()


class Recorder:
    """Collects the reference's quantised latents while it runs, without touching its code."""

    def __init__(self, classes):
        self.rounds = []
        self.bits_args = []
        self._round = torch.round
        self._classes = classes
        self._saved = []

    def __enter__(self):
        rec = self

        def round_hook(x, *a, **k):
            y = rec._round(x, *a, **k)
            rec.rounds.append(y.detach().to(torch.int16).clone())   # a copy: the caller goes on to modify y in place
            return y
        torch.round = round_hook
        for cls in self._classes:
            for name in ("get_y_bits_probs", "get_z_bits_probs"):
                orig = getattr(cls, name)
                self._saved.append((cls, name, cls.__dict__[name]))

                def wrapped(sym, other, _orig=orig, _name=name):
                    rec.bits_args.append((_name, sym.detach().to(torch.int16).clone()))
                    return _orig(sym, other)
                setattr(cls, name, staticmethod(wrapped))
        return self

    def __exit__(self, *exc):
        torch.round = self._round
        for cls, name, orig in self._saved:
            setattr(cls, name, orig)


def second_run_record(out, name):
    """Round 6 (LSSVC_GOLDEN_SECOND=1, LSSVC_GOLDEN_THREADS=n): the SAME case run by the reference a second time on another
    thread count, stored as <name>_ref_t2.npz relative to the committed fixture A: per frame the bits, PSNR and sums of this run, and
    for every quantised-latent plane the positions where it differs from A's plane with this run's values there (plus the sha1 of the
    whole plane). torch's CPU convolutions split their fp32 sums by thread, a value on a rounding tie (LSSVC_net.py:193,
    img_entropy_models.py:237) falls either way, and the closed loop carries the difference on: this file is the reference's
    disagreement WITH ITSELF, the yardstick tests/test_gpu_golden_full.py derives its tie allowance from."""
    a = np.load(os.path.join(HERE, name + ".npz"))
    assert str(a["clip_sha1"]) == str(out["clip_sha1"]) and (a["meta"] == out["meta"]).all()
    rec = {"meta": out["meta"], "scale_gain": out["scale_gain"], "clip_sha1": out["clip_sha1"],
           "reference_threads": out["reference_threads"], "reference_threads_a": a["reference_threads"],
           "reference_seconds": out["reference_seconds"]}
    for k, v in out.items():
        if k.endswith(("_bits", "_psnr", "_sum")):
            rec[k] = v
        elif "_sym_" in k:
            d = np.flatnonzero(v != a[k]).astype(np.int32)
            rec[k.replace("_sym_", "_symdiff_") + "_idx"] = d
            rec[k.replace("_sym_", "_symdiff_") + "_val"] = v[d]
            rec[k.replace("_sym_", "_symsha1_")] = np.array(hashlib.sha1(np.ascontiguousarray(v).tobytes()).hexdigest())
    return rec, name + "_ref_t2"


def run_case(name, IntraSS, LSSVC_extend, imresize):
    from src.models.LSSVC_net import LSSVC
    from src.models.dmc_net import DMC
    frames, ph, pw, scale, gain, seed = CASES[name]
    pad = interlayer_padding(ph, pw, scale)
    (H, W), (h, w) = pad["HR_padded_size"], pad["LR_padded_size"]
    sd_i = synth_state_dict("intra_ss", seed, gain)
    sd_p = synth_state_dict("lssvc_extend", seed, gain)
    inet = IntraSS.from_state_dict(dict(sd_i)).eval()
    pnet = LSSVC_extend()
    pnet.load_dict(sd_p)
    pnet.eval()
    clip = synth_clip_exact(frames, ph, pw, seed=seed)
    out = {"meta": np.array([frames, ph, pw, H, W, h, w, seed], dtype=np.int64),
           "scale_gain": np.array([scale, gain], dtype=np.float64), "clip_sha1": np.array(sha1(clip))}
    dpb = None
    secs = []
    with torch.no_grad():
        for t in range(frames):
            x_el = torch.nn.functional.pad(clip[t:t + 1].float() / 255.0, pad["P_HR"], mode="constant", value=0)   # test.py:192-197
            x_bl = imresize(x_el, sizes=(h, w), kernel="cubic").clamp_(0, 1)                                       # test.py:199
            mine = imresize_bicubic(x_el, (h, w)).clamp_(0, 1)
            assert torch.equal(mine, x_bl), "bicubic restatement differs from the reference's imresize: max %g" % (mine - x_bl).abs().max().item()
            dense = frames <= 3 or t in DENSE_FRAMES
            if dense:
                out["f%d_x_bl" % t] = sample("x_bl", x_bl)
            out["f%d_x_bl_sha1" % t] = np.array(sha1(x_bl))
            inet.set_scale_information(scale, (H, W), (0, 0, 0, 0))
            pnet.set_scale_information(scale, (H, W), (0, 0, 0, 0))
            t0 = time.time()
            with Recorder((LSSVC, DMC)) as rec:
                if t == 0:
                    r = inet.encode_decode(x_bl, x_el, None, None, h, w, H, W)
                    dpb = {"ref_frame_bl": r["x_hat_bl"], "ref_frame_el": r["x_hat_el"], "ref_feature_bl": None,
                           "ref_feature_el": r["feature_el"]}
                else:
                    r = pnet.encode_decode(x_bl, x_el, dpb, None, None, W, H, w, h)
                    dpb = r["dpb"]
            secs.append(time.time() - t0)
            if t == 0:
                # BL z, BL y, EL z, EL y by element count (each is rounded more than once with equal results)
                n_bl = sd_i["base_layer_model.g_s.0.conv1.weight"].shape[0]
                want = {"bl_z": n_bl * (h // 64) * (w // 64), "bl_y": n_bl * (h // 16) * (w // 16),
                        "el_z": 64 * (H // 64) * (W // 64), "el_y": 96 * (H // 16) * (W // 16)}
                for key, n in want.items():
                    hits = [x for x in rec.rounds if x.numel() == n]
                    assert hits, (key, n, [tuple(x.shape) for x in rec.rounds])
                    for x in hits[1:]:
                        assert torch.equal(x.reshape(-1), hits[0].reshape(-1)), key
                    out["f0_sym_" + key] = hits[0].reshape(-1).numpy()   # (1,C,H,W) or the bottleneck's (C,1,H*W): channel-major
            else:
                # call order: BL y, mv_y, z, mv_z (dmc_net.py:466-469) then EL y, mv_y, z, mv_z (LSSVC_net.py:504-507)
                names = [n for n, _ in rec.bits_args]
                assert names == ["get_y_bits_probs", "get_y_bits_probs", "get_z_bits_probs", "get_z_bits_probs"] * 2, names
                for key, (_, v) in zip(("bl_y", "bl_mv_y", "bl_z", "bl_mv_z", "el_y", "el_mv_y", "el_z", "el_mv_z"), rec.bits_args):
                    out["f%d_sym_%s" % (t, key)] = v.reshape(-1).numpy()
                if dense:
                    out["f%d_mv_hat" % t] = sample("mv_hat", r["mv_hat"])
                    out["f%d_warp_frame" % t] = sample("warp_frame", r["warp_frame"])
                    out["f%d_feature_bl" % t] = sample("feature_bl", dpb["ref_feature_bl"])
                out["f%d_mv_hat_sum" % t] = sums(r["mv_hat"])
                out["f%d_warp_frame_sum" % t] = sums(r["warp_frame"])
                out["f%d_feature_bl_sum" % t] = sums(dpb["ref_feature_bl"])
            out["f%d_bits" % t] = np.array([r["bit_bl"], r["bit_el"]], dtype=np.float64)
            if dense:
                out["f%d_x_hat_bl" % t] = sample("x_hat_bl", dpb["ref_frame_bl"])      # un-clamped, as returned
                out["f%d_x_hat_el" % t] = sample("x_hat_el", dpb["ref_frame_el"])
                out["f%d_feature_el" % t] = sample("feature_el", dpb["ref_feature_el"])
            out["f%d_x_hat_bl_sum" % t] = sums(dpb["ref_frame_bl"])
            out["f%d_x_hat_el_sum" % t] = sums(dpb["ref_frame_el"])
            out["f%d_feature_el_sum" % t] = sums(dpb["ref_feature_el"])
            dpb["ref_frame_bl"].clamp_(0, 1)                                            # test.py:249-250
            dpb["ref_frame_el"].clamp_(0, 1)
            out["f%d_psnr" % t] = np.array([psnr(x_bl, dpb["ref_frame_bl"]), psnr(x_el, dpb["ref_frame_el"])])
            print(name, "frame", t, "bits", out["f%d_bits" % t], "psnr", out["f%d_psnr" % t], "%.1f s" % secs[-1],
                  "peak RSS %.1f GB" % (resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6), flush=True)
    out["reference_seconds"] = np.array(secs)
    out["reference_threads"] = np.array(torch.get_num_threads())
    if os.environ.get("LSSVC_GOLDEN_SECOND"):
        out, name = second_run_record(out, name)
    path = os.path.join(os.environ.get("LSSVC_GOLDEN_OUT", HERE), name + ".npz")      # (override: dry runs that must not touch the committed fixtures)
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    torch.set_num_threads(int(os.environ.get("LSSVC_GOLDEN_THREADS", "8")))
    IntraSS, LSSVC_extend = import_reference()
    from src.utils.core import imresize  # reference's MATLAB-style bicubic (core.py:364-432)
    for case in (sys.argv[1:] or list(CASES)):
        run_case(case, IntraSS, LSSVC_extend, imresize)
