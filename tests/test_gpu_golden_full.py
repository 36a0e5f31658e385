"""Parity at the benchmark's own sizes, against outputs of the REFERENCE itself (tests/golden/make_golden_full.py):

    x2_1080p_ipp    BASELINE configs[1]: EL 1152x1920 / BL 576x960, I + first P + steady P
    x1_5_1080p_ip   the non-integer ratio at full size: EL 1152x1920 / BL 768x1280, I + first P
    x2_2160p_ipp    BASELINE configs[3]'s shape: EL 2176x3840 / BL 1088x1920, I + first P + steady P
    x2_1080p_gop32  BASELINE configs[1] in full: all 32 frames of the closed loop (test.py:182-250) at 1152x1920 / 576x960; bits,
                    PSNR, whole-tensor sums and symbols of every frame, strided samples of frames 0, 1, 2, 15, 31 (round 5: in the
                    default run for the default precision; the exact-fp32 mode behind --runslow). Result: every frame inside the
                    bars with the plain rule on 31 of 32 frames (f16x3) and by the tie allowance on one -- frame 19, 2.08e-5 bpp,
                    ONE tie event of 9 symbols (profiles/r05_golden_gop32_gpu.txt).
                    test_free_running_gop32_against_reference codes the same GOP WITHOUT re-aligning the loop after a tie.
    x2_2160p_gop12  BASELINE configs[3] in full (round 5): the 12-frame closed loop (IP12) at EL 2176x3840 / BL 1088x1920, 75 minutes of the
                    reference on 7 threads (36.9 GB peak); bits, PSNR, sums and symbols of every frame, strided samples of frames 0, 1, 2, 11

Bars (BASELINE.json north_star): |d bpp| <= 1e-5 and |d PSNR| <= 1e-4 dB per layer per frame, in both conv precisions.
The fixtures also hold the reference's QUANTISED LATENTS, which makes the comparison exact where a plain replay cannot
be: 2.3 M values are rounded per P-frame and a differently ordered fp32 sum moves one of them across a rounding tie about
once per frame at this size (noise floor 2.6e-7 at the quantiser inputs; the first run of this test
saw exactly that). One flipped symbol changes the reconstruction by ~2e-3 in a 100x100-pixel neighbourhood and from then
on the closed loop drifts locally, although bits and PSNR stay inside the bars. So every frame is checked twice:

  ENCODER pass (the public estimate-mode API, from a DPB that is aligned with the reference's): PSNR at the bar; every symbol
      against the reference's -- differences must be off by exactly one and rare as EVENTS (spatial clusters per latent plane
      of 0.2-1.6 M symbols: a tie in the 4-step spatial prior can take near-tie neighbours with it), i.e. ties, not errors; bits
      at 1e-5 bpp, plus an allowance per symbol that fell the other way (the I-frame of x2_2160p_ipp has two such BL symbols in
      the f32 mode: 38 bits = 1.8e-5 bpp of its 2.09 M pixels). How many events, symbols and bits: TWICE what the reference does
      to itself between two runs on other thread counts (helpers.tie_allowance; the comment below the imports).
  DECODER pass (the decoder role of the same codec functions, fed the REFERENCE's symbols): every tensor the model hands
      back (reconstructions, features, mv_hat, warp_frame) against the reference's samples and whole-tensor sums, with no
      rounding in the way. Its outputs are the DPB of the next frame, which keeps the loop aligned with the reference.
"""
import numpy as np
import pytest
import torch

from helpers import load_full_case, replay_full, full_sample, decode_from_symbols, tie_clusters, tie_allowance, reference_self_disagreement

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
# THE TIE ALLOWANCE IS DERIVED, NOT CHOSEN (round 6): tests/golden/x2_1080p_gop32_ref_t2.npz is the REFERENCE run a
# second time on this GOP with 8 threads instead of 6, and helpers.reference_self_disagreement() reads from it what the reference does
# to itself: symbols off by exactly one, at most 11 per latent plane in ONE spatial cluster, at most 16.7 bits of a layer's count per
# differing symbol, worst frame 4.30e-5 bpp. helpers.tie_allowance() is TWICE that, per plane, scaled with the plane's size:
#   max_flips   differing symbols per plane, clusters' followers included      (2 x 11 = 22 at 1080p; rounds 4-5 had CHOSEN 48)
#   max_events  independent tie events per plane (helpers.tie_clusters)         (2 x 1 = 2; chosen: 4)
#   flip_bits   what one such symbol may move a layer's bit count beyond 1e-5 bpp (2 x 16.7 = 33.4; chosen: 40)
# With no symbol off the reference's the plain bar holds: 1e-5 bpp, 1e-4 dB.


@pytest.fixture(params=["f16x3", "f32"])
def precision(request):
    from lssvc_amd import hip_ops
    old = hip_ops.CONV_PRECISION
    hip_ops.set_conv_precision(request.param)
    yield request.param
    hip_ops.set_conv_precision(old)


def _runslow(request):
    import os
    return bool(request.config.getoption("--runslow")) or os.environ.get("LSSVC_SLOW") == "1"


@pytest.mark.parametrize("case", ["x2_1080p_ipp", "x1_5_1080p_ip", "x2_2160p_ipp", "x2_1080p_gop32", "x2_2160p_gop12"])
def test_full_size_frames_match_reference(case, precision, request):
    from lssvc_amd import IntraSS, LSSVC_extend
    import os
    if case != "x2_1080p_ipp" and precision == "f32" and not _runslow(request):
        pytest.skip("the default run holds the exact-fp32 mode to the reference at full size on x2_1080p_ipp (and on the five small goldens); "
                    "its other full-size twins are behind --runslow -- the default precision (f16x3) runs every case")
    if case == "x2_2160p_gop12" and not _runslow(request):
        pytest.skip("configs[3]'s whole 12-frame 2160p GOP: a minute of GPU box time per precision, behind --runslow (profiles/r05_golden_2160p_gop12_gpu.txt holds its run)")
    if not os.path.exists(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", case + ".npz")):
        pytest.skip("fixture %s.npz not generated (tests/golden/make_golden_full.py %s)" % (case, case))
    from lssvc_amd.preprocess import psnr
    from lssvc_amd.synth import synth_state_dict
    from helpers import full_case_inputs
    z, m = load_full_case(case)
    inet = IntraSS.from_state_dict(synth_state_dict("intra_ss", m["seed"], m["gain"])).to(DEV).eval()
    inet.update(force=True)                     # the decoder role reads the bottleneck medians from the tables
    pnet = None
    if m["frames"] > 1:
        pnet = LSSVC_extend()
        pnet.load_dict(synth_state_dict("lssvc_extend", m["seed"], m["gain"]))
        pnet.to(DEV).eval()
        pnet.update(force=True)
    inputs, exact = full_case_inputs(case)
    hr = (m["H"], m["W"])
    dpb, report, n_sym = None, [], 0
    for t, (x_bl, x_el) in enumerate(inputs):
        x_bl, x_el = x_bl.to(DEV), x_el.to(DEV)
        net = inet if t == 0 else pnet
        net.set_scale_information(m["scale"], hr, (0, 0, 0, 0))
        # ---------------- encoder pass: public API, estimate mode
        taps = net.taps = {}
        if t == 0:
            r = inet.encode_decode(x_bl, x_el, None, None, m["h"], m["w"], m["H"], m["W"])
            enc = {"ref_frame_bl": r["x_hat_bl"], "ref_frame_el": r["x_hat_el"]}
        else:
            r = pnet.encode_decode(x_bl, x_el, dpb, None, None, m["W"], m["H"], m["w"], m["h"])
            enc = r["dpb"]
        net.taps = None
        flips = {}
        for key in [k[len("f%d_sym_" % t):] for k in z.files if k.startswith("f%d_sym_" % t)]:
            want = z["f%d_sym_%s" % (t, key)]
            got = taps[key].reshape(-1).numpy()
            assert got.shape == want.shape, (key, got.shape, want.shape)
            d = got.astype(np.int32) - want.astype(np.int32)
            flips[key] = tie_clusters(d, key, m["H"], m["W"], m["h"], m["w"])
            n_sym += d.size
        bits = z["f%d_bits" % t]
        d_bpp = (abs(r["bit_bl"] - bits[0]) / (m["h"] * m["w"]), abs(r["bit_el"] - bits[1]) / (m["H"] * m["W"]))
        want_psnr = z["f%d_psnr" % t]
        p_enc = (psnr(x_bl, enc["ref_frame_bl"].clamp(0, 1)), psnr(x_el, enc["ref_frame_el"].clamp(0, 1)))
        nflip = {k: "%d in %d event(s)" % (v[0], v[2]) for k, v in flips.items() if v[0]}
        print("%s %s frame %d: d bpp (%.2e, %.2e), encoder d PSNR (%.1e, %.1e), flipped symbols %s, inputs bit-equal %s" % (
            case, precision, t, d_bpp[0], d_bpp[1], p_enc[0] - want_psnr[0], p_enc[1] - want_psnr[1], nflip or "none", exact), flush=True)
        flip_bits = 0.0
        for key, (n, mx, events) in flips.items():
            al = tie_allowance(z["f%d_sym_%s" % (t, key)].size, key)
            flip_bits = al["flip_bits"]
            assert n <= al["max_flips"] and mx <= al["max_abs_diff"] and events <= al["max_events"], (t, key, n, mx, events, al["max_flips"], al["max_events"])
        n_bl = sum(v[0] for k, v in flips.items() if k.startswith("bl"))
        n_el = sum(v[0] for k, v in flips.items() if k.startswith("el"))
        assert abs(r["bit_bl"] - bits[0]) <= 1e-5 * m["h"] * m["w"] + flip_bits * n_bl, (t, r["bit_bl"], bits[0], n_bl)
        assert abs(r["bit_el"] - bits[1]) <= 1e-5 * m["H"] * m["W"] + flip_bits * n_el, (t, r["bit_el"], bits[1], n_el)
        assert abs(p_enc[0] - want_psnr[0]) <= 1e-4 and abs(p_enc[1] - want_psnr[1]) <= 1e-4, (t, p_enc, want_psnr)
        del r, enc
        # ---------------- decoder pass on the reference's symbols: every tensor, tight, and the next frame's DPB
        d = decode_from_symbols({k[len("f%d_sym_" % t):]: z[k] for k in z.files if k.startswith("f%d_sym_" % t)}, m["H"], m["W"], m["h"], m["w"],
                                t, inet, pnet, dpb)
        if t == 0:
            dpb = {"ref_frame_bl": d["x_hat_bl"], "ref_frame_el": d["x_hat_el"], "ref_feature_bl": None, "ref_feature_el": d["feature_el"]}
        else:
            dpb = d["dpb"]
        dense = ("f%d_x_hat_el" % t) in z.files                    # the 32-frame case stores strided samples of five frames only
        for k, name in (("ref_frame_bl", "x_hat_bl"), ("ref_frame_el", "x_hat_el")):
            x = dpb[k].cpu()                                        # un-clamped, as returned
            if dense:
                np.testing.assert_allclose(full_sample(name, x).numpy(), z["f%d_%s" % (t, name)], atol=2e-4, rtol=0)
            assert x.double().abs().sum().item() == pytest.approx(z["f%d_%s_sum" % (t, name)][1], rel=1e-5)
        fe = dpb["ref_feature_el"].cpu()
        if dense:
            np.testing.assert_allclose(full_sample("feature_el", fe).numpy(), z["f%d_feature_el" % t], atol=5e-4, rtol=1e-4)
        assert fe.double().abs().sum().item() == pytest.approx(z["f%d_feature_el_sum" % t][1], rel=1e-5)
        if t > 0:
            if dense:
                np.testing.assert_allclose(full_sample("mv_hat", d["mv_hat"].cpu()).numpy(), z["f%d_mv_hat" % t], atol=2e-4, rtol=0)
                np.testing.assert_allclose(full_sample("warp_frame", d["warp_frame"].cpu()).numpy(), z["f%d_warp_frame" % t], atol=2e-4, rtol=0)
            if ("f%d_warp_frame_sum" % t) in z.files:
                assert d["warp_frame"].double().abs().sum().item() == pytest.approx(z["f%d_warp_frame_sum" % t][1], rel=1e-5)
            assert d["mv_hat"].double().abs().sum().item() == pytest.approx(z["f%d_mv_hat_sum" % t][1], rel=1e-5, abs=1e-3)
            fb = dpb["ref_feature_bl"].cpu()
            if dense:
                np.testing.assert_allclose(full_sample("feature_bl", fb).numpy(), z["f%d_feature_bl" % t], atol=5e-4, rtol=1e-4)
            assert fb.double().abs().sum().item() == pytest.approx(z["f%d_feature_bl_sum" % t][1], rel=1e-5)
        dpb["ref_frame_bl"].clamp_(0, 1)                            # test.py:249-250
        dpb["ref_frame_el"].clamp_(0, 1)
        p_dec = (psnr(x_bl, dpb["ref_frame_bl"]), psnr(x_el, dpb["ref_frame_el"]))
        assert abs(p_dec[0] - want_psnr[0]) <= 1e-4 and abs(p_dec[1] - want_psnr[1]) <= 1e-4, (t, p_dec, want_psnr)
        report.append((t, sum(v[0] for v in flips.values())))
    print("%s %s: (frame, flipped symbols) = %s; %d symbols compared, %d flipped" % (
        case, precision, [r for r in report if r[1]] or "none", n_sym, sum(r[1] for r in report)))


def test_free_running_gop32_against_reference(request):
    """What a rounding tie costs a caller who just codes the GOP: BASELINE configs[1]'s 32 frames through the public API
    alone, every frame's DPB the previous frame's own output -- no decoder pass on the reference's symbols, no re-alignment -- in
    the default precision, against the reference's per-frame bits and PSNR (tests/golden/x2_1080p_gop32.npz). Up to the first frame
    in which a symbol falls on the other side of a tie the plain bars hold (1e-5 bpp, 1e-4 dB). From there on the loop is a
    slightly different, equally valid closed loop: the flipped symbol changes the reconstruction in its neighbourhood and the
    difference feeds forward through the DPB, so later frames are RECORDED (gpurun_out/free_running_gop32.json, printed) and held to
    TWICE WHAT THE REFERENCE DOES TO ITSELF on this GOP (round 6: tests/golden/x2_1080p_gop32_ref_t2.npz, the reference on 8 threads
    against the reference on 6, free-running like this test; helpers.reference_self_disagreement): per frame 2 x 4.30e-5 bpp and 2 x 11
    differing symbols, over the GOP 2 x 46 differing symbols in 2 x 10 frames and 2 x the reference's GOP-average |d bpp|; PSNR at the
    plain 1e-4 dB bar on every frame (the reference moves its own by 9.5e-7 dB). Rounds 4-5 held these frames to bars this test had
    chosen for itself (1e-3 bpp, 1e-2 dB)."""
    import json
    import os
    from lssvc_amd import IntraSS, LSSVC_extend, hip_ops
    from lssvc_amd.preprocess import psnr
    from lssvc_amd.synth import synth_state_dict
    from helpers import full_case_inputs
    case = "x2_1080p_gop32"
    old = hip_ops.CONV_PRECISION
    hip_ops.set_conv_precision("f16x3")
    try:
        z, m = load_full_case(case)
        inet = IntraSS.from_state_dict(synth_state_dict("intra_ss", m["seed"], m["gain"])).to(DEV).eval()
        pnet = LSSVC_extend()
        pnet.load_dict(synth_state_dict("lssvc_extend", m["seed"], m["gain"]))
        pnet.to(DEV).eval()
        inputs, _ = full_case_inputs(case)
        hr = (m["H"], m["W"])
        dpb, rows, first_flip = None, [], None
        yard = reference_self_disagreement(case)
        for t, (x_bl, x_el) in enumerate(inputs):
            x_bl, x_el = x_bl.to(DEV), x_el.to(DEV)
            net = inet if t == 0 else pnet
            net.set_scale_information(m["scale"], hr, (0, 0, 0, 0))
            taps = net.taps = {}
            if t == 0:
                r = inet.encode_decode(x_bl, x_el, None, None, m["h"], m["w"], m["H"], m["W"])
                dpb = {"ref_frame_bl": r["x_hat_bl"], "ref_frame_el": r["x_hat_el"], "ref_feature_bl": None, "ref_feature_el": r["feature_el"]}
            else:
                r = pnet.encode_decode(x_bl, x_el, dpb, None, None, m["W"], m["H"], m["w"], m["h"])
                dpb = r["dpb"]
            net.taps = None
            dpb["ref_frame_bl"].clamp_(0, 1)                            # test.py:249-250
            dpb["ref_frame_el"].clamp_(0, 1)
            differing = 0
            for key in [k[len("f%d_sym_" % t):] for k in z.files if k.startswith("f%d_sym_" % t)]:
                differing += int(np.count_nonzero(taps[key].reshape(-1).numpy().astype(np.int32) - z["f%d_sym_%s" % (t, key)].astype(np.int32)))
            if differing and first_flip is None:
                first_flip = t
            bits, want_psnr = z["f%d_bits" % t], z["f%d_psnr" % t]
            row = {"frame": t, "d_bpp_bl": (r["bit_bl"] - float(bits[0])) / (m["h"] * m["w"]), "d_bpp_el": (r["bit_el"] - float(bits[1])) / (m["H"] * m["W"]),
                   "d_psnr_bl": psnr(x_bl, dpb["ref_frame_bl"]) - float(want_psnr[0]), "d_psnr_el": psnr(x_el, dpb["ref_frame_el"]) - float(want_psnr[1]),
                   "symbols_differing_from_reference": differing}
            rows.append(row)
            print("free-running frame %2d: d bpp (%+.2e, %+.2e)  d PSNR (%+.1e, %+.1e) dB  symbols differing %d" % (
                t, row["d_bpp_bl"], row["d_bpp_el"], row["d_psnr_bl"], row["d_psnr_el"], differing), flush=True)
            aligned = first_flip is None
            bpp_bar = 1e-5 if aligned else 2 * yard["max_d_bpp"]
            assert abs(row["d_bpp_bl"]) <= bpp_bar and abs(row["d_bpp_el"]) <= bpp_bar, (row, first_flip, bpp_bar)
            assert abs(row["d_psnr_bl"]) <= 1e-4 and abs(row["d_psnr_el"]) <= 1e-4, (row, first_flip)
            assert differing <= 2 * yard["max_symbols_per_plane"], (row, yard["max_symbols_per_plane"])
        n = len(rows)
        avg = {k: sum(r_[k] for r_ in rows) / n for k in ("d_bpp_bl", "d_bpp_el", "d_psnr_bl", "d_psnr_el")}
        worst = {k: max(abs(r_[k]) for r_ in rows) for k in avg}
        print("free-running GOP: first frame with a symbol off the reference's: %s; GOP average %s; worst frame %s" % (first_flip, avg, worst))
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        with open(os.path.join(root, "gpurun_out", "free_running_gop32.json"), "w") as f:
            json.dump({"case": case, "precision": "f16x3", "first_frame_with_a_differing_symbol": first_flip, "gop_average": avg, "worst_frame_abs": worst, "frames": rows,
                       "reference_against_itself": {k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in yard.items() if k != "plane_symbols"}}, f, indent=1)
        n_frames = sum(1 for r_ in rows if r_["symbols_differing_from_reference"])
        n_syms = sum(r_["symbols_differing_from_reference"] for r_ in rows)
        print("against the reference's disagreement with itself (8 vs 6 threads): frames with a differing symbol %d (reference %d), differing symbols %d (%d), "
              "worst frame %.2e bpp (%.2e), GOP average %.2e bpp (%.2e)" % (n_frames, yard["frames_with_differing_symbols"], n_syms, yard["symbols"],
                                                                          max(worst["d_bpp_bl"], worst["d_bpp_el"]), yard["max_d_bpp"], max(abs(avg["d_bpp_bl"]), abs(avg["d_bpp_el"])), yard["gop_avg_d_bpp"]))
        assert n_frames <= 2 * yard["frames_with_differing_symbols"] and n_syms <= 2 * yard["symbols"], (n_frames, n_syms, yard)
        assert abs(avg["d_bpp_bl"]) <= 2 * yard["gop_avg_d_bpp"] and abs(avg["d_bpp_el"]) <= 2 * yard["gop_avg_d_bpp"], (avg, yard["gop_avg_d_bpp"])
        assert abs(avg["d_psnr_bl"]) <= 1e-4 and abs(avg["d_psnr_el"]) <= 1e-4, avg
    finally:
        hip_ops.set_conv_precision(old)
