"""Frame-level parity on the GPU: replay every golden case (test.py's frame loop) through the HIP
models via the reference's own API and compare with the reference's stored outputs.
Bars (BASELINE.json north_star): |dPSNR| <= 1e-4 dB, |d bpp| <= 1e-5 per frame."""
import numpy as np
import pytest
import torch

from helpers import CASES, load_case, replay

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _nets(m):
    from lssvc_amd import IntraSS, LSSVC_extend
    from lssvc_amd.synth import synth_state_dict
    inet = IntraSS.from_state_dict(synth_state_dict("intra_ss", m["seed"], m["gain"])).to(DEV).eval()
    pnet = LSSVC_extend()
    pnet.load_dict(synth_state_dict("lssvc_extend", m["seed"], m["gain"]))
    pnet.to(DEV).eval()
    return inet, pnet


HEAVY_IN_BOTH_MODES_ONLY_WITH_RUNSLOW = ("test_gop_drift_symbol_aware", "test_gop_drift_vs_oracle", "test_frames_384x640_vs_oracle", "test_dataset_picture_sizes_vs_oracle")


@pytest.fixture(params=["f16x3", "f32"])
def precision(request):
    """Every frame-level bar is held in both conv arithmetic modes. Round 5: the default run keeps the exact-fp32 mode for the golden cases
    against the reference and for configs[0]; the closed loops against the oracle's fixtures run in the DEFAULT mode (f16x3), their exact-fp32
    twins behind --runslow -- the suite's host seconds differ 2x from box to box (7:20 on one, 13:03 on another, same code), and the
    driver's step limit is what a slow box must stay inside."""
    import os
    from lssvc_amd import hip_ops
    if request.param == "f32" and request.node.originalname in HEAVY_IN_BOTH_MODES_ONLY_WITH_RUNSLOW \
            and not (request.config.getoption("--runslow") or os.environ.get("LSSVC_SLOW") == "1"):
        pytest.skip("exact-fp32 twin of a closed loop against the oracle: --runslow")
    old = hip_ops.CONV_PRECISION
    hip_ops.set_conv_precision(request.param)
    yield request.param
    hip_ops.set_conv_precision(old)


@pytest.mark.parametrize("case", CASES)
def test_frames_match_reference(case, precision):
    z, m = load_case(case)
    inet, pnet = _nets(m)

    def i_fn(xb, xe, hr):
        inet.set_scale_information(m["scale"], hr, m["pad"])
        return inet.encode_decode(xb, xe, None, None, m["h"], m["w"], m["H"], m["W"])

    def p_fn(xb, xe, dpb, hr, s):
        pnet.set_scale_information(s, hr, m["pad"])
        return pnet.encode_decode(xb, xe, dpb, None, None, m["W"], m["H"], m["w"], m["h"])

    for t, r, raw, dpb, p_bl, p_el in replay(case, i_fn, p_fn, device=DEV):
        bits = z["f%d_bits" % t]
        d_bpp_bl = abs(r["bit_bl"] - bits[0]) / (m["h"] * m["w"])
        d_bpp_el = abs(r["bit_el"] - bits[1]) / (m["H"] * m["W"])
        assert d_bpp_bl <= 1e-5 and d_bpp_el <= 1e-5, (t, r["bit_bl"], bits[0], r["bit_el"], bits[1])
        want_psnr = z["f%d_psnr" % t]
        assert abs(p_bl - want_psnr[0]) <= 1e-4 and abs(p_el - want_psnr[1]) <= 1e-4, (t, p_bl, p_el, want_psnr)
        np.testing.assert_allclose(raw["x_hat_bl"].cpu().numpy(), z["f%d_x_hat_bl" % t], atol=2e-4, rtol=0)
        want = z["f%d_x_hat_el" % t]
        got = raw["x_hat_el"].cpu().numpy()
        if want.shape != got.shape:
            got = got[:, :, ::2, ::2]
        np.testing.assert_allclose(got, want, atol=2e-4, rtol=0)
        fe = dpb["ref_feature_el"].cpu()
        np.testing.assert_allclose(fe[:, :, ::8, ::8].numpy(), z["f%d_feature_el" % t], atol=5e-4, rtol=1e-4)
        assert fe.double().abs().sum().item() == pytest.approx(z["f%d_feature_el_sum" % t][1], rel=1e-5)
        if t > 0:
            np.testing.assert_allclose(r["mv_hat"].cpu()[:, :, ::2, ::2].numpy(), z["f%d_mv_hat" % t], atol=2e-4, rtol=0)
            np.testing.assert_allclose(r["warp_frame"].cpu()[:, :, ::2, ::2].numpy(), z["f%d_warp_frame" % t],
                                       atol=2e-4, rtol=0)
            np.testing.assert_allclose(dpb["ref_feature_bl"].cpu()[:, :, ::4, ::4].numpy(), z["f%d_feature_bl" % t],
                                       atol=5e-4, rtol=1e-4)


def test_rejects_cpu_device():
    from lssvc_amd import IntraSS
    from lssvc_amd.synth import synth_state_dict
    net = IntraSS.from_state_dict(synth_state_dict("intra_ss", 0, 0.6))
    with pytest.raises(RuntimeError):
        net.to("cpu")


_ORACLE_GOPS = {}


def _oracle_gop(n, H, W, seed, gain, scale=2.0, bl=None):
    """The CPU oracle's closed-loop coding of a synthetic clip (1 I + n-1 P), shared by the precision-parametrised tests: per frame
    (bit_bl, bit_el, psnr_bl, psnr_el, symbols). bl: base-layer size (default H/2 x W/2). Read from the committed fixture
    (tests/golden/oracle_gops/, generated in the build container by tests/golden/make_oracle_gops.py -- the oracle's CPU minutes
    do not belong on the GPU box); a configuration without one is computed here."""
    from helpers import load_oracle_gop, compute_oracle_gop
    bl = bl or (H // 2, W // 2)
    key = (n, H, W, seed, gain, scale, bl)
    if key not in _ORACLE_GOPS:
        _ORACLE_GOPS[key] = load_oracle_gop(*key) or compute_oracle_gop(*key)
    return _ORACLE_GOPS[key]


def _gpu_gop_against_oracle(n, H, W, seed, gain, want_kernels=(), scale=2.0, bl=None):
    from lssvc_amd import IntraSS, LSSVC_extend, hip_ops
    from lssvc_amd.synth import synth_state_dict
    from lssvc_amd.preprocess import psnr
    clip, x_bl, rows = _oracle_gop(n, H, W, seed, gain, scale, bl)
    h, w = x_bl.shape[2:]
    inet = IntraSS.from_state_dict(synth_state_dict("intra_ss", seed, gain)).to(DEV).eval()
    pnet = LSSVC_extend()
    pnet.load_dict(synth_state_dict("lssvc_extend", seed, gain))
    pnet.to(DEV).eval()
    inet.set_scale_information(scale, (H, W), (0, 0, 0, 0))
    pnet.set_scale_information(scale, (H, W), (0, 0, 0, 0))
    if want_kernels:
        # the first frame of a type is range-audited, which splits the fused DepthConvBlock kernels into their convs
        # (bit-identical); this run must dispatch the kernels the benchmark is timed on
        inet.range_audit = pnet.range_audit = False
    dg, seen = None, set()
    for t in range(n):
        xb, xe = x_bl[t:t + 1], clip[t:t + 1]
        log_this = bool(want_kernels) and t in (0, 1)
        if log_this:
            hip_ops.OP_LOG = []
        try:
            if t == 0:
                g = inet.encode_decode(xb.to(DEV), xe.to(DEV), None, None)
                dg = {"ref_frame_bl": g["x_hat_bl"], "ref_frame_el": g["x_hat_el"], "ref_feature_bl": None,
                      "ref_feature_el": g["feature_el"]}
            else:
                g = pnet.encode_decode(xb.to(DEV), xe.to(DEV), dg)
                dg = g["dpb"]
        finally:
            if log_this:
                seen |= {op["kernel"] for op in hip_ops.OP_LOG}
                hip_ops.OP_LOG = None
        dg["ref_frame_bl"].clamp_(0, 1)
        dg["ref_frame_el"].clamp_(0, 1)
        o_bl, o_el, o_pbl, o_pel = rows[t][:4]
        assert abs(g["bit_bl"] - o_bl) / (h * w) <= 1e-5, (t, g["bit_bl"], o_bl)
        assert abs(g["bit_el"] - o_el) / (H * W) <= 1e-5, (t, g["bit_el"], o_el)
        assert abs(psnr(xe, dg["ref_frame_el"].cpu()) - o_pel) <= 1e-4, t
        assert abs(psnr(xb, dg["ref_frame_bl"].cpu()) - o_pbl) <= 1e-4, t
    for k in want_kernels:
        assert any(s.startswith(k) for s in seen), (k, sorted(seen))


TIE_STATS = {}             # precision -> [symbols compared, symbols that differ from the oracle's]
# The tie allowance (round 6): TWICE what the reference does to itself between two runs on other thread counts, read from
# tests/golden/x2_1080p_gop32_ref_t2.npz (helpers.tie_allowance / reference_self_disagreement; tests/test_gpu_golden_full.py says how it
# was made): 2 independent tie events (helpers.tie_clusters) and 22 differing symbols per latent plane, each off by exactly one, and
# 33.4 bits of a layer's count per differing symbol. Rounds 3-5 used numbers this file had chosen (2 / 16 / 40).
def _allowance():
    from helpers import tie_allowance
    return tie_allowance()


def _gpu_gop_symbol_aware(n, H, W, seed, gain, scale=2.0, bl=None):
    """The closed loop against the oracle with rounding ties told apart from errors (what tests/test_gpu_golden_full.py does
    against the reference's own symbols; LSSVC_net.py:193, img_entropy_models.py:237 are the round() calls). A differently
    ordered fp32 sum moves a value that lies within ~3e-7 of k + 1/2 across the tie about once per 10^6 symbols (DESIGN.md
    section 9); at 256x384 pixels one such symbol is 1.9e-4 bpp, beyond the 1e-5 bar by construction, and from then on a
    plain closed loop drifts away from the oracle's. So per frame:
      ENCODER pass (public API, estimate mode) from a DPB aligned with the oracle's: every symbol against the oracle's --
          at most the derived number of independent tie events per plane (spatial clusters), each difference by exactly one; bits inside 1e-5 bpp + the derived allowance per
          flipped symbol (no flip: the plain bar); PSNR inside 1e-4 dB when nothing flipped.
      DECODER pass (decoder role of the same codec functions on the ORACLE's symbols): PSNR of both layers inside 1e-4 dB,
          always; its outputs are the next frame's DPB.
    Flip counts are accumulated in TIE_STATS (test_tie_rate_report)."""
    from lssvc_amd import IntraSS, LSSVC_extend, hip_ops
    from lssvc_amd.synth import synth_state_dict
    from lssvc_amd.preprocess import psnr
    from helpers import decode_from_symbols, tie_clusters
    al = _allowance()
    clip, x_bl, rows = _oracle_gop(n, H, W, seed, gain, scale, bl)
    h, w = x_bl.shape[2:]
    inet = IntraSS.from_state_dict(synth_state_dict("intra_ss", seed, gain)).to(DEV).eval()
    pnet = LSSVC_extend()
    pnet.load_dict(synth_state_dict("lssvc_extend", seed, gain))
    pnet.to(DEV).eval()
    inet.update(force=True)                                  # the decoder role reads the bottleneck medians from the tables
    pnet.update(force=True)
    stats = TIE_STATS.setdefault(hip_ops.CONV_PRECISION, [0, 0])
    dg, report = None, []
    for t in range(n):
        xb, xe = x_bl[t:t + 1], clip[t:t + 1]
        net = inet if t == 0 else pnet
        net.set_scale_information(scale, (H, W), (0, 0, 0, 0))
        taps = net.taps = {}
        if t == 0:
            g = inet.encode_decode(xb.to(DEV), xe.to(DEV), None, None)
            enc = {"ref_frame_bl": g["x_hat_bl"], "ref_frame_el": g["x_hat_el"]}
        else:
            g = pnet.encode_decode(xb.to(DEV), xe.to(DEV), dg)
            enc = g["dpb"]
        net.taps = None
        o_bl, o_el, o_pbl, o_pel, syms = rows[t]
        assert set(taps) == set(syms), (sorted(taps), sorted(syms))
        flips = {"bl": 0, "el": 0}
        for key, want in syms.items():
            got = taps[key].reshape(-1).numpy()
            assert got.shape == want.shape, (key, got.shape, want.shape)
            d = got.astype(np.int32) - want.astype(np.int32)
            nz, mx, events = tie_clusters(d, key, H, W, h, w)
            assert nz <= al["max_flips"] and mx <= al["max_abs_diff"] and events <= al["max_events"], (t, key, nz, mx, events)
            flips[key[:2]] += nz
            stats[0] += d.size
            stats[1] += nz
        assert abs(g["bit_bl"] - o_bl) <= 1e-5 * h * w + al["flip_bits"] * flips["bl"], (t, g["bit_bl"], o_bl, flips)
        assert abs(g["bit_el"] - o_el) <= 1e-5 * H * W + al["flip_bits"] * flips["el"], (t, g["bit_el"], o_el, flips)
        if flips["bl"] == 0:
            assert abs(psnr(xb, enc["ref_frame_bl"].cpu().clamp(0, 1)) - o_pbl) <= 1e-4, t
            if flips["el"] == 0:
                assert abs(psnr(xe, enc["ref_frame_el"].cpu().clamp(0, 1)) - o_pel) <= 1e-4, t
        del g, enc
        d = decode_from_symbols(syms, H, W, h, w, t, inet, pnet, dg)
        dg = d["dpb"] if t else {"ref_frame_bl": d["x_hat_bl"], "ref_frame_el": d["x_hat_el"], "ref_feature_bl": None, "ref_feature_el": d["feature_el"]}
        dg["ref_frame_bl"].clamp_(0, 1)
        dg["ref_frame_el"].clamp_(0, 1)
        assert abs(psnr(xb, dg["ref_frame_bl"].cpu()) - o_pbl) <= 1e-4, t
        assert abs(psnr(xe, dg["ref_frame_el"].cpu()) - o_pel) <= 1e-4, t
        if flips["bl"] or flips["el"]:
            report.append((t, flips["bl"], flips["el"]))
    print("symbol-aware loop %dx%d seed %d %s: frames with flipped symbols (t, BL, EL) = %s" % (H, W, seed, hip_ops.CONV_PRECISION, report or "none"))


@pytest.mark.parametrize("seed", [7, pytest.param(8, marks=pytest.mark.slow), pytest.param(9, marks=pytest.mark.slow)])
def test_gop_drift_symbol_aware(seed, precision):
    """The GOP of test_gop_drift_vs_oracle on three more seeds, with ties handled instead of avoided: all 32 frames on every seed
    (round 5: the oracle's side is a fixture, tests/golden/oracle_gops/, so only GPU seconds are spent here); seed 7 runs by default,
    seeds 8 and 9 behind --runslow (profiles/r04_slow_gpu_tests.txt holds a run)."""
    _gpu_gop_symbol_aware(32, 128, 128, seed, 0.55)


def test_gop_drift_vs_oracle(precision):
    """Errors feed forward through the DPB: code a full 32-frame GOP (1 I + 31 P, the GOP length of BASELINE
    configs[1]) and hold EVERY frame to the per-frame bars (|d bpp| <= 1e-5, |d PSNR| <= 1e-4 dB) against the CPU
    oracle run closed-loop on the same inputs, at the bench's weight gain."""
    _gpu_gop_against_oracle(32, 128, 128, 5, 0.55)


def test_frames_384x640_vs_oracle(precision):
    """I + P + P at EL 384x640 / BL 192x320 against the CPU oracle: the smallest size at which the full-resolution 3x3
    and 7x7 convs dispatch the persistent kernels and the narrow-output convs their RPW = 4 instantiations, i.e. the
    kernels the 1080p benchmark is timed on, inside the whole network, at the north-star bars."""
    want = ("conv3_f16x3p_kernel<4", "conv3_f16x3p_kernel<3", "conv_f16x3_kernel<1, 4, 7, 1>",
            "conv7_f16x3p_kernel<2", "conv7_f16x3p_kernel<4", "ffn_f16x3_kernel", "dwpre_f16x3_kernel") if precision == "f16x3" else ()
    _gpu_gop_against_oracle(3, 384, 640, 3, 0.55, want_kernels=want)


@pytest.mark.parametrize("seed", [7, 8, pytest.param(9, marks=pytest.mark.slow)])
@pytest.mark.parametrize("ph,pw,scale,frames", [
    (240, 416, 2.0, 3), (240, 416, 1.5, 3),
    # (the oracle's side of every shape is a fixture since round 5; the larger shapes stay behind --runslow / LSSVC_SLOW=1 for their GPU seconds)
    pytest.param(480, 832, 2.0, 2, marks=pytest.mark.slow), pytest.param(720, 1280, 1.5, 2, marks=pytest.mark.slow)])
def test_dataset_picture_sizes_vs_oracle(ph, pw, scale, frames, seed, precision):
    """The picture sizes of the reference's own test set below 720p (HEVC class C 832x480 and class D 416x240,
    recommend_test_config.json) at both of its ratios, padded as test.py pads them (common.py:48-86): EL 512x896 / BL 256x448,
    EL 384x576 / BL 256x384 (ratio 1.5) and EL 256x512 / BL 128x256 -- map widths of 14, 9 and 8 sixty-fourths, which none of
    the other shapes has -- and class E 1280x720 at ratio 1.5 (EL 768x1344 / BL 512x896: 21 sixty-fourths); I + P + P at
    416x240, I + P at the larger sizes (the oracle's seconds bound the suite), against the CPU oracle, seeds 7, 8 and 9, with
    the symbol-aware comparison (_gpu_gop_symbol_aware): round 3 ran the plain comparison, found that seed 7 at 416x240 /
    ratio 1.5 in the f32 mode flips ONE base-layer symbol of the I-frame at a tie (18.49 bits = 1.9e-4 bpp at 256x384 pixels)
    and picked seed 8; a tie is now recognised for what it is on every seed, counted, and the loop carries on from the
    oracle's symbols. Run by default: 416x240 at ratio 2; the other shapes are marked slow (their oracle runs take 45-80 s each)."""
    from lssvc_amd.preprocess import interlayer_padding
    pad = interlayer_padding(ph, pw, scale)
    (H, W), bl = pad["HR_padded_size"], pad["LR_padded_size"]
    _gpu_gop_symbol_aware(frames, H, W, seed, 0.55, scale=scale, bl=bl)


def test_tie_rate_report():
    """Runs after the symbol-aware loops of this file (pytest keeps file order): symbols compared and symbols that fell on the
    other side of a rounding tie than the oracle's, per conv precision, as flips per 10^6 symbols; kept in
    gpurun_out/tie_stats.json (DESIGN.md section 9 quotes it). The fp32 noise floor predicts O(1) per 10^6."""
    import json
    import os
    if not TIE_STATS:
        pytest.skip("no symbol-aware loop ran in this session")
    out = {k: {"symbols": v[0], "flipped": v[1], "flips_per_million": round(1e6 * v[1] / max(v[0], 1), 3)} for k, v in TIE_STATS.items()}
    print("tie statistics:", out)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", "tie_stats.json"), "w") as f:
        json.dump(out, f, indent=1)
    for k, v in out.items():
        assert v["flips_per_million"] <= 25.0, (k, v)


def test_config1_single_iframe_256(precision):
    """BASELINE configs[0] (SURVEY 8d "Config 1"): IntraSS, one 256x256 frame, x2, estimate mode, the reference's own
    CPU-runnable case: x_el = rand(1,3,256,256) under manual_seed(0), x_bl = bicubic(x_el, 128x128).clamp(0,1).
    GPU against the CPU oracle on the same tensors: |dbpp| <= 1e-5, |dPSNR| <= 1e-4 dB."""
    from lssvc_amd import IntraSS
    from lssvc_amd.preprocess import imresize_bicubic, psnr
    from lssvc_amd.synth import synth_state_dict
    from lssvc_oracle.intra import intra_forward
    torch.manual_seed(0)
    x_el = torch.rand(1, 3, 256, 256)
    x_bl = imresize_bicubic(x_el, (128, 128)).clamp_(0, 1)
    sd = synth_state_dict("intra_ss", 0, 0.6)
    with torch.no_grad():
        want = intra_forward(sd, x_bl, x_el, (256, 256))
    net = IntraSS.from_state_dict(sd).to(DEV).eval()
    net.set_scale_information(2.0, (256, 256), (0, 0, 0, 0))
    got = net.encode_decode(x_bl.to(DEV), x_el.to(DEV), None, None, 128, 128, 256, 256)
    assert abs(got["bit_bl"] - want["bit_bl"]) / (128 * 128) <= 1e-5 and abs(got["bit_el"] - want["bit_el"]) / (256 * 256) <= 1e-5
    for k, x in (("x_hat_bl", x_bl), ("x_hat_el", x_el)):
        g, w = got[k].cpu().clamp(0, 1), want[k].clamp(0, 1)
        assert abs(psnr(x, g) - psnr(x, w)) <= 1e-4
        np.testing.assert_allclose(got[k].cpu().numpy(), want[k].numpy(), atol=2e-4, rtol=0)


def test_boundary_tensors_are_zero_copy_channels_last():
    """What the model hands to the caller (hip_ops.T.to_nchw) is a channels_last view of its NHWC buffer: same storage,
    NCHW shape. Handing it back costs no transpose, in-place edits by the caller (test.py:249-250 clamps the frames)
    are seen, and a plain NCHW-contiguous tensor with equal contents still goes through the transpose."""
    from lssvc_amd.hip_ops import T
    x = torch.rand(1, 8, 6, 10, device=DEV) * 3 - 1
    t = T.from_nchw(x)                                          # NCHW-contiguous input: transposed once
    y = t.to_nchw()
    assert y.shape == x.shape and torch.equal(y, x) and y.is_contiguous(memory_format=torch.channels_last)
    assert y.data_ptr() == t.buf.data_ptr()                     # no copy on the way out ...
    t2 = T.from_nchw(y)
    assert t2.buf.data_ptr() == t.buf.data_ptr()                # ... and none on the way back in
    y.clamp_(0, 1)
    assert torch.equal(T.from_nchw(y).to_nchw(), x.clamp(0, 1)) and torch.equal(t.to_nchw(), x.clamp(0, 1))
    z = y.contiguous()                                          # an NCHW-contiguous copy takes the transpose path
    tz = T.from_nchw(z)
    assert tz.buf.data_ptr() != t.buf.data_ptr() and torch.equal(tz.to_nchw(), y)
    c = t.to_nchw(copy=True)
    assert c.is_contiguous() and c.data_ptr() != t.buf.data_ptr() and torch.equal(c, y)
    s = t.slice(0, 4).to_nchw()                                 # a channel slice cannot be a dense view: copied
    assert s.is_contiguous() and torch.equal(s, y[:, 0:4])


def test_full_size_runs_are_bit_deterministic():
    """BASELINE configs[1] shape (EL 1152x1920 / BL 576x960): an I-frame and a P-frame coded twice from the same inputs
    must agree bit for bit (bits, reconstructions, features). The persistent / warp-specialised kernels hand data
    between waves through LDS rings and barriers; a missing fence would show up here as run-to-run noise long before
    it moved a PSNR."""
    from lssvc_amd import IntraSS, LSSVC_extend
    from lssvc_amd.preprocess import make_layers
    from lssvc_amd.synth import synth_clip, synth_state_dict
    inet = IntraSS.from_state_dict(synth_state_dict("intra_ss", 0, 0.55)).to(DEV).eval()
    pnet = LSSVC_extend()
    pnet.load_dict(synth_state_dict("lssvc_extend", 0, 0.55))
    pnet.to(DEV).eval()
    clip = synth_clip(2, 1080, 1920, seed=2).to(DEV).float() / 255.0
    runs = []
    for _ in range(2):
        xb0, xe0, pad = make_layers(clip[0:1], 2.0)
        xb1, xe1, _ = make_layers(clip[1:2], 2.0)
        inet.set_scale_information(2.0, pad["HR_padded_size"], (0, 0, 0, 0))
        pnet.set_scale_information(2.0, pad["HR_padded_size"], (0, 0, 0, 0))
        ri = inet.encode_decode(xb0, xe0, None, None)
        dpb = {"ref_frame_bl": ri["x_hat_bl"].clone().clamp_(0, 1), "ref_frame_el": ri["x_hat_el"].clone().clamp_(0, 1),
               "ref_feature_bl": None, "ref_feature_el": ri["feature_el"]}
        rp = pnet.encode_decode(xb1, xe1, dpb)
        runs.append((ri["bit_bl"], ri["bit_el"], rp["bit_bl"], rp["bit_el"], ri["x_hat_el"].clone(), ri["feature_el"].clone(),
                     rp["dpb"]["ref_frame_el"].clone(), rp["dpb"]["ref_feature_el"].clone(), rp["dpb"]["ref_feature_bl"].clone(),
                     rp["mv_hat"].clone()))
    a, b = runs
    assert a[:4] == b[:4], (a[:4], b[:4])
    for x, y in zip(a[4:], b[4:]):
        assert torch.equal(x, y)
    assert 0.0 < a[3] / (1152 * 1920) < 8.0                      # sane bpp
