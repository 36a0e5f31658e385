"""CPU-only checks: the C-ABI library loads and exports every symbol include/lssvc_hip.h declares,
the weight re-layouts are equivalent re-orderings, checkpoints are validated strictly."""
import os
import re

import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from lssvc_amd import _lib
    header = open(os.path.join(ROOT, "include", "lssvc_hip.h")).read()
    declared = set(re.findall(r"\b(lssvc_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert getattr(_lib.lib, name) is not None
    assert _lib.lib.lssvc_version() == 1
    assert _lib.lib.lssvc_reduce_workspace_bytes() >= 8 * 256


def test_argument_validation_without_gpu():
    """Shape errors are caught on the host before any launch, with a readable message."""
    import ctypes as C
    from lssvc_amd import _lib
    d = _lib.ConvDesc()
    d.n_in = 5
    assert _lib.lib.lssvc_conv2d(C.byref(d), None) != 0
    assert b"n_in" in _lib.lib.lssvc_last_error()
    with pytest.raises(_lib.LssvcHipError):
        _lib.check(_lib.lib.lssvc_conv2d(C.byref(d), None))


def _unlayout(wp, cout, splits, kh, kw):
    """Invert weights.layout_conv: [chunk][ky][kx][m][8] -> (Cout, Cin, KH, KW)."""
    nchunk = wp.shape[0]
    w = wp.permute(3, 0, 4, 1, 2).reshape(wp.shape[3], nchunk * 8, kh, kw)[:cout]
    segs, a = [], 0
    for c in splits:
        segs.append(w[:, a:a + c])
        a += (c + 7) // 8 * 8
    return torch.cat(segs, 1)


def test_conv_layout_roundtrip():
    from weights_torch_ref import layout_conv
    w = torch.randn(51, 3 + 48, 3, 3)
    b = torch.randn(51)
    wp, bp, cout, m_pad = layout_conv(w, b, [3, 48], False)
    assert wp.shape == (1 + 6, 3, 3, 64, 8) and m_pad == 64 and cout == 51
    assert torch.equal(_unlayout(wp, 51, [3, 48], 3, 3), w) and torch.equal(bp[:51], b) and bp[51:].abs().sum() == 0


def test_pixel_shuffle_permutation():
    from weights_torch_ref import layout_conv
    w = torch.randn(32, 16, 3, 3)
    b = torch.randn(32)
    x = torch.randn(1, 16, 6, 7)
    wp, bp, cout, m_pad = layout_conv(w, b, [16], True)
    wq = _unlayout(wp, 32, [16], 3, 3)
    y = F.conv2d(x, wq, bp[:32], padding=1)                       # channels are (q, c)-major: m = q*8 + c
    y = y.reshape(1, 4, 8, 6, 7)
    out = torch.zeros(1, 8, 12, 14)
    for q in range(4):
        out[:, :, (q >> 1)::2, (q & 1)::2] = y[:, q]
    assert torch.allclose(out, F.pixel_shuffle(F.conv2d(x, w, b, padding=1), 2), atol=1e-6)


@pytest.mark.parametrize("stride", [1, 2])
def test_conv_transpose_rewrite(stride):
    from weights_torch_ref import conv_t_as_conv
    w = torch.randn(10, 6, 3, 3)
    b = torch.randn(6)
    x = torch.randn(1, 10, 5, 7)
    want = F.conv_transpose2d(x, w, b, stride=stride, padding=1, output_padding=stride - 1)
    w2, b2, pad, ps = conv_t_as_conv(w, b, stride)
    if stride == 1:
        got = F.conv2d(x, w2, b2, padding=pad)
    else:
        y = F.conv2d(F.pad(x, (0, 1, 0, 1)), w2, b2)             # 2x2 taps read (i..i+1, j..j+1), zero beyond the edge
        y = y.reshape(1, 4, 6, 5, 7)
        got = torch.zeros(1, 6, 10, 14)
        for q in range(4):
            got[:, :, (q >> 1)::2, (q & 1)::2] = y[:, q]
    assert torch.allclose(got, want, atol=1e-5)


def test_strict_checkpoint_validation():
    from lssvc_amd.synth import synth_state_dict
    from lssvc_amd.weights import validate, CheckpointError, strip_module_prefix
    sd = synth_state_dict("lssvc_extend", 0, 0.6)
    validate(strip_module_prefix({"module." + k: v for k, v in sd.items()}), "lssvc_extend")
    bad = dict(sd)
    bad.pop("align.fusion.weight")
    with pytest.raises(CheckpointError):
        validate(bad, "lssvc_extend")
    bad = dict(sd)
    bad["extra.weight"] = torch.zeros(1)
    with pytest.raises(CheckpointError):
        validate(bad, "lssvc_extend")
    bad = dict(sd)
    bad["align.fusion.weight"] = torch.zeros(48, 6, 3, 3)
    with pytest.raises(CheckpointError):
        validate(bad, "lssvc_extend")


def test_synth_is_order_independent_and_seeded():
    from lssvc_amd.synth import synth_state_dict, synth_clip
    a, b = synth_state_dict("intra_ss", 3, 0.6), synth_state_dict("intra_ss", 3, 0.6)
    assert all(torch.equal(a[k], b[k]) for k in a)
    c = synth_state_dict("intra_ss", 4, 0.6)
    assert not torch.equal(a["g_a.conv1.weight"], c["g_a.conv1.weight"])
    clip = synth_clip(2, 64, 64, seed=1)
    assert clip.dtype == torch.uint8 and clip.shape == (2, 3, 64, 64) and torch.equal(clip, synth_clip(2, 64, 64, seed=1))


@pytest.mark.parametrize("case", ["x2_128_ipp", "x1_5_192_ip", "x2_128x256_ip"])
def test_bl_frame_generation_matches_reference(case):
    """preprocess.imresize_bicubic reproduces the reference's imresize on the golden clips
    (the fixtures' x_bl were produced by src/utils/core.py:imresize on x_el)."""
    import numpy as np
    from helpers import load_case
    from lssvc_amd.preprocess import imresize_bicubic
    z, m = load_case(case)
    x_el = torch.from_numpy(z["x_el_u8"]).float() / 255.0
    got = imresize_bicubic(x_el, (m["h"], m["w"])).clamp_(0, 1)
    np.testing.assert_allclose(got.numpy(), z["x_bl"], atol=1e-6, rtol=0)


def test_interlayer_padding_1080p():
    from lssvc_amd.preprocess import interlayer_padding
    p = interlayer_padding(1080, 1920, 2.0)
    assert p["HR_padded_size"] == (1152, 1920) and p["LR_padded_size"] == (576, 960) and p["LR_size"] == (540, 960)
    p = interlayer_padding(1080, 1920, 1.5)
    assert p["HR_padded_size"][0] % 96 == 0 and p["LR_padded_size"][0] % 64 == 0


def test_f16x3_layout_reconstructs_weights_and_prescales():
    """layout_conv_f16x3: hi + lo (times the returned power-of-two unscale) gives the fp32 weights back to ~2^-21
    relative, in the [chunk16][ky][kx][m][16] order, concat segments zero-padded to 16; the prescale keeps the lo plane
    out of fp16's subnormal range for typical weight magnitudes."""
    import math
    from weights_torch_ref import layout_conv_f16x3
    g = torch.Generator().manual_seed(3)
    w = torch.randn(40, 24 + 8, 3, 3, generator=g) * 0.02
    planes, unscale = layout_conv_f16x3(w, [24, 8], False)
    assert planes.shape == (2, 3, 3, 3, 48, 16) and planes.dtype == torch.float16            # 24 -> 2 chunks, 8 -> 1 chunk; M 40 -> 48
    assert math.log2(unscale) == round(math.log2(unscale)) and 2 ** 11 <= float(w.abs().max()) / unscale < 2 ** 12
    rec = (planes[0].float() + planes[1].float()) * unscale                                  # [chunk][ky][kx][m][16]
    seg0 = rec[0:2].permute(3, 0, 4, 1, 2).reshape(48, 32, 3, 3)[:40, :24]
    seg1 = rec[2].permute(2, 3, 0, 1)[:40, :8]
    assert (seg0 - w[:, :24]).abs().max() <= 2 ** -20 * w.abs().max() and (seg1 - w[:, 24:]).abs().max() <= 2 ** -20 * w.abs().max()
    assert rec[0:2].permute(3, 0, 4, 1, 2).reshape(48, 32, 3, 3)[:, 24:].abs().max() == 0   # segment padding is zero
    assert rec[:, :, :, 40:].abs().max() == 0                                                 # M padding is zero
    lo = planes[1].float().abs()
    assert (lo[lo > 0] >= 2.0 ** -14).float().mean() > 0.99                                  # lo parts are normal fp16 numbers


def test_ffn_layout_chained_k_order():
    """layout_ffn_f16x3 puts W1 / W2 in the K order in which ffn_f16x3.hip chains accumulator fragments into B
    operands: K position (pair p, k = 8g + j) <-> channel 16 * (2p + (j >> 2)) + 4g + (j & 3)."""
    from weights_torch_ref import layout_ffn_f16x3, layout_pw_natural_f16x3
    g = torch.Generator().manual_seed(4)
    c, hidden = 48, 192
    w1 = torch.randn(hidden, c, 1, 1, generator=g) * 0.05
    w2 = torch.randn(c, hidden, 1, 1, generator=g) * 0.05
    (a, ua), (b, ub) = layout_ffn_f16x3(w1, w2)
    t, cf, s = hidden // 32, c // 16, 2
    A = ((a[:a.numel() // 2].float() + a[a.numel() // 2:].float()) * ua).reshape(t, 2, s, 16, 32)
    B = ((b[:b.numel() // 2].float() + b[b.numel() // 2:].float()) * ub).reshape(t, cf, 16, 32)
    for tt, f, ss, i, k in [(0, 0, 0, 0, 0), (5, 1, 1, 15, 31), (2, 1, 0, 7, 13), (3, 0, 1, 3, 5), (4, 1, 1, 9, 20)]:
        gg, j = k // 8, k % 8
        ic = (2 * ss + (j >> 2)) * 16 + 4 * gg + (j & 3)
        want = w1[(2 * tt + f) * 16 + i, ic, 0, 0].item() if ic < c else 0.0
        assert abs(A[tt, f, ss, i, k].item() - want) <= 1e-6
    for tt, m, i, k in [(0, 0, 0, 0), (5, 2, 15, 31), (2, 1, 7, 13), (1, 0, 4, 22)]:
        gg, j = k // 8, k % 8
        hc = (2 * tt + (j >> 2)) * 16 + 4 * gg + (j & 3)
        assert abs(B[tt, m, i, k].item() - w2[m * 16 + i, hc, 0, 0].item()) <= 1e-6
    wp = torch.randn(48, 40, 1, 1, generator=g) * 0.05                                        # leading conv: natural K order, K padded to 64
    blob, up = layout_pw_natural_f16x3(wp)
    P = ((blob[:blob.numel() // 2].float() + blob[blob.numel() // 2:].float()) * up).reshape(3, 2, 16, 32)
    assert abs(P[2, 1, 5, 3].item() - wp[37, 35, 0, 0].item()) <= 1e-6 and P[:, 1, :, 8:].abs().max() == 0


def _store(sd):
    from lssvc_amd.weights import WeightStore
    return WeightStore(sd, torch.device("cpu"))            # host-only use of the library's weight preparation (no launch)


def test_weight_prep_matches_torch_layouts():
    """lssvc_prepare_weights (csrc/weight_prep.cpp: what WeightStore and the engine's checkpoint loader both call) against
    the torch restatement of every layout (tests/weights_torch_ref.py, the implementation rounds 1-3 shipped): the same
    bytes for convs (plain, concatenated inputs, sub-pixel), their fp16 hi / lo planes and prescale, both ConvTranspose
    rewrites, depthwise, GDN of both flavours and the FFN images."""
    import weights_torch_ref as R
    g = torch.Generator().manual_seed(11)
    rn = lambda *sh: torch.randn(*sh, generator=g)
    sd = {"a.weight": rn(51, 3 + 48, 3, 3) * 0.05, "a.bias": rn(51), "s.weight": rn(32, 16, 3, 3), "s.bias": rn(32), "nb.weight": rn(16, 8, 1, 1),
          "t.weight": rn(10, 6, 3, 3), "t.bias": rn(6), "d.weight": rn(24, 1, 3, 3), "d.bias": rn(24),
          "f.conv.0.weight": rn(192, 48, 1, 1) * 0.05, "f.conv.0.bias": rn(192), "f.conv.2.weight": rn(48, 192, 1, 1) * 0.05, "f.conv.2.bias": rn(48),
          "p.weight": rn(48, 40, 1, 1) * 0.05, "p.bias": rn(48),
          "gi.beta": rn(16).abs() + 0.5, "gi.gamma": rn(16, 16).abs() * 0.1, "gi.beta_reparam.lower_bound.bound": torch.tensor([(1e-6 + 2.0 ** -36) ** 0.5]),
          "gi.beta_reparam.pedestal": torch.tensor([2.0 ** -36]), "gi.gamma_reparam.lower_bound.bound": torch.tensor([2.0 ** -18]),
          "gi.gamma_reparam.pedestal": torch.tensor([2.0 ** -36]), "gp.beta": rn(32).abs() + 0.5, "gp.gamma": rn(32, 32) * 0.1}
    W = _store(sd)
    for name, splits, ps in (("a", [3, 48], False), ("s", [16], True), ("nb", [8], False)):
        wp, bp, cout, m_pad, kh, kw = W.conv(name, splits, ps)
        rw, rb, rc, rm = R.layout_conv(sd[name + ".weight"], sd.get(name + ".bias"), splits, ps)
        assert (cout, m_pad, kh, kw) == (rc, rm) + tuple(sd[name + ".weight"].shape[2:])
        assert torch.equal(wp, rw.reshape(-1)) and torch.equal(bp, rb)
        planes, unscale = W.conv_f16x3(name, splits, ps)
        rp, ru = R.layout_conv_f16x3(sd[name + ".weight"], splits, ps)
        assert unscale == ru and torch.equal(planes.view(torch.int16), rp.reshape(-1).view(torch.int16))
    for stride in (1, 2):
        wp, bp, cout, m_pad, kh, kw, pad, ps = W.conv_t("t", stride)
        w2, b2, rpad, rps = R.conv_t_as_conv(sd["t.weight"], sd["t.bias"], stride)
        rw, rb, rc, rm = R.layout_conv(w2, b2, [w2.shape[1]], False)
        assert (cout, m_pad, kh, kw, pad, ps) == (rc, rm, w2.shape[2], w2.shape[3], rpad, rps)
        assert torch.equal(wp, rw.reshape(-1)) and torch.equal(bp, rb)
    dw, db = W.dwconv("d")
    assert torch.equal(dw, sd["d.weight"].reshape(24, 9).t().reshape(-1)) and torch.equal(db, sd["d.bias"])
    for name, flavour in (("gi", "intra"), ("gp", "inter")):
        wp, bp, cout, m_pad, _, _ = W.gdn(name, flavour)
        beta, gamma = sd[name + ".beta"], sd[name + ".gamma"]
        if flavour == "intra":
            beta = torch.max(beta, sd[name + ".beta_reparam.lower_bound.bound"]) ** 2 - sd[name + ".beta_reparam.pedestal"]
            gamma = torch.max(gamma, sd[name + ".gamma_reparam.lower_bound.bound"]) ** 2 - sd[name + ".gamma_reparam.pedestal"]
        else:
            beta = torch.max(beta, torch.ones_like(beta) * R._BETA_BOUND) ** 2 - R._PEDESTAL
            gamma = torch.max(gamma, torch.ones_like(gamma) * R._REPARAM_OFFSET) ** 2 - R._PEDESTAL
        c = gamma.shape[0]
        rw, rb, _, _ = R.layout_conv(gamma.reshape(c, c, 1, 1), beta, [c], False)
        rp, ru = R.layout_conv_f16x3(gamma.reshape(c, c, 1, 1), [c], False)
        planes, unscale = W.gdn_f16x3(name, flavour)
        assert torch.equal(wp, rw.reshape(-1)) and torch.equal(bp, rb) and unscale == ru
        assert torch.equal(planes.view(torch.int16), rp.reshape(-1).view(torch.int16))
    rec = W.ffn_f16x3("f", "p")
    (a, ua), (b, ub) = R.layout_ffn_f16x3(sd["f.conv.0.weight"], sd["f.conv.2.weight"])
    blob, up = R.layout_pw_natural_f16x3(sd["p.weight"])
    assert (rec["u1"], rec["u2"], rec["up"], rec["hidden"], rec["C"], rec["pre_cin"]) == (ua, ub, up, 192, 48, 40)
    for got, want in ((rec["w1"], a), (rec["w2"], b), (rec["wp"], blob)):
        assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    assert torch.equal(rec["b1"], sd["f.conv.0.bias"]) and torch.equal(rec["b2"], sd["f.conv.2.bias"]) and torch.equal(rec["bp"], sd["p.bias"])
    rec2 = W.ffn_f16x3("f")
    assert "wp" not in rec2 and torch.equal(rec2["w1"].view(torch.int16), a.view(torch.int16))


def test_weight_prep_tables_and_whole_checkpoints():
    """The BitEstimator / EntropyBottleneck tables (softplus / tanh in double, rounded once: within one ulp of torch's fp32
    kernels) and every conv of both synthetic checkpoints through the library's preparation, against the torch restatement;
    a missing tensor and a wrong split are errors, not garbage."""
    import torch.nn.functional as F2
    import weights_torch_ref as R
    from lssvc_amd import _lib
    from lssvc_amd.synth import synth_state_dict
    sd = synth_state_dict("lssvc_extend", 3, 0.6)
    W = _store(sd)
    be = W.bit_estimator("bit_estimator_z")
    rows = []
    for i in (1, 2, 3):
        rows += [F2.softplus(sd["bit_estimator_z.f%d.h" % i]), sd["bit_estimator_z.f%d.b" % i], torch.tanh(sd["bit_estimator_z.f%d.a" % i])]
    rows += [F2.softplus(sd["bit_estimator_z.f4.h"]), sd["bit_estimator_z.f4.b"]]
    want = torch.stack([r.reshape(-1) for r in rows], 0)
    assert be.shape == want.shape and torch.allclose(be, want, rtol=2e-7, atol=0)
    assert torch.equal(be[1], want[1])                                                       # the plain rows are copies
    si = synth_state_dict("intra_ss", 3, 0.6)
    Wi = _store(si)
    eb = Wi.entropy_bottleneck("entropy_bottleneck")
    assert eb.shape == (59, 64) and torch.equal(eb[58], si["entropy_bottleneck.quantiles"][:, 0, 1])
    m0 = F2.softplus(si["entropy_bottleneck._matrices.0"])
    assert torch.allclose(eb[0:3], torch.stack([m0[:, j, 0] for j in range(3)]), rtol=2e-7, atol=0)
    f3 = torch.tanh(si["entropy_bottleneck._factors.3"])
    assert torch.allclose(eb[55:58], torch.stack([f3[:, j, 0] for j in range(3)]), rtol=2e-7, atol=0)
    n = 0
    for store, d in ((W, sd), (Wi, si)):
        for k, v in d.items():
            if k.endswith(".weight") and v.dim() == 4 and v.shape[1] > 1 and "deconv" not in k and v.shape[2] == v.shape[3] and n < 400:
                name = k[:-7]
                planes, unscale = store.conv_f16x3(name, [v.shape[1]])
                rp, ru = R.layout_conv_f16x3(v, [v.shape[1]], False)
                assert unscale == ru and torch.equal(planes.view(torch.int16), rp.reshape(-1).view(torch.int16)), k
                n += 1
    assert n > 200
    with pytest.raises(_lib.LssvcHipError, match="no tensor"):
        W.conv("no.such.layer", [3])
    with pytest.raises(_lib.LssvcHipError, match="add up"):
        W.conv_f16x3(next(k[:-7] for k, v in sd.items() if k.endswith(".weight") and v.dim() == 4), [1, 2])


def test_weight_prep_rejects_checkpoints_of_other_shapes():
    """lssvc_prepare_weights is fed raw checkpoint dumps by the engine (lssvc_engine_load_checkpoint), so a checkpoint
    whose tensors have other shapes than the architecture's -- other EntropyBottleneck filter sizes, short biases, a depthwise
    weight that is not (C,1,3,3), a ConvFFN whose bias does not match, a non-positive input segment -- must be an error message
    BEFORE anything is written, never an out-of-bounds read or a heap overflow."""
    from lssvc_amd import _lib
    from lssvc_amd.synth import synth_state_dict
    si = synth_state_dict("intra_ss", 3, 0.6)
    bad = dict(si)
    bad["entropy_bottleneck._matrices.1"] = torch.zeros(64, 5, 3)                    # filters (3, 5, ...) would need 67 rows
    with pytest.raises(_lib.LssvcHipError, match="_matrices.1 must be"):
        _store(bad).entropy_bottleneck("entropy_bottleneck")
    bad = dict(si)
    bad["entropy_bottleneck.quantiles"] = torch.zeros(64, 1, 2)
    with pytest.raises(_lib.LssvcHipError, match="quantiles must be"):
        _store(bad).entropy_bottleneck("entropy_bottleneck")
    sd = synth_state_dict("lssvc_extend", 3, 0.6)
    bad = dict(sd)
    bad["bit_estimator_z.f2.a"] = torch.zeros(1, 7, 1, 1)
    with pytest.raises(_lib.LssvcHipError, match="elements, expected"):
        _store(bad).bit_estimator("bit_estimator_z")
    g = torch.Generator().manual_seed(1)
    rn = lambda *sh: torch.randn(*sh, generator=g)
    with pytest.raises(_lib.LssvcHipError, match="bias has 5 elements"):
        _store({"a.weight": rn(16, 8, 3, 3), "a.bias": rn(5)}).conv("a", [8])
    with pytest.raises(_lib.LssvcHipError, match="depthwise weight must be"):
        _store({"d.weight": rn(24, 3, 3), "d.bias": rn(24)}).dwconv("d")
    with pytest.raises(_lib.LssvcHipError, match="bias has"):
        _store({"d.weight": rn(24, 1, 3, 3), "d.bias": rn(8)}).dwconv("d")
    with pytest.raises(_lib.LssvcHipError, match="bias has"):
        _store({"t.weight": rn(10, 6, 3, 3), "t.bias": rn(10)}).conv_t("t", 2)
    ffn = {"f.conv.0.weight": rn(64, 32, 1, 1), "f.conv.0.bias": rn(8), "f.conv.2.weight": rn(32, 64, 1, 1), "f.conv.2.bias": rn(32)}
    with pytest.raises(_lib.LssvcHipError, match="conv.0.bias has 8 elements"):
        _store(ffn).ffn_f16x3("f")
    with pytest.raises(_lib.LssvcHipError, match="input segment 1 has -8 channels"):
        _store({"a.weight": rn(16, 8, 3, 3)}).conv("a", [16, -8])
    with pytest.raises(_lib.LssvcHipError, match="GDN wants|elements, expected"):
        _store({"g.beta": rn(16), "g.gamma": rn(16, 16), "g.beta_reparam.lower_bound.bound": rn(2), "g.beta_reparam.pedestal": rn(1),
                "g.gamma_reparam.lower_bound.bound": rn(1), "g.gamma_reparam.pedestal": rn(1)}).gdn("g", "intra")


def test_framing_matches_reference_fixture(tmp_path):
    """tests/golden/framing.json was written by the reference's own stream_helper.py (make_framing_golden.py imports it
    from /root/reference): our framing must produce the same file bytes from the same strings, parse the reference's
    files back, and agree on get_downsampled_shape."""
    import json
    from lssvc_amd import bitstream as B
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "framing.json")))
    assert len(fx["i_frames"]) >= 5 and len(fx["p_frames"]) >= 4 and len(fx["shapes"]) >= 10
    p = str(tmp_path / "f.bin")
    for c in fx["i_frames"]:
        y, z, want = bytes.fromhex(c["y"]), bytes.fromhex(c["z"]), bytes.fromhex(c["file"])
        B.encode_i(c["height"], c["width"], y, z, p)
        assert open(p, "rb").read() == want
        with open(p, "wb") as f:
            f.write(want)
        assert B.decode_i(p) == (c["height"], c["width"], y, z) and B.filesize(p) == len(want)
    for c in fx["p_frames"]:
        s, want = bytes.fromhex(c["string"]), bytes.fromhex(c["file"])
        B.encode_p(s, p)
        assert open(p, "rb").read() == want
        assert B.decode_p(p) == s
    for c in fx["shapes"]:
        assert list(B.get_downsampled_shape(*c["args"])) == c["out"]


def test_counted_waits_of_the_persistent_kernels_cover_their_weight_dma():
    """the persistent 3x3 kernels' producers publish an LDS buffer after `s_waitcnt vmcnt(N)` -- N younger patch loads stay in
    flight, the weight DMA issued before them must have landed. Correct only while the compiler emits at least N vector-memory
    instructions between that DMA and the wait; round 5 checked the ISA by hand. tools/p3_waitcnt_check.py does it mechanically on the
    in-tree objects (the ones that are linked into liblssvc_hip.so): every instantiation of the patch-ring schedule (round 5) and of the
    register-prefetch and split-roles schedules (round 6) must pass, and the tool must actually have found their counted waits (the
    hand-written ones carry an expcnt(6) mark in the ISA: conv3_f16x3p_kernel.h, p3_waitcnt)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    objs = [os.path.join(root, "lssvc_amd", "csrc", n) for n in ("conv3_f16x3p.o", "conv3_f16x3p_r.o", "conv3_f16x3p_r2.o", "conv3_f16x3p_r3.o")]
    if not all(os.path.exists(o) for o in objs) or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("objects not built here (python -c 'import __graft_entry__ as g; g.build()') or no llvm-objdump")
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "p3_waitcnt_check.py")] + objs, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert "VIOLATION" not in p.stdout and " 0 with a counted wait that does not cover" in p.stdout
    marked = lambda ln: ln.split("marked waits (N, loads in front): ")[1]
    ring = [ln for ln in p.stdout.splitlines() if "<3, 0, 0, 0, 1, 0, 0, 0, 0, 0>" in ln]                  # 48-channel kernels: three patch buffers
    pf = [ln for ln in p.stdout.splitlines() if "<4, 0, 0, 0, 2, 0, 0, 1, 0, 0>" in ln]                    # stride 2 with the register prefetch
    roles = [ln for ln in p.stdout.splitlines() if "<4, 0, 0, 0, 2, 0, 0, 3, 0, 0>" in ln]                 # stride 2, split roles (three patch waves: 12 loads each)
    narrow = [ln for ln in p.stdout.splitlines() if "<1, 0, 0, 0, 1, 0, 4, 3, 0, 1>" in ln]                # narrow head, split roles
    assert ring and "(8, -8)" in ring[0] and "(8, 8)" in marked(ring[0]), ring
    assert pf and "(9, 9)" in pf[0] and "(9, 9)" in marked(pf[0]), pf
    assert roles and "(12, 12)" in marked(roles[0]), roles
    assert narrow and "(" in marked(narrow[0]), narrow
