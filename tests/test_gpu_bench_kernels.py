"""Parity of the kernel instantiations the BENCHMARK runs (round-1 verdict, "What's weak" 1).

The dispatcher picks a kernel by tile count: conv3_f16x3p_kernel<MF, INACT> (persistent, warp-specialised) only for 3x3
stride-1 convs with >= 256 tiles, the RPW = 4 instantiations of the tiled kernels only for grids of >= 512
workgroups. The small shapes of test_gpu_ops.py never reach those, so every case here is sized to DISPATCH the
kernel under test (asserted through the op log) and compared with an fp64 reference of the same op:
max |err| of the f16x3 kernel <= 8x the exact-fp32 kernel's own error (the error budget of DESIGN.md section 9).
A second set pins "persistent == tiled, bit for bit" by running the same launch with the dispatch threshold
moved (lssvc_set_option)."""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def hip():
    from lssvc_amd import hip_ops
    return hip_ops


def _W(sd):
    from lssvc_amd.weights import WeightStore
    return WeightStore(sd, torch.device(DEV))


def nhwc(hip, x):
    return hip.T.from_nchw(x.to(DEV))


def back(t):
    return t.to_nchw().cpu()


def _set(name, value):
    from lssvc_amd._lib import lib, check
    check(lib.lssvc_set_option(name.encode(), value))


def _get(name):
    from lssvc_amd._lib import lib, check
    v = C.c_int32()
    check(lib.lssvc_get_option(name.encode(), C.byref(v)))
    return v.value


def _run(hip, mode, fn):
    """fn() under conv precision `mode` with the op log on -> (result, kernel name of the last conv launch)."""
    old = hip.CONV_PRECISION
    try:
        hip.set_conv_precision(mode)
        hip.OP_LOG = []
        out = fn()
        return out, hip.OP_LOG[-1]["kernel"]
    finally:
        hip.OP_LOG = None
        hip.set_conv_precision(old)          # (round 5: this used to LEAVE the process in "f32", and every test file after this one ran in that mode)


# cins, cout, H, W, in_act, act, residual, pixel_shuffle, expected persistent instantiation
P3_CASES = [
    ([64], 64, 300, 340, None, None, False, False, 4),            # the dominant kernel of the bench
    ([64], 64, 293, 331, "lrelu", "lrelu", True, False, 4),       # ResBlock.conv2 form, H % 24 != 0, W % 16 != 0
    ([48], 48, 300, 340, "lrelu", "lrelu", False, False, 3),      # ResBlock.conv1 @ full resolution
    ([48], 48, 301, 333, None, None, True, False, 3),
    ([96], 48, 300, 340, None, None, False, False, 3),            # context_fusion conv1_out (two 48-ch inputs below)
    ([48, 48], 48, 290, 350, None, None, False, False, 3),
    ([64, 16], 48, 300, 340, None, None, False, False, 3),        # recon first_conv: 80 -> 48, two-input concat
    ([128], 192, 170, 180, None, None, False, False, 4),          # 3 M tiles
    ([128], 64, 300, 340, "lrelu", None, False, False, 4),
    ([128], 256, 150, 170, None, "lrelu", False, True, 4),        # subpel: pixel-shuffle epilogue (conv2_up.0 form)
    ([96], 256, 130, 170, None, None, False, True, 4),
    ([32], 32, 300, 340, None, "relu", False, False, 2),
    ([16], 16, 300, 340, None, None, True, False, 1),
    ([192], 96, 200, 260, None, "lrelu", False, False, 3),        # res_encoder.res2.conv1 form
]


@pytest.mark.parametrize("cins,cout,H,W,in_act,act,residual,shuffle,mf", P3_CASES)
def test_persistent_3x3_matches_fp64(hip, cins, cout, H, W, in_act, act, residual, shuffle, mf):
    g = torch.Generator().manual_seed(hash((tuple(cins), cout, H, W)) & 0xFFFF)
    xs = [torch.randn(1, c, H, W, generator=g) for c in cins]
    cin = sum(cins)
    w = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
    b = torch.randn(cout, generator=g)
    r = torch.randn(1, cout, H, W, generator=g) if residual else None
    x64 = torch.cat(xs, 1).double()
    if in_act == "lrelu":
        x64 = F.leaky_relu(x64, 0.1)
    ref = F.conv2d(x64, w.double(), b.double(), padding=1)
    if shuffle:
        ref = F.pixel_shuffle(ref, 2)
    if act == "lrelu":
        ref = F.leaky_relu(ref, 0.01)
    elif act == "relu":
        ref = F.relu(ref)
    if residual:
        ref = ref + r.double()
    name = "s.0" if shuffle else "c"
    Wt = _W({name + ".weight": w, name + ".bias": b})

    def launch():
        kw = dict(in_act=in_act, in_slope=0.1, act=act, slope=0.01, residual=nhwc(hip, r) if residual else None)
        ins = [nhwc(hip, x) for x in xs]
        return back(hip.subpel(Wt, "s", ins, **kw) if shuffle else hip.conv(Wt, "c", ins, **kw))

    got16, k16 = _run(hip, "f16x3", launch)
    got32, k32 = _run(hip, "f32", launch)
    inact = "true" if in_act else "false"
    want = "conv3_f16x3p_kernel<%d, %s>" % (mf, inact)
    if mf == 1:
        want = "conv3n_f16x3p_kernel<%s, fast> roles" % inact           # round 6: <= 16 output channels take the narrow-head instantiation
    assert k16 == want, k16                                           # really a persistent kernel
    assert k32.startswith("conv_mfma_kernel"), k32
    assert got16.shape == ref.shape
    e16 = (got16.double() - ref).abs().max().item()
    e32 = (got32.double() - ref).abs().max().item()
    assert e16 <= 8 * e32 + 1e-6, (e16, e32)
    # and a second, kernel-independent bar: fp32-class absolute accuracy
    assert e16 <= 2e-5 * max(1.0, ref.abs().max().item()), e16


@pytest.mark.parametrize("cins,cout,H,W,in_act,act,residual,shuffle,mf", P3_CASES[:2] + P3_CASES[4:7] + P3_CASES[9:10])
def test_persistent_3x3_is_bit_identical_to_tiled(hip, cins, cout, H, W, in_act, act, residual, shuffle, mf):
    """DESIGN section 10 claims the persistent kernel reproduces the tiled kernel bit for bit (same K order inside a
    step, same accumulator layout, same epilogue). Pinned here by running one launch under both dispatch decisions."""
    g = torch.Generator().manual_seed(hash((tuple(cins), cout, H)) & 0xFFFF)
    xs = [torch.randn(1, c, H, W, generator=g) for c in cins]
    cin = sum(cins)
    w = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
    b = torch.randn(cout, generator=g)
    r = torch.randn(1, cout, H, W, generator=g) if residual else None
    name = "s.0" if shuffle else "c"
    Wt = _W({name + ".weight": w, name + ".bias": b})

    def launch():
        kw = dict(in_act=in_act, in_slope=0.1, act=act, slope=0.01, residual=nhwc(hip, r) if residual else None)
        ins = [nhwc(hip, x) for x in xs]
        return back(hip.subpel(Wt, "s", ins, **kw) if shuffle else hip.conv(Wt, "c", ins, **kw))

    old = _get("f16x3_persist")
    try:
        _set("f16x3_persist", 1)
        c, kc = _run(hip, "f16x3", launch)                    # 24x16 tiles, epilogue at the tile boundary (conv3_f16x3p.hip, default)
        _set("f16x3_persist", 0)
        b_, kb = _run(hip, "f16x3", launch)                   # tiled kernel
    finally:
        _set("f16x3_persist", old)
    assert kc.startswith("conv3_f16x3p_kernel") and kb.startswith("conv_f16x3_kernel"), (kc, kb)
    assert torch.equal(c, b_)


@pytest.mark.parametrize("cins,cout,H,W,in_act,act,residual,shuffle,mf", [P3_CASES[1], P3_CASES[5], P3_CASES[7], ([32], 32, 300, 340, None, None, False, False, 2)])
@pytest.mark.parametrize("stride", [1, 2])
def test_persistent_3x3_pre_split_inputs_are_bit_identical(hip, cins, cout, H, W, in_act, act, residual, shuffle, mf, stride):
    """Round 5: the persistent 3x3 kernels fed PRE-SPLIT inputs (lssvc_presplit: fp16 hi | lo per 16-channel
    chunk, input activation applied; the halo patch then goes global -> LDS by LDS-DMA, zero padding from a block of zeros) against
    the same conv on the fp32 tensors: the split values are what the kernel makes of the fp32 input itself, so the outputs are equal
    bit for bit -- partial tiles on both edges, two-input concat, 3 M tiles, a residual, stride 2 with its de-interleaved patch
    columns. (The A/B that decided NOT to convert the pipeline to this format: profiles/r05_p3_split_ab.txt.)"""
    if stride == 2 and (mf < 3 or residual):
        pytest.skip("the stride-2 persistent kernel serves >= 48 output channels; the residual case is a stride-1 shape")
    g = torch.Generator().manual_seed(hash((tuple(cins), cout, H, stride)) & 0xFFFF)
    xs = [torch.randn(1, c, H, W, generator=g) * 2.0 for c in cins]
    cin = sum(cins)
    w = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
    b = torch.randn(cout, generator=g)
    Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    r = torch.randn(1, cout, Ho, Wo, generator=g) if residual else None
    Wt = _W({"c.weight": w, "c.bias": b})
    old = _get("f16x3_persist_min_tiles")
    try:
        _set("f16x3_persist_min_tiles", 1)
        ins = [nhwc(hip, x) for x in xs]
        res = nhwc(hip, r) if residual else None

        def plain():
            return back(hip.conv(Wt, "c", ins, stride=stride, in_act=in_act, in_slope=0.1, act=act, slope=0.01, residual=res))

        def split():
            sp = [hip.presplit(t, in_act, 0.1) for t in ins]
            return back(hip.conv(Wt, "c", sp, stride=stride, act=act, slope=0.01, residual=res))

        a, ka = _run(hip, "f16x3", plain)
        b_, kb = _run(hip, "f16x3", split)
    finally:
        _set("f16x3_persist_min_tiles", old)
    assert ka.startswith("conv3" ) and "f16x3p_kernel" in ka and kb.endswith(" split"), (ka, kb)
    assert torch.equal(a, b_)


@pytest.mark.slow            # (an experimental kernel path that is off by default: behind --runslow since round 5)
@pytest.mark.parametrize("cins,cout,H,W,in_act,act,residual,shuffle,mf", [c for c in P3_CASES if c[8] >= 3])
def test_persistent_3x3_staged_epilogue_is_bit_identical(hip, cins, cout, H, W, in_act, act, residual, shuffle, mf):
    """The staged epilogue (conv3_f16x3p.hip, STAGE: the consumers park the finished tile in the operand buffers they have
    just consumed and the producer waves store it) against the same kernel with the epilogue in the consumer waves (option
    p3_stage = 0), on every MF >= 3 shape of the list: partial tiles at the right and bottom edges, 3 M tiles, two- and
    three-phase tiles (the parking area alternates between the two buffer pairs when a tile has an odd number of phases),
    residual, pixel-shuffle store. Twice each, so that a hand-off that only sometimes loses a row shows up."""
    g = torch.Generator().manual_seed(hash((tuple(cins), cout, W)) & 0xFFFF)
    xs = [torch.randn(1, c, H, W, generator=g) for c in cins]
    cin = sum(cins)
    w = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
    b = torch.randn(cout, generator=g)
    r = torch.randn(1, cout, H, W, generator=g) if residual else None
    name = "s.0" if shuffle else "c"
    Wt = _W({name + ".weight": w, name + ".bias": b})

    def launch():
        kw = dict(in_act=in_act, in_slope=0.1, act=act, slope=0.01, residual=nhwc(hip, r) if residual else None)
        ins = [nhwc(hip, x) for x in xs]
        return back(hip.subpel(Wt, "s", ins, **kw) if shuffle else hip.conv(Wt, "c", ins, **kw))

    old = _get("p3_stage")
    try:
        _set("p3_stage", 0)
        plain, k0 = _run(hip, "f16x3", launch)
        _set("p3_stage", 1)
        staged = [_run(hip, "f16x3", launch)[0] for _ in range(2)]
    finally:
        _set("p3_stage", old)
    assert k0.startswith("conv3_f16x3p_kernel")
    for s in staged:
        assert torch.equal(s, plain), (s - plain).abs().max().item()


# cins, cout, H_in, W_in, in_act, act, residual  (stride-2 3x3, padding 1: out = ceil(in / 2))
P3S2_CASES = [
    ([48], 64, 360, 420, None, None, False),             # the bench's 48 -> 64 @1152x1920 -> 576x960 form (>= 256 tiles of 8x16 outputs)
    ([64], 64, 373, 411, "lrelu", "lrelu", False),       # odd input sizes: partial tiles on both edges
    ([4], 64, 360, 420, None, None, False),              # RGB input padded to 4 channels: one phase per tile
    ([48, 8], 64, 380, 400, None, "lrelu", False),       # two-input concat, 56 channels (the bench's 52 / 56 -> 64 forms)
    ([64], 96, 260, 300, None, "lrelu", False),          # MF = 3, two M tiles
    ([128], 96, 260, 310, "lrelu", None, True),          # MF = 3, 8 phases, with a residual
]


@pytest.mark.parametrize("cins,cout,H,W,in_act,act,residual", P3S2_CASES)
def test_persistent_3x3_stride2(hip, cins, cout, H, W, in_act, act, residual):
    """The stride-2 form of the persistent kernel (8x16-pixel output tiles, patch columns de-interleaved): fp32-class accuracy
    against fp64, bit-identical to the tiled stride-2 kernel (same K order, same epilogue), and really dispatched."""
    g = torch.Generator().manual_seed(hash((tuple(cins), cout, H, 2)) & 0xFFFF)
    xs = [torch.randn(1, c, H, W, generator=g) for c in cins]
    cin = sum(cins)
    w = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
    b = torch.randn(cout, generator=g)
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    r = torch.randn(1, cout, Ho, Wo, generator=g) if residual else None
    x64 = torch.cat(xs, 1).double()
    if in_act == "lrelu":
        x64 = F.leaky_relu(x64, 0.1)
    ref = F.conv2d(x64, w.double(), b.double(), stride=2, padding=1)
    if act == "lrelu":
        ref = F.leaky_relu(ref, 0.01)
    if residual:
        ref = ref + r.double()
    Wt = _W({"c.weight": w, "c.bias": b})

    def launch():
        return back(hip.conv(Wt, "c", [nhwc(hip, x) for x in xs], stride=2, in_act=in_act, in_slope=0.1, act=act, slope=0.01,
                             residual=nhwc(hip, r) if residual else None))

    old = _get("f16x3_persist_s2")
    try:
        _set("f16x3_persist_s2", 1)
        got, k1 = _run(hip, "f16x3", launch)
        again, _ = _run(hip, "f16x3", launch)
        _set("f16x3_persist_s2", 0)
        tiled, k0 = _run(hip, "f16x3", launch)
    finally:
        _set("f16x3_persist_s2", old)
    got32, _ = _run(hip, "f32", launch)
    assert k1.startswith("conv3s2_f16x3p_kernel<%d" % (4 if cout % 64 == 0 else 3)), k1
    assert k0.startswith("conv_f16x3_kernel"), k0
    assert got.shape == ref.shape
    assert torch.equal(got, tiled) and torch.equal(got, again)
    e16 = (got.double() - ref).abs().max().item()
    e32 = (got32.double() - ref).abs().max().item()
    assert e16 <= 8 * e32 + 1e-6, (e16, e32)


def test_persistent_3x3_small_grids(hip):
    """Grids with fewer than 8 workgroups (tile ranges keyed by blockIdx & 7 left tiles uncomputed): force
    the persistent kernel onto convs with 1..7 tiles and compare with the tiled kernel."""
    old_min, old_on = _get("f16x3_persist_min_tiles"), _get("f16x3_persist")
    old_narrow = _get("p3_narrow")
    try:
        _set("p3_narrow", 0)                              # (round 6: the 16-channel case would take the narrow-head instantiation; this test is about the 24x16 one)
        # 1..7 tiles (24x16) / a few more 16x16 ones, partial tiles, several M tiles, a lone last tile to flush
        for H, W, cout in ((24, 16, 64), (24, 48, 64), (48, 48, 64), (30, 70, 48), (24, 16, 128), (50, 20, 16), (72, 16, 64),
                           (100, 40, 192)):
            g = torch.Generator().manual_seed(H * W + cout)
            x = torch.randn(1, 64, H, W, generator=g)
            w = torch.randn(cout, 64, 3, 3, generator=g) / 24
            b = torch.randn(cout, generator=g)
            Wt = _W({"c.weight": w, "c.bias": b})
            _set("f16x3_persist_min_tiles", 1)
            _set("f16x3_persist", 0)
            b_, kb = _run(hip, "f16x3", lambda: back(hip.conv(Wt, "c", nhwc(hip, x))))
            assert kb.startswith("conv_f16x3_kernel"), kb
            _set("f16x3_persist", 1)
            a, ka = _run(hip, "f16x3", lambda: back(hip.conv(Wt, "c", nhwc(hip, x))))
            assert ka.startswith("conv3_f16x3p_kernel"), ka
            assert torch.equal(a, b_), (H, W, cout)
    finally:
        _set("f16x3_persist_min_tiles", old_min)
        _set("f16x3_persist", old_on)
        _set("p3_narrow", old_narrow)


# ---- RPW = 4 instantiations of the tiled f16x3 kernels (7x7 SpyNet convs; 3x3 with 2-3 output channels) -----------
TILED_CASES = [
    # cins, cout, k, H, W, act, residual, expected kernel
    ([8], 32, 7, 384, 400, "relu", False, "conv_f16x3_kernel<2, 4, 7, 1>"),       # moduleBasic.conv1
    ([32], 64, 7, 260, 520, "relu", False, "conv_f16x3_kernel<4, 4, 7, 1>"),      # conv2
    ([64], 32, 7, 260, 520, "relu", False, "conv_f16x3_kernel<2, 4, 7, 1>"),      # conv3 (2.4 % of a P-frame each)
    ([32], 16, 7, 384, 400, "relu", False, "conv_f16x3_kernel<1, 4, 7, 1>"),      # conv4
    ([16], 2, 7, 384, 400, None, True, "conv_f16x3_kernel<1, 4, 7, 1>"),          # conv5 + flow residual
    ([64], 2, 3, 384, 400, None, False, "conv3n_f16x3p_kernel<false, flat> roles"),  # mv_resampler.recon_conv / weight maps (round 6: the narrow-head persistent kernel,
    ([48], 3, 3, 384, 400, None, False, "conv3n_f16x3p_kernel<false, flat> roles"),  # recon_conv                         test_narrow_heads_* hold it to the tiled kernel bit for bit)
]


@pytest.mark.parametrize("persist7", [1, 0])
@pytest.mark.parametrize("cins,cout,k,H,W,act,residual,kernel", TILED_CASES)
def test_tiled_rpw4_matches_fp64(hip, cins, cout, k, H, W, act, residual, kernel, persist7):
    """persist7 = 1 (default): the 7x7 convs with a fused-epilogue-friendly output (Cout % 4 == 0) run on the persistent
    warp-specialised 7x7 kernel (conv7_f16x3p.hip; Cout >= 32); persist7 = 0 and the narrow-output convs: the RPW = 4 tiled kernels."""
    g = torch.Generator().manual_seed(hash((tuple(cins), cout, k, H)) & 0xFFFF)
    xs = [torch.randn(1, c, H, W, generator=g) for c in cins]
    cin = sum(cins)
    w = torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)
    b = torch.randn(cout, generator=g)
    r = torch.randn(1, cout, H, W, generator=g) if residual else None
    ref = F.conv2d(torch.cat(xs, 1).double(), w.double(), b.double(), padding=k // 2)
    if act == "relu":
        ref = F.relu(ref)
    if residual:
        ref = ref + r.double()
    Wt = _W({"c.weight": w, "c.bias": b})

    def launch():
        return back(hip.conv(Wt, "c", [nhwc(hip, x) for x in xs], act=act, residual=nhwc(hip, r) if residual else None))

    old = _get("f16x3_persist7")
    try:
        _set("f16x3_persist7", persist7)
        got16, k16 = _run(hip, "f16x3", launch)
        if persist7 and k == 7 and cout % 4 == 0 and cout >= 32:                      # bit-identical to the tiled kernel
            _set("f16x3_persist7", 0)
            tiled, kt = _run(hip, "f16x3", launch)
            assert kt == kernel and torch.equal(got16, tiled), (k16, kt)
    finally:
        _set("f16x3_persist7", old)
    got32, _ = _run(hip, "f32", launch)
    if persist7 and k == 7 and cout % 4 == 0 and cout >= 32:
        assert k16 == "conv7_f16x3p_kernel<%d, false>" % min(4, (cout + 15) // 16), k16
    else:
        assert k16 == kernel, k16
    e16 = (got16.double() - ref).abs().max().item()
    e32 = (got32.double() - ref).abs().max().item()
    assert e16 <= 8 * e32 + 1e-6, (e16, e32)
    assert e16 <= 2e-5 * max(1.0, ref.abs().max().item()), e16


P7_EXTRA = [
    # cins, cout, H, W: several chunks per tile, two-input concat, partial tiles in both directions, few tiles (forced)
    ([64], 32, 301, 333), ([32, 32], 64, 290, 350), ([48], 48, 300, 340), ([128], 32, 262, 270), ([16], 64, 50, 40),
]


@pytest.mark.parametrize("cins,cout,H,W", P7_EXTRA)
def test_persistent_7x7_is_bit_identical_to_tiled(hip, cins, cout, H, W):
    g = torch.Generator().manual_seed(hash((tuple(cins), cout, H)) & 0xFFFF)
    xs = [torch.randn(1, c, H, W, generator=g) for c in cins]
    cin = sum(cins)
    w = torch.randn(cout, cin, 7, 7, generator=g) / math.sqrt(cin * 49)
    b = torch.randn(cout, generator=g)
    r = torch.randn(1, cout, H, W, generator=g)
    Wt = _W({"c.weight": w, "c.bias": b})

    def launch():
        return back(hip.conv(Wt, "c", [nhwc(hip, x) for x in xs], in_act="lrelu", in_slope=0.2, act="lrelu", slope=0.1, residual=nhwc(hip, r)))

    old, old_min = _get("f16x3_persist7"), _get("f16x3_persist_min_tiles")
    try:
        _set("f16x3_persist_min_tiles", 1)
        _set("f16x3_persist7", 1)
        a, ka = _run(hip, "f16x3", launch)
        _set("f16x3_persist7", 0)
        b_, kb = _run(hip, "f16x3", launch)
    finally:
        _set("f16x3_persist7", old)
        _set("f16x3_persist_min_tiles", old_min)
    assert ka.startswith("conv7_f16x3p_kernel") and kb.startswith("conv_f16x3_kernel"), (ka, kb)
    assert torch.equal(a, b_)
    ref = F.leaky_relu(F.conv2d(F.leaky_relu(torch.cat(xs, 1).double(), 0.2), w.double(), b.double(), padding=3), 0.1) + r.double()
    assert (a.double() - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("cin,cout,H,W", [(64, 64, 576, 960), (48, 48, 576, 960), (64, 64, 290, 350), (64, 64, 1152, 1920)])
def test_persistent_3x3_hand_off_is_race_free(hip, cin, cout, H, W):
    """The persistent kernel's producer and consumer waves hand LDS buffers over through per-wave slots, not barriers: the
    consumers run ahead of each other and of the producers by up to a phase. Forty launches with an input activation (the
    longest staging path; a shared hand-off counter once let a consumer read the last patch rows before they were written,
    nine launches out of ten) must all equal the tiled kernel bit for bit."""
    g = torch.Generator().manual_seed(cin + H)
    x = torch.randn(1, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
    b = torch.randn(cout, generator=g)
    Wt = _W({"c.weight": w, "c.bias": b})
    xin = nhwc(hip, x)

    def launch():
        return hip.conv(Wt, "c", [xin], in_act="lrelu", in_slope=0.1, act="lrelu", slope=0.01)

    from lssvc_amd._lib import lib
    old_on, old_min = _get("f16x3_persist"), _get("f16x3_persist_min_tiles")
    try:
        hip.set_conv_precision("f16x3")
        _set("f16x3_persist", 0)
        ref = launch().buf.clone()
        _set("f16x3_persist", 1)
        _set("f16x3_persist_min_tiles", 1)
        bad = 0
        for _ in range(40):
            out = launch()
            assert lib.lssvc_conv2d_last_kernel().decode().startswith("conv3_f16x3p_kernel<%d, true>" % min(4, cout // 16))
            bad += 0 if torch.equal(out.buf, ref) else 1
    finally:
        hip.set_conv_precision("f32")
        _set("f16x3_persist", old_on)
        _set("f16x3_persist_min_tiles", old_min)
    assert bad == 0, "%d of 40 launches differ" % bad


def test_persistent_7x7_hand_off_is_race_free(hip):
    """Same hand-off scheme in the 7x7 kernel (patch written in seven slices, one per phase): repeated launches = tiled kernel."""
    from lssvc_amd._lib import lib
    g = torch.Generator().manual_seed(77)
    x = torch.randn(1, 32, 576, 960, generator=g)
    w = torch.randn(64, 32, 7, 7, generator=g) / math.sqrt(32 * 49)
    b = torch.randn(64, generator=g)
    Wt = _W({"c.weight": w, "c.bias": b})
    xin = nhwc(hip, x)

    def launch():
        return hip.conv(Wt, "c", [xin], in_act="lrelu", in_slope=0.1, act="relu")

    old = _get("f16x3_persist7")
    try:
        hip.set_conv_precision("f16x3")
        _set("f16x3_persist7", 0)
        ref = launch().buf.clone()
        assert lib.lssvc_conv2d_last_kernel().decode().startswith("conv_f16x3_kernel")
        _set("f16x3_persist7", 1)
        bad = 0
        for _ in range(30):
            out = launch()
            assert lib.lssvc_conv2d_last_kernel().decode() == "conv7_f16x3p_kernel<4, true>"
            bad += 0 if torch.equal(out.buf, ref) else 1
    finally:
        hip.set_conv_precision("f32")
        _set("f16x3_persist7", old)
    assert bad == 0, "%d of 30 launches differ" % bad


# ---- round 6: the small-tile, narrow-head and prefetching instantiations of the persistent 3x3 kernel, the lean GDN epilogue ----------
def _opts(**kw):
    """Set library options for the duration of a with-block."""
    import contextlib

    @contextlib.contextmanager
    def cm():
        old = {k: _get(k) for k in kw}
        try:
            for k, v in kw.items():
                _set(k, v)
            yield
        finally:
            for k, v in old.items():
                _set(k, v)
    return cm()


def _conv_case(hip, cins, cout, H, W, stride=1, seed=0, **kw):
    g = torch.Generator().manual_seed(seed + hash((tuple(cins), cout, H, W)) & 0xFFFF)
    xs = [torch.randn(1, c, H, W, generator=g) for c in cins]
    cin = sum(cins)
    w = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
    b = torch.randn(cout, generator=g)
    Ho, Wo = (H + stride - 1) // stride, (W + stride - 1) // stride
    shuffle = kw.pop("pixel_shuffle", False)
    r = torch.randn(1, cout, Ho, Wo, generator=g) if kw.pop("residual", False) else None
    name = "s.0" if shuffle else "c"
    Wt = _W({name + ".weight": w, name + ".bias": b})

    def launch():
        ins = [nhwc(hip, x) for x in xs]
        k2 = dict(kw, residual=nhwc(hip, r) if r is not None else None)
        return back(hip.subpel(Wt, "s", ins, **k2) if shuffle else hip.conv(Wt, "c", ins, stride=stride, **k2))
    return launch


SMALL_MAP_CASES = [
    # cins, cout, H, W, kwargs: the maps of the quarter-resolution / prior / hyper networks at 1080p (< 256 tiles of 24x16) and odd sizes
    ([64], 64, 144, 240, {}),
    ([64], 64, 141, 233, {"in_act": "lrelu", "in_slope": 0.1, "act": "lrelu", "residual": True}),      # partial tiles on both edges
    ([128], 128, 72, 120, {"act": "lrelu"}),                                                       # 8 phases: the register prefetch
    ([96], 96, 36, 60, {}),                                                                        # MF = 3
    ([64, 64], 128, 70, 118, {"in_act": "lrelu", "in_slope": 0.1}),                                 # two-input concat
    ([64], 256, 72, 120, {"pixel_shuffle": True}),                                                 # subpel store, 4 M tiles
    ([192], 160, 36, 60, {"act": "relu"}),                                                         # ragged M (10 fragments)
]


@pytest.mark.parametrize("cins,cout,H,W,kw", SMALL_MAP_CASES)
def test_small_tilings_are_bit_identical_to_tiled(hip, cins, cout, H, W, kw):
    """Round 6: 3x3 convs on maps the 24x16 tiling cannot spread over 256 CUs run on the persistent kernel's
    small-tile instantiations (16x16 / 8x16 / 4x16 pixels, MF or MF / 2 channel fragments per tile; conv3_f16x3p.hip: p3_pick_tiling).
    Same MFMA sequence per accumulator, same epilogue: every forced (MF, rows per wave) pair, with and without the register
    prefetch, and the dispatcher's own choice must equal the tiled kernel bit for bit."""
    launch = _conv_case(hip, cins, cout, H, W, **dict(kw))
    frags = (cout + 15) // 16
    with _opts(p3_small=0, p3_force=0):
        tiled, k0 = _run(hip, "f16x3", launch)
    assert k0.startswith("conv_f16x3_kernel"), k0
    with _opts(p3_small=1, p3_force=0):
        auto, ka = _run(hip, "f16x3", launch)
    assert ka.startswith("conv3r_f16x3p_kernel<"), ka
    assert torch.equal(auto, tiled), ka
    seen = set()
    for mf in (4, 3, 2, 1):
        if mf > frags:
            continue
        for rpw in (4, 2, 1):
            for sm in (2, 3):                                    # register prefetch always / never
                with _opts(p3_small=sm, p3_force=mf * 16 + rpw):
                    got, k = _run(hip, "f16x3", launch)
                assert k.startswith("conv3r_f16x3p_kernel<%d," % mf) and ("rpw %d" % rpw) in k and (("pf2" in k) == (sm == 2)), k
                assert torch.equal(got, tiled), k
                seen.add(k)
    assert len(seen) >= 6


NARROW_CASES = [
    # cins, cout, H, W, kwargs, epilogue
    ([64], 2, 384, 400, {}, "flat"),                                         # mv_resampler.recon_conv / weight maps
    ([64], 2, 371, 393, {"residual": True}, "flat"),                         # a 2-channel residual (scalar loads), partial tiles
    ([48], 3, 384, 400, {"in_act": "lrelu", "in_slope": 0.1}, "flat"),       # recon_conv form with an input activation
    ([64], 8, 300, 340, {"act": "lrelu"}, "fast"),
    ([32, 32], 16, 300, 340, {"act": "relu", "residual": True}, "fast"),     # two inputs, 16 channels, float4 residual
    ([16], 12, 290, 330, {}, "fast"),                                        # Cout % 4 == 0 but not a whole fragment
    ([48], 3, 301, 333, {"act": "relu", "residual": True, "out_scale": 0.5}, "flat"),        # every branch of the straight-line flat epilogue (conv_epilogue_plain)
    ([64], 2, 290, 350, {"act": "lrelu", "residual": True, "out_scale": 20.0}, "flat"),      # a flow head's form: activation, residual flow, scale
    ([64], 6, 300, 340, {"in_act": "lrelu", "in_slope": 0.1, "residual": True}, "flat"),     # two lane groups, the second with two of its four channels
]


@pytest.mark.parametrize("cins,cout,H,W,kw,epi", NARROW_CASES)
def test_narrow_heads_are_bit_identical_to_round5_kernels(hip, cins, cout, H, W, kw, epi):
    """Round 6: 3x3 convs with at most 16 output channels -- flow heads, picture heads -- on the persistent kernel's
    narrow instantiation (MF = 1, 16x16 tiles, two workgroups per CU, register prefetch; the tiled kernel's own epilogue when Cout % 4
    != 0): equal, bit for bit, to what round 5 ran (the tiled kernel, or the 24x16 persistent one for Cout % 4 == 0), with and
    without the prefetch, and twice in a row."""
    launch = _conv_case(hip, cins, cout, H, W, **dict(kw))
    inact = "true" if kw.get("in_act") else "false"
    with _opts(p3_narrow=0):
        old, k0 = _run(hip, "f16x3", launch)
    assert k0.startswith("conv_f16x3_kernel<1,") or k0.startswith("conv3_f16x3p_kernel<1,"), k0
    with _opts(p3_narrow=0, f16x3_persist=0):
        tiled, kt = _run(hip, "f16x3", launch)
    assert kt.startswith("conv_f16x3_kernel<1,"), kt
    with _opts(p3_narrow=1, p3_pf2=1):
        new, k1 = _run(hip, "f16x3", launch)
        again, _ = _run(hip, "f16x3", launch)
    assert k1 == "conv3n_f16x3p_kernel<%s, %s> roles" % (inact, epi), k1
    with _opts(p3_narrow=1, p3_pf2=4):
        pf2, k3 = _run(hip, "f16x3", launch)
    assert k3 == "conv3n_f16x3p_kernel<%s, %s> pf2" % (inact, epi), k3
    with _opts(p3_narrow=1, p3_pf2=0):
        plain, k2 = _run(hip, "f16x3", launch)
    assert k2 == "conv3n_f16x3p_kernel<%s, %s>" % (inact, epi), k2
    assert torch.equal(new, old) and torch.equal(new, tiled) and torch.equal(new, again) and torch.equal(plain, old) and torch.equal(pf2, old)


@pytest.mark.parametrize("cins,cout,H,W,in_act,act,residual", P3S2_CASES)
def test_stride2_register_prefetch_is_bit_identical(hip, cins, cout, H, W, in_act, act, residual):
    """Round 6: the stride-2 persistent kernel's producer schedules -- split roles (3: one wave owns the weight DMA,
    three stage the patch through two register sets), register prefetch (4), pair loads (2; measured slower, kept for the A/B), round 5's
    one register set (0) and the default (1: roles up to five phases per tile, the prefetch from six on): one arithmetic, four schedules."""
    launch = _conv_case(hip, cins, cout, H, W, stride=2, in_act=in_act, in_slope=0.1, act=act, slope=0.01, residual=residual)
    outs = {}
    phases = sum((c + 15) // 16 for c in cins)
    for pf in (0, 1, 2, 3, 4):
        with _opts(p3_pf2=pf):
            outs[pf], k = _run(hip, "f16x3", launch)
            again, _ = _run(hip, "f16x3", launch)
        assert k.startswith("conv3s2_f16x3p_kernel<") and k.endswith({0: ">", 1: "roles" if phases <= 5 else "pf2", 2: "pair", 3: "roles", 4: "pf2"}[pf]), k
        assert torch.equal(outs[pf], again), k
    assert all(torch.equal(outs[0], outs[pf]) for pf in (1, 2, 3, 4))


@pytest.mark.parametrize("c,H,W", [(64, 60, 100), (96, 37, 53), (128, 48, 80), (192, 20, 31), (48, 64, 64)])
@pytest.mark.parametrize("flavour,inverse", [("intra", False), ("intra", True), ("inter", False), ("inter", True)])
@pytest.mark.parametrize("residual", [False, True])
def test_gdn_lean_epilogue_is_bit_identical(hip, c, H, W, flavour, inverse, residual):
    """Round 6 (the part that paid): the GDN / IGDN 1x1 kernels' normalising epilogue as straight-line code
    (conv_mfma_kernel.h: conv_epilogue_gdn -- the kind of normalisation a compile-time parameter, float4 accesses only) against the
    general epilogue it replaces (option gdn_fast = 0): same operations in the same order, equal bit for bit, in both conv
    precisions (the exact-fp32 kernel shares the routine)."""
    from lssvc_amd.synth import _make
    sd = {"g.beta": _make({"key": "g.beta", "shape": [c], "kind": "gdn_beta"}, 3, 1.0),
          "g.gamma": _make({"key": "g.gamma", "shape": [c, c], "kind": "gdn_gamma"}, 3, 1.0),
          "g.beta_reparam.pedestal": torch.tensor([2.0 ** -36]), "g.gamma_reparam.pedestal": torch.tensor([2.0 ** -36]),
          "g.beta_reparam.lower_bound.bound": torch.tensor([(1e-6 + 2.0 ** -36) ** 0.5]),
          "g.gamma_reparam.lower_bound.bound": torch.tensor([2.0 ** -18])}
    Wt = _W(sd)
    g = torch.Generator().manual_seed(c * H + W)
    x = torch.randn(1, c, H, W, generator=g) * 2
    r = torch.randn(1, c, H, W, generator=g) if residual else None

    def launch():
        return back(hip.gdn(Wt, "g", nhwc(hip, x), flavour, inverse=inverse, residual=nhwc(hip, r) if residual else None))

    for mode in ("f16x3", "f32"):
        with _opts(gdn_fast=0):
            general, k0 = _run(hip, mode, launch)
        with _opts(gdn_fast=1):
            lean, k1 = _run(hip, mode, launch)
        assert k0 == k1 and (("f16x3" in k0) == (mode == "f16x3")), (k0, k1)
        assert torch.equal(general, lean), (mode, k0, (general - lean).abs().max().item())


@pytest.mark.parametrize("cins,cout,H,W,stride,opts,kernel", [
    ([64], 2, 576, 960, 1, {}, "conv3n_f16x3p_kernel<true, flat> roles"),                                # narrow head, two workgroups per CU, split roles
    ([64], 2, 576, 960, 1, {"p3_pf2": 4}, "conv3n_f16x3p_kernel<true, flat> pf2"),                       # ... register prefetch
    ([64], 8, 288, 480, 1, {}, "conv3n_f16x3p_kernel<true, fast> roles"),                                # ... fast epilogue
    ([64], 64, 144, 240, 1, {}, "conv3r_f16x3p_kernel<4, true, rpw 4>"),                                 # small tiling, plain schedule
    ([128], 128, 72, 120, 1, {}, "conv3r_f16x3p_kernel<4, true, rpw 2, pf2>"),                           # small tiling, 8 phases: register prefetch
    ([48], 64, 576, 960, 2, {}, "conv3s2_f16x3p_kernel<4, true> roles"),                                 # stride 2, split roles (3 phases per tile: odd)
    ([64], 96, 288, 480, 2, {}, "conv3s2_f16x3p_kernel<3, true> roles"),                                 # ... MF = 3, 4 phases
    ([48], 64, 576, 960, 2, {"p3_pf2": 4}, "conv3s2_f16x3p_kernel<4, true> pf2"),                        # stride 2, register prefetch
    ([128], 96, 288, 480, 2, {}, "conv3s2_f16x3p_kernel<3, true> pf2"),                                  # ... the default from six phases on
    ([48], 48, 320, 400, 1, {"p3_force": 3 * 16 + 8}, "conv3r_f16x3p_kernel<3, true, rpw 8, roles>"),    # 32x16 tiles, split roles
    ([64], 64, 320, 400, 1, {"p3_big_pair": 2}, "conv3_f16x3p_kernel<4, true> roles"),                   # 24x16 tiles, split roles (experiment)
    ([64], 64, 320, 400, 1, {"p3_big_pair": 4}, "conv3_f16x3p_kernel<4, true> late"),                    # 24x16 tiles, late loads (experiment)
    ([96], 48, 320, 400, 1, {"p3_force": 3 * 16 + 8, "p3_big_pair": 4}, "conv3r_f16x3p_kernel<3, true, rpw 8, late>"),      # 32x16 tiles, late loads
    ([64], 64, 576, 960, 2, {"p3_pf2": 2}, "conv3s2_f16x3p_kernel<4, true> pair"),                       # stride 2, pair loads
])
def test_round6_schedules_hand_off_is_race_free(hip, cins, cout, H, W, stride, opts, kernel):
    """The producer schedules of round 6 (split roles and register prefetch with counted waits, pair loads, two workgroups per CU) publish LDS buffers
    through the same per-wave slots as the round-5 kernel: thirty launches each, with an input activation, must all equal the tiled
    kernel bit for bit (a hand-off that signalled a buffer before its patch had landed would show up as a launch that differs)."""
    from lssvc_amd._lib import lib
    launch = None
    g = torch.Generator().manual_seed(sum(cins) + H + stride)
    xs = [nhwc(hip, torch.randn(1, c, H, W, generator=g)) for c in cins]
    cin = sum(cins)
    Wt = _W({"c.weight": torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9), "c.bias": torch.randn(cout, generator=g)})

    def launch():
        return hip.conv(Wt, "c", xs, stride=stride, in_act="lrelu", in_slope=0.1, act="lrelu", slope=0.01)

    try:
        hip.set_conv_precision("f16x3")
        with _opts(f16x3_persist=0, f16x3_persist_s2=0):
            ref = launch().buf.clone()
            assert lib.lssvc_conv2d_last_kernel().decode().startswith("conv_f16x3_kernel")
        bad = 0
        with _opts(**opts):
            for _ in range(30):
                out = launch()
                assert lib.lssvc_conv2d_last_kernel().decode() == kernel, lib.lssvc_conv2d_last_kernel().decode()
                bad += 0 if torch.equal(out.buf, ref) else 1
    finally:
        hip.set_conv_precision("f32")
    assert bad == 0, "%d of 30 launches differ" % bad


@pytest.mark.parametrize("cins,H,W,kw", [([48], 300, 340, {}), ([48, 48], 290, 350, {"in_act": "lrelu", "in_slope": 0.1, "act": "lrelu", "residual": True}), ([64, 16], 301, 333, {}),
                                         ([96], 300, 340, {"act": "lrelu"})])
def test_tall_tiles_are_bit_identical(hip, cins, H, W, kw):
    """Round 6: the 48-channel layers of the full-resolution maps run on 32x16-pixel tiles (8 rows per consumer wave; conv3_f16x3p.hip:
    p3_pick_tiling) -- forced here onto maps small enough for a test, with and without the register prefetch, against the 24x16
    tiling and the tiled kernel."""
    launch = _conv_case(hip, cins, 48, H, W, **dict(kw))
    with _opts(p3_small=0, p3_force=0):
        r5, k5 = _run(hip, "f16x3", launch)
    assert k5.startswith("conv3_f16x3p_kernel<3,"), k5
    with _opts(f16x3_persist=0):
        tiled, kt = _run(hip, "f16x3", launch)
    assert kt.startswith("conv_f16x3_kernel"), kt
    for sm in (2, 3):
        with _opts(p3_small=sm, p3_force=3 * 16 + 8):
            got, k = _run(hip, "f16x3", launch)
        assert k.startswith("conv3r_f16x3p_kernel<3,") and "rpw 8" in k and (("pf2" in k) == (sm == 2)), k
        assert torch.equal(got, r5) and torch.equal(got, tiled), k
    for bp in (2, 4, 0):                            # split roles / late loads on the tall tiles: forced, and by the default rules
        with _opts(p3_force=3 * 16 + 8, p3_big_pair=bp):
            got, k = _run(hip, "f16x3", launch)
            again, _ = _run(hip, "f16x3", launch)
        phases = sum((c + 15) // 16 for c in cins)
        late = bp == 0 and phases == 6 and not kw.get("in_act")       # the default rule: late loads for six-phase tiles without an input activation
        want = "roles" if bp == 2 or (bp == 0 and phases == 3) else "late" if (bp == 4 or late) else ("pf2" if phases % 3 == 0 else "rpw 8>")
        assert k.startswith("conv3r_f16x3p_kernel<3,") and "rpw 8" in k and want in k, (k, want)
        assert torch.equal(got, r5) and torch.equal(got, again), k


def test_tall_tiles_are_dispatched_for_the_full_resolution_48_channel_layers(hip):
    launch = _conv_case(hip, [48], 48, 1152, 1920)
    got, k = _run(hip, "f16x3", launch)
    assert k == "conv3r_f16x3p_kernel<3, false, rpw 8, roles>", k
    with _opts(p3_big_pair=3):
        pf2, k3 = _run(hip, "f16x3", launch)
    assert k3 == "conv3r_f16x3p_kernel<3, false, rpw 8, pf2>" and torch.equal(got, pf2), k3
    with _opts(p3_small=0):
        r5, k5 = _run(hip, "f16x3", launch)
    assert k5 == "conv3_f16x3p_kernel<3, false>" and torch.equal(got, r5)
