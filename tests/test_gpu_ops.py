"""GPU parity of every HIP kernel against the CPU oracle's building blocks (fp32 torch CPU) on
seeded inputs, called through the C ABI (ctypes). Tolerances are stated per test: convolutions
differ from the CPU only in fp32 summation order; the pointwise kernels replay the reference's
ATen op order and are held to a few ulp."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def hip():
    from lssvc_amd import hip_ops
    return hip_ops


def nhwc(hip, x):
    return hip.T.from_nchw(x.to(DEV))


def back(t):
    return t.to_nchw().cpu()


class FakeW:
    """Minimal WeightStore over an ad-hoc dict."""

    def __new__(cls, sd):
        from lssvc_amd.weights import WeightStore
        return WeightStore(sd, torch.device(DEV))


def close(got, want, rtol=2e-5, atol=2e-5):
    scale = max(1.0, want.abs().max().item())
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=rtol, atol=atol * scale)


CONV_CASES = [
    # cins, cout, k, stride, H, W
    ([64], 64, 3, 1, 40, 56),
    ([48], 48, 3, 1, 33, 47),
    ([3], 64, 3, 1, 24, 40),
    ([3, 64], 64, 3, 2, 32, 48),
    ([64, 64], 48, 3, 1, 20, 36),
    ([8], 32, 7, 1, 24, 24),
    ([32], 64, 7, 1, 18, 30),
    ([16], 2, 7, 1, 16, 32),
    ([64], 64, 1, 1, 17, 19),
    ([128, 256], 384, 1, 1, 8, 8),
    ([192], 192, 3, 1, 9, 15),
    ([3], 192, 1, 2, 64, 64),
    ([170], 149, 3, 1, 8, 8),
    ([64], 3, 3, 1, 32, 32),
]


@pytest.mark.parametrize("cins,cout,k,stride,H,W", CONV_CASES)
def test_conv_plain(hip, cins, cout, k, stride, H, W):
    g = torch.Generator().manual_seed(hash((tuple(cins), cout, k, stride)) & 0xFFFF)
    xs = [torch.randn(1, c, H, W, generator=g) for c in cins]
    w = torch.randn(cout, sum(cins), k, k, generator=g) / math.sqrt(sum(cins) * k * k)
    b = torch.randn(cout, generator=g)
    pad = k // 2 if not (k == 1 and stride == 2) else 0
    want = F.conv2d(torch.cat(xs, 1), w, b, stride=stride, padding=pad)
    Wt = FakeW({"c.weight": w, "c.bias": b})
    got = back(hip.conv(Wt, "c", [nhwc(hip, x) for x in xs], stride=stride, pad=pad))
    close(got, want)


def test_conv_fused_epilogues(hip):
    g = torch.Generator().manual_seed(7)
    x = torch.randn(1, 64, 24, 40, generator=g)
    r = torch.randn(1, 48, 24, 40, generator=g)
    w = torch.randn(48, 64, 3, 3, generator=g) / 24
    b = torch.randn(48, generator=g)
    Wt = FakeW({"c.weight": w, "c.bias": b})
    want = F.leaky_relu(F.conv2d(F.leaky_relu(x, 0.1), w, b, padding=1), 0.1) + r
    got = back(hip.conv(Wt, "c", nhwc(hip, x), in_act="lrelu", in_slope=0.1, act="lrelu", slope=0.1,
                        residual=nhwc(hip, r)))
    close(got, want)
    want = 1.5 * F.relu(F.conv2d(x, w, b, padding=1))
    got = back(hip.conv(Wt, "c", nhwc(hip, x), act="relu", out_scale=1.5))
    close(got, want)


@pytest.mark.parametrize("cin,cps,k", [(64, 64, 3), (192, 3, 3), (128, 64, 1), (64, 2, 3), (96, 48, 3)])
def test_subpel(hip, cin, cps, k):
    g = torch.Generator().manual_seed(cin + cps)
    x = torch.randn(1, cin, 12, 20, generator=g)
    w = torch.randn(cps * 4, cin, k, k, generator=g) / math.sqrt(cin * k * k)
    b = torch.randn(cps * 4, generator=g)
    want = F.leaky_relu(F.pixel_shuffle(F.conv2d(x, w, b, padding=k // 2), 2), 0.01)
    Wt = FakeW({"s.0.weight": w, "s.0.bias": b})
    got = back(hip.subpel(Wt, "s", nhwc(hip, x), act="lrelu"))
    close(got, want)


@pytest.mark.parametrize("stride,cin,cout", [(2, 64, 128), (2, 128, 2), (1, 144, 192), (2, 96, 144)])
def test_conv_transpose(hip, stride, cin, cout):
    g = torch.Generator().manual_seed(stride * 100 + cout)
    x = torch.randn(1, cin, 9, 15, generator=g)
    w = torch.randn(cin, cout, 3, 3, generator=g) / math.sqrt(cin * 9 / stride ** 2)
    b = torch.randn(cout, generator=g)
    want = F.conv_transpose2d(x, w, b, stride=stride, padding=1, output_padding=stride - 1)
    Wt = FakeW({"t.weight": w, "t.bias": b})
    got = back(hip.conv_t(Wt, "t", nhwc(hip, x), stride))
    close(got, want)


@pytest.mark.parametrize("flavour,inverse", [("intra", False), ("intra", True), ("inter", False), ("inter", True)])
def test_gdn(hip, flavour, inverse):
    from lssvc_oracle.blocks import Params, gdn_intra, gdn_inter
    from lssvc_amd.synth import _make
    c = 64
    sd = {"g.beta": _make({"key": "g.beta", "shape": [c], "kind": "gdn_beta"}, 3, 1.0),
          "g.gamma": _make({"key": "g.gamma", "shape": [c, c], "kind": "gdn_gamma"}, 3, 1.0),
          "g.beta_reparam.pedestal": torch.tensor([2.0 ** -36]), "g.gamma_reparam.pedestal": torch.tensor([2.0 ** -36]),
          "g.beta_reparam.lower_bound.bound": torch.tensor([(1e-6 + 2.0 ** -36) ** 0.5]),
          "g.gamma_reparam.lower_bound.bound": torch.tensor([2.0 ** -18])}
    x = torch.randn(1, c, 20, 28, generator=torch.Generator().manual_seed(5)) * 2
    r = torch.randn(1, c, 20, 28, generator=torch.Generator().manual_seed(6))
    fn = gdn_intra if flavour == "intra" else gdn_inter
    want = fn(x, Params(sd), "g", inverse=inverse) + r
    for mode in ("f32", "f16x3"):                     # the exact kernel, and GDN on the f16x3 streaming kernel
        try:
            hip.set_conv_precision(mode)
            hip.OP_LOG = []
            got = back(hip.gdn(FakeW(sd), "g", nhwc(hip, x), flavour, inverse=inverse, residual=nhwc(hip, r)))
            kernel = hip.OP_LOG[-1]["kernel"]
        finally:
            hip.OP_LOG = None
            hip.set_conv_precision("f32")
        assert ("f16x3" in kernel) == (mode == "f16x3"), kernel
        close(got, want, rtol=1e-5, atol=1e-5)


def test_dwconv(hip):
    g = torch.Generator().manual_seed(11)
    x = torch.randn(1, 48, 19, 23, generator=g)
    w = torch.randn(48, 1, 3, 3, generator=g) / 3
    b = torch.randn(48, generator=g)
    want = F.conv2d(x, w, b, padding=1, groups=48)
    got = back(hip.dwconv3x3(FakeW({"d.weight": w, "d.bias": b}), "d", nhwc(hip, x)))
    close(got, want, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("c,hin,win,hout,wout,scale", [(64, 16, 24, 32, 48, 1.0), (2, 16, 24, 32, 48, 2.0),
                                                      (2, 32, 48, 16, 24, 0.5), (64, 128, 128, 192, 192, 1.0),
                                                      (96, 8, 8, 12, 12, 1.0), (3, 17, 23, 40, 31, 1.0)])
def test_resize(hip, c, hin, win, hout, wout, scale):
    x = torch.randn(1, c, hin, win, generator=torch.Generator().manual_seed(c + hin))
    want = F.interpolate(x, size=(hout, wout), mode="bilinear", align_corners=False) * scale
    got = back(hip.resize(nhwc(hip, x), hout, wout, scale=scale))
    close(got, want, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("c,H,W,pad", [(3, 9, 11, (1, 2, 3, 4)), (64, 16, 24, (-2, -1, -3, 0)), (96, 8, 8, (2, -3, -1, 5)), (5, 7, 6, (0, 0, -2, -2)),
                                       (4, 5, 5, (-5, 6, 0, 0))])
def test_pad_crop_matches_f_pad(hip, c, H, W, pad):
    """get_depadded_feature (IntraSS.py:124-135): F.pad with zeros, negative entries cropping; the last case crops the whole
    source away on one side (all zeros)."""
    x = torch.randn(1, c, H, W, generator=torch.Generator().manual_seed(c))
    want = F.pad(x, pad, mode="constant", value=0)
    got = back(hip.pad_crop(nhwc(hip, x), pad))
    assert got.shape == want.shape and torch.equal(got, want)
    wide = nhwc(hip, torch.cat([x, x], 1))                       # a channel slice of a wider buffer (ld > C)
    assert torch.equal(back(hip.pad_crop(wide.slice(c, 2 * c), pad)), want)


def _with_option(name, value, fn):
    import ctypes as C
    from lssvc_amd._lib import lib, check
    old = C.c_int32()
    check(lib.lssvc_get_option(name.encode(), C.byref(old)))
    try:
        check(lib.lssvc_set_option(name.encode(), value))
        return fn()
    finally:
        check(lib.lssvc_set_option(name.encode(), old.value))


@pytest.mark.parametrize("c,H,W", [(48, 19, 23), (64, 32, 48), (4, 2, 2), (96, 7, 40), (32, 33, 2)])
def test_dwconv_block_kernel_is_bit_identical(hip, c, H, W):
    """Option pointwise_blocks: the depthwise 3x3 with a 2x2 output block per thread (4x4 neighbourhood loaded once) against
    the one-output-per-thread kernel, odd sizes and borders included; and against ATen."""
    g = torch.Generator().manual_seed(c + H)
    x = torch.randn(1, c, H, W, generator=g)
    w = torch.randn(c, 1, 3, 3, generator=g) / 3
    b = torch.randn(c, generator=g)
    Wt = FakeW({"d.weight": w, "d.bias": b})
    xin = nhwc(hip, x)
    blocks = _with_option("pointwise_blocks", 1, lambda: back(hip.dwconv3x3(Wt, "d", xin)))
    single = _with_option("pointwise_blocks", 0, lambda: back(hip.dwconv3x3(Wt, "d", xin)))
    assert torch.equal(blocks, single)
    close(blocks, F.conv2d(x, w, b, padding=1, groups=c), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("c,hin,win,scale", [(64, 16, 24, 1.0), (48, 33, 17, 2.0), (8, 2, 2, 1.0), (96, 37, 61, 0.5)])
def test_resize_x2_kernel_is_bit_identical_to_the_generic_one(hip, c, hin, win, scale):
    """Exact x2 upsampling of 4-channel-aligned views runs on resize_up2_kernel (9 loads per 2x2 output block); option
    pointwise_blocks = 0 sends it to the generic kernel: same src_index / lerp arithmetic, equal bits."""
    x = torch.randn(1, c, hin, win, generator=torch.Generator().manual_seed(c + hin))
    xin = nhwc(hip, x)
    fast = _with_option("pointwise_blocks", 1, lambda: back(hip.resize(xin, 2 * hin, 2 * win, scale=scale)))
    slow = _with_option("pointwise_blocks", 0, lambda: back(hip.resize(xin, 2 * hin, 2 * win, scale=scale)))
    assert torch.equal(fast, slow)
    want = F.interpolate(x, size=(2 * hin, 2 * win), mode="bilinear", align_corners=False) * scale
    close(fast, want, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("c,H,W,mag", [(3, 32, 48, 3.0), (64, 24, 40, 8.0), (48, 64, 64, 50.0), (96, 8, 12, 1.0)])
def test_flow_warp(hip, c, H, W, mag):
    from lssvc_oracle.blocks import flow_warp
    g = torch.Generator().manual_seed(c)
    x = torch.randn(1, c, H, W, generator=g)
    flow = torch.randn(1, 2, H, W, generator=g) * mag
    want = flow_warp(x, flow)
    got = back(hip.flow_warp(nhwc(hip, x), nhwc(hip, flow)))
    # bilinear weights carry the coordinate rounding of a [-1,1]-normalised fp32 grid: ~1e-7 * size
    close(got, want, rtol=0, atol=3e-5)


def test_pool_blend_add(hip):
    g = torch.Generator().manual_seed(2)
    x = torch.randn(1, 3, 32, 48, generator=g)
    close(back(hip.pool2x2(nhwc(hip, x), is_max=False)), F.avg_pool2d(x, 2, 2), rtol=1e-6, atol=1e-7)
    y = torch.randn(1, 32, 16, 24, generator=g)
    close(back(hip.pool2x2(nhwc(hip, y), is_max=True)), F.max_pool2d(y, 2, 2), rtol=0, atol=0)
    a, b = torch.randn(1, 48, 16, 24, generator=g), torch.randn(1, 48, 16, 24, generator=g)
    l = torch.randn(1, 2, 16, 24, generator=g) * 3
    wm = torch.softmax(l, dim=1)
    close(back(hip.softmax2_blend(nhwc(hip, a), nhwc(hip, b), nhwc(hip, l))), a * wm[:, 0:1] + b * wm[:, 1:2],
          rtol=1e-6, atol=1e-6)
    close(back(hip.add(nhwc(hip, a), nhwc(hip, b))), a + b, rtol=0, atol=0)
    close(back(hip.lrelu(nhwc(hip, a), 0.1)), F.leaky_relu(a, 0.1), rtol=0, atol=0)
    parts = [nhwc(hip, a), nhwc(hip, y[:, :, :16, :24].contiguous()), nhwc(hip, x[:, :, :16, :24].contiguous())]
    close(back(hip.cat(parts)), torch.cat([a, y[:, :, :16, :24], x[:, :, :16, :24]], 1), rtol=0, atol=0)


@pytest.mark.parametrize("c,wide", [(3, 4), (2, 4), (1, 4), (4, 8), (6, 8)])
def test_copy_widens_with_zero_channels(hip, c, wide):
    """lssvc_copy into a view with more channels writes the extra ones as zeros (hip_ops.pad4: the 2-3 channel inputs of the f16x3
    convs in one launch); the destination starts as NaN so that an unwritten element shows."""
    x = torch.randn(1, c, 17, 23, generator=torch.Generator().manual_seed(c))
    dst = hip.T(torch.full((17 * 23 * wide,), float("nan"), device=DEV), 17, 23, wide, wide)
    got = back(hip.copy(nhwc(hip, x), dst))
    want = torch.cat([x, torch.zeros(1, wide - c, 17, 23)], 1)
    assert torch.equal(got, want)
    if c < 4:
        assert torch.equal(back(hip.pad4(nhwc(hip, x))), torch.cat([x, torch.zeros(1, 4 - c, 17, 23)], 1))


@pytest.mark.parametrize("H,W,mag", [(16, 24, 0.0), (36, 60, 3.0), (72, 120, 9.0), (34, 46, 1.5)])
def test_spynet_prep_equals_resize_copy_warp(hip, H, W, mag):
    """lssvc_spynet_prep (one SpyNet level's input in one launch) against the three launches it replaces, bit for bit."""
    g = torch.Generator().manual_seed(H + W)
    im1, im2 = torch.rand(1, 3, H, W, generator=g), torch.rand(1, 3, H, W, generator=g)
    flow = torch.randn(1, 2, H // 2, W // 2, generator=g) * mag
    a, b, f = nhwc(hip, im1), nhwc(hip, im2), nhwc(hip, flow)
    want = hip.T(torch.full((H * W * 8,), float("nan"), device=DEV), H, W, 8, 8)
    up = hip.resize(f, H, W, scale=2.0, out=want.slice(6, 8))
    hip.copy(a, want.slice(0, 3))
    hip.flow_warp(b, up, out=want.slice(3, 6))
    got = hip.spynet_prep(a, b, f, hip.T(torch.full((H * W * 8,), float("nan"), device=DEV), H, W, 8, 8))
    assert torch.equal(back(got), back(want))


@pytest.mark.parametrize("c,H,W", [(3, 64, 96), (3, 72, 120), (5, 32, 40), (3, 36, 60)])
def test_avgpool_pyramid_equals_three_pools(hip, c, H, W):
    """lssvc_avgpool_pyramid3 against three lssvc_pool2x2 launches, bit for bit (36x60: not a multiple of 8, the level-by-level path)."""
    x = nhwc(hip, torch.rand(1, c, H, W, generator=torch.Generator().manual_seed(H)))
    got = hip.avgpool_pyramid3(x)
    want = [x]
    for _ in range(3):
        want.append(hip.pool2x2(want[-1], is_max=False))
    for a, b in zip(got[1:], want[1:]):
        assert torch.equal(back(a), back(b))
    close(back(got[1]), F.avg_pool2d(back(x), 2, 2), rtol=1e-6, atol=1e-7)


def test_layout_roundtrip(hip):
    x = torch.randn(1, 37, 19, 45)
    t = nhwc(hip, x)
    assert torch.equal(t.torch_hwc().cpu(), x[0].permute(1, 2, 0))
    assert torch.equal(back(t), x)


def test_offset_diversity(hip):
    from lssvc_oracle.blocks import Params
    from lssvc_oracle.inter import offset_diversity
    from lssvc_amd.synth import synth_state_dict
    from lssvc_amd.inter import LSSVC_extend
    sd = synth_state_dict("lssvc_extend", 0, 0.6)
    net = LSSVC_extend()
    net.load_dict(sd)
    net.to(DEV)
    g = torch.Generator().manual_seed(9)
    H, W = 32, 48
    x = torch.randn(1, 48, H, W, generator=g)
    c1 = torch.randn(1, 48, H, W, generator=g)
    wf = torch.rand(1, 3, H, W, generator=g)
    mv = torch.randn(1, 2, H, W, generator=g) * 2
    want = offset_diversity(x, torch.cat((c1, wf, mv), 1), mv, Params(sd, "align."))
    got = back(net._offset_diversity(nhwc(hip, x), [nhwc(hip, c1), nhwc(hip, wf), nhwc(hip, mv)], nhwc(hip, mv)))
    close(got, want, rtol=0, atol=1e-4)


# ----------------------------------------------------------------------------------------- entropy kernels
def test_laplace_and_factorized_bits(hip):
    from lssvc_oracle.blocks import Params
    from lssvc_oracle import entropy as E
    from lssvc_amd.synth import synth_state_dict
    sd = synth_state_dict("lssvc_extend", 1, 0.6)
    Wt = FakeW(sd)
    g = torch.Generator().manual_seed(4)
    y = torch.randn(1, 128, 9, 15, generator=g) * 6
    mu = torch.randn(1, 128, 9, 15, generator=g)
    sg = torch.exp(torch.randn(1, 128, 9, 15, generator=g) * 2.5) * 0.5 - 0.05      # includes sigma <= 0 -> clamp
    slots = hip.BitSlots(torch.device(DEV))
    yq, yh = hip.T.empty(9, 15, 128, DEV), hip.T.empty(9, 15, 128, DEV)
    hip.laplace_quant_bits(nhwc(hip, y), nhwc(hip, mu), nhwc(hip, sg), slots, 0, y_q=yq, y_hat=yh)
    q = torch.round(y - mu)
    assert torch.equal(back(yq), q) and torch.equal(back(yh), q + mu)
    hip.laplace_bits(yq, nhwc(hip, sg), slots, 1)
    z = torch.randn(1, 128, 5, 7, generator=g) * 4
    zh = hip.T.empty(5, 7, 128, DEV)
    hip.factorized_quant_bits(nhwc(hip, z), Wt.bit_estimator("bit_estimator_z"), slots, 2, z_hat=zh)
    assert torch.equal(back(zh), torch.round(z))
    s = slots.fetch()
    want = E.laplace_bits(q, sg).item()
    assert s[0] == pytest.approx(want, rel=2e-6) and s[1] == pytest.approx(want, rel=2e-6)
    assert s[2] == pytest.approx(E.factorized_bits(torch.round(z), Params(sd, "bit_estimator_z.")).item(), rel=2e-6)


def test_gaussian_and_bottleneck(hip):
    from lssvc_oracle.blocks import Params
    from lssvc_oracle import entropy as E
    from lssvc_amd.synth import synth_state_dict
    sd = synth_state_dict("intra_ss", 1, 0.6)
    Wt = FakeW(sd)
    g = torch.Generator().manual_seed(8)
    y = torch.randn(1, 96, 8, 12, generator=g) * 5
    mu = torch.randn(1, 96, 8, 12, generator=g)
    sc = torch.exp(torch.randn(1, 96, 8, 12, generator=g) * 2) * 0.3 - 0.02
    slots = hip.BitSlots(torch.device(DEV))
    yh = hip.T.empty(8, 12, 96, DEV)
    hip.gaussian_conditional(nhwc(hip, y), nhwc(hip, sc), nhwc(hip, mu), slots, 0, y_hat=yh)
    y_hat, lik = E.gaussian_conditional(y, sc, mu)
    assert torch.equal(back(yh), y_hat)
    z = torch.randn(1, 64, 4, 6, generator=g) * 3
    zh = hip.T.empty(4, 6, 64, DEV)
    hip.entropy_bottleneck(nhwc(hip, z), Wt.entropy_bottleneck("entropy_bottleneck"), slots, 1, z_hat=zh)
    z_hat, zlik = E.entropy_bottleneck(z, Params(sd, "entropy_bottleneck."))
    assert torch.equal(back(zh), z_hat)
    s = slots.fetch()
    assert s[0] == pytest.approx(torch.log(lik).double().sum().item(), rel=1e-5)
    assert s[1] == pytest.approx(torch.log(zlik).double().sum().item(), rel=1e-5)


def test_four_part_step_schedule(hip):
    """The kernel + host schedule reproduce the reference's (chunk, mask) pairing on fixed sigma/mu."""
    from lssvc_amd.inter import MASK_OF_CHUNK
    g = torch.Generator().manual_seed(3)
    H, W, C = 6, 10, 128
    y = torch.randn(1, C, H, W, generator=g) * 4
    yq, yh, sh = hip.T.zeros(H, W, C, DEV), hip.T.zeros(H, W, C, DEV), hip.T.zeros(H, W, C, DEV)
    want_q, want_h, want_s = torch.zeros_like(y), torch.zeros_like(y), torch.zeros_like(y)
    pos = ((0, 0), (0, 1), (1, 0), (1, 1))
    for step in range(4):
        mu = torch.randn(1, C, H, W, generator=g)
        sg = torch.rand(1, C, H, W, generator=g) + 0.1
        hip.four_part_step(nhwc(hip, y), nhwc(hip, mu), nhwc(hip, sg), MASK_OF_CHUNK[step], yq, yh, sh)
        for c in range(4):
            r, cc = pos[MASK_OF_CHUNK[step][c]]
            sl = (slice(None), slice(c * 32, (c + 1) * 32), slice(r, None, 2), slice(cc, None, 2))
            q = torch.round(y[sl] - mu[sl])
            want_q[sl], want_h[sl], want_s[sl] = q, q + mu[sl], sg[sl]
    assert torch.equal(back(yq), want_q) and torch.equal(back(yh), want_h) and torch.equal(back(sh), want_s)


# ----------------------------------------------------------------------------------------- f16x3 precision mode
F16X3_CASES = [([64], 64, 3, 40, 56), ([48], 48, 3, 33, 47), ([64, 64], 48, 3, 20, 36), ([32], 64, 7, 18, 30),
               ([16], 2, 7, 16, 32), ([192], 192, 3, 9, 15), ([96], 256, 3, 12, 20), ([8], 32, 7, 24, 24)]


@pytest.mark.parametrize("cins,cout,k,H,W", F16X3_CASES)
def test_conv_f16x3_matches_fp64(hip, cins, cout, k, H, W):
    """The fp16-MFMA 3-term split (hi*hi + hi*lo + lo*hi) is fp32-class: against an fp64 reference its error is
    within a small factor of the exact-fp32 kernel's own error, and ~1000x below plain fp16's."""
    g = torch.Generator().manual_seed(hash((tuple(cins), cout, k)) & 0xFFFF)
    xs = [torch.randn(1, c, H, W, generator=g) for c in cins]
    w = torch.randn(cout, sum(cins), k, k, generator=g) / math.sqrt(sum(cins) * k * k)
    b = torch.randn(cout, generator=g)
    ref = F.conv2d(torch.cat(xs, 1).double(), w.double(), b.double(), padding=k // 2)
    Wt = FakeW({"c.weight": w, "c.bias": b})
    try:
        hip.set_conv_precision("f16x3")
        got16 = back(hip.conv(Wt, "c", [nhwc(hip, x) for x in xs]))
    finally:
        hip.set_conv_precision("f32")
    got32 = back(hip.conv(Wt, "c", [nhwc(hip, x) for x in xs]))
    e16 = (got16.double() - ref).abs().max().item()
    e32 = (got32.double() - ref).abs().max().item()
    half = F.conv2d(torch.cat(xs, 1).half().float(), w.half().float(), b, padding=k // 2)
    e_half = (half.double() - ref).abs().max().item()
    assert e16 <= 8 * e32 + 1e-6, (e16, e32)
    assert e16 <= e_half / 100, (e16, e_half)


@pytest.mark.parametrize("cins,cout,H,W", [([48], 64, 64, 96), ([64], 96, 35, 51), ([64, 64], 128, 20, 36), ([96], 192, 18, 30),
                                          ([16], 16, 9, 9), ([48, 48], 48, 130, 66)])
def test_conv3x3_stride2_f16x3_matches_fp64(hip, cins, cout, H, W):
    """Stride-2 3x3 convs in the f16x3 mode (even/odd patch columns de-interleaved in LDS): same error bar as stride 1,
    with an input LeakyReLU, output activation and odd sizes in the mix."""
    g = torch.Generator().manual_seed(hash((tuple(cins), cout, H)) & 0xFFFF)
    xs = [torch.randn(1, c, H, W, generator=g) for c in cins]
    w = torch.randn(cout, sum(cins), 3, 3, generator=g) / math.sqrt(sum(cins) * 9)
    b = torch.randn(cout, generator=g)
    ref = F.leaky_relu(F.conv2d(F.leaky_relu(torch.cat(xs, 1).double(), 0.1), w.double(), b.double(), stride=2, padding=1), 0.01)
    Wt = FakeW({"c.weight": w, "c.bias": b})
    out = {}
    for mode in ("f16x3", "f32"):
        try:
            hip.set_conv_precision(mode)
            out[mode] = back(hip.conv(Wt, "c", [nhwc(hip, x) for x in xs], stride=2, in_act="lrelu", in_slope=0.1, act="lrelu", slope=0.01))
        finally:
            hip.set_conv_precision("f32")
    assert out["f16x3"].shape == ref.shape
    e16 = (out["f16x3"].double() - ref).abs().max().item()
    e32 = (out["f32"].double() - ref).abs().max().item()
    assert e16 <= 8 * e32 + 1e-6, (e16, e32)


def test_conv_f16x3_fused_paths(hip):
    g = torch.Generator().manual_seed(21)
    x = torch.randn(1, 64, 24, 40, generator=g)
    r = torch.randn(1, 64, 48, 80, generator=g)
    w = torch.randn(256, 64, 3, 3, generator=g) / 24
    b = torch.randn(256, generator=g)
    Wt = FakeW({"s.0.weight": w, "s.0.bias": b})
    want = F.leaky_relu(F.pixel_shuffle(F.conv2d(F.leaky_relu(x, 0.1), w, b, padding=1), 2), 0.01)
    try:
        hip.set_conv_precision("f16x3")
        got = back(hip.subpel(Wt, "s", nhwc(hip, x), in_act="lrelu", in_slope=0.1, act="lrelu"))
    finally:
        hip.set_conv_precision("f32")
    close(got, want, rtol=1e-5, atol=1e-5)


PW_CASES = [([64], 256, 40, 56), ([256], 64, 33, 47), ([48], 48, 20, 36), ([64, 64], 64, 18, 30), ([128, 256], 384, 9, 15),
            ([512], 128, 12, 20), ([1024], 384, 8, 8), ([32], 128, 17, 19), ([48], 32, 16, 16), ([8], 16, 5, 7)]


@pytest.mark.parametrize("cins,cout,H,W", PW_CASES)
def test_conv1x1_f16x3_matches_fp64(hip, cins, cout, H, W):
    g = torch.Generator().manual_seed(hash((tuple(cins), cout)) & 0xFFFF)
    xs = [torch.randn(1, c, H, W, generator=g) for c in cins]
    w = torch.randn(cout, sum(cins), 1, 1, generator=g) / math.sqrt(sum(cins))
    b = torch.randn(cout, generator=g)
    r = torch.randn(1, cout, H, W, generator=g)
    ref = F.leaky_relu(F.conv2d(torch.cat(xs, 1).double(), w.double(), b.double()), 0.1) + r.double()
    Wt = FakeW({"c.weight": w, "c.bias": b})
    out = {}
    for mode in ("f16x3", "f32"):
        try:
            hip.set_conv_precision(mode)
            out[mode] = back(hip.conv(Wt, "c", [nhwc(hip, x) for x in xs], act="lrelu", slope=0.1, residual=nhwc(hip, r)))
        finally:
            hip.set_conv_precision("f32")
    e16 = (out["f16x3"].double() - ref).abs().max().item()
    e32 = (out["f32"].double() - ref).abs().max().item()
    assert e16 <= 8 * e32 + 1e-6, (e16, e32)


def test_subpel1x1_f16x3(hip):
    g = torch.Generator().manual_seed(33)
    x = torch.randn(1, 128, 12, 20, generator=g)
    w = torch.randn(256, 128, 1, 1, generator=g) / 11
    b = torch.randn(256, generator=g)
    want = F.pixel_shuffle(F.conv2d(x, w, b), 2)
    Wt = FakeW({"s.0.weight": w, "s.0.bias": b})
    try:
        hip.set_conv_precision("f16x3")
        got = back(hip.subpel(Wt, "s", nhwc(hip, x)))
    finally:
        hip.set_conv_precision("f32")
    close(got, want, rtol=1e-5, atol=1e-5)


# ----------------------------------------------------------------------------------------- fused DepthConvBlock tail
def _ffn_weights(g, c, hidden, pre_cin=None):
    sd = {"f.conv.0.weight": torch.randn(hidden, c, 1, 1, generator=g) / math.sqrt(c), "f.conv.0.bias": torch.randn(hidden, generator=g) * 0.1,
          "f.conv.2.weight": torch.randn(c, hidden, 1, 1, generator=g) / math.sqrt(hidden), "f.conv.2.bias": torch.randn(c, generator=g) * 0.1}
    if pre_cin:
        sd["p.weight"] = torch.randn(c, pre_cin, 1, 1, generator=g) / math.sqrt(pre_cin)
        sd["p.bias"] = torch.randn(c, generator=g) * 0.1
    return sd


def _ffn_ref(sd, o1):
    h = F.leaky_relu(F.conv2d(o1, sd["f.conv.0.weight"].double(), sd["f.conv.0.bias"].double()), 0.1)
    return o1 + F.leaky_relu(F.conv2d(h, sd["f.conv.2.weight"].double(), sd["f.conv.2.bias"].double()), 0.1)


@pytest.mark.parametrize("c,hidden,pre_cin,H,W", [(64, 256, None, 24, 40), (48, 192, None, 19, 23), (32, 128, None, 16, 16),
                                                 (64, 256, 64, 21, 35), (48, 192, 48, 16, 48), (32, 128, 32, 7, 9),
                                                 (48, 192, 64, 12, 20), (32, 64, 24, 10, 10), (64, 128, 40, 9, 33),
                                                 # streamed-weights variant: wide blocks / wide leading conv
                                                 (128, 512, 128, 18, 30), (128, 512, None, 11, 13), (64, 256, 128, 16, 24),
                                                 (96, 384, 96, 9, 17), (128, 1024, 128, 8, 16), (96, 384, None, 5, 7)])
def test_ffn_fused_matches_fp64(hip, c, hidden, pre_cin, H, W):
    """lssvc_ffn_f16x3 (DepthConv.conv2 + identity + ConvFFN in one launch) against an fp64 reference; error budget as
    for the unfused f16x3 convs: within 8x of what the exact-fp32 kernels give for the same chain."""
    g = torch.Generator().manual_seed(c * 1000 + hidden + (pre_cin or 0))
    sd = _ffn_weights(g, c, hidden, pre_cin)
    Wt = FakeW(sd)
    try:
        hip.set_conv_precision("f16x3")
        assert hip.ffn_fusable(Wt, "f", "p" if pre_cin else None, c, pre_cin or 0)
        if pre_cin:
            t = torch.randn(1, pre_cin, H, W, generator=g)
            ident = torch.randn(1, c, H, W, generator=g)
            o1 = F.conv2d(t.double(), sd["p.weight"].double(), sd["p.bias"].double()) + ident.double()
            got = back(hip.ffn_block(Wt, "f", pre_name="p", pre_in=nhwc(hip, t), ident=nhwc(hip, ident)))
            hip.set_conv_precision("f32")
            u = hip.conv(Wt, "p", nhwc(hip, t), residual=nhwc(hip, ident))
        else:
            x = torch.randn(1, c, H, W, generator=g)
            o1 = x.double()
            got = back(hip.ffn_block(Wt, "f", x=nhwc(hip, x)))
            hip.set_conv_precision("f32")
            u = nhwc(hip, x)
        v = hip.conv(Wt, "f.conv.0", u, act="lrelu", slope=0.1)
        unfused = back(hip.conv(Wt, "f.conv.2", v, act="lrelu", slope=0.1, residual=u))
    finally:
        hip.set_conv_precision("f32")
    ref = _ffn_ref(sd, o1)
    e16 = (got.double() - ref).abs().max().item()
    e32 = (unfused.double() - ref).abs().max().item()
    assert e16 <= 8 * e32 + 1e-6, (e16, e32)


def test_ffn_fused_on_channel_slices_and_in_place(hip):
    """Views with ld > C (slices of a concat buffer) for every operand, and out aliasing ident."""
    g = torch.Generator().manual_seed(77)
    c, hidden, H, W = 48, 192, 13, 21
    sd = _ffn_weights(g, c, hidden, 48)
    Wt = FakeW(sd)
    wide_t = torch.randn(1, 96, H, W, generator=g)
    wide_i = torch.randn(1, 112, H, W, generator=g)
    o1 = F.conv2d(wide_t[:, 48:96].double(), sd["p.weight"].double(), sd["p.bias"].double()) + wide_i[:, 16:64].double()
    ref = _ffn_ref(sd, o1)
    try:
        hip.set_conv_precision("f16x3")
        bt, bi = nhwc(hip, wide_t), nhwc(hip, wide_i)
        out = hip.ffn_block(Wt, "f", pre_name="p", pre_in=bt.slice(48, 96), ident=bi.slice(16, 64), out=bi.slice(16, 64))
    finally:
        hip.set_conv_precision("f32")
    res = back(bi)
    close(res[:, 16:64], ref.float(), rtol=2e-5, atol=2e-5)
    assert torch.equal(res[:, :16], wide_i[:, :16]) and torch.equal(res[:, 64:], wide_i[:, 64:])


def test_depth_conv_block_fused_equals_unfused(hip):
    from lssvc_amd import blocks
    g = torch.Generator().manual_seed(5)
    c, H, W = 64, 20, 28
    sd = {"b.block.0.conv1.0.weight": torch.randn(c, c, 1, 1, generator=g) / 8, "b.block.0.conv1.0.bias": torch.randn(c, generator=g) * 0.1,
          "b.block.0.depth_conv.weight": torch.randn(c, 1, 3, 3, generator=g) / 3, "b.block.0.depth_conv.bias": torch.randn(c, generator=g) * 0.1,
          "b.block.0.conv2.weight": torch.randn(c, c, 1, 1, generator=g) / 8, "b.block.0.conv2.bias": torch.randn(c, generator=g) * 0.1}
    for k, v in _ffn_weights(g, c, 4 * c).items():
        sd[k.replace("f.", "b.block.1.")] = v
    Wt = FakeW(sd)
    x = torch.randn(1, c, H, W, generator=g)
    outs = {}
    try:
        hip.set_conv_precision("f16x3")
        for fuse in (True, False):
            hip.FUSE_FFN = fuse
            outs[fuse] = back(blocks.depth_conv_block(Wt, "b", nhwc(hip, x)))
    finally:
        hip.FUSE_FFN = True
        hip.set_conv_precision("f32")
    close(outs[True], outs[False], rtol=1e-5, atol=1e-5)


# ----------------------------------------------------------------------------------------- fused DepthConv front half
@pytest.mark.parametrize("cins,c,H,W", [([64], 64, 37, 45), ([48], 48, 16, 16), ([32], 32, 50, 18), ([32, 32], 64, 33, 70),
                                        ([16, 16, 16], 48, 19, 21), ([24], 32, 7, 5), ([64], 64, 128, 96)])
def test_conv1x1_dw3x3_fused_is_bit_identical(hip, cins, c, H, W):
    """lssvc_conv1x1_dw3x3_f16x3 == lssvc_conv2d (1x1, LeakyReLU) followed by lssvc_dwconv3x3, bit for bit: same K order
    in the 1x1, same tap order in the depthwise conv, zero padding applied to the intermediate."""
    g = torch.Generator().manual_seed(sum(cins) * 100 + H)
    cin = sum(cins)
    sd = {"c.weight": torch.randn(c, cin, 1, 1, generator=g) / math.sqrt(cin), "c.bias": torch.randn(c, generator=g) * 0.2,
          "d.weight": torch.randn(c, 1, 3, 3, generator=g) / 3, "d.bias": torch.randn(c, generator=g) * 0.2}
    Wt = FakeW(sd)
    xs = [torch.randn(1, ci, H, W, generator=g) for ci in cins]
    try:
        hip.set_conv_precision("f16x3")
        ins = [nhwc(hip, x) for x in xs]
        fused = hip.conv1x1_dw3x3(Wt, "c", "d", ins, slope=0.01)
        assert fused is not None
        t = hip.conv(Wt, "c", ins, act="lrelu", slope=0.01)
        two = hip.dwconv3x3(Wt, "d", t)
    finally:
        hip.set_conv_precision("f32")
    assert torch.equal(back(fused), back(two))
    ref = F.conv2d(F.leaky_relu(F.conv2d(torch.cat(xs, 1), sd["c.weight"], sd["c.bias"]), 0.01), sd["d.weight"], sd["d.bias"], padding=1, groups=c)
    close(back(fused), ref, rtol=2e-5, atol=2e-5)


def test_conv1x1_dw3x3_declines_uncovered_shapes(hip):
    g = torch.Generator().manual_seed(1)
    sd = {"c.weight": torch.randn(128, 128, 1, 1, generator=g), "c.bias": torch.randn(128, generator=g),
          "d.weight": torch.randn(128, 1, 3, 3, generator=g), "d.bias": torch.randn(128, generator=g)}
    try:
        hip.set_conv_precision("f16x3")
        assert hip.conv1x1_dw3x3(FakeW(sd), "c", "d", nhwc(hip, torch.randn(1, 128, 8, 8, generator=g))) is None
        hip.set_conv_precision("f32")
        assert hip.conv1x1_dw3x3(FakeW(sd), "c", "d", nhwc(hip, torch.randn(1, 128, 8, 8, generator=g))) is None
    finally:
        hip.set_conv_precision("f32")


@pytest.mark.parametrize("cins,cout,k,stride", [([3, 48], 64, 3, 2), ([2], 64, 3, 2), ([3], 64, 3, 1), ([64, 3], 32, 3, 1), ([2, 2, 4], 16, 7, 1),
                                                ([3, 64], 64, 1, 1)])
def test_narrow_inputs_take_the_f16x3_path(hip, cins, cout, k, stride):
    """2-3 channel inputs (RGB, flow) are padded to 4 channels so the conv can run in the f16x3 mode; the result must
    match the exact-fp32 kernel on the unpadded views within the usual f16x3 error budget."""
    g = torch.Generator().manual_seed(sum(cins) + cout + k)
    H, W = 22, 38
    xs = [torch.randn(1, c, H, W, generator=g) for c in cins]
    cin = sum(cins)
    w = torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)
    b = torch.randn(cout, generator=g)
    ref = F.conv2d(torch.cat(xs, 1).double(), w.double(), b.double(), stride=stride, padding=k // 2)
    Wt = FakeW({"c.weight": w, "c.bias": b})
    out = {}
    for mode in ("f16x3", "f32"):
        try:
            hip.set_conv_precision(mode)
            hip.OP_LOG = []
            out[mode] = back(hip.conv(Wt, "c", [nhwc(hip, x) for x in xs], stride=stride))
            kernel = hip.OP_LOG[-1]["kernel"]
        finally:
            hip.OP_LOG = None
            hip.set_conv_precision("f32")
        if mode == "f16x3":
            assert "f16x3" in kernel, kernel                   # really left the fp32 kernel
    e16 = (out["f16x3"].double() - ref).abs().max().item()
    e32 = (out["f32"].double() - ref).abs().max().item()
    assert e16 <= 8 * e32 + 1e-6, (e16, e32)


@pytest.mark.parametrize("c,hidden,H,W", [(64, 256, 21, 35), (128, 512, 11, 13)])
def test_ffn_fused_outer_skip(hip, c, hidden, H, W):
    """The optional `skip` operand of lssvc_ffn_f16x3 (block(x) + skip, lssvc_modules.py:363): equals the fused block
    followed by a separate add, bit for bit, on both the resident and the streamed kernel; `out` may alias `skip`."""
    g = torch.Generator().manual_seed(c + hidden)
    sd = _ffn_weights(g, c, hidden, c)
    Wt = FakeW(sd)
    t, ident, skip = (torch.randn(1, c, H, W, generator=g) for _ in range(3))
    try:
        hip.set_conv_precision("f16x3")
        plain = hip.ffn_block(Wt, "f", pre_name="p", pre_in=nhwc(hip, t), ident=nhwc(hip, ident))
        want = back(hip.add(plain, nhwc(hip, skip)))
        got = back(hip.ffn_block(Wt, "f", pre_name="p", pre_in=nhwc(hip, t), ident=nhwc(hip, ident), skip=nhwc(hip, skip)))
        sk = nhwc(hip, skip)
        inplace = back(hip.ffn_block(Wt, "f", pre_name="p", pre_in=nhwc(hip, t), ident=nhwc(hip, ident), skip=sk, out=sk))
    finally:
        hip.set_conv_precision("f32")
    assert torch.equal(got, want) and torch.equal(inplace, want)


@pytest.mark.parametrize("cin,cout,k,H,W", [(64, 64, 3, 40, 56), (48, 48, 3, 300, 340), (64, 64, 3, 293, 331), (96, 96, 3, 19, 23), (64, 64, 1, 17, 33)])
def test_second_residual_equals_separate_add(hip, cin, cout, k, H, W):
    """lssvc_conv_desc.residual2 (the `skip + res_block(x)` sums of the context-fusion nets folded into the block's last conv):
    (act(conv) + residual) + residual2 in one launch is bit-identical to the conv followed by lssvc_add, in both precisions
    and on the tiled as well as the persistent kernels."""
    g = torch.Generator().manual_seed(cin + H)
    x = torch.randn(1, cin, H, W, generator=g)
    r1, r2 = torch.randn(1, cout, H, W, generator=g), torch.randn(1, cout, H, W, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)
    b = torch.randn(cout, generator=g)
    Wt = FakeW({"c.weight": w, "c.bias": b})
    for mode in ("f16x3", "f32"):
        try:
            hip.set_conv_precision(mode)
            two = back(hip.add(nhwc(hip, r2), hip.conv(Wt, "c", nhwc(hip, x), act="lrelu", slope=0.1, residual=nhwc(hip, r1))))
            one = back(hip.conv(Wt, "c", nhwc(hip, x), act="lrelu", slope=0.1, residual=nhwc(hip, r1), residual2=nhwc(hip, r2)))
        finally:
            hip.set_conv_precision("f32")
        assert torch.equal(one, two), mode
    want = F.leaky_relu(F.conv2d(x, w, b, padding=k // 2), 0.1) + r1 + r2
    close(one, want)


def test_fp16_range_audit_moves_saturating_layers_to_fp32():
    """The f16x3 kernels saturate activations at +-65504 while staging them as fp16 hi/lo parts; the reference's fp32 convs
    do not. A checkpoint whose FUNCTION is unchanged but whose intermediate tensor is 2^17 times larger (conv1 scaled by
    2^17, conv2 by 2^-17 around a positively homogeneous LeakyReLU; exact powers of two) must therefore give the same
    I-frame: the range audit of the first frame sees max |input| of g_a.0.conv2 beyond 2^15, moves that layer to the exact
    fp32 kernel, warns, and recomputes. Without the audit the same checkpoint saturates silently and the frame is wrong."""
    import warnings
    from lssvc_amd import IntraSS, hip_ops
    from lssvc_amd.synth import synth_state_dict, synth_clip
    from lssvc_amd.preprocess import imresize_bicubic
    old_precision = hip_ops.CONV_PRECISION
    hip_ops.set_conv_precision("f16x3")
    try:
        _range_audit_case(IntraSS, hip_ops, synth_state_dict, synth_clip, imresize_bicubic, warnings)
    finally:
        hip_ops.set_conv_precision(old_precision)


def _range_audit_case(IntraSS, hip_ops, synth_state_dict, synth_clip, imresize_bicubic, warnings):
    sd = synth_state_dict("intra_ss", 0, 0.6)
    big = dict(sd)
    K = 2.0 ** 17
    big["base_layer_model.g_a.0.conv1.weight"] = sd["base_layer_model.g_a.0.conv1.weight"] * K
    big["base_layer_model.g_a.0.conv1.bias"] = sd["base_layer_model.g_a.0.conv1.bias"] * K
    big["base_layer_model.g_a.0.conv2.weight"] = sd["base_layer_model.g_a.0.conv2.weight"] / K
    x_el = synth_clip(1, 128, 128, seed=1).float().div(255.0).to(DEV)
    x_bl = imresize_bicubic(x_el.cpu(), (64, 64)).clamp_(0, 1).to(DEV)

    def code(state, audit):
        net = IntraSS.from_state_dict(state).to(DEV).eval()
        net.range_audit = audit
        net.set_scale_information(2.0, (128, 128), (0, 0, 0, 0))
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            r = net.encode_decode(x_bl, x_el, None, None)
        return net, r, [str(m.message) for m in w]

    ref_net, ref, ref_w = code(sd, True)
    assert not ref_w and not ref_net.W.force_f32                      # the ordinary checkpoint stays entirely on f16x3
    rep = next(iter(ref_net.audit_report.values()))
    assert len(rep) > 50 and 0 < max(rep.values()) < hip_ops.F16_INPUT_LIMIT
    net, got, msgs = code(big, True)
    assert net.W.force_f32 == {"base_layer_model.g_a.0.conv2"}, net.W.force_f32
    assert len(msgs) == 1 and "g_a.0.conv2" in msgs[0]
    assert abs(got["bit_bl"] - ref["bit_bl"]) / (64 * 64) <= 1e-5 and abs(got["bit_el"] - ref["bit_el"]) / (128 * 128) <= 1e-5
    assert (got["x_hat_el"] - ref["x_hat_el"]).abs().max().item() <= 2e-4
    assert net.get_f32_layers() == ["base_layer_model.g_a.0.conv2"]
    r2 = net.encode_decode(x_bl, x_el, None, None)                     # later frames: no audit, the layer stays on fp32
    assert r2["bit_bl"] == got["bit_bl"] and torch.equal(r2["x_hat_el"], got["x_hat_el"])
    fresh = IntraSS.from_state_dict(big).to(DEV).eval()                # a second process is handed the list instead of auditing
    fresh.range_audit = False
    fresh.set_f32_layers(net.get_f32_layers())
    fresh.set_scale_information(2.0, (128, 128), (0, 0, 0, 0))
    r3 = fresh.encode_decode(x_bl, x_el, None, None)
    assert r3["bit_bl"] == got["bit_bl"] and torch.equal(r3["x_hat_el"], got["x_hat_el"])
    _, blind, _ = code(big, False)                                     # audit off: silent saturation, visibly wrong
    assert abs(blind["bit_bl"] - ref["bit_bl"]) / (64 * 64) > 1e-3 or (blind["x_hat_el"] - ref["x_hat_el"]).abs().max().item() > 1e-2
