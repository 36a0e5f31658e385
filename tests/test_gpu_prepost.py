"""The frame pre/post-processing kernels (csrc/prepost.hip) against the torch restatements of what the
reference's test.py does on the host (lssvc_amd.preprocess / harness helpers, themselves pinned on the CPU against the
reference's formulas and the golden x_bl fixtures). Tolerances: 1e-6 on pixels (a different fp32 summation order in the
10-tap bicubic filter), exact on the normalised 8-bit planes, 1e-10 relative on the fp64 squared-error sums."""
import numpy as np
import pytest
import torch

from helpers import load_case

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def prep():
    from lssvc_amd.prepost import FramePrep
    return FramePrep(DEV)


@pytest.mark.parametrize("H,W,ratio", [(1080, 1920, 2.0), (128, 128, 2.0), (360, 640, 1.5), (270, 482, 2.0)])
def test_rgb8_frame_and_bicubic_base_layer(prep, H, W, ratio):
    from lssvc_amd import preprocess
    from lssvc_amd.synth import synth_clip
    u8 = synth_clip(1, H, W, seed=H)[0].to(DEV)                                   # (3,H,W) uint8
    x_bl, x_el, pad = prep.make_layers_rgb8(u8, ratio)
    # the pinned restatement runs on the CPU like the reference (true division by 255; torch's CUDA kernel multiplies by
    # the reciprocal instead, 1 ulp off)
    want_bl, want_el, want_pad = preprocess.make_layers(u8.cpu()[None].float() / 255.0, ratio)
    assert pad == want_pad and tuple(x_el.shape) == tuple(want_el.shape) and tuple(x_bl.shape) == tuple(want_bl.shape)
    assert torch.equal(x_el.cpu(), want_el)                                        # u8 / 255 and zero padding: exact
    assert (x_bl.cpu() - want_bl).abs().max().item() <= 1e-6
    assert x_bl.min().item() >= 0.0 and x_bl.max().item() <= 1.0


@pytest.mark.parametrize("H,W,ratio", [(1080, 1920, 2.0), (360, 640, 1.5), (270, 482, 2.0), (720, 1280, 4.0)])
def test_bicubic_row_kernel_is_bit_identical_to_the_per_output_kernel(prep, H, W, ratio):
    """Round 6: lssvc_resample2d as a separable pass (vertical sums once per source column, in the LDS) against the per-output kernel it
    replaces (option resample_rows = 0): the same operations on every element in the same order."""
    from lssvc_amd._lib import lib, check
    from lssvc_amd.synth import synth_clip
    u8 = synth_clip(1, H, W, seed=W)[0].to(DEV)
    outs = []
    for rows in (1, 0, 1):
        check(lib.lssvc_set_option(b"resample_rows", rows))
        try:
            outs.append(prep.make_layers_rgb8(u8, ratio)[0].clone())
        finally:
            check(lib.lssvc_set_option(b"resample_rows", 1))
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_bicubic_matches_reference_fixture(prep):
    """x_bl of the golden cases was produced by the reference's own imresize (tests/golden/make_golden.py)."""
    from lssvc_amd.hip_ops import T
    z, m = load_case("x2_128x256_ip")
    x_el = (torch.from_numpy(z["x_el_u8"][0:1]).float() / 255.0).to(DEV)
    got = prep.bicubic(T.from_nchw(x_el), (m["h"], m["w"])).to_nchw().cpu()
    np.testing.assert_allclose(got.numpy(), z["x_bl"][0:1], atol=1e-6, rtol=0)


@pytest.mark.parametrize("H,W,Hp,Wp", [(128, 192, 128, 192), (1080, 1920, 1152, 1920), (66, 34, 128, 64)])
def test_yuv420_to_frame(prep, H, W, Hp, Wp):
    import colour_torch_ref as CT
    g = np.random.default_rng(H)
    y, u, v = (g.integers(0, 256, s, dtype=np.uint8) for s in ((H, W), (H // 2, W // 2), (H // 2, W // 2)))
    want, wy, wu, wv = CT.yuv420_to_rgb(y, u, v, "cpu")      # the torch restatement on the CPU, like the reference (x / 255 is a
    #                                                          true division there; torch's CUDA kernel multiplies by 1/255)
    f, (py, pu, pv) = prep.frame_from_yuv420(*(torch.from_numpy(a).to(DEV) for a in (y, u, v)), (Hp, Wp))
    got = f.to_nchw().cpu()
    assert (got[:, :, :H, :W] - want).abs().max().item() <= 1e-6
    assert got[:, :, H:, :].abs().max().item() == 0 if Hp > H else True
    assert got[:, :, :, W:].abs().max().item() == 0 if Wp > W else True
    assert torch.equal(py.cpu(), wy) and torch.equal(pu.cpu(), wu) and torch.equal(pv.cpu(), wv)


def test_rgb_to_yuv420_and_sqdiff(prep):
    import colour_torch_ref as CT
    from lssvc_amd.hip_ops import T
    g = torch.Generator().manual_seed(3)
    a = torch.rand(1, 3, 72, 96, generator=g) * 1.4 - 0.2                          # includes values outside [0,1]
    b = torch.rand(1, 3, 72, 96, generator=g)
    ta, tb = T.from_nchw(a.to(DEV)), T.from_nchw(b.to(DEV))
    for clamp in (False, True):
        src = a.clamp(0, 1) if clamp else a
        wy, wu, wv = CT.rgb_to_yuv420(src[:, :, :64, :80])
        y, u, v = prep.rgb_to_yuv420(ta, 64, 80, clamp01=clamp)
        for got, want in ((y, wy), (u, wu), (v, wv)):
            assert (got.cpu() - want).abs().max().item() <= 1e-6
        prep.sqdiff_frames(ta, tb, 64, 80, 0, clamp01=clamp)
        d = (src[:, :, :64, :80] - b[:, :, :64, :80]).float()
        assert prep.fetch()[0] == pytest.approx((d * d).double().sum().item(), rel=1e-10)
    prep.sqdiff_planes(y, v.new_zeros(y.shape), 5)
    assert prep.fetch()[5] == pytest.approx((y.cpu().float() ** 2).double().sum().item(), rel=1e-10)
