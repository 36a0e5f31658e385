"""ORACLE (test infrastructure only) -- I-frame path: IntraSS.forward in estimate mode
(IntraSS.py:137-172) with its BL codec IntraNoAR.get_layer_information (priors.py:368-388).
See blocks.py for the rules this package follows.
"""
import torch

from .blocks import (Params, conv, subpel, lrelu, res_block, residual_block, residual_block_with_stride,
                     residual_block_upsample, gdn_intra, bilinear)
from .entropy import entropy_bottleneck, gaussian_conditional, bits_from_likelihoods


def _seq_lrelu_convs(x, p, name, layout):
    """Run an nn.Sequential of convs / subpel convs separated by LeakyReLU(0.01).
    layout: list of (index, kind, stride) with kind in {'conv','subpel'}; lrelu between items."""
    q = p.sub(name)
    for i, (idx, kind, stride) in enumerate(layout):
        if i:
            x = lrelu(x)
        x = conv(x, q, str(idx), stride=stride) if kind == "conv" else subpel(x, q, str(idx))
    return x


def bl_g_a(x, p):
    """IntraNoAR.g_a (priors.py:116-124)."""
    q = p.sub("g_a")
    x = residual_block_with_stride(x, q, "0")
    x = residual_block(x, q, "1")
    x = residual_block_with_stride(x, q, "2")
    x = residual_block(x, q, "3")
    x = residual_block_with_stride(x, q, "4")
    x = residual_block(x, q, "5")
    return conv(x, q, "6", stride=2)


def bl_g_s(x, p):
    """IntraNoAR.g_s (priors.py:150-159)."""
    q = p.sub("g_s")
    x = residual_block(x, q, "0")
    x = residual_block_upsample(x, q, "1")
    x = residual_block(x, q, "2")
    x = residual_block_upsample(x, q, "3")
    x = residual_block(x, q, "4")
    x = residual_block_upsample(x, q, "5")
    x = residual_block(x, q, "6")
    return subpel(x, q, "7")


def bl_layer_information(x, p):
    """IntraNoAR.get_layer_information (priors.py:368-388) -> bits, x_hat, y_hat (+ y, z)."""
    y = bl_g_a(x, p)
    z = _seq_lrelu_convs(y, p, "h_a", [(0, "conv", 1), (2, "conv", 1), (4, "conv", 2), (6, "conv", 1), (8, "conv", 2)])
    z_hat, z_lik = entropy_bottleneck(z, p.sub("entropy_bottleneck"))
    g = _seq_lrelu_convs(z_hat, p, "h_s", [(0, "conv", 1), (2, "subpel", 1), (4, "conv", 1), (6, "subpel", 1), (8, "conv", 1)])
    scales, means = g.chunk(2, 1)
    y_hat, y_lik = gaussian_conditional(y, scales, means)
    x_hat = bl_g_s(y_hat, p)
    med = p.sub("entropy_bottleneck")["quantiles"][:, 0, 1].view(1, -1, 1, 1)
    return {"bits": bits_from_likelihoods(y_lik, z_lik), "x_hat": x_hat, "y_hat": y_hat, "y": y, "z": z,
            "y_q": torch.round(y - means), "z_q": torch.round(z - med)}      # the integers the coder sees (tests: symbol planes)


def texture_resampler(x, p, shape_hr):
    """Intra TextureResampler (layers.py:258-270)."""
    f = conv(lrelu(conv(x, p, "conv_adaptor.0")), p, "conv_adaptor.2")
    return bilinear(f, shape_hr)


def layer_prior_resampler(y_hat_bl, p, shape_hr):
    """Intra LayerPriorResampler (layers.py:273-285)."""
    f = conv(lrelu(conv(y_hat_bl, p, "conv_adaptor.0")), p, "conv_adaptor.2")
    return bilinear(f, (shape_hr[0] // 16, shape_hr[1] // 16))


def texture_extractor(f, p):
    """MultiScaleTextureExtractor (layers.py:288-308)."""
    l1 = res_block(conv(f, p, "conv1"), p, "res_block1")
    l2 = res_block(conv(l1, p, "conv2", stride=2), p, "res_block2")
    l3 = res_block(conv(l2, p, "conv3", stride=2), p, "res_block3")
    return l1, l2, l3


def context_fusion(t1, t2, t3, p):
    """MultiScaleTextureFusion (layers.py:311-339); the inter-frame MultiScaleContextFusion nets
    (dmc_net.py:34-62, lssvc_modules.py:203-232) share this exact dataflow."""
    c3_up = res_block(subpel(t3, p, "conv3_up"), p, "res_block3_up")
    c3_out = res_block(conv(t3, p, "conv3_out"), p, "res_block3_out")
    cat2 = torch.cat((c3_up, t2), dim=1)
    c2_up = res_block(subpel(cat2, p, "conv2_up"), p, "res_block2_up")
    c2_out = res_block(conv(cat2, p, "conv2_out"), p, "res_block2_out")
    c1_out = res_block(conv(torch.cat((c2_up, t1), dim=1), p, "conv1_out"), p, "res_block1_out")
    return t1 + c1_out, t2 + c2_out, t3 + c3_out


def res_encoder(x, c1, c2, c3, p, gdn):
    """Contextual analysis transform: Intra ResEncoder (layers.py:342-367) and DMC ResEncoder
    (dmc_net.py:65-90) -- identical dataflow, `gdn` picks the GDN flavour."""
    f = gdn(conv(torch.cat([x, c1], 1), p, "conv1", stride=2), p, "gdn1")
    f = res_block(torch.cat([f, c2], 1), p, "res1", slope=0.1, start_from_relu=False, end_with_relu=True)
    f = gdn(conv(f, p, "conv2", stride=2), p, "gdn2")
    f = res_block(torch.cat([f, c3], 1), p, "res2", slope=0.1, start_from_relu=False, end_with_relu=True)
    f = gdn(conv(f, p, "conv3", stride=2), p, "gdn3")
    return conv(f, p, "conv4", stride=2)


def res_decoder(y_hat, c2, c3, p, gdn):
    """Contextual synthesis transform: Intra ResDecoder (layers.py:370-395), DMC ResDecoder (dmc_net.py:93-118)."""
    f = gdn(subpel(y_hat, p, "up1"), p, "gdn1", inverse=True)
    f = gdn(subpel(f, p, "up2"), p, "gdn2", inverse=True)
    f = res_block(torch.cat([f, c3], 1), p, "res1", slope=0.1, start_from_relu=False, end_with_relu=True)
    f = gdn(subpel(f, p, "up3"), p, "gdn3", inverse=True)
    f = res_block(torch.cat([f, c2], 1), p, "res2", slope=0.1, start_from_relu=False, end_with_relu=True)
    return subpel(f, p, "up4")


def recon_generation(res, ctx1, p):
    """ReconGeneration called as recon_net(res_hat, context1) => cat(res, ctx1)
    (layers.py:398-411 with IntraSS.py:161; dmc_net.py:143-156 with dmc_net.py:452)."""
    f = conv(torch.cat((res, ctx1), dim=1), p, "feature_conv.0")
    f = res_block(f, p, "feature_conv.1")
    f = res_block(f, p, "feature_conv.2")
    return f, conv(f, p, "recon_conv")


def prior_fusion(hyper, layer, ctx3, p):
    """Intra PriorFusion (layers.py:473-492)."""
    c = conv(lrelu(conv(ctx3, p, "context_parameters.0", stride=2), 0.1), p, "context_parameters.2", stride=2)
    t = torch.cat([hyper, layer, c], dim=1)
    t = lrelu(conv(t, p, "params_net.0"))
    t = lrelu(conv(t, p, "params_net.2"))
    return conv(t, p, "params_net.4")


def depad(feature, pad_size, p=1):
    """get_depadded_feature (IntraSS.py:124-135, LSSVC_net.py:271-282): F.pad by pad_size / p, zeros in, negative = crop."""
    if feature is None:
        return None
    return torch.nn.functional.pad(feature, tuple(int(v / p) for v in pad_size), mode="constant", value=0)


def intra_forward(sd, x_bl, x_el, shape_hr, extras=False, pad_size=(0, 0, 0, 0)):
    """IntraSS.forward (IntraSS.py:137-172); pad_size=(0,0,0,0) is what test.py:212 always passes."""
    p = Params(sd)
    bl = bl_layer_information(x_bl, p.sub("base_layer_model"))
    x_hat_bl, y_hat_bl = bl["x_hat"], bl["y_hat"]
    x_hat_bl_full = x_hat_bl
    x_hat_bl, y_hat_bl = depad(x_hat_bl, pad_size), depad(y_hat_bl, pad_size, 16)       # IntraSS.py:146-147

    tex = texture_resampler(x_hat_bl, p.sub("texture_resampler"), shape_hr)
    t1, t2, t3 = texture_extractor(tex, p.sub("texture_extractor"))
    c1, c2, c3 = context_fusion(t1, t2, t3, p.sub("context_fusion_net"))

    y = res_encoder(x_el, c1, c2, c3, p.sub("g_a"), gdn_intra)
    z = _seq_lrelu_convs(y, p, "h_a", [(0, "conv", 1), (2, "conv", 2), (4, "conv", 2)])
    z_hat, z_lik = entropy_bottleneck(z, p.sub("entropy_bottleneck"))
    hyper = _seq_lrelu_convs(z_hat, p, "h_s", [(0, "subpel", 1), (2, "subpel", 1), (4, "conv", 1)])
    layer = layer_prior_resampler(y_hat_bl, p.sub("layer_prior_resampler"), shape_hr)
    params = prior_fusion(hyper, layer, c3, p.sub("prior_fusion_net"))
    scales, means = params.chunk(2, 1)
    y_hat, y_lik = gaussian_conditional(y, scales, means)
    res_hat = res_decoder(y_hat, c2, c3, p.sub("g_s"), gdn_intra)
    feature, x_hat = recon_generation(res_hat, c1, p.sub("recon_net"))

    out = {"bit_bl": bl["bits"].item(), "bit_el": bits_from_likelihoods(y_lik, z_lik).item(),
           "x_hat_bl": x_hat_bl_full, "x_hat_el": x_hat, "feature_el": feature}
    if extras:
        med = p.sub("entropy_bottleneck")["quantiles"][:, 0, 1].view(1, -1, 1, 1)
        out.update({"y_bl": bl["y"], "z_bl": bl["z"], "y_hat_bl": y_hat_bl, "ctx": (c1, c2, c3), "y": y, "z": z,
                    "scales": scales, "means": means, "y_hat": y_hat,
                    "sym": {"bl_y": bl["y_q"], "bl_z": bl["z_q"], "el_y": torch.round(y - means), "el_z": torch.round(z - med)}})
    return out
