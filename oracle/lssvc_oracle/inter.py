"""ORACLE (test infrastructure only) -- P-frame path: LSSVC.forward_one_frame in estimate mode
(LSSVC_net.py:445-528) with its BL codec DMC.get_inter_layer_information (dmc_net.py:421-488).
See blocks.py for the rules this package follows.
"""
import torch
import torch.nn.functional as F

from .blocks import (Params, conv, conv_t, subpel, lrelu, res_block, gdn_inter, depth_conv_block, bilinear, down2,
                     up2, flow_warp, spynet)
from .entropy import laplace_bits, factorized_bits
from .intra import context_fusion, res_encoder, res_decoder, recon_generation, texture_extractor


# ============================================================================= base layer (DMC)
def _mv_encoder_bl(mv, p):
    """DMC.mv_encoder (dmc_net.py:174-188)."""
    x = mv
    for base in (0, 4, 8):
        x = conv(x, p, str(base), stride=2)
        x = gdn_inter(x, p, str(base + 1))
        x = res_block(x, p, str(base + 2), start_from_relu=False)
        x = lrelu(x, 0.1)
    return conv(x, p, "12", stride=2)


def _mv_decoder_bl(y, p):
    """DMC.mv_decoder (dmc_net.py:208-221)."""
    x = lrelu(conv_t(y, p, "0", 2, 1), 0.1)
    x = res_block(x, p, "2", start_from_relu=False)
    x = gdn_inter(x, p, "3", inverse=True)
    x = gdn_inter(conv_t(x, p, "4", 2, 1), p, "5", inverse=True)
    x = gdn_inter(conv_t(x, p, "6", 2, 1), p, "7", inverse=True)
    return conv_t(x, p, "8", 2, 1)


def _prior_encoder(y, p):
    """conv s1 -> lrelu -> conv s2 -> lrelu -> conv s2 (dmc_net.py:190-196,230-236; LSSVC_net.py:55-61,90-96)."""
    x = lrelu(conv(y, p, "0"))
    x = lrelu(conv(x, p, "2", stride=2))
    return conv(x, p, "4", stride=2)


def _prior_decoder_bl(z_hat, p):
    """convT s2 -> lrelu -> convT s2 -> lrelu -> convT s1 (dmc_net.py:198-206,238-246)."""
    x = lrelu(conv_t(z_hat, p, "0", 2, 1))
    x = lrelu(conv_t(x, p, "2", 2, 1))
    return conv_t(x, p, "4", 1, 0)


def _temporal_prior_encoder_bl(c1, c2, c3, p):
    """TemporalPriorEncoder (dmc_net.py:121-140)."""
    f = gdn_inter(conv(c1, p, "conv1", stride=2), p, "gdn1")
    f = gdn_inter(conv(torch.cat([f, c2], 1), p, "conv2", stride=2), p, "gdn2")
    f = gdn_inter(conv(torch.cat([f, c3], 1), p, "conv3", stride=2), p, "gdn3")
    return conv(f, p, "conv4", stride=2)


def bl_motion_compensation(ref, feature, mv, p):
    """DMC.motion_compensation (dmc_net.py:352-368)."""
    mv2 = down2(mv) / 2
    mv3 = down2(mv2) / 2
    f = conv(ref, p, "feature_adaptor_I") if feature is None else conv(feature, p, "feature_adaptor_P")
    r1, r2, r3 = texture_extractor(f, p.sub("feature_extractor"))     # same dataflow (dmc_net.py:11-31)
    c1, c2, c3 = flow_warp(r1, mv), flow_warp(r2, mv2), flow_warp(r3, mv3)
    return context_fusion(c1, c2, c3, p.sub("context_fusion_net"))


def bl_inter_layer_information(x, ref_frame, ref_feature, p):
    """DMC.get_inter_layer_information in eval mode (dmc_net.py:421-488)."""
    est_mv = spynet(x, ref_frame, p, "optic_flow")
    mv_y = _mv_encoder_bl(est_mv, p.sub("mv_encoder"))
    mv_z = _prior_encoder(mv_y, p.sub("mv_prior_encoder"))
    mv_z_hat = torch.round(mv_z)
    mv_scales, mv_means = _prior_decoder_bl(mv_z_hat, p.sub("mv_prior_decoder")).chunk(2, 1)
    mv_y_q = torch.round(mv_y - mv_means)
    mv_y_hat = mv_y_q + mv_means
    mv_hat = _mv_decoder_bl(mv_y_hat, p.sub("mv_decoder"))

    c1, c2, c3 = bl_motion_compensation(ref_frame, ref_feature, mv_hat, p)
    y = res_encoder(x, c1, c2, c3, p.sub("res_encoder"), gdn_inter)
    z = _prior_encoder(y, p.sub("res_prior_encoder"))
    z_hat = torch.round(z)
    hier = _prior_decoder_bl(z_hat, p.sub("res_prior_decoder"))
    temporal = _temporal_prior_encoder_bl(c1, c2, c3, p.sub("temporal_prior_encoder"))
    q = p.sub("res_entropy_parameter")
    g = lrelu(conv(torch.cat((temporal, hier), dim=1), q, "0"))
    g = conv(lrelu(conv(g, q, "2")), q, "4")
    scales, means = g.chunk(2, 1)
    y_q = torch.round(y - means)
    y_hat = y_q + means

    res = res_decoder(y_hat, c2, c3, p.sub("res_decoder"), gdn_inter)
    feature, recon = recon_generation(res, c1, p.sub("recon_generation_net"))

    bits = (laplace_bits(y_q, scales) + factorized_bits(z_hat, p.sub("bit_estimator_z"))
            + laplace_bits(mv_y_q, mv_scales) + factorized_bits(mv_z_hat, p.sub("bit_estimator_z_mv")))
    return {"bits": bits, "recon_image": recon, "feature": feature, "y_hat": y_hat, "mv_hat": mv_hat,
            "y_q": y_q, "z_hat": z_hat, "mv_y_q": mv_y_q, "mv_z_hat": mv_z_hat, "scales": scales,
            "mv_scales": mv_scales, "est_mv": est_mv}


# ============================================================================= enhancement layer
def _resampler_tail(up, p):
    """conv2 (conv-lrelu-conv) -> 2 DepthConvBlocks, + skip (lssvc_modules.py:361-363,394-396,426-428)."""
    up = conv(lrelu(conv(up, p, "conv2.0")), p, "conv2.2")
    ref = depth_conv_block(depth_conv_block(up, p, "feature_refine.0"), p, "feature_refine.1")
    return ref + up


def mv_resampler(mv_bl, p, shape_hr, s):
    """MvResampler (lssvc_modules.py:339-365)."""
    f = conv(lrelu(conv(mv_bl, p, "conv1.0")), p, "conv1.2")
    f = _resampler_tail(bilinear(f, shape_hr), p)
    return s * conv(f, p, "recon_conv")


def texture_resampler(tex_bl, p, shape_hr):
    """LSSVC TextureResampler (lssvc_modules.py:368-397); adaptor picked by channel count."""
    which = "base_layer_adaptor" if tex_bl.shape[1] == 64 else "enhance_layer_adaptor"
    f = conv(tex_bl, p, "conv_adaptor." + which)
    f = conv(lrelu(conv(f, p, "conv1.0")), p, "conv1.2")
    return _resampler_tail(bilinear(f, shape_hr), p)


def layer_prior_resampler(y_hat_bl, p, shape):
    """LSSVC LayerPriorResampler (lssvc_modules.py:400-429); `shape` is already shape_hr//16."""
    which = "base_layer_adaptor" if y_hat_bl.shape[1] == 96 else "enhance_layer_adaptor"
    f = conv(y_hat_bl, p, "conv_adaptor." + which)
    f = conv(lrelu(conv(f, p, "conv1.0")), p, "conv1.2")
    return _resampler_tail(bilinear(f, shape), p)


def _mv_ctx_prior_encoder(mv_up, p):
    """LSSVC.mv_ctx_prior_encoder (LSSVC_net.py:108-116)."""
    x = mv_up
    for base in (0, 2, 4):
        x = gdn_inter(conv(x, p, str(base), stride=2), p, str(base + 1))
    return conv(x, p, "6", stride=2)


def _mv_res_encoder(mv, mv_ctx, p):
    """MVResEncoder (lssvc_modules.py:445-469)."""
    e = p.sub("encoder1")
    f = gdn_inter(conv(mv, e, "0", stride=2), e, "1")
    f = lrelu(res_block(f, e, "2", start_from_relu=False), 0.1)
    e = p.sub("encoder2")
    x = torch.cat([f, mv_ctx], dim=1)
    for base in (0, 4):
        x = gdn_inter(conv(x, e, str(base), stride=2), e, str(base + 1))
        x = lrelu(res_block(x, e, str(base + 2), start_from_relu=False), 0.1)
    return conv(x, e, "8", stride=2)


def _mv_res_decoder(mv_y_hat, mv_ctx, p):
    """MVResDecoder (lssvc_modules.py:472-494)."""
    d = p.sub("decoder1")
    x = lrelu(subpel(mv_y_hat, d, "0"), 0.1)
    x = res_block(x, d, "2", start_from_relu=False)
    x = gdn_inter(x, d, "3", inverse=True)
    x = gdn_inter(subpel(x, d, "4"), d, "5", inverse=True)
    x = gdn_inter(subpel(x, d, "6"), d, "7", inverse=True)
    d = p.sub("decoder2")
    x = lrelu(conv(torch.cat([x, mv_ctx], dim=1), d, "0"), 0.1)
    return subpel(x, d, "2")


def offset_diversity(x, aux, flow, p):
    """OffsetDiversity (lssvc_modules.py:75-112): 2 offset sets x 16 groups of 3 channels, warped by
    40*tanh(offset)+flow, masked by sigmoid, fused by a grouped 1x1 conv. The view() calls are
    restated literally so the (set, group) channel interleave of the reference is preserved."""
    b, c, h, w = x.shape
    groups, sets = 16, 2
    out = conv(aux, p, "conv_offset.0", stride=2)
    out = conv(lrelu(out, 0.1), p, "conv_offset.2")
    out = conv(lrelu(out, 0.1), p, "conv_offset.4")
    out = up2(out)
    o1, o2, mask = torch.chunk(out, 3, dim=1)
    mask = torch.sigmoid(mask)
    offset = 40 * torch.tanh(torch.cat((o1, o2), dim=1))
    offset = offset + flow.repeat(1, groups * sets, 1, 1)
    offset = offset.view(b * groups * sets, 2, h, w)
    mask = mask.view(b * groups * sets, 1, h, w)
    xs = x.view(b * groups, c // groups, h, w).repeat(sets, 1, 1, 1)
    xs = flow_warp(xs, offset) * mask
    xs = xs.view(b, c * sets, h, w)
    return F.conv2d(xs, p["fusion.weight"], p["fusion.bias"], groups=groups)


def _feature_extractor_el(f, p):
    """LSSVC FeatureExtractor / TextureExtractor (lssvc_modules.py:157-200): 48/64/96-channel pyramid."""
    return texture_extractor(f, p)


def el_motion_compensation(ref, feature_el, mv, p):
    """LSSVC.motion_compensation (LSSVC_net.py:229-244) with multi_scale_feature_extractor (:195-202)."""
    warpframe = flow_warp(ref, mv)
    mv2 = down2(mv) / 2
    mv3 = down2(mv2) / 2
    if feature_el is None:
        f = conv(ref, p, "feature_adaptor_EL_I")
    elif feature_el.shape[1] == 64:
        f = conv(feature_el, p, "feature_adaptor_EL_first_P")
    else:
        f = conv(feature_el, p, "feature_adaptor_EL")
    r1, r2, r3 = _feature_extractor_el(f, p.sub("feature_extractor"))
    c1_init = flow_warp(r1, mv)
    c1 = offset_diversity(r1, torch.cat((c1_init, warpframe, mv), dim=1), mv, p.sub("align"))
    c2, c3 = flow_warp(r2, mv2), flow_warp(r3, mv3)
    return context_fusion(c1, c2, c3, p.sub("context_fusion_net")), warpframe


def _weight_maps(temp, spat, p):
    """HybridWeightGenerator (lssvc_modules.py:115-154): softmax over 2 channels per scale."""
    maps = []
    for i, name in enumerate(("generator1", "generator2", "generator3")):
        q = p.sub(name)
        f = conv(torch.cat([temp[i], spat[i]], dim=1), q, "0")
        f = res_block(f, q, "1", end_with_relu=True)
        maps.append(torch.softmax(conv(f, q, "2"), dim=1).chunk(2, 1))
    return maps


def el_context(texture_bl, mv, ref, feature_el, p):
    """LSSVC.hybrid_temporal_layer_context_fusion (LSSVC_net.py:246-259). Note context_fusion_net
    runs a second time on the blended contexts (:258)."""
    temp, warpframe = el_motion_compensation(ref, feature_el, mv, p)
    if texture_bl is not None:
        tex = texture_resampler(texture_bl, p.sub("texture_resampler"), p.shape_hr)
        spat = _feature_extractor_el(tex, p.sub("texture_extractor"))
        maps = _weight_maps(temp, spat, p.sub("weight_map_generator"))
        ctx = [temp[i] * maps[i][0] + spat[i] * maps[i][1] for i in range(3)]
    else:
        ctx = list(temp)
    c1, c2, c3 = context_fusion(ctx[0], ctx[1], ctx[2], p.sub("context_fusion_net"))
    return c1, c2, c3, warpframe


def _res_encoder_el(x, c1, c2, c3, p):
    """LSSVC ResEncoder, no GDN (lssvc_modules.py:235-254). res blocks start from relu (default)."""
    f = conv(torch.cat([x, c1], 1), p, "conv1", stride=2)
    f = res_block(torch.cat([f, c2], 1), p, "res1", slope=0.1, end_with_relu=True)
    f = conv(f, p, "conv2", stride=2)
    f = res_block(torch.cat([f, c3], 1), p, "res2", slope=0.1, end_with_relu=True)
    return conv(conv(f, p, "conv3", stride=2), p, "conv4", stride=2)


def _res_decoder_el(y_hat, c2, c3, p):
    """LSSVC ResDecoder (lssvc_modules.py:257-276)."""
    f = subpel(subpel(y_hat, p, "up1"), p, "up2")
    f = res_block(torch.cat([f, c3], 1), p, "res1", slope=0.1, end_with_relu=True)
    f = subpel(f, p, "up3")
    f = res_block(torch.cat([f, c2], 1), p, "res2", slope=0.1, end_with_relu=True)
    return subpel(f, p, "up4")


def _unet(x, p):
    """UNet of DepthConvBlocks (lssvc_modules.py:295-336)."""
    x1 = depth_conv_block(x, p, "conv1")
    x2 = depth_conv_block(F.max_pool2d(x1, 2, 2), p, "conv2")
    x3 = depth_conv_block(F.max_pool2d(x2, 2, 2), p, "conv3")
    for i in range(4):
        x3 = depth_conv_block(x3, p, "context_refine.%d" % i)
    d3 = depth_conv_block(torch.cat((x2, subpel(x3, p, "up3")), dim=1), p, "up_conv3")
    return depth_conv_block(torch.cat((x1, subpel(d3, p, "up2")), dim=1), p, "up_conv2")


def _recon_generation_el(res, ctx1, p):
    """LSSVC ReconGeneration called as (recon_image_feature, context1) (lssvc_modules.py:279-292, LSSVC_net.py:492)."""
    f = conv(torch.cat((res, ctx1), dim=1), p, "first_conv")
    f = _unet(_unet(f, p.sub("unet_1")), p.sub("unet_2"))
    return f, conv(f, p, "recon_conv")


def _res_prior_decoder_el(z_hat, p):
    """LSSVC.res_prior_decoder (LSSVC_net.py:63-73): conv, subpel1x1, conv, subpel1x1, conv with lrelu between."""
    x = lrelu(conv(z_hat, p, "0"))
    x = lrelu(subpel(x, p, "2"))
    x = lrelu(conv(x, p, "4"))
    x = lrelu(subpel(x, p, "6"))
    return conv(x, p, "8")


# (chunk c, mask m) pairs per step; mask m = 2x2 position (0,0),(0,1),(1,0),(1,1)  (LSSVC_net.py:361-413)
FOUR_PART_SCHEDULE = (((0, 0), (1, 1), (2, 2), (3, 3)),
                      ((0, 3), (1, 2), (2, 1), (3, 0)),
                      ((0, 2), (1, 3), (2, 0), (3, 1)),
                      ((0, 1), (1, 0), (2, 3), (3, 2)))


def four_part_prior(y, common_params, p):
    """LSSVC.forward_four_part_prior (LSSVC_net.py:338-443), write=False branch."""
    _, _, h, w = y.shape
    masks = []
    for (r, c) in ((0, 0), (0, 1), (1, 0), (1, 1)):
        m = torch.zeros(1, 1, h, w, dtype=y.dtype)
        m[:, :, r::2, c::2] = 1
        masks.append(m)
    y_c = y.chunk(4, 1)
    zero = lambda: [torch.zeros_like(y_c[0]) for _ in range(4)]
    y_res, y_q, y_hat, s_hat = zero(), zero(), zero(), zero()
    scales, means = common_params.chunk(2, 1)
    sc, mn = scales.chunk(4, 1), means.chunk(4, 1)
    y_hat_so_far = None
    for step, pairs in enumerate(FOUR_PART_SCHEDULE):
        if step > 0:
            params = torch.cat((y_hat_so_far, common_params), dim=1)
            t = conv(params, p, "y_spatial_prior_adaptor_%d" % step)
            for i in range(3):
                t = depth_conv_block(t, p, "y_spatial_prior.%d" % i)
            parts = t.chunk(8, 1)
            sc, mn = parts[:4], parts[4:]
        cur = []
        for (c, m) in pairs:
            mask = masks[m]
            s_m, m_m = sc[c] * mask, mn[c] * mask
            r_ = (y_c[c] - m_m) * mask
            q_ = torch.round(r_)
            h_ = q_ + m_m
            y_res[c] = y_res[c] + r_
            y_q[c] = y_q[c] + q_
            y_hat[c] = y_hat[c] + h_
            s_hat[c] = s_hat[c] + s_m
            cur.append(h_)
        cur = torch.cat(cur, dim=1)
        y_hat_so_far = cur if y_hat_so_far is None else y_hat_so_far + cur
    cat = lambda parts: torch.cat(parts, dim=1)
    return cat(y_res), cat(y_q), cat(y_hat), cat(s_hat)


class _ELParams(Params):
    shape_hr = None


def inter_forward(sd, x_bl, x_el, dpb, shape_hr, scale, extras=False, pad_size=(0, 0, 0, 0)):
    """LSSVC.forward_one_frame (LSSVC_net.py:445-528); pad_size=(0,0,0,0) is what test.py:213 always passes."""
    p = _ELParams(sd)
    p.shape_hr = tuple(shape_hr)
    bl = bl_inter_layer_information(x_bl, dpb["ref_frame_bl"], dpb["ref_feature_bl"], p.sub("base_layer_model"))
    feature_bl, mv_bl_hat, y_bl_hat = bl["feature"], bl["mv_hat"], bl["y_hat"]
    from .intra import depad                                                   # ILP, LSSVC_net.py:454-456
    texture_bl, mv_bl_hat, y_bl_hat = depad(feature_bl, pad_size), depad(mv_bl_hat, pad_size), depad(y_bl_hat, pad_size, 16)

    mv_up = mv_resampler(mv_bl_hat, p.sub("mv_resampler"), shape_hr, scale)
    mv_ctx_prior = _mv_ctx_prior_encoder(mv_up, p.sub("mv_ctx_prior_encoder"))
    t = p.sub("mv_ctx_transform.transform")
    mv_ctx = res_block(conv(mv_up, t, "0", stride=2), t, "1")

    mv = spynet(x_el, dpb["ref_frame_el"], p, "optic_flow")
    mv_y = _mv_res_encoder(mv, mv_ctx, p.sub("mv_encoder"))
    mv_z = _prior_encoder(mv_y, p.sub("mv_prior_encoder"))
    mv_z_hat = torch.round(mv_z)
    q = p.sub("mv_prior_decoder")
    hyper = conv(lrelu(subpel(lrelu(subpel(mv_z_hat, q, "0")), q, "2")), q, "4")
    q = p.sub("mv_prior_fusion")
    g = lrelu(conv(torch.cat([hyper, mv_ctx_prior], dim=1), q, "0"))
    g = conv(lrelu(conv(g, q, "2")), q, "4")
    mv_scales, mv_means = g.chunk(2, 1)
    mv_y_q = torch.round(mv_y - mv_means)
    mv_y_hat = mv_y_q + mv_means
    mv_hat = _mv_res_decoder(mv_y_hat, mv_ctx, p.sub("mv_decoder"))

    c1, c2, c3, warp_frame = el_context(texture_bl, mv_hat, dpb["ref_frame_el"], dpb["ref_feature_el"], p)

    y = _res_encoder_el(x_el, c1, c2, c3, p.sub("res_encoder"))
    z = _prior_encoder(y, p.sub("res_prior_encoder"))
    z_hat = torch.round(z)
    hier = _res_prior_decoder_el(z_hat, p.sub("res_prior_decoder"))
    q = p.sub("temporal_prior_encoder")
    temporal = conv(lrelu(conv(c3, q, "0", stride=2), 0.1), q, "2", stride=2)
    layer_prior = layer_prior_resampler(y_bl_hat, p.sub("layer_prior_resampler"), (shape_hr[0] // 16, shape_hr[1] // 16))
    q = p.sub("prior_fusion_net")
    params = torch.cat([hier, temporal, layer_prior], dim=1)
    params = depth_conv_block(depth_conv_block(params, q, "prior_fusion_conv.0"), q, "prior_fusion_conv.1")
    y_res, y_q, y_hat, scales_hat = four_part_prior(y, params, p)

    res = _res_decoder_el(y_hat, c2, c3, p.sub("res_decoder"))
    feature, recon_el = _recon_generation_el(res, c1, p.sub("recon_generation_net"))

    bits_el = (laplace_bits(y_q, scales_hat) + laplace_bits(mv_y_q, mv_scales)
               + factorized_bits(z_hat, p.sub("bit_estimator_z")) + factorized_bits(mv_z_hat, p.sub("bit_estimator_z_mv")))
    out = {"dpb": {"ref_frame_bl": bl["recon_image"], "ref_feature_bl": feature_bl, "ref_frame_el": recon_el,
                   "ref_feature_el": feature},
           "bit_bl": bl["bits"].item(), "bit_el": bits_el.item(), "mv_hat": mv_hat, "warp_frame": warp_frame}
    if extras:
        out.update({"bl": bl, "mv_up": mv_up, "mv": mv, "y": y, "y_q": y_q, "y_hat": y_hat, "scales_hat": scales_hat,
                    "mv_y_q": mv_y_q, "mv_scales": mv_scales, "z_hat": z_hat, "mv_z_hat": mv_z_hat, "ctx": (c1, c2, c3),
                    "params": params, "pre_round": {"mv_z": mv_z, "mv_y": mv_y - mv_means, "z": z, "y": y_res},
                    "sym": {"bl_y": bl["y_q"], "bl_mv_y": bl["mv_y_q"], "bl_z": bl["z_hat"], "bl_mv_z": bl["mv_z_hat"],
                            "el_y": y_q, "el_mv_y": mv_y_q, "el_z": z_hat, "el_mv_z": mv_z_hat}})
    return out
