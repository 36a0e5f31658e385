"""LSSVC_extend -- MI355X-native drop-in for the reference's P-frame scalable codec.

Mirrors the surface test.py uses (test.py:553-559, 213, 230-237):
    LSSVC_extend() -> .load_dict(sd) -> .to(device) -> .eval() -> .set_scale_information(...) -> .encode_decode(...)
Reference: src/models/LSSVC_net.py:12-528 (EL, `forward_one_frame` :445-528),
src/models/dmc_net.py:159-488 (BL `DMC.get_inter_layer_information` :421-488),
src/InterModules/lssvc_modules.py, src/InterModules/video_net_component.py.
"""
import torch

from . import hip_ops as ops
from .hip_ops import T
from . import blocks as B
from .intra import _HostModel
from .weights import strip_module_prefix, validate

# (chunk, mask) pairs per step of the 4-step spatial/channel prior (LSSVC_net.py:361-413):
# MASK_OF_CHUNK[step][chunk] = 2x2 position index, 0:(0,0) 1:(0,1) 2:(1,0) 3:(1,1)
MASK_OF_CHUNK = ((0, 1, 2, 3), (3, 2, 1, 0), (2, 3, 0, 1), (1, 0, 3, 2))


class LSSVC_extend(_HostModel):
    def __init__(self):
        super().__init__()
        self._sd = None

    def load_dict(self, pretrained_dict, strict=True):
        """LSSVC.load_dict (LSSVC_net.py:141-149): strip 'module.' and load strictly."""
        sd = strip_module_prefix(dict(pretrained_dict))
        validate(sd, "lssvc_extend")
        self._sd = sd
        if self.device is not None:
            self.to(self.device)

    # ============================================================================ base layer (DMC)
    def _prior_encoder(self, p, y):
        """conv s1 - lrelu - conv s2 - lrelu - conv s2 (dmc_net.py:190-196,230-236; LSSVC_net.py:55-61,90-96)."""
        W = self.W
        t = ops.conv(W, p + ".0", y, act="lrelu")
        t = ops.conv(W, p + ".2", t, stride=2, act="lrelu")
        return ops.conv(W, p + ".4", t, stride=2)

    def _prior_decoder_bl(self, p, z_hat, out=None):
        """convT s2 - lrelu - convT s2 - lrelu - convT s1 (dmc_net.py:198-206,238-246)."""
        W = self.W
        t = ops.conv_t(W, p + ".0", z_hat, 2, act="lrelu")
        t = ops.conv_t(W, p + ".2", t, 2, act="lrelu")
        return ops.conv_t(W, p + ".4", t, 1, out=out)

    def _bl_forward(self, x, ref_frame, ref_feature):
        """DMC.get_inter_layer_information, eval mode (dmc_net.py:421-488). Bit slots 0..3 = y, z, mv_y, mv_z."""
        W, S, p = self.W, self.slots, "base_layer_model"
        est_mv = B.spynet(W, p + ".optic_flow", x, ref_frame)

        # mv_encoder (dmc_net.py:174-188)
        t, e = est_mv, p + ".mv_encoder"
        for base in (0, 4, 8):
            t = ops.conv(W, "%s.%d" % (e, base), t, stride=2)
            t = ops.gdn(W, "%s.%d" % (e, base + 1), t, "inter")
            t = B.res_block(W, "%s.%d" % (e, base + 2), t, start_from_relu=False)
            t = ops.lrelu(t, 0.1)
        mv_y = ops.conv(W, e + ".12", t, stride=2)
        mv_z = self._prior_encoder(p + ".mv_prior_encoder", mv_y)
        mv_z_hat = mv_z.like()
        ops.factorized_quant_bits(mv_z, W.bit_estimator(p + ".bit_estimator_z_mv"), S, 3, z_hat=mv_z_hat)
        mv_scales, mv_means = self._prior_decoder_bl(p + ".mv_prior_decoder", mv_z_hat).chunk(2)
        mv_y_hat = mv_y.like()
        ops.laplace_quant_bits(mv_y, mv_means, mv_scales, S, 2, y_hat=mv_y_hat)

        # mv_decoder (dmc_net.py:208-221)
        d = p + ".mv_decoder"
        t = ops.conv_t(W, d + ".0", mv_y_hat, 2, act="lrelu", slope=0.1)
        t = B.res_block(W, d + ".2", t, start_from_relu=False)
        t = ops.gdn(W, d + ".3", t, "inter", inverse=True)
        t = ops.gdn(W, d + ".5", ops.conv_t(W, d + ".4", t, 2), "inter", inverse=True)
        t = ops.gdn(W, d + ".7", ops.conv_t(W, d + ".6", t, 2), "inter", inverse=True)
        mv_hat = ops.conv_t(W, d + ".8", t, 2)

        # motion_compensation (dmc_net.py:352-368)
        mv2 = ops.resize(mv_hat, mv_hat.H // 2, mv_hat.W // 2, scale=0.5)
        mv3 = ops.resize(mv2, mv2.H // 2, mv2.W // 2, scale=0.5)
        f = ops.conv(W, p + ".feature_adaptor_I", ref_frame) if ref_feature is None \
            else ops.conv(W, p + ".feature_adaptor_P", ref_feature)
        r1, r2, r3 = B.pyramid_extractor(W, p + ".feature_extractor", f)
        c1, c2, c3 = B.context_fusion(W, p + ".context_fusion_net", ops.flow_warp(r1, mv_hat), ops.flow_warp(r2, mv2),
                                      ops.flow_warp(r3, mv3))

        y = B.res_encoder_gdn(W, p + ".res_encoder", x, c1, c2, c3, "inter")
        z = self._prior_encoder(p + ".res_prior_encoder", y)
        z_hat = z.like()
        ops.factorized_quant_bits(z, W.bit_estimator(p + ".bit_estimator_z"), S, 1, z_hat=z_hat)

        # params = cat(temporal 192, hierarchical 192) -> res_entropy_parameter (dmc_net.py:440-445)
        q = p + ".temporal_prior_encoder"
        t = ops.gdn(W, q + ".gdn1", ops.conv(W, q + ".conv1", c1, stride=2), "inter")
        t = ops.gdn(W, q + ".gdn2", ops.conv(W, q + ".conv2", [t, c2], stride=2), "inter")
        t = ops.gdn(W, q + ".gdn3", ops.conv(W, q + ".conv3", [t, c3], stride=2), "inter")
        temporal = ops.conv(W, q + ".conv4", t, stride=2)
        hier = self._prior_decoder_bl(p + ".res_prior_decoder", z_hat)
        q = p + ".res_entropy_parameter"
        t = ops.conv(W, q + ".0", [temporal, hier], act="lrelu")
        t = ops.conv(W, q + ".2", t, act="lrelu")
        scales, means = ops.conv(W, q + ".4", t).chunk(2)
        y_hat = y.like()
        ops.laplace_quant_bits(y, means, scales, S, 0, y_hat=y_hat)

        res = B.res_decoder_gdn(W, p + ".res_decoder", y_hat, c2, c3, "inter")
        feature, recon = B.recon_generation(W, p + ".recon_generation_net", res, c1)
        return {"recon": recon, "feature": feature, "y_hat": y_hat, "mv_hat": mv_hat}

    # ============================================================================ enhancement layer
    def _resampler_tail(self, p, up, out=None):
        """conv2 (conv-lrelu-conv) -> 2 DepthConvBlocks -> + skip (lssvc_modules.py:361-363,394-396,426-428)."""
        W = self.W
        up = ops.conv(W, p + ".conv2.2", ops.conv(W, p + ".conv2.0", up, act="lrelu"))
        ref = B.depth_conv_block(W, p + ".feature_refine.1", B.depth_conv_block(W, p + ".feature_refine.0", up))
        return ops.add(ref, up, out=out)

    def _mv_resampler(self, mv_bl):
        """MvResampler (lssvc_modules.py:339-365); the trailing `s * mv` is the last conv's output scale."""
        W, p = self.W, "mv_resampler"
        f = ops.conv(W, p + ".conv1.2", ops.conv(W, p + ".conv1.0", mv_bl, act="lrelu"))
        f = self._resampler_tail(p, ops.resize(f, *self.shape_hr))
        return ops.conv(W, p + ".recon_conv", f, out_scale=float(self.scale_factor))

    def _texture_resampler(self, tex_bl):
        """LSSVC TextureResampler (lssvc_modules.py:368-397)."""
        W, p = self.W, "texture_resampler"
        which = "base_layer_adaptor" if tex_bl.C == 64 else "enhance_layer_adaptor"
        f = ops.conv(W, p + ".conv_adaptor." + which, tex_bl)
        f = ops.conv(W, p + ".conv1.2", ops.conv(W, p + ".conv1.0", f, act="lrelu"))
        return self._resampler_tail(p, ops.resize(f, *self.shape_hr))

    def _layer_prior_resampler(self, y_hat_bl, out=None):
        """LSSVC LayerPriorResampler (lssvc_modules.py:400-429), target = shape_hr // 16 (LSSVC_net.py:226)."""
        W, p = self.W, "layer_prior_resampler"
        which = "base_layer_adaptor" if y_hat_bl.C == 96 else "enhance_layer_adaptor"
        f = ops.conv(W, p + ".conv_adaptor." + which, y_hat_bl)
        f = ops.conv(W, p + ".conv1.2", ops.conv(W, p + ".conv1.0", f, act="lrelu"))
        return self._resampler_tail(p, ops.resize(f, self.shape_hr[0] // 16, self.shape_hr[1] // 16), out=out)

    def _offset_diversity(self, x, aux, flow):
        """OffsetDiversity (lssvc_modules.py:75-112). `aux` = [context1_init, warpframe, mv] (virtual concat)."""
        W, p = self.W, "align"
        t = ops.conv(W, p + ".conv_offset.0", aux, stride=2, act="lrelu", slope=0.1)
        t = ops.conv(W, p + ".conv_offset.2", t, act="lrelu", slope=0.1)
        t = ops.conv(W, p + ".conv_offset.4", t)
        om = ops.resize(t, t.H * 2, t.W * 2)
        fw = W.vector(p + ".fusion.weight")     # (48, 6, 1, 1) flattened = [48][6]
        fb = W.vector(p + ".fusion.bias")
        return ops.offset_diversity_tail(x, om, flow, fw, fb)

    def _motion_compensation(self, ref, feature_el, mv):
        """LSSVC.motion_compensation + multi_scale_feature_extractor (LSSVC_net.py:195-202,229-244)."""
        W = self.W
        warpframe = ops.flow_warp(ref, mv)
        mv2 = ops.resize(mv, mv.H // 2, mv.W // 2, scale=0.5)
        mv3 = ops.resize(mv2, mv2.H // 2, mv2.W // 2, scale=0.5)
        if feature_el is None:
            f = ops.conv(W, "feature_adaptor_EL_I", ref)
        elif feature_el.C == 64:
            f = ops.conv(W, "feature_adaptor_EL_first_P", feature_el)
        else:
            f = ops.conv(W, "feature_adaptor_EL", feature_el)
        r1, r2, r3 = B.pyramid_extractor(W, "feature_extractor", f)
        c1_init = ops.flow_warp(r1, mv)
        c1 = self._offset_diversity(r1, [c1_init, warpframe, mv], mv)
        c2, c3 = ops.flow_warp(r2, mv2), ops.flow_warp(r3, mv3)
        return B.context_fusion(W, "context_fusion_net", c1, c2, c3), warpframe

    def _el_context(self, texture_bl, mv, ref, feature_el):
        """LSSVC.hybrid_temporal_layer_context_fusion (LSSVC_net.py:246-259)."""
        W = self.W
        temp, warpframe = self._motion_compensation(ref, feature_el, mv)
        if texture_bl is not None:
            spat = B.pyramid_extractor(W, "texture_extractor", self._texture_resampler(texture_bl))
            ctx = []
            for i, g in enumerate(("generator1", "generator2", "generator3")):   # HybridWeightGenerator :115-154
                q = "weight_map_generator." + g
                f = ops.conv(W, q + ".0", [temp[i], spat[i]])
                f = B.res_block(W, q + ".1", f, end_with_relu=True)
                logits = ops.conv(W, q + ".2", f)
                ctx.append(ops.softmax2_blend(temp[i], spat[i], logits))
        else:
            ctx = list(temp)
        c1, c2, c3 = B.context_fusion(W, "context_fusion_net", ctx[0], ctx[1], ctx[2])
        return c1, c2, c3, warpframe

    def _unet(self, p, x):
        """UNet (lssvc_modules.py:295-336)."""
        W = self.W
        x1 = B.depth_conv_block(W, p + ".conv1", x)
        x2 = B.depth_conv_block(W, p + ".conv2", ops.pool2x2(x1, is_max=True))
        x3 = B.depth_conv_block(W, p + ".conv3", ops.pool2x2(x2, is_max=True))
        for i in range(4):
            x3 = B.depth_conv_block(W, "%s.context_refine.%d" % (p, i), x3)
        d3 = B.depth_conv_block(W, p + ".up_conv3", [x2, ops.subpel(W, p + ".up3", x3)])
        return B.depth_conv_block(W, p + ".up_conv2", [x1, ops.subpel(W, p + ".up2", d3)])

    def _four_part_prior(self, y, common):
        """LSSVC.forward_four_part_prior, write=False (LSSVC_net.py:338-443): four masked quantise steps;
        steps 2-4 recompute (sigma, mu) from cat(y_hat_so_far, common_params)."""
        W = self.W
        y_q, y_hat, s_hat = T.zeros(y.H, y.W, y.C, y.device), T.zeros(y.H, y.W, y.C, y.device), T.zeros(y.H, y.W, y.C, y.device)
        scales, means = common.chunk(2)
        for step in range(4):
            if step > 0:
                t = ops.conv(W, "y_spatial_prior_adaptor_%d" % step, [y_hat, common])
                for i in range(3):
                    t = B.depth_conv_block(W, "y_spatial_prior.%d" % i, t)
                scales, means = t.chunk(2)
            ops.four_part_step(y, means, scales, MASK_OF_CHUNK[step], y_q, y_hat, s_hat)
        return y_q, y_hat, s_hat

    def _forward(self, xb, xe, ref_bl, ref_el, feat_bl, feat_el):
        """LSSVC.forward_one_frame (LSSVC_net.py:445-528) on NHWC views. EL bit slots 4..7 = y, mv_y, z, mv_z."""
        W, S = self.W, self.slots
        H, Wd = self.shape_hr
        assert (xe.H, xe.W) == (H, Wd), "x_el is %dx%d but shape_hr is %dx%d" % (xe.H, xe.W, H, Wd)
        bl = self._bl_forward(xb, ref_bl, feat_bl)

        mv_up = self._mv_resampler(bl["mv_hat"])
        # mv_ctx_prior_encoder (LSSVC_net.py:108-116)
        t, e = mv_up, "mv_ctx_prior_encoder"
        for base in (0, 2, 4):
            t = ops.gdn(W, "%s.%d" % (e, base + 1), ops.conv(W, "%s.%d" % (e, base), t, stride=2), "inter")
        mv_ctx_prior = ops.conv(W, e + ".6", t, stride=2)
        mv_ctx = B.res_block(W, "mv_ctx_transform.transform.1", ops.conv(W, "mv_ctx_transform.transform.0", mv_up, stride=2))

        mv = B.spynet(W, "optic_flow", xe, ref_el)
        # MVResEncoder (lssvc_modules.py:445-469)
        e = "mv_encoder.encoder1"
        t = ops.gdn(W, e + ".1", ops.conv(W, e + ".0", mv, stride=2), "inter")
        t = ops.lrelu(B.res_block(W, e + ".2", t, start_from_relu=False), 0.1)
        e = "mv_encoder.encoder2"
        t = ops.gdn(W, e + ".1", ops.conv(W, e + ".0", [t, mv_ctx], stride=2), "inter")
        t = ops.lrelu(B.res_block(W, e + ".2", t, start_from_relu=False), 0.1)
        t = ops.gdn(W, e + ".5", ops.conv(W, e + ".4", t, stride=2), "inter")
        t = ops.lrelu(B.res_block(W, e + ".6", t, start_from_relu=False), 0.1)
        mv_y = ops.conv(W, e + ".8", t, stride=2)
        mv_z = self._prior_encoder("mv_prior_encoder", mv_y)
        mv_z_hat = mv_z.like()
        ops.factorized_quant_bits(mv_z, W.bit_estimator("bit_estimator_z_mv"), S, 7, z_hat=mv_z_hat)
        q = "mv_prior_decoder"
        t = ops.subpel(W, q + ".0", mv_z_hat, act="lrelu")
        t = ops.subpel(W, q + ".2", t, act="lrelu")
        hyper = ops.conv(W, q + ".4", t)
        q = "mv_prior_fusion"
        t = ops.conv(W, q + ".0", [hyper, mv_ctx_prior], act="lrelu")
        t = ops.conv(W, q + ".2", t, act="lrelu")
        mv_scales, mv_means = ops.conv(W, q + ".4", t).chunk(2)
        mv_y_hat = mv_y.like()
        ops.laplace_quant_bits(mv_y, mv_means, mv_scales, S, 5, y_hat=mv_y_hat)
        # MVResDecoder (lssvc_modules.py:472-494)
        d = "mv_decoder.decoder1"
        t = ops.subpel(W, d + ".0", mv_y_hat, act="lrelu", slope=0.1)
        t = B.res_block(W, d + ".2", t, start_from_relu=False)
        t = ops.gdn(W, d + ".3", t, "inter", inverse=True)
        t = ops.gdn(W, d + ".5", ops.subpel(W, d + ".4", t), "inter", inverse=True)
        t = ops.gdn(W, d + ".7", ops.subpel(W, d + ".6", t), "inter", inverse=True)
        d = "mv_decoder.decoder2"
        t = ops.conv(W, d + ".0", [t, mv_ctx], act="lrelu", slope=0.1)
        mv_hat = ops.subpel(W, d + ".2", t)

        c1, c2, c3, warp_frame = self._el_context(bl["feature"], mv_hat, ref_el, feat_el)

        # ResEncoder without GDN (lssvc_modules.py:235-254); the concat feeding each ResBlock is built in place
        p = "res_encoder"
        t = T.empty(H // 2, Wd // 2, 64 + c2.C, self.device)
        ops.conv(W, p + ".conv1", [xe, c1], stride=2, out=t.slice(0, 64))
        ops.copy(c2, t.slice(64, t.C))
        t = B.res_block(W, p + ".res1", t, slope=0.1, end_with_relu=True)
        u = T.empty(H // 4, Wd // 4, 96 + c3.C, self.device)
        ops.conv(W, p + ".conv2", t, stride=2, out=u.slice(0, 96))
        ops.copy(c3, u.slice(96, u.C))
        u = B.res_block(W, p + ".res2", u, slope=0.1, end_with_relu=True)
        y = ops.conv(W, p + ".conv4", ops.conv(W, p + ".conv3", u, stride=2), stride=2)

        z = self._prior_encoder("res_prior_encoder", y)
        z_hat = z.like()
        ops.factorized_quant_bits(z, W.bit_estimator("bit_estimator_z"), S, 6, z_hat=z_hat)

        # prior fusion input cat(hyper 128, temporal 128, layer 128) written in place (lssvc_modules.py:440-442)
        fused = T.empty(y.H, y.W, 384, self.device)
        q = "res_prior_decoder"                                     # LSSVC_net.py:63-73
        t = ops.conv(W, q + ".0", z_hat, act="lrelu")
        t = ops.subpel(W, q + ".2", t, act="lrelu")
        t = ops.conv(W, q + ".4", t, act="lrelu")
        t = ops.subpel(W, q + ".6", t, act="lrelu")
        ops.conv(W, q + ".8", t, out=fused.slice(0, 128))
        q = "temporal_prior_encoder"                                # LSSVC_net.py:75-79
        t = ops.conv(W, q + ".0", c3, stride=2, act="lrelu", slope=0.1)
        ops.conv(W, q + ".2", t, stride=2, out=fused.slice(128, 256))
        self._layer_prior_resampler(bl["y_hat"], out=fused.slice(256, 384))
        params = B.depth_conv_block(W, "prior_fusion_net.prior_fusion_conv.1",
                                    B.depth_conv_block(W, "prior_fusion_net.prior_fusion_conv.0", fused))
        y_q, y_hat, scales_hat = self._four_part_prior(y, params)
        ops.laplace_bits(y_q, scales_hat, S, 4)

        # ResDecoder (lssvc_modules.py:257-276)
        p = "res_decoder"
        t = ops.subpel(W, p + ".up1", y_hat)
        u = T.empty(H // 4, Wd // 4, 96 + c3.C, self.device)
        ops.subpel(W, p + ".up2", t, out=u.slice(0, 96))
        ops.copy(c3, u.slice(96, u.C))
        u = B.res_block(W, p + ".res1", u, slope=0.1, end_with_relu=True)
        t = T.empty(H // 2, Wd // 2, 64 + c2.C, self.device)
        ops.subpel(W, p + ".up3", u, out=t.slice(0, 64))
        ops.copy(c2, t.slice(64, t.C))
        t = B.res_block(W, p + ".res2", t, slope=0.1, end_with_relu=True)
        res = ops.subpel(W, p + ".up4", t)

        # ReconGeneration (lssvc_modules.py:279-292), called as (recon_image_feature, context1) (LSSVC_net.py:492)
        p = "recon_generation_net"
        f = ops.conv(W, p + ".first_conv", [res, c1])
        feature = self._unet(p + ".unet_2", self._unet(p + ".unet_1", f))
        recon_el = ops.conv(W, p + ".recon_conv", feature)
        return bl, feature, recon_el, mv_hat, warp_frame

    # ---------------------------------------------------------------------------------------------
    def forward_one_frame(self, x_bl, x_el, ref_frame_bl, ref_frame_el, ref_feature_bl, ref_feature_el):
        self._require_device()
        nhwc = lambda t: None if t is None else T.from_nchw(t)
        bl, feature, recon_el, mv_hat, warp_frame = self._forward(
            nhwc(x_bl), nhwc(x_el), nhwc(ref_frame_bl), nhwc(ref_frame_el), nhwc(ref_feature_bl), nhwc(ref_feature_el))
        dpb = {"ref_frame_bl": bl["recon"].to_nchw(), "ref_feature_bl": bl["feature"].to_nchw(),
               "ref_frame_el": recon_el.to_nchw(), "ref_feature_el": feature.to_nchw()}
        out = {"dpb": dpb, "mv_hat": mv_hat.to_nchw(), "warp_frame": warp_frame.to_nchw(),
               "encoding_time_EL": 0.0, "decoding_time_EL": 0.0, "encoding_time_BL": 0.0, "decoding_time_BL": 0.0}
        s = self.slots.fetch()
        out["bit_bl"] = s[0] + s[1] + s[2] + s[3]          # y + z + mv_y + mv_z  (dmc_net.py:473)
        out["bit_el"] = s[4] + s[5] + s[6] + s[7]          # y + mv_y + z + mv_z  (LSSVC_net.py:508)
        return out

    def encode_decode(self, x_bl, x_el, dpb, output_path_bl=None, output_path_el=None,
                      pic_width=None, pic_height=None, pic_width_bl=None, pic_height_bl=None):
        """LSSVC.encode_decode (LSSVC_net.py:172-185). output_path_el None <=> estimate mode."""
        if output_path_el is not None:
            raise NotImplementedError("write_stream=1 (real bitstream) is not built yet in lssvc_amd; use estimate mode")
        return self.forward_one_frame(x_bl, x_el, dpb["ref_frame_bl"], dpb["ref_frame_el"], dpb["ref_feature_bl"],
                                      dpb["ref_feature_el"])

    def update(self, force=False):
        raise NotImplementedError("update() builds CDF tables for write_stream=1, which is not built yet")
