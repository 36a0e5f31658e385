"""LSSVC_extend -- MI355X-native drop-in for the reference's P-frame scalable codec.

Mirrors the surface test.py uses (test.py:553-559, 213, 230-237):
    LSSVC_extend() -> .load_dict(sd) -> .to(device) -> .eval() -> .set_scale_information(...) -> .encode_decode(...)
Reference: src/models/LSSVC_net.py:12-528 (EL, `forward_one_frame` :445-528),
src/models/dmc_net.py:159-488 (BL `DMC.get_inter_layer_information` :421-488),
src/InterModules/lssvc_modules.py, src/InterModules/video_net_component.py.
"""
import os
import time

import torch

from . import hip_ops as ops
from .hip_ops import T
from . import blocks as B
from . import bitstream, tables
from .entropy_coder import SymbolSink, SymbolSource
from .intra import _HostModel, _channel_indexes, LAPLACE_IDX
from .weights import strip_module_prefix, validate

EL_BIND = os.environ.get("LSSVC_EL_BIND", "1") == "1"

# (chunk, mask) pairs per step of the 4-step spatial/channel prior (LSSVC_net.py:361-413):
# MASK_OF_CHUNK[step][chunk] = 2x2 position index, 0:(0,0) 1:(0,1) 2:(1,0) 3:(1,1)
MASK_OF_CHUNK = ((0, 1, 2, 3), (3, 2, 1, 0), (2, 3, 0, 1), (1, 0, 3, 2))
# inverse view used by the folded symbol planes: CHUNK_OF_MASK[step][mask] = channel chunk coded at that 2x2 position
CHUNK_OF_MASK = tuple(tuple(m.index(k) for k in range(4)) for m in MASK_OF_CHUNK)


class LSSVC_extend(_HostModel):
    def __init__(self):
        super().__init__()
        self._sd = None
        self._ahead = None             # look-ahead protocol: {"for": frame id, "geom": ..., "bits": its four BL bit counts} of the base layer coded ahead
        self._ahead_stream = None      # the stream BL(t+1) is launched on
        # Persistent buffers of the look-ahead plans, PER GEOMETRY (BL size, EL size, scale, padding, conv precision: everything that
        # fixes their shapes and the launch sequences that have their addresses baked in) and by frame parity: the base layer coded
        # ahead (STASH_KEYS) and the EL reconstruction + feature. One model object serves every dataset and ratio of a harness run
        # (harness._load_nets), so a second size must get its own buffers AND its own plans; at most MAX_GEOMS are kept.
        self._lookahead_bufs = {}      # geom -> {"stash": [None, None], "el_out": [None, None]}
        self._geom = None              # the geometry of the frame being issued

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

    def _pull_factorized(self, source, table, c, h, w):
        """BitEstimator.decode_stream (video_entropy_models.py:240-245): z_hat = decoded symbols."""
        return self._pull(source, table, T.empty(h, w, c, self.device))

    def _pull_laplace(self, source, scales, means, chunk_of_mask=None, out=None):
        """GaussianEncoder.decode_stream (video_entropy_models.py:321-326) + the `+ means` that follows it."""
        if out is None:
            out = T.empty(scales.H, scales.W, scales.C, self.device)
        return self._pull(source, self._tables["laplace"], out, sigma=scales, idx_params=LAPLACE_IDX, mean=means, chunk_of_mask=chunk_of_mask)

    def _bl_codec(self, x, ref_frame, ref_feature, sink=None, source=None, fk=None, slot_base=0):
        """DMC base layer in one of two roles sharing every decoder-side kernel:
        encoder (x given): get_inter_layer_information (dmc_net.py:421-488) / DMCExtend.compress
        (dmc_net_extend.py:55-107, symbols pushed to `sink` in the order mv_z, mv_y, z, y);
        decoder (source given): DMCExtend.decompress (dmc_net_extend.py:109-146).
        Estimate-mode bit slots slot_base + 0..3 = y, z, mv_y, mv_z (slot_base 8: the look-ahead base layer, _frame_body_ahead).
        fk: the frame's ops.Fork -- chains that do not depend on the motion-vector codec (the feature pyramid of the
        reference, later the temporal prior) are issued as parallel branches (side streams 1 and 2)."""
        W, S, p = self.W, self.slots, "base_layer_model"
        tb = self._tables
        decoding = source is not None
        fk = fk if fk is not None else ops.Fork(self.device, enabled=False)
        with fk.branch(1):                      # needs only the DPB: runs beside ME + the whole MV codec
            f = ops.conv(W, p + ".feature_adaptor_I", ref_frame) if ref_feature is None \
                else ops.conv(W, p + ".feature_adaptor_P", ref_feature)
            r1, r2, r3 = B.pyramid_extractor(W, p + ".feature_extractor", f)
        if not decoding:
            est_mv = B.spynet(W, p + ".optic_flow", x, ref_frame)
            # mv_encoder (dmc_net.py:174-188)
            t, e = est_mv, p + ".mv_encoder"
            # (the LeakyReLU(0.1) behind each ResBlock, dmc_net.py:178/182/186, is applied by the conv that reads the tensor -- its
            # only reader -- while staging it: one launch less per stage, the same function of the same values)
            for base in (0, 4, 8):
                t = ops.conv(W, "%s.%d" % (e, base), t, stride=2, **(dict(in_act="lrelu", in_slope=0.1) if base else {}))
                t = ops.gdn(W, "%s.%d" % (e, base + 1), t, "inter")
                t = B.res_block(W, "%s.%d" % (e, base + 2), t, start_from_relu=False)
            mv_y = ops.conv(W, e + ".12", t, stride=2, in_act="lrelu", in_slope=0.1)
            mv_z = self._prior_encoder(p + ".mv_prior_encoder", mv_y)
            mv_z_hat = mv_z.like()
            ops.factorized_quant_bits(mv_z, W.bit_estimator(p + ".bit_estimator_z_mv"), S, slot_base + 3, z_hat=mv_z_hat)
            self._tap("bl_mv_z", mv_z_hat)
            if sink:
                self._push(sink, mv_z_hat, None, tb["bl_z_mv"])
        else:
            zh, zw = bitstream.get_downsampled_shape(ref_frame.H, ref_frame.W, 64)
            mv_z_hat = self._pull_factorized(source, tb["bl_z_mv"], 64, zh, zw)
        mv_scales, mv_means = self._prior_decoder_bl(p + ".mv_prior_decoder", mv_z_hat).chunk(2)
        if not decoding:
            mv_y_hat = mv_y.like()
            mv_y_q = mv_y.like() if (sink or self.taps is not None) else None
            ops.laplace_quant_bits(mv_y, mv_means, mv_scales, S, slot_base + 2, y_q=mv_y_q, y_hat=mv_y_hat)
            self._tap("bl_mv_y", mv_y_q)
            if sink:
                self._push(sink, mv_y_q, mv_scales, tb["laplace"], LAPLACE_IDX)
        else:
            mv_y_hat = self._pull_laplace(source, mv_scales, mv_means)

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
        fk.join(1)
        cf = p + ".context_fusion_net"
        homes = B.context_homes(W, p + ".res_encoder", ref_frame.H // 2, ref_frame.W // 2, W.raw(cf + ".conv2_out.weight").shape[0],
                                W.raw(cf + ".conv3_out.weight").shape[0], self.device)      # contexts 2 / 3 are made where the bottlenecks read them
        c1, c2, c3 = B.context_fusion(W, cf, ops.flow_warp(r1, mv_hat), ops.flow_warp(r2, mv2), ops.flow_warp(r3, mv3),
                                      outs=(None, homes[2], homes[3]))

        with fk.branch(1):                      # temporal prior (dmc_net.py:121-140): contexts only, beside encoder + hyper codec
            q = p + ".temporal_prior_encoder"
            t = ops.gdn(W, q + ".gdn1", ops.conv(W, q + ".conv1", c1, stride=2), "inter")
            t = ops.gdn(W, q + ".gdn2", ops.conv(W, q + ".conv2", [t, c2], stride=2), "inter")
            t = ops.gdn(W, q + ".gdn3", ops.conv(W, q + ".conv3", [t, c3], stride=2), "inter")
            temporal = ops.conv(W, q + ".conv4", t, stride=2)
        if not decoding:
            y = B.res_encoder_gdn(W, p + ".res_encoder", x, c1, c2, c3, "inter", homes=homes)
            z = self._prior_encoder(p + ".res_prior_encoder", y)
            z_hat = z.like()
            ops.factorized_quant_bits(z, W.bit_estimator(p + ".bit_estimator_z"), S, slot_base + 1, z_hat=z_hat)
            self._tap("bl_z", z_hat)
            if sink:
                self._push(sink, z_hat, None, tb["bl_z"])
        else:
            z_hat = self._pull_factorized(source, tb["bl_z"], 64, zh, zw)

        # params = cat(temporal 192, hierarchical 192) -> res_entropy_parameter (dmc_net.py:440-445)
        hier = self._prior_decoder_bl(p + ".res_prior_decoder", z_hat)
        fk.join(1)
        q = p + ".res_entropy_parameter"
        t = ops.conv(W, q + ".0", [temporal, hier], act="lrelu")
        t = ops.conv(W, q + ".2", t, act="lrelu")
        scales, means = ops.conv(W, q + ".4", t).chunk(2)
        if not decoding:
            y_hat = y.like()
            y_q = y.like() if (sink or self.taps is not None) else None
            ops.laplace_quant_bits(y, means, scales, S, slot_base + 0, y_q=y_q, y_hat=y_hat)
            self._tap("bl_y", y_q)
            if sink:
                self._push(sink, y_q, scales, tb["laplace"], LAPLACE_IDX)
                self._prefetch(sink)                  # the layer's last plane: its copy goes ahead of the synthesis / reconstruction kernels
        else:
            y_hat = self._pull_laplace(source, scales, means)

        res = B.res_decoder_gdn(W, p + ".res_decoder", y_hat, c2, c3, "inter", homes=homes)
        feature, recon = B.recon_generation(W, p + ".recon_generation_net", res, c1)
        return {"recon": recon, "feature": feature, "y_hat": y_hat, "mv_hat": mv_hat}

    # ============================================================================ enhancement layer
    def _resampler_tail(self, p, up, out=None):
        """conv2 (conv-lrelu-conv) -> 2 DepthConvBlocks -> + skip (lssvc_modules.py:361-363,394-396,426-428)."""
        W = self.W
        up = ops.conv(W, p + ".conv2.2", ops.conv(W, p + ".conv2.0", up, act="lrelu"))
        # feature_refine(up) + up: the outer skip rides on the second block's fused tail
        return B.depth_conv_block(W, p + ".feature_refine.1", B.depth_conv_block(W, p + ".feature_refine.0", up), out=out, skip=up)

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

    def _ref_pyramid(self, ref, feature_el):
        """multi_scale_feature_extractor (LSSVC_net.py:195-202): feature pyramid of the reference; needs only the DPB."""
        W = self.W
        if feature_el is None:
            f = ops.conv(W, "feature_adaptor_EL_I", ref)
        elif feature_el.C == 64:
            f = ops.conv(W, "feature_adaptor_EL_first_P", feature_el)
        else:
            f = ops.conv(W, "feature_adaptor_EL", feature_el)
        return B.pyramid_extractor(W, "feature_extractor", f)

    def _motion_compensation(self, ref, ref_pyr, mv):
        """LSSVC.motion_compensation (LSSVC_net.py:229-244) on the reference's feature pyramid."""
        W = self.W
        warpframe = ops.flow_warp(ref, mv)
        mv2 = ops.resize(mv, mv.H // 2, mv.W // 2, scale=0.5)
        mv3 = ops.resize(mv2, mv2.H // 2, mv2.W // 2, scale=0.5)
        r1, r2, r3 = ref_pyr
        c1_init = ops.flow_warp(r1, mv)
        c1 = self._offset_diversity(r1, [c1_init, warpframe, mv], mv)
        c2, c3 = ops.flow_warp(r2, mv2), ops.flow_warp(r3, mv3)
        return B.context_fusion(W, "context_fusion_net", c1, c2, c3), warpframe

    def _el_context(self, spat, mv, ref, ref_pyr, outs=(None, None, None)):
        """LSSVC.hybrid_temporal_layer_context_fusion (LSSVC_net.py:246-259); spat = the texture pyramid of the up-sampled
        base-layer feature (`texture_extractor(texture_resampler(texture_bl))`), ref_pyr = _ref_pyramid(...)."""
        W = self.W
        temp, warpframe = self._motion_compensation(ref, ref_pyr, mv)
        if spat is not None:
            ctx = []
            for i, g in enumerate(("generator1", "generator2", "generator3")):   # HybridWeightGenerator :115-154
                q = "weight_map_generator." + g
                f = ops.conv(W, q + ".0", [temp[i], spat[i]])
                f = B.res_block(W, q + ".1", f, end_with_relu=True)
                logits = ops.conv(W, q + ".2", f)
                ctx.append(ops.softmax2_blend(temp[i], spat[i], logits))
        else:
            ctx = list(temp)
        c1, c2, c3 = B.context_fusion(W, "context_fusion_net", ctx[0], ctx[1], ctx[2], outs=outs)
        return c1, c2, c3, warpframe

    def _unet(self, p, x, out=None):
        """UNet (lssvc_modules.py:295-336)."""
        W = self.W
        x1 = B.depth_conv_block(W, p + ".conv1", x)
        x2 = B.depth_conv_block(W, p + ".conv2", ops.pool2x2(x1, is_max=True))
        x3 = B.depth_conv_block(W, p + ".conv3", ops.pool2x2(x2, is_max=True))
        for i in range(4):
            x3 = B.depth_conv_block(W, "%s.context_refine.%d" % (p, i), x3)
        d3 = B.depth_conv_block(W, p + ".up_conv3", [x2, ops.subpel(W, p + ".up3", x3)])
        return B.depth_conv_block(W, p + ".up_conv2", [x1, ops.subpel(W, p + ".up2", d3)], out=out)

    def _four_part_prior(self, y, common, sink=None, source=None):
        """LSSVC.forward_four_part_prior (LSSVC_net.py:338-443) / compress_four_part_prior (write=True) /
        decompress_four_part_prior (LSSVC_net_extend.py:193-263): four masked steps; steps 2-4 recompute
        (sigma, mu) from cat(y_hat_so_far, common_params). Encoder: quantise y; decoder: pull the folded
        C/4-channel symbol plane of each step from the stream and unfold it."""
        W = self.W
        ref = common.slice(0, common.C // 2)
        mk = lambda: T.zeros(ref.H, ref.W, ref.C, self.device)
        y_q, y_hat, s_hat = (mk() if source is None else None), mk(), (mk() if source is None else None)
        scales, means = common.chunk(2)
        for step in range(4):
            if step > 0:
                t = ops.conv(W, "y_spatial_prior_adaptor_%d" % step, [y_hat, common])
                for i in range(3):
                    t = B.depth_conv_block(W, "y_spatial_prior.%d" % i, t)
                scales, means = t.chunk(2)
            if source is None:
                ops.four_part_step(y, means, scales, MASK_OF_CHUNK[step], y_q, y_hat, s_hat)
            else:
                self._pull_laplace(source, scales, means, chunk_of_mask=CHUNK_OF_MASK[step], out=y_hat)
        if sink:
            for step in range(4):          # y_q_w_0..3 / scales_w_0..3 (LSSVC_net.py:432-442), pushed after the loop
                self._push(sink, y_q, s_hat, self._tables["laplace"], LAPLACE_IDX, chunk_of_mask=CHUNK_OF_MASK[step])
        return y_q, y_hat, s_hat

    def _el_codec(self, xe, bl, ref_el, feat_el, sink=None, source=None, fk=None, pre=None, out=None):
        """LSSVC enhancement layer; `bl` = base-layer outputs {feature, mv_hat, y_hat} (encoder-side in estimate
        mode, DECODED in write mode, LSSVC_net_extend.py:143-147). Encoder: forward_one_frame
        (LSSVC_net.py:458-508) / compress (LSSVC_net_extend.py:24-84, symbols pushed in the order mv_z, mv_y, z,
        y_w0..3); decoder: decompress (LSSVC_net_extend.py:86-136). EL estimate bit slots 4..7 = y, mv_y, z, mv_z.
        fk / pre: the frame's ops.Fork and what its branch 0 already issued from the inputs and the DPB alone
        ({"mv": ME_Spynet_DCVC flow, "ref_pyr": _ref_pyramid}); both optional (then everything runs here, in order)."""
        W, S = self.W, self.slots
        tb = self._tables
        H, Wd = self.shape_hr
        decoding = source is not None
        fk = fk if fk is not None else ops.Fork(self.device, enabled=False)
        pre = pre or {}
        # ILP (LSSVC_net.py:454-456): the three base-layer tensors the EL reads, de-padded (a no-op for test.py's zeros)
        bl = {"feature": self._depad(bl["feature"]), "mv_hat": self._depad(bl["mv_hat"]), "y_hat": self._depad(bl["y_hat"], 16)}
        fused = T.empty(H // 16, Wd // 16, 384, self.device)       # cat(hyper 128, temporal 128, layer 128), filled in place
        with fk.branch(1):        # BL texture -> EL pyramid (lssvc_modules.py:368-397,157-177): beside the whole EL MV codec
            spat = B.pyramid_extractor(W, "texture_extractor", self._texture_resampler(bl["feature"]))
        with fk.branch(2):        # BL latent -> EL prior (lssvc_modules.py:400-429): small maps, needed only by the prior fusion
            self._layer_prior_resampler(bl["y_hat"], out=fused.slice(256, 384))
        mv_up = self._mv_resampler(bl["mv_hat"])
        # mv_ctx_prior_encoder (LSSVC_net.py:108-116)
        t, e = mv_up, "mv_ctx_prior_encoder"
        for base in (0, 2, 4):
            t = ops.gdn(W, "%s.%d" % (e, base + 1), ops.conv(W, "%s.%d" % (e, base), t, stride=2), "inter")
        mv_ctx_prior = ops.conv(W, e + ".6", t, stride=2)
        mv_ctx = B.res_block(W, "mv_ctx_transform.transform.1", ops.conv(W, "mv_ctx_transform.transform.0", mv_up, stride=2))

        zh, zw = bitstream.get_downsampled_shape(H, Wd, 64)
        if not decoding:
            if "mv" in pre:
                fk.join(0)
                mv = pre["mv"]
            else:
                mv = B.spynet(W, "optic_flow", xe, ref_el)
            # MVResEncoder (lssvc_modules.py:445-469)
            e = "mv_encoder.encoder1"
            t = ops.gdn(W, e + ".1", ops.conv(W, e + ".0", mv, stride=2), "inter")
            t = ops.lrelu(B.res_block(W, e + ".2", t, start_from_relu=False), 0.1)
            e = "mv_encoder.encoder2"
            t = ops.gdn(W, e + ".1", ops.conv(W, e + ".0", [t, mv_ctx], stride=2), "inter")
            t = B.res_block(W, e + ".2", t, start_from_relu=False)          # (+ LeakyReLU(0.1): applied by its one reader, as in the BL)
            t = ops.gdn(W, e + ".5", ops.conv(W, e + ".4", t, stride=2, in_act="lrelu", in_slope=0.1), "inter")
            t = B.res_block(W, e + ".6", t, start_from_relu=False)
            mv_y = ops.conv(W, e + ".8", t, stride=2, in_act="lrelu", in_slope=0.1)
            mv_z = self._prior_encoder("mv_prior_encoder", mv_y)
            mv_z_hat = mv_z.like()
            ops.factorized_quant_bits(mv_z, W.bit_estimator("bit_estimator_z_mv"), S, 7, z_hat=mv_z_hat)
            self._tap("el_mv_z", mv_z_hat)
            if sink:
                self._push(sink, mv_z_hat, None, tb["el_z_mv"])
        else:
            mv_z_hat = self._pull_factorized(source, tb["el_z_mv"], 64, zh, zw)
        q = "mv_prior_decoder"
        t = ops.subpel(W, q + ".0", mv_z_hat, act="lrelu")
        t = ops.subpel(W, q + ".2", t, act="lrelu")
        hyper = ops.conv(W, q + ".4", t)
        q = "mv_prior_fusion"
        t = ops.conv(W, q + ".0", [hyper, mv_ctx_prior], act="lrelu")
        t = ops.conv(W, q + ".2", t, act="lrelu")
        mv_scales, mv_means = ops.conv(W, q + ".4", t).chunk(2)
        if not decoding:
            mv_y_hat = mv_y.like()
            mv_y_q = mv_y.like() if (sink or self.taps is not None) else None
            ops.laplace_quant_bits(mv_y, mv_means, mv_scales, S, 5, y_q=mv_y_q, y_hat=mv_y_hat)
            self._tap("el_mv_y", mv_y_q)
            if sink:
                self._push(sink, mv_y_q, mv_scales, tb["laplace"], LAPLACE_IDX)
        else:
            mv_y_hat = self._pull_laplace(source, mv_scales, mv_means)
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

        if "ref_pyr" in pre:
            fk.join(0)
            ref_pyr = pre["ref_pyr"]
        else:
            ref_pyr = self._ref_pyramid(ref_el, feat_el)
        fk.join(1)
        # contexts 2 / 3 are made where the bottleneck ResBlocks of the residual encoder and decoder read them: cat(64, context2) at
        # 1/2 and cat(96, context3) at 1/4 resolution, one buffer each for the frame (B.context_homes: no concat copies)
        homes = B.context_homes(W, "res_encoder", H // 2, Wd // 2, W.raw("context_fusion_net.conv2_out.weight").shape[0],
                                W.raw("context_fusion_net.conv3_out.weight").shape[0], self.device)
        c1, c2, c3, warp_frame = self._el_context(spat, mv_hat, ref_el, ref_pyr, outs=(None, homes[2], homes[3]))

        with fk.branch(1):        # temporal prior (LSSVC_net.py:75-79): context3 only, beside the encoder and the hyper codec
            q = "temporal_prior_encoder"
            t = ops.conv(W, q + ".0", c3, stride=2, act="lrelu", slope=0.1)
            ops.conv(W, q + ".2", t, stride=2, out=fused.slice(128, 256))

        if not decoding:
            # ResEncoder without GDN (lssvc_modules.py:235-254); the concat feeding each ResBlock is built in place
            p = "res_encoder"
            t, u = homes[0], homes[1]
            assert t.C == 64 + c2.C and u.C == 96 + c3.C
            ops.conv(W, p + ".conv1", [xe, c1], stride=2, out=t.slice(0, 64))
            t = B.res_block(W, p + ".res1", t, slope=0.1, end_with_relu=True)
            ops.conv(W, p + ".conv2", t, stride=2, out=u.slice(0, 96))
            u = B.res_block(W, p + ".res2", u, slope=0.1, end_with_relu=True)
            y = ops.conv(W, p + ".conv4", ops.conv(W, p + ".conv3", u, stride=2), stride=2)
            z = self._prior_encoder("res_prior_encoder", y)
            z_hat = z.like()
            ops.factorized_quant_bits(z, W.bit_estimator("bit_estimator_z"), S, 6, z_hat=z_hat)
            self._tap("el_z", z_hat)
            if sink:
                self._push(sink, z_hat, None, tb["el_z"])
        else:
            y = None
            z_hat = self._pull_factorized(source, tb["el_z"], 128, zh, zw)

        # prior fusion input cat(hyper 128, temporal 128, layer 128) written in place (lssvc_modules.py:440-442)
        q = "res_prior_decoder"                                     # LSSVC_net.py:63-73
        t = ops.conv(W, q + ".0", z_hat, act="lrelu")
        t = ops.subpel(W, q + ".2", t, act="lrelu")
        t = ops.conv(W, q + ".4", t, act="lrelu")
        t = ops.subpel(W, q + ".6", t, act="lrelu")
        ops.conv(W, q + ".8", t, out=fused.slice(0, 128))
        fk.join(1)
        fk.join(2)
        params = B.depth_conv_block(W, "prior_fusion_net.prior_fusion_conv.1",
                                    B.depth_conv_block(W, "prior_fusion_net.prior_fusion_conv.0", fused))
        y_q, y_hat, scales_hat = self._four_part_prior(y, params, sink=sink, source=source)
        if sink is not None:
            self._prefetch(sink)                      # (as in _bl_codec)
        if not decoding:
            ops.laplace_bits(y_q, scales_hat, S, 4)
            self._tap("el_y", y_q)

        # ResDecoder (lssvc_modules.py:257-276)
        p = "res_decoder"
        t = ops.subpel(W, p + ".up1", y_hat)
        u = homes[1]                                        # (the encoder's parts of the two buffers are dead by now: y came from them)
        ops.subpel(W, p + ".up2", t, out=u.slice(0, 96))
        u = B.res_block(W, p + ".res1", u, slope=0.1, end_with_relu=True)
        t = homes[0]
        ops.subpel(W, p + ".up3", u, out=t.slice(0, 64))
        t = B.res_block(W, p + ".res2", t, slope=0.1, end_with_relu=True)
        res = ops.subpel(W, p + ".up4", t)

        # ReconGeneration (lssvc_modules.py:279-292), called as (recon_image_feature, context1) (LSSVC_net.py:492)
        p = "recon_generation_net"
        f = ops.conv(W, p + ".first_conv", [res, c1])
        out = out or {}                                    # (look-ahead plans: the two results the next frame reads, written in place)
        feature = self._unet(p + ".unet_2", self._unet(p + ".unet_1", f), out=out.get("feature"))
        recon_el = ops.conv(W, p + ".recon_conv", feature, out=out.get("recon"))
        return feature, recon_el, mv_hat, warp_frame

    # ---------------------------------------------------------------------------------------------
    def _fork_el_head(self, x_el, ref_el, feat_el):
        """-> (fk, pre): the frame's Fork with branch 0 already issued -- what the EL needs from the inputs and the DPB alone
        (encoder: motion estimation + reference pyramid; decoder, x_el None: the reference pyramid), beside the whole BL codec."""
        fk = ops.Fork(self.device)
        pre = {}
        if fk.enabled:
            with fk.branch(0):
                if x_el is not None:
                    pre["mv"] = B.spynet(self.W, "optic_flow", x_el, ref_el)
                pre["ref_pyr"] = self._ref_pyramid(ref_el, feat_el)
        return fk, pre

    def _frame_body(self, t):
        fk, pre = self._fork_el_head(t["x_el"], t["ref_frame_el"], t["ref_feature_el"])
        bl = self._bl_codec(t["x_bl"], t["ref_frame_bl"], t["ref_feature_bl"], fk=fk)
        feature, recon_el, mv_hat, warp_frame = self._el_codec(t["x_el"], bl, t["ref_frame_el"], t["ref_feature_el"], fk=fk, pre=pre)
        fk.close()
        return {"recon_bl": bl["recon"], "feature_bl": bl["feature"], "recon_el": recon_el, "feature_el": feature,
                "mv_hat": mv_hat, "warp_frame": warp_frame}

    # ---- look-ahead: the base layer of frame t+1 beside the enhancement layer of frame t ---------------------------------------
    # The base layer of a P-frame needs the previous frame's BASE layer only (dmc_net.py:421-488: its reference frame and feature),
    # the enhancement layer needs its own previous frame and the base layer of the SAME frame (LSSVC_net.py:445-528). So BL(t+1) and
    # EL(t) are independent, and a caller that can name the next frame (test.py's loop can: the frames are all there) may have them
    # in flight together: same launches, same order inside either layer, bit-identical results -- but the small-map launches and the
    # tails of the big ones of one layer now fill with the other layer's work. They are TWO frame plans (two
    # hipGraphs), BL(t+1) launched on a second stream and EL(t) on the caller's; forking BL(t+1)'s own side chains from a branch of
    # one captured graph -- a fork inside a fork -- crashes hipGraph capture on this ROCm. BL(t+1) reads frame t's BL reconstruction
    # clamped to [0,1] -- what test.py:249-250 makes of the DPB before the next frame; a caller of this mode promises that clamp --
    # and leaves its four results in persistent buffers (self._stash, by frame parity) for the next call, which codes its EL only.
    STASH_KEYS = ("recon", "feature", "y_hat", "mv_hat")
    MAX_GEOMS = 3           # (intra._HostModel.MAX_PLANS is sized for this many geometries' plan sets)

    def _bufs(self, geom=None):
        """The look-ahead buffers of a geometry (default: the frame being issued), least recently used geometries dropped together
        with every frame plan that has their addresses baked in."""
        geom = self._geom if geom is None else geom
        b = self._lookahead_bufs.pop(geom, None)
        if b is None:
            assert not torch.cuda.is_current_stream_capturing()
            b = {"stash": [None, None], "el_out": [None, None]}
            while len(self._lookahead_bufs) >= self.MAX_GEOMS:
                old = next(iter(self._lookahead_bufs))
                del self._lookahead_bufs[old]
                self._plans = {k: v for k, v in self._plans.items() if old not in k}
        self._lookahead_bufs[geom] = b                     # (re-)insert as most recently used
        return b

    @property
    def _stash(self):
        return self._bufs()["stash"]

    @property
    def _el_out(self):
        return self._bufs()["el_out"]

    def _ahead_bl_body(self, t, src_parity, out_parity):
        """BL(t+1): reference = clamp(frame t's BL reconstruction) and its feature, from the stash (src_parity) or, behind a frame
        coded whole, from the inputs ref_recon / ref_feature; bit slots 8..11; results -> self._stash[out_parity]."""
        src = self._stash[src_parity] if src_parity is not None else {"recon": t["ref_recon"], "feature": t["ref_feature"]}
        rec = src["recon"]
        assert rec.ld == rec.C
        ref = ops.clamp_(ops.copy(rec, T.empty(rec.H, rec.W, rec.C, self.device)))      # = test.py:249's clamp_(0, 1); two launches of the library (round 5: torch.clamp, an ATen
        #                                                                                 kernel inside the frame plan -- the one launch that was not the extension's)
        fk = ops.Fork(self.device)
        self.slots.lane = 1                                # (its own reduction workspaces: this plan runs beside the EL's)
        try:
            nb = self._bl_codec(t["next_x_bl"], ref, src["feature"], fk=fk, slot_base=8)
        finally:
            self.slots.lane = 0
        fk.close()
        if self._stash[out_parity] is None:                # (first, eager call of this plan: never inside a capture)
            assert not torch.cuda.is_current_stream_capturing()
            self._stash[out_parity] = {k: T(torch.empty(nb[k].H * nb[k].W * nb[k].C, dtype=torch.float32, device=self.device),
                                            nb[k].H, nb[k].W, nb[k].C, nb[k].C) for k in self.STASH_KEYS}
        for k in self.STASH_KEYS:
            ops.copy(nb[k], self._stash[out_parity][k])
        return {}

    def _ahead_el_body(self, t, parity):
        """EL(t) on the base layer that was coded ahead (self._stash[parity]). Its reconstruction and feature -- the next frame's
        references -- go to persistent buffers by frame parity (self._el_out); when the caller hands exactly those back as this
        frame's references (test.py's loop does: the DPB), they are read where they are instead of being copied into the plan's
        inputs (566 MB of feature per 1080p frame)."""
        bl = self._stash[parity]
        ref_el, feat_el = t["ref_frame_el"], t["ref_feature_el"]
        if feat_el is None:                                # bound: the previous frame's outputs, in place
            ref_el, feat_el = self._el_out[1 - parity]["recon"], self._el_out[1 - parity]["feature"]
        if self._el_out[parity] is None:                   # (first, eager call of a plan: never inside a capture)
            assert not torch.cuda.is_current_stream_capturing()
            self._el_out[parity] = {k: T(torch.empty(v.H * v.W * v.C, dtype=torch.float32, device=self.device), v.H, v.W, v.C, v.C)
                                    for k, v in (("recon", ref_el), ("feature", feat_el))}
        fk, pre = self._fork_el_head(t["x_el"], ref_el, feat_el)
        feature, recon_el, mv_hat, warp_frame = self._el_codec(t["x_el"], bl, ref_el, feat_el, fk=fk, pre=pre, out=self._el_out[parity] if EL_BIND else None)
        fk.close()
        return {"recon_bl": bl["recon"], "feature_bl": bl["feature"], "recon_el": recon_el, "feature_el": feature,
                "mv_hat": mv_hat, "warp_frame": warp_frame}

    def _run_body(self, key, tensors, body):
        if self.graph_mode:
            return self._run_planned(key, tensors, body)
        ins = {k: (None if v is None else T.from_nchw(v)) for k, v in tensors.items()}
        return self._with_range_audit(key, lambda: body(ins))

    @staticmethod
    def _frame_key(tensors):
        """Frame type + sizes: first-P (no BL feature, 64-channel EL feature) and steady-P differ in their shapes."""
        return ("p",) + tuple(None if v is None else tuple(v.shape) for v in tensors.values())

    def forward_one_frame(self, x_bl, x_el, ref_frame_bl, ref_frame_el, ref_feature_bl, ref_feature_el, next_x_bl=None, frame_id=None):
        """LSSVC.forward_one_frame (LSSVC_net.py:445-528): estimate mode.
        frame_id (an int that grows by one per frame of the sequence) switches the look-ahead protocol on (_ahead_bl_body):
        next_x_bl is the NEXT frame's base-layer input (None behind the last frame), whose base layer is coded beside this frame's
        enhancement layer and kept for the call with frame_id + 1. The caller clamps the DPB's reference frames to [0, 1] between
        frames, as test.py:249-250 does. Results are those of the plain calls, bit for bit."""
        self._require_device()
        assert tuple(x_el.shape[2:]) == self.shape_hr, "x_el is %dx%d but shape_hr is %s" % (x_el.shape[2], x_el.shape[3], self.shape_hr)
        tensors = {"x_bl": x_bl, "x_el": x_el, "ref_frame_bl": ref_frame_bl, "ref_frame_el": ref_frame_el,
                   "ref_feature_bl": ref_feature_bl, "ref_feature_el": ref_feature_el}
        t_issue = time.perf_counter()
        stashed = None
        if frame_id is None:
            # (Round 4 also sent single-stream mode, LSSVC_STREAMS=0, down this branch: with the look-ahead plans captured there,
            # replayed graphs -- the plain plans included -- gave deterministically wrong results. Round 5 found the cause:
            # T.zeros went through hipMemsetAsync, which a stream capture records as a MEMSET NODE, and with several captured
            # frame plans alive this ROCm replays such nodes wrongly (the zero-initialised SpyNet flow / four-part buffers came back
            # stale; every other node was right). lssvc_fill_zero is a kernel launch now, and the mode needs no guard:
            # tools/debug_lookahead_single.py, profiles/r05_lookahead_root_cause.txt)
            self._ahead = None
            frame_id = None
            r = self._run_body(self._frame_key(tensors), tensors, self._frame_body)
        else:
            fid = int(frame_id)
            # everything the look-ahead buffers' shapes and the plans that bake their addresses in depend on (one
            # model object codes several sizes and ratios in a harness run)
            geom = ("geom", tuple(x_bl.shape[1:]), tuple(x_el.shape[1:]), float(self.scale_factor), self.shape_hr, self.pad_size, ops.CONV_PRECISION)
            self._geom = geom
            if next_x_bl is not None:
                assert tuple(next_x_bl.shape) == tuple(x_bl.shape), "next_x_bl is %s, x_bl %s" % (tuple(next_x_bl.shape), tuple(x_bl.shape))
            stashed = self._ahead if (self._ahead is not None and self._ahead["for"] == fid and self._ahead["geom"] == geom) else None
            self._ahead = None
            main = torch.cuda.current_stream(self.device)
            if self._ahead_stream is None:
                self._ahead_stream = torch.cuda.Stream(self.device)      # (same priority as the caller's: a high-priority BL(t+1) costs 13 %, profiles/r04_lookahead_ab.txt)
            side = self._ahead_stream
            shape = lambda v: None if v is None else tuple(v.shape)

            def code_ahead(src):
                tb = {"next_x_bl": next_x_bl}
                if src is not None:
                    tb.update(ref_recon=src["recon_bl"].to_nchw(), ref_feature=src["feature_bl"].to_nchw())
                key = ("p-ahead-bl", geom, src is None, fid & 1) + tuple(shape(v) for v in tb.values())
                side.wait_stream(main)                     # (the previous frame, the caller's clamp of the DPB; not this frame's EL)
                with torch.cuda.stream(side):
                    self._run_body(key, tb, lambda ins: self._ahead_bl_body(ins, (fid & 1) if src is None else None, (fid + 1) & 1))

            if stashed is None:                            # no base layer coded ahead for this frame: the whole frame, then BL(t+1)
                r = self._run_body(self._frame_key(tensors), tensors, self._frame_body)
                if next_x_bl is not None:
                    code_ahead(r)
            else:
                if next_x_bl is not None:
                    code_ahead(None)                       # BL(t+1) first, on the side stream ...
                te = {"x_el": x_el, "ref_frame_el": ref_frame_el, "ref_feature_el": ref_feature_el}
                prev = self._el_out[1 - (fid & 1)]
                if EL_BIND and prev is not None and all(v.dim() == 4 and v.data_ptr() == prev[k].buf.data_ptr() + 4 * prev[k].off and tuple(v.shape) == (1, prev[k].C, prev[k].H, prev[k].W)
                                            and v.stride() == (prev[k].H * prev[k].W * prev[k].C, 1, prev[k].W * prev[k].C, prev[k].C)
                                            for k, v in (("recon", ref_frame_el), ("feature", ref_feature_el))):
                    te.update(ref_frame_el=None, ref_feature_el=None)       # the previous frame's outputs where they are (_ahead_el_body)
                r = self._run_body(("p-ahead-el", geom, fid & 1) + tuple(shape(v) for v in te.values()), te,
                                   lambda ins: self._ahead_el_body(ins, fid & 1))      # ... EL(t) beside it
            main.wait_stream(side)
        self.last_issue_s = time.perf_counter() - t_issue        # host time to put the frame on the stream (no GPU wait)
        r = {k: self._own(v) for k, v in r.items()}
        dpb = {"ref_frame_bl": r["recon_bl"].to_nchw(remember=True), "ref_feature_bl": r["feature_bl"].to_nchw(remember=True),
               "ref_frame_el": r["recon_el"].to_nchw(remember=True), "ref_feature_el": r["feature_el"].to_nchw(remember=True)}
        out = {"dpb": dpb, "mv_hat": r["mv_hat"].to_nchw(), "warp_frame": r["warp_frame"].to_nchw(),
               "encoding_time_EL": 0.0, "decoding_time_EL": 0.0, "encoding_time_BL": 0.0, "decoding_time_BL": 0.0}
        s = self.slots.fetch()
        b = stashed["bits"] if stashed is not None else s[0:4]
        out["bit_bl"] = b[0] + b[1] + b[2] + b[3]          # y + z + mv_y + mv_z  (dmc_net.py:473)
        out["bit_el"] = s[4] + s[5] + s[6] + s[7]          # y + mv_y + z + mv_z  (LSSVC_net.py:508)
        if frame_id is not None and next_x_bl is not None:
            self._ahead = {"for": int(frame_id) + 1, "geom": self._geom, "bits": s[8:12]}
        return out

    def encode_decode_extend(self, x_bl, x_el, dpb, output_path_bl, output_path_el):
        """LSSVC_extend.encode_decode_extend (LSSVC_net_extend.py:138-191) with DMCExtend.encode_decode_extend
        (dmc_net_extend.py:148-173): each layer is compressed to a real rANS string, framed, written, read back
        and DECODED; the decoded tensors become the DPB. The EL sees the DECODED base layer."""
        self._require_device()
        if self._tables is None:
            raise ValueError("Uninitialized CDFs. Run update() first")
        nhwc = lambda t: None if t is None else T.from_nchw(t)
        xb, xe = nhwc(x_bl), nhwc(x_el)
        ref_bl, ref_el = nhwc(dpb["ref_frame_bl"]), nhwc(dpb["ref_frame_el"])
        feat_bl, feat_el = nhwc(dpb["ref_feature_bl"]), nhwc(dpb["ref_feature_el"])
        ins = {"x_bl": xb, "x_el": xe, "ref_frame_bl": ref_bl, "ref_frame_el": ref_el, "ref_feature_bl": feat_bl, "ref_feature_el": feat_el}
        self._with_range_audit(("p",) + tuple(None if v is None else (1, v.C, v.H, v.W) for v in ins.values()),
                               lambda: self._frame_body(ins))                # first frame of a type only (hip_ops.RangeAudit)
        sync = lambda: torch.cuda.synchronize(self.device)
        # ---- base layer ----
        # (every codec call below issues its independent chains as parallel branches, as the estimate-mode frame does: a
        # ops.Fork per call, closed -- all branches joined -- before the call's results are used)
        sync(); t0 = time.time()
        sink = SymbolSink(self._begin_layer())
        fk = ops.Fork(self.device)
        bl_e = self._bl_codec(xb, ref_bl, feat_bl, sink=sink, fk=fk)
        fk.close()
        bitstream.encode_p(sink.flush(), output_path_bl)
        bit_bl = bitstream.filesize(output_path_bl) * 8
        sync(); t1 = time.time()
        fk = ops.Fork(self.device)
        bl = self._bl_codec(None, ref_bl, feat_bl, source=SymbolSource(bitstream.decode_p(output_path_bl), self._begin_layer()), fk=fk)
        fk.close()
        recon_bl = bl["recon"].to_nchw(copy=True).clamp_(0, 1)                         # dmc_net_extend.py:138
        sync(); t2 = time.time()
        # ---- enhancement layer ----
        sink = SymbolSink(self._begin_layer())
        fk, pre = self._fork_el_head(xe, ref_el, feat_el)
        feature_e, recon_e, mv_hat, warp_frame = self._el_codec(xe, bl, ref_el, feat_el, sink=sink, fk=fk, pre=pre)
        fk.close()
        bitstream.encode_p(sink.flush(), output_path_el)
        bit_el = bitstream.filesize(output_path_el) * 8
        est = self.slots.fetch()
        sync(); t3 = time.time()
        fk, pre = self._fork_el_head(None, ref_el, feat_el)
        feature, recon_el, _, _ = self._el_codec(None, bl, ref_el, feat_el, source=SymbolSource(bitstream.decode_p(output_path_el), self._begin_layer()),
                                                 fk=fk, pre=pre)
        fk.close()
        sync(); t4 = time.time()
        out_dpb = {"ref_frame_bl": recon_bl, "ref_feature_bl": bl["feature"].to_nchw(),
                   "ref_frame_el": recon_el.to_nchw(), "ref_feature_el": feature.to_nchw()}
        return {"dpb": out_dpb, "bit_bl": bit_bl, "bit_el": bit_el,
                "encoding_time_BL": t1 - t0, "decoding_time_BL": t2 - t1, "encoding_time_EL": t3 - t2, "decoding_time_EL": t4 - t3,
                "mv_hat": mv_hat.to_nchw(), "warp_frame": warp_frame.to_nchw(),
                # extras (not in the reference's dict)
                "bit_bl_estimate": est[0] + est[1] + est[2] + est[3], "bit_el_estimate": est[4] + est[5] + est[6] + est[7],
                "encoder_side": {"ref_frame_bl": bl_e["recon"].to_nchw(), "ref_feature_bl": bl_e["feature"].to_nchw(),
                                 "ref_frame_el": recon_e.to_nchw(), "ref_feature_el": feature_e.to_nchw()}}

    def encode(self, x_bl, x_el, dpb, output_path_bl, output_path_el):
        """Encoder only: the compress half of encode_decode_extend (DMCExtend.compress dmc_net_extend.py:55-104 +
        LSSVC_extend.compress LSSVC_net_extend.py:24-86) -- writes the two layer files and returns {"dpb": ...} built from the
        encoder-side reconstruction, which is bit for bit what decode() makes of those files, so an encoder process never has
        to decode: ≈53 instead of ≈101 ms per 1080p P-frame. ref_frame_bl comes back clamped, as the decoder returns it
        (dmc_net_extend.py:138)."""
        self._require_device()
        if self._tables is None:
            raise ValueError("Uninitialized CDFs. Run update() first")
        nhwc = lambda t: None if t is None else T.from_nchw(t)
        xb, xe = nhwc(x_bl), nhwc(x_el)
        ref_bl, ref_el = nhwc(dpb["ref_frame_bl"]), nhwc(dpb["ref_frame_el"])
        feat_bl, feat_el = nhwc(dpb["ref_feature_bl"]), nhwc(dpb["ref_feature_el"])
        ins = {"x_bl": xb, "x_el": xe, "ref_frame_bl": ref_bl, "ref_frame_el": ref_el, "ref_feature_bl": feat_bl, "ref_feature_el": feat_el}
        self._with_range_audit(("p",) + tuple(None if v is None else (1, v.C, v.H, v.W) for v in ins.values()),
                               lambda: self._frame_body(ins))
        fk, pre = self._fork_el_head(xe, ref_el, feat_el)         # EL motion estimation + reference pyramid run beside the BL codec
        sink = SymbolSink(self._begin_layer())
        bl = self._bl_codec(xb, ref_bl, feat_bl, sink=sink, fk=fk)
        bitstream.encode_p(sink.flush(), output_path_bl)
        sink = SymbolSink(self._begin_layer())
        feature, recon_el, mv_hat, warp_frame = self._el_codec(xe, bl, ref_el, feat_el, sink=sink, fk=fk, pre=pre)
        fk.close()
        bitstream.encode_p(sink.flush(), output_path_el)
        return {"dpb": {"ref_frame_bl": bl["recon"].to_nchw(copy=True).clamp_(0, 1), "ref_feature_bl": bl["feature"].to_nchw(),
                        "ref_frame_el": recon_el.to_nchw(), "ref_feature_el": feature.to_nchw()},
                "bit_bl": bitstream.filesize(output_path_bl) * 8, "bit_el": bitstream.filesize(output_path_el) * 8,
                "mv_hat": mv_hat.to_nchw(), "warp_frame": warp_frame.to_nchw()}

    def decode(self, dpb, input_path_bl, input_path_el):
        """Decoder only: reconstruct a P-frame from its two layer files and the previous frame's DPB (the decode half
        of encode_decode_extend = DMCExtend.decompress dmc_net_extend.py:106-146 + LSSVC_extend.decompress
        LSSVC_net_extend.py:88-136). Returns {"dpb": ...} like encode_decode."""
        self._require_device()
        if self._tables is None:
            raise ValueError("Uninitialized CDFs. Run update() first")
        nhwc = lambda t: None if t is None else T.from_nchw(t)
        ref_bl, ref_el = nhwc(dpb["ref_frame_bl"]), nhwc(dpb["ref_frame_el"])
        feat_bl, feat_el = nhwc(dpb["ref_feature_bl"]), nhwc(dpb["ref_feature_el"])
        fk, pre = self._fork_el_head(None, ref_el, feat_el)       # the EL reference pyramid runs beside the BL decoder; while the host
        #                                                           decodes a plane, the side streams keep the GPU busy
        bl = self._bl_codec(None, ref_bl, feat_bl, source=SymbolSource(bitstream.decode_p(input_path_bl), self._begin_layer()), fk=fk)
        recon_bl = bl["recon"].to_nchw(copy=True).clamp_(0, 1)                         # dmc_net_extend.py:138
        feature, recon_el, _, _ = self._el_codec(None, bl, ref_el, feat_el, source=SymbolSource(bitstream.decode_p(input_path_el), self._begin_layer()),
                                                 fk=fk, pre=pre)
        fk.close()
        return {"dpb": {"ref_frame_bl": recon_bl, "ref_feature_bl": bl["feature"].to_nchw(remember=True),
                        "ref_frame_el": recon_el.to_nchw(remember=True), "ref_feature_el": feature.to_nchw(remember=True)}}

    # ---- the reference's lower-level EL API (round 5): same names, arguments, result keys --------------------------------------
    def _el_dpb_inputs(self, dpb):
        nhwc = lambda t: None if t is None else T.from_nchw(t)
        bl = {"feature": nhwc(dpb["texture"]), "y_hat": nhwc(dpb["y_hat_bl"]), "mv_hat": nhwc(dpb["mv_hat_bl"])}
        return bl, nhwc(dpb["ref_frame_el"]), nhwc(dpb["ref_feature_el"])

    def compress(self, x, dpb):
        """LSSVC_extend.compress (LSSVC_net_extend.py:24-84): the enhancement layer of one P-frame -> ONE rANS string (symbols in
        the order mv_z, mv_y, z, y_w0..3) and the encoder-side reconstruction. dpb: ref_frame_el, ref_feature_el and the base-layer
        information texture / y_hat_bl / mv_hat_bl (un-depadded: get_depadded_feature is applied here, as in the reference).
        -> {"string": bytes, "dpb": {ref_frame_el, ref_feature_el, warp_frame, mv_hat}}."""
        self._require_device()
        if self._tables is None:
            raise ValueError("Uninitialized CDFs. Run update() first")
        bl, ref_el, feat_el = self._el_dpb_inputs(dpb)
        xe = T.from_nchw(x)
        sink = SymbolSink(self._begin_layer())
        fk, pre = self._fork_el_head(xe, ref_el, feat_el)
        feature, recon_el, mv_hat, warp_frame = self._el_codec(xe, bl, ref_el, feat_el, sink=sink, fk=fk, pre=pre)
        fk.close()
        return {"string": sink.flush(),
                "dpb": {"ref_frame_el": recon_el.to_nchw(), "ref_feature_el": feature.to_nchw(), "warp_frame": warp_frame.to_nchw(), "mv_hat": mv_hat.to_nchw()}}

    def decompress(self, string, height, width, dpb):
        """LSSVC_extend.decompress (LSSVC_net_extend.py:86-136): the enhancement layer back from its string; height / width are the
        padded EL picture size the latent shapes follow from (get_downsampled_shape). -> {"dpb": {ref_frame_el, ref_feature_el}}."""
        self._require_device()
        if self._tables is None:
            raise ValueError("Uninitialized CDFs. Run update() first")
        assert (int(height), int(width)) == self.shape_hr, "decompress(%d x %d) but shape_hr is %s" % (height, width, self.shape_hr)
        bl, ref_el, feat_el = self._el_dpb_inputs(dpb)
        fk, pre = self._fork_el_head(None, ref_el, feat_el)
        feature, recon_el, _, _ = self._el_codec(None, bl, ref_el, feat_el, source=SymbolSource(string, self._begin_layer()), fk=fk, pre=pre)
        fk.close()
        return {"dpb": {"ref_frame_el": recon_el.to_nchw(), "ref_feature_el": feature.to_nchw()}}

    def encode_decode(self, x_bl, x_el, dpb, output_path_bl=None, output_path_el=None,
                      pic_width=None, pic_height=None, pic_width_bl=None, pic_height_bl=None, next_x_bl=None, frame_id=None):
        """LSSVC.encode_decode (LSSVC_net.py:172-185). output_path_el None <=> estimate mode (next_x_bl / frame_id: its look-ahead
        protocol, see forward_one_frame)."""
        if output_path_el is not None:
            return self.encode_decode_extend(x_bl, x_el, dpb, output_path_bl, output_path_el)
        return self.forward_one_frame(x_bl, x_el, dpb["ref_frame_bl"], dpb["ref_frame_el"], dpb["ref_feature_bl"],
                                      dpb["ref_feature_el"], next_x_bl=next_x_bl, frame_id=frame_id)

    def update(self, force=False):
        """LSSVC_extend.update + DMCExtend.update (LSSVC_net_extend.py:17-22, dmc_net_extend.py:49-53)."""
        if self._tables is not None and not force:
            return
        sd = self._sd
        self._tables = {"laplace": tables.laplace_tables(),
                        "el_z": tables.bit_estimator_tables(sd, "bit_estimator_z"),
                        "el_z_mv": tables.bit_estimator_tables(sd, "bit_estimator_z_mv"),
                        "bl_z": tables.bit_estimator_tables(sd, "base_layer_model.bit_estimator_z"),
                        "bl_z_mv": tables.bit_estimator_tables(sd, "base_layer_model.bit_estimator_z_mv")}
