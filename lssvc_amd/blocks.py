"""LSSVC's NN building blocks expressed as sequences of fused HIP ops (NHWC, fp32).

Each function mirrors one reference nn.Module (cited per function) but is laid out for the
MI355X kernels: torch.cat in front of a conv becomes a multi-input conv, PixelShuffle is the conv's
store pattern, bias / LeakyReLU / residual adds / GDN normalisation are conv epilogues, channel
chunk() is a strided view. `W` is a weights.WeightStore, `p` the state-dict prefix of the module.
"""
from . import hip_ops as ops
from .hip_ops import T


import os as _os

FOLD_SKIP_ADDS = True      # fold `skip + res_block(x)` into the block's last conv (second residual operand); False: separate add


def _vec4(t):
    """16-byte addressable view: what lssvc_conv2d's straight-line epilogue needs of out / residual / residual2 (vec4_ok)."""
    return t.C % 4 == 0 and t.ld % 4 == 0 and t.v.ptr % 16 == 0


def _can_fold(x, skip, out, slope):
    """Exactly the conditions under which lssvc_conv2d accepts a second residual (csrc/conv_mfma.hip: fast_epi == 1); anything
    else takes the separate add instead of a hard error (e.g. LSSVC_FAST_EPI=0, a documented debug switch)."""
    return FOLD_SKIP_ADDS and _os.environ.get("LSSVC_FAST_EPI", "1") != "0" and 0.0 <= slope <= 1.0 and _vec4(x) and _vec4(skip) \
        and (out is None or _vec4(out))


def res_block(W, p, x, slope=0.01, start_from_relu=True, end_with_relu=False, out=None, skip=None):
    """ResBlock (layers.py:229-255, video_net_component.py:170-188): x + [lrelu]conv2(lrelu(conv1([lrelu]x))).
    skip: an outer sum `skip + block(x)` that follows the block (lssvc_modules.py:226-231), computed in the same launch as
    (conv2 + x) + skip -- the reference's rounding order, fp32 addition being commutative."""
    t = ops.conv(W, p + ".conv1", x, in_act="lrelu" if start_from_relu else None, in_slope=slope, act="lrelu", slope=slope)
    if skip is None:
        return ops.conv(W, p + ".conv2", t, act="lrelu" if end_with_relu else None, slope=slope, residual=x, out=out)
    if _can_fold(x, skip, out, slope):
        return ops.conv(W, p + ".conv2", t, act="lrelu" if end_with_relu else None, slope=slope, residual=x, residual2=skip, out=out)
    return ops.add(skip, ops.conv(W, p + ".conv2", t, act="lrelu" if end_with_relu else None, slope=slope, residual=x), out=out)


def residual_block(W, p, x):
    """ResidualBlock (layers.py:122-145): lrelu(conv2(lrelu(conv1 x))) + x."""
    t = ops.conv(W, p + ".conv1", x, act="lrelu")
    return ops.conv(W, p + ".conv2", t, act="lrelu", residual=x)


def residual_block_with_stride(W, p, x):
    """ResidualBlockWithStride (layers.py:60-91): GDN(conv2(lrelu(conv1_s2 x))) + conv1x1_s2(x)."""
    t = ops.conv(W, p + ".conv1", x, stride=2, act="lrelu")
    t = ops.conv(W, p + ".conv2", t)
    skip = ops.conv(W, p + ".downsample", x, stride=2, pad=0)
    return ops.gdn(W, p + ".gdn", t, "intra", residual=skip)


def residual_block_upsample(W, p, x):
    """ResidualBlockUpsample (layers.py:94-119): IGDN(conv(lrelu(subpel x))) + subpel'(x)."""
    t = ops.subpel(W, p + ".subpel_conv", x, act="lrelu")
    t = ops.conv(W, p + ".conv", t)
    skip = ops.subpel(W, p + ".upsample", x)
    return ops.gdn(W, p + ".igdn", t, "intra", inverse=True, residual=skip)


def depth_conv_block(W, p, inputs, out=None, skip=None):
    """DepthConvBlock (lssvc_modules.py:15-72). `inputs` may be a list (virtual concat) when the
    block has a 1x1 adaptor (Cin != Cout); otherwise a single T (it is also the identity branch).
    `skip`: an outer skip connection added to the block's result (block(x) + skip), folded into the last launch."""
    q = p + ".block.0"
    if W.has(q + ".adaptor.weight"):
        ident = ops.conv(W, q + ".adaptor", inputs)
    else:
        assert isinstance(inputs, T), "DepthConv without adaptor needs a materialised input"
        ident = inputs
    t = ops.conv1x1_dw3x3(W, q + ".conv1.0", q + ".depth_conv", inputs, slope=0.01)     # one launch when the shape allows
    if t is None:
        t = ops.conv(W, q + ".conv1.0", inputs, act="lrelu", slope=0.01)
        t = ops.dwconv3x3(W, q + ".depth_conv", t)
    f = p + ".block.1"
    if ops.ffn_fusable(W, f, q + ".conv2", ident.C, t.C):
        # conv2 + identity + the whole ConvFFN in one launch; the 4C-wide hidden tensor never reaches HBM
        return ops.ffn_block(W, f, pre_name=q + ".conv2", pre_in=t, ident=ident, slope=0.1, out=out, skip=skip)
    o1 = ops.conv(W, q + ".conv2", t, residual=ident)
    t = ops.conv(W, f + ".conv.0", o1, act="lrelu", slope=0.1)
    if skip is None:
        return ops.conv(W, f + ".conv.2", t, act="lrelu", slope=0.1, residual=o1, out=out)
    return ops.add(ops.conv(W, f + ".conv.2", t, act="lrelu", slope=0.1, residual=o1), skip, out=out)


def pyramid_extractor(W, p, f):
    """MultiScaleTextureExtractor / FeatureExtractor / TextureExtractor (layers.py:288-308,
    dmc_net.py:11-31, lssvc_modules.py:157-200): conv+ResBlock at 1x, 1/2, 1/4."""
    l1 = res_block(W, p + ".res_block1", ops.conv(W, p + ".conv1", f))
    l2 = res_block(W, p + ".res_block2", ops.conv(W, p + ".conv2", l1, stride=2))
    l3 = res_block(W, p + ".res_block3", ops.conv(W, p + ".conv3", l2, stride=2))
    return l1, l2, l3


def context_fusion(W, p, t1, t2, t3, outs=(None, None, None)):
    """MultiScaleTextureFusion / MultiScaleContextFusion (layers.py:311-339, dmc_net.py:34-62,
    lssvc_modules.py:203-232). `outs` lets the caller place the three results (e.g. into concat slices)."""
    c3_up = res_block(W, p + ".res_block3_up", ops.subpel(W, p + ".conv3_up", t3))
    o3 = res_block(W, p + ".res_block3_out", ops.conv(W, p + ".conv3_out", t3), skip=t3, out=outs[2])        # context3 + ...
    c2_up = res_block(W, p + ".res_block2_up", ops.subpel(W, p + ".conv2_up", [c3_up, t2]))
    o2 = res_block(W, p + ".res_block2_out", ops.conv(W, p + ".conv2_out", [c3_up, t2]), skip=t2, out=outs[1])
    o1 = res_block(W, p + ".res_block1_out", ops.conv(W, p + ".conv1_out", [c2_up, t1]), skip=t1, out=outs[0])
    return o1, o2, o3


def context_homes(W, p_codec, H2, W2, c2_channels, c3_channels, device):
    """The wide buffers the bottleneck ResBlocks of a contextual encoder / decoder pair read -- cat(transform part, context) at 1/2 and
    1/4 resolution -- allocated ONCE per frame with the context living in its slice from the moment it is made: (wide2, wide3,
    context2 view, context3 view). Hand the views to context_fusion(outs=...) and the wides to res_encoder_gdn / res_decoder_gdn
    (`homes=`): the encoder and then the decoder fill the transform part, nobody copies the context (two copies each per P-frame
    layer before, 141 + 53 MB at 1080p). p_codec: the encoder's prefix, for its conv1 / conv2 output widths."""
    n2, n3 = W.raw(p_codec + ".conv1.weight").shape[0], W.raw(p_codec + ".conv2.weight").shape[0]
    w2, w3 = T.empty(H2, W2, n2 + c2_channels, device), T.empty(H2 // 2, W2 // 2, n3 + c3_channels, device)
    return w2, w3, w2.slice(n2, n2 + c2_channels), w3.slice(n3, n3 + c3_channels)


def _wide(homes, k, t, c, dev):
    """The cat(t, c) buffer of bottleneck k: the frame's resident one (context already in place) or a fresh one + a copy of c."""
    n = t.C
    if homes is not None:
        wide = homes[k]
        assert wide.C == n + c.C and c.buf is wide.buf and c.off == wide.off + n, "context %d does not live in its wide buffer" % k
        return wide, n
    wide = T.empty(t.H, t.W, n + c.C, dev)
    ops.copy(c, wide.slice(n, n + c.C))
    return wide, n


def res_encoder_gdn(W, p, x, c1, c2, c3, flavour, homes=None):
    """Contextual analysis transform with GDN: Intra ResEncoder (layers.py:342-367) and DMC ResEncoder
    (dmc_net.py:65-90). The concat that feeds each bottleneck ResBlock is materialised in place: the GDN
    writes its half of the wide buffer, the context is copied into the other half -- or already lives there (`homes`)."""
    dev = x.device
    t = ops.conv(W, p + ".conv1", [x, c1], stride=2)
    wide, n = _wide(homes, 0, t, c2, dev)
    ops.gdn(W, p + ".gdn1", t, flavour, out=wide.slice(0, n))
    f = res_block(W, p + ".res1", wide, slope=0.1, start_from_relu=False, end_with_relu=True)
    t = ops.conv(W, p + ".conv2", f, stride=2)
    wide, n = _wide(homes, 1, t, c3, dev)
    ops.gdn(W, p + ".gdn2", t, flavour, out=wide.slice(0, n))
    f = res_block(W, p + ".res2", wide, slope=0.1, start_from_relu=False, end_with_relu=True)
    t = ops.gdn(W, p + ".gdn3", ops.conv(W, p + ".conv3", f, stride=2), flavour)
    return ops.conv(W, p + ".conv4", t, stride=2)


def res_decoder_gdn(W, p, y_hat, c2, c3, flavour, homes=None):
    """Contextual synthesis transform with IGDN (layers.py:370-395, dmc_net.py:93-118)."""
    dev = y_hat.device
    t = ops.gdn(W, p + ".gdn1", ops.subpel(W, p + ".up1", y_hat), flavour, inverse=True)
    t = ops.subpel(W, p + ".up2", t)
    wide, n = _wide(homes, 1, t, c3, dev)
    ops.gdn(W, p + ".gdn2", t, flavour, inverse=True, out=wide.slice(0, n))
    f = res_block(W, p + ".res1", wide, slope=0.1, start_from_relu=False, end_with_relu=True)
    t = ops.subpel(W, p + ".up3", f)
    wide, n = _wide(homes, 0, t, c2, dev)
    ops.gdn(W, p + ".gdn3", t, flavour, inverse=True, out=wide.slice(0, n))
    f = res_block(W, p + ".res2", wide, slope=0.1, start_from_relu=False, end_with_relu=True)
    return ops.subpel(W, p + ".up4", f)


def recon_generation(W, p, res, ctx1):
    """ReconGeneration(res, ctx1) -> (feature, recon) (layers.py:398-411 / dmc_net.py:143-156; the callers
    pass (res_hat, context1), IntraSS.py:161, dmc_net.py:452, so the concat order is res first)."""
    f = ops.conv(W, p + ".feature_conv.0", [res, ctx1])
    f = res_block(W, p + ".feature_conv.1", f)
    f = res_block(W, p + ".feature_conv.2", f)
    return f, ops.conv(W, p + ".recon_conv", f)


def spynet(W, p, im1, im2):
    """ME_Spynet / ME_Spynet_DCVC (video_net_component.py:213-248,292-326)."""
    levels = 4
    l1, l2 = ops.avgpool_pyramid3(im1), ops.avgpool_pyramid3(im2)      # the two image pyramids: one launch each
    coarse = l2[levels - 1]
    flow = T.zeros(coarse.H // 2, coarse.W // 2, 2, im1.device)
    for lvl in range(levels):
        a, b = l1[levels - 1 - lvl], l2[levels - 1 - lvl]
        x = ops.spynet_prep(a, b, flow, T.empty(a.H, a.W, 8, im1.device))      # cat(im1, warp(im2, up), up), up = 2 * upsampled flow
        up = x.slice(6, 8)
        m = "%s.moduleBasic.%d" % (p, lvl)
        t = ops.conv(W, m + ".conv1", x, act="relu")
        t = ops.conv(W, m + ".conv2", t, act="relu")
        t = ops.conv(W, m + ".conv3", t, act="relu")
        t = ops.conv(W, m + ".conv4", t, act="relu")
        flow = ops.conv(W, m + ".conv5", t, residual=up)
    return flow
