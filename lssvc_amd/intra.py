"""IntraSS -- MI355X-native drop-in for the reference's I-frame scalable codec.

Mirrors the public surface test.py uses (test.py:545-547, 212, 220-223):
    IntraSS.from_state_dict(sd) -> .to(device) -> .eval() -> .set_scale_information(...) -> .encode_decode(...)
Reference: src/models/IntraSS.py:74-336 (EL) and src/models/priors.py:112-452 (`IntraNoAR`, BL).
The whole estimate-mode forward (IntraSS.py:137-172) runs as HIP kernels; tensors cross the API as
NCHW fp32 torch tensors exactly like the reference's.
"""
import math
import os as _os
import time as _time

import torch

from . import hip_ops as ops
from .hip_ops import T
from . import blocks as B
from . import bitstream, tables
from .entropy_coder import SymbolSink, SymbolSource
from .weights import WeightStore, strip_module_prefix, validate

GAUSS_IDX = tables.index_params(tables.GAUSSIAN, 1.0)       # GaussianConditional.build_indexes (+1)
LAPLACE_IDX = tables.index_params(tables.LAPLACE, 0.0)      # GaussianEncoder.build_indexes


def _channel_indexes(c, h, w):
    """BitEstimator / EntropyBottleneck build_indexes: index = channel, NCHW order (video_entropy_models.py:225-231)."""
    import numpy as np
    return np.repeat(np.arange(c, dtype=np.int32), h * w)


_CDF_BUFFERS = ("._offset", "._quantized_cdf", "._cdf_length")


SHARED_GRAPH_POOL = _os.environ.get("LSSVC_SHARED_GRAPH_POOL", "1") == "1"


class FramePlan:
    """The static launch plan of one frame type at one size: the estimate-mode forward issues a
    FIXED sequence of ~250 (I) / ~400 (P) kernel launches whose shapes depend only on the frame size, so after one
    eager call (which also lays out the weights and grants the kernels their LDS) the sequence is captured ONCE into a
    hipGraph and every later frame is: copy the inputs into the plan's static NHWC buffers, one hipGraphLaunch, one
    D2H read of the bit counters. Host work per frame drops from ~8 ms of Python + ctypes to ~0.2 ms, which is what
    binds below ~480p (configs[0], 256x256).
    Outputs are views of the graph's private memory pool: they are overwritten by the next call on the same plan
    (test.py's loop consumes each frame's outputs before coding the next one; a caller that keeps them clones them)."""

    def __init__(self, in_shapes, device):
        # in_shapes: {name: (C, H, W) or None}
        self.inputs = {k: (T.empty(v[1], v[2], v[0], device) if v is not None else None) for k, v in in_shapes.items()}
        self.calls = 0
        self.graph = None
        self.outs = None

    def load(self, tensors):
        """Copy the caller's NCHW tensors into the static NHWC input buffers (stream-ordered, no host sync)."""
        import ctypes as C
        from ._lib import lib, check
        for k, dst in self.inputs.items():
            x = tensors[k]
            if dst is None:
                assert x is None
                continue
            assert tuple(x.shape) == (1, dst.C, dst.H, dst.W) and x.dtype == torch.float32 and x.is_cuda, (k, tuple(x.shape))
            c, h, w = dst.C, dst.H, dst.W
            if c > 1 and x.stride() == (h * w * c, 1, w * c, c):
                dst.buf.view(-1)[dst.off:dst.off + c * h * w].copy_(x.permute(0, 2, 3, 1).reshape(-1))      # channels_last (our own outputs): flat copy
            else:
                x = x.contiguous()
                check(lib.lssvc_nchw_to_nhwc(C.c_void_p(x.data_ptr()), dst.ref, ops.stream_ptr()))


class _HostModel:
    """Small shared shell: device placement, eval(), scale information (IntraSS.py:229-232, LSSVC_net.py:266-269)."""

    def __init__(self):
        self.device = None
        self.W = None
        self.shape_hr = (256, 256)
        self.scale_factor = 2.0
        self.pad_size = (0, 0, 0, 0)
        self.training = False
        self._tables = None
        self._medians = {}
        self.graph_mode = _os.environ.get("LSSVC_GRAPH", "0") == "1"
        self.alias_outputs = _os.environ.get("LSSVC_GRAPH_ALIAS", "0") == "1"
        self._plans = {}
        self._graph_pools = {}    # lane (0: plans on the caller's stream, 1: the look-ahead base-layer plans) -> graph memory pool handle
        self.last_issue_s = 0.0
        self.range_audit = ops.RANGE_AUDIT_DEFAULT     # audit the first frame of every type for fp16 range (hip_ops.RangeAudit)
        self._audited = set()
        self.audit_report = {}    # frame-type key -> {layer: max |input|} of the audited frame
        self.taps = None          # diagnostics: set to a dict and every encoder pass stores its quantised latents in it

    def set_graph_mode(self, on=True, alias_outputs=False):
        """Estimate-mode frames through captured hipGraphs (FramePlan). Results are bit-identical to the eager path. The
        returned tensors are copies the caller owns, as with the reference; alias_outputs=True hands out views of the
        plan's own memory instead, which the next call of the same frame type overwrites (no copy: for loops like
        test.py's that consume a frame's outputs before coding the next frame)."""
        self.graph_mode = bool(on)
        self.alias_outputs = bool(alias_outputs)
        if not on:
            self._plans = {}
        return self

    def _own(self, t):
        """A frame output for the caller: in graph mode (unless alias_outputs) a copy outside the plan's memory."""
        if not self.graph_mode or self.alias_outputs:
            return t
        return ops.copy(t, T.empty(t.H, t.W, t.C, t.device))

    # Plans kept per model. One GEOMETRY (picture size x ratio x padding x precision) needs up to 7: I | first-P, steady-P | and the
    # look-ahead protocol's first-P, two parities, last, each as a BL and an EL plan where they differ. LSSVC_extend keeps look-ahead
    # buffers for MAX_GEOMS geometries (inter.py), so the plan cache must hold that many sets or a harness alternating two sizes would
    # evict plans before they replay -- each re-creation an eager first call, a device sync and a recapture (the default
    # was 10 against 3 geometries).
    PLANS_PER_GEOMETRY = 7
    MAX_PLANS = 3 * PLANS_PER_GEOMETRY + 2      # = inter.LSSVC_extend.MAX_GEOMS sets + the I plans of an IntraSS used on its own

    def _run_planned(self, key, tensors, body):
        """body(T inputs dict) -> dict of T outputs. First call of a key: eager. Second: capture + replay. Later: replay.
        The key is extended by everything a captured launch sequence bakes in besides the tensor shapes (scale factor,
        padded size, inter-layer padding, conv precision, single- or multi-stream order); plans are evicted least recently
        used, so a harness that walks through sizes and ratios does not pile up graph pools."""
        frame_type = key
        key = key + (float(self.scale_factor), self.shape_hr, self.pad_size, ops.CONV_PRECISION, ops.MULTI_STREAM)
        plan = self._plans.pop(key, None)
        if plan is None:
            shapes = {k: (None if v is None else tuple(v.shape[1:])) for k, v in tensors.items()}
            plan = FramePlan(shapes, self.device)
            while len(self._plans) >= self.MAX_PLANS:
                self._plans.pop(next(iter(self._plans)))             # dicts keep insertion order: the first key is the LRU
        self._plans[key] = plan                                      # (re-)insert as most recently used
        lane = 1 if frame_type[0] == "p-ahead-bl" else 0
        if plan.calls == 0:
            plan.calls = 1
            ins = {k: (None if v is None else T.from_nchw(v)) for k, v in tensors.items()}
            # The eager first call of a frame type (weights laid out, LDS granted, fp16 range audit) runs its chains in program order
            # on one stream -- same launches, same results as with side streams (tests/test_gpu_graph.py) -- because that way a freed
            # buffer is recycled at once instead of at its branch's join: 14 instead of 26 GB of working set for a 1080p P-frame. The
            # plan that follows gets its memory from a graph pool, so the allocator's cache is handed back right away rather than at
            # the next capture. Together: peak reserved HBM of the 1080p bench's priming 51 -> 3x GiB.
            streams, ops.MULTI_STREAM = ops.MULTI_STREAM, False
            try:
                r = self._with_range_audit(frame_type, lambda: body(ins))
            finally:
                ops.MULTI_STREAM = streams
            torch.cuda.synchronize(self.device)
            torch.cuda.empty_cache()
            return r
        plan.load(tensors)
        if plan.graph is None:
            torch.cuda.synchronize(self.device)
            g = torch.cuda.CUDAGraph()
            # One graph memory pool per model and LANE instead of one per plan (round 5; 98 GiB -> about a third reserved for the 1080p
            # bench): the frame plans of a model that run one after the other on the caller's stream (I; first-P / steady-P; the
            # look-ahead EL plans) share a pool -- a plan's intermediates are dead when its replay ends, its outputs stay allocated --
            # and the look-ahead BASE-LAYER plans, which run beside them on the second stream, share another. Every plan writes its
            # intermediates before it reads them and takes its inputs from buffers outside the pools, so the replay order is free.
            kw = {"pool": self._graph_pool(lane).id} if SHARED_GRAPH_POOL else {}
            with torch.cuda.graph(g, **kw):
                plan.outs = body(plan.inputs)
            plan.graph = g
        plan.graph.replay()
        plan.calls += 1
        return plan.outs

    def _graph_pool(self, lane):
        if lane not in self._graph_pools:
            if hasattr(torch.cuda, "MemPool"):
                self._graph_pools[lane] = torch.cuda.MemPool()
            else:                                                      # older torch: a pool handle has the same sharing semantics
                class _Handle:
                    id = torch.cuda.graph_pool_handle()
                self._graph_pools[lane] = _Handle()
        return self._graph_pools[lane]

    def to(self, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("lssvc_amd models run only on an MI355X HIP device (got %s); "
                               "there is no CPU fallback" % device)
        self.device = device
        self.W = WeightStore(self._sd, device)
        self.slots = ops.BitSlots(device)
        self.stage = ops.SymbolStage(device)
        return self

    def cuda(self, index=0):
        return self.to("cuda:%d" % index)

    def eval(self):
        self.training = False
        return self

    def set_scale_information(self, scale, shape_hr, pad_size):
        self.scale_factor = scale
        self.shape_hr = (int(shape_hr[0]), int(shape_hr[1]))
        self.pad_size = tuple(int(v) for v in pad_size)      # test.py:212-213 always passes (0,0,0,0)

    def _depad(self, t, p=1):
        """get_depadded_feature (IntraSS.py:124-135, LSSVC_net.py:271-282): F.pad by pad_size / p (zeros; negative = crop)."""
        return None if t is None else ops.pad_crop(t, tuple(int(v / p) for v in self.pad_size))

    def _with_range_audit(self, key, run):
        """run() -> the frame's outputs. In the f16x3 mode the first frame of type `key` is first run DRY under a RangeAudit
        (its outputs are discarded: the audited pass splits the fused DepthConvBlock kernels, whose sums are ordered
        differently, and every result handed out must come from the one normal launch sequence); layers whose input comes
        within 2x of what their fp16 staging can hold are moved to the exact fp32 kernel, with a warning; then the frame runs
        normally. Until no layer moves any more, at most three rounds (an fp32 layer changes nothing upstream of itself)."""
        if not self.range_audit or ops.CONV_PRECISION != "f16x3" or key in self._audited:
            return run()
        self._audited.add(key)
        for _ in range(3):
            audit = ops.RANGE_AUDIT = ops.RangeAudit(self.device)
            try:
                run()
            finally:
                ops.RANGE_AUDIT = None
            bad = audit.finish()
            self.audit_report[key] = audit.report
            if not bad:
                break
            import warnings
            self.W.force_f32 |= set(bad)
            self._plans = {}                   # captured launch sequences bake the kernel choice in: drop them
            worst = max(bad, key=bad.get)
            warnings.warn("fp16 range audit (%s): %d conv layer(s) moved to the exact fp32 kernel, e.g. %s with max |input| = %.3g; a "
                          "separate decoder process must be given the same set (get_f32_layers / set_f32_layers)"
                          % (key[0], len(bad), worst, bad[worst]))
        return run()

    def get_f32_layers(self):
        """Conv layers the range audit moved to the exact fp32 kernel (sorted names). With write_stream=1 the choice of kernel
        is part of what encoder and decoder must share: a decoder running in another process has to be given this list
        (set_f32_layers) before its first frame, like the checkpoint itself."""
        return sorted(self.W.force_f32)

    def set_f32_layers(self, names):
        self._require_device()
        self.W.force_f32 = set(names)
        self._plans = {}                       # captured launch sequences bake the kernel choice in

    def _tap(self, name, t):
        """Diagnostic tap (tests/test_gpu_golden_full.py): the quantised latent `t` as an int16 NCHW host tensor."""
        if self.taps is not None and t is not None:
            self.taps[name] = t.to_nchw(copy=True).round().to(torch.int16).cpu()

    # ---- write_stream = 1 plumbing: symbol planes between the kernels and the host coder (entropy_coder.py) ----
    def _begin_layer(self):
        """Start coding / decoding one layer: its int16 planes are staged in self.stage (device + pinned host)."""
        H, W = self.shape_hr
        return self.stage.begin(6 * 256 * (H // 16) * (W // 16) + 4096)

    def _push(self, sink, q, sigma, tables, idx_params=None, chunk_of_mask=None):
        """Encoder: hand the symbols of `q` (table index from `sigma`, or the channel number) to a sink."""
        sink.push(*ops.export_symbols(q, sigma, idx_params, chunk_of_mask, stage=getattr(sink, "stage", None)), tables)

    def _prefetch(self, sink):
        """Encoder, after a layer's last _push: start the planes' trip to the host now (SymbolStage.prefetch)."""
        if getattr(sink, "stage", None) is self.stage:
            self.stage.prefetch()

    def _pull(self, source, tables, out, sigma=None, idx_params=None, mean=None, channel_add=None, chunk_of_mask=None):
        """Decoder: out = decoded symbols (+ mean / per-channel medians); the table index plane comes from `sigma`
        (GaussianConditional / GaussianEncoder.build_indexes) or is the channel number (factorised tables)."""
        st = getattr(source, "stage", None)
        if sigma is not None:
            idx = ops.export_indexes(sigma, idx_params, chunk_of_mask, stage=st)
        else:
            idx = _channel_indexes(out.C, out.H, out.W)
            if st is not None:
                idx = idx.astype("int16")
        return ops.import_symbols(source.pull(idx, tables), out, mean=mean, channel_add=channel_add, chunk_of_mask=chunk_of_mask, stage=st)

    def _require_device(self):
        if self.W is None:
            raise RuntimeError("call .to('cuda:N') before encode_decode()")


def _lrelu_conv_seq(W, p, x, layout, out=None):
    """nn.Sequential of convs / subpel convs with LeakyReLU(0.01) between them; the activation is
    fused into the producing conv. layout: [(index, 'conv'|'subpel', stride), ...]."""
    last = len(layout) - 1
    for i, (idx, kind, stride) in enumerate(layout):
        act = "lrelu" if i < last else None
        o = out if i == last else None
        if kind == "conv":
            x = ops.conv(W, "%s.%d" % (p, idx), x, stride=stride, act=act, out=o)
        else:
            x = ops.subpel(W, "%s.%d" % (p, idx), x, act=act, out=o)
    return x


class IntraSS(_HostModel):
    def __init__(self, state_dict):
        super().__init__()
        self._sd = state_dict
        self.N_bl = state_dict["base_layer_model.g_s.0.conv1.weight"].shape[0]

    @classmethod
    def from_state_dict(cls, state_dict, base_layer_model_path=None):
        """IntraSS.from_state_dict (IntraSS.py:190-214): strip 'module.', drop the scale table, strict load."""
        sd = strip_module_prefix(dict(state_dict))
        if base_layer_model_path is not None:
            bl = torch.load(base_layer_model_path, map_location="cpu")
            bl = bl.get("state_dict", bl)
            for k, v in bl.items():
                sd["base_layer_model." + k] = v
        sd.pop("gaussian_conditional.scale_table", None)
        validate(sd, "intra_ss", resizable=_CDF_BUFFERS)
        return cls(sd)

    # ---------------------------------------------------------------------------------------------
    # Every codec stage below runs in one of two roles with the SAME kernels on the shared part:
    #   encoder (x given):      analysis transform -> quantise -> [push symbols to a SymbolSink] -> synthesis
    #   decoder (source given): pull symbols from a SymbolSource -> synthesis
    # so that the decoder's sigma -> table index is computed by exactly the code that the encoder ran.
    def _bl_codec(self, x_bl, sinks=None, sources=None, lat_hw=None):
        """IntraNoAR: get_layer_information (priors.py:368-388) / compress (:422-437) / decompress (:439-452).
        sinks / sources = (y_coder, z_coder): the reference writes y and z as two rANS strings."""
        W, p = self.W, "base_layer_model"
        T_ = self._tables
        N = self.N_bl
        if sources is None:
            g = p + ".g_a"
            t = B.residual_block_with_stride(W, g + ".0", x_bl)
            t = B.residual_block(W, g + ".1", t)
            t = B.residual_block_with_stride(W, g + ".2", t)
            t = B.residual_block(W, g + ".3", t)
            t = B.residual_block_with_stride(W, g + ".4", t)
            t = B.residual_block(W, g + ".5", t)
            y = ops.conv(W, g + ".6", t, stride=2)
            z = _lrelu_conv_seq(W, p + ".h_a", y, [(0, "conv", 1), (2, "conv", 1), (4, "conv", 2), (6, "conv", 1), (8, "conv", 2)])
            z_hat = z.like()
            z_q = z.like() if (sinks or self.taps is not None) else None
            ops.entropy_bottleneck(z, W.entropy_bottleneck(p + ".entropy_bottleneck"), self.slots, 1, z_hat=z_hat, z_q=z_q)
            self._tap("bl_z", z_q)
            if sinks:
                self._push(sinks[1], z_q, None, T_["bl_eb"][0])
        else:
            zh, zw = lat_hw
            z_hat = T.empty(zh, zw, N, self.device)
            self._pull(sources[1], T_["bl_eb"][0], z_hat, channel_add=self._dev_medians("bl_eb"))
        params = _lrelu_conv_seq(W, p + ".h_s", z_hat,
                                 [(0, "conv", 1), (2, "subpel", 1), (4, "conv", 1), (6, "subpel", 1), (8, "conv", 1)])
        scales, means = params.chunk(2)
        if sources is None:
            y_hat = y.like()
            y_q = y.like() if (sinks or self.taps is not None) else None
            ops.gaussian_conditional(y, scales, means, self.slots, 0, y_hat=y_hat, y_q=y_q)
            self._tap("bl_y", y_q)
            if sinks:
                self._push(sinks[0], y_q, scales, T_["gauss"], GAUSS_IDX)
                self._prefetch(sinks[0])              # the layer's last plane: its copy goes ahead of the synthesis kernels
        else:
            y_hat = T.empty(scales.H, scales.W, scales.C, self.device)
            self._pull(sources[0], T_["gauss"], y_hat, sigma=scales, idx_params=GAUSS_IDX, mean=means)
        g = p + ".g_s"
        t = B.residual_block(W, g + ".0", y_hat)
        t = B.residual_block_upsample(W, g + ".1", t)
        t = B.residual_block(W, g + ".2", t)
        t = B.residual_block_upsample(W, g + ".3", t)
        t = B.residual_block(W, g + ".4", t)
        t = B.residual_block_upsample(W, g + ".5", t)
        t = B.residual_block(W, g + ".6", t)
        x_hat = ops.subpel(W, g + ".7", t)
        return x_hat, y_hat

    def _el_contexts(self, x_hat_bl):
        """multi_scale_context_mining (IntraSS.py:119-122) of the (de-padded) base-layer reconstruction -> ctx1, ctx2, ctx3."""
        W = self.W
        H, Wd = self.shape_hr
        t = ops.conv(W, "texture_resampler.conv_adaptor.0", x_hat_bl, act="lrelu")
        t = ops.conv(W, "texture_resampler.conv_adaptor.2", t)
        tex = ops.resize(t, H, Wd)
        t1, t2, t3 = B.pyramid_extractor(W, "texture_extractor", tex)
        return B.context_fusion(W, "context_fusion_net", t1, t2, t3)

    def _el_analysis(self, xe, c1, c2, c3):
        """get_y_z_ctx's transforms (IntraSS.py:239-243): y = g_a(x, ctx), z = h_a(y)."""
        W = self.W
        y = B.res_encoder_gdn(W, "g_a", xe, c1, c2, c3, "intra")
        return y, _lrelu_conv_seq(W, "h_a", y, [(0, "conv", 1), (2, "conv", 2), (4, "conv", 2)])

    def _el_entropy(self, y, z, c3, y_hat_bl, sinks=None, sources=None, lat_hw=None):
        """The hyper codec + prior fusion + conditional coding of y: forward (IntraSS.py:150-158) / compress (:304-314) /
        the entropy half of decompress (:316-331). Encoder (y, z given): -> y_hat, symbols pushed to `sinks`; decoder: from `sources`."""
        W = self.W
        T_ = self._tables
        H, Wd = self.shape_hr
        if sources is None:
            z_hat = z.like()
            z_q = z.like() if (sinks or self.taps is not None) else None
            ops.entropy_bottleneck(z, W.entropy_bottleneck("entropy_bottleneck"), self.slots, 3, z_hat=z_hat, z_q=z_q)
            self._tap("el_z", z_q)
            if sinks:
                self._push(sinks[1], z_q, None, T_["eb"][0])
        else:
            zh, zw = lat_hw
            z_hat = T.empty(zh, zw, 64, self.device)
            self._pull(sources[1], T_["eb"][0], z_hat, channel_add=self._dev_medians("eb"))

        # PriorFusion input cat(hyper 192, layer 96, context_params 192) is written in place (layers.py:489-492)
        fused = T.empty(H // 16, Wd // 16, 480, self.device)
        _lrelu_conv_seq(W, "h_s", z_hat, [(0, "subpel", 1), (2, "subpel", 1), (4, "conv", 1)], out=fused.slice(0, 192))
        t = ops.conv(W, "layer_prior_resampler.conv_adaptor.0", y_hat_bl, act="lrelu")
        t = ops.conv(W, "layer_prior_resampler.conv_adaptor.2", t)
        ops.resize(t, H // 16, Wd // 16, out=fused.slice(192, 288))
        t = ops.conv(W, "prior_fusion_net.context_parameters.0", c3, stride=2, act="lrelu", slope=0.1)
        ops.conv(W, "prior_fusion_net.context_parameters.2", t, stride=2, out=fused.slice(288, 480))
        t = ops.conv(W, "prior_fusion_net.params_net.0", fused, act="lrelu")
        t = ops.conv(W, "prior_fusion_net.params_net.2", t, act="lrelu")
        params = ops.conv(W, "prior_fusion_net.params_net.4", t)
        scales, means = params.chunk(2)
        if sources is None:
            y_hat = y.like()
            y_q = y.like() if (sinks or self.taps is not None) else None
            ops.gaussian_conditional(y, scales, means, self.slots, 2, y_hat=y_hat, y_q=y_q)
            self._tap("el_y", y_q)
            if sinks:
                self._push(sinks[0], y_q, scales, T_["gauss"], GAUSS_IDX)
                self._prefetch(sinks[0])              # the layer's last plane: its copy goes ahead of the synthesis kernels
        else:
            y_hat = T.empty(scales.H, scales.W, scales.C, self.device)
            self._pull(sources[0], T_["gauss"], y_hat, sigma=scales, idx_params=GAUSS_IDX, mean=means)
        return y_hat

    def _el_codec(self, xe, x_hat_bl, y_hat_bl, sinks=None, sources=None, lat_hw=None):
        """IntraSS EL: forward (IntraSS.py:148-161) / compress (:304-314) / decompress (:316-336)."""
        W = self.W
        x_hat_bl, y_hat_bl = self._depad(x_hat_bl), self._depad(y_hat_bl, 16)                  # IntraSS.py:146-147
        c1, c2, c3 = self._el_contexts(x_hat_bl)
        y, z = self._el_analysis(xe, c1, c2, c3) if sources is None else (None, None)
        y_hat = self._el_entropy(y, z, c3, y_hat_bl, sinks=sinks, sources=sources, lat_hw=lat_hw)
        res_hat = B.res_decoder_gdn(W, "g_s", y_hat, c2, c3, "intra")
        feature, x_hat = B.recon_generation(W, "recon_net", res_hat, c1)
        return feature, x_hat

    # ---- the reference's lower-level EL API (round 5): same names, arguments, result keys --------------------------------------
    def get_y_z_ctx(self, x_bl, x_el):
        """IntraSS.get_y_z_ctx (IntraSS.py:239-243): x_bl is the (de-padded) base-layer RECONSTRUCTION. -> y, z, (ctx1, ctx2, ctx3), NCHW."""
        self._require_device()
        c1, c2, c3 = self._el_contexts(T.from_nchw(x_bl))
        y, z = self._el_analysis(T.from_nchw(x_el), c1, c2, c3)
        return y.to_nchw(), z.to_nchw(), (c1.to_nchw(), c2.to_nchw(), c3.to_nchw())

    def compress(self, y=None, z=None, ctx3=None, y_hat_bl=None):
        """IntraSS.compress (IntraSS.py:304-314): latents -> {"strings": [[y_string], [z_string]], "shape": z's H x W}."""
        self._require_device()
        if self._tables is None:
            raise ValueError("Uninitialized CDFs. Run update() first")
        st = self._begin_layer()
        sinks = (SymbolSink(st), SymbolSink(st))
        self._el_entropy(T.from_nchw(y), T.from_nchw(z), T.from_nchw(ctx3), T.from_nchw(y_hat_bl), sinks=sinks)
        return {"strings": [[sinks[0].flush()], [sinks[1].flush()]], "shape": tuple(z.shape[-2:])}

    def decompress(self, strings, DPB_layer, shape):
        """IntraSS.decompress (IntraSS.py:316-336): strings = [[y_string], [z_string]], DPB_layer = {x_hat_bl, y_hat_bl} (de-padded),
        shape = z's H x W. -> {"x_hat", "feature"}."""
        self._require_device()
        if self._tables is None:
            raise ValueError("Uninitialized CDFs. Run update() first")
        c1, c2, c3 = self._el_contexts(T.from_nchw(DPB_layer["x_hat_bl"]))
        st = self._begin_layer()
        y_hat = self._el_entropy(None, None, c3, T.from_nchw(DPB_layer["y_hat_bl"]), sources=(SymbolSource(strings[0][0], st), SymbolSource(strings[1][0], st)),
                                 lat_hw=(int(shape[0]), int(shape[1])))
        res_hat = B.res_decoder_gdn(self.W, "g_s", y_hat, c2, c3, "intra")
        feature, x_hat = B.recon_generation(self.W, "recon_net", res_hat, c1)
        return {"x_hat": x_hat.to_nchw(), "feature": feature.to_nchw()}

    def _frame_body(self, t):
        x_hat_bl, y_hat_bl = self._bl_codec(t["x_bl"])
        feature, x_hat = self._el_codec(t["x_el"], x_hat_bl, y_hat_bl)
        return {"x_hat_bl": x_hat_bl, "x_hat_el": x_hat, "feature_el": feature}

    def forward(self, x_bl, x_el):
        """IntraSS.forward (IntraSS.py:137-172): estimate mode."""
        self._require_device()
        H, Wd = self.shape_hr
        assert tuple(x_el.shape[2:]) == (H, Wd), "x_el is %dx%d but shape_hr is %dx%d" % (x_el.shape[2], x_el.shape[3], H, Wd)
        tensors = {"x_bl": x_bl, "x_el": x_el}
        t_issue = _time.perf_counter()
        if self.graph_mode:
            r = self._run_planned(("i", tuple(x_bl.shape), tuple(x_el.shape)), tensors, self._frame_body)
        else:
            ins = {k: T.from_nchw(v) for k, v in tensors.items()}
            r = self._with_range_audit(("i", tuple(x_bl.shape), tuple(x_el.shape)), lambda: self._frame_body(ins))
        self.last_issue_s = _time.perf_counter() - t_issue       # host time to put the frame on the stream (no GPU wait)
        x_hat_bl, x_hat, feature = self._own(r["x_hat_bl"]), self._own(r["x_hat_el"]), self._own(r["feature_el"])
        out = {"x_hat_bl": x_hat_bl.to_nchw(remember=True), "x_hat_el": x_hat.to_nchw(remember=True), "feature_el": feature.to_nchw(remember=True)}
        s = self.slots.fetch()
        out["bit_bl"] = (s[0] + s[1]) / (-math.log(2))
        out["bit_el"] = (s[2] + s[3]) / (-math.log(2))
        return out

    def encode_decode(self, x_bl, x_el, bin_path_bl, bin_path_el,
                      pic_height_bl=None, pic_width_bl=None, pic_height_el=None, pic_width_el=None):
        """IntraSS.encode_decode (IntraSS.py:245-302). bin_path None <=> estimate mode; otherwise both layers are
        written to real bitstreams, read back and DECODED, and the decoded tensors are returned."""
        if bin_path_bl is None:
            return self.forward(x_bl, x_el)
        self._require_device()
        if self._tables is None:
            raise ValueError("Uninitialized CDFs. Run update() first")       # img_entropy_models.py:265-267
        assert pic_height_el is not None and pic_width_el is not None
        xb, xe = T.from_nchw(x_bl), T.from_nchw(x_el)
        self._with_range_audit(("i", tuple(x_bl.shape), tuple(x_el.shape)), lambda: self._frame_body({"x_bl": xb, "x_el": xe}))   # first frame only
        # ---- encode ----
        st = self._begin_layer()
        sinks = (SymbolSink(st), SymbolSink(st))
        x_hat_bl_e, y_hat_bl_e = self._bl_codec(xb, sinks=sinks)
        bitstream.encode_i(pic_height_bl, pic_width_bl, sinks[0].flush(), sinks[1].flush(), bin_path_bl)
        bit_bl = bitstream.filesize(bin_path_bl) * 8
        st = self._begin_layer()
        sinks = (SymbolSink(st), SymbolSink(st))
        feature_e, x_hat_e = self._el_codec(xe, x_hat_bl_e, y_hat_bl_e, sinks=sinks)
        bitstream.encode_i(pic_height_el, pic_width_el, sinks[0].flush(), sinks[1].flush(), bin_path_el)
        bit_el = bitstream.filesize(bin_path_el) * 8
        est = self.slots.fetch()
        # ---- decode ----
        h, w, y_string, z_string = bitstream.decode_i(bin_path_bl)
        st = self._begin_layer()
        x_hat_bl, y_hat_bl = self._bl_codec(None, sources=(SymbolSource(y_string, st), SymbolSource(z_string, st)),
                                            lat_hw=bitstream.get_downsampled_shape(h, w, 64))
        h, w, y_string, z_string = bitstream.decode_i(bin_path_el)
        st = self._begin_layer()
        feature, x_hat = self._el_codec(None, x_hat_bl, y_hat_bl, sources=(SymbolSource(y_string, st), SymbolSource(z_string, st)),
                                        lat_hw=bitstream.get_downsampled_shape(h, w, 64))
        return {"bit_bl": bit_bl, "bit_el": bit_el, "x_hat_bl": x_hat_bl.to_nchw(), "x_hat_el": x_hat.to_nchw(),
                "feature_el": feature.to_nchw(),
                # extras (not in the reference's dict): estimated bits and the encoder-side reconstructions
                "bit_bl_estimate": (est[0] + est[1]) / (-math.log(2)), "bit_el_estimate": (est[2] + est[3]) / (-math.log(2)),
                "encoder_side": {"x_hat_bl": x_hat_bl_e.to_nchw(), "x_hat_el": x_hat_e.to_nchw(), "feature_el": feature_e.to_nchw()}}

    def encode(self, x_bl, x_el, bin_path_bl, bin_path_el, pic_height_bl, pic_width_bl, pic_height_el, pic_width_el):
        """Encoder only: the compress half of encode_decode above (IntraNoAR.compress priors.py:422-437 + IntraSS.compress
        IntraSS.py:304-314) -- writes the two layer files and returns the encoder-side reconstruction, which is bit for bit
        what decode() makes of those files (tests/test_gpu_stream.py), so an encoder process never has to decode. The host
        coder runs while the GPU is still busy with the synthesis transforms (SymbolStage.prefetch)."""
        self._require_device()
        if self._tables is None:
            raise ValueError("Uninitialized CDFs. Run update() first")
        xb, xe = T.from_nchw(x_bl), T.from_nchw(x_el)
        self._with_range_audit(("i", tuple(x_bl.shape), tuple(x_el.shape)), lambda: self._frame_body({"x_bl": xb, "x_el": xe}))
        st = self._begin_layer()
        sinks = (SymbolSink(st), SymbolSink(st))
        x_hat_bl, y_hat_bl = self._bl_codec(xb, sinks=sinks)
        bitstream.encode_i(pic_height_bl, pic_width_bl, sinks[0].flush(), sinks[1].flush(), bin_path_bl)
        st = self._begin_layer()
        sinks = (SymbolSink(st), SymbolSink(st))
        feature, x_hat = self._el_codec(xe, x_hat_bl, y_hat_bl, sinks=sinks)
        bitstream.encode_i(pic_height_el, pic_width_el, sinks[0].flush(), sinks[1].flush(), bin_path_el)
        return {"bit_bl": bitstream.filesize(bin_path_bl) * 8, "bit_el": bitstream.filesize(bin_path_el) * 8,
                "x_hat_bl": x_hat_bl.to_nchw(), "x_hat_el": x_hat.to_nchw(), "feature_el": feature.to_nchw()}

    def decode(self, bin_path_bl, bin_path_el):
        """Decoder only: reconstruct an I-frame from its two layer files (the decode half of encode_decode above =
        IntraNoAR.decompress priors.py:439-452 + IntraSS.decompress IntraSS.py:316-336). Needs set_scale_information()
        and update() like the encoder; everything else comes from the streams (their headers carry the picture size).
        Returns the same keys a caller builds the DPB from: x_hat_bl, x_hat_el, feature_el."""
        self._require_device()
        if self._tables is None:
            raise ValueError("Uninitialized CDFs. Run update() first")
        h, w, y_string, z_string = bitstream.decode_i(bin_path_bl)
        st = self._begin_layer()
        x_hat_bl, y_hat_bl = self._bl_codec(None, sources=(SymbolSource(y_string, st), SymbolSource(z_string, st)),
                                            lat_hw=bitstream.get_downsampled_shape(h, w, 64))
        h, w, y_string, z_string = bitstream.decode_i(bin_path_el)
        st = self._begin_layer()
        feature, x_hat = self._el_codec(None, x_hat_bl, y_hat_bl, sources=(SymbolSource(y_string, st), SymbolSource(z_string, st)),
                                        lat_hw=bitstream.get_downsampled_shape(h, w, 64))
        return {"x_hat_bl": x_hat_bl.to_nchw(remember=True), "x_hat_el": x_hat.to_nchw(remember=True),
                "feature_el": feature.to_nchw(remember=True)}

    def update(self, force=False):
        """IntraSS.update (IntraSS.py:234-237): build the CDF tables the real bitstream needs (test.py:561-564)."""
        if self._tables is not None and not force:
            return
        self._tables = {"gauss": tables.gaussian_tables(), "eb": tables.bottleneck_tables(self._sd, "entropy_bottleneck"),
                        "bl_eb": tables.bottleneck_tables(self._sd, "base_layer_model.entropy_bottleneck")}
        self._medians = {}

    def _dev_medians(self, which):
        if which not in self._medians:
            self._medians[which] = self.W._dev(torch.from_numpy(self._tables[which][1]))      # (registered: plan_compiler stores it with the weights)
        return self._medians[which]
