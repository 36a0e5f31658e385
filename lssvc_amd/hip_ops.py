"""Host-side operator layer: NHWC tensor views over torch device memory + thin wrappers that fill
the C-ABI descriptors of include/lssvc_hip.h and enqueue the HIP kernels on torch's current stream.

PyTorch is used here only as the device allocator / stream provider (plumbing); every FLOP of
the hot path runs in liblssvc_hip.so.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import lib, check, View, ConvDesc

_NULL_VIEW = View(None, 0, 0, 0, 0)

# Conv arithmetic for the MFMA-bound layers (3x3 / 7x7, stride 1): "f16x3" (default: fp16 MFMA on hi/lo-split
# operands, fp32 accumulate -- fp32-class accuracy, holds the parity bars, ~2-3x faster) or "f32" (exact fp32
# MFMA everywhere). Set with set_conv_precision() or env LSSVC_CONV_PRECISION.
import os as _os
CONV_PRECISION = _os.environ.get("LSSVC_CONV_PRECISION", "f16x3")


def set_conv_precision(mode):
    global CONV_PRECISION
    if mode not in ("f32", "f16x3"):
        raise ValueError("conv precision must be 'f32' or 'f16x3'")
    CONV_PRECISION = mode


# optional op log: list of (kind, name, macs) appended by conv() when enabled (bench / profiling)
OP_LOG = None
CONV_CHECK = _os.environ.get("LSSVC_CONV_CHECK", "0") == "1"


def reserve_device_memory(device, gib=None):
    """Grow PyTorch's caching allocator to `gib` GiB in one hipMalloc and hand the block back to the cache. The codec
    allocates and frees hundreds of activation buffers per frame; without this the pool keeps growing (synchronous
    hipMalloc calls) over the first few GOPs -- a 1080p two-layer P-frame peaks around 10 GiB, far inside the 288 GiB."""
    if gib is None:
        gib = float(_os.environ.get("LSSVC_RESERVE_GIB", "12"))
    free, _total = torch.cuda.mem_get_info(device)
    want = min(int(gib * 2 ** 30), int(free * 0.5))
    if want > 0:
        block = torch.empty(want, dtype=torch.uint8, device=device)
        del block


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# ---- side streams: independent chains of one frame as parallel branches ---------------------------------------------
# A frame is a fixed DAG of ~400 launches, most of them on maps so small that one launch cannot fill 256 CUs (the prior /
# hyper / motion-vector codecs). Chains that do not depend on each other (EL SpyNet || the whole BL codec; the BL-texture
# pyramid || the EL motion-vector codec; ...) are issued on side streams between fork/join events, so that under a
# captured frame plan they become parallel branches of the hipGraph and in eager mode concurrent HIP streams. The kernels
# and their per-chain order are unchanged, so results are bit-identical to the single-stream order.
#
# Memory: PyTorch's caching allocator recycles a freed block for the next allocation on the block's OWN stream, which is
# only safe if every other stream that touched the block has been joined since; the plan compiler's arena hands a freed
# block to whichever stream allocates next. Every buffer touched by a launch inside branch i is therefore kept alive until
# branch i is JOINED: from then on the joining stream has waited for the branch, and every later allocation is either on
# that stream or inside a branch context, whose entry waits for that stream again. (Rounds 2-3 kept them to the end of the
# frame body, which made a P-frame plan's arena 17.6 GB at 1080p.)
MULTI_STREAM = _os.environ.get("LSSVC_STREAMS", "1") == "1"
import threading as _threading


class _Tls(_threading.local):
    fork = None           # the Fork whose branch is open (None outside branches)
    keep = None           # buffers touched while a branch is open (None outside branches); per host thread, because several
    #                       GOPs may be in flight on one GPU, each coded by its own thread on its own stream (bench.py)


_TLS = _Tls()
_SIDE_STREAMS = {}        # (device, main stream) -> its side streams


class _Branch:
    def __init__(self, fork, i):
        self.fork, self.i, self.ctx, self.outer_keep = fork, i, None, None

    def __enter__(self):
        f = self.fork
        if f.enabled:
            # (a branch of ANOTHER Fork may be open around this one -- the look-ahead base layer runs as one branch of the frame and
            # forks its own chains inside it: that Fork must have been made with its own `tag`, i.e. its own side streams)
            assert _TLS.keep is None or _TLS.fork is not f, "branches of one Fork do not nest"
            self.outer_keep, self.outer_fork = _TLS.keep, _TLS.fork
            s = f.streams[self.i]
            s.wait_stream(torch.cuda.current_stream())
            if PLAN_RECORDER is not None:
                PLAN_RECORDER.wait(s.cuda_stream, torch.cuda.current_stream().cuda_stream)
            self.ctx = torch.cuda.stream(s)
            self.ctx.__enter__()
            _TLS.keep = f.keep.setdefault(self.i, [])
            _TLS.fork = f
            f.open.add(self.i)
        return self

    def __exit__(self, *exc):
        if self.fork.enabled:
            _TLS.keep, _TLS.fork = self.outer_keep, self.outer_fork
            self.ctx.__exit__(*exc)
        return False


class Fork:
    """Fork/join bookkeeping of one frame body: `with fork.branch(i): ...` issues the enclosed launches on side stream i
    (which first waits for everything issued so far on the current stream); `fork.join(i)` makes the current stream wait
    for that branch. A branch index may be reused after its join. Disabled (LSSVC_STREAMS=0 / enabled=False) it is a no-op
    and everything runs in program order on the current stream."""

    def __init__(self, device, enabled=None, n=3, tag=""):
        self.enabled = MULTI_STREAM if enabled is None else bool(enabled)
        self.keep, self.open = {}, set()          # keep: branch index -> buffers its launches have touched since its last join
        if self.enabled:
            main = torch.cuda.current_stream(device)
            key = (device.index, n, tag, main.cuda_stream if not torch.cuda.is_current_stream_capturing() else "capture")
            if key not in _SIDE_STREAMS:
                _SIDE_STREAMS[key] = [torch.cuda.Stream(device) for _ in range(n)]
            self.streams = _SIDE_STREAMS[key]

    def branch(self, i):
        return _Branch(self, i)

    def join(self, i):
        assert _TLS.keep is None or _TLS.fork is not self, "join from the forking stream, not from inside a branch of the same Fork"
        if self.enabled and i in self.open:
            torch.cuda.current_stream().wait_stream(self.streams[i])
            if PLAN_RECORDER is not None:
                PLAN_RECORDER.wait(torch.cuda.current_stream().cuda_stream, self.streams[i].cuda_stream)
            self.open.discard(i)
            done = self.keep.pop(i, None)          # the joining stream has waited for the branch: its buffers may be recycled ...
            if done and _TLS.keep is not None:
                _TLS.keep.extend(done)             # ... unless the joining stream is itself a branch (of another Fork): then with that one

    def close(self):
        for i in sorted(self.open):
            self.join(i)


# ---- fp16 range audit of the f16x3 conv mode ------------------------------------------------------------------------------
# The f16x3 kernels split every activation into fp16 hi + lo parts while staging and SATURATE at +-65504 (GDN's 1x1 squares
# first, so its limit is 255.9); the reference computes in fp32 and has no such limit. Pretrained checkpoints are not
# available here, so instead of trusting that "all tensors on this path are O(1-100)" the models run the FIRST frame of every
# frame type (I / first-P / steady-P, per size) under a RangeAudit: max |x| of every f16x3 conv input is reduced on the device
# (lssvc_absmax; the fused DepthConvBlock kernels are split into their convs for that one frame so that their internal
# tensors are seen too), and every layer whose input comes within a factor two of the limit is moved to the exact fp32
# kernel for good, with a warning, and the frame is recomputed. LSSVC_RANGE_AUDIT=0 switches the audit off.
RANGE_AUDIT = None                    # the RangeAudit of the frame being audited, else None
RANGE_AUDIT_DEFAULT = _os.environ.get("LSSVC_RANGE_AUDIT", "1") == "1"
F16_INPUT_LIMIT = 2.0 ** 15           # half of fp16's largest finite value: a 2x margin for the frames that follow
F16_SQUARE_INPUT_LIMIT = 2.0 ** 7.5   # GDN squares while staging: x^2 <= 2^15


class RangeAudit:
    def __init__(self, device, max_layers=4096):
        self.slots = torch.zeros(max_layers, dtype=torch.float32, device=device)
        self.names, self.limits = [], []
        self.report = {}

    def watch(self, name, tensors, squared):
        i = len(self.names)
        if i >= self.slots.numel():
            raise RuntimeError("RangeAudit: more than %d audited layers in one frame" % self.slots.numel())
        self.names.append(name)
        self.limits.append(F16_SQUARE_INPUT_LIMIT if squared else F16_INPUT_LIMIT)
        ptr = C.c_void_p(self.slots.data_ptr() + 4 * i)
        for t in tensors:
            check(lib.lssvc_absmax(t.ref, ptr, stream_ptr()))

    def finish(self):
        """-> {layer name: max |input|} of the layers over their limit; self.report holds every audited layer."""
        torch.cuda.synchronize(self.slots.device)
        vals = self.slots[:len(self.names)].cpu().tolist()
        bad = {}
        for name, v, lim in zip(self.names, vals, self.limits):
            self.report[name] = max(v, self.report.get(name, 0.0))
            if not v < lim:
                bad[name] = v
        return bad


POISON_EMPTY = _os.environ.get("LSSVC_POISON_EMPTY", "0") == "1"
GUARD_EMPTY = False
_GUARDS = []


def check_guards():
    """-> [(shape, 'before' | 'after', first bad offset)] of the guarded buffers (GUARD_EMPTY) whose sentinel zones were written."""
    bad = []
    for whole, G, shape in _GUARDS:
        for side, zone in (("before", whole[:G]), ("after", whole[-G:])):
            ne = (zone != 12345.0).nonzero()
            if ne.numel():
                bad.append((shape, side, int(ne[0]) if side == "after" else int(ne[-1]) - G, int(ne.numel())))
    _GUARDS.clear()
    return bad
ARENA = None                  # plan_compiler.Arena while a frame plan is being recorded
PLAN_RECORDER = None          # plan_compiler.Recorder while a frame plan is being recorded (fork / join edges are reported to it)


class T:
    """An H x W x C fp32 view (batch 1) with pixel pitch `ld` into a flat torch buffer."""
    __slots__ = ("buf", "H", "W", "C", "ld", "off", "_v", "split")

    def __init__(self, buf, H, W, Cc, ld, off=0, split=False):
        self.buf, self.H, self.W, self.C, self.ld, self.off = buf, H, W, Cc, ld, off
        self.split = split        # True: a PRE-SPLIT tensor (lssvc_hip.h LSSVC_PREC_SPLIT_IN): fp16 hi | lo per 16-channel chunk, only 3x3 f16x3 convs read it
        self._v = View(buf.data_ptr() + 4 * off, H, W, Cc, ld)

    @staticmethod
    def empty(H, W, Cc, device):
        if ARENA is not None:                    # a frame plan is being compiled (plan_compiler.py): activations from ONE arena
            return T(ARENA.alloc_f32(H * W * Cc), H, W, Cc, Cc)
        if GUARD_EMPTY:              # debugging aid: every activation buffer between two sentinel zones, checked by check_guards()
            G = 1024
            whole = torch.full((H * W * Cc + 2 * G,), 12345.0, dtype=torch.float32, device=device)
            _GUARDS.append((whole, G, (H, W, Cc)))
            return T(whole, H, W, Cc, Cc, off=G)
        if POISON_EMPTY:             # debugging aid (LSSVC_POISON_EMPTY=1): fresh activation buffers start as NaN, so a read of
            return T(torch.full((H * W * Cc,), float("nan"), dtype=torch.float32, device=device), H, W, Cc, Cc)      # unwritten memory shows
        return T(torch.empty(H * W * Cc, dtype=torch.float32, device=device), H, W, Cc, Cc)

    @staticmethod
    def zeros(H, W, Cc, device):
        t = T.empty(H, W, Cc, device)
        check(lib.lssvc_fill_zero(C.c_void_p(t.buf.data_ptr() + 4 * t.off), 4 * H * W * Cc, stream_ptr()))     # a library launch, so plans see it
        return t

    @property
    def device(self):
        return self.buf.device

    @property
    def v(self):
        k = _TLS.keep
        if k is not None:
            k.append(self.buf)              # touched by a side-stream launch: alive until that branch is joined
        return self._v

    @property
    def ref(self):
        k = _TLS.keep
        if k is not None:
            k.append(self.buf)
        return C.byref(self._v)

    def slice(self, c0, c1):
        assert 0 <= c0 < c1 <= self.C
        return T(self.buf, self.H, self.W, c1 - c0, self.ld, self.off + c0)

    def chunk(self, n):
        step = self.C // n
        return [self.slice(i * step, (i + 1) * step) for i in range(n)]

    def like(self, Cc=None):
        return T.empty(self.H, self.W, self.C if Cc is None else Cc, self.device)

    # ---- boundary layout (the reference hands NCHW tensors across its model API) -------------------
    @staticmethod
    def from_nchw(x):
        """A (1,C,H,W) fp32 device tensor as an NHWC view. A channels_last tensor (what to_nchw() hands out) is wrapped in
        place; a plain NCHW-contiguous one (the caller's input frames) is transposed once."""
        assert x.dim() == 4 and x.shape[0] == 1 and x.dtype == torch.float32 and x.is_cuda, \
            "expected a (1,C,H,W) fp32 device tensor, got %s %s %s" % (tuple(x.shape), x.dtype, x.device)
        _, c, h, w = x.shape
        if c > 1 and x.stride() == (h * w * c, 1, w * c, c) and x.data_ptr() % 16 == 0:
            return T(x.permute(0, 2, 3, 1).reshape(-1), h, w, c, c)          # zero-copy: same storage, NHWC order
        x = x.contiguous()
        t = T.empty(h, w, c, x.device)
        check(lib.lssvc_nchw_to_nhwc(C.c_void_p(x.data_ptr()), t.ref, stream_ptr()))
        return t

    def to_nchw(self, remember=False, copy=False):
        """The view as a (1,C,H,W) tensor for the caller. Dense views are returned WITHOUT a copy as a channels_last
        tensor over the same storage (torch.channels_last: shape NCHW, memory NHWC), so the caller's in-place edits
        (test.py:249-250 clamps the reconstructions) act on this buffer and handing the tensor back in costs nothing.
        copy=True (or a channel slice / single-channel view) gives an independent NCHW-contiguous tensor."""
        if not copy and self.C > 1 and self.ld == self.C and self.off % 4 == 0 and self.buf.numel() >= self.off + self.H * self.W * self.C:
            flat = self.buf.view(-1)[self.off:self.off + self.H * self.W * self.C]
            return flat.view(1, self.H, self.W, self.C).permute(0, 3, 1, 2)
        out = torch.empty(1, self.C, self.H, self.W, dtype=torch.float32, device=self.device)
        check(lib.lssvc_nhwc_to_nchw(self.ref, C.c_void_p(out.data_ptr()), stream_ptr()))
        return out

    def torch_hwc(self):
        """Debug/test helper: a dense (H,W,C) torch copy."""
        full = self.buf.view(-1)[self.off:self.off + (self.H * self.W - 1) * self.ld + self.C]
        if self.ld == self.C:
            return full.view(self.H, self.W, self.C).clone()
        idx = (torch.arange(self.H * self.W, device=self.device)[:, None] * self.ld
               + torch.arange(self.C, device=self.device)[None, :])
        return self.buf.view(-1)[self.off + idx].view(self.H, self.W, self.C)


_ACT = {None: _lib.ACT_NONE, "lrelu": _lib.ACT_LRELU, "relu": _lib.ACT_RELU}
_INACT = {None: _lib.INACT_NONE, "lrelu": _lib.INACT_LRELU, "square": _lib.INACT_SQUARE}


def _conv_launch(inputs, prepared, KH, KW, stride, pad_t, pad_l, out, *, in_act=None, in_slope=0.01, epilogue=0,
                 gdn_x=None, act=None, slope=0.01, residual=None, out_scale=1.0, pixel_shuffle=False, name="", w16=None,
                 residual2=None):
    w_dev, b_dev, cout, m_pad = prepared
    d = ConvDesc()
    for i, t in enumerate(inputs):
        d.inp[i] = t.v
    d.n_in = len(inputs)
    d.weight = w_dev.data_ptr()
    d.bias = b_dev.data_ptr() if b_dev is not None else None
    d.KH, d.KW, d.stride, d.pad_t, d.pad_l = KH, KW, stride, pad_t, pad_l
    d.Cout, d.M_pad = cout, m_pad
    d.in_act, d.in_slope = _INACT[in_act], in_slope
    d.epilogue = epilogue
    d.gdn_x = gdn_x.v if gdn_x is not None else _NULL_VIEW
    d.act, d.slope = _ACT[act], slope
    d.residual = residual.v if residual is not None else _NULL_VIEW
    d.residual2 = residual2.v if residual2 is not None else _NULL_VIEW
    d.out_scale = out_scale
    d.pixel_shuffle = 1 if pixel_shuffle else 0
    d.out = out.v
    if RANGE_AUDIT is not None and w16 is not None:
        RANGE_AUDIT.watch(name, inputs, in_act == "square")
    if w16 is not None and CONV_CHECK:
        # debug aid (LSSVC_CONV_CHECK=1): run the launch in fp32 into a scratch output first, compare afterwards
        scratch = T.empty(out.H, out.W, out.C, out.device)
        d.out = scratch.v
        check(lib.lssvc_conv2d(C.byref(d), stream_ptr()))
        d.out = out.v
        d.precision, d.weight16, d.weight16_unscale = _lib.PREC_F16X3, w16[0].data_ptr(), w16[1]
        check(lib.lssvc_conv2d(C.byref(d), stream_ptr()))
        a, b = scratch.torch_hwc(), out.torch_hwc()
        err = (a - b).abs().max().item()
        ref = a.abs().max().item()
        if not err <= 1e-4 * max(ref, 1.0):
            print("CONV_CHECK %s k%d in=%s out=%dx%dx%d(ld %d) kernel=%s: max|f16x3-f32| %.3e (max|f32| %.3e) res=%s ps=%s"
                  % (name, KH, [(t.C, t.ld) for t in inputs], out.H, out.W, out.C, out.ld,
                     lib.lssvc_conv2d_last_kernel().decode(), err, ref, residual is not None, pixel_shuffle), flush=True)
        return out
    if w16 is not None:
        d.precision, d.weight16, d.weight16_unscale = _lib.PREC_F16X3, w16[0].data_ptr(), w16[1]
    if any(t.split for t in inputs):
        assert w16 is not None and all(t.split for t in inputs) and in_act is None, "pre-split inputs: f16x3 conv, all inputs, no input activation"
        d.precision |= _lib.PREC_SPLIT_IN
    if OP_LOG is None:
        check(lib.lssvc_conv2d(C.byref(d), stream_ptr()))
        return out
    # profiling mode (bench.py): bracket the launch with HIP events on the launch stream and log the
    # algorithmic MACs (true Cin/Cout, no padding) and the kernel instantiation that ran
    hout, wout = (out.H // 2, out.W // 2) if pixel_shuffle else (out.H, out.W)
    cin = sum(t.C for t in inputs)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(lib.lssvc_conv2d(C.byref(d), stream_ptr()))
    e1.record()
    OP_LOG.append({"kind": "conv%dx%ds%d" % (KH, KW, stride), "name": name, "macs": hout * wout * cout * KH * KW * cin,
                   "hout": hout, "wout": wout, "cin": cin, "cout": cout,
                   "variant": lib.lssvc_conv2d_variant(hout, wout, m_pad, stride), "ks": KH, "stride": stride,
                   "vec": all(t.C % 4 == 0 and t.ld % 4 == 0 and t.v.ptr % 16 == 0 for t in inputs),
                   "f16x3": w16 is not None, "kernel": lib.lssvc_conv2d_last_kernel().decode(),
                   "bytes": 4 * (hout * wout * cin * (stride * stride) + out.H * out.W * out.C
                                 + (out.H * out.W * out.C if residual is not None else 0)
                                 + (out.H * out.W * out.C if residual2 is not None else 0)),
                   "events": (e0, e1)})
    return out


PAD_NARROW_INPUTS = _os.environ.get("LSSVC_PAD_NARROW", "1") == "1"


def pad4(t):
    """A 4-channel copy of a 1-3 channel view (extra channels zero). Not cached: the source may be rewritten in place."""
    return copy(t, T.empty(t.H, t.W, 4, t.device))          # one launch: lssvc_copy writes the channels the source lacks as zeros


def conv(W, name, inputs, *, stride=1, act=None, slope=0.01, in_act=None, in_slope=0.01, residual=None,
         pixel_shuffle=False, out_scale=1.0, out=None, pad=None, residual2=None):
    """nn.Conv2d (+ optional fused pieces). `inputs`: a T or a list of up to 3 T's read as torch.cat(dim=1)."""
    if isinstance(inputs, T):
        inputs = [inputs]
    splits = [t.C for t in inputs]                      # what the weights are laid out for
    if CONV_PRECISION == "f16x3" and PAD_NARROW_INPUTS and any(t.C < 4 for t in inputs):
        # RGB / flow inputs (2-3 channels) would force the whole conv onto the exact-fp32 kernel (its loads are not
        # 16-byte addressable); a zero 4th channel costs one small copy and meets zero weights (every concat segment
        # is zero-padded to its chunk size in both weight layouts), so the result is unchanged
        inputs = [pad4(t) if t.C < 4 else t for t in inputs]
    w_dev, b_dev, cout, m_pad, KH, KW = W.conv(name, splits, pixel_shuffle)
    if pad is None:
        pad = KH // 2
    x = inputs[0]
    hout = (x.H + 2 * pad - KH) // stride + 1
    wout = (x.W + 2 * pad - KW) // stride + 1
    if out is None:
        out = T.empty(hout * 2, wout * 2, cout // 4, x.device) if pixel_shuffle else T.empty(hout, wout, cout, x.device)
    w16 = None
    if CONV_PRECISION == "f16x3" and (stride == 1 and KH in (1, 3, 7) or stride == 2 and KH == 3) \
            and name not in W.force_f32 and all(t.C % 4 == 0 and t.ld % 4 == 0 for t in inputs):
        w16 = W.conv_f16x3(name, splits, pixel_shuffle)
    return _conv_launch(inputs, (w_dev, b_dev, cout, m_pad), KH, KW, stride, pad, pad, out, in_act=in_act,
                        in_slope=in_slope, act=act, slope=slope, residual=residual, out_scale=out_scale,
                        pixel_shuffle=pixel_shuffle, name=name, w16=w16, residual2=residual2)


def subpel(W, name, inputs, **kw):
    """subpel_conv3x3 / subpel_conv1x1: conv -> PixelShuffle(2) fused in the store; `name` is the nn.Sequential."""
    return conv(W, name + ".0", inputs, pixel_shuffle=True, **kw)


def conv_t(W, name, x, stride, *, act=None, slope=0.01, out=None):
    """nn.ConvTranspose2d(k=3, padding=1 [, stride=2, output_padding=1]) via its equivalent conv."""
    w_dev, b_dev, cout, m_pad, KH, KW, pad, ps = W.conv_t(name, stride)
    if out is None:
        out = T.empty(x.H * 2, x.W * 2, cout // 4, x.device) if ps else T.empty(x.H, x.W, cout, x.device)
    return _conv_launch([x], (w_dev, b_dev, cout, m_pad), KH, KW, 1, pad, pad, out, act=act, slope=slope,
                        pixel_shuffle=ps, name=name)


_GDN_EPI = {("intra", False): _lib.EPI_X_MUL_RSQRT, ("intra", True): _lib.EPI_X_MUL_SQRT,
            ("inter", False): _lib.EPI_X_DIV_SQRT, ("inter", True): _lib.EPI_X_MUL_SQRT}


GDN_F16X3 = _os.environ.get("LSSVC_GDN_F16X3", "1") == "1"


def gdn(W, name, x, flavour, inverse=False, *, residual=None, act=None, slope=0.01, out=None):
    """GDN / IGDN as a 1x1 conv on x^2 with the normalisation fused into the epilogue."""
    w_dev, b_dev, cout, m_pad, _, _ = W.gdn(name, flavour)
    if out is None:
        out = x.like()
    w16 = None
    if CONV_PRECISION == "f16x3" and GDN_F16X3 and x.C % 4 == 0 and x.ld % 4 == 0 and name not in W.force_f32:
        w16 = W.gdn_f16x3(name, flavour)
    return _conv_launch([x], (w_dev, b_dev, cout, m_pad), 1, 1, 1, 0, 0, out, in_act="square",
                        epilogue=_GDN_EPI[(flavour, inverse)], gdn_x=x, act=act, slope=slope, residual=residual,
                        name=name, w16=w16)


def dwconv3x3(W, name, x, out=None):
    w_dev, b_dev = W.dwconv(name)
    if out is None:
        out = x.like()
    check(lib.lssvc_dwconv3x3(x.ref, C.c_void_p(w_dev.data_ptr()), C.c_void_p(b_dev.data_ptr()), out.ref, stream_ptr()))
    return out


FUSE_DW = _os.environ.get("LSSVC_FUSE_DW", "1") == "1"


def conv1x1_dw3x3(W, conv_name, dw_name, inputs, *, slope=0.01, out=None):
    """depthwise3x3(lrelu(conv1x1(cat(inputs)))) in one launch (DepthConv.conv1 -> depth_conv, lssvc_modules.py:15-44),
    or None if this shape is not covered (the caller then issues the two ops)."""
    if isinstance(inputs, T):
        inputs = [inputs]
    if not (FUSE_DW and CONV_PRECISION == "f16x3") or len(inputs) > 3 or RANGE_AUDIT is not None or conv_name in W.force_f32:
        return None                      # (an audited frame runs the two ops separately so that the tensor between them is seen)
    if any(t.C % 4 or t.ld % 4 for t in inputs) or sum((t.C + 15) // 16 for t in inputs) > 4:
        return None
    w_dev, b_dev, cout, m_pad, KH, KW = W.conv(conv_name, [t.C for t in inputs], False)
    if KH != 1 or cout not in (32, 48, 64) or b_dev is None:
        return None
    w16 = W.conv_f16x3(conv_name, [t.C for t in inputs], False)
    dw_w, dw_b = W.dwconv(dw_name)
    x = inputs[0]
    if out is None:
        out = T.empty(x.H, x.W, cout, x.device)
    d = ConvDesc()
    for i, t in enumerate(inputs):
        d.inp[i] = t.v
    d.n_in = len(inputs)
    d.weight, d.bias = w_dev.data_ptr(), b_dev.data_ptr()
    d.KH, d.KW, d.stride, d.pad_t, d.pad_l = 1, 1, 1, 0, 0
    d.Cout, d.M_pad = cout, m_pad
    d.in_act, d.in_slope = _INACT[None], 0.01
    d.epilogue = 0
    d.gdn_x = _NULL_VIEW
    d.act, d.slope = _ACT["lrelu"], slope
    d.residual = _NULL_VIEW
    d.out_scale = 1.0
    d.pixel_shuffle = 0
    d.out = out.v
    d.precision, d.weight16, d.weight16_unscale = _lib.PREC_F16X3, w16[0].data_ptr(), w16[1]
    args = (C.byref(d), C.c_void_p(dw_w.data_ptr()), C.c_void_p(dw_b.data_ptr()), stream_ptr())
    if OP_LOG is None:
        check(lib.lssvc_conv1x1_dw3x3_f16x3(*args))
        return out
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(lib.lssvc_conv1x1_dw3x3_f16x3(*args))
    e1.record()
    cin = sum(t.C for t in inputs)
    OP_LOG.append({"kind": "conv1x1+dw3x3", "name": conv_name, "macs": out.H * out.W * cout * (cin + 9), "hout": out.H, "wout": out.W,
                   "cin": cin, "cout": cout, "variant": 0, "ks": 1, "stride": 1, "vec": True, "f16x3": True,
                   "kernel": "dwpre_f16x3_kernel<%d, %d, 16>" % (cout // 16, (sum((t.C + 15) // 16 for t in inputs) + 1) // 2),
                   "bytes": 4 * out.H * out.W * (cin + cout), "events": (e0, e1)})
    return out


FUSE_FFN = _os.environ.get("LSSVC_FUSE_FFN", "1") == "1"
_FFN_LDS_LIMIT = 160 * 1024


def ffn_fusable(W, ffn_prefix, pre_name, c_out, pre_cin):
    """Can DepthConv.conv2 (+identity) + ConvFFN of this block run as one lssvc_ffn_f16x3 launch?"""
    if not (FUSE_FFN and CONV_PRECISION == "f16x3" and c_out in (32, 48, 64, 96, 128)) or RANGE_AUDIT is not None:
        return False                     # (an audited frame runs the block's convs one by one: o1 and the hidden tensor are seen)
    if W.force_f32 and (pre_name in W.force_f32 or ffn_prefix + ".conv.0" in W.force_f32 or ffn_prefix + ".conv.2" in W.force_f32):
        return False
    w1 = W.raw(ffn_prefix + ".conv.0.weight")
    if w1.shape[1] != c_out or w1.shape[0] % 32 or not W.has(ffn_prefix + ".conv.0.bias") or not W.has(ffn_prefix + ".conv.2.bias"):
        return False
    if pre_name is not None and (pre_cin % 8 or pre_cin > 128 or not W.has(pre_name + ".bias")):
        return False
    return lib.lssvc_ffn_f16x3_lds_bytes(c_out, w1.shape[0], pre_cin if pre_name else 0) <= _FFN_LDS_LIMIT


def ffn_block(W, ffn_prefix, *, x=None, pre_name=None, pre_in=None, ident=None, slope=0.1, out=None, skip=None):
    """out = o1 + lrelu(conv.2(lrelu(conv.0(o1)))) with o1 = x, or o1 = pre_name(pre_in) + ident, in one launch
    (DepthConvBlock's per-pixel tail, lssvc_modules.py:38-72)."""
    rec = W.ffn_f16x3(ffn_prefix, pre_name)
    ref = x if pre_name is None else ident
    if out is None:
        out = ref.like()
    d = _lib.FfnDesc()
    d.x = x.v if pre_name is None else _NULL_VIEW
    if pre_name is not None:
        d.pre_in, d.ident = pre_in.v, ident.v
        d.pre_w16, d.pre_unscale, d.pre_bias = rec["wp"].data_ptr(), rec["up"], rec["bp"].data_ptr()
    else:
        d.pre_in, d.ident = _NULL_VIEW, _NULL_VIEW
    d.w1_16, d.w1_unscale, d.b1, d.hidden = rec["w1"].data_ptr(), rec["u1"], rec["b1"].data_ptr(), rec["hidden"]
    d.w2_16, d.w2_unscale, d.b2 = rec["w2"].data_ptr(), rec["u2"], rec["b2"].data_ptr()
    d.slope = slope
    d.out = out.v
    d.skip = skip.v if skip is not None else _NULL_VIEW
    if OP_LOG is None:
        check(lib.lssvc_ffn_f16x3(C.byref(d), stream_ptr()))
        return out
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(lib.lssvc_ffn_f16x3(C.byref(d), stream_ptr()))
    e1.record()
    c, hid = rec["C"], rec["hidden"]
    pre_cin = rec.get("pre_cin", 0) if pre_name is not None else 0
    npx = out.H * out.W
    OP_LOG.append({"kind": "ffn_fused", "name": ffn_prefix, "macs": npx * (2 * c * hid + pre_cin * c),
                   "hout": out.H, "wout": out.W, "cin": c, "cout": c, "variant": 0, "ks": 1, "stride": 1, "vec": True,
                   "f16x3": True, "kernel": "%s<%d, %s>" % ("ffn_stream_f16x3_kernel" if lib.lssvc_ffn_f16x3_is_streamed(c, hid, pre_cin)
                                                             else "ffn_f16x3_kernel", c // 16, "true" if pre_name else "false"),
                   "bytes": 4 * npx * (2 * c + (pre_cin if pre_name else 0)), "events": (e0, e1)})
    return out


def resize(x, H, W_, scale=1.0, out=None):
    if out is None:
        out = T.empty(int(H), int(W_), x.C, x.device)
    check(lib.lssvc_resize_bilinear(x.ref, out.ref, scale, stream_ptr()))
    return out


def flow_warp(x, flow, out=None):
    if out is None:
        out = x.like()
    check(lib.lssvc_flow_warp(x.ref, flow.ref, out.ref, stream_ptr()))
    return out


def spynet_prep(im1, im2, flow_lo, out):
    """out (H x W x 8) = cat(im1, warp(im2, up), up) with up = 2 * bilinear_x2(flow_lo): one SpyNet level's input in one launch."""
    check(lib.lssvc_spynet_prep(im1.ref, im2.ref, flow_lo.ref, out.ref, stream_ptr()))
    return out


def avgpool_pyramid3(x):
    """[x, avg_pool(x), avg_pool^2(x), avg_pool^3(x)]: one launch when H, W are multiples of 8, else level by level."""
    if x.H % 8 or x.W % 8:
        out = [x]
        for _ in range(3):
            out.append(pool2x2(out[-1], is_max=False))
        return out
    lv = [T.empty(x.H >> k, x.W >> k, x.C, x.device) for k in (1, 2, 3)]
    check(lib.lssvc_avgpool_pyramid3(x.ref, lv[0].ref, lv[1].ref, lv[2].ref, stream_ptr()))
    return [x] + lv


def pool2x2(x, is_max, out=None):
    if out is None:
        out = T.empty(x.H // 2, x.W // 2, x.C, x.device)
    check(lib.lssvc_pool2x2(x.ref, out.ref, 1 if is_max else 0, stream_ptr()))
    return out


def softmax2_blend(a, b, logits, out=None):
    if out is None:
        out = a.like()
    check(lib.lssvc_softmax2_blend(a.ref, b.ref, logits.ref, out.ref, stream_ptr()))
    return out


def add(a, b, out=None):
    if out is None:
        out = a.like()
    check(lib.lssvc_add(a.ref, b.ref, out.ref, stream_ptr()))
    return out


def copy(a, out):
    check(lib.lssvc_copy(a.ref, out.ref, stream_ptr()))
    return out


def clamp_(t, lo=0.0, hi=1.0):
    """torch.Tensor.clamp_(lo, hi) on a densely stored T (test.py:249-250's clamp of the reconstructions), as a launch of the library
    (lssvc_clamp_inplace): a frame plan that holds it can be recorded for the C engine, an ATen kernel cannot."""
    assert t.ld == t.C, "clamp_: the view must be dense"
    check(lib.lssvc_clamp_inplace(C.c_void_p(t.buf.data_ptr() + 4 * t.off), t.H * t.W * t.C, lo, hi, stream_ptr()))
    return t


def pad_crop(x, pad):
    """F.pad(x, (left, right, top, bottom), value=0) of an NHWC view, negative entries cropping: the reference's
    get_depadded_feature (IntraSS.py:124-135, LSSVC_net.py:271-282). Pure data movement (lssvc_pad_crop); all zeros in
    `pad` returns x itself."""
    l, r, t, b = (int(v) for v in pad)
    if l == r == t == b == 0:
        return x
    H, W = x.H + t + b, x.W + l + r
    assert H > 0 and W > 0, "pad_size %s leaves nothing of a %dx%d map" % (pad, x.H, x.W)
    out = T.empty(H, W, x.C, x.device)
    check(lib.lssvc_pad_crop(x.ref, out.ref, l, t, stream_ptr()))      # a library launch: compiled frame plans record it
    return out


def cat(parts):
    """Materialise torch.cat(dim=1) (only where a fused multi-input conv cannot absorb it)."""
    total = sum(p.C for p in parts)
    out = T.empty(parts[0].H, parts[0].W, total, parts[0].device)
    a = 0
    for p in parts:
        copy(p, out.slice(a, a + p.C))
        a += p.C
    return out


def presplit(x, in_act=None, in_slope=0.01, out=None):
    """fp32 view -> pre-split view (fp16 hi | lo per 16-channel chunk, the input activation applied): lssvc_presplit."""
    if out is None:
        cp = (x.C + 15) // 16 * 16
        if ARENA is not None:
            out = T(ARENA.alloc_f32(x.H * x.W * cp), x.H, x.W, x.C, cp, split=True)
        else:
            out = T(torch.empty(x.H * x.W * cp, dtype=torch.float32, device=x.device), x.H, x.W, x.C, cp, split=True)
    check(lib.lssvc_presplit(x.ref, out.ref, _INACT[in_act], in_slope, stream_ptr()))
    return out


def lrelu(x, slope, out=None):
    if out is None:
        out = x.like()
    check(lib.lssvc_lrelu(x.ref, out.ref, slope, stream_ptr()))
    return out


def offset_diversity_tail(x, om, flow, fusion_w, fusion_b, out=None):
    if out is None:
        out = x.like()
    check(lib.lssvc_offset_diversity(x.ref, om.ref, flow.ref, C.c_void_p(fusion_w.data_ptr()),
                                     C.c_void_p(fusion_b.data_ptr()), out.ref, stream_ptr()))
    return out


# ---- entropy ----------------------------------------------------------------------------------------
class BitSlots:
    """A small device array of fp64 accumulators + the reduction workspace. One D2H copy per frame."""

    def __init__(self, device, n=16):
        self.vals = torch.zeros(n, dtype=torch.float64, device=device)
        self.device = device
        words = int(lib.lssvc_reduce_workspace_bytes()) // 8
        self._free = [torch.empty(words, dtype=torch.float64, device=device) for _ in range(16)]  # up front: none is ever
        self._ws = {}                                                                             # allocated inside a capture
        self.lane = 0             # 1 while the look-ahead base layer is issued (inter.py): its plan is captured on the SAME capture-time
        #                           side streams as the frame's other plan and replayed beside it, so the stream alone does not tell them apart

    def slot(self, i):
        return C.c_void_p(self.vals.data_ptr() + 8 * i)

    @property
    def wsp(self):
        """The reduction workspace of the CURRENT stream (reductions on different streams may run concurrently)."""
        sid = (self.lane, torch.cuda.current_stream().cuda_stream)
        ws = self._ws.get(sid)
        if ws is None:
            if not self._free:
                raise RuntimeError("BitSlots: more than 16 (lane, stream) pairs issued bit reductions")
            ws = self._ws[sid] = self._free.pop()
        return C.c_void_p(ws.data_ptr())

    def fetch(self):
        return self.vals.cpu().tolist()     # the per-frame device->host sync (reference: .item(), IntraSS.py:166)


def _opt(t):
    return t.ref if t is not None else C.byref(_NULL_VIEW)


def laplace_quant_bits(y, mean, sigma, slots, slot, y_q=None, y_hat=None):
    check(lib.lssvc_laplace_quant_bits(y.ref, mean.ref, sigma.ref, _opt(y_q), _opt(y_hat), slots.slot(slot), slots.wsp,
                                       stream_ptr()))


def laplace_bits(y_q, sigma, slots, slot):
    check(lib.lssvc_laplace_bits(y_q.ref, sigma.ref, slots.slot(slot), slots.wsp, stream_ptr()))


def four_part_step(y, mean, sigma, mask_of_chunk, y_q, y_hat, sigma_hat):
    arr = (C.c_int32 * 4)(*mask_of_chunk)
    check(lib.lssvc_four_part_step(y.ref, mean.ref, sigma.ref, arr, y_q.ref, y_hat.ref, sigma_hat.ref, stream_ptr()))


def factorized_quant_bits(z, params, slots, slot, z_hat=None):
    check(lib.lssvc_factorized_quant_bits(z.ref, C.c_void_p(params.data_ptr()), _opt(z_hat), slots.slot(slot), slots.wsp,
                                          stream_ptr()))


def gaussian_conditional(y, scale, mean, slots, slot, y_hat=None, y_q=None):
    check(lib.lssvc_gaussian_conditional(y.ref, scale.ref, mean.ref, _opt(y_hat), _opt(y_q), slots.slot(slot), slots.wsp,
                                         stream_ptr()))


def entropy_bottleneck(z, params, slots, slot, z_hat=None, z_q=None):
    check(lib.lssvc_entropy_bottleneck(z.ref, C.c_void_p(params.data_ptr()), _opt(z_hat), _opt(z_q), slots.slot(slot),
                                       slots.wsp, stream_ptr()))


# ---- write_stream = 1: symbol / index planes between the device and the host coder ---------------------------------------
# The reference moves every latent through Python lists (`.tolist()` of ~2.3 M symbols per P-frame,
# video_entropy_models.py:234-236,317-319). Here a layer's planes are int16, written by the export kernels into ONE device
# staging buffer and brought down by ONE asynchronous copy into pinned host memory when the layer's string is flushed; the
# C coder reads them in place. STREAM_PROF (a dict, or None) collects where the time of a write_stream frame goes.
STREAM_PROF = None


def _prof(key, t0):
    if STREAM_PROF is not None:
        import time
        STREAM_PROF[key] = STREAM_PROF.get(key, 0.0) + time.perf_counter() - t0


def _prof_start(sync_device=None):
    """-> perf_counter() or None; with profiling on, the device is drained first so that the interval that follows holds
    only the work it brackets (the GPU forward that precedes a copy is then booked under 'gpu_wait_s')."""
    if STREAM_PROF is None:
        return None
    import time
    if sync_device is not None:
        t0 = time.perf_counter()
        torch.cuda.synchronize(sync_device)
        STREAM_PROF["gpu_wait_s"] = STREAM_PROF.get("gpu_wait_s", 0.0) + time.perf_counter() - t0
    return time.perf_counter()


class PlaneRef:
    """n int16 entries at element offset `off` of a SymbolStage."""
    __slots__ = ("off", "n")

    def __init__(self, off, n):
        self.off, self.n = off, n


class SymbolStage:
    """Device + pinned-host int16 staging of one layer's symbol and index planes."""

    def __init__(self, device):
        self.device = device
        self.capacity = self.used = 0
        self.dev = self.host = self.host_np = None
        self._down = (0, 0)
        self._pre = None
        self.uploaded = None                                                   # event behind the last asynchronous H2D out of the pinned buffer
        self.flag = torch.zeros(1, dtype=torch.int32, device=device)           # set by a symbol that does not fit 16 bits
        self.flag_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self.done = torch.cuda.Event()

    def begin(self, capacity):
        """Start a layer: everything staged before is dead (its copies were waited for when its string was flushed; an
        upload still in flight out of the pinned buffer is waited for here, before the host writes into it again -- what
        plan_runtime.cpp does with hipStreamSynchronize before every decode step)."""
        if self.uploaded is not None:
            self.uploaded.synchronize()
            self.uploaded = None
        if capacity > self.capacity:
            self.capacity = int(capacity)
            self.dev = torch.empty(self.capacity, dtype=torch.int16, device=self.device)
            self.host = torch.empty(self.capacity, dtype=torch.int16).pin_memory()
            self.host_np = self.host.numpy()
        self.used = 0
        self._down = (0, 0)
        self._pre = None
        self.flag.zero_()
        return self

    def alloc(self, n):
        off = self.used
        self.used += (n + 7) // 8 * 8                                           # 16-byte aligned planes
        if self.used > self.capacity:
            raise RuntimeError("SymbolStage: %d entries staged, capacity %d" % (self.used, self.capacity))
        return PlaneRef(off, n)

    def dev_ptr(self, ref):
        return C.c_void_p(self.dev.data_ptr() + 2 * ref.off)

    def host_ptr(self, ref):
        return C.c_void_p(self.host.data_ptr() + 2 * ref.off)

    def numpy(self, ref):
        return self.host_np[ref.off:ref.off + ref.n]

    def prefetch(self):
        """Encoder side, called right after a layer's LAST plane has been exported: put the copy of everything staged so far
        on the stream NOW, with an event behind it. The kernels the codec issues after this point (the layer's synthesis
        transform and reconstruction network: a third of a P-frame) then run while the host codes the planes, instead of
        the host waiting for them before it starts."""
        if self.used == 0 or STREAM_PROF is not None:        # (a profiled pass books GPU, copy and coder time separately: no overlap)
            return
        self.host[0:self.used].copy_(self.dev[0:self.used], non_blocking=True)
        self.flag_host.copy_(self.flag, non_blocking=True)
        self.done.record()
        self._pre = (0, self.used)

    def download(self, lo, hi):
        """dev[lo:hi] -> pinned host, asynchronously on the current stream; returns after the copy has landed."""
        if lo >= self._down[0] and hi <= self._down[1]:
            return                                                              # a sink sharing the stage already brought it down
        t0 = _prof_start(self.device)
        if self._pre is not None and lo >= self._pre[0] and hi <= self._pre[1]:
            self._down = self._pre                                              # prefetch(): only its event is waited for
        else:
            self._down = (lo, hi)
            self.host[lo:hi].copy_(self.dev[lo:hi], non_blocking=True)
            self.flag_host.copy_(self.flag, non_blocking=True)
            self.done.record()
        self.done.synchronize()
        if t0 is not None:
            _prof("d2h_s", t0)
            STREAM_PROF["d2h_bytes"] = STREAM_PROF.get("d2h_bytes", 0) + 2 * (hi - lo)
        if int(self.flag_host[0]) != 0:
            raise RuntimeError("a quantised latent does not fit the 16-bit symbol planes")

    def upload(self, ref):
        t0 = _prof_start()
        self.dev[ref.off:ref.off + ref.n].copy_(self.host[ref.off:ref.off + ref.n], non_blocking=True)
        if self.uploaded is None:
            self.uploaded = torch.cuda.Event()
        self.uploaded.record()
        if t0 is not None:
            _prof("h2d_s", t0)
            STREAM_PROF["h2d_bytes"] = STREAM_PROF.get("h2d_bytes", 0) + 2 * ref.n


def export_symbols(q, sigma, index_params=None, chunk_of_mask=None, stage=None):
    """Symbol / index planes of a latent in NCHW order (either may be None). Without `stage`: host int32 numpy planes
    (synchronous; tests and the compat modules). With a SymbolStage: the planes are written as int16 into the stage's
    device buffer and (PlaneRef | None, PlaneRef) is returned -- nothing is copied or waited for here.
    index_params = (log_min, log_step, add, levels) when sigma is given; chunk_of_mask folds C -> C/4."""
    ref = q if q is not None else sigma
    c_out = ref.C // 4 if chunk_of_mask is not None else ref.C
    n = ref.H * ref.W * c_out
    lo, step, add, levels = index_params if index_params is not None else (0.0, 1.0, 0.0, 1)
    cm = (C.c_int32 * 4)(*chunk_of_mask) if chunk_of_mask is not None else None
    if stage is not None:
        r_sym = stage.alloc(n) if q is not None else None
        r_idx = stage.alloc(n)
        check(lib.lssvc_export_symbols_i16(_opt(q), _opt(sigma), cm, lo, step, add, levels,
                                           stage.dev_ptr(r_sym) if r_sym is not None else None, stage.dev_ptr(r_idx),
                                           C.c_void_p(stage.flag.data_ptr()), stream_ptr()))
        return r_sym, r_idx
    sym = torch.empty(n, dtype=torch.int32, device=ref.device) if q is not None else None
    idx = torch.empty(n, dtype=torch.int32, device=ref.device)
    check(lib.lssvc_export_symbols(_opt(q), _opt(sigma), cm, lo, step, add, levels,
                                   C.c_void_p(sym.data_ptr()) if sym is not None else None, C.c_void_p(idx.data_ptr()),
                                   stream_ptr()))
    t0 = _prof_start(ref.device)
    out = (sym.cpu().numpy() if sym is not None else None), idx.cpu().numpy()
    if t0 is not None:
        _prof("d2h_s", t0)
        STREAM_PROF["d2h_bytes"] = STREAM_PROF.get("d2h_bytes", 0) + 4 * n * (2 if sym is not None else 1)
    return out


def export_indexes(sigma, index_params, chunk_of_mask=None, stage=None):
    """Decoder side: the table-index plane of `sigma` ON THE HOST (the coder needs it to decode): int16 view of the
    stage's pinned buffer, or an int32 array without a stage. Waits for the device."""
    if stage is None:
        return export_symbols(None, sigma, index_params, chunk_of_mask)[1]
    _, r = export_symbols(None, sigma, index_params, chunk_of_mask, stage=stage)
    stage.download(r.off, r.off + r.n)
    return stage.numpy(r)


def import_symbols(symbols, out, mean=None, channel_add=None, chunk_of_mask=None, stage=None):
    """Host NCHW plane -> device, out = sym + mean + channel_add[c] (unfolding C/4 -> C with chunk_of_mask). An int16 plane
    that lives in `stage`'s pinned buffer goes up by an asynchronous copy; anything else through a pageable copy."""
    cm = (C.c_int32 * 4)(*chunk_of_mask) if chunk_of_mask is not None else None
    add = C.c_void_p(channel_add.data_ptr()) if channel_add is not None else None
    if stage is not None and symbols.dtype.name == "int16" and symbols.ctypes.data >= stage.host.data_ptr() \
            and symbols.ctypes.data + 2 * symbols.size <= stage.host.data_ptr() + 2 * stage.capacity:
        ref = PlaneRef((symbols.ctypes.data - stage.host.data_ptr()) // 2, symbols.size)
        stage.upload(ref)
        check(lib.lssvc_import_symbols_i16(stage.dev_ptr(ref), _opt(mean), add, cm, out.ref, stream_ptr()))
        return out
    assert symbols.dtype.name in ("int16", "int32"), symbols.dtype
    t0 = _prof_start()
    dev = torch.from_numpy(symbols).to(out.device)
    if t0 is not None:
        _prof("h2d_s", t0)
        STREAM_PROF["h2d_bytes"] = STREAM_PROF.get("h2d_bytes", 0) + symbols.nbytes
    fn = lib.lssvc_import_symbols_i16 if symbols.dtype.name == "int16" else lib.lssvc_import_symbols
    check(fn(C.c_void_p(dev.data_ptr()), _opt(mean), add, cm, out.ref, stream_ptr()))
    return out
