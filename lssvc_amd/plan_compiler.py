"""Compile a frame plan: the Python model objects are the FRONT END, a plan file is what the C++ runtime executes.

SURVEY 8b's last row asks for engine-level entry points a caller without Python can use (`lssvc_engine_create / load_* /
iframe / pframe`, owning the weights and the captured graph). The estimate-mode forward of a frame type at a given size is a
FIXED sequence of C-ABI launches, so instead of restating the ~1000 lines of graph logic in C++ this
module records that sequence once while the front end runs it for real:

  * every activation is taken from ONE arena (hip_ops.ARENA: first-fit with lifetime reuse, same stream rules as PyTorch's
    caching allocator), so each device pointer in a launch is (region, offset): the arena, the prepared weight tensors
    (lssvc_amd.weights layouts, stored in the file), zero-initialised scratch (bit accumulators, reduction workspaces), or one
    of the caller's input / output buffers;
  * a launch is stored as (function name, stream index, arguments), structs as byte images with their pointer fields listed
    for rebasing; fork / join edges between the side streams of a frame (hip_ops.Fork) are stored as waits.

csrc/plan_runtime.cpp loads such a file, allocates the regions, rebases the pointers, replays the launches (once eagerly,
then as a hipGraph captured inside the library with hipStreamBeginCapture) and exposes lssvc_engine_* -- no Python, no torch.
What stays in the front end: checkpoint validation, the weight re-layout (weights.py) and the graph logic, i.e. everything
that runs once per (checkpoint, frame size); a plan is specific to that pair, like an engine file of any inference runtime.
"""
import ctypes as C
import struct
import weakref

import torch

from . import _lib
from . import hip_ops as ops

MAGIC = b"LSSVCPL2"       # 2: weight regions carry the recipe that rebuilds them from a raw checkpoint, structs carry scalar fixes
# host steps of the write_stream = 1 plans, stored in the launch list where the front end performed them (names start with "__")
HOST_D2H, HOST_H2D, HOST_ENCODE, HOST_FLUSH, HOST_SET_STREAM, HOST_DECODE, HOST_DECODE_CH, HOST_D2H_ASYNC, HOST_D2H_WAIT = (
    "__d2h__", "__h2d__", "__encode__", "__flush__", "__set_stream__", "__decode__", "__decode_ch__", "__d2h_async__", "__d2h_wait__")
REGION_ARENA, REGION_WEIGHTS, REGION_SCRATCH, REGION_INPUT, REGION_OUTPUT = range(5)
TAG_NULL, TAG_PTR, TAG_STRUCT, TAG_F32, TAG_I32, TAG_I64, TAG_I32ARRAY, TAG_STREAM = range(8)
FN_WAIT = "__wait__"


class Arena:
    """One device buffer, first-fit allocation with coalescing; a block returns to the free list when the tensor view that
    was handed out dies (the same moment PyTorch's allocator would recycle it, so the stream-ordering argument of
    hip_ops.Fork carries over unchanged)."""

    def __init__(self, nbytes, device):
        self.buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        self.free = [(0, int(nbytes))]
        self.peak = 0

    def alloc_f32(self, n):
        need = (4 * n + 255) // 256 * 256
        for i, (off, size) in enumerate(self.free):
            if size >= need:
                if size == need:
                    self.free.pop(i)
                else:
                    self.free[i] = (off + need, size - need)
                self.peak = max(self.peak, off + need)
                t = self.buf[off:off + 4 * n].view(torch.float32)
                weakref.finalize(t, self._release, off, need)
                return t
        raise RuntimeError("plan arena of %d bytes is too small" % self.buf.numel())

    def _release(self, off, size):
        self.free.append((off, size))
        self.free.sort()
        merged = []
        for o, s in self.free:
            if merged and merged[-1][0] + merged[-1][1] == o:
                merged[-1] = (merged[-1][0], merged[-1][1] + s)
            else:
                merged.append((o, s))
        self.free = merged


def _pointer_fields(ctype, base=0):
    """[(byte offset, ...)] of every c_void_p inside a ctypes Structure type, nested structs and arrays included."""
    out = []
    for name, ft in ctype._fields_:
        off = base + getattr(ctype, name).offset
        if ft is C.c_void_p:
            out.append(off)
        elif isinstance(ft, type) and issubclass(ft, C.Structure):
            out += _pointer_fields(ft, off)
        elif isinstance(ft, type) and issubclass(ft, C.Array) and issubclass(ft._type_, C.Structure):
            for k in range(ft._length_):
                out += _pointer_fields(ft._type_, off + k * C.sizeof(ft._type_))
    return out


_PTR_FIELDS = {}
# (float field, pointer field) pairs of the launch descriptors whose float is a by-product of preparing the pointer's weights
_SCALAR_FIELDS = {"ConvDesc": (("weight16_unscale", "weight16"),),
                  "FfnDesc": (("pre_unscale", "pre_w16"), ("w1_unscale", "w1_16"), ("w2_unscale", "w2_16"))}
# (recipe kind, blob index) -> index into the recipe's scalars (include/lssvc_hip.h: lssvc_prepare_weights)
_SCALAR_OF_BLOB = {(_lib.PREP_CONV_F16X3, 0): 0, (_lib.PREP_GDN, 2): 0, (_lib.PREP_FFN_F16X3, 0): 0, (_lib.PREP_FFN_F16X3, 1): 1, (_lib.PREP_FFN_F16X3, 4): 2}


class Recorder:
    def __init__(self, device, arena_bytes, weight_stores, scratch, inputs, outputs):
        """scratch / inputs / outputs: {name: tensor}; weight_stores: the models' WeightStore objects."""
        self.device = device
        self.arena = Arena(arena_bytes, device)
        self.regions = [{"kind": REGION_ARENA, "name": "arena", "ptr": self.arena.buf.data_ptr(), "nbytes": self.arena.buf.numel(), "data": None}]
        for kind, group in ((REGION_SCRATCH, scratch), (REGION_INPUT, inputs), (REGION_OUTPUT, outputs)):
            for name, t in group.items():
                self.regions.append({"kind": kind, "name": name, "ptr": t.data_ptr(), "nbytes": t.numel() * t.element_size(), "data": None,
                                     "shape": tuple(t.shape)})
        self.weight_stores = weight_stores
        self._weight_regions = {}
        self.launches = []
        self.streams = {}
        self._orig = {}
        self.tables = []          # entropy_coder.Tables objects of the host steps, in first-use order
        self._table_ids = {}

    def host(self, name, *ints):
        """A host step between launches (write_stream = 1 plans): performed by the runtime on the main stream, in order."""
        self.launches.append((name, 0, [(TAG_I64, int(v)) for v in ints]))

    def table_id(self, tables):
        if id(tables) not in self._table_ids:
            self._table_ids[id(tables)] = len(self.tables)
            self.tables.append(tables)
        return self._table_ids[id(tables)]

    # ---- pointers ------------------------------------------------------------------------------------------------
    def _resolve(self, p):
        for i, r in enumerate(self.regions):
            if r["ptr"] <= p < r["ptr"] + max(r["nbytes"], 1):
                return i, p - r["ptr"]
        for ws in self.weight_stores:                      # a prepared weight tensor (registered by WeightStore.prepare / _dev)
            for ptr, (nbytes, host) in ws.regions.items():
                if ptr <= p < ptr + max(nbytes, 1):
                    if ptr not in self._weight_regions:
                        self.regions.append({"kind": REGION_WEIGHTS, "name": "w%d" % len(self._weight_regions), "ptr": ptr, "nbytes": nbytes,
                                             "data": host, "recipe": ws.recipes.get(ptr)})
                        self._weight_regions[ptr] = len(self.regions) - 1
                    return self._weight_regions[ptr], p - ptr
        raise RuntimeError("launch uses device memory at 0x%x that is neither arena, weights, scratch nor a declared input/output" % p)

    def _stream(self, handle):
        h = int(handle or 0)
        if h not in self.streams:
            self.streams[h] = len(self.streams)
        return self.streams[h]

    def wait(self, waiter, signaler):
        """stream `waiter` waits for everything issued so far on `signaler` (a fork or a join of hip_ops.Fork)."""
        self.launches.append((FN_WAIT, self._stream(waiter), [(TAG_I32, self._stream(signaler))]))

    # ---- launches ------------------------------------------------------------------------------------------------
    def _encode(self, name, args):
        restype, argtypes = _lib.SIGNATURES[name]
        enc, stream = [], 0
        last_ptr = max(i for i, t in enumerate(argtypes) if t is C.c_void_p)
        for i, (a, t) in enumerate(zip(args, argtypes)):
            if t is C.c_void_p and isinstance(a, C.Array):     # a small host int32 array handed to a `const int32_t *` (chunk_of_mask)
                enc.append((TAG_I32ARRAY, [int(x) for x in a]))
            elif t is C.c_void_p:
                v = a.value if isinstance(a, C.c_void_p) else a
                if i == last_ptr:                              # every launching entry point ends with `void *stream`
                    stream = self._stream(v)
                    enc.append((TAG_STREAM, 0))
                elif v is None or v == 0:
                    enc.append((TAG_NULL, 0))
                else:
                    enc.append((TAG_PTR,) + self._resolve(int(v)))
            elif t is C.c_float:
                enc.append((TAG_F32, float(a)))
            elif t is C.c_int32:
                enc.append((TAG_I32, int(a)))
            elif t is C.c_int64:
                enc.append((TAG_I64, int(a)))
            elif t is C.POINTER(C.c_int32):
                enc.append((TAG_I32ARRAY, [int(x) for x in a]))
            else:                                              # POINTER(struct): byref(obj) or a pointer instance
                obj = a._obj if hasattr(a, "_obj") else a.contents
                st = type(obj)
                if st not in _PTR_FIELDS:
                    _PTR_FIELDS[st] = _pointer_fields(st)
                blob = bytearray(C.string_at(C.addressof(obj), C.sizeof(obj)))
                fixes = []
                for off in _PTR_FIELDS[st]:
                    (p,) = struct.unpack_from("<Q", blob, off)
                    if p:
                        fixes.append((off,) + self._resolve(p))
                        struct.pack_into("<Q", blob, off, 0)
                # floats that are a function of the CHECKPOINT (the 2^-e of a layer's fp16 weight planes): the runtime patches
                # them from the recipe of the weight region the paired pointer field resolves to
                sfixes = []
                for f_scalar, f_ptr in _SCALAR_FIELDS.get(st.__name__, ()):
                    p = getattr(obj, f_ptr)
                    if p:
                        reg, _ = self._resolve(int(p))
                        recipe = self.regions[reg].get("recipe")
                        if recipe is not None:
                            idx = _SCALAR_OF_BLOB[(recipe[0][0], recipe[1])]
                            sfixes.append((getattr(st, f_scalar).offset, reg, idx))
                enc.append((TAG_STRUCT, bytes(blob), fixes, sfixes))
        return stream, enc

    def __enter__(self):
        lib = _lib.lib
        for name in _lib.SIGNATURES:
            restype, argtypes = _lib.SIGNATURES[name]
            if restype is not C.c_int or not argtypes or argtypes[-1] is not C.c_void_p or not name.startswith("lssvc_") \
                    or name.startswith("lssvc_rans") or name in ("lssvc_pmf_to_quantized_cdf", "lssvc_conv2d_variant"):
                continue
            fn = getattr(lib, name)
            self._orig[name] = fn

            def wrapper(*args, _fn=fn, _name=name):
                stream, enc = self._encode(_name, args)
                self.launches.append((_name, stream, enc))
                return _fn(*args)
            setattr(lib, name, wrapper)
        ops.ARENA = self.arena
        ops.PLAN_RECORDER = self
        return self

    def __exit__(self, *exc):
        for name, fn in self._orig.items():
            setattr(_lib.lib, name, fn)
        ops.ARENA = None
        ops.PLAN_RECORDER = None
        return False

    # ---- file ------------------------------------------------------------------------------------------------------
    def save(self, path, kind, scale, shape_hr, meta=()):
        """meta: extra (name, int) pairs, e.g. channel counts of optional inputs."""
        def s48(x):
            b = x.encode()[:47]
            return b + b"\0" * (48 - len(b))
        out = bytearray()
        out += MAGIC
        out += struct.pack("<5I", len(self.regions), len(self.launches), max(len(self.streams), 1), len(meta), len(self.tables))
        out += struct.pack("<d2i", float(scale), int(shape_hr[0]), int(shape_hr[1]))
        out += s48(kind)
        for name, v in meta:
            out += s48(name) + struct.pack("<q", int(v))
        def s120(x):
            b = x.encode()
            assert len(b) < 120, x
            return b + b"\0" * (120 - len(b))
        for r in self.regions:
            nbytes = self.arena.peak if r["kind"] == REGION_ARENA else r["nbytes"]
            shape = list(r.get("shape", ())) + [0] * 4
            out += struct.pack("<IQ4q", r["kind"], nbytes, *shape[:4]) + s48(r["name"])
            if r["kind"] == REGION_WEIGHTS:
                rec = r.get("recipe")
                if rec is None:
                    out += struct.pack("<I", 0)                 # no recipe: the bytes follow the launch list (tables' medians)
                else:
                    (kind, name, name2, splits, flag), blob = rec
                    sp = list(splits) + [0] * 3
                    out += struct.pack("<I6i", 1, kind, blob, len(splits), sp[0], sp[1], sp[2]) + struct.pack("<i", flag) + s120(name) + s120(name2)
        for name, stream, args in self.launches:
            out += s48(name) + struct.pack("<2I", stream, len(args))
            for a in args:
                tag = a[0]
                out += struct.pack("<I", tag)
                if tag == TAG_PTR:
                    out += struct.pack("<IQ", a[1], a[2])
                elif tag == TAG_STRUCT:
                    out += struct.pack("<2I", len(a[1]), len(a[2])) + a[1] + b"\0" * (-len(a[1]) % 8)
                    for off, reg, roff in a[2]:
                        out += struct.pack("<2IQ", off, reg, roff)
                    out += struct.pack("<I", len(a[3]))
                    for off, reg, idx in a[3]:
                        out += struct.pack("<3I", off, reg, idx)
                elif tag == TAG_F32:
                    out += struct.pack("<f", a[1])
                elif tag == TAG_I32:
                    out += struct.pack("<i", a[1])
                elif tag == TAG_I64:
                    out += struct.pack("<q", a[1])
                elif tag == TAG_I32ARRAY:
                    out += struct.pack("<I", len(a[1])) + struct.pack("<%di" % len(a[1]), *a[1])
        for t in self.tables:                                   # CDF tables of the host coder: rows, stride, cdfs, sizes, offsets (int32)
            out += struct.pack("<2I", t.cdfs.shape[0], t.cdfs.shape[1]) + t.cdfs.tobytes() + t.sizes.tobytes() + t.offsets.tobytes()
        with open(path, "wb") as f:
            f.write(out)
            for r in self.regions:                              # payloads of the weight regions without a recipe, in region order, 256-byte aligned
                if r["kind"] == REGION_WEIGHTS and r.get("recipe") is None:
                    pad = -f.tell() % 256
                    f.write(b"\0" * pad)
                    f.write(r["data"].contiguous().cpu().numpy().tobytes())
        return {"launches": len(self.launches), "regions": len(self.regions), "arena_bytes": self.arena.peak, "streams": len(self.streams),
                "weight_recipes": sum(1 for r in self.regions if r["kind"] == REGION_WEIGHTS and r.get("recipe") is not None),
                "embedded_weight_bytes": sum(r["nbytes"] for r in self.regions if r["kind"] == REGION_WEIGHTS and r.get("recipe") is None),
                "host_steps": sum(1 for n, _, _ in self.launches if n.startswith("__") and n != FN_WAIT), "tables": len(self.tables)}


class StreamHooks:
    """While a write_stream = 1 pass is recorded: every step the front end performs on the HOST between launches -- staging
    copies of the int16 planes (hip_ops.SymbolStage), the rANS coder calls (entropy_coder) -- is written into the launch
    list at the point where it happened, with stage offsets instead of addresses, table ids instead of objects and the
    coders numbered in creation order (= the order of the strings in the layer files)."""

    def __init__(self, rec, stage):
        self.rec, self.stage = rec, stage
        self.n_enc = self.n_dec = self.n_out = 0
        self._saved = []

    def _off(self, a):
        """element offset of a numpy int16 view inside the stage's pinned buffer, or None"""
        base = self.stage.host.data_ptr()
        p = a.ctypes.data
        if a.dtype.name == "int16" and base <= p and p + 2 * a.size <= base + 2 * self.stage.capacity:
            return (p - base) // 2
        return None

    def _patch(self, cls, name, fn):
        self._saved.append((cls, name, cls.__dict__[name]))
        setattr(cls, name, fn)

    def __enter__(self):
        from . import entropy_coder as ec
        hooks, rec = self, self.rec
        st_cls, enc_cls, dec_cls = ops.SymbolStage, ec.RansEncoder, ec.RansDecoder
        o_begin, o_down, o_up, o_pre = st_cls.begin, st_cls.download, st_cls.upload, st_cls.prefetch
        o_einit, o_eenc, o_eflush = enc_cls.__init__, enc_cls.encode_with_indexes, enc_cls.flush
        o_dinit, o_dset, o_ddec = dec_cls.__init__, dec_cls.set_stream, dec_cls.decode_stream

        def begin(st, capacity):
            assert st is hooks.stage and capacity <= st.capacity, "the stage must not grow while a plan is recorded"
            r = o_begin(st, capacity)
            _lib.check(_lib.lib.lssvc_fill_zero(C.c_void_p(st.flag.data_ptr()), 4, ops.stream_ptr()))     # the recorded form of flag.zero_()
            return r

        def prefetch(st):
            if st.used:
                rec.host(HOST_D2H_ASYNC, 0, st.used)
            return o_pre(st)

        def download(st, lo, hi):
            if not (lo >= st._down[0] and hi <= st._down[1]):
                if st._pre is not None and lo >= st._pre[0] and hi <= st._pre[1]:
                    rec.host(HOST_D2H_WAIT)                               # the copy is on the stream already (prefetch)
                else:
                    rec.host(HOST_D2H, lo, hi)
            return o_down(st, lo, hi)

        def upload(st, ref):
            rec.host(HOST_H2D, ref.off, ref.n)
            return o_up(st, ref)

        def einit(enc):
            o_einit(enc)
            enc._plan_id = hooks.n_enc
            hooks.n_enc += 1

        def eenc(enc, symbols, indexes, tables):
            so, io = hooks._off(symbols), hooks._off(indexes)
            assert so is not None and io is not None, "stream plans code from the staged int16 planes only"
            rec.host(HOST_ENCODE, enc._plan_id, so, io, symbols.size, rec.table_id(tables))
            return o_eenc(enc, symbols, indexes, tables)

        def eflush(enc):
            rec.host(HOST_FLUSH, enc._plan_id, hooks.n_out)
            hooks.n_out += 1
            return o_eflush(enc)

        def dinit(dec):
            o_dinit(dec)
            dec._plan_id = hooks.n_dec
            hooks.n_dec += 1

        def dset(dec, data):
            rec.host(HOST_SET_STREAM, dec._plan_id, dec._plan_id)        # string k of the layer files feeds decoder k
            return o_dset(dec, data)

        def ddec(dec, indexes, tables, out=None):
            oo = hooks._off(out) if out is not None else None
            assert oo is not None, "stream plans decode into the staged int16 planes only"
            io = hooks._off(indexes)
            if io is not None:
                rec.host(HOST_DECODE, dec._plan_id, io, indexes.size, rec.table_id(tables), oo)
            else:                                                       # the channel-number plane of a factorised table (intra._channel_indexes)
                c = int(indexes[-1]) + 1
                assert indexes.size % c == 0 and int(indexes[0]) == 0
                rec.host(HOST_DECODE_CH, dec._plan_id, c, indexes.size // c, rec.table_id(tables), oo)
            return o_ddec(dec, indexes, tables, out=out)

        for cls, name, fn in ((st_cls, "begin", begin), (st_cls, "download", download), (st_cls, "upload", upload), (st_cls, "prefetch", prefetch),
                              (enc_cls, "__init__", einit), (enc_cls, "encode_with_indexes", eenc), (enc_cls, "flush", eflush),
                              (dec_cls, "__init__", dinit), (dec_cls, "set_stream", dset), (dec_cls, "decode_stream", ddec)):
            self._patch(cls, name, fn)
        return self

    def __exit__(self, *exc):
        for cls, name, fn in reversed(self._saved):
            setattr(cls, name, fn)
        return False


def entropy_params_crc(sd):
    """CRC-32 over the tensors update() builds the CDF tables from (every fp32 tensor of a BitEstimator / EntropyBottleneck: names
    containing 'bit_estimator' or 'entropy_bottleneck'), in name order: name bytes, then the raw fp32 bytes. plan_runtime.cpp
    computes the same over the checkpoint it was given."""
    import zlib
    crc = 0
    for k in sorted(sd):
        v = sd[k]
        if ("bit_estimator" in k or "entropy_bottleneck" in k) and v.is_floating_point() and v.dim() <= 4:
            crc = zlib.crc32(k.encode(), crc)
            crc = zlib.crc32(v.detach().to("cpu", torch.float32).contiguous().numpy().tobytes(), crc)
    return crc


def _record(model, body_inputs, run, outputs, path, kind, arena_gib, meta=(), stream_mode=False):
    """Run `run()` (which must issue the frame through `model` from the NCHW tensors in body_inputs and write the NCHW
    results into the tensors in `outputs`) once eagerly to warm everything up, then once under the recorder."""
    device = model.device
    assert not model.graph_mode, "compile plans from a model in eager mode"
    # warm-up THROUGH the fp16 range audit (a frame type this model has not coded yet is audited here, exactly as its first
    # encode_decode would be): weights laid out, LDS granted, and W.force_f32 final before anything is recorded
    model._with_range_audit(("plan", kind, model.shape_hr, float(model.scale_factor)), run)
    torch.cuda.synchronize(device)
    import zlib
    f32 = sorted(model.W.force_f32)
    meta = tuple(meta) + tuple(zip(("pad_left", "pad_right", "pad_top", "pad_bottom"), model.pad_size)) + (
        ("f32_layers_n", len(f32)), ("f32_layers_crc", zlib.crc32("\n".join(f32).encode())))
    if stream_mode:
        # a stream plan carries the CDF tables update() built from THIS checkpoint: the engine refuses to bind it to another one
        # (plan_runtime.cpp: entropy_params_crc), whose weights would code against the wrong tables
        meta += (("entropy_params_crc", entropy_params_crc(model.W.sd)),)
    scratch = {"bits": model.slots.vals}
    for i, ws in enumerate(list(model.slots._ws.values()) + list(model.slots._free)):
        scratch["reduce_ws%d" % i] = ws
    if stream_mode:                                         # the int16 staging buffer of the symbol / index planes and its overflow flag
        scratch["stage_dev"] = model.stage.dev
        scratch["stage_flag"] = model.stage.flag
    rec = Recorder(device, int(arena_gib * 2 ** 30), [model.W], scratch, body_inputs, outputs)
    with rec:
        if stream_mode:
            with StreamHooks(rec, model.stage):
                run()
        else:
            run()
    torch.cuda.synchronize(device)
    info = rec.save(path, kind, model.scale_factor, model.shape_hr, meta)
    return info


def compile_iframe(inet, x_bl, x_el, path, arena_gib=None):
    """Plan of IntraSS.forward (estimate mode) for frames shaped like x_bl / x_el (NCHW fp32, contiguous, on the device).
    Inputs of the plan: x_bl, x_el. Outputs: x_hat_bl, x_hat_el, feature_el (NCHW) and the four bit accumulators."""
    from .hip_ops import T
    H, W = inet.shape_hr
    arena_gib = arena_gib if arena_gib is not None else max(0.25, 8.0 * H * W / (1152.0 * 1920.0))
    dev = inet.device
    outs = {"x_hat_bl": torch.empty(1, 3, x_bl.shape[2], x_bl.shape[3], device=dev), "x_hat_el": torch.empty(1, 3, H, W, device=dev),
            "feature_el": torch.empty(1, 64, H, W, device=dev)}
    ins = {"x_bl": x_bl.contiguous(), "x_el": x_el.contiguous()}

    def run():
        r = inet._frame_body({k: T.from_nchw(v) for k, v in ins.items()})
        for k in outs:
            _lib.check(_lib.lib.lssvc_nhwc_to_nchw(r[k].ref, C.c_void_p(outs[k].data_ptr()), ops.stream_ptr()))

    info = _record(inet, ins, run, outs, path, "iframe", arena_gib)
    return info, outs


def compile_pframe(pnet, x_bl, x_el, dpb, path, arena_gib=None):
    """Plan of LSSVC.forward_one_frame (estimate mode). The DPB decides the frame type: ref_feature_bl None and a 64-channel
    ref_feature_el give the first-P plan (after an I-frame), otherwise the steady-P plan. All inputs NCHW contiguous."""
    from .hip_ops import T
    H, W = pnet.shape_hr
    arena_gib = arena_gib if arena_gib is not None else max(0.5, 26.0 * H * W / (1152.0 * 1920.0))
    dev = pnet.device
    ins = {"x_bl": x_bl.contiguous(), "x_el": x_el.contiguous(), "ref_frame_bl": dpb["ref_frame_bl"].contiguous(),
           "ref_frame_el": dpb["ref_frame_el"].contiguous(), "ref_feature_el": dpb["ref_feature_el"].contiguous()}
    if dpb["ref_feature_bl"] is not None:
        ins["ref_feature_bl"] = dpb["ref_feature_bl"].contiguous()
    h, w = x_bl.shape[2], x_bl.shape[3]
    outs = {"recon_bl": torch.empty(1, 3, h, w, device=dev), "feature_bl": torch.empty(1, 64, h, w, device=dev),
            "recon_el": torch.empty(1, 3, H, W, device=dev), "feature_el": torch.empty(1, 48, H, W, device=dev),
            "mv_hat": torch.empty(1, 2, H, W, device=dev), "warp_frame": torch.empty(1, 3, H, W, device=dev)}

    def run():
        t = {k: T.from_nchw(v) for k, v in ins.items()}
        t.setdefault("ref_feature_bl", None)
        r = pnet._frame_body(t)
        for k in outs:
            _lib.check(_lib.lib.lssvc_nhwc_to_nchw(r[k].ref, C.c_void_p(outs[k].data_ptr()), ops.stream_ptr()))

    first = dpb["ref_feature_bl"] is None
    info = _record(pnet, ins, run, outs, path, "pframe_first" if first else "pframe", arena_gib,
                   meta=(("ref_feature_el_channels", ins["ref_feature_el"].shape[1]),))
    return info, outs


def compile_pframe_layers(pnet, x_bl, x_el, dpb, path_bl, path_el, arena_gib=None):
    """Round 6: a P-frame as TWO plans, so that a caller without Python gets the look-ahead of
    LSSVC_extend.forward_one_frame(next_x_bl=...): the BASE LAYER alone (DMC.get_inter_layer_information, dmc_net.py:421-488) and the
    ENHANCEMENT LAYER given the base layer's results (LSSVC.forward_one_frame's own part, LSSVC_net.py:456-528). The engine
    (lssvc_engine_pframe_lookahead) runs EL(t) on the caller's stream and BL(t+1) on a second one -- the base layer of a P-frame needs
    the previous frame's BASE layer only -- and hands BL results from plan to plan as dense NHWC buffers.
      base-layer plan   in: x_bl, ref_frame_bl (clamped to [0, 1] inside the plan: test.py:249-250's clamp, idempotent on a DPB the
                        caller has clamped already), ref_feature_bl (steady-P only). out: bl_recon, bl_feature, bl_y_hat, bl_mv_hat
                        (NHWC, what the enhancement layer reads) + recon_bl, feature_bl (NCHW, the caller's DPB); bit slots 0..3
      enhancement plan  in: x_el, ref_frame_el, ref_feature_el (NCHW) + the four bl_* buffers. out: recon_el, feature_el, mv_hat,
                        warp_frame (NCHW); bit slots 4..7
    Same launches in the same order inside either layer as the whole-frame plan (compile_pframe): bit-identical results
    (tests/test_gpu_engine.py). The DPB decides first-P / steady-P as in compile_pframe. Returns (info_bl, info_el, outs_bl, outs_el)."""
    from .hip_ops import T
    H, W = pnet.shape_hr
    h, w = x_bl.shape[2], x_bl.shape[3]
    dev = pnet.device
    first = dpb["ref_feature_bl"] is None
    keys = pnet.STASH_KEYS
    ins_bl = {"x_bl": x_bl.contiguous(), "ref_frame_bl": dpb["ref_frame_bl"].contiguous()}
    if not first:
        ins_bl["ref_feature_bl"] = dpb["ref_feature_bl"].contiguous()

    def bl_body():
        t = {k: T.from_nchw(v) for k, v in ins_bl.items()}
        ref = t["ref_frame_bl"]
        ref = ops.clamp_(ops.copy(ref, T.empty(ref.H, ref.W, ref.C, dev)))
        fk = ops.Fork(dev)
        bl = pnet._bl_codec(t["x_bl"], ref, t.get("ref_feature_bl"), fk=fk)
        fk.close()
        return bl

    probe = bl_body()                                      # eager, once: the shapes of the four results
    shapes = {k: (probe[k].H, probe[k].W, probe[k].C) for k in keys}
    del probe
    flat = lambda k: torch.empty(shapes[k][0] * shapes[k][1] * shapes[k][2], dtype=torch.float32, device=dev)
    as_t = lambda buf, k: T(buf, shapes[k][0], shapes[k][1], shapes[k][2], shapes[k][2])
    outs_bl = {"bl_" + k: flat(k) for k in keys}
    outs_bl.update(recon_bl=torch.empty(1, 3, h, w, device=dev), feature_bl=torch.empty(1, shapes["feature"][2], h, w, device=dev))

    def run_bl():
        bl = bl_body()
        for k in keys:
            ops.copy(bl[k], as_t(outs_bl["bl_" + k], k))
        _nchw_out(bl["recon"], outs_bl["recon_bl"])
        _nchw_out(bl["feature"], outs_bl["feature_bl"])

    meta_bl = tuple(("bl_%s_%s" % (k, d), shapes[k][i]) for k in keys for i, d in enumerate(("h", "w", "c")))
    info_bl = _record(pnet, ins_bl, run_bl, outs_bl, path_bl, "pframe_first_bl" if first else "pframe_bl",
                      arena_gib if arena_gib is not None else max(0.5, 8.0 * H * W / (1152.0 * 1920.0)), meta=meta_bl)
    ins_el = {"x_el": x_el.contiguous(), "ref_frame_el": dpb["ref_frame_el"].contiguous(), "ref_feature_el": dpb["ref_feature_el"].contiguous()}
    ins_el.update({"bl_" + k: outs_bl["bl_" + k].clone() for k in keys})
    outs_el = {"recon_el": torch.empty(1, 3, H, W, device=dev), "feature_el": torch.empty(1, 48, H, W, device=dev),
               "mv_hat": torch.empty(1, 2, H, W, device=dev), "warp_frame": torch.empty(1, 3, H, W, device=dev)}

    def run_el():
        xe, ref_el, feat_el = (T.from_nchw(ins_el[k]) for k in ("x_el", "ref_frame_el", "ref_feature_el"))
        bl = {k: as_t(ins_el["bl_" + k], k) for k in keys}
        fk, pre = pnet._fork_el_head(xe, ref_el, feat_el)
        feature, recon_el, mv_hat, warp_frame = pnet._el_codec(xe, bl, ref_el, feat_el, fk=fk, pre=pre)
        fk.close()
        _nchw_out(recon_el, outs_el["recon_el"])
        _nchw_out(feature, outs_el["feature_el"])
        _nchw_out(mv_hat, outs_el["mv_hat"])
        _nchw_out(warp_frame, outs_el["warp_frame"])

    info_el = _record(pnet, ins_el, run_el, outs_el, path_el, "pframe_first_el" if first else "pframe_el",
                      arena_gib if arena_gib is not None else max(0.5, 24.0 * H * W / (1152.0 * 1920.0)),
                      meta=(("ref_feature_el_channels", ins_el["ref_feature_el"].shape[1]),) + meta_bl)
    return info_bl, info_el, outs_bl, outs_el


# ---- write_stream = 1: encoder and decoder plans (GPU launches + the host coder's steps between them) ------------------------
def _nchw_out(t, dst, clamp=False):
    _lib.check(_lib.lib.lssvc_nhwc_to_nchw(t.ref, C.c_void_p(dst.data_ptr()), ops.stream_ptr()))
    if clamp:
        _lib.check(_lib.lib.lssvc_clamp_inplace(C.c_void_p(dst.data_ptr()), dst.numel(), 0.0, 1.0, ops.stream_ptr()))


def compile_iframe_stream(inet, x_bl, x_el, path_enc, path_dec, arena_gib=None):
    """The two write_stream = 1 plans of an I-frame: IntraSS.encode_decode with bin paths, split into its encoder half
    (IntraNoAR.compress + IntraSS.compress, priors.py:422-437, IntraSS.py:304-314 -> the y and z strings of both layers)
    and its decoder half (decompress, priors.py:439-452, IntraSS.py:316-336). inet.update() must have been called.
    Returns (info_enc, info_dec, strings): the four rANS strings the recorded encoder pass produced."""
    from .hip_ops import T
    from .entropy_coder import SymbolSink, SymbolSource
    from . import bitstream
    assert inet._tables is not None, "call update() first"
    H, W = inet.shape_hr
    h, w = x_bl.shape[2], x_bl.shape[3]
    arena_gib = arena_gib if arena_gib is not None else max(0.25, 8.0 * H * W / (1152.0 * 1920.0))
    dev = inet.device
    mk = lambda: {"x_hat_bl": torch.empty(1, 3, h, w, device=dev), "x_hat_el": torch.empty(1, 3, H, W, device=dev),
                  "feature_el": torch.empty(1, 64, H, W, device=dev)}
    ins = {"x_bl": x_bl.contiguous(), "x_el": x_el.contiguous()}
    meta = (("pic_height_bl", h), ("pic_width_bl", w), ("pic_height_el", H), ("pic_width_el", W))
    strings = []

    def finish(outs, x_hat_bl, x_hat, feature):
        _nchw_out(x_hat_bl, outs["x_hat_bl"])
        _nchw_out(x_hat, outs["x_hat_el"])
        _nchw_out(feature, outs["feature_el"])

    outs_e = mk()

    def run_enc():
        del strings[:]
        st = inet._begin_layer()
        sinks = (SymbolSink(st), SymbolSink(st))
        x_hat_bl, y_hat_bl = inet._bl_codec(T.from_nchw(ins["x_bl"]), sinks=sinks)
        strings.extend([sinks[0].flush(), sinks[1].flush()])
        st = inet._begin_layer()
        sinks = (SymbolSink(st), SymbolSink(st))
        feature, x_hat = inet._el_codec(T.from_nchw(ins["x_el"]), x_hat_bl, y_hat_bl, sinks=sinks)
        strings.extend([sinks[0].flush(), sinks[1].flush()])
        finish(outs_e, x_hat_bl, x_hat, feature)

    info_e = _record(inet, ins, run_enc, outs_e, path_enc, "iframe_enc", arena_gib, meta, stream_mode=True)
    coded = list(strings)
    outs_d = mk()

    def run_dec():
        st = inet._begin_layer()
        x_hat_bl, y_hat_bl = inet._bl_codec(None, sources=(SymbolSource(coded[0], st), SymbolSource(coded[1], st)),
                                            lat_hw=bitstream.get_downsampled_shape(h, w, 64))
        st = inet._begin_layer()
        feature, x_hat = inet._el_codec(None, x_hat_bl, y_hat_bl, sources=(SymbolSource(coded[2], st), SymbolSource(coded[3], st)),
                                        lat_hw=bitstream.get_downsampled_shape(H, W, 64))
        finish(outs_d, x_hat_bl, x_hat, feature)

    info_d = _record(inet, {}, run_dec, outs_d, path_dec, "iframe_dec", arena_gib, meta, stream_mode=True)
    for k in outs_e:
        assert torch.equal(outs_e[k], outs_d[k]), "decoder plan does not reproduce the encoder's %s" % k
    return info_e, info_d, coded


def compile_pframe_stream(pnet, x_bl, x_el, dpb, path_enc, path_dec, arena_gib=None):
    """The two write_stream = 1 plans of a P-frame (LSSVC_extend.encode_decode_extend, LSSVC_net_extend.py:138-191, with
    DMCExtend's, dmc_net_extend.py:148-173): encoder half (compress: one rANS string per layer) and decoder half (decompress).
    The DPB decides first-P / steady-P as in compile_pframe. Outputs of both: the next DPB -- recon_bl (clamped to [0, 1], as
    the reference's base-layer decoder returns it, dmc_net_extend.py:138), feature_bl, recon_el, feature_el."""
    from .hip_ops import T
    from .entropy_coder import SymbolSink, SymbolSource
    assert pnet._tables is not None, "call update() first"
    H, W = pnet.shape_hr
    h, w = x_bl.shape[2], x_bl.shape[3]
    arena_gib = arena_gib if arena_gib is not None else max(0.5, 26.0 * H * W / (1152.0 * 1920.0))
    dev = pnet.device
    refs = {"ref_frame_bl": dpb["ref_frame_bl"].contiguous(), "ref_frame_el": dpb["ref_frame_el"].contiguous(),
            "ref_feature_el": dpb["ref_feature_el"].contiguous()}
    if dpb["ref_feature_bl"] is not None:
        refs["ref_feature_bl"] = dpb["ref_feature_bl"].contiguous()
    ins = dict(refs, x_bl=x_bl.contiguous(), x_el=x_el.contiguous())
    mk = lambda: {"recon_bl": torch.empty(1, 3, h, w, device=dev), "feature_bl": torch.empty(1, 64, h, w, device=dev),
                  "recon_el": torch.empty(1, 3, H, W, device=dev), "feature_el": torch.empty(1, 48, H, W, device=dev)}
    first = dpb["ref_feature_bl"] is None
    meta = (("ref_feature_el_channels", refs["ref_feature_el"].shape[1]),)
    strings = []

    def nhwc(d, k):
        return T.from_nchw(d[k]) if k in d else None

    def finish(outs, bl, recon_el, feature):
        _nchw_out(bl["recon"], outs["recon_bl"], clamp=True)
        _nchw_out(bl["feature"], outs["feature_bl"])
        _nchw_out(recon_el, outs["recon_el"])
        _nchw_out(feature, outs["feature_el"])

    outs_e = mk()

    def run_enc():
        del strings[:]
        xe, ref_el, feat_el = nhwc(ins, "x_el"), nhwc(ins, "ref_frame_el"), nhwc(ins, "ref_feature_el")
        fk, pre = pnet._fork_el_head(xe, ref_el, feat_el)           # as LSSVC_extend.encode: side-stream branches are part of the plan
        sink = SymbolSink(pnet._begin_layer())
        bl = pnet._bl_codec(nhwc(ins, "x_bl"), nhwc(ins, "ref_frame_bl"), nhwc(ins, "ref_feature_bl"), sink=sink, fk=fk)
        strings.append(sink.flush())
        sink = SymbolSink(pnet._begin_layer())
        feature, recon_el, _, _ = pnet._el_codec(xe, bl, ref_el, feat_el, sink=sink, fk=fk, pre=pre)
        fk.close()
        strings.append(sink.flush())
        finish(outs_e, bl, recon_el, feature)

    info_e = _record(pnet, ins, run_enc, outs_e, path_enc, "pframe_first_enc" if first else "pframe_enc", arena_gib, meta, stream_mode=True)
    coded = list(strings)
    outs_d = mk()

    def run_dec():
        ref_el, feat_el = nhwc(refs, "ref_frame_el"), nhwc(refs, "ref_feature_el")
        fk, pre = pnet._fork_el_head(None, ref_el, feat_el)         # as LSSVC_extend.decode
        bl = pnet._bl_codec(None, nhwc(refs, "ref_frame_bl"), nhwc(refs, "ref_feature_bl"), source=SymbolSource(coded[0], pnet._begin_layer()), fk=fk)
        feature, recon_el, _, _ = pnet._el_codec(None, bl, ref_el, feat_el, source=SymbolSource(coded[1], pnet._begin_layer()), fk=fk, pre=pre)
        fk.close()
        finish(outs_d, bl, recon_el, feature)

    info_d = _record(pnet, refs, run_dec, outs_d, path_dec, "pframe_first_dec" if first else "pframe_dec", arena_gib, meta, stream_mode=True)
    for k in outs_e:
        assert torch.equal(outs_e[k], outs_d[k]), "decoder plan does not reproduce the encoder's %s" % k
    return info_e, info_d, coded
