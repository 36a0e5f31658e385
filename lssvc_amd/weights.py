"""Checkpoint handling: strict validation against the reference state-dict layout and one-time
re-layout of every tensor into what the HIP kernels consume -- done by the library itself (lssvc_prepare_weights,
csrc/weight_prep.cpp), the same call the engine makes for a caller without Python (lssvc_engine_load_checkpoint); this
module only names the layers, owns the device copies and remembers each blob's recipe for the plan compiler.

Layouts produced here (see include/lssvc_hip.h):
  conv     OIHW fp32  ->  [chunk][ky][kx][m][8]   (input channels of every concatenated input segment
                           zero-padded to a multiple of 8, output channels padded to 16; for a
                           sub-pixel conv the output axis is permuted to (dy,dx)-major so that the
                           kernel's PixelShuffle store is a plain float4 per lane)
  convT    (Cin,Cout,3,3) -> an equivalent plain conv: stride 1 = flipped 3x3; stride 2 (pad 1,
                           output_padding 1) = 2x2 conv producing the 4 output phases + pixel shuffle
  GDN      beta/gamma de-reparametrised once (gdn.py:31-33 / video_net_component.py:86-93) into a
           1x1 conv (weight gamma, bias beta) applied to x^2 with a fused x*rsqrt / x*sqrt / x/sqrt epilogue
  depthwise (C,1,3,3) -> [9][C]
  BitEstimator / EntropyBottleneck parameters -> [rows][C] tables with softplus / tanh pre-applied
"""
import ctypes as C

import torch

from .synth import load_manifest

CK = 8


class CheckpointError(RuntimeError):
    pass


def strip_module_prefix(sd):
    """'module.' prefix removal (IntraSS.py:193-198, LSSVC_net.py:141-149) and optional 'state_dict' unwrap."""
    if "state_dict" in sd and not torch.is_tensor(sd["state_dict"]):
        sd = sd["state_dict"]
    return {(k[7:] if k[:7] == "module." else k): v for k, v in sd.items()}


def validate(sd, manifest_name, ignore=(), resizable=()):
    """Same contract as nn.Module.load_state_dict(strict=True): every expected key present with the
    expected shape, nothing unexpected. `resizable` keys (CDF buffers) may have any shape."""
    want = {e["key"]: tuple(e["shape"]) for e in load_manifest(manifest_name)}
    missing = [k for k in want if k not in sd]
    unexpected = [k for k in sd if k not in want and k not in ignore]
    bad = [k for k in want if k in sd and not any(k.endswith(r) for r in resizable) and tuple(sd[k].shape) != want[k]]
    if missing or unexpected or bad:
        raise CheckpointError("Error(s) in loading state_dict: missing %s; unexpected %s; size mismatch %s"
                              % (missing[:5], unexpected[:5], bad[:5]))


class WeightStore:
    """Lazily prepared, cached device copies of one model's tensors."""

    def __init__(self, sd, device):
        self.sd = {k: (v.detach().to("cpu", torch.float32).contiguous() if v.is_floating_point() else v.detach().cpu())
                   for k, v in sd.items()}
        self.device = device
        self._cache = {}
        self.regions = {}               # device pointer -> (bytes, host tensor) of every prepared tensor
        self.recipes = {}               # device pointer -> (recipe, blob index): how plan_runtime.cpp rebuilds it from a raw checkpoint
        self.force_f32 = set()          # layers the fp16 range audit moved to the exact fp32 conv kernel (hip_ops.RangeAudit)
        self._table = None              # the checkpoint as a C array of lssvc_tensor (built on first use)

    def _capturing(self):
        return torch.device(self.device).type == "cuda" and torch.cuda.is_current_stream_capturing()

    def has(self, key):
        return key in self.sd

    def raw(self, key):
        return self.sd[key]

    def _dev(self, t):
        """A device copy of an ad-hoc host tensor that is NOT a function of the checkpoint alone (the bottleneck medians of
        update()'s tables): registered without a recipe, so a compiled plan stores its bytes."""
        host = t.contiguous()
        assert not self._capturing(), "a weight tensor would be uploaded inside a hipGraph capture"
        dev = host.to(self.device)
        self.regions[dev.data_ptr()] = (dev.numel() * dev.element_size(), host)
        return dev

    # ---- the checkpoint as the library sees it -------------------------------------------------------------------
    def _ckpt(self):
        if self._table is None:
            from ._lib import Tensor
            names = [k for k, v in self.sd.items() if v.is_floating_point() and v.dim() <= 4]
            arr = (Tensor * len(names))()
            for i, k in enumerate(names):
                v = self.sd[k]
                arr[i].name = k.encode()
                arr[i].data = v.data_ptr()
                arr[i].ndim = v.dim()
                for d in range(v.dim()):
                    arr[i].shape[d] = v.shape[d]
            self._table = (arr, len(names))
        return self._table

    def prepare(self, kind, name, name2="", splits=(), flag=0):
        """One lssvc_prepare_weights call -> (device blobs, scalars, dims); every blob is registered with its recipe."""
        from ._lib import lib, check, PrepSpec, PREP_MAX_BLOBS
        spec = PrepSpec()
        spec.kind, spec.name, spec.name2, spec.n_splits, spec.flag = kind, name.encode(), name2.encode(), len(splits), int(flag)
        for i, c in enumerate(splits):
            spec.splits[i] = int(c)
        arr, n = self._ckpt()
        nb, nbytes, scalars, dims = C.c_int32(), (C.c_int64 * PREP_MAX_BLOBS)(), (C.c_float * 4)(), (C.c_int32 * 8)()
        check(lib.lssvc_prepare_weights(arr, n, C.byref(spec), C.byref(nb), nbytes, scalars, dims, None))
        hosts = [torch.empty(int(nbytes[i]), dtype=torch.uint8) for i in range(nb.value)]
        ptrs = (C.c_void_p * PREP_MAX_BLOBS)(*[h.data_ptr() for h in hosts])
        check(lib.lssvc_prepare_weights(arr, n, C.byref(spec), C.byref(nb), nbytes, scalars, dims, ptrs))
        recipe = (kind, name, name2, tuple(int(c) for c in splits), int(flag))
        devs = []
        # (a layer variant first used inside a capture would get its weights from the graph's private pool through a captured copy of
        # pageable memory: every plan's first call is eager precisely so that this never happens -- fail loudly if it ever does)
        assert not self._capturing(), "weights of %s would be prepared inside a hipGraph capture" % name
        for i, h in enumerate(hosts):
            dev = h.to(self.device)
            self.regions[dev.data_ptr()] = (dev.numel(), h)
            self.recipes[dev.data_ptr()] = (recipe, i)
            devs.append(dev)
        return devs, [float(x) for x in scalars], [int(x) for x in dims]

    @staticmethod
    def _f32(t):
        return t.view(torch.float32)

    def conv(self, name, splits, pixel_shuffle=False):
        from ._lib import PREP_CONV
        key = ("conv", name, tuple(splits), pixel_shuffle)
        if key not in self._cache:
            (wp, bp), _, d = self.prepare(PREP_CONV, name, splits=splits, flag=pixel_shuffle)
            self._cache[key] = (self._f32(wp), self._f32(bp), d[0], d[1], d[2], d[3])
        return self._cache[key]

    def conv_f16x3(self, name, splits, pixel_shuffle=False):
        from ._lib import PREP_CONV_F16X3
        key = ("conv16", name, tuple(splits), pixel_shuffle)
        if key not in self._cache:
            (planes,), sc, _ = self.prepare(PREP_CONV_F16X3, name, splits=splits, flag=pixel_shuffle)
            self._cache[key] = (planes.view(torch.float16), sc[0])
        return self._cache[key]

    def ffn_f16x3(self, ffn_prefix, pre_name=None):
        """Device blobs for lssvc_ffn_f16x3: ConvFFN `ffn_prefix`.conv.{0,2} and, optionally, the leading 1x1 conv."""
        from ._lib import PREP_FFN_F16X3
        key = ("ffn16", ffn_prefix, pre_name)
        if key not in self._cache:
            blobs, sc, d = self.prepare(PREP_FFN_F16X3, ffn_prefix, name2=pre_name or "")
            rec = {"w1": blobs[0].view(torch.float16), "u1": sc[0], "w2": blobs[1].view(torch.float16), "u2": sc[1], "hidden": d[0], "C": d[1],
                   "b1": self._f32(blobs[2]), "b2": self._f32(blobs[3])}
            if pre_name is not None:
                rec.update({"wp": blobs[4].view(torch.float16), "up": sc[2], "bp": self._f32(blobs[5]), "pre_cin": d[2]})
            self._cache[key] = rec
        return self._cache[key]

    def conv_t(self, name, stride):
        from ._lib import PREP_CONVT
        key = ("convT", name, stride)
        if key not in self._cache:
            (wp, bp), _, d = self.prepare(PREP_CONVT, name, flag=stride)
            self._cache[key] = (self._f32(wp), self._f32(bp), d[0], d[1], d[2], d[3], d[4], bool(d[5]))
        return self._cache[key]

    def dwconv(self, name):
        from ._lib import PREP_DWCONV
        key = ("dw", name)
        if key not in self._cache:
            (w, b), _, _ = self.prepare(PREP_DWCONV, name)
            self._cache[key] = (self._f32(w), self._f32(b))
        return self._cache[key]

    def gdn(self, name, flavour):
        """flavour 'intra' (gdn.py + others.py reparam buffers) | 'inter' (video_net_component.py constants)."""
        from ._lib import PREP_GDN
        key = ("gdn", name, flavour)
        if key not in self._cache:
            (wp, bp, planes), sc, d = self.prepare(PREP_GDN, name, flag=0 if flavour == "intra" else 1)
            self._cache[key] = (self._f32(wp), self._f32(bp), d[0], d[1], 1, 1)
            self._cache[("gdn16", name, flavour)] = (planes.view(torch.float16), sc[0])
        return self._cache[key]

    def gdn_f16x3(self, name, flavour):
        self.gdn(name, flavour)
        return self._cache[("gdn16", name, flavour)]

    def vector(self, key):
        from ._lib import PREP_VECTOR
        k = ("vec", key)
        if k not in self._cache:
            (v,), _, _ = self.prepare(PREP_VECTOR, key)
            self._cache[k] = self._f32(v)
        return self._cache[k]

    def bit_estimator(self, name):
        """[11][C]: softplus(h_i), b_i, tanh(a_i) for i=1..3, softplus(h_4), b_4 (video_entropy_models.py:110-129)."""
        from ._lib import PREP_BIT_ESTIMATOR
        key = ("be", name)
        if key not in self._cache:
            (t,), _, d = self.prepare(PREP_BIT_ESTIMATOR, name)
            self._cache[key] = self._f32(t).view(11, d[0])
        return self._cache[key]

    def entropy_bottleneck(self, name):
        """[59][C]: softplus(matrices) 3+9+9+9+3, biases 3+3+3+3+1, tanh(factors) 3x4, median
        (img_entropy_models.py:483-502, 432-434)."""
        from ._lib import PREP_ENTROPY_BOTTLENECK
        key = ("eb", name)
        if key not in self._cache:
            (t,), _, d = self.prepare(PREP_ENTROPY_BOTTLENECK, name)
            self._cache[key] = self._f32(t).view(59, d[0])
        return self._cache[key]
