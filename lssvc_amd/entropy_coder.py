"""Host entropy coder of write_stream=1: thin numpy/ctypes wrappers over the C-ABI rANS coder in
liblssvc_hip.so (include/lssvc_hip.h, "host entropy coder"). API mirrors what the reference's Python
calls on its pybind11 modules (video_entropy_models.py:8-61): reset / encode_with_indexes / flush,
set_stream / decode_stream, pmf_to_quantized_cdf -- but with int32 numpy planes instead of Python lists."""
import ctypes as C
import time

import numpy as np

from ._lib import lib, check, CdfTable


def pmf_to_quantized_cdf(pmf, precision=16):
    p = np.ascontiguousarray(pmf, dtype=np.float32).reshape(-1)
    out = np.empty(p.size + 1, dtype=np.uint32)
    check(lib.lssvc_pmf_to_quantized_cdf(p.ctypes.data, p.size, precision, out.ctypes.data))
    return out


class Tables:
    """A set of quantised CDFs (rows), their used lengths and symbol offsets (CdfHelper in the reference)."""

    def __init__(self, cdfs, sizes, offsets):
        self.cdfs = np.ascontiguousarray(cdfs, dtype=np.int32)
        self.sizes = np.ascontiguousarray(sizes, dtype=np.int32).reshape(-1)
        self.offsets = np.ascontiguousarray(offsets, dtype=np.int32).reshape(-1)
        assert self.cdfs.ndim == 2 and self.sizes.size == self.cdfs.shape[0] == self.offsets.size
        self.c = CdfTable(self.cdfs.ctypes.data, self.cdfs.shape[0], self.cdfs.shape[1], self.sizes.ctypes.data,
                          self.offsets.ctypes.data)

    @staticmethod
    def from_pmfs(pmf, tail_mass, pmf_length, offsets):
        """EntropyCoder.pmf_to_cdf (video_entropy_models.py:24-30): row i = cdf(pmf[i,:len_i] ++ tail_i)."""
        pmf = np.asarray(pmf, dtype=np.float32)
        lengths = np.asarray(pmf_length, dtype=np.int64).reshape(-1)
        cdfs = np.zeros((pmf.shape[0], int(lengths.max()) + 2), dtype=np.int32)
        for i in range(pmf.shape[0]):
            prob = np.concatenate([pmf[i, :lengths[i]], np.asarray(tail_mass[i], dtype=np.float32).reshape(-1)])
            c = pmf_to_quantized_cdf(prob)
            cdfs[i, :c.size] = c
        return Tables(cdfs, lengths + 2, offsets)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32).reshape(-1)


def _prof(key, t0, count_key=None, n=0):
    from . import hip_ops
    if hip_ops.STREAM_PROF is not None:
        hip_ops.STREAM_PROF[key] = hip_ops.STREAM_PROF.get(key, 0.0) + time.perf_counter() - t0
        if count_key:
            hip_ops.STREAM_PROF[count_key] = hip_ops.STREAM_PROF.get(count_key, 0) + int(n)


class RansEncoder:
    def __init__(self):
        self.h = C.c_void_p(lib.lssvc_rans_encoder_new())

    def reset(self):
        lib.lssvc_rans_encoder_reset(self.h)

    def encode_with_indexes(self, symbols, indexes, tables):
        t0 = time.perf_counter()
        if getattr(symbols, "dtype", None) == np.int16 and getattr(indexes, "dtype", None) == np.int16 \
                and symbols.flags.c_contiguous and indexes.flags.c_contiguous:
            assert symbols.size == indexes.size
            check(lib.lssvc_rans_encode_with_indexes_i16(self.h, symbols.ctypes.data, indexes.ctypes.data, symbols.size, C.byref(tables.c)))
            n = symbols.size
        else:
            s, i = _i32(symbols), _i32(indexes)
            assert s.size == i.size
            check(lib.lssvc_rans_encode_with_indexes(self.h, s.ctypes.data, i.ctypes.data, s.size, C.byref(tables.c)))
            n = s.size
        _prof("rans_enc_s", t0, "enc_symbols", n)

    def flush(self):
        t0 = time.perf_counter()
        n = lib.lssvc_rans_encoder_flush(self.h)
        out = C.string_at(lib.lssvc_rans_encoder_bytes(self.h), n)
        _prof("rans_enc_s", t0)
        return out

    def __del__(self):
        if lib is not None:
            lib.lssvc_rans_encoder_free(self.h)


class RansDecoder:
    def __init__(self):
        self.h = C.c_void_p(lib.lssvc_rans_decoder_new())

    def set_stream(self, data):
        check(lib.lssvc_rans_decoder_set_stream(self.h, data, len(data)))

    def decode_stream(self, indexes, tables, out=None):
        """-> decoded symbols. int16 indexes give int16 symbols (written into `out` if given: a pinned staging view)."""
        t0 = time.perf_counter()
        if getattr(indexes, "dtype", None) == np.int16 and indexes.flags.c_contiguous:
            if out is None:
                out = np.empty(indexes.size, dtype=np.int16)
            assert out.dtype == np.int16 and out.size == indexes.size and out.flags.c_contiguous
            check(lib.lssvc_rans_decode_stream_i16(self.h, indexes.ctypes.data, indexes.size, C.byref(tables.c), out.ctypes.data))
        else:
            i = _i32(indexes)
            out = np.empty(i.size, dtype=np.int32)
            check(lib.lssvc_rans_decode_stream(self.h, i.ctypes.data, i.size, C.byref(tables.c), out.ctypes.data))
        _prof("rans_dec_s", t0, "dec_symbols", out.size)
        return out

    def __del__(self):
        if lib is not None:
            lib.lssvc_rans_decoder_free(self.h)


class SymbolSink:
    """Encoder side of one rANS string: latents are pushed in coding order (as the reference's
    entropy_coder.reset_encoder / *.encode(...) / flush_encoder sequence, dmc_net_extend.py:89-95).
    With a hip_ops.SymbolStage the pushes are PlaneRefs into the stage's device buffer: nothing crosses PCIe until flush(),
    which brings the whole staged region down in one asynchronous copy into pinned memory and codes the planes in place."""

    def __init__(self, stage=None):
        self.enc = RansEncoder()
        self.enc.reset()
        self.stage = stage
        self.pending = []
        self.symbols = 0

    def push(self, symbols, indexes, tables):
        if self.stage is not None and hasattr(indexes, "off"):
            self.pending.append((symbols, indexes, tables))
        else:
            self.enc.encode_with_indexes(symbols, indexes, tables)
            self.symbols += int(np.asarray(indexes).size)

    def flush(self):
        if self.pending:
            st = self.stage
            st.download(min(min(r.off for r in (r_sym, r_idx) if r is not None) for r_sym, r_idx, _ in self.pending), st.used)
            for r_sym, r_idx, tables in self.pending:
                self.enc.encode_with_indexes(st.numpy(r_sym), st.numpy(r_idx), tables)
                self.symbols += r_idx.n
            self.pending = []
        return self.enc.flush()


class SymbolSource:
    """Decoder side: pulls latents from one rANS string in the same order. With a SymbolStage, int16 index planes that live
    in its pinned buffer are decoded into a fresh region of the same buffer (which the import kernel's upload reads)."""

    def __init__(self, string, stage=None):
        self.dec = RansDecoder()
        self.dec.set_stream(string)
        self.stage = stage

    def pull(self, indexes, tables):
        out = None
        if self.stage is not None and getattr(indexes, "dtype", None) == np.int16:
            out = self.stage.numpy(self.stage.alloc(indexes.size))
        return self.dec.decode_stream(indexes, tables, out=out)
