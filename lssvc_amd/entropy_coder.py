"""Host entropy coder of write_stream=1: thin numpy/ctypes wrappers over the C-ABI rANS coder in
liblssvc_hip.so (include/lssvc_hip.h, "host entropy coder"). API mirrors what the reference's Python
calls on its pybind11 modules (video_entropy_models.py:8-61): reset / encode_with_indexes / flush,
set_stream / decode_stream, pmf_to_quantized_cdf -- but with int32 numpy planes instead of Python lists."""
import ctypes as C

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


class RansEncoder:
    def __init__(self):
        self.h = C.c_void_p(lib.lssvc_rans_encoder_new())

    def reset(self):
        lib.lssvc_rans_encoder_reset(self.h)

    def encode_with_indexes(self, symbols, indexes, tables):
        s, i = _i32(symbols), _i32(indexes)
        assert s.size == i.size
        check(lib.lssvc_rans_encode_with_indexes(self.h, s.ctypes.data, i.ctypes.data, s.size, C.byref(tables.c)))

    def flush(self):
        n = lib.lssvc_rans_encoder_flush(self.h)
        return C.string_at(lib.lssvc_rans_encoder_bytes(self.h), n)

    def __del__(self):
        if lib is not None:
            lib.lssvc_rans_encoder_free(self.h)


class RansDecoder:
    def __init__(self):
        self.h = C.c_void_p(lib.lssvc_rans_decoder_new())

    def set_stream(self, data):
        check(lib.lssvc_rans_decoder_set_stream(self.h, data, len(data)))

    def decode_stream(self, indexes, tables):
        i = _i32(indexes)
        out = np.empty(i.size, dtype=np.int32)
        check(lib.lssvc_rans_decode_stream(self.h, i.ctypes.data, i.size, C.byref(tables.c), out.ctypes.data))
        return out

    def __del__(self):
        if lib is not None:
            lib.lssvc_rans_decoder_free(self.h)


class SymbolSink:
    """Encoder side of one rANS string: latents are pushed in coding order (as the reference's
    entropy_coder.reset_encoder / *.encode(...) / flush_encoder sequence, dmc_net_extend.py:89-95)."""

    def __init__(self):
        self.enc = RansEncoder()
        self.enc.reset()

    def push(self, symbols, indexes, tables):
        self.enc.encode_with_indexes(symbols, indexes, tables)

    def flush(self):
        return self.enc.flush()


class SymbolSource:
    """Decoder side: pulls latents from one rANS string in the same order."""

    def __init__(self, string):
        self.dec = RansDecoder()
        self.dec.set_stream(string)

    def pull(self, indexes, tables):
        return self.dec.decode_stream(indexes, tables)
