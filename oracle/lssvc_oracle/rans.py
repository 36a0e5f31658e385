"""ORACLE (test infrastructure only) -- ctypes wrapper over oracle/c/rans_oracle.c (plain-C restatement of
the reference's BufferedRansEncoder / RansDecoder / pmf_to_quantized_cdf). See that file's header for the
parity status (CDF quantiser pinned, rANS byte stream unpinned)."""
import ctypes as C
import os

import numpy as np

_LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "_build", "librans_oracle.so")
_lib = C.CDLL(_LIB)
_lib.oracle_encoder_new.restype = C.c_void_p
_lib.oracle_decoder_new.restype = C.c_void_p
_lib.oracle_encoder_flush.restype = C.c_int64
_I32 = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")


def _tables(cdfs, sizes, offsets):
    cdfs = np.ascontiguousarray(cdfs, dtype=np.int32)
    return cdfs, np.ascontiguousarray(sizes, dtype=np.int32), np.ascontiguousarray(offsets, dtype=np.int32)


class Encoder:
    def __init__(self):
        self.h = C.c_void_p(_lib.oracle_encoder_new())

    def encode_with_indexes(self, symbols, indexes, cdfs, sizes, offsets):
        s = np.ascontiguousarray(symbols, dtype=np.int32).reshape(-1)
        i = np.ascontiguousarray(indexes, dtype=np.int32).reshape(-1)
        cdfs, sizes, offsets = _tables(cdfs, sizes, offsets)
        _lib.oracle_encode_with_indexes(self.h, s.ctypes.data_as(C.c_void_p), i.ctypes.data_as(C.c_void_p), C.c_int64(s.size),
                                        cdfs.ctypes.data_as(C.c_void_p), C.c_int32(cdfs.shape[1]),
                                        sizes.ctypes.data_as(C.c_void_p), offsets.ctypes.data_as(C.c_void_p))

    def flush(self):
        out = C.POINTER(C.c_uint8)()
        n = _lib.oracle_encoder_flush(self.h, C.byref(out))
        data = bytes(bytearray(out[:n]))
        _lib.oracle_free(out)
        return data

    def reset(self):
        _lib.oracle_encoder_reset(self.h)

    def __del__(self):
        _lib.oracle_encoder_free(self.h)


class Decoder:
    def __init__(self):
        self.h = C.c_void_p(_lib.oracle_decoder_new())

    def set_stream(self, data):
        _lib.oracle_decoder_set_stream(self.h, C.c_char_p(data), C.c_int64(len(data)))

    def decode_stream(self, indexes, cdfs, sizes, offsets):
        i = np.ascontiguousarray(indexes, dtype=np.int32).reshape(-1)
        cdfs, sizes, offsets = _tables(cdfs, sizes, offsets)
        out = np.empty(i.size, dtype=np.int32)
        _lib.oracle_decode_stream(self.h, i.ctypes.data_as(C.c_void_p), C.c_int64(i.size), cdfs.ctypes.data_as(C.c_void_p),
                                  C.c_int32(cdfs.shape[1]), sizes.ctypes.data_as(C.c_void_p),
                                  offsets.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        return out

    def __del__(self):
        _lib.oracle_decoder_free(self.h)


def pmf_to_quantized_cdf(pmf, precision=16):
    p = np.ascontiguousarray(pmf, dtype=np.float32)
    out = np.empty(p.size + 1, dtype=np.uint32)
    _lib.oracle_pmf_to_quantized_cdf(p.ctypes.data_as(C.c_void_p), C.c_int32(p.size), C.c_int32(precision),
                                     out.ctypes.data_as(C.c_void_p))
    return out
