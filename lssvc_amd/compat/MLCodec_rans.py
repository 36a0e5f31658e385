"""MLCodec_rans stand-in over liblssvc_hip.so's host rANS coder (include/lssvc_hip.h, lssvc_rans_*).

Two API sets live here, because the reference's Python and its C++ source drifted apart:

  * what `src/cpp/rans/rans_interface.cpp:246-261` exports (and priors.py:627-770, video_entropy_models.py:8-61 use):
        BufferedRansEncoder().encode_with_indexes(symbols, indexes, cdfs, cdfs_sizes, offsets) ; .flush() -> bytes ; .reset()
        RansDecoder().set_stream(bytes) ; .decode_stream(indexes, cdfs, cdfs_sizes, offsets) -> int32 array
  * what `src/entropy_models/img_entropy_models.py:16-27,305-361` calls (the CompressAI-era names):
        RansEncoder().encode_with_indexes(symbols, indexes, cdfs, cdfs_sizes, offsets) -> bytes
        RansDecoder().decode_with_indexes(stream, indexes, cdfs, cdfs_sizes, offsets) -> list[int]
    plus the helpers video_entropy_models.py touches on its decoder/encoder objects
    (set_cdf / decode_stream_only_indexes / get_encoded_stream, :33-61,96-101).

Arguments may be Python lists (the reference passes `.tolist()`), numpy arrays or anything np.asarray takes;
`cdfs` is the 2-D [n_cdfs][max_len] int table, `cdfs_sizes[i]` the entries row i uses (pmf length + 2),
`offsets[i]` the symbol value of slot 0. Errors the reference's C++ asserts on (index / size out of range)
raise ValueError here instead of aborting the process."""
import ctypes as C

import numpy as np

from .._lib import lib, CdfTable

__name__ = "MLCodec_rans"


def _check(status):
    if status != 0:
        raise ValueError(lib.lssvc_last_error().decode("utf-8", "replace"))


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32).reshape(-1)


class _Table:
    """C view of (cdfs, cdfs_sizes, offsets); keeps the numpy buffers alive. Converting a list-of-lists table costs
    more than coding a small plane, and the reference passes the same table objects over and over, so the last
    conversion is cached per coder object keyed by the identity of the three arguments."""

    def __init__(self, cdfs, sizes, offsets):
        self.cdfs = np.ascontiguousarray(cdfs, dtype=np.int32)
        if self.cdfs.ndim != 2:
            raise ValueError("cdfs must be a 2-D table [n_cdfs][max_length]")
        self.sizes, self.offsets = _i32(sizes), _i32(offsets)
        if not (self.sizes.size == self.offsets.size == self.cdfs.shape[0]):
            raise ValueError("cdfs (%d rows), cdfs_sizes (%d) and offsets (%d) disagree"
                             % (self.cdfs.shape[0], self.sizes.size, self.offsets.size))
        self.c = CdfTable(self.cdfs.ctypes.data, self.cdfs.shape[0], self.cdfs.shape[1], self.sizes.ctypes.data,
                          self.offsets.ctypes.data)


class _TableCache:
    def __init__(self):
        self._key, self._keep, self._table = None, None, None

    def get(self, cdfs, sizes, offsets):
        key = (id(cdfs), id(sizes), id(offsets))
        if key != self._key:
            self._table = _Table(cdfs, sizes, offsets)
            self._key, self._keep = key, (cdfs, sizes, offsets)      # hold the objects so the ids stay theirs
        return self._table


class BufferedRansEncoder:
    """rans_interface.hpp:50-69. Symbols are buffered by encode_with_indexes and entropy-coded by flush()."""

    def __init__(self):
        self._h = C.c_void_p(lib.lssvc_rans_encoder_new())
        self._tables = _TableCache()
        self._last = b""

    def encode_with_indexes(self, symbols, indexes, cdfs, cdfs_sizes, offsets):
        s, i = _i32(symbols), _i32(indexes)
        if s.size != i.size:
            raise ValueError("symbols (%d) and indexes (%d) differ in length" % (s.size, i.size))
        t = self._tables.get(cdfs, cdfs_sizes, offsets)
        _check(lib.lssvc_rans_encode_with_indexes(self._h, s.ctypes.data, i.ctypes.data, s.size, C.byref(t.c)))

    def flush(self):
        """Entropy-code everything pending -> bytes; the pending list is emptied (rans_interface.cpp:144-178)."""
        n = lib.lssvc_rans_encoder_flush(self._h)
        self._last = C.string_at(lib.lssvc_rans_encoder_bytes(self._h), n)
        lib.lssvc_rans_encoder_reset(self._h)
        return self._last

    def reset(self):
        lib.lssvc_rans_encoder_reset(self._h)

    def get_encoded_stream(self):
        """The last flush()ed stream as a uint8 array (video_entropy_models.py:99-100 calls .tobytes() on it)."""
        return np.frombuffer(self._last, dtype=np.uint8)

    def __del__(self):
        if lib is not None and getattr(self, "_h", None):
            lib.lssvc_rans_encoder_free(self._h)


class RansEncoder:
    """One-shot encoder (img_entropy_models.py:16-24): encode_with_indexes(...) -> bytes."""

    def __init__(self):
        self._enc = BufferedRansEncoder()

    def encode_with_indexes(self, symbols, indexes, cdfs, cdfs_sizes, offsets):
        self._enc.reset()
        self._enc.encode_with_indexes(symbols, indexes, cdfs, cdfs_sizes, offsets)
        return self._enc.flush()


class RansDecoder:
    """rans_interface.hpp:71-96 (set_stream / decode_stream, cursor persists across decode_stream calls) plus the
    one-shot decode_with_indexes of img_entropy_models.py:26-27,354-360."""

    def __init__(self):
        self._h = C.c_void_p(lib.lssvc_rans_decoder_new())
        self._tables = _TableCache()
        self._stream = None
        self._fixed = None

    def set_stream(self, stream):
        if isinstance(stream, str):                      # pybind11 hands std::string over; accept latin-1 text too
            stream = stream.encode("latin-1")
        elif not isinstance(stream, (bytes, bytearray)):
            stream = np.ascontiguousarray(stream, dtype=np.uint8).tobytes()      # video_entropy_models.py:103
        self._stream = bytes(stream)                     # the C side reads it in place: keep it alive
        _check(lib.lssvc_rans_decoder_set_stream(self._h, self._stream, len(self._stream)))

    def _decode(self, indexes, table):
        i = _i32(indexes)
        out = np.empty(i.size, dtype=np.int32)
        _check(lib.lssvc_rans_decode_stream(self._h, i.ctypes.data, i.size, C.byref(table.c), out.ctypes.data))
        return out

    def decode_stream(self, indexes, cdfs, cdfs_sizes, offsets):
        if self._stream is None:
            raise ValueError("decode_stream before set_stream")
        return self._decode(indexes, self._tables.get(cdfs, cdfs_sizes, offsets))

    def decode_with_indexes(self, stream, indexes, cdfs, cdfs_sizes, offsets):
        self.set_stream(stream)
        return self.decode_stream(indexes, cdfs, cdfs_sizes, offsets).tolist()

    # -- helpers video_entropy_models.py:35-36,59-61 calls on the decoder object
    def set_cdf(self, cdfs, cdfs_sizes, offsets):
        self._fixed = _Table(cdfs, cdfs_sizes, offsets)

    def decode_stream_only_indexes(self, indexes):
        if self._fixed is None:
            raise ValueError("decode_stream_only_indexes before set_cdf")
        return self._decode(indexes, self._fixed)

    def __del__(self):
        if lib is not None and getattr(self, "_h", None):
            lib.lssvc_rans_decoder_free(self._h)
