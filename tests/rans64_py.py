"""An INDEPENDENT pure-Python (arbitrary-precision int) implementation of the published rANS64 coder --
Fabian Giesen's ryg_rans `rans64.h` (public domain; the reference pins it at commit c9d162d in its CMake and does
not vendor it) plus the two bit-bypass primitives CompressAI added to its copy -- and of the symbol protocol the
reference wraps around it (src/cpp/rans/rans_interface.cpp:85-244: 16-bit precision, reverse-order encode, escape
slot + 4-bit bypass digits). Written from the algorithm's published definition, NOT from lssvc_amd/csrc/rans_host.cpp
or oracle/c/rans_oracle.c: it is the second opinion those two are checked against (tests/test_rans_kat.py).
Test infrastructure only; pure Python loops, small cases."""
import struct

L = 1 << 31            # RANS64_L: lower bound of the normalisation interval
PRECISION = 16         # rans_interface.cpp: constexpr int precision = 16
BYPASS_BITS = 4        # bypass_precision
MAX_BYPASS = (1 << BYPASS_BITS) - 1


def _symbols(symbols, indexes, cdfs, sizes, offsets):
    """The (start, range, bypass) list BufferedRansEncoder::encode_with_indexes builds."""
    out = []
    for s, ci in zip(symbols, indexes):
        cdf = cdfs[ci]
        max_value = int(sizes[ci]) - 2
        value = int(s) - int(offsets[ci])
        raw = 0
        if value < 0:
            raw = -2 * value - 1
            value = max_value
        elif value >= max_value:
            raw = 2 * (value - max_value)
            value = max_value
        out.append((int(cdf[value]), int(cdf[value + 1]) - int(cdf[value]), False))
        if value == max_value:
            n_bypass = 0
            while (raw >> (n_bypass * BYPASS_BITS)) != 0:
                n_bypass += 1
            val = n_bypass
            while val >= MAX_BYPASS:
                out.append((MAX_BYPASS, MAX_BYPASS + 1, True))
                val -= MAX_BYPASS
            out.append((val, val + 1, True))
            for j in range(n_bypass):
                v = (raw >> (j * BYPASS_BITS)) & MAX_BYPASS
                out.append((v, v + 1, True))
    return out


def encode(symbols, indexes, cdfs, sizes, offsets):
    """-> bytes of BufferedRansEncoder.encode_with_indexes(...) + flush()."""
    syms = _symbols(symbols, indexes, cdfs, sizes, offsets)
    x = L                                                   # Rans64EncInit
    words = []                                              # emitted back to front
    for start, rng, bypass in reversed(syms):
        if not bypass:                                      # Rans64EncPut(start, freq, scale_bits = 16)
            assert 0 < rng <= (1 << PRECISION) and start + rng <= (1 << PRECISION)
            x_max = ((L >> PRECISION) << 32) * rng
            if x >= x_max:
                words.append(x & 0xFFFFFFFF)
                x >>= 32
            x = ((x // rng) << PRECISION) + (x % rng) + start
        else:                                               # Rans64EncPutBits(val, nbits = 4)
            freq = 1 << (16 - BYPASS_BITS)
            x_max = ((L >> 16) << 32) * freq
            if x >= x_max:
                words.append(x & 0xFFFFFFFF)
                x >>= 32
            x = (x << BYPASS_BITS) | start
        assert L <= x < (1 << 63)
    words.append(x >> 32)                                   # Rans64EncFlush: ptr[0] = low, ptr[1] = high
    words.append(x & 0xFFFFFFFF)
    words.reverse()
    return struct.pack("<%dI" % len(words), *words)


def decode(data, indexes, cdfs, sizes, offsets):
    """-> list of symbols, RansDecoder.set_stream + decode_stream."""
    words = struct.unpack("<%dI" % (len(data) // 4), data)
    pos = 2
    x = words[0] | (words[1] << 32)                          # Rans64DecInit

    def get_bits():
        nonlocal x, pos
        v = x & MAX_BYPASS                                   # Rans64DecGetBits
        x >>= BYPASS_BITS
        if x < L:
            x = (x << 32) | words[pos]
            pos += 1
        return v

    out = []
    mask = (1 << PRECISION) - 1
    for ci in indexes:
        cdf = cdfs[ci]
        max_value = int(sizes[ci]) - 2
        cum = x & mask                                       # Rans64DecGet
        s = 0
        while int(cdf[s + 1]) <= cum:                        # first slot with cdf[s+1] > cum
            s += 1
        assert s <= max_value
        start, rng = int(cdf[s]), int(cdf[s + 1]) - int(cdf[s])
        x = rng * (x >> PRECISION) + (x & mask) - start      # Rans64DecAdvance
        if x < L:
            x = (x << 32) | words[pos]
            pos += 1
        value = s
        if value == max_value:
            val = get_bits()
            n_bypass = val
            while val == MAX_BYPASS:
                val = get_bits()
                n_bypass += val
            raw = 0
            for j in range(n_bypass):
                raw |= get_bits() << (j * BYPASS_BITS)
            value = raw >> 1
            if raw & 1:
                value = -value - 1
            else:
                value += max_value
        out.append(value + int(offsets[ci]))
    return out
