// rans_host.cpp -- the sequential entropy coder of write_stream=1, kept on the host (BASELINE north_star)
// behind the C ABI of include/lssvc_hip.h.
//
// Wire format = the reference's: one rANS64 stream per (frame, layer[, latent]) with 16-bit probabilities,
// symbols pushed in coding order and entropy-coded in reverse at flush, out-of-table symbols escaped through
// the table's last slot followed by 4-bit raw digits (reference: src/cpp/rans/rans_interface.cpp:85-244,
// rANS64 core = rygorous/ryg_rans rans64.h, public domain). What differs from the reference is the host
// interface: symbols and indexes arrive as flat int32 or int16 planes (no Python lists; the product path hands over
// int16 planes in one pinned staging buffer filled by an asynchronous D2H copy, lssvc_amd/entropy_coder.py), tables
// are passed by pointer, the pending symbols are packed in 32 bits, and the decoder finds
// the symbol by binary search and checks every read against the end of the stream.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "common.h"

namespace {

constexpr uint64_t kL = 1ull << 31;        // lower bound of the normalised state interval
constexpr uint32_t kProbBits = 16;
constexpr uint32_t kRawBits = 4;           // bypass digit width
constexpr uint32_t kRawMax = (1u << kRawBits) - 1;

struct Encoder {
    // pending symbols: bits 0-15 start, bits 16-31 (range - 1), raw digits flagged in a parallel bitset
    std::vector<uint32_t> syms;
    std::vector<uint8_t> is_raw;
    std::vector<uint32_t> words;
    const uint8_t *out = nullptr;
    int64_t out_bytes = 0;
    void push(uint32_t start, uint32_t range, bool raw) {
        syms.push_back(start | ((range - 1) << 16));
        is_raw.push_back(raw ? 1 : 0);
    }
};

struct Decoder {
    std::vector<uint32_t> words;
    size_t pos = 0;
    uint64_t x = 0;
    bool overrun = false;
    uint32_t next_word() {
        if (pos >= words.size()) {
            overrun = true;
            return 0;
        }
        return words[pos++];
    }
    uint32_t raw_digit() {
        const uint32_t v = (uint32_t)(x & kRawMax);
        x >>= kRawBits;
        if (x < kL) x = (x << 32) | next_word();
        return v;
    }
};

bool table_ok(const lssvc_cdf_table *t) {
    return t && t->cdfs && t->sizes && t->offsets && t->n_cdfs > 0 && t->stride >= 2;
}

}  // namespace

using namespace lssvc;

extern "C" void *lssvc_rans_encoder_new(void) { return new Encoder(); }
extern "C" void lssvc_rans_encoder_free(void *h) { delete static_cast<Encoder *>(h); }
extern "C" void lssvc_rans_encoder_reset(void *h) {
    Encoder *e = static_cast<Encoder *>(h);
    e->syms.clear();
    e->is_raw.clear();
}

template <typename PlaneT>
static int encode_with_indexes_impl(void *h, const PlaneT *symbols, const PlaneT *indexes, int64_t n, const lssvc_cdf_table *t) {
    LSSVC_CHECK(h && symbols && indexes && n >= 0 && table_ok(t), "rans_encode_with_indexes: bad arguments");
    Encoder *e = static_cast<Encoder *>(h);
    e->syms.reserve(e->syms.size() + (size_t)n + 16);
    e->is_raw.reserve(e->is_raw.size() + (size_t)n + 16);
    // an error part-way through a plane leaves the pending list as it was at entry (no half-appended plane)
    struct Rollback {
        Encoder *e;
        size_t n0;
        bool armed;
        ~Rollback() {
            if (armed) {
                e->syms.resize(n0);
                e->is_raw.resize(n0);
            }
        }
    } rollback{e, e->syms.size(), true};
    for (int64_t i = 0; i < n; ++i) {
        const int32_t ci = indexes[i];
        LSSVC_CHECK(ci >= 0 && ci < t->n_cdfs, "rans_encode_with_indexes: index %d out of range [0,%d) at %lld", ci, t->n_cdfs,
                    (long long)i);
        const int32_t *cdf = t->cdfs + (size_t)ci * t->stride;
        const int32_t escape = t->sizes[ci] - 2;              // last table slot = "escape, raw digits follow"
        LSSVC_CHECK(escape >= 0 && t->sizes[ci] <= t->stride, "rans_encode_with_indexes: bad cdf size %d", t->sizes[ci]);
        int32_t v = symbols[i] - t->offsets[ci];
        uint32_t raw = 0;
        if (v < 0) {                                          // below the table: odd codes
            raw = (uint32_t)(-2 * (int64_t)v - 1);
            v = escape;
        } else if (v >= escape) {                             // above (or at) the escape slot: even codes
            raw = (uint32_t)(2 * ((int64_t)v - escape));
            v = escape;
        }
        // a zero-width or out-of-range slot would wrap to a huge frequency and write a corrupt stream silently
        // (the reference divides by zero in Rans64EncPut here): refuse it
        LSSVC_CHECK(cdf[v] >= 0 && cdf[v + 1] > cdf[v] && cdf[v + 1] <= (1 << kProbBits),
                    "rans_encode_with_indexes: cdf row %d is not strictly increasing within [0, 2^%d] at slot %d (%d, %d)", ci,
                    (int)kProbBits, v, cdf[v], cdf[v + 1]);
        e->push((uint32_t)cdf[v], (uint32_t)(cdf[v + 1] - cdf[v]), false);
        if (v == escape) {
            // (64-bit shifts: a 32-bit value with its top digit set needs 8 digits, and `raw >> 32` on a uint32_t is
            // undefined -- on x86 it is `raw >> 0`, i.e. an endless loop. The reference has that hazard for
            // |symbol - offset| >= 2^27, rans_interface.cpp:121-123; here such symbols code and decode correctly.)
            uint32_t digits = 0;
            while (((uint64_t)raw >> (digits * kRawBits)) != 0) ++digits;
            uint32_t count = digits;                          // digit count, unary-ish in base 15
            while (count >= kRawMax) {
                e->push(kRawMax, 1, true);
                count -= kRawMax;
            }
            e->push(count, 1, true);
            for (uint32_t j = 0; j < digits; ++j) e->push((raw >> (j * kRawBits)) & kRawMax, 1, true);
        }
    }
    rollback.armed = false;
    return 0;
}

extern "C" int lssvc_rans_encode_with_indexes(void *h, const int32_t *symbols, const int32_t *indexes, int64_t n,
                                              const lssvc_cdf_table *t) {
    return encode_with_indexes_impl<int32_t>(h, symbols, indexes, n, t);
}
extern "C" int lssvc_rans_encode_with_indexes_i16(void *h, const int16_t *symbols, const int16_t *indexes, int64_t n,
                                                  const lssvc_cdf_table *t) {
    return encode_with_indexes_impl<int16_t>(h, symbols, indexes, n, t);
}

// Entropy-codes the pending symbols (last pushed first, so the decoder pops them in pushing order) and
// returns the byte count; lssvc_rans_encoder_bytes() then points at the stream until the next call.
extern "C" int64_t lssvc_rans_encoder_flush(void *h) {
    Encoder *e = static_cast<Encoder *>(h);
    const size_t n = e->syms.size();
    e->words.assign(n + 2, 0);
    uint32_t *ptr = e->words.data() + e->words.size();
    uint64_t x = kL;
    for (size_t i = n; i-- > 0;) {
        const uint32_t start = e->syms[i] & 0xFFFFu;
        if (!e->is_raw[i]) {
            const uint64_t freq = (e->syms[i] >> 16) + 1;
            if (x >= ((kL >> kProbBits) << 32) * freq) {
                *--ptr = (uint32_t)x;
                x >>= 32;
            }
            x = ((x / freq) << kProbBits) + (x % freq) + start;
        } else {                                              // raw 4-bit digit = uniform symbol of width 2^12/2^16
            if (x >= ((kL >> 16) << 32) * (uint64_t)(1u << (16 - kRawBits))) {
                *--ptr = (uint32_t)x;
                x >>= 32;
            }
            x = (x << kRawBits) | start;
        }
    }
    ptr -= 2;
    ptr[0] = (uint32_t)x;
    ptr[1] = (uint32_t)(x >> 32);
    e->out = reinterpret_cast<const uint8_t *>(ptr);
    e->out_bytes = (int64_t)((e->words.data() + e->words.size()) - ptr) * 4;
    e->syms.clear();
    e->is_raw.clear();
    return e->out_bytes;
}
extern "C" const uint8_t *lssvc_rans_encoder_bytes(void *h) { return static_cast<Encoder *>(h)->out; }

extern "C" void *lssvc_rans_decoder_new(void) { return new Decoder(); }
extern "C" void lssvc_rans_decoder_free(void *h) { delete static_cast<Decoder *>(h); }

extern "C" int lssvc_rans_decoder_set_stream(void *h, const uint8_t *bytes, int64_t n) {
    LSSVC_CHECK(h && bytes && n >= 8 && n % 4 == 0, "rans_decoder_set_stream: stream of %lld bytes is not a rANS64 stream", (long long)n);
    Decoder *d = static_cast<Decoder *>(h);
    d->words.resize((size_t)n / 4);
    memcpy(d->words.data(), bytes, (size_t)n);
    d->x = (uint64_t)d->words[0] | ((uint64_t)d->words[1] << 32);
    d->pos = 2;
    d->overrun = false;
    return 0;
}

template <typename PlaneT>
static int decode_stream_impl(void *h, const PlaneT *indexes, int64_t n, const lssvc_cdf_table *t, PlaneT *out) {
    LSSVC_CHECK(h && indexes && out && n >= 0 && table_ok(t), "rans_decode_stream: bad arguments");
    Decoder *d = static_cast<Decoder *>(h);
    LSSVC_CHECK(d->words.size() >= 2, "rans_decode_stream: no stream set");
    for (int64_t i = 0; i < n; ++i) {
        const int32_t ci = indexes[i];
        LSSVC_CHECK(ci >= 0 && ci < t->n_cdfs, "rans_decode_stream: index %d out of range [0,%d) at %lld", ci, t->n_cdfs, (long long)i);
        const int32_t *cdf = t->cdfs + (size_t)ci * t->stride;
        const int32_t size = t->sizes[ci];
        const int32_t escape = size - 2;
        const uint32_t cum = (uint32_t)(d->x & ((1u << kProbBits) - 1));
        int32_t lo = 0, hi = size - 1;                        // largest s with cdf[s] <= cum  (cdf strictly increasing)
        while (hi - lo > 1) {
            const int32_t mid = (lo + hi) >> 1;
            if ((uint32_t)cdf[mid] <= cum) lo = mid; else hi = mid;
        }
        const uint32_t start = (uint32_t)cdf[lo], freq = (uint32_t)(cdf[lo + 1] - cdf[lo]);
        d->x = (uint64_t)freq * (d->x >> kProbBits) + cum - start;
        if (d->x < kL) d->x = (d->x << 32) | d->next_word();
        int32_t v = lo;
        if (v == escape) {
            uint32_t digit = d->raw_digit();
            uint32_t digits = digit;
            while (digit == kRawMax) {
                digit = d->raw_digit();
                digits += digit;
            }
            LSSVC_CHECK(digits <= 8, "rans_decode_stream: corrupt escape (%u digits)", digits);
            uint32_t raw = 0;
            for (uint32_t j = 0; j < digits; ++j) raw |= d->raw_digit() << (j * kRawBits);
            v = (int32_t)(raw >> 1);
            v = (raw & 1) ? -v - 1 : v + escape;
        }
        const int32_t s = v + t->offsets[ci];
        if constexpr (sizeof(PlaneT) == 2)
            LSSVC_CHECK(s >= -32768 && s <= 32767, "rans_decode_stream: symbol %d at %lld does not fit the 16-bit plane", s, (long long)i);
        out[i] = (PlaneT)s;
    }
    LSSVC_CHECK(!d->overrun, "rans_decode_stream: read past the end of the stream");
    return 0;
}

extern "C" int lssvc_rans_decode_stream(void *h, const int32_t *indexes, int64_t n, const lssvc_cdf_table *t, int32_t *out) {
    return decode_stream_impl<int32_t>(h, indexes, n, t, out);
}
extern "C" int lssvc_rans_decode_stream_i16(void *h, const int16_t *indexes, int64_t n, const lssvc_cdf_table *t, int16_t *out) {
    return decode_stream_impl<int16_t>(h, indexes, n, t, out);
}

// Round a pmf to `precision`-bit frequencies whose CDF is strictly increasing (every symbol stays
// codable): scale, renormalise to 2^precision, then for every empty bin take one count from the
// currently least-frequent bin that can spare it (reference: src/cpp/ops/ops.cpp:24-82).
extern "C" int lssvc_pmf_to_quantized_cdf(const float *pmf, int32_t n, int32_t precision, uint32_t *cdf) {
    LSSVC_CHECK(pmf && cdf && n > 0 && precision > 0 && precision <= 16, "pmf_to_quantized_cdf: bad arguments");
    std::vector<uint32_t> c((size_t)n + 1, 0);
    uint32_t total = 0;
    for (int32_t i = 0; i < n; ++i) {
        c[i + 1] = (uint32_t)(std::round(pmf[i] * (float)(1 << precision)) + 0.5);
        total += c[i + 1];
    }
    LSSVC_CHECK(total > 0, "pmf_to_quantized_cdf: empty pmf");
    uint32_t run = 0;
    for (int32_t i = 0; i <= n; ++i) {
        run += (uint32_t)(((1ull << precision) * c[i]) / total);
        c[i] = run;
    }
    c[n] = 1u << precision;
    for (int32_t i = 0; i < n; ++i) {
        if (c[i] != c[i + 1]) continue;
        uint32_t best = ~0u;
        int32_t donor = -1;
        for (int32_t j = 0; j < n; ++j) {
            const uint32_t f = c[j + 1] - c[j];
            if (f > 1 && f < best) {
                best = f;
                donor = j;
            }
        }
        LSSVC_CHECK(donor >= 0, "pmf_to_quantized_cdf: cannot make the cdf strictly increasing");
        if (donor < i) for (int32_t j = donor + 1; j <= i; ++j) c[j]--;
        else for (int32_t j = i + 1; j <= donor; ++j) c[j]++;
    }
    memcpy(cdf, c.data(), ((size_t)n + 1) * sizeof(uint32_t));
    return 0;
}
