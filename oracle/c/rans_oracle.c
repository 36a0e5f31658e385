/* ORACLE (test infrastructure only) -- plain-C restatement of the reference's host entropy coder.
 *
 *  - rANS64 core: third-party rygorous/ryg_rans `rans64.h` @ c9d162d (public domain, fetched by the
 *    reference's CMake, NOT vendored under /root/reference: 3rdparty/ryg_rans/CMakeLists.txt.in:8-9).
 *    Restated here from its published algorithm: 64-bit state, lower bound L = 2^31, 32-bit
 *    renormalisation words written backwards, scale_bits-bit probabilities.
 *    Reference call sites: rans_interface.cpp:54,73,149,159,167,181,205,213.
 *  - wrapper: BufferedRansEncoder::{encode_with_indexes,flush} and RansDecoder::{set_stream,
 *    decode_stream} with the 4-bit bypass escape (rans_interface.cpp:85-244).
 *  - pmf_to_quantized_cdf (ops.cpp:24-82).
 *
 * Parity status: pmf_to_quantized_cdf is pinned against the reference's own ops.cpp (built by
 * oracle/Makefile into oracle/_ref, golden vectors in tests/golden/cdf_vectors.json).
 * The rANS byte stream is "parity unpinned": rans_interface.cpp cannot be built here (rans64.h
 * absent) and the reference holds no byte-level test vectors; it is pinned only by round trips.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define RANS64_L (1ull << 31)
#define PRECISION 16
#define BYPASS_PRECISION 4
#define MAX_BYPASS_VAL ((1 << BYPASS_PRECISION) - 1)

/* ---- rans64.h (ryg_rans) ---------------------------------------------------------------------- */
static void enc_put(uint64_t *r, uint32_t **pptr, uint32_t start, uint32_t freq, uint32_t scale_bits) {
    uint64_t x = *r;
    uint64_t x_max = ((RANS64_L >> scale_bits) << 32) * freq;
    if (x >= x_max) {
        *pptr -= 1;
        **pptr = (uint32_t)x;
        x >>= 32;
    }
    *r = ((x / freq) << scale_bits) + (x % freq) + start;
}
static void enc_flush(uint64_t *r, uint32_t **pptr) {
    uint64_t x = *r;
    *pptr -= 2;
    (*pptr)[0] = (uint32_t)(x >> 0);
    (*pptr)[1] = (uint32_t)(x >> 32);
}
static void dec_init(uint64_t *r, uint32_t **pptr) {
    uint64_t x = (uint64_t)((*pptr)[0]) << 0;
    x |= (uint64_t)((*pptr)[1]) << 32;
    *pptr += 2;
    *r = x;
}
static uint32_t dec_get(uint64_t *r, uint32_t scale_bits) { return (uint32_t)(*r & ((1u << scale_bits) - 1)); }
static void dec_advance(uint64_t *r, uint32_t **pptr, uint32_t start, uint32_t freq, uint32_t scale_bits) {
    uint64_t mask = (1ull << scale_bits) - 1;
    uint64_t x = *r;
    x = freq * (x >> scale_bits) + (x & mask) - start;
    if (x < RANS64_L) {
        x = (x << 32) | **pptr;
        *pptr += 1;
    }
    *r = x;
}
/* ---- rans_interface.cpp:38-79 ------------------------------------------------------------------ */
static void enc_put_bits(uint64_t *r, uint32_t **pptr, uint32_t val, uint32_t nbits) {
    uint64_t x = *r;
    uint32_t freq = 1u << (16 - nbits);
    uint64_t x_max = ((RANS64_L >> 16) << 32) * freq;
    if (x >= x_max) {
        *pptr -= 1;
        **pptr = (uint32_t)x;
        x >>= 32;
    }
    *r = (x << nbits) | val;
}
static uint32_t dec_get_bits(uint64_t *r, uint32_t **pptr, uint32_t nbits) {
    uint64_t x = *r;
    uint32_t val = (uint32_t)(x & ((1u << nbits) - 1));
    x >>= nbits;
    if (x < RANS64_L) {
        x = (x << 32) | **pptr;
        *pptr += 1;
    }
    *r = x;
    return val;
}

typedef struct { uint16_t start, range; uint8_t bypass; } sym_t;
typedef struct { sym_t *s; size_t n, cap; } encoder_t;

static void push(encoder_t *e, uint16_t start, uint16_t range, uint8_t bypass) {
    if (e->n == e->cap) {
        e->cap = e->cap ? e->cap * 2 : 1024;
        e->s = (sym_t *)realloc(e->s, e->cap * sizeof(sym_t));
    }
    e->s[e->n].start = start; e->s[e->n].range = range; e->s[e->n].bypass = bypass;
    e->n++;
}

void *oracle_encoder_new(void) { return calloc(1, sizeof(encoder_t)); }
void oracle_encoder_free(void *h) { encoder_t *e = (encoder_t *)h; free(e->s); free(e); }
void oracle_encoder_reset(void *h) { ((encoder_t *)h)->n = 0; }

/* rans_interface.cpp:85-145 */
void oracle_encode_with_indexes(void *h, const int32_t *symbols, const int32_t *indexes, int64_t n, const int32_t *cdfs,
                                int32_t cdf_stride, const int32_t *cdf_sizes, const int32_t *offsets) {
    encoder_t *e = (encoder_t *)h;
    for (int64_t i = 0; i < n; ++i) {
        const int32_t ci = indexes[i];
        const int32_t *cdf = cdfs + (size_t)ci * cdf_stride;
        const int32_t max_value = cdf_sizes[ci] - 2;
        int32_t value = symbols[i] - offsets[ci];
        uint32_t raw_val = 0;
        if (value < 0) { raw_val = (uint32_t)(-2 * value - 1); value = max_value; }
        else if (value >= max_value) { raw_val = (uint32_t)(2 * (value - max_value)); value = max_value; }
        push(e, (uint16_t)cdf[value], (uint16_t)(cdf[value + 1] - cdf[value]), 0);
        if (value == max_value) {
            int32_t n_bypass = 0;
            /* 64-bit shift: the reference's 32-bit `raw_val >> 32` (rans_interface.cpp:121-123) is undefined for raw
             * values with the top digit set (|symbol - offset| >= 2^27) and loops forever on x86; outside that range
             * the two forms are identical */
            while (((uint64_t)raw_val >> (n_bypass * BYPASS_PRECISION)) != 0) ++n_bypass;
            int32_t val = n_bypass;
            while (val >= MAX_BYPASS_VAL) { push(e, MAX_BYPASS_VAL, MAX_BYPASS_VAL + 1, 1); val -= MAX_BYPASS_VAL; }
            push(e, (uint16_t)val, (uint16_t)(val + 1), 1);
            for (int32_t j = 0; j < n_bypass; ++j) {
                const int32_t v1 = (raw_val >> (j * BYPASS_PRECISION)) & MAX_BYPASS_VAL;
                push(e, (uint16_t)v1, (uint16_t)(v1 + 1), 1);
            }
        }
    }
}

/* rans_interface.cpp:147-174. Returns the byte count; *out is malloc'ed (caller frees with oracle_free). */
int64_t oracle_encoder_flush(void *h, uint8_t **out) {
    encoder_t *e = (encoder_t *)h;
    uint64_t rans = RANS64_L;
    size_t words = e->n + 2;          /* the reference sizes the buffer _syms.size(); +2 keeps the flush words in range */
    uint32_t *buf = (uint32_t *)malloc(words * sizeof(uint32_t));
    uint32_t *ptr = buf + words;
    while (e->n) {
        const sym_t s = e->s[--e->n];
        if (!s.bypass) enc_put(&rans, &ptr, s.start, s.range, PRECISION);
        else enc_put_bits(&rans, &ptr, s.start, BYPASS_PRECISION);
    }
    enc_flush(&rans, &ptr);
    const int64_t nbytes = (int64_t)((buf + words) - ptr) * 4;
    *out = (uint8_t *)malloc((size_t)nbytes);
    memcpy(*out, ptr, (size_t)nbytes);
    free(buf);
    return nbytes;
}
void oracle_free(void *p) { free(p); }

typedef struct { uint64_t rans; uint32_t *buf, *ptr; } decoder_t;
void *oracle_decoder_new(void) { return calloc(1, sizeof(decoder_t)); }
void oracle_decoder_free(void *h) { decoder_t *d = (decoder_t *)h; free(d->buf); free(d); }
/* rans_interface.cpp:176-182 */
void oracle_decoder_set_stream(void *h, const uint8_t *bytes, int64_t n) {
    decoder_t *d = (decoder_t *)h;
    free(d->buf);
    d->buf = (uint32_t *)malloc((size_t)n + 8);
    memcpy(d->buf, bytes, (size_t)n);
    d->ptr = d->buf;
    dec_init(&d->rans, &d->ptr);
}
/* rans_interface.cpp:184-244 */
void oracle_decode_stream(void *h, const int32_t *indexes, int64_t n, const int32_t *cdfs, int32_t cdf_stride,
                          const int32_t *cdf_sizes, const int32_t *offsets, int32_t *out) {
    decoder_t *d = (decoder_t *)h;
    for (int64_t i = 0; i < n; ++i) {
        const int32_t ci = indexes[i];
        const int32_t *cdf = cdfs + (size_t)ci * cdf_stride;
        const int32_t max_value = cdf_sizes[ci] - 2;
        const uint32_t cum = dec_get(&d->rans, PRECISION);
        int32_t s = 0;
        while (s < cdf_sizes[ci] && !((uint32_t)cdf[s] > cum)) ++s;   /* std::find_if(first v > cum) */
        s -= 1;
        dec_advance(&d->rans, &d->ptr, (uint32_t)cdf[s], (uint32_t)(cdf[s + 1] - cdf[s]), PRECISION);
        int32_t value = s;
        if (value == max_value) {
            int32_t val = (int32_t)dec_get_bits(&d->rans, &d->ptr, BYPASS_PRECISION);
            int32_t n_bypass = val;
            while (val == MAX_BYPASS_VAL) { val = (int32_t)dec_get_bits(&d->rans, &d->ptr, BYPASS_PRECISION); n_bypass += val; }
            int32_t raw_val = 0;
            for (int j = 0; j < n_bypass; ++j) {
                val = (int32_t)dec_get_bits(&d->rans, &d->ptr, BYPASS_PRECISION);
                raw_val |= val << (j * BYPASS_PRECISION);
            }
            value = raw_val >> 1;
            if (raw_val & 1) value = -value - 1; else value += max_value;
        }
        out[i] = value + offsets[ci];
    }
}

/* ops.cpp:24-82. cdf_out has n + 1 entries. */
void oracle_pmf_to_quantized_cdf(const float *pmf, int32_t n, int32_t precision, uint32_t *cdf) {
    cdf[0] = 0;
    for (int i = 0; i < n; ++i) cdf[i + 1] = (uint32_t)(roundf(pmf[i] * (float)(1 << precision)) + 0.5);
    uint32_t total = 0;
    for (int i = 0; i <= n; ++i) total += cdf[i];
    for (int i = 0; i <= n; ++i) cdf[i] = (uint32_t)(((1ull << precision) * cdf[i]) / total);
    for (int i = 1; i <= n; ++i) cdf[i] += cdf[i - 1];
    cdf[n] = 1u << precision;
    for (int i = 0; i < n; ++i) {
        if (cdf[i] == cdf[i + 1]) {
            uint32_t best_freq = ~0u;
            int best_steal = -1;
            for (int j = 0; j < n; ++j) {
                uint32_t freq = cdf[j + 1] - cdf[j];
                if (freq > 1 && freq < best_freq) { best_freq = freq; best_steal = j; }
            }
            if (best_steal < i) { for (int j = best_steal + 1; j <= i; ++j) cdf[j]--; }
            else { for (int j = i + 1; j <= best_steal; ++j) cdf[j]++; }
        }
    }
}
