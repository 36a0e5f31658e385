// weight_prep.cpp -- checkpoint tensors -> the layouts the HIP kernels consume, on the host, in C++.
//
// One-time work per checkpoint ("load a table of named fp32 tensors"): the reference's nn.Module.load_state_dict
// (src/models/IntraSS.py:190-214, src/models/LSSVC_net.py:141-149) keeps OIHW fp32 tensors; the kernels of this library want
//   conv        OIHW fp32 -> [chunk8][ky][kx][m][8] (every concatenated input segment zero-padded to 8 channels, output channels
//               padded to 16; sub-pixel convs with the output axis permuted to (dy,dx)-major so PixelShuffle is the store pattern)
//   conv f16x3  fp16 hi / lo planes [hi|lo][chunk16][ky][kx][m][16] of w * 2^e (per-layer power-of-two prescale)
//   convT       ConvTranspose2d(k 3, pad 1[, stride 2, output_padding 1]) rewritten as a plain conv
//   GDN         beta / gamma de-reparametrised (gdn.py:31-33, video_net_component.py:86-93) into a 1x1 conv on x^2
//   depthwise   (C,1,3,3) -> [9][C]
//   FFN         the two chained-K LDS images of lssvc_ffn_f16x3 (+ the leading 1x1 conv in natural K order)
//   BitEstimator / EntropyBottleneck -> [rows][C] tables with softplus / tanh applied (video_entropy_models.py:110-129,
//               img_entropy_models.py:483-502)
// The Python front end (lssvc_amd/weights.py: WeightStore) and the engine (plan_runtime.cpp: lssvc_engine_load_checkpoint) both
// call lssvc_prepare_weights, so the two paths hold byte-identical device weights by construction.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "common.h"

namespace {

using lssvc::fail;

struct Src {
    const float *p = nullptr;
    int nd = 0;
    int64_t d[4] = {1, 1, 1, 1};
    int64_t numel() const { return d[0] * d[1] * d[2] * d[3]; }
};

const lssvc_tensor *find(const lssvc_tensor *t, int n, const std::string &name) {
    for (int i = 0; i < n; ++i)
        if (t[i].name && name == t[i].name) return &t[i];
    return nullptr;
}

bool get(const lssvc_tensor *t, int n, const std::string &name, Src &s, bool required = true) {
    const lssvc_tensor *x = find(t, n, name);
    if (!x || !x->data) {
        if (required) fail("prepare_weights: the checkpoint has no tensor '%s'", name.c_str());
        return false;
    }
    if (x->ndim < 0 || x->ndim > 4) {
        fail("prepare_weights: tensor '%s' has %d dimensions", name.c_str(), (int)x->ndim);
        return false;
    }
    s.p = x->data;
    s.nd = x->ndim;
    int64_t total = 1;
    for (int i = 0; i < 4; ++i) {
        s.d[i] = i < x->ndim ? x->shape[i] : 1;
        if (s.d[i] < 1 || s.d[i] > (int64_t(1) << 28) || (total *= s.d[i]) > (int64_t(1) << 32)) {      // (a checkpoint tensor of > 2^32 elements is not one of this model's)
            fail("prepare_weights: tensor '%s' has a bad shape", name.c_str());
            return false;
        }
    }
    return true;
}

// every tensor a prepared layout reads is checked against the element count the layout assumes BEFORE anything is written: a
// checkpoint with other shapes (another architecture, a truncated dump) is an error message, not an out-of-bounds access
bool want_numel(const Src &s, int64_t n, const char *layer, const char *what) {
    if (s.numel() == n) return true;
    fail("prepare_weights(%s): %s has %lld elements, expected %lld", layer, what, (long long)s.numel(), (long long)n);
    return false;
}

inline int64_t pad_to(int64_t n, int64_t m) { return (n + m - 1) / m * m; }

// ---- OIHW (after the optional pixel-shuffle permutation of the output axis) -> [chunk][ky][kx][m][CK], zero-padded -------
// w: (cout, cin, kh, kw). splits: channel counts of the concatenated input segments. Returns the padded K (channels).
struct Oihw {
    std::vector<float> w;      // (cout, cin, kh, kw), already permuted
    int64_t cout, cin, kh, kw;
};

Oihw shuffle_rows(const float *w, int64_t cout, int64_t cin, int64_t kh, int64_t kw, bool pixel_shuffle) {
    Oihw o{std::vector<float>((size_t)(cout * cin * kh * kw)), cout, cin, kh, kw};
    const int64_t row = cin * kh * kw, cps = cout / 4;
    for (int64_t m = 0; m < cout; ++m) {
        // weights.py: w.reshape(cps, 4, ...).permute(1, 0, ...): new row q * cps + c takes old row c * 4 + q
        const int64_t src = pixel_shuffle ? (m % cps) * 4 + m / cps : m;
        std::memcpy(&o.w[(size_t)(m * row)], w + src * row, (size_t)row * sizeof(float));
    }
    return o;
}

std::vector<float> chunked(const Oihw &o, const int32_t *splits, int n_splits, int CK, int64_t m_pad, int64_t &nchunk) {
    int64_t kpad = 0;
    for (int i = 0; i < n_splits; ++i) kpad += pad_to(splits[i], CK);
    nchunk = kpad / CK;
    std::vector<float> out((size_t)(nchunk * o.kh * o.kw * m_pad * CK), 0.f);
    int64_t a = 0, ka = 0;
    for (int i = 0; i < n_splits; ++i) {
        for (int64_t c = 0; c < splits[i]; ++c) {
            const int64_t k = ka + c, chunk = k / CK, j = k % CK;
            for (int64_t m = 0; m < o.cout; ++m)
                for (int64_t y = 0; y < o.kh; ++y)
                    for (int64_t x = 0; x < o.kw; ++x)
                        out[(size_t)((((chunk * o.kh + y) * o.kw + x) * m_pad + m) * CK + j)] = o.w[(size_t)(((m * o.cin + a + c) * o.kh + y) * o.kw + x)];
        }
        a += splits[i];
        ka += pad_to(splits[i], CK);
    }
    return out;
}

// ---- w * 2^e -> fp16 hi, fp16 lo (weights.py: layout_conv_f16x3 / _f16x3_planes) -------------------------------------------
constexpr int kWeightExp = 12;       // max |w * 2^e| in [2^11, 2^12)

float split_planes(const std::vector<float> &w, _Float16 *hi, _Float16 *lo) {
    float wmax = 0.f;
    bool finite = true;
    for (float v : w) {
        const float a = std::fabs(v);
        if (!std::isfinite(a)) finite = false;
        if (a > wmax) wmax = a;
    }
    int e = 0;
    if (wmax != 0.f && finite) {
        int ex;
        (void)std::frexp(wmax, &ex);
        e = kWeightExp - ex;
        e = e < -14 ? -14 : (e > 24 ? 24 : e);
    }
    const float s = std::ldexp(1.0f, e);
    for (size_t i = 0; i < w.size(); ++i) {
        const float v = w[i] * s;               // power of two: exact
        const _Float16 h = (_Float16)v;         // round to nearest even, as torch .half()
        hi[i] = h;
        lo[i] = (_Float16)(v - (float)h);
    }
    return std::ldexp(1.0f, -e);
}

struct Out {
    void **blobs;                // NULL on the size query
    int64_t *bytes;
    int n = 0;
    void *take(int64_t nbytes) {
        bytes[n] = nbytes;
        void *p = blobs ? blobs[n] : nullptr;
        ++n;
        return p;
    }
};

int conv_f32(const Oihw &o, const float *bias, const int32_t *splits, int n_splits, bool bias_shuffled, Out &out, int32_t *dims) {
    const int64_t m_pad = pad_to(o.cout, 16);
    int64_t nchunk = 0;
    {   // sizes first
        int64_t kpad = 0;
        for (int i = 0; i < n_splits; ++i) kpad += pad_to(splits[i], 8);
        nchunk = kpad / 8;
    }
    float *wp = (float *)out.take(nchunk * o.kh * o.kw * m_pad * 8 * (int64_t)sizeof(float));
    float *bp = (float *)out.take(m_pad * (int64_t)sizeof(float));
    dims[0] = (int32_t)o.cout, dims[1] = (int32_t)m_pad, dims[2] = (int32_t)o.kh, dims[3] = (int32_t)o.kw;
    if (!wp) return 0;
    const std::vector<float> c = chunked(o, splits, n_splits, 8, m_pad, nchunk);
    std::memcpy(wp, c.data(), c.size() * sizeof(float));
    for (int64_t m = 0; m < m_pad; ++m) bp[m] = 0.f;
    if (bias) {
        const int64_t cps = o.cout / 4;
        for (int64_t m = 0; m < o.cout; ++m) bp[m] = bias[bias_shuffled ? (m % cps) * 4 + m / cps : m];
    }
    return 0;
}

int conv_f16(const Oihw &o, const int32_t *splits, int n_splits, Out &out, float *unscale) {
    const int64_t m_pad = pad_to(o.cout, 16);
    int64_t kpad = 0;
    for (int i = 0; i < n_splits; ++i) kpad += pad_to(splits[i], 16);
    const int64_t n = (kpad / 16) * o.kh * o.kw * m_pad * 16;
    _Float16 *pl = (_Float16 *)out.take(2 * n * (int64_t)sizeof(_Float16));
    if (!pl) return 0;
    int64_t nchunk;
    const std::vector<float> c = chunked(o, splits, n_splits, 16, m_pad, nchunk);
    *unscale = split_planes(c, pl, pl + n);
    return 0;
}

int sum_splits(const lssvc_prep_spec *s, int64_t cin) {
    int64_t t = 0;
    if (s->n_splits < 1 || s->n_splits > 3) return fail("prepare_weights(%s): %d input segments", s->name, (int)s->n_splits);
    for (int i = 0; i < s->n_splits; ++i) {
        if (s->splits[i] < 1) return fail("prepare_weights(%s): input segment %d has %d channels", s->name, i, (int)s->splits[i]);
        t += s->splits[i];
    }
    if (t != cin) return fail("prepare_weights(%s): input segments do not add up to %lld channels", s->name, (long long)cin);
    return 0;
}

// ---- ConvTranspose2d as a conv (weights.py: conv_t_as_conv) ----------------------------------------------------------------
int conv_transpose(const lssvc_tensor *t, int n, const lssvc_prep_spec *s, Out &out, int32_t *dims) {
    Src w, b;
    if (!get(t, n, std::string(s->name) + ".weight", w) || !get(t, n, std::string(s->name) + ".bias", b)) return 1;
    const int64_t cin = w.d[0], cout = w.d[1];
    if (w.nd != 4 || w.d[2] != 3 || w.d[3] != 3) return fail("prepare_weights(%s): ConvTranspose2d weight must be (Cin, Cout, 3, 3)", s->name);
    if (!want_numel(b, cout, s->name, "bias")) return 1;
    if (s->flag != 1 && s->flag != 2) return fail("prepare_weights(%s): ConvTranspose2d stride %d", s->name, (int)s->flag);
    const int32_t one_split[1] = {(int32_t)cin};
    if (s->flag == 1) {             // stride 1: flipped 3x3, (Cout, Cin)
        Oihw o{std::vector<float>((size_t)(cout * cin * 9)), cout, cin, 3, 3};
        for (int64_t m = 0; m < cout; ++m)
            for (int64_t c = 0; c < cin; ++c)
                for (int y = 0; y < 3; ++y)
                    for (int x = 0; x < 3; ++x) o.w[(size_t)(((m * cin + c) * 3 + y) * 3 + x)] = w.p[((c * cout + m) * 3 + (2 - y)) * 3 + (2 - x)];
        if (int rc = conv_f32(o, b.p, one_split, 1, false, out, dims)) return rc;
        dims[4] = 1, dims[5] = 0;
        return 0;
    }
    // stride 2: out[2i+a, 2j+b] = sum_{dy,dx in {0,1}} in[i+dy, j+dx] * w[:, :, ky(a,dy), kx(b,dx)]; rows (q = 2a+b, co)-major = the
    // kernel's pixel-shuffle order
    auto tap = [](int phase, int delta) { return phase == 0 ? (delta == 0 ? 1 : -1) : (delta == 0 ? 2 : 0); };
    Oihw o{std::vector<float>((size_t)(4 * cout * cin * 4), 0.f), 4 * cout, cin, 2, 2};
    for (int a = 0; a < 2; ++a)
        for (int bb = 0; bb < 2; ++bb)
            for (int dy = 0; dy < 2; ++dy)
                for (int dx = 0; dx < 2; ++dx) {
                    const int ky = tap(a, dy), kx = tap(bb, dx);
                    if (ky < 0 || kx < 0) continue;
                    for (int64_t co = 0; co < cout; ++co)
                        for (int64_t c = 0; c < cin; ++c)
                            o.w[(size_t)(((((a * 2 + bb) * cout + co) * cin + c) * 2 + dy) * 2 + dx)] = w.p[((c * cout + co) * 3 + ky) * 3 + kx];
                }
    std::vector<float> b4((size_t)(4 * cout));
    for (int q = 0; q < 4; ++q)
        for (int64_t co = 0; co < cout; ++co) b4[(size_t)(q * cout + co)] = b.p[co];
    if (int rc = conv_f32(o, b4.data(), one_split, 1, false, out, dims)) return rc;
    dims[4] = 0, dims[5] = 1;
    return 0;
}

// ---- GDN -------------------------------------------------------------------------------------------------------------------
int gdn(const lssvc_tensor *t, int n, const lssvc_prep_spec *s, Out &out, float *scalars, int32_t *dims) {
    const std::string nm = s->name;
    Src beta, gamma;
    if (!get(t, n, nm + ".beta", beta) || !get(t, n, nm + ".gamma", gamma)) return 1;
    const int64_t c = gamma.d[0];
    if (gamma.nd != 2 || gamma.d[1] != c || beta.numel() != c) return fail("prepare_weights(%s): GDN wants beta (C) and gamma (C, C)", s->name);
    std::vector<float> be((size_t)c), ga((size_t)(c * c));
    if (s->flag == 0) {             // 'intra': gdn.py + others.py reparametrisation buffers from the checkpoint
        Src bb, bp, gb, gp;
        if (!get(t, n, nm + ".beta_reparam.lower_bound.bound", bb) || !get(t, n, nm + ".beta_reparam.pedestal", bp) ||
            !get(t, n, nm + ".gamma_reparam.lower_bound.bound", gb) || !get(t, n, nm + ".gamma_reparam.pedestal", gp))
            return 1;
        if (!want_numel(bb, 1, s->name, "beta_reparam.lower_bound.bound") || !want_numel(bp, 1, s->name, "beta_reparam.pedestal") ||
            !want_numel(gb, 1, s->name, "gamma_reparam.lower_bound.bound") || !want_numel(gp, 1, s->name, "gamma_reparam.pedestal"))
            return 1;
        for (int64_t i = 0; i < c; ++i) {
            const float m = std::fmax(beta.p[i], bb.p[0]);
            be[(size_t)i] = m * m - bp.p[0];
        }
        for (int64_t i = 0; i < c * c; ++i) {
            const float m = std::fmax(gamma.p[i], gb.p[0]);
            ga[(size_t)i] = m * m - gp.p[0];
        }
    } else {                        // 'inter': video_net_component.py constants
        const double off = std::ldexp(1.0, -18), ped = off * off;
        const float beta_bound = (float)std::sqrt(1e-6 + ped), gamma_bound = (float)off, pedf = (float)ped;
        for (int64_t i = 0; i < c; ++i) {
            const float m = std::fmax(beta.p[i], 1.0f * beta_bound);
            be[(size_t)i] = m * m - pedf;
        }
        for (int64_t i = 0; i < c * c; ++i) {
            const float m = std::fmax(gamma.p[i], 1.0f * gamma_bound);
            ga[(size_t)i] = m * m - pedf;
        }
    }
    Oihw o{ga, c, c, 1, 1};
    const int32_t one_split[1] = {(int32_t)c};
    if (int rc = conv_f32(o, be.data(), one_split, 1, false, out, dims)) return rc;
    return conv_f16(o, one_split, 1, out, &scalars[0]);
}

// ---- FFN: chained-K images (weights.py: layout_ffn_f16x3, layout_pw_natural_f16x3; csrc/ffn_f16x3.hip) ---------------------
// K position (pair p, k = 8 g + j) of a B operand assembled from two accumulator fragments of the previous GEMM is channel
// (2 p + (j >> 2)) * 16 + 4 g + (j & 3).
inline int64_t chained(int64_t p, int k) {
    const int g = k / 8, j = k % 8;
    return (2 * p + (j >> 2)) * 16 + 4 * g + (j & 3);
}

int ffn(const lssvc_tensor *t, int n, const lssvc_prep_spec *s, Out &out, float *scalars, int32_t *dims) {
    const std::string nm = s->name;
    Src w1, w2, b1, b2;
    if (!get(t, n, nm + ".conv.0.weight", w1) || !get(t, n, nm + ".conv.2.weight", w2) || !get(t, n, nm + ".conv.0.bias", b1) ||
        !get(t, n, nm + ".conv.2.bias", b2))
        return 1;
    const int64_t hidden = w1.d[0], c = w1.d[1];
    if (w1.nd != 4 || w2.nd != 4 || w1.d[2] != 1 || w1.d[3] != 1 || w2.d[2] != 1 || w2.d[3] != 1) return fail("prepare_weights(%s): ConvFFN weights must be (Cout, Cin, 1, 1)", s->name);
    if (!want_numel(b1, hidden, s->name, "conv.0.bias") || !want_numel(b2, c, s->name, "conv.2.bias")) return 1;
    if (c % 16 || hidden % 32 || w2.d[0] != c || w2.d[1] != hidden) return fail("prepare_weights(%s): ConvFFN shapes (%lld, %lld)", s->name, (long long)hidden, (long long)c);
    const int64_t cf = c / 16, tt = hidden / 32, ss = (cf + 1) / 2;
    const bool pre = s->name2[0] != 0;
    Src wp, bp;
    if (pre && (!get(t, n, std::string(s->name2) + ".weight", wp) || !get(t, n, std::string(s->name2) + ".bias", bp))) return 1;
    if (pre && (wp.nd != 4 || wp.d[2] != 1 || wp.d[3] != 1 || wp.d[0] != c || !want_numel(bp, wp.d[0], s->name2, "bias")))
        return fail("prepare_weights(%s): the leading conv must be a 1x1 conv with %lld output channels and a bias", s->name2, (long long)c);
    const int64_t n1 = tt * 2 * ss * 16 * 32, n2 = tt * cf * 16 * 32;
    _Float16 *o1 = (_Float16 *)out.take(2 * n1 * 2);
    _Float16 *o2 = (_Float16 *)out.take(2 * n2 * 2);
    float *ob1 = (float *)out.take(hidden * 4);
    float *ob2 = (float *)out.take(c * 4);
    int64_t pre_cin = 0, sp = 0, npre = 0;
    _Float16 *op = nullptr;
    float *obp = nullptr;
    if (pre) {
        pre_cin = wp.d[1];
        if (wp.d[0] % 16) return fail("prepare_weights(%s): the leading conv needs Cout %% 16 == 0", s->name2);
        sp = (pre_cin + 31) / 32;
        npre = (wp.d[0] / 16) * sp * 16 * 32;
        op = (_Float16 *)out.take(2 * npre * 2);
        obp = (float *)out.take(wp.d[0] * 4);
    }
    dims[0] = (int32_t)hidden, dims[1] = (int32_t)c, dims[2] = (int32_t)pre_cin;
    if (!o1) return 0;
    {   // W1 [t][f][s][i][k]: hidden channel (t * 2 + f) * 16 + i, input channel chained(s, k) (zero past C)
        std::vector<float> a((size_t)n1);
        size_t q = 0;
        for (int64_t ti = 0; ti < tt; ++ti)
            for (int f = 0; f < 2; ++f)
                for (int64_t si = 0; si < ss; ++si)
                    for (int i = 0; i < 16; ++i)
                        for (int k = 0; k < 32; ++k) {
                            const int64_t ch = chained(si, k), h = (ti * 2 + f) * 16 + i;
                            a[q++] = ch < c ? w1.p[h * c + ch] : 0.f;
                        }
        scalars[0] = split_planes(a, o1, o1 + n1);
    }
    {   // W2 [t][m][i][k]: output channel m * 16 + i, hidden channel chained(t, k)
        std::vector<float> a((size_t)n2);
        size_t q = 0;
        for (int64_t ti = 0; ti < tt; ++ti)
            for (int64_t m = 0; m < cf; ++m)
                for (int i = 0; i < 16; ++i)
                    for (int k = 0; k < 32; ++k) a[q++] = w2.p[(m * 16 + i) * hidden + chained(ti, k)];
        scalars[1] = split_planes(a, o2, o2 + n2);
    }
    std::memcpy(ob1, b1.p, (size_t)hidden * 4);
    std::memcpy(ob2, b2.p, (size_t)c * 4);
    if (pre) {  // [m][s][i][k] in natural K order
        std::vector<float> a((size_t)npre);
        size_t q = 0;
        for (int64_t m = 0; m < wp.d[0] / 16; ++m)
            for (int64_t si = 0; si < sp; ++si)
                for (int i = 0; i < 16; ++i)
                    for (int k = 0; k < 32; ++k) {
                        const int64_t ch = si * 32 + k;
                        a[q++] = ch < pre_cin ? wp.p[(m * 16 + i) * pre_cin + ch] : 0.f;
                    }
        scalars[2] = split_planes(a, op, op + npre);
        std::memcpy(obp, bp.p, (size_t)wp.d[0] * 4);
    }
    return 0;
}

// ---- parameter tables ----------------------------------------------------------------------------------------------------
// softplus / tanh in double, rounded once to fp32 (F.softplus: x > 20 -> x)
inline float softplus(float x) { return x > 20.f ? x : (float)std::log1p(std::exp((double)x)); }
inline float tanh32(float x) { return (float)std::tanh((double)x); }

int bit_estimator(const lssvc_tensor *t, int n, const lssvc_prep_spec *s, Out &out, int32_t *dims) {
    const std::string nm = s->name;
    Src h[4], b[4], a[3];
    for (int i = 0; i < 4; ++i) {
        const std::string f = nm + ".f" + std::to_string(i + 1);
        if (!get(t, n, f + ".h", h[i]) || !get(t, n, f + ".b", b[i])) return 1;
        if (i < 3 && !get(t, n, f + ".a", a[i])) return 1;
    }
    const int64_t c = h[0].numel();
    for (int i = 0; i < 4; ++i)
        if (!want_numel(h[i], c, s->name, "f*.h") || !want_numel(b[i], c, s->name, "f*.b") || (i < 3 && !want_numel(a[i], c, s->name, "f*.a"))) return 1;
    float *o = (float *)out.take(11 * c * 4);
    dims[0] = (int32_t)c;
    if (!o) return 0;
    int r = 0;
    for (int i = 0; i < 3; ++i) {
        for (int64_t k = 0; k < c; ++k) o[r * c + k] = softplus(h[i].p[k]);
        ++r;
        std::memcpy(o + r * c, b[i].p, (size_t)c * 4);
        ++r;
        for (int64_t k = 0; k < c; ++k) o[r * c + k] = tanh32(a[i].p[k]);
        ++r;
    }
    for (int64_t k = 0; k < c; ++k) o[r * c + k] = softplus(h[3].p[k]);
    ++r;
    std::memcpy(o + r * c, b[3].p, (size_t)c * 4);
    return 0;
}

int entropy_bottleneck(const lssvc_tensor *t, int n, const lssvc_prep_spec *s, Out &out, int32_t *dims) {
    const std::string nm = s->name;
    Src m[5], b[5], f[4], q;
    for (int i = 0; i < 5; ++i)
        if (!get(t, n, nm + "._matrices." + std::to_string(i), m[i]) || !get(t, n, nm + "._biases." + std::to_string(i), b[i])) return 1;
    for (int i = 0; i < 4; ++i)
        if (!get(t, n, nm + "._factors." + std::to_string(i), f[i])) return 1;
    if (!get(t, n, nm + ".quantiles", q)) return 1;
    const int64_t c = m[0].d[0];
    {   // filters (1, 3, 3, 3, 3, 1) (img_entropy_models.py:400-431): matrices (C, f[i+1], f[i]), biases (C, f[i+1], 1), factors (C, f[i+1], 1), quantiles (C, 1, 3);
        // checked before the 59-row blob is taken, so that a checkpoint with other filter sizes cannot overflow it
        static const int64_t fl[6] = {1, 3, 3, 3, 3, 1};
        for (int i = 0; i < 5; ++i) {
            if (m[i].nd != 3 || m[i].d[0] != c || m[i].d[1] != fl[i + 1] || m[i].d[2] != fl[i]) return fail("prepare_weights(%s): _matrices.%d must be (C, %lld, %lld)", s->name, i, (long long)fl[i + 1], (long long)fl[i]);
            if (b[i].nd != 3 || b[i].d[0] != c || b[i].d[1] != fl[i + 1] || b[i].d[2] != 1) return fail("prepare_weights(%s): _biases.%d must be (C, %lld, 1)", s->name, i, (long long)fl[i + 1]);
            if (i < 4 && (f[i].nd != 3 || f[i].d[0] != c || f[i].d[1] != fl[i + 1] || f[i].d[2] != 1)) return fail("prepare_weights(%s): _factors.%d must be (C, %lld, 1)", s->name, i, (long long)fl[i + 1]);
        }
        if (q.nd != 3 || q.d[0] != c || q.d[1] != 1 || q.d[2] != 3) return fail("prepare_weights(%s): quantiles must be (C, 1, 3)", s->name);
    }
    float *o = (float *)out.take(59 * c * 4);
    dims[0] = (int32_t)c;
    if (!o) return 0;
    int r = 0;
    for (int i = 0; i < 5; ++i)                              // softplus(matrices): (C, f_out, f_in), rows [j][k]
        for (int64_t j = 0; j < m[i].d[1]; ++j)
            for (int64_t k = 0; k < m[i].d[2]; ++k, ++r)
                for (int64_t ch = 0; ch < c; ++ch) o[r * c + ch] = softplus(m[i].p[(ch * m[i].d[1] + j) * m[i].d[2] + k]);
    for (int i = 0; i < 5; ++i)                              // biases (C, f_out, 1)
        for (int64_t j = 0; j < b[i].d[1]; ++j, ++r)
            for (int64_t ch = 0; ch < c; ++ch) o[r * c + ch] = b[i].p[ch * b[i].d[1] + j];
    for (int i = 0; i < 4; ++i)                              // tanh(factors) (C, f_out, 1)
        for (int64_t j = 0; j < f[i].d[1]; ++j, ++r)
            for (int64_t ch = 0; ch < c; ++ch) o[r * c + ch] = tanh32(f[i].p[ch * f[i].d[1] + j]);
    for (int64_t ch = 0; ch < c; ++ch) o[r * c + ch] = q.p[ch * 3 + 1];      // median: quantiles (C, 1, 3)[:, 0, 1]
    ++r;
    if (r != 59) return fail("prepare_weights(%s): EntropyBottleneck table has %d rows, expected 59", s->name, r);
    return 0;
}

}  // namespace

extern "C" int lssvc_prepare_weights(const lssvc_tensor *ckpt, int32_t n_tensors, const lssvc_prep_spec *spec, int32_t *n_blobs,
                                     int64_t blob_bytes[LSSVC_PREP_MAX_BLOBS], float scalars[4], int32_t dims[8], void *const *blobs) {
    LSSVC_CHECK(ckpt && spec && n_blobs && blob_bytes && scalars && dims && n_tensors >= 0, "prepare_weights: bad arguments");
    LSSVC_CHECK(memchr(spec->name, 0, sizeof(spec->name)) && memchr(spec->name2, 0, sizeof(spec->name2)), "prepare_weights: unterminated layer name");
    for (int i = 0; i < 4; ++i) scalars[i] = 1.0f;
    for (int i = 0; i < 8; ++i) dims[i] = 0;
    Out out{const_cast<void **>(blobs), blob_bytes, 0};
    const std::string nm = spec->name;
    int rc = 0;
    switch (spec->kind) {
    case LSSVC_PREP_CONV:
    case LSSVC_PREP_CONV_F16X3: {
        Src w, b;
        if (!get(ckpt, n_tensors, nm + ".weight", w)) return 1;
        if (w.nd != 4) return fail("prepare_weights(%s): conv weight must have 4 dimensions", spec->name);
        if (sum_splits(spec, w.d[1])) return 1;
        const bool ps = spec->flag != 0;
        if (ps && w.d[0] % 4) return fail("prepare_weights(%s): a sub-pixel conv needs Cout %% 4 == 0", spec->name);
        const Oihw o = shuffle_rows(w.p, w.d[0], w.d[1], w.d[2], w.d[3], ps);
        if (spec->kind == LSSVC_PREP_CONV) {
            const bool has_b = get(ckpt, n_tensors, nm + ".bias", b, false);
            if (has_b && !want_numel(b, w.d[0], spec->name, "bias")) return 1;
            rc = conv_f32(o, has_b ? b.p : nullptr, spec->splits, spec->n_splits, ps, out, dims);
        } else {
            rc = conv_f16(o, spec->splits, spec->n_splits, out, &scalars[0]);
        }
        break;
    }
    case LSSVC_PREP_CONVT: rc = conv_transpose(ckpt, n_tensors, spec, out, dims); break;
    case LSSVC_PREP_DWCONV: {
        Src w, b;
        if (!get(ckpt, n_tensors, nm + ".weight", w) || !get(ckpt, n_tensors, nm + ".bias", b)) return 1;
        const int64_t c = w.d[0];
        if (w.nd != 4 || w.d[1] != 1 || w.d[2] != 3 || w.d[3] != 3) return fail("prepare_weights(%s): depthwise weight must be (C, 1, 3, 3)", spec->name);
        if (!want_numel(b, c, spec->name, "bias")) return 1;
        float *o = (float *)out.take(9 * c * 4);
        float *ob = (float *)out.take(c * 4);
        dims[0] = (int32_t)c;
        if (o) {
            for (int64_t ch = 0; ch < c; ++ch)
                for (int k = 0; k < 9; ++k) o[k * c + ch] = w.p[ch * 9 + k];
            std::memcpy(ob, b.p, (size_t)c * 4);
        }
        break;
    }
    case LSSVC_PREP_GDN: rc = gdn(ckpt, n_tensors, spec, out, scalars, dims); break;
    case LSSVC_PREP_VECTOR: {
        Src v;
        if (!get(ckpt, n_tensors, nm, v)) return 1;
        float *o = (float *)out.take(v.numel() * 4);
        dims[0] = (int32_t)v.numel();
        if (o) std::memcpy(o, v.p, (size_t)v.numel() * 4);
        break;
    }
    case LSSVC_PREP_BIT_ESTIMATOR: rc = bit_estimator(ckpt, n_tensors, spec, out, dims); break;
    case LSSVC_PREP_ENTROPY_BOTTLENECK: rc = entropy_bottleneck(ckpt, n_tensors, spec, out, dims); break;
    case LSSVC_PREP_FFN_F16X3: rc = ffn(ckpt, n_tensors, spec, out, scalars, dims); break;
    default: return fail("prepare_weights: unknown kind %d", spec->kind);
    }
    *n_blobs = out.n;
    return rc;
}
