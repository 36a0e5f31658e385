// ffn_f16x3.hip -- the per-pixel tail of a DepthConvBlock as ONE kernel (f16x3 precision mode).
//
// Replaces, for C <= 64 channels (reference: src/models/lssvc_modules.py:15-72, DepthConv.conv2 + identity and
// ConvFFN):
//     o1  = W_pre * t + b_pre + ident                     (optional leading 1x1 conv with its residual)
//     out = o1 + lrelu(W2 * lrelu(W1 * o1 + b1) + b2)     (ConvFFN: C -> hidden -> C, slope 0.1, + o1)
// Unfused, this chain moves 14 C-channel tensor passes through HBM (the 4C-wide hidden tensor is written and read
// back); fused it moves 3 (t, ident, out), and the hidden activations never leave registers.
//
// GEMM chaining without data movement: D[m][n] of v_mfma_f32_16x16x32_f16 leaves lane (i = l & 15, g = l >> 4) with
// rows m = 4g..4g+3 of column n = i, and the B operand of the next GEMM wants 8 consecutive k of column n = i in
// the same lane. Two accumulator fragments (channels 16a + 4g + j and 16b + 4g + j) therefore ARE a B fragment if the
// next weight matrix's K axis is ordered k = 8g + j' -> channel 16*(j' < 4 ? a : b) + 4g + (j' & 3). The host lays
// W1 and W2 out in that order (lssvc_amd/weights.py: layout_ffn_f16x3), so o1 -> hidden -> out stays in registers.
//
// Workgroup = 8 waves sharing the LDS-resident weights (hi/lo fp16 planes of W_pre, W1, W2 + the fp32 biases; up to
// 147 KB for C = 64, hidden = 256, i.e. one workgroup per CU), persistent over flat 32-pixel groups; no barrier
// after the staging. Every product is the 3-term split of the f16x3 mode (lo*hi + hi*lo + hi*hi, fp32 accumulate).
#include "common.h"
#include "conv_f16x3_kernel.h"

namespace lssvc {

struct FfnP {
    V x, pre_in, ident, out;
    const _Float16 *pre_w, *w1, *w2;     // each: [hi plane | lo plane], already in the LDS image order
    const float *pre_b, *b1, *b2;
    float pre_u, u1, u2, slope;
    int hidden, n_pre, n1, n2;           // halfs per plane
    int sa;                              // K-steps (of 32 channels) of the leading conv
};

__device__ __forceinline__ void split8(const float (&v)[8], f16x8 &h, f16x8 &l) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x = fminf(fmaxf(v[j], -65504.f), 65504.f);
        const _Float16 hh = (_Float16)x;
        h[j] = hh;
        l[j] = (_Float16)(x - (float)hh);
    }
}

template <int NA, int RPW>
__device__ __forceinline__ void mfma3(f32x4 (&acc)[NA][RPW], const f16x8 (&ah)[NA], const f16x8 (&al)[NA], const f16x8 (&bh)[RPW],
                                      const f16x8 (&bl)[RPW]) {
#pragma unroll
    for (int f = 0; f < NA; ++f)
#pragma unroll
        for (int r = 0; r < RPW; ++r) acc[f][r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[f], bh[r], acc[f][r], 0, 0, 0);
#pragma unroll
    for (int f = 0; f < NA; ++f)
#pragma unroll
        for (int r = 0; r < RPW; ++r) acc[f][r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[f], bl[r], acc[f][r], 0, 0, 0);
#pragma unroll
    for (int f = 0; f < NA; ++f)
#pragma unroll
        for (int r = 0; r < RPW; ++r) acc[f][r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[f], bh[r], acc[f][r], 0, 0, 0);
}

constexpr int kFfnThreads = 512;

template <int CF, bool PRE>
__global__ __launch_bounds__(kFfnThreads, 1) void ffn_f16x3_kernel(const FfnP p) {
    constexpr int RPW = 2;
    constexpr int S = (CF + 1) / 2;                 // K-steps of the C -> hidden GEMM (two C fragments per step)
    constexpr int SA_MAX = 2;                       // leading conv: at most 64 input channels
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // LDS image: [W1 hi | W1 lo | W2 hi | W2 lo | Wpre hi | Wpre lo] (fp16) then [b1 | b2 | b_pre] (fp32)
    _Float16 *w1h = reinterpret_cast<_Float16 *>(smem);
    _Float16 *w1l = w1h + p.n1;
    _Float16 *w2h = w1l + p.n1;
    _Float16 *w2l = w2h + p.n2;
    _Float16 *wph = w2l + p.n2;
    _Float16 *wpl = wph + p.n_pre;
    float *b1s = reinterpret_cast<float *>(wpl + p.n_pre);
    float *b2s = b1s + p.hidden;
    float *bps = b2s + 16 * CF;

    const int tid = threadIdx.x;
    {
        auto copy = [&](_Float16 *dst, const _Float16 *src, int halfs) {
            for (int i = tid * 8; i < halfs; i += kFfnThreads * 8)
                *reinterpret_cast<f16x8 *>(dst + i) = *reinterpret_cast<const f16x8 *>(src + i);
        };
        copy(w1h, p.w1, 2 * p.n1);
        copy(w2h, p.w2, 2 * p.n2);
        if (PRE) copy(wph, p.pre_w, 2 * p.n_pre);
        for (int i = tid; i < p.hidden; i += kFfnThreads) b1s[i] = p.b1[i];
        for (int i = tid; i < 16 * CF; i += kFfnThreads) {
            b2s[i] = p.b2[i];
            if (PRE) bps[i] = p.pre_b[i];
        }
    }
    __syncthreads();

    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 15;
    const int lg = lane >> 4;
    const int C = p.out.C;
    const long long npix = (long long)p.out.H * p.out.W;
    const long long ngroups = (npix + 16 * RPW - 1) / (16 * RPW);
    const long long wave_id = (long long)blockIdx.x * (kFfnThreads / 64) + wave;
    const long long wave_stride = (long long)gridDim.x * (kFfnThreads / 64);
    const int T = p.hidden >> 5;
    const int a_off = li * 32 + lg * 8;             // this lane's 8 halfs inside a [16][32] fragment image

    // prefetch registers: the next group's inputs in flight while the current one is in the matrix pipe
    float4 nx[CF][RPW];                             // o1 (no leading conv) or ident (leading conv), accumulator layout
    float4 nt[PRE ? SA_MAX : 1][RPW][2];            // leading conv input, B-fragment layout (8 channels per lane)
    auto load_group = [&](long long grp) {
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            long long q = (grp * RPW + r) * 16 + li;
            if (q >= npix) q = 0;
            const V &src = PRE ? p.ident : p.x;
#pragma unroll
            for (int f = 0; f < CF; ++f) nx[f][r] = *reinterpret_cast<const float4 *>(src.p + (size_t)q * src.ld + f * 16 + 4 * lg);
            if (PRE) {
#pragma unroll
                for (int s = 0; s < SA_MAX; ++s) {
                    const int c0 = 32 * s + 8 * lg;
                    const int left = s < p.sa ? p.pre_in.C - c0 : 0;
                    const float *t = p.pre_in.p + (size_t)q * p.pre_in.ld + (left > 0 ? c0 : 0);
                    nt[s][r][0] = *reinterpret_cast<const float4 *>(t);
                    nt[s][r][1] = *reinterpret_cast<const float4 *>(t + (left > 4 ? 4 : 0));
                }
            }
        }
    };
    if (wave_id < ngroups) load_group(wave_id);

    for (long long grp = wave_id; grp < ngroups; grp += wave_stride) {
        long long pix[RPW];
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const long long q = (grp * RPW + r) * 16 + li;
            pix[r] = q < npix ? q : -1;
        }
        // ---- o1 in accumulator layout: o1[f][r][j] = channel 16f + 4g + j of pixel pix[r]
        f32x4 o1[CF][RPW];
        if (PRE) {
            f16x8 th[SA_MAX][RPW], tl[SA_MAX][RPW];
#pragma unroll
            for (int s = 0; s < SA_MAX; ++s) {
                const int left = s < p.sa ? p.pre_in.C - (32 * s + 8 * lg) : 0;
#pragma unroll
                for (int r = 0; r < RPW; ++r) {
                    const float raw[8] = {nt[s][r][0].x, nt[s][r][0].y, nt[s][r][0].z, nt[s][r][0].w,
                                          nt[s][r][1].x, nt[s][r][1].y, nt[s][r][1].z, nt[s][r][1].w};
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = j < left ? raw[j] : 0.f;
                    split8(v, th[s][r], tl[s][r]);
                }
            }
            f32x4 ident[CF][RPW];
#pragma unroll
            for (int f = 0; f < CF; ++f)
#pragma unroll
                for (int r = 0; r < RPW; ++r) {
                    ident[f][r] = f32x4{nx[f][r].x, nx[f][r].y, nx[f][r].z, nx[f][r].w};
                    o1[f][r] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            if (grp + wave_stride < ngroups) load_group(grp + wave_stride);
#pragma unroll
            for (int s = 0; s < SA_MAX; ++s) {
                if (s < p.sa) {
                    f16x8 ah[CF], al[CF];
#pragma unroll
                    for (int f = 0; f < CF; ++f) {
                        const int o = (f * p.sa + s) * 512 + a_off;
                        ah[f] = *reinterpret_cast<const f16x8 *>(wph + o);
                        al[f] = *reinterpret_cast<const f16x8 *>(wpl + o);
                    }
                    mfma3<CF, RPW>(o1, ah, al, th[s], tl[s]);
                }
            }
#pragma unroll
            for (int f = 0; f < CF; ++f) {
                const f32x4 b = *reinterpret_cast<const f32x4 *>(bps + f * 16 + 4 * lg);
#pragma unroll
                for (int r = 0; r < RPW; ++r) o1[f][r] = (o1[f][r] * p.pre_u + b) + ident[f][r];
            }
        } else {
#pragma unroll
            for (int f = 0; f < CF; ++f)
#pragma unroll
                for (int r = 0; r < RPW; ++r) o1[f][r] = f32x4{nx[f][r].x, nx[f][r].y, nx[f][r].z, nx[f][r].w};
            if (grp + wave_stride < ngroups) load_group(grp + wave_stride);
        }

        // ---- B fragments of the C -> hidden GEMM straight from o1 (K order permuted on the host to match)
        f16x8 bh[S][RPW], bl[S][RPW];
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j] = o1[2 * s][r][j];
                    v[4 + j] = (2 * s + 1 < CF) ? o1[(2 * s + 1 < CF) ? 2 * s + 1 : 0][r][j] : 0.f;
                }
                split8(v, bh[s][r], bl[s][r]);
            }

        f32x4 oacc[CF][RPW];
#pragma unroll
        for (int f = 0; f < CF; ++f)
#pragma unroll
            for (int r = 0; r < RPW; ++r) oacc[f][r] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int t = 0; t < T; ++t) {
            // hidden channels 32t .. 32t+31 = two accumulator fragments
            f32x4 hacc[2][RPW];
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int r = 0; r < RPW; ++r) hacc[f][r] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < S; ++s) {
                f16x8 ah[2], al[2];
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    const int o = ((t * 2 + f) * S + s) * 512 + a_off;
                    ah[f] = *reinterpret_cast<const f16x8 *>(w1h + o);
                    al[f] = *reinterpret_cast<const f16x8 *>(w1l + o);
                }
                mfma3<2, RPW>(hacc, ah, al, bh[s], bl[s]);
            }
            f16x8 hh[RPW], hl[RPW];
            {
                const f32x4 b0 = *reinterpret_cast<const f32x4 *>(b1s + (2 * t) * 16 + 4 * lg);
                const f32x4 b1v = *reinterpret_cast<const f32x4 *>(b1s + (2 * t + 1) * 16 + 4 * lg);
#pragma unroll
                for (int r = 0; r < RPW; ++r) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float a = hacc[0][r][j] * p.u1 + b0[j];
                        const float b = hacc[1][r][j] * p.u1 + b1v[j];
                        v[j] = a > 0.f ? a : a * p.slope;
                        v[4 + j] = b > 0.f ? b : b * p.slope;
                    }
                    split8(v, hh[r], hl[r]);
                }
            }
            f16x8 ah[CF], al[CF];
#pragma unroll
            for (int m = 0; m < CF; ++m) {
                const int o = (t * CF + m) * 512 + a_off;
                ah[m] = *reinterpret_cast<const f16x8 *>(w2h + o);
                al[m] = *reinterpret_cast<const f16x8 *>(w2l + o);
            }
            mfma3<CF, RPW>(oacc, ah, al, hh, hl);
        }

        // ---- out = o1 + lrelu(W2 h + b2)
#pragma unroll
        for (int f = 0; f < CF; ++f) {
            if (f * 16 + 4 * lg >= C) continue;
            const f32x4 b = *reinterpret_cast<const f32x4 *>(b2s + f * 16 + 4 * lg);
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                if (pix[r] < 0) continue;
                float4 o;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a = oacc[f][r][j] * p.u2 + b[j];
                    v[j] = o1[f][r][j] + (a > 0.f ? a : a * p.slope);
                }
                o = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4 *>(p.out.p + (size_t)pix[r] * p.out.ld + f * 16 + 4 * lg) = o;
            }
        }
    }
}

template <int CF, bool PRE>
static int launch_ffn(const FfnP &p, size_t lds, hipStream_t st) {
    static const int cus = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) return v;
        return 256;
    }();
    static size_t granted = 0;
    if (lds > granted) {
        LSSVC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(ffn_f16x3_kernel<CF, PRE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        granted = lds;
    }
    const long long npix = (long long)p.out.H * p.out.W;
    const long long ngroups = (npix + 31) / 32;
    long long blocks = (ngroups + kFfnThreads / 64 - 1) / (kFfnThreads / 64);
    if (blocks > cus) blocks = cus;                 // one persistent 8-wave workgroup per CU
    hipLaunchKernelGGL((ffn_f16x3_kernel<CF, PRE>), dim3((unsigned)blocks), dim3(kFfnThreads), lds, st, p);
    return launch_status("ffn_f16x3");
}

}  // namespace lssvc

using namespace lssvc;

extern "C" int64_t lssvc_ffn_f16x3_lds_bytes(int32_t C, int32_t hidden, int32_t pre_cin) {
    const int cf = (C + 15) / 16, s = (cf + 1) / 2, t = hidden / 32;
    const long long n1 = (long long)t * 2 * s * 512, n2 = (long long)t * cf * 512;
    const long long npre = pre_cin > 0 ? (long long)cf * ((pre_cin + 31) / 32) * 512 : 0;
    return 2 * 2 * (n1 + n2 + npre) + 4 * ((long long)hidden + 2 * 16 * cf);
}

extern "C" int lssvc_ffn_f16x3(const lssvc_ffn_desc *d, void *stream) {
    LSSVC_CHECK(d != nullptr, "ffn_f16x3: null descriptor");
    LSSVC_CHECK(view_ok(&d->out) && vec4_ok(&d->out), "ffn_f16x3: bad out view (needs C %% 4 == 0, 16-byte aligned)");
    const int C = d->out.C;
    LSSVC_CHECK(C % 16 == 0 && C >= 32 && C <= 64, "ffn_f16x3: C = %d not in {32, 48, 64}", C);
    LSSVC_CHECK(d->hidden > 0 && d->hidden % 32 == 0, "ffn_f16x3: hidden = %d must be a positive multiple of 32", d->hidden);
    LSSVC_CHECK(d->w1_16 && d->w2_16 && d->b1 && d->b2, "ffn_f16x3: missing FFN weights");
    const bool pre = d->pre_w16 != nullptr;
    FfnP p{};
    p.out = mk(&d->out);
    if (pre) {
        LSSVC_CHECK(view_ok(&d->pre_in) && vec4_ok(&d->pre_in) && same_hw(&d->pre_in, &d->out), "ffn_f16x3: bad pre_in view");
        LSSVC_CHECK(d->pre_in.C % 8 == 0 && d->pre_in.C <= 64, "ffn_f16x3: leading conv takes Cin %% 8 == 0, <= 64 (got %d)", d->pre_in.C);
        LSSVC_CHECK(view_ok(&d->ident) && vec4_ok(&d->ident) && same_shape(&d->ident, &d->out), "ffn_f16x3: bad ident view");
        LSSVC_CHECK(d->pre_bias != nullptr, "ffn_f16x3: leading conv needs a bias vector");
        p.pre_in = mk(&d->pre_in);
        p.ident = mk(&d->ident);
        p.x = mk_null();
        p.sa = (d->pre_in.C + 31) / 32;
    } else {
        LSSVC_CHECK(view_ok(&d->x) && vec4_ok(&d->x) && same_shape(&d->x, &d->out), "ffn_f16x3: bad x view");
        p.x = mk(&d->x);
        p.pre_in = mk_null();
        p.ident = mk_null();
        p.sa = 0;
    }
    const int cf = C / 16, s = (cf + 1) / 2, t = d->hidden / 32;
    p.hidden = d->hidden;
    p.n1 = t * 2 * s * 512;
    p.n2 = t * cf * 512;
    p.n_pre = pre ? cf * p.sa * 512 : 0;
    p.pre_w = reinterpret_cast<const _Float16 *>(d->pre_w16);
    p.w1 = reinterpret_cast<const _Float16 *>(d->w1_16);
    p.w2 = reinterpret_cast<const _Float16 *>(d->w2_16);
    p.pre_b = d->pre_bias;
    p.b1 = d->b1;
    p.b2 = d->b2;
    p.pre_u = d->pre_unscale != 0.f ? d->pre_unscale : 1.f;
    p.u1 = d->w1_unscale != 0.f ? d->w1_unscale : 1.f;
    p.u2 = d->w2_unscale != 0.f ? d->w2_unscale : 1.f;
    p.slope = d->slope;
    const size_t lds = (size_t)lssvc_ffn_f16x3_lds_bytes(C, d->hidden, pre ? d->pre_in.C : 0);
    LSSVC_CHECK(lds <= 160 * 1024, "ffn_f16x3: %zu bytes of weights do not fit the 160 KB LDS (C=%d hidden=%d)", lds, C, d->hidden);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (pre) {
        if (cf == 4) return launch_ffn<4, true>(p, lds, st);
        if (cf == 3) return launch_ffn<3, true>(p, lds, st);
        return launch_ffn<2, true>(p, lds, st);
    }
    if (cf == 4) return launch_ffn<4, false>(p, lds, st);
    if (cf == 3) return launch_ffn<3, false>(p, lds, st);
    return launch_ffn<2, false>(p, lds, st);
}
