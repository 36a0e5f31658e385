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
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "conv_f16x3_kernel.h"

namespace lssvc {

struct FfnP {
    V x, pre_in, ident, out, skip;     // skip (optional): added to the block's result, e.g. the resamplers' outer skip connection
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

// lrelu(acc * u + b) for one accumulator fragment, written on 4-wide vectors so it compiles to packed fp32 ops; the
// LeakyReLU is max(y, s*y) (exact for 0 <= s <= 1, which the host guarantees for this kernel)
__device__ __forceinline__ f32x4 act4(const f32x4 acc, float u, const f32x4 b, float slope) {
    const f32x4 y = acc * u + b;
    const f32x4 n = y * slope;
    return f32x4{fmaxf(y[0], n[0]), fmaxf(y[1], n[1]), fmaxf(y[2], n[2]), fmaxf(y[3], n[3])};
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
    // 32-bit pixel / group indices (the host refuses images of 2^31 pixels or more): this instantiation sits at the VGPR cap
    const int npix = p.out.H * p.out.W;
    const int ngroups = (npix + 16 * RPW - 1) / (16 * RPW);
    const int wave_id = (int)blockIdx.x * (kFfnThreads / 64) + wave;
    const int wave_stride = (int)gridDim.x * (kFfnThreads / 64);
    const int T = p.hidden >> 5;
    const int a_off = li * 32 + lg * 8;             // this lane's 8 halfs inside a [16][32] fragment image

    // prefetch registers: the next group's inputs in flight while the current one is in the matrix pipe
    float4 nx[CF][RPW];                             // o1 (no leading conv) or ident (leading conv), accumulator layout
    float4 nt[PRE ? SA_MAX : 1][RPW][2];            // leading conv input, B-fragment layout (8 channels per lane)
    // what = 1: the accumulator-layout operand (o1 / ident), 2: the leading conv's input, 3: both
    auto load_group = [&](int grp, int what) {
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            int q = (grp * RPW + r) * 16 + li;
            if (q >= npix) q = 0;
            const V &src = PRE ? p.ident : p.x;
            if (what & 1) {
#pragma unroll
                for (int f = 0; f < CF; ++f) nx[f][r] = *reinterpret_cast<const float4 *>(src.p + (size_t)q * src.ld + f * 16 + 4 * lg);
            }
            if (PRE && (what & 2)) {
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
    // With a leading conv only ITS input is prefetched a group ahead; `ident` is requested at the head of the group it belongs
    // to and first needed after the leading conv's MFMAs. (Both a group ahead = 64 registers of loads in flight across the
    // hidden loop: the CF = 4 instantiation then spills, and a spill reload's s_waitcnt vmcnt(0) -- loads return in order --
    // waits for the whole prefetch at once, which is worse than not prefetching.)
    constexpr int AHEAD = PRE ? 2 : 3;
    if (wave_id < ngroups) load_group(wave_id, AHEAD);

    for (int grp = wave_id; grp < ngroups; grp += wave_stride) {
        int pix[RPW];
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int q = (grp * RPW + r) * 16 + li;
            pix[r] = q < npix ? q : -1;
        }
        if (PRE) load_group(grp, 1);
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
#pragma unroll
            for (int f = 0; f < CF; ++f)
#pragma unroll
                for (int r = 0; r < RPW; ++r) o1[f][r] = f32x4{0.f, 0.f, 0.f, 0.f};
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
                for (int r = 0; r < RPW; ++r) o1[f][r] = (o1[f][r] * p.pre_u + b) + f32x4{nx[f][r].x, nx[f][r].y, nx[f][r].z, nx[f][r].w};   // ident: first use of this group's load
            }
        } else {
#pragma unroll
            for (int f = 0; f < CF; ++f)
#pragma unroll
                for (int r = 0; r < RPW; ++r) o1[f][r] = f32x4{nx[f][r].x, nx[f][r].y, nx[f][r].z, nx[f][r].w};
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

        // The next group's inputs are requested HERE, with the whole hidden loop ahead to hide them. Requested earlier (as soon
        // as their registers were free, before the leading conv) they were waited for at once: this instantiation spills a few
        // registers around that stage, a spill reload is a vector-memory load, loads return in order, so the reload's
        // s_waitcnt vmcnt(0) also waited for the sixteen prefetch loads issued just before it.
        // (unconditional -- past the end it re-reads group 0 -- and fenced: as a conditional block the compiler moved it above the
        // arithmetic that consumes this group's older loads, whose waits then had to assume the worst and drained it as well)
        __builtin_amdgcn_sched_barrier(0);
        load_group(grp + wave_stride < ngroups ? grp + wave_stride : 0, AHEAD);
        __builtin_amdgcn_sched_barrier(0);
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
            // W2's fragments are requested BEFORE the activation / split arithmetic (which does not depend on them): their LDS
            // round trip then hides behind ~110 VALU instructions instead of stalling the MFMAs that follow
            f16x8 ah[CF], al[CF];
#pragma unroll
            for (int m = 0; m < CF; ++m) {
                const int o = (t * CF + m) * 512 + a_off;
                ah[m] = *reinterpret_cast<const f16x8 *>(w2h + o);
                al[m] = *reinterpret_cast<const f16x8 *>(w2l + o);
            }
            __builtin_amdgcn_sched_barrier(0);
            f16x8 hh[RPW], hl[RPW];
            {
                const f32x4 b0 = *reinterpret_cast<const f32x4 *>(b1s + (2 * t) * 16 + 4 * lg);
                const f32x4 b1v = *reinterpret_cast<const f32x4 *>(b1s + (2 * t + 1) * 16 + 4 * lg);
#pragma unroll
                for (int r = 0; r < RPW; ++r) {
                    const f32x4 ya = act4(hacc[0][r], p.u1, b0, p.slope), yb = act4(hacc[1][r], p.u1, b1v, p.slope);
                    const float v[8] = {ya[0], ya[1], ya[2], ya[3], yb[0], yb[1], yb[2], yb[3]};
                    split8(v, hh[r], hl[r]);
                }
            }
            mfma3<CF, RPW>(oacc, ah, al, hh, hl);
        }

        // ---- out = o1 + lrelu(W2 h + b2) (+ skip)
        // Loads and stores share the vmcnt counter: a wait for a load issued AFTER a store also waits for that store to be
        // acknowledged. Written as one loop (load skip, compute, store per fragment) every store of a group waited for
        // the one before it -- a full memory round trip each, eight per group -- whether or not there was a skip operand at
        // all (the wait sits at the join of the two paths). Hence two straight-line variants, chosen wave-uniformly: no skip
        // operand = no load and no wait between the stores; with one, all its loads are issued before the first store.
        auto finish = [&](auto has_skip) {
            constexpr bool SK = decltype(has_skip)::value;
            float4 sk[SK ? CF : 1][SK ? RPW : 1];
            if constexpr (SK) {
#pragma unroll
                for (int f = 0; f < CF; ++f)
#pragma unroll
                    for (int r = 0; r < RPW; ++r) {
                        const bool ok = f * 16 + 4 * lg < C && pix[r] >= 0;
                        sk[f][r] = *reinterpret_cast<const float4 *>(p.skip.p + (ok ? (size_t)pix[r] * p.skip.ld + f * 16 + 4 * lg : (size_t)0));
                    }
            }
#pragma unroll
            for (int f = 0; f < CF; ++f) {
                const f32x4 b = *reinterpret_cast<const f32x4 *>(b2s + f * 16 + 4 * lg);
#pragma unroll
                for (int r = 0; r < RPW; ++r) {
                    float v[4];
                    float skv[4] = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (SK) skv[0] = sk[f][r].x, skv[1] = sk[f][r].y, skv[2] = sk[f][r].z, skv[3] = sk[f][r].w;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float a = oacc[f][r][j] * p.u2 + b[j];
                        v[j] = (o1[f][r][j] + (a > 0.f ? a : a * p.slope)) + skv[j];
                    }
                    if (f * 16 + 4 * lg < C && pix[r] >= 0)
                        *reinterpret_cast<float4 *>(p.out.p + (size_t)pix[r] * p.out.ld + f * 16 + 4 * lg) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        };
        if (p.skip.p) finish(std::true_type{});
        else finish(std::false_type{});
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Streamed-weights variant for wider blocks (C up to 128, hidden up to 1024: 2*C*hidden fp16 hi/lo weights are up to
// 1 MB, far beyond LDS). Same arithmetic and register chaining; what changes is where the FFN weights live: the hidden
// dimension is walked in slices of 32 channels, and slice t+1 of W1/W2 (4S + 2CF KiB, already in fragment order) is
// moved global -> LDS by LDS-DMA into the other half of a two-slot ring while slice t is in the matrix pipe; one
// workgroup barrier per slice. Each wave keeps ONE 16-pixel group (o1, the output accumulators and the C -> hidden B
// fragments: 96 registers at C = 128) for a whole pass over the hidden dimension, so the weights are re-streamed from
// L2 once per 128 pixels of a workgroup (4 KB per pixel at C = 128, hidden = 512: about twice the activation bytes).
// The leading conv's weights (<= 64 KB) stay resident.
template <int CF, bool PRE>
__global__ __launch_bounds__(kFfnThreads, 1) void ffn_stream_f16x3_kernel(const FfnP p) {
    constexpr int S = (CF + 1) / 2;
    constexpr int SA_MAX = 4;                        // leading conv: at most 128 input channels
    constexpr int W1_HALFS = 2 * S * 512, W2_HALFS = CF * 512;          // per plane, per 32-channel hidden slice
    constexpr int SLICE_HALFS = 2 * W1_HALFS + 2 * W2_HALFS;            // [W1 hi | W1 lo | W2 hi | W2 lo]
    constexpr int NI = SLICE_HALFS / 512;                               // 1 KiB DMA instructions per slice
    constexpr int NDMA = (NI + kFfnThreads / 64 - 1) / (kFfnThreads / 64);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16 *ring = reinterpret_cast<_Float16 *>(smem);                // 2 slots
    _Float16 *wph = ring + 2 * SLICE_HALFS;
    _Float16 *wpl = wph + p.n_pre;
    float *b1s = reinterpret_cast<float *>(wpl + p.n_pre);
    float *b2s = b1s + p.hidden;
    float *bps = b2s + 16 * CF;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 15;
    const int lg = lane >> 4;
    const int a_off = li * 32 + lg * 8;
    const int C = p.out.C;
    const int T = p.hidden >> 5;
    {
        if (PRE)
            for (int i = tid * 8; i < 2 * p.n_pre; i += kFfnThreads * 8)
                *reinterpret_cast<f16x8 *>(wph + i) = *reinterpret_cast<const f16x8 *>(p.pre_w + i);
        for (int i = tid; i < p.hidden; i += kFfnThreads) b1s[i] = p.b1[i];
        for (int i = tid; i < 16 * CF; i += kFfnThreads) {
            b2s[i] = p.b2[i];
            if (PRE) bps[i] = p.pre_b[i];
        }
    }
    // one hidden slice -> ring slot: the four pieces are contiguous runs of the host blobs ([hi plane | lo plane])
    auto issue_slice = [&](int t, int slot) {
        unsigned char *dst = reinterpret_cast<unsigned char *>(ring + slot * SLICE_HALFS);
#pragma unroll
        for (int q = 0; q < NDMA; ++q) {
            int j = wave + (kFfnThreads / 64) * q;          // wave-uniform 1 KiB piece of the slice image
            if (j >= NI) j = NI - 1;
            const int h0 = j * 512 + lane * 8;              // first half of this lane's 16 bytes, in slice-image halfs
            const _Float16 *src;
            if (h0 < W1_HALFS) src = p.w1 + (size_t)t * W1_HALFS + h0;
            else if (h0 < 2 * W1_HALFS) src = p.w1 + p.n1 + (size_t)t * W1_HALFS + (h0 - W1_HALFS);
            else if (h0 < 2 * W1_HALFS + W2_HALFS) src = p.w2 + (size_t)t * W2_HALFS + (h0 - 2 * W1_HALFS);
            else src = p.w2 + p.n2 + (size_t)t * W2_HALFS + (h0 - 2 * W1_HALFS - W2_HALFS);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(dst + j * 1024), 16, 0, 0);
        }
    };

    const long long npix = (long long)p.out.H * p.out.W;
    const long long npass = (npix + 127) / 128;             // 8 waves x 16 pixels
    const long long my_passes = blockIdx.x < npass ? (npass - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    if (my_passes == 0) return;
    const long long total = my_passes * T;
    issue_slice(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's LDS-DMA pieces have landed (explicit: see dma_fence note)
    __syncthreads();                                         // resident weights + slice 0 are in LDS

    long long g = 0;
    for (long long pass = blockIdx.x; pass < npass; pass += gridDim.x) {
        long long q = pass * 128 + wave * 16 + li;
        const bool live = q < npix;
        if (!live) q = 0;
        // ---- o1 in accumulator layout
        f32x4 o1[CF];
        if (PRE) {
#pragma unroll
            for (int f = 0; f < CF; ++f) o1[f] = f32x4{0.f, 0.f, 0.f, 0.f};
            // all K steps' loads first, unconditionally (steps past p.sa re-read step 0 and are not used): inside the run-time
            // `if (s < p.sa)` the compiler waited vmcnt(0) after every step's loads, four to eight serial round trips per pass
            float4 pr0[SA_MAX], pr1[SA_MAX];
#pragma unroll
            for (int s = 0; s < SA_MAX; ++s) {
                const int c0 = 32 * (s < p.sa ? s : 0) + 8 * lg;
                const int left = p.pre_in.C - c0;
                const float *tp = p.pre_in.p + (size_t)q * p.pre_in.ld + (left > 0 ? c0 : 0);
                pr0[s] = *reinterpret_cast<const float4 *>(tp);
                pr1[s] = *reinterpret_cast<const float4 *>(tp + (left > 4 ? 4 : 0));
            }
#pragma unroll
            for (int s = 0; s < SA_MAX; ++s) {
                if (s < p.sa) {
                    const int c0 = 32 * s + 8 * lg;
                    const int left = p.pre_in.C - c0;
                    const float4 r0 = pr0[s], r1 = pr1[s];
                    const float raw[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = j < left ? raw[j] : 0.f;
                    f16x8 th[1], tl[1];
                    split8(v, th[0], tl[0]);
                    f16x8 ah[CF], al[CF];
#pragma unroll
                    for (int f = 0; f < CF; ++f) {
                        const int o = (f * p.sa + s) * 512 + a_off;
                        ah[f] = *reinterpret_cast<const f16x8 *>(wph + o);
                        al[f] = *reinterpret_cast<const f16x8 *>(wpl + o);
                    }
                    f32x4 acc1[CF][1];
#pragma unroll
                    for (int f = 0; f < CF; ++f) acc1[f][0] = o1[f];
                    mfma3<CF, 1>(acc1, ah, al, th, tl);
#pragma unroll
                    for (int f = 0; f < CF; ++f) o1[f] = acc1[f][0];
                }
            }
#pragma unroll
            for (int f = 0; f < CF; ++f) {
                const float4 idv = *reinterpret_cast<const float4 *>(p.ident.p + (size_t)q * p.ident.ld + f * 16 + 4 * lg);
                const f32x4 b = *reinterpret_cast<const f32x4 *>(bps + f * 16 + 4 * lg);
                o1[f] = (o1[f] * p.pre_u + b) + f32x4{idv.x, idv.y, idv.z, idv.w};
            }
        } else {
#pragma unroll
            for (int f = 0; f < CF; ++f) {
                const float4 xv = *reinterpret_cast<const float4 *>(p.x.p + (size_t)q * p.x.ld + f * 16 + 4 * lg);
                o1[f] = f32x4{xv.x, xv.y, xv.z, xv.w};
            }
        }
        f16x8 bh[S][1], bl[S][1];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = o1[2 * s][j];
                v[4 + j] = (2 * s + 1 < CF) ? o1[(2 * s + 1 < CF) ? 2 * s + 1 : 0][j] : 0.f;
            }
            split8(v, bh[s][0], bl[s][0]);
        }
        f32x4 oacc[CF][1];
#pragma unroll
        for (int f = 0; f < CF; ++f) oacc[f][0] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int t = 0; t < T; ++t) {
            const int slot = (int)(g & 1);
            if (g + 1 < total) issue_slice(t + 1 < T ? t + 1 : 0, slot ^ 1);      // that slot was released by the last barrier
            const _Float16 *w1h = ring + slot * SLICE_HALFS;
            const _Float16 *w1l = w1h + W1_HALFS;
            const _Float16 *w2h = w1l + W1_HALFS;
            const _Float16 *w2l = w2h + W2_HALFS;
            f32x4 hacc[2][1];
            hacc[0][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            hacc[1][0] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < S; ++s) {
                f16x8 ah[2], al[2];
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    const int o = (f * S + s) * 512 + a_off;
                    ah[f] = *reinterpret_cast<const f16x8 *>(w1h + o);
                    al[f] = *reinterpret_cast<const f16x8 *>(w1l + o);
                }
                mfma3<2, 1>(hacc, ah, al, bh[s], bl[s]);
            }
            f16x8 hh[1], hl[1];
            {
                const f32x4 b0 = *reinterpret_cast<const f32x4 *>(b1s + (2 * t) * 16 + 4 * lg);
                const f32x4 b1v = *reinterpret_cast<const f32x4 *>(b1s + (2 * t + 1) * 16 + 4 * lg);
                const f32x4 ya = act4(hacc[0][0], p.u1, b0, p.slope), yb = act4(hacc[1][0], p.u1, b1v, p.slope);
                const float v[8] = {ya[0], ya[1], ya[2], ya[3], yb[0], yb[1], yb[2], yb[3]};
                split8(v, hh[0], hl[0]);
            }
            f16x8 ah[CF], al[CF];
#pragma unroll
            for (int m = 0; m < CF; ++m) {
                const int o = m * 512 + a_off;
                ah[m] = *reinterpret_cast<const f16x8 *>(w2h + o);
                al[m] = *reinterpret_cast<const f16x8 *>(w2l + o);
            }
            mfma3<CF, 1>(oacc, ah, al, hh, hl);
            // LDS-DMA data is ordered for other waves' ds_reads only by the ISSUING wave's vmcnt wait followed by a
            // barrier. hipcc (ROCm 7.2) happens to emit that wait before s_barrier on its own; spelled out so a
            // toolchain change cannot turn it into an LDS race.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                 // slice g consumed by everyone; slice g+1 has landed
            ++g;
        }
        // (two straight-line variants, as in ffn_f16x3_kernel: no wait between the stores)
        auto finish = [&](auto has_skip) {
            constexpr bool SK = decltype(has_skip)::value;
            float4 sk[SK ? CF : 1];
            if constexpr (SK) {
#pragma unroll
                for (int f = 0; f < CF; ++f) {
                    const bool ok = f * 16 + 4 * lg < C && live;
                    sk[f] = *reinterpret_cast<const float4 *>(p.skip.p + (ok ? (size_t)q * p.skip.ld + f * 16 + 4 * lg : (size_t)0));
                }
            }
#pragma unroll
            for (int f = 0; f < CF; ++f) {
                const f32x4 b = *reinterpret_cast<const f32x4 *>(b2s + f * 16 + 4 * lg);
                float v[4];
                float skv[4] = {0.f, 0.f, 0.f, 0.f};
                if constexpr (SK) skv[0] = sk[f].x, skv[1] = sk[f].y, skv[2] = sk[f].z, skv[3] = sk[f].w;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a = oacc[f][0][j] * p.u2 + b[j];
                    v[j] = (o1[f][j] + (a > 0.f ? a : a * p.slope)) + skv[j];
                }
                if (f * 16 + 4 * lg < C && live)
                    *reinterpret_cast<float4 *>(p.out.p + (size_t)q * p.out.ld + f * 16 + 4 * lg) = make_float4(v[0], v[1], v[2], v[3]);
            }
        };
        if (p.skip.p) finish(std::true_type{});
        else finish(std::false_type{});
    }
}

template <int CF, bool PRE>
static int launch_ffn_stream(const FfnP &p, size_t lds, hipStream_t st) {
    const int cus = device_cus();
    static LdsGrant grant;
    if (grant.ensure(reinterpret_cast<const void *>(ffn_stream_f16x3_kernel<CF, PRE>), lds)) return 1;
    const long long npix = (long long)p.out.H * p.out.W;
    long long blocks = (npix + 127) / 128;
    if (blocks > cus) blocks = cus;
    hipLaunchKernelGGL((ffn_stream_f16x3_kernel<CF, PRE>), dim3((unsigned)blocks), dim3(kFfnThreads), lds, st, p);
    return launch_status("ffn_stream_f16x3");
}

template <int CF, bool PRE>
static int launch_ffn(const FfnP &p, size_t lds, hipStream_t st) {
    const int cus = device_cus();
    static LdsGrant grant;
    if (grant.ensure(reinterpret_cast<const void *>(ffn_f16x3_kernel<CF, PRE>), lds)) return 1;
    const long long npix = (long long)p.out.H * p.out.W;
    const long long ngroups = (npix + 31) / 32;
    long long blocks = (ngroups + kFfnThreads / 64 - 1) / (kFfnThreads / 64);
    if (blocks > cus) blocks = cus;                 // one persistent 8-wave workgroup per CU
    hipLaunchKernelGGL((ffn_f16x3_kernel<CF, PRE>), dim3((unsigned)blocks), dim3(kFfnThreads), lds, st, p);
    return launch_status("ffn_f16x3");
}

}  // namespace lssvc

using namespace lssvc;

static long long ffn_resident_lds(int C, int hidden, int pre_cin) {
    const int cf = (C + 15) / 16, s = (cf + 1) / 2, t = hidden / 32;
    const long long n1 = (long long)t * 2 * s * 512, n2 = (long long)t * cf * 512;
    const long long npre = pre_cin > 0 ? (long long)cf * ((pre_cin + 31) / 32) * 512 : 0;
    return 2 * 2 * (n1 + n2 + npre) + 4 * ((long long)hidden + 2 * 16 * cf);
}
static long long ffn_stream_lds(int C, int hidden, int pre_cin) {
    const int cf = (C + 15) / 16, s = (cf + 1) / 2;
    const long long slice = 2 * (2 * s * 512) + 2 * (cf * 512);                      // halfs per ring slot
    const long long npre = pre_cin > 0 ? (long long)cf * ((pre_cin + 31) / 32) * 512 : 0;
    return 2 * (2 * slice + 2 * npre) + 4 * ((long long)hidden + 2 * 16 * cf);
}

/* LDS the fused kernel needs for this shape: the all-resident variant if it fits 160 KB, else the streamed one. */
extern "C" int64_t lssvc_ffn_f16x3_lds_bytes(int32_t C, int32_t hidden, int32_t pre_cin) {
    const long long r = ffn_resident_lds(C, hidden, pre_cin);
    if (r <= 160 * 1024 && C <= 64 && pre_cin <= 64) return r;
    return ffn_stream_lds(C, hidden, pre_cin);
}

extern "C" int lssvc_ffn_f16x3_is_streamed(int32_t C, int32_t hidden, int32_t pre_cin) {
    return !(ffn_resident_lds(C, hidden, pre_cin) <= 160 * 1024 && C <= 64 && pre_cin <= 64);
}

extern "C" int lssvc_ffn_f16x3(const lssvc_ffn_desc *d, void *stream) {
    LSSVC_CHECK(d != nullptr, "ffn_f16x3: null descriptor");
    LSSVC_CHECK(view_ok(&d->out) && vec4_ok(&d->out), "ffn_f16x3: bad out view (needs C %% 4 == 0, 16-byte aligned)");
    const int C = d->out.C;
    LSSVC_CHECK(C % 16 == 0 && C >= 32 && C <= 128, "ffn_f16x3: C = %d must be a multiple of 16 in [32, 128]", C);
    LSSVC_CHECK(d->hidden > 0 && d->hidden % 32 == 0, "ffn_f16x3: hidden = %d must be a positive multiple of 32", d->hidden);
    LSSVC_CHECK(d->w1_16 && d->w2_16 && d->b1 && d->b2, "ffn_f16x3: missing FFN weights");
    LSSVC_CHECK(d->slope >= 0.0f && d->slope <= 1.0f, "ffn_f16x3: LeakyReLU slope %g outside [0, 1]", (double)d->slope);
    LSSVC_CHECK((long long)d->out.H * d->out.W < (1LL << 31) - 64, "ffn_f16x3: %d x %d pixels do not fit 32-bit indexing", d->out.H, d->out.W);
    const bool pre = d->pre_w16 != nullptr;
    FfnP p{};
    p.out = mk(&d->out);
    if (pre) {
        LSSVC_CHECK(view_ok(&d->pre_in) && vec4_ok(&d->pre_in) && same_hw(&d->pre_in, &d->out), "ffn_f16x3: bad pre_in view");
        LSSVC_CHECK(d->pre_in.C % 8 == 0 && d->pre_in.C <= 128, "ffn_f16x3: leading conv takes Cin %% 8 == 0, <= 128 (got %d)", d->pre_in.C);
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
    if (d->skip.ptr) {
        LSSVC_CHECK(view_ok(&d->skip) && vec4_ok(&d->skip) && same_shape(&d->skip, &d->out), "ffn_f16x3: bad skip view");
        p.skip = mk(&d->skip);
    } else {
        p.skip = mk_null();
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
    const int pre_cin = pre ? d->pre_in.C : 0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    static const int force_stream = getenv("LSSVC_FFN_STREAM") ? atoi(getenv("LSSVC_FFN_STREAM")) : 0;
    const long long resident = ffn_resident_lds(C, d->hidden, pre_cin);
    if (resident <= 160 * 1024 && C <= 64 && pre_cin <= 64 && !force_stream) {
        const size_t lds = (size_t)resident;
        if (pre) {
            if (cf == 4) return launch_ffn<4, true>(p, lds, st);
            if (cf == 3) return launch_ffn<3, true>(p, lds, st);
            return launch_ffn<2, true>(p, lds, st);
        }
        if (cf == 4) return launch_ffn<4, false>(p, lds, st);
        if (cf == 3) return launch_ffn<3, false>(p, lds, st);
        return launch_ffn<2, false>(p, lds, st);
    }
    const size_t lds = (size_t)ffn_stream_lds(C, d->hidden, pre_cin);
    LSSVC_CHECK(lds <= 160 * 1024, "ffn_f16x3: %zu bytes do not fit the 160 KB LDS (C=%d hidden=%d pre_cin=%d)", lds, C, d->hidden, pre_cin);
#define LSSVC_FFN_STREAM_CASE(n) \
    if (cf == n) return pre ? launch_ffn_stream<n, true>(p, lds, st) : launch_ffn_stream<n, false>(p, lds, st);
    LSSVC_FFN_STREAM_CASE(2) LSSVC_FFN_STREAM_CASE(3) LSSVC_FFN_STREAM_CASE(4) LSSVC_FFN_STREAM_CASE(6) LSSVC_FFN_STREAM_CASE(8)
#undef LSSVC_FFN_STREAM_CASE
    return fail("ffn_f16x3: no streamed kernel for C = %d", C);
}
