// conv_mfma_kernel.h -- the kernel template behind lssvc_conv2d (see conv_mfma.hip for the design notes).
// Instantiated per (kernel size, stride, vector-addressable inputs) in conv_inst_*.hip so the
// translation units build in parallel.
#pragma once
#include "common.h"

namespace lssvc {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvP {
    V in[LSSVC_CONV_MAX_INPUTS];
    int n_in;
    const float *w;
    const float *bias;
    int KH, KW, stride, pad_t, pad_l;
    int Cout, M_pad;
    int in_act;
    float in_slope;
    int epilogue;
    V gdn_x;
    int act;
    float slope;
    V res;
    V res2;          // second residual (fast epilogue only)
    float out_scale;
    int pixel_shuffle;
    V out;
    int Hout, Wout;  // conv-space output size (before pixel shuffle)
    int tiles_x, tiles_y, m_tiles;
    int in_vec[LSSVC_CONV_MAX_INPUTS];
    int n_chunks;    // total 8-channel chunks over all input segments
    int n_chunks16;  // total 16-channel chunks over all input segments (f16x3 kernels)
    const void *w16; // f16x3 mode: fp16 weights [plane hi|lo][chunk16][ky][kx][m][16]
    long long w16_plane;   // elements per plane
    float w16_unscale;     // f16x3 mode: accumulators *= this (power of two) before the epilogue
    int debug;       // perf-ablation switches (env LSSVC_CONV_DEBUG; results are WRONG when set): 1 skip prefetch loads, 2 skip LDS stores, 4 skip barriers, 8 no stagger, 16 skip LDS fragment reads, 32 skip the epilogue, 64 skip the fp32 -> fp16 split (K-sliced 1x1), 256 in-kernel stamps
    int out_vec, res_vec, gdn_vec;
    int fast_epi;    // host: Cout % 4 == 0, 16-byte addressable out / residual, no pixel shuffle, no GDN -> straight-line epilogue
    int in_split;    // every input is a PRE-SPLIT tensor (LSSVC_PREC_SPLIT_IN): [pixel][16-channel chunk][hi x16 | lo x16] fp16, activation applied
    int out_split;   // (never set: writing pre-split OUTPUTS from an epilogue was not built -- the round-5 A/B stopped at pre-split inputs made by lssvc_presplit)
    int gdn_fast;    // host: GDN / IGDN epilogue with Cout % 4 == 0, 16-byte addressable out / gdn_x / residual, no shuffle, out_scale 1 -> conv_epilogue_gdn
};

constexpr int CP = 12;  // LDS row pitch in floats (CK=8 + 4 pad)

// ---- fused epilogue shared by the conv kernels: bias -> GDN -> activation -> residual -> scale ->
//      (pixel-shuffle) store. Lane (li, lg) of wave `wave` holds, for fragment (f, r), channels
//      m0 + 16f + 4lg .. +3 of one pixel per fragment row r. ------------------------------------------------
template <int MF, int RPW>
__device__ __forceinline__ void conv_unscale(const ConvP &p, f32x4 (&acc)[MF][RPW]) {
#pragma unroll
    for (int f = 0; f < MF; ++f)
#pragma unroll
        for (int r = 0; r < RPW; ++r) acc[f][r] *= p.w16_unscale;
}

// Straight-line epilogue for the common case (p.fast_epi): unscale -> bias -> activation -> residual -> scale -> float4
// store. No scalar fallbacks, so no branches for the compiler to hang conservative waits on, and the arithmetic is
// written on 2-wide vectors so that it compiles to v_pk_mul_f32 / v_pk_add_f32 (the epilogue is VALU-bound on the
// kernels that keep 128 accumulators per wave). LeakyReLU / ReLU / none are one formula, max(v, s*v) with
// s = slope / 0 / 1: exact for 0 <= s <= 1 (the host only sets fast_epi then). `unscale` is the f16x3 weight
// prescale (1 for the fp32 kernels); multiplying by a power of two and by 1.0f is exact.
typedef float f32x2 __attribute__((ext_vector_type(2)));

// INTERIOR (wave-uniform, decided by the caller from the tile's position): every pixel row of this wave lies inside the
// image and every channel fragment below Cout, so no load or store needs a per-lane guard. With the guards the compiler
// wraps EACH conditional load / store in its own s_and_saveexec / s_cbranch_execz block -- ~50 branches per epilogue;
// interior tiles (all but the last tile row / column / M tile) get straight-line code instead. RES likewise lifts the
// "is there a residual" test out of the per-row code.
typedef __attribute__((address_space(3))) const float *lds_cfloat_ptr;

// BIAS_LDS: the bias vector comes from LDS (persistent kernels) -- a COMPILE-TIME choice: with a run-time one the compiler joins a
// path holding a global load in front of the first store and waits vmcnt(0) there, i.e. for every store still in flight.
// PIXF: (r, col) -> conv-space pixel index of column col (0..15) of fragment row r, or -1 if outside the image.
//
// LANE TRANSPOSITION. An MFMA 16x16 result has lane (lg, li) = 16 lg + li holding channels 4 lg .. 4 lg + 3 of pixel li: 16
// CONSECUTIVE lanes touch 16 different pixels, 16 bytes each. The memory pipeline coalesces a wave's accesses over runs of
// consecutive lanes: measured (tools/probes/store_probe.hip) this native pattern sustains 16.6 B/clk per CU, any pattern
// whose 4 consecutive lanes cover 64 contiguous bytes 59 B/clk -- and the 96 KB tile of a 64-channel 3x3 conv spent
// 6.5 k of its 33 k cycles issuing stores. Each result dword is therefore moved with ds_bpermute (the LDS crossbar, no LDS
// memory) to lane 4 li + lg, so that 4 consecutive lanes hold the 16 channels of one pixel of a fragment; bias and
// activation are applied before the move (native lane = native channel), residuals are loaded and added and the
// result stored in the transposed layout. Same arithmetic per element, same results.
template <int MF, int RPW, bool PS, int RES, bool INTERIOR, bool ACT, bool BIAS_LDS, typename PIXF, int R0 = 0>      // RES: number of residual operands (0, 1, 2); rows R0 .. RPW-1
__device__ __forceinline__ void conv_epilogue_fast_impl(const ConvP &p, f32x4 (&acc)[MF][RPW], const PIXF &pixf, int m0,
                                                        int lg, float unscale, lds_cfloat_ptr bias_lds) {
    const float s_neg = p.act == LSSVC_ACT_LRELU ? p.slope : (p.act == LSSVC_ACT_RELU ? 0.0f : 1.0f);
    constexpr bool has_res = RES > 0;
    const f32x2 us = {unscale, unscale}, sn = {s_neg, s_neg};      // (out_scale == 1 here: the host keeps other convs off this path)
    const int lane = threadIdx.x & 63;
    const int tcol = lane >> 2, tq = lane & 3;                    // transposed layout: pixel column, channel quad of the fragment
    const int bp_addr = ((lane & 3) * 16 + (lane >> 2)) * 4;      // ds_bpermute source lane (byte address): native lane (lg = tq, li = tcol)
    f32x2 bb[MF][2];
#pragma unroll
    for (int f = 0; f < MF; ++f) {
        const int mb = m0 + f * 16 + 4 * lg;
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (BIAS_LDS) {
            // the persistent kernels keep the (zero-padded, M_pad long) bias vector in LDS: a dependent GLOBAL load at the
            // head of every tile's epilogue costs a full L2 round trip with the matrix pipe idle
            const f32x4 v = *reinterpret_cast<__attribute__((address_space(3))) const f32x4 *>(bias_lds + mb);
            t = make_float4(v[0], v[1], v[2], v[3]);
        } else if (p.bias && (INTERIOR || mb < p.Cout)) {
            t = *reinterpret_cast<const float4 *>(p.bias + mb);
        }
        bb[f][0] = f32x2{t.x, t.y};
        bb[f][1] = f32x2{t.z, t.w};
    }
    // pixel-shuffle store (fast_epi == 2): channel m = q*cps + c goes to sub-pixel q = dy*2+dx, channel c (the host
    // permuted the weights so); a lane's 4 channels never straddle q because cps % 4 == 0
    int ps_off[PS ? MF : 1];
    if (PS) {
        const int cps = p.Cout >> 2;
#pragma unroll
        for (int f = 0; f < MF; ++f) {
            const int mt = m0 + f * 16 + 4 * tq;
            const int q = mt / cps, c = mt - q * cps;
            ps_off[f] = ((q >> 1) * p.out.W + (q & 1)) * p.out.ld + c;      // < 2 * W * ld: fits int
        }
    }
    float4 rs[2][MF], rs2[2][RES > 1 ? MF : 1];
    auto load_res = [&](int r, float4 (&d)[MF], float4 (&d2)[RES > 1 ? MF : 1]) {
        const long long px = pixf(r, tcol);
#pragma unroll
        for (int f = 0; f < MF; ++f) {
            const int mt = m0 + f * 16 + 4 * tq;
            d[f] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (RES > 1) d2[f] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (has_res && (INTERIOR || (px >= 0 && mt < p.Cout))) {
                d[f] = *reinterpret_cast<const float4 *>(p.res.p + (size_t)px * p.res.ld + mt);
                if (RES > 1) d2[f] = *reinterpret_cast<const float4 *>(p.res2.p + (size_t)px * p.res2.ld + mt);
            }
        }
    };
    load_res(R0, rs[R0 & 1], rs2[R0 & 1]);
#pragma unroll
    for (int r = R0; r < RPW; ++r) {
        if (r + 1 < RPW) load_res(r + 1, rs[(r + 1) & 1], rs2[(r + 1) & 1]);      // issued before row r's stores (see conv_epilogue_flat)
        const long long px = pixf(r, tcol);
        const size_t opix = (size_t)((INTERIOR || px >= 0) ? px : 0);
        float *orow = p.out.p + opix * p.out.ld + m0 + 4 * tq;
        float *srow = nullptr;                                   // pixel-shuffle: the 2x2 output block of this conv pixel
        if (PS) {
            const int oy = (int)(opix / p.Wout), ox = (int)(opix - (size_t)oy * p.Wout);
            srow = p.out.p + ((size_t)(2 * oy) * p.out.W + 2 * ox) * p.out.ld;
        }
        f32x2 t0[MF], t1[MF];
#pragma unroll
        for (int f = 0; f < MF; ++f) {
            f32x2 v0 = f32x2{acc[f][r][0], acc[f][r][1]} * us + bb[f][0];
            f32x2 v1 = f32x2{acc[f][r][2], acc[f][r][3]} * us + bb[f][1];
            if (ACT) {                                                        // no activation: max(v, 1 * v) == v, skipped
                const f32x2 n0 = v0 * sn, n1 = v1 * sn;
                v0 = f32x2{fmaxf(v0.x, n0.x), fmaxf(v0.y, n0.y)};
                v1 = f32x2{fmaxf(v1.x, n1.x), fmaxf(v1.y, n1.y)};
            }
            t0[f] = f32x2{__int_as_float(__builtin_amdgcn_ds_bpermute(bp_addr, __float_as_int(v0.x))),
                          __int_as_float(__builtin_amdgcn_ds_bpermute(bp_addr, __float_as_int(v0.y)))};
            t1[f] = f32x2{__int_as_float(__builtin_amdgcn_ds_bpermute(bp_addr, __float_as_int(v1.x))),
                          __int_as_float(__builtin_amdgcn_ds_bpermute(bp_addr, __float_as_int(v1.y)))};
        }
#pragma unroll
        for (int f = 0; f < MF; ++f) {
            const int mt = m0 + f * 16 + 4 * tq;
            f32x2 v0 = t0[f], v1 = t1[f];
            if (RES > 0) {
                v0 = v0 + f32x2{rs[r & 1][f].x, rs[r & 1][f].y};
                v1 = v1 + f32x2{rs[r & 1][f].z, rs[r & 1][f].w};
            }
            if (RES > 1) {
                v0 = v0 + f32x2{rs2[r & 1][f].x, rs2[r & 1][f].y};
                v1 = v1 + f32x2{rs2[r & 1][f].z, rs2[r & 1][f].w};
            }
            float *dst = orow + f * 16;
            if (PS) dst = srow + ps_off[f];
            if (INTERIOR || (px >= 0 && mt < p.Cout))
                *reinterpret_cast<float4 *>(dst) = make_float4(v0.x, v0.y, v1.x, v1.y);
        }
    }
}

template <int MF, int RPW, bool BIAS_LDS = false, int R0 = 0, typename PIXF>
__device__ __forceinline__ void conv_epilogue_fast_f(const ConvP &p, f32x4 (&acc)[MF][RPW], const PIXF &pix, int m0,
                                                     int lg, float unscale = 1.0f, bool interior = false,
                                                     lds_cfloat_ptr bias_lds = nullptr) {
    // (every condition is wave-uniform: scalar branches)
    const bool act = p.act != LSSVC_ACT_NONE;
#define LSSVC_EPI_CALL(PS_, RES_)                                                                                              \
    do {                                                                                                                       \
        if (interior) {                                                                                                        \
            if (act) conv_epilogue_fast_impl<MF, RPW, PS_, RES_, true, true, BIAS_LDS, PIXF, R0>(p, acc, pix, m0, lg, unscale, bias_lds);  \
            else conv_epilogue_fast_impl<MF, RPW, PS_, RES_, true, false, BIAS_LDS, PIXF, R0>(p, acc, pix, m0, lg, unscale, bias_lds);     \
        } else {                                                                                                               \
            conv_epilogue_fast_impl<MF, RPW, PS_, RES_, false, true, BIAS_LDS, PIXF, R0>(p, acc, pix, m0, lg, unscale, bias_lds);          \
        }                                                                                                                      \
    } while (0)
    if (p.fast_epi == 2) LSSVC_EPI_CALL(true, 0);                            // pixel-shuffle store (never with a residual)
    else if (p.res2.p != nullptr) LSSVC_EPI_CALL(false, 2);
    else if (p.res.p != nullptr) LSSVC_EPI_CALL(false, 1);
    else LSSVC_EPI_CALL(false, 0);
#undef LSSVC_EPI_CALL
}


// Memory-op ordering matters here: on gfx9/CDNA loads and stores share the vmcnt counter, so a wait for a load that
// was issued AFTER a store also waits for that store's acknowledgement. The bias is therefore loaded once up front,
// and the side inputs of row r+1 (GDN's x, the residual) are issued BEFORE row r's stores: every wait then names
// only loads that are older than all stores in flight, and the stores of a tile stream out back to back
// (measured on the persistent 3x3 kernel: 530 ns per store instruction before, i.e. one full round trip each).
template <int MF>
struct EpiSide {
    float4 g[MF];     // GDN input x
    float4 rs[MF];    // residual
};

// The GDN / IGDN epilogue for the common case (p.gdn_fast; round 6): what conv_epilogue_flat computes per element -- bias, sqrt, the
// normalisation x * (1 / s) | x * s | x / s, activation, residual -- in the same order with the same roundings, but with the kind of
// normalisation a COMPILE-TIME parameter and float4 loads / stores only. The general routine carries all three normalisations, the
// scalar fall-backs of every load and store, the pixel-shuffle store and the output scale behind run-time tests: 7 k instructions per
// 32-pixel group in the GDN 1x1 kernels (two correctly rounded divisions compiled per element, one of them never executed), which ran
// at 2.0 TB/s where the same kernel without the normalisation streams at 4.6 (profiles/r06_gdn_ab.txt).
template <int MF, int RPW, int EPI>
__device__ __forceinline__ void conv_epilogue_gdn_impl(const ConvP &p, f32x4 (&acc)[MF][RPW], const long long (&pix)[RPW], int m0, int lg) {
    float4 bb[MF];
    bool mv[MF];                                        // this lane's 4 channels of fragment f exist (Cout % 4 == 0: all four or none)
#pragma unroll
    for (int f = 0; f < MF; ++f) {
        const int mb = m0 + f * 16 + 4 * lg;
        mv[f] = mb < p.Cout;
        bb[f] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias && mv[f]) bb[f] = *reinterpret_cast<const float4 *>(p.bias + mb);
    }
    const bool has_res = p.res.p != nullptr;
    float4 gx[2][MF], rs[2][MF];
    auto load_side = [&](int r, float4 (&g)[MF], float4 (&q)[MF]) {
        const size_t opix = (size_t)(pix[r] >= 0 ? pix[r] : 0);
#pragma unroll
        for (int f = 0; f < MF; ++f) {
            const int mb = mv[f] ? m0 + f * 16 + 4 * lg : 0;
            g[f] = *reinterpret_cast<const float4 *>(p.gdn_x.p + opix * p.gdn_x.ld + mb);
            if (has_res) q[f] = *reinterpret_cast<const float4 *>(p.res.p + opix * p.res.ld + mb);
        }
    };
    load_side(0, gx[0], rs[0]);
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        if (r + 1 < RPW) load_side(r + 1, gx[(r + 1) & 1], rs[(r + 1) & 1]);      // issued before row r's stores (see conv_epilogue_flat)
        const size_t opix = (size_t)(pix[r] >= 0 ? pix[r] : 0);
#pragma unroll
        for (int f = 0; f < MF; ++f) {
            const int mb = m0 + f * 16 + 4 * lg;
            float v[4] = {acc[f][r][0] + bb[f].x, acc[f][r][1] + bb[f].y, acc[f][r][2] + bb[f].z, acc[f][r][3] + bb[f].w};
            const float x[4] = {gx[r & 1][f].x, gx[r & 1][f].y, gx[r & 1][f].z, gx[r & 1][f].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float sq = sqrtf(v[j]);
                if (EPI == LSSVC_EPI_X_MUL_RSQRT) v[j] = x[j] * (1.0f / sq);
                else if (EPI == LSSVC_EPI_X_MUL_SQRT) v[j] = x[j] * sq;
                else v[j] = x[j] / sq;
            }
            if (p.act == LSSVC_ACT_LRELU) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : v[j] * p.slope;
            } else if (p.act == LSSVC_ACT_RELU) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : 0.f;
            }
            if (has_res) {
                v[0] += rs[r & 1][f].x; v[1] += rs[r & 1][f].y; v[2] += rs[r & 1][f].z; v[3] += rs[r & 1][f].w;
            }
            if (pix[r] >= 0 && mv[f]) *reinterpret_cast<float4 *>(p.out.p + opix * p.out.ld + mb) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

template <int MF, int RPW>
__device__ __forceinline__ void conv_epilogue_gdn(const ConvP &p, f32x4 (&acc)[MF][RPW], const long long (&pix)[RPW], int m0, int lg) {
    if (p.epilogue == LSSVC_EPI_X_MUL_RSQRT) conv_epilogue_gdn_impl<MF, RPW, LSSVC_EPI_X_MUL_RSQRT>(p, acc, pix, m0, lg);      // (wave-uniform)
    else if (p.epilogue == LSSVC_EPI_X_MUL_SQRT) conv_epilogue_gdn_impl<MF, RPW, LSSVC_EPI_X_MUL_SQRT>(p, acc, pix, m0, lg);
    else conv_epilogue_gdn_impl<MF, RPW, -1>(p, acc, pix, m0, lg);
}

template <int MF, int RPW, bool GDN = true, typename PIXF>
__device__ __forceinline__ void conv_epilogue_flat(const ConvP &p, f32x4 (&acc)[MF][RPW], const long long (&pix)[RPW], const PIXF &pixf,
                                                   int m0, int lg, bool interior = false, lds_cfloat_ptr bias_lds = nullptr) {
    // pix[r] = conv-space pixel index oy*Wout + ox of this lane's column of fragment row r, or -1 if outside; pixf(r, col) the
    // same for any column of the tile (conv_epilogue_fast_impl); interior (wave-uniform): the caller knows that no pixel of
    // the wave's rows is outside and m0 + 16 MF <= Cout
    if (p.fast_epi) {
        conv_epilogue_fast_f<MF, RPW, false>(p, acc, pixf, m0, lg, 1.0f, interior);
        return;
    }
    if constexpr (GDN) {
        if (p.gdn_fast) {
            conv_epilogue_gdn<MF, RPW>(p, acc, pix, m0, lg);
            return;
        }
    }
    const int cps = p.Cout >> 2;  // channels after pixel shuffle
    float4 bb[MF];
#pragma unroll
    for (int f = 0; f < MF; ++f) {
        const int mb = m0 + f * 16 + 4 * lg;
        bb[f] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias_lds) {
            // the persistent kernels keep the bias in the LDS (zero where there is none): a GLOBAL load here queues behind the producers'
            // patch requests in the CU's memory pipeline and the tile's epilogue waits two microseconds for sixteen bytes (round 6:
            // profiles/r06_roles_ablation.txt -- the 2- and 3-channel heads spent 15 % of their time there)
            const f32x4 v = *reinterpret_cast<__attribute__((address_space(3))) const f32x4 *>(bias_lds + mb);
            bb[f] = make_float4(v[0], v[1], v[2], v[3]);
        } else if (p.bias && mb < p.Cout) bb[f] = *reinterpret_cast<const float4 *>(p.bias + mb);  // bias is M_pad long
    }
    auto gather4 = [&](const float *src, int mb, bool vec) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (vec && mb + 3 < p.Cout) {
            t = *reinterpret_cast<const float4 *>(src);
        } else {
            if (mb + 0 < p.Cout) t.x = src[0];
            if (mb + 1 < p.Cout) t.y = src[1];
            if (mb + 2 < p.Cout) t.z = src[2];
            if (mb + 3 < p.Cout) t.w = src[3];
        }
        return t;
    };
    auto load_side = [&](int r, EpiSide<MF> &s) {
        if (pix[r] < 0) return;
        const size_t opix = (size_t)pix[r];
#pragma unroll
        for (int f = 0; f < MF; ++f) {
            const int mb = m0 + f * 16 + 4 * lg;
            if (mb >= p.Cout) continue;
            if (GDN && p.epilogue != LSSVC_EPI_NONE) s.g[f] = gather4(p.gdn_x.p + opix * p.gdn_x.ld + mb, mb, p.gdn_vec);
            if (p.res.p) s.rs[f] = gather4(p.res.p + opix * p.res.ld + mb, mb, p.res_vec);
        }
    };
    EpiSide<MF> side[2];
    load_side(0, side[0]);
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        __builtin_amdgcn_sched_barrier(0);        // keep the prefetch one row deep (else every row's loads get hoisted: spills)
        if (r + 1 < RPW) load_side(r + 1, side[(r + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        if (pix[r] < 0) continue;
        const EpiSide<MF> &sd = side[r & 1];
        const size_t opix = (size_t)pix[r];
        const int oy = (int)(opix / p.Wout), ox = (int)(opix - (size_t)oy * p.Wout);   // only the pixel-shuffle store needs them
#pragma unroll
        for (int f = 0; f < MF; ++f) {
            const int mb = m0 + f * 16 + 4 * lg;
            if (mb >= p.Cout) continue;
            float v[4] = {acc[f][r][0] + bb[f].x, acc[f][r][1] + bb[f].y, acc[f][r][2] + bb[f].z, acc[f][r][3] + bb[f].w};
            const bool full = (mb + 3 < p.Cout);
            if (GDN && p.epilogue != LSSVC_EPI_NONE) {
                const float x[4] = {sd.g[f].x, sd.g[f].y, sd.g[f].z, sd.g[f].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float sq = sqrtf(v[j]);
                    if (p.epilogue == LSSVC_EPI_X_MUL_RSQRT) v[j] = x[j] * (1.0f / sq);
                    else if (p.epilogue == LSSVC_EPI_X_MUL_SQRT) v[j] = x[j] * sq;
                    else v[j] = x[j] / sq;
                }
            }
            if (p.act == LSSVC_ACT_LRELU) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : v[j] * p.slope;
            } else if (p.act == LSSVC_ACT_RELU) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : 0.f;
            }
            if (p.res.p) {
                v[0] += sd.rs[f].x; v[1] += sd.rs[f].y; v[2] += sd.rs[f].z; v[3] += sd.rs[f].w;
            }
            if (p.out_scale != 1.0f) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] *= p.out_scale;
            }
            if (!p.pixel_shuffle) {
                float *dst = p.out.p + opix * p.out.ld + mb;
                if (full && p.out_vec) {
                    *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    for (int j = 0; j < 4; ++j)
                        if (mb + j < p.Cout) dst[j] = v[j];
                }
            } else {
                // m = q*cps + c, q = dy*2+dx  (weights were permuted on the host)
                if (full && p.out_vec && (cps & 3) == 0) {
                    const int q = mb / cps, c = mb - q * cps;
                    float *dst = p.out.p + ((size_t)(2 * oy + (q >> 1)) * p.out.W + 2 * ox + (q & 1)) * p.out.ld + c;
                    *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    for (int j = 0; j < 4; ++j) {
                        const int m = mb + j;
                        if (m >= p.Cout) break;
                        const int q = m / cps, c = m - q * cps;
                        p.out.p[((size_t)(2 * oy + (q >> 1)) * p.out.W + 2 * ox + (q & 1)) * p.out.ld + c] = v[j];
                    }
                }
            }
        }
    }
}

// conv_epilogue_flat's plain case -- no GDN, no pixel shuffle, outputs that are not 16-byte addressable (the 2- and 3-channel heads) -- as
// straight-line code for the persistent kernels (round 6): the same operations on every element in the same order (bias, activation,
// residual, out_scale), scalar accesses, the bias from the LDS, every row's residuals requested before the first store. The general
// routine compiles to ~500 instructions per row with its pixel-shuffle address arithmetic; the narrow-head kernel spent 15 % of its
// time in it (profiles/r06_roles_ablation.txt).
template <int MF, int RPW>
__device__ __forceinline__ void conv_epilogue_plain(const ConvP &p, f32x4 (&acc)[MF][RPW], const long long (&pix)[RPW], int m0, int lg,
                                                    lds_cfloat_ptr bias_lds) {
#pragma unroll
    for (int f = 0; f < MF; ++f) {
        const int mb = m0 + f * 16 + 4 * lg;
        if (mb >= p.Cout) continue;
        const int nch = p.Cout - mb;                                   // channels of this lane's four that exist (>= 1)
        const f32x4 bb = *reinterpret_cast<__attribute__((address_space(3))) const f32x4 *>(bias_lds + mb);      // (zero where there is no bias)
        float rs[RPW][4];
        if (p.res.p) {
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const float *rp = p.res.p + (size_t)(pix[r] < 0 ? 0 : pix[r]) * p.res.ld + mb;
#pragma unroll
                for (int j = 0; j < 4; ++j) rs[r][j] = (pix[r] >= 0 && j < nch) ? rp[j] : 0.f;
            }
        }
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            if (pix[r] < 0) continue;
            float v[4] = {acc[f][r][0] + bb[0], acc[f][r][1] + bb[1], acc[f][r][2] + bb[2], acc[f][r][3] + bb[3]};
            if (p.act == LSSVC_ACT_LRELU) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : v[j] * p.slope;
            } else if (p.act == LSSVC_ACT_RELU) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : 0.f;
            }
            if (p.res.p) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += rs[r][j];
            }
            if (p.out_scale != 1.0f) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] *= p.out_scale;
            }
            float *dst = p.out.p + (size_t)pix[r] * p.out.ld + mb;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < nch) dst[j] = v[j];
        }
    }
}

template <int MF, int RPW, bool GDN = true>
__device__ __forceinline__ void conv_epilogue(const ConvP &p, f32x4 (&acc)[MF][RPW], int oy0, int ox0, int m0, int wave,
                                              int li, int lg) {
    long long pix[RPW];
    const int ox = ox0 + li;
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int oy = oy0 + wave * RPW + r;
        pix[r] = (oy < p.Hout && ox < p.Wout) ? (long long)oy * p.Wout + ox : -1;
    }
    const bool interior = oy0 + wave * RPW + RPW <= p.Hout && ox0 + 16 <= p.Wout && m0 + 16 * MF <= p.Cout;
    const int oy_w = oy0 + wave * RPW;
    auto pixf = [&](int r, int col) {
        return (oy_w + r < p.Hout && ox0 + col < p.Wout) ? (long long)(oy_w + r) * p.Wout + ox0 + col : -1LL;
    };
    conv_epilogue_flat<MF, RPW, GDN>(p, acc, pix, pixf, m0, lg, interior);
}

// One K "phase" = one 8-channel chunk x RPP kernel rows. Small kernels (<=3x3) take all rows in one
// phase (the whole KSxKS filter slab of the chunk sits in LDS); 7x7 takes one kernel row per phase.
template <int KS>
struct PhaseRows {
    static constexpr int value = (KS <= 3) ? KS : 1;
};

struct KState {
    int seg, c0, ky, kc;
};

template <int MF, int RPW, int KS, int S, bool VEC>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvP p) {
    constexpr int TM = 16 * MF;
    constexpr int TH = 4 * RPW;
    constexpr int PH = (TH - 1) * S + KS, PW = 15 * S + KS;      // LDS halo patch, pixels
    constexpr int RPP = PhaseRows<KS>::value;                     // kernel rows per phase
    // 1x1 convs have one tap per chunk, far too little MFMA work per barrier: they take KCH channel
    // chunks per phase instead, each in its own LDS patch plane ("tap" t of the phase = chunk t).
    constexpr int KCH = 1;   // (KS == 1 && S == 1) ? 4 : 1 measured slower: 1x1 convs are prologue/epilogue-bound, not barrier-bound
    constexpr int NT = RPP * KS * KCH;                            // (tap | chunk) units per phase
    constexpr int PLANE_ITEMS = PH * PW * 2;                      // float4 items per patch plane
    constexpr int PATCH_ITEMS = PLANE_ITEMS * KCH;
    constexpr int W_ITEMS = NT * TM * 2;
    constexpr int NP = (PATCH_ITEMS + 255) / 256;                 // prefetch registers (float4) per thread
    constexpr int NW = (W_ITEMS + 255) / 256;
    __shared__ __attribute__((aligned(16))) float patch[KCH * PH * PW * CP];
    __shared__ __attribute__((aligned(16))) float wts[NT * TM * CP];

    // XCD-aware tile order: consecutive workgroup ids are dealt round-robin over the 8 XCDs, so
    // remap (bijectively) to give each XCD a contiguous run of tiles: neighbouring pixel tiles share
    // halos and the M tiles of one pixel tile share the whole input patch through that XCD's L2.
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int mt = bid % p.m_tiles;
    const int pt = bid / p.m_tiles;
    const int tx = pt % p.tiles_x, ty = pt / p.tiles_x;
    const int oy0 = ty * TH, ox0 = tx * 16, m0 = mt * TM;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 15;  // row of A / column of B
    const int lg = lane >> 4;  // k group

    f32x4 acc[MF][RPW];
#pragma unroll
    for (int a = 0; a < MF; ++a)
#pragma unroll
        for (int b = 0; b < RPW; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int gy0 = oy0 * S - p.pad_t, gx0 = ox0 * S - p.pad_l;

    float4 preg[NP], wreg[NW];
    const bool sq = p.in_act == LSSVC_INACT_SQUARE;
    const float in_slope = p.in_act == LSSVC_INACT_LRELU ? p.in_slope : 1.0f;

    // Loop-invariant staging geometry, computed once: for every float4 this thread stages, the source
    // pixel (or -1 when it falls in the zero padding / past the tile) and the weight row offset.
    int ppix[NP];   // input pixel index gy*W + gx, or -1
    int woff[NW];   // float offset inside one (chunk, ky) weight slab, or -1
    int wlds[NW];   // float offset inside the LDS weight tile
    {
        const int Hin = p.in[0].H, Win = p.in[0].W;     // all inputs share H, W
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int idx = tid + i * 256;
            const int pix = (idx % PLANE_ITEMS) >> 1;
            const int py = pix / PW, px = pix - py * PW;
            const int gy = gy0 + py, gx = gx0 + px;
            const bool ok = idx < PATCH_ITEMS && gy >= 0 && gy < Hin && gx >= 0 && gx < Win;
            ppix[i] = ok ? gy * Win + gx : -1;
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int idx = tid + i * 256;
            const int tap = idx / (TM * 2);          // 0 .. RPP*KS-1
            const int r = idx - tap * (TM * 2);
            const int m = r >> 1, half = r & 1;
            const bool ok = idx < W_ITEMS && (m0 + m) < p.M_pad;
            woff[i] = ok ? (tap * p.M_pad + m0 + m) * 8 + half * 4 : -1;
            wlds[i] = (tap * TM + m) * CP + half * 4;
        }
    }
    const int half4 = (tid & 1) * 4;     // idx & 1 == tid & 1 for every item (256 is even)

    // global -> registers (issued before the MFMA phase of the previous step; no wait here)
    // VEC: every input view is 16-byte addressable (C, ld multiples of 4). Loads are then unconditional
    // (out-of-range items read pixel 0 / channel 0 and are zeroed when written to LDS), which keeps the
    // prefetch free of branches so no wait lands between the loads and the MFMA phase.
    int c_left = 0;   // channels of the segment at and after the chunk held in preg (validity of each quad)
    auto load_patch = [&](const KState &k) {
        const V X = p.in[k.seg];
        c_left = X.C - k.c0;
        if constexpr (VEC) {
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int cq = ((tid + i * 256) / PLANE_ITEMS) * 8 + half4;      // channel quad inside the phase
                const int pp = ppix[i] >= 0 ? ppix[i] : 0;
                preg[i] = *reinterpret_cast<const float4 *>(X.p + (size_t)pp * X.ld + (cq < c_left ? k.c0 + cq : 0));
            }
        } else {
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int cq = ((tid + i * 256) / PLANE_ITEMS) * 8 + half4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ppix[i] >= 0 && cq < c_left) {
                    const float *src = X.p + (size_t)ppix[i] * X.ld + k.c0 + cq;
                    v.x = src[0];
                    if (cq + 1 < c_left) v.y = src[1];
                    if (cq + 2 < c_left) v.z = src[2];
                    if (cq + 3 < c_left) v.w = src[3];
                }
                preg[i] = v;
            }
        }
    };
    auto store_patch = [&]() {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int idx = tid + i * 256;
            if (idx < PATCH_ITEMS) {
                float4 v = preg[i];
                if constexpr (VEC) {
                    const int cq = (idx / PLANE_ITEMS) * 8 + half4;
                    if (ppix[i] < 0 || cq >= c_left) v = make_float4(0.f, 0.f, 0.f, 0.f);
                }
                // input activation, branch-free: v * g with g = v (square) or (v > 0 ? 1 : slope); "none" is
                // slope 1 (x * 1.0f is exact). Zero padding stays zero under all three.
                v.x *= sq ? v.x : (v.x > 0.f ? 1.0f : in_slope);
                v.y *= sq ? v.y : (v.y > 0.f ? 1.0f : in_slope);
                v.z *= sq ? v.z : (v.z > 0.f ? 1.0f : in_slope);
                v.w *= sq ? v.w : (v.w > 0.f ? 1.0f : in_slope);
                *reinterpret_cast<float4 *>(patch + ((tid >> 1) + i * 128) * CP + half4) = v;
            }
        }
    };
    auto load_w = [&](const KState &k) {
        const float *wsrc = p.w + ((size_t)(k.kc * KS + k.ky) * KS) * p.M_pad * 8;   // rows ky .. ky+RPP-1 are contiguous
        // with KCH > 1 "tap" t is chunk kc+t: chunks past the end of the tensor are clamped to a valid
        // address (their patch plane is all zeros, so the product vanishes whatever is read)
        const int t_max = KCH > 1 ? p.n_chunks - k.kc - 1 : NT;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            int o = woff[i] >= 0 ? woff[i] : 0;
            if (KCH > 1 && (tid + i * 256) / (TM * 2) > t_max) o = 0;
            wreg[i] = *reinterpret_cast<const float4 *>(wsrc + o);
        }
    };
    auto store_w = [&]() {
#pragma unroll
        for (int i = 0; i < NW; ++i)
            if (tid + i * 256 < W_ITEMS)
                *reinterpret_cast<float4 *>(wts + wlds[i]) = woff[i] >= 0 ? wreg[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto advance = [&](KState k) {
        k.ky += RPP;
        if (k.ky >= KS) {
            k.ky = 0;
            const int left = (p.in[k.seg].C - k.c0 + LSSVC_CONV_CK - 1) / LSSVC_CONV_CK;   // chunks left in this segment
            k.kc += left < KCH ? left : KCH;
            k.c0 += LSSVC_CONV_CK * KCH;
            if (k.c0 >= p.in[k.seg].C) {
                k.c0 = 0;
                ++k.seg;
            }
        }
        return k;
    };

    // ---- software-pipelined K loop. Per phase: issue the global loads of phase i+1 into registers,
    //      run the MFMAs of phase i out of LDS, barrier, write the registers to LDS, barrier. Loads and
    //      their LDS stores sit in the same iteration, so no prefetch register is live across the
    //      back-edge and the only vmcnt wait is the one in front of the ds_writes, after the MFMAs. ------
    KState cur{0, 0, 0, 0};
    load_patch(cur);
    load_w(cur);
    // Two workgroups share each SIMD's matrix pipe. Started together they stay in lockstep (both stage,
    // then both issue MFMAs) and the pipe idles during every staging step; delaying the workgroup whose
    // first wave sits in an odd wave slot by about half a phase makes one stage while the other computes.
    // Pure scheduling: results do not depend on it.
    {
        __shared__ int odd_slot;
        if (tid == 0) odd_slot = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (3 << 11)) & 1;   // HW_REG_HW_ID.wave_id
        __syncthreads();
        if (odd_slot && !(p.debug & 8)) __builtin_amdgcn_s_sleep(64);
    }
    store_patch();
    store_w();
    __syncthreads();
    while (true) {
        const KState nxt = advance(cur);
        const bool more = nxt.seg < p.n_in;
        if (more && !(p.debug & 1)) {
            load_w(nxt);
            if (nxt.ky == 0) load_patch(nxt);
        }
        // taps of this phase, with the LDS fragment reads of tap t+1 issued before the MFMAs of tap t
        // (two fragment register sets); sched_barrier keeps the compiler from sinking the reads back
        // down to their first use, which would expose the LDS latency once per tap.
        float2 fa[2][MF], fb[2][RPW];
        auto read_frags = [&](int t, float2 (&a)[MF], float2 (&b)[RPW]) {
            const int plane = KCH > 1 ? t : 0;
            const int tt = KCH > 1 ? 0 : t;
            const int ry = tt / KS, kx = tt - ry * KS;
            const int ky = cur.ky + ry;
#pragma unroll
            for (int f = 0; f < MF; ++f)
                a[f] = *reinterpret_cast<const float2 *>(wts + (t * TM + f * 16 + li) * CP + 2 * lg);
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int row = wave * RPW + r;
                b[r] = *reinterpret_cast<const float2 *>(patch + (plane * PH * PW + (row * S + ky) * PW + li * S + kx) * CP + 2 * lg);
            }
        };
        read_frags(0, fa[0], fb[0]);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t + 1 < NT && !(p.debug & 16)) read_frags(t + 1, fa[(t + 1) & 1], fb[(t + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int f = 0; f < MF; ++f)
#pragma unroll
                for (int r = 0; r < RPW; ++r)
                    acc[f][r] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[t & 1][f].x, fb[t & 1][r].x, acc[f][r], 0, 0, 0);
#pragma unroll
            for (int f = 0; f < MF; ++f)
#pragma unroll
                for (int r = 0; r < RPW; ++r)
                    acc[f][r] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[t & 1][f].y, fb[t & 1][r].y, acc[f][r], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!more) break;
        if (!(p.debug & 4)) __syncthreads();  // every wave has finished reading this phase's LDS tiles
        if (!(p.debug & 2)) {
            if (nxt.ky == 0) store_patch();
            store_w();
        }
        if (!(p.debug & 4)) __syncthreads();
        cur = nxt;
    }

    conv_epilogue<MF, RPW>(p, acc, oy0, ox0, m0, wave, li, lg);
}

template <int MF, int RPW, int KS, int S, bool VEC>
static int launch(const ConvP &p, hipStream_t st) {
    const int TH = 4 * RPW;
    ConvP q = p;
    q.tiles_x = (p.Wout + 15) / 16;
    q.tiles_y = (p.Hout + TH - 1) / TH;
    q.m_tiles = (p.M_pad / 16 + MF - 1) / MF;
    const long long blocks = (long long)q.tiles_x * q.tiles_y * q.m_tiles;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return fail("conv2d: bad grid %lld", blocks);
    hipLaunchKernelGGL((conv_mfma_kernel<MF, RPW, KS, S, VEC>), dim3((unsigned)blocks), dim3(256), 0, st, q);
    return launch_status("conv2d");
}

template <int KS, int S, bool VEC>
int dispatch_tile(const ConvP &p, int MF, int RPW, hipStream_t st) {
#define LSSVC_CONV_CASE(mf, rpw) \
    if (MF == mf && RPW == rpw) return launch<mf, rpw, KS, S, VEC>(p, st);
    LSSVC_CONV_CASE(1, 1) LSSVC_CONV_CASE(1, 2) LSSVC_CONV_CASE(1, 4)
    LSSVC_CONV_CASE(2, 1) LSSVC_CONV_CASE(2, 2) LSSVC_CONV_CASE(2, 4)
    LSSVC_CONV_CASE(3, 1) LSSVC_CONV_CASE(3, 2) LSSVC_CONV_CASE(3, 4)
    LSSVC_CONV_CASE(4, 1) LSSVC_CONV_CASE(4, 2) LSSVC_CONV_CASE(4, 4)
#undef LSSVC_CONV_CASE
    return fail("conv2d: no kernel for MF=%d RPW=%d", MF, RPW);
}


// one explicit instantiation per (KS, S, VEC), defined in conv_inst_*.hip
#define LSSVC_DECLARE_CONV(KS, S)                                                             \
    extern template int dispatch_tile<KS, S, true>(const ConvP &, int, int, hipStream_t);  \
    extern template int dispatch_tile<KS, S, false>(const ConvP &, int, int, hipStream_t);
LSSVC_DECLARE_CONV(1, 1) LSSVC_DECLARE_CONV(1, 2) LSSVC_DECLARE_CONV(2, 1)
LSSVC_DECLARE_CONV(3, 1) LSSVC_DECLARE_CONV(3, 2) LSSVC_DECLARE_CONV(7, 1)
#undef LSSVC_DECLARE_CONV

}  // namespace lssvc
