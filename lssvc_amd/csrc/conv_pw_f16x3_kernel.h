// conv_pw_f16x3_kernel.h -- 1x1 (pointwise) convolutions in the f16x3 precision mode, as a streaming GEMM.
//
// A 1x1 conv has no spatial reuse: per output pixel it reads Cin and writes Cout values once, so at these
// channel counts it is bound by memory traffic and per-workgroup overheads, not by the matrix pipe
// (DESIGN.md section 8). This kernel is therefore laid out as a stream:
//   - the weights of the workgroup's M tile (all of K, hi/lo fp16 planes) are staged into LDS ONCE; after one
//     barrier the workgroup never synchronises again,
//   - workgroups are persistent: each wave walks pixel groups (RPW x 16 consecutive pixels of the flattened
//     image) with a grid-wide stride,
//   - the pixel operand goes straight from global memory into MFMA B fragments: lane (i, g) loads 8 consecutive
//     channels of pixel i (two float4), i.e. every pixel contributes whole 128-byte lines per K-step; the fp32
//     values get the fused input activation, are split hi/lo in registers and feed
//     v_mfma_f32_16x16x32_f16 three times (lo*hi + hi*lo + hi*hi), next K-step's loads already in flight,
//   - the shared fused epilogue (bias / activation / residual / scale / pixel-shuffle) stores float4s.
// K order: a K-step of 32 = two 16-channel chunks of the (virtually concatenated) inputs; lane group g>>1
// picks the chunk, g&1 the 8-channel half -- the same fragment convention as conv_f16x3_kernel.h, so the
// host-side weight layout [chunk16][m][16] is shared (KH = KW = 1).
#pragma once
#include "conv_f16x3_kernel.h"

namespace lssvc {

constexpr int kPwMaxLds = 64 * 1024;      // two workgroups per CU
constexpr int kPwBigLds = 144 * 1024;     // one workgroup per CU: taken when it buys a larger M tile (fewer re-reads of X)

template <int MF, int RPW, bool GDN = false>
__global__ __launch_bounds__(256, RPW == 1 ? 4 : 2) void conv_pw_f16x3_kernel(const ConvP p) {
    constexpr int TM = 16 * MF;
    extern __shared__ __attribute__((aligned(16))) _Float16 wlds[];     // [plane hi|lo][chunk16 (padded to even)][TM][16]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 15;
    const int lg = lane >> 4;
    const int tsel = lg >> 1;
    const int ch8 = (lg & 1) * 8;

    const int nchunk = p.n_chunks16;                  // 16-channel chunks over all input segments
    const int nstep = (nchunk + 1) >> 1;
    const int nslot = nstep * 2;
    const int m_tile = blockIdx.x % p.m_tiles;
    const int m0 = m_tile * TM;
    const int plane = nslot * TM * CK16;              // elements per LDS plane

    // ---- stage this M tile's weights once --------------------------------------------------------------
    {
        const _Float16 *g_h = reinterpret_cast<const _Float16 *>(p.w16);
        const _Float16 *g_l = g_h + p.w16_plane;
        const int items = nslot * TM * 2;             // 16-byte items per plane
        for (int idx = tid; idx < items; idx += 256) {
            const int c = idx / (TM * 2);
            const int r = idx - c * (TM * 2);
            const int m = r >> 1, half = r & 1;
            f16x8 h = {0, 0, 0, 0, 0, 0, 0, 0}, l = {0, 0, 0, 0, 0, 0, 0, 0};
            if (c < nchunk && m0 + m < p.M_pad) {
                const size_t o = ((size_t)c * p.M_pad + m0 + m) * CK16 + half * 8;
                h = *reinterpret_cast<const f16x8 *>(g_h + o);
                l = *reinterpret_cast<const f16x8 *>(g_l + o);
            }
            const int d = (c * TM + m) * CK16 + half * 8;
            *reinterpret_cast<f16x8 *>(wlds + d) = h;
            *reinterpret_cast<f16x8 *>(wlds + plane + d) = l;
        }
    }
    __syncthreads();

    const bool sq = p.in_act == LSSVC_INACT_SQUARE;
    const float in_slope = p.in_act == LSSVC_INACT_LRELU ? p.in_slope : 1.0f;
    const long long npix = (long long)p.Hout * p.Wout;
    const long long ngroups = (npix + 16 * RPW - 1) / (16 * RPW);
    const long long wave_id = (long long)(blockIdx.x / p.m_tiles) * 4 + wave;
    const long long wave_stride = (long long)(gridDim.x / p.m_tiles) * 4;

    // chunk -> (segment base pointer, pixel pitch, first channel, channels left) for this lane group's chunk of step s
    auto chunk_src = [&](int c, const float *&base, int &ld, int &c0, int &left) {
        int seg = 0, first = 0;
#pragma unroll
        for (int i = 0; i < LSSVC_CONV_MAX_INPUTS - 1; ++i) {
            const int n = (p.in[seg].C + 15) >> 4;
            if (seg < p.n_in - 1 && c >= first + n) {
                first += n;
                ++seg;
            }
        }
        base = p.in[seg].p;
        ld = p.in[seg].ld;
        c0 = (c - first) * 16;
        left = p.in[seg].C - c0;
    };

    for (long long grp = wave_id; grp < ngroups; grp += wave_stride) {
        long long pix[RPW];
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const long long q = (grp * RPW + r) * 16 + li;
            pix[r] = q < npix ? q : -1;
        }
        f32x4 acc[MF][RPW];
#pragma unroll
        for (int a = 0; a < MF; ++a)
#pragma unroll
            for (int b = 0; b < RPW; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

        float4 raw[2][RPW][2];
        int left_of[2];
        auto load_step = [&](int s, float4 (&dst)[RPW][2], int &left8) {
            const int c = 2 * s + tsel;
            const float *base; int ld, c0, left;
            chunk_src(c < nchunk ? c : nchunk - 1, base, ld, c0, left);
            left8 = c < nchunk ? left - ch8 : 0;                    // channels available from this lane's first channel on
            const int cc = left8 > 0 ? c0 + ch8 : 0;
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const float *src = base + (size_t)(pix[r] >= 0 ? pix[r] : 0) * ld + cc;
                dst[r][0] = *reinterpret_cast<const float4 *>(src);
                dst[r][1] = *reinterpret_cast<const float4 *>(src + (left8 > 4 ? 4 : 0));
            }
        };
        auto compute_step = [&](int s, const float4 (&src)[RPW][2], int left8) {
            f16x8 bh[RPW], bl[RPW];
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                float v[8] = {src[r][0].x, src[r][0].y, src[r][0].z, src[r][0].w, src[r][1].x, src[r][1].y, src[r][1].z, src[r][1].w};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float x = (pix[r] >= 0 && j < left8) ? v[j] : 0.f;     // left8 is a multiple of 4 (inputs are 4-channel aligned)
                    x *= sq ? x : (x > 0.f ? 1.0f : in_slope);
                    x = fminf(fmaxf(x, -65504.f), 65504.f);
                    const _Float16 h = (_Float16)x;
                    bh[r][j] = h;
                    bl[r][j] = (_Float16)(x - (float)h);
                }
            }
            const int slot = 2 * s + tsel;
            f16x8 ah[MF], al[MF];
#pragma unroll
            for (int f = 0; f < MF; ++f) {
                const int o = (slot * TM + f * 16 + li) * CK16 + ch8;
                ah[f] = *reinterpret_cast<const f16x8 *>(wlds + o);
                al[f] = *reinterpret_cast<const f16x8 *>(wlds + plane + o);
            }
            // three passes over the MF x RPW accumulators so consecutive MFMAs never share an accumulator
#pragma unroll
            for (int f = 0; f < MF; ++f)
#pragma unroll
                for (int r = 0; r < RPW; ++r) acc[f][r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[f], bh[r], acc[f][r], 0, 0, 0);
#pragma unroll
            for (int f = 0; f < MF; ++f)
#pragma unroll
                for (int r = 0; r < RPW; ++r) acc[f][r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[f], bl[r], acc[f][r], 0, 0, 0);
#pragma unroll
            for (int f = 0; f < MF; ++f)
#pragma unroll
                for (int r = 0; r < RPW; ++r) acc[f][r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[f], bh[r], acc[f][r], 0, 0, 0);
        };
        load_step(0, raw[0], left_of[0]);
        for (int s = 0; s < nstep; s += 2) {          // two steps per trip so the register buffers have static names
            if (s + 1 < nstep) load_step(s + 1, raw[1], left_of[1]);
            compute_step(s, raw[0], left_of[0]);
            if (s + 1 < nstep) {
                if (s + 2 < nstep) load_step(s + 2, raw[0], left_of[0]);
                compute_step(s + 1, raw[1], left_of[1]);
            }
        }
        conv_unscale<MF, RPW>(p, acc);
        conv_epilogue_flat<MF, RPW, GDN>(p, acc, pix, [&](int r, int col) { const long long q = (grp * RPW + r) * 16 + col; return q < npix ? q : -1LL; }, m0, lg);
    }
}

template <int MF, int RPW, bool GDN = false>
static int launch_pw_f16x3(const ConvP &p, hipStream_t st) {
    ConvP q = p;
    q.m_tiles = (p.M_pad / 16 + MF - 1) / MF;
    const int nslot = ((p.n_chunks16 + 1) / 2) * 2;
    const size_t lds = (size_t)2 * nslot * 16 * MF * CK16 * sizeof(_Float16);
    if (lds > (size_t)kPwBigLds) return fail("conv2d(pw f16x3): %zu bytes of weights do not fit LDS", lds);
    static LdsGrant grant;
    if (grant.ensure(reinterpret_cast<const void *>(conv_pw_f16x3_kernel<MF, RPW, GDN>), lds, 64 * 1024)) return 1;
    static const int per_cu = [] {
        int v = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, conv_pw_f16x3_kernel<MF, RPW, GDN>, 256, 0) != hipSuccess || v < 1) v = 1;
        return v;
    }();
    const int resident = per_cu * device_cus();
    const long long npix = (long long)p.Hout * p.Wout;
    const long long ngroups = (npix + 16 * RPW - 1) / (16 * RPW);
    long long per_m = (ngroups + 3) / 4;                                    // workgroups that have work, per M tile
    long long res = resident;
    if (lds > (size_t)kPwMaxLds) res = resident / 2 > 0 ? resident / 2 : 1;     // one workgroup per CU fits
    const long long cap = res / q.m_tiles > 0 ? res / q.m_tiles : 1;
    if (per_m > cap) per_m = cap;
    const long long blocks = per_m * q.m_tiles;
    hipLaunchKernelGGL((conv_pw_f16x3_kernel<MF, RPW, GDN>), dim3((unsigned)blocks), dim3(256), lds, st, q);
    return launch_status("conv2d(pw f16x3)");
}


// ---- deep-K variant (K >= 128): the K loop above keeps one step in flight and its conditionals make the compiler wait
// vmcnt(0) before every load group, so with 6-24 MFMAs per 32-channel step it is bound by global-load latency
// (1024->384 @72x120: 34 TFLOP/s). Here the loop body is straight-line: D = 4 register sets form a ring, step s+D is
// loaded right after step s has been consumed (addresses past the end are clamped, their B operand zeroed by a
// select, so there is no tail branch), the input LeakyReLU is max(x, s*x), and the single-input case (every FFN
// contraction) gets its own instantiation without the segment lookup.
template <int MF, int RPW, bool MULTI>
__global__ __launch_bounds__(256, 2) void conv_pwk_f16x3_kernel(const ConvP p) {
    constexpr int TM = 16 * MF;
    constexpr int D = MF >= 4 ? 2 : 4;       // ring depth: 16 * D registers; MF = 4 already holds 32 accumulators + 64 fragment registers
    extern __shared__ __attribute__((aligned(16))) _Float16 wlds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 15;
    const int lg = lane >> 4;
    const int tsel = lg >> 1;
    const int ch8 = (lg & 1) * 8;
    const int nchunk = p.n_chunks16;
    const int nstep = (nchunk + 1) >> 1;
    const int nslot = nstep * 2;
    const int m_tile = blockIdx.x % p.m_tiles;
    const int m0 = m_tile * TM;
    const int plane = nslot * TM * CK16;
    {
        const _Float16 *g_h = reinterpret_cast<const _Float16 *>(p.w16);
        const _Float16 *g_l = g_h + p.w16_plane;
        const int items = nslot * TM * 2;
        for (int idx = tid; idx < items; idx += 256) {
            const int c = idx / (TM * 2);
            const int r = idx - c * (TM * 2);
            const int m = r >> 1, half = r & 1;
            f16x8 h = {0, 0, 0, 0, 0, 0, 0, 0}, l = {0, 0, 0, 0, 0, 0, 0, 0};
            if (c < nchunk && m0 + m < p.M_pad) {
                const size_t o = ((size_t)c * p.M_pad + m0 + m) * CK16 + half * 8;
                h = *reinterpret_cast<const f16x8 *>(g_h + o);
                l = *reinterpret_cast<const f16x8 *>(g_l + o);
            }
            const int d = (c * TM + m) * CK16 + half * 8;
            *reinterpret_cast<f16x8 *>(wlds + d) = h;
            *reinterpret_cast<f16x8 *>(wlds + plane + d) = l;
        }
    }
    __syncthreads();

    const float in_slope = p.in_act == LSSVC_INACT_LRELU ? p.in_slope : 1.0f;
    const long long npix = (long long)p.Hout * p.Wout;
    const long long ngroups = (npix + 16 * RPW - 1) / (16 * RPW);
    const long long wave_id = (long long)(blockIdx.x / p.m_tiles) * 4 + wave;
    const long long wave_stride = (long long)(gridDim.x / p.m_tiles) * 4;
    const int n0 = (p.in[0].C + 15) >> 4, n1 = p.n_in > 1 ? (p.in[1].C + 15) >> 4 : 0;
    const int nstep_pad = (nstep + D - 1) / D * D;

    for (long long grp = wave_id; grp < ngroups; grp += wave_stride) {
        long long pix[RPW];
        size_t poff[RPW];
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const long long q = (grp * RPW + r) * 16 + li;
            pix[r] = q < npix ? q : -1;
            poff[r] = (size_t)(q < npix ? q : 0);
        }
        f32x4 acc[MF][RPW];
#pragma unroll
        for (int a = 0; a < MF; ++a)
#pragma unroll
            for (int b = 0; b < RPW; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

        float4 raw[D][RPW][2];
        int left_of[D];
        auto load_step = [&](int s, float4 (&dst)[RPW][2], int &left8) {
            int c = 2 * s + tsel;
            const bool in_range = c < nchunk;
            c = in_range ? c : nchunk - 1;
            const float *base = p.in[0].p;
            int ld = p.in[0].ld, cfirst = 0, cseg = p.in[0].C;
            if (MULTI) {                                   // selects, not branches
                const bool s1 = c >= n0, s2 = c >= n0 + n1;
                base = s2 ? p.in[2].p : (s1 ? p.in[1].p : base);
                ld = s2 ? p.in[2].ld : (s1 ? p.in[1].ld : ld);
                cseg = s2 ? p.in[2].C : (s1 ? p.in[1].C : cseg);
                cfirst = s2 ? n0 + n1 : (s1 ? n0 : 0);
            }
            const int c0 = (c - cfirst) * 16;
            const int avail = cseg - c0 - ch8;             // channels from this lane's first channel to the end of the segment
            left8 = in_range ? avail : 0;
            const int cc = avail > 0 ? c0 + ch8 : 0;
            const int second = avail > 4 ? 4 : 0;
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const float *src = base + poff[r] * ld + cc;
                dst[r][0] = *reinterpret_cast<const float4 *>(src);
                dst[r][1] = *reinterpret_cast<const float4 *>(src + second);
            }
        };
        auto compute_step = [&](int s, const float4 (&src)[RPW][2], int left8) {
            f16x8 bh[RPW], bl[RPW];
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const float v[8] = {src[r][0].x, src[r][0].y, src[r][0].z, src[r][0].w, src[r][1].x, src[r][1].y, src[r][1].z, src[r][1].w};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float x = (pix[r] >= 0 && j < left8) ? v[j] : 0.f;
                    x = fmaxf(x, in_slope * x);
                    x = fminf(fmaxf(x, -65504.f), 65504.f);
                    const _Float16 h = (_Float16)x;
                    bh[r][j] = h;
                    bl[r][j] = (_Float16)(x - (float)h);
                }
            }
            const int sc = s < nstep ? s : nstep - 1;      // steps of the padded tail re-read the last slot against B = 0
            const int slot = 2 * sc + tsel;
            f16x8 ah[MF], al[MF];
#pragma unroll
            for (int f = 0; f < MF; ++f) {
                const int o = (slot * TM + f * 16 + li) * CK16 + ch8;
                ah[f] = *reinterpret_cast<const f16x8 *>(wlds + o);
                al[f] = *reinterpret_cast<const f16x8 *>(wlds + plane + o);
            }
#pragma unroll
            for (int f = 0; f < MF; ++f)
#pragma unroll
                for (int r = 0; r < RPW; ++r) acc[f][r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[f], bh[r], acc[f][r], 0, 0, 0);
#pragma unroll
            for (int f = 0; f < MF; ++f)
#pragma unroll
                for (int r = 0; r < RPW; ++r) acc[f][r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[f], bl[r], acc[f][r], 0, 0, 0);
#pragma unroll
            for (int f = 0; f < MF; ++f)
#pragma unroll
                for (int r = 0; r < RPW; ++r) acc[f][r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[f], bh[r], acc[f][r], 0, 0, 0);
        };
#pragma unroll
        for (int d = 0; d < D; ++d) load_step(d, raw[d], left_of[d]);
        for (int s = 0; s < nstep_pad; s += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                compute_step(s + d, raw[d], (s + d) < nstep ? left_of[d] : 0);
                load_step(s + d + D, raw[d], left_of[d]);
            }
        }
        conv_unscale<MF, RPW>(p, acc);
        conv_epilogue_flat<MF, RPW, false>(p, acc, pix, [&](int r, int col) { const long long q = (grp * RPW + r) * 16 + col; return q < npix ? q : -1LL; }, m0, lg);
    }
}

template <int MF, int RPW>
static int launch_pwk_f16x3(const ConvP &p, hipStream_t st) {
    ConvP q = p;
    q.m_tiles = (p.M_pad / 16 + MF - 1) / MF;
    const int nslot = ((p.n_chunks16 + 1) / 2) * 2;
    const size_t lds = (size_t)2 * nslot * 16 * MF * CK16 * sizeof(_Float16);
    if (lds > (size_t)kPwMaxLds) return fail("conv2d(pwk f16x3): %zu bytes of weights do not fit LDS", lds);
    const int cus = device_cus();
    const long long npix = (long long)p.Hout * p.Wout;
    const long long ngroups = (npix + 16 * RPW - 1) / (16 * RPW);
    long long per_m = (ngroups + 3) / 4;
    const long long res = 2LL * cus;
    const long long cap = res / q.m_tiles > 0 ? res / q.m_tiles : 1;
    if (per_m > cap) per_m = cap;
    const long long blocks = per_m * q.m_tiles;
    if (p.n_in > 1) hipLaunchKernelGGL((conv_pwk_f16x3_kernel<MF, RPW, true>), dim3((unsigned)blocks), dim3(256), lds, st, q);
    else hipLaunchKernelGGL((conv_pwk_f16x3_kernel<MF, RPW, false>), dim3((unsigned)blocks), dim3(256), lds, st, q);
    return launch_status("conv2d(pwk f16x3)");
}


// ---- K-sliced variant for large K on small maps (e.g. 1024 -> 384 @72x120): with all of K resident only a 16-row M tile
// fits LDS, so X is re-read M/16 times and those layers are L2-bound. Here the M tile keeps its full 16*MF rows and
// K is cut into slices of 256 channels (8 K-steps, 16*MF KiB of hi/lo weights): the workgroup re-stages the weight
// slice every 8 steps (two barriers), while each wave owns ONE 64-pixel group whose MF x 4 accumulators live in
// registers across all slices; the X ring prefetch runs straight through the slice boundaries. One group per wave,
// no persistence: meant for launches with few pixel groups.
template <int MF, bool MULTI, int RPW = 4>
__global__ __launch_bounds__(256, 2) void conv_pwks_f16x3_kernel(const ConvP p) {
    constexpr int TM = 16 * MF, D = 2, SLICE_STEPS = 8, SLICE_SLOTS = 2 * SLICE_STEPS;      // D: X steps in flight per wave (rings of 4 and 8 measured 7-25 % slower, profiles/r04_pwks_ring_ab.txt)
    extern __shared__ __attribute__((aligned(16))) _Float16 wlds[];     // [hi|lo][slot in slice][TM][16]
    constexpr int plane = SLICE_SLOTS * TM * CK16;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 15;
    const int lg = lane >> 4;
    const int tsel = lg >> 1;
    const int ch8 = (lg & 1) * 8;
    const int nchunk = p.n_chunks16;
    const int nstep = (nchunk + 1) >> 1;
    const int m_tile = blockIdx.x % p.m_tiles;
    const int m0 = m_tile * TM;
    const float in_slope = p.in_act == LSSVC_INACT_LRELU ? p.in_slope : 1.0f;
    const long long npix = (long long)p.Hout * p.Wout;
    const long long grp = (long long)(blockIdx.x / p.m_tiles) * 4 + wave;          // this wave's 64-pixel group (may be past the end)
    const int n0 = (p.in[0].C + 15) >> 4, n1 = p.n_in > 1 ? (p.in[1].C + 15) >> 4 : 0;
    const int nstep_pad = (nstep + SLICE_STEPS - 1) / SLICE_STEPS * SLICE_STEPS;

    long long pix[RPW];
    size_t poff[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const long long q = (grp * RPW + r) * 16 + li;
        pix[r] = q < npix ? q : -1;
        poff[r] = (size_t)(q < npix ? q : 0);
    }
    f32x4 acc[MF][RPW];
#pragma unroll
    for (int a = 0; a < MF; ++a)
#pragma unroll
        for (int b = 0; b < RPW; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto stage_slice = [&](int s0) {                       // weights of K-steps s0 .. s0+7 for this M tile
        const _Float16 *g_h = reinterpret_cast<const _Float16 *>(p.w16);
        const _Float16 *g_l = g_h + p.w16_plane;
        // SLICE_SLOTS * TM * 2 sixteen-byte items per plane = TM / 8 per thread, in batches of four whose eight loads go out
        // together: as a run-time loop (load, load, wait, store, store per item) this was eight serial L2 round trips per slice,
        // most of what a launch on a 72x120 map took
        constexpr int NIT = SLICE_SLOTS * TM * 2 / 256, NB = NIT < 4 ? NIT : 4;
        static_assert(SLICE_SLOTS * TM * 2 % 256 == 0, "slice items per thread");
        if constexpr (RPW >= 4 && MF >= 4) {               // (the 64-pixel MF = 4 variant is at its register limit: item by item, as before)
            for (int idx = tid; idx < SLICE_SLOTS * TM * 2; idx += 256) {
                const int c = idx / (TM * 2);
                const int r = idx - c * (TM * 2);
                const int m = r >> 1, half = r & 1;
                const int cg = 2 * s0 + c;
                f16x8 h = {0, 0, 0, 0, 0, 0, 0, 0}, l = {0, 0, 0, 0, 0, 0, 0, 0};
                if (cg < nchunk && m0 + m < p.M_pad) {
                    const size_t o = ((size_t)cg * p.M_pad + m0 + m) * CK16 + half * 8;
                    h = *reinterpret_cast<const f16x8 *>(g_h + o);
                    l = *reinterpret_cast<const f16x8 *>(g_l + o);
                }
                const int d = (c * TM + m) * CK16 + half * 8;
                *reinterpret_cast<f16x8 *>(wlds + d) = h;
                *reinterpret_cast<f16x8 *>(wlds + plane + d) = l;
            }
            return;
        }
#pragma unroll
        for (int i0 = 0; i0 < NIT; i0 += NB) {
            f16x8 h[NB], l[NB];
            bool ok[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int idx = tid + (i0 + j) * 256;
                const int c = idx / (TM * 2);
                const int r = idx - c * (TM * 2);
                const int m = r >> 1, half = r & 1;
                const int cg = 2 * s0 + c;
                ok[j] = i0 + j < NIT && cg < nchunk && m0 + m < p.M_pad;
                const size_t o = ok[j] ? ((size_t)cg * p.M_pad + m0 + m) * CK16 + half * 8 : 0;
                h[j] = *reinterpret_cast<const f16x8 *>(g_h + o);
                l[j] = *reinterpret_cast<const f16x8 *>(g_l + o);
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                if (i0 + j >= NIT) break;
                const int idx = tid + (i0 + j) * 256;
                const int c = idx / (TM * 2);
                const int r = idx - c * (TM * 2);
                const int m = r >> 1, half = r & 1;
                const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                const int d = (c * TM + m) * CK16 + half * 8;
                *reinterpret_cast<f16x8 *>(wlds + d) = ok[j] ? h[j] : z;
                *reinterpret_cast<f16x8 *>(wlds + plane + d) = ok[j] ? l[j] : z;
            }
        }
    };
    float4 raw[D][RPW][2];
    int left_of[D];
    auto load_step = [&](int s, float4 (&dst)[RPW][2], int &left8) {
        int c = 2 * s + tsel;
        const bool in_range = c < nchunk;
        c = in_range ? c : nchunk - 1;
        const float *base = p.in[0].p;
        int ld = p.in[0].ld, cfirst = 0, cseg = p.in[0].C;
        if (MULTI) {
            const bool s1 = c >= n0, s2 = c >= n0 + n1;
            base = s2 ? p.in[2].p : (s1 ? p.in[1].p : base);
            ld = s2 ? p.in[2].ld : (s1 ? p.in[1].ld : ld);
            cseg = s2 ? p.in[2].C : (s1 ? p.in[1].C : cseg);
            cfirst = s2 ? n0 + n1 : (s1 ? n0 : 0);
        }
        const int c0 = (c - cfirst) * 16;
        const int avail = cseg - c0 - ch8;
        left8 = in_range ? avail : 0;
        const int cc = avail > 0 ? c0 + ch8 : 0;
        const int second = avail > 4 ? 4 : 0;
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const float *src = base + poff[r] * ld + cc;
            dst[r][0] = *reinterpret_cast<const float4 *>(src);
            dst[r][1] = *reinterpret_cast<const float4 *>(src + second);
        }
    };
    auto compute_step = [&](int slot_step, const float4 (&src)[RPW][2], int left8) {
        f16x8 bh[RPW], bl[RPW];
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const float v[8] = {src[r][0].x, src[r][0].y, src[r][0].z, src[r][0].w, src[r][1].x, src[r][1].y, src[r][1].z, src[r][1].w};
            if (p.debug & 64) {                                   // ablation (results WRONG): no fp32 -> fp16 hi / lo conversion
                bh[r] = __builtin_bit_cast(f16x8, src[r][0]);
                bl[r] = __builtin_bit_cast(f16x8, src[r][1]);
                continue;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float x = (pix[r] >= 0 && j < left8) ? v[j] : 0.f;
                x = fmaxf(x, in_slope * x);
                x = fminf(fmaxf(x, -65504.f), 65504.f);
                const _Float16 h = (_Float16)x;
                bh[r][j] = h;
                bl[r][j] = (_Float16)(x - (float)h);
            }
        }
        const int slot = 2 * slot_step + tsel;
        f16x8 ah[MF], al[MF];
#pragma unroll
        for (int f = 0; f < MF; ++f) {
            const int o = (slot * TM + f * 16 + li) * CK16 + ch8;
            ah[f] = *reinterpret_cast<const f16x8 *>(wlds + o);
            al[f] = *reinterpret_cast<const f16x8 *>(wlds + plane + o);
        }
#pragma unroll
        for (int f = 0; f < MF; ++f)
#pragma unroll
            for (int r = 0; r < RPW; ++r) acc[f][r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[f], bh[r], acc[f][r], 0, 0, 0);
#pragma unroll
        for (int f = 0; f < MF; ++f)
#pragma unroll
            for (int r = 0; r < RPW; ++r) acc[f][r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[f], bl[r], acc[f][r], 0, 0, 0);
#pragma unroll
        for (int f = 0; f < MF; ++f)
#pragma unroll
            for (int r = 0; r < RPW; ++r) acc[f][r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[f], bh[r], acc[f][r], 0, 0, 0);
    };
#pragma unroll
    for (int d = 0; d < D; ++d) load_step(d, raw[d], left_of[d]);
    for (int s0 = 0; s0 < nstep_pad; s0 += SLICE_STEPS) {
        __syncthreads();                                   // every wave is done with the previous slice
        stage_slice(s0);
        __syncthreads();
#pragma unroll
        for (int u = 0; u < SLICE_STEPS; u += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int s = s0 + u + d;
                compute_step(u + d, raw[d], s < nstep ? left_of[d] : 0);      // padded-tail slots hold zero weights
                load_step(s + D, raw[d], left_of[d]);
            }
        }
    }
    conv_unscale<MF, RPW>(p, acc);
    conv_epilogue_flat<MF, RPW, false>(p, acc, pix, [&](int r, int col) { const long long q = (grp * RPW + r) * 16 + col; return q < npix ? q : -1LL; }, m0, lg);
}

template <int MF>
static int launch_pwks_f16x3(const ConvP &p, hipStream_t st) {
    ConvP q = p;
    q.m_tiles = (p.M_pad / 16 + MF - 1) / MF;
    const size_t lds = (size_t)2 * 16 * 16 * MF * CK16 * sizeof(_Float16);          // 16 KiB per M fragment
    const long long npix = (long long)p.Hout * p.Wout;
    // 64-pixel groups (4 accumulator rows per wave) unless that leaves most of the chip without a second workgroup per CU to
    // hide the X loads and the re-staging barriers behind: then 32-pixel groups, twice the workgroups (the weight slices are
    // re-read from L2 twice as often; at these sizes they are a few MB)
    static const int small_on = getenv("LSSVC_PWKS_SMALL") ? atoi(getenv("LSSVC_PWKS_SMALL")) : 1;
    const long long blocks64 = (((npix + 63) / 64 + 3) / 4) * q.m_tiles;
    const bool small = small_on && blocks64 < 2LL * device_cus();
    const long long ngroups = small ? (npix + 31) / 32 : (npix + 63) / 64;
    const long long blocks = ((ngroups + 3) / 4) * q.m_tiles;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return fail("conv2d(pwks f16x3): bad grid %lld", blocks);
    if (small) {
        if (p.n_in > 1) hipLaunchKernelGGL((conv_pwks_f16x3_kernel<MF, true, 2>), dim3((unsigned)blocks), dim3(256), lds, st, q);
        else hipLaunchKernelGGL((conv_pwks_f16x3_kernel<MF, false, 2>), dim3((unsigned)blocks), dim3(256), lds, st, q);
    } else {
        if (p.n_in > 1) hipLaunchKernelGGL((conv_pwks_f16x3_kernel<MF, true, 4>), dim3((unsigned)blocks), dim3(256), lds, st, q);
        else hipLaunchKernelGGL((conv_pwks_f16x3_kernel<MF, false, 4>), dim3((unsigned)blocks), dim3(256), lds, st, q);
    }
    return launch_status("conv2d(pwks f16x3)");
}


// ---- "all-M" variant for small K (<= 64 input channels, i.e. <= 2 K-steps) whose whole weight matrix fits LDS:
// the wave converts its pixel fragments ONCE, keeps them in registers and walks every M tile itself (so X is
// read and split once instead of once per M tile, and all output channels of a pixel are written by one wave),
// while the raw loads of its next pixel group are already in flight.
template <int MF, int RPW, bool GDN = false>
__global__ __launch_bounds__(256, 2) void conv_pw_allm_f16x3_kernel(const ConvP p) {
    constexpr int TM = 16 * MF;
    constexpr int NS = 2;                                                   // K-steps held in registers
    extern __shared__ __attribute__((aligned(16))) _Float16 wlds[];     // [plane][slot][M_all][16]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 15;
    const int lg = lane >> 4;
    const int tsel = lg >> 1;
    const int ch8 = (lg & 1) * 8;
    const int nchunk = p.n_chunks16;
    const int nstep = (nchunk + 1) >> 1;                                    // 1 or 2
    const int nslot = nstep * 2;
    const int m_all = p.m_tiles * TM;                                       // rows held in LDS (>= M_pad)
    const int plane = nslot * m_all * CK16;
    {
        const _Float16 *g_h = reinterpret_cast<const _Float16 *>(p.w16);
        const _Float16 *g_l = g_h + p.w16_plane;
        const int items = nslot * m_all * 2;
        for (int idx = tid; idx < items; idx += 256) {
            const int c = idx / (m_all * 2);
            const int r = idx - c * (m_all * 2);
            const int m = r >> 1, half = r & 1;
            f16x8 h = {0, 0, 0, 0, 0, 0, 0, 0}, l = {0, 0, 0, 0, 0, 0, 0, 0};
            if (c < nchunk && m < p.M_pad) {
                const size_t o = ((size_t)c * p.M_pad + m) * CK16 + half * 8;
                h = *reinterpret_cast<const f16x8 *>(g_h + o);
                l = *reinterpret_cast<const f16x8 *>(g_l + o);
            }
            const int d = (c * m_all + m) * CK16 + half * 8;
            *reinterpret_cast<f16x8 *>(wlds + d) = h;
            *reinterpret_cast<f16x8 *>(wlds + plane + d) = l;
        }
    }
    __syncthreads();

    const bool sq = p.in_act == LSSVC_INACT_SQUARE;
    const float in_slope = p.in_act == LSSVC_INACT_LRELU ? p.in_slope : 1.0f;
    const long long npix = (long long)p.Hout * p.Wout;
    const long long ngroups = (npix + 16 * RPW - 1) / (16 * RPW);
    const long long wave_id = (long long)blockIdx.x * 4 + wave;
    const long long wave_stride = (long long)gridDim.x * 4;

    // this lane group's source chunk per K-step is the same for every pixel group: resolve it once
    const float *base[NS];
    int ld[NS], cc[NS], left8[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int c = 2 * s + tsel;
        int seg = 0, first = 0;
#pragma unroll
        for (int i = 0; i < LSSVC_CONV_MAX_INPUTS - 1; ++i) {
            const int n = (p.in[seg].C + 15) >> 4;
            if (seg < p.n_in - 1 && c >= first + n) {
                first += n;
                ++seg;
            }
        }
        const int c0 = (c - first) * 16;
        left8[s] = (s < nstep && c < nchunk) ? p.in[seg].C - c0 - ch8 : 0;
        base[s] = p.in[seg].p;
        ld[s] = p.in[seg].ld;
        cc[s] = left8[s] > 0 ? c0 + ch8 : 0;
    }

    float4 raw[NS][RPW][2];
    auto load_group = [&](long long grp) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                long long q = (grp * RPW + r) * 16 + li;
                if (q >= npix) q = 0;
                const float *src = base[s] + (size_t)q * ld[s] + cc[s];
                raw[s][r][0] = *reinterpret_cast<const float4 *>(src);
                raw[s][r][1] = *reinterpret_cast<const float4 *>(src + (left8[s] > 4 ? 4 : 0));
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
        f16x8 bh[NS][RPW], bl[NS][RPW];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const float v[8] = {raw[s][r][0].x, raw[s][r][0].y, raw[s][r][0].z, raw[s][r][0].w,
                                    raw[s][r][1].x, raw[s][r][1].y, raw[s][r][1].z, raw[s][r][1].w};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float x = (pix[r] >= 0 && j < left8[s]) ? v[j] : 0.f;
                    x *= sq ? x : (x > 0.f ? 1.0f : in_slope);
                    x = fminf(fmaxf(x, -65504.f), 65504.f);
                    const _Float16 h = (_Float16)x;
                    bh[s][r][j] = h;
                    bl[s][r][j] = (_Float16)(x - (float)h);
                }
            }
        if (grp + wave_stride < ngroups) load_group(grp + wave_stride);      // in flight during the whole M loop
        for (int mt = 0; mt < p.m_tiles; ++mt) {
            f32x4 acc[MF][RPW];
#pragma unroll
            for (int a = 0; a < MF; ++a)
#pragma unroll
                for (int b = 0; b < RPW; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (s < nstep) {
                    f16x8 ah[MF], al[MF];
#pragma unroll
                    for (int f = 0; f < MF; ++f) {
                        const int o = ((2 * s + tsel) * m_all + mt * TM + f * 16 + li) * CK16 + ch8;
                        ah[f] = *reinterpret_cast<const f16x8 *>(wlds + o);
                        al[f] = *reinterpret_cast<const f16x8 *>(wlds + plane + o);
                    }
#pragma unroll
                    for (int f = 0; f < MF; ++f)
#pragma unroll
                        for (int r = 0; r < RPW; ++r) acc[f][r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[f], bh[s][r], acc[f][r], 0, 0, 0);
#pragma unroll
                    for (int f = 0; f < MF; ++f)
#pragma unroll
                        for (int r = 0; r < RPW; ++r) acc[f][r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[f], bl[s][r], acc[f][r], 0, 0, 0);
#pragma unroll
                    for (int f = 0; f < MF; ++f)
#pragma unroll
                        for (int r = 0; r < RPW; ++r) acc[f][r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[f], bh[s][r], acc[f][r], 0, 0, 0);
                }
            }
            conv_unscale<MF, RPW>(p, acc);
            conv_epilogue_flat<MF, RPW, GDN>(p, acc, pix, [&](int r, int col) { const long long q = (grp * RPW + r) * 16 + col; return q < npix ? q : -1LL; }, mt * TM, lg);
        }
    }
}

template <int MF, int RPW, bool GDN = false>
static int launch_pw_allm_f16x3(const ConvP &p, hipStream_t st) {
    ConvP q = p;
    q.m_tiles = (p.M_pad / 16 + MF - 1) / MF;
    const int nslot = ((p.n_chunks16 + 1) / 2) * 2;
    const size_t lds = (size_t)2 * nslot * q.m_tiles * 16 * MF * CK16 * sizeof(_Float16);
    if (lds > (size_t)kPwMaxLds || p.n_chunks16 > 4) return fail("conv2d(pw all-M f16x3): shape does not fit");
    static const int per_cu = [] {
        int v = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, conv_pw_allm_f16x3_kernel<MF, RPW, GDN>, 256, kPwMaxLds) != hipSuccess || v < 1) v = 1;
        return v;
    }();
    const int resident = per_cu * device_cus();
    const long long npix = (long long)p.Hout * p.Wout;
    const long long ngroups = (npix + 16 * RPW - 1) / (16 * RPW);
    long long blocks = (ngroups + 3) / 4;
    // LDS use varies per launch; size the persistent grid by what the actual footprint admits (2 per CU at most by registers)
    long long cap = resident;
    const long long by_lds = (long long)(160 * 1024 / (lds > 1024 ? lds : 1024)) * 256;
    if (cap > by_lds) cap = by_lds;
    if (cap > 2 * 256) cap = 2 * 256;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL((conv_pw_allm_f16x3_kernel<MF, RPW, GDN>), dim3((unsigned)blocks), dim3(256), lds, st, q);
    return launch_status("conv2d(pw all-M f16x3)");
}

int dispatch_pw_f16x3(const ConvP &p, hipStream_t st, char *kernel_name);

}  // namespace lssvc
