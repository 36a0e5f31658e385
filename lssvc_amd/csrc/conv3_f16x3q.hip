// conv3_f16x3q.hip -- 3x3 stride-1 convolution in the f16x3 mode: PERSISTENT, PING-PONG wave groups.
//
// Successor of the producer / consumer kernel (conv3_f16x3p.hip). What round 1 measured on that one (DESIGN.md
// section 10): matrix pipe 53 % busy; the four producer waves spend ~650 VALU instructions per phase on addressing
// and fp32 -> fp16 hi/lo conversion against ~720 vector-issue slots the consumer's MFMAs leave free on their SIMD
// (a 16x16x32 MFMA holds the issue port 8 of its 16 cycles), so producers and consumers finish a phase neck and neck
// and every jitter stalls the matrix pipe; and once per tile all four consumers stop together for the epilogue
// (16 % of the time) while the producers -- and their half of the register file -- sit idle.
//
// Here every wave is a consumer. One workgroup of 8 waves per CU works on a PAIR of vertically adjacent
// (4*RPW) x 16-pixel tiles (same 16*MF output channels): waves 0-3 (group A, one per SIMD) own the upper tile, waves
// 4-7 (group B, the second wave of each SIMD) the lower one, and the two groups run half a phase apart:
//
//     slot 2k    : A computes phase k   (MFMA + ds_read_b128 only)   | B stages its patch of phase k (+ its epilogue)
//     slot 2k+1  : A stages phase k+1 (+ epilogue), DMAs W(k+1)      | B computes phase k
//
// so each SIMD always has one wave in the matrix pipe and its partner in the vector / memory pipes:
//   * the epilogue of one group overlaps the MFMAs of the other (it no longer costs matrix-pipe time),
//   * staging is done by the waves that own the registers anyway, with per-TILE hoisted addressing (the patch geometry
//     and the weight-DMA lane offsets are computed when the tile changes, not per phase): ~45 VALU per staged float4
//     item become ~30,
//   * the weights of a phase are fetched once (LDS-DMA, double-buffered) and used by BOTH tiles,
//   * the accumulators of both groups are live, i.e. the CU holds a 48 x 16 x 64 output block in registers.
// LDS: one patch image per group (single-buffered: a group writes it only in the slot in which it does not compute)
// + two weight images = 2 x 30 KB + 2 x 37 KB = 134 KB at MF = 4, as before. Two workgroup barriers per phase.
//
// Arithmetic: f16x3_step_pair for taps (0,1) (2,3) (4,5) (6,7) and f16x3_step_odd for tap 8, the same K order and
// accumulator layout as conv_f16x3_kernel, the same fused epilogue: results are bit-identical to the tiled kernel
// (tests/test_gpu_bench_kernels.py pins that).
#include "conv_f16x3_kernel.h"

namespace lssvc {

#ifndef LSSVC_P3_RPW
#define LSSVC_P3_RPW 6
#endif
constexpr int kQThreads = 512;
constexpr int kQGroup = 256;             // threads per wave group

template <int MF>
struct QGeom {
    static constexpr int RPW = LSSVC_P3_RPW, HALF = RPW / 2, TH = RPW * 4, TM = 16 * MF;
    static constexpr int PH = TH + 2, PW = 18, NTAP = 9;
    static constexpr int PATCH_HALFS = PH * PW * CK16;            // per plane
    static constexpr int PATCH_ITEMS = PH * PW * 4;               // float4 items
    static constexpr int NP = (PATCH_ITEMS + kQGroup - 1) / kQGroup;
    static constexpr int W_HALFS = NTAP * TM * CK16;              // per plane
    static constexpr int W_ITEMS = NTAP * TM * 2;                 // 16-byte items per plane
    static constexpr int W_INSTR = 2 * W_ITEMS / 64;              // wave-level DMA instructions for both planes (= 9 MF)
    static constexpr int NDMA = (W_INSTR + 3) / 4;                // per wave of group A
    static constexpr int LDS_BYTES = (2 * 2 * PATCH_HALFS + 2 * 2 * W_HALFS) * 2;
};

template <int MF, bool INACT>
__global__ __launch_bounds__(kQThreads, 1) void conv3_f16x3q_kernel(const ConvP p) {
    using G = QGeom<MF>;
    constexpr int RPW = G::RPW, TM = G::TM, PW = G::PW, NTAP = G::NTAP, NP = G::NP, NDMA = G::NDMA;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16 *const patch0 = reinterpret_cast<_Float16 *>(smem);                    // [group][plane][PH*PW][16]
    _Float16 *const wts0 = patch0 + 4 * G::PATCH_HALFS;                             // [buf][plane][tap][m][16]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int grp = wave >> 2;                 // 0: group A (upper tile, weight DMA), 1: group B (lower tile)
    const int cw = wave & 3;                   // wave inside the group: pixel rows cw*RPW .. of the group's tile
    const int lt = tid & (kQGroup - 1);

    // ---- this workgroup's tile pairs: the XCD it runs on owns a contiguous range, its workgroups interleave inside it
    const int pairs_y = (p.tiles_y + 1) >> 1;
    const int npairs = p.tiles_x * pairs_y * p.m_tiles;
    const int nx = (int)gridDim.x < 8 ? (int)gridDim.x : 8;
    const int xcd = blockIdx.x % nx, kb = blockIdx.x / nx;
    const int nb_x = ((int)gridDim.x - xcd + nx - 1) / nx;
    const int tq = npairs / nx, tr = npairs % nx;
    const int t_begin = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
    const int t_cnt = tq + (xcd < tr ? 1 : 0);
    const int n_it = kb < t_cnt ? (t_cnt - kb + nb_x - 1) / nb_x : 0;
    if (n_it == 0) return;
    const int ppt = p.n_chunks16;              // phases per tile
    const int total = n_it * ppt;

    auto tile_origin = [&](int it, int &oy0, int &ox0, int &m0) {
        const int pair = t_begin + kb + it * nb_x;
        const int mt = pair % p.m_tiles, pt = pair / p.m_tiles;
        const int tx = pt % p.tiles_x, py = pt / p.tiles_x;
        oy0 = (2 * py + grp) * G::TH;          // a lower tile past the image computes zeros and stores nothing
        ox0 = tx * 16;
        m0 = mt * TM;
    };

    const int li = lane & 15;
    const int lg = lane >> 4;
    const int tsel = lg >> 1;
    const int ch8 = (lg & 1) * 8;
    const int quad4 = (lt & 3) * 4;
    const float in_slope = p.in_slope;
    const int Hin = p.in[0].H, Win = p.in[0].W;
    const _Float16 *const w16 = reinterpret_cast<const _Float16 *>(p.w16);
    _Float16 *const my_patch_h = patch0 + grp * 2 * G::PATCH_HALFS;
    _Float16 *const my_patch_l = my_patch_h + G::PATCH_HALFS;

    // ---- staging state (the "producer side" of this wave): next phase to stage, per-tile hoisted geometry
    int s_it = 0;                              // tile of the next phase to stage
    KState s_k{0, 0, 0, 0};                    // its segment / channel offset / global chunk
    int ppix[NP];                              // input pixel index of each staged float4 item, -1 = zero padding / past the patch
    int woff[NDMA];                            // group A: element offset of each weight-DMA lane inside one chunk's [hi | lo] image
    float4 preg[NP];
    unsigned live = 0;

    auto tile_geometry = [&](int it) {
        int oy0, ox0, m0;
        tile_origin(it, oy0, ox0, m0);
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int idx = lt + i * kQGroup;
            const int pix = idx >> 2;
            const int py = pix / PW, px = pix - py * PW;
            const int gy = oy0 - p.pad_t + py, gx = ox0 - p.pad_l + px;
            const bool ok = idx < G::PATCH_ITEMS && gy >= 0 && gy < Hin && gx >= 0 && gx < Win;
            ppix[i] = ok ? gy * Win + gx : -1;
        }
        if (grp == 0) {
#pragma unroll
            for (int t = 0; t < NDMA; ++t) {
                int j = cw + 4 * t;                                  // wave-uniform DMA instruction index
                if (j >= G::W_INSTR) j = G::W_INSTR - 1;             // surplus slots rewrite the last KiB with the same bytes
                const int i = j * 64 + lane;                         // 16-byte item of the [hi plane | lo plane] image
                const int plane = i >= G::W_ITEMS ? 1 : 0;
                const int r = i - plane * G::W_ITEMS;
                const int tap = r / (2 * TM);
                const int rr = r - tap * 2 * TM;
                int m = m0 + (rr >> 1);
                if (m >= p.M_pad) m = p.M_pad - 1;                   // rows past M_pad: any finite weights, masked by the epilogue
                woff[t] = plane * (int)p.w16_plane + (tap * p.M_pad + m) * CK16 + (rr & 1) * 8;
            }
        }
    };
    // first half of staging phase (s_it, s_k): weight DMA (group A) + the global loads of the patch
    auto stage_issue = [&](int wbuf) {
        if (grp == 0 && !(p.debug & 1)) {
            unsigned char *dst = reinterpret_cast<unsigned char *>(wts0 + wbuf * 2 * G::W_HALFS);
            const _Float16 *src0 = w16 + (size_t)s_k.kc * NTAP * p.M_pad * CK16;
#pragma unroll
            for (int t = 0; t < NDMA; ++t) {
                int j = cw + 4 * t;
                if (j >= G::W_INSTR) j = G::W_INSTR - 1;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src0 + woff[t]),
                                                 (__attribute__((address_space(3))) void *)(dst + j * 1024), 16, 0, 0);
            }
        }
        const V X = p.in[s_k.seg];
        const bool cvalid = quad4 < X.C - s_k.c0;
        const int cc = cvalid ? s_k.c0 + quad4 : 0;
        live = 0;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const bool ok = ppix[i] >= 0 && cvalid;
            const size_t off = ok ? (size_t)ppix[i] * X.ld + cc : 0;
            preg[i] = *reinterpret_cast<const float4 *>(X.p + off);
            live |= ok ? (1u << i) : 0u;
        }
    };
    // second half: activation, hi/lo split, LDS stores; then advance the staging cursor
    auto stage_finish = [&]() {
        if (!(p.debug & 2)) {
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int idx = lt + i * kQGroup;
                const bool on = (live >> i) & 1u;
                const float raw[4] = {preg[i].x, preg[i].y, preg[i].z, preg[i].w};
                f16x4 h, l;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float x = on ? raw[j] : 0.f;
                    if (INACT) x = fmaxf(x, in_slope * x);                    // LeakyReLU for 0 <= slope <= 1: exact
                    x = fminf(fmaxf(x, -65504.f), 65504.f);
                    h[j] = (_Float16)x;
                    l[j] = (_Float16)(x - (float)h[j]);
                }
                const int o = (idx >> 2) * CK16 + quad4;
                if (i + 1 < NP || idx < G::PATCH_ITEMS) {
                    *reinterpret_cast<f16x4 *>(my_patch_h + o) = h;
                    *reinterpret_cast<f16x4 *>(my_patch_l + o) = l;
                }
            }
        }
        s_k.c0 += CK16;
        ++s_k.kc;
        if (s_k.c0 >= p.in[s_k.seg].C) {
            s_k.c0 = 0;
            ++s_k.seg;
            if (s_k.seg >= p.n_in) {
                s_k = KState{0, 0, 0, 0};
                ++s_it;
                if (s_it < n_it) tile_geometry(s_it);
            }
        }
    };

    // ---- compute state
    f32x4 acc[MF][RPW];
#pragma unroll
    for (int a = 0; a < MF; ++a)
#pragma unroll
        for (int b = 0; b < RPW; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    int c_it = 0, c_kt = 0;
    bool epi_pending = false;
    int epi_it = 0;

    auto compute = [&](int wbuf) {
        const _Float16 *ph_ = my_patch_h, *pl_ = my_patch_l;
        const _Float16 *wh_ = wts0 + wbuf * 2 * G::W_HALFS;
        const _Float16 *wl_ = wh_ + G::W_HALFS;
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const bool odd = u == 4;                                         // tap 8
            const int tap = odd ? 8 : 2 * u + tsel;
            const int ky = tap / 3, kx = tap - ky * 3;
            const _Float16 *wa1 = (odd && !tsel) ? wl_ : wh_, *wa2 = (odd && !tsel) ? wh_ : wl_;
            const _Float16 *pb1 = (odd && tsel) ? pl_ : ph_;
            f16x8 a1[MF], a2[MF];
#pragma unroll
            for (int f = 0; f < MF; ++f) {
                const int o = (tap * TM + f * 16 + li) * CK16 + ch8;
                a1[f] = *reinterpret_cast<const f16x8 *>(wa1 + o);
                a2[f] = *reinterpret_cast<const f16x8 *>(wa2 + o);
            }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                f16x8 b1[G::HALF], b2[G::HALF];
#pragma unroll
                for (int r = 0; r < G::HALF; ++r) {
                    const int row = cw * RPW + half * G::HALF + r;
                    const int o = ((row + ky) * PW + li + kx) * CK16 + ch8;
                    b1[r] = *reinterpret_cast<const f16x8 *>(pb1 + o);
                    if (!odd) b2[r] = *reinterpret_cast<const f16x8 *>(pl_ + o);
                }
                if (odd) {
#pragma unroll
                    for (int f = 0; f < MF; ++f)
#pragma unroll
                        for (int r = 0; r < G::HALF; ++r)
                            acc[f][half * G::HALF + r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[f], b1[r], acc[f][half * G::HALF + r], 0, 0, 0);
#pragma unroll
                    for (int f = 0; f < MF; ++f)
#pragma unroll
                        for (int r = 0; r < G::HALF; ++r)
                            acc[f][half * G::HALF + r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2[f], b1[r], acc[f][half * G::HALF + r], 0, 0, 0);
                } else {
#pragma unroll
                    for (int f = 0; f < MF; ++f)
#pragma unroll
                        for (int r = 0; r < G::HALF; ++r)
                            acc[f][half * G::HALF + r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2[f], b1[r], acc[f][half * G::HALF + r], 0, 0, 0);
#pragma unroll
                    for (int f = 0; f < MF; ++f)
#pragma unroll
                        for (int r = 0; r < G::HALF; ++r)
                            acc[f][half * G::HALF + r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[f], b2[r], acc[f][half * G::HALF + r], 0, 0, 0);
#pragma unroll
                    for (int f = 0; f < MF; ++f)
#pragma unroll
                        for (int r = 0; r < G::HALF; ++r)
                            acc[f][half * G::HALF + r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[f], b1[r], acc[f][half * G::HALF + r], 0, 0, 0);
                }
            }
        }
        if (++c_kt == ppt) {
            c_kt = 0;
            epi_pending = true;
            epi_it = c_it++;
        }
    };
    auto epilogue = [&]() {
        int oy0, ox0, m0;
        tile_origin(epi_it, oy0, ox0, m0);
        if (!(p.debug & 32)) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                f32x4 part[MF][G::HALF];
#pragma unroll
                for (int f = 0; f < MF; ++f)
#pragma unroll
                    for (int r = 0; r < G::HALF; ++r) part[f][r] = acc[f][half * G::HALF + r];
                long long pix[G::HALF];
#pragma unroll
                for (int r = 0; r < G::HALF; ++r) {
                    const int oy = oy0 + cw * RPW + half * G::HALF + r, ox = ox0 + li;
                    pix[r] = (oy < p.Hout && ox < p.Wout) ? (long long)oy * p.Wout + ox : -1;
                }
                conv_epilogue_fast<MF, G::HALF>(p, part, pix, m0, lg, p.w16_unscale);       // the dispatcher only sends p.fast_epi convs here
            }
        }
#pragma unroll
        for (int a = 0; a < MF; ++a)
#pragma unroll
            for (int b = 0; b < RPW; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        epi_pending = false;
    };

    // LDS-DMA data is ordered for other waves' ds_reads only by the ISSUING wave's vmcnt wait followed by a barrier.
    auto slot_barrier = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };

    tile_geometry(0);
    if (grp == 0) {
        stage_issue(0);
        stage_finish();
    }
    slot_barrier();                                          // A's patch of phase 0 and W(0) are in LDS
    if (grp == 0) {
        for (int k = 0; k < total; ++k) {
            compute(k & 1);                                  // slot 2k
            slot_barrier();
            const bool more = k + 1 < total;                 // slot 2k+1: stage phase k+1 around this tile's epilogue
            if (more) stage_issue((k + 1) & 1);
            if (epi_pending) epilogue();
            if (more) stage_finish();
            slot_barrier();
        }
    } else {
        for (int k = 0; k < total; ++k) {
            stage_issue(0);                                  // slot 2k: stage phase k around the previous tile's epilogue
            if (epi_pending) epilogue();
            stage_finish();
            slot_barrier();
            compute(k & 1);                                  // slot 2k+1
            slot_barrier();
        }
        if (epi_pending) epilogue();
    }
}

template <int MF, bool INACT>
static int launch_q(const ConvP &p, hipStream_t st) {
    using G = QGeom<MF>;
    const int cus = device_cus();
    static LdsGrant grant;
    if (grant.ensure(reinterpret_cast<const void *>(conv3_f16x3q_kernel<MF, INACT>), G::LDS_BYTES)) return 1;
    ConvP q = p;
    q.tiles_x = (p.Wout + 15) / 16;
    q.tiles_y = (p.Hout + G::TH - 1) / G::TH;
    q.m_tiles = (p.M_pad / 16 + MF - 1) / MF;
    const long long npairs = (long long)q.tiles_x * ((q.tiles_y + 1) / 2) * q.m_tiles;
    if (npairs <= 0 || npairs > 0x7fffffffLL) return fail("conv2d(f16x3q): bad tile count %lld", npairs);
    if (p.w16_plane * 2 > 0x7fffffffLL) return fail("conv2d(f16x3q): weight image too large for 32-bit lane offsets");
    long long blocks = cus;                       // one persistent 8-wave workgroup per CU
    if (blocks > npairs) blocks = npairs;
    hipLaunchKernelGGL((conv3_f16x3q_kernel<MF, INACT>), dim3((unsigned)blocks), dim3(kQThreads), G::LDS_BYTES, st, q);
    return launch_status("conv2d(f16x3q)");
}

static int q_pick_mf(int frags) {
    if (frags > 4 && frags % 4 != 0 && frags % 3 == 0) return 3;        // e.g. 96 = 2 x 48 rather than 64 + 32
    return frags >= 4 ? 4 : frags;
}

int dispatch_conv3_f16x3q(const ConvP &p, hipStream_t st, char *kernel_name) {
    const int mf = q_pick_mf(p.M_pad / 16);
    const bool inact = p.in_act == LSSVC_INACT_LRELU;
    snprintf(kernel_name, 96, "conv3_f16x3q_kernel<%d, %s>", mf, inact ? "true" : "false");
#define LSSVC_Q_CASE(m) \
    if (mf == m) return inact ? launch_q<m, true>(p, st) : launch_q<m, false>(p, st);
    LSSVC_Q_CASE(4) LSSVC_Q_CASE(3) LSSVC_Q_CASE(2) LSSVC_Q_CASE(1)
#undef LSSVC_Q_CASE
    return fail("conv2d(f16x3q): no kernel for MF=%d", mf);
}

}  // namespace lssvc
