// conv3_f16x3d.hip -- 3x3 stride-1 convolution in the f16x3 mode: persistent, warp-specialised (producer / consumer
// waves as conv3_f16x3p.hip), with the tile EPILOGUE DEFERRED into the next tile's MFMA stream.
//
// Why. In-kernel stamps on the 24x16-tile kernel (tools/p3_stamps.py, 64->64 @1152x1920, 1.75 GHz): a consumer wave
// spends 69 % of its time in the MFMA phases, 9 % waiting at phase barriers and 21 % in the epilogue -- 7.9 k cycles
// per tile for ~100 VALU instructions and 24 stores. The stores are the cost: a CU retires global stores at ~12 B/clk
// (its share of the HBM write bandwidth; every CU reaches its epilogue at about the same time), the 96 KB of a tile take
// 96 KB / 12 B/clk = 7.9 k cycles to ISSUE, the wave sits in the store queue, and its SIMD's matrix pipe idles.
// Making the epilogue's instruction stream shorter (straight-line interior path, bias in LDS, activation / residual
// specialisation) changed nothing measurable: it is not instruction-bound.
//
// Here a finished tile's accumulators are moved to a second register set (`old`) and the consumer goes straight on
// with the next tile; the stores of `old` are dealt out over that tile's phases, one 16-byte store per lane after each
// tap-pair MFMA step: a row per phase = 4 KB per CU per ~900 cycles = 4.5 B/clk, well under what the store path
// sustains, so the store queue never backs up into the MFMA stream and the VALU part of the epilogue (~14 instructions
// per store) rides in the issue slots the MFMAs leave free.
// Two register sets of accumulators only fit with 4 pixel rows per consumer wave (2 x 64 registers at MF = 4) instead
// of 6: tile = 16 x 16 pixels x 16 MF channels. That costs ~4 % (more halo, weight DMA and barriers per MFMA;
// measured in round 1: RPW 6 -> 4 = -1.7 % on the whole benchmark) and buys back the 21 %.
//
// Flush schedule (static, so that every register index is a compile-time constant): with P = phases per tile
// (16-channel chunks of Cin), RP = 1 row per phase if P >= 4, else 2 (P >= 2; convs with Cin <= 16 stay on
// conv3_f16x3p); phase q of the next tile flushes rows q RP .. q RP + RP - 1, their RP x MF (row, fragment) items
// spread over the four flush points behind tap-pair steps 0..3. Whatever is still pending when a workgroup runs out of
// tiles is flushed at once.
// Arithmetic, K order, accumulator layout and the per-item epilogue arithmetic are those of conv_f16x3_kernel /
// conv_epilogue_fast: results are bit-identical to the tiled kernel (tests/test_gpu_bench_kernels.py).
#include <type_traits>

#include "conv_f16x3_kernel.h"

namespace lssvc {

#ifndef LSSVC_D_SCHED_MASK
#define LSSVC_D_SCHED_MASK 0x0108
#endif
constexpr int kDThreads = 512;
constexpr int kDConsumers = 4;           // waves 0..3
constexpr int kDProducerThreads = kDThreads - 64 * kDConsumers;

template <int MF>
struct DGeom {
    static constexpr int RPW = 4, HALF = 2, TH = RPW * kDConsumers, TM = 16 * MF;
    static constexpr int PH = TH + 2, PW = 18, NTAP = 9, NSTEP = 5;
    static constexpr int PATCH_HALFS = PH * PW * CK16;            // per plane
    static constexpr int PATCH_ITEMS = PH * PW * 4;               // float4 items
    static constexpr int NP = (PATCH_ITEMS + kDProducerThreads - 1) / kDProducerThreads;
    static constexpr int W_HALFS = NTAP * TM * CK16;              // per plane
    static constexpr int W_ITEMS = NTAP * TM * 2;                 // 16-byte items per plane
    static constexpr int W_INSTR = 2 * W_ITEMS / 64;              // wave-level DMA instructions for both planes (= 9 MF)
    static constexpr int NPROD = kDProducerThreads / 64;
    static constexpr int NDMA = (W_INSTR + NPROD - 1) / NPROD;
    static constexpr int LDS_BYTES = 2 * (2 * PATCH_HALFS + 2 * W_HALFS) * 2;      // + the bias vector (launch_d)
};

struct DPhase {
    int it;        // index into this workgroup's tile sequence
    KState k;      // segment / channel offset / global chunk index of the phase
};

template <int MF, bool INACT, bool RES>
__global__ __launch_bounds__(kDThreads, 1) void conv3_f16x3d_kernel(const ConvP p) {
    using G = DGeom<MF>;
    using P3Phase = DPhase;
    constexpr int RPW = G::RPW, TM = G::TM, PW = G::PW, NTAP = G::NTAP, NP = G::NP;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16 *const patch0 = reinterpret_cast<_Float16 *>(smem);                    // [buf][plane][PH*PW][16]
    _Float16 *const wts0 = patch0 + 4 * G::PATCH_HALFS;                             // [buf][plane][tap][m][16]
    float *const bias_s = reinterpret_cast<float *>(wts0 + 4 * G::W_HALFS);         // [m_tiles * TM], zero past M_pad

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    // ---- this workgroup's tiles: the XCD it runs on owns a contiguous range, its workgroups interleave inside it
    const int ntiles = p.tiles_x * p.tiles_y * p.m_tiles;
    const int nx = (int)gridDim.x < 8 ? (int)gridDim.x : 8;
    const int xcd = blockIdx.x % nx, kb = blockIdx.x / nx;
    const int nb_x = ((int)gridDim.x - xcd + nx - 1) / nx;
    const int tq = ntiles / nx, tr = ntiles % nx;
    const int t_begin = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
    const int t_cnt = tq + (xcd < tr ? 1 : 0);
    const int n_it = kb < t_cnt ? (t_cnt - kb + nb_x - 1) / nb_x : 0;
    if (n_it == 0) return;
    const int phases_per_tile = p.n_chunks16;
    for (int i = tid; i < p.m_tiles * TM; i += kDThreads) bias_s[i] = (p.bias && i < p.M_pad) ? p.bias[i] : 0.f;   // visible after barrier (A)

    auto tile_origin = [&](int it, int &oy0, int &ox0, int &m0) {
        const int tile = t_begin + kb + it * nb_x;
        const int mt = tile % p.m_tiles, pt = tile / p.m_tiles;
        const int tx = pt % p.tiles_x, ty = pt / p.tiles_x;
        oy0 = ty * G::TH;
        ox0 = tx * 16;
        m0 = mt * TM;
    };

    if (wave >= kDConsumers) {
        // =================================================================================== PRODUCER waves
        const int lt = tid - 64 * kDConsumers;                   // 0 .. 255
        const int pw = wave - kDConsumers;
        const int quad4 = (lt & 3) * 4;
        const float in_slope = p.in_slope;
        const int Hin = p.in[0].H, Win = p.in[0].W;
        const _Float16 *w16 = reinterpret_cast<const _Float16 *>(p.w16);

        auto next_phase = [&](P3Phase ph) {
            ph.k.c0 += CK16;
            ++ph.k.kc;
            if (ph.k.c0 >= p.in[ph.k.seg].C) {
                ph.k.c0 = 0;
                ++ph.k.seg;
                if (ph.k.seg >= p.n_in) {
                    ph.k = KState{0, 0, 0, 0};
                    ++ph.it;
                }
            }
            return ph;
        };
        // Per-TILE staging geometry, computed when the tile changes instead of every phase (the producers' vector-issue
        // slots are what they compete for with the consumers' MFMAs): input pixel of every staged float4 item (-1 = zero
        // padding / past the patch) and the lane offsets of the weight DMA inside one chunk's [hi | lo] image.
        int ppix[NP], woff[G::NDMA];
        int geom_it = -1;
        auto tile_geometry = [&](int it) {
            int oy0, ox0, m0;
            tile_origin(it, oy0, ox0, m0);
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int idx = lt + i * kDProducerThreads;
                const int pix = idx >> 2;
                const int py = pix / PW, px = pix - py * PW;
                const int gy = oy0 - p.pad_t + py, gx = ox0 - p.pad_l + px;
                const bool ok = idx < G::PATCH_ITEMS && gy >= 0 && gy < Hin && gx >= 0 && gx < Win;
                ppix[i] = ok ? gy * Win + gx : -1;
            }
#pragma unroll
            for (int t = 0; t < G::NDMA; ++t) {
                int j = pw + G::NPROD * t;                       // wave-uniform DMA instruction index
                if (j >= G::W_INSTR) j = G::W_INSTR - 1;         // surplus slots rewrite the last KiB with the same bytes
                const int i = j * 64 + lane;                     // 16-byte item of the [hi plane | lo plane] image
                const int plane = i >= G::W_ITEMS ? 1 : 0;
                const int r = i - plane * G::W_ITEMS;
                const int tap = r / (2 * TM);
                const int rr = r - tap * 2 * TM;
                int m = m0 + (rr >> 1);
                if (m >= p.M_pad) m = p.M_pad - 1;               // rows past M_pad: any finite weights, masked by the epilogue
                woff[t] = plane * (int)p.w16_plane + (tap * p.M_pad + m) * CK16 + (rr & 1) * 8;
            }
            geom_it = it;
        };
        // everything one phase needs, into LDS buffer `buf`: weights by DMA, patch through registers
        auto fill = [&](const P3Phase &ph, int buf) {
            if (ph.it != geom_it) tile_geometry(ph.it);
            if (!(p.debug & 1)) {
                unsigned char *dst = reinterpret_cast<unsigned char *>(wts0 + buf * 2 * G::W_HALFS);
                const _Float16 *src0 = w16 + (size_t)ph.k.kc * NTAP * p.M_pad * CK16;
#pragma unroll
                for (int t = 0; t < G::NDMA; ++t) {
                    int j = pw + G::NPROD * t;
                    if (j >= G::W_INSTR) j = G::W_INSTR - 1;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src0 + woff[t]),
                                                     (__attribute__((address_space(3))) void *)(dst + j * 1024), 16, 0, 0);
                }
            }
            if (p.debug & 2) return;
            const V X = p.in[ph.k.seg];
            const bool cvalid = quad4 < X.C - ph.k.c0;
            const int cc = cvalid ? ph.k.c0 + quad4 : 0;
            float4 preg[NP];
            unsigned pmask = 0;
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const bool ok = ppix[i] >= 0 && cvalid;
                const size_t off = ok ? (size_t)ppix[i] * X.ld + cc : 0;
                preg[i] = *reinterpret_cast<const float4 *>(X.p + off);
                pmask |= ok ? (1u << i) : 0u;
            }
            _Float16 *ph_ = patch0 + buf * 2 * G::PATCH_HALFS;
            _Float16 *pl_ = ph_ + G::PATCH_HALFS;
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int idx = lt + i * kDProducerThreads;
                const bool live = (pmask >> i) & 1u;
                const float raw[4] = {preg[i].x, preg[i].y, preg[i].z, preg[i].w};
                f16x4 h, l;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float x = live ? raw[j] : 0.f;
                    if (INACT) x = fmaxf(x, in_slope * x);                    // LeakyReLU for 0 <= slope <= 1: exact
                    x = fminf(fmaxf(x, -65504.f), 65504.f);
                    h[j] = (_Float16)x;
                    l[j] = (_Float16)(x - (float)h[j]);
                }
                const int o = (idx >> 2) * CK16 + quad4;
                if (i + 1 < NP || idx < G::PATCH_ITEMS) {
                    *reinterpret_cast<f16x4 *>(ph_ + o) = h;
                    *reinterpret_cast<f16x4 *>(pl_ + o) = l;
                }
            }
        };

        P3Phase ph{0, KState{0, 0, 0, 0}};
        fill(ph, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the weight DMA of this wave has landed (made explicit, see below)
        __syncthreads();                                   // (A) phase 0 is in buffer 0
        const int total = n_it * phases_per_tile;
        for (int k = 0; k < total; ++k) {
            if (k + 1 < total) {
                ph = next_phase(ph);
                fill(ph, (k + 1) & 1);                     // the consumers read buffer k & 1 meanwhile
            }
            // LDS-DMA data is ordered for the consumers' ds_reads only by the issuing wave's vmcnt wait + a barrier;
            // the compiler emits that wait today, this line makes it a property of the source
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                               // (B_k) buffer (k+1)&1 complete, buffer k&1 released
        }
        return;
    }

    // ======================================================================================= CONSUMER waves
    const int li = lane & 15;
    const int lg = lane >> 4;
    const int tsel = lg >> 1;
    const int ch8 = (lg & 1) * 8;

    f32x4 acc[MF][RPW], old[MF][RPW];
#pragma unroll
    for (int a = 0; a < MF; ++a)
#pragma unroll
        for (int b = 0; b < RPW; ++b) {
            acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
            old[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    bool pending = false;                                  // `old` holds a finished tile that is not (fully) stored yet
    int e_oy0 = 0, e_ox0 = 0, e_m0 = 0;                    // its origin

    // ---- the deferred epilogue, one (row, fragment) item = 4 channels of one pixel per lane. BRANCH-FREE on purpose: a phase
    // is one basic block (the compiler overlaps the fragment reads of a step with the MFMAs of the previous one only inside a
    // block), so lanes outside the image / past Cout store to a per-thread trash slot instead of being branched around, the
    // activation is the max(v, s v) form for every conv, and a conv without a residual is a separate instantiation (RES).
    // Addressing is split so that little stays live across a phase: per pending TILE the channel part of every fragment's
    // address (f_off, with the pixel-shuffle sub-pixel folded in: the divisions happen once per tile), per PHASE the pixel
    // part of its two rows (row_off), per item one 64-bit add.
    const float unscale = p.w16_unscale;
    const float s_neg = p.act == LSSVC_ACT_LRELU ? p.slope : (p.act == LSSVC_ACT_RELU ? 0.0f : 1.0f);
    const bool shuffle = p.fast_epi == 2;
    float *const trash = p.gdn_x.p + 1024 + ((size_t)blockIdx.x * kDThreads + tid) * 4;     // launch_d: scratch behind 4 KB of zeros
    int f_off[MF];                                         // element offset of fragment f's 4 channels inside a pixel (block)
    int f_mb[MF];                                          // its first channel (bias / residual index)
    unsigned f_ok = 0;                                     // bit f: fragment below Cout
    auto set_tile = [&](int it_) {
        tile_origin(it_, e_oy0, e_ox0, e_m0);
        f_ok = 0;
#pragma unroll
        for (int f = 0; f < MF; ++f) {
            const int mb = e_m0 + f * 16 + 4 * lg;
            f_mb[f] = mb;
            if (mb < p.Cout) f_ok |= 1u << f;
            if (shuffle) {                                 // m = q * cps + c -> sub-pixel q = dy * 2 + dx, channel c
                const int cps = p.Cout >> 2;
                const int q = mb / cps, c = mb - q * cps;
                f_off[f] = ((q >> 1) * p.out.W + (q & 1)) * p.out.ld + c;
            } else {
                f_off[f] = mb;
            }
        }
    };
    struct Row {
        size_t out_off, res_off;                           // element offset of the pixel (plain) / its 2x2 block (shuffle)
        bool ok;
    };
    auto row_of = [&](int row_abs) {
        Row r;
        const int oy = e_oy0 + wave * RPW + row_abs, ox = e_ox0 + li;
        r.ok = oy < p.Hout && ox < p.Wout;
        const size_t pix = r.ok ? (size_t)oy * p.Wout + ox : 0;
        r.res_off = pix * (RES ? p.res.ld : 0);
        r.out_off = shuffle ? ((size_t)(2 * (r.ok ? oy : 0)) * p.out.W + 2 * (r.ok ? ox : 0)) * p.out.ld : pix * p.out.ld;
        return r;
    };
    auto load_res = [&](const Row &r, int f) {
        const bool ok = r.ok && ((f_ok >> f) & 1u);
        return *reinterpret_cast<const float4 *>(p.res.p + r.res_off + (ok ? f_mb[f] : 0));
    };
    auto flush_item = [&](const f32x4 &a, const Row &r, int f, const float4 &rs) {
        const f32x4 bv = *reinterpret_cast<__attribute__((address_space(3))) const f32x4 *>((lds_cfloat_ptr)bias_s + f_mb[f]);
        const f32x2 us = {unscale, unscale}, sn = {s_neg, s_neg};
        f32x2 v0 = f32x2{a[0], a[1]} * us + f32x2{bv[0], bv[1]};
        f32x2 v1 = f32x2{a[2], a[3]} * us + f32x2{bv[2], bv[3]};
        const f32x2 n0 = v0 * sn, n1 = v1 * sn;
        v0 = f32x2{fmaxf(v0.x, n0.x), fmaxf(v0.y, n0.y)};
        v1 = f32x2{fmaxf(v1.x, n1.x), fmaxf(v1.y, n1.y)};
        if (RES) {
            v0 = v0 + f32x2{rs.x, rs.y};
            v1 = v1 + f32x2{rs.z, rs.w};
        }
        const bool ok = r.ok && ((f_ok >> f) & 1u);
        float *dst = ok ? p.out.p + r.out_off + f_off[f] : trash;
        *reinterpret_cast<float4 *>(dst) = make_float4(v0.x, v0.y, v1.x, v1.y);
    };
    // Flush schedule: the pending tile's four rows go out during phases 0 and 1 of the tile that follows it, two rows =
    // 2 MF items per phase, spread over the four flush points behind the tap-pair steps. `old` is rotated by two rows after
    // phase 0, so the items of a phase always sit in old[.][0..1]: every register index is a compile-time constant.
    constexpr int UNITS = 2 * MF, UPP = (UNITS + 3) / 4;

    auto phase_body = [&](auto flush_tag, int kt, const _Float16 *ph_, const _Float16 *pl_, const _Float16 *wh_, const _Float16 *wl_) {
        constexpr bool FLUSH = decltype(flush_tag)::value;
        float4 rs_next[UPP];
        Row rows[2];
        if (FLUSH) {
            rows[0] = row_of(2 * kt);
            rows[1] = row_of(2 * kt + 1);
        }
        if (FLUSH && RES) {
#pragma unroll
            for (int i = 0; i < UPP; ++i) rs_next[i] = load_res(rows[i / MF], i % MF);
        }
#pragma unroll
        for (int u = 0; u < G::NSTEP; ++u) {
            const bool odd = 2 * u + 1 >= NTAP;                              // the ninth tap: f16x3_step_odd
            const int tap = odd ? 2 * u : 2 * u + tsel;
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
                    const int row = wave * RPW + half * G::HALF + r;
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
            if (FLUSH && u < 4) {                                            // flush point u: items u UPP .. u UPP + UPP - 1
                // keep the item code where it is written: MFMAs and LDS fragment reads may move across (0x8 | 0x100), the
                // epilogue's VALU / VMEM may not, or the scheduler hoists all eight items to the top of the phase and spills
                __builtin_amdgcn_sched_barrier(LSSVC_D_SCHED_MASK);
                float4 rs_cur[UPP];
#pragma unroll
                for (int i = 0; i < UPP; ++i) rs_cur[i] = RES ? rs_next[i] : make_float4(0.f, 0.f, 0.f, 0.f);
                if (RES && u < 3) {                                          // next point's residuals first: they must be OLDER than
#pragma unroll                                                               // this point's stores (vmcnt retires in order)
                    for (int i = 0; i < UPP; ++i) {
                        const int j = (u + 1) * UPP + i;
                        if (j < UNITS) rs_next[i] = load_res(rows[j / MF], j % MF);
                    }
                }
#pragma unroll
                for (int i = 0; i < UPP; ++i) {
                    const int j = u * UPP + i;
                    if (j < UNITS) flush_item(old[j % MF][j / MF], rows[j / MF], j % MF, rs_cur[i]);
                }
                __builtin_amdgcn_sched_barrier(LSSVC_D_SCHED_MASK);
            }
        }
    };

    __syncthreads();                                       // (A)
    int it = 0, kt = 0;                                    // tile index in this workgroup's sequence, phase inside the tile
    const int total = n_it * phases_per_tile;
    for (int k = 0; k < total; ++k) {
        const int buf = k & 1;
        const _Float16 *ph_ = patch0 + buf * 2 * G::PATCH_HALFS;
        const _Float16 *pl_ = ph_ + G::PATCH_HALFS;
        const _Float16 *wh_ = wts0 + buf * 2 * G::W_HALFS;
        const _Float16 *wl_ = wh_ + G::W_HALFS;
        if (pending && kt < 2 && !(p.debug & 32)) {
            phase_body(std::true_type{}, kt, ph_, pl_, wh_, wl_);
#pragma unroll
            for (int a = 0; a < MF; ++a) {                 // rotate: rows 2, 3 become the next phase's rows 0, 1
                old[a][0] = old[a][2];
                old[a][1] = old[a][3];
            }
            if (kt == 1) pending = false;
        } else {
            phase_body(std::false_type{}, kt, ph_, pl_, wh_, wl_);
        }
        __syncthreads();                                   // (B_k)
        if (++kt == phases_per_tile) {                     // tile finished: its results become the pending set
#pragma unroll
            for (int a = 0; a < MF; ++a)
#pragma unroll
                for (int b = 0; b < RPW; ++b) {
                    old[a][b] = acc[a][b];
                    acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            set_tile(it);
            pending = true;
            kt = 0;
            ++it;
        }
    }
    if (pending && !(p.debug & 32)) {                      // the workgroup's last tile: flush it now
#pragma unroll
        for (int row = 0; row < RPW; ++row) {
            const Row r = row_of(row);
#pragma unroll
            for (int f = 0; f < MF; ++f) {
                float4 rs = make_float4(0.f, 0.f, 0.f, 0.f);
                if (RES) rs = load_res(r, f);
                flush_item(old[f][row], r, f, rs);
            }
        }
    }
}

// 4 KB of zeros (a stand-in residual for lanes outside the image) followed by one 16-byte trash slot per resident thread
// (where those lanes' stores go): allocated once per device on first use -- the first use of a shape is always an eager
// call, never inside a graph capture (FramePlan warms up eagerly).
static float *d_scratch() {
    static float *buf[kMaxDevices] = {nullptr};
    const int d = current_device();
    if (!buf[d]) {
        const size_t bytes = (1024 + (size_t)1024 * kDThreads * 4) * sizeof(float);
        float *ptr = nullptr;
        if (hipMalloc(reinterpret_cast<void **>(&ptr), bytes) != hipSuccess) return nullptr;
        if (hipMemset(ptr, 0, bytes) != hipSuccess) return nullptr;
        buf[d] = ptr;
    }
    return buf[d];
}

template <int MF, bool INACT, bool RES>
static int launch_d(const ConvP &p, hipStream_t st) {
    using G = DGeom<MF>;
    const int cus = device_cus();
    ConvP q = p;
    q.tiles_x = (p.Wout + 15) / 16;
    q.tiles_y = (p.Hout + G::TH - 1) / G::TH;
    q.m_tiles = (p.M_pad / 16 + MF - 1) / MF;
    const size_t lds = (size_t)G::LDS_BYTES + (size_t)q.m_tiles * G::TM * sizeof(float);
    if (lds > 160 * 1024) return fail("conv2d(f16x3d): %zu bytes of LDS", lds);
    static LdsGrant grant;
    if (grant.ensure(reinterpret_cast<const void *>(conv3_f16x3d_kernel<MF, INACT, RES>), lds)) return 1;
    const long long ntiles = (long long)q.tiles_x * q.tiles_y * q.m_tiles;
    if (ntiles <= 0 || ntiles > 0x7fffffffLL) return fail("conv2d(f16x3d): bad tile count %lld", ntiles);
    if (p.w16_plane * 2 > 0x7fffffffLL) return fail("conv2d(f16x3d): weight image too large for 32-bit lane offsets");
    long long blocks = cus < 1024 ? cus : 1024;   // one persistent 8-wave workgroup per CU
    if (blocks > ntiles) blocks = ntiles;
    float *scratch = d_scratch();
    if (!scratch) return fail("conv2d(f16x3d): cannot allocate the epilogue scratch buffer");
    q.gdn_x = V{scratch, 0, 0, 0, 0};             // (GDN epilogues never come here: fast_epi only)
    hipLaunchKernelGGL((conv3_f16x3d_kernel<MF, INACT, RES>), dim3((unsigned)blocks), dim3(kDThreads), lds, st, q);
    return launch_status("conv2d(f16x3d)");
}

int dispatch_conv3_f16x3d(const ConvP &p, hipStream_t st, char *kernel_name) {
    const int frags = p.M_pad / 16;
    const int mf = (frags > 4 && frags % 4 != 0 && frags % 3 == 0) ? 3 : (frags >= 4 ? 4 : frags);
    // one phase per tile (Cin <= 16): nothing to defer into; fewer than 48 output channels: too little MFMA work per
    // staged patch for the smaller tile -- both stay on the 24x16 kernel
    if (p.n_chunks16 < 2 || mf < 3 || p.res2.p != nullptr) return dispatch_conv3_f16x3p(p, st, kernel_name);
    const bool inact = p.in_act == LSSVC_INACT_LRELU, res = p.res.p != nullptr;
    snprintf(kernel_name, 96, "conv3_f16x3d_kernel<%d, %s, %s>", mf, inact ? "true" : "false", res ? "true" : "false");
#define LSSVC_D_CASE(m)                                                                   \
    if (mf == m) {                                                                        \
        if (inact) return res ? launch_d<m, true, true>(p, st) : launch_d<m, true, false>(p, st); \
        return res ? launch_d<m, false, true>(p, st) : launch_d<m, false, false>(p, st);  \
    }
    LSSVC_D_CASE(4) LSSVC_D_CASE(3)
#undef LSSVC_D_CASE
    return fail("conv2d(f16x3d): no kernel for MF=%d", mf);
}

}  // namespace lssvc
