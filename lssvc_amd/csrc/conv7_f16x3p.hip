// conv7_f16x3p.hip -- 7x7 stride-1 convolution (SpyNet's moduleBasic, video_net_component.py:191-211) in the f16x3 mode:
// PERSISTENT, double-buffered, WARP-SPECIALISED, the 7x7 counterpart of conv3_f16x3p.hip.
//
// One workgroup of 8 waves per CU walks 24x16-pixel x 16 MF-channel tiles; waves 0-3 (one per SIMD) are consumers
// (ds_read_b128 fragments + MFMA + the fused epilogue), waves 4-7 producers (fp32 patch -> fp16 hi/lo planes in LDS,
// weights by LDS-DMA). A PHASE is one kernel row of one 16-channel chunk: 7 taps = three tap-pair K steps + the packed
// odd step = 11 MFMAs per (fragment, pixel row) instead of 12. What differs from the 3x3 kernel:
//   * the (24+6) x (16+6)-pixel halo patch of a chunk serves SEVEN phases, so it is double-buffered per CHUNK (2 x 42 KB)
//     and the producers write the next chunk's patch in seven slices, one per phase, while the consumers walk the kernel
//     rows of the current one; only the 7-tap weight slab (2 x 29 KB at MF = 4) changes every phase;
//   * a fragment row of kernel row ky is patch row (r + ky): the consumers' B addresses move by one patch row per phase.
// LDS: 2 x 42 KB + 2 x 29 KB = 139 KB at MF = 4. Arithmetic, K order inside a step, accumulator layout and the epilogue are
// those of conv_f16x3_kernel<MF, RPW, 7, 1> (one kernel row per phase there too): results are bit-identical to it
// (tests/test_gpu_bench_kernels.py).
#include <type_traits>
#include <utility>

#include "conv_f16x3_kernel.h"

namespace lssvc {

constexpr int kP7Threads = 512;
constexpr int kP7Consumers = 4;          // waves 0..3
constexpr int kP7ProducerThreads = kP7Threads - 64 * kP7Consumers;

template <int MF>
struct P7Geom {
    static constexpr int KS = 7, RPW = 6, TH = RPW * kP7Consumers, TM = 16 * MF;
    static constexpr int PH = TH + KS - 1, PW = 16 + KS - 1, NTAP = KS, NSTEP = (NTAP + 1) / 2;
    static constexpr int PATCH_HALFS = PH * PW * CK16;            // per plane
    static constexpr int PATCH_ITEMS = PH * PW * 4;               // float4 items of a whole patch
    static constexpr int SLICE_ITEMS = ((PATCH_ITEMS + KS - 1) / KS + 3) & ~3;   // ... written per phase (whole pixels: a lane keeps its channel quad)
    static constexpr int NP = (SLICE_ITEMS + kP7ProducerThreads - 1) / kP7ProducerThreads;
    static constexpr int W_HALFS = NTAP * TM * CK16;              // per plane, one kernel row
    static constexpr int W_ITEMS = NTAP * TM * 2;                 // 16-byte items per plane
    static constexpr int W_INSTR = (2 * W_ITEMS + 63) / 64;       // wave-level DMA instructions for both planes
    static constexpr int NPROD = kP7ProducerThreads / 64;
    static constexpr int NDMA = (W_INSTR + NPROD - 1) / NPROD;
    static constexpr int LDS_BYTES = 2 * (2 * PATCH_HALFS + 2 * W_HALFS) * 2;      // + the bias vector (launch_p7)
    static_assert((2 * W_ITEMS) % 64 == 0, "weight slab must be a whole number of 1 KiB DMA pieces");
};

// position in the flattened phase sequence of a workgroup: tile, chunk (segment / channel offset / global chunk), kernel row
struct P7Pos {
    int it;
    KState k;      // seg, c0, ky, kc
};

template <int MF, bool INACT>
__global__ __launch_bounds__(kP7Threads, 1) void conv7_f16x3p_kernel(const ConvP p) {
    using G = P7Geom<MF>;
    constexpr int KS = G::KS, RPW = G::RPW, TM = G::TM, PW = G::PW, NTAP = G::NTAP, NSTEP = G::NSTEP, NP = G::NP;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16 *const patch0 = reinterpret_cast<_Float16 *>(smem);                    // [chunk parity][plane][PH*PW][16]
    _Float16 *const wts0 = patch0 + 4 * G::PATCH_HALFS;                             // [phase parity][plane][tap][m][16]
    float *const bias_s = reinterpret_cast<float *>(wts0 + 4 * G::W_HALFS);         // [m_tiles * TM], zero past M_pad

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    const int ntiles = p.tiles_x * p.tiles_y * p.m_tiles;
    const int nx = (int)gridDim.x < 8 ? (int)gridDim.x : 8;
    const int xcd = blockIdx.x % nx, kb = blockIdx.x / nx;
    const int nb_x = ((int)gridDim.x - xcd + nx - 1) / nx;
    const int tq = ntiles / nx, tr = ntiles % nx;
    const int t_begin = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
    const int t_cnt = tq + (xcd < tr ? 1 : 0);
    const int n_it = kb < t_cnt ? (t_cnt - kb + nb_x - 1) / nb_x : 0;
    if (n_it == 0) return;
    const int chunks_per_tile = p.n_chunks16;
    const int total = n_it * chunks_per_tile * KS;          // phases of this workgroup
    for (int i = tid; i < p.m_tiles * TM; i += kP7Threads) bias_s[i] = (p.bias && i < p.M_pad) ? p.bias[i] : 0.f;   // visible after barrier (A)

    auto tile_origin = [&](int it, int &oy0, int &ox0, int &m0) {
        const int tile = t_begin + kb + it * nb_x;
        const int mt = tile % p.m_tiles, pt = tile / p.m_tiles;
        const int tx = pt % p.tiles_x, ty = pt / p.tiles_x;
        oy0 = ty * G::TH;
        ox0 = tx * 16;
        m0 = mt * TM;
    };
    auto next_chunk = [&](P7Pos q) {                         // first phase of the chunk after q's
        q.k.ky = 0;
        q.k.c0 += CK16;
        ++q.k.kc;
        if (q.k.c0 >= p.in[q.k.seg].C) {
            q.k.c0 = 0;
            ++q.k.seg;
            if (q.k.seg >= p.n_in) {
                q.k = KState{0, 0, 0, 0};
                ++q.it;
            }
        }
        return q;
    };
    auto next_phase = [&](P7Pos q) {
        if (q.k.ky + 1 < KS) {
            ++q.k.ky;
            return q;
        }
        return next_chunk(q);
    };

    if (wave >= kP7Consumers) {
        // =================================================================================== PRODUCER waves
        const int lt = tid - 64 * kP7Consumers;                   // 0 .. 255
        const int pw = wave - kP7Consumers;
        const int quad4 = (lt & 3) * 4;
        const float in_slope = p.in_slope;
        const int Hin = p.in[0].H, Win = p.in[0].W;
        const _Float16 *w16 = reinterpret_cast<const _Float16 *>(p.w16);

        // weight-DMA lane offsets inside one (chunk, ky) slab [hi | lo]: recomputed when the tile (its M tile) changes
        int woff[G::NDMA];
        int geom_it = -1;
        auto w_geometry = [&](int it) {
            int oy0, ox0, m0;
            tile_origin(it, oy0, ox0, m0);
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
        // weights of phase q -> weight buffer `wbuf`
        auto fill_weights = [&](const P7Pos &q, int wbuf) {
            if (q.it != geom_it) w_geometry(q.it);
            if (p.debug & 1) return;
            unsigned char *dst = reinterpret_cast<unsigned char *>(wts0 + wbuf * 2 * G::W_HALFS);
            const _Float16 *src0 = w16 + ((size_t)q.k.kc * KS + q.k.ky) * NTAP * p.M_pad * CK16;
#pragma unroll
            for (int t = 0; t < G::NDMA; ++t) {
                int j = pw + G::NPROD * t;
                if (j >= G::W_INSTR) j = G::W_INSTR - 1;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src0 + woff[t]),
                                                 (__attribute__((address_space(3))) void *)(dst + j * 1024), 16, 0, 0);
            }
        };
        // slice `part` (0..KS-1) of the patch of chunk c (its tile c.it, channels c.k.seg / c.k.c0) -> patch buffer `pbuf`
        auto fill_patch_slice = [&](const P7Pos &c, int part, int pbuf) {
            if (p.debug & 2) return;
            int oy0, ox0, m0;
            tile_origin(c.it, oy0, ox0, m0);
            const V X = p.in[c.k.seg];
            const bool cvalid = quad4 < X.C - c.k.c0;
            const int cc = cvalid ? c.k.c0 + quad4 : 0;
            float4 preg[NP];
            unsigned pmask = 0;
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int sl = lt + i * kP7ProducerThreads;                       // item inside the slice
                const int idx = part * G::SLICE_ITEMS + sl;
                const int pix = idx >> 2;
                const int py = pix / PW, px = pix - py * PW;
                const int gy = oy0 - p.pad_t + py, gx = ox0 - p.pad_l + px;
                const bool in_slice = sl < G::SLICE_ITEMS && idx < G::PATCH_ITEMS;
                const bool ok = in_slice && gy >= 0 && gy < Hin && gx >= 0 && gx < Win && cvalid;
                const size_t off = ok ? (size_t)(gy * Win + gx) * X.ld + cc : 0;
                preg[i] = *reinterpret_cast<const float4 *>(X.p + off);
                pmask |= ok ? (1u << i) : 0u;
            }
            _Float16 *ph_ = patch0 + pbuf * 2 * G::PATCH_HALFS;
            _Float16 *pl_ = ph_ + G::PATCH_HALFS;
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int sl = lt + i * kP7ProducerThreads;
                const int idx = part * G::SLICE_ITEMS + sl;
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
                if (sl < G::SLICE_ITEMS && idx < G::PATCH_ITEMS) {
                    const int o = (idx >> 2) * CK16 + quad4;
                    *reinterpret_cast<f16x4 *>(ph_ + o) = h;
                    *reinterpret_cast<f16x4 *>(pl_ + o) = l;
                }
            }
        };

        // Schedule. Consumer phase k reads weight buffer k & 1 and the patch buffer of its chunk (global chunk counter
        // parity). fill(k) -- everything consumer phase k needs that is not in LDS yet -- runs during consumer phase k-1:
        //   weights of phase k; and ONE slice of a patch: ky(k) >= 1 -> slice ky-1 of the NEXT chunk's patch (its buffer was
        //   released when the consumers left the previous chunk); ky(k) == 0 -> the last slice (KS-1) of phase k's own chunk.
        P7Pos q{0, KState{0, 0, 0, 0}};
        int cc = 0;                                        // global chunk counter of q
        fill_weights(q, 0);
#pragma unroll 1
        for (int part = 0; part < KS; ++part) fill_patch_slice(q, part, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's weight DMA has landed
        __syncthreads();                                   // (A) phase 0 is staged
        for (int k = 0; k < total; ++k) {
            if (k + 1 < total) {
                const P7Pos n = next_phase(q);
                const bool new_chunk = n.k.ky == 0;
                fill_weights(n, (k + 1) & 1);
                if (new_chunk) {
                    fill_patch_slice(n, KS - 1, (cc + 1) & 1);               // last slice of the chunk that starts at phase k+1
                } else {
                    const P7Pos nc = next_chunk(q);                          // the chunk after the current one
                    if (nc.it < n_it) fill_patch_slice(nc, n.k.ky - 1, (cc + 1) & 1);
                }
                if (new_chunk) ++cc;
                q = n;
            }
            // LDS-DMA data is ordered for the consumers' ds_reads only by the issuing wave's vmcnt wait + a barrier
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                               // (B_k)
        }
        return;
    }

    // ======================================================================================= CONSUMER waves
    const int li = lane & 15;
    const int lg = lane >> 4;
    const int tsel = lg >> 1;
    const int ch8 = (lg & 1) * 8;

    f32x4 acc[MF][RPW];
#pragma unroll
    for (int a = 0; a < MF; ++a)
#pragma unroll
        for (int b = 0; b < RPW; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // per-lane byte offsets (see conv3_f16x3p.hip): A = weights [tap][m][16], B = patch [row][col][16]
    const unsigned a_lane = (unsigned)(li * CK16 + ch8) * 2u;
    const unsigned b_lane = (unsigned)(((wave * RPW) * PW + li) * CK16 + ch8) * 2u;
    auto lds_addr = [](const _Float16 *ptr) { return (unsigned)(size_t)(lds_cfloat_ptr)(const float *)(const void *)ptr; };
    auto lds_read = [](unsigned addr) { return *reinterpret_cast<const __attribute__((address_space(3))) f16x8 *>((size_t)addr); };

    __syncthreads();                                       // (A)
    int it = 0, ky = 0, chunk = 0, cc = 0;                 // tile, kernel row, chunk inside the tile, global chunk counter
    for (int k = 0; k < total; ++k) {
        const unsigned wh_b = lds_addr(wts0 + (k & 1) * 2 * G::W_HALFS), wl_b = wh_b + (unsigned)G::W_HALFS * 2u;
        const unsigned ph_b = lds_addr(patch0 + (cc & 1) * 2 * G::PATCH_HALFS) + (unsigned)(ky * PW * CK16) * 2u;   // kernel row ky
        const unsigned pl_b = ph_b + (unsigned)G::PATCH_HALFS * 2u;
        constexpr int GR = 2, NG = RPW / GR, NUNIT = NG * NSTEP;
        f16x8 fa1[2][MF], fa2[2][MF], fb1[2][GR], fb2[2][GR];
        auto load_a = [&](int u, f16x8 (&a1)[MF], f16x8 (&a2)[MF]) {
            const bool odd = 2 * u + 1 >= NTAP;                              // tap 6: f16x3_step_odd
            const unsigned t0 = (unsigned)(2 * u) * TM * CK16 * 2u, t1 = odd ? t0 : (unsigned)(2 * u + 1) * TM * CK16 * 2u;
            const unsigned tap_b = a_lane + (tsel ? t1 : t0);
            const unsigned p1 = ((odd && !tsel) ? wl_b : wh_b) + tap_b, p2 = ((odd && !tsel) ? wh_b : wl_b) + tap_b;
#pragma unroll
            for (int f = 0; f < MF; ++f) {
                a1[f] = lds_read(p1 + (unsigned)(f * 16 * CK16) * 2u);
                a2[f] = lds_read(p2 + (unsigned)(f * 16 * CK16) * 2u);
            }
        };
        auto load_b = [&](int u, int g, f16x8 (&b1)[GR], f16x8 (&b2)[GR]) {
            const bool odd = 2 * u + 1 >= NTAP;
            const unsigned o0 = (unsigned)((2 * u) * CK16) * 2u, o1 = odd ? o0 : (unsigned)((2 * u + 1) * CK16) * 2u;    // kx = tap
            const unsigned tap_b = b_lane + (tsel ? o1 : o0);
            const unsigned p1 = ((odd && tsel) ? pl_b : ph_b) + tap_b, p2 = pl_b + tap_b;
#pragma unroll
            for (int r = 0; r < GR; ++r) {
                const unsigned ro = (unsigned)((g * GR + r) * PW * CK16) * 2u;
                b1[r] = lds_read(p1 + ro);
                if (!odd) b2[r] = lds_read(p2 + ro);
            }
        };
        auto unit = [&](auto tc) {
            constexpr int t = decltype(tc)::value;
            constexpr int u = t / NG, g = t % NG;
            constexpr bool odd = 2 * u + 1 >= NTAP;
            constexpr bool more = t + 1 < NUNIT;
            constexpr int nu = (t + 1) / NG, ng = (t + 1) % NG;
            constexpr bool nodd = 2 * nu + 1 >= NTAP;
            constexpr bool pre_a = (NG >= 2 ? g == NG - 2 : true) && u + 1 < NSTEP;
            constexpr int NR = (pre_a ? 2 * MF : 0) + (more ? (nodd ? GR : 2 * GR) : 0);
            constexpr int NM = (odd ? 2 : 3) * MF * GR;
            if (pre_a) load_a(u + 1, fa1[(u + 1) & 1], fa2[(u + 1) & 1]);
            if (more) load_b(nu, ng, fb1[(t + 1) & 1], fb2[(t + 1) & 1]);
            const f16x8(&a1)[MF] = fa1[u & 1];
            const f16x8(&a2)[MF] = fa2[u & 1];
            const f16x8(&b1)[GR] = fb1[t & 1];
            const f16x8(&b2)[GR] = fb2[t & 1];
            if (odd) {
#pragma unroll
                for (int f = 0; f < MF; ++f)
#pragma unroll
                    for (int r = 0; r < GR; ++r)
                        acc[f][g * GR + r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[f], b1[r], acc[f][g * GR + r], 0, 0, 0);
#pragma unroll
                for (int f = 0; f < MF; ++f)
#pragma unroll
                    for (int r = 0; r < GR; ++r)
                        acc[f][g * GR + r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2[f], b1[r], acc[f][g * GR + r], 0, 0, 0);
            } else {
#pragma unroll
                for (int f = 0; f < MF; ++f)
#pragma unroll
                    for (int r = 0; r < GR; ++r)
                        acc[f][g * GR + r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2[f], b1[r], acc[f][g * GR + r], 0, 0, 0);
#pragma unroll
                for (int f = 0; f < MF; ++f)
#pragma unroll
                    for (int r = 0; r < GR; ++r)
                        acc[f][g * GR + r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[f], b2[r], acc[f][g * GR + r], 0, 0, 0);
#pragma unroll
                for (int f = 0; f < MF; ++f)
#pragma unroll
                    for (int r = 0; r < GR; ++r)
                        acc[f][g * GR + r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[f], b1[r], acc[f][g * GR + r], 0, 0, 0);
            }
            constexpr int NI = (2 * NR <= NM) ? NR : NM / 2;                 // issue order: (2 MFMA, 1 ds_read) x reads, then MFMAs
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            if (NR > NI) __builtin_amdgcn_sched_group_barrier(0x100, NR - NI, 0);
            if (NM > 2 * NI) __builtin_amdgcn_sched_group_barrier(0x008, NM - 2 * NI, 0);
            __builtin_amdgcn_sched_barrier(0);
        };
        load_a(0, fa1[0], fa2[0]);
        load_b(0, 0, fb1[0], fb2[0]);
        __builtin_amdgcn_sched_barrier(0);
        [&]<int... I>(std::integer_sequence<int, I...>) { (unit(std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, NUNIT>{});

        __syncthreads();                                   // (B_k)
        if (++ky == KS) {
            ky = 0;
            ++cc;
            if (++chunk == chunks_per_tile) {
                chunk = 0;
                int oy0, ox0, m0;
                tile_origin(it, oy0, ox0, m0);
                if (!(p.debug & 32)) {
                    const int oy_w = oy0 + wave * RPW;
                    auto pix = [&](int r, int col) {
                        return (ox0 + col < p.Wout && oy_w + r < p.Hout) ? (long long)(oy_w + r) * p.Wout + ox0 + col : -1LL;
                    };
                    const bool interior = oy0 + wave * RPW + RPW <= p.Hout && ox0 + 16 <= p.Wout && m0 + TM <= p.Cout;
                    conv_epilogue_fast_f<MF, RPW, true>(p, acc, pix, m0, lg, p.w16_unscale, interior, (lds_cfloat_ptr)bias_s);
                }
#pragma unroll
                for (int a = 0; a < MF; ++a)
#pragma unroll
                    for (int b = 0; b < RPW; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
                ++it;
            }
        }
    }
}

template <int MF, bool INACT>
static int launch_p7(const ConvP &p, hipStream_t st) {
    using G = P7Geom<MF>;
    const int cus = device_cus();
    ConvP q = p;
    q.tiles_x = (p.Wout + 15) / 16;
    q.tiles_y = (p.Hout + G::TH - 1) / G::TH;
    q.m_tiles = (p.M_pad / 16 + MF - 1) / MF;
    const size_t lds = (size_t)G::LDS_BYTES + (size_t)q.m_tiles * G::TM * sizeof(float);
    if (lds > 160 * 1024) return fail("conv2d(f16x3p 7x7): %zu bytes of LDS", lds);
    static LdsGrant grant;
    if (grant.ensure(reinterpret_cast<const void *>(conv7_f16x3p_kernel<MF, INACT>), lds)) return 1;
    const long long ntiles = (long long)q.tiles_x * q.tiles_y * q.m_tiles;
    if (ntiles <= 0 || ntiles > 0x7fffffffLL) return fail("conv2d(f16x3p 7x7): bad tile count %lld", ntiles);
    if (p.w16_plane * 2 > 0x7fffffffLL) return fail("conv2d(f16x3p 7x7): weight image too large for 32-bit lane offsets");
    long long blocks = cus;                       // one persistent 8-wave workgroup per CU
    if (blocks > ntiles) blocks = ntiles;
    hipLaunchKernelGGL((conv7_f16x3p_kernel<MF, INACT>), dim3((unsigned)blocks), dim3(kP7Threads), lds, st, q);
    return launch_status("conv2d(f16x3p 7x7)");
}

// Entry conditions as for the 3x3 kernel: fast epilogue (Cout % 4 == 0, >= 32 channels, no GDN / shuffle / scale), an input activation in
// the max(x, s x) form, and enough tiles to give every CU one (f16x3_persist_min_tiles, shared with the 3x3 kernel).
bool conv7_f16x3p_wanted(const ConvP &p) {
    const int on = option_get(OPT_P3_ON), min_tiles = option_get(OPT_P3_MIN_TILES);
    if (!on || !p.fast_epi || p.res2.p != nullptr) return false;
    if (p.in_act == LSSVC_INACT_LRELU && !(p.in_slope >= 0.0f && p.in_slope <= 1.0f)) return false;
    const int frags = p.M_pad / 16, mf = frags >= 4 ? 4 : frags;
    if (frags < 2 && !option_get(OPT_P7_NARROW)) return false;       // 16 output channels: the producers' patch conversion outweighs 11 MFMAs per row; tiled kernel wins (round 6 re-measured: option p7_narrow)
    const long long ntiles = (long long)((p.Wout + 15) / 16) * ((p.Hout + 23) / 24) * ((frags + mf - 1) / mf);
    return ntiles >= min_tiles;
}

int dispatch_conv7_f16x3p(const ConvP &p, hipStream_t st, char *kernel_name) {
    const int frags = p.M_pad / 16, mf = frags >= 4 ? 4 : frags;
    const bool inact = p.in_act == LSSVC_INACT_LRELU;
    snprintf(kernel_name, 96, "conv7_f16x3p_kernel<%d, %s>", mf, inact ? "true" : "false");
#define LSSVC_P7_CASE(m) \
    if (mf == m) return inact ? launch_p7<m, true>(p, st) : launch_p7<m, false>(p, st);
    LSSVC_P7_CASE(4) LSSVC_P7_CASE(3) LSSVC_P7_CASE(2) LSSVC_P7_CASE(1)
#undef LSSVC_P7_CASE
    return fail("conv2d(f16x3p 7x7): no kernel for MF=%d", mf);
}

}  // namespace lssvc
