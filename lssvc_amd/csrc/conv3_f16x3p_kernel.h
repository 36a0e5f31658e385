// conv3_f16x3p_kernel.h -- 3x3 stride-1 convolution in the f16x3 mode: PERSISTENT, double-buffered, WARP-SPECIALISED.
//
// Why. In the tiled kernel (conv_f16x3_kernel.h) every wave does everything: address
// arithmetic + global loads, fp32 -> fp16 hi/lo conversion + LDS stores, LDS fragment reads, MFMAs, epilogue. All of
// that is one in-order instruction stream per wave, and the waves of a workgroup are barrier-locked into doing the
// same part at the same time, so the matrix pipe idles while they stage. Ablations on MI355X (3x3 64->64): loads
// -19 %, conversion -8 %, fragment reads -7 %, epilogue -16 %, loop bookkeeping -26 %; the kernel ran at 215-255
// TFLOP/s where a bare LDS-fed MFMA loop sustains ~620 (f16x3-equivalent; tools/probes/mfma_probe.hip).
// Here the work is split by wave role, one workgroup of 8 waves per CU, persistent over (4*RPW)x16-pixel x 16*MF-channel
// tiles (RPW = 6 rows per consumer by default: 24x16 pixels, 96 accumulator registers):
//   * waves 0-3, one per SIMD, are CONSUMERS: ds_read_b128 fragments + MFMA, nothing else, RPW pixel rows x MF channel
//     fragments each; after a tile's last phase they run the fused epilogue;
//   * waves 4-7, one per SIMD, are PRODUCERS: they fetch the next phase's (4*RPW+2)x18x16-channel fp32 halo patch, apply the
//     input activation, split it into fp16 hi/lo planes in LDS, and move the next phase's weights (already fp16 in LDS
//     image order) global -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPRs, no ds_write). Their VALU / VMEM issue
//     slots interleave with the consumer's MFMAs on the same SIMD (an MFMA holds the issue port 8 of its 16 cycles);
//     their memory latency is hidden by not being on the consumers' path at all.
//   * both operand images are double-buffered in LDS (2 x (30 KB patch + 37 KB weights) = 134 KB at RPW = 6); a phase is one
//     16-channel chunk, all 9 taps, and the phase sequence runs on across tile boundaries, so tiles have no head or tail.
//     The buffers are handed over through per-wave LDS slots (fills completed / phases completed), not barriers: a consumer
//     waits only for the fill of its next phase, only the producers wait for the slowest consumer.
// Arithmetic, operand order inside a K-step, accumulator layout and the fused epilogue are those of
// conv_f16x3_kernel (f16x3_step_pair per two taps x 16 channels, f16x3_step_odd for the ninth tap): results are
// bit-identical to it.
#pragma once
#include <type_traits>
#include <utility>

#include "conv_f16x3_kernel.h"

namespace lssvc {

#ifndef LSSVC_P3_RPW
#define LSSVC_P3_RPW 6      // rows per consumer wave (Makefile P3_RPW); even
#endif
#ifndef LSSVC_P3_MFMA_PER_READ
#define LSSVC_P3_MFMA_PER_READ 2   // consumer issue pattern: this many MFMAs, then one LDS fragment read
#endif
constexpr int kP3Threads = 512;
constexpr int kP3Consumers = 4;          // waves 0..3
constexpr int kP3ProducerThreads = kP3Threads - 64 * kP3Consumers;

// S = 2 (stride-2 convs, round 4): the input patch of a tile is 4x its output, so the tile is 8x16 output pixels (RPW = 2 rows per
// consumer wave, (2*8+1) x 33 input pixels per 16-channel chunk: 2 x 36 KB of patch + 2 x 37 KB of weights at MF = 4); fragment
// column li reads patch column 2*li + kx, a 64-byte lane stride that would be a 4-way bank conflict on ds_read_b128, so the even
// and the odd columns of a patch row are stored as two runs ([even 0,2,.. | odd 1,3,..]: column c lives at (c & 1) * PWE + (c >> 1)),
// as in the tiled kernel. These launches are paced by their producers (4 input pixels per output pixel: they are HBM-bound at a
// matrix-pipe occupancy of about 0.4), which is the point: the tiled kernel ran them at half that.
// RPWT (round 6): rows per consumer wave chosen per instantiation instead of by stride (0 = the defaults: LSSVC_P3_RPW at stride 1, 2 at
// stride 2). Small tiles -- 16x16, 8x16, 4x16 pixels at RPWT = 4, 2, 1 -- give the 72x120 ... 288x480 maps of the prior / hyper /
// quarter-resolution networks at least one tile per CU (the 24x16 tiling leaves them 23 ... 180 tiles for 256 CUs), and at MF = 1 a
// 16x16 tiling needs 62 KB of LDS, so that TWO workgroups share a CU (narrow-output convs: see the kernel's FLAT / PF2 notes).
template <int MF, int S = 1, int RPWT = 0>
struct P3Geom {
    static constexpr int RPW = RPWT ? RPWT : (S == 2 ? 2 : LSSVC_P3_RPW), HALF = RPW / 2, TH = RPW * kP3Consumers, TM = 16 * MF;
    static constexpr int PH = (TH - 1) * S + 3, PW = 15 * S + 3, PWE = (PW + 1) / 2, NTAP = 9, NSTEP = 5;
    static constexpr int PATCH_HALFS = PH * PW * CK16;            // per plane
    static constexpr int PATCH_ITEMS = PH * PW * 4;               // float4 items
    static constexpr int NP = (PATCH_ITEMS + kP3ProducerThreads - 1) / kP3ProducerThreads;
    static constexpr int W_HALFS = NTAP * TM * CK16;              // per plane
    static constexpr int W_ITEMS = NTAP * TM * 2;                 // 16-byte items per plane
    static constexpr int W_INSTR = 2 * W_ITEMS / 64;              // wave-level DMA instructions for both planes (= 9 MF)
    static constexpr int NPROD = kP3ProducerThreads / 64;
    static constexpr int NDMA = (W_INSTR + NPROD - 1) / NPROD;
    // PATCH RING (round 5): with three patch buffers instead of two the producers run the patch one phase further ahead than the weights
    // (patch of phase k+2 beside the weights of phase k+1, while the consumers run phase k): twice the patch bytes in flight per CU and no
    // fill that is late because its buffer was released late -- the 48-channel kernels waited 18 % of their cycles for fills. Where the LDS
    // holds it: 3 x 30 KB of patch + 2 x 27.6 KB of weights at MF = 3; at MF = 4 the sum is 163 584 of the 163 840 bytes and the bias
    // vector, the hand-off slots and the trash slots no longer fit, at stride 2 the patch is 36 KB: those keep two buffers.
#ifdef LSSVC_P3_NO_RING      // (A/B build: two patch buffers everywhere, the round-4 schedule)
    static constexpr int NPB = 2;
#else
    static constexpr int NPB = (S == 1 && RPWT == 0 && (3 * 2 * PATCH_HALFS + 2 * 2 * W_HALFS) * 2 + 2048 + 64 + 2 * kP3ProducerThreads * 8 <= 160 * 1024) ? 3 : 2;
#endif
    static constexpr int LDS_BYTES = 2 * (2 * PATCH_HALFS + 2 * W_HALFS) * 2;      // two-buffer layout; + the bias vector (launch_p3)
    static constexpr int LDS_BYTES_RING = (NPB * 2 * PATCH_HALFS + 2 * 2 * W_HALFS) * 2;
    // ---- STAGE variant (staged epilogue, see the kernel): the whole 160 KB, laid out [patch 0][weights 0][F][weights 1][patch 1][tail]
    // so that either operand buffer pair plus the free middle F is ONE contiguous run the finished tile can be parked in
    static constexpr int PB = 2 * PATCH_HALFS * 2, WB = 2 * W_HALFS * 2;          // bytes of one patch / weight buffer (both planes)
    static constexpr int BIAS_MAX = 1024, TAIL = BIAS_MAX + 64 + 2 * kP3ProducerThreads * 8;      // bias (Cout <= 256), hand-off slots, trash slots
    static constexpr int TOTAL = 160 * 1024;
    static constexpr int FB = (TOTAL - 2 * (PB + WB) - TAIL) / 16 * 16;
    static constexpr int ROWSET = kP3Consumers * 16 * TM * 4;                     // bytes of one output row of every consumer wave (fp32)
    static constexpr int SR_FIT = (PB + WB + FB) / ROWSET < RPW ? (PB + WB + FB) / ROWSET : RPW;   // rows per wave that fit: 5 of 6 at MF = 4, all 6 below
    static constexpr int SR = (MF == 4 && SR_FIT > 4) ? 4 : SR_FIT;        // (MF = 4: 4, so that a producer lane's 16 items + their residuals stay in registers)
    static constexpr int STAGE_BYTES = SR * ROWSET;
    static constexpr int OFF_P0 = 0, OFF_W0 = PB, OFF_F = PB + WB, OFF_W1 = PB + WB + FB, OFF_P1 = PB + WB + FB + WB, OFF_TAIL = 2 * (PB + WB) + FB;
    static_assert(FB >= 0 && STAGE_BYTES <= PB + WB + FB && OFF_TAIL + TAIL <= TOTAL, "staged layout");
};

// Diagnostic builds only (make P3_ABLATE=n into a separate library, tools/r6_roles_ablation.sh): the split-roles schedule with one of its
// parts switched off -- 1: no weight DMA after the first phase, 2: no MFMAs, 4: no patch loads after the first two, 8: no conversion / LDS
// stores of the patch. Results are wrong by design; the shipped library is built with 0 and none of this exists in its code.
#ifndef LSSVC_P3_ABLATE
#define LSSVC_P3_ABLATE 0
#endif
constexpr int kP3Ablate = LSSVC_P3_ABLATE;

// s_waitcnt immediate of gfx9 / gfx950: vmcnt in bits 3:0 and 15:14, expcnt 6:4 (7 = no wait), lgkmcnt 11:8 (15 = no wait)
// COUNTED waits (0 < vm < 63: some loads are meant to stay in flight) carry expcnt(6) instead of 7: a compute kernel has no exports, the
// counter is always 0 and the field costs nothing, but the disassembly shows `s_waitcnt vmcnt(N) expcnt(6)` -- the mark by which
// tools/p3_waitcnt_check.py tells the hand-written counted waits from the compiler's own and checks N against the loads in front of them.
constexpr int p3_waitcnt(int vm, int lgkm) { return (vm & 15) | ((vm > 0 && vm < 63 ? 6 : 7) << 4) | ((lgkm & 15) << 8) | (((vm >> 4) & 3) << 14); }

struct P3Phase {
    int it;        // index into this workgroup's tile sequence
    KState k;      // segment / channel offset / global chunk index of the phase
};

// STAMP: diagnostic build only (LSSVC_CONV_DEBUG & 256; never dispatched otherwise): consumer wave 0..3 of every workgroup
// accumulates s_memtime deltas of its compute / barrier-wait / epilogue sections and writes them, with the s_memrealtime
// span of the loop, to the buffer passed in p.gdn_x.p (unused on this path): 8 x int64 per wave.
//
// STAGE (staged epilogue; round 4): with the epilogue in the consumer waves the matrix pipe idles while they store -- a store
// instruction blocks the issuing wave until the CU's memory pipeline takes it, and a residual is a full memory round trip with
// nothing else to do (ablation, 64->64 @1152x1920: 444 us -> 381 us without the epilogue, 48->48: 315 -> 231; with a residual
// at 576x960: 149 -> 93). Only another WAVE can issue those stores beside the MFMAs, and the only way to hand a wave's
// accumulators to another wave is LDS, which the operand buffers fill. But when a tile's last phase k has been computed, the
// operand pair k & 1 is dead until the producers refill it for phase k + 2: in the STAGE layout that pair and the free middle
// of the LDS are one contiguous run, the consumers park the finished tile there (bias and activation applied, fp32, 16 bytes
// per lane; SR = 5 of a wave's 6 rows at MF = 4, all 6 at MF <= 3 -- the sixth row of an MF = 4 tile is stored by the consumer
// as before) and go straight on to the next tile; each producer wave then moves one consumer wave's rows to global memory
// (adding the residuals, plain or pixel-shuffle store) with the NEXT fill's patch loads already in flight, and only after all
// four have drained is the pair refilled. Same arithmetic per element as the direct epilogue: results are bit-identical.
// SPLIT (round 5): the inputs are PRE-SPLIT tensors (lssvc_hip.h: LSSVC_PREC_SPLIT_IN) -- per pixel and 16-channel chunk 64 bytes,
// [hi: 16 x fp16 | lo: 16 x fp16], written by lssvc_presplit (an epilogue that writes the format was not built) with the input activation already
// applied; same bytes per element as fp32. The patch then goes global -> LDS by LDS-DMA like the weights: no patch registers, no
// conversion, no ds_write in the producer waves; every 16-byte unit of the LDS image [plane][patch pixel][16 halfs] is fetched by
// one lane from wherever it lives (zero padding: from a 64-byte block of zeros), so the LDS layout the consumers read is unchanged.
static __device__ __attribute__((aligned(64))) unsigned g_p3_zero_block[16];      // zero-initialised (one per translation unit)

// PF2 (round 6; two LDS patch buffers only): the producers keep the patch of phase k+2 IN FLIGHT IN REGISTERS while they convert and
// store the patch of phase k+1 (two register sets, alternating). With one set the loads of a fill are requested only after the previous
// fill has been converted, so at most one phase's patch (36 KB per CU at stride 2) is in flight, and only for part of the phase: the
// stride-2 kernels -- HBM-bound: four input pixels per output pixel -- ran a phase in 3.3 us whatever their channel count, 2.4-3.0 TB/s.
// (At stride 1 / MF = 4, the dominant kernel, the same idea in its LDS form was measured SLOWER -- the matrix pipe is the limit there and
// the extra loads in flight get in the way of the epilogue's stores -- and stays off.)
// PAIR (PF = 2, round 6): a fill loads the patches of TWO phases at once when they are neighbouring 16-channel chunks of one tensor --
// the two 64-byte halves of the same 128-byte lines -- converts and publishes the first, and fills the next phase from registers, without
// a load. In-kernel stamps of the stride-2 kernel (profiles/r06_s2_stamps.txt): 3.1 k of a producer's 6.2 k cycles per phase go into
// ISSUING nine loads -- the CU's memory pipeline is full of outstanding line requests, each of which this access pattern uses only half
// of (one 64-byte slice of a 128-byte line per phase; the other half is requested again a phase later, when the line has long left the
// 32 KB L1). Requested together the second half rides on the first one's line fetch.
// FLAT (round 6): the epilogue of convs the straight-line epilogue does not cover (Cout % 4 != 0: the 2-channel flow and 3-channel
// picture heads; scalar residuals; out_scale) -- conv_epilogue_flat, the tiled kernel's own, so results stay bit-identical to it.
// WG2: two workgroups per CU (launch bound: four waves per SIMD, 128 registers) -- the narrow-output instantiations, whose consumers
// have a quarter of the MFMA work per patch byte and whose producers are the limit: twice the producer waves per CU.
template <int MF, bool INACT, bool STAMP = false, bool STAGE = false, int S = 1, bool SPLIT = false, int RPWT = 0, int PF = 0, bool FLAT = false, bool WG2 = false>
__global__ __launch_bounds__(kP3Threads, WG2 ? 4 : 1) void conv3_f16x3p_kernel(const ConvP p) {
    static_assert(!(STAGE && S != 1), "the staged epilogue is laid out for stride 1");
    static_assert(!(SPLIT && INACT), "a pre-split input carries its activation already");
    constexpr bool PF2 = PF == 1, PAIR = PF == 2, ROLES = PF == 3, LATE = PF == 4;
    static_assert(!(PF && (STAGE || SPLIT || (STAMP && PF != 4))), "the register prefetch / pair loads are written for the plain two-buffer schedule");
    using G = P3Geom<MF, S, RPWT>;
    constexpr int RPW = G::RPW, TM = G::TM, PW = G::PW, NTAP = G::NTAP, NSTEP = G::NSTEP;
    // ROLES (PF = 3, round 6): producer wave 3 moves ALL of a phase's weights (LDS-DMA only), waves 0-2 stage the patch (plain loads
    // only). A wave whose instruction stream holds no LDS-DMA gets exact in-order vmcnt waits from the compiler -- with a DMA outstanding
    // it guards every use of a loaded register with vmcnt(0), which is what kept the register prefetch (PF = 1) from ever having two
    // fills in flight: the second step of its unrolled loop drains.
    constexpr int PT = ROLES ? kP3ProducerThreads - 64 : kP3ProducerThreads;      // threads that stage the patch
    constexpr int NP = (G::PATCH_ITEMS + PT - 1) / PT;                             // float4 items per staging thread and phase
    constexpr int NDMA = ROLES ? G::W_INSTR : G::NDMA;                             // weight-DMA instructions per DMA-issuing wave and phase
    constexpr int SR = STAGE ? G::SR : 0;
    // patch buffers (P3Geom: PATCH RING). Round 6: the narrow heads' split-roles schedule runs a ring of three as well -- its patch waves fill
    // up to two phases ahead of the consumers, whose phases take as long as a fill (profiles/r06_roles_ablation.txt); 3 x 20.3 KB of patch +
    // 2 x 9 KB of weights + 1 KB of trash slots (one wave's worth, shared) = 79.8 of the 80 KB two workgroups per CU leave
    constexpr bool RING3R = ROLES && WG2;
    constexpr int NPB = STAGE ? 2 : RING3R ? 3 : PF ? 2 : G::NPB;
    constexpr int TRASH_LANES = RING3R ? 64 : kP3ProducerThreads;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16 *const patch0 = reinterpret_cast<_Float16 *>(smem);                    // [buf][plane][PH*PW][16]
    _Float16 *const wts0 = STAGE ? reinterpret_cast<_Float16 *>(smem + G::OFF_W0) : patch0 + NPB * 2 * G::PATCH_HALFS;      // [buf][plane][tap][m][16]
    float *const bias_s = STAGE ? reinterpret_cast<float *>(smem + G::OFF_TAIL) : reinterpret_cast<float *>(wts0 + 4 * G::W_HALFS);   // [m_tiles * TM], zero past M_pad
    // operand buffer b of the double buffer (STAGE: buffer 1 sits at the far end, see P3Geom)
    auto patch_buf = [&](int b) { return STAGE ? reinterpret_cast<_Float16 *>(smem + (b ? G::OFF_P1 : G::OFF_P0)) : patch0 + b * 2 * G::PATCH_HALFS; };
    auto wts_buf = [&](int b) { return STAGE ? reinterpret_cast<_Float16 *>(smem + (b ? G::OFF_W1 : G::OFF_W0)) : wts0 + b * 2 * G::W_HALFS; };
    // where the tile whose last phase used buffer pair b is parked: pair 0 + F from the front, F + pair 1 up to the back
    auto stage_off = [&](int b) { return (unsigned)(b ? G::OFF_TAIL - G::STAGE_BYTES : 0); };
    // Hand-off slots: fills finished by each producer wave / phases finished by each consumer wave.
    // One s_barrier per phase makes every consumer wait for the SLOWEST consumer of that phase (stamps: ~600 of 6600
    // cycles); with the counters a consumer only waits for the data of its next phase, which the producers finish a
    // couple of thousand cycles ahead, and only the producers -- which have the slack -- wait for the last consumer.
    // One slot per wave (a shared counter could be satisfied by a fast wave signalling twice while a slow one has not signalled
    // at all): slot = number of fills / phases that wave has completed.
    int *const sync_s = STAGE ? reinterpret_cast<int *>(smem + G::OFF_TAIL + G::BIAS_MAX)
                              : reinterpret_cast<int *>(bias_s + p.m_tiles * TM);     // [0..3] producer waves: fills done; [4..7] consumer waves: phases done;
    //                                                                                   STAGE: [8..11] consumer waves: tiles parked; [12..15] producer waves: tiles drained
    _Float16 *const trash_s = reinterpret_cast<_Float16 *>(sync_s + (STAGE ? 16 : 8)); // one 8-byte slot per producer lane and plane: staged items past the patch land here

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    // ---- this workgroup's tiles: the XCD it runs on owns a contiguous range, its workgroups interleave inside it
    const int ntiles = p.tiles_x * p.tiles_y * p.m_tiles;
    // (grids smaller than 8 workgroups split the tiles into gridDim.x ranges instead of 8, so none is orphaned)
    const int nx = (int)gridDim.x < 8 ? (int)gridDim.x : 8;
    const int xcd = blockIdx.x % nx, kb = blockIdx.x / nx;
    const int nb_x = ((int)gridDim.x - xcd + nx - 1) / nx;
    const int tq = ntiles / nx, tr = ntiles % nx;
    const int t_begin = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
    const int t_cnt = tq + (xcd < tr ? 1 : 0);
    const int n_it = kb < t_cnt ? (t_cnt - kb + nb_x - 1) / nb_x : 0;
    if (n_it == 0) return;
    const int phases_per_tile = p.n_chunks16;
    for (int i = tid; i < p.m_tiles * TM; i += kP3Threads) bias_s[i] = (p.bias && i < p.M_pad) ? p.bias[i] : 0.f;   // visible after barrier (A)
    if (tid < (STAGE ? 16 : 8)) sync_s[tid] = 0;
    // spin until all four slots of a group are >= target (monotonic: every target is reached, and the grid drains, because a
    // fill never waits for a phase that needs it and a phase never waits for a fill that needs it)
    auto wait_for = [&](int *slots, int target) {
        while (__builtin_amdgcn_ballot_w64(__hip_atomic_load(slots + (lane & 3), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < target) != 0)
            __builtin_amdgcn_s_sleep(1);
    };
    // (the caller has waited, with explicit s_waitcnt, for exactly the LDS / DMA traffic the slot stands for; a RELEASE store
    // would make the compiler wait for vmcnt(0), i.e. also for the producers' prefetched loads of the phase after next)
    // Under the HIP memory model the relaxed store is formally a race against the plain LDS writes it publishes; the
    // ordering is carried by the s_waitcnt in front of every call. `make P3_RELEASE_SIGNAL=1` builds the same kernel with a
    // RELEASE store instead (slower, ordering by the compiler): if a toolchain change ever breaks the bit-identity tests
    // (tests/test_gpu_bench_kernels.py: hand_off_is_race_free), that build tells a broken hand-off from anything else.
    auto signal = [&](int *slot, int value) {
        asm volatile("" ::: "memory");
#ifdef LSSVC_P3_RELEASE_SIGNAL
        if (lane == 0) __hip_atomic_store(slot, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
        if (lane == 0) __hip_atomic_store(slot, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
    };

    // parked tile: [consumer wave][row][pixel column][TM floats], the 16-byte channel quads of a pixel rotated by its column so that
    // the 8-lane groups of a ds_write_b128 (8 pixel columns, one quad) spread over the banks
    constexpr int QN = 4 * MF;
    auto tile_origin = [&](int it, int &oy0, int &ox0, int &m0) {
        const int tile = t_begin + kb + it * nb_x;
        const int mt = tile % p.m_tiles, pt = tile / p.m_tiles;
        const int tx = pt % p.tiles_x, ty = pt / p.tiles_x;
        oy0 = ty * G::TH;
        ox0 = tx * 16;
        m0 = mt * TM;
    };

    if (wave >= kP3Consumers) {
        // =================================================================================== PRODUCER waves
        const int lt = tid - 64 * kP3Consumers;                   // 0 .. 255
        const int pw = wave - kP3Consumers;
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
        // padding / past the patch) and the lane offsets of the weight DMA inside one chunk's [hi | lo] image. The patch of
        // phase k+2 is loaded while the weights of phase k+1 are staged, so the two halves are cached per tile separately.
        // SPLIT: 16-byte units of one plane / both planes of the LDS patch image, wave-level DMA instructions, per producer wave
        constexpr int UPP = G::PH * PW * 2, UT = 2 * UPP, P_INSTR = (UT + 63) / 64, NPDMA = (P_INSTR + G::NPROD - 1) / G::NPROD;
        int ppix[NP], woff[NDMA], pcode[SPLIT ? NPDMA : 1];
        int pgeom_it = -1, wgeom_it = -1;
        auto patch_geometry = [&](int it) {
            int oy0, ox0, m0;
            tile_origin(it, oy0, ox0, m0);
            if constexpr (SPLIT) {
                // unit u of the LDS image = (plane, patch position, half of the 16 channels) -> (input pixel << 2 | plane * 2 + half),
                // -1 for zero padding and for the lanes past the image in the last instruction
#pragma unroll
                for (int t = 0; t < NPDMA; ++t) {
                    const int u = (pw + G::NPROD * t) * 64 + lane;
                    const int plane = u >= UPP ? 1 : 0;
                    const int r = u - plane * UPP;
                    const int ppos = r >> 1, half = r & 1;
                    const int py = ppos / PW, pxs = ppos - py * PW;
                    const int px = S == 2 ? (pxs < G::PWE ? 2 * pxs : 2 * (pxs - G::PWE) + 1) : pxs;      // stride 2: de-interleaved columns
                    const int gy = oy0 * S - p.pad_t + py, gx = ox0 * S - p.pad_l + px;
                    const bool ok = u < UT && gy >= 0 && gy < Hin && gx >= 0 && gx < Win;
                    pcode[t] = ok ? (((gy * Win + gx) << 2) | (plane * 2 + half)) : -1;
                }
                pgeom_it = it;
                return;
            }
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int idx = lt + i * PT;
                const int pix = idx >> 2;
                const int py = pix / PW, px = pix - py * PW;
                const int gy = oy0 * S - p.pad_t + py, gx = ox0 * S - p.pad_l + px;
                const bool ok = idx < G::PATCH_ITEMS && gy >= 0 && gy < Hin && gx >= 0 && gx < Win;
                ppix[i] = ok ? gy * Win + gx : -1;
            }
            pgeom_it = it;
        };
        auto weight_geometry = [&](int it) {
            int oy0, ox0, m0;
            tile_origin(it, oy0, ox0, m0);
#pragma unroll
            for (int t = 0; t < NDMA; ++t) {
                int j = ROLES ? t : pw + G::NPROD * t;           // wave-uniform DMA instruction index
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
            wgeom_it = it;
        };
        long long s_dma = 0, s_ld = 0, s_wait = 0, s_cvt = 0, s_bar = 0, s_geo = 0;      // STAMP build only
        // weights of phase `ph` -> LDS buffer `buf`, by DMA
        auto stage_weights = [&](const P3Phase &ph, int buf) {
            long long ts = 0;
            if (STAMP) ts = __builtin_amdgcn_s_memtime();
            if (ph.it != wgeom_it) weight_geometry(ph.it);
            {
                unsigned char *dst = reinterpret_cast<unsigned char *>(wts_buf(buf));
                const _Float16 *src0 = w16 + (size_t)ph.k.kc * NTAP * p.M_pad * CK16;
#pragma unroll
                for (int t = 0; t < NDMA; ++t) {
                    int j = ROLES ? t : pw + G::NPROD * t;
                    if (j >= G::W_INSTR) j = G::W_INSTR - 1;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src0 + woff[t]),
                                                     (__attribute__((address_space(3))) void *)(dst + j * 1024), 16, 0, 0);
                }
            }
            if (STAMP) s_dma += __builtin_amdgcn_s_memtime() - ts;
        };
        // SPLIT: patch of phase `ph` -> LDS buffer `buf`, by DMA
        auto dma_patch = [&](const P3Phase &ph, int buf) {
            if constexpr (SPLIT) {
                long long ts = 0;
                if (STAMP) ts = __builtin_amdgcn_s_memtime();
                if (ph.it != pgeom_it) patch_geometry(ph.it);
                const V X = p.in[ph.k.seg];
                const unsigned char *base = reinterpret_cast<const unsigned char *>(X.p + ph.k.c0);      // this chunk's 64 bytes of pixel 0
                const size_t pitch = (size_t)X.ld * 4;
                unsigned char *dst = reinterpret_cast<unsigned char *>(patch_buf(buf));
#pragma unroll
                for (int t = 0; t < NPDMA; ++t) {
                    const int j = pw + G::NPROD * t;
                    if (j >= P_INSTR) break;
                    const unsigned char *src = pcode[t] >= 0 ? base + (size_t)(pcode[t] >> 2) * pitch + (size_t)((pcode[t] & 3) * 16)
                                                             : reinterpret_cast<const unsigned char *>(g_p3_zero_block);
                    if (j * 64 + lane < UT)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                         (__attribute__((address_space(3))) void *)(dst + j * 1024), 16, 0, 0);
                }
                if (STAMP) s_ld += __builtin_amdgcn_s_memtime() - ts;
            }
        };
        // patch of phase `ph` -> registers (fp32, as loaded)
        float4 preg[NP];
        unsigned pmask = 0;
        auto load_patch_to = [&](const P3Phase &ph, float4 (&preg)[NP], unsigned &pmask) __attribute__((always_inline)) {
            pmask = 0;
            long long ts = 0;
            if (STAMP) ts = __builtin_amdgcn_s_memtime();
            if (ph.it != pgeom_it) patch_geometry(ph.it);
            if (STAMP) {
                const long long t = __builtin_amdgcn_s_memtime();
                s_geo += t - ts;
                ts = t;
            }
            const V X = p.in[ph.k.seg];
            const bool cvalid = quad4 < X.C - ph.k.c0;
            const int cc = cvalid ? ph.k.c0 + quad4 : 0;
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const bool ok = ppix[i] >= 0 && cvalid;
                const size_t off = ok ? (size_t)ppix[i] * X.ld + cc : 0;
                preg[i] = *reinterpret_cast<const float4 *>(X.p + off);
                pmask |= ok ? (1u << i) : 0u;
            }
            if (STAMP) s_ld += __builtin_amdgcn_s_memtime() - ts;
        };
        auto load_patch = [&](const P3Phase &ph) __attribute__((always_inline)) { load_patch_to(ph, preg, pmask); };
        // registers -> fp16 hi / lo planes of LDS buffer `buf`
        auto store_patch_from = [&](int buf, const float4 (&preg)[NP], const unsigned pmask) __attribute__((always_inline)) {
            long long ts = 0;
            if (STAMP) ts = __builtin_amdgcn_s_memtime();
            _Float16 *ph_ = patch_buf(buf);
            _Float16 *pl_ = ph_ + G::PATCH_HALFS;
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int idx = lt + i * PT;
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
                // items past the patch (last round only) go to the lane's trash slot: a guarded store makes the compiler sink
                // that round's global load INTO the guarded block, behind every other store, and two of the four producer
                // waves then sit out a full memory latency at the end of every fill
                const bool in_patch = i + 1 < NP || idx < G::PATCH_ITEMS;
                int ppos = idx >> 2;                               // patch pixel py * PW + px ...
                if (S == 2) {                                      // ... stride 2: even columns first, then the odd ones (P3Geom)
                    const int py = ppos / PW, px = ppos - py * PW;
                    ppos = py * PW + (px & 1) * G::PWE + (px >> 1);
                }
                const int o = ppos * CK16 + quad4;
                *reinterpret_cast<f16x4 *>(in_patch ? ph_ + o : trash_s + (lt % TRASH_LANES) * 4) = h;
                *reinterpret_cast<f16x4 *>(in_patch ? pl_ + o : trash_s + (TRASH_LANES + lt % TRASH_LANES) * 4) = l;
            }
            if (STAMP) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                s_cvt += __builtin_amdgcn_s_memtime() - ts;
            }
        };
        auto store_patch = [&](int buf) __attribute__((always_inline)) { store_patch_from(buf, preg, pmask); };
        // STAGE, at a tile boundary: the patch registers are made plain values here (the empty asm reads them, so the compiler
        // waits for their loads in front of it) -- behind the drain, whose stores are younger than these loads, its wait for
        // them would also be a wait for the stores' acknowledgements.
        auto settle_patch = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < NP; ++i) asm volatile("" : "+v"(preg[i].x), "+v"(preg[i].y), "+v"(preg[i].z), "+v"(preg[i].w));
        };
        // STAGE: the rows consumer wave `pw` parked for tile `it` in operand pair b -> global memory (+ residuals; plain or
        // pixel-shuffle store): conv_epilogue_fast_impl's arithmetic after its lane transposition, element for element, in
        // THREE steps so that nothing the consumers wait for waits for a store:
        //   drain_prefetch  residual loads (with the next fill's patch loads, before the tile is even parked)
        //   drain_read      parked rows -> registers, + residual; after it the pair may be overwritten
        //   drain_store     -> global; issued AFTER the fill has been signalled: loads, stores and the weight
        //                   DMA share one in-order counter, so a wait for the DMA behind these stores would also wait for their
        //                   acknowledgement, which takes as long as the burst of all 256 CUs' tiles takes to reach memory
        constexpr int NI = STAGE ? SR * MF : 1;               // 16-byte items per lane: SR rows x 16 pixels x QN quads over 64 lanes
        constexpr int NH = (NI + 1) / 2;                      // residuals are held for half of them at a time (registers)
        f32x4 dv[NI];
        float4 drs[NH];
        int d_oy0 = 0, d_ox0 = 0, d_m0 = 0;
        // (each step recomputes its addresses from an OPAQUE copy of the lane id: the compiler would otherwise keep the 20 pixel
        // and channel offsets of the prefetch alive for the later steps, ~60 registers that push the producers into scratch)
        auto opaque_lane = [&]() __attribute__((always_inline)) {
            int ln = lane;
            asm volatile("" : "+v"(ln));
            return ln;
        };
        auto d_item = [&](int ln, int i, int &qd, int &col, int &r) __attribute__((always_inline)) {
            const int idx = ln + 64 * i;
            qd = idx % QN;
            const int pc = idx / QN;
            col = pc & 15;
            r = pc >> 4;
        };
        auto d_valid = [&](int qd, int col, int r, unsigned &px, unsigned &mt) __attribute__((always_inline)) {
            const int oy = d_oy0 + pw * RPW + r, ox = d_ox0 + col, m = d_m0 + 4 * qd;
            const bool v = oy < p.Hout && ox < p.Wout && m < p.Cout;
            px = v ? (unsigned)oy * (unsigned)p.Wout + (unsigned)ox : 0u;
            mt = v ? (unsigned)m : 0u;
            return v;
        };
        // residuals of items [i0, i0 + NH) of `src` -> drs
        auto d_load_res = [&](const V &src, int i0) __attribute__((always_inline)) {
            const int ln = opaque_lane();
#pragma unroll
            for (int j = 0; j < NH; ++j) {
                if (i0 + j >= NI) break;
                int qd, col, r;
                unsigned px, mt;
                d_item(ln, i0 + j, qd, col, r);
                d_valid(qd, col, r, px, mt);
                drs[j] = *reinterpret_cast<const float4 *>(src.p + (size_t)(px * (unsigned)src.ld + mt));
            }
        };
        auto d_add_res = [&](int i0) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < NH; ++j)
                if (i0 + j < NI) dv[i0 + j] = dv[i0 + j] + f32x4{drs[j].x, drs[j].y, drs[j].z, drs[j].w};
        };
        auto drain_prefetch = [&](int it) __attribute__((always_inline)) {
            if constexpr (STAGE) {
                tile_origin(it, d_oy0, d_ox0, d_m0);
                if ((p.debug & 32) || !p.res.p) return;
                d_load_res(p.res, 0);                         // the first half's residuals: in flight while the consumers park the tile
            }
        };
        auto drain_read = [&](int b) __attribute__((always_inline)) {
            if constexpr (STAGE) {
                if (p.debug & 32) return;
                const unsigned sbase = (unsigned)(size_t)(lds_cfloat_ptr)(const float *)(const void *)smem + stage_off(b) + (unsigned)(pw * SR * 16 * TM * 4);
                const int ln = opaque_lane();
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    int qd, col, r;
                    d_item(ln, i, qd, col, r);
                    const unsigned a = sbase + (unsigned)(((r * 16 + col) * TM + ((qd + col) % QN) * 4) * 4);
                    dv[i] = *reinterpret_cast<const __attribute__((address_space(3))) f32x4 *>((size_t)a);
                }
                if (p.res.p) {
                    d_add_res(0);
                    d_load_res(p.res, NH);
                    d_add_res(NH);
                }
                if (p.res2.p) {                               // second residual (few convs): added after the first, as the direct epilogue does
                    d_load_res(p.res2, 0);
                    d_add_res(0);
                    d_load_res(p.res2, NH);
                    d_add_res(NH);
                }
            }
        };
        auto drain_store = [&]() __attribute__((always_inline)) {
            if constexpr (STAGE) {
                if (p.debug & 32) return;
                const bool ps = p.fast_epi == 2;
                const unsigned cps = (unsigned)(p.Cout >> 2);
                const int ln = opaque_lane();
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    int qd, col, r;
                    unsigned px, mt;
                    d_item(ln, i, qd, col, r);
                    const bool v = d_valid(qd, col, r, px, mt);
                    unsigned o = px * (unsigned)p.out.ld + mt;       // element offsets (the host keeps tensors of >= 2^32 elements off this path)
                    if (ps) {                                 // channel m = q * cps + c goes to sub-pixel q = dy * 2 + dx, channel c
                        const unsigned oy = px / (unsigned)p.Wout, ox = px - oy * (unsigned)p.Wout;
                        const unsigned q = mt / cps, c = mt - q * cps;
                        o = ((2u * oy + (q >> 1)) * (unsigned)p.out.W + 2u * ox + (q & 1u)) * (unsigned)p.out.ld + c;
                    }
                    if (v) *reinterpret_cast<float4 *>(p.out.p + (size_t)o) = make_float4(dv[i][0], dv[i][1], dv[i][2], dv[i][3]);
                }
            }
        };

        // Schedule: fill(k+1) = weights by DMA + patch through registers, while the consumers run phase k. (Requesting the patch
        // of phase k+2 before the buffer of phase k+1 is released -- only the LDS writes need the buffer -- was tried: the
        // producers then had 2 k cycles of slack per phase, the consumers waited as long as before and the extra loads in
        // flight slowed the epilogue's stores; 64->64 @1080p 428 -> 458 us.)
        const int total = n_it * phases_per_tile;
        P3Phase ph{0, KState{0, 0, 0, 0}};
        // the patch part of a fill: through registers (load, convert, ds_write) or, SPLIT, by DMA
        auto fill_patch = [&](const P3Phase &f, int buf) __attribute__((always_inline)) {
            if constexpr (SPLIT) {
                dma_patch(f, buf);
            } else {
                load_patch(f);
                store_patch(buf);
            }
        };
        if constexpr (NPB == 3 && !ROLES) {
            // ---- PATCH RING schedule: weights(k+1) and patch(k+2) while the consumers run phase k
            P3Phase php = ph;                               // the phase whose patch is filled next
            stage_weights(ph, 0);
            fill_patch(php, 0);
            if (total > 1) {
                php = next_phase(php);
                fill_patch(php, 1);
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __syncthreads();                               // (A) phase 0 is in weight buffer 0 / patch buffer 0, the patch of phase 1 in patch buffer 1
            int pb = 2;                                    // patch buffer of phase k+2
            for (int k = 0; k + 1 < total; ++k) {
                long long tb = 0;
                if (STAMP) tb = __builtin_amdgcn_s_memtime();
                if (k >= 1) wait_for(sync_s + 4, k);       // phase k-1 is over: weight buffer (k+1)&1 and patch buffer (k+2)%3 = (k-1)%3 are free
                if (STAMP) s_bar += __builtin_amdgcn_s_memtime() - tb;
                ph = next_phase(ph);
                const bool ablate = STAMP && (p.debug & 7);
                if (!(ablate && (p.debug & 2))) stage_weights(ph, (k + 1) & 1);
                asm volatile("" ::: "memory");             // the counted wait below relies on this order: every weight DMA is OLDER than every patch request
                const bool more = k + 2 < total && !(ablate && (p.debug & 1));
                if (more) {
                    php = next_phase(php);
                    if constexpr (SPLIT) dma_patch(php, pb);
                    else load_patch(php);
                }
                // fill(k+1) = the weights just requested + the patch of phase k+1, whose LDS stores were waited for at the end of the previous
                // iteration: signalled as soon as the weight DMA has landed -- the NP (SPLIT: NPDMA) younger patch requests stay in flight
                if (more) {
                    if constexpr (SPLIT) asm volatile("s_waitcnt vmcnt(%0) expcnt(6)" ::"n"(P_INSTR % G::NPROD == 0 ? NPDMA : NPDMA - 1) : "memory");      // (some waves issue one patch DMA fewer)
                    else asm volatile("s_waitcnt vmcnt(%0) expcnt(6)" ::"n"(NP) : "memory");      // (expcnt(6): the counted-wait mark, see p3_waitcnt)
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                // (published by a ds_write written in asm: before a compiler-visible LDS store the compiler drains vmcnt(0) as long as an
                // LDS-DMA is in flight -- the wait above has just named the DMA that matters -- and the patch loads would be waited for too)
#ifdef LSSVC_P3_RELEASE_SIGNAL
                signal(sync_s + pw, k + 1);                // (diagnostic build: a RELEASE store, ordering by the compiler -- it drains every load first)
#else
                if (lane == 0) {
                    const unsigned slot_addr = (unsigned)(size_t)(__attribute__((address_space(3))) int *)(sync_s + pw);
                    asm volatile("ds_write_b32 %0, %1" ::"v"(slot_addr), "v"(k + 1) : "memory");
                }
#endif
                if (more) {
                    if constexpr (!SPLIT) {
                        if (!(ablate && (p.debug & 4))) store_patch(pb);
                        else
#pragma unroll
                            for (int i = 0; i < NP; ++i) asm volatile("" ::"v"(preg[i].x), "v"(preg[i].y), "v"(preg[i].z), "v"(preg[i].w));
                    }
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // the patch of phase k+2 is in the LDS before fill(k+2) is signalled
                    pb = pb == 2 ? 0 : pb + 1;
                }
            }
            if (STAMP && lane == 0 && p.gdn_x.p) {
                long long *o = reinterpret_cast<long long *>(p.gdn_x.p) + ((size_t)gridDim.x * kP3Consumers + (size_t)blockIdx.x * 4 + pw) * 8;
                o[0] = s_dma; o[1] = s_ld; o[2] = s_wait; o[3] = s_cvt; o[4] = s_bar; o[5] = total; o[6] = s_geo; o[7] = 0;
            }
            return;
        }
        if constexpr (ROLES) {
            if (pw == G::NPROD - 1) {
                // ---- the DMA wave: all of every phase's weights, nothing else
                stage_weights(ph, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();                           // (A)
                for (int k = 0; k + 1 < total; ++k) {
                    if (k >= 1) wait_for(sync_s + 4, k);   // weight buffer (k+1)&1 was read in phase k-1
                    ph = next_phase(ph);
                    if constexpr (!(kP3Ablate & 1)) stage_weights(ph, (k + 1) & 1);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    signal(sync_s + pw, k + 1);
                }
                return;
            }
            // ---- the patch waves: two register sets, the patch of phase k+2 requested before the patch of phase k+1 is converted
            float4 preg2[NP];
            unsigned pmask2 = 0;
            P3Phase php = ph;
            load_patch_to(php, preg, pmask);
            store_patch_from(0, preg, pmask);
            if (total > 1) {
                php = next_phase(php);
                load_patch_to(php, preg, pmask);           // patch of phase 1: in flight across barrier (A)
            }
            __builtin_amdgcn_s_waitcnt(p3_waitcnt(63, 0)); // this wave's LDS stores of phase 0 are done (no wait for the loads)
            __syncthreads();                               // (A)
            int pb = 1;                                    // patch buffer of fill k+1: (k+1) & 1, or (k+1) % 3 with the ring
            auto step = [&](int k, float4 (&cur)[NP], unsigned &cmask, float4 (&nxt)[NP], unsigned &nmask, auto load) __attribute__((always_inline)) {
                if constexpr (NPB == 3) {
                    if (k >= 2) wait_for(sync_s + 4, k - 1);      // ring: patch buffer (k+1) % 3 was read in phase k-2
                } else {
                    if (k >= 1) wait_for(sync_s + 4, k);   // patch buffer (k+1)&1 was read in phase k-1
                }
                if constexpr (decltype(load)::value) {
                    if (k + 2 < total) php = next_phase(php);
                    if constexpr (!(kP3Ablate & 4)) {
                        load_patch_to(php, nxt, nmask);    // (unconditional: see the PF2 schedule)
                        __builtin_amdgcn_sched_barrier(0);
                        __builtin_amdgcn_s_waitcnt(p3_waitcnt(NP, 15));      // the patch of phase k+1 has landed; the NP new loads stay in flight
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
                    __builtin_amdgcn_s_waitcnt(p3_waitcnt(0, 15));
                }
                if constexpr (!(kP3Ablate & 8)) store_patch_from(pb, cur, cmask);
                else {
#pragma unroll
                    for (int i = 0; i < NP; ++i) asm volatile("" ::"v"(cur[i].x), "v"(cur[i].y), "v"(cur[i].z), "v"(cur[i].w));      // (the loads stay alive)
                }
                __builtin_amdgcn_s_waitcnt(p3_waitcnt(63, 0));           // this wave's LDS stores are done
                signal(sync_s + pw, k + 1);
                pb = pb + 1 == NPB ? 0 : pb + 1;
            };
            int k = 0;
            for (; k + 2 < total; k += 2) {
                step(k, preg, pmask, preg2, pmask2, std::true_type{});
                step(k + 1, preg2, pmask2, preg, pmask, std::true_type{});
            }
            if (k + 1 < total) step(k, preg, pmask, preg2, pmask2, std::false_type{});
            return;
        }
        if constexpr (PAIR) {
            // ---- PAIR schedule (see the kernel's PAIR note). Fill f = k + 1 while the consumers run phase k; when phase f + 1 is the next
            // chunk of the same tensor in the same tile, its patch is loaded beside phase f's and fill f + 1 needs no load.
            float4 preg2[NP];
            unsigned pmask2 = 0;
            auto publish = [&](int fill) __attribute__((always_inline)) {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // this wave's DMA has landed, its LDS stores are done
                signal(sync_s + pw, fill);
            };
            stage_weights(ph, 0);
            P3Phase nx = total > 1 ? next_phase(ph) : ph;
            bool paired = total > 1 && nx.it == ph.it && nx.k.seg == ph.k.seg;
            load_patch_to(ph, preg, pmask);
            if (paired) load_patch_to(nx, preg2, pmask2);
            store_patch_from(0, preg, pmask);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __syncthreads();                               // (A) phase 0 is in buffer 0
            int k = 0;
            while (k + 1 < total) {
                if (k >= 1) wait_for(sync_s + 4, k);       // buffer (k+1)&1 was read in phase k-1: every consumer has left it
                ph = next_phase(ph);                       // phase k+1
                stage_weights(ph, (k + 1) & 1);
                if (paired) {                              // its patch came with the previous fill's
                    store_patch_from((k + 1) & 1, preg2, pmask2);
                    paired = false;
                } else {
                    nx = k + 2 < total ? next_phase(ph) : ph;
                    paired = k + 2 < total && nx.it == ph.it && nx.k.seg == ph.k.seg;
                    load_patch_to(ph, preg, pmask);
                    if (paired) load_patch_to(nx, preg2, pmask2);
                    store_patch_from((k + 1) & 1, preg, pmask);
                }
                publish(k + 1);
                ++k;
            }
            return;
        }
        if constexpr (LATE) {
            // ---- LATE-LOADS schedule (PF = 4, round 6; for the MFMA-bound tilings). The plain schedule below starts a fill when the consumers
            // release its buffer and then REQUESTS the patch: weight-DMA issue + patch-load issue + memory latency + conversion = 4.9 k cycles
            // (stamps) in front of the signal, against a consumer phase of 6.0 k -- the consumers waited 11.7 % of their cycles for fills. Here
            // the patch of phase k+1 is requested right AFTER fill(k) has been signalled and sits in registers while the consumers run phase
            // k-1's successor; once they release the buffer only the conversion, the weight DMA and its latency are in front of the signal.
            // One register set; no LDS-DMA is ever outstanding while a loaded register is used (the DMA is issued after the patch stores and
            // drained before the signal), so the compiler's own waits are the exact ones.
            P3Phase php = ph;
            stage_weights(ph, 0);
            fill_patch(ph, 0);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            if (total > 1) php = next_phase(php);
            load_patch_to(php, preg, pmask);               // patch of phase 1 (a one-phase launch: phase 0 once more, to registers nobody reads)
            __syncthreads();                               // (A) phase 0 is in buffer 0
            for (int k = 0; k + 1 < total; ++k) {
                long long tb = 0;
                if (STAMP) tb = __builtin_amdgcn_s_memtime();
                if (k >= 1) wait_for(sync_s + 4, k);       // buffer (k+1)&1 was read in phase k-1: every consumer has left it
                if (STAMP) s_bar += __builtin_amdgcn_s_memtime() - tb;
                ph = next_phase(ph);
                __builtin_amdgcn_s_waitcnt(p3_waitcnt(0, 15));            // the patch of phase k+1, requested an iteration ago, has landed
                store_patch_from((k + 1) & 1, preg, pmask);
                stage_weights(ph, (k + 1) & 1);
                long long tw = 0;
                if (STAMP) tw = __builtin_amdgcn_s_memtime();
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // this wave's DMA has landed, its LDS stores are done
                if (STAMP) s_wait += __builtin_amdgcn_s_memtime() - tw;       // (stamps: the weight DMA's latency)
                signal(sync_s + pw, k + 1);
                if (k + 2 < total) php = next_phase(php);
                load_patch_to(php, preg, pmask);           // (unconditional: one definition of the register set per iteration, see the PF2 schedule)
            }
            if (STAMP && lane == 0 && p.gdn_x.p) {
                long long *o = reinterpret_cast<long long *>(p.gdn_x.p) + ((size_t)gridDim.x * kP3Consumers + (size_t)blockIdx.x * 4 + pw) * 8;
                o[0] = s_dma; o[1] = s_ld; o[2] = s_wait; o[3] = s_cvt; o[4] = s_bar; o[5] = total; o[6] = s_geo; o[7] = 0;
            }
            return;
        }
        if constexpr (PF2) {
            // ---- REGISTER PREFETCH schedule (see the kernel's PF2 note): while the consumers run phase k, fill(k+1) = weights by DMA + the
            // patch that has been in flight in one register set since the previous iteration; the loads of phase k+2 are requested into the
            // other set BEFORE that patch is converted, so a whole phase's patch is in flight per CU at any time.
            float4 preg2[NP];
            unsigned pmask2 = 0;
            P3Phase php = ph;                               // the phase whose patch is requested next
            stage_weights(ph, 0);
            load_patch_to(php, preg, pmask);
            store_patch_from(0, preg, pmask);
            if (total > 1) {
                php = next_phase(php);
                load_patch_to(php, preg, pmask);            // patch of phase 1: in flight across barrier (A)
                __builtin_amdgcn_s_waitcnt(p3_waitcnt(NP, 0));                         // the weight DMA and the loads of phase 0 are older than these NP loads
            } else {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            }
            __syncthreads();                               // (A) phase 0 is in buffer 0
            // (`load`: a compile-time flag. Inside the loop both steps load UNCONDITIONALLY -- past the last phase the last patch once more, to
            // registers nobody reads -- so that each register set has ONE definition per iteration: with the load under a condition the set
            // becomes a merge of old and new values at the loop's back edge, the compiler copies registers there, and a copy of a register
            // that is the target of a load in flight waits for vmcnt(0) -- the very loads this schedule keeps in flight.)
            auto step = [&](int k, float4 (&cur)[NP], unsigned &cmask, float4 (&nxt)[NP], unsigned &nmask, auto load) __attribute__((always_inline)) {
                if (k >= 1) wait_for(sync_s + 4, k);       // buffer (k+1)&1 was read in phase k-1: every consumer has left it
                ph = next_phase(ph);
                stage_weights(ph, (k + 1) & 1);
                asm volatile("" ::: "memory");             // the counted wait below relies on this order: the weight DMA is OLDER than the new patch requests
                if constexpr (decltype(load)::value) {
                    if (k + 2 < total) php = next_phase(php);
                    load_patch_to(php, nxt, nmask);
                    // the patch of phase k+1 (cur) and the weight DMA have landed; the NP new loads stay in flight. A BUILTIN wait, not asm text:
                    // the compiler's own counter model must see it, or it drains vmcnt(0) in front of the next LDS access of this wave
                    // (the slot poll, the patch stores) for as long as it believes an LDS-DMA may be outstanding
                    __builtin_amdgcn_sched_barrier(0);         // (no part of the conversion below may be scheduled in front of the wait: the compiler would
                    __builtin_amdgcn_s_waitcnt(p3_waitcnt(NP, 15));    // guard it with a vmcnt(0) of its own, in the middle of the new requests)
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                    __builtin_amdgcn_s_waitcnt(p3_waitcnt(0, 15));
                }
                store_patch_from((k + 1) & 1, cur, cmask);
                __builtin_amdgcn_s_waitcnt(p3_waitcnt(63, 0));                      // this wave's LDS stores are done
                // (published by a ds_write written in asm, as in the PATCH RING schedule: a compiler-visible LDS store would drain vmcnt(0),
                // i.e. wait for the loads of phase k+2 as well)
#ifdef LSSVC_P3_RELEASE_SIGNAL
                signal(sync_s + pw, k + 1);
#else
                if (lane == 0) {
                    const unsigned slot_addr = (unsigned)(size_t)(__attribute__((address_space(3))) int *)(sync_s + pw);
                    asm volatile("ds_write_b32 %0, %1" ::"v"(slot_addr), "v"(k + 1) : "memory");
                }
#endif
            };
            int k = 0;
            for (; k + 2 < total; k += 2) {                // steps k and k+1 both exist
                step(k, preg, pmask, preg2, pmask2, std::true_type{});
                step(k + 1, preg2, pmask2, preg, pmask, std::true_type{});
            }
            if (k + 1 < total) step(k, preg, pmask, preg2, pmask2, std::false_type{});      // the last fill of an even phase count
            return;
        }
        stage_weights(ph, 0);
        fill_patch(ph, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the weight DMA of this wave has landed
        __syncthreads();                                   // (A) phase 0 is in buffer 0
        int drained = 0;                                   // STAGE: tiles this wave has taken out of the LDS
        for (int k = 0; k + 1 < total; ++k) {              // fill(k+1) while the consumers run phase k
            long long tb = 0;
            if (STAMP) tb = __builtin_amdgcn_s_memtime();
            if (k >= 1) wait_for(sync_s + 4, k);           // buffer (k+1)&1 was read in phase k-1: every consumer has left it
            if (STAMP) s_bar += __builtin_amdgcn_s_memtime() - tb;
            ph = next_phase(ph);
            const bool boundary = STAGE && k >= 1 && k % phases_per_tile == 0;
            long long tw = 0;
            if (STAMP && boundary) tw = __builtin_amdgcn_s_memtime();
            if (boundary) {
                // phase k-1 closed a tile: it is parked in the very pair this fill is about to overwrite. The tile's residual loads
                // first (registers only); the parked rows into registers; the patch loads; when all four producer waves have their
                // rows, the DMA and the patch stores may touch the pair; the tile's stores go out after the fill has been signalled.
                drain_prefetch(drained);
                wait_for(sync_s + 8, drained + 1);
                drain_read((k + 1) & 1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's reads of the parked rows have returned
                ++drained;
                signal(sync_s + 12 + pw, drained);
                if constexpr (!SPLIT) load_patch(ph);                  // (after the residuals have been added: 80 registers fewer in flight)
                wait_for(sync_s + 12, drained);
                stage_weights(ph, (k + 1) & 1);
                if constexpr (SPLIT) dma_patch(ph, (k + 1) & 1);
                else store_patch((k + 1) & 1);
            } else if (STAMP && (p.debug & 7)) {
                // ablations of the stamp build (LSSVC_CONV_DEBUG = 256 + bits; results are wrong): 1 no patch traffic after the first
                // fill, 2 no weight DMA after the first fill, 4 patch loads only (no conversion, no LDS stores)
                if (!(p.debug & 2)) stage_weights(ph, (k + 1) & 1);
                if (!(p.debug & 1)) {
                    if constexpr (SPLIT) {
                        dma_patch(ph, (k + 1) & 1);
                    } else {
                        load_patch(ph);
                        if (!(p.debug & 4)) store_patch((k + 1) & 1);
                        else
#pragma unroll
                            for (int i = 0; i < NP; ++i) asm volatile("" ::"v"(preg[i].x), "v"(preg[i].y), "v"(preg[i].z), "v"(preg[i].w));
                    }
                }
            } else {
                stage_weights(ph, (k + 1) & 1);
                fill_patch(ph, (k + 1) & 1);
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // this wave's DMA has landed, its LDS stores are done
            signal(sync_s + pw, k + 1);
            if (STAMP && boundary) s_wait += __builtin_amdgcn_s_memtime() - tw;      // (STAGE stamps: cycles of the boundary fills, wait for the consumers excluded)
            if (boundary) drain_store();
        }
        if (STAGE) {                                       // the tiles parked after the last fill (the last one, two if a tile is one phase)
            while (drained < n_it) {
                drain_prefetch(drained);
                wait_for(sync_s + 8, drained + 1);
                drain_read(((drained + 1) * phases_per_tile - 1) & 1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                ++drained;
                signal(sync_s + 12 + pw, drained);
                drain_store();
            }
        }
        if (STAMP && lane == 0 && p.gdn_x.p) {
            long long *o = reinterpret_cast<long long *>(p.gdn_x.p) + ((size_t)gridDim.x * kP3Consumers + (size_t)blockIdx.x * 4 + pw) * 8;
            o[0] = s_dma; o[1] = s_ld; o[2] = s_wait; o[3] = s_cvt; o[4] = s_bar; o[5] = total; o[6] = s_geo; o[7] = 0;
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

    __syncthreads();                                       // (A)
    int it = 0, kt = 0;                                    // tile index in this workgroup's sequence, phase inside the tile
    int pring = 0;                                         // PATCH RING: k % 3
    const int total = n_it * phases_per_tile;
    long long t_comp = 0, t_bar = 0, t_epi = 0, t_zero = 0, t_mark = 0, t_real0 = 0, t_cyc0 = 0;
    if (STAMP) {
        t_real0 = __builtin_amdgcn_s_memrealtime();
        t_cyc0 = t_mark = __builtin_amdgcn_s_memtime();
    }
    for (int k = 0; k < total; ++k) {
        int fill_seen = 0;                                 // producers' slots as read during the last unit of this phase
        const int buf = k & 1;
        const _Float16 *ph_ = patch_buf(NPB == 3 ? pring : buf);           // hi plane; the lo plane follows it (PATCH RING: buffer k % 3)
        const _Float16 *wh_ = wts_buf(buf);
        if (NPB == 3) pring = pring == 2 ? 0 : pring + 1;
        // The phase as NSTEP x NG units (K step u, row group g of GR rows), software-pipelined and INTERLEAVED by hand.
        // In-kernel stamps: the straightforward loop takes 6.6 k cycles per phase for 336 MFMAs = 5.4 k issue cycles, with
        // the producers idle or not and with the fragment reads prefetched or not -- it is neither LDS latency nor
        // producer interference but ISSUE ORDER: a wave issues in order, an MFMA holds the issue port for 8 of its 16
        // cycles, so two 4-cycle instructions fit in every MFMA gap for free and a third delays the next MFMA. The
        // compiler emits the ~95 ds_reads and ~60 address instructions of a phase in clumps. Here the fragment reads of
        // unit t+1 (two register sets, every index a compile-time constant) are written before the MFMAs of unit t and
        // sched_group_barrier lays the unit out as (2 MFMA, 1 ds_read) x reads, then the remaining MFMAs.
        constexpr int GR = RPW >= 2 ? 2 : 1, NG = RPW / GR, NUNIT = NG * NSTEP;
        static_assert(RPW % GR == 0, "row groups");
        f16x8 fa1[2][MF], fa2[2][MF], fb1[2][GR], fb2[2][GR];
        // Addresses: everything that depends on the lane is folded into ONE byte offset per operand and K step (the tap of
        // a lane group is 2u + tsel, so the tap part is a per-lane select between two constants); fragment f / row r are
        // compile-time byte offsets that the ds_read carries as its immediate. Base pointers as LDS byte addresses.
        const unsigned a_lane = (unsigned)(li * CK16 + ch8) * 2u;                                  // bytes
        const unsigned b_lane = (unsigned)(((wave * RPW * S) * PW + li) * CK16 + ch8) * 2u;
        const unsigned wh_b = (unsigned)(size_t)(lds_cfloat_ptr)(const float *)(const void *)wh_;   // LDS byte address of the hi plane
        const unsigned wl_b = wh_b + (unsigned)G::W_HALFS * 2u;
        const unsigned ph_b = (unsigned)(size_t)(lds_cfloat_ptr)(const float *)(const void *)ph_;
        const unsigned pl_b = ph_b + (unsigned)G::PATCH_HALFS * 2u;
        auto lds_read = [](unsigned addr) {
            return *reinterpret_cast<const __attribute__((address_space(3))) f16x8 *>((size_t)addr);
        };
        auto load_a = [&](int u, f16x8 (&a1)[MF], f16x8 (&a2)[MF]) {
            const bool odd = 2 * u + 1 >= NTAP;                              // the ninth tap: f16x3_step_odd
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
            const int tap0 = 2 * u, tap1 = odd ? tap0 : 2 * u + 1;
            // tap (ky, kx) -> patch offset of fragment column 0: stride 1 column kx; stride 2 column kx of the de-interleaved row
            auto tap_off = [](int tap) { const int ky = tap / 3, kx = tap % 3; return (unsigned)((ky * PW + (S == 2 ? (kx & 1) * G::PWE + (kx >> 1) : kx)) * CK16) * 2u; };
            const unsigned o0 = tap_off(tap0), o1 = tap_off(tap1);
            const unsigned tap_b = b_lane + (tsel ? o1 : o0);
            const unsigned p1 = ((odd && tsel) ? pl_b : ph_b) + tap_b, p2 = pl_b + tap_b;
#pragma unroll
            for (int r = 0; r < GR; ++r) {
                const unsigned ro = (unsigned)((g * GR + r) * S * PW * CK16) * 2u;
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
            // B fragments are prefetched one unit ahead, the A fragments of K step u+1 during row group NG-2 of step u: a
            // unit that ends in a burst of reads would make the NEXT unit's first MFMA wait for them
            constexpr bool pre_a = (NG >= 2 ? g == NG - 2 : true) && u + 1 < NSTEP;
            constexpr int NR = (pre_a ? 2 * MF : 0) + (more ? (nodd ? GR : 2 * GR) : 1);      // ds_reads issued in this unit
            constexpr int NM = (odd ? 2 : 3) * MF * GR;                                       // MFMAs of this unit
            if (pre_a) load_a(u + 1, fa1[(u + 1) & 1], fa2[(u + 1) & 1]);
            if (more) load_b(nu, ng, fb1[(t + 1) & 1], fb2[(t + 1) & 1]);
            // last unit: nothing left to prefetch in this buffer -- look at the producers' slots for the NEXT phase instead, so
            // that the round trip of that read (several hundred cycles with the LDS queue full) hides behind this unit's MFMAs
            if (!more) fill_seen = __hip_atomic_load(sync_s + (lane & 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
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
            // issue order of this unit: (2 MFMA, 1 ds_read) per prefetched read, then the rest of the MFMAs
            constexpr int MPR = LSSVC_P3_MFMA_PER_READ;
            constexpr int NI = (MPR * NR <= NM) ? NR : NM / MPR;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, MPR, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            if (NR > NI) __builtin_amdgcn_sched_group_barrier(0x100, NR - NI, 0);
            if (NM > MPR * NI) __builtin_amdgcn_sched_group_barrier(0x008, NM - MPR * NI, 0);
            __builtin_amdgcn_sched_barrier(0);                                                 // units do not mix
        };
        if constexpr (!(ROLES && (kP3Ablate & 2))) {
        load_b(0, 0, fb1[0], fb2[0]);                      // B first: the first MFMA needs b1[0] and a2[0], LDS returns in order,
        load_a(0, fa1[0], fa2[0]);                         // so it can start after 6 of these 12 reads instead of 10
        __builtin_amdgcn_sched_barrier(0);
        [&]<int... I>(std::integer_sequence<int, I...>) { (unit(std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, NUNIT>{});
        }
        if (STAMP) {
            const long long t = __builtin_amdgcn_s_memtime();
            t_comp += t - t_mark;
            t_mark = t;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (every fragment read of this phase has returned: the MFMAs consumed them)
        signal(sync_s + 4 + wave, k + 1);                  // this wave is done with buffer k & 1
        if (k + 1 < total) {                               // every producer wave has finished fill(k+1)? (normally long ago)
            if (__builtin_amdgcn_ballot_w64(fill_seen < k + 1) != 0) wait_for(sync_s + 0, k + 1);
            asm volatile("" ::: "memory");                 // the next phase's fragment reads stay behind this check
            if (STAMP) {
                const long long t = __builtin_amdgcn_s_memtime();
                t_bar += t - t_mark;
                t_mark = t;
            }
        }
        if (++kt == phases_per_tile) {
            int oy0, ox0, m0;
            tile_origin(it, oy0, ox0, m0);
            const int oy_w = oy0 + wave * RPW;
            auto pix = [&](int r, int col) {              // pixel index of column col of this wave's row r, -1 outside the image
                return (ox0 + col < p.Wout && oy_w + r < p.Hout) ? (long long)(oy_w + r) * p.Wout + ox0 + col : -1LL;
            };
            // interior: this wave's rows, the tile's 16 columns and its 16*MF channels all lie inside the output
            const bool interior = oy0 + wave * RPW + RPW <= p.Hout && ox0 + 16 <= p.Wout && m0 + TM <= p.Cout;
            if constexpr (STAGE) {
                // Park the tile in the operand pair of the phase just computed (+ the free middle of the LDS): every consumer
                // must have left that pair (they all read its weights), and the previous tile must be out of the LDS (the two
                // parking areas share part of the middle; normally long done: its drain ran a whole tile ago).
                wait_for(sync_s + 4, k + 1);
                wait_for(sync_s + 12, it);
                if (!(p.debug & 32)) {
                    const float us = p.w16_unscale;
                    const float s_neg = p.act == LSSVC_ACT_LRELU ? p.slope : (p.act == LSSVC_ACT_RELU ? 0.0f : 1.0f);
                    const bool act = p.act != LSSVC_ACT_NONE;
                    const unsigned sb = (unsigned)(size_t)(lds_cfloat_ptr)(const float *)(const void *)smem + stage_off(buf) +
                                        (unsigned)((wave * SR * 16 + li) * TM * 4);
#pragma unroll
                    for (int f = 0; f < MF; ++f) {
                        const f32x4 bv = *reinterpret_cast<__attribute__((address_space(3))) const f32x4 *>((lds_cfloat_ptr)bias_s + m0 + f * 16 + 4 * lg);
                        const unsigned qoff = (unsigned)(((f * 4 + lg + li) % QN) * 16);
#pragma unroll
                        for (int r = 0; r < SR; ++r) {
                            f32x4 v = acc[f][r] * us + bv;                                   // (no contraction: -ffp-contract=off)
                            if (act) {
                                const f32x4 n = v * s_neg;
                                v = f32x4{fmaxf(v[0], n[0]), fmaxf(v[1], n[1]), fmaxf(v[2], n[2]), fmaxf(v[3], n[3])};
                            }
                            *reinterpret_cast<__attribute__((address_space(3))) f32x4 *>((size_t)(sb + (unsigned)(r * 16 * TM * 4) + qoff)) = v;
                        }
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the parked rows are in the LDS
                signal(sync_s + 8 + wave, it + 1);
                if constexpr (SR < RPW)                      // the rows that are not parked (MF = 4): stored from here as before, after the signal
                    if (!(p.debug & 32)) conv_epilogue_fast_f<MF, RPW, true, SR>(p, acc, pix, m0, lg, p.w16_unscale, interior, (lds_cfloat_ptr)bias_s);
            } else if constexpr (FLAT) {
                // the tiled kernel's own epilogue (conv_f16x3_kernel: conv_unscale + conv_epilogue): scalar stores / residuals where the
                // output is not 16-byte addressable, out_scale -- the 2- and 3-channel heads
                if (!(p.debug & 32)) {
                    conv_unscale<MF, RPW>(p, acc);
                    long long pixv[RPW];
#pragma unroll
                    for (int r = 0; r < RPW; ++r) pixv[r] = pix(r, li);
                    if (!p.pixel_shuffle) conv_epilogue_plain<MF, RPW>(p, acc, pixv, m0, lg, (lds_cfloat_ptr)bias_s);
                    else conv_epilogue_flat<MF, RPW, false>(p, acc, pixv, pix, m0, lg, interior, (lds_cfloat_ptr)bias_s);
                }
            } else if (!(p.debug & 32)) {
                // ONE pass over the wave's rows, straight from the accumulators: the epilogue issues row r+1's residual loads
                // before row r's stores, so no wait inside it ever names a store (a second pass would start by waiting for
                // the first one's stores to be acknowledged: 6 of the 8 thousand cycles per tile this section used to take)
                conv_epilogue_fast_f<MF, RPW, true>(p, acc, pix, m0, lg, p.w16_unscale, interior,     // the dispatcher only sends p.fast_epi convs here
                                                    (lds_cfloat_ptr)bias_s);
            }
            if (STAMP) {
                const long long t = __builtin_amdgcn_s_memtime();
                t_zero -= t;
            }
#pragma unroll
            for (int a = 0; a < MF; ++a)
#pragma unroll
                for (int b = 0; b < RPW; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (STAMP) {
                const long long t = __builtin_amdgcn_s_memtime();
                t_zero += t;
            }
            kt = 0;
            ++it;
            if (STAMP) {
                const long long t = __builtin_amdgcn_s_memtime();
                t_epi += t - t_mark;
                t_mark = t;
            }
        }
    }
    if (STAMP && lane == 0 && p.gdn_x.p) {
        long long *o = reinterpret_cast<long long *>(p.gdn_x.p) + ((size_t)blockIdx.x * kP3Consumers + wave) * 8;
        o[0] = t_comp; o[1] = t_bar; o[2] = t_epi;
        o[3] = __builtin_amdgcn_s_memtime() - t_cyc0;
        o[4] = __builtin_amdgcn_s_memrealtime() - t_real0;
        o[5] = total; o[6] = n_it; o[7] = t_zero;
    }
}

template <int MF, bool INACT, int S = 1, bool SPLIT = false>
static int launch_p3(const ConvP &p, hipStream_t st) {
    using G = P3Geom<MF, S>;
    const int cus = device_cus();
    ConvP q = p;
    q.tiles_x = (p.Wout + 15) / 16;
    q.tiles_y = (p.Hout + G::TH - 1) / G::TH;
    q.m_tiles = (p.M_pad / 16 + MF - 1) / MF;
    const size_t lds = (size_t)G::LDS_BYTES_RING + (size_t)q.m_tiles * G::TM * sizeof(float) + 32 + 2 * kP3ProducerThreads * 8;      // + bias vector + hand-off slots + trash slots
    if (lds > 160 * 1024) return fail("conv2d(f16x3p): %zu bytes of LDS", lds);
    static LdsGrant grant;
    if (grant.ensure(reinterpret_cast<const void *>(conv3_f16x3p_kernel<MF, INACT, false, false, S, SPLIT>), lds)) return 1;
    const long long ntiles = (long long)q.tiles_x * q.tiles_y * q.m_tiles;
    if (ntiles <= 0 || ntiles > 0x7fffffffLL) return fail("conv2d(f16x3p): bad tile count %lld", ntiles);
    if (p.w16_plane * 2 > 0x7fffffffLL) return fail("conv2d(f16x3p): weight image too large for 32-bit lane offsets");
    long long blocks = cus;                       // one persistent 8-wave workgroup per CU
    if (const int forced = option_get(OPT_P3_BLOCKS); forced > 0) blocks = forced;      // experiments (tools/p3_scaling.py)
    if (blocks > ntiles) blocks = ntiles;
    if constexpr (S != 1) {
        if constexpr (MF == 4 && !INACT && !SPLIT) {
            if (p.debug & 256) {              // diagnostic: in-kernel stamps (tools/p3_stamps.py, P3_STRIDE=2)
                static LdsGrant grant_s;
                if (grant_s.ensure(reinterpret_cast<const void *>(conv3_f16x3p_kernel<4, false, true, false, S>), lds)) return 1;
                hipLaunchKernelGGL((conv3_f16x3p_kernel<4, false, true, false, S>), dim3((unsigned)blocks), dim3(kP3Threads), lds, st, q);
                return launch_status("conv2d(f16x3p, stride 2, stamps)");
            }
        }
        hipLaunchKernelGGL((conv3_f16x3p_kernel<MF, INACT, false, false, S, SPLIT>), dim3((unsigned)blocks), dim3(kP3Threads), lds, st, q);
        return launch_status("conv2d(f16x3p, stride 2)");
    } else if constexpr (SPLIT) {
        auto elems = [](const V &v) { return v.p ? (unsigned long long)v.H * v.W * v.ld : 0ull; };
        const bool small32 = elems(p.out) < (1ull << 32) && elems(p.res) < (1ull << 32) && elems(p.res2) < (1ull << 32);
        if (MF >= 3 && option_get(OPT_P3_STAGE) && small32 && (size_t)q.m_tiles * G::TM * sizeof(float) <= (size_t)G::BIAS_MAX) {
            static LdsGrant grant_g;
            if (grant_g.ensure(reinterpret_cast<const void *>(conv3_f16x3p_kernel<MF, false, false, true, 1, true>), G::TOTAL)) return 1;
            hipLaunchKernelGGL((conv3_f16x3p_kernel<MF, false, false, true, 1, true>), dim3((unsigned)blocks), dim3(kP3Threads), G::TOTAL, st, q);
            return launch_status("conv2d(f16x3p, split in, staged)");
        }
        if (MF == 4 && (p.debug & 256)) {             // diagnostic: in-kernel stamps (tools/p3_stamps.py)
            static LdsGrant grant_s;
            if (grant_s.ensure(reinterpret_cast<const void *>(conv3_f16x3p_kernel<4, false, true, false, 1, true>), lds)) return 1;
            hipLaunchKernelGGL((conv3_f16x3p_kernel<4, false, true, false, 1, true>), dim3((unsigned)blocks), dim3(kP3Threads), lds, st, q);
            return launch_status("conv2d(f16x3p, split in, stamps)");
        }
        hipLaunchKernelGGL((conv3_f16x3p_kernel<MF, false, false, false, 1, true>), dim3((unsigned)blocks), dim3(kP3Threads), lds, st, q);
        return launch_status("conv2d(f16x3p, split in)");
    } else {
    if (MF == 4 && !INACT && (p.debug & 256) && option_get(OPT_P3_STAGE) && (size_t)q.m_tiles * G::TM * sizeof(float) <= (size_t)G::BIAS_MAX) {
        static LdsGrant grant_ss;
        if (grant_ss.ensure(reinterpret_cast<const void *>(conv3_f16x3p_kernel<4, false, true, true>), G::TOTAL)) return 1;
        hipLaunchKernelGGL((conv3_f16x3p_kernel<4, false, true, true>), dim3((unsigned)blocks), dim3(kP3Threads), G::TOTAL, st, q);
        return launch_status("conv2d(f16x3p, staged, stamps)");
    }
    if ((MF == 4 || MF == 3) && !INACT && (p.debug & 256)) {             // diagnostic: in-kernel stamps (tools/p3_stamps.py)
        static LdsGrant grant_s;
        if (grant_s.ensure(reinterpret_cast<const void *>(conv3_f16x3p_kernel<MF, false, true>), lds)) return 1;
        hipLaunchKernelGGL((conv3_f16x3p_kernel<MF, false, true>), dim3((unsigned)blocks), dim3(kP3Threads), lds, st, q);
        return launch_status("conv2d(f16x3p, stamps)");
    }
    // staged epilogue (the kernel's STAGE note; option p3_stage, OFF by default: measured 5-12 % SLOWER than the direct epilogue,
    // profiles/r04_p3_stage_ab.txt -- the producer waves have no slack to spare for it): MF >= 3 and the bias of every M tile
    // must fit the fixed tail of the staged layout
    auto elems = [](const V &v) { return v.p ? (unsigned long long)v.H * v.W * v.ld : 0ull; };      // the drain addresses with 32-bit element offsets
    const bool small32 = elems(p.out) < (1ull << 32) && elems(p.res) < (1ull << 32) && elems(p.res2) < (1ull << 32);
    if (MF >= 3 && option_get(OPT_P3_STAGE) && small32 && (size_t)q.m_tiles * G::TM * sizeof(float) <= (size_t)G::BIAS_MAX) {
        static LdsGrant grant_g;
        if (grant_g.ensure(reinterpret_cast<const void *>(conv3_f16x3p_kernel<MF, INACT, false, true>), G::TOTAL)) return 1;
        hipLaunchKernelGGL((conv3_f16x3p_kernel<MF, INACT, false, true>), dim3((unsigned)blocks), dim3(kP3Threads), G::TOTAL, st, q);
        return launch_status("conv2d(f16x3p, staged)");
    }
    hipLaunchKernelGGL((conv3_f16x3p_kernel<MF, INACT>), dim3((unsigned)blocks), dim3(kP3Threads), lds, st, q);
    return launch_status("conv2d(f16x3p)");
    }
}


// Round 6: the instantiations with their own tile height (RPWT), the register prefetch (PF2), the tiled kernel's epilogue (FLAT) and two
// workgroups per CU (WG2); conv3_f16x3p_r.hip. No STAGE / SPLIT / STAMP forms.
template <int MF, bool INACT, int S, int RPWT, int PF, bool FLAT, bool WG2, bool STAMPK = false>
static int launch_p3r(const ConvP &p, hipStream_t st) {
    using G = P3Geom<MF, S, RPWT>;
    const int cus = device_cus();
    ConvP q = p;
    q.tiles_x = (p.Wout + 15) / 16;
    q.tiles_y = (p.Hout + G::TH - 1) / G::TH;
    q.m_tiles = (p.M_pad / 16 + MF - 1) / MF;
    constexpr bool RING3R = PF == 3 && WG2;             // (the kernel's RING3R: three patch buffers, one wave's worth of trash slots)
    constexpr int NPB = RING3R ? 3 : PF ? 2 : G::NPB;
    const size_t lds = (size_t)(NPB * 2 * G::PATCH_HALFS + 2 * 2 * G::W_HALFS) * 2 + (size_t)q.m_tiles * G::TM * sizeof(float) + 32 +
                       2 * (RING3R ? 64 : kP3ProducerThreads) * 8;      // + bias vector + hand-off slots + trash slots
    if (lds > (WG2 ? 80 : 160) * 1024) return fail("conv2d(f16x3p r): %zu bytes of LDS", lds);
    auto kern = conv3_f16x3p_kernel<MF, INACT, STAMPK, false, S, false, RPWT, PF, FLAT, WG2>;
    static LdsGrant grant;
    if (grant.ensure(reinterpret_cast<const void *>(kern), lds)) return 1;
    const long long ntiles = (long long)q.tiles_x * q.tiles_y * q.m_tiles;
    if (ntiles <= 0 || ntiles > 0x7fffffffLL) return fail("conv2d(f16x3p r): bad tile count %lld", ntiles);
    if (p.w16_plane * 2 > 0x7fffffffLL) return fail("conv2d(f16x3p r): weight image too large for 32-bit lane offsets");
    long long blocks = WG2 ? 2 * cus : cus;
    if (const int forced = option_get(OPT_P3_BLOCKS); forced > 0) blocks = forced;
    if (blocks > ntiles) blocks = ntiles;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(kP3Threads), lds, st, q);
    return launch_status("conv2d(f16x3p r)");
}

// conv3_f16x3p_r.hip
int launch_p3_small(const ConvP &p, int mf, int rpw, bool inact, int pf, hipStream_t st);      // stride 1, fused fast epilogue, 4*rpw x 16 tiles
int launch_p3_narrow(const ConvP &p, bool inact, bool flat, int pf, hipStream_t st);                     // stride 1, <= 16 output channels, 16x16 tiles, 2 workgroups per CU
int launch_p3s2_pf(const ConvP &p, int mf, bool inact, int pf, hipStream_t st);                  // stride 2 with the register prefetch (pf 1) / pair loads (pf 2)
int launch_p3_big_pair(const ConvP &p, int mf, bool inact, hipStream_t st);
int launch_p3_big_roles(const ConvP &p, int mf, int rpw, bool inact, hipStream_t st);      // conv3_f16x3p_r3.hip
int launch_p3_big_late(const ConvP &p, int mf, int rpw, bool inact, hipStream_t st);       // conv3_f16x3p_r3.hip
int launch_p3_late_stamps(const ConvP &p, hipStream_t st);                                   // conv3_f16x3p_r3.hip: MF = 4, no input activation, in-kernel stamps
int launch_p3_tall(const ConvP &p, int mf, bool inact, int pf, hipStream_t st);                  // stride 1, 32x16 tiles (experiment)                      // stride 1, 24x16 tiles, pair loads (experiment)                         // stride 2 with the register prefetch

}  // namespace lssvc
