// dwpre_f16x3.hip -- DepthConv's front half as ONE kernel: t = LeakyReLU(conv1x1(x)); out = depthwise3x3(t) + bias.
//
// Reference: src/models/lssvc_modules.py:15-44 (DepthConv.conv1 -> depth_conv). Unfused this is four tensor passes
// (read x, write t, read t, write out) by two bandwidth-bound kernels; here the intermediate t never reaches HBM:
//   phase 1  each of the 8 waves runs the 1x1 conv (f16x3 arithmetic, identical K order to conv_pw_allm_f16x3_kernel,
//            so t is bit-identical to the unfused path) on 16-pixel groups of the (TH+2) x 18 halo patch of the
//            workgroup's TH x 16 output tile, applies bias + LeakyReLU, forces pixels outside the image to 0 (the
//            depthwise conv zero-pads t) and stores fp32 t to LDS (pixel pitch C + 4 floats: conflict-free b128 writes);
//   phase 2  thread (pixel row, 4-channel quad) accumulates the 9 taps from LDS in the order of dwconv3x3_kernel
//            (ky, kx ascending, fmaf from 0, bias last: bit-identical again) and stores float4s, 16 quads = one
//            256-byte pixel row per 16 lanes.
// The halo costs (TH+2)*18 / (TH*16) extra conv1 work and x reads (1.27x at TH = 16, served by L2); the weights
// (<= 64x64 hi/lo = 16 KB) and dw taps stay in LDS / registers for the life of the persistent workgroup, and the next
// tile's x fragments are prefetched into the SAME registers: with DEEP (option dwpre_deep, default) a wave reloads each of
// its 16-pixel groups for the next tile as soon as phase 1 has consumed that group, so that global loads are in flight
// during both phases (the kernel is HBM-bound, and with the prefetch issued only at the start of phase 2 the memory pipeline
// idled through every phase 1: 3.5 TB/s against the 5.3 TB/s a plain device copy of the same tensors reaches,
// tools/bw_copy_probe.py; a second register set for a whole-tile-ahead prefetch spilled 28 registers at C = 64); without
// it all groups are reloaded at the start of phase 2 (round-2 behaviour).
#include <cstdlib>

#include "conv_f16x3_kernel.h"

namespace lssvc {

struct DwPreP {
    V in[LSSVC_CONV_MAX_INPUTS];
    int n_in, n_chunks16, C, M_pad;
    const _Float16 *w16;
    long long w16_plane;
    float w16_unscale;
    const float *bias, *dw_w, *dw_b;   // conv1 bias [M_pad], dw taps [9][C], dw bias [C]
    float in_slope, slope;             // input LeakyReLU of conv1 (1 = none), LeakyReLU after conv1
    V out;
    int tiles_x, tiles_y;
};

constexpr int kDwThreads = 512;

// TH output rows per tile; CF = C / 16 accumulator fragments; NS = K-steps of conv1 (32 input channels each)
template <int CF, int NS, int TH, bool DEEP>
__global__ __launch_bounds__(kDwThreads, 1) void dwpre_f16x3_kernel(const DwPreP p) {
    constexpr int PW = 18, PH = TH + 2, NPIX = PH * PW, NGRP = (NPIX + 15) / 16, WAVES = kDwThreads / 64;
    constexpr int GPW = (NGRP + WAVES - 1) / WAVES;                 // halo groups per wave
    constexpr int C = 16 * CF, PITCH = C + 4;                       // fp32 pixel pitch of t in LDS
    constexpr int NSLOT = 2 * NS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *t_lds = reinterpret_cast<float *>(smem);                                     // [NPIX][PITCH]
    _Float16 *wlds = reinterpret_cast<_Float16 *>(t_lds + NPIX * PITCH);                // [hi|lo][slot][C][16]
    constexpr int WPLANE = NSLOT * C * CK16;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 15;
    const int lg = lane >> 4;
    const int tsel = lg >> 1;
    const int ch8 = (lg & 1) * 8;
    {
        const _Float16 *g_h = p.w16, *g_l = p.w16 + p.w16_plane;
        for (int idx = tid; idx < NSLOT * C * 2; idx += kDwThreads) {
            const int c = idx / (C * 2);
            const int r = idx - c * (C * 2);
            const int m = r >> 1, half = r & 1;
            f16x8 h = {0, 0, 0, 0, 0, 0, 0, 0}, l = {0, 0, 0, 0, 0, 0, 0, 0};
            if (c < p.n_chunks16 && m < p.M_pad) {
                const size_t o = ((size_t)c * p.M_pad + m) * CK16 + half * 8;
                h = *reinterpret_cast<const f16x8 *>(g_h + o);
                l = *reinterpret_cast<const f16x8 *>(g_l + o);
            }
            const int d = (c * C + m) * CK16 + half * 8;
            *reinterpret_cast<f16x8 *>(wlds + d) = h;
            *reinterpret_cast<f16x8 *>(wlds + WPLANE + d) = l;
        }
    }
    // phase-2 role of this thread: channel quad q4 (fixed), pixel rows tid / (C/4) + k * (threads / (C/4))
    constexpr int QUADS = C / 4, PIX_PER_PASS = kDwThreads / QUADS, NPASS = (TH * 16 + PIX_PER_PASS - 1) / PIX_PER_PASS;
    const int q4 = (tid % QUADS) * 4;
    float4 taps[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) taps[k] = *reinterpret_cast<const float4 *>(p.dw_w + k * p.C + q4);
    const float4 dwb = *reinterpret_cast<const float4 *>(p.dw_b + q4);
    float4 b1[CF];
#pragma unroll
    for (int f = 0; f < CF; ++f) b1[f] = *reinterpret_cast<const float4 *>(p.bias + f * 16 + 4 * lg);
    __syncthreads();

    const int H = p.out.H, Wd = p.out.W;
    const int ntiles = p.tiles_x * p.tiles_y;
    const int n0 = (p.in[0].C + 15) >> 4, n1 = p.n_in > 1 ? (p.in[1].C + 15) >> 4 : 0;

    // this lane's source chunk per K-step is tile-independent
    const float *sbase[NS];
    int sld[NS], scc[NS], sleft[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        int c = 2 * s + tsel;
        const bool in_range = c < p.n_chunks16;
        c = in_range ? c : p.n_chunks16 - 1;
        const bool s1 = c >= n0, s2 = c >= n0 + n1;
        const V X = s2 ? p.in[2] : (s1 ? p.in[1] : p.in[0]);
        const int c0 = (c - (s2 ? n0 + n1 : (s1 ? n0 : 0))) * 16;
        const int avail = X.C - c0 - ch8;
        sbase[s] = X.p;
        sld[s] = X.ld;
        sleft[s] = in_range ? avail : 0;
        scc[s] = avail > 0 ? c0 + ch8 : 0;
    }

    float4 raw[GPW][NS][2];
    auto load_group = [&](int tile, int g) {
        const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
        {
            const int hp = (wave + WAVES * g) * 16 + li;                 // halo pixel index of this lane's column
            const int py = hp / PW, px = hp - py * PW;
            const int gy = ty * TH - 1 + py, gx = tx * 16 - 1 + px;
            const bool ok = hp < NPIX && gy >= 0 && gy < H && gx >= 0 && gx < Wd;
            const size_t pixoff = ok ? (size_t)gy * Wd + gx : 0;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const float *src = sbase[s] + pixoff * sld[s] + scc[s];
                raw[g][s][0] = *reinterpret_cast<const float4 *>(src);
                raw[g][s][1] = *reinterpret_cast<const float4 *>(src + (sleft[s] > 4 ? 4 : 0));
            }
        }
    };
    auto load_tile = [&](int tile) {
#pragma unroll
        for (int g = 0; g < GPW; ++g) load_group(tile, g);
    };

    // one tile: phase 1 from `raw` (loaded earlier), barrier, [shallow prefetch], phase 2, barrier
    auto do_tile = [&](int tile, int prefetch_tile) {
        const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
        // ---------------------------------------------------------------- phase 1: t = lrelu(W1 x + b1) on the halo patch
#pragma unroll
        for (int g = 0; g < GPW; ++g) {
            const int grp = wave + WAVES * g;
            const int hp = grp * 16 + li;
            const int py = hp / PW, px = hp - py * PW;
            const int gy = ty * TH - 1 + py, gx = tx * 16 - 1 + px;
            const bool inside = hp < NPIX && gy >= 0 && gy < H && gx >= 0 && gx < Wd;
            f32x4 acc[CF][1];
#pragma unroll
            for (int f = 0; f < CF; ++f) acc[f][0] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const float v[8] = {raw[g][s][0].x, raw[g][s][0].y, raw[g][s][0].z, raw[g][s][0].w,
                                    raw[g][s][1].x, raw[g][s][1].y, raw[g][s][1].z, raw[g][s][1].w};
                f16x8 bh[1], bl[1];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float x = (inside && j < sleft[s]) ? v[j] : 0.f;
                    x = fmaxf(x, p.in_slope * x);
                    x = fminf(fmaxf(x, -65504.f), 65504.f);
                    const _Float16 h = (_Float16)x;
                    bh[0][j] = h;
                    bl[0][j] = (_Float16)(x - (float)h);
                }
                f16x8 ah[CF], al[CF];
#pragma unroll
                for (int f = 0; f < CF; ++f) {
                    const int o = ((2 * s + tsel) * C + f * 16 + li) * CK16 + ch8;
                    ah[f] = *reinterpret_cast<const f16x8 *>(wlds + o);
                    al[f] = *reinterpret_cast<const f16x8 *>(wlds + WPLANE + o);
                }
#pragma unroll
                for (int f = 0; f < CF; ++f) acc[f][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[f], bh[0], acc[f][0], 0, 0, 0);
#pragma unroll
                for (int f = 0; f < CF; ++f) acc[f][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[f], bl[0], acc[f][0], 0, 0, 0);
#pragma unroll
                for (int f = 0; f < CF; ++f) acc[f][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[f], bh[0], acc[f][0], 0, 0, 0);
            }
            if (hp < NPIX) {
#pragma unroll
                for (int f = 0; f < CF; ++f) {
                    float o[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float bj = j == 0 ? b1[f].x : (j == 1 ? b1[f].y : (j == 2 ? b1[f].z : b1[f].w));
                        float y = acc[f][0][j] * p.w16_unscale + bj;
                        y = y > 0.f ? y : y * p.slope;
                        o[j] = inside ? y : 0.f;                           // zero padding of the depthwise conv
                    }
                    *reinterpret_cast<float4 *>(t_lds + hp * PITCH + f * 16 + 4 * lg) = make_float4(o[0], o[1], o[2], o[3]);
                }
            }
            if (DEEP && prefetch_tile >= 0) load_group(prefetch_tile, g);      // raw[g] is consumed: refill it for the next tile now
        }
        __syncthreads();
        if (!DEEP && prefetch_tile >= 0) load_tile(prefetch_tile);             // shallow mode: in flight during phase 2 only
        // ---------------------------------------------------------------- phase 2: depthwise 3x3 out of LDS
#pragma unroll
        for (int k = 0; k < NPASS; ++k) {
            const int op = tid / QUADS + k * PIX_PER_PASS;                     // output pixel inside the tile
            if (op < TH * 16) {
                const int oy = op >> 4, ox = op & 15;
                const int gy = ty * TH + oy, gx = tx * 16 + ox;
                float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const float4 v = *reinterpret_cast<const float4 *>(t_lds + ((oy + ky) * PW + ox + kx) * PITCH + q4);
                        const float4 w = taps[ky * 3 + kx];
                        a.x = fmaf(v.x, w.x, a.x);
                        a.y = fmaf(v.y, w.y, a.y);
                        a.z = fmaf(v.z, w.z, a.z);
                        a.w = fmaf(v.w, w.w, a.w);
                    }
                a.x += dwb.x; a.y += dwb.y; a.z += dwb.z; a.w += dwb.w;
                if (gy < H && gx < Wd) *reinterpret_cast<float4 *>(p.out.p + ((size_t)gy * Wd + gx) * p.out.ld + q4) = a;
            }
        }
        __syncthreads();                                                       // t_lds is rewritten by the next tile
    };

    const int stride = (int)gridDim.x;
    int tile = blockIdx.x;
    if (tile < ntiles) load_tile(tile);
    for (; tile < ntiles; tile += stride) do_tile(tile, tile + stride < ntiles ? tile + stride : -1);
}

template <int CF, int NS, int TH, bool DEEP>
static int launch_dwpre_impl(const DwPreP &p, hipStream_t st) {
    constexpr int C = 16 * CF;
    constexpr size_t lds = (size_t)(TH + 2) * 18 * (C + 4) * 4 + (size_t)2 * (2 * NS) * C * CK16 * 2;
    static_assert(lds <= 160 * 1024, "dwpre tile does not fit LDS");
    const int cus = device_cus();
    static LdsGrant grant;
    if (grant.ensure(reinterpret_cast<const void *>(dwpre_f16x3_kernel<CF, NS, TH, DEEP>), lds)) return 1;
    DwPreP q = p;
    q.tiles_x = (p.out.W + 15) / 16;
    q.tiles_y = (p.out.H + TH - 1) / TH;
    long long blocks = (long long)q.tiles_x * q.tiles_y;
    if (blocks > cus) blocks = cus;
    hipLaunchKernelGGL((dwpre_f16x3_kernel<CF, NS, TH, DEEP>), dim3((unsigned)blocks), dim3(kDwThreads), lds, st, q);
    return launch_status("conv1x1_dw3x3_f16x3");
}

template <int CF, int NS, int TH>
static int launch_dwpre(const DwPreP &p, hipStream_t st) {
    return option_get(OPT_DWPRE_DEEP) ? launch_dwpre_impl<CF, NS, TH, true>(p, st) : launch_dwpre_impl<CF, NS, TH, false>(p, st);
}

}  // namespace lssvc

using namespace lssvc;

extern "C" int lssvc_conv1x1_dw3x3_f16x3(const lssvc_conv_desc *d, const float *dw_weight, const float *dw_bias, void *stream) {
    LSSVC_CHECK(d != nullptr && dw_weight != nullptr && dw_bias != nullptr, "conv1x1_dw3x3: null argument");
    LSSVC_CHECK(d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad_t == 0 && d->pad_l == 0, "conv1x1_dw3x3: the leading conv must be 1x1 stride 1");
    LSSVC_CHECK(d->precision == LSSVC_PREC_F16X3 && d->weight16 != nullptr && d->bias != nullptr, "conv1x1_dw3x3: needs f16x3 weights and a bias");
    LSSVC_CHECK(d->epilogue == LSSVC_EPI_NONE && !d->pixel_shuffle && d->residual.ptr == nullptr && d->out_scale == 1.0f,
                "conv1x1_dw3x3: the leading conv takes no GDN / shuffle / residual / scale");
    LSSVC_CHECK(d->in_act != LSSVC_INACT_SQUARE && (d->in_act != LSSVC_INACT_LRELU || (d->in_slope >= 0.f && d->in_slope <= 1.f)),
                "conv1x1_dw3x3: unsupported input activation");
    LSSVC_CHECK(d->act == LSSVC_ACT_NONE || d->act == LSSVC_ACT_LRELU, "conv1x1_dw3x3: activation must be none or LeakyReLU");
    LSSVC_CHECK(d->n_in >= 1 && d->n_in <= LSSVC_CONV_MAX_INPUTS, "conv1x1_dw3x3: n_in=%d", d->n_in);
    LSSVC_CHECK(view_ok(&d->out) && vec4_ok(&d->out), "conv1x1_dw3x3: bad out view");
    const int C = d->Cout;
    LSSVC_CHECK(C == d->out.C && (C == 32 || C == 48 || C == 64) && d->M_pad == C, "conv1x1_dw3x3: C = %d not in {32, 48, 64}", C);
    DwPreP p{};
    long long chunks16 = 0;
    for (int i = 0; i < d->n_in; ++i) {
        LSSVC_CHECK(view_ok(&d->in[i]) && vec4_ok(&d->in[i]) && same_hw(&d->in[i], &d->out), "conv1x1_dw3x3: bad input view %d", i);
        p.in[i] = mk(&d->in[i]);
        chunks16 += (d->in[i].C + 15) / 16;
    }
    LSSVC_CHECK(chunks16 <= 4, "conv1x1_dw3x3: at most 64 input channels (got %lld chunks of 16)", chunks16);
    p.n_in = d->n_in;
    p.n_chunks16 = (int)chunks16;
    p.C = C;
    p.M_pad = d->M_pad;
    p.w16 = reinterpret_cast<const _Float16 *>(d->weight16);
    p.w16_plane = chunks16 * (long long)d->M_pad * 16;
    p.w16_unscale = d->weight16_unscale != 0.f ? d->weight16_unscale : 1.f;
    p.bias = d->bias;
    p.dw_w = dw_weight;
    p.dw_b = dw_bias;
    p.in_slope = d->in_act == LSSVC_INACT_LRELU ? d->in_slope : 1.0f;
    p.slope = d->act == LSSVC_ACT_LRELU ? d->slope : 1.0f;
    p.out = mk(&d->out);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int ns = (int)(chunks16 + 1) / 2;
#define LSSVC_DW_CASE(cf, nsv) \
    if (C == 16 * cf && ns == nsv) return launch_dwpre<cf, nsv, 16>(p, st);
    LSSVC_DW_CASE(2, 1) LSSVC_DW_CASE(2, 2) LSSVC_DW_CASE(3, 1) LSSVC_DW_CASE(3, 2) LSSVC_DW_CASE(4, 1) LSSVC_DW_CASE(4, 2)
#undef LSSVC_DW_CASE
    return fail("conv1x1_dw3x3: no kernel for C=%d, %d K-steps", C, ns);
}
