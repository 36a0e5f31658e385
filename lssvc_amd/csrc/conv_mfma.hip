// conv_mfma.hip -- NHWC implicit-GEMM convolution on fp32 MFMA (v_mfma_f32_16x16x4_f32) for gfx950.
//
// GEMM view:  D[m][n] = sum_k W[m][k] * X[k][n]
//   m = output channel (A operand = weights), n = output pixel (B operand = input window),
//   k = (ky, kx, input channel).  With channels on the MFMA row axis each lane ends up holding
//   4 consecutive channels of one pixel, so the fused epilogue and the NHWC store are float4 wide.
//
// Workgroup = 256 threads = 4 waves (one per SIMD).  Output tile = (4*RPW rows) x 16 columns of
// pixels x (16*MF) channels; wave w owns rows [w*RPW, (w+1)*RPW), all MF channel fragments.
// K is walked in chunks of CK=8 input channels:
//   - the input halo patch ((TH-1)*s+KH) x (15*s+KW) pixels x 8 channels is staged in LDS once per
//     chunk (zero padding, virtual concat of up to 3 inputs and the input activation are applied
//     while staging),
//   - the filter slab of one kernel row (KW taps x 16*MF channels x 8) is staged per ky.
// LDS rows are 12 floats (8 + 4 pad): a ds_read_b64 at row stride 48 B is bank-conflict free for
// the 16x16x4 operand pattern (lane = (i, g): row i, floats 2g..2g+1), see MI355X_MICROARCH LDS table.
// One ds_read_b64 pair feeds two MFMAs (k = channels {0,2,4,6} then {1,3,5,7} across the 4 lane groups).
//
// fp32 MFMA is exact fp32 (fmaf chain), so this is the parity path against the fp32 CPU oracle.
#include <cstdlib>

#include "conv_pw_f16x3_kernel.h"

namespace lssvc {

// ---- runtime tuning switches: environment at first use, lssvc_set_option() afterwards ----------------
static std::atomic<int> g_opt[OPT_COUNT];
static std::atomic<bool> g_opt_set[OPT_COUNT];
static const char *const kOptEnv[OPT_COUNT] = {"LSSVC_F16X3_PERSIST", "LSSVC_F16X3_PERSIST_MIN_TILES", "LSSVC_F16X3_PERSIST7", "LSSVC_POINTWISE_BLOCKS", "LSSVC_DWPRE_DEEP", "LSSVC_P3_BLOCKS", "LSSVC_P3_STAGE", "LSSVC_F16X3_PERSIST_S2",
                                              "LSSVC_P3_SMALL", "LSSVC_P3_NARROW", "LSSVC_P3_PF2", "LSSVC_P3_FORCE", "LSSVC_GDN_FAST_OPT", "LSSVC_P3_BIG_PAIR", "LSSVC_P7_NARROW", "LSSVC_RESAMPLE_ROWS"};
static const char *const kOptName[OPT_COUNT] = {"f16x3_persist", "f16x3_persist_min_tiles", "f16x3_persist7", "pointwise_blocks", "dwpre_deep", "p3_blocks", "p3_stage", "f16x3_persist_s2",
                                               "p3_small", "p3_narrow", "p3_pf2", "p3_force", "gdn_fast", "p3_big_pair", "p7_narrow", "resample_rows"};
static const int kOptDefault[OPT_COUNT] = {1, 256, 1, 1, 1, 0, 0, 1, 1, 1, 1, 0, 1, 0, 0, 1};      // p3_stage: off (measured slower); kept for the record and its test
int option_get(int which) {
    if (!g_opt_set[which].load(std::memory_order_acquire)) {
        const char *e = getenv(kOptEnv[which]);
        g_opt[which].store(e ? atoi(e) : kOptDefault[which], std::memory_order_relaxed);
        g_opt_set[which].store(true, std::memory_order_release);
    }
    return g_opt[which].load(std::memory_order_relaxed);
}

static char *last_kernel_name() {
    static thread_local char name[96] = {0};
    return name;
}

static long long grid_blocks(const ConvP &p, int MF, int RPW) {
    const int TH = 4 * RPW;
    return (long long)((p.Wout + 15) / 16) * ((p.Hout + TH - 1) / TH) * ((p.M_pad / 16 + MF - 1) / MF);
}

// Tile choice: MF = 16-channel fragments per workgroup (prefer 4, or 3 when the channel count divides
// by 48 but not 64); RPW = pixel rows per wave: 4 (16x16 pixel tile) for stride 1, 2 for stride 2
// (the halo patch doubles), halved further while the grid would not fill the 256 CUs twice over.
static void pick_variant(const ConvP &p, int &MF, int &RPW) {
    const int frags = p.M_pad / 16;
    MF = frags >= 4 ? 4 : frags;
    if (frags % 4 != 0 && frags % 3 == 0) MF = 3;
    RPW = (p.stride == 2) ? 2 : 4;
    while (RPW > 1 && (p.Hout <= 2 * RPW || grid_blocks(p, MF, RPW) < 512)) RPW >>= 1;
    // still fewer workgroups than CUs (the prior networks' 36x60 / 72x120 maps): these launches are bound by the latency of
    // staging a K chunk, not by the matrix pipe, so narrower M tiles = more workgroups in flight beat the patch re-staging
    static const int narrow = getenv("LSSVC_TILED_NARROW") ? atoi(getenv("LSSVC_TILED_NARROW")) : 1;
    if (narrow && RPW == 1)
        while (MF > 1 && MF % 2 == 0 && grid_blocks(p, MF, RPW) < 256) MF >>= 1;
}

}  // namespace lssvc

using namespace lssvc;

extern "C" const char *lssvc_conv2d_last_kernel(void) { return last_kernel_name(); }

extern "C" int lssvc_set_option(const char *name, int32_t value) {
    LSSVC_CHECK(name != nullptr, "set_option: null name");
    for (int i = 0; i < OPT_COUNT; ++i)
        if (!strcmp(name, kOptName[i])) {
            g_opt[i].store(value, std::memory_order_relaxed);
            g_opt_set[i].store(true, std::memory_order_release);
            return 0;
        }
    return fail("set_option: unknown option '%s'", name);
}
extern "C" int lssvc_get_option(const char *name, int32_t *value) {
    LSSVC_CHECK(name != nullptr && value != nullptr, "get_option: null argument");
    for (int i = 0; i < OPT_COUNT; ++i)
        if (!strcmp(name, kOptName[i])) {
            *value = option_get(i);
            return 0;
        }
    return fail("get_option: unknown option '%s'", name);
}

extern "C" int lssvc_conv2d_variant(int32_t Hout, int32_t Wout, int32_t M_pad, int32_t stride) {
    ConvP p;
    memset(&p, 0, sizeof(p));
    p.Hout = Hout; p.Wout = Wout; p.M_pad = M_pad; p.stride = stride;
    int MF, RPW;
    pick_variant(p, MF, RPW);
    return MF * 16 + RPW;
}

extern "C" int lssvc_conv2d(const lssvc_conv_desc *d, void *stream) {
    LSSVC_CHECK(d != nullptr, "conv2d: null descriptor");
    LSSVC_CHECK(d->n_in >= 1 && d->n_in <= LSSVC_CONV_MAX_INPUTS, "conv2d: n_in=%d", d->n_in);
    LSSVC_CHECK(d->weight != nullptr, "conv2d: null weight");
    LSSVC_CHECK(d->KH == d->KW && (d->KH == 1 || d->KH == 2 || d->KH == 3 || d->KH == 7), "conv2d: kernel %dx%d", d->KH, d->KW);
    LSSVC_CHECK(d->stride == 1 || d->stride == 2, "conv2d: stride %d", d->stride);
    LSSVC_CHECK(d->Cout >= 1 && d->M_pad == (d->Cout + 15) / 16 * 16, "conv2d: Cout=%d M_pad=%d", d->Cout, d->M_pad);
    LSSVC_CHECK(view_ok(&d->out), "conv2d: bad out view");
    ConvP p;
    memset(&p, 0, sizeof(p));
    for (int i = 0; i < d->n_in; ++i) {
        LSSVC_CHECK(view_ok(&d->in[i]), "conv2d: bad input view %d", i);
        LSSVC_CHECK(same_hw(&d->in[i], &d->in[0]), "conv2d: input %d is %dx%d, input 0 is %dx%d", i, d->in[i].H,
                    d->in[i].W, d->in[0].H, d->in[0].W);
        p.in[i] = mk(&d->in[i]);
        p.in_vec[i] = vec4_ok(&d->in[i]);
    }
    p.n_in = d->n_in;
    for (int i = 0; i < d->n_in; ++i) p.n_chunks += (d->in[i].C + LSSVC_CONV_CK - 1) / LSSVC_CONV_CK;
    p.w = d->weight;
    p.bias = d->bias;
    p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad_t = d->pad_t; p.pad_l = d->pad_l;
    p.Cout = d->Cout; p.M_pad = d->M_pad;
    p.in_act = d->in_act; p.in_slope = d->in_slope;
    p.epilogue = d->epilogue;
    p.act = d->act; p.slope = d->slope;
    p.out_scale = d->out_scale;
    p.pixel_shuffle = d->pixel_shuffle;
    p.out = mk(&d->out);
    p.out_vec = vec4_ok(&d->out);
    if (d->pixel_shuffle) {
        LSSVC_CHECK(d->Cout % 4 == 0 && d->out.C == d->Cout / 4 && d->out.H % 2 == 0 && d->out.W % 2 == 0,
                    "conv2d: pixel-shuffle out view %dx%dx%d does not match Cout=%d", d->out.H, d->out.W, d->out.C, d->Cout);
        LSSVC_CHECK(d->residual.ptr == nullptr && d->epilogue == LSSVC_EPI_NONE,
                    "conv2d: residual / GDN epilogue cannot be combined with pixel shuffle");
        p.Hout = d->out.H / 2; p.Wout = d->out.W / 2;
    } else {
        LSSVC_CHECK(d->out.C == d->Cout, "conv2d: out.C=%d != Cout=%d", d->out.C, d->Cout);
        p.Hout = d->out.H; p.Wout = d->out.W;
    }
    // the last output pixel must touch at least one real input pixel (guards swapped H/W, wrong stride)
    LSSVC_CHECK((p.Hout - 1) * d->stride - d->pad_t < d->in[0].H && (p.Wout - 1) * d->stride - d->pad_l < d->in[0].W,
                "conv2d: output %dx%d too large for input %dx%d (k=%dx%d s=%d)", p.Hout, p.Wout, d->in[0].H,
                d->in[0].W, d->KH, d->KW, d->stride);
    if (d->residual.ptr) {
        LSSVC_CHECK(view_ok(&d->residual) && same_shape(&d->residual, &d->out), "conv2d: residual shape mismatch");
        p.res = mk(&d->residual);
        p.res_vec = vec4_ok(&d->residual);
    } else {
        p.res = mk_null();
    }
    p.res2 = mk_null();
    if (d->residual2.ptr) {
        LSSVC_CHECK(d->residual.ptr != nullptr, "conv2d: residual2 without residual");
        LSSVC_CHECK(view_ok(&d->residual2) && same_shape(&d->residual2, &d->out) && vec4_ok(&d->residual2),
                    "conv2d: residual2 must match `out` and be 16-byte addressable");
        p.res2 = mk(&d->residual2);
    }
    if (d->epilogue != LSSVC_EPI_NONE) {
        LSSVC_CHECK(view_ok(&d->gdn_x) && same_shape(&d->gdn_x, &d->out), "conv2d: gdn_x shape mismatch");
        p.gdn_x = mk(&d->gdn_x);
        p.gdn_vec = vec4_ok(&d->gdn_x);
    } else {
        p.gdn_x = mk_null();
    }
    {
        static const int fast_on = getenv("LSSVC_FAST_EPI") ? atoi(getenv("LSSVC_FAST_EPI")) : 1;
        const bool common = fast_on && d->epilogue == LSSVC_EPI_NONE && (d->Cout % 4 == 0) && p.out_vec && d->out_scale == 1.0f &&
                            (d->act != LSSVC_ACT_LRELU || (d->slope >= 0.0f && d->slope <= 1.0f));
        p.fast_epi = 0;
        if (common && !d->pixel_shuffle && (!d->residual.ptr || p.res_vec)) p.fast_epi = 1;
        if (common && d->pixel_shuffle && !d->residual.ptr && (d->Cout % 16 == 0)) p.fast_epi = 2;   // cps % 4 == 0
    }
    {
        static const int gdn_fast_on = getenv("LSSVC_GDN_FAST") ? atoi(getenv("LSSVC_GDN_FAST")) : 1;
        p.gdn_fast = (gdn_fast_on && option_get(OPT_GDN_FAST) && d->epilogue != LSSVC_EPI_NONE && (d->Cout % 4 == 0) && p.out_vec && p.gdn_vec && !d->pixel_shuffle &&
                      d->out_scale == 1.0f && (!d->residual.ptr || p.res_vec)) ? 1 : 0;
    }
    LSSVC_CHECK(!d->residual2.ptr || p.fast_epi == 1, "conv2d: residual2 needs the plain fused epilogue (no GDN / shuffle / scale, Cout %% 4 == 0)");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    {
        static const int dbg = getenv("LSSVC_CONV_DEBUG") ? atoi(getenv("LSSVC_CONV_DEBUG")) : 0;
        p.debug = dbg;
        if ((dbg & 256) && d->epilogue == LSSVC_EPI_NONE && d->gdn_x.ptr) p.gdn_x = V{d->gdn_x.ptr, 0, 0, 0, 0};   // stamp buffer (diagnostic)
    }

    int MF, RPW;
    pick_variant(p, MF, RPW);

    const int ks = d->KH, sd = d->stride;
    char *kname = last_kernel_name();
    bool vec = true;
    for (int i = 0; i < p.n_in; ++i) vec = vec && p.in_vec[i];
    LSSVC_CHECK((d->precision & ~(LSSVC_PREC_MASK | LSSVC_PREC_SPLIT_IN)) == 0, "conv2d: precision 0x%x", d->precision);
    if ((d->precision & LSSVC_PREC_MASK) == LSSVC_PREC_F16X3) {
        // fp16-MFMA 3-term split: built for the MFMA-bound layers (3x3 / 7x7, stride 1, 16-byte addressable inputs);
        // everything else (1x1, strided, 2..3-channel inputs, GDN) stays on the exact-fp32 kernel
        LSSVC_CHECK(d->weight16 != nullptr, "conv2d: precision f16x3 needs weight16");
        long long chunks16 = 0;
        for (int i = 0; i < d->n_in; ++i) chunks16 += (d->in[i].C + 15) / 16;
        p.n_chunks16 = (int)chunks16;
        p.w16 = d->weight16;
        p.w16_unscale = d->weight16_unscale != 0.0f ? d->weight16_unscale : 1.0f;
        p.w16_plane = chunks16 * ks * ks * (long long)p.M_pad * 16;
        if (d->precision & LSSVC_PREC_SPLIT_IN) {
            // pre-split inputs: only the persistent 3x3 kernels read them (there is no fallback that would silently convert)
            LSSVC_CHECK(ks == 3 && d->in_act == LSSVC_INACT_NONE, "conv2d: pre-split inputs need a 3x3 conv without input activation (k=%d in_act=%d)", ks, d->in_act);
            for (int i = 0; i < d->n_in; ++i)
                LSSVC_CHECK(d->in[i].ld % 16 == 0 && d->in[i].ld >= (d->in[i].C + 15) / 16 * 16 && (reinterpret_cast<uintptr_t>(d->in[i].ptr) & 63) == 0,
                            "conv2d: input %d is not a pre-split view (C=%d ld=%d)", i, d->in[i].C, d->in[i].ld);
            p.in_split = 1;
            LSSVC_CHECK(option_get(sd == 1 ? OPT_P3_ON : OPT_P3_S2) && p.fast_epi, "conv2d: pre-split inputs need the persistent 3x3 kernel (fused fast epilogue)");
            if (sd == 1) return dispatch_conv3_f16x3p(p, st, kname);
            LSSVC_CHECK(p3_pick_mf_public(p.M_pad / 16) >= 3, "conv2d: stride-2 conv with pre-split inputs needs >= 48 output channels");
            return dispatch_conv3s2_f16x3p(p, st, kname);
        }
        if (vec && sd == 1 && ks == 3 && d->in_act != LSSVC_INACT_SQUARE && conv3_f16x3p_wanted(p))
            return dispatch_conv3_f16x3p(p, st, kname);
        static const int s2_on = getenv("LSSVC_F16X3_S2") ? atoi(getenv("LSSVC_F16X3_S2")) : 1;
        if (s2_on && vec && sd == 2 && ks == 3 && d->in_act != LSSVC_INACT_SQUARE && conv3s2_f16x3p_wanted(p))
            return dispatch_conv3s2_f16x3p(p, st, kname);
        if (s2_on && vec && sd == 2 && ks == 3 && RPW <= 2) {
            snprintf(kname, 96, "conv_f16x3_kernel<%d, %d, 3, 2>", MF, RPW);
            return dispatch_tile_f16x3_s2<3>(p, MF, RPW, st);
        }
        if (vec && sd == 1 && ks == 7 && d->in_act != LSSVC_INACT_SQUARE && option_get(OPT_P7_ON) && conv7_f16x3p_wanted(p))
            return dispatch_conv7_f16x3p(p, st, kname);
        if (vec && sd == 1 && (ks == 3 || ks == 7)) {
            snprintf(kname, 96, "conv_f16x3_kernel<%d, %d, %d, 1>", MF, RPW, ks);
            return ks == 3 ? dispatch_tile_f16x3<3, 1>(p, MF, RPW, st) : dispatch_tile_f16x3<7, 1>(p, MF, RPW, st);
        }
        // 1x1: streaming kernel with LDS-resident weights; GDN (square + normalise) stays on the exact path
        if (vec && sd == 1 && ks == 1 && ((chunks16 + 1) / 2) * 2 * 1024 <= kPwBigLds &&
            (d->epilogue == LSSVC_EPI_NONE || ((chunks16 + 1) / 2) * 2 * 1024 <= kPwMaxLds))
            return dispatch_pw_f16x3(p, st, kname);
    }
    snprintf(kname, 96, "conv_mfma_kernel<%d, %d, %d, %d, %s>", MF, RPW, ks, sd, vec ? "true" : "false");
#define LSSVC_CONV_KS(K, SD) \
    if (ks == K && sd == SD) return vec ? dispatch_tile<K, SD, true>(p, MF, RPW, st) : dispatch_tile<K, SD, false>(p, MF, RPW, st);
    LSSVC_CONV_KS(1, 1) LSSVC_CONV_KS(1, 2) LSSVC_CONV_KS(2, 1) LSSVC_CONV_KS(3, 1) LSSVC_CONV_KS(3, 2) LSSVC_CONV_KS(7, 1)
#undef LSSVC_CONV_KS
    return fail("conv2d: no kernel for %dx%d stride %d", d->KH, d->KW, sd);
}
