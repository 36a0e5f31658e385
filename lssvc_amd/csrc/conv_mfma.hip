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
    float out_scale;
    int pixel_shuffle;
    V out;
    int Hout, Wout;  // conv-space output size (before pixel shuffle)
    int tiles_x, tiles_y, m_tiles;
    int PH, PW;      // LDS patch size in pixels
    int in_vec[LSSVC_CONV_MAX_INPUTS];
    int out_vec, res_vec, gdn_vec;
};

constexpr int CP = 12;  // LDS row pitch in floats (CK=8 + 4 pad)

__device__ __forceinline__ float in_activate(float v, int mode, float slope) {
    if (mode == LSSVC_INACT_LRELU) return v > 0.f ? v : v * slope;
    if (mode == LSSVC_INACT_SQUARE) return v * v;
    return v;
}

template <int MF, int RPW>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int TM = 16 * MF;
    constexpr int TH = 4 * RPW;
    float *patch = smem;                          // [PH*PW][CP]
    float *wts = smem + p.PH * p.PW * CP;         // [KW][TM][CP]

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

    const int s = p.stride;
    const int gy0 = oy0 * s - p.pad_t, gx0 = ox0 * s - p.pad_l;
    const int patch_items = p.PH * p.PW * 2;
    const int w_items = p.KW * TM * 2;

    int kc = 0;  // global chunk index into the weight tensor
    for (int seg = 0; seg < p.n_in; ++seg) {
        const V X = p.in[seg];
        const int vec = p.in_vec[seg];
        for (int c0 = 0; c0 < X.C; c0 += LSSVC_CONV_CK, ++kc) {
            // ---- stage the input patch for channels [c0, c0+8) ----------------------------------
            for (int idx = tid; idx < patch_items; idx += 256) {
                const int pix = idx >> 1, half = idx & 1;
                const int py = pix / p.PW, px = pix - py * p.PW;
                const int gy = gy0 + py, gx = gx0 + px;
                const int c = c0 + half * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (gy >= 0 && gy < X.H && gx >= 0 && gx < X.W && c < X.C) {
                    const float *src = X.p + ((size_t)gy * X.W + gx) * X.ld + c;
                    if (vec) {
                        v = *reinterpret_cast<const float4 *>(src);
                    } else {
                        v.x = src[0];
                        if (c + 1 < X.C) v.y = src[1];
                        if (c + 2 < X.C) v.z = src[2];
                        if (c + 3 < X.C) v.w = src[3];
                    }
                    if (p.in_act != LSSVC_INACT_NONE) {
                        v.x = in_activate(v.x, p.in_act, p.in_slope);
                        v.y = in_activate(v.y, p.in_act, p.in_slope);
                        v.z = in_activate(v.z, p.in_act, p.in_slope);
                        v.w = in_activate(v.w, p.in_act, p.in_slope);
                    }
                }
                *reinterpret_cast<float4 *>(patch + pix * CP + half * 4) = v;
            }
            for (int ky = 0; ky < p.KH; ++ky) {
                // ---- stage the filter slab of kernel row ky -------------------------------------
                const float *wsrc = p.w + ((size_t)(kc * p.KH + ky) * p.KW) * p.M_pad * 8;
                for (int idx = tid; idx < w_items; idx += 256) {
                    const int kx = idx / (TM * 2);
                    const int r = idx - kx * (TM * 2);
                    const int m = r >> 1, half = r & 1;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (m0 + m < p.M_pad)
                        v = *reinterpret_cast<const float4 *>(wsrc + ((size_t)kx * p.M_pad + m0 + m) * 8 + half * 4);
                    *reinterpret_cast<float4 *>(wts + (kx * TM + m) * CP + half * 4) = v;
                }
                __syncthreads();
                // ---- MFMA over the KW taps of this kernel row -----------------------------------
                for (int kx = 0; kx < p.KW; ++kx) {
                    float2 a[MF], b[RPW];
#pragma unroll
                    for (int f = 0; f < MF; ++f)
                        a[f] = *reinterpret_cast<const float2 *>(wts + (kx * TM + f * 16 + li) * CP + 2 * lg);
#pragma unroll
                    for (int r = 0; r < RPW; ++r) {
                        const int row = wave * RPW + r;
                        const int ppix = (row * s + ky) * p.PW + li * s + kx;
                        b[r] = *reinterpret_cast<const float2 *>(patch + ppix * CP + 2 * lg);
                    }
#pragma unroll
                    for (int f = 0; f < MF; ++f)
#pragma unroll
                        for (int r = 0; r < RPW; ++r) {
                            acc[f][r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[f].x, b[r].x, acc[f][r], 0, 0, 0);
                            acc[f][r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[f].y, b[r].y, acc[f][r], 0, 0, 0);
                        }
                }
                __syncthreads();
            }
        }
    }

    // ---- fused epilogue: bias -> GDN -> activation -> residual -> scale -> (pixel-shuffle) store ----
    const int ox = ox0 + li;
    const int cps = p.Cout >> 2;  // channels after pixel shuffle
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int oy = oy0 + wave * RPW + r;
        if (oy >= p.Hout || ox >= p.Wout) continue;
        const size_t opix = (size_t)oy * p.Wout + ox;
#pragma unroll
        for (int f = 0; f < MF; ++f) {
            const int mb = m0 + f * 16 + 4 * lg;
            if (mb >= p.Cout) continue;
            float v[4] = {acc[f][r][0], acc[f][r][1], acc[f][r][2], acc[f][r][3]};
            const bool full = (mb + 3 < p.Cout);
            if (p.bias) {
                const float4 bb = *reinterpret_cast<const float4 *>(p.bias + mb);  // bias is M_pad long
                v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
            }
            if (p.epilogue != LSSVC_EPI_NONE) {
                float x[4] = {0.f, 0.f, 0.f, 0.f};
                const float *xs = p.gdn_x.p + opix * p.gdn_x.ld + mb;
                if (full && p.gdn_vec) {
                    const float4 t = *reinterpret_cast<const float4 *>(xs);
                    x[0] = t.x; x[1] = t.y; x[2] = t.z; x[3] = t.w;
                } else {
                    for (int j = 0; j < 4; ++j)
                        if (mb + j < p.Cout) x[j] = xs[j];
                }
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
                const float *rs = p.res.p + opix * p.res.ld + mb;
                if (full && p.res_vec) {
                    const float4 t = *reinterpret_cast<const float4 *>(rs);
                    v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
                } else {
                    for (int j = 0; j < 4; ++j)
                        if (mb + j < p.Cout) v[j] += rs[j];
                }
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

template <int MF, int RPW>
static int launch(const ConvP &p, hipStream_t st) {
    const int TH = 4 * RPW;
    ConvP q = p;
    q.tiles_x = (p.Wout + 15) / 16;
    q.tiles_y = (p.Hout + TH - 1) / TH;
    q.m_tiles = (p.M_pad / 16 + MF - 1) / MF;
    q.PH = (TH - 1) * p.stride + p.KH;
    q.PW = 15 * p.stride + p.KW;
    const size_t lds = (size_t)(q.PH * q.PW + p.KW * 16 * MF) * CP * sizeof(float);
    if (lds > 64 * 1024) return fail("conv2d: LDS tile of %zu bytes exceeds 64 KiB", lds);
    const long long blocks = (long long)q.tiles_x * q.tiles_y * q.m_tiles;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return fail("conv2d: bad grid %lld", blocks);
    hipLaunchKernelGGL((conv_mfma_kernel<MF, RPW>), dim3((unsigned)blocks), dim3(256), lds, st, q);
    return launch_status("conv2d");
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
}

}  // namespace lssvc

using namespace lssvc;

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
    LSSVC_CHECK(d->KH >= 1 && d->KH <= 7 && d->KW >= 1 && d->KW <= 7, "conv2d: kernel %dx%d", d->KH, d->KW);
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
    if (d->epilogue != LSSVC_EPI_NONE) {
        LSSVC_CHECK(view_ok(&d->gdn_x) && same_shape(&d->gdn_x, &d->out), "conv2d: gdn_x shape mismatch");
        p.gdn_x = mk(&d->gdn_x);
        p.gdn_vec = vec4_ok(&d->gdn_x);
    } else {
        p.gdn_x = mk_null();
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);

    int MF, RPW;
    pick_variant(p, MF, RPW);

#define LSSVC_CONV_CASE(mf, rpw) \
    if (MF == mf && RPW == rpw) return launch<mf, rpw>(p, st);
    LSSVC_CONV_CASE(1, 1) LSSVC_CONV_CASE(1, 2) LSSVC_CONV_CASE(1, 4)
    LSSVC_CONV_CASE(2, 1) LSSVC_CONV_CASE(2, 2) LSSVC_CONV_CASE(2, 4)
    LSSVC_CONV_CASE(3, 1) LSSVC_CONV_CASE(3, 2) LSSVC_CONV_CASE(3, 4)
    LSSVC_CONV_CASE(4, 1) LSSVC_CONV_CASE(4, 2) LSSVC_CONV_CASE(4, 4)
#undef LSSVC_CONV_CASE
    return fail("conv2d: no kernel for MF=%d RPW=%d", MF, RPW);
}
