/*
 * lssvc_hip.h -- C ABI of liblssvc_hip.so, the MI355X (gfx950) kernels behind LSSVC's per-frame
 * encode/decode hot path.
 *
 * The reference has no FFI registry for this path: its "operators" are the PyTorch ATen calls made
 * by the nn.Modules under src/IntraModules and src/InterModules. Each entry
 * point below replaces one such operator class; the reference call sites are cited per function.
 * The host-side mirror of the reference's model API (IntraSS / LSSVC_extend) lives in Python
 * (lssvc_amd/) and reaches these symbols through ctypes -- see INTEGRATION.md.
 *
 * Conventions
 *   - All tensors are fp32, batch 1, NHWC ("pixel-major") in device memory: element (y, x, c) of a
 *     view lives at ptr[(y * W + x) * ld + c]; ld >= C lets a view be a channel slice of a wider
 *     buffer (that is how torch.cat / chunk along channels are made free).
 *   - Every call enqueues work on `stream` (a hipStream_t passed as void*) and returns at once.
 *   - Return value: 0 on success, non-zero on error; lssvc_last_error() gives the message
 *     (thread-local). Shapes are validated on the host BEFORE any launch.
 */
#ifndef LSSVC_HIP_H
#define LSSVC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lssvc_view {
    float *ptr; /* device pointer */
    int32_t H, W, C;
    int32_t ld; /* elements between consecutive pixels (>= C) */
} lssvc_view;

/* ---- activation / epilogue selectors ------------------------------------------------------- */
enum { LSSVC_ACT_NONE = 0, LSSVC_ACT_LRELU = 1, LSSVC_ACT_RELU = 2 };
enum { LSSVC_INACT_NONE = 0, LSSVC_INACT_LRELU = 1, LSSVC_INACT_SQUARE = 2 };
/* GDN epilogues: v = conv(x^2, gamma) + beta;  out = one of
 *   X_MUL_RSQRT : x * (1/sqrt(v))   IntraModules GDN forward      (gdn.py:29-44)
 *   X_MUL_SQRT  : x * sqrt(v)       both IGDN flavours            (gdn.py:38-39, video_net_component.py:100-101)
 *   X_DIV_SQRT  : x / sqrt(v)       InterModules GDN forward      (video_net_component.py:102-103) */
enum { LSSVC_EPI_NONE = 0, LSSVC_EPI_X_MUL_RSQRT = 1, LSSVC_EPI_X_MUL_SQRT = 2, LSSVC_EPI_X_DIV_SQRT = 3 };

/* Conv arithmetic. F16X3: operands split x = hi + lo in fp16, hi*hi + hi*lo + lo*hi accumulated in fp32 on
 * v_mfma_f32_16x16x32_f16 -- fp32-class accuracy (holds the 1e-5 bpp / 1e-4 dB bars; plain fp16 does not) at
 * 3/8 of the fp32-MFMA pipe cycles x 16x the rate. Used for 3x3 / 7x7 stride-1 layers; others stay F32. */
enum { LSSVC_PREC_F32 = 0, LSSVC_PREC_F16X3 = 1 };
/* Flags OR-ed into lssvc_conv_desc.precision beside LSSVC_PREC_F16X3 (round 5): PRE-SPLIT activation tensors. A pre-split view
 * holds, per pixel and 16-channel chunk, 64 bytes: [hi: 16 x fp16 | lo: 16 x fp16] with hi = fp16(x), lo = fp16(x - hi) of the
 * saturated value x -- what every f16x3 conv makes of its fp32 input while staging it -- so it takes the bytes of the fp32 tensor
 * (ptr 64-byte aligned, ld %% 16 == 0 in 4-byte units, channels >= C of the last chunk zero; 16-channel slices of it are views
 * again). SPLIT_IN: every input of the conv is pre-split, with the conv's input activation ALREADY APPLIED by whoever wrote it
 * (in_act must be NONE): the persistent 3x3 kernels then move their halo patches global -> LDS by DMA, no conversion in the
 * kernel. Same hi / lo values either way: results are bit-identical to the fp32-input form. */
#define LSSVC_PREC_MASK 0xff
#define LSSVC_PREC_SPLIT_IN 0x100

#define LSSVC_CONV_MAX_INPUTS 3
#define LSSVC_CONV_CK 8 /* input channels per K-chunk; each input segment is zero-padded to a multiple */

/*
 * 2-D convolution as an LDS-tiled implicit GEMM on fp32 MFMA (v_mfma_f32_16x16x4_f32).
 * Replaces nn.Conv2d / F.conv2d (3x3 s1/s2, 7x7, 1x1, GDN's 1x1), nn.ConvTranspose2d (host rewrites
 * it as a 2x2 or flipped 3x3 conv) and the torch.cat in front of it (up to 3 input views are read as
 * one virtual concat), with fused bias, input activation, GDN epilogue, output activation, residual
 * add, output scale and PixelShuffle(2) store.
 * Reference call sites: layers.py:36-57, video_net_component.py:11-32,191-210, lssvc_modules.py:15-72.
 *
 * weight layout (prepared on the host, see lssvc_amd/weights.py):
 *   w[chunk][ky][kx][m][8]  with chunk over the CK=8-padded concatenated input channels and
 *   m over output channels zero-padded to a multiple of 16 (M_pad).
 * With pixel_shuffle = 1 the m axis must already be permuted to (dy,dx)-major: m = q*(Cout/4) + c.
 * bias: M_pad floats (same permutation) or NULL.
 */
typedef struct lssvc_conv_desc {
    lssvc_view in[LSSVC_CONV_MAX_INPUTS];
    int32_t n_in;
    const float *weight;
    const float *bias;
    int32_t KH, KW, stride, pad_t, pad_l;
    int32_t Cout;  /* true output channels (before pixel shuffle) */
    int32_t M_pad; /* Cout rounded up to a multiple of 16 */
    int32_t in_act;
    float in_slope;
    int32_t epilogue; /* LSSVC_EPI_* ; needs gdn_x */
    lssvc_view gdn_x; /* same H,W,Cout as the conv result */
    int32_t act;
    float slope;
    lssvc_view residual; /* added after the activation; ptr NULL = none; same shape as `out` */
    float out_scale;     /* multiplied last; 1.0f = none */
    int32_t pixel_shuffle; /* 0 or 1 (r = 2) */
    lssvc_view out;      /* H_out x W_out x Cout, or 2H_out x 2W_out x Cout/4 with pixel_shuffle */
    int32_t precision;   /* LSSVC_PREC_F32 (exact fp32 MFMA, the parity reference path) or LSSVC_PREC_F16X3 */
    const void *weight16; /* F16X3 only: fp16 weights [hi|lo][chunk16][ky][kx][m][16] (lssvc_amd/weights.py) */
    float weight16_unscale; /* F16X3 only: weight16 holds w * 2^e (a power of two chosen so that the lo parts are
                             * normal fp16 numbers); the accumulators are multiplied by this 2^-e (exact) before the
                             * epilogue. 0 is read as 1. */
    lssvc_view residual2; /* a second tensor added after `residual` (out = (act(conv) + residual) + residual2): the skip
                           * sums that follow a ResBlock, e.g. MultiScaleContextFusion's `context1 + res_block1_out(...)`
                           * (lssvc_modules.py:226-231), folded into the block's last conv. ptr NULL = none; needs
                           * `residual`, 16-byte addressable views and the plain (no GDN / shuffle / scale) epilogue. */
} lssvc_conv_desc;

int lssvc_conv2d(const lssvc_conv_desc *d, void *stream);

/* Which template instantiation lssvc_conv2d picks for a conv-space output of Hout x Wout x M_pad at
 * `stride`: returns MF*16 + RPW (kernel name conv_mfma_kernel<MF,RPW>; tile = 4*RPW rows x 16 cols x
 * 16*MF channels). Lets profilers attribute launches to kernels; no GPU work. */
int lssvc_conv2d_variant(int32_t Hout, int32_t Wout, int32_t M_pad, int32_t stride);
/* Name (as rocprofv3 prints it, without the lssvc:: prefix) of the kernel the calling thread's last lssvc_conv2d launched. */
const char *lssvc_conv2d_last_kernel(void);

/* The per-pixel tail of a DepthConvBlock in one launch (f16x3 arithmetic; src/models/lssvc_modules.py:15-72):
 *     o1  = pre_w * pre_in + pre_bias + ident      (DepthConv.conv2 + identity/adaptor; skipped when pre_w16 == NULL: o1 = x)
 *     out = o1 + lrelu(w2 * lrelu(w1 * o1 + b1) + b2) [+ skip]                 (ConvFFN, both slopes = `slope`; `skip`,
 *           when its ptr is not NULL, is an outer skip connection added last, e.g. lssvc_modules.py:363)
 * C = out.C in {32, 48, 64, 96, 128}; hidden %% 32 == 0; pre_in.C %% 8 == 0 and <= 128. Up to C = 64 / pre_in.C = 64
 * with all weights within the 160 KB LDS they stay resident; otherwise the hidden dimension is streamed through LDS in
 * 32-channel slices by LDS-DMA (lssvc_ffn_f16x3_lds_bytes gives the LDS either way; it must be <= 160 KB). Weight blobs are fp16 [hi plane | lo plane] images of w * 2^e in the fragment
 * order the kernel reads (lssvc_amd/weights.py: layout_ffn_f16x3), *_unscale = 2^-e; biases are fp32, padded to
 * 16*ceil(C/16) (b1: hidden). `out` may alias `x` or `ident`. */
typedef struct {
    lssvc_view x;
    lssvc_view pre_in;
    const void *pre_w16;
    float pre_unscale;
    const float *pre_bias;
    lssvc_view ident;
    const void *w1_16;
    float w1_unscale;
    const float *b1;
    int32_t hidden;
    const void *w2_16;
    float w2_unscale;
    const float *b2;
    float slope;
    lssvc_view out;
    lssvc_view skip;   /* optional, same shape as out; may alias out */
} lssvc_ffn_desc;
int lssvc_ffn_f16x3(const lssvc_ffn_desc *d, void *stream);
int64_t lssvc_ffn_f16x3_lds_bytes(int32_t C, int32_t hidden, int32_t pre_cin);
/* 1 if lssvc_ffn_f16x3 runs the streamed-weights kernel for this shape (profiling label only). */
int lssvc_ffn_f16x3_is_streamed(int32_t C, int32_t hidden, int32_t pre_cin);

/* DepthConv's front half in one launch (f16x3 arithmetic; src/models/lssvc_modules.py:15-44):
 *     out = depthwise3x3(act(conv1x1(in...) + bias)) + dw_bias
 * `d` describes the leading 1x1 conv exactly as for lssvc_conv2d (precision F16X3, up to 3 concatenated inputs with at
 * most 64 channels in 16-channel chunks, Cout = C in {32, 48, 64}, act none / LeakyReLU, no residual / GDN / shuffle);
 * d->out is the FINAL H x W x C view. dw_weight is [9][C] (tap-major), dw_bias [C], as for lssvc_dwconv3x3. Results
 * are bit-identical to lssvc_conv2d followed by lssvc_dwconv3x3. */
int lssvc_conv1x1_dw3x3_f16x3(const lssvc_conv_desc *d, const float *dw_weight, const float *dw_bias, void *stream);

/* Depthwise 3x3, stride 1, pad 1 (lssvc_modules.py:23-24). weight: [9][C], bias: [C]. */
int lssvc_dwconv3x3(const lssvc_view *in, const float *weight, const float *bias, const lssvc_view *out,
                    void *stream);

/* F.interpolate(mode='bilinear', align_corners=False) to out->H x out->W, result times `scale`
 * (layers.py:269,284; lssvc_modules.py:360,393,425; video_net_component.py:355-368 incl. the
 * "*2.0" / "/2" that always follows it on flows). */
int lssvc_resize_bilinear(const lssvc_view *in, const lssvc_view *out, float scale, void *stream);

/* flow_warp = grid_sample(bilinear, border, align_corners=True) with the reference's
 * linspace(-1,1)+flow/((size-1)/2) grid (video_net_component.py:329-352). flow: H x W x 2 (dx,dy). */
int lssvc_flow_warp(const lssvc_view *in, const lssvc_view *flow, const lssvc_view *out, void *stream);

/* F.avg_pool2d / nn.MaxPool2d, kernel 2 stride 2 (video_net_component.py:230-233, lssvc_modules.py:298). */
int lssvc_pool2x2(const lssvc_view *in, const lssvc_view *out, int32_t is_max, void *stream);

/* out = a*wa + b*wb, weights = softmax over the 2 channels of `logits` (lssvc_modules.py:133-153 with
 * LSSVC_net.py:253-255). */
int lssvc_softmax2_blend(const lssvc_view *a, const lssvc_view *b, const lssvc_view *logits,
                         const lssvc_view *out, void *stream);

/* One level of ME_Spynet / ME_Spynet_DCVC's coarse-to-fine loop (video_net_component.py:231-246, 308-324) up to the conv stack:
 * up = 2 * bilinear_x2(flow_lo); out = cat(im1, warp(im2, up), up) as an H x W x 8 view. One launch for what is a resize, a copy
 * and a flow warp otherwise, with their arithmetic. im1 / im2: H x W x 3, flow_lo: (H/2) x (W/2) x 2 (zeros at the coarsest level). */
int lssvc_spynet_prep(const lssvc_view *im1, const lssvc_view *im2, const lssvc_view *flow_lo, const lssvc_view *out, void *stream);

/* l1, l2, l3 = F.avg_pool2d(kernel 2, stride 2) applied once, twice and three times to `in` (ME_Spynet's image pyramid,
 * video_net_component.py:225-229) in one launch, with lssvc_pool2x2's arithmetic level by level. in: H x W x C with H, W % 8 == 0. */
int lssvc_avgpool_pyramid3(const lssvc_view *in, const lssvc_view *l1, const lssvc_view *l2, const lssvc_view *l3, void *stream);

/* out = a + b (channel-sliced views allowed) -- the residual sums outside convs. */
int lssvc_add(const lssvc_view *a, const lssvc_view *b, const lssvc_view *out, void *stream);
/* out = in (strided copy: materialises a torch.cat slice). `out` may have MORE channels than `in` (same H, W): the channels `in`
 * lacks are written as zeros -- the zero-padded 4-channel copy of an RGB / flow tensor in one launch. */
int lssvc_copy(const lssvc_view *in, const lssvc_view *out, void *stream);
/* out = lrelu(in, slope) (the stand-alone nn.LeakyReLU between blocks, e.g. dmc_net.py:178). */
int lssvc_lrelu(const lssvc_view *in, const lssvc_view *out, float slope, void *stream);
/* out = F.pad(in, (left, right, top, bottom), value 0), negative entries cropping; right / bottom follow from out's size
 * (get_depadded_feature: IntraSS.py:124-135, LSSVC_net.py:271-282). Writes every element of `out`. */
int lssvc_pad_crop(const lssvc_view *in, const lssvc_view *out, int32_t left, int32_t top, void *stream);
/* fp32 NHWC view -> PRE-SPLIT view of the same shape (see LSSVC_PREC_SPLIT_IN): out = split(clamp(act(in), +-65504)), act = none or
 * LeakyReLU(in_slope) -- the staging arithmetic of the f16x3 convs (the reference has no counterpart: its convs read fp32,
 * layers.py:36-57). EXPERIMENTAL (round 5: measured 0-7 %, not adopted; the model never calls it): the only producer of pre-split views --
 * no conv epilogue writes the format. `in` and `out` must not overlap (the layout permutes bytes across threads; checked). */
int lssvc_presplit(const lssvc_view *in, const lssvc_view *out, int32_t in_act, float in_slope, void *stream);
/* Zero `nbytes` of device memory / clamp n floats in place (what torch.zeros and the caller's `clamp_(0, 1)` of the
 * reconstructions, test.py:249-250, are for a caller without PyTorch; compiled frame plans record these as launches). */
int lssvc_fill_zero(void *ptr, int64_t nbytes, void *stream);
int lssvc_clamp_inplace(float *x, int64_t n, float lo, float hi, void *stream);
/* Range audit of the f16x3 conv mode: *out_max = max(*out_max, max |x| over the view) as a non-negative float (NaN / Inf
 * give +Inf); the caller zeroes it. The f16x3 kernels split activations into fp16 hi/lo parts and saturate at +-65504 while
 * staging (GDN squares first), which the reference's fp32 convs (e.g. src/IntraModules/gdn.py:29-44) do not: the host runs
 * this on every f16x3 conv input of the first frame of each type and moves layers that come near the limit to the exact
 * fp32 kernel (lssvc_amd/hip_ops.py RangeAudit). */
int lssvc_absmax(const lssvc_view *x, float *out_max, void *stream);

/* OffsetDiversity tail (lssvc_modules.py:96-110): from the up-sampled conv_offset output `om`
 * (H x W x 96: o1[32] | o2[32] | mask[32]) and flow (H x W x 2): 32 warps of 3-channel groups of x
 * (H x W x 48) by 40*tanh(offset)+flow, times sigmoid(mask), then the grouped (g=16) 1x1 fusion
 * conv (fusion_w: [48][6] as in nn.Conv2d.weight, fusion_b: [48]). The reference's
 * view(B,96,H,W) channel interleave is reproduced exactly. */
int lssvc_offset_diversity(const lssvc_view *x, const lssvc_view *om, const lssvc_view *flow,
                           const float *fusion_w, const float *fusion_b, const lssvc_view *out, void *stream);

/* NCHW (boundary layout of the reference's tensors) <-> NHWC view. */
int lssvc_nchw_to_nhwc(const float *src, const lssvc_view *dst, void *stream);
int lssvc_nhwc_to_nchw(const lssvc_view *src, float *dst, void *stream);

/* ---- entropy models ---------------------------------------------------------------------------
 * Bit counts are accumulated in fp64, deterministically (per-block partials + one final pass, no
 * float atomics): bits_out[0] = sum over elements. `workspace` must hold lssvc_reduce_workspace_bytes().
 */
int64_t lssvc_reduce_workspace_bytes(void);

/* Quantise around a mean and price with a zero-mean Laplace(sigma) (LSSVC_net.py:154-161,
 * dmc_net.py:370-377,429-431): q = rint(y - mean); y_hat = q + mean;
 * bits += clamp(-log2(cdf(q+.5) - cdf(q-.5) + 1e-5), 0, 50), sigma clamped to [1e-5, 1e10].
 * y_q / y_hat may be NULL views (ptr = NULL) when not needed. */
int lssvc_laplace_quant_bits(const lssvc_view *y, const lssvc_view *mean, const lssvc_view *sigma,
                             const lssvc_view *y_q, const lssvc_view *y_hat, double *bits_out,
                             void *workspace, void *stream);

/* One step of the 4-step spatial/channel checkerboard (LSSVC_net.py:288-443). For channel quarter
 * c (C/4 channels each) only the 2x2 position mask_of_chunk[c] is touched:
 *   q = rint(y - mean); y_q = q; y_hat = q + mean; sigma_hat = sigma; elsewhere untouched.
 * y_q / y_hat / sigma_hat accumulate across the 4 steps (host zero-fills before step 1). */
int lssvc_four_part_step(const lssvc_view *y, const lssvc_view *mean, const lssvc_view *sigma,
                         const int32_t mask_of_chunk[4], const lssvc_view *y_q, const lssvc_view *y_hat,
                         const lssvc_view *sigma_hat, void *stream);

/* Laplace bits of already-quantised symbols (LSSVC_net.py:504). */
int lssvc_laplace_bits(const lssvc_view *y_q, const lssvc_view *sigma, double *bits_out, void *workspace,
                       void *stream);

/* z_hat = rint(z); bits += clamp(-log2(BitEstimator(z_hat+.5) - BitEstimator(z_hat-.5) + 1e-5), 0, 50)
 * (LSSVC_net.py:163-167, video_entropy_models.py:110-166). params: [11][C] rows =
 * softplus(h1),b1,tanh(a1), softplus(h2),b2,tanh(a2), softplus(h3),b3,tanh(a3), softplus(h4),b4. */
int lssvc_factorized_quant_bits(const lssvc_view *z, const float *params, const lssvc_view *z_hat,
                                double *bits_out, void *workspace, void *stream);

/* GaussianConditional.forward, eval mode (img_entropy_models.py:650-685): y_hat = round(y-mu)+mu,
 * lik = .5erfc(-(.5-|v|)/s/sqrt2) - .5erfc(-(-.5-|v|)/s/sqrt2), v = (round(y-mu)+mu)-mu,
 * s = max(scale, 0.11), lik >= 1e-9; sum_out[0] = sum ln(lik) (natural log; the caller divides
 * by -ln2 as IntraSS.py:163 does). y_q (optional) receives the symbols round(y-mu) for the coder. */
int lssvc_gaussian_conditional(const lssvc_view *y, const lssvc_view *scale, const lssvc_view *mean,
                               const lssvc_view *y_hat, const lssvc_view *y_q, double *sum_out, void *workspace,
                               void *stream);

/* EntropyBottleneck.forward, eval mode (img_entropy_models.py:483-554): z_hat = round(z-med)+med,
 * lik = |sigmoid(s*u) - sigmoid(s*l)| >= 1e-9 with the 1-3-3-3-3-1 per-channel MLP.
 * params: [59][C] rows = softplus(matrices) (3+9+9+9+3), biases (3+3+3+3+1), tanh(factors) (3*4), median. */
int lssvc_entropy_bottleneck(const lssvc_view *z, const float *params, const lssvc_view *z_hat,
                             const lssvc_view *z_q, double *sum_out, void *workspace, void *stream);

/* Symbol / index planes for the host coder, flat NCHW int32 as the reference flattens them
 * (video_entropy_models.py:234-236,315-319): sym = (int) q, idx = table index of sigma (or the channel
 * number when sigma is NULL: BitEstimator / EntropyBottleneck tables are per channel). With chunk_of_mask
 * (4 entries, a permutation of 0..3) the C-channel inputs are folded to C/4 channels as the 4-step prior
 * writes them (LSSVC_net.py:432-442): 2x2 position m takes channel chunk chunk_of_mask[m]. */
int lssvc_export_symbols(const lssvc_view *q, const lssvc_view *sigma, const int32_t *chunk_of_mask, float log_min,
                         float log_step, float add, int32_t levels, int32_t *sym_nchw, int32_t *idx_nchw, void *stream);
/* Decoder side: out = sym + mean + channel_add[c] (means / medians optional); with chunk_of_mask only the
 * channel chunk of each 2x2 position is written (LSSVC_net_extend.py:208-213). sym_nchw is a device pointer. */
int lssvc_import_symbols(const int32_t *sym_nchw, const lssvc_view *mean, const float *channel_add,
                         const int32_t *chunk_of_mask, const lssvc_view *out, void *stream);
/* The same two with 16-bit planes (what replaces the reference's `.tolist()` of ~2.3 M symbols per
 * P-frame, video_entropy_models.py:234-236,317-319, is an int16 plane in pinned memory): half the PCIe bytes of the
 * int32 form. A symbol outside [-32768, 32767] sets *overflow (device int32, zeroed by the caller) so that the host can
 * refuse the plane instead of coding a truncated value. */
int lssvc_export_symbols_i16(const lssvc_view *q, const lssvc_view *sigma, const int32_t *chunk_of_mask, float log_min,
                             float log_step, float add, int32_t levels, int16_t *sym_nchw, int16_t *idx_nchw, int32_t *overflow,
                             void *stream);
int lssvc_import_symbols_i16(const int16_t *sym_nchw, const lssvc_view *mean, const float *channel_add,
                             const int32_t *chunk_of_mask, const lssvc_view *out, void *stream);

/* sigma -> CDF-table index planes for the host coder (video_entropy_models.py:309-313,
 * img_entropy_models.py:687-691): idx = clamp((ln max(s,1e-5) - ln smin)/step + add, 0, levels-1). */
int lssvc_build_indexes(const lssvc_view *sigma, float log_min, float log_step, float add, int32_t levels,
                        int32_t *idx_nhwc, void *stream);

/* ---- frame pre/post-processing around the codec (csrc/prepost.hip) ---------------------------------
 * What the reference's test.py does to every frame on the host before and after encode_decode
 * (test.py:185-201, 249-311), here on the device so that only 8-bit planes go up and a few scalars come back. */

/* 8-bit planar 4:2:0 (device pointers; y: HxW, u, v: H/2 x W/2) -> RGB fp32 NHWC frame [0,1]: ycbcr420_to_rgb
 * (src/utils/functional.py:42-58: chroma x2 by linear interpolation at scipy.ndimage.zoom's sample positions, BT.709,
 * clip). `frame` may be larger than HxW: the rest is the zero inter-layer padding (test.py:191-193).
 * y_norm / u_norm / v_norm (optional): the planes / 255 that the per-plane PSNRs are taken against. */
int lssvc_yuv420_to_frame(const uint8_t *y, const uint8_t *u, const uint8_t *v, int32_t H, int32_t W, const lssvc_view *frame,
                          float *y_norm, float *u_norm, float *v_norm, void *stream);
/* 8-bit planar RGB (3 x H x W) -> fp32 NHWC frame, x / 255, zero-padded to the frame's size. */
int lssvc_rgb8_to_frame(const uint8_t *rgb, int32_t H, int32_t W, const lssvc_view *frame, void *stream);
/* Separable K-tap resampling with host-built tables ([n_out][K] weights and source indexes per axis), vertical pass
 * then horizontal pass, result clamped to [clamp_lo, clamp_hi]: the MATLAB-style antialiased bicubic `imresize`
 * (src/utils/core.py:276-432) that makes the base-layer frame (test.py:194-199); tables: lssvc_amd/preprocess.py. */
int lssvc_resample2d(const lssvc_view *in, const lssvc_view *out, const float *w_v, const int32_t *idx_v, int32_t k_v,
                     const float *w_h, const int32_t *idx_h, int32_t k_h, float clamp_lo, float clamp_hi, void *stream);
/* rgb_to_ycbcr420 (functional.py:16-39) of the top-left h x w crop of an RGB frame (optionally clamped to [0,1] first,
 * test.py:249-250): y (h x w), u, v (h/2 x w/2) fp32 planes. */
int lssvc_rgb_to_yuv420(const lssvc_view *rgb, int32_t h, int32_t w, int32_t clamp01, float *y, float *u, float *v, void *stream);
/* out[0] = sum over the top-left h x w crop, all channels, of (clamp01 ? clamp(a,0,1) : a  -  b)^2, accumulated in fp64
 * in a fixed order (PSNR = 10 log10(1 / (out / n)), test.py:104-118). workspace: lssvc_reduce_workspace_bytes(). */
int lssvc_sqdiff_sum(const lssvc_view *a, const lssvc_view *b, int32_t h, int32_t w, int32_t clamp01, double *out, void *workspace,
                     void *stream);
int lssvc_sqdiff_sum_flat(const float *a, const float *b, int64_t n, double *out, void *workspace, void *stream);

/* ---- host entropy coder (write_stream = 1) ------------------------------------------------------
 * Replaces the reference's pybind11 modules MLCodec_rans (BufferedRansEncoder / RansDecoder,
 * src/cpp/rans/rans_interface.cpp:85-261) and MLCodec_CXX (pmf_to_quantized_cdf, src/cpp/ops/ops.cpp:24-91).
 * Runs on the host; symbols / indexes are flat int32 planes (NCHW order, as the reference flattens them). */
typedef struct lssvc_cdf_table {
    const int32_t *cdfs;    /* [n_cdfs][stride] quantised CDFs (16-bit precision) */
    int32_t n_cdfs, stride;
    const int32_t *sizes;   /* [n_cdfs] entries used per row (pmf length + 2) */
    const int32_t *offsets; /* [n_cdfs] symbol value of table slot 0 */
} lssvc_cdf_table;

void *lssvc_rans_encoder_new(void);
void lssvc_rans_encoder_free(void *enc);
void lssvc_rans_encoder_reset(void *enc);
/* append n symbols to the pending list (BufferedRansEncoder.encode_with_indexes) */
int lssvc_rans_encode_with_indexes(void *enc, const int32_t *symbols, const int32_t *indexes, int64_t n,
                                   const lssvc_cdf_table *table);
int lssvc_rans_encode_with_indexes_i16(void *enc, const int16_t *symbols, const int16_t *indexes, int64_t n,
                                       const lssvc_cdf_table *table);
/* entropy-code everything pending; returns the stream length in bytes, lssvc_rans_encoder_bytes() its data */
int64_t lssvc_rans_encoder_flush(void *enc);
const uint8_t *lssvc_rans_encoder_bytes(void *enc);

void *lssvc_rans_decoder_new(void);
void lssvc_rans_decoder_free(void *dec);
int lssvc_rans_decoder_set_stream(void *dec, const uint8_t *bytes, int64_t n);
/* decode n symbols; the cursor persists across calls (RansDecoder.decode_stream) */
int lssvc_rans_decode_stream(void *dec, const int32_t *indexes, int64_t n, const lssvc_cdf_table *table, int32_t *out);
/* 16-bit planes: same streams; a decoded symbol outside [-32768, 32767] is an error */
int lssvc_rans_decode_stream_i16(void *dec, const int16_t *indexes, int64_t n, const lssvc_cdf_table *table, int16_t *out);

/* cdf_out has n + 1 entries */
int lssvc_pmf_to_quantized_cdf(const float *pmf, int32_t n, int32_t precision, uint32_t *cdf_out);

/* ---- checkpoint -> kernel layouts, on the host (csrc/weight_prep.cpp) -------------------------------------------------------
 * What nn.Module.load_state_dict + the first forward do for the reference (src/models/IntraSS.py:190-214,
 * src/models/LSSVC_net.py:141-149): here a checkpoint is a table of named fp32 tensors exactly as the reference's state dict
 * holds them (OIHW conv weights, GDN beta / gamma, BitEstimator / EntropyBottleneck parameters; 'module.' prefix already
 * stripped), and one call turns one layer's tensors into the blobs the kernels consume. The Python front end
 * (lssvc_amd/weights.py) and the engine (lssvc_engine_load_checkpoint) both go through it. Host memory in, host memory out. */
typedef struct lssvc_tensor {
    const char *name;      /* state-dict key, e.g. "res_encoder.conv1.weight" */
    const float *data;     /* contiguous, row-major */
    int32_t ndim;          /* <= 4 */
    int64_t shape[4];
} lssvc_tensor;

enum {
    LSSVC_PREP_CONV = 1,            /* <name>.weight [, .bias]           -> [chunk8][ky][kx][m][8] fp32, bias[M_pad]; dims Cout, M_pad, KH, KW */
    LSSVC_PREP_CONV_F16X3 = 2,      /* <name>.weight                     -> fp16 [hi|lo][chunk16][ky][kx][m][16] of w * 2^e; scalars[0] = 2^-e */
    LSSVC_PREP_CONVT = 3,           /* ConvTranspose2d, flag = stride    -> as CONV of the equivalent conv; dims[4] = pad, dims[5] = pixel_shuffle */
    LSSVC_PREP_DWCONV = 4,          /* (C,1,3,3) depthwise               -> [9][C], bias[C] */
    LSSVC_PREP_GDN = 5,             /* flag 0 = gdn.py / 1 = video_net_component.py -> 1x1 conv of gamma on x^2: fp32 weights, beta as bias, fp16 planes */
    LSSVC_PREP_VECTOR = 6,          /* tensor <name> as it is, flattened */
    LSSVC_PREP_BIT_ESTIMATOR = 7,   /* <name>.f1..f4.{h,b,a}             -> [11][C] */
    LSSVC_PREP_ENTROPY_BOTTLENECK = 8, /* <name>._matrices/_biases/_factors/quantiles -> [59][C] */
    LSSVC_PREP_FFN_F16X3 = 9        /* ConvFFN <name>.conv.{0,2} [+ leading 1x1 conv <name2>] -> w1, w2 (fp16 planes), b1, b2 [, wp, bp];
                                       scalars = unscale of w1, w2 [, wp]; dims hidden, C, pre_cin */
};
#define LSSVC_PREP_MAX_BLOBS 8
typedef struct lssvc_prep_spec {
    int32_t kind;
    char name[120];
    char name2[120];
    int32_t splits[3];     /* CONV / CONV_F16X3: channel counts of the concatenated input segments (sum = Cin) */
    int32_t n_splits;
    int32_t flag;          /* CONV*: pixel_shuffle; CONVT: stride; GDN: flavour */
} lssvc_prep_spec;
/* blobs == NULL: size query (n_blobs, blob_bytes, dims). Otherwise blobs[i] points at blob_bytes[i] bytes of host memory to fill. */
int lssvc_prepare_weights(const lssvc_tensor *checkpoint, int32_t n_tensors, const lssvc_prep_spec *spec, int32_t *n_blobs,
                          int64_t blob_bytes[LSSVC_PREP_MAX_BLOBS], float scalars[4], int32_t dims[8], void *const *blobs);

/* ---- engine: compiled frame plans (csrc/plan_runtime.cpp) ------------------------------------------------
 * Frame-level entry points for a caller without Python or PyTorch (last row). They replace, per frame,
 * IntraSS.encode_decode(bin_path=None) = IntraSS.forward (src/models/IntraSS.py:245-249,137-172) and
 * LSSVC.encode_decode(output_path_el=None) = LSSVC.forward_one_frame (src/models/LSSVC_net.py:172-185,445-528).
 * A plan file holds ONE frame type at ONE size: the fixed sequence of library launches the Python front end issues for it
 * (lssvc_amd/plan_compiler.py: compile_iframe / compile_pframe), with every weight tensor named by the RECIPE that builds it
 * from a raw checkpoint (lssvc_engine_load_checkpoint); the weights themselves come from the caller. The engine owns
 * the device memory, the side streams and the hipGraph it captures from the sequence (hipStreamBeginCapture inside the
 * library, after one eager pass). All tensors are fp32 NCHW, batch 1, as at the reference's model API; input / output
 * pointers may be device or host memory (hipMemcpyDefault); an output pointer may be NULL if the caller does not want it.
 * The caller owns the DPB between frames, exactly as with the reference: clamp the two reconstructions to [0, 1]
 * (test.py:249-250; lssvc_clamp_inplace) and hand them back with the two features as the next frame's references. */
void *lssvc_engine_create(int32_t device);
void lssvc_engine_destroy(void *engine);
/* The checkpoint of one model (0 = IntraSS, 1 = LSSVC / LSSVC_extend) as a table of named fp32 tensors: what torch.load +
 * load_state_dict give the reference (IntraSS.py:190-214, LSSVC_net.py:141-149; a leading 'module.' is dropped the same way).
 * The engine keeps its own copy; plans loaded afterwards take their weights from it (prepared on the device once per layer by
 * lssvc_prepare_weights and shared between the model's plans), so a plan file holds launches only and is independent of the
 * checkpoint -- except for what its compilation baked in: the frame size, and the kernel choice of the fp16 range audit
 * (entry "f32_layers_n" of lssvc_engine_plan_meta; 0 for every checkpoint whose activations stay inside fp16's range). Call
 * before lssvc_engine_load_intra / _inter / _stream of that model. */
int lssvc_engine_load_checkpoint(void *engine, int32_t model, const lssvc_tensor *tensors, int32_t n_tensors);
int lssvc_engine_load_intra(void *engine, const char *iframe_plan_path);
/* first_p: the plan of the first P-frame after an I-frame (no BL reference feature, 64-channel EL reference feature),
 * steady_p: every later P-frame of the GOP */
int lssvc_engine_load_inter(void *engine, const char *first_p_plan_path, const char *steady_p_plan_path);
/* set_scale_information (IntraSS.py:229-232, LSSVC_net.py:266-269): checks the loaded plans were compiled for this
 * scale and padded EL size; plans are size-specific */
int lssvc_engine_set_scale(void *engine, float scale, int32_t H_el_padded, int32_t W_el_padded);
/* bits[0] = bit_bl, bits[1] = bit_el (estimated, as the reference's result dict) */
int lssvc_engine_iframe(void *engine, const float *x_bl, const float *x_el, double bits[2], float *x_hat_bl, float *x_hat_el,
                        float *feature_el, void *stream);
/* ref_feature_bl NULL selects the first-P plan. Outputs = the reference's result['dpb'] (un-clamped) + mv_hat + warp_frame */
int lssvc_engine_pframe(void *engine, const float *x_bl, const float *x_el, const float *ref_frame_bl, const float *ref_frame_el,
                        const float *ref_feature_bl, const float *ref_feature_el, double bits[2], float *recon_bl, float *feature_bl,
                        float *recon_el, float *feature_el, float *mv_hat, float *warp_frame, void *stream);
/* Round 6 -- the look-ahead of LSSVC_extend.forward_one_frame(next_x_bl=...) (lssvc_amd/inter.py; reference loop test.py:182-250, the
 * layers LSSVC_net.py:445-528 / dmc_net.py:421-488) for a caller without Python. A P-frame as TWO plans per frame type
 * (plan_compiler.compile_pframe_layers): its base layer alone, and its enhancement layer given the base layer's results. The base layer
 * of a P-frame reads the previous frame's BASE layer only, so the engine codes BL(t+1) -- from next_x_bl -- on a second stream of its
 * own while EL(t) runs on the caller's, keeps it, and the next call codes its enhancement layer only. Contract: consecutive frames of
 * one sequence; the x_bl of a call is the tensor named as next_x_bl in the previous one (then x_bl / ref_frame_bl / ref_feature_bl
 * are not read again); next_x_bl NULL behind the last frame; the caller clamps the DPB's reference frames to [0, 1] between frames, as
 * test.py:249-250 does (the base-layer plans clamp their own copy, so a base layer coded ahead sees the same values). ref_feature_bl
 * NULL = the first P-frame after an I-frame (drops anything coded ahead); lssvc_engine_lookahead_reset does the same on demand (a seek).
 * Same launches per layer as lssvc_engine_pframe: results are bit-identical to it (tests/test_gpu_engine.py, tests/engine_demo.c). */
int lssvc_engine_load_inter_layers(void *engine, const char *bl_first_plan, const char *bl_steady_plan, const char *el_first_plan,
                                   const char *el_steady_plan);
int lssvc_engine_pframe_lookahead(void *engine, const float *x_bl, const float *x_el, const float *next_x_bl, const float *ref_frame_bl,
                                  const float *ref_frame_el, const float *ref_feature_bl, const float *ref_feature_el, double bits[2],
                                  float *recon_bl, float *feature_bl, float *recon_el, float *feature_el, float *mv_hat, float *warp_frame,
                                  void *stream);
int lssvc_engine_lookahead_reset(void *engine);
/* which: 0 intra, 1 first-P, 2 steady-P; 3 / 4 I-frame encoder / decoder, 5 / 6 first-P, 7 / 8 steady-P (write_stream = 1 plans)
 * -> launches (host steps included), streams, arena bytes, weight bytes, H, W */
int lssvc_engine_plan_info(void *engine, int32_t which, int64_t *out6);
/* One named integer of a plan's header: "pad_left" / "pad_right" / "pad_top" / "pad_bottom" (the inter-layer padding the plan
 * was compiled for: set_scale_information's pad_size, IntraSS.py:229-232), "f32_layers_n" / "f32_layers_crc" (the conv layers
 * the fp16 range audit moved to the exact fp32 kernel), "pic_height_bl" ... (stream plans). lssvc_engine_set_scale refuses a
 * set of plans that disagree on the padding or, within one model, on the fp32 layers. Unknown name = error. */
int lssvc_engine_plan_meta(void *engine, int32_t which, const char *name, int64_t *out);

/* write_stream = 1 through the engine (SURVEY 8b's lssvc_pframe_symbols / lssvc_pframe_decode): the ENCODER and the DECODER
 * half of a frame as separate plans (plan_compiler.py: compile_iframe_stream / compile_pframe_stream), replacing
 * IntraSS.compress / decompress (src/models/IntraSS.py:304-336, priors.py:422-452) and LSSVC_extend.compress / decompress
 * (src/models/LSSVC_net_extend.py:24-136, dmc_net_extend.py:55-146). Such a plan also holds the host coder's steps (staging
 * copies of the int16 planes, rANS calls, the CDF tables update() built) and is replayed eagerly. A layer "file" is the byte
 * image the reference writes to disk (src/utils/stream_helper.py:61-99): I-frame = big-endian u32 height, width, len_y,
 * len_z + the y and z strings; P-frame = u32 len + one string -- the bytes are those of the Python path's .bin files.
 * Any path may be NULL (an encoder-only or decoder-only process loads its half). */
int lssvc_engine_load_stream(void *engine, const char *iframe_enc_plan, const char *iframe_dec_plan, const char *first_p_enc_plan,
                             const char *first_p_dec_plan, const char *steady_p_enc_plan, const char *steady_p_dec_plan);
/* *_len receives the file size (also when the buffer is too small, which is an error). Outputs: the reconstruction the decoder
 * will produce from these bytes (bit-identical), i.e. the DPB of the next frame */
int lssvc_engine_encode_iframe(void *engine, const float *x_bl, const float *x_el, uint8_t *bl_file, int64_t bl_cap, int64_t *bl_len,
                               uint8_t *el_file, int64_t el_cap, int64_t *el_len, float *x_hat_bl, float *x_hat_el, float *feature_el,
                               void *stream);
int lssvc_engine_decode_iframe(void *engine, const uint8_t *bl_file, int64_t bl_len, const uint8_t *el_file, int64_t el_len,
                               float *x_hat_bl, float *x_hat_el, float *feature_el, void *stream);
/* ref_feature_bl NULL selects the first-P plans. recon_bl comes back clamped to [0, 1], as the reference's base-layer decoder
 * returns it (dmc_net_extend.py:138); the caller clamps recon_el before handing it back (test.py:249-250) */
int lssvc_engine_encode_pframe(void *engine, const float *x_bl, const float *x_el, const float *ref_frame_bl, const float *ref_frame_el,
                               const float *ref_feature_bl, const float *ref_feature_el, uint8_t *bl_file, int64_t bl_cap, int64_t *bl_len,
                               uint8_t *el_file, int64_t el_cap, int64_t *el_len, float *recon_bl, float *feature_bl, float *recon_el,
                               float *feature_el, void *stream);
int lssvc_engine_decode_pframe(void *engine, const uint8_t *bl_file, int64_t bl_len, const uint8_t *el_file, int64_t el_len,
                               const float *ref_frame_bl, const float *ref_frame_el, const float *ref_feature_bl, const float *ref_feature_el,
                               float *recon_bl, float *feature_bl, float *recon_el, float *feature_el, void *stream);

/* Runtime tuning switches (each also reads an environment variable at first use):
 *   "f16x3_persist"            1/0   use the persistent warp-specialised 3x3 kernel (LSSVC_F16X3_PERSIST)
 *   "f16x3_persist_min_tiles"  n     ... for convs with at least n output tiles (LSSVC_F16X3_PERSIST_MIN_TILES, 256)
 *   "f16x3_persist7"           1/0   the persistent warp-specialised kernel for 7x7 convs too (LSSVC_F16X3_PERSIST7)
 *   "dwpre_deep"               1/0   the fused 1x1 + depthwise kernel prefetches the next tile's inputs one whole tile ahead
 *                                    (a second register set) instead of only during its depthwise phase (LSSVC_DWPRE_DEEP)
 *   "pointwise_blocks"         1/0   x2 bilinear resize and depthwise 3x3 compute a 2x2 output block per thread (the input
 *                                    neighbourhood is loaded once: 9 / 16 loads instead of 16 / 36) (LSSVC_POINTWISE_BLOCKS)
 *   round 6 (conv3_f16x3p.hip, conv_mfma_kernel.h; profiles/r06_*_ab.txt hold the A/Bs):
 *   "p3_small"                 1/0   3x3 stride-1 convs with fewer than f16x3_persist_min_tiles tiles of 24x16 pixels run on the persistent
 *                                    kernel's small-tile instantiations (16x16 / 8x16 / 4x16 tiles, picked by a cost model); 2 / 3: with the
 *                                    register prefetch always / never (1: from 8 phases on) (LSSVC_P3_SMALL)
 *   "p3_narrow"                1/0   3x3 stride-1 convs with <= 16 output channels on the narrow-head instantiation (16x16 tiles, two
 *                                    workgroups per CU; the tiled kernel's epilogue when Cout % 4 != 0) (LSSVC_P3_NARROW)
 *   "p3_pf2"                   0..4  producers of the stride-2 and narrow-head instantiations: 0 one register set (round 5); 1 (default)
 *                                    split roles -- one producer wave owns the weight DMA, three stage the patch through two register
 *                                    sets -- up to five 16-channel phases per tile and the register prefetch from six on; 2 pair loads
 *                                    (measured slower; stride 2 only); 3 roles always; 4 register prefetch always (LSSVC_P3_PF2)
 *   "p7_narrow"                1/0   experiment: 7x7 persistent kernel with one 16-channel fragment for Cout <= 16 (slower) (LSSVC_P7_NARROW)
 *   "p3_force"                 n     experiments: force the small tiling MF * 16 + rows-per-wave (0 = the cost model) (LSSVC_P3_FORCE)
 *   "p3_big_pair"              0..4  producers of the big tilings: 0 (default) on the 32x16 tiling split roles for three-phase tiles (48 -> 48
 *                                    layers: +1 ... +3 %) and late loads for six-phase tiles without an input activation (96 -> 48: +3 ... +4 %),
 *                                    round 5's schedule elsewhere; 1 experiment: 24x16 with pair loads (slower); 2 split roles wherever built
 *                                    (24x16: slower at MF = 4); 3 neither anywhere; 4 late loads wherever built (LSSVC_P3_BIG_PAIR)
 *   "gdn_fast"                 1/0   GDN / IGDN epilogue as straight-line code where the views allow it (LSSVC_GDN_FAST_OPT)
 *   "resample_rows"            1/0   lssvc_resample2d evaluates the vertical pass once per source column and output row (a workgroup per
 *                                    256 outputs of a row, column sums in the LDS) instead of once per output (LSSVC_RESAMPLE_ROWS)
 * Results do not depend on them (the kernels they choose between are bit-identical); tests use them to pin that. */
int lssvc_set_option(const char *name, int32_t value);
int lssvc_get_option(const char *name, int32_t *value);

const char *lssvc_last_error(void);
int lssvc_version(void);

#ifdef __cplusplus
}
#endif
#endif /* LSSVC_HIP_H */
