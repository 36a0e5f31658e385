// entropy.hip -- quantisation + likelihood / bit-count kernels of LSSVC's entropy models (gfx950).
//
// Every kernel evaluates the per-element CDF difference in fp32 exactly as the reference's ATen op
// sequence does (-ffp-contract=off), then accumulates in fp64: lane -> wavefront (DPP shuffles) ->
// workgroup (LDS) -> one partial per workgroup in a workspace; a second single-workgroup pass sums
// the partials in index order. No float atomics: the result is bit-reproducible run to run, which
// the encoder/decoder symmetry of write_stream=1 relies on.
#include "common.h"

namespace lssvc {

constexpr float kLn2f = 0.6931471805599453f;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// blockDim = 256. Returns the workgroup total in thread 0.
__device__ __forceinline__ double block_sum(double v) {
    __shared__ double part[4];
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0) t = part[0] + part[1] + part[2] + part[3];
    return t;
}

__global__ void reduce_final_kernel(const double *__restrict__ partials, int n, double *__restrict__ out) {
    // fixed order: thread t sums partials t, t+256, ... then the block tree is fixed too
    double v = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) v += partials[i];
    const double t = block_sum(v);
    if (threadIdx.x == 0) out[0] = t;
}

static inline unsigned reduce_blocks(long long total) {
    long long b = (total + 255) / 256;
    if (b > kReduceMaxBlocks) b = kReduceMaxBlocks;
    if (b < 1) b = 1;
    return (unsigned)b;
}

// ---- Laplace -------------------------------------------------------------------------------------
// torch.distributions.Laplace(0, s).cdf(v) = 0.5 - 0.5 * sign(v) * expm1(-|v| / s)
__device__ __forceinline__ float laplace_cdf(float v, float s) {
    const float sg = (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f);
    return 0.5f - 0.5f * sg * expm1f(-fabsf(v) / s);
}
__device__ __forceinline__ float laplace_bits_of(float q, float sigma) {
    const float s = fminf(fmaxf(sigma, 1e-5f), 1e10f);
    const float probs = laplace_cdf(q + 0.5f, s) - laplace_cdf(q - 0.5f, s);
    float b = -1.0f * logf(probs + 1e-5f) / kLn2f;
    return fminf(fmaxf(b, 0.f), 50.f);
}

// mode 0: quantise y around mean, write y_q / y_hat (if given), price; mode 1: y already holds symbols
__global__ void laplace_kernel(V y, V mean, V sigma, V y_q, V y_hat, int mode, long long total, double *partials) {
    double acc = 0.0;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int c = (int)(idx % y.C);
        const size_t pix = (size_t)(idx / y.C);
        float q;
        if (mode == 0) {
            const float mu = mean.p[pix * mean.ld + c];
            q = rintf(y.p[pix * y.ld + c] - mu);
            if (y_q.p) y_q.p[pix * y_q.ld + c] = q;
            if (y_hat.p) y_hat.p[pix * y_hat.ld + c] = q + mu;
        } else {
            q = y.p[pix * y.ld + c];
        }
        acc += (double)laplace_bits_of(q, sigma.p[pix * sigma.ld + c]);
    }
    const double t = block_sum(acc);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

// ---- 4-step checkerboard -------------------------------------------------------------------------
__global__ void four_part_kernel(V y, V mean, V sigma, int m0, int m1, int m2, int m3, V y_q, V y_hat, V s_hat,
                                 long long total) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % y.C);
    const size_t pix = (size_t)(idx / y.C);
    const int x = (int)(pix % y.W), yy = (int)(pix / y.W);
    const int chunk = c / (y.C >> 2);
    const int m = chunk == 0 ? m0 : (chunk == 1 ? m1 : (chunk == 2 ? m2 : m3));
    if ((yy & 1) != (m >> 1) || (x & 1) != (m & 1)) return;
    const float mu = mean.p[pix * mean.ld + c];
    const float q = rintf(y.p[pix * y.ld + c] - mu);
    y_q.p[pix * y_q.ld + c] = q;
    y_hat.p[pix * y_hat.ld + c] = q + mu;
    s_hat.p[pix * s_hat.ld + c] = sigma.p[pix * sigma.ld + c];
}

// ---- BitEstimator (factorised prior of the P-frame hyper latents) --------------------------------
// params [11][C]: sp_h1,b1,ta1, sp_h2,b2,ta2, sp_h3,b3,ta3, sp_h4,b4
__device__ __forceinline__ float bit_estimator(float x, const float *__restrict__ P, int C, int c) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        x = x * P[(3 * i) * C + c] + P[(3 * i + 1) * C + c];
        x = x + tanhf(x) * P[(3 * i + 2) * C + c];
    }
    x = x * P[9 * C + c] + P[10 * C + c];
    return 1.f / (1.f + expf(-x));
}
__global__ void factorized_kernel(V z, const float *__restrict__ P, V z_hat, long long total, double *partials) {
    double acc = 0.0;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int c = (int)(idx % z.C);
        const size_t pix = (size_t)(idx / z.C);
        const float q = rintf(z.p[pix * z.ld + c]);
        if (z_hat.p) z_hat.p[pix * z_hat.ld + c] = q;
        const float prob = bit_estimator(q + 0.5f, P, z.C, c) - bit_estimator(q - 0.5f, P, z.C, c);
        float b = -1.0f * logf(prob + 1e-5f) / kLn2f;
        acc += (double)fminf(fmaxf(b, 0.f), 50.f);
    }
    const double t = block_sum(acc);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

// ---- GaussianConditional (I-frame y) --------------------------------------------------------------
__global__ void gaussian_kernel(V y, V scale, V mean, V y_hat, V y_q, long long total, double *partials) {
    const float kC = -0.70710678118654752440f;  // float(-(2 ** -0.5))
    double acc = 0.0;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int c = (int)(idx % y.C);
        const size_t pix = (size_t)(idx / y.C);
        const float mu = mean.p[pix * mean.ld + c];
        const float q = rintf(y.p[pix * y.ld + c] - mu);
        const float out = q + mu;
        if (y_hat.p) y_hat.p[pix * y_hat.ld + c] = out;
        if (y_q.p) y_q.p[pix * y_q.ld + c] = q;
        const float v = fabsf(out - mu);
        const float s = fmaxf(scale.p[pix * scale.ld + c], 0.11f);
        const float upper = 0.5f * erfcf(kC * ((0.5f - v) / s));
        const float lower = 0.5f * erfcf(kC * ((-0.5f - v) / s));
        const float lik = fmaxf(upper - lower, 1e-9f);
        acc += (double)logf(lik);
    }
    const double t = block_sum(acc);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

// ---- EntropyBottleneck (I-frame z) ----------------------------------------------------------------
// params [59][C]: m0[3] m1[3][3] m2[3][3] m3[3][3] m4[3] | b0[3] b1[3] b2[3] b3[3] b4[1] | f0[3] f1[3] f2[3] f3[3] | median
__device__ __forceinline__ float eb_logits(float v, const float *__restrict__ P, int C, int c) {
    auto R = [&](int row) { return P[row * C + c]; };
    float l[3], n[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float t = R(j) * v + R(33 + j);
        l[j] = t + R(46 + j) * tanhf(t);
    }
#pragma unroll
    for (int layer = 1; layer <= 3; ++layer) {
        const int mb = 3 + (layer - 1) * 9, bb = 33 + 3 * layer, fb = 46 + 3 * layer;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float t = R(mb + 3 * j) * l[0];
            t = fmaf(R(mb + 3 * j + 1), l[1], t);
            t = fmaf(R(mb + 3 * j + 2), l[2], t);
            t = t + R(bb + j);
            n[j] = t + R(fb + j) * tanhf(t);
        }
        l[0] = n[0]; l[1] = n[1]; l[2] = n[2];
    }
    float t = R(30) * l[0];
    t = fmaf(R(31), l[1], t);
    t = fmaf(R(32), l[2], t);
    return t + R(45);
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__global__ void bottleneck_kernel(V z, const float *__restrict__ P, V z_hat, V z_q, long long total, double *partials) {
    double acc = 0.0;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int c = (int)(idx % z.C);
        const size_t pix = (size_t)(idx / z.C);
        const float med = P[58 * z.C + c];
        const float zq = rintf(z.p[pix * z.ld + c] - med);
        const float out = zq + med;
        if (z_hat.p) z_hat.p[pix * z_hat.ld + c] = out;
        if (z_q.p) z_q.p[pix * z_q.ld + c] = zq;
        const float lower = eb_logits(out - 0.5f, P, z.C, c);
        const float upper = eb_logits(out + 0.5f, P, z.C, c);
        const float sum = lower + upper;
        const float sg = -((sum > 0.f) ? 1.f : ((sum < 0.f) ? -1.f : 0.f));
        const float lik = fmaxf(fabsf(sigmoidf_(sg * upper) - sigmoidf_(sg * lower)), 1e-9f);
        acc += (double)logf(lik);
    }
    const double t = block_sum(acc);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

// ---- sigma -> table index --------------------------------------------------------------------------
// sigma -> table index exactly as the reference computes it in fp32 (video_entropy_models.py:309-313):
// (log(max(s, 1e-5)) - log_min) / step + add, clamp, truncate. The logarithm is taken in fp64 and rounded once:
// that is the correctly rounded fp32 log, which ATen's CPU log returns for all but ~2e-4 of inputs, whereas the
// device logf (1 ulp) differs from it for ~10 % of inputs -- and a 1-ulp difference next to a level boundary
// changes the integer (the decoder of another implementation would then pick a different CDF).
__device__ __forceinline__ int32_t sigma_index(float sigma, float log_min, float log_step, float add, int levels) {
    const float s = fmaxf(sigma, 1e-5f);
    float v = ((float)log((double)s) - log_min) / log_step + add;
    v = fminf(fmaxf(v, 0.f), (float)(levels - 1));
    return (int32_t)v;
}

__global__ void build_indexes_kernel(V sigma, float log_min, float log_step, float add, int levels, int32_t *idx_out,
                                     long long total) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % sigma.C);
    const size_t pix = (size_t)(idx / sigma.C);
    idx_out[pix * sigma.C + c] = sigma_index(sigma.p[pix * sigma.ld + c], log_min, log_step, add, levels);
}


// ---- symbol / index planes for the host coder (write_stream = 1) -------------------------------------
// The host coder consumes flat NCHW-ordered int32 planes, exactly the order in which the reference flattens
// its tensors (x.reshape(-1) of an NCHW tensor, video_entropy_models.py:234-236,315-319).

// chunk_of_mask < 0: plain export of a C-channel tensor. Otherwise the 4-step fold (LSSVC_net.py:432-442):
// out channel j at 2x2 position m takes channel chunk_of_mask[m]*C4 + j of the C-channel inputs.
// PlaneT = int32_t (the reference's width) or int16_t (half the PCIe bytes; a symbol that does not fit sets *overflow, which
// the host checks before it codes the plane -- real latents are a few tens at most).
template <typename PlaneT>
__global__ void export_symbols_kernel(V q, V sigma, int cm0, int cm1, int cm2, int cm3, float log_min, float log_step, float add,
                                      int levels, PlaneT *sym, PlaneT *idx, int32_t *overflow, int C_out, int H, int W, long long total) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int j = (int)(t % C_out);
    const long long pix = t / C_out;
    int c = j;
    if (cm0 >= 0) {
        const int x = (int)(pix % W), y = (int)(pix / W);
        const int m = (y & 1) * 2 + (x & 1);
        c = (m == 0 ? cm0 : (m == 1 ? cm1 : (m == 2 ? cm2 : cm3))) * C_out + j;
    }
    const size_t o = (size_t)j * H * W + pix;
    if (sym) {
        const int32_t s = (int32_t)q.p[(size_t)pix * q.ld + c];
        if constexpr (sizeof(PlaneT) == 2) {
            if (s < -32768 || s > 32767) atomicOr(overflow, 1);
        }
        sym[o] = (PlaneT)s;
    }
    if (idx) idx[o] = (PlaneT)(sigma.p ? sigma_index(sigma.p[(size_t)pix * sigma.ld + c], log_min, log_step, add, levels) : j);
}

// out[pix][c] = sym + mean + add[c]; with chunk_of_mask >= 0 the 4-step unfold (LSSVC_net_extend.py:208-213):
// only channel chunk_of_mask[m]*C4 + j at position m is written.
template <typename PlaneT>
__global__ void import_symbols_kernel(const PlaneT *__restrict__ sym, V mean, const float *__restrict__ add, int cm0, int cm1,
                                      int cm2, int cm3, V out, int C_in, int H, int W, long long total) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int j = (int)(t % C_in);
    const long long pix = t / C_in;
    int c = j;
    if (cm0 >= 0) {
        const int x = (int)(pix % W), y = (int)(pix / W);
        const int m = (y & 1) * 2 + (x & 1);
        c = (m == 0 ? cm0 : (m == 1 ? cm1 : (m == 2 ? cm2 : cm3))) * C_in + j;
    }
    float v = (float)sym[(size_t)j * H * W + pix];
    if (mean.p) v += mean.p[(size_t)pix * mean.ld + c];
    if (add) v += add[c];
    out.p[(size_t)pix * out.ld + c] = v;
}

}  // namespace lssvc

using namespace lssvc;

extern "C" int64_t lssvc_reduce_workspace_bytes(void) { return (int64_t)kReduceMaxBlocks * sizeof(double); }

static int finish_reduce(unsigned blocks, void *workspace, double *out, hipStream_t st, const char *what) {
    hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(256), 0, st, (const double *)workspace, (int)blocks, out);
    return launch_status(what);
}
static V opt(const lssvc_view *v) { return (v && v->ptr) ? mk(v) : mk_null(); }

extern "C" int lssvc_laplace_quant_bits(const lssvc_view *y, const lssvc_view *mean, const lssvc_view *sigma,
                                        const lssvc_view *y_q, const lssvc_view *y_hat, double *bits_out, void *workspace,
                                        void *stream) {
    LSSVC_CHECK(view_ok(y) && view_ok(mean) && view_ok(sigma) && bits_out && workspace, "laplace_quant_bits: bad arguments");
    LSSVC_CHECK(same_shape(y, mean) && same_shape(y, sigma), "laplace_quant_bits: shape mismatch");
    LSSVC_CHECK((!y_q || !y_q->ptr || same_shape(y, y_q)) && (!y_hat || !y_hat->ptr || same_shape(y, y_hat)),
                "laplace_quant_bits: output shape mismatch");
    const long long total = (long long)y->H * y->W * y->C;
    const unsigned blocks = reduce_blocks(total);
    hipLaunchKernelGGL(laplace_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, mk(y), mk(mean), mk(sigma), opt(y_q),
                       opt(y_hat), 0, total, (double *)workspace);
    if (int e = launch_status("laplace_quant_bits")) return e;
    return finish_reduce(blocks, workspace, bits_out, (hipStream_t)stream, "laplace_quant_bits/reduce");
}

extern "C" int lssvc_laplace_bits(const lssvc_view *y_q, const lssvc_view *sigma, double *bits_out, void *workspace,
                                  void *stream) {
    LSSVC_CHECK(view_ok(y_q) && view_ok(sigma) && bits_out && workspace, "laplace_bits: bad arguments");
    LSSVC_CHECK(same_shape(y_q, sigma), "laplace_bits: shape mismatch");
    const long long total = (long long)y_q->H * y_q->W * y_q->C;
    const unsigned blocks = reduce_blocks(total);
    hipLaunchKernelGGL(laplace_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, mk(y_q), mk_null(), mk(sigma),
                       mk_null(), mk_null(), 1, total, (double *)workspace);
    if (int e = launch_status("laplace_bits")) return e;
    return finish_reduce(blocks, workspace, bits_out, (hipStream_t)stream, "laplace_bits/reduce");
}

extern "C" int lssvc_four_part_step(const lssvc_view *y, const lssvc_view *mean, const lssvc_view *sigma,
                                    const int32_t mask_of_chunk[4], const lssvc_view *y_q, const lssvc_view *y_hat,
                                    const lssvc_view *sigma_hat, void *stream) {
    LSSVC_CHECK(view_ok(y) && view_ok(mean) && view_ok(sigma) && view_ok(y_q) && view_ok(y_hat) && view_ok(sigma_hat) && mask_of_chunk,
                "four_part_step: bad arguments");
    LSSVC_CHECK(y->C % 4 == 0 && same_shape(y, mean) && same_shape(y, sigma) && same_shape(y, y_q) && same_shape(y, y_hat) &&
                    same_shape(y, sigma_hat), "four_part_step: shape mismatch");
    for (int i = 0; i < 4; ++i) LSSVC_CHECK(mask_of_chunk[i] >= 0 && mask_of_chunk[i] < 4, "four_part_step: mask %d", mask_of_chunk[i]);
    const long long total = (long long)y->H * y->W * y->C;
    hipLaunchKernelGGL(four_part_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mk(y), mk(mean),
                       mk(sigma), mask_of_chunk[0], mask_of_chunk[1], mask_of_chunk[2], mask_of_chunk[3], mk(y_q), mk(y_hat),
                       mk(sigma_hat), total);
    return launch_status("four_part_step");
}

extern "C" int lssvc_factorized_quant_bits(const lssvc_view *z, const float *params, const lssvc_view *z_hat, double *bits_out,
                                           void *workspace, void *stream) {
    LSSVC_CHECK(view_ok(z) && params && bits_out && workspace, "factorized_quant_bits: bad arguments");
    LSSVC_CHECK(!z_hat || !z_hat->ptr || same_shape(z, z_hat), "factorized_quant_bits: z_hat shape mismatch");
    const long long total = (long long)z->H * z->W * z->C;
    const unsigned blocks = reduce_blocks(total);
    hipLaunchKernelGGL(factorized_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, mk(z), params, opt(z_hat), total,
                       (double *)workspace);
    if (int e = launch_status("factorized_quant_bits")) return e;
    return finish_reduce(blocks, workspace, bits_out, (hipStream_t)stream, "factorized_quant_bits/reduce");
}

extern "C" int lssvc_gaussian_conditional(const lssvc_view *y, const lssvc_view *scale, const lssvc_view *mean,
                                          const lssvc_view *y_hat, const lssvc_view *y_q, double *sum_out, void *workspace,
                                          void *stream) {
    LSSVC_CHECK(view_ok(y) && view_ok(scale) && view_ok(mean) && sum_out && workspace, "gaussian_conditional: bad arguments");
    LSSVC_CHECK(same_shape(y, scale) && same_shape(y, mean) && (!y_hat || !y_hat->ptr || same_shape(y, y_hat)),
                "gaussian_conditional: shape mismatch");
    const long long total = (long long)y->H * y->W * y->C;
    const unsigned blocks = reduce_blocks(total);
    hipLaunchKernelGGL(gaussian_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, mk(y), mk(scale), mk(mean), opt(y_hat),
                       opt(y_q), total, (double *)workspace);
    if (int e = launch_status("gaussian_conditional")) return e;
    return finish_reduce(blocks, workspace, sum_out, (hipStream_t)stream, "gaussian_conditional/reduce");
}

extern "C" int lssvc_entropy_bottleneck(const lssvc_view *z, const float *params, const lssvc_view *z_hat, const lssvc_view *z_q,
                                        double *sum_out, void *workspace, void *stream) {
    LSSVC_CHECK(view_ok(z) && params && sum_out && workspace, "entropy_bottleneck: bad arguments");
    LSSVC_CHECK(!z_hat || !z_hat->ptr || same_shape(z, z_hat), "entropy_bottleneck: z_hat shape mismatch");
    const long long total = (long long)z->H * z->W * z->C;
    const unsigned blocks = reduce_blocks(total);
    hipLaunchKernelGGL(bottleneck_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, mk(z), params, opt(z_hat), opt(z_q), total,
                       (double *)workspace);
    if (int e = launch_status("entropy_bottleneck")) return e;
    return finish_reduce(blocks, workspace, sum_out, (hipStream_t)stream, "entropy_bottleneck/reduce");
}

extern "C" int lssvc_build_indexes(const lssvc_view *sigma, float log_min, float log_step, float add, int32_t levels,
                                   int32_t *idx_nhwc, void *stream) {
    LSSVC_CHECK(view_ok(sigma) && idx_nhwc && levels > 0 && log_step > 0.f, "build_indexes: bad arguments");
    const long long total = (long long)sigma->H * sigma->W * sigma->C;
    hipLaunchKernelGGL(build_indexes_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mk(sigma),
                       log_min, log_step, add, levels, idx_nhwc, total);
    return launch_status("build_indexes");
}


static int chunk_masks_ok(const int32_t *cm) {
    if (!cm) return 1;
    int seen = 0;
    for (int i = 0; i < 4; ++i) {
        if (cm[i] < 0 || cm[i] > 3) return 0;
        seen |= 1 << cm[i];
    }
    return seen == 15;
}

template <typename PlaneT>
static int export_symbols_impl(const lssvc_view *q, const lssvc_view *sigma, const int32_t *chunk_of_mask, float log_min,
                               float log_step, float add, int32_t levels, PlaneT *sym_nchw, PlaneT *idx_nchw, int32_t *overflow,
                               void *stream) {
    const lssvc_view *ref = (q && q->ptr) ? q : sigma;
    LSSVC_CHECK(view_ok(ref) && (sym_nchw || idx_nchw), "export_symbols: bad arguments");
    LSSVC_CHECK(!sym_nchw || view_ok(q), "export_symbols: symbols requested without q");
    LSSVC_CHECK(!(q && q->ptr && sigma && sigma->ptr) || same_shape(q, sigma), "export_symbols: q / sigma shape mismatch");
    LSSVC_CHECK(chunk_masks_ok(chunk_of_mask) && (!chunk_of_mask || ref->C % 4 == 0), "export_symbols: bad chunk_of_mask");
    LSSVC_CHECK(!(sigma && sigma->ptr) || (levels > 0 && log_step > 0.f), "export_symbols: bad index parameters");
    const int C_out = chunk_of_mask ? ref->C / 4 : ref->C;
    const long long total = (long long)ref->H * ref->W * C_out;
    const int32_t none[4] = {-1, -1, -1, -1};
    const int32_t *cm = chunk_of_mask ? chunk_of_mask : none;
    hipLaunchKernelGGL(export_symbols_kernel<PlaneT>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, opt(q),
                       opt(sigma), cm[0], cm[1], cm[2], cm[3], log_min, log_step, add, levels, sym_nchw, idx_nchw, overflow, C_out,
                       ref->H, ref->W, total);
    return launch_status("export_symbols");
}

extern "C" int lssvc_export_symbols(const lssvc_view *q, const lssvc_view *sigma, const int32_t *chunk_of_mask, float log_min,
                                    float log_step, float add, int32_t levels, int32_t *sym_nchw, int32_t *idx_nchw, void *stream) {
    return export_symbols_impl<int32_t>(q, sigma, chunk_of_mask, log_min, log_step, add, levels, sym_nchw, idx_nchw, nullptr, stream);
}

extern "C" int lssvc_export_symbols_i16(const lssvc_view *q, const lssvc_view *sigma, const int32_t *chunk_of_mask, float log_min,
                                        float log_step, float add, int32_t levels, int16_t *sym_nchw, int16_t *idx_nchw,
                                        int32_t *overflow, void *stream) {
    LSSVC_CHECK(overflow || !sym_nchw, "export_symbols_i16: symbols requested without an overflow flag");
    LSSVC_CHECK(levels <= 32768, "export_symbols_i16: %d table levels do not fit 16 bits", levels);
    return export_symbols_impl<int16_t>(q, sigma, chunk_of_mask, log_min, log_step, add, levels, sym_nchw, idx_nchw, overflow, stream);
}

template <typename PlaneT>
static int import_symbols_impl(const PlaneT *sym_nchw, const lssvc_view *mean, const float *channel_add,
                               const int32_t *chunk_of_mask, const lssvc_view *out, void *stream) {
    LSSVC_CHECK(sym_nchw && view_ok(out), "import_symbols: bad arguments");
    LSSVC_CHECK(!(mean && mean->ptr) || same_shape(mean, out), "import_symbols: mean shape mismatch");
    LSSVC_CHECK(chunk_masks_ok(chunk_of_mask) && (!chunk_of_mask || out->C % 4 == 0), "import_symbols: bad chunk_of_mask");
    const int C_in = chunk_of_mask ? out->C / 4 : out->C;
    const long long total = (long long)out->H * out->W * C_in;
    const int32_t none[4] = {-1, -1, -1, -1};
    const int32_t *cm = chunk_of_mask ? chunk_of_mask : none;
    hipLaunchKernelGGL(import_symbols_kernel<PlaneT>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, sym_nchw,
                       opt(mean), channel_add, cm[0], cm[1], cm[2], cm[3], mk(out), C_in, out->H, out->W, total);
    return launch_status("import_symbols");
}

extern "C" int lssvc_import_symbols(const int32_t *sym_nchw, const lssvc_view *mean, const float *channel_add,
                                    const int32_t *chunk_of_mask, const lssvc_view *out, void *stream) {
    return import_symbols_impl<int32_t>(sym_nchw, mean, channel_add, chunk_of_mask, out, stream);
}

extern "C" int lssvc_import_symbols_i16(const int16_t *sym_nchw, const lssvc_view *mean, const float *channel_add,
                                        const int32_t *chunk_of_mask, const lssvc_view *out, void *stream) {
    return import_symbols_impl<int16_t>(sym_nchw, mean, channel_add, chunk_of_mask, out, stream);
}

// ---- error plumbing ---------------------------------------------------------------------------------
namespace lssvc {
char *err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}
int fail(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return 1;
}
}  // namespace lssvc
extern "C" const char *lssvc_last_error(void) { return lssvc::err_buf(); }
extern "C" int lssvc_version(void) { return 1; }
