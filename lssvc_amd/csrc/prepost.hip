// prepost.hip -- the caller's side of the frame loop as HIP kernels: what test.py does to every
// frame before and after encode_decode (test.py:185-201,249-311) and the metric arithmetic of src/utils/functional.py,
// src/utils/core.py:364-432. All HBM-bound, one pass each:
//   lssvc_yuv420_to_frame   8-bit planar 4:2:0 -> RGB fp32 NHWC frame, zero-padded to the inter-layer size
//                            (ycbcr420_to_rgb functional.py:42-58: chroma x2 by linear interpolation at scipy.ndimage.zoom's
//                            sample positions, BT.709, clip) + the normalised source planes the per-plane PSNRs use
//   lssvc_rgb8_to_frame     8-bit planar RGB -> fp32 NHWC frame, zero-padded (x / 255)
//   lssvc_resample2d        separable K-tap resampling with host-built tap tables, vertical pass then horizontal pass as
//                            core.py:276-345 orders them: the MATLAB-bicubic base-layer frame (imresize, antialiased)
//   lssvc_rgb_to_yuv420     rgb_to_ycbcr420 (functional.py:16-39) of a (cropped) frame -> y, u, v planes
//   lssvc_sqdiff_sum        sum (a - b)^2 in fp64, fixed order (PSNR = 10 log10(1 / mean), test.py:104-118), optional
//                            clamp of `a` to [0, 1] and crop; frame views or flat planes
#include "common.h"

namespace lssvc {

constexpr float KR = 0.2126f, KG = 0.7152f, KB = 0.0722f;      // ITU-R BT.709 (functional.py:10-13)

__device__ __forceinline__ double pp_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ double pp_block_sum(double v) {
    __shared__ double part[4];
    v = pp_wave_sum(v);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0) t = part[0] + part[1] + part[2] + part[3];
    return t;
}
__global__ void pp_reduce_final_kernel(const double *__restrict__ partials, int n, double *__restrict__ out) {
    double v = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) v += partials[i];
    const double t = pp_block_sum(v);
    if (threadIdx.x == 0) out[0] = t;
}

// align_corners=True linear interpolation of a half-size chroma plane at full-size position i:
// src = i * (n_in - 1) / (n_out - 1), computed in fp32 like ATen's area_pixel_compute_source_index
__device__ __forceinline__ void ac_index(int i, int n_in, float scale, int &i0, int &i1, float &l0, float &l1) {
    const float src = scale * (float)i;
    i0 = (int)src;
    if (i0 > n_in - 1) i0 = n_in - 1;
    i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
    l1 = src - (float)i0;
    l0 = 1.0f - l1;
}

__global__ void yuv420_to_frame_kernel(const uint8_t *__restrict__ yp, const uint8_t *__restrict__ up, const uint8_t *__restrict__ vp,
                                       int H, int W, V out, float *__restrict__ y_n, float *__restrict__ u_n, float *__restrict__ v_n,
                                       float sy, float sx, long long total) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int x = (int)(idx % out.W), y = (int)(idx / out.W);
    float *dst = out.p + (size_t)idx * out.ld;
    if (y >= H || x >= W) {                                      // inter-layer padding: zeros (test.py:191-193)
        dst[0] = 0.f; dst[1] = 0.f; dst[2] = 0.f;
        return;
    }
    const int h2 = H >> 1, w2 = W >> 1;
    const float yt = (float)yp[(size_t)y * W + x] / 255.0f;
    int y0, y1, x0, x1;
    float hy0, hy1, wx0, wx1;
    ac_index(y, h2, sy, y0, y1, hy0, hy1);
    ac_index(x, w2, sx, x0, x1, wx0, wx1);
    auto chroma = [&](const uint8_t *p) {
        const float a = (float)p[(size_t)y0 * w2 + x0] / 255.0f, b = (float)p[(size_t)y0 * w2 + x1] / 255.0f;
        const float c = (float)p[(size_t)y1 * w2 + x0] / 255.0f, d = (float)p[(size_t)y1 * w2 + x1] / 255.0f;
        return hy0 * (wx0 * a + wx1 * b) + hy1 * (wx0 * c + wx1 * d);      // upsample_bilinear2d's form
    };
    const float cb = chroma(up), cr = chroma(vp);
    const float r = yt + (2.0f - 2.0f * KR) * (cr - 0.5f);
    const float b = yt + (2.0f - 2.0f * KB) * (cb - 0.5f);
    const float g = (yt - KR * r - KB * b) / KG;
    dst[0] = fminf(fmaxf(r, 0.f), 1.f);
    dst[1] = fminf(fmaxf(g, 0.f), 1.f);
    dst[2] = fminf(fmaxf(b, 0.f), 1.f);
    if (y_n) y_n[(size_t)y * W + x] = yt;
    if (u_n && !(y & 1) && !(x & 1)) {
        const size_t o = (size_t)(y >> 1) * w2 + (x >> 1);
        u_n[o] = (float)up[o] / 255.0f;
        v_n[o] = (float)vp[o] / 255.0f;
    }
}

__global__ void rgb8_to_frame_kernel(const uint8_t *__restrict__ rgb, int H, int W, V out, long long total) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int x = (int)(idx % out.W), y = (int)(idx / out.W);
    float *dst = out.p + (size_t)idx * out.ld;
    if (y >= H || x >= W) {
        dst[0] = 0.f; dst[1] = 0.f; dst[2] = 0.f;
        return;
    }
    const size_t o = (size_t)y * W + x, plane = (size_t)H * W;
    dst[0] = (float)rgb[o] / 255.0f;
    dst[1] = (float)rgb[plane + o] / 255.0f;
    dst[2] = (float)rgb[2 * plane + o] / 255.0f;
}

// out[y][x][c] = clamp( sum_kx wh[x][kx] * ( sum_ky wv[y][ky] * in[iv[y][ky]][ih[x][kx]][c] ) ): the vertical pass is
// evaluated per horizontal tap, taps summed in index order; tables are [n_out][K].
__global__ void resample2d_kernel(V in, V out, const float *__restrict__ wv, const int32_t *__restrict__ iv, int Kv,
                                  const float *__restrict__ wh, const int32_t *__restrict__ ih, int Kh, float lo, float hi,
                                  long long total) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % out.C);
    const long long pix = idx / out.C;
    const int x = (int)(pix % out.W), y = (int)(pix / out.W);
    const float *wvy = wv + (size_t)y * Kv;
    const int32_t *ivy = iv + (size_t)y * Kv;
    float acc = 0.f;
    for (int kx = 0; kx < Kh; ++kx) {
        const int sx = ih[(size_t)x * Kh + kx];
        float col = 0.f;
        for (int ky = 0; ky < Kv; ++ky) col += wvy[ky] * in.p[((size_t)ivy[ky] * in.W + sx) * in.ld + c];
        acc += wh[(size_t)x * Kh + kx] * col;
    }
    out.p[(size_t)pix * out.ld + c] = fminf(fmaxf(acc, lo), hi);
}

// The same sums, separably (round 6): a workgroup owns kRsTx output pixels of one output row; it first evaluates the VERTICAL pass once per
// source column its outputs touch (col[sx][c] = sum_ky wv[y][ky] * in[iv[y][ky]][sx][c], taps in index order, coalesced along sx and c) into
// the LDS, then every output sums its horizontal taps over those columns in index order -- the operations of resample2d_kernel on every
// element in the same order (bit-identical: tests/test_gpu_prepost.py), an eighth of its loads at 1080p -> 540p (237 -> ~25 us; the
// bench's per-frame pre-processing sits in the clock since round 6). Source-column ranges that do not fit the LDS budget (a down-scale
// beyond ~x15) are evaluated per output as before.
constexpr int kRsTx = 256, kRsCap = 12288;      // outputs per workgroup; floats of LDS for the column sums (48 KB)
__global__ void resample2d_rows_kernel(V in, V out, const float *__restrict__ wv, const int32_t *__restrict__ iv, int Kv,
                                       const float *__restrict__ wh, const int32_t *__restrict__ ih, int Kh, float lo, float hi) {
    extern __shared__ float rs_cols[];
    __shared__ int s_min, s_max;
    const int y = blockIdx.y, x0 = blockIdx.x * kRsTx, C = out.C;
    const int nx = out.W - x0 < kRsTx ? out.W - x0 : kRsTx;
    if (threadIdx.x == 0) {
        s_min = 0x7fffffff;
        s_max = -1;
    }
    __syncthreads();
    int mn = 0x7fffffff, mx = -1;
    for (int i = threadIdx.x; i < nx * Kh; i += blockDim.x) {
        const int sx = ih[(size_t)x0 * Kh + i];
        mn = sx < mn ? sx : mn;
        mx = sx > mx ? sx : mx;
    }
    atomicMin(&s_min, mn);
    atomicMax(&s_max, mx);
    __syncthreads();
    const int s0 = s_min, n = s_max - s0 + 1;
    const float *wvy = wv + (size_t)y * Kv;
    const int32_t *ivy = iv + (size_t)y * Kv;
    const bool staged = (long long)n * C <= kRsCap;
    if (staged) {
        for (int i = threadIdx.x; i < n * C; i += blockDim.x) {
            const int sx = s0 + i / C, c = i % C;
            float col = 0.f;
            for (int ky = 0; ky < Kv; ++ky) col += wvy[ky] * in.p[((size_t)ivy[ky] * in.W + sx) * in.ld + c];
            rs_cols[i] = col;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nx * C; i += blockDim.x) {
        const int x = x0 + i / C, c = i % C;
        float acc = 0.f;
        for (int kx = 0; kx < Kh; ++kx) {
            const int sx = ih[(size_t)x * Kh + kx];
            float col;
            if (staged) {
                col = rs_cols[(sx - s0) * C + c];
            } else {
                col = 0.f;
                for (int ky = 0; ky < Kv; ++ky) col += wvy[ky] * in.p[((size_t)ivy[ky] * in.W + sx) * in.ld + c];
            }
            acc += wh[(size_t)x * Kh + kx] * col;
        }
        out.p[((size_t)y * out.W + x) * out.ld + c] = fminf(fmaxf(acc, lo), hi);
    }
}

// rgb_to_ycbcr420 of the top-left h x w crop (h, w even): one thread per 2x2 block
__global__ void rgb_to_yuv420_kernel(V rgb, int h, int w, int clamp01, float *__restrict__ yo, float *__restrict__ uo,
                                     float *__restrict__ vo, long long total) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int w2 = w >> 1;
    const int bx = (int)(idx % w2), by = (int)(idx / w2);
    float cbs[4], crs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int y = 2 * by + (j >> 1), x = 2 * bx + (j & 1);
        const float *s = rgb.p + ((size_t)y * rgb.W + x) * rgb.ld;
        float r = s[0], g = s[1], b = s[2];
        if (clamp01) {
            r = fminf(fmaxf(r, 0.f), 1.f); g = fminf(fmaxf(g, 0.f), 1.f); b = fminf(fmaxf(b, 0.f), 1.f);
        }
        const float yy = KR * r + KG * g + KB * b;
        cbs[j] = 0.5f * (b - yy) / (1.0f - KB) + 0.5f;
        crs[j] = 0.5f * (r - yy) / (1.0f - KR) + 0.5f;
        yo[(size_t)y * w + x] = fminf(fmaxf(yy, 0.f), 1.f);
    }
    // mean over the 2x2 block in the order torch's mean(dim=(1,3)) of the (h/2,2,w/2,2) view visits it
    const float cb = (cbs[0] + cbs[1] + cbs[2] + cbs[3]) / 4.0f, cr = (crs[0] + crs[1] + crs[2] + crs[3]) / 4.0f;
    uo[(size_t)by * w2 + bx] = fminf(fmaxf(cb, 0.f), 1.f);
    vo[(size_t)by * w2 + bx] = fminf(fmaxf(cr, 0.f), 1.f);
}

// sum over the h x w crop and all channels of (clamp?(a) - b)^2
__global__ void sqdiff_view_kernel(V a, V b, int h, int w, int clamp01, long long total, double *partials) {
    double acc = 0.0;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int c = (int)(idx % a.C);
        const long long pix = idx / a.C;
        const int x = (int)(pix % w), y = (int)(pix / w);
        float va = a.p[((size_t)y * a.W + x) * a.ld + c];
        if (clamp01) va = fminf(fmaxf(va, 0.f), 1.f);
        const float d = va - b.p[((size_t)y * b.W + x) * b.ld + c];
        acc += (double)(d * d);
    }
    const double t = pp_block_sum(acc);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}
__global__ void sqdiff_flat_kernel(const float *__restrict__ a, const float *__restrict__ b, long long total, double *partials) {
    double acc = 0.0;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const float d = a[idx] - b[idx];
        acc += (double)(d * d);
    }
    const double t = pp_block_sum(acc);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

static inline unsigned pp_blocks(long long total) {
    long long b = (total + 255) / 256;
    return (unsigned)(b < 1 ? 1 : b);
}
static inline unsigned pp_reduce_blocks(long long total) {
    long long b = (total + 255) / 256;
    if (b > kReduceMaxBlocks) b = kReduceMaxBlocks;
    return (unsigned)(b < 1 ? 1 : b);
}

}  // namespace lssvc

using namespace lssvc;

extern "C" int lssvc_yuv420_to_frame(const uint8_t *y, const uint8_t *u, const uint8_t *v, int32_t H, int32_t W,
                                     const lssvc_view *frame, float *y_norm, float *u_norm, float *v_norm, void *stream) {
    LSSVC_CHECK(y && u && v && view_ok(frame) && frame->C == 3, "yuv420_to_frame: bad arguments");
    LSSVC_CHECK(H >= 4 && W >= 4 && !(H & 1) && !(W & 1) && frame->H >= H && frame->W >= W, "yuv420_to_frame: %dx%d into %dx%d", H, W,
                frame->H, frame->W);
    LSSVC_CHECK((u_norm == nullptr) == (v_norm == nullptr), "yuv420_to_frame: u_norm / v_norm go together");
    const long long total = (long long)frame->H * frame->W;
    const float sy = (float)(H / 2 - 1) / (float)(H - 1), sx = (float)(W / 2 - 1) / (float)(W - 1);
    hipLaunchKernelGGL(yuv420_to_frame_kernel, dim3(pp_blocks(total)), dim3(256), 0, (hipStream_t)stream, y, u, v, H, W, mk(frame),
                       y_norm, u_norm, v_norm, sy, sx, total);
    return launch_status("yuv420_to_frame");
}

extern "C" int lssvc_rgb8_to_frame(const uint8_t *rgb, int32_t H, int32_t W, const lssvc_view *frame, void *stream) {
    LSSVC_CHECK(rgb && view_ok(frame) && frame->C == 3 && H > 0 && W > 0 && frame->H >= H && frame->W >= W,
                "rgb8_to_frame: bad arguments");
    const long long total = (long long)frame->H * frame->W;
    hipLaunchKernelGGL(rgb8_to_frame_kernel, dim3(pp_blocks(total)), dim3(256), 0, (hipStream_t)stream, rgb, H, W, mk(frame), total);
    return launch_status("rgb8_to_frame");
}

extern "C" int lssvc_resample2d(const lssvc_view *in, const lssvc_view *out, const float *w_v, const int32_t *idx_v, int32_t k_v,
                                const float *w_h, const int32_t *idx_h, int32_t k_h, float clamp_lo, float clamp_hi, void *stream) {
    LSSVC_CHECK(view_ok(in) && view_ok(out) && in->C == out->C && w_v && idx_v && w_h && idx_h && k_v > 0 && k_h > 0,
                "resample2d: bad arguments");
    const long long total = (long long)out->H * out->W * out->C;
    if (option_get(OPT_RESAMPLE_ROWS) && out->H <= 65535 && (long long)kRsTx * out->C <= 65536) {
        hipLaunchKernelGGL(resample2d_rows_kernel, dim3((out->W + kRsTx - 1) / kRsTx, out->H), dim3(256), kRsCap * sizeof(float), (hipStream_t)stream,
                           mk(in), mk(out), w_v, idx_v, k_v, w_h, idx_h, k_h, clamp_lo, clamp_hi);
        return launch_status("resample2d(rows)");
    }
    hipLaunchKernelGGL(resample2d_kernel, dim3(pp_blocks(total)), dim3(256), 0, (hipStream_t)stream, mk(in), mk(out), w_v, idx_v, k_v,
                       w_h, idx_h, k_h, clamp_lo, clamp_hi, total);
    return launch_status("resample2d");
}

extern "C" int lssvc_rgb_to_yuv420(const lssvc_view *rgb, int32_t h, int32_t w, int32_t clamp01, float *y, float *u, float *v,
                                   void *stream) {
    LSSVC_CHECK(view_ok(rgb) && rgb->C == 3 && y && u && v && h > 0 && w > 0 && !(h & 1) && !(w & 1) && h <= rgb->H && w <= rgb->W,
                "rgb_to_yuv420: bad arguments");
    const long long total = (long long)(h / 2) * (w / 2);
    hipLaunchKernelGGL(rgb_to_yuv420_kernel, dim3(pp_blocks(total)), dim3(256), 0, (hipStream_t)stream, mk(rgb), h, w, clamp01, y, u, v,
                       total);
    return launch_status("rgb_to_yuv420");
}

extern "C" int lssvc_sqdiff_sum(const lssvc_view *a, const lssvc_view *b, int32_t h, int32_t w, int32_t clamp01, double *out,
                                void *workspace, void *stream) {
    LSSVC_CHECK(view_ok(a) && view_ok(b) && a->C == b->C && out && workspace, "sqdiff_sum: bad arguments");
    LSSVC_CHECK(h > 0 && w > 0 && h <= a->H && w <= a->W && h <= b->H && w <= b->W, "sqdiff_sum: crop %dx%d of %dx%d / %dx%d", h, w,
                a->H, a->W, b->H, b->W);
    const long long total = (long long)h * w * a->C;
    const unsigned blocks = pp_reduce_blocks(total);
    hipLaunchKernelGGL(sqdiff_view_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, mk(a), mk(b), h, w, clamp01, total,
                       (double *)workspace);
    if (int e = launch_status("sqdiff_sum")) return e;
    hipLaunchKernelGGL(pp_reduce_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double *)workspace, (int)blocks, out);
    return launch_status("sqdiff_sum/reduce");
}

extern "C" int lssvc_sqdiff_sum_flat(const float *a, const float *b, int64_t n, double *out, void *workspace, void *stream) {
    LSSVC_CHECK(a && b && n > 0 && out && workspace, "sqdiff_sum_flat: bad arguments");
    const unsigned blocks = pp_reduce_blocks(n);
    hipLaunchKernelGGL(sqdiff_flat_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, b, (long long)n, (double *)workspace);
    if (int e = launch_status("sqdiff_sum_flat")) return e;
    hipLaunchKernelGGL(pp_reduce_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double *)workspace, (int)blocks, out);
    return launch_status("sqdiff_sum_flat/reduce");
}
