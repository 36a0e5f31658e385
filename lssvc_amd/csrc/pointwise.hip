// pointwise.hip -- the HBM-bound kernels of the LSSVC hot path (gfx950): depthwise 3x3, bilinear
// resize, flow warp, 2x2 pooling, softmax-2 blend, adds/copies, the fused OffsetDiversity tail and
// the NCHW<->NHWC boundary transposes. All work on NHWC views; one thread handles 4 consecutive
// channels of one pixel (16-byte accesses) whenever the views allow it, so a wave reads/writes
// contiguous 1-KiB runs when C >= 64 and whole pixels otherwise.
//
// Compiled with -ffp-contract=off: the reference evaluates these as separate ATen ops (mul, add, ...),
// so no FMA contraction is allowed if results are to track the fp32 CPU oracle.
#include <cstdlib>
#include "common.h"

namespace lssvc {

__device__ __forceinline__ float4 ld4(const V &v, size_t pix, int c, bool vec) {
    const float *s = v.p + pix * v.ld + c;
    if (vec) return *reinterpret_cast<const float4 *>(s);
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    r.x = s[0];
    if (c + 1 < v.C) r.y = s[1];
    if (c + 2 < v.C) r.z = s[2];
    if (c + 3 < v.C) r.w = s[3];
    return r;
}
__device__ __forceinline__ void st4(const V &v, size_t pix, int c, bool vec, float4 r) {
    float *d = v.p + pix * v.ld + c;
    if (vec) {
        *reinterpret_cast<float4 *>(d) = r;
        return;
    }
    d[0] = r.x;
    if (c + 1 < v.C) d[1] = r.y;
    if (c + 2 < v.C) d[2] = r.z;
    if (c + 3 < v.C) d[3] = r.w;
}

// Generic launch geometry: total = H*W*ceil(C/4) items, 256 threads per block.
struct Items {
    long long total;
    int cg;  // channel groups of 4
};
static inline Items items_of(const lssvc_view *v) {
    Items it;
    it.cg = (v->C + 3) / 4;
    it.total = (long long)v->H * v->W * it.cg;
    return it;
}
// The kernels below decode (pixel, channel group) from a 32-bit thread index: four 64-bit div/mod per thread were ~300
// instructions, more than the rest of most of these kernels. Every launcher checks its item count with LSSVC_ITEMS_OK.
static inline unsigned blocks_for(long long total) { return (unsigned)((total + 255) / 256); }
#define LSSVC_ITEMS_OK(total, what) LSSVC_CHECK((total) > 0 && (total) < (1LL << 31), what ": %lld items do not fit 32-bit indexing", (long long)(total))

// ------------------------------------------------------------------------------------------------
// depthwise 3x3, stride 1, zero pad 1; weight [9][C]
__global__ void dwconv3x3_kernel(V in, const float *__restrict__ w, const float *__restrict__ bias, V out, int cg,
                                 long long total) {
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;      // 32-bit index arithmetic (the host refuses totals >= 2^31)
    if (idx >= (unsigned)total) return;
    const unsigned pix = idx / (unsigned)cg;
    const int g = (int)(idx - pix * (unsigned)cg);
    const int y = (int)(pix / (unsigned)in.W), x = (int)(pix - (unsigned)y * (unsigned)in.W);
    const int c = g * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int yy = y + ky - 1;
        if (yy < 0 || yy >= in.H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int xx = x + kx - 1;
            if (xx < 0 || xx >= in.W) continue;
            const float4 v = *reinterpret_cast<const float4 *>(in.p + ((size_t)yy * in.W + xx) * in.ld + c);
            const float4 k = *reinterpret_cast<const float4 *>(w + (ky * 3 + kx) * in.C + c);
            acc.x = fmaf(v.x, k.x, acc.x);
            acc.y = fmaf(v.y, k.y, acc.y);
            acc.z = fmaf(v.z, k.z, acc.z);
            acc.w = fmaf(v.w, k.w, acc.w);
        }
    }
    const float4 b = *reinterpret_cast<const float4 *>(bias + c);
    acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w;
    *reinterpret_cast<float4 *>(out.p + (size_t)pix * out.ld + c) = acc;
}

// The same depthwise conv with a 2x2 output block per thread: the 4x4 input neighbourhood is loaded once (4 loads per output
// instead of 9 -- the kernel is bound by those L2 reads). Every output accumulates its own nine taps in the generic kernel's
// order and skips the same out-of-image taps: bit-identical results.
__global__ void dwconv3x3_b2_kernel(V in, const float *__restrict__ w, const float *__restrict__ bias, V out, int cg, int bw,
                                    long long total) {
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= (unsigned)total) return;
    const unsigned blk = idx / (unsigned)cg;
    const int g = (int)(idx - blk * (unsigned)cg);
    const int by = (int)(blk / (unsigned)bw), bx = (int)(blk - (unsigned)by * (unsigned)bw);
    const int y0 = 2 * by, x0 = 2 * bx, c = g * 4;
    float4 v[4][4];
    bool okr[4], okc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        okr[a] = y0 - 1 + a >= 0 && y0 - 1 + a < in.H;
        okc[a] = x0 - 1 + a >= 0 && x0 - 1 + a < in.W;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const bool ok = okr[a] && okc[b];
            const size_t off = ok ? ((size_t)(y0 - 1 + a) * in.W + (x0 - 1 + b)) * in.ld + c : (size_t)c;
            v[a][b] = *reinterpret_cast<const float4 *>(in.p + off);
        }
    float4 k[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) k[t] = *reinterpret_cast<const float4 *>(w + t * in.C + c);
    const float4 bv = *reinterpret_cast<const float4 *>(bias + c);
    // all four results first, then the stores back to back: with compute + store per output inside its own bounds check the
    // compiler repeated the (already satisfied) load waits in every block, down to vmcnt(0) -- which, after the first store,
    // waits for THAT STORE: four store round trips in a row per thread
    float4 res[2][2];
#pragma unroll
    for (int oy = 0; oy < 2; ++oy)
#pragma unroll
        for (int ox = 0; ox < 2; ++ox) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    if (!(okr[oy + ky] && okc[ox + kx])) continue;
                    const float4 &p = v[oy + ky][ox + kx];
                    const float4 &q = k[ky * 3 + kx];
                    acc.x = fmaf(p.x, q.x, acc.x);
                    acc.y = fmaf(p.y, q.y, acc.y);
                    acc.z = fmaf(p.z, q.z, acc.z);
                    acc.w = fmaf(p.w, q.w, acc.w);
                }
            acc.x += bv.x; acc.y += bv.y; acc.z += bv.z; acc.w += bv.w;
            asm volatile("" : "+v"(acc.x), "+v"(acc.y), "+v"(acc.z), "+v"(acc.w));      // (or the compiler sinks the arithmetic back into the store's block)
            res[oy][ox] = acc;
        }
#pragma unroll
    for (int oy = 0; oy < 2; ++oy)
#pragma unroll
        for (int ox = 0; ox < 2; ++ox)
            if (y0 + oy < in.H && x0 + ox < in.W)
                *reinterpret_cast<float4 *>(out.p + ((size_t)(y0 + oy) * out.W + x0 + ox) * out.ld + c) = res[oy][ox];
}

// ------------------------------------------------------------------------------------------------
// bilinear resize, align_corners=False (ATen area_pixel_compute_source_index + guard_index_and_lambda)
__device__ __forceinline__ void src_index(float scale, int dst, int size, int &i0, int &i1, float &l0, float &l1) {
    // ATen's CPU kernel is built with FMA contraction: scale * (dst + 0.5) - 0.5 is one fused op there.
    float real = fmaf(scale, (float)dst + 0.5f, -0.5f);
    if (real < 0.f) real = 0.f;
    int idx = (int)real;
    if (idx > size - 1) idx = size - 1;
    float lam = real - (float)idx;
    lam = fminf(fmaxf(lam, 0.f), 1.f);
    i0 = idx;
    i1 = idx + (idx < size - 1 ? 1 : 0);
    l1 = lam;
    l0 = 1.f - lam;
}

__global__ void resize_bilinear_kernel(V in, V out, float sy, float sx, float post, int cg, long long total, int vin,
                                       int vout) {
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;      // 32-bit index arithmetic (the host refuses totals >= 2^31)
    if (idx >= (unsigned)total) return;
    const unsigned pix = idx / (unsigned)cg;
    const int g = (int)(idx - pix * (unsigned)cg);
    const int y = (int)(pix / (unsigned)out.W), x = (int)(pix - (unsigned)y * (unsigned)out.W);
    const int c = g * 4;
    int y0, y1, x0, x1;
    float hy0, hy1, wx0, wx1;
    src_index(sy, y, in.H, y0, y1, hy0, hy1);
    src_index(sx, x, in.W, x0, x1, wx0, wx1);
    const float4 a = ld4(in, (size_t)y0 * in.W + x0, c, vin);
    const float4 b = ld4(in, (size_t)y0 * in.W + x1, c, vin);
    const float4 d = ld4(in, (size_t)y1 * in.W + x0, c, vin);
    const float4 e = ld4(in, (size_t)y1 * in.W + x1, c, vin);
    // separable form of ATen's upsample_generic kernel: t = a*w0; t += b*w1 (contracted to an FMA on the CPU)
    auto lerp2 = [&](float a_, float b_, float d_, float e_) {
        const float top = fmaf(b_, wx1, a_ * wx0);
        const float bot = fmaf(e_, wx1, d_ * wx0);
        return fmaf(bot, hy1, top * hy0);
    };
    float4 r;
    r.x = lerp2(a.x, b.x, d.x, e.x);
    r.y = lerp2(a.y, b.y, d.y, e.y);
    r.z = lerp2(a.z, b.z, d.z, e.z);
    r.w = lerp2(a.w, b.w, d.w, e.w);
    if (post != 1.0f) { r.x *= post; r.y *= post; r.z *= post; r.w *= post; }
    st4(out, (size_t)pix, c, vout, r);
}

// Exact x2 upsampling (the inter-layer up-samplers and SpyNet's flow pyramid: 2.1 % of a frame): one thread per INPUT pixel
// and channel group writes the 2x2 output block. The four outputs read input rows / columns {i-1, i, i+1} (clamped), so 9
// loads serve what takes the generic kernel 16 -- it is bound by those L2 reads, not by the store. Indices and weights come
// from the same src_index() calls and the same lerp as the generic kernel: bit-identical results (tests/test_gpu_ops.py).
__global__ void resize_up2_kernel(V in, V out, float post, int cg, long long total) {
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= (unsigned)total) return;
    const unsigned pix = idx / (unsigned)cg;
    const int g = (int)(idx - pix * (unsigned)cg);
    const int iy = (int)(pix / (unsigned)in.W), ix = (int)(pix - (unsigned)iy * (unsigned)in.W);
    const int c = g * 4;
    const int ry[3] = {iy > 0 ? iy - 1 : 0, iy, iy < in.H - 1 ? iy + 1 : in.H - 1};
    const int rx[3] = {ix > 0 ? ix - 1 : 0, ix, ix < in.W - 1 ? ix + 1 : in.W - 1};
    float4 v[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) v[a][b] = *reinterpret_cast<const float4 *>(in.p + ((size_t)ry[a] * in.W + rx[b]) * in.ld + c);
    // Output row 2 iy + dy reads input rows (y0, y1) = (iy - 1, iy) for dy = 0 and (iy, iy + 1) for dy = 1, i.e. (v[0], v[1]) and
    // (v[1], v[2]); at the borders src_index() clamps exactly like ry[] does, except at iy = 0, dy = 0 where it returns
    // (0, 1) with weight 0 on row 1 -- taken from v[2] there so that even the zero-weighted operand is the generic kernel's.
    // Columns likewise. The horizontal lerps are shared between the two output rows that use the same input row.
    float hy[2][2], wx[2][2];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        int i0, i1;
        src_index(0.5f, 2 * iy + d, in.H, i0, i1, hy[d][0], hy[d][1]);
        src_index(0.5f, 2 * ix + d, in.W, i0, i1, wx[d][0], wx[d][1]);
    }
    const bool top0 = iy == 0, left0 = ix == 0;
    float4 h[3][2];                                    // h[row][dx]: horizontal lerp of input row `row` for output column 2 ix + dx
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float4 c1 = left0 ? v[a][2] : v[a][1];   // x1 of dx = 0
        auto hl = [&](const float4 &p, const float4 &q, float w0, float w1) {
            return make_float4(fmaf(q.x, w1, p.x * w0), fmaf(q.y, w1, p.y * w0), fmaf(q.z, w1, p.z * w0), fmaf(q.w, w1, p.w * w0));
        };
        h[a][0] = hl(v[a][0], c1, wx[0][0], wx[0][1]);
        h[a][1] = hl(v[a][1], v[a][2], wx[1][0], wx[1][1]);
    }
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const float4 t = dy == 0 ? h[0][dx] : h[1][dx];
            const float4 bsel = top0 ? h[2][dx] : h[1][dx];
            const float4 u = dy == 0 ? bsel : h[2][dx];
            float4 r = make_float4(fmaf(u.x, hy[dy][1], t.x * hy[dy][0]), fmaf(u.y, hy[dy][1], t.y * hy[dy][0]),
                                   fmaf(u.z, hy[dy][1], t.z * hy[dy][0]), fmaf(u.w, hy[dy][1], t.w * hy[dy][0]));
            if (post != 1.0f) { r.x *= post; r.y *= post; r.z *= post; r.w *= post; }
            *reinterpret_cast<float4 *>(out.p + ((size_t)(2 * iy + dy) * out.W + 2 * ix + dx) * out.ld + c) = r;
        }
}

// ------------------------------------------------------------------------------------------------
// flow warp = grid_sample(bilinear, border, align_corners=True) on the reference's normalised grid.
struct Bilin {
    int x0, x1, y0, y1;
    float nw, ne, sw, se;
};
// linspace(-1, 1, n)[i] as ATen's CPU kernel computes it (two-sided, fp32 step)
__device__ __forceinline__ float linspace_m1_1(int i, int n) {
    const float step = 2.0f / (float)(n - 1);
    return (i < n / 2) ? (-1.0f + step * (float)i) : (1.0f - step * (float)(n - 1 - i));
}
__device__ __forceinline__ Bilin warp_coords(int x, int y, float fx, float fy, int W, int H) {
    // torch_warp: grid = linspace + flow / ((size-1)/2)        (video_net_component.py:333-342)
    const float gx = linspace_m1_1(x, W) + fx / ((float)(((double)W - 1.0) / 2.0));
    const float gy = linspace_m1_1(y, H) + fy / ((float)(((double)H - 1.0) / 2.0));
    // grid_sample, align_corners=True: ((g + 1) * (size-1)/2), border clip, floor + lerp weights
    float ix = (gx + 1.f) * ((float)(W - 1) / 2.f);
    float iy = (gy + 1.f) * ((float)(H - 1) / 2.f);
    ix = fminf((float)(W - 1), fmaxf(ix, 0.f));
    iy = fminf((float)(H - 1), fmaxf(iy, 0.f));
    const float xw = floorf(ix), yn = floorf(iy);
    const float w = ix - xw, e = 1.f - w, n = iy - yn, s = 1.f - n;
    Bilin b;
    b.nw = s * e; b.ne = s * w; b.sw = n * e; b.se = n * w;
    b.x0 = (int)xw; b.y0 = (int)yn;
    b.x1 = min(b.x0 + 1, W - 1);  // weight is exactly 0 whenever the +1 neighbour would fall outside
    b.y1 = min(b.y0 + 1, H - 1);
    return b;
}

__global__ void flow_warp_kernel(V in, V flow, V out, int cg, long long total, int vin, int vout) {
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;      // 32-bit index arithmetic (the host refuses totals >= 2^31)
    if (idx >= (unsigned)total) return;
    const unsigned pix = idx / (unsigned)cg;
    const int g = (int)(idx - pix * (unsigned)cg);
    const int y = (int)(pix / (unsigned)out.W), x = (int)(pix - (unsigned)y * (unsigned)out.W);
    const int c = g * 4;
    const float *f = flow.p + (size_t)pix * flow.ld;
    const Bilin b = warp_coords(x, y, f[0], f[1], in.W, in.H);
    const float4 nw = ld4(in, (size_t)b.y0 * in.W + b.x0, c, vin);
    const float4 ne = ld4(in, (size_t)b.y0 * in.W + b.x1, c, vin);
    const float4 sw = ld4(in, (size_t)b.y1 * in.W + b.x0, c, vin);
    const float4 se = ld4(in, (size_t)b.y1 * in.W + b.x1, c, vin);
    float4 r;
    r.x = nw.x * b.nw + ne.x * b.ne + sw.x * b.sw + se.x * b.se;
    r.y = nw.y * b.nw + ne.y * b.ne + sw.y * b.sw + se.y * b.se;
    r.z = nw.z * b.nw + ne.z * b.ne + sw.z * b.sw + se.z * b.se;
    r.w = nw.w * b.nw + ne.w * b.ne + sw.w * b.sw + se.w * b.se;
    st4(out, (size_t)pix, c, vout, r);
}

// ------------------------------------------------------------------------------------------------
// One SpyNet level's network input in ONE launch (ME_Spynet.forward's loop body, video_net_component.py:231-246 / 308-324):
//   up  = 2 * bilinear_x2(flow_lo)                      -> out[6:8]   (resize_bilinear_kernel's arithmetic, post = 2)
//   out[3:6] = warp(im2, up)   (flow_warp_kernel's)     out[0:3] = im1
// was three launches (resize, copy, warp) per level, 8 levels per P-frame. One thread per pixel; `up` is used as computed, which
// is what the warp launch read back from memory: results are bit-identical to the three launches (tests/test_gpu_ops.py).
__global__ void spynet_prep_kernel(V im1, V im2, V flow, V out, float sy, float sx, long long total) {
    const unsigned pix = blockIdx.x * 256u + threadIdx.x;
    if (pix >= (unsigned)total) return;
    const int y = (int)(pix / (unsigned)out.W), x = (int)(pix - (unsigned)y * (unsigned)out.W);
    int y0, y1, x0, x1;
    float hy0, hy1, wx0, wx1;
    src_index(sy, y, flow.H, y0, y1, hy0, hy1);
    src_index(sx, x, flow.W, x0, x1, wx0, wx1);
    const float *a = flow.p + ((size_t)y0 * flow.W + x0) * flow.ld, *b = flow.p + ((size_t)y0 * flow.W + x1) * flow.ld;
    const float *d = flow.p + ((size_t)y1 * flow.W + x0) * flow.ld, *e = flow.p + ((size_t)y1 * flow.W + x1) * flow.ld;
    float up[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float top = fmaf(b[k], wx1, a[k] * wx0);
        const float bot = fmaf(e[k], wx1, d[k] * wx0);
        up[k] = fmaf(bot, hy1, top * hy0) * 2.0f;
    }
    float *o = out.p + (size_t)pix * out.ld;
    const float *s1 = im1.p + (size_t)pix * im1.ld;
    const Bilin w = warp_coords(x, y, up[0], up[1], im2.W, im2.H);
    const float *nw = im2.p + ((size_t)w.y0 * im2.W + w.x0) * im2.ld, *ne = im2.p + ((size_t)w.y0 * im2.W + w.x1) * im2.ld;
    const float *sw = im2.p + ((size_t)w.y1 * im2.W + w.x0) * im2.ld, *se = im2.p + ((size_t)w.y1 * im2.W + w.x1) * im2.ld;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        o[k] = s1[k];
        o[3 + k] = nw[k] * w.nw + ne[k] * w.ne + sw[k] * w.sw + se[k] * w.se;
    }
    o[6] = up[0];
    o[7] = up[1];
}

// ------------------------------------------------------------------------------------------------
__global__ void pool2x2_kernel(V in, V out, int is_max, int cg, long long total, int vin, int vout) {
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;      // 32-bit index arithmetic (the host refuses totals >= 2^31)
    if (idx >= (unsigned)total) return;
    const unsigned pix = idx / (unsigned)cg;
    const int g = (int)(idx - pix * (unsigned)cg);
    const int y = (int)(pix / (unsigned)out.W), x = (int)(pix - (unsigned)y * (unsigned)out.W);
    const int c = g * 4;
    const size_t p00 = (size_t)(2 * y) * in.W + 2 * x;
    const float4 a = ld4(in, p00, c, vin), b = ld4(in, p00 + 1, c, vin);
    const float4 d = ld4(in, p00 + in.W, c, vin), e = ld4(in, p00 + in.W + 1, c, vin);
    float4 r;
    if (is_max) {
        r.x = fmaxf(fmaxf(a.x, b.x), fmaxf(d.x, e.x));
        r.y = fmaxf(fmaxf(a.y, b.y), fmaxf(d.y, e.y));
        r.z = fmaxf(fmaxf(a.z, b.z), fmaxf(d.z, e.z));
        r.w = fmaxf(fmaxf(a.w, b.w), fmaxf(d.w, e.w));
    } else {  // ATen avg_pool2d: running sum in window order, then / 4
        r.x = (a.x + b.x + d.x + e.x) / 4.f;
        r.y = (a.y + b.y + d.y + e.y) / 4.f;
        r.z = (a.z + b.z + d.z + e.z) / 4.f;
        r.w = (a.w + b.w + d.w + e.w) / 4.f;
    }
    st4(out, (size_t)pix, c, vout, r);
}

// Three levels of 2x2 average pooling in one launch (SpyNet's image pyramids: avg_pool2d applied three times,
// video_net_component.py:225-229): one thread per pixel and channel of the COARSEST level walks its 8x8 input block level by
// level -- each level the running sum in window order, then / 4, exactly pool2x2_kernel's arithmetic on the level below, so the
// three outputs are bit-identical to three launches. Needs H % 8 == 0 and W % 8 == 0 (else the caller pools level by level).
__global__ void avgpool_pyramid3_kernel(V in, V l1, V l2, V l3, long long total) {
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= (unsigned)total) return;
    const unsigned pix = idx / (unsigned)in.C;
    const int c = (int)(idx - pix * (unsigned)in.C);
    const int y3 = (int)(pix / (unsigned)l3.W), x3 = (int)(pix - (unsigned)y3 * (unsigned)l3.W);
    float s3 = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {                      // level-2 pixels of this level-3 pixel, in window order
        const int y2 = 2 * y3 + (q >> 1), x2 = 2 * x3 + (q & 1);
        float s2 = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int y1 = 2 * y2 + (r >> 1), x1 = 2 * x2 + (r & 1);
            const float *p0 = in.p + ((size_t)(2 * y1) * in.W + 2 * x1) * in.ld + c;
            const float v1 = (p0[0] + p0[in.ld] + p0[(size_t)in.W * in.ld] + p0[((size_t)in.W + 1) * in.ld]) / 4.f;
            l1.p[((size_t)y1 * l1.W + x1) * l1.ld + c] = v1;
            s2 += v1;
        }
        const float v2 = s2 / 4.f;
        l2.p[((size_t)y2 * l2.W + x2) * l2.ld + c] = v2;
        s3 += v2;
    }
    l3.p[(size_t)pix * l3.ld + c] = s3 / 4.f;
}

// ------------------------------------------------------------------------------------------------
__global__ void softmax2_blend_kernel(V a, V b, V logits, V out, int cg, long long total, int vec) {
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;      // 32-bit index arithmetic (the host refuses totals >= 2^31)
    if (idx >= (unsigned)total) return;
    const unsigned pix = idx / (unsigned)cg;
    const int g = (int)(idx - pix * (unsigned)cg);
    const int c = g * 4;
    const float *l = logits.p + (size_t)pix * logits.ld;
    const float m = fmaxf(l[0], l[1]);
    const float e0 = expf(l[0] - m), e1 = expf(l[1] - m);
    const float sum = e0 + e1;
    const float w0 = e0 / sum, w1 = e1 / sum;
    const float4 p = ld4(a, (size_t)pix, c, vec), q = ld4(b, (size_t)pix, c, vec);
    float4 r;
    r.x = p.x * w0 + q.x * w1;
    r.y = p.y * w0 + q.y * w1;
    r.z = p.z * w0 + q.z * w1;
    r.w = p.w * w0 + q.w * w1;
    st4(out, (size_t)pix, c, vec, r);
}

// mode 0: out = a + b ; 1: out = a (out may have MORE channels than a: the extra ones are written as zeros) ; 2: out = lrelu(a)
__global__ void binary_kernel(V a, V b, V out, int mode, float slope, int cg, long long total, int vec) {
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;      // 32-bit index arithmetic (the host refuses totals >= 2^31)
    if (idx >= (unsigned)total) return;
    const unsigned pix = idx / (unsigned)cg;
    const int g = (int)(idx - pix * (unsigned)cg);
    const int c = g * 4;
    float4 r = c < a.C ? ld4(a, (size_t)pix, c, vec) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (mode == 0) {
        const float4 q = ld4(b, (size_t)pix, c, vec);
        r.x += q.x; r.y += q.y; r.z += q.z; r.w += q.w;
    } else if (mode == 2) {
        r.x = r.x > 0.f ? r.x : r.x * slope;
        r.y = r.y > 0.f ? r.y : r.y * slope;
        r.z = r.z > 0.f ? r.z : r.z * slope;
        r.w = r.w > 0.f ? r.w : r.w * slope;
    }
    st4(out, (size_t)pix, c, vec, r);
}

// ------------------------------------------------------------------------------------------------
// OffsetDiversity tail. One thread = one pixel x one fusion group G (slots n = 2G, 2G+1):
//   slot n: x channel group (n % 16) (3 channels), offset channels (2n, 2n+1) of cat(o1,o2) = om[0:64],
//   mask channel om[64+n]; warped*mask values land in the virtual 96-channel tensor at n*3+k;
//   grouped 1x1 conv, group G: inputs 6G..6G+5 -> outputs 3G..3G+2.
__global__ void offset_diversity_kernel(V x, V om, V flow, const float *__restrict__ fw, const float *__restrict__ fb,
                                        V out, long long total) {
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;      // 32-bit index arithmetic (the host refuses totals >= 2^31)
    if (idx >= (unsigned)total) return;
    const int G = (int)(idx & 15u);
    const unsigned pix = idx >> 4;
    const int py = (int)(pix / (unsigned)x.W), px = (int)(pix - (unsigned)py * (unsigned)x.W);
    const float *o = om.p + (size_t)pix * om.ld;
    const float *f = flow.p + (size_t)pix * flow.ld;
    const float f0 = f[0], f1 = f[1];
    // this thread's two offset pairs o[4G..4G+3] and two mask logits o[64+2G..+1]: one 16-byte and one 8-byte load when aligned
    float ov[4], mv[2];
    if (((om.ld & 3) | ((int)(size_t)om.p & 15)) == 0) {
        const float4 t4 = *reinterpret_cast<const float4 *>(o + 4 * G);
        const float2 t2 = *reinterpret_cast<const float2 *>(o + 64 + 2 * G);
        ov[0] = t4.x; ov[1] = t4.y; ov[2] = t4.z; ov[3] = t4.w;
        mv[0] = t2.x; mv[1] = t2.y;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) ov[j] = o[4 * G + j];
        mv[0] = o[64 + 2 * G];
        mv[1] = o[64 + 2 * G + 1];
    }
    float v[6];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = 2 * G + t;
        // flow.repeat(1, 32, 1, 1): offset channel j gets flow[j % 2]
        const float dx = 40.f * tanhf(ov[2 * t]) + f0;
        const float dy = 40.f * tanhf(ov[2 * t + 1]) + f1;
        const float mk = 1.f / (1.f + expf(-mv[t]));
        const Bilin b = warp_coords(px, py, dx, dy, x.W, x.H);
        const int cb = (n & 15) * 3;
        const float *nw = x.p + ((size_t)b.y0 * x.W + b.x0) * x.ld + cb;
        const float *ne = x.p + ((size_t)b.y0 * x.W + b.x1) * x.ld + cb;
        const float *sw = x.p + ((size_t)b.y1 * x.W + b.x0) * x.ld + cb;
        const float *se = x.p + ((size_t)b.y1 * x.W + b.x1) * x.ld + cb;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float s = nw[k] * b.nw + ne[k] * b.ne + sw[k] * b.sw + se[k] * b.se;
            v[t * 3 + k] = s * mk;
        }
    }
    float *dst = out.p + (size_t)pix * out.ld + 3 * G;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float *wr = fw + (3 * G + j) * 6;
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < 6; ++t) acc = fmaf(wr[t], v[t], acc);
        dst[j] = acc + fb[3 * G + j];
    }
}

// ------------------------------------------------------------------------------------------------
// boundary layout changes: 32 pixels x 32 channels tiles through LDS so both sides stay coalesced
__global__ void nchw_to_nhwc_kernel(const float *__restrict__ src, V dst) {
    __shared__ float tile[32][33];
    const long long hw = (long long)dst.H * dst.W;
    const long long p0 = (long long)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 8 rows per pass
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r;
        const long long p = p0 + tx;
        tile[r][tx] = (c < dst.C && p < hw) ? src[(size_t)c * hw + p] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const long long p = p0 + r;
        const int c = c0 + tx;
        if (c < dst.C && p < hw) dst.p[(size_t)p * dst.ld + c] = tile[tx][r];
    }
}
__global__ void nhwc_to_nchw_kernel(V src, float *__restrict__ dst) {
    __shared__ float tile[32][33];
    const long long hw = (long long)src.H * src.W;
    const long long p0 = (long long)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const long long p = p0 + r;
        const int c = c0 + tx;
        tile[r][tx] = (c < src.C && p < hw) ? src.p[(size_t)p * src.ld + c] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r;
        const long long p = p0 + tx;
        if (c < src.C && p < hw) dst[(size_t)c * hw + p] = tile[tx][r];
    }
}

}  // namespace lssvc

using namespace lssvc;

extern "C" int lssvc_dwconv3x3(const lssvc_view *in, const float *weight, const float *bias, const lssvc_view *out,
                               void *stream) {
    LSSVC_CHECK(view_ok(in) && view_ok(out) && weight && bias, "dwconv3x3: bad arguments");
    LSSVC_CHECK(same_shape(in, out), "dwconv3x3: in %dx%dx%d vs out %dx%dx%d", in->H, in->W, in->C, out->H, out->W, out->C);
    LSSVC_CHECK(vec4_ok(in) && vec4_ok(out), "dwconv3x3: views must be 4-channel aligned (C=%d ld=%d)", in->C, in->ld);
    if (option_get(OPT_POINTWISE_BLOCKS) && in->H > 1 && in->W > 1) {
        const int cg = (out->C + 3) / 4, bw = (out->W + 1) / 2;
        const long long total = (long long)((out->H + 1) / 2) * bw * cg;
        LSSVC_ITEMS_OK(total, "dwconv3x3");
        hipLaunchKernelGGL(dwconv3x3_b2_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, mk(in), weight, bias, mk(out),
                           cg, bw, total);
        return launch_status("dwconv3x3(2x2 blocks)");
    }
    const Items it = items_of(out);
    LSSVC_ITEMS_OK(it.total, "dwconv3x3");
    hipLaunchKernelGGL(dwconv3x3_kernel, dim3(blocks_for(it.total)), dim3(256), 0, (hipStream_t)stream, mk(in), weight,
                       bias, mk(out), it.cg, it.total);
    return launch_status("dwconv3x3");
}

extern "C" int lssvc_resize_bilinear(const lssvc_view *in, const lssvc_view *out, float scale, void *stream) {
    LSSVC_CHECK(view_ok(in) && view_ok(out), "resize_bilinear: bad views");
    LSSVC_CHECK(in->C == out->C, "resize_bilinear: C %d vs %d", in->C, out->C);
    if (option_get(OPT_POINTWISE_BLOCKS) && out->H == 2 * in->H && out->W == 2 * in->W && vec4_ok(in) && vec4_ok(out) && in->H > 1 && in->W > 1) {
        const Items ii = items_of(in);                 // one thread per input pixel and channel group
        LSSVC_ITEMS_OK(ii.total, "resize_bilinear");
        hipLaunchKernelGGL(resize_up2_kernel, dim3(blocks_for(ii.total)), dim3(256), 0, (hipStream_t)stream, mk(in), mk(out), scale,
                           ii.cg, ii.total);
        return launch_status("resize_bilinear(x2)");
    }
    const Items it = items_of(out);
    const float sy = (float)in->H / (float)out->H, sx = (float)in->W / (float)out->W;
    LSSVC_ITEMS_OK(it.total, "resize_bilinear");
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3(blocks_for(it.total)), dim3(256), 0, (hipStream_t)stream, mk(in),
                       mk(out), sy, sx, scale, it.cg, it.total, (int)vec4_ok(in), (int)vec4_ok(out));
    return launch_status("resize_bilinear");
}

extern "C" int lssvc_flow_warp(const lssvc_view *in, const lssvc_view *flow, const lssvc_view *out, void *stream) {
    LSSVC_CHECK(view_ok(in) && view_ok(flow) && view_ok(out), "flow_warp: bad views");
    LSSVC_CHECK(flow->C == 2 && same_hw(flow, out) && same_shape(in, out), "flow_warp: in %dx%dx%d flow %dx%dx%d out %dx%dx%d",
                in->H, in->W, in->C, flow->H, flow->W, flow->C, out->H, out->W, out->C);
    LSSVC_CHECK(in->H > 1 && in->W > 1, "flow_warp: needs H,W > 1");
    const Items it = items_of(out);
    LSSVC_ITEMS_OK(it.total, "flow_warp");
    hipLaunchKernelGGL(flow_warp_kernel, dim3(blocks_for(it.total)), dim3(256), 0, (hipStream_t)stream, mk(in), mk(flow),
                       mk(out), it.cg, it.total, (int)vec4_ok(in), (int)vec4_ok(out));
    return launch_status("flow_warp");
}

extern "C" int lssvc_spynet_prep(const lssvc_view *im1, const lssvc_view *im2, const lssvc_view *flow_lo, const lssvc_view *out, void *stream) {
    LSSVC_CHECK(view_ok(im1) && view_ok(im2) && view_ok(flow_lo) && view_ok(out), "spynet_prep: bad views");
    LSSVC_CHECK(im1->C == 3 && same_shape(im1, im2) && same_hw(im1, out) && out->C == 8 && flow_lo->C == 2 && im1->H > 1 && im1->W > 1,
                "spynet_prep: im1 %dx%dx%d im2 %dx%dx%d flow %dx%dx%d out %dx%dx%d", im1->H, im1->W, im1->C, im2->H, im2->W, im2->C,
                flow_lo->H, flow_lo->W, flow_lo->C, out->H, out->W, out->C);
    const long long total = (long long)out->H * out->W;
    LSSVC_ITEMS_OK(total, "spynet_prep");
    hipLaunchKernelGGL(spynet_prep_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, mk(im1), mk(im2), mk(flow_lo), mk(out),
                       (float)flow_lo->H / (float)out->H, (float)flow_lo->W / (float)out->W, total);
    return launch_status("spynet_prep");
}

extern "C" int lssvc_avgpool_pyramid3(const lssvc_view *in, const lssvc_view *l1, const lssvc_view *l2, const lssvc_view *l3, void *stream) {
    LSSVC_CHECK(view_ok(in) && view_ok(l1) && view_ok(l2) && view_ok(l3), "avgpool_pyramid3: bad views");
    LSSVC_CHECK(in->H % 8 == 0 && in->W % 8 == 0 && l1->H == in->H / 2 && l1->W == in->W / 2 && l2->H == in->H / 4 && l2->W == in->W / 4 &&
                l3->H == in->H / 8 && l3->W == in->W / 8 && l1->C == in->C && l2->C == in->C && l3->C == in->C,
                "avgpool_pyramid3: in %dx%dx%d levels %dx%d %dx%d %dx%d", in->H, in->W, in->C, l1->H, l1->W, l2->H, l2->W, l3->H, l3->W);
    const long long total = (long long)l3->H * l3->W * l3->C;
    LSSVC_ITEMS_OK(total, "avgpool_pyramid3");
    hipLaunchKernelGGL(avgpool_pyramid3_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, mk(in), mk(l1), mk(l2), mk(l3), total);
    return launch_status("avgpool_pyramid3");
}

extern "C" int lssvc_pool2x2(const lssvc_view *in, const lssvc_view *out, int32_t is_max, void *stream) {
    LSSVC_CHECK(view_ok(in) && view_ok(out), "pool2x2: bad views");
    LSSVC_CHECK(in->C == out->C && out->H == in->H / 2 && out->W == in->W / 2, "pool2x2: in %dx%dx%d out %dx%dx%d", in->H,
                in->W, in->C, out->H, out->W, out->C);
    const Items it = items_of(out);
    LSSVC_ITEMS_OK(it.total, "pool2x2");
    hipLaunchKernelGGL(pool2x2_kernel, dim3(blocks_for(it.total)), dim3(256), 0, (hipStream_t)stream, mk(in), mk(out),
                       (int)is_max, it.cg, it.total, (int)vec4_ok(in), (int)vec4_ok(out));
    return launch_status("pool2x2");
}

extern "C" int lssvc_softmax2_blend(const lssvc_view *a, const lssvc_view *b, const lssvc_view *logits,
                                    const lssvc_view *out, void *stream) {
    LSSVC_CHECK(view_ok(a) && view_ok(b) && view_ok(logits) && view_ok(out), "softmax2_blend: bad views");
    LSSVC_CHECK(same_shape(a, b) && same_shape(a, out) && same_hw(a, logits) && logits->C == 2, "softmax2_blend: shape mismatch");
    const Items it = items_of(out);
    const int vec = vec4_ok(a) && vec4_ok(b) && vec4_ok(out);
    LSSVC_ITEMS_OK(it.total, "softmax2_blend");
    hipLaunchKernelGGL(softmax2_blend_kernel, dim3(blocks_for(it.total)), dim3(256), 0, (hipStream_t)stream, mk(a), mk(b),
                       mk(logits), mk(out), it.cg, it.total, vec);
    return launch_status("softmax2_blend");
}

static int binary(const lssvc_view *a, const lssvc_view *b, const lssvc_view *out, int mode, float slope, void *stream,
                  const char *what) {
    LSSVC_CHECK(view_ok(a) && view_ok(out) && (mode != 0 || view_ok(b)), "%s: bad views", what);
    LSSVC_CHECK((mode == 1 ? (same_hw(a, out) && out->C >= a->C) : same_shape(a, out)) && (mode != 0 || same_shape(a, b)), "%s: shape mismatch", what);
    const Items it = items_of(out);
    const int vec = vec4_ok(a) && vec4_ok(out) && (mode != 0 || vec4_ok(b));
    LSSVC_ITEMS_OK(it.total, "binary");
    hipLaunchKernelGGL(binary_kernel, dim3(blocks_for(it.total)), dim3(256), 0, (hipStream_t)stream, mk(a),
                       mode == 0 ? mk(b) : mk_null(), mk(out), mode, slope, it.cg, it.total, vec);
    return launch_status(what);
}
extern "C" int lssvc_add(const lssvc_view *a, const lssvc_view *b, const lssvc_view *out, void *stream) {
    return binary(a, b, out, 0, 0.f, stream, "add");
}
extern "C" int lssvc_copy(const lssvc_view *in, const lssvc_view *out, void *stream) {
    return binary(in, nullptr, out, 1, 0.f, stream, "copy");
}
extern "C" int lssvc_lrelu(const lssvc_view *in, const lssvc_view *out, float slope, void *stream) {
    return binary(in, nullptr, out, 2, slope, stream, "lrelu");
}

// fp32 NHWC -> PRE-SPLIT layout (lssvc_hip.h, LSSVC_PREC_SPLIT_IN): per pixel and 16-channel chunk 64 bytes, [hi: 16 x fp16 | lo: 16 x
// fp16] of act(x) saturated at +-65504 -- exactly what the f16x3 conv kernels make of an fp32 input while staging it (conv_f16x3_kernel.h
// store_patch, conv3_f16x3p.hip store_patch). One thread = one pixel x one 4-channel quad of the padded channel range; channels >= C of the
// last chunk are written as zeros.
typedef _Float16 ps_f16x4 __attribute__((ext_vector_type(4)));
__global__ void presplit_kernel(V in, V out, int in_act, float slope, int qpp, long long total, int vec) {
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= (unsigned)total) return;
    const unsigned pix = idx / (unsigned)qpp;
    const int c = (int)(idx - pix * (unsigned)qpp) * 4;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < in.C) r = ld4(in, (size_t)pix, c, vec);
    float v[4] = {r.x, r.y, r.z, r.w};
    ps_f16x4 h, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float x = v[j];
        if (in_act == LSSVC_INACT_LRELU) x = fmaxf(x, slope * x);                  // 0 <= slope <= 1 (checked by the host): exact
        x = fminf(fmaxf(x, -65504.f), 65504.f);
        h[j] = (_Float16)x;
        l[j] = (_Float16)(x - (float)h[j]);
    }
    _Float16 *o = reinterpret_cast<_Float16 *>(out.p + (size_t)pix * out.ld + (c >> 4) * 16) + (c & 15);
    *reinterpret_cast<ps_f16x4 *>(o) = h;
    *reinterpret_cast<ps_f16x4 *>(o + 16) = l;
}
extern "C" int lssvc_presplit(const lssvc_view *in, const lssvc_view *out, int32_t in_act, float in_slope, void *stream) {
    LSSVC_CHECK(view_ok(in) && view_ok(out) && same_shape(in, out), "presplit: bad views");
    LSSVC_CHECK(in_act == LSSVC_INACT_NONE || (in_act == LSSVC_INACT_LRELU && in_slope >= 0.0f && in_slope <= 1.0f), "presplit: in_act %d slope %g", in_act, in_slope);
    LSSVC_CHECK(out->ld % 16 == 0 && out->ld >= (out->C + 15) / 16 * 16 && (reinterpret_cast<uintptr_t>(out->ptr) & 63) == 0,
                "presplit: the pre-split view needs a 64-byte aligned base and a pixel pitch that is a multiple of 16 and covers its padded chunks (C=%d ld=%d)", out->C, out->ld);
    {   // the pre-split layout permutes bytes inside every 64-byte chunk across threads: in-place (or overlapping) use would corrupt data silently
        const uintptr_t a0 = reinterpret_cast<uintptr_t>(in->ptr), a1 = a0 + (size_t)in->H * in->W * in->ld * sizeof(float);
        const uintptr_t b0 = reinterpret_cast<uintptr_t>(out->ptr), b1 = b0 + (size_t)out->H * out->W * out->ld * sizeof(float);
        LSSVC_CHECK(a1 <= b0 || b1 <= a0, "presplit: `in` and `out` overlap (the conversion cannot run in place)");
    }
    const int qpp = (out->C + 15) / 16 * 4;
    const long long total = (long long)out->H * out->W * qpp;
    LSSVC_ITEMS_OK(total, "presplit");
    hipLaunchKernelGGL(presplit_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, mk(in), mk(out), (int)in_act, in_slope, qpp,
                       total, (int)vec4_ok(in));
    return launch_status("presplit");
}

// F.pad(x, (left, right, top, bottom), value 0) with negative entries cropping, as ONE launch that writes every element of
// `out` (source pixel (y - top, x - left) where it exists, zero elsewhere): get_depadded_feature, IntraSS.py:124-135.
__global__ void pad_crop_kernel(V in, V out, int left, int top, int cg, long long total, int vec) {
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= (unsigned)total) return;
    const unsigned pix = idx / (unsigned)cg;
    const int c = (int)(idx - pix * (unsigned)cg) * 4;
    const int y = (int)(pix / (unsigned)out.W), x = (int)(pix - (unsigned)y * (unsigned)out.W);
    const int sy = y - top, sx = x - left;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (sy >= 0 && sy < in.H && sx >= 0 && sx < in.W) r = ld4(in, (size_t)sy * in.W + sx, c, vec);
    st4(out, (size_t)pix, c, vec, r);
}
extern "C" int lssvc_pad_crop(const lssvc_view *in, const lssvc_view *out, int32_t left, int32_t top, void *stream) {
    LSSVC_CHECK(view_ok(in) && view_ok(out) && in->C == out->C, "pad_crop: bad views");
    const Items it = items_of(out);
    const int vec = vec4_ok(in) && vec4_ok(out);
    LSSVC_ITEMS_OK(it.total, "pad_crop");
    hipLaunchKernelGGL(pad_crop_kernel, dim3(blocks_for(it.total)), dim3(256), 0, (hipStream_t)stream, mk(in), mk(out), (int)left, (int)top,
                       it.cg, it.total, vec);
    return launch_status("pad_crop");
}

// ---- range audit: max |x| of a view (NaN / Inf count as +Inf) ------------------------------------------------------------
// Non-negative floats order like their bit patterns, so the grid reduces with one integer atomicMax per wave: no workspace,
// no fp64, order-independent. The caller zeroes *out_max (or keeps accumulating the maximum over several views into it).
__global__ void absmax_kernel(V x, unsigned int *out_max, long long total) {
    unsigned int m = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % x.C);
        const size_t pix = (size_t)(i / x.C);
        const float v = fabsf(x.p[pix * x.ld + c]);
        const unsigned int b = (v == v) ? __float_as_uint(v) : 0x7f800000u;
        m = b > m ? b : m;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned int other = (unsigned int)__shfl_xor((int)m, o, 64);
        m = other > m ? other : m;
    }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out_max, m);
}

// ---- plumbing the compiled frame plans need as library calls (csrc/plan_runtime.cpp replays only C-ABI launches) ----------
__global__ void clamp_flat_kernel(float *x, float lo, float hi, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) x[i] = fminf(fmaxf(x[i], lo), hi);
}

// Zero fill as a KERNEL launch, not hipMemsetAsync (round 5): inside a stream capture a memset becomes a memset NODE of the hipGraph
// (DESIGN section 6.1 for what that did to replayed frame plans); a kernel node is ordered like every other launch of the frame.
__global__ void fill_zero_kernel(unsigned char *p, long long nbytes) {
    const long long head = ((16 - (long long)(reinterpret_cast<uintptr_t>(p) & 15)) & 15) < nbytes ? ((16 - (long long)(reinterpret_cast<uintptr_t>(p) & 15)) & 15) : nbytes;
    const long long n16 = (nbytes - head) / 16;
    uint4 *q = reinterpret_cast<uint4 *>(p + head);
    const long long tid = (long long)blockIdx.x * 256 + threadIdx.x, stride = (long long)gridDim.x * 256;
    for (long long i = tid; i < n16; i += stride) q[i] = make_uint4(0u, 0u, 0u, 0u);
    for (long long i = tid; i < head; i += stride) p[i] = 0;
    for (long long i = head + n16 * 16 + tid; i < nbytes; i += stride) p[i] = 0;
}
extern "C" int lssvc_fill_zero(void *ptr, int64_t nbytes, void *stream) {
    LSSVC_CHECK(ptr && nbytes >= 0, "fill_zero: bad arguments");
    if (nbytes == 0) return 0;
    static const int use_memset = getenv("LSSVC_FILL_MEMSET") ? atoi(getenv("LSSVC_FILL_MEMSET")) : 0;      // the round-4 form, for the record
    if (use_memset) {
        LSSVC_HIP(hipMemsetAsync(ptr, 0, (size_t)nbytes, (hipStream_t)stream));
        return 0;
    }
    const long long blocks = (nbytes / 16 + 255) / 256 + 1;
    hipLaunchKernelGGL(fill_zero_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<unsigned char *>(ptr), (long long)nbytes);
    return launch_status("fill_zero");
}

extern "C" int lssvc_clamp_inplace(float *x, int64_t n, float lo, float hi, void *stream) {
    LSSVC_CHECK(x && n >= 0, "clamp_inplace: bad arguments");
    if (n == 0) return 0;
    const long long blocks = (n + 255) / 256;
    hipLaunchKernelGGL(clamp_flat_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, (hipStream_t)stream, x, lo, hi,
                       (long long)n);
    return launch_status("clamp_inplace");
}

extern "C" int lssvc_absmax(const lssvc_view *x, float *out_max, void *stream) {
    LSSVC_CHECK(view_ok(x) && out_max, "absmax: bad arguments");
    const long long total = (long long)x->H * x->W * x->C;
    const long long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, (hipStream_t)stream, mk(x),
                       reinterpret_cast<unsigned int *>(out_max), total);
    return launch_status("absmax");
}

extern "C" int lssvc_offset_diversity(const lssvc_view *x, const lssvc_view *om, const lssvc_view *flow,
                                      const float *fusion_w, const float *fusion_b, const lssvc_view *out, void *stream) {
    LSSVC_CHECK(view_ok(x) && view_ok(om) && view_ok(flow) && view_ok(out) && fusion_w && fusion_b, "offset_diversity: bad arguments");
    LSSVC_CHECK(x->C == 48 && om->C == 96 && flow->C == 2 && out->C == 48, "offset_diversity: channels x=%d om=%d flow=%d out=%d",
                x->C, om->C, flow->C, out->C);
    LSSVC_CHECK(same_hw(x, om) && same_hw(x, flow) && same_hw(x, out) && x->H > 1 && x->W > 1, "offset_diversity: size mismatch");
    const long long total = (long long)x->H * x->W * 16;
    LSSVC_ITEMS_OK(total, "offset_diversity");
    hipLaunchKernelGGL(offset_diversity_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, mk(x), mk(om),
                       mk(flow), fusion_w, fusion_b, mk(out), total);
    return launch_status("offset_diversity");
}

extern "C" int lssvc_nchw_to_nhwc(const float *src, const lssvc_view *dst, void *stream) {
    LSSVC_CHECK(src && view_ok(dst), "nchw_to_nhwc: bad arguments");
    const long long hw = (long long)dst->H * dst->W;
    dim3 grid((unsigned)((hw + 31) / 32), (unsigned)((dst->C + 31) / 32));
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, mk(dst));
    return launch_status("nchw_to_nhwc");
}
extern "C" int lssvc_nhwc_to_nchw(const lssvc_view *src, float *dst, void *stream) {
    LSSVC_CHECK(dst && view_ok(src), "nhwc_to_nchw: bad arguments");
    const long long hw = (long long)src->H * src->W;
    dim3 grid((unsigned)((hw + 31) / 32), (unsigned)((src->C + 31) / 32));
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, grid, dim3(256), 0, (hipStream_t)stream, mk(src), dst);
    return launch_status("nhwc_to_nchw");
}
