// Round 6, VERDICT r5 item 1c ("find why it streams at 45 % of achievable"): how fast can the CUs pull the PATCHES of the persistent 3x3
// kernels out of an NHWC fp32 tensor when nothing else is in the way -- no LDS stores, no weight DMA, no consumers -- and what does the rate
// depend on? The producers of conv3_f16x3p_kernel read, per 16-channel phase, a TH x TW-pixel patch as 64-byte pieces (16 channels x 4 bytes)
// at a stride of C x 4 bytes: half of a 128-byte line per request, the other half asked for one phase later. This probe replays exactly
// that address stream with a persistent grid and varies
//   the piece   64 B (one phase's 16 channels) | 128 B (two phases at once: a whole line) | 256 B (four phases: the whole pixel at C = 64)
//   the depth   float4 requests in flight per thread (one register set | two: the next piece requested before the current one is used)
//   the waves   loading waves per CU (4 | 8 | 12 | 16; 3 = the split-roles schedule's patch waves, 6 = two such workgroups)
// against a plain linear read of the same tensor. Rates are ALGORITHMIC tensor bytes (H x W x C x 4, each pixel once) per second, the
// accounting of the conv kernels' read side; halo pixels are read again by the neighbouring tile, as in the kernels.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/patch_stream_probe.hip -o /tmp/psp && /tmp/psp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                        \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

struct Geo {
    int H, W, C;            // tensor
    int th, tw;             // output tile (pixels of the INPUT grid the tile owns: 16x16 narrow, 16x32 for the stride-2 kernel's 8x16 outputs)
    int ph, pw;             // patch = tile + halo
    int tiles_y, tiles_x;
    int scatter;            // 0: a workgroup walks all pieces of a tile (the kernels' order); 1: unit n -> workgroup n % grid, so the halves of a
                            // line are asked for by DIFFERENT CUs (mostly on different XCDs, i.e. behind different L2s)
};

// One piece of PIECE_CH channels of one patch: items = ph * pw * PIECE_CH / 4 float4, item -> (pixel, part). NL = items per thread (compile time).
// the largest workgroup an instantiation is built for: its register sets (NL float4, once or twice) must fit the per-wave budget without spills
constexpr int max_threads(int nl, bool two) {
    const int regs = nl * 4 * (two ? 2 : 1) + 3 * nl + 56;
    return regs <= 128 ? 1024 : regs <= 168 ? 768 : regs <= 256 ? 512 : 256;
}

template <int PIECE_CH, int NL, bool TWO_SETS>
__global__ __launch_bounds__(max_threads(NL, TWO_SETS)) void patch_stream(const float *__restrict__ x, unsigned *__restrict__ out, Geo g) {
    constexpr int PARTS = PIECE_CH / 4;
    const int nthreads = blockDim.x, tid = threadIdx.x;
    const int pieces = g.C / PIECE_CH, items = g.ph * g.pw * PARTS;
    unsigned acc = 0;
    float4 r[2][NL];
    int off[NL];                                   // item -> element offset inside the patch: the same for every tile (as in the kernels)
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        const int it = tid + j * nthreads, pix = it / PARTS, part = it % PARTS;
        off[j] = it < items ? ((pix / g.pw) * g.W + pix % g.pw) * g.C + part * 4 : -1;
    }
    auto request = [&](long long u, float4 *dst) {
        const int piece = (int)(u % pieces);
        const long long t = u / pieces;
        const int ty = (int)(t / g.tiles_x), tx = (int)(t % g.tiles_x);
        int y0 = ty * g.th - 1, x0 = tx * g.tw - 1;      // (edge patches shifted inside instead of zero-filled: same request count)
        y0 = y0 < 0 ? 0 : (y0 + g.ph > g.H ? g.H - g.ph : y0);
        x0 = x0 < 0 ? 0 : (x0 + g.pw > g.W ? g.W - g.pw : x0);
        const float *base = x + ((long long)y0 * g.W + x0) * g.C + piece * PIECE_CH;
#pragma unroll
        for (int j = 0; j < NL; ++j) dst[j] = off[j] >= 0 ? *reinterpret_cast<const float4 *>(base + off[j]) : make_float4(0, 0, 0, 0);
    };
    auto use = [&](const float4 *src) {
#pragma unroll
        for (int j = 0; j < NL; ++j) acc ^= __float_as_uint(src[j].x) ^ __float_as_uint(src[j].y) ^ __float_as_uint(src[j].z) ^ __float_as_uint(src[j].w);
    };
    // unit n of this workgroup = (its n / pieces-th tile, piece n % pieces): a workgroup walks ALL pieces of a tile before the next tile, as the
    // kernels do (the other half of a 64-byte piece's line is asked for by the same CU, one unit later)
    const long long n_tiles = (long long)g.tiles_y * g.tiles_x;
    const long long mine = g.scatter ? (n_tiles * pieces - blockIdx.x + gridDim.x - 1) / gridDim.x : (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x * pieces;
    auto unit = [&](long long n) { return g.scatter ? blockIdx.x + n * gridDim.x : (blockIdx.x + (n / pieces) * gridDim.x) * pieces + n % pieces; };
    if (TWO_SETS) {
        if (mine > 0) request(unit(0), r[0]);
        for (long long n = 0; n < mine; n += 2) {
            if (n + 1 < mine) request(unit(n + 1), r[1]);
            use(r[0]);
            if (n + 2 < mine) request(unit(n + 2), r[0]);
            if (n + 1 < mine) use(r[1]);
        }
    } else {
        for (long long n = 0; n < mine; ++n) {
            request(unit(n), r[0]);
            use(r[0]);
        }
    }
    if (acc == 0x12345678u) out[blockIdx.x * nthreads + tid] = acc;      // (never: keeps the loads alive)
}

template <int NL>
__global__ __launch_bounds__(1024) void linear_stream(const float4 *__restrict__ x, unsigned *__restrict__ out, long long n4) {
    unsigned acc = 0;
    const long long stride = (long long)gridDim.x * blockDim.x * NL;
    for (long long i = (long long)blockIdx.x * blockDim.x * NL + threadIdx.x; i < n4; i += stride) {
        float4 r[NL];
#pragma unroll
        for (int j = 0; j < NL; ++j) r[j] = i + (long long)j * blockDim.x < n4 ? x[i + (long long)j * blockDim.x] : make_float4(0, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NL; ++j) acc ^= __float_as_uint(r[j].x) ^ __float_as_uint(r[j].y) ^ __float_as_uint(r[j].z) ^ __float_as_uint(r[j].w);
    }
    if (acc == 0x12345678u) out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

static float time_us(hipEvent_t a, hipEvent_t b, int reps) {
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1000.f / reps;
}

template <int PIECE_CH, int NL, bool TWO>
static void run(const char *what, const float *x, unsigned *out, Geo g, int threads, int wg_per_cu, int cus) {
    const int items = g.ph * g.pw * PIECE_CH / 4;
    if ((items + threads - 1) / threads != NL || g.C % PIECE_CH) return;      // this instantiation is for another (threads, piece) pair
    if (threads > max_threads(NL, TWO) || NL * 4 * (TWO ? 2 : 1) + 3 * NL + 56 > 500) return;      // would spill
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    const int reps = 10;
    dim3 grid(cus * wg_per_cu), block(threads);
    hipLaunchKernelGGL((patch_stream<PIECE_CH, NL, TWO>), grid, block, 0, 0, x, out, g);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((patch_stream<PIECE_CH, NL, TWO>), grid, block, 0, 0, x, out, g);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    const float us = time_us(a, b, reps);
    const double bytes = (double)g.H * g.W * g.C * 4;
    printf("  %-28s piece %3d B  %2d waves x %d WG/CU  %2d float4 per thread %s  %8.1f us  %5.2f TB/s\n", what, PIECE_CH * 4, threads / 64, wg_per_cu, NL,
           TWO ? "x 2 sets" : "        ", us, bytes / us * 1e-6);
    CK(hipEventDestroy(a));
    CK(hipEventDestroy(b));
}

template <int PIECE_CH>
static void sweep(const char *what, const float *x, unsigned *out, Geo g, int cus) {
    for (int threads : {192, 256, 512, 768, 1024})
        for (int wg : {1, 2}) {
            if (threads * wg > 1024) continue;      // (<= 16 waves per CU loading)
#define R(NL)                                                       \
    run<PIECE_CH, NL, false>(what, x, out, g, threads, wg, cus);    \
    run<PIECE_CH, NL, true>(what, x, out, g, threads, wg, cus);
            R(1) R(2) R(3) R(4) R(5) R(6) R(7) R(8) R(9) R(10) R(11) R(12) R(13) R(14) R(16) R(18) R(20) R(21) R(24) R(27) R(28) R(36)
#undef R
        }
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("%s, %d CUs\n", prop.name, cus);
    for (int C : {64, 48}) {
        const int H = 1152, W = 1920;
        const size_t n = (size_t)H * W * C;
        float *x;
        unsigned *out;
        CK(hipMalloc(&x, n * 4));
        CK(hipMalloc(&out, (size_t)cus * 2 * 1024 * 4));
        std::vector<float> h(n);
        for (size_t i = 0; i < n; ++i) h[i] = (float)(i % 977) * 0.001f;
        CK(hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice));
        printf("tensor %dx%dx%d fp32 NHWC, %.0f MB\n", H, W, C, n * 4e-6);
        {   // linear reads
            hipEvent_t a, b;
            CK(hipEventCreate(&a));
            CK(hipEventCreate(&b));
            for (int wg : {1, 2, 4, 8}) {
                hipLaunchKernelGGL((linear_stream<8>), dim3(cus * wg), dim3(256), 0, 0, (const float4 *)x, out, (long long)(n / 4));
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(a));
                for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((linear_stream<8>), dim3(cus * wg), dim3(256), 0, 0, (const float4 *)x, out, (long long)(n / 4));
                CK(hipEventRecord(b));
                CK(hipEventSynchronize(b));
                const float us = time_us(a, b, 10);
                printf("  linear read, 8 float4 per thread, 4 waves x %d WG/CU: %8.1f us  %5.2f TB/s\n", wg, us, n * 4.0 / us * 1e-6);
            }
        }
        Geo narrow{H, W, C, 16, 16, 18, 18, (H + 15) / 16, (W + 15) / 16, 0};      // conv3n: 16x16 tiles, 18x18 patch
        Geo s2{H, W, C, 16, 32, 17, 33, (H + 15) / 16, (W + 31) / 32, 0};          // stride 2: 8x16 outputs <- 17x33 inputs
        Geo big{H, W, C, 24, 16, 26, 18, (H + 23) / 24, (W + 15) / 16, 0};         // the 24x16 kernel
        printf(" narrow-head patches (18x18 of 16x16):\n");
        sweep<16>("narrow 18x18", x, out, narrow, cus);
        sweep<32>("narrow 18x18", x, out, narrow, cus);
        if (C % 64 == 0) sweep<64>("narrow 18x18", x, out, narrow, cus);
        if (C == 48) sweep<48>("narrow 18x18", x, out, narrow, cus);
        printf(" stride-2 patches (17x33 of 16x32):\n");
        sweep<16>("stride-2 17x33", x, out, s2, cus);
        sweep<32>("stride-2 17x33", x, out, s2, cus);
        if (C % 64 == 0) sweep<64>("stride-2 17x33", x, out, s2, cus);
        if (C == 48) sweep<48>("stride-2 17x33", x, out, s2, cus);
        printf(" 24x16 patches (26x18):\n");
        sweep<16>("24x16 26x18", x, out, big, cus);
        printf(" the two 64-byte halves of a line asked for by different workgroups (unit n -> workgroup n %% grid):\n");
        narrow.scatter = s2.scatter = 1;
        sweep<16>("narrow 18x18, scattered", x, out, narrow, cus);
        sweep<16>("stride-2 17x33, scattered", x, out, s2, cus);
        CK(hipFree(x));
        CK(hipFree(out));
    }
    return 0;
}
