// The store-side twin of slice_load_probe.hip: a tile's 384 pixels x 64 fp32 channels written as S-byte slices per pass, S = 64 / 128 /
// 256 -- the persistent 3x3 kernel's epilogue writes 64-byte slices (one 16-channel fragment of 16 pixels per store instruction).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/slice_store_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>

template <int LPP>
__global__ __launch_bounds__(256) void probe(float4 *x, int ntiles) {
    constexpr int TP = 384, PASSES = 16 / LPP, PPI = 256 / LPP;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        float4 *base = x + (size_t)t * TP * 16;
        for (int pass = 0; pass < PASSES; ++pass) {
#pragma unroll
            for (int i = 0; i < TP / PPI; ++i) {
                const int p = i * PPI + threadIdx.x / LPP;
                base[(size_t)p * 16 + pass * LPP + threadIdx.x % LPP] = make_float4((float)t, (float)pass, (float)i, 1.f);
            }
        }
    }
}

// the epilogue's actual order: for every group of 16 pixels the four 64-byte slices back to back (same lines, consecutive instructions)
__global__ __launch_bounds__(256) void probe_rowwise(float4 *x, int ntiles) {
    constexpr int TP = 384;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        float4 *base = x + (size_t)t * TP * 16;
#pragma unroll
        for (int r = 0; r < TP / 64; ++r) {                    // each wave: 6 groups of 16 pixels
            const int p = (wave * (TP / 64) + r) * 16 + (lane >> 2);
#pragma unroll
            for (int f = 0; f < 4; ++f) base[(size_t)p * 16 + f * 4 + (lane & 3)] = make_float4((float)t, (float)f, (float)r, 1.f);
        }
    }
}

template <int LPP>
static void run(float4 *x, int ntiles, const char *what) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(probe<LPP>, dim3(512), dim3(256), 0, 0, x, ntiles);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double bytes = (double)ntiles * 384 * 256 * 10;
    printf("%-28s %7.1f us per sweep  %6.2f TB/s\n", what, ms * 100, bytes / (ms * 1e-3) * 1e-12);
}

int main() {
    const int ntiles = 1152 * 1920 / 384;
    float4 *x;
    (void)hipMalloc(&x, (size_t)ntiles * 384 * 16 * sizeof(float4));
    run<4>(x, ntiles, "64-byte slices, 4 passes");
    run<8>(x, ntiles, "128-byte slices, 2 passes");
    run<16>(x, ntiles, "256-byte pixels, 1 pass");
    run<4>(x, ntiles, "64-byte slices, 4 passes");
    {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0);
            for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(probe_rowwise, dim3(512), dim3(256), 0, 0, x, ntiles);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        printf("%-28s %7.1f us per sweep  %6.2f TB/s\n", "64-byte slices, row-wise", ms * 100, (double)ntiles * 384 * 256 * 10 / (ms * 1e-3) * 1e-12);
    }
    return 0;
}
