// Does the GRANULARITY of a phase's loads matter? The persistent 3x3 kernel's producers read a tile's pixels one 16-channel slice
// (64 B of each pixel's 256 B) per phase; this probe reads the same bytes as S-byte slices per pass, S = 64 / 128 / 256, over tiles of
// 468 pixels x 64 channels (fp32), one 256-thread workgroup per tile at a time, 256 persistent workgroups x 2 per CU.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/slice_load_probe.hip -o /tmp/slice_probe && /tmp/slice_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int LPP>   // lanes per pixel: 4 (64-B slices, 4 passes), 8 (128 B, 2 passes), 16 (256 B, 1 pass)
__global__ __launch_bounds__(256) void probe(const float4 *x, float *sink, int ntiles) {
    constexpr int TP = 468, PASSES = 16 / LPP, PPI = 256 / LPP;     // pixels per tile, passes per tile, pixels per instruction round
    float4 acc = make_float4(0, 0, 0, 0);
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const float4 *base = x + (size_t)t * TP * 16;
        for (int pass = 0; pass < PASSES; ++pass) {
            float4 v[(TP + PPI - 1) / PPI];
#pragma unroll
            for (int i = 0; i < (TP + PPI - 1) / PPI; ++i) {
                int p = i * PPI + threadIdx.x / LPP;
                p = p < TP ? p : TP - 1;
                v[i] = base[(size_t)p * 16 + pass * LPP + threadIdx.x % LPP];
            }
#pragma unroll
            for (int i = 0; i < (TP + PPI - 1) / PPI; ++i) { acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w; }
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}

template <int LPP>
static void run(const float4 *x, float *sink, int ntiles, const char *what) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(probe<LPP>, dim3(512), dim3(256), 0, 0, x, sink, ntiles);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)ntiles * 468 * 256 * 10;
    printf("%-28s %7.1f us per sweep  %6.2f TB/s\n", what, ms * 100, bytes / (ms * 1e-3) * 1e-12);
}

int main() {
    const int ntiles = 1152 * 1920 / 384;          // as many tiles as a 1152x1920 map has 24x16 tiles (5760); 468 = with halo
    const size_t n = (size_t)ntiles * 468 * 16;
    float4 *x; float *sink;
    hipMalloc(&x, n * sizeof(float4)); hipMalloc(&sink, 16);
    hipMemset(x, 0, n * sizeof(float4));
    run<4>(x, sink, ntiles, "64-byte slices, 4 passes");
    run<8>(x, sink, ntiles, "128-byte slices, 2 passes");
    run<16>(x, sink, ntiles, "256-byte pixels, 1 pass");
    run<4>(x, sink, ntiles, "64-byte slices, 4 passes");
    return 0;
}
