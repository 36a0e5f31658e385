// load_probe.hip -- per-CU global LOAD throughput of the lane patterns the MFMA kernels use (diagnostic, not part of the library)
//   hipcc --offload-arch=gfx950 -O3 -o load_probe.bin tools/probes/load_probe.hip && ./load_probe.bin [grid]
// Every wave streams rows of 16 pixels x C floats (C = 64) with dwordx4 loads in one of the patterns:
//   0: MFMA B-fragment order: lane (lg, li) reads channels 8 lg .. 8 lg + 7 of pixel li (2 x 16 B): 16 consecutive lanes = 16 pixels
//   1: 4 consecutive lanes = 64 contiguous bytes of one pixel (the producers' patch order)
//   2: wave-contiguous 1 KiB per instruction
//   3: MFMA D-fragment order (epilogue residual): lane (lg, li) reads channels 4 lg .. + 3 of fragment f of pixel li
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PAT>
__global__ __launch_bounds__(256, 1) void probe(const float *in, float *sink, int iters, long long *cycles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int C = 64;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const size_t grp = ((size_t)it * gridDim.x + blockIdx.x) * 4 + wave;        // 16 pixels = 4 KiB per group
        const float *row = in + grp * 16 * C;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            f32x4 v;
            if (PAT == 0) v = *reinterpret_cast<const f32x4 *>(row + li * C + (f >> 1) * 32 + lg * 8 + (f & 1) * 4);
            if (PAT == 1) v = *reinterpret_cast<const f32x4 *>(row + (lane >> 2) * C + f * 16 + (lane & 3) * 4);
            if (PAT == 2) v = *reinterpret_cast<const f32x4 *>(row + f * 256 + lane * 4);
            if (PAT == 3) v = *reinterpret_cast<const f32x4 *>(row + li * C + f * 16 + lg * 4);
            s += v;
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cycles[blockIdx.x * 4 + wave] = t1 - t0;
    if (s[0] + s[1] + s[2] + s[3] == 123.456f) sink[0] = s[0];
}

int main(int argc, char **argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 64;
    const int iters = 512;
    const size_t n = (size_t)iters * 256 * 4 * 16 * 64;       // sized for the largest grid
    float *in, *sink;
    long long *cyc, host[1024];
    if (hipMalloc(&in, n * 4) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess || hipMalloc(&cyc, sizeof(host)) != hipSuccess) return 1;
    (void)hipMemset(in, 0, n * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    static const char *names[] = {"B-fragment order (16 lanes = 16 pixels, 32 B each)", "4 lanes = 64 B of a pixel", "wave-contiguous", "D-fragment order (16 lanes = 16 pixels, 16 B each)"};
    for (int pat = 0; pat < 4; ++pat) {
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            (void)hipEventRecord(e0);
            if (pat == 0) probe<0><<<grid, 256>>>(in, sink, iters, cyc);
            if (pat == 1) probe<1><<<grid, 256>>>(in, sink, iters, cyc);
            if (pat == 2) probe<2><<<grid, 256>>>(in, sink, iters, cyc);
            if (pat == 3) probe<3><<<grid, 256>>>(in, sink, iters, cyc);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        (void)hipMemcpy(host, cyc, sizeof(host), hipMemcpyDeviceToHost);
        double s = 0;
        for (int i = 0; i < grid * 4; ++i) s += host[i];
        const double bytes = (double)grid * 4 * iters * 4096;
        printf("grid %3d  %-52s: %.1f us, %.2f TB/s, %.1f B/clk/CU\n", grid, names[pat], best * 1e3, bytes / best / 1e9, bytes / grid / (s / (grid * 4)));
    }
    return 0;
}
