// Does v_mfma_f32_16x16x16_f16 (K = 16) cost half of v_mfma_f32_16x16x32_f16 on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool K16>
__global__ __launch_bounds__(256, 1) void probe(float *out, int iters) {
    const int lane = threadIdx.x & 63;
    f16x8 a8, b8;
    f16x4 a4, b4;
    for (int j = 0; j < 8; ++j) { a8[j] = (_Float16)(0.01f * (lane + j)); b8[j] = (_Float16)(0.02f * (lane - j)); }
    for (int j = 0; j < 4; ++j) { a4[j] = a8[j]; b4[j] = b8[j]; }
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (K16) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[i], 0, 0, 0);
            else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[i], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    float *out;
    hipMalloc(&out, sizeof(float) * 256 * 256);
    const int iters = 20000;
    for (int v = 0; v < 2; ++v) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (v) hipLaunchKernelGGL((probe<true>), dim3(256), dim3(256), 0, 0, out, iters);
            else hipLaunchKernelGGL((probe<false>), dim3(256), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        const double n = 16.0 * iters;   // MFMAs per wave
        printf("%s: %.3f ms, %.2f ns per MFMA per wave\n", v ? "16x16x16_f16" : "16x16x32_f16", ms, ms * 1e6 / n);
    }
    return 0;
}
