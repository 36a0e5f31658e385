// Bare v_mfma_f32_16x16x32_f16 issue-rate probe: W waves per CU-workgroup, NACC independent accumulators,
// operands in registers (optionally re-read from LDS each step). Prints achieved TFLOP/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, bool LDS>
__global__ __launch_bounds__(512, 1) void probe(float *out, int iters) {
    __shared__ __attribute__((aligned(16))) _Float16 sm[64 * 8 * 16];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 64 * 8 * 16; i += blockDim.x) sm[i] = (_Float16)(0.001f * (i & 63));
    __syncthreads();
    f16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = *reinterpret_cast<const f16x8 *>(sm + (lane + i * 64) * 8);
        b[i] = *reinterpret_cast<const f16x8 *>(sm + (lane + (i + 4) * 64) * 8);
    }
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        if (LDS) {
            for (int i = 0; i < 4; ++i) {
                a[i] = *reinterpret_cast<const f16x8 *>(sm + ((lane + i * 64 + it) & 511) * 8);
                b[i] = *reinterpret_cast<const f16x8 *>(sm + ((lane + (i + 4) * 64 + it) & 511) * 8);
            }
        }
#pragma unroll
        for (int rep = 0; rep < 3; ++rep)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, bool LDS>
void run(const char *name, int waves, int blocks_per_cu) {
    int cus = 256;
    float *out;
    hipMalloc(&out, sizeof(float) * 512 * cus * 4);
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<NACC, LDS>), dim3(cus * blocks_per_cu), dim3(64 * waves), 0, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = 2.0 * 16 * 16 * 32 * 3.0 * NACC * iters * waves * cus * blocks_per_cu;
        if (rep == 2) printf("%-28s waves/CU=%2d  %8.3f ms  %8.1f TFLOP/s\n", name, waves * blocks_per_cu, ms, flops / ms / 1e9);
    }
    hipFree(out);
}

int main() {
    run<16, false>("16 acc, regs", 4, 1);
    run<16, false>("16 acc, regs", 8, 1);
    run<16, false>("16 acc, regs", 8, 2);
    run<16, true>("16 acc, 8 ds_read_b128/48", 8, 1);
    run<16, true>("16 acc, 8 ds_read_b128/48", 4, 1);
    run<8, false>("8 acc, regs", 8, 1);
    return 0;
}
