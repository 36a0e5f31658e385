// clock_probe.hip -- what the chip sustains for the persistent 3x3 kernel's instruction mix (round 5; MI355X_MICROARCH.md "DVFS
// give-back" item 6): v_mfma_f32_16x16x32_f16 loops, one computing wave per SIMD on every CU (launched as 8-wave workgroups like the conv kernel, so that a wave has its 256 registers), after >= 2 s of back-to-back launches, with the
// in-kernel clock (delta s_memtime / delta s_memrealtime x 100 MHz, median over workgroups) beside the rate:
//   zeros        operands all zero, in registers                       (the clock the chip holds when the MFMAs toggle nothing)
//   random-reg   random fp16 operands (hi | lo pairs of random fp32 values, as the f16x3 convs feed them), in registers
//   random-lds   the same, every fragment re-read from LDS by ds_read_b128 at the kernel's ratio: 20 reads per 72 MFMAs
//                (8 weight + 12 pixel fragments per K step of an MF = 4 x RPW = 6 tile)
// Build: hipcc -O3 --offload-arch=gfx950 clock_probe.hip -o clock_probe.bin ; run: ./clock_probe.bin [seconds per arm]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: register operands; 1: LDS re-reads at 20 : 72
template <int MODE>
__global__ __launch_bounds__(512, 1) void probe(const _Float16 *__restrict__ src, float *out, long long *stamps, int iters) {
    extern __shared__ __attribute__((aligned(16))) _Float16 sm[];        // 64 KiB of operands
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += 512) reinterpret_cast<f16x8 *>(sm)[i] = reinterpret_cast<const f16x8 *>(src)[i];
    __syncthreads();
    if (wave >= 4) return;                 // 8 waves are launched so that a wave gets the conv kernel's 256 registers (no AGPR shuffling); 4 compute
    f16x8 a1[4], a2[4], b1[6], b2[6];
    f32x4 acc[4][6];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        a1[f] = reinterpret_cast<const f16x8 *>(sm)[(wave * 4 + f) * 64 + lane];
        a2[f] = reinterpret_cast<const f16x8 *>(sm)[1024 + (wave * 4 + f) * 64 + lane];
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        b1[r] = reinterpret_cast<const f16x8 *>(sm)[2048 + (wave * 6 + r) * 64 + lane];
        b2[r] = reinterpret_cast<const f16x8 *>(sm)[2048 + 1536 + ((wave * 6 + r) & 7) * 64 + lane];
    }
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int r = 0; r < 6; ++r) acc[f][r] = f32x4{0.f, 0.f, 0.f, 0.f};
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1) {
            const int o = it * 80;                // fresh operands every K step: the reads walk through the 64 KiB of random data
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                a1[f] = reinterpret_cast<const f16x8 *>(sm)[(((wave * 4 + f) * 64 + lane) + o) & 1023];
                a2[f] = reinterpret_cast<const f16x8 *>(sm)[1024 + ((((wave * 4 + f) * 64 + lane) + o) & 1023)];
            }
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                b1[r] = reinterpret_cast<const f16x8 *>(sm)[2048 + ((((wave * 6 + r) * 64 + lane) + o) & 1023)];
                b2[r] = reinterpret_cast<const f16x8 *>(sm)[3072 + ((((wave * 6 + r) * 64 + lane) + o) & 1023)];
            }
        }
        // one K step of the f16x3 pair form: wl * xh, wh * xl, wh * xh over the 4 x 6 tile = 72 MFMAs
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int r = 0; r < 6; ++r) acc[f][r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2[f], b1[r], acc[f][r], 0, 0, 0);
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int r = 0; r < 6; ++r) acc[f][r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[f], b2[r], acc[f][r], 0, 0, 0);
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int r = 0; r < 6; ++r) acc[f][r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[f], b1[r], acc[f][r], 0, 0, 0);
    }
    const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int r = 0; r < 6; ++r) s += acc[f][r][0] + acc[f][r][1] + acc[f][r][2] + acc[f][r][3];
    out[blockIdx.x * 256 + (threadIdx.x & 255)] = s;
    if (lane == 0) {
        stamps[(blockIdx.x * 4 + wave) * 2] = t1 - t0;
        stamps[(blockIdx.x * 4 + wave) * 2 + 1] = r1 - r0;
    }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>
static void arm(const char *name, const _Float16 *src, float seconds) {
    int cus = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    float *out;
    long long *stamps;
    CK(hipMalloc(&out, sizeof(float) * cus * 256));
    CK(hipMalloc(&stamps, sizeof(long long) * cus * 8));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    const int iters = 20000;                          // 72 MFMAs x 16 cycles x 20000 = 23 M cycles: ~12 ms per launch
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms = 0.f, total = 0.f;
    int launches = 0;
    while (total < seconds * 1e3f) {                  // back-to-back launches; the last one is the measurement
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(probe<MODE>, dim3(cus), dim3(512), 65536, 0, src, out, stamps, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        total += ms;
        ++launches;
    }
    std::vector<long long> st(cus * 8);
    CK(hipMemcpy(st.data(), stamps, sizeof(long long) * cus * 8, hipMemcpyDeviceToHost));
    std::vector<double> clk;
    for (int i = 0; i < cus * 4; ++i) clk.push_back((double)st[2 * i] / (double)st[2 * i + 1] * 100.0);
    std::sort(clk.begin(), clk.end());
    const double flops = (double)cus * 4 * iters * 72.0 * 16 * 16 * 32 * 2;
    const double cyc = (double)st[0];
    printf("%-11s %7.1f issued TFLOP/s (%6.1f f16x3-equivalent)   in-kernel clock %4.0f MHz (min %4.0f max %4.0f)   MFMA duty %.3f   [%d launches, last %.2f ms]\n",
           name, flops / ms * 1e-9, flops / ms * 1e-9 / 3.0, clk[clk.size() / 2], clk.front(), clk.back(), (double)iters * 72 * 16 / cyc, launches, ms);
    CK(hipFree(out));
    CK(hipFree(stamps));
}

int main(int argc, char **argv) {
    const float seconds = argc > 1 ? (float)atof(argv[1]) : 2.5f;
    std::vector<_Float16> h(32768), z(32768, (_Float16)0.f);
    unsigned s = 12345u;
    for (size_t i = 0; i < h.size(); i += 2) {        // (hi, lo) pairs of random fp32 values in [-4, 4): what the split feeds the MFMAs
        s = s * 1664525u + 1013904223u;
        const float x = ((int)(s >> 8) - (1 << 23)) * (4.0f / (1 << 23));
        const _Float16 hi = (_Float16)x;
        h[i] = hi;
        h[i + 1] = (_Float16)(x - (float)hi);
    }
    _Float16 *dz, *dr;
    CK(hipMalloc(&dz, 65536));
    CK(hipMalloc(&dr, 65536));
    CK(hipMemcpy(dz, z.data(), 65536, hipMemcpyHostToDevice));
    CK(hipMemcpy(dr, h.data(), 65536, hipMemcpyHostToDevice));
    arm<0>("zeros", dz, seconds);
    arm<0>("random-reg", dr, seconds);
    arm<1>("random-lds", dr, seconds);
    arm<1>("zeros-lds", dz, seconds);
    return 0;
}
