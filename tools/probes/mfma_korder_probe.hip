// Is v_mfma_f32_16x16x32_f16 invariant under a permutation of its K positions (the same permutation applied to A and B)?
// If the 32 products of a dot product are summed exactly and rounded once, yes; if they are summed in a fixed tree with intermediate
// roundings, no. Decides whether a GDN fused into a conv epilogue (accumulator fragments used as B operands: a K-permuted layout, as in
// ffn_f16x3.hip) can be BIT-identical to the GDN 1x1 kernel (natural K order).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_korder_probe.hip -o /tmp/korder && /tmp/korder
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// A: [16 rows][32 k], B: [32 k][16 cols] as fp16; perm: k' = perm[k]. Lane (i = l & 15, g = l >> 4) supplies k = 8 g .. 8 g + 7.
__global__ void probe(const _Float16 *A, const _Float16 *B, const int *perm, const float *C0, float *D, int trials) {
    const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
    for (int t = blockIdx.x; t < trials; t += gridDim.x) {
        const _Float16 *a = A + (size_t)t * 16 * 32, *b = B + (size_t)t * 32 * 16;
        f16x8 fa, fb, pa, pb;
        for (int j = 0; j < 8; ++j) {
            const int k = 8 * g + j, kp = perm[k];
            fa[j] = a[i * 32 + k];  fb[j] = b[k * 16 + i];
            pa[j] = a[i * 32 + kp]; pb[j] = b[kp * 16 + i];
        }
        f32x4 c = {C0[(size_t)t * 256 + (4 * g + 0) * 16 + i], C0[(size_t)t * 256 + (4 * g + 1) * 16 + i], C0[(size_t)t * 256 + (4 * g + 2) * 16 + i], C0[(size_t)t * 256 + (4 * g + 3) * 16 + i]};
        const f32x4 d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, c, 0, 0, 0);
        const f32x4 d2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(pa, pb, c, 0, 0, 0);
        for (int j = 0; j < 4; ++j) {
            D[((size_t)t * 2 + 0) * 256 + (4 * g + j) * 16 + i] = d1[j];
            D[((size_t)t * 2 + 1) * 256 + (4 * g + j) * 16 + i] = d2[j];
        }
    }
}

int main() {
    const int trials = 20000;
    std::vector<_Float16> A((size_t)trials * 512), B((size_t)trials * 512);
    std::vector<float> C0((size_t)trials * 256);
    srand(1);
    auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    for (int t = 0; t < trials; ++t) {
        const int mode = t % 4;          // 0: O(1) values; 1: wide exponent spread; 2: hi/lo-like (A small, B O(100)); 3: squares-like (B >= 0)
        for (int e = 0; e < 512; ++e) {
            float a = rnd(), b = rnd();
            if (mode == 1) { a *= powf(2.f, (float)(rand() % 20 - 10)); b *= powf(2.f, (float)(rand() % 20 - 10)); }
            if (mode == 2) { a *= 1e-3f; b *= 100.f; }
            if (mode == 3) { b = b * b * 50.f; }
            A[(size_t)t * 512 + e] = (_Float16)a; B[(size_t)t * 512 + e] = (_Float16)b;
        }
        for (int e = 0; e < 256; ++e) C0[(size_t)t * 256 + e] = (t & 8) ? rnd() * 10.f : 0.f;
    }
    // permutations: the FFN / accumulator-fragment order (k = 8 g + j' -> 16 * (j' < 4 ? 0 : 1) + 4 g + (j' & 3)), a reversal, a random one
    std::vector<std::vector<int>> perms(3, std::vector<int>(32));
    for (int k = 0; k < 32; ++k) { const int g = k >> 3, j = k & 7; perms[0][k] = 16 * (j < 4 ? 0 : 1) + 4 * g + (j & 3); perms[1][k] = 31 - k; perms[2][k] = k; }
    for (int k = 31; k > 0; --k) std::swap(perms[2][k], perms[2][rand() % (k + 1)]);
    _Float16 *dA, *dB; int *dP; float *dC, *dD;
    hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dP, 128); hipMalloc(&dC, C0.size() * 4); hipMalloc(&dD, (size_t)trials * 512 * 4);
    hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dC, C0.data(), C0.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> D((size_t)trials * 512);
    const char *names[3] = {"accumulator-fragment (FFN) order", "reversed", "random"};
    for (int p = 0; p < 3; ++p) {
        hipMemcpy(dP, perms[p].data(), 128, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(256), dim3(64), 0, 0, dA, dB, dP, dC, dD, trials);
        hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
        long long diff = 0, total = 0; double worst = 0;
        for (int t = 0; t < trials; ++t)
            for (int e = 0; e < 256; ++e) {
                const float x = D[((size_t)t * 2) * 256 + e], y = D[((size_t)t * 2 + 1) * 256 + e];
                ++total;
                if (memcmp(&x, &y, 4)) { ++diff; const double r = fabs((double)x - y) / (fabs((double)x) + 1e-30); if (r > worst) worst = r; }
            }
        printf("K permutation %-34s: %lld of %lld results differ (worst relative difference %.3g)\n", names[p], diff, total, worst);
    }
    return 0;
}
