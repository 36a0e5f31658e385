// store_probe.hip -- how fast can 4 waves of a CU issue the epilogue's global stores?  (diagnostic, not part of the library)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/store_probe tools/probes/store_probe.hip && /tmp/store_probe
// 256 workgroups x 4 waves; each wave writes TILES x 24 x 1 KiB with one of the patterns:
//   0: the conv epilogue's (16 pixels x 64-byte channel quads per instruction, pixel stride 256 B; the four fragments of a
//      row follow each other, so a pixel's 256 B are completed by 4 consecutive instructions)
//   1: wave-contiguous 1 KiB per instruction
//   2: 4 pixels x 256 B per instruction (what a lane transpose would give)
//   3: pattern 0 but 8-byte stores (2 instructions of 512 B)
// with `gap` s_sleep(127)s between tiles (0 = back to back) to mimic the MFMA phases.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PAT, int MAP = 0, int REUSE = 0>
__global__ __launch_bounds__(256, 1) void probe(float *out, int tiles, int gap, long long *cycles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int W = 1920, C = 64;
    f32x4 v = {1.f * lane, 2.f, 3.f, 4.f};
    const int xcd = blockIdx.x % 8, kb = blockIdx.x / 8;
    long long t_epi = 0;
    for (int t = 0; t < tiles; ++t) {
        // MAP 1: the conv kernel's mapping (an XCD owns a contiguous range of 720 tiles, its 32 workgroups interleave inside it)
        const int tile = MAP ? xcd * 720 + kb + t * 32 : blockIdx.x + t * gridDim.x;          // 120 tiles per row of 16-wide tiles
        const int ty = tile / 120, tx = tile % 120;
        const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const int oy = ty * 24 + wave * 6 + r;
            float *row = out + ((size_t)oy * W + tx * 16) * C;
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                if (PAT == 0) *reinterpret_cast<f32x4 *>(row + li * C + f * 16 + lg * 4) = v;
                if (PAT == 4) __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(row + li * C + f * 16 + lg * 4));
                if (PAT == 5) {
                    float *q = row + li * C + f * 16 + lg * 4;
                    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(q), "v"(v) : "memory");
                }
                if (PAT == 6) {
                    float *q = row + li * C + f * 16 + lg * 4;
                    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(q), "v"(v) : "memory");
                }
                if (PAT == 7) {
                    float *q = row + li * C + f * 16 + lg * 4;
                    asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(q), "v"(v) : "memory");
                }
                if (PAT == 8)      // 8 pixels x 128-byte lines per instruction (DPP half-row swap of two fragments)
                    *reinterpret_cast<f32x4 *>(row + ((li & 7) + 8 * (f & 1)) * C + (f >> 1) * 32 + ((li >> 3) * 4 + lg) * 4) = v;
                if (PAT == 11)     // 4 consecutive lanes = one pixel's 64 B of fragment f
                    *reinterpret_cast<f32x4 *>(row + (lane >> 2) * C + f * 16 + (lane & 3) * 4) = v;
                if (PAT == 12)     // 8 consecutive lanes = 128 B
                    *reinterpret_cast<f32x4 *>(row + ((lane >> 3) + 8 * (f & 1)) * C + (f >> 1) * 32 + (lane & 7) * 4) = v;
                if (PAT == 9)      // 48-channel pixels (192 B): the epilogue pattern, 3 fragments
                    if (f < 3) *reinterpret_cast<f32x4 *>(row + li * 48 + f * 16 + lg * 4) = v;
                if (PAT == 10)     // 48-channel pixels, wave-contiguous
                    if (f < 3) *reinterpret_cast<f32x4 *>(row + f * 256 + lane * 4) = v;
                if (REUSE) v = v * 1.5f + 1.0f;                 // the next store's data overwrites this one's registers
                if (PAT == 1) *reinterpret_cast<f32x4 *>(row + f * 256 + lane * 4) = v;
                if (PAT == 2) *reinterpret_cast<f32x4 *>(row + (f * 4 + lg) * C + li * 4) = v;
                if (PAT == 3) {
                    *reinterpret_cast<float2 *>(row + li * C + f * 16 + lg * 4) = float2{v.x, v.y};
                    *reinterpret_cast<float2 *>(row + li * C + f * 16 + lg * 4 + 2) = float2{v.z, v.w};
                }
            }
        }
        t_epi += __builtin_amdgcn_s_memtime() - t0;
        for (int g = 0; g < gap; ++g) __builtin_amdgcn_s_sleep(127);
    }
    if (lane == 0) cycles[blockIdx.x * 4 + wave] = t_epi;
}

int main(int argc, char **argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 256;
    const size_t n = (size_t)1152 * 1920 * 64;
    float *out;
    long long *cyc, host[1024];
    hipMalloc(&out, n * 4);
    hipMalloc(&cyc, 1024 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int tiles = 22;
    for (int gap = 0; gap <= 0; gap += 4)
        for (int pat = 0; pat < 13; ++pat) {
            float best = 1e9;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                if (pat == 0) probe<0><<<grid, 256>>>(out, tiles, gap, cyc);
                if (pat == 1) probe<1><<<grid, 256>>>(out, tiles, gap, cyc);
                if (pat == 2) probe<2><<<grid, 256>>>(out, tiles, gap, cyc);
                if (pat == 3) probe<3><<<grid, 256>>>(out, tiles, gap, cyc);
                if (pat == 4) probe<4><<<grid, 256>>>(out, tiles, gap, cyc);
                if (pat == 5) probe<5><<<grid, 256>>>(out, tiles, gap, cyc);
                if (pat == 6) probe<6><<<grid, 256>>>(out, tiles, gap, cyc);
                if (pat == 7) probe<7><<<grid, 256>>>(out, tiles, gap, cyc);
                if (pat == 8) probe<8><<<grid, 256>>>(out, tiles, gap, cyc);
                if (pat == 9) probe<9><<<grid, 256>>>(out, tiles, gap, cyc);
                if (pat == 10) probe<10><<<grid, 256>>>(out, tiles, gap, cyc);
                if (pat == 11) probe<11><<<grid, 256>>>(out, tiles, gap, cyc);
                if (pat == 12) probe<12><<<grid, 256>>>(out, tiles, gap, cyc);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            hipMemcpy(host, cyc, sizeof(host), hipMemcpyDeviceToHost);
            double s = 0;
            for (int i = 0; i < grid * 4; ++i) s += host[i];
            const double per_tile = s / (grid * 4) / tiles;         // s_memtime counts shader clocks
            static const char *names[] = {"epilogue pattern", "wave-contiguous 1 KiB", "4 pixels x 256 B", "8-byte stores", "epilogue pattern, nontemporal",
                                          "epilogue pattern, sc0 sc1", "epilogue pattern, sc1", "epilogue pattern, sc0", "8 pixels x 128 B", "48-ch epilogue pattern (72 KiB/tile)", "48-ch contiguous (72 KiB/tile)", "4 consecutive lanes = 64 B", "8 consecutive lanes = 128 B"};
            printf("grid %d gap %d  %-40s: %.1f us for %.0f MB = %.2f TB/s; store section %.0f cycles/tile = %.1f B/clk/CU\n", grid, gap, names[pat],
                   best * 1e3, (double)grid * tiles * 98304 / 1e6, (double)grid * tiles * 98304 / best / 1e9, per_tile, 98304.0 / per_tile);
        }
    return 0;
}
