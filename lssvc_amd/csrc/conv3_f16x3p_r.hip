// conv3_f16x3p_r.hip -- round-6 instantiations of the persistent warp-specialised 3x3 kernel (conv3_f16x3p_kernel.h):
//   small tiles  (RPWT = 1, 2, 4: 4x16, 8x16, 16x16 pixels) for the maps the 24x16 tiling cannot spread over 256 CUs,
//   narrow heads (MF = 1, 16x16 tiles, two workgroups per CU; with the tiled kernel's epilogue when Cout % 4 != 0),
//   stride 2 with the register prefetch / the pair loads (conv3_f16x3p_r2.hip holds the 24x16 pair-load instantiations).
// Same arithmetic per accumulator as every other f16x3 3x3 kernel: bit-identical results (tests/test_gpu_bench_kernels.py).
#include "conv3_f16x3p_kernel.h"

namespace lssvc {

int launch_p3_small(const ConvP &p, int mf, int rpw, bool inact, int pf, hipStream_t st) {
#define LSSVC_P3R_CASE(m, r)                                                                                              \
    if (mf == m && rpw == r) {                                                                                            \
        if (pf == 1) return inact ? launch_p3r<m, true, 1, r, 1, false, false>(p, st) : launch_p3r<m, false, 1, r, 1, false, false>(p, st);   \
        return inact ? launch_p3r<m, true, 1, r, 0, false, false>(p, st) : launch_p3r<m, false, 1, r, 0, false, false>(p, st);           \
    }
    LSSVC_P3R_CASE(4, 1) LSSVC_P3R_CASE(4, 2) LSSVC_P3R_CASE(4, 4)
    LSSVC_P3R_CASE(3, 1) LSSVC_P3R_CASE(3, 2) LSSVC_P3R_CASE(3, 4)
    LSSVC_P3R_CASE(2, 1) LSSVC_P3R_CASE(2, 2) LSSVC_P3R_CASE(2, 4)
    LSSVC_P3R_CASE(1, 1) LSSVC_P3R_CASE(1, 2) LSSVC_P3R_CASE(1, 4)
#undef LSSVC_P3R_CASE
    return fail("conv2d(f16x3p r): no kernel for MF=%d RPW=%d", mf, rpw);
}

}  // namespace lssvc
