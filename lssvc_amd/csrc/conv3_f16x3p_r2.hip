// conv3_f16x3p_r2.hip -- round-6 instantiations of the persistent warp-specialised 3x3 kernel, second translation unit (see
// conv3_f16x3p_r.hip): the narrow heads, stride 2 with register prefetch / pair loads, and (one instantiation, an experiment) the 24x16 tiling with pair loads.
#include "conv3_f16x3p_kernel.h"

namespace lssvc {

int launch_p3_narrow(const ConvP &p, bool inact, bool flat, int pf, hipStream_t st) {
#define LSSVC_P3N_CASE(f)                                                                                                                   \
    if (pf == f) {                                                                                                                          \
        if (flat) return inact ? launch_p3r<1, true, 1, 4, f, true, true>(p, st) : launch_p3r<1, false, 1, 4, f, true, true>(p, st);        \
        return inact ? launch_p3r<1, true, 1, 4, f, false, true>(p, st) : launch_p3r<1, false, 1, 4, f, false, true>(p, st);                \
    }
    LSSVC_P3N_CASE(3) LSSVC_P3N_CASE(1) LSSVC_P3N_CASE(0)
#undef LSSVC_P3N_CASE
    return fail("conv2d(f16x3p narrow): prefetch mode %d", pf);
}

int launch_p3s2_pf(const ConvP &p, int mf, bool inact, int pf, hipStream_t st) {
#define LSSVC_P3S_CASE(m, f) \
    if (mf == m && pf == f) return inact ? launch_p3r<m, true, 2, 0, f, false, false>(p, st) : launch_p3r<m, false, 2, 0, f, false, false>(p, st);
    LSSVC_P3S_CASE(4, 1) LSSVC_P3S_CASE(4, 2) LSSVC_P3S_CASE(3, 1) LSSVC_P3S_CASE(3, 2) LSSVC_P3S_CASE(4, 3) LSSVC_P3S_CASE(3, 3)
#undef LSSVC_P3S_CASE
    return fail("conv2d(f16x3p, stride 2, prefetch %d): no kernel for MF=%d", pf, mf);
}

// (experiment, `p3_big_pair`: the 24x16 tiling with pair loads -- 2 ... 13 % SLOWER on every shape of the bench, profiles/r06_pair_loads_ab.txt;
// only the dominant instantiation is kept so that the A/B can be re-run)
// experiment (`p3_force` = MF * 16 + 8): 32x16-pixel tiles (8 rows per consumer wave). At MF = 3 the phase has the MFMA count of the
// dominant MF = 4 / 24x16 kernel (336) on a patch with a smaller halo share; two patch buffers instead of the ring's three.
int launch_p3_tall(const ConvP &p, int mf, bool inact, int pf, hipStream_t st) {
#define LSSVC_P3T_CASE(m, f) \
    if (mf == m && pf == f) return inact ? launch_p3r<m, true, 1, 8, f, false, false>(p, st) : launch_p3r<m, false, 1, 8, f, false, false>(p, st);
    LSSVC_P3T_CASE(3, 0) LSSVC_P3T_CASE(3, 1) LSSVC_P3T_CASE(4, 0)
#undef LSSVC_P3T_CASE
    return fail("conv2d(f16x3p, 32x16 tiles): no kernel for MF=%d pf=%d", mf, pf);
}

int launch_p3_big_pair(const ConvP &p, int mf, bool inact, hipStream_t st) {
    if (mf == 4 && !inact) return launch_p3r<4, false, 1, 0, 2, false, false>(p, st);
    return fail("conv2d(f16x3p, pair loads): only the MF = 4 instantiation without input activation is built");
}

}  // namespace lssvc
