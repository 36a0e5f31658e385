// conv3_f16x3p_r3.hip -- round-6 instantiations of the persistent warp-specialised 3x3 kernel, third translation unit: the split-roles producer
// schedule (PF = 3: one producer wave owns the weight DMA, three stage the patch through two register sets) on the big tilings -- 24x16
// pixels at MF = 4 / 3 / 2 and 32x16 at MF = 3 -- selected with option p3_big_pair = 2.
#include "conv3_f16x3p_kernel.h"

namespace lssvc {

int launch_p3_big_roles(const ConvP &p, int mf, int rpw, bool inact, hipStream_t st) {
#define LSSVC_P3B_CASE(m, r) \
    if (mf == m && rpw == (r ? r : LSSVC_P3_RPW)) return inact ? launch_p3r<m, true, 1, r, 3, false, false>(p, st) : launch_p3r<m, false, 1, r, 3, false, false>(p, st);
    LSSVC_P3B_CASE(4, 0) LSSVC_P3B_CASE(3, 0) LSSVC_P3B_CASE(2, 0) LSSVC_P3B_CASE(3, 8)
#undef LSSVC_P3B_CASE
    return fail("conv2d(f16x3p, split roles): no kernel for MF=%d, %d rows per wave", mf, rpw);
}

// the late-loads schedule (PF = 4: the next patch requested right after a fill is signalled, conv3_f16x3p_kernel.h) on the same tilings
int launch_p3_big_late(const ConvP &p, int mf, int rpw, bool inact, hipStream_t st) {
#define LSSVC_P3L_CASE(m, r) \
    if (mf == m && rpw == (r ? r : LSSVC_P3_RPW)) return inact ? launch_p3r<m, true, 1, r, 4, false, false>(p, st) : launch_p3r<m, false, 1, r, 4, false, false>(p, st);
    LSSVC_P3L_CASE(4, 0) LSSVC_P3L_CASE(3, 0) LSSVC_P3L_CASE(2, 0) LSSVC_P3L_CASE(3, 8)
#undef LSSVC_P3L_CASE
    return fail("conv2d(f16x3p, late loads): no kernel for MF=%d, %d rows per wave", mf, rpw);
}

// diagnostic (LSSVC_CONV_DEBUG = 256 with p3_big_pair = 4; tools/p3_stamps.py): the late-loads schedule with in-kernel stamps
int launch_p3_late_stamps(const ConvP &p, hipStream_t st) { return launch_p3r<4, false, 1, 0, 4, false, false, true>(p, st); }

}  // namespace lssvc
