// instantiations + tile choice of the streaming 1x1 conv kernel (f16x3 mode)
#include <cstdlib>

#include "conv_pw_f16x3_kernel.h"
namespace lssvc {
int dispatch_pw_f16x3(const ConvP &p, hipStream_t st, char *kernel_name) {
    static const int rpw = getenv("LSSVC_PW_RPW") ? atoi(getenv("LSSVC_PW_RPW")) : 2;
    const int frags = p.M_pad / 16;
    const int nslot = ((p.n_chunks16 + 1) / 2) * 2;
    int mf_fit = kPwMaxLds / (nslot * 1024);          // each 16-channel M fragment costs nslot KiB of LDS (hi + lo planes)
    if (mf_fit < 1 && kPwBigLds / (nslot * 1024) < 1) return fail("conv2d(pw f16x3): Cin too large for the LDS-resident weight tile");
    int MF = frags < 4 ? frags : 4;
    if (frags % 4 != 0 && frags % 3 == 0) MF = 3;
    if (MF > mf_fit) {
        // K is large: a bigger M tile (one workgroup per CU, up to 144 KB of weights) halves the re-reads of X from L2
        static const int big = getenv("LSSVC_PW_BIG_LDS") ? atoi(getenv("LSSVC_PW_BIG_LDS")) : 0;   // measured neutral on the bench workload
        const int mf_big = kPwBigLds / (nslot * 1024);
        int want = MF;
        MF = mf_fit;
        if (big && mf_big > mf_fit) MF = mf_big < want ? mf_big : want;
    }
    if (p.epilogue != LSSVC_EPI_NONE || p.in_act == LSSVC_INACT_SQUARE) {
        // GDN / IGDN (square input, normalising epilogue). Up to 64 channels the whole gamma matrix fits LDS: the all-M kernel
        // (pixels converted once, next group's loads in flight during the M loop) with the GDN epilogue compiled in; the
        // generic streaming kernel has no cross-group prefetch and ran these at 1.3-1.8 TB/s
        static const int gdn_allm = getenv("LSSVC_GDN_ALLM") ? atoi(getenv("LSSVC_GDN_ALLM")) : 1;
        {
            int mf = frags < 4 ? frags : 4;
            if (frags % 4 != 0 && frags % 3 == 0) mf = 3;
            const int m_tiles = (frags + mf - 1) / mf;
            if (gdn_allm && p.n_chunks16 <= 4 && (long long)nslot * m_tiles * mf * 1024 <= kPwMaxLds) {
                snprintf(kernel_name, 96, "conv_pw_allm_f16x3_kernel<%d, 2, true>", mf);
                if (mf == 4) return launch_pw_allm_f16x3<4, 2, true>(p, st);
                if (mf == 3) return launch_pw_allm_f16x3<3, 2, true>(p, st);
                if (mf == 2) return launch_pw_allm_f16x3<2, 2, true>(p, st);
                return launch_pw_allm_f16x3<1, 2, true>(p, st);
            }
        }
        if (MF > mf_fit) MF = mf_fit;
        if (MF < 1) return fail("conv2d(pw f16x3): Cin too large for the LDS-resident weight tile");
        snprintf(kernel_name, 96, "conv_pw_f16x3_kernel<%d, 2, true>", MF);
        if (MF == 4) return launch_pw_f16x3<4, 2, true>(p, st);
        if (MF == 3) return launch_pw_f16x3<3, 2, true>(p, st);
        if (MF == 2) return launch_pw_f16x3<2, 2, true>(p, st);
        return launch_pw_f16x3<1, 2, true>(p, st);
    }
    static const int allm = getenv("LSSVC_PW_ALLM") ? atoi(getenv("LSSVC_PW_ALLM")) : 1;
    {   // small K and the whole weight matrix in LDS: convert the pixels once, loop the M tiles inside the wave
        int mf = frags < 4 ? frags : 4;
        if (frags % 4 != 0 && frags % 3 == 0) mf = 3;
        const int m_tiles = (frags + mf - 1) / mf;
        if (allm && p.n_chunks16 <= 4 && (long long)nslot * m_tiles * mf * 1024 <= kPwMaxLds) {
            snprintf(kernel_name, 96, "conv_pw_allm_f16x3_kernel<%d, 2>", mf);
            if (mf == 4) return launch_pw_allm_f16x3<4, 2>(p, st);
            if (mf == 3) return launch_pw_allm_f16x3<3, 2>(p, st);
            if (mf == 2) return launch_pw_allm_f16x3<2, 2>(p, st);
            return launch_pw_allm_f16x3<1, 2>(p, st);
        }
    }
    // large K on a small map: K-sliced kernel with a full-height M tile (X is re-read M/64 instead of M/16 times)
    static const int ksliced = getenv("LSSVC_PW_KSLICED") ? atoi(getenv("LSSVC_PW_KSLICED")) : 1;
    {
        int mfw = frags < 4 ? frags : 4;
        if (frags % 4 != 0 && frags % 3 == 0) mfw = 3;
        const long long groups64 = ((long long)p.Hout * p.Wout + 63) / 64;
        if (ksliced && p.n_chunks16 >= 16 && mfw > mf_fit && groups64 * ((frags + mfw - 1) / mfw) <= 4096 &&
            p.in_act != LSSVC_INACT_SQUARE && (p.in_act != LSSVC_INACT_LRELU || (p.in_slope >= 0.0f && p.in_slope <= 1.0f))) {
            snprintf(kernel_name, 96, "conv_pwks_f16x3_kernel<%d>", mfw);
            if (mfw == 4) return launch_pwks_f16x3<4>(p, st);
            if (mfw == 3) return launch_pwks_f16x3<3>(p, st);
            if (mfw == 2) return launch_pwks_f16x3<2>(p, st);
            return launch_pwks_f16x3<1>(p, st);
        }
    }
    static const int deepk = getenv("LSSVC_PW_DEEPK") ? atoi(getenv("LSSVC_PW_DEEPK")) : 1;
    if (deepk && p.n_chunks16 >= 8 && MF <= mf_fit && p.in_act != LSSVC_INACT_SQUARE &&
        (p.in_act != LSSVC_INACT_LRELU || (p.in_slope >= 0.0f && p.in_slope <= 1.0f))) {
        snprintf(kernel_name, 96, "conv_pwk_f16x3_kernel<%d, 2>", MF);
        if (MF == 4) return launch_pwk_f16x3<4, 2>(p, st);
        if (MF == 3) return launch_pwk_f16x3<3, 2>(p, st);
        if (MF == 2) return launch_pwk_f16x3<2, 2>(p, st);
        return launch_pwk_f16x3<1, 2>(p, st);
    }
    snprintf(kernel_name, 96, "conv_pw_f16x3_kernel<%d, %d>", MF, rpw);
    if (rpw == 1) {
        if (MF == 4) return launch_pw_f16x3<4, 1>(p, st);
        if (MF == 3) return launch_pw_f16x3<3, 1>(p, st);
        if (MF == 2) return launch_pw_f16x3<2, 1>(p, st);
        return launch_pw_f16x3<1, 1>(p, st);
    }
    if (MF == 4) return launch_pw_f16x3<4, 2>(p, st);
    if (MF == 3) return launch_pw_f16x3<3, 2>(p, st);
    if (MF == 2) return launch_pw_f16x3<2, 2>(p, st);
    return launch_pw_f16x3<1, 2>(p, st);
}
}  // namespace lssvc
