// conv3_f16x3p.hip -- dispatch of the persistent warp-specialised 3x3 kernels (conv3_f16x3p_kernel.h) and the instantiations of
// rounds 2-5 (24x16-pixel tiles at stride 1, 8x16 at stride 2); the small-tile / narrow-output instantiations of round 6 live in
// conv3_f16x3p_r.hip so that the two translation units build in parallel.
#include "conv3_f16x3p_kernel.h"

namespace lssvc {

static int p3_pick_mf(int frags) {
    if (frags > 4 && frags % 4 != 0 && frags % 3 == 0) return 3;        // e.g. 96 = 2 x 48 rather than 64 + 32
    return frags >= 4 ? 4 : frags;
}

int p3_pick_mf_public(int frags) { return p3_pick_mf(frags); }

static long long p3_tiles(const ConvP &p, int mf, int rpw) {
    return (long long)((p.Wout + 15) / 16) * ((p.Hout + 4 * rpw - 1) / (4 * rpw)) * ((p.M_pad / 16 + mf - 1) / mf);
}

// ---- round 6: which tiling serves a 3x3 stride-1 conv. A launch lasts (tiles per CU, rounded UP) x (time of one tile), so a map that gives
// the 256 CUs 1.4 tiles of 24x16 each runs as long as one that gives them 2. Candidates: the 24x16 tiling of rounds 2-5 (rows per wave 6:
// the instantiations of this file) and the small-tile instantiations of conv3_f16x3p_r.hip (16 / 8 / 4 rows) with the M tilings MF, MF / 2
// (channel halves as separate tiles: twice the patch traffic, half the work per tile), costed with the consumers' MFMA cycles per phase
// plus per-phase and per-tile overheads fitted to profiles/r06_small_map_ab.txt. Maps with at least `f16x3_persist_min_tiles` (256) tiles
// of 24x16 keep that tiling (measured: no small tiling beats it there); below, only the small tilings are candidates, and none is taken
// unless it gives half the CUs a tile (the tiled kernel keeps the launch then).
struct P3Small {
    int mf, rpw;
};
static bool p3_pick_tiling(const ConvP &p, P3Small &c) {
    const int cus = device_cus(), frags = p.M_pad / 16, phases = p.n_chunks16, mf0 = p3_pick_mf(frags);
    const long long tiles24 = p3_tiles(p, mf0, LSSVC_P3_RPW);
    const bool big_ok = tiles24 >= option_get(OPT_P3_MIN_TILES);
    c.mf = mf0;
    c.rpw = LSSVC_P3_RPW;
    if (p.in_split || !option_get(OPT_P3_SMALL)) return big_ok;
    if (const int forced = option_get(OPT_P3_FORCE); forced > 0) {      // experiments: mf * 16 + rows per wave
        c.mf = (forced >> 4) & 15;
        c.rpw = forced & 15;
        return c.mf >= 1 && c.mf <= 4 && (c.rpw == 1 || c.rpw == 2 || c.rpw == 4 || (c.rpw == 8 && c.mf >= 3)) && c.mf <= frags;
    }
    if (big_ok) {
        // (measured: from 256 tiles of 24x16 on no SMALL tiling beats it -- profiles/r06_small_map_ab.txt, second table.) One TALLER tiling
        // does, for the 48-channel layers of the full-resolution maps: 32x16 tiles (8 rows per wave) give an MF = 3 phase the MFMA count of
        // the dominant MF = 4 / 24x16 kernel on a patch with a smaller halo share -- +2 ... +7 % where the map has >= 16 tiles per CU
        // (profiles/r06_tall_tiles_ab.txt; with few tiles per CU the coarser tiling loses to its own tail, -17 % at 288x480)
        if (option_get(OPT_P3_SMALL) == 1 && mf0 == 3 && frags == 3 && p3_tiles(p, 3, 8) >= 16LL * cus) c.rpw = 8;
        return true;
    }
    auto cost = [&](int mf, int rpw) {
        const long long tiles = p3_tiles(p, mf, rpw);
        const double tile_cycles = phases * (14.0 * 16.0 * mf * rpw + 900.0) + 150.0 * mf * rpw + 1500.0;
        return (double)((tiles + cus - 1) / cus) * tile_cycles;
    };
    double best = 0.0;
    bool found = false;
    for (int mf = 4; mf >= 1; --mf) {
        if (mf > frags || (frags % mf != 0 && mf != mf0)) continue;      // M tilings without a ragged last tile (or the default one)
        for (int rpw = 4; rpw >= 1; rpw >>= 1) {
            if (p3_tiles(p, mf, rpw) * 2 < cus) continue;
            const double t = cost(mf, rpw);
            if (!found || t < best) {
                best = t;
                c.mf = mf;
                c.rpw = rpw;
                found = true;
            }
        }
    }
    return found;
}

// Worth it once every CU gets at least one 32x16 tile (measured on the bench workload: thresholds 256 / 512 / 1024 /
// 2048 tiles give 14.0 / 13.9 / 13.6 / 13.5 frames/s). Round 6: below that the small-tile instantiations; convs with at most 16
// output channels (M_pad == 16) take the narrow instantiation (16x16 tiles, two workgroups per CU), whatever their epilogue.
static bool p3_narrow_wanted(const ConvP &p) {
    return option_get(OPT_P3_NARROW) && p.M_pad == 16 && !p.in_split && p3_tiles(p, 1, 4) >= option_get(OPT_P3_MIN_TILES) &&
           (p.fast_epi == 1 || (p.fast_epi == 0 && p.epilogue == LSSVC_EPI_NONE && !p.pixel_shuffle && !p.res2.p));
}
bool conv3_f16x3p_wanted(const ConvP &p) {
    const int on = option_get(OPT_P3_ON), min_tiles = option_get(OPT_P3_MIN_TILES);     // shared by both persistent kernels
    if (!on) return false;
    if (p.in_act == LSSVC_INACT_LRELU && !(p.in_slope >= 0.0f && p.in_slope <= 1.0f)) return false;   // max(x, s*x) form
    if (p3_narrow_wanted(p)) return true;
    if (!p.fast_epi) return false;
    (void)min_tiles;
    P3Small c;
    return p3_pick_tiling(p, c);
}

// stride 2: 8x16-pixel output tiles (P3Geom<MF, 2>); MF >= 3 only (narrower outputs leave the consumers one or two fragments
// of work per 33x17-pixel patch, the tiled kernel's two workgroups per CU do as well there)
bool conv3s2_f16x3p_wanted(const ConvP &p) {
    const int on = option_get(OPT_P3_S2), min_tiles = option_get(OPT_P3_MIN_TILES);
    if (!on || !p.fast_epi) return false;
    if (p.in_act == LSSVC_INACT_LRELU && !(p.in_slope >= 0.0f && p.in_slope <= 1.0f)) return false;
    const int mf = p3_pick_mf(p.M_pad / 16);
    if (mf < 3) return false;
    const long long ntiles = (long long)((p.Wout + 15) / 16) * ((p.Hout + 7) / 8) * ((p.M_pad / 16 + mf - 1) / mf);
    return ntiles >= min_tiles;
}

int dispatch_conv3s2_f16x3p(const ConvP &p, hipStream_t st, char *kernel_name) {
    const int mf = p3_pick_mf(p.M_pad / 16);
    const bool inact = p.in_act == LSSVC_INACT_LRELU;
    snprintf(kernel_name, 96, "conv3s2_f16x3p_kernel<%d, %s>%s", mf, inact ? "true" : "false", p.in_split ? " split" : "");
    if (p.in_split) {
        if (mf == 4) return launch_p3<4, false, 2, true>(p, st);
        if (mf == 3) return launch_p3<3, false, 2, true>(p, st);
        return fail("conv2d(f16x3p, stride 2, split in): no kernel for MF=%d", mf);
    }
    // round 6 producer schedules (conv3_f16x3p_kernel.h), option p3_pf2: 0 = round 5's one register set; 1 (default) = split roles (one producer
    // wave owns the weight DMA, three stage the patch through two register sets) up to five 16-channel phases per tile, the register prefetch
    // from six on (128-channel inputs: the prefetch is +13 %, the roles +7 %; profiles/r06_roles_ab.txt); 2 = pair loads; 3 = roles always;
    // 4 = register prefetch always. The stamp build has the plain schedule only.
    if (const int opt = option_get(OPT_P3_PF2); opt && !(p.debug & 256)) {
        const int pf = opt == 2 ? 2 : opt == 3 ? 3 : opt == 4 ? 1 : (p.n_chunks16 <= 5 ? 3 : 1);
        snprintf(kernel_name, 96, "conv3s2_f16x3p_kernel<%d, %s> %s", mf, inact ? "true" : "false", pf == 2 ? "pair" : pf == 3 ? "roles" : "pf2");
        return launch_p3s2_pf(p, mf, inact, pf, st);
    }
    if (mf == 4) return inact ? launch_p3<4, true, 2>(p, st) : launch_p3<4, false, 2>(p, st);
    if (mf == 3) return inact ? launch_p3<3, true, 2>(p, st) : launch_p3<3, false, 2>(p, st);
    return fail("conv2d(f16x3p, stride 2): no kernel for MF=%d", mf);
}

int dispatch_conv3_f16x3p(const ConvP &p, hipStream_t st, char *kernel_name) {
    const int mf = p3_pick_mf(p.M_pad / 16);
    const bool inact = p.in_act == LSSVC_INACT_LRELU;
    snprintf(kernel_name, 96, "conv3_f16x3p_kernel<%d, %s>%s", mf, inact ? "true" : "false", p.in_split ? " split" : "");
    if (p.in_split) {
#define LSSVC_P3_CASE(m) \
    if (mf == m) return launch_p3<m, false, 1, true>(p, st);
        LSSVC_P3_CASE(4) LSSVC_P3_CASE(3) LSSVC_P3_CASE(2) LSSVC_P3_CASE(1)
#undef LSSVC_P3_CASE
    }
    if (p3_narrow_wanted(p) && !(p.debug & 256)) {      // (the stamp build, LSSVC_CONV_DEBUG = 256, has the 24x16 kernel's schedule only)
        const int opt = option_get(OPT_P3_PF2);
        const int pf = !opt ? 0 : (opt == 4 || opt == 2) ? 1 : 3;      // split roles: +5 ... +8 % over the register prefetch on these (profiles/r06_roles_ab.txt); pair loads are not built for them (slower)
        snprintf(kernel_name, 96, "conv3n_f16x3p_kernel<%s, %s>%s", inact ? "true" : "false", p.fast_epi ? "fast" : "flat", pf == 3 ? " roles" : pf == 1 ? " pf2" : "");
        return launch_p3_narrow(p, inact, !p.fast_epi, pf, st);
    }
    if (P3Small c; !p.in_split && !(p.debug & 256) && p3_pick_tiling(p, c) && c.rpw != LSSVC_P3_RPW) {
        // p3_small: 1 = the register prefetch from 8 phases on (+3 ... +8 %; below that it is neutral: profiles/r06_small_map_ab.txt), 2 = always,
        // 3 = never
        const int sm = option_get(OPT_P3_SMALL);
        int pf = (sm == 2 || (sm == 1 && p.n_chunks16 >= 8)) ? 1 : 0;
        if (c.rpw == 8 && sm == 1) pf = p.n_chunks16 % 3 == 0 ? 1 : 0;      // 32x16 tiles: the prefetch pays with 3 or 6 phases per tile (48- / 96-channel inputs), not with 4 or 5
        snprintf(kernel_name, 96, "conv3r_f16x3p_kernel<%d, %s, rpw %d%s>", c.mf, inact ? "true" : "false", c.rpw, pf == 2 ? ", pair" : pf == 1 ? ", pf2" : "");
        // split roles on the 32x16 tiling: +1 ... +3 % on the 48 -> 48 layers (three phases per tile), -1 % with six; on the 24x16 tilings
        // -2 ... -6 % at MF = 4 (profiles/r06_big_roles_ab.txt). p3_big_pair: 0 = this rule, 2 = roles wherever built, 3 = nowhere
        if (const int bp = option_get(OPT_P3_BIG_PAIR); c.rpw == 8 && c.mf == 3 && (bp == 2 || (bp == 0 && p.n_chunks16 == 3 && sm == 1))) {
            snprintf(kernel_name, 96, "conv3r_f16x3p_kernel<%d, %s, rpw %d, roles>", c.mf, inact ? "true" : "false", c.rpw);
            return launch_p3_big_roles(p, c.mf, 8, inact, st);
        }
        if (const int bp = option_get(OPT_P3_BIG_PAIR); c.rpw == 8 && c.mf == 3 && (bp == 4 || (bp == 0 && p.n_chunks16 == 6 && !inact && sm == 1))) {      // (96 -> 48: +4 %)
            snprintf(kernel_name, 96, "conv3r_f16x3p_kernel<%d, %s, rpw %d, late>", c.mf, inact ? "true" : "false", c.rpw);
            return launch_p3_big_late(p, c.mf, 8, inact, st);
        }
        if (c.rpw == 8) return launch_p3_tall(p, c.mf, inact, c.mf == 3 ? pf : 0, st);
        return launch_p3_small(p, c.mf, c.rpw, inact, pf, st);
    }
    // experiment: late loads (conv3_f16x3p_kernel.h, PF = 4) on the 24x16 tiling: -2 ... +3 % at MF = 4 without an input activation (two boxes:
    // +1 % on average, inside the noise on the 576x960 layers), -5 % with one (profiles/r06_late_loads_ab.txt): not a default here
    if (option_get(OPT_P3_BIG_PAIR) == 4 && mf == 4 && !inact && (p.debug & 256)) {      // the stamp build of the late-loads schedule
        snprintf(kernel_name, 96, "conv3_f16x3p_kernel<%d, %s> late, stamps", mf, inact ? "true" : "false");
        return launch_p3_late_stamps(p, st);
    }
    if (option_get(OPT_P3_BIG_PAIR) == 4 && mf >= 2 && !(p.debug & 256)) {
        snprintf(kernel_name, 96, "conv3_f16x3p_kernel<%d, %s> late", mf, inact ? "true" : "false");
        return launch_p3_big_late(p, mf, LSSVC_P3_RPW, inact, st);
    }
    if (option_get(OPT_P3_BIG_PAIR) == 2 && mf >= 2 && !(p.debug & 256)) {      // experiment: the 24x16 tiling with split roles
        snprintf(kernel_name, 96, "conv3_f16x3p_kernel<%d, %s> roles", mf, inact ? "true" : "false");
        return launch_p3_big_roles(p, mf, LSSVC_P3_RPW, inact, st);
    }
    if (option_get(OPT_P3_BIG_PAIR) == 1 && mf == 4 && !inact && !(p.debug & 256)) {      // experiment: the 24x16 tiling with pair loads
        snprintf(kernel_name, 96, "conv3_f16x3p_kernel<%d, %s> pair", mf, inact ? "true" : "false");
        return launch_p3_big_pair(p, mf, inact, st);
    }
#define LSSVC_P3_CASE(m) \
    if (mf == m) return inact ? launch_p3<m, true>(p, st) : launch_p3<m, false>(p, st);
    LSSVC_P3_CASE(4) LSSVC_P3_CASE(3) LSSVC_P3_CASE(2) LSSVC_P3_CASE(1)
#undef LSSVC_P3_CASE
    return fail("conv2d(f16x3p): no kernel for MF=%d", mf);
}

}  // namespace lssvc
