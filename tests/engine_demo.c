/* engine_demo.c -- a caller of the engine entry points with no Python, no PyTorch and no HIP code of its own: plain C against
 * include/lssvc_hip.h, linked with liblssvc_hip.so. It codes a clip frame by frame exactly as the reference's test.py loop
 * does (test.py:212-250): I-frame through lssvc_engine_iframe, P-frames through lssvc_engine_pframe with the DPB handed
 * back by the caller after clamping the two reconstructions to [0, 1]. All buffers here are host memory (the engine accepts
 * host or device pointers).
 *
 *   engine_demo <intra.ckpt> <inter.ckpt> <iframe.plan> <first_p.plan> <steady_p.plan> <case.bin> <out.bin>
 *               [<bl_first.plan> <bl_steady.plan> <el_first.plan> <el_steady.plan>]
 *
 * With the four layer plans (round 6; plan_compiler.compile_pframe_layers) the P-frames go through lssvc_engine_pframe_lookahead: the
 * loop has all frames at hand, as test.py's has, so it names frame t+1's base-layer input in the call of frame t and the engine codes
 * that base layer on its second stream beside frame t's enhancement layer. Same out.bin, bit for bit.
 *
 * *.ckpt: the two RAW checkpoints (tests/ckpt_blob.h: the reference's state dicts dumped tensor by tensor, OIHW fp32, no
 *         re-layout); the engine prepares its device weights from them (lssvc_engine_load_checkpoint) and the plan files hold
 *         launches only.
 * case.bin: int32 n_frames, H, W, h, w; float scale; then per frame x_bl (3*h*w floats) and x_el (3*H*W floats), NCHW.
 * out.bin:  per frame: double bit_bl, bit_el; then recon_bl (3hw), recon_el (3HW), feature_el (Cf*H*W, Cf = 64 for the
 *           I-frame, 48 for P-frames), and for P-frames feature_bl (64hw), mv_hat (2HW), warp_frame (3HW); un-clamped.
 * tests/test_gpu_engine.py builds the plans and the case with the Python front end, runs this program and compares. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lssvc_hip.h"
#include "ckpt_blob.h"

static void die(const char *what) {
    fprintf(stderr, "engine_demo: %s: %s\n", what, lssvc_last_error());
    exit(1);
}

static float *alloc_f(size_t n) {
    float *p = (float *)malloc(n * sizeof(float));
    if (!p) {
        fprintf(stderr, "engine_demo: out of memory\n");
        exit(1);
    }
    return p;
}

static void clamp01(float *x, size_t n) {
    for (size_t i = 0; i < n; ++i) x[i] = x[i] < 0.f ? 0.f : (x[i] > 1.f ? 1.f : x[i]);
}

int main(int argc, char **argv) {
    if (argc != 8 && argc != 12) {
        fprintf(stderr, "usage: engine_demo <intra.ckpt> <inter.ckpt> <iframe.plan> <first_p.plan> <steady_p.plan> <case.bin> <out.bin> "
                        "[<bl_first.plan> <bl_steady.plan> <el_first.plan> <el_steady.plan>]\n");
        return 2;
    }
    const int lookahead = argc == 12;
    const char *ckpt_i = argv[1], *ckpt_p = argv[2];
    argv += 2;
    FILE *in = fopen(argv[4], "rb"), *out = fopen(argv[5], "wb");
    if (!in || !out) {
        fprintf(stderr, "engine_demo: cannot open %s / %s\n", argv[4], argv[5]);
        return 2;
    }
    int32_t hdr[5];
    float scale;
    if (fread(hdr, 4, 5, in) != 5 || fread(&scale, 4, 1, in) != 1) return 2;
    const int n_frames = hdr[0], H = hdr[1], W = hdr[2], h = hdr[3], w = hdr[4];
    const size_t el = (size_t)H * W, bl = (size_t)h * w;

    void *eng = lssvc_engine_create(0);
    if (!eng) die("engine_create");
    {   /* the checkpoints first: plans name their weights by layer, the bytes come from here */
        int32_t n_i = 0, n_p = 0;
        lssvc_tensor *sd_i = read_checkpoint_blob(ckpt_i, &n_i);
        if (lssvc_engine_load_checkpoint(eng, 0, sd_i, n_i)) die("load_checkpoint(IntraSS)");
        free_checkpoint_blob(sd_i, n_i);                         /* the engine keeps its own copy */
        if (n_frames > 1) {
            lssvc_tensor *sd_p = read_checkpoint_blob(ckpt_p, &n_p);
            if (lssvc_engine_load_checkpoint(eng, 1, sd_p, n_p)) die("load_checkpoint(LSSVC)");
            free_checkpoint_blob(sd_p, n_p);
        }
        printf("checkpoints: %d + %d tensors\n", n_i, n_p);
    }
    if (lssvc_engine_load_intra(eng, argv[1])) die("load_intra");
    if (n_frames > 1 && lssvc_engine_load_inter(eng, argv[2], argv[3])) die("load_inter");
    if (n_frames > 1 && lookahead && lssvc_engine_load_inter_layers(eng, argv[6], argv[7], argv[8], argv[9])) die("load_inter_layers");
    if (lssvc_engine_set_scale(eng, scale, H, W)) die("set_scale");
    for (int which = 0; which < (n_frames > 1 ? 3 : 1); ++which) {
        int64_t info[6];
        if (lssvc_engine_plan_info(eng, which, info)) die("plan_info");
        printf("plan %d: %lld launches on %lld streams, arena %.1f MB, weights %.1f MB, %lldx%lld\n", which, (long long)info[0],
               (long long)info[1], info[2] / 1e6, info[3] / 1e6, (long long)info[4], (long long)info[5]);
    }

    float *all_bl = alloc_f((size_t)n_frames * 3 * bl), *all_el = alloc_f((size_t)n_frames * 3 * el);      /* the loop has every frame at hand, as test.py's */
    for (int t = 0; t < n_frames; ++t)
        if (fread(all_bl + (size_t)t * 3 * bl, 4, 3 * bl, in) != 3 * bl || fread(all_el + (size_t)t * 3 * el, 4, 3 * el, in) != 3 * el) return 2;
    float *ref_bl = alloc_f(3 * bl), *ref_el = alloc_f(3 * el), *feat_bl = alloc_f(64 * bl), *feat_el = alloc_f(64 * el);
    float *mv = alloc_f(2 * el), *warp = alloc_f(3 * el);
    int have_feat_bl = 0;
    for (int t = 0; t < n_frames; ++t) {
        const float *x_bl = all_bl + (size_t)t * 3 * bl, *x_el = all_el + (size_t)t * 3 * el;
        const float *next_bl = t + 1 < n_frames ? all_bl + (size_t)(t + 1) * 3 * bl : NULL;
        double bits[2];
        size_t cf;
        if (t == 0) {                                            /* test.py:219-227 */
            if (lssvc_engine_iframe(eng, x_bl, x_el, bits, ref_bl, ref_el, feat_el, NULL)) die("iframe");
            cf = 64;
            have_feat_bl = 0;
        } else {                                                 /* test.py:229-247: the DPB of frame t-1 goes in, the new one comes out */
            float *n_ref_bl = alloc_f(3 * bl), *n_ref_el = alloc_f(3 * el), *n_feat_bl = alloc_f(64 * bl), *n_feat_el = alloc_f(48 * el);
            if (lookahead) {
                if (lssvc_engine_pframe_lookahead(eng, x_bl, x_el, next_bl, ref_bl, ref_el, have_feat_bl ? feat_bl : NULL, feat_el, bits, n_ref_bl,
                                                  n_feat_bl, n_ref_el, n_feat_el, mv, warp, NULL))
                    die("pframe_lookahead");
            } else if (lssvc_engine_pframe(eng, x_bl, x_el, ref_bl, ref_el, have_feat_bl ? feat_bl : NULL, feat_el, bits, n_ref_bl, n_feat_bl,
                                           n_ref_el, n_feat_el, mv, warp, NULL))
                die("pframe");
            memcpy(ref_bl, n_ref_bl, 3 * bl * 4);
            memcpy(ref_el, n_ref_el, 3 * el * 4);
            memcpy(feat_bl, n_feat_bl, 64 * bl * 4);
            memcpy(feat_el, n_feat_el, 48 * el * 4);
            free(n_ref_bl), free(n_ref_el), free(n_feat_bl), free(n_feat_el);
            cf = 48;
            have_feat_bl = 1;
        }
        printf("frame %d: bit_bl %.6f bit_el %.6f\n", t, bits[0], bits[1]);
        fwrite(bits, 8, 2, out);
        fwrite(ref_bl, 4, 3 * bl, out);
        fwrite(ref_el, 4, 3 * el, out);
        fwrite(feat_el, 4, cf * el, out);
        if (t > 0) {
            fwrite(feat_bl, 4, 64 * bl, out);
            fwrite(mv, 4, 2 * el, out);
            fwrite(warp, 4, 3 * el, out);
        }
        clamp01(ref_bl, 3 * bl);                                 /* test.py:249-250 */
        clamp01(ref_el, 3 * el);
    }
    lssvc_engine_destroy(eng);
    fclose(in);
    fclose(out);
    return 0;
}
