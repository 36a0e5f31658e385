/* engine_stream_demo.c -- write_stream = 1 without Python: plain C against include/lssvc_hip.h, linked with liblssvc_hip.so.
 * Two roles, run as two separate processes by tests/test_gpu_engine.py, so that the decoder sees nothing but the files:
 *
 *   engine_stream_demo enc <dir> <case.bin>   ENCODER: loads <dir>/{i,p1,p}_enc.plan, codes the clip frame by frame as the
 *        reference's loop does (test.py:212-250 with write_stream = 1): lssvc_engine_encode_iframe, then
 *        lssvc_engine_encode_pframe with the DPB it got back (EL reconstruction clamped to [0, 1] by the caller, the BL one
 *        comes back clamped). Writes <dir>/<t>_BL.bin and <dir>/<t>_EL.bin -- the reference's layer files -- and <dir>/enc.out:
 *        per frame recon_bl (3hw), recon_el (3HW, un-clamped), feature_el (Cf HW; Cf = 64 after the I-frame, 48 after a
 *        P-frame) and for P-frames feature_bl (64hw).
 *   engine_stream_demo dec <dir> <case.bin>   DECODER: loads <dir>/{i,p1,p}_dec.plan, reads only the header of case.bin (frame
 *        count and sizes) and the .bin files, reconstructs every frame, writes <dir>/dec.out in the same layout.
 *
 * Both roles first read <dir>/intra.ckpt and <dir>/inter.ckpt, the two RAW checkpoints (tests/ckpt_blob.h), and hand them to
 * lssvc_engine_load_checkpoint: the plan files hold launches, host steps and CDF tables; the network weights come from the caller.
 *
 * case.bin: int32 n_frames, H, W, h, w; float scale; then per frame x_bl (3*h*w floats) and x_el (3*H*W floats), NCHW. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lssvc_hip.h"
#include "ckpt_blob.h"

static void die(const char *what) {
    fprintf(stderr, "engine_stream_demo: %s: %s\n", what, lssvc_last_error());
    exit(1);
}

static void *xmalloc(size_t n) {
    void *p = malloc(n ? n : 1);
    if (!p) {
        fprintf(stderr, "engine_stream_demo: out of memory\n");
        exit(1);
    }
    return p;
}

static void clamp01(float *x, size_t n) {
    for (size_t i = 0; i < n; ++i) x[i] = x[i] < 0.f ? 0.f : (x[i] > 1.f ? 1.f : x[i]);
}

static void path_of(char *dst, size_t cap, const char *dir, const char *name) { snprintf(dst, cap, "%s/%s", dir, name); }

static void write_file(const char *path, const uint8_t *data, int64_t n) {
    FILE *f = fopen(path, "wb");
    if (!f || fwrite(data, 1, (size_t)n, f) != (size_t)n) {
        fprintf(stderr, "engine_stream_demo: cannot write %s\n", path);
        exit(1);
    }
    fclose(f);
}

static int64_t read_file(const char *path, uint8_t *data, int64_t cap) {
    FILE *f = fopen(path, "rb");
    if (!f) {
        fprintf(stderr, "engine_stream_demo: cannot read %s\n", path);
        exit(1);
    }
    const int64_t n = (int64_t)fread(data, 1, (size_t)cap, f);
    fclose(f);
    return n;
}

int main(int argc, char **argv) {
    if (argc != 4 || (strcmp(argv[1], "enc") && strcmp(argv[1], "dec"))) {
        fprintf(stderr, "usage: engine_stream_demo enc|dec <dir> <case.bin>\n");
        return 2;
    }
    const int encoder = strcmp(argv[1], "enc") == 0;
    const char *dir = argv[2];
    FILE *in = fopen(argv[3], "rb");
    if (!in) return 2;
    int32_t hdr[5];
    float scale;
    if (fread(hdr, 4, 5, in) != 5 || fread(&scale, 4, 1, in) != 1) return 2;
    const int n_frames = hdr[0], H = hdr[1], W = hdr[2], h = hdr[3], w = hdr[4];
    const size_t el = (size_t)H * W, bl = (size_t)h * w;
    char p0[1024], p1[1024], p2[1024], name[64];

    void *eng = lssvc_engine_create(0);
    if (!eng) die("engine_create");
    {
        static const char *const ck[2] = {"intra.ckpt", "inter.ckpt"};
        for (int m = 0; m < (n_frames > 1 ? 2 : 1); ++m) {
            int32_t n = 0;
            path_of(p0, sizeof p0, dir, ck[m]);
            lssvc_tensor *sd = read_checkpoint_blob(p0, &n);
            if (lssvc_engine_load_checkpoint(eng, m, sd, n)) die("load_checkpoint");
            free_checkpoint_blob(sd, n);
        }
    }
    path_of(p0, sizeof p0, dir, encoder ? "i_enc.plan" : "i_dec.plan");
    path_of(p1, sizeof p1, dir, encoder ? "p1_enc.plan" : "p1_dec.plan");
    path_of(p2, sizeof p2, dir, encoder ? "p_enc.plan" : "p_dec.plan");
    if (encoder ? lssvc_engine_load_stream(eng, p0, NULL, p1, NULL, p2, NULL) : lssvc_engine_load_stream(eng, NULL, p0, NULL, p1, NULL, p2))
        die("load_stream");
    if (lssvc_engine_set_scale(eng, scale, H, W)) die("set_scale");

    path_of(p0, sizeof p0, dir, encoder ? "enc.out" : "dec.out");
    FILE *out = fopen(p0, "wb");
    if (!out) return 2;
    float *x_bl = xmalloc(3 * bl * 4), *x_el = xmalloc(3 * el * 4);
    float *recon_bl = xmalloc(3 * bl * 4), *recon_el = xmalloc(3 * el * 4), *feature_el = xmalloc(64 * el * 4), *feature_bl = xmalloc(64 * bl * 4);
    float *ref_bl = xmalloc(3 * bl * 4), *ref_el = xmalloc(3 * el * 4), *ref_feature_el = xmalloc(64 * el * 4), *ref_feature_bl = xmalloc(64 * bl * 4);
    const int64_t cap = 64 + (int64_t)(16 * el);           /* far more than a layer of these test clips needs */
    uint8_t *f_bl = xmalloc((size_t)cap), *f_el = xmalloc((size_t)cap);
    int have_feature_bl = 0;
    int64_t total = 0;
    for (int t = 0; t < n_frames; ++t) {
        int64_t n_bl = 0, n_el = 0;
        const size_t cf = t == 0 ? 64 : 48;
        if (encoder) {
            if (fread(x_bl, 4, 3 * bl, in) != 3 * bl || fread(x_el, 4, 3 * el, in) != 3 * el) return 2;
            if (t == 0) {
                if (lssvc_engine_encode_iframe(eng, x_bl, x_el, f_bl, cap, &n_bl, f_el, cap, &n_el, recon_bl, recon_el, feature_el, NULL)) die("encode_iframe");
            } else if (lssvc_engine_encode_pframe(eng, x_bl, x_el, ref_bl, ref_el, have_feature_bl ? ref_feature_bl : NULL, ref_feature_el, f_bl, cap,
                                                  &n_bl, f_el, cap, &n_el, recon_bl, feature_bl, recon_el, feature_el, NULL)) {
                die("encode_pframe");
            }
            snprintf(name, sizeof name, "%d_BL.bin", t);
            path_of(p1, sizeof p1, dir, name);
            write_file(p1, f_bl, n_bl);
            snprintf(name, sizeof name, "%d_EL.bin", t);
            path_of(p1, sizeof p1, dir, name);
            write_file(p1, f_el, n_el);
        } else {
            snprintf(name, sizeof name, "%d_BL.bin", t);
            path_of(p1, sizeof p1, dir, name);
            n_bl = read_file(p1, f_bl, cap);
            snprintf(name, sizeof name, "%d_EL.bin", t);
            path_of(p1, sizeof p1, dir, name);
            n_el = read_file(p1, f_el, cap);
            if (t == 0) {
                if (lssvc_engine_decode_iframe(eng, f_bl, n_bl, f_el, n_el, recon_bl, recon_el, feature_el, NULL)) die("decode_iframe");
            } else if (lssvc_engine_decode_pframe(eng, f_bl, n_bl, f_el, n_el, ref_bl, ref_el, have_feature_bl ? ref_feature_bl : NULL, ref_feature_el,
                                                  recon_bl, feature_bl, recon_el, feature_el, NULL)) {
                die("decode_pframe");
            }
        }
        total += n_bl + n_el;
        fwrite(recon_bl, 4, 3 * bl, out);
        fwrite(recon_el, 4, 3 * el, out);
        fwrite(feature_el, 4, cf * el, out);
        if (t > 0) fwrite(feature_bl, 4, 64 * bl, out);
        /* the caller owns the DPB: clamp the reconstructions (test.py:249-250) and hand everything back */
        memcpy(ref_bl, recon_bl, 3 * bl * 4);
        memcpy(ref_el, recon_el, 3 * el * 4);
        clamp01(ref_bl, 3 * bl);
        clamp01(ref_el, 3 * el);
        memcpy(ref_feature_el, feature_el, cf * el * 4);
        if (t > 0) {
            memcpy(ref_feature_bl, feature_bl, 64 * bl * 4);
            have_feature_bl = 1;
        }
        printf("%s frame %d: BL %lld + EL %lld bytes\n", encoder ? "encoded" : "decoded", t, (long long)n_bl, (long long)n_el);
    }
    printf("%s %d frames, %lld bytes of layer files\n", encoder ? "encoded" : "decoded", n_frames, (long long)total);
    fclose(out);
    fclose(in);
    lssvc_engine_destroy(eng);
    return 0;
}
