/* ckpt_blob.h -- the raw checkpoint file the C demo programs read: a state dict as the reference's torch.load returns it
 * (IntraSS.py:190-214, LSSVC_net.py:141-149), dumped tensor by tensor with no re-layout (tests/helpers.py:
 * write_checkpoint_blob; any ten-line exporter of a .pth file does):
 *     "LSSVCCK1", int32 n, then n x { int32 name_len, name bytes, int32 ndim, int64 shape[4], float data[prod(shape)] }
 * -> an array of lssvc_tensor for lssvc_engine_load_checkpoint. */
#ifndef LSSVC_TESTS_CKPT_BLOB_H
#define LSSVC_TESTS_CKPT_BLOB_H
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lssvc_hip.h"

static lssvc_tensor *read_checkpoint_blob(const char *path, int32_t *n_out) {
    FILE *f = fopen(path, "rb");
    char magic[8];
    int32_t n = 0;
    if (!f || fread(magic, 1, 8, f) != 8 || memcmp(magic, "LSSVCCK1", 8) != 0 || fread(&n, 4, 1, f) != 1 || n <= 0 || n > 100000) {
        fprintf(stderr, "cannot read checkpoint blob %s\n", path);
        exit(2);
    }
    lssvc_tensor *t = (lssvc_tensor *)calloc((size_t)n, sizeof(lssvc_tensor));
    for (int32_t i = 0; i < n; ++i) {
        int32_t len = 0, ndim = 0;
        if (fread(&len, 4, 1, f) != 1 || len <= 0 || len > 1000) exit(2);
        char *name = (char *)calloc((size_t)len + 1, 1);
        if (fread(name, 1, (size_t)len, f) != (size_t)len || fread(&ndim, 4, 1, f) != 1 || ndim < 0 || ndim > 4) exit(2);
        int64_t shape[4], numel = 1;
        if (fread(shape, 8, 4, f) != 4) exit(2);
        for (int d = 0; d < ndim; ++d) numel *= shape[d];
        float *data = (float *)malloc((size_t)(numel > 0 ? numel : 1) * sizeof(float));
        if (!data || fread(data, 4, (size_t)numel, f) != (size_t)numel) exit(2);
        t[i].name = name;
        t[i].data = data;
        t[i].ndim = ndim;
        memcpy(t[i].shape, shape, sizeof(shape));
    }
    fclose(f);
    *n_out = n;
    return t;
}

static void free_checkpoint_blob(lssvc_tensor *t, int32_t n) {
    for (int32_t i = 0; i < n; ++i) {
        free((void *)t[i].name);
        free((void *)t[i].data);
    }
    free(t);
}
#endif
