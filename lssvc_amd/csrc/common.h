// Shared host/device helpers for liblssvc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/lssvc_hip.h"

namespace lssvc {

// thread-local error text behind lssvc_last_error()
char *err_buf();
int fail(const char *fmt, ...);

#define LSSVC_CHECK(cond, ...)                 \
    do {                                       \
        if (!(cond)) return lssvc::fail(__VA_ARGS__); \
    } while (0)

#define LSSVC_HIP(call)                                                             \
    do {                                                                            \
        hipError_t e_ = (call);                                                     \
        if (e_ != hipSuccess) return lssvc::fail("%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)

inline int launch_status(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("%s launch: %s", what, hipGetErrorString(e));
    return 0;
}

inline bool view_ok(const lssvc_view *v) {
    return v && v->ptr && v->H > 0 && v->W > 0 && v->C > 0 && v->ld >= v->C;
}
inline bool same_hw(const lssvc_view *a, const lssvc_view *b) { return a->H == b->H && a->W == b->W; }
inline bool same_shape(const lssvc_view *a, const lssvc_view *b) { return same_hw(a, b) && a->C == b->C; }
inline bool vec4_ok(const lssvc_view *v) {
    return (v->C % 4 == 0) && (v->ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(v->ptr) & 15) == 0);
}

// Device-side copy of a view (plain struct, passed by value in kernel args).
struct V {
    float *p;
    int H, W, C, ld;
};
inline V mk(const lssvc_view *v) { return V{v->ptr, v->H, v->W, v->C, v->ld}; }
inline V mk_null() { return V{nullptr, 0, 0, 0, 0}; }

constexpr int kReduceMaxBlocks = 1024;  // partial-sum slots in the reduction workspace

}  // namespace lssvc
