// Shared host/device helpers for liblssvc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
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

// ---- per-device launch state. A process may drive several GPUs (one stream each); the CU count and the
// dynamic-LDS grant of a kernel (hipFuncSetAttribute acts on the CURRENT device) are therefore cached per
// device ordinal, not per process.
constexpr int kMaxDevices = 64;          // far above any node (8 GPUs); an ordinal beyond it shares the last slot's cache,
inline int current_device() {            // which only costs a repeated hipFuncSetAttribute, never a wrong grant: see ensure()
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0) d = 0;
    return d < kMaxDevices ? d : kMaxDevices - 1;
}
inline bool device_slot_is_shared() {
    int d = 0;
    return hipGetDevice(&d) == hipSuccess && d >= kMaxDevices - 1;
}
inline int device_cus() {
    static std::atomic<int> cus[kMaxDevices];
    const int d = current_device();
    int v = device_slot_is_shared() ? 0 : cus[d].load(std::memory_order_relaxed);
    if (v > 0) return v;
    int real = 0;
    if (hipGetDevice(&real) != hipSuccess) real = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, real) != hipSuccess || v <= 0) v = 256;
    cus[d].store(v, std::memory_order_relaxed);
    return v;
}
// One instance per kernel instantiation (a function-local static): raises the kernel's dynamic-LDS limit on the
// current device when this launch needs more than was granted there so far (`base` = what needs no grant).
struct LdsGrant {
    std::atomic<size_t> granted[kMaxDevices];
    int ensure(const void *fn, size_t bytes, size_t base = 0) {
        const int d = current_device();
        size_t g = device_slot_is_shared() ? 0 : granted[d].load(std::memory_order_relaxed);   // shared slot: always (re)grant
        if (g < base) g = base;
        if (bytes <= g) return 0;
        LSSVC_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        granted[d].store(bytes, std::memory_order_relaxed);
        return 0;
    }
};

// runtime tuning switches (lssvc_set_option / environment), see conv_mfma.hip
int option_get(int which);
enum { OPT_P3_ON = 0, OPT_P3_MIN_TILES = 1, OPT_P7_ON = 2, OPT_POINTWISE_BLOCKS = 3, OPT_DWPRE_DEEP = 4, OPT_P3_BLOCKS = 5, OPT_P3_STAGE = 6, OPT_P3_S2 = 7,
       OPT_P3_SMALL = 8, OPT_P3_NARROW = 9, OPT_P3_PF2 = 10, OPT_P3_FORCE = 11, OPT_GDN_FAST = 12, OPT_P3_BIG_PAIR = 13, OPT_P7_NARROW = 14, OPT_RESAMPLE_ROWS = 15, OPT_COUNT = 16 };

}  // namespace lssvc
