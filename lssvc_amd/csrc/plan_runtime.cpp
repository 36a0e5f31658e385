// plan_runtime.cpp -- executes compiled frame plans: the engine-level entry points of the C ABI (include/lssvc_hip.h,
// "engine"), for callers without Python or PyTorch.
//
// A plan file (written by lssvc_amd/plan_compiler.py) is the estimate-mode forward of ONE frame type at ONE size -- what the
// reference runs per frame in IntraSS.forward (src/models/IntraSS.py:137-172) or LSSVC.forward_one_frame
// (src/models/LSSVC_net.py:445-528) -- as the fixed sequence of library launches the Python front end issued for it: memory
// regions (one activation arena, the prepared weight tensors with their data, zeroed scratch, the caller's input / output
// buffers), and launches whose device pointers are (region, offset) pairs. The engine allocates the regions, rebases the
// pointers, replays the launches once eagerly (kernels raise their LDS limits on first use, which a stream capture forbids)
// and then captures them -- side streams and their fork / join waits included -- into a hipGraph with
// hipStreamBeginCapture; a frame after that is: copy the caller's inputs into the plan's input buffers, hipGraphLaunch, copy
// the outputs and the bit counters out (the graph itself only ever sees the engine's own memory, so it is captured once).
// Results are bit-identical to the Python path: same kernels, same arguments, same order.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "common.h"

using namespace lssvc;

namespace {

enum { REGION_ARENA = 0, REGION_WEIGHTS = 1, REGION_SCRATCH = 2, REGION_INPUT = 3, REGION_OUTPUT = 4 };
enum { TAG_NULL = 0, TAG_PTR = 1, TAG_STRUCT = 2, TAG_F32 = 3, TAG_I32 = 4, TAG_I64 = 5, TAG_I32ARRAY = 6, TAG_STREAM = 7 };

struct Region {
    uint32_t kind = 0;
    uint64_t nbytes = 0;
    int64_t shape[4] = {0, 0, 0, 0};
    std::string name;
    void *ptr = nullptr;
};

struct Fix {
    uint32_t field, region;
    uint64_t offset;
};

struct Arg {
    uint32_t tag = TAG_NULL;
    uint32_t region = 0;
    uint64_t offset = 0;
    float f = 0.f;
    int64_t i = 0;
    std::vector<unsigned char> blob;     // struct image (pointer fields rebased at load) or an int32 array
    std::vector<Fix> fixes;
    void *ptr = nullptr;                 // resolved TAG_PTR
};

struct Launch {
    std::string fn;
    uint32_t stream = 0;
    std::vector<Arg> args;
    int id = -1;
};

enum Fn {
    FN_WAIT, FN_CONV2D, FN_CONV1X1_DW, FN_FFN, FN_DWCONV, FN_RESIZE, FN_WARP, FN_POOL, FN_SOFTMAX2, FN_ADD, FN_COPY, FN_LRELU,
    FN_OFFSET_DIVERSITY, FN_NCHW_TO_NHWC, FN_NHWC_TO_NCHW, FN_LAPLACE_QUANT_BITS, FN_FOUR_PART_STEP, FN_LAPLACE_BITS,
    FN_FACTORIZED, FN_GAUSSIAN, FN_BOTTLENECK, FN_FILL_ZERO, FN_CLAMP, FN_COUNT
};
const char *const kFnNames[FN_COUNT] = {
    "__wait__", "lssvc_conv2d", "lssvc_conv1x1_dw3x3_f16x3", "lssvc_ffn_f16x3", "lssvc_dwconv3x3", "lssvc_resize_bilinear",
    "lssvc_flow_warp", "lssvc_pool2x2", "lssvc_softmax2_blend", "lssvc_add", "lssvc_copy", "lssvc_lrelu", "lssvc_offset_diversity",
    "lssvc_nchw_to_nhwc", "lssvc_nhwc_to_nchw", "lssvc_laplace_quant_bits", "lssvc_four_part_step", "lssvc_laplace_bits",
    "lssvc_factorized_quant_bits", "lssvc_gaussian_conditional", "lssvc_entropy_bottleneck", "lssvc_fill_zero", "lssvc_clamp_inplace"};

struct Plan {
    std::string kind;
    double scale = 0;
    int H = 0, W = 0;
    std::vector<std::pair<std::string, int64_t>> meta;
    std::vector<Region> regions;
    std::vector<Launch> launches;
    uint32_t n_streams = 1;
    std::vector<hipStream_t> side;       // streams 1 .. n-1 (stream 0 is the caller's)
    std::vector<hipEvent_t> events;
    hipGraphExec_t exec = nullptr;
    int runs = 0;
    int bits_region = -1;

    ~Plan() {
        if (exec) (void)hipGraphExecDestroy(exec);
        for (auto e : events) (void)hipEventDestroy(e);
        for (auto s : side) (void)hipStreamDestroy(s);
        for (auto &r : regions)
            if (r.ptr) (void)hipFree(r.ptr);
    }
    int region_index(const char *name, uint32_t kind) const {
        for (size_t i = 0; i < regions.size(); ++i)
            if (regions[i].kind == kind && regions[i].name == name) return (int)i;
        return -1;
    }
    int64_t meta_value(const char *name, int64_t dflt) const {
        for (auto &m : meta)
            if (m.first == name) return m.second;
        return dflt;
    }
};

struct Reader {
    FILE *f;
    bool ok = true;
    template <typename T>
    T get() {
        T v{};
        if (fread(&v, sizeof(T), 1, f) != 1) ok = false;
        return v;
    }
    std::string str48() {
        char b[49] = {0};
        if (fread(b, 1, 48, f) != 48) ok = false;
        return std::string(b);
    }
    void bytes(void *dst, size_t n) {
        if (n && fread(dst, 1, n, f) != n) ok = false;
    }
};

int bind(Plan &p);

int load_plan(const char *path, Plan &p) {
    FILE *f = fopen(path, "rb");
    LSSVC_CHECK(f != nullptr, "engine: cannot open plan file %s", path);
    std::unique_ptr<FILE, int (*)(FILE *)> closer(f, fclose);
    Reader r{f};
    char magic[8];
    r.bytes(magic, 8);
    LSSVC_CHECK(r.ok && memcmp(magic, "LSSVCPL1", 8) == 0, "engine: %s is not a frame plan", path);
    const uint32_t n_regions = r.get<uint32_t>(), n_launches = r.get<uint32_t>();
    p.n_streams = r.get<uint32_t>();
    const uint32_t n_meta = r.get<uint32_t>();
    r.get<uint32_t>();
    p.scale = r.get<double>();
    p.H = r.get<int32_t>();
    p.W = r.get<int32_t>();
    p.kind = r.str48();
    for (uint32_t i = 0; i < n_meta; ++i) {
        std::string name = r.str48();
        p.meta.emplace_back(name, r.get<int64_t>());
    }
    p.regions.resize(n_regions);
    for (auto &g : p.regions) {
        g.kind = r.get<uint32_t>();
        g.nbytes = r.get<uint64_t>();
        for (int k = 0; k < 4; ++k) g.shape[k] = r.get<int64_t>();
        g.name = r.str48();
    }
    p.launches.resize(n_launches);
    for (auto &l : p.launches) {
        l.fn = r.str48();
        l.stream = r.get<uint32_t>();
        const uint32_t n_args = r.get<uint32_t>();
        LSSVC_CHECK(r.ok && n_args <= 16 && l.stream < p.n_streams, "engine: corrupt plan (launch header)");
        l.args.resize(n_args);
        for (auto &a : l.args) {
            a.tag = r.get<uint32_t>();
            switch (a.tag) {
            case TAG_PTR:
                a.region = r.get<uint32_t>();
                a.offset = r.get<uint64_t>();
                break;
            case TAG_STRUCT: {
                const uint32_t len = r.get<uint32_t>(), nfix = r.get<uint32_t>();
                LSSVC_CHECK(r.ok && len <= 4096 && nfix <= 64, "engine: corrupt plan (struct argument)");
                a.blob.resize((len + 7) / 8 * 8);
                r.bytes(a.blob.data(), a.blob.size());
                a.fixes.resize(nfix);
                for (auto &x : a.fixes) {
                    x.field = r.get<uint32_t>();
                    x.region = r.get<uint32_t>();
                    x.offset = r.get<uint64_t>();
                }
                break;
            }
            case TAG_F32: a.f = r.get<float>(); break;
            case TAG_I32: a.i = r.get<int32_t>(); break;
            case TAG_I64: a.i = r.get<int64_t>(); break;
            case TAG_I32ARRAY: {
                const uint32_t n = r.get<uint32_t>();
                LSSVC_CHECK(r.ok && n <= 64, "engine: corrupt plan (array argument)");
                a.blob.resize(4 * n);
                r.bytes(a.blob.data(), a.blob.size());
                break;
            }
            case TAG_NULL:
            case TAG_STREAM: break;
            default: return fail("engine: corrupt plan (argument tag %u)", a.tag);
            }
        }
        for (int k = 0; k < FN_COUNT; ++k)
            if (l.fn == kFnNames[k]) l.id = k;
        LSSVC_CHECK(l.id >= 0, "engine: plan uses %s, which this runtime does not replay", l.fn.c_str());
    }
    LSSVC_CHECK(r.ok, "engine: truncated plan file %s", path);
    // ---- regions: allocate, upload the weights (payloads follow the launch list, 256-byte aligned, in region order)
    std::vector<unsigned char> host;
    for (auto &g : p.regions) {
        LSSVC_HIP(hipMalloc(&g.ptr, g.nbytes ? g.nbytes : 16));
        if (g.kind == REGION_SCRATCH) LSSVC_HIP(hipMemset(g.ptr, 0, g.nbytes));
        if (g.kind == REGION_WEIGHTS) {
            const long pos = ftell(f);
            fseek(f, (256 - pos % 256) % 256, SEEK_CUR);
            host.resize(g.nbytes);
            r.bytes(host.data(), g.nbytes);
            LSSVC_CHECK(r.ok, "engine: truncated plan file %s (weights)", path);
            LSSVC_HIP(hipMemcpy(g.ptr, host.data(), g.nbytes, hipMemcpyHostToDevice));
        }
    }
    p.bits_region = p.region_index("bits", REGION_SCRATCH);
    LSSVC_CHECK(p.bits_region >= 0, "engine: plan has no bit-counter region");
    for (uint32_t s = 1; s < p.n_streams; ++s) {
        hipStream_t st;
        LSSVC_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        p.side.push_back(st);
    }
    return bind(p);
}
}  // namespace

namespace {
// resolve every (region, offset) against the regions' base addresses
int bind(Plan &p) {
    for (auto &l : p.launches)
        for (auto &a : l.args) {
            if (a.tag == TAG_PTR) {
                LSSVC_CHECK(a.region < p.regions.size() && p.regions[a.region].ptr, "engine: unbound region %u", a.region);
                a.ptr = static_cast<char *>(p.regions[a.region].ptr) + a.offset;
            } else if (a.tag == TAG_STRUCT) {
                for (auto &x : a.fixes) {
                    LSSVC_CHECK(x.region < p.regions.size() && p.regions[x.region].ptr && x.field + 8 <= a.blob.size(),
                                "engine: unbound region %u", x.region);
                    void *q = static_cast<char *>(p.regions[x.region].ptr) + x.offset;
                    memcpy(a.blob.data() + x.field, &q, 8);
                }
            }
        }
    return 0;
}

int replay(Plan &p, hipStream_t main) {
    auto stream_of = [&](uint32_t s) { return s == 0 ? main : p.side[s - 1]; };
    size_t ev = 0;
    for (auto &l : p.launches) {
        hipStream_t st = stream_of(l.stream);
        auto P = [&](int i) -> void * { return l.args[i].tag == TAG_PTR ? l.args[i].ptr : nullptr; };
        auto V = [&](int i) { return reinterpret_cast<const lssvc_view *>(l.args[i].blob.data()); };
        auto F = [&](int i) { return l.args[i].f; };
        auto I = [&](int i) { return (int32_t)l.args[i].i; };
        int rc = 0;
        switch (l.id) {
        case FN_WAIT: {                                  // stream l.stream waits for what stream args[0] has been given so far
            if (ev == p.events.size()) {
                hipEvent_t e;
                LSSVC_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                p.events.push_back(e);
            }
            LSSVC_HIP(hipEventRecord(p.events[ev], stream_of((uint32_t)l.args[0].i)));
            LSSVC_HIP(hipStreamWaitEvent(st, p.events[ev], 0));
            ++ev;
            break;
        }
        case FN_CONV2D: rc = lssvc_conv2d(reinterpret_cast<const lssvc_conv_desc *>(l.args[0].blob.data()), st); break;
        case FN_CONV1X1_DW:
            rc = lssvc_conv1x1_dw3x3_f16x3(reinterpret_cast<const lssvc_conv_desc *>(l.args[0].blob.data()), (const float *)P(1),
                                           (const float *)P(2), st);
            break;
        case FN_FFN: rc = lssvc_ffn_f16x3(reinterpret_cast<const lssvc_ffn_desc *>(l.args[0].blob.data()), st); break;
        case FN_DWCONV: rc = lssvc_dwconv3x3(V(0), (const float *)P(1), (const float *)P(2), V(3), st); break;
        case FN_RESIZE: rc = lssvc_resize_bilinear(V(0), V(1), F(2), st); break;
        case FN_WARP: rc = lssvc_flow_warp(V(0), V(1), V(2), st); break;
        case FN_POOL: rc = lssvc_pool2x2(V(0), V(1), I(2), st); break;
        case FN_SOFTMAX2: rc = lssvc_softmax2_blend(V(0), V(1), V(2), V(3), st); break;
        case FN_ADD: rc = lssvc_add(V(0), V(1), V(2), st); break;
        case FN_COPY: rc = lssvc_copy(V(0), V(1), st); break;
        case FN_LRELU: rc = lssvc_lrelu(V(0), V(1), F(2), st); break;
        case FN_OFFSET_DIVERSITY: rc = lssvc_offset_diversity(V(0), V(1), V(2), (const float *)P(3), (const float *)P(4), V(5), st); break;
        case FN_NCHW_TO_NHWC: rc = lssvc_nchw_to_nhwc((const float *)P(0), V(1), st); break;
        case FN_NHWC_TO_NCHW: rc = lssvc_nhwc_to_nchw(V(0), (float *)P(1), st); break;
        case FN_LAPLACE_QUANT_BITS: rc = lssvc_laplace_quant_bits(V(0), V(1), V(2), V(3), V(4), (double *)P(5), P(6), st); break;
        case FN_FOUR_PART_STEP:
            rc = lssvc_four_part_step(V(0), V(1), V(2), reinterpret_cast<const int32_t *>(l.args[3].blob.data()), V(4), V(5), V(6), st);
            break;
        case FN_LAPLACE_BITS: rc = lssvc_laplace_bits(V(0), V(1), (double *)P(2), P(3), st); break;
        case FN_FACTORIZED: rc = lssvc_factorized_quant_bits(V(0), (const float *)P(1), V(2), (double *)P(3), P(4), st); break;
        case FN_GAUSSIAN: rc = lssvc_gaussian_conditional(V(0), V(1), V(2), V(3), V(4), (double *)P(5), P(6), st); break;
        case FN_BOTTLENECK: rc = lssvc_entropy_bottleneck(V(0), (const float *)P(1), V(2), V(3), (double *)P(4), P(5), st); break;
        case FN_FILL_ZERO: rc = lssvc_fill_zero(P(0), l.args[1].i, st); break;
        case FN_CLAMP: rc = lssvc_clamp_inplace((float *)P(0), l.args[1].i, F(2), F(3), st); break;
        default: return fail("engine: no replay for %s", l.fn.c_str());
        }
        if (rc) return rc;
    }
    return 0;
}

struct Engine {
    int device = 0;
    hipStream_t own = nullptr;           // frames run here when the caller passes stream NULL (the null stream cannot be captured)
    std::unique_ptr<Plan> intra, first_p, steady_p;
    ~Engine() {
        intra.reset(), first_p.reset(), steady_p.reset();
        if (own) (void)hipStreamDestroy(own);
    }
    hipStream_t stream(void *s) const { return s ? (hipStream_t)s : own; }
};

// Run one plan: caller's inputs in, first call eager, second call capture, later calls hipGraphLaunch, outputs out.
int run_plan(Plan &p, const std::vector<std::pair<const char *, const void *>> &ins,
             const std::vector<std::pair<const char *, void *>> &outs, double *slots16, hipStream_t st) {
    for (auto &kv : ins) {
        const int i = p.region_index(kv.first, REGION_INPUT);
        if (i < 0) {
            LSSVC_CHECK(kv.second == nullptr, "engine: this plan takes no input named %s", kv.first);
            continue;
        }
        LSSVC_CHECK(kv.second != nullptr, "engine: input %s is required by this plan", kv.first);
        LSSVC_HIP(hipMemcpyAsync(p.regions[i].ptr, kv.second, p.regions[i].nbytes, hipMemcpyDefault, st));
    }
    if (p.runs == 0 || std::getenv("LSSVC_ENGINE_EAGER")) {
        if (int e = replay(p, st)) return e;
    } else {
        if (!p.exec) {
            hipGraph_t graph = nullptr;
            LSSVC_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            const int rc = replay(p, st);
            const hipError_t ce = hipStreamEndCapture(st, &graph);
            if (rc) return rc;
            LSSVC_CHECK(ce == hipSuccess && graph, "engine: hipStreamEndCapture: %s", hipGetErrorString(ce));
            const hipError_t ie = hipGraphInstantiate(&p.exec, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            LSSVC_CHECK(ie == hipSuccess, "engine: hipGraphInstantiate: %s", hipGetErrorString(ie));
        }
        LSSVC_HIP(hipGraphLaunch(p.exec, st));
    }
    ++p.runs;
    for (auto &kv : outs) {
        if (!kv.second) continue;                        // the caller does not want this one
        const int i = p.region_index(kv.first, REGION_OUTPUT);
        LSSVC_CHECK(i >= 0, "engine: this plan has no output named %s", kv.first);
        LSSVC_HIP(hipMemcpyAsync(kv.second, p.regions[i].ptr, p.regions[i].nbytes, hipMemcpyDefault, st));
    }
    LSSVC_HIP(hipMemcpyAsync(slots16, p.regions[p.bits_region].ptr, 16 * sizeof(double), hipMemcpyDeviceToHost, st));
    LSSVC_HIP(hipStreamSynchronize(st));                 // the bit counts are host values, as in the reference (.item())
    return 0;
}



}  // namespace

extern "C" void *lssvc_engine_create(int32_t device) {
    if (hipSetDevice(device) != hipSuccess) {
        fail("engine: hipSetDevice(%d) failed", device);
        return nullptr;
    }
    Engine *e = new Engine();
    e->device = device;
    if (hipStreamCreateWithFlags(&e->own, hipStreamNonBlocking) != hipSuccess) {
        fail("engine: hipStreamCreate failed");
        delete e;
        return nullptr;
    }
    return e;
}

extern "C" void lssvc_engine_destroy(void *h) { delete static_cast<Engine *>(h); }

static int load_into(std::unique_ptr<Plan> &slot, const char *path, const char *want_a, const char *want_b) {
    std::unique_ptr<Plan> p(new Plan());
    if (int e = load_plan(path, *p)) return e;
    LSSVC_CHECK(p->kind == want_a || (want_b && p->kind == want_b), "engine: %s holds a '%s' plan", path, p->kind.c_str());
    slot = std::move(p);
    return 0;
}

extern "C" int lssvc_engine_load_intra(void *h, const char *iframe_plan) {
    LSSVC_CHECK(h && iframe_plan, "engine_load_intra: bad arguments");
    Engine *e = static_cast<Engine *>(h);
    LSSVC_HIP(hipSetDevice(e->device));
    return load_into(e->intra, iframe_plan, "iframe", nullptr);
}

extern "C" int lssvc_engine_load_inter(void *h, const char *first_p_plan, const char *steady_p_plan) {
    LSSVC_CHECK(h && first_p_plan && steady_p_plan, "engine_load_inter: bad arguments");
    Engine *e = static_cast<Engine *>(h);
    LSSVC_HIP(hipSetDevice(e->device));
    if (int rc = load_into(e->first_p, first_p_plan, "pframe_first", nullptr)) return rc;
    return load_into(e->steady_p, steady_p_plan, "pframe", nullptr);
}

extern "C" int lssvc_engine_set_scale(void *h, float scale, int32_t H, int32_t W) {
    LSSVC_CHECK(h, "engine_set_scale: bad arguments");
    Engine *e = static_cast<Engine *>(h);
    for (Plan *p : {e->intra.get(), e->first_p.get(), e->steady_p.get()})
        if (p)
            LSSVC_CHECK(std::fabs(p->scale - (double)scale) < 1e-9 && p->H == H && p->W == W,
                        "engine_set_scale: the loaded '%s' plan was compiled for scale %g, %dx%d (asked for %g, %dx%d)", p->kind.c_str(),
                        p->scale, p->H, p->W, (double)scale, H, W);
    return 0;
}

extern "C" int lssvc_engine_iframe(void *h, const float *x_bl, const float *x_el, double bits[2], float *x_hat_bl, float *x_hat_el,
                                   float *feature_el, void *stream) {
    LSSVC_CHECK(h && bits, "engine_iframe: bad arguments");
    Engine *e = static_cast<Engine *>(h);
    LSSVC_CHECK(e->intra != nullptr, "engine_iframe: no I-frame plan loaded");
    double s[16];
    if (int rc = run_plan(*e->intra, {{"x_bl", x_bl}, {"x_el", x_el}}, {{"x_hat_bl", x_hat_bl}, {"x_hat_el", x_hat_el}, {"feature_el", feature_el}},
                          s, e->stream(stream)))
        return rc;
    bits[0] = (s[0] + s[1]) / -std::log(2.0);            // IntraSS.py:163, priors.py:377: sum of log-likelihoods / -ln 2
    bits[1] = (s[2] + s[3]) / -std::log(2.0);
    return 0;
}

extern "C" int lssvc_engine_pframe(void *h, const float *x_bl, const float *x_el, const float *ref_frame_bl, const float *ref_frame_el,
                                   const float *ref_feature_bl, const float *ref_feature_el, double bits[2], float *recon_bl,
                                   float *feature_bl, float *recon_el, float *feature_el, float *mv_hat, float *warp_frame, void *stream) {
    LSSVC_CHECK(h && bits, "engine_pframe: bad arguments");
    Engine *e = static_cast<Engine *>(h);
    Plan *p = ref_feature_bl ? e->steady_p.get() : e->first_p.get();     // no BL feature yet = the first P-frame after an I-frame
    LSSVC_CHECK(p != nullptr, "engine_pframe: no %s plan loaded", ref_feature_bl ? "steady-P" : "first-P");
    double s[16];
    if (int rc = run_plan(*p, {{"x_bl", x_bl}, {"x_el", x_el}, {"ref_frame_bl", ref_frame_bl}, {"ref_frame_el", ref_frame_el},
                               {"ref_feature_bl", ref_feature_bl}, {"ref_feature_el", ref_feature_el}},
                          {{"recon_bl", recon_bl}, {"feature_bl", feature_bl}, {"recon_el", recon_el}, {"feature_el", feature_el},
                           {"mv_hat", mv_hat}, {"warp_frame", warp_frame}}, s, e->stream(stream)))
        return rc;
    bits[0] = s[0] + s[1] + s[2] + s[3];                 // dmc_net.py:473
    bits[1] = s[4] + s[5] + s[6] + s[7];                 // LSSVC_net.py:508
    return 0;
}

extern "C" int lssvc_engine_plan_info(void *h, int32_t which, int64_t *out6) {
    LSSVC_CHECK(h && out6, "engine_plan_info: bad arguments");
    Engine *e = static_cast<Engine *>(h);
    Plan *p = which == 0 ? e->intra.get() : (which == 1 ? e->first_p.get() : e->steady_p.get());
    LSSVC_CHECK(p != nullptr, "engine_plan_info: plan %d is not loaded", which);
    uint64_t arena = 0, weights = 0;
    for (auto &g : p->regions) {
        if (g.kind == REGION_ARENA) arena += g.nbytes;
        if (g.kind == REGION_WEIGHTS) weights += g.nbytes;
    }
    out6[0] = (int64_t)p->launches.size();
    out6[1] = p->n_streams;
    out6[2] = (int64_t)arena;
    out6[3] = (int64_t)weights;
    out6[4] = p->H;
    out6[5] = p->W;
    return 0;
}
