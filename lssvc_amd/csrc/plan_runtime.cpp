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
//
// write_stream = 1 plans (compile_iframe_stream / compile_pframe_stream) are the encoder or the decoder half of a frame
// (IntraSS.compress / decompress, src/models/IntraSS.py:304-336 with priors.py:422-452; LSSVC_extend.compress / decompress,
// src/models/LSSVC_net_extend.py:24-136 with dmc_net_extend.py:55-146). Besides launches they hold the HOST steps the front
// end performed between them -- int16 plane copies to / from a pinned staging buffer, rANS coder calls with the CDF tables
// stored in the file -- which the runtime performs in place, in order, on the main stream; such a plan is replayed eagerly
// every frame (a decoder waits for the host a dozen times per frame: there is no graph to capture). The layer files the
// encoder entry points return are byte for byte what the Python path writes (src/utils/stream_helper.py:61-99).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <map>
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
    // weight regions of "LSSVCPL2" plans: not bytes but the RECIPE that rebuilds them from a raw checkpoint
    // (lssvc_prepare_weights); their device memory belongs to the engine's cache and is shared between its plans
    bool has_recipe = false, shared = false;
    lssvc_prep_spec spec{};
    int32_t blob = 0;
    float scalars[4] = {1.f, 1.f, 1.f, 1.f};
};

struct Fix {
    uint32_t field, region;
    uint64_t offset;
};
struct ScalarFix {                       // float at `field` of a struct image = scalars[index] of a recipe region (the 2^-e of its fp16 planes)
    uint32_t field, region, index;
};

// a checkpoint handed to the engine (lssvc_engine_load_checkpoint): owned copies of the tensors + the device blobs prepared
// from them so far, keyed by recipe
struct Prepared {
    std::vector<void *> dev;
    std::vector<int64_t> bytes;
    float scalars[4];
};
struct Checkpoint {
    std::vector<std::string> names;
    std::vector<std::vector<float>> data;
    std::vector<lssvc_tensor> table;
    std::map<std::string, Prepared> cache;
    ~Checkpoint() {
        for (auto &kv : cache)
            for (void *d : kv.second.dev)
                if (d) (void)hipFree(d);
    }
};

struct Arg {
    uint32_t tag = TAG_NULL;
    uint32_t region = 0;
    uint64_t offset = 0;
    float f = 0.f;
    int64_t i = 0;
    std::vector<unsigned char> blob;     // struct image (pointer fields rebased at load) or an int32 array
    std::vector<Fix> fixes;
    std::vector<ScalarFix> sfixes;
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
    FN_FACTORIZED, FN_GAUSSIAN, FN_BOTTLENECK, FN_FILL_ZERO, FN_CLAMP, FN_EXPORT_I16, FN_IMPORT_I16, FN_PAD_CROP, FN_SPYNET_PREP, FN_AVGPOOL_PYRAMID3,
    FN_H_D2H, FN_H_H2D, FN_H_ENCODE, FN_H_FLUSH, FN_H_SET_STREAM, FN_H_DECODE, FN_H_DECODE_CH, FN_H_D2H_ASYNC, FN_H_D2H_WAIT, FN_COUNT
};
const char *const kFnNames[FN_COUNT] = {
    "__wait__", "lssvc_conv2d", "lssvc_conv1x1_dw3x3_f16x3", "lssvc_ffn_f16x3", "lssvc_dwconv3x3", "lssvc_resize_bilinear",
    "lssvc_flow_warp", "lssvc_pool2x2", "lssvc_softmax2_blend", "lssvc_add", "lssvc_copy", "lssvc_lrelu", "lssvc_offset_diversity",
    "lssvc_nchw_to_nhwc", "lssvc_nhwc_to_nchw", "lssvc_laplace_quant_bits", "lssvc_four_part_step", "lssvc_laplace_bits",
    "lssvc_factorized_quant_bits", "lssvc_gaussian_conditional", "lssvc_entropy_bottleneck", "lssvc_fill_zero", "lssvc_clamp_inplace",
    "lssvc_export_symbols_i16", "lssvc_import_symbols_i16", "lssvc_pad_crop", "lssvc_spynet_prep", "lssvc_avgpool_pyramid3",
    "__d2h__", "__h2d__", "__encode__", "__flush__", "__set_stream__", "__decode__", "__decode_ch__", "__d2h_async__", "__d2h_wait__"};

// argument shapes of every replayed entry point, checked when a plan is loaded (a truncated or mismatched file must be a clean
// error, not an out-of-bounds pointer): V = lssvc_view image, C / F = conv / ffn descriptor image, P = device pointer or NULL,
// f / i / l = float / int32 / int64, A = int32 array or NULL, s = the stream slot, w = stream index of a wait
const char *const kFnArgs[FN_COUNT] = {
    "w", "Cs", "CPPs", "Fs", "VPPVs", "VVfs", "VVVs", "VVis", "VVVVs", "VVVs", "VVs", "VVfs", "VVVPPVs", "PVs", "VPs", "VVVVVPPs", "VVVAVVVs", "VVPPs",
    "VPVPPs", "VVVVVPPs", "VPVVPPs", "Pls", "Plffs", "VVAfffiPPPs", "PVPAVs", "VViis", "VVVVs", "VVVVs",
    nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};

struct Table {                           // one entropy_coder.Tables: quantised CDF rows + used lengths + symbol offsets
    std::vector<int32_t> cdfs, sizes, offsets;
    lssvc_cdf_table c{};
};

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
    std::string audit_failed;     // non-empty: the first frame's range audit failed; every later run of this plan returns the same error
    int bits_region = -1;
    // fp16 range audit of the FIRST frame a plan codes with the engine's checkpoint (replay): one float per audited f16x3 launch
    float *audit_dev = nullptr;
    std::vector<std::pair<int, float>> audit_recs;      // (launch index, limit) per slot
    // write_stream = 1 plans
    bool has_host_steps = false;
    std::vector<Table> tables;
    int stage_region = -1, flag_region = -1;
    int16_t *stage_host = nullptr;       // pinned mirror of the staging region
    int32_t *flag_host = nullptr;
    hipEvent_t staged = nullptr;         // behind the asynchronous plane copy of SymbolStage.prefetch
    std::vector<void *> encoders, decoders;
    std::vector<std::pair<const uint8_t *, int64_t>> in_strings;
    std::vector<std::vector<uint8_t>> out_strings;
    std::vector<int16_t> channel_idx;

    ~Plan() {
        for (void *e : encoders)
            if (e) lssvc_rans_encoder_free(e);
        for (void *d : decoders)
            if (d) lssvc_rans_decoder_free(d);
        if (stage_host) (void)hipHostFree(stage_host);
        if (flag_host) (void)hipHostFree(flag_host);
        if (staged) (void)hipEventDestroy(staged);
        if (exec) (void)hipGraphExecDestroy(exec);
        if (audit_dev) (void)hipFree(audit_dev);
        for (auto e : events) (void)hipEventDestroy(e);
        for (auto s : side) (void)hipStreamDestroy(s);
        for (auto &r : regions)
            if (r.ptr && !r.shared) (void)hipFree(r.ptr);
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

// device blobs of one recipe, prepared once per checkpoint
int prepared_for(Checkpoint &ck, const lssvc_prep_spec &spec, Prepared **out) {
    std::string key = std::to_string(spec.kind) + "|" + spec.name + "|" + spec.name2 + "|" + std::to_string(spec.flag);
    for (int i = 0; i < spec.n_splits; ++i) key += "|" + std::to_string(spec.splits[i]);
    auto it = ck.cache.find(key);
    if (it == ck.cache.end()) {
        int32_t nb = 0, dims[8];
        int64_t bytes[LSSVC_PREP_MAX_BLOBS];
        Prepared pr{};
        if (int rc = lssvc_prepare_weights(ck.table.data(), (int32_t)ck.table.size(), &spec, &nb, bytes, pr.scalars, dims, nullptr)) return rc;
        std::vector<std::vector<unsigned char>> host(nb);
        void *ptrs[LSSVC_PREP_MAX_BLOBS] = {nullptr};
        for (int i = 0; i < nb; ++i) {
            host[i].resize((size_t)(bytes[i] ? bytes[i] : 16));
            ptrs[i] = host[i].data();
        }
        if (int rc = lssvc_prepare_weights(ck.table.data(), (int32_t)ck.table.size(), &spec, &nb, bytes, pr.scalars, dims, ptrs)) return rc;
        for (int i = 0; i < nb; ++i) {
            void *d = nullptr;
            LSSVC_HIP(hipMalloc(&d, bytes[i] ? bytes[i] : 16));
            pr.dev.push_back(d);
            pr.bytes.push_back(bytes[i]);
            LSSVC_HIP(hipMemcpy(d, host[i].data(), bytes[i], hipMemcpyHostToDevice));
        }
        it = ck.cache.emplace(key, std::move(pr)).first;
    }
    *out = &it->second;
    return 0;
}

int load_plan(const char *path, Plan &p, Checkpoint *ck) {
    FILE *f = fopen(path, "rb");
    LSSVC_CHECK(f != nullptr, "engine: cannot open plan file %s", path);
    std::unique_ptr<FILE, int (*)(FILE *)> closer(f, fclose);
    Reader r{f};
    char magic[8];
    r.bytes(magic, 8);
    LSSVC_CHECK(r.ok && memcmp(magic, "LSSVCPL2", 8) == 0, "engine: %s is not a frame plan of this format (LSSVCPL2)", path);
    const uint32_t n_regions = r.get<uint32_t>(), n_launches = r.get<uint32_t>();
    LSSVC_CHECK(r.ok && n_regions <= (1u << 16) && n_launches <= (1u << 20), "engine: corrupt plan (%u regions, %u launches)", n_regions, n_launches);
    p.n_streams = r.get<uint32_t>();
    const uint32_t n_meta = r.get<uint32_t>();
    const uint32_t n_tables = r.get<uint32_t>();
    LSSVC_CHECK(r.ok && n_tables <= 64, "engine: corrupt plan (table count)");
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
        if (g.kind == REGION_WEIGHTS) {
            g.has_recipe = r.get<uint32_t>() != 0;
            if (g.has_recipe) {
                g.spec.kind = r.get<int32_t>();
                g.blob = r.get<int32_t>();
                g.spec.n_splits = r.get<int32_t>();
                for (int k = 0; k < 3; ++k) g.spec.splits[k] = r.get<int32_t>();
                g.spec.flag = r.get<int32_t>();
                r.bytes(g.spec.name, sizeof(g.spec.name));
                r.bytes(g.spec.name2, sizeof(g.spec.name2));
                g.spec.name[sizeof(g.spec.name) - 1] = g.spec.name2[sizeof(g.spec.name2) - 1] = 0;
                LSSVC_CHECK(r.ok && g.blob >= 0 && g.blob < LSSVC_PREP_MAX_BLOBS && g.spec.n_splits >= 0 && g.spec.n_splits <= 3, "engine: corrupt plan (weight recipe)");
            }
        }
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
                const uint32_t nsfix = r.get<uint32_t>();
                LSSVC_CHECK(r.ok && nsfix <= 8, "engine: corrupt plan (struct argument)");
                a.sfixes.resize(nsfix);
                for (auto &x : a.sfixes) {
                    x.field = r.get<uint32_t>();
                    x.region = r.get<uint32_t>();
                    x.index = r.get<uint32_t>();
                    LSSVC_CHECK(r.ok && x.index < 4 && (uint64_t)x.field + 4 <= (uint64_t)len && x.region < n_regions, "engine: corrupt plan (scalar fix)");
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
        if (l.id >= FN_H_D2H) {
            p.has_host_steps = true;
            for (auto &a : l.args) LSSVC_CHECK(a.tag == TAG_I64, "engine: corrupt plan (host step argument)");
        }
    }
    for (auto &l : p.launches) {
        const char *sig = kFnArgs[l.id];
        if (!sig) continue;                                                   // host steps: int64 arguments, checked above and in host_step
        LSSVC_CHECK(strlen(sig) == l.args.size(), "engine: corrupt plan (%s with %zu arguments)", l.fn.c_str(), l.args.size());
        for (size_t i = 0; i < l.args.size(); ++i) {
            const Arg &a = l.args[i];
            bool ok = false;
            switch (sig[i]) {
            case 'V': ok = a.tag == TAG_STRUCT && a.blob.size() >= sizeof(lssvc_view); break;
            case 'C': ok = a.tag == TAG_STRUCT && a.blob.size() >= sizeof(lssvc_conv_desc); break;
            case 'F': ok = a.tag == TAG_STRUCT && a.blob.size() >= sizeof(lssvc_ffn_desc); break;
            case 'P': ok = a.tag == TAG_PTR || a.tag == TAG_NULL; break;
            case 'f': ok = a.tag == TAG_F32; break;
            case 'i': ok = a.tag == TAG_I32; break;
            case 'l': ok = a.tag == TAG_I64; break;
            case 'A': ok = a.tag == TAG_NULL || (a.tag == TAG_I32ARRAY && a.blob.size() >= 16); break;
            case 's': ok = a.tag == TAG_STREAM; break;
            case 'w': ok = a.tag == TAG_I32 && a.i >= 0 && a.i < (int64_t)p.n_streams; break;
            }
            LSSVC_CHECK(ok, "engine: corrupt plan (argument %zu of %s)", i, l.fn.c_str());
            if (a.tag == TAG_PTR) LSSVC_CHECK(a.region < n_regions && a.offset <= p.regions[a.region].nbytes, "engine: corrupt plan (pointer outside its region in %s)", l.fn.c_str());
            for (auto &x : a.fixes)
                LSSVC_CHECK(x.region < n_regions && x.offset <= p.regions[x.region].nbytes && (size_t)x.field + 8 <= a.blob.size(),
                            "engine: corrupt plan (struct pointer outside its region in %s)", l.fn.c_str());
        }
    }
    p.tables.resize(n_tables);
    for (auto &t : p.tables) {
        const uint32_t rows = r.get<uint32_t>(), stride = r.get<uint32_t>();
        LSSVC_CHECK(r.ok && rows > 0 && rows <= 4096 && stride > 0 && stride <= 65538, "engine: corrupt plan (table shape)");
        t.cdfs.resize((size_t)rows * stride);
        t.sizes.resize(rows);
        t.offsets.resize(rows);
        r.bytes(t.cdfs.data(), t.cdfs.size() * 4);
        r.bytes(t.sizes.data(), rows * 4);
        r.bytes(t.offsets.data(), rows * 4);
        t.c.cdfs = t.cdfs.data();
        t.c.n_cdfs = (int32_t)rows;
        t.c.stride = (int32_t)stride;
        t.c.sizes = t.sizes.data();
        t.c.offsets = t.offsets.data();
    }
    LSSVC_CHECK(r.ok, "engine: truncated plan file %s", path);
    // ---- regions: allocate, upload the weights (payloads follow the launch list, 256-byte aligned, in region order)
    std::vector<unsigned char> host;
    for (auto &g : p.regions) {
        if (g.kind == REGION_WEIGHTS && g.has_recipe) {        // rebuilt from the caller's checkpoint, shared between this engine's plans
            LSSVC_CHECK(ck != nullptr, "engine: %s refers to checkpoint tensors ('%s'): call lssvc_engine_load_checkpoint for this model first", path, g.spec.name);
            Prepared *pr = nullptr;
            if (int rc = prepared_for(*ck, g.spec, &pr)) return rc;
            LSSVC_CHECK((size_t)g.blob < pr->dev.size() && (uint64_t)pr->bytes[g.blob] == g.nbytes,
                        "engine: layer '%s' of the checkpoint does not have the shape %s was compiled for (%llu bytes expected, %lld prepared)", g.spec.name, path,
                        (unsigned long long)g.nbytes, (size_t)g.blob < pr->bytes.size() ? (long long)pr->bytes[g.blob] : -1LL);
            g.ptr = pr->dev[g.blob];
            g.shared = true;
            memcpy(g.scalars, pr->scalars, sizeof(g.scalars));
            continue;
        }
        LSSVC_HIP(hipMalloc(&g.ptr, g.nbytes ? g.nbytes : 16));
        if (g.kind == REGION_SCRATCH) LSSVC_HIP(hipMemset(g.ptr, 0, g.nbytes));
        if (g.kind == REGION_WEIGHTS) {
            const long pos = ftell(f);
            LSSVC_CHECK(pos >= 0 && fseek(f, (256 - pos % 256) % 256, SEEK_CUR) == 0, "engine: cannot seek in %s", path);
            host.resize(g.nbytes);
            r.bytes(host.data(), g.nbytes);
            LSSVC_CHECK(r.ok, "engine: truncated plan file %s (weights)", path);
            LSSVC_HIP(hipMemcpy(g.ptr, host.data(), g.nbytes, hipMemcpyHostToDevice));
        }
    }
    p.bits_region = p.region_index("bits", REGION_SCRATCH);
    LSSVC_CHECK(p.bits_region >= 0, "engine: plan has no bit-counter region");
    if (p.has_host_steps) {
        p.stage_region = p.region_index("stage_dev", REGION_SCRATCH);
        p.flag_region = p.region_index("stage_flag", REGION_SCRATCH);
        LSSVC_CHECK(p.stage_region >= 0 && p.flag_region >= 0, "engine: a plan with host steps needs the staging regions");
        LSSVC_HIP(hipHostMalloc((void **)&p.stage_host, p.regions[p.stage_region].nbytes, hipHostMallocDefault));
        LSSVC_HIP(hipHostMalloc((void **)&p.flag_host, sizeof(int32_t), hipHostMallocDefault));
        LSSVC_HIP(hipEventCreateWithFlags(&p.staged, hipEventDisableTiming));
    }
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
                    LSSVC_CHECK(x.region < p.regions.size() && p.regions[x.region].ptr && (uint64_t)x.field + 8 <= (uint64_t)a.blob.size(),
                                "engine: unbound region %u", x.region);
                    void *q = static_cast<char *>(p.regions[x.region].ptr) + x.offset;
                    memcpy(a.blob.data() + x.field, &q, 8);
                }
                for (auto &x : a.sfixes) {
                    const Region &g = p.regions[x.region];
                    LSSVC_CHECK(g.has_recipe, "engine: scalar fix on a region without a recipe");
                    memcpy(a.blob.data() + x.field, &g.scalars[x.index], 4);
                }
            }
        }
    return 0;
}

// One host step of a write_stream = 1 plan (entropy_coder.py / hip_ops.SymbolStage, as recorded): plane offsets are int16
// element offsets into the staging region and its pinned mirror.
int host_step(Plan &p, const Launch &l, hipStream_t main) {
    auto A = [&](size_t i) { return l.args[i].i; };
    const int64_t cap = (int64_t)(p.regions[p.stage_region].nbytes / 2);
    int16_t *dev = static_cast<int16_t *>(p.regions[p.stage_region].ptr);
    auto in_stage = [&](int64_t off, int64_t n) { return off >= 0 && n >= 0 && off + n <= cap; };
    auto coder = [&](std::vector<void *> &v, int64_t id, bool enc) -> void * {
        if (id < 0 || id >= 16) return nullptr;
        if ((size_t)id >= v.size()) v.resize(id + 1, nullptr);
        if (!v[id]) v[id] = enc ? lssvc_rans_encoder_new() : lssvc_rans_decoder_new();
        return v[id];
    };
    auto table = [&](int64_t id) -> const lssvc_cdf_table * { return id >= 0 && (size_t)id < p.tables.size() ? &p.tables[id].c : nullptr; };
    switch (l.id) {
    case FN_H_D2H: {                                     // SymbolStage.download: planes [lo, hi) + the overflow flag, then wait
        LSSVC_CHECK(l.args.size() == 2 && in_stage(A(0), A(1) - A(0)), "engine: corrupt plan (d2h step)");
        LSSVC_HIP(hipMemcpyAsync(p.stage_host + A(0), dev + A(0), 2 * (size_t)(A(1) - A(0)), hipMemcpyDeviceToHost, main));
        LSSVC_HIP(hipMemcpyAsync(p.flag_host, p.regions[p.flag_region].ptr, sizeof(int32_t), hipMemcpyDeviceToHost, main));
        LSSVC_HIP(hipStreamSynchronize(main));
        LSSVC_CHECK(*p.flag_host == 0, "engine: a quantised latent does not fit the 16-bit symbol planes");
        return 0;
    }
    case FN_H_D2H_ASYNC:                                 // SymbolStage.prefetch: the same copy, not waited for; the kernels that follow overlap the host coder
        LSSVC_CHECK(l.args.size() == 2 && in_stage(A(0), A(1) - A(0)), "engine: corrupt plan (d2h step)");
        LSSVC_HIP(hipMemcpyAsync(p.stage_host + A(0), dev + A(0), 2 * (size_t)(A(1) - A(0)), hipMemcpyDeviceToHost, main));
        LSSVC_HIP(hipMemcpyAsync(p.flag_host, p.regions[p.flag_region].ptr, sizeof(int32_t), hipMemcpyDeviceToHost, main));
        LSSVC_HIP(hipEventRecord(p.staged, main));
        return 0;
    case FN_H_D2H_WAIT:                                  // ... SymbolStage.download of a prefetched range: wait for that copy only
        LSSVC_HIP(hipEventSynchronize(p.staged));
        LSSVC_CHECK(*p.flag_host == 0, "engine: a quantised latent does not fit the 16-bit symbol planes");
        return 0;
    case FN_H_H2D:                                       // SymbolStage.upload
        LSSVC_CHECK(l.args.size() == 2 && in_stage(A(0), A(1)), "engine: corrupt plan (h2d step)");
        LSSVC_HIP(hipMemcpyAsync(dev + A(0), p.stage_host + A(0), 2 * (size_t)A(1), hipMemcpyHostToDevice, main));
        return 0;
    case FN_H_ENCODE: {                                  // RansEncoder.encode_with_indexes(symbols, indexes, tables)
        LSSVC_CHECK(l.args.size() == 5 && in_stage(A(1), A(3)) && in_stage(A(2), A(3)) && table(A(4)), "engine: corrupt plan (encode step)");
        void *e = coder(p.encoders, A(0), true);
        LSSVC_CHECK(e != nullptr, "engine: corrupt plan (encoder id)");
        return lssvc_rans_encode_with_indexes_i16(e, p.stage_host + A(1), p.stage_host + A(2), A(3), table(A(4)));
    }
    case FN_H_FLUSH: {                                   // RansEncoder.flush -> string A(1) of the frame
        LSSVC_CHECK(l.args.size() == 2 && A(1) >= 0 && A(1) < 8, "engine: corrupt plan (flush step)");
        void *e = coder(p.encoders, A(0), true);
        LSSVC_CHECK(e != nullptr, "engine: corrupt plan (encoder id)");
        const int64_t n = lssvc_rans_encoder_flush(e);
        LSSVC_CHECK(n >= 0, "engine: rANS flush failed");
        if ((size_t)A(1) >= p.out_strings.size()) p.out_strings.resize(A(1) + 1);
        const uint8_t *b = lssvc_rans_encoder_bytes(e);
        p.out_strings[A(1)].assign(b, b + n);
        lssvc_rans_encoder_reset(e);
        return 0;
    }
    case FN_H_SET_STREAM: {                              // RansDecoder.set_stream(string A(1) of the layer files)
        LSSVC_CHECK(l.args.size() == 2 && A(1) >= 0 && (size_t)A(1) < p.in_strings.size(), "engine: this plan reads string %lld, %zu were given",
                    (long long)A(1), p.in_strings.size());
        void *d = coder(p.decoders, A(0), false);
        LSSVC_CHECK(d != nullptr, "engine: corrupt plan (decoder id)");
        return lssvc_rans_decoder_set_stream(d, p.in_strings[A(1)].first, p.in_strings[A(1)].second);
    }
    case FN_H_DECODE: {                                  // RansDecoder.decode_stream(staged index plane) -> staged symbol plane
        LSSVC_CHECK(l.args.size() == 5 && in_stage(A(1), A(2)) && in_stage(A(4), A(2)) && table(A(3)), "engine: corrupt plan (decode step)");
        void *d = coder(p.decoders, A(0), false);
        LSSVC_CHECK(d != nullptr, "engine: corrupt plan (decoder id)");
        LSSVC_HIP(hipStreamSynchronize(main));           // (earlier uploads out of the pinned buffer have left it)
        return lssvc_rans_decode_stream_i16(d, p.stage_host + A(1), A(2), table(A(3)), p.stage_host + A(4));
    }
    case FN_H_DECODE_CH: {                               // ... with index = channel number, NCHW order (factorised tables)
        LSSVC_CHECK(l.args.size() == 5 && A(1) > 0 && A(1) <= 4096 && A(2) > 0 && in_stage(A(4), A(1) * A(2)) && table(A(3)),
                    "engine: corrupt plan (decode step)");
        void *d = coder(p.decoders, A(0), false);
        LSSVC_CHECK(d != nullptr, "engine: corrupt plan (decoder id)");
        p.channel_idx.resize((size_t)(A(1) * A(2)));
        for (int64_t c = 0; c < A(1); ++c)
            for (int64_t i = 0; i < A(2); ++i) p.channel_idx[(size_t)(c * A(2) + i)] = (int16_t)c;
        LSSVC_HIP(hipStreamSynchronize(main));
        return lssvc_rans_decode_stream_i16(d, p.channel_idx.data(), A(1) * A(2), table(A(3)), p.stage_host + A(4));
    }
    default: return fail("engine: no host step for %s", l.fn.c_str());
    }
}

// fp16 range of the f16x3 kernels' inputs (hip_ops.RangeAudit): they split activations into fp16 hi / lo parts and saturate at
// +-65504 (GDN's 1x1 squares first), which the reference's fp32 convs do not. The Python front end audits the first frame of every
// type and moves layers that come within 2x of the limit to the exact fp32 kernel; a plan bakes that kernel choice in. The engine
// binds plans to ANY checkpoint of the architecture, so the first frame a plan codes is audited here as well: max |x| of every
// input view of its f16x3 conv / fused-DepthConv / FFN launches (the tensors INSIDE the fused kernels are not visible here; the
// Python audit splits those kernels). A value over the limit is an error that says so -- recompile the plans from this checkpoint
// -- instead of a silently saturated frame.
constexpr float kF16InputLimit = 32768.0f, kF16SquareInputLimit = 181.01934f;      // 2^15, 2^7.5: hip_ops.F16_INPUT_LIMIT / F16_SQUARE_INPUT_LIMIT

static int audit_views(Plan &p, int launch, const lssvc_view *const *views, int n, float limit, hipStream_t st, size_t &slot) {
    if (slot >= p.audit_recs.size()) p.audit_recs.emplace_back(launch, limit);
    for (int i = 0; i < n; ++i)
        if (views[i] && views[i]->ptr)
            if (int rc = lssvc_absmax(views[i], p.audit_dev + slot, st)) return rc;
    ++slot;
    return 0;
}

int replay(Plan &p, hipStream_t main, bool audit = false) {
    auto stream_of = [&](uint32_t s) { return s == 0 ? main : p.side[s - 1]; };
    size_t ev = 0, slot = 0;
    if (audit) {
        if (!p.audit_dev) LSSVC_HIP(hipMalloc(&p.audit_dev, sizeof(float) * (p.launches.size() + 1)));
        LSSVC_HIP(hipMemsetAsync(p.audit_dev, 0, sizeof(float) * (p.launches.size() + 1), main));
        p.audit_recs.clear();
        LSSVC_HIP(hipStreamSynchronize(main));      // (the side streams' first launches may come before main's next one: the zeroing must have landed)
    }
    int li = -1;
    for (auto &l : p.launches) {
        hipStream_t st = stream_of(l.stream);
        ++li;
        if (audit && (l.id == FN_CONV2D || l.id == FN_CONV1X1_DW)) {
            const lssvc_conv_desc *d = reinterpret_cast<const lssvc_conv_desc *>(l.args[0].blob.data());
            if ((d->precision & LSSVC_PREC_MASK) == LSSVC_PREC_F16X3 && !(d->precision & LSSVC_PREC_SPLIT_IN)) {
                const lssvc_view *v[LSSVC_CONV_MAX_INPUTS] = {nullptr, nullptr, nullptr};
                for (int i = 0; i < d->n_in && i < LSSVC_CONV_MAX_INPUTS; ++i) v[i] = &d->in[i];
                if (int rc = audit_views(p, li, v, LSSVC_CONV_MAX_INPUTS, d->in_act == LSSVC_INACT_SQUARE ? kF16SquareInputLimit : kF16InputLimit, st, slot)) return rc;
            }
        } else if (audit && l.id == FN_FFN) {
            const lssvc_ffn_desc *d = reinterpret_cast<const lssvc_ffn_desc *>(l.args[0].blob.data());
            const lssvc_view *v[3] = {&d->x, &d->pre_in, &d->ident};
            if (int rc = audit_views(p, li, v, 3, kF16InputLimit, st, slot)) return rc;
        }
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
        case FN_EXPORT_I16:
            rc = lssvc_export_symbols_i16(V(0), V(1), l.args[2].tag == TAG_I32ARRAY ? reinterpret_cast<const int32_t *>(l.args[2].blob.data()) : nullptr,
                                          F(3), F(4), F(5), I(6), (int16_t *)P(7), (int16_t *)P(8), (int32_t *)P(9), st);
            break;
        case FN_IMPORT_I16:
            rc = lssvc_import_symbols_i16((const int16_t *)P(0), V(1), (const float *)P(2),
                                          l.args[3].tag == TAG_I32ARRAY ? reinterpret_cast<const int32_t *>(l.args[3].blob.data()) : nullptr, V(4), st);
            break;
        case FN_PAD_CROP: rc = lssvc_pad_crop(V(0), V(1), I(2), I(3), st); break;
        case FN_SPYNET_PREP: rc = lssvc_spynet_prep(V(0), V(1), V(2), V(3), st); break;
        case FN_AVGPOOL_PYRAMID3: rc = lssvc_avgpool_pyramid3(V(0), V(1), V(2), V(3), st); break;
        default:
            if (l.id >= FN_H_D2H) {
                rc = host_step(p, l, main);
                break;
            }
            return fail("engine: no replay for %s", l.fn.c_str());
        }
        if (rc) return rc;
    }
    return 0;
}

struct Engine {
    int device = 0;
    hipStream_t own = nullptr;           // frames run here when the caller passes stream NULL (the null stream cannot be captured)
    std::unique_ptr<Plan> intra, first_p, steady_p;
    std::unique_ptr<Plan> i_enc, i_dec, p1_enc, p1_dec, p_enc, p_dec;      // write_stream = 1 halves
    // round 6: the P-frame as a base-layer plan and an enhancement-layer plan (plan_compiler.compile_pframe_layers), [0] first-P, [1] steady-P;
    // lssvc_engine_pframe_lookahead runs EL(t) on the caller's stream and BL(t+1) on `ahead_stream`
    std::unique_ptr<Plan> bl_plan[2], el_plan[2];
    hipStream_t ahead_stream = nullptr;
    hipEvent_t ev_handed = nullptr, ev_ahead = nullptr;   // BL(t) handed to the EL plan / BL(t+1) finished
    Plan *ahead = nullptr;                                // the base-layer plan whose outputs hold the NEXT frame's base layer (coded ahead), or null
    double ahead_bits[16] = {0};
    double *ahead_slots_host = nullptr;                   // pinned: the look-ahead base layer's bit slots come down asynchronously
    std::unique_ptr<Checkpoint> ckpt[2];                                    // [0] IntraSS, [1] LSSVC: raw tensors + prepared device weights
    ~Engine() {
        intra.reset(), first_p.reset(), steady_p.reset();
        i_enc.reset(), i_dec.reset(), p1_enc.reset(), p1_dec.reset(), p_enc.reset(), p_dec.reset();
        for (int i = 0; i < 2; ++i) bl_plan[i].reset(), el_plan[i].reset();
        ckpt[0].reset(), ckpt[1].reset();
        if (ahead_slots_host) (void)hipHostFree(ahead_slots_host);
        if (ev_handed) (void)hipEventDestroy(ev_handed);
        if (ev_ahead) (void)hipEventDestroy(ev_ahead);
        if (ahead_stream) (void)hipStreamDestroy(ahead_stream);
        if (own) (void)hipStreamDestroy(own);
    }
    hipStream_t stream(void *s) const { return s ? (hipStream_t)s : own; }
};

// after the audited first frame has finished: every audited launch's max |input| against its limit
static int audit_verdict(Plan &p, hipStream_t st) {
    if (p.audit_recs.empty()) return 0;
    std::vector<float> v(p.audit_recs.size());
    LSSVC_HIP(hipMemcpyAsync(v.data(), p.audit_dev, sizeof(float) * v.size(), hipMemcpyDeviceToHost, st));
    LSSVC_HIP(hipStreamSynchronize(st));
    for (size_t i = 0; i < v.size(); ++i)
        if (!(v[i] < p.audit_recs[i].second))
            return fail("engine: with this checkpoint the input of launch %d of the '%s' plan (%s) reaches |x| = %g, outside what its fp16-split kernel can hold "
                        "(limit %g): the plan's kernel choice was audited against another checkpoint -- compile the plans from this one (its range audit moves "
                        "such layers to the exact fp32 kernel)", p.audit_recs[i].first, p.kind.c_str(), p.launches[(size_t)p.audit_recs[i].first].fn.c_str(), (double)v[i],
                        (double)p.audit_recs[i].second);
    return 0;
}

// Run one plan: caller's inputs in, first call eager, second call capture, later calls hipGraphLaunch, outputs out.
// no_sync (round 6, the look-ahead base layer): once the plan replays as a graph the call returns with the work queued -- `slots16` must
// then be pinned host memory, filled when the stream gets there; the eager first call and the capturing second call stay synchronous
int run_plan(Plan &p, const std::vector<std::pair<const char *, const void *>> &ins,
             const std::vector<std::pair<const char *, void *>> &outs, double *slots16, hipStream_t st, bool no_sync = false) {
    for (auto &kv : ins) {
        const int i = p.region_index(kv.first, REGION_INPUT);
        if (i < 0) {
            LSSVC_CHECK(kv.second == nullptr, "engine: this plan takes no input named %s", kv.first);
            continue;
        }
        LSSVC_CHECK(kv.second != nullptr, "engine: input %s is required by this plan", kv.first);
        LSSVC_HIP(hipMemcpyAsync(p.regions[i].ptr, kv.second, p.regions[i].nbytes, hipMemcpyDefault, st));
    }
    LSSVC_CHECK(!p.has_host_steps, "engine: a '%s' plan runs through the encode / decode entry points", p.kind.c_str());
    LSSVC_CHECK(p.audit_failed.empty(), "%s", p.audit_failed.c_str());
    const bool audit = p.runs == 0 && !std::getenv("LSSVC_ENGINE_NO_AUDIT");
    if (p.runs == 0 || std::getenv("LSSVC_ENGINE_EAGER")) {
        if (int e = replay(p, st, audit)) return e;
    } else {
        if (!p.exec) {
            hipGraph_t graph = nullptr;
            LSSVC_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            const int rc = replay(p, st);
            const hipError_t ce = hipStreamEndCapture(st, &graph);
            if (rc) {
                if (graph) (void)hipGraphDestroy(graph);
                return rc;
            }
            LSSVC_CHECK(ce == hipSuccess && graph, "engine: hipStreamEndCapture: %s", hipGetErrorString(ce));
            const hipError_t ie = hipGraphInstantiate(&p.exec, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            LSSVC_CHECK(ie == hipSuccess, "engine: hipGraphInstantiate: %s", hipGetErrorString(ie));
        }
        LSSVC_HIP(hipGraphLaunch(p.exec, st));
    }
    // the verdict comes BEFORE the run is counted and before anything is handed to the caller: a failed audit leaves runs == 0 and is
    // sticky, so a caller that carries on after the error gets the same error again, never an un-audited captured replay
    if (audit)
        if (int e = audit_verdict(p, st)) {
            p.audit_failed = err_buf();
            return e;
        }
    ++p.runs;
    for (auto &kv : outs) {
        if (!kv.second) continue;                        // the caller does not want this one
        const int i = p.region_index(kv.first, REGION_OUTPUT);
        LSSVC_CHECK(i >= 0, "engine: this plan has no output named %s", kv.first);
        LSSVC_HIP(hipMemcpyAsync(kv.second, p.regions[i].ptr, p.regions[i].nbytes, hipMemcpyDefault, st));
    }
    LSSVC_HIP(hipMemcpyAsync(slots16, p.regions[p.bits_region].ptr, 16 * sizeof(double), hipMemcpyDeviceToHost, st));
    if (no_sync && p.runs > 2) return 0;                 // (replaying: the caller waits for the stream when it needs the counts)
    LSSVC_HIP(hipStreamSynchronize(st));                 // the bit counts are host values, as in the reference (.item())
    return 0;
}

// Run the encoder or decoder half of a frame: inputs in, eager replay with its host steps, outputs out; the strings the
// decoder reads are in p.in_strings, the strings the encoder wrote come back in p.out_strings.
int run_stream_plan(Plan &p, const std::vector<std::pair<const char *, const void *>> &ins,
                    const std::vector<std::pair<const char *, void *>> &outs, hipStream_t st) {
    LSSVC_CHECK(p.has_host_steps, "engine: a '%s' plan has no coder steps", p.kind.c_str());
    LSSVC_CHECK(p.audit_failed.empty(), "%s", p.audit_failed.c_str());
    for (auto &kv : ins) {
        const int i = p.region_index(kv.first, REGION_INPUT);
        if (i < 0) {
            LSSVC_CHECK(kv.second == nullptr, "engine: this plan takes no input named %s", kv.first);
            continue;
        }
        LSSVC_CHECK(kv.second != nullptr, "engine: input %s is required by this plan", kv.first);
        LSSVC_HIP(hipMemcpyAsync(p.regions[i].ptr, kv.second, p.regions[i].nbytes, hipMemcpyDefault, st));
    }
    p.out_strings.clear();
    for (void *e : p.encoders)
        if (e) lssvc_rans_encoder_reset(e);
    const bool audit = p.runs == 0 && !std::getenv("LSSVC_ENGINE_NO_AUDIT");
    if (int e = replay(p, st, audit)) return e;
    if (audit)
        if (int e = audit_verdict(p, st)) {
            p.audit_failed = err_buf();
            return e;
        }
    ++p.runs;
    for (auto &kv : outs) {
        if (!kv.second) continue;
        const int i = p.region_index(kv.first, REGION_OUTPUT);
        LSSVC_CHECK(i >= 0, "engine: this plan has no output named %s", kv.first);
        LSSVC_HIP(hipMemcpyAsync(kv.second, p.regions[i].ptr, p.regions[i].nbytes, hipMemcpyDefault, st));
    }
    LSSVC_HIP(hipStreamSynchronize(st));
    return 0;
}

// ---- layer files (src/utils/stream_helper.py:61-99; lssvc_amd/bitstream.py): big-endian u32 headers
void put_be32(uint8_t *d, uint64_t v) {
    d[0] = (uint8_t)(v >> 24), d[1] = (uint8_t)(v >> 16), d[2] = (uint8_t)(v >> 8), d[3] = (uint8_t)v;
}
uint32_t get_be32(const uint8_t *d) { return ((uint32_t)d[0] << 24) | ((uint32_t)d[1] << 16) | ((uint32_t)d[2] << 8) | d[3]; }

// I-frame layer file = (height, width, len_y, len_z) + y string + z string
int write_i_file(uint8_t *dst, int64_t cap, int64_t *len, int64_t height, int64_t width, const std::vector<uint8_t> &y, const std::vector<uint8_t> &z) {
    const int64_t need = 16 + (int64_t)y.size() + (int64_t)z.size();
    *len = need;
    LSSVC_CHECK(dst && cap >= need, "engine: the layer file needs %lld bytes, the buffer holds %lld", (long long)need, (long long)cap);
    put_be32(dst, (uint64_t)height), put_be32(dst + 4, (uint64_t)width), put_be32(dst + 8, y.size()), put_be32(dst + 12, z.size());
    memcpy(dst + 16, y.data(), y.size());
    memcpy(dst + 16 + y.size(), z.data(), z.size());
    return 0;
}
// P-frame layer file = (len) + one rANS string
int write_p_file(uint8_t *dst, int64_t cap, int64_t *len, const std::vector<uint8_t> &s) {
    const int64_t need = 4 + (int64_t)s.size();
    *len = need;
    LSSVC_CHECK(dst && cap >= need, "engine: the layer file needs %lld bytes, the buffer holds %lld", (long long)need, (long long)cap);
    put_be32(dst, s.size());
    memcpy(dst + 4, s.data(), s.size());
    return 0;
}
int read_i_file(const uint8_t *f, int64_t n, int64_t want_h, int64_t want_w, std::vector<std::pair<const uint8_t *, int64_t>> &strings) {
    LSSVC_CHECK(f && n >= 16, "engine: truncated I-frame stream");
    const int64_t h = get_be32(f), w = get_be32(f + 4), ly = get_be32(f + 8), lz = get_be32(f + 12);
    LSSVC_CHECK(16 + ly + lz <= n, "engine: truncated I-frame stream");
    LSSVC_CHECK(h == want_h && w == want_w, "engine: the stream codes a %lldx%lld picture, the loaded plan was compiled for %lldx%lld",
                (long long)h, (long long)w, (long long)want_h, (long long)want_w);
    strings.emplace_back(f + 16, ly);
    strings.emplace_back(f + 16 + ly, lz);
    return 0;
}
int read_p_file(const uint8_t *f, int64_t n, std::vector<std::pair<const uint8_t *, int64_t>> &strings) {
    LSSVC_CHECK(f && n >= 4, "engine: truncated P-frame stream");
    const int64_t len = get_be32(f);
    LSSVC_CHECK(4 + len <= n, "engine: truncated P-frame stream");
    strings.emplace_back(f + 4, len);
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

// CRC-32 (zlib's polynomial) over the tensors the CDF tables are a function of: plan_compiler.py entropy_params_crc
static uint32_t crc32_update(uint32_t crc, const unsigned char *p, size_t n) {
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        init = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xFF] ^ (crc >> 8);
    return ~crc;
}
static int64_t entropy_params_crc(const Checkpoint &ck) {
    std::vector<size_t> idx;
    for (size_t i = 0; i < ck.names.size(); ++i)
        if (ck.names[i].find("bit_estimator") != std::string::npos || ck.names[i].find("entropy_bottleneck") != std::string::npos) idx.push_back(i);
    std::sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return ck.names[a] < ck.names[b]; });
    uint32_t crc = 0;
    for (size_t i : idx) {
        crc = crc32_update(crc, reinterpret_cast<const unsigned char *>(ck.names[i].data()), ck.names[i].size());
        crc = crc32_update(crc, reinterpret_cast<const unsigned char *>(ck.data[i].data()), ck.data[i].size() * sizeof(float));
    }
    return (int64_t)crc;
}

static int load_into(Engine *eng, std::unique_ptr<Plan> &slot, const char *path, const char *want_a, const char *want_b) {
    std::unique_ptr<Plan> p(new Plan());
    const bool intra_model = want_a[0] == 'i';                  // "iframe*" plans run IntraSS, "pframe*" plans LSSVC
    const Checkpoint *ck = eng->ckpt[intra_model ? 0 : 1].get();
    if (int e = load_plan(path, *p, eng->ckpt[intra_model ? 0 : 1].get())) return e;
    LSSVC_CHECK(p->kind == want_a || (want_b && p->kind == want_b), "engine: %s holds a '%s' plan", path, p->kind.c_str());
    // a write_stream plan holds the CDF tables update() built from the checkpoint it was compiled with (and the bottleneck medians);
    // bound to another checkpoint its strings would be coded against the wrong tables and decode nowhere else
    if (const int64_t want = p->meta_value("entropy_params_crc", -1); want >= 0 && ck) {
        const int64_t have = entropy_params_crc(*ck);
        LSSVC_CHECK(have == want, "engine: %s was compiled with another checkpoint's entropy parameters (CRC %08llx, this checkpoint %08llx): its CDF tables "
                    "do not belong to these weights -- compile the stream plans from this checkpoint", path, (unsigned long long)want, (unsigned long long)have);
    }
    slot = std::move(p);
    return 0;
}

extern "C" int lssvc_engine_load_checkpoint(void *h, int32_t model, const lssvc_tensor *tensors, int32_t n) {
    LSSVC_CHECK(h && tensors && n > 0 && (model == 0 || model == 1), "engine_load_checkpoint: bad arguments (model 0 = IntraSS, 1 = LSSVC)");
    Engine *e = static_cast<Engine *>(h);
    LSSVC_CHECK(!(model == 0 ? (e->intra || e->i_enc || e->i_dec) : (e->first_p || e->steady_p || e->p1_enc || e->p1_dec || e->p_enc || e->p_dec)),
                "engine_load_checkpoint: plans of this model are already loaded with another checkpoint");
    LSSVC_HIP(hipSetDevice(e->device));
    std::unique_ptr<Checkpoint> ck(new Checkpoint());
    ck->names.reserve(n), ck->data.reserve(n), ck->table.reserve(n);
    for (int32_t i = 0; i < n; ++i) {
        const lssvc_tensor &t = tensors[i];
        LSSVC_CHECK(t.name && t.data && t.ndim >= 0 && t.ndim <= 4, "engine_load_checkpoint: bad tensor %d", i);
        int64_t numel = 1;
        for (int d = 0; d < t.ndim; ++d) {
            LSSVC_CHECK(t.shape[d] >= 0 && t.shape[d] < (1LL << 31), "engine_load_checkpoint: bad shape of '%s'", t.name);
            numel *= t.shape[d];
        }
        const char *nm = strncmp(t.name, "module.", 7) == 0 ? t.name + 7 : t.name;       // IntraSS.py:193-198, LSSVC_net.py:141-149
        ck->names.emplace_back(nm);
        ck->data.emplace_back(t.data, t.data + numel);
    }
    for (int32_t i = 0; i < n; ++i) {
        lssvc_tensor t = tensors[i];
        t.name = ck->names[i].c_str();
        t.data = ck->data[i].data();
        ck->table.push_back(t);
    }
    e->ckpt[model] = std::move(ck);
    return 0;
}

extern "C" int lssvc_engine_load_intra(void *h, const char *iframe_plan) {
    LSSVC_CHECK(h && iframe_plan, "engine_load_intra: bad arguments");
    Engine *e = static_cast<Engine *>(h);
    LSSVC_HIP(hipSetDevice(e->device));
    return load_into(e, e->intra, iframe_plan, "iframe", nullptr);
}

extern "C" int lssvc_engine_load_inter(void *h, const char *first_p_plan, const char *steady_p_plan) {
    LSSVC_CHECK(h && first_p_plan && steady_p_plan, "engine_load_inter: bad arguments");
    Engine *e = static_cast<Engine *>(h);
    LSSVC_HIP(hipSetDevice(e->device));
    if (int rc = load_into(e, e->first_p, first_p_plan, "pframe_first", nullptr)) return rc;
    return load_into(e, e->steady_p, steady_p_plan, "pframe", nullptr);
}

// Round 6: the four plans of the look-ahead protocol (plan_compiler.compile_pframe_layers): base layer / enhancement layer of the first
// P-frame after an I-frame and of a steady-state P-frame.
extern "C" int lssvc_engine_load_inter_layers(void *h, const char *bl_first, const char *bl_steady, const char *el_first, const char *el_steady) {
    LSSVC_CHECK(h && bl_first && bl_steady && el_first && el_steady, "engine_load_inter_layers: bad arguments");
    Engine *e = static_cast<Engine *>(h);
    LSSVC_HIP(hipSetDevice(e->device));
    if (int rc = load_into(e, e->bl_plan[0], bl_first, "pframe_first_bl", nullptr)) return rc;
    if (int rc = load_into(e, e->bl_plan[1], bl_steady, "pframe_bl", nullptr)) return rc;
    if (int rc = load_into(e, e->el_plan[0], el_first, "pframe_first_el", nullptr)) return rc;
    if (int rc = load_into(e, e->el_plan[1], el_steady, "pframe_el", nullptr)) return rc;
    // what the base-layer plans hand over must be what the enhancement-layer plans were compiled to read
    for (int i = 0; i < 2; ++i)
        for (const char *k : {"bl_recon", "bl_feature", "bl_y_hat", "bl_mv_hat"})
            for (int j = 0; j < 2; ++j) {
                const int a = e->bl_plan[i]->region_index(k, REGION_OUTPUT), b = e->el_plan[j]->region_index(k, REGION_INPUT);
                LSSVC_CHECK(a >= 0 && b >= 0 && e->bl_plan[i]->regions[a].nbytes == e->el_plan[j]->regions[b].nbytes,
                            "engine_load_inter_layers: the plans do not agree on %s (compile the four from one model at one size)", k);
            }
    if (!e->ahead_stream) LSSVC_HIP(hipStreamCreateWithFlags(&e->ahead_stream, hipStreamNonBlocking));
    if (!e->ev_handed) LSSVC_HIP(hipEventCreateWithFlags(&e->ev_handed, hipEventDisableTiming));
    if (!e->ev_ahead) LSSVC_HIP(hipEventCreateWithFlags(&e->ev_ahead, hipEventDisableTiming));
    if (!e->ahead_slots_host) LSSVC_HIP(hipHostMalloc(reinterpret_cast<void **>(&e->ahead_slots_host), 16 * sizeof(double), hipHostMallocDefault));
    e->ahead = nullptr;
    return 0;
}

extern "C" int lssvc_engine_lookahead_reset(void *h) {
    LSSVC_CHECK(h, "engine_lookahead_reset: bad arguments");
    Engine *e = static_cast<Engine *>(h);
    if (e->ahead_stream) LSSVC_HIP(hipStreamSynchronize(e->ahead_stream));
    e->ahead = nullptr;
    return 0;
}

extern "C" int lssvc_engine_set_scale(void *h, float scale, int32_t H, int32_t W) {
    LSSVC_CHECK(h, "engine_set_scale: bad arguments");
    Engine *e = static_cast<Engine *>(h);
    Plan *const all[13] = {e->intra.get(), e->first_p.get(), e->steady_p.get(), e->i_enc.get(), e->i_dec.get(), e->p1_enc.get(), e->p1_dec.get(),
                           e->p_enc.get(), e->p_dec.get(), e->bl_plan[0].get(), e->bl_plan[1].get(), e->el_plan[0].get(), e->el_plan[1].get()};
    const Plan *first = nullptr;
    for (Plan *p : all) {
        if (!p) continue;
        LSSVC_CHECK(std::fabs(p->scale - (double)scale) < 1e-9 && p->H == H && p->W == W,
                    "engine_set_scale: the loaded '%s' plan was compiled for scale %g, %dx%d (asked for %g, %dx%d)", p->kind.c_str(),
                    p->scale, p->H, p->W, (double)scale, H, W);
        // the inter-layer padding (set_scale_information's pad_size) is baked into a plan's launches: every plan of one
        // session must have been compiled for the same one
        if (!first) first = p;
        for (const char *k : {"pad_left", "pad_right", "pad_top", "pad_bottom"})
            LSSVC_CHECK(p->meta_value(k, 0) == first->meta_value(k, 0), "engine_set_scale: plans '%s' and '%s' were compiled for different %s (%lld vs %lld)",
                        first->kind.c_str(), p->kind.c_str(), k, (long long)first->meta_value(k, 0), (long long)p->meta_value(k, 0));
    }
    // encoder and decoder of one model must run the same kernels (the fp16 range audit may have moved layers to the exact
    // fp32 kernel while the plans were compiled; a mismatch would give streams the other side cannot decode)
    Plan *const groups[2][10] = {{e->i_enc.get(), e->i_dec.get(), e->intra.get(), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr},
                                 {e->p1_enc.get(), e->p1_dec.get(), e->p_enc.get(), e->p_dec.get(), e->first_p.get(), e->steady_p.get(), e->bl_plan[0].get(),
                                  e->bl_plan[1].get(), e->el_plan[0].get(), e->el_plan[1].get()}};
    for (auto &g : groups) {
        const Plan *ref = nullptr;
        for (Plan *p : g) {
            if (!p) continue;
            if (!ref) ref = p;
            LSSVC_CHECK(p->meta_value("f32_layers_crc", 0) == ref->meta_value("f32_layers_crc", 0) &&
                            p->meta_value("f32_layers_n", 0) == ref->meta_value("f32_layers_n", 0),
                        "engine_set_scale: plans '%s' and '%s' were compiled with different sets of fp32-fallback layers (range audit): "
                        "recompile them from one model", ref->kind.c_str(), p->kind.c_str());
        }
    }
    return 0;
}

extern "C" int lssvc_engine_plan_meta(void *h, int32_t which, const char *name, int64_t *out) {
    LSSVC_CHECK(h && name && out, "engine_plan_meta: bad arguments");
    Engine *e = static_cast<Engine *>(h);
    Plan *const all[9] = {e->intra.get(), e->first_p.get(), e->steady_p.get(), e->i_enc.get(), e->i_dec.get(), e->p1_enc.get(),
                          e->p1_dec.get(), e->p_enc.get(), e->p_dec.get()};
    LSSVC_CHECK(which >= 0 && which < 9 && all[which], "engine_plan_meta: plan %d is not loaded", which);
    for (auto &m : all[which]->meta)
        if (m.first == name) {
            *out = m.second;
            return 0;
        }
    return fail("engine_plan_meta: plan %d has no entry '%s'", which, name);
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

// Round 6: a P-frame with the NEXT frame's base layer coded beside this frame's enhancement layer (what LSSVC_extend.forward_one_frame does
// with next_x_bl / frame_id; lssvc_amd/inter.py). The base layer of a P-frame reads the previous frame's BASE layer only, so BL(t+1) runs
// on the engine's second stream while EL(t) runs on the caller's; its results wait in the base-layer plan's output regions for the next
// call, which then codes its enhancement layer only. Contract (the Python protocol's): the caller codes consecutive frames of one
// sequence, passes as x_bl the tensor it named as next_x_bl in the previous call, and clamps the DPB's reference frames to [0, 1]
// between frames (test.py:249-250; the base-layer plan clamps its own copy of the reference, so the look-ahead sees the same values).
// lssvc_engine_lookahead_reset() (or a call with ref_feature_bl == NULL: the first P-frame after an I-frame) drops a base layer coded
// ahead. Same launches per layer as lssvc_engine_pframe: bit-identical results.
extern "C" int lssvc_engine_pframe_lookahead(void *h, const float *x_bl, const float *x_el, const float *next_x_bl, const float *ref_frame_bl,
                                             const float *ref_frame_el, const float *ref_feature_bl, const float *ref_feature_el, double bits[2],
                                             float *recon_bl, float *feature_bl, float *recon_el, float *feature_el, float *mv_hat, float *warp_frame,
                                             void *stream) {
    LSSVC_CHECK(h && bits && x_el && ref_frame_el && ref_feature_el, "engine_pframe_lookahead: bad arguments");
    Engine *e = static_cast<Engine *>(h);
    const int steady = ref_feature_bl ? 1 : 0;
    LSSVC_CHECK(e->bl_plan[0] && e->bl_plan[1] && e->el_plan[0] && e->el_plan[1], "engine_pframe_lookahead: load the four layer plans first (lssvc_engine_load_inter_layers)");
    hipStream_t main = e->stream(stream), side = e->ahead_stream;
    if (!steady) e->ahead = nullptr;                         // a new GOP: nothing coded ahead belongs to it
    double s[16];
    Plan *src = e->ahead;                                    // the plan whose outputs hold BL(t)
    if (src) {
        LSSVC_HIP(hipStreamWaitEvent(main, e->ev_ahead, 0)); // BL(t) was finished on the second stream during the previous call
        for (int i = 0; i < 4; ++i) s[i] = e->ahead_bits[i];
    } else {
        LSSVC_CHECK(x_bl && ref_frame_bl, "engine_pframe_lookahead: no base layer was coded ahead for this frame: x_bl and ref_frame_bl are required");
        src = e->bl_plan[steady].get();
        if (int rc = run_plan(*src, {{"x_bl", x_bl}, {"ref_frame_bl", ref_frame_bl}, {"ref_feature_bl", ref_feature_bl}}, {}, s, main)) return rc;
    }
    e->ahead = nullptr;
    // ---- BL(t): to the enhancement-layer plan's inputs and to the caller
    Plan &el = *e->el_plan[steady];
    auto out_ptr = [&](Plan &p, const char *k) -> const void * {
        const int i = p.region_index(k, REGION_OUTPUT);
        return i >= 0 ? p.regions[i].ptr : nullptr;
    };
    auto region_bytes = [&](Plan &p, const char *k) -> size_t {
        const int i = p.region_index(k, REGION_OUTPUT);
        return i >= 0 ? (size_t)p.regions[i].nbytes : 0;
    };
    if (recon_bl) LSSVC_HIP(hipMemcpyAsync(recon_bl, out_ptr(*src, "recon_bl"), region_bytes(*src, "recon_bl"), hipMemcpyDefault, main));
    if (feature_bl) LSSVC_HIP(hipMemcpyAsync(feature_bl, out_ptr(*src, "feature_bl"), region_bytes(*src, "feature_bl"), hipMemcpyDefault, main));
    // ---- BL(t+1) on the second stream, from BL(t)'s reconstruction and feature (its plan clamps the reference itself)
    Plan *nxt = next_x_bl ? e->bl_plan[1].get() : nullptr;
    if (nxt) {
        // the steady base-layer plan reads its references from its INPUT regions; when BL(t) lives in that very plan's outputs (every frame
        // but the first two) the two copies below are what frees the outputs for BL(t+1)
        for (const char *k : {"ref_frame_bl", "ref_feature_bl"}) {
            const int i = nxt->region_index(k, REGION_INPUT);
            LSSVC_CHECK(i >= 0, "engine_pframe_lookahead: the steady base-layer plan has no input %s", k);
            const char *from = k[4] == 'f' && k[5] == 'r' ? "recon_bl" : "feature_bl";
            LSSVC_HIP(hipMemcpyAsync(nxt->regions[i].ptr, out_ptr(*src, from), nxt->regions[i].nbytes, hipMemcpyDefault, main));
        }
    }
    // (the enhancement-layer plan's run copies the four bl_* buffers into its input regions on `main` before anything of BL(t+1) may
    // overwrite them: run_plan's input copies come first, the hand-over event is recorded behind them)
    std::vector<std::pair<const char *, const void *>> el_in = {{"x_el", x_el}, {"ref_frame_el", ref_frame_el}, {"ref_feature_el", ref_feature_el},
                                                                {"bl_recon", out_ptr(*src, "bl_recon")}, {"bl_feature", out_ptr(*src, "bl_feature")},
                                                                {"bl_y_hat", out_ptr(*src, "bl_y_hat")}, {"bl_mv_hat", out_ptr(*src, "bl_mv_hat")}};
    for (auto &kv : el_in) {
        const int i = el.region_index(kv.first, REGION_INPUT);
        LSSVC_CHECK(i >= 0 && kv.second, "engine_pframe_lookahead: enhancement-layer input %s", kv.first);
        LSSVC_HIP(hipMemcpyAsync(el.regions[i].ptr, kv.second, el.regions[i].nbytes, hipMemcpyDefault, main));
    }
    LSSVC_HIP(hipEventRecord(e->ev_handed, main));
    if (nxt) {
        LSSVC_HIP(hipStreamWaitEvent(side, e->ev_handed, 0));
        if (int rc = run_plan(*nxt, {{"x_bl", next_x_bl}}, {}, e->ahead_slots_host, side, true)) return rc;      // (references are in place; queued, not waited for)
        LSSVC_HIP(hipEventRecord(e->ev_ahead, side));
    }
    // ---- EL(t) on the caller's stream (its inputs are in place: none is passed again)
    double t[16];
    if (int rc = run_plan(el, {}, {{"recon_el", recon_el}, {"feature_el", feature_el}, {"mv_hat", mv_hat}, {"warp_frame", warp_frame}}, t, main)) return rc;
    bits[0] = s[0] + s[1] + s[2] + s[3];                 // dmc_net.py:473
    bits[1] = t[4] + t[5] + t[6] + t[7];                 // LSSVC_net.py:508
    if (nxt) {
        LSSVC_HIP(hipEventSynchronize(e->ev_ahead));     // (normally long done: the base layer is a fifth of the frame)
        for (int i = 0; i < 16; ++i) e->ahead_bits[i] = e->ahead_slots_host[i];
        e->ahead = nxt;
    }
    return 0;
}

extern "C" int lssvc_engine_plan_info(void *h, int32_t which, int64_t *out6) {
    LSSVC_CHECK(h && out6, "engine_plan_info: bad arguments");
    Engine *e = static_cast<Engine *>(h);
    Plan *const all[9] = {e->intra.get(), e->first_p.get(), e->steady_p.get(), e->i_enc.get(), e->i_dec.get(), e->p1_enc.get(),
                          e->p1_dec.get(), e->p_enc.get(), e->p_dec.get()};
    LSSVC_CHECK(which >= 0 && which < 9, "engine_plan_info: plan index %d", which);
    Plan *p = all[which];
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

// ---- write_stream = 1 ------------------------------------------------------------------------------------------------
extern "C" int lssvc_engine_load_stream(void *h, const char *iframe_enc, const char *iframe_dec, const char *first_p_enc, const char *first_p_dec,
                                        const char *steady_p_enc, const char *steady_p_dec) {
    LSSVC_CHECK(h, "engine_load_stream: bad arguments");
    Engine *e = static_cast<Engine *>(h);
    LSSVC_HIP(hipSetDevice(e->device));
    struct { std::unique_ptr<Plan> *slot; const char *path, *kind; } todo[6] = {
        {&e->i_enc, iframe_enc, "iframe_enc"}, {&e->i_dec, iframe_dec, "iframe_dec"}, {&e->p1_enc, first_p_enc, "pframe_first_enc"},
        {&e->p1_dec, first_p_dec, "pframe_first_dec"}, {&e->p_enc, steady_p_enc, "pframe_enc"}, {&e->p_dec, steady_p_dec, "pframe_dec"}};
    for (auto &t : todo)
        if (t.path)
            if (int rc = load_into(e, *t.slot, t.path, t.kind, nullptr)) return rc;
    return 0;
}

extern "C" int lssvc_engine_encode_iframe(void *h, const float *x_bl, const float *x_el, uint8_t *bl_file, int64_t bl_cap, int64_t *bl_len,
                                          uint8_t *el_file, int64_t el_cap, int64_t *el_len, float *x_hat_bl, float *x_hat_el,
                                          float *feature_el, void *stream) {
    LSSVC_CHECK(h && bl_len && el_len, "engine_encode_iframe: bad arguments");
    Engine *e = static_cast<Engine *>(h);
    Plan *p = e->i_enc.get();
    LSSVC_CHECK(p != nullptr, "engine_encode_iframe: no I-frame encoder plan loaded");
    if (int rc = run_stream_plan(*p, {{"x_bl", x_bl}, {"x_el", x_el}}, {{"x_hat_bl", x_hat_bl}, {"x_hat_el", x_hat_el}, {"feature_el", feature_el}},
                                 e->stream(stream)))
        return rc;
    LSSVC_CHECK(p->out_strings.size() == 4, "engine_encode_iframe: the plan produced %zu strings, not y and z of two layers", p->out_strings.size());
    if (int rc = write_i_file(bl_file, bl_cap, bl_len, p->meta_value("pic_height_bl", 0), p->meta_value("pic_width_bl", 0), p->out_strings[0], p->out_strings[1]))
        return rc;
    return write_i_file(el_file, el_cap, el_len, p->meta_value("pic_height_el", 0), p->meta_value("pic_width_el", 0), p->out_strings[2], p->out_strings[3]);
}

extern "C" int lssvc_engine_decode_iframe(void *h, const uint8_t *bl_file, int64_t bl_len, const uint8_t *el_file, int64_t el_len, float *x_hat_bl,
                                          float *x_hat_el, float *feature_el, void *stream) {
    LSSVC_CHECK(h, "engine_decode_iframe: bad arguments");
    Engine *e = static_cast<Engine *>(h);
    Plan *p = e->i_dec.get();
    LSSVC_CHECK(p != nullptr, "engine_decode_iframe: no I-frame decoder plan loaded");
    p->in_strings.clear();
    if (int rc = read_i_file(bl_file, bl_len, p->meta_value("pic_height_bl", 0), p->meta_value("pic_width_bl", 0), p->in_strings)) return rc;
    if (int rc = read_i_file(el_file, el_len, p->meta_value("pic_height_el", 0), p->meta_value("pic_width_el", 0), p->in_strings)) return rc;
    return run_stream_plan(*p, {}, {{"x_hat_bl", x_hat_bl}, {"x_hat_el", x_hat_el}, {"feature_el", feature_el}}, e->stream(stream));
}

extern "C" int lssvc_engine_encode_pframe(void *h, const float *x_bl, const float *x_el, const float *ref_frame_bl, const float *ref_frame_el,
                                          const float *ref_feature_bl, const float *ref_feature_el, uint8_t *bl_file, int64_t bl_cap,
                                          int64_t *bl_len, uint8_t *el_file, int64_t el_cap, int64_t *el_len, float *recon_bl, float *feature_bl,
                                          float *recon_el, float *feature_el, void *stream) {
    LSSVC_CHECK(h && bl_len && el_len, "engine_encode_pframe: bad arguments");
    Engine *e = static_cast<Engine *>(h);
    Plan *p = ref_feature_bl ? e->p_enc.get() : e->p1_enc.get();
    LSSVC_CHECK(p != nullptr, "engine_encode_pframe: no %s encoder plan loaded", ref_feature_bl ? "steady-P" : "first-P");
    if (int rc = run_stream_plan(*p, {{"x_bl", x_bl}, {"x_el", x_el}, {"ref_frame_bl", ref_frame_bl}, {"ref_frame_el", ref_frame_el},
                                      {"ref_feature_bl", ref_feature_bl}, {"ref_feature_el", ref_feature_el}},
                                 {{"recon_bl", recon_bl}, {"feature_bl", feature_bl}, {"recon_el", recon_el}, {"feature_el", feature_el}},
                                 e->stream(stream)))
        return rc;
    LSSVC_CHECK(p->out_strings.size() == 2, "engine_encode_pframe: the plan produced %zu strings, not one per layer", p->out_strings.size());
    if (int rc = write_p_file(bl_file, bl_cap, bl_len, p->out_strings[0])) return rc;
    return write_p_file(el_file, el_cap, el_len, p->out_strings[1]);
}

extern "C" int lssvc_engine_decode_pframe(void *h, const uint8_t *bl_file, int64_t bl_len, const uint8_t *el_file, int64_t el_len,
                                          const float *ref_frame_bl, const float *ref_frame_el, const float *ref_feature_bl,
                                          const float *ref_feature_el, float *recon_bl, float *feature_bl, float *recon_el, float *feature_el,
                                          void *stream) {
    LSSVC_CHECK(h, "engine_decode_pframe: bad arguments");
    Engine *e = static_cast<Engine *>(h);
    Plan *p = ref_feature_bl ? e->p_dec.get() : e->p1_dec.get();
    LSSVC_CHECK(p != nullptr, "engine_decode_pframe: no %s decoder plan loaded", ref_feature_bl ? "steady-P" : "first-P");
    p->in_strings.clear();
    if (int rc = read_p_file(bl_file, bl_len, p->in_strings)) return rc;
    if (int rc = read_p_file(el_file, el_len, p->in_strings)) return rc;
    return run_stream_plan(*p, {{"ref_frame_bl", ref_frame_bl}, {"ref_frame_el", ref_frame_el}, {"ref_feature_bl", ref_feature_bl},
                                {"ref_feature_el", ref_feature_el}},
                           {{"recon_bl", recon_bl}, {"feature_bl", feature_bl}, {"recon_el", recon_el}, {"feature_el", feature_el}}, e->stream(stream));
}
