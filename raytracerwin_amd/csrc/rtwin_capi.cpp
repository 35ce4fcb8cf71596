// rtwin_capi.cpp -- the C ABI declared in include/rtwin.h, over the HIP runtime.
// There is deliberately no CPU fallback: every device entry point needs a HIP device.
#include "../../include/rtwin.h"

#include <hip/hip_runtime_api.h>

#include <chrono>
#include <climits>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "rtw_device.h"
#include "rtw_host.h"

static_assert(sizeof(rtw_material_node) == sizeof(RtwMaterialNode), "material node layout");
static_assert(RTW_MAX_BOUNCE == RTW_MAX_BOUNCE_DEV, "bounce limit");
static_assert(RTW_UNIT_TABLE_SIZE == RTW_TABLE_SIZE, "table size");

#ifdef RTW_TIMING
namespace rtw { int read_timing(unsigned long long* out, int n); }
#endif

namespace {

thread_local std::string t_error;
int fail(int code, const std::string& msg) { t_error = msg; return code; }
int hip_fail(hipError_t e, const char* what)
{
    return fail(RTW_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return hip_fail(e_, #expr); } while (0)

// host copy of the unit-vector table, generated once per process
std::mutex g_table_mutex;
std::vector<float> g_unit_table;
const std::vector<float>& host_unit_table()
{
    std::lock_guard<std::mutex> l(g_table_mutex);
    if (g_unit_table.empty()) {
        g_unit_table.resize((size_t)RTW_TABLE_SIZE * 3);
        int th = (int)std::thread::hardware_concurrency();
        rtw::fill_unit_table(g_unit_table.data(), th > 0 ? th : 1);
    }
    return g_unit_table;
}

// device copy of the table: ONE per (process, device), shared by the contexts on that device (201 MB: a second context costs nothing)
struct DeviceTable { float* d = nullptr; int users = 0; };
std::map<int, DeviceTable> g_device_tables;
hipError_t acquire_device_table(int device, float** out)
{
    const std::vector<float>& tab = host_unit_table();
    std::lock_guard<std::mutex> l(g_table_mutex);
    DeviceTable& t = g_device_tables[device];
    if (!t.d) {
        hipError_t e = hipMalloc((void**)&t.d, tab.size() * sizeof(float));
        if (e != hipSuccess) { t.d = nullptr; return e; }
        e = hipMemcpy(t.d, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(t.d); t.d = nullptr; return e; }
    }
    t.users++;
    *out = t.d;
    return hipSuccess;
}
void release_device_table(int device)
{
    std::lock_guard<std::mutex> l(g_table_mutex);
    DeviceTable& t = g_device_tables[device];
    if (t.users > 0 && --t.users == 0 && t.d) { (void)hipFree(t.d); t.d = nullptr; }
}

}  // namespace

#define RTW_MAX_PARTS 4
struct rtw_context {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipStream_t aux_stream = nullptr;   // sky-only tiles of the bins + wave pipeline run here, beside the main stream
    hipEvent_t fork_event = nullptr, join_event = nullptr;
    int batch_pos = 0;                  // rtw_render_passes: 0 = a pass on its own, 1 = first of a run, 2 = inside a run, 3 = last of a run
    bool aux_unjoined = false;          // a sky kernel of the current run is not joined yet
    float* d_unit = nullptr;
    float* d_gamma = nullptr;
    float* d_lut = nullptr;
    unsigned long long* d_stats = nullptr;
    bool stats_enabled = false;
    void* d_workspace = nullptr;        // per-launch queues / level store of the bounce recursion (grown on demand)
    size_t workspace_bytes = 0;
    int pipeline = 4;                   // 4 = pass-batched: screen bins + a ray per lane, K passes per set of launches (default), 3 = screen bins + a wave per secondary ray, one pass
                                        // per set of launches (the one-pass reference), 0 = one kernel, one thread per pixel (reference-order counters)
    uint32_t* h_counters = nullptr;     // pinned: queue / pending lengths copied back after each pass
    hipEvent_t counters_event = nullptr;// recorded after that copy; the value is only read once the event has completed
    bool counters_pending = false;
    long long counters_shape = -1;      // launch shape (work items, samples) the copy in flight belongs to
    long long known_shape = -1;         // launch shape of known_paths
    int known_paths = -1;               // queue length of the latest pass whose copy has completed
    int known_rounds[32];               // wavefront: trace-list lengths of that pass
    int cu_count = 256;
    const void* clean_ws = nullptr;     // workspace and counters offset whose counters the last pass left zeroed (bins + wave pipeline)
    size_t clean_off = 0;
    int hint_period = 16;               // the queue lengths are read back every hint_period-th pass (a 256-byte copy costs the stream ~10 us; the lengths drift slowly)
    int hint_tick = 0;
    int kernel_timing = 0;              // 1: record events around the three kernels of each pass
    hipEvent_t timing_events[4] = { nullptr, nullptr, nullptr, nullptr };
    // pass-batched pipeline (pipeline 4): its own workspace (the list counters sit at its start and are left zeroed by every group)
    // a group runs as up to RTW_MAX_PARTS parts on as many streams (part 0 on the context's stream), each with its own workspace; the events order them
    void* d_group_ws[RTW_MAX_PARTS] = { nullptr, nullptr, nullptr, nullptr };
    size_t group_ws_bytes[RTW_MAX_PARTS] = { 0, 0, 0, 0 };
    bool group_clean[RTW_MAX_PARTS] = { false, false, false, false };
    hipStream_t part_stream[RTW_MAX_PARTS] = { nullptr, nullptr, nullptr, nullptr };        // [0] unused: part 0 runs on `stream`
    hipEvent_t split_fork = nullptr, part_resolved[RTW_MAX_PARTS] = { nullptr, nullptr, nullptr, nullptr }, part_done[RTW_MAX_PARTS] = { nullptr, nullptr, nullptr, nullptr };
    int backface_filter = 1;                  // option: the persistent trace kernel stages the triangles' planes and never notes a leaf that faces away
    int group_parts = 2;                      // option: parts a split group runs as, at most (measured on C2 / C4 at 20 passes: 2 parts -8 % / -1 % against one, 3 and 4 parts +10..25 %)
    int group_split = 1, split_min = 8;       // options: halves when a group has at least split_min passes ...
    int split_paths = 400000;                 // ... and each half at least this many paths (a rank's share of a small frame at 8 ranks stays whole: measured 0.0123 whole, 0.0144 ms split)
    bool lane_sky_only = false;
    int sky_blocks = 4;                 // the sky kernel's blocks per CU at most (its lanes loop over the tiles); 1, 2, 4, 16 measured: all within 1.5 % on C2 / C4, at 1 and 20 passes per call
    int lane = 0, lane_count = 1, lane_sky_passes = 0, lane_sky_first = 0;        // set by rtw_render_passes around render_group: which part of how many; the whole group's passes (the parts share out its sky tiles)
    uint32_t* h_gcounters = nullptr;    // pinned: list lengths of a finished group
    hipEvent_t gcounters_event = nullptr;
    bool gcounters_pending = false;
    struct GroupKey { const void* scene = nullptr; long long capacity = -1; int max_bounce = -1, preview = -1, n_passes = -1;
                      bool operator==(const GroupKey& o) const { return scene == o.scene && capacity == o.capacity && max_bounce == o.max_bounce && preview == o.preview && n_passes == o.n_passes; } };
    GroupKey gcounters_key, known_gkey;
    int gcounters_passes = 1;           // passes of the group whose lengths are in flight
    int last_group_passes = 0;          // passes of the latest group (rtw_last_group_passes)
    int known_ground[32];
    int known_goverflow[24];
    int known_gtrace[16];
    int budget_nodes = 0;               // ... trees with more nodes than this get the budget
    int visit_budget = 384;             // one-mesh scenes: node visits a ray gets in the ray-per-lane kernel before it goes to the wave-per-ray one (0: no limit)
    int group_paths = 32 << 20;         // passes are grouped until a launch holds about this many paths (measured against 16 Mi: SetupScene -5 %, C5 at 20 passes -7 %, the others unchanged; 64 and 128 Mi: no further change) ...
    int primary_passes = 0;             // passes of a tile one wave of the primary kernel walks the tile's bin for at once (0: chosen per launch; -1: one ray set at a time, the round-2 order)
    int group_max = 256;                // ... and at most this many passes (a power of two)
    int wave_below = 80000;             // a trace round with fewer rays (x 5 for trees of more than 4096 nodes) runs a wave per ray (measured with groups as two halves: C2 -3 % against 160 000; big trees keep 400 000)
    int device_build = 1;               // rtw_scene_commit builds the tree (KdNode::Build's decisions) and the layouts derived from it on the device (0: on the host)
    int last_pipeline = -1;             // the pipeline the latest render call actually ran (rtw_last_pass_pipeline)
    size_t workspace_limit = (size_t)24 << 30;  // a group's workspace may not exceed this (option "workspace_limit_mb"); hipMalloc failing counts as exceeding it
    bool ws_refused = false;            // the latest ensure_group_workspace was refused (limit or out of memory): the caller retries with a smaller group
    bool ws_single = false;             // the group at hand is one pass: the limit does not apply (only hipMalloc can refuse it)
    int group_cap = 0;                  // > 0: groups of the current rtw_render_passes call hold at most this many passes (set after a refusal)
    int fallbacks = 0;                  // how many times a group was re-formed smaller after a refusal (rtw_context_memory_bytes' caller can see it: option-free diagnostics)
};

struct rtw_scene {
    rtw_context* ctx = nullptr;
    std::vector<std::unique_ptr<rtw::HostMesh>> meshes;
    bool committed = false;
    bool has_emissive = false;          // some material tree holds an Emissive node
    bool materials_finite = true;       // every material colour and parameter is a finite number
    bool has_analytic = false;          // some shape is a sphere / plane / capsule
    bool texture_carry = false;         // ... and comes after a textured mesh: its hits can inherit that mesh's sampled colour
                                        // (one RayHitResult serves all shapes, Src/RayTracerScene.cpp:99-125) -> single-kernel pipeline
    int prune = 1;
    int traversal = 1;
    RtwSceneDev* d_scene = nullptr;
    std::vector<void*> allocs;
    std::vector<const RtwNode*> d_nodes_of;     // per shape: the device-built tree and leaf records (null: built on the host)
    std::vector<const RtwTri*> d_tris_of;
    // screen-space bins of the reference camera, one set per (width, height, bin shape) this scene has been rendered at
    struct JobTable { int task_rows, rank, world, spp; const uint32_t* d_order; int n_jobs, n_busy; };     // n_busy: jobs of tiles with a non-empty bin (they come first)
    // pass-batched pipeline: the launch's tiles split into busy ones (heaviest bins first) and sky-only ones, and the primary kernel's jobs
    struct GroupTable { int task_rows, rank, world, spp; const uint32_t* d_busy; const uint32_t* d_sky; const uint32_t* d_jobs; int n_busy, n_sky, n_jobs; };
    struct BinSet { int width, height, bin_w, bin_h; RtwBinsDev* d_bins; const float* d_dx; const float* d_dy;
                    std::vector<uint32_t> weight;           // per bin: leaves listed, over all shapes (1000+ for a shape without bins)
                    std::vector<JobTable> jobs;             // job tables per (task partition, sub-sample count) rendered so far
                    std::vector<GroupTable> gtables; };
    std::vector<BinSet> bin_sets;
};

struct rtw_framebuffer {
    rtw_context* ctx = nullptr;
    int width = 0, height = 0;
    void* accum = nullptr;
    void* argb = nullptr;
    bool owned = false;
};

namespace {
template <typename T>
int upload(rtw_scene* scene, const std::vector<T>& v, const T** out)
{
    *out = nullptr;
    if (v.empty()) return RTW_OK;
    void* d = nullptr;
    HIP_TRY(hipMalloc(&d, v.size() * sizeof(T)));
    scene->allocs.push_back(d);
    HIP_TRY(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = (const T*)d;
    return RTW_OK;
}

// make sure the context's level workspace holds `bytes`; growing waits for the stream first
int ensure_workspace(rtw_context* ctx, size_t bytes)
{
    if (bytes <= ctx->workspace_bytes) return RTW_OK;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->d_workspace) { (void)hipFree(ctx->d_workspace); ctx->d_workspace = nullptr; ctx->workspace_bytes = 0; }
    ctx->clean_ws = nullptr;
    HIP_TRY(hipMalloc(&ctx->d_workspace, bytes));
    ctx->workspace_bytes = bytes;
    return RTW_OK;
}
}  // namespace

extern "C" {

const char* rtw_last_error(void) { return t_error.c_str(); }
#ifdef HIPEMU_COMPUTE_UNITS      /* defined by tests/cpu_emul's stand-in for the HIP headers: this is the host build the sanitizers run, not the product */
const char* rtw_version(void) { return "rtwin 0.3 (HOST EMULATION for sanitizer runs -- not the product)"; }
#else
#ifndef RTW_KERNELS_SHA
#define RTW_KERNELS_SHA "unknown"
#endif
const char* rtw_version(void) { return "rtwin 0.3 (gfx950; kernels " RTW_KERNELS_SHA ")"; }      // the hash of the device sources + flags (csrc/Makefile)
#endif

int rtw_context_create(int device_index, rtw_context** out)
{
    if (!out) return fail(RTW_ERR_INVALID, "out is null");
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(RTW_ERR_NO_DEVICE, "no HIP device: librtwin has no CPU fallback");
    if (device_index < 0 || device_index >= count) return fail(RTW_ERR_INVALID, "device index out of range");
    HIP_TRY(hipSetDevice(device_index));
    std::unique_ptr<rtw_context> c(new rtw_context());
    c->device = device_index;
    { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_index) == hipSuccess && cus > 0) c->cu_count = cus; }
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->own_stream = true;
    HIP_TRY(acquire_device_table(device_index, &c->d_unit));
    float thr[256], lut[256];
    rtw::gamma_thresholds(thr);
    rtw::texel_lut(lut);
    HIP_TRY(hipMalloc((void**)&c->d_gamma, sizeof thr));
    HIP_TRY(hipMalloc((void**)&c->d_lut, sizeof lut));
    HIP_TRY(hipMemcpy(c->d_gamma, thr, sizeof thr, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->d_lut, lut, sizeof lut, hipMemcpyHostToDevice));
    HIP_TRY(hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&c->split_fork, hipEventDisableTiming));
    for (int j = 0; j < RTW_MAX_PARTS; j++) {
        if (j > 0) HIP_TRY(hipStreamCreateWithFlags(&c->part_stream[j], hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&c->part_resolved[j], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&c->part_done[j], hipEventDisableTiming));
    }
    HIP_TRY(hipEventCreateWithFlags(&c->fork_event, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->join_event, hipEventDisableTiming));
    HIP_TRY(hipHostMalloc((void**)&c->h_counters, 256, hipHostMallocDefault));
    HIP_TRY(hipEventCreateWithFlags(&c->counters_event, hipEventDisableTiming));
    HIP_TRY(hipHostMalloc((void**)&c->h_gcounters, 256, hipHostMallocDefault));
    HIP_TRY(hipEventCreateWithFlags(&c->gcounters_event, hipEventDisableTiming));
    for (int i = 0; i < 64; i++) c->h_gcounters[i] = 0;
    for (int r = 0; r < 32; r++) c->known_ground[r] = -1;
    for (int r = 0; r < 24; r++) c->known_goverflow[r] = -1;
    for (int r = 0; r < 16; r++) c->known_gtrace[r] = -1;
    for (int i = 0; i < 64; i++) c->h_counters[i] = 0;
    for (int r = 0; r < 32; r++) c->known_rounds[r] = -1;
    HIP_TRY(hipMalloc((void**)&c->d_stats, 8 * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(c->d_stats, 0, 8 * sizeof(unsigned long long)));
    *out = c.release();
    return RTW_OK;
}

int rtw_context_destroy(rtw_context* ctx)
{
    if (!ctx) return RTW_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->d_unit) release_device_table(ctx->device);
    (void)hipFree(ctx->d_workspace); (void)hipFree(ctx->d_gamma); (void)hipFree(ctx->d_lut); (void)hipFree(ctx->d_stats);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->aux_stream) { (void)hipStreamSynchronize(ctx->aux_stream); (void)hipStreamDestroy(ctx->aux_stream); }
    if (ctx->fork_event) (void)hipEventDestroy(ctx->fork_event);
    if (ctx->join_event) (void)hipEventDestroy(ctx->join_event);
    for (int j = 0; j < RTW_MAX_PARTS; j++) {
        if (ctx->part_stream[j]) { (void)hipStreamSynchronize(ctx->part_stream[j]); (void)hipStreamDestroy(ctx->part_stream[j]); }
        if (ctx->d_group_ws[j]) (void)hipFree(ctx->d_group_ws[j]);
        if (ctx->part_resolved[j]) (void)hipEventDestroy(ctx->part_resolved[j]);
        if (ctx->part_done[j]) (void)hipEventDestroy(ctx->part_done[j]);
    }
    if (ctx->split_fork) (void)hipEventDestroy(ctx->split_fork);
    if (ctx->h_gcounters) (void)hipHostFree(ctx->h_gcounters);
    if (ctx->gcounters_event) (void)hipEventDestroy(ctx->gcounters_event);
    if (ctx->h_counters) (void)hipHostFree(ctx->h_counters);
    if (ctx->counters_event) (void)hipEventDestroy(ctx->counters_event);
    for (int i = 0; i < 4; i++) if (ctx->timing_events[i]) (void)hipEventDestroy(ctx->timing_events[i]);
    delete ctx;
    return RTW_OK;
}

int rtw_context_set_stream(rtw_context* ctx, void* hip_stream)
{
    if (!ctx) return fail(RTW_ERR_INVALID, "context is null");
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->own_stream) { (void)hipStreamDestroy(ctx->stream); ctx->own_stream = false; }
    ctx->stream = (hipStream_t)hip_stream;
    return RTW_OK;
}

int rtw_context_set_option(rtw_context* ctx, const char* name, int value)
{
    if (!ctx || !name) return fail(RTW_ERR_INVALID, "null argument");
    if (std::strcmp(name, "pipeline") == 0) {
        if (value != 0 && value != 3 && value != 4) return fail(RTW_ERR_INVALID, "pipeline must be 4 (pass-batched, default), 3 (one pass per set of launches) or 0 (one kernel)");
        ctx->pipeline = value;
        return RTW_OK;
    }
    if (std::strcmp(name, "visit_budget") == 0) { ctx->visit_budget = value < 0 ? 0 : value; return RTW_OK; }
    if (std::strcmp(name, "workspace_limit_mb") == 0) { ctx->workspace_limit = value <= 0 ? ((size_t)24 << 30) : ((size_t)value << 20); return RTW_OK; }
    if (std::strcmp(name, "backface_filter") == 0) { ctx->backface_filter = value ? 1 : 0; return RTW_OK; }
    if (std::strcmp(name, "group_parts") == 0) { ctx->group_parts = value < 1 ? 1 : (value > RTW_MAX_PARTS ? RTW_MAX_PARTS : value); return RTW_OK; }
    if (std::strcmp(name, "group_split") == 0) { ctx->group_split = value != 0; return RTW_OK; }
    if (std::strcmp(name, "split_paths") == 0) { ctx->split_paths = value < 0 ? 0 : value; return RTW_OK; }
    if (std::strcmp(name, "split_min") == 0) { ctx->split_min = value < 2 ? 2 : value; return RTW_OK; }
    if (std::strcmp(name, "budget_nodes") == 0) { ctx->budget_nodes = value < 0 ? 0 : value; return RTW_OK; }
    if (std::strcmp(name, "wave_below") == 0) { ctx->wave_below = value < 0 ? 0 : value; return RTW_OK; }
    if (std::strcmp(name, "device_build") == 0) { ctx->device_build = value ? 1 : 0; return RTW_OK; }
    if (std::strcmp(name, "sky_blocks") == 0) { ctx->sky_blocks = value < 1 ? 1 : value; return RTW_OK; }
    if (std::strcmp(name, "primary_passes") == 0) { ctx->primary_passes = value < -1 ? -1 : (value > 4 ? 4 : value); return RTW_OK; }
    if (std::strcmp(name, "group_paths") == 0) { ctx->group_paths = value < 1 ? 1 : value; return RTW_OK; }
    if (std::strcmp(name, "group_max") == 0) {
        if (value < 1 || value > 256 || (value & (value - 1)) != 0) return fail(RTW_ERR_INVALID, "group_max must be a power of two in 1..256");
        ctx->group_max = value;
        return RTW_OK;
    }
    if (std::strcmp(name, "hint_period") == 0) { ctx->hint_period = value < 1 ? 1 : value; return RTW_OK; }
    if (std::strcmp(name, "kernel_timing") == 0) {
        if (value && !ctx->timing_events[0]) for (int i = 0; i < 4; i++) HIP_TRY(hipEventCreate(&ctx->timing_events[i]));
        ctx->kernel_timing = value ? 1 : 0;
        return RTW_OK;
    }
    return fail(RTW_ERR_INVALID, std::string("unknown option ") + name);
}

int rtw_last_pass_kernel_ms(rtw_context* ctx, float out3[3])
{
    if (!ctx || !out3) return fail(RTW_ERR_INVALID, "null argument");
    if (!ctx->kernel_timing || !ctx->timing_events[0]) return fail(RTW_ERR_STATE, "kernel_timing is off");
    HIP_TRY(hipEventSynchronize(ctx->timing_events[3]));
    for (int i = 0; i < 3; i++) HIP_TRY(hipEventElapsedTime(&out3[i], ctx->timing_events[i], ctx->timing_events[i + 1]));
    return RTW_OK;
}

int rtw_last_group_passes(rtw_context* ctx)
{
    if (!ctx) return fail(RTW_ERR_INVALID, "context is null");
    return ctx->last_group_passes;
}

int rtw_last_pass_pipeline(rtw_context* ctx)
{
    if (!ctx) return fail(RTW_ERR_INVALID, "context is null");
    return ctx->last_pipeline;
}

int rtw_context_synchronize(rtw_context* ctx)
{
    if (!ctx) return fail(RTW_ERR_INVALID, "context is null");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RTW_OK;
}

// ---- scene ------------------------------------------------------------------------------------------
int rtw_scene_create(rtw_context* ctx, rtw_scene** out)
{
    // ctx may be NULL: a host-only scene (parse + build + inspect), every device query then fails
    if (!out) return fail(RTW_ERR_INVALID, "null argument");
    *out = new rtw_scene();
    (*out)->ctx = ctx;
    return RTW_OK;
}

int rtw_scene_destroy(rtw_scene* scene)
{
    if (!scene) return RTW_OK;
    if (scene->ctx) {
        (void)hipSetDevice(scene->ctx->device);
        (void)hipStreamSynchronize(scene->ctx->stream);
    }
    for (void* p : scene->allocs) (void)hipFree(p);
    delete scene;
    return RTW_OK;
}

static int default_material(rtw::HostMesh& m)
{
    m.material.clear();     // a shape added without a material has none (RayTrace then returns black)
    return 0;
}

int rtw_scene_add_mesh_obj(rtw_scene* scene, const char* obj_path, int* out_shape)
{
    if (!scene || !obj_path) return fail(RTW_ERR_INVALID, "null argument");
    if (scene->committed) return fail(RTW_ERR_STATE, "scene already committed");
    if ((int)scene->meshes.size() >= RTW_DEV_MAX_SHAPES) return fail(RTW_ERR_LIMIT, "too many shapes");
    std::unique_ptr<rtw::HostMesh> m(new rtw::HostMesh());
    const std::string err = rtw::load_obj(obj_path, *m);
    if (!err.empty()) return fail(RTW_ERR_IO, err);
    if ((int)m->material_names.size() > RTW_MAX_MESH_MATERIALS) return fail(RTW_ERR_LIMIT, "too many materials in mesh");
    default_material(*m);
    scene->meshes.push_back(std::move(m));
    if (out_shape) *out_shape = (int)scene->meshes.size() - 1;
    return RTW_OK;
}

int rtw_scene_add_mesh(rtw_scene* scene, const float* positions, int n_positions, const float* texcoords, int n_texcoords,
                       const float* normals, int n_normals, const int32_t* idx_p, const int32_t* idx_t, const int32_t* idx_n,
                       const int32_t* tri_material, int n_tris, const float* shape_bounds6, int* out_shape)
{
    if (!scene || !positions || !texcoords || !normals || !idx_p || !idx_t || !idx_n || n_tris < 0)
        return fail(RTW_ERR_INVALID, "null argument");
    if (scene->committed) return fail(RTW_ERR_STATE, "scene already committed");
    if ((int)scene->meshes.size() >= RTW_DEV_MAX_SHAPES) return fail(RTW_ERR_LIMIT, "too many shapes");
    std::unique_ptr<rtw::HostMesh> m(new rtw::HostMesh());
    auto copy3 = [](std::vector<rtw::Vec3>& dst, const float* src, int n) {
        dst.resize((size_t)n);
        if (n > 0) std::memcpy(dst.data(), src, (size_t)n * 12);
    };
    copy3(m->points, positions, n_positions);
    copy3(m->texcoords, texcoords, n_texcoords);
    copy3(m->normals, normals, n_normals);
    m->point_idx.assign(idx_p, idx_p + (size_t)n_tris * 3);
    m->texcoord_idx.assign(idx_t, idx_t + (size_t)n_tris * 3);
    m->normal_idx.assign(idx_n, idx_n + (size_t)n_tris * 3);
    if (tri_material) m->poly_material.assign(tri_material, tri_material + n_tris);
    else m->poly_material.assign((size_t)n_tris, -1);
    int max_mat = -1;
    for (int32_t v : m->poly_material) if (v > max_mat) max_mat = v;
    if (max_mat >= RTW_MAX_MESH_MATERIALS) return fail(RTW_ERR_LIMIT, "material id too large");
    m->material_names.resize((size_t)(max_mat + 1));
    m->texture_paths.resize((size_t)(max_mat + 1));
    m->textures.resize((size_t)(max_mat + 1));
    const std::string err = rtw::finish_arrays(*m, shape_bounds6);
    if (!err.empty()) return fail(RTW_ERR_INVALID, err);
    scene->meshes.push_back(std::move(m));
    if (out_shape) *out_shape = (int)scene->meshes.size() - 1;
    return RTW_OK;
}

// RSphere / RPlane / RCapsule (Src/Shapes.h:46-112): a shape record without arrays.  The culling box is RAabb::ExpandBySphere
// (Src/RAabb.h:46-54) from the default box (Src/RAabb.cpp:13-17), in the reference's float operations.
static void expand_by_sphere(rtw::HostMesh& m, const float c[3], float r)
{
    for (int k = 0; k < 3; k++) {
        if (c[k] - r < m.bmin[k]) m.bmin[k] = c[k] - r;
        if (c[k] + r > m.bmax[k]) m.bmax[k] = c[k] + r;
    }
}
static int add_analytic(rtw_scene* scene, int kind, const float* a, const float* b, float radius, int* out_shape)
{
    if (!scene || !a || (kind != RTW_SHAPE_SPHERE && !b)) return fail(RTW_ERR_INVALID, "null argument");
    if (scene->committed) return fail(RTW_ERR_STATE, "scene already committed");
    if ((int)scene->meshes.size() >= RTW_DEV_MAX_SHAPES) return fail(RTW_ERR_LIMIT, "too many shapes");
    std::unique_ptr<rtw::HostMesh> m(new rtw::HostMesh());
    m->kind = kind; m->radius = radius;
    for (int k = 0; k < 3; k++) { m->pa[k] = a[k]; m->pb[k] = b ? b[k] : 0.0f; m->bmin[k] = FLT_MAX; m->bmax[k] = -FLT_MAX; }
    if (kind == RTW_SHAPE_SPHERE) expand_by_sphere(*m, m->pa, radius);
    if (kind == RTW_SHAPE_CAPSULE) { expand_by_sphere(*m, m->pa, radius); expand_by_sphere(*m, m->pb, radius); }
    scene->meshes.push_back(std::move(m));
    if (out_shape) *out_shape = (int)scene->meshes.size() - 1;
    return RTW_OK;
}
int rtw_scene_add_sphere(rtw_scene* scene, const float center[3], float radius, int* out_shape)
{
    return add_analytic(scene, RTW_SHAPE_SPHERE, center, nullptr, radius, out_shape);
}
int rtw_scene_add_plane(rtw_scene* scene, const float normal[3], const float point[3], int* out_shape)
{
    return add_analytic(scene, RTW_SHAPE_PLANE, normal, point, 0.0f, out_shape);
}
int rtw_scene_add_capsule(rtw_scene* scene, const float start[3], const float end[3], float radius, int* out_shape)
{
    return add_analytic(scene, RTW_SHAPE_CAPSULE, start, end, radius, out_shape);
}
// RTriangle::Create(p0, p1, p2) (Src/Shapes.h:106-130): the culling box is the three points' (RAabb::Expand)
int rtw_scene_add_triangle(rtw_scene* scene, const float p0[3], const float p1[3], const float p2[3], int* out_shape)
{
    if (!p2) return fail(RTW_ERR_INVALID, "null argument");
    int idx = -1;
    const int rc = add_analytic(scene, RTW_SHAPE_TRIANGLE, p0, p1, 0.0f, &idx);
    if (rc != RTW_OK) return rc;
    rtw::HostMesh& m = *scene->meshes[(size_t)idx];
    for (int k = 0; k < 3; k++) m.pc[k] = p2[k];
    const float* pts[3] = { m.pa, m.pb, m.pc };
    for (int i = 0; i < 3; i++)
        for (int k = 0; k < 3; k++) {
            if (pts[i][k] < m.bmin[k]) m.bmin[k] = pts[i][k];
            if (pts[i][k] > m.bmax[k]) m.bmax[k] = pts[i][k];
        }
    rtw::triangle_plane(m.pa, m.pb, m.pc, m.pn, &m.pd1);
    if (out_shape) *out_shape = idx;
    return RTW_OK;
}

int rtw_scene_set_texture(rtw_scene* scene, int shape, int material_id, const uint8_t* texels, int width, int height, int channels)
{
    if (!scene || !texels) return fail(RTW_ERR_INVALID, "null argument");
    if (scene->committed) return fail(RTW_ERR_STATE, "scene already committed");
    if (shape < 0 || shape >= (int)scene->meshes.size()) return fail(RTW_ERR_INVALID, "bad shape index");
    if (material_id < 0 || material_id >= RTW_MAX_MESH_MATERIALS) return fail(RTW_ERR_INVALID, "bad material id");
    if (width <= 0 || height <= 0 || (channels != 3 && channels != 4)) return fail(RTW_ERR_INVALID, "bad texture format");
    rtw::HostMesh& m = *scene->meshes[(size_t)shape];
    if ((int)m.textures.size() <= material_id) { m.textures.resize((size_t)material_id + 1); m.texture_paths.resize((size_t)material_id + 1); }
    rtw::HostTexture t; t.width = width; t.height = height; t.valid = true;
    t.rgba8.resize((size_t)width * (size_t)height);
    for (size_t i = 0; i < t.rgba8.size(); i++) {
        const uint8_t* s = texels + i * (size_t)channels;
        t.rgba8[i] = (uint32_t)s[0] | ((uint32_t)s[1] << 8) | ((uint32_t)s[2] << 16) | ((uint32_t)(channels == 4 ? s[3] : 255) << 24);
    }
    m.textures[(size_t)material_id] = std::move(t);
    // the reference sizes its texture table by the triangle count once an MTL is present
    if (m.n_textures_slots == 0) m.n_textures_slots = m.n_tris();
    return RTW_OK;
}

int rtw_scene_set_material(rtw_scene* scene, int shape, const rtw_material_node* nodes, int n_nodes)
{
    if (!scene || !nodes) return fail(RTW_ERR_INVALID, "null argument");
    if (scene->committed) return fail(RTW_ERR_STATE, "scene already committed");
    if (shape < 0 || shape >= (int)scene->meshes.size()) return fail(RTW_ERR_INVALID, "bad shape index");
    if (n_nodes < 1 || n_nodes > RTW_MAX_MATERIAL_NODES) return fail(RTW_ERR_LIMIT, "material node count out of range");
    // children after parents; nesting of Combine nodes bounded by the device evaluation stack
    std::vector<int> combine_depth((size_t)n_nodes, 0);
    for (int i = 0; i < n_nodes; i++) {
        const rtw_material_node& n = nodes[i];
        if (n.type < RTW_MAT_DIFFUSE || n.type > RTW_MAT_NULL) return fail(RTW_ERR_INVALID, "unknown material type");
        if (n.type == RTW_MAT_BLEND || n.type == RTW_MAT_COMBINE) {
            if (n.child_a <= i || n.child_a >= n_nodes || n.child_b <= i || n.child_b >= n_nodes)
                return fail(RTW_ERR_INVALID, "material children must follow their parent");
            const int d = combine_depth[(size_t)i] + (n.type == RTW_MAT_COMBINE ? 1 : 0);
            if (d > 2) return fail(RTW_ERR_LIMIT, "Combine nesting deeper than 2");
            if (d > combine_depth[(size_t)n.child_a]) combine_depth[(size_t)n.child_a] = d;
            if (d > combine_depth[(size_t)n.child_b]) combine_depth[(size_t)n.child_b] = d;
        }
    }
    rtw::HostMesh& m = *scene->meshes[(size_t)shape];
    m.material.resize((size_t)n_nodes);
    std::memcpy(m.material.data(), nodes, (size_t)n_nodes * sizeof(rtw_material_node));
    return RTW_OK;
}

int rtw_scene_set_traversal(rtw_scene* scene, int mode)
{
    if (!scene) return fail(RTW_ERR_INVALID, "scene is null");
    if (mode != 0 && mode != 1) return fail(RTW_ERR_INVALID, "traversal must be 0 (binary preorder walk only) or 1 (accelerated walks allowed)");
    scene->traversal = mode;
    if (scene->committed && scene->ctx) {
        HIP_TRY(hipSetDevice(scene->ctx->device));
        HIP_TRY(hipStreamSynchronize(scene->ctx->stream));
        HIP_TRY(hipMemcpy((char*)scene->d_scene + offsetof(RtwSceneDev, traversal), &scene->traversal, sizeof(int32_t), hipMemcpyHostToDevice));
    }
    return RTW_OK;
}

int rtw_scene_set_prune(rtw_scene* scene, int enabled)
{
    if (!scene) return fail(RTW_ERR_INVALID, "scene is null");
    scene->prune = enabled ? 1 : 0;
    if (scene->committed && scene->ctx) {
        HIP_TRY(hipSetDevice(scene->ctx->device));
        HIP_TRY(hipStreamSynchronize(scene->ctx->stream));
        HIP_TRY(hipMemcpy((char*)scene->d_scene + offsetof(RtwSceneDev, prune), &scene->prune, sizeof(int32_t), hipMemcpyHostToDevice));
    }
    return RTW_OK;
}

static double commit_now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define COMMIT_MARK(what) do { if (trace) { (void)hipStreamSynchronize(scene->ctx->stream); const double t_ = commit_now(); std::fprintf(stderr, "  commit: %-28s %.3f ms\n", what, t_ - t_last); t_last = t_; } } while (0)
int rtw_scene_commit(rtw_scene* scene)
{
    static const bool trace = std::getenv("RTW_COMMIT_TRACE") != nullptr;
    double t_last = commit_now();
    if (!scene) return fail(RTW_ERR_INVALID, "scene is null");
    if (scene->committed) return fail(RTW_ERR_STATE, "scene already committed");
    if (!scene->ctx) {      // host-only scene: build the flattened trees for inspection, nothing to upload
        for (auto& m : scene->meshes) { rtw::build_tree(*m); rtw::build_tnodes(*m, RTW_TNODES_TOP_BUDGET); rtw::build_flat(*m); }
        scene->committed = true;
        return RTW_OK;
    }
    HIP_TRY(hipSetDevice(scene->ctx->device));
    std::unique_ptr<RtwSceneDev> h(new RtwSceneDev());
    std::memset(h.get(), 0, sizeof(RtwSceneDev));
    h->n_shapes = (int)scene->meshes.size();
    h->prune = scene->prune;
    h->traversal = scene->traversal;
#ifdef RTW_DEBUG_OPTIONS
    if (const char* dm = std::getenv("RTW_DEBUG_TABLE_MASK")) h->debug_table_mask = (int32_t)std::strtol(dm, nullptr, 0);
#endif
    h->unit_table = scene->ctx->d_unit;
    h->gamma_thr = scene->ctx->d_gamma;
    h->texel_lut = scene->ctx->d_lut;
    h->stats = scene->ctx->d_stats;
    bool textured_mesh_before = false;
    for (size_t s = 0; s < scene->meshes.size(); s++) {
        rtw::HostMesh& m = *scene->meshes[s];
        RtwShapeDev& d = h->shapes[s];
        int rc;
        const bool on_device = scene->ctx->device_build && m.kind == RTW_SHAPE_MESH && m.n_tris() > 0;
        if (on_device) {
            // KdNode::Build's recursion and the layouts derived from the tree run on the device (rtw_build_kernels.h); the host keeps copies of
            // the small arrays for the screen bins, the older pipelines' collapsed trees and the inspection calls
            rtw::DeviceBuildIn in;
            in.points = reinterpret_cast<const float*>(m.points.data()); in.n_points = (int)m.points.size();       // (.data(): an OBJ without vt / vn lines has empty arrays)
            in.texcoords = reinterpret_cast<const float*>(m.texcoords.data()); in.n_texcoords = (int)m.texcoords.size();
            in.normals = reinterpret_cast<const float*>(m.normals.data()); in.n_normals = (int)m.normals.size();
            in.idx_p = m.point_idx.data(); in.idx_t = m.texcoord_idx.data(); in.idx_n = m.normal_idx.data(); in.tri_material = m.poly_material.data(); in.n_tris = m.n_tris();
            rtw::DeviceBuildOut o;
            const hipError_t be = (hipError_t)rtw::device_build_mesh(in, RTW_TNODES_TOP_BUDGET, &o, scene->ctx->stream);
            if (be != hipSuccess) return hip_fail(be, "device tree build");
            COMMIT_MARK("device_build_mesh");
            scene->allocs.push_back(o.nodes); scene->allocs.push_back(o.tnodes); scene->allocs.push_back(o.tris); scene->allocs.push_back(o.shade);
            for (int l = 0; l < 3; l++) scene->allocs.push_back(o.flat[l]);
            m.nodes.resize((size_t)o.n_nodes); m.tnodes.resize((size_t)o.n_nodes); m.tris.resize((size_t)in.n_tris); m.shade.resize((size_t)in.n_tris);
            HIP_TRY(hipMemcpy(m.nodes.data(), o.nodes, m.nodes.size() * sizeof(RtwNode), hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(m.tnodes.data(), o.tnodes, m.tnodes.size() * sizeof(RtwPNode), hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(m.tris.data(), o.tris, m.tris.size() * sizeof(RtwTri), hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(m.shade.data(), o.shade, m.shade.size() * sizeof(RtwShade), hipMemcpyDeviceToHost));
            m.tnodes_top = o.tnodes_top; m.max_depth = o.max_depth;
            for (int l = 0; l < 3; l++) {
                m.flat_n[l] = o.flat_n[l]; m.flat_pad[l] = o.flat_pad[l];
                m.flat[l].resize((size_t)6 * (size_t)o.flat_pad[l]);
                HIP_TRY(hipMemcpy(m.flat[l].data(), o.flat[l], m.flat[l].size() * sizeof(float), hipMemcpyDeviceToHost));
                d.flat[l] = o.flat[l]; d.flat_n[l] = o.flat_n[l]; d.flat_pad[l] = o.flat_pad[l];
            }
            d.nodes = o.nodes; d.tnodes = o.tnodes; d.tnodes_top = o.tnodes_top; d.tris = o.tris; d.shade = o.shade;
            scene->d_nodes_of.resize(scene->meshes.size(), nullptr); scene->d_tris_of.resize(scene->meshes.size(), nullptr);
            scene->d_nodes_of[s] = o.nodes; scene->d_tris_of[s] = o.tris;
            COMMIT_MARK("read back");
        } else {
            rtw::build_tree(m);
            COMMIT_MARK("host build_tree");
        }
        if (!on_device && (rc = upload(scene, m.nodes, &d.nodes)) != RTW_OK) return rc;
        if (!on_device) {
            rtw::build_tnodes(m, RTW_TNODES_TOP_BUDGET);
            if ((rc = upload(scene, m.tnodes, &d.tnodes)) != RTW_OK) return rc;
            d.tnodes_top = m.tnodes_top;
            rtw::build_flat(m);
            for (int l = 0; l < 3; l++) {
                if ((rc = upload(scene, m.flat[l], &d.flat[l])) != RTW_OK) return rc;
                d.flat_n[l] = m.flat_n[l]; d.flat_pad[l] = m.flat_pad[l];
            }
            if ((rc = upload(scene, m.tris, &d.tris)) != RTW_OK) return rc;
            if ((rc = upload(scene, m.shade, &d.shade)) != RTW_OK) return rc;
        }
        if (m.kind == RTW_SHAPE_MESH && !m.tris.empty()) {      // the triangles' planes on their own (see RtwShapeDev::planes)
            std::vector<float> planes(m.tris.size() * 4);
            for (size_t t = 0; t < m.tris.size(); t++) { planes[t * 4] = m.tris[t].nx; planes[t * 4 + 1] = m.tris[t].ny; planes[t * 4 + 2] = m.tris[t].nz; planes[t * 4 + 3] = m.tris[t].d1; }
            if ((rc = upload(scene, planes, &d.planes)) != RTW_OK) return rc;
        }
        COMMIT_MARK("layouts + uploads");
        std::vector<uint32_t> atlas;
        for (size_t t = 0; t < m.textures.size() && t < RTW_DEV_MAX_TEXTURES; t++) {
            const rtw::HostTexture& tx = m.textures[t];
            if (!tx.valid) continue;
            d.textures[t].offset = (uint32_t)atlas.size();
            d.textures[t].width = tx.width; d.textures[t].height = tx.height; d.textures[t].valid = 1;
            atlas.insert(atlas.end(), tx.rgba8.begin(), tx.rgba8.end());
        }
        if ((rc = upload(scene, atlas, &d.texels)) != RTW_OK) return rc;
        COMMIT_MARK("texture atlas");
        for (int k = 0; k < 3; k++) { d.bmin[k] = m.bmin[k]; d.bmax[k] = m.bmax[k]; }
        d.n_nodes = (int)m.nodes.size(); d.n_tris = (int)m.tris.size();
        d.n_textures = m.n_textures_slots;
        d.has_material = m.material.empty() ? 0 : 1;
        d.n_material_nodes = (int)m.material.size();
        for (size_t k = 0; k < m.material.size(); k++) d.material[k] = m.material[k];
        for (const RtwMaterialNode& mn : m.material) {
            if (mn.type == RTW_MAT_EMISSIVE) scene->has_emissive = true;
            if (!std::isfinite(mn.r) || !std::isfinite(mn.g) || !std::isfinite(mn.b) || !std::isfinite(mn.param)) scene->materials_finite = false;
        }
        d.kind = m.kind; d.radius = m.radius;
        for (int k = 0; k < 3; k++) { d.pa[k] = m.pa[k]; d.pb[k] = m.pb[k]; d.pc[k] = m.pc[k]; d.pn[k] = m.pn[k]; }
        d.pd1 = m.pd1;
        if (m.kind != RTW_SHAPE_MESH) {
            scene->has_analytic = true;
            if (textured_mesh_before) scene->texture_carry = true;
        } else {
            for (const rtw::HostTexture& tx : m.textures) if (tx.valid) textured_mesh_before = true;
        }
    }
    void* dsc = nullptr;
    HIP_TRY(hipMalloc(&dsc, sizeof(RtwSceneDev)));
    scene->allocs.push_back(dsc);
    HIP_TRY(hipMemcpy(dsc, h.get(), sizeof(RtwSceneDev), hipMemcpyHostToDevice));
    scene->d_scene = (RtwSceneDev*)dsc;
    scene->committed = true;
    return RTW_OK;
}

int rtw_scene_mesh_info(const rtw_scene* scene, int shape, int32_t info[8], float shape_bounds6[6])
{
    if (!scene || shape < 0 || shape >= (int)scene->meshes.size()) return fail(RTW_ERR_INVALID, "bad shape index");
    const rtw::HostMesh& m = *scene->meshes[(size_t)shape];
    if (info) {
        info[0] = (int)m.points.size(); info[1] = (int)m.texcoords.size(); info[2] = (int)m.normals.size();
        info[3] = m.n_tris(); info[4] = (int)m.material_names.size(); info[5] = (int)m.nodes.size();
        int ntex = 0; for (const auto& t : m.textures) ntex += t.valid ? 1 : 0;
        info[6] = ntex; info[7] = m.max_depth;
    }
    if (shape_bounds6) for (int k = 0; k < 3; k++) { shape_bounds6[k] = m.bmin[k]; shape_bounds6[k + 3] = m.bmax[k]; }
    return RTW_OK;
}

int rtw_scene_mesh_nodes(const rtw_scene* scene, int shape, float* bounds6, int32_t* skip, int32_t* tri, int max_nodes)
{
    if (!scene || shape < 0 || shape >= (int)scene->meshes.size()) return fail(RTW_ERR_INVALID, "bad shape index");
    if (!scene->committed) return fail(RTW_ERR_STATE, "scene not committed");
    const rtw::HostMesh& m = *scene->meshes[(size_t)shape];
    const int n = (int)m.nodes.size() < max_nodes ? (int)m.nodes.size() : max_nodes;
    for (int i = 0; i < n; i++) {
        const RtwNode& nd = m.nodes[(size_t)i];
        if (bounds6) { float* b = bounds6 + (size_t)i * 6; b[0] = nd.min_x; b[1] = nd.min_y; b[2] = nd.min_z; b[3] = nd.max_x; b[4] = nd.max_y; b[5] = nd.max_z; }
        if (skip) skip[i] = nd.skip;
        if (tri) tri[i] = nd.tri >= 0 ? m.tris[(size_t)nd.tri].orig : -1;
    }
    return (int)m.nodes.size();
}

int rtw_scene_mesh_flat(const rtw_scene* scene, int shape, int level, float* boxes6, int max_entries)
{
    if (!scene || shape < 0 || shape >= (int)scene->meshes.size()) return fail(RTW_ERR_INVALID, "bad shape index");
    if (!scene->committed) return fail(RTW_ERR_STATE, "scene not committed");
    if (level < 0 || level > 2) return fail(RTW_ERR_INVALID, "level must be 0, 1 or 2");
    const rtw::HostMesh& m = *scene->meshes[(size_t)shape];
    const int n = m.flat_n[level] < max_entries ? m.flat_n[level] : max_entries;
    if (boxes6) for (int i = 0; i < n; i++) for (int c = 0; c < 6; c++)
        boxes6[(size_t)i * 6 + c] = m.flat[level][(size_t)i * 6 + (size_t)(c < 3 ? 2 * c : 2 * (c - 3) + 1)];
    return m.flat_n[level];
}

int rtw_scene_mesh_bins(const rtw_scene* scene, int shape, int width, int height, int bin_w, int bin_h,
                        uint32_t* offsets, int64_t max_offsets, uint32_t* entries, int64_t max_entries, int64_t* counts2)
{
    if (!scene || shape < 0 || shape >= (int)scene->meshes.size()) return fail(RTW_ERR_INVALID, "bad shape index");
    if (!scene->committed) return fail(RTW_ERR_STATE, "scene not committed");
    if (!counts2) return fail(RTW_ERR_INVALID, "null argument");
    std::vector<uint32_t> off, ent;
    bool ok;
    if (scene->ctx && scene->ctx->device_build && scene->d_nodes_of.size() > (size_t)shape && scene->d_nodes_of[(size_t)shape]) {
        // what the renderer uses: the lists built on the device
        if (width <= 0 || height <= 0 || bin_w <= 0 || bin_h <= 0) return fail(RTW_ERR_INVALID, "bad bin geometry");
        HIP_TRY(hipSetDevice(scene->ctx->device));
        const size_t n_bins = (size_t)((width + bin_w - 1) / bin_w) * (size_t)((height + bin_h - 1) / bin_h);
        off.resize(n_bins + 1);
        uint32_t *d_off = nullptr, *d_ent = nullptr; int has = 0;
        const hipError_t be = (hipError_t)rtw::device_build_bins(scene->d_nodes_of[(size_t)shape], scene->d_tris_of[(size_t)shape], (int)scene->meshes[(size_t)shape]->nodes.size(),
                                                                 width, height, bin_w, bin_h, &d_off, &d_ent, off.data(), &has, scene->ctx->stream);
        if (be != hipSuccess) return hip_fail(be, "device bins build");
        ok = has != 0;
        if (ok) {
            ent.resize(off.back() ? off.back() : 1, 0u);
            const hipError_t ce = off.back() ? hipMemcpy(ent.data(), d_ent, (size_t)off.back() * 4, hipMemcpyDeviceToHost) : hipSuccess;
            (void)hipFree(d_off); (void)hipFree(d_ent);
            if (ce != hipSuccess) return hip_fail(ce, "bins read-back");
        }
    } else {
        ok = rtw::build_bins(*scene->meshes[(size_t)shape], width, height, bin_w, bin_h, off, ent);
    }
    counts2[0] = ok ? (int64_t)off.size() : 0; counts2[1] = ok ? (int64_t)off.back() : 0;
    if (!ok) return 0;                   // this mesh gets no bins (not wholly in front of the camera, or the frame does not tile)
    if (offsets) std::memcpy(offsets, off.data(), (size_t)std::min<int64_t>(max_offsets, (int64_t)off.size()) * 4);
    if (entries) std::memcpy(entries, ent.data(), (size_t)std::min<int64_t>(max_entries, (int64_t)off.back()) * 4);
    return 1;
}

// ---- ray-level queries ------------------------------------------------------------------------------------
namespace {
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
};
int need_committed(rtw_scene* scene)
{
    if (!scene) return fail(RTW_ERR_INVALID, "scene is null");
    if (!scene->ctx) return fail(RTW_ERR_NO_DEVICE, "host-only scene (created without a context): no device, no CPU fallback");
    if (!scene->committed) return fail(RTW_ERR_STATE, "scene not committed");
    hipError_t e = hipSetDevice(scene->ctx->device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    return RTW_OK;
}
}  // namespace

int rtw_trace_closest(rtw_scene* scene, const float* rays, int64_t n, float* hits11, int32_t* shape, int32_t* tri)
{
    int rc = need_committed(scene); if (rc != RTW_OK) return rc;
    if (n < 0 || (n > 0 && (!rays || !hits11 || !shape || !tri))) return fail(RTW_ERR_INVALID, "null argument");
    if (n == 0) return RTW_OK;
    hipStream_t st = scene->ctx->stream;
    DevBuf dr, dh, ds, dt;
    HIP_TRY(dr.alloc((size_t)n * 28)); HIP_TRY(dh.alloc((size_t)n * 44)); HIP_TRY(ds.alloc((size_t)n * 4)); HIP_TRY(dt.alloc((size_t)n * 4));
    HIP_TRY(hipMemcpyAsync(dr.p, rays, (size_t)n * 28, hipMemcpyHostToDevice, st));
    hipError_t e = (hipError_t)rtw::launch_closest(scene->d_scene, (const float*)dr.p, n, (float*)dh.p, (int*)ds.p, (int*)dt.p, scene->ctx->stats_enabled, st);
    if (e != hipSuccess) return hip_fail(e, "closest_kernel launch");
    HIP_TRY(hipMemcpyAsync(hits11, dh.p, (size_t)n * 44, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(shape, ds.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(tri, dt.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return RTW_OK;
}

int rtw_ray_trace(rtw_scene* scene, const float* rays, const uint32_t* keys2, int64_t n, int max_bounce, int use_base_color,
                  uint32_t seed, int width, int height, float* rgb)
{
    int rc = need_committed(scene); if (rc != RTW_OK) return rc;
    if (n < 0 || (n > 0 && (!rays || !keys2 || !rgb))) return fail(RTW_ERR_INVALID, "null argument");
    if (max_bounce < 0 || max_bounce > RTW_MAX_BOUNCE) return fail(RTW_ERR_LIMIT, "max_bounce out of range");
    if (width <= 0 || height <= 0) return fail(RTW_ERR_INVALID, "bad image size");
    if (n == 0) return RTW_OK;
    hipStream_t st = scene->ctx->stream;
    DevBuf dr, dk, dc;
    HIP_TRY(dr.alloc((size_t)n * 28)); HIP_TRY(dk.alloc((size_t)n * 8)); HIP_TRY(dc.alloc((size_t)n * 12));
    HIP_TRY(hipMemcpyAsync(dr.p, rays, (size_t)n * 28, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dk.p, keys2, (size_t)n * 8, hipMemcpyHostToDevice, st));
    rc = ensure_workspace(scene->ctx, rtw::level_workspace_bytes(n, max_bounce)); if (rc != RTW_OK) return rc;
    scene->ctx->clean_ws = nullptr;      // the level store below overwrites the pipelines' counters
    hipError_t e = (hipError_t)rtw::launch_ray_trace(scene->d_scene, (const float*)dr.p, (const uint32_t*)dk.p, n, max_bounce, use_base_color, seed,
                                                     (unsigned long long)width * (unsigned long long)height, (float*)dc.p, scene->ctx->d_workspace,
                                                     scene->ctx->stats_enabled, st);
    if (e != hipSuccess) return hip_fail(e, "ray_trace_kernel launch");
    HIP_TRY(hipMemcpyAsync(rgb, dc.p, (size_t)n * 12, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return RTW_OK;
}

int rtw_texture_sample(rtw_scene* scene, int shape, int material_id, const float* uv, int64_t n, float* rgba)
{
    int rc = need_committed(scene); if (rc != RTW_OK) return rc;
    if (shape < 0 || shape >= (int)scene->meshes.size()) return fail(RTW_ERR_INVALID, "bad shape index");
    const rtw::HostMesh& m = *scene->meshes[(size_t)shape];
    if (material_id < 0 || material_id >= (int)m.textures.size() || material_id >= RTW_DEV_MAX_TEXTURES || !m.textures[(size_t)material_id].valid)
        return fail(RTW_ERR_INVALID, "material has no texture");
    if (n < 0 || (n > 0 && (!uv || !rgba))) return fail(RTW_ERR_INVALID, "null argument");
    if (n == 0) return RTW_OK;
    hipStream_t st = scene->ctx->stream;
    DevBuf du, dc;
    HIP_TRY(du.alloc((size_t)n * 8)); HIP_TRY(dc.alloc((size_t)n * 16));
    HIP_TRY(hipMemcpyAsync(du.p, uv, (size_t)n * 8, hipMemcpyHostToDevice, st));
    hipError_t e = (hipError_t)rtw::launch_texture_sample(scene->d_scene, shape, material_id, (const float*)du.p, n, (float*)dc.p, st);
    if (e != hipSuccess) return hip_fail(e, "texture_sample_kernel launch");
    HIP_TRY(hipMemcpyAsync(rgba, dc.p, (size_t)n * 16, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return RTW_OK;
}

// ---- framebuffer -------------------------------------------------------------------------------------------
int rtw_framebuffer_create(rtw_context* ctx, int width, int height, rtw_framebuffer** out)
{
    if (!ctx || !out || width <= 0 || height <= 0) return fail(RTW_ERR_INVALID, "bad argument");
    if ((int64_t)width * height > (int64_t)1 << 30) return fail(RTW_ERR_LIMIT, "framebuffer too large");
    HIP_TRY(hipSetDevice(ctx->device));
    std::unique_ptr<rtw_framebuffer> fb(new rtw_framebuffer());
    fb->ctx = ctx; fb->width = width; fb->height = height; fb->owned = true;
    const size_t n = (size_t)width * (size_t)height;
    HIP_TRY(hipMalloc(&fb->accum, n * 16));
    HIP_TRY(hipMalloc(&fb->argb, n * 4));
    HIP_TRY(hipMemsetAsync(fb->accum, 0, n * 16, ctx->stream));
    HIP_TRY(hipMemsetAsync(fb->argb, 0, n * 4, ctx->stream));
    *out = fb.release();
    return RTW_OK;
}

int rtw_framebuffer_wrap(rtw_context* ctx, int width, int height, void* accum_dev, void* argb_dev, rtw_framebuffer** out)
{
    if (!ctx || !out || width <= 0 || height <= 0 || !accum_dev || !argb_dev) return fail(RTW_ERR_INVALID, "bad argument");
    rtw_framebuffer* fb = new rtw_framebuffer();
    fb->ctx = ctx; fb->width = width; fb->height = height; fb->owned = false;
    fb->accum = accum_dev; fb->argb = argb_dev;
    *out = fb;
    return RTW_OK;
}

int rtw_framebuffer_destroy(rtw_framebuffer* fb)
{
    if (!fb) return RTW_OK;
    if (fb->owned) {
        (void)hipSetDevice(fb->ctx->device);
        (void)hipStreamSynchronize(fb->ctx->stream);
        (void)hipFree(fb->accum); (void)hipFree(fb->argb);
    }
    delete fb;
    return RTW_OK;
}

int rtw_framebuffer_clear(rtw_framebuffer* fb)
{
    if (!fb) return fail(RTW_ERR_INVALID, "framebuffer is null");
    HIP_TRY(hipSetDevice(fb->ctx->device));
    const size_t n = (size_t)fb->width * (size_t)fb->height;
    HIP_TRY(hipMemsetAsync(fb->accum, 0, n * 16, fb->ctx->stream));
    HIP_TRY(hipMemsetAsync(fb->argb, 0, n * 4, fb->ctx->stream));
    return RTW_OK;
}

int rtw_framebuffer_read_float(rtw_framebuffer* fb, float* accum4)
{
    if (!fb || !accum4) return fail(RTW_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(fb->ctx->device));
    const size_t n = (size_t)fb->width * (size_t)fb->height;
    HIP_TRY(hipMemcpyAsync(accum4, fb->accum, n * 16, hipMemcpyDeviceToHost, fb->ctx->stream));
    HIP_TRY(hipStreamSynchronize(fb->ctx->stream));
    for (size_t i = 0; i < n; i++) {        // the count travels as int bits on the device
        int32_t c; std::memcpy(&c, &accum4[i * 4 + 3], 4);
        accum4[i * 4 + 3] = (float)c;
    }
    return RTW_OK;
}

int rtw_framebuffer_resolve_argb(rtw_framebuffer* fb, uint32_t* argb)
{
    if (!fb || !argb) return fail(RTW_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(fb->ctx->device));
    const size_t n = (size_t)fb->width * (size_t)fb->height;
    HIP_TRY(hipMemcpyAsync(argb, fb->argb, n * 4, hipMemcpyDeviceToHost, fb->ctx->stream));
    HIP_TRY(hipStreamSynchronize(fb->ctx->stream));
    return RTW_OK;
}

int rtw_framebuffer_device_pointers(rtw_framebuffer* fb, void** accum_dev, void** argb_dev)
{
    if (!fb) return fail(RTW_ERR_INVALID, "framebuffer is null");
    if (accum_dev) *accum_dev = fb->accum;
    if (argb_dev) *argb_dev = fb->argb;
    return RTW_OK;
}

// Bins of every shape of the scene for one frame size and bin shape, built and uploaded on first use.
static int scene_bins(rtw_scene* scene, int width, int height, int bin_w, int bin_h, rtw_scene::BinSet** out)
{
    for (rtw_scene::BinSet& b : scene->bin_sets)
        if (b.width == width && b.height == height && b.bin_w == bin_w && b.bin_h == bin_h) { *out = &b; return RTW_OK; }
    std::vector<RtwBinsDev> h(scene->meshes.size());
    const size_t n_bins = (size_t)((width + bin_w - 1) / bin_w) * (size_t)((height + bin_h - 1) / bin_h);
    std::vector<uint32_t> weight(n_bins, 0u);
    for (size_t s = 0; s < scene->meshes.size(); s++) {
        std::vector<uint32_t> off, ent;
        h[s].off = nullptr; h[s].ent = nullptr;
        if (scene->meshes[s]->kind != RTW_SHAPE_MESH) { for (auto& w : weight) w += 1u; continue; }      // tested by every sample of every tile
        if (scene->ctx->device_build && scene->d_nodes_of.size() > s && scene->d_nodes_of[s]) {
            // the same lists from the device: counted, scanned, filled and sorted there (rtw_build_kernels.h); only the offsets come back (tile weights)
            off.resize(n_bins + 1);
            uint32_t *d_off = nullptr, *d_ent = nullptr; int has = 0;
            const hipError_t be = (hipError_t)rtw::device_build_bins(scene->d_nodes_of[s], scene->d_tris_of[s], (int)scene->meshes[s]->nodes.size(), width, height, bin_w, bin_h,
                                                                     &d_off, &d_ent, off.data(), &has, scene->ctx->stream);
            if (be != hipSuccess) return hip_fail(be, "device bins build");
            if (!has) { for (auto& w : weight) w += 1000u; continue; }
            scene->allocs.push_back(d_off); scene->allocs.push_back(d_ent);
            for (size_t b = 0; b < n_bins; b++) weight[b] += off[b + 1] - off[b];
            h[s].off = d_off; h[s].ent = d_ent;
            continue;
        }
        if (!rtw::build_bins(*scene->meshes[s], width, height, bin_w, bin_h, off, ent)) { for (auto& w : weight) w += 1000u; continue; }
        for (size_t b = 0; b < n_bins; b++) weight[b] += off[b + 1] - off[b];
        int rc;
        if ((rc = upload(scene, off, &h[s].off)) != RTW_OK) return rc;
        if ((rc = upload(scene, ent, &h[s].ent)) != RTW_OK) return rc;
    }
    // the camera's per-column dx and per-row dy, as ThreadWorker_Render computes them for every pixel (Src/RayTracerProgram.cpp:141-142)
    std::vector<float> dx((size_t)width), dy((size_t)height);
    const float aspect = (float)width / (float)height;
    for (int x = 0; x < width; x++) dx[(size_t)x] = -(float)(x - width / 2) / (width * 2) * aspect;
    for (int y = 0; y < height; y++) dy[(size_t)y] = -(float)(y - height / 2) / (height * 2);
    const RtwBinsDev* d = nullptr; const float* ddx = nullptr; const float* ddy = nullptr;
    rtw_scene::BinSet bsnew;
    int rc = upload(scene, h, &d); if (rc != RTW_OK) return rc;
    bsnew.weight = weight;
    if ((rc = upload(scene, dx, &ddx)) != RTW_OK) return rc;
    if ((rc = upload(scene, dy, &ddy)) != RTW_OK) return rc;
    bsnew.width = width; bsnew.height = height; bsnew.bin_w = bin_w; bsnew.bin_h = bin_h; bsnew.d_bins = const_cast<RtwBinsDev*>(d); bsnew.d_dx = ddx; bsnew.d_dy = ddy;
    scene->bin_sets.push_back(bsnew);
    *out = &scene->bin_sets.back();
    return RTW_OK;
}

// Job table of one launch shape (p already tiled by choose_tiles): the launch's wave tiles in order of decreasing bin-list
// length (stable), so that the waves that take longest start first and the sky-only tiles (no leaf listed: they come last and can
// go to the lean sky kernel) fill the tail; with several sub-samples a tile with a long list is one job per sub-sample
// (sub-sample + 1 in bits 24..27).  Built on the host on first use, cached per (task partition, sub-sample count).
static int scene_job_table(rtw_scene* scene, rtw_scene::BinSet& bs, const RtwRenderParams& p, int spp, const rtw_scene::JobTable** out)
{
    *out = nullptr;
    const bool linear = p.world <= 1;
    if (linear && (p.row0 != 0 || p.nrows != p.height)) return RTW_OK;          // a partial range: tiles as numbered
    const int key_rows = linear ? 0 : p.task_rows, key_rank = linear ? 0 : p.rank, key_world = linear ? 1 : p.world;
    for (const rtw_scene::JobTable& t : bs.jobs)
        if (t.task_rows == key_rows && t.rank == key_rank && t.world == key_world && t.spp == spp) { *out = &t; return RTW_OK; }
    const int n_tiles = p.count >> 6;
    if (n_tiles >= (1 << 24)) return RTW_OK;                                    // tile numbers must fit 24 bits
    const int tiles_per_row = p.tiles_per_row;
    std::vector<uint32_t> w((size_t)n_tiles, 0u);
    for (int wt = 0; wt < n_tiles; wt++) {
        const int band = wt / tiles_per_row, tx = wt - band * tiles_per_row;
        const int vr = band * p.tile_h;
        int y;
        if (linear) y = p.row0 + vr;
        else { const int j = vr / p.task_rows, r = vr - j * p.task_rows; y = (j * p.world + p.rank) * p.task_rows + r; }
        if (vr >= p.nrows || y >= p.height) continue;                           // a dead tile (past the last task's rows)
        w[(size_t)wt] = bs.weight[(size_t)(y / p.tile_h) * (size_t)tiles_per_row + (size_t)tx];
    }
    std::vector<uint32_t> by_weight((size_t)n_tiles);
    for (int i = 0; i < n_tiles; i++) by_weight[(size_t)i] = (uint32_t)i;
    std::stable_sort(by_weight.begin(), by_weight.end(), [&](uint32_t a, uint32_t b) { return w[a] > w[b]; });
    std::vector<uint32_t> order;
    int n_busy = 0;
    for (uint32_t t : by_weight) {
        if (spp > 1 && w[t] >= 40u) for (int i = 0; i < spp; i++) order.push_back(t | ((uint32_t)(i + 1) << 24));
        else order.push_back(t);
        if (w[t] > 0u) n_busy = (int)order.size();
    }
    rtw_scene::JobTable jt; jt.task_rows = key_rows; jt.rank = key_rank; jt.world = key_world; jt.spp = spp; jt.d_order = nullptr;
    int rc = upload(scene, order, &jt.d_order); if (rc != RTW_OK) return rc;
    jt.n_jobs = (int)order.size(); jt.n_busy = n_busy;
    bs.jobs.push_back(jt);
    *out = &bs.jobs.back();
    return RTW_OK;
}

// Tiled work mapping of the bins + wave pipeline: the rows of the launch are cut into tile_w x tile_h tiles (64 pixels,
// one wave each) that coincide with bins of the screen.  Returns false when the launch's rows cannot be tiled that way.
static bool choose_tiles(RtwRenderParams& p)
{
    const int W = p.width, H = p.height;
    if (W <= 0 || H <= 0) return false;
    long long vrows;                     // rows of this launch (virtual rows of the rank's tasks, or screen rows of the range)
    int row0 = 0;
    if (p.world <= 1 && p.task_rows == 0) {
        if (p.begin % W != 0 || p.count % W != 0) return false;
        row0 = p.begin / W; vrows = p.count / W;
    } else if (p.world <= 1) {
        row0 = 0; vrows = H;
    } else {
        vrows = p.count / W;
    }
    for (int th = 4; th >= 1; th >>= 1) {
        const int tw = 64 / th;
        if (W % tw != 0 || H % th != 0 || row0 % th != 0) continue;
        if (p.world > 1 && p.task_rows % th != 0) continue;
        if (p.world <= 1 && vrows % th != 0) continue;
        const long long bands = (vrows + th - 1) / th;
        const long long count = bands * (W / tw) * 64;
        if (count > INT32_MAX) return false;
        p.tile_w = tw; p.tile_h = th; p.tiles_per_row = W / tw;
        p.tile_shift = tw == 16 ? 4 : (tw == 32 ? 5 : 6);
        p.row0 = row0; p.nrows = (int)vrows;
        p.count = (int)count;
        return true;
    }
    return false;
}

// ---- pass-batched pipeline (pipeline 4) -----------------------------------------------------------------------------------
// Tile mapping of a group launch: 16 x 4-pixel tiles (32 x 2 / 64 x 1 when a rank's task rows do not divide by 4), one wave each, aligned to
// the screen's bin grid.  Frames that do not tile get partial tiles at the right / bottom edge, a pixel range that does not start / end on
// a tile row gets dead lanes (range_begin / range_end).
static bool choose_group_tiles(RtwRenderParams& p, int range_begin, int range_end)
{
    const int W = p.width, H = p.height;
    if (W <= 0 || H <= 0) return false;
    int th = 4;
    if (p.world > 1) { th = p.task_rows % 4 == 0 ? 4 : (p.task_rows % 2 == 0 ? 2 : 1); }
    const int tw = 64 / th;
    long long vrows; int row0 = 0;
    if (p.world <= 1) {
        const int r_first = range_begin / W, r_last = range_end / W;
        row0 = (r_first / th) * th;
        vrows = (long long)r_last - row0 + 1;
    } else {
        vrows = p.count / W;            // the rank's virtual rows (whole tasks; rows past the frame's bottom are dead)
    }
    const long long bands = (vrows + th - 1) / th;
    const long long tiles_per_row = (W + tw - 1) / tw;
    const long long count = bands * tiles_per_row * 64;
    if (count > INT32_MAX || bands * tiles_per_row >= (1 << 24)) return false;
    p.tile_w = tw; p.tile_h = th; p.tiles_per_row = (int)tiles_per_row;
    p.tile_shift = tw == 16 ? 4 : (tw == 32 ? 5 : 6);
    p.row0 = row0; p.nrows = (int)vrows;
    p.count = (int)count;
    return true;
}

// busy / sky tile lists and the primary kernel's job list of one (task partition, sub-sample count), full frames only
static int scene_group_table(rtw_scene* scene, rtw_scene::BinSet& bs, const RtwRenderParams& p, int spp, const rtw_scene::GroupTable** out)
{
    *out = nullptr;
    const bool linear = p.world <= 1;
    const int key_rows = linear ? 0 : p.task_rows, key_rank = linear ? 0 : p.rank, key_world = linear ? 1 : p.world;
    for (const rtw_scene::GroupTable& t : bs.gtables)
        if (t.task_rows == key_rows && t.rank == key_rank && t.world == key_world && t.spp == spp) { *out = &t; return RTW_OK; }
    const int n_tiles = p.count >> 6;
    const int tiles_per_row = p.tiles_per_row;
    std::vector<uint32_t> w((size_t)n_tiles, 0u);
    std::vector<uint8_t> dead((size_t)n_tiles, 0);
    for (int wt = 0; wt < n_tiles; wt++) {
        const int band = wt / tiles_per_row, tx = wt - band * tiles_per_row;
        const int vr = band * p.tile_h;
        int y;
        if (linear) y = p.row0 + vr;
        else { const int j = vr / p.task_rows, r = vr - j * p.task_rows; y = (j * p.world + p.rank) * p.task_rows + r; }
        if (vr >= p.nrows || y >= p.height) { dead[(size_t)wt] = 1; continue; }
        w[(size_t)wt] = bs.weight[(size_t)(y / p.tile_h) * (size_t)tiles_per_row + (size_t)tx];
    }
    std::vector<uint32_t> busy, sky, jobs;
    for (int wt = 0; wt < n_tiles; wt++) {
        if (dead[(size_t)wt]) continue;
        (w[(size_t)wt] > 0u ? busy : sky).push_back((uint32_t)wt);
    }
    std::stable_sort(busy.begin(), busy.end(), [&](uint32_t a, uint32_t b) { return w[a] > w[b]; });
    for (size_t b = 0; b < busy.size(); b++) {
        // with several sub-samples a tile with a long list is one job PER sub-sample (its waves share the list walk)
        if (spp > 1 && w[busy[b]] >= 40u) for (int i = 0; i < spp; i++) jobs.push_back((uint32_t)b | ((uint32_t)(i + 1) << 24));
        else jobs.push_back((uint32_t)b);
    }
    rtw_scene::GroupTable gt; gt.task_rows = key_rows; gt.rank = key_rank; gt.world = key_world; gt.spp = spp;
    gt.d_busy = gt.d_sky = gt.d_jobs = nullptr;
    int rc;
    if ((rc = upload(scene, busy, &gt.d_busy)) != RTW_OK) return rc;
    if ((rc = upload(scene, sky, &gt.d_sky)) != RTW_OK) return rc;
    if ((rc = upload(scene, jobs, &gt.d_jobs)) != RTW_OK) return rc;
    gt.n_busy = (int)busy.size(); gt.n_sky = (int)sky.size(); gt.n_jobs = (int)jobs.size();
    bs.gtables.push_back(gt);
    *out = &bs.gtables.back();
    return RTW_OK;
}

static int ensure_group_workspace(rtw_context* cx, size_t bytes)
{
    void*& ws = cx->d_group_ws[cx->lane];
    size_t& have = cx->group_ws_bytes[cx->lane];
    cx->ws_refused = false;
    if (bytes <= have) return RTW_OK;
    if (bytes > cx->workspace_limit && !cx->ws_single) { cx->ws_refused = true; return fail(RTW_ERR_HIP, "group workspace above the context's workspace limit"); }      // (a one-pass group is always tried: the limit shapes groups, it does not refuse frames)
    // grow-only; growing waits for the streams (rtw_render_reserve does it ahead of a call that must not stall)
    HIP_TRY(hipStreamSynchronize(cx->stream));
    if (cx->aux_stream) HIP_TRY(hipStreamSynchronize(cx->aux_stream));
    for (int j = 1; j < RTW_MAX_PARTS; j++) if (cx->part_stream[j]) HIP_TRY(hipStreamSynchronize(cx->part_stream[j]));
    if (ws) { (void)hipFree(ws); ws = nullptr; have = 0; }
    cx->group_clean[cx->lane] = false;
    const hipError_t e = hipMalloc(&ws, bytes);
    if (e != hipSuccess) {      // no room on the device (a GPU shared with torch, other ranks ...): the caller forms a smaller group
        ws = nullptr; (void)hipGetLastError();
        cx->ws_refused = true;
        return hip_fail(e, "hipMalloc of the group workspace");
    }
    have = bytes;
    return RTW_OK;
}

static int check_render_args(const rtw_scene* scene, const rtw_framebuffer* fb, int max_bounce, int pass_index, int sub_samples)
{
    if (fb->ctx != scene->ctx) return fail(RTW_ERR_INVALID, "scene and framebuffer belong to different contexts");
    if (max_bounce < 0 || max_bounce > RTW_MAX_BOUNCE) return fail(RTW_ERR_LIMIT, "max_bounce out of range");
    if (sub_samples < 1 || sub_samples > 4) return fail(RTW_ERR_INVALID, "sub_samples must be 1..4");
    if (pass_index < 0) return fail(RTW_ERR_INVALID, "pass_index must be >= 0");
    return RTW_OK;
}

// Can this scene / context go through the pass-batched pipeline?  (The reference-order walk used for work counters, traversal 0, stays with the single kernel.)
static bool group_pipeline_ok(const rtw_scene* scene)
{
    return scene->ctx->pipeline == 4 && scene->traversal != 0;
}

// how many of the `remaining` passes the next group takes: enough for about group_paths paths per launch
static int group_passes(const rtw_context* cx, long long paths_per_pass, int remaining, int max_bounce, bool carry)
{
    long long k = 1;
    while (k < cx->group_max && k * paths_per_pass < cx->group_paths) k <<= 1;
    // memory: at most ~24 GiB of workspace (of 288) and 2^30 slots
    for (;;) {
        const size_t bytes = rtw::group_workspace_bytes((size_t)(paths_per_pass * k), max_bounce, carry, nullptr);
        if (k == 1 || (bytes <= ((size_t)24 << 30) && paths_per_pass * k < ((long long)1 << 30))) break;
        k >>= 1;
    }
    if (k < remaining && remaining < 2 * k) k = (remaining + 1) / 2;      // 20 passes at 16 per group: 10 + 10, not 16 + 4 (a small last group pays every launch's latency for little work)
    return (int)(k < remaining ? k : remaining);
}

// The next group of a call: how many of the `remaining` passes it takes (the policy above, under the cap a refusal of workspace has set) and whether it
// runs as two halves on two streams.
static int next_group(const rtw_context* cx, long long per_pass, int remaining, int max_bounce, bool carry, bool preview, int* parts)
{
    *parts = 1;
    if (preview) return 1;
    int k = group_passes(cx, per_pass > 0 ? per_pass : 1, remaining, max_bounce, carry);
    if (cx->group_cap > 0 && k > cx->group_cap) k = cx->group_cap;
    if (cx->group_split && k >= cx->split_min && !cx->stats_enabled && !cx->kernel_timing) {
        // as many parts as the option allows, each with at least split_min / 2 passes and split_paths paths
        int n = cx->group_parts;
        while (n > 1 && (k / n < (cx->split_min + 1) / 2 || per_pass * (k / n) < cx->split_paths)) n--;
        *parts = n;
    }
    return k;
}

// what rtw_render_passes and rtw_render_reserve share: the rank's share of the frame and its paths per pass (busy tiles x 64 x sub-samples; builds the
// screen bins and tile tables of this frame shape on first use)
struct GroupCall { RtwRenderParams p; long long per_pass; };
static int plan_group_call(rtw_scene* scene, rtw_framebuffer* fb, int task_rows, int rank, int world, int max_bounce, int first_pass, int sub_samples, GroupCall* out)
{
    rtw_context* cx = scene->ctx;
    if (task_rows < 1 || world < 1 || rank < 0 || rank >= world) return fail(RTW_ERR_INVALID, "bad task partition");
    int rc = check_render_args(scene, fb, max_bounce, first_pass, sub_samples); if (rc != RTW_OK) return rc;
    RtwRenderParams p; std::memset(&p, 0, sizeof p);
    const int n_tasks = (fb->height + task_rows - 1) / task_rows;
    const int mine = n_tasks > rank ? (n_tasks - rank + world - 1) / world : 0;
    p.begin = 0; p.task_rows = task_rows; p.rank = rank; p.world = world;
    const int64_t cnt = world == 1 ? (int64_t)fb->width * fb->height : (int64_t)mine * task_rows * fb->width;
    if (cnt > INT32_MAX) return fail(RTW_ERR_LIMIT, "too many work items");
    p.count = (int)cnt;
    out->p = p;
    // paths per pass: the busy tiles' pixels x sub-samples (known once the tile lists exist; the whole share until then)
    long long per_pass = (long long)p.count * sub_samples;
    if (p.count > 0) {
        RtwRenderParams t = p; t.width = fb->width; t.height = fb->height;
        rtw_scene::BinSet* bs = nullptr; const rtw_scene::GroupTable* gt = nullptr;
        if (!cx->stats_enabled && choose_group_tiles(t, 0, fb->width * fb->height - 1) && scene_bins(scene, t.width, t.height, t.tile_w, t.tile_h, &bs) == RTW_OK &&
            scene_group_table(scene, *bs, t, sub_samples, &gt) == RTW_OK && gt) per_pass = (long long)gt->n_busy * 64 * sub_samples;
    }
    out->per_pass = per_pass;
    return RTW_OK;
}

// One group: passes first_pass .. first_pass + n_passes - 1 over the pixels begin .. end of this rank's share of the frame.
// ranged: a pixel range (rtw_render_range), else the whole frame / the rank's tasks.  batch_pos as rtw_context::batch_pos.
static int render_group(rtw_scene* scene, rtw_framebuffer* fb, RtwRenderParams p, bool ranged, int range_begin, int range_end,
                        int max_bounce, int use_base_color, int first_pass, int n_passes, int sub_samples, uint32_t seed)
{
    rtw_context* cx = scene->ctx;
    p.width = fb->width; p.height = fb->height;
    p.max_bounce = max_bounce; p.preview = use_base_color ? 1 : 0; p.pass_index = first_pass; p.sub_samples = sub_samples; p.seed = seed;
    if (!choose_group_tiles(p, range_begin, range_end)) return fail(RTW_ERR_LIMIT, "frame too large for the tile mapping");
    rtw_scene::BinSet* bs = nullptr;
    int rc = scene_bins(scene, p.width, p.height, p.tile_w, p.tile_h, &bs); if (rc != RTW_OK) return rc;
    p.bins = bs->d_bins; p.cam_dx = bs->d_dx; p.cam_dy = bs->d_dy;
    RtwGroupParams g; std::memset(&g, 0, sizeof g);
    g.first_pass = first_pass; g.n_passes = n_passes;
    g.range_begin = range_begin; g.range_end = range_end; g.first_tile = 0;
    if (cx->stats_enabled) ranged = true;       // work counters: every camera ray goes through the primary kernel, which counts (the sky kernel does not)
    if (ranged) {           // every tile of the range is a job; the primary kernel finds out which see the sky only
        g.n_busy = p.count >> 6; g.n_sky = 0; g.n_jobs = g.n_busy;
        g.busy_tiles = nullptr; g.sky_tiles = nullptr; g.jobs = nullptr;
    } else {
        const rtw_scene::GroupTable* gt = nullptr;
        rc = scene_group_table(scene, *bs, p, sub_samples, &gt); if (rc != RTW_OK) return rc;
        g.n_busy = gt->n_busy; g.n_sky = gt->n_sky; g.n_jobs = gt->n_jobs;
        g.busy_tiles = gt->d_busy; g.sky_tiles = gt->d_sky; g.jobs = gt->d_jobs;
    }
    {   // the scene's leading spheres / planes / capsules / triangles are tested by the lane that sets a segment up (group_lead_query)
        int lead = 0;
        while (lead < (int)scene->meshes.size() && scene->meshes[(size_t)lead]->kind != RTW_SHAPE_MESH) lead++;
        p.lead_shapes = lead;
    }
    g.rp = p;
    // the primary kernel's waves: the bin of a tile is walked once for up to four rays per lane (its sub-samples and / or several passes) -- as many
    // passes per wave as still leave the launch about two rounds of waves (C2 at 20 passes: 2 per wave 0.0367, 4 per wave 0.0376 ms per pass; at 192: 4 wins)
    g.primary_passes = cx->primary_passes < 0 ? -1 : 1;        // -1: a ray set at a time (the round-2 kernel's order; kept for comparison)
    if (cx->primary_passes >= 0 && !scene->has_analytic && scene->meshes.size() == 1) {
        const int fit = 4 / sub_samples;
        int ppw = cx->primary_passes > 0 ? (cx->primary_passes < fit ? cx->primary_passes : fit) : fit;
        if (cx->primary_passes == 0)
            while (ppw > 1 && (long long)g.n_jobs * ((n_passes + ppw - 1) / ppw) < (long long)cx->cu_count * 24) ppw >>= 1;
        g.primary_passes = ppw < 1 ? 1 : ppw;
    }
    const bool carry = scene->texture_carry;
    const size_t capacity = rtw::group_capacity((size_t)g.n_busy * 64 * (size_t)sub_samples, n_passes);
    if (capacity >= ((size_t)1 << 31)) return fail(RTW_ERR_LIMIT, "too many paths in one launch");
    // the workspace of THIS group (grow-only): a caller that must not stall inside a later, longer call reserves it with rtw_render_reserve
    cx->ws_single = n_passes == 1 && cx->lane_count == 1;
    rc = ensure_group_workspace(cx, rtw::group_workspace_bytes(capacity, max_bounce, carry, nullptr));
    cx->ws_single = false;
    if (rc != RTW_OK) return rc;
    rtw::GroupTuning tune;
    tune.capacity = capacity;
    tune.aux_stream = cx->aux_stream; tune.fork_event = cx->fork_event; tune.join_event = cx->join_event;
    tune.do_fork = cx->batch_pos == 0 || cx->batch_pos == 1 || !cx->aux_unjoined;
    tune.do_join = cx->batch_pos == 0 || cx->batch_pos == 3; tune.aux_unjoined = &cx->aux_unjoined;
    tune.gamma_thr = cx->d_gamma;
    tune.has_analytic = scene->has_analytic; tune.carry = carry;
    tune.counters_clean = cx->group_clean[cx->lane];
    if (cx->lane_count > 1) {       // a part of a split group: no sky tiles (the group's sky kernel was enqueued ahead of the parts, see rtw_render_passes); its resolve kernel after the previous part's
        tune.sky_passes = cx->lane_sky_passes; tune.sky_first_pass = cx->lane_sky_first;
        tune.sky_mode = cx->lane_sky_only ? 2 : 1;
        tune.aux_stream = nullptr;
        tune.resolve_after = cx->lane > 0 ? cx->part_resolved[cx->lane - 1] : nullptr;
        tune.resolve_done = cx->part_resolved[cx->lane];
    }
    tune.cu_count = cx->cu_count; tune.sky_blocks = cx->sky_blocks;
    {      // the first mesh's upper tree levels live in the trace blocks' LDS
        for (size_t k = 0; k < scene->meshes.size(); k++)
            if (scene->meshes[k]->kind == RTW_SHAPE_MESH && scene->meshes[k]->tnodes_top > 0) { tune.staged_shape = (int)k; tune.staged_top = scene->meshes[k]->tnodes_top; tune.staged_all = tune.staged_top == (int)scene->meshes[k]->tnodes.size(); break; }
    }
    if (tune.staged_shape >= 0 && tune.staged_all) {        // ... and the triangles' planes beside a tree that lives in LDS entirely, if they fit (1 024-thread block: 32 KiB of candidate lists)
        const size_t need = (size_t)RTW_PERSIST_CAND_BYTES + (size_t)tune.staged_top * 32 + scene->meshes[(size_t)tune.staged_shape]->tris.size() * 16;
        tune.staged_planes = cx->backface_filter != 0 && need <= (size_t)150 * 1024;
        tune.staged_tris = (int)scene->meshes[(size_t)tune.staged_shape]->tris.size();
    }
    tune.single_mesh = scene->meshes.size() == 1 && scene->meshes[0]->kind == RTW_SHAPE_MESH && !scene->meshes[0]->nodes.empty();
    tune.lead_mesh = !carry && p.lead_shapes > 0 && p.lead_shapes == (int)scene->meshes.size() - 1 && scene->meshes.back()->kind == RTW_SHAPE_MESH && !scene->meshes.back()->nodes.empty() &&
                     tune.staged_shape == p.lead_shapes;

    // big trees: rays with very long walks (a few per cent need 4 x the mean) go to the wave-per-ray kernel instead of keeping a launch waiting
    tune.visit_budget = (cx->visit_budget > 0 && (tune.single_mesh || tune.lead_mesh) && scene->meshes.back()->nodes.size() > (size_t)cx->budget_nodes) ? cx->visit_budget : INT32_MAX;
    {   // long walks (a tree of more than 4 096 nodes): the wave-per-ray kernel pays up to longer lists
        bool big = false;
        for (const auto& m : scene->meshes) if (m->nodes.size() > 4096) big = true;
        tune.wave_below = big ? cx->wave_below * 5 : cx->wave_below;
    }
    tune.timing = (cx->kernel_timing && cx->lane == 0) ? cx->timing_events : nullptr;
    // list lengths of the latest finished group with the same shape (a stale or missing value only costs speed)
    // (keyed by the launch shape WITHOUT the number of passes: the lengths are kept per pass and scaled to the group at hand, so a warm-up call of any
    // length primes a longer one)
    rtw_context::GroupKey key; key.scene = scene; key.capacity = (long long)((size_t)g.n_busy * 64 * (size_t)sub_samples); key.max_bounce = max_bounce; key.preview = p.preview; key.n_passes = 0;
    if (cx->gcounters_pending && hipEventQuery(cx->gcounters_event) == hipSuccess) {
        cx->known_gkey = cx->gcounters_key; cx->gcounters_pending = false;
        const int np = cx->gcounters_passes > 0 ? cx->gcounters_passes : 1;
        for (int r = 0; r < 32; r++) cx->known_ground[r] = (int)((cx->h_gcounters[r] + (uint32_t)np - 1u) / (uint32_t)np);          // per pass, rounded up
        for (int r = 0; r < 16; r++) cx->known_goverflow[r] = (int)((cx->h_gcounters[40 + r] + (uint32_t)np - 1u) / (uint32_t)np);
        for (int r = 0; r < 16; r++) cx->known_gtrace[r] = (int)((cx->h_gcounters[24 + r] + (uint32_t)np - 1u) / (uint32_t)np);
    }
    auto scaled = [&](int per_pass) { const long long v = (long long)per_pass * n_passes; return (int)(v > INT32_MAX ? INT32_MAX : v); };
    for (int r = 0; r < 32; r++) tune.round_hint[r] = (cx->known_gkey == key) ? scaled(cx->known_ground[r]) : -1;
    for (int r = 0; r < 32; r++) { tune.overflow_hint[r] = -1; tune.trace_hint[r] = -1; }
    for (int r = 0; r < 16; r++) tune.overflow_hint[r] = (cx->known_gkey == key) ? scaled(cx->known_goverflow[r]) : -1;
    for (int r = 0; r < 16; r++) tune.trace_hint[r] = (cx->known_gkey == key) ? scaled(cx->known_gtrace[r]) : -1;
    tune.skip_trace = p.lead_shapes > 0 && p.lead_shapes == (int)scene->meshes.size();
    bool& clean = cx->group_clean[cx->lane];
    const bool sky_only = tune.sky_mode == 2;       // (touches neither the workspace nor its counters)
    if (!sky_only) clean = false;
    const hipError_t e = (hipError_t)rtw::launch_render_group(scene->d_scene, fb->accum, fb->argb, cx->d_group_ws[cx->lane], g, tune, cx->stats_enabled,
                                                              cx->lane ? cx->part_stream[cx->lane] : cx->stream);
    if (e != hipSuccess) return hip_fail(e, "group launch");
    if (sky_only) return RTW_OK;
    clean = true;
    cx->last_pipeline = 4;
    cx->last_group_passes = cx->lane_count > 1 ? cx->lane_sky_passes : n_passes;
    if (cx->lane == 0 && !cx->gcounters_pending && g.n_busy > 0 && (!(cx->known_gkey == key) || (cx->hint_tick++ % cx->hint_period) == 0)) {
        if (hipMemcpyAsync(cx->h_gcounters, (char*)cx->d_group_ws[0] + 256, 256, hipMemcpyDeviceToHost, cx->stream) == hipSuccess &&
            hipEventRecord(cx->gcounters_event, cx->stream) == hipSuccess) {
            cx->gcounters_pending = true; cx->gcounters_key = key; cx->gcounters_passes = n_passes;
        }
    }
    return RTW_OK;
}

// ---- the hot path -------------------------------------------------------------------------------------------
// One pass through pipeline 3 (bins + a wave per secondary ray; the one-pass reference of the pass-batched pipeline) or pipeline 0 (one kernel, one
// thread per pixel: every frame shape and range, the reference-order walk for the work counters).
static int render_common(rtw_scene* scene, rtw_framebuffer* fb, RtwRenderParams& p, int max_bounce, int use_base_color,
                         int pass_index, int sub_samples, uint32_t seed)
{
    int rc = check_render_args(scene, fb, max_bounce, pass_index, sub_samples); if (rc != RTW_OK) return rc;
    p.width = fb->width; p.height = fb->height;
    p.max_bounce = max_bounce; p.preview = use_base_color ? 1 : 0; p.pass_index = pass_index; p.sub_samples = sub_samples; p.seed = seed;
    rtw_context* cx = scene->ctx;
    hipError_t e;
    // the bins + wave pipeline needs whole rows, the flat hierarchy (traversal != 0) and a frame that tiles; a scene whose analytic hits can inherit a texel
    // (one RayHitResult serves all shapes) goes through the single kernel; so does everything else that does not fit
    int pipeline = cx->pipeline >= 3 ? 3 : 0;
    if (scene->texture_carry) pipeline = 0;
    int sky_job0 = 0;
    if (pipeline == 3) {
        RtwRenderParams tiled = p;
        rtw_scene::BinSet* bs = nullptr;
        if (scene->traversal != 0 && choose_tiles(tiled) && scene_bins(scene, p.width, p.height, tiled.tile_w, tiled.tile_h, &bs) == RTW_OK) {
            p = tiled; p.bins = bs->d_bins; p.cam_dx = bs->d_dx; p.cam_dy = bs->d_dy;
            const rtw_scene::JobTable* jt = nullptr;
            if (scene_job_table(scene, *bs, p, sub_samples, &jt) != RTW_OK) jt = nullptr;
            p.tile_order = jt ? jt->d_order : nullptr;
            p.n_jobs = jt ? jt->n_jobs : 0;
            sky_job0 = jt ? jt->n_busy : 0;
        } else {
            pipeline = 0;
        }
    }
    if (pipeline == 3) {
        rtw::PipelineLayout layout;
        rc = ensure_workspace(cx, rtw::pipeline_workspace_bytes(p.count, max_bounce, &layout)); if (rc != RTW_OK) return rc;
        rtw::PipelineTuning tune;
        // queue length of the previous pass with the same launch shape (a stale or missing value only costs speed)
        const long long shape = (long long)p.count * 64 + sub_samples;
        if (cx->counters_pending && hipEventQuery(cx->counters_event) == hipSuccess) {
            cx->known_paths = (int)cx->h_counters[0]; cx->known_shape = cx->counters_shape; cx->counters_pending = false;
            for (int r = 0; r < 32; r++) cx->known_rounds[r] = (int)cx->h_counters[4 + r];
        }
        tune.expected_paths = (cx->known_shape == shape) ? cx->known_paths : -1;
        for (int r = 0; r < 32; r++) tune.round_hint[r] = (cx->known_shape == shape) ? cx->known_rounds[r] : -1;
        tune.aux_stream = cx->aux_stream; tune.fork_event = cx->fork_event; tune.join_event = cx->join_event;
        tune.sky_job0 = sky_job0; tune.gamma_thr = cx->d_gamma;
        tune.do_fork = cx->batch_pos == 0 || cx->batch_pos == 1 || !cx->aux_unjoined;      // (a run whose earlier passes launched no sky kernel has not forked yet)
        tune.do_join = cx->batch_pos == 0 || cx->batch_pos == 3; tune.aux_unjoined = &cx->aux_unjoined;
        tune.cu_count = cx->cu_count;
        tune.timing = cx->kernel_timing ? cx->timing_events : nullptr;
        {   // leading spheres / planes / capsules are tested by the lane that sets a segment up (see RtwRenderParams::lead_shapes)
            int lead = 0;
            while (lead < (int)scene->meshes.size() && scene->meshes[(size_t)lead]->kind != RTW_SHAPE_MESH) lead++;
            p.lead_shapes = lead;
            tune.has_analytic = scene->has_analytic;
            tune.skip_trace = lead > 0 && lead == (int)scene->meshes.size();       // nothing is left for the trace kernels
        }
        const size_t coff = layout.counters_off;
        tune.counters_clean = cx->clean_ws == cx->d_workspace && cx->clean_off == coff && cx->d_workspace != nullptr;
        cx->clean_ws = nullptr;
        e = (hipError_t)rtw::launch_render_pipeline(scene->d_scene, fb->accum, fb->argb, cx->d_workspace, p, tune, cx->stats_enabled, cx->stream);
        if (e == hipSuccess) { cx->clean_ws = cx->d_workspace; cx->clean_off = coff; }
        if (e == hipSuccess && !cx->counters_pending && (cx->known_shape != shape || (cx->hint_tick++ % cx->hint_period) == 0)) {     // one copy in flight at a time; its value is used once it has landed
            if (hipMemcpyAsync(cx->h_counters, (char*)cx->d_workspace + coff + 256, 256, hipMemcpyDeviceToHost, cx->stream) == hipSuccess &&       // (the pass files its counters 64 words further)
                hipEventRecord(cx->counters_event, cx->stream) == hipSuccess) {
                cx->counters_pending = true; cx->counters_shape = shape;
            }
        }
    } else {
        rc = ensure_workspace(cx, rtw::level_workspace_bytes(p.count, max_bounce)); if (rc != RTW_OK) return rc;
        cx->clean_ws = nullptr;
        e = (hipError_t)rtw::launch_render(scene->d_scene, fb->accum, fb->argb, cx->d_workspace, p, cx->stats_enabled, cx->stream);
    }
    if (e != hipSuccess) return hip_fail(e, "render_kernel launch");
    cx->last_pipeline = pipeline;
    return RTW_OK;
}

int rtw_render_range(rtw_scene* scene, rtw_framebuffer* fb, int begin, int end, int max_bounce, int use_base_color,
                     int pass_index, int sub_samples, uint32_t seed)
{
    int rc = need_committed(scene); if (rc != RTW_OK) return rc;
    if (!fb) return fail(RTW_ERR_INVALID, "framebuffer is null");
    const int npix = fb->width * fb->height;
    if (begin < 0 || end >= npix) return fail(RTW_ERR_INVALID, "pixel range outside the framebuffer");
    RtwRenderParams p; std::memset(&p, 0, sizeof p);
    p.begin = begin; p.count = end >= begin ? end - begin + 1 : 0;      // an empty range renders nothing, like the reference loop
    p.task_rows = 0; p.rank = 0; p.world = 1;
    if (group_pipeline_ok(scene)) {
        rc = check_render_args(scene, fb, max_bounce, pass_index, sub_samples); if (rc != RTW_OK) return rc;
        if (p.count == 0) return RTW_OK;
        return render_group(scene, fb, p, !(begin == 0 && end == npix - 1), begin, end, max_bounce, use_base_color, pass_index, 1, sub_samples, seed);
    }
    return render_common(scene, fb, p, max_bounce, use_base_color, pass_index, sub_samples, seed);
}

int rtw_render_tasks(rtw_scene* scene, rtw_framebuffer* fb, int task_rows, int rank, int world, int max_bounce, int use_base_color,
                     int pass_index, int sub_samples, uint32_t seed)
{
    int rc = need_committed(scene); if (rc != RTW_OK) return rc;
    if (!fb) return fail(RTW_ERR_INVALID, "framebuffer is null");
    if (task_rows < 1 || world < 1 || rank < 0 || rank >= world) return fail(RTW_ERR_INVALID, "bad task partition");
    RtwRenderParams p; std::memset(&p, 0, sizeof p);
    const int n_tasks = (fb->height + task_rows - 1) / task_rows;
    const int mine = n_tasks > rank ? (n_tasks - rank + world - 1) / world : 0;
    p.begin = 0; p.task_rows = task_rows; p.rank = rank;
    p.world = world;
    if (world == 1) { p.count = fb->width * fb->height; }
    else {
        const int64_t cnt = (int64_t)mine * task_rows * fb->width;
        if (cnt > INT32_MAX) return fail(RTW_ERR_LIMIT, "too many work items");
        p.count = (int)cnt;
    }
    if (group_pipeline_ok(scene)) {
        rc = check_render_args(scene, fb, max_bounce, pass_index, sub_samples); if (rc != RTW_OK) return rc;
        if (p.count == 0) return RTW_OK;
        return render_group(scene, fb, p, false, 0, fb->width * fb->height - 1, max_bounce, use_base_color, pass_index, 1, sub_samples, seed);
    }
    return render_common(scene, fb, p, max_bounce, use_base_color, pass_index, sub_samples, seed);
}

// n_passes accumulated passes (UpdateBitmapPixels' sample loop, Src/RayTracerProgram.cpp:317-361) over this rank's tasks.
int rtw_render_passes(rtw_scene* scene, rtw_framebuffer* fb, int task_rows, int rank, int world, int max_bounce, int use_base_color,
                      int first_pass, int n_passes, int sub_samples, uint32_t seed)
{
    int rc = need_committed(scene); if (rc != RTW_OK) return rc;
    if (!fb) return fail(RTW_ERR_INVALID, "framebuffer is null");
    if (n_passes < 0 || first_pass < 0) return fail(RTW_ERR_INVALID, "bad pass range");
    rtw_context* cx = scene->ctx;
    if (group_pipeline_ok(scene)) {
        // groups of passes share one set of launches (rtw_group_kernels.h); the groups of this call fork the second stream once and join it once
        GroupCall gc;
        rc = plan_group_call(scene, fb, task_rows, rank, world, max_bounce, first_pass, sub_samples, &gc); if (rc != RTW_OK) return rc;
        if (gc.p.count == 0) return RTW_OK;
        const RtwRenderParams& p = gc.p;
        const long long per_pass = gc.per_pass;
        const int last_pixel = fb->width * fb->height - 1;
        cx->group_cap = 0;
        int done = 0;
        while (done < n_passes) {
            int parts = 1;
            const int k = next_group(cx, per_pass, n_passes - done, max_bounce, scene->texture_carry, use_base_color != 0, &parts);
            const bool split = parts > 1;
            rc = RTW_OK;        // (a refused group of the previous round has been re-formed)
            const bool first = done == 0, last = done + k >= n_passes;
            cx->batch_pos = (first && last) ? 0 : (first ? 1 : (last ? 3 : 2));
            if (split) {
                // The group as `parts` parts on as many streams.  Its kernels are bound by latency, not by a throughput roof: every stage of a
                // part's chain of launches lasts about as long as its longest ray, whatever the number of rays, so several shorter chains side by side use
                // the chip where one long chain leaves it idle.  Order kept: a part's share of the sky tiles gets all k passes of its pixels in a row
                // (first kernel of the part's stream); a part's resolve kernel waits for the previous part's, so a busy tile's passes are added in pass order.
                const int base = k / parts, rem = k % parts;
                {   // every part's workspace before anything of the group is launched: a refusal then leaves nothing half done
                    const size_t need = rtw::group_workspace_bytes(rtw::group_capacity((size_t)per_pass, base + (rem ? 1 : 0)), max_bounce, scene->texture_carry, nullptr);
                    for (int j = 0; j < parts && rc == RTW_OK; j++) { cx->lane = j; rc = ensure_group_workspace(cx, need); }
                    cx->lane = 0;
                }
                if (rc == RTW_OK) {
                    (void)hipEventRecord(cx->split_fork, cx->stream);
                    for (int j = 1; j < parts; j++) (void)hipStreamWaitEvent(cx->part_stream[j], cx->split_fork, 0);
                    int off = 0;
                    cx->lane_count = parts; cx->lane_sky_passes = k; cx->lane_sky_first = first_pass + done;
                    // the sky tiles of the whole group (all k passes of their pixels in a row) FIRST, on the LAST part's stream: that part's chain is enqueued
                    // last anyway (about 100 us of host time after the first part's), so the sky kernel runs where nothing else would yet.  (Tried: the sky
                    // kernel on a third, lowest-priority stream -- the last part's first kernel then still starts when the sky kernel ends (the two queues
                    // share a dispatch pipe), C2 +4 %; every part's first kernel enqueued before any part's remaining launches: no change.)
                    cx->lane = parts - 1; cx->lane_sky_only = true;
                    rc = render_group(scene, fb, p, false, 0, last_pixel, max_bounce, use_base_color, first_pass + done, base, sub_samples, seed);
                    cx->lane_sky_only = false;
                    for (int j = 0; j < parts && rc == RTW_OK; j++) {
                        const int kj = base + (j < rem ? 1 : 0);
                        cx->lane = j;
                        rc = render_group(scene, fb, p, false, 0, last_pixel, max_bounce, use_base_color, first_pass + done + off, kj, sub_samples, seed);
                        off += kj;
                    }
                    cx->lane = 0; cx->lane_count = 1; cx->lane_sky_passes = 0;
                    for (int j = 1; j < parts; j++) {
                        (void)hipEventRecord(cx->part_done[j], cx->part_stream[j]);
                        (void)hipStreamWaitEvent(cx->stream, cx->part_done[j], 0);
                    }

                }
            } else
                rc = render_group(scene, fb, p, false, 0, last_pixel, max_bounce, use_base_color, first_pass + done, k, sub_samples, seed);
            cx->batch_pos = 0;
            if (rc != RTW_OK && cx->ws_refused && k > 1) {
                // no room for this group's workspace (the context's limit, or the device is full): nothing of the group has been launched -- form smaller
                // groups for the rest of the call instead of failing (k = 1 needs ~300-600 bytes per path of one pass)
                cx->ws_refused = false;
                cx->group_cap = split ? (k + parts - 1) / parts : k / 2;
                cx->fallbacks++;
                continue;
            }
            if (cx->aux_unjoined && (rc != RTW_OK || last)) {     // the run ends here (an error, or a last group that launched no sky kernel)
                (void)hipEventRecord(cx->join_event, cx->aux_stream);
                (void)hipStreamWaitEvent(cx->stream, cx->join_event, 0);
                cx->aux_unjoined = false;
            }
            if (rc != RTW_OK) { cx->group_cap = 0; return rc; }
            done += k;
        }
        cx->group_cap = 0;
        return RTW_OK;
    }
    // pipelines 3 / 0: a pass per set of launches; the passes of this call form a run (the second stream is forked before the first and joined after the last)
    for (int i = 0; i < n_passes; i++) {
        cx->batch_pos = n_passes == 1 ? 0 : (i + 1 == n_passes ? 3 : (i == 0 ? 1 : 2));
        rc = rtw_render_tasks(scene, fb, task_rows, rank, world, max_bounce, use_base_color, first_pass + i, sub_samples, seed);
        cx->batch_pos = 0;
        if (cx->aux_unjoined && (rc != RTW_OK || i + 1 == n_passes)) {      // the run ends here (an error, or a last pass that launched no sky kernel)
            (void)hipEventRecord(cx->join_event, cx->aux_stream);
            (void)hipStreamWaitEvent(cx->stream, cx->join_event, 0);
            cx->aux_unjoined = false;
        }
        if (rc != RTW_OK) return rc;
    }
    return RTW_OK;
}

// Everything a later rtw_render_passes call with these arguments needs that would otherwise be made inside it: the screen bins and tile tables of this
// frame shape and the group workspace(s) of the largest group that call will form.  Renders nothing.
int rtw_render_reserve(rtw_scene* scene, rtw_framebuffer* fb, int task_rows, int rank, int world, int max_bounce, int n_passes, int sub_samples)
{
    int rc = need_committed(scene); if (rc != RTW_OK) return rc;
    if (!fb) return fail(RTW_ERR_INVALID, "framebuffer is null");
    if (n_passes < 0) return fail(RTW_ERR_INVALID, "bad pass count");
    rtw_context* cx = scene->ctx;
    if (!group_pipeline_ok(scene) || n_passes == 0) return RTW_OK;      // the older pipelines size their workspace by the frame, on first use
    GroupCall gc;
    rc = plan_group_call(scene, fb, task_rows, rank, world, max_bounce, 0, sub_samples, &gc); if (rc != RTW_OK) return rc;
    if (gc.p.count == 0) return RTW_OK;
    RtwRenderParams t = gc.p; t.width = fb->width; t.height = fb->height;
    if (!choose_group_tiles(t, 0, fb->width * fb->height - 1)) return fail(RTW_ERR_LIMIT, "frame too large for the tile mapping");
    size_t need[RTW_MAX_PARTS] = { 0, 0, 0, 0 };
    const int saved_cap = cx->group_cap;
    cx->group_cap = 0;
    for (int done = 0; done < n_passes;) {
        int parts = 1;
        const int k = next_group(cx, gc.per_pass, n_passes - done, max_bounce, scene->texture_carry, false, &parts);
        const int kk = (k + parts - 1) / parts;
        const size_t b = rtw::group_workspace_bytes(rtw::group_capacity((size_t)gc.per_pass, kk), max_bounce, scene->texture_carry, nullptr);
        for (int j = 0; j < parts; j++) if (b > need[j]) need[j] = b;
        done += k;
    }
    cx->group_cap = saved_cap;
    for (int j = 0; j < RTW_MAX_PARTS && rc == RTW_OK; j++) if (need[j] > 0) {
        cx->lane = j; rc = ensure_group_workspace(cx, need[j]);
        // ... and the part's list counters zeroed on the part's own stream: its first group needs no memset, and a stream's first operation (the runtime
        // sets its hardware queue up then) does not fall into a timed call
        if (rc == RTW_OK && !cx->group_clean[j]) {
            hipStream_t ps = j ? cx->part_stream[j] : cx->stream;
            if (hipMemsetAsync(cx->d_group_ws[j], 0, 256, ps) == hipSuccess && hipStreamSynchronize(ps) == hipSuccess) cx->group_clean[j] = true;
        }
    }
    cx->lane = 0;
    if (rc != RTW_OK && cx->ws_refused) { cx->ws_refused = false; return RTW_OK; }      // not an error: the call will form smaller groups
    return rc;
}

// device memory this context holds right now: the group workspaces, the older pipelines' workspace and its share of the unit-vector table
// (201 MB per device, shared by the contexts on it); scenes and framebuffers are the caller's objects and not counted
long long rtw_context_memory_bytes(const rtw_context* ctx)
{
    if (!ctx) return 0;
    size_t g = 0; for (int j = 0; j < RTW_MAX_PARTS; j++) g += ctx->group_ws_bytes[j];
    return (long long)(g + ctx->workspace_bytes + (size_t)RTW_TABLE_SIZE * 3 * sizeof(float) + 2 * 1024 + 64);
}
// ... of which workspace (grows with the largest group rendered so far; rtw_context_trim gives it back)
long long rtw_context_workspace_bytes(const rtw_context* ctx)
{
    if (!ctx) return 0;
    size_t g = 0; for (int j = 0; j < RTW_MAX_PARTS; j++) g += ctx->group_ws_bytes[j];
    return (long long)(g + ctx->workspace_bytes);
}
int rtw_context_trim(rtw_context* ctx)
{
    if (!ctx) return fail(RTW_ERR_INVALID, "context is null");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->aux_stream) HIP_TRY(hipStreamSynchronize(ctx->aux_stream));
    for (int j = 0; j < RTW_MAX_PARTS; j++) {
        if (ctx->part_stream[j]) HIP_TRY(hipStreamSynchronize(ctx->part_stream[j]));
        if (ctx->d_group_ws[j]) { (void)hipFree(ctx->d_group_ws[j]); ctx->d_group_ws[j] = nullptr; ctx->group_ws_bytes[j] = 0; ctx->group_clean[j] = false; }
    }
    if (ctx->d_workspace) { (void)hipFree(ctx->d_workspace); ctx->d_workspace = nullptr; ctx->workspace_bytes = 0; ctx->clean_ws = nullptr; }
    return RTW_OK;
}
int rtw_context_fallbacks(const rtw_context* ctx) { return ctx ? ctx->fallbacks : 0; }

// ---- multi-GPU gather over RCCL -------------------------------------------------------------------------------------------------
}  // extern "C"
#include <dlfcn.h>
#include <rccl/rccl.h>      // types only: the library is loaded with dlopen on first use
namespace {
struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};
std::mutex g_rccl_mutex;
RcclApi g_rccl;
const RcclApi* rccl_api()
{
    std::lock_guard<std::mutex> l(g_rccl_mutex);
    if (g_rccl.handle) return &g_rccl;
    // the copy the process already holds (PyTorch ships its own librccl.so), else the ROCm one
    const char* names[] = { "librccl.so.1", "librccl.so" };
    void* h = nullptr;
    if (const char* named = std::getenv("RTW_RCCL_LIBRARY")) {      // a deployment's own RCCL build (include/rtwin.h)
        h = dlopen(named, RTLD_NOW | RTLD_GLOBAL);
        if (!h) { const char* e = dlerror(); g_rccl.error = std::string("RTW_RCCL_LIBRARY: ") + (e ? e : "cannot load"); return nullptr; }       // (dlerror() clears the message: one call)
    }
    for (const char* n : names) if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    for (const char* n : names) if (!h) h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) { const char* e = dlerror(); g_rccl.error = std::string("cannot load librccl: ") + (e ? e : "?"); return nullptr; }
    RcclApi a; a.handle = h;
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(h, "ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
    a.GroupStart = (decltype(a.GroupStart))dlsym(h, "ncclGroupStart");
    a.GroupEnd = (decltype(a.GroupEnd))dlsym(h, "ncclGroupEnd");
    a.Send = (decltype(a.Send))dlsym(h, "ncclSend");
    a.Recv = (decltype(a.Recv))dlsym(h, "ncclRecv");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.GroupStart || !a.GroupEnd || !a.Send || !a.Recv) {
        g_rccl.error = "librccl lacks a needed symbol"; return nullptr;
    }
    g_rccl = a;
    return &g_rccl;
}
int rccl_fail(const RcclApi* a, ncclResult_t r, const char* what)
{
    return fail(RTW_ERR_HIP, std::string(what) + ": " + ((a && a->GetErrorString) ? a->GetErrorString(r) : "rccl error"));
}
}  // namespace

struct rtw_comm {
    rtw_context* ctx = nullptr;
    ncclComm_t comm = nullptr;
    bool owned = false;
    int rank = 0, world = 1;
    void* d_stage = nullptr;            // compact row blocks of rtw_gather_rows (this rank's on a sender, every peer's on the root); grow-only
    size_t stage_bytes = 0;
    long long messages = 0;             // ncclSend / ncclRecv operations issued so far (rtw_comm_messages)
};

extern "C" {

int rtw_comm_unique_id(uint8_t id[RTW_COMM_ID_BYTES])
{
    static_assert(sizeof(ncclUniqueId) == RTW_COMM_ID_BYTES, "ncclUniqueId size");
    if (!id) return fail(RTW_ERR_INVALID, "null argument");
    const RcclApi* a = rccl_api();
    if (!a) return fail(RTW_ERR_STATE, g_rccl.error);
    ncclUniqueId u;
    const ncclResult_t r = a->GetUniqueId(&u);
    if (r != ncclSuccess) return rccl_fail(a, r, "ncclGetUniqueId");
    std::memcpy(id, &u, RTW_COMM_ID_BYTES);
    return RTW_OK;
}

int rtw_comm_create(rtw_context* ctx, const uint8_t id[RTW_COMM_ID_BYTES], int rank, int world, rtw_comm** out)
{
    if (!ctx || !id || !out || world < 1 || rank < 0 || rank >= world) return fail(RTW_ERR_INVALID, "bad argument");
    *out = nullptr;
    const RcclApi* a = rccl_api();
    if (!a) return fail(RTW_ERR_STATE, g_rccl.error);
    HIP_TRY(hipSetDevice(ctx->device));
    ncclUniqueId u; std::memcpy(&u, id, RTW_COMM_ID_BYTES);
    ncclComm_t c = nullptr;
    const ncclResult_t r = a->CommInitRank(&c, world, u, rank);
    if (r != ncclSuccess) return rccl_fail(a, r, "ncclCommInitRank");
    rtw_comm* k = new rtw_comm(); k->ctx = ctx; k->comm = c; k->owned = true; k->rank = rank; k->world = world;
    *out = k;
    return RTW_OK;
}

int rtw_comm_wrap(rtw_context* ctx, void* nccl_comm, int rank, int world, rtw_comm** out)
{
    if (!ctx || !nccl_comm || !out || world < 1 || rank < 0 || rank >= world) return fail(RTW_ERR_INVALID, "bad argument");
    if (!rccl_api()) return fail(RTW_ERR_STATE, g_rccl.error);
    rtw_comm* k = new rtw_comm(); k->ctx = ctx; k->comm = (ncclComm_t)nccl_comm; k->owned = false; k->rank = rank; k->world = world;
    *out = k;
    return RTW_OK;
}

int rtw_comm_destroy(rtw_comm* comm)
{
    if (!comm) return RTW_OK;
    if (comm->owned && comm->comm) {
        (void)hipSetDevice(comm->ctx->device);
        (void)hipStreamSynchronize(comm->ctx->stream);
        const RcclApi* a = rccl_api();
        if (a) (void)a->CommDestroy(comm->comm);
    }
    if (comm->d_stage) { (void)hipSetDevice(comm->ctx->device); (void)hipStreamSynchronize(comm->ctx->stream); (void)hipFree(comm->d_stage); }
    delete comm;
    return RTW_OK;
}

int rtw_gather_rows(rtw_comm* comm, rtw_framebuffer* fb, int task_rows, int mode)
{
    if (!comm || !fb) return fail(RTW_ERR_INVALID, "null argument");
    if (fb->ctx != comm->ctx) return fail(RTW_ERR_INVALID, "communicator and framebuffer belong to different contexts");
    if (task_rows < 1 || (mode != RTW_GATHER_ALL && mode != RTW_GATHER_ARGB)) return fail(RTW_ERR_INVALID, "bad argument");
    if (comm->world == 1) return RTW_OK;
    if (comm->world > 65) return fail(RTW_ERR_LIMIT, "rtw_gather_rows: at most 65 ranks");
    const RcclApi* a = rccl_api();
    if (!a) return fail(RTW_ERR_STATE, g_rccl.error);
    HIP_TRY(hipSetDevice(comm->ctx->device));
    hipStream_t st = comm->ctx->stream;
    const int W = fb->width, H = fb->height, world = comm->world;
    if ((int64_t)task_rows * W > INT32_MAX) return fail(RTW_ERR_LIMIT, "task too large");
    const int n_tasks = (H + task_rows - 1) / task_rows;
    const bool with_accum = mode == RTW_GATHER_ALL;
    // ONE message per peer: a sender packs its task rows (scattered over the frame) into a compact block, the root receives every peer's block
    // into its staging buffer inside one RCCL group and unpacks them all with one launch.  (Round 2 sent every 10-row task on its own: ~100 small
    // messages per gather at 1080p, whose per-message cost was of the order of the whole timed region.)
    auto pixels_of = [&](int r) -> size_t {
        if (n_tasks <= r) return 0;
        const int mine = (n_tasks - r + world - 1) / world;
        size_t px = (size_t)mine * (size_t)task_rows * (size_t)W;
        if ((n_tasks - 1) % world == r) px -= (size_t)(n_tasks * task_rows - H) * (size_t)W;      // the frame's last task may be short
        return px;
    };
    rtw::GatherBlocks b;
    size_t total = 0;
    if (comm->rank == 0) {
        b.first_rank = 1; b.n = world - 1;
        for (int r = 1; r < world; r++) { b.off[r - 1] = total; b.px[r - 1] = (uint32_t)pixels_of(r); total += rtw::gather_block_bytes(pixels_of(r), with_accum); }
    } else {
        b.first_rank = comm->rank; b.n = 1; b.off[0] = 0; b.px[0] = (uint32_t)pixels_of(comm->rank);
        total = rtw::gather_block_bytes(pixels_of(comm->rank), with_accum);
    }
    if (total > comm->stage_bytes) {        // grow-only; the first gather of a frame shape pays it (bench.py's warm-up gather does)
        HIP_TRY(hipStreamSynchronize(st));
        if (comm->d_stage) { (void)hipFree(comm->d_stage); comm->d_stage = nullptr; comm->stage_bytes = 0; }
        HIP_TRY(hipMalloc(&comm->d_stage, total));
        comm->stage_bytes = total;
    }
    ncclResult_t r = ncclSuccess;
    if (comm->rank != 0) {
        if (b.px[0] == 0) return RTW_OK;        // more ranks than tasks: nothing of this rank's to move (the root posts no receive for it either)
        const hipError_t e = (hipError_t)rtw::launch_gather_rows(false, fb->argb, fb->accum, comm->d_stage, W, task_rows, world, b, with_accum, st);
        if (e != hipSuccess) return hip_fail(e, "gather pack launch");
        r = a->Send(comm->d_stage, total, ncclChar, 0, comm->comm, st);
        comm->messages++;
    } else {
        if ((r = a->GroupStart()) == ncclSuccess) {
            for (int k = 0; k < b.n && r == ncclSuccess; k++) {
                if (b.px[k] == 0) continue;
                r = a->Recv((char*)comm->d_stage + b.off[k], rtw::gather_block_bytes(b.px[k], with_accum), ncclChar, b.first_rank + k, comm->comm, st);
                comm->messages++;
            }
            const ncclResult_t e = a->GroupEnd();
            if (r == ncclSuccess) r = e;
        }
        if (r == ncclSuccess) {
            const hipError_t e = (hipError_t)rtw::launch_gather_rows(true, fb->argb, fb->accum, comm->d_stage, W, task_rows, world, b, with_accum, st);
            if (e != hipSuccess) return hip_fail(e, "gather unpack launch");
        }
    }
    if (r != ncclSuccess) return rccl_fail(a, r, "rtw_gather_rows");
    return RTW_OK;
}

long long rtw_comm_messages(const rtw_comm* comm) { return comm ? comm->messages : 0; }

// rank 0's *value to every rank of the communicator (one 4-byte message per peer, through the staging block); returns when it has arrived
int rtw_comm_broadcast_int(rtw_comm* comm, int* value)
{
    if (!comm || !value) return fail(RTW_ERR_INVALID, "null argument");
    if (comm->world == 1) return RTW_OK;
    const RcclApi* a = rccl_api();
    if (!a) return fail(RTW_ERR_STATE, g_rccl.error);
    HIP_TRY(hipSetDevice(comm->ctx->device));
    hipStream_t st = comm->ctx->stream;
    if (comm->stage_bytes < 256) {
        HIP_TRY(hipStreamSynchronize(st));
        if (comm->d_stage) { (void)hipFree(comm->d_stage); comm->d_stage = nullptr; comm->stage_bytes = 0; }
        HIP_TRY(hipMalloc(&comm->d_stage, 256));
        comm->stage_bytes = 256;
    }
    ncclResult_t r = ncclSuccess;
    if (comm->rank == 0) {
        HIP_TRY(hipMemcpyAsync(comm->d_stage, value, sizeof(int), hipMemcpyHostToDevice, st));
        if ((r = a->GroupStart()) == ncclSuccess) {
            for (int peer = 1; peer < comm->world && r == ncclSuccess; peer++) { r = a->Send(comm->d_stage, sizeof(int), ncclChar, peer, comm->comm, st); comm->messages++; }
            const ncclResult_t e = a->GroupEnd();
            if (r == ncclSuccess) r = e;
        }
        HIP_TRY(hipStreamSynchronize(st));
    } else {
        r = a->Recv(comm->d_stage, sizeof(int), ncclChar, 0, comm->comm, st);
        comm->messages++;
        if (r == ncclSuccess) {
            HIP_TRY(hipMemcpyAsync(value, comm->d_stage, sizeof(int), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
        }
    }
    if (r != ncclSuccess) return rccl_fail(a, r, "rtw_comm_broadcast_int");
    return RTW_OK;
}

// ---- stats ------------------------------------------------------------------------------------------------------
int rtw_stats_enable(rtw_context* ctx, int enabled)
{
    if (!ctx) return fail(RTW_ERR_INVALID, "context is null");
    ctx->stats_enabled = enabled != 0;
    return RTW_OK;
}
int rtw_stats_reset(rtw_context* ctx)
{
    if (!ctx) return fail(RTW_ERR_INVALID, "context is null");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemsetAsync(ctx->d_stats, 0, 8 * sizeof(unsigned long long), ctx->stream));
    return RTW_OK;
}
int rtw_stats_get(rtw_context* ctx, rtw_stats* out)
{
    if (!ctx || !out) return fail(RTW_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    unsigned long long h[8];
    HIP_TRY(hipMemcpyAsync(h, ctx->d_stats, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
#ifdef RTW_BOUNDS_DEBUG
    std::fprintf(stderr, "RTW_BOUNDS_DEBUG: site=%llu value=%lld\n", h[6], (long long)h[7]);
#endif
    out->rays = h[0]; out->box_tests = h[1]; out->tri_tests = h[2]; out->shaded_hits = h[3]; out->tex_samples = h[4]; out->camera_rays = h[5];
    return RTW_OK;
}

#ifdef RTW_TIMING
int rtw_debug_read_timing(unsigned long long* out, int n) { return rtw::read_timing(out, n); }
#endif

// ---- tables + file helpers (host only, no device needed) ------------------------------------------------------------
uint32_t rtw_rand31(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t counter) { return rtw::rand31(seed, pixel, sample, counter); }
int rtw_unit_table_entry(uint32_t index, float out3[3])
{
    if (!out3 || index >= RTW_TABLE_SIZE) return fail(RTW_ERR_INVALID, "bad table index");
    rtw::unit_table_entry(index, out3);
    return RTW_OK;
}
int rtw_gamma_thresholds(float out256[256]) { if (!out256) return fail(RTW_ERR_INVALID, "null"); rtw::gamma_thresholds(out256); return RTW_OK; }
int rtw_texel_lut(float out256[256]) { if (!out256) return fail(RTW_ERR_INVALID, "null"); rtw::texel_lut(out256); return RTW_OK; }

int rtw_png_load(const char* path, uint8_t** texels_out, int* width, int* height, int* channels)
{
    if (!path || !texels_out || !width || !height || !channels) return fail(RTW_ERR_INVALID, "null argument");
    std::vector<uint8_t> px; int w = 0, h = 0, c = 0;
    const std::string err = rtw::png_load(path, px, w, h, c);
    if (!err.empty()) return fail(RTW_ERR_IO, err);
    uint8_t* out = (uint8_t*)std::malloc(px.size() ? px.size() : 1);
    if (!out) return fail(RTW_ERR_IO, "out of memory");
    std::memcpy(out, px.data(), px.size());
    *texels_out = out; *width = w; *height = h; *channels = c;
    return RTW_OK;
}
void rtw_png_free(uint8_t* texels) { std::free(texels); }

int rtw_png_save_argb(const char* path, const uint32_t* argb, int width, int height)
{
    if (!path || !argb || width <= 0 || height <= 0) return fail(RTW_ERR_INVALID, "bad argument");
    std::vector<uint8_t> rgb((size_t)width * (size_t)height * 3);
    for (size_t i = 0; i < (size_t)width * (size_t)height; i++) {     // GetUint32ColorRed/Green/Blue (Src/ColorBuffer.h:43-68)
        rgb[i * 3] = (uint8_t)((argb[i] >> 16) & 0xFF); rgb[i * 3 + 1] = (uint8_t)((argb[i] >> 8) & 0xFF); rgb[i * 3 + 2] = (uint8_t)(argb[i] & 0xFF);
    }
    const std::string err = rtw::png_save_rgb(path, rgb.data(), width, height);
    if (!err.empty()) return fail(RTW_ERR_IO, err);
    return RTW_OK;
}

}  // extern "C"
