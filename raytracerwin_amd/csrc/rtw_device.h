// rtw_device.h -- launch wrappers implemented in rtw_device.hip (internal).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <climits>

#include "rtw_types.h"

#define RTW_MAX_BOUNCE_DEV 16

namespace rtw {

// bytes of level workspace a launch of `work_items` threads needs (rounded up to whole 256-thread blocks)
inline size_t level_workspace_bytes(long long work_items, int max_bounce)
{
    const size_t threads = (size_t)((work_items + 255) / 256) * 256;
    return threads * (size_t)(max_bounce > 0 ? max_bounce : 1) * 3 * 16;
}
struct PipelineLayout { size_t queue_off, pend_off, counters_off, rad_off, hit_off, state_off, tlist0_off, tlist1_off, ws_off, total, capacity; };
// workspace of the bins + wave pipeline (pipeline 3) for a launch of `work_items` pixels (device bytes)
size_t pipeline_workspace_bytes(long long work_items, int max_bounce, PipelineLayout* out);
struct PipelineTuning {
    int expected_paths;    // queue length seen by the previous pass on this context, -1 = unknown
    int round_hint[32];    // trace-list lengths of the previous pass per round, -1 = unknown
    // jobs [sky_job0, n_jobs) of the job table are sky-only tiles, rendered by primary_sky_kernel on aux_stream
    // between the two events (fork after the previous work of the main stream, join before the pass ends); aux_stream null = one kernel
    hipStream_t aux_stream; hipEvent_t fork_event, join_event; int sky_job0; const float* gamma_thr;
    bool has_analytic = false;  // some shape is a sphere / plane / capsule: the kernels' instantiations that hold those tests are launched
    // a run of passes inside one rtw_render_passes call forks the second stream before its first pass and joins it after its last:
    // between those passes nothing else can be enqueued, the sky kernels are ordered on their own stream and write pixels no other
    // kernel of the run writes
    bool do_fork = true, do_join = true;
    bool* aux_unjoined = nullptr;   // set when a sky kernel was launched and the join was left to a later pass
    bool skip_trace = false;    // every shape is a leading analytic shape: the shading lanes do the whole scene query, no trace launches
    bool counters_clean;   // the counters are known to be zero (left so by the previous pass): no memset
    int cu_count;
    hipEvent_t* timing;    // null, or 4 events recorded before the primary kernel and after each of the three stages
};
int launch_render_pipeline(const RtwSceneDev* sc, void* accum, void* argb, void* workspace, const RtwRenderParams& p, const PipelineTuning& tune, bool stats, hipStream_t stream);
// device address of the pipeline's counters inside the workspace (for the asynchronous read-back of the queue length)
size_t pipeline_counters_offset(long long work_items, int max_bounce);
// ---- pass-batched pipeline (pipeline 4, rtw_group_kernels.h) ----
// slots of a group's workspace (rtw_group_kernels.h group_slot): one per path and pass
inline size_t group_capacity(size_t paths_per_pass, int n_passes) { return paths_per_pass * (size_t)(n_passes > 0 ? n_passes : 1); }
struct GroupLayout { size_t counters_off, rad_off, state_off, hit_off, carry_off, levels_off, list0_off, list1_off, overflow_off, tlist0_off, tlist1_off, total; };
// device bytes of a group's workspace: `capacity` path slots (busy tiles x 64 x sub-samples x passes of the group, rounded up to a power of two of passes)
size_t group_workspace_bytes(size_t capacity, int max_bounce, bool carry, GroupLayout* out);
struct GroupTuning {
    size_t capacity = 0;
    hipStream_t aux_stream = nullptr; hipEvent_t fork_event = nullptr, join_event = nullptr;   // sky-only tiles run on aux_stream beside the rest of the group
    bool do_fork = true, do_join = true; bool* aux_unjoined = nullptr;
    const float* gamma_thr = nullptr;
    bool has_analytic = false;     // some shape is a sphere / plane / capsule / triangle
    bool carry = false;            // ... and one of them follows a textured mesh: hit records carry the mesh hit whose texel the hit keeps
    bool counters_clean = false;   // the list counters are zero (left so by the previous group)
    int round_hint[32];            // list lengths of the previous group per round (-1 = unknown): they size the launches
    int cu_count = 256;
    bool single_mesh = false;      // the scene is one mesh: the trace rounds run persistent waves that refill their lanes
    bool lead_mesh = false;        // ... or leading analytic shapes followed by ONE mesh (shape staged_shape), no texel inheritance: the same kernel, continuing the lead query
    // a group rendered as several parts on as many streams (rtwin_capi.cpp: rtw_render_passes): the group's sky tiles are rendered for ALL its passes
    // (sky_first_pass, sky_passes) by one launch enqueued ahead of the parts (sky_mode 2); a part's resolve kernel waits for the previous part's
    // (resolve_after) and signals the next (resolve_done)
    int sky_blocks = 4;  // blocks of the sky kernel per CU at most (its lanes loop over the tiles)
    int sky_passes = 0, sky_first_pass = 0, sky_mode = 0; hipEvent_t resolve_after = nullptr, resolve_done = nullptr;       // sky_mode 0: a whole group (sky on aux_stream beside it), 1: a part of a split group (no sky tiles), 2: ONLY the sky tiles of a split group, all its passes
    int visit_budget = INT32_MAX;            // persistent trace waves: a ray that needs more node visits than this is handed to a wave-per-ray launch that follows
    int trace_hint[32];                      // scenes with leading analytic shapes: how many rays of a round still went to the trace launch (previous group)
    bool skip_trace = false;                 // every shape is a leading analytic shape: the shading lanes do the whole query, no trace launches
    int overflow_hint[32];                   // how many rays that were in the previous group, per round (-1 = unknown)
    int wave_below = 0;                      // a trace round whose list (previous group's length) is shorter than this runs a wave per ray
    bool staged_all = false;                 // ... and that is the shape's whole tree
    int staged_tris = 0;                     // triangles of the staged shape
    bool staged_planes = false;              // ... and the triangles' planes are staged beside it (persistent kernel: back-facing leaves are never noted)
    int staged_shape = -1, staged_top = 0;   // the trace blocks stage the first staged_top tnodes records of this shape in LDS (-1: nothing staged)
    hipEvent_t* timing = nullptr;  // null, or 4 events: before the primary kernel, after it, after the bounce rounds, after resolve
};
int launch_render_group(const RtwSceneDev* sc, void* accum, void* argb, void* workspace, const RtwGroupParams& g, const GroupTuning& tune, bool stats, hipStream_t stream);
// ---- KdNode::Build and the layouts derived from the tree, on the device (rtw_build_kernels.h) ----
struct DeviceBuildIn {      // the arrays RMeshShape owns (host memory)
    const float* points; int n_points; const float* texcoords; int n_texcoords; const float* normals; int n_normals;
    const int32_t* idx_p; const int32_t* idx_t; const int32_t* idx_n; const int32_t* tri_material; int n_tris;
};
struct DeviceBuildOut {     // device memory, the caller's from here on (hipFree)
    RtwNode* nodes; RtwPNode* tnodes; RtwTri* tris; RtwShade* shade; float* flat[3];
    int n_nodes, tnodes_top, max_depth, flat_n[3], flat_pad[3];
};
int device_build_mesh(const DeviceBuildIn& in, int top_budget, DeviceBuildOut* out, hipStream_t stream);
int device_build_bins(const RtwNode* d_nodes, const RtwTri* d_tris, int n_nodes, int width, int height, int bin_w, int bin_h,
                      uint32_t** off_out, uint32_t** ent_out, uint32_t* h_off, int* has_bins, hipStream_t stream);
#define RTW_PERSIST_CAND_BYTES (8 * 1024 * 4)   // candidate lists of a persistent trace block: RTW_GT_CAP_STAGED words x 1024 threads
#define RTW_TNODES_TOP_BUDGET 3072 // records (32 B each) of a tree's upper levels a block of the ray-per-lane trace kernel stages in LDS: 96 KiB
int launch_render(const RtwSceneDev* sc, void* accum, void* argb, void* ws, const RtwRenderParams& p, bool stats, hipStream_t stream);
int launch_closest(const RtwSceneDev* sc, const float* rays, long long n, float* hits11, int* shape, int* tri, bool stats, hipStream_t stream);
int launch_ray_trace(const RtwSceneDev* sc, const float* rays, const uint32_t* keys2, long long n, int max_bounce, int preview,
                     uint32_t seed, unsigned long long npix, float* rgb, void* ws, bool stats, hipStream_t stream);
int launch_texture_sample(const RtwSceneDev* sc, int shape, int mat, const float* uv, long long n, float* rgba, hipStream_t stream);
// ---- multi-GPU gather (rtw_gather_rows): every rank's task rows travel as ONE compact block per rank -- [ARGB of its pixels in task order |
// their accumulators] -- packed by the sender, unpacked into the framebuffer by the root, all peers in one launch ----
struct GatherBlocks { int n = 0, first_rank = 0; uint64_t off[64]; uint32_t px[64]; };     // blocks of ranks first_rank .. first_rank + n - 1 inside the staging buffer
size_t gather_block_bytes(size_t pixels, bool with_accum);
int launch_gather_rows(bool unpack, void* argb, void* accum, void* stage, int width, int task_rows, int world, const GatherBlocks& b, bool with_accum, hipStream_t stream);

}  // namespace rtw
