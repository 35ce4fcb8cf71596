/*
 * rtwin.h -- C ABI of the MI355X-native ray-trace hot path (librtwin.so).
 *
 * The reference (aosyang/RayTracerWin) has no FFI: its hot path is reached only through
 * in-process C++ calls.  The narrowest seam is the pixel-range task call
 *     void ThreadWorker_Render(int begin, int end, int MaxBounceCount, const RenderOption&)
 * (Src/RayTracerProgram.cpp:131), fed by a scene built with
 *     RayTracerScene::AddShape(RMeshShape::Create(path), material)   (Src/RayTracerScene.cpp:25)
 * and drained through the global accuBuffer[] / bitcolor[] arrays (Src/RayTracerProgram.cpp:49,77).
 * Every entry point below names the reference interface it replaces.  Plain pointers and
 * sizes only; all device memory is owned by the library behind opaque handles (or wrapped
 * from the caller via rtw_framebuffer_wrap).  Every function returns RTW_OK (0) or a
 * negative rtw_status; rtw_last_error() gives the message for the calling thread.
 *
 * There is no CPU fallback: without a HIP device every device entry point fails with
 * RTW_ERR_NO_DEVICE.
 */
#ifndef RTWIN_H
#define RTWIN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    RTW_OK = 0,
    RTW_ERR_INVALID = -1,       /* bad argument / handle */
    RTW_ERR_NO_DEVICE = -2,     /* no usable HIP device */
    RTW_ERR_HIP = -3,           /* a HIP runtime call failed */
    RTW_ERR_IO = -4,            /* file could not be read / written */
    RTW_ERR_STATE = -5,         /* scene not committed, etc. */
    RTW_ERR_LIMIT = -6          /* exceeds a compiled-in limit (RTW_MAX_*) */
} rtw_status;

#define RTW_MAX_BOUNCE 16          /* deepest MaxBounceTimes the device path stack holds */
#define RTW_MAX_MATERIAL_NODES 64
#define RTW_MAX_MESH_MATERIALS 64  /* distinct usemtl names per mesh */
#define RTW_UNIT_TABLE_SIZE 0xFFFFFFu   /* MaxUnitVectorNums, Src/Math.cpp:17 */

/* Flattened ISurfaceMaterial tree (Src/SurfaceMaterials.h:47-141); root is node 0 and
 * children always have larger indices than their parent. */
typedef enum {
    RTW_MAT_DIFFUSE = 0,          /* SurfaceMaterial_Diffuse(albedo)                      */
    RTW_MAT_DIFFUSE_CHECKER = 1,  /* SurfaceMaterial_DiffuseChecker(albedo, param=size)   */
    RTW_MAT_REFLECTIVE = 2,       /* SurfaceMaterial_Reflective(albedo, param=fuzziness)  */
    RTW_MAT_EMISSIVE = 3,         /* SurfaceMaterial_Emissive(colour)                     */
    RTW_MAT_BLEND = 4,            /* SurfaceMaterial_Blend(A, B, param=factor)            */
    RTW_MAT_COMBINE = 5,          /* SurfaceMaterial_Combine(A, B)                        */
    RTW_MAT_NULL = 6              /* SurfaceMaterial_Null                                 */
} rtw_material_type;

typedef struct {
    int32_t type;
    float r, g, b;
    float param;
    int32_t child_a, child_b;
    int32_t pad;
} rtw_material_node;              /* 32 bytes */

typedef struct {
    uint64_t rays;          /* FindIntersectionWithScene-equivalent queries            */
    uint64_t box_tests;     /* node / shape box tests actually executed                */
    uint64_t tri_tests;     /* triangle tests executed                                 */
    uint64_t shaded_hits;   /* hits that ran the shading-input block                   */
    uint64_t tex_samples;   /* bilinear texture samples                                */
    uint64_t camera_rays;
} rtw_stats;

typedef struct rtw_context rtw_context;
typedef struct rtw_scene rtw_scene;
typedef struct rtw_framebuffer rtw_framebuffer;

/* ---- context: one per (process, GPU).  Replaces the process-global state of
 * RayTracerProgram::Run (Src/RayTracerProgram.cpp:437-443: srand + the unit-vector table
 * of RMath::InitPseudoRandomUnitVector, Src/Math.cpp:24-31). ---- */
int rtw_context_create(int device_index, rtw_context** out);
int rtw_context_destroy(rtw_context* ctx);
/* run every later launch of this context on the caller's hipStream_t (e.g. torch's) */
int rtw_context_set_stream(rtw_context* ctx, void* hip_stream);
int rtw_context_synchronize(rtw_context* ctx);
/* Switches; results never depend on them (every combination is tested bit-identical).
 *   "pipeline"      4 (default) = pass-batched: the passes of one rtw_render_passes call are rendered in groups that share one set of launches (screen
 *                   bins for the camera rays, a ray per lane for the bounce rounds; rtw_group_kernels.h), every frame shape, pixel range and scene;
 *                   3 = one pass per set of launches: screen bins for the camera rays + one wave-per-ray trace launch and one shade launch per bounce
 *                   (the one-pass reference the pass-batched pipeline is compared with); 0 = one kernel, one thread per pixel (with
 *                   rtw_scene_set_traversal(0): the reference's own visit order, for the work counters).
 *   "group_max" (256) passes per group at most (a power of two), "group_paths" (32 Mi) paths a launch should hold; "group_split" (1) a group of at least
 *                   "split_min" (8) passes runs as up to "group_parts" (2; at most 4) parts of at least "split_paths" (400 000) paths each, on as many streams
 *                   (a workspace per part);
 *   "wave_below" (80 000; x 5 for trees of more than 4 096 nodes) a trace round with fewer rays runs a wave per ray; "visit_budget" (384) one-mesh scenes:
 *                   a ray's node visits in the ray-per-lane kernel before it goes to the wave-per-ray one (a rare ray that walks a thousand nodes kept its
 *                   whole wave waiting), for trees with more than "budget_nodes" (0) nodes;
 *   "backface_filter" (1) one-mesh scenes whose tree fits LDS: the persistent trace blocks also stage the triangles' planes and never note a leaf whose
 *                   triangle faces away from the ray's origin (RRay::TestIntersectionWithTriangle's first rejection, which does not depend on the segment);
 *   "primary_passes" (0 = chosen per launch; 1..4; -1 = one ray set at a time) one-mesh scenes: passes of a tile one wave of the primary kernel takes, its bin
 *                   walked once for up to four rays per lane (sub-samples x passes <= 4); "sky_blocks" (4) blocks of the sky kernel per CU at most;
 *   "workspace_limit_mb" (0 = 24 GiB) a group's workspace may not exceed this: larger groups are re-formed smaller (see rtw_context_memory_bytes);
 *   "device_build" (1) tree, layouts and screen bins built on the device; "hint_period" (16) pipeline 3: the queue lengths that size the next launches
 *                   are read back every n-th pass; "kernel_timing" 1 = record events around the stages of each pass / group (rtw_last_pass_kernel_ms). */
int rtw_context_set_option(rtw_context* ctx, const char* name, int value);
/* with option "kernel_timing" = 1: HIP-event durations (ms) of the latest pass's three stages -- primary kernel(s),
 * the per-bounce trace / shade launches, resolve -- measured on the context's stream; waits for that pass. */
int rtw_last_pass_kernel_ms(rtw_context* ctx, float out3[3]);
/* passes the latest group of the pass-batched pipeline held (rtw_last_pass_kernel_ms times that group) */
int rtw_last_group_passes(rtw_context* ctx);
/* the pipeline the latest render call of this context actually ran (0, 3 or 4; -1 before the first call): lets a caller see a fallback */
int rtw_last_pass_pipeline(rtw_context* ctx);
const char* rtw_last_error(void);
const char* rtw_version(void);

/* ---- scene: RayTracerScene (Src/RayTracerScene.h:43-62) ---- */
int rtw_scene_create(rtw_context* ctx, rtw_scene** out);
int rtw_scene_destroy(rtw_scene* scene);
/* RayTracerScene::AddShape(RMeshShape::Create(path), ...) : OBJ + sibling MTL + PNG
 * textures with the reference's parser semantics (Src/MeshShape.cpp:65-278). */
int rtw_scene_add_mesh_obj(rtw_scene* scene, const char* obj_path, int* out_shape);
/* RayTracerScene::AddShape(RSphere::Create(center, radius), ...) / RPlane::Create(normal, point) / RCapsule::Create(start, end,
 * radius) (Src/Shapes.h:46-112; intersections Src/RRay.cpp:25-87 and Src/Shapes.cpp:18-125).  Shapes keep their insertion order
 * (it decides which of two hits at one distance wins, and which hit's sampled colour a later sphere / plane / capsule-side hit
 * inherits: one RayHitResult serves all shapes of a query, Src/RayTracerScene.cpp:99-125).  A plane has no culling box
 * (RPlane::HasCullingBounds).  The material is set with rtw_scene_set_material as for a mesh. */
int rtw_scene_add_sphere(rtw_scene* scene, const float center[3], float radius, int* out_shape);
int rtw_scene_add_plane(rtw_scene* scene, const float normal[3], const float point[3], int* out_shape);
int rtw_scene_add_capsule(rtw_scene* scene, const float start[3], const float end[3], float radius, int* out_shape);
/* RTriangle::Create(p0, p1, p2) (Src/Shapes.h:106-130): one single-sided triangle with its face normal, culled by the box of its
 * three points (so an axis-aligned one is as invisible as the reference's: its zero-thickness box never passes the slab test). */
int rtw_scene_add_triangle(rtw_scene* scene, const float p0[3], const float p1[3], const float p2[3], int* out_shape);
/* Same, from arrays the caller already holds (the members of RMeshShape,
 * Src/MeshShape.h:25-37).  positions/texcoords/normals are 3 floats per element,
 * idx_* are 3 ints per triangle (0-based), tri_material is 1 int per triangle (-1 = none).
 * shape_bounds6 may be NULL (then the bounds of all positions, Src/MeshShape.cpp:109). */
int rtw_scene_add_mesh(rtw_scene* scene,
                       const float* positions, int n_positions,
                       const float* texcoords, int n_texcoords,
                       const float* normals, int n_normals,
                       const int32_t* idx_p, const int32_t* idx_t, const int32_t* idx_n,
                       const int32_t* tri_material, int n_tris,
                       const float* shape_bounds6, int* out_shape);
/* RTexture::LoadTexturePNG result for one material id (Src/Texture.cpp:119-151):
 * 8-bit texels, channels 3 (RGB, alpha = 1) or 4 (RGBA), row-major, top row first. */
int rtw_scene_set_texture(rtw_scene* scene, int shape, int material_id,
                          const uint8_t* texels, int width, int height, int channels);
/* RShape::SetSurfaceMaterial (Src/Shapes.cpp:9) */
int rtw_scene_set_material(rtw_scene* scene, int shape, const rtw_material_node* nodes, int n_nodes);
/* build + flatten + upload; the scene is immutable afterwards (all reference queries are const) */
int rtw_scene_commit(rtw_scene* scene);
/* info[0..7] = points, texcoords, normals, triangles, materials, nodes, textures, max tree depth */
int rtw_scene_mesh_info(const rtw_scene* scene, int shape, int32_t info[8], float shape_bounds6[6]);
/* flattened tree for inspection: per node 6 floats (min, max), skip index, triangle (-1 internal) */
int rtw_scene_mesh_nodes(const rtw_scene* scene, int shape, float* bounds6, int32_t* skip, int32_t* tri, int max_nodes);
/* traversal pruning (result-preserving segment clip); default on */
int rtw_scene_set_prune(rtw_scene* scene, int enabled);
/* 1 (default): screen bins, link tree and flat hierarchy may be used (candidate leaves gathered in the reference's order before their
 * triangle tests); 0: every ray walks the binary tree in preorder, visiting exactly the boxes KdNode::TestRayIntersection
 * visits (Src/KdTree.cpp:128-195; one thread per pixel; used for reference-faithful work counters).  Same results. */
int rtw_scene_set_traversal(rtw_scene* scene, int mode);
/* The flat hierarchy over the leaves in preorder that the wave-per-ray walk uses: level 0 = each
 * leaf's own box (KdNode::Bounds of the leaf, Src/KdTree.cpp:37-60), level 1 / 2 = unions of 16 /
 * 256 consecutive leaves.  boxes6 = min.xyz, max.xyz per entry.  Returns the level's entry count. */
int rtw_scene_mesh_flat(const rtw_scene* scene, int shape, int level, float* boxes6, int max_entries);
/* Screen-space bins of the reference's fixed camera (Src/RayTracerProgram.cpp:133-165) for a
 * width x height frame cut into bin_w x bin_h pixel bins: CSR offsets (bins + 1) and, per bin, the
 * node indices (ascending) of the leaves whose triangle a camera ray of the bin's pixels could accept
 * (it faces the camera by the reference's own float test and its projection touches the bin).
 * counts2 = {number of offsets, number of entries}.  Returns 1, or 0 when this mesh gets no bins. */
int rtw_scene_mesh_bins(const rtw_scene* scene, int shape, int width, int height, int bin_w, int bin_h,
                        uint32_t* offsets, int64_t max_offsets, uint32_t* entries, int64_t max_entries, int64_t* counts2);

/* ---- ray-level queries (parity surface) ---- */
/* RayTracerScene::FindIntersectionWithScene (Src/RayTracerScene.cpp:99-125) for n rays.
 * rays: 7 floats each (origin, direction, distance).  hits: 11 floats each (HitPosition,
 * HitNormal, Distance, SampledColor, SampledAlpha); shape/tri: hit shape index (-1 miss)
 * and original triangle index.  Host buffers. */
int rtw_trace_closest(rtw_scene* scene, const float* rays, int64_t n,
                      float* hits11, int32_t* shape, int32_t* tri);
/* RayTracerScene::RayTrace (Src/RayTracerScene.cpp:31-97) for n explicit rays; keys2 holds
 * (pixel, sample) per ray, which select the random stream.  rgb: 3 floats each. */
int rtw_ray_trace(rtw_scene* scene, const float* rays, const uint32_t* keys2, int64_t n,
                  int max_bounce, int use_base_color, uint32_t seed, int width, int height, float* rgb);
/* RTexture::Sample (Src/Texture.cpp:23-57) probes: uv 2 floats each -> rgba 4 floats each */
int rtw_texture_sample(rtw_scene* scene, int shape, int material_id, const float* uv, int64_t n, float* rgba);

/* ---- framebuffer: accuBuffer[] + bitcolor[] (Src/RayTracerProgram.cpp:49-77) ---- */
int rtw_framebuffer_create(rtw_context* ctx, int width, int height, rtw_framebuffer** out);
/* wrap caller-owned device memory: accum = width*height*16 bytes (sum.xyz, int count),
 * argb = width*height*4 bytes.  Lets torch own the buffers that RCCL gathers. */
int rtw_framebuffer_wrap(rtw_context* ctx, int width, int height, void* accum_dev, void* argb_dev, rtw_framebuffer** out);
int rtw_framebuffer_destroy(rtw_framebuffer* fb);
int rtw_framebuffer_clear(rtw_framebuffer* fb);
/* copy out: accum4 = width*height*4 floats (sum.xyz, count as float); argb = 0xAARRGGBB */
int rtw_framebuffer_read_float(rtw_framebuffer* fb, float* accum4);
int rtw_framebuffer_resolve_argb(rtw_framebuffer* fb, uint32_t* argb);
/* the device addresses of the two arrays (accum: width*height float4 = sum.xyz + int count; argb: width*height 0xAARRGGBB words), e.g. for a display
 * hook that blits bitcolor[] straight out of device memory (Src/Windows/RenderWindow.cpp:150-187 blits the host array); valid while the framebuffer
 * lives; order reads after the renders on the context's stream.  Either pointer argument may be NULL. */
int rtw_framebuffer_device_pointers(rtw_framebuffer* fb, void** accum_dev, void** argb_dev);

/* ---- the hot path ---- */
/* ThreadWorker_Render(begin, end, MaxBounceCount, RenderOption{use_base_color})
 * (Src/RayTracerProgram.cpp:131-188) for one pass: pixel indices begin..end INCLUSIVE,
 * row-major.  sub_samples in 1..4 (the reference always uses 4); pass_index and seed
 * select the random streams.  Asynchronous on the context's stream; calls on disjoint
 * ranges may be issued back to back. */
int rtw_render_range(rtw_scene* scene, rtw_framebuffer* fb, int begin, int end, int max_bounce,
                     int use_base_color, int pass_index, int sub_samples, uint32_t seed);
/* One pass over this rank's share of the frame: the reference's NumTaskRows-row tasks
 * (Src/RayTracerProgram.cpp:282,294-301) dealt round-robin, task t belongs to rank
 * t % world.  Pixels keep their global indices (and random streams), so the image is
 * identical for every world size. */
int rtw_render_tasks(rtw_scene* scene, rtw_framebuffer* fb, int task_rows, int rank, int world,
                     int max_bounce, int use_base_color, int pass_index, int sub_samples, uint32_t seed);
/* UpdateBitmapPixels' sample loop (Src/RayTracerProgram.cpp:317-361): n_passes accumulated passes
 * first_pass .. first_pass + n_passes - 1 over this rank's tasks, as rtw_render_tasks would render
 * them one by one (same final images).  The passes are rendered in groups that share one set of launches; a pixel's pass colours are
 * added to its accumulator one by one in pass order, its accumulator entry is written and its ARGB word (divide + gamma of the accumulator after
 * the group's last pass) computed and written once per group: intermediate images are not materialised inside a call (call it with
 * n_passes = 1 to show every pass). */
int rtw_render_passes(rtw_scene* scene, rtw_framebuffer* fb, int task_rows, int rank, int world,
                      int max_bounce, int use_base_color, int first_pass, int n_passes,
                      int sub_samples, uint32_t seed);

/* Prepare, without rendering, everything a later rtw_render_passes call with these arguments would otherwise make inside it: the screen bins and tile
 * tables of this frame shape and the workspace of the largest group of passes that call will form (the workspace is grow-only; growing it waits for the
 * context's streams -- measured 0.3 ms inside a timed call).  UpdateBitmapPixels has no counterpart: its buffers are static arrays. */
int rtw_render_reserve(rtw_scene* scene, rtw_framebuffer* fb, int task_rows, int rank, int world,
                       int max_bounce, int n_passes, int sub_samples);
/* Device memory the context holds: workspaces + its share of the 201 MB unit-vector table (one copy per device, shared by the contexts on it).
 * The workspace is sized by the largest group of passes rendered (or reserved) so far: per path slot 116 B + 48 B x max_bounce, slots = pixels of the
 * frame's busy tiles x sub-samples x passes per group; e.g. TorusKnot 1080p depth 4: 21 MB for one pass per call,
 * 0.43 GB for a 20-pass call.  When the device has no room for a group's workspace (or it exceeds the option "workspace_limit_mb"), the call renders
 * the same image in smaller groups instead of failing (rtw_context_fallbacks counts how often).  rtw_context_trim gives the workspaces back. */
long long rtw_context_memory_bytes(const rtw_context* ctx);
long long rtw_context_workspace_bytes(const rtw_context* ctx);
int rtw_context_trim(rtw_context* ctx);
int rtw_context_fallbacks(const rtw_context* ctx);

/* ---- multi-GPU: the one exchange of the path.  Every rank renders its own tasks (rtw_render_tasks / rtw_render_passes with rank, world)
 * into a full-size framebuffer; rtw_gather_rows then moves every rank's rows to rank 0 over RCCL (xGMI): one ncclSend / ncclRecv
 * per peer of a compact block of the rank's rows, on the context's stream.  No reference counterpart (the
 * reference is one process); it completes what UpdateBitmapPixels' WaitForAllTasksDone (Src/RayTracerProgram.cpp:303) is to one process.
 * librccl is loaded on first use (dlopen: the library the environment variable RTW_RCCL_LIBRARY names if it is set, else the copy already in the
 * process, e.g. PyTorch's, else librccl.so.1); single-GPU users never load it. ---- */
typedef struct rtw_comm rtw_comm;
#define RTW_COMM_ID_BYTES 128
/* rank 0: a fresh ncclUniqueId, to be handed to every rank by the caller (MPI, torch.distributed, a file ...) */
int rtw_comm_unique_id(uint8_t id[RTW_COMM_ID_BYTES]);
/* every rank: ncclCommInitRank on the context's device */
int rtw_comm_create(rtw_context* ctx, const uint8_t id[RTW_COMM_ID_BYTES], int rank, int world, rtw_comm** out);
/* or: use a communicator the caller already has (an ncclComm_t); it stays the caller's */
int rtw_comm_wrap(rtw_context* ctx, void* nccl_comm, int rank, int world, rtw_comm** out);
int rtw_comm_destroy(rtw_comm* comm);
#define RTW_GATHER_ALL 0        /* accumulator (16 B / pixel) + ARGB (4 B / pixel) */
#define RTW_GATHER_ARGB 1       /* the displayable image only, 4 B / pixel */
/* rows of task t (task_rows rows each, as rtw_render_tasks deals them: t belongs to rank t % world) travel from their owner to rank 0:
 * ONE message per peer -- a sender packs its rows into a compact block (a kernel), the root receives every peer's block inside one RCCL group and
 * unpacks them with one launch; the staging blocks (a rank's share of the frame x 4 or 20 bytes per pixel; the root holds every peer's) are
 * allocated by the first gather of a frame shape.  Asynchronous on the context's stream; every rank of the communicator must call it. */
int rtw_gather_rows(rtw_comm* comm, rtw_framebuffer* fb, int task_rows, int mode);
/* ncclSend / ncclRecv operations this communicator has issued so far (a gather costs a sender one, the root world - 1) */
long long rtw_comm_messages(const rtw_comm* comm);
/* rank 0's *value to every rank (a decision every rank must take alike, e.g. RayTracerProgram::IsTerminating polled on rank 0: a rank that left the
 * progressive loop alone would leave the others blocked in the next gather).  Synchronous; every rank of the communicator must call it. */
int rtw_comm_broadcast_int(rtw_comm* comm, int* value);

/* work counters of launches since the last reset (only counted while enabled) */
int rtw_stats_enable(rtw_context* ctx, int enabled);
int rtw_stats_reset(rtw_context* ctx);
int rtw_stats_get(rtw_context* ctx, rtw_stats* out);

/* tables the device uses, for pinning against the oracle */
uint32_t rtw_rand31(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t counter);
int rtw_unit_table_entry(uint32_t index, float out3[3]);
int rtw_gamma_thresholds(float out256[256]);
int rtw_texel_lut(float out256[256]);

/* ---- host-side file helpers: RTexture::LoadTexturePNG / SaveBufferToPNG
 * (Src/Texture.cpp:59-199, 201-283) ---- */
int rtw_png_load(const char* path, uint8_t** texels_out, int* width, int* height, int* channels);
void rtw_png_free(uint8_t* texels);
int rtw_png_save_argb(const char* path, const uint32_t* argb, int width, int height);

#ifdef __cplusplus
}
#endif
#endif /* RTWIN_H */
