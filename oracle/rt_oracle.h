/*
 * rt_oracle.h -- C API of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference's
 * per-pixel ray-trace hot path (aosyang/RayTracerWin).  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load it; the product library
 * (raytracerwin_amd/csrc) never links, imports or calls anything in oracle/.
 *
 * Parity status: PINNED against the real reference, compiled from its own
 * sources by oracle/Makefile into oracle/_ref/ (see oracle/ref_harness.cpp and
 * tests/golden/).
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Material node types (flattened ISurfaceMaterial tree, root = node 0).
 * Reference: Src/SurfaceMaterials.h:47-141. */
enum {
    ORC_MAT_DIFFUSE = 0,        /* rgb = albedo */
    ORC_MAT_DIFFUSE_CHECKER = 1,/* rgb = albedo, param = pattern size */
    ORC_MAT_REFLECTIVE = 2,     /* rgb = albedo, param = fuzziness */
    ORC_MAT_EMISSIVE = 3,       /* rgb = colour */
    ORC_MAT_BLEND = 4,          /* param = blend factor, child_a, child_b */
    ORC_MAT_COMBINE = 5,        /* child_a, child_b */
    ORC_MAT_NULL = 6
};

typedef struct {
    int32_t type;
    float r, g, b;
    float param;
    int32_t child_a, child_b;
    int32_t pad;
} orc_material_node;            /* 32 bytes, same layout as rt_material_node */

typedef struct {
    uint64_t rays;          /* FindIntersectionWithScene calls */
    uint64_t box_tests;     /* RRay::TestIntersectionWithAabb calls (incl. shape bound) */
    uint64_t tri_tests;     /* TestIntersectionWithTriangle calls */
    uint64_t shaded_hits;   /* mesh hits that ran the shading-input block */
    uint64_t tex_samples;   /* RTexture::Sample calls */
    uint64_t camera_rays;
} orc_stats;

typedef struct orc_scene orc_scene;
typedef struct orc_framebuffer orc_framebuffer;

/* unit-vector transcendental mode for the fuzzy-reflection direction */
enum { ORC_UNITVEC_LIBM = 0, ORC_UNITVEC_F64 = 1 };

orc_scene* orc_scene_create(void);
void orc_scene_destroy(orc_scene*);

/* Parse an OBJ (+ sibling MTL) with the reference's parser semantics and build
 * the one-triangle-per-leaf tree.  Returns shape index or <0. */
int orc_scene_add_mesh_obj(orc_scene*, const char* obj_path);
/* RSphere / RPlane / RCapsule (Src/Shapes.h:46-112, Src/Shapes.cpp:18-125, Src/RRay.cpp:25-87); shapes keep insertion order */
enum { ORC_SHAPE_MESH = 0, ORC_SHAPE_SPHERE = 1, ORC_SHAPE_PLANE = 2, ORC_SHAPE_CAPSULE = 3, ORC_SHAPE_TRIANGLE = 4 };
int orc_scene_add_sphere(orc_scene*, const float center[3], float radius);
int orc_scene_add_plane(orc_scene*, const float normal[3], const float point[3]);
int orc_scene_add_capsule(orc_scene*, const float start[3], const float end[3], float radius);
int orc_scene_add_triangle(orc_scene*, const float p0[3], const float p1[3], const float p2[3]);   /* RTriangle (Src/Shapes.h:106-130) */
int orc_scene_set_material(orc_scene*, int shape, const orc_material_node* nodes, int n);

/* mesh introspection (for parser / topology digests) */
int orc_mesh_counts(const orc_scene*, int shape, int32_t out[8]);
/* which: 0 points(3f) 1 texcoords(3f) 2 normals(3f) 3 point idx 4 tc idx 5 normal idx 6 tri material */
int orc_mesh_copy(const orc_scene*, int shape, int which, void* dst, int64_t max_bytes);
int orc_mesh_num_materials(const orc_scene*, int shape);
/* texture path that the MTL's map_Kd names for material id (empty if none) */
int orc_mesh_texture_path(const orc_scene*, int shape, int material_id, char* buf, int buflen);
/* hand over decoded 8-bit texels (channels 3 = RGB, 4 = RGBA) */
int orc_mesh_set_texture(orc_scene*, int shape, int material_id, const uint8_t* px, int w, int h, int channels);
/* preorder dump of the pointer tree: per node 6 floats bounds + int tri (-1 internal) */
int orc_mesh_tree_preorder(const orc_scene*, int shape, float* bounds6, int32_t* tri, int max_nodes);
int orc_shape_bounds(const orc_scene*, int shape, float out6[6]);

/* FindIntersectionWithScene for n rays (7 floats each: o.xyz d.xyz dist).
 * hits: 12 floats each: pos3 normal3 dist color3 alpha, shape index as float bits in [11]?  -> separate arrays */
int orc_trace_closest(const orc_scene*, const float* rays, int64_t n,
                      float* hit_f11, int32_t* hit_shape, int32_t* hit_tri);

/* RTexture::Sample probes on (shape, material id) */
int orc_texture_sample(const orc_scene*, int shape, int material_id, const float* uv, int64_t n, float* rgba);

/* RayTrace for explicit rays with explicit RNG keys (pixel,sample per ray) */
int orc_ray_trace(const orc_scene*, const float* rays, const uint32_t* keys2, int64_t n,
                  int max_bounce, int use_base_color, uint32_t seed, int width, int height, float* rgb);

orc_framebuffer* orc_framebuffer_create(int width, int height);
void orc_framebuffer_destroy(orc_framebuffer*);
void orc_framebuffer_clear(orc_framebuffer*);
int orc_framebuffer_read(const orc_framebuffer*, float* accum4 /*sum.xyz,n as float*/, uint32_t* argb);

/* ThreadWorker_Render restated (Src/RayTracerProgram.cpp:131-188) with run-time
 * width/height, sub-sample count ns in 1..4, pass index, seed. */
int orc_render_range(const orc_scene*, orc_framebuffer*, int begin, int end, int max_bounce,
                     int use_base_color, int pass_index, int ns, uint32_t seed);

/* One full pass through a ThreadTaskQueue-style pool of `threads` workers pulling
 * `task_rows`-row tasks (Src/RayTracerProgram.cpp:270-327).  Returns seconds. */
double orc_render_pass_pool(const orc_scene*, orc_framebuffer*, int max_bounce, int use_base_color,
                            int pass_index, int ns, uint32_t seed, int threads, int task_rows);

void orc_set_unitvec_mode(orc_scene*, int mode);
void orc_set_combine_order(orc_scene*, int b_first);
void orc_stats_reset(void);
void orc_stats_get(orc_stats* out);

/* raw RNG + tables, exposed so tests can pin them against the product */
uint32_t orc_rand31(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t counter);
void orc_unit_table_entry(uint32_t index, float out3[3]);
void orc_gamma_thresholds(float out256[256]);
void orc_texel_lut(float out256[256]);
int orc_hw_threads(void);
void orc_make_pixel_colors(const float* rgb, int64_t n, uint32_t* out);

/* test instrumentation: while set, every pixel rendered folds the hit / miss history of its paths into buf[pixel] (npix words, caller-owned) */
void orc_set_history_buffer(uint32_t* buf);

#ifdef __cplusplus
}
#endif
#endif
