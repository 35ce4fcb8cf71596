/*
 * rt_oracle.c -- CPU oracle: a plain-C restatement of the reference's per-pixel
 * ray-trace hot path (aosyang/RayTracerWin, all citations relative to the
 * reference root, e.g. Src/KdTree.cpp:128).
 *
 * TEST INFRASTRUCTURE ONLY (see rt_oracle.h).  Deliberately structured like the
 * reference, not like the product: recursive pointer tree with one triangle per
 * leaf, un-pruned left-then-right DFS, per-test normal recomputation, float4
 * texels, recursive RayTrace.  Every float expression keeps the reference's
 * operand order; build with -ffp-contract=off and without -ffast-math.
 *
 * The only deliberate departure from the reference is the random source: the
 * reference draws from process-global rand() and a global table cursor
 * (Src/Math.h:17-20, Src/Math.cpp:33-40); the oracle draws the SAME sequence of
 * decisions from a counter-based generator keyed by (seed, pixel, sample, draw#)
 * and addresses the unit-vector table by (pass, pixel, sub-sample, read#).
 * oracle/ref_harness.cpp replays exactly these draws into the real reference by
 * interposing rand(), which is how this file is pinned (tests/golden/).
 */
#define _GNU_SOURCE
#include "rt_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

/* ------------------------------------------------------------------------- */
/* RVec3 (Src/RVector.h:100-233)                                              */
/* ------------------------------------------------------------------------- */
typedef struct { float x, y, z; } vec3;
typedef struct { float x, y, z, w; } vec4;

#define FLT_EQUAL_ZERO(a) (fabsf(a) < FLT_EPSILON)   /* Src/MathHelper.h:12 */
#define ORC_PI 3.1415926f                            /* Src/MathHelper.h:13 */

static inline vec3 v3(float x, float y, float z) { vec3 r = { x, y, z }; return r; }
static inline vec3 v3add(vec3 a, vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline vec3 v3sub(vec3 a, vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline vec3 v3muls(vec3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
static inline vec3 v3divs(vec3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
static inline vec3 v3mul(vec3 a, vec3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline float v3dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline vec3 v3cross(vec3 a, vec3 b)
{
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float v3mag(vec3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
/* Src/RVector.h:142-145 : all three components must be non-zero */
static inline int v3_is_nonzero(vec3 a)
{
    return !FLT_EQUAL_ZERO(a.x) && !FLT_EQUAL_ZERO(a.y) && !FLT_EQUAL_ZERO(a.z);
}
/* Src/RVector.h:169-183 */
static inline vec3 v3normalized(vec3 a)
{
    float sqr_mag = a.x * a.x + a.y * a.y + a.z * a.z;
    if (!FLT_EQUAL_ZERO(sqr_mag)) {
        float one_over_mag = 1.0f / sqrtf(sqr_mag);
        a.x *= one_over_mag; a.y *= one_over_mag; a.z *= one_over_mag;
    }
    return a;
}
/* Src/MathHelper.cpp:26-38 */
static inline float q_rsqrt(float number)
{
    const float x2 = number * 0.5F;
    const float threehalfs = 1.5F;
    union { float f; uint32_t i; } conv;
    conv.f = number;
    conv.i = 0x5f3759df - (conv.i >> 1);
    conv.f *= (threehalfs - (x2 * conv.f * conv.f));
    return conv.f;
}
/* Src/RVector.h:185-199 */
static inline vec3 v3normalized_fast(vec3 a)
{
    float sqr_mag = a.x * a.x + a.y * a.y + a.z * a.z;
    if (!FLT_EQUAL_ZERO(sqr_mag)) {
        float one_over_mag = q_rsqrt(sqr_mag);
        a.x *= one_over_mag; a.y *= one_over_mag; a.z *= one_over_mag;
    }
    return a;
}
/* Src/RVector.h:218-221 : *this - normal * 2.0f * Dot(*this, normal) */
static inline vec3 v3reflect(vec3 v, vec3 n)
{
    return v3sub(v, v3muls(v3muls(n, 2.0f), v3dot(v, n)));
}
static inline float lerpf(float a, float b, float t) { return a + (b - a) * t; } /* Src/MathHelper.h:38 */
static inline float minf_ref(float a, float b) { return (a < b) ? a : b; }     /* Src/MathHelper.h:36 */
static inline float maxf_ref(float a, float b) { return (a > b) ? a : b; }     /* Src/MathHelper.h:33 */

/* ------------------------------------------------------------------------- */
/* Counter-based replacement for rand() (Src/Math.h:17-20)                    */
/* ------------------------------------------------------------------------- */
#define ORC_TABLE_SIZE 0xFFFFFFu     /* MaxUnitVectorNums, Src/Math.cpp:17 */
#define ORC_TABLE_STRIDE 16u
#define ORC_TABLE_SEED 0x52544142u

static inline uint32_t mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
static inline uint32_t path_key(uint32_t seed, uint32_t pixel, uint32_t sample)
{
    uint32_t h = mix32(seed ^ 0x9E3779B9u);
    h = mix32(h + pixel);
    h = mix32(h + sample);
    return h;
}
static inline uint32_t key_rand31(uint32_t key, uint32_t counter) { return mix32(key + counter) >> 1; }
uint32_t orc_rand31(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t counter)
{
    return key_rand31(path_key(seed, pixel, sample), counter);
}
/* (float)rand() / RAND_MAX with RAND_MAX = 2^31-1 (-> 2147483648.0f as float) */
static inline float rand31_to_float(uint32_t r) { return (float)(int32_t)r / 2147483648.0f; }

/* RMath::RandomUnitVector (Src/Math.h:34-40) from two explicit uniform draws */
static inline vec3 unit_vector_libm(float r1, float r2)
{
    float t1 = 2.0f * ORC_PI * r1;
    float t2 = acosf(1.0f - 2.0f * r2);
    float sin_t2 = sinf(t2);
    return v3(sinf(t1) * sin_t2, cosf(t1) * sin_t2, cosf(t2));
}
/* same expression with the transcendentals evaluated in double and rounded once:
 * this is what the device computes for the fuzzy-reflection direction */
static inline vec3 unit_vector_f64(float r1, float r2)
{
    float t1 = 2.0f * ORC_PI * r1;
    float t2 = (float)acos((double)(1.0f - 2.0f * r2));
    float sin_t2 = (float)sin((double)t2);
    return v3((float)sin((double)t1) * sin_t2, (float)cos((double)t1) * sin_t2, (float)cos((double)t2));
}
/* entry i of the table InitPseudoRandomUnitVector fills (Src/Math.cpp:24-31) when
 * rand() call j returns draw j of the table stream */
static inline vec3 unit_table_entry(uint32_t index)
{
    uint32_t key = path_key(ORC_TABLE_SEED, 0xFFFFFFFFu, 0xFFFFFFFFu);
    float r1 = rand31_to_float(key_rand31(key, 2u * index));
    float r2 = rand31_to_float(key_rand31(key, 2u * index + 1u));
    return unit_vector_libm(r1, r2);
}
void orc_unit_table_entry(uint32_t index, float out3[3])
{
    vec3 v = unit_table_entry(index);
    out3[0] = v.x; out3[1] = v.y; out3[2] = v.z;
}
static inline uint32_t table_phase(uint32_t seed) { return mix32(seed ^ 0x7AB1E5u) % ORC_TABLE_SIZE; }

/* per-path random state */
typedef struct {
    uint32_t key;          /* path_key(seed, pixel, sample) */
    uint32_t counter;      /* number of rand() draws so far */
    uint64_t table_base;   /* first table index of this path (before modulo) */
    uint32_t table_reads;  /* PseudoRandomUnitVector calls so far */
} path_rng;

static inline float rng_random(path_rng* r) { return rand31_to_float(key_rand31(r->key, r->counter++)); }

/* ------------------------------------------------------------------------- */
/* stats                                                                      */
/* ------------------------------------------------------------------------- */
static __thread orc_stats t_stats;
static orc_stats g_stats;
static pthread_mutex_t g_stats_mutex = PTHREAD_MUTEX_INITIALIZER;
static void stats_flush(void)
{
    pthread_mutex_lock(&g_stats_mutex);
    g_stats.rays += t_stats.rays; g_stats.box_tests += t_stats.box_tests;
    g_stats.tri_tests += t_stats.tri_tests; g_stats.shaded_hits += t_stats.shaded_hits;
    g_stats.tex_samples += t_stats.tex_samples; g_stats.camera_rays += t_stats.camera_rays;
    pthread_mutex_unlock(&g_stats_mutex);
    memset(&t_stats, 0, sizeof t_stats);
}
void orc_stats_reset(void) { memset(&t_stats, 0, sizeof t_stats); memset(&g_stats, 0, sizeof g_stats); }
void orc_stats_get(orc_stats* out) { stats_flush(); *out = g_stats; }

/* ------------------------------------------------------------------------- */
/* RRay / RAabb / RayHitResult (Src/RRay.h:13-37, Src/RAabb.h:12-62)          */
/* ------------------------------------------------------------------------- */
typedef struct { vec3 origin, dir; float distance; } ray_t;
typedef struct { vec3 pmin, pmax; } aabb_t;
typedef struct {
    vec3 pos, normal; float distance; vec3 color; float alpha;
} hit_t;

static inline void hit_init(hit_t* h)   /* Src/RRay.h:15-20 */
{
    memset(h, 0, sizeof *h);
    h->distance = 0.0f; h->color = v3(1.0f, 1.0f, 1.0f); h->alpha = 1.0f;
}
static inline void aabb_init(aabb_t* b) /* Src/RAabb.cpp:13-17 */
{
    b->pmin = v3(FLT_MAX, FLT_MAX, FLT_MAX); b->pmax = v3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
}
static inline void aabb_expand(aabb_t* b, vec3 p) /* Src/RAabb.h:20-28 */
{
    if (p.x < b->pmin.x) b->pmin.x = p.x;
    if (p.y < b->pmin.y) b->pmin.y = p.y;
    if (p.z < b->pmin.z) b->pmin.z = p.z;
    if (p.x > b->pmax.x) b->pmax.x = p.x;
    if (p.y > b->pmax.y) b->pmax.y = p.y;
    if (p.z > b->pmax.z) b->pmax.z = p.z;
}

/* RRay::TestIntersectionWithAabb (Src/RRay.cpp:89-136): slab test of the infinite line */
static int ray_test_aabb(const ray_t* r, const aabb_t* b)
{
    float tmin = -FLT_MAX, tmax = FLT_MAX;
    t_stats.box_tests++;
    if (!FLT_EQUAL_ZERO(r->dir.x)) {
        float inv_x = 1.0f / r->dir.x;
        float tx1 = (b->pmin.x - r->origin.x) * inv_x;
        float tx2 = (b->pmax.x - r->origin.x) * inv_x;
        tmin = maxf_ref(tmin, minf_ref(tx1, tx2));
        tmax = minf_ref(tmax, maxf_ref(tx1, tx2));
    }
    if (!FLT_EQUAL_ZERO(r->dir.y)) {
        float inv_y = 1.0f / r->dir.y;
        float ty1 = (b->pmin.y - r->origin.y) * inv_y;
        float ty2 = (b->pmax.y - r->origin.y) * inv_y;
        tmin = maxf_ref(tmin, minf_ref(ty1, ty2));
        tmax = minf_ref(tmax, maxf_ref(ty1, ty2));
    }
    if (!FLT_EQUAL_ZERO(r->dir.z)) {
        float inv_z = 1.0f / r->dir.z;
        float tz1 = (b->pmin.z - r->origin.z) * inv_z;
        float tz2 = (b->pmax.z - r->origin.z) * inv_z;
        tmin = maxf_ref(tmin, minf_ref(tz1, tz2));
        tmax = minf_ref(tmax, maxf_ref(tz1, tz2));
    }
    return tmax > tmin;
}

/* RRay::TestIntersectionWithTriangleAndFaceNormal (Src/RRay.cpp:147-213) */
static int ray_test_triangle_normal(const ray_t* r, const vec3 tri[3], vec3 normal, hit_t* result)
{
    vec3 end_point = v3add(r->origin, v3muls(r->dir, r->distance));
    vec3 point = tri[0];
    float d0 = v3dot(normal, r->origin);
    float d1 = v3dot(normal, point);
    float d2 = d0 - d1;
    if (d2 < 0) return 0;
    if (v3dot(end_point, normal) - d1 > 0) return 0;
    vec3 l = v3sub(end_point, r->origin);
    float d3 = v3dot(normal, l);
    if (FLT_EQUAL_ZERO(d3)) return 0;
    float df = -(d2 / d3);
    vec3 cp = v3add(r->origin, v3muls(l, df));
    for (int i = 0; i < 3; i++) {
        vec3 edge = v3sub(tri[(i + 1) % 3], tri[i]);
        vec3 edge_normal = v3cross(edge, normal);
        if (v3dot(edge_normal, v3sub(cp, tri[i])) > 0) return 0;
    }
    if (result) {
        result->pos = cp;
        result->normal = normal;
        result->distance = v3mag(v3muls(l, df));
    }
    return 1;
}
/* RRay::TestIntersectionWithTriangle (Src/RRay.cpp:138-145) */
static int ray_test_triangle(const ray_t* r, const vec3 tri[3], hit_t* result)
{
    vec3 p0p1 = v3sub(tri[1], tri[0]);
    vec3 p0p2 = v3sub(tri[2], tri[0]);
    vec3 normal = v3normalized(v3cross(p0p1, p0p2));
    t_stats.tri_tests++;
    return ray_test_triangle_normal(r, tri, normal, result);
}

/* ------------------------------------------------------------------------- */
/* KdTree (Src/KdTree.h:25-95, Src/KdTree.cpp)                                */
/* ------------------------------------------------------------------------- */
typedef struct { int p0, p1, p2, index; } tri_data;
typedef struct kd_node {
    struct kd_node *left, *right;
    tri_data triangle;
    aabb_t bounds;
} kd_node;

/* GetLargestAxisOfBounds (Src/KdTree.cpp:10-35) */
static int largest_axis(const aabb_t* b)
{
    vec3 size = v3sub(b->pmax, b->pmin);
    if (size.x > size.y) return (size.x > size.z) ? 0 : 2;
    return (size.y > size.z) ? 1 : 2;
}

/* KdNode::Build (Src/KdTree.cpp:37-126) */
static kd_node* kd_build(const vec3* points, const tri_data* tris, int n)
{
    kd_node* node = (kd_node*)calloc(1, sizeof(kd_node));
    aabb_init(&node->bounds);
    node->triangle.p0 = node->triangle.p1 = node->triangle.p2 = node->triangle.index = -1;
    for (int i = 0; i < n; i++) {
        aabb_expand(&node->bounds, points[tris[i].p0]);
        aabb_expand(&node->bounds, points[tris[i].p1]);
        aabb_expand(&node->bounds, points[tris[i].p2]);
    }
    if (n == 1) { node->triangle = tris[0]; return node; }

    vec3 mid = v3(0, 0, 0);
    for (int i = 0; i < n; i++) {
        vec3 v0 = points[tris[i].p0], v1 = points[tris[i].p1], v2 = points[tris[i].p2];
        mid = v3add(mid, v3divs(v3add(v3add(v0, v1), v2), 3.0f));
    }
    mid = v3divs(mid, (float)n);

    tri_data* left = (tri_data*)malloc(sizeof(tri_data) * (size_t)n);
    tri_data* right = (tri_data*)malloc(sizeof(tri_data) * (size_t)n);
    int nl = 0, nr = 0;
    const int axis = largest_axis(&node->bounds);
    for (int i = 0; i < n; i++) {
        vec3 v0 = points[tris[i].p0], v1 = points[tris[i].p1], v2 = points[tris[i].p2];
        vec3 c = v3divs(v3add(v3add(v0, v1), v2), 3.0f);
        int to_left = 0;
        switch (axis) {
        case 0: to_left = (c.x < mid.x); break;
        case 1: to_left = (c.y < mid.y); break;
        case 2: to_left = (c.z < mid.z); break;
        }
        if (to_left) left[nl++] = tris[i]; else right[nr++] = tris[i];
    }
    if (nl == n || nr == n) {
        const int half = n / 2;
        memcpy(left, tris, sizeof(tri_data) * (size_t)half); nl = half;
        memcpy(right, tris + half, sizeof(tri_data) * (size_t)(n - half)); nr = n - half;
    }
    if (nl > 0) node->left = kd_build(points, left, nl);
    if (nr > 0) node->right = kd_build(points, right, nr);
    free(left); free(right);
    return node;
}
static void kd_free(kd_node* n) { if (!n) return; kd_free(n->left); kd_free(n->right); free(n); }

/* KdNode::TestRayIntersection (Src/KdTree.cpp:128-195) */
static int kd_test_ray(const kd_node* node, ray_t* test_ray, const vec3* points, hit_t* out, int* tri_index)
{
    if (!ray_test_aabb(test_ray, &node->bounds)) return 0;
    int is_leaf = 1, result = 0;
    if (node->left) { result |= kd_test_ray(node->left, test_ray, points, out, tri_index); is_leaf = 0; }
    if (node->right) { result |= kd_test_ray(node->right, test_ray, points, out, tri_index); is_leaf = 0; }
    if (is_leaf) {
        const vec3 tri[3] = { points[node->triangle.p0], points[node->triangle.p1], points[node->triangle.p2] };
        hit_t h; hit_init(&h);
        if (ray_test_triangle(test_ray, tri, &h)) {
            test_ray->distance = h.distance;
            if (out) *out = h;
            if (tri_index) *tri_index = node->triangle.index;
            return 1;
        }
        return 0;
    }
    return result;
}

/* ------------------------------------------------------------------------- */
/* RTexture (Src/Texture.h, Src/Texture.cpp:23-57, 119-151)                   */
/* ------------------------------------------------------------------------- */
typedef struct { vec4* pixels; int width, height; } texture_t;

static inline float texel_to_linear(uint8_t b) { return powf((float)b / 255, 2.2f); } /* Src/ColorBuffer.h:70-78 */
void orc_texel_lut(float out256[256]) { for (int i = 0; i < 256; i++) out256[i] = texel_to_linear((uint8_t)i); }

static vec4 v4lerp(vec4 a, vec4 b, float t)
{
    vec4 r = { lerpf(a.x, b.x, t), lerpf(a.y, b.y, t), lerpf(a.z, b.z, t), lerpf(a.w, b.w, t) };
    return r;
}
static vec4 texture_sample(const texture_t* t, float u, float v)
{
    t_stats.tex_samples++;
    float cu = u - floorf(u);
    float cv = v - floorf(v);
    float fx = cu * (t->width - 1);
    float fy = cv * (t->height - 1);
    int x0 = (int)floorf(fx), y0 = (int)floorf(fy);
    int x1 = (int)ceilf(fx), y1 = (int)ceilf(fy);
    float dx = fx - x0, dy = fy - y0;
    const vec4* p = t->pixels; const int w = t->width;
    return v4lerp(v4lerp(p[y0 * w + x0], p[y0 * w + x1], dx),
                  v4lerp(p[y1 * w + x0], p[y1 * w + x1], dx), dy);
}

/* ------------------------------------------------------------------------- */
/* RMeshShape (Src/MeshShape.h, Src/MeshShape.cpp)                            */
/* ------------------------------------------------------------------------- */
#define ORC_MAX_MATERIALS 64
typedef struct {
    vec3 *points, *texcoords, *normals;
    int n_points, n_texcoords, n_normals;
    int *point_idx, *texcoord_idx, *normal_idx;   /* 3 per triangle */
    int *poly_material;                           /* per triangle */
    int n_tris;
    int n_material_names;
    char* material_names[ORC_MAX_MATERIALS];
    char* texture_paths[ORC_MAX_MATERIALS];
    texture_t** textures;   /* sized by triangle count, indexed by material id (Src/MeshShape.cpp:220,266) */
    int n_textures;
    kd_node* root;
    int n_nodes;
} mesh_t;

typedef struct {
    orc_material_node nodes[64];
    int n;
} material_t;

typedef struct {
    aabb_t aabb;            /* RShape::Aabb */
    int has_culling_bounds;
    int kind;               /* ORC_SHAPE_* */
    vec3 c;                 /* triangle: a, b, c = Points[0..2] (Src/Shapes.h:106-130) */
    vec3 a, b;              /* sphere: a = Center; plane: a = Normal, b = Point; capsule: a = Start, b = End (Src/Shapes.h:46-112) */
    float radius;
    mesh_t* mesh;
    material_t material;
    int has_material;
} shape_t;

struct orc_scene {
    shape_t* shapes; int n_shapes;
    int unitvec_mode;
    int combine_b_first;
};

#define PUSH(arr, n, cap, val) do { if ((n) == (cap)) { (cap) = (cap) ? (cap) * 2 : 1024; (arr) = realloc((arr), sizeof(*(arr)) * (size_t)(cap)); } (arr)[(n)++] = (val); } while (0)

/* GetNthNumericValue (Src/MeshShape.cpp:38-48): (n+1)-th integer of "a/b/c" */
static int nth_numeric(int n, const char* tok)
{
    int value = -1; int failed = 0;
    const char* p = tok;
    for (int i = 0; i <= n; i++) {
        if (failed) break;
        while (*p == ' ' || *p == '\t') p++;
        char* e; long v = strtol(p, &e, 10);
        if (e == p) { value = 0; failed = 1; break; }
        value = (int)v; p = e;
        if (*p) p++; else failed = 1;     /* >> dummy */
    }
    return value;
}
/* Split(Line, ' ') (Src/MeshShape.cpp:50-62): getline semantics */
static int split_spaces(char* line, char** toks, int max)
{
    int n = 0; char* p = line;
    if (!*p) return 0;
    for (;;) {
        if (n < max) toks[n] = p;
        n++;
        char* s = strchr(p, ' ');
        if (!s) break;
        *s = 0; p = s + 1;
        if (!*p) break;                   /* trailing delimiter: no empty last token */
    }
    return n < max ? n : max;
}
static const char* skip_token(const char* p)
{
    while (*p == ' ' || *p == '\t') p++;
    while (*p && *p != ' ' && *p != '\t') p++;
    return p;
}
static float next_float(const char** pp)
{
    char* e; float v = strtof(*pp, &e);
    if (e == *pp) return 0.0f;
    *pp = e; return v;
}

static FILE* open_with_fallback(const char* filename, char* resolved, size_t n)
{
    /* Src/MeshShape.cpp:67-83 */
    snprintf(resolved, n, "%s", filename);
    FILE* f = fopen(resolved, "rb");
    for (int i = 0; !f && i < 2; i++) {
        char tmp[4096]; snprintf(tmp, sizeof tmp, "../%s", resolved);
        snprintf(resolved, n, "%s", tmp);
        f = fopen(resolved, "rb");
    }
    return f;
}

static mesh_t* mesh_load(const char* filename, aabb_t* shape_aabb)
{
    char mesh_filename[4096];
    FILE* f = open_with_fallback(filename, mesh_filename, sizeof mesh_filename);
    if (!f) { fprintf(stderr, "orc: unable to open %s\n", filename); return NULL; }
    mesh_t* m = (mesh_t*)calloc(1, sizeof(mesh_t));
    int cp = 0, ct = 0, cn = 0, cpi = 0, cti = 0, cni = 0, cpm = 0;
    int npi = 0, nti = 0, nni = 0, npm = 0;
    int current_material = -1;
    char* line = NULL; size_t cap = 0; ssize_t len;
    while ((len = getline(&line, &cap, f)) >= 0) {
        if (len > 0 && line[len - 1] == '\n') line[--len] = 0;
        char* sp = strchr(line, ' ');
        size_t klen = sp ? (size_t)(sp - line) : (size_t)len;
        if (klen == 1 && line[0] == 'v') {
            const char* p = skip_token(line);
            float a = next_float(&p), b = next_float(&p), c = next_float(&p);
            vec3 v = v3(a, b, c);
            PUSH(m->points, m->n_points, cp, v);
            aabb_expand(shape_aabb, v);
        } else if (klen == 2 && line[0] == 'v' && line[1] == 't') {
            const char* p = skip_token(line);
            float a = next_float(&p), b = next_float(&p);
            PUSH(m->texcoords, m->n_texcoords, ct, v3(a, b, 0.0f));
        } else if (klen == 2 && line[0] == 'v' && line[1] == 'n') {
            const char* p = skip_token(line);
            float a = next_float(&p), b = next_float(&p), c = next_float(&p);
            PUSH(m->normals, m->n_normals, cn, v3(a, b, c));
        } else if (klen == 1 && line[0] == 'f') {
            char* toks[16];
            int nt = split_spaces(line, toks, 16);
            int nverts = nt - 1;
            static const int tri_idx[3] = { 0, 1, 2 };
            static const int quad_idx[6] = { 0, 1, 2, 0, 2, 3 };
            const int* poly = NULL; int npoly = 0;
            if (nverts == 3) { poly = tri_idx; npoly = 3; }
            else if (nverts == 4) { poly = quad_idx; npoly = 6; }
            for (int i = 0; i < npoly; i++) {
                const char* tok = toks[poly[i] + 1];
                PUSH(m->point_idx, npi, cpi, nth_numeric(0, tok) - 1);
                PUSH(m->texcoord_idx, nti, cti, nth_numeric(1, tok) - 1);
                PUSH(m->normal_idx, nni, cni, nth_numeric(2, tok) - 1);
                if (i % 3 == 0) PUSH(m->poly_material, npm, cpm, current_material);
            }
        } else if (klen == 6 && !strncmp(line, "usemtl", 6)) {
            char* toks[8];
            int nt = split_spaces(line, toks, 8);
            const char* name = nt > 1 ? toks[1] : "";
            int found = -1;
            for (int i = 0; i < m->n_material_names; i++) if (!strcmp(m->material_names[i], name)) { found = i; break; }
            if (found < 0 && m->n_material_names < ORC_MAX_MATERIALS) {
                m->material_names[m->n_material_names] = strdup(name);
                found = m->n_material_names++;
            }
            current_material = found;
        }
    }
    fclose(f);
    m->n_tris = npi / 3;

    /* validate indices: the reference would read out of bounds, the oracle refuses */
    for (int i = 0; i < npi; i++) {
        if (m->point_idx[i] < 0 || m->point_idx[i] >= m->n_points ||
            m->normal_idx[i] < 0 || m->normal_idx[i] >= m->n_normals ||
            m->texcoord_idx[i] < 0 || m->texcoord_idx[i] >= m->n_texcoords) {
            fprintf(stderr, "orc: index out of range in %s (vertex %d)\n", filename, i);
            free(line); return NULL;
        }
    }

    /* .mtl (Src/MeshShape.cpp:202-272) */
    char mtl[4096]; snprintf(mtl, sizeof mtl, "%s", mesh_filename);
    char* ext = strstr(mtl, ".obj");
    if (ext) {
        memcpy(ext, ".mtl", 4);
        FILE* mf = fopen(mtl, "rb");
        if (mf) {
            char base[4096] = "";
            char* s1 = strrchr(mtl, '/'); char* s2 = strrchr(mtl, '\\');
            char* s = s1 > s2 ? s1 : s2;
            if (s) { size_t bl = (size_t)(s - mtl) + 1; memcpy(base, mtl, bl); base[bl] = 0; }
            m->n_textures = m->n_tris;
            m->textures = (texture_t**)calloc((size_t)(m->n_textures > 0 ? m->n_textures : 1), sizeof(texture_t*));
            current_material = -1;
            while ((len = getline(&line, &cap, mf)) >= 0) {
                if (len > 0 && line[len - 1] == '\n') line[--len] = 0;
                char* sp = strchr(line, ' ');
                size_t klen = sp ? (size_t)(sp - line) : (size_t)len;
                if (klen == 6 && !strncmp(line, "newmtl", 6)) {
                    char name[1024] = ""; sscanf(skip_token(line), "%1023s", name);
                    current_material = -1;
                    for (int i = 0; i < m->n_material_names; i++) if (!strcmp(m->material_names[i], name)) { current_material = i; break; }
                } else if (klen == 6 && !strncmp(line, "map_Kd", 6)) {
                    if (current_material == -1) continue;
                    char rel[2048] = ""; sscanf(skip_token(line), "%2047s", rel);
                    char path[8192]; snprintf(path, sizeof path, "%s%s", base, rel);
                    char* bs;
                    while ((bs = strstr(path, "\\\\")) != NULL) { *bs = '/'; memmove(bs + 1, bs + 2, strlen(bs + 2) + 1); }
                    free(m->texture_paths[current_material]);
                    m->texture_paths[current_material] = strdup(path);
                }
            }
            fclose(mf);
        }
    }
    free(line);

    /* KdTree::Build (Src/KdTree.cpp:202-220) */
    if (m->n_tris > 0) {
        tri_data* td = (tri_data*)malloc(sizeof(tri_data) * (size_t)m->n_tris);
        for (int i = 0; i < m->n_tris; i++) {
            td[i].p0 = m->point_idx[i * 3]; td[i].p1 = m->point_idx[i * 3 + 1]; td[i].p2 = m->point_idx[i * 3 + 2];
            td[i].index = i;
        }
        m->root = kd_build(m->points, td, m->n_tris);
        free(td);
        m->n_nodes = 2 * m->n_tris - 1;
    }
    return m;
}

/* RMath::Barycentric (Src/Math.cpp:56-68) */
static void barycentric(vec3 p, vec3 a, vec3 b, vec3 c, float* u, float* v, float* w)
{
    vec3 v0 = v3sub(b, a), v1 = v3sub(c, a), v2 = v3sub(p, a);
    float d00 = v3dot(v0, v0), d01 = v3dot(v0, v1), d11 = v3dot(v1, v1);
    float d20 = v3dot(v2, v0), d21 = v3dot(v2, v1);
    float denom = d00 * d11 - d01 * d01;
    *v = (d11 * d20 - d01 * d21) / denom;
    *w = (d00 * d21 - d01 * d20) / denom;
    *u = 1.0f - *v - *w;
}

/* RMeshShape::TestRayIntersection (Src/MeshShape.cpp:280-332) */
static int mesh_test_ray(const mesh_t* m, const ray_t* in_ray, hit_t* out, int* tri_out)
{
    if (!m->root) return 0;
    int tri = -1;
    ray_t test_ray = *in_ray;                    /* KdTree::TestRayIntersection copies the ray (Src/KdTree.cpp:229) */
    if (!kd_test_ray(m->root, &test_ray, m->points, out, &tri)) return 0;
    if (tri_out) *tri_out = tri;
    if (out) {
        t_stats.shaded_hits++;
        vec3 p = out->pos;
        int v0 = tri * 3, v1 = tri * 3 + 1, v2 = tri * 3 + 2;
        vec3 a = m->points[m->point_idx[v0]], b = m->points[m->point_idx[v1]], c = m->points[m->point_idx[v2]];
        float u, v, w;
        barycentric(p, a, b, c, &u, &v, &w);
        vec3 n0 = m->normals[m->normal_idx[v0]], n1 = m->normals[m->normal_idx[v1]], n2 = m->normals[m->normal_idx[v2]];
        out->normal = v3normalized_fast(v3add(v3add(v3muls(n0, u), v3muls(n1, v)), v3muls(n2, w)));
        int mat = m->poly_material[tri];
        if (mat != -1 && mat < m->n_textures) {
            const texture_t* tex = m->textures[mat];
            if (tex) {
                vec3 t0 = m->texcoords[m->texcoord_idx[v0]], t1 = m->texcoords[m->texcoord_idx[v1]], t2 = m->texcoords[m->texcoord_idx[v2]];
                vec3 tc = v3add(v3add(v3muls(t0, u), v3muls(t1, v)), v3muls(t2, w));
                vec4 s = texture_sample(tex, tc.x, 1.0f - tc.y);
                out->color = v3(s.x, s.y, s.z);
                out->alpha = s.w;
            }
        }
    }
    return 1;
}

/* ------------------------------------------------------------------------- */
/* RayTracerScene (Src/RayTracerScene.cpp)                                    */
/* ------------------------------------------------------------------------- */
static const float BounceRayStartOffset = 0.0001f;   /* Src/SurfaceMaterials.cpp:13 */

/* RAabb::ExpandBySphere (Src/RAabb.h:46-54) */
static inline void aabb_expand_by_sphere(aabb_t* b, vec3 c, float r)
{
    if (c.x - r < b->pmin.x) b->pmin.x = c.x - r;
    if (c.y - r < b->pmin.y) b->pmin.y = c.y - r;
    if (c.z - r < b->pmin.z) b->pmin.z = c.z - r;
    if (c.x + r > b->pmax.x) b->pmax.x = c.x + r;
    if (c.y + r > b->pmax.y) b->pmax.y = c.y + r;
    if (c.z + r > b->pmax.z) b->pmax.z = c.z + r;
}

/* RRay::TestIntersectionWithSphere (Src/RRay.cpp:25-66): quadratic in the parameter of Origin + t * (Direction * Distance).
 * Writes position, normal and distance only -- the sampled colour / alpha of `result` stay what they were. */
static int ray_test_sphere(const ray_t* r, vec3 c, float radius, hit_t* result)
{
    float dx = r->dir.x * r->distance;
    float dy = r->dir.y * r->distance;
    float dz = r->dir.z * r->distance;
    vec3 o = r->origin;
    float qa = dx * dx + dy * dy + dz * dz;
    float qb = 2 * dx * (o.x - c.x) + 2 * dy * (o.y - c.y) + 2 * dz * (o.z - c.z);
    float qc = c.x * c.x + c.y * c.y + c.z * c.z + o.x * o.x + o.y * o.y + o.z * o.z +
               -2 * (c.x * o.x + c.y * o.y + c.z * o.z) - radius * radius;
    float d = qb * qb - 4 * qa * qc;
    if (d >= 0) {
        float t = (-qb - sqrtf(d)) / (qa * 2);
        if (t <= 0) return 0;
        vec3 hp = v3(o.x + t * dx, o.y + t * dy, o.z + t * dz);
        float dist = v3mag(v3sub(hp, o));
        if (dist > r->distance) return 0;
        if (result) {
            result->pos = hp;
            result->normal = v3normalized(v3sub(hp, c));
            result->distance = dist;
        }
        return 1;
    }
    return 0;
}

/* RRay::TestIntersectionWithPlane (Src/RRay.cpp:68-87); `fabsf(denom) > 1e-6` compares against a double constant */
static int ray_test_plane(const ray_t* r, vec3 n, vec3 p, hit_t* result)
{
    float denom = v3dot(n, r->dir);
    if ((double)fabsf(denom) > 1e-6) {
        vec3 p0l0 = v3sub(p, r->origin);
        float t = v3dot(p0l0, n) / denom;
        if (t >= 0 && t < r->distance) {
            if (result) {
                result->pos = v3add(r->origin, v3muls(r->dir, t));
                result->normal = n;
                result->distance = t;
            }
            return 1;
        }
    }
    return 0;
}

/* RCapsule::TestRayCylinderIntersection (Src/Shapes.cpp:64-125); `fabs(a) < FLT_EPSILON` is the double overload on a float */
static int ray_test_cylinder(const ray_t* r, vec3 start, vec3 end, float radius, hit_t* result)
{
    vec3 d = v3sub(end, start);
    vec3 m = v3sub(r->origin, start);
    float dd = v3dot(d, d);
    float nd = v3dot(r->dir, d);
    float mn = v3dot(m, r->dir);
    float md = v3dot(m, d);
    float mm = v3dot(m, m);
    if (v3dot(v3sub(r->origin, start), v3sub(end, start)) < 0 && v3dot(r->dir, v3sub(end, start)) < 0) return 0;
    if (v3dot(v3sub(r->origin, end), v3sub(start, end)) < 0 && v3dot(r->dir, v3sub(start, end)) < 0) return 0;
    float a = dd - nd * nd;
    float b = dd * mn - nd * md;
    float c = dd * (mm - radius * radius) - md * md;
    if (fabs((double)a) < (double)FLT_EPSILON) return 0;
    if ((b * b - a * c) < 0) return 0;
    float rt = (-b - sqrtf(b * b - a * c)) / a;
    if (rt < 0) return 0;
    vec3 v = v3add(r->origin, v3muls(r->dir, rt));
    if (v3dot(v3sub(v, start), v3sub(end, start)) < 0) return 0;
    if (v3dot(v3sub(v, end), v3sub(start, end)) < 0) return 0;
    if (result) {
        result->distance = rt;
        result->pos = v3add(r->origin, v3muls(r->dir, rt));
        vec3 side = v3cross(v3sub(end, start), v3sub(result->pos, start));
        result->normal = v3normalized(v3cross(side, v3sub(end, start)));
    }
    return 1;
}

/* RCapsule::TestRayIntersection (Src/Shapes.cpp:34-62): the side first; failing that the two end spheres into fresh
 * RayHitResults, the nearer (the second on a tie) copied WHOLE into *OutResult -- which resets its sampled colour / alpha. */
static int ray_test_capsule(const ray_t* r, vec3 start, vec3 end, float radius, hit_t* result)
{
    if (!ray_test_cylinder(r, start, end, radius, result)) {
        hit_t r1, r2; hit_init(&r1); hit_init(&r2);
        int b1 = ray_test_sphere(r, start, radius, &r1);
        int b2 = ray_test_sphere(r, end, radius, &r2);
        if (result) {
            if (b1 && b2) *result = r1.distance < r2.distance ? r1 : r2;
            else if (b1) *result = r1;
            else if (b2) *result = r2;
        }
        return b1 || b2;
    }
    return 1;
}

static int shape_test_ray(const shape_t* s, const ray_t* r, hit_t* out, int* tri_out)
{
    switch (s->kind) {
    case ORC_SHAPE_MESH: return s->mesh ? mesh_test_ray(s->mesh, r, out, tri_out) : 0;
    case ORC_SHAPE_SPHERE: if (!ray_test_sphere(r, s->a, s->radius, out)) return 0; break;
    case ORC_SHAPE_PLANE: if (!ray_test_plane(r, s->a, s->b, out)) return 0; break;
    case ORC_SHAPE_CAPSULE: if (!ray_test_capsule(r, s->a, s->b, s->radius, out)) return 0; break;
    case ORC_SHAPE_TRIANGLE: { vec3 t[3] = { s->a, s->b, s->c }; if (!ray_test_triangle(r, t, out)) return 0; break; }   /* RTriangle::TestRayIntersection (Src/Shapes.cpp:127-130) */
    default: return 0;
    }
    if (tri_out) *tri_out = -1;     /* (this API's convention: no triangle index for an analytic shape) */
    return 1;
}

/* FindIntersectionWithScene (Src/RayTracerScene.cpp:99-125).  ONE RayHitResult serves all shapes: a sphere, plane or capsule
 * side that hits after a textured mesh did keeps the mesh hit's sampled colour and alpha. */
static int find_intersection(const orc_scene* sc, ray_t test_ray, hit_t* out, int* tri_out)
{
    int hit_shape = -1;
    t_stats.rays++;
    for (int i = 0; i < sc->n_shapes; i++) {
        const shape_t* s = &sc->shapes[i];
        if (!s->has_culling_bounds || ray_test_aabb(&test_ray, &s->aabb)) {
            int hit = shape_test_ray(s, &test_ray, out, tri_out);
            if (hit) { test_ray.distance = out->distance; hit_shape = i; }
        }
    }
    return hit_shape;
}

typedef struct { vec3 attenuation, emissive; } bounce_t;

static vec3 pseudo_random_unit_vector(path_rng* rng, uint32_t phase)
{
    /* RMath::PseudoRandomUnitVector (Src/Math.cpp:33-40) with a per-path cursor */
    uint64_t idx = (rng->table_base + rng->table_reads + phase) % ORC_TABLE_SIZE;
    rng->table_reads++;
    return unit_table_entry((uint32_t)idx);
}
/* RMath::RandomHemisphereDirection (Src/Math.cpp:42-54) */
static vec3 random_hemisphere_direction(vec3 normal, path_rng* rng, uint32_t phase)
{
    vec3 v = pseudo_random_unit_vector(rng, phase);
    if (v3dot(v, normal) > 0.0f) return v;
    return v3reflect(v, normal);
}

typedef struct { const orc_scene* sc; path_rng* rng; uint32_t phase; } eval_ctx;

/* SurfaceMaterial_DiffuseChecker::IsBrighterArea (Src/SurfaceMaterials.cpp:69-92) */
static int checker_brighter(vec3 p, float recip)
{
    int r = 0;
    float fx = p.x * recip, fy = p.y * recip, fz = p.z * recip;
    if (fx - floorf(fx) > 0.5f) r = !r;
    if (fz - floorf(fz) > 0.5f) r = !r;
    if (fy - floorf(fy) > 0.5f) r = !r;
    return r;
}
static float checker_recip(float size) { return FLT_EQUAL_ZERO(size) ? 1.0f : 1.0f / size; } /* :42-50 */

/* ISurfaceMaterial::BounceViewRay (Src/SurfaceMaterials.cpp:20-187) */
static bounce_t material_bounce(const eval_ctx* cx, const material_t* mt, int node, const ray_t* in, const hit_t* hit, ray_t* out)
{
    const orc_material_node* n = &mt->nodes[node];
    bounce_t r; r.attenuation = v3(0, 0, 0); r.emissive = v3(0, 0, 0);
    vec3 albedo = v3(n->r, n->g, n->b);
    switch (n->type) {
    case ORC_MAT_DIFFUSE:
    case ORC_MAT_DIFFUSE_CHECKER: {
        float factor = 1.0f;
        if (n->type == ORC_MAT_DIFFUSE_CHECKER)
            factor = checker_brighter(hit->pos, checker_recip(n->param)) ? 1.0f : 0.5f;
        float ray_distance = in->distance - hit->distance;
        vec3 dir = random_hemisphere_direction(hit->normal, cx->rng, cx->phase);
        out->origin = v3add(hit->pos, v3muls(dir, BounceRayStartOffset));
        out->dir = dir; out->distance = ray_distance;
        float d = maxf_ref(0.0f, v3dot(hit->normal, dir));
        r.attenuation = v3muls(albedo, d);
        if (n->type == ORC_MAT_DIFFUSE_CHECKER) r.attenuation = v3muls(r.attenuation, factor);
        break;
    }
    case ORC_MAT_REFLECTIVE: {
        float ray_distance = in->distance - hit->distance;
        vec3 nd = v3reflect(in->dir, hit->normal);
        if (n->param > 0.0f) {
            float r1 = rng_random(cx->rng);
            float r2 = rng_random(cx->rng);
            vec3 uv = cx->sc->unitvec_mode == ORC_UNITVEC_F64 ? unit_vector_f64(r1, r2) : unit_vector_libm(r1, r2);
            nd = v3add(nd, v3muls(uv, n->param));
            nd = v3normalized(nd);
        }
        out->origin = v3add(hit->pos, v3muls(nd, BounceRayStartOffset));
        out->dir = nd; out->distance = ray_distance;
        r.attenuation = albedo;
        break;
    }
    case ORC_MAT_EMISSIVE:
        *out = *in;
        r.emissive = albedo;
        break;
    case ORC_MAT_BLEND: {
        float bf = n->param < 0.0f ? 0.0f : (n->param > 1.0f ? 1.0f : n->param);   /* RMath::Clamp, :145 */
        int child = rng_random(cx->rng) > bf ? n->child_a : n->child_b;
        return material_bounce(cx, mt, child, in, hit, out);
    }
    case ORC_MAT_COMBINE: {
        /* operand evaluation order of A(...) + B(...) is unspecified (:169-172) */
        bounce_t a, b;
        if (cx->sc->combine_b_first) {
            b = material_bounce(cx, mt, n->child_b, in, hit, out);
            a = material_bounce(cx, mt, n->child_a, in, hit, out);
        } else {
            a = material_bounce(cx, mt, n->child_a, in, hit, out);
            b = material_bounce(cx, mt, n->child_b, in, hit, out);
        }
        r.attenuation = v3add(a.attenuation, b.attenuation);
        r.emissive = v3add(a.emissive, b.emissive);
        break;
    }
    case ORC_MAT_NULL: {
        float ray_distance = in->distance - hit->distance;
        out->origin = v3add(hit->pos, v3muls(in->dir, BounceRayStartOffset));
        out->dir = in->dir; out->distance = ray_distance;
        r.attenuation = v3(1, 1, 1);
        break;
    }
    }
    return r;
}

/* ISurfaceMaterial::PreviewColor */
static vec3 material_preview(const eval_ctx* cx, const material_t* mt, int node, const hit_t* hit)
{
    const orc_material_node* n = &mt->nodes[node];
    vec3 albedo = v3(n->r, n->g, n->b);
    switch (n->type) {
    case ORC_MAT_DIFFUSE:
        return v3muls(albedo, v3dot(hit->normal, v3(0, 1, 0)) * 0.5f + 0.5f);
    case ORC_MAT_DIFFUSE_CHECKER: {
        float factor = checker_brighter(hit->pos, checker_recip(n->param)) ? 1.0f : 0.5f;
        return v3muls(v3muls(albedo, v3dot(hit->normal, v3(0, 1, 0)) * 0.5f + 0.5f), factor);
    }
    case ORC_MAT_REFLECTIVE: return albedo;
    case ORC_MAT_EMISSIVE: return albedo;
    case ORC_MAT_BLEND: {
        float bf = n->param < 0.0f ? 0.0f : (n->param > 1.0f ? 1.0f : n->param);
        int child = rng_random(cx->rng) > bf ? n->child_a : n->child_b;
        return material_preview(cx, mt, child, hit);
    }
    case ORC_MAT_COMBINE: {
        vec3 a, b;
        if (cx->sc->combine_b_first) { b = material_preview(cx, mt, n->child_b, hit); a = material_preview(cx, mt, n->child_a, hit); }
        else { a = material_preview(cx, mt, n->child_a, hit); b = material_preview(cx, mt, n->child_b, hit); }
        return v3add(a, b);
    }
    default: return v3(0, 0, 0);
    }
}

/* Test instrumentation: a per-pixel signature of the hit / miss HISTORY of the pixel's paths -- every scene query folds (bounce level, shape hit
 * or -1, triangle) into the word of the pixel being rendered.  Two renders whose signatures agree in a pixel took the same branches there. */
static uint32_t* g_history = NULL;
static __thread uint32_t* t_history_word = NULL;
void orc_set_history_buffer(uint32_t* buf) { g_history = buf; }
static inline void history_note(int level, int shape, int tri)
{
    if (t_history_word) *t_history_word = mix32(*t_history_word ^ (uint32_t)(level * 0x9E3779B1u) ^ ((uint32_t)(shape + 1) << 24) ^ (uint32_t)(tri + 1));
}

/* RayTracerScene::RayTrace (Src/RayTracerScene.cpp:31-97) */
static vec3 ray_trace(const orc_scene* sc, const ray_t* in_ray, int max_bounce, int use_base_color, path_rng* rng, uint32_t phase)
{
    if (max_bounce == 0) return v3(0, 0, 0);
    vec3 final_color = v3(0, 0, 0);
    hit_t result; hit_init(&result);
    int hit_tri = -1;
    int hit_shape = find_intersection(sc, *in_ray, &result, t_history_word ? &hit_tri : NULL);
    history_note(max_bounce, hit_shape, hit_shape != -1 ? hit_tri : -1);
    if (hit_shape != -1) {
        const shape_t* s = &sc->shapes[hit_shape];
        eval_ctx cx = { sc, rng, phase };
        if (use_base_color) {
            if (s->has_material)
                final_color = v3add(final_color, v3mul(material_preview(&cx, &s->material, 0, &result), result.color));
        } else if (s->has_material) {
            ray_t out_ray; memset(&out_ray, 0, sizeof out_ray);
            bounce_t b = material_bounce(&cx, &s->material, 0, in_ray, &result, &out_ray);
            if (rng_random(rng) <= result.alpha) {
                if (v3_is_nonzero(b.attenuation)) {
                    vec3 child = ray_trace(sc, &out_ray, max_bounce - 1, use_base_color, rng, phase);
                    final_color = v3add(final_color, v3mul(v3mul(b.attenuation, child), result.color));
                }
                final_color = v3add(final_color, b.emissive);
            } else {
                float ray_distance = in_ray->distance - result.distance;
                out_ray.origin = v3add(result.pos, v3muls(in_ray->dir, BounceRayStartOffset));
                out_ray.dir = in_ray->dir; out_ray.distance = ray_distance;
                final_color = v3add(final_color, ray_trace(sc, &out_ray, max_bounce - 1, use_base_color, rng, phase));
            }
        }
    } else {
        float t = 0.5f * (in_ray->dir.y + 1.0f);
        return v3add(v3muls(v3(1.0f, 1.0f, 1.0f), 1.0f - t), v3muls(v3(0.5f, 0.7f, 1.0f), t));
    }
    return final_color;
}

/* ------------------------------------------------------------------------- */
/* ColorBuffer.h / accumulate (Src/ColorBuffer.h:70-109, RayTracerProgram.cpp:51-77) */
/* ------------------------------------------------------------------------- */
struct orc_framebuffer {
    int width, height;
    vec3* accum; int* num; uint32_t* bitcolor;
};
static inline vec3 linear_to_gamma(vec3 c)
{
    static const float exponent = 1.0f / 2.2f;
    return v3(powf(c.x, exponent), powf(c.y, exponent), powf(c.z, exponent));
}
static inline uint32_t make_pixel_color(vec3 c)
{
    int r = (int)(minf_ref(maxf_ref(c.x, 0.0f), 1.0f) * 255);
    int g = (int)(minf_ref(maxf_ref(c.y, 0.0f), 1.0f) * 255);
    int b = (int)(minf_ref(maxf_ref(c.z, 0.0f), 1.0f) * 255);
    return (uint32_t)(((uint32_t)255 << 24) | ((uint32_t)(uint8_t)r << 16) | ((uint32_t)(uint8_t)g << 8) | (uint32_t)(uint8_t)b);
}
/* smallest float c with int(powf(c, 1/2.2f) * 255) >= k, for k = 0..255 */
void orc_gamma_thresholds(float out256[256])
{
    const float exponent = 1.0f / 2.2f;
    out256[0] = 0.0f;
    for (int k = 1; k < 256; k++) {
        union { float f; uint32_t i; } lo, hi, mid;
        lo.f = 0.0f; hi.f = 1.0f;      /* invariant: q(lo) < k <= q(hi) */
        while (hi.i - lo.i > 1) {
            mid.i = lo.i + (hi.i - lo.i) / 2;
            int q = (int)(powf(mid.f, exponent) * 255);
            if (q >= k) hi = mid; else lo = mid;
        }
        out256[k] = hi.f;
    }
}

orc_framebuffer* orc_framebuffer_create(int width, int height)
{
    orc_framebuffer* fb = (orc_framebuffer*)calloc(1, sizeof *fb);
    fb->width = width; fb->height = height;
    size_t n = (size_t)width * (size_t)height;
    fb->accum = (vec3*)calloc(n, sizeof(vec3));
    fb->num = (int*)calloc(n, sizeof(int));
    fb->bitcolor = (uint32_t*)calloc(n, sizeof(uint32_t));
    return fb;
}
void orc_framebuffer_destroy(orc_framebuffer* fb)
{
    if (!fb) return;
    free(fb->accum); free(fb->num); free(fb->bitcolor); free(fb);
}
void orc_framebuffer_clear(orc_framebuffer* fb)
{
    size_t n = (size_t)fb->width * (size_t)fb->height;
    memset(fb->accum, 0, n * sizeof(vec3)); memset(fb->num, 0, n * sizeof(int)); memset(fb->bitcolor, 0, n * 4);
}
int orc_framebuffer_read(const orc_framebuffer* fb, float* accum4, uint32_t* argb)
{
    size_t n = (size_t)fb->width * (size_t)fb->height;
    if (accum4) for (size_t i = 0; i < n; i++) {
        accum4[i * 4] = fb->accum[i].x; accum4[i * 4 + 1] = fb->accum[i].y; accum4[i * 4 + 2] = fb->accum[i].z;
        accum4[i * 4 + 3] = (float)fb->num[i];
    }
    if (argb) memcpy(argb, fb->bitcolor, n * 4);
    return 0;
}

/* camera ray of sub-sample i of a pixel (Src/RayTracerProgram.cpp:133-165) */
static ray_t camera_ray(int width, int height, int pixel_index, int i, path_rng* rng)
{
    const vec3 view_point = v3(0, 0, 7.0f);
    const float aspect = (float)width / (float)height;
    int x = pixel_index % width, y = pixel_index / width;
    float dx = -(float)(x - width / 2) / (width * 2) * aspect;
    float dy = -(float)(y - height / 2) / (height * 2);
    const float inv_pixel_radius = 1.0f / (width * 4);
    const float ox[4] = { 0.0f, inv_pixel_radius, 0.0f, inv_pixel_radius };
    const float oy[4] = { 0.0f, 0.0f, inv_pixel_radius, inv_pixel_radius };
    const float offset_radius = inv_pixel_radius * 0.5f;
    float offset_x = ox[i], offset_y = oy[i];
    offset_x += (rng_random(rng) - 0.5f) * offset_radius;
    offset_y += (rng_random(rng) - 0.5f) * offset_radius;
    ray_t r;
    r.origin = view_point;
    r.dir = v3normalized(v3(dx + offset_x, dy + offset_y, -0.5f));
    r.distance = 1000.0f;
    return r;
}

static void path_rng_init(path_rng* rng, uint32_t seed, int npix, int pixel, int pass, int sub)
{
    rng->key = path_key(seed, (uint32_t)pixel, (uint32_t)(pass * 4 + sub));
    rng->counter = 0;
    rng->table_base = (((uint64_t)pass * (uint64_t)npix + (uint64_t)pixel) * 4u + (uint64_t)sub) * ORC_TABLE_STRIDE;
    rng->table_reads = 0;
}

/* ThreadWorker_Render (Src/RayTracerProgram.cpp:131-188) */
int orc_render_range(const orc_scene* sc, orc_framebuffer* fb, int begin, int end, int max_bounce,
                     int use_base_color, int pass_index, int ns, uint32_t seed)
{
    const int w = fb->width, h = fb->height, npix = w * h;
    if (ns < 1 || ns > 4 || begin < 0 || end >= npix) return -1;
    const uint32_t phase = table_phase(seed);
    for (int p = begin; p <= end; p++) {
        vec3 c = v3(0, 0, 0);
        t_history_word = g_history ? &g_history[p] : NULL;
        for (int i = 0; i < ns; i++) {
            path_rng rng; path_rng_init(&rng, seed, npix, p, pass_index, i);
            ray_t ray = camera_ray(w, h, p, i, &rng);
            t_stats.camera_rays++;
            c = v3add(c, ray_trace(sc, &ray, max_bounce, use_base_color, &rng, phase));
        }
        c = v3divs(c, (float)ns);
        if (use_base_color) {
            fb->bitcolor[p] = make_pixel_color(linear_to_gamma(c));
        } else {
            fb->accum[p] = v3add(fb->accum[p], c);
            fb->num[p]++;
            fb->bitcolor[p] = make_pixel_color(linear_to_gamma(v3divs(fb->accum[p], (float)fb->num[p])));
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* ThreadTaskQueue-style pool (Src/ThreadTaskQueue.h, RayTracerProgram.cpp:190-327) */
/* ------------------------------------------------------------------------- */
typedef struct { int start, end; } pool_task;
typedef struct {
    pool_task* tasks; int n_tasks, next, done;
    pthread_mutex_t mutex; pthread_cond_t done_cond;
    const orc_scene* sc; orc_framebuffer* fb;
    int max_bounce, use_base_color, pass_index, ns; uint32_t seed;
} pool_t;

static void* pool_worker(void* arg)
{
    pool_t* p = (pool_t*)arg;
    for (;;) {
        pool_task t;
        pthread_mutex_lock(&p->mutex);
        if (p->next >= p->n_tasks) { pthread_mutex_unlock(&p->mutex); break; }
        t = p->tasks[p->next++];
        pthread_mutex_unlock(&p->mutex);
        orc_render_range(p->sc, p->fb, t.start, t.end, p->max_bounce, p->use_base_color, p->pass_index, p->ns, p->seed);
        pthread_mutex_lock(&p->mutex);
        p->done++;
        if (p->done == p->n_tasks) pthread_cond_signal(&p->done_cond);
        pthread_mutex_unlock(&p->mutex);
    }
    stats_flush();
    return NULL;
}
int orc_hw_threads(void) { long n = sysconf(_SC_NPROCESSORS_ONLN); return n > 0 ? (int)n : 1; }

double orc_render_pass_pool(const orc_scene* sc, orc_framebuffer* fb, int max_bounce, int use_base_color,
                            int pass_index, int ns, uint32_t seed, int threads, int task_rows)
{
    if (threads < 1) threads = orc_hw_threads();
    if (task_rows < 1) task_rows = 10;
    const int w = fb->width, h = fb->height, max_idx = w * h - 1;
    pool_t p; memset(&p, 0, sizeof p);
    p.n_tasks = (h + task_rows - 1) / task_rows;
    p.tasks = (pool_task*)malloc(sizeof(pool_task) * (size_t)p.n_tasks);
    for (int i = 0, k = 0; i < h; i += task_rows, k++) {
        p.tasks[k].start = i * w;
        int e = (i + task_rows) * w - 1;
        p.tasks[k].end = e < max_idx ? e : max_idx;
    }
    pthread_mutex_init(&p.mutex, NULL); pthread_cond_init(&p.done_cond, NULL);
    p.sc = sc; p.fb = fb; p.max_bounce = max_bounce; p.use_base_color = use_base_color;
    p.pass_index = pass_index; p.ns = ns; p.seed = seed;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
    for (int i = 0; i < threads; i++) pthread_create(&th[i], NULL, pool_worker, &p);
    for (int i = 0; i < threads; i++) pthread_join(th[i], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    free(th); free(p.tasks);
    pthread_mutex_destroy(&p.mutex); pthread_cond_destroy(&p.done_cond);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* ------------------------------------------------------------------------- */
/* scene API                                                                  */
/* ------------------------------------------------------------------------- */
orc_scene* orc_scene_create(void) { orc_scene* s = (orc_scene*)calloc(1, sizeof(orc_scene)); s->combine_b_first = 1; /* g++ 11 evaluates B first: pinned by oracle/_ref */ return s; }

static void mesh_free(mesh_t* m)
{
    if (!m) return;
    free(m->points); free(m->texcoords); free(m->normals);
    free(m->point_idx); free(m->texcoord_idx); free(m->normal_idx); free(m->poly_material);
    for (int i = 0; i < ORC_MAX_MATERIALS; i++) { free(m->material_names[i]); free(m->texture_paths[i]); }
    for (int i = 0; i < m->n_textures; i++) if (m->textures[i]) { free(m->textures[i]->pixels); free(m->textures[i]); }
    free(m->textures);
    kd_free(m->root);
    free(m);
}
void orc_scene_destroy(orc_scene* sc)
{
    if (!sc) return;
    for (int i = 0; i < sc->n_shapes; i++) mesh_free(sc->shapes[i].mesh);
    free(sc->shapes); free(sc);
}
int orc_scene_add_mesh_obj(orc_scene* sc, const char* obj_path)
{
    shape_t s; memset(&s, 0, sizeof s);
    aabb_init(&s.aabb);
    s.has_culling_bounds = 1;
    s.mesh = mesh_load(obj_path, &s.aabb);
    if (!s.mesh) return -1;
    sc->shapes = (shape_t*)realloc(sc->shapes, sizeof(shape_t) * (size_t)(sc->n_shapes + 1));
    sc->shapes[sc->n_shapes] = s;
    return sc->n_shapes++;
}
static int scene_push_shape(orc_scene* sc, const shape_t* s)
{
    sc->shapes = (shape_t*)realloc(sc->shapes, sizeof(shape_t) * (size_t)(sc->n_shapes + 1));
    sc->shapes[sc->n_shapes] = *s;
    return sc->n_shapes++;
}
/* RSphere / RPlane / RCapsule constructors (Src/Shapes.h:46-112) */
int orc_scene_add_sphere(orc_scene* sc, const float center[3], float radius)
{
    shape_t s; memset(&s, 0, sizeof s);
    aabb_init(&s.aabb); s.has_culling_bounds = 1; s.kind = ORC_SHAPE_SPHERE;
    s.a = v3(center[0], center[1], center[2]); s.radius = radius;
    aabb_expand_by_sphere(&s.aabb, s.a, radius);
    return scene_push_shape(sc, &s);
}
int orc_scene_add_plane(orc_scene* sc, const float normal[3], const float point[3])
{
    shape_t s; memset(&s, 0, sizeof s);
    aabb_init(&s.aabb); s.has_culling_bounds = 0; s.kind = ORC_SHAPE_PLANE;      /* RPlane::HasCullingBounds (Src/Shapes.cpp:28-32) */
    s.a = v3(normal[0], normal[1], normal[2]); s.b = v3(point[0], point[1], point[2]);
    return scene_push_shape(sc, &s);
}
int orc_scene_add_capsule(orc_scene* sc, const float start[3], const float end[3], float radius)
{
    shape_t s; memset(&s, 0, sizeof s);
    aabb_init(&s.aabb); s.has_culling_bounds = 1; s.kind = ORC_SHAPE_CAPSULE;
    s.a = v3(start[0], start[1], start[2]); s.b = v3(end[0], end[1], end[2]); s.radius = radius;
    aabb_expand_by_sphere(&s.aabb, s.a, radius);
    aabb_expand_by_sphere(&s.aabb, s.b, radius);
    return scene_push_shape(sc, &s);
}
int orc_scene_add_triangle(orc_scene* sc, const float p0[3], const float p1[3], const float p2[3])
{
    shape_t s; memset(&s, 0, sizeof s);
    aabb_init(&s.aabb); s.has_culling_bounds = 1; s.kind = ORC_SHAPE_TRIANGLE;
    s.a = v3(p0[0], p0[1], p0[2]); s.b = v3(p1[0], p1[1], p1[2]); s.c = v3(p2[0], p2[1], p2[2]);
    aabb_expand(&s.aabb, s.a); aabb_expand(&s.aabb, s.b); aabb_expand(&s.aabb, s.c);
    return scene_push_shape(sc, &s);
}
int orc_scene_set_material(orc_scene* sc, int shape, const orc_material_node* nodes, int n)
{
    if (shape < 0 || shape >= sc->n_shapes || n < 1 || n > 64) return -1;
    for (int i = 0; i < n; i++) {
        if (nodes[i].type == ORC_MAT_BLEND || nodes[i].type == ORC_MAT_COMBINE)
            if (nodes[i].child_a <= i || nodes[i].child_a >= n || nodes[i].child_b <= i || nodes[i].child_b >= n) return -2;
    }
    memcpy(sc->shapes[shape].material.nodes, nodes, sizeof(orc_material_node) * (size_t)n);
    sc->shapes[shape].material.n = n;
    sc->shapes[shape].has_material = 1;
    return 0;
}
void orc_set_unitvec_mode(orc_scene* sc, int mode) { sc->unitvec_mode = mode; }
void orc_set_combine_order(orc_scene* sc, int b_first) { sc->combine_b_first = b_first; }

static const mesh_t* get_mesh(const orc_scene* sc, int shape)
{
    if (shape < 0 || shape >= sc->n_shapes) return NULL;
    return sc->shapes[shape].mesh;
}
int orc_mesh_counts(const orc_scene* sc, int shape, int32_t out[8])
{
    const mesh_t* m = get_mesh(sc, shape); if (!m) return -1;
    out[0] = m->n_points; out[1] = m->n_texcoords; out[2] = m->n_normals; out[3] = m->n_tris;
    out[4] = m->n_material_names; out[5] = m->n_nodes; out[6] = m->n_textures; out[7] = 0;
    return 0;
}
int orc_mesh_copy(const orc_scene* sc, int shape, int which, void* dst, int64_t max_bytes)
{
    const mesh_t* m = get_mesh(sc, shape); if (!m) return -1;
    const void* src = NULL; int64_t bytes = 0;
    switch (which) {
    case 0: src = m->points; bytes = (int64_t)m->n_points * 12; break;
    case 1: src = m->texcoords; bytes = (int64_t)m->n_texcoords * 12; break;
    case 2: src = m->normals; bytes = (int64_t)m->n_normals * 12; break;
    case 3: src = m->point_idx; bytes = (int64_t)m->n_tris * 12; break;
    case 4: src = m->texcoord_idx; bytes = (int64_t)m->n_tris * 12; break;
    case 5: src = m->normal_idx; bytes = (int64_t)m->n_tris * 12; break;
    case 6: src = m->poly_material; bytes = (int64_t)m->n_tris * 4; break;
    default: return -2;
    }
    if (bytes > max_bytes) return -3;
    memcpy(dst, src, (size_t)bytes);
    return 0;
}
int orc_mesh_num_materials(const orc_scene* sc, int shape)
{
    const mesh_t* m = get_mesh(sc, shape); return m ? m->n_material_names : -1;
}
int orc_mesh_texture_path(const orc_scene* sc, int shape, int material_id, char* buf, int buflen)
{
    const mesh_t* m = get_mesh(sc, shape);
    if (!m || material_id < 0 || material_id >= ORC_MAX_MATERIALS || buflen < 1) return -1;
    snprintf(buf, (size_t)buflen, "%s", m->texture_paths[material_id] ? m->texture_paths[material_id] : "");
    return 0;
}
int orc_mesh_set_texture(orc_scene* sc, int shape, int material_id, const uint8_t* px, int w, int h, int channels)
{
    mesh_t* m = (mesh_t*)get_mesh(sc, shape);
    if (!m || material_id < 0 || material_id >= m->n_textures || (channels != 3 && channels != 4)) return -1;
    texture_t* t = (texture_t*)calloc(1, sizeof *t);
    t->width = w; t->height = h;
    t->pixels = (vec4*)malloc(sizeof(vec4) * (size_t)w * (size_t)h);
    for (int64_t i = 0; i < (int64_t)w * h; i++) {          /* Src/Texture.cpp:119-151 */
        const uint8_t* s = px + i * channels;
        t->pixels[i].x = texel_to_linear(s[0]);
        t->pixels[i].y = texel_to_linear(s[1]);
        t->pixels[i].z = texel_to_linear(s[2]);
        t->pixels[i].w = channels == 4 ? (float)s[3] / 255 : 1.0f;
    }
    if (m->textures[material_id]) { free(m->textures[material_id]->pixels); free(m->textures[material_id]); }
    m->textures[material_id] = t;
    return 0;
}
static int tree_dump(const kd_node* n, float* bounds6, int32_t* tri, int max_nodes, int at)
{
    if (!n) return at;
    if (at < max_nodes) {
        bounds6[at * 6 + 0] = n->bounds.pmin.x; bounds6[at * 6 + 1] = n->bounds.pmin.y; bounds6[at * 6 + 2] = n->bounds.pmin.z;
        bounds6[at * 6 + 3] = n->bounds.pmax.x; bounds6[at * 6 + 4] = n->bounds.pmax.y; bounds6[at * 6 + 5] = n->bounds.pmax.z;
        tri[at] = (n->left || n->right) ? -1 : n->triangle.index;
    }
    at++;
    at = tree_dump(n->left, bounds6, tri, max_nodes, at);
    at = tree_dump(n->right, bounds6, tri, max_nodes, at);
    return at;
}
int orc_mesh_tree_preorder(const orc_scene* sc, int shape, float* bounds6, int32_t* tri, int max_nodes)
{
    const mesh_t* m = get_mesh(sc, shape); if (!m) return -1;
    return tree_dump(m->root, bounds6, tri, max_nodes, 0);
}
int orc_shape_bounds(const orc_scene* sc, int shape, float out6[6])
{
    if (shape < 0 || shape >= sc->n_shapes) return -1;
    const aabb_t* b = &sc->shapes[shape].aabb;
    out6[0] = b->pmin.x; out6[1] = b->pmin.y; out6[2] = b->pmin.z; out6[3] = b->pmax.x; out6[4] = b->pmax.y; out6[5] = b->pmax.z;
    return 0;
}

int orc_trace_closest(const orc_scene* sc, const float* rays, int64_t n, float* hit_f11, int32_t* hit_shape, int32_t* hit_tri)
{
    for (int64_t i = 0; i < n; i++) {
        ray_t r; r.origin = v3(rays[i * 7], rays[i * 7 + 1], rays[i * 7 + 2]);
        r.dir = v3(rays[i * 7 + 3], rays[i * 7 + 4], rays[i * 7 + 5]); r.distance = rays[i * 7 + 6];
        hit_t h; hit_init(&h); int tri = -1;
        int s = find_intersection(sc, r, &h, &tri);
        float* o = hit_f11 + i * 11;
        o[0] = h.pos.x; o[1] = h.pos.y; o[2] = h.pos.z; o[3] = h.normal.x; o[4] = h.normal.y; o[5] = h.normal.z;
        o[6] = h.distance; o[7] = h.color.x; o[8] = h.color.y; o[9] = h.color.z; o[10] = h.alpha;
        hit_shape[i] = s; hit_tri[i] = s >= 0 ? tri : -1;
    }
    return 0;
}
int orc_texture_sample(const orc_scene* sc, int shape, int material_id, const float* uv, int64_t n, float* rgba)
{
    const mesh_t* m = get_mesh(sc, shape);
    if (!m || material_id < 0 || material_id >= m->n_textures || !m->textures[material_id]) return -1;
    for (int64_t i = 0; i < n; i++) {
        vec4 s = texture_sample(m->textures[material_id], uv[i * 2], uv[i * 2 + 1]);
        rgba[i * 4] = s.x; rgba[i * 4 + 1] = s.y; rgba[i * 4 + 2] = s.z; rgba[i * 4 + 3] = s.w;
    }
    return 0;
}
int orc_ray_trace(const orc_scene* sc, const float* rays, const uint32_t* keys2, int64_t n,
                  int max_bounce, int use_base_color, uint32_t seed, int width, int height, float* rgb)
{
    const uint32_t phase = table_phase(seed);
    for (int64_t i = 0; i < n; i++) {
        ray_t r; r.origin = v3(rays[i * 7], rays[i * 7 + 1], rays[i * 7 + 2]);
        r.dir = v3(rays[i * 7 + 3], rays[i * 7 + 4], rays[i * 7 + 5]); r.distance = rays[i * 7 + 6];
        path_rng rng;
        uint32_t pixel = keys2[i * 2], sample = keys2[i * 2 + 1];
        path_rng_init(&rng, seed, width * height, (int)pixel, (int)(sample / 4), (int)(sample % 4));
        vec3 c = ray_trace(sc, &r, max_bounce, use_base_color, &rng, phase);
        rgb[i * 3] = c.x; rgb[i * 3 + 1] = c.y; rgb[i * 3 + 2] = c.z;
    }
    return 0;
}

/* MakePixelColor(LinearToGamma(c)) for n colours (3 floats each): Src/ColorBuffer.h:81-109 */
void orc_make_pixel_colors(const float* rgb, int64_t n, uint32_t* out)
{
    for (int64_t i = 0; i < n; i++) out[i] = make_pixel_color(linear_to_gamma(v3(rgb[i * 3], rgb[i * 3 + 1], rgb[i * 3 + 2])));
}
