// rtw_types.h -- plain-data layouts shared by the host scene builder and the HIP kernels.
//
// HBM layout (per committed scene; everything read-only while rendering):
//   nodes   RtwNode[2T-1]   32 B   preorder, left child = i+1, `skip` = next node after the subtree
//   tris    RtwTri[T]       64 B   leaf order: 3 positions + face normal + plane offset + original index
//   shade   RtwShade[T]     64 B   leaf order: 3 vertex normals, 3 uv pairs, material id
//   texels  uint32[...]     RGBA8 atlas, one slab per texture; linearised through a 256-entry LUT
//   unit    float[3*0xFFFFFF]  the reference's pseudo-random unit-vector table (per context)
#pragma once
#include <stdint.h>

struct RtwNode {            // 2 x 16 B
    float min_x, min_y, min_z; int32_t skip;
    float max_x, max_y, max_z; int32_t tri;      // leaf slot, or -1 for an internal node
};
static_assert(sizeof(RtwNode) == 32, "RtwNode");

struct RtwPNode {           // 2 x 16 B
    float min_x, max_x, min_y, max_y;
    float min_z, max_z; int32_t skip; int32_t link;   // link >= 0: a leaf's slot; < 0: internal, left child = record -1 - link
};
static_assert(sizeof(RtwPNode) == 32, "RtwPNode");

struct RtwTri {             // 4 x 16 B
    float p0x, p0y, p0z, nx;
    float p1x, p1y, p1z, ny;
    float p2x, p2y, p2z, nz;
    float d1; int32_t orig; int32_t pad0, pad1;  // d1 = dot(N, p0); orig = triangle index in the mesh
};
static_assert(sizeof(RtwTri) == 64, "RtwTri");

struct RtwShade {           // 4 x 16 B
    float n0x, n0y, n0z, n1x;
    float n1y, n1z, n2x, n2y;
    float n2z, u0, v0, u1;
    float v1, u2, v2; int32_t material;
};
static_assert(sizeof(RtwShade) == 64, "RtwShade");

struct RtwTexture {         // 16 B
    uint32_t offset;        // first texel in the atlas
    int32_t width, height;
    int32_t valid;
};

struct RtwMaterialNode {    // mirrors rtw_material_node
    int32_t type; float r, g, b; float param; int32_t child_a, child_b; int32_t pad;
};

#define RTW_DEV_MAX_TEXTURES 64
#define RTW_DEV_MAX_MATERIAL_NODES 64
#define RTW_DEV_MAX_SHAPES 32
#define RTW_SHAPE_MESH 0
#define RTW_SHAPE_SPHERE 1
#define RTW_SHAPE_PLANE 2
#define RTW_SHAPE_CAPSULE 3
#define RTW_SHAPE_TRIANGLE 4

struct RtwShapeDev {
    const RtwNode* nodes;
    const RtwTri* tris;
    const RtwShade* shade;
    const uint32_t* texels;
    // flat hierarchy over the leaves in preorder (16 consecutive entries of a level share one entry of the next):
    // level 0 = the leaves' own boxes, 1 = groups of 16 leaves, 2 = groups of 256.  Entry i of a level is the 24 bytes
    // flat[l][6 i ..]: (min x, max x), (min y, max y), (min z, max z) -- one pair per axis, so that (pair - origin) * reciprocal
    // is one packed subtract and one packed multiply; flat_pad[l] entries are allocated (whole waves may read past flat_n[l]).
    const float* flat[3];
    int32_t flat_n[3], flat_pad[3];
    // the same binary tree once more for the ray-per-lane walk (gtrace_kernel): RtwNode records with EXPLICIT links -- `skip` as in
    // `nodes`, `tri` >= 0 a leaf's slot, < 0 an internal node whose left child is record -1 - tri -- stored so that the tnodes_top
    // records of the tree's upper levels come first, in preorder (a block stages them in LDS), followed by the subtrees below them,
    // each contiguous in preorder.  The walk visits the records in the reference's order whatever their place in memory.
    // A record is RtwPNode: the box as (min, max) PAIRS per axis, so that (pair - origin) * reciprocal is one packed subtract and one
    // packed multiply per axis (v_pk_add_f32 / v_pk_mul_f32: component-wise the reference's float operations).
    const struct RtwPNode* tnodes;
    const float* planes;            // per leaf slot 4 floats: the triangle's face normal and plane offset (RtwTri nx, ny, nz, d1) on their own, 16 bytes a leaf: the
                                    // ray-per-lane walk drops a leaf whose triangle faces away from the ray's origin (RRay::TestIntersectionWithTriangle's first
                                    // rejection, Src/RRay.cpp:156-158, which does not depend on the segment) before it is ever noted; null: not built
    float bmin[3], bmax[3];         // RShape::Aabb (all `v` lines)
    int32_t n_nodes, n_tris;
    int32_t n_textures;             // size of the reference's Textures vector (= triangle count when an MTL exists)
    int32_t has_material;
    int32_t n_material_nodes;
    // RTW_SHAPE_MESH: everything above; a sphere / plane / capsule has no arrays, only these (Src/Shapes.h:46-112):
    //   sphere: pa = Center, radius;  plane: pa = Normal, pb = Point;  capsule: pa = Start, pb = End, radius;  triangle: see pc
    int32_t kind;
    float pa[3], pb[3], radius;
    float pc[3], pn[3], pd1;        // triangle (RTriangle, Src/Shapes.h:106-130): pa, pb, pc = Points[0..2]; pn = its face normal and pd1 = dot(pn, pa) as
                                    // RRay::TestIntersectionWithTriangle computes them at every test (Src/RRay.cpp:138-145), here once on the host
    int32_t tnodes_top;
    RtwTexture textures[RTW_DEV_MAX_TEXTURES];
    RtwMaterialNode material[RTW_DEV_MAX_MATERIAL_NODES];
};

struct RtwSceneDev {
    int32_t n_shapes;
    int32_t prune;
    int32_t traversal;              // 1 (default): bins, flat hierarchy and link tree may be used; 0: binary preorder walk only (reference visit counts)
    int32_t debug_table_mask;       // timing experiments only: nonzero -> unit-table index &= mask (changes the image)
    const float* unit_table;        // 3 floats per entry
    const float* gamma_thr;         // 256
    const float* texel_lut;         // 256
    unsigned long long* stats;      // 8 counters or nullptr
    RtwShapeDev shapes[RTW_DEV_MAX_SHAPES];
};

struct RtwRenderParams {
    int32_t width, height;
    int32_t begin, count;           // linear work items: pixel = map(begin + i)
    int32_t task_rows, rank, world; // world <= 1: contiguous range
    int32_t max_bounce, preview, pass_index, sub_samples;
    uint32_t seed;
    // tiled work mapping (tile_w != 0): wave w of the launch renders the tile_w x tile_h pixel tile number w of its rows,
    // aligned to the screen's bin grid; `bins` then holds, per shape, the leaves whose box can be met by a camera ray of each bin
    int32_t tile_w, tile_h, tile_shift, tiles_per_row;
    int32_t row0, nrows;            // contiguous range: first screen row and number of rows; task partition: number of virtual rows
    int32_t lead_shapes;            // the scene's first lead_shapes shapes are spheres / planes / capsules / triangles;
                                    // the lane that sets up a path's next segment (shade_hit_step) tests them there, one ray per lane, and
                                    // leaves the partial scene query in the path's hit record; the wave-per-ray trace continues from it
    int32_t n_jobs;                 // entries of tile_order
    const struct RtwBinsDev* bins;  // [n_shapes] or null
    const uint32_t* tile_order;     // full-frame launches: the order in which the primary kernel takes the tiles (null = as numbered)
    const float* cam_dx;            // tiled mapping: dx of every pixel column / dy of every pixel row (Src/RayTracerProgram.cpp:141-142),
    const float* cam_dy;            // computed once on the host with the same float operations
};

// screen-space bins of one shape for one (width, height, tile shape): CSR over the bins, entries = node index of a leaf, ascending
struct RtwBinsDev {
    const uint32_t* off;            // n_bins + 1 offsets, or null: no bins for this shape (camera too close to it) -> tree walk
    const uint32_t* ent;
};

// ---- pass-batched pipeline (pipeline 4): the passes first_pass .. first_pass + n_passes - 1 of one rtw_render_passes call share one
// set of launches.  Tiles of the launch are either BUSY (some shape lists a leaf in the tile's bin, or has no bins) or SKY-ONLY.
// A path's slot = ((b * sub_samples + sub) * n_passes + k) * 64 + lane   (b = rank of its tile in the busy list, k = pass - first_pass).
struct RtwGroupParams {
    RtwRenderParams rp;             // frame, tile mapping, bounce limit, seed, bins, camera tables (pass_index unused)
    int32_t first_pass, n_passes;
    int32_t n_busy, n_sky, n_jobs;
    const uint32_t* busy_tiles;     // tile numbers of the busy tiles, heaviest bins first; null: tile = first_tile + b (every tile of the range)
    const uint32_t* sky_tiles;      // tile numbers of the sky-only tiles
    const uint32_t* jobs;           // primary kernel: b | (sub-sample + 1) << 24 (0 in bits 24..27: every sub-sample); null: job = b
    int32_t range_begin, range_end; // inclusive pixel range that is rendered (a lane outside it is dead)
    int32_t first_tile;
    int32_t primary_passes;         // passes one wave of the primary kernel takes for its tile (> 1: x sub-samples <= 4; -1: one ray set at a time, no shared walk)
};

// random-stream constants (shared with the oracle by specification, not by code)
#define RTW_TABLE_SIZE 0xFFFFFFu
#define RTW_TABLE_STRIDE 16u
#define RTW_TABLE_SEED 0x52544142u
