// rtw_host.h -- host-side scene library (internal).  C++17, no HIP types.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "rtw_types.h"

namespace rtw {

struct Vec3 { float x, y, z; };

struct HostTexture {
    int width = 0, height = 0;
    std::vector<uint32_t> rgba8;      // R | G<<8 | B<<16 | A<<24
    bool valid = false;
};

// The arrays RMeshShape owns (Src/MeshShape.h:25-37) plus the flattened tree.
struct HostMesh {
    std::vector<Vec3> points, texcoords, normals;
    std::vector<int32_t> point_idx, texcoord_idx, normal_idx;   // 3 per triangle
    std::vector<int32_t> poly_material;                         // per triangle
    std::vector<std::string> material_names;
    std::vector<std::string> texture_paths;                     // per material id ("" = none)
    std::vector<HostTexture> textures;                          // per material id
    int n_textures_slots = 0;          // size of the reference's Textures vector (0 without an MTL)
    float bmin[3], bmax[3];            // RShape::Aabb
    int kind = RTW_SHAPE_MESH;         // a sphere / plane / capsule is a shape record without arrays (Src/Shapes.h:46-112)
    float pa[3] = { 0, 0, 0 }, pb[3] = { 0, 0, 0 }, radius = 0;
    float pc[3] = { 0, 0, 0 }, pn[3] = { 0, 0, 0 }, pd1 = 0;
    std::vector<RtwMaterialNode> material;
    // built by build_tree():
    std::vector<RtwNode> nodes;
    std::vector<RtwTri> tris;          // leaf order
    std::vector<RtwShade> shade;       // leaf order
    int max_depth = 0;
    std::vector<RtwPNode> tnodes;      // the tree with explicit links, upper levels first (RtwShapeDev::tnodes)
    int tnodes_top = 0;
    std::vector<float> flat[3];        // flat hierarchy over the leaves in preorder, see RtwShapeDev::flat
    int flat_n[3] = { 0, 0, 0 }, flat_pad[3] = { 0, 0, 0 };
    int n_tris() const { return (int)(point_idx.size() / 3); }
};

// OBJ + MTL (+ PNG) with the reference parser's semantics (Src/MeshShape.cpp:65-278).
// Returns empty string on success, else an error message.
std::string load_obj(const std::string& path, HostMesh& out);
// Validate index ranges; compute bounds if `bounds6` is null.
std::string finish_arrays(HostMesh& m, const float* bounds6);
// Face normal and plane offset of a triangle as RRay::TestIntersectionWithTriangle computes them (Src/RRay.cpp:138-145), the
// same operations build_tree() uses for a mesh's triangle records.
void triangle_plane(const float p0[3], const float p1[3], const float p2[3], float n[3], float* d1);
// KdNode::Build restated (Src/KdTree.cpp:37-126) + flatten to preorder/skip-link form.
void build_tree(HostMesh& m);
// The tree with explicit links, the at most `top_budget` records of its upper levels first (needs build_tree()).
void build_tnodes(HostMesh& m, int top_budget);
// Flat hierarchy: leaf boxes in preorder and the unions of every 16 / 256 consecutive leaves (needs build_tree()).
void build_flat(HostMesh& m);
// Screen-space bins of the reference camera (Src/RayTracerProgram.cpp:133-165: origin (0,0,7), image plane z = -0.5)
// for a width x height frame cut into bin_w x bin_h pixel bins: for each bin the leaves (node indices, ascending) whose
// box the line of any camera ray of the bin's pixels can meet.  Returns false (no bins) when some leaf box is not
// wholly in front of the camera.
bool build_bins(const HostMesh& m, int width, int height, int bin_w, int bin_h, std::vector<uint32_t>& off, std::vector<uint32_t>& ent);

// tables
uint32_t rand31(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t counter);
void unit_table_entry(uint32_t index, float out3[3]);
void fill_unit_table(float* dst /* 3 * RTW_TABLE_SIZE floats */, int threads);
void gamma_thresholds(float out256[256]);
void texel_lut(float out256[256]);

// PNG (8-bit RGB / RGBA, non-interlaced or Adam7) on zlib
std::string png_load(const std::string& path, std::vector<uint8_t>& texels, int& w, int& h, int& channels);
std::string png_save_rgb(const std::string& path, const uint8_t* rgb, int w, int h);

}  // namespace rtw
