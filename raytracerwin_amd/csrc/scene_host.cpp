// scene_host.cpp -- host-side scene library: OBJ/MTL loading with the reference parser's
// observable semantics, the reference's split decisions for the one-triangle-per-leaf
// bounding tree flattened to a stack-free preorder array, and the host-generated tables
// the kernels read (unit vectors, gamma thresholds, texel LUT).
//
// Build with -ffp-contract=off: split decisions, face normals and table entries must be
// the same bits the reference computes (Src/KdTree.cpp:37-126, Src/RRay.cpp:138-145,
// Src/Math.h:34-40).
#include "rtw_host.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <thread>

namespace rtw {

namespace {

inline Vec3 sub(Vec3 a, Vec3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline Vec3 add(Vec3 a, Vec3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
inline Vec3 cross(Vec3 a, Vec3 b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
inline float dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline bool near_zero(float a) { return std::fabs(a) < FLT_EPSILON; }   // FLT_EQUAL_ZERO, Src/MathHelper.h:12

// RVec3::GetNormalizedVec3 (Src/RVector.h:169-183): tiny vectors are returned un-normalised
inline Vec3 normalized(Vec3 v)
{
    float sq = v.x * v.x + v.y * v.y + v.z * v.z;
    if (!near_zero(sq)) {
        float inv = 1.0f / std::sqrt(sq);
        v.x *= inv; v.y *= inv; v.z *= inv;
    }
    return v;
}

// --- line tokenising ------------------------------------------------------------------
// "keyword" = text before the first blank (Src/MeshShape.cpp:23-36)
std::string keyword_of(const std::string& line)
{
    size_t sp = line.find(' ');
    return sp == std::string::npos ? line : line.substr(0, sp);
}
// fields separated by ONE blank each; a trailing blank does not open an empty field
// (std::getline semantics of Src/MeshShape.cpp:50-62)
std::vector<std::string> blank_fields(const std::string& line)
{
    std::vector<std::string> out;
    size_t pos = 0;
    while (pos < line.size()) {
        size_t sp = line.find(' ', pos);
        if (sp == std::string::npos) { out.push_back(line.substr(pos)); break; }
        out.push_back(line.substr(pos, sp - pos));
        pos = sp + 1;
    }
    return out;
}
// whitespace-delimited reader for the numeric lines (operator>> semantics)
struct Cursor {
    const char* p;
    explicit Cursor(const std::string& s) : p(s.c_str()) {}
    void skip_word() { while (*p == ' ' || *p == '\t') p++; while (*p && *p != ' ' && *p != '\t') p++; }
    float number() { char* e; float v = std::strtof(p, &e); if (e == p) return 0.0f; p = e; return v; }
    std::string word()
    {
        while (*p == ' ' || *p == '\t') p++;
        const char* b = p;
        while (*p && *p != ' ' && *p != '\t' && *p != '\r') p++;
        return std::string(b, p);
    }
};
// k-th integer of a "p/t/n" vertex reference (Src/MeshShape.cpp:38-48)
int slash_field(const std::string& tok, int k)
{
    const char* p = tok.c_str();
    int value = -1;
    for (int i = 0; i <= k; i++) {
        char* e; long v = std::strtol(p, &e, 10);
        if (e == p) return 0;          // failed extraction leaves 0
        value = (int)v; p = e;
        if (!*p) { if (i < k) return value; break; }
        p++;                            // the separator
    }
    return value;
}

bool open_with_parent_fallback(const std::string& name, std::ifstream& f, std::string& resolved)
{
    resolved = name;
    f.open(resolved, std::ios::binary);
    for (int i = 0; !f.is_open() && i < 2; i++) {       // Src/MeshShape.cpp:70-83
        resolved = "../" + resolved;
        f.clear(); f.open(resolved, std::ios::binary);
    }
    return f.is_open();
}

}  // namespace

std::string load_obj(const std::string& path, HostMesh& m)
{
    std::ifstream f; std::string resolved;
    if (!open_with_parent_fallback(path, f, resolved)) return "unable to open " + path;

    int current = -1;
    std::string line;
    while (std::getline(f, line)) {
        const std::string key = keyword_of(line);
        if (key == "v" || key == "vn") {
            Cursor c(line); c.skip_word();
            Vec3 v; v.x = c.number(); v.y = c.number(); v.z = c.number();
            (key == "v" ? m.points : m.normals).push_back(v);
        } else if (key == "vt") {
            Cursor c(line); c.skip_word();
            Vec3 v; v.x = c.number(); v.y = c.number(); v.z = 0.0f;
            m.texcoords.push_back(v);
        } else if (key == "f") {
            const std::vector<std::string> fld = blank_fields(line);
            const int corners = (int)fld.size() - 1;
            static const int fan3[3] = { 0, 1, 2 }, fan4[6] = { 0, 1, 2, 0, 2, 3 };   // quad -> (0,1,2),(0,2,3)
            const int* order = corners == 3 ? fan3 : corners == 4 ? fan4 : nullptr;
            const int n = corners == 3 ? 3 : corners == 4 ? 6 : 0;                  // other polygons are dropped
            for (int i = 0; i < n; i++) {
                const std::string& ref = fld[(size_t)order[i] + 1];
                m.point_idx.push_back(slash_field(ref, 0) - 1);
                m.texcoord_idx.push_back(slash_field(ref, 1) - 1);
                m.normal_idx.push_back(slash_field(ref, 2) - 1);
                if (i % 3 == 0) m.poly_material.push_back(current);
            }
        } else if (key == "usemtl") {
            const std::vector<std::string> fld = blank_fields(line);
            const std::string name = fld.size() > 1 ? fld[1] : std::string();
            current = -1;
            for (size_t i = 0; i < m.material_names.size(); i++) if (m.material_names[i] == name) { current = (int)i; break; }
            if (current < 0) { m.material_names.push_back(name); current = (int)m.material_names.size() - 1; }
        }
    }
    f.close();
    m.texture_paths.assign(m.material_names.size(), std::string());
    m.textures.assign(m.material_names.size(), HostTexture());

    // sibling .mtl: newmtl / map_Kd only (Src/MeshShape.cpp:202-272)
    std::string mtl = resolved;
    size_t ext = mtl.find(".obj");
    if (ext != std::string::npos) {
        mtl.replace(ext, 4, ".mtl");
        std::ifstream mf(mtl, std::ios::binary);
        if (mf.is_open()) {
            std::string base;
            size_t slash = mtl.find_last_of("\\/");
            if (slash != std::string::npos) base = mtl.substr(0, slash + 1);
            m.n_textures_slots = m.n_tris();           // Textures.resize(PolyMaterialId.size())
            current = -1;
            while (std::getline(mf, line)) {
                const std::string key = keyword_of(line);
                if (key == "newmtl") {
                    Cursor c(line); c.skip_word();
                    const std::string name = c.word();
                    current = -1;
                    for (size_t i = 0; i < m.material_names.size(); i++) if (m.material_names[i] == name) { current = (int)i; break; }
                } else if (key == "map_Kd" && current != -1) {
                    Cursor c(line); c.skip_word();
                    std::string tex = base + c.word();
                    for (size_t bs = tex.find("\\\\"); bs != std::string::npos; bs = tex.find("\\\\")) tex.replace(bs, 2, "/");
                    m.texture_paths[(size_t)current] = tex;
                    std::vector<uint8_t> px; int w = 0, h = 0, ch = 0;
                    HostTexture t;
                    if (png_load(tex, px, w, h, ch).empty()) {      // a bad PNG degrades to "no texture" (Src/Texture.cpp:161-196)
                        t.width = w; t.height = h; t.valid = true;
                        t.rgba8.resize((size_t)w * (size_t)h);
                        for (size_t i = 0; i < t.rgba8.size(); i++) {
                            const uint8_t* s = &px[i * (size_t)ch];
                            t.rgba8[i] = (uint32_t)s[0] | ((uint32_t)s[1] << 8) | ((uint32_t)s[2] << 16) | ((uint32_t)(ch == 4 ? s[3] : 255) << 24);
                        }
                    }
                    m.textures[(size_t)current] = std::move(t);
                }
            }
        }
    }
    return finish_arrays(m, nullptr);
}

std::string finish_arrays(HostMesh& m, const float* bounds6)
{
    const size_t nv = m.point_idx.size();
    if (nv % 3 != 0 || m.texcoord_idx.size() != nv || m.normal_idx.size() != nv || m.poly_material.size() != nv / 3)
        return "inconsistent index array sizes";
    for (size_t i = 0; i < nv; i++) {
        // the reference would index out of bounds here; the library refuses the mesh instead
        if (m.point_idx[i] < 0 || m.point_idx[i] >= (int)m.points.size()) return "position index out of range";
        if (m.texcoord_idx[i] < 0 || m.texcoord_idx[i] >= (int)m.texcoords.size()) return "texcoord index out of range";
        if (m.normal_idx[i] < 0 || m.normal_idx[i] >= (int)m.normals.size()) return "normal index out of range";
    }
    for (size_t i = 0; i < nv / 3; i++)
        if (m.poly_material[i] < -1 || m.poly_material[i] >= RTW_DEV_MAX_TEXTURES) return "material id out of range";
    if (bounds6) {
        for (int k = 0; k < 3; k++) { m.bmin[k] = bounds6[k]; m.bmax[k] = bounds6[k + 3]; }
    } else {
        // RShape::Aabb grows over every `v` line, referenced or not (Src/MeshShape.cpp:109)
        for (int k = 0; k < 3; k++) { m.bmin[k] = FLT_MAX; m.bmax[k] = -FLT_MAX; }
        for (const Vec3& p : m.points) {
            const float c[3] = { p.x, p.y, p.z };
            for (int k = 0; k < 3; k++) { if (c[k] < m.bmin[k]) m.bmin[k] = c[k]; if (c[k] > m.bmax[k]) m.bmax[k] = c[k]; }
        }
    }
    if (m.textures.size() < m.material_names.size()) m.textures.resize(m.material_names.size());
    return std::string();
}

// ---------------------------------------------------------------------------------------
// tree: the reference's recursion (bounds of the node's vertices; leaf iff one triangle;
// split at the mean centroid along the largest extent, strict '<' goes left; a one-sided
// split falls back to first half / second half of the current order), emitted in preorder.
// ---------------------------------------------------------------------------------------
namespace {

struct BuildRef { int32_t a, b, c, index; };

struct TreeBuilder {
    HostMesh& m;
    int deepest = 0;
    explicit TreeBuilder(HostMesh& mesh) : m(mesh) {}

    static int split_axis(const float lo[3], const float hi[3])
    {
        const float sx = hi[0] - lo[0], sy = hi[1] - lo[1], sz = hi[2] - lo[2];
        if (sx > sy) return sx > sz ? 0 : 2;     // Src/KdTree.cpp:13-34 (ties go to Z, then Y)
        return sy > sz ? 1 : 2;
    }
    Vec3 centroid(const BuildRef& t) const
    {
        const Vec3 s = add(add(m.points[(size_t)t.a], m.points[(size_t)t.b]), m.points[(size_t)t.c]);
        return { s.x / 3.0f, s.y / 3.0f, s.z / 3.0f };
    }

    void emit(const std::vector<BuildRef>& refs, int depth)
    {
        if (depth > deepest) deepest = depth;
        const size_t self = m.nodes.size();
        m.nodes.emplace_back();
        float lo[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, hi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
        for (const BuildRef& t : refs) {
            for (int32_t vi : { t.a, t.b, t.c }) {
                const Vec3& p = m.points[(size_t)vi];
                const float c[3] = { p.x, p.y, p.z };
                for (int k = 0; k < 3; k++) { if (c[k] < lo[k]) lo[k] = c[k]; if (c[k] > hi[k]) hi[k] = c[k]; }
            }
        }
        RtwNode nd;
        nd.min_x = lo[0]; nd.min_y = lo[1]; nd.min_z = lo[2];
        nd.max_x = hi[0]; nd.max_y = hi[1]; nd.max_z = hi[2];
        nd.tri = -1; nd.skip = 0;

        if (refs.size() == 1) {
            nd.tri = (int32_t)m.tris.size();
            add_leaf_records(refs[0]);
        } else {
            Vec3 mean = { 0, 0, 0 };
            for (const BuildRef& t : refs) mean = add(mean, centroid(t));
            const float n = (float)refs.size();
            mean.x /= n; mean.y /= n; mean.z /= n;
            const int axis = split_axis(lo, hi);
            const float cut = axis == 0 ? mean.x : axis == 1 ? mean.y : mean.z;
            std::vector<BuildRef> left, right;
            for (const BuildRef& t : refs) {
                const Vec3 c = centroid(t);
                const float v = axis == 0 ? c.x : axis == 1 ? c.y : c.z;
                (v < cut ? left : right).push_back(t);
            }
            if (left.size() == refs.size() || right.size() == refs.size()) {
                const size_t half = refs.size() / 2;
                left.assign(refs.begin(), refs.begin() + (long)half);
                right.assign(refs.begin() + (long)half, refs.end());
            }
            if (!left.empty()) emit(left, depth + 1);
            if (!right.empty()) emit(right, depth + 1);
        }
        nd.skip = (int32_t)m.nodes.size();
        m.nodes[self] = nd;
    }

    void add_leaf_records(const BuildRef& t)
    {
        const Vec3 p0 = m.points[(size_t)t.a], p1 = m.points[(size_t)t.b], p2 = m.points[(size_t)t.c];
        // the reference recomputes this per test (Src/RRay.cpp:138-145); it only depends on the triangle
        const Vec3 n = normalized(cross(sub(p1, p0), sub(p2, p0)));
        RtwTri r;
        r.p0x = p0.x; r.p0y = p0.y; r.p0z = p0.z; r.nx = n.x;
        r.p1x = p1.x; r.p1y = p1.y; r.p1z = p1.z; r.ny = n.y;
        r.p2x = p2.x; r.p2y = p2.y; r.p2z = p2.z; r.nz = n.z;
        r.d1 = dot(n, p0); r.orig = t.index; r.pad0 = r.pad1 = 0;
        m.tris.push_back(r);
        const size_t v = (size_t)t.index * 3;
        const Vec3 n0 = m.normals[(size_t)m.normal_idx[v]], n1 = m.normals[(size_t)m.normal_idx[v + 1]], n2 = m.normals[(size_t)m.normal_idx[v + 2]];
        const Vec3 t0 = m.texcoords[(size_t)m.texcoord_idx[v]], t1 = m.texcoords[(size_t)m.texcoord_idx[v + 1]], t2 = m.texcoords[(size_t)m.texcoord_idx[v + 2]];
        RtwShade s;
        s.n0x = n0.x; s.n0y = n0.y; s.n0z = n0.z; s.n1x = n1.x; s.n1y = n1.y; s.n1z = n1.z; s.n2x = n2.x; s.n2y = n2.y; s.n2z = n2.z;
        s.u0 = t0.x; s.v0 = t0.y; s.u1 = t1.x; s.v1 = t1.y; s.u2 = t2.x; s.v2 = t2.y;
        s.material = m.poly_material[(size_t)t.index];
        m.shade.push_back(s);
    }
};

}  // namespace

void triangle_plane(const float p0[3], const float p1[3], const float p2[3], float n[3], float* d1)
{
    const Vec3 a = { p0[0], p0[1], p0[2] }, b = { p1[0], p1[1], p1[2] }, c = { p2[0], p2[1], p2[2] };
    const Vec3 nn = normalized(cross(sub(b, a), sub(c, a)));
    n[0] = nn.x; n[1] = nn.y; n[2] = nn.z;
    *d1 = dot(nn, a);
}

void build_tree(HostMesh& m)
{
    m.nodes.clear(); m.tris.clear(); m.shade.clear(); m.max_depth = 0;
    const int n = m.n_tris();
    if (n == 0) return;
    std::vector<BuildRef> all((size_t)n);
    for (int i = 0; i < n; i++) all[(size_t)i] = { m.point_idx[(size_t)i * 3], m.point_idx[(size_t)i * 3 + 1], m.point_idx[(size_t)i * 3 + 2], i };
    m.nodes.reserve((size_t)2 * (size_t)n);
    TreeBuilder b(m);
    b.emit(all, 1);
    m.max_depth = b.deepest;
}

// ---------------------------------------------------------------------------------------
// The tree with explicit links for the ray-per-lane walk.  Depth limit D = the deepest level such that the nodes of depth <= D
// number at most top_budget; those come first (preorder), then, for every internal node of depth D in preorder, the nodes below it
// (preorder).  A record's links name records of this array; following them visits the nodes in the order KdNode::TestRayIntersection
// does (Src/KdTree.cpp:128-195), exactly like `nodes` with left = i + 1.
// ---------------------------------------------------------------------------------------
void build_tnodes(HostMesh& m, int top_budget)
{
    m.tnodes.clear(); m.tnodes_top = 0;
    const int n = (int)m.nodes.size();
    if (n == 0) return;
    std::vector<int> depth((size_t)n, 0);
    std::vector<int> per_level;
    {   // preorder: the left child of i is i + 1, the right child is the left child's skip target
        depth[0] = 0;
        for (int i = 0; i < n; i++) {
            if ((int)per_level.size() <= depth[(size_t)i]) per_level.resize((size_t)depth[(size_t)i] + 1, 0);
            per_level[(size_t)depth[(size_t)i]]++;
            if (m.nodes[(size_t)i].tri < 0) {
                const int l = i + 1, r = m.nodes[(size_t)l].skip;
                depth[(size_t)l] = depth[(size_t)i] + 1;
                if (r < n) depth[(size_t)r] = depth[(size_t)i] + 1;
            }
        }
    }
    int D = -1, count = 0;
    for (size_t d = 0; d < per_level.size(); d++) { if (count + per_level[d] > top_budget) break; count += per_level[d]; D = (int)d; }
    std::vector<int> order; order.reserve((size_t)n);
    for (int i = 0; i < n; i++) if (depth[(size_t)i] <= D) order.push_back(i);
    m.tnodes_top = (int)order.size();
    if (D < 0) { for (int i = 0; i < n; i++) order.push_back(i); }      // nothing fits: plain preorder
    else {
        for (int i = 0; i < n; i++) {
            if (depth[(size_t)i] != D || m.nodes[(size_t)i].tri >= 0) continue;
            for (int j = i + 1; j < m.nodes[(size_t)i].skip; j++) order.push_back(j);      // the subtree below i, preorder
        }
    }
    std::vector<int> place((size_t)n + 1, n);
    for (int k = 0; k < n; k++) place[(size_t)order[(size_t)k]] = k;
    m.tnodes.resize((size_t)n);
    for (int k = 0; k < n; k++) {
        const RtwNode& src = m.nodes[(size_t)order[(size_t)k]];
        RtwPNode t;
        t.min_x = src.min_x; t.max_x = src.max_x; t.min_y = src.min_y; t.max_y = src.max_y; t.min_z = src.min_z; t.max_z = src.max_z;
        t.skip = place[(size_t)src.skip];
        t.link = src.tri >= 0 ? src.tri : -1 - place[(size_t)order[(size_t)k] + 1];
        m.tnodes[(size_t)k] = t;
    }
}

// ---------------------------------------------------------------------------------------
// Flat hierarchy.  Leaves in preorder; level l + 1 entry k = union of level l entries [16k, 16k + 16).
// A ray whose line meets a leaf's own box meets every union that contains it (the slab test is monotone
// in the bounds), so testing unions first only culls; the leaf's own box then gets the reference's test.
// ---------------------------------------------------------------------------------------
void build_flat(HostMesh& m)
{
    for (int l = 0; l < 3; l++) { m.flat[l].clear(); m.flat_n[l] = m.flat_pad[l] = 0; }
    const int n0 = (int)m.tris.size();
    if (n0 == 0) return;
    int n = n0;
    for (int l = 0; l < 3; l++) {
        m.flat_n[l] = n;
        m.flat_pad[l] = ((n + 63) / 64) * 64 + 64;          // whole waves may read past n (values unused)
        m.flat[l].assign((size_t)6 * (size_t)m.flat_pad[l], 0.0f);
        n = (n + 15) / 16;
    }
    // entry i of a level = six floats (min x, max x, min y, max y, min z, max z); c = 0..2 min x/y/z, 3..5 max x/y/z
    auto at = [&](int l, int c, int i) -> float& { return m.flat[l][(size_t)i * 6 + (size_t)(c < 3 ? 2 * c : 2 * (c - 3) + 1)]; };
    for (const RtwNode& nd : m.nodes) {
        if (nd.tri < 0) continue;
        at(0, 0, nd.tri) = nd.min_x; at(0, 1, nd.tri) = nd.min_y; at(0, 2, nd.tri) = nd.min_z;
        at(0, 3, nd.tri) = nd.max_x; at(0, 4, nd.tri) = nd.max_y; at(0, 5, nd.tri) = nd.max_z;
    }
    for (int l = 1; l < 3; l++) {
        for (int k = 0; k < m.flat_n[l]; k++) {
            for (int c = 0; c < 3; c++) {
                float lo = FLT_MAX, hi = -FLT_MAX;
                for (int i = 16 * k; i < 16 * k + 16 && i < m.flat_n[l - 1]; i++) {
                    if (at(l - 1, c, i) < lo) lo = at(l - 1, c, i);
                    if (at(l - 1, c + 3, i) > hi) hi = at(l - 1, c + 3, i);
                }
                at(l, c, k) = lo; at(l, c + 3, k) = hi;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// Screen-space bins of the reference's fixed camera.  Pixel (x, y), sub-sample offset (ox, oy) looks along
// (dx + ox, dy + oy, -0.5) with dx = -(x - W/2) / (2W) * (W/H), dy = -(y - H/2) / (2H) (Src/RayTracerProgram.cpp:141-165),
// so a point q (relative to the camera, q.z < 0) is seen at x = W/2 + H q.x / q.z, y = H/2 + H q.y / q.z (the offsets
// move a sample by less than half a pixel).  A bin lists the leaves whose triangle a camera ray of the bin's pixels could
// ACCEPT: the triangle faces the camera (the reference's own float test, identical for every camera ray) and its projection,
// grown by a one-pixel margin, touches the bin (separating-axis test against the bin's rectangle).  A leaf left out would have
// been box- or triangle-tested by the reference and rejected, which changes no result; the kernel still runs the reference's
// box and triangle tests on every listed leaf, per ray.
// ---------------------------------------------------------------------------------------
bool build_bins(const HostMesh& m, int width, int height, int bin_w, int bin_h, std::vector<uint32_t>& off, std::vector<uint32_t>& ent)
{
    off.clear(); ent.clear();
    if (width <= 0 || height <= 0 || bin_w <= 0 || bin_h <= 0) return false;
    const int bx = (width + bin_w - 1) / bin_w, by = (height + bin_h - 1) / bin_h;      // the bins at the right / bottom edge may be partial
    const double cx = (double)(width / 2), cy = (double)(height / 2), H = (double)height;
    // A sub-sample looks through a point up to 1.25 / (4 W) off its pixel centre in dx and dy (Src/RayTracerProgram.cpp:147-162), i.e.
    // 1.25 H / (2 W) PIXELS (a pixel is 1 / (2 H) wide in those units): under half a pixel for landscape frames, several pixels for tall
    // narrow ones.  The margin covers that plus a pixel of slack.
    const double margin = 1.0 + 1.25 * (double)height / (2.0 * (double)width);
    struct Item { int node, bin; };
    std::vector<Item> items;
    items.reserve(m.tris.size() * 4);
    for (size_t i = 0; i < m.nodes.size(); i++) {
        const RtwNode& nd = m.nodes[i];
        if (nd.tri < 0) continue;
        const double zmax = (double)nd.max_z - 7.0;
        if (!(zmax < -0.01)) return false;                      // not wholly in front of the camera: no bins for this mesh
        const RtwTri& t = m.tris[(size_t)nd.tri];
        // Every camera ray starts at (0, 0, 7): the reference's first rejection, "origin behind the triangle's plane"
        // (d2 = N.O - N.P0 < 0, Src/RRay.cpp:156-160), is the same float computation for all of them, so a triangle that
        // fails it is accepted by no camera ray and needs no bin at all.
        {
            const float ox = 0.0f, oy = 0.0f, oz = 7.0f;
            const float d0 = t.nx * ox + t.ny * oy + t.nz * oz;
            const float d2 = d0 - t.d1;
            if (d2 < 0) continue;
        }
        // the triangle's projection (an accepted hit lies inside the triangle, so its pixel lies inside this projection)
        const double vx[3] = { t.p0x, t.p1x, t.p2x }, vy[3] = { t.p0y, t.p1y, t.p2y }, vz[3] = { t.p0z, t.p1z, t.p2z };
        double sx[3], sy[3];
        bool finite = true;
        for (int k = 0; k < 3; k++) {
            const double qz = vz[k] - 7.0;
            sx[k] = cx + H * vx[k] / qz; sy[k] = cy + H * vy[k] / qz;
            finite = finite && sx[k] == sx[k] && sy[k] == sy[k] && std::fabs(sx[k]) < 1e12 && std::fabs(sy[k]) < 1e12;
        }
        if (!finite) return false;
        const double xa = std::min(sx[0], std::min(sx[1], sx[2])), xb = std::max(sx[0], std::max(sx[1], sx[2]));
        const double ya = std::min(sy[0], std::min(sy[1], sy[2])), yb = std::max(sy[0], std::max(sy[1], sy[2]));
        double fx0 = std::floor(xa - margin), fx1 = std::ceil(xb + margin), fy0 = std::floor(ya - margin), fy1 = std::ceil(yb + margin);
        if (fx1 < 0 || fy1 < 0 || fx0 > width - 1 || fy0 > height - 1) continue;      // off screen
        if (fx0 < 0) fx0 = 0;
        if (fy0 < 0) fy0 = 0;
        if (fx1 > width - 1) fx1 = width - 1;
        if (fy1 > height - 1) fy1 = height - 1;
        const int bx0 = (int)fx0 / bin_w, bx1 = (int)fx1 / bin_w, by0 = (int)fy0 / bin_h, by1 = (int)fy1 / bin_h;
        // edges of the projected triangle with outward normals (skipped when the projection is degenerate)
        const double area2 = (sx[1] - sx[0]) * (sy[2] - sy[0]) - (sx[2] - sx[0]) * (sy[1] - sy[0]);
        const double scale = std::fabs(xb - xa) + std::fabs(yb - ya) + 1.0;
        const bool use_edges = std::fabs(area2) > 1e-9 * scale * scale;
        for (int yb_ = by0; yb_ <= by1; yb_++) {
            for (int xb_ = bx0; xb_ <= bx1; xb_++) {
                bool separated = false;
                if (use_edges) {
                    // the bin's pixels, grown by the margin (a sample looks through a point less than half a pixel off its pixel)
                    const double rx0 = (double)(xb_ * bin_w) - margin, rx1 = (double)(xb_ * bin_w + bin_w - 1) + margin;
                    const double ry0 = (double)(yb_ * bin_h) - margin, ry1 = (double)(yb_ * bin_h + bin_h - 1) + margin;
                    for (int e = 0; e < 3 && !separated; e++) {
                        const int a = e, b2 = (e + 1) % 3, c = (e + 2) % 3;
                        double nx = sy[b2] - sy[a], ny = -(sx[b2] - sx[a]);            // normal of edge a -> b
                        if (nx * (sx[c] - sx[a]) + ny * (sy[c] - sy[a]) > 0) { nx = -nx; ny = -ny; }      // pointing away from the third vertex
                        const double len = std::sqrt(nx * nx + ny * ny);
                        if (!(len > 0)) continue;
                        // the rectangle's corner that is deepest along -n: if even it is outside, the whole rectangle is
                        const double px = nx > 0 ? rx0 : rx1, py = ny > 0 ? ry0 : ry1;
                        if ((nx * (px - sx[a]) + ny * (py - sy[a])) / len > 1e-6) separated = true;
                    }
                }
                if (!separated) items.push_back({ (int)i, yb_ * bx + xb_ });
            }
        }
    }
    off.assign((size_t)bx * (size_t)by + 1, 0u);
    for (const Item& it : items) off[(size_t)it.bin + 1]++;
    for (size_t i = 1; i < off.size(); i++) off[i] += off[i - 1];
    ent.assign(off.back() ? off.back() : 1, 0u);
    std::vector<uint32_t> fill(off.begin(), off.end() - 1);
    for (const Item& it : items) ent[fill[(size_t)it.bin]++] = (uint32_t)it.node;       // items are in ascending node order, so every bin's list is too
    return true;
}

// ---------------------------------------------------------------------------------------
// random stream + host-generated tables
// ---------------------------------------------------------------------------------------
namespace {
inline uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
inline uint32_t stream_key(uint32_t seed, uint32_t pixel, uint32_t sample)
{
    uint32_t h = mix32(seed ^ 0x9E3779B9u);
    h = mix32(h + pixel);
    return mix32(h + sample);
}
inline float uniform_from(uint32_t r31) { return (float)(int32_t)r31 / 2147483648.0f; }   // (float)rand() / RAND_MAX
}  // namespace

uint32_t rand31(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t counter)
{
    return mix32(stream_key(seed, pixel, sample) + counter) >> 1;
}

// entry i of the table RMath::InitPseudoRandomUnitVector would fill (Src/Math.cpp:24-31,
// Src/Math.h:34-40) when the 2i-th and (2i+1)-th rand() calls return the table stream's draws
void unit_table_entry(uint32_t index, float out3[3])
{
    const uint32_t key = stream_key(RTW_TABLE_SEED, 0xFFFFFFFFu, 0xFFFFFFFFu);
    const float r1 = uniform_from(mix32(key + 2u * index) >> 1);
    const float r2 = uniform_from(mix32(key + 2u * index + 1u) >> 1);
    const float t1 = 2.0f * 3.1415926f * r1;
    const float t2 = acosf(1.0f - 2.0f * r2);
    const float sin_t2 = sinf(t2);
    out3[0] = sinf(t1) * sin_t2;
    out3[1] = cosf(t1) * sin_t2;
    out3[2] = cosf(t2);
}

void fill_unit_table(float* dst, int threads)
{
    if (threads < 1) threads = 1;
    std::vector<std::thread> pool;
    const uint32_t n = RTW_TABLE_SIZE;
    for (int t = 0; t < threads; t++) {
        pool.emplace_back([=]() {
            const uint32_t lo = (uint32_t)((uint64_t)n * (uint64_t)t / (uint64_t)threads);
            const uint32_t hi = (uint32_t)((uint64_t)n * (uint64_t)(t + 1) / (uint64_t)threads);
            for (uint32_t i = lo; i < hi; i++) unit_table_entry(i, dst + (size_t)i * 3);
        });
    }
    for (auto& th : pool) th.join();
}

// MakePixelColor(LinearToGamma(c)) (Src/ColorBuffer.h:81-109) is a monotone staircase in c;
// thr[k] = smallest float whose 8-bit value is >= k.  The device resolves a channel with an
// 8-step search in this table instead of calling a device powf that would not match libm.
void gamma_thresholds(float out[256])
{
    const float exponent = 1.0f / 2.2f;
    out[0] = 0.0f;
    for (int k = 1; k < 256; k++) {
        uint32_t lo = 0, hi; float one = 1.0f; std::memcpy(&hi, &one, 4);
        while (hi - lo > 1) {
            const uint32_t mid = lo + (hi - lo) / 2;
            float c; std::memcpy(&c, &mid, 4);
            const int q = (int)(powf(c, exponent) * 255);
            if (q >= k) hi = mid; else lo = mid;
        }
        std::memcpy(&out[k], &hi, 4);
    }
}

// GammaToLinear of an 8-bit channel (Src/Texture.cpp:129-131, Src/ColorBuffer.h:70-78)
void texel_lut(float out[256])
{
    for (int i = 0; i < 256; i++) out[i] = powf((float)i / 255, 2.2f);
}

}  // namespace rtw
