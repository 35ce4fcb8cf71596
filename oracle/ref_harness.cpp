// ref_harness.cpp -- headless driver that is LINKED AGAINST THE REFERENCE'S OWN
// TRANSLATION UNITS (compiled in place from /root/reference by oracle/Makefile,
// outputs only under oracle/_ref/).  Nothing from the reference is copied here:
// this file only calls the reference's public (and, for introspection, private)
// members and re-states the 30-line camera loop of ThreadWorker_Render
// (Src/RayTracerProgram.cpp:131-188) with run-time width/height, because the
// original is hard-wired to 800x800.
//
// TEST INFRASTRUCTURE ONLY.  It generates tests/golden/* (via
// tests/golden/make_golden.py) and can serve as the "reference" CPU baseline.
//
// Determinism: rand()/srand() are defined HERE, so every rand()-driven decision
// in the reference TUs (AA jitter, Blend choice, alpha test, fuzzy reflection,
// the unit-vector table contents) is replayed from the same counter-based
// generator the oracle uses.  The table cursor (function-local static in
// Src/Math.cpp:35-39) cannot be set, but it can be advanced by calling the public
// RMath::PseudoRandomUnitVector(), and located by comparing what it returns with
// the table entry values, which are a pure function of the index.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <thread>
#include <chrono>
#include <atomic>

#define private public
#define protected public
#include "MeshShape.h"
#include "RayTracerScene.h"
#include "RayTracerProgram.h"
#undef private
#undef protected
#include "Math.h"
#include "Texture.h"
#include "ColorBuffer.h"
#include "ThreadTaskQueue.h"
#include "rt_oracle.h"   // only for the orc_material_node layout

// ---------------------------------------------------------------------------
// interposed rand()
// ---------------------------------------------------------------------------
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
static const uint32_t kTableSize = 0xFFFFFFu, kTableStride = 16u, kTableSeed = 0x52544142u;
static inline uint32_t table_phase(uint32_t seed) { return mix32(seed ^ 0x7AB1E5u) % kTableSize; }

enum RandMode { RM_ZERO, RM_KEYED, RM_THREAD_LOCAL, RM_GLIBC_LIKE, RM_RECORD, RM_SCRIPT };
static std::vector<int> g_script;      // RM_RECORD: the keyed stream's values in call order; RM_SCRIPT: played back to the reference's own loop
static size_t g_script_pos = 0;
static RandMode g_mode = RM_ZERO;
static thread_local uint32_t t_key = 0, t_counter = 0;
static thread_local uint64_t t_xs = 0x9E3779B97F4A7C15ull;
static uint64_t g_locked_state = 12345;
static std::mutex g_rand_mutex;

extern "C" int rand(void)
{
    switch (g_mode) {
    case RM_ZERO: return 0;
    case RM_KEYED: return (int)(mix32(t_key + t_counter++) >> 1);
    case RM_RECORD: { const int v = (int)(mix32(t_key + t_counter++) >> 1); g_script.push_back(v); return v; }
    case RM_SCRIPT: {
        if (g_script_pos >= g_script.size()) { fprintf(stderr, "harness: the reference's loop drew more numbers than the recorded pass\n"); exit(6); }
        return g_script[g_script_pos++];
    }
    case RM_THREAD_LOCAL: {
        t_xs ^= t_xs << 13; t_xs ^= t_xs >> 7; t_xs ^= t_xs << 17;
        return (int)((t_xs >> 33) & 0x7FFFFFFF);
    }
    case RM_GLIBC_LIKE: {   // one process-wide lock per call, like glibc's rand()
        std::lock_guard<std::mutex> l(g_rand_mutex);
        g_locked_state = g_locked_state * 6364136223846793005ull + 1442695040888963407ull;
        return (int)((g_locked_state >> 33) & 0x7FFFFFFF);
    }
    }
    return 0;
}
extern "C" void srand(unsigned) {}

static void set_key(uint32_t key) { t_key = key; t_counter = 0; }

// table entry i, computed by the reference's own inline RandomUnitVector()
static RVec3 table_entry(uint32_t index)
{
    RandMode m = g_mode; uint32_t k = t_key, c = t_counter;
    g_mode = RM_KEYED; t_key = path_key(kTableSeed, 0xFFFFFFFFu, 0xFFFFFFFFu); t_counter = 2u * index;
    RVec3 v = RMath::RandomUnitVector();
    g_mode = m; t_key = k; t_counter = c;
    return v;
}

static uint64_t g_cursor = 0;   // index the reference's cursor will return next (mod table size)
static bool g_table_ready = false;
static void init_table()
{
    if (g_table_ready) return;
    g_mode = RM_KEYED; set_key(path_key(kTableSeed, 0xFFFFFFFFu, 0xFFFFFFFFu));
    RMath::InitPseudoRandomUnitVector();
    g_mode = RM_ZERO;                       // RandRangedInt(0, Max) -> 0 : cursor starts at 0
    RVec3 v = RMath::PseudoRandomUnitVector();
    RVec3 e = table_entry(0);
    if (memcmp(&v, &e, 12) != 0) { fprintf(stderr, "harness: table entry 0 mismatch\n"); exit(3); }
    g_cursor = 1;
    g_table_ready = true;
}
static void cursor_advance_to(uint64_t target)
{
    target %= kTableSize;
    uint64_t n = (target + kTableSize - (g_cursor % kTableSize)) % kTableSize;
    for (uint64_t i = 0; i < n; i++) RMath::PseudoRandomUnitVector();
    g_cursor = target;
}
// after a path that started at `base`: find how many entries it consumed
static void cursor_relocate(uint64_t base)
{
    RandMode m = g_mode; g_mode = RM_ZERO;
    RVec3 v = RMath::PseudoRandomUnitVector();
    g_mode = m;
    for (uint32_t k = 0; k < 4096; k++) {
        RVec3 e = table_entry((uint32_t)((base + k) % kTableSize));
        if (memcmp(&v, &e, 12) == 0) { g_cursor = (base + k + 1) % kTableSize; return; }
    }
    fprintf(stderr, "harness: lost the table cursor\n"); exit(4);
}

// ---------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------
template <typename T> static std::vector<T> read_file(const char* path, size_t count)
{
    std::vector<T> v(count);
    FILE* f = fopen(path, "rb");
    if (!f || fread(v.data(), sizeof(T), count, f) != count) { fprintf(stderr, "harness: cannot read %s\n", path); exit(2); }
    fclose(f);
    return v;
}
static void write_file(const std::string& path, const void* p, size_t bytes)
{
    FILE* f = fopen(path.c_str(), "wb");
    if (!f || fwrite(p, 1, bytes, f) != bytes) { fprintf(stderr, "harness: cannot write %s\n", path.c_str()); exit(2); }
    fclose(f);
}

static std::unique_ptr<ISurfaceMaterial> build_material(const std::vector<orc_material_node>& n, int i)
{
    const orc_material_node& m = n[i];
    RVec3 c(m.r, m.g, m.b);
    switch (m.type) {
    case ORC_MAT_DIFFUSE: return std::unique_ptr<ISurfaceMaterial>(new SurfaceMaterial_Diffuse(c));
    case ORC_MAT_DIFFUSE_CHECKER: return std::unique_ptr<ISurfaceMaterial>(new SurfaceMaterial_DiffuseChecker(c, m.param));
    case ORC_MAT_REFLECTIVE: return std::unique_ptr<ISurfaceMaterial>(new SurfaceMaterial_Reflective(c, m.param));
    case ORC_MAT_EMISSIVE: return std::unique_ptr<ISurfaceMaterial>(new SurfaceMaterial_Emissive(c));
    case ORC_MAT_BLEND: return std::unique_ptr<ISurfaceMaterial>(new SurfaceMaterial_Blend(build_material(n, m.child_a), build_material(n, m.child_b), m.param));
    case ORC_MAT_COMBINE: return std::unique_ptr<ISurfaceMaterial>(new SurfaceMaterial_Combine(build_material(n, m.child_a), build_material(n, m.child_b)));
    default: return std::unique_ptr<ISurfaceMaterial>(new SurfaceMaterial_Null());
    }
}
static std::vector<orc_material_node> load_material(const char* path)
{
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "harness: cannot read %s\n", path); exit(2); }
    fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<orc_material_node> v((size_t)sz / sizeof(orc_material_node));
    if (fread(v.data(), sizeof(orc_material_node), v.size(), f) != v.size()) exit(2);
    fclose(f);
    return v;
}

static RayTracerProgram* g_program = nullptr;
static RMeshShape* g_mesh = nullptr;
static bool g_scene_file = false;

static std::unique_ptr<ISurfaceMaterial> material_arg(const char* matfile, bool dash_is_diffuse)
{
    if (matfile && strcmp(matfile, "none") == 0) return std::unique_ptr<ISurfaceMaterial>();
    if (matfile && strcmp(matfile, "-") != 0) return build_material(load_material(matfile), 0);
    if (!dash_is_diffuse) return std::unique_ptr<ISurfaceMaterial>();
    return std::unique_ptr<ISurfaceMaterial>(new SurfaceMaterial_Diffuse(RVec3(1, 1, 1)));
}

// A ".scene" file lists the shapes in insertion order, one per line (numbers are decimal renderings of exact floats):
//   sphere cx cy cz r MAT | plane nx ny nz px py pz MAT | capsule sx sy sz ex ey ez r MAT | triangle 9 numbers MAT | mesh OBJ MAT
// MAT = a material-node file, "-" (Diffuse(1,1,1)) or "none" (no material).  The shapes are made by the reference's own
// RSphere / RPlane / RCapsule / RMeshShape::Create and added with RayTracerScene::AddShape.
static void setup_scene_file(const char* path)
{
    FILE* f = fopen(path, "r");
    if (!f) { fprintf(stderr, "harness: cannot read %s\n", path); exit(2); }
    char kind[32], a[4096], m[4096];
    while (fscanf(f, "%31s", kind) == 1) {
        double v[7];
        if (!strcmp(kind, "sphere")) {
            if (fscanf(f, "%lf %lf %lf %lf %4095s", &v[0], &v[1], &v[2], &v[3], m) != 5) exit(2);
            g_program->GetScene()->AddShape(RSphere::Create(RVec3((float)v[0], (float)v[1], (float)v[2]), (float)v[3]), material_arg(m, true));
        } else if (!strcmp(kind, "plane")) {
            if (fscanf(f, "%lf %lf %lf %lf %lf %lf %4095s", &v[0], &v[1], &v[2], &v[3], &v[4], &v[5], m) != 7) exit(2);
            g_program->GetScene()->AddShape(RPlane::Create(RVec3((float)v[0], (float)v[1], (float)v[2]), RVec3((float)v[3], (float)v[4], (float)v[5])), material_arg(m, true));
        } else if (!strcmp(kind, "capsule")) {
            if (fscanf(f, "%lf %lf %lf %lf %lf %lf %lf %4095s", &v[0], &v[1], &v[2], &v[3], &v[4], &v[5], &v[6], m) != 8) exit(2);
            g_program->GetScene()->AddShape(RCapsule::Create(RVec3((float)v[0], (float)v[1], (float)v[2]), RVec3((float)v[3], (float)v[4], (float)v[5]), (float)v[6]), material_arg(m, true));
        } else if (!strcmp(kind, "triangle")) {
            double w[9];
            if (fscanf(f, "%lf %lf %lf %lf %lf %lf %lf %lf %lf %4095s", &w[0], &w[1], &w[2], &w[3], &w[4], &w[5], &w[6], &w[7], &w[8], m) != 10) exit(2);
            g_program->GetScene()->AddShape(RTriangle::Create(RVec3((float)w[0], (float)w[1], (float)w[2]), RVec3((float)w[3], (float)w[4], (float)w[5]),
                                                              RVec3((float)w[6], (float)w[7], (float)w[8])), material_arg(m, true));
        } else if (!strcmp(kind, "mesh")) {
            if (fscanf(f, "%4095s %4095s", a, m) != 2) exit(2);
            auto mesh = RMeshShape::Create(a);
            g_mesh = mesh.get();
            g_program->GetScene()->AddShape(std::move(mesh), material_arg(m, true));
        } else { fprintf(stderr, "harness: bad scene line '%s'\n", kind); exit(2); }
    }
    fclose(f);
}

static void setup_scene(const char* obj, const char* matfile)
{
    g_program = new RayTracerProgram();       // sets CurrentInstance; no window is opened
    const size_t L = strlen(obj);
    if (L > 6 && strcmp(obj + L - 6, ".scene") == 0) { g_scene_file = true; setup_scene_file(obj); return; }
    auto mesh = RMeshShape::Create(obj);
    g_mesh = mesh.get();
    g_program->GetScene()->AddShape(std::move(mesh), material_arg(matfile, true));
}

static void dump_tree(const KdNode* n, std::vector<float>& bounds, std::vector<int>& tri)
{
    if (!n) return;
    bounds.push_back(n->Bounds.pMin.x); bounds.push_back(n->Bounds.pMin.y); bounds.push_back(n->Bounds.pMin.z);
    bounds.push_back(n->Bounds.pMax.x); bounds.push_back(n->Bounds.pMax.y); bounds.push_back(n->Bounds.pMax.z);
    tri.push_back((n->Left || n->Right) ? -1 : n->Triangle.Index);
    dump_tree(n->Left.get(), bounds, tri);
    dump_tree(n->Right.get(), bounds, tri);
}

// camera loop of ThreadWorker_Render with run-time W/H and ns sub-samples
struct FrameParams { int W, H, ns, depth, preview; uint32_t seed; };

static bool g_no_table = false;     // the scene's materials never read the unit-vector table (preview, mirrors): leave the cursor alone
static RVec3 render_pixel(const FrameParams& fp, int PixelIndex, int pass, bool keyed)
{
    const RVec3 ViewPoint(0, 0, 7.0f);
    const RayTracerScene* Scene = g_program->GetScene();
    const float Aspect = (float)fp.W / (float)fp.H;
    RenderOption opt; opt.UseBaseColor = fp.preview != 0;
    int x = PixelIndex % fp.W, y = PixelIndex / fp.W;
    float dx = -(float)(x - fp.W / 2) / (fp.W * 2) * Aspect;
    float dy = -(float)(y - fp.H / 2) / (fp.H * 2);
    RVec3 c = RVec3::Zero();
    const float inv_pixel_radius = 1.0f / (fp.W * 4);
    const float ox[4] = { 0.0f, inv_pixel_radius, 0.0f, inv_pixel_radius };
    const float oy[4] = { 0.0f, 0.0f, inv_pixel_radius, inv_pixel_radius };
    const float offset_radius = inv_pixel_radius * 0.5f;
    const uint64_t npix = (uint64_t)fp.W * fp.H;
    const uint32_t phase = table_phase(fp.seed);
    for (int i = 0; i < fp.ns; i++) {
        uint64_t base = 0;
        if (keyed) {
            set_key(path_key(fp.seed, (uint32_t)PixelIndex, (uint32_t)(pass * 4 + i)));
            base = (((uint64_t)pass * npix + (uint64_t)PixelIndex) * 4u + (uint64_t)i) * kTableStride + phase;
            if (!g_no_table) cursor_advance_to(base);
        }
        float offset_x = ox[i];
        float offset_y = oy[i];
        offset_x += (RMath::Random() - 0.5f) * offset_radius;
        offset_y += (RMath::Random() - 0.5f) * offset_radius;
        RVec3 Dir(dx + offset_x, dy + offset_y, -0.5f);
        RRay ray(ViewPoint, Dir.GetNormalizedVec3(), 1000.0f);
        c += Scene->RayTrace(ray, fp.depth, opt);
        if (keyed && !g_no_table) cursor_relocate(base);
    }
    c /= (float)fp.ns;
    return c;
}

struct AccPixel { RVec3 sum; int n; };

// The reference's own pixel-range worker and the buffers it writes (external linkage, Src/RayTracerProgram.cpp:49,77,131).
// AccumulatePixel is defined in that .cpp only; this declaration has its two data members in the same order (16 bytes).
void ThreadWorker_Render(int begin, int end, int MaxBounceCount, const RenderOption& InOption);
void FormatTimeString(char* Buffer, int BufferSize, int Milliseconds);
extern Pixel bitcolor[];
struct AccumulatePixel { RVec3 AccumulatedColor; int Num; };
extern AccumulatePixel accuBuffer[];

struct TimeTask { int Start, End; };
typedef ThreadTaskQueue<TimeTask> TimeQueue;

int main(int argc, char** argv)
{
    if (argc < 2) { fprintf(stderr, "usage: ref_harness <cmd> ...\n"); return 1; }
    std::string cmd = argv[1];

    if (cmd == "dump_mesh" && argc == 4) {
        setup_scene(argv[2], "-");
        std::string pre = argv[3];
        RMeshShape* m = g_mesh;
        int counts[8] = { (int)m->Points.size(), (int)m->Texcoords.size(), (int)m->Normals.size(),
                          (int)m->PointIndices.size() / 3, 0, 0, (int)m->Textures.size(), 0 };
        std::vector<float> bounds; std::vector<int> tri;
        if (m->Spatial) dump_tree(m->Spatial->RootNode.get(), bounds, tri);
        counts[5] = (int)tri.size();
        int maxmat = -1; for (int id : m->PolyMaterialId) if (id > maxmat) maxmat = id;
        counts[4] = maxmat + 1;
        write_file(pre + ".counts.i32", counts, sizeof counts);
        write_file(pre + ".points.f32", m->Points.data(), m->Points.size() * 12);
        write_file(pre + ".texcoords.f32", m->Texcoords.data(), m->Texcoords.size() * 12);
        write_file(pre + ".normals.f32", m->Normals.data(), m->Normals.size() * 12);
        write_file(pre + ".pidx.i32", m->PointIndices.data(), m->PointIndices.size() * 4);
        write_file(pre + ".tidx.i32", m->TexcoordIndices.data(), m->TexcoordIndices.size() * 4);
        write_file(pre + ".nidx.i32", m->NormalIndices.data(), m->NormalIndices.size() * 4);
        write_file(pre + ".matid.i32", m->PolyMaterialId.data(), m->PolyMaterialId.size() * 4);
        write_file(pre + ".tree_bounds.f32", bounds.data(), bounds.size() * 4);
        write_file(pre + ".tree_tri.i32", tri.data(), tri.size() * 4);
        float sb[6] = { m->GetBounds().pMin.x, m->GetBounds().pMin.y, m->GetBounds().pMin.z,
                        m->GetBounds().pMax.x, m->GetBounds().pMax.y, m->GetBounds().pMax.z };
        write_file(pre + ".shape_bounds.f32", sb, sizeof sb);
        std::vector<int> texinfo;
        for (size_t i = 0; i < m->Textures.size() && i < 64; i++) {
            RTexture* t = m->Textures[i].get();
            texinfo.push_back(t ? t->Width : 0); texinfo.push_back(t ? t->Height : 0);
        }
        write_file(pre + ".texinfo.i32", texinfo.data(), texinfo.size() * 4);
        return 0;
    }

    if (cmd == "closest" && argc == 6) {
        setup_scene(argv[2], "-");
        size_t n = (size_t)atoll(argv[4]);
        auto rays = read_file<float>(argv[3], n * 7);
        std::vector<float> out(n * 13);
        for (size_t i = 0; i < n; i++) {
            RRay r(RVec3(rays[i * 7], rays[i * 7 + 1], rays[i * 7 + 2]), RVec3(rays[i * 7 + 3], rays[i * 7 + 4], rays[i * 7 + 5]), rays[i * 7 + 6]);
            RayHitResult h;
            int s = g_program->GetScene()->FindIntersectionWithScene(r, h);
            int tri = -1;
            if (g_scene_file) tri = -2;     // several shapes: the triangle index is not recorded
            else if (s >= 0) {   // same query again, straight at the tree, to learn the triangle index
                RayHitResult h2;
                g_mesh->Spatial->TestRayIntersection(r, g_mesh->Points.data(), &h2, &tri);
            }
            float* o = &out[i * 13];
            o[0] = h.HitPosition.x; o[1] = h.HitPosition.y; o[2] = h.HitPosition.z;
            o[3] = h.HitNormal.x; o[4] = h.HitNormal.y; o[5] = h.HitNormal.z; o[6] = h.Distance;
            o[7] = h.SampledColor.x; o[8] = h.SampledColor.y; o[9] = h.SampledColor.z; o[10] = h.SampledAlpha;
            memcpy(&o[11], &s, 4); memcpy(&o[12], &tri, 4);
        }
        write_file(argv[5], out.data(), out.size() * 4);
        return 0;
    }

    if (cmd == "texsample" && argc == 7) {
        setup_scene(argv[2], "-");
        int mat = atoi(argv[3]);
        size_t n = (size_t)atoll(argv[5]);
        auto uv = read_file<float>(argv[4], n * 2);
        if (mat < 0 || mat >= (int)g_mesh->Textures.size() || !g_mesh->Textures[mat]) { fprintf(stderr, "no texture %d\n", mat); return 5; }
        std::vector<float> out(n * 4);
        for (size_t i = 0; i < n; i++) {
            RVec4 s = g_mesh->Textures[mat]->Sample(uv[i * 2], uv[i * 2 + 1]);
            out[i * 4] = s.x; out[i * 4 + 1] = s.y; out[i * 4 + 2] = s.z; out[i * 4 + 3] = s.w;
        }
        write_file(argv[6], out.data(), out.size() * 4);
        return 0;
    }

    // frame OBJ MAT W H NS DEPTH PREVIEW SEED PASS0 NPASS BEGIN END OUT
    if (cmd == "frame" && argc == 15) {
        setup_scene(argv[2], argv[3]);
        FrameParams fp; fp.W = atoi(argv[4]); fp.H = atoi(argv[5]); fp.ns = atoi(argv[6]); fp.depth = atoi(argv[7]);
        fp.preview = atoi(argv[8]); fp.seed = (uint32_t)strtoul(argv[9], nullptr, 0);
        int pass0 = atoi(argv[10]), npass = atoi(argv[11]), begin = atoi(argv[12]), end = atoi(argv[13]);
        init_table();
        g_mode = RM_KEYED;
        size_t n = (size_t)(end - begin + 1);
        std::vector<AccPixel> acc(n); for (auto& a : acc) { a.sum = RVec3(0, 0, 0); a.n = 0; }
        std::vector<uint32_t> argb(n);
        for (int pass = pass0; pass < pass0 + npass; pass++) {
            for (int p = begin; p <= end; p++) {
                RVec3 c = render_pixel(fp, p, pass, true);
                AccPixel& a = acc[(size_t)(p - begin)];
                if (fp.preview) {
                    a.sum = c; a.n = 1;
                    argb[(size_t)(p - begin)] = MakePixelColor(LinearToGamma(c));
                } else {
                    a.sum += c; a.n++;                                              // AccumulatePixel::AddPixel
                    argb[(size_t)(p - begin)] = MakePixelColor(LinearToGamma(a.sum / (float)a.n));   // GetGammaSpacePixel
                }
            }
        }
        std::vector<float> out(n * 4);
        for (size_t i = 0; i < n; i++) { out[i * 4] = acc[i].sum.x; out[i * 4 + 1] = acc[i].sum.y; out[i * 4 + 2] = acc[i].sum.z; out[i * 4 + 3] = (float)acc[i].n; }
        std::string o = argv[14];
        write_file(o + ".accum.f32", out.data(), out.size() * 4);
        write_file(o + ".argb.u32", argb.data(), argb.size() * 4);
        return 0;
    }

    // raytrace OBJ MAT RAYS KEYS N DEPTH PREVIEW SEED W H OUT
    if (cmd == "raytrace" && argc == 13) {
        setup_scene(argv[2], argv[3]);
        size_t n = (size_t)atoll(argv[6]);
        auto rays = read_file<float>(argv[4], n * 7);
        auto keys = read_file<uint32_t>(argv[5], n * 2);
        int depth = atoi(argv[7]); RenderOption opt; opt.UseBaseColor = atoi(argv[8]) != 0;
        uint32_t seed = (uint32_t)strtoul(argv[9], nullptr, 0);
        uint64_t npix = (uint64_t)atoi(argv[10]) * (uint64_t)atoi(argv[11]);
        init_table();
        g_mode = RM_KEYED;
        const uint32_t phase = table_phase(seed);
        std::vector<float> out(n * 3);
        for (size_t i = 0; i < n; i++) {
            uint32_t pixel = keys[i * 2], sample = keys[i * 2 + 1];
            set_key(path_key(seed, pixel, sample));
            uint64_t base = (((uint64_t)(sample / 4) * npix + pixel) * 4u + (sample % 4)) * kTableStride + phase;
            cursor_advance_to(base);
            RRay r(RVec3(rays[i * 7], rays[i * 7 + 1], rays[i * 7 + 2]), RVec3(rays[i * 7 + 3], rays[i * 7 + 4], rays[i * 7 + 5]), rays[i * 7 + 6]);
            RVec3 c = g_program->GetScene()->RayTrace(r, depth, opt);
            cursor_relocate(base);
            out[i * 3] = c.x; out[i * 3 + 1] = c.y; out[i * 3 + 2] = c.z;
        }
        write_file(argv[12], out.data(), out.size() * 4);
        return 0;
    }

    // time OBJ MAT W H NS DEPTH THREADS PASSES RANDMODE(tl|glibc) [ROWS_LIMIT]
    // One pass = all pixels through the reference's ThreadTaskQueue in 10-row tasks
    // (Src/RayTracerProgram.cpp:282,294-301).  ROWS_LIMIT bounds the sample to the
    // centre band of rows (0 = whole frame).
    if (cmd == "time" && (argc == 11 || argc == 12)) {
        setup_scene(argv[2], argv[3]);
        FrameParams fp; fp.W = atoi(argv[4]); fp.H = atoi(argv[5]); fp.ns = atoi(argv[6]); fp.depth = atoi(argv[7]);
        fp.preview = 0; fp.seed = 12345;
        int threads = atoi(argv[8]), passes = atoi(argv[9]);
        if (threads <= 0) threads = (int)std::thread::hardware_concurrency();
        if (threads <= 0) threads = 1;
        int rows_limit = argc == 12 ? atoi(argv[11]) : 0;
        int row0 = 0, row1 = fp.H;
        if (rows_limit > 0 && rows_limit < fp.H) { row0 = (fp.H - rows_limit) / 2; row1 = row0 + rows_limit; }
        g_mode = RM_THREAD_LOCAL;
        RMath::InitPseudoRandomUnitVector();
        g_mode = strcmp(argv[10], "glibc") == 0 ? RM_GLIBC_LIKE : RM_THREAD_LOCAL;
        std::vector<AccPixel> acc((size_t)fp.W * fp.H);
        TimeQueue& q = TimeQueue::Get();
        std::atomic<int> remaining(0); std::atomic<bool> quit(false);
        std::vector<std::thread> workers;
        for (int t = 0; t < threads; t++) {
            workers.emplace_back([&, t]() {
                t_xs = 0x9E3779B97F4A7C15ull * (uint64_t)(t + 1);
                for (;;) {
                    TimeTask task;
                    {
                        std::unique_lock<std::mutex> lk(q.GetMutex());
                        q.GetWorkerThreadCondition().wait(lk, [&] { return q.GetNumTasks() > 0 || quit.load(); });
                        if (quit.load() && q.GetNumTasks() == 0) return;
                        q.PopTask(&task);
                    }
                    for (int p = task.Start; p <= task.End; p++) {
                        RVec3 c = render_pixel(fp, p, 0, false);
                        acc[(size_t)p].sum += c; acc[(size_t)p].n++;
                    }
                    remaining.fetch_sub(1);
                    q.NotifySingleTaskDone();
                }
            });
        }
        const int NumTaskRows = 10;
        double best = 1e30, total = 0;
        for (int pass = 0; pass < passes; pass++) {
            int ntasks = 0; for (int i = row0; i < row1; i += NumTaskRows) ntasks++;
            remaining.store(ntasks);
            auto t0 = std::chrono::steady_clock::now();
            for (int i = row0; i < row1; i += NumTaskRows) {
                TimeTask t; t.Start = i * fp.W;
                int e = (i + NumTaskRows) * fp.W - 1; int mx = row1 * fp.W - 1;
                t.End = e < mx ? e : mx;
                q.PushTask(t);
            }
            while (remaining.load() > 0) std::this_thread::sleep_for(std::chrono::microseconds(200));
            double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            total += s; if (s < best) best = s;
        }
        quit.store(true);
        q.GetWorkerThreadCondition().notify_all();
        for (auto& w : workers) w.join();
        double chk = 0; for (auto& a : acc) chk += a.sum.x;
        printf("{\"threads\": %d, \"passes\": %d, \"rows\": %d, \"pixels\": %lld, \"best_s\": %.6f, \"mean_s\": %.6f, \"rand\": \"%s\", \"checksum\": %.3f}\n",
               threads, passes, row1 - row0, (long long)(row1 - row0) * fp.W, best, total / passes, argv[10], chk);
        return 0;
    }

    // worker OBJ MAT DEPTH PREVIEW SEED PASS0 NPASS BEGIN END OUT
    // The reference's OWN ThreadWorker_Render (800 x 800, four sub-samples: compile-time constants) over pixels BEGIN..END for NPASS
    // passes, writing its own accuBuffer[] / bitcolor[].  The loop cannot be told which (pixel, sample) it is at, so each pass is
    // first run through this file's restatement with the keyed generator RECORDING every value rand() returns, and the reference's
    // loop then gets exactly that sequence played back; it must consume it to the last number.  Only for scenes whose materials never
    // read the unit-vector table (a preview pass, mirrors): that table's cursor cannot be set from inside the reference's loop.
    if (cmd == "worker" && argc == 12) {
        setup_scene(argv[2], argv[3]);
        FrameParams fp; fp.W = bitmapWidth; fp.H = bitmapHeight; fp.ns = 4; fp.depth = atoi(argv[4]);
        fp.preview = atoi(argv[5]); fp.seed = (uint32_t)strtoul(argv[6], nullptr, 0);
        int pass0 = atoi(argv[7]), npass = atoi(argv[8]), begin = atoi(argv[9]), end = atoi(argv[10]);
        if (begin < 0 || end >= bitmapWidth * bitmapHeight || end < begin) { fprintf(stderr, "harness: bad range\n"); return 2; }
        RenderOption opt; opt.UseBaseColor = fp.preview != 0;
        g_no_table = true;
        double restated = 0;
        for (int pass = pass0; pass < pass0 + npass; pass++) {
            g_script.clear(); g_script_pos = 0;
            g_mode = RM_RECORD;
            for (int p = begin; p <= end; p++) { RVec3 c = render_pixel(fp, p, pass, true); restated += c.x; }
            g_mode = RM_SCRIPT;
            ThreadWorker_Render(begin, end, fp.depth, opt);
            if (g_script_pos != g_script.size()) { fprintf(stderr, "harness: the reference's loop drew %zu numbers, the recorded pass %zu\n", g_script_pos, g_script.size()); return 6; }
        }
        g_mode = RM_ZERO;
        const size_t n = (size_t)(end - begin + 1);
        std::vector<float> out(n * 4);
        for (size_t i = 0; i < n; i++) {
            const AccumulatePixel& a = accuBuffer[(size_t)begin + i];
            out[i * 4] = a.AccumulatedColor.x; out[i * 4 + 1] = a.AccumulatedColor.y; out[i * 4 + 2] = a.AccumulatedColor.z; out[i * 4 + 3] = (float)a.Num;
        }
        std::string o = argv[11];
        write_file(o + ".accum.f32", out.data(), out.size() * 4);
        write_file(o + ".argb.u32", &bitcolor[begin], n * sizeof(Pixel));
        return 0;
    }

    // savepng ARGB W H OUT.png : RTexture::SaveBufferToPNG (Src/Texture.cpp:201-283) of a 0xAARRGGBB buffer
    if (cmd == "savepng" && argc == 6) {
        const int w = atoi(argv[3]), h = atoi(argv[4]);
        auto px = read_file<uint32_t>(argv[2], (size_t)w * h);
        return RTexture::SaveBufferToPNG(argv[5], px.data(), w, h) ? 0 : 7;
    }

    // timestring MS... : the reference's own FormatTimeString (Src/RayTracerProgram.cpp:242-268, external linkage), one line per argument
    if (cmd == "timestring" && argc >= 3) {
        for (int i = 2; i < argc; i++) {
            char buf[256];
            FormatTimeString(buf, (int)sizeof buf, atoi(argv[i]));
            printf("%s %s\n", argv[i], buf);
        }
        return 0;
    }

    fprintf(stderr, "ref_harness: bad command line\n");
    return 1;
}
