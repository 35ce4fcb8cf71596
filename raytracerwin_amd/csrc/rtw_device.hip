// rtw_device.hip -- HIP kernels for gfx950 (CDNA4) and their launch wrappers.
//
// Compile with -ffp-contract=off and correctly rounded fp32 divide/sqrt (hipcc default):
// every float expression below keeps the operand order of the reference so that the GPU
// result is bit-identical to the CPU oracle.  No MFMA: there is no dense contraction on
// this path; the hot loop is a stack-free walk over 32-byte nodes (one ray per lane).
#include <hip/hip_runtime.h>
#include <float.h>
#include <time.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <cstring>

#include "rtw_types.h"
#include "rtw_device.h"

#ifdef RTW_TIMING
__device__ unsigned long long g_rtw_timing[12 * 16384];
#endif

namespace {

struct f3 { float x, y, z; };
__device__ __forceinline__ f3 mk(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ f3 operator/(f3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ f3 cross(f3 a, f3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ bool near_zero(float a) { return fabsf(a) < FLT_EPSILON; }   // FLT_EQUAL_ZERO
__device__ __forceinline__ f3 reflect(f3 v, f3 n) { return v - (n * 2.0f) * dot(v, n); } // Src/RVector.h:218
__device__ __forceinline__ bool all_nonzero(f3 a) { return !near_zero(a.x) && !near_zero(a.y) && !near_zero(a.z); } // :142
__device__ __forceinline__ float ref_min(float a, float b) { return (a < b) ? a : b; }  // Math::Min
__device__ __forceinline__ float ref_max(float a, float b) { return (a > b) ? a : b; }  // Math::Max

__device__ __forceinline__ f3 normalized(f3 v)          // RVec3::GetNormalizedVec3
{
    float sq = v.x * v.x + v.y * v.y + v.z * v.z;
    if (!near_zero(sq)) { float inv = 1.0f / sqrtf(sq); v.x *= inv; v.y *= inv; v.z *= inv; }
    return v;
}
__device__ __forceinline__ float q_rsqrt(float number)  // Math::Q_rsqrt (Src/MathHelper.cpp:26-38)
{
    const float x2 = number * 0.5f;
    float f = __uint_as_float(0x5f3759dfu - (__float_as_uint(number) >> 1));
    f *= (1.5f - (x2 * f * f));
    return f;
}
__device__ __forceinline__ f3 normalized_fast(f3 v)     // RVec3::GetNormalizedVec3_Fast
{
    float sq = v.x * v.x + v.y * v.y + v.z * v.z;
    if (!near_zero(sq)) { float inv = q_rsqrt(sq); v.x *= inv; v.y *= inv; v.z *= inv; }
    return v;
}

// ---- random stream (specification shared with the oracle) --------------------------------
__device__ __forceinline__ uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__device__ __forceinline__ uint32_t stream_key(uint32_t seed, uint32_t pixel, uint32_t sample)
{
    uint32_t h = mix32(seed ^ 0x9E3779B9u);
    h = mix32(h + pixel);
    return mix32(h + sample);
}
struct PathRng {
    uint32_t key, counter;
    uint64_t table_base;    // first unit-table index of the path, phase included, before the modulo
    uint32_t table_reads;
    // the unit-table entry the NEXT hemisphere draw will read, fetched ahead of the shading chain (valid while
    // pre_reads == table_reads); a latency optimisation only: the value is the one the draw would load
    uint32_t pre_reads; float pre_x, pre_y, pre_z;
    __device__ __forceinline__ float random()            // RMath::Random (Src/Math.h:17-20)
    {
        uint32_t r = mix32(key + counter++) >> 1;
        return (float)(int32_t)r / 2147483648.0f;
    }
};
__device__ __forceinline__ uint32_t table_phase(uint32_t seed) { return mix32(seed ^ 0x7AB1E5u) % RTW_TABLE_SIZE; }   // 32-bit: fine
__device__ __forceinline__ void rng_init(PathRng& r, uint32_t seed, uint32_t phase, uint64_t npix, uint32_t pixel, uint32_t pass, uint32_t sub)
{
    r.key = stream_key(seed, pixel, pass * 4u + sub);
    r.counter = 0;
    r.table_base = (((uint64_t)pass * npix + (uint64_t)pixel) * 4u + (uint64_t)sub) * RTW_TABLE_STRIDE + phase;
    r.table_reads = 0;
    r.pre_reads = 0xFFFFFFFFu; r.pre_x = r.pre_y = r.pre_z = 0.0f;
}

struct Counters { uint32_t rays, boxes, tris, hits, tex, cams; };

// Loads/stores with an explicit address space.  Pointers that come out of the scene descriptor or out of a
// struct are "generic" to the compiler, which then emits flat_* instructions (slower, and they tie the LDS
// and vector-memory counters together); these helpers give it global_load_dwordx4 / ds_read_* instead.
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(1))) const float4 rtw_g_f4;
typedef __attribute__((address_space(1))) const float rtw_g_f1;
typedef __attribute__((address_space(3))) const float rtw_l_f1;
typedef __attribute__((address_space(3))) const uint32_t rtw_l_u1;
typedef __attribute__((address_space(3))) uint32_t rtw_l_u1w;
typedef __attribute__((address_space(4))) const float4 rtw_c_f4;
typedef __attribute__((address_space(4))) const uint32_t rtw_c_u1;
__device__ __forceinline__ uint32_t cldu(const uint32_t* p, int i) { return ((rtw_c_u1*)p)[i]; }
// wave-uniform address in read-only memory: becomes an s_load (scalar cache), no vector memory instruction
__device__ __forceinline__ float4 cld4(const float4* p, int i) { return ((rtw_c_f4*)p)[i]; }
__device__ __forceinline__ float4 gld4(const float4* p, size_t i) { return ((rtw_g_f4*)p)[i]; }
__device__ __forceinline__ float gld1(const float* p, size_t i) { return ((rtw_g_f1*)p)[i]; }
__device__ __forceinline__ float lld1(const float* p, int i) { return ((rtw_l_f1*)p)[i]; }
typedef __attribute__((address_space(3))) const float4 rtw_l_f4;
__device__ __forceinline__ float4 lld4(const float4* p, int i) { return ((rtw_l_f4*)p)[i]; }
__device__ __forceinline__ uint32_t lldu(const uint32_t* p, int i) { return ((rtw_l_u1*)p)[i]; }
__device__ __forceinline__ void lstu(uint32_t* p, int i, uint32_t v) { ((rtw_l_u1w*)p)[i] = v; }
#else
__device__ __forceinline__ float4 cld4(const float4* p, int i) { return p[i]; }
__device__ __forceinline__ uint32_t cldu(const uint32_t* p, int i) { return p[i]; }
__device__ __forceinline__ float4 gld4(const float4* p, size_t i) { return p[i]; }
__device__ __forceinline__ float gld1(const float* p, size_t i) { return p[i]; }
__device__ __forceinline__ float lld1(const float* p, int i) { return p[i]; }
__device__ __forceinline__ float4 lld4(const float4* p, int i) { return p[i]; }
__device__ __forceinline__ uint32_t lldu(const uint32_t* p, int i) { return p[i]; }
__device__ __forceinline__ void lstu(uint32_t* p, int i, uint32_t v) { p[i] = v; }
#endif

struct Ray { f3 o, d; float dist; };
struct Hit { f3 pos, normal; float dist; f3 color; float alpha; };

// ---- RRay::TestIntersectionWithAabb (Src/RRay.cpp:89-136), exact form ------------------------
__device__ __forceinline__ bool slab_exact(const Ray& r, float mnx, float mny, float mnz, float mxx, float mxy, float mxz, float& tmin_out, float& tmax_out)
{
    float tmin = -FLT_MAX, tmax = FLT_MAX;
    if (!near_zero(r.d.x)) {
        float inv = 1.0f / r.d.x;
        float t1 = (mnx - r.o.x) * inv, t2 = (mxx - r.o.x) * inv;
        tmin = ref_max(tmin, ref_min(t1, t2)); tmax = ref_min(tmax, ref_max(t1, t2));
    }
    if (!near_zero(r.d.y)) {
        float inv = 1.0f / r.d.y;
        float t1 = (mny - r.o.y) * inv, t2 = (mxy - r.o.y) * inv;
        tmin = ref_max(tmin, ref_min(t1, t2)); tmax = ref_min(tmax, ref_max(t1, t2));
    }
    if (!near_zero(r.d.z)) {
        float inv = 1.0f / r.d.z;
        float t1 = (mnz - r.o.z) * inv, t2 = (mxz - r.o.z) * inv;
        tmin = ref_max(tmin, ref_min(t1, t2)); tmax = ref_min(tmax, ref_max(t1, t2));
    }
    tmin_out = tmin; tmax_out = tmax;
    return tmax > tmin;
}

// A ray is "tame" when every direction component is a normal number of magnitude >= FLT_EPSILON
// and the origin is finite and moderate: then no slab axis is skipped and no NaN/inf can appear,
// so Math::Min/Max equal v_min/v_max and the reciprocal can be hoisted out of the node loop.
__device__ __forceinline__ bool ray_is_tame_impl(const Ray& r);
__device__ __forceinline__ bool ray_is_tame(const Ray& r)
{
    const bool t = ray_is_tame_impl(r);
    return t;
}
__device__ __forceinline__ bool ray_is_tame_impl(const Ray& r)
{
    const float big = 1.0e15f;
    return fabsf(r.d.x) >= FLT_EPSILON && fabsf(r.d.y) >= FLT_EPSILON && fabsf(r.d.z) >= FLT_EPSILON &&
           fabsf(r.d.x) < big && fabsf(r.d.y) < big && fabsf(r.d.z) < big &&
           fabsf(r.o.x) < big && fabsf(r.o.y) < big && fabsf(r.o.z) < big;
}

// ---- leaf: RRay::TestIntersectionWithTriangleAndFaceNormal (Src/RRay.cpp:147-213) ------------
// N and d1 = dot(N, p0) come precomputed from the host (they depend on the triangle only).
__device__ __forceinline__ bool triangle_test(const Ray& r, float cur_dist, const float4 a, const float4 b, const float4 c, float d1,
                                              f3& cp_out, float& dist_out)
{
    const f3 n = mk(a.w, b.w, c.w);
    const f3 p0 = mk(a.x, a.y, a.z), p1 = mk(b.x, b.y, b.z), p2 = mk(c.x, c.y, c.z);
    const f3 end = r.o + r.d * cur_dist;
    const float d0 = dot(n, r.o);
    const float d2 = d0 - d1;
    if (d2 < 0) return false;
    if (dot(end, n) - d1 > 0) return false;
    const f3 l = end - r.o;
    const float d3 = dot(n, l);
    if (near_zero(d3)) return false;
    const float df = -(d2 / d3);
    const f3 cp = r.o + l * df;
    if (dot(cross(p1 - p0, n), cp - p0) > 0) return false;
    if (dot(cross(p2 - p1, n), cp - p1) > 0) return false;
    if (dot(cross(p0 - p2, n), cp - p2) > 0) return false;
    const f3 ldf = l * df;
    cp_out = cp;
    dist_out = sqrtf(ldf.x * ldf.x + ldf.y * ldf.y + ldf.z * ldf.z);
    return true;
}

// ---- KdNode::TestRayIntersection (Src/KdTree.cpp:128-195) as a stack-free preorder walk --------
// Reference order: box miss -> skip subtree; internal hit -> left child (i+1) then right; leaf hit ->
// triangle test, shrink the segment on accept.  Equal-distance ties keep the LAST accepted leaf.
// PRUNE adds a segment clip (tmin <= dist+eps, tmax >= -eps): it only skips boxes that cannot hold an
// accepted hit, so the sequence of accepted hits -- and every output bit -- is unchanged.
template <bool TAME, bool STATS>
__device__ __forceinline__ bool tree_walk(const RtwSceneDev* __restrict__ sc, const RtwNode* __restrict__ nodes, const RtwTri* __restrict__ tris, int n_nodes, int n_tris,
                                          const Ray& r, bool prune, float& cur_dist, f3& hit_pos, int& hit_slot, Counters& ct)
{
    float ix = 0.f, iy = 0.f, iz = 0.f, eps_t = 0.f;
    if (TAME) {
        ix = 1.0f / r.d.x; iy = 1.0f / r.d.y; iz = 1.0f / r.d.z;
        // positional slack 2e-5 (scene units) expressed along the ray, plus a relative term
        eps_t = 2.0e-5f * fmaxf(fabsf(ix), fmaxf(fabsf(iy), fabsf(iz)));
    }
    bool any = false;
    int i = 0;
    const float4* nd4 = reinterpret_cast<const float4*>(nodes);
    const float4* tr4 = reinterpret_cast<const float4*>(tris);
    while (i < n_nodes) {
        const float4 lo = gld4(nd4, 2 * (size_t)i), hi = gld4(nd4, 2 * (size_t)i + 1);
        const int skip = __float_as_int(lo.w), leaf = __float_as_int(hi.w);
        bool hit; float tmin, tmax;
        if (TAME) {
            const float x1 = (lo.x - r.o.x) * ix, x2 = (hi.x - r.o.x) * ix;
            const float y1 = (lo.y - r.o.y) * iy, y2 = (hi.y - r.o.y) * iy;
            const float z1 = (lo.z - r.o.z) * iz, z2 = (hi.z - r.o.z) * iz;
            tmin = fmaxf(fmaxf(fminf(x1, x2), fminf(y1, y2)), fminf(z1, z2));
            tmax = fminf(fminf(fmaxf(x1, x2), fmaxf(y1, y2)), fmaxf(z1, z2));
            hit = tmax > tmin;
            if (prune) hit = hit && !(tmin > cur_dist + (eps_t + 1.0e-4f * cur_dist)) && !(tmax < -eps_t);
        } else {
            hit = slab_exact(r, lo.x, lo.y, lo.z, hi.x, hi.y, hi.z, tmin, tmax);
        }
        if (STATS) ct.boxes++;
        if (hit && leaf >= 0) {
            const float4 a = gld4(tr4, 4 * (size_t)leaf), b = gld4(tr4, 4 * (size_t)leaf + 1), c = gld4(tr4, 4 * (size_t)leaf + 2), d = gld4(tr4, 4 * (size_t)leaf + 3);
            if (STATS) ct.tris++;
            f3 cp; float dist;
            if (triangle_test(r, cur_dist, a, b, c, d.x, cp, dist)) {
                cur_dist = dist; hit_pos = cp; hit_slot = leaf; any = true;
            }
        }
        i = (hit && leaf < 0) ? i + 1 : skip;
    }
    return any;
}

// ---- packet walk: one wave, 64 coherent rays, ONE shared position in the tree ------------------------------
// For camera rays of neighbouring pixels.  The node index is wave-uniform (node and triangle records come
// through scalar loads, the walk needs no per-lane stack or list), every lane tests the current box with its
// own ray, and a subtree is entered when ANY lane's ray hits its box.  A lane only ever triangle-tests a leaf
// whose own box its ray hits, in preorder, with its own shrinking segment -- which is all the reference's
// result depends on (a ray that hits a box hits every enclosing box), so each lane gets the bits it would get
// walking alone.  Lanes with `active` false (no ray, ray outside the shape bound, or not tame) just ride along.
// EXACT: the lanes' rays are not "tame" (a near-zero direction component, NaN, ...): the box test then is the
// reference's own form (skipped axes, Math::Min/Max ternaries), with the three reciprocals hoisted; no pruning.
template <bool STATS, bool EXACT>
__device__ __forceinline__ bool packet_walk(const RtwNode* nodes, const RtwTri* tris, int n_nodes, const Ray& r, bool active, bool prune,
                                            float& cur_dist, f3& hit_pos, int& hit_slot, Counters& ct)
{
    const bool skx = near_zero(r.d.x), sky = near_zero(r.d.y), skz = near_zero(r.d.z);
    const float ix = (EXACT && skx) ? 0.0f : 1.0f / r.d.x, iy = (EXACT && sky) ? 0.0f : 1.0f / r.d.y, iz = (EXACT && skz) ? 0.0f : 1.0f / r.d.z;
    const float eps_t = 2.0e-5f * fmaxf(fabsf(ix), fmaxf(fabsf(iy), fabsf(iz)));
    const float4* nd4 = reinterpret_cast<const float4*>(nodes);
    const float4* tr4 = reinterpret_cast<const float4*>(tris);
    bool any = false;
    int i = 0;
    while (i < n_nodes) {
        const int iu = __builtin_amdgcn_readfirstlane(i);
        const float4 lo = cld4(nd4, 2 * iu), hi = cld4(nd4, 2 * iu + 1);
        const int skip = __float_as_int(lo.w), leaf = __float_as_int(hi.w);
        const float x1 = (lo.x - r.o.x) * ix, x2 = (hi.x - r.o.x) * ix;
        const float y1 = (lo.y - r.o.y) * iy, y2 = (hi.y - r.o.y) * iy;
        const float z1 = (lo.z - r.o.z) * iz, z2 = (hi.z - r.o.z) * iz;
        float tmin, tmax;
        if (EXACT) {            // RRay::TestIntersectionWithAabb as written (Src/RRay.cpp:93-126)
            tmin = -FLT_MAX; tmax = FLT_MAX;
            if (!skx) { tmin = ref_max(tmin, ref_min(x1, x2)); tmax = ref_min(tmax, ref_max(x1, x2)); }
            if (!sky) { tmin = ref_max(tmin, ref_min(y1, y2)); tmax = ref_min(tmax, ref_max(y1, y2)); }
            if (!skz) { tmin = ref_max(tmin, ref_min(z1, z2)); tmax = ref_min(tmax, ref_max(z1, z2)); }
        } else {
            tmin = fmaxf(fmaxf(fminf(x1, x2), fminf(y1, y2)), fminf(z1, z2));
            tmax = fminf(fminf(fmaxf(x1, x2), fmaxf(y1, y2)), fmaxf(z1, z2));
        }
        bool hit = active && (tmax > tmin);
        if (!EXACT && prune) hit = hit && !(tmin > cur_dist + (eps_t + 1.0e-4f * cur_dist)) && !(tmax < -eps_t);
        if (STATS) ct.boxes += active ? 1u : 0u;
        const bool any_hit = __ballot(hit) != 0ull;
        if (leaf >= 0) {
            if (any_hit) {
                const float4 a = cld4(tr4, 4 * leaf), b = cld4(tr4, 4 * leaf + 1), c = cld4(tr4, 4 * leaf + 2), d = cld4(tr4, 4 * leaf + 3);
                if (hit) {
                    if (STATS) ct.tris++;
                    f3 cp; float dist;
                    if (triangle_test(r, cur_dist, a, b, c, d.x, cp, dist)) { cur_dist = dist; hit_pos = cp; hit_slot = leaf; any = true; }
                }
            }
            i = skip;
        } else {
            i = any_hit ? iu + 1 : skip;
        }
    }
    return any;
}

// what a helper needs to know about the lane it runs on: does this lane add the per-ray work counters?
struct TravCtx { bool count; };
__device__ __forceinline__ TravCtx make_trav() { TravCtx t; t.count = true; return t; }

// ---- RTexture::Sample (Src/Texture.cpp:23-57) on RGBA8 texels + the host LUT ---------------------
__device__ __forceinline__ void texel_fetch(const uint32_t* __restrict__ tex, const float* __restrict__ lut, int idx, float& r, float& g, float& b, float& a)
{
    const uint32_t t = tex[idx];
    r = lut[t & 255u]; g = lut[(t >> 8) & 255u]; b = lut[(t >> 16) & 255u];
    a = (float)(t >> 24) / 255;
}
__device__ __forceinline__ float lerpf(float a, float b, float t) { return a + (b - a) * t; }
__device__ __forceinline__ void texture_sample(const uint32_t* __restrict__ texels, const RtwTexture& t, const float* __restrict__ lut,
                                               float u, float v, f3& rgb, float& alpha)
{
    const float cu = u - floorf(u), cv = v - floorf(v);
    const float fx = cu * (t.width - 1), fy = cv * (t.height - 1);
    int x0 = (int)floorf(fx), y0 = (int)floorf(fy), x1 = (int)ceilf(fx), y1 = (int)ceilf(fy);
    const float dx = fx - x0, dy = fy - y0;
    // finite uv always land inside the image; NaN/inf uv (undefined behaviour in the reference) must not fault
    x0 = min(max(x0, 0), t.width - 1); x1 = min(max(x1, 0), t.width - 1);
    y0 = min(max(y0, 0), t.height - 1); y1 = min(max(y1, 0), t.height - 1);
    const uint32_t* base = texels + t.offset;
    float r00, g00, b00, a00, r01, g01, b01, a01, r10, g10, b10, a10, r11, g11, b11, a11;
    texel_fetch(base, lut, y0 * t.width + x0, r00, g00, b00, a00);
    texel_fetch(base, lut, y0 * t.width + x1, r01, g01, b01, a01);
    texel_fetch(base, lut, y1 * t.width + x0, r10, g10, b10, a10);
    texel_fetch(base, lut, y1 * t.width + x1, r11, g11, b11, a11);
    rgb.x = lerpf(lerpf(r00, r01, dx), lerpf(r10, r11, dx), dy);
    rgb.y = lerpf(lerpf(g00, g01, dx), lerpf(g10, g11, dx), dy);
    rgb.z = lerpf(lerpf(b00, b01, dx), lerpf(b10, b11, dx), dy);
    alpha = lerpf(lerpf(a00, a01, dx), lerpf(a10, a11, dx), dy);
}

// ---- RSphere / RPlane / RCapsule (Src/Shapes.cpp:18-125) ---------------------------------------------------------------
// `seg` is TestRay.Distance at the time of the test (FindIntersectionWithScene shortens it hit by hit).
// RRay::TestIntersectionWithSphere (Src/RRay.cpp:25-66): quadratic in the parameter of Origin + t * (Direction * Distance)
__device__ __forceinline__ bool sphere_test(const Ray& r, float seg, f3 c, float radius, f3& pos, float& dist)
{
    const float dx = r.d.x * seg, dy = r.d.y * seg, dz = r.d.z * seg;
    const f3 o = r.o;
    const float qa = dx * dx + dy * dy + dz * dz;
    const float qb = 2.0f * dx * (o.x - c.x) + 2.0f * dy * (o.y - c.y) + 2.0f * dz * (o.z - c.z);
    const float qc = c.x * c.x + c.y * c.y + c.z * c.z + o.x * o.x + o.y * o.y + o.z * o.z +
                     -2.0f * (c.x * o.x + c.y * o.y + c.z * o.z) - radius * radius;
    const float d = qb * qb - 4.0f * qa * qc;
    if (!(d >= 0.0f)) return false;
    const float t = (-qb - sqrtf(d)) / (qa * 2.0f);
    if (t <= 0.0f) return false;
    const f3 hp = mk(o.x + t * dx, o.y + t * dy, o.z + t * dz);
    const f3 rel = hp - o;
    const float len = sqrtf(rel.x * rel.x + rel.y * rel.y + rel.z * rel.z);
    if (len > seg) return false;
    pos = hp; dist = len;
    return true;
}
// RRay::TestIntersectionWithPlane (Src/RRay.cpp:68-87); its `fabsf(denom) > 1e-6` compares in double
__device__ __forceinline__ bool plane_test(const Ray& r, float seg, f3 n, f3 pt, f3& pos, float& dist)
{
    const float denom = dot(n, r.d);
    if (!((double)fabsf(denom) > 1e-6)) return false;
    const float t = dot(pt - r.o, n) / denom;
    if (!(t >= 0.0f && t < seg)) return false;
    pos = r.o + r.d * t; dist = t;
    return true;
}
// RCapsule::TestRayCylinderIntersection (Src/Shapes.cpp:64-125)
__device__ __forceinline__ bool cylinder_test(const Ray& r, f3 start, f3 end, float radius, f3& pos, float& dist)
{
    const f3 d = end - start, m = r.o - start;
    const float dd = dot(d, d), nd = dot(r.d, d), mn = dot(m, r.d), md = dot(m, d), mm = dot(m, m);
    if (dot(r.o - start, end - start) < 0.0f && dot(r.d, end - start) < 0.0f) return false;
    if (dot(r.o - end, start - end) < 0.0f && dot(r.d, start - end) < 0.0f) return false;
    const float a = dd - nd * nd;
    const float b = dd * mn - nd * md;
    const float c = dd * (mm - radius * radius) - md * md;
    if (fabsf(a) < FLT_EPSILON) return false;
    if ((b * b - a * c) < 0.0f) return false;
    const float rt = (-b - sqrtf(b * b - a * c)) / a;
    if (rt < 0.0f) return false;
    const f3 v = r.o + r.d * rt;
    if (dot(v - start, end - start) < 0.0f) return false;
    if (dot(v - end, start - end) < 0.0f) return false;
    pos = v; dist = rt;
    return true;
}
// One analytic shape against the ray.  `part` tells what wrote the result: 0 = a sphere, a plane or a capsule's side (they set
// position, normal and distance only: the result's sampled colour and alpha stay what an earlier shape left there; so does an
// RTriangle), 1 / 2 = the
// capsule's end sphere at Start / End (RCapsule::TestRayIntersection copies a fresh RayHitResult in, which resets them;
// of two end hits the nearer, the second on a tie: Src/Shapes.cpp:34-62).
__device__ __forceinline__ bool analytic_test(const RtwShapeDev& sh, const Ray& r, float seg, f3& pos, float& dist, int& part)
{
    const f3 a = mk(sh.pa[0], sh.pa[1], sh.pa[2]), b = mk(sh.pb[0], sh.pb[1], sh.pb[2]);
    part = 0;
    if (sh.kind == RTW_SHAPE_SPHERE) return sphere_test(r, seg, a, sh.radius, pos, dist);
    if (sh.kind == RTW_SHAPE_PLANE) return plane_test(r, seg, a, b, pos, dist);
    if (sh.kind == RTW_SHAPE_TRIANGLE)      // RTriangle::TestRayIntersection -> RRay::TestIntersectionWithTriangle (Src/Shapes.cpp:127-130)
        return triangle_test(r, seg, make_float4(a.x, a.y, a.z, sh.pn[0]), make_float4(b.x, b.y, b.z, sh.pn[1]),
                             make_float4(sh.pc[0], sh.pc[1], sh.pc[2], sh.pn[2]), sh.pd1, pos, dist);
    if (sh.kind != RTW_SHAPE_CAPSULE) return false;
    if (cylinder_test(r, a, b, sh.radius, pos, dist)) return true;
    f3 p1 = mk(0, 0, 0), p2 = mk(0, 0, 0); float d1 = 0.0f, d2 = 0.0f;
    const bool b1 = sphere_test(r, seg, a, sh.radius, p1, d1);
    const bool b2 = sphere_test(r, seg, b, sh.radius, p2, d2);
    if (b1 && (!b2 || d1 < d2)) { pos = p1; dist = d1; part = 1; return true; }
    if (b2) { pos = p2; dist = d2; part = 2; return true; }
    return false;
}
// the normal that test wrote, from the hit position it wrote (the same operations on the same values)
__device__ __forceinline__ f3 analytic_normal(const RtwShapeDev& sh, f3 pos, int part)
{
    const f3 a = mk(sh.pa[0], sh.pa[1], sh.pa[2]), b = mk(sh.pb[0], sh.pb[1], sh.pb[2]);
    if (sh.kind == RTW_SHAPE_PLANE) return a;
    if (sh.kind == RTW_SHAPE_TRIANGLE) return mk(sh.pn[0], sh.pn[1], sh.pn[2]);
    if (sh.kind == RTW_SHAPE_SPHERE || part == 1) return normalized(pos - a);
    if (part == 2) return normalized(pos - b);
    const f3 side = cross(b - a, pos - a);
    return normalized(cross(side, b - a));
}

// ---- RMeshShape::TestRayIntersection (Src/MeshShape.cpp:280-332) ------------------------------------
// The part of RMeshShape::TestRayIntersection after the tree query (Src/MeshShape.cpp:288-327): barycentrics,
// fast-normalised smooth normal, texture sample.
template <bool STATS>
__device__ __forceinline__ void mesh_finish(const RtwSceneDev* __restrict__ sc, const RtwShapeDev& sh, const TravCtx& tc,
                                            f3 pos, float cur, int slot, Hit& out, int& tri_index, Counters& ct)
{
    if (STATS && tc.count) ct.hits++;
    const float4* tr4 = reinterpret_cast<const float4*>(sh.tris);
    const float4 ta = gld4(tr4, 4 * (size_t)slot), tb = gld4(tr4, 4 * (size_t)slot + 1), tcc = gld4(tr4, 4 * (size_t)slot + 2), td = gld4(tr4, 4 * (size_t)slot + 3);
    tri_index = __float_as_int(td.y);
    const f3 a = mk(ta.x, ta.y, ta.z), b = mk(tb.x, tb.y, tb.z), c = mk(tcc.x, tcc.y, tcc.z);
    // RMath::Barycentric (Src/Math.cpp:56-68)
    const f3 v0 = b - a, v1 = c - a, v2 = pos - a;
    const float d00 = dot(v0, v0), d01 = dot(v0, v1), d11 = dot(v1, v1), d20 = dot(v2, v0), d21 = dot(v2, v1);
    const float denom = d00 * d11 - d01 * d01;
    const float bv = (d11 * d20 - d01 * d21) / denom;
    const float bw = (d00 * d21 - d01 * d20) / denom;
    const float bu = 1.0f - bv - bw;
    const float4* sh4 = reinterpret_cast<const float4*>(sh.shade);
    const float4 s0 = gld4(sh4, 4 * (size_t)slot), s1 = gld4(sh4, 4 * (size_t)slot + 1), s2 = gld4(sh4, 4 * (size_t)slot + 2), s3 = gld4(sh4, 4 * (size_t)slot + 3);
    const f3 n0 = mk(s0.x, s0.y, s0.z), n1 = mk(s0.w, s1.x, s1.y), n2 = mk(s1.z, s1.w, s2.x);
    out.pos = pos;
    out.dist = cur;
    out.normal = normalized_fast((n0 * bu + n1 * bv) + n2 * bw);
    out.color = mk(1.0f, 1.0f, 1.0f);      // *OutResult = HitResult resets the sampled colour (Src/KdTree.cpp:176)
    out.alpha = 1.0f;
    const int mat = __float_as_int(s3.w);
    if (mat != -1 && mat < sh.n_textures && mat < RTW_DEV_MAX_TEXTURES && sh.textures[mat].valid) {
        const float tu = (s2.y * bu + s2.w * bv) + s3.y * bw;      // t0*u + t1*v + t2*w
        const float tv = (s2.z * bu + s3.x * bv) + s3.z * bw;
        if (STATS && tc.count) ct.tex++;
        texture_sample(sh.texels, sh.textures[mat], sc->texel_lut, tu, 1.0f - tv, out.color, out.alpha);
    }
}

// A recorded hit (shape, position, distance, leaf slot or analytic part) -> the RayHitResult RayTrace shades.  Records are only
// kept for scenes in which no analytic hit can inherit a texel (the host sends the others through the single kernel).
template <bool STATS>
__device__ __forceinline__ void hit_finish(const RtwSceneDev* __restrict__ sc, const RtwShapeDev& sh, const TravCtx& tc,
                                           f3 pos, float cur, int slot, Hit& out, int& tri_index, Counters& ct)
{
    if (sh.kind == RTW_SHAPE_MESH) { mesh_finish<STATS>(sc, sh, tc, pos, cur, slot, out, tri_index, ct); return; }
    out.pos = pos; out.dist = cur; out.normal = analytic_normal(sh, pos, slot);
    out.color = mk(1.0f, 1.0f, 1.0f); out.alpha = 1.0f;
    tri_index = -1;
}

// RMeshShape::TestRayIntersection for one ray per lane: the reference's own walk of the binary tree (tree_walk), then the shading inputs
template <bool STATS>
__device__ __forceinline__ bool mesh_query(const RtwSceneDev* __restrict__ sc, const RtwShapeDev& sh, const TravCtx& tc,
                                           const Ray& r, float seg_dist, Hit& out, int& tri_index, Counters& ct)
{
    float cur = seg_dist; f3 pos = mk(0, 0, 0); int slot = -1;
    bool any;
    Counters walk = { 0, 0, 0, 0, 0, 0 };
    if (ray_is_tame(r)) any = tree_walk<true, STATS>(sc, sh.nodes, sh.tris, sh.n_nodes, sh.n_tris, r, sc->prune != 0, cur, pos, slot, walk);
    else any = tree_walk<false, STATS>(sc, sh.nodes, sh.tris, sh.n_nodes, sh.n_tris, r, false, cur, pos, slot, walk);
    if (STATS && tc.count) { ct.boxes += walk.boxes; ct.tris += walk.tris; }
    if (!any) return false;
    mesh_finish<STATS>(sc, sh, tc, pos, cur, slot, out, tri_index, ct);
    return true;
}

// ---- RayTracerScene::FindIntersectionWithScene (Src/RayTracerScene.cpp:99-125) ------------------------
template <bool STATS>
__device__ __forceinline__ int find_intersection(const RtwSceneDev* __restrict__ sc, const TravCtx& tc, const Ray& in, Hit& out, int& tri_index, Counters& ct)
{
    int hit_shape = -1;
    float seg = in.dist;
    if (STATS && tc.count) ct.rays++;
    out.color = mk(1.0f, 1.0f, 1.0f); out.alpha = 1.0f;      // RayHitResult() (Src/RRay.h:15-20); ONE result serves every shape below
    for (int s = 0; s < sc->n_shapes; s++) {
        const RtwShapeDev& sh = sc->shapes[s];
        if (sh.kind != RTW_SHAPE_PLANE) {                    // RPlane::HasCullingBounds() is false (Src/Shapes.cpp:28-32)
            float t0, t1;
            if (STATS && tc.count) ct.boxes++;
            if (!slab_exact(in, sh.bmin[0], sh.bmin[1], sh.bmin[2], sh.bmax[0], sh.bmax[1], sh.bmax[2], t0, t1)) continue;
        }
        if (sh.kind != RTW_SHAPE_MESH) {
            f3 pos; float dist; int part;
            if (analytic_test(sh, in, seg, pos, dist, part)) {
                out.pos = pos; out.dist = dist; out.normal = analytic_normal(sh, pos, part);
                if (part != 0) { out.color = mk(1.0f, 1.0f, 1.0f); out.alpha = 1.0f; }
                tri_index = -1;
                seg = dist; hit_shape = s;
            }
            continue;
        }
        if (mesh_query<STATS>(sc, sh, tc, in, seg, out, tri_index, ct)) { seg = out.dist; hit_shape = s; }
    }
    return hit_shape;
}

// The first `lead` shapes of FindIntersectionWithScene for one ray per lane (they are spheres / planes / capsules): the same
// tests in the same order on the same running distance; the query is continued from (seg, hit_shape, hit_slot, hit_pos).
template <bool STATS>
__device__ __forceinline__ void lead_find(const RtwSceneDev* __restrict__ sc, int lead, const Ray& in, float& seg, int& hit_shape, int& hit_slot, f3& hit_pos, Counters& ct)
{
    for (int s = 0; s < lead; s++) {
        const RtwShapeDev& sh = sc->shapes[s];
        if (sh.kind != RTW_SHAPE_PLANE) {
            float t0, t1;
            if (STATS) ct.boxes++;
            if (!slab_exact(in, sh.bmin[0], sh.bmin[1], sh.bmin[2], sh.bmax[0], sh.bmax[1], sh.bmax[2], t0, t1)) continue;
        }
        f3 pos; float dist; int part;
        if (analytic_test(sh, in, seg, pos, dist, part)) { seg = dist; hit_shape = s; hit_slot = part; hit_pos = pos; }
    }
}

// ---- materials (Src/SurfaceMaterials.cpp:20-187), flattened tree, explicit evaluation stack -------------
struct Bounce { f3 att, em; };

__device__ __forceinline__ f3 unit_vector_f64(float r1, float r2)     // RMath::RandomUnitVector, double transcendentals
{
    const float t1 = 2.0f * 3.1415926f * r1;
    const float t2 = (float)acos((double)(1.0f - 2.0f * r2));
    const float sin_t2 = (float)sin((double)t2);
    return mk((float)sin((double)t1) * sin_t2, (float)cos((double)t1) * sin_t2, (float)cos((double)t2));
}
// x mod (2^24 - 1) by digit folding (2^24 == 1 mod 2^24-1).  Written out because hipcc (ROCm 7.2) lowered
// `uint64 % 0xFFFFFF` on gfx950 to a sequence that left values outside the table (GPU memory fault at
// 1920x1080, where the per-path cursor first exceeds the table size); exact for every 64-bit x.
__device__ __forceinline__ uint32_t mod_table_size(uint64_t x)
{
    uint32_t s = (uint32_t)(x & 0xFFFFFFu) + (uint32_t)((x >> 24) & 0xFFFFFFu) + (uint32_t)(x >> 48);   // < 2^26
    s = (s & 0xFFFFFFu) + (s >> 24);                                                                     // < 2^24 + 4
    if (s >= RTW_TABLE_SIZE) s -= RTW_TABLE_SIZE;
    if (s >= RTW_TABLE_SIZE) s -= RTW_TABLE_SIZE;
    return s;
}
// issue the loads of the unit-table entry the next hemisphere draw of this path will use
__device__ __forceinline__ void prefetch_unit_vector(const RtwSceneDev* __restrict__ sc, PathRng& rng)
{
    uint32_t idx = mod_table_size(rng.table_base + rng.table_reads);
    if (sc->debug_table_mask) idx &= (uint32_t)sc->debug_table_mask;
    const float* e = sc->unit_table;
    rng.pre_x = gld1(e, (size_t)idx * 3); rng.pre_y = gld1(e, (size_t)idx * 3 + 1); rng.pre_z = gld1(e, (size_t)idx * 3 + 2);
    rng.pre_reads = rng.table_reads;
}
__device__ __forceinline__ f3 hemisphere_direction(const RtwSceneDev* __restrict__ sc, f3 normal, PathRng& rng)
{
    uint32_t idx = mod_table_size(rng.table_base + rng.table_reads);   // RMath::PseudoRandomUnitVector, per-path cursor
    if (sc->debug_table_mask) idx &= (uint32_t)sc->debug_table_mask;
    const bool fetched = rng.pre_reads == rng.table_reads;
    rng.table_reads++;
    const float* e = sc->unit_table;
    const f3 v = fetched ? mk(rng.pre_x, rng.pre_y, rng.pre_z) : mk(gld1(e, (size_t)idx * 3), gld1(e, (size_t)idx * 3 + 1), gld1(e, (size_t)idx * 3 + 2));
    if (dot(v, normal) > 0.0f) return v;                                          // Src/Math.cpp:42-54
    return reflect(v, normal);
}
__device__ __forceinline__ bool checker_brighter(f3 p, float size)
{
    const float recip = near_zero(size) ? 1.0f : 1.0f / size;
    bool r = false;
    const float fx = p.x * recip, fy = p.y * recip, fz = p.z * recip;
    if (fx - floorf(fx) > 0.5f) r = !r;
    if (fz - floorf(fz) > 0.5f) r = !r;
    if (fy - floorf(fy) > 0.5f) r = !r;
    return r;
}

__device__ __forceinline__ Bounce leaf_bounce(const RtwSceneDev* __restrict__ sc, const RtwMaterialNode& n, const Ray& in, const Hit& h, Ray& out, PathRng& rng)
{
    Bounce b; b.att = mk(0, 0, 0); b.em = mk(0, 0, 0);
    const f3 albedo = mk(n.r, n.g, n.b);
    switch (n.type) {
    case 0: case 1: {
        const float factor = (n.type == 1) ? (checker_brighter(h.pos, n.param) ? 1.0f : 0.5f) : 1.0f;
        const float rd = in.dist - h.dist;
        const f3 dir = hemisphere_direction(sc, h.normal, rng);
        out.o = h.pos + dir * 0.0001f; out.d = dir; out.dist = rd;
        const float d = ref_max(0.0f, dot(h.normal, dir));
        b.att = albedo * d;
        if (n.type == 1) b.att = b.att * factor;
        break;
    }
    case 2: {
        const float rd = in.dist - h.dist;
        f3 nd = reflect(in.d, h.normal);
        if (n.param > 0.0f) {
            const float r1 = rng.random();
            const float r2 = rng.random();
            nd = nd + unit_vector_f64(r1, r2) * n.param;
            nd = normalized(nd);
        }
        out.o = h.pos + nd * 0.0001f; out.d = nd; out.dist = rd;
        b.att = albedo;
        break;
    }
    case 3: out = in; b.em = albedo; break;
    default: {
        const float rd = in.dist - h.dist;
        out.o = h.pos + in.d * 0.0001f; out.d = in.d; out.dist = rd;
        b.att = mk(1, 1, 1);
        break;
    }
    }
    return b;
}
__device__ __forceinline__ f3 leaf_preview(const RtwMaterialNode& n, const Hit& h)
{
    const f3 albedo = mk(n.r, n.g, n.b);
    switch (n.type) {
    case 0: return albedo * (dot(h.normal, mk(0, 1, 0)) * 0.5f + 0.5f);
    case 1: return (albedo * (dot(h.normal, mk(0, 1, 0)) * 0.5f + 0.5f)) * (checker_brighter(h.pos, n.param) ? 1.0f : 0.5f);
    case 2: case 3: return albedo;
    default: return mk(0, 0, 0);
    }
}

// Evaluate BounceViewRay (PREVIEW=false) or PreviewColor (PREVIEW=true) of a material tree.
// Combine evaluates its B operand first (what g++ 11 does for A(...) + B(...); pinned by oracle/_ref),
// then adds A + B.  `att` carries the preview colour when PREVIEW.
#define RTW_EVAL_DEPTH 2
// Combine nesting is limited to RTW_EVAL_DEPTH = 2 so that the evaluation stack is two named register
// sets (no dynamically indexed private array, hence no scratch memory).
template <bool PREVIEW>
__device__ __forceinline__ Bounce material_eval(const RtwSceneDev* __restrict__ sc, const RtwShapeDev& sh, const Ray& in, const Hit& h, Ray& out, PathRng& rng)
{
    int node0 = 0, node1 = 0, phase0 = 0, phase1 = 0;
    Bounce saved0, saved1;
    saved0.att = saved0.em = saved1.att = saved1.em = mk(0, 0, 0);
    int sp = 0;
    int cur = 0;
    Bounce res; res.att = mk(0, 0, 0); res.em = mk(0, 0, 0);
    for (;;) {
        const RtwMaterialNode n = sh.material[cur];
        if (n.type == 4) {                       // Blend: Random() > factor ? A : B  (factor clamped to [0,1])
            const float bf = n.param < 0.0f ? 0.0f : (n.param > 1.0f ? 1.0f : n.param);
            cur = rng.random() > bf ? n.child_a : n.child_b;
            continue;
        }
        if (n.type == 5 && sp < RTW_EVAL_DEPTH) {
            if (sp == 0) { node0 = cur; phase0 = 0; } else { node1 = cur; phase1 = 0; }
            sp++;
            cur = n.child_b;
            continue;
        }
        if (PREVIEW) { res.att = leaf_preview(n, h); res.em = mk(0, 0, 0); }
        else res = leaf_bounce(sc, n, in, h, out, rng);
        bool have_result = true;
        while (have_result) {
            if (sp == 0) return res;
            const bool top0 = (sp == 1);
            const int phase = top0 ? phase0 : phase1;
            if (phase == 0) {                    // B done: keep it, evaluate A
                if (top0) { saved0 = res; phase0 = 1; cur = sh.material[node0].child_a; }
                else { saved1 = res; phase1 = 1; cur = sh.material[node1].child_a; }
                have_result = false;
            } else {                             // A done: A + B
                const Bounce sb = top0 ? saved0 : saved1;
                res.att = res.att + sb.att;
                res.em = res.em + sb.em;
                sp--;
            }
        }
    }
}

// ---- RayTracerScene::RayTrace (Src/RayTracerScene.cpp:31-97): recursion -> walk forward, fold back ----------
// The per-level factors of the recursion live in a global workspace, structure-of-arrays over the
// launch's threads (slot (level, j) of thread t = ws[(level * 3 + j) * stride + t]): coalesced, no
// scratch.  j = 0: attenuation + kind (0 opaque with child, 2 transparent pass-through),
// j = 1: sampled colour, j = 2: emissive.
struct LevelStore {
    float4* __restrict__ ws; size_t stride; size_t tid;
    int rec_levels;     // 0: structure of arrays over the launch's threads (coalesced); > 0: one record of rec_levels x 3 float4 per
                        // path slot (the slots of the per-bounce pipeline are sparse: a path's levels then share cache lines / pages);
                        // < 0: per level an array of 3 x float4 records over the slots (neighbouring slots neighbours, a record's 48 bytes together)
    __device__ __forceinline__ float4& at(int level, int j) const
    {
        return rec_levels > 0 ? ws[(tid * (size_t)rec_levels + (size_t)level) * 3 + (size_t)j]
             : rec_levels < 0 ? ws[((size_t)level * stride + tid) * 3 + (size_t)j] : ws[((size_t)level * 3 + (size_t)j) * stride + tid];
    }
};

template <bool STATS>
__device__ f3 trace_path(const RtwSceneDev* __restrict__ sc, const TravCtx& tc, Ray ray, int max_bounce, bool preview, PathRng& rng, Counters& ct, const LevelStore& lv)
{
    int nlev = 0;
    f3 L = mk(0, 0, 0);
    int depth = max_bounce;
    for (;;) {
        if (depth == 0) { L = mk(0, 0, 0); break; }
        Hit h; int tri;
        const int s = find_intersection<STATS>(sc, tc, ray, h, tri, ct);
        if (s < 0) {                                                     // sky (Src/RayTracerScene.cpp:89-94)
            const float t = 0.5f * (ray.d.y + 1.0f);
            L = mk(1.0f, 1.0f, 1.0f) * (1.0f - t) + mk(0.5f, 0.7f, 1.0f) * t;
            break;
        }
        const RtwShapeDev& sh = sc->shapes[s];
        if (!sh.has_material) { L = mk(0, 0, 0); break; }
        Ray out = ray;
        if (preview) {
            const Bounce p = material_eval<true>(sc, sh, ray, h, out, rng);
            L = mk(0, 0, 0) + p.att * h.color;
            break;
        }
        const Bounce b = material_eval<false>(sc, sh, ray, h, out, rng);
        if (rng.random() <= h.alpha) {
            if (all_nonzero(b.att)) {
                lv.at(nlev, 0) = make_float4(b.att.x, b.att.y, b.att.z, __int_as_float(0));
                lv.at(nlev, 1) = make_float4(h.color.x, h.color.y, h.color.z, 0.0f);
                lv.at(nlev, 2) = make_float4(b.em.x, b.em.y, b.em.z, 0.0f);
                nlev++;
                ray = out; depth--;
                continue;
            }
            L = mk(0, 0, 0) + b.em;
            break;
        }
        // transparent texel: same direction, remaining distance, no colour factor
        lv.at(nlev, 0) = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(2));
        nlev++;
        const float rd = ray.dist - h.dist;
        ray.o = h.pos + ray.d * 0.0001f; ray.dist = rd;
        depth--;
    }
    for (int k = nlev - 1; k >= 0; k--) {
        const float4 a = lv.at(k, 0);
        if (__float_as_int(a.w) == 0) {
            const float4 c = lv.at(k, 1), e = lv.at(k, 2);
            L = (mk(0, 0, 0) + (mk(a.x, a.y, a.z) * L) * mk(c.x, c.y, c.z)) + mk(e.x, e.y, e.z);
        } else {
            L = mk(0, 0, 0) + L;
        }
    }
    return L;
}

// ---- camera + resolve (Src/RayTracerProgram.cpp:131-188, Src/ColorBuffer.h:81-109) ---------------------------
__device__ __forceinline__ Ray camera_ray(int width, int height, int pixel, int i, PathRng& rng)
{
    const float aspect = (float)width / (float)height;
    const int x = pixel % width, y = pixel / width;
    const float dx = -(float)(x - width / 2) / (width * 2) * aspect;
    const float dy = -(float)(y - height / 2) / (height * 2);
    const float inv_pixel_radius = 1.0f / (width * 4);
    const float offset_radius = inv_pixel_radius * 0.5f;
    float ox = (i & 1) ? inv_pixel_radius : 0.0f;
    float oy = (i & 2) ? inv_pixel_radius : 0.0f;
    ox += (rng.random() - 0.5f) * offset_radius;
    oy += (rng.random() - 0.5f) * offset_radius;
    Ray r;
    r.o = mk(0, 0, 7.0f);
    r.d = normalized(mk(dx + ox, dy + oy, -0.5f));
    r.dist = 1000.0f;
    return r;
}

// the same ray from the host's per-column / per-row tables (tiled pipelines): dx and dy hold exactly the floats above
__device__ __forceinline__ Ray camera_ray_xy(const RtwRenderParams& p, int x, int y, int i, PathRng& rng)
{
    const float dx = p.cam_dx[x], dy = p.cam_dy[y];
    const float inv_pixel_radius = 1.0f / (p.width * 4);
    const float offset_radius = inv_pixel_radius * 0.5f;
    float ox = (i & 1) ? inv_pixel_radius : 0.0f;
    float oy = (i & 2) ? inv_pixel_radius : 0.0f;
    ox += (rng.random() - 0.5f) * offset_radius;
    oy += (rng.random() - 0.5f) * offset_radius;
    Ray r;
    r.o = mk(0, 0, 7.0f);
    r.d = normalized(mk(dx + ox, dy + oy, -0.5f));
    r.dist = 1000.0f;
    return r;
}

// 8-bit value of MakePixelColor(LinearToGamma(c)) for one channel: the largest k with thr[k] <= c.
// A fast exp2/log2 guess is corrected against the exact host thresholds, so the guess quality
// affects speed only.
__device__ __forceinline__ uint32_t gamma_channel(const float* __restrict__ thr, float c)
{
    if (!(c > 0.0f)) return 0u;
    if (c >= 1.0f) return 255u;
    int k = (int)(__builtin_amdgcn_exp2f(__builtin_amdgcn_logf(c) * (1.0f / 2.2f)) * 255.0f);
    k = k < 0 ? 0 : (k > 255 ? 255 : k);
    while (k < 255 && thr[k + 1] <= c) k++;
    while (k > 0 && thr[k] > c) k--;
    return (uint32_t)k;
}
// guess of gamma_channel's k (quality affects speed only)
__device__ __forceinline__ int gamma_guess(float c)
{
    if (!(c > 0.0f)) return 0;          // (NaN too: no float-to-int conversion of a NaN)
    if (c >= 1.0f) return 254;
    const int k = (int)(__builtin_amdgcn_exp2f(__builtin_amdgcn_logf(c) * (1.0f / 2.2f)) * 255.0f);
    return k < 0 ? 0 : (k > 254 ? 254 : k);
}
// the largest k with thr[k] <= c, given the guess g in 0..254 and the two table entries around it
__device__ __forceinline__ uint32_t gamma_fix(const float* __restrict__ thr, float c, int g, float t0, float t1)
{
    if (!(c > 0.0f)) return 0u;
    if (c >= 1.0f) return 255u;
    if (t0 <= c && !(t1 <= c)) return (uint32_t)g;          // thr[g] <= c < thr[g + 1]: the guess was right (almost always)
    int k = g;
    while (k < 255 && thr[k + 1] <= c) k++;
    while (k > 0 && thr[k] > c) k--;
    return (uint32_t)k;
}
__device__ __forceinline__ uint32_t pack_pixel(const float* __restrict__ thr, f3 c)
{
    // the three channels' table reads are issued together (one LDS round trip instead of six dependent ones)
    const int gx = gamma_guess(c.x), gy = gamma_guess(c.y), gz = gamma_guess(c.z);
    const float x0 = thr[gx], x1 = thr[gx + 1], y0 = thr[gy], y1 = thr[gy + 1], z0 = thr[gz], z1 = thr[gz + 1];
    return (255u << 24) | (gamma_fix(thr, c.x, gx, x0, x1) << 16) | (gamma_fix(thr, c.y, gy, y0, y1) << 8) | gamma_fix(thr, c.z, gz, z0, z1);
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// counters: one atomic per wave and counter (the whole wave must call this together)
__device__ __forceinline__ void flush_counters(const RtwSceneDev* __restrict__ sc, const Counters& ct)
{
    const uint32_t r = wave_sum(ct.rays), b = wave_sum(ct.boxes), t = wave_sum(ct.tris), h = wave_sum(ct.hits), x = wave_sum(ct.tex), c = wave_sum(ct.cams);
    if ((threadIdx.x & 63) != 0) return;
    if (!sc->stats) return;
    atomicAdd(&sc->stats[0], (unsigned long long)r);
    atomicAdd(&sc->stats[1], (unsigned long long)b);
    atomicAdd(&sc->stats[2], (unsigned long long)t);
    atomicAdd(&sc->stats[3], (unsigned long long)h);
    atomicAdd(&sc->stats[4], (unsigned long long)x);
    atomicAdd(&sc->stats[5], (unsigned long long)c);
}

// AccumulatePixel::AddPixel + GetGammaSpacePixel, or the preview write (Src/RayTracerProgram.cpp:174-186)
__device__ __forceinline__ void resolve_pixel(const float* __restrict__ thr, float4* __restrict__ accum, uint32_t* __restrict__ argb,
                                              int pixel, f3 c, bool preview)
{
    if (preview) {
        argb[pixel] = pack_pixel(thr, c);
    } else {
        const float4 a = accum[pixel];
        const f3 sum = mk(a.x, a.y, a.z) + c;
        const int n = __float_as_int(a.w) + 1;
        accum[pixel] = make_float4(sum.x, sum.y, sum.z, __int_as_float(n));
        argb[pixel] = pack_pixel(thr, n == 1 ? sum : sum / (float)n);       // x / 1.0f == x: the first pass skips three divides
    }
}

// tiled mapping: wave (wi >> 6) renders one tile_w x tile_h tile of its rows, lanes row-major inside the tile
__device__ __forceinline__ bool work_to_xy(const RtwRenderParams& p, int wi, int& x, int& y)
{
    const int wt = wi >> 6, l = wi & 63;
    int band = (int)(((float)wt + 0.5f) / (float)p.tiles_per_row);
    if (band * p.tiles_per_row > wt) band--;
    if ((band + 1) * p.tiles_per_row <= wt) band++;
    const int tx = wt - band * p.tiles_per_row;
    const int vr = band * p.tile_h + (l >> p.tile_shift);
    x = tx * p.tile_w + (l & (p.tile_w - 1));
    if (p.world <= 1) y = p.row0 + vr;
    else { const int j = vr / p.task_rows, r = vr - j * p.task_rows; y = (j * p.world + p.rank) * p.task_rows + r; }
    return vr < p.nrows && y < p.height;
}
__device__ __forceinline__ int work_to_pixel(const RtwRenderParams& p, int wi)
{
    if (p.tile_w != 0) {
        int x, y;
        return work_to_xy(p, wi, x, y) ? y * p.width + x : p.width * p.height;
    }
    if (p.world <= 1) return p.begin + wi;
    const int per_task = p.task_rows * p.width;
    const int j = wi / per_task, r = wi - j * per_task;
    return (j * p.world + p.rank) * per_task + r;
}

// ---- kernels -----------------------------------------------------------------------------------------
template <bool STATS>
__global__ __launch_bounds__(256) void render_kernel(const RtwSceneDev* __restrict__ sc, float4* __restrict__ accum,
                                                     uint32_t* __restrict__ argb, float4* __restrict__ ws, RtwRenderParams p)
{
    __shared__ float thr[256];
    thr[threadIdx.x] = sc->gamma_thr[threadIdx.x];
    __syncthreads();
    const TravCtx tc = make_trav();
    const int wi = blockIdx.x * blockDim.x + threadIdx.x;
    const int npix = p.width * p.height;
    const int pixel = wi < p.count ? work_to_pixel(p, wi) : npix;
    Counters ct = { 0, 0, 0, 0, 0, 0 };
    if (pixel < npix) {
        const uint32_t phase = table_phase(p.seed);
        LevelStore lv; lv.ws = ws; lv.stride = (size_t)gridDim.x * blockDim.x; lv.tid = (size_t)wi; lv.rec_levels = 0;
        f3 c = mk(0, 0, 0);
        for (int i = 0; i < p.sub_samples; i++) {
            PathRng rng; rng_init(rng, p.seed, phase, (uint64_t)npix, (uint32_t)pixel, (uint32_t)p.pass_index, (uint32_t)i);
            const Ray ray = camera_ray(p.width, p.height, pixel, i, rng);
            if (STATS) ct.cams++;
            c = c + trace_path<STATS>(sc, tc, ray, p.max_bounce, p.preview != 0, rng, ct, lv);
        }
        c = c / (float)p.sub_samples;
        resolve_pixel(thr, accum, argb, pixel, c, p.preview != 0);
    }
    if (STATS) flush_counters(sc, ct);
}

// ---- the bins + wave pipeline (pipeline 3; rtw_wave_kernels.h): one pass per set of launches.  Kept as the one-pass reference the pass-batched
// pipeline is compared with: primary_bins_kernel (+ primary_sky_kernel beside it), then per bounce one wave-per-ray trace launch and one shade
// launch, then resolve_kernel.  A path's slot in the dense arrays = work item * sub_samples + sub-sample. ----
struct PipeBufs {
    uint32_t* __restrict__ queue;      // round 0's trace list: the slots of the paths the primary kernel's shading step left alive
    uint32_t* __restrict__ pend;       // work items (pixels) with at least one sample that hit something: resolve_kernel sums their samples
    float4* __restrict__ rad;          // radiance per path id = work item * 4 + sub-sample (only slots of pending pixels are used)
    uint32_t* __restrict__ counters;   // [0] queue length, [1] pending length, [4 + r] length of round r's trace list; [64 ..] their values at the end of the previous pass
    float4* __restrict__ ws;           // level store: max_bounce x 3 float4 per slot
    float4* __restrict__ hitslot;      // 2 x float4 per slot: hit position + distance, shape + leaf slot (+ "more to trace" flag with leading analytic shapes)
    float4* __restrict__ state;        // 3 x float4 per slot: origin + distance, direction + draw counter, key / table reads / levels / depth
    uint32_t* __restrict__ tlist0;     // trace lists (slots whose next segment must be traced), ping-pong
    uint32_t* __restrict__ tlist1;
    uint32_t capacity;                 // slots the dense arrays hold
};

__device__ __forceinline__ void wave_push(uint32_t* __restrict__ list, uint32_t* __restrict__ counter, bool flag, uint32_t value)
{
    const unsigned long long m = __ballot(flag);
    if (m == 0ull) return;
    const int lane = (int)(threadIdx.x & 63u);
    const int leader = __ffsll((long long)m) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, leader);
    if (flag) list[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = value;
}

__device__ __forceinline__ f3 sky_color(float dir_y)        // Src/RayTracerScene.cpp:92-93
{
    const float t = 0.5f * (dir_y + 1.0f);
    return mk(1.0f, 1.0f, 1.0f) * (1.0f - t) + mk(0.5f, 0.7f, 1.0f) * t;
}

__device__ __forceinline__ const uint32_t* wf_list(const PipeBufs& pb, int which) { return which ? pb.tlist1 : pb.tlist0; }

// slot q = work item * sub_samples + sub-sample  <->  path id = work item * 4 + sub-sample
__device__ __forceinline__ uint32_t slot_of_path(const RtwRenderParams& p, uint32_t wi, uint32_t sub) { return wi * (uint32_t)p.sub_samples + sub; }
__device__ __forceinline__ uint32_t pid_of_slot(const RtwRenderParams& p, uint32_t q)
{
    switch (p.sub_samples) {
    case 1: return q * 4u;
    case 2: return (q >> 1) * 4u + (q & 1u);
    case 4: return q;
    default: { const uint32_t wi = q / 3u; return wi * 4u + (q - wi * 3u); }
    }
}

// RayTrace's per-hit block (Src/RayTracerScene.cpp:47-94) for one path whose segment has just been traced: shade the
// recorded hit (r0 = position + distance, r1 = shape + leaf slot; shape < 0 = the segment missed), push a level, set up
// the next segment; or finish the path (fold the levels back in the reference's association order, write the radiance).
// Returns true when the path goes on: its state is then saved in slot q.
template <bool STATS, bool AN>
__device__ __forceinline__ bool shade_hit_step(const RtwSceneDev* __restrict__ sc, const PipeBufs& pb, const RtwRenderParams& p, uint32_t q, uint32_t pid,
                                               Ray ray, PathRng rng, int depth, int nlev, float4 r0, float4 r1, Counters& ct)
{
    const TravCtx tc = make_trav();
    if (!p.preview) prefetch_unit_vector(sc, rng);     // the table read (an HBM miss) overlaps the record loads below
    LevelStore lv; lv.ws = pb.ws; lv.stride = (size_t)pb.capacity; lv.tid = (size_t)q; lv.rec_levels = p.max_bounce > 0 ? p.max_bounce : 1;
    f3 L = mk(0, 0, 0);
    bool done = false;
    const int hs = __float_as_int(r1.x), slot = __float_as_int(r1.y);
    if (hs < 0) { L = sky_color(ray.d.y); done = true; }
    else {
        const RtwShapeDev& sh = sc->shapes[hs];
        Hit h; int tri_index;
        if (AN) hit_finish<STATS>(sc, sh, tc, mk(r0.x, r0.y, r0.z), r0.w, slot, h, tri_index, ct);
        else mesh_finish<STATS>(sc, sh, tc, mk(r0.x, r0.y, r0.z), r0.w, slot, h, tri_index, ct);
        if (!sh.has_material) { L = mk(0, 0, 0); done = true; }
        else {
            Ray out = ray;
            if (p.preview) {
                const Bounce pv = material_eval<true>(sc, sh, ray, h, out, rng);
                L = mk(0, 0, 0) + pv.att * h.color; done = true;
            } else {
                const Bounce b = material_eval<false>(sc, sh, ray, h, out, rng);
                if (rng.random() <= h.alpha) {
                    if (all_nonzero(b.att)) {
                        lv.at(nlev, 0) = make_float4(b.att.x, b.att.y, b.att.z, __int_as_float(0));
                        lv.at(nlev, 1) = make_float4(h.color.x, h.color.y, h.color.z, 0.0f);
                        lv.at(nlev, 2) = make_float4(b.em.x, b.em.y, b.em.z, 0.0f);
                        nlev++;
                        ray = out;
                    } else { L = mk(0, 0, 0) + b.em; done = true; }
                } else {
                    lv.at(nlev, 0) = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(2));
                    nlev++;
                    const float rd = ray.dist - h.dist;
                    ray.o = h.pos + ray.d * 0.0001f; ray.dist = rd;
                }
                if (!done) { depth--; if (depth == 0) { L = mk(0, 0, 0); done = true; } }
            }
        }
    }
    if (done) {
        for (int kk = nlev - 1; kk >= 0; kk--) {
            const float4 a = lv.at(kk, 0);
            if (__float_as_int(a.w) == 0) {
                const float4 c = lv.at(kk, 1), e = lv.at(kk, 2);
                L = (mk(0, 0, 0) + (mk(a.x, a.y, a.z) * L) * mk(c.x, c.y, c.z)) + mk(e.x, e.y, e.z);
            } else {
                L = mk(0, 0, 0) + L;
            }
        }
        pb.rad[pid] = make_float4(L.x, L.y, L.z, 0.0f);
        return false;
    }
    pb.state[(size_t)q * 3] = make_float4(ray.o.x, ray.o.y, ray.o.z, ray.dist);
    pb.state[(size_t)q * 3 + 1] = make_float4(ray.d.x, ray.d.y, ray.d.z, __uint_as_float(rng.counter));
    pb.state[(size_t)q * 3 + 2] = make_float4(__uint_as_float(rng.key), __uint_as_float(rng.table_reads),
                                              __uint_as_float(((uint32_t)depth << 16) | (uint32_t)nlev), 0.0f);
    if (AN && p.lead_shapes > 0) {  // the scene's leading spheres / planes / capsules: tested here, a ray per lane (in the trace kernel a whole
                                    // wave would repeat each test 64 times); the record is where the trace of this segment starts from
        int hs2 = -1, hslot = -1; f3 hp = mk(0, 0, 0); float seg = ray.dist;
        lead_find<STATS>(sc, p.lead_shapes, ray, seg, hs2, hslot, hp, ct);
        // can a later shape be hit at all?  Only if the ray's line meets its culling box (the reference's own early-out,
        // Src/RayTracerScene.cpp:109; a plane has none).  If no box is met the record above is the query's result.
        // With pruning on, a tame ray whose segment [0, seg] ends before the box or starts past it cannot be accepted by anything
        // inside either (the walk's own conservative cull, with its margins, applied to the shape's box: skipped tests are tests
        // the reference runs and rejects).
        const int n_shapes = sc->n_shapes;
        bool more = false; uint32_t tested = 0u;
        const bool cull = sc->prune != 0 && ray_is_tame(ray);
        const float eps_t = 2.0e-5f * fmaxf(fabsf(1.0f / ray.d.x), fmaxf(fabsf(1.0f / ray.d.y), fabsf(1.0f / ray.d.z)));
        for (int s = p.lead_shapes; s < n_shapes; s++) {
            const RtwShapeDev& sh = sc->shapes[s];
            float t0, t1;
            if (sh.kind == RTW_SHAPE_PLANE) { more = true; continue; }
            tested++;
            if (!slab_exact(ray, sh.bmin[0], sh.bmin[1], sh.bmin[2], sh.bmax[0], sh.bmax[1], sh.bmax[2], t0, t1)) continue;
            if (cull && sh.kind == RTW_SHAPE_MESH && (t0 > seg + (eps_t + 1.0e-4f * seg) || t1 < -eps_t)) continue;     // (meshes only: their walk applies the same cull)
            more = true;
        }
        if (STATS && !more) { ct.rays++; ct.boxes += tested; }       // (a ray that goes on to the trace kernel is counted there)
        pb.hitslot[(size_t)q * 2] = make_float4(hp.x, hp.y, hp.z, seg);
        pb.hitslot[(size_t)q * 2 + 1] = make_float4(__int_as_float(hs2), __int_as_float(hslot), __int_as_float(more ? 1 : 0), 0.0f);
    }
    return true;
}

// round >= 1: the paths of trace list (round - 1) have had their segment traced (the primary kernel was round 0)
template <bool STATS, bool AN>
__global__ __launch_bounds__(256, 3) void shade_kernel(const RtwSceneDev* __restrict__ sc, PipeBufs pb, RtwRenderParams p, int round)
{
    const bool from_queue = round == 1;         // the queue is round 0's trace list
    const uint32_t n = from_queue ? pb.counters[0] : pb.counters[4 + round - 1];
    const uint32_t* __restrict__ src = from_queue ? pb.queue : wf_list(pb, (round - 1) & 1);
    uint32_t* __restrict__ dst = round & 1 ? pb.tlist1 : pb.tlist0;
    const uint32_t nthreads = gridDim.x * blockDim.x;
    const int npix = p.width * p.height;
    const uint32_t phase = table_phase(p.seed);
    Counters ct = { 0, 0, 0, 0, 0, 0 };
    const uint32_t rounds_of_wave = (n + nthreads - 1) / nthreads;       // wave-uniform trip count: every lane joins the pushes
    for (uint32_t it = 0, k = blockIdx.x * blockDim.x + threadIdx.x; it < rounds_of_wave; it++, k += nthreads) {
        const bool live = k < n;
        const uint32_t q = live ? src[k] : 0u;
        bool go_on = false;                                                // this path has another segment to trace
        if (live && q < pb.capacity) {
            const uint32_t pid = pid_of_slot(p, q);
            const int wi = (int)(pid >> 2), sub = (int)(pid & 3u);
            const int pixel = work_to_pixel(p, wi);
            PathRng rng; Ray ray;
            const float4 s0 = pb.state[(size_t)q * 3], s1 = pb.state[(size_t)q * 3 + 1], s2 = pb.state[(size_t)q * 3 + 2];
            ray.o = mk(s0.x, s0.y, s0.z); ray.dist = s0.w; ray.d = mk(s1.x, s1.y, s1.z);
            rng.counter = __float_as_uint(s1.w); rng.key = __float_as_uint(s2.x); rng.table_reads = __float_as_uint(s2.y);
            rng.table_base = (((uint64_t)p.pass_index * (uint64_t)npix + (uint64_t)pixel) * 4u + (uint64_t)sub) * RTW_TABLE_STRIDE + phase;
            const int nlev = (int)(__float_as_uint(s2.z) & 0xFFFFu), depth = (int)(__float_as_uint(s2.z) >> 16);
            rng.pre_reads = 0xFFFFFFFFu; rng.pre_x = rng.pre_y = rng.pre_z = 0.0f;
            const float4 r0 = pb.hitslot[(size_t)q * 2], r1 = pb.hitslot[(size_t)q * 2 + 1];
            go_on = shade_hit_step<STATS, AN>(sc, pb, p, q, pid, ray, rng, depth, nlev, r0, r1, ct);
        }
        wave_push(dst, &pb.counters[4 + round], go_on, q);
    }
    if (STATS) flush_counters(sc, ct);
}

#include "rtw_wave_kernels.h"
#include "rtw_group_kernels.h"
#include "rtw_build_kernels.h"

__global__ __launch_bounds__(256) void resolve_kernel(const RtwSceneDev* __restrict__ sc, float4* __restrict__ accum,
                                                      uint32_t* __restrict__ argb, PipeBufs pb, RtwRenderParams p)
{
    __shared__ float thr[256];
    thr[threadIdx.x] = sc->gamma_thr[threadIdx.x];
    __syncthreads();
    const uint32_t n = pb.counters[1];
    const uint32_t nthreads = gridDim.x * blockDim.x;
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += nthreads) {
        const uint32_t wi = pb.pend[k];
        const int pixel = work_to_pixel(p, (int)wi);
        f3 c = mk(0, 0, 0);
        for (int i = 0; i < 4; i++) {
            if (i >= p.sub_samples) break;
            const float4 r = pb.rad[(size_t)wi * 4 + i];
            c = c + mk(r.x, r.y, r.z);
        }
        c = c / (float)p.sub_samples;
        resolve_pixel(thr, accum, argb, pixel, c, p.preview != 0);
    }
    // the pass's last kernel: the last block to finish files the counters for the host (queue lengths size the next pass's launches) and zeroes them
    // for the next pass (which saves it a memset launch).  Every block has read what it needs of them before it takes its ticket.
    __shared__ uint32_t last_block;
    __syncthreads();
    if (threadIdx.x == 0) last_block = atomicAdd(&pb.counters[40], 1u) == gridDim.x - 1u ? 1u : 0u;
    __syncthreads();
    if (last_block && threadIdx.x < 64) {
        const uint32_t v = threadIdx.x == 40 ? 0u : pb.counters[threadIdx.x];
        pb.counters[64 + threadIdx.x] = v;
        pb.counters[threadIdx.x] = 0u;
    }
}

template <bool STATS>
__global__ __launch_bounds__(256) void closest_kernel(const RtwSceneDev* __restrict__ sc, const float* __restrict__ rays, long long n,
                                                      float* __restrict__ hits11, int* __restrict__ shape, int* __restrict__ tri)
{
    const TravCtx tc = make_trav();
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Ray r; r.o = mk(rays[i * 7], rays[i * 7 + 1], rays[i * 7 + 2]); r.d = mk(rays[i * 7 + 3], rays[i * 7 + 4], rays[i * 7 + 5]); r.dist = rays[i * 7 + 6];
    Hit h; h.pos = mk(0, 0, 0); h.normal = mk(0, 0, 0); h.dist = 0.0f; h.color = mk(1, 1, 1); h.alpha = 1.0f;   // RayHitResult()
    Counters ct = { 0, 0, 0, 0, 0, 0 };
    int t = -1;
    const int s = find_intersection<STATS>(sc, tc, r, h, t, ct);
    float* o = hits11 + i * 11;
    o[0] = h.pos.x; o[1] = h.pos.y; o[2] = h.pos.z; o[3] = h.normal.x; o[4] = h.normal.y; o[5] = h.normal.z; o[6] = h.dist;
    o[7] = h.color.x; o[8] = h.color.y; o[9] = h.color.z; o[10] = h.alpha;
    shape[i] = s; tri[i] = s >= 0 ? t : -1;
    if (STATS) flush_counters(sc, ct);
}

template <bool STATS>
__global__ __launch_bounds__(256) void ray_trace_kernel(const RtwSceneDev* __restrict__ sc, const float* __restrict__ rays,
                                                        const uint32_t* __restrict__ keys2, long long n, int max_bounce, int preview,
                                                        uint32_t seed, unsigned long long npix, float* __restrict__ rgb, float4* __restrict__ ws)
{
    const TravCtx tc = make_trav();
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Ray r; r.o = mk(rays[i * 7], rays[i * 7 + 1], rays[i * 7 + 2]); r.d = mk(rays[i * 7 + 3], rays[i * 7 + 4], rays[i * 7 + 5]); r.dist = rays[i * 7 + 6];
    Counters ct = { 0, 0, 0, 0, 0, 0 };
    const uint32_t pixel = keys2[i * 2], sample = keys2[i * 2 + 1];
    PathRng rng; rng_init(rng, seed, table_phase(seed), npix, pixel, sample / 4u, sample % 4u);
    LevelStore lv; lv.ws = ws; lv.stride = (size_t)gridDim.x * blockDim.x; lv.tid = (size_t)i; lv.rec_levels = 0;
    const f3 c = trace_path<STATS>(sc, tc, r, max_bounce, preview != 0, rng, ct, lv);
    rgb[i * 3] = c.x; rgb[i * 3 + 1] = c.y; rgb[i * 3 + 2] = c.z;
    if (STATS) flush_counters(sc, ct);
}

__global__ void texture_sample_kernel(const RtwSceneDev* __restrict__ sc, int shape, int mat, const float* __restrict__ uv, long long n, float* __restrict__ rgba)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const RtwShapeDev& sh = sc->shapes[shape];
    f3 c; float a;
    texture_sample(sh.texels, sh.textures[mat], sc->texel_lut, uv[i * 2], uv[i * 2 + 1], c, a);
    rgba[i * 4] = c.x; rgba[i * 4 + 1] = c.y; rgba[i * 4 + 2] = c.z; rgba[i * 4 + 3] = a;
}

// ---- multi-GPU gather: a rank's task rows <-> one compact block ---------------------------------------------------------------
// The tasks of rank r (task_rows rows each, t = r, r + world, ...: Src/RayTracerProgram.cpp:282,294-301 dealt round-robin) lie scattered
// over the frame; RCCL moves ONE message per peer, so a sender first PACKS its rows into a compact block -- [ARGB of its pixels in task
// order | their accumulators] -- and the root UNPACKS every peer's block into its framebuffer (blockIdx.y = peer).  Pixel l of rank r's
// block is pixel (r + (l / task_px) * world) * task_px + l % task_px of the frame (only the frame's last task can be short, and it is
// the last of its owner's).  VEC: four pixels per thread (the frame's width divides by 4, so every offset does).
struct GatherPlan {
    uint32_t* argb; float4* accum;      // the framebuffer
    char* stage;                        // compact blocks: rank r's starts at stage + block_off[r - first_rank]
    int32_t task_px;                    // task_rows * width
    int32_t world, first_rank, with_accum;
    uint64_t block_off[64];             // byte offset of peer k's block (k = blockIdx.y), a multiple of 256
    uint32_t block_px[64];              // its pixels
};
template <bool UNPACK, bool VEC>
__global__ __launch_bounds__(256) void gather_rows_kernel(GatherPlan g)
{
    const uint32_t k = blockIdx.y;
    const uint32_t n = g.block_px[k];
    const uint32_t r = (uint32_t)g.first_rank + k;
    char* blk = g.stage + g.block_off[k];
    uint32_t* b_argb = reinterpret_cast<uint32_t*>(blk);
    float4* b_accum = reinterpret_cast<float4*>(blk + (((size_t)n * 4 + 255) & ~(size_t)255));
    const uint32_t step = VEC ? 4u : 1u;
    for (uint32_t l = (blockIdx.x * blockDim.x + threadIdx.x) * step; l < n; l += gridDim.x * blockDim.x * step) {
        const uint32_t j = l / (uint32_t)g.task_px, within = l - j * (uint32_t)g.task_px;
        const size_t pixel = (size_t)(r + j * (uint32_t)g.world) * (size_t)g.task_px + within;
        if (VEC) {
            float4* fa = reinterpret_cast<float4*>(g.argb + pixel); float4* ba = reinterpret_cast<float4*>(b_argb + l);      // (a 16-byte move; no arithmetic)
            if (UNPACK) *fa = *ba; else *ba = *fa;
        } else {
            if (UNPACK) g.argb[pixel] = b_argb[l]; else b_argb[l] = g.argb[pixel];
        }
        if (g.with_accum) {
            for (uint32_t e = 0; e < step; e++) {
                if (UNPACK) g.accum[pixel + e] = b_accum[l + e]; else b_accum[l + e] = g.accum[pixel + e];
            }
        }
    }
}

}  // namespace

// ---- launch wrappers ---------------------------------------------------------------------------------------
namespace rtw {

int launch_render(const RtwSceneDev* sc, void* accum, void* argb, void* ws, const RtwRenderParams& p, bool stats, hipStream_t stream)
{
    if (p.count <= 0) return 0;
    const int block = 256;
    const int grid = (p.count + block - 1) / block;
    if (stats) hipLaunchKernelGGL(render_kernel<true>, dim3(grid), dim3(block), 0, stream, sc, (float4*)accum, (uint32_t*)argb, (float4*)ws, p);
    else hipLaunchKernelGGL(render_kernel<false>, dim3(grid), dim3(block), 0, stream, sc, (float4*)accum, (uint32_t*)argb, (float4*)ws, p);
    return (int)hipGetLastError();
}

size_t pipeline_workspace_bytes(long long work_items, int max_bounce, PipelineLayout* out)
{
    // queue | pend | counters | rad | per-slot arrays: hit records, state, trace lists, level store (a slot per possible path: work items x 4)
    const size_t n = (size_t)(work_items > 0 ? work_items : 1);
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    PipelineLayout l;
    l.capacity = n * 4;
    l.queue_off = 0;
    l.pend_off = l.queue_off + up(l.capacity * 4);
    l.counters_off = l.pend_off + up(n * 4);
    l.rad_off = l.counters_off + 1024;      // 64 counters | 64 words: their values at the end of the previous pass (read back by the host) | spare
    l.hit_off = l.rad_off + up(l.capacity * 16);
    l.state_off = l.hit_off + up(l.capacity * 32);
    l.tlist0_off = l.state_off + up(l.capacity * 48);
    l.tlist1_off = l.tlist0_off + up(l.capacity * 4);
    l.ws_off = l.tlist1_off + up(l.capacity * 4);
    l.total = l.ws_off + l.capacity * (size_t)(max_bounce > 0 ? max_bounce : 1) * 3 * 16;
    if (out) *out = l;
    return l.total;
}

size_t pipeline_counters_offset(long long work_items, int max_bounce)
{
    PipelineLayout l;
    pipeline_workspace_bytes(work_items, max_bounce, &l);
    return l.counters_off;
}

// One pass through the bins + wave pipeline: primary_bins_kernel (+ primary_sky_kernel on the second stream), then trace(r - 1) / shade(r) for
// r = 1 .. max_bounce - 1 (the primary kernel was shade(0)), then resolve_kernel.
int launch_render_pipeline(const RtwSceneDev* sc, void* accum, void* argb, void* workspace, const RtwRenderParams& p, const PipelineTuning& tune, bool stats, hipStream_t stream)
{
    if (p.count <= 0) return 0;
    PipelineLayout l;
    pipeline_workspace_bytes(p.count, p.max_bounce, &l);
    char* w = (char*)workspace;
    PipeBufs pb;
    pb.queue = (uint32_t*)(w + l.queue_off); pb.pend = (uint32_t*)(w + l.pend_off); pb.counters = (uint32_t*)(w + l.counters_off);
    pb.rad = (float4*)(w + l.rad_off); pb.hitslot = (float4*)(w + l.hit_off); pb.state = (float4*)(w + l.state_off);
    pb.tlist0 = (uint32_t*)(w + l.tlist0_off); pb.tlist1 = (uint32_t*)(w + l.tlist1_off); pb.ws = (float4*)(w + l.ws_off); pb.capacity = (uint32_t)l.capacity;
    hipError_t e = hipSuccess;
    if (!tune.counters_clean) e = hipMemsetAsync(pb.counters, 0, 256, stream);     // else: the previous pass's resolve_kernel left them zeroed
    if (e != hipSuccess) return (int)e;
    // Paths to size the launches for.  The true queue length is only known on the device; the host passes the length the previous pass had
    // (frames of a progressive render barely differ) plus a margin.  Any shortfall is absorbed by stride loops, any excess by blocks that exit at once.
    long long owners = (long long)l.capacity;
    if (tune.expected_paths >= 0) {
        const long long want = (long long)tune.expected_paths + tune.expected_paths / 4 + 1024;
        if (want < owners) owners = want;
    }
    const int block = 256;
    const int grid = (p.count + block - 1) / block;
    int resolve_blocks = grid < 1024 ? grid : 1024;
    if (tune.expected_paths >= 0) {      // a thread per pending pixel (at most one per queued path), not per pixel of the frame
        const int want = (tune.expected_paths + tune.expected_paths / 4 + 1024 + 255) / 256;
        if (want < resolve_blocks) resolve_blocks = want;
    }
    if (tune.timing) (void)hipEventRecord(tune.timing[0], stream);
    bool forked = false;
    {
        RtwRenderParams ph = p;
        if (tune.aux_stream && p.tile_order && tune.sky_job0 > 0 && tune.sky_job0 < p.n_jobs && !stats) {
            // sky-only tiles on the second stream, beside everything else of this pass
            forked = !tune.do_fork || (hipEventRecord(tune.fork_event, stream) == hipSuccess && hipStreamWaitEvent(tune.aux_stream, tune.fork_event, 0) == hipSuccess);
            if (forked) {
                const int sky_jobs = p.n_jobs - tune.sky_job0;
                int sgrid = (sky_jobs + 3) / 4;
                if (sgrid > tune.cu_count * 64) sgrid = tune.cu_count * 64;
                hipLaunchKernelGGL(primary_sky_kernel, dim3(sgrid), dim3(block), 0, tune.aux_stream, tune.gamma_thr, (float4*)accum, (uint32_t*)argb, p, tune.sky_job0);
                if (tune.do_join) (void)hipEventRecord(tune.join_event, tune.aux_stream);
                else if (tune.aux_unjoined) *tune.aux_unjoined = true;
                ph.n_jobs = tune.sky_job0;
            }
        }
        const int jobs_grid = ph.tile_order ? (ph.n_jobs + 3) / 4 : grid;        // a wave per job (a tile, or a tile's sub-sample); else waves take jobs in turn
        const int pgrid = jobs_grid < tune.cu_count * 64 ? jobs_grid : tune.cu_count * 64;
        if (tune.has_analytic) {
            if (stats) hipLaunchKernelGGL((primary_bins_kernel<true, true>), dim3(pgrid), dim3(block), 0, stream, sc, (float4*)accum, (uint32_t*)argb, pb, ph);
            else hipLaunchKernelGGL((primary_bins_kernel<false, true>), dim3(pgrid), dim3(block), 0, stream, sc, (float4*)accum, (uint32_t*)argb, pb, ph);
        } else {
            if (stats) hipLaunchKernelGGL((primary_bins_kernel<true, false>), dim3(pgrid), dim3(block), 0, stream, sc, (float4*)accum, (uint32_t*)argb, pb, ph);
            else hipLaunchKernelGGL((primary_bins_kernel<false, false>), dim3(pgrid), dim3(block), 0, stream, sc, (float4*)accum, (uint32_t*)argb, pb, ph);
        }
    }
    if (tune.timing) (void)hipEventRecord(tune.timing[1], stream);
    auto items_of = [&](int round) {
        if (round <= 1) return owners;      // the queue is round 0's trace list = round 1's input
        const long long items = tune.round_hint[round - 1] >= 0 ? (long long)tune.round_hint[round - 1] + tune.round_hint[round - 1] / 4 + 256 : owners;
        return items > owners ? owners : items;
    };
    constexpr int NTV = 128;                // wave-per-ray trace launches: 128-thread blocks (measured a little faster than 256: fewer waves coupled to one block)
    const size_t dyn = (size_t)(NTV / 64) * RTW_WAVE_LDS_WORDS * 4;
    const long long trace_cap = (long long)tune.cu_count * (1024 / NTV) * 8;
    for (int r = 1; r < p.max_bounce; r++) {
        if (!tune.skip_trace) {
            const long long rays = items_of(r);
            if (p.lead_shapes > 0) {        // leading analytic shapes: the records are half done, most rays need no trace (see trace_wave_lead_kernel)
                int shift = 0;              // measured on SetupScene: 4 entries per wave at a time 1.64 ms, 16: 1.67, 64: 1.84, 1: 1.80 (flagged rays cluster: small chunks spread them over the waves)
                while (shift < 2 && (rays >> (shift + 1)) >= 16384) shift++;
                long long blocks = ((rays >> shift) + NTV / 64) / (NTV / 64);
                if (blocks < 1) blocks = 1;
                if (blocks > trace_cap) blocks = trace_cap;
                if (stats) hipLaunchKernelGGL((trace_wave_lead_kernel<true, NTV>), dim3((unsigned)blocks), dim3(NTV), dyn, stream, sc, pb, p, r - 1, shift);
                else hipLaunchKernelGGL((trace_wave_lead_kernel<false, NTV>), dim3((unsigned)blocks), dim3(NTV), dyn, stream, sc, pb, p, r - 1, shift);
            } else {
                long long blocks = (rays + NTV / 64 - 1) / (NTV / 64);
                if (blocks < 1) blocks = 1;
                if (blocks > trace_cap) blocks = trace_cap;
                if (tune.has_analytic) {    // analytic shapes somewhere after a mesh: the general wave-per-ray query
                    if (stats) hipLaunchKernelGGL((trace_wave_kernel<true, NTV, true>), dim3((unsigned)blocks), dim3(NTV), dyn, stream, sc, pb, p, r - 1);
                    else hipLaunchKernelGGL((trace_wave_kernel<false, NTV, true>), dim3((unsigned)blocks), dim3(NTV), dyn, stream, sc, pb, p, r - 1);
                } else {
                    if (stats) hipLaunchKernelGGL((trace_wave_kernel<true, NTV, false>), dim3((unsigned)blocks), dim3(NTV), dyn, stream, sc, pb, p, r - 1);
                    else hipLaunchKernelGGL((trace_wave_kernel<false, NTV, false>), dim3((unsigned)blocks), dim3(NTV), dyn, stream, sc, pb, p, r - 1);
                }
            }
        }
        long long sb = (items_of(r) + 255) / 256;
        if (sb < 1) sb = 1;
        if (sb > 262144) sb = 262144;
        if (tune.has_analytic) {
            if (stats) hipLaunchKernelGGL((shade_kernel<true, true>), dim3((unsigned)sb), dim3(256), 0, stream, sc, pb, p, r);
            else hipLaunchKernelGGL((shade_kernel<false, true>), dim3((unsigned)sb), dim3(256), 0, stream, sc, pb, p, r);
        } else {
            if (stats) hipLaunchKernelGGL((shade_kernel<true, false>), dim3((unsigned)sb), dim3(256), 0, stream, sc, pb, p, r);
            else hipLaunchKernelGGL((shade_kernel<false, false>), dim3((unsigned)sb), dim3(256), 0, stream, sc, pb, p, r);
        }
    }
    if (tune.timing) (void)hipEventRecord(tune.timing[2], stream);
    hipLaunchKernelGGL(resolve_kernel, dim3(resolve_blocks), dim3(block), 0, stream, sc, (float4*)accum, (uint32_t*)argb, pb, p);
    if (forked && tune.do_join) {           // the pass (or the run of passes) is complete when both streams are
        (void)hipStreamWaitEvent(stream, tune.join_event, 0);
        if (tune.aux_unjoined) *tune.aux_unjoined = false;
    }
    if (tune.timing) (void)hipEventRecord(tune.timing[3], stream);
    return (int)hipGetLastError();
}

// ---- pass-batched pipeline (rtw_group_kernels.h) -----------------------------------------------------------------------------
size_t group_workspace_bytes(size_t capacity, int max_bounce, bool carry, GroupLayout* out)
{
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t cap = capacity > 0 ? capacity : 1;
    GroupLayout l;
    l.counters_off = 0;                         // 128 words at the start: the offset never moves when the workspace grows
    l.rad_off = 1024;
    l.state_off = l.rad_off + up(cap * 16);
    l.hit_off = l.state_off + up(cap * 48);
    l.carry_off = l.hit_off + up(cap * 32);
    l.levels_off = l.carry_off + (carry ? up(cap * 16) : 0);
    l.list0_off = l.levels_off + up(cap * (size_t)(max_bounce > 0 ? max_bounce : 1) * 48);
    l.list1_off = l.list0_off + up(cap * 4);
    l.overflow_off = l.list1_off + up(cap * 4);
    l.tlist0_off = l.overflow_off + up(cap * 4);
    l.tlist1_off = l.tlist0_off + up(cap * 4);
    l.total = l.tlist1_off + up(cap * 4);
    if (out) *out = l;
    return l.total;
}

int launch_render_group(const RtwSceneDev* sc, void* accum, void* argb, void* workspace, const RtwGroupParams& g, const GroupTuning& tune, bool stats, hipStream_t stream)
{
    const RtwRenderParams& p = g.rp;
    if (g.n_passes <= 0 || (g.n_busy <= 0 && g.n_sky <= 0)) return 0;
    if (tune.sky_mode == 2) {       // only the sky tiles of a split group, all the group's passes
        if (g.n_sky > 0) {
            RtwGroupParams gs = g;
            gs.first_pass = tune.sky_first_pass; gs.n_passes = tune.sky_passes;
            int sgrid = (gs.n_sky + 3) / 4;
            if (sgrid > tune.cu_count * tune.sky_blocks) sgrid = tune.cu_count * tune.sky_blocks;
            hipLaunchKernelGGL(gsky_kernel, dim3(sgrid), dim3(256), 0, stream, tune.gamma_thr, (float4*)accum, (uint32_t*)argb, gs);
        }
        return (int)hipGetLastError();
    }
    GroupLayout l;
    group_workspace_bytes(tune.capacity, p.max_bounce, tune.carry, &l);
    char* w = (char*)workspace;
    GroupBufs gb;
    gb.rad = (float4*)(w + l.rad_off); gb.state = (float4*)(w + l.state_off); gb.hit = (float4*)(w + l.hit_off);
    gb.carry = tune.carry ? (float4*)(w + l.carry_off) : nullptr; gb.levels = (float4*)(w + l.levels_off);
    gb.list0 = (uint32_t*)(w + l.list0_off); gb.list1 = (uint32_t*)(w + l.list1_off); gb.overflow = (uint32_t*)(w + l.overflow_off); gb.tlist0 = (uint32_t*)(w + l.tlist0_off); gb.tlist1 = (uint32_t*)(w + l.tlist1_off); gb.counters = (uint32_t*)(w + l.counters_off);
    gb.capacity = (uint32_t)tune.capacity; gb.carry_on = tune.carry ? 1 : 0;
    if (!tune.counters_clean) { hipError_t e = hipMemsetAsync(gb.counters, 0, 256, stream); if (e != hipSuccess) return (int)e; }
    bool forked = false;
    if (tune.sky_mode == 1) {
        // a part of a split group: the sky tiles are not its business
    } else if (g.n_sky > 0) {
        RtwGroupParams gs = g;
        hipStream_t ss = stream;
        if (tune.aux_stream) {
            forked = !tune.do_fork || (hipEventRecord(tune.fork_event, stream) == hipSuccess && hipStreamWaitEvent(tune.aux_stream, tune.fork_event, 0) == hipSuccess);
            if (forked) ss = tune.aux_stream;
        }
        int sgrid = (g.n_sky + 3) / 4;
        if (sgrid > tune.cu_count * tune.sky_blocks) sgrid = tune.cu_count * tune.sky_blocks;
        hipLaunchKernelGGL(gsky_kernel, dim3(sgrid), dim3(256), 0, ss, tune.gamma_thr, (float4*)accum, (uint32_t*)argb, gs);
        if (forked) {
            if (tune.do_join) (void)hipEventRecord(tune.join_event, tune.aux_stream);
            else if (tune.aux_unjoined) *tune.aux_unjoined = true;
        }
    }
    if (tune.timing) (void)hipEventRecord(tune.timing[0], stream);
    if (g.n_busy > 0) {
        const long long live_paths = (long long)g.n_busy * 64 * p.sub_samples * g.n_passes;
        {
            const int ppw = (g.primary_passes > 1 && g.primary_passes * p.sub_samples <= 4) ? g.primary_passes : 1;        // as the kernel reads it
            const long long waves = (long long)g.n_jobs * ((g.n_passes + ppw - 1) / ppw);
            const unsigned pgrid = (unsigned)((waves + 3) / 4);
            if (tune.has_analytic) {
                if (stats) hipLaunchKernelGGL((gprimary_kernel<true, true>), dim3(pgrid), dim3(256), 0, stream, sc, gb, g);
                else hipLaunchKernelGGL((gprimary_kernel<false, true>), dim3(pgrid), dim3(256), 0, stream, sc, gb, g);
            } else {
                if (stats) hipLaunchKernelGGL((gprimary_kernel<true, false>), dim3(pgrid), dim3(256), 0, stream, sc, gb, g);
                else hipLaunchKernelGGL((gprimary_kernel<false, false>), dim3(pgrid), dim3(256), 0, stream, sc, gb, g);
            }
        }
        if (tune.timing) (void)hipEventRecord(tune.timing[1], stream);
        auto blocks_for = [&](int list_round) {      // blocks of 256 lanes for the paths of list `list_round` (a grid-stride loop takes any excess)
            long long items = live_paths;
            if (tune.round_hint[list_round] >= 0) {
                const long long want = (long long)tune.round_hint[list_round] + tune.round_hint[list_round] / 4 + 1024;
                if (want < items) items = want;
            }
            long long bl = (items + 255) / 256;
            if (bl < 1) bl = 1;
            if (bl > 65536) bl = 65536;
            return (unsigned)bl;
        };
        if (!p.preview) {
            for (int r = 1; r < p.max_bounce; r++) {       // the primary kernel was shade(0); trace(r - 1) then shade(r)
                const unsigned sb_full = blocks_for(r - 1);  // the round's whole list (the shade launch)
                // the rays the trace launch takes: the whole list, or (leading analytic shapes) the part the shading lanes could not finish
                const int trace_hint = p.lead_shapes > 0 ? tune.trace_hint[r - 1] : tune.round_hint[r - 1];
                unsigned tb = sb_full;
                if (p.lead_shapes > 0 && trace_hint >= 0) { long long bl = ((long long)trace_hint + trace_hint / 4 + 1024 + 255) / 256; if (bl < (long long)tb) tb = (unsigned)(bl < 1 ? 1 : bl); }
                if (!tune.skip_trace) {
#define RTW_LAUNCH_GT(ST, AN_, NT_, CAP_, STG, BLOCKS, DYN)                                                                                      \
                do {                                                                                                                            \
                    if ((DYN) > 65536) (void)hipFuncSetAttribute((const void*)gtrace_kernel<ST, AN_, NT_, CAP_, STG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(DYN)); \
                    hipLaunchKernelGGL((gtrace_kernel<ST, AN_, NT_, CAP_, STG>), dim3(BLOCKS), dim3(NT_), (DYN), stream, sc, gb, r - 1, tune.staged_shape, p.lead_shapes); \
                } while (0)
#define RTW_LAUNCH_GT4(NT_, CAP_, STG, BLOCKS, DYN)                                                                                             \
                do {                                                                                                                            \
                    if (tune.has_analytic) { if (stats) RTW_LAUNCH_GT(true, true, NT_, CAP_, STG, BLOCKS, DYN); else RTW_LAUNCH_GT(false, true, NT_, CAP_, STG, BLOCKS, DYN); } \
                    else { if (stats) RTW_LAUNCH_GT(true, false, NT_, CAP_, STG, BLOCKS, DYN); else RTW_LAUNCH_GT(false, false, NT_, CAP_, STG, BLOCKS, DYN); } \
                } while (0)
                if (!tune.carry && trace_hint >= 0 && trace_hint < tune.wave_below) {
                    // a short list: a wave per ray (128-thread blocks, a wave takes rays in turn)
                    constexpr int NTV = 128;
                    long long blocks = ((long long)trace_hint + trace_hint / 4 + 64 + NTV / 64 - 1) / (NTV / 64);
                    const long long cap = (long long)tune.cu_count * 64;
                    if (blocks < 1) blocks = 1;
                    if (blocks > cap) blocks = cap;
                    const size_t dyn = (size_t)(NTV / 64) * RTW_WAVE_LDS_WORDS * 4;
                    if (tune.has_analytic) {
                        if (stats) hipLaunchKernelGGL((gtrace_wave_kernel<true, true, NTV>), dim3((unsigned)blocks), dim3(NTV), dyn, stream, sc, gb, r - 1, 0, p.lead_shapes);
                        else hipLaunchKernelGGL((gtrace_wave_kernel<false, true, NTV>), dim3((unsigned)blocks), dim3(NTV), dyn, stream, sc, gb, r - 1, 0, p.lead_shapes);
                    } else {
                        if (stats) hipLaunchKernelGGL((gtrace_wave_kernel<true, false, NTV>), dim3((unsigned)blocks), dim3(NTV), dyn, stream, sc, gb, r - 1, 0, p.lead_shapes);
                        else hipLaunchKernelGGL((gtrace_wave_kernel<false, false, NTV>), dim3((unsigned)blocks), dim3(NTV), dyn, stream, sc, gb, r - 1, 0, p.lead_shapes);
                    }
                } else if ((tune.single_mesh || tune.lead_mesh) && tune.staged_top > 0) {
                    // one mesh: persistent waves that refill their lanes (one block per CU when the tree's upper levels are staged)
                    const int budget_r = tune.visit_budget;
#define RTW_LAUNCH_GPL(NT_, CAP_, STG, LD, PL, BLOCKS, DYN)                                                                                      \
                    do {                                                                                                                        \
                        if (stats) { if ((DYN) > 65536) (void)hipFuncSetAttribute((const void*)gtrace_persist_kernel<true, NT_, CAP_, STG, LD, PL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(DYN)); \
                            hipLaunchKernelGGL((gtrace_persist_kernel<true, NT_, CAP_, STG, LD, PL>), dim3(BLOCKS), dim3(NT_), (DYN), stream, sc, gb, r - 1, budget_r, tune.staged_shape); } \
                        else { if ((DYN) > 65536) (void)hipFuncSetAttribute((const void*)gtrace_persist_kernel<false, NT_, CAP_, STG, LD, PL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(DYN)); \
                            hipLaunchKernelGGL((gtrace_persist_kernel<false, NT_, CAP_, STG, LD, PL>), dim3(BLOCKS), dim3(NT_), (DYN), stream, sc, gb, r - 1, budget_r, tune.staged_shape); } \
                    } while (0)
#define RTW_LAUNCH_GP(NT_, CAP_, STG, PL, BLOCKS, DYN) do { if (tune.lead_mesh) RTW_LAUNCH_GPL(NT_, CAP_, STG, true, PL, BLOCKS, DYN); else RTW_LAUNCH_GPL(NT_, CAP_, STG, false, PL, BLOCKS, DYN); } while (0)
                    {
                        unsigned sbl = (tb + 3) / 4;
                        if (sbl > (unsigned)tune.cu_count) sbl = (unsigned)tune.cu_count;
                        const size_t dyn = (size_t)RTW_GT_CAP_STAGED * 1024 * 4 + (size_t)tune.staged_top * 32;
                        if (tune.staged_all && tune.staged_planes) RTW_LAUNCH_GP(1024, RTW_GT_CAP_STAGED, 2, true, sbl, dyn + (size_t)tune.staged_tris * 16);
                        else if (tune.staged_all) RTW_LAUNCH_GP(1024, RTW_GT_CAP_STAGED, 2, false, sbl, dyn);
                        else RTW_LAUNCH_GP(1024, RTW_GT_CAP_STAGED, 1, false, sbl, dyn);
                    }
#undef RTW_LAUNCH_GP
#undef RTW_LAUNCH_GPL
                    if (tune.visit_budget < INT32_MAX) {        // the rays that ran out of budget: a wave each
                        constexpr int NTV = 128;
                        // a generous grid: the count varies from group to group, a wave-per-ray loop with too few waves is slow (measured: 0.9 ms for a few
                        // thousand rays on 256 waves), blocks without a ray leave at once
                        long long want = tune.overflow_hint[r - 1] >= 0 ? (long long)tune.overflow_hint[r - 1] * 4 + 8192 : 16384;
                        long long blocks = (want + NTV / 64 - 1) / (NTV / 64);
                        const long long cap = (long long)tune.cu_count * 64;
                        if (blocks > cap) blocks = cap;
                        const size_t dyn = (size_t)(NTV / 64) * RTW_WAVE_LDS_WORDS * 4;
                        if (tune.lead_mesh) {
                            if (stats) hipLaunchKernelGGL((gtrace_wave_kernel<true, true, NTV>), dim3((unsigned)blocks), dim3(NTV), dyn, stream, sc, gb, r - 1, 1, p.lead_shapes);
                            else hipLaunchKernelGGL((gtrace_wave_kernel<false, true, NTV>), dim3((unsigned)blocks), dim3(NTV), dyn, stream, sc, gb, r - 1, 1, p.lead_shapes);
                        } else {
                            if (stats) hipLaunchKernelGGL((gtrace_wave_kernel<true, false, NTV>), dim3((unsigned)blocks), dim3(NTV), dyn, stream, sc, gb, r - 1, 1, 0);
                            else hipLaunchKernelGGL((gtrace_wave_kernel<false, false, NTV>), dim3((unsigned)blocks), dim3(NTV), dyn, stream, sc, gb, r - 1, 1, 0);
                        }
                    }
                } else if (tune.staged_shape >= 0 && tune.staged_top > 0) {
                    const unsigned sbl = (tb + 3) / 4;          // 1024-thread blocks
                    const size_t dyn = (size_t)RTW_GT_CAP_STAGED * 1024 * 4 + (size_t)tune.staged_top * 32;
                    if (tune.staged_all) RTW_LAUNCH_GT4(1024, RTW_GT_CAP_STAGED, 2, sbl, dyn);
                    else RTW_LAUNCH_GT4(1024, RTW_GT_CAP_STAGED, 1, sbl, dyn);
                } else {
                    const size_t dyn = (size_t)RTW_GT_CAP * 256 * 4;
                    RTW_LAUNCH_GT4(256, RTW_GT_CAP, 0, tb, dyn);
                }
#undef RTW_LAUNCH_GT4
#undef RTW_LAUNCH_GT
                }
                if (tune.has_analytic) {
                    if (stats) hipLaunchKernelGGL((gshade_kernel<true, true>), dim3(sb_full), dim3(256), 0, stream, sc, gb, g, r);
                    else hipLaunchKernelGGL((gshade_kernel<false, true>), dim3(sb_full), dim3(256), 0, stream, sc, gb, g, r);
                } else {
                    if (stats) hipLaunchKernelGGL((gshade_kernel<true, false>), dim3(sb_full), dim3(256), 0, stream, sc, gb, g, r);
                    else hipLaunchKernelGGL((gshade_kernel<false, false>), dim3(sb_full), dim3(256), 0, stream, sc, gb, g, r);
                }
            }
        }
        if (tune.timing) (void)hipEventRecord(tune.timing[2], stream);
        if (tune.resolve_after) (void)hipStreamWaitEvent(stream, tune.resolve_after, 0);      // a pixel's passes are added in pass order
        hipLaunchKernelGGL(gresolve_kernel, dim3((unsigned)((g.n_busy + 3) / 4)), dim3(256), 0, stream, sc, (float4*)accum, (uint32_t*)argb, gb, g);
        if (tune.resolve_done) (void)hipEventRecord(tune.resolve_done, stream);
    } else if (tune.timing) {
        (void)hipEventRecord(tune.timing[1], stream); (void)hipEventRecord(tune.timing[2], stream);
    }
    if (forked && tune.do_join) {
        (void)hipStreamWaitEvent(stream, tune.join_event, 0);
        if (tune.aux_unjoined) *tune.aux_unjoined = false;
    }
    if (tune.timing) (void)hipEventRecord(tune.timing[3], stream);
    return (int)hipGetLastError();
}

// ---- KdNode::Build + derived layouts on the device (rtw_build_kernels.h) ------------------------------------------------------
namespace {
struct TempBufs {
    std::vector<void*> v;
    ~TempBufs() { for (void* p : v) (void)hipFree(p); }
    template <typename T> hipError_t get(T** out, size_t count) { void* d = nullptr; const hipError_t e = hipMalloc(&d, (count ? count : 1) * sizeof(T)); if (e == hipSuccess) v.push_back(d); *out = (T*)d; return e; }
};
}  // namespace

#define RTW_HIP_OK(expr) do { const hipError_t e_ = (expr); if (e_ != hipSuccess) return (int)e_; } while (0)
#define BUILD_MARK(what) do { if (btrace) { (void)hipStreamSynchronize(stream); timespec ts_; clock_gettime(CLOCK_MONOTONIC, &ts_); const double t_ = ts_.tv_sec * 1e3 + ts_.tv_nsec * 1e-6; fprintf(stderr, "    build: %-20s %.3f ms\n", what, t_ - bt_last); bt_last = t_; } } while (0)
int device_build_mesh(const DeviceBuildIn& in, int top_budget, DeviceBuildOut* out, hipStream_t stream)
{
    static const bool btrace = std::getenv("RTW_COMMIT_TRACE") != nullptr;
    double bt_last = 0; { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); bt_last = ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
    const int n = in.n_tris, n_nodes = 2 * n - 1;
    if (n <= 0) return (int)hipErrorInvalidValue;
    TempBufs tmp;
    BUILD_MARK("entry");
    float *d_pts, *d_tcs, *d_nrm; int32_t *d_ip, *d_it, *d_in, *d_mat, *d_leaf, *d_depth, *d_place, *d_ntop; BuildTri* d_rec[2];
    BuildNode* d_lvl[2]; uint32_t* d_cnt;      // d_cnt[0..1]: node counts of the two level lists, d_cnt[2..65]: nodes per depth
    RTW_HIP_OK(tmp.get(&d_pts, (size_t)in.n_points * 3)); RTW_HIP_OK(tmp.get(&d_tcs, (size_t)in.n_texcoords * 3)); RTW_HIP_OK(tmp.get(&d_nrm, (size_t)in.n_normals * 3));
    RTW_HIP_OK(tmp.get(&d_ip, (size_t)n * 3)); RTW_HIP_OK(tmp.get(&d_it, (size_t)n * 3)); RTW_HIP_OK(tmp.get(&d_in, (size_t)n * 3)); RTW_HIP_OK(tmp.get(&d_mat, (size_t)n));
    RTW_HIP_OK(tmp.get(&d_rec[0], (size_t)n)); RTW_HIP_OK(tmp.get(&d_rec[1], (size_t)n)); RTW_HIP_OK(tmp.get(&d_leaf, (size_t)n));
    RTW_HIP_OK(tmp.get(&d_depth, (size_t)n_nodes)); RTW_HIP_OK(tmp.get(&d_place, (size_t)n_nodes + 1)); RTW_HIP_OK(tmp.get(&d_ntop, 1));
    RTW_HIP_OK(tmp.get(&d_lvl[0], (size_t)n + 2)); RTW_HIP_OK(tmp.get(&d_lvl[1], (size_t)n + 2)); RTW_HIP_OK(tmp.get(&d_cnt, 66));
    RTW_HIP_OK(hipMemcpyAsync(d_pts, in.points, (size_t)in.n_points * 12, hipMemcpyHostToDevice, stream));
    RTW_HIP_OK(hipMemcpyAsync(d_tcs, in.texcoords, (size_t)in.n_texcoords * 12, hipMemcpyHostToDevice, stream));
    RTW_HIP_OK(hipMemcpyAsync(d_nrm, in.normals, (size_t)in.n_normals * 12, hipMemcpyHostToDevice, stream));
    RTW_HIP_OK(hipMemcpyAsync(d_ip, in.idx_p, (size_t)n * 12, hipMemcpyHostToDevice, stream));
    RTW_HIP_OK(hipMemcpyAsync(d_it, in.idx_t, (size_t)n * 12, hipMemcpyHostToDevice, stream));
    RTW_HIP_OK(hipMemcpyAsync(d_in, in.idx_n, (size_t)n * 12, hipMemcpyHostToDevice, stream));
    RTW_HIP_OK(hipMemcpyAsync(d_mat, in.tri_material, (size_t)n * 4, hipMemcpyHostToDevice, stream));
    BUILD_MARK("temps + uploads");
    // outputs (owned by the caller afterwards; an early return frees what was allocated so far)
    DeviceBuildOut o; std::memset(&o, 0, sizeof o);
    struct OutGuard {
        DeviceBuildOut* o; bool keep = false;
        ~OutGuard() { if (!keep) { (void)hipFree(o->nodes); (void)hipFree(o->tnodes); (void)hipFree(o->tris); (void)hipFree(o->shade); for (int l = 0; l < 3; l++) (void)hipFree(o->flat[l]); } }
    } guard{ &o };
    o.n_nodes = n_nodes;
    RTW_HIP_OK(hipMalloc((void**)&o.nodes, (size_t)n_nodes * sizeof(RtwNode)));
    RTW_HIP_OK(hipMalloc((void**)&o.tnodes, (size_t)n_nodes * sizeof(RtwPNode)));
    RTW_HIP_OK(hipMalloc((void**)&o.tris, (size_t)n * sizeof(RtwTri)));
    RTW_HIP_OK(hipMalloc((void**)&o.shade, (size_t)n * sizeof(RtwShade)));
    {
        int cnt = n;
        for (int l = 0; l < 3; l++) {
            o.flat_n[l] = cnt; o.flat_pad[l] = ((cnt + 63) / 64) * 64 + 64;
            RTW_HIP_OK(hipMalloc((void**)&o.flat[l], (size_t)o.flat_pad[l] * 24));
            RTW_HIP_OK(hipMemsetAsync(o.flat[l], 0, (size_t)o.flat_pad[l] * 24, stream));
            cnt = (cnt + 15) / 16;
        }
    }
    BUILD_MARK("output buffers");
    // the recursion, level by level
    hipLaunchKernelGGL(build_tri_prep_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, d_pts, d_ip, n, d_rec[0]);      // centroid + own box of every triangle, once
    RTW_HIP_OK(hipMemsetAsync(d_cnt, 0, 66 * 4, stream));
    {
        const BuildNode root = { 0, n, 0, 0 };
        const uint32_t one = 1u;
        RTW_HIP_OK(hipMemcpyAsync(d_lvl[0], &root, sizeof root, hipMemcpyHostToDevice, stream));
        RTW_HIP_OK(hipMemcpyAsync(&d_cnt[0], &one, 4, hipMemcpyHostToDevice, stream));
        RTW_HIP_OK(hipMemcpyAsync(&d_cnt[2], &one, 4, hipMemcpyHostToDevice, stream));
        RTW_HIP_OK(hipStreamSynchronize(stream));       // (root / one live on this frame)
    }
    int level = 0;
    for (;;) {
        for (int k = 0; k < 8; k++, level++) {          // eight levels per read-back
            const int c = level & 1;
            RTW_HIP_OK(hipMemsetAsync(&d_cnt[1 - c], 0, 4, stream));
            // a wave per node; at most n nodes on a level, the stride loop takes any excess
            const int block_per_node = level <= 7 ? 1 : 0;      // at most 128 nodes: a block each (its four waves share a big node)
            unsigned blocks = block_per_node ? (unsigned)(1 << level) : (unsigned)(((long long)(n < (1 << level) || level > 20 ? n : (1 << level)) + 3) / 4);
            if (blocks > 8192u) blocks = 8192u;
            if (blocks < 1u) blocks = 1u;
            hipLaunchKernelGGL(build_level_kernel, dim3(blocks), dim3(256), 0, stream, d_pts, d_ip, d_rec[c], d_rec[1 - c], d_leaf, d_lvl[c], &d_cnt[c], d_lvl[1 - c], &d_cnt[1 - c],
                               o.nodes, d_depth, &d_cnt[2], block_per_node);
        }
        uint32_t left = 0;
        RTW_HIP_OK(hipMemcpyAsync(&left, &d_cnt[level & 1], 4, hipMemcpyDeviceToHost, stream));
        RTW_HIP_OK(hipStreamSynchronize(stream));
        if (left == 0u) break;
        if (level > 4 * n + 64) return (int)hipErrorUnknown;        // cannot happen: every level splits every segment
    }
    BUILD_MARK("levels");
    // depth statistics -> how many levels fit the LDS budget of the trace kernels (the host build's rule)
    uint32_t per_level[64];
    RTW_HIP_OK(hipMemcpyAsync(per_level, &d_cnt[2], sizeof per_level, hipMemcpyDeviceToHost, stream));
    RTW_HIP_OK(hipStreamSynchronize(stream));
    int D = -1, count = 0, deepest = 0;
    for (int d = 0; d < 64; d++) if (per_level[d] > 0) deepest = d;
    for (int d = 0; d < 64; d++) { if (per_level[d] == 0 || count + (int)per_level[d] > top_budget) break; count += (int)per_level[d]; D = d; }
    o.max_depth = deepest + 1;
    hipLaunchKernelGGL(build_tnode_places_kernel, dim3(1), dim3(1024), 0, stream, d_depth, n_nodes, D, d_place, d_ntop);
    hipLaunchKernelGGL(build_tnodes_kernel, dim3((n_nodes + 255) / 256), dim3(256), 0, stream, o.nodes, d_place, n_nodes, o.tnodes);
    hipLaunchKernelGGL(build_leaf_records_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, d_pts, d_tcs, d_nrm, d_ip, d_it, d_in, d_mat, d_leaf, n, o.tris, o.shade);
    hipLaunchKernelGGL(build_flat0_kernel, dim3((n_nodes + 255) / 256), dim3(256), 0, stream, o.nodes, n_nodes, o.flat[0]);
    for (int l = 1; l < 3; l++)
        hipLaunchKernelGGL(build_flat_up_kernel, dim3((o.flat_n[l] + 255) / 256), dim3(256), 0, stream, o.flat[l - 1], o.flat_n[l - 1], o.flat[l], o.flat_n[l]);
    int32_t ntop = 0;
    RTW_HIP_OK(hipMemcpyAsync(&ntop, d_ntop, 4, hipMemcpyDeviceToHost, stream));
    RTW_HIP_OK(hipStreamSynchronize(stream));
    RTW_HIP_OK(hipGetLastError());
    BUILD_MARK("derived layouts");
    o.tnodes_top = D < 0 ? 0 : ntop;
    *out = o;
    guard.keep = true;
    return 0;
}
// Screen bins of one mesh on the device.  *off_out (n_bins + 1 offsets) and *ent_out are the caller's (hipFree); h_off receives the offsets
// (the host sorts the tiles by list length).  Returns 0 and *has_bins = 0 when the mesh gets no bins (as build_bins returns false).
int device_build_bins(const RtwNode* d_nodes, const RtwTri* d_tris, int n_nodes, int width, int height, int bin_w, int bin_h,
                      uint32_t** off_out, uint32_t** ent_out, uint32_t* h_off, int* has_bins, hipStream_t stream)
{
    BinsGeom g; g.width = width; g.height = height; g.bin_w = bin_w; g.bin_h = bin_h;
    g.bx = (width + bin_w - 1) / bin_w; g.by = (height + bin_h - 1) / bin_h;
    g.margin = 1.0 + 1.25 * (double)height / (2.0 * (double)width);
    const int n_bins = g.bx * g.by;
    *off_out = nullptr; *ent_out = nullptr; *has_bins = 0;
    TempBufs tmp;
    uint32_t *d_counts, *d_flag;
    RTW_HIP_OK(tmp.get(&d_counts, (size_t)n_bins)); RTW_HIP_OK(tmp.get(&d_flag, 1));
    uint32_t* d_off = nullptr;
    RTW_HIP_OK(hipMalloc((void**)&d_off, ((size_t)n_bins + 1) * 4));
    RTW_HIP_OK(hipMemsetAsync(d_counts, 0, (size_t)n_bins * 4, stream));
    RTW_HIP_OK(hipMemsetAsync(d_flag, 0, 4, stream));
    const unsigned nb = (unsigned)((n_nodes + 255) / 256);
    hipLaunchKernelGGL(bins_pass_kernel<0>, dim3(nb), dim3(256), 0, stream, d_nodes, d_tris, n_nodes, g, d_counts, (const uint32_t*)nullptr, (uint32_t*)nullptr, d_flag);
    hipLaunchKernelGGL(bins_scan_kernel, dim3(1), dim3(1024), 0, stream, d_counts, d_off, n_bins);
    uint32_t flag = 0;
    RTW_HIP_OK(hipMemcpyAsync(h_off, d_off, ((size_t)n_bins + 1) * 4, hipMemcpyDeviceToHost, stream));
    RTW_HIP_OK(hipMemcpyAsync(&flag, d_flag, 4, hipMemcpyDeviceToHost, stream));
    RTW_HIP_OK(hipStreamSynchronize(stream));
    if (flag != 0u) { (void)hipFree(d_off); return 0; }
    const uint32_t total = h_off[n_bins];
    uint32_t* d_ent = nullptr;
    RTW_HIP_OK(hipMalloc((void**)&d_ent, ((size_t)total + 1) * 4));
    hipLaunchKernelGGL(bins_pass_kernel<1>, dim3(nb), dim3(256), 0, stream, d_nodes, d_tris, n_nodes, g, d_counts, (const uint32_t*)d_off, d_ent, d_flag);
    hipLaunchKernelGGL(bins_sort_kernel, dim3((unsigned)((n_bins + 3) / 4)), dim3(256), 0, stream, (const uint32_t*)d_off, d_ent, n_bins);       // a wave per bin
    RTW_HIP_OK(hipStreamSynchronize(stream));
    RTW_HIP_OK(hipGetLastError());
    *off_out = d_off; *ent_out = d_ent; *has_bins = 1;
    return 0;
}
#undef RTW_HIP_OK

#ifdef RTW_TIMING
int read_timing(unsigned long long* out, int n) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rtw_timing), sizeof(unsigned long long) * (size_t)n); }
#endif

int launch_closest(const RtwSceneDev* sc, const float* rays, long long n, float* hits11, int* shape, int* tri, bool stats, hipStream_t stream)
{
    if (n <= 0) return 0;
    const int block = 256;
    const unsigned grid = (unsigned)((n + block - 1) / block);
    if (stats) hipLaunchKernelGGL(closest_kernel<true>, dim3(grid), dim3(block), 0, stream, sc, rays, n, hits11, shape, tri);
    else hipLaunchKernelGGL(closest_kernel<false>, dim3(grid), dim3(block), 0, stream, sc, rays, n, hits11, shape, tri);
    return (int)hipGetLastError();
}

int launch_ray_trace(const RtwSceneDev* sc, const float* rays, const uint32_t* keys2, long long n, int max_bounce, int preview,
                     uint32_t seed, unsigned long long npix, float* rgb, void* ws, bool stats, hipStream_t stream)
{
    if (n <= 0) return 0;
    const int block = 256;
    const unsigned grid = (unsigned)((n + block - 1) / block);
    if (stats) hipLaunchKernelGGL(ray_trace_kernel<true>, dim3(grid), dim3(block), 0, stream, sc, rays, keys2, n, max_bounce, preview, seed, npix, rgb, (float4*)ws);
    else hipLaunchKernelGGL(ray_trace_kernel<false>, dim3(grid), dim3(block), 0, stream, sc, rays, keys2, n, max_bounce, preview, seed, npix, rgb, (float4*)ws);
    return (int)hipGetLastError();
}

int launch_texture_sample(const RtwSceneDev* sc, int shape, int mat, const float* uv, long long n, float* rgba, hipStream_t stream)
{
    if (n <= 0) return 0;
    const int block = 256;
    const unsigned grid = (unsigned)((n + block - 1) / block);
    hipLaunchKernelGGL(texture_sample_kernel, dim3(grid), dim3(block), 0, stream, sc, shape, mat, uv, n, rgba);
    return (int)hipGetLastError();
}

// ---- multi-GPU gather: pack / unpack of the ranks' compact row blocks (gather_rows_kernel) --------------------------------------
size_t gather_block_bytes(size_t pixels, bool with_accum)
{
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    return up(pixels * 4) + (with_accum ? up(pixels * 16) : 0);
}

int launch_gather_rows(bool unpack, void* argb, void* accum, void* stage, int width, int task_rows, int world, const GatherBlocks& b, bool with_accum, hipStream_t stream)
{
    if (b.n <= 0) return 0;
    GatherPlan g;
    g.argb = (uint32_t*)argb; g.accum = (float4*)accum; g.stage = (char*)stage;
    g.task_px = task_rows * width; g.world = world; g.first_rank = b.first_rank; g.with_accum = with_accum ? 1 : 0;
    uint32_t most = 0;
    for (int k = 0; k < 64; k++) { g.block_off[k] = k < b.n ? b.off[k] : 0; g.block_px[k] = k < b.n ? b.px[k] : 0; if (g.block_px[k] > most) most = g.block_px[k]; }
    if (most == 0) return 0;
    const bool vec = width % 4 == 0;
    unsigned gx = (unsigned)((most / (vec ? 4u : 1u) + 255u) / 256u);
    if (gx < 1u) gx = 1u;
    if (gx > 4096u) gx = 4096u;
    const dim3 grid(gx, (unsigned)b.n);
    if (unpack) { if (vec) hipLaunchKernelGGL((gather_rows_kernel<true, true>), grid, dim3(256), 0, stream, g); else hipLaunchKernelGGL((gather_rows_kernel<true, false>), grid, dim3(256), 0, stream, g); }
    else { if (vec) hipLaunchKernelGGL((gather_rows_kernel<false, true>), grid, dim3(256), 0, stream, g); else hipLaunchKernelGGL((gather_rows_kernel<false, false>), grid, dim3(256), 0, stream, g); }
    return (int)hipGetLastError();
}

}  // namespace rtw
