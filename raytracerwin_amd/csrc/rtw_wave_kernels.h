// rtw_wave_kernels.h -- the "bins + wave" pipeline (pipeline 3).  Included inside rtw_device.hip's anonymous
// namespace (device build only); it uses that file's ray / triangle / shading helpers unchanged.
//
//   primary_bins_kernel  one thread per pixel, one wave per screen tile.  The camera is the reference's fixed one
//                        (Src/RayTracerProgram.cpp:133-165), so the leaves a tile's camera rays can meet are known
//                        before the frame starts (RtwBinsDev, built on the host): the wave runs through that short
//                        list -- wave-uniform index, records through scalar loads, nothing pointer-chased -- and each
//                        lane gives its own ray the reference's box test and triangle test, in preorder, with its
//                        own shrinking segment.  Misses are finished here, hits are queued with their hit record.
//   trace_wave_kernel    every secondary segment is traced by a WHOLE wave, one ray at a time: 64 lanes test 64 boxes of the flat
//                        hierarchy per step (no stack, wave-uniform control flow, no divergence), candidate leaves come out in preorder
//                        and are triangle-tested 64 at a time with the reference's shrinking segment.
//
// Why the results are the reference's bits: KdNode::TestRayIntersection (Src/KdTree.cpp:128-195) tests exactly the
// leaves whose own box the ray's line meets, in preorder, each with the segment left by the previous accepted hit.
// Both kernels meet a superset-filtered list of leaves in preorder, apply the reference's own box test to the leaf's
// own box and the reference's triangle test with the running segment; every float operation is the one the other
// pipelines execute (same helpers).

#define RTW_WAVE_LDS_WORDS 256      // per wave: 64 level-2 hits, 64 level-1 hits, 128 candidate leaves
#ifndef RTW_TRACEWAVE_MINW
#define RTW_TRACEWAVE_MINW 1     // (8 = 64 VGPRs with a small spill measured slower than the natural 75 VGPRs / 6 waves per SIMD)
#endif

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }
__device__ __forceinline__ int mbcnt(unsigned long long m)      // set bits of m below this lane
{
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
__device__ __forceinline__ float readlane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
// LDS words written by some lanes of the wave are read by others: nothing to wait for in hardware (a wave's LDS
// operations execute in order), the fences only keep the compiler from moving the accesses across each other
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct FlatRay { f3 o; float ix, iy, iz, eps_t; bool skx, sky, skz; };   // sk*: axis skipped by the reference's test (|d| < FLT_EPSILON)

// reference box test of a tame ray (RRay::TestIntersectionWithAabb, Src/RRay.cpp:89-136) on entry idx of a flat
// level, plus the conservative segment clip when `prune`
// EXACT: the ray is not "tame" (a direction component below FLT_EPSILON, NaN, ...): the reference's test as written (skipped axes,
// Math::Min/Max), no clip.
typedef float rtw_v2f __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) const rtw_v2f rtw_g_v2;
template <bool EXACT>
__device__ __forceinline__ bool flat_box_hit(const float* __restrict__ b, int idx, const FlatRay& fr, bool prune, float far_t)
{
    // the entry's three (min, max) pairs; (pair - origin) * reciprocal is one packed subtract + one packed multiply per axis,
    // component-wise the reference's float operations
    rtw_v2f bx, by, bz;
    { const rtw_g_v2* e = (rtw_g_v2*)b + (size_t)idx * 3; bx = e[0]; by = e[1]; bz = e[2]; }
    const float x1 = (bx.x - fr.o.x) * fr.ix, x2 = (bx.y - fr.o.x) * fr.ix;
    const float y1 = (by.x - fr.o.y) * fr.iy, y2 = (by.y - fr.o.y) * fr.iy;
    const float z1 = (bz.x - fr.o.z) * fr.iz, z2 = (bz.y - fr.o.z) * fr.iz;
    if (EXACT) {
        float tmin = -FLT_MAX, tmax = FLT_MAX;
        if (!fr.skx) { tmin = ref_max(tmin, ref_min(x1, x2)); tmax = ref_min(tmax, ref_max(x1, x2)); }
        if (!fr.sky) { tmin = ref_max(tmin, ref_min(y1, y2)); tmax = ref_min(tmax, ref_max(y1, y2)); }
        if (!fr.skz) { tmin = ref_max(tmin, ref_min(z1, z2)); tmax = ref_min(tmax, ref_max(z1, z2)); }
        return tmax > tmin;
    }
    const float tmin = fmaxf(fmaxf(fminf(x1, x2), fminf(y1, y2)), fminf(z1, z2));
    const float tmax = fminf(fminf(fmaxf(x1, x2), fmaxf(y1, y2)), fmaxf(z1, z2));
    bool h = tmax > tmin;
    if (prune) h = h && !(tmin > far_t) && !(tmax < -fr.eps_t);
    return h;
}

// a shape's flat hierarchy, copied out of the scene descriptor once (no reloads in the walk)
struct FlatSrc {
    const float* lvl[3]; const float4* tris;
    int n[3], pad[3];           // entries and array stride of each level (copied out of the shape once: no reloads in the walk)
    float bmin[3], bmax[3];     // the shape's bound (RShape::Aabb)
};
__device__ __forceinline__ FlatSrc flat_src_of(const RtwShapeDev& sh)
{
    FlatSrc g;
    for (int l = 0; l < 3; l++) { g.lvl[l] = sh.flat[l]; g.n[l] = sh.flat_n[l]; g.pad[l] = sh.flat_pad[l]; }
    g.tris = reinterpret_cast<const float4*>(sh.tris);
    for (int k = 0; k < 3; k++) { g.bmin[k] = sh.bmin[k]; g.bmax[k] = sh.bmax[k]; }
    return g;
}

// triangle tests of the n candidate leaves in lds_c[0..n) (ascending = preorder), 64 at a time.  A test must see
// the segment left by every earlier accepted hit, so after an accept the later lanes are tested again with the
// shortened segment (the reference tests them one after another).
template <bool STATS>
__device__ __forceinline__ void wave_triangles(const float4* __restrict__ tr4, const uint32_t* __restrict__ lds_c, int n, const Ray& r,
                                               float& cur_dist, f3& hit_pos, int& hit_slot, bool& any, Counters& ct)
{
    const int lane = lane_id();
    for (int j = 0; j < n; j += 64) {
        const bool mine = j + lane < n;
        const int leaf = mine ? (int)lldu(lds_c, j + lane) : 0;
        float4 a, b, c, d;
        a = gld4(tr4, 4 * (size_t)leaf); b = gld4(tr4, 4 * (size_t)leaf + 1); c = gld4(tr4, 4 * (size_t)leaf + 2); d = gld4(tr4, 4 * (size_t)leaf + 3);
        if (STATS) ct.tris += mine ? 1u : 0u;
        int settled = -1;
        for (;;) {
            f3 cp = mk(0, 0, 0); float dist = 0.0f;
            const bool acc = mine && lane > settled && triangle_test(r, cur_dist, a, b, c, d.x, cp, dist);
            const unsigned long long am = __ballot(acc);
            if (am == 0ull) break;
            const int first = __ffsll((long long)am) - 1;
            cur_dist = readlane_f(dist, first);
            hit_pos = mk(readlane_f(cp.x, first), readlane_f(cp.y, first), readlane_f(cp.z, first));
            hit_slot = __builtin_amdgcn_readlane(leaf, first);
            any = true;
            settled = first;
        }
    }
}

// KdTree::TestRayIntersection for ONE tame ray held identically by all 64 lanes.  Depth-first over the three flat
// levels, four entries (64 children) per step, so the leaves come out in ascending slot order = preorder.
template <bool STATS, bool EXACT>
__device__ __forceinline__ bool wave_walk_flat(const FlatSrc& src, uint32_t* __restrict__ lds, const Ray& r, const FlatRay& fr, bool prune,
                                               float& cur_dist, f3& hit_pos, int& hit_slot, Counters& ct)
{
    uint32_t* l2 = lds; uint32_t* l1 = lds + 64; uint32_t* lc = lds + 128;
    const int lane = lane_id();
    const int n0 = src.n[0], n1 = src.n[1], n2 = src.n[2];
    bool any = false;
    int ncand = 0;
    for (int t0 = 0; t0 < n2; t0 += 64) {
        float far_t = cur_dist + (fr.eps_t + 1.0e-4f * cur_dist);
        const int i2 = t0 + lane;
        const bool v2 = i2 < n2;
        const bool h2 = v2 && flat_box_hit<EXACT>(src.lvl[2], i2, fr, prune, far_t);
        if (STATS) ct.boxes += v2 ? 1u : 0u;
        const unsigned long long m2 = __ballot(h2);
        if (m2 == 0ull) continue;
        const int c2 = __popcll(m2) * 16;
        wave_lds_sync();                            // every lane has read its l2 word of the previous round
        if (h2) lstu(l2, mbcnt(m2), (uint32_t)i2);
        wave_lds_sync();
        for (int g2 = 0; g2 < c2; g2 += 64) {
            const int g = g2 + lane;
            const int i1 = (g < c2 ? (int)lldu(l2, g >> 4) : 0) * 16 + (g & 15);
            const bool v1 = g < c2 && i1 < n1;
            const bool h1 = v1 && flat_box_hit<EXACT>(src.lvl[1], i1, fr, prune, far_t);
            if (STATS) ct.boxes += v1 ? 1u : 0u;
            const unsigned long long m1 = __ballot(h1);
            if (m1 == 0ull) continue;
            const int c1 = __popcll(m1) * 16;
            wave_lds_sync();                        // every lane has read its l1 word of the previous round
            if (h1) lstu(l1, mbcnt(m1), (uint32_t)i1);
            wave_lds_sync();
            for (int g1 = 0; g1 < c1; g1 += 64) {
                const int gg = g1 + lane;
                const int i0 = (gg < c1 ? (int)lldu(l1, gg >> 4) : 0) * 16 + (gg & 15);
                const bool v0 = gg < c1 && i0 < n0;
                const bool h0 = v0 && flat_box_hit<EXACT>(src.lvl[0], i0, fr, prune, far_t);
                if (STATS) ct.boxes += v0 ? 1u : 0u;
                const unsigned long long m0 = __ballot(h0);
                if (m0 == 0ull) continue;
                if (h0) lstu(lc, ncand + mbcnt(m0), (uint32_t)i0);
                ncand += __popcll(m0);
                if (ncand > 64) {                   // the list holds 128: make room before the next 64
                    wave_lds_sync();
                    wave_triangles<STATS>(src.tris, lc, ncand, r, cur_dist, hit_pos, hit_slot, any, ct);
                    wave_lds_sync();
                    ncand = 0;
                    far_t = cur_dist + (fr.eps_t + 1.0e-4f * cur_dist);
                }
            }
        }
    }
    if (ncand > 0) {
        wave_lds_sync();
        wave_triangles<STATS>(src.tris, lc, ncand, r, cur_dist, hit_pos, hit_slot, any, ct);
        wave_lds_sync();
    }
    return any;
}

// FindIntersectionWithScene (Src/RayTracerScene.cpp:99-125) of one ray held by the whole wave, without the shading
// tail (only the record of the last shape that hit is read afterwards).  `shape0` describes shape 0; later shapes are read from the scene.
// The query covers shapes [first_shape, n_shapes) and continues from the caller's (hit_shape, hit_slot, hit_pos, seg): for a fresh
// query that is (-1, -1, 0, ray.dist) with first_shape 0; with leading analytic shapes already tested by the shading lane
// (RtwRenderParams::lead_shapes) it is that partial result.
// AN = the scene may hold spheres / planes / capsules; mesh-only scenes run instantiations without that code (registers, occupancy).
template <bool STATS, bool AN>
__device__ __forceinline__ void wave_find_intersection(const RtwSceneDev* __restrict__ sc, int first_shape, int n_shapes, bool prune, const FlatSrc& shape0,
                                                       uint32_t* __restrict__ lds, const Ray& ray,
                                                       int& hit_shape, int& hit_slot, f3& hit_pos, float& seg, Counters& ct)
{
    const bool one = lane_id() == 0;
    if (STATS && one) ct.rays++;
    const bool tame = ray_is_tame(ray);
    FlatRay fr;             // the ray's reciprocals (0 on an axis the reference's test skips), shared by the bound test and the walk
    fr.o = ray.o;
    fr.skx = near_zero(ray.d.x); fr.sky = near_zero(ray.d.y); fr.skz = near_zero(ray.d.z);
    fr.ix = (!tame && fr.skx) ? 0.0f : 1.0f / ray.d.x; fr.iy = (!tame && fr.sky) ? 0.0f : 1.0f / ray.d.y; fr.iz = (!tame && fr.skz) ? 0.0f : 1.0f / ray.d.z;
    fr.eps_t = 2.0e-5f * fmaxf(fabsf(fr.ix), fmaxf(fabsf(fr.iy), fabsf(fr.iz)));
    for (int s = first_shape; s < n_shapes; s++) {
        const FlatSrc g = (s == 0) ? shape0 : flat_src_of(sc->shapes[s]);
        const int kind = AN ? sc->shapes[s].kind : RTW_SHAPE_MESH;
        if (AN && kind == RTW_SHAPE_PLANE) {      // no culling box (RPlane::HasCullingBounds); every lane computes the one ray's test alike
            f3 pos; float dist; int part;
            if (analytic_test(sc->shapes[s], ray, seg, pos, dist, part)) { seg = dist; hit_shape = s; hit_slot = part; hit_pos = pos; }
            continue;
        }
        if (STATS && one) ct.boxes++;
        bool in_bound;
        if (tame) {         // RRay::TestIntersectionWithAabb with no axis skipped and no NaN: same operations, reciprocals reused
            const float x1 = (g.bmin[0] - ray.o.x) * fr.ix, x2 = (g.bmax[0] - ray.o.x) * fr.ix;
            const float y1 = (g.bmin[1] - ray.o.y) * fr.iy, y2 = (g.bmax[1] - ray.o.y) * fr.iy;
            const float z1 = (g.bmin[2] - ray.o.z) * fr.iz, z2 = (g.bmax[2] - ray.o.z) * fr.iz;
            const float tmin = fmaxf(fmaxf(fminf(x1, x2), fminf(y1, y2)), fminf(z1, z2));
            const float tmax = fminf(fminf(fmaxf(x1, x2), fmaxf(y1, y2)), fmaxf(z1, z2));
            in_bound = tmax > tmin;
        } else {
            float t0, t1;
            in_bound = slab_exact(ray, g.bmin[0], g.bmin[1], g.bmin[2], g.bmax[0], g.bmax[1], g.bmax[2], t0, t1);
        }
        if (!in_bound) continue;
        float cur = seg; f3 pos = mk(0, 0, 0); int slot = -1;
        bool any;
        if (AN && kind != RTW_SHAPE_MESH) { // a sphere or a capsule: the record's slot says which part was hit
            any = analytic_test(sc->shapes[s], ray, seg, pos, cur, slot);
        } else if (g.n[0] > 0) {
            if (tame) {
                any = wave_walk_flat<STATS, false>(g, lds, ray, fr, prune, cur, pos, slot, ct);
            } else {                // rare (a direction component below FLT_EPSILON, NaN, ...): same walk, the reference's test as written
                any = wave_walk_flat<STATS, true>(g, lds, ray, fr, false, cur, pos, slot, ct);
            }
        } else {                    // a shape without a flat hierarchy: the reference's own walk, all lanes alike
            const RtwShapeDev& sh = sc->shapes[s];
            Counters walk = { 0, 0, 0, 0, 0, 0 };
            if (tame) any = tree_walk<true, STATS>(sc, sh.nodes, sh.tris, sh.n_nodes, sh.n_tris, ray, prune, cur, pos, slot, walk);
            else any = tree_walk<false, STATS>(sc, sh.nodes, sh.tris, sh.n_nodes, sh.n_tris, ray, false, cur, pos, slot, walk);
            if (STATS && one) { ct.boxes += walk.boxes; ct.tris += walk.tris; }
        }
        if (any) { seg = cur; hit_shape = s; hit_slot = slot; hit_pos = pos; }
    }
}

// ---- tiles no leaf can be met from: sky only -----------------------------------------------------------------------------
// The jobs [job0, n_jobs) of the job table are tiles whose bins are empty for every shape: each sample sees the sky
// (Src/RayTracerScene.cpp:89-94).  A kernel of its own, launched beside primary_bins_kernel on a second stream: it needs a
// fraction of that kernel's registers (no spills of scalar registers into vector lanes) and nothing later in the pass waits for it.
__global__ __launch_bounds__(256) void primary_sky_kernel(const float* __restrict__ gamma_thr, float4* __restrict__ accum, uint32_t* __restrict__ argb,
                                                          RtwRenderParams p, int job0)
{
    __shared__ float thr[256];
    thr[threadIdx.x] = gamma_thr[threadIdx.x];
    __syncthreads();
    const int npix = p.width * p.height;
    const uint32_t phase = table_phase(p.seed);
    const int wave0 = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), nwaves = (int)(gridDim.x * (blockDim.x >> 6));
    for (int wk = job0 + wave0; wk < p.n_jobs; wk += nwaves) {
        const int wt = (int)(cldu(p.tile_order, wk) & 0xFFFFFFu);
        const int wi = wt * 64 + lane_id();
        int px = 0, py = 0;
        const bool live = work_to_xy(p, wi, px, py);
        if (!live) continue;
        const int pixel = py * p.width + px;
        float4 acc_prev = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!p.preview) acc_prev = accum[pixel];
        f3 csum = mk(0, 0, 0);
        for (int i = 0; i < p.sub_samples; i++) {
            PathRng rng; rng_init(rng, p.seed, phase, (uint64_t)npix, (uint32_t)pixel, (uint32_t)p.pass_index, (uint32_t)i);
            const Ray ray = camera_ray_xy(p, px, py, i, rng);
            const f3 si = p.max_bounce != 0 ? sky_color(ray.d.y) : mk(0, 0, 0);       // RayTrace(.., 0) is black (Src/RayTracerScene.cpp:39)
            csum = csum + si;
        }
        const f3 c = p.sub_samples == 1 ? csum : csum / (float)p.sub_samples;
        if (p.preview) {
            argb[pixel] = pack_pixel(thr, c);
        } else {
            const f3 sum = mk(acc_prev.x, acc_prev.y, acc_prev.z) + c;
            const int n = __float_as_int(acc_prev.w) + 1;
            accum[pixel] = make_float4(sum.x, sum.y, sum.z, __int_as_float(n));
            argb[pixel] = pack_pixel(thr, n == 1 ? sum : sum / (float)n);
        }
    }
}

// ---- primary rays through the screen bins ------------------------------------------------------------------------
template <bool STATS, bool AN>
__global__ __launch_bounds__(256) void primary_bins_kernel(const RtwSceneDev* __restrict__ sc, float4* __restrict__ accum,
                                                           uint32_t* __restrict__ argb, PipeBufs pb, RtwRenderParams p)
{
    __shared__ float thr[256];
    thr[threadIdx.x] = sc->gamma_thr[threadIdx.x];
    __syncthreads();
    const int npix = p.width * p.height;
    Counters ct = { 0, 0, 0, 0, 0, 0 };
    const uint32_t phase = table_phase(p.seed);
    const int n_shapes = sc->n_shapes;
    const bool prune = sc->prune != 0;
    // persistent waves: the grid holds a few blocks per CU and every wave takes tiles in turn, so that the gamma table
    // above and the launch of a wave are paid once per many tiles, not once per 64 pixels
    const int n_jobs = p.tile_order ? p.n_jobs : (p.count >> 6);
    const int wave0 = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), nwaves = (int)(gridDim.x * (blockDim.x >> 6));
  for (int wk = wave0; wk < n_jobs; wk += nwaves) {
    // Jobs in the order of the table: tiles with the longest bin lists first (their waves take longest; the many empty tiles
    // then fill the tail).  A job is a tile (low 24 bits); with several sub-samples a long-listed tile is one job PER
    // sub-sample (bits 24..27 = sub-sample + 1): four waves share its list walk, all its pixels are resolved by
    // resolve_kernel from the kept sample colours.
    const uint32_t job = p.tile_order ? cldu(p.tile_order, wk) : (uint32_t)wk;
    const int wt = (int)(job & 0xFFFFFFu);
    const int only_sample = (int)((job >> 24) & 15u) - 1;    // -1: this wave does every sub-sample of its tile
    const int wi = wt * 64 + lane_id();
    int px = 0, py = 0;
    const bool live = work_to_xy(p, wi, px, py);
    if (!live) { px = 0; py = 0; }
    const int pixel = live ? py * p.width + px : npix;
    // the accumulator entry is needed at the very end: ask for it now
    float4 acc_prev = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live && !p.preview) acc_prev = accum[pixel];
    // the wave's bin: all its pixels lie in one tile of the screen's bin grid (tile rows never straddle a bin row)
    const unsigned long long live_mask = __ballot(live);
    int bin = 0;
    if (live_mask != 0ull) {
        const int first = __ffsll((long long)live_mask) - 1;
        const int fy = __builtin_amdgcn_readlane(py, first), fx = __builtin_amdgcn_readlane(px, first);
        bin = (fy >> (6 - p.tile_shift)) * p.tiles_per_row + (fx >> p.tile_shift);      // tile_w = 1 << tile_shift, tile_h = 64 / tile_w
    }
    // Can any sample of this wave's pixels hit anything?  Only if a shape has leaves in the tile's bin (or has no bins).
    // Then every sample's colour is kept (resolve_kernel sums a pending pixel's samples in order); else nothing is kept.
    bool near_wave = false;
    for (int k = 0; k < n_shapes; k++) {
        const uint32_t* __restrict__ boff = p.bins[k].off;
        near_wave = near_wave || boff == nullptr || cldu(boff, bin) != cldu(boff, bin + 1);     // (a sphere / plane / capsule has no bins)
    }
    f3 csum = mk(0, 0, 0);                                   // s[0] + s[1] + ... in sample order, as the reference adds them
    uint32_t queued = 0u;
    bool hit_any = false;                                    // some sample of this pixel hit something: the pixel is resolved later
    for (int i = 0; i < p.sub_samples; i++) {                // wave-uniform loop
        if (only_sample >= 0 && i != only_sample) continue;
        PathRng rng; rng_init(rng, p.seed, phase, (uint64_t)npix, (uint32_t)(live ? pixel : 0), (uint32_t)p.pass_index, (uint32_t)i);
        const Ray ray = camera_ray_xy(p, px, py, i, rng);
        if (STATS && live) ct.cams++;
        f3 si = mk(0, 0, 0);
        bool queue_it = false;
        float4 hr0 = make_float4(0.f, 0.f, 0.f, 0.f), hr1 = hr0;
        if (p.max_bounce != 0 && !near_wave && !STATS) {     // no leaf of any shape can be met from this tile: every sample sees the sky
            if (live) si = sky_color(ray.d.y);
        } else if (p.max_bounce != 0) {                      // RayTrace(.., 0) is black (Src/RayTracerScene.cpp:39)
            // FindIntersectionWithScene of the camera ray (Src/RayTracerScene.cpp:99-125), shapes in insertion order
            int hit_shape = -1, hit_slot = -1;
            f3 hit_pos = mk(0, 0, 0);
            float seg = ray.dist;
            if (STATS && live) ct.rays++;
            const bool tame = ray_is_tame(ray);
            // a handful of rays per frame are not tame (a direction component below FLT_EPSILON at the image's centre column / row):
            // they go through the same lists with the reference's test as written (skipped axes); only such waves pay for it
            const bool any_untame = __ballot(live && !tame) != 0ull;
            const bool skx = near_zero(ray.d.x), sky = near_zero(ray.d.y), skz = near_zero(ray.d.z);
            for (int k = 0; k < n_shapes; k++) {
                const RtwShapeDev& sh = sc->shapes[k];
                const int kind = AN ? sh.kind : RTW_SHAPE_MESH;
                float t0, t1;
                const bool inbox = live && ((AN && kind == RTW_SHAPE_PLANE) ||      // a plane has no culling box (RPlane::HasCullingBounds)
                                            slab_exact(ray, sh.bmin[0], sh.bmin[1], sh.bmin[2], sh.bmax[0], sh.bmax[1], sh.bmax[2], t0, t1));
                if (STATS && live && kind != RTW_SHAPE_PLANE) ct.boxes++;
                if (__ballot(inbox) == 0ull) continue;
                float cur = seg; f3 pos = mk(0, 0, 0); int slot = -1;
                bool any = false;
                const uint32_t* __restrict__ boff = p.bins[k].off;
                if (AN && kind != RTW_SHAPE_MESH) {          // a sphere / plane / capsule: every lane tests its own ray; slot = the part hit
                    if (inbox) any = analytic_test(sh, ray, seg, pos, cur, slot);
                } else if (boff == nullptr) {                // no bins for this shape: packet walk of its tree
                    any = packet_walk<STATS, false>(sh.nodes, sh.tris, sh.n_nodes, ray, inbox && tame, prune, cur, pos, slot, ct);
                    if (any_untame) {
                        const bool a2 = packet_walk<STATS, true>(sh.nodes, sh.tris, sh.n_nodes, ray, inbox && !tame, false, cur, pos, slot, ct);
                        any = any || a2;
                    }
                } else {
                    const bool active = inbox && tame;
                    const bool active_exact = inbox && !tame;
                    const uint32_t* __restrict__ bent = p.bins[k].ent;
                    const int e0 = (int)cldu(boff, bin), e1 = (int)cldu(boff, bin + 1);
                    const float4* nd4 = reinterpret_cast<const float4*>(sh.nodes);
                    const float4* tr4 = reinterpret_cast<const float4*>(sh.tris);
                    const float ix = (!tame && skx) ? 0.0f : 1.0f / ray.d.x, iy = (!tame && sky) ? 0.0f : 1.0f / ray.d.y, iz = (!tame && skz) ? 0.0f : 1.0f / ray.d.z;
                    const float eps_t = 2.0e-5f * fmaxf(fabsf(ix), fmaxf(fabsf(iy), fabsf(iz)));
                    const int lane = lane_id();
                    // 64 entries at a time: lane j fetches entry j's leaf box and triangle record (all loads in flight together),
                    // then the wave goes through the entries in order and every lane tests its own ray against the broadcast record
                    for (int ec = e0; ec < e1; ec += 64) {
                        const int cnt = e1 - ec < 64 ? e1 - ec : 64;
                        const int mnode = lane < cnt ? (int)bent[ec + lane] : 0;
                        const float4 mlo = gld4(nd4, 2 * (size_t)mnode), mhi = gld4(nd4, 2 * (size_t)mnode + 1);
                        const int mleaf = __float_as_int(mhi.w) < 0 ? 0 : __float_as_int(mhi.w);
                        const float4 ta = gld4(tr4, 4 * (size_t)mleaf), tb = gld4(tr4, 4 * (size_t)mleaf + 1), tc = gld4(tr4, 4 * (size_t)mleaf + 2);
                        const float td = gld4(tr4, 4 * (size_t)mleaf + 3).x;
                        for (int j = 0; j < cnt; j++) {
                            const float lox = readlane_f(mlo.x, j), loy = readlane_f(mlo.y, j), loz = readlane_f(mlo.z, j);
                            const float hix = readlane_f(mhi.x, j), hiy = readlane_f(mhi.y, j), hiz = readlane_f(mhi.z, j);
                            const float x1 = (lox - ray.o.x) * ix, x2 = (hix - ray.o.x) * ix;
                            const float y1 = (loy - ray.o.y) * iy, y2 = (hiy - ray.o.y) * iy;
                            const float z1 = (loz - ray.o.z) * iz, z2 = (hiz - ray.o.z) * iz;
                            const float tmin = fmaxf(fmaxf(fminf(x1, x2), fminf(y1, y2)), fminf(z1, z2));
                            const float tmax = fminf(fminf(fmaxf(x1, x2), fmaxf(y1, y2)), fmaxf(z1, z2));
                            bool hit = active && (tmax > tmin);
                            if (prune) hit = hit && !(tmin > cur + (eps_t + 1.0e-4f * cur)) && !(tmax < -eps_t);
                            if (any_untame) {                // wave-uniform: RRay::TestIntersectionWithAabb as written for the lanes that need it
                                float emin = -FLT_MAX, emax = FLT_MAX;
                                if (!skx) { emin = ref_max(emin, ref_min(x1, x2)); emax = ref_min(emax, ref_max(x1, x2)); }
                                if (!sky) { emin = ref_max(emin, ref_min(y1, y2)); emax = ref_min(emax, ref_max(y1, y2)); }
                                if (!skz) { emin = ref_max(emin, ref_min(z1, z2)); emax = ref_min(emax, ref_max(z1, z2)); }
                                if (active_exact) hit = emax > emin;
                            }
                            if (STATS) ct.boxes += (active || active_exact) ? 1u : 0u;
                            if (__ballot(hit) == 0ull) continue;
                            const int leaf = __builtin_amdgcn_readlane(mleaf, j);
                            const float4 a = make_float4(readlane_f(ta.x, j), readlane_f(ta.y, j), readlane_f(ta.z, j), readlane_f(ta.w, j));
                            const float4 b = make_float4(readlane_f(tb.x, j), readlane_f(tb.y, j), readlane_f(tb.z, j), readlane_f(tb.w, j));
                            const float4 c = make_float4(readlane_f(tc.x, j), readlane_f(tc.y, j), readlane_f(tc.z, j), readlane_f(tc.w, j));
                            const float d1 = readlane_f(td, j);
                            if (hit) {
                                if (STATS) ct.tris++;
                                f3 cp; float dist;
                                if (triangle_test(ray, cur, a, b, c, d1, cp, dist)) { cur = dist; pos = cp; slot = leaf; any = true; }
                            }
                        }
                    }
                }
                if (any) { seg = cur; hit_shape = k; hit_slot = slot; hit_pos = pos; }
            }
            if (live) {
                if (hit_shape < 0) {
                    si = sky_color(ray.d.y);
                } else {
                    queue_it = true;
                    hr0 = make_float4(hit_pos.x, hit_pos.y, hit_pos.z, seg);
                    hr1 = make_float4(__int_as_float(hit_shape), __int_as_float(hit_slot), 0.0f, 0.0f);
                }
            }
        }
        csum = csum + si;
        if (live && (near_wave || only_sample >= 0) && !queue_it) pb.rad[(size_t)wi * 4 + i] = make_float4(si.x, si.y, si.z, 0.0f);
        if (queue_it) {                                      // shade the hit here: the path's slot needs no queue position
            hit_any = true;
            if (shade_hit_step<STATS, AN>(sc, pb, p, slot_of_path(p, (uint32_t)wi, (uint32_t)i), (uint32_t)wi * 4u + (uint32_t)i, ray, rng, p.max_bounce, 0, hr0, hr1, ct))
                queued |= 1u << i;                           // it goes on: its slot joins round 0's trace list below
        }
    }
    // ONE atomic per wave for both lists (the two counters are one 64-bit word): the queue gets an entry per queued sample,
    // the pending list one per pixel with a queued sample
    // a tile split by sub-sample: every pixel is pending (listed once, by the wave of sub-sample 0)
    const bool pending = only_sample >= 0 ? live : hit_any;
    const bool list_pixel = only_sample >= 0 ? (live && only_sample == 0) : pending;
    {
        const unsigned long long m0 = __ballot((queued & 1u) != 0u), m1 = __ballot((queued & 2u) != 0u), m2 = __ballot((queued & 4u) != 0u),
                                 m3 = __ballot((queued & 8u) != 0u), mp = __ballot(list_pixel);
        if ((mp | m0 | m1 | m2 | m3) != 0ull) {
            const uint32_t c0 = (uint32_t)__popcll(m0), c1 = (uint32_t)__popcll(m1), c2 = (uint32_t)__popcll(m2), c3 = (uint32_t)__popcll(m3);
            unsigned long long base = 0ull;
            if (lane_id() == 0)
                base = atomicAdd(reinterpret_cast<unsigned long long*>(pb.counters), (unsigned long long)(c0 + c1 + c2 + c3) | ((unsigned long long)__popcll(mp) << 32));
            const uint32_t qb = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);
            const uint32_t pbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32));
            const uint32_t e0 = slot_of_path(p, (uint32_t)wi, 0u);
            const uint32_t es = 1u;
            if (queued & 1u) pb.queue[qb + (uint32_t)mbcnt(m0)] = e0;
            if (queued & 2u) pb.queue[qb + c0 + (uint32_t)mbcnt(m1)] = e0 + es;
            if (queued & 4u) pb.queue[qb + c0 + c1 + (uint32_t)mbcnt(m2)] = e0 + 2u * es;
            if (queued & 8u) pb.queue[qb + c0 + c1 + c2 + (uint32_t)mbcnt(m3)] = e0 + 3u * es;
            if (list_pixel) pb.pend[pbase + (uint32_t)mbcnt(mp)] = (uint32_t)wi;
        }
    }
    if (live && !pending) {
        const f3 c = p.sub_samples == 1 ? csum : csum / (float)p.sub_samples;     // x / 1.0f == x
        if (p.preview) {
            argb[pixel] = pack_pixel(thr, c);
        } else {                                             // AccumulatePixel::AddPixel + GetGammaSpacePixel (resolve_pixel, with the entry loaded above)
            const f3 sum = mk(acc_prev.x, acc_prev.y, acc_prev.z) + c;
            const int n = __float_as_int(acc_prev.w) + 1;
            accum[pixel] = make_float4(sum.x, sum.y, sum.z, __int_as_float(n));
            argb[pixel] = pack_pixel(thr, n == 1 ? sum : sum / (float)n);
        }
    }
  }
    if (STATS) flush_counters(sc, ct);
}

// ---- one round of secondary segments: a wave per ray -----------------------------------------------------------------
// The ray comes in through scalar loads, the walk is wave_walk_flat; a wave takes rays in turn.
template <bool STATS, int NT, bool AN>
__global__ __launch_bounds__(NT) void trace_wave_kernel(const RtwSceneDev* __restrict__ sc, PipeBufs pb, RtwRenderParams p, int round)
{
    HIP_DYNAMIC_SHARED(uint32_t, wave_dyn);              // [NT / 64 waves x RTW_WAVE_LDS_WORDS]
    const bool from_queue = round == 0;                  // the path queue IS round 0's trace list
    const uint32_t n = from_queue ? pb.counters[0] : pb.counters[4 + round];
    if ((uint32_t)blockIdx.x * (uint32_t)(NT / 64) >= n) return;       // (whole block) no ray for this block's first wave
    uint32_t* lds = wave_dyn + (threadIdx.x >> 6) * RTW_WAVE_LDS_WORDS;
    const FlatSrc shape0 = flat_src_of(sc->shapes[0]);
    const uint32_t* __restrict__ src = from_queue ? pb.queue : wf_list(pb, round & 1);
    const int n_shapes = sc->n_shapes;
    const bool prune = sc->prune != 0;
    const uint32_t wave = (blockIdx.x * (uint32_t)NT + threadIdx.x) >> 6, nwaves = gridDim.x * (uint32_t)(NT / 64);
    Counters ct = { 0, 0, 0, 0, 0, 0 };
    for (uint32_t k = wave; k < n; k += nwaves) {
        const int ku = __builtin_amdgcn_readfirstlane((int)k);
        const int q = (int)cldu(src, ku);
        const float4 s0 = cld4(pb.state, q * 3), s1 = cld4(pb.state, q * 3 + 1);
        Ray ray; ray.o = mk(s0.x, s0.y, s0.z); ray.dist = s0.w; ray.d = mk(s1.x, s1.y, s1.z);
        int hs = -1, slot = -1; f3 pos = mk(0, 0, 0); float seg = ray.dist;
        wave_find_intersection<STATS, AN>(sc, 0, n_shapes, prune, shape0, lds, ray, hs, slot, pos, seg, ct);
        if (lane_id() == 0) {
            pb.hitslot[(size_t)q * 2] = make_float4(pos.x, pos.y, pos.z, seg);
            pb.hitslot[(size_t)q * 2 + 1] = make_float4(__int_as_float(hs), __int_as_float(slot), 0.0f, 0.0f);
        }
    }
    if (STATS) flush_counters(sc, ct);
}

// The trace step of a scene whose leading shapes are analytic (RtwRenderParams::lead_shapes > 0).  The lane that set a segment up
// has tested those shapes and, with the reference's own culling test, the boxes of the shapes after them; a ray that meets none
// of those boxes has its complete record already (flag word 0) and is passed over here.  A wave takes 2^chunk_shift list entries
// at a time -- lane j reads entry j's flag -- and runs the wave-per-ray query for the flagged ones only.
template <bool STATS, int NT>
__global__ __launch_bounds__(NT) void trace_wave_lead_kernel(const RtwSceneDev* __restrict__ sc, PipeBufs pb, RtwRenderParams p, int round, int chunk_shift)
{
    HIP_DYNAMIC_SHARED(uint32_t, wave_dyn);              // [NT / 64 waves x RTW_WAVE_LDS_WORDS]
    const bool from_queue = round == 0;
    const uint32_t n = from_queue ? pb.counters[0] : pb.counters[4 + round];
    const uint32_t chunk = 1u << chunk_shift;
    if ((uint32_t)blockIdx.x * (uint32_t)(NT / 64) * chunk >= n) return;
    uint32_t* lds = wave_dyn + (threadIdx.x >> 6) * RTW_WAVE_LDS_WORDS;
    const FlatSrc staged = flat_src_of(sc->shapes[0]);
    const uint32_t* __restrict__ src = from_queue ? pb.queue : wf_list(pb, round & 1);
    const int n_shapes = sc->n_shapes;
    const bool prune = sc->prune != 0;
    const uint32_t wave = (blockIdx.x * (uint32_t)NT + threadIdx.x) >> 6, nwaves = gridDim.x * (uint32_t)(NT / 64);
    const uint32_t lane = (uint32_t)lane_id();
    Counters ct = { 0, 0, 0, 0, 0, 0 };
    for (uint32_t c = wave; (unsigned long long)c * chunk < n; c += nwaves) {
        const uint32_t k = c * chunk + lane;
        const bool mine = lane < chunk && k < n;
        const uint32_t qm = mine ? src[k] : 0u;
        const bool need = mine && qm < pb.capacity && __float_as_int(pb.hitslot[(size_t)qm * 2 + 1].z) != 0;
        unsigned long long todo = __ballot(need);
        while (todo != 0ull) {
            const int l = __ffsll((long long)todo) - 1;
            todo &= todo - 1ull;
            const int q = __builtin_amdgcn_readlane((int)qm, l);
            const float4 s0 = cld4(pb.state, q * 3), s1 = cld4(pb.state, q * 3 + 1);
            Ray ray; ray.o = mk(s0.x, s0.y, s0.z); ray.dist = s0.w; ray.d = mk(s1.x, s1.y, s1.z);
            const float4 h0 = cld4(pb.hitslot, q * 2), h1 = cld4(pb.hitslot, q * 2 + 1);
            f3 pos = mk(h0.x, h0.y, h0.z); float seg = h0.w; int hs = __float_as_int(h1.x), slot = __float_as_int(h1.y);
            wave_find_intersection<STATS, true>(sc, p.lead_shapes, n_shapes, prune, staged, lds, ray, hs, slot, pos, seg, ct);
            if (lane == 0u) {
                pb.hitslot[(size_t)q * 2] = make_float4(pos.x, pos.y, pos.z, seg);
                pb.hitslot[(size_t)q * 2 + 1] = make_float4(__int_as_float(hs), __int_as_float(slot), 0.0f, 0.0f);
            }
        }
    }
    if (STATS) flush_counters(sc, ct);
}
