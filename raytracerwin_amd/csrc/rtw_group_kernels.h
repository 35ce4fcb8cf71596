// rtw_group_kernels.h -- the pass-batched pipeline (pipeline 4, the default).  Included inside rtw_device.hip's anonymous
// namespace after rtw_wave_kernels.h; it uses that file's ray / triangle / shading helpers unchanged.
//
// The passes of the reference's sample loop (UpdateBitmapPixels, Src/RayTracerProgram.cpp:317-361) are independent apart from the
// order in which a pixel's pass colours are added to its accumulator (AccumulatePixel::AddPixel, :51-75).  A GROUP of K consecutive
// passes therefore shares one set of launches -- K times the rays per launch, 1 / K of the launches.  A pixel's pass colours are added to its
// accumulator one by one in pass order (the same float additions as pass-by-pass calls); its accumulator entry and its ARGB word are WRITTEN once per
// group (the ARGB word = GetGammaSpacePixel after the group's last pass): the images between the passes of a group are not materialised.
//
//   gsky_kernel      sky-only tiles (no leaf of any shape can be met from them): one lane per pixel, the K passes in a row with the
//                    accumulator entry in registers (read once, written once per group); the ARGB word is resolved once, after the group's last pass.
//   gprimary_kernel  one wave per (busy tile, pass[, sub-sample]): the camera rays through the tile's screen bin -- the bins walk of
//                    primary_bins_kernel -- and the first shading step.  A path that ends writes its radiance, one that goes on saves
//                    its state in its slot and joins round 0's ray list (ONE atomic per wave).
//   gtrace_kernel    ONE LANE PER RAY over the reference's own binary tree in preorder (skip links, no stack): with K passes in one
//                    launch there are enough rays to fill the chip a ray per lane, and a lane's node visit costs ~30 instructions
//                    where a whole wave per ray spent ~450 per ray on mostly idle lanes.  Leaves whose box is hit are only NOTED
//                    (a per-lane list in LDS); the triangle tests run afterwards, all lanes together, in list order = preorder, each
//                    with the segment left by the previous accepted hit -- the reference's sequence (Src/KdTree.cpp:128-195).
//   gshade_kernel    one lane per path: RayTrace's per-hit block (Src/RayTracerScene.cpp:47-94).
//   gresolve_kernel  one lane per pixel of the busy tiles: for each pass of the group in order, the pass colour from the samples'
//                    radiances and AddPixel; then GetGammaSpacePixel and the ARGB store, once.
//
// Every float operation is the one the other pipelines execute, in the same order; the tests compare them bit for bit.

// RTW_TIMING build (tools/wave_stats.py): per-wave clocks and lane-occupancy counters of the persistent trace kernel, filed in g_rtw_timing
#ifdef RTW_TIMING
#define RTW_TM(...) __VA_ARGS__
#else
#define RTW_TM(...)
#endif
#define RTW_GT_CAP 16           // candidate leaves a lane of gtrace_kernel notes before the wave runs its triangle tests (256-thread blocks, nothing staged)
#ifndef RTW_GT_UNROLL
#define RTW_GT_UNROLL 4      // measured 1 / 2 / 3 / 4 / 8 visits per look: C2 0.895 / 0.879 / 0.890 / 0.860 / 0.870 ms per 20 passes, C4 4.94 / 4.83 / 4.71 / 4.73 / 4.74
#endif
#ifndef RTW_GT_CAP_STAGED
#define RTW_GT_CAP_STAGED 8     // ... in the 1024-thread blocks that stage a tree's upper levels in LDS
#endif

struct GroupBufs {
    float4* __restrict__ rad;       // [capacity]      radiance of a finished path
    float4* __restrict__ state;     // [capacity * 3]  origin + distance | direction + draw counter | pixel, table reads, depth << 16 | levels, -
    float4* __restrict__ hit;       // [capacity * 2]  hit position + distance | shape, leaf slot or part, carry shape, carry slot
    float4* __restrict__ carry;     // [capacity]      texel-inheritance scenes: position of the mesh hit whose sampled colour the hit keeps
    float4* __restrict__ levels;    // [max_bounce][capacity][3]
    uint32_t* __restrict__ list0;   // ray lists (slots), ping-pong
    uint32_t* __restrict__ list1;
    uint32_t* __restrict__ tlist0;  // scenes whose first shapes are analytic (rp.lead_shapes > 0): the subset of a round's list whose query is not finished
    uint32_t* __restrict__ tlist1;  // by the lane that set the segment up -- the trace kernels take these; lengths: counters[24 + round]
    uint32_t* __restrict__ overflow;// rays the budgeted ray-per-lane walk handed over to the wave-per-ray kernel; its length: counters[40 + round]
    uint32_t* __restrict__ counters;// [r] = length of round r's list; [64 + r] = the same at the end of the previous group (for the host)
    uint32_t capacity;
    int32_t carry_on;               // the scene has an analytic shape after a textured mesh (Src/RRay.cpp:53-58,75-80: its hits keep the texel)
};

// A path's slot: lane-fastest -- the 64 paths of one (tile, sub-sample, pass) are neighbours and a tile's paths one compact region, so a wave that
// works on paths in list order (the primary kernel's order, thinned by every round) reads and writes whole cache lines; exactly paths-per-pass x
// passes slots.  The records stay one per slot (state 48 B, hit 32 B, a level 48 B): measured against the round-2 numbering (a pixel's passes
// neighbours, lanes 16 x 48 B apart: C2 +5 %, C3 +8 %, C4 +6 %, C5 +6 %, SetupScene +17 % slower) and against structure-of-arrays records over the
// same slots (a path's record then lies in three places: as slow as the round-2 numbering once the lists thin out).
__device__ __forceinline__ uint32_t group_slot(const RtwGroupParams& g, uint32_t b, uint32_t lane, uint32_t sub, uint32_t k)
{
    return ((b * (uint32_t)g.rp.sub_samples + sub) * (uint32_t)g.n_passes + k) * 64u + lane;
}
#define GST(gb_, slot_, j_) (gb_).state[(size_t)(slot_) * 3 + (j_)]
#define GHT(gb_, slot_, j_) (gb_).hit[(size_t)(slot_) * 2 + (j_)]
// pixel of (tile, lane); false: the lane has no pixel (past the frame's right / bottom edge, outside the rendered range)
__device__ __forceinline__ bool group_xy(const RtwGroupParams& g, int wt, int lane, int& x, int& y)
{
    bool live = work_to_xy(g.rp, wt * 64 + lane, x, y);
    live = live && x < g.rp.width;
    const int pixel = y * g.rp.width + x;
    return live && pixel >= g.range_begin && pixel <= g.range_end;
}

// A list push by a whole BLOCK of NW waves (every thread calls it, the same number of times): one atomic per block instead of one per wave.
// The lists of a round have one counter; ~20 000 waves pushing to it one atomic each measured 6 us per pass of the C2 primary kernel (a third
// of it): returning atomics on one address execute one after another in the L2.  `mine` = entries this wave appends (wave-uniform);
// returns the index of the wave's first entry.  part: NW + 1 words of LDS.
template <int NW>
__device__ __forceinline__ uint32_t block_reserve(uint32_t* __restrict__ counter, uint32_t mine, uint32_t* part)
{
    const uint32_t w = threadIdx.x >> 6;
    if ((threadIdx.x & 63u) == 0u) part[w] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (int i = 0; i < NW; i++) { const uint32_t c = part[i]; part[i] = tot; tot += c; }
        part[NW] = tot != 0u ? atomicAdd(counter, tot) : 0u;
    }
    __syncthreads();
    const uint32_t at = part[NW] + part[w];
    __syncthreads();            // the words are free again for the next call
    return at;
}
template <int NW>
__device__ __forceinline__ void block_push(uint32_t* __restrict__ list, uint32_t* __restrict__ counter, bool flag, uint32_t value, uint32_t* part)
{
    const unsigned long long m = __ballot(flag);
    const uint32_t at = block_reserve<NW>(counter, (uint32_t)__popcll(m), part);
    if (flag) list[at + (uint32_t)mbcnt(m)] = value;
}

// RayTrace's per-hit block (Src/RayTracerScene.cpp:47-94) for one path whose segment has just been traced (r0 = hit position +
// distance, r1 = shape, leaf slot or part, carry).  Returns true when the path goes on (ray, rng, depth, nlev updated, its level
// pushed); false when it ends: L is its radiance, the levels folded back in the reference's association order.
template <bool STATS, bool AN>
__device__ __forceinline__ bool group_shade_step(const RtwSceneDev* __restrict__ sc, const RtwRenderParams& p, const GroupBufs& gb, uint32_t slot,
                                                 Ray& ray, PathRng& rng, int& depth, int& nlev, float4 r0, float4 r1, float4 r2, f3& L, Counters& ct)
{
    const TravCtx tc = make_trav();
    if (!p.preview) prefetch_unit_vector(sc, rng);     // the table read (an HBM miss) overlaps the record loads below
#ifndef RTW_LEVEL_SKIP
#define RTW_LEVEL_SKIP 1
#endif
    // a level's three fields lie in three arrays over the slots; a colour of exactly (1, 1, 1) and an emission of exactly (+0, +0, +0) are not stored
    // (bits 2 and 3 of the kind word say so): x * 1 == x and the fold below still adds the zero, so the radiance has the same bits
    LevelStore lv; lv.ws = gb.levels; lv.stride = (size_t)gb.capacity; lv.tid = (size_t)slot; lv.rec_levels = RTW_LEVEL_SKIP ? 0 : -1;
    L = mk(0, 0, 0);
    bool done = false;
    const int hs = __float_as_int(r1.x), hslot = __float_as_int(r1.y);
    if (hs < 0) { L = sky_color(ray.d.y); done = true; }
    else {
        const RtwShapeDev& sh = sc->shapes[hs];
        Hit h; int tri_index;
        if (AN) {
            hit_finish<STATS>(sc, sh, tc, mk(r0.x, r0.y, r0.z), r0.w, hslot, h, tri_index, ct);
            // one RayHitResult serves all shapes of a query (Src/RayTracerScene.cpp:99-125): a sphere / plane / capsule-side / RTriangle hit keeps
            // the sampled colour and alpha an earlier mesh hit of the same query left there
            const int cs = __float_as_int(r1.z);
            if (gb.carry_on && sh.kind != RTW_SHAPE_MESH && hslot == 0 && cs >= 0) {
                Hit hc; int ti; Counters none = { 0, 0, 0, 0, 0, 0 };
                mesh_finish<false>(sc, sc->shapes[cs], tc, mk(r2.x, r2.y, r2.z), r2.w, __float_as_int(r1.w), hc, ti, none);
                h.color = hc.color; h.alpha = hc.alpha;
            }
        } else {
            mesh_finish<STATS>(sc, sh, tc, mk(r0.x, r0.y, r0.z), r0.w, hslot, h, tri_index, ct);
        }
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
#if RTW_LEVEL_SKIP
                        const bool c1 = __float_as_uint(h.color.x) == 0x3F800000u && __float_as_uint(h.color.y) == 0x3F800000u && __float_as_uint(h.color.z) == 0x3F800000u;
                        const bool e0 = (__float_as_uint(b.em.x) | __float_as_uint(b.em.y) | __float_as_uint(b.em.z)) == 0u;
                        lv.at(nlev, 0) = make_float4(b.att.x, b.att.y, b.att.z, __int_as_float((c1 ? 4 : 0) | (e0 ? 8 : 0)));
                        if (!c1) lv.at(nlev, 1) = make_float4(h.color.x, h.color.y, h.color.z, 0.0f);
                        if (!e0) lv.at(nlev, 2) = make_float4(b.em.x, b.em.y, b.em.z, 0.0f);
#else
                        lv.at(nlev, 0) = make_float4(b.att.x, b.att.y, b.att.z, __int_as_float(0));
                        lv.at(nlev, 1) = make_float4(h.color.x, h.color.y, h.color.z, 0.0f);
                        lv.at(nlev, 2) = make_float4(b.em.x, b.em.y, b.em.z, 0.0f);
#endif
                        nlev++;
                        ray = out;
                    } else { L = mk(0, 0, 0) + b.em; done = true; }
                } else {                 // transparent texel: same direction, remaining distance, no colour factor
                    lv.at(nlev, 0) = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(2));
                    nlev++;
                    const float rd = ray.dist - h.dist;
                    ray.o = h.pos + ray.d * 0.0001f; ray.dist = rd;
                }
                if (!done) { depth--; if (depth == 0) { L = mk(0, 0, 0); done = true; } }
            }
        }
    }
    if (!done) return true;
    for (int kk = nlev - 1; kk >= 0; kk--) {
        const float4 a = lv.at(kk, 0);
        const int kind = __float_as_int(a.w);
        if ((kind & 3) == 0) {
            float4 c = make_float4(1.0f, 1.0f, 1.0f, 0.0f), e = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (!(kind & 4)) c = lv.at(kk, 1);
            if (!(kind & 8)) e = lv.at(kk, 2);
            L = (mk(0, 0, 0) + (mk(a.x, a.y, a.z) * L) * mk(c.x, c.y, c.z)) + mk(e.x, e.y, e.z);
        } else {
            L = mk(0, 0, 0) + L;
        }
    }
    return false;
}

// A scene whose first `lead` shapes are spheres / planes / capsules / triangles (the reference's default scene: six of them before the mesh): the
// lane that sets a segment up runs FindIntersectionWithScene's first `lead` iterations itself, a ray per lane (Src/RayTracerScene.cpp:99-125:
// the same tests in the same order on the same running distance), and gives the boxes of the shapes after them the reference's own culling test
// (plus, for meshes, the walk's conservative segment clip).  A ray that meets none of those boxes has its complete result already and never
// enters a trace kernel; the others continue there from this record.  Returns true when the query must go on.
template <bool STATS>
__device__ __forceinline__ bool group_lead_query(const RtwSceneDev* __restrict__ sc, int lead, const Ray& ray, float4& r0, float4& r1, Counters& ct)
{
    int hs = -1, hslot = -1; f3 hp = mk(0, 0, 0); float seg = ray.dist;
    lead_find<STATS>(sc, lead, ray, seg, hs, hslot, hp, ct);
    const int n_shapes = sc->n_shapes;
    bool more = false; uint32_t tested = 0u;
    const bool cull = sc->prune != 0 && ray_is_tame(ray);
    const float eps_t = 2.0e-5f * fmaxf(fabsf(1.0f / ray.d.x), fmaxf(fabsf(1.0f / ray.d.y), fabsf(1.0f / ray.d.z)));
    for (int s = lead; s < n_shapes; s++) {
        const RtwShapeDev& sh = sc->shapes[s];
        float t0, t1;
        if (sh.kind == RTW_SHAPE_PLANE) { more = true; continue; }
        tested++;
        if (!slab_exact(ray, sh.bmin[0], sh.bmin[1], sh.bmin[2], sh.bmax[0], sh.bmax[1], sh.bmax[2], t0, t1)) continue;
        if (cull && sh.kind == RTW_SHAPE_MESH && (t0 > seg + (eps_t + 1.0e-4f * seg) || t1 < -eps_t)) continue;     // (meshes only: their walk applies the same clip)
        more = true;
    }
    if (STATS && !more) { ct.rays++; ct.boxes += tested; }       // (a ray that goes on is counted by the trace kernel)
    r0 = make_float4(hp.x, hp.y, hp.z, seg);
    r1 = make_float4(__int_as_float(hs), __int_as_float(hslot), __int_as_float(-1), __int_as_float(-1));
    return more;
}

__device__ __forceinline__ void group_save_state(const GroupBufs& gb, uint32_t slot, const Ray& ray, const PathRng& rng, int depth, int nlev, int pixel, uint32_t pass_sub)
{
    GST(gb, slot, 0) = make_float4(ray.o.x, ray.o.y, ray.o.z, ray.dist);
    GST(gb, slot, 1) = make_float4(ray.d.x, ray.d.y, ray.d.z, __uint_as_float(rng.counter));
    GST(gb, slot, 2) = make_float4(__int_as_float(pixel), __uint_as_float(rng.table_reads),
                                                 __uint_as_float(((uint32_t)depth << 16) | (uint32_t)nlev), __uint_as_float(pass_sub));       // pass_sub: (pass in the group) << 2 | sub-sample
}

// ---- sky-only tiles: K passes per pixel with the accumulator entry in registers ----------------------------------------------
__global__ __launch_bounds__(256) void gsky_kernel(const float* __restrict__ gamma_thr, float4* __restrict__ accum, uint32_t* __restrict__ argb, RtwGroupParams g)
{
    __shared__ float thr[256];
    thr[threadIdx.x] = gamma_thr[threadIdx.x];
    __syncthreads();
    const RtwRenderParams& p = g.rp;
    const int npix = p.width * p.height;
    const uint32_t phase = table_phase(p.seed);
    const int wave0 = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), nwaves = (int)(gridDim.x * (blockDim.x >> 6));
    for (int wk = wave0; wk < g.n_sky; wk += nwaves) {
        const int wt = (int)cldu(g.sky_tiles, wk);
        int px = 0, py = 0;
        if (!group_xy(g, wt, lane_id(), px, py)) continue;
        const int pixel = py * p.width + px;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!p.preview) acc = accum[pixel];
        f3 sum = mk(acc.x, acc.y, acc.z);
        int n = __float_as_int(acc.w);
        f3 c = mk(0, 0, 0);
        for (int k = 0; k < g.n_passes; k++) {
            f3 csum = mk(0, 0, 0);
            for (int i = 0; i < p.sub_samples; i++) {
                PathRng rng; rng_init(rng, p.seed, phase, (uint64_t)npix, (uint32_t)pixel, (uint32_t)(g.first_pass + k), (uint32_t)i);
                const Ray ray = camera_ray_xy(p, px, py, i, rng);
                const f3 si = p.max_bounce != 0 ? sky_color(ray.d.y) : mk(0, 0, 0);       // RayTrace(.., 0) is black (Src/RayTracerScene.cpp:39)
                csum = csum + si;
            }
            c = p.sub_samples == 1 ? csum : csum / (float)p.sub_samples;                   // x / 1.0f == x
            if (!p.preview) { sum = sum + c; n++; }                                        // AccumulatePixel::AddPixel, pass after pass in pass order
        }
        // bitcolor[] as it stands after the group's LAST pass: GetGammaSpacePixel of the accumulator (a preview pass shows its own colour).  The images
        // the reference would show BETWEEN the passes of a group are not materialised: nothing can look at them inside one call.
        argb[pixel] = pack_pixel(thr, p.preview ? c : (n == 1 ? sum : sum / (float)n));
        if (!p.preview) accum[pixel] = make_float4(sum.x, sum.y, sum.z, __int_as_float(n));
    }
}

// ---- camera rays of the busy tiles + first shading step -------------------------------------------------------------------------
// The bins walk of one mesh for up to four camera rays per lane at once (the sub-samples of a pixel and / or the same pixel in several passes: they all start
// at the camera and differ by their jitter only): an entry's leaf box and triangle record are fetched and broadcast once, every ray keeps its own
// segment and meets the entries in the list's order (= preorder), so each ray's sequence of tests -- and every bit of its result -- is what it is
// when the rays are walked one after the other.  All rays tame (the caller checks).
template <bool STATS>
__device__ __forceinline__ void bins_walk_rays(const RtwShapeDev& sh, const uint32_t* __restrict__ boff, const uint32_t* __restrict__ bent, int bin, int lane, int nr, bool prune,
                                               const f3 o, const f3 (&d)[4], const bool (&act)[4], float (&cur)[4], f3 (&pos)[4], int (&slot_hit)[4], bool (&any)[4], Counters& ct)
{
    float ix[4], iy[4], iz[4], eps_t[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        ix[r] = 1.0f / d[r].x; iy[r] = 1.0f / d[r].y; iz[r] = 1.0f / d[r].z;
        eps_t[r] = 2.0e-5f * fmaxf(fabsf(ix[r]), fmaxf(fabsf(iy[r]), fabsf(iz[r])));
    }
    const int e0 = (int)cldu(boff, bin), e1 = (int)cldu(boff, bin + 1);
    const float4* nd4 = reinterpret_cast<const float4*>(sh.nodes);
    const float4* tr4 = reinterpret_cast<const float4*>(sh.tris);
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
            const float ax = lox - o.x, bx = hix - o.x, ay = loy - o.y, by = hiy - o.y, az = loz - o.z, bz = hiz - o.z;
            bool hit[4]; bool some = false;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                hit[r] = false;
                if (r < nr) {
                    const float x1 = ax * ix[r], x2 = bx * ix[r], y1 = ay * iy[r], y2 = by * iy[r], z1 = az * iz[r], z2 = bz * iz[r];
                    const float tmin = fmaxf(fmaxf(fminf(x1, x2), fminf(y1, y2)), fminf(z1, z2));
                    const float tmax = fminf(fminf(fmaxf(x1, x2), fmaxf(y1, y2)), fmaxf(z1, z2));
                    bool h = act[r] && (tmax > tmin);
                    if (prune) h = h && !(tmin > cur[r] + (eps_t[r] + 1.0e-4f * cur[r])) && !(tmax < -eps_t[r]);
                    if (STATS) ct.boxes += act[r] ? 1u : 0u;
                    hit[r] = h; some = some || h;
                }
            }
            if (__ballot(some) == 0ull) continue;
            const int leaf = __builtin_amdgcn_readlane(mleaf, j);
            const float4 a = make_float4(readlane_f(ta.x, j), readlane_f(ta.y, j), readlane_f(ta.z, j), readlane_f(ta.w, j));
            const float4 bq = make_float4(readlane_f(tb.x, j), readlane_f(tb.y, j), readlane_f(tb.z, j), readlane_f(tb.w, j));
            const float4 c = make_float4(readlane_f(tc.x, j), readlane_f(tc.y, j), readlane_f(tc.z, j), readlane_f(tc.w, j));
            const float d1 = readlane_f(td, j);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                if (r < nr && hit[r]) {
                    if (STATS) ct.tris++;
                    Ray rr; rr.o = o; rr.d = d[r]; rr.dist = cur[r];
                    f3 cp; float dist;
                    if (triangle_test(rr, cur[r], a, bq, c, d1, cp, dist)) { cur[r] = dist; pos[r] = cp; slot_hit[r] = leaf; any[r] = true; }
                }
            }
        }
    }
}

__device__ __forceinline__ float pick4(int r, float a, float b, float c, float d) { return r == 0 ? a : (r == 1 ? b : (r == 2 ? c : d)); }

template <bool STATS, bool AN>
#ifndef RTW_GPRIMARY_MINB
#define RTW_GPRIMARY_MINB 3      // measured in one session: 3 blocks of 256 per CU (<= 168 VGPRs) makes the primary kernel 13 % faster on unitychan (C4 0.343 -> 0.315 ms per pass), neutral on TorusKnot; 4 blocks (round 3, 128 VGPRs): C2 +3 %, C3 +4 % slower
#endif
__global__ __launch_bounds__(256, RTW_GPRIMARY_MINB) void gprimary_kernel(const RtwSceneDev* __restrict__ sc, GroupBufs gb, RtwGroupParams g)
{
    const RtwRenderParams& p = g.rp;
    const int npix = p.width * p.height;
    Counters ct = { 0, 0, 0, 0, 0, 0 };
    const uint32_t phase = table_phase(p.seed);
    const int n_shapes = sc->n_shapes;
    const bool prune = sc->prune != 0;
    // a wave takes `primary_passes` passes of its job (tile, or tile and sub-sample): at most four rays per lane
    const uint32_t ppw = (g.primary_passes > 1 && g.primary_passes * p.sub_samples <= 4) ? (uint32_t)g.primary_passes : 1u;
    const uint32_t wpj = ((uint32_t)g.n_passes + ppw - 1u) / ppw;
    const uint32_t total = (uint32_t)g.n_jobs * wpj;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    __shared__ uint32_t part[5];
    uint32_t q_out = 0u, tq_out = 0u, b_out = 0u, k_out = 0u, s_out = 0u;
    if (wave < total) {
        // the passes of a tile are neighbouring waves (its bin list stays hot); jobs in the table's order, heaviest bins first
        const uint32_t jidx = wave / wpj, kbase = (wave - jidx * wpj) * ppw;
        const uint32_t npw = (uint32_t)g.n_passes - kbase < ppw ? (uint32_t)g.n_passes - kbase : ppw;
        const uint32_t job = g.jobs ? cldu(g.jobs, (int)jidx) : jidx;
        const uint32_t b = job & 0xFFFFFFu;
        const int only_sample = (int)((job >> 24) & 15u) - 1;        // -1: this wave does every sub-sample of its tile
        const uint32_t nsub = only_sample >= 0 ? 1u : (uint32_t)p.sub_samples;
        const int nr = (int)(nsub * npw);                             // ray sets of this wave: r = (pass - kbase) * nsub + (index of the sub-sample)
        const int wt = g.busy_tiles ? (int)cldu(g.busy_tiles, (int)b) : g.first_tile + (int)b;
        const int lane = lane_id();
        int px = 0, py = 0;
        const bool live = group_xy(g, wt, lane, px, py);
        if (!live) { px = 0; py = 0; }
        const int pixel = live ? py * p.width + px : 0;
        // the wave's bin: all its pixels lie in one tile of the screen's bin grid (tile rows never straddle a bin row); a tile with a
        // live lane starts inside the frame (its first lane has the tile's smallest x and y)
        int bin = 0;
        if (__ballot(live) != 0ull) {
            int tx = 0, ty = 0;
            (void)work_to_xy(p, wt * 64, tx, ty);
            bin = (ty >> (6 - p.tile_shift)) * p.tiles_per_row + (tx >> p.tile_shift);
        }
        bool near_wave = false;          // can any sample of this wave's pixels hit anything?
        for (int k = 0; k < n_shapes; k++) {
            const uint32_t* __restrict__ boff = p.bins[k].off;
            near_wave = near_wave || boff == nullptr || cldu(boff, bin) != cldu(boff, bin + 1);
        }
        // ---- one mesh with bins and several ray sets: the bins walk once for all of them
        float w_cur[4] = { 0.f, 0.f, 0.f, 0.f }; f3 w_pos[4]; int w_slot[4] = { -1, -1, -1, -1 }; bool w_any[4] = { false, false, false, false };
#pragma unroll
        for (int r = 0; r < 4; r++) w_pos[r] = mk(0, 0, 0);
        bool shared_walk = false;
        if (!AN && g.primary_passes > 0 && n_shapes == 1 && nr > 1 && nr <= 4 && near_wave && p.max_bounce != 0 && p.bins[0].off != nullptr) {
            f3 w_d[4]; bool w_act[4]; bool all_tame = true;
            const RtwShapeDev& sh = sc->shapes[0];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                w_d[r] = mk(0, 0, -1); w_act[r] = false;
                if (r < nr) {
                    const uint32_t po = (uint32_t)r / nsub, i = only_sample >= 0 ? (uint32_t)only_sample : (uint32_t)r - po * nsub;
                    PathRng rng; rng_init(rng, p.seed, phase, (uint64_t)npix, (uint32_t)pixel, (uint32_t)g.first_pass + kbase + po, i);
                    const Ray ray = camera_ray_xy(p, px, py, (int)i, rng);
                    float t0, t1;
                    w_d[r] = ray.d; w_cur[r] = ray.dist;
                    w_act[r] = live && slab_exact(ray, sh.bmin[0], sh.bmin[1], sh.bmin[2], sh.bmax[0], sh.bmax[1], sh.bmax[2], t0, t1);
                    all_tame = all_tame && (!live || ray_is_tame(ray));
                }
            }
            if (__ballot(!all_tame) == 0ull) {
                shared_walk = true;
                bins_walk_rays<STATS>(sh, p.bins[0].off, p.bins[0].ent, bin, lane, nr, prune, mk(0, 0, 7.0f), w_d, w_act, w_cur, w_pos, w_slot, w_any, ct);
            }
        }
        uint32_t queued = 0u, tqueued = 0u;
        for (int r = 0; r < nr; r++) {                           // wave-uniform loop
            const uint32_t po = (uint32_t)r / nsub, kpass = kbase + po;
            const int i = only_sample >= 0 ? only_sample : (int)((uint32_t)r - po * nsub);
            const int pass = g.first_pass + (int)kpass;
            PathRng rng; rng_init(rng, p.seed, phase, (uint64_t)npix, (uint32_t)pixel, (uint32_t)pass, (uint32_t)i);
            const Ray ray = camera_ray_xy(p, px, py, i, rng);
            if (STATS && live) ct.cams++;
            const uint32_t slot = group_slot(g, b, (uint32_t)lane, (uint32_t)i, kpass);
            f3 si = mk(0, 0, 0);
            bool have_hit = false;
            float4 hr0 = make_float4(0.f, 0.f, 0.f, 0.f), hr1 = hr0, hr2 = hr0;
            if (p.max_bounce != 0 && !near_wave && !STATS) {     // no leaf of any shape can be met from this tile: every sample sees the sky
                if (live) si = sky_color(ray.d.y);
            } else if (shared_walk) {                            // the walk above found this ray's closest hit
                if (STATS && live) { ct.rays++; ct.boxes++; }
                const bool any = r == 0 ? w_any[0] : (r == 1 ? w_any[1] : (r == 2 ? w_any[2] : w_any[3]));
                if (live) {
                    if (!any) si = sky_color(ray.d.y);
                    else {
                        have_hit = true;
                        hr0 = make_float4(pick4(r, w_pos[0].x, w_pos[1].x, w_pos[2].x, w_pos[3].x), pick4(r, w_pos[0].y, w_pos[1].y, w_pos[2].y, w_pos[3].y),
                                          pick4(r, w_pos[0].z, w_pos[1].z, w_pos[2].z, w_pos[3].z), pick4(r, w_cur[0], w_cur[1], w_cur[2], w_cur[3]));
                        const int hslot = r == 0 ? w_slot[0] : (r == 1 ? w_slot[1] : (r == 2 ? w_slot[2] : w_slot[3]));
                        hr1 = make_float4(__int_as_float(0), __int_as_float(hslot), __int_as_float(-1), __int_as_float(-1));
                    }
                }
            } else if (p.max_bounce != 0) {                      // RayTrace(.., 0) is black (Src/RayTracerScene.cpp:39)
                // FindIntersectionWithScene of the camera ray (Src/RayTracerScene.cpp:99-125), shapes in insertion order
                int hit_shape = -1, hit_slot = -1, carry_shape = -1, carry_slot = -1;
                f3 hit_pos = mk(0, 0, 0), carry_pos = mk(0, 0, 0);
                float seg = ray.dist, carry_dist = 0.0f;
                if (STATS && live) ct.rays++;
                const bool tame = ray_is_tame(ray);
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
                    float cur = seg; f3 pos = mk(0, 0, 0); int slot_hit = -1;
                    bool any = false;
                    const uint32_t* __restrict__ boff = p.bins[k].off;
                    if (AN && kind != RTW_SHAPE_MESH) {          // a sphere / plane / capsule / triangle: every lane tests its own ray; slot = the part hit
                        if (inbox) any = analytic_test(sh, ray, seg, pos, cur, slot_hit);
                    } else if (boff == nullptr) {                // no bins for this shape: packet walk of its tree
                        any = packet_walk<STATS, false>(sh.nodes, sh.tris, sh.n_nodes, ray, inbox && tame, prune, cur, pos, slot_hit, ct);
                        if (any_untame) {
                            const bool a2 = packet_walk<STATS, true>(sh.nodes, sh.tris, sh.n_nodes, ray, inbox && !tame, false, cur, pos, slot_hit, ct);
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
                                const float4 bq = make_float4(readlane_f(tb.x, j), readlane_f(tb.y, j), readlane_f(tb.z, j), readlane_f(tb.w, j));
                                const float4 c = make_float4(readlane_f(tc.x, j), readlane_f(tc.y, j), readlane_f(tc.z, j), readlane_f(tc.w, j));
                                const float d1 = readlane_f(td, j);
                                if (hit) {
                                    if (STATS) ct.tris++;
                                    f3 cp; float dist;
                                    if (triangle_test(ray, cur, a, bq, c, d1, cp, dist)) { cur = dist; pos = cp; slot_hit = leaf; any = true; }
                                }
                            }
                        }
                    }
                    if (any) {
                        seg = cur; hit_shape = k; hit_slot = slot_hit; hit_pos = pos;
                        if (AN && kind == RTW_SHAPE_MESH) { carry_shape = k; carry_slot = slot_hit; carry_pos = pos; carry_dist = cur; }
                        else if (AN && slot_hit != 0) carry_shape = -1;      // a capsule end resets the sampled colour (Src/Shapes.cpp:34-62)
                    }
                }
                if (live) {
                    if (hit_shape < 0) si = sky_color(ray.d.y);
                    else {
                        have_hit = true;
                        hr0 = make_float4(hit_pos.x, hit_pos.y, hit_pos.z, seg);
                        hr1 = make_float4(__int_as_float(hit_shape), __int_as_float(hit_slot), __int_as_float(carry_shape), __int_as_float(carry_slot));
                        hr2 = make_float4(carry_pos.x, carry_pos.y, carry_pos.z, carry_dist);
                    }
                }
            }
            if (have_hit) {                                      // shade the hit here: the path's slot needs no queue position
                Ray rs = ray; PathRng rg = rng; int depth = p.max_bounce, nlev = 0;
                f3 L;
                if (group_shade_step<STATS, AN>(sc, p, gb, slot, rs, rg, depth, nlev, hr0, hr1, hr2, L, ct)) {
                    group_save_state(gb, slot, rs, rg, depth, nlev, pixel, (kpass << 2) | (uint32_t)i);
                    queued |= 1u << r;
                    if (AN && p.lead_shapes > 0) {
                        float4 q0, q1;
                        if (group_lead_query<STATS>(sc, p.lead_shapes, rs, q0, q1, ct)) tqueued |= 1u << r;
                        GHT(gb, slot, 0) = q0; GHT(gb, slot, 1) = q1;
                    }
                } else {
                    si = mk(0, 0, 0) + L;
                    gb.rad[slot] = make_float4(si.x, si.y, si.z, 0.0f);
                }
            } else if (live) {
                gb.rad[slot] = make_float4(si.x, si.y, si.z, 0.0f);
            }
        }
        q_out = queued; tq_out = tqueued; b_out = b; k_out = kbase; s_out = (nsub << 8) | (uint32_t)(only_sample >= 0 ? only_sample : 0);
    }
    {
        // ONE atomic per BLOCK: round 0's list gets an entry per ray set that goes on (every wave of the block comes here, also one without a job)
        const uint32_t queued = q_out, tqueued = tq_out, b = b_out, kbase = k_out, nsub = s_out >> 8 ? s_out >> 8 : 1u, sub0 = s_out & 255u;
        const int lane = lane_id();
        const unsigned long long m0 = __ballot((queued & 1u) != 0u), m1 = __ballot((queued & 2u) != 0u), m2 = __ballot((queued & 4u) != 0u),
                                 m3 = __ballot((queued & 8u) != 0u);
        const uint32_t c0 = (uint32_t)__popcll(m0), c1 = (uint32_t)__popcll(m1), c2 = (uint32_t)__popcll(m2), c3 = (uint32_t)__popcll(m3);
        const uint32_t qb = block_reserve<4>(&gb.counters[0], c0 + c1 + c2 + c3, part);
        auto slot_of = [&](uint32_t r) { const uint32_t po = r / nsub; return group_slot(g, b, (uint32_t)lane, nsub == 1u ? sub0 : r - po * nsub, kbase + po); };
        if (queued & 1u) gb.list0[qb + (uint32_t)mbcnt(m0)] = slot_of(0u);
        if (queued & 2u) gb.list0[qb + c0 + (uint32_t)mbcnt(m1)] = slot_of(1u);
        if (queued & 4u) gb.list0[qb + c0 + c1 + (uint32_t)mbcnt(m2)] = slot_of(2u);
        if (queued & 8u) gb.list0[qb + c0 + c1 + c2 + (uint32_t)mbcnt(m3)] = slot_of(3u);
        if (AN && p.lead_shapes > 0) {       // the subset that still has to be traced
            for (uint32_t r = 0; r < 4u; r++)
                block_push<4>(gb.tlist0, &gb.counters[24], (tqueued >> r) & 1u, slot_of(r), part);
        }
    }
    if (STATS) flush_counters(sc, ct);
}

// ---- one round of secondary segments: a ray per lane --------------------------------------------------------------------------
// The reference's box test of one node for one lane's ray (RRay::TestIntersectionWithAabb, Src/RRay.cpp:89-136): tame rays through
// hoisted reciprocals and v_min / v_max (the same values), the others -- a direction component below FLT_EPSILON, NaN, ... -- with
// the test as written (skipped axes, Math::Min / Max ternaries); the conservative segment clip only for tame rays.
struct LaneRay { f3 o; float ix, iy, iz, eps_t; bool tame, skx, sky, skz; };
__device__ __forceinline__ LaneRay lane_ray_of(const Ray& r)
{
    LaneRay q;
    q.o = r.o;
    q.tame = ray_is_tame(r);
    q.skx = near_zero(r.d.x); q.sky = near_zero(r.d.y); q.skz = near_zero(r.d.z);
    q.ix = (!q.tame && q.skx) ? 0.0f : 1.0f / r.d.x; q.iy = (!q.tame && q.sky) ? 0.0f : 1.0f / r.d.y; q.iz = (!q.tame && q.skz) ? 0.0f : 1.0f / r.d.z;
    q.eps_t = 2.0e-5f * fmaxf(fabsf(q.ix), fmaxf(fabsf(q.iy), fabsf(q.iz)));
    return q;
}

// KdNode::TestRayIntersection (Src/KdTree.cpp:128-195) of every lane's own ray on one mesh: the preorder walk of tree_walk with the
// triangle tests postponed, over the explicit-link records (RtwShapeDev::tnodes).  `go`: this lane's ray passed the shape's culling
// box.  `lnodes`: the first `ltop` records, staged in LDS by the block (ALLDS: that is the whole tree, the walk never leaves LDS).
// `cand`: the block's candidate lists, entry j of thread t at [j * NT + t], CAP entries per lane.  On return cur / pos / leaf hold
// what the reference's recursion leaves in TestRay.Distance / *OutResult / *TriangleIndex.
//
// The box test (RRay::TestIntersectionWithAabb, Src/RRay.cpp:89-136) of a tame ray: (pair - origin) * reciprocal per axis as packed
// operations, v_min / v_max (no NaN, no skipped axis: the same values as the reference's ternaries), then the conservative segment
// clip.  A wave that holds a ray that is not tame takes the loop that also evaluates the test as written.
template <bool STATS, int NT, int CAP, bool ALLDS>
__device__ __forceinline__ bool lane_mesh_walk(const RtwShapeDev& sh, uint32_t* __restrict__ cand, const float4* __restrict__ lnodes, int ltop,
                                               const Ray& r, const LaneRay& q, bool go, bool prune, float& cur, f3& pos, int& leaf_out, Counters& ct)
{
    const float4* nd4 = reinterpret_cast<const float4*>(sh.tnodes);
    const float4* tr4 = reinterpret_cast<const float4*>(sh.tris);
    const int n_nodes = sh.n_nodes;
    const int tid = (int)threadIdx.x;
    const bool any_untame = __ballot(go && !q.tame) != 0ull;
    const rtw_v2f ox = { q.o.x, q.o.x }, oy = { q.o.y, q.o.y }, oz = { q.o.z, q.o.z };
    const rtw_v2f vx = { q.ix, q.ix }, vy = { q.iy, q.iy }, vz = { q.iz, q.iz };
    const float neg_eps = prune ? -q.eps_t : -INFINITY;
    bool any = false;
    int i = go ? 0 : n_nodes;
    int ncand = 0;
    for (;;) {
        const float far_t = prune ? cur + (q.eps_t + 1.0e-4f * cur) : INFINITY;       // the segment only changes in the triangle phase below
        // ---- walk: every lane steps through its own records until all are done or some lane's list is full ----
        if (!any_untame) {
            for (;;) {
                if (__ballot(i < n_nodes) == 0ull) break;
            #pragma unroll
                for (int u = 0; u < RTW_GT_UNROLL; u++) {        // several visits between two looks at the wave (see gtrace_persist_kernel)
                const bool walking = (i < n_nodes) & (ncand < CAP);
                if (walking) {
                    float4 a, b;
                    if (ALLDS || i < ltop) { a = lld4(lnodes, 2 * i); b = lld4(lnodes, 2 * i + 1); }
                    else { a = gld4(nd4, 2 * (size_t)i); b = gld4(nd4, 2 * (size_t)i + 1); }
                    const int skip = __float_as_int(b.z), link = __float_as_int(b.w);
                    const rtw_v2f bx = { a.x, a.y }, by = { a.z, a.w }, bz = { b.x, b.y };
                    const rtw_v2f tx = (bx - ox) * vx, ty = (by - oy) * vy, tz = (bz - oz) * vz;
                    const float tmin = fmaxf(fmaxf(fminf(tx.x, tx.y), fminf(ty.x, ty.y)), fminf(tz.x, tz.y));
                    const float tmax = fminf(fminf(fmaxf(tx.x, tx.y), fmaxf(ty.x, ty.y)), fmaxf(tz.x, tz.y));
                    const bool hit = (tmax > tmin) & !(tmin > far_t) & !(tmax < neg_eps);
                    const bool leaf = link >= 0;
                    if (STATS) ct.boxes++;
                    if (hit & leaf) { lstu(cand, ncand * NT + tid, (uint32_t)link); ncand++; }
                    i = (hit & !leaf) ? ~link : skip;
                }
                }
                if (__ballot(ncand == CAP) != 0ull) break;
            }
        } else {
            for (;;) {
                const bool walking = i < n_nodes;
                if (__ballot(walking) == 0ull) break;
                            if (walking) {
                    float4 a, b;
                    if (ALLDS || i < ltop) { a = lld4(lnodes, 2 * i); b = lld4(lnodes, 2 * i + 1); }
                    else { a = gld4(nd4, 2 * (size_t)i); b = gld4(nd4, 2 * (size_t)i + 1); }
                    const int skip = __float_as_int(b.z), link = __float_as_int(b.w);
                    const float x1 = (a.x - q.o.x) * q.ix, x2 = (a.y - q.o.x) * q.ix;
                    const float y1 = (a.z - q.o.y) * q.iy, y2 = (a.w - q.o.y) * q.iy;
                    const float z1 = (b.x - q.o.z) * q.iz, z2 = (b.y - q.o.z) * q.iz;
                    const float tmin = fmaxf(fmaxf(fminf(x1, x2), fminf(y1, y2)), fminf(z1, z2));
                    const float tmax = fminf(fminf(fmaxf(x1, x2), fmaxf(y1, y2)), fmaxf(z1, z2));
                    bool hit = (tmax > tmin) & !(tmin > far_t) & !(tmax < neg_eps);
                    float emin = -FLT_MAX, emax = FLT_MAX;      // RRay::TestIntersectionWithAabb as written, for the lanes that need it
                    if (!q.skx) { emin = ref_max(emin, ref_min(x1, x2)); emax = ref_min(emax, ref_max(x1, x2)); }
                    if (!q.sky) { emin = ref_max(emin, ref_min(y1, y2)); emax = ref_min(emax, ref_max(y1, y2)); }
                    if (!q.skz) { emin = ref_max(emin, ref_min(z1, z2)); emax = ref_min(emax, ref_max(z1, z2)); }
                    if (!q.tame) hit = emax > emin;
                    const bool leaf = link >= 0;
                    if (STATS) ct.boxes++;
                    if (hit & leaf) { lstu(cand, ncand * NT + tid, (uint32_t)link); ncand++; }
                    i = (hit & !leaf) ? ~link : skip;
                }
                if (__ballot(ncand == CAP) != 0ull) break;
            }
        }
        // ---- the noted leaves' triangle tests, all lanes together, each lane through its own list in order; the next record is
        // fetched while the current one is tested ----
        {
            bool mine = 0 < ncand;
            int leaf = mine ? (int)lldu(cand, tid) : 0;
            float4 ta = gld4(tr4, 4 * (size_t)leaf), tb = gld4(tr4, 4 * (size_t)leaf + 1), tc = gld4(tr4, 4 * (size_t)leaf + 2);
            float td = gld1(reinterpret_cast<const float*>(tr4), 16 * (size_t)leaf + 12);
            for (int j = 0; __ballot(mine) != 0ull; j++) {
                const bool nmine = j + 1 < ncand;
                const int nleaf = nmine ? (int)lldu(cand, (j + 1) * NT + tid) : 0;
                const float4 na = gld4(tr4, 4 * (size_t)nleaf), nb = gld4(tr4, 4 * (size_t)nleaf + 1), nc = gld4(tr4, 4 * (size_t)nleaf + 2);
                const float nd = gld1(reinterpret_cast<const float*>(tr4), 16 * (size_t)nleaf + 12);
                if (mine) {
                    if (STATS) ct.tris++;
                    f3 cp; float dist;
                    if (triangle_test(r, cur, ta, tb, tc, td, cp, dist)) { cur = dist; pos = cp; leaf_out = leaf; any = true; }
                }
                mine = nmine; leaf = nleaf; ta = na; tb = nb; tc = nc; td = nd;
            }
        }
        ncand = 0;
        if (__ballot(i < n_nodes) == 0ull) break;
    }
    return any;
}

// staged_shape: the shape whose upper tree levels the block holds in LDS (lnodes, ltop), or -1
template <bool STATS, bool AN, int NT, int CAP, bool ALLDS>
__device__ __forceinline__ void lane_find_intersection(const RtwSceneDev* __restrict__ sc, uint32_t* __restrict__ cand, const float4* __restrict__ lnodes, int ltop, int staged_shape,
                                                       const Ray& ray, bool have, int first_shape,
                                                       int& hit_shape, int& hit_slot, f3& hit_pos, float& seg,
                                                       int& carry_shape, int& carry_slot, f3& carry_pos, float& carry_dist, Counters& ct)
{
    const int n_shapes = sc->n_shapes;
    const bool prune = sc->prune != 0;
    const LaneRay q = lane_ray_of(ray);
    if (STATS && have) ct.rays++;
    for (int s = first_shape; s < n_shapes; s++) {
        const RtwShapeDev& sh = sc->shapes[s];
        const int kind = AN ? sh.kind : RTW_SHAPE_MESH;
        float t0, t1;
        const bool inbox = have && ((AN && kind == RTW_SHAPE_PLANE) ||
                                    slab_exact(ray, sh.bmin[0], sh.bmin[1], sh.bmin[2], sh.bmax[0], sh.bmax[1], sh.bmax[2], t0, t1));
        if (STATS && have && kind != RTW_SHAPE_PLANE) ct.boxes++;
        if (__ballot(inbox) == 0ull) continue;
        float cur = seg; f3 pos = mk(0, 0, 0); int slot = -1;
        bool any = false;
        if (AN && kind != RTW_SHAPE_MESH) {
            if (inbox) any = analytic_test(sh, ray, seg, pos, cur, slot);
        } else if (sh.n_nodes > 0) {
            if (ALLDS && s == staged_shape) any = lane_mesh_walk<STATS, NT, CAP, true>(sh, cand, lnodes, ltop, ray, q, inbox, prune, cur, pos, slot, ct);
            else any = lane_mesh_walk<STATS, NT, CAP, false>(sh, cand, lnodes, s == staged_shape ? ltop : 0, ray, q, inbox, prune, cur, pos, slot, ct);
        }
        if (any) {
            seg = cur; hit_shape = s; hit_slot = slot; hit_pos = pos;
            if (AN && kind == RTW_SHAPE_MESH) { carry_shape = s; carry_slot = slot; carry_pos = pos; carry_dist = cur; }
            else if (AN && slot != 0) carry_shape = -1;
        }
    }
}

// NT threads per block.  STAGE: the block first copies the upper levels of shape `staged_shape`'s tree (RtwShapeDev::tnodes_top records)
// into LDS -- for the config meshes' 2 399 / 1 935 nodes that is the whole tree -- and its waves walk them there: a step of the walk is
// one dependent node fetch, ~100 ns from LDS against ~600 ns through L2.
// STAGE 0: nothing staged; 1: the upper levels (the walk goes on in global memory below them); 2: the whole tree fits (ALLDS)
template <bool STATS, bool AN, int NT, int CAP, int STAGE>
__global__ __launch_bounds__(NT) void gtrace_kernel(const RtwSceneDev* __restrict__ sc, GroupBufs gb, int round, int staged_shape, int lead)
{
    HIP_DYNAMIC_SHARED(uint32_t, gt_dyn);                 // [CAP * NT candidate words | staged records]
    uint32_t* cand = gt_dyn;
    // lead > 0: the scene's first `lead` shapes were tested by the lane that set the segment up; the list holds the rays that go on, their records the state so far
    const uint32_t nl = lead > 0 ? gb.counters[24 + round] : gb.counters[round];
    const uint32_t n = nl < gb.capacity ? nl : gb.capacity;
    if ((uint32_t)blockIdx.x * (uint32_t)NT >= n) return;        // (whole block) the grid is sized from the previous group's list length
    const float4* lnodes = nullptr; int ltop = 0;
    if (STAGE) {
        const RtwShapeDev& s0 = sc->shapes[staged_shape];
        ltop = s0.tnodes_top;
        float4* dst = reinterpret_cast<float4*>(gt_dyn + CAP * NT);
        const float4* src4 = reinterpret_cast<const float4*>(s0.tnodes);
        for (int i = (int)threadIdx.x; i < ltop * 2; i += NT) dst[i] = gld4(src4, (size_t)i);
        lnodes = dst;
        __syncthreads();
    }
    const uint32_t* __restrict__ src = lead > 0 ? (round & 1 ? gb.tlist1 : gb.tlist0) : (round & 1 ? gb.list1 : gb.list0);
    const uint32_t nthreads = gridDim.x * (uint32_t)NT;
    const uint32_t lane = (uint32_t)lane_id();
    Counters ct = { 0, 0, 0, 0, 0, 0 };
    for (uint32_t base = blockIdx.x * (uint32_t)NT + (threadIdx.x & ~63u); base < n; base += nthreads) {     // wave-uniform
        const uint32_t k = base + lane;
        const bool have = k < n;
        const uint32_t slot = have ? src[k] : 0u;
        Ray ray; ray.o = mk(0, 0, 0); ray.d = mk(0, 0, 1); ray.dist = 0.0f;
        if (have) {
            const float4 s0 = GST(gb, slot, 0), s1 = GST(gb, slot, 1);
            ray.o = mk(s0.x, s0.y, s0.z); ray.dist = s0.w; ray.d = mk(s1.x, s1.y, s1.z);
        }
        int hs = -1, hslot = -1, cs = -1, cslot = -1; f3 pos = mk(0, 0, 0), cpos = mk(0, 0, 0); float seg = ray.dist, cdist = 0.0f;
        if (AN && lead > 0 && have) {        // continue the query from what the shading lane found among the leading shapes
            const float4 h0 = GHT(gb, slot, 0), h1 = GHT(gb, slot, 1);
            pos = mk(h0.x, h0.y, h0.z); seg = h0.w; hs = __float_as_int(h1.x); hslot = __float_as_int(h1.y);
        }
        lane_find_intersection<STATS, AN, NT, CAP, (STAGE == 2)>(sc, cand, lnodes, ltop, STAGE ? staged_shape : -1, ray, have, AN ? lead : 0, hs, hslot, pos, seg, cs, cslot, cpos, cdist, ct);
        if (have) {
            GHT(gb, slot, 0) = make_float4(pos.x, pos.y, pos.z, seg);
            GHT(gb, slot, 1) = make_float4(__int_as_float(hs), __int_as_float(hslot), __int_as_float(cs), __int_as_float(cslot));
            if (AN && gb.carry_on) gb.carry[slot] = make_float4(cpos.x, cpos.y, cpos.z, cdist);
        }
    }
    if (STATS) flush_counters(sc, ct);
}

// ---- the same round for a scene that is ONE mesh: persistent waves that refill their lanes ------------------------------------
// In the kernel above a wave walks 64 rays in lock step until the slowest is done (the mean lane is busy ~40 % of that time) and a
// block keeps its CU until its slowest wave is done.  Here the grid holds one block per CU, every block owns an interleaved share of
// the list and its waves draw 64-entry batches from it through a counter in LDS: whenever RTW_GT_REFILL lanes have finished their walk (or some lane's candidate list is full) the wave runs the
// noted leaves' triangle tests for all its lanes, writes the finished lanes' hit records and hands those lanes new rays -- out of the
// 64 it fetched ahead into registers while it walked (one atomic cursor for all waves was measured: 60 k same-address atomics per
// launch serialise, 14 us per refill).  A lane's sequence of box tests and triangle tests is untouched by what its neighbours do.
// Entry of the round's list that this lane stages for batch `batch` (n entries, n_batches = ceil(n / 64)); count = the batch's entries (they
// sit in lanes 0 .. count - 1); 0xFFFFFFFF: none.  SPREAD: batch b is entries b, b + n_batches, b + 2 n_batches, ... (lane l reads the l-th
// 64th of the list): neighbours in the list are neighbours on the screen and cost alike, so 64 consecutive entries make cheap batches and
// dear ones (per-wave durations 2 x apart, the kernel as slow as its slowest wave) while a spread batch is a sample of the whole list
// (slowest wave / mean 1.84 -> 1.41 on C2; the waves run ~13 % more iterations each, their lanes' rays being less alike).
#ifndef RTW_GT_SPREAD
#define RTW_GT_SPREAD 1
#endif
template <bool SPREAD>
__device__ __forceinline__ uint32_t batch_entry(uint32_t batch, uint32_t n_batches, uint32_t n, uint32_t& count)
{
    count = 0u;
    if (batch >= n_batches) return 0xFFFFFFFFu;
    const uint32_t lane = (uint32_t)lane_id();
    if (SPREAD) {
        count = 63u * n_batches + batch < n ? 64u : (n - batch + n_batches - 1u) / n_batches;      // lanes l with l * n_batches + batch < n
        const uint32_t k = lane * n_batches + batch;
        return k < n ? k : 0xFFFFFFFFu;
    }
    const uint32_t first = batch * 64u;
    count = n - first < 64u ? n - first : 64u;
    return first + lane < n ? first + lane : 0xFFFFFFFFu;
}
#ifndef RTW_GT_REFILL
#define RTW_GT_REFILL 16
#endif
// LEAD: the scene is `ms` leading spheres / planes / capsules / triangles followed by this ONE mesh (the reference's default scene): the lane that set
// the segment up has tested the leading shapes (group_lead_query); the list is the rays that can still meet the mesh's box, their hit records hold the
// query so far, and the mesh's hits are measured against the segment those shapes left.
// PLANES (only with the whole tree in LDS): the triangles' planes are staged too, and a leaf whose triangle faces away from the ray's origin -- the first
// rejection of RRay::TestIntersectionWithTriangle, d2 = N . O - N . p0 < 0 (Src/RRay.cpp:156-158), independent of the segment -- is not noted at all:
// about half the leaves a ray meets on a closed mesh, so half the triangle iterations (whose lanes run a quarter full).
template <bool STATS, int NT, int CAP, int STAGE, bool LEAD, bool PLANES>
__global__ __launch_bounds__(NT) void gtrace_persist_kernel(const RtwSceneDev* __restrict__ sc, GroupBufs gb, int round, int budget, int ms)
{
    HIP_DYNAMIC_SHARED(uint32_t, gt_dyn);                 // [CAP * NT candidate words | staged records]
    RTW_TM(const unsigned long long tm_entry = wall_clock64(); unsigned long long tm_walk = 0ull, tm_tri = 0ull, tm_refill = 0ull; uint32_t tm_wtrips = 0u, tm_wlanes = 0u, tm_ttrips = 0u, tm_tlanes = 0u, tm_events = 0u, tm_rays = 0u;)
    uint32_t* cand = gt_dyn;
    const uint32_t nl = LEAD ? gb.counters[24 + round] : gb.counters[round];
    const uint32_t n = nl < gb.capacity ? nl : gb.capacity;
    const RtwShapeDev& sh = sc->shapes[LEAD ? ms : 0];
    const float4* lnodes = nullptr; int ltop = 0;
    // Who takes which 64-entry batch of the list.  A tree that lives in LDS entirely (STAGE 2: the config meshes TorusKnot and BlenderMonkey):
    // static, wave w of the grid takes batches w, w + waves, ... -- neighbouring batches (neighbours on the screen) on neighbouring waves.
    // A larger tree (unitychan: most node records come through L2, a batch's cost varies more): every block owns batches b, b + blocks, ...
    // and its waves DRAW them in turn from a counter in LDS, so that a wave that met cheap rays takes more.  Measured at 20 passes per group:
    // dynamic C4 0.277 -> 0.261, C5 1.126 -> 1.053 ms per pass, but C2 0.057 -> 0.064 and C3 0.155 -> 0.157 (hence the split).
    constexpr bool DRAW = STAGE != 2;
    constexpr bool SPREAD = RTW_GT_SPREAD != 0 && STAGE == 2;       // (measured: bounce rounds -4 % on C2, -2.5 % on C3; +1 % on C4 / C5, whose deep node records like neighbours together)
    __shared__ uint32_t blk_next;
    if (threadIdx.x == 0) blk_next = (uint32_t)(NT / 64);       // each wave starts with the batch of its index in the block
    if (STAGE) {
        ltop = sh.tnodes_top;
        float4* dst = reinterpret_cast<float4*>(gt_dyn + CAP * NT);
        const float4* src4 = reinterpret_cast<const float4*>(sh.tnodes);
        for (int k = (int)threadIdx.x; k < ltop * 2; k += NT) dst[k] = gld4(src4, (size_t)k);
        lnodes = dst;
    }
    const float4* lplanes = nullptr;
    if (PLANES) {
        float4* dst = reinterpret_cast<float4*>(gt_dyn + CAP * NT) + (size_t)sh.tnodes_top * 2;
        const float4* src4 = reinterpret_cast<const float4*>(sh.planes);
        for (int k = (int)threadIdx.x; k < sh.n_tris; k += NT) dst[k] = gld4(src4, (size_t)k);
        lplanes = dst;
    }
    __syncthreads();
    constexpr bool ALLDS = STAGE == 2;
    const float4* nd4 = reinterpret_cast<const float4*>(sh.tnodes);
    const float4* tr4 = reinterpret_cast<const float4*>(sh.tris);
    const int n_nodes = sh.n_nodes;
    const bool prune = sc->prune != 0;
    const uint32_t* __restrict__ src = LEAD ? (round & 1 ? gb.tlist1 : gb.tlist0) : (round & 1 ? gb.list1 : gb.list0);
    const int tid = (int)threadIdx.x;
    Counters ct = { 0, 0, 0, 0, 0, 0 };
    // the lane's ray and the state of its walk
    bool have = false, tame = true, skx = false, sky = false, skz = false;
    uint32_t slot = 0u;
    Ray r; r.o = mk(0, 0, 0); r.d = mk(0, 0, 1); r.dist = 0.0f;
    float ix = 0.0f, iy = 0.0f, iz = 0.0f, eps_t = 0.0f, cur = 0.0f;
    f3 pos = mk(0, 0, 0);
    int i = n_nodes, leaf_out = -1, ncand = 0, visits = 0;
    int lead_hs = -1, lead_slot = -1;       // LEAD: what the leading shapes' tests left (shape, part); the mesh's hit, if any, replaces it
    // the wave's share of the list: every nw-th batch of 64 entries (neighbouring entries come from neighbouring pixels and cost alike:
    // contiguous shares were measured 2 x out of balance), fetched one batch ahead into staging registers
    uint32_t st_count, st_used = 0u, st_slot = 0u;
    float4 st_s0 = make_float4(0.f, 0.f, 0.f, 0.f), st_s1 = st_s0;
    const uint32_t n_batches = (n + 63u) >> 6;
    uint32_t static_next = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (uint32_t)(NT / 64) + (threadIdx.x >> 6))) + gridDim.x * (uint32_t)(NT / 64);
    {
        const uint32_t j = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        const uint32_t batch = DRAW ? j * gridDim.x + blockIdx.x : blockIdx.x * (uint32_t)(NT / 64) + j;
        const uint32_t k = batch_entry<SPREAD>(batch, n_batches, n, st_count);
        if (k < n) { st_slot = src[k]; st_s0 = GST(gb, st_slot, 0); st_s1 = GST(gb, st_slot, 1); }
    }
    RTW_TM(const unsigned long long tm_start = __builtin_amdgcn_s_memtime();)
    for (;;) {
        RTW_TM(const unsigned long long tm_e0 = __builtin_amdgcn_s_memtime(); tm_events++;)
        // ---- hand the idle lanes new rays, out of the 64 the wave fetched ahead (lane e of the staging registers holds entry e) ----
        if (st_used < st_count) {
            const unsigned long long idle = __ballot(!have);
            const uint32_t c = (uint32_t)__popcll(idle);
            if (c != 0u) {
                const uint32_t avail = st_count - st_used, take = c < avail ? c : avail;
                const uint32_t rank = (uint32_t)mbcnt(idle);
                const int from = (int)((st_used + rank) & 63u) * 4;
                const uint32_t g_slot = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)st_slot);
                const float g0 = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(st_s0.x))), g1 = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(st_s0.y)));
                const float g2 = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(st_s0.z))), g3 = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(st_s0.w)));
                const float g4 = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(st_s1.x))), g5 = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(st_s1.y)));
                const float g6 = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(st_s1.z)));
                if (!have && rank < take) {
                    slot = g_slot;
                    r.o = mk(g0, g1, g2); r.dist = g3; r.d = mk(g4, g5, g6);
                    have = true;
                    if (STATS) { ct.rays++; ct.boxes++; }
                    tame = ray_is_tame(r);
                    skx = near_zero(r.d.x); sky = near_zero(r.d.y); skz = near_zero(r.d.z);
                    ix = (!tame && skx) ? 0.0f : 1.0f / r.d.x; iy = (!tame && sky) ? 0.0f : 1.0f / r.d.y; iz = (!tame && skz) ? 0.0f : 1.0f / r.d.z;
                    eps_t = 2.0e-5f * fmaxf(fabsf(ix), fmaxf(fabsf(iy), fabsf(iz)));
                    cur = r.dist; pos = mk(0, 0, 0); leaf_out = -1; ncand = 0; visits = 0;
                    if (LEAD) {         // FindIntersectionWithScene so far (Src/RayTracerScene.cpp:99-125): the segment is already shortened by the leading shapes' hits
                        const float4 h0 = GHT(gb, slot, 0), h1 = GHT(gb, slot, 1);
                        pos = mk(h0.x, h0.y, h0.z); cur = h0.w; lead_hs = __float_as_int(h1.x); lead_slot = __float_as_int(h1.y);
                    }
                    float t0, t1;       // the shape's culling box (Src/RayTracerScene.cpp:109)
                    i = slab_exact(r, sh.bmin[0], sh.bmin[1], sh.bmin[2], sh.bmax[0], sh.bmax[1], sh.bmax[2], t0, t1) ? 0 : n_nodes;
                }
                st_used += take;
                RTW_TM(tm_rays += take;)
                if (st_used == st_count) {          // fetch the next 64 now: they arrive while the wave walks
                    uint32_t batch;
                    if (DRAW) {
                        uint32_t j = 0u;
                        if (lane_id() == 0) j = atomicAdd(&blk_next, 1u);
                        j = (uint32_t)__builtin_amdgcn_readfirstlane((int)j);
                        batch = j < 0x00FFFFFFu ? j * gridDim.x + blockIdx.x : 0xFFFFFFFFu;        // (the counter cannot wrap: one failed draw per wave ends it)
                    } else {
                        batch = static_next; static_next += gridDim.x * (uint32_t)(NT / 64);
                    }
                    st_used = 0u;
                    const uint32_t k = batch_entry<SPREAD>(batch, n_batches, n, st_count);
                    if (k < n) { st_slot = src[k]; st_s0 = GST(gb, st_slot, 0); st_s1 = GST(gb, st_slot, 1); }
                }
            }
        }
        const bool exhausted = st_used >= st_count;
        RTW_TM(const unsigned long long tm_e1 = __builtin_amdgcn_s_memtime(); tm_refill += tm_e1 - tm_e0;)
        if (__ballot(have) == 0ull) break;
        const bool any_untame = __ballot(have && !tame) != 0ull;
        const float far_t = prune ? cur + (eps_t + 1.0e-4f * cur) : INFINITY;
        const float neg_eps = prune ? -eps_t : -INFINITY;
        // ---- walk ----
        if (!any_untame) {
            const rtw_v2f ox = { r.o.x, r.o.x }, oy = { r.o.y, r.o.y }, oz = { r.o.z, r.o.z };
            const rtw_v2f vx = { ix, ix }, vy = { iy, iy }, vz = { iz, iz };
            for (;;) {
                if (__ballot(have & (i < n_nodes)) == 0ull) break;
                            // RTW_GT_UNROLL visits between two looks at the wave (the three ballots, their scalar compares and branches: a CU has ONE scalar
                // unit for its sixteen waves); a lane whose list is full sits the extra visits out, its sequence of visits is the same
#pragma unroll
                for (int u = 0; u < RTW_GT_UNROLL; u++) {
                const bool walking = have & (i < n_nodes) & (ncand < CAP);
                RTW_TM(tm_wtrips++; tm_wlanes += (uint32_t)__popcll(__ballot(walking));)
                if (walking) {
                    float4 a, b;
                    if (ALLDS || i < ltop) { a = lld4(lnodes, 2 * i); b = lld4(lnodes, 2 * i + 1); }
                    else { a = gld4(nd4, 2 * (size_t)i); b = gld4(nd4, 2 * (size_t)i + 1); }
                    const int skip = __float_as_int(b.z), link = __float_as_int(b.w);
                    const rtw_v2f bx = { a.x, a.y }, by = { a.z, a.w }, bz = { b.x, b.y };
                    const rtw_v2f tx = (bx - ox) * vx, ty = (by - oy) * vy, tz = (bz - oz) * vz;
                    const float tmin = fmaxf(fmaxf(fminf(tx.x, tx.y), fminf(ty.x, ty.y)), fminf(tz.x, tz.y));
                    const float tmax = fminf(fminf(fmaxf(tx.x, tx.y), fmaxf(ty.x, ty.y)), fmaxf(tz.x, tz.y));
                    const bool hit = (tmax > tmin) & !(tmin > far_t) & !(tmax < neg_eps);
                    const bool leaf = link >= 0;
                    if (STATS) ct.boxes++;
                    if (hit & leaf) {
                        bool facing = true;
                        if (PLANES) { const float4 pl = lld4(lplanes, link); facing = !(dot(mk(pl.x, pl.y, pl.z), r.o) - pl.w < 0); }       // triangle_test's own d2 < 0 rejection, on the same values
                        if (facing) { lstu(cand, ncand * NT + tid, (uint32_t)link); ncand++; }
                    }
                    i = (hit & !leaf) ? ~link : skip;
                    visits++;
                }
                }
                if (__ballot(ncand == CAP) != 0ull) break;
                if (!exhausted && __popcll(__ballot(have & (i >= n_nodes))) >= RTW_GT_REFILL) break;
            }
        } else {
            for (;;) {
                const bool walking = have & (i < n_nodes);
                if (__ballot(walking) == 0ull) break;
                            if (walking) {
                    float4 a, b;
                    if (ALLDS || i < ltop) { a = lld4(lnodes, 2 * i); b = lld4(lnodes, 2 * i + 1); }
                    else { a = gld4(nd4, 2 * (size_t)i); b = gld4(nd4, 2 * (size_t)i + 1); }
                    const int skip = __float_as_int(b.z), link = __float_as_int(b.w);
                    const float x1 = (a.x - r.o.x) * ix, x2 = (a.y - r.o.x) * ix;
                    const float y1 = (a.z - r.o.y) * iy, y2 = (a.w - r.o.y) * iy;
                    const float z1 = (b.x - r.o.z) * iz, z2 = (b.y - r.o.z) * iz;
                    const float tmin = fmaxf(fmaxf(fminf(x1, x2), fminf(y1, y2)), fminf(z1, z2));
                    const float tmax = fminf(fminf(fmaxf(x1, x2), fmaxf(y1, y2)), fmaxf(z1, z2));
                    bool hit = (tmax > tmin) & !(tmin > far_t) & !(tmax < neg_eps);
                    float emin = -FLT_MAX, emax = FLT_MAX;      // RRay::TestIntersectionWithAabb as written, for the lanes that need it
                    if (!skx) { emin = ref_max(emin, ref_min(x1, x2)); emax = ref_min(emax, ref_max(x1, x2)); }
                    if (!sky) { emin = ref_max(emin, ref_min(y1, y2)); emax = ref_min(emax, ref_max(y1, y2)); }
                    if (!skz) { emin = ref_max(emin, ref_min(z1, z2)); emax = ref_min(emax, ref_max(z1, z2)); }
                    if (!tame) hit = emax > emin;
                    const bool leaf = link >= 0;
                    if (STATS) ct.boxes++;
                    if (hit & leaf) { lstu(cand, ncand * NT + tid, (uint32_t)link); ncand++; }
                    i = (hit & !leaf) ? ~link : skip;
                    visits++;
                }
                if (__ballot(ncand == CAP) != 0ull) break;
                if (!exhausted && __popcll(__ballot(have & (i >= n_nodes))) >= RTW_GT_REFILL) break;
            }
        }
        RTW_TM(const unsigned long long tm_e2 = __builtin_amdgcn_s_memtime(); tm_walk += tm_e2 - tm_e1;)
        // ---- a ray that has used up its budget of node visits leaves for the wave-per-ray kernel that follows (it starts over there:
        // a whole wave on one ray shortens the chain of dependent steps that would otherwise keep this launch waiting) ----
        {
            const bool over = have & (i < n_nodes) & (visits > budget);
            const unsigned long long om = __ballot(over);
            if (om != 0ull) {
                uint32_t base = 0u;
                if (lane_id() == 0) base = atomicAdd(&gb.counters[40 + round], (uint32_t)__popcll(om));
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                if (over) {
                    gb.overflow[base + (uint32_t)mbcnt(om)] = slot;
                    if (STATS) ct.rays--;        // the query is counted by the kernel that takes it over (the box / triangle tests run so far were run)
                    // (LEAD: the hit record still holds the leading shapes' result; the wave-per-ray kernel continues from it)
                    have = false; ncand = 0; i = n_nodes;
                }
            }
        }
        // ---- the noted leaves' triangle tests: every lane through its own list in order, the next record fetched ahead ----
        {
            bool mine = 0 < ncand;
            int leaf = mine ? (int)lldu(cand, tid) : 0;
            // (the record's d1 = dot(N, p0), by the same float operations on the device and on the host: computed here, one load a test less: C4 -1.5 %, C5 -0.8 %)
            float4 ta = gld4(tr4, 4 * (size_t)leaf), tb = gld4(tr4, 4 * (size_t)leaf + 1), tc = gld4(tr4, 4 * (size_t)leaf + 2);
            for (int j = 0; __ballot(mine) != 0ull; j++) {
                RTW_TM(tm_ttrips++; tm_tlanes += (uint32_t)__popcll(__ballot(mine));)
                const bool nmine = j + 1 < ncand;
                const int nleaf = nmine ? (int)lldu(cand, (j + 1) * NT + tid) : 0;
                const float4 na = gld4(tr4, 4 * (size_t)nleaf), nb = gld4(tr4, 4 * (size_t)nleaf + 1), nc = gld4(tr4, 4 * (size_t)nleaf + 2);
                if (mine) {
                    if (STATS) ct.tris++;
                    f3 cp; float dist;
                    const float td = dot(mk(ta.w, tb.w, tc.w), mk(ta.x, ta.y, ta.z));
                    if (triangle_test(r, cur, ta, tb, tc, td, cp, dist)) { cur = dist; pos = cp; leaf_out = leaf; }
                }
                mine = nmine; leaf = nleaf; ta = na; tb = nb; tc = nc;
            }
            ncand = 0;
        }
        // ---- lanes whose walk is complete: FindIntersectionWithScene's result for the one shape ----
        if (have & (i >= n_nodes)) {
            const int hs = leaf_out >= 0 ? (LEAD ? ms : 0) : (LEAD ? lead_hs : -1);
            GHT(gb, slot, 0) = make_float4(pos.x, pos.y, pos.z, cur);
            GHT(gb, slot, 1) = make_float4(__int_as_float(hs), __int_as_float(leaf_out >= 0 ? leaf_out : (LEAD ? lead_slot : -1)), __int_as_float(-1), __int_as_float(-1));
            have = false;
        }
        RTW_TM(tm_tri += __builtin_amdgcn_s_memtime() - tm_e2;)
    }
#ifdef RTW_TIMING
    {
        const unsigned w = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
        if ((threadIdx.x & 63u) == 0 && w < 16384 && round == RTW_TIMING) {        // -DRTW_TIMING=<round + 0>: which trace round files its waves
            unsigned long long* o = &g_rtw_timing[12 * w];
            o[0] = tm_entry; o[1] = wall_clock64(); o[2] = tm_wtrips; o[3] = tm_wlanes; o[4] = tm_ttrips; o[5] = tm_tlanes; o[6] = tm_events; o[7] = tm_walk; o[8] = tm_tri; o[9] = tm_refill;
            o[10] = tm_rays; o[11] = __builtin_amdgcn_s_memtime() - tm_start;
        }
    }
#endif
    if (STATS) flush_counters(sc, ct);
}

// The same round for a SHORT list: a whole wave per ray over the flat hierarchy (wave_find_intersection, rtw_wave_kernels.h).  A ray
// per lane needs hundreds of thousands of rays to fill the chip and ~100 dependent steps per ray; below that the wave-per-ray walk's
// handful of dependent steps wins (the host picks by the previous group's list length).  Not for texel-inheritance scenes.
// from_overflow: the list is the one the budgeted ray-per-lane kernel of this round filled.
template <bool STATS, bool AN, int NT>
__global__ __launch_bounds__(NT) void gtrace_wave_kernel(const RtwSceneDev* __restrict__ sc, GroupBufs gb, int round, int from_overflow, int lead)
{
    HIP_DYNAMIC_SHARED(uint32_t, wave_dyn);              // [NT / 64 waves x RTW_WAVE_LDS_WORDS]
    const uint32_t nl = from_overflow ? gb.counters[40 + round] : (lead > 0 ? gb.counters[24 + round] : gb.counters[round]);
    const uint32_t n = nl < gb.capacity ? nl : gb.capacity;
    if ((uint32_t)blockIdx.x * (uint32_t)(NT / 64) >= n) return;
    uint32_t* lds = wave_dyn + (threadIdx.x >> 6) * RTW_WAVE_LDS_WORDS;
    const FlatSrc shape0 = flat_src_of(sc->shapes[0]);
    const uint32_t* __restrict__ src = from_overflow ? gb.overflow : (lead > 0 ? (round & 1 ? gb.tlist1 : gb.tlist0) : (round & 1 ? gb.list1 : gb.list0));
    const int n_shapes = sc->n_shapes;
    const bool prune = sc->prune != 0;
    const uint32_t wave = (blockIdx.x * (uint32_t)NT + threadIdx.x) >> 6, nwaves = gridDim.x * (uint32_t)(NT / 64);
    Counters ct = { 0, 0, 0, 0, 0, 0 };
    for (uint32_t k = wave; k < n; k += nwaves) {
        const int ku = __builtin_amdgcn_readfirstlane((int)k);
        const int q = (int)cldu(src, ku);
        const float4 s0 = GST(gb, q, 0), s1 = GST(gb, q, 1);
        Ray ray; ray.o = mk(s0.x, s0.y, s0.z); ray.dist = s0.w; ray.d = mk(s1.x, s1.y, s1.z);
        int hs = -1, slot = -1; f3 pos = mk(0, 0, 0); float seg = ray.dist;
        if (AN && lead > 0) {
            const float4 h0 = GHT(gb, q, 0), h1 = GHT(gb, q, 1);
            pos = mk(h0.x, h0.y, h0.z); seg = h0.w; hs = __float_as_int(h1.x); slot = __float_as_int(h1.y);
        }
        wave_find_intersection<STATS, AN>(sc, AN ? lead : 0, n_shapes, prune, shape0, lds, ray, hs, slot, pos, seg, ct);
        if (lane_id() == 0) {
            GHT(gb, q, 0) = make_float4(pos.x, pos.y, pos.z, seg);
            GHT(gb, q, 1) = make_float4(__int_as_float(hs), __int_as_float(slot), __int_as_float(-1), __int_as_float(-1));
        }
    }
    if (STATS) flush_counters(sc, ct);
}

// ---- shading of one round's hits: a path per lane ---------------------------------------------------------------------------------
template <bool STATS, bool AN>
#ifndef RTW_GSHADE_MINB
#define RTW_GSHADE_MINB 3
#endif
__global__ __launch_bounds__(256, RTW_GSHADE_MINB) void gshade_kernel(const RtwSceneDev* __restrict__ sc, GroupBufs gb, RtwGroupParams g, int round)
{
    // round >= 1: the paths of list (round - 1) have had their segment traced; the ones that go on join list `round`
    const RtwRenderParams& p = g.rp;
    const uint32_t n = gb.counters[round - 1] < gb.capacity ? gb.counters[round - 1] : gb.capacity;
    const uint32_t* __restrict__ src = (round - 1) & 1 ? gb.list1 : gb.list0;
    uint32_t* __restrict__ dst = round & 1 ? gb.list1 : gb.list0;
    const uint32_t nthreads = gridDim.x * blockDim.x;
    const int npix = p.width * p.height;
    const uint32_t phase = table_phase(p.seed);
    __shared__ uint32_t part[5];
    Counters ct = { 0, 0, 0, 0, 0, 0 };
    const uint32_t trips = (n + nthreads - 1) / nthreads;        // grid-uniform trip count: every thread joins the pushes
    for (uint32_t it = 0, k = blockIdx.x * blockDim.x + threadIdx.x; it < trips; it++, k += nthreads) {
        const bool live = k < n;
        const uint32_t slot = live ? src[k] : 0u;
        bool go_on = false, trace_on = false;
        if (live && slot < gb.capacity) {
            const float4 s0 = GST(gb, slot, 0), s1 = GST(gb, slot, 1), s2 = GST(gb, slot, 2);
            Ray ray; ray.o = mk(s0.x, s0.y, s0.z); ray.dist = s0.w; ray.d = mk(s1.x, s1.y, s1.z);
            const int pixel = __float_as_int(s2.x);
            const uint32_t pass_sub = __float_as_uint(s2.w);
            const uint32_t pass = (uint32_t)g.first_pass + (pass_sub >> 2), sub = pass_sub & 3u;
            PathRng rng;
            rng.key = stream_key(p.seed, (uint32_t)pixel, pass * 4u + sub);
            rng.counter = __float_as_uint(s1.w); rng.table_reads = __float_as_uint(s2.y);
            rng.table_base = (((uint64_t)pass * (uint64_t)npix + (uint64_t)pixel) * 4u + (uint64_t)sub) * RTW_TABLE_STRIDE + phase;
            rng.pre_reads = 0xFFFFFFFFu; rng.pre_x = rng.pre_y = rng.pre_z = 0.0f;
            int nlev = (int)(__float_as_uint(s2.z) & 0xFFFFu), depth = (int)(__float_as_uint(s2.z) >> 16);
            const float4 r0 = GHT(gb, slot, 0), r1 = GHT(gb, slot, 1);
            float4 r2 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (AN && gb.carry_on) r2 = gb.carry[slot];
            f3 L;
            go_on = group_shade_step<STATS, AN>(sc, p, gb, slot, ray, rng, depth, nlev, r0, r1, r2, L, ct);
            if (go_on) {
                group_save_state(gb, slot, ray, rng, depth, nlev, pixel, pass_sub);
                if (AN && p.lead_shapes > 0) {
                    float4 q0, q1;
                    trace_on = group_lead_query<STATS>(sc, p.lead_shapes, ray, q0, q1, ct);
                    GHT(gb, slot, 0) = q0; GHT(gb, slot, 1) = q1;
                }
            } else { const f3 c = mk(0, 0, 0) + L; gb.rad[slot] = make_float4(c.x, c.y, c.z, 0.0f); }
        }
        block_push<4>(dst, &gb.counters[round], go_on, slot, part);        // (the trip count is the same for every thread of the grid)
        if (AN && p.lead_shapes > 0) block_push<4>(round & 1 ? gb.tlist1 : gb.tlist0, &gb.counters[24 + round], trace_on, slot, part);
    }
    if (STATS) flush_counters(sc, ct);
}

// ---- the group's pixels of the busy tiles: accumulate + resolve, pass by pass ---------------------------------------------------
__global__ __launch_bounds__(256) void gresolve_kernel(const RtwSceneDev* __restrict__ sc, float4* __restrict__ accum, uint32_t* __restrict__ argb,
                                                       GroupBufs gb, RtwGroupParams g)
{
    __shared__ float thr[256];
    thr[threadIdx.x] = sc->gamma_thr[threadIdx.x];
    __syncthreads();
    const RtwRenderParams& p = g.rp;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t b = t >> 6, lane = t & 63u;
    if (b < (uint32_t)g.n_busy) {
        const int wt = g.busy_tiles ? (int)g.busy_tiles[b] : g.first_tile + (int)b;
        int px = 0, py = 0;
        if (group_xy(g, wt, (int)lane, px, py)) {
            const int pixel = py * p.width + px;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!p.preview) acc = accum[pixel];
            f3 sum = mk(acc.x, acc.y, acc.z);
            int n = __float_as_int(acc.w);
            f3 c = mk(0, 0, 0);
            for (int k = 0; k < g.n_passes; k++) {
                c = mk(0, 0, 0);
                for (int i = 0; i < p.sub_samples; i++) {
                    const float4 r = gb.rad[group_slot(g, b, lane, (uint32_t)i, (uint32_t)k)];
                    c = c + mk(r.x, r.y, r.z);
                }
                c = c / (float)p.sub_samples;
                if (!p.preview) { sum = sum + c; n++; }                                    // AddPixel, in pass order
            }
            argb[pixel] = pack_pixel(thr, p.preview ? c : (n == 1 ? sum : sum / (float)n));       // GetGammaSpacePixel after the group's last pass (see gsky_kernel)
            if (!p.preview) accum[pixel] = make_float4(sum.x, sum.y, sum.z, __int_as_float(n));
        }
    }
    // the group's last kernel: file the list lengths for the host (they size the next group's launches) and zero them
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        gb.counters[64 + threadIdx.x] = gb.counters[threadIdx.x];
        gb.counters[threadIdx.x] = 0u;
    }
}
