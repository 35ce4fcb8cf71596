// rtw_build_kernels.h -- KdNode::Build (Src/KdTree.cpp:37-126) on the device, with the reference's split decisions, and the layouts
// derived from the tree (leaf records, the explicit-link copy, the flat hierarchy).  Included inside rtw_device.hip's anonymous namespace.
//
// The reference's recursion, level by level: a node is its segment [start, start + count) of the triangle order.  One wave per node:
//   bounds      min / max over the segment's vertices (order-free, exact), through the triangles' own boxes computed once
//   mid point   sum of (v0 + v1 + v2) / 3.0f over the segment IN LIST ORDER, fp32, then / count: 64 centroids at a time go through LDS and
//               one dependent chain of adds per axis folds them in order, exactly the reference's sequence
//   axis        GetLargestAxisOfBounds' `>` cascade (ties go to Z, then Y)
//   partition   stable: left iff centroid[axis] < mid[axis] (strict), both sides keep the list order (ballot + prefix counts);
//               a one-sided split becomes first half / second half of the current order
// A subtree over n triangles holds 2 n - 1 nodes, so preorder numbers need no second pass: left child = i + 1, right child =
// i + 2 n_left, skip = i + 2 n - 1; the stable partitions leave the triangles in preorder leaf order, so a leaf's slot is its segment start.

struct BuildNode { int32_t start, count, index, depth; };

__device__ __forceinline__ float wave_min(float v) { for (int o = 32; o > 0; o >>= 1) { const float w = __shfl_xor(v, o); v = w < v ? w : v; } return v; }
__device__ __forceinline__ float wave_max(float v) { for (int o = 32; o > 0; o >>= 1) { const float w = __shfl_xor(v, o); v = w > v ? w : v; } return v; }

__device__ __forceinline__ f3 build_centroid(const float* __restrict__ pts, const int32_t* __restrict__ idx, int tri)
{
    const int a = idx[tri * 3], b = idx[tri * 3 + 1], c = idx[tri * 3 + 2];
    const f3 pa = mk(pts[a * 3], pts[a * 3 + 1], pts[a * 3 + 2]), pb = mk(pts[b * 3], pts[b * 3 + 1], pts[b * 3 + 2]), pc = mk(pts[c * 3], pts[c * 3 + 1], pts[c * 3 + 2]);
    const f3 s = (pa + pb) + pc;
    return mk(s.x / 3.0f, s.y / 3.0f, s.z / 3.0f);
}

// (the zero a box bound shows: +0 and -0 compare equal, the reference's sequential Expand keeps whichever it met FIRST -- triangle by triangle,
// p0 p1 p2 -- see build_first_zero_rec)

// Per-triangle data the levels read over and over, computed once: the centroid (v0 + v1 + v2) / 3.0f (the reference's operations, so its bits) and
// the triangle's own box.  The level kernel PERMUTES these records along with the triangle order (ping-pong buffers), so that a node's segment
// is contiguous memory: a node with 16 000 triangles is one wave, and with the three-level gather order -> index -> point of the first version
// it spent ~2 us of latency per 64 triangles and pass (3.7 ms for unitychan's 24 levels; 1 ms with the records carried along).
struct BuildTri { float cx, cy, cz; int32_t tri; float lox, loy, loz, pad0; float hix, hiy, hiz, pad1; };      // 48 B
__global__ __launch_bounds__(256) void build_tri_prep_kernel(const float* __restrict__ pts, const int32_t* __restrict__ idx, int n, BuildTri* __restrict__ out)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const f3 c = build_centroid(pts, idx, t);
    BuildTri r;
    r.cx = c.x; r.cy = c.y; r.cz = c.z; r.tri = t;
    float lo[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, hi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    for (int k = 0; k < 3; k++) {
        const int v = idx[t * 3 + k];
        for (int a = 0; a < 3; a++) { const float x = pts[v * 3 + a]; if (x < lo[a]) lo[a] = x; if (x > hi[a]) hi[a] = x; }
    }
    r.lox = lo[0]; r.loy = lo[1]; r.loz = lo[2]; r.pad0 = 0.0f; r.hix = hi[0]; r.hiy = hi[1]; r.hiz = hi[2]; r.pad1 = 0.0f;
    out[t] = r;
}
// the zero (its sign) the segment's vertices show first on axis `a`, in the order the reference expands its box (whole wave; only called when the
// box's bound on that axis is a zero)
__device__ __forceinline__ float build_first_zero_rec(const float* __restrict__ pts, const int32_t* __restrict__ idx, const BuildTri* __restrict__ recs, int start, int n, int a)
{
    int first = 0x7FFFFFFF;
    for (int t = lane_id(); t < n && first == 0x7FFFFFFF; t += 64) {
        const int tri = recs[start + t].tri;
        for (int k = 0; k < 3; k++) if (pts[idx[tri * 3 + k] * 3 + a] == 0.0f) { first = t * 3 + k; break; }
    }
    for (int o = 32; o > 0; o >>= 1) { const int w = __shfl_xor(first, o); first = w < first ? w : first; }
    if (first == 0x7FFFFFFF) return 0.0f;
    return pts[idx[recs[start + first / 3].tri * 3 + first % 3] * 3 + a];
}

// One node of a level: its box, its split, its children.  NW = 1: one wave does it all.  NW = 4: the four waves of the block share a BIG node --
// each takes a quarter of the segment for the order-free parts (box: min / max; the count of triangles going left; the stable scatter, whose
// quarters stay in order), wave 0 alone folds the centroids (that chain is sequential by definition) -- because the top of the tree is a handful
// of nodes with thousands of triangles each, and one wave per node left the rest of the chip idle for most of the build's time.
#define RTW_BUILD_BIG 2048
template <int NW>
__device__ __forceinline__ void build_split_node(const float* __restrict__ pts, const int32_t* __restrict__ idx, const BuildTri* __restrict__ src, BuildTri* __restrict__ dst,
                                                 int32_t* __restrict__ leaf_order, const BuildNode nd, BuildNode* __restrict__ next, uint32_t* __restrict__ n_next,
                                                 RtwNode* __restrict__ nodes, int32_t* __restrict__ node_depth, uint32_t* __restrict__ level_count,
                                                 float* cb, float* part_f, int* part_i, int q)
{
    const int lane = lane_id();
    const int start = nd.start, n = nd.count;
    const float4* __restrict__ s4 = reinterpret_cast<const float4*>(src + start);      // record t: s4[3 t] = centroid | tri, s4[3 t + 1] = lo, s4[3 t + 2] = hi
    // this wave's part of the segment (whole 64s)
    const int per = NW == 1 ? n : ((n + 64 * NW - 1) / (64 * NW)) * 64;
    const int ps = NW == 1 ? 0 : (per * q < n ? per * q : n), pe = NW == 1 ? n : (ps + per < n ? ps + per : n);
    // ---- bounds (RAabb::Expand over every vertex of the segment = over the triangles' own boxes; min / max are order-free) ----
    float lox = FLT_MAX, loy = FLT_MAX, loz = FLT_MAX, hix = -FLT_MAX, hiy = -FLT_MAX, hiz = -FLT_MAX;
    for (int t = ps + lane; t < pe; t += 256) {            // four independent loads in flight per lane
        float4 lo[4], hi[4];
        for (int u = 0; u < 4; u++) {
            const int tt = t + 64 * u;
            const bool in = tt < pe;
            lo[u] = in ? s4[3 * tt + 1] : make_float4(FLT_MAX, FLT_MAX, FLT_MAX, 0.f);
            hi[u] = in ? s4[3 * tt + 2] : make_float4(-FLT_MAX, -FLT_MAX, -FLT_MAX, 0.f);
        }
        for (int u = 0; u < 4; u++) {
            if (lo[u].x < lox) lox = lo[u].x; if (lo[u].y < loy) loy = lo[u].y; if (lo[u].z < loz) loz = lo[u].z;
            if (hi[u].x > hix) hix = hi[u].x; if (hi[u].y > hiy) hiy = hi[u].y; if (hi[u].z > hiz) hiz = hi[u].z;
        }
    }
    lox = wave_min(lox); loy = wave_min(loy); loz = wave_min(loz); hix = wave_max(hix); hiy = wave_max(hiy); hiz = wave_max(hiz);
    if (NW > 1) {
        if (lane == 0) { float* o = part_f + 6 * q; o[0] = lox; o[1] = loy; o[2] = loz; o[3] = hix; o[4] = hiy; o[5] = hiz; }
        __syncthreads();
        for (int k = 0; k < NW; k++) {
            const float* o = part_f + 6 * k;
            if (o[0] < lox) lox = o[0]; if (o[1] < loy) loy = o[1]; if (o[2] < loz) loz = o[2];
            if (o[3] > hix) hix = o[3]; if (o[4] > hiy) hiy = o[4]; if (o[5] > hiz) hiz = o[5];
        }
        __syncthreads();
    }
    // +0 and -0 compare equal: the reference's sequential Expand keeps whichever zero it met FIRST; the parallel reduction may hold the other
    if (lox == 0.0f) lox = build_first_zero_rec(pts, idx, src, start, n, 0);
    if (loy == 0.0f) loy = build_first_zero_rec(pts, idx, src, start, n, 1);
    if (loz == 0.0f) loz = build_first_zero_rec(pts, idx, src, start, n, 2);
    if (hix == 0.0f) hix = build_first_zero_rec(pts, idx, src, start, n, 0);
    if (hiy == 0.0f) hiy = build_first_zero_rec(pts, idx, src, start, n, 1);
    if (hiz == 0.0f) hiz = build_first_zero_rec(pts, idx, src, start, n, 2);
    RtwNode rec;
    rec.min_x = lox; rec.min_y = loy; rec.min_z = loz; rec.max_x = hix; rec.max_y = hiy; rec.max_z = hiz;
    rec.skip = nd.index + 2 * n - 1;
    if (n == 1) {                    // a leaf: its slot in leaf order is its place in the triangle order  (never a shared node)
        const int tri = __float_as_int(s4[0].w);
        rec.tri = start;
        if (lane == 0) { nodes[nd.index] = rec; node_depth[nd.index] = nd.depth; leaf_order[start] = tri; }
        return;
    }
    rec.tri = -1;
    // ---- NodeMidPoint: the centroids summed in list order (one chain of dependent adds per axis), then / NumTriangles; the next 64 centroids
    // are fetched while the chain folds the current ones (lanes 0, 1, 2 fold x, y, z out of LDS) ----
    float mx = 0.0f, my = 0.0f, mz = 0.0f;
    if (NW == 1 || q == 0) {
        float m = 0.0f;
        float4 cn = lane < n ? s4[3 * lane] : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int t0 = 0; t0 < n; t0 += 64) {
            const int cnt = n - t0 < 64 ? n - t0 : 64;
            const float4 c = cn;
            const int tn = t0 + 64 + lane;
            cn = tn < n ? s4[3 * tn] : make_float4(0.f, 0.f, 0.f, 0.f);
            cb[lane * 3] = c.x; cb[lane * 3 + 1] = c.y; cb[lane * 3 + 2] = c.z;
            wave_lds_sync();
            if (lane < 3) {
                for (int j0 = 0; j0 < cnt; j0 += 16) {      // sixteen LDS reads in flight, then their sixteen dependent adds (a read per add waited ~70 cycles each)
                    float v[16];
#pragma unroll
                    for (int u = 0; u < 16; u++) v[u] = cb[(j0 + u) * 3 + lane];
#pragma unroll
                    for (int u = 0; u < 16; u++) if (j0 + u < cnt) m = m + v[u];
                }
            }
            wave_lds_sync();
        }
        const float fn = (float)n;
        mx = readlane_f(m, 0) / fn; my = readlane_f(m, 1) / fn; mz = readlane_f(m, 2) / fn;
        if (NW > 1 && lane == 0) { part_f[0] = mx; part_f[1] = my; part_f[2] = mz; }
    }
    if (NW > 1) { __syncthreads(); mx = part_f[0]; my = part_f[1]; mz = part_f[2]; __syncthreads(); }
    // ---- GetLargestAxisOfBounds ----
    const float sx = hix - lox, sy = hiy - loy, sz = hiz - loz;
    const int axis = sx > sy ? (sx > sz ? 0 : 2) : (sy > sz ? 1 : 2);
    const float cut = axis == 0 ? mx : (axis == 1 ? my : mz);
    // ---- how many go left (in this wave's part) ----
    int my_left = 0;
    for (int t0 = ps; t0 < pe; t0 += 256) {
        float4 c[4];
        for (int u = 0; u < 4; u++) { const int tt = t0 + 64 * u + lane; c[u] = tt < pe ? s4[3 * tt] : make_float4(0.f, 0.f, 0.f, 0.f); }
        for (int u = 0; u < 4; u++) {
            const int tt = t0 + 64 * u + lane;
            const float v = axis == 0 ? c[u].x : (axis == 1 ? c[u].y : c[u].z);
            my_left += (int)__popcll(__ballot(tt < pe && v < cut));
        }
    }
    int n_left = my_left, left_before = 0;          // left_before: lefts in the parts before this wave's
    if (NW > 1) {
        if (lane == 0) part_i[q] = my_left;
        __syncthreads();
        n_left = 0;
        for (int k = 0; k < NW; k++) { if (k < q) left_before += part_i[k]; n_left += part_i[k]; }
        __syncthreads();
    }
    const bool one_sided = n_left == 0 || n_left == n;
    if (one_sided) { n_left = n / 2; left_before = ps < n_left ? ps : n_left; }
    const int right_before = ps - left_before;
    // ---- the two sides, each in list order: the records move with their triangles ----
    float4* __restrict__ d4 = reinterpret_cast<float4*>(dst + start);
    int done_l = 0, done_r = 0;
    for (int t0 = ps; t0 < pe; t0 += 128) {
        float4 r0[2], r1[2], r2[2];
        for (int u = 0; u < 2; u++) {
            const int tt = t0 + 64 * u + lane;
            if (tt < pe) { r0[u] = s4[3 * tt]; r1[u] = s4[3 * tt + 1]; r2[u] = s4[3 * tt + 2]; }
            else { r0[u] = make_float4(0.f, 0.f, 0.f, 0.f); r1[u] = r0[u]; r2[u] = r0[u]; }
        }
        for (int u = 0; u < 2; u++) {
            const int tt = t0 + 64 * u + lane;
            const bool mine = tt < pe;
            bool left = false;
            if (mine) {
                if (one_sided) left = tt < n_left;
                else { const float v = axis == 0 ? r0[u].x : (axis == 1 ? r0[u].y : r0[u].z); left = v < cut; }
            }
            const unsigned long long ml = __ballot(mine && left), mr = __ballot(mine && !left);
            if (mine) {
                const int at = left ? left_before + done_l + mbcnt(ml) : n_left + right_before + done_r + mbcnt(mr);
                d4[3 * at] = r0[u]; d4[3 * at + 1] = r1[u]; d4[3 * at + 2] = r2[u];
            }
            done_l += (int)__popcll(ml); done_r += (int)__popcll(mr);
        }
    }
    if (lane == 0 && (NW == 1 || q == 0)) {
        nodes[nd.index] = rec; node_depth[nd.index] = nd.depth;
        const uint32_t at = atomicAdd(n_next, 2u);
        BuildNode l, r;
        l.start = start; l.count = n_left; l.index = nd.index + 1; l.depth = nd.depth + 1;
        r.start = start + n_left; r.count = n - n_left; r.index = nd.index + 2 * n_left; r.depth = nd.depth + 1;
        next[at] = l; next[at + 1] = r;
        atomicAdd(&level_count[nd.depth + 1 < 63 ? nd.depth + 1 : 63], 2u);
    }
}

// one level of the recursion: the nodes of cur[0 .. *n_cur) are split, their children appended to next[].  A block takes four nodes: its waves one
// each, except that a node of RTW_BUILD_BIG triangles or more is shared by the whole block (block-uniform branch: every thread reads the same record)
__global__ __launch_bounds__(256) void build_level_kernel(const float* __restrict__ pts, const int32_t* __restrict__ idx,
                                                          const BuildTri* __restrict__ src, BuildTri* __restrict__ dst, int32_t* __restrict__ leaf_order,
                                                          const BuildNode* __restrict__ cur, const uint32_t* __restrict__ n_cur, BuildNode* __restrict__ next, uint32_t* __restrict__ n_next,
                                                          RtwNode* __restrict__ nodes, int32_t* __restrict__ node_depth, uint32_t* __restrict__ level_count, int block_per_node)
{
    __shared__ float cbuf[4][192];
    __shared__ float part_f[24];
    __shared__ int part_i[4];
    const int q = (int)(threadIdx.x >> 6);
    float* cb = cbuf[q];
    const uint32_t nn = *n_cur;
    if (block_per_node) {       // the first levels: few nodes, big ones -- a block each, so that the big nodes of a level run side by side
        for (uint32_t w = blockIdx.x; w < nn; w += gridDim.x) {
            const BuildNode nd = cur[w];
            if (nd.count >= RTW_BUILD_BIG) build_split_node<4>(pts, idx, src, dst, leaf_order, nd, next, n_next, nodes, node_depth, level_count, cb, part_f, part_i, q);
            else if (q == 0) build_split_node<1>(pts, idx, src, dst, leaf_order, nd, next, n_next, nodes, node_depth, level_count, cb, part_f, part_i, 0);
        }
        return;
    }
    for (uint32_t g0 = blockIdx.x * 4u; g0 < nn; g0 += gridDim.x * 4u) {
        for (uint32_t i = 0; i < 4u && g0 + i < nn; i++) {
            const BuildNode nd = cur[g0 + i];
            if (nd.count >= RTW_BUILD_BIG) build_split_node<4>(pts, idx, src, dst, leaf_order, nd, next, n_next, nodes, node_depth, level_count, cb, part_f, part_i, q);
        }
        const uint32_t w = g0 + (uint32_t)q;
        if (w < nn) {
            const BuildNode nd = cur[w];
            if (nd.count < RTW_BUILD_BIG) build_split_node<1>(pts, idx, src, dst, leaf_order, nd, next, n_next, nodes, node_depth, level_count, cb, part_f, part_i, 0);
        }
    }
}

// leaf records in leaf order: the triangle with the face normal RRay::TestIntersectionWithTriangle recomputes at every test
// (Src/RRay.cpp:138-145: normalize(cross(p1 - p0, p2 - p0)), tiny vectors left as they are) and the shading inputs
// RMeshShape::TestRayIntersection gathers per hit (Src/MeshShape.cpp:303-326)
__global__ void build_leaf_records_kernel(const float* __restrict__ pts, const float* __restrict__ tcs, const float* __restrict__ nrm,
                                          const int32_t* __restrict__ idx_p, const int32_t* __restrict__ idx_t, const int32_t* __restrict__ idx_n,
                                          const int32_t* __restrict__ mat, const int32_t* __restrict__ leaf_order, int n_tris, RtwTri* __restrict__ tris, RtwShade* __restrict__ shade)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_tris) return;
    const int t = leaf_order[s];
    const int a = idx_p[t * 3], b = idx_p[t * 3 + 1], c = idx_p[t * 3 + 2];
    const f3 p0 = mk(pts[a * 3], pts[a * 3 + 1], pts[a * 3 + 2]), p1 = mk(pts[b * 3], pts[b * 3 + 1], pts[b * 3 + 2]), p2 = mk(pts[c * 3], pts[c * 3 + 1], pts[c * 3 + 2]);
    const f3 n = normalized(cross(p1 - p0, p2 - p0));
    RtwTri r;
    r.p0x = p0.x; r.p0y = p0.y; r.p0z = p0.z; r.nx = n.x;
    r.p1x = p1.x; r.p1y = p1.y; r.p1z = p1.z; r.ny = n.y;
    r.p2x = p2.x; r.p2y = p2.y; r.p2z = p2.z; r.nz = n.z;
    r.d1 = dot(n, p0); r.orig = t; r.pad0 = r.pad1 = 0;
    tris[s] = r;
    RtwShade sh;
    const int na = idx_n[t * 3], nb = idx_n[t * 3 + 1], nc = idx_n[t * 3 + 2], ta = idx_t[t * 3], tb = idx_t[t * 3 + 1], tc = idx_t[t * 3 + 2];
    sh.n0x = nrm[na * 3]; sh.n0y = nrm[na * 3 + 1]; sh.n0z = nrm[na * 3 + 2];
    sh.n1x = nrm[nb * 3]; sh.n1y = nrm[nb * 3 + 1]; sh.n1z = nrm[nb * 3 + 2];
    sh.n2x = nrm[nc * 3]; sh.n2y = nrm[nc * 3 + 1]; sh.n2z = nrm[nc * 3 + 2];
    sh.u0 = tcs[ta * 3]; sh.v0 = tcs[ta * 3 + 1]; sh.u1 = tcs[tb * 3]; sh.v1 = tcs[tb * 3 + 1]; sh.u2 = tcs[tc * 3]; sh.v2 = tcs[tc * 3 + 1];
    sh.material = mat[t];
    shade[s] = sh;
}

// place[i] of every node in the explicit-link array: the nodes of depth <= D first (preorder), the deeper ones after them (preorder);
// one block scans the two flags over the preorder index
__global__ __launch_bounds__(1024) void build_tnode_places_kernel(const int32_t* __restrict__ node_depth, int n_nodes, int depth_limit, int32_t* __restrict__ place, int32_t* __restrict__ n_top_out)
{
    __shared__ int part[1024];
    __shared__ int n_top_sh;
    // first the number of top nodes
    int mine = 0;
    for (int i = threadIdx.x; i < n_nodes; i += 1024) mine += node_depth[i] <= depth_limit ? 1 : 0;
    part[threadIdx.x] = mine;
    __syncthreads();
    if (threadIdx.x == 0) { int s = 0; for (int k = 0; k < 1024; k++) s += part[k]; n_top_sh = s; *n_top_out = s; }
    __syncthreads();
    const int n_top = n_top_sh;
    // then both running counts over contiguous chunks of the preorder index
    const int per = (n_nodes + 1023) / 1024;
    const int lo = threadIdx.x * per, hi = lo + per < n_nodes ? lo + per : n_nodes;
    int tops = 0;
    for (int i = lo; i < hi; i++) tops += node_depth[i] <= depth_limit ? 1 : 0;
    __syncthreads();
    part[threadIdx.x] = tops;
    __syncthreads();
    if (threadIdx.x == 0) { int s = 0; for (int k = 0; k < 1024; k++) { const int v = part[k]; part[k] = s; s += v; } }
    __syncthreads();
    int t = part[threadIdx.x];            // top nodes before this chunk
    for (int i = lo; i < hi; i++) {
        const bool top = node_depth[i] <= depth_limit;
        place[i] = top ? t : n_top + (i - t);       // deep nodes before i = i - (top nodes before i)
        t += top ? 1 : 0;
    }
    if (threadIdx.x == 0) place[n_nodes] = n_nodes;
}

__global__ void build_tnodes_kernel(const RtwNode* __restrict__ nodes, const int32_t* __restrict__ place, int n_nodes, RtwPNode* __restrict__ tnodes)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const RtwNode s = nodes[i];
    RtwPNode t;
    t.min_x = s.min_x; t.max_x = s.max_x; t.min_y = s.min_y; t.max_y = s.max_y; t.min_z = s.min_z; t.max_z = s.max_z;
    t.skip = place[s.skip];
    t.link = s.tri >= 0 ? s.tri : -1 - place[i + 1];
    tnodes[place[i]] = t;
}

// flat hierarchy: level 0 = the leaves' own boxes by slot; level l + 1 entry k = union of level l entries [16 k, 16 k + 16)
__global__ void build_flat0_kernel(const RtwNode* __restrict__ nodes, int n_nodes, float* __restrict__ flat0)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const RtwNode s = nodes[i];
    if (s.tri < 0) return;
    float* e = flat0 + (size_t)s.tri * 6;
    e[0] = s.min_x; e[1] = s.max_x; e[2] = s.min_y; e[3] = s.max_y; e[4] = s.min_z; e[5] = s.max_z;
}
__global__ void build_flat_up_kernel(const float* __restrict__ lower, int n_lower, float* __restrict__ upper, int n_upper)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_upper) return;
    float lo[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, hi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    for (int i = 16 * k; i < 16 * k + 16 && i < n_lower; i++)
        for (int c = 0; c < 3; c++) {
            const float a = lower[(size_t)i * 6 + 2 * c], b = lower[(size_t)i * 6 + 2 * c + 1];
            if (a < lo[c]) lo[c] = a;
            if (b > hi[c]) hi[c] = b;
        }
    for (int c = 0; c < 3; c++) { upper[(size_t)k * 6 + 2 * c] = lo[c]; upper[(size_t)k * 6 + 2 * c + 1] = hi[c]; }
}

// ---- screen-space bins of the reference's fixed camera, built on the device ---------------------------------------------------------
// The host routine (scene_host.cpp build_bins) operation for operation, in double precision: a leaf is listed in the bins its triangle's
// projection (grown by the jitter margin) touches when the triangle faces the camera by the reference's own float test.  Three passes:
// count per bin, scan, fill (any order), then every bin's list sorted ascending = preorder.
struct BinsGeom { int width, height, bin_w, bin_h, bx, by; double margin; };

// the bins of one leaf: calls f(bin) for every bin the leaf is listed in; returns false when the MESH gets no bins at all
template <typename F>
__device__ __forceinline__ bool bins_of_leaf(const RtwNode& nd, const RtwTri& t, const BinsGeom& g, F f)
{
    const double cx = (double)(g.width / 2), cy = (double)(g.height / 2), H = (double)g.height;
    const double zmax = (double)nd.max_z - 7.0;
    if (!(zmax < -0.01)) return false;                      // not wholly in front of the camera
    {
        const float ox = 0.0f, oy = 0.0f, oz = 7.0f;
        const float d0 = t.nx * ox + t.ny * oy + t.nz * oz;
        const float d2 = d0 - t.d1;
        if (d2 < 0) return true;                            // faces away from every camera ray: listed nowhere
    }
    const double vx[3] = { t.p0x, t.p1x, t.p2x }, vy[3] = { t.p0y, t.p1y, t.p2y }, vz[3] = { t.p0z, t.p1z, t.p2z };
    double sx[3], sy[3];
    bool finite = true;
    for (int k = 0; k < 3; k++) {
        const double qz = vz[k] - 7.0;
        sx[k] = cx + H * vx[k] / qz; sy[k] = cy + H * vy[k] / qz;
        finite = finite && sx[k] == sx[k] && sy[k] == sy[k] && fabs(sx[k]) < 1e12 && fabs(sy[k]) < 1e12;
    }
    if (!finite) return false;
    const double xa = fmin(sx[0], fmin(sx[1], sx[2])), xb = fmax(sx[0], fmax(sx[1], sx[2]));
    const double ya = fmin(sy[0], fmin(sy[1], sy[2])), yb = fmax(sy[0], fmax(sy[1], sy[2]));
    double fx0 = floor(xa - g.margin), fx1 = ceil(xb + g.margin), fy0 = floor(ya - g.margin), fy1 = ceil(yb + g.margin);
    if (fx1 < 0 || fy1 < 0 || fx0 > g.width - 1 || fy0 > g.height - 1) return true;      // off screen
    if (fx0 < 0) fx0 = 0;
    if (fy0 < 0) fy0 = 0;
    if (fx1 > g.width - 1) fx1 = g.width - 1;
    if (fy1 > g.height - 1) fy1 = g.height - 1;
    const int bx0 = (int)fx0 / g.bin_w, bx1 = (int)fx1 / g.bin_w, by0 = (int)fy0 / g.bin_h, by1 = (int)fy1 / g.bin_h;
    const double area2 = (sx[1] - sx[0]) * (sy[2] - sy[0]) - (sx[2] - sx[0]) * (sy[1] - sy[0]);
    const double scale = fabs(xb - xa) + fabs(yb - ya) + 1.0;
    const bool use_edges = fabs(area2) > 1e-9 * scale * scale;
    for (int yb_ = by0; yb_ <= by1; yb_++) {
        for (int xb_ = bx0; xb_ <= bx1; xb_++) {
            bool separated = false;
            if (use_edges) {
                const double rx0 = (double)(xb_ * g.bin_w) - g.margin, rx1 = (double)(xb_ * g.bin_w + g.bin_w - 1) + g.margin;
                const double ry0 = (double)(yb_ * g.bin_h) - g.margin, ry1 = (double)(yb_ * g.bin_h + g.bin_h - 1) + g.margin;
                for (int e = 0; e < 3 && !separated; e++) {
                    const int a = e, b2 = (e + 1) % 3, c = (e + 2) % 3;
                    double nx = sy[b2] - sy[a], ny = -(sx[b2] - sx[a]);
                    if (nx * (sx[c] - sx[a]) + ny * (sy[c] - sy[a]) > 0) { nx = -nx; ny = -ny; }
                    const double len = sqrt(nx * nx + ny * ny);
                    if (!(len > 0)) continue;
                    const double px = nx > 0 ? rx0 : rx1, py = ny > 0 ? ry0 : ry1;
                    if ((nx * (px - sx[a]) + ny * (py - sy[a])) / len > 1e-6) separated = true;
                }
            }
            if (!separated) f(yb_ * g.bx + xb_);
        }
    }
    return true;
}

// pass 0: counts[bin] += 1 per listed leaf; pass 1: entries at off[bin] + (a ticket from fill[bin])
template <int PASS>
__global__ void bins_pass_kernel(const RtwNode* __restrict__ nodes, const RtwTri* __restrict__ tris, int n_nodes, BinsGeom g,
                                 uint32_t* __restrict__ counts, const uint32_t* __restrict__ off, uint32_t* __restrict__ ent, uint32_t* __restrict__ no_bins)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const RtwNode nd = nodes[i];
    if (nd.tri < 0) return;
    const RtwTri t = tris[nd.tri];
    const bool ok = bins_of_leaf(nd, t, g, [&](int bin) {
        if (PASS == 0) atomicAdd(&counts[bin], 1u);
        else ent[off[bin] + atomicAdd(&counts[bin], 1u)] = (uint32_t)i;
    });
    if (!ok) *no_bins = 1u;
}

// exclusive scan of counts[0..n) into off[0..n], off[n] = total; counts zeroed for the fill pass (one block)
__global__ __launch_bounds__(1024) void bins_scan_kernel(uint32_t* __restrict__ counts, uint32_t* __restrict__ off, int n)
{
    __shared__ uint32_t part[1024];
    const int per = (n + 1023) / 1024;
    const int lo = threadIdx.x * per, hi = lo + per < n ? lo + per : n;
    uint32_t s = 0;
    for (int i = lo; i < hi; i++) s += counts[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t t = 0; for (int k = 0; k < 1024; k++) { const uint32_t v = part[k]; part[k] = t; t += v; } off[n] = t; }
    __syncthreads();
    uint32_t t = part[threadIdx.x];
    for (int i = lo; i < hi; i++) { off[i] = t; t += counts[i]; counts[i] = 0u; }
}

// every bin's entries ascending (node index = preorder).  A wave per bin: the list goes to LDS and every entry's place is its rank -- the count
// of smaller entries (a bin holds a node once, so ranks are distinct).  (A lane per bin with an insertion sort in global memory took 2.2 ms on
// unitychan's bins, whose longest list has 197 entries: the whole launch waited for that lane.)  Lists longer than the LDS row keep the slow way.
#define RTW_BINS_SORT_ROW 1024
__global__ __launch_bounds__(256) void bins_sort_kernel(const uint32_t* __restrict__ off, uint32_t* __restrict__ ent, int n_bins)
{
    __shared__ uint32_t rows[4][RTW_BINS_SORT_ROW];
    const int b = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = lane_id();
    if (b >= n_bins) return;
    const uint32_t lo = off[b], hi = off[b + 1], n = hi - lo;
    if (n < 2u) return;
    if (n <= (uint32_t)RTW_BINS_SORT_ROW) {
        uint32_t* row = rows[threadIdx.x >> 6];
        for (uint32_t i = (uint32_t)lane; i < n; i += 64u) row[i] = ent[lo + i];
        wave_lds_sync();
        for (uint32_t i = (uint32_t)lane; i < n; i += 64u) {
            const uint32_t v = row[i];
            uint32_t rank = 0u;
            for (uint32_t j = 0; j < n; j++) rank += row[j] < v ? 1u : 0u;
            ent[lo + rank] = v;
        }
        return;
    }
    if (lane != 0) return;
    for (uint32_t i = lo + 1; i < hi; i++) {
        const uint32_t v = ent[i];
        uint32_t j = i;
        while (j > lo && ent[j - 1] > v) { ent[j] = ent[j - 1]; j--; }
        ent[j] = v;
    }
}
