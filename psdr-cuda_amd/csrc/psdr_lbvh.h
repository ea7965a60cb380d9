// psdr_lbvh.h -- BVH construction ON THE DEVICE (included by psdr_hip.hip).
//
// What it replaces: the OptiX GAS build the reference runs on the GPU at every Scene::configure()
// (include/psdr/scene/optix.h:277-340, src/scene/scene.cpp:247-248).  The host binned-SAH builder
// (psdr_bvh_build.h) needs the triangle table on the host -- a D2H copy of 96 bytes per triangle, a stream
// sync, a single-threaded build and an H2D copy: ~10 ms at 50 k triangles, ~0.3 s at 1 M.  This builder never
// leaves the device: Morton codes of the triangle centroids, rocPRIM radix sort, the radix tree over the sorted
// triangles built node-parallel (Karras 2012) with every subtree of <= 4 triangles collapsed into a leaf, boxes
// fitted bottom-up with one arrival counter per node.  One 16-byte read-back (depth, area, validity) per build.
// The same bottom-up pass, rerun, is the device REFIT of such a tree.
// Measured (tools/bvh_build_probe.py): 263 k triangles 2.2 ms against 127 ms on the host, 50 k 0.9 against 25 ms.  The tree
// is a Morton tree, not an SAH tree -- PathTracer(3) renderC 1.35x (two bunnies) to 1.5x (rooms with wall-sized triangles)
// slower than on the SAH tree: psdr_bvh_build picks it for large tables (>= kLbvhAutoTris triangles), where the host
// build would dominate a configure(), or when PSDR_BVH_BUILD=device asks for it (PSDR_BVH_BUILD=host: never).
// Node format, leaf encoding and padding are the host builder's (BvhNode, ~((first << 3) | (count - 1))).
#pragma once

constexpr int kLbvhLeaf = 4;                       // triangles per leaf (the host builder's kMaxLeaf)
constexpr int kLbvhAutoTris = 1 << 18;

struct LbvhInfo { int32_t depth; float area; int32_t bad; float pad; };

__device__ __forceinline__ uint32_t lbvh_expand10(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
// order-preserving float <-> uint map for atomicMin / atomicMax
__device__ __forceinline__ uint32_t lbvh_f2o(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float lbvh_o2f(uint32_t o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o); }

__device__ __forceinline__ void lbvh_tri_box(const float *__restrict__ r, float *lo, float *hi) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float p = r[k], q = r[k] + r[3 + k], w = r[k] + r[6 + k];
        lo[k] = fminf(p, fminf(q, w)); hi[k] = fmaxf(p, fmaxf(q, w));
    }
}

// scene bounds (ordered uints: lo[3], hi[3]) + validity of the vertices
__global__ __launch_bounds__(kBlock) void k_lbvh_bounds(const float *__restrict__ tri_info, int T, uint32_t *__restrict__ bounds, LbvhInfo *__restrict__ info) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    bool bad = false;
    if (i < T) {
        const float *r = tri_info + (size_t) i * PSDR_TRI_STRIDE;
        lbvh_tri_box(r, lo, hi);
#pragma unroll
        for (int k = 0; k < 9; ++k) bad = bad || !isfinite(r[k]);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { lo[k] = fminf(lo[k], __shfl_down(lo[k], off, 64)); hi[k] = fmaxf(hi[k], __shfl_down(hi[k], off, 64)); }
    }
    if ((threadIdx.x & 63) == 0 && lo[0] <= hi[0]) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { atomicMin(bounds + k, lbvh_f2o(lo[k])); atomicMax(bounds + 3 + k, lbvh_f2o(hi[k])); }
    }
    if (bad) info->bad = 1;
}

__global__ __launch_bounds__(kBlock) void k_lbvh_codes(const float *__restrict__ tri_info, int T, const uint32_t *__restrict__ bounds,
                                                       uint32_t *__restrict__ keys, int32_t *__restrict__ vals, LbvhInfo *__restrict__ info) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    float slo[3], ext[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { slo[k] = lbvh_o2f(bounds[k]); ext[k] = lbvh_o2f(bounds[3 + k]) - slo[k]; }
    if (i == 0) info->pad = fmaxf(1e-6f, 1e-5f * fmaxf(ext[0], fmaxf(ext[1], ext[2])));     // the host builder's padding of leaf boxes
    if (i >= T) return;
    float lo[3], hi[3];
    lbvh_tri_box(tri_info + (size_t) i * PSDR_TRI_STRIDE, lo, hi);
    uint32_t q[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float c = 0.5f * (lo[k] + hi[k]);
        const float t = ext[k] > 0.f ? (c - slo[k]) / ext[k] : 0.f;
        q[k] = (uint32_t) fminf(fmaxf(t * 1024.f, 0.f), 1023.f);
    }
    keys[i] = (lbvh_expand10(q[0]) << 2) | (lbvh_expand10(q[1]) << 1) | lbvh_expand10(q[2]);
    vals[i] = i;
}

// leaf triangle records in Morton order
__global__ __launch_bounds__(kBlock) void k_lbvh_tris(const float *__restrict__ tri_info, int T, const int32_t *__restrict__ ids_sorted, float4 *__restrict__ btris) {
    const int t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= T) return;
    const int id = ids_sorted[t];
    const float *r = tri_info + (size_t) id * PSDR_TRI_STRIDE;
    float4 a{r[0], r[1], r[2], 0.f};
    a.w = __int_as_float(id);
    btris[(size_t) t * 3] = a;
    btris[(size_t) t * 3 + 1] = float4{r[3], r[4], r[5], 0.f};
    btris[(size_t) t * 3 + 2] = float4{r[6], r[7], r[8], 0.f};
}

// common-prefix length of the keys of sorted triangles i and j (ties broken by the position); -1 outside the array
__device__ __forceinline__ int lbvh_delta(const uint32_t *__restrict__ key, int T, int i, int j) {
    if (j < 0 || j >= T) return -1;
    const uint32_t a = key[i], b = key[j];
    return a == b ? 32 + __clz((uint32_t) (i ^ j)) : __clz(a ^ b);
}

// One thread per inner node of the radix tree over the T sorted triangles (Karras, "Maximizing parallelism in the
// construction of BVHs, octrees and k-d trees", HPG 2012): its range, the split, the two children.  A child whose range
// holds <= kLbvhLeaf triangles IS a leaf (the range is contiguous in btris: ~((first << 3) | (count - 1))) -- leaves follow
// the cells of the radix tree instead of cutting the Morton curve every four triangles -- and the nodes below it stay
// unused.  Children learn their parent and side (parent * 2 + side); leaf_parent is indexed by the leaf's first triangle.
__global__ __launch_bounds__(kBlock) void k_lbvh_hierarchy(const uint32_t *__restrict__ key, int T, BvhNode *__restrict__ nodes,
                                                           int32_t *__restrict__ node_parent, int32_t *__restrict__ leaf_parent) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= T - 1) return;
    const int d = lbvh_delta(key, T, i, i + 1) - lbvh_delta(key, T, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = lbvh_delta(key, T, i, i - d);
    int lmax = 2;
    while (lbvh_delta(key, T, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (lbvh_delta(key, T, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int lo = min(i, j), hi = max(i, j);
    if (i == 0) node_parent[0] = -1;
    if (hi - lo + 1 <= kLbvhLeaf && i != 0) return;                    // inside a leaf
    const int dnode = lbvh_delta(key, T, i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) / 2;
        if (lbvh_delta(key, T, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + min(d, 0);
    const int n0 = gamma - lo + 1, n1 = hi - gamma;
    int32_t c0, c1;
    if (n0 <= kLbvhLeaf) { c0 = ~((lo << 3) | (n0 - 1)); leaf_parent[lo] = i * 2; } else { c0 = gamma; node_parent[gamma] = i * 2; }
    if (n1 <= kLbvhLeaf) { c1 = ~(((gamma + 1) << 3) | (n1 - 1)); leaf_parent[gamma + 1] = i * 2 + 1; } else { c1 = gamma + 1; node_parent[gamma + 1] = i * 2 + 1; }
    nodes[i].c0 = c0; nodes[i].c1 = c1;
}

// bottom-up fit: the thread of a leaf's first triangle boxes the leaf, writes the box into its slot of the parent and
// climbs; the SECOND arrival at a node (arrival counter) finds both child boxes, forms their union and carries it on.
// Also the inner-node area sum (the refit's degradation measure).  Rerun after k_refit_leaves it is the refit.
__global__ __launch_bounds__(kBlock) void k_lbvh_fit(int T, BvhNode *nodes, const float4 *__restrict__ btris, const int32_t *__restrict__ node_parent,
                                                     const int32_t *__restrict__ leaf_parent, uint32_t *arrivals, float *area_sum, float pad) {
    const int t0 = blockIdx.x * kBlock + threadIdx.x;
    float area = 0.f;
    int link = t0 < T ? leaf_parent[t0] : -1;
    if (link >= 0) {
        const int32_t code = (link & 1) ? nodes[link >> 1].c1 : nodes[link >> 1].c0;
        const int cnt = ((~code) & 7) + 1;
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int j = 0; j < cnt; ++j) {
            const float4 a = btris[(size_t) (t0 + j) * 3], b = btris[(size_t) (t0 + j) * 3 + 1], e = btris[(size_t) (t0 + j) * 3 + 2];
            const float p[3] = {a.x, a.y, a.z}, q[3] = {a.x + b.x, a.y + b.y, a.z + b.z}, w[3] = {a.x + e.x, a.y + e.y, a.z + e.z};
#pragma unroll
            for (int k = 0; k < 3; ++k) { lo[k] = fminf(lo[k], fminf(p[k], fminf(q[k], w[k]))); hi[k] = fmaxf(hi[k], fmaxf(p[k], fmaxf(q[k], w[k]))); }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) { lo[k] -= pad; hi[k] += pad; }
        while (link >= 0) {
            const int p = link >> 1, side = link & 1;
            float *mlo = side ? nodes[p].lo1 : nodes[p].lo0, *mhi = side ? nodes[p].hi1 : nodes[p].hi0;
#pragma unroll
            for (int k = 0; k < 3; ++k) { __hip_atomic_store(mlo + k, lo[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(mhi + k, hi[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            __threadfence();
            if (atomicAdd(arrivals + p, 1u) == 0u) break;              // first at this node: the sibling finishes it
            __threadfence();
            const float *olo = side ? nodes[p].lo0 : nodes[p].lo1, *ohi = side ? nodes[p].hi0 : nodes[p].hi1;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                lo[k] = fminf(lo[k], __hip_atomic_load(olo + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                hi[k] = fmaxf(hi[k], __hip_atomic_load(ohi + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            }
            const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
            area += dx * dy + dy * dz + dz * dx;
            link = node_parent[p];
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) area += __shfl_down(area, off, 64);
    if ((threadIdx.x & 63) == 0 && area != 0.f) atomicAdd(area_sum, area);
}

__global__ __launch_bounds__(kBlock) void k_lbvh_depth(int T, const int32_t *__restrict__ node_parent, const int32_t *__restrict__ leaf_parent, LbvhInfo *info) {
    const int t = blockIdx.x * kBlock + threadIdx.x;
    int depth = 0;
    int link = t < T ? leaf_parent[t] : -1;
    while (link >= 0) { ++depth; link = node_parent[link >> 1]; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) depth = max(depth, __shfl_down(depth, off, 64));
    if ((threadIdx.x & 63) == 0 && depth > 0) atomicMax(&info->depth, depth);
}

// hot rows of the reverse-mode gradient cache without the table on the host: keys = triangle areas (positive floats order
// like their bit patterns), sorted descending; the host reads the first kMaxHotRows ids
__global__ __launch_bounds__(kBlock) void k_lbvh_area_keys(const float *__restrict__ tri_info, int T, uint32_t *__restrict__ keys, int32_t *__restrict__ vals) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= T) return;
    const float a = tri_info[(size_t) i * PSDR_TRI_STRIDE + 21];
    keys[i] = a > 0.f && isfinite(a) ? __float_as_uint(a) : 0u;
    vals[i] = i;
}
