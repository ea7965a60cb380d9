// psdr_hip.hip -- host side and C ABI of include/psdr_hip.h (+ k_trace).  The render kernels live in
// psdr_kernels.h and are compiled per scene flag set in psdr_variant.hip; this file owns the scene
// handle, the BVH build, the LDS plan and picks the kernel variant of the scene.
#define PSDR_WIDE_TREE 2      // k_trace serves every scene: both walks, chosen by SceneView::nodes4
#include "psdr_host.h"
#include "psdr_bvh_build.h"

#include <cstdlib>
#include <cstring>
#include <rocprim/rocprim.hpp>

namespace {
// ------------------------------------------------------------------------------- k_trace
__global__ __launch_bounds__(kBlock) void k_trace(LaunchCtx cx, int m, const float *__restrict__ ox, const float *__restrict__ oy,
                                                  const float *__restrict__ oz, const float *__restrict__ dx,
                                                  const float *__restrict__ dy, const float *__restrict__ dz,
                                                  const float *__restrict__ tmax, int32_t *__restrict__ out_shape,
                                                  int32_t *__restrict__ out_tri, float *__restrict__ out_u, float *__restrict__ out_v) {
    TraversalStack st; setup_lds(cx, st);
    const SceneView &sc = cx.sc;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < m; i += gridDim.x * kBlock) {
        const Hit h = closest_hit<false, -1>(sc, st, Vec3f{ox[i], oy[i], oz[i]}, Vec3f{dx[i], dy[i], dz[i]}, tmax[i]);
        out_tri[i] = h.tri;
        out_shape[i] = h.tri >= 0 ? (sc.d.tri_mesh[h.tri] & ~PSDR_TRI_FACE_NORMALS) : -1;
        out_u[i] = h.u; out_v[i] = h.v;
    }
}


// ------------------------------------------------------------------------------- k_wf_trace
// The DENSE trace kernel of the traced wavefront (psdr_kernels.h run_camera_wavefront; replaces optixLaunch + the closest-hit programs,
// src/scene/scene_optix.cpp:81-126, cuda/psdr_cuda.cu:9-45, for the rays that enter a tree box).  In a room only a fifth of the rays meet an
// object, and inside the render kernels a wave pays for its slowest lane at every closest_hit: on the 50 k-triangle interior the fused
// PathTracer spent two thirds of its instructions walking trees with one or two lanes busy (tools/walk_stats: 18 % of the bounce rays enter a
// tree and visit 8.6 nodes, a wave of 64 almost always holds one).  Here the walk is a kernel of its own:
//   * input: a queue of ray requests (o | dest), (d | -) -- only rays that enter a box, written by the stage that sampled them;
//   * one persistent workgroup of 1024 threads per CU with the 4-wide tree in its 160 KB of LDS (the whole tree of a 5 k-triangle mesh, the
//     top levels of a larger forest; 80-byte rows: the 16-byte words of 16 random rows then spread over all bank groups) and the
//     traversal stacks (one column per lane) behind it;
//   * LANE REFILL: a lane whose ray is done writes its hit row and takes the next request at the following leaf boundary -- the wave
//     keeps a block of 64 requests in registers and hands them out by rank (ds_bpermute), so every node step runs with (nearly) all lanes busy;
//   * output: hit[dest] = (tri, u, v, t) of the closest TREE hit (tri < 0: none), the same leaf test as every other walk of the library.
constexpr int kTraceBlock = 1024, kTraceNodeStride = 80, kTraceStackMax = 16;
#ifndef PSDR_TRACE_SERVICE_MIN
#define PSDR_TRACE_SERVICE_MIN 16
#endif
#ifndef PSDR_TRACE_VOTE_NODE
#define PSDR_TRACE_VOTE_NODE 1
#endif
#ifndef PSDR_TRACE_VOTE_LEAF
#define PSDR_TRACE_VOTE_LEAF 1
#endif
constexpr int kTraceServiceMin = PSDR_TRACE_SERVICE_MIN, kTraceVoteNode = PSDR_TRACE_VOTE_NODE, kTraceVoteLeaf = PSDR_TRACE_VOTE_LEAF;
struct TraceArgs {
    // boxes first: read through the kernel-argument segment pointer with a run-time index (an index into the by-value struct sends it to scratch)
    float4 blas_lo[kMaxBlas], blas_hi[kMaxBlas];      // hi.w = root of the tree in the 4-wide node array
    const Bvh4Node *nodes4; const float4 *btris;
    const float4 *req; const int32_t *count; float4 *hit; int32_t *ovf;
    const int32_t *sec_edge_faces;     // IGN instances: a request whose second row carries a secondary-edge index skips that edge's two faces
    long long sub_cap;
    int32_t n_blas, n_lnodes, off_stack, stack_entries, ovf_stride;
};
static_assert(offsetof(TraceArgs, blas_lo) == 0, "k_wf_trace reads the boxes through the kernel-argument segment pointer");

// MULTI: more than one tree (nearest box first, the remaining boxes re-tested against the current hit after every walk, as closest_hit does);
// OVF: the worst-case stack of the forest is deeper than the LDS columns -- entries beyond them live in a per-lane global column (rare: the deepest
// stack of a ray on the bunny / the interior is 11 of 24 / 22 possible entries).
// IGN: rays that start ON a secondary edge (the probe pass of a traced secondary-edge launch): rb.w = the edge, its adjacent faces are not hit.
// WPE: resident waves per SIMD the instance is compiled for -- 4: one workgroup per CU with all of its LDS; 8: two workgroups per CU (<= 64
// VGPRs), each with half of the LDS (shorter stack columns, the rest of the stacks in the overflow columns; fewer staged nodes)
template <bool MULTI, bool OVF, bool IGN = false, int WPE = 4>
__global__ __launch_bounds__(kTraceBlock, WPE) void k_wf_trace(TraceArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int32_t kDone = 0x7fffffff;
    {
        float4 *dst = reinterpret_cast<float4 *>(psdr_dyn_lds);
        const float4 *src = reinterpret_cast<const float4 *>(a.nodes4);
        for (int i = threadIdx.x; i < a.n_lnodes * 4; i += kTraceBlock) dst[(i >> 2) * (kTraceNodeStride / 16) + (i & 3)] = src[i];
    }
    // the 64 request sub-queues as one list of 64-request blocks; wave w of the grid takes blocks w, w + W, ...
    __shared__ int s_pref[kWfSub + 1];
    if (threadIdx.x < kWfSub) {
        int c = (a.count[threadIdx.x * kWfCountStride] + 63) / 64;
#pragma unroll
        for (int off = 1; off < kWfSub; off <<= 1) { const int o = __shfl_up(c, off, 64); if ((int) threadIdx.x >= off) c += o; }
        s_pref[threadIdx.x + 1] = c;
        if (threadIdx.x == 0) s_pref[0] = 0;
    }
    __syncthreads();
    const int total = s_pref[kWfSub];
    const int lane = threadIdx.x & 63;
    const int W = (int) gridDim.x * (kTraceBlock / 64);
    int next_b = __builtin_amdgcn_readfirstlane((int) blockIdx.x * (kTraceBlock / 64) + (int) (threadIdx.x >> 6));
    // LDS-address-space pointer: with a generic one the compiler merges the two arms of the pop below ("LDS column or overflow column") into ONE
    // flat_load_dword on a selected address -- every pop of the walk then takes the texture-address path and waits for vmcnt AND lgkmcnt
    typedef __attribute__((address_space(3))) int32_t lds_int;
    lds_int *stack = (lds_int *) reinterpret_cast<int32_t *>(psdr_dyn_lds + a.off_stack) + threadIdx.x;
    const int S = a.stack_entries;
    int32_t *ovf = OVF ? a.ovf + ((size_t) blockIdx.x * kTraceBlock + threadIdx.x) : nullptr;
    typedef __attribute__((address_space(4))) const float4 kernarg_float4;
    const kernarg_float4 *boxes = (kernarg_float4 *) __builtin_amdgcn_kernarg_segment_ptr();

    bool active = false;
    Vec3f o(0.f), d(0.f), inv(0.f);
    Hit best; best.tri = -1; best.u = best.v = -1.f; best.t = INFINITY;
    uint32_t dest = 0, cand = 0;
    int ig0 = -1, ig1 = -1;
    int32_t cur = kDone;                                           // >= 0: inner node, < 0: leaf, kDone: between trees / finished / idle
    int sp = 0, li = 0;                                            // li: next triangle of the leaf in `cur`
    float4 ra{0.f, 0.f, 0.f, 0.f}, rb{0.f, 0.f, 0.f, 0.f};        // this lane's request of the wave's current block
    int blk_n = 0, consumed = 0;
    bool exhausted = false;                                        // wave-uniform: no request left for this wave
    auto pop = [&]() {
        if (sp > 0) { --sp; if (!OVF || sp < S) cur = stack[sp * kTraceBlock]; else cur = ovf[(size_t) (sp - S) * a.ovf_stride]; }
        else cur = kDone;
    };
    // VOTE scheduling: every iteration the wave runs ONE step of the phase more of its lanes wait for -- a node step (lanes at an inner node)
    // or a leaf step (two triangles, lanes at a leaf) -- instead of walking inner nodes until the last lane has reached a leaf
    // (tools/simd_sim: 1.25x fewer wave steps than while-while on the rays that enter the bunny's box, both with refill); finished rays are
    // written out and idle lanes refilled when kTraceServiceMin lanes wait for it (or nothing else is left to do).
    for (;;) {
        const bool inner = cur >= 0 && cur != kDone, leaf = cur < 0;
        const bool ended = active && cur == kDone;                  // finished, or between two trees
        const unsigned long long m_inner = __ballot(inner), m_leaf = __ballot(leaf);
        const int n_inner = (int) __popcll(m_inner), n_leaf = (int) __popcll(m_leaf);
        const int n_service = (int) __popcll(__ballot(ended || (!active && !exhausted)));
        if (n_service >= kTraceServiceMin || (n_inner + n_leaf == 0)) {
            // ---- a ray whose walk is over: its hit row
            if (ended && (!MULTI || cand == 0u)) {
                a.hit[dest] = float4{__int_as_float(best.tri), best.u, best.v, best.t};
                active = false;
            }
            // ---- refill the idle lanes from the wave's block of requests
            unsigned long long idle = __ballot(!active);
            while (idle != 0ull) {
                if (consumed >= blk_n) {
                    if (next_b >= total) { exhausted = true; break; }
                    int lo = 0, hi = kWfSub - 1;
                    while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_pref[mid + 1] <= next_b) lo = mid + 1; else hi = mid; }
                    const int off = (next_b - s_pref[lo]) * 64;
                    blk_n = min(64, a.count[lo * kWfCountStride] - off);
                    const float4 *r = a.req + 2 * ((size_t) lo * a.sub_cap + off + lane);
                    if (lane < blk_n) { ra = r[0]; rb = r[1]; }
                    consumed = 0; next_b += W;
                }
                const int take = min((int) __popcll(idle), blk_n - consumed);
                const int rank = (int) __popcll(idle & ((1ull << lane) - 1ull));
                const int from = (consumed + rank) & 63;
                const float ox = __shfl(ra.x, from, 64), oy = __shfl(ra.y, from, 64), oz = __shfl(ra.z, from, 64), dw = __shfl(ra.w, from, 64);
                const float dx = __shfl(rb.x, from, 64), dy = __shfl(rb.y, from, 64), dz = __shfl(rb.z, from, 64);
                int edge = -1;
                if (IGN) edge = __float_as_int(__shfl(rb.w, from, 64));
                if (!active && rank < take) {
                    if (IGN) { ig0 = ig1 = -1; if (edge >= 0) { ig0 = a.sec_edge_faces[2 * edge]; ig1 = a.sec_edge_faces[2 * edge + 1]; } }
                    o = Vec3f{ox, oy, oz}; d = Vec3f{dx, dy, dz}; inv = Vec3f{1.f / dx, 1.f / dy, 1.f / dz};
                    dest = (uint32_t) __float_as_int(dw);
                    best.tri = -1; best.u = best.v = -1.f; best.t = INFINITY;
                    sp = 0; li = 0; active = true;
                    if (MULTI) { cur = kDone; cand = (1u << a.n_blas) - 1u; }
                    else { cur = __float_as_int(boxes[kMaxBlas].w); cand = 0u; }
                }
                consumed += take;
                idle = __ballot(!active);
            }
            if (__ballot(active) == 0ull) break;
            // ---- the next tree of a ray between two walks: the nearest box its segment [0, t_best] still enters (wave-uniform loop, boxes in SGPRs)
            if (MULTI) {
                const bool need = active && cur == kDone && cand != 0u;
                if (__ballot(need) != 0ull) {
                    float near_t = INFINITY; int32_t root = kDone; uint32_t pick = 0u;
                    for (int k = 0; k < a.n_blas; ++k) {
                        const float4 blo = boxes[k], bhi = boxes[kMaxBlas + k];
                        const float lo3[3] = {blo.x, blo.y, blo.z}, hi3[3] = {bhi.x, bhi.y, bhi.z};
                        float te;
                        const bool h = slab(lo3, hi3, o, inv, best.t, te);
                        const bool c = need && ((cand >> k) & 1u) != 0u;
                        if (c && !h) cand &= ~(1u << k);
                        else if (c && te < near_t) { near_t = te; root = __float_as_int(bhi.w); pick = 1u << k; }
                    }
                    if (need) { cand &= ~pick; cur = pick ? root : kDone; li = 0; if (!pick) cand = 0u; }
                }
            }
            continue;
        }
        if (n_inner * kTraceVoteNode >= n_leaf * kTraceVoteLeaf) {
            // ---- one node step of the lanes at an inner node
            if (inner) {
                Bvh4Node n;
                if (cur < a.n_lnodes) n = *reinterpret_cast<const Bvh4Node *>(psdr_dyn_lds + __umul24((uint32_t) cur, (uint32_t) kTraceNodeStride));
                else n = a.nodes4[cur];
                const float ax = __int_as_float((int) ((n.exps & 0xffu) << 23)) * inv.x, ay = __int_as_float((int) (((n.exps >> 8) & 0xffu) << 23)) * inv.y,
                            az = __int_as_float((int) (((n.exps >> 16) & 0xffu) << 23)) * inv.z;
                const float bx = (n.org[0] - o.x) * inv.x, by = (n.org[1] - o.y) * inv.y, bz = (n.org[2] - o.z) * inv.z;
                const bool px = inv.x >= 0.f, py = inv.y >= 0.f, pz = inv.z >= 0.f;
                const uint32_t nx = px ? n.qlo[0] : n.qhi[0], fx = px ? n.qhi[0] : n.qlo[0];
                const uint32_t ny = py ? n.qlo[1] : n.qhi[1], fy = py ? n.qhi[1] : n.qlo[1];
                const uint32_t nz = pz ? n.qlo[2] : n.qhi[2], fz = pz ? n.qhi[2] : n.qlo[2];
                uint32_t key[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float tnx = (float) ((nx >> (8 * c)) & 0xffu) * ax + bx, tfx = (float) ((fx >> (8 * c)) & 0xffu) * ax + bx;
                    const float tny = (float) ((ny >> (8 * c)) & 0xffu) * ay + by, tfy = (float) ((fy >> (8 * c)) & 0xffu) * ay + by;
                    const float tnz = (float) ((nz >> (8 * c)) & 0xffu) * az + bz, tfz = (float) ((fz >> (8 * c)) & 0xffu) * az + bz;
                    const float tn = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, 0.f));
                    const float tf = fminf(fminf(tfx, tfy), fminf(tfz, best.t));
                    // tn >= 0: its bit pattern orders like the value; the two low mantissa bits carry the slot (the nearest child is visited first,
                    // the others are pushed in slot order: a full sort saves 1.5 % of the node visits, tools/walk_stats, and costs 25 instructions)
                    const uint32_t k = ((uint32_t) __float_as_int(tn) & ~3u) | (uint32_t) c;
                    const bool hit = (tn <= tf) & (n.child[c] != kNoChild);
                    key[c] = hit ? k : 0xffffffffu;
                }
                const uint32_t kmin = min(min(key[0], key[1]), min(key[2], key[3]));
                const int slot = (int) (kmin & 3u);
                if (OVF && __ballot(sp + 3 > S) != 0ull) {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (key[c] != 0xffffffffu && c != slot) {
                            if (sp < S) stack[sp * kTraceBlock] = n.child[c]; else ovf[(size_t) (sp - S) * a.ovf_stride] = n.child[c];
                            ++sp;
                        }
                } else {
                    // room for three more everywhere in the wave: unconditional stores, the stack pointer moves where the child counts
#pragma unroll
                    for (int c = 0; c < 4; ++c) { stack[sp * kTraceBlock] = n.child[c]; sp += (key[c] != 0xffffffffu && c != slot) ? 1 : 0; }
                }
                const int32_t c01 = (slot & 1) ? n.child[1] : n.child[0], c23 = (slot & 1) ? n.child[3] : n.child[2];
                if (kmin != 0xffffffffu) cur = (slot & 2) ? c23 : c01;
                else pop();
                li = 0;
            }
        } else {
            // ---- one leaf step of the lanes at a leaf: two triangles (six loads in flight; an odd end tests its last triangle twice -- the second test
            // fails t < t_best)
            if (leaf) {
                const int enc = ~cur, first = enc >> 3, cnt = (enc & 7) + 1;
                const int j = li + 1 < cnt ? li + 1 : li;
                const float4 *bt = a.btris + (size_t) (first + li) * 3, *bu = a.btris + (size_t) (first + j) * 3;
                const float4 a0 = bt[0], b0 = bt[1], c0 = bt[2];
                const float4 a1 = bu[0], b1 = bu[1], c1 = bu[2];
                leaf_triangle_test<IGN>(a0, b0, c0, o, d, best, ig0, ig1);
                leaf_triangle_test<IGN>(a1, b1, c1, o, d, best, ig0, ig1);
                li += 2;
                if (li >= cnt) { pop(); li = 0; }
            }
        }
    }
#else
    (void) a;
#endif
}

// ------------------------------------------------------------------------------- BVH refit
// Between two Scene::configure() calls of an optimisation loop the topology stays and the vertices move a
// little: instead of the host rebuild (D2H of the triangle table, SAH build, H2D: ~2.5 ms for 5 k
// triangles, ~10 ms for 50 k) the tree is REFITTED on the device -- leaf triangles re-read from the new
// table, boxes recomputed bottom-up level by level (breadth-first node order = levels are contiguous).
// Always correct (boxes are recomputed from the actual triangles); the summed box area of the inner nodes
// tells when the tree has degraded enough to be rebuilt.
__global__ __launch_bounds__(kBlock) void k_refit_leaves(float4 *__restrict__ btris, const float *__restrict__ tri_info, int n) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int id = __float_as_int(btris[(size_t) i * 3].w);
    const float *r = tri_info + (size_t) id * PSDR_TRI_STRIDE;
    float4 a{r[0], r[1], r[2], 0.f};
    a.w = __int_as_float(id);
    btris[(size_t) i * 3] = a;
    btris[(size_t) i * 3 + 1] = float4{r[3], r[4], r[5], 0.f};
    btris[(size_t) i * 3 + 2] = float4{r[6], r[7], r[8], 0.f};
}
__device__ __forceinline__ void child_box(const BvhNode *nodes, const float4 *btris, int32_t c, float pad, float *lo, float *hi) {
    for (int k = 0; k < 3; ++k) { lo[k] = INFINITY; hi[k] = -INFINITY; }
    if (c < 0) {
        const int enc = ~c, first = enc >> 3, cnt = (enc & 7) + 1;
        for (int i = 0; i < cnt; ++i) {
            const float4 a = btris[(size_t) (first + i) * 3], b = btris[(size_t) (first + i) * 3 + 1], e = btris[(size_t) (first + i) * 3 + 2];
            const float p[3] = {a.x, a.y, a.z}, q[3] = {a.x + b.x, a.y + b.y, a.z + b.z}, w[3] = {a.x + e.x, a.y + e.y, a.z + e.z};
            for (int k = 0; k < 3; ++k) { lo[k] = fminf(lo[k], fminf(p[k], fminf(q[k], w[k]))); hi[k] = fmaxf(hi[k], fmaxf(p[k], fmaxf(q[k], w[k]))); }
        }
        for (int k = 0; k < 3; ++k) { lo[k] -= pad; hi[k] += pad; }
    } else {
        const BvhNode &n = nodes[c];
        for (int k = 0; k < 3; ++k) { lo[k] = fminf(n.lo0[k], n.lo1[k]); hi[k] = fmaxf(n.hi0[k], n.hi1[k]); }
    }
}
__global__ __launch_bounds__(kBlock) void k_refit_level(BvhNode *__restrict__ nodes, const float4 *__restrict__ btris, int first, int count, float pad,
                                                        float *__restrict__ area_sum) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    float area = 0.f;
    if (i < count) {
        BvhNode n = nodes[first + i];
        child_box(nodes, btris, n.c0, pad, n.lo0, n.hi0);
        child_box(nodes, btris, n.c1, pad, n.lo1, n.hi1);
        nodes[first + i] = n;
        const float dx = fmaxf(n.hi0[0], n.hi1[0]) - fminf(n.lo0[0], n.lo1[0]), dy = fmaxf(n.hi0[1], n.hi1[1]) - fminf(n.lo0[1], n.lo1[1]),
                    dz = fmaxf(n.hi0[2], n.hi1[2]) - fminf(n.lo0[2], n.lo1[2]);
        area = dx * dy + dy * dz + dz * dx;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) area += __shfl_down(area, off, 64);
    if ((threadIdx.x & 63) == 0 && area != 0.f) atomicAdd(area_sum, area);
}

// Two-level trees: what the kernel arguments carry (inline triangles, tree boxes) after a device refit --
// gathered into one small buffer, read back with one copy.
__global__ void k_gather_top(float4 *__restrict__ top, const BvhNode *__restrict__ nodes, const float *__restrict__ tri_info,
                             const int32_t *__restrict__ inline_ids, int n_inline, int n_blas) {
    const int i = threadIdx.x;
    if (i < n_inline) {
        const int id = inline_ids[i];
        const float *r = tri_info + (size_t) id * PSDR_TRI_STRIDE;
        float4 a{r[0], r[1], r[2], 0.f};
        a.w = __int_as_float(id);
        top[i * 3] = a; top[i * 3 + 1] = float4{r[3], r[4], r[5], 0.f}; top[i * 3 + 2] = float4{r[6], r[7], r[8], 0.f};
    }
    if (i < n_blas) {            // roots are the first n_blas nodes (level 0 of the forest)
        const BvhNode &n = nodes[i];
        top[psdr_host::kMaxInlineTris * 3 + i] = float4{fminf(n.lo0[0], n.lo1[0]), fminf(n.lo0[1], n.lo1[1]), fminf(n.lo0[2], n.lo1[2]), 0.f};
        top[psdr_host::kMaxInlineTris * 3 + kMaxBlas + i] = float4{fmaxf(n.hi0[0], n.hi1[0]), fmaxf(n.hi0[1], n.hi1[1]), fmaxf(n.hi0[2], n.hi1[2]), 0.f};
    }
}

// ------------------------------------------------------------------------ 4-wide tree (boxes)
// One thread per 4-wide node: the boxes of its (up to) four children -- each stored in a BVH2 node (src = 2 * node + side) -- are
// quantised to 8 bits per plane relative to their union: origin = lower corner, per-axis scale 2^e with 255 * 2^e >= extent, lower planes
// rounded down, upper planes rounded up.  Runs after every build and refit of the BVH2 (a pure function of its boxes).
__global__ __launch_bounds__(kBlock) void k_bvh4_fill(Bvh4Node *__restrict__ out, const BvhNode *__restrict__ nodes, const int32_t *__restrict__ child,
                                                      const int32_t *__restrict__ src, int n4) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n4) return;
    float lo[4][3], hi[4][3], org[3] = {INFINITY, INFINITY, INFINITY}, top[3] = {-INFINITY, -INFINITY, -INFINITY};
    Bvh4Node n;
    for (int c = 0; c < 4; ++c) {
        n.child[c] = child[(size_t) i * 4 + c];
        const int32_t s = src[(size_t) i * 4 + c];
        if (s < 0) continue;
        const BvhNode &b = nodes[s >> 1];
        for (int a = 0; a < 3; ++a) {
            lo[c][a] = (s & 1) ? b.lo1[a] : b.lo0[a]; hi[c][a] = (s & 1) ? b.hi1[a] : b.hi0[a];
            org[a] = fminf(org[a], lo[c][a]); top[a] = fmaxf(top[a], hi[c][a]);
        }
    }
    n.exps = 0; n.pad[0] = n.pad[1] = 0;
    for (int a = 0; a < 3; ++a) {
        n.org[a] = org[a];
        n.qlo[a] = 0xffffffffu; n.qhi[a] = 0u;               // empty slots: lower plane above the upper one
        const float ext = top[a] - org[a];
        int e = 0;
        (void) frexpf(ext * (1.f / 255.f), &e);               // ext / 255 = m * 2^e, m in [0.5, 1): 255 * 2^e >= ext
        int E = min(max(e + 127, 1), 254);
        for (;;) {
            const float scale = __int_as_float(E << 23), inv = 1.f / scale;
            uint32_t ql = 0, qh = 0; bool ok = true;
            for (int c = 0; c < 4; ++c) {
                if (src[(size_t) i * 4 + c] < 0) { ql |= 0xffu << (8 * c); continue; }
                int l = (int) floorf((lo[c][a] - org[a]) * inv), u = (int) ceilf((hi[c][a] - org[a]) * inv);
                l = max(min(l, 255), 0);
                while (l > 0 && fmaf((float) l, scale, org[a]) > lo[c][a]) --l;                   // the dequantised plane must not cut into the box
                while (u <= 255 && fmaf((float) u, scale, org[a]) < hi[c][a]) ++u;
                if (u > 255) { ok = false; break; }
                ql |= (uint32_t) l << (8 * c); qh |= (uint32_t) max(u, 0) << (8 * c);
            }
            if (ok || E >= 254) { n.qlo[a] = ql; n.qhi[a] = qh; break; }
            ++E;
        }
        n.exps |= (uint32_t) E << (8 * a);
    }
    out[i] = n;
}

// --------------------------------------------------------------------- primary-edge slot order
// A primary-edge slot draws a random silhouette edge, so consecutive slots land on unrelated pixels and
// the two Li evaluations of a wave start from 64 unrelated camera rays.  The sample streams are stateless
// (slot id -> stream), hence the slots may be evaluated in ANY order: this pre-pass computes the pixel of
// every slot (one draw + the edge CDF search) and a radix sort by pixel gives an order in which the lanes
// of a wave share their pixel neighbourhood, like the interior term.  The image is the same sum.
__global__ __launch_bounds__(kBlock) void k_primary_edge_keys(SceneView sc, RngJump jump, long long i0, long long n, uint32_t *__restrict__ keys,
                                                              uint32_t *__restrict__ vals, uint32_t invalid_key) {
    const long long j = (long long) blockIdx.x * kBlock + threadIdx.x;
    if (j >= n) return;
    Rng rng; rng.init((uint64_t) (i0 + j), jump);
    float u = rng.next(), pmf;
    const int k = sample_reuse(sc.d.prim_cmf, sc.d.prim_pmf, sc.d.prim_sum, sc.d.num_prim_edges, u, pmf);
    const float *pe = sc.d.prim_edge + (size_t) k * PSDR_PEDGE_STRIDE;
    const float px = pe[0] * (1.f - u) + pe[2] * u, py = pe[1] * (1.f - u) + pe[3] * u;
    const int W = sc.d.width, H = sc.d.height;
    const int ix = (int) floorf(px * (float) W), iy = (int) floorf(py * (float) H);
    const bool valid = ix >= 0 && ix < W && iy >= 0 && iy < H;
    // 8x8 pixel tiles, row-major inside: neighbours in the order are neighbours on the screen
    const uint32_t tile = valid ? (uint32_t) ((iy >> 3) * ((W + 7) >> 3) + (ix >> 3)) : 0u;
    keys[j] = valid ? ((tile << 6) | (uint32_t) (((iy & 7) << 3) | (ix & 7))) : invalid_key;          // (behind every pixel: primary_edge_order)
    vals[j] = (uint32_t) j;
}

#include "psdr_lbvh.h"
__global__ void k_scatter_hot(int32_t *__restrict__ map, const int32_t *__restrict__ tris, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) map[tris[i]] = i;
}

thread_local std::string g_err;
}  // namespace

namespace psdr_host {
int bvh4_refill(psdr_scene_s *h, hipStream_t s);
int bvh4_build(psdr_scene_s *h, const std::vector<BvhNode> &nodes, const std::vector<int32_t> &roots2, bool forest, hipStream_t s);
int fail(const std::string &m) { g_err = m; return 1; }

// Device memory a scratch buffer of this handle may grow to NOW: what the driver reports free, plus what the stream-ordered pool holds without using it (blocks
// hipFreeAsync returned earlier: hipMallocAsync serves from them first), plus the block that is being replaced -- minus 64 MB of headroom for the runtime.
static size_t scratch_available(size_t have) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return 0;
    int dev = 0; hipMemPool_t pool = nullptr;
    uint64_t reserved = 0, used = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess && pool != nullptr &&
        hipMemPoolGetAttribute(pool, hipMemPoolAttrReservedMemCurrent, &reserved) == hipSuccess && hipMemPoolGetAttribute(pool, hipMemPoolAttrUsedMemCurrent, &used) == hipSuccess &&
        reserved > used)
        free_b += (size_t) (reserved - used);
    (void) hipGetLastError();
    const size_t gross = free_b + have, headroom = 64ull << 20;
    return gross > headroom ? gross - headroom : 0;
}
// The largest chunk (a power-of-two fraction of `chunk`, at least 2^18 slots) whose per-slot workspace fits the device now: a launch whose default chunk
// (2^26 slots: 16-30 GB of streams / records) does not fit -- torch's caching allocator or another process holds the memory -- runs in more, smaller chunks
// instead of failing.
long long fit_chunk(long long chunk, size_t bytes_per_slot, size_t fixed_bytes, size_t have) {
    const size_t avail = scratch_available(have);
    while (chunk > (1ll << 18) && (size_t) chunk * bytes_per_slot + fixed_bytes > avail && (size_t) chunk * bytes_per_slot + fixed_bytes > have) chunk >>= 1;
    return chunk;
}
// A/B switch of the TLB experiment (option scratch_plain): blocks of >= 64 MB come from hipMalloc (one mapping the driver may back with
// large fragments) instead of the stream-ordered pool.  Process-wide; such blocks are never handed back to the pool.
static int g_scratch_plain = 0;
int scratch_reserve(void **buf, size_t *have, size_t need, hipStream_t s, const char *what) {
    if (need <= *have) return 0;
    if (g_scratch_plain != 0 && (need >= (64ull << 20) || (*buf != nullptr && *have >= (64ull << 20)))) {
        HIP_TRY(hipStreamSynchronize(s));
        if (*buf) { if (*have >= (64ull << 20)) HIP_TRY(hipFree(*buf)); else HIP_TRY(hipFreeAsync(*buf, s)); }
        *buf = nullptr; *have = 0;
        if (hipMalloc(buf, need) != hipSuccess) { (void) hipGetLastError(); *buf = nullptr; return fail(std::string("psdr: could not allocate ") + std::to_string(need >> 20) + " MB for the " + what); }
        *have = need;
        return 0;
    }
    const size_t avail = scratch_available(*have);
    if (need > avail)
        return fail(std::string("psdr: the ") + what + " needs " + std::to_string(need >> 20) + " MB of device memory, " + std::to_string(avail >> 20) +
                    " MB are available on this device: render fewer samples per call (spp range) or set the option chunk_log2");
    // the new block first where both fit (a failure then leaves the old one in place), otherwise the old one goes first
    void *nb = nullptr;
    if (*buf != nullptr && need <= scratch_available(0)) {
        if (hipMallocAsync(&nb, need, s) != hipSuccess) { (void) hipGetLastError(); nb = nullptr; }
    }
    if (*buf) HIP_TRY(hipFreeAsync(*buf, s));
    *buf = nullptr; *have = 0;
    if (nb == nullptr) {
        if (hipMallocAsync(&nb, need, s) != hipSuccess) {
            (void) hipGetLastError();
            return fail(std::string("psdr: could not allocate ") + std::to_string(need >> 20) + " MB of device memory for the " + what);
        }
    }
    *buf = nb; *have = need;
    return 0;
}

// Grid of the grid-stride kernels: at most per_cu workgroups per CU (16 by default; the forward camera kernels
// choose theirs, psdr_kernels.h camera_blocks_per_cu; the reverse kernels zero and flush a gradient cache per
// workgroup and are flat between 8 and 24).
int launch_blocks(const psdr_scene_s *h, long long n, int per_cu) {
    const int forced = h->opt.blocks_per_cu;
    const long long need = (n + kBlock - 1) / kBlock;
    const long long cap = (long long) h->num_cus * (forced ? forced : per_cu);
    return (int) std::max(1LL, std::min(need, cap));
}

// LDS plan of one launch: stacks sized by the tree depth, the rest of a 40 KB budget (4 workgroups
// per CU) filled with the top of the BVH, then the leaf triangles, then the TriangleInfo rows.
constexpr int kLdsBudget = 40 * 1024;
static bool tiny_only(const psdr_scene_s *h) { return h->n_tiny > 0 && h->n_blas == 0; }
// entries of face_cmf / face_pmf the emitters reference (host copy of emitter_i, psdr_bvh_build)
static int emitter_faces(const psdr_scene_s *h) {
    int n = 0;
    for (int e = 0; e < h->desc.num_emitters && (size_t) (e + 1) * PSDR_EMITTER_I_STRIDE <= h->emitter_i.size(); ++e)
        n = std::max(n, h->emitter_i[(size_t) e * PSDR_EMITTER_I_STRIDE + 3] + h->emitter_i[(size_t) e * PSDR_EMITTER_I_STRIDE + 2]);
    return n;
}
// A scene without a tree whose small tables fit the LDS block of the kSceneTiny kernel instances (a few KB): every launch on it stages
// them (plan_lds) and variant_of picks those instances.  PSDR_TINY_VARIANTS=0: the general instances (A/B, tools).
constexpr int kLdsTexels = 256, kLdsTexelTangents = 3;
// the mesh-level tables (mesh -> bsdf / emitter, BSDF and emitter records, the emitters' face distributions) fit the LDS block of the kernels
static bool small_tables_fit(const psdr_scene_s *h) {
    const psdr_scene_desc &d = h->desc;
    return d.env_emitter < 0 && d.num_meshes <= 64 && d.num_bsdfs <= 32 && d.num_emitters >= 0 && d.num_emitters <= 8 &&
           (d.num_emitters == 0 || (d.face_cmf && d.face_pmf)) && emitter_faces(h) <= 64;
}
static bool tiny_tables_ok(const psdr_scene_s *h) {
    return h->opt.tiny_variants != 0 && tiny_only(h) && small_tables_fit(h);
}
// The kernels of a two-level scene read the mesh-level tables from LDS and have no other path (psdr_device.h Tab<FL>::lds_small): such a tree
// is only built, and only kept, while they fit (psdr_bvh_build, ensure_tree_kind).
static bool forest_tables(const psdr_scene_s *h) { return PSDR_FOREST_LDS_TABLES && h->n_blas > 0; }
// Which tree the launches on this scene walk: the 4-wide quantised tree exactly where the scene's kernel variant is compiled for it (flag set 6:
// rough conductor + two-level tree, psdr_variant.hip) -- measured 5-10 % ahead on the 50 k-triangle interior, level or behind elsewhere
// (profiles/r03_bvh4_ab.txt).  k_trace (this unit) carries both walks and follows the scene.
static bool use_wide_tree(const psdr_scene_s *h, bool forest) {
    if (h->opt.wide == 0) return false;                 // never (the variant-6 kernels then cannot run: tools only)
    return forest && h->has_rough && h->desc.env_emitter < 0;
}

// traversal-stack entries per lane: none when no launch on this scene ever walks a tree
static int stack_entries_of(const psdr_scene_s *h) { return tiny_only(h) ? 0 : (h->wide ? h->stack_need4 + 1 : std::min(kBvhStack, h->bvh_depth + 2)); }
static int staged_nodes_of(const psdr_scene_s *h) { return h->wide ? h->num_nodes4 : h->num_nodes; }
int plan_lds(const psdr_scene_s *h, LaunchCtx &cx, int reserved) {
    // the plain diffuse variant runs renderC at 5 workgroups per CU: 28 KB each (C4 PathTracer(3) 40.0 -> 37.6 ms,
    // C3 3.26 -> 3.15 ms; 32 KB is already one workgroup less)
    const int forced = h->opt.lds_budget;
    const bool lean = !h->has_rough && h->desc.env_emitter < 0;
    const int budget = forced ? forced : (lean ? 28 * 1024 : kLdsBudget);
    const int stack_bytes = stack_entries_of(h) * kBlock * 4;
    int room = std::max(0, budget - reserved - stack_bytes);
    // a scene without a tree (all primitives in the kernel arguments) ALWAYS stages its <= 16 TriangleInfo rows (1.5 KB): the kSceneTiny
    // kernel instances have no global-memory fallback for them (psdr_device.h load_tri_f)
    if (tiny_only(h)) room = std::max(room, staged_nodes_of(h) * kLdsNodeStride + h->num_btris * 48 + h->desc.num_tris * 96);
    SceneView &sc = cx.sc;
    int off = 0;
    sc.n_lnodes = std::min(staged_nodes_of(h), room / kLdsNodeStride); sc.off_lnodes = off; off += sc.n_lnodes * kLdsNodeStride; room -= sc.n_lnodes * kLdsNodeStride;
    sc.n_lbtris = (sc.n_lnodes == staged_nodes_of(h)) ? std::min(h->num_btris, room / 48) : 0;
    sc.off_lbtris = off; off += sc.n_lbtris * 48; room -= sc.n_lbtris * 48;
    sc.n_ltri = (sc.n_lbtris == h->num_btris) ? std::min(h->desc.num_tris, room / 96) : 0;
    sc.off_ltri = off; off += sc.n_ltri * 96;
    sc.off_lprim = off; off += h->n_tiny * kTinyHitWords * 4;          // hit rows of the kernel-argument primitives (resolve_tiny_hit)
    // the small tables of a scene without a tree (psdr_device.h Tab<FL>, staged by setup_lds): 16-byte aligned blocks
    sc.lt_trimesh = sc.lt_meshbsdf = sc.lt_meshemitter = sc.lt_bsdf = sc.lt_emf = sc.lt_emi = sc.lt_fcmf = sc.lt_fpmf = sc.lt_uv = sc.lt_tex = sc.lt_ecmf = sc.lt_epmf = -1;
    sc.lt_nfaces = 0; sc.lt_occ = -1; sc.occ = nullptr;
    const bool all_tables = tiny_tables_ok(h);
    if (all_tables || forest_tables(h)) {
        auto take = [&](int words) { const int o = off; off += (words * 4 + 15) / 16 * 16; return o; };
        const psdr_scene_desc &d = h->desc;
        sc.lt_nfaces = emitter_faces(h);
        if (all_tables) sc.lt_trimesh = take(d.num_tris);          // per-triangle tables: only where every triangle is a kernel-argument primitive
        sc.lt_meshbsdf = take(d.num_meshes); sc.lt_meshemitter = take(d.num_meshes);
        sc.lt_bsdf = take(std::max(d.num_bsdfs, 1) * PSDR_BSDF_STRIDE);
        sc.lt_emf = take(d.num_emitters * PSDR_EMITTER_F_STRIDE); sc.lt_emi = take(d.num_emitters * PSDR_EMITTER_I_STRIDE);
        sc.lt_fcmf = take(sc.lt_nfaces); sc.lt_fpmf = take(sc.lt_nfaces);
        if (d.num_emitters > 1 && d.emitter_cmf && d.emitter_pmf) { sc.lt_ecmf = take(d.num_emitters); sc.lt_epmf = take(d.num_emitters); }
        if (all_tables && d.tri_uv) sc.lt_uv = take(d.num_tris * PSDR_TRIUV_STRIDE);
        if (d.num_texels > 0 && d.num_texels <= kLdsTexels) sc.lt_tex = take(d.num_texels * (1 + kLdsTexelTangents));      // value pool + up to 3 tangent pools (forward mode)
        if (all_tables && h->have_occ && h->d_occ != nullptr) { sc.occ = h->d_occ; sc.lt_occ = take(d.num_tris * d.num_tris); }   // occluder rows of the light rays (<= 32 x 32 words)
    }
    sc.lt_end = off;
    cx.off_stack = off;
    return off + stack_bytes;
}
int lds_bytes(const LaunchCtx &cx, const psdr_scene_s *h) { return cx.off_stack + stack_entries_of(h) * kBlock * 4; }

// the part of the tree that travels in the kernel arguments (tiny scenes, two-level trees)
void fill_top(const psdr_scene_s *h, SceneView &sc) {
    sc.n_tiny = h->n_tiny; sc.aa_cnt = h->n_tiny > 0 ? h->aa_cnt : 0;
    sc.emit_rows = (h->n_tiny > 0 && h->opt.emitter_pretest != 0) ? h->emit_rows : 0u;
    std::memcpy(sc.tiny, h->tiny, sizeof(h->tiny));
    std::memcpy(sc.tiny_meta, h->tiny_meta, sizeof(h->tiny_meta));
    sc.n_blas = h->n_blas;
    std::memcpy(sc.blas_lo, h->blas_lo, sizeof(h->blas_lo));
    std::memcpy(sc.blas_hi, h->blas_hi, sizeof(h->blas_hi));
    for (int k = 0; k < h->n_blas; ++k) std::memcpy(&sc.blas_hi[k].w, &h->blas_root4[k], 4);
    sc.nodes4 = h->wide ? h->d_nodes4 : nullptr; sc.root4 = h->root4;
}

int make_ctx(psdr_scene_s *h, const psdr_render_opts *o, int sampler, LaunchCtx &cx) {
    if (!h->have_tables) return fail("Scene not loaded yet!");
    if (!h->have_bvh) return fail("Input scene must be configured!");
    if (o->integrator != PSDR_INTEGRATOR_FIELD && h->desc.num_emitters <= 0) return fail("No Emitter!");
    if (o->integrator == PSDR_INTEGRATOR_DIRECT && !(o->bsdf_samples >= 0 && o->light_samples >= 0 && o->bsdf_samples + o->light_samples > 0))
        return fail("DirectIntegrator: bsdf_samples + light_samples must be positive");
    cx.sc.d = h->desc; cx.sc.nodes = h->d_nodes; cx.sc.btris = h->d_btris; cx.sc.root = h->root;
    fill_top(h, cx.sc);
    plan_lds(h, cx);
    cx.sc.literal_forms = (o->flags & PSDR_FLAG_LITERAL_FORMS) ? 1 : 0;
    cx.lp = LiParams{o->integrator, o->bsdf_samples, o->light_samples, o->max_depth, o->hide_emitters, o->field};
    cx.jump = make_rng_jump(o->rng_offset[sampler]);
    return 0;
}

int check_counts(const psdr_scene_s *h, const psdr_render_opts *o) {
    const long long WH = (long long) h->desc.width * h->desc.height;
    if (WH <= 0) return fail("Invalid film resolution");
    if (WH * std::max(o->spp, 1) > 0x7fffffffLL) return fail("Too many samples (width*height*spp > INT_MAX)");   // integrator.cpp:74
    if (o->spp_begin < 0 || o->spp_end > o->spp || o->spp_begin > o->spp_end) return fail("Invalid spp shard range");
    if (o->sppe_begin < 0 || o->sppe_end > o->sppe || o->sppe_begin > o->sppe_end) return fail("Invalid sppe shard range");
    if (o->sppse_begin < 0 || o->sppse_end > o->sppse || o->sppse_begin > o->sppse_end) return fail("Invalid sppse shard range");
    return 0;
}

// Strategy choice: a PURE FUNCTION of the scene and the options (round 3; it used to follow the survival ratio the previous PathTracer
// call on the handle had measured, so the same call could run either strategy depending on the call history -- and the two agree only
// to the rounding of separately compiled fp32 kernels, up to 1e-3 on a low-spp GGX scene where one firefly sample flips).
// Measured on MI355X (tools/perf_cases.py): in a closed room almost every path survives every bounce, compaction buys nothing and the
// fused kernel is 1.1-1.9x ahead; in an open scene (bunny_light: 2.5 of 7 possible rays per path) the wavefront is 1.35-1.5x ahead.
// "Room" = what psdr_bvh_build recognised as one: all primitives in the kernel arguments (the Cornell box) or a two-level tree (>= 6 inline
// wall triangles around the meshes); everything else (objects under a light, height fields, environment-lit scenes) counts as open.
bool open_scene(const psdr_scene_s *h) { return !(h->n_blas > 0 || (h->n_tiny > 0 && h->n_blas == 0)); }
bool use_wavefront(const psdr_scene_s *h, const psdr_render_opts *o) {
    if (o->integrator != PSDR_INTEGRATOR_PATH || o->max_depth > 250) return false;
    if (h->desc.num_tris >= (1 << 29)) return false;                // a stream record keeps its triangle in 29 bits (psdr_kernels.h kWfTriMask)
    if (o->flags & PSDR_FLAG_FUSED) return false;
    if (o->flags & PSDR_FLAG_WAVEFRONT) return true;
    if (o->max_depth < 2) return false;
    if (open_scene(h)) return true;                                 // most paths die early, compaction pays
    const long long n = (long long) h->desc.width * h->desc.height * (o->spp_end - o->spp_begin);
    // two-level scene (a room with objects): the TRACED wavefront -- in the fused kernel a wave pays the slowest lane's tree walk at every
    // closest_hit although only a fifth of the rays enter a tree; with the walks in the dense trace kernel between the stages C4's shard takes
    // 18.2 ms against 25.8 fused (21.8 class-binned), the 50 k-triangle interior 2.8 against 4.6, cbox_bunny at 4 M slots 1.6 against 2.1
    // (profiles/r04_traced_first.txt); below 2^16 slots the seven launches cost more than the walks
    if (traced_wavefront(h)) return n >= (1ll << 16);
    // without the trace kernel (PSDR_WF_TRACED=0): the class-binned streams (psdr_kernels.h) are ahead of the fused kernel on large launches only --
    // C4 shard (67 M slots) 27.5 against 29.6 ms; at 4 M slots their extra launches and stream traffic cost more than they save (3.7 against 2.9 ms)
    return h->n_blas > 0 && h->wf_binned && n >= (1ll << 25);
}

SinkLayout make_sink_layout(const psdr_scene_s *h, const psdr_grads *g) {
    SinkLayout L{};
    int off = 0;
    L.cam_off = off; off += 16;
    if (g->g_env_f && h->desc.env_emitter >= 0) { L.env_off = off; L.env_n = PSDR_ENV_WORDS; off += PSDR_ENV_WORDS; }
    const int nt = h->desc.num_texels, nr = h->desc.num_emitters * 3;
    if (g->g_texels && nt > 0 && nt <= 2048) { L.tex_off = off; L.tex_n = nt; off += nt; }
    if (g->g_emitter_rad && nr > 0 && nr <= 256) { L.rad_off = off; L.rad_n = nr; off += nr; }
    if (g->g_tri_info && h->hot_rows > 0) {
        const int rows = std::min(h->hot_rows, (kSinkCacheWords - off) / PSDR_TRI_STRIDE);
        // slots >= rows (cache too small for all of them) fall back to global atomics in add_tri
        if (rows > 0) { L.hot_off = off; L.hot_rows = rows; L.hot_map = h->d_hot_map; L.hot_tris = h->d_hot_tris; off += rows * PSDR_TRI_STRIDE; }
    }
    L.total = off;
    L.stride = off | 1;
    L.rep = 1;
    // 2 copies bought 6 % on C2, more nothing: 4 at most, so that a small scene's cache stays a few KB of LDS
    const int max_rep = h->opt.sink_rep;
    while (L.rep * 2 <= max_rep && L.rep * 2 * L.stride <= kSinkCacheWords) L.rep *= 2;
    // lane-private rows: the leading hot slots are emitter 0's triangles (build_bvh)
    const bool priv_on = h->opt.sink_private != 0;
    L.priv_tri[0] = L.priv_tri[1] = -1; L.priv_slot[0] = L.priv_slot[1] = 0; L.priv_emitter = -1;
    if (priv_on && L.hot_rows > 0 && h->desc.num_emitters > 0 && h->desc.env_emitter != 0) {
        const int32_t *ei = h->emitter_i.data();
        const int n = std::min(std::min(ei[2], 2), L.hot_rows);
        for (int r = 0; r < n; ++r) { L.priv_tri[r] = ei[1] + r; L.priv_slot[r] = h->hot_identity ? ei[1] + r : r; }
        L.priv_rows = n;
        if (n > 0) { L.priv_off = (L.rep * L.stride + 3) / 4 * 4; if (L.rad_n) L.priv_emitter = 0; }
    }
    return L;
}

// The tree kind follows the kernel variant, the variant follows material_mask -- which psdr_scene_set_tables can change without a rebuild (a
// scene that gains or loses its rough conductors): bring the 4-wide tree in line before the launch (one read-back of the BVH2 nodes + the collapse).
int ensure_tree_kind(psdr_scene_s *h, hipStream_t s) {
    if (!h->have_bvh || h->num_nodes <= 0) return 0;
    // a two-level tree stands only while the mesh-level tables fit the LDS block of its kernels (psdr_scene_set_tables may have grown them)
    if (forest_tables(h) && !small_tables_fit(h)) return psdr_bvh_build(h, s);
    const bool forest = h->n_blas > 0;
    if (use_wide_tree(h, forest) == h->wide) return 0;
    std::vector<BvhNode> host_nodes((size_t) h->num_nodes);
    HIP_TRY(hipMemcpyAsync(host_nodes.data(), h->d_nodes, host_nodes.size() * sizeof(BvhNode), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    std::vector<int32_t> roots;
    if (forest) for (int k = 0; k < h->n_blas; ++k) { int32_t r; std::memcpy(&r, &h->blas_lo[k].w, 4); roots.push_back(r); }
    else roots.push_back(h->root);
    return bvh4_build(h, host_nodes, roots, forest, s);
}

int begin_call(psdr_scene_s *h, hipStream_t s) {
    h->last_stream = s;
    if (int rc = ensure_tree_kind(h, s)) return rc;
    h->slots[0] = h->slots[1] = h->slots[2] = 0; h->last_path_depth = 0;
    HIP_TRY(hipMemsetAsync(h->d_counters, 0, sizeof(unsigned long long) * kRayCounters * kRayCounterStride, s));
    return 0;
}


// Order in which the primary-edge slots [i0, i0 + n) are evaluated (see k_primary_edge_keys); *order = nullptr
// keeps the natural order (tiny launches, or PSDR_SORT_EDGES=0).
int primary_edge_order(psdr_scene_s *h, const LaunchCtx &cx, long long i0, long long n, const uint32_t **order, hipStream_t s) {
    *order = nullptr;
    if (!h->sort_edges || n < 65536 || n > 0x7fffffffLL) return 0;
    size_t temp = 0;
    // key = (8x8 tile, pixel in the tile); a slot outside the film gets the one bit above them.  The sort runs over the bits the image needs (1024^2: 21 of 32)
    const unsigned long long tiles = (unsigned long long) ((cx.sc.d.width + 7) >> 3) * (unsigned long long) ((cx.sc.d.height + 7) >> 3);
    unsigned tile_bits = 0;
    while ((1ull << tile_bits) < tiles) ++tile_bits;
    if (tile_bits + 6 >= 32) return 0;                                    // (an image of > 2^25 tiles: natural order)
    const unsigned end_bit = tile_bits + 7;
    const uint32_t invalid_key = 1u << (tile_bits + 6);
    uint32_t *nul = nullptr;
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, temp, nul, nul, nul, nul, (size_t) n, 0, end_bit, s));
    const size_t need = 4 * sizeof(uint32_t) * (size_t) n + temp + 256;
    if (int rc = scratch_reserve(&h->d_sort, &h->sort_bytes, need, s, "primary-edge sort scratch")) return rc;
    uint32_t *k_in = reinterpret_cast<uint32_t *>(h->d_sort), *k_out = k_in + n, *v_in = k_out + n, *v_out = v_in + n;
    void *tmp = reinterpret_cast<void *>((reinterpret_cast<uintptr_t>(v_out + n) + 255) & ~(uintptr_t) 255);
    hipLaunchKernelGGL(k_primary_edge_keys, dim3((unsigned) ((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, cx.sc, cx.jump, i0, n, k_in, v_in, invalid_key);
    HIP_TRY(hipGetLastError());
    HIP_TRY(rocprim::radix_sort_pairs(tmp, temp, k_in, k_out, v_in, v_out, (size_t) n, 0, end_bit, s));
    *order = v_out;
    return 0;
}

// ---- device tree (psdr_lbvh.h).  Layout of the scratch block: what the refit needs first, the build temporaries behind.
struct LbvhScratch {
    LbvhInfo *info; uint32_t *bounds; int32_t *node_parent, *leaf_parent; uint32_t *arrivals;
    uint32_t *keys, *keys2; int32_t *vals, *vals2; void *sort_tmp; size_t sort_tmp_bytes, total;
};
LbvhScratch lbvh_layout(void *base, int T, size_t sort_tmp_bytes) {
    LbvhScratch L{};
    size_t off = 0;
    auto take = [&](size_t bytes) { void *p = base ? (char *) base + off : nullptr; off += (bytes + 255) & ~(size_t) 255; return p; };
    L.info = (LbvhInfo *) take(sizeof(LbvhInfo)); L.bounds = (uint32_t *) take(6 * 4);
    L.node_parent = (int32_t *) take((size_t) T * 4); L.leaf_parent = (int32_t *) take((size_t) T * 4);
    L.arrivals = (uint32_t *) take((size_t) T * 4);
    L.keys = (uint32_t *) take((size_t) T * 4); L.keys2 = (uint32_t *) take((size_t) T * 4);
    L.vals = (int32_t *) take((size_t) T * 4); L.vals2 = (int32_t *) take((size_t) T * 4);
    L.sort_tmp = take(sort_tmp_bytes); L.sort_tmp_bytes = sort_tmp_bytes;
    L.total = off;
    return L;
}
int lbvh_fit(psdr_scene_s *h, const LbvhScratch &L, float pad, hipStream_t s) {
    const int T = h->desc.num_tris;
    HIP_TRY(hipMemsetAsync(L.arrivals, 0, (size_t) T * 4, s));
    HIP_TRY(hipMemsetAsync(h->d_refit_area, 0, sizeof(float), s));
    hipLaunchKernelGGL(k_lbvh_fit, dim3((T + kBlock - 1) / kBlock), dim3(kBlock), 0, s, T, h->d_nodes, h->d_btris, L.node_parent, L.leaf_parent, L.arrivals,
                       h->d_refit_area, pad);
    HIP_TRY(hipGetLastError());
    return 0;
}
// refit of a device-built tree: leaf triangles re-read from the table (k_refit_leaves), then the bottom-up pass again
int lbvh_refit(psdr_scene_s *h, hipStream_t s) {
    const int T = h->tree_tris;
    const LbvhScratch L = lbvh_layout(h->d_lbvh, T, 0);
    hipLaunchKernelGGL(k_refit_leaves, dim3((T + kBlock - 1) / kBlock), dim3(kBlock), 0, s, h->d_btris, h->desc.tri_info, T);
    return lbvh_fit(h, L, h->bvh_pad, s);
}
// Builds the tree on the device.  fallback = true (and 0 returned): the host builder has to do it (too few triangles, tree
// deeper than the traversal stack).
int lbvh_build(psdr_scene_s *h, hipStream_t s, bool &fallback) {
    const int T = h->desc.num_tris;
    fallback = T <= kLbvhLeaf;
    if (fallback) return 0;
    size_t t1 = 0, t2 = 0;
    uint32_t *nk = nullptr; int32_t *nv = nullptr;
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, t1, nk, nk, nv, nv, (size_t) T, 0, 30, s));
    HIP_TRY(rocprim::radix_sort_pairs_desc(nullptr, t2, nk, nk, nv, nv, (size_t) T, 0, 32, s));
    const size_t sort_tmp = std::max(t1, t2);
    const size_t need = lbvh_layout(nullptr, T, sort_tmp).total;
    if (need > h->lbvh_bytes) {
        if (h->d_lbvh) (void) hipFree(h->d_lbvh);
        h->d_lbvh = nullptr; h->lbvh_bytes = 0;
        HIP_TRY(hipMalloc(&h->d_lbvh, need));
        h->lbvh_bytes = need;
    }
    const LbvhScratch L = lbvh_layout(h->d_lbvh, T, sort_tmp);
    if ((size_t) (T - 1) > h->cap_nodes) {
        if (h->d_nodes) (void) hipFree(h->d_nodes);
        h->cap_nodes = std::max<size_t>((size_t) T, 16);
        HIP_TRY(hipMalloc(&h->d_nodes, h->cap_nodes * sizeof(BvhNode)));
    }
    if ((size_t) T * 3 > h->cap_btris) {
        if (h->d_btris) (void) hipFree(h->d_btris);
        h->cap_btris = (size_t) T * 3;
        HIP_TRY(hipMalloc(&h->d_btris, h->cap_btris * sizeof(float4)));
    }
    if (!h->d_refit_area) HIP_TRY(hipMalloc(&h->d_refit_area, sizeof(float)));
    const dim3 gT((T + kBlock - 1) / kBlock), blk(kBlock);
    HIP_TRY(hipMemsetAsync(L.info, 0, sizeof(LbvhInfo), s));
    HIP_TRY(hipMemsetAsync(L.bounds, 0xff, 3 * 4, s));
    HIP_TRY(hipMemsetAsync(L.bounds + 3, 0, 3 * 4, s));
    hipLaunchKernelGGL(k_lbvh_bounds, gT, blk, 0, s, h->desc.tri_info, T, L.bounds, L.info);
    hipLaunchKernelGGL(k_lbvh_codes, gT, blk, 0, s, h->desc.tri_info, T, L.bounds, L.keys, L.vals, L.info);
    HIP_TRY(hipGetLastError());
    size_t tmp_bytes = L.sort_tmp_bytes;
    HIP_TRY(rocprim::radix_sort_pairs(L.sort_tmp, tmp_bytes, L.keys, L.keys2, L.vals, L.vals2, (size_t) T, 0, 30, s));
    HIP_TRY(hipMemsetAsync(L.leaf_parent, 0xff, (size_t) T * 4, s));
    hipLaunchKernelGGL(k_lbvh_tris, gT, blk, 0, s, h->desc.tri_info, T, L.vals2, h->d_btris);
    hipLaunchKernelGGL(k_lbvh_hierarchy, gT, blk, 0, s, L.keys2, T, h->d_nodes, L.node_parent, L.leaf_parent);
    HIP_TRY(hipGetLastError());
    // the padding of the leaf boxes is a function of the scene extent: fetch it with the validity flag before the fit
    LbvhInfo info{};
    HIP_TRY(hipMemcpyAsync(&info, L.info, sizeof(info), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (info.bad) return fail("psdr_bvh_build: non-finite vertex");
    if (int rc = lbvh_fit(h, L, info.pad, s)) return rc;
    hipLaunchKernelGGL(k_lbvh_depth, gT, blk, 0, s, T, L.node_parent, L.leaf_parent, L.info);
    // hot rows of the gradient cache: emitter triangles, then the largest triangles (area keys sorted on the device)
    constexpr int kMaxHotRows = 200;
    hipLaunchKernelGGL(k_lbvh_area_keys, gT, blk, 0, s, h->desc.tri_info, T, L.keys, L.vals);
    HIP_TRY(hipGetLastError());
    tmp_bytes = L.sort_tmp_bytes;
    HIP_TRY(rocprim::radix_sort_pairs_desc(L.sort_tmp, tmp_bytes, L.keys, L.keys2, L.vals, L.vals2, (size_t) T, 0, 32, s));
    float area = 0.f;
    const int top = std::min(T, kMaxHotRows);
    std::vector<int32_t> by_area((size_t) top);
    HIP_TRY(hipMemcpyAsync(&info, L.info, sizeof(info), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&area, h->d_refit_area, sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(by_area.data(), L.vals2, (size_t) top * 4, hipMemcpyDeviceToHost, s));
    h->emitter_i.assign((size_t) std::max(h->desc.num_emitters, 0) * PSDR_EMITTER_I_STRIDE, 0);
    if (h->desc.num_emitters > 0 && h->desc.emitter_i)
        HIP_TRY(hipMemcpyAsync(h->emitter_i.data(), h->desc.emitter_i, h->emitter_i.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (info.depth > kBvhStack - 2) { fallback = true; return 0; }
    std::vector<int32_t> tris;
    {
        auto add = [&](int t) {
            if (t < 0 || t >= T || (int) tris.size() >= kMaxHotRows) return;
            if (std::find(tris.begin(), tris.end(), t) == tris.end()) tris.push_back(t);
        };
        for (int e = 0; e < h->desc.num_emitters; ++e) {
            const int32_t *ei = h->emitter_i.data() + (size_t) e * PSDR_EMITTER_I_STRIDE;
            for (int f = 0; f < ei[2] && f < 64; ++f) add(ei[1] + f);
        }
        for (int i = 0; i < top; ++i) add(by_area[(size_t) i]);
    }
    if ((size_t) T > h->hot_cap) {
        if (h->d_hot_map) (void) hipFree(h->d_hot_map);
        if (h->d_hot_tris) (void) hipFree(h->d_hot_tris);
        h->hot_cap = (size_t) T;
        HIP_TRY(hipMalloc(&h->d_hot_map, h->hot_cap * sizeof(int32_t)));
        HIP_TRY(hipMalloc(&h->d_hot_tris, kMaxHotRows * sizeof(int32_t)));
    }
    HIP_TRY(hipMemsetAsync(h->d_hot_map, 0xff, (size_t) T * sizeof(int32_t), s));
    HIP_TRY(hipMemcpyAsync(h->d_hot_tris, tris.data(), tris.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_scatter_hot, dim3(1), dim3(256), 0, s, h->d_hot_map, h->d_hot_tris, (int) tris.size());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));            // `tris` dies at return
    h->hot_rows = (int) tris.size(); h->hot_identity = false;
    h->root = 0; h->bvh_depth = info.depth; h->num_nodes = T - 1; h->num_btris = T;
    h->n_tiny = 0; h->aa_cnt = 0; h->n_blas = 0; h->n_inline = 0; h->have_occ = false; h->emit_rows = 0;
    if (use_wide_tree(h, false)) {
        // the 4-wide tree over the device-built BVH2: its topology is decided on the host (one read-back of the node array; the collapse is
        // O(T)): 263 k triangles +~15 ms on top of the 2 ms device build -- only where a launch would walk it
        std::vector<BvhNode> host_nodes((size_t) T - 1);
        HIP_TRY(hipMemcpyAsync(host_nodes.data(), h->d_nodes, host_nodes.size() * sizeof(BvhNode), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (int rc = bvh4_build(h, host_nodes, std::vector<int32_t>{0}, false, s)) return rc;
    } else { h->wide = false; h->num_nodes4 = 0; h->stack_need4 = 0; }
    h->tree_tris = T; h->refits_since_build = 0; h->num_builds++; h->bvh_pad = info.pad; h->built_area = area;
    h->level_start.clear();
    h->refit_ok = true; h->lbvh = true; h->have_bvh = true;
    return 0;
}

// The 4-wide tree over the BVH2 now on the device (`nodes` = its host copy, `roots2` = its roots): collapse on the host, topology up,
// boxes by k_bvh4_fill.  bvh4_refill: the boxes again after a refit of the BVH2.
int bvh4_refill(psdr_scene_s *h, hipStream_t s) {
    if (h->num_nodes4 <= 0) return 0;
    hipLaunchKernelGGL(k_bvh4_fill, dim3((h->num_nodes4 + kBlock - 1) / kBlock), dim3(kBlock), 0, s, h->d_nodes4, h->d_nodes, h->d_topo4,
                       h->d_topo4 + (size_t) h->num_nodes4 * 4, h->num_nodes4);
    HIP_TRY(hipGetLastError());
    return 0;
}
int bvh4_build(psdr_scene_s *h, const std::vector<BvhNode> &nodes, const std::vector<int32_t> &roots2, bool forest, hipStream_t s) {
    h->wide = use_wide_tree(h, forest);
    // the dense trace kernel of the traced wavefront walks the 4-wide forest of every two-level scene, whatever its render kernels walk
    if (!h->wide && !forest) { h->num_nodes4 = 0; h->stack_need4 = 0; return 0; }
    Bvh4Topology tp;
    collapse_bvh4(nodes, roots2, tp);
    h->num_nodes4 = tp.n4; h->stack_need4 = tp.stack_need;
    h->root4 = roots2.size() == 1 ? tp.roots[0] : 0;
    for (size_t k = 0; k < roots2.size() && k < (size_t) kMaxBlas; ++k) h->blas_root4[k] = tp.roots[k];
    if (tp.n4 == 0) return 0;
    if ((size_t) tp.n4 > h->cap_nodes4) {
        if (h->d_nodes4) (void) hipFree(h->d_nodes4);
        if (h->d_topo4) (void) hipFree(h->d_topo4);
        h->d_nodes4 = nullptr; h->d_topo4 = nullptr;
        h->cap_nodes4 = (size_t) tp.n4;
        HIP_TRY(hipMalloc(&h->d_nodes4, h->cap_nodes4 * sizeof(Bvh4Node)));
        HIP_TRY(hipMalloc(&h->d_topo4, h->cap_nodes4 * 8 * sizeof(int32_t)));
    }
    HIP_TRY(hipMemcpyAsync(h->d_topo4, tp.child.data(), (size_t) tp.n4 * 4 * sizeof(int32_t), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(h->d_topo4 + (size_t) tp.n4 * 4, tp.src.data(), (size_t) tp.n4 * 4 * sizeof(int32_t), hipMemcpyHostToDevice, s));
    if (int rc = bvh4_refill(h, s)) return rc;
    HIP_TRY(hipStreamSynchronize(s));              // the topology vectors die at return
    return 0;
}

// ---- the dense trace kernel of the traced wavefront
bool traced_wavefront(const psdr_scene_s *h) { return h->traced_enabled && h->n_blas > 0 && h->num_nodes4 > 0 && h->d_nodes4 != nullptr; }
int launch_wf_trace(psdr_scene_s *h, const float4 *req, const int32_t *count, long long sub_cap, float4 *hit, hipStream_t s, bool ign) {
    TraceArgs a{};
    a.sec_edge_faces = h->desc.sec_edge_faces;
    if (ign && !a.sec_edge_faces) ign = false;
    std::memcpy(a.blas_lo, h->blas_lo, sizeof(a.blas_lo));
    std::memcpy(a.blas_hi, h->blas_hi, sizeof(a.blas_hi));
    for (int k = 0; k < h->n_blas; ++k) std::memcpy(&a.blas_hi[k].w, &h->blas_root4[k], 4);
    a.nodes4 = h->d_nodes4; a.btris = h->d_btris; a.req = req; a.count = count; a.hit = hit; a.sub_cap = sub_cap; a.n_blas = h->n_blas;
    // LDS: stacks first in the budget (S + 1 columns: the unconditional stores of a node step may touch the entry above the top), the tree in the rest.
    // TWO workgroups per CU (8 waves per SIMD, <= 64 VGPRs), each with half of the LDS -- stack columns of 8 entries (deeper ones, rare, in the
    // overflow columns), fewer staged nodes -- where the walk gains more from the second set of waves than it loses to the smaller staging:
    // a forest that does not fit the LDS anyway (C5: trace stage 326 -> 307 us) or a large launch (C4 shard: 754 -> 690 us); the 4 M-slot launch on
    // the bunny, whose whole tree fits one workgroup's LDS, is 2.5 % faster with one (profiles/r04_trace_wg2_abk.txt).  Option trace_wg2: -1 this
    // rule, 0 never, n > 0 always with columns of n entries.
    const int full_S = std::min(h->stack_need4, kTraceStackMax);
    const bool tree_fits = (long long) h->num_nodes4 * kTraceNodeStride + (long long) (full_S + 1) * kTraceBlock * 4 + 1024 <= (long long) h->lds_limit;
    const bool wg2 = h->lds_limit >= 160 * 1024 && (h->opt.trace_wg2 > 0 || (h->opt.trace_wg2 < 0 && (!tree_fits || sub_cap * kWfSub >= (1ll << 25))));
    const int S = std::min(h->stack_need4, wg2 ? std::min(h->opt.trace_wg2 > 0 ? h->opt.trace_wg2 : 8, kTraceStackMax) : kTraceStackMax);
    const bool ovf = h->stack_need4 > S;
    const int stack_bytes = (S + 1) * kTraceBlock * 4;
    const int room = (wg2 ? h->lds_limit / 2 : h->lds_limit) - 1024 - stack_bytes;                 // 1 KB: the kernel's static LDS
    a.n_lnodes = std::max(0, std::min(h->num_nodes4, room / kTraceNodeStride));
    a.off_stack = a.n_lnodes * kTraceNodeStride;
    a.stack_entries = S;
    const int grid = wg2 ? 2 * h->num_cus : h->num_cus;
    if (ovf) {
        a.ovf_stride = grid * kTraceBlock;
        const size_t need = (size_t) a.ovf_stride * (size_t) (h->stack_need4 - S) * sizeof(int32_t);
        if (int rc = scratch_reserve(reinterpret_cast<void **>(&h->d_trace_ovf), &h->trace_ovf_bytes, need, s, "trace kernel's overflow stack columns")) return rc;
        a.ovf = h->d_trace_ovf;
    }
    const int dyn = a.off_stack + stack_bytes;
#define PSDR_LAUNCH_TRACE_I(MULTI, OVF, IGN, WPE)                                                                                                           \
    do { static bool attr_set = false;                                                                                                                      \
         if (!attr_set) { HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_wf_trace<MULTI, OVF, IGN, WPE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024)); attr_set = true; } \
         hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wf_trace<MULTI, OVF, IGN, WPE>), dim3(grid), dim3(kTraceBlock), dyn, s, a); } while (0)
#define PSDR_LAUNCH_TRACE(MULTI, OVF) do { if (ign) { if (wg2) PSDR_LAUNCH_TRACE_I(MULTI, OVF, true, 8); else PSDR_LAUNCH_TRACE_I(MULTI, OVF, true, 4); } else if (wg2) PSDR_LAUNCH_TRACE_I(MULTI, OVF, false, 8); else PSDR_LAUNCH_TRACE_I(MULTI, OVF, false, 4); } while (0)
    if (h->n_blas > 1) { if (ovf) PSDR_LAUNCH_TRACE(true, true); else PSDR_LAUNCH_TRACE(true, false); }
    else { if (ovf) PSDR_LAUNCH_TRACE(false, true); else PSDR_LAUNCH_TRACE(false, false); }
#undef PSDR_LAUNCH_TRACE_I
#undef PSDR_LAUNCH_TRACE
    HIP_TRY(hipGetLastError());
    return 0;
}

// hit rows [slots * rays_per_slot], one mask word per slot, the request queue (64 sub-queues, every ray can be a request) and its counters (zeroed here)
int probe_buffers(psdr_scene_s *h, long long slots, int rays_per_slot, ProbeBuffers &pb, hipStream_t s, int per_cu) {
    const long long rays = slots * rays_per_slot;
    const long long blocks = ((long long) launch_blocks(h, slots, per_cu) + kWfSub - 1) / kWfSub * kWfSub;
    const long long trips = (slots + blocks * kBlock - 1) / (blocks * kBlock);
    pb.sub_cap = (blocks / kWfSub) * trips * kBlock * rays_per_slot;          // block b appends to queue b % kWfSub
    const size_t cnt_bytes = (size_t) kWfSub * kWfCountStride * sizeof(int32_t);
    const size_t need = cnt_bytes + (size_t) rays * sizeof(float4) + (((size_t) slots * 4 + 255) & ~(size_t) 255) + (size_t) 2 * pb.sub_cap * kWfSub * sizeof(float4);
    if (int rc = scratch_reserve(&h->d_probe, &h->probe_bytes, need, s, "probe buffers (hit rows, masks, trace requests)")) return rc;
    char *p = reinterpret_cast<char *>(h->d_probe);
    pb.count = reinterpret_cast<int32_t *>(p); p += cnt_bytes;
    pb.hit = reinterpret_cast<float4 *>(p); p += (size_t) rays * sizeof(float4);
    pb.mask = reinterpret_cast<uint32_t *>(p); p += ((size_t) slots * 4 + 255) & ~(size_t) 255;
    pb.req = reinterpret_cast<float4 *>(p);
    HIP_TRY(hipMemsetAsync(pb.count, 0, cnt_bytes, s));
    return 0;
}

// kernel variant of the scene: bit 0 = environment map present, bit 1 = a rough conductor may be present
const VariantOps *variant_of(const psdr_scene_s *h) {
    int fl = (h->desc.env_emitter >= 0 ? kSceneEnv : 0) | (h->has_rough ? kSceneRough : 0) | (h->n_blas > 0 ? kSceneForest : 0);
    if (tiny_tables_ok(h)) fl |= kSceneTiny;
    switch (fl) {
        case 0: return variant_ops_0();
        case 1: return variant_ops_1();
        case 2: return variant_ops_2();
        case 3: return variant_ops_3();
        case 4: return variant_ops_4();
        case 8: return variant_ops_8();
        case 10: return variant_ops_10();
        default: return variant_ops_6();      // 6; psdr_bvh_build never builds a two-level tree under an environment map
    }
}
}  // namespace psdr_host
using namespace psdr_host;

extern "C" {

const char *psdr_last_error(void) { return g_err.c_str(); }
const char *psdr_version(void) { return "psdr-hip 0.1 gfx950"; }
int psdr_abi_struct_sizes(int32_t out[4]) {
    out[0] = (int32_t) sizeof(psdr_scene_desc); out[1] = (int32_t) sizeof(psdr_render_opts);
    out[2] = (int32_t) sizeof(psdr_tangents); out[3] = (int32_t) sizeof(psdr_grads);
    return 0;
}

int psdr_scene_create(psdr_scene_t *out) {
    if (!out) return fail("psdr_scene_create: null output");
    psdr_scene_s *h = new psdr_scene_s();
    hipError_t e = hipMalloc(&h->d_counters, sizeof(unsigned long long) * kRayCounters * kRayCounterStride);
    if (e != hipSuccess) { delete h; return fail(std::string("hipMalloc: ") + hipGetErrorString(e)); }
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) {
        h->num_cus = prop.multiProcessorCount;
        h->lds_limit = (int) std::min<size_t>(prop.sharedMemPerBlock, 160 * 1024);
    }
    *out = h;
    return 0;
}

// Developer options (A/B switches of tools and tests; DESIGN.md names the default of each).  Takes effect at the next psdr_bvh_build / render call.
int psdr_scene_set_option(psdr_scene_t h, const char *name, double value) {
    if (!h || !name) return fail("psdr_scene_set_option: null argument");
    const std::string n(name);
    const int iv = (int) value;
    if (n == "bvh_refit") h->refit_enabled = iv != 0;                   // 0: rebuild the tree on the host at every psdr_bvh_build
    else if (n == "forest_min_inline") { h->forest_min_inline = iv; h->have_bvh = false; }   // a two-level tree needs at least this many inline triangles (default 0 since round 6; 6 = only rooms: walls around objects)
    else if (n == "scratch_plain") g_scratch_plain = iv;                 // experiment: scratch blocks of >= 64 MB from hipMalloc instead of the stream-ordered pool (process-wide)
    else if (n == "own_pixels") h->opt.own_pixels = iv;                  // 0: the camera kernels always add to the image with atomics
    else if (n == "emitter_pretest") h->opt.emitter_pretest = iv;         // 0: BSDF-sampled rays that only matter on an emitter are traced like the others (A/B, tests)
    else if (n == "occ_rows") { h->opt.occ_rows = iv; h->have_bvh = false; }       // 0: the light rays of a scene without a tree test every row (A/B, tests)
    else if (n == "aa_prims") { h->aa_enabled = iv != 0; h->have_bvh = false; }   // 0: every kernel-argument primitive in plane form (no slab rows)
    else if (n == "tiny_scene") h->tiny_enabled = iv != 0;              // 0: walk a tree even for <= 16 triangles
    else if (n == "two_level") h->two_level_enabled = iv != 0;          // 0: one tree over all triangles
    else if (n == "wf_binned") h->wf_binned = iv != 0;                  // 0: wavefront streams not binned by cost class (only without the trace kernel)
    else if (n == "wf_traced") h->traced_enabled = iv != 0;             // 0: no dense trace kernel between the wavefront stages
    else if (n == "bvh_build") h->bvh_device_mode = iv;                 // 1: always on the device, 0: always on the host, -1: by size
    else if (n == "sort_edges") h->sort_edges = iv != 0;                // 0: primary-edge slots in natural order
    else if (n == "blocks_per_cu") h->opt.blocks_per_cu = iv;
    else if (n == "camera_blocks") h->opt.camera_blocks = iv;
    else if (n == "tiny_variants") h->opt.tiny_variants = iv;
    else if (n == "wide") h->opt.wide = iv;
    else if (n == "lds_budget") h->opt.lds_budget = iv;
    else if (n == "sink_rep") h->opt.sink_rep = std::max(1, std::min(16, iv));
    else if (n == "sink_private") h->opt.sink_private = iv;
    else if (n == "rev_split") h->opt.rev_split = iv;
    else if (n == "tangent_live") h->opt.tangent_live = iv;              // 0: forward-mode kernels load the tangents of every triangle row (no liveness mask)
    else if (n == "wf_geo") h->opt.wf_geo = iv;                          // 0: geometry tangents of the PathTracer always through the fused kernel
    else if (n == "logd") h->opt.logd = iv;                              // 0: PathTracer forward mode never runs the log-derivative kernel
    else if (n == "keep_records") h->opt.keep_records = iv;              // 0: psdr_render_c ignores PSDR_FLAG_KEEP_RECORDS
    else if (n == "rev_sorted") h->opt.rev_sorted = iv;                  // 0: the reverse camera kernels scatter every row adjoint on the spot (no deferred, sorted adds)
    else if (n == "sedge_split") h->opt.sedge_split = iv;
    else if (n == "probe") h->opt.probe = iv;
    else if (n == "trace_wg2") h->opt.trace_wg2 = iv;                    // dense trace kernel as two workgroups per CU: -1 by forest and launch size, 0 never, n > 0 always (stack columns of n entries)
    else if (n == "chunk_log2") h->opt.chunk_log2 = std::max(0, std::min(30, iv));
    else if (n == "bvh_maxleaf") h->opt.bvh_maxleaf = std::max(1, std::min(8, iv));
    else if (n == "bvh_tcost") h->opt.bvh_tcost = (float) value;
    else return fail("psdr_scene_set_option: unknown option '" + n + "'");
    h->kept.valid = false;                  // records kept under other options are not reused
    if (n == "tiny_scene" || n == "two_level" || n == "bvh_build" || n == "wide" || n == "bvh_maxleaf" || n == "bvh_tcost" || n == "tiny_variants") h->have_bvh = false;
    return 0;
}

int psdr_scene_destroy(psdr_scene_t h) {
    if (!h) return 0;
    if (h->d_nodes) (void) hipFree(h->d_nodes);
    if (h->d_btris) (void) hipFree(h->d_btris);
    if (h->d_counters) (void) hipFree(h->d_counters);
    if (h->d_refit_area) (void) hipFree(h->d_refit_area);
    if (h->d_sort) (void) hipFree(h->d_sort);
    if (h->d_ws) (void) hipFree(h->d_ws);
    if (h->d_live) (void) hipFree(h->d_live);
    if (h->d_logd_bad) (void) hipFree(h->d_logd_bad);
    if (h->d_hot_map) (void) hipFree(h->d_hot_map);
    if (h->d_hot_tris) (void) hipFree(h->d_hot_tris);
    if (h->d_occ) (void) hipFree(h->d_occ);
    if (h->d_top) (void) hipFree(h->d_top);
    if (h->d_inline_ids) (void) hipFree(h->d_inline_ids);
    if (h->d_lbvh) (void) hipFree(h->d_lbvh);
    if (h->d_nodes4) (void) hipFree(h->d_nodes4);
    if (h->d_topo4) (void) hipFree(h->d_topo4);
    if (h->d_rev) (void) hipFree(h->d_rev);
    if (h->d_se_list) (void) hipFree(h->d_se_list);
    if (h->d_rev_deep) (void) hipFree(h->d_rev_deep);
    if (h->d_pe_rep) (void) hipFree(h->d_pe_rep);
    if (h->d_trace_ovf) (void) hipFree(h->d_trace_ovf);
    if (h->d_probe) (void) hipFree(h->d_probe);
    delete h;
    return 0;
}

int psdr_scene_set_tables(psdr_scene_t h, const psdr_scene_desc *desc) {
    if (!h || !desc) return fail("psdr_scene_set_tables: null argument");
    // validate a local copy; the handle keeps its previous tables when the call fails
    psdr_scene_desc d = *desc;
    if (d.num_tris <= 0 || !d.tri_info || !d.tri_mesh) return fail("Missing meshes!");
    if (!d.cam) return fail("Missing sensor!");
    if (reinterpret_cast<uintptr_t>(d.tri_info) & 15) return fail("psdr_scene_set_tables: tri_info must be 16-byte aligned");
    if (d.width <= 0 || d.height <= 0) return fail("Invalid film resolution");
    if (d.num_meshes <= 0 || !d.mesh_bsdf || !d.mesh_emitter) return fail("psdr_scene_set_tables: mesh_bsdf / mesh_emitter missing");
    if (d.num_bsdfs > 0 && (!d.bsdf_rec || !d.texels)) return fail("psdr_scene_set_tables: bsdf_rec / texels missing");
    if (d.num_emitters > 0 && (!d.emitter_f || !d.emitter_i)) return fail("psdr_scene_set_tables: emitter tables missing");
    if (d.num_emitters > 1 && (!d.emitter_cmf || !d.emitter_pmf)) return fail("psdr_scene_set_tables: emitter distribution missing");
    if (d.num_sec_edges > 0 && (!d.sec_edge || !d.sec_cmf || !d.sec_pmf)) return fail("psdr_scene_set_tables: secondary-edge tables missing");
    if (d.num_prim_edges > 0 && (!d.prim_edge || !d.prim_cmf || !d.prim_pmf)) return fail("psdr_scene_set_tables: primary-edge tables missing");
    if (d.num_prim_edges <= 0) d.prim_edge_z = nullptr;
    if (d.num_guide_cells > 0 && (!d.guide_cmf || !d.guide_pmf)) return fail("psdr_scene_set_tables: guiding tables missing");
    // no environment map unless its record is given (a zero-initialised desc means "none")
    if (!d.env_f) d.env_emitter = -1;
    if (d.env_emitter >= 0) {
        if (d.env_emitter >= d.num_emitters || !d.env_cmf || !d.env_pmf || d.env_reso[0] <= 0 || d.env_reso[1] <= 0 || d.env_tex[1] < 2 ||
            d.env_tex[2] < 2)
            return fail("psdr_scene_set_tables: inconsistent environment-map record");
    }
    if (h->have_tables && (d.tri_info != h->desc.tri_info || d.num_tris != h->desc.num_tris)) h->have_bvh = false;
    // psdr_bvh_build keeps a host copy of emitter_i (hot gradient rows, the LDS table block of tiny scenes): new emitter tables need it again
    if (h->have_tables && (d.emitter_i != h->desc.emitter_i || d.num_emitters != h->desc.num_emitters)) h->have_bvh = false;
    if (!h->have_tables || std::memcmp(&h->desc, &d, sizeof(d)) != 0) { ++h->tables_gen; h->kept.valid = false; }
    h->desc = d;
    // material_mask = 0: unknown -> serve every BSDF type
    h->has_rough = d.material_mask == 0 || (d.material_mask & (1u << PSDR_BSDF_ROUGHCONDUCTOR)) != 0;
    h->have_tables = true;
    return 0;
}

// A copy between host and device buffers ON THE CALLER'S STREAM that is complete when the call returns (the host build reads the data /
// the source vectors go out of scope) -- not hipMemcpy: the null stream would drag every other stream of the process into the wait.
static hipError_t copy_on_stream(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t s) {
    if (hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, s)) return e;
    return hipStreamSynchronize(s);
}
// SceneView::emit_rows from the handle's current primitive table and its host copy of the emitter table
static void update_emit_rows(psdr_scene_s *h) {
    std::vector<char> is_em((size_t) std::max(h->desc.num_tris, 0), 0);
    for (int e = 0; e < h->desc.num_emitters && (size_t) (e + 1) * PSDR_EMITTER_I_STRIDE <= h->emitter_i.size(); ++e) {
        const int32_t *ei = h->emitter_i.data() + (size_t) e * PSDR_EMITTER_I_STRIDE;
        if (e == h->desc.env_emitter) { h->emit_rows = 0; return; }
        for (int f = 0; f < ei[2]; ++f) if (ei[1] + f >= 0 && ei[1] + f < h->desc.num_tris) is_em[(size_t) (ei[1] + f)] = 1;
    }
    h->emit_rows = h->n_tiny > 0 ? tiny_emitter_rows(h->tiny_meta, h->n_tiny, h->aa_cnt, is_em) : 0u;
}
int psdr_bvh_build(psdr_scene_t h, void *stream) {
    if (!h || !h->have_tables) return fail("Scene not loaded yet!");
    hipStream_t s = (hipStream_t) stream;
    const int T = h->desc.num_tris;
    h->kept.valid = false;                  // records of a value sweep belong to the tree they were traced with
    // ---- refit: same triangle count as the tree on the device and the tree has not degraded
    const bool tiny = h->tiny_enabled && T <= kTinyTris;        // the triangles travel in the kernel arguments: host copy needed
    // The host copy of emitter_i (LDS table sizes, two-level eligibility, hot gradient rows) is refreshed by the build paths below only: a
    // refit must not run on when the emitter layout changed under an unchanged triangle count -- the kernels would stage too few entries of
    // face_cmf / face_pmf.  One small device-to-host copy per refit candidate; a changed layout takes the full build.
    bool emitters_same = (size_t) std::max(h->desc.num_emitters, 0) * PSDR_EMITTER_I_STRIDE == h->emitter_i.size();
    if (emitters_same && !h->emitter_i.empty() && h->refit_enabled && !tiny && h->refit_ok && h->tree_tris == T && h->num_nodes > 0) {
        std::vector<int32_t> now(h->emitter_i.size());
        emitters_same = h->desc.emitter_i != nullptr;
        if (emitters_same) {
            HIP_TRY(copy_on_stream(now.data(), h->desc.emitter_i, now.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
            emitters_same = std::memcmp(now.data(), h->emitter_i.data(), now.size() * sizeof(int32_t)) == 0;
        }
    }
    const bool forest_stands = !(forest_tables(h) && !small_tables_fit(h));      // a two-level tree whose kernels could no longer stage the tables: rebuild as one tree
    if (h->refit_enabled && !tiny && h->refit_ok && emitters_same && forest_stands && h->tree_tris == T && h->num_nodes > 0 && h->refits_since_build < kMaxRefits) {
        float prev_area = h->built_area;
        if (h->refits_since_build > 0) {        // of the PREVIOUS refit (done long ago), read on the stream that wrote it
            HIP_TRY(hipMemcpyAsync(&prev_area, h->d_refit_area, sizeof(float), hipMemcpyDeviceToHost, h->refit_stream));
            HIP_TRY(hipStreamSynchronize(h->refit_stream));
        }
        if (prev_area <= kRefitAreaGrowth * h->built_area && h->lbvh) {
            h->refit_stream = s;
            if (int rc = lbvh_refit(h, s)) return rc;
            if (int rc = bvh4_refill(h, s)) return rc;
            h->refits_since_build++;
            h->num_refits++;
            h->have_bvh = true;
            return 0;
        }
        if (prev_area <= kRefitAreaGrowth * h->built_area) {
            bool refit_fits = true;
            h->refit_stream = s;
            HIP_TRY(hipMemsetAsync(h->d_refit_area, 0, sizeof(float), s));
            hipLaunchKernelGGL(k_refit_leaves, dim3((h->num_btris + kBlock - 1) / kBlock), dim3(kBlock), 0, s, h->d_btris, h->desc.tri_info, h->num_btris);
            for (int l = (int) h->level_start.size() - 2; l >= 0; --l) {
                const int first = h->level_start[l], count = h->level_start[l + 1] - first;
                hipLaunchKernelGGL(k_refit_level, dim3((count + kBlock - 1) / kBlock), dim3(kBlock), 0, s, h->d_nodes, h->d_btris, first, count, h->bvh_pad,
                                   h->d_refit_area);
            }
            HIP_TRY(hipGetLastError());
            if (int rc = bvh4_refill(h, s)) return rc;          // the 4-wide nodes from the refitted boxes
            if (h->n_blas > 0) {          // the kernel arguments carry the inline primitives and the tree boxes: fetch the refitted ones
                float4 top[kMaxInlineTris * 3 + 2 * kMaxBlas];
                hipLaunchKernelGGL(k_gather_top, dim3(1), dim3(64), 0, s, h->d_top, h->d_nodes, h->desc.tri_info, h->d_inline_ids, h->n_inline, h->n_blas);
                HIP_TRY(hipMemcpyAsync(top, h->d_top, sizeof(top), hipMemcpyDeviceToHost, s));
                HIP_TRY(hipStreamSynchronize(s));
                std::vector<float4> tris(top, top + 3 * (size_t) h->n_inline), prims;
                pack_tiny_prims(tris, prims);
                // moved wall vertices may no longer pair into parallelograms: more than kTinyTris primitives -> full rebuild below
                // (which then picks a single tree)
                refit_fits = (int) prims.size() / 3 <= kTinyTris;
                if (refit_fits) {
                    h->n_tiny = tiny_plane_form(prims, h->tiny, h->tiny_meta, &h->aa_cnt, h->aa_enabled);
                    update_emit_rows(h);
                    for (int k = 0; k < h->n_blas; ++k) {
                        const float w = h->blas_lo[k].w;
                        h->blas_lo[k] = top[kMaxInlineTris * 3 + k]; h->blas_lo[k].w = w;
                        h->blas_hi[k] = top[kMaxInlineTris * 3 + kMaxBlas + k];
                    }
                }
            }
            if (refit_fits) {
                h->refits_since_build++;
                h->num_refits++;
                h->have_bvh = true;
                return 0;
            }
        }
    }
    // ---- build on the device: large tables (the host build is a D2H of the table + a single-threaded SAH build), or on request
    h->lbvh = false;
    if (!tiny && (h->bvh_device_mode == 1 || (h->bvh_device_mode < 0 && T >= kLbvhAutoTris))) {
        bool fallback = false;
        if (int rc = lbvh_build(h, s, fallback)) return rc;
        if (!fallback) return 0;
    }
    std::vector<float> rows((size_t) T * PSDR_TRI_STRIDE);
    HIP_TRY(hipMemcpyAsync(rows.data(), h->desc.tri_info, rows.size() * sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    h->emitter_i.assign((size_t) std::max(h->desc.num_emitters, 0) * PSDR_EMITTER_I_STRIDE, 0);
    if (h->desc.num_emitters > 0 && h->desc.emitter_i)
        HIP_TRY(copy_on_stream(h->emitter_i.data(), h->desc.emitter_i, h->emitter_i.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    // ---- which tree: two-level (a few small meshes + a few large ones) or one tree over everything
    std::vector<int32_t> tri_mesh;
    bool forest = false;
    if (h->two_level_enabled && h->tiny_enabled && T > kTinyTris && h->desc.num_meshes > 0 && h->desc.env_emitter < 0) {
        tri_mesh.resize((size_t) T);
        HIP_TRY(copy_on_stream(tri_mesh.data(), h->desc.tri_mesh, tri_mesh.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        forest = ForestBuilder::eligible(tri_mesh.data(), T, h->desc.num_meshes) && (!PSDR_FOREST_LDS_TABLES || small_tables_fit(h));
    }
    Builder b;
    ForestBuilder fb;
    b.kMaxLeaf = fb.max_leaf = h->opt.bvh_maxleaf; b.kTraversalCost = fb.traversal_cost = h->opt.bvh_tcost;
    std::vector<float4> top_prims;
    int32_t root = 0;
    if (forest) {
        if (const char *err = fb.run(rows.data(), tri_mesh.data(), T, h->desc.num_meshes)) return fail(err);
        pack_tiny_prims(fb.inline_tris, top_prims);
        // too many inline primitives: one tree.  Too few: nothing room-like to keep out of the trees (two bunnies and a light quad:
        // PathTracer(3) 2.9 ms on one tree against 3.1 on the forest) -- the two-level tree is for walls around objects
        if ((int) top_prims.size() / 3 > kTinyTris || (int) fb.inline_tris.size() / 3 < h->forest_min_inline) { forest = false; fb = ForestBuilder(); }
    }
    if (!forest) { if (const char *err = b.run(rows.data(), T, root)) return fail(err); }
    std::vector<BvhNode> &nodes = forest ? fb.nodes : b.nodes;
    std::vector<float4> &btris = forest ? fb.btris : b.btris;
    if (nodes.size() > h->cap_nodes) {
        if (h->d_nodes) (void) hipFree(h->d_nodes);
        h->cap_nodes = std::max<size_t>(nodes.size(), 16);
        HIP_TRY(hipMalloc(&h->d_nodes, h->cap_nodes * sizeof(BvhNode)));
    }
    if (btris.size() > h->cap_btris) {
        if (h->d_btris) (void) hipFree(h->d_btris);
        h->cap_btris = btris.size();
        HIP_TRY(hipMalloc(&h->d_btris, h->cap_btris * sizeof(float4)));
    }
    if (!nodes.empty()) HIP_TRY(hipMemcpyAsync(h->d_nodes, nodes.data(), nodes.size() * sizeof(BvhNode), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(h->d_btris, btris.data(), btris.size() * sizeof(float4), hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));     // host vectors die at return
    {   // hot rows of the reverse-mode gradient cache: emitter triangles first, then by decreasing area
        constexpr int kMaxHotRows = 200;
        std::vector<int32_t> map((size_t) T, -1), tris;
        auto add = [&](int t) { if (t >= 0 && t < T && map[t] < 0 && (int) tris.size() < kMaxHotRows) { map[t] = (int32_t) tris.size(); tris.push_back(t); } };
        // a scene whose triangles travel in the kernel arguments caches EVERY row, slot = triangle: its kernels (flag kSceneTiny) address the
        // row without the map, the range test and the global-atomic arm (DeviceSink::add_tri)
        h->hot_identity = tiny;
        if (tiny) for (int t = 0; t < T; ++t) add(t);
        for (int e = 0; e < h->desc.num_emitters; ++e) {
            const int32_t *ei = h->emitter_i.data() + (size_t) e * PSDR_EMITTER_I_STRIDE;
            for (int f = 0; f < ei[2] && f < 64; ++f) add(ei[1] + f);
        }
        std::vector<int> by_area((size_t) T);
        for (int i = 0; i < T; ++i) by_area[i] = i;
        const int top = std::min(T, kMaxHotRows);
        std::partial_sort(by_area.begin(), by_area.begin() + top, by_area.end(),
                          [&](int a, int c) { return rows[(size_t) a * PSDR_TRI_STRIDE + 21] > rows[(size_t) c * PSDR_TRI_STRIDE + 21]; });
        for (int i = 0; i < top; ++i) add(by_area[i]);
        if ((size_t) T > h->hot_cap) {
            if (h->d_hot_map) (void) hipFree(h->d_hot_map);
            if (h->d_hot_tris) (void) hipFree(h->d_hot_tris);
            h->hot_cap = (size_t) T;
            HIP_TRY(hipMalloc(&h->d_hot_map, h->hot_cap * sizeof(int32_t)));
            HIP_TRY(hipMalloc(&h->d_hot_tris, kMaxHotRows * sizeof(int32_t)));
        }
        HIP_TRY(copy_on_stream(h->d_hot_map, map.data(), map.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
        HIP_TRY(copy_on_stream(h->d_hot_tris, tris.data(), tris.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
        h->hot_rows = (int) tris.size();
    }
    h->root = root;
    h->bvh_depth = forest ? fb.max_depth : b.max_depth; h->num_nodes = (int) nodes.size(); h->num_btris = (int) btris.size() / 3;
    if (int rc = bvh4_build(h, nodes, forest ? fb.roots : std::vector<int32_t>{root}, forest, s)) return rc;
    h->have_bvh = true;
    h->n_tiny = 0; h->aa_cnt = 0; h->n_blas = 0; h->n_inline = 0; h->have_occ = false; h->emit_rows = 0;
    if (forest) {
        h->n_tiny = tiny_plane_form(top_prims, h->tiny, h->tiny_meta, &h->aa_cnt, h->aa_enabled);
        update_emit_rows(h);
        h->n_inline = (int) fb.inline_ids.size();
        h->n_blas = (int) fb.roots.size();
        for (int k = 0; k < h->n_blas; ++k) {
            float lo[3], hi[3];
            fb.tree_box(k, lo, hi);
            h->blas_lo[k] = float4{lo[0], lo[1], lo[2], 0.f}; h->blas_hi[k] = float4{hi[0], hi[1], hi[2], 0.f};
            std::memcpy(&h->blas_lo[k].w, &fb.roots[(size_t) k], 4);
        }
        if (!h->d_top) HIP_TRY(hipMalloc(&h->d_top, sizeof(float4) * (kMaxInlineTris * 3 + 2 * kMaxBlas)));
        if (!h->d_inline_ids) HIP_TRY(hipMalloc(&h->d_inline_ids, sizeof(int32_t) * kMaxInlineTris));
        if (h->n_inline) HIP_TRY(copy_on_stream(h->d_inline_ids, fb.inline_ids.data(), sizeof(int32_t) * (size_t) h->n_inline, hipMemcpyHostToDevice, s));
    } else if (tiny) {
        std::vector<float4> prims;
        std::vector<int> row_of_prim;
        pack_tiny_prims(b.btris, prims);           // walls as parallelograms: half the tests
        h->n_tiny = tiny_plane_form(prims, h->tiny, h->tiny_meta, &h->aa_cnt, h->aa_enabled, &row_of_prim);
        update_emit_rows(h);
        // occluder rows of the light rays (SceneView::occ): which primitives can lie between a triangle and an emitter triangle
        h->have_occ = false;
        if (T <= 32 && h->n_tiny <= 32) {
            std::vector<char> is_em((size_t) T, 0);
            for (int e = 0; e < h->desc.num_emitters; ++e) {
                const int32_t *ei = h->emitter_i.data() + (size_t) e * PSDR_EMITTER_I_STRIDE;
                for (int f = 0; f < ei[2]; ++f) if (ei[1] + f >= 0 && ei[1] + f < T) is_em[(size_t) (ei[1] + f)] = 1;
            }
            std::vector<uint32_t> occ;
            if (h->opt.occ_rows != 0) tiny_occluder_rows(prims, row_of_prim, T, is_em, occ);
            else occ.assign((size_t) T * T, 0xffffffffu);
            if (occ.size() > h->occ_cap) {
                if (h->d_occ) (void) hipFree(h->d_occ);
                h->occ_cap = std::max<size_t>(occ.size(), 32 * 32);
                HIP_TRY(hipMalloc(&h->d_occ, h->occ_cap * sizeof(uint32_t)));
            }
            HIP_TRY(copy_on_stream(h->d_occ, occ.data(), occ.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s));
            h->have_occ = true;
            h->occ_max_rows = 0;                                     // the most rows any (triangle, emitter triangle) entry names (psdr_scene_info)
            for (int r = 0; r < T; ++r) for (int e = 0; e < T; ++e) if (is_em[(size_t) e]) h->occ_max_rows = std::max(h->occ_max_rows, (int) __builtin_popcount(occ[(size_t) r * T + e] & ((1u << h->n_tiny) - 1u)));
        }
    }
    // what the refit path needs: the levels of the breadth-first node order, the padding, the reference area
    h->tree_tris = T; h->refits_since_build = 0; h->num_builds++; h->bvh_pad = forest ? fb.pad : b.pad;
    h->level_start.clear();
    double area = 0.0;
    auto node_area = [](const BvhNode &n) {
        const float dx = std::max(n.hi0[0], n.hi1[0]) - std::min(n.lo0[0], n.lo1[0]), dy = std::max(n.hi0[1], n.hi1[1]) - std::min(n.lo0[1], n.lo1[1]),
                    dz = std::max(n.hi0[2], n.hi1[2]) - std::min(n.lo0[2], n.lo1[2]);
        return (double) (dx * dy + dy * dz + dz * dx);
    };
    if (forest) {
        h->level_start = fb.level_start;
        for (const BvhNode &n : nodes) area += node_area(n);
    } else if (root >= 0) {
        std::vector<int> depth(b.nodes.size(), 0);
        for (size_t i = 0; i < b.nodes.size(); ++i) {
            const BvhNode &n = b.nodes[i];
            if (n.c0 >= 0) depth[n.c0] = depth[i] + 1;
            if (n.c1 >= 0) depth[n.c1] = depth[i] + 1;
            if (i == 0 || depth[i] != depth[i - 1]) h->level_start.push_back((int) i);
            area += node_area(n);
        }
        h->level_start.push_back((int) b.nodes.size());
    }
    h->refit_ok = true;
    if (forest) for (int32_t r : fb.roots) if (r < 0) h->refit_ok = false;     // k_gather_top reads the roots as nodes 0 .. n_blas-1
    h->built_area = (float) area;
    if (!h->d_refit_area) HIP_TRY(hipMalloc(&h->d_refit_area, sizeof(float)));
    return 0;
}

int psdr_trace(psdr_scene_t h, int32_t m, const float *ox, const float *oy, const float *oz, const float *dx, const float *dy,
               const float *dz, const float *tmax, int32_t *out_shape, int32_t *out_tri, float *out_u, float *out_v, void *stream) {
    if (!h || !h->have_tables) return fail("Scene not loaded yet!");
    if (!h->have_bvh) return fail("Input scene must be configured!");
    if (m <= 0) return 0;
    LaunchCtx cx{};
    cx.sc.d = h->desc; cx.sc.nodes = h->d_nodes; cx.sc.btris = h->d_btris; cx.sc.root = h->root;
    fill_top(h, cx.sc);
    plan_lds(h, cx);
    hipLaunchKernelGGL(k_trace, dim3(launch_blocks(h, m)), dim3(kBlock), lds_bytes(cx, h), (hipStream_t) stream, cx, m, ox, oy, oz, dx, dy, dz, tmax,
                       out_shape, out_tri, out_u, out_v);
    HIP_TRY(hipGetLastError());
    return 0;
}

int psdr_render_c(psdr_scene_t h, const psdr_render_opts *o, float *out_img, void *stream) {
    if (!h || !o || !out_img) return fail("psdr_render_c: null argument");
    if (!h->have_tables) return fail("Scene not loaded yet!");
    if (int rc = check_counts(h, o)) return rc;
    hipStream_t s = (hipStream_t) stream;
    if (int rc = begin_call(h, s)) return rc;
    if (o->integrator == PSDR_INTEGRATOR_PATH) h->last_path_depth = o->max_depth;
    const long long WH = (long long) h->desc.width * h->desc.height;
    HIP_TRY(hipMemsetAsync(out_img, 0, sizeof(float) * WH * 3, s));
    return variant_of(h)->render_c(h, o, out_img, s);
}

int psdr_render_d_fwd(psdr_scene_t h, const psdr_render_opts *o, int32_t K, const psdr_tangents *tangents, float *out_img,
                      float *out_dimg, void *stream) {
    if (!h || !o || !out_img || !out_dimg || !tangents) return fail("psdr_render_d_fwd: null argument");
    if (!h->have_tables) return fail("Scene not loaded yet!");
    if (int rc = check_counts(h, o)) return rc;
    hipStream_t s = (hipStream_t) stream;
    if (int rc = begin_call(h, s)) return rc;
    if (K != 1 && K != 3) return fail("psdr_render_d_fwd: K must be 1 or 3");
    return variant_of(h)->render_fwd(h, o, K, tangents, out_img, out_dimg, s);
}

int psdr_render_d_rev(psdr_scene_t h, const psdr_render_opts *o, const float *adj_img, float *out_img, const psdr_grads *grads,
                      void *stream) {
    if (!h || !o || !adj_img || !grads) return fail("psdr_render_d_rev: null argument");
    if (!h->have_tables) return fail("Scene not loaded yet!");
    if (int rc = check_counts(h, o)) return rc;
    if (o->flags & PSDR_FLAG_LITERAL_FORMS) return fail("psdr_render_d_rev: PSDR_FLAG_LITERAL_FORMS is a forward-mode diagnostic (no literal-form adjoint)");
    if (o->integrator == PSDR_INTEGRATOR_PATH && o->max_depth > kMaxRevDepthDeep)
        return fail("psdr_render_d_rev: PathTracer max_depth > 250 is not supported in reverse mode");
    hipStream_t s = (hipStream_t) stream;
    if (int rc = begin_call(h, s)) return rc;
    return variant_of(h)->render_rev(h, o, adj_img, out_img, grads, s);
}

int psdr_guide_build(psdr_scene_t h, const psdr_render_opts *o, const int32_t reso[4], int32_t nrounds, float *out_mass, void *stream) {
    if (!h || !o || !reso || !out_mass) return fail("psdr_guide_build: null argument");
    if (nrounds <= 0) return fail("psdr_guide_build: nrounds must be positive");
    if (h->desc.num_sec_edges <= 0) return fail("psdr_guide_build: scene has no secondary edges");
    hipStream_t s = (hipStream_t) stream;
    if (int rc = begin_call(h, s)) return rc;
    LaunchCtx cx;
    if (int rc = make_ctx(h, o, 2, cx)) return rc;
    cx.sc.d.guide_cmf = nullptr; cx.sc.d.num_guide_cells = 0;
    const long long cells = (long long) reso[0] * reso[1] * reso[2];
    const long long n = cells * reso[3];
    if (n <= 0 || n > 0x7fffffffLL) return fail("psdr_guide_build: invalid resolution");
    HIP_TRY(hipMemsetAsync(out_mass, 0, sizeof(float) * cells, s));
    return variant_of(h)->guide(h, cx, reso, nrounds, n, out_mass, s);
}

int psdr_bvh_stats(psdr_scene_t h, int32_t out[4]) {
    if (!h || !out) return fail("psdr_bvh_stats: null argument");
    out[0] = h->num_builds; out[1] = h->num_refits; out[2] = h->num_nodes; out[3] = h->bvh_depth;
    return 0;
}

int psdr_scene_info(psdr_scene_t h, int32_t out[8]) {
    if (!h || !out) return fail("psdr_scene_info: null argument");
    const int n_slab = (h->aa_cnt & 255) + ((h->aa_cnt >> 8) & 255) + (h->aa_cnt >> 16);
    out[0] = h->n_tiny - (h->aa_cnt != 0 ? kAaSlots - n_slab : 0); out[1] = h->n_blas; out[2] = h->n_inline; out[3] = h->num_btris; out[4] = h->lbvh ? 1 : 0;
    out[5] = n_slab; out[6] = h->have_occ ? 1 : 0; out[7] = h->have_occ ? h->occ_max_rows : 0;
    return 0;
}

int psdr_get_counters(psdr_scene_t h, uint64_t out[4]) {
    if (!h || !out) return fail("psdr_get_counters: null argument");
    unsigned long long all[kRayCounters * kRayCounterStride], c[1] = {0};
    // ordered behind the launches of the last render call on ITS stream (a blocking null-stream copy is not
    // ordered against a non-blocking caller stream)
    HIP_TRY(hipMemcpyAsync(all, h->d_counters, sizeof(all), hipMemcpyDeviceToHost, h->last_stream));
    HIP_TRY(hipStreamSynchronize(h->last_stream));
    for (int i = 0; i < kRayCounters; ++i) c[0] += all[i * kRayCounterStride];
#ifdef PSDR_STAGE_CLOCKS
    {   // developer build: the phase clocks of the traced bounce stage (psdr_kernels.h PSDR_CLK_MARK), summed over the waves of the last call
        unsigned long long ph[12] = {0};
        for (int i = 0; i < kRayCounters; ++i) for (int k = 0; k < 12; ++k) ph[k] += all[i * kRayCounterStride + 1 + k];
        std::fprintf(stderr, "stage clocks:");
        for (int k = 0; k < 12; ++k) std::fprintf(stderr, " %llu", ph[k]);
        std::fprintf(stderr, "\n");
    }
#endif
    out[0] = c[0]; out[1] = h->slots[0]; out[2] = h->slots[1]; out[3] = h->slots[2];
    return 0;
}

}  // extern "C"

