// psdr_bvh_build.h -- host-side BVH builder shared by the HIP library and the host test harness.
#pragma once
#include "psdr_device.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace psdr {

// ---------------------------------------------------------------------------- BVH builder
// Host binned-SAH builder (replaces the OptiX GAS build, include/psdr/scene/optix.h:277-340).
// Scenes of this path are small (12 .. ~50k triangles) and the tree is rebuilt on every
// Scene::configure() as the reference does (scene.cpp:247-248).
struct BuildTri { float lo[3], hi[3], c[3]; };
struct Builder {
    const float *rows; int T;
    std::vector<BuildTri> tris;
    std::vector<int> order;
    std::vector<BvhNode> nodes;
    std::vector<float4> btris;
    int max_depth = 0;
    float pad = 0.f;
    int kMaxLeaf = 4;                  // leaf encoding holds 1..8
    float kTraversalCost = 1.0f;       // node visit / triangle test (both ~40-50 VALU ops)
    Builder() {                        // experiment knobs (tools only): PSDR_BVH_MAXLEAF, PSDR_BVH_TCOST
        if (const char *e = std::getenv("PSDR_BVH_MAXLEAF")) kMaxLeaf = std::max(1, std::min(8, std::atoi(e)));
        if (const char *e = std::getenv("PSDR_BVH_TCOST")) kTraversalCost = (float) std::atof(e);
    }

    static float area(const float *lo, const float *hi) {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
    void bounds(int first, int count, float *lo, float *hi, float *clo, float *chi) const {
        for (int k = 0; k < 3; ++k) { lo[k] = clo[k] = INFINITY; hi[k] = chi[k] = -INFINITY; }
        for (int i = first; i < first + count; ++i) {
            const BuildTri &t = tris[order[i]];
            for (int k = 0; k < 3; ++k) {
                lo[k] = std::min(lo[k], t.lo[k]); hi[k] = std::max(hi[k], t.hi[k]);
                clo[k] = std::min(clo[k], t.c[k]); chi[k] = std::max(chi[k], t.c[k]);
            }
        }
    }
    int32_t make_leaf(int first, int count) {
        const int base = (int) btris.size() / 3;
        for (int i = first; i < first + count; ++i) {
            const int id = order[i];
            const float *r = rows + (size_t) id * PSDR_TRI_STRIDE;
            float4 a{r[0], r[1], r[2], 0.f};
            std::memcpy(&a.w, &id, 4);
            btris.push_back(a);
            btris.push_back(float4{r[3], r[4], r[5], 0.f});
            btris.push_back(float4{r[6], r[7], r[8], 0.f});
        }
        return ~((base << 3) | (count - 1));
    }
    // returns encoded child; writes the padded box of the subtree into lo/hi
    int32_t build(int first, int count, int depth, float *lo, float *hi) {
        float clo[3], chi[3];
        bounds(first, count, lo, hi, clo, chi);
        max_depth = std::max(max_depth, depth);
        auto leaf_here = [&]() {
            const int32_t leaf = make_leaf(first, count);
            for (int k = 0; k < 3; ++k) { lo[k] -= pad; hi[k] += pad; }
            return leaf;
        };
        if (count == 1) return leaf_here();
        int mid = -1;
        int ax = 0;
        for (int k = 1; k < 3; ++k) if (chi[k] - clo[k] > chi[ax] - clo[ax]) ax = k;
        if (depth < 20 && chi[ax] > clo[ax]) {
            constexpr int NB = 16;
            float best = INFINITY; int best_ax = -1, best_b = -1;
            for (int a = 0; a < 3; ++a) {
                if (!(chi[a] > clo[a])) continue;
                int cnt[NB] = {0}; float blo[NB][3], bhi[NB][3];
                for (int b = 0; b < NB; ++b) for (int k = 0; k < 3; ++k) { blo[b][k] = INFINITY; bhi[b][k] = -INFINITY; }
                const float scale = NB / (chi[a] - clo[a]);
                for (int i = first; i < first + count; ++i) {
                    const BuildTri &t = tris[order[i]];
                    int b = std::min(NB - 1, std::max(0, (int) ((t.c[a] - clo[a]) * scale)));
                    cnt[b]++;
                    for (int k = 0; k < 3; ++k) { blo[b][k] = std::min(blo[b][k], t.lo[k]); bhi[b][k] = std::max(bhi[b][k], t.hi[k]); }
                }
                float ra[NB]; int rc[NB];
                float lo_[3] = {INFINITY, INFINITY, INFINITY}, hi_[3] = {-INFINITY, -INFINITY, -INFINITY}; int c = 0;
                for (int b = NB - 1; b > 0; --b) {
                    for (int k = 0; k < 3; ++k) { lo_[k] = std::min(lo_[k], blo[b][k]); hi_[k] = std::max(hi_[k], bhi[b][k]); }
                    c += cnt[b]; rc[b] = c; ra[b] = c ? area(lo_, hi_) : 0.f;
                }
                for (int k = 0; k < 3; ++k) { lo_[k] = INFINITY; hi_[k] = -INFINITY; }
                c = 0;
                for (int b = 0; b < NB - 1; ++b) {
                    for (int k = 0; k < 3; ++k) { lo_[k] = std::min(lo_[k], blo[b][k]); hi_[k] = std::max(hi_[k], bhi[b][k]); }
                    c += cnt[b];
                    if (c == 0 || rc[b + 1] == 0) continue;
                    const float cost = area(lo_, hi_) * c + ra[b + 1] * rc[b + 1];
                    if (cost < best) { best = cost; best_ax = a; best_b = b; }
                }
            }
            // SAH termination for small groups: a leaf of n triangles costs n tests; splitting costs one node
            // visit (two slab tests, an LDS stack push) plus the area-weighted tests of the children.  Large flat
            // primitives (walls) end up one quad per leaf instead of sharing a room-sized box with their neighbours.
            if (count <= kMaxLeaf) {
                const float parent = area(lo, hi);
                const float split_cost = best_ax >= 0 && parent > 0.f ? kTraversalCost + best / parent : INFINITY;
                if (!(split_cost < (float) count)) return leaf_here();
            }
            if (best_ax >= 0) {
                const float scale = NB / (chi[best_ax] - clo[best_ax]);
                auto it = std::partition(order.begin() + first, order.begin() + first + count, [&](int id) {
                    int b = std::min(NB - 1, std::max(0, (int) ((tris[id].c[best_ax] - clo[best_ax]) * scale)));
                    return b <= best_b;
                });
                mid = (int) (it - order.begin());
                if (mid == first || mid == first + count) mid = -1;
            }
        }
        if (mid < 0 && count <= kMaxLeaf) return leaf_here();
        if (mid < 0) {   // median split keeps the depth bounded
            mid = first + count / 2;
            std::nth_element(order.begin() + first, order.begin() + mid, order.begin() + first + count,
                             [&](int a, int b) { return tris[a].c[ax] < tris[b].c[ax]; });
        }
        const int idx = (int) nodes.size();
        nodes.push_back(BvhNode{});
        BvhNode n{};
        n.c0 = build(first, mid - first, depth + 1, n.lo0, n.hi0);
        n.c1 = build(mid, first + count - mid, depth + 1, n.lo1, n.hi1);
        nodes[idx] = n;
        return idx;
    }
    // Builds the tree over `T` TriangleInfo rows; returns nullptr on success or an error text.
    const char *run(const float *rows_, int T_, int32_t &root) {
        rows = rows_; T = T_;
        tris.resize(T); order.resize(T);
        float slo[3] = {INFINITY, INFINITY, INFINITY}, shi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int i = 0; i < T; ++i) {
            const float *r = rows + (size_t) i * PSDR_TRI_STRIDE;
            BuildTri &t = tris[i];
            for (int k = 0; k < 3; ++k) {
                const float p = r[k], q = r[k] + r[3 + k], w = r[k] + r[6 + k];
                if (!std::isfinite(p) || !std::isfinite(q) || !std::isfinite(w)) return "psdr_bvh_build: non-finite vertex";
                t.lo[k] = std::min(p, std::min(q, w)); t.hi[k] = std::max(p, std::max(q, w));
                t.c[k] = 0.5f * (t.lo[k] + t.hi[k]);
                slo[k] = std::min(slo[k], t.lo[k]); shi[k] = std::max(shi[k], t.hi[k]);
            }
            order[i] = i;
        }
        const float ext = std::max(shi[0] - slo[0], std::max(shi[1] - slo[1], shi[2] - slo[2]));
        pad = std::max(1e-6f, 1e-5f * ext);       // keeps flat (axis-aligned) triangles inside a non-degenerate slab
        nodes.clear(); btris.clear();
        nodes.reserve(T); btris.reserve((size_t) T * 3);
        float lo[3], hi[3];
        root = build(0, T, 0, lo, hi);
        if (max_depth > kBvhStack - 2) return "psdr_bvh_build: tree too deep for the traversal stack";
        // breadth-first relabelling: a prefix of the node array is the top of the tree (the part the
        // kernels stage in LDS)
        if (root >= 0) {
            std::vector<int> order_bfs; order_bfs.reserve(nodes.size());
            std::vector<int> newid(nodes.size(), -1);
            order_bfs.push_back(root);
            for (size_t i = 0; i < order_bfs.size(); ++i) {
                const BvhNode &n = nodes[order_bfs[i]];
                if (n.c0 >= 0) order_bfs.push_back(n.c0);
                if (n.c1 >= 0) order_bfs.push_back(n.c1);
            }
            for (size_t i = 0; i < order_bfs.size(); ++i) newid[order_bfs[i]] = (int) i;
            std::vector<BvhNode> out(order_bfs.size());
            for (size_t i = 0; i < order_bfs.size(); ++i) {
                BvhNode n = nodes[order_bfs[i]];
                if (n.c0 >= 0) n.c0 = newid[n.c0];
                if (n.c1 >= 0) n.c1 = newid[n.c1];
                out[i] = n;
            }
            nodes.swap(out);
            root = 0;
        }
        return nullptr;
    }
};


}  // namespace psdr
