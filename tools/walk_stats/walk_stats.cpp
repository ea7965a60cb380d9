// walk_stats.cpp -- developer tool (host only): statistics of the 4-wide walk on the product's own forest (psdr_bvh_build.h ForestBuilder +
// collapse_bvh4, boxes quantised as csrc/psdr_hip.hip k_bvh4_fill does): node / leaf visits per ray, how deep the traversal stack gets, and
// which share of the node visits falls into the first N nodes of the level-ordered array (= what an LDS stage of N nodes would serve).
#include "../../psdr-cuda_amd/csrc/psdr_bvh_build.h"
#include <cstdio>
#include <vector>
using namespace psdr;

static void fill4(const std::vector<BvhNode> &nodes, const Bvh4Topology &tp, std::vector<Bvh4Node> &out) {
    out.resize((size_t) tp.n4);
    for (int i = 0; i < tp.n4; ++i) {
        float lo[4][3], hi[4][3], org[3] = {INFINITY, INFINITY, INFINITY}, top[3] = {-INFINITY, -INFINITY, -INFINITY};
        Bvh4Node n{};
        for (int c = 0; c < 4; ++c) {
            n.child[c] = tp.child[(size_t) i * 4 + c];
            const int32_t s = tp.src[(size_t) i * 4 + c];
            if (s < 0) continue;
            const BvhNode &b = nodes[(size_t) (s >> 1)];
            for (int a = 0; a < 3; ++a) {
                lo[c][a] = (s & 1) ? b.lo1[a] : b.lo0[a]; hi[c][a] = (s & 1) ? b.hi1[a] : b.hi0[a];
                org[a] = std::min(org[a], lo[c][a]); top[a] = std::max(top[a], hi[c][a]);
            }
        }
        for (int a = 0; a < 3; ++a) {
            n.org[a] = org[a];
            int e = 0; (void) std::frexp((top[a] - org[a]) * (1.f / 255.f), &e);
            int E = std::min(std::max(e + 127, 1), 254);
            for (;;) {
                union { int i; float f; } sc; sc.i = E << 23;
                const float scale = sc.f, inv = 1.f / scale;
                uint32_t ql = 0, qh = 0; bool ok = true;
                for (int c = 0; c < 4; ++c) {
                    if (tp.src[(size_t) i * 4 + c] < 0) { ql |= 0xffu << (8 * c); continue; }
                    int l = (int) std::floor((lo[c][a] - org[a]) * inv), u = (int) std::ceil((hi[c][a] - org[a]) * inv);
                    l = std::max(std::min(l, 255), 0);
                    while (l > 0 && std::fma((float) l, scale, org[a]) > lo[c][a]) --l;
                    while (u <= 255 && std::fma((float) u, scale, org[a]) < hi[c][a]) ++u;
                    if (u > 255) { ok = false; break; }
                    ql |= (uint32_t) l << (8 * c); qh |= (uint32_t) std::max(u, 0) << (8 * c);
                }
                if (ok || E >= 254) { n.qlo[a] = ql; n.qhi[a] = qh; break; }
                ++E;
            }
            n.exps |= (uint32_t) E << (8 * a);
        }
        out[(size_t) i] = n;
    }
}

struct Stat { int nodes = 0, leaves = 0, tris = 0, max_sp = 0; };
static void walk4(const std::vector<Bvh4Node> &n4, const std::vector<float4> &btris, int32_t root, const Vec3f &o, const Vec3f &d, Hit &best, Stat &st,
                  std::vector<long long> &visits, bool sorted) {
    const Vec3f inv{1.f / d.x, 1.f / d.y, 1.f / d.z};
    int sp = 0; int32_t stack[128]; int32_t cur = root;
    constexpr int32_t kDone = 0x7fffffff;
    auto f = [](int i) { union { int i; float f; } c; c.i = i; return c.f; };
    auto fi = [](float x) { union { int i; float f; } c; c.f = x; return (uint32_t) c.i; };
    while (cur != kDone) {
        while (cur >= 0 && cur != kDone) {
            const Bvh4Node &n = n4[(size_t) cur];
            visits[(size_t) cur]++; st.nodes++;
            const float ax = f((int) ((n.exps & 0xffu) << 23)) * inv.x, ay = f((int) (((n.exps >> 8) & 0xffu) << 23)) * inv.y, az = f((int) (((n.exps >> 16) & 0xffu) << 23)) * inv.z;
            const float bx = (n.org[0] - o.x) * inv.x, by = (n.org[1] - o.y) * inv.y, bz = (n.org[2] - o.z) * inv.z;
            const bool px = inv.x >= 0.f, py = inv.y >= 0.f, pz = inv.z >= 0.f;
            const uint32_t nx = px ? n.qlo[0] : n.qhi[0], fx = px ? n.qhi[0] : n.qlo[0], ny = py ? n.qlo[1] : n.qhi[1], fy = py ? n.qhi[1] : n.qlo[1],
                           nz = pz ? n.qlo[2] : n.qhi[2], fz = pz ? n.qhi[2] : n.qlo[2];
            uint32_t key[4]; int32_t ch[4];
            for (int c = 0; c < 4; ++c) {
                const float tnx = (float) ((nx >> (8 * c)) & 0xffu) * ax + bx, tfx = (float) ((fx >> (8 * c)) & 0xffu) * ax + bx;
                const float tny = (float) ((ny >> (8 * c)) & 0xffu) * ay + by, tfy = (float) ((fy >> (8 * c)) & 0xffu) * ay + by;
                const float tnz = (float) ((nz >> (8 * c)) & 0xffu) * az + bz, tfz = (float) ((fz >> (8 * c)) & 0xffu) * az + bz;
                const float tn = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, 0.f)), tf = fminf(fminf(tfx, tfy), fminf(tfz, best.t));
                const bool hit = tn <= tf && n.child[c] != kNoChild;
                key[c] = hit ? fi(tn) : 0xffffffffu; ch[c] = n.child[c];
            }
            if (sorted) {
                auto cx = [&](int i, int j) { if (key[j] < key[i]) { std::swap(key[i], key[j]); std::swap(ch[i], ch[j]); } };
                cx(0, 1); cx(2, 3); cx(0, 2); cx(1, 3); cx(1, 2);
            } else {          // nearest first, the rest in slot order
                int m = 0; for (int c = 1; c < 4; ++c) if (key[c] < key[m]) m = c;
                std::swap(key[0], key[m]); std::swap(ch[0], ch[m]);
            }
            if (key[3] != 0xffffffffu) stack[sp++] = ch[3];
            if (key[2] != 0xffffffffu) stack[sp++] = ch[2];
            if (key[1] != 0xffffffffu) stack[sp++] = ch[1];
            st.max_sp = std::max(st.max_sp, sp);
            cur = key[0] != 0xffffffffu ? ch[0] : (sp > 0 ? stack[--sp] : kDone);
        }
        if (cur == kDone) break;
        const int enc = ~cur, first = enc >> 3, cnt = (enc & 7) + 1;
        for (int i = 0; i < cnt; ++i) leaf_triangle_test(btris[(size_t) (first + i) * 3], btris[(size_t) (first + i) * 3 + 1], btris[(size_t) (first + i) * 3 + 2], o, d, best);
        st.leaves++; st.tris += cnt;
        cur = sp > 0 ? stack[--sp] : kDone;
    }
}

// out: [0] BVH2 nodes, [1] 4-wide nodes, [2] worst-case stack need, [3] leaf triangles, [4] rays that entered a tree, [5] node visits / such ray, [6] leaf visits,
// [7] triangles tested, [8..8+32) histogram of the deepest stack per ray, [40..48) share of node visits within the first 128, 256, 512, 1024, 2048, 4096, 8192, all nodes
extern "C" int walk_stats(const psdr_scene_desc *d, int m, const float *o, const float *dir, int sorted, double *out) {
    ForestBuilder fb;
    if (fb.run(d->tri_info, d->tri_mesh, d->num_tris, d->num_meshes)) return 1;
    Bvh4Topology tp; collapse_bvh4(fb.nodes, fb.roots, tp);
    std::vector<Bvh4Node> n4; fill4(fb.nodes, tp, n4);
    const int nb = (int) fb.roots.size();
    std::vector<float> lo((size_t) nb * 3), hi((size_t) nb * 3);
    for (int k = 0; k < nb; ++k) fb.tree_box(k, &lo[(size_t) k * 3], &hi[(size_t) k * 3]);
    std::vector<long long> visits((size_t) tp.n4, 0);
    double rays = 0, nodes = 0, leaves = 0, tris = 0; std::vector<double> hist(32, 0.0);
    for (int i = 0; i < m; ++i) {
        const Vec3f O{o[3 * i], o[3 * i + 1], o[3 * i + 2]}, D{dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]};
        Hit best; best.tri = -1; best.u = best.v = -1.f; best.t = INFINITY;
        for (size_t t = 0; t < fb.inline_tris.size(); t += 3) leaf_triangle_test(fb.inline_tris[t], fb.inline_tris[t + 1], fb.inline_tris[t + 2], O, D, best);
        const Vec3f inv{1.f / D.x, 1.f / D.y, 1.f / D.z};
        Stat st; bool any = false;
        // nearest box first, as closest_hit does
        uint32_t cand = (1u << nb) - 1u;
        while (cand) {
            int pick = -1; float near_t = INFINITY;
            for (int k = 0; k < nb; ++k) {
                if (!((cand >> k) & 1u)) continue;
                float te;
                if (!slab(&lo[(size_t) k * 3], &hi[(size_t) k * 3], O, inv, best.t, te)) cand &= ~(1u << k);
                else if (te < near_t) { near_t = te; pick = k; }
            }
            if (pick < 0) break;
            cand &= ~(1u << pick);
            any = true;
            walk4(n4, fb.btris, tp.roots[(size_t) pick], O, D, best, st, visits, sorted != 0);
        }
        if (any) { rays++; nodes += st.nodes; leaves += st.leaves; tris += st.tris; hist[(size_t) std::min(st.max_sp, 31)]++; }
    }
    out[0] = (double) fb.nodes.size(); out[1] = tp.n4; out[2] = tp.stack_need; out[3] = (double) fb.btris.size() / 3; out[4] = rays;
    out[5] = nodes / std::max(rays, 1.0); out[6] = leaves / std::max(rays, 1.0); out[7] = tris / std::max(rays, 1.0);
    for (int i = 0; i < 32; ++i) out[8 + i] = hist[(size_t) i] / std::max(rays, 1.0);
    const int cuts[8] = {128, 256, 512, 1024, 2048, 4096, 8192, 1 << 30};
    for (int c = 0; c < 8; ++c) { double s = 0; for (int i = 0; i < tp.n4 && i < cuts[c]; ++i) s += (double) visits[(size_t) i]; out[40 + c] = s / std::max(nodes, 1.0); }
    return 0;
}
