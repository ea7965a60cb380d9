// simd_sim.cpp -- developer tool (host only): replays the product's BVH walk (csrc/psdr_device.h closest_hit,
// same tree, same while-while order) ray by ray, records per outer iteration how many inner-node steps and leaf
// triangle tests each ray makes, and prices wave64 execution models on those records:
//   A  today's kernel: 64 consecutive rays per wave, every outer iteration costs the wave the MAX over its lanes
//   B  a workgroup-local ray queue (R rays per lane): a lane that finishes its ray takes the next one at the
//      following outer-iteration boundary
// Output: useful lane-work / (64 x wave cost) = SIMD efficiency of the walk.
#include "../../psdr-cuda_amd/csrc/psdr_bvh_build.h"
#include <cstdio>
#include <vector>
using namespace psdr;

struct Ev { uint16_t inner, leaf; };
struct RayRec { std::vector<Ev> ev; };

static void walk2(const SceneView &sc, const Vec3f &o, const Vec3f &d, RayRec &rec, int max_inner_batch, Hit &best);
static void walk(const SceneView &sc, const Vec3f &o, const Vec3f &d, RayRec &rec, int max_inner_batch) {
    Hit best; best.tri = -1; best.u = best.v = -1.f; best.t = INFINITY;
    walk2(sc, o, d, rec, max_inner_batch, best);
}
static void walk2(const SceneView &sc, const Vec3f &o, const Vec3f &d, RayRec &rec, int max_inner_batch, Hit &best) {
    const Vec3f inv{1.f / d.x, 1.f / d.y, 1.f / d.z};
    int sp = 0; int32_t stack[64];
    int32_t cur = sc.root;
    constexpr int32_t kDone = 0x7fffffff;
    while (cur != kDone) {
        Ev e{0, 0};
        while (cur >= 0 && cur != kDone) {
            const BvhNode &n = sc.nodes[cur];
            float t0, t1;
            const bool h0 = slab(n.lo0, n.hi0, o, inv, best.t, t0), h1 = slab(n.lo1, n.hi1, o, inv, best.t, t1);
            e.inner++;
            if (h0 && h1) { const bool f0 = t0 <= t1; stack[sp++] = f0 ? n.c1 : n.c0; cur = f0 ? n.c0 : n.c1; }
            else if (h0 || h1) cur = h0 ? n.c0 : n.c1;
            else cur = sp > 0 ? stack[--sp] : kDone;
            if (max_inner_batch > 0 && e.inner >= max_inner_batch) break;
        }
        if (cur != kDone && cur < 0) {
            const int enc = ~cur, first = enc >> 3, cnt = (enc & 7) + 1;
            for (int i = 0; i < cnt; ++i) leaf_triangle_test(sc.btris[(first + i) * 3], sc.btris[(first + i) * 3 + 1], sc.btris[(first + i) * 3 + 2], o, d, best);
            e.leaf = (uint16_t) cnt;
            cur = sp > 0 ? stack[--sp] : kDone;
        }
        rec.ev.push_back(e);
    }
}

// model E: two-level tree.  Meshes with >= min_blas triangles get their own tree (BLAS); the triangles of the
// other meshes and the BLAS boxes are tested by every lane in a uniform loop (phase 1, no divergence); the rays
// whose segment [0, t_best] enters a BLAS box are compacted across a batch of `bsz` rays (workgroup, LDS) and
// walk that tree in full waves (phase 2, one round per deferred BLAS, near to far).
extern "C" int simd_sim_two_level(const psdr_scene_desc *d, int m, const float *o, const float *dir, double cn, double ct, double cfix, int bsz,
                                  int min_blas, double c_tri1, double c_box1, double c_fix1, double c_item, double *out) {
    const int T = d->num_tris;
    std::vector<int> mesh(T);
    int nm = 0;
    for (int i = 0; i < T; ++i) { mesh[i] = d->tri_mesh[i] & ~PSDR_TRI_FACE_NORMALS; nm = std::max(nm, mesh[i] + 1); }
    std::vector<std::vector<int>> ids(nm);
    for (int i = 0; i < T; ++i) ids[mesh[i]].push_back(i);
    struct Blas { Builder b; SceneView sc; std::vector<float> rows; float lo[3], hi[3]; };
    std::vector<Blas *> blas; std::vector<int> inl;
    for (int k = 0; k < nm; ++k) {
        if ((int) ids[k].size() < min_blas) { inl.insert(inl.end(), ids[k].begin(), ids[k].end()); continue; }
        Blas *B = new Blas(); B->rows.resize(ids[k].size() * PSDR_TRI_STRIDE);
        for (size_t i = 0; i < ids[k].size(); ++i) std::memcpy(&B->rows[i * PSDR_TRI_STRIDE], d->tri_info + (size_t) ids[k][i] * PSDR_TRI_STRIDE, PSDR_TRI_STRIDE * 4);
        int32_t root = 0; if (B->b.run(B->rows.data(), (int) ids[k].size(), root)) return 1;
        B->sc = SceneView{}; B->sc.nodes = B->b.nodes.data(); B->sc.btris = B->b.btris.data(); B->sc.root = root;
        for (int a = 0; a < 3; ++a) { B->lo[a] = INFINITY; B->hi[a] = -INFINITY; }
        for (size_t i = 0; i < ids[k].size(); ++i) { const float *r = &B->rows[i * PSDR_TRI_STRIDE]; for (int a = 0; a < 3; ++a) {
            const float p = r[a], q = r[a] + r[3 + a], w = r[a] + r[6 + a];
            B->lo[a] = std::min(B->lo[a], std::min(p, std::min(q, w))); B->hi[a] = std::max(B->hi[a], std::max(p, std::max(q, w))); } }
        blas.push_back(B);
    }
    const double c1 = inl.size() * c_tri1 + blas.size() * c_box1 + c_fix1;
    struct Def { int b; float t; };
    std::vector<std::vector<Def>> defer(m);
    std::vector<Hit> best(m);
    double n_def = 0;
    for (int i = 0; i < m; ++i) {
        const Vec3f O{o[3 * i], o[3 * i + 1], o[3 * i + 2]}, D{dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]};
        Hit &h = best[i]; h.tri = -1; h.u = h.v = -1.f; h.t = INFINITY;
        for (int t : inl) { const float *r = d->tri_info + (size_t) t * PSDR_TRI_STRIDE;
            float4 a{r[0], r[1], r[2], 0.f}; std::memcpy(&a.w, &t, 4);
            leaf_triangle_test(a, float4{r[3], r[4], r[5], 0.f}, float4{r[6], r[7], r[8], 0.f}, O, D, h); }
        const Vec3f inv{1.f / D.x, 1.f / D.y, 1.f / D.z};
        for (size_t k = 0; k < blas.size(); ++k) { float te; if (slab(blas[k]->lo, blas[k]->hi, O, inv, h.t, te)) defer[i].push_back(Def{(int) k, te}); }
        std::sort(defer[i].begin(), defer[i].end(), [](const Def &a, const Def &b) { return a.t < b.t; });
        n_def += defer[i].size();
    }
    double cost = 0; double items = 0, waves2 = 0;
    for (int g = 0; g + bsz <= m; g += bsz) {
        cost += c1 * (bsz / 64);
        for (int round = 0;; ++round) {
            std::vector<RayRec> recs;
            for (int i = g; i < g + bsz; ++i) {
                if ((int) defer[i].size() <= round) continue;
                const Def &df = defer[i][round];
                if (!(df.t <= best[i].t)) continue;
                RayRec rr;
                Hit h = best[i];
                const int before = h.tri;
                walk2(blas[df.b]->sc, Vec3f{o[3 * i], o[3 * i + 1], o[3 * i + 2]}, Vec3f{dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]}, rr, 0, h);
                (void) before; best[i] = h;
                recs.push_back(std::move(rr));
            }
            bool more = false;
            for (int i = g; i < g + bsz; ++i) if ((int) defer[i].size() > round + 1) more = true;
            items += recs.size();
            for (size_t w = 0; w < recs.size(); w += 64) {
                waves2++;
                size_t it = 0; bool any = true;
                cost += c_item;
                while (any) {
                    any = false; int mi = 0, ml = 0;
                    for (size_t l = w; l < std::min(recs.size(), w + 64); ++l) if (it < recs[l].ev.size()) { any = true; mi = std::max<int>(mi, recs[l].ev[it].inner); ml = std::max<int>(ml, recs[l].ev[it].leaf); }
                    if (any) cost += mi * cn + ml * ct + cfix;
                    ++it;
                }
            }
            if (!more) break;
        }
    }
    const int mb = m / bsz * bsz;
    out[0] = cost / (mb / 64); out[1] = c1; out[2] = items / mb; out[3] = waves2 / (mb / bsz); out[4] = (double) inl.size(); out[5] = (double) blas.size();
    out[6] = n_def / m;
    return 0;
}

extern "C" int simd_sim(const psdr_scene_desc *d, int m, const float *o, const float *dir, double cn, double ct, double cfix, int R, int max_inner_batch,
                        double *out) {
    Builder b; int32_t root = 0;
    if (b.run(d->tri_info, d->num_tris, root)) return 1;
    SceneView sc{}; sc.d = *d; sc.nodes = b.nodes.data(); sc.btris = b.btris.data(); sc.root = root;
    std::vector<RayRec> rays(m);
    double useful = 0, steps = 0, leaves = 0, iters = 0;
    for (int i = 0; i < m; ++i) {
        walk(sc, Vec3f{o[3 * i], o[3 * i + 1], o[3 * i + 2]}, Vec3f{dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]}, rays[i], max_inner_batch);
        for (const Ev &e : rays[i].ev) { useful += e.inner * cn + e.leaf * ct; steps += e.inner; leaves += e.leaf; }
        iters += rays[i].ev.size();
    }
    // model A
    double costA = 0;
    for (int w = 0; w + 64 <= m; w += 64) {
        size_t it = 0; bool any = true;
        while (any) {
            any = false; int mi = 0, ml = 0;
            for (int l = 0; l < 64; ++l) if (it < rays[w + l].ev.size()) { any = true; mi = std::max<int>(mi, rays[w + l].ev[it].inner); ml = std::max<int>(ml, rays[w + l].ev[it].leaf); }
            if (any) costA += mi * cn + ml * ct + cfix;
            ++it;
        }
    }
    // model B: workgroups of 256 lanes (4 waves), batch = 256 * R consecutive rays, shared queue.  Waves are advanced
    // round-robin by accumulated cost (the wave that is behind takes the next iteration).
    double costB = 0;
    const int batch = 256 * R;
    for (int g = 0; g + batch <= m; g += batch) {
        int next = g;
        int ray[4][64]; size_t pos[4][64]; double tw[4] = {0, 0, 0, 0}; bool alive[4] = {true, true, true, true};
        for (int w = 0; w < 4; ++w) for (int l = 0; l < 64; ++l) { ray[w][l] = -1; pos[w][l] = 0; }
        for (;;) {
            int w = -1;
            for (int k = 0; k < 4; ++k) if (alive[k] && (w < 0 || tw[k] < tw[w])) w = k;
            if (w < 0) break;
            // refill
            for (int l = 0; l < 64; ++l) if (ray[w][l] < 0 && next < g + batch) { ray[w][l] = next++; pos[w][l] = 0; }
            int mi = 0, ml = 0; bool any = false;
            for (int l = 0; l < 64; ++l) if (ray[w][l] >= 0) {
                const RayRec &r = rays[ray[w][l]];
                if (pos[w][l] < r.ev.size()) { any = true; mi = std::max<int>(mi, r.ev[pos[w][l]].inner); ml = std::max<int>(ml, r.ev[pos[w][l]].leaf); pos[w][l]++; }
                if (pos[w][l] >= r.ev.size()) ray[w][l] = -1;
            }
            if (!any) { alive[w] = false; continue; }
            tw[w] += mi * cn + ml * ct + cfix;
        }
        for (int k = 0; k < 4; ++k) costB += tw[k];
    }
    // model C/D: "vote" scheduling -- the wave runs ONE block per iteration, a node step (for all lanes at an inner
    // node) or one triangle test (for all lanes holding a leaf triangle), whichever has more lanes waiting; with
    // R > 0 finished lanes take the next ray of the workgroup's queue (checked every block)
    auto vote = [&](int Rq, double thresh) {
        double cost = 0;
        std::vector<std::vector<uint8_t>> ops(m);
        for (int i = 0; i < m; ++i) for (const Ev &e : rays[i].ev) { ops[i].insert(ops[i].end(), e.inner, 0); ops[i].insert(ops[i].end(), e.leaf, 1); }
        const int bt = Rq > 0 ? 256 * Rq : 64;
        for (int g = 0; g + bt <= m; g += bt) {
            const int nw = Rq > 0 ? 4 : 1;
            int next = g;
            int ray[4][64]; size_t pos[4][64]; double tw[4] = {0, 0, 0, 0}; bool alive[4] = {true, true, true, true};
            for (int w = 0; w < nw; ++w) for (int l = 0; l < 64; ++l) { ray[w][l] = -1; pos[w][l] = 0; }
            for (;;) {
                int w = -1;
                for (int k = 0; k < nw; ++k) if (alive[k] && (w < 0 || tw[k] < tw[w])) w = k;
                if (w < 0) break;
                for (int l = 0; l < 64; ++l) {
                    while (ray[w][l] < 0 && next < g + bt) { ray[w][l] = next++; pos[w][l] = 0; if (ops[ray[w][l]].empty()) ray[w][l] = -1; }
                }
                int nI = 0, nL = 0;
                for (int l = 0; l < 64; ++l) if (ray[w][l] >= 0) { if (ops[ray[w][l]][pos[w][l]] == 0) nI++; else nL++; }
                if (nI + nL == 0) { alive[w] = false; continue; }
                const int kind = (nI >= thresh * nL) ? 0 : 1;
                if ((kind == 0 && nI == 0) || (kind == 1 && nL == 0)) { /* cannot happen with thresh > 0 */ }
                for (int l = 0; l < 64; ++l) if (ray[w][l] >= 0 && ops[ray[w][l]][pos[w][l]] == kind) {
                    if (++pos[w][l] >= ops[ray[w][l]].size()) ray[w][l] = -1;
                }
                tw[w] += (kind == 0 ? cn : ct) + cfix;
            }
            for (int k = 0; k < nw; ++k) cost += tw[k];
        }
        const int mb = m / bt * bt;
        return mb ? cost / (mb / 64) : 0.0;
    };
    out[9] = vote(0, 1.0); out[10] = vote(R, 1.0); out[11] = vote(0, 0.5); out[12] = vote(R, 0.5);
    if (d->num_guide_cells == -12345) { FILE *f = fopen("/tmp/raycost.bin", "wb"); for (int i = 0; i < m; ++i) { float c = 0; for (const Ev &e : rays[i].ev) c += e.inner * cn + e.leaf * ct; fwrite(&c, 4, 1, f); } fclose(f); }
    out[0] = useful / m; out[1] = steps / m; out[2] = leaves / m; out[3] = iters / m;
    out[4] = useful / (64.0 * costA) * (m / 64 * 64) / m;       // efficiency A
    out[5] = costA / (m / 64);                                    // wave cost per 64 rays, A
    const int mb = m / batch * batch;
    out[6] = mb ? costB / (mb / 64) : 0;                          // wave cost per 64 rays, B
    out[7] = (double) b.nodes.size(); out[8] = b.max_depth;
    return 0;
}
