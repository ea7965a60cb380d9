// hostcheck.cpp -- TEST-ONLY harness: runs the PRODUCT's estimator code (csrc/psdr_device.h, PSDR_HD
// functions) on the host, sample by sample, so that `-m "not gpu"` tests can compare it with the
// oracle where no GPU exists.  It is never imported by the psdr_cuda package and is not a fallback:
// the render path (libpsdr_hip.so) only ever executes these functions inside HIP kernels.
#include "../../psdr-cuda_amd/csrc/psdr_bvh_build.h"
#include "../../psdr-cuda_amd/csrc/psdr_reverse.h"

#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

using namespace psdr;

namespace {
struct HostScene {
    SceneView sc{};
    Builder b;
};
bool setup(HostScene &hs, const psdr_scene_desc *d) {
    hs.sc.d = *d;
    if (!hs.sc.d.env_f) hs.sc.d.env_emitter = -1;
    int32_t root = 0;
    if (hs.b.run(d->tri_info, d->num_tris, root)) return false;
    hs.sc.nodes = hs.b.nodes.data(); hs.sc.btris = hs.b.btris.data(); hs.sc.root = root;
    // tiny scenes take the all-triangles path of closest_hit, as psdr_bvh_build arranges on the device
    const char *e = std::getenv("PSDR_TINY_SCENE");
    if (d->num_tris <= kTinyTris && !(e && std::atoi(e) == 0)) {
        std::vector<float4> prims;
        pack_tiny_prims(hs.b.btris, prims);
        hs.sc.n_tiny = tiny_plane_form(prims, hs.sc.tiny, hs.sc.tiny_meta, &hs.sc.aa_cnt);
        // SceneView::emit_rows as psdr_bvh_build sets it (the estimators' emitter pre-test; the host check's TangentView flags carry neither kSceneTiny nor
        // kSceneForest, so the estimators here do not USE it -- hostcheck_emitter_rows hands it to the tests)
        std::vector<char> is_em((size_t) d->num_tris, 0);
        for (int e = 0; e < d->num_emitters && d->emitter_i; ++e) {
            const int32_t *ei = d->emitter_i + (size_t) e * PSDR_EMITTER_I_STRIDE;
            for (int f = 0; f < ei[2]; ++f) if (ei[1] + f >= 0 && ei[1] + f < d->num_tris) is_em[(size_t) (ei[1] + f)] = 1;
        }
        hs.sc.emit_rows = tiny_emitter_rows(hs.sc.tiny_meta, hs.sc.n_tiny, hs.sc.aa_cnt, is_em);
    }
    return true;
}
template <class F> void pfor(long long n, int nt, F f) {
    std::vector<std::thread> th;
    long long chunk = (n + nt - 1) / nt;
    for (int t = 0; t < nt; ++t) {
        long long a = t * chunk, b = std::min(n, a + chunk);
        if (a >= b) break;
        th.emplace_back([=] { f(a, b, t); });
    }
    for (auto &x : th) x.join();
}
}  // namespace

extern "C" {

int hostcheck_trace(const psdr_scene_desc *d, int m, const float *o, const float *dir, int *tri, float *u, float *v) {
    HostScene hs;
    if (!setup(hs, d)) return 1;
    TraversalStack st;
    for (int i = 0; i < m; ++i) {
        Hit h = closest_hit(hs.sc, st, Vec3f{o[3 * i], o[3 * i + 1], o[3 * i + 2]}, Vec3f{dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]}, INFINITY);
        tri[i] = h.tri; u[i] = h.u; v[i] = h.v;
    }
    return 0;
}

// the same search restricted to the rows a per-ray mask names (closest_hit MASKED: what the light rays of a scene without a tree run with SceneView::occ)
int hostcheck_trace_rows(const psdr_scene_desc *d, int m, const float *o, const float *dir, const uint32_t *rows, int *tri, float *u, float *v, float *t) {
    HostScene hs;
    if (!setup(hs, d)) return 1;
    if (hs.sc.n_tiny <= 0) return 2;
    TraversalStack st;
    for (int i = 0; i < m; ++i) {
        Hit h = closest_hit<false, 2, true>(hs.sc, st, Vec3f{o[3 * i], o[3 * i + 1], o[3 * i + 2]}, Vec3f{dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]}, INFINITY, -1, -1, 0, rows[i]);
        tri[i] = h.tri; u[i] = h.u; v[i] = h.v; t[i] = h.t;
    }
    return 0;
}

int hostcheck_emitter_rows(const psdr_scene_desc *d, uint32_t *rows) {
    HostScene hs;
    if (!setup(hs, d)) return 1;
    *rows = hs.sc.emit_rows;
    return 0;
}

// rows in use and slab-form slots per axis of a tiny scene's primitive list (tiny_plane_form)
int hostcheck_tiny_layout(const psdr_scene_desc *d, int *out) {
    HostScene hs;
    if (!setup(hs, d)) return 1;
    out[0] = hs.sc.n_tiny; out[1] = hs.sc.aa_cnt & 255; out[2] = (hs.sc.aa_cnt >> 8) & 255; out[3] = hs.sc.aa_cnt >> 16;
    return 0;
}

// occluder rows of a scene without a tree as psdr_bvh_build computes them (psdr_bvh_build.h tiny_occluder_rows): occ[num_tris^2], row_of_tri[num_tris] = the
// row of `tiny` that holds the triangle; emitter triangles from the caller's emitter_i table
int hostcheck_occluder_rows(const psdr_scene_desc *d, uint32_t *occ_out, int *row_of_tri) {
    HostScene hs;
    if (!setup(hs, d)) return 1;
    if (d->num_tris > kTinyTris) return 2;
    std::vector<float4> prims;
    std::vector<int> row_of_prim;
    pack_tiny_prims(hs.b.btris, prims);
    float4 rows[kTinyRows * 4]; int32_t meta[kTinyRows * 4]; int32_t aa = 0;
    tiny_plane_form(prims, rows, meta, &aa, true, &row_of_prim);
    std::vector<char> is_em((size_t) d->num_tris, 0);
    for (int e = 0; e < d->num_emitters; ++e) {
        const int32_t *ei = d->emitter_i + (size_t) e * PSDR_EMITTER_I_STRIDE;
        for (int f = 0; f < ei[2]; ++f) if (ei[1] + f >= 0 && ei[1] + f < d->num_tris) is_em[(size_t) (ei[1] + f)] = 1;
    }
    std::vector<uint32_t> occ;
    tiny_occluder_rows(prims, row_of_prim, d->num_tris, is_em, occ);
    std::memcpy(occ_out, occ.data(), occ.size() * sizeof(uint32_t));
    for (int i = 0; i < (int) prims.size() / 3; ++i) {
        int32_t ids; std::memcpy(&ids, &prims[(size_t) i * 3].w, 4);
        row_of_tri[ids & 0xffff] = row_of_prim[(size_t) i];
        if (((uint32_t) ids >> 16) != 0xffffu) row_of_tri[(uint32_t) ids >> 16] = row_of_prim[(size_t) i];
    }
    return 0;
}

// mode 0: renderC; mode 1: renderD forward (K = 1), all three terms
int hostcheck_render(const psdr_scene_desc *d, const psdr_render_opts *o, int mode, const psdr_tangents *tan, float *img, float *dimg,
                     int nthreads) {
    HostScene hs;
    if (!setup(hs, d)) return 1;
    hs.sc.literal_forms = (o->flags & PSDR_FLAG_LITERAL_FORMS) ? 1 : 0;
    const int W = d->width, H = d->height;
    const long long WH = (long long) W * H;
    const size_t n3 = (size_t) WH * 3;
    nthreads = std::max(1, nthreads);
    std::vector<std::vector<double>> acc(nthreads, std::vector<double>(n3, 0.0)), dacc(nthreads, std::vector<double>(mode ? n3 : 0, 0.0));
    LiParams lp{o->integrator, o->bsdf_samples, o->light_samples, o->max_depth, o->hide_emitters, o->field};
    TangentView<1, kSceneAll> tv1; tv1.t[0] = tan ? *tan : psdr_tangents{};
    const TangentView<0, kSceneAll> tv0{};   // the host check always carries the env-map and rough-conductor code
    const int nsp = o->spp_end - o->spp_begin;
    if (o->spp > 0 && nsp > 0) {
        const RngJump jump = make_rng_jump(o->rng_offset[0]);
        pfor(WH * nsp, nthreads, [&](long long a, long long b, int t) {
            TraversalStack st; uint32_t nr = 0;
            for (long long j = a; j < b; ++j) {
                const int pixel = (int) (j / nsp), s = o->spp_begin + (int) (j % nsp);
                const uint64_t slot = (uint64_t) pixel * o->spp + s;
                if (mode == 0) {
                    Vec3f r = camera_sample<float, float>(hs.sc, tv0, st, lp, jump, pixel, slot, nr);
                    acc[t][pixel * 3] += r.x / o->spp; acc[t][pixel * 3 + 1] += r.y / o->spp; acc[t][pixel * 3 + 2] += r.z / o->spp;
                } else {
                    const bool geo = tv1.t[0].d_tri_info || tv1.t[0].d_cam_to_world;
                    Vec3<Dual<1>> r = geo ? camera_sample<Dual<1>, Dual<1>>(hs.sc, tv1, st, lp, jump, pixel, slot, nr)
                                          : camera_sample<float, Dual<1>>(hs.sc, tv1, st, lp, jump, pixel, slot, nr);
                    acc[t][pixel * 3] += r.x.v / o->spp; acc[t][pixel * 3 + 1] += r.y.v / o->spp; acc[t][pixel * 3 + 2] += r.z.v / o->spp;
                    dacc[t][pixel * 3] += r.x.d[0] / o->spp; dacc[t][pixel * 3 + 1] += r.y.d[0] / o->spp; dacc[t][pixel * 3 + 2] += r.z.d[0] / o->spp;
                }
            }
        });
    }
    if (mode == 1 && o->sppe > 0 && o->sppe_end > o->sppe_begin && d->num_prim_edges > 0) {
        const RngJump jump = make_rng_jump(o->rng_offset[1]);
        const long long i0 = WH * o->sppe_begin, n = WH * (o->sppe_end - o->sppe_begin);
        pfor(n, nthreads, [&](long long a, long long b, int t) {
            TraversalStack st; uint32_t nr = 0;
            for (long long j = a; j < b; ++j) {
                float tg[1][3];
                int pix = primary_edge_sample<1>(hs.sc, tv1, st, lp, jump, (uint64_t) (i0 + j), 1.f / o->sppe, tg, nr);
                if (pix >= 0) for (int c = 0; c < 3; ++c) dacc[t][pix * 3 + c] += tg[0][c];
            }
        });
    }
    if (mode == 1 && o->sppse > 0 && o->sppse_end > o->sppse_begin && d->num_sec_edges > 0 && o->integrator == PSDR_INTEGRATOR_DIRECT) {
        const RngJump jump = make_rng_jump(o->rng_offset[2]);
        const long long i0 = WH * o->sppse_begin, n = WH * (o->sppse_end - o->sppse_begin);
        const bool guided = d->guide_cmf && d->num_guide_cells > 0;
        pfor(n, nthreads, [&](long long a, long long b, int t) {
            TraversalStack st; uint32_t nr = 0;
            for (long long j = a; j < b; ++j) {
                Rng rng; rng.init((uint64_t) (i0 + j), jump);
                float s3[3] = {rng.next(), rng.next(), rng.next()};
                const float pdf0 = guided ? guide_sample_reuse(hs.sc, s3) : 1.f;
                Vec3<Dual<1>> v;
                int pix = secondary_edge_sample<Dual<1>>(hs.sc, tv1, st, s3, v, nr);
                if (pix >= 0) {
                    v = zero_nonfinite(v);
                    const float scale = (pdf0 > kEpsilon ? 1.f / pdf0 : 1.f) / o->sppse;
                    dacc[t][pix * 3] += v.x.d[0] * scale; dacc[t][pix * 3 + 1] += v.y.d[0] * scale; dacc[t][pix * 3 + 2] += v.z.d[0] * scale;
                }
            }
        });
    }
    for (size_t i = 0; i < n3; ++i) {
        double s = 0, ds = 0;
        for (int t = 0; t < nthreads; ++t) { s += acc[t][i]; if (mode) ds += dacc[t][i]; }
        img[i] = (float) s;
        if (mode && dimg) dimg[i] = (float) ds;
    }
    return 0;
}

int hostcheck_guide(const psdr_scene_desc *d, const int *reso, int nrounds, float *mass, int nthreads) {
    HostScene hs;
    if (!setup(hs, d)) return 1;
    hs.sc.d.guide_cmf = nullptr; hs.sc.d.num_guide_cells = 0;
    const long long cells = (long long) reso[0] * reso[1] * reso[2], n = cells * reso[3];
    std::vector<double> m(cells, 0.0);
    const TangentView<0, kSceneAll> tv0{};   // the host check always carries the env-map and rough-conductor code
    const RngJump nojump{1ull, 0ull};
    pfor(cells, std::max(1, nthreads), [&](long long a, long long b, int) {
        TraversalStack st; uint32_t nr = 0;
        for (long long cell = a; cell < b; ++cell) for (int q = 0; q < reso[3]; ++q) {
            const long long j = cell * reso[3] + q;
            const int c0 = (int) (cell / (reso[1] * reso[2])), rem = (int) (cell - (long long) c0 * reso[1] * reso[2]), c1 = rem / reso[2], c2 = rem - c1 * reso[2];
            Rng rng; rng.init((uint64_t) j, nojump);
            float accv = 0.f;
            for (int r = 0; r < nrounds; ++r) {
                float s3[3] = {rng.next(), rng.next(), rng.next()};
                s3[0] = (s3[0] + c0) * (1.f / reso[0]); s3[1] = (s3[1] + c1) * (1.f / reso[1]); s3[2] = (s3[2] + c2) * (1.f / reso[2]);
                Vec3f v; secondary_edge_sample<float>(hs.sc, tv0, st, s3, v, nr);
                v = zero_nonfinite(v);
                if (reso[3] > 1) v = v / (float) reso[3];
                accv += fmaxf(v.x, fmaxf(v.y, v.z));
            }
            if (nrounds > 1) accv /= (float) nrounds;
            m[cell] += accv;
        }
    });
    (void) n;
    for (long long c = 0; c < cells; ++c) mass[c] = (float) m[c];
    return 0;
}
}

namespace {
struct HostSink {
    static constexpr int flags = kSceneAll;
    static constexpr bool has_env = true;
    psdr_grads g;
    void add_env(int w, float v) const { put(g.g_env_f, w, v); }
    static void put(float *b, size_t i, float v) { if (b && v != 0.f && std::isfinite(v)) b[i] += v; }
    void add_tri(int tri, int word, float v) const { put(g.g_tri_info, (size_t) tri * PSDR_TRI_STRIDE + word, v); }
    void add_texel(int idx, float v) const { put(g.g_texels, idx, v); }
    void add_rad(int e, int c, float v) const { put(g.g_emitter_rad, (size_t) e * 3 + c, v); }
    void add_cam(int w, float v) const { put(g.g_cam_to_world, w, v); }
    void add_sedge(int e, int w, float v) const { put(g.g_sec_edge, (size_t) e * PSDR_SEDGE_STRIDE + w, v); }
    void add_pedge(int e, int w, float v) const { put(g.g_prim_edge, (size_t) e * PSDR_PEDGE_STRIDE + w, v); }
};
}  // namespace

extern "C" int hostcheck_render_rev(const psdr_scene_desc *d, const psdr_render_opts *o, const float *adj, float *img, const psdr_grads *grads) {
    HostScene hs;
    if (!setup(hs, d)) return 1;
    const int W = d->width, H = d->height;
    const long long WH = (long long) W * H;
    LiParams lp{o->integrator, o->bsdf_samples, o->light_samples, o->max_depth, o->hide_emitters, o->field};
    HostSink sink; sink.g = *grads;
    TraversalStack st; uint32_t nr = 0;
    std::vector<double> acc((size_t) WH * 3, 0.0);
    const int nsp = o->spp_end - o->spp_begin;
    if (o->spp > 0 && nsp > 0) {
        const RngJump jump = make_rng_jump(o->rng_offset[0]);
        for (long long j = 0; j < WH * nsp; ++j) {
            const int pixel = (int) (j / nsp), s = o->spp_begin + (int) (j % nsp);
            const float inv = 1.f / o->spp;
            const Vec3f a{adj[pixel * 3] * inv, adj[pixel * 3 + 1] * inv, adj[pixel * 3 + 2] * inv};
            PrimaryGrad pg; PathRec rec;
            const bool geo = grads->g_tri_info != nullptr || grads->g_cam_to_world != nullptr;
            const Vec3f r = geo ? camera_sample_reverse<true>(sink, pg, rec, hs.sc, st, lp, jump, pixel, (uint64_t) pixel * o->spp + s, a, nr)
                                : camera_sample_reverse<false>(sink, pg, rec, hs.sc, st, lp, jump, pixel, (uint64_t) pixel * o->spp + s, a, nr);
            if (pg.tri >= 0) for (int w = 0; w < kPrimaryWords; ++w) sink.add_tri(pg.tri, w, pg.w[w]);
            acc[pixel * 3] += r.x * inv; acc[pixel * 3 + 1] += r.y * inv; acc[pixel * 3 + 2] += r.z * inv;
        }
    }
    if (o->sppe > 0 && o->sppe_end > o->sppe_begin && d->num_prim_edges > 0) {
        const RngJump jump = make_rng_jump(o->rng_offset[1]);
        for (long long j = WH * o->sppe_begin; j < WH * o->sppe_end; ++j)
            primary_edge_reverse(sink, hs.sc, st, lp, jump, (uint64_t) j, 1.f / o->sppe, adj, nr);
    }
    if (o->sppse > 0 && o->sppse_end > o->sppse_begin && d->num_sec_edges > 0 && o->integrator == PSDR_INTEGRATOR_DIRECT) {
        const RngJump jump = make_rng_jump(o->rng_offset[2]);
        const bool guided = d->guide_cmf && d->num_guide_cells > 0;
        for (long long j = WH * o->sppse_begin; j < WH * o->sppse_end; ++j) {
            Rng rng; rng.init((uint64_t) j, jump);
            float s3[3] = {rng.next(), rng.next(), rng.next()};
            const float pdf0 = guided ? guide_sample_reuse(hs.sc, s3) : 1.f;
            secondary_edge_reverse(sink, hs.sc, st, s3, (pdf0 > kEpsilon ? 1.f / pdf0 : 1.f) / o->sppse, adj, nr);
        }
    }
    if (img) for (size_t i = 0; i < acc.size(); ++i) img[i] = (float) acc[i];
    return 0;
}
