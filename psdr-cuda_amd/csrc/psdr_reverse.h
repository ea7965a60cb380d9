// psdr_reverse.h -- reverse-mode (adjoint) evaluation of the interior and boundary estimators.
//
// The reference obtains parameter gradients by letting Enoki walk an N-wide tape backwards
// (enoki.backward(loss), docs/inverse_diff_render.rst): every gather<TriangleInfoD> (scene.cpp:300),
// bitmap gather (bitmap.cpp:72-75) and radiance read turns into an atomic scatter_add.  Here the
// adjoint of one sample is written out by hand and evaluated in registers: forward re-evaluation of
// the sample, then the chain rule backwards, then a scatter-add of the per-sample gradient into the
// gradient tables through a GradSink.  No tape ever touches HBM.
//
// Path structure (DirectIntegrator == one vertex with B/L loops; PathTracer == max_depth vertices):
//   L = Le_0 + T_0,   T_k = c_k + f_k * T_{k+1}
// with c_k the emitter terms gathered at vertex k and f_k the BSDF-sampled throughput.  The adjoint of
// T_k is a * beta_k (known going forward), the adjoint of f_k needs the suffix radiance T_{k+1}, so a
// first sweep records (c_k, f_k) per vertex and a second sweep replays the same random numbers and
// differentiates vertex by vertex.
//
// RoughConductor: the BSDF-internal partials (eval / pdf / sample w.r.t. wi, wo, alpha, eta, k) are
// taken with small local dual numbers (Dual<K> over the handful of BSDF inputs); everything
// geometric is hand-derived.
#pragma once
#include "psdr_device.h"

namespace psdr {

// ---------------------------------------------------------------------- adjoint helpers
PSDR_HD Vec3f operator*(float s, const Vec3f &a) { return {a.x * s, a.y * s, a.z * s}; }
PSDR_HD void acc(Vec3f &d, const Vec3f &v) { d.x += v.x; d.y += v.y; d.z += v.z; }
PSDR_HD float hsum3(const Vec3f &a) { return a.x + a.y + a.z; }

// n = normalize(v)
PSDR_HD Vec3f normalize_vjp(const Vec3f &v, const Vec3f &n, const Vec3f &an) {
    const float inv = 1.f / norm(v);
    return (an - n * dot(n, an)) * inv;
}
// Frame(n) = (s, t, n): adjoint of coordinate_system w.r.t. n
PSDR_HD Vec3f frame_vjp(const Vec3f &n, const Vec3f &as, const Vec3f &at) {
    const float sg = copysignf(1.f, n.z);
    const float a = -1.f / (sg + n.z);
    const float ab = as.y * sg + at.x;
    const float aa = as.x * n.x * n.x * sg + at.y * n.y * n.y + ab * n.x * n.y;
    Vec3f an;
    an.x = as.x * 2.f * n.x * a * sg - as.z * sg + ab * n.y * a;
    an.y = at.y * 2.f * n.y * a - at.z + ab * n.x * a;
    an.z = aa * a * a;
    return an;
}

// Adjoint accumulators of one path vertex (the quantities later terms read)
struct VertexAdj {
    Vec3f p, s, t, n, wi;     // position, shading frame, local incident direction
    float u, v;               // texture coordinates
    PSDR_HD void clear() { p = s = t = n = wi = Vec3f(0.f); u = v = 0.f; }
};

// Pending adjoint of ONE path vertex' triangle row: its position (scattered as p0 + u e1 + v e2), face normal and area.  A vertex
// receives adjoints from three consecutive iterations of the replay sweep (the BSDF sample that found it, its own frame /
// direction chain, the direction chain of the vertex after it), all at the same (u, v): they are summed here and the row is
// touched once -- 13 adds per vertex instead of 34 (an LDS float add costs ~3 cycles per active LANE on gfx950,
// tools/micro/lds_atomics2.hip: the adds, not the arithmetic, were a third of the PathTracer geometry kernel).
struct RowAdj {
    Vec3f p, fn; float area;
    PSDR_HD void clear() { p = fn = Vec3f(0.f); area = 0.f; }
};
PSDR_HD float finite_or_zero(float v) { return isfinite(v) ? v : 0.f; }
PSDR_HD void acc_finite(Vec3f &d, const Vec3f &v) { d.x += finite_or_zero(v.x); d.y += finite_or_zero(v.y); d.z += finite_or_zero(v.z); }
template <class Sink> PSDR_HD void scatter_point(Sink &sink, int tri, float u, float v, const Vec3f &ap);
template <class Sink> PSDR_HD void scatter_vec(Sink &sink, int tri, int word, const Vec3f &a);
template <class Sink> PSDR_HD void flush_row_normal_area(Sink &sink, int tri, const RowAdj &r) {
    scatter_vec(sink, tri, 18, r.fn);
    sink.add_tri(tri, 21, r.area);
}

// A sink may take a whole row adjoint at once (DeviceSink::add_row: lane-private accumulators for the emitter's rows)
template <class Sink, class = void> struct SinkTakesRows : std::false_type {};
template <class Sink> struct SinkTakesRows<Sink, std::void_t<decltype(std::declval<Sink &>().add_row(0, 0.f, 0.f, std::declval<const Vec3f &>(), std::declval<const Vec3f &>(), 0.f))>> : std::true_type {};
template <class Sink> PSDR_HD void scatter_row(Sink &sink, int tri, float u, float v, const Vec3f &ap, const Vec3f &afn, float aarea) {
    if constexpr (SinkTakesRows<Sink>::value) { if (sink.add_row(tri, u, v, ap, afn, aarea)) return; }
    scatter_point(sink, tri, u, v, ap);
    scatter_vec(sink, tri, 18, afn);
    sink.add_tri(tri, 21, aarea);
}

// A sink may DEFER a complete row adjoint to a convergent point of its kernel (DeviceSink::defer_row: the row waits in a per-lane LDS column and
// leaves sorted by row with the other lanes' rows, psdr_kernels.h sink_add_row_wave); every other sink scatters it on the spot.
template <class Sink, class = void> struct SinkDefersRows : std::false_type {};
template <class Sink> struct SinkDefersRows<Sink, std::void_t<decltype(std::declval<Sink &>().defer_row(0, 0.f, 0.f, std::declval<const RowAdj &>()))>> : std::true_type {};
template <class Sink> PSDR_HD void complete_row(Sink &sink, int tri, float u, float v, const RowAdj &r) {
    if constexpr (SinkDefersRows<Sink>::value) sink.defer_row(tri, u, v, r);
    else scatter_row(sink, tri, u, v, r.p, r.fn, r.area);
}

// Gradient sink interface (duck-typed): add_tri(tri, word, g), add_texel(idx, g), add_rad(e, c, g),
// add_cam(word, g), add_sedge(edge, word, g), add_pedge(edge, word, g).

template <class Sink> PSDR_HD void scatter_point(Sink &sink, int tri, float u, float v, const Vec3f &ap) {
    // p = p0 + u e1 + v e2 (barycentrics detached)
    sink.add_tri(tri, 0, ap.x); sink.add_tri(tri, 1, ap.y); sink.add_tri(tri, 2, ap.z);
    sink.add_tri(tri, 3, u * ap.x); sink.add_tri(tri, 4, u * ap.y); sink.add_tri(tri, 5, u * ap.z);
    sink.add_tri(tri, 6, v * ap.x); sink.add_tri(tri, 7, v * ap.y); sink.add_tri(tri, 8, v * ap.z);
}
template <class Sink> PSDR_HD void scatter_vec(Sink &sink, int tri, int word, const Vec3f &a) {
    sink.add_tri(tri, word, a.x); sink.add_tri(tri, word + 1, a.y); sink.add_tri(tri, word + 2, a.z);
}

// shading normal of a triangle at (bu, bv): forward + adjoint (returns d sh_n/d(bu,bv) contributions)
struct ShNormal { Vec3f v, n; bool face; };
PSDR_HD ShNormal shading_normal(const TriRow<float> &T, bool face, float bu, float bv) {
    ShNormal r; r.face = face;
    if (face) { r.v = r.n = T.fn; return r; }
    r.v = bary_point(T.n0, T.n1 - T.n0, T.n2 - T.n0, bu, bv);
    r.n = normalize(r.v);
    return r;
}
template <class Sink>
PSDR_HD void shading_normal_vjp(Sink &sink, int tri, const TriRow<float> &T, const ShNormal &sn, float bu, float bv, const Vec3f &an,
                                float &abu, float &abv) {
    if (sn.face) { scatter_vec(sink, tri, 18, an); return; }
    const Vec3f av = normalize_vjp(sn.v, sn.n, an);
    scatter_vec(sink, tri, 9, av * (1.f - bu - bv));
    scatter_vec(sink, tri, 12, av * bu);
    scatter_vec(sink, tri, 15, av * bv);
    abu += dot(av, T.n1 - T.n0); abv += dot(av, T.n2 - T.n0);
}

// ------------------------------------------------------------------ BSDF with adjoints
// Texture lookup adjoint: scatter a_out (C channels) to the texels, return d/d(u,v).
template <class Sink, int C>
PSDR_HD void bitmap_vjp(Sink &sink, const SceneView &sc, const int32_t *slot, float u, float v, const float *a_out, float &au, float &av,
                        bool flip_v = true) {
    const int off = slot[0], w = slot[1], h = slot[2];
    if (w == 1 && h == 1) {
#pragma unroll
        for (int c = 0; c < C; ++c) sink.add_texel(off + c, a_out[c]);
        return;
    }
    const float *tx = sc.d.texels;
    float vv = flip_v ? -v : v;
    u = u - floorf(u); vv = vv - floorf(vv);
    u = u * (float) (w - 1); vv = vv * (float) (h - 1);
    int px = (int) floorf(u), py = (int) floorf(vv);
    const float w1x = u - (float) px, w1y = vv - (float) py, w0x = 1.f - w1x, w0y = 1.f - w1y;
    px = px < w - 2 ? px : w - 2; py = py < h - 2 ? py : h - 2;
    const size_t idx = (size_t) py * w + px;
    float du = 0.f, dv = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const size_t i00 = off + idx * C + c, i10 = off + (idx + 1) * C + c, i01 = off + (idx + w) * C + c, i11 = off + (idx + w + 1) * C + c;
        const float a = a_out[c];
        sink.add_texel((int) i00, a * w0y * w0x); sink.add_texel((int) i10, a * w0y * w1x);
        sink.add_texel((int) i01, a * w1y * w0x); sink.add_texel((int) i11, a * w1y * w1x);
        const float v00 = tx[i00], v10 = tx[i10], v01 = tx[i01], v11 = tx[i11];
        du += a * (w0y * (v10 - v00) + w1y * (v11 - v01));
        dv += a * ((w0x * v01 + w1x * v11) - (w0x * v00 + w1x * v10));
    }
    au += du * (float) (w - 1);
    av += (flip_v ? -dv : dv) * (float) (h - 1);        // v was negated (flip_v)
}

// EnvironmentMap::eval_direction adjoint (envmap.cpp:41-59): scatters a_out (adjoint of the returned
// radiance) to the map's texels, m_scale and m_from_world, returns the adjoint of the direction w.
template <class Sink> PSDR_HD Vec3f env_eval_vjp(Sink &sink, const SceneView &sc, const Vec3f &w, const Vec3f &a_out) {
    const float *f = sc.d.env_f;
    const Vec3f v = env_mul3(f + PSDR_ENV_FROM_WORLD, w);
    float tu = atan2f(v.x, -v.z) * kInvTwoPi, tv = safe_acos_(v.y) * kInvPi;
    tu -= floorf(tu); tv -= floorf(tv);
    const TangentView<0, kSceneEnv> tv0{};
    float rgb[3];
    bitmap_eval<float, 3>(sc, tv0, sc.d.env_tex, tu, tv, rgb, false);
    const float scale = f[PSDR_ENV_SCALE];
    sink.add_env(PSDR_ENV_SCALE, a_out.x * rgb[0] + a_out.y * rgb[1] + a_out.z * rgb[2]);
    const float a_rgb[3] = {a_out.x * scale, a_out.y * scale, a_out.z * scale};
    float a_tu = 0.f, a_tv = 0.f;
    bitmap_vjp<Sink, 3>(sink, sc, sc.d.env_tex, tu, tv, a_rgb, a_tu, a_tv, false);
    // tu = atan2(v.x, -v.z) / 2pi ; tv = acos(clamp(v.y)) / pi
    const float r2 = v.x * v.x + v.z * v.z;
    const float au = a_tu * kInvTwoPi / r2;
    Vec3f a_v{au * (-v.z), 0.f, au * v.x};
    if (v.y > -1.f && v.y < 1.f) a_v.y = -a_tv * kInvPi / sqrtf(1.f - v.y * v.y);
    const float av3[3] = {a_v.x, a_v.y, a_v.z}, w3[3] = {w.x, w.y, w.z};
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) sink.add_env(PSDR_ENV_FROM_WORLD + r * 3 + c, av3[r] * w3[c]);
    const float *m = f + PSDR_ENV_FROM_WORLD;
    return {m[0] * a_v.x + m[3] * a_v.y + m[6] * a_v.z, m[1] * a_v.x + m[4] * a_v.y + m[7] * a_v.z, m[2] * a_v.x + m[5] * a_v.y + m[8] * a_v.z};
}

// value and partials of one RoughConductor quantity w.r.t. the packed inputs
//   [0..2] wi, [3..5] wo, [6] alpha_u, [7] alpha_v   (Dual<8>)
using D8 = Dual<8>;
PSDR_HD D8 seed8(float v, int i) { D8 r(v); r.d[i] = 1.f; return r; }

struct RcParams { float au, av; Vec3f eta, k, spec; };

// Hand-written reverse mode of the geometric part of the GGX rough conductor (ggx.cpp:9-32,
// roughconductor.cpp:40-75):   H = normalize(wi + wo),  D(H; au, av),  G1(v, H; au, av),
//     geo = D G1(wi) G1(wo) / (4 wi.z)   (eval)      or      D G1(wi) / (4 wi.z)   (pdf, WITH_GO = false),
//     c   = wi . H                                        (the Fresnel cosine; a_c = 0 for the pdf)
// Values are recomputed exactly as GGX<float> computes them (same masks); ~100 flops instead of running
// the forward code on Dual<8> numbers (9x the registers and instructions of the value).
struct GgxAdj { Vec3f wi, wo; float au, av; float geo, c; bool zero; };
template <bool WITH_GO>
PSDR_HD GgxAdj ggx_geo_vjp(const Vec3f &wi, const Vec3f &wo, float au, float av, float a_geo, float a_c) {
    GgxAdj r; r.wi = Vec3f(0.f); r.wo = Vec3f(0.f); r.au = r.av = 0.f; r.geo = 0.f; r.c = 0.f; r.zero = true;
    const Vec3f s = wo + wi;
    const float L = norm(s);
    const Vec3f H = s / L;
    const GGX<float> g{au, av};
    const float q = sqr(H.x / au) + sqr(H.y / av) + sqr(H.z);
    const float D = g.eval(H);
    if (D == 0.f) return r;
    const float g1i = g.smith_g1(wi, H), g1o = WITH_GO ? g.smith_g1(wo, H) : 1.f;
    const float inv4 = 1.f / (4.f * wi.z);
    r.geo = D * (g1i * g1o) * inv4;
    r.c = dot(wi, H);
    r.zero = false;
    // geo = D g1i g1o inv4
    const float a_D = a_geo * g1i * g1o * inv4, a_g1i = a_geo * D * g1o * inv4, a_g1o = WITH_GO ? a_geo * D * g1i * inv4 : 0.f;
    r.wi.z += -a_geo * r.geo / wi.z;
    Vec3f a_H = wi * a_c;
    acc(r.wi, H * a_c);
    // G1(v) = 2 / (1 + sqrt(1 + xy / vz^2)),  xy = (au vx)^2 + (av vy)^2; the masks are piecewise constant
    auto g1_vjp = [&](const Vec3f &v, float a_r, Vec3f &a_v) {
        const float xy = sqr(au * v.x) + sqr(av * v.y);
        if (a_r == 0.f || xy == 0.f || dot(v, H) * v.z <= 0.f) return;
        const float iz2 = 1.f / sqr(v.z), t = xy * iz2, sq = sqrtf(1.f + t);
        const float a_t = -a_r / (sqr(1.f + sq) * sq);
        const float a_xy = a_t * iz2;
        a_v.z += a_t * (-2.f * t / v.z);
        a_v.x += a_xy * 2.f * au * au * v.x; a_v.y += a_xy * 2.f * av * av * v.y;
        r.au += a_xy * 2.f * au * v.x * v.x; r.av += a_xy * 2.f * av * v.y * v.y;
    };
    g1_vjp(wi, a_g1i, r.wi);
    if (WITH_GO) g1_vjp(wo, a_g1o, r.wo);
    // D = 1 / (pi au av q^2)
    const float a_q = a_D * (-2.f * D / q);
    r.au += -a_D * D / au - a_q * 2.f * sqr(H.x) / (au * au * au);
    r.av += -a_D * D / av - a_q * 2.f * sqr(H.y) / (av * av * av);
    a_H.x += a_q * 2.f * H.x / (au * au); a_H.y += a_q * 2.f * H.y / (av * av); a_H.z += a_q * 2.f * H.z;
    // H = s / |s|,  s = wi + wo
    const Vec3f a_s = (a_H - H * dot(H, a_H)) / L;
    acc(r.wi, a_s); acc(r.wo, a_s);
    return r;
}

template <class Sink> struct BsdfRev {
    const SceneView &sc;
    Bsdf<float, float> b;
    PSDR_HD BsdfRev(const SceneView &s, int id) : sc(s), b(s, TangentView<0, Sink::flags>{}, id) {}

    template <class TVT> PSDR_HD RcParams rc_params(const TVT &tv0, const Its<float> &its) const {
        RcParams p;
        p.au = b.tex1(sc, tv0, PSDR_SLOT_ALPHA_U, its); p.av = b.tex1(sc, tv0, PSDR_SLOT_ALPHA_V, its);
        p.eta = b.tex3(sc, tv0, PSDR_SLOT_ETA, its); p.k = b.tex3(sc, tv0, PSDR_SLOT_K, its);
        p.spec = b.tex3(sc, tv0, PSDR_SLOT_REFLECTANCE, its);
        return p;
    }

    // adjoint of value = eval(its, wo): a_f (RGB) -> a_wi, a_wo, texels, a_uv
    template <class TVT> PSDR_HD void eval_vjp(Sink &sink, const TVT &tv0, const Its<float> &its, const Vec3f &wo, const Vec3f &af, Vec3f &awi,
                          Vec3f &awo, float &auvx, float &auvy) const {
        if (!(its.wi.z > 0.f && wo.z > 0.f)) return;
        if (b.is_diffuse(tv0)) {
            const Vec3f rho = b.tex3(sc, tv0, PSDR_SLOT_REFLECTANCE, its);
            const float c = wo.z * kInvPi;
            const float ar[3] = {af.x * c, af.y * c, af.z * c};
            bitmap_vjp<Sink, 3>(sink, sc, b.slot(PSDR_SLOT_REFLECTANCE), its.uvx, its.uvy, ar, auvx, auvy);
            awo.z += dot(af, rho) * kInvPi;
            return;
        }
        const RcParams p = rc_params(tv0, its);
        // value pass of the geometric part g(wi, wo, alpha) = D*G/(4 cos_i) and cos = wi.H (adjoint below)
        const GgxAdj v0 = ggx_geo_vjp<true>(its.wi, wo, p.au, p.av, 0.f, 0.f);
        if (v0.zero) return;
        struct { float v; } geo{v0.geo}, c{v0.c};
        // Fresnel per channel with Dual<3> over (cos, eta, k)
        using D3 = Dual<3>;
        const float *eta = &p.eta.x, *kk = &p.k.x, *spec = &p.spec.x, *afp = &af.x;
        float a_geo = 0.f, a_cos = 0.f, a_eta[3], a_k[3], a_spec[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            D3 dc(c.v); dc.d[0] = 1.f;
            D3 de(eta[ch]); de.d[1] = 1.f;
            D3 dk(kk[ch]); dk.d[2] = 1.f;
            const D3 F = fresnel_conductor(de, dk, dc);
            // value_ch = F * geo * spec
            const float a = afp[ch];
            a_spec[ch] = a * F.v * geo.v;
            a_geo += a * F.v * spec[ch];
            const float aF = a * geo.v * spec[ch];
            a_cos += aF * F.d[0]; a_eta[ch] = aF * F.d[1]; a_k[ch] = aF * F.d[2];
        }
        const GgxAdj ga = ggx_geo_vjp<true>(its.wi, wo, p.au, p.av, a_geo, a_cos);
        acc(awi, ga.wi); acc(awo, ga.wo);
        bitmap_vjp<Sink, 1>(sink, sc, b.slot(PSDR_SLOT_ALPHA_U), its.uvx, its.uvy, &ga.au, auvx, auvy);
        bitmap_vjp<Sink, 1>(sink, sc, b.slot(PSDR_SLOT_ALPHA_V), its.uvx, its.uvy, &ga.av, auvx, auvy);
        bitmap_vjp<Sink, 3>(sink, sc, b.slot(PSDR_SLOT_ETA), its.uvx, its.uvy, a_eta, auvx, auvy);
        bitmap_vjp<Sink, 3>(sink, sc, b.slot(PSDR_SLOT_K), its.uvx, its.uvy, a_k, auvx, auvy);
        bitmap_vjp<Sink, 3>(sink, sc, b.slot(PSDR_SLOT_REFLECTANCE), its.uvx, its.uvy, a_spec, auvx, auvy);
    }

    // adjoint of pdf(its, wo) (only RoughConductor carries derivatives; Diffuse::__pdf is detached)
    template <class TVT> PSDR_HD void pdf_vjp(Sink &sink, const TVT &tv0, const Its<float> &its, const Vec3f &wo, float apdf, Vec3f &awi, Vec3f &awo,
                         float &auvx, float &auvy) const {
        if (b.is_diffuse(tv0) || apdf == 0.f) return;
        const RcParams p = rc_params(tv0, its);
        const GgxAdj ga = ggx_geo_vjp<false>(its.wi, wo, p.au, p.av, apdf, 0.f);
        if (ga.zero) return;
        acc(awi, ga.wi); acc(awo, ga.wo);
        bitmap_vjp<Sink, 1>(sink, sc, b.slot(PSDR_SLOT_ALPHA_U), its.uvx, its.uvy, &ga.au, auvx, auvy);
        bitmap_vjp<Sink, 1>(sink, sc, b.slot(PSDR_SLOT_ALPHA_V), its.uvx, its.uvy, &ga.av, auvx, auvy);
    }

    // adjoint of the SAMPLED pdf: pdf_s = pdf(its, wo_s(wi, alpha; xi))  (roughconductor.cpp:79-92 keeps this
    // dependency alive; Diffuse: constant).  Written out by hand -- the value chain of GGX::sample / visible11 (psdr_device.h) once,
    // then its adjoint backwards: ~250 flops instead of pushing five tangents through the chain on Dual<5> numbers (6x the
    // registers of every intermediate, in a kernel that spills).  The reflected direction's half vector IS the sampled normal m,
    // so pdf_s = D(m) G1(wi, m) / (4 wi.z) with m = m(wi, alpha; xi).  Branches (clamps, safe_sqrt, the masks of D and G1) follow the
    // value like the dual-number forms do (psdr_math.h max_ / min_ / safe_sqrt).
    template <class TVT> PSDR_HD void sampled_pdf_vjp(Sink &sink, const TVT &tv0, const Its<float> &its, const float s[3], float apdf, Vec3f &awi,
                                 float &auvx, float &auvy) const {
        if (b.is_diffuse(tv0) || apdf == 0.f) return;
        const RcParams p = rc_params(tv0, its);
        const float au = p.au, av = p.av;
        const Vec3f wi = its.wi;
        // ---- values
        const Vec3f v{au * wi.x, av * wi.y, wi.z};
        const float L = norm(v);
        const Vec3f wp = v / L;
        const float st2 = wp.x * wp.x + wp.y * wp.y;
        const bool has_phi = !(fabsf(st2) <= 4.f * kEpsilon);
        float sp = 0.f, cp = 1.f, inv = 0.f;
        bool sp_free = false, cp_free = false;                  // inside the clamp: the derivative passes
        if (has_phi) {
            inv = 1.f / sqrtf(st2);
            const float rs = wp.y * inv, rc = wp.x * inv;
            sp_free = rs > -1.f && rs < 1.f; cp_free = rc > -1.f && rc < 1.f;
            sp = fminf(fmaxf(rs, -1.f), 1.f); cp = fminf(fmaxf(rc, -1.f), 1.f);
        }
        float px, py;
        concentric_disk(s[0], s[1], px, py);
        const float c = wp.z, sh = 0.5f * (1.f + c), k0 = sqrtf(fmaxf(1.f - px * px, 0.f));
        const float y = k0 * (1.f - sh) + py * sh;
        const float zarg = 1.f - (px * px + y * y), z = zarg > 0.f ? sqrtf(zarg) : 0.f;
        const float sarg = 1.f - c * c, si = sarg > 0.f ? sqrtf(sarg) : 0.f;
        const float nrm = 1.f / (si * y + c * z);
        const float slx = (c * y - si * z) * nrm, sly = px * nrm;
        const float b0 = cp * slx - sp * sly, b1 = sp * slx + cp * sly;
        const Vec3f u{-(b0 * au), -(b1 * av), 1.f};
        const float Lu = norm(u);
        const Vec3f m = u / Lu;
        const float q = sqr(m.x / au) + sqr(m.y / av) + sqr(m.z);
        const float D = 1.f / (kPi * au * av * sqr(q));
        if (!(D * m.z > 1e-5f)) return;                          // D masked to zero: pdf_s = 0, nothing depends on anything
        const float xy = sqr(au * wi.x) + sqr(av * wi.y);
        const bool g_zero = dot(wi, m) * wi.z <= 0.f, g_one = xy == 0.f;
        const float iz2 = 1.f / sqr(wi.z), tt = xy * iz2, sq = sqrtf(1.f + tt);
        const float G1 = g_zero ? 0.f : (g_one ? 1.f : 2.f / (1.f + sq));
        const float inv4 = 1.f / (4.f * wi.z);
        const float F = D * G1 * inv4;
        // ---- adjoints
        float a_au = 0.f, a_av = 0.f;
        Vec3f a_wi(0.f);
        const float a_D = apdf * G1 * inv4, a_G1 = apdf * D * inv4;
        a_wi.z += -apdf * F / wi.z;
        if (!g_zero && !g_one && a_G1 != 0.f) {                  // G1 = 2 / (1 + sqrt(1 + xy / wi.z^2))
            const float a_t = -a_G1 / (sqr(1.f + sq) * sq);                        // d/dt [2 / (1 + sqrt(1 + t))] = -1 / ((1 + sq)^2 sq)
            const float a_xy = a_t * iz2;
            a_wi.z += a_t * (-2.f * tt / wi.z);
            a_wi.x += a_xy * 2.f * au * au * wi.x; a_wi.y += a_xy * 2.f * av * av * wi.y;
            a_au += a_xy * 2.f * au * wi.x * wi.x; a_av += a_xy * 2.f * av * wi.y * wi.y;
        }
        // D = 1 / (pi au av q^2)
        const float a_q = a_D * (-2.f * D / q);
        a_au += -a_D * D / au - a_q * 2.f * sqr(m.x) / (au * au * au);
        a_av += -a_D * D / av - a_q * 2.f * sqr(m.y) / (av * av * av);
        const Vec3f a_m{a_q * 2.f * m.x / (au * au), a_q * 2.f * m.y / (av * av), a_q * 2.f * m.z};
        // m = u / |u|, u = (-b0 au, -b1 av, 1)
        const Vec3f a_u = (a_m - m * dot(m, a_m)) / Lu;
        const float a_s0 = -a_u.x, a_s1 = -a_u.y;               // s0 = b0 au, s1 = b1 av
        a_au += a_s0 * b0; a_av += a_s1 * b1;
        const float a_b0 = a_s0 * au, a_b1 = a_s1 * av;
        float a_cp = a_b0 * slx + a_b1 * sly, a_sp = -a_b0 * sly + a_b1 * slx;
        const float a_slx = a_b0 * cp + a_b1 * sp, a_sly = -a_b0 * sp + a_b1 * cp;
        // visible11: slx = (c y - si z) nrm, sly = px nrm, nrm = 1 / (si y + c z)
        const float a_nrm = a_slx * (c * y - si * z) + a_sly * px;
        float a_c = a_slx * y * nrm, a_y = a_slx * c * nrm, a_si = -a_slx * z * nrm, a_z = -a_slx * si * nrm;
        const float a_den = -a_nrm * nrm * nrm;
        a_si += a_den * y; a_y += a_den * si; a_c += a_den * z; a_z += a_den * c;
        if (si > 0.f) a_c += a_si * (-c / si);                   // safe_sqrt: no derivative at or below zero
        if (z > 0.f) a_y += a_z * (-y / z);
        a_c += 0.5f * a_y * (py - k0);                           // y = k0 (1 - sh) + py sh, sh = (1 + c) / 2
        // phi: sp = clamp(wp.y inv), cp = clamp(wp.x inv), inv = st2^(-1/2)
        Vec3f a_wp{0.f, 0.f, a_c};
        if (has_phi) {
            if (!sp_free) a_sp = 0.f;
            if (!cp_free) a_cp = 0.f;
            const float a_inv = a_sp * wp.y + a_cp * wp.x;
            a_wp.y += a_sp * inv; a_wp.x += a_cp * inv;
            const float a_st2 = a_inv * (-0.5f * inv * inv * inv);
            a_wp.x += a_st2 * 2.f * wp.x; a_wp.y += a_st2 * 2.f * wp.y;
        }
        // wp = v / |v|, v = (au wi.x, av wi.y, wi.z)
        const Vec3f a_v = (a_wp - wp * dot(wp, a_wp)) / L;
        a_wi.x += a_v.x * au; a_wi.y += a_v.y * av; a_wi.z += a_v.z;
        a_au += a_v.x * wi.x; a_av += a_v.y * wi.y;
        acc(awi, a_wi);
        bitmap_vjp<Sink, 1>(sink, sc, b.slot(PSDR_SLOT_ALPHA_U), its.uvx, its.uvy, &a_au, auvx, auvy);
        bitmap_vjp<Sink, 1>(sink, sc, b.slot(PSDR_SLOT_ALPHA_V), its.uvx, its.uvy, &a_av, auvx, auvy);
    }
};

// ---------------------------------------------------------------- per-vertex evaluation
// Moeller-Trumbore adjoint (include/psdr/utils.h:66-77): given (a_u, a_v, a_t) returns the adjoints
// of p0, e1, e2 and of the ray origin / direction.
struct MtAdj { Vec3f p0, e1, e2, o, d; };
PSDR_HD MtAdj mt_vjp(const Vec3f &p0, const Vec3f &e1, const Vec3f &e2, const RayT<float> &ray, float au, float av, float at) {
    const Vec3f h = cross(ray.d, e2);
    const float f = 1.f / dot(e1, h);
    const Vec3f s = ray.o - p0, q = cross(s, e1);
    const float a_f = au * dot(s, h) + av * dot(ray.d, q) + at * dot(e2, q);
    Vec3f as = h * (au * f), ah = s * (au * f);
    Vec3f ad = q * (av * f), aq = ray.d * (av * f);
    Vec3f ae2 = q * (at * f);
    acc(aq, e2 * (at * f));
    const float a_a = -a_f * f * f;
    Vec3f ae1 = h * a_a;
    acc(ah, e1 * a_a);
    acc(as, cross(e1, aq)); acc(ae1, cross(aq, s));      // q = s x e1
    acc(ad, cross(e2, ah)); acc(ae2, cross(ah, ray.d));  // h = d x e2
    MtAdj r; r.p0 = -as; r.e1 = ae1; r.e2 = ae2; r.o = as; r.d = ad;
    return r;
}

// camera ray adjoint: o = to_world[:,3] (w = 1), d = R * dcam  (perspective.cpp:130-135)
template <class Sink> PSDR_HD void camera_ray_vjp(Sink &sink, const SceneView &sc, const Vec3f &dcam, const Vec3f &ao, const Vec3f &ad) {
    const float o3[3] = {ao.x, ao.y, ao.z}, d3[3] = {ad.x, ad.y, ad.z};
    {   // o = to_world[:,3] / w
        const float *c = sc.d.cam + PSDR_CAM_TO_WORLD;
        const float iw = 1.f / uniform_word(c, 15);
        sink.add_cam(15, -(ao.x * uniform_word(c, 3) + ao.y * uniform_word(c, 7) + ao.z * uniform_word(c, 11)) * iw * iw);
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        sink.add_cam(r * 4 + 3, o3[r] / uniform_word(sc.d.cam, PSDR_CAM_TO_WORLD + 15));
        sink.add_cam(r * 4 + 0, d3[r] * dcam.x); sink.add_cam(r * 4 + 1, d3[r] * dcam.y); sink.add_cam(r * 4 + 2, d3[r] * dcam.z);
    }
}
PSDR_HD Vec3f camera_space_dir(const SceneView &sc, float sx, float sy) {
    const float *m = sc.d.cam + PSDR_CAM_SAMPLE_TO_CAMERA;
    float v4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v4[r] = uniform_word(m, r * 4) * sx + uniform_word(m, r * 4 + 1) * sy + uniform_word(m, r * 4 + 3);
    return normalize(Vec3f{v4[0] / v4[3], v4[1] / v4[3], v4[2] / v4[3]});
}

constexpr int kMaxRevDepth = 8;          // path vertices whose record fits in LDS (8 KB per vertex and workgroup)
constexpr int kMaxRevDepthDeep = 250;    // beyond: the record lives in HBM (render_rev), as deep as the forward wavefront goes
constexpr int kPathRecWords = 8;

// Per-lane record of (c_k, f_k) and the two hit triangles along the path (8 words per vertex).  On the device it lives in LDS
// (one column per lane) -- a register array indexed by the run-time vertex number would be demoted to
// scratch, and the scratch of all resident waves (hundreds of MB) thrashes the caches.  Paths deeper than kMaxRevDepth keep
// it in HBM instead: one column per resident thread of the (persistent) grid, word w of thread t at deep[w * threads + t].
struct PathRec {
#if defined(__HIP_DEVICE_COMPILE__)
    float *base;                 // &lds[threadIdx.x], stride kBlock
    float *deep = nullptr;       // HBM block of the launch (wave-uniform: stays in scalar registers) or null
    uint32_t deep_stride = 0;    // threads of the grid
    uint32_t deep_col = 0;       // this thread's column
    __device__ __forceinline__ void put(int k, int c, float v) {
        if (deep) deep[(size_t) (k * kPathRecWords + c) * deep_stride + deep_col] = v; else base[(k * kPathRecWords + c) * kBlock] = v;
    }
    __device__ __forceinline__ float get(int k, int c) const {
        return deep ? deep[(size_t) (k * kPathRecWords + c) * deep_stride + deep_col] : base[(k * kPathRecWords + c) * kBlock];
    }
#else
    float a[kMaxRevDepthDeep * kPathRecWords];
    void put(int k, int c, float v) { a[k * kPathRecWords + c] = v; }
    float get(int k, int c) const { return a[k * kPathRecWords + c]; }
#endif
    // triangles the two rays of vertex k arrived at in sweep 1 (-1: miss): sweep 2 re-intersects THAT triangle
    // instead of walking the tree again
    PSDR_HD void put_tri(int k, int which, int tri) { put(k, 6 + which, __int_as_float_hd(tri)); }
    PSDR_HD int tri(int k, int which) const { return __float_as_int_hd(get(k, 6 + which)); }
    PSDR_HD void put_cf(int k, const Vec3f &c, const Vec3f &f) { put(k, 0, c.x); put(k, 1, c.y); put(k, 2, c.z); put(k, 3, f.x); put(k, 4, f.y); put(k, 5, f.z); }
    PSDR_HD Vec3f c(int k) const { return {get(k, 0), get(k, 1), get(k, 2)}; }
    PSDR_HD Vec3f f(int k) const { return {get(k, 3), get(k, 4), get(k, 5)}; }
};

// Split reverse launch (tree scenes): the value sweep runs as its own kernel at the occupancy of a forward kernel and leaves,
// per path, what the adjoint kernel needs instead of tracing again -- the primary triangle, the number of vertices, and per
// vertex the suffix radiance T_{k+1} and the two triangles its rays arrived at.  One column per path, word w at p[w * stride].
// cf = 1 (the value sweep ran as the TRACED WAVEFRONT, psdr_kernels.h): per vertex (c_k, f_k, tri, tri) as the stage that evaluated the vertex
// left them; the adjoint kernel forms the suffix radiances itself.
// (Round 5 also built the adjoint sweep as one launch per path vertex on a 10-word layout with a 23-word state column: parity-green, 1.3-1.5x slower than the one
// adjoint kernel on C2 / C4 / C5 -- profiles/r05_vertex_rev_*.txt, DESIGN.md round 5 -- and removed in round 6.)
constexpr int kRevDiskHead = 2, kRevDiskPerVertex = 5, kRevDiskPerVertexCf = 8;
struct RevDisk {
    float *p; long long stride; int cf = 0;
    PSDR_HD void put(int w, float v) const { p[(long long) w * stride] = v; }
    PSDR_HD float get(int w) const { return p[(long long) w * stride]; }
    PSDR_HD void puti(int w, int v) const { put(w, __int_as_float_hd(v)); }
    PSDR_HD int geti(int w) const { return __float_as_int_hd(get(w)); }
};

// What one vertex hands back to the path loop
struct VertexOut {
    Vec3f c;            // sum of the emitter contributions gathered at this vertex (unit throughput)
    Vec3f f;            // throughput of the first BSDF sample (D form: eval * G * J / pdf0)
    Its<float> next;    // BSDF-sampled next vertex (path-space form)
    bool next_valid;
};

template <int FL> PSDR_HD Its<float> make_path_vertex(const SceneView &sc, const Vec3f &origin, int tri, float hu, float hv, const TriRow<float> &T) {
    Its<float> n;
    const int tm = Tab<FL>::tri_mesh(sc, tri);
    n.valid = true; n.tri = tri; n.mesh = tm & ~PSDR_TRI_FACE_NORMALS; n.hu = hu; n.hv = hv;
    n.n = T.fn; n.J = 1.f;
    n.p = bary_point(T.p0, T.e1, T.e2, hu, hv);
    Vec3f dir = n.p - origin;
    n.t = norm(dir);
    dir = dir / n.t;
    const ShNormal sn = shading_normal(T, (tm & PSDR_TRI_FACE_NORMALS) != 0, hu, hv);
    n.sh = Frame<float>(sn.n);
    n.wi = n.sh.to_local(-dir);
    const float *q = sc.d.tri_uv ? Tab<FL>::tri_uv(sc, tri) : nullptr;
    n.uvx = q ? (q[2] - q[0]) * hu + ((q[4] - q[0]) * hv + q[0]) : 0.f;
    n.uvy = q ? (q[3] - q[1]) * hu + ((q[5] - q[1]) * hv + q[1]) : 0.f;
    return n;
}

// Evaluates the terms gathered at vertex `its` in their D-mode forms (direct.cpp:64-160) and, when
// BACKWARD, differentiates them:
//   a_c : adjoint of every emitter contribution (= a * beta_k)
//   a_f : adjoint of the continuation throughput f_k (= a * beta_k * T_{k+1}); zero for DirectIntegrator
// Adjoints of THIS vertex' position / frame / wi / uv accumulate in `va`; everything that belongs to
// other triangles (next vertex, emitter) is scattered straight into the sink.
//   REPLAY (one BSDF and one light sample per vertex): the value sweep records the triangle each of the two
//   rays arrived at in rec[k]; the BACKWARD sweep re-intersects that triangle instead of tracing again.
// Order of the two estimators in the ADJOINT sweep of a replayed vertex (geometry sinks): 0 = BSDF sample first, 1 = light sample first, 2 = light
// sample first in the rough-conductor instances only.  Measured (profiles/r04_rev_sink_ab.txt): the rough instances spill less with the light
// sample first (adjoint kernel of the split launch 42 -> 15 spilled VGPRs; C5 DirectIntegrator reverse 3.14 -> 2.95 ms, PathTracer(3) 5.59 -> 5.46);
// the diffuse ones do not spill either way and are level (C2 all gradients 5.32 = 5.32 ms; 240 -> 230 VGPRs, far from a third wave's 168).
#ifndef PSDR_REV_LIGHT_FIRST
#define PSDR_REV_LIGHT_FIRST 2
#endif
template <bool BACKWARD, bool REPLAY, class Sink, class Rec = PathRec>
PSDR_HD VertexOut vertex_eval(Sink &sink, const SceneView &sc, TraversalStack &st, Rng &rng, const Its<float> &its, int nB, int nL,
                              const Vec3f &a_c, const Vec3f &a_f, VertexAdj &va, uint32_t &nrays, Rec &rec, int k, RowAdj *next_row = nullptr, bool emitter_only = false) {
    const TangentView<0, Sink::flags> tv0{};
    VertexOut out; out.c = Vec3f(0.f); out.f = Vec3f(0.f); out.next_valid = false;
    // the bounding mesh of the environment map has no BSDF (direct.cpp:54-57): nothing is gathered there and
    // the path ends (the caller stops on !next_valid, so the skipped random numbers are never missed)
    const int bsdf_id = Tab<Sink::flags>::mesh_bsdf(sc, its.mesh);
    if (bsdf_id < 0) return out;
    constexpr bool kEnv = Sink::has_env;
    BsdfRev<Sink> brev(sc, bsdf_id);
    const MatCache<float> mat = brev.b.fetch(sc, tv0, its);        // the vertex' material parameters, looked up once
    brev.b.mc = &mat;
    const Bsdf<float, float> &bsdf = brev.b;
    if (REPLAY && !BACKWARD) { rec.put_tri(k, 0, -1); rec.put_tri(k, 1, -1); }
    // The two estimators of a vertex as closures: the numbers are drawn in the reference's order (three for the BSDF sample, two for
    // the light sample), the ORDER OF EVALUATION is the caller's below.
    auto bsdf_term = [&](const int i, const float (&s)[3]) {
        Vec3f wo_s; float pdf_s;
        const bool ok = bsdf.sample(sc, tv0, its, s, true, wo_s, pdf_s);
        if (!ok) return;
        const RayT<float> ray1{its.p, its.sh.to_world(wo_s)};
        Hit h1;
        if (REPLAY && BACKWARD) h1.tri = rec.tri(k, 0);
        else {
            // emitter_only (direct_step): the hit matters only on an emitter -- the emitters' primitives first, no trace for a ray that meets none of them
            bool wanted = true;
            if constexpr (!Sink::has_env && (Sink::flags & kSceneForest) != 0 && PSDR_EMITTER_PRETEST) {
                if (emitter_only && sc.emit_rows != 0u) wanted = closest_hit<false, 2, true>(sc, st, ray1.o, ray1.d, INFINITY, -1, -1, 0, sc.emit_rows).tri >= 0;
            }
            if (wanted) {
                nrays++;
                h1 = closest_hit<false, tree_mode<Sink::flags>()>(sc, st, ray1.o, ray1.d, INFINITY, -1, -1, kPreBsdfRay);
            } else h1.tri = -1;
            if (REPLAY) rec.put_tri(k, 0, h1.tri);
        }
        if (h1.tri < 0) return;
        const int tm = Tab<Sink::flags>::tri_mesh(sc, h1.tri), mesh1 = tm & ~PSDR_TRI_FACE_NORMALS;
        const TriRow<float> T1 = load_tri<float>(sc, tv0, h1.tri);
        if (REPLAY && BACKWARD) h1 = hit_on_triangle(h1.tri, T1.p0, T1.e1, T1.e2, ray1.o, ray1.d);
        const Vec3f p1 = bary_point(T1.p0, T1.e1, T1.e2, h1.u, h1.v);
        const Vec3f dvec = p1 - its.p;
        const float t1 = norm(dvec);
        const Vec3f wo = dvec / t1;
        const Vec3f wol = its.sh.to_local(wo);
        const Vec3f f = bsdf.eval(sc, tv0, its, wol, true);
        const float cosv = -dot(T1.fn, wo);
        const float G = fabsf(cosv) / (t1 * t1);
        const float pdf0 = pdf_s * G;
        const float cfac = 1.f / pdf_s;                       // G * J / pdf0 with J = 1, pdf0 = pdf_s * G
        const Vec3f valv = f * cfac;
        const int e1 = Tab<Sink::flags>::mesh_emitter(sc, mesh1);
        Vec3f Le1(0.f);
        float w = 1.f / (float) nB, dw_dpdf0 = 0.f;
        const bool env1 = kEnv && e1 >= 0 && e1 == sc.d.env_emitter;
        if (e1 >= 0) {
            if (env1) Le1 = env_eval_direction<float>(sc, tv0, wo);
            else {
                const ShNormal sn1 = shading_normal(T1, (tm & PSDR_TRI_FACE_NORMALS) != 0, h1.u, h1.v);
                if (-dot(wo, sn1.n) > 0.f) { const float *r = Tab<Sink::flags>::emitter_f(sc, e1); Le1 = Vec3f{r[0], r[1], r[2]}; }
            }
            if (nL > 0) {
                const float *ef = Tab<Sink::flags>::emitter_f(sc, e1);
                const float pe = env1 ? env_position_pdf(sc, its.p, p1, T1.fn) : ef[3] * ef[4];
                const float a2 = pdf0 * pdf0, b2 = pe * pe, den = a2 + b2;
                w *= a2 / den;
                dw_dpdf0 = (2.f * pdf0 * b2 / (den * den)) / (float) nB;
            }
            out.c = out.c + Le1 * valv * w;
        }
        if (i == 0) { out.f = valv; out.next = make_path_vertex<Sink::flags>(sc, its.p, h1.tri, h1.u, h1.v, T1); out.next_valid = true; }
        if (!BACKWARD) return;
        Vec3f a_val = a_c * Le1 * w;
        const float a_w = dot(a_c, Le1 * valv);
        if (i == 0) acc(a_val, a_f);
        Vec3f a_wo(0.f);
        bool le_dir = false;
        if (env1) {
            if constexpr (kEnv) { a_wo = env_eval_vjp(sink, sc, wo, a_c * valv * w); le_dir = true; }
        } else if (e1 >= 0 && (Le1.x != 0.f || Le1.y != 0.f || Le1.z != 0.f)) {
            const Vec3f gr = a_c * valv * w;
            sink.add_rad(e1, 0, gr.x); sink.add_rad(e1, 1, gr.y); sink.add_rad(e1, 2, gr.z);
        }
        if (a_val.x == 0.f && a_val.y == 0.f && a_val.z == 0.f && a_w == 0.f && !le_dir) return;
        float a_pdf0 = a_w * dw_dpdf0;
        const Vec3f a_fv = a_val * cfac;
        const float a_cfac = dot(a_val, f);
        const float a_G = a_cfac / pdf0;                       // cfac = G * J / pdf0
        const float a_J = a_cfac * cfac;
        a_pdf0 += -a_cfac * cfac / pdf0;
        const float a_pdf_s = a_pdf0 * G;                      // pdf0 = pdf_s * detach(G)
        const float a_cosv = a_G * (cosv < 0.f ? -1.f : 1.f) / (t1 * t1);
        float a_t1 = -2.f * a_G * G / t1;
        // the row of the next vertex: J = A / detach(A), cosv = -fn . wo, and (below) its position
        const bool merge = next_row != nullptr && i == 0;
        if (merge) { next_row->area += finite_or_zero(a_J / T1.area); acc_finite(next_row->fn, wo * (-a_cosv)); }
        else { sink.add_tri(h1.tri, 21, a_J / T1.area); scatter_vec(sink, h1.tri, 18, wo * (-a_cosv)); }
        acc(a_wo, T1.fn * (-a_cosv));
        Vec3f a_wol(0.f);
        brev.eval_vjp(sink, tv0, its, wol, a_fv, va.wi, a_wol, va.u, va.v);
        brev.sampled_pdf_vjp(sink, tv0, its, s, a_pdf_s, va.wi, va.u, va.v);
        acc(a_wo, its.sh.s * a_wol.x + its.sh.t * a_wol.y + its.sh.n * a_wol.z);   // wol = to_local(wo)
        acc(va.s, wo * a_wol.x); acc(va.t, wo * a_wol.y); acc(va.n, wo * a_wol.z);
        Vec3f a_dvec = a_wo / t1;                              // wo = dvec / t1 ; t1 = |dvec|
        a_t1 += -dot(a_wo, wo) / t1;
        acc(a_dvec, wo * a_t1);
        if (merge) acc_finite(next_row->p, a_dvec);
        else scatter_point(sink, h1.tri, h1.u, h1.v, a_dvec);
        acc(va.p, -a_dvec);
    };
    auto light_term = [&](const int i, const float s0, const float s1) {
        (void) i;
        float r0 = s0, r1 = s1;                                // mirrors sample_emitter_position
        int e = 0; float epdf = 1.f;
        if (sc.d.num_emitters > 1) e = sample_reuse(Tab<Sink::flags>::emitter_cmf(sc), Tab<Sink::flags>::emitter_pmf(sc), sc.d.emitter_sum, sc.d.num_emitters, r1, epdf);
        const bool env_s = kEnv && e == sc.d.env_emitter;       // sampled from the environment map: detached, J = 1
        Vec3f psp; float pspdf, ba = 0.f, bb = 0.f, e_area = 1.f; int etri = -1;
        if (env_s) {
            const PosSample<float> pse = env_sample_position<float>(sc, its.p, r0, r1);
            psp = pse.p; pspdf = pse.pdf * epdf;
        } else {
            const float *ef = Tab<Sink::flags>::emitter_f(sc, e);
            const int32_t *ei = Tab<Sink::flags>::emitter_i(sc, e);
            float fp;
            const int fidx = sample_reuse(Tab<Sink::flags>::face_cmf(sc) + ei[3], Tab<Sink::flags>::face_pmf(sc) + ei[3], ef[5], ei[2], r0, fp);
            const float tt = sqrtf(fmaxf(1.f - r0, 0.f));
            ba = 1.f - tt; bb = tt * r1;
            etri = ei[1] + fidx;
            const TriRow<float> Te = load_tri<float>(sc, tv0, etri);
            psp = bary_point(Te.p0, Te.e1, Te.e2, ba, bb);
            pspdf = ef[4] * epdf;
            e_area = Te.area;
        }
        const Vec3f wov = psp - its.p;
        const float d2 = dot(wov, wov), dist = sqrtf(fmaxf(d2, 0.f));
        const Vec3f wo = wov / dist;
        Hit h2;
        // (direct_step: a light sample the BSDF's cosine tests zero is not traced -- recorded as a miss)
        if (PSDR_SKIP_UNLIT && !(REPLAY && BACKWARD) && !(its.sh.to_local(wo).z > 0.f && its.wi.z > 0.f)) { if (REPLAY) rec.put_tri(k, 1, -1); return; }
        if (REPLAY && BACKWARD) h2.tri = rec.tri(k, 1);
        else {
            nrays++;
            h2 = closest_hit<false, tree_mode<Sink::flags>()>(sc, st, its.p, wo, INFINITY, -1, -1, kPreLightRay);
            if (REPLAY) rec.put_tri(k, 1, h2.tri);
        }
        if (h2.tri < 0) return;
        const int tm2 = Tab<Sink::flags>::tri_mesh(sc, h2.tri), mesh2 = tm2 & ~PSDR_TRI_FACE_NORMALS;
        const int e2 = Tab<Sink::flags>::mesh_emitter(sc, mesh2);
        const TriRow<float> T2 = load_tri<float>(sc, tv0, h2.tri);
        if (REPLAY && BACKWARD) h2 = hit_on_triangle(h2.tri, T2.p0, T2.e1, T2.e2, its.p, wo);
        const Vec3f p2 = bary_point(T2.p0, T2.e1, T2.e2, h2.u, h2.v);
        const float t2 = norm(p2 - its.p);
        if (!(t2 > dist - kShadowEpsilon && e2 >= 0)) return;
        const bool env2 = kEnv && e2 == sc.d.env_emitter;
        Vec3f Le2;
        // the forward pass looks the map up along the path-space direction to the HIT point p2 (its1.wi)
        const Vec3f dir2 = (p2 - its.p) / t2;
        if (env2) Le2 = env_eval_direction<float>(sc, tv0, dir2);
        else {
            const ShNormal sn2 = shading_normal(T2, (tm2 & PSDR_TRI_FACE_NORMALS) != 0, h2.u, h2.v);
            if (!(-dot((p2 - its.p) / t2, sn2.n) > 0.f)) return;         // Le = 0 from behind
            const float *rr = Tab<Sink::flags>::emitter_f(sc, e2);
            Le2 = Vec3f{rr[0], rr[1], rr[2]};
        }
        const float cosv = -dot(T2.fn, wo);
        const float G = fabsf(cosv) / d2;
        const Vec3f wl = its.sh.to_local(wo);
        const Vec3f f = bsdf.eval(sc, tv0, its, wl, true);
        const float cfac = G / pspdf;                                   // G * ps.J / ps.pdf, J = 1
        const Vec3f valv = f * cfac;
        const float pdfb = bsdf.pdf(sc, tv0, its, wl, true);
        const float pdf1 = pdfb * G;
        float w = 1.f / (float) nL, dw_dpdf1 = 0.f;
        if (nB > 0) {
            const float a2 = pspdf * pspdf, b2 = pdf1 * pdf1, den = a2 + b2;
            w *= a2 / den;
            dw_dpdf1 = (-2.f * a2 * pdf1 / (den * den)) / (float) nL;
        }
        out.c = out.c + Le2 * valv * w;
        if (!BACKWARD) return;
        const Vec3f gr = a_c * valv * w;
        Vec3f a_wo(0.f);
        if (env2) {
            if constexpr (kEnv) {
                const Vec3f a_dir2 = env_eval_vjp(sink, sc, dir2, gr);
                const Vec3f a_p2 = (a_dir2 - dir2 * dot(dir2, a_dir2)) / t2;      // dir2 = (p2 - p) / |p2 - p|
                scatter_point(sink, h2.tri, h2.u, h2.v, a_p2);
                acc(va.p, -a_p2);
            }
        } else { sink.add_rad(e2, 0, gr.x); sink.add_rad(e2, 1, gr.y); sink.add_rad(e2, 2, gr.z); }
        const Vec3f a_val = a_c * Le2 * w;
        const float a_w = dot(a_c, Le2 * valv);
        const float a_pdfb = a_w * dw_dpdf1 * G;                        // pdf1 = pdfb * detach(G)
        const Vec3f a_fv = a_val * cfac;
        const float a_cfac = dot(a_val, f);
        const float a_G = a_cfac / pspdf;
        const float a_earea = a_cfac * cfac / e_area;                   // ps.J = A_e / detach(A_e)
        const float a_cosv = a_G * (cosv < 0.f ? -1.f : 1.f) / d2;
        float a_d2 = -a_G * G / d2;
        const bool one_row = !env_s && h2.tri == etri;                  // sampled point and hit on the same triangle (the rule): one row update
        if (!one_row) { if (!env_s) sink.add_tri(etri, 21, a_earea); scatter_vec(sink, h2.tri, 18, wo * (-a_cosv)); }
        acc(a_wo, T2.fn * (-a_cosv));
        Vec3f a_wl(0.f);
        brev.eval_vjp(sink, tv0, its, wl, a_fv, va.wi, a_wl, va.u, va.v);
        brev.pdf_vjp(sink, tv0, its, wl, a_pdfb, va.wi, a_wl, va.u, va.v);
        acc(a_wo, its.sh.s * a_wl.x + its.sh.t * a_wl.y + its.sh.n * a_wl.z);
        acc(va.s, wo * a_wl.x); acc(va.t, wo * a_wl.y); acc(va.n, wo * a_wl.z);
        Vec3f a_wov = a_wo / dist;
        const float a_dist = -dot(a_wo, wo) / dist;
        a_d2 += a_dist / (2.f * dist);
        acc(a_wov, wov * (2.f * a_d2));
        if (one_row) scatter_row(sink, etri, ba, bb, a_wov, wo * (-a_cosv), a_earea);
        else if (!env_s) scatter_point(sink, etri, ba, bb, a_wov);
        acc(va.p, -a_wov);
    };
    // geometry-adjoint sweeps only (the material sinks take no rows: their texel kernel at five waves lost 20 % to the order, 2.12 -> 2.60 ms)
    if constexpr (REPLAY && BACKWARD && SinkTakesRows<Sink>::value && (PSDR_REV_LIGHT_FIRST == 1 || (PSDR_REV_LIGHT_FIRST == 2 && (Sink::flags & kSceneRough) != 0))) {
        // one sample of each kind at most (REPLAY): the light sample is differentiated FIRST -- the BSDF sample leaves the next vertex
        // (a whole hit record and its row adjoint) behind, which then is not alive across the light sample's chain.  Same five numbers;
        // the vertex' accumulators start from zero, so their two addends commute.
        float s[3] = {0.f, 0.f, 0.f}, s0 = 0.f, s1 = 0.f;
        if (nB > 0) { s[0] = rng.next(); s[1] = rng.next(); s[2] = rng.next(); }
        if (nL > 0) { s0 = rng.next(); s1 = rng.next(); }
        if (nL > 0) light_term(0, s0, s1);
        if (nB > 0) bsdf_term(0, s);
    } else {
        for (int i = 0; i < nB; ++i) {
            const float s[3] = {rng.next(), rng.next(), rng.next()};
            bsdf_term(i, s);
        }
        for (int i = 0; i < nL; ++i) {
            const float s0 = rng.next(), s1 = rng.next();
            light_term(i, s0, s1);
        }
    }
    return out;
}

template <int FLAGS = kSceneRough> struct NullSink {
    static constexpr int flags = FLAGS;
    static constexpr bool has_env = (FLAGS & kSceneEnv) != 0;
    PSDR_HD void add_env(int, float) {}
    PSDR_HD void add_tri(int, int, float) {}
    PSDR_HD void add_texel(int, float) {}
    PSDR_HD void add_rad(int, int, float) {}
    PSDR_HD void add_cam(int, float) {}
    PSDR_HD void add_sedge(int, int, float) {}
    PSDR_HD void add_pedge(int, int, float) {}
};

// Gradient of the PRIMARY triangle row of one camera sample (p0 e1 e2 n0 n1 n2 fn = words 0..20).
// Lanes of a wave share their pixel, hence almost always their primary triangle: the kernel sums
// these rows across the wave (segmented by triangle id) before touching memory.
constexpr int kPrimaryWords = 21;
struct PrimaryGrad {
    int tri;
    float w[kPrimaryWords];
    PSDR_HD void clear() { tri = -1;
#pragma unroll
        for (int i = 0; i < kPrimaryWords; ++i) w[i] = 0.f; }
};
// Routes add_tri(primary triangle, word < 21) into registers, everything else to the real sink.
template <class Sink> struct PrimarySink {
    static constexpr int flags = Sink::flags;
    static constexpr bool has_env = Sink::has_env;
    Sink &real; PrimaryGrad &pg;
    PSDR_HD void add_env(int w, float v) { real.add_env(w, v); }
    PSDR_HD PrimarySink(Sink &r, PrimaryGrad &p) : real(r), pg(p) {}
    PSDR_HD void add_tri(int tri, int word, float v) {
        if (tri == pg.tri && word < kPrimaryWords) { if (v != 0.f && isfinite(v)) pg.w[word] += v; }
        else real.add_tri(tri, word, v);
    }
    PSDR_HD void add_texel(int i, float v) { real.add_texel(i, v); }
    PSDR_HD void add_rad(int e, int c, float v) { real.add_rad(e, c, v); }
    PSDR_HD void add_cam(int w, float v) { real.add_cam(w, v); }
    PSDR_HD void add_sedge(int e, int w, float v) { real.add_sedge(e, w, v); }
    PSDR_HD void add_pedge(int e, int w, float v) { real.add_pedge(e, w, v); }
};

// Drops every geometry adjoint AT COMPILE TIME (material-only gradients: texels, emitter radiance, the
// environment-map record): with these empty members the whole geometric adjoint chain of the estimators
// -- frames, Moeller-Trumbore, path-space directions, the LDS adds of triangle rows -- is dead code.
template <class Sink> struct MaterialSink {
    static constexpr int flags = Sink::flags;
    static constexpr bool has_env = Sink::has_env;
    Sink &real;
    PSDR_HD MaterialSink(Sink &r, PrimaryGrad &) : real(r) {}
    PSDR_HD void add_tri(int, int, float) {}
    PSDR_HD void add_cam(int, float) {}
    PSDR_HD void add_sedge(int, int, float) {}
    PSDR_HD void add_pedge(int, int, float) {}
    PSDR_HD void add_texel(int i, float v) { real.add_texel(i, v); }
    PSDR_HD void add_rad(int e, int c, float v) { real.add_rad(e, c, v); }
    PSDR_HD void add_env(int w, float v) { real.add_env(w, v); }
};
template <bool GEO, class RealSink> struct CameraSinkOf { using type = PrimarySink<RealSink>; };
template <class RealSink> struct CameraSinkOf<false, RealSink> { using type = MaterialSink<RealSink>; };

// Back-propagates the adjoints of a PATH-SPACE vertex (k >= 1) into its triangle row and returns the
// adjoint of the previous vertex' position (wi_k = to_local_k(-(p_k - p_{k-1}) / t)).
template <class Sink> PSDR_HD Vec3f path_vertex_backward(Sink &sink, const SceneView &sc, const Its<float> &v, const Vec3f &prev_p, VertexAdj va, RowAdj &row) {
    const TangentView<0, Sink::flags> tv0{};
    const TriRow<float> T = load_tri<float>(sc, tv0, v.tri);
    const bool face = (Tab<Sink::flags>::tri_mesh(sc, v.tri) & PSDR_TRI_FACE_NORMALS) != 0;
    const ShNormal sn = shading_normal(T, face, v.hu, v.hv);
    Vec3f dir = v.p - prev_p;
    const float t = norm(dir);
    dir = dir / t;
    const Vec3f a_dir = -(v.sh.s * va.wi.x + v.sh.t * va.wi.y + v.sh.n * va.wi.z);
    acc(va.s, dir * (-va.wi.x)); acc(va.t, dir * (-va.wi.y)); acc(va.n, dir * (-va.wi.z));
    const Vec3f a_shn = va.n + frame_vjp(sn.n, va.s, va.t);
    float abu = 0.f, abv = 0.f;
    if (face) acc_finite(row.fn, a_shn);
    else shading_normal_vjp(sink, v.tri, T, sn, v.hu, v.hv, a_shn, abu, abv);   // barycentrics detached: abu/abv dropped
    const Vec3f a_dvec = (a_dir - dir * dot(dir, a_dir)) / t;
    acc_finite(row.p, va.p + a_dvec);
    return -a_dvec;
}

// One camera sample in reverse mode (Integrator::__render<true> + enoki.backward).
//   adj = dLoss/d(pixel) / spp.   Returns the primal sample value.
//   GEO: a geometry gradient (triangle table / camera) is wanted -> solid-angle form of the primary hit and
//        the geometric adjoint chain; otherwise the on-surface form, exactly like forward mode
//        (psdr_device.h Li), and every geometry adjoint is compiled out (MaterialSink)
//   INTEG >= 0: integrator fixed at compile time (as in the forward kernels); -1: run-time switch
//   STAGE 0: both sweeps in one kernel.  1: the value sweep only, record written to `disk`.  2: the adjoint sweep only, from `disk`.
template <bool GEO, int INTEG = -1, int STAGE = 0, class RealSink>
PSDR_HD Vec3f camera_sample_reverse(RealSink &real_sink, PrimaryGrad &pg, PathRec &rec, const SceneView &sc, TraversalStack &st, const LiParams &lp,
                                    const RngJump &jump, int pixel, uint64_t slot, const Vec3f &adj, uint32_t &nrays, const RevDisk &disk = RevDisk{nullptr, 0}) {
    constexpr bool geo = GEO;
    pg.clear();
    using Sink = typename CameraSinkOf<GEO, RealSink>::type;
    Sink sink(real_sink, pg);
    // the two sweeps scatter straight into the real sink; only the adjoint chain of the PRIMARY vertex (run last,
    // all on one triangle row that the lanes of a wave share) goes through the register accumulators of
    // PrimarySink -- so those 21 registers are not live across the sweeps
    auto &sweep = [&]() -> auto & { if constexpr (GEO) return real_sink; else return sink; }();
    using SweepSink = std::remove_reference_t<decltype(sweep)>;
    const TangentView<0, Sink::flags> tv0{};
    Rng rng; rng.init(slot, jump);
    const float j0 = rng.next(), j1 = rng.next();
    const int W = sc.d.width;
    const float sx = ((float) (pixel % W) + j0) / (float) W, sy = ((float) (pixel / W) + j1) / (float) sc.d.height;
    const Vec3f dcam = camera_space_dir(sc, sx, sy);
    const RayT<float> ray = primary_ray<float>(sc, tv0, sx, sy);
    Hit h0;
    int nv = 0;
    if constexpr (STAGE == 2) {
        h0.tri = disk.geti(0);
        if (h0.tri < 0) return Vec3f(0.f);
        nv = disk.geti(1);
    } else {
        nrays++;
        h0 = closest_hit<false, tree_mode<RealSink::flags>()>(sc, st, ray.o, ray.d, INFINITY, -1, -1, kPrePrimaryRay);
        if (h0.tri < 0) { if constexpr (STAGE == 1) disk.puti(0, -1); return Vec3f(0.f); }
    }
    pg.tri = h0.tri;
    const int tm0 = Tab<RealSink::flags>::tri_mesh(sc, h0.tri);
    const bool face0 = (tm0 & PSDR_TRI_FACE_NORMALS) != 0;
    const TriRow<float> T0 = load_tri<float>(sc, tv0, h0.tri);
    if constexpr (STAGE == 2) h0 = hit_on_triangle(h0.tri, T0.p0, T0.e1, T0.e2, ray.o, ray.d);       // the leaf test's arithmetic: the same (u, v)
    // solid-angle form (scene.cpp:355-376)
    float bu, bv, t0;
    Its<float> its;
    if (geo) {
        moeller_trumbore(T0.p0, T0.e1, T0.e2, ray, bu, bv, t0);
        its.p = bary_point(T0.p0, T0.e1, T0.e2, bu, bv);        // = ray(t0), evaluated on the triangle (psdr_device.h intersect)
    } else {
        bu = h0.u; bv = h0.v;
        its.p = bary_point(T0.p0, T0.e1, T0.e2, bu, bv);
        t0 = norm(its.p - ray.o);
    }
    its.valid = true; its.tri = h0.tri; its.mesh = tm0 & ~PSDR_TRI_FACE_NORMALS; its.hu = h0.u; its.hv = h0.v;
    its.n = T0.fn; its.J = 1.f; its.t = t0;
    const ShNormal sn0 = shading_normal(T0, face0, bu, bv);
    its.sh = Frame<float>(sn0.n);
    its.wi = its.sh.to_local(-ray.d);
    const float *q = sc.d.tri_uv ? Tab<RealSink::flags>::tri_uv(sc, h0.tri) : nullptr;
    its.uvx = q ? (q[2] - q[0]) * bu + ((q[4] - q[0]) * bv + q[0]) : 0.f;
    its.uvy = q ? (q[3] - q[1]) * bu + ((q[5] - q[1]) * bv + q[1]) : 0.f;

    PSDR_CLK_MARK_ST(st, 0);                    // record head + the primary vertex rebuilt
    VertexAdj va0; va0.clear();                 // adjoints of the primary vertex; its solid-angle chain runs last
    Vec3f result(0.f);
    const int integ = INTEG >= 0 ? INTEG : lp.integrator;
    if (integ == PSDR_INTEGRATOR_FIELD) {
        // FieldExtractionIntegrator (field.cpp:34-54): position / depth / geoNormal carry derivatives
        float a_t = 0.f;
        switch (lp.field) {
            case PSDR_FIELD_SILHOUETTE: result = Vec3f(1.f); break;
            case PSDR_FIELD_POSITION: result = its.p; va0.p = adj; break;
            case PSDR_FIELD_DEPTH: result = Vec3f(t0); a_t = adj.x + adj.y + adj.z; break;
            case PSDR_FIELD_GEONORMAL: result = its.n; scatter_vec(sink, h0.tri, 18, adj); break;
            case PSDR_FIELD_SHNORMAL: result = its.sh.n; va0.n = adj; break;
            default: result = Vec3f{its.uvx, its.uvy, 0.f}; va0.u = adj.x; va0.v = adj.y; break;
        }
        if (!(isfinite(result.x) && isfinite(result.y) && isfinite(result.z))) return zero_nonfinite(result);
        if (!geo) return result;
        // p = p0 + bu e1 + bv e2 with (bu, bv, t) = MT(tri0, ray)  (depth == t)
        const Vec3f a_shn = va0.n + frame_vjp(sn0.n, va0.s, va0.t);
        float abu = dot(va0.p, T0.e1), abv = dot(va0.p, T0.e2);
        shading_normal_vjp(sink, h0.tri, T0, sn0, bu, bv, a_shn, abu, abv);
        if (q) { abu += va0.u * (q[2] - q[0]) + va0.v * (q[3] - q[1]); abv += va0.u * (q[4] - q[0]) + va0.v * (q[5] - q[1]); }
        const MtAdj ma = mt_vjp(T0.p0, T0.e1, T0.e2, ray, abu, abv, a_t);
        scatter_vec(sink, h0.tri, 0, ma.p0 + va0.p); scatter_vec(sink, h0.tri, 3, ma.e1 + va0.p * bu); scatter_vec(sink, h0.tri, 6, ma.e2 + va0.p * bv);
        camera_ray_vjp(sink, sc, dcam, ma.o, ma.d);
        return result;
    }

    const bool direct = integ == PSDR_INTEGRATOR_DIRECT;
    const int nB = direct ? lp.bsdf_samples : 1, nL = direct ? lp.light_samples : 1;
    const int depth = direct ? 1 : (lp.max_depth < kMaxRevDepthDeep ? lp.max_depth : kMaxRevDepthDeep);
    // sweep 2 replays the hits of sweep 1 (always for the PathTracer; DirectIntegrator with <= 1 sample of each kind)
    const bool replay = STAGE != 0 || INTEG == PSDR_INTEGRATOR_PATH || (nB <= 1 && nL <= 1);      // a split launch is only made when the hits can be replayed

    const int e0 = Tab<RealSink::flags>::mesh_emitter(sc, its.mesh);
    const bool env0 = Sink::has_env && !lp.hide_emitters && e0 >= 0 && e0 == sc.d.env_emitter;
    const bool le0 = !lp.hide_emitters && e0 >= 0 && !env0 && its.wi.z > 0.f;
    if (le0) { const float *r = Tab<RealSink::flags>::emitter_f(sc, e0); result = Vec3f{r[0], r[1], r[2]}; }
    if (env0) result = env_eval_direction<float>(sc, tv0, ray.d);

    // ---- sweep 1 (values): record (c_k, f_k), build the suffix radiances T_k
    if constexpr (STAGE != 2) {
        NullSink<Sink::flags> ns; VertexAdj dummy; dummy.clear();
        Rng r1 = rng;
        Its<float> cur = its;
        Vec3f beta(1.f);
        for (int k = 0; k < depth; ++k) {
            const bool eo = direct || k + 1 >= depth;          // nothing continues from this vertex' BSDF sample: its hit matters only on an emitter
            const VertexOut vo = replay ? vertex_eval<false, true, NullSink<Sink::flags>>(ns, sc, st, r1, cur, nB, nL, Vec3f(0.f), Vec3f(0.f), dummy, nrays, rec, k, nullptr, eo)
                                        : vertex_eval<false, false, NullSink<Sink::flags>>(ns, sc, st, r1, cur, nB, nL, Vec3f(0.f), Vec3f(0.f), dummy, nrays, rec, k, nullptr, eo);
            rec.put_cf(k, vo.c, vo.f); nv = k + 1;
            result = result + beta * vo.c;
            if (!vo.next_valid) break;
            beta = beta * vo.f; cur = vo.next;
            if (!(beta.x != 0.f || beta.y != 0.f || beta.z != 0.f)) break;
        }
    }
    // masked(value, ~isfinite(value)) = 0 (integrator.cpp:87): a zeroed sample has no gradient either
    if constexpr (STAGE != 2) {
        if (!(isfinite(result.x) && isfinite(result.y) && isfinite(result.z))) { if constexpr (STAGE == 1) disk.puti(0, -1); return zero_nonfinite(result); }
    }
    Vec3f a_d_le0(0.f);                         // d Le(primary) / d ray direction (environment map seen directly)
    if constexpr (STAGE != 1) {
        if (le0) { sweep.add_rad(e0, 0, adj.x); sweep.add_rad(e0, 1, adj.y); sweep.add_rad(e0, 2, adj.z); }
        if (env0) { if constexpr (Sink::has_env) a_d_le0 = env_eval_vjp(sweep, sc, ray.d, adj); }
    }
    // suffix radiances T_{k+1} overwrite c_k in place (T_nv = 0): afterwards rec.c(k) == T_{k+1}
    if constexpr (STAGE != 2) {
        Vec3f T(0.f);
        for (int k = nv - 1; k >= 0; --k) { const Vec3f Tk = rec.c(k) + rec.f(k) * T; rec.put(k, 0, T.x); rec.put(k, 1, T.y); rec.put(k, 2, T.z); T = Tk; }
    }
    if constexpr (STAGE == 1) {
        disk.puti(0, h0.tri); disk.puti(1, nv);
        for (int k = 0; k < nv; ++k) {
            const int w = kRevDiskHead + k * kRevDiskPerVertex;
            disk.put(w, rec.get(k, 0)); disk.put(w + 1, rec.get(k, 1)); disk.put(w + 2, rec.get(k, 2));
            disk.puti(w + 3, rec.tri(k, 0)); disk.puti(w + 4, rec.tri(k, 1));
        }
        return result;
    }
    if constexpr (STAGE == 2) {
        if (disk.cf) {
            // records of the wavefront value sweep: (c_k, f_k) per vertex -> suffix radiances T_{k+1} in place of c_k, as sweep 1 leaves them
            Vec3f T(0.f);
            for (int k = nv - 1; k >= 0; --k) {
                const int w = kRevDiskHead + k * kRevDiskPerVertexCf;
                const Vec3f ck{disk.get(w), disk.get(w + 1), disk.get(w + 2)}, fk{disk.get(w + 3), disk.get(w + 4), disk.get(w + 5)};
                rec.put(k, 0, T.x); rec.put(k, 1, T.y); rec.put(k, 2, T.z);
                rec.put_tri(k, 0, disk.geti(w + 6)); rec.put_tri(k, 1, disk.geti(w + 7));
                T = ck + fk * T;
            }
        } else
        for (int k = 0; k < nv; ++k) {
            const int w = kRevDiskHead + k * kRevDiskPerVertex;
            rec.put(k, 0, disk.get(w)); rec.put(k, 1, disk.get(w + 1)); rec.put(k, 2, disk.get(w + 2));
            rec.put_tri(k, 0, disk.geti(w + 3)); rec.put_tri(k, 1, disk.geti(w + 4));
        }
    }

    PSDR_CLK_MARK_ST(st, 1);                    // the path record read, suffix radiances
    // ---- sweep 2: replay the same random numbers, differentiate vertex by vertex
    {
        Its<float> cur = its;
        Vec3f beta(1.f);
        Vec3f prev_p = its.p;        // vertex k-1: position and where its pending row adjoint goes
        int prev_tri = -1; float prev_u = 0.f, prev_v = 0.f;
        RowAdj pend; pend.clear();            // row adjoint of vertex k-1, complete but for the position adjoint the direction chain of vertex k sends back
        RowAdj row_cur; row_cur.clear();      // row of vertex k, fed by the BSDF sample of iteration k-1
        for (int k = 0; k < nv; ++k) {
            const Vec3f a_c = adj * beta;
            const Vec3f a_f = (k + 1 < nv) ? a_c * rec.c(k) : Vec3f(0.f);
            VertexAdj va; va.clear();
            RowAdj row_next; row_next.clear();
            const bool eo = direct || k + 1 >= depth;
            const VertexOut vo = replay ? vertex_eval<true, true, SweepSink>(sweep, sc, st, rng, cur, nB, nL, a_c, a_f, k == 0 ? va0 : va, nrays, rec, k, &row_next, eo)
                                        : vertex_eval<true, false, SweepSink>(sweep, sc, st, rng, cur, nB, nL, a_c, a_f, k == 0 ? va0 : va, nrays, rec, k, &row_next, eo);
            PSDR_CLK_MARK_ST(st, 2 + (k > 0 ? 1 : 0));      // vertex_eval backward: vertex 0 / the others
            if (k >= 1) {
                // a vertex' row leaves ONCE and COMPLETE (position, face normal, area: complete_row) -- one iteration late, when its successor's direction
                // chain has sent its position adjoint back; a sink that defers rows (DeviceSink) adds them sorted by row at the kernel's convergent point
                const Vec3f a_prev = path_vertex_backward(sweep, sc, cur, prev_p, va, row_cur);
                if (k == 1) acc(va0.p, a_prev);
                else { acc_finite(pend.p, a_prev); complete_row(sweep, prev_tri, prev_u, prev_v, pend); }
                pend = row_cur;
            }
            PSDR_CLK_MARK_ST(st, 4);            // the direction chain back to the vertex behind, its row parked
            if (!vo.next_valid || k + 1 >= nv) {
                if (k >= 1) complete_row(sweep, cur.tri, cur.hu, cur.hv, pend);
                if (vo.next_valid) complete_row(sweep, vo.next.tri, vo.next.hu, vo.next.hv, row_next);
                break;
            }
            beta = beta * vo.f;
            prev_p = cur.p; prev_tri = cur.tri; prev_u = cur.hu; prev_v = cur.hv;
            cur = vo.next; row_cur = row_next;
        }
    }
    // ---- primary vertex: wi = to_local(-d), frame(sh_n(bu,bv)), uv(bu,bv), p = p0 + bu e1 + bv e2, (bu,bv,t) = MT(tri0, ray)
    if (geo) {
        // The primary vertex is REBUILT here (ray, triangle row, Moeller-Trumbore, shading frame: ~150 VALU) instead
        // of staying live across the two sweeps: ~60 registers less in a kernel that spills at 2 waves / SIMD.
        // The triangle index goes through an opaque copy so that the compiler does not merge the two evaluations.
        int tri_b = h0.tri;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(tri_b));
#endif
        const Vec3f dcam_b = camera_space_dir(sc, sx, sy);
        const RayT<float> ray_b = primary_ray<float>(sc, tv0, sx, sy);
        const TriRow<float> Tb = load_tri<float>(sc, tv0, tri_b);
        float bu_b, bv_b, t_b;
        moeller_trumbore(Tb.p0, Tb.e1, Tb.e2, ray_b, bu_b, bv_b, t_b);
        const ShNormal sn_b = shading_normal(Tb, face0, bu_b, bv_b);
        const Frame<float> sh_b(sn_b.n);
        const float *qb = sc.d.tri_uv ? Tab<RealSink::flags>::tri_uv(sc, tri_b) : nullptr;
        Vec3f a_d = a_d_le0 - (sh_b.s * va0.wi.x + sh_b.t * va0.wi.y + sh_b.n * va0.wi.z);
        acc(va0.s, ray_b.d * (-va0.wi.x)); acc(va0.t, ray_b.d * (-va0.wi.y)); acc(va0.n, ray_b.d * (-va0.wi.z));
        const Vec3f a_shn = va0.n + frame_vjp(sn_b.n, va0.s, va0.t);
        float abu = dot(va0.p, Tb.e1), abv = dot(va0.p, Tb.e2);
        shading_normal_vjp(sink, tri_b, Tb, sn_b, bu_b, bv_b, a_shn, abu, abv);
        if (qb) { abu += va0.u * (qb[2] - qb[0]) + va0.v * (qb[3] - qb[1]); abv += va0.u * (qb[4] - qb[0]) + va0.v * (qb[5] - qb[1]); }
        const MtAdj ma = mt_vjp(Tb.p0, Tb.e1, Tb.e2, ray_b, abu, abv, 0.f);
        scatter_vec(sink, tri_b, 0, ma.p0 + va0.p); scatter_vec(sink, tri_b, 3, ma.e1 + va0.p * bu_b); scatter_vec(sink, tri_b, 6, ma.e2 + va0.p * bv_b);
        camera_ray_vjp(sink, sc, dcam_b, ma.o, a_d + ma.d);
    }
    PSDR_CLK_MARK_ST(st, 5);                    // the primary vertex' chain
    return result;
}

// One primary-edge slot in reverse mode (integrator.cpp:98-119): value = x_dot_n * dL / pdf / sppe with
// x_dot_n = dot(lerp(p0, p1, u), n) the only differentiable factor -> gradient w.r.t. the edge table.
// Returns the edge (or -1) and w[4] = the gradient of its (p0.x, p0.y, p1.x, p1.y) words, so that a kernel can
// combine the lanes of a wave that landed on the same edge before touching memory.
template <int INTEG, int FL>
PSDR_HD int primary_edge_reverse_values(const SceneView &sc, TraversalStack &st, const LiParams &lp, const RngJump &jump, uint64_t slot,
                                        float inv_sppe, const float *__restrict__ adj_img, uint32_t &nrays, float w[4]) {
    w[0] = w[1] = w[2] = w[3] = 0.f;
    Rng rng; rng.init(slot, jump);
    float u = rng.next(), pmf;
    const int k = sample_reuse(sc.d.prim_cmf, sc.d.prim_pmf, sc.d.prim_sum, sc.d.num_prim_edges, u, pmf);
    const float *pe = sc.d.prim_edge + (size_t) k * PSDR_PEDGE_STRIDE;
    const float nx = pe[4], ny = pe[5], pdf = pmf / pe[6];
    const float px = pe[0] * (1.f - u) + pe[2] * u, py = pe[1] * (1.f - u) + pe[3] * u;
    const int W = sc.d.width, H = sc.d.height;
    const int ix = (int) floorf(px * (float) W), iy = (int) floorf(py * (float) H);
    bool valid = ix >= 0 && ix < W && iy >= 0 && iy < H && pmf > 0.f && pe[6] > 0.f;          // (pmf > 0, length > 0: see primary_edge_sample)
    const TangentView<0, FL> tv0{};
    if (sc.d.prim_edge_z != nullptr && valid) valid = primary_edge_point_visible(sc, tv0, st, k, u, px, py, nrays);
    Vec3f L2[2];
    if constexpr (pair_walk_ok<FL>()) {
        // two-level scenes: both camera rays through ONE walk (closest_hit_pair), then the estimator on each side's known hit
        Hit hp0, hp1;
        hp0.tri = hp1.tri = -1; hp0.u = hp0.v = hp0.t = hp1.u = hp1.v = hp1.t = 0.f;
        if (valid) primary_edge_camera_hits(sc, tv0, st, px, py, nx, ny, hp0, hp1);
#pragma unroll 1
        for (int side = 0; side < 2; ++side) {
            const float sg = side == 0 ? -kEdgeEpsilon : kEdgeEpsilon;
            const RayT<float> ray = primary_ray<float>(sc, tv0, px + sg * nx, py + sg * ny);
            const Hit hs = side == 0 ? hp0 : hp1;
            L2[side] = Li<float, float, INTEG, true>(sc, tv0, st, lp, rng, ray, valid, nrays, &hs);
        }
    } else {
#pragma unroll 1
        for (int side = 0; side < 2; ++side) {
            const float sg = side == 0 ? -kEdgeEpsilon : kEdgeEpsilon;
            const RayT<float> ray = primary_ray<float>(sc, tv0, px + sg * nx, py + sg * ny);
            L2[side] = Li<float, float, INTEG>(sc, tv0, st, lp, rng, ray, valid, nrays);
        }
    }
    if (!valid) return -1;
    const Vec3f Ln = L2[0], Lp = L2[1];
    const float *a = adj_img + (size_t) (iy * W + ix) * 3;
    const float xdn = px * nx + py * ny;
    float g = 0.f;
    const float dL[3] = {(Ln.x - Lp.x) / pdf, (Ln.y - Lp.y) / pdf, (Ln.z - Lp.z) / pdf};
#pragma unroll
    for (int c = 0; c < 3; ++c) if (isfinite(xdn * dL[c])) g += a[c] * dL[c];
    g *= inv_sppe;
    if (g == 0.f || !isfinite(g)) return -1;
    w[0] = g * (1.f - u) * nx; w[1] = g * (1.f - u) * ny; w[2] = g * u * nx; w[3] = g * u * ny;
    return k;
}
template <int INTEG = -1, class Sink>
PSDR_HD void primary_edge_reverse(Sink &sink, const SceneView &sc, TraversalStack &st, const LiParams &lp, const RngJump &jump, uint64_t slot,
                                  float inv_sppe, const float *__restrict__ adj_img, uint32_t &nrays) {
    float w[4];
    const int k = primary_edge_reverse_values<INTEG, Sink::flags>(sc, st, lp, jump, slot, inv_sppe, adj_img, nrays, w);
    if (k < 0) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) sink.add_pedge(k, i, w[i]);
}

// One secondary-edge slot in reverse mode (direct.cpp:225-316): result = value0 * dot(n, u2(theta)).
template <class Sink>
PSDR_HD void secondary_edge_reverse(Sink &sink, const SceneView &sc, TraversalStack &st, const float s3[3], float scale,
                                    const float *__restrict__ adj_img, uint32_t &nrays, bool count_first = true) {
    const TangentView<0, Sink::flags> tv0{};
    float s1 = s3[0], pdf0;
    const int k = sample_reuse(sc.d.sec_cmf, sc.d.sec_pmf, sc.d.sec_sum, sc.d.num_sec_edges, s1, pdf0);
    const float *se = sc.d.sec_edge + (size_t) k * PSDR_SEDGE_STRIDE;
    const Vec3f ep0{se[0], se[1], se[2]}, ee1{se[3], se[4], se[5]}, n0{se[6], se[7], se[8]}, n1{se[9], se[10], se[11]}, ep2{se[12], se[13], se[14]};
    const bool is_boundary = se[15] != 0.f;
    const Vec3f p0 = ee1 * s1 + ep0;
    const float e1len = norm(ee1);
    const Vec3f edge = ee1 / e1len, edge2 = ep2 - ep0;
    pdf0 /= e1len;
    const PosSample<float> ps2 = sample_emitter_position<float>(sc, tv0, p0, s3[1], s3[2], false);
    const Vec3f p2 = ps2.p, bn = ps2.n;
    Vec3f e = p2 - p0;
    const float distSqr = dot(e, e);
    e = e / sqrtf(fmaxf(distSqr, 0.f));
    const float cosTheta = -dot(bn, e);
    const float d0n = dot(n0, e), d1n = dot(n1, e);
    const int sgn0 = d0n > kEdgeEpsilon ? 1 : (d0n < -kEdgeEpsilon ? -1 : 0), sgn1 = d1n > kEdgeEpsilon ? 1 : (d1n < -kEdgeEpsilon ? -1 : 0);
    bool valid = cosTheta > kEpsilon && (is_boundary ? sgn0 != 0 : sgn0 * sgn1 < 0);
    const float bpdf = pdf0 * ps2.pdf * (distSqr / cosTheta);
    const Vec3f dir = normalize(p2 - p0);
    const int f0 = sc.d.sec_edge_faces ? sc.d.sec_edge_faces[2 * k] : -1, f1 = sc.d.sec_edge_faces ? sc.d.sec_edge_faces[2 * k + 1] : -1;
    uint32_t counted_before = 0;                 // split launch: secondary_edge_survives already traced (and counted) these two
    uint32_t &n12 = count_first ? nrays : counted_before;
    const Its<float> its2 = intersect<float>(sc, tv0, st, RayT<float>{p0, dir}, valid, kDetached, n12, f0, f1);
    valid = valid && its2.valid && norm(its2.p - p2) < kShadowEpsilon;
    const Its<float> its1c = intersect<float>(sc, tv0, st, RayT<float>{p0, -dir}, valid, kDetached, n12, f0, f1);
    if (!(valid && its1c.valid)) return;
    const Vec3f p1 = its1c.p;
    int pixel; float qx, qy, sensor_val;
    if (!sample_direct(sc, p1, pixel, qx, qy, sensor_val)) return;
    const Vec3f dcam = camera_space_dir(sc, qx, qy);
    const RayT<float> cam = primary_ray<float>(sc, tv0, qx, qy);
    nrays++;
    const Hit hc = closest_hit<false, tree_mode<Sink::flags>()>(sc, st, cam.o, cam.d, INFINITY);
    if (hc.tri < 0) return;
    const TriRow<float> Tc = load_tri<float>(sc, tv0, hc.tri);
    float cu, cv, ct;
    moeller_trumbore(Tc.p0, Tc.e1, Tc.e2, cam, cu, cv, ct);
    const Vec3f x1 = bary_point(Tc.p0, Tc.e1, Tc.e2, cu, cv);            // its1.p (solid-angle form, on the triangle)
    if (!(camera_return_distance(sc, its1c.tri, its1c.hu, its1c.hv, hc.tri) < (double) kShadowEpsilon)) return;        // psdr_device.h secondary_edge_sample
    const float dist = norm(p2 - p1), cos2 = fabsf(dot(bn, dir));
    const Vec3f ev = cross(edge, dir);
    const float sinphi = norm(ev);
    const Vec3f proj = normalize(cross(ev, bn));
    const float sinphi2 = norm(cross(dir, proj));
    if (!(sinphi > kEpsilon && sinphi2 > kEpsilon)) return;
    const float base_v = (its1c.t / dist) * (sinphi / sinphi2) * cos2;
    const Vec3f d0 = -cam.d;
    const Vec3f d0_local = its1c.sh.to_local(d0);
    if (Tab<Sink::flags>::mesh_bsdf(sc, its1c.mesh) < 0) return;
    const Bsdf<float, float> bsdf(sc, tv0, Tab<Sink::flags>::mesh_bsdf(sc, its1c.mesh));
    Vec3f bsdf_val = bsdf.eval(sc, tv0, its1c, d0_local, true);
    bsdf_val = bsdf_val * fabsf((its1c.wi.z * dot(d0, its1c.n)) / (d0_local.z * dot(dir, its1c.n)));
    Vec3f value0 = bsdf_val * Le<float>(sc, tv0, its2, true) * (base_v * sensor_val / bpdf);
    const Vec3f n = normalize(cross(bn, proj));
    value0 = value0 * (copysignf(1.f, dot(ev, edge2)) * copysignf(1.f, dot(ev, n)));
    const TriRow<float> TA = load_tri<float>(sc, tv0, its2.tri);
    const Vec3f wv = p0 - x1;
    const RayT<float> shadow{x1, normalize(wv)};
    // adjoint seed: a_dn = <adj_pixel, value0> * scale (non-finite components of the value are masked)
    float su, sv, stt;
    moeller_trumbore(TA.p0, TA.e1, TA.e2, shadow, su, sv, stt);
    const Vec3f u2 = bary_point(TA.p0, TA.e1, TA.e2, su, sv);
    const float dn = dot(n, u2);
    const float *a = adj_img + (size_t) pixel * 3;
    float a_dn = 0.f;
    const float v0[3] = {value0.x, value0.y, value0.z};
#pragma unroll
    for (int c = 0; c < 3; ++c) if (isfinite(v0[c] * dn)) a_dn += a[c] * v0[c];
    a_dn *= scale;
    if (a_dn == 0.f || !isfinite(a_dn)) return;
    const Vec3f a_u2 = n * a_dn;
    const MtAdj ma = mt_vjp(TA.p0, TA.e1, TA.e2, shadow, dot(a_u2, TA.e1), dot(a_u2, TA.e2), 0.f);
    scatter_vec(sink, its2.tri, 0, ma.p0); scatter_vec(sink, its2.tri, 3, ma.e1); scatter_vec(sink, its2.tri, 6, ma.e2);
    const Vec3f a_w = normalize_vjp(wv, shadow.d, ma.d);
    // bp0 = e1 * s1 + p0 (edge table)
    sink.add_sedge(k, 0, a_w.x); sink.add_sedge(k, 1, a_w.y); sink.add_sedge(k, 2, a_w.z);
    sink.add_sedge(k, 3, a_w.x * s1); sink.add_sedge(k, 4, a_w.y * s1); sink.add_sedge(k, 5, a_w.z * s1);
    const Vec3f a_x1 = ma.o - a_w;
    // x1 = p0 + u e1 + v e2 with (u, v, .) = MT(triangle C, camera ray)
    const MtAdj mc = mt_vjp(Tc.p0, Tc.e1, Tc.e2, cam, dot(a_x1, Tc.e1), dot(a_x1, Tc.e2), 0.f);
    scatter_vec(sink, hc.tri, 0, mc.p0 + a_x1); scatter_vec(sink, hc.tri, 3, mc.e1 + a_x1 * cu); scatter_vec(sink, hc.tri, 6, mc.e2 + a_x1 * cv);
    camera_ray_vjp(sink, sc, dcam, mc.o, mc.d);
}

}  // namespace psdr
