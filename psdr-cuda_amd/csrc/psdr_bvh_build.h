// psdr_bvh_build.h -- host-side BVH builder shared by the HIP library and the host test harness.
#pragma once
#include "psdr_device.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace psdr {

// -------------------------------------------------------------------- tiny-scene primitives
// The triangles of a tiny scene travel in the kernel arguments (SceneView::tiny) and every ray tests all of them.  Two
// triangles that form a PARALLELOGRAM -- a wall of the Cornell box, fan-triangulated (a, b, c), (a, c, d) -- are tested as
// ONE primitive: Moeller-Trumbore on (p0, e1, e2) = (b, a - b, c - b), the corner that belongs to one triangle only and
// the edges to the shared diagonal, yields the plane coordinates (s, t) of the hit; 0 <= s, t <= 1 is the parallelogram,
// s + t <= 1 the first triangle, otherwise the second; the barycentrics of either triangle are an affine map of (s, t)
// with coefficients in {-1, 0, 1} (a lone triangle is a primitive whose map is the identity: bit-identical to the leaf
// test).  12 wall triangles -> 6 tests on the Cornell box.  Pairs are formed only when all four corners are exact in
// fp32 (p0 + e1, p0 + e2, p0 + e1 + e2 reproduce the stored vertices).
//   prim = (p0 | idA + (idB << 16), idB = 0xffff: none), (e1 | codeA), (e2 | codeB)
//   u = k0 + k1 s + k2 t,  v = k3 + k4 s + k5 t,  code = sum (k_i + 2) << 3 i
inline void pack_tiny_prims(const std::vector<float4> &tris, std::vector<float4> &prims) {
    const int n = (int) tris.size() / 3;
    std::vector<char> used((size_t) n, 0);
    struct V { float x[3]; bool operator==(const V &o) const { return x[0] == o.x[0] && x[1] == o.x[1] && x[2] == o.x[2]; } };
    auto vtx = [&](int t, int k) {
        const float4 &a = tris[(size_t) t * 3], &b = tris[(size_t) t * 3 + 1], &c = tris[(size_t) t * 3 + 2];
        V v{{a.x, a.y, a.z}};
        if (k == 1) { v.x[0] += b.x; v.x[1] += b.y; v.x[2] += b.z; }
        if (k == 2) { v.x[0] += c.x; v.x[1] += c.y; v.x[2] += c.z; }
        return v;
    };
    auto tri_id = [&](int t) { int32_t id; std::memcpy(&id, &tris[(size_t) t * 3].w, 4); return id; };
    // barycentric map of a triangle whose vertices sit at the plane coordinates c[0..2] (each 0 = (0,0), 1 = (1,0), 2 = (0,1), 3 = (1,1))
    auto map_code = [](const int c[3], int32_t &code) {
        static const int cs[4] = {0, 1, 0, 1}, ct[4] = {0, 0, 1, 1};
        const int a = cs[c[1]] - cs[c[0]], b = cs[c[2]] - cs[c[0]], cc = ct[c[1]] - ct[c[0]], d = ct[c[2]] - ct[c[0]];
        const int det = a * d - b * cc;
        if (det != 1 && det != -1) return false;
        const int m00 = d * det, m01 = -b * det, m10 = -cc * det, m11 = a * det;          // inverse of [[a, b], [cc, d]]
        const int k[6] = {-(m00 * cs[c[0]] + m01 * ct[c[0]]), m00, m01, -(m10 * cs[c[0]] + m11 * ct[c[0]]), m10, m11};
        code = 0;
        for (int q = 0; q < 6; ++q) code |= (k[q] + 2) << (3 * q);
        return true;
    };
    prims.clear();
    for (int i = 0; i < n; ++i) {
        if (used[i]) continue;
        used[i] = 1;
        bool paired = false;
        for (int j = i + 1; j < n && !paired; ++j) {
            if (used[j]) continue;
            for (int k0 = 0; k0 < 3 && !paired; ++k0) {          // k0: the vertex of i that is NOT on the shared diagonal
                const V p0 = vtx(i, k0), s1 = vtx(i, (k0 + 1) % 3), s2 = vtx(i, (k0 + 2) % 3);
                V e1, e2, c1, c2, c3;
                for (int a = 0; a < 3; ++a) {
                    e1.x[a] = s1.x[a] - p0.x[a]; e2.x[a] = s2.x[a] - p0.x[a];
                    c1.x[a] = p0.x[a] + e1.x[a]; c2.x[a] = p0.x[a] + e2.x[a]; c3.x[a] = e1.x[a] + (e2.x[a] + p0.x[a]);      // bary_point order
                }
                if (!(c1 == s1) || !(c2 == s2)) continue;
                const V corner[4] = {p0, s1, s2, c3};
                int ca[3], cb[3];
                bool ok = true;
                for (int k = 0; k < 3 && ok; ++k) {
                    ca[k] = cb[k] = -1;
                    const V va = vtx(i, k), vb = vtx(j, k);
                    for (int q = 0; q < 4; ++q) { if (va == corner[q]) ca[k] = q; if (q > 0 && vb == corner[q]) cb[k] = q; }
                    ok = ca[k] >= 0 && cb[k] > 0;
                }
                if (!ok || cb[0] == cb[1] || cb[0] == cb[2] || cb[1] == cb[2]) continue;
                int32_t codeA, codeB;
                if (!map_code(ca, codeA) || !map_code(cb, codeB)) continue;
                float4 a{p0.x[0], p0.x[1], p0.x[2], 0.f}, b{e1.x[0], e1.x[1], e1.x[2], 0.f}, c{e2.x[0], e2.x[1], e2.x[2], 0.f};
                const int32_t ids = tri_id(i) | (tri_id(j) << 16);
                std::memcpy(&a.w, &ids, 4); std::memcpy(&b.w, &codeA, 4); std::memcpy(&c.w, &codeB, 4);
                prims.push_back(a); prims.push_back(b); prims.push_back(c);
                used[j] = 1; paired = true;
            }
        }
        if (!paired) {                                             // lone triangle: identity map
            float4 a = tris[(size_t) i * 3], b = tris[(size_t) i * 3 + 1], c = tris[(size_t) i * 3 + 2];
            const int cid[3] = {0, 1, 2};
            int32_t code; map_code(cid, code);
            const int32_t ids = tri_id(i) | (0xffff << 16);
            std::memcpy(&a.w, &ids, 4); std::memcpy(&b.w, &code, 4); std::memcpy(&c.w, &code, 4);
            prims.push_back(a); prims.push_back(b); prims.push_back(c);
        }
    }
}

// What the kernels receive (SceneView::tiny / tiny_meta): every primitive in PLANE FORM, computed here in double,
//   row 0 = (n, c0)   unit normal and n . p0:            t = -(n . o - c0) / (n . d)
//   row 1 = (a1, c1)  dual basis vector of e1, a1 . p0 + 1/2:  s - 1/2 = a1 . (o + t d) - c1
//   row 2 = (a2, c2)  dual basis vector of e2, a2 . p0:  the second plane coordinate likewise
//   row 3 = (bound on (s - 1/2) + (t - 1/2): 1 parallelogram / 0 triangle, ids, -, -)
// (a1 = e2 x n / |e1 x e2|, a2 = n x e1 / |e1 x e2| with the unit normal: a1 . e1 = a2 . e2 = 1, a1 . e2 = a2 . e1 = 0) -- 17 VALU
// operations per test against Moeller-Trumbore's 31 (psdr_device.h tiny_prim_test), and meta = (ids, codeA, codeB, the bound as float bits).
//
// AXIS-ALIGNED RECTANGLES (round 4: the walls, floor, ceiling and light of a room) get a slab-style row instead, in one of kAaSlots FIXED
// slots at the front of the arrays -- three per axis of the normal (slots 0-2: x, 3-5: y, 6-8: z; a fourth rectangle of an axis stays a
// plane-form primitive), so that each slot's test is compiled for its axis and reads its row at a compile-time address.  With the plane
// x_k = c and the in-plane axes (a, b) = the other two in ascending order:
//   row 0 = (c, centre_a, centre_b, half extent along a)
//   row 1 = (half extent along b, rho (float bits, low 4 bits = the slot), -, ids)
//   row 2 = S, the 2 x 2 map from the test's coordinates (u', v') = (p_a - centre_a, p_b - centre_b) to (s - 1/2, t - 1/2) -- a signed,
//           scaled permutation
// aa_prim_test: t = (c - o_k) / d_k with the ray's three reciprocals, two FMAs and two subtractions for (u', v'), |u'| <= h_a, |v'| <= h_b:
// 15 VALU and no v_rcp_f32 per test.  The second triangle of the pair is the half (s - 1/2) + (t - 1/2) > 0 = (S00 + S10) u' + (S01 + S11) v'
// > 0; divided by B = S01 + S11 that is rho u' + v' > 0 (B > 0) or its complement (B < 0: the two triangles swap their places in ids / codes
// here, so the kernels test one form).  aa_cnt = the slots in use per axis (x | y << 8 | z << 16); with any in use the plane-form
// primitives start at row kAaSlots, otherwise at row 0.  Returns the number of rows in use (SceneView::n_tiny).
inline int tiny_plane_form(const std::vector<float4> &prims_in, float4 *rows, int32_t *meta, int32_t *aa_cnt_out = nullptr, bool allow_aa = true, std::vector<int> *row_of_prim = nullptr) {
    const int n = (int) prims_in.size() / 3;
    struct Aa { int axis; double c, cu, cv, hu, hv, S[4]; };
    auto bits = [](const float4 &r) { int32_t v; std::memcpy(&v, &r.w, 4); return v; };
    auto classify = [&](const float4 &a, const float4 &b, const float4 &c, Aa &o) {
        if (((uint32_t) bits(a) >> 16) == 0xffffu) return false;                      // a lone triangle
        const double p0[3] = {a.x, a.y, a.z}, e1[3] = {b.x, b.y, b.z}, e2[3] = {c.x, c.y, c.z};
        int i1 = -1, i2 = -1, n1 = 0, n2 = 0;
        for (int k = 0; k < 3; ++k) { if (e1[k] != 0.0) { i1 = k; n1++; } if (e2[k] != 0.0) { i2 = k; n2++; } }
        if (n1 != 1 || n2 != 1 || i1 == i2) return false;
        for (int k = 0; k < 3; ++k) if (!std::isfinite(p0[k]) || !std::isfinite(e1[k]) || !std::isfinite(e2[k])) return false;
        o.axis = 3 - i1 - i2;
        const int ia = o.axis == 0 ? 1 : 0, ib = o.axis == 2 ? 1 : 2;
        double ctr[3];
        for (int k = 0; k < 3; ++k) ctr[k] = p0[k] + 0.5 * (e1[k] + e2[k]);
        o.c = p0[o.axis]; o.cu = ctr[ia]; o.cv = ctr[ib];
        o.hu = 0.5 * std::fabs(i1 == ia ? e1[ia] : e2[ia]); o.hv = 0.5 * std::fabs(i1 == ib ? e1[ib] : e2[ib]);
        if (i1 == ia) { o.S[0] = 1.0 / e1[i1]; o.S[1] = 0.0; o.S[2] = 0.0; o.S[3] = 1.0 / e2[i2]; }
        else          { o.S[0] = 0.0; o.S[1] = 1.0 / e1[i1]; o.S[2] = 1.0 / e2[i2]; o.S[3] = 0.0; }
        return true;
    };
    // slot of every primitive: an axis-aligned rectangle takes the next free slot of its axis, the others follow in their order
    std::vector<int> row_of((size_t) n, -1); std::vector<Aa> aa((size_t) n);
    int cnt[3] = {0, 0, 0};
    if (allow_aa && aa_cnt_out)
        for (int i = 0; i < n; ++i) {
            Aa &q = aa[(size_t) i];
            if (classify(prims_in[(size_t) i * 3], prims_in[(size_t) i * 3 + 1], prims_in[(size_t) i * 3 + 2], q) && cnt[q.axis] < kAaPerAxis) row_of[(size_t) i] = q.axis * kAaPerAxis + cnt[q.axis]++;
        }
    const int n_aa = cnt[0] + cnt[1] + cnt[2];
    int next = n_aa > 0 ? kAaSlots : 0;
    for (int r = 0; r < next; ++r) { for (int q = 0; q < 4; ++q) { rows[r * 4 + q] = float4{0.f, 0.f, 0.f, 0.f}; meta[r * 4 + q] = 0; } rows[r * 4].w = -1.f; }   // an unused slot never hits
    if (aa_cnt_out) *aa_cnt_out = cnt[0] | (cnt[1] << 8) | (cnt[2] << 16);
    for (int src = 0; src < n; ++src) {
        const bool is_aa = row_of[(size_t) src] >= 0;
        const int i = is_aa ? row_of[(size_t) src] : next++;
        const float4 &a = prims_in[(size_t) src * 3], &b = prims_in[(size_t) src * 3 + 1], &c = prims_in[(size_t) src * 3 + 2];
        int32_t ids = bits(a), codeA = bits(b), codeB = bits(c);
        const float lim = ((uint32_t) ids >> 16) != 0xffffu ? 1.f : 0.f;           // bound on (s - 1/2) + (t - 1/2)
        if (is_aa) {
            const Aa &q = aa[(size_t) src];
            const double A = q.S[0] + q.S[2], B = q.S[1] + q.S[3];
            if (B < 0.0) { ids = (int32_t) (((uint32_t) ids >> 16) | ((uint32_t) ids << 16)); std::swap(codeA, codeB); }
            const float rho = (float) (A / B);
            int32_t packed; std::memcpy(&packed, &rho, 4); packed = (packed & ~15) | i;
            rows[i * 4] = float4{(float) q.c, (float) q.cu, (float) q.cv, (float) q.hu};
            rows[i * 4 + 1] = float4{(float) q.hv, 0.f, 0.f, 0.f};
            std::memcpy(&rows[i * 4 + 1].y, &packed, 4); std::memcpy(&rows[i * 4 + 1].w, &ids, 4);
            rows[i * 4 + 2] = float4{(float) q.S[0], (float) q.S[1], (float) q.S[2], (float) q.S[3]};
        } else {
            const double p0[3] = {a.x, a.y, a.z}, e1[3] = {b.x, b.y, b.z}, e2[3] = {c.x, c.y, c.z};
            double nn[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
            const double len = std::sqrt(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]);
            double a1[3] = {0, 0, 0}, a2[3] = {0, 0, 0};
            if (len > 0.0 && std::isfinite(len)) {
                for (int k = 0; k < 3; ++k) nn[k] /= len;
                const double x1[3] = {e2[1] * nn[2] - e2[2] * nn[1], e2[2] * nn[0] - e2[0] * nn[2], e2[0] * nn[1] - e2[1] * nn[0]};      // e2 x n
                const double x2[3] = {nn[1] * e1[2] - nn[2] * e1[1], nn[2] * e1[0] - nn[0] * e1[2], nn[0] * e1[1] - nn[1] * e1[0]};      // n x e1
                for (int k = 0; k < 3; ++k) { a1[k] = x1[k] / len; a2[k] = x2[k] / len; }
            } else nn[0] = nn[1] = nn[2] = 0.0;                        // degenerate: n . d = 0 for every ray, never hit
            auto dot3 = [](const double *x, const double *y) { return x[0] * y[0] + x[1] * y[1] + x[2] * y[2]; };
            rows[i * 4] = float4{(float) nn[0], (float) nn[1], (float) nn[2], (float) dot3(nn, p0)};
            rows[i * 4 + 1] = float4{(float) a1[0], (float) a1[1], (float) a1[2], (float) (dot3(a1, p0) + 0.5)};      // the test works on s - 1/2, t - 1/2
            rows[i * 4 + 2] = float4{(float) a2[0], (float) a2[1], (float) a2[2], (float) (dot3(a2, p0) + 0.5)};
            rows[i * 4 + 3] = float4{lim, 0.f, 0.f, 0.f}; std::memcpy(&rows[i * 4 + 3].y, &ids, 4);
        }
        meta[i * 4] = ids; meta[i * 4 + 1] = codeA; meta[i * 4 + 2] = codeB;
        std::memcpy(&meta[i * 4 + 3], &lim, 4);
        if (row_of_prim) { row_of_prim->resize((size_t) n, -1); (*row_of_prim)[(size_t) src] = i; }
    }
    return next;
}

// SceneView::emit_rows: the rows of the kernel-argument table (tiny_plane_form's meta: ids = two 16-bit triangle ids, 0xffff = none) that hold an emitter triangle --
// or 0 when some emitter triangle is NOT among them (a mesh light inside a tree: the pre-test of direct_step then does not apply).  aa_cnt: slab slots in use per axis.
inline uint32_t tiny_emitter_rows(const int32_t *meta, int n_rows, int32_t aa_cnt, const std::vector<char> &is_emitter_tri) {
    const int cnt[3] = {aa_cnt & 255, (aa_cnt >> 8) & 255, aa_cnt >> 16};
    size_t want = 0, found = 0;
    for (char c : is_emitter_tri) want += c ? 1 : 0;
    uint32_t rows = 0;
    for (int i = 0; i < n_rows && i < 32; ++i) {
        if (aa_cnt != 0 && i < kAaSlots && (i % kAaPerAxis) >= cnt[i / kAaPerAxis]) continue;          // an unused slab slot
        const uint32_t ids = (uint32_t) meta[i * 4];
        const uint32_t t[2] = {ids & 0xffffu, ids >> 16};
        for (int a = 0; a < 2; ++a)
            if (t[a] != 0xffffu && t[a] < is_emitter_tri.size() && is_emitter_tri[t[a]]) { rows |= 1u << i; found++; }
    }
    return (want > 0 && found == want) ? rows : 0u;
}

// Occluder rows of a scene without a tree (SceneView::occ, psdr_device.h closest_hit MASKED): for every triangle r and every EMITTER triangle e the rows of
// the kernel-argument table a light ray from a point of r's primitive to a point of e's primitive has to test -- e's own primitive, plus every primitive P
// whose supporting plane has corners of (r's primitive, e's primitive) strictly on BOTH sides: a segment whose end points lie in one closed half space of
// P's plane does not cross it (a segment IN the plane gives n . d = 0: no hit in either test form).  Entries of a non-emitter e: all ones.  In a convex
// room (the Cornell box of BASELINE configs 1 / 2) every entry is e's primitive alone.  Evaluated in double from the packed primitives (p0 | ids, e1, e2).
// tol: a corner within tol of the plane counts as ON it (exact for the box: its corners are shared fp32 numbers).
inline void tiny_occluder_rows(const std::vector<float4> &prims, const std::vector<int> &row_of_prim, int num_tris, const std::vector<char> &is_emitter_tri,
                               std::vector<uint32_t> &occ) {
    const int n = (int) prims.size() / 3;
    occ.assign((size_t) num_tris * num_tris, 0xffffffffu);
    struct P { double c[4][3]; int nc; double nrm[3], d; int tris[2]; int row; };
    std::vector<P> ps((size_t) n);
    double scale = 0.0;
    for (int i = 0; i < n; ++i) {
        const float4 &a = prims[(size_t) i * 3], &b = prims[(size_t) i * 3 + 1], &c = prims[(size_t) i * 3 + 2];
        int32_t ids; std::memcpy(&ids, &a.w, 4);
        P &q = ps[(size_t) i];
        const bool quad = ((uint32_t) ids >> 16) != 0xffffu;
        q.tris[0] = ids & 0xffff; q.tris[1] = quad ? (int) ((uint32_t) ids >> 16) : -1;
        const double p0[3] = {a.x, a.y, a.z}, e1[3] = {b.x, b.y, b.z}, e2[3] = {c.x, c.y, c.z};
        q.nc = quad ? 4 : 3;
        for (int k = 0; k < 3; ++k) { q.c[0][k] = p0[k]; q.c[1][k] = p0[k] + e1[k]; q.c[2][k] = p0[k] + e2[k]; q.c[3][k] = p0[k] + e1[k] + e2[k]; }
        q.nrm[0] = e1[1] * e2[2] - e1[2] * e2[1]; q.nrm[1] = e1[2] * e2[0] - e1[0] * e2[2]; q.nrm[2] = e1[0] * e2[1] - e1[1] * e2[0];
        const double len = std::sqrt(q.nrm[0] * q.nrm[0] + q.nrm[1] * q.nrm[1] + q.nrm[2] * q.nrm[2]);
        for (int k = 0; k < 3; ++k) q.nrm[k] = len > 0.0 ? q.nrm[k] / len : 0.0;
        q.d = q.nrm[0] * p0[0] + q.nrm[1] * p0[1] + q.nrm[2] * p0[2];
        q.row = row_of_prim[(size_t) i];
        for (int v = 0; v < q.nc; ++v) for (int k = 0; k < 3; ++k) scale = std::max(scale, std::fabs(q.c[v][k]));
    }
    const double tol = 1e-6 * std::max(scale, 1e-30);
    int n_emitter_prims = 0;
    for (int i = 0; i < n; ++i) {
        bool em = false;
        for (int a = 0; a < 2; ++a) { const int t = ps[(size_t) i].tris[a]; em = em || (t >= 0 && t < num_tris && is_emitter_tri[(size_t) t]); }
        n_emitter_prims += em ? 1 : 0;
    }
    for (int r = 0; r < n; ++r)
        for (int e = 0; e < n; ++e) {
            uint32_t m = 1u << ps[(size_t) e].row;
            {   // r's primitive in e's plane (r == e, or a coplanar neighbour): such a ray runs IN the emitter's plane and never meets it (n . d = 0); the full
                // search reports whatever lies behind.  What a light ray's hit is USED for is "an emitter, at the sample's distance or beyond" (direct.cpp:138-141):
                // with ONE emitter primitive in the scene nothing behind can be one, and e's row alone gives the same outcome (no contribution); with several
                // the entry names every row
                const P &q = ps[(size_t) e];
                bool coplanar = true;
                for (int v = 0; v < ps[(size_t) r].nc; ++v) {
                    const double sd = q.nrm[0] * ps[(size_t) r].c[v][0] + q.nrm[1] * ps[(size_t) r].c[v][1] + q.nrm[2] * ps[(size_t) r].c[v][2] - q.d;
                    coplanar = coplanar && std::fabs(sd) <= tol;
                }
                if (coplanar && n_emitter_prims > 1) m = 0xffffffffu;
            }
            for (int p = 0; p < n; ++p) {
                if (p == e) continue;
                const P &q = ps[(size_t) p];
                double lo = 0.0, hi = 0.0;
                bool finite = true;
                auto side = [&](const P &x) { for (int v = 0; v < x.nc; ++v) { const double sd = q.nrm[0] * x.c[v][0] + q.nrm[1] * x.c[v][1] + q.nrm[2] * x.c[v][2] - q.d;
                                                                          if (!std::isfinite(sd)) finite = false; lo = std::min(lo, sd); hi = std::max(hi, sd); } };
                side(ps[(size_t) r]); side(ps[(size_t) e]);
                if (!finite || (lo < -tol && hi > tol)) m |= 1u << q.row;             // corners on both sides (or nothing known): the plane may be crossed
            }
            for (int a = 0; a < 2; ++a)
                for (int b = 0; b < 2; ++b) {
                    const int tr = ps[(size_t) r].tris[a], te = ps[(size_t) e].tris[b];
                    if (tr >= 0 && te >= 0 && tr < num_tris && te < num_tris && is_emitter_tri[(size_t) te]) occ[(size_t) tr * num_tris + te] = m;
                }
        }
}

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
    float kTraversalCost = 2.0f;       // node visit / triangle test: a visit is two slab tests, a stack access and two dependent loads --
                                       // measured (tools/bvh_param_sweep.py): 2.0 is 4-5 % ahead of 1.0 on cbox_bunny and the 50 k-triangle interior, 3.0 no better

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


// ------------------------------------------------------------------------ 4-wide tree (topology)
// The kernels walk a 4-wide tree (psdr_device.h Bvh4Node) derived from the BVH2: every 4-wide node adopts grandchildren of its BVH2 node
// until it has four children, always opening the inner child with the largest box (the greedy surface-area collapse).  Only the
// TOPOLOGY is decided here; the quantised boxes are (re)computed on the device from the BVH2 nodes after every build and refit
// (psdr_hip.hip k_bvh4_fill): child[i][c] = 4-wide node index / leaf code / kNoChild, src[i][c] = 2 * (BVH2 node holding the box) + side.
// Nodes come out in level order across all roots (a prefix = the top of every tree: LDS staging).
struct Bvh4Topology {
    std::vector<int32_t> child, src;      // [n4][4]
    std::vector<int32_t> roots;           // per BVH2 root: 4-wide root (encoded like a child)
    int n4 = 0, stack_need = 0;           // traversal-stack entries a walk can need (worst case over all root-to-leaf paths) 
};
inline void collapse_bvh4(const std::vector<BvhNode> &nodes, const std::vector<int32_t> &roots2, Bvh4Topology &out) {
    out.child.clear(); out.src.clear(); out.roots.clear(); out.n4 = 0; out.stack_need = 0;
    std::vector<int32_t> queue;                       // BVH2 node of every 4-wide node, in creation (= level) order
    std::vector<int32_t> id4(nodes.size(), -1);
    auto make = [&](int32_t n2) { id4[(size_t) n2] = (int32_t) queue.size(); queue.push_back(n2); return id4[(size_t) n2]; };
    for (int32_t r : roots2) out.roots.push_back(r >= 0 ? make(r) : r);
    auto area_of = [&](int32_t s) {
        const BvhNode &n = nodes[(size_t) (s >> 1)];
        const float *lo = (s & 1) ? n.lo1 : n.lo0, *hi = (s & 1) ? n.hi1 : n.hi0;
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    };
    auto child_of = [&](int32_t s) { const BvhNode &n = nodes[(size_t) (s >> 1)]; return (s & 1) ? n.c1 : n.c0; };
    for (size_t q = 0; q < queue.size(); ++q) {
        const int32_t n2 = queue[q];
        int32_t slots[4] = {2 * n2, 2 * n2 + 1, -1, -1};
        int cnt = 2;
        while (cnt < 4) {
            int best = -1; float best_area = -1.f;
            for (int i = 0; i < cnt; ++i)
                if (child_of(slots[i]) >= 0 && area_of(slots[i]) > best_area) { best_area = area_of(slots[i]); best = i; }
            if (best < 0) break;
            const int32_t c = child_of(slots[best]);
            slots[best] = 2 * c; slots[cnt++] = 2 * c + 1;
        }
        for (int i = 0; i < 4; ++i) {
            if (i >= cnt) { out.child.push_back(kNoChild); out.src.push_back(-1); continue; }
            const int32_t c = child_of(slots[i]);
            out.child.push_back(c >= 0 ? make(c) : c);
            out.src.push_back(slots[i]);
        }
    }
    out.n4 = (int) queue.size();
    // stack entries: a node with k children leaves k - 1 of them on the stack while the walk is below one of them
    std::vector<int> need((size_t) out.n4, 0);
    for (int i = out.n4 - 1; i >= 0; --i) {
        int k = 0, deepest = 0;
        for (int c = 0; c < 4; ++c) {
            const int32_t ch = out.child[(size_t) i * 4 + c];
            if (ch == kNoChild) continue;
            ++k;
            if (ch >= 0) deepest = std::max(deepest, need[(size_t) ch]);
        }
        need[(size_t) i] = std::max(k - 1, 0) + deepest;
    }
    for (int32_t r : out.roots) if (r >= 0) out.stack_need = std::max(out.stack_need, need[(size_t) r]);
}

// ------------------------------------------------------------------- two-level tree (forest)
// Scenes made of a few small meshes (walls, lights) and a few large ones (objects): every mesh with at least
// kMinBlasTris triangles gets its OWN tree; the triangles of the others stay "inline" (tested by every ray in a
// wave-uniform loop, SceneView::tiny).  The trees share one node array, ordered by LEVEL across the forest (all
// roots first, then all depth-1 nodes, ...): a prefix is still "the top of every tree" (LDS staging) and a level is
// still one contiguous range (device refit, bottom-up).  Used when the scene fits the kernel-argument tables:
// <= kTinyTris inline triangles and 1..kMaxBlas trees; otherwise the single tree above serves the scene.
struct ForestBuilder {
    std::vector<BvhNode> nodes;
    std::vector<float4> btris;
    std::vector<float4> inline_tris;          // p0 | id, e1, e2 per inline triangle
    std::vector<int32_t> inline_ids;
    std::vector<int32_t> roots;               // encoded like a child, one per tree
    std::vector<int> level_start;
    int max_depth = 0;
    float pad = 0.f;
    int max_leaf = 4; float traversal_cost = 2.0f;      // Builder::kMaxLeaf / kTraversalCost of the per-mesh trees

    static bool eligible(const int32_t *tri_mesh, int T, int num_meshes) {
        std::vector<int> cnt((size_t) std::max(num_meshes, 1), 0);
        for (int i = 0; i < T; ++i) {
            const int m = tri_mesh[i] & ~PSDR_TRI_FACE_NORMALS;
            if (m < 0 || m >= num_meshes) return false;
            cnt[m]++;
        }
        int n_inline = 0, n_blas = 0;
        for (int c : cnt) { if (c >= kMinBlasTris) n_blas++; else n_inline += c; }
        // an inline primitive packs two GLOBAL triangle ids into 16 bits each (pack_tiny_prims, 0xffff = none): a room whose walls
        // are listed behind >= 65535 triangles of other meshes keeps the single tree
        for (int i = 0xffff; i < T; ++i)
            if (cnt[tri_mesh[i] & ~PSDR_TRI_FACE_NORMALS] < kMinBlasTris) return false;
        return n_blas >= 1 && n_blas <= kMaxBlas && n_inline <= 2 * kTinyTris;      // primitives after pairing are checked by the caller
    }

    const char *run(const float *rows, const int32_t *tri_mesh, int T, int num_meshes) {
        std::vector<std::vector<int32_t>> ids((size_t) num_meshes);
        for (int i = 0; i < T; ++i) ids[tri_mesh[i] & ~PSDR_TRI_FACE_NORMALS].push_back(i);
        // the padding of the whole scene (Builder::run derives it from ITS triangles): one value for every tree
        float slo[3] = {INFINITY, INFINITY, INFINITY}, shi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int i = 0; i < T; ++i) {
            const float *r = rows + (size_t) i * PSDR_TRI_STRIDE;
            for (int k = 0; k < 3; ++k) {
                const float p = r[k], q = r[k] + r[3 + k], w = r[k] + r[6 + k];
                if (!std::isfinite(p) || !std::isfinite(q) || !std::isfinite(w)) return "psdr_bvh_build: non-finite vertex";
                slo[k] = std::min(slo[k], std::min(p, std::min(q, w))); shi[k] = std::max(shi[k], std::max(p, std::max(q, w)));
            }
        }
        pad = std::max(1e-6f, 1e-5f * std::max(shi[0] - slo[0], std::max(shi[1] - slo[1], shi[2] - slo[2])));
        struct Tree { std::vector<BvhNode> nodes; std::vector<int> depth; int32_t root; int btri_base; };
        std::vector<Tree> trees;
        std::vector<float> sub;
        for (int m = 0; m < num_meshes; ++m) {
            const std::vector<int32_t> &id = ids[m];
            if ((int) id.size() < kMinBlasTris) {
                for (int32_t t : id) {
                    const float *r = rows + (size_t) t * PSDR_TRI_STRIDE;
                    float4 a{r[0], r[1], r[2], 0.f};
                    std::memcpy(&a.w, &t, 4);
                    inline_tris.push_back(a); inline_tris.push_back(float4{r[3], r[4], r[5], 0.f}); inline_tris.push_back(float4{r[6], r[7], r[8], 0.f});
                    inline_ids.push_back(t);
                }
                continue;
            }
            sub.resize(id.size() * PSDR_TRI_STRIDE);
            for (size_t i = 0; i < id.size(); ++i) std::memcpy(&sub[i * PSDR_TRI_STRIDE], rows + (size_t) id[i] * PSDR_TRI_STRIDE, PSDR_TRI_STRIDE * sizeof(float));
            Builder b;
            b.kMaxLeaf = max_leaf; b.kTraversalCost = traversal_cost;
            int32_t root = 0;
            if (const char *err = b.run(sub.data(), (int) id.size(), root)) return err;
            // Builder padded its boxes with ITS pad; re-pad consistently is unnecessary (boxes only need to contain
            // the triangles), but the device refit uses one pad: take the larger
            pad = std::max(pad, b.pad);
            Tree t; t.root = root; t.btri_base = (int) btris.size() / 3;
            for (size_t i = 0; i < b.btris.size(); i += 3) {          // local -> global triangle ids
                float4 a = b.btris[i];
                int32_t local; std::memcpy(&local, &a.w, 4);
                const int32_t global = id[(size_t) local];
                std::memcpy(&a.w, &global, 4);
                btris.push_back(a); btris.push_back(b.btris[i + 1]); btris.push_back(b.btris[i + 2]);
            }
            auto fix_leaf = [&](int32_t c) { if (c >= 0) return c; const int enc = ~c; return (int32_t) ~((((enc >> 3) + t.btri_base) << 3) | (enc & 7)); };
            t.nodes = b.nodes;
            t.depth.assign(t.nodes.size(), 0);
            for (size_t i = 0; i < t.nodes.size(); ++i) {             // breadth-first order: children come after their parent
                BvhNode &n = t.nodes[i];
                if (n.c0 >= 0) t.depth[n.c0] = t.depth[i] + 1; else n.c0 = fix_leaf(n.c0);
                if (n.c1 >= 0) t.depth[n.c1] = t.depth[i] + 1; else n.c1 = fix_leaf(n.c1);
            }
            if (t.root < 0) t.root = fix_leaf(t.root);
            max_depth = std::max(max_depth, b.max_depth);
            trees.push_back(std::move(t));
        }
        // forest-wide level order
        std::vector<std::vector<int>> newid(trees.size());
        int next = 0, level = 0;
        bool any = true;
        for (size_t k = 0; k < trees.size(); ++k) newid[k].assign(trees[k].nodes.size(), -1);
        while (any) {
            any = false;
            level_start.push_back(next);
            for (size_t k = 0; k < trees.size(); ++k)
                for (size_t i = 0; i < trees[k].nodes.size(); ++i)
                    if (trees[k].depth[i] == level) { newid[k][i] = next++; any = true; }
            if (!any) level_start.pop_back();
            ++level;
        }
        level_start.push_back(next);
        nodes.assign((size_t) next, BvhNode{});
        for (size_t k = 0; k < trees.size(); ++k) {
            for (size_t i = 0; i < trees[k].nodes.size(); ++i) {
                BvhNode n = trees[k].nodes[i];
                if (n.c0 >= 0) n.c0 = newid[k][n.c0];
                if (n.c1 >= 0) n.c1 = newid[k][n.c1];
                nodes[(size_t) newid[k][i]] = n;
            }
            roots.push_back(trees[k].root >= 0 ? newid[k][trees[k].root] : trees[k].root);
        }
        if (max_depth > kBvhStack - 2) return "psdr_bvh_build: tree too deep for the traversal stack";
        return nullptr;
    }
    // box of tree k = union of its root's child boxes (a single-leaf tree: the box of its triangles)
    void tree_box(int k, float *lo, float *hi) const {
        const int32_t r = roots[(size_t) k];
        for (int a = 0; a < 3; ++a) { lo[a] = INFINITY; hi[a] = -INFINITY; }
        if (r >= 0) {
            const BvhNode &n = nodes[(size_t) r];
            for (int a = 0; a < 3; ++a) { lo[a] = std::min(n.lo0[a], n.lo1[a]); hi[a] = std::max(n.hi0[a], n.hi1[a]); }
        } else {
            const int enc = ~r, first = enc >> 3, cnt = (enc & 7) + 1;
            for (int i = 0; i < cnt; ++i) {
                const float4 a = btris[(size_t) (first + i) * 3], b = btris[(size_t) (first + i) * 3 + 1], c = btris[(size_t) (first + i) * 3 + 2];
                const float p[3] = {a.x, a.y, a.z}, q[3] = {a.x + b.x, a.y + b.y, a.z + b.z}, w[3] = {a.x + c.x, a.y + c.y, a.z + c.z};
                for (int x = 0; x < 3; ++x) { lo[x] = std::min(lo[x], std::min(p[x], std::min(q[x], w[x]))); hi[x] = std::max(hi[x], std::max(p[x], std::max(q[x], w[x]))); }
            }
            for (int x = 0; x < 3; ++x) { lo[x] -= pad; hi[x] += pad; }
        }
    }
};

}  // namespace psdr
