// psdr_device.h -- scene views, BVH traversal, hit reconstruction, BSDFs, emitter sampling, camera
// and the per-sample estimators (Li, primary/secondary edge samples) of the gfx950 renderer.
//
// Mirrors, function by function, the reference hot path; each routine cites
// the reference file:line it implements.  Scalar type R is float (renderC) or Dual<K> (renderD,
// forward mode).
#pragma once
#include "psdr_math.h"
#include "../../include/psdr_hip.h"

namespace psdr {

// ------------------------------------------------------------------------------ BVH
// BVH2, 64-byte nodes holding BOTH children's boxes so one 64 B fetch decides the descent.
// child >= 0: inner node index.  child < 0: leaf, ~child = (first << 3) | (count - 1) into the
// reordered triangle array `btris` (3 x float4 per triangle: p0|tri_id, e1, e2).
struct __attribute__((aligned(16))) BvhNode {
    float lo0[3], hi0[3];
    float lo1[3], hi1[3];
    int32_t c0, c1;
    int32_t pad[2];
};
static_assert(sizeof(BvhNode) == 64, "BvhNode must be 64 bytes");

// What the kernels WALK: a 4-wide tree with quantised child boxes, one 64-byte line per node (round 3).  The walk of incoherent rays
// is bound by the rate at which a CU gathers distinct cache lines (tools/micro/gather_rate.hip: ~2.3-2.9 cycles per line, 186 cycles
// for the 64 node fetches of a wave, whatever the occupancy) -- four times the VALU time of a BVH2 step -- so the lever is FEWER LINES
// PER RAY: four children per line instead of two, half the node visits.  The BVH2 above stays the master structure (SAH build, LBVH
// build, device refit); the 4-wide nodes are derived from it on the device (psdr_hip.hip k_bvh4_fill) after every build / refit.
//   child box c on axis a:  [org[a] + qlo[a].byte(c) * 2^(exps.byte(a) - 127),  org[a] + qhi[a].byte(c) * 2^(exps.byte(a) - 127)]
//   (lower planes rounded down, upper planes rounded up: the boxes only grow)
struct __attribute__((aligned(16))) Bvh4Node {
    float org[3];
    uint32_t exps;
    uint32_t qlo[3], qhi[3];
    int32_t child[4];          // >= 0: Bvh4Node index; < 0: leaf, encoded as in BvhNode; kNoChild: empty slot
    uint32_t pad[2];
};
static_assert(sizeof(Bvh4Node) == 64, "Bvh4Node must be 64 bytes");
constexpr int32_t kNoChild = 0x7ffffffe;

constexpr int kBvhStack = 40;     // builder guarantees depth <= kBvhStack - 2
constexpr int kBlock = 256;

constexpr int kTinyTris = 16;
constexpr int kAaPerAxis = 3, kAaSlots = 3 * kAaPerAxis;   // slab-form rows of axis-aligned rectangles in front of the plane-form primitives (tiny_plane_form)
constexpr int kTinyRows = kTinyTris + kAaSlots;
constexpr int kMaxBlas = 16;
constexpr int kMinBlasTris = 64;      // meshes below this size stay inline in the top level
struct SceneView {
    psdr_scene_desc d;
    const BvhNode *nodes;  // BVH2 (master structure; walked by the host tests only)
    const float4 *btris;
    int32_t root;          // encoded like a child (negative = single leaf)
    const Bvh4Node *nodes4; // the 4-wide tree the kernels walk
    int32_t root4;         // its root (encoded like a child); the roots of a two-level scene's trees: blas_hi[k].w
    // LDS-staged prefix of the scene (device only; offsets in bytes into the dynamic LDS block):
    // the first n_lnodes BVH nodes (breadth-first order = the top of the tree), the first
    // n_lbtris leaf triangles and the first n_ltri TriangleInfo rows.
    int32_t n_lnodes, n_lbtris, n_ltri;
    int32_t off_lnodes, off_lbtris, off_ltri;
    int32_t off_lprim;     // the hit rows of the kernel-argument primitives, kTinyHitWords words each (resolve_tiny_hit)
    // Tiny scenes (<= kTinyTris triangles, e.g. the 12-triangle Cornell box): the leaf triangles travel IN THE
    // KERNEL ARGUMENTS and closest_hit tests all of them in order -- wave-uniform s_load from the kernarg
    // segment into SGPRs, no tree, no per-lane node fetches, no stack traffic, no divergence between lanes.
    int32_t n_tiny;
    int32_t aa_cnt;                             // slab-form slots in use per axis (x | y << 8 | z << 16): rows 0 .. kAaSlots-1; the plane-form primitives then start at row kAaSlots (tiny_plane_form)
    float4 tiny[kTinyRows * 4];                 // plane form: (n | c0), (a1 | c1), (a2 | c2), (bound, ids, -, -) per primitive: 64 bytes, ONE scalar load (tiny_plane_form)
    int32_t tiny_meta[kTinyRows * 4];           // (ids, codeA, codeB, bound on s + t - 1 as float bits: 1 parallelogram / 0 triangle) per primitive
    // Two-level tree (psdr_bvh_build.h ForestBuilder; scenes of a few small meshes plus a few large ones -- a room with
    // objects): the triangles of the small meshes are the primitives above, every large mesh has its OWN tree in `nodes`,
    // and its box + root travel in the kernel arguments too.  A closest-hit query first tests the inline primitives
    // and the tree boxes in a wave-uniform loop (SGPR operands, no divergence); only a ray whose segment [0, t_best]
    // enters a box walks that tree.  In a room most rays never do (cbox_bunny: 7-14 % of the bounce rays).
    int32_t n_blas;
    float4 blas_lo[kMaxBlas], blas_hi[kMaxBlas];      // lo.w = root of the tree (encoded like a child), hi.w = its root in the 4-wide tree
    // kSceneTiny launches: EVERY table a path vertex reads is staged in LDS (byte offsets into the dynamic LDS block; same layouts as the
    // caller's tables) -- the per-vertex chain tri_mesh -> mesh_bsdf -> bsdf_rec -> texels and the emitter lookups are LDS reads (~64 clk)
    // instead of dependent global loads (~300-500 clk each).  lt_tex < 0: the texel pool is too large and stays in global memory.
    int32_t lt_trimesh, lt_meshbsdf, lt_meshemitter, lt_bsdf, lt_emf, lt_emi, lt_fcmf, lt_fpmf, lt_uv, lt_tex, lt_ecmf, lt_epmf;
    int32_t lt_nfaces, lt_end;                          // entries of face_cmf / face_pmf staged; end of the block (bytes)
    int32_t literal_forms;                              // PSDR_FLAG_LITERAL_FORMS: the reference's literal fp32 expressions (psdr_hip.h)
    // Occluder rows of a scene without a tree (round 6; psdr_bvh_build.h tiny_occluder_rows): occ[r * num_tris + e] = the rows of `tiny` a light ray from a
    // point of triangle r towards a point of EMITTER triangle e has to test -- e's own primitive plus every primitive whose plane separates a corner of
    // r's primitive from a corner of e's (all ones where e is no emitter triangle).  In a convex room that is e's primitive alone: the light rays, 3 of a
    // PathTracer(3) path's 7, test one primitive instead of six.  nullptr: every row.  lt_occ: its staged copy (kSceneTiny launches), < 0: none.
    const uint32_t *occ;
    int32_t lt_occ;
    // Rows of `tiny` that hold an EMITTER triangle, when every emitter triangle of the scene is a kernel-argument primitive (0 otherwise: unknown).  A BSDF-sampled ray
    // whose hit matters only if it is an emitter -- DirectIntegrator (direct.cpp:86-91: `active1 &= neq(its1.shape->emitter(), nullptr)`), the last vertex of a PathTracer
    // path -- is tested against these rows first; if it meets none of them nothing along it can contribute and it is not traced (direct_step, round 6).
    uint32_t emit_rows;
};

#if defined(__HIP_DEVICE_COMPILE__)
extern __shared__ __attribute__((aligned(16))) unsigned char psdr_dyn_lds[];
#define PSDR_LDS_TABLE(T, off) reinterpret_cast<const T *>(psdr_dyn_lds + (off))
#else
#define PSDR_LDS_TABLE(T, off) static_cast<const T *>(nullptr)
#endif

// K sets of forward-mode tangent tables (struct of K pointer groups).  FLAGS: compile-time properties of
// the SCENE the kernel instance serves, carried by the type every estimator already receives, so that
// scenes without the feature do not pay registers / instruction cache for its code:
//   kSceneEnv    an EnvironmentMap exists (lat-long lookup, cell sampling; C2 renderD K=3: 5.3 ms without
//                the code, 7.1 ms with a run-time branch)
//   kSceneRough  a RoughConductor BSDF exists (GGX + conductor Fresnel, and in reverse mode their Dual<8>
//                adjoints: diffuse-only scenes run renderD K=3 15 % and reverse mode 40 % faster without)
//   kSceneForest the scene has a two-level tree (SceneView::n_blas > 0): the box loop, the per-tree walks and the
//                class-binned streams exist only in these instances (with a run-time switch instead the Cornell-box
//                kernels spilled twice as many SGPRs: C2 renderC 1.41 -> 1.80 ms)
//   kSceneTiny   ALL primitives of the scene travel in the kernel arguments (SceneView::n_tiny > 0, no tree at all -- the 12-triangle
//                Cornell box of C1 / C2): the instance carries no tree walk, no traversal stack and no global-memory fallback of the
//                LDS-staged TriangleInfo rows -- fewer registers (one more wave per SIMD), a third of the code
//   kScenePre    (kernel instances only, never a scene's flag set) the tree walks of this kernel's rays were done beforehand by the dense trace kernel
//                (psdr_hip.hip k_wf_trace): closest_hit tests the kernel-argument primitives and merges the tree hit it finds in TraversalStack::pre
constexpr int kSceneEnv = 1, kSceneRough = 2, kSceneAll = 3, kSceneForest = 4, kSceneTiny = 8, kScenePre = 16;
// closest_hit's FOREST argument of a flag set: 3 = two-level with the tree hits traced beforehand, 2 = kernel-argument primitives only, 1 = two-level,
// 0 = one tree (or a tiny scene served by a general instance)
template <int FLAGS> constexpr int tree_mode() { return (FLAGS & kSceneTiny) ? 2 : ((FLAGS & kSceneForest) ? ((FLAGS & kScenePre) ? 3 : 1) : 0); }
template <int K, int FLAGS = kSceneRough> struct TangentView {
    static constexpr int flags = FLAGS, k = K;
    static constexpr bool has_env = (FLAGS & kSceneEnv) != 0, has_rough = (FLAGS & kSceneRough) != 0, forest = (FLAGS & kSceneForest) != 0, tiny = (FLAGS & kSceneTiny) != 0;
    psdr_tangents t[K > 0 ? K : 1];
    // one bit per triangle: some tangent set moves this row of tri_info (render_fwd computes it per launch, psdr_kernels.h k_tangent_live); nullptr:
    // unknown.  A translation of one mesh leaves the rows of every other mesh at rest -- in a room the walls, where most path vertices land -- and
    // load_tri then fetches the 22 value words only instead of 22 + 22 K.
    const uint32_t *live = nullptr;
};

struct Hit { int tri; float u, v, t; };
constexpr int kPreBsdfRay = 0, kPreLightRay = 1, kPrePrimaryRay = 2;     // TraversalStack::pre slots: the two rays of a path vertex (direct_step), the ray that found the vertex (Li)

// ------------------------------------------------------------------------ small-table access
// The tables a path vertex looks up besides its TriangleInfo row.  Instances compiled for kSceneTiny read LDS copies of ALL of them (the launch
// staged them, make_ctx / setup_lds); the two-level instances (kSceneForest: a room around a few large meshes) read LDS copies of the MESH-level
// ones -- mesh -> bsdf / emitter, the BSDF and emitter records, the emitters' face distributions, a small texel pool -- and keep the
// per-triangle tables (tri_mesh, tri_uv: one entry per triangle of the large meshes) in global memory; every other instance reads the caller's
// tables.  A vertex' chain tri_mesh -> mesh_bsdf -> bsdf_rec -> texel is then ONE global load instead of four dependent ones -- what a
// kernel at 2-3 waves per SIMD (the dual-number and adjoint instances) waits for most.  FL = scene flag set.
#ifndef PSDR_FOREST_LDS_TABLES
#define PSDR_FOREST_LDS_TABLES 1
#endif
template <int FL> struct Tab {
    static constexpr bool lds = (FL & kSceneTiny) != 0;
    static constexpr bool lds_small = lds || (PSDR_FOREST_LDS_TABLES && (FL & kSceneForest) != 0);
    static PSDR_HD int tri_mesh(const SceneView &sc, int tri) { return lds ? PSDR_LDS_TABLE(int32_t, sc.lt_trimesh)[tri] : sc.d.tri_mesh[tri]; }
    static PSDR_HD int mesh_bsdf(const SceneView &sc, int mesh) { return lds_small ? PSDR_LDS_TABLE(int32_t, sc.lt_meshbsdf)[mesh] : sc.d.mesh_bsdf[mesh]; }
    static PSDR_HD int mesh_emitter(const SceneView &sc, int mesh) { return lds_small ? PSDR_LDS_TABLE(int32_t, sc.lt_meshemitter)[mesh] : sc.d.mesh_emitter[mesh]; }
    static PSDR_HD const int32_t *bsdf_rec(const SceneView &sc, int id) {
        return (lds_small ? PSDR_LDS_TABLE(int32_t, sc.lt_bsdf) : sc.d.bsdf_rec) + (size_t) (id < 0 ? 0 : id) * PSDR_BSDF_STRIDE;
    }
    static PSDR_HD const float *emitter_f(const SceneView &sc, int e) { return (lds_small ? PSDR_LDS_TABLE(float, sc.lt_emf) : sc.d.emitter_f) + (size_t) e * PSDR_EMITTER_F_STRIDE; }
    static PSDR_HD const int32_t *emitter_i(const SceneView &sc, int e) { return (lds_small ? PSDR_LDS_TABLE(int32_t, sc.lt_emi) : sc.d.emitter_i) + (size_t) e * PSDR_EMITTER_I_STRIDE; }
    static PSDR_HD const float *face_cmf(const SceneView &sc) { return lds_small ? PSDR_LDS_TABLE(float, sc.lt_fcmf) : sc.d.face_cmf; }
    static PSDR_HD const float *face_pmf(const SceneView &sc) { return lds_small ? PSDR_LDS_TABLE(float, sc.lt_fpmf) : sc.d.face_pmf; }
    static PSDR_HD const float *emitter_cmf(const SceneView &sc) { return lds_small ? PSDR_LDS_TABLE(float, sc.lt_ecmf) : sc.d.emitter_cmf; }
    static PSDR_HD const float *emitter_pmf(const SceneView &sc) { return lds_small ? PSDR_LDS_TABLE(float, sc.lt_epmf) : sc.d.emitter_pmf; }
    static PSDR_HD const float *tri_uv(const SceneView &sc, int tri) { return (lds ? PSDR_LDS_TABLE(float, sc.lt_uv) : sc.d.tri_uv) + (size_t) tri * PSDR_TRIUV_STRIDE; }
};

// Traversal stack: LDS on the device (one column per lane: conflict-free, no scratch traffic),
// a plain array on the host (tests).
// -DPSDR_STAGE_CLOCKS (developer build, tools/r05_clk.sh): where the waves of one kernel spend their wall time.  A mark drains the memory counters and
// reads s_memtime; the per-phase sums of all waves land behind the ray counters (psdr_get_counters prints them).  Off: the macros are empty.
#ifdef PSDR_STAGE_CLOCKS
struct StageClk { unsigned long long t[12]; unsigned long long last; };
// the clocks of a wave live in LDS (one StageClk per wave, written by the first active lane): no register of the instrumented kernel stays live for them
#define PSDR_CLK_MARK_P(clk, i) do { if (clk) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    if ((int) (threadIdx.x & 63) == __ffsll((long long) __ballot(1)) - 1) { (clk)->t[i] += now_ - (clk)->last; (clk)->last = now_; } } } while (0)
#define PSDR_CLK_MARK_ST(st, i) PSDR_CLK_MARK_P((st).clk, i)
#else
#define PSDR_CLK_MARK_P(clk, i) do { } while (0)
#define PSDR_CLK_MARK_ST(st, i) do { } while (0)
#endif
struct TraversalStack {
    // kScenePre instances: the closest TREE hits of the vertex' two rays (tri < 0: none), found beforehand by the dense trace kernel
    Hit pre[3];
#ifdef PSDR_STAGE_CLOCKS
    StageClk *clk = nullptr;
#endif
#if defined(__HIP_DEVICE_COMPILE__)
    int32_t *base;   // &lds[threadIdx.x], stride kBlock
    __device__ __forceinline__ void put(int i, int32_t v) { base[i * kBlock] = v; }
    __device__ __forceinline__ int32_t get(int i) const { return base[i * kBlock]; }
#else
    int32_t a[kBvhStack];
    void put(int i, int32_t v) { a[i] = v; }
    int32_t get(int i) const { return a[i]; }
#endif
};

PSDR_HD int __float_as_int_hd(float f) { union { float f; int i; } c; c.f = f; return c.i; }
PSDR_HD float __int_as_float_hd(int i) { union { float f; int i; } c; c.i = i; return c.f; }

PSDR_HD bool slab(const float *lo, const float *hi, const Vec3f &o, const Vec3f &inv, float tmax, float &t_entry) {
    // hit test of one box + entry distance.  NaNs (0*inf) drop out of fmin/fmax.
    // (lo - o) * inv is kept as two operations: folding it into fma(lo, inv, -o * inv) loses the exact
    // difference near the origin of the ray and was measured 2.5x SLOWER on cbox_bunny paths.  Also measured and dropped: both
    // children of a node on packed fp32 (boxes interleaved per axis, v_pk_add_f32 / v_pk_mul_f32: 12 instead of 24 instructions
    // per visit, same bits) -- C4 fused 29.3 -> 31.3 ms, C3 direct renderC 1.15 -> 1.34 ms: a packed fp32 instruction is no
    // cheaper than the two it replaces here, and the broadcast operands cost moves.
    const float ax = (lo[0] - o.x) * inv.x, bx = (hi[0] - o.x) * inv.x;
    const float ay = (lo[1] - o.y) * inv.y, by = (hi[1] - o.y) * inv.y;
    const float az = (lo[2] - o.z) * inv.z, bz = (hi[2] - o.z) * inv.z;
    t_entry = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.f));
    const float t1 = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), tmax));
    return t_entry <= t1;
}

// The leaf test of closest_hit on ONE known triangle (reverse mode replays the hits recorded in its value
// sweep): the same arithmetic, hence the same (u, v, t) as the traversal that found the triangle.
PSDR_HD Hit hit_on_triangle(int tri, const Vec3f &p0, const Vec3f &e1, const Vec3f &e2, const Vec3f &o, const Vec3f &d) {
    const Vec3f h = cross(d, e2);
    const float det = dot(e1, h);
    const float f = 1.f / det;
    const Vec3f s{o.x - p0.x, o.y - p0.y, o.z - p0.z};
    const Vec3f q = cross(s, e1);
    Hit r; r.tri = tri; r.u = f * dot(s, h); r.v = f * dot(d, q); r.t = f * dot(e2, q);
    return r;
}

// Moeller-Trumbore on one leaf triangle (p0 | id, e1, e2); keeps the closer hit in `best` (the OptiX built-in
// triangle test is closed source).  A precomputed plane form (18 FMAs) was measured: no faster in the tree
// walk (latency-, not ALU-bound) and less accurate.
// IGN: skip the triangles ig0 / ig1 (the faces adjacent to a secondary edge, for the rays that start ON that edge).
template <bool IGN = false>
PSDR_HD void leaf_triangle_test(const float4 &a, const float4 &b, const float4 &c, const Vec3f &o, const Vec3f &d, Hit &best, int ig0 = -1, int ig1 = -1) {
    const Vec3f e1{b.x, b.y, b.z}, e2{c.x, c.y, c.z};
    const Vec3f h = cross(d, e2);
    const float det = dot(e1, h);
    const float f = 1.f / det;
    const Vec3f s{o.x - a.x, o.y - a.y, o.z - a.z};
    const float u = f * dot(s, h);
    const Vec3f q = cross(s, e1);
    const float v = f * dot(d, q);
    const float t = f * dot(e2, q);
    // det != 0 and u <= 1 of the textbook test are implied: det == 0 makes u, v infinite or NaN (every comparison
    // below fails), and v >= 0 with u + v <= 1 gives u <= 1 in floating point too (fl(u + v) >= u)
    // closest_hit starts from the open upper bound next_above(tmax), so "t <= tmax, the first of equal hits wins"
    // is the single strict comparison t < best.t
    bool hit = u >= 0.f && v >= 0.f && u + v <= 1.f && t >= kRayEpsilon && t < best.t;
    if (IGN) { const int id = __float_as_int_hd(a.w); hit = hit && id != ig0 && id != ig1; }
    if (hit) { best.t = t; best.u = u; best.v = v; best.tri = __float_as_int_hd(a.w); }
}

// The triangles of a leaf that live in global memory (L2): TWO at a time -- the six loads of a pair are in flight together, so a leaf of four
// costs two memory latencies instead of four (the loop over one triangle at a time waited for each in turn).  An odd end tests its last
// triangle twice: the second test fails t < best.t, nothing changes.
// Measured (profiles/r03_leaf_pair_ab.txt): the rough-conductor two-level kernels (flag set 6: C5) gain 2-10 %; the lean diffuse kernels at 5 waves
// per SIMD LOSE 4-10 % to the twelve extra registers -- so the pairing is a per-translation-unit choice (psdr_variant.hip).
#ifndef PSDR_LEAF_PAIR
#define PSDR_LEAF_PAIR 0
#endif
template <bool IGN>
PSDR_HD void leaf_from_memory(const float4 *bt, int cnt, const Vec3f &o, const Vec3f &d, Hit &best, int ig0, int ig1) {
#if PSDR_LEAF_PAIR
    for (int i = 0; i < cnt; i += 2) {
        const int j = i + 1 < cnt ? i + 1 : i;
        const float4 a0 = bt[i * 3], b0 = bt[i * 3 + 1], c0 = bt[i * 3 + 2];
        const float4 a1 = bt[j * 3], b1 = bt[j * 3 + 1], c1 = bt[j * 3 + 2];
        leaf_triangle_test<IGN>(a0, b0, c0, o, d, best, ig0, ig1);
        leaf_triangle_test<IGN>(a1, b1, c1, o, d, best, ig0, ig1);
    }
#else
    for (int i = 0; i < cnt; ++i) leaf_triangle_test<IGN>(bt[i * 3], bt[i * 3 + 1], bt[i * 3 + 2], o, d, best, ig0, ig1);
#endif
}

#ifndef PSDR_TINY_UNROLL
#define PSDR_TINY_UNROLL 6
#endif
// Replaces __raygen__/__closesthit__/__miss__ (cuda/psdr_cuda.cu:9-45): closest hit with
// t in [RayEpsilon, tmax], both faces; (u,v) = barycentric weights of vertex 1 and 2.
// One primitive of a tiny scene (psdr_bvh_build.h pack_tiny_prims): a triangle or a parallelogram of two triangles, in PLANE FORM
// (tiny_plane_form): t from the plane equation, the plane coordinates (s, t) of the hit point from the dual basis of the two edges --
// 17 VALU operations with the rows in SGPRs, against 31 for Moeller-Trumbore (the primitive tests are half of the C2 kernels'
// arithmetic).  best.(u, v) hold the plane coordinates and best_i the index of the winning primitive; resolve_tiny_hit turns them into
// (triangle, u, v) once per ray.  BRANCH-FREE on purpose: a triangle is the parallelogram test with the bound s + t <= 1 instead of
// <= 2 (a wave-uniform select) and the hit update is four v_cndmask -- with `if (hit)` around it the ids / codes were re-fetched by
// scalar loads INSIDE the divergent branch and every primitive waited for its own s_load (three basic blocks per primitive: nothing
// could be scheduled across them); now the unrolled loop body is one block and the scalar loads of the following primitives are in
// flight while one is tested.
template <bool IGN = false>
PSDR_HD void tiny_prim_test(const float4 &r0, const float4 &r1, const float4 &r2, const float4 &r3, int i, const Vec3f &o, const Vec3f &d, Hit &best, int &best_i,
                            int ig0 = -1, int ig1 = -1) {
    const float dn = r0.x * d.x + (r0.y * d.y + r0.z * d.z);
    const float on = r0.x * o.x + (r0.y * o.y + (r0.z * o.z - r0.w));
    const float t = -on * (1.f / dn);
    const Vec3f p{o.x + t * d.x, o.y + t * d.y, o.z + t * d.z};
    // (u, v) = plane coordinates MINUS ONE HALF (the rows' constants carry the shift): 0 <= s <= 1 is |u| <= 1/2 -- one comparison with the free
    // |.| source modifier instead of two
    const float u = r1.x * p.x + (r1.y * p.y + (r1.z * p.z - r1.w));
    const float v = r2.x * p.x + (r2.y * p.y + (r2.z * p.z - r2.w));
    const float lim = r3.x;                                        // wave-uniform (kernel argument): bound on u + v -- 1 for a parallelogram (s + t <= 2), 0 for a triangle
    // a ray in the plane (dn = 0) gives t = +-inf or NaN, p and (u, v) NaN: every comparison fails
    bool hit = (fabsf(u) <= 0.5f) & (fabsf(v) <= 0.5f) & (u + v <= lim) & (t >= kRayEpsilon) & (t < best.t);
    if (IGN) { const int id2 = __float_as_int_hd(r3.y); const bool quad = lim > 0.5f; const int id = (quad && u + v > 0.f) ? (int) ((uint32_t) id2 >> 16) : (id2 & 0xffff); hit = hit & (id != ig0) & (id != ig1); }
    best.t = hit ? t : best.t; best.u = hit ? u : best.u; best.v = hit ? v : best.v; best_i = hit ? i : best_i;
}
// An AXIS-ALIGNED RECTANGLE (tiny_plane_form: the plane x_AX = c, in-plane axes in ascending order, rows (c, centre_a, centre_b, h_a),
// (h_b, rho | index, axis, ids)): the slab form -- t from ONE subtraction and the ray's reciprocal of that axis (three v_rcp_f32 per ray
// instead of one per primitive), the hit point's two in-plane coordinates relative to the rectangle's centre against its half extents.
// 15 VALU per test where the plane form takes 31 and a quarter-rate reciprocal.  best.(u, v) = the unscaled offsets; the winner's word
// (rho | index) tells resolve_tiny_hit which half of the pair they fall in, the hit rows in LDS carry the scale (setup_lds).
// d_AX = 0: inv = +-inf, t = +-inf or NaN (o on the plane) -- every comparison fails, as in the plane form.
template <int AX, bool IGN = false>
PSDR_HD void aa_prim_test(const float4 &ra, float hb, int packed, int id2, const Vec3f &o, const Vec3f &d, const Vec3f &inv, Hit &best, int &best_i, int ig0 = -1, int ig1 = -1) {
    const float on = AX == 0 ? o.x : (AX == 1 ? o.y : o.z), in = AX == 0 ? inv.x : (AX == 1 ? inv.y : inv.z);
    const float oa = AX == 0 ? o.y : o.x, da = AX == 0 ? d.y : d.x, ob = AX == 2 ? o.y : o.z, db = AX == 2 ? d.y : d.z;
    const float t = (ra.x - on) * in;
    const float u = (oa + t * da) - ra.y, v = (ob + t * db) - ra.z;
    bool hit = (fabsf(u) <= ra.w) & (fabsf(v) <= hb) & (t >= kRayEpsilon) & (t < best.t);
    if (IGN) { const int id = (u * __int_as_float_hd(packed) + v > 0.f) ? (int) ((uint32_t) id2 >> 16) : (id2 & 0xffff); hit = hit & (id != ig0) & (id != ig1); }
    // (four selects: `if (hit) { ... }` -- exec-masked moves -- measured 4 % slower on renderC, 15 % on the K = 3 duals: profiles/r04_aa_ifhit_ab.txt)
    best.t = hit ? t : best.t; best.u = hit ? u : best.u; best.v = hit ? v : best.v; best_i = hit ? packed : best_i;
}
// The triangle and its barycentrics from the winning primitive's plane coordinates.  Per primitive and half (the second triangle of a
// parallelogram is the half s + t > 1; a lone triangle fills both halves alike) eight words: the triangle id and the affine map
// (u, v) = (k0, k3) + (k1, k2 | k4, k5) (s - 1/2, t - 1/2), coefficients small integers and halves (tiny_hit_row decodes them from the
// 3-bit codes of pack_tiny_prims).  Device: the rows are staged in LDS by setup_lds, two 16-byte reads per ray (the decode itself was
// ~35 VALU instructions per ray); host: decoded on the spot.
constexpr int kTinyHitWords = 16;           // per primitive: two halves of (tri, k0, k1, k2, k3, k4, k5, -)
PSDR_HD void tiny_hit_row(const int32_t *meta, int half, int32_t &tri, float k[6], const float *S = nullptr) {
    const int ids = meta[0];
    const bool quad = ((uint32_t) ids >> 16) != 0xffffu, second = quad && half != 0;
    const int code = second ? meta[2] : meta[1];
    tri = second ? (int) ((uint32_t) ids >> 16) : (ids & 0xffff);
    float c[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) c[i] = (float) (((code >> (3 * i)) & 7) - 2);
    // tiny_prim_test keeps the plane coordinates shifted by one half: fold the shift into the constant
    k[0] = c[0] + 0.5f * (c[1] + c[2]); k[1] = c[1]; k[2] = c[2];
    k[3] = c[3] + 0.5f * (c[4] + c[5]); k[4] = c[4]; k[5] = c[5];
    if (S) {
        // an axis-aligned rectangle: best.(u, v) are world offsets from its centre, (s - 1/2, t - 1/2) = S (u', v')
        const float k1 = k[1], k2 = k[2], k4 = k[4], k5 = k[5];
        k[1] = k1 * S[0] + k2 * S[2]; k[2] = k1 * S[1] + k2 * S[3];
        k[4] = k4 * S[0] + k5 * S[2]; k[5] = k4 * S[1] + k5 * S[3];
    }
}
// best_i of closest_hit's primitive loops: a plane-form primitive leaves its row index (< kTinyRows), an axis-aligned rectangle its
// (rho | slot) word -- the bits of a float of normal magnitude, never a small integer
constexpr int kNoPrim = 63;
PSDR_HD void resolve_tiny_hit(const SceneView &sc, Hit &best, int packed) {
    if (packed == kNoPrim) return;                                  // no hit
    const bool plane = (uint32_t) packed < (uint32_t) kNoPrim;
    const float rho = plane ? 1.f : __int_as_float_hd(packed);      // the low four bits move rho by 2^-19 of itself: which triangle a point ON the diagonal belongs to
    const int best_i = plane ? packed : (packed & 15);
    const int half = best.u * rho + best.v > 0.f ? 1 : 0;
    float k[6];
#if defined(__HIP_DEVICE_COMPILE__)
    const float4 *row = reinterpret_cast<const float4 *>(psdr_dyn_lds + sc.off_lprim) + best_i * (kTinyHitWords / 4) + half * 2;
    const float4 a = row[0], b = row[1];
    best.tri = __float_as_int_hd(a.x);
    k[0] = a.y; k[1] = a.z; k[2] = a.w; k[3] = b.x; k[4] = b.y; k[5] = b.z;
#else
    tiny_hit_row(sc.tiny_meta + best_i * 4, half, best.tri, k, !plane ? &sc.tiny[best_i * 4 + 2].x : nullptr);
#endif
    const float u = best.u, v = best.v;
    best.u = k[0] + (k[1] * u + k[2] * v);
    best.v = k[3] + (k[4] * u + k[5] * v);
}

// Byte stride of a staged 64-byte node in LDS.  Every lane reads ITS node as four 16-byte words; rows 64 bytes apart put word c of all rows
// in two of the eight 16-byte bank groups (tools/micro/gather_rate.hip mode 4: 143 cycles per wave fetch), 80 bytes would spread them
// uniformly -- measured level on every tree workload (profiles/r03_lds_stride_ab.txt: the walk does not wait on these reads), so the
// rows stay packed and 25 % more of the tree fits.
#ifndef PSDR_LDS_NODE_STRIDE
#define PSDR_LDS_NODE_STRIDE 64
#endif
constexpr int kLdsNodeStride = PSDR_LDS_NODE_STRIDE;
// Tree walk from `root` (encoded like a child), keeping the closer hit in `best` (strict: the first of equal hits wins).
// "while-while" traversal: every lane first walks inner nodes until it holds a leaf (or is done), THEN
// the wave tests leaf triangles together -- the two phases have very different lengths, and in one
// merged loop lanes sitting at a leaf would idle through the others' node steps and vice versa.
template <bool IGN = false>
PSDR_HD void walk_tree(const SceneView &sc, TraversalStack &st, const Vec3f &o, const Vec3f &d, const Vec3f &inv, int32_t root, Hit &best, int ig0 = -1, int ig1 = -1) {
    int sp = 0;
    int32_t cur = root;
    constexpr int32_t kDone = 0x7fffffff;        // never a node index (node count < 2^28)
    while (cur != kDone) {
        while (cur >= 0 && cur != kDone) {
            BvhNode n;
#if defined(__HIP_DEVICE_COMPILE__)
            if (cur < sc.n_lnodes) n = *reinterpret_cast<const BvhNode *>(psdr_dyn_lds + sc.off_lnodes + cur * kLdsNodeStride);
            else
#endif
                n = sc.nodes[cur];
            float t0, t1;
            const bool h0 = slab(n.lo0, n.hi0, o, inv, best.t, t0), h1 = slab(n.lo1, n.hi1, o, inv, best.t, t1);
            if (h0 && h1) {
                const bool first0 = t0 <= t1;
                st.put(sp++, first0 ? n.c1 : n.c0);
                cur = first0 ? n.c0 : n.c1;
            } else if (h0 || h1) {
                cur = h0 ? n.c0 : n.c1;
            } else {
                cur = sp > 0 ? st.get(--sp) : kDone;
            }
        }
        if (cur == kDone) break;
        {
            const int enc = ~cur, first = enc >> 3, cnt = (enc & 7) + 1;
#if defined(__HIP_DEVICE_COMPILE__)
            const bool staged = first + cnt <= sc.n_lbtris;
            const float4 *lt = reinterpret_cast<const float4 *>(psdr_dyn_lds + sc.off_lbtris);
#endif
#if defined(__HIP_DEVICE_COMPILE__)
            if (staged) {
                for (int i = 0; i < cnt; ++i) leaf_triangle_test<IGN>(lt[(first + i) * 3], lt[(first + i) * 3 + 1], lt[(first + i) * 3 + 2], o, d, best, ig0, ig1);
            } else
#endif
                leaf_from_memory<IGN>(sc.btris + (size_t) first * 3, cnt, o, d, best, ig0, ig1);
            cur = sp > 0 ? st.get(--sp) : kDone;
        }
    }
}

// PSDR_WIDE_TREE (per translation unit: psdr_variant.hip sets it from the flag set, psdr_hip.hip uses 2): 1 = the kernels walk the 4-wide
// quantised tree, 0 = the BVH2, 2 = both walks in the kernel, chosen per scene (SceneView::nodes4 != nullptr)
#ifndef PSDR_WIDE_TREE
#define PSDR_WIDE_TREE 0
#endif
#if defined(__HIP_DEVICE_COMPILE__)
// The walk of the 4-wide tree (device).  Same while-while structure as walk_tree: inner nodes until the lane holds a leaf, then the leaf
// triangles.  Per node: one 64-byte fetch (LDS for the staged top of the tree), the slab test of the four children on the quantised
// planes -- t = q * (scale * inv) + (org - o) * inv: one v_cvt_f32_ubyteN and one v_fma per plane --, the hit children ordered by entry
// distance with a 5-exchange network on keys (entry distance bits | slot), the nearest visited next, the others pushed far-to-near.
template <bool IGN = false>
__device__ __forceinline__ void walk_tree4(const SceneView &sc, TraversalStack &st, const Vec3f &o, const Vec3f &d, const Vec3f &inv, int32_t root, Hit &best, int ig0 = -1, int ig1 = -1) {
    int sp = 0;
    int32_t cur = root;
    constexpr int32_t kDone = 0x7fffffff;
    while (cur != kDone) {
        while (cur >= 0 && cur != kDone) {
            Bvh4Node n;
            if (cur < sc.n_lnodes) n = *reinterpret_cast<const Bvh4Node *>(psdr_dyn_lds + sc.off_lnodes + cur * kLdsNodeStride);
            else n = sc.nodes4[cur];
            const float ax = __int_as_float_hd((int) ((n.exps & 0xffu) << 23)) * inv.x, ay = __int_as_float_hd((int) (((n.exps >> 8) & 0xffu) << 23)) * inv.y,
                        az = __int_as_float_hd((int) (((n.exps >> 16) & 0xffu) << 23)) * inv.z;
            const float bx = (n.org[0] - o.x) * inv.x, by = (n.org[1] - o.y) * inv.y, bz = (n.org[2] - o.z) * inv.z;
            // the ray enters through the lower plane of an axis it travels up along, through the upper plane otherwise: near / far planes
            // picked once per node (six selects on the packed plane words) instead of a min and a max per child and axis
            const bool px = inv.x >= 0.f, py = inv.y >= 0.f, pz = inv.z >= 0.f;
            const uint32_t nx = px ? n.qlo[0] : n.qhi[0], fx = px ? n.qhi[0] : n.qlo[0];
            const uint32_t ny = py ? n.qlo[1] : n.qhi[1], fy = py ? n.qhi[1] : n.qlo[1];
            const uint32_t nz = pz ? n.qlo[2] : n.qhi[2], fz = pz ? n.qhi[2] : n.qlo[2];
            uint32_t key[4]; int32_t ch[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float tnx = (float) ((nx >> (8 * c)) & 0xffu) * ax + bx, tfx = (float) ((fx >> (8 * c)) & 0xffu) * ax + bx;
                const float tny = (float) ((ny >> (8 * c)) & 0xffu) * ay + by, tfy = (float) ((fy >> (8 * c)) & 0xffu) * ay + by;
                const float tnz = (float) ((nz >> (8 * c)) & 0xffu) * az + bz, tfz = (float) ((fz >> (8 * c)) & 0xffu) * az + bz;
                // NaNs (0 * inf, inf - inf) drop out of fmin / fmax: that slab then does not cut -- conservative, as in slab()
                const float tn = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, 0.f));
                const float tf = fminf(fminf(tfx, tfy), fminf(tfz, best.t));
                const bool hit = tn <= tf && n.child[c] != kNoChild;
                key[c] = hit ? (uint32_t) __float_as_int_hd(tn) : 0xffffffffu;        // tn >= 0: the bit pattern orders like the value
                ch[c] = n.child[c];
            }
            // order the (entry distance, child) pairs: 5 exchanges
            auto cx = [&](int i, int j) {
                const bool sw = key[j] < key[i];
                const uint32_t ki = sw ? key[j] : key[i], kj = sw ? key[i] : key[j]; const int32_t ci = sw ? ch[j] : ch[i], cj = sw ? ch[i] : ch[j];
                key[i] = ki; key[j] = kj; ch[i] = ci; ch[j] = cj;
            };
            cx(0, 1); cx(2, 3); cx(0, 2); cx(1, 3); cx(1, 2);
            if (key[3] != 0xffffffffu) st.put(sp++, ch[3]);
            if (key[2] != 0xffffffffu) st.put(sp++, ch[2]);
            if (key[1] != 0xffffffffu) st.put(sp++, ch[1]);
            cur = key[0] != 0xffffffffu ? ch[0] : (sp > 0 ? st.get(--sp) : kDone);
        }
        if (cur == kDone) break;
        {
            const int enc = ~cur, first = enc >> 3, cnt = (enc & 7) + 1;
            const bool staged = first + cnt <= sc.n_lbtris;
            const float4 *lt = reinterpret_cast<const float4 *>(psdr_dyn_lds + sc.off_lbtris);
            if (staged) {
                for (int i = 0; i < cnt; ++i) leaf_triangle_test<IGN>(lt[(first + i) * 3], lt[(first + i) * 3 + 1], lt[(first + i) * 3 + 2], o, d, best, ig0, ig1);
            } else
                leaf_from_memory<IGN>(sc.btris + (size_t) first * 3, cnt, o, d, best, ig0, ig1);
            cur = sp > 0 ? st.get(--sp) : kDone;
        }
    }
}
#endif
// the tree walk of this build: the 4-wide tree on the device, the BVH2 on the host (tests/hostcheck, tools/simd_sim)
template <bool IGN = false>
PSDR_HD void walk(const SceneView &sc, TraversalStack &st, const Vec3f &o, const Vec3f &d, const Vec3f &inv, int32_t root2, int32_t root4, Hit &best, int ig0 = -1, int ig1 = -1) {
#if defined(__HIP_DEVICE_COMPILE__) && PSDR_WIDE_TREE == 1
    (void) root2; walk_tree4<IGN>(sc, st, o, d, inv, root4, best, ig0, ig1);
#elif defined(__HIP_DEVICE_COMPILE__) && PSDR_WIDE_TREE == 2
    // both walks in the kernel, chosen per scene (uniform): the 4-wide tree pays on large trees (50 k triangles: 5-8 %), not on a 5 k-triangle mesh
    if (sc.nodes4 != nullptr) walk_tree4<IGN>(sc, st, o, d, inv, root4, best, ig0, ig1);
    else walk_tree<IGN>(sc, st, o, d, inv, root2, best, ig0, ig1);
#else
    (void) root4; walk_tree<IGN>(sc, st, o, d, inv, root2, best, ig0, ig1);
#endif
}

// TWO rays from ONE origin through one walk (the two camera rays of a primary-edge sample: epsilon apart on the film, so they visit the same nodes).  The
// walks are bound by the latency of their node fetches, not by the slab arithmetic: the pair pays each fetch, each stack round trip and each loop
// iteration once and runs the slab / triangle tests of both rays on it.  A child is entered when either ray enters it (each ray prunes with its OWN closest
// hit, so a ray that does not reach a box fails every test below it); every triangle that either single walk would test is tested for that ray here, hence
// the same closest hit (ties between two triangles at the same distance may resolve in another order: the single walk orders children by its own ray).
#ifndef PSDR_PE_PAIR
#define PSDR_PE_PAIR 1
#endif
PSDR_HD void walk_tree_pair(const SceneView &sc, TraversalStack &st, const Vec3f &o, const Vec3f &dA, const Vec3f &invA, const Vec3f &dB, const Vec3f &invB, int32_t root,
                            Hit &bestA, Hit &bestB) {
    int sp = 0;
    int32_t cur = root;
    constexpr int32_t kDone = 0x7fffffff;
    while (cur != kDone) {
        while (cur >= 0 && cur != kDone) {
            BvhNode n;
#if defined(__HIP_DEVICE_COMPILE__)
            if (cur < sc.n_lnodes) n = *reinterpret_cast<const BvhNode *>(psdr_dyn_lds + sc.off_lnodes + cur * kLdsNodeStride);
            else
#endif
                n = sc.nodes[cur];
            float a0, a1, b0, b1;
            const bool hA0 = slab(n.lo0, n.hi0, o, invA, bestA.t, a0), hA1 = slab(n.lo1, n.hi1, o, invA, bestA.t, a1);
            const bool hB0 = slab(n.lo0, n.hi0, o, invB, bestB.t, b0), hB1 = slab(n.lo1, n.hi1, o, invB, bestB.t, b1);
            const bool h0 = hA0 || hB0, h1 = hA1 || hB1;
            if (h0 && h1) {
                const float t0 = fminf(hA0 ? a0 : INFINITY, hB0 ? b0 : INFINITY), t1 = fminf(hA1 ? a1 : INFINITY, hB1 ? b1 : INFINITY);
                const bool first0 = t0 <= t1;
                st.put(sp++, first0 ? n.c1 : n.c0);
                cur = first0 ? n.c0 : n.c1;
            } else if (h0 || h1) {
                cur = h0 ? n.c0 : n.c1;
            } else {
                cur = sp > 0 ? st.get(--sp) : kDone;
            }
        }
        if (cur == kDone) break;
        {
            const int enc = ~cur, first = enc >> 3, cnt = (enc & 7) + 1;
#if defined(__HIP_DEVICE_COMPILE__)
            if (first + cnt <= sc.n_lbtris) {                      // (two loops: an LDS pointer and a global one must not meet in one variable -- flat loads)
                const float4 *lt = reinterpret_cast<const float4 *>(psdr_dyn_lds + sc.off_lbtris) + first * 3;
                for (int i = 0; i < cnt; ++i) {
                    const float4 ta = lt[i * 3], tb = lt[i * 3 + 1], tc = lt[i * 3 + 2];
                    leaf_triangle_test<false>(ta, tb, tc, o, dA, bestA);
                    leaf_triangle_test<false>(ta, tb, tc, o, dB, bestB);
                }
            } else
#endif
            {
                const float4 *bt = sc.btris + (size_t) first * 3;
                for (int i = 0; i < cnt; ++i) {
                    const float4 ta = bt[i * 3], tb = bt[i * 3 + 1], tc = bt[i * 3 + 2];
                    leaf_triangle_test<false>(ta, tb, tc, o, dA, bestA);
                    leaf_triangle_test<false>(ta, tb, tc, o, dB, bestB);
                }
            }
            cur = sp > 0 ? st.get(--sp) : kDone;
        }
    }
}

PSDR_HD bool blas_box(const SceneView &sc, int k, const Vec3f &o, const Vec3f &inv, float tmax, float &t_entry) {
    const float lo[3] = {sc.blas_lo[k].x, sc.blas_lo[k].y, sc.blas_lo[k].z}, hi[3] = {sc.blas_hi[k].x, sc.blas_hi[k].y, sc.blas_hi[k].z};
    return slab(lo, hi, o, inv, tmax, t_entry);
}

// FOREST: 2 = the instance serves scenes WITHOUT a tree only (kSceneTiny: the walk below is not even compiled), 1 = two-level scenes only,
// 0 = never two-level (no box loop / per-tree walks in the code), -1 = decided at run time (k_trace, host tests).
// MASKED (kSceneTiny light rays, SceneView::occ): bit i of `rows` clear = row i of `tiny` cannot lie between the origin and the target.  `rows` is WAVE-UNIFORM
// (wave_or_hd of the lanes' entries: a row is tested when any lane wants it -- an extra test never changes a closest hit), so a skipped row costs one scalar
// bit test and the scalar loads of the tested ones stay uniform.
PSDR_HD uint32_t wave_or_hd(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    // one round per DISTINCT value among the active lanes (normally one: every lane names the light's row), all on the scalar unit
    unsigned long long todo = __ballot(1);
    uint32_t u = 0u;
    while (todo != 0ull) {
        const uint32_t f = (uint32_t) __builtin_amdgcn_readlane((int) x, __ffsll((long long) todo) - 1);
        u |= f;
        todo &= ~__ballot(x == f);
    }
    return u;
#else
    return x;
#endif
}
template <bool IGN = false, int FOREST = -1, bool MASKED = false>
PSDR_HD Hit closest_hit(const SceneView &sc, TraversalStack &st, const Vec3f &o, const Vec3f &d, float tmax, int ig0 = -1, int ig1 = -1, int pre_slot = 0, uint32_t rows = 0xffffffffu) {
    Hit best; best.tri = -1; best.u = best.v = -1.f;
    best.t = (tmax > 0.f && tmax < INFINITY) ? __int_as_float_hd(__float_as_int_hd(tmax) + 1) : tmax;   // accept t <= tmax
    const Vec3f inv{1.f / d.x, 1.f / d.y, 1.f / d.z};
    const bool forest = FOREST < 0 ? sc.n_blas > 0 : FOREST == 1;
    if (FOREST == 3 || FOREST == 2 || sc.n_tiny > 0 || forest) {
        // n_tiny PRIMITIVES (triangles, or parallelograms of two triangles: pack_tiny_prims), unrolled by 6 (the six
        // walls of the Cornell box): the scalar loads of the following primitives are in flight while one is tested
        int best_i = kNoPrim;
        // One s_load per primitive.  Device: the rows are read through the kernel-argument segment pointer itself -- SceneView is the first
        // member of the first argument of every kernel that traces (static_assert in psdr_host.h) -- because an index into the by-value struct
        // that the compiler cannot fold makes it copy the whole struct to scratch in some kernels (2 KB per lane; C4 PathTracer(3) renderC
        // 24 -> 54 ms when a second loop did that).
#if defined(__HIP_DEVICE_COMPILE__)
        typedef __attribute__((address_space(4))) const float4 kernarg_float4;
        const kernarg_float4 *prim_rows = (kernarg_float4 *) ((__attribute__((address_space(4))) const char *) __builtin_amdgcn_kernarg_segment_ptr() + offsetof(SceneView, tiny));
#else
        const float4 *prim_rows = sc.tiny;
#endif
        // the axis-aligned rectangles: kAaPerAxis slots per axis at fixed rows, each tested by the instance compiled for its axis behind ONE
        // wave-uniform branch on the axis' count.  (The compiler sinks each row's two scalar loads into its slot's branch; holding all rows in
        // SGPRs up front costs 25 more spilled SGPRs and 10 % of the kernel, requesting a row while its predecessor is tested changes nothing at
        // five waves per SIMD: profiles/r04_aa_slab_ab.txt)
        const int aa_cnt = sc.aa_cnt;
        if (aa_cnt != 0) {
            float4 ra[kAaSlots]; float hb[kAaSlots]; int pk[kAaSlots], id2[kAaSlots];
#pragma unroll
            for (int i = 0; i < kAaSlots; ++i) {
                ra[i] = prim_rows[i * 4];
                if (IGN) { const float4 r = prim_rows[i * 4 + 1]; hb[i] = r.x; pk[i] = __float_as_int_hd(r.y); id2[i] = __float_as_int_hd(r.w); }
                else {
#if defined(__HIP_DEVICE_COMPILE__)
                    const float2 r = *(__attribute__((address_space(4))) const float2 *) &prim_rows[i * 4 + 1];
#else
                    const float2 r{prim_rows[i * 4 + 1].x, prim_rows[i * 4 + 1].y};
#endif
                    hb[i] = r.x; pk[i] = __float_as_int_hd(r.y); id2[i] = 0;
                }
            }
            const int cx = aa_cnt & 255, cy = (aa_cnt >> 8) & 255, cz = aa_cnt >> 16;
#define PSDR_AA(AX, S) do { if (!MASKED || ((rows >> (S)) & 1u) != 0u) aa_prim_test<AX, IGN>(ra[S], hb[S], pk[S], id2[S], o, d, inv, best, best_i, ig0, ig1); } while (0)
            if (cx > 0) { PSDR_AA(0, 0); if (cx > 1) { PSDR_AA(0, 1); if (cx > 2) PSDR_AA(0, 2); } }
            if (cy > 0) { PSDR_AA(1, 3); if (cy > 1) { PSDR_AA(1, 4); if (cy > 2) PSDR_AA(1, 5); } }
            if (cz > 0) { PSDR_AA(2, 6); if (cz > 1) { PSDR_AA(2, 7); if (cz > 2) PSDR_AA(2, 8); } }
#undef PSDR_AA
        }
        // the other primitives in plane form (triangles, or parallelograms of two triangles: pack_tiny_prims)
#pragma unroll PSDR_TINY_UNROLL
        for (int i = aa_cnt != 0 ? kAaSlots : 0; i < sc.n_tiny; ++i)
            if (!MASKED || ((rows >> i) & 1u) != 0u) tiny_prim_test<IGN>(prim_rows[i * 4], prim_rows[i * 4 + 1], prim_rows[i * 4 + 2], prim_rows[i * 4 + 3], i, o, d, best, best_i, ig0, ig1);
        resolve_tiny_hit(sc, best, best_i);
        if (FOREST == 3) {
            // the trees were walked beforehand (same leaf test, tmax = infinity): the closest tree hit replaces the primitive hit exactly where the
            // walk below would have -- strictly closer
            const Hit ph = st.pre[pre_slot];
            if (ph.tri >= 0 && ph.t < best.t) best = ph;
            return best;
        }
        if (FOREST == 2 || !forest) return best;
        // two-level tree: the trees whose box the segment [0, t_best] enters, NEAREST box first (a hit in a near object
        // prunes the far ones); every round re-tests the remaining boxes against the current t_best -- wave-uniform
        // loops over SGPR operands, 16 VALU instructions per box
        if (sc.n_blas == 1) {
            float te;
            if (blas_box(sc, 0, o, inv, best.t, te)) walk<IGN>(sc, st, o, d, inv, __float_as_int_hd(sc.blas_lo[0].w), __float_as_int_hd(sc.blas_hi[0].w), best, ig0, ig1);
            return best;
        }
        uint32_t cand = (1u << sc.n_blas) - 1u;
        while (cand) {
            int32_t root = 0, root4 = 0; float near_t = INFINITY; uint32_t pick = 0;
            for (int k = 0; k < sc.n_blas; ++k) {
                if (!((cand >> k) & 1u)) continue;
                float te;
                if (!blas_box(sc, k, o, inv, best.t, te)) cand &= ~(1u << k);
                else if (te < near_t) { near_t = te; root = __float_as_int_hd(sc.blas_lo[k].w); root4 = __float_as_int_hd(sc.blas_hi[k].w); pick = 1u << k; }
            }
            if (!pick) break;
            cand &= ~pick;
            walk<IGN>(sc, st, o, d, inv, root, root4, best, ig0, ig1);
        }
        return best;
    }
    walk<IGN>(sc, st, o, d, inv, sc.root, sc.root4, best, ig0, ig1);
    return best;
}

// closest hits of two rays from one origin on a two-level scene whose kernels walk the BVH2 (pair_walk_ok): the kernel-argument primitives per ray (one
// copy of that loop), then every tree either ray's segment enters through walk_tree_pair.  The trees are taken in table order (the single-ray search takes
// the nearest box first): the closest hit does not depend on the order, only the pruning does, and these scenes hold one to a few trees.
template <int FLAGS> constexpr bool pair_walk_ok() { return PSDR_PE_PAIR != 0 && tree_mode<FLAGS>() == 1 && PSDR_WIDE_TREE == 0; }
PSDR_HD void closest_hit_pair(const SceneView &sc, TraversalStack &st, const Vec3f &o, const Vec3f &dA, const Vec3f &dB, Hit &hA, Hit &hB) {
#pragma unroll 1
    for (int r = 0; r < 2; ++r) {
        const Hit h = closest_hit<false, 2>(sc, st, o, r ? dB : dA, INFINITY);
        if (r) hB = h; else hA = h;
    }
    const Vec3f invA{1.f / dA.x, 1.f / dA.y, 1.f / dA.z}, invB{1.f / dB.x, 1.f / dB.y, 1.f / dB.z};
    for (int k = 0; k < sc.n_blas; ++k) {
        float te;
        const bool eA = blas_box(sc, k, o, invA, hA.t, te), eB = blas_box(sc, k, o, invB, hB.t, te);
        if (eA || eB) walk_tree_pair(sc, st, o, dA, invA, dB, invB, __float_as_int_hd(sc.blas_lo[k].w), hA, hB);
    }
}

// ------------------------------------------------------------------------ table loads
// Two scalar types run through the estimators:
//   G  geometry  (positions, directions, frames, hit records):  float or Dual<K>
//   M  material / radiance / throughput / result:               float or Dual<K>
// with G in {float, M}.  <float,float> = renderC; <Dual,Dual> = renderD with geometry tangents;
// <float,Dual> = renderD when only texels / emitter radiance carry tangents (albedo, roughness):
// the geometry then stays in plain fp32 registers.
template <class R> struct Loader;
template <> struct Loader<float> {
    template <int KK, int E> static PSDR_HD float f(const float *tab, const TangentView<KK, E> &, const float *const psdr_tangents::*, size_t i) { return tab[i]; }
};
template <int K> struct Loader<Dual<K>> {
    template <int E> static PSDR_HD Dual<K> f(const float *tab, const TangentView<K, E> &tv, const float *const psdr_tangents::*m, size_t i) {
        Dual<K> r; r.v = tab[i];
#pragma unroll
        for (int k = 0; k < K; ++k) { const float *p = tv.t[k].*m; r.d[k] = p ? p[i] : 0.f; }
        return r;
    }
};
template <class R, int FLAGS = kSceneRough> using TV = TangentView<ad_traits<R>::K, FLAGS>;
template <class R, class TVT> PSDR_HD R ldf(const float *tab, const TVT &tv, const float *const psdr_tangents::*m, size_t i) {
    return Loader<R>::f(tab, tv, m, i);
}
template <class R, class TVT> PSDR_HD Vec3<R> ld3(const float *tab, const TVT &tv, const float *const psdr_tangents::*m, size_t i) {
    return {ldf<R>(tab, tv, m, i), ldf<R>(tab, tv, m, i + 1), ldf<R>(tab, tv, m, i + 2)};
}
// G -> M promotion (identity, or float -> Dual with zero tangents)
template <class M> PSDR_HD M to_m(float x) { return M(x); }
template <class M, int K> PSDR_HD M to_m(const Dual<K> &x) { return x; }
template <class M, class G> PSDR_HD Vec3<M> to_m3(const Vec3<G> &a) { return {to_m<M>(a.x), to_m<M>(a.y), to_m<M>(a.z)}; }
// Vec3<M> * G-scalar when G = float and M = Dual is covered by the Vec3<Dual<K>> * float overload.

// TriangleInfo_ row (include/psdr/types.h:135-146), gathered by global triangle id (scene.cpp:300)
template <class R> struct TriRow { Vec3<R> p0, e1, e2, n0, n1, n2, fn; R area; };
template <class TVT> PSDR_HD TriRow<float> load_tri_f(const SceneView &sc, const TVT &, int id) {
    TriRow<float> t;
    float4 r[6];
#if defined(__HIP_DEVICE_COMPILE__)
    if (TVT::tiny || id < sc.n_ltri) {          // a kSceneTiny launch stages every row (make_ctx checks it)
        const float4 *q = reinterpret_cast<const float4 *>(psdr_dyn_lds + sc.off_ltri) + (size_t) id * 6;
#pragma unroll
        for (int i = 0; i < 6; ++i) r[i] = q[i];
    } else
#endif
    {
        const float4 *q = reinterpret_cast<const float4 *>(sc.d.tri_info) + (size_t) id * 6;
#pragma unroll
        for (int i = 0; i < 6; ++i) r[i] = q[i];
    }
    t.p0 = {r[0].x, r[0].y, r[0].z}; t.e1 = {r[0].w, r[1].x, r[1].y}; t.e2 = {r[1].z, r[1].w, r[2].x};
    t.n0 = {r[2].y, r[2].z, r[2].w}; t.n1 = {r[3].x, r[3].y, r[3].z}; t.n2 = {r[3].w, r[4].x, r[4].y};
    t.fn = {r[4].z, r[4].w, r[5].x}; t.area = r[5].y;
    return t;
}
#ifndef PSDR_TANGENT_LIVE
#define PSDR_TANGENT_LIVE 1
#endif
template <class R, class TVT> PSDR_HD TriRow<R> load_tri(const SceneView &sc, const TVT &tv, int id) {
    if constexpr (!is_ad<R>()) return load_tri_f(sc, tv, id);
    if constexpr (is_ad<R>() && PSDR_TANGENT_LIVE) {
        if (tv.live != nullptr && ((tv.live[id >> 5] >> (id & 31)) & 1u) == 0u) {
            const TriRow<float> f = load_tri_f(sc, tv, id);          // a row at rest: six 16-byte loads, zero tangents
            TriRow<R> t;
            t.p0 = lift<R>(f.p0); t.e1 = lift<R>(f.e1); t.e2 = lift<R>(f.e2); t.n0 = lift<R>(f.n0); t.n1 = lift<R>(f.n1); t.n2 = lift<R>(f.n2);
            t.fn = lift<R>(f.fn); t.area = R(f.area);
            return t;
        }
    }
    const size_t o = (size_t) id * PSDR_TRI_STRIDE;
    const float *a = sc.d.tri_info;
    constexpr auto m = &psdr_tangents::d_tri_info;
    TriRow<R> t;
    t.p0 = ld3<R>(a, tv, m, o); t.e1 = ld3<R>(a, tv, m, o + 3); t.e2 = ld3<R>(a, tv, m, o + 6);
    t.n0 = ld3<R>(a, tv, m, o + 9); t.n1 = ld3<R>(a, tv, m, o + 12); t.n2 = ld3<R>(a, tv, m, o + 15);
    t.fn = ld3<R>(a, tv, m, o + 18); t.area = ldf<R>(a, tv, m, o + 21);
    return t;
}

// ---------------------------------------------------------------------- intersection
// Intersection_ (include/psdr/core/intersection.h:24-52)
template <class R> struct Its {
    bool valid;
    int tri, mesh;
    Vec3<R> wi, p, n;
    R t, J, uvx, uvy;
    Frame<R> sh;
    float hu, hv;   // traversal barycentrics (detached), kept for the reverse pass
};
template <class R> struct RayT { Vec3<R> o, d; };

// ray_intersect_triangle (include/psdr/utils.h:66-77), no range tests
template <class R> PSDR_HD void moeller_trumbore(const Vec3<R> &p0, const Vec3<R> &e1, const Vec3<R> &e2, const RayT<R> &ray,
                                                 R &u, R &v, R &t) {
    const Vec3<R> h = cross(ray.d, e2);
    const R f = 1.f / dot(e1, h);
    const Vec3<R> s = ray.o - p0;
    u = f * dot(s, h);
    const Vec3<R> q = cross(s, e1);
    v = f * dot(ray.d, q);
    t = f * dot(e2, q);
}

enum HitForm { kDetached = 0, kPathSpace = 1, kSolidAngle = 2 };
template <class R, class TVT> PSDR_HD void fill_its_from_hit(Its<R> &its, const SceneView &sc, const TVT &tv, const Hit &h, const RayT<R> &ray, HitForm form);

// Scene::ray_intersect<ad, path_space> (src/scene/scene.cpp:290-384)
//   kDetached  : C types; barycentrics from the traversal, J = 1
//   kPathSpace : D types, barycentrics DETACHED (point rides on the moving triangle), J = A/detach(A)
//   kSolidAngle: D types, differentiable Moeller-Trumbore on the chosen triangle, J = 1
// ig0 / ig1 >= 0: triangles the ray must not hit (rays that start on a secondary edge, psdr_scene_desc::sec_edge_faces).
template <class R, class TVT, bool MASKED = false> PSDR_HD Its<R> intersect(const SceneView &sc, const TVT &tv, TraversalStack &st, const RayT<R> &ray,
                                                       bool active, HitForm form, uint32_t &nrays, int ig0 = -1, int ig1 = -1, int pre_slot = 0, uint32_t rows = 0xffffffffu) {
    Its<R> its;
    its.valid = false; its.tri = its.mesh = -1; its.J = R(1.f); its.t = R(INFINITY);
    if (!active) return its;
    nrays++;
    constexpr int F = tree_mode<TVT::flags>();
    Hit h;
    if constexpr (MASKED) h = closest_hit<false, F, true>(sc, st, val(ray.o), val(ray.d), INFINITY, -1, -1, pre_slot, wave_or_hd(rows));
    else h = (ig0 >= 0 || ig1 >= 0) ? closest_hit<true, F>(sc, st, val(ray.o), val(ray.d), INFINITY, ig0, ig1, pre_slot)
                                    : closest_hit<false, F>(sc, st, val(ray.o), val(ray.d), INFINITY, -1, -1, pre_slot);
    if (h.tri < 0) return its;
    fill_its_from_hit<R>(its, sc, tv, h, ray, form);
    return its;
}
// the rows of SceneView::tiny a light ray from triangle tri_r to emitter triangle tri_e tests (SceneView::occ)
PSDR_HD uint32_t occ_rows(const SceneView &sc, int tri_r, int tri_e) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (sc.lt_occ >= 0 && tri_e >= 0) ? PSDR_LDS_TABLE(uint32_t, sc.lt_occ)[tri_r * sc.d.num_tris + tri_e] : 0xffffffffu;
#else
    return (sc.occ != nullptr && tri_e >= 0) ? sc.occ[(size_t) tri_r * sc.d.num_tris + tri_e] : 0xffffffffu;
#endif
}
// The hit record of a KNOWN hit (triangle + traversal barycentrics): everything intersect() derives behind its closest_hit.  The geometry-dual
// stages of the traced wavefront rebuild a path vertex with it from its stream record -- the same arithmetic, hence the same vertex and tangents,
// as the fused kernel's `its = nits`.  (intersect() fills ITS OWN record through fill_its_from_hit: with a second record returned by value the
// compiler kept parts of it in scratch -- 21 scratch instructions in the C2 renderC kernel, +2.4 % instructions, 4x the counter traffic.)
template <class R, class TVT> PSDR_HD Its<R> its_from_hit(const SceneView &sc, const TVT &tv, const Hit &h, const RayT<R> &ray, HitForm form) {
    Its<R> its;
    its.J = R(1.f); its.t = R(INFINITY);
    fill_its_from_hit<R>(its, sc, tv, h, ray, form);
    return its;
}
template <class R, class TVT> PSDR_HD void fill_its_from_hit(Its<R> &its, const SceneView &sc, const TVT &tv, const Hit &h, const RayT<R> &ray, HitForm form) {
    its.valid = true; its.tri = h.tri; its.hu = h.u; its.hv = h.v;
    const int tm = Tab<TVT::flags>::tri_mesh(sc, h.tri);
    its.mesh = tm & ~PSDR_TRI_FACE_NORMALS;
    const TriRow<R> T = load_tri<R>(sc, tv, h.tri);
    its.n = T.fn;
    R bu, bv;
    if (form != kSolidAngle) {
        bu = R(h.u); bv = R(h.v);
        if (is_ad<R>() && form == kPathSpace) its.J = T.area / detach(T.area);
        its.p = bary_point(T.p0, T.e1, T.e2, bu, bv);
        Vec3<R> dir = its.p - ray.o;
        its.t = norm(dir);
        dir = dir / its.t;
        Vec3<R> sh_n = (tm & PSDR_TRI_FACE_NORMALS) ? its.n : normalize(bary_point(T.n0, T.n1 - T.n0, T.n2 - T.n0, bu, bv));
        its.sh = Frame<R>(sh_n);
        its.wi = its.sh.to_local(-dir);
    } else {
        R t;
        moeller_trumbore(T.p0, T.e1, T.e2, ray, bu, bv, t);
        Vec3<R> sh_n = (tm & PSDR_TRI_FACE_NORMALS) ? its.n : normalize(bary_point(T.n0, T.n1 - T.n0, T.n2 - T.n0, bu, bv));
        // The reference writes p = ray(t) (scene.cpp:368).  Same point, same derivative, evaluated ON the triangle:
        // o + t d sits up to |o - p| * 2^-22 off the surface (4e-4 at the bunny scenes' 400 units), and a grazing
        // continuation ray that starts below it re-hits its own face above RayEpsilon = 1e-3 -- a one-sided loss of
        // ~2e-3 of the interior gradient in ANY fp32 evaluation of the literal form (fp32 against fp64,
        // tests/test_projections_gpu.py); with the point on the surface fp32 agrees with fp64 to ~1e-4.
        its.p = sc.literal_forms ? ray.o + ray.d * t : bary_point(T.p0, T.e1, T.e2, bu, bv);        // literal: scene.cpp:368
        its.t = t;
        its.sh = Frame<R>(sh_n);
        its.wi = its.sh.to_local(-ray.d);
    }
    if (sc.d.tri_uv) {
        const float *q = Tab<TVT::flags>::tri_uv(sc, h.tri);
        its.uvx = (q[2] - q[0]) * bu + ((q[4] - q[0]) * bv + q[0]);
        its.uvy = (q[3] - q[1]) * bu + ((q[5] - q[1]) * bv + q[1]);
    } else { its.uvx = R(0.f); its.uvy = R(0.f); }
}

// Path vertex rebuilt from its stream record (wavefront mode): triangle id, detached barycentrics and
// the unit direction of arrival.  Same arithmetic as the path-space branch of intersect().
template <class TVT> PSDR_HD Its<float> path_vertex_from_record(const SceneView &sc, const TVT &tv, int tri, float hu, float hv, const Vec3f &dir) {
    Its<float> its;
    its.valid = true; its.tri = tri; its.hu = hu; its.hv = hv; its.J = 1.f; its.t = 0.f;
    // (measured and dropped: the texture-coordinate row fetched WITH the triangle row instead of after it -- one dependent round trip less in the
    // bounce stage's chain, 1 344.7 = 1 344.3 us per C4 stage: the stage does not wait for a single trip, profiles/r04_flat_ab.txt)
    const int tm = Tab<TVT::flags>::tri_mesh(sc, tri);
    its.mesh = tm & ~PSDR_TRI_FACE_NORMALS;
    const TriRow<float> T = load_tri_f(sc, tv, tri);
    its.n = T.fn;
    its.p = bary_point(T.p0, T.e1, T.e2, hu, hv);
    const Vec3f sh_n = (tm & PSDR_TRI_FACE_NORMALS) ? its.n : normalize(bary_point(T.n0, T.n1 - T.n0, T.n2 - T.n0, hu, hv));
    its.sh = Frame<float>(sh_n);
    its.wi = its.sh.to_local(-dir);
    if (sc.d.tri_uv) {
        const float *q = Tab<TVT::flags>::tri_uv(sc, tri);
        its.uvx = (q[2] - q[0]) * hu + ((q[4] - q[0]) * hv + q[0]);
        its.uvy = (q[3] - q[1]) * hu + ((q[5] - q[1]) * hv + q[1]);
    } else { its.uvx = 0.f; its.uvy = 0.f; }
    return its;
}

template <class R, class TVT> PSDR_HD int emitter_of(const SceneView &sc, const TVT &, const Its<R> &its) { return its.valid ? Tab<TVT::flags>::mesh_emitter(sc, its.mesh) : -1; }

template <class M, class TVT> PSDR_HD Vec3<M> radiance(const SceneView &sc, const TVT &tv, int e) {
    const float *f = Tab<TVT::flags>::emitter_f(sc, e);
    Vec3<M> r = {M(f[0]), M(f[1]), M(f[2])};
    if constexpr (is_ad<M>()) {
#pragma unroll
        for (int k = 0; k < ad_traits<M>::K; ++k) {
            const float *p = tv.t[k].d_emitter_rad;
            if (p) { r.x.d[k] = p[e * 3]; r.y.d[k] = p[e * 3 + 1]; r.z.d[k] = p[e * 3 + 2]; }
        }
    }
    return r;
}
// ------------------------------------------------------------------------------ BSDF
// Bitmap<c>::eval (src/core/bitmap.cpp:41-89); 3-channel textures are stored interleaved RGB.
// U = type of the texture coordinates (geometry), M = type of the texels.
// One texel (value + tangents).  LDS = true: from the staged copy of the pool, [value pool | K tangent pools] (kSceneTiny launches
// whose pool is small, SceneView::lt_tex >= 0; the kernel stages the tangent pools itself: stage_tangent_texels).
template <class M, bool LDS, class TVT> PSDR_HD M texel(const SceneView &sc, const TVT &tv, size_t i) {
    if constexpr (!LDS) return ldf<M>(sc.d.texels, tv, &psdr_tangents::d_texels, i);
    else {
        const float *lx = PSDR_LDS_TABLE(float, sc.lt_tex);
        if constexpr (!is_ad<M>()) return lx[i];
        else {
            M r; r.v = lx[i];
#pragma unroll
            for (int k = 0; k < ad_traits<M>::K; ++k) r.d[k] = lx[(size_t) (k + 1) * sc.d.num_texels + i];
            return r;
        }
    }
}
template <class M, int C, bool LDS, class U, class TVT>
PSDR_HD void bitmap_eval_from(const SceneView &sc, const TVT &tv, const int32_t *slot, U u, U v, M *out, bool flip_v) {
    const int off = slot[0], w = slot[1], h = slot[2];
    if (w == 1 && h == 1) {
#pragma unroll
        for (int c = 0; c < C; ++c) out[c] = texel<M, LDS>(sc, tv, off + c);
        return;
    }
    if (flip_v) v = -v;
    u = u - floorf(val(u)); v = v - floorf(val(v));
    u = u * (float) (w - 1); v = v * (float) (h - 1);
    int px = (int) floorf(val(u)), py = (int) floorf(val(v));
    const U w1x = u - (float) px, w1y = v - (float) py;
    const U w0x = 1.f - w1x, w0y = 1.f - w1y;
    px = px < w - 2 ? px : w - 2; py = py < h - 2 ? py : h - 2;
    const size_t idx = (size_t) py * w + px;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const M v00 = texel<M, LDS>(sc, tv, off + idx * C + c), v10 = texel<M, LDS>(sc, tv, off + (idx + 1) * C + c);
        const M v01 = texel<M, LDS>(sc, tv, off + (idx + w) * C + c), v11 = texel<M, LDS>(sc, tv, off + (idx + w + 1) * C + c);
        out[c] = (v00 * w0x + v10 * w1x) * w0y + (v01 * w0x + v11 * w1x) * w1y;
    }
}
template <class M, int C, class U, class TVT>
PSDR_HD void bitmap_eval(const SceneView &sc, const TVT &tv, const int32_t *slot, U u, U v, M *out, bool flip_v = true) {
    if constexpr (Tab<TVT::flags>::lds_small) {
        if (sc.lt_tex >= 0) { bitmap_eval_from<M, C, true>(sc, tv, slot, u, v, out, flip_v); return; }
    }
    bitmap_eval_from<M, C, false>(sc, tv, slot, u, v, out, flip_v);
}

// ------------------------------------------------------------------- environment map
constexpr float kInvTwoPi = 0.15915494309189533577f;
// EnvironmentMap::eval_direction (src/emitter/envmap.cpp:41-59): lat-long lookup along the world
// direction w.  Tangents: the direction (geometry), m_from_world / m_scale (d_env_f), the texels.
template <class M, class G, class TVT> PSDR_HD Vec3<M> env_eval_direction(const SceneView &sc, const TVT &tv, const Vec3<G> &w) {
    const float *f = sc.d.env_f;
    constexpr auto m = &psdr_tangents::d_env_f;
    const Vec3<M> wm = to_m3<M>(w);
    auto row = [&](int r) {
        return ldf<M>(f, tv, m, PSDR_ENV_FROM_WORLD + r * 3) * wm.x + ldf<M>(f, tv, m, PSDR_ENV_FROM_WORLD + r * 3 + 1) * wm.y +
               ldf<M>(f, tv, m, PSDR_ENV_FROM_WORLD + r * 3 + 2) * wm.z;
    };
    const M vx = row(0), vy = row(1), vz = row(2);
    M u = atan2_(vx, -vz) * kInvTwoPi, v = safe_acos_(vy) * kInvPi;
    u = u - floorf(val(u)); v = v - floorf(val(v));
    M rgb[3];
    bitmap_eval<M, 3>(sc, tv, sc.d.env_tex, u, v, rgb, false);
    const M scale = ldf<M>(f, tv, m, PSDR_ENV_SCALE);
    return {rgb[0] * scale, rgb[1] * scale, rgb[2] * scale};
}
// Intersection::Le -> AreaLight::eval (src/emitter/area.cpp:20-29) / EnvironmentMap::eval (envmap.cpp:29-38)
template <class M, class G, class TVT> PSDR_HD Vec3<M> Le(const SceneView &sc, const TVT &tv, const Its<G> &its, bool active) {
    const int e = active ? emitter_of(sc, tv, its) : -1;
    if (e < 0) return zero3<M>();
    if (TVT::has_env && e == sc.d.env_emitter) return env_eval_direction<M>(sc, tv, -its.sh.to_world(its.wi));
    if (!(val(its.wi.z) > 0.f)) return zero3<M>();
    return radiance<M>(sc, tv, e);
}


// GGXDistribution (src/bsdf/ggx.cpp:9-106)
template <class R> struct GGX {
    R au, av;
    PSDR_HD R eval(const Vec3<R> &m) const {
        const R r = 1.f / (kPi * au * av * sqr(sqr(m.x / au) + sqr(m.y / av) + sqr(m.z)));
        return val(r) * val(m.z) > 1e-5f ? r : R(0.f);
    }
    PSDR_HD R smith_g1(const Vec3<R> &v, const Vec3<R> &m) const {
        const R xy = sqr(au * v.x) + sqr(av * v.y);
        R r = 2.f / (1.f + sqrt_(1.f + xy / sqr(v.z)));
        if (val(xy) == 0.f) r = R(1.f);
        if (val(dot(v, m)) * val(v.z) <= 0.f) r = R(0.f);
        return r;
    }
    // sample_visible_11, ggx.cpp:96-106
    PSDR_HD void visible11(const R &cos_i, float sx, float sy, R &ox, R &oy) const {
        float px, py;
        concentric_disk(sx, sy, px, py);
        const R s = 0.5f * (1.f + cos_i);
        const R y = sqrtf(fmaxf(1.f - px * px, 0.f)) * (1.f - s) + py * s;
        const R z = safe_sqrt(1.f - (px * px + y * y));
        const R sin_i = safe_sqrt(1.f - sqr(cos_i));
        const R nrm = 1.f / (sin_i * y + cos_i * z);
        ox = (cos_i * y - sin_i * z) * nrm; oy = px * nrm;
    }
    // ggx.cpp:34-76 with Frame::sin_phi / cos_phi (frame.h:100-116)
    PSDR_HD Vec3<R> sample(const Vec3<R> &wi, float sx, float sy) const {
        const Vec3<R> wp = normalize(Vec3<R>(au * wi.x, av * wi.y, wi.z));
        const R st2 = wp.x * wp.x + wp.y * wp.y;
        R sin_phi(0.f), cos_phi(1.f);
        if (!(fabsf(val(st2)) <= 4.f * kEpsilon)) {
            const R inv = 1.f / sqrt_(st2);
            sin_phi = clamp_(wp.y * inv, -1.f, 1.f); cos_phi = clamp_(wp.x * inv, -1.f, 1.f);
        }
        R slx, sly;
        visible11(wp.z, sx, sy, slx, sly);
        const R s0 = (cos_phi * slx - sin_phi * sly) * au, s1 = (sin_phi * slx + cos_phi * sly) * av;
        return normalize(Vec3<R>(-s0, -s1, R(1.f)));
    }
};

// conductor Fresnel (include/psdr/utils.h:148-164), one channel
template <class R> PSDR_HD R fresnel_conductor(const R &eta, const R &k, const R &cos_i) {
    const R c2 = sqr(cos_i), s2 = 1.f - c2, s4 = sqr(s2);
    const R t1 = sqr(eta) - sqr(k) - s2;
    const R a2pb2 = safe_sqrt(sqr(t1) + 4.f * sqr(k * eta));
    const R a = safe_sqrt(0.5f * (a2pb2 + t1));
    const R T1 = a2pb2 + c2, T2 = 2.f * cos_i * a;
    const R rs = (T1 - T2) / (T1 + T2);
    const R T3 = a2pb2 * c2 + s4, T4 = T2 * s2;
    return 0.5f * (rs + rs * (T3 - T4) / (T3 + T4));
}

// BSDF evaluated at a hit of geometry type G with parameters of type M; all results are M.
// The material parameters of ONE vertex, looked up once (reverse mode: sample / eval / pdf and their adjoints would fetch the
// five textures of a rough conductor ~34 times per vertex, with the gradient adds in between keeping the compiler from merging them)
template <class M> struct MatCache { M au, av; Vec3<M> eta, k, refl; };
template <class G, class M> struct Bsdf {
    const int32_t *rec;
    const MatCache<M> *mc = nullptr;       // set: tex1 / tex3 answer from it
    PSDR_HD Bsdf(const SceneView &sc, int id) : rec(sc.d.bsdf_rec + (size_t) (id < 0 ? 0 : id) * PSDR_BSDF_STRIDE) {}
    template <class TVT> PSDR_HD Bsdf(const SceneView &sc, const TVT &, int id) : rec(Tab<TVT::flags>::bsdf_rec(sc, id)) {}
    template <class TVT> PSDR_HD MatCache<M> fetch(const SceneView &sc, const TVT &tv, const Its<G> &its) const {
        MatCache<M> c;
        c.refl = tex3(sc, tv, PSDR_SLOT_REFLECTANCE, its);
        if (is_diffuse(tv)) { c.au = c.av = M(0.f); c.eta = c.k = zero3<M>(); return c; }
        c.au = tex1(sc, tv, PSDR_SLOT_ALPHA_U, its); c.av = tex1(sc, tv, PSDR_SLOT_ALPHA_V, its);
        c.eta = tex3(sc, tv, PSDR_SLOT_ETA, its); c.k = tex3(sc, tv, PSDR_SLOT_K, its);
        return c;
    }
    PSDR_HD int type() const { return rec[0]; }
    // diffuse unless the kernel instance carries the rough-conductor code (TangentView FLAGS)
    template <class TVT> PSDR_HD bool is_diffuse(const TVT &) const { return !TVT::has_rough || rec[0] == PSDR_BSDF_DIFFUSE; }
    PSDR_HD const int32_t *slot(int s) const { return rec + 1 + 3 * s; }
    template <class TVT> PSDR_HD Vec3<M> tex3(const SceneView &sc, const TVT &tv, int s, const Its<G> &its) const {
        if (mc) return s == PSDR_SLOT_ETA ? mc->eta : (s == PSDR_SLOT_K ? mc->k : mc->refl);
        M o[3]; bitmap_eval<M, 3>(sc, tv, slot(s), its.uvx, its.uvy, o); return {o[0], o[1], o[2]};
    }
    template <class TVT> PSDR_HD M tex1(const SceneView &sc, const TVT &tv, int s, const Its<G> &its) const {
        if (mc) return s == PSDR_SLOT_ALPHA_U ? mc->au : mc->av;
        M o[1]; bitmap_eval<M, 1>(sc, tv, slot(s), its.uvx, its.uvy, o); return o[0];
    }
    // Diffuse::__eval (diffuse.cpp:25-35) / RoughConductor::__eval (roughconductor.cpp:40-58); value = f * cos(theta_o)
    template <class TVT> PSDR_HD Vec3<M> eval(const SceneView &sc, const TVT &tv, const Its<G> &its, const Vec3<G> &wo, bool active) const {
        if (!(active && val(its.wi.z) > 0.f && val(wo.z) > 0.f)) return zero3<M>();
        if (is_diffuse(tv)) return tex3(sc, tv, PSDR_SLOT_REFLECTANCE, its) * (wo.z * kInvPi);
        const GGX<M> g{tex1(sc, tv, PSDR_SLOT_ALPHA_U, its), tex1(sc, tv, PSDR_SLOT_ALPHA_V, its)};
        const Vec3<M> wi_m = to_m3<M>(its.wi), wo_m = to_m3<M>(wo);
        const Vec3<M> H = normalize(wo_m + wi_m);
        const M D = g.eval(H);
        if (val(D) == 0.f) return zero3<M>();
        const M res = D * (g.smith_g1(wi_m, H) * g.smith_g1(wo_m, H)) / (4.f * wi_m.z);
        const Vec3<M> eta = tex3(sc, tv, PSDR_SLOT_ETA, its), k = tex3(sc, tv, PSDR_SLOT_K, its);
        const M c = dot(wi_m, H);
        const Vec3<M> F{fresnel_conductor(eta.x, k.x, c), fresnel_conductor(eta.y, k.y, c), fresnel_conductor(eta.z, k.z, c)};
        return F * res * tex3(sc, tv, PSDR_SLOT_REFLECTANCE, its);
    }
    // Diffuse::__pdf (diffuse.cpp:70-81: detached) / RoughConductor::__pdf (roughconductor.cpp:61-75: mask not applied)
    template <class TVT> PSDR_HD M pdf(const SceneView &sc, const TVT &tv, const Its<G> &its, const Vec3<G> &wo, bool active) const {
        if (is_diffuse(tv)) {
            const float ci = val(its.wi.z), co = val(wo.z);
            return M((active && ci > 0.f && co > 0.f) ? kInvPi * co : 0.f);
        }
        const Vec3<M> wi_m = to_m3<M>(its.wi);
        const Vec3<M> m = normalize(to_m3<M>(wo) + wi_m);
        const GGX<M> g{tex1(sc, tv, PSDR_SLOT_ALPHA_U, its), tex1(sc, tv, PSDR_SLOT_ALPHA_V, its)};
        return g.eval(m) * g.smith_g1(wi_m, m) / (4.f * wi_m.z);
    }
    // Diffuse::__sample (diffuse.cpp:48-57, uses tail<2>) / RoughConductor::__sample (roughconductor.cpp:78-92).
    // Returns the sampled direction as plain floats (it only steers the traced ray; in D mode the
    // BSDF is re-evaluated from the hit points) and its pdf as M (GGX: carries d/d(alpha, wi)).
    template <class TVT> PSDR_HD bool sample(const SceneView &sc, const TVT &tv, const Its<G> &its, const float s[3], bool active, Vec3f &wo,
                                             M &pdf_) const {
        if (is_diffuse(tv)) {
            wo = cosine_hemisphere(s[1], s[2]);
            pdf_ = M(kInvPi * wo.z);
            return active && val(its.wi.z) > 0.f;
        }
        const GGX<M> g{tex1(sc, tv, PSDR_SLOT_ALPHA_U, its), tex1(sc, tv, PSDR_SLOT_ALPHA_V, its)};
        const Vec3<M> wi_m = to_m3<M>(its.wi);
        const Vec3<M> m = g.sample(wi_m, s[0], s[1]);
        const Vec3<M> wo_m = m * (2.f * dot(wi_m, m)) - wi_m;
        const Vec3<M> h = normalize(wo_m + wi_m);
        pdf_ = g.eval(h) * g.smith_g1(wi_m, h) / (4.f * wi_m.z);
        wo = val(wo_m);
        return active && val(its.wi.z) > 0.f && val(pdf_) != 0.f && wo.z > 0.f;
    }
};

// -------------------------------------------------------------------------- emitters
template <class R> struct PosSample { Vec3<R> p, n; R J; float pdf; bool valid; int tri; };          // tri: the emitter triangle the point lies on (-1: the environment map)

// HyperCubeDistribution<2>::sample_reuse / pdf (src/core/cube_distrb.cpp:42-62) of the env-map cells;
// cell (x, y) has index x * reso[1] + y (cube_distrb.cpp:19-26)
PSDR_HD float env_cells_sample_reuse(const SceneView &sc, float &u0, float &u1) {
    const int r0 = sc.d.env_reso[0], r1 = sc.d.env_reso[1], n = r0 * r1;
    float pmf;
    const int idx = sample_reuse(sc.d.env_cmf, sc.d.env_pmf, sc.d.env_sum, n, u1, pmf);
    const int c0 = idx / r1, c1 = idx - c0 * r1;
    u0 = (u0 + (float) c0) * (1.f / (float) r0);
    u1 = (u1 + (float) c1) * (1.f / (float) r1);
    return pmf * (float) n;
}
PSDR_HD float env_cells_pdf(const SceneView &sc, float u0, float u1) {
    const int r0 = sc.d.env_reso[0], r1 = sc.d.env_reso[1];
    const int i0 = (int) floorf(u0 * (float) r0), i1 = (int) floorf(u1 * (float) r1);
    if (!(i0 >= 0 && i0 < r0 && i1 >= 0 && i1 < r1)) return 0.f;
    return sc.d.env_pmf[i0 * r1 + i1] / sc.d.env_sum * (float) (r0 * r1);
}
PSDR_HD Vec3f env_mul3(const float *m, const Vec3f &v) {      // transform_dir, transform.h:91-94
    return {m[0] * v.x + m[1] * v.y + m[2] * v.z, m[3] * v.x + m[4] * v.y + m[5] * v.z, m[6] * v.x + m[7] * v.y + m[8] * v.z};
}
// EnvironmentMap::__sample_position (envmap.cpp:72-95): direction from the luminance cells
// (sample_direction, :98-111), projected on the scene AABB (ray_intersect_scene_aabb, utils.h:128-145).
// Everything is detached in the reference, J = 1.
template <class R> PSDR_HD PosSample<R> env_sample_position(const SceneView &sc, const Vec3f &ref_p, float s0, float s1) {
    const float *f = sc.d.env_f;
    float pdf = env_cells_sample_reuse(sc, s0, s1);
    const float theta = s1 * kPi, phi = s0 * (2.f * kPi);
    float st, ct, sp, cp;
    sincosf(theta, &st, &ct); sincosf(phi, &sp, &cp);
    Vec3f d{sp * st, ct, -cp * st};                          // sphdir(theta, phi) -> (y, z, -x)
    const float inv_sin_theta = 1.f / sqrtf(fmaxf(d.x * d.x + d.z * d.z, kEpsilon * kEpsilon));
    if (pdf > kEpsilon) pdf *= inv_sin_theta * (.5f / (kPi * kPi));
    d = env_mul3(f + PSDR_ENV_TO_WORLD, d);
    const float o[3] = {ref_p.x, ref_p.y, ref_p.z}, dd[3] = {d.x, d.y, d.z};
    float t = 0.f; int ax = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float t1 = (f[PSDR_ENV_LOWER + i] - o[i]) / dd[i], t2 = (f[PSDR_ENV_UPPER + i] - o[i]) / dd[i];
        const float tp = fmaxf(t1, t2);
        if (i == 0 || tp < t) { t = tp; ax = i; }
    }
    float n[3] = {0.f, 0.f, 0.f};
    n[ax] = -copysignf(1.f, dd[ax]);
    const float Gv = -(n[0] * d.x + n[1] * d.y + n[2] * d.z) / (t * t);
    PosSample<R> ps;
    ps.p = lift<R>(Vec3f{d.x * t + ref_p.x, d.y * t + ref_p.y, d.z * t + ref_p.z});
    ps.n = lift<R>(Vec3f{n[0], n[1], n[2]});
    ps.J = R(1.f);
    ps.pdf = pdf * Gv;
    ps.valid = true;
    ps.tri = -1;
    return ps;
}
// EnvironmentMap::__sample_position_pdf (envmap.cpp:124-143), detached
PSDR_HD float env_position_pdf(const SceneView &sc, const Vec3f &ref_p, const Vec3f &p, const Vec3f &n) {
    const float *f = sc.d.env_f;
    Vec3f d = p - ref_p;
    const float dist2 = dot(d, d);
    d = d / sqrtf(fmaxf(dist2, 0.f));
    const float Gv = fabsf(dot(d, n)) / dist2;
    d = env_mul3(f + PSDR_ENV_FROM_WORLD, d);
    const float factor = Gv * (1.f / sqrtf(fmaxf(d.x * d.x + d.z * d.z, kEpsilon * kEpsilon))) * (.5f / (kPi * kPi));
    float u = atan2f(d.x, -d.z) * kInvTwoPi, v = safe_acos_(d.y) * kInvPi;
    u -= floorf(u); v -= floorf(v);
    return env_cells_pdf(sc, u, v) * factor;
}

// Scene::sample_emitter_position (scene.cpp:427-447) -> AreaLight::sample_position (area.cpp:32-46)
// -> Mesh::__sample_position (mesh.cpp:306-330), or EnvironmentMap::sample_position
template <class R, class TVT>
PSDR_HD PosSample<R> sample_emitter_position(const SceneView &sc, const TVT &tv, const Vec3f &ref_p, float s0, float s1, bool with_J) {
    PosSample<R> ps;
    int e = 0; float epdf = 1.f;
    if (sc.d.num_emitters > 1) e = sample_reuse(Tab<TVT::flags>::emitter_cmf(sc), Tab<TVT::flags>::emitter_pmf(sc), sc.d.emitter_sum, sc.d.num_emitters, s1, epdf);
    if (TVT::has_env && e == sc.d.env_emitter) {
        ps = env_sample_position<R>(sc, ref_p, s0, s1);
        ps.pdf *= epdf;
        return ps;
    }
    const float *ef = Tab<TVT::flags>::emitter_f(sc, e);
    const int32_t *ei = Tab<TVT::flags>::emitter_i(sc, e);
    float fp;
    const int f = sample_reuse(Tab<TVT::flags>::face_cmf(sc) + ei[3], Tab<TVT::flags>::face_pmf(sc) + ei[3], ef[5], ei[2], s0, fp);
    const float t = sqrtf(fmaxf(1.f - s0, 0.f));           // warp::square_to_uniform_triangle, warp.h:76-80
    const TriRow<R> T = load_tri<R>(sc, tv, ei[1] + f);
    ps.J = R(1.f);
    if (is_ad<R>() && with_J) ps.J = T.area / detach(T.area);
    ps.p = bary_point(T.p0, T.e1, T.e2, R(1.f - t), R(t * s1));
    ps.n = T.fn;
    ps.pdf = ef[4] * epdf;
    ps.valid = true;
    ps.tri = ei[1] + f;
    return ps;
}
// Scene::emitter_position_pdf (scene.cpp:451-453) -> area.cpp:60-62 -> mesh.cpp:333-342, or envmap.cpp:124-143
template <class R, class TVT> PSDR_HD float emitter_position_pdf(const SceneView &sc, const TVT &tv, const Vec3f &ref_p, const Its<R> &its) {
    const int e = emitter_of(sc, tv, its);
    if (e < 0) return 0.f;
    if (TVT::has_env && e == sc.d.env_emitter) return env_position_pdf(sc, ref_p, val(its.p), val(its.n));
    const float *ef = Tab<TVT::flags>::emitter_f(sc, e);
    return ef[3] * ef[4];
}

// ---------------------------------------------------------------------------- camera
// A word of a launch-constant table at a wave-uniform index, read through the CONSTANT address space: in a kernel that also writes global
// memory (the reverse-mode kernels: gradient atomics) the compiler cannot prove a plain load invariant and fetches it per lane through the
// vector memory path -- the camera record alone was 23 of the 53 vector loads per slot of the C2 all-gradients kernel.  Nothing a launch
// reads this way is written by it (the camera's gradient goes to psdr_grads::g_cam_to_world).
#ifndef PSDR_UNIFORM_WORD
#define PSDR_UNIFORM_WORD 1
#endif
PSDR_HD float uniform_word(const float *tab, int i) {
#if defined(__HIP_DEVICE_COMPILE__) && PSDR_UNIFORM_WORD
    return ((const __attribute__((address_space(4))) float *) tab)[i];
#else
    return tab[i];
#endif
}
// PerspectiveCamera::sample_primary_ray (src/sensor/perspective.cpp:120-136)
template <class R, class TVT> PSDR_HD RayT<R> primary_ray(const SceneView &sc, const TVT &tv, float sx, float sy) {
    const float *m = sc.d.cam + PSDR_CAM_SAMPLE_TO_CAMERA;
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = uniform_word(m, r * 4) * sx + uniform_word(m, r * 4 + 1) * sy + uniform_word(m, r * 4 + 3);
    const float iw = 1.f / v[3];
    const Vec3f d = normalize(Vec3f{v[0] * iw, v[1] * iw, v[2] * iw});
    const float *c = sc.d.cam + PSDR_CAM_TO_WORLD;
    constexpr auto tm = &psdr_tangents::d_cam_to_world;
    auto tw = [&](int r, int col) { if constexpr (is_ad<R>()) return ldf<R>(c, tv, tm, r * 4 + col); else return R(uniform_word(c, r * 4 + col)); };
    RayT<R> ray;
    const R w = tw(3, 3);
    ray.o = Vec3<R>(tw(0, 3) / w, tw(1, 3) / w, tw(2, 3) / w);
    ray.d = Vec3<R>(tw(0, 0) * d.x + tw(0, 1) * d.y + tw(0, 2) * d.z, tw(1, 0) * d.x + tw(1, 1) * d.y + tw(1, 2) * d.z,
                    tw(2, 0) * d.x + tw(2, 1) * d.y + tw(2, 2) * d.z);
    return ray;
}
// PerspectiveCamera::sample_direct (perspective.cpp:139-155), all detached
PSDR_HD bool sample_direct(const SceneView &sc, const Vec3f &p, int &pixel, float &qx, float &qy, float &sensor_val) {
    const float *m = sc.d.cam + PSDR_CAM_WORLD_TO_SAMPLE;
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = m[r * 4] * p.x + m[r * 4 + 1] * p.y + m[r * 4 + 2] * p.z + m[r * 4 + 3];
    qx = v[0] / v[3]; qy = v[1] / v[3];
    const int W = sc.d.width, H = sc.d.height;
    const int ix = (int) floorf(qx * (float) W), iy = (int) floorf(qy * (float) H);
    const bool ok = ix >= 0 && ix < W && iy >= 0 && iy < H;
    pixel = ok ? iy * W + ix : -1;
    const float *c = sc.d.cam;
    Vec3f dir{p.x - c[PSDR_CAM_POS], p.y - c[PSDR_CAM_POS + 1], p.z - c[PSDR_CAM_POS + 2]};
    const float d2 = dot(dir, dir);
    dir = dir / sqrtf(fmaxf(d2, 0.f));
    const float ict = 1.f / (c[PSDR_CAM_DIR] * dir.x + c[PSDR_CAM_DIR + 1] * dir.y + c[PSDR_CAM_DIR + 2] * dir.z);
    sensor_val = (1.f / d2) * (ict * ict * ict) * c[PSDR_CAM_INV_AREA];
    return ok;
}

// ------------------------------------------------------------------------ integrators
#ifndef PSDR_EMITTER_PRETEST
#define PSDR_EMITTER_PRETEST 1          // BSDF-sampled rays that only matter if they end on an emitter are tested against the emitters' primitives first (0: always traced -- A/B builds)
#endif
#ifndef PSDR_SKIP_UNLIT
#define PSDR_SKIP_UNLIT 1               // light samples whose BSDF value is zero by its cosine tests are not traced (0: traced anyway -- A/B builds)
#endif
#ifndef PSDR_OCC_ROWS
#define PSDR_OCC_ROWS 1                 // kSceneTiny instances: light rays test the rows SceneView::occ names (0: every row -- A/B builds)
#endif
template <class R> PSDR_HD R mis_weight(const R &p1, const R &p2) { const R a = sqr(p1), b = sqr(p2); return a / (a + b); }  // direct.cpp:18-21

struct LiParams {                 // uniform per launch
    int integrator, bsdf_samples, light_samples, max_depth, hide_emitters, field;
};

// The loop bodies of DirectIntegrator::__Li (src/integrator/direct.cpp:64-160) evaluated at `its`.
// next_* report the FIRST BSDF-sampled vertex so the build-defined PathTracer can
// continue the path from it.  The D-mode forms (path-space hits, BSDF re-evaluated from the hit
// points, detached G in the pdfs) are used whenever the result type M carries tangents.
// NI: the record type the caller wants the BSDF-sampled next vertex in -- Its<G> (the fused kernels continue the path from it), or Its<float> where only
// its plain values are needed (the geometry-dual stages of the traced wavefront: a 196-byte Its<Dual<1>> handed out through the pointer went through
// scratch, store and reload: 300 B of scratch traffic per record, 20 of the stage's 30 GB)
// Member-wise copy of a hit record (scalars only).  `*next = its1` on an Its<Dual<K>> is lowered to a 196-byte memcpy that the optimiser does NOT split
// into registers again: the record went to scratch and came back (12 scratch_store_dwordx4 + ~25 loads per path vertex of every geometry-dual kernel).
template <class R> PSDR_HD void copy_its(Its<R> &d, const Its<R> &a) {
    d.valid = a.valid; d.tri = a.tri; d.mesh = a.mesh; d.hu = a.hu; d.hv = a.hv;
    d.wi.x = a.wi.x; d.wi.y = a.wi.y; d.wi.z = a.wi.z; d.p.x = a.p.x; d.p.y = a.p.y; d.p.z = a.p.z; d.n.x = a.n.x; d.n.y = a.n.y; d.n.z = a.n.z;
    d.t = a.t; d.J = a.J; d.uvx = a.uvx; d.uvy = a.uvy;
    d.sh.s.x = a.sh.s.x; d.sh.s.y = a.sh.s.y; d.sh.s.z = a.sh.s.z; d.sh.t.x = a.sh.t.x; d.sh.t.y = a.sh.t.y; d.sh.t.z = a.sh.t.z;
    d.sh.n.x = a.sh.n.x; d.sh.n.y = a.sh.n.y; d.sh.n.z = a.sh.n.z;
}
template <class NI, class G> PSDR_HD NI its_cast(const Its<G> &a) {
    if constexpr (std::is_same<NI, Its<G>>::value) { if constexpr (is_ad<G>()) { NI r; copy_its(r, a); return r; } else return a; }
    else {
        Its<float> r;
        r.valid = a.valid; r.tri = a.tri; r.mesh = a.mesh; r.hu = a.hu; r.hv = a.hv;
        r.wi = val(a.wi); r.p = val(a.p); r.n = val(a.n); r.t = val(a.t); r.J = val(a.J); r.uvx = val(a.uvx); r.uvy = val(a.uvy);
        r.sh.s = val(a.sh.s); r.sh.t = val(a.sh.t); r.sh.n = val(a.sh.n);
        return r;
    }
}
template <class G, class M, class TVT, class NI = Its<G>>
PSDR_HD Vec3<M> direct_step(const SceneView &sc, const TVT &tv, TraversalStack &st, Rng &rng, const Its<G> &its, bool active, int nB,
                            int nL, uint32_t &nrays, NI *next_its, Vec3<M> *next_f, bool *next_valid, int *light_tri = nullptr, bool emitter_only = false) {
    constexpr bool ad = is_ad<M>();
    constexpr HitForm form = is_ad<G>() ? kPathSpace : kDetached;
    Vec3<M> result = zero3<M>();
    int bsdf_id = active ? Tab<TVT::flags>::mesh_bsdf(sc, its.mesh) : 0;
    if (bsdf_id < 0) { active = false; bsdf_id = 0; }      // bounding mesh of the environment map, direct.cpp:54-57
    const Bsdf<G, M> bsdf(sc, tv, bsdf_id);
    // one BSDF sample (direct.cpp:64-118) / one emitter sample (direct.cpp:120-160) at `its`
    auto bsdf_sample = [&](const float (&s)[3], int i) {
        Vec3f wo_s; M pdf_s;
        bool a1 = bsdf.sample(sc, tv, its, s, active, wo_s, pdf_s);
        const Vec3f dir1 = val(its.sh.s) * wo_s.x + val(its.sh.t) * wo_s.y + val(its.sh.n) * wo_s.z;
        // emitter_only (the caller has no use for the hit unless it is an emitter): the ray against the emitters' primitives first -- the very tests the full search
        // would run on those rows; a ray that meets none of them cannot end on an emitter, whatever lies along it: not traced.  (Not under an environment map: every
        // ray that leaves the scene ends on its bounding mesh.  Two-level scenes only: where every ray costs six slab tests and nothing else -- the instances of a scene without a
        // tree -- the second copy of the primitive loop costs more registers and code than the five tests it saves on one ray in seven: C2 18 800 against 19 400.)
        if constexpr (!TVT::has_env && TVT::forest && PSDR_EMITTER_PRETEST && !is_ad<G>()) {          // (not in the geometry-dual instances: their largest kernel then spills in front of an exec restore, tools/check_spill_exec.py)
            if (emitter_only && sc.emit_rows != 0u && a1) a1 = closest_hit<false, 2, true>(sc, st, val(its.p), dir1, INFINITY, -1, -1, 0, sc.emit_rows).tri >= 0;
        }
        const RayT<G> ray1{its.p, lift<G>(dir1)};
        const Its<G> its1 = intersect<G>(sc, tv, st, ray1, a1, form, nrays, -1, -1, kPreBsdfRay);
        const bool a_hit = a1 && its1.valid;
        a1 = a_hit && emitter_of(sc, tv, its1) >= 0;
        Vec3<M> bsdf_val = zero3<M>(); M pdf0(0.f);
        if (a_hit) {
            if constexpr (ad) {
                const Vec3<G> wo = (its1.p - its.p) / its1.t;
                bsdf_val = bsdf.eval(sc, tv, its, its.sh.to_local(wo), true);
                const G Gv = abs_(dot(its1.n, -wo)) / sqr(its1.t);
                pdf0 = pdf_s * val(Gv);
                bsdf_val = bsdf_val * (to_m<M>(Gv * its1.J) / pdf0);
            } else {
                bsdf_val = bsdf.eval(sc, tv, its, lift<G>(wo_s), true);
                const G Gv = abs_(dot(its1.n, -ray1.d)) / sqr(its1.t);
                pdf0 = pdf_s * Gv;
                bsdf_val = bsdf_val / pdf_s;
            }
        }
        if (a1) {
            M w(1.f / (float) nB);
            if (nL > 0) w = w * mis_weight(pdf0, M(emitter_position_pdf(sc, tv, val(its.p), its1)));
            result = result + Le<M>(sc, tv, its1, true) * bsdf_val * w;
        }
        if (next_its && i == 0) { *next_its = its_cast<NI>(its1); *next_f = bsdf_val; *next_valid = a_hit; }
    };
    auto light_sample = [&](float s0, float s1, int i) {
        const PosSample<G> ps = sample_emitter_position<G>(sc, tv, val(its.p), s0, s1, is_ad<G>());
        Vec3<G> wo = ps.p - its.p;
        const G d2 = dot(wo, wo), dist = safe_sqrt(d2);
        wo = wo / dist;
        const RayT<G> ray1{its.p, wo};
        // An emitter sample below the vertex' horizon, or a vertex seen from behind: both BSDFs evaluate to zero there (diffuse.cpp:28-31, roughconductor.cpp:43-45:
        // cos theta_i > 0 and cos theta_o > 0), whatever the ray would find -- it is not traced (round 6; on a bunny half of the light samples).  Same value, fewer rays.
        const Vec3<G> wl = its.sh.to_local(wo);
        const bool lit = PSDR_SKIP_UNLIT ? (val(wl.z) > 0.f && val(its.wi.z) > 0.f) : true;
        // a scene without a tree: only the rows that can lie between this vertex' primitive and the sampled emitter triangle (SceneView::occ)
        auto trace_light = [&]() {
            if constexpr (TVT::tiny && PSDR_OCC_ROWS) return intersect<G, TVT, true>(sc, tv, st, ray1, ps.valid && lit, form, nrays, -1, -1, kPreLightRay, occ_rows(sc, its.tri, ps.tri));
            else return intersect<G>(sc, tv, st, ray1, ps.valid && lit, form, nrays, -1, -1, kPreLightRay);
        };
        const Its<G> its1 = trace_light();
        if (light_tri && i == 0) *light_tri = its1.valid ? its1.tri : -1;          // the value sweep of a split reverse launch records it (psdr_reverse.h RevDisk)
        if (!(its1.valid && val(its1.t) > val(dist) - kShadowEpsilon && emitter_of(sc, tv, its1) >= 0)) return;
        const G Gv = abs_(dot(its1.n, -wo)) / d2;
        const Vec3<M> bsdf_val = bsdf.eval(sc, tv, its, wl, true) * to_m<M>(Gv * ps.J / ps.pdf);
        M pdf1 = bsdf.pdf(sc, tv, its, wl, true);
        if constexpr (ad) pdf1 = pdf1 * val(Gv); else pdf1 = pdf1 * Gv;
        M w(1.f / (float) nL);
        if (nB > 0) w = w * mis_weight(M(ps.pdf), pdf1);
        result = result + Le<M>(sc, tv, its1, true) * bsdf_val * w;
    };
    if (nB == 1 && nL == 1 && next_its != nullptr && ad_traits<M>::K <= 1 && !is_ad<G>()) {
        // a PathTracer vertex (one sample of each kind, the BSDF sample's hit is the path's next vertex): the five numbers are drawn in the
        // reference's order, then the EMITTER sample is evaluated first -- its hit record dies with it, so the next vertex (~25 registers,
        // with the BSDF value and pdf) is not alive across a second trace.  0 + a + b = 0 + b + a: the same sum, bit for bit.  C2 K = 1
        // duals 55 -> 21 spilled VGPRs, 1.20 -> 1.14 ms; not for DirectIntegrator(1, 1) (nothing lives on: renderC 0.49 -> 0.52 ms) nor the
        // K = 3 duals (1.71 -> 1.76 ms): profiles/r04_sample_order_ab.txt; the geometry duals (G = Dual) gain nothing and keep the plain order
        const float s[3] = {rng.next(), rng.next(), rng.next()};
        const float s0 = rng.next(), s1 = rng.next();
        if (active) { light_sample(s0, s1, 0); bsdf_sample(s, 0); }
        return result;
    }
    for (int i = 0; i < nB; ++i) {
        const float s[3] = {rng.next(), rng.next(), rng.next()};
        if (!active) continue;
        bsdf_sample(s, i);
    }
    for (int i = 0; i < nL; ++i) {
        const float s0 = rng.next(), s1 = rng.next();
        if (!active) continue;
        light_sample(s0, s1, i);
    }
    return result;
}

template <bool KNOWN, class G, class TVT>
PSDR_HD Its<G> known_or_traced(const SceneView &sc, const TVT &tv, TraversalStack &st, const RayT<G> &ray, bool active, uint32_t &nrays, const Hit *primary) {
    if constexpr (KNOWN) {
        Its<G> its;
        its.valid = false; its.tri = its.mesh = -1; its.J = G(1.f); its.t = G(INFINITY);
        if (active) { nrays++; if (primary->tri >= 0) fill_its_from_hit<G>(its, sc, tv, *primary, ray, is_ad<G>() ? kSolidAngle : kDetached); }
        return its;
    } else {
        (void) primary;
        return intersect<G>(sc, tv, st, ray, active, is_ad<G>() ? kSolidAngle : kDetached, nrays, -1, -1, kPrePrimaryRay);
    }
}
// DirectIntegrator::__Li (direct.cpp:47-163); FieldExtractionIntegrator::__Li (field.cpp:34-54);
// PathTracer = iterated direct step (no reference implementation; depth 1 == DirectIntegrator(1,1)).
template <class G, class M, int INTEG = -1, bool KNOWN = false, class TVT>
PSDR_HD Vec3<M> Li(const SceneView &sc, const TVT &tv, TraversalStack &st, const LiParams &lp, Rng &rng, const RayT<G> &ray, bool active,
                   uint32_t &nrays, const Hit *primary = nullptr) {
    // INTEG >= 0: integrator fixed at compile time (the camera kernels are specialised per integrator so
    // the other integrators' code does not occupy registers / instruction cache); -1: run-time switch
    const int integ = INTEG >= 0 ? INTEG : lp.integrator;
    // renderD traces its primary ray in the solid-angle form (scene.cpp:355-376: p = o + t d, (u,v,t) from a
    // differentiable Moeller-Trumbore) when GEOMETRY carries tangents.  With material-only tangents the
    // two forms have the same value and derivative, and the on-surface form p = p0 + u e1 + v e2 is used:
    // o + t d sits up to ~1e-4 off the wall (fp32 t at distance ~1000), which lets ~1e-3 of the grazing
    // continuation rays re-hit their own wall -- isolated O(1) sample flips between any two fp32 builds
    // KNOWN: the closest hit of `ray` is *primary (closest_hit_pair found the two camera rays of a primary-edge sample in one walk)
    Its<G> its = known_or_traced<KNOWN, G>(sc, tv, st, ray, active, nrays, primary);
    active = active && its.valid;
    if (integ == PSDR_INTEGRATOR_FIELD) {
        if (!active) return zero3<M>();
        switch (lp.field) {
            case PSDR_FIELD_SILHOUETTE: return Vec3<M>(1.f);
            case PSDR_FIELD_POSITION: return to_m3<M>(its.p);
            case PSDR_FIELD_DEPTH: return Vec3<M>(to_m<M>(its.t), to_m<M>(its.t), to_m<M>(its.t));
            case PSDR_FIELD_GEONORMAL: return to_m3<M>(its.n);
            case PSDR_FIELD_SHNORMAL: return to_m3<M>(its.sh.n);
            default: return Vec3<M>(to_m<M>(its.uvx), to_m<M>(its.uvy), M(0.f));
        }
    }
    Vec3<M> result = lp.hide_emitters ? zero3<M>() : Le<M>(sc, tv, its, active);
    if (integ == PSDR_INTEGRATOR_DIRECT)
        return result + direct_step<G, M>(sc, tv, st, rng, its, active, lp.bsdf_samples, lp.light_samples, nrays, (Its<G> *) nullptr, nullptr, nullptr, nullptr, true);
    Vec3<M> beta(1.f);
    for (int depth = 0; depth < lp.max_depth; ++depth) {
        Its<G> nits; Vec3<M> nf; bool nvalid = false;
        const Vec3<M> c = direct_step<G, M>(sc, tv, st, rng, its, active, 1, 1, nrays, &nits, &nf, &nvalid, nullptr, depth + 1 >= lp.max_depth);          // last vertex: nothing continues from its BSDF sample
        if (active) {
            result = result + beta * c;
            active = nvalid;
            if (active) {
                beta = beta * nf;
                if constexpr (is_ad<G>()) copy_its(its, nits); else its = nits;          // (member-wise: see copy_its)
                const Vec3f b = val(beta);
                if (!(b.x != 0.f || b.y != 0.f || b.z != 0.f)) active = false;
            }
        }
    }
    return result;
}

// masked(value, ~isfinite(value)) = 0 (integrator.cpp:87), per component; a non-finite tangent is
// dropped with it (the reference's harness zeroes those afterwards, run_test.py:130).
PSDR_HD float zero_nonfinite(float x) { return isfinite(x) ? x : 0.f; }
template <int K> PSDR_HD Dual<K> zero_nonfinite(const Dual<K> &x) {
    Dual<K> r; const bool fv = isfinite(x.v); r.v = fv ? x.v : 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) r.d[k] = (fv && isfinite(x.d[k])) ? x.d[k] : 0.f;
    return r;
}
template <class R> PSDR_HD Vec3<R> zero_nonfinite(const Vec3<R> &v) { return {zero_nonfinite(v.x), zero_nonfinite(v.y), zero_nonfinite(v.z)}; }

// PathTracer, forward mode, tangents on DIFFUSE ALBEDO TEXELS ONLY (the headline's renderD w.r.t. the diffuse albedo, examples/run_test.py:126-129): a path's
// contribution is a product of albedos times factors no tangent reaches (the cosine-hemisphere pdf, the MIS weights, the emitted radiance), so
//     d/dP [ beta_k c_k ] = beta_k c_k * sum_{j <= k} (d rho_j / rho_j)          per channel,
// and the whole estimator runs on PLAIN FLOATS with three running sums per tangent set instead of dual numbers through every BSDF value, weight and
// throughput product (C2: the K = 1 kernel 0.99 -> see DESIGN.md round 5).  Exact wherever the albedo of a texel that carries a tangent is not zero;
// the launch checks that on the device (psdr_kernels.h k_logd_check) and runs the dual-number kernel otherwise.  tv: the K tangent sets (texels only).
template <int K, class TVT>
PSDR_HD void albedo_logd(const SceneView &sc, const TVT &tv, const Its<float> &its, float (&g)[K][3]) {
    const int bsdf_id = Tab<TVT::flags>::mesh_bsdf(sc, its.mesh);
#pragma unroll
    for (int k = 0; k < K; ++k) g[k][0] = g[k][1] = g[k][2] = 0.f;
    if (bsdf_id < 0) return;
    const Bsdf<float, Dual<K>> bsdf(sc, tv, bsdf_id);
    const Vec3<Dual<K>> rho = bsdf.tex3(sc, tv, PSDR_SLOT_REFLECTANCE, its);
    const float ix = rho.x.v != 0.f ? 1.f / rho.x.v : 0.f, iy = rho.y.v != 0.f ? 1.f / rho.y.v : 0.f, iz = rho.z.v != 0.f ? 1.f / rho.z.v : 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) { g[k][0] = rho.x.d[k] * ix; g[k][1] = rho.y.d[k] * iy; g[k][2] = rho.z.d[k] * iz; }
}
template <int K, class TVT>
PSDR_HD Vec3<Dual<K>> li_path_logd(const SceneView &sc, const TVT &tv, TraversalStack &st, const LiParams &lp, Rng &rng, const RayT<float> &ray, uint32_t &nrays) {
    const TangentView<0, TVT::flags> tv0{};
    Its<float> its = intersect<float>(sc, tv0, st, ray, true, kDetached, nrays, -1, -1, kPrePrimaryRay);
    bool active = its.valid;
    Vec3f result = lp.hide_emitters ? Vec3f(0.f) : Le<float>(sc, tv0, its, active);
    float rd[K][3], s[K][3];
#pragma unroll
    for (int k = 0; k < K; ++k) { rd[k][0] = rd[k][1] = rd[k][2] = 0.f; s[k][0] = s[k][1] = s[k][2] = 0.f; }
    Vec3f beta(1.f);
    for (int depth = 0; depth < lp.max_depth; ++depth) {
        Its<float> nits; Vec3f nf; bool nvalid = false;
        if (active) {
            // the vertex' own quotient joins the running sums BEFORE the estimator runs: nothing but the sums is alive across its two rays
            float g[K][3];
            albedo_logd<K>(sc, tv, its, g);
#pragma unroll
            for (int k = 0; k < K; ++k) { s[k][0] += g[k][0]; s[k][1] += g[k][1]; s[k][2] += g[k][2]; }
        }
        const Vec3f c = direct_step<float, float>(sc, tv0, st, rng, its, active, 1, 1, nrays, &nits, &nf, &nvalid, nullptr, depth + 1 >= lp.max_depth);
        if (active) {
            const Vec3f bc = beta * c;
            result = result + bc;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                rd[k][0] = fmaf(bc.x, s[k][0], rd[k][0]); rd[k][1] = fmaf(bc.y, s[k][1], rd[k][1]); rd[k][2] = fmaf(bc.z, s[k][2], rd[k][2]);
            }
            active = nvalid;
            if (active) {
                beta = beta * nf;
                its = nits;
                if (!(beta.x != 0.f || beta.y != 0.f || beta.z != 0.f)) active = false;
            }
        }
    }
    Vec3<Dual<K>> out;
    out.x.v = result.x; out.y.v = result.y; out.z.v = result.z;
#pragma unroll
    for (int k = 0; k < K; ++k) { out.x.d[k] = rd[k][0]; out.y.d[k] = rd[k][1]; out.z.d[k] = rd[k][2]; }
    return out;
}
template <int K, class TVT>
PSDR_HD Vec3<Dual<K>> camera_sample_logd(const SceneView &sc, const TVT &tv, TraversalStack &st, const LiParams &lp, const RngJump &jump, int pixel, uint64_t slot, uint32_t &nrays) {
    Rng rng; rng.init(slot, jump);
    const float j0 = rng.next(), j1 = rng.next();
    const int W = sc.d.width;
    const float sx = ((float) (pixel % W) + j0) / (float) W, sy = ((float) (pixel / W) + j1) / (float) sc.d.height;
    const TangentView<0, TVT::flags> tv0{};
    const RayT<float> ray = primary_ray<float>(sc, tv0, sx, sy);
    return zero_nonfinite(li_path_logd<K>(sc, tv, st, lp, rng, ray, nrays));
}


// Wavefront mode, stage 0: camera ray, primary hit, Le and the direct step at the primary vertex;
// reports the BSDF-sampled continuation (next vertex record + throughput).
template <class M, class TVT>
PSDR_HD Vec3<M> wavefront_camera_vertex(const SceneView &sc, const TVT &tv, TraversalStack &st, const LiParams &lp, const RngJump &jump,
                                        int pixel, uint64_t slot, uint32_t &nrays, Its<float> &next, Vec3<M> &beta, Vec3f &origin, bool &alive, Rng *rng_after = nullptr) {
    Rng rng; rng.init(slot, jump);
    const float j0 = rng.next(), j1 = rng.next();
    const int W = sc.d.width;
    const float sx = ((float) (pixel % W) + j0) / (float) W, sy = ((float) (pixel / W) + j1) / (float) sc.d.height;
    const RayT<float> ray = primary_ray<float>(sc, tv, sx, sy);
    const Its<float> its = intersect<float>(sc, tv, st, ray, true, kDetached, nrays);     // geometry is plain fp32 here, see Li
    alive = false;
    if (!its.valid) return zero3<M>();
    Vec3<M> result = lp.hide_emitters ? zero3<M>() : Le<M>(sc, tv, its, true);
    bool nvalid = false;
    result = result + direct_step<float, M>(sc, tv, st, rng, its, true, 1, 1, nrays, &next, &beta, &nvalid, nullptr, lp.max_depth <= 1);
    if (rng_after) *rng_after = rng;           // the stream continues at the next vertex (classify_next)
    origin = its.p;
    if (nvalid) { const Vec3f b = val(beta); alive = b.x != 0.f || b.y != 0.f || b.z != 0.f; }
    return result;
}
// Wavefront mode with binned streams: the camera stage stops at the primary hit (emitted radiance seen directly) and hands
// the hit over as a stream record -- the direct step at the primary vertex then runs as bounce stage 0, class-pure like the
// others (its sample streams continue behind the two pixel jitter draws: the stage-0 jump).
template <class M, class TVT>
PSDR_HD Vec3<M> wavefront_primary_vertex(const SceneView &sc, const TVT &tv, TraversalStack &st, const LiParams &lp, const RngJump &jump,
                                         int pixel, uint64_t slot, uint32_t &nrays, Its<float> &next, Vec3f &dir, bool &alive, Rng *rng_after = nullptr, float *sxy = nullptr) {
    Rng rng; rng.init(slot, jump);
    const float j0 = rng.next(), j1 = rng.next();
    if (rng_after) *rng_after = rng;           // bounce stage 0 draws from here on (classify_next)
    const int W = sc.d.width;
    const float sx = ((float) (pixel % W) + j0) / (float) W, sy = ((float) (pixel / W) + j1) / (float) sc.d.height;
    if (sxy) { sxy[0] = sx; sxy[1] = sy; }     // the geometry-dual stages rebuild the camera ray (with its tangents) from the film position
    const RayT<float> ray = primary_ray<float>(sc, tv, sx, sy);
    const Its<float> its = intersect<float>(sc, tv, st, ray, true, kDetached, nrays);
    alive = its.valid;
    if (!its.valid) return zero3<M>();
    next = its;
    { const Vec3f d = its.p - ray.o; dir = d / norm(d); }      // the direction intersect() derives wi from
    return lp.hide_emitters ? zero3<M>() : Le<M>(sc, tv, its, true);
}
// Plain-float copy of a hit record that carries tangents (classify_next aims the next vertex' rays with it).
template <class G> PSDR_HD Its<float> detach_its(const Its<G> &a) {
    Its<float> r;
    r.valid = a.valid; r.tri = a.tri; r.mesh = a.mesh; r.hu = a.hu; r.hv = a.hv;
    r.wi = val(a.wi); r.p = val(a.p); r.n = val(a.n); r.t = val(a.t); r.J = val(a.J); r.uvx = val(a.uvx); r.uvy = val(a.uvy);
    r.sh.s = val(a.sh.s); r.sh.t = val(a.sh.t); r.sh.n = val(a.sh.n);
    return r;
}
// Wavefront mode, stage k >= 1: the direct step at a path vertex read back from the stream.
template <class M, class TVT>
PSDR_HD Vec3<M> wavefront_bounce_vertex(const SceneView &sc, const TVT &tv, TraversalStack &st, const RngJump &jump_k, uint64_t slot,
                                        const Its<float> &its, uint32_t &nrays, Its<float> &next, Vec3<M> &f, bool &alive, int *light_tri = nullptr,
                                        Rng *rng_after = nullptr, bool last = false) {
    Rng rng; rng.init(slot, jump_k);
    bool nvalid = false;
    if (light_tri) *light_tri = -1;
    const Vec3<M> c = direct_step<float, M>(sc, tv, st, rng, its, true, 1, 1, nrays, &next, &f, &nvalid, light_tri, last);
    if (rng_after) *rng_after = rng;           // five draws per vertex: the next stage's streams start here (classify_next)
    alive = nvalid;
    return c;
}

// One camera sample slot: Integrator::__render (src/integrator/integrator.cpp:64-95), before the splat
template <class G, class M, int INTEG = -1, class TVT>
PSDR_HD Vec3<M> camera_sample(const SceneView &sc, const TVT &tv, TraversalStack &st, const LiParams &lp, const RngJump &jump,
                              int pixel, uint64_t slot, uint32_t &nrays) {
    Rng rng; rng.init(slot, jump);
    const float j0 = rng.next(), j1 = rng.next();
    const int W = sc.d.width;
    const int px = pixel % W, py = pixel / W;
    const float sx = ((float) px + j0) / (float) W, sy = ((float) py + j1) / (float) sc.d.height;
    const RayT<G> ray = primary_ray<G>(sc, tv, sx, sy);
    return zero_nonfinite(Li<G, M, INTEG>(sc, tv, st, lp, rng, ray, true, nrays));
}

// PSDR_PRIMARY_EDGE_VIS_CHECK (macros.h:13; integrator.cpp:105-108, perspective.cpp:171-196), active when the caller supplies
// psdr_scene_desc::prim_edge_z (PSDR_PRIMARY_EDGE_VIS_CHECK, integrator.cpp:105-108, perspective.cpp:171-196): the point of
// edge k at parameter u must itself be visible from the camera -- the camera ray through it, cut 100 ShadowEpsilon before
// the point, hits nothing.  Row k = (1 / depth of the end points along the viewing direction, the adjacent faces): the
// reference interpolates sample-space z and unprojects, which in exact arithmetic is the same distance depth / cos; the
// edge's own faces, met AT that distance, are left out of the search instead of to fp32 round-off.
template <class TVT>
PSDR_HD bool primary_edge_point_visible(const SceneView &sc, const TVT &tv0, TraversalStack &st, int k, float u, float px, float py, uint32_t &nrays) {
    const float *z = sc.d.prim_edge_z + (size_t) k * 4;
    const float depth = 1.f / (z[0] * (1.f - u) + z[1] * u);
    const Vec3f cd = normalize(Vec3f{sc.d.cam[PSDR_CAM_DIR], sc.d.cam[PSDR_CAM_DIR + 1], sc.d.cam[PSDR_CAM_DIR + 2]});
    const RayT<float> ray = primary_ray<float>(sc, tv0, px, py);
    const float tmax = depth / dot(ray.d, cd) - 100.f * kShadowEpsilon;
    nrays++;
    return closest_hit<true, tree_mode<TVT::flags>()>(sc, st, ray.o, ray.d, tmax, __float_as_int_hd(z[2]), __float_as_int_hd(z[3])).tri < 0;
}

// the closest hits of the two camera rays of a primary-edge sample (the film point moved by -+ kEdgeEpsilon along the edge normal), one walk
template <class TVT0>
PSDR_HD void primary_edge_camera_hits(const SceneView &sc, const TVT0 &tv0, TraversalStack &st, float px, float py, float nx, float ny, Hit &h0, Hit &h1) {
    const RayT<float> r0 = primary_ray<float>(sc, tv0, px - kEdgeEpsilon * nx, py - kEdgeEpsilon * ny), r1 = primary_ray<float>(sc, tv0, px + kEdgeEpsilon * nx, py + kEdgeEpsilon * ny);
    closest_hit_pair(sc, st, r0.o, r0.d, r1.d, h0, h1);          // (a perspective camera: both rays leave its position)
}
// One primary-edge slot: Integrator::render_primary_edges (integrator.cpp:98-119) +
// PerspectiveCamera::sample_primary_edge (perspective.cpp:158-200).  Returns the pixel (or -1);
// tan[k][c] = d value / d P_k (the primal part is exactly zero: value -= detach(value)).
template <int K, int INTEG = -1, int FL = kSceneRough>
PSDR_HD int primary_edge_sample(const SceneView &sc, const TangentView<K, FL> &tv, TraversalStack &st, const LiParams &lp, const RngJump &jump,
                                uint64_t slot, float inv_sppe, float tan[K][3], uint32_t &nrays) {
    Rng rng; rng.init(slot, jump);
    float u = rng.next(), pmf;
    const int k = sample_reuse(sc.d.prim_cmf, sc.d.prim_pmf, sc.d.prim_sum, sc.d.num_prim_edges, u, pmf);
    const float *pe = sc.d.prim_edge + (size_t) k * PSDR_PEDGE_STRIDE;
    const float nx = pe[4], ny = pe[5];
    const float pdf = pmf / pe[6];
    const float px = pe[0] * (1.f - u) + pe[2] * u, py = pe[1] * (1.f - u) + pe[3] * u;
    const int W = sc.d.width, H = sc.d.height;
    const int ix = (int) floorf(px * (float) W), iy = (int) floorf(py * (float) H);
    // (pmf > 0: a table built on the device keeps the capacity of the candidate list, zero rows behind the kept ones -- never drawn unless NO edge is kept;
    //  pe[6] > 0: a capacity-ONE table whose only candidate was dropped still "draws" its zero row with pmf 1 -- sample_reuse's size == 1 shortcut -- and its
    //  length 0 would make the pdf infinite: ADVICE r4)
    bool valid = ix >= 0 && ix < W && iy >= 0 && iy < H && pmf > 0.f && pe[6] > 0.f;
    const TangentView<0, FL> tv0{};
    if (sc.d.prim_edge_z != nullptr && valid) valid = primary_edge_point_visible(sc, tv0, st, k, u, px, py, nrays);
    // Li on the two sides of the edge (ray_n first, then ray_p: the order the reference draws them in,
    // integrator.cpp:107-112) -- one loop body, so the estimator is instantiated once
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
    const Vec3f dL{(Ln.x - Lp.x) / pdf, (Ln.y - Lp.y) / pdf, (Ln.z - Lp.z) / pdf};
    const float xdn = px * nx + py * ny;
    const bool fin[3] = {isfinite(xdn * dL.x), isfinite(xdn * dL.y), isfinite(xdn * dL.z)};
#pragma unroll
    for (int t = 0; t < K; ++t) {
        const float *dp = tv.t[t].d_prim_edge ? tv.t[t].d_prim_edge + (size_t) k * PSDR_PEDGE_STRIDE : nullptr;
        const float dxdn = dp ? ((dp[0] * (1.f - u) + dp[2] * u) * nx + (dp[1] * (1.f - u) + dp[3] * u) * ny) : 0.f;
        const float g[3] = {dxdn * dL.x, dxdn * dL.y, dxdn * dL.z};
#pragma unroll
        for (int c = 0; c < 3; ++c) tan[t][c] = (fin[c] && isfinite(g[c])) ? g[c] * inv_sppe : 0.f;
    }
    return iy * W + ix;
}

// |x1 - p1| with x1 = where the camera ray through the film projection of p1 meets the plane of triangle `tri`
// (perspective.cpp:139-155, then :120-136, then utils.h:66-77), p1 = point (hu, hv) of triangle tri_c, in DOUBLE from the fp32 tables: the visibility test of
// eval_secondary_edge (direct.cpp:262).  ~100 fp64 operations for the ~1.5 % of the boundary samples that get here.
PSDR_HD double camera_return_distance(const SceneView &sc, int tri_c, float hu, float hv, int tri) {
    const float *cam = sc.d.cam;
    const float *w2s = cam + PSDR_CAM_WORLD_TO_SAMPLE, *s2c = cam + PSDR_CAM_SAMPLE_TO_CAMERA, *tw = cam + PSDR_CAM_TO_WORLD;
    // p1 rebuilt ON its triangle in double from the traversal barycentrics: an fp32 p1 sits up to 5e-5 off the plane,
    // which a grazing camera ray turns into |x1 - p1| ~ 1e-3
    const float *rc = sc.d.tri_info + (size_t) tri_c * PSDR_TRI_STRIDE;
    const double px = (double) rc[0] + (double) hu * rc[3] + (double) hv * rc[6], py = (double) rc[1] + (double) hu * rc[4] + (double) hv * rc[7],
                 pz = (double) rc[2] + (double) hu * rc[5] + (double) hv * rc[8];
    double v[4], c[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = (double) w2s[r * 4] * px + (double) w2s[r * 4 + 1] * py + (double) w2s[r * 4 + 2] * pz + (double) w2s[r * 4 + 3];
    const double qx = v[0] / v[3], qy = v[1] / v[3];
#pragma unroll
    for (int r = 0; r < 4; ++r) c[r] = (double) s2c[r * 4] * qx + (double) s2c[r * 4 + 1] * qy + (double) s2c[r * 4 + 3];
    double dx = c[0] / c[3], dy = c[1] / c[3], dz = c[2] / c[3];
    const double inv = 1.0 / sqrt(dx * dx + dy * dy + dz * dz);
    dx *= inv; dy *= inv; dz *= inv;
    const double w = tw[15];
    const double o[3] = {tw[3] / w, tw[7] / w, tw[11] / w};
    const double d[3] = {tw[0] * dx + tw[1] * dy + tw[2] * dz, tw[4] * dx + tw[5] * dy + tw[6] * dz, tw[8] * dx + tw[9] * dy + tw[10] * dz};
    const float *row = sc.d.tri_info + (size_t) tri * PSDR_TRI_STRIDE;
    const double p0[3] = {row[0], row[1], row[2]}, e1[3] = {row[3], row[4], row[5]}, e2[3] = {row[6], row[7], row[8]};
    const double h[3] = {d[1] * e2[2] - d[2] * e2[1], d[2] * e2[0] - d[0] * e2[2], d[0] * e2[1] - d[1] * e2[0]};
    const double f = 1.0 / (e1[0] * h[0] + e1[1] * h[1] + e1[2] * h[2]);
    const double s_[3] = {o[0] - p0[0], o[1] - p0[1], o[2] - p0[2]};
    const double u = f * (s_[0] * h[0] + s_[1] * h[1] + s_[2] * h[2]);
    const double q[3] = {s_[1] * e1[2] - s_[2] * e1[1], s_[2] * e1[0] - s_[0] * e1[2], s_[0] * e1[1] - s_[1] * e1[0]};
    const double vv = f * (d[0] * q[0] + d[1] * q[1] + d[2] * q[2]);
    const double x[3] = {p0[0] + u * e1[0] + vv * e2[0] - px, p0[1] + u * e1[1] + vv * e2[1] - py, p0[2] + u * e1[2] + vv * e2[2] - pz};
    return sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
}

// HyperCubeDistribution<3>::sample_reuse (src/core/cube_distrb.cpp:41-48)
PSDR_HD float guide_sample_reuse(const SceneView &sc, float s[3]) {
    float pmf;
    const int n = sc.d.num_guide_cells;
    const int idx = sample_reuse(sc.d.guide_cmf, sc.d.guide_pmf, sc.d.guide_sum, n, s[2], pmf);
    const int r1 = sc.d.guide_reso[1], r2 = sc.d.guide_reso[2];
    const int c0 = idx / (r1 * r2), rem = idx - c0 * r1 * r2, c1 = rem / r2, c2 = rem - c1 * r2;
    s[0] = (s[0] + (float) c0) * (1.f / (float) sc.d.guide_reso[0]);
    s[1] = (s[1] + (float) c1) * (1.f / (float) r1);
    s[2] = (s[2] + (float) c2) * (1.f / (float) r2);
    return pmf * (float) n;
}

// DirectIntegrator::eval_secondary_edge (direct.cpp:225-316) + Scene::sample_boundary_segment_direct
// (scene.cpp:456-492).  R = float: returns value0 (guiding, pixel -1); R = Dual<K>: tangent-only value.
// What decides whether a secondary-edge slot evaluates anything at all (sample_boundary_segment_direct + the first two rays of
// eval_secondary_edge, direct.cpp:225-262): the boundary segment must reach the emitter sample from the edge point and continue
// backwards onto a surface.  A few per cent of the slots pass.  The same draws and the same float arithmetic as
// secondary_edge_sample / secondary_edge_reverse, which repeat it for the survivors of a split launch (k_secondary_edge_filter).
template <int FL>
PSDR_HD bool secondary_edge_survives(const SceneView &sc, TraversalStack &st, const float s3[3], uint32_t &nrays) {
    const TangentView<0, FL> tv0{};
    float s1 = s3[0], pdf0;
    const int k = sample_reuse(sc.d.sec_cmf, sc.d.sec_pmf, sc.d.sec_sum, sc.d.num_sec_edges, s1, pdf0);
    const float *se = sc.d.sec_edge + (size_t) k * PSDR_SEDGE_STRIDE;
    const Vec3f ep0{se[0], se[1], se[2]}, ee1{se[3], se[4], se[5]}, n0{se[6], se[7], se[8]}, n1{se[9], se[10], se[11]};
    const bool is_boundary = se[15] != 0.f;
    const Vec3f p0 = ee1 * s1 + ep0;
    const PosSample<float> ps2 = sample_emitter_position<float>(sc, tv0, p0, s3[1], s3[2], false);
    const Vec3f p2 = ps2.p, bn = ps2.n;
    Vec3f e = p2 - p0;
    const float distSqr = dot(e, e);
    e = e / sqrtf(fmaxf(distSqr, 0.f));
    const float cosTheta = -dot(bn, e);
    const float d0n = dot(n0, e), d1n = dot(n1, e);
    const int sgn0 = d0n > kEdgeEpsilon ? 1 : (d0n < -kEdgeEpsilon ? -1 : 0), sgn1 = d1n > kEdgeEpsilon ? 1 : (d1n < -kEdgeEpsilon ? -1 : 0);
    bool valid = cosTheta > kEpsilon && (is_boundary ? sgn0 != 0 : sgn0 * sgn1 < 0);
    const Vec3f dir = normalize(p2 - p0);
    const bool skip = sc.d.sec_edge_faces != nullptr && !sc.literal_forms;
    const int f0 = skip ? sc.d.sec_edge_faces[2 * k] : -1, f1 = skip ? sc.d.sec_edge_faces[2 * k + 1] : -1;
    const Its<float> its2 = intersect<float>(sc, tv0, st, RayT<float>{p0, dir}, valid, kDetached, nrays, f0, f1, 0);
    valid = valid && its2.valid && norm(its2.p - p2) < kShadowEpsilon;
    const Its<float> its1c = intersect<float>(sc, tv0, st, RayT<float>{p0, -dir}, valid, kDetached, nrays, f0, f1, 1);
    return valid && its1c.valid;
}

// What the dense trace kernel needs of a secondary-edge slot BEFORE that test (the probe pass of a traced launch, psdr_kernels.h k_se_probe): the
// geometric part of the decision, the edge point, the direction towards the emitter sample and the edge (its adjacent faces are skipped by
// both rays).  The same draws and the same arithmetic as secondary_edge_survives.
template <int FL>
PSDR_HD bool secondary_edge_rays(const SceneView &sc, const float s3[3], Vec3f &p0, Vec3f &dir, int &edge) {
    const TangentView<0, FL> tv0{};
    float s1 = s3[0], pdf0;
    const int k = sample_reuse(sc.d.sec_cmf, sc.d.sec_pmf, sc.d.sec_sum, sc.d.num_sec_edges, s1, pdf0);
    const float *se = sc.d.sec_edge + (size_t) k * PSDR_SEDGE_STRIDE;
    const Vec3f ep0{se[0], se[1], se[2]}, ee1{se[3], se[4], se[5]}, n0{se[6], se[7], se[8]}, n1{se[9], se[10], se[11]};
    const bool is_boundary = se[15] != 0.f;
    p0 = ee1 * s1 + ep0;
    const PosSample<float> ps2 = sample_emitter_position<float>(sc, tv0, p0, s3[1], s3[2], false);
    const Vec3f p2 = ps2.p, bn = ps2.n;
    Vec3f e = p2 - p0;
    const float distSqr = dot(e, e);
    e = e / sqrtf(fmaxf(distSqr, 0.f));
    const float cosTheta = -dot(bn, e);
    const float d0n = dot(n0, e), d1n = dot(n1, e);
    const int sgn0 = d0n > kEdgeEpsilon ? 1 : (d0n < -kEdgeEpsilon ? -1 : 0), sgn1 = d1n > kEdgeEpsilon ? 1 : (d1n < -kEdgeEpsilon ? -1 : 0);
    dir = normalize(p2 - p0);
    edge = (sc.d.sec_edge_faces != nullptr && !sc.literal_forms) ? k : -1;
    return cosTheta > kEpsilon && (is_boundary ? sgn0 != 0 : sgn0 * sgn1 < 0);
}

template <class R, class TVT>
PSDR_HD int secondary_edge_sample(const SceneView &sc, const TVT &tv, TraversalStack &st, const float s3[3], Vec3<R> &out, uint32_t &nrays, bool count_first = true) {
    constexpr bool ad = is_ad<R>();
    out = zero3<R>();
    const TangentView<0, TVT::flags> tv0{};
    // -- sample_boundary_segment_direct
    float s1 = s3[0], pdf0;
    const int k = sample_reuse(sc.d.sec_cmf, sc.d.sec_pmf, sc.d.sec_sum, sc.d.num_sec_edges, s1, pdf0);
    const size_t off = (size_t) k * PSDR_SEDGE_STRIDE;
    const float *se = sc.d.sec_edge + off;
    constexpr auto sm = &psdr_tangents::d_sec_edge;
    const Vec3<R> ep0 = ld3<R>(sc.d.sec_edge, tv, sm, off), ee1 = ld3<R>(sc.d.sec_edge, tv, sm, off + 3);
    const Vec3f n0{se[6], se[7], se[8]}, n1{se[9], se[10], se[11]}, ep2{se[12], se[13], se[14]};
    const bool is_boundary = se[15] != 0.f;
    const Vec3<R> bp0 = ee1 * R(s1) + ep0;
    const Vec3f e1v = val(ee1);
    const float e1len = norm(e1v);
    const Vec3f edge = e1v / e1len, edge2 = ep2 - val(ep0), p0 = val(bp0);
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
    // -- eval_secondary_edge
    const Vec3f dir = normalize(p2 - p0);
    // the two rays that start ON the edge skip its adjacent faces when the caller supplies them (psdr_hip.h sec_edge_faces)
    const bool skip = sc.d.sec_edge_faces != nullptr && !sc.literal_forms;
    const int f0 = skip ? sc.d.sec_edge_faces[2 * k] : -1, f1 = skip ? sc.d.sec_edge_faces[2 * k + 1] : -1;
    uint32_t counted_before = 0;                 // split launch: secondary_edge_survives already traced (and counted) these two
    uint32_t &n12 = count_first ? nrays : counted_before;
    const Its<float> its2 = intersect<float>(sc, tv0, st, RayT<float>{p0, dir}, valid, kDetached, n12, f0, f1);
    valid = valid && its2.valid && norm(its2.p - p2) < kShadowEpsilon;
    const Its<float> its1c = intersect<float>(sc, tv0, st, RayT<float>{p0, -dir}, valid, kDetached, n12, f0, f1);
    valid = valid && its1c.valid;
    if (!valid) return -1;
    const Vec3f p1 = its1c.p;
    int pixel; float qx, qy, sensor_val;
    if (!sample_direct(sc, p1, pixel, qx, qy, sensor_val)) return -1;
    const RayT<R> camera_ray = primary_ray<R>(sc, tv, qx, qy);
    const Its<R> its1 = intersect<R>(sc, tv, st, camera_ray, true, ad ? kSolidAngle : kDetached, nrays);
    // "the camera sees p1" (direct.cpp:262: |its1.p - p1| < ShadowEpsilon).  The camera ray goes through the film
    // projection of p1, so the test measures how far projection + ray generation (two fp32 matrices, inverse to each
    // other only to fp32 accuracy) carry the ray from p1 on a grazing surface 1000 units away -- of the order of
    // ShadowEpsilon itself.  In fp32 the decision flips for high-weight samples (5e-3 of the cbox_bunny boundary
    // term against an fp64 evaluation); camera_return_distance evaluates the same quantity in double.
    if (!its1.valid) return -1;
    if (sc.literal_forms) { if (!(norm(val(its1.p) - p1) < kShadowEpsilon)) return -1; }                 // direct.cpp:262 as written
    else if (!(camera_return_distance(sc, its1c.tri, its1c.hu, its1c.hv, its1.tri) < (double) kShadowEpsilon)) return -1;
    const float dist = norm(p2 - p1), cos2 = fabsf(dot(bn, dir));
    const Vec3f ev = cross(edge, dir);
    const float sinphi = norm(ev);
    const Vec3f proj = normalize(cross(ev, bn));
    const float sinphi2 = norm(cross(dir, proj));
    const float base_v = (its1c.t / dist) * (sinphi / sinphi2) * cos2;
    if (!(sinphi > kEpsilon && sinphi2 > kEpsilon)) return -1;
    const Vec3f d0 = -val(camera_ray.d);
    const Vec3f d0_local = its1c.sh.to_local(d0);
    if (Tab<TVT::flags>::mesh_bsdf(sc, its1c.mesh) < 0) return -1;          // bounding mesh: null BSDF evaluates to zero
    const Bsdf<float, float> bsdf(sc, tv0, Tab<TVT::flags>::mesh_bsdf(sc, its1c.mesh));
    Vec3f bsdf_val = bsdf.eval(sc, tv0, its1c, d0_local, true);
    const float correction = fabsf((its1c.wi.z * dot(d0, its1c.n)) / (d0_local.z * dot(dir, its1c.n)));
    bsdf_val = bsdf_val * correction;
    Vec3f value0 = bsdf_val * Le<float>(sc, tv0, its2, true) * (base_v * sensor_val / bpdf);
    if constexpr (ad) {
        const Vec3f n = normalize(cross(bn, proj));
        value0 = value0 * (copysignf(1.f, dot(ev, edge2)) * copysignf(1.f, dot(ev, n)));
        const TriRow<R> T = load_tri<R>(sc, tv, its2.tri);
        const RayT<R> shadow{its1.p, normalize(bp0 - its1.p)};
        R u, v, t;
        moeller_trumbore(T.p0, T.e1, T.e2, shadow, u, v, t);
        const Vec3<R> u2 = bary_point(detach(T.p0), detach(T.e1), detach(T.e2), u, v);
        const R dn = dot(lift<R>(n), u2);
        const Vec3<R> res{dn * value0.x, dn * value0.y, dn * value0.z};
        out = res - detach(res);
        return pixel;
    } else {
        out = value0;
        return -1;
    }
}

}  // namespace psdr
