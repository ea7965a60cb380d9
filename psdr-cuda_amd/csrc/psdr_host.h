// psdr_host.h -- what the translation units of libpsdr_hip.so share: the launch context, the scene
// handle, the host-side launch helpers (defined in psdr_hip.hip) and the table of kernel variants.
//
// The kernels are compiled once per SCENE FLAG SET (TangentView FLAGS in psdr_device.h: environment map
// present, rough conductor present) in separate translation units (psdr_variant.hip with
// -DPSDR_VARIANT_FLAGS=n), built in parallel; the C ABI in psdr_hip.hip picks the variant of the scene.
#pragma once
#include "psdr_device.h"
#include "psdr_reverse.h"

#include <algorithm>
#include <type_traits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace psdr;

struct LaunchCtx {
    SceneView sc;
    LiParams lp;
    RngJump jump;
    int32_t off_stack;          // byte offset of the traversal stacks inside the dynamic LDS block
    int32_t off_pathrec;        // reverse mode: per-lane (c_k, f_k) path records behind the stacks
    int32_t off_sink;           // reverse mode: the gradient cache of the workgroup (DeviceSink) behind those
};

static_assert(offsetof(LaunchCtx, sc) == 0, "closest_hit reads SceneView::tiny through the kernel-argument segment pointer: LaunchCtx (SceneView first) must be the first kernel argument");

// Dynamic LDS block of every kernel:  [ staged BVH nodes | staged leaf triangles | staged
// TriangleInfo rows | traversal stacks (stack_entries x 256 lanes) ].  The stacks are sized from the
// depth of THIS scene's tree, so a 12-triangle scene uses 5 KB instead of 40 KB and the freed LDS
// holds the scene itself (LDS latency ~64 clk vs ~200 for an L2 hit).
__device__ __forceinline__ void setup_lds(const LaunchCtx &cx, TraversalStack &st) {
#if defined(__HIP_DEVICE_COMPILE__)
    const SceneView &sc = cx.sc;
    float4 *dst = reinterpret_cast<float4 *>(psdr_dyn_lds);
    // the top of the tree this launch walks (level order): 4-wide nodes or BVH2 nodes, 64 bytes either way
    const float4 *src = sc.nodes4 != nullptr ? reinterpret_cast<const float4 *>(sc.nodes4) : reinterpret_cast<const float4 *>(sc.nodes);
    for (int i = threadIdx.x; i < sc.n_lnodes * 4; i += kBlock) dst[sc.off_lnodes / 16 + (i >> 2) * (kLdsNodeStride / 16) + (i & 3)] = src[i];
    for (int i = threadIdx.x; i < sc.n_lbtris * 3; i += kBlock) dst[sc.off_lbtris / 16 + i] = sc.btris[i];
    src = reinterpret_cast<const float4 *>(sc.d.tri_info);
    for (int i = threadIdx.x; i < sc.n_ltri * 6; i += kBlock) dst[sc.off_ltri / 16 + i] = src[i];
    {
        // hit rows of the kernel-argument primitives (resolve_tiny_hit): the codes come by scalar loads (uniform index), 16 lanes decode one row
        float *m = reinterpret_cast<float *>(psdr_dyn_lds + sc.off_lprim);
#pragma unroll 1
        for (int i = 0; i < sc.n_tiny; ++i) {
            int32_t tri; float k[6];
            const int w = threadIdx.x & 7;
            const float4 Sm = sc.tiny[i * 4 + 2];
            const float S[4] = {Sm.x, Sm.y, Sm.z, Sm.w};
            tiny_hit_row(sc.tiny_meta + i * 4, (threadIdx.x >> 3) & 1, tri, k, (sc.aa_cnt != 0 && i < kAaSlots) ? S : nullptr);
            const float val = w == 0 ? __int_as_float(tri) : (w == 1 ? k[0] : w == 2 ? k[1] : w == 3 ? k[2] : w == 4 ? k[3] : w == 5 ? k[4] : w == 6 ? k[5] : 0.f);
            if (threadIdx.x < kTinyHitWords) m[i * kTinyHitWords + threadIdx.x] = val;
        }
    }
    if (sc.lt_meshbsdf >= 0) {
        // a scene without a tree: the small tables of a path vertex too (psdr_device.h Tab<FL>), same layouts as the caller's
        auto stage = [&](int off, const void *table, int words) {
            if (off < 0 || table == nullptr) return;
            const uint32_t *g = static_cast<const uint32_t *>(table);
            uint32_t *l = reinterpret_cast<uint32_t *>(psdr_dyn_lds + off);
#pragma unroll 1
            for (int i = threadIdx.x; i < words; i += kBlock) l[i] = g[i];
        };
        stage(sc.lt_trimesh, sc.d.tri_mesh, sc.d.num_tris);
        stage(sc.lt_meshbsdf, sc.d.mesh_bsdf, sc.d.num_meshes);
        stage(sc.lt_meshemitter, sc.d.mesh_emitter, sc.d.num_meshes);
        stage(sc.lt_bsdf, sc.d.bsdf_rec, sc.d.num_bsdfs * PSDR_BSDF_STRIDE);
        stage(sc.lt_emf, sc.d.emitter_f, sc.d.num_emitters * PSDR_EMITTER_F_STRIDE);
        stage(sc.lt_emi, sc.d.emitter_i, sc.d.num_emitters * PSDR_EMITTER_I_STRIDE);
        stage(sc.lt_ecmf, sc.d.emitter_cmf, sc.d.num_emitters);
        stage(sc.lt_epmf, sc.d.emitter_pmf, sc.d.num_emitters);
        stage(sc.lt_fcmf, sc.d.face_cmf, sc.lt_nfaces);
        stage(sc.lt_fpmf, sc.d.face_pmf, sc.lt_nfaces);
        stage(sc.lt_uv, sc.d.tri_uv, sc.d.num_tris * PSDR_TRIUV_STRIDE);
        stage(sc.lt_tex, sc.d.texels, sc.d.num_texels);
        stage(sc.lt_occ, sc.occ, sc.d.num_tris * sc.d.num_tris);
    }
    st.base = reinterpret_cast<int32_t *>(psdr_dyn_lds + cx.off_stack) + threadIdx.x;
    __syncthreads();
#else
    (void) cx; (void) st;
#endif
}
// Forward-mode kernels of the kSceneTiny instances: the K tangent texel pools behind the staged value pool (psdr_device.h texel<M, true>).
template <class TVT> __device__ __forceinline__ void setup_lds(const LaunchCtx &cx, TraversalStack &st, const TVT &tv) {
    setup_lds(cx, st);
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (Tab<TVT::flags>::lds_small && TVT::k > 0) {
        constexpr int K = TVT::k;
        static_assert(K <= 3, "plan_lds reserves three tangent texel pools");
        if (cx.sc.lt_tex >= 0) {
            float *l = reinterpret_cast<float *>(psdr_dyn_lds + cx.sc.lt_tex);
            const int nt = cx.sc.d.num_texels;
            for (int k = 0; k < K; ++k) {
                const float *g = tv.t[k].d_texels;
                for (int i = threadIdx.x; i < nt; i += kBlock) l[(k + 1) * nt + i] = g ? g[i] : 0.f;
            }
            __syncthreads();
        }
    }
#endif
}

// ------------------------------------------------------------------------ reverse mode
// Gradient sink of the reverse kernels: the "per-parameter gradient scatter-add".
//   level 0  camera matrix: per-lane REGISTERS, wave-reduced once at kernel end
//   level 1  primary-triangle row: summed across the wave (lanes share their pixel) by the kernel
//   level 2  per-workgroup LDS cache (ds_add_f32) for everything hot: all texels and emitter
//            radiances (if they fit), and the "hot" triangle rows = the whole table when it is small,
//            otherwise the emitter meshes' rows (every light sample lands on them) plus the
//            largest-area triangles (walls, floors: the rows most path vertices land on)
//            The cache is REPLICATED up to 16 times when it is small (lane -> copy lane % rep): ds_add_f32 of
//            lanes that land on the same word serialise (64 lanes on one albedo texel or one wall row = 64 LDS
//            cycles; the C2 geometry-gradient kernel kept the LDS pipeline busy 80 cycles per instruction)
//   level 3  hardware global_atomic_add_f32 on the gradient table for the incoherent remainder
// The cache is flushed with one global atomic per non-zero cached word per workgroup.  It lives in the DYNAMIC LDS
// block and is as large as the launch's layout needs: a fixed 24 KB array held the material-gradient kernel at
// 3 workgroups per CU on a scene whose cache is 5 KB (C2 PathTracer(3) texel gradient 4.5 -> 3.4 ms, 2.8 ms
// with the fourth wave).
// Non-finite pieces are dropped (forward mode zeroes non-finite tangents, zero_nonfinite).
constexpr int kSinkCacheWords = 6144;        // at most 24 KB of LDS; a launch allocates what its layout needs (sink_bytes)
struct SinkLayout {
    int tex_off, tex_n;                      // texel cache (tex_n = 0: not cached)
    int rad_off, rad_n;
    int cam_off;                             // 16 words
    int env_off, env_n;                      // environment-map record (PSDR_ENV_WORDS) if wanted
    int hot_off, hot_rows;                   // cached triangle rows: slot = hot_map[tri] (-1 = not cached)
    const int32_t *hot_map, *hot_tris;       // [T] tri -> slot, [hot_rows] slot -> tri
    int total;
    int rep, stride;                         // copies of the cache (power of two) and their distance in words (odd)
    // lane-private accumulators (plain LDS read-add-write, no atomics) for the rows EVERY light sample lands on: the first one
    // or two triangles of emitter 0 (an area-light quad) and its radiance.  kPrivWords words per lane behind the cache copies.
    int priv_off, priv_rows;                 // priv_rows = 0: off
    int priv_tri[2], priv_slot[2], priv_emitter;
    int priv_regs;                           // 1: the kernel keeps these accumulators in REGISTERS (psdr_kernels.h RegPrivSink): no LDS block behind the cache
    // deferred row adjoints (DeviceSink::defer_row): pend_rows columns of kPendWords words per lane behind everything else (0: off -- rows are scattered on the spot)
    int pend_off, pend_rows;
};
constexpr int kPendWords = 10;               // triangle, (u, v), position (3), face normal (3), area
constexpr int kPrivRowWords = 13;            // position (p0, e1, e2: 9), face normal (3), area (1)
constexpr int kPrivWords = 2 * kPrivRowWords + 3;

// The scene handle behind psdr_scene_t: the caller's tables, the BVH on the device and per-handle scratch.
// Developer options of a handle (psdr_scene_set_option; the library reads no environment variable): A/B switches of the strategies below, for
// tools and tests.  The defaults are what every measurement in DESIGN.md refers to.
struct psdr_scene_options {
    int blocks_per_cu = 0;                 // cap of the grid-stride grids (0: per kernel)
    int camera_blocks = 0;                 // workgroups per CU of the forward camera kernels (0: camera_blocks_per_cu)
    int tiny_variants = 1;                 // scenes without a tree run the kSceneTiny kernel instances
    int wide = -1;                         // 0: never derive the 4-wide tree for the render kernels (-1: by scene)
    int lds_budget = 0;                    // LDS per workgroup for stacks + staged scene (0: 28 KB / 40 KB by variant)
    int sink_rep = 4;                      // copies of the LDS gradient cache at most
    int sink_private = 1;                  // lane-private accumulators for the emitter's rows
    int rev_split = -1;                    // reverse mode as value kernel + adjoint kernel: 1 / 0 force, -1 by scene and launch size
    int tangent_live = 1;                  // forward mode with geometry tangents: one bit per triangle "some tangent set moves this row" (TangentView::live); 0: every row loads its tangents
    int wf_geo = 1;                        // PathTracer forward mode with geometry tangents on a two-level scene as the traced wavefront (k_wfg_*); 0: the fused kernel
    int own_pixels = 1;                    // camera kernels store a pixel whose samples all sit in one wave (64 % samples per pixel == 0) instead of adding to it with atomics (0: always atomics)
    int emitter_pretest = 1;               // direct_step: BSDF-sampled rays whose hit matters only on an emitter meet the emitters' primitives first (SceneView::emit_rows); 0: off
    int occ_rows = 1;                      // scenes without a tree: light rays test only the rows that can lie between the vertex and the emitter sample (0: every row; takes effect at the next psdr_bvh_build)
    int logd = 1;                          // PathTracer forward mode with tangents on diffuse albedo texels only: the log-derivative kernel (0: always dual numbers)
    int keep_records = 1;                  // psdr_render_c honours PSDR_FLAG_KEEP_RECORDS (0: ignored -- A/B, tests)
    int rev_sorted = 1;                    // reverse camera kernels with geometry gradients: complete row adjoints wait in LDS and leave sorted by row at the slot's end (0: scattered on the spot)
    int sedge_split = -1;                  // secondary-edge term as filter + survivor kernel: 1 / 0 force, -1 from 2^18 slots
    int chunk_log2 = 0;                    // log2 of the slots per chunk of the chunked launches (0: 2^24 / 2^25, the traced wavefront 2^26)
    int probe = 1;                         // two-level scenes: fused kernels as probe pass + dense trace kernel + final pass where that is built (0: one kernel)
    int trace_wg2 = -1;                    // the dense trace kernel as two workgroups per CU: -1 by forest and launch size, 0 never, n > 0 always (stack columns of n entries in LDS)
    int bvh_maxleaf = 4;                   // host SAH builder: leaf size limit (1..8)
    float bvh_tcost = 2.0f;                //                   cost of a node visit in triangle tests
};

struct psdr_scene_s {
    psdr_scene_options opt;
    psdr_scene_desc desc{};                // caller-owned device tables (psdr_scene_set_tables)
    bool have_tables = false;
    bool has_rough = true;                 // a RoughConductor may be present (desc.material_mask)
    int num_cus = 256;
    int lds_limit = 64 * 1024;             // hipDeviceProp_t::sharedMemPerBlock (160 KB on gfx950)

    // BVH on the device (psdr_bvh_build)
    BvhNode *d_nodes = nullptr;
    float4 *d_btris = nullptr;
    size_t cap_nodes = 0, cap_btris = 0;
    int32_t root = 0;
    bool have_bvh = false;
    int bvh_depth = 0, num_nodes = 0, num_btris = 0;
    // the 4-wide tree the kernels walk (psdr_device.h Bvh4Node): topology from the host collapse, boxes from k_bvh4_fill
    Bvh4Node *d_nodes4 = nullptr;
    int32_t *d_topo4 = nullptr;            // [n4][4] child, then [n4][4] src
    size_t cap_nodes4 = 0;
    int num_nodes4 = 0, stack_need4 = 0;
    bool wide = false;                     // the render kernels of this scene walk the 4-wide tree (psdr_hip.hip use_wide_tree); the dense trace kernel
                                           // always does -- the 4-wide forest of a two-level scene exists whenever num_nodes4 > 0
    int32_t *d_trace_ovf = nullptr;        // k_wf_trace: stack entries beyond the LDS columns
    size_t trace_ovf_bytes = 0;
    bool traced_enabled = true;
    int32_t root4 = 0, blas_root4[kMaxBlas] = {};
    // device refit of the tree between rebuilds (psdr_hip.hip k_refit_*)
    bool refit_enabled = true;
    int tree_tris = -1, refits_since_build = 0, num_builds = 0, num_refits = 0;
    std::vector<int> level_start;          // breadth-first node order: first node of every level (+ end)
    float bvh_pad = 0.f, built_area = 0.f;
    float *d_refit_area = nullptr;

    // tree built on the device (psdr_lbvh.h): -1 = by size, 0 = never (host SAH), 1 = always; scratch kept for the refit
    int bvh_device_mode = -1;
    bool lbvh = false;
    int lbvh_leaves = 0;
    void *d_lbvh = nullptr;
    size_t lbvh_bytes = 0;

    // tiny scenes: the leaf triangles as they travel in the kernel arguments (SceneView::tiny)
    bool tiny_enabled = true, aa_enabled = true;
    int n_tiny = 0, aa_cnt = 0;     // rows in use / slab-form slots per axis
    float4 tiny[kTinyRows * 4] = {};           // plane form / slab form of the first n_aa (tiny_plane_form)
    int32_t tiny_meta[kTinyRows * 4] = {};
    // two-level tree (psdr_bvh_build.h ForestBuilder): boxes + roots of the per-mesh trees, as they travel in the
    // kernel arguments; d_top / d_inline_ids serve the refresh after a device refit
    bool two_level_enabled = true, refit_ok = true, wf_binned = true;
    int n_blas = 0, n_inline = 0;
    float4 blas_lo[kMaxBlas] = {}, blas_hi[kMaxBlas] = {};
    float4 *d_top = nullptr;
    int32_t *d_inline_ids = nullptr;

    // counters of the last render call (psdr_get_counters)
    unsigned long long *d_counters = nullptr;
    hipStream_t last_stream = nullptr;     // stream of the last render call (psdr_get_counters reads behind it)
    hipStream_t refit_stream = nullptr;    // stream of the last device refit
    uint64_t slots[3] = {0, 0, 0};
    int last_path_depth = 0;

    // reverse-mode gradient sink: triangle rows cached in LDS (chosen at build time)
    std::vector<int32_t> emitter_i;        // host copy of desc.emitter_i (the emitter meshes' rows are hot)
    int32_t *d_hot_map = nullptr, *d_hot_tris = nullptr;
    uint32_t *d_occ = nullptr; size_t occ_cap = 0; bool have_occ = false; int occ_max_rows = 0;      // occluder rows of a scene without a tree (SceneView::occ; psdr_bvh_build), [num_tris^2]
    int hot_rows = 0;
    bool hot_identity = false;             // every row cached, slot == triangle (scenes without a tree)
    size_t hot_cap = 0;

    // primary-edge slots sorted by pixel (psdr_hip.hip primary_edge_order)
    bool sort_edges = true;
    void *d_sort = nullptr;
    size_t sort_bytes = 0;

    // reverse mode, primary edges: copies of the edge gradient table (PrimaryEdgeSink)
    void *d_pe_rep = nullptr;
    size_t pe_rep_bytes = 0;

    // reverse mode, paths deeper than the LDS record: per-thread path records
    void *d_rev_deep = nullptr;
    size_t rev_deep_bytes = 0;

    // split secondary-edge launch: survivor counter (first 256 bytes) + slot list
    void *d_se_list = nullptr;
    size_t se_list_bytes = 0;

    // split reverse launch: per-path records between the value kernel and the adjoint kernel
    void *d_rev = nullptr;
    size_t rev_bytes = 0;
    // ... kept across calls by psdr_render_c(PSDR_FLAG_KEEP_RECORDS) for the psdr_render_d_rev of the same samples and tables
    void *d_logd_bad = nullptr; size_t logd_bad_bytes = 0;      // the log-derivative launches' gate flag
    struct KeptRecords { bool valid = false; psdr_render_opts o{}; uint64_t gen = 0; long long n = 0; int kind = 0; } kept;      // kind 1: the traced wavefront's records (c, f per vertex), 0: the value kernel's (suffix radiances)
    uint64_t tables_gen = 0;               // bumped whenever psdr_scene_set_tables installs a descriptor that differs from the current one

    // wavefront PathTracer: path-state streams + stream counters
    void *d_ws = nullptr;
    size_t ws_bytes = 0;
    // forward mode, geometry tangents: liveness bits of the tri_info rows (TangentView::live)
    void *d_live = nullptr;
    size_t live_bytes = 0;
    // probe / final launches: hit rows, masks, trace requests
    void *d_probe = nullptr;
    size_t probe_bytes = 0;
    // psdr_bvh_build: a two-level tree only with at least this many inline triangles.  Rounds 2-5 kept scenes without something room-like (< 6 inline triangles: two
    // bunnies, a light quad and a backdrop) on ONE tree -- measured on 4 M-slot PathTracer launches (2.9 against 3.1 ms).  At BASELINE configs[2]'s size every launch
    // form that runs its tree walks in the dense trace kernel wins there too (bunny_light 512^2 x 128: DirectIntegrator three-term reverse 39.6 -> 33.9 ms, forward
    // 40.8 -> 35.8, PathTracer(3) renderC 8.0 -> 6.2, PathTracer(3) reverse 25.2 -> 12.8; profiles/r06_bunny_light_forest.txt): default 0 since round 6
    uint32_t emit_rows = 0;                // SceneView::emit_rows of the current tree (psdr_bvh_build)
    int forest_min_inline = 0;
};

constexpr int kRayCounters = 64, kRayCounterStride = 16;     // d_counters: 64 counters, 128 bytes apart
// wavefront streams and trace-request queues: 64 sub-streams / sub-queues, their counters on their own 128-byte lines (psdr_kernels.h)
constexpr int kWfSub = 64, kWfCountStride = 32;

namespace psdr_host {
int fail(const std::string &m);
// Growth of a per-handle scratch buffer INSIDE a render call: stream-ordered (hipFreeAsync / hipMallocAsync on the call's stream --
// no device-wide synchronise on the hot path of the first large call or after a size change), and refused with a clear message when the device does
// not have the memory (the traced wavefront takes 16 GB for a 2^26-slot chunk: fine for eight ranks on eight GPUs, not for two processes on one).
int scratch_reserve(void **buf, size_t *have, size_t need, hipStream_t s, const char *what);
long long fit_chunk(long long chunk, size_t bytes_per_slot, size_t fixed_bytes, size_t have);      // the largest power-of-two fraction of `chunk` whose workspace the device can hold now
inline int workspace_reserve(psdr_scene_s *h, size_t need, hipStream_t s) { return scratch_reserve(&h->d_ws, &h->ws_bytes, need, s, "wavefront workspace (path-state streams + trace requests)"); }
int launch_blocks(const psdr_scene_s *h, long long n, int per_cu = 16);
int plan_lds(const psdr_scene_s *h, LaunchCtx &cx, int reserved = 0);
int lds_bytes(const LaunchCtx &cx, const psdr_scene_s *h);
void fill_top(const psdr_scene_s *h, SceneView &sc);
constexpr int kMaxInlineTris = 2 * kTinyTris;      // inline triangles of a two-level tree before pairing
int make_ctx(psdr_scene_s *h, const psdr_render_opts *o, int sampler, LaunchCtx &cx);
bool use_wavefront(const psdr_scene_s *h, const psdr_render_opts *o);
// traced wavefront (psdr_hip.hip k_wf_trace): the scene has a two-level tree whose 4-wide forest is on the device
bool traced_wavefront(const psdr_scene_s *h);
int launch_wf_trace(psdr_scene_s *h, const float4 *req, const int32_t *count, long long sub_cap, float4 *hit, hipStream_t s, bool ign = false);
// Probe / final launches of the fused kernels on two-level scenes (psdr_kernels.h): the buffers between the probe pass, the trace kernel and the final pass
struct ProbeBuffers { float4 *hit; uint32_t *mask; float4 *req; int32_t *count; long long sub_cap; };
int probe_buffers(psdr_scene_s *h, long long slots, int rays_per_slot, ProbeBuffers &pb, hipStream_t s, int per_cu = 16);   // per_cu: workgroups per CU of the probe launch (sizes the request queues)
SinkLayout make_sink_layout(const psdr_scene_s *h, const psdr_grads *g);
inline int sink_bytes_base(const SinkLayout &L) { return (L.priv_rows > 0 && !L.priv_regs) ? (L.priv_off + kPrivWords * kBlock) * 4 : (L.rep * L.stride * 4 + 15) / 16 * 16; }
inline int sink_bytes(const SinkLayout &L) { return L.pend_rows > 0 ? (L.pend_off + L.pend_rows * kPendWords * kBlock) * 4 : sink_bytes_base(L); }
// reserve the deferred-row block behind the cache and the private block: call when the layout is otherwise final
inline void sink_reserve_pending(SinkLayout &L, int rows) { L.pend_rows = 0; L.pend_off = (sink_bytes_base(L) / 4 + 3) / 4 * 4; L.pend_rows = rows; }
int check_counts(const psdr_scene_s *h, const psdr_render_opts *o);
int begin_call(psdr_scene_s *h, hipStream_t s);
int primary_edge_order(psdr_scene_s *h, const LaunchCtx &cx, long long i0, long long n, const uint32_t **order, hipStream_t s);
constexpr int kMaxRefits = 64;              // full SAH rebuild at least this often
constexpr float kRefitAreaGrowth = 1.3f;    // ... or when the summed inner-box area grew by 30 %

// entry points of one kernel variant (one scene flag set)
struct VariantOps {
    int (*render_c)(psdr_scene_s *h, const psdr_render_opts *o, float *img, hipStream_t s);
    int (*render_fwd)(psdr_scene_s *h, const psdr_render_opts *o, int K, const psdr_tangents *tangents, float *img, float *dimg, hipStream_t s);
    int (*render_rev)(psdr_scene_s *h, const psdr_render_opts *o, const float *adj_img, float *out_img, const psdr_grads *grads, hipStream_t s);
    int (*guide)(psdr_scene_s *h, LaunchCtx &cx, const int32_t reso[4], int nrounds, long long n, float *out_mass, hipStream_t s);
};
const VariantOps *variant_ops_0();
const VariantOps *variant_ops_1();
const VariantOps *variant_ops_2();
const VariantOps *variant_ops_3();
const VariantOps *variant_ops_4();      // two-level tree (kSceneForest), diffuse
const VariantOps *variant_ops_6();      // two-level tree, rough conductor
const VariantOps *variant_ops_8();      // no tree at all (kSceneTiny: every primitive in the kernel arguments), diffuse
const VariantOps *variant_ops_10();     // no tree, rough conductor
}  // namespace psdr_host

#define HIP_TRY(expr)                                                                              \
    do { hipError_t e_ = (expr); if (e_ != hipSuccess) return psdr_host::fail(std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)
