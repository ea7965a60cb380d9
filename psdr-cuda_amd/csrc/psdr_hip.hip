// psdr_hip.hip -- gfx950 kernels and the C ABI of include/psdr_hip.h.
//
// Kernel inventory (all hand-written HIP for CDNA4, wave64):
//   k_trace            closest-hit BVH traversal over SoA ray streams       (replaces OptiX launch)
//   k_camera<R>        one lane = one camera sample slot: raygen + Li + segmented wave splat
//   k_primary_edge<K>  one lane = one primary-edge slot (two detached Li evaluations)
//   k_secondary_edge<R> one lane = one secondary-edge slot (3 rays + boundary integrand)
//   k_guide            guiding-grid mass accumulation
// Traversal stacks live in LDS (one column per lane); image accumulation uses a segmented
// wave reduction followed by one hardware f32 atomic per (pixel run, channel).
#include "psdr_device.h"
#include "psdr_reverse.h"
#include "psdr_bvh_build.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

using namespace psdr;

// ============================================================================ device helpers
namespace {

__device__ __forceinline__ float wave_shfl_down(float v, int off) { return __shfl_down(v, off, 64); }

// Segmented sum over a wave for NON-DECREASING integer keys (camera slots are pixel-major, so a
// wave covers a few consecutive pixels).  After the loop the first lane of every key run holds
// the run total.  All 64 lanes must call this.
template <int N> __device__ __forceinline__ bool wave_segmented_sum(int key, float (&v)[N]) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int okey = __shfl_down(key, off, 64);
        const bool take = (lane + off < 64) && (okey == key);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const float o = wave_shfl_down(v[i], off);
            if (take) v[i] += o;
        }
    }
    const int pkey = __shfl_up(key, 1, 64);
    return lane == 0 || pkey != key;
}

// Sum of v over runs of ADJACENT lanes holding the same key (keys in any order); the first lane of
// each run gets the total.
template <int N> __device__ __forceinline__ bool wave_run_sum(int key, float (&v)[N]) {
    const int lane = threadIdx.x & 63;
    const int prev = __shfl_up(key, 1, 64);
    const bool head = lane == 0 || prev != key;
    const unsigned long long heads = __ballot(head);
    const int seg = __popcll(heads & (~0ull >> (63 - lane)));       // run index: non-decreasing
    wave_segmented_sum<N>(seg, v);
    return head;
}

__device__ __forceinline__ void count_rays(unsigned long long *counters, uint32_t nrays) {
    uint32_t s = nrays;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(counters, (unsigned long long) s);
}

struct LaunchCtx {
    SceneView sc;
    LiParams lp;
    RngJump jump;
    int32_t off_stack;          // byte offset of the traversal stacks inside the dynamic LDS block
    int32_t off_pathrec;        // reverse mode: per-lane (c_k, f_k) path records behind the stacks
};

// Dynamic LDS block of every kernel:  [ staged BVH nodes | staged leaf triangles | staged
// TriangleInfo rows | traversal stacks (stack_entries x 256 lanes) ].  The stacks are sized from the
// depth of THIS scene's tree, so a 12-triangle scene uses 5 KB instead of 40 KB and the freed LDS
// holds the scene itself (LDS latency ~64 clk vs ~200 for an L2 hit).
__device__ __forceinline__ void setup_lds(const LaunchCtx &cx, TraversalStack &st) {
#if defined(__HIP_DEVICE_COMPILE__)
    const SceneView &sc = cx.sc;
    float4 *dst = reinterpret_cast<float4 *>(psdr_dyn_lds);
    const float4 *src = reinterpret_cast<const float4 *>(sc.nodes);
    for (int i = threadIdx.x; i < sc.n_lnodes * 4; i += kBlock) dst[sc.off_lnodes / 16 + i] = src[i];
    for (int i = threadIdx.x; i < sc.n_lbtris * 3; i += kBlock) dst[sc.off_lbtris / 16 + i] = sc.btris[i];
    src = reinterpret_cast<const float4 *>(sc.d.tri_info);
    for (int i = threadIdx.x; i < sc.n_ltri * 6; i += kBlock) dst[sc.off_ltri / 16 + i] = src[i];
    st.base = reinterpret_cast<int32_t *>(psdr_dyn_lds + cx.off_stack) + threadIdx.x;
    __syncthreads();
#else
    (void) cx; (void) st;
#endif
}

// ------------------------------------------------------------------------------- k_trace
__global__ __launch_bounds__(kBlock) void k_trace(LaunchCtx cx, int m, const float *__restrict__ ox, const float *__restrict__ oy,
                                                  const float *__restrict__ oz, const float *__restrict__ dx,
                                                  const float *__restrict__ dy, const float *__restrict__ dz,
                                                  const float *__restrict__ tmax, int32_t *__restrict__ out_shape,
                                                  int32_t *__restrict__ out_tri, float *__restrict__ out_u, float *__restrict__ out_v) {
    TraversalStack st; setup_lds(cx, st);
    const SceneView &sc = cx.sc;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < m; i += gridDim.x * kBlock) {
        const Hit h = closest_hit(sc, st, Vec3f{ox[i], oy[i], oz[i]}, Vec3f{dx[i], dy[i], dz[i]}, tmax[i]);
        out_tri[i] = h.tri;
        out_shape[i] = h.tri >= 0 ? (sc.d.tri_mesh[h.tri] & ~PSDR_TRI_FACE_NORMALS) : -1;
        out_u[i] = h.u; out_v[i] = h.v;
    }
}

// ------------------------------------------------------------------------------ k_camera
// n = W*H*nsp slots of this shard, pixel-major: slot j -> pixel j / nsp, sample s_begin + j % nsp.
// Occupancy targets (waves per SIMD) of the camera kernel.  The kernel is latency/dependency bound
// (rocprof r01: 43 % of wave cycles waiting at 2 waves/SIMD), so trading registers for resident
// waves pays: measured 6.2 -> 4.6 ms (renderC, 4 waves) and 12.5 -> 7.0 ms (renderD K=3 material-only, 2 waves,
// no spills) on C2; 5 waves (renderC) and 3-4 waves (Dual<3>) spill and lose.
#ifndef PSDR_WAVES_C
#define PSDR_WAVES_C 4
#endif
#ifndef PSDR_WAVES_DM
#define PSDR_WAVES_DM 3
#endif
#ifndef PSDR_WAVES_DG
#define PSDR_WAVES_DG 2
#endif
template <class G, class R> constexpr int camera_waves() { return !is_ad<R>() ? PSDR_WAVES_C : (is_ad<G>() ? PSDR_WAVES_DG : PSDR_WAVES_DM); }
template <class G, class R, int INTEG, bool ENV>
__global__ __launch_bounds__(kBlock, (camera_waves<G, R>())) void k_camera(LaunchCtx cx, TV<R, ENV> tv, int spp, int s_begin, int nsp, long long n, float inv_spp,
                                                   float *__restrict__ img, float *__restrict__ dimg, long long plane,
                                                   unsigned long long *counters) {
    constexpr int K = ad_traits<R>::K;
    constexpr int NV = 3 * (1 + K);
    TraversalStack st; setup_lds(cx, st);
    uint32_t nrays = 0;
    const long long nceil = (n + kBlock - 1) / kBlock * kBlock;
    for (long long j = (long long) blockIdx.x * kBlock + threadIdx.x; j < nceil; j += (long long) gridDim.x * kBlock) {
        const bool in = j < n;
        const int pixel = in ? (int) (j / nsp) : 0x7fffffff;
        float v[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = 0.f;
        if (in) {
            const int s = s_begin + (int) (j % nsp);
            const uint64_t slot = (uint64_t) pixel * (uint64_t) spp + (uint64_t) s;
            const Vec3<R> r = camera_sample<G, R, INTEG>(cx.sc, tv, st, cx.lp, cx.jump, pixel, slot, nrays);
            v[0] = val(r.x) * inv_spp; v[1] = val(r.y) * inv_spp; v[2] = val(r.z) * inv_spp;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                v[3 + 3 * k] = tangent(r.x, k) * inv_spp; v[4 + 3 * k] = tangent(r.y, k) * inv_spp; v[5 + 3 * k] = tangent(r.z, k) * inv_spp;
            }
        }
        const bool head = wave_segmented_sum<NV>(pixel, v);
        if (head && in) {
            float *p = img + (size_t) pixel * 3;
            if (v[0] != 0.f) atomicAdd(p, v[0]);
            if (v[1] != 0.f) atomicAdd(p + 1, v[1]);
            if (v[2] != 0.f) atomicAdd(p + 2, v[2]);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                float *q = dimg + (size_t) k * plane + (size_t) pixel * 3;
                if (v[3 + 3 * k] != 0.f) atomicAdd(q, v[3 + 3 * k]);
                if (v[4 + 3 * k] != 0.f) atomicAdd(q + 1, v[4 + 3 * k]);
                if (v[5 + 3 * k] != 0.f) atomicAdd(q + 2, v[5 + 3 * k]);
            }
        }
    }
    count_rays(counters, nrays);
}

// ----------------------------------------------------------------------- wavefront mode
// PathTracer as a wavefront: stage 0 (camera ray, primary vertex) then one kernel per bounce over SoA
// path-state streams in HBM.  Live paths are compacted between stages with a wave ballot + prefix
// popcount and ONE atomic per wave on the stream counter, so every bounce kernel runs on dense
// waves.  Record = pixel, slot, triangle, (u,v), arrival direction, throughput (+K tangents):
// 44 + 12 K bytes, all streams coalesced.
struct PathStream {
    int32_t *pixel; uint32_t *slot; int32_t *tri; float *hu, *hv; float *dir; float *beta;   // dir: [3][cap], beta: [3(1+K)][cap]
    long long cap;
};

template <class M>
__device__ __forceinline__ void stream_push(const PathStream &out, int *counter, bool alive, int pixel, uint32_t slot, const Its<float> &next,
                                            const Vec3f &dir, const Vec3<M> &beta) {
    constexpr int K = ad_traits<M>::K;
    const unsigned long long mask = __ballot(alive);
    if (mask == 0ull) return;
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == __ffsll((long long) mask) - 1) base = atomicAdd(counter, (int) __popcll(mask));
    base = __shfl(base, __ffsll((long long) mask) - 1, 64);
    if (!alive) return;
    const long long i = base + __popcll(mask & ((1ull << lane) - 1ull));
    out.pixel[i] = pixel; out.slot[i] = slot; out.tri[i] = next.tri; out.hu[i] = next.hu; out.hv[i] = next.hv;
    out.dir[i] = dir.x; out.dir[out.cap + i] = dir.y; out.dir[2 * out.cap + i] = dir.z;
    out.beta[i] = val(beta.x); out.beta[out.cap + i] = val(beta.y); out.beta[2 * out.cap + i] = val(beta.z);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        out.beta[(3 + 3 * k) * out.cap + i] = tangent(beta.x, k); out.beta[(4 + 3 * k) * out.cap + i] = tangent(beta.y, k);
        out.beta[(5 + 3 * k) * out.cap + i] = tangent(beta.z, k);
    }
}

template <class M>
__device__ __forceinline__ void splat_runs(int pixel, bool valid, const Vec3<M> &r, float scale, float *img, float *dimg, long long plane) {
    constexpr int K = ad_traits<M>::K;
    constexpr int NV = 3 * (1 + K);
    float v[NV];
    v[0] = val(r.x) * scale; v[1] = val(r.y) * scale; v[2] = val(r.z) * scale;
#pragma unroll
    for (int k = 0; k < K; ++k) { v[3 + 3 * k] = tangent(r.x, k) * scale; v[4 + 3 * k] = tangent(r.y, k) * scale; v[5 + 3 * k] = tangent(r.z, k) * scale; }
    if (!valid) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = 0.f;
    }
    const bool head = wave_run_sum<NV>(valid ? pixel : -1, v);
    if (head && valid) {
        float *p = img + (size_t) pixel * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) if (v[c] != 0.f) atomicAdd(p + c, v[c]);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float *q = dimg + (size_t) k * plane + (size_t) pixel * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) if (v[3 + 3 * k + c] != 0.f) atomicAdd(q + c, v[3 + 3 * k + c]);
        }
    }
}

template <class M, bool ENV>
__global__ __launch_bounds__(kBlock, (is_ad<M>() ? 2 : 4)) void k_wf_camera(LaunchCtx cx, TV<M, ENV> tv, int spp, int s_begin, int nsp, long long j0, long long n,
                                                        float inv_spp, float *__restrict__ img, float *__restrict__ dimg, long long plane,
                                                        PathStream out, int *out_count, int want_next, unsigned long long *counters) {
    TraversalStack st; setup_lds(cx, st);
    uint32_t nrays = 0;
    const long long nceil = (n + kBlock - 1) / kBlock * kBlock;
    for (long long jj = (long long) blockIdx.x * kBlock + threadIdx.x; jj < nceil; jj += (long long) gridDim.x * kBlock) {
        const bool in = jj < n;
        const long long j = j0 + jj;
        const int pixel = in ? (int) (j / nsp) : 0;
        Vec3<M> r = zero3<M>(), beta = zero3<M>();
        Its<float> next; next.tri = -1; next.hu = next.hv = 0.f;
        Vec3f origin(0.f), dir(0.f);
        bool alive = false;
        uint32_t slot = 0;
        if (in) {
            slot = (uint32_t) ((uint64_t) pixel * (uint64_t) spp + (uint64_t) (s_begin + (int) (j % nsp)));
            r = zero_nonfinite(wavefront_camera_vertex<M>(cx.sc, tv, st, cx.lp, cx.jump, pixel, slot, nrays, next, beta, origin, alive));
            if (alive) { Vec3f d = next.p - origin; const float t = norm(d); dir = d / t; }
        }
        splat_runs<M>(pixel, in, r, inv_spp, img, dimg, plane);
        stream_push<M>(out, out_count, alive && want_next, pixel, slot, next, dir, beta);
    }
    count_rays(counters, nrays);
}

template <class M, bool ENV>
__global__ __launch_bounds__(kBlock, (is_ad<M>() ? 2 : 4)) void k_wf_bounce(LaunchCtx cx, TV<M, ENV> tv, float inv_spp, float *__restrict__ img,
                                                        float *__restrict__ dimg, long long plane, PathStream in, const int *in_count,
                                                        PathStream out, int *out_count, int want_next, unsigned long long *counters) {
    constexpr int K = ad_traits<M>::K;
    TraversalStack st; setup_lds(cx, st);
    uint32_t nrays = 0;
    const long long n = *in_count;
    const long long nceil = (n + kBlock - 1) / kBlock * kBlock;
    for (long long j = (long long) blockIdx.x * kBlock + threadIdx.x; j < nceil; j += (long long) gridDim.x * kBlock) {
        const bool live = j < n;
        int pixel = -1; uint32_t slot = 0;
        Vec3<M> r = zero3<M>(), beta = zero3<M>();
        Its<float> next; next.tri = -1; next.hu = next.hv = 0.f;
        Vec3f dir(0.f);
        bool alive = false;
        if (live) {
            pixel = in.pixel[j]; slot = in.slot[j];
            const Vec3f din{in.dir[j], in.dir[in.cap + j], in.dir[2 * in.cap + j]};
            beta.x = M(in.beta[j]); beta.y = M(in.beta[in.cap + j]); beta.z = M(in.beta[2 * in.cap + j]);
            if constexpr (K > 0) {
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    beta.x.d[k] = in.beta[(3 + 3 * k) * in.cap + j]; beta.y.d[k] = in.beta[(4 + 3 * k) * in.cap + j];
                    beta.z.d[k] = in.beta[(5 + 3 * k) * in.cap + j];
                }
            }
            const Its<float> its = path_vertex_from_record(cx.sc, tv, in.tri[j], in.hu[j], in.hv[j], din);
            Vec3<M> f;
            const Vec3<M> c = wavefront_bounce_vertex<M>(cx.sc, tv, st, cx.jump, (uint64_t) slot, its, nrays, next, f, alive);
            r = zero_nonfinite(beta * c);
            if (alive) {
                beta = beta * f;
                const Vec3f b = val(beta);
                alive = b.x != 0.f || b.y != 0.f || b.z != 0.f;
                Vec3f d = next.p - its.p; const float t = norm(d); dir = d / t;
            }
        }
        splat_runs<M>(pixel, live, r, inv_spp, img, dimg, plane);
        stream_push<M>(out, out_count, alive && want_next, pixel, slot, next, dir, beta);
    }
    count_rays(counters, nrays);
}

// ------------------------------------------------------------------------ k_primary_edge
template <int K, bool ENV>
__global__ __launch_bounds__(kBlock) void k_primary_edge(LaunchCtx cx, TangentView<K, ENV> tv, long long i0, long long n, float inv_sppe,
                                                         float *__restrict__ dimg, long long plane, unsigned long long *counters) {
    TraversalStack st; setup_lds(cx, st);
    uint32_t nrays = 0;
    for (long long j = (long long) blockIdx.x * kBlock + threadIdx.x; j < n; j += (long long) gridDim.x * kBlock) {
        float tan[K][3];
        const int pixel = primary_edge_sample<K>(cx.sc, tv, st, cx.lp, cx.jump, (uint64_t) (i0 + j), inv_sppe, tan, nrays);
        if (pixel >= 0) {
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    if (tan[k][c] != 0.f) atomicAdd(dimg + (size_t) k * plane + (size_t) pixel * 3 + c, tan[k][c]);
        }
    }
    count_rays(counters, nrays);
}

// ---------------------------------------------------------------------- k_secondary_edge
template <int K, bool ENV>
__global__ __launch_bounds__(kBlock) void k_secondary_edge(LaunchCtx cx, TangentView<K, ENV> tv, long long i0, long long n, float inv_sppse,
                                                           float *__restrict__ dimg, long long plane, unsigned long long *counters) {
    using R = Dual<K>;
    TraversalStack st; setup_lds(cx, st);
    uint32_t nrays = 0;
    const bool guided = cx.sc.d.guide_cmf != nullptr && cx.sc.d.num_guide_cells > 0;
    for (long long j = (long long) blockIdx.x * kBlock + threadIdx.x; j < n; j += (long long) gridDim.x * kBlock) {
        Rng rng; rng.init((uint64_t) (i0 + j), cx.jump);
        float s3[3] = {rng.next(), rng.next(), rng.next()};
        const float pdf0 = guided ? guide_sample_reuse(cx.sc, s3) : 1.f;
        Vec3<R> value;
        const int pixel = secondary_edge_sample<R>(cx.sc, tv, st, s3, value, nrays);
        if (pixel >= 0) {
            value = zero_nonfinite(value);
            const float scale = (pdf0 > kEpsilon ? 1.f / pdf0 : 1.f) * inv_sppse;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const float g[3] = {value.x.d[k] * scale, value.y.d[k] * scale, value.z.d[k] * scale};
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    if (g[c] != 0.f) atomicAdd(dimg + (size_t) k * plane + (size_t) pixel * 3 + c, g[c]);
            }
        }
    }
    count_rays(counters, nrays);
}

// ------------------------------------------------------------------------------- k_guide
// DirectIntegrator::preprocess_secondary_edges (direct.cpp:166-204): one lane = one (cell, j) sample
// stream; nrounds evaluations each; mass[cell] += hmax(value0 / reso3) / nrounds.
template <bool ENV>
__global__ __launch_bounds__(kBlock) void k_guide(LaunchCtx cx, int r0, int r1, int r2, int per, int nrounds, long long n,
                                                  float *__restrict__ mass, unsigned long long *counters) {
    TraversalStack st; setup_lds(cx, st);
    uint32_t nrays = 0;
    const TangentView<0, ENV> tv0{};
    const RngJump nojump{1ull, 0ull};
    for (long long j = (long long) blockIdx.x * kBlock + threadIdx.x; j < n; j += (long long) gridDim.x * kBlock) {
        const int cell = (int) (j / per);
        const int c0 = cell / (r1 * r2), rem = cell - c0 * r1 * r2, c1 = rem / r2, c2 = rem - c1 * r2;
        Rng rng; rng.init((uint64_t) j, nojump);
        float acc = 0.f;
        for (int r = 0; r < nrounds; ++r) {
            float s3[3] = {rng.next(), rng.next(), rng.next()};
            s3[0] = (s3[0] + (float) c0) * (1.f / (float) r0);
            s3[1] = (s3[1] + (float) c1) * (1.f / (float) r1);
            s3[2] = (s3[2] + (float) c2) * (1.f / (float) r2);
            Vec3f v;
            secondary_edge_sample<float>(cx.sc, tv0, st, s3, v, nrays);
            v = zero_nonfinite(v);
            if (per > 1) v = v / (float) per;
            acc += fmaxf(v.x, fmaxf(v.y, v.z));
        }
        if (nrounds > 1) acc /= (float) nrounds;
        if (acc != 0.f) atomicAdd(mass + cell, acc);
    }
    count_rays(counters, nrays);
}

// ------------------------------------------------------------------------ reverse mode
// Gradient sink of the reverse kernels: the "per-parameter gradient scatter-add".
//   level 0  camera matrix: per-lane REGISTERS, wave-reduced once at kernel end
//   level 1  primary-triangle row: summed across the wave (lanes share their pixel) by the kernel
//   level 2  per-workgroup LDS cache (ds_add_f32) for everything hot: all texels and emitter
//            radiances (if they fit), and the "hot" triangle rows = the whole table when it is small,
//            otherwise the emitter meshes' rows (every light sample lands on them) plus the
//            largest-area triangles (walls, floors: the rows most path vertices land on)
//   level 3  hardware global_atomic_add_f32 on the gradient table for the incoherent remainder
// The cache is flushed with one global atomic per non-zero cached word per workgroup.
// Non-finite pieces are dropped (forward mode zeroes non-finite tangents, zero_nonfinite).
constexpr int kSinkCacheWords = 6144;        // 24 KB of LDS next to the 40 KB of traversal stacks
struct SinkLayout {
    int tex_off, tex_n;                      // texel cache (tex_n = 0: not cached)
    int rad_off, rad_n;
    int cam_off;                             // 16 words
    int env_off, env_n;                      // environment-map record (PSDR_ENV_WORDS) if wanted
    int hot_off, hot_rows;                   // cached triangle rows: slot = hot_map[tri] (-1 = not cached)
    const int32_t *hot_map, *hot_tris;       // [T] tri -> slot, [hot_rows] slot -> tri
    int total;
};
template <bool ENV> struct DeviceSink {
    static constexpr bool has_env = ENV;
    psdr_grads g;
    SinkLayout L;
    float *lds;
    float cam[16];
    __device__ __forceinline__ static bool ok(float v) { return v != 0.f && isfinite(v); }
    __device__ __forceinline__ void glob(float *base, size_t i, float v) const { if (base != nullptr && ok(v)) atomicAdd(base + i, v); }
    __device__ __forceinline__ void add_tri(int tri, int word, float v) const {
        if (g.g_tri_info == nullptr || !ok(v)) return;
        const int slot = L.hot_rows ? L.hot_map[tri] : -1;
        if (slot >= 0 && slot < L.hot_rows) atomicAdd(lds + L.hot_off + slot * PSDR_TRI_STRIDE + word, v);
        else atomicAdd(g.g_tri_info + (size_t) tri * PSDR_TRI_STRIDE + word, v);
    }
    __device__ __forceinline__ void add_texel(int idx, float v) const {
        if (g.g_texels == nullptr || !ok(v)) return;
        if (L.tex_n) atomicAdd(lds + L.tex_off + idx, v); else atomicAdd(g.g_texels + idx, v);
    }
    __device__ __forceinline__ void add_rad(int e, int c, float v) const {
        if (g.g_emitter_rad == nullptr || !ok(v)) return;
        if (L.rad_n) atomicAdd(lds + L.rad_off + e * 3 + c, v); else atomicAdd(g.g_emitter_rad + e * 3 + c, v);
    }
    __device__ __forceinline__ void add_cam(int word, float v) { if (ok(v)) cam[word] += v; }
    __device__ __forceinline__ void add_env(int word, float v) const { if (L.env_n && ok(v)) atomicAdd(lds + L.env_off + word, v); }
    __device__ __forceinline__ void add_sedge(int e, int word, float v) const { glob(g.g_sec_edge, (size_t) e * PSDR_SEDGE_STRIDE + word, v); }
    __device__ __forceinline__ void add_pedge(int e, int word, float v) const { glob(g.g_prim_edge, (size_t) e * PSDR_PEDGE_STRIDE + word, v); }

    __device__ __forceinline__ void begin(float *cache) {
        lds = cache;
        for (int i = threadIdx.x; i < L.total; i += kBlock) cache[i] = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) cam[i] = 0.f;
        __syncthreads();
    }
    __device__ __forceinline__ void end() {
        if (g.g_cam_to_world != nullptr) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float v = cam[i];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
                if ((threadIdx.x & 63) == 0 && v != 0.f) atomicAdd(lds + L.cam_off + i, v);
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < L.total; i += kBlock) {
            const float v = lds[i];
            if (v == 0.f) continue;
            if (i >= L.cam_off && i < L.cam_off + 16) { atomicAdd(g.g_cam_to_world + (i - L.cam_off), v); continue; }
            if (L.tex_n && i >= L.tex_off && i < L.tex_off + L.tex_n) { atomicAdd(g.g_texels + (i - L.tex_off), v); continue; }
            if (L.rad_n && i >= L.rad_off && i < L.rad_off + L.rad_n) { atomicAdd(g.g_emitter_rad + (i - L.rad_off), v); continue; }
            if (L.env_n && i >= L.env_off && i < L.env_off + L.env_n) { atomicAdd(g.g_env_f + (i - L.env_off), v); continue; }
            const int rel = i - L.hot_off;
            if (rel >= 0 && rel < L.hot_rows * PSDR_TRI_STRIDE)
                atomicAdd(g.g_tri_info + (size_t) L.hot_tris[rel / PSDR_TRI_STRIDE] * PSDR_TRI_STRIDE + rel % PSDR_TRI_STRIDE, v);
        }
    }
};

#ifndef PSDR_WAVES_REV
#define PSDR_WAVES_REV 2
#endif
template <bool ENV>
__global__ __launch_bounds__(kBlock, PSDR_WAVES_REV) void k_camera_rev(LaunchCtx cx, DeviceSink<ENV> sink, int spp, int s_begin, int nsp, long long n, float inv_spp,
                                                       const float *__restrict__ adj_img, float *__restrict__ img,
                                                       unsigned long long *counters) {
    __shared__ float cache[kSinkCacheWords];
    TraversalStack st; setup_lds(cx, st);
    sink.begin(cache);
    uint32_t nrays = 0;
    const bool geo = sink.g.g_tri_info != nullptr || sink.g.g_cam_to_world != nullptr;
    const long long nceil = (n + kBlock - 1) / kBlock * kBlock;
    for (long long j = (long long) blockIdx.x * kBlock + threadIdx.x; j < nceil; j += (long long) gridDim.x * kBlock) {
        const bool in = j < n;
        const int pixel = in ? (int) (j / nsp) : 0x7fffffff;
        float v[3] = {0.f, 0.f, 0.f};
        PrimaryGrad pg; pg.clear();
        PathRec rec;
#if defined(__HIP_DEVICE_COMPILE__)
        rec.base = reinterpret_cast<float *>(psdr_dyn_lds + cx.off_pathrec) + threadIdx.x;
#endif
        if (in) {
            const int s = s_begin + (int) (j % nsp);
            const uint64_t slot = (uint64_t) pixel * (uint64_t) spp + (uint64_t) s;
            const float *a = adj_img + (size_t) pixel * 3;
            const Vec3f adj{a[0] * inv_spp, a[1] * inv_spp, a[2] * inv_spp};
            const Vec3f r = camera_sample_reverse(sink, pg, rec, cx.sc, st, cx.lp, cx.jump, pixel, slot, adj, nrays, geo);
            v[0] = r.x * inv_spp; v[1] = r.y * inv_spp; v[2] = r.z * inv_spp;
        }
        // primary-triangle row: one add per run of lanes that hit the same triangle
        if (sink.g.g_tri_info != nullptr) {
            const bool head = wave_run_sum<kPrimaryWords>(pg.tri, pg.w);
            if (head && pg.tri >= 0) {
#pragma unroll
                for (int w = 0; w < kPrimaryWords; ++w) sink.add_tri(pg.tri, w, pg.w[w]);
            }
        }
        if (img != nullptr) {
            const bool head = wave_segmented_sum<3>(pixel, v);
            if (head && in) {
                float *p = img + (size_t) pixel * 3;
                if (v[0] != 0.f) atomicAdd(p, v[0]);
                if (v[1] != 0.f) atomicAdd(p + 1, v[1]);
                if (v[2] != 0.f) atomicAdd(p + 2, v[2]);
            }
        }
    }
    sink.end();
    count_rays(counters, nrays);
}

template <bool ENV>
__global__ __launch_bounds__(kBlock) void k_primary_edge_rev(LaunchCtx cx, DeviceSink<ENV> sink, long long i0, long long n, float inv_sppe,
                                                             const float *__restrict__ adj_img, unsigned long long *counters) {
    __shared__ float cache[kSinkCacheWords];
    TraversalStack st; setup_lds(cx, st);
    sink.begin(cache);
    uint32_t nrays = 0;
    for (long long j = (long long) blockIdx.x * kBlock + threadIdx.x; j < n; j += (long long) gridDim.x * kBlock)
        primary_edge_reverse(sink, cx.sc, st, cx.lp, cx.jump, (uint64_t) (i0 + j), inv_sppe, adj_img, nrays);
    sink.end();
    count_rays(counters, nrays);
}

template <bool ENV>
__global__ __launch_bounds__(kBlock) void k_secondary_edge_rev(LaunchCtx cx, DeviceSink<ENV> sink, long long i0, long long n, float inv_sppse,
                                                               const float *__restrict__ adj_img, unsigned long long *counters) {
    __shared__ float cache[kSinkCacheWords];
    TraversalStack st; setup_lds(cx, st);
    sink.begin(cache);
    uint32_t nrays = 0;
    const bool guided = cx.sc.d.guide_cmf != nullptr && cx.sc.d.num_guide_cells > 0;
    for (long long j = (long long) blockIdx.x * kBlock + threadIdx.x; j < n; j += (long long) gridDim.x * kBlock) {
        Rng rng; rng.init((uint64_t) (i0 + j), cx.jump);
        float s3[3] = {rng.next(), rng.next(), rng.next()};
        const float pdf0 = guided ? guide_sample_reuse(cx.sc, s3) : 1.f;
        secondary_edge_reverse(sink, cx.sc, st, s3, (pdf0 > kEpsilon ? 1.f / pdf0 : 1.f) * inv_sppse, adj_img, nrays);
    }
    sink.end();
    count_rays(counters, nrays);
}

// ============================================================================ host side
thread_local std::string g_err;
int fail(const std::string &m) { g_err = m; return 1; }
#define HIP_TRY(expr)                                                                              \
    do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)

}  // namespace

struct psdr_scene_s {
    psdr_scene_desc desc{};
    bool have_tables = false;
    BvhNode *d_nodes = nullptr;
    float4 *d_btris = nullptr;
    size_t cap_nodes = 0, cap_btris = 0;
    int32_t root = 0;
    bool have_bvh = false;
    unsigned long long *d_counters = nullptr;
    uint64_t slots[3] = {0, 0, 0};
    int num_cus = 256;
    std::vector<int32_t> emitter_i;
    int bvh_depth = 0, num_nodes = 0, num_btris = 0;
    int32_t *d_hot_map = nullptr, *d_hot_tris = nullptr; int hot_rows = 0; size_t hot_cap = 0;   // reverse sink: LDS-cached triangle rows
    void *d_ws = nullptr; size_t ws_bytes = 0;
    int last_path_depth = 0; float path_survival = -1.f;   // rays traced / rays of fully surviving paths (last PathTracer call)          // wavefront path-state streams + counters          // host copy of desc.emitter_i (hot-row ranges of the reverse sink)
};

namespace {

int launch_blocks(const psdr_scene_s *h, long long n) {
    const long long need = (n + kBlock - 1) / kBlock;
    const long long cap = (long long) h->num_cus * 16;      // grid-stride beyond 16 resident-ish blocks per CU
    return (int) std::max(1LL, std::min(need, cap));
}

// LDS plan of one launch: stacks sized by the tree depth, the rest of a 40 KB budget (4 workgroups
// per CU) filled with the top of the BVH, then the leaf triangles, then the TriangleInfo rows.
constexpr int kLdsBudget = 40 * 1024;
int plan_lds(const psdr_scene_s *h, LaunchCtx &cx, int reserved = 0) {
    const int stack_entries = std::min(kBvhStack, h->bvh_depth + 2);
    const int stack_bytes = stack_entries * kBlock * 4;
    int room = std::max(0, kLdsBudget - reserved - stack_bytes);
    SceneView &sc = cx.sc;
    int off = 0;
    sc.n_lnodes = std::min(h->num_nodes, room / 64); sc.off_lnodes = off; off += sc.n_lnodes * 64; room -= sc.n_lnodes * 64;
    sc.n_lbtris = (sc.n_lnodes == h->num_nodes) ? std::min(h->num_btris, room / 48) : 0;
    sc.off_lbtris = off; off += sc.n_lbtris * 48; room -= sc.n_lbtris * 48;
    sc.n_ltri = (sc.n_lbtris == h->num_btris) ? std::min(h->desc.num_tris, room / 96) : 0;
    sc.off_ltri = off; off += sc.n_ltri * 96;
    cx.off_stack = off;
    return off + stack_bytes;
}
int lds_bytes(const LaunchCtx &cx, const psdr_scene_s *h) { return cx.off_stack + std::min(kBvhStack, h->bvh_depth + 2) * kBlock * 4; }

int make_ctx(psdr_scene_s *h, const psdr_render_opts *o, int sampler, LaunchCtx &cx) {
    if (!h->have_tables) return fail("Scene not loaded yet!");
    if (!h->have_bvh) return fail("Input scene must be configured!");
    if (o->integrator != PSDR_INTEGRATOR_FIELD && h->desc.num_emitters <= 0) return fail("No Emitter!");
    if (o->integrator == PSDR_INTEGRATOR_DIRECT && !(o->bsdf_samples >= 0 && o->light_samples >= 0 && o->bsdf_samples + o->light_samples > 0))
        return fail("DirectIntegrator: bsdf_samples + light_samples must be positive");
    cx.sc.d = h->desc; cx.sc.nodes = h->d_nodes; cx.sc.btris = h->d_btris; cx.sc.root = h->root;
    plan_lds(h, cx);
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

template <class G, class R, bool ENV>
int run_camera(psdr_scene_s *h, const psdr_render_opts *o, const TV<R, ENV> &tv, float *img, float *dimg, hipStream_t s) {
    const long long WH = (long long) h->desc.width * h->desc.height;
    const int nsp = o->spp_end - o->spp_begin;
    if (o->spp <= 0 || nsp <= 0) return 0;
    LaunchCtx cx;
    if (int rc = make_ctx(h, o, 0, cx)) return rc;
    const long long n = WH * nsp;
    h->slots[0] += (uint64_t) n;
#define PSDR_LAUNCH_CAMERA(INTEG)                                                                                                  \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_camera<G, R, INTEG, ENV>), dim3(launch_blocks(h, n)), dim3(kBlock), lds_bytes(cx, h), s, cx, tv, o->spp, \
                       o->spp_begin, nsp, n, 1.f / (float) o->spp, img, dimg, WH * 3, h->d_counters)
    switch (o->integrator) {
        case PSDR_INTEGRATOR_DIRECT: PSDR_LAUNCH_CAMERA(PSDR_INTEGRATOR_DIRECT); break;
        case PSDR_INTEGRATOR_PATH: PSDR_LAUNCH_CAMERA(PSDR_INTEGRATOR_PATH); break;
        case PSDR_INTEGRATOR_FIELD: PSDR_LAUNCH_CAMERA(PSDR_INTEGRATOR_FIELD); break;
        default: return fail("Unknown integrator");
    }
#undef PSDR_LAUNCH_CAMERA
    HIP_TRY(hipGetLastError());
    return 0;
}

// PathTracer interior term as a wavefront (see k_wf_camera / k_wf_bounce).  M = float or Dual<K>
// with plain-fp32 geometry.
constexpr long long kWfChunk = 1ll << 25;
template <class M, bool ENV>
int run_camera_wavefront(psdr_scene_s *h, const psdr_render_opts *o, const TV<M, ENV> &tv, float *img, float *dimg, hipStream_t s) {
    constexpr int K = ad_traits<M>::K;
    const long long WH = (long long) h->desc.width * h->desc.height;
    const int nsp = o->spp_end - o->spp_begin;
    if (o->spp <= 0 || nsp <= 0) return 0;
    const long long n = WH * nsp;
    const long long cap = std::min(n, kWfChunk);
    const int depth = o->max_depth;
    const size_t words = 8 + 3 * (1 + K);
    const size_t need = 2 * words * 4 * (size_t) cap + 256 * sizeof(int);
    if (need > h->ws_bytes) {
        if (h->d_ws) (void) hipFree(h->d_ws);
        h->d_ws = nullptr; h->ws_bytes = 0;
        HIP_TRY(hipMalloc(&h->d_ws, need));
        h->ws_bytes = need;
    }
    int *cnt = reinterpret_cast<int *>(h->d_ws);
    PathStream st[2];
    for (int i = 0; i < 2; ++i) {
        float *b = reinterpret_cast<float *>(reinterpret_cast<char *>(h->d_ws) + 256 * sizeof(int)) + (size_t) i * words * cap;
        st[i].cap = cap;
        st[i].pixel = reinterpret_cast<int32_t *>(b); st[i].slot = reinterpret_cast<uint32_t *>(b + cap); st[i].tri = reinterpret_cast<int32_t *>(b + 2 * cap);
        st[i].hu = b + 3 * cap; st[i].hv = b + 4 * cap; st[i].dir = b + 5 * cap; st[i].beta = b + 8 * cap;
    }
    h->slots[0] += (uint64_t) n;
    const float inv_spp = 1.f / (float) o->spp;
    for (long long j0 = 0; j0 < n; j0 += cap) {
        const long long cn = std::min(cap, n - j0);
        HIP_TRY(hipMemsetAsync(cnt, 0, 256 * sizeof(int), s));
        LaunchCtx cx;
        if (int rc = make_ctx(h, o, 0, cx)) return rc;
        const int blocks = launch_blocks(h, cn);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wf_camera<M, ENV>), dim3(blocks), dim3(kBlock), lds_bytes(cx, h), s, cx, tv, o->spp, o->spp_begin, nsp, j0, cn,
                           inv_spp, img, dimg, WH * 3, st[0], cnt + 1, depth > 1 ? 1 : 0, h->d_counters);
        HIP_TRY(hipGetLastError());
        for (int k = 1; k < depth; ++k) {
            cx.jump = make_rng_jump(o->rng_offset[0] + 2 + 5 * (uint64_t) k);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wf_bounce<M, ENV>), dim3(blocks), dim3(kBlock), lds_bytes(cx, h), s, cx, tv, inv_spp, img, dimg, WH * 3,
                               st[(k - 1) & 1], cnt + k, st[k & 1], cnt + k + 1, k + 1 < depth ? 1 : 0, h->d_counters);
            HIP_TRY(hipGetLastError());
        }
    }
    return 0;
}

// Strategy choice.  Measured on MI355X (tools/perf_cases.py): in a closed box almost every path
// survives, compaction buys nothing and the fused kernel is 1.1-1.9x faster; in an open scene
// (bunny_light: 2.5 of 7 possible rays per path) the wavefront is 1.35x faster.  The library keeps
// the survival ratio of the previous PathTracer call on this handle and switches on it.
bool use_wavefront(const psdr_scene_s *h, const psdr_render_opts *o) {
    if (o->integrator != PSDR_INTEGRATOR_PATH || o->max_depth > 250) return false;
    if (o->flags & PSDR_FLAG_FUSED) return false;
    if (o->flags & PSDR_FLAG_WAVEFRONT) return true;
    return o->max_depth >= 2 && h->path_survival >= 0.f && h->path_survival < 0.55f;
}

template <int K, bool ENV>
int render_fwd(psdr_scene_s *h, const psdr_render_opts *o, const psdr_tangents *tangents, float *img, float *dimg, hipStream_t s) {
    const long long WH = (long long) h->desc.width * h->desc.height;
    TangentView<K, ENV> tv;
    for (int k = 0; k < K; ++k) tv.t[k] = tangents[k];
    HIP_TRY(hipMemsetAsync(img, 0, sizeof(float) * WH * 3, s));
    HIP_TRY(hipMemsetAsync(dimg, 0, sizeof(float) * WH * 3 * K, s));
    // geometry stays in plain fp32 when only material / emitter tables carry tangents
    bool geo = false;
    for (int k = 0; k < K; ++k) geo = geo || tangents[k].d_tri_info || tangents[k].d_cam_to_world;
    if (geo) { if (int rc = run_camera<Dual<K>, Dual<K>, ENV>(h, o, tv, img, dimg, s)) return rc; }
    else if (use_wavefront(h, o)) { if (int rc = run_camera_wavefront<Dual<K>, ENV>(h, o, tv, img, dimg, s)) return rc; }
    else { if (int rc = run_camera<float, Dual<K>, ENV>(h, o, tv, img, dimg, s)) return rc; }
    if (o->sppe > 0 && o->sppe_end > o->sppe_begin && h->desc.num_prim_edges > 0) {
        LaunchCtx cx;
        if (int rc = make_ctx(h, o, 1, cx)) return rc;
        const long long i0 = WH * o->sppe_begin, n = WH * (o->sppe_end - o->sppe_begin);
        h->slots[1] += (uint64_t) n;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_primary_edge<K, ENV>), dim3(launch_blocks(h, n)), dim3(kBlock), lds_bytes(cx, h), s, cx, tv, i0, n,
                           1.f / (float) o->sppe, dimg, WH * 3, h->d_counters);
        HIP_TRY(hipGetLastError());
    }
    if (o->sppse > 0 && o->sppse_end > o->sppse_begin && h->desc.num_sec_edges > 0 && o->integrator == PSDR_INTEGRATOR_DIRECT) {
        LaunchCtx cx;
        if (int rc = make_ctx(h, o, 2, cx)) return rc;
        const long long i0 = WH * o->sppse_begin, n = WH * (o->sppse_end - o->sppse_begin);
        h->slots[2] += (uint64_t) n;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_secondary_edge<K, ENV>), dim3(launch_blocks(h, n)), dim3(kBlock), lds_bytes(cx, h), s, cx, tv, i0, n,
                           1.f / (float) o->sppse, dimg, WH * 3, h->d_counters);
        HIP_TRY(hipGetLastError());
    }
    return 0;
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
    return L;
}

int begin_call(psdr_scene_s *h, hipStream_t s) {
    h->slots[0] = h->slots[1] = h->slots[2] = 0; h->last_path_depth = 0;
    HIP_TRY(hipMemsetAsync(h->d_counters, 0, sizeof(unsigned long long) * 4, s));
    return 0;
}

}  // namespace

// ================================================================================= C ABI
namespace {
template <bool ENV>
int render_rev(psdr_scene_s *h, const psdr_render_opts *o, const float *adj_img, float *out_img, const psdr_grads *grads, hipStream_t s) {
    const long long WH = (long long) h->desc.width * h->desc.height;
    if (out_img) HIP_TRY(hipMemsetAsync(out_img, 0, sizeof(float) * WH * 3, s));
    DeviceSink<ENV> sink{}; sink.g = *grads; sink.L = make_sink_layout(h, grads);
    const int nsp = o->spp_end - o->spp_begin;
    if (o->spp > 0 && nsp > 0) {
        LaunchCtx cx;
        if (int rc = make_ctx(h, o, 0, cx)) return rc;
        const long long n = WH * nsp;
        h->slots[0] += (uint64_t) n;
        const int depth = o->integrator == PSDR_INTEGRATOR_PATH ? std::min(o->max_depth, kMaxRevDepth) : 1;
        const int rec_bytes = depth * 6 * kBlock * 4;
        plan_lds(h, cx, rec_bytes + kSinkCacheWords * 4);          // stage less of the scene: the record + cache live in LDS too
        cx.off_pathrec = lds_bytes(cx, h);
        hipLaunchKernelGGL(k_camera_rev<ENV>, dim3(launch_blocks(h, n)), dim3(kBlock), lds_bytes(cx, h) + rec_bytes, s, cx, sink, o->spp, o->spp_begin, nsp, n,
                           1.f / (float) o->spp, adj_img, out_img, h->d_counters);
        HIP_TRY(hipGetLastError());
    }
    if (o->sppe > 0 && o->sppe_end > o->sppe_begin && h->desc.num_prim_edges > 0 && grads->g_prim_edge) {
        LaunchCtx cx;
        if (int rc = make_ctx(h, o, 1, cx)) return rc;
        const long long i0 = WH * o->sppe_begin, n = WH * (o->sppe_end - o->sppe_begin);
        h->slots[1] += (uint64_t) n;
        hipLaunchKernelGGL(k_primary_edge_rev<ENV>, dim3(launch_blocks(h, n)), dim3(kBlock), lds_bytes(cx, h), s, cx, sink, i0, n, 1.f / (float) o->sppe, adj_img,
                           h->d_counters);
        HIP_TRY(hipGetLastError());
    }
    if (o->sppse > 0 && o->sppse_end > o->sppse_begin && h->desc.num_sec_edges > 0 && o->integrator == PSDR_INTEGRATOR_DIRECT) {
        LaunchCtx cx;
        if (int rc = make_ctx(h, o, 2, cx)) return rc;
        const long long i0 = WH * o->sppse_begin, n = WH * (o->sppse_end - o->sppse_begin);
        h->slots[2] += (uint64_t) n;
        hipLaunchKernelGGL(k_secondary_edge_rev<ENV>, dim3(launch_blocks(h, n)), dim3(kBlock), lds_bytes(cx, h), s, cx, sink, i0, n, 1.f / (float) o->sppse,
                           adj_img, h->d_counters);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}
}  // namespace

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
    hipError_t e = hipMalloc(&h->d_counters, sizeof(unsigned long long) * 4);
    if (e != hipSuccess) { delete h; return fail(std::string("hipMalloc: ") + hipGetErrorString(e)); }
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) h->num_cus = prop.multiProcessorCount;
    *out = h;
    return 0;
}

int psdr_scene_destroy(psdr_scene_t h) {
    if (!h) return 0;
    if (h->d_nodes) (void) hipFree(h->d_nodes);
    if (h->d_btris) (void) hipFree(h->d_btris);
    if (h->d_counters) (void) hipFree(h->d_counters);
    if (h->d_ws) (void) hipFree(h->d_ws);
    if (h->d_hot_map) (void) hipFree(h->d_hot_map);
    if (h->d_hot_tris) (void) hipFree(h->d_hot_tris);
    delete h;
    return 0;
}

int psdr_scene_set_tables(psdr_scene_t h, const psdr_scene_desc *desc) {
    if (!h || !desc) return fail("psdr_scene_set_tables: null argument");
    if (desc->num_tris <= 0 || !desc->tri_info || !desc->tri_mesh) return fail("Missing meshes!");
    if (!desc->cam) return fail("Missing sensor!");
    if (h->have_tables && (desc->tri_info != h->desc.tri_info || desc->num_tris != h->desc.num_tris)) h->have_bvh = false;
    h->desc = *desc;
    // no environment map unless its record is given (a zero-initialised desc means "none")
    if (!h->desc.env_f) h->desc.env_emitter = -1;
    if (h->desc.env_emitter >= 0) {
        if (h->desc.env_emitter >= h->desc.num_emitters || !h->desc.env_cmf || !h->desc.env_pmf || h->desc.env_reso[0] <= 0 ||
            h->desc.env_reso[1] <= 0 || h->desc.env_tex[1] < 2 || h->desc.env_tex[2] < 2)
            return fail("psdr_scene_set_tables: inconsistent environment-map record");
    }
    h->have_tables = true;
    return 0;
}

int psdr_bvh_build(psdr_scene_t h, void *stream) {
    if (!h || !h->have_tables) return fail("Scene not loaded yet!");
    hipStream_t s = (hipStream_t) stream;
    const int T = h->desc.num_tris;
    std::vector<float> rows((size_t) T * PSDR_TRI_STRIDE);
    HIP_TRY(hipMemcpyAsync(rows.data(), h->desc.tri_info, rows.size() * sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    h->emitter_i.assign((size_t) std::max(h->desc.num_emitters, 0) * PSDR_EMITTER_I_STRIDE, 0);
    if (h->desc.num_emitters > 0 && h->desc.emitter_i)
        HIP_TRY(hipMemcpy(h->emitter_i.data(), h->desc.emitter_i, h->emitter_i.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    Builder b;
    int32_t root = 0;
    if (const char *err = b.run(rows.data(), T, root)) return fail(err);
    if (b.nodes.size() > h->cap_nodes) {
        if (h->d_nodes) (void) hipFree(h->d_nodes);
        h->cap_nodes = std::max<size_t>(b.nodes.size(), 16);
        HIP_TRY(hipMalloc(&h->d_nodes, h->cap_nodes * sizeof(BvhNode)));
    }
    if (b.btris.size() > h->cap_btris) {
        if (h->d_btris) (void) hipFree(h->d_btris);
        h->cap_btris = b.btris.size();
        HIP_TRY(hipMalloc(&h->d_btris, h->cap_btris * sizeof(float4)));
    }
    if (!b.nodes.empty()) HIP_TRY(hipMemcpyAsync(h->d_nodes, b.nodes.data(), b.nodes.size() * sizeof(BvhNode), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(h->d_btris, b.btris.data(), b.btris.size() * sizeof(float4), hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));     // host vectors die at return
    {   // hot rows of the reverse-mode gradient cache: emitter triangles first, then by decreasing area
        constexpr int kMaxHotRows = 200;
        std::vector<int32_t> map((size_t) T, -1), tris;
        auto add = [&](int t) { if (t >= 0 && t < T && map[t] < 0 && (int) tris.size() < kMaxHotRows) { map[t] = (int32_t) tris.size(); tris.push_back(t); } };
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
        HIP_TRY(hipMemcpy(h->d_hot_map, map.data(), map.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(h->d_hot_tris, tris.data(), tris.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        h->hot_rows = (int) tris.size();
    }
    h->root = root;
    h->bvh_depth = b.max_depth; h->num_nodes = (int) b.nodes.size(); h->num_btris = (int) b.btris.size() / 3;
    h->have_bvh = true;
    return 0;
}

int psdr_trace(psdr_scene_t h, int32_t m, const float *ox, const float *oy, const float *oz, const float *dx, const float *dy,
               const float *dz, const float *tmax, int32_t *out_shape, int32_t *out_tri, float *out_u, float *out_v, void *stream) {
    if (!h || !h->have_tables) return fail("Scene not loaded yet!");
    if (!h->have_bvh) return fail("Input scene must be configured!");
    if (m <= 0) return 0;
    LaunchCtx cx{};
    cx.sc.d = h->desc; cx.sc.nodes = h->d_nodes; cx.sc.btris = h->d_btris; cx.sc.root = h->root;
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
    if (h->desc.env_emitter >= 0) {
        const TangentView<0, true> tv0{};
        if (use_wavefront(h, o)) return run_camera_wavefront<float, true>(h, o, tv0, out_img, nullptr, s);
        return run_camera<float, float, true>(h, o, tv0, out_img, nullptr, s);
    }
    const TangentView<0, false> tv0{};
    if (use_wavefront(h, o)) return run_camera_wavefront<float, false>(h, o, tv0, out_img, nullptr, s);
    return run_camera<float, float, false>(h, o, tv0, out_img, nullptr, s);
}

int psdr_render_d_fwd(psdr_scene_t h, const psdr_render_opts *o, int32_t K, const psdr_tangents *tangents, float *out_img,
                      float *out_dimg, void *stream) {
    if (!h || !o || !out_img || !out_dimg || !tangents) return fail("psdr_render_d_fwd: null argument");
    if (!h->have_tables) return fail("Scene not loaded yet!");
    if (int rc = check_counts(h, o)) return rc;
    hipStream_t s = (hipStream_t) stream;
    if (int rc = begin_call(h, s)) return rc;
    const bool env = h->desc.env_emitter >= 0;
    switch (K) {
        case 1: return env ? render_fwd<1, true>(h, o, tangents, out_img, out_dimg, s) : render_fwd<1, false>(h, o, tangents, out_img, out_dimg, s);
        case 3: return env ? render_fwd<3, true>(h, o, tangents, out_img, out_dimg, s) : render_fwd<3, false>(h, o, tangents, out_img, out_dimg, s);
        default: return fail("psdr_render_d_fwd: K must be 1 or 3");
    }
}

int psdr_render_d_rev(psdr_scene_t h, const psdr_render_opts *o, const float *adj_img, float *out_img, const psdr_grads *grads,
                      void *stream) {
    if (!h || !o || !adj_img || !grads) return fail("psdr_render_d_rev: null argument");
    if (!h->have_tables) return fail("Scene not loaded yet!");
    if (int rc = check_counts(h, o)) return rc;
    if (o->integrator == PSDR_INTEGRATOR_PATH && o->max_depth > kMaxRevDepth) return fail("psdr_render_d_rev: max_depth > 8 is not supported");
    hipStream_t s = (hipStream_t) stream;
    if (int rc = begin_call(h, s)) return rc;
    return h->desc.env_emitter >= 0 ? render_rev<true>(h, o, adj_img, out_img, grads, s) : render_rev<false>(h, o, adj_img, out_img, grads, s);
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
    if (h->desc.env_emitter >= 0)
        hipLaunchKernelGGL(k_guide<true>, dim3(launch_blocks(h, n)), dim3(kBlock), lds_bytes(cx, h), s, cx, reso[0], reso[1], reso[2], reso[3], nrounds, n,
                           out_mass, h->d_counters);
    else
        hipLaunchKernelGGL(k_guide<false>, dim3(launch_blocks(h, n)), dim3(kBlock), lds_bytes(cx, h), s, cx, reso[0], reso[1], reso[2], reso[3], nrounds, n,
                           out_mass, h->d_counters);
    HIP_TRY(hipGetLastError());
    return 0;
}

int psdr_get_counters(psdr_scene_t h, uint64_t out[4]) {
    if (!h || !out) return fail("psdr_get_counters: null argument");
    unsigned long long c[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpy(c, h->d_counters, sizeof(c), hipMemcpyDeviceToHost));
    out[0] = c[0]; out[1] = h->slots[0]; out[2] = h->slots[1]; out[3] = h->slots[2];
    if (h->last_path_depth > 0 && h->slots[0] > 0 && h->slots[1] == 0 && h->slots[2] == 0)
        h->path_survival = (float) ((double) c[0] / ((double) h->slots[0] * (1.0 + 2.0 * h->last_path_depth)));
    return 0;
}

}  // extern "C"
