// psdr_kernels.h -- the gfx950 kernels (all hand-written HIP for CDNA4, wave64) and their launch drivers,
// templated on the scene flag set FL (psdr_device.h TangentView).  Included by psdr_variant.hip once per
// flag set.
//   k_camera<G,M,INTEG,FL>  one lane = one camera sample slot: raygen + Li + segmented wave splat
//   k_wf_camera / k_wf_bounce  the same PathTracer as a wavefront over SoA path-state streams
//   k_primary_edge<K,FL>    one lane = one primary-edge slot (two detached Li evaluations)
//   k_secondary_edge<K,FL>  one lane = one secondary-edge slot (3 rays + boundary integrand)
//   k_guide<FL>             guiding-grid mass accumulation
//   k_camera_rev / k_primary_edge_rev / k_secondary_edge_rev   reverse mode (gradient scatter-add)
// Traversal stacks live in LDS (one column per lane); image accumulation uses a segmented
// wave reduction followed by one hardware f32 atomic per (pixel run, channel).
#pragma once
#include "psdr_host.h"

namespace {
using namespace psdr_host;

__device__ __forceinline__ float wave_shfl_down(float v, int off) { return __shfl_down(v, off, 64); }

// Segmented sum over a wave for NON-DECREASING integer keys (camera slots are pixel-major, so a
// wave covers a few consecutive pixels).  After the loop the first lane of every key run holds
// the run total.  All 64 lanes must call this.
// Sum over all 64 lanes on the DPP path (no LDS round trips): an inclusive scan inside every row of 16 (row_shr 1, 2, 4, 8), then the row totals
// carried into the following rows (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3): lane 63 holds the total.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ float dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true));
}
__device__ __forceinline__ float wave_total(float v) {
    v = dpp_add<0x111, 0xf>(v); v = dpp_add<0x112, 0xf>(v); v = dpp_add<0x114, 0xf>(v); v = dpp_add<0x118, 0xf>(v);
    v = dpp_add<0x142, 0xa>(v); v = dpp_add<0x143, 0xc>(v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
#ifndef PSDR_DPP_ALWAYS
#define PSDR_DPP_ALWAYS 0                   // test builds (variants/): the DPP totals in every instance
#endif
template <int N, bool DPP = (N <= 6 || PSDR_DPP_ALWAYS)> __device__ __forceinline__ bool wave_segmented_sum(int key, float (&v)[N]) {
    const int lane = threadIdx.x & 63;
    // the common case of the camera kernels: spp is a multiple of 64 and the whole wave sits on ONE pixel -- six DPP adds per value instead of six
    // ds_bpermute round trips (each an LDS instruction + its address arithmetic + the wait)
    // (DPP by default only for few values: a PERFORMANCE choice -- the K = 3 dual kernels spill hundreds of registers and gain nothing from 12 more
    // live values at the splat.  Round 3 read a wrong gradient of a build with more DPP totals as "DPP returns stale registers in a spilled kernel";
    // round 4 found the cause elsewhere: a spill-placement defect of the compiler that any change of register pressure can expose -- DESIGN.md
    // "the order-dependent gradient", guarded at build time by tools/check_spill_exec.py, not by this flag)
    if (DPP && __ballot(key != __builtin_amdgcn_readfirstlane(key)) == 0ull) {
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = wave_total(v[i]);
        return lane == 0;
    }
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
template <int N, bool DPP = (N <= 6 || PSDR_DPP_ALWAYS)> __device__ __forceinline__ bool wave_run_sum(int key, float (&v)[N]) {
    const int lane = threadIdx.x & 63;
    const int prev = __shfl_up(key, 1, 64);
    const bool head = lane == 0 || prev != key;
    const unsigned long long heads = __ballot(head);
    const int seg = __popcll(heads & (~0ull >> (63 - lane)));       // run index: non-decreasing
    wave_segmented_sum<N, DPP>(seg, v);
    return head;
}

// Rays traced by the launch: one atomic per wave at kernel end, spread over kRayCounters counters on their own
// 128-byte lines (same-address L2 atomics serialise at 3-5 ns each; psdr_get_counters sums them).
__device__ __forceinline__ float *dyn_lds_floats(int byte_offset) {
#if defined(__HIP_DEVICE_COMPILE__)
    return reinterpret_cast<float *>(psdr_dyn_lds + byte_offset);
#else
    (void) byte_offset; return nullptr;
#endif
}
__device__ __forceinline__ void count_rays(unsigned long long *counters, uint32_t nrays) {
    uint32_t s = nrays;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(counters + (blockIdx.x % kRayCounters) * kRayCounterStride, (unsigned long long) s);
}

// Hit rows of a probe / trace / final launch (see k_se_probe): what the final pass of a kernel reads.
struct ProbeView { const float4 *hit; const uint32_t *mask; };
template <int NR>
__device__ __forceinline__ void probe_load(TraversalStack &st, const ProbeView &pv, long long slot, uint32_t m) {
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        st.pre[r].tri = -1;
        if ((m >> r) & 1u) { const float4 h = pv.hit[slot * NR + r]; st.pre[r] = Hit{__float_as_int(h.x), h.y, h.z, h.w}; }
    }
}

// ------------------------------------------------------------------------------ k_camera
// n = W*H*nsp slots of this shard, pixel-major: slot j -> pixel j / nsp, sample s_begin + j % nsp.
// n <= INT_MAX (check_counts), so the division by the run-time nsp is a 32 x 32 -> 64 bit multiply by a host-computed
// reciprocal and a shift (round-up method for 31-bit dividends: l = ceil(log2 d), m = floor(2^(31+l) / d) + 1 < 2^32,
// q = (j * m) >> (31 + l), exact for every j < 2^31): 4 VALU instructions per slot where the expanded 32-bit division took ~30.
struct SlotDiv {
    uint32_t d, m; int sh;
    SlotDiv() = default;
    SlotDiv(int nsp) : d((uint32_t) nsp) {
        int l = 0;
        while ((1u << l) < d) ++l;
        m = (uint32_t) ((1ull << (31 + l)) / d + 1ull); sh = 31 + l;
    }
};
PSDR_HD void slot_to_pixel(long long j, const SlotDiv &nsp, int &pixel, int &s) {
    const uint32_t ju = (uint32_t) j, q = (uint32_t) (((uint64_t) ju * nsp.m) >> nsp.sh);
    pixel = (int) q; s = (int) (ju - q * nsp.d);
}
// Occupancy targets (waves per SIMD) of the camera kernel.  The kernel is latency/dependency bound
// (rocprof r01: 43 % of wave cycles waiting at 2 waves/SIMD), so trading registers for resident
// waves pays: measured 6.2 -> 4.6 ms (renderC, 4 waves) and 12.5 -> 7.0 ms (renderD K=3 material-only, 2 waves,
// no spills) on C2.  These are the defaults of the general variants; camera_waves() adds a wave where the
// leaner instances take it.
#ifndef PSDR_WAVES_C
#define PSDR_WAVES_C 4
#endif
#ifndef PSDR_WAVES_DM
#define PSDR_WAVES_DM 3
#endif
#ifndef PSDR_WAVES_C_NOTREE
// renderC of the PathTracer on a scene without a tree: SIX waves (80 VGPRs, 9 spilled).  The kernel waits -- wait_any 0.37 of its wave cycles at five waves, and 3.7 % fewer
// instructions bought nothing there (profiles/r06_pretest_tiny_abk.txt) -- so a sixth resident wave pays although it spills: C2 0.846 -> 0.808 ms; the price is scratch
// traffic, 6 -> 93 MB per launch in the counters.  Seven waves: 0.86 ms, 1.1 GB (profiles/r06_six_waves_ab.txt)
#define PSDR_WAVES_C_NOTREE 6
#endif
#ifndef PSDR_WAVES_DG
#define PSDR_WAVES_DG 2
#endif
// geometry duals: K = 1 fits 3 waves/SIMD without spills in DirectIntegrator form (C2 direct 2.5 -> 2.0 ms, path3
// 6.7 -> 5.8 ms); K = 3 would spill > 400 registers there and stays at PSDR_WAVES_DG
// the plain diffuse / area-light variant (FL == 0) is lean enough for one more wave: renderC 5 waves/SIMD (C2
// PathTracer(3) 2.06 -> 1.89 ms), material duals of the PathTracer 4 (2.85 -> 2.66 ms); the rough-conductor
// variants lose 10 % there (C5 renderC 5.0 -> 5.5 ms) and the DirectIntegrator K = 3 instance 20 %
#ifndef PSDR_LOGD
#define PSDR_LOGD 1
#endif
#ifndef PSDR_LOGD_WAVES_K1
#define PSDR_LOGD_WAVES_K1 6       // resident waves per SIMD of the K = 1 log-derivative kernel on a scene without a tree (0: those of the dual-number kernel it stands in for: 5);
#endif                            // C2: 4 waves 1.05 ms, 5 (10 VGPRs spilled, 85-143 MB of counter traffic per launch) 0.94-0.95, 6 0.885-0.89 with 712 MB of scratch traffic (9 % of the HBM
                                  // peak for the launch's duration).  Round 5 kept 5 for the traffic figure; round 6 takes the time: the headline is what this kernel is for, the traffic is
                                  // reported beside it (roofline.traffic) -- profiles/r05_logd_waves.txt, profiles/r06_six_waves_ab.txt
template <class G, class R, int INTEG, int FL, bool NOTREE = false> constexpr int camera_waves() {
    constexpr bool lean = (FL & (kSceneEnv | kSceneRough)) == 0;          // plain diffuse / area light (with or without a two-level tree)
    if (!is_ad<R>()) return lean ? ((NOTREE && INTEG == PSDR_INTEGRATOR_PATH) ? PSDR_WAVES_C_NOTREE : PSDR_WAVES_C + 1) : PSDR_WAVES_C;
    // NOTREE (the launch serves a scene whose primitives all travel in the kernel arguments, run_camera): the rough-conductor PathTracer
    // spills 79 (geometry duals, K = 1) / 204 (material duals, K = 3) VGPRs at 3 waves, none at 2 -- without a tree walk whose latency the third
    // wave would hide, 2 waves are faster (cbox_rough: 8.15 -> 6.9 ms and 8.65 -> 6.4 ms); with trees the third wave wins (interior: 7.4
    // against 8.8 and 8.1 ms)
    constexpr bool rough_path_notree = NOTREE && (FL & kSceneRough) != 0 && INTEG == PSDR_INTEGRATOR_PATH;
    if (is_ad<G>()) return ad_traits<G>::K == 1 ? (rough_path_notree ? 2 : 3) : PSDR_WAVES_DG;
    // PathTracer material duals of the lean variant: K = 1 fits 4 waves / SIMD without spilling (C2 1.67 ms); K = 3 spills
    // 150 VGPRs there and runs faster at 3 (3.37 -> 2.99 ms)
    if (rough_path_notree && ad_traits<R>::K == 3) return 2;
    // the instances of a scene without a tree (NOTREE, lean) take one more: PathTracer K = 1 at 5 waves (C2 1.215 -> 1.163 ms, no spills), K = 3 at 4
    // (1.642 -> 1.624 ms)
    if (lean && NOTREE && INTEG == PSDR_INTEGRATOR_PATH) return ad_traits<R>::K == 1 ? PSDR_WAVES_DM + 2 : PSDR_WAVES_DM + 1;
    return (lean && INTEG == PSDR_INTEGRATOR_PATH && ad_traits<R>::K == 1) ? PSDR_WAVES_DM + 1 : PSDR_WAVES_DM;
}
// Log-derivative launches (psdr_device.h li_path_logd): which k_camera instances they stand in for, and the gate word among the ray counters' padding
// (zeroed with them at the start of every call; k_logd_check writes 1 = "every texel that carries a tangent has a non-zero albedo")
constexpr int kLogdGateWord = kRayCounterStride - 1;
template <class G, class R, int INTEG, int FL> constexpr bool logd_instance() {
    return PSDR_LOGD && !is_ad<G>() && is_ad<R>() && INTEG == PSDR_INTEGRATOR_PATH && (FL & (kSceneRough | kSceneEnv | kScenePre)) == 0;
}
template <class G, class R, int INTEG, int FL, bool NOTREE = false>
__global__ __launch_bounds__(kBlock, (camera_waves<G, R, INTEG, FL, NOTREE>())) void k_camera(LaunchCtx cx, TV<R, FL> tv, int spp, int s_begin, SlotDiv nsp, long long n, float inv_spp,
                                                   float *__restrict__ img, float *__restrict__ dimg, long long plane,
                                                   unsigned long long *counters, long long j0, ProbeView pv, int own) {
    // own: every pixel's samples of this launch sit in ONE wave (64 % samples per pixel == 0, run_camera) -- the run's head lane STORES the pixel (the image was
    // zeroed, the edge terms add to the derivative images afterwards) instead of three to twelve returning L2 atomics of 32 bytes each: C2 1.57 M atomics per launch
    constexpr int K = ad_traits<R>::K;
    constexpr int NV = 3 * (1 + K);
    // the dual-number PathTracer instances a log-derivative launch stands in for (k_camera_logd below): both are launched, the gate word says which one runs
    if constexpr (logd_instance<G, R, INTEG, FL>()) { if (counters[kLogdGateWord] == 1ull) return; }
    TraversalStack st; setup_lds(cx, st, tv);
    uint32_t nrays = 0;
    const long long nceil = (n + kBlock - 1) / kBlock * kBlock;
    for (long long j = (long long) blockIdx.x * kBlock + threadIdx.x; j < nceil; j += (long long) gridDim.x * kBlock) {
        const bool in = j < n;
        int pixel = 0x7fffffff, s_in = 0;
        if (in) slot_to_pixel(j0 + j, nsp, pixel, s_in);
        float v[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = 0.f;
        // final pass of a probe / trace / final launch (DirectIntegrator(1, 1) on a two-level scene): the slot's tree hits
        if constexpr ((FL & kScenePre) != 0) { if (in) probe_load<3>(st, pv, j, pv.mask[j]); }
        if (in) {
            const int s = s_begin + s_in;
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
            if (own) {
                p[0] = v[0]; p[1] = v[1]; p[2] = v[2];
#pragma unroll
                for (int k = 0; k < K; ++k) { float *q = dimg + (size_t) k * plane + (size_t) pixel * 3; q[0] = v[3 + 3 * k]; q[1] = v[4 + 3 * k]; q[2] = v[5 + 3 * k]; }
            } else {
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
    }
    count_rays(counters, nrays);
}

// ---------------------------------------------------------------- log-derivative launches (round 5)
// k_logd_check: the gate.  One thread per texel: a texel some tangent set moves must have an albedo the quotient d rho / rho can be formed with.
template <int K>
__global__ __launch_bounds__(kBlock) void k_logd_check(const float *__restrict__ texels, TangentView<K, 0> tv, int n, unsigned long long *counters, int *bad) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    bool b = false;
    if (i < n) {
        bool moved = false;
#pragma unroll
        for (int k = 0; k < K; ++k) moved = moved || tv.t[k].d_texels[i] != 0.f;
        b = moved && !(fabsf(texels[i]) > 1e-20f);
    }
    if (__ballot(b) != 0ull && (threadIdx.x & 63) == 0) atomicOr(bad, 1);
}
__global__ void k_logd_gate(unsigned long long *counters, const int *bad) { if (threadIdx.x == 0 && blockIdx.x == 0) counters[kLogdGateWord] = *bad ? 0ull : 1ull; }
// the camera kernel of a PathTracer whose tangents sit on diffuse albedo texels only: k_camera<float, Dual<K>, PATH, FL> with the estimator on plain floats
template <int K, int FL, bool NOTREE>
__global__ __launch_bounds__(kBlock, ((PSDR_LOGD_WAVES_K1 > 0 && K == 1 && NOTREE) ? PSDR_LOGD_WAVES_K1 : camera_waves<float, Dual<K>, PSDR_INTEGRATOR_PATH, FL, NOTREE>())) void k_camera_logd(LaunchCtx cx, TV<Dual<K>, FL> tv, int spp, int s_begin, SlotDiv nsp,
                                                   long long n, float inv_spp, float *__restrict__ img, float *__restrict__ dimg, long long plane, unsigned long long *counters, int own) {
    constexpr int NV = 3 * (1 + K);
    if (counters[kLogdGateWord] != 1ull) return;
    TraversalStack st; setup_lds(cx, st, tv);
    uint32_t nrays = 0;
    const long long nceil = (n + kBlock - 1) / kBlock * kBlock;
    for (long long j = (long long) blockIdx.x * kBlock + threadIdx.x; j < nceil; j += (long long) gridDim.x * kBlock) {
        const bool in = j < n;
        int pixel = 0x7fffffff, s_in = 0;
        if (in) slot_to_pixel(j, nsp, pixel, s_in);
        float v[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = 0.f;
        if (in) {
            const uint64_t slot = (uint64_t) pixel * (uint64_t) spp + (uint64_t) (s_begin + s_in);
            const Vec3<Dual<K>> r = camera_sample_logd<K>(cx.sc, tv, st, cx.lp, cx.jump, pixel, slot, nrays);
            v[0] = r.x.v * inv_spp; v[1] = r.y.v * inv_spp; v[2] = r.z.v * inv_spp;
#pragma unroll
            for (int k = 0; k < K; ++k) { v[3 + 3 * k] = r.x.d[k] * inv_spp; v[4 + 3 * k] = r.y.d[k] * inv_spp; v[5 + 3 * k] = r.z.d[k] * inv_spp; }
        }
        const bool head = wave_segmented_sum<NV>(pixel, v);
        if (head && in) {
            float *p = img + (size_t) pixel * 3;
            if (own) {
                p[0] = v[0]; p[1] = v[1]; p[2] = v[2];
#pragma unroll
                for (int k = 0; k < K; ++k) { float *q = dimg + (size_t) k * plane + (size_t) pixel * 3; q[0] = v[3 + 3 * k]; q[1] = v[4 + 3 * k]; q[2] = v[5 + 3 * k]; }
            } else {
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
    }
    count_rays(counters, nrays);
}

// ----------------------------------------------------------------------- wavefront mode
// PathTracer as a wavefront: stage 0 (camera ray, primary vertex) then one kernel per bounce over SoA
// path-state streams in HBM.  Live paths are compacted between stages with a wave ballot + prefix
// popcount and one atomic per wave on a stream counter, so every bounce kernel runs on dense waves.
// Record = pixel, slot, triangle, (u,v), arrival direction, throughput (+K tangents): 44 + 12 K bytes, all
// streams coalesced.
// The stream is split into kWfSub SUB-STREAMS, each with its own counter on its own 128-byte line; block b
// appends to (and, in the next stage, consumes) sub-stream b % kWfSub.  With ONE counter the 262 k waves of a
// C2 stage serialised on a single L2 atomic (3-5 ns each: ~1 ms per stage, C2 PathTracer(3) 6.6 ms);
// wave-private segments without any atomic were measured too: as fast on C2 (3.0 ms) but 10 % slower on
// open scenes, where the waves' survivor counts differ and the next stage inherits the imbalance.
// Two-level scenes (SceneView::n_blas > 0) run the wavefront BINNED BY COST CLASS: in a room most bounce rays only meet
// the walls (uniform, ~200 VALU) while the few that enter an object's box walk its tree (1 500 - 4 000), and a wave
// pays for its most expensive lane -- SIMD efficiency 16-23 % on the bounce rays of cbox_bunny (tools/simd_sim).  The
// sample streams are stateless, so the two rays a surviving path will trace at its NEXT vertex are known when its
// record is written: classify_next samples them and tests them against the tree boxes (2 bits: BSDF ray / light ray
// enters a box), and the record goes to the sub-stream of its class.  The next stage consumes the sub-streams chunk
// by chunk (256 records, expensive classes first, dynamic grabs of kWfGrab chunks from one counter): its waves are
// class-pure -- three quarters of them never touch a tree, the rest walk with all lanes busy.
constexpr int kWfGroups = 16, kWfClasses = 4, kWfGrab = 8;
constexpr int kWfStageInts = (kWfSub + 1) * kWfCountStride;       // per stage: 64 sub-stream counters + the chunk-grab counter
struct PathStream {
    int32_t *pixel; uint32_t *slot; int32_t *tri; float *hu, *hv; float *dir; float *beta;   // dir: [3][cap], beta: [3(1+K)][cap]
    float *acc;           // [3(1+K)][cap] radiance gathered so far: a sample is splatted ONCE, when its path ends, so that a non-finite
                          // contribution zeroes the whole sample as the fused kernel and the reference do (integrator.cpp:87)
    long long cap;
    int32_t *count;       // [kWfSub * kWfCountStride] records in each sub-stream (+ the grab counter behind them)
    long long sub_cap;    // records per sub-stream
    int32_t binned;       // 1: sub-stream = (3 - class) * kWfGroups + chunk % kWfGroups, consumed by dynamic chunk grabs
    // traced wavefront (kScenePre instances): hit[2 * i + ray] = closest TREE hit (tri, u, v, t) of ray 0 (BSDF-sampled) / 1 (light) of record i,
    // written by the dense trace kernel (psdr_hip.hip k_wf_trace) between the stage that pushed the record and the stage that consumes it;
    // bits 29-30 of tri[i] tell which of the two rays have one
    float4 *hit;
    // geometry-dual stages (k_wfg_*): `dir` holds the VALUE of the position of the vertex behind the record's (the film position (sx, sy, -) in the records the
    // camera stage pushes), prev_t [3 K][cap] its tangents
    float *prev_t;
};
constexpr int kWfClsShift = 29;
constexpr int32_t kWfTriMask = (1 << kWfClsShift) - 1;
// Requests of the dense trace kernel: the rays of the NEXT stage that enter a tree box, as (o | dest), (d | -) rows in kWfSub sub-queues
// (block b appends to queue b % kWfSub, one atomic per wave); dest = 2 * record + ray addresses PathStream::hit of the stream the record went to.
struct TraceQueue {
    float4 *req;          // [2 * kWfSub * sub_cap]
    int32_t *count;       // [kWfSub * kWfCountStride]
    long long sub_cap;    // requests per sub-queue
};
// REC instances of the traced wavefront = the VALUE SWEEP of a split reverse launch (render_rev): every stage leaves, in the per-path record
// the adjoint kernel reads (psdr_reverse.h RevDisk, cf format), what it evaluated -- the camera stage the primary triangle, bounce stage k
// (c_k, f_k) and the triangles the vertex' two rays arrived at, the stage a path ends in its vertex count (or -1: a non-finite sample has no
// gradient, integrator.cpp:87).  Column of a path = its slot index inside the chunk of slots the launch serves.
struct WfRec {
    float *disk; long long stride;      // word w of column c at disk[w * stride + c]
    int spp, s_begin, nsp; long long j0;
    int stage;                          // vertex this bounce stage evaluates
    __device__ __forceinline__ long long column(int pixel, uint32_t slot) const {
        return (long long) pixel * nsp + (long long) (slot - (uint32_t) pixel * (uint32_t) spp - (uint32_t) s_begin) - j0;
    }
};

template <class M>
__device__ __forceinline__ void stream_write(const PathStream &out, long long i, int pixel, uint32_t slot, const Its<float> &next, const Vec3f &dir, const Vec3<M> &beta,
                                             const Vec3<M> &acc, int cls = 0) {
    constexpr int K = ad_traits<M>::K;
    out.pixel[i] = pixel; out.slot[i] = slot; out.tri[i] = next.tri | (cls << kWfClsShift); out.hu[i] = next.hu; out.hv[i] = next.hv;
    out.dir[i] = dir.x; out.dir[out.cap + i] = dir.y; out.dir[2 * out.cap + i] = dir.z;
    out.beta[i] = val(beta.x); out.beta[out.cap + i] = val(beta.y); out.beta[2 * out.cap + i] = val(beta.z);
    out.acc[i] = val(acc.x); out.acc[out.cap + i] = val(acc.y); out.acc[2 * out.cap + i] = val(acc.z);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        out.beta[(3 + 3 * k) * out.cap + i] = tangent(beta.x, k); out.beta[(4 + 3 * k) * out.cap + i] = tangent(beta.y, k);
        out.beta[(5 + 3 * k) * out.cap + i] = tangent(beta.z, k);
        out.acc[(3 + 3 * k) * out.cap + i] = tangent(acc.x, k); out.acc[(4 + 3 * k) * out.cap + i] = tangent(acc.y, k);
        out.acc[(5 + 3 * k) * out.cap + i] = tangent(acc.z, k);
    }
}

template <class M>
__device__ __forceinline__ void stream_push(const PathStream &out, bool alive, int pixel, uint32_t slot, const Its<float> &next,
                                            const Vec3f &dir, const Vec3<M> &beta, const Vec3<M> &acc) {
    const unsigned long long mask = __ballot(alive);
    if (mask == 0ull) return;
    const int lane = threadIdx.x & 63, sub = blockIdx.x % kWfSub;
    int base = 0;
    if (lane == __ffsll((long long) mask) - 1) base = atomicAdd(out.count + sub * kWfCountStride, (int) __popcll(mask));
    base = __shfl(base, __ffsll((long long) mask) - 1, 64);
    if (!alive) return;
    const long long i = (long long) sub * out.sub_cap + base + __popcll(mask & ((1ull << lane) - 1ull));
    stream_write<M>(out, i, pixel, slot, next, dir, beta, acc);
}

// Cost class of the two rays the path will trace at the vertex `next` (arrival direction `dir`) in the stage whose
// sample streams start at `jump_next`: bit 0 = the BSDF-sampled ray enters a tree box, bit 1 = the light ray does.  The
// same vertex reconstruction, the same draws and the same sampling routines as direct_step, on plain floats; the box
// test ignores t_best (conservative: class 0 NEVER walks a tree).
// `next` is the complete vertex the current stage just produced (position, frame, wi): no need to rebuild it from the record as the
// next stage will -- a direction one ulp apart can at worst put a record into a cheaper class than its walk turns out to be.
// TRACED (the traced wavefront): the two directions are handed back -- they become the trace requests -- and the light ray is tested against the
// boxes without its length too: closest_hit looks for the closest hit along the whole ray, and so does the trace kernel.
template <bool TRACED = false, class TVT>
__device__ __forceinline__ int classify_next(const SceneView &sc, const TVT &, Rng rng, const Its<float> &next, const Vec3f &,
                                             Vec3f *d_bsdf = nullptr, Vec3f *d_light = nullptr, bool emitter_only = false, TraversalStack *stp = nullptr) {
    const TangentView<0, TVT::flags> tv0{};
    const Its<float> &its = next;
    const int bsdf_id = Tab<TVT::flags>::mesh_bsdf(sc, its.mesh);          // the staged copies (two-level instances), as every estimator reads them
    if (bsdf_id < 0) return 0;
    auto enters = [&](const Vec3f &o, const Vec3f &d, float tmax) {
        const Vec3f inv{1.f / d.x, 1.f / d.y, 1.f / d.z};
        bool any = false;
        for (int k = 0; k < sc.n_blas; ++k) { float te; any = any || blas_box(sc, k, o, inv, tmax, te); }
        return any;
    };
    // `rng` = the stream of this slot where the stage that produced `next` left it: the next stage's jump is this stage's plus the draws it
    // made (2 for the pixel jitter, 5 per path vertex: run_camera_wavefront), so the two TEA mixes of a second Rng::init (~290 VALU
    // instructions per record, a fifth of a bounce stage) are not repeated
    const float s[3] = {rng.next(), rng.next(), rng.next()};
    const float s0 = rng.next(), s1 = rng.next();
    int cls = 0;
    const Bsdf<float, float> bsdf(sc, tv0, bsdf_id);
    Vec3f wo_s; float pdf_s;
    if (bsdf.sample(sc, tv0, its, s, true, wo_s, pdf_s)) {
        const Vec3f d1 = its.sh.s * wo_s.x + its.sh.t * wo_s.y + its.sh.n * wo_s.z;
        // emitter_only (the stage that consumes the record has no use for the BSDF sample's hit unless it is an emitter: direct_step): a ray that meets no emitter
        // primitive is not traced there -- no request for it
        bool wanted = true;
        if constexpr (!TVT::has_env && TVT::forest && PSDR_EMITTER_PRETEST) {
            if (emitter_only && sc.emit_rows != 0u && stp != nullptr) wanted = closest_hit<false, 2, true>(sc, *stp, its.p, d1, INFINITY, -1, -1, 0, sc.emit_rows).tri >= 0;
        }
        if (wanted && enters(its.p, d1, INFINITY)) cls |= 1;
        if (TRACED) *d_bsdf = d1;
    }
    const PosSample<float> ps = sample_emitter_position<float>(sc, tv0, its.p, s0, s1, false);
    if (ps.valid) {
        Vec3f wo = ps.p - its.p;
        const float dist = sqrtf(fmaxf(dot(wo, wo), 0.f));
        wo = wo / dist;
        const bool lit = PSDR_SKIP_UNLIT ? (its.sh.to_local(wo).z > 0.f && its.wi.z > 0.f) : true;          // direct_step does not trace an unlit light sample
        if (lit && enters(its.p, wo, TRACED ? INFINITY : dist)) cls |= 2;
        if (TRACED) *d_light = wo;
    }
    return cls;
}

// Binned push: `chunk` = index of the 256-record chunk this workgroup is processing (spreads the records over
// kWfGroups sub-streams per class whatever block happens to process the chunk).
template <class M>
__device__ __forceinline__ void stream_push_binned(const PathStream &out, bool alive, int cls, long long chunk, int pixel, uint32_t slot, const Its<float> &next,
                                                   const Vec3f &dir, const Vec3<M> &beta, const Vec3<M> &acc) {
    const int lane = threadIdx.x & 63, group = (int) (chunk % kWfGroups);
#pragma unroll
    for (int c = 0; c < kWfClasses; ++c) {
        const bool mine = alive && cls == c;
        const unsigned long long mask = __ballot(mine);
        if (mask == 0ull) continue;
        const int sub = (kWfClasses - 1 - c) * kWfGroups + group;              // expensive classes first
        int base = 0;
        if (lane == __ffsll((long long) mask) - 1) base = atomicAdd(out.count + sub * kWfCountStride, (int) __popcll(mask));
        base = __shfl(base, __ffsll((long long) mask) - 1, 64);
        if (mine) stream_write<M>(out, (long long) sub * out.sub_cap + base + __popcll(mask & ((1ull << lane) - 1ull)), pixel, slot, next, dir, beta, acc);
    }
}

// -DPSDR_STAGE_CLOCKS (psdr_device.h): the marks of the traced bounce stage travel as an extra argument
#ifdef PSDR_STAGE_CLOCKS
#define PSDR_CLK_ARG , StageClk *clk = nullptr
#define PSDR_CLK_PASS , clk
#define PSDR_CLK_MARK(i) PSDR_CLK_MARK_P(clk, i)
#else
#define PSDR_CLK_ARG
#define PSDR_CLK_PASS
#define PSDR_CLK_MARK(i) do { } while (0)
#endif
// Traced wavefront: plain sub-streams (block b -> sub-stream b % kWfSub, like stream_push) + the trace requests of the record's two rays.
template <class M>
__device__ __forceinline__ void stream_push_traced(const PathStream &out, const TraceQueue &q, bool alive, int cls, int pixel, uint32_t slot, const Its<float> &next,
                                                   const Vec3f &dir, const Vec3<M> &beta, const Vec3<M> &acc, const Vec3f &d_bsdf, const Vec3f &d_light PSDR_CLK_ARG) {
    const unsigned long long mask = __ballot(alive);
    if (mask == 0ull) return;
    const int lane = threadIdx.x & 63, sub = blockIdx.x % kWfSub, leader = __ffsll((long long) mask) - 1;
    const unsigned long long m1 = __ballot(alive && (cls & 1)), m2 = __ballot(alive && (cls & 2));
    const int n1 = (int) __popcll(m1), n2 = (int) __popcll(m2);
    int base = 0, rbase = 0;
    if (lane == leader) {
        base = atomicAdd(out.count + sub * kWfCountStride, (int) __popcll(mask));
        if (n1 + n2 > 0) rbase = atomicAdd(q.count + sub * kWfCountStride, n1 + n2);
    }
    base = __shfl(base, leader, 64); rbase = __shfl(rbase, leader, 64);
    PSDR_CLK_MARK(6);
    if (!alive) return;
    const unsigned long long below = (1ull << lane) - 1ull;
    const long long i = (long long) sub * out.sub_cap + base + __popcll(mask & below);
    stream_write<M>(out, i, pixel, slot, next, dir, beta, acc, cls);
    if (cls & 1) {
        const long long r = (long long) sub * q.sub_cap + rbase + __popcll(m1 & below);
        q.req[2 * r] = float4{next.p.x, next.p.y, next.p.z, __int_as_float((int) (2 * i))};
        q.req[2 * r + 1] = float4{d_bsdf.x, d_bsdf.y, d_bsdf.z, __int_as_float(-1)};
    }
    if (cls & 2) {
        const long long r = (long long) sub * q.sub_cap + rbase + n1 + __popcll(m2 & below);
        q.req[2 * r] = float4{next.p.x, next.p.y, next.p.z, __int_as_float((int) (2 * i + 1))};
        q.req[2 * r + 1] = float4{d_light.x, d_light.y, d_light.z, __int_as_float(-1)};
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
    if (img == nullptr) return;                                     // a reverse launch that does not want the primal image (uniform)
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

#ifndef PSDR_WF_WAVES
#define PSDR_WF_WAVES 4
#endif
#ifndef PSDR_WF_WAVES_T
#define PSDR_WF_WAVES_T 4          // bounce stages of the traced wavefront (no tree walk in the kernel)
#endif

// TRACED (two-level scenes, run_camera_wavefront): the stage stops at the primary hit like the binned one, the records go to plain sub-streams and the
// two rays of bounce stage 0 that enter a tree box become requests of the dense trace kernel (stream_push_traced).
template <class M, int FL, bool TRACED = false, bool REC = false>
__global__ __launch_bounds__(kBlock, (is_ad<M>() ? 2 : PSDR_WF_WAVES)) void k_wf_camera(LaunchCtx cx, TV<M, FL> tv, int spp, int s_begin, SlotDiv nsp, long long j0, long long n,
                                                        float inv_spp, float *__restrict__ img, float *__restrict__ dimg, long long plane,
                                                        PathStream out, int want_next, unsigned long long *counters, RngJump jump_next, TraceQueue tq, WfRec wr) {
    TraversalStack st; setup_lds(cx, st, tv);
    uint32_t nrays = 0;
    // want_next: bit 0 = surviving paths are pushed to `out`; bit 1 = the stage that consumes them is the path's LAST (its BSDF sample matters only if it ends on an emitter)
    const bool next_last = (want_next & 2) != 0;
    want_next &= 1;
    const long long nceil = (n + kBlock - 1) / kBlock * kBlock;
    for (long long jj = (long long) blockIdx.x * kBlock + threadIdx.x; jj < nceil; jj += (long long) gridDim.x * kBlock) {
        const bool in = jj < n;
        const long long j = j0 + jj;
        int pixel = 0, s_in = 0;
        if (in) slot_to_pixel(j, nsp, pixel, s_in);
        Vec3<M> r = zero3<M>(), beta = zero3<M>();
        Its<float> next; next.tri = -1; next.hu = next.hv = 0.f;
        Vec3f origin(0.f), dir(0.f);
        bool alive = false;
        uint32_t slot = 0;
        Rng rng_next; rng_next.state = rng_next.inc = 0;
        bool primary_only = false;
        if constexpr (TRACED) primary_only = want_next != 0;
        else if constexpr ((FL & kSceneForest) != 0) primary_only = out.binned != 0 && want_next;
        if (in) {
            slot = (uint32_t) ((uint64_t) pixel * (uint64_t) spp + (uint64_t) (s_begin + s_in));
            if (primary_only) {
                r = wavefront_primary_vertex<M>(cx.sc, tv, st, cx.lp, cx.jump, pixel, slot, nrays, next, dir, alive, &rng_next);
                beta = Vec3<M>{M(1.f), M(1.f), M(1.f)};
            } else {
                r = wavefront_camera_vertex<M>(cx.sc, tv, st, cx.lp, cx.jump, pixel, slot, nrays, next, beta, origin, alive, &rng_next);
                if (alive) { Vec3f d = next.p - origin; const float t = norm(d); dir = d / t; }
            }
        }
        // a path that goes on carries its radiance along; one that ends here is splatted now
        const bool goes_on = want_next && alive;
        if constexpr (REC) { if (in) wr.disk[jj] = __int_as_float(alive ? next.tri : -1); }          // head word 0: the primary triangle (-1: no gradient)
        splat_runs<M>(pixel, in && !goes_on, zero_nonfinite(r), inv_spp, img, dimg, plane);
        if (want_next) {
            if constexpr (TRACED) {
                Vec3f d_bsdf(0.f), d_light(0.f);
                const int cls = alive ? classify_next<true>(cx.sc, tv, rng_next, next, dir, &d_bsdf, &d_light, next_last, &st) : 0;
                stream_push_traced<M>(out, tq, alive, cls, pixel, slot, next, dir, beta, r, d_bsdf, d_light);
            } else if constexpr ((FL & kSceneForest) != 0) {
                if (out.binned) stream_push_binned<M>(out, alive, alive ? classify_next(cx.sc, tv, rng_next, next, dir, nullptr, nullptr, next_last, &st) : 0, jj / kBlock, pixel, slot, next, dir, beta, r);
                else stream_push<M>(out, alive, pixel, slot, next, dir, beta, r);
            } else stream_push<M>(out, alive, pixel, slot, next, dir, beta, r);
        }
    }
    count_rays(counters, nrays);
}

// One bounce: block b consumes its share of sub-stream b % kWfSub (grid-stride over the blocks of that
// sub-stream; gridDim.x is a multiple of kWfSub) and appends the surviving paths to the same sub-stream of `out`.
// One record of a bounce stage: rebuild the vertex, direct step, splat, push the continuation.
// The value words of a stream record (14 coalesced loads).
struct WfRaw { int pixel; uint32_t slot; int tri; float hu, hv, dx, dy, dz, bx, by, bz, ax, ay, az; };
__device__ __forceinline__ WfRaw wf_load_raw(const PathStream &in, long long j, bool live) {
    WfRaw w{-1, 0u, 0, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (live) {
        w.pixel = in.pixel[j]; w.slot = in.slot[j]; w.tri = in.tri[j]; w.hu = in.hu[j]; w.hv = in.hv[j];
        w.dx = in.dir[j]; w.dy = in.dir[in.cap + j]; w.dz = in.dir[2 * in.cap + j];
        w.bx = in.beta[j]; w.by = in.beta[in.cap + j]; w.bz = in.beta[2 * in.cap + j];
        w.ax = in.acc[j]; w.ay = in.acc[in.cap + j]; w.az = in.acc[2 * in.cap + j];
    }
    return w;
}
template <class M, int FL, bool TRACED = false, bool REC = false>
__device__ __forceinline__ void wf_bounce_record(const LaunchCtx &cx, const TV<M, FL> &tv, TraversalStack &st, float inv_spp, float *__restrict__ img,
                                                 float *__restrict__ dimg, long long plane, const PathStream &in, const PathStream &out, int want_next,
                                                 const RngJump &jump_next, bool live, long long j, long long chunk, uint32_t &nrays, const TraceQueue &tq,
                                                 const WfRec &wr, const WfRaw &raw, bool next_last PSDR_CLK_ARG) {
    constexpr int K = ad_traits<M>::K;
    PSDR_CLK_MARK(0);                            // the record has arrived (and the stores of the trip before it have left)
    int pixel = -1; uint32_t slot = 0;
    Vec3<M> r = zero3<M>(), beta = zero3<M>();
    Its<float> next; next.tri = -1; next.hu = next.hv = 0.f;
    Vec3f dir(0.f);
    bool alive = false;
    Rng rng_next; rng_next.state = rng_next.inc = 0;
    if (live) {
        pixel = raw.pixel; slot = raw.slot;
        const Vec3f din{raw.dx, raw.dy, raw.dz};
        beta.x = M(raw.bx); beta.y = M(raw.by); beta.z = M(raw.bz);
        if constexpr (K > 0) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                beta.x.d[k] = in.beta[(3 + 3 * k) * in.cap + j]; beta.y.d[k] = in.beta[(4 + 3 * k) * in.cap + j];
                beta.z.d[k] = in.beta[(5 + 3 * k) * in.cap + j];
            }
        }
        Vec3<M> acc;
        acc.x = M(raw.ax); acc.y = M(raw.ay); acc.z = M(raw.az);
        if constexpr (K > 0) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                acc.x.d[k] = in.acc[(3 + 3 * k) * in.cap + j]; acc.y.d[k] = in.acc[(4 + 3 * k) * in.cap + j];
                acc.z.d[k] = in.acc[(5 + 3 * k) * in.cap + j];
            }
        }
        int tri_word = raw.tri;
        Vec3<M> f, c;
        if constexpr (TRACED) {
            // the tree hits of this vertex' two rays, traced since the record was pushed (none where its class bit is clear: the ray enters no box)
            const int cls = (tri_word >> kWfClsShift) & 3;
            st.pre[kPreBsdfRay].tri = st.pre[kPreLightRay].tri = -1;
            if (cls & 1) { const float4 h = in.hit[2 * j]; st.pre[kPreBsdfRay] = Hit{__float_as_int(h.x), h.y, h.z, h.w}; }
            if (cls & 2) { const float4 h = in.hit[2 * j + 1]; st.pre[kPreLightRay] = Hit{__float_as_int(h.x), h.y, h.z, h.w}; }
        }
        PSDR_CLK_MARK(1);                        // hit rows
        const Its<float> its = path_vertex_from_record(cx.sc, tv, tri_word & kWfTriMask, raw.hu, raw.hv, din);
        PSDR_CLK_MARK(2);                        // the vertex rebuilt from its triangle row
        if constexpr (TRACED) {
            TV<M, FL | kScenePre> tvp;
#pragma unroll
            for (int k = 0; k < (K > 0 ? K : 1); ++k) tvp.t[k] = tv.t[k];
            tvp.live = tv.live;
            int light_tri = -1;
            c = wavefront_bounce_vertex<M>(cx.sc, tvp, st, cx.jump, (uint64_t) slot, its, nrays, next, f, alive, REC ? &light_tri : nullptr, &rng_next, want_next == 0);
            if constexpr (REC) {
                // vertex `stage` of this path: (c_k, f_k) and the triangles its two rays arrived at (the adjoint kernel re-intersects those)
                float *col = wr.disk + wr.column(pixel, slot) + (long long) (kRevDiskHead + wr.stage * kRevDiskPerVertexCf) * wr.stride;
                const Vec3f cv = val(c), fv = alive ? val(f) : Vec3f(0.f);
                col[0] = cv.x; col[wr.stride] = cv.y; col[2 * wr.stride] = cv.z;
                col[3 * wr.stride] = fv.x; col[4 * wr.stride] = fv.y; col[5 * wr.stride] = fv.z;
                col[6 * wr.stride] = __int_as_float(alive ? next.tri : -1); col[7 * wr.stride] = __int_as_float(light_tri);
            }
        } else c = wavefront_bounce_vertex<M>(cx.sc, tv, st, cx.jump, (uint64_t) slot, its, nrays, next, f, alive, nullptr, &rng_next, want_next == 0);
        r = acc + beta * c;
        if (alive) {
            beta = beta * f;
            const Vec3f b = val(beta);
            alive = b.x != 0.f || b.y != 0.f || b.z != 0.f;
            Vec3f d = next.p - its.p; const float t = norm(d); dir = d / t;
        }
    }
    PSDR_CLK_MARK(3);                            // direct_step with the traced hits
    const bool goes_on = want_next && alive;
    if constexpr (REC) {
        if (live && !goes_on) {
            // the path ends at this vertex: its vertex count; a non-finite sample is zeroed as a whole and has no gradient
            const Vec3f rv = val(r);
            float *col = wr.disk + wr.column(pixel, slot);
            if (isfinite(rv.x) && isfinite(rv.y) && isfinite(rv.z)) col[wr.stride] = __int_as_float(wr.stage + 1);
            else col[0] = __int_as_float(-1);
        }
    }
    splat_runs<M>(pixel, live && !goes_on, zero_nonfinite(r), inv_spp, img, dimg, plane);
    PSDR_CLK_MARK(4);                            // splat of the paths that end here
    if (want_next) {
        if constexpr (TRACED) {
            Vec3f d_bsdf(0.f), d_light(0.f);
            const int cls = alive ? classify_next<true>(cx.sc, tv, rng_next, next, dir, &d_bsdf, &d_light, next_last, &st) : 0;
            PSDR_CLK_MARK(5);                    // the next vertex' two rays aimed
            stream_push_traced<M>(out, tq, alive, cls, pixel, slot, next, dir, beta, r, d_bsdf, d_light PSDR_CLK_PASS);
            PSDR_CLK_MARK(7);                    // record + requests stored (drained: a developer build waits for its stores here)
        } else if constexpr ((FL & kSceneForest) != 0) {
            if (out.binned) stream_push_binned<M>(out, alive, alive ? classify_next(cx.sc, tv, rng_next, next, dir, nullptr, nullptr, next_last, &st) : 0, chunk, pixel, slot, next, dir, beta, r);
            else stream_push<M>(out, alive, pixel, slot, next, dir, beta, r);
        } else stream_push<M>(out, alive, pixel, slot, next, dir, beta, r);
    }
}

// One bounce.  Plain streams: block b consumes its share of sub-stream b % kWfSub (grid-stride over the blocks of that
// sub-stream; gridDim.x is a multiple of kWfSub) and appends the surviving paths to the same sub-stream of `out`.
// Binned streams: the 64 sub-streams form one list of 256-record chunks (expensive classes first) that the workgroups
// grab kWfGrab at a time from one counter.
template <class M, int FL, bool TRACED = false, bool REC = false>
__global__ __launch_bounds__(kBlock, (is_ad<M>() ? 2 : (TRACED ? PSDR_WF_WAVES_T : PSDR_WF_WAVES))) void k_wf_bounce(LaunchCtx cx, TV<M, FL> tv, float inv_spp, float *__restrict__ img,
                                                        float *__restrict__ dimg, long long plane, PathStream in,
                                                        PathStream out, int want_next, unsigned long long *counters, RngJump jump_next, TraceQueue tq, WfRec wr) {
    TraversalStack st; setup_lds(cx, st, tv);
    uint32_t nrays = 0;
    const bool next_last = (want_next & 2) != 0;          // (bits of want_next: k_wf_camera)
    want_next &= 1;
    bool binned = false;
    if constexpr ((FL & kSceneForest) != 0 && !TRACED) binned = in.binned != 0;
    if (binned) {
        __shared__ int s_pref[kWfSub + 1];
        __shared__ int s_grab;
        if (threadIdx.x < kWfSub) {
            int c = (in.count[threadIdx.x * kWfCountStride] + kBlock - 1) / kBlock;          // chunks of this sub-stream
#pragma unroll
            for (int off = 1; off < kWfSub; off <<= 1) { const int o = __shfl_up(c, off, 64); if ((int) threadIdx.x >= off) c += o; }
            s_pref[threadIdx.x + 1] = c;
            if (threadIdx.x == 0) s_pref[0] = 0;
        }
        __syncthreads();
        const int total = s_pref[kWfSub];
        for (;;) {
            if (threadIdx.x == 0) s_grab = atomicAdd(in.count + kWfSub * kWfCountStride, kWfGrab);
            __syncthreads();
            const int c0 = s_grab;
            __syncthreads();
            if (c0 >= total) break;
            const int c1 = min(c0 + kWfGrab, total);
            for (int c = c0; c < c1; ++c) {
                int lo = 0, hi = kWfSub - 1;                                                    // sub-stream of chunk c
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_pref[mid + 1] <= c) lo = mid + 1; else hi = mid; }
                const int off = (c - s_pref[lo]) * kBlock + (int) threadIdx.x;
                const bool live = off < in.count[lo * kWfCountStride];
                wf_bounce_record<M, FL, false, false>(cx, tv, st, inv_spp, img, dimg, plane, in, out, want_next, jump_next, live, (long long) lo * in.sub_cap + off, c, nrays, tq, wr,
                                                      wf_load_raw(in, (long long) lo * in.sub_cap + off, live), next_last);
            }
        }
    } else {
        const int sub = blockIdx.x % kWfSub, per = gridDim.x / kWfSub;
        const long long in_base = (long long) sub * in.sub_cap;
        const int n = in.count[sub * kWfCountStride];
        // (measured and dropped: fetching the NEXT trip's record while the current one is processed -- 1 571 against 1 548 us per C4 stage: the stage does
        // not wait for its record)
#ifdef PSDR_STAGE_CLOCKS
        __shared__ StageClk s_ck[kBlock / 64];
        StageClk &ck = s_ck[threadIdx.x >> 6];
        const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
        if ((threadIdx.x & 63) == 0) { for (int i = 0; i < 12; ++i) ck.t[i] = 0; ck.last = t_begin; }
        StageClk *clk = REC ? nullptr : &ck;          // a recording stage shares its call with the adjoint kernel, whose clocks the call then reports
#endif
        for (int base = (blockIdx.x / kWfSub) * kBlock; base < n; base += per * kBlock)
            wf_bounce_record<M, FL, TRACED, REC>(cx, tv, st, inv_spp, img, dimg, plane, in, out, want_next, jump_next, base + (int) threadIdx.x < n, in_base + base + threadIdx.x,
                                                 base / kBlock, nrays, tq, wr, wf_load_raw(in, in_base + base + threadIdx.x, base + (int) threadIdx.x < n), next_last PSDR_CLK_PASS);
#ifdef PSDR_STAGE_CLOCKS
        if ((threadIdx.x & 63) == 0 && !REC) {
            ck.t[11] = __builtin_amdgcn_s_memtime() - t_begin;       // wave lifetime inside the loop
            for (int i = 0; i < 12; ++i) atomicAdd(counters + (blockIdx.x % kRayCounters) * kRayCounterStride + 1 + i, ck.t[i]);
        }
#endif
    }
    count_rays(counters, nrays);
}

// ---------------------------------------------------------------- traced wavefront with GEOMETRY duals (round 5)
// renderD + enoki.forward w.r.t. a geometry parameter (vertex positions, a mesh transform, the camera pose: examples/run_test.py:126-129, the only AD mode
// the reference's harness uses) ran the PathTracer as the fused kernel k_camera<Dual<K>, Dual<K>> on every scene -- on a two-level scene the kernel the
// dense trace kernel had beaten in renderC (C4 shard 43 ms against 13.4).  Here the same estimator runs as the traced wavefront: the trace kernel stays
// plain float (a hit carries no tangent), the stages re-derive every vertex DIFFERENTIABLY from its stream record as scene.cpp:346-368 does from the
// OptiX hit -- the primary vertex in solid-angle form from the film position (camera ray with its tangents, Moeller-Trumbore on the hit triangle), the
// others in path-space form from (triangle, detached barycentrics) and the position of the vertex behind them, which travels in the record with its
// tangents (its_from_hit: the arithmetic of the fused kernel's intersect(), hence its vertex and tangents).
template <int K>
__device__ __forceinline__ void stream_write_geo(const PathStream &out, long long i, int pixel, uint32_t slot, const Its<float> &next, const Vec3<Dual<K>> &prev, const Vec3<Dual<K>> &beta,
                                                 const Vec3<Dual<K>> &acc, int cls) {
    stream_write<Dual<K>>(out, i, pixel, slot, next, val(prev), beta, acc, cls);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        out.prev_t[(3 * k) * out.cap + i] = prev.x.d[k]; out.prev_t[(3 * k + 1) * out.cap + i] = prev.y.d[k]; out.prev_t[(3 * k + 2) * out.cap + i] = prev.z.d[k];
    }
}
template <int K>
__device__ __forceinline__ void stream_push_traced_geo(const PathStream &out, const TraceQueue &q, bool alive, int cls, int pixel, uint32_t slot, const Its<float> &next,
                                                       const Vec3<Dual<K>> &prev, const Vec3<Dual<K>> &beta, const Vec3<Dual<K>> &acc, const Vec3f &d_bsdf, const Vec3f &d_light) {
    const unsigned long long mask = __ballot(alive);
    if (mask == 0ull) return;
    const int lane = threadIdx.x & 63, sub = blockIdx.x % kWfSub, leader = __ffsll((long long) mask) - 1;
    const unsigned long long m1 = __ballot(alive && (cls & 1)), m2 = __ballot(alive && (cls & 2));
    const int n1 = (int) __popcll(m1), n2 = (int) __popcll(m2);
    int base = 0, rbase = 0;
    if (lane == leader) {
        base = atomicAdd(out.count + sub * kWfCountStride, (int) __popcll(mask));
        if (n1 + n2 > 0) rbase = atomicAdd(q.count + sub * kWfCountStride, n1 + n2);
    }
    base = __shfl(base, leader, 64); rbase = __shfl(rbase, leader, 64);
    if (!alive) return;
    const unsigned long long below = (1ull << lane) - 1ull;
    const long long i = (long long) sub * out.sub_cap + base + __popcll(mask & below);
    stream_write_geo<K>(out, i, pixel, slot, next, prev, beta, acc, cls);
    if (cls & 1) {
        const long long r = (long long) sub * q.sub_cap + rbase + __popcll(m1 & below);
        q.req[2 * r] = float4{next.p.x, next.p.y, next.p.z, __int_as_float((int) (2 * i))};
        q.req[2 * r + 1] = float4{d_bsdf.x, d_bsdf.y, d_bsdf.z, __int_as_float(-1)};
    }
    if (cls & 2) {
        const long long r = (long long) sub * q.sub_cap + rbase + n1 + __popcll(m2 & below);
        q.req[2 * r] = float4{next.p.x, next.p.y, next.p.z, __int_as_float((int) (2 * i + 1))};
        q.req[2 * r + 1] = float4{d_light.x, d_light.y, d_light.z, __int_as_float(-1)};
    }
}
// waves per SIMD of the dual-number stages: the lean (diffuse) instances need 172 VGPRs at two waves and take a third with a handful of spills (C4 shard
// 32.9 -> 31.3 ms), the rough-conductor instances lose 13 % there (C5 3.71 -> 4.21 ms): profiles/r05_geo_wavefront_abk.txt
#ifndef PSDR_WFG_WAVES
#define PSDR_WFG_WAVES 0
#endif
template <int FL> constexpr int wfg_waves() { return PSDR_WFG_WAVES > 0 ? PSDR_WFG_WAVES : (((FL & kSceneRough) != 0) ? 2 : 3); }
// Camera stage: the primary hit (its walk stays in the kernel: camera rays are coherent), the emitter seen directly, the requests of bounce stage 0.
template <int K, int FL>
__global__ __launch_bounds__(kBlock, (wfg_waves<FL>())) void k_wfg_camera(LaunchCtx cx, TangentView<K, FL> tv, int spp, int s_begin, SlotDiv nsp, long long j0, long long n, float inv_spp,
                                                                      float *__restrict__ img, float *__restrict__ dimg, long long plane, PathStream out, unsigned long long *counters, TraceQueue tq) {
    using M = Dual<K>;
    TraversalStack st; setup_lds(cx, st, tv);
    uint32_t nrays = 0;
    const long long nceil = (n + kBlock - 1) / kBlock * kBlock;
    for (long long jj = (long long) blockIdx.x * kBlock + threadIdx.x; jj < nceil; jj += (long long) gridDim.x * kBlock) {
        const bool in = jj < n;
        int pixel = 0, s_in = 0;
        if (in) slot_to_pixel(j0 + jj, nsp, pixel, s_in);
        Vec3<M> r = zero3<M>();
        Its<float> next; next.tri = -1; next.hu = next.hv = 0.f;
        Vec3f dir(0.f);
        bool alive = false;
        uint32_t slot = 0;
        float sxy[2] = {0.f, 0.f};
        Rng rng_next; rng_next.state = rng_next.inc = 0;
        if (in) {
            slot = (uint32_t) ((uint64_t) pixel * (uint64_t) spp + (uint64_t) (s_begin + s_in));
            r = wavefront_primary_vertex<M>(cx.sc, tv, st, cx.lp, cx.jump, pixel, slot, nrays, next, dir, alive, &rng_next, sxy);
        }
        splat_runs<M>(pixel, in && !alive, zero_nonfinite(r), inv_spp, img, dimg, plane);
        Vec3f d_bsdf(0.f), d_light(0.f);
        const int cls = alive ? classify_next<true>(cx.sc, tv, rng_next, next, dir, &d_bsdf, &d_light) : 0;
        const Vec3<M> film{M(sxy[0]), M(sxy[1]), M(0.f)}, one{M(1.f), M(1.f), M(1.f)};
        stream_push_traced_geo<K>(out, tq, alive, cls, pixel, slot, next, film, one, r, d_bsdf, d_light);
    }
    count_rays(counters, nrays);
}
// Bounce stage k: FIRST = the primary vertex (solid-angle form from the film position), else a path-space vertex.
template <int K, int FL, bool FIRST>
__global__ __launch_bounds__(kBlock, (wfg_waves<FL>())) void k_wfg_bounce(LaunchCtx cx, TangentView<K, FL> tv, float inv_spp, float *__restrict__ img, float *__restrict__ dimg, long long plane,
                                                                      PathStream in, PathStream out, int want_next, unsigned long long *counters, TraceQueue tq) {
    using M = Dual<K>;
    using G = Dual<K>;
    TraversalStack st; setup_lds(cx, st, tv);
    uint32_t nrays = 0;
    TV<M, FL | kScenePre> tvp;
#pragma unroll
    for (int k = 0; k < K; ++k) tvp.t[k] = tv.t[k];
    tvp.live = tv.live;
    const int sub = blockIdx.x % kWfSub, per = gridDim.x / kWfSub;
    const long long in_base = (long long) sub * in.sub_cap;
    const int n = in.count[sub * kWfCountStride];
    for (int base = (blockIdx.x / kWfSub) * kBlock; base < n; base += per * kBlock) {
        const bool live = base + (int) threadIdx.x < n;
        const long long j = in_base + base + threadIdx.x;
        const WfRaw raw = wf_load_raw(in, j, live);
        int pixel = -1; uint32_t slot = 0;
        Vec3<M> r = zero3<M>(), beta = zero3<M>(), prevp = zero3<M>();
        Its<float> nextf; nextf.tri = -1; nextf.hu = nextf.hv = 0.f;
        bool alive = false;
        Rng rng_next; rng_next.state = rng_next.inc = 0;
        if (live) {
            pixel = raw.pixel; slot = raw.slot;
            beta.x = M(raw.bx); beta.y = M(raw.by); beta.z = M(raw.bz);
            Vec3<M> acc; acc.x = M(raw.ax); acc.y = M(raw.ay); acc.z = M(raw.az);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                beta.x.d[k] = in.beta[(3 + 3 * k) * in.cap + j]; beta.y.d[k] = in.beta[(4 + 3 * k) * in.cap + j]; beta.z.d[k] = in.beta[(5 + 3 * k) * in.cap + j];
                acc.x.d[k] = in.acc[(3 + 3 * k) * in.cap + j]; acc.y.d[k] = in.acc[(4 + 3 * k) * in.cap + j]; acc.z.d[k] = in.acc[(5 + 3 * k) * in.cap + j];
            }
            const int tri_word = raw.tri, cls = (tri_word >> kWfClsShift) & 3;
            st.pre[kPreBsdfRay].tri = st.pre[kPreLightRay].tri = -1;
            if (cls & 1) { const float4 h = in.hit[2 * j]; st.pre[kPreBsdfRay] = Hit{__float_as_int(h.x), h.y, h.z, h.w}; }
            if (cls & 2) { const float4 h = in.hit[2 * j + 1]; st.pre[kPreLightRay] = Hit{__float_as_int(h.x), h.y, h.z, h.w}; }
            const Hit hk{tri_word & kWfTriMask, raw.hu, raw.hv, 0.f};
            Its<G> its;
            if constexpr (FIRST) {
                const RayT<G> ray = primary_ray<G>(cx.sc, tvp, raw.dx, raw.dy);                    // (sx, sy): the camera ray with the pose's tangents
                its = its_from_hit<G>(cx.sc, tvp, hk, ray, kSolidAngle);
            } else {
                RayT<G> ray;
                ray.o.x = G(raw.dx); ray.o.y = G(raw.dy); ray.o.z = G(raw.dz);
#pragma unroll
                for (int k = 0; k < K; ++k) { ray.o.x.d[k] = in.prev_t[(3 * k) * in.cap + j]; ray.o.y.d[k] = in.prev_t[(3 * k + 1) * in.cap + j]; ray.o.z.d[k] = in.prev_t[(3 * k + 2) * in.cap + j]; }
                ray.d = zero3<G>();                                                                 // unused by the path-space form
                its = its_from_hit<G>(cx.sc, tvp, hk, ray, kPathSpace);
            }
            Rng rng; rng.init((uint64_t) slot, cx.jump);
            Vec3<M> f = zero3<M>(); bool nvalid = false;
            const Vec3<M> c = direct_step<G, M>(cx.sc, tvp, st, rng, its, true, 1, 1, nrays, &nextf, &f, &nvalid, nullptr, want_next == 0);          // the next vertex as plain values (its_cast)
            rng_next = rng;
            r = acc + beta * c;
            alive = nvalid;
            if (alive) {
                beta = beta * f;
                const Vec3f b = val(beta);
                alive = b.x != 0.f || b.y != 0.f || b.z != 0.f;
                prevp = its.p;
            }
        }
        const bool goes_on = want_next && alive;
        splat_runs<M>(pixel, live && !goes_on, zero_nonfinite(r), inv_spp, img, dimg, plane);
        if (want_next) {
            Vec3f d_bsdf(0.f), d_light(0.f);
            const int cls = alive ? classify_next<true>(cx.sc, tv, rng_next, nextf, Vec3f(0.f), &d_bsdf, &d_light) : 0;
            stream_push_traced_geo<K>(out, tq, alive, cls, pixel, slot, nextf, prevp, beta, r, d_bsdf, d_light);
        }
    }
    count_rays(counters, nrays);
}

// ------------------------------------------------------------------------ k_primary_edge
#ifndef PSDR_WAVES_PE
#define PSDR_WAVES_PE 4
#endif
// forward instances of the lean two-level variant (K = 1: 119 -> 96 VGPRs, 14 spilled) take a fifth wave: C3 three-term forward 2.85 -> 2.78 ms
// (k_primary_edge 1 310 -> 1 248 us); the reverse kernel loses 3-11 % there, three waves lose everywhere (profiles/r04_occupancy_abk.txt)
#ifndef PSDR_WAVES_PE_FWD
#define PSDR_WAVES_PE_FWD (PSDR_WAVES_PE + 1)
#endif
template <int K, int FL> constexpr int primary_edge_waves() { return ((FL & (kSceneEnv | kSceneRough)) == 0 && (FL & kSceneForest) != 0 && K == 1) ? PSDR_WAVES_PE_FWD : PSDR_WAVES_PE; }
template <int K, int FL, int INTEG>
__global__ __launch_bounds__(kBlock, (primary_edge_waves<K, FL>())) void k_primary_edge(LaunchCtx cx, TangentView<K, FL> tv, long long i0, long long n, float inv_sppe,
                                                         float *__restrict__ dimg, long long plane, unsigned long long *counters,
                                                         const uint32_t *__restrict__ order) {
    TraversalStack st; setup_lds(cx, st);
    uint32_t nrays = 0;
    const long long nceil = (n + kBlock - 1) / kBlock * kBlock;
    for (long long j = (long long) blockIdx.x * kBlock + threadIdx.x; j < nceil; j += (long long) gridDim.x * kBlock) {
        float tan[K][3];
        int pixel = -1;
        if (j < n) {
            const long long jj = order ? (long long) order[j] : j;          // pixel-sorted evaluation order (psdr_hip.hip)
            pixel = primary_edge_sample<K, INTEG>(cx.sc, tv, st, cx.lp, cx.jump, (uint64_t) (i0 + jj), inv_sppe, tan, nrays);
        }
        // pixel-sorted slots: the lanes of a wave mostly sit on ONE pixel of a silhouette -- 64 float atomics on one address serialise in L2 (bunny_light: the
        // forward kernel 16.1 ms where the reverse kernel, which sums its runs of equal edges, takes 11.1).  One atomic per run of equal pixels.
        float v[3 * K];
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int c = 0; c < 3; ++c) v[3 * k + c] = pixel >= 0 ? tan[k][c] : 0.f;
        const bool head = wave_run_sum<3 * K>(pixel, v);
        if (head && pixel >= 0) {
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    if (v[3 * k + c] != 0.f) atomicAdd(dimg + (size_t) k * plane + (size_t) pixel * 3 + c, v[3 * k + c]);
        }
    }
    count_rays(counters, nrays);
}

// ---------------------------------------------------------------------- k_secondary_edge
// Split launch of the secondary-edge term (forward and reverse): only a few per cent of the slots get past the first two rays, yet the
// dual-number / adjoint code that the survivors need holds the full kernels at 2 waves/SIMD while all of them walk the tree.  The filter
// traces those two rays for every slot at the occupancy of a plain kernel and compacts the survivors' slot numbers (one atomic per
// wave); the full kernel then runs over that list.
// ---- probe / final launches (two-level scenes).  A fused kernel pays its slowest lane's tree walk at every closest_hit although few of its rays
// enter a tree box; where the rays of a slot are known before any of them is traced, the launch becomes PROBE pass (the rays that enter a box
// become requests of the dense trace kernel, psdr_hip.hip k_wf_trace; one mask word per slot says which) -> trace kernel -> FINAL pass = the same
// kernel compiled with kScenePre: closest_hit tests the kernel-argument primitives and merges the hit row of the ray (TraversalStack::pre).
// The requests of one lane's rays: `want` bit r set = ray r (direction d[r], common origin o) becomes request dest0 + r; block b appends to
// sub-queue b % kWfSub with one atomic per wave.  `edge` >= 0: a secondary edge whose adjacent faces both rays skip.
template <int NR>
__device__ __forceinline__ void probe_push(const TraceQueue &q, uint32_t want, const Vec3f &o, const Vec3f (&d)[NR], uint32_t dest0, int edge) {
    const int lane = threadIdx.x & 63, sub = blockIdx.x % kWfSub;
    unsigned long long m[NR]; int total = 0;
#pragma unroll
    for (int r = 0; r < NR; ++r) { m[r] = __ballot((want >> r) & 1u); total += (int) __popcll(m[r]); }
    if (total == 0) return;
    int base = 0;
    if (lane == 0) base = atomicAdd(q.count + sub * kWfCountStride, total);
    base = __shfl(base, 0, 64);
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        if ((want >> r) & 1u) {
            const long long i = (long long) sub * q.sub_cap + base + __popcll(m[r] & below);
            q.req[2 * i] = float4{o.x, o.y, o.z, __int_as_float((int) (dest0 + (uint32_t) r))};
            q.req[2 * i + 1] = float4{d[r].x, d[r].y, d[r].z, __int_as_float(edge)};
        }
        base += (int) __popcll(m[r]);
    }
}
__device__ __forceinline__ bool enters_any_box(const SceneView &sc, const Vec3f &o, const Vec3f &d) {
    const Vec3f inv{1.f / d.x, 1.f / d.y, 1.f / d.z};
    bool any = false;
    for (int k = 0; k < sc.n_blas; ++k) { float te; any = any || blas_box(sc, k, o, inv, INFINITY, te); }
    return any;
}

// Guiding-grid build through the filter launches (guide_launch): slot j = (round r = j / n, sample stream l = j % n) of the n = cells x per streams
// of DirectIntegrator::preprocess_secondary_edges (direct.cpp:166-204); stream l draws three numbers per round, stratified into its cell.
struct GuideGrid { int r0, r1, r2, per, n; float scale; };          // n == 0: an ordinary secondary-edge launch
__device__ __forceinline__ void guide_slot_sample(const GuideGrid &gg, long long j, float s3[3], int &cell) {
    const int r = (int) (j / gg.n), l = (int) (j - (long long) r * gg.n);
    cell = l / gg.per;
    const int c0 = cell / (gg.r1 * gg.r2), rem = cell - c0 * gg.r1 * gg.r2, c1 = rem / gg.r2, c2 = rem - c1 * gg.r2;
    Rng rng; rng.init((uint64_t) l, rng_jump_hd(3ull * (uint64_t) r));
    s3[0] = rng.next(); s3[1] = rng.next(); s3[2] = rng.next();
    s3[0] = (s3[0] + (float) c0) * (1.f / (float) gg.r0);
    s3[1] = (s3[1] + (float) c1) * (1.f / (float) gg.r1);
    s3[2] = (s3[2] + (float) c2) * (1.f / (float) gg.r2);
}
// Probe pass of the secondary-edge filter: mask bit 2 = the slot passes the geometric part of the test (everything else never traces), bits 0 / 1 =
// its ray towards the emitter sample / away from it enters a tree box.
template <int FL>
__global__ __launch_bounds__(kBlock, 6) void k_se_probe(LaunchCtx cx, long long i0, long long n, TraceQueue tq, uint32_t *__restrict__ mask, GuideGrid gg) {
    // slots [i0, i0 + n) of the launch; hit rows, masks and request destinations are relative to i0 (one chunk of the launch)
    TraversalStack st; setup_lds(cx, st);
    const bool guided = cx.sc.d.guide_cmf != nullptr && cx.sc.d.num_guide_cells > 0;
    const long long nceil = (n + kBlock - 1) / kBlock * kBlock;
    for (long long j = (long long) blockIdx.x * kBlock + threadIdx.x; j < nceil; j += (long long) gridDim.x * kBlock) {
        uint32_t m = 0; Vec3f p0(0.f), d[2] = {Vec3f(0.f), Vec3f(0.f)}; int edge = -1;
        if (j < n) {
            float s3[3];
            if (gg.n > 0) { int cell; guide_slot_sample(gg, i0 + j, s3, cell); }
            else {
                Rng rng; rng.init((uint64_t) (i0 + j), cx.jump);
                s3[0] = rng.next(); s3[1] = rng.next(); s3[2] = rng.next();
            }
            if (guided) (void) guide_sample_reuse(cx.sc, s3);
            Vec3f dir;
            if (secondary_edge_rays<FL>(cx.sc, s3, p0, dir, edge)) {
                d[0] = dir; d[1] = -dir;
                m = 4u | (enters_any_box(cx.sc, p0, d[0]) ? 1u : 0u) | (enters_any_box(cx.sc, p0, d[1]) ? 2u : 0u);
            }
            mask[j] = m;
        }
        probe_push<2>(tq, m & 3u, p0, d, (uint32_t) (2 * j), edge);
    }
}

// Probe pass of a camera launch with DirectIntegrator(1, 1) on a two-level scene: the primary hit (its walk stays in this kernel: camera rays are
// coherent) goes to hit row 3 j + 2, the vertex' two rays (classify_next: the draws and the sampling routines of direct_step) that enter a tree
// box become trace requests for rows 3 j + 0 / 1; mask bits 0 / 1 / 2 say which rows the final pass reads.
// (Measured and dropped: the same for the primary-edge kernels -- their Li evaluations start ON the silhouette of the mesh, nearly every ray walks the
// tree anyway and the fused kernel loses little to divergence: C3 1.29 ms fused against 0.35 + 0.53 + 0.66 probe / trace / final, C4 shard 14.8 against 22.)
template <int FL> __global__ __launch_bounds__(kBlock, PSDR_WAVES_C) void k_direct_probe(LaunchCtx cx, TangentView<0, FL> tv0, int spp, int s_begin, SlotDiv nsp, long long j0, long long n,
                                                                                       TraceQueue tq, float4 *__restrict__ hit, uint32_t *__restrict__ mask, RngJump jump_next) {
    TraversalStack st; setup_lds(cx, st, tv0);
    uint32_t nrays = 0;
    const long long nceil = (n + kBlock - 1) / kBlock * kBlock;
    for (long long jj = (long long) blockIdx.x * kBlock + threadIdx.x; jj < nceil; jj += (long long) gridDim.x * kBlock) {
        uint32_t m = 0; Vec3f p0(0.f), d[2] = {Vec3f(0.f), Vec3f(0.f)};
        if (jj < n) {
            int pixel, s_in;
            slot_to_pixel(j0 + jj, nsp, pixel, s_in);
            const uint32_t slot = (uint32_t) ((uint64_t) pixel * (uint64_t) spp + (uint64_t) (s_begin + s_in));
            Its<float> next; next.tri = -1; next.hu = next.hv = 0.f;
            Vec3f dir(0.f); bool alive = false;
            Rng rng_next; rng_next.state = rng_next.inc = 0;
            (void) wavefront_primary_vertex<float>(cx.sc, tv0, st, cx.lp, cx.jump, pixel, slot, nrays, next, dir, alive, &rng_next);
            if (alive) {
                hit[3 * jj + 2] = float4{__int_as_float(next.tri), next.hu, next.hv, next.t};
                m = 4u | (uint32_t) classify_next<true>(cx.sc, tv0, rng_next, next, dir, &d[0], &d[1], true, &st);          // DirectIntegrator: the BSDF sample's hit matters only on an emitter
                p0 = next.p;
            }
            mask[jj] = m;
        }
        probe_push<2>(tq, m & 3u, p0, d, (uint32_t) (3 * jj), -1);
    }
}

template <int FL>
__global__ __launch_bounds__(kBlock, 4) void k_secondary_edge_filter(LaunchCtx cx, long long i0, long long n, uint32_t *__restrict__ list, int *__restrict__ list_n,
                                                                      unsigned long long *counters, ProbeView pv, GuideGrid gg, long long list_base) {
    // slots [i0, i0 + n); the survivor list takes list_base + j (the slot's index in the whole launch: several chunks append to one list)
    TraversalStack st; setup_lds(cx, st);
    uint32_t nrays = 0;
    const bool guided = cx.sc.d.guide_cmf != nullptr && cx.sc.d.num_guide_cells > 0;
    const long long nceil = (n + kBlock - 1) / kBlock * kBlock;
    constexpr int kSeBuf = 256;
    __shared__ uint32_t s_buf[kBlock / 64][kSeBuf];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int held = 0;                                    // wave-uniform: survivors waiting in this wave's buffer
    auto flush = [&]() {
        if (held == 0) return;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        int base = 0;
        if (lane == 0) base = atomicAdd(list_n, held);
        base = __shfl(base, 0, 64);
        for (int i = lane; i < held; i += 64) list[base + i] = s_buf[wave][i];
        __builtin_amdgcn_wave_barrier();
        held = 0;
    };
    for (long long j = (long long) blockIdx.x * kBlock + threadIdx.x; j < nceil; j += (long long) gridDim.x * kBlock) {
        bool keep = false;
        bool probed = true;
        if constexpr ((FL & kScenePre) != 0) {
            // final pass of a traced launch: a slot that failed the geometric test in the probe pass evaluates nothing; the others find their tree hits
            const uint32_t m = j < n ? pv.mask[j] : 0u;
            probed = (m & 4u) != 0u;
            if (probed) probe_load<2>(st, pv, j, m);
        }
        if (j < n && probed) {
            float s3[3];
            if (gg.n > 0) { int cell; guide_slot_sample(gg, i0 + j, s3, cell); }
            else {
                Rng rng; rng.init((uint64_t) (i0 + j), cx.jump);
                s3[0] = rng.next(); s3[1] = rng.next(); s3[2] = rng.next();
            }
            if (guided) (void) guide_sample_reuse(cx.sc, s3);
            keep = secondary_edge_survives<FL>(cx.sc, st, s3, nrays);
        }
        // survivors are gathered per WAVE in LDS and appended with one atomic per kSeBuf / 2 of them: one atomic per wave and trip on the single
        // counter was the kernel -- a million same-address L2 atomics at 3-5 ns each on the C4 shard (4 of its 12 ms), 0.26 of 0.89 ms on C3
        const unsigned long long mask = __ballot(keep);
        if (mask != 0ull) {
            if (keep) s_buf[wave][held + (int) __popcll(mask & ((1ull << lane) - 1ull))] = (uint32_t) (list_base + j);
            held += (int) __popcll(mask);
        }
        if (held > kSeBuf - 64) flush();
    }
    flush();
    count_rays(counters, nrays);
}

template <int K, int FL>
__global__ __launch_bounds__(kBlock) void k_secondary_edge(LaunchCtx cx, TangentView<K, FL> tv, long long i0, long long n, float inv_sppse,
                                                           float *__restrict__ dimg, long long plane, unsigned long long *counters,
                                                           const uint32_t *__restrict__ list, const int *__restrict__ list_n) {
    using R = Dual<K>;
    TraversalStack st; setup_lds(cx, st);
    uint32_t nrays = 0;
    const bool guided = cx.sc.d.guide_cmf != nullptr && cx.sc.d.num_guide_cells > 0;
    if (list != nullptr) n = *list_n;
    for (long long jj = (long long) blockIdx.x * kBlock + threadIdx.x; jj < n; jj += (long long) gridDim.x * kBlock) {
        const long long j = list != nullptr ? (long long) list[jj] : jj;
        Rng rng; rng.init((uint64_t) (i0 + j), cx.jump);
        float s3[3] = {rng.next(), rng.next(), rng.next()};
        const float pdf0 = guided ? guide_sample_reuse(cx.sc, s3) : 1.f;
        Vec3<R> value;
        const int pixel = secondary_edge_sample<R>(cx.sc, tv, st, s3, value, nrays, list == nullptr);
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
template <int FL>
__global__ __launch_bounds__(kBlock) void k_guide(LaunchCtx cx, int r0, int r1, int r2, int per, int nrounds, long long n,
                                                  float *__restrict__ mass, unsigned long long *counters) {
    TraversalStack st; setup_lds(cx, st);
    uint32_t nrays = 0;
    const TangentView<0, FL> tv0{};
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

// The guiding-grid build as filter + survivors (two-level scenes): the slots of ALL rounds pass the probe / dense trace / filter launches of the
// secondary-edge term (a few per cent survive the first two rays of eval_secondary_edge); this kernel evaluates the survivors and adds their mass.
template <int FL>
__global__ __launch_bounds__(kBlock) void k_guide_survivors(LaunchCtx cx, GuideGrid gg, const uint32_t *__restrict__ list, const int *__restrict__ list_n, float *__restrict__ mass,
                                                            unsigned long long *counters) {
    TraversalStack st; setup_lds(cx, st);
    uint32_t nrays = 0;
    const TangentView<0, FL> tv0{};
    const int n = *list_n;
    for (int jj = blockIdx.x * kBlock + threadIdx.x; jj < n; jj += gridDim.x * kBlock) {
        float s3[3]; int cell;
        guide_slot_sample(gg, (long long) list[jj], s3, cell);
        Vec3f v;
        secondary_edge_sample<float>(cx.sc, tv0, st, s3, v, nrays, false);          // the filter traced (and counted) the first two rays
        v = zero_nonfinite(v);
        const float a = fmaxf(v.x, fmaxf(v.y, v.z)) * gg.scale;                     // hmax(value0 / per) / nrounds
        if (a != 0.f) atomicAdd(mass + cell, a);
    }
    count_rays(counters, nrays);
}

#ifndef PSDR_TINY_DIRECT_ROWS
#define PSDR_TINY_DIRECT_ROWS 1
#endif
#ifndef PSDR_SINK_CLASS_TEST
#define PSDR_SINK_CLASS_TEST 1
#endif
template <int FL> struct DeviceSink {
    static constexpr int flags = FL;
    static constexpr bool has_env = (FL & kSceneEnv) != 0;
    psdr_grads g;
    SinkLayout L;
    // the cache is addressed through LDS-address-space pointers: with generic pointers the compiler merges the two arms of
    // "cached ? LDS add : global add" into ONE flat_atomic_add_f32 on a selected address (137 of them in the PathTracer
    // geometry kernel) -- a flat atomic goes through the texture-address path, counts against vmcnt AND lgkmcnt, and every
    // later wait for a load or a path-record read then waits for it as well
    typedef __attribute__((address_space(3))) float lds_float;
    lds_float *lds;       // THIS lane's copy of the cache (lane & (rep - 1))
    lds_float *lds0;      // copy 0
    __device__ __forceinline__ static void lds_add(lds_float *p, float v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    float cam[16];
    // finite and not zero: ONE v_cmp_class_f32 (normal numbers of either sign; denormals are flushed in these kernels) instead of two compares and an s_and
    __device__ __forceinline__ static bool ok(float v) {
#if PSDR_SINK_CLASS_TEST
        return __builtin_amdgcn_classf(v, 0x108);
#else
        return v != 0.f && isfinite(v);
#endif
    }
    __device__ __forceinline__ void glob(float *base, size_t i, float v) const { if (base != nullptr && ok(v)) atomicAdd(base + i, v); }
    // The ~20 words of one vertex' row adjoint arrive one add_tri at a time: the tri -> cache slot lookup (a
    // global load that the compiler cannot hoist across the atomics) is remembered for the last triangle
    // (C2 PathTracer(3): 370 -> 90 global loads per slot).
    int last_tri, last_slot;
    __device__ __forceinline__ void add_tri(int tri, int word, float v) {
        if (g.g_tri_info == nullptr || !ok(v)) return;
        // scenes without a tree: every row is cached at slot == triangle (render_rev checks it) -- no map lookup, no range test, no global-atomic arm
        if constexpr ((FL & kSceneTiny) != 0 && PSDR_TINY_DIRECT_ROWS) { lds_add(lds + L.hot_off + tri * PSDR_TRI_STRIDE + word, v); return; }
        if (tri != last_tri) { last_tri = tri; last_slot = L.hot_rows ? L.hot_map[tri] : -1; }
        const int slot = last_slot;
        if (slot >= 0 && slot < L.hot_rows) lds_add(lds + L.hot_off + slot * PSDR_TRI_STRIDE + word, v);
        else atomicAdd(g.g_tri_info + (size_t) tri * PSDR_TRI_STRIDE + word, v);
    }
    // One whole row adjoint (position at (u, v), face normal, area).  The rows every light sample lands on have lane-private
    // accumulators: a plain LDS read-add-write per word (conflict-free: word * kBlock + lane) instead of an LDS float atomic
    // at ~3 cycles per active lane.  false: not such a row, the caller scatters word by word.
    lds_float *priv;      // this lane's column of the private block
    __device__ __forceinline__ static float finite(float v) { return isfinite(v) ? v : 0.f; }
    __device__ __forceinline__ bool add_row(int tri, float u, float v, const Vec3f &ap, const Vec3f &afn, float aarea) {
        if (L.priv_rows == 0 || g.g_tri_info == nullptr) return false;
        const bool r0 = tri == L.priv_tri[0], r1 = tri == L.priv_tri[1];
        if (!(r0 || r1)) return false;
        lds_float *q = priv + (r1 ? kPrivRowWords * kBlock : 0);
        const Vec3f a{finite(ap.x), finite(ap.y), finite(ap.z)};
        const float w[kPrivRowWords] = {a.x, a.y, a.z, u * a.x, u * a.y, u * a.z, v * a.x, v * a.y, v * a.z, finite(afn.x), finite(afn.y), finite(afn.z), finite(aarea)};
#pragma unroll
        for (int i = 0; i < kPrivRowWords; ++i) q[i * kBlock] += w[i];
        return true;
    }
    __device__ __forceinline__ void add_texel(int idx, float v) const {
        if (g.g_texels == nullptr || !ok(v)) return;
        if (L.tex_n) lds_add(lds + L.tex_off + idx, v); else atomicAdd(g.g_texels + idx, v);
    }
    __device__ __forceinline__ void add_rad(int e, int c, float v) const {
        if (g.g_emitter_rad == nullptr || !ok(v)) return;
        if (e == L.priv_emitter) { priv[(2 * kPrivRowWords + c) * kBlock] += v; return; }
        if (L.rad_n) lds_add(lds + L.rad_off + e * 3 + c, v); else atomicAdd(g.g_emitter_rad + e * 3 + c, v);
    }
    __device__ __forceinline__ void add_cam(int word, float v) { if (ok(v)) cam[word] += v; }
    // Complete row adjoints wait in a per-lane LDS column (plain stores, one column per lane: conflict-free) until the kernel's convergent point, where
    // flush_pending_wave adds them sorted by row (sink_add_row_wave).  L.pend_rows == 0 (the launch reserved no block) or a full column: scatter now.
    int n_pending;
    __device__ __forceinline__ void defer_row(int tri, float u, float v, const RowAdj &r) {
        if (g.g_tri_info == nullptr) return;
        if (L.pend_rows == 0 || n_pending >= L.pend_rows) {
            // (as scatter_row, spelled out: psdr_reverse.h's template is declared behind this header's includes)
            const float w[kPrivRowWords] = {r.p.x, r.p.y, r.p.z, u * r.p.x, u * r.p.y, u * r.p.z, v * r.p.x, v * r.p.y, v * r.p.z, r.fn.x, r.fn.y, r.fn.z, r.area};
#pragma unroll
            for (int i = 0; i < kPrivRowWords; ++i) add_tri(tri, i < 9 ? i : i + 9, w[i]);
            return;
        }
        lds_float *q = lds0 + L.pend_off + threadIdx.x + n_pending * (kPendWords * kBlock);
        q[0] = __int_as_float(tri); q[kBlock] = u; q[2 * kBlock] = v;
        q[3 * kBlock] = r.p.x; q[4 * kBlock] = r.p.y; q[5 * kBlock] = r.p.z; q[6 * kBlock] = r.fn.x; q[7 * kBlock] = r.fn.y; q[8 * kBlock] = r.fn.z; q[9 * kBlock] = r.area;
        ++n_pending;
    }
    __device__ __forceinline__ void add_env(int word, float v) const { if (L.env_n && ok(v)) lds_add(lds + L.env_off + word, v); }
    __device__ __forceinline__ void add_sedge(int e, int word, float v) const { glob(g.g_sec_edge, (size_t) e * PSDR_SEDGE_STRIDE + word, v); }
    __device__ __forceinline__ void add_pedge(int e, int word, float v) const { glob(g.g_prim_edge, (size_t) e * PSDR_PEDGE_STRIDE + word, v); }

    __device__ __forceinline__ void begin(float *cache) {
        lds0 = (lds_float *) cache;
        last_tri = -1; last_slot = -1; n_pending = 0;
        lds = lds0 + (threadIdx.x & (L.rep - 1)) * L.stride;
        for (int i = threadIdx.x; i < L.rep * L.stride; i += kBlock) lds0[i] = 0.f;
        priv = lds0 + L.priv_off + threadIdx.x;
        if (L.priv_rows > 0 && !L.priv_regs) {
#pragma unroll 1
            for (int i = 0; i < kPrivWords; ++i) priv[i * kBlock] = 0.f;
        }
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
                if ((threadIdx.x & 63) == 0 && v != 0.f) lds_add(lds0 + L.cam_off + i, v);
            }
        }
        if (L.priv_rows > 0 && !L.priv_regs) {
            // private columns -> the cache words they stand for: thread t sums word (t & 31) over 32 of the 256 lanes
            __syncthreads();
            const int w = threadIdx.x & 31, part = threadIdx.x >> 5;
            if (w < kPrivWords) {
                float sum = 0.f;
                for (int l = 0; l < 32; ++l) sum += lds0[L.priv_off + w * kBlock + part * 32 + ((l + threadIdx.x) & 31)];
                if (sum != 0.f) {
                    if (w < 2 * kPrivRowWords) {
                        const int r = w / kPrivRowWords, c = w % kPrivRowWords;
                        const int word = c < 9 ? c : c + 9;                                    // 9..11 -> face normal (18..20), 12 -> area (21)
                        if (r < L.priv_rows) lds_add(lds0 + L.hot_off + L.priv_slot[r] * PSDR_TRI_STRIDE + word, sum);
                    } else lds_add(lds0 + L.rad_off + L.priv_emitter * 3 + (w - 2 * kPrivRowWords), sum);
                }
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < L.total; i += kBlock) {
            float v = lds0[i];
            for (int r = 1; r < L.rep; ++r) v += lds0[r * L.stride + i];
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

// The same sink with the lane-private accumulators in REGISTERS (SinkLayout::priv_regs; the PathTracer's geometry-adjoint kernel: 2 waves per SIMD,
// ~40 of its 256 VGPRs free).  The LDS version costs a read and a write per word and light sample -- 78 LDS round trips per slot whose latency a
// kernel at 2 waves per SIMD cannot hide -- and 30 KB of LDS that tree scenes could not spare at all (they fell back to LDS atomics).  Here a row
// adjoint is 26 selects + adds, the totals leave through six DPP adds per word at kernel end.
#ifndef PSDR_SINK_REG_PRIV
#define PSDR_SINK_REG_PRIV 1
#endif
template <int FL> struct RegPrivSink : DeviceSink<FL> {
    typedef DeviceSink<FL> Base;
    float pacc[kPrivWords];
    __device__ __forceinline__ explicit RegPrivSink(const Base &b) : Base(b) {}
    __device__ __forceinline__ bool add_row(int tri, float u, float v, const Vec3f &ap, const Vec3f &afn, float aarea) {
        if (this->L.priv_rows == 0 || this->g.g_tri_info == nullptr) return false;
        const bool r0 = tri == this->L.priv_tri[0], r1 = tri == this->L.priv_tri[1];
        if (!(r0 || r1)) return false;
        const Vec3f a{Base::finite(ap.x), Base::finite(ap.y), Base::finite(ap.z)};
        const float w[kPrivRowWords] = {a.x, a.y, a.z, u * a.x, u * a.y, u * a.z, v * a.x, v * a.y, v * a.z, Base::finite(afn.x), Base::finite(afn.y), Base::finite(afn.z),
                                        Base::finite(aarea)};
#pragma unroll
        for (int i = 0; i < kPrivRowWords; ++i) { pacc[i] += r0 ? w[i] : 0.f; pacc[kPrivRowWords + i] += r1 ? w[i] : 0.f; }
        return true;
    }
    __device__ __forceinline__ void add_rad(int e, int c, float v) {
        if (this->g.g_emitter_rad == nullptr || !Base::ok(v)) return;
        if (e == this->L.priv_emitter) {
#pragma unroll
            for (int k = 0; k < 3; ++k) pacc[2 * kPrivRowWords + k] += c == k ? v : 0.f;
            return;
        }
        Base::add_rad(e, c, v);
    }
    __device__ __forceinline__ void begin(float *cache) {
        Base::begin(cache);
#pragma unroll
        for (int i = 0; i < kPrivWords; ++i) pacc[i] = 0.f;
    }
    __device__ __forceinline__ void end() {
        if (this->L.priv_rows > 0) {
#pragma unroll
            for (int w = 0; w < kPrivWords; ++w) {
                const float sum = wave_total(pacc[w]);             // every lane of the workgroup is here
                if ((threadIdx.x & 63) == 0 && sum != 0.f) {
                    if (w < 2 * kPrivRowWords) {
                        const int r = w / kPrivRowWords, c = w % kPrivRowWords;
                        const int word = c < 9 ? c : c + 9;                                    // 9..11 -> face normal (18..20), 12 -> area (21)
                        if (r < this->L.priv_rows) Base::lds_add(this->lds0 + this->L.hot_off + this->L.priv_slot[r] * PSDR_TRI_STRIDE + word, sum);
                    } else if (this->L.priv_emitter >= 0) Base::lds_add(this->lds0 + this->L.rad_off + this->L.priv_emitter * 3 + (w - 2 * kPrivRowWords), sum);
                }
            }
        }
        Base::end();
    }
};
// (Measured and dropped, round 5: the texel triples of the material-only reverse kernels summed per vertex in registers, parked in a per-lane LDS column and
// added sorted by texel at the slot's end -- TexDeferSink.  The texel kernel sits on a spill cliff at five waves: the five registers of the open entry alone took
// C2's texel gradient from 2.05 to 2.45 ms with the deferral switched OFF at run time, 2.96 ms with it on: profiles/r05_rev_sorted_abk.txt.)
// which kernels keep them in registers: the host (render_rev) sets SinkLayout::priv_regs by the same rule
// (not the rough-conductor instances: their adjoint kernel already fills 256 VGPRs and spilled 180 more with the accumulators: C5 7.75 -> 8.27 ms.
// The wrong camera gradient round 3 saw in that configuration was the compiler's spill-placement defect, DESIGN.md "the order-dependent gradient")
#ifndef PSDR_SINK_REG_PRIV_ROUGH
#define PSDR_SINK_REG_PRIV_ROUGH 0          // test builds (variants/): the register accumulators in the rough-conductor instances too
#endif
template <int FL, bool GEO, int INTEG> constexpr bool reg_priv_kernel() { return PSDR_SINK_REG_PRIV && GEO && INTEG == PSDR_INTEGRATOR_PATH && ((FL & kSceneRough) == 0 || PSDR_SINK_REG_PRIV_ROUGH); }

#ifndef PSDR_WAVES_REV
#define PSDR_WAVES_REV 2
#endif
#ifndef PSDR_WAVES_REV_VALUE
#define PSDR_WAVES_REV_VALUE 3          // value kernel of a split reverse launch
#endif
#ifndef PSDR_WAVES_REV_VALUE_TINY
#define PSDR_WAVES_REV_VALUE_TINY 3     // ... of a PathTracer on a scene without a tree: the recording primal render of psdr_render_c(PSDR_FLAG_KEEP_RECORDS)
#endif
#ifndef PSDR_WAVES_REV_MAT
#define PSDR_WAVES_REV_MAT 3
#endif
// geometry adjoints: 2 waves/SIMD (248 VGPRs for the PathTracer, more for the rough-conductor variants); the
// diffuse DirectIntegrator instance needs 187 and runs faster at 3 (C2 direct all gradients 4.1 -> 3.4 ms), the
// others lose 50-100 % there to spills
#ifndef PSDR_WAVES_REV_DIRECT_PRE
#define PSDR_WAVES_REV_DIRECT_PRE 3     // final pass of the DirectIntegrator's camera term on a two-level scene (hits pre-traced: no walk, no stack)
#endif
template <int FL, bool GEO, int INTEG> constexpr int rev_waves() {
    if (!GEO) {
        if ((FL & (kSceneEnv | kSceneRough)) != 0) return PSDR_WAVES_REV_MAT;
        // plain diffuse variant: 130 VGPRs, C2 texel gradient 3.4 -> 2.8 ms at 4; its instance for scenes without a tree (117 VGPRs) at 5: 2.17 -> 2.09 ms
        return (FL & kSceneTiny) ? PSDR_WAVES_REV_MAT + 2 : PSDR_WAVES_REV_MAT + 1;
    }
    if (INTEG == PSDR_INTEGRATOR_DIRECT && !(FL & kSceneRough) && (FL & kScenePre) != 0) return PSDR_WAVES_REV_DIRECT_PRE;
    return (INTEG == PSDR_INTEGRATOR_DIRECT && !(FL & kSceneRough)) ? 3 : PSDR_WAVES_REV;
}
// STAGE 0: value sweep + adjoint sweep per slot.  Split launch (tree scenes, render_rev): STAGE 1 = the value sweep alone at 3
// workgroups per CU -- the tree walks are latency-bound and the adjoint code's registers hold the fused kernel at 2 -- writing a
// record per path to `disk`; STAGE 2 = the adjoint sweep from that record: no traversal, no stacks in LDS.
template <int FL, bool GEO, int INTEG, int STAGE = 0>
__global__ __launch_bounds__(kBlock, (STAGE == 1 ? (((FL & kSceneTiny) != 0 && INTEG == PSDR_INTEGRATOR_PATH) ? PSDR_WAVES_REV_VALUE_TINY : PSDR_WAVES_REV_VALUE) : rev_waves<FL, GEO, INTEG>())) void k_camera_rev(LaunchCtx cx, DeviceSink<FL> sink_arg, int spp, int s_begin, SlotDiv nsp, long long j0,
                                                       long long n, float inv_spp, const float *__restrict__ adj_img, float *__restrict__ img,
                                                       unsigned long long *counters, float *__restrict__ disk, long long disk_stride, float *__restrict__ deep, int disk_cf,
                                                       ProbeView pv) {
    TraversalStack st; setup_lds(cx, st);
    typename std::conditional<reg_priv_kernel<FL, GEO, INTEG>() && STAGE != 1, RegPrivSink<FL>, DeviceSink<FL> &>::type sink(sink_arg);
    if (STAGE != 1) sink.begin(dyn_lds_floats(cx.off_sink));
    uint32_t nrays = 0;
    const long long nceil = (n + kBlock - 1) / kBlock * kBlock;
#ifdef PSDR_STAGE_CLOCKS
    __shared__ StageClk s_ck[kBlock / 64];
    StageClk &ck = s_ck[threadIdx.x >> 6];
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) { for (int i = 0; i < 12; ++i) ck.t[i] = 0; ck.last = t_begin; }
    if (STAGE != 1 && GEO) st.clk = &ck;          // STAGE 0 (both sweeps in one kernel): phase 1 is the value sweep
#endif
    for (long long jj = (long long) blockIdx.x * kBlock + threadIdx.x; jj < nceil; jj += (long long) gridDim.x * kBlock) {
        const bool in = jj < n;
        int pixel = 0x7fffffff, s_in = 0;
        if (in) slot_to_pixel(j0 + jj, nsp, pixel, s_in);
        float v[3] = {0.f, 0.f, 0.f};
        PrimaryGrad pg; pg.clear();
        PathRec rec;
#if defined(__HIP_DEVICE_COMPILE__)
        rec.base = reinterpret_cast<float *>(psdr_dyn_lds + cx.off_pathrec) + threadIdx.x;
        if (INTEG != PSDR_INTEGRATOR_DIRECT && deep != nullptr) { rec.deep = deep; rec.deep_stride = gridDim.x * kBlock; rec.deep_col = blockIdx.x * kBlock + threadIdx.x; }
#endif
        if constexpr ((FL & kScenePre) != 0) { if (in) probe_load<3>(st, pv, jj, pv.mask[jj]); }
        if (in) {
            const int s = s_begin + s_in;
            const uint64_t slot = (uint64_t) pixel * (uint64_t) spp + (uint64_t) s;
            Vec3f adj(0.f);                  // the value kernel of a split launch has no use for it (and psdr_render_c's recording render has none)
            if constexpr (STAGE != 1) { const float *a = adj_img + (size_t) pixel * 3; adj = Vec3f{a[0] * inv_spp, a[1] * inv_spp, a[2] * inv_spp}; }
            const RevDisk dk{disk + jj, disk_stride, disk_cf};
            const Vec3f r = camera_sample_reverse<GEO, INTEG, STAGE>(sink, pg, rec, cx.sc, st, cx.lp, cx.jump, pixel, slot, adj, nrays, dk);
            v[0] = r.x * inv_spp; v[1] = r.y * inv_spp; v[2] = r.z * inv_spp;
        }
        // primary-triangle row: one add per run of lanes that hit the same triangle
        if (STAGE != 1 && GEO && sink.g.g_tri_info != nullptr) {
            const bool head = wave_run_sum<kPrimaryWords, (FL & kSceneRough) == 0 || PSDR_DPP_ALWAYS>(pg.tri, pg.w);
            if (head && pg.tri >= 0) {
#pragma unroll
                for (int w = 0; w < kPrimaryWords; ++w) sink.add_tri(pg.tri, w, pg.w[w]);
            }
        }
        PSDR_CLK_MARK_ST(st, 6);          // the primary triangle's row (wave run sums)
        // the complete row adjoints of this slot's path vertices (camera_sample_reverse -> complete_row -> DeviceSink::defer_row), sorted by row
        if constexpr (STAGE != 1 && GEO) sink_flush_pending_wave(sink);
        PSDR_CLK_MARK_ST(st, 7);          // the parked rows added sorted
        if (STAGE != 2 && img != nullptr) {
            const bool head = wave_segmented_sum<3>(pixel, v);
            if (head && in) {
                float *p = img + (size_t) pixel * 3;
                if (v[0] != 0.f) atomicAdd(p, v[0]);
                if (v[1] != 0.f) atomicAdd(p + 1, v[1]);
                if (v[2] != 0.f) atomicAdd(p + 2, v[2]);
            }
        }
    }
    if (STAGE != 1) sink.end();
#ifdef PSDR_STAGE_CLOCKS
    if (st.clk) {
        PSDR_CLK_MARK_ST(st, 8);          // the gradient cache flushed
        if ((threadIdx.x & 63) == 0) {
            ck.t[11] = __builtin_amdgcn_s_memtime() - t_begin;
            for (int i = 0; i < 12; ++i) atomicAdd(counters + (blockIdx.x % kRayCounters) * kRayCounterStride + 1 + i, ck.t[i]);
        }
    }
#endif
    count_rays(counters, nrays);
}

// ---- adds of a whole wave SORTED BY KEY (convergent points only: all 64 lanes call).  An LDS float add costs ~3 cycles per ACTIVE LANE whatever the
// addresses (tools/micro/lds_atomics2.hip), and the lanes of a wave land on a handful of rows (the walls of a room): the lanes are ranked by key
// (one ballot per distinct key), values travel to their rank with ds_permute, a segmented scan inside every row of 16 lanes (four v_fmac with DPP
// row_shr operands, masks computed once per key set) leaves each run's total in its last lane, and only those lanes add -- one LDS add per run of
// equal keys and row of 16 instead of one per lane.
#ifndef PSDR_SORTED_ROWS_BATCHED
#define PSDR_SORTED_ROWS_BATCHED 1
#endif
template <int CTRL> __device__ __forceinline__ int dpp_mov_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
template <int CTRL> __device__ __forceinline__ float dpp_mov_f(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true)); }
struct WaveSort {
    int to_bytes;          // this lane's entry travels to lane to_bytes / 4
    int skey;              // key + 1 of the entry that arrived here (0: none)
    float m1, m2, m4, m8;  // 1: the lane 1 / 2 / 4 / 8 to the left (same row of 16) holds the same key
    bool tail;             // last lane of a run: it holds the run's total after the scan
    __device__ __forceinline__ float total(float v) const {
        float s = __int_as_float(__builtin_amdgcn_ds_permute(to_bytes, __float_as_int(v)));
        s = fmaf(dpp_mov_f<0x111>(s), m1, s); s = fmaf(dpp_mov_f<0x112>(s), m2, s); s = fmaf(dpp_mov_f<0x114>(s), m4, s); s = fmaf(dpp_mov_f<0x118>(s), m8, s);
        return s;
    }
};
// key >= 0: this lane has an entry; the caller checked that at least one lane does
__device__ __forceinline__ WaveSort wave_sort_keys(int key) {
    const int lane = threadIdx.x & 63;
    const unsigned long long part = __ballot(key >= 0), below = (1ull << lane) - 1ull;
    unsigned long long rem = part;
    int rank = 0, base = 0;
    while (rem) {
        const int t = __builtin_amdgcn_readlane(key, __ffsll((long long) rem) - 1);
        const unsigned long long same = __ballot(key == t);
        if (key == t) rank = base + (int) __popcll(same & below);
        base += (int) __popcll(same);
        rem &= ~same;
    }
    if (key < 0) rank = base + (int) __popcll(~part & below);                 // a full permutation: the lanes without an entry fill the tail
    WaveSort ws;
    ws.to_bytes = rank << 2;
    ws.skey = __builtin_amdgcn_ds_permute(ws.to_bytes, key + 1);
    const int k = ws.skey;
    ws.m1 = (k != 0 && dpp_mov_i<0x111>(k) == k) ? 1.f : 0.f; ws.m2 = (k != 0 && dpp_mov_i<0x112>(k) == k) ? 1.f : 0.f;
    ws.m4 = (k != 0 && dpp_mov_i<0x114>(k) == k) ? 1.f : 0.f; ws.m8 = (k != 0 && dpp_mov_i<0x118>(k) == k) ? 1.f : 0.f;
    ws.tail = k != 0 && dpp_mov_i<0x101>(k) != k;                              // row_shl:1 -- the lane to the right (0 at the end of a row of 16)
    return ws;
}
// One complete row adjoint per lane (tri < 0: none): position at (u, v), face normal, area -- the 13 words of scatter_row.
template <class S> __device__ __forceinline__ void sink_add_row_wave(S &sink, int tri, float u, float v, const RowAdj &r) {
    bool valid = tri >= 0 && sink.g.g_tri_info != nullptr;
    if (__ballot(valid) == 0ull) return;
    const Vec3f p{S::finite(r.p.x), S::finite(r.p.y), S::finite(r.p.z)}, fn{S::finite(r.fn.x), S::finite(r.fn.y), S::finite(r.fn.z)};
    const float area = S::finite(r.area);
    if (valid && sink.add_row(tri, u, v, p, fn, area)) valid = false;          // the emitter's rows: lane-private accumulators
    int slot = -1;
    if (valid) {
        if constexpr ((S::flags & kSceneTiny) != 0 && PSDR_TINY_DIRECT_ROWS) slot = tri;
        else { slot = sink.L.hot_rows ? sink.L.hot_map[tri] : -1; if (slot >= sink.L.hot_rows) slot = -1; }
    }
    const float w[kPrivRowWords] = {p.x, p.y, p.z, u * p.x, u * p.y, u * p.z, v * p.x, v * p.y, v * p.z, fn.x, fn.y, fn.z, area};
    if (valid && slot < 0) {
        // a row outside the cache (the triangles of the large meshes): one hardware atomic per word, as DeviceSink::add_tri
#pragma unroll
        for (int i = 0; i < kPrivRowWords; ++i) if (w[i] != 0.f) atomicAdd(sink.g.g_tri_info + (size_t) tri * PSDR_TRI_STRIDE + (i < 9 ? i : i + 9), w[i]);
    }
    if (__ballot(slot >= 0) == 0ull) return;
    // (measured and dropped: ranking by one ballot per POSSIBLE key in a loop of uniform length for the <= 32 rows of a scene without a tree -- 4.97 ms either way)
    const WaveSort ws = wave_sort_keys(slot);
    typename S::lds_float *row = sink.lds + sink.L.hot_off + (ws.skey - 1) * PSDR_TRI_STRIDE;
#if PSDR_SORTED_ROWS_BATCHED
    // the 13 words travel together: all permutes in flight at once, the 13 independent scans interleaved, then the adds of the run tails (word by word the
    // wave waited for every permute and every scan alone: C2 all gradients 5.14 -> 4.97 ms, the C4 shard's reverse 35.0 -> 34.4; profiles/r05_rev_sorted_abk.txt)
    float t[kPrivRowWords];
#pragma unroll
    for (int i = 0; i < kPrivRowWords; ++i) t[i] = __int_as_float(__builtin_amdgcn_ds_permute(ws.to_bytes, __float_as_int(slot >= 0 ? w[i] : 0.f)));
#pragma unroll
    for (int i = 0; i < kPrivRowWords; ++i) t[i] = fmaf(dpp_mov_f<0x111>(t[i]), ws.m1, t[i]);
#pragma unroll
    for (int i = 0; i < kPrivRowWords; ++i) t[i] = fmaf(dpp_mov_f<0x112>(t[i]), ws.m2, t[i]);
#pragma unroll
    for (int i = 0; i < kPrivRowWords; ++i) t[i] = fmaf(dpp_mov_f<0x114>(t[i]), ws.m4, t[i]);
#pragma unroll
    for (int i = 0; i < kPrivRowWords; ++i) t[i] = fmaf(dpp_mov_f<0x118>(t[i]), ws.m8, t[i]);
    if (ws.tail) {
#pragma unroll
        for (int i = 0; i < kPrivRowWords; ++i) if (t[i] != 0.f) S::lds_add(row + (i < 9 ? i : i + 9), t[i]);
    }
#else
#pragma unroll
    for (int i = 0; i < kPrivRowWords; ++i) {
        const float t = ws.total(slot >= 0 ? w[i] : 0.f);
        if (ws.tail && t != 0.f) S::lds_add(row + (i < 9 ? i : i + 9), t);
    }
#endif
}
// The rows a lane deferred during its slot (DeviceSink::defer_row), added sorted by row at the kernel's convergent point: round r takes every lane's r-th row.
template <class S> __device__ __forceinline__ void sink_flush_pending_wave(S &sink) {
    if (sink.L.pend_rows == 0) return;
    for (int r = 0; r < sink.L.pend_rows; ++r) {
        const bool has = r < sink.n_pending;
        if (__ballot(has) == 0ull) break;
        int tri = -1; float u = 0.f, v = 0.f; RowAdj row; row.clear();
        if (has) {
            const typename S::lds_float *q = sink.lds0 + sink.L.pend_off + threadIdx.x + r * (kPendWords * kBlock);
            tri = __float_as_int(q[0]); u = q[kBlock]; v = q[2 * kBlock];
            row.p.x = q[3 * kBlock]; row.p.y = q[4 * kBlock]; row.p.z = q[5 * kBlock]; row.fn.x = q[6 * kBlock]; row.fn.y = q[7 * kBlock]; row.fn.z = q[8 * kBlock]; row.area = q[9 * kBlock];
        }
        sink_add_row_wave(sink, tri, u, v, row);
    }
    sink.n_pending = 0;
}

// The primary-edge term only produces gradients of the edge table (the two Li values are detached): a sink
// without the LDS gradient cache, so the kernel is not held at 2 workgroups per CU by 24 KB of static LDS.
// The long edges of a scene (the box walls) take most of the length-weighted samples, the slots are pixel-sorted, so wave after wave
// adds to the same 32-byte row: same-line L2 atomics serialise and cost 3.9 of this kernel's 5.1 ms on cbox_bunny (C3, 4 M slots).  The
// adds go to one of `reps` copies of the table (by workgroup), k_sum_replicas folds them into the caller's table afterwards.
template <int FL> struct PrimaryEdgeSink {
    static constexpr int flags = FL;
    static constexpr bool has_env = (FL & kSceneEnv) != 0;
    float *g_prim_edge;       // the caller's table (reps == 1) or the first copy
    long long rep_stride;     // floats between copies
    int reps;
    __device__ __forceinline__ void add_pedge(int e, int word, float v) const {
        if (v != 0.f && isfinite(v)) atomicAdd(g_prim_edge + (size_t) (blockIdx.x % (unsigned) reps) * rep_stride + (size_t) e * PSDR_PEDGE_STRIDE + word, v);
    }
};
__global__ __launch_bounds__(kBlock) void k_sum_replicas(float *__restrict__ dst, const float *__restrict__ rep, long long n, int reps) {
    const long long i = (long long) blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int r = 0; r < reps; ++r) s += rep[(size_t) r * n + i];
    if (s != 0.f) dst[i] += s;
}
template <int FL, int INTEG>
__global__ __launch_bounds__(kBlock, PSDR_WAVES_PE) void k_primary_edge_rev(LaunchCtx cx, PrimaryEdgeSink<FL> sink, long long i0, long long n, float inv_sppe,
                                                                            const float *__restrict__ adj_img, unsigned long long *counters,
                                                                            const uint32_t *__restrict__ order) {
    TraversalStack st; setup_lds(cx, st);
    uint32_t nrays = 0;
    const long long nceil = (n + kBlock - 1) / kBlock * kBlock;
    for (long long j = (long long) blockIdx.x * kBlock + threadIdx.x; j < nceil; j += (long long) gridDim.x * kBlock) {
        float w[4] = {0.f, 0.f, 0.f, 0.f};
        int edge = -1;
        if (j < n)
            edge = primary_edge_reverse_values<INTEG, FL>(cx.sc, st, cx.lp, cx.jump, (uint64_t) (i0 + (order ? (long long) order[j] : j)), inv_sppe, adj_img,
                                                          nrays, w);
        // pixel-sorted slots: neighbouring lanes mostly sit on the same edge -> one atomic per run of equal edges
        const bool head = wave_run_sum<4>(edge, w);
        if (head && edge >= 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) sink.add_pedge(edge, i, w[i]);
        }
    }
    count_rays(counters, nrays);
}

template <int FL>
__global__ __launch_bounds__(kBlock) void k_secondary_edge_rev(LaunchCtx cx, DeviceSink<FL> sink, long long i0, long long n, float inv_sppse,
                                                               const float *__restrict__ adj_img, unsigned long long *counters,
                                                               const uint32_t *__restrict__ list, const int *__restrict__ list_n) {
    TraversalStack st; setup_lds(cx, st);
    sink.begin(dyn_lds_floats(cx.off_sink));
    uint32_t nrays = 0;
    const bool guided = cx.sc.d.guide_cmf != nullptr && cx.sc.d.num_guide_cells > 0;
    if (list != nullptr) n = *list_n;
    for (long long jj = (long long) blockIdx.x * kBlock + threadIdx.x; jj < n; jj += (long long) gridDim.x * kBlock) {
        const long long j = list != nullptr ? (long long) list[jj] : jj;
        Rng rng; rng.init((uint64_t) (i0 + j), cx.jump);
        float s3[3] = {rng.next(), rng.next(), rng.next()};
        const float pdf0 = guided ? guide_sample_reuse(cx.sc, s3) : 1.f;
        secondary_edge_reverse(sink, cx.sc, st, s3, (pdf0 > kEpsilon ? 1.f / pdf0 : 1.f) * inv_sppse, adj_img, nrays, list == nullptr);
    }
    sink.end();
    count_rays(counters, nrays);
}


// ============================================================================ launch drivers
// Grid of the forward camera kernels: finer workgroups balance image regions of different cost better (C2 +4.5 %,
// C4 PathTracer(3) shard 37.4 -> 34.2 ms at 40 per CU instead of 16), but every workgroup stages its share of the
// scene in LDS first: on a 4 M-slot launch of an open scene (bunny_light) 40 per CU is 9-15 % SLOWER.  So: 40 when
// nothing is staged (tiny scene) or when a workgroup still makes >= 8 trips of its grid-stride loop, else 16.
inline int camera_blocks_per_cu(const psdr_scene_s *h, long long n) {
    if (h->opt.camera_blocks > 0) return h->opt.camera_blocks;          // experiments (psdr_scene_set_option)
    if (h->has_rough) return 16;
    if (h->n_tiny > 0 && h->n_blas == 0) return 40;
    // tree scenes (lean variants): pixels on a mesh cost several times a wall pixel and neighbouring pixels share a workgroup, so the hardware's
    // workgroup scheduler balances the launch better the finer the grid -- two slots per thread, 16..64 workgroups per CU (cbox_bunny 256^2 spp 64:
    // PathTracer(3) 2.51 -> 2.37 ms, (6) 4.17 -> 3.90 against 16; the C4 shard's fused launch 25.8 -> 24.8 against 40)
    const long long fit = n / ((long long) kBlock * h->num_cus * 2);
    return (int) std::max(16LL, std::min(64LL, fit));
}
// slots per chunk of the launches that keep per-slot state between kernels (wavefront streams, reverse-mode records, probe buffers): 2^log2_default,
// or what psdr_scene_set_option("chunk_log2", ...) says (tests: chunk boundaries on small scenes)
// workgroups per CU of the large launches that measured faster on a finer grid (traced wavefront stages, primary-edge reverse kernel, secondary-edge filter of a
// two-level scene): 16, up to 40 where a workgroup still makes >= 8 trips of its grid-stride loop (profiles/r04_wf_grid_abk.txt)
inline int big_launch_per_cu(const psdr_scene_s *h, long long n) { return (int) std::max(16LL, std::min(40LL, n / ((long long) kBlock * h->num_cus * 8))); }
inline long long launch_chunk(const psdr_scene_s *h, int log2_default) { return 1ll << (h->opt.chunk_log2 > 0 ? h->opt.chunk_log2 : log2_default); }
// DirectIntegrator(1, 1) camera launches on a two-level scene run as probe pass + dense trace kernel + final pass (from 2^16 slots)
inline bool probe_direct(const psdr_scene_s *h, const psdr_render_opts *o, long long n) {
    return traced_wavefront(h) && h->opt.probe != 0 && o->bsdf_samples == 1 && o->light_samples == 1 && !(o->flags & PSDR_FLAG_FUSED) && n >= (1ll << 16);
}
template <class G, class R, int FL>
int run_camera(psdr_scene_s *h, const psdr_render_opts *o, const TV<R, FL> &tv, float *img, float *dimg, hipStream_t s) {
    const long long WH = (long long) h->desc.width * h->desc.height;
    const int nsp = o->spp_end - o->spp_begin;
    if (o->spp <= 0 || nsp <= 0) return 0;
    LaunchCtx cx;
    if (int rc = make_ctx(h, o, 0, cx)) return rc;
    const long long n = WH * nsp;
    h->slots[0] += (uint64_t) n;
    // every pixel's samples of the launch in one wave: the camera kernels store their pixels instead of adding to them (k_camera `own`; chunks start at multiples of 2^24)
    const int own = (h->opt.own_pixels != 0 && nsp <= 64 && 64 % nsp == 0) ? 1 : 0;
    // DirectIntegrator(1, 1) on a two-level scene: probe pass (primary hit + requests for the vertex' two rays that enter a tree box) -> dense trace
    // kernel -> this kernel compiled with kScenePre (no tree walk: primitives + hit rows), chunk by chunk.  C3 forward geometry duals 1.31 -> 0.73 ms
    // (0.21 + 0.19 + 0.33), C4 shard reverse 16.0 -> 14.0 ms
    if constexpr ((FL & kSceneForest) != 0) {
        if (o->integrator == PSDR_INTEGRATOR_DIRECT && probe_direct(h, o, n)) {
            const TangentView<0, FL> tv0{};
            TV<R, FL | kScenePre> tvp;
            for (int k = 0; k < (ad_traits<R>::K > 0 ? ad_traits<R>::K : 1); ++k) tvp.t[k] = tv.t[k];
            tvp.live = tv.live;
            LaunchCtx cxp = cx;
            plan_lds(h, cxp, 1 << 30);
            cxp.sc.n_lnodes = cxp.sc.n_lbtris = cxp.sc.n_ltri = 0;
            const long long chunk = std::min<long long>(n, launch_chunk(h, 25));
            for (long long c0 = 0; c0 < n; c0 += chunk) {
                const long long nc = std::min(chunk, n - c0);
                ProbeBuffers pb;
                const int probe_per_cu = big_launch_per_cu(h, nc);          // k_direct_probe 951 -> 879 us at 40 per CU on the C4 shard (profiles/r04_wf_grid_abk.txt)
                if (int rc = probe_buffers(h, nc, 3, pb, s, probe_per_cu)) return rc;
                const TraceQueue tq{pb.req, pb.count, pb.sub_cap};
                const int blocks = (launch_blocks(h, nc, probe_per_cu) + kWfSub - 1) / kWfSub * kWfSub;
                hipLaunchKernelGGL(k_direct_probe<FL>, dim3(blocks), dim3(kBlock), lds_bytes(cx, h), s, cx, tv0, o->spp, o->spp_begin, SlotDiv(nsp), c0, nc, tq, pb.hit, pb.mask,
                                   make_rng_jump(o->rng_offset[0] + 2));
                HIP_TRY(hipGetLastError());
                if (int rc = launch_wf_trace(h, pb.req, pb.count, pb.sub_cap, pb.hit, s)) return rc;
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_camera<G, R, PSDR_INTEGRATOR_DIRECT, FL | kScenePre, false>), dim3(launch_blocks(h, nc, camera_blocks_per_cu(h, nc))), dim3(kBlock), cxp.off_stack, s,
                                   cxp, tvp, o->spp, o->spp_begin, SlotDiv(nsp), nc, 1.f / (float) o->spp, img, dimg, WH * 3, h->d_counters, c0, ProbeView{pb.hit, pb.mask}, own);
                HIP_TRY(hipGetLastError());
            }
            return 0;
        }
    }
    // PathTracer, material tangents on albedo texels only, a variant without rough conductors / environment map: the log-derivative kernel and, behind the
    // same gate, the dual-number kernel (one of the two returns at once)
    bool gated = false;
    if constexpr (logd_instance<G, R, PSDR_INTEGRATOR_PATH, FL>()) {
        constexpr int K = ad_traits<R>::K;
        bool texels_only = o->integrator == PSDR_INTEGRATOR_PATH && h->opt.logd != 0 && h->desc.num_texels > 0 && h->desc.texels != nullptr;
        for (int k = 0; k < K && texels_only; ++k) {
            const psdr_tangents &t = tv.t[k];
            texels_only = t.d_texels != nullptr && !t.d_tri_info && !t.d_emitter_rad && !t.d_cam_to_world && !t.d_sec_edge && !t.d_prim_edge && !t.d_env_f;
        }
        if (texels_only) {
            if (int rc = scratch_reserve(&h->d_logd_bad, &h->logd_bad_bytes, sizeof(int), s, "log-derivative gate")) return rc;
            int *bad = reinterpret_cast<int *>(h->d_logd_bad);
            HIP_TRY(hipMemsetAsync(bad, 0, sizeof(int), s));
            TangentView<K, 0> tvk;
            for (int k = 0; k < K; ++k) tvk.t[k] = tv.t[k];
            hipLaunchKernelGGL(k_logd_check<K>, dim3((unsigned) ((h->desc.num_texels + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, h->desc.texels, tvk, h->desc.num_texels, h->d_counters, bad);
            hipLaunchKernelGGL(k_logd_gate, dim3(1), dim3(64), 0, s, h->d_counters, bad);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_camera_logd<K, FL, ((FL & kSceneTiny) != 0)>), dim3(launch_blocks(h, n, camera_blocks_per_cu(h, n))), dim3(kBlock), lds_bytes(cx, h), s, cx, tv,
                               o->spp, o->spp_begin, SlotDiv(nsp), n, 1.f / (float) o->spp, img, dimg, WH * 3, h->d_counters, own);
            HIP_TRY(hipGetLastError());
            gated = true;
        }
    }
#define PSDR_LAUNCH_CAMERA_T(INTEG, NOTREE)                                                                                        \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_camera<G, R, INTEG, FL, NOTREE>), dim3(launch_blocks(h, n, camera_blocks_per_cu(h, n))), dim3(kBlock), lds_bytes(cx, h), s, cx, tv, o->spp, \
                       o->spp_begin, nsp, n, 1.f / (float) o->spp, img, dimg, WH * 3, h->d_counters, 0ll, ProbeView{nullptr, nullptr}, own)
#define PSDR_LAUNCH_CAMERA(INTEG) PSDR_LAUNCH_CAMERA_T(INTEG, ((FL & kSceneTiny) != 0))
    // scenes without a tree are served by their own flag sets (kSceneTiny): occupancy follows from FL alone
    switch (o->integrator) {
        case PSDR_INTEGRATOR_DIRECT: PSDR_LAUNCH_CAMERA(PSDR_INTEGRATOR_DIRECT); break;
        case PSDR_INTEGRATOR_PATH: PSDR_LAUNCH_CAMERA(PSDR_INTEGRATOR_PATH); break;
        case PSDR_INTEGRATOR_FIELD: PSDR_LAUNCH_CAMERA(PSDR_INTEGRATOR_FIELD); break;
        default: return fail("Unknown integrator");
    }
#undef PSDR_LAUNCH_CAMERA
#undef PSDR_LAUNCH_CAMERA_T
    HIP_TRY(hipGetLastError());
    // the gate word belongs to THIS kernel pair: closed again behind it, so that a later launch of a dual-number instance in the same call (another chunk, another
    // tangent group) is not silenced by it
    if (gated) HIP_TRY(hipMemsetAsync(h->d_counters + kLogdGateWord, 0, sizeof(unsigned long long), s));
    return 0;
}

// PathTracer interior term as a wavefront (see k_wf_camera / k_wf_bounce).  M = float or Dual<K>
// with plain-fp32 geometry.

constexpr int kWfMaxDepth = 256;         // counter sets (use_wavefront: max_depth <= 250)
// rec != nullptr: the launch is the VALUE SWEEP of a split reverse launch over the slots [rec->c0, rec->c0 + rec->nc) of the shard -- the traced
// wavefront with REC kernels that leave the per-path records the adjoint kernel reads (WfRec); float only, two-level scenes only.
struct WfRecArgs { float *disk; long long stride, c0, nc; };
template <class M, int FL>
int run_camera_wavefront(psdr_scene_s *h, const psdr_render_opts *o, const TV<M, FL> &tv, float *img, float *dimg, hipStream_t s, const WfRecArgs *rec = nullptr) {
    constexpr int K = ad_traits<M>::K;
    const long long WH = (long long) h->desc.width * h->desc.height;
    const int nsp = o->spp_end - o->spp_begin;
    if (o->spp <= 0 || nsp <= 0) return 0;
    const long long n_all = WH * nsp;
    const long long n = rec ? rec->nc : n_all;
    // class-binned streams pay where most paths survive every bounce (rooms); in an open scene the streams thin out quickly and the 64
    // sub-streams of mostly empty chunks cost more than the classes save (bunny_light PathTracer(3) 4.9 against 3.3 ms plain,
    // PathTracer(6) 8.7 against 3.6)
    // two-level scenes: the TRACED wavefront -- tree walks in the dense trace kernel between the stages (psdr_hip.hip k_wf_trace), plain streams;
    // PSDR_WF_TRACED=0 (psdr_scene_create) falls back to the class-binned streams of round 2
    bool traced = false;
    if constexpr ((FL & kSceneForest) != 0) traced = traced_wavefront(h);
    const bool binned = !traced && (FL & kSceneForest) != 0 && h->n_blas > 0 && h->wf_binned;          // a two-level scene is a room (use_wavefront)
    // chunk of the traced wavefront: 2^26 slots (240 B of workspace per slot: 16 GB of the 288) -- the stage kernels are persistent and pay their
    // ramp-up and tail once per launch; the C4 shard (2^26 slots) as ONE chunk 14.3 -> 14.1 ms of kernel time, and every smaller chunk is slower
    // still (2^24: 15.0, 2^22: 19.7, 2^20: 34.4 -- the streams staying inside the 256 MB last-level cache buys nothing; profiles/r04_chunk_sweep.txt)
    const int depth = o->max_depth;
    const size_t words = 8 + 6 * (1 + K) + (traced ? 8 : 0);          // traced: + the two hit rows of a record
    // (a chunk that does not fit the device right now is halved until it does: fit_chunk; the records of a recording launch fix the chunk themselves)
    const long long cap = rec ? n : std::min(n, fit_chunk(binned ? launch_chunk(h, 24) : launch_chunk(h, traced ? 26 : 25), 2 * words * 4 * 5 / 4 + (traced ? 80 : 0), 64u << 20, h->ws_bytes));
    // plain: block b appends to sub-stream b % kWfSub, at most ceil(blocks / kWfSub) * trips * kBlock records each.
    // binned: the records of chunk c go to group c % kWfGroups of their class; a class can take all of a group
    // grid of the traced wavefront's stage kernels: 16 workgroups per CU, up to 40 where a workgroup still makes >= 8 trips of its grid-stride loop (finer
    // workgroups balance the stages' uneven records better: C4 shard, 2^26 slots, 13.99 -> 13.45 ms of kernel time at 40, 13.59 at 24, 13.65 at 80; C5's
    // 2^22 slots are best at 10-16 and 9 % slower at 40 -- profiles/r04_wf_grid_abk.txt)
    const auto wf_per_cu = [&](long long slots) { return traced ? big_launch_per_cu(h, slots) : 16; };
    const long long max_blocks = ((long long) launch_blocks(h, cap, wf_per_cu(cap)) + kWfSub - 1) / kWfSub * kWfSub;   // >= the grid of any chunk
    const long long binned_sub_cap = (((cap + kBlock - 1) / kBlock + kWfSub) / kWfGroups + 2) * kBlock;
    const long long cap_alloc = ((binned ? binned_sub_cap * kWfSub : cap + max_blocks * kBlock) + 63) / 64 * 64;      // the hit rows are float4
    const size_t cnt_ints = (size_t) (kWfMaxDepth + 1) * kWfStageInts + (traced ? (size_t) (kWfMaxDepth + 1) * kWfSub * kWfCountStride : 0);
    const size_t cnt_bytes = cnt_ints * sizeof(int32_t);
    const size_t req_bytes = traced ? (size_t) 2 * cap_alloc * 2 * sizeof(float4) : 0;      // at most two requests per record, two rows each
    const size_t need = 2 * words * 4 * (size_t) cap_alloc + cnt_bytes + req_bytes;
    if (int rc = workspace_reserve(h, need, s)) return rc;
    int32_t *cnt = reinterpret_cast<int32_t *>(h->d_ws);
    int32_t *req_cnt = cnt + (size_t) (kWfMaxDepth + 1) * kWfStageInts;
    PathStream st[2];
    for (int i = 0; i < 2; ++i) {
        float *b = reinterpret_cast<float *>(reinterpret_cast<char *>(h->d_ws) + cnt_bytes) + (size_t) i * words * cap_alloc;
        const long long c = cap_alloc;
        st[i].cap = c;
        st[i].pixel = reinterpret_cast<int32_t *>(b); st[i].slot = reinterpret_cast<uint32_t *>(b + c); st[i].tri = reinterpret_cast<int32_t *>(b + 2 * c);
        st[i].hu = b + 3 * c; st[i].hv = b + 4 * c; st[i].dir = b + 5 * c; st[i].beta = b + 8 * c; st[i].acc = b + (8 + 3 * (1 + K)) * c;
        st[i].hit = traced ? reinterpret_cast<float4 *>(b + (8 + 6 * (1 + K)) * c) : nullptr;
        st[i].prev_t = nullptr;
        st[i].binned = binned ? 1 : 0;
    }
    TraceQueue tq{};
    tq.req = traced ? reinterpret_cast<float4 *>(reinterpret_cast<char *>(h->d_ws) + cnt_bytes + 2 * words * 4 * (size_t) cap_alloc) : nullptr;
    if (!rec) h->slots[0] += (uint64_t) n;
    const float inv_spp = 1.f / (float) o->spp;
    const long long j_first = rec ? rec->c0 : 0;
    for (long long j0 = j_first; j0 < j_first + n; j0 += cap) {
        const long long cn = std::min(cap, j_first + n - j0);
        LaunchCtx cx;
        if (int rc = make_ctx(h, o, 0, cx)) return rc;
        HIP_TRY(hipMemsetAsync(cnt, 0, (size_t) (std::min(depth, kWfMaxDepth) + 1) * kWfStageInts * sizeof(int32_t), s));
        if (traced) HIP_TRY(hipMemsetAsync(req_cnt, 0, (size_t) (std::min(depth, kWfMaxDepth) + 1) * kWfSub * kWfCountStride * sizeof(int32_t), s));
        const int blocks = (launch_blocks(h, cn, wf_per_cu(cn)) + kWfSub - 1) / kWfSub * kWfSub;
        const long long trips = (cn + (long long) blocks * kBlock - 1) / ((long long) blocks * kBlock);
        st[0].sub_cap = st[1].sub_cap = binned ? binned_sub_cap : (blocks / kWfSub) * trips * kBlock;
        st[0].count = cnt;
        if constexpr ((FL & kSceneForest) != 0) {
            if (traced) {
                // camera stage = primary hit (its walk stays in the kernel: camera rays are coherent) + the requests of bounce stage 0; then per
                // bounce: trace kernel (requests -> hit rows of the stream they belong to), bounce stage on the records + their hit rows
                tq.sub_cap = 2 * st[0].sub_cap;
                tq.count = req_cnt;
                // the bounce stages walk no tree: nothing of it staged, no stacks -- the hit rows of the kernel-argument primitives and the small tables only
                LaunchCtx cxp = cx;
                plan_lds(h, cxp, 1 << 30);
                cxp.sc.n_lnodes = cxp.sc.n_lbtris = cxp.sc.n_ltri = 0;
                const int dyn_p = cxp.off_stack;
                WfRec wr{};
                bool recording = false;
                if constexpr (K == 0) {
                    if (rec) { recording = true; wr = WfRec{rec->disk + (j0 - rec->c0), rec->stride, o->spp, o->spp_begin, nsp, j0, 0}; }
                }
                if (recording) {
                    if constexpr (K == 0)
                        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wf_camera<M, FL, true, true>), dim3(blocks), dim3(kBlock), lds_bytes(cx, h), s, cx, tv, o->spp, o->spp_begin, nsp, j0, cn,
                                           inv_spp, img, dimg, WH * 3, st[0], 1 | (depth == 1 ? 2 : 0), h->d_counters, make_rng_jump(o->rng_offset[0] + 2), tq, wr);
                } else
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wf_camera<M, FL, true>), dim3(blocks), dim3(kBlock), lds_bytes(cx, h), s, cx, tv, o->spp, o->spp_begin, nsp, j0, cn,
                                   inv_spp, img, dimg, WH * 3, st[0], 1 | (depth == 1 ? 2 : 0), h->d_counters, make_rng_jump(o->rng_offset[0] + 2), tq, wr);
                HIP_TRY(hipGetLastError());
                for (int k = 0; k < depth; ++k) {
                    if (int rc = launch_wf_trace(h, tq.req, tq.count, tq.sub_cap, st[k & 1].hit, s)) return rc;
                    st[(k + 1) & 1].count = cnt + (size_t) (k + 1) * kWfStageInts;
                    tq.count = req_cnt + (size_t) (k + 1) * kWfSub * kWfCountStride;
                    cxp.jump = make_rng_jump(o->rng_offset[0] + 2 + 5 * (uint64_t) k);
                    wr.stage = k;
                    if (recording) {
                        if constexpr (K == 0)
                            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wf_bounce<M, FL, true, true>), dim3(blocks), dim3(kBlock), dyn_p, s, cxp, tv, inv_spp, img, dimg, WH * 3,
                                               st[k & 1], st[(k + 1) & 1], (k + 1 < depth ? 1 : 0) | (k + 2 == depth ? 2 : 0), h->d_counters, make_rng_jump(o->rng_offset[0] + 2 + 5 * (uint64_t) (k + 1)), tq, wr);
                    } else
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wf_bounce<M, FL, true>), dim3(blocks), dim3(kBlock), dyn_p, s, cxp, tv, inv_spp, img, dimg, WH * 3,
                                       st[k & 1], st[(k + 1) & 1], (k + 1 < depth ? 1 : 0) | (k + 2 == depth ? 2 : 0), h->d_counters, make_rng_jump(o->rng_offset[0] + 2 + 5 * (uint64_t) (k + 1)), tq, wr);
                    HIP_TRY(hipGetLastError());
                }
                continue;
            }
        }
        if (binned) {
            // camera stage = primary hit only; the direct step at the primary vertex is bounce stage 0 (binned like the rest):
            // stage k reads stream k & 1 (counter set k) and appends to stream (k + 1) & 1 (set k + 1)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wf_camera<M, FL>), dim3(blocks), dim3(kBlock), lds_bytes(cx, h), s, cx, tv, o->spp, o->spp_begin, nsp, j0, cn,
                               inv_spp, img, dimg, WH * 3, st[0], 1 | (depth == 1 ? 2 : 0), h->d_counters, make_rng_jump(o->rng_offset[0] + 2), tq, WfRec{});
            HIP_TRY(hipGetLastError());
            for (int k = 0; k < depth; ++k) {
                st[(k + 1) & 1].count = cnt + (size_t) (k + 1) * kWfStageInts;
                cx.jump = make_rng_jump(o->rng_offset[0] + 2 + 5 * (uint64_t) k);
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wf_bounce<M, FL>), dim3(blocks), dim3(kBlock), lds_bytes(cx, h), s, cx, tv, inv_spp, img, dimg, WH * 3,
                                   st[k & 1], st[(k + 1) & 1], (k + 1 < depth ? 1 : 0) | (k + 2 == depth ? 2 : 0), h->d_counters, make_rng_jump(o->rng_offset[0] + 2 + 5 * (uint64_t) (k + 1)), tq, WfRec{});
                HIP_TRY(hipGetLastError());
            }
            continue;
        }
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wf_camera<M, FL>), dim3(blocks), dim3(kBlock), lds_bytes(cx, h), s, cx, tv, o->spp, o->spp_begin, nsp, j0, cn,
                           inv_spp, img, dimg, WH * 3, st[0], (depth > 1 ? 1 : 0) | (depth == 2 ? 2 : 0), h->d_counters, make_rng_jump(o->rng_offset[0] + 2 + 5), tq, WfRec{});
        HIP_TRY(hipGetLastError());
        for (int k = 1; k < depth; ++k) {
            st[k & 1].count = cnt + (size_t) k * kWfStageInts;      // a fresh (zeroed) counter set per stage
            cx.jump = make_rng_jump(o->rng_offset[0] + 2 + 5 * (uint64_t) k);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wf_bounce<M, FL>), dim3(blocks), dim3(kBlock), lds_bytes(cx, h), s, cx, tv, inv_spp, img, dimg, WH * 3,
                               st[(k - 1) & 1], st[k & 1], (k + 1 < depth ? 1 : 0) | (k + 2 == depth ? 2 : 0), h->d_counters, make_rng_jump(o->rng_offset[0] + 2 + 5 * (uint64_t) (k + 1)), tq, WfRec{});
            HIP_TRY(hipGetLastError());
        }
    }
    return 0;
}

// PathTracer renderD forward with GEOMETRY tangents on a two-level scene: the traced wavefront with dual-number stages (k_wfg_camera / k_wfg_bounce).
template <int K, int FL>
int run_camera_wavefront_geo(psdr_scene_s *h, const psdr_render_opts *o, const TangentView<K, FL> &tv, float *img, float *dimg, hipStream_t s) {
    const long long WH = (long long) h->desc.width * h->desc.height;
    const int nsp = o->spp_end - o->spp_begin;
    if (o->spp <= 0 || nsp <= 0) return 0;
    const long long n = WH * nsp;
    const int depth = o->max_depth;
    const size_t words = 8 + 6 * (1 + K) + 3 * K + 8;             // record + the tangents of the position behind it + the two hit rows
    const long long cap = std::min(n, fit_chunk(launch_chunk(h, 26), 2 * words * 4 * 5 / 4 + 80, 64u << 20, h->ws_bytes));
    const auto wf_per_cu = [&](long long slots) { return big_launch_per_cu(h, slots); };
    const long long max_blocks = ((long long) launch_blocks(h, cap, wf_per_cu(cap)) + kWfSub - 1) / kWfSub * kWfSub;
    const long long cap_alloc = ((cap + max_blocks * kBlock) + 63) / 64 * 64;
    const size_t cnt_ints = (size_t) (kWfMaxDepth + 1) * kWfStageInts + (size_t) (kWfMaxDepth + 1) * kWfSub * kWfCountStride;
    const size_t cnt_bytes = cnt_ints * sizeof(int32_t);
    const size_t req_bytes = (size_t) 2 * cap_alloc * 2 * sizeof(float4);
    const size_t need = 2 * words * 4 * (size_t) cap_alloc + cnt_bytes + req_bytes;
    if (int rc = workspace_reserve(h, need, s)) return rc;
    int32_t *cnt = reinterpret_cast<int32_t *>(h->d_ws);
    int32_t *req_cnt = cnt + (size_t) (kWfMaxDepth + 1) * kWfStageInts;
    PathStream st[2];
    for (int i = 0; i < 2; ++i) {
        float *b = reinterpret_cast<float *>(reinterpret_cast<char *>(h->d_ws) + cnt_bytes) + (size_t) i * words * cap_alloc;
        const long long c = cap_alloc;
        st[i].cap = c;
        st[i].pixel = reinterpret_cast<int32_t *>(b); st[i].slot = reinterpret_cast<uint32_t *>(b + c); st[i].tri = reinterpret_cast<int32_t *>(b + 2 * c);
        st[i].hu = b + 3 * c; st[i].hv = b + 4 * c; st[i].dir = b + 5 * c; st[i].beta = b + 8 * c; st[i].acc = b + (8 + 3 * (1 + K)) * c;
        st[i].prev_t = b + (8 + 6 * (1 + K)) * c;
        st[i].hit = reinterpret_cast<float4 *>(b + (8 + 6 * (1 + K) + 3 * K) * c);
        st[i].binned = 0;
    }
    TraceQueue tq{};
    tq.req = reinterpret_cast<float4 *>(reinterpret_cast<char *>(h->d_ws) + cnt_bytes + 2 * words * 4 * (size_t) cap_alloc);
    h->slots[0] += (uint64_t) n;
    const float inv_spp = 1.f / (float) o->spp;
    for (long long j0 = 0; j0 < n; j0 += cap) {
        const long long cn = std::min(cap, n - j0);
        LaunchCtx cx;
        if (int rc = make_ctx(h, o, 0, cx)) return rc;
        HIP_TRY(hipMemsetAsync(cnt, 0, (size_t) (std::min(depth, kWfMaxDepth) + 1) * kWfStageInts * sizeof(int32_t), s));
        HIP_TRY(hipMemsetAsync(req_cnt, 0, (size_t) (std::min(depth, kWfMaxDepth) + 1) * kWfSub * kWfCountStride * sizeof(int32_t), s));
        const int blocks = (launch_blocks(h, cn, wf_per_cu(cn)) + kWfSub - 1) / kWfSub * kWfSub;
        const long long trips = (cn + (long long) blocks * kBlock - 1) / ((long long) blocks * kBlock);
        st[0].sub_cap = st[1].sub_cap = (blocks / kWfSub) * trips * kBlock;
        st[0].count = cnt;
        tq.sub_cap = 2 * st[0].sub_cap;
        tq.count = req_cnt;
        LaunchCtx cxp = cx;                          // the bounce stages walk nothing: no tree staged, no stacks
        plan_lds(h, cxp, 1 << 30);
        cxp.sc.n_lnodes = cxp.sc.n_lbtris = cxp.sc.n_ltri = 0;
        const int dyn_p = cxp.off_stack;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wfg_camera<K, FL>), dim3(blocks), dim3(kBlock), lds_bytes(cx, h), s, cx, tv, o->spp, o->spp_begin, SlotDiv(nsp), j0, cn, inv_spp, img, dimg,
                           WH * 3, st[0], h->d_counters, tq);
        HIP_TRY(hipGetLastError());
        for (int k = 0; k < depth; ++k) {
            if (int rc = launch_wf_trace(h, tq.req, tq.count, tq.sub_cap, st[k & 1].hit, s)) return rc;
            st[(k + 1) & 1].count = cnt + (size_t) (k + 1) * kWfStageInts;
            tq.count = req_cnt + (size_t) (k + 1) * kWfSub * kWfCountStride;
            cxp.jump = make_rng_jump(o->rng_offset[0] + 2 + 5 * (uint64_t) k);
            if (k == 0)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wfg_bounce<K, FL, true>), dim3(blocks), dim3(kBlock), dyn_p, s, cxp, tv, inv_spp, img, dimg, WH * 3, st[0], st[1], k + 1 < depth ? 1 : 0,
                                   h->d_counters, tq);
            else
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wfg_bounce<K, FL, false>), dim3(blocks), dim3(kBlock), dyn_p, s, cxp, tv, inv_spp, img, dimg, WH * 3, st[k & 1], st[(k + 1) & 1],
                                   k + 1 < depth ? 1 : 0, h->d_counters, tq);
            HIP_TRY(hipGetLastError());
        }
    }
    return 0;
}

// Filter pass of a split secondary-edge launch: *list / *list_n on the device, nullptr when the launch is too small to split.
// gg (guide_launch): the slots are those of a guiding-grid build.  On a two-level scene the pass runs as probe -> dense trace kernel -> filter on
// primitives + hit rows, in CHUNKS of 2^25 slots that append to one list (the probe buffers are ~100 B per slot: 6.7 GB for the C4 shard in one piece --
// ADVICE r4; a chunk's buffers are 3.3 GB and reused).
template <int FL>
int secondary_edge_filter(psdr_scene_s *h, const LaunchCtx &cx, long long i0, long long n, const uint32_t **list, const int **list_n, hipStream_t s, const GuideGrid *ggp = nullptr) {
    *list = nullptr; *list_n = nullptr;
    const GuideGrid gg = ggp ? *ggp : GuideGrid{0, 0, 0, 0, 0, 0.f};
    const int split_env = h->opt.sedge_split;
    if (!ggp && (split_env == 0 || (split_env < 0 && n < (1ll << 18)))) return 0;
    if (n > 0x7fffffffLL) return ggp ? fail("psdr_guide_build: more than 2^31 (stream, round) slots") : 0;
    const size_t need = 256 + (size_t) n * sizeof(uint32_t);
    if (int rc = scratch_reserve(&h->d_se_list, &h->se_list_bytes, need, s, "secondary-edge survivor list")) return rc;
    int *cnt = reinterpret_cast<int *>(h->d_se_list);
    uint32_t *lst = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(h->d_se_list) + 256);
    HIP_TRY(hipMemsetAsync(cnt, 0, sizeof(int), s));
    if constexpr ((FL & kSceneForest) != 0) {
        if (traced_wavefront(h) && h->opt.probe != 0) {
            // traced launch: probe pass -> dense trace kernel -> the filter on primitives + hit rows (C4 shard: 12.2 ms as one kernel)
            LaunchCtx cxp = cx;
            plan_lds(h, cxp, 1 << 30);
            cxp.sc.n_lnodes = cxp.sc.n_lbtris = cxp.sc.n_ltri = 0;
            const long long chunk = std::min<long long>(n, launch_chunk(h, 25));
            for (long long c0 = 0; c0 < n; c0 += chunk) {
                const long long nc = std::min(chunk, n - c0);
                ProbeBuffers pb;
                if (int rc = probe_buffers(h, nc, 2, pb, s)) return rc;
                const TraceQueue tq{pb.req, pb.count, pb.sub_cap};
                const int blocks = (launch_blocks(h, nc) + kWfSub - 1) / kWfSub * kWfSub;
                hipLaunchKernelGGL(k_se_probe<FL>, dim3(blocks), dim3(kBlock), cxp.off_stack, s, cxp, i0 + c0, nc, tq, pb.mask, gg);
                HIP_TRY(hipGetLastError());
                if (int rc = launch_wf_trace(h, pb.req, pb.count, pb.sub_cap, pb.hit, s, true)) return rc;
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_secondary_edge_filter<FL | kScenePre>), dim3(launch_blocks(h, nc, big_launch_per_cu(h, nc))), dim3(kBlock), cxp.off_stack, s, cxp, i0 + c0, nc, lst, cnt,
                                   h->d_counters, ProbeView{pb.hit, pb.mask}, gg, c0);
                HIP_TRY(hipGetLastError());
            }
            *list = lst; *list_n = cnt;
            return 0;
        }
    }
    hipLaunchKernelGGL(k_secondary_edge_filter<FL>, dim3(launch_blocks(h, n)), dim3(kBlock), lds_bytes(cx, h), s, cx, i0, n, lst, cnt, h->d_counters, ProbeView{nullptr, nullptr}, gg, 0ll);
    HIP_TRY(hipGetLastError());
    *list = lst; *list_n = cnt;
    return 0;
}

// One bit per triangle: does any of the K tangent sets move its tri_info row (TangentView::live)?  One thread per triangle, one ballot per wave.
struct TangentRows { const float *p[3]; };
__global__ __launch_bounds__(kBlock) void k_tangent_live(TangentRows tr, int K, int T, uint32_t *__restrict__ live) {
    const int tri = blockIdx.x * kBlock + threadIdx.x;
    bool any = false;
    if (tri < T) {
        for (int k = 0; k < K; ++k) {
            const float *p = tr.p[k];
            if (p == nullptr) continue;
            const float4 *q = reinterpret_cast<const float4 *>(p + (size_t) tri * PSDR_TRI_STRIDE);
#pragma unroll
            for (int i = 0; i < PSDR_TRI_STRIDE / 4; ++i) { const float4 v = q[i]; any = any || v.x != 0.f || v.y != 0.f || v.z != 0.f || v.w != 0.f; }
        }
    }
    const unsigned long long m = __ballot(any);
    if ((threadIdx.x & 63) == 0 && tri < T) { live[tri >> 5] = (uint32_t) m; live[(tri >> 5) + 1] = (uint32_t) (m >> 32); }
}

template <int K, int FL>
int render_fwd(psdr_scene_s *h, const psdr_render_opts *o, const psdr_tangents *tangents, float *img, float *dimg, hipStream_t s) {
    const long long WH = (long long) h->desc.width * h->desc.height;
    TangentView<K, FL> tv;
    for (int k = 0; k < K; ++k) tv.t[k] = tangents[k];
    HIP_TRY(hipMemsetAsync(img, 0, sizeof(float) * WH * 3, s));
    HIP_TRY(hipMemsetAsync(dimg, 0, sizeof(float) * WH * 3 * K, s));
    // geometry stays in plain fp32 when only material / emitter tables carry tangents
    bool geo = false;
    for (int k = 0; k < K; ++k) geo = geo || tangents[k].d_tri_info || tangents[k].d_cam_to_world;
    if (geo && PSDR_TANGENT_LIVE && h->opt.tangent_live != 0) {
        static_assert(K <= 3, "TangentRows holds three sets");
        const int T = h->desc.num_tris;
        const size_t need = ((size_t) (T + 63) / 64 * 2 + 2) * sizeof(uint32_t);
        if (int rc = scratch_reserve(&h->d_live, &h->live_bytes, need, s, "tangent liveness mask")) return rc;
        TangentRows tr{};
        for (int k = 0; k < K; ++k) tr.p[k] = tangents[k].d_tri_info;
        hipLaunchKernelGGL(k_tangent_live, dim3((unsigned) ((T + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, tr, K, T, reinterpret_cast<uint32_t *>(h->d_live));
        HIP_TRY(hipGetLastError());
        tv.live = reinterpret_cast<const uint32_t *>(h->d_live);
    }
    bool geo_wavefront = false;
    if constexpr ((FL & kSceneForest) != 0) geo_wavefront = geo && h->opt.wf_geo != 0 && traced_wavefront(h) && use_wavefront(h, o);
    if (geo_wavefront) {
        // geometry tangents of the PathTracer on a two-level scene: the traced wavefront with dual-number stages (C4 shard: 43 ms fused)
        if constexpr ((FL & kSceneForest) != 0) { if (int rc = run_camera_wavefront_geo<K, FL>(h, o, tv, img, dimg, s)) return rc; }
    } else if (geo) { if (int rc = run_camera<Dual<K>, Dual<K>, FL>(h, o, tv, img, dimg, s)) return rc; }
    else if (use_wavefront(h, o)) { if (int rc = run_camera_wavefront<Dual<K>, FL>(h, o, tv, img, dimg, s)) return rc; }
    else { if (int rc = run_camera<float, Dual<K>, FL>(h, o, tv, img, dimg, s)) return rc; }
    if (o->sppe > 0 && o->sppe_end > o->sppe_begin && h->desc.num_prim_edges > 0) {
        LaunchCtx cx;
        if (int rc = make_ctx(h, o, 1, cx)) return rc;
        const long long i0 = WH * o->sppe_begin, n = WH * (o->sppe_end - o->sppe_begin);
        h->slots[1] += (uint64_t) n;
        const uint32_t *order = nullptr;
        if (int rc = primary_edge_order(h, cx, i0, n, &order, s)) return rc;
#define PSDR_LAUNCH_PE(INTEG)                                                                                                        \
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_primary_edge<K, FL, INTEG>), dim3(launch_blocks(h, n)), dim3(kBlock), lds_bytes(cx, h), s, cx, tv, i0, n,   \
                           1.f / (float) o->sppe, dimg, WH * 3, h->d_counters, order)
        switch (o->integrator) {
            case PSDR_INTEGRATOR_DIRECT: PSDR_LAUNCH_PE(PSDR_INTEGRATOR_DIRECT); break;
            case PSDR_INTEGRATOR_PATH: PSDR_LAUNCH_PE(PSDR_INTEGRATOR_PATH); break;
            default: PSDR_LAUNCH_PE(PSDR_INTEGRATOR_FIELD); break;
        }
#undef PSDR_LAUNCH_PE
        HIP_TRY(hipGetLastError());
    }
    if (o->sppse > 0 && o->sppse_end > o->sppse_begin && h->desc.num_sec_edges > 0 && o->integrator == PSDR_INTEGRATOR_DIRECT) {
        LaunchCtx cx;
        if (int rc = make_ctx(h, o, 2, cx)) return rc;
        const long long i0 = WH * o->sppse_begin, n = WH * (o->sppse_end - o->sppse_begin);
        h->slots[2] += (uint64_t) n;
        const uint32_t *list = nullptr; const int *list_n = nullptr;
        if (int rc = secondary_edge_filter<FL>(h, cx, i0, n, &list, &list_n, s)) return rc;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_secondary_edge<K, FL>), dim3(launch_blocks(h, list ? std::max(n / 16, 1ll << 16) : n)), dim3(kBlock), lds_bytes(cx, h), s, cx, tv, i0, n,
                           1.f / (float) o->sppse, dimg, WH * 3, h->d_counters, list, list_n);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

// Does the reverse launch of these camera slots with geometry gradients run its value sweep as the traced wavefront with recording stages (render_rev)?
// Then psdr_render_c(PSDR_FLAG_KEEP_RECORDS) can BE that value sweep.  One rule, used by both.
template <int FL> bool rev_value_sweep_is_wavefront(const psdr_scene_s *h, const psdr_render_opts *o, long long n) {
    if constexpr ((FL & kSceneForest) == 0) return false;
    if (o->integrator != PSDR_INTEGRATOR_PATH) return false;
    const int depth = std::min(o->max_depth, kMaxRevDepthDeep);
    const bool has_tree = h->num_nodes > 0 && (h->n_tiny == 0 || h->n_blas > 0);
    const int split_env = h->opt.rev_split;
    const bool split = split_env != 0 && (split_env == 1 || (has_tree && o->max_depth >= 2 && n >= (1ll << 20)));
    return split && depth <= kMaxRevDepth && traced_wavefront(h) && use_wavefront(h, o);
}
inline bool same_camera_samples(const psdr_render_opts &a, const psdr_render_opts &b) {
    return a.integrator == b.integrator && a.max_depth == b.max_depth && a.hide_emitters == b.hide_emitters && a.spp == b.spp && a.spp_begin == b.spp_begin &&
           a.spp_end == b.spp_end && a.rng_offset[0] == b.rng_offset[0] && ((a.flags ^ b.flags) & (PSDR_FLAG_FUSED | PSDR_FLAG_WAVEFRONT)) == 0;
}
inline int rev_record_words(const psdr_scene_s *, int depth, bool wavefront) {
    return kRevDiskHead + (wavefront ? kRevDiskPerVertexCf : kRevDiskPerVertex) * depth;
}

// value_only (psdr_render_c with PSDR_FLAG_KEEP_RECORDS where the traced wavefront does not apply -- a scene without a tree, a small launch): the PathTracer's
// primal render runs as the VALUE KERNEL of a split reverse launch (k_camera_rev STAGE 1: image + one record per path, kept on the handle), and the
// psdr_render_d_rev of the same samples runs its adjoint kernel (STAGE 2) only.  Returns 1 when that does not apply (the caller renders as usual).
template <int FL>
int render_rev_impl(psdr_scene_s *h, const psdr_render_opts *o, const float *adj_img, float *out_img, const psdr_grads *grads, hipStream_t s, bool value_only) {
    const long long WH = (long long) h->desc.width * h->desc.height;
    if (value_only) {
        const long long nv = WH * (o->spp_end - o->spp_begin);
        if (o->integrator != PSDR_INTEGRATOR_PATH || o->max_depth > kMaxRevDepth || o->max_depth < 1 || h->opt.rev_split == 0 || o->spp <= 0 || nv <= 0 ||
            nv > launch_chunk(h, 26) || nv > fit_chunk(std::max(nv, 1ll << 18), (size_t) (kRevDiskHead + kRevDiskPerVertex * o->max_depth) * 4, 64u << 20, h->rev_bytes)) return 1;
    }
    if (out_img) HIP_TRY(hipMemsetAsync(out_img, 0, sizeof(float) * WH * 3, s));
    DeviceSink<FL> sink{}; sink.g = *grads; sink.L = make_sink_layout(h, grads);
    // the instances of scenes without a tree address a triangle's cached row directly (DeviceSink::add_tri): every row cached, slot == triangle
    if ((FL & kSceneTiny) != 0 && PSDR_TINY_DIRECT_ROWS && grads->g_tri_info != nullptr && !(h->hot_identity && sink.L.hot_rows == h->desc.num_tris))
        return fail("psdr_render_d_rev: the gradient cache of a scene without a tree does not hold every triangle row");
    const int nsp = o->spp_end - o->spp_begin;
    if (o->spp > 0 && nsp > 0) {
        LaunchCtx cx;
        if (int rc = make_ctx(h, o, 0, cx)) return rc;
        const long long n = WH * nsp;
        h->slots[0] += (uint64_t) n;
        const int depth = o->integrator == PSDR_INTEGRATOR_PATH ? std::min(o->max_depth, kMaxRevDepthDeep) : 1;
        // up to kMaxRevDepth vertices the per-lane path record lives in LDS; deeper paths keep it in HBM, one column per thread of the grid
        const bool deep_rec = depth > kMaxRevDepth;
        const int rec_bytes = deep_rec ? 0 : depth * kPathRecWords * kBlock * 4;
        float *deep = nullptr;
        if (deep_rec) {
            const size_t need = (size_t) launch_blocks(h, n) * kBlock * depth * kPathRecWords * sizeof(float);
            if (int rc = scratch_reserve(&h->d_rev_deep, &h->rev_deep_bytes, need, s, "deep path records of the reverse launch")) return rc;
            deep = reinterpret_cast<float *>(h->d_rev_deep);
        }
        const bool geo = value_only || grads->g_tri_info != nullptr || grads->g_cam_to_world != nullptr;
        const int wg_per_cu = (o->integrator == PSDR_INTEGRATOR_DIRECT && !(FL & kSceneRough)) ? 3 : 2;       // rev_waves<FL, true, INTEG>
        // Split launch: a scene with a tree to walk, geometry gradients (the register-heavy kernel), hits replayable.
        const int split_env = h->opt.rev_split;
        const bool replayable = o->integrator == PSDR_INTEGRATOR_PATH || (o->integrator == PSDR_INTEGRATOR_DIRECT && o->bsdf_samples <= 1 && o->light_samples <= 1);
        const bool has_tree = h->num_nodes > 0 && (h->n_tiny == 0 || h->n_blas > 0);
        // measured (tools/rev_split_probe.py, 4 M slots): PathTracer(3) cbox_bunny 6.8 -> 4.8 ms, bunny_light 5.6 -> 3.9, 50 k-triangle interior
        // 11.4 -> 9.4, PathTracer(6) 21 -> 12 / 13 -> 7.3 / 36 -> 20; the DirectIntegrator (three rays per slot) only gains on the
        // large tree (3.8 -> 3.3 ms; bunny scenes 2.0 -> 2.4); the 12-triangle box neither way
        const bool worth = o->integrator == PSDR_INTEGRATOR_PATH ? o->max_depth >= 2 : h->num_nodes >= 16384;
        // the records a recording psdr_render_c (value_only, above) left of exactly these samples on exactly these tables: this call is their adjoint kernel
        const bool kept_fused = !value_only && geo && o->integrator == PSDR_INTEGRATOR_PATH && !deep_rec && split_env != 0 && h->kept.valid && h->kept.kind == 0 &&
                                h->kept.gen == h->tables_gen && h->kept.n == n && out_img == nullptr && same_camera_samples(h->kept.o, *o) && n <= launch_chunk(h, 26);
        const bool split = value_only || kept_fused || (geo && replayable && split_env != 0 && (split_env == 1 || (has_tree && worth && n >= (1ll << 20))));
        // adjoint kernel of a split launch: nothing of the tree staged, no stacks -- only what plan_lds places without any room (the hit rows of
        // the kernel-argument primitives and the small tables of the two-level / tiny instances, Tab<FL>::lds_small), then record and cache
        // the PathTracer's geometry-adjoint kernels keep the lane-private emitter accumulators in registers (RegPrivSink): no LDS block, never switched off
        sink.L.priv_regs = (geo && o->integrator == PSDR_INTEGRATOR_PATH && reg_priv_kernel<FL, true, PSDR_INTEGRATOR_PATH>()) ? 1 : 0;
        LaunchCtx cx2 = cx;
        plan_lds(h, cx2, 1 << 30);
        const int base2 = cx2.off_stack;                           // bytes in front of where the stacks would start
        cx2.sc.n_lnodes = cx2.sc.n_lbtris = cx2.sc.n_ltri = 0;
        // lane-private emitter rows (30 KB) only where they do not cost a resident workgroup (C5 PathTracer(3) fused: stacks 22 KB +
        // record 24 KB + cache 19 KB + 30 KB = one workgroup per CU instead of two, 11.5 -> 19.4 ms)
        {
            LaunchCtx probe = cx;
            const int floor_bytes = (split ? base2 : plan_lds(h, probe, 1 << 30)) + rec_bytes;                  // (tables +) stacks only + record
            if (!sink.L.priv_regs && sink.L.priv_rows > 0 && floor_bytes + sink_bytes(sink.L) > h->lds_limit / wg_per_cu) { sink.L.priv_rows = 0; sink.L.priv_emitter = -1; }
        }
        // deferred row adjoints (DeviceSink::defer_row -> sorted adds at the slot's end): one column per path vertex behind the primary one, where the
        // block does not cost a resident workgroup
        sink.L.pend_rows = 0; sink.L.pend_off = 0;
        // (rows: measured on the PathTracer only -- C2 all gradients 5.28 -> 4.99 ms, the C4 shard's adjoint kernel 19.5 -> 18.7; the DirectIntegrator's
        // single deferred row is level to 3 % slower: profiles/r05_rev_sorted_abk.txt)
        if (geo && h->opt.rev_sorted != 0 && grads->g_tri_info != nullptr && o->integrator == PSDR_INTEGRATOR_PATH) {
            LaunchCtx probe = cx;
            const int floor_bytes = (split ? base2 : plan_lds(h, probe, 1 << 30)) + rec_bytes;
            for (int rows = std::min(depth, 4); rows >= 1; --rows) {
                SinkLayout t = sink.L;
                sink_reserve_pending(t, rows);
                if (floor_bytes + sink_bytes(t) <= h->lds_limit / wg_per_cu) { sink.L = t; break; }
            }
        }
        const int cache_bytes = sink_bytes(sink.L);
        plan_lds(h, cx, split ? rec_bytes : rec_bytes + cache_bytes);   // stage less of the scene: the record (+ cache) live in LDS too
        cx.off_pathrec = lds_bytes(cx, h);
        cx.off_sink = cx.off_pathrec + rec_bytes;
        cx2.off_stack = base2; cx2.off_pathrec = base2; cx2.off_sink = base2 + rec_bytes;
        // a scene without a tree runs both kernels of a (forced) split launch on the layout of the fused one
        const bool no_tree = lds_bytes(cx, h) == cx.off_stack;
        if (split && no_tree) cx2 = cx;
        const int dyn1 = cx.off_pathrec + rec_bytes;                   // value kernel of a split launch
        const int dyn_bytes = (split && !no_tree) ? cx2.off_sink + cache_bytes : cx.off_sink + cache_bytes;
        if (std::max(dyn_bytes, split ? dyn1 : 0) > h->lds_limit) return fail("psdr_render_d_rev: the launch needs " + std::to_string(dyn_bytes) + " bytes of LDS per workgroup (path record of " +
                                                  std::to_string(depth) + " levels + traversal stacks + gradient cache), the device offers " + std::to_string(h->lds_limit));
        // material-only gradients (no triangle / camera table wanted) run the variant without the geometric adjoints;
        // the integrator is a compile-time parameter as in the forward kernels (direct: no path record / replay loop)
#define PSDR_LAUNCH_REV_K(GEO, INTEG, STAGE, CX, BYTES, J0, N, IMG, DISK, STRIDE)                                                    \
        do { if ((BYTES) > 48 * 1024) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_camera_rev<FL, GEO, INTEG, STAGE>), hipFuncAttributeMaxDynamicSharedMemorySize, (BYTES))); \
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_camera_rev<FL, GEO, INTEG, STAGE>), dim3(launch_blocks(h, (N))), dim3(kBlock), (BYTES), s, CX, sink, \
                           o->spp, o->spp_begin, nsp, (long long) (J0), (long long) (N), 1.f / (float) o->spp, adj_img, IMG, h->d_counters, DISK, (long long) (STRIDE), deep, disk_cf, ProbeView{nullptr, nullptr}); } while (0)
#define PSDR_LAUNCH_REV(GEO, INTEG) PSDR_LAUNCH_REV_K(GEO, INTEG, 0, cx, dyn_bytes, 0, n, out_img, (float *) nullptr, 0)
        // the value sweep of a split PathTracer launch on a two-level scene runs as the TRACED WAVEFRONT (dense trace kernel between the stages):
        // in the fused value kernel a wave pays its slowest lane's tree walk at every closest_hit (C4 shard: 29 ms of the 57; renderC by the
        // wavefront: 16 ms)
        bool wf_value = false;
        if constexpr ((FL & kSceneForest) != 0) wf_value = split && !value_only && !kept_fused && !no_tree && o->integrator == PSDR_INTEGRATOR_PATH && !deep_rec && traced_wavefront(h) && use_wavefront(h, o);
        // (rev_value_sweep_is_wavefront states the same rule for psdr_render_c's PSDR_FLAG_KEEP_RECORDS; `geo` and the option rev_split == 1 on a scene
        // without a tree are the caller's side of it)
        const int disk_cf = wf_value ? 1 : 0;
        bool direct_probe = false;
        if constexpr ((FL & kSceneForest) != 0) direct_probe = !split && !no_tree && o->integrator == PSDR_INTEGRATOR_DIRECT && probe_direct(h, o, n);
        if (split) {
            // records of one chunk of slots live in a scratch buffer (2 + 5 depth words per slot; 2 + 8 depth from the wavefront); chunks bound its size
            const int words = kRevDiskHead + (wf_value ? kRevDiskPerVertexCf : kRevDiskPerVertex) * depth;
            // (kept records of exactly this launch on the handle: the buffer exists and holds them -- no sizing against the free memory)
            const bool reuse_possible = h->kept.valid && h->kept.gen == h->tables_gen && h->kept.n == n && (size_t) n * words * sizeof(float) <= h->rev_bytes;
            // 2^26 slots per chunk (104 B of records per slot with the wavefront's cf format: 7 GB): the C4 shard's PathTracer(3) reverse as ONE chunk
            // 36.5 -> 35.7 ms of kernel time against four of 2^24 (its value sweep is the traced wavefront; profiles/r04_chunk_sweep.txt)
            const long long chunk = std::min<long long>(n, reuse_possible ? launch_chunk(h, 26) : fit_chunk(launch_chunk(h, 26), (size_t) words * 4 + (wf_value ? 240 : 0), 128u << 20, h->rev_bytes + h->ws_bytes));
            const size_t need = (size_t) chunk * words * sizeof(float);
            if (need > h->rev_bytes) h->kept.valid = false;          // the buffer is about to be replaced
            if (int rc = scratch_reserve(&h->d_rev, &h->rev_bytes, need, s, "per-path records of the split reverse launch")) return rc;
            float *disk = reinterpret_cast<float *>(h->d_rev);
            // the records psdr_render_c(PSDR_FLAG_KEEP_RECORDS) left of exactly these samples on exactly these tables: the value sweep is already there
            const bool reuse = (wf_value ? h->kept.kind == 1 : kept_fused) && h->kept.valid && h->kept.gen == h->tables_gen && h->kept.n == n && chunk >= n && out_img == nullptr &&
                               same_camera_samples(h->kept.o, *o) && need <= h->rev_bytes;
            if (!reuse) h->kept.valid = false;          // this launch overwrites them
            for (long long c0 = 0; c0 < n; c0 += chunk) {
                const long long nc = std::min(chunk, n - c0);
                if (o->integrator == PSDR_INTEGRATOR_PATH) {
                    if (reuse) {
                        // (nothing: the adjoint kernel below reads the kept records)
                    } else if (wf_value) {
                        if constexpr ((FL & kSceneForest) != 0) {
                            const TangentView<0, FL> tv0{};
                            const WfRecArgs ra{disk, chunk, c0, nc};
                            if (int rc = run_camera_wavefront<float, FL>(h, o, tv0, out_img, nullptr, s, &ra)) return rc;
                        }
                    } else
                    PSDR_LAUNCH_REV_K(true, PSDR_INTEGRATOR_PATH, 1, cx, dyn1, c0, nc, out_img, disk, chunk);
                    if (value_only) {
                        HIP_TRY(hipGetLastError());
                        h->kept.valid = true; h->kept.o = *o; h->kept.gen = h->tables_gen; h->kept.n = n; h->kept.kind = 0;
                        return 0;
                    }
                    PSDR_LAUNCH_REV_K(true, PSDR_INTEGRATOR_PATH, 2, cx2, dyn_bytes, c0, nc, (float *) nullptr, disk, chunk);
                } else {
                    PSDR_LAUNCH_REV_K(true, PSDR_INTEGRATOR_DIRECT, 1, cx, dyn1, c0, nc, out_img, disk, chunk);
                    PSDR_LAUNCH_REV_K(true, PSDR_INTEGRATOR_DIRECT, 2, cx2, dyn_bytes, c0, nc, (float *) nullptr, disk, chunk);
                }
            }
            // kept records are ONE-SHOT: the reverse call that consumed them clears them (a caller who rewrites tables in place under an unchanged descriptor and a
            // repeated rng_offset must not meet the records of the old tables: ADVICE r5); the Python surface renders a recording primal before every backward
            h->kept.valid = false;
        } else if (direct_probe) {
            if constexpr ((FL & kSceneForest) != 0) {
                // DirectIntegrator(1, 1), one kernel, on a two-level scene: probe pass -> dense trace kernel -> the reverse kernel on primitives + hit rows
                const TangentView<0, FL> tv0{};
                DeviceSink<FL | kScenePre> sinkp{}; sinkp.g = sink.g; sinkp.L = sink.L;
                LaunchCtx cxf;
                if (int rc = make_ctx(h, o, 0, cxf)) return rc;                                   // the probe pass stages the scene like a forward kernel
                LaunchCtx cxp = cx;                                                              // the final pass: record + cache, nothing of the tree, no stacks
                plan_lds(h, cxp, 1 << 30);
                cxp.sc.n_lnodes = cxp.sc.n_lbtris = cxp.sc.n_ltri = 0;
                cxp.off_pathrec = cxp.off_stack; cxp.off_sink = cxp.off_pathrec + rec_bytes;
                const int dyn_p = cxp.off_sink + cache_bytes;
                const long long chunk = std::min<long long>(n, launch_chunk(h, 25));
                for (long long c0 = 0; c0 < n; c0 += chunk) {
                    const long long nc = std::min(chunk, n - c0);
                    ProbeBuffers pb;
                    const int probe_per_cu = big_launch_per_cu(h, nc);
                    if (int rc = probe_buffers(h, nc, 3, pb, s, probe_per_cu)) return rc;
                    const TraceQueue tq{pb.req, pb.count, pb.sub_cap};
                    const int blocks = (launch_blocks(h, nc, probe_per_cu) + kWfSub - 1) / kWfSub * kWfSub;
                    hipLaunchKernelGGL(k_direct_probe<FL>, dim3(blocks), dim3(kBlock), lds_bytes(cxf, h), s, cxf, tv0, o->spp, o->spp_begin, SlotDiv(nsp), c0, nc, tq, pb.hit, pb.mask,
                                       make_rng_jump(o->rng_offset[0] + 2));
                    HIP_TRY(hipGetLastError());
                    if (int rc = launch_wf_trace(h, pb.req, pb.count, pb.sub_cap, pb.hit, s)) return rc;
#define PSDR_LAUNCH_REV_P(GEO)                                                                                                            \
                    do { if (dyn_p > 48 * 1024) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_camera_rev<FL | kScenePre, GEO, PSDR_INTEGRATOR_DIRECT, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, dyn_p)); \
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_camera_rev<FL | kScenePre, GEO, PSDR_INTEGRATOR_DIRECT, 0>), dim3(launch_blocks(h, nc)), dim3(kBlock), dyn_p, s, cxp, sinkp,  \
                                       o->spp, o->spp_begin, nsp, c0, nc, 1.f / (float) o->spp, adj_img, out_img, h->d_counters, (float *) nullptr, 0ll, deep, 0, ProbeView{pb.hit, pb.mask}); } while (0)
                    if (geo) PSDR_LAUNCH_REV_P(true); else PSDR_LAUNCH_REV_P(false);
#undef PSDR_LAUNCH_REV_P
                    HIP_TRY(hipGetLastError());
                }
            }
        } else
        switch (o->integrator) {
            case PSDR_INTEGRATOR_DIRECT: if (geo) PSDR_LAUNCH_REV(true, PSDR_INTEGRATOR_DIRECT); else PSDR_LAUNCH_REV(false, PSDR_INTEGRATOR_DIRECT); break;
            case PSDR_INTEGRATOR_PATH: if (geo) PSDR_LAUNCH_REV(true, PSDR_INTEGRATOR_PATH); else PSDR_LAUNCH_REV(false, PSDR_INTEGRATOR_PATH); break;
            default: if (geo) PSDR_LAUNCH_REV(true, PSDR_INTEGRATOR_FIELD); else PSDR_LAUNCH_REV(false, PSDR_INTEGRATOR_FIELD); break;
        }
#undef PSDR_LAUNCH_REV_K
#undef PSDR_LAUNCH_REV
        HIP_TRY(hipGetLastError());
    }
    if (o->sppe > 0 && o->sppe_end > o->sppe_begin && h->desc.num_prim_edges > 0 && grads->g_prim_edge) {
        LaunchCtx cx;
        if (int rc = make_ctx(h, o, 1, cx)) return rc;
        const long long i0 = WH * o->sppe_begin, n = WH * (o->sppe_end - o->sppe_begin);
        h->slots[1] += (uint64_t) n;
        // replicated gradient table (see PrimaryEdgeSink): up to 64 copies within 64 MB of the sort scratch's neighbour buffer
        const long long pe_words = (long long) h->desc.num_prim_edges * PSDR_PEDGE_STRIDE;
        int reps = 1;
        while (reps < 64 && (long long) (reps * 2) * pe_words * 4 <= (64ll << 20)) reps *= 2;
        if (n < (1ll << 18)) reps = 1;
        if (reps > 1) {
            const size_t need = (size_t) reps * pe_words * sizeof(float);
            if (int rc = scratch_reserve(&h->d_pe_rep, &h->pe_rep_bytes, need, s, "primary-edge gradient replicas")) return rc;
            HIP_TRY(hipMemsetAsync(h->d_pe_rep, 0, need, s));
        }
        const PrimaryEdgeSink<FL> pe_sink{reps > 1 ? reinterpret_cast<float *>(h->d_pe_rep) : grads->g_prim_edge, pe_words, reps};
        const uint32_t *order = nullptr;
        if (int rc = primary_edge_order(h, cx, i0, n, &order, s)) return rc;
#define PSDR_LAUNCH_PER(INTEG)                                                                                                       \
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_primary_edge_rev<FL, INTEG>), dim3(launch_blocks(h, n, big_launch_per_cu(h, n))), dim3(kBlock), lds_bytes(cx, h), s, cx, pe_sink, i0, n, \
                           1.f / (float) o->sppe, adj_img, h->d_counters, order)
        switch (o->integrator) {
            case PSDR_INTEGRATOR_DIRECT: PSDR_LAUNCH_PER(PSDR_INTEGRATOR_DIRECT); break;
            case PSDR_INTEGRATOR_PATH: PSDR_LAUNCH_PER(PSDR_INTEGRATOR_PATH); break;
            default: PSDR_LAUNCH_PER(PSDR_INTEGRATOR_FIELD); break;
        }
#undef PSDR_LAUNCH_PER
        HIP_TRY(hipGetLastError());
        if (reps > 1) {
            hipLaunchKernelGGL(k_sum_replicas, dim3((unsigned) ((pe_words + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, grads->g_prim_edge,
                               reinterpret_cast<const float *>(h->d_pe_rep), pe_words, reps);
            HIP_TRY(hipGetLastError());
        }
    }
    if (o->sppse > 0 && o->sppse_end > o->sppse_begin && h->desc.num_sec_edges > 0 && o->integrator == PSDR_INTEGRATOR_DIRECT) {
        LaunchCtx cx;
        if (int rc = make_ctx(h, o, 2, cx)) return rc;
        const long long i0 = WH * o->sppse_begin, n = WH * (o->sppse_end - o->sppse_begin);
        h->slots[2] += (uint64_t) n;
        sink.L.priv_rows = 0; sink.L.priv_emitter = -1;            // boundary samples land on the emitter once per slot: no private rows
        sink.L.pend_rows = 0; sink.L.pend_off = 0;                 // (nothing deferred in this kernel)
        const int cache_bytes = sink_bytes(sink.L);
        plan_lds(h, cx, cache_bytes);
        cx.off_sink = lds_bytes(cx, h);
        const int dyn_bytes = cx.off_sink + cache_bytes;
        if (dyn_bytes > 48 * 1024)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_secondary_edge_rev<FL>), hipFuncAttributeMaxDynamicSharedMemorySize, dyn_bytes));
        const uint32_t *list = nullptr; const int *list_n = nullptr;
        {
            LaunchCtx cxf;                                          // the filter stages the scene like a forward kernel (no gradient cache in LDS)
            if (int rc = make_ctx(h, o, 2, cxf)) return rc;
            if (int rc = secondary_edge_filter<FL>(h, cxf, i0, n, &list, &list_n, s)) return rc;
        }
        hipLaunchKernelGGL(k_secondary_edge_rev<FL>, dim3(launch_blocks(h, list ? std::max(n / 16, 1ll << 16) : n)), dim3(kBlock), dyn_bytes, s, cx, sink, i0, n, 1.f / (float) o->sppse,
                           adj_img, h->d_counters, list, list_n);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

template <int FL>
int guide_launch(psdr_scene_s *h, LaunchCtx &cx, const int32_t reso[4], int nrounds, long long n, float *out_mass, hipStream_t s) {
    // two-level scenes: all (stream, round) slots through the filter launches of the secondary-edge term (probe -> dense trace kernel -> filter), the
    // survivors evaluated by k_guide_survivors -- k_guide walks the trees per lane for every one of its nrounds x 2+ rays although a few per cent of the
    // samples get past the first two (the reference's size: 40000 x 5 x 5 cells x 2 streams x 32 rounds = 64 M samples on cbox_bunny)
    if constexpr ((FL & kSceneForest) != 0) {
        if (traced_wavefront(h) && h->opt.probe != 0 && n * (long long) nrounds <= 0x7fffffffLL && n * (long long) nrounds >= (1ll << 18)) {
            const GuideGrid gg{reso[0], reso[1], reso[2], reso[3], (int) n, 1.f / ((float) reso[3] * (float) nrounds)};
            const uint32_t *list = nullptr; const int *list_n = nullptr;
            if (int rc = secondary_edge_filter<FL>(h, cx, 0, n * nrounds, &list, &list_n, s, &gg)) return rc;
            hipLaunchKernelGGL(k_guide_survivors<FL>, dim3(launch_blocks(h, std::max(n * nrounds / 16, 1ll << 16))), dim3(kBlock), lds_bytes(cx, h), s, cx, gg, list, list_n, out_mass, h->d_counters);
            HIP_TRY(hipGetLastError());
            return 0;
        }
    }
    hipLaunchKernelGGL(k_guide<FL>, dim3(launch_blocks(h, n)), dim3(kBlock), lds_bytes(cx, h), s, cx, reso[0], reso[1], reso[2], reso[3], nrounds, n,
                       out_mass, h->d_counters);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <int FL>
int render_rev(psdr_scene_s *h, const psdr_render_opts *o, const float *adj_img, float *out_img, const psdr_grads *grads, hipStream_t s) {
    return render_rev_impl<FL>(h, o, adj_img, out_img, grads, s, false);
}

template <int FL>
int render_c_launch(psdr_scene_s *h, const psdr_render_opts *o, float *out_img, hipStream_t s) {
    const TangentView<0, FL> tv0{};
    if constexpr ((FL & kSceneForest) != 0) {
        // PSDR_FLAG_KEEP_RECORDS: this render IS the value sweep of the reverse launch that follows (render_rev: wf_value) -- recording stages, records kept
        const long long n = (long long) h->desc.width * h->desc.height * (o->spp_end - o->spp_begin);
        // (records + streams of the whole launch must fit the device NOW -- 104 + ~230 bytes per slot; otherwise the plain render below, and the reverse call runs its own,
        // chunked value sweep: no failure for want of memory)
        const int depth_k = std::min(o->max_depth, kMaxRevDepthDeep);
        if ((o->flags & PSDR_FLAG_KEEP_RECORDS) != 0 && h->opt.keep_records != 0 && o->spp > 0 && n > 0 && n <= launch_chunk(h, 26) && rev_value_sweep_is_wavefront<FL>(h, o, n) &&
            n <= fit_chunk(std::max(n, 1ll << 18), (size_t) rev_record_words(h, depth_k, true) * 4 + 240, 128u << 20, h->rev_bytes + h->ws_bytes)) {
            const int depth = depth_k;
            const size_t need = (size_t) n * rev_record_words(h, depth, true) * sizeof(float);
            h->kept.valid = false;
            if (int rc = scratch_reserve(&h->d_rev, &h->rev_bytes, need, s, "per-path records of the split reverse launch")) return rc;
            const WfRecArgs ra{reinterpret_cast<float *>(h->d_rev), n, 0, n};
            if (int rc = run_camera_wavefront<float, FL>(h, o, tv0, out_img, nullptr, s, &ra)) return rc;
            h->slots[0] += (uint64_t) n;
            h->kept.valid = true; h->kept.o = *o; h->kept.gen = h->tables_gen; h->kept.n = n; h->kept.kind = 1;
            return 0;
        }
    }
    if ((o->flags & PSDR_FLAG_KEEP_RECORDS) != 0 && h->opt.keep_records != 0 && !use_wavefront(h, o)) {
        // where the reverse launch that follows would run BOTH sweeps in one kernel (a scene without a tree: the headline's cbox): this render is its value
        // kernel (render_rev value_only); 1 = does not apply
        const psdr_grads none{};
        const int rc = render_rev_impl<FL>(h, o, nullptr, out_img, &none, s, true);
        if (rc != 1) return rc;
    }
    if (use_wavefront(h, o)) return run_camera_wavefront<float, FL>(h, o, tv0, out_img, nullptr, s);
    return run_camera<float, float, FL>(h, o, tv0, out_img, nullptr, s);
}

template <int FL>
int render_fwd_launch(psdr_scene_s *h, const psdr_render_opts *o, int K, const psdr_tangents *tangents, float *img, float *dimg, hipStream_t s) {
    switch (K) {
        case 1: return render_fwd<1, FL>(h, o, tangents, img, dimg, s);
        case 3: return render_fwd<3, FL>(h, o, tangents, img, dimg, s);
        default: return fail("psdr_render_d_fwd: K must be 1 or 3");
    }
}

}  // namespace
