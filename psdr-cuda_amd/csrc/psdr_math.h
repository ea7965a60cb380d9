// psdr_math.h -- scalar/vector/dual-number arithmetic, RNG and warps for the gfx950 kernels.
//
// Everything here is PSDR_HD (host+device) so that tests can drive the very same estimator code
// on the host (csrc/host_check.cpp); the product only ever runs it inside HIP kernels.
//
// AD model: the reference differentiates with Enoki DiffArray (include/psdr/types.h:17-20).  Here
// forward mode is a dual number carried in registers: Dual<K> = value + K tangents, i.e. K
// directional derivatives per render pass.  detach() == enoki::detach.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#define PSDR_HD __host__ __device__ __forceinline__

namespace psdr {

// include/psdr/constants.h:8-28
constexpr float kEpsilon = 1e-5f, kRayEpsilon = 1e-3f, kShadowEpsilon = 1e-3f, kEdgeEpsilon = 1e-5f;
constexpr float kPi = 3.14159265358979323846f, kInvPi = 0.31830988618379067154f;

// ----------------------------------------------------------------------------- Dual<K>
template <int K> struct Dual {
    float v;
    float d[K];
    PSDR_HD Dual() {}
    PSDR_HD Dual(float x) : v(x) {
#pragma unroll
        for (int i = 0; i < K; ++i) d[i] = 0.f;
    }
};

template <class R> struct ad_traits { static constexpr int K = 0; };
template <int K_> struct ad_traits<Dual<K_>> { static constexpr int K = K_; };
template <class R> constexpr bool is_ad() { return ad_traits<R>::K > 0; }

PSDR_HD float val(float x) { return x; }
template <int K> PSDR_HD float val(const Dual<K> &x) { return x.v; }
PSDR_HD float detach(float x) { return x; }
template <int K> PSDR_HD Dual<K> detach(const Dual<K> &x) { return Dual<K>(x.v); }
PSDR_HD float tangent(float, int) { return 0.f; }
template <int K> PSDR_HD float tangent(const Dual<K> &x, int i) { return x.d[i]; }

#define PSDR_DUAL_OP(op, VEXPR, DEXPR)                                                        \
    template <int K> PSDR_HD Dual<K> operator op(const Dual<K> &a, const Dual<K> &b) {        \
        Dual<K> r; r.v = VEXPR; const float av = a.v, bv = b.v; (void) av; (void) bv;         \
        _Pragma("unroll") for (int i = 0; i < K; ++i) { const float ad = a.d[i], bd = b.d[i]; (void) ad; (void) bd; r.d[i] = DEXPR; } \
        return r;                                                                             \
    }
PSDR_DUAL_OP(+, a.v + b.v, ad + bd)
PSDR_DUAL_OP(-, a.v - b.v, ad - bd)
PSDR_DUAL_OP(*, a.v * b.v, ad * bv + av * bd)
#undef PSDR_DUAL_OP
template <int K> PSDR_HD Dual<K> operator/(const Dual<K> &a, const Dual<K> &b) {
    Dual<K> r;
    const float inv = 1.f / b.v;
    r.v = a.v * inv;
#pragma unroll
    for (int i = 0; i < K; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
    return r;
}
template <int K> PSDR_HD Dual<K> operator+(const Dual<K> &a, float b) { Dual<K> r = a; r.v += b; return r; }
template <int K> PSDR_HD Dual<K> operator+(float b, const Dual<K> &a) { Dual<K> r = a; r.v += b; return r; }
template <int K> PSDR_HD Dual<K> operator-(const Dual<K> &a, float b) { Dual<K> r = a; r.v -= b; return r; }
template <int K> PSDR_HD Dual<K> operator-(float b, const Dual<K> &a) {
    Dual<K> r; r.v = b - a.v;
#pragma unroll
    for (int i = 0; i < K; ++i) r.d[i] = -a.d[i];
    return r;
}
template <int K> PSDR_HD Dual<K> operator*(const Dual<K> &a, float b) {
    Dual<K> r; r.v = a.v * b;
#pragma unroll
    for (int i = 0; i < K; ++i) r.d[i] = a.d[i] * b;
    return r;
}
template <int K> PSDR_HD Dual<K> operator*(float b, const Dual<K> &a) { return a * b; }
template <int K> PSDR_HD Dual<K> operator/(const Dual<K> &a, float b) { return a * (1.f / b); }
template <int K> PSDR_HD Dual<K> operator/(float a, const Dual<K> &b) {
    Dual<K> r; const float inv = 1.f / b.v; r.v = a * inv;
#pragma unroll
    for (int i = 0; i < K; ++i) r.d[i] = -r.v * b.d[i] * inv;
    return r;
}
template <int K> PSDR_HD Dual<K> operator-(const Dual<K> &a) { return 0.f - a; }

PSDR_HD float rsqrt_(float x) { return 1.f / sqrtf(x); }
PSDR_HD float sqrt_(float x) { return sqrtf(x); }
template <int K> PSDR_HD Dual<K> sqrt_(const Dual<K> &a) {
    Dual<K> r; r.v = sqrtf(a.v);
    const float h = r.v > 0.f ? 0.5f / r.v : 0.f;
#pragma unroll
    for (int i = 0; i < K; ++i) r.d[i] = a.d[i] * h;
    return r;
}
PSDR_HD float abs_(float x) { return fabsf(x); }
template <int K> PSDR_HD Dual<K> abs_(const Dual<K> &a) { return a.v < 0.f ? -a : a; }
PSDR_HD float sqr(float x) { return x * x; }
template <int K> PSDR_HD Dual<K> sqr(const Dual<K> &a) { return a * a; }
// enoki::max / min / safe_sqrt / clamp (value decides, derivative follows the selected branch)
template <class R> PSDR_HD R max_(const R &a, float b) { return val(a) > b ? a : R(b); }
template <class R> PSDR_HD R min_(const R &a, float b) { return val(a) < b ? a : R(b); }
template <class R> PSDR_HD R safe_sqrt(const R &a) { return sqrt_(max_(a, 0.f)); }
template <class R> PSDR_HD R clamp_(const R &a, float lo, float hi) { return max_(min_(a, hi), lo); }
// atan2 / safe_acos with their derivatives (EnvironmentMap::eval_direction, src/emitter/envmap.cpp:52)
PSDR_HD float atan2_(float y, float x) { return atan2f(y, x); }
template <int K> PSDR_HD Dual<K> atan2_(const Dual<K> &y, const Dual<K> &x) {
    Dual<K> r; r.v = atan2f(y.v, x.v);
    const float inv = 1.f / (x.v * x.v + y.v * y.v);
#pragma unroll
    for (int k = 0; k < K; ++k) r.d[k] = (x.v * y.d[k] - y.v * x.d[k]) * inv;
    return r;
}
PSDR_HD float safe_acos_(float x) { return acosf(fminf(fmaxf(x, -1.f), 1.f)); }
template <int K> PSDR_HD Dual<K> safe_acos_(const Dual<K> &x) {
    const float c = fminf(fmaxf(x.v, -1.f), 1.f);
    Dual<K> r; r.v = acosf(c);
    const bool inside = x.v > -1.f && x.v < 1.f;          // the clamp has zero derivative
    const float g = inside ? -1.f / sqrtf(1.f - c * c) : 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) r.d[k] = g * x.d[k];
    return r;
}
PSDR_HD bool finite_(float x) { return isfinite(x); }

// ------------------------------------------------------------------------------ vectors
template <class R> struct Vec3 {
    R x, y, z;
    PSDR_HD Vec3() {}
    PSDR_HD Vec3(const R &a, const R &b, const R &c) : x(a), y(b), z(c) {}
    PSDR_HD explicit Vec3(float a) : x(a), y(a), z(a) {}
};
using Vec3f = Vec3<float>;
template <class R> PSDR_HD Vec3<R> operator+(const Vec3<R> &a, const Vec3<R> &b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <class R> PSDR_HD Vec3<R> operator-(const Vec3<R> &a, const Vec3<R> &b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <class R> PSDR_HD Vec3<R> operator*(const Vec3<R> &a, const Vec3<R> &b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
template <class R> PSDR_HD Vec3<R> operator*(const Vec3<R> &a, const R &s) { return {a.x * s, a.y * s, a.z * s}; }
template <class R> PSDR_HD Vec3<R> operator/(const Vec3<R> &a, const R &s) { return {a.x / s, a.y / s, a.z / s}; }
template <class R> PSDR_HD Vec3<R> operator-(const Vec3<R> &a) { return {-a.x, -a.y, -a.z}; }
template <int K> PSDR_HD Vec3<Dual<K>> operator*(const Vec3<Dual<K>> &a, float s) { return {a.x * s, a.y * s, a.z * s}; }
template <class R> PSDR_HD R dot(const Vec3<R> &a, const Vec3<R> &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <class R> PSDR_HD Vec3<R> cross(const Vec3<R> &a, const Vec3<R> &b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
template <class R> PSDR_HD R norm(const Vec3<R> &a) { return sqrt_(dot(a, a)); }
template <class R> PSDR_HD Vec3<R> normalize(const Vec3<R> &a) { return a / norm(a); }
PSDR_HD Vec3f val(const Vec3f &a) { return a; }
template <int K> PSDR_HD Vec3f val(const Vec3<Dual<K>> &a) { return {a.x.v, a.y.v, a.z.v}; }
template <class R> PSDR_HD Vec3<R> detach(const Vec3<R> &a) { return {detach(a.x), detach(a.y), detach(a.z)}; }
template <class R> PSDR_HD Vec3<R> lift(const Vec3f &a) { return {R(a.x), R(a.y), R(a.z)}; }
template <class R> PSDR_HD Vec3<R> zero3() { return {R(0.f), R(0.f), R(0.f)}; }
// p0 + e1*s + e2*t  (include/psdr/utils.h:48-57 `bilinear`)
template <class R> PSDR_HD Vec3<R> bary_point(const Vec3<R> &p0, const Vec3<R> &e1, const Vec3<R> &e2, const R &s, const R &t) {
    return e1 * s + (e2 * t + p0);
}

// -------------------------------------------------------------------------------- frame
// Duff et al. orthonormal basis: include/psdr/core/frame.h:9-28
template <class R> struct Frame {
    Vec3<R> s, t, n;
    PSDR_HD Frame() {}
    PSDR_HD explicit Frame(const Vec3<R> &v) : n(v) {
        const float sg = copysignf(1.f, val(v.z));
        const R a = -1.f / (sg + v.z);
        const R b = v.x * v.y * a;
        s = Vec3<R>(sqr(v.x) * a * sg + 1.f, b * sg, v.x * (-sg));
        t = Vec3<R>(b, sg + sqr(v.y) * a, -v.y);
    }
    PSDR_HD Vec3<R> to_local(const Vec3<R> &v) const { return {dot(v, s), dot(v, t), dot(v, n)}; }
    PSDR_HD Vec3<R> to_world(const Vec3<R> &v) const { return s * v.x + t * v.y + n * v.z; }
};

// --------------------------------------------------------------------------------- warp
// include/psdr/core/warp.h:13-48 (Shirley-Chiu concentric map).  The inputs are random numbers:
// they never carry derivatives, so this stays in plain float.
// sin / cos of an angle in [-pi, pi] produced from random numbers: on the device the hardware v_sin_f32 /
// v_cos_f32 (argument in revolutions, ~1e-6 absolute error) instead of the ~60-instruction range-reduced
// library sincosf -- the class of approximation Enoki's CUDA backend uses.
PSDR_HD void sincos_fast(float x, float &s, float &c) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float r = x * 0.15915494309189535f;
    s = __builtin_amdgcn_sinf(r); c = __builtin_amdgcn_cosf(r);
#else
    sincosf(x, &s, &c);
#endif
}
PSDR_HD void concentric_disk(float sx, float sy, float &dx, float &dy) {
    const float x = 2.f * sx - 1.f, y = 2.f * sy - 1.f;
    const bool q13 = fabsf(x) < fabsf(y);
    const float r = q13 ? y : x, rp = q13 ? x : y;
    float phi = 0.25f * kPi * rp / r;
    if (q13) phi = 0.5f * kPi - phi;
    if (x == 0.f && y == 0.f) phi = 0.f;
    float s, c;
    sincos_fast(phi, s, c);
    dx = r * c; dy = r * s;
}
// warp.h:52-61
PSDR_HD Vec3f cosine_hemisphere(float sx, float sy) {
    float px, py;
    concentric_disk(sx, sy, px, py);
    return {px, py, sqrtf(fmaxf(1.f - (px * px + py * py), 0.f))};
}

// ---------------------------------------------------------------------------------- RNG
// Sampler (src/core/sampler.cpp:7-54): one PCG32 stream per sample slot, seeded by two TEA mixes.
// Streams are STATELESS here: the state after `offset` earlier draws is recomputed with the PCG
// jump-ahead (acc_mult, acc_plus_unit computed once on the host), so no 16 B/slot state array
// has to live in HBM between render calls (the reference keeps one, scene.cpp:65-79).
PSDR_HD uint64_t tea64(uint64_t v0, uint64_t v1) {
    uint32_t sum = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        sum += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cull) ^ (v1 + (uint64_t) sum) ^ ((v1 >> 5) + 0xc8013ea4ull);
        v1 += ((v0 << 4) + 0xad90777dull) ^ (v0 + (uint64_t) sum) ^ ((v0 >> 5) + 0x7e95761eull);
    }
    return v0 + (v1 << 32);
}
struct RngJump { uint64_t mult, plus_unit; };   // state' = mult*state + inc*plus_unit
struct Rng {
    uint64_t state, inc;
    static constexpr uint64_t MULT = 0x5851f42d4c957f2dull;
    PSDR_HD uint32_t next_u32() {
        const uint64_t old = state;
        state = old * MULT + inc;
        const uint32_t xs = (uint32_t) (((old >> 18u) ^ old) >> 27u);
        const uint32_t rot = (uint32_t) (old >> 59u);
        return (xs >> rot) | (xs << ((~rot + 1u) & 31));
    }
    PSDR_HD float next() {
        const uint32_t bits = (next_u32() >> 9) | 0x3f800000u;
        return __uint_as_float_hd(bits) - 1.f;
    }
    PSDR_HD static float __uint_as_float_hd(uint32_t b) {
        union { uint32_t u; float f; } c; c.u = b; return c.f;
    }
    PSDR_HD void init(uint64_t slot, const RngJump &j) {
        const uint64_t sv = slot + 0x853c49e6748fea9bull;      // PCG32_DEFAULT_STATE, sampler.h:35
        const uint64_t initstate = tea64(sv, slot), initseq = tea64(slot, sv);
        state = 0; inc = (initseq << 1u) | 1u;
        next_u32(); state += initstate; next_u32();
        state = j.mult * state + inc * j.plus_unit;
    }
};
// (mult, plus_unit) for a jump of `delta` draws (pcg32 advance with inc = 1), on the device too (log2(delta) steps)
PSDR_HD RngJump rng_jump_hd(uint64_t delta) {
    uint64_t cur_mult = Rng::MULT, cur_plus = 1, acc_mult = 1, acc_plus = 0;
    while (delta > 0) {
        if (delta & 1) { acc_mult *= cur_mult; acc_plus = acc_plus * cur_mult + cur_plus; }
        cur_plus = (cur_mult + 1) * cur_plus; cur_mult *= cur_mult; delta >>= 1;
    }
    return {acc_mult, acc_plus};
}
// host: (mult, plus_unit) for a jump of `delta` draws (pcg32 advance with inc = 1)
inline RngJump make_rng_jump(uint64_t delta) {
    uint64_t cur_mult = Rng::MULT, cur_plus = 1, acc_mult = 1, acc_plus = 0;
    while (delta > 0) {
        if (delta & 1) { acc_mult *= cur_mult; acc_plus = acc_plus * cur_mult + cur_plus; }
        cur_plus = (cur_mult + 1) * cur_plus; cur_mult *= cur_mult; delta >>= 1;
    }
    return {acc_mult, acc_plus};
}

// DiscreteDistribution::sample_reuse, src/core/pmf.cpp:30-50 + enoki::binary_search.
// cmf/pmf tables are tiny and hot: they sit in L2 (and in the scalar cache when uniform).
PSDR_HD int sample_reuse(const float *__restrict__ cmf, const float *__restrict__ pmf, float sum, int size, float &u,
                         float &pmf_norm) {
    if (size == 1) { pmf_norm = 1.f; return 0; }
    u *= sum;
    int lo = 0, hi = size - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cmf[mid] < u) lo = (mid + 1 < hi ? mid + 1 : hi); else hi = mid;
    }
    if (lo > 0) u -= cmf[lo - 1];
    const float p = pmf[lo];
    if (p > 0.f) u /= p;
    u = fminf(fmaxf(u, 0.f), 1.f);
    pmf_norm = p / sum;
    return lo;
}

}  // namespace psdr
