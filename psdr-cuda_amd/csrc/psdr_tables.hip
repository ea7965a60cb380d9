// psdr_tables.hip -- the differentiable table chain of Scene::configure as HIP kernels (round 3): world vertices -> TriangleInfo rows
// (process_mesh, reference src/shape/mesh.cpp:20-51), secondary-edge records of every candidate edge (Mesh::configure, mesh.cpp:251-270 +
// the coplanar filter of scene.cpp:219-244) and primary-edge records of every candidate edge of a sensor (perspective.cpp:39-111), each
// with its hand-written adjoint.  One forward and one reverse entry point per table (include/psdr_hip.h psdr_geo_*); the host mirror wraps
// them in torch.autograd.Function objects (psdr_cuda/tables_native.py).  Round 4: the compaction of the kept edges, their length
// distributions, mesh areas and the emitter tables are kernels here as well (k_compact_*, k_mesh_areas, k_emitter_rows): no count or sum travels to the host.  Replaces ~150 eager torch launches per configure() and ~250 in its
// backward.  Everything is fp32; the scatter-adds into vertex / row adjoints are hardware global_atomic_add_f32.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string>

#include "../../include/psdr_hip.h"

namespace psdr_host { int fail(const std::string &m); }

namespace {
constexpr int kB = 256;
constexpr float kEpsilon = 1e-5f, kEdgeEpsilon = 1e-5f;      // include/psdr/constants.h

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 ld3(const float *p) { return {p[0], p[1], p[2]}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// the filters compare a dot product with 1 - 1e-5: evaluated as the eager torch chain does (three products, two sums, no contraction), so
// that an edge whose faces are coplanar to within an ulp is kept or dropped alike
__device__ __forceinline__ float dot_plain(V3 a, V3 b) { return __fadd_rn(__fadd_rn(__fmul_rn(a.x, b.x), __fmul_rn(a.y, b.y)), __fmul_rn(a.z, b.z)); }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ void st3(float *p, V3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
__device__ __forceinline__ void add3(float *p, V3 a) {
    if (a.x != 0.f) atomicAdd(p, a.x);
    if (a.y != 0.f) atomicAdd(p + 1, a.y);
    if (a.z != 0.f) atomicAdd(p + 2, a.z);
}
// y = x / |x|;  a_x = (a_y - y (y . a_y)) / |x|
__device__ __forceinline__ V3 normalize_vjp(V3 y, float len, V3 ay) { const float inv = len > 0.f ? 1.f / len : 0.f; return (ay - y * dot(y, ay)) * inv; }

// ------------------------------------------------------------------ TriangleInfo rows (process_mesh, mesh.cpp:20-51)
// pass 1: per face, the un-normalised normal cross(e1, e2) added to its three vertices (area-weighted vertex normals: mesh.cpp:33-41 sums
// face_normal * area and divides by the summed area before normalising -- the same direction).
// The sums are accumulated in DOUBLE (hardware global_atomic_add_f64) and rounded to fp32 once: a double holds the sum of fp32 terms whose
// exponents span less than 29 bits EXACTLY, so the rounded sum does not depend on the order the atomics land in.  With fp32 atomics the vertex
// normals moved in their last bit from one configure() to the next, and a path tracer turns that into samples that flip at a silhouette: the
// same scene rendered twice differed by 1e-3 in a pixel, the bunny's vertex gradient by 5e-4 rel-L2 (tools/r05_clk.sh, round 5).
__global__ __launch_bounds__(kB) void k_face_accum(int T, const float *__restrict__ v, const int32_t *__restrict__ faces, double *__restrict__ dsum) {
    const int t = blockIdx.x * kB + threadIdx.x;
    if (t >= T) return;
    const int id[3] = {faces[3 * t], faces[3 * t + 1], faces[3 * t + 2]};
    const V3 p0 = ld3(v + 3 * (size_t) id[0]), c = cross(ld3(v + 3 * (size_t) id[1]) - p0, ld3(v + 3 * (size_t) id[2]) - p0);
    for (int k = 0; k < 3; ++k) {
        double *d = dsum + 3 * (size_t) id[k];
        if (c.x != 0.f) atomicAdd(d, (double) c.x);
        if (c.y != 0.f) atomicAdd(d + 1, (double) c.y);
        if (c.z != 0.f) atomicAdd(d + 2, (double) c.z);
    }
}
__global__ __launch_bounds__(kB) void k_vsum_round(int n, const double *__restrict__ dsum, float *__restrict__ vsum) {
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i < n) vsum[i] = (float) dsum[i];
}
// pass 2: the 22-word rows p0 e1 e2 n0 n1 n2 face_normal face_area (types.h:135-146)
__global__ __launch_bounds__(kB) void k_tri_rows(int T, const float *__restrict__ v, const int32_t *__restrict__ faces, const float *__restrict__ vsum,
                                                 float *__restrict__ rows, int stride) {
    const int t = blockIdx.x * kB + threadIdx.x;
    if (t >= T) return;
    const int id[3] = {faces[3 * t], faces[3 * t + 1], faces[3 * t + 2]};
    const V3 p0 = ld3(v + 3 * (size_t) id[0]), e1 = ld3(v + 3 * (size_t) id[1]) - p0, e2 = ld3(v + 3 * (size_t) id[2]) - p0;
    const V3 c = cross(e1, e2);
    const float len = sqrtf(dot(c, c));
    float *r = rows + (size_t) t * stride;
    st3(r, p0); st3(r + 3, e1); st3(r + 6, e2);
    for (int k = 0; k < 3; ++k) {
        const V3 s = ld3(vsum + 3 * (size_t) id[k]);
        st3(r + 9 + 3 * k, s * (1.f / sqrtf(dot(s, s))));
    }
    st3(r + 18, c * (1.f / len));
    r[21] = 0.5f * len;
    for (int k = 22; k < stride; ++k) r[k] = 0.f;          // the padding words of a PSDR_TRI_STRIDE row
}
// adjoint, pass 1: the vertex-normal adjoints of the rows gathered into the adjoint of the per-vertex sums
__global__ __launch_bounds__(kB) void k_tri_rows_rev_vn(int T, const int32_t *__restrict__ faces, const float *__restrict__ vsum, const float *__restrict__ a_rows,
                                                        int stride, float *__restrict__ a_vsum) {
    const int t = blockIdx.x * kB + threadIdx.x;
    if (t >= T) return;
    const float *ar = a_rows + (size_t) t * stride;
    for (int k = 0; k < 3; ++k) {
        const V3 an = ld3(ar + 9 + 3 * k);
        if (an.x == 0.f && an.y == 0.f && an.z == 0.f) continue;
        const int i = faces[3 * t + k];
        const V3 s = ld3(vsum + 3 * (size_t) i);
        const float len = sqrtf(dot(s, s));
        add3(a_vsum + 3 * (size_t) i, normalize_vjp(s * (1.f / len), len, an));
    }
}
// adjoint, pass 2: per face, everything that flows into its three vertices
__global__ __launch_bounds__(kB) void k_tri_rows_rev(int T, const float *__restrict__ v, const int32_t *__restrict__ faces, const float *__restrict__ a_rows, int stride,
                                                     const float *__restrict__ a_vsum, float *__restrict__ a_v) {
    const int t = blockIdx.x * kB + threadIdx.x;
    if (t >= T) return;
    const int i0 = faces[3 * t], i1 = faces[3 * t + 1], i2 = faces[3 * t + 2];
    const V3 p0 = ld3(v + 3 * (size_t) i0), e1 = ld3(v + 3 * (size_t) i1) - p0, e2 = ld3(v + 3 * (size_t) i2) - p0;
    const V3 c = cross(e1, e2);
    const float len = sqrtf(dot(c, c));
    const float *ar = a_rows + (size_t) t * stride;
    // adjoint of c = cross(e1, e2): from the three vertex sums, the face normal c / |c| and the area |c| / 2
    V3 ac = ld3(a_vsum + 3 * (size_t) i0) + ld3(a_vsum + 3 * (size_t) i1) + ld3(a_vsum + 3 * (size_t) i2);
    const V3 fn = c * (1.f / len);
    ac = ac + normalize_vjp(fn, len, ld3(ar + 18)) + fn * (0.5f * ar[21]);
    // c = e1 x e2:  a_e1 = e2 x a_c,  a_e2 = a_c x e1
    const V3 ae1 = ld3(ar + 3) + cross(e2, ac), ae2 = ld3(ar + 6) + cross(ac, e1);
    add3(a_v + 3 * (size_t) i1, ae1); add3(a_v + 3 * (size_t) i2, ae2);
    add3(a_v + 3 * (size_t) i0, ld3(ar) - ae1 - ae2);
}

// -------------------------------------------------- secondary-edge records (mesh.cpp:251-270, scene.cpp:219-244)
// edges[e] = v0, v1, face0, face1 (-1: boundary), opposite vertex of face0 (global ids).  info = p0 e1 n0 n1 p2 is_boundary; keep = the
// two face normals are not parallel (dot < 1 - EdgeEpsilon; a boundary edge has n1 = 0 and always stays)
__global__ __launch_bounds__(kB) void k_sec_edges(int E, const int32_t *__restrict__ edges, const float *__restrict__ v, const float *__restrict__ rows, int stride,
                                                  float *__restrict__ info, uint8_t *__restrict__ keep) {
    const int e = blockIdx.x * kB + threadIdx.x;
    if (e >= E) return;
    const int32_t *ed = edges + 5 * (size_t) e;
    const bool bnd = ed[3] < 0;
    const V3 p0 = ld3(v + 3 * (size_t) ed[0]), e1 = ld3(v + 3 * (size_t) ed[1]) - p0, p2 = ld3(v + 3 * (size_t) ed[4]);
    const V3 n0 = ld3(rows + (size_t) ed[2] * stride + 18), n1 = bnd ? V3{0.f, 0.f, 0.f} : ld3(rows + (size_t) ed[3] * stride + 18);
    float *o = info + 16 * (size_t) e;
    st3(o, p0); st3(o + 3, e1); st3(o + 6, n0); st3(o + 9, n1); st3(o + 12, p2); o[15] = bnd ? 1.f : 0.f;
    keep[e] = dot_plain(n0, n1) < 1.f - kEdgeEpsilon ? 1 : 0;
}
__global__ __launch_bounds__(kB) void k_sec_edges_rev(int E, const int32_t *__restrict__ edges, const float *__restrict__ a_info, float *__restrict__ a_v,
                                                      float *__restrict__ a_rows, int stride) {
    const int e = blockIdx.x * kB + threadIdx.x;
    if (e >= E) return;
    const int32_t *ed = edges + 5 * (size_t) e;
    const float *a = a_info + 16 * (size_t) e;
    const V3 ae1 = ld3(a + 3);
    add3(a_v + 3 * (size_t) ed[0], ld3(a) - ae1); add3(a_v + 3 * (size_t) ed[1], ae1); add3(a_v + 3 * (size_t) ed[4], ld3(a + 12));
    add3(a_rows + (size_t) ed[2] * stride + 18, ld3(a + 6));
    if (ed[3] >= 0) add3(a_rows + (size_t) ed[3] * stride + 18, ld3(a + 9));
}

// ------------------------------------------------------- primary-edge records of one sensor (perspective.cpp:39-111)
// cam = world_to_sample (16, row-major) | camera position (3) | viewing direction (3).  Kept: silhouette edges as seen from the camera
// (face-normal meshes: not both faces turned away and not coplanar; smooth meshes: exactly one face turned towards the camera; boundary
// edges always).  rows8 = film positions of the end points (differentiable), edge normal and length on the film (detached), 0;
// z4 = 1 / depth of the end points along the viewing direction + the adjacent faces as int bits (PSDR_PRIMARY_EDGE_VIS_CHECK).
__device__ __forceinline__ void project(const float *m, V3 p, float &qx, float &qy, float &iw) {
    const float hx = m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], hy = m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7];
    const float w = m[12] * p.x + m[13] * p.y + m[14] * p.z + m[15];
    iw = 1.f / w; qx = hx * iw; qy = hy * iw;
}
__global__ __launch_bounds__(kB) void k_prim_edges(int E, const int32_t *__restrict__ edges, const uint8_t *__restrict__ face_normals, const float *__restrict__ v,
                                                   const float *__restrict__ rows, int stride, const float *__restrict__ cam, float *__restrict__ rows8,
                                                   float *__restrict__ z4, uint8_t *__restrict__ keep) {
    const int e = blockIdx.x * kB + threadIdx.x;
    if (e >= E) return;
    const int32_t *ed = edges + 5 * (size_t) e;
    const bool valid = ed[3] >= 0;
    const V3 cp = ld3(cam + 16), cd = ld3(cam + 19);
    auto unit = [](V3 a) { return a * (1.f / sqrtf(dot(a, a))); };
    const float *r0 = rows + (size_t) ed[2] * stride, *r1 = rows + (size_t) (valid ? ed[3] : 0) * stride;
    const V3 e0 = unit(cp - ld3(r0)), e1 = unit(valid ? cp - ld3(r1) : cp);          // masked gather: normalize(cam_pos - 0) on a boundary edge
    const V3 n0 = ld3(r0 + 18), n1 = valid ? ld3(r1 + 18) : V3{0.f, 0.f, 0.f};
    const float d0 = dot_plain(e0, n0), d1 = dot_plain(e1, n1), dn = dot_plain(n0, n1);
    const bool keep_face = !(valid && ((d0 < kEpsilon && d1 < kEpsilon) || dn > 1.f - kEpsilon));
    const bool keep_smooth = !valid || ((d0 > kEpsilon) != (d1 > kEpsilon));
    keep[e] = (face_normals[e] ? keep_face : keep_smooth) ? 1 : 0;
    const V3 p0 = ld3(v + 3 * (size_t) ed[0]), p1 = ld3(v + 3 * (size_t) ed[1]);
    float q0x, q0y, q1x, q1y, iw;
    project(cam, p0, q0x, q0y, iw); project(cam, p1, q1x, q1y, iw);
    float ex = q1x - q0x, ey = q1y - q0y;
    const float ln = sqrtf(ex * ex + ey * ey);
    ex /= ln; ey /= ln;
    float *o = rows8 + 8 * (size_t) e;
    o[0] = q0x; o[1] = q0y; o[2] = q1x; o[3] = q1y; o[4] = -ey; o[5] = ex; o[6] = ln; o[7] = 0.f;
    const V3 cdu = unit(cd);
    float *z = z4 + 4 * (size_t) e;
    z[0] = 1.f / dot(p0 - cp, cdu); z[1] = 1.f / dot(p1 - cp, cdu);
    z[2] = __int_as_float(ed[2]); z[3] = __int_as_float(valid ? ed[3] : ed[2]);
}
// adjoint of q = (M p).xy / (M p).w with respect to p and M (rows 0, 1, 3 of world_to_sample)
__device__ __forceinline__ void project_vjp(const float *m, V3 p, float aqx, float aqy, V3 &ap, float *am /* [16] lane-private */) {
    float qx, qy, iw;
    project(m, p, qx, qy, iw);
    const float ahx = aqx * iw, ahy = aqy * iw, aw = -(aqx * qx + aqy * qy) * iw;
    ap = V3{ahx * m[0] + ahy * m[4] + aw * m[12], ahx * m[1] + ahy * m[5] + aw * m[13], ahx * m[2] + ahy * m[6] + aw * m[14]};
    am[0] += ahx * p.x; am[1] += ahx * p.y; am[2] += ahx * p.z; am[3] += ahx;
    am[4] += ahy * p.x; am[5] += ahy * p.y; am[6] += ahy * p.z; am[7] += ahy;
    am[12] += aw * p.x; am[13] += aw * p.y; am[14] += aw * p.z; am[15] += aw;
}
__global__ __launch_bounds__(kB) void k_prim_edges_rev(int E, const int32_t *__restrict__ edges, const float *__restrict__ v, const float *__restrict__ cam,
                                                       const float *__restrict__ a_rows8, float *__restrict__ a_v, float *__restrict__ a_w2s) {
    const int e = blockIdx.x * kB + threadIdx.x;
    float am[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) am[i] = 0.f;
    if (e < E) {
        const int32_t *ed = edges + 5 * (size_t) e;
        const float *a = a_rows8 + 8 * (size_t) e;
        if (a[0] != 0.f || a[1] != 0.f || a[2] != 0.f || a[3] != 0.f) {
            V3 ap;
            project_vjp(cam, ld3(v + 3 * (size_t) ed[0]), a[0], a[1], ap, am); add3(a_v + 3 * (size_t) ed[0], ap);
            project_vjp(cam, ld3(v + 3 * (size_t) ed[1]), a[2], a[3], ap, am); add3(a_v + 3 * (size_t) ed[1], ap);
        }
    }
    // the 12 matrix words: wave sum, one atomic per wave and word
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (i >= 8 && i < 12) continue;
        float s = am[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if ((threadIdx.x & 63) == 0 && s != 0.f) atomicAdd(a_w2s + i, s);
    }
}

// ------------------------------------------------------------------------------ world positions
// transform_pos (include/psdr/core/transform.h:84-88) of every vertex by ITS mesh's to_world matrix: y = (A v + t) / (p . v + q).  Products and
// sums as the eager torch chain forms them (three products, its reduction order, then the translation; no contraction): an edge whose
// faces are coplanar to an ulp must see the same vertices either way.
__global__ __launch_bounds__(kB) void k_world_vertices(int V, const float *__restrict__ v_raw, const int32_t *__restrict__ vmesh, const float *__restrict__ mats,
                                                       float *__restrict__ out) {
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i >= V) return;
    const float *m = mats + 16 * (size_t) vmesh[i];
    const V3 x = ld3(v_raw + 3 * (size_t) i);
    float h[4];
    {
#pragma clang fp contract(off)
        for (int r = 0; r < 4; ++r) {
            const float s02 = m[4 * r] * x.x + m[4 * r + 2] * x.z;           // torch's reduction of three addends on this device: (a + c) + b
            const float s012 = s02 + m[4 * r + 1] * x.y;
            h[r] = s012 + m[4 * r + 3];
        }
    }
    // an affine matrix (w = 1 exactly: every scene file of the reference) divides exactly; a projective one by this unit's approximate division
    const float w = h[3];
    st3(out + 3 * (size_t) i, w == 1.f ? V3{h[0], h[1], h[2]} : V3{h[0] / w, h[1] / w, h[2] / w});
}
// a_v = (A^T a - p (y . a)) / w
__global__ __launch_bounds__(kB) void k_world_vertices_rev(int V, const float *__restrict__ v_raw, const int32_t *__restrict__ vmesh, const float *__restrict__ mats,
                                                           const float *__restrict__ y, const float *__restrict__ a_y, float *__restrict__ a_raw) {
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i >= V) return;
    const float *m = mats + 16 * (size_t) vmesh[i];
    const V3 x = ld3(v_raw + 3 * (size_t) i), yy = ld3(y + 3 * (size_t) i), a = ld3(a_y + 3 * (size_t) i);
    const float w = m[12] * x.x + m[13] * x.y + m[14] * x.z + m[15], iw = 1.f / w, ya = dot(yy, a);
    st3(a_raw + 3 * (size_t) i, V3{(m[0] * a.x + m[4] * a.y + m[8] * a.z - m[12] * ya) * iw, (m[1] * a.x + m[5] * a.y + m[9] * a.z - m[13] * ya) * iw,
                                   (m[2] * a.x + m[6] * a.y + m[10] * a.z - m[14] * ya) * iw});
}

// ------------------------------------------------------------- kept edges -> table + length distribution, count on the device
// The reference compresses the kept edges and builds a DiscreteDistribution over their lengths inside one Enoki trace (scene.cpp:219-244,
// perspective.cpp:96-111, pmf.cpp:7-21); an eager host chain has to read the NUMBER of kept edges back to size the compacted table.  Here the
// table keeps its capacity E: the kept rows come first in their original order, the tail is zero, and the distribution is normalised on the
// device (pmf / sum, cmf / sum, so the host passes sum = 1 without knowing it).  cmf[i] = 1 for i >= n - 1: the binary search of
// sample_reuse (lower bound of u < 1) then never leaves the kept rows, exactly as over a table of n rows.  header = {n as int bits, sum}.
constexpr int kS = 1024;
struct ScanTmp { int wi[kS / 64]; float wf[kS / 64]; };
// exclusive scan of flag, inclusive scan of w over the kS threads of the block (+ the block totals)
__device__ __forceinline__ void block_scan(int flag, float w, int &excl, float &incl, int &btotal, float &bsum, ScanTmp &t) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int ci = flag; float cf = w;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int ui = __shfl_up(ci, off, 64); const float uf = __shfl_up(cf, off, 64);
        if (lane >= off) { ci += ui; cf += uf; }
    }
    __syncthreads();                       // (t may still be read from an earlier call)
    if (lane == 63) { t.wi[wave] = ci; t.wf[wave] = cf; }
    __syncthreads();
    int bi = 0; float bf = 0.f;
    btotal = 0; bsum = 0.f;
#pragma unroll
    for (int k = 0; k < kS / 64; ++k) {
        if (k == wave) { bi = btotal; bf = bsum; }
        btotal += t.wi[k]; bsum += t.wf[k];
    }
    excl = bi + ci - flag; incl = bf + cf;
}
__device__ __forceinline__ float edge_weight(const float *row, int w0, int wn) {
    if (wn == 1) return row[w0];
    return sqrtf(row[w0] * row[w0] + row[w0 + 1] * row[w0 + 1] + row[w0 + 2] * row[w0 + 2]);
}
__global__ __launch_bounds__(kS) void k_compact_count(int E, const uint8_t *__restrict__ keep, const float *__restrict__ rows, int S, int w0, int wn,
                                                      int *__restrict__ bc, float *__restrict__ bs) {
    __shared__ ScanTmp t;
    const int e = blockIdx.x * kS + threadIdx.x;
    const int flag = e < E && keep[e] ? 1 : 0;
    const float w = flag ? edge_weight(rows + (size_t) e * S, w0, wn) : 0.f;
    int excl, btotal; float incl, bsum;
    block_scan(flag, w, excl, incl, btotal, bsum, t);
    if (threadIdx.x == 0) { bc[blockIdx.x] = btotal; bs[blockIdx.x] = bsum; }
}
__global__ __launch_bounds__(kS) void k_compact_top(int nb, const int *__restrict__ bc, const float *__restrict__ bs, int *__restrict__ bo, float *__restrict__ fo,
                                                    float *__restrict__ header) {
    __shared__ ScanTmp t;
    int ci = 0; float cf = 0.f;
    for (int base = 0; base < nb; base += kS) {
        const int b = base + threadIdx.x;
        const int c = b < nb ? bc[b] : 0; const float f = b < nb ? bs[b] : 0.f;
        int excl, btotal; float incl, bsum;
        block_scan(c, f, excl, incl, btotal, bsum, t);
        if (b < nb) { bo[b] = ci + excl; fo[b] = cf + (incl - f); }
        ci += btotal; cf += bsum;
    }
    if (threadIdx.x == 0) { header[0] = __int_as_float(ci); header[1] = cf; }
}
__global__ __launch_bounds__(kS) void k_compact_write(int E, const uint8_t *__restrict__ keep, const float *__restrict__ rows, int S, int w0, int wn,
                                                      const uint32_t *__restrict__ aux, int aux_stride, int A, const int *__restrict__ bo, const float *__restrict__ fo,
                                                      const float *__restrict__ header, float *__restrict__ rows_out, uint32_t *__restrict__ aux_out,
                                                      int32_t *__restrict__ pos, float *__restrict__ pmf, float *__restrict__ cmf) {
    __shared__ ScanTmp t;
    const int e = blockIdx.x * kS + threadIdx.x;
    const int flag = e < E && keep[e] ? 1 : 0;
    const float *row = rows + (size_t) e * S;
    const float w = flag ? edge_weight(row, w0, wn) : 0.f;
    int excl, btotal; float incl, bsum;
    block_scan(flag, w, excl, incl, btotal, bsum, t);
    if (e >= E) return;
    const int n = __float_as_int(header[0]);
    const float total = header[1], inv = total > 0.f ? 1.f / total : 0.f;
    if (flag) {
        const int p = bo[blockIdx.x] + excl;
        for (int c = 0; c < S; ++c) rows_out[(size_t) p * S + c] = row[c];
        for (int c = 0; c < A; ++c) aux_out[(size_t) p * A + c] = aux[(size_t) e * aux_stride + c];
        pmf[p] = w * inv;
        cmf[p] = p >= n - 1 ? 1.f : (fo[blockIdx.x] + incl) * inv;
        pos[e] = p;
    } else pos[e] = -1;
    if (e >= n) {                          // slot e of the tail
        for (int c = 0; c < S; ++c) rows_out[(size_t) e * S + c] = 0.f;
        for (int c = 0; c < A; ++c) aux_out[(size_t) e * A + c] = 0u;
        pmf[e] = 0.f; cmf[e] = 1.f;
    }
}
__global__ __launch_bounds__(kB) void k_compact_rev(int E, int S, const int32_t *__restrict__ pos, const float *__restrict__ a_out, float *__restrict__ a_rows) {
    const long long i = (long long) blockIdx.x * kB + threadIdx.x;
    if (i >= (long long) E * S) return;
    const int e = (int) (i / S), c = (int) (i - (long long) e * S), p = pos[e];
    a_rows[i] = p >= 0 ? a_out[(size_t) p * S + c] : 0.f;
}

// ------------------------------------------------------------------ mesh areas + emitter tables (scene.cpp:183-196, area.cpp:10-16)
// One workgroup per mesh: its total area (Mesh::m_total_area, mesh.cpp:244-246) and, for the mesh of an area light, the face-area
// distribution (mesh.cpp:248-249: pmf = areas, cmf = their inclusive prefix sum, at emitter_i[.][3] of the concatenated tables).
__global__ __launch_bounds__(kS) void k_mesh_areas(const float *__restrict__ rows, int stride, const int32_t *__restrict__ face_offset, const int32_t *__restrict__ mesh_emitter,
                                                   const int32_t *__restrict__ emitter_i, const float *__restrict__ env_weight, float *__restrict__ mesh_area,
                                                   float *__restrict__ face_pmf, float *__restrict__ face_cmf) {
    __shared__ ScanTmp t;
    const int m = blockIdx.x, f0 = face_offset[m], f1 = face_offset[m + 1], em = mesh_emitter[m];
    const bool distrb = em >= 0 && env_weight[em] < 0.f;
    const int off = distrb ? emitter_i[4 * em + 3] : 0;
    float carry = 0.f;
    for (int base = f0; base < f1; base += kS) {
        const int f = base + threadIdx.x;
        const float a = f < f1 ? rows[(size_t) f * stride + 21] : 0.f;
        int excl, btotal; float incl, bsum;
        block_scan(0, a, excl, incl, btotal, bsum, t);
        if (distrb && f < f1) { face_pmf[off + f - f0] = a; face_cmf[off + f - f0] = carry + incl; }
        carry += bsum;
    }
    if (threadIdx.x == 0) mesh_area[m] = carry;
}
// Scene::m_emitters_distrb (scene.cpp:183-196): sampling weight of an area light = area * luminance(radiance) (area.cpp:10-16), of the
// environment map the host-made env_weight; normalised here (pmf / sum, running cmf), so the host passes emitter_sum = 1.
__global__ void k_emitter_rows(int Ne, const int32_t *__restrict__ emitter_i, const float *__restrict__ radiance, const float *__restrict__ env_weight,
                               const float *__restrict__ mesh_area, const float *__restrict__ face_cmf, float *__restrict__ emitter_f, float *__restrict__ emitter_pmf,
                               float *__restrict__ emitter_cmf) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float total = 0.f;
    for (int e = 0; e < Ne; ++e) {
        const float *r = radiance + 3 * e;
        const float w = env_weight[e] >= 0.f ? env_weight[e] : mesh_area[emitter_i[4 * e]] * (r[0] * .2126f + r[1] * .7152f + r[2] * .0722f);
        emitter_pmf[e] = w; total += w;
    }
    const float inv = 1.f / total;
    float run = 0.f;
    for (int e = 0; e < Ne; ++e) {
        const bool env = env_weight[e] >= 0.f;
        const float w = emitter_pmf[e] * inv, area = mesh_area[emitter_i[4 * e]];
        run += w;
        emitter_pmf[e] = w; emitter_cmf[e] = run;
        float *o = emitter_f + PSDR_EMITTER_F_STRIDE * e;
        o[0] = env ? 0.f : radiance[3 * e]; o[1] = env ? 0.f : radiance[3 * e + 1]; o[2] = env ? 0.f : radiance[3 * e + 2];
        o[3] = w; o[4] = env ? 0.f : 1.f / area; o[5] = env ? 0.f : face_cmf[emitter_i[4 * e + 3] + emitter_i[4 * e + 2] - 1];
        o[6] = 0.f; o[7] = 0.f;
    }
}

inline dim3 grid(int n) { return dim3((unsigned) ((n + kB - 1) / kB)); }
}  // namespace

#define TAB_TRY(expr)                                                                              \
    do { hipError_t e_ = (expr); if (e_ != hipSuccess) return psdr_host::fail(std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)

extern "C" {

int psdr_geo_world_vertices_fwd(int32_t V, const float *v_raw, const int32_t *vmesh, const float *mats, float *v_world, void *stream) {
    if (V <= 0 || !v_raw || !vmesh || !mats || !v_world) return psdr_host::fail("psdr_geo_world_vertices_fwd: invalid argument");
    hipLaunchKernelGGL(k_world_vertices, grid(V), dim3(kB), 0, (hipStream_t) stream, V, v_raw, vmesh, mats, v_world);
    TAB_TRY(hipGetLastError());
    return 0;
}
int psdr_geo_world_vertices_rev(int32_t V, const float *v_raw, const int32_t *vmesh, const float *mats, const float *v_world, const float *a_world, float *a_raw,
                                void *stream) {
    if (V <= 0 || !v_raw || !vmesh || !mats || !v_world || !a_world || !a_raw) return psdr_host::fail("psdr_geo_world_vertices_rev: invalid argument");
    hipLaunchKernelGGL(k_world_vertices_rev, grid(V), dim3(kB), 0, (hipStream_t) stream, V, v_raw, vmesh, mats, v_world, a_world, a_raw);
    TAB_TRY(hipGetLastError());
    return 0;
}
int psdr_geo_tri_rows_fwd(int32_t V, int32_t T, const float *v, const int32_t *faces, float *vsum, float *rows, int32_t row_stride, void *stream) {
    if (V <= 0 || T <= 0 || !v || !faces || !vsum || !rows || row_stride < 22) return psdr_host::fail("psdr_geo_tri_rows_fwd: invalid argument");
    hipStream_t s = (hipStream_t) stream;
    // the double accumulators live in the row table until the rows are written (24 V <= 72 T bytes of its >= 88 T; a mesh with more unused
    // vertices than that takes a stream-ordered scratch)
    const size_t acc_bytes = sizeof(double) * 3 * (size_t) V;
    double *dsum = reinterpret_cast<double *>(rows);
    const bool own = acc_bytes > sizeof(float) * (size_t) row_stride * (size_t) T || (reinterpret_cast<uintptr_t>(rows) & 7u) != 0;
    if (own) TAB_TRY(hipMallocAsync(reinterpret_cast<void **>(&dsum), acc_bytes, s));
    TAB_TRY(hipMemsetAsync(dsum, 0, acc_bytes, s));
    hipLaunchKernelGGL(k_face_accum, grid(T), dim3(kB), 0, s, T, v, faces, dsum);
    hipLaunchKernelGGL(k_vsum_round, grid(3 * V), dim3(kB), 0, s, 3 * V, dsum, vsum);
    if (own) TAB_TRY(hipFreeAsync(dsum, s));
    hipLaunchKernelGGL(k_tri_rows, grid(T), dim3(kB), 0, s, T, v, faces, vsum, rows, row_stride);
    TAB_TRY(hipGetLastError());
    return 0;
}
int psdr_geo_tri_rows_rev(int32_t V, int32_t T, const float *v, const int32_t *faces, const float *vsum, const float *a_rows, int32_t row_stride,
                          float *a_vsum, float *a_v, void *stream) {
    if (V <= 0 || T <= 0 || !v || !faces || !vsum || !a_rows || !a_vsum || !a_v || row_stride < 22) return psdr_host::fail("psdr_geo_tri_rows_rev: invalid argument");
    hipStream_t s = (hipStream_t) stream;
    TAB_TRY(hipMemsetAsync(a_vsum, 0, sizeof(float) * 3 * (size_t) V, s));
    hipLaunchKernelGGL(k_tri_rows_rev_vn, grid(T), dim3(kB), 0, s, T, faces, vsum, a_rows, row_stride, a_vsum);
    hipLaunchKernelGGL(k_tri_rows_rev, grid(T), dim3(kB), 0, s, T, v, faces, a_rows, row_stride, a_vsum, a_v);
    TAB_TRY(hipGetLastError());
    return 0;
}
int psdr_geo_sec_edges_fwd(int32_t E, const int32_t *edges, const float *v, const float *rows, int32_t row_stride, float *info, uint8_t *keep, void *stream) {
    if (E <= 0 || !edges || !v || !rows || !info || !keep) return psdr_host::fail("psdr_geo_sec_edges_fwd: invalid argument");
    hipLaunchKernelGGL(k_sec_edges, grid(E), dim3(kB), 0, (hipStream_t) stream, E, edges, v, rows, row_stride, info, keep);
    TAB_TRY(hipGetLastError());
    return 0;
}
int psdr_geo_sec_edges_rev(int32_t E, const int32_t *edges, const float *a_info, float *a_v, float *a_rows, int32_t row_stride, void *stream) {
    if (E <= 0 || !edges || !a_info || !a_v || !a_rows) return psdr_host::fail("psdr_geo_sec_edges_rev: invalid argument");
    hipLaunchKernelGGL(k_sec_edges_rev, grid(E), dim3(kB), 0, (hipStream_t) stream, E, edges, a_info, a_v, a_rows, row_stride);
    TAB_TRY(hipGetLastError());
    return 0;
}
int psdr_geo_prim_edges_fwd(int32_t E, const int32_t *edges, const uint8_t *face_normals, const float *v, const float *rows, int32_t row_stride,
                            const float *cam22, float *rows8, float *z4, uint8_t *keep, void *stream) {
    if (E <= 0 || !edges || !face_normals || !v || !rows || !cam22 || !rows8 || !z4 || !keep) return psdr_host::fail("psdr_geo_prim_edges_fwd: invalid argument");
    hipLaunchKernelGGL(k_prim_edges, grid(E), dim3(kB), 0, (hipStream_t) stream, E, edges, face_normals, v, rows, row_stride, cam22, rows8, z4, keep);
    TAB_TRY(hipGetLastError());
    return 0;
}
int psdr_geo_prim_edges_rev(int32_t E, const int32_t *edges, const float *v, const float *cam22, const float *a_rows8, float *a_v, float *a_w2s, void *stream) {
    if (E <= 0 || !edges || !v || !cam22 || !a_rows8 || !a_v || !a_w2s) return psdr_host::fail("psdr_geo_prim_edges_rev: invalid argument");
    hipLaunchKernelGGL(k_prim_edges_rev, grid(E), dim3(kB), 0, (hipStream_t) stream, E, edges, v, cam22, a_rows8, a_v, a_w2s);
    TAB_TRY(hipGetLastError());
    return 0;
}

int psdr_geo_compact_edges_fwd(int32_t E, const float *rows, int32_t S, const uint8_t *keep, int32_t w0, int32_t wn, const void *aux, int32_t aux_stride, int32_t A,
                               void *scratch, float *rows_out, void *aux_out, int32_t *pos, float *pmf, float *cmf, float *header, void *stream) {
    if (E <= 0 || !rows || S <= 0 || !keep || w0 < 0 || (wn != 1 && wn != 3) || w0 + wn > S || (A > 0 && (!aux || !aux_out || aux_stride < A)) || !scratch || !rows_out || !pos ||
        !pmf || !cmf || !header)
        return psdr_host::fail("psdr_geo_compact_edges_fwd: invalid argument");
    hipStream_t s = (hipStream_t) stream;
    const int nb = (E + kS - 1) / kS;
    int *bc = (int *) scratch, *bo = bc + nb;
    float *bs = (float *) (bo + nb), *fo = bs + nb;
    hipLaunchKernelGGL(k_compact_count, dim3(nb), dim3(kS), 0, s, E, keep, rows, S, w0, wn, bc, bs);
    hipLaunchKernelGGL(k_compact_top, dim3(1), dim3(kS), 0, s, nb, bc, bs, bo, fo, header);
    hipLaunchKernelGGL(k_compact_write, dim3(nb), dim3(kS), 0, s, E, keep, rows, S, w0, wn, (const uint32_t *) aux, aux_stride, A, bo, fo, header, rows_out,
                       (uint32_t *) aux_out, pos, pmf, cmf);
    TAB_TRY(hipGetLastError());
    return 0;
}
int psdr_geo_compact_edges_rev(int32_t E, int32_t S, const int32_t *pos, const float *a_rows_out, float *a_rows, void *stream) {
    if (E <= 0 || S <= 0 || !pos || !a_rows_out || !a_rows) return psdr_host::fail("psdr_geo_compact_edges_rev: invalid argument");
    const long long n = (long long) E * S;
    hipLaunchKernelGGL(k_compact_rev, dim3((unsigned) ((n + kB - 1) / kB)), dim3(kB), 0, (hipStream_t) stream, E, S, pos, a_rows_out, a_rows);
    TAB_TRY(hipGetLastError());
    return 0;
}

int psdr_geo_emitter_tables(int32_t M, const float *rows, int32_t row_stride, const int32_t *face_offset, const int32_t *mesh_emitter, int32_t Ne, const int32_t *emitter_i,
                            const float *radiance, const float *env_weight, float *mesh_area, float *emitter_f, float *emitter_pmf, float *emitter_cmf, float *face_pmf,
                            float *face_cmf, void *stream) {
    if (M <= 0 || !rows || row_stride < 22 || !face_offset || !mesh_emitter || !mesh_area || Ne < 0 ||
        (Ne > 0 && (!emitter_i || !radiance || !env_weight || !emitter_f || !emitter_pmf || !emitter_cmf || !face_pmf || !face_cmf)))
        return psdr_host::fail("psdr_geo_emitter_tables: invalid argument");
    hipStream_t s = (hipStream_t) stream;
    hipLaunchKernelGGL(k_mesh_areas, dim3(M), dim3(kS), 0, s, rows, row_stride, face_offset, mesh_emitter, emitter_i, env_weight, mesh_area, face_pmf, face_cmf);
    if (Ne > 0) hipLaunchKernelGGL(k_emitter_rows, dim3(1), dim3(64), 0, s, Ne, emitter_i, radiance, env_weight, mesh_area, face_cmf, emitter_f, emitter_pmf, emitter_cmf);
    TAB_TRY(hipGetLastError());
    return 0;
}

}  // extern "C"
