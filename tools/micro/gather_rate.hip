// micro-benchmark: the rate at which a CU fetches RANDOM rows of a table that fits the L2 (what the BVH walk does: every lane loads its
// own node each step).  Rows of 64 / 32 / 16 bytes from a 3 MB table (49 152 nodes of 64 B: the tree of a ~50 k-triangle scene), global
// memory (vector L1 -> L2) against the same rows in LDS.
//   mode 0  one lane = one 64-byte row, 4 x global_load_dwordx4        (today's BvhNode fetch)
//   mode 1  one lane = one 32-byte row, 2 x global_load_dwordx4
//   mode 2  one lane = one 16-byte row, 1 x global_load_dwordx4
//   mode 3  FOUR lanes fetch one 64-byte row together (lane & 3 = which 16 bytes): one global_load_dwordx4 serves 16 rows
//   mode 4  one lane = one 64-byte row from LDS (4 x ds_read_b128), 32 KB of rows per workgroup
//   mode 5  one lane = one 128-byte row, 8 x global_load_dwordx4        (a 4-wide node with full-precision boxes)
// Reported: rows per second over the whole chip, and cycles per wave-level row fetch per CU.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/gather_rate.hip -o tools/micro/bin/gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ __launch_bounds__(256) void k_gather(const float4 *__restrict__ tab, uint32_t rows, int iters, int mode, float *out) {
    __shared__ float4 lds[2048];      // 32 KB
    uint32_t x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    if (mode == 4) { for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = tab[i]; __syncthreads(); }
    float acc = 0.f;
    for (int i = 0; i < iters; ++i) {
        x = x * 1664525u + 1013904223u;
        if (mode == 0) {
            const float4 *p = tab + (size_t) ((x >> 8) % rows) * 4;
            const float4 a = p[0], b = p[1], c = p[2], d = p[3];
            acc += a.x + b.y + c.z + d.w;
            x ^= __float_as_uint(acc) & 0xff;              // the next address depends on the data (as the child index of a node does)
        } else if (mode == 1) {
            const float4 *p = tab + (size_t) ((x >> 8) % (rows * 2)) * 2;
            const float4 a = p[0], b = p[1];
            acc += a.x + b.y;
            x ^= __float_as_uint(acc) & 0xff;
        } else if (mode == 2) {
            const float4 a = tab[(x >> 8) % (rows * 4)];
            acc += a.x;
            x ^= __float_as_uint(acc) & 0xff;
        } else if (mode == 3) {
            // the row index of lane group g = lane >> 2 comes from that group's first lane
            const uint32_t xr = __shfl(x, (threadIdx.x & 63) & ~3, 64);
            const float4 a = tab[(size_t) ((xr >> 8) % rows) * 4 + (threadIdx.x & 3)];
            acc += a.x;
            x ^= __float_as_uint(acc) & 0xff;
        } else if (mode == 4) {
            const float4 *p = lds + ((x >> 8) % 512u) * 4;
            const float4 a = p[0], b = p[1], c = p[2], d = p[3];
            acc += a.x + b.y + c.z + d.w;
            x ^= __float_as_uint(acc) & 0xff;
        } else {
            const float4 *p = tab + (size_t) ((x >> 8) % (rows / 2)) * 8;
            const float4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4], f = p[5], g = p[6], h = p[7];
            acc += a.x + b.y + c.z + d.w + e.x + f.y + g.z + h.w;
            x ^= __float_as_uint(acc) & 0xff;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
    hipDeviceProp_t prop; (void) hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const uint32_t rows = 49152;
    std::vector<float> h((size_t) rows * 16);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float) (i % 977) * 1e-3f;
    float4 *tab; (void) hipMalloc(&tab, h.size() * 4); (void) hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    float *out; (void) hipMalloc(&out, (size_t) cus * 8 * 256 * 4);
    hipEvent_t a, b; (void) hipEventCreate(&a); (void) hipEventCreate(&b);
    const char *names[6] = {"64 B row / lane, 4 x dwordx4 (global)", "32 B row / lane, 2 x dwordx4 (global)", "16 B row / lane, 1 x dwordx4 (global)",
                            "64 B row / 4 lanes, 1 x dwordx4 (global)", "64 B row / lane, 4 x ds_read_b128 (LDS)", "128 B row / lane, 8 x dwordx4 (global)"};
    const int iters = 4000;
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_gather, dim3(cus * 8), dim3(256), 0, 0, tab, rows, iters, 0, out);
    (void) hipDeviceSynchronize();
    printf("%-44s %6s %9s %14s %28s\n", "fetch", "w/SIMD", "ms", "G rows/s", "cycles per wave row-fetch per CU");
    for (int mode = 0; mode < 6; ++mode)
        for (int w = 2; w <= 8; w *= 2) {
            const int blocks = cus * w;
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                (void) hipEventRecord(a);
                hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(256), 0, 0, tab, rows, iters, mode, out);
                (void) hipEventRecord(b); (void) hipEventSynchronize(b);
                (void) hipEventElapsedTime(&ms, a, b);
            }
            const double lanes_rows = (double) blocks * 256 * iters / (mode == 3 ? 4 : 1);
            const double wave_fetches_per_cu = (double) w * 4 * iters;          // wave-level fetch steps each CU executed
            printf("%-44s %6d %9.3f %14.2f %28.1f\n", names[mode], w, ms, lanes_rows / (ms * 1e-3) / 1e9, ms * 1e-3 * 2.4e9 / wave_fetches_per_cu);
        }
    return 0;
}
