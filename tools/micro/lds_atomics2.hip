// micro-benchmark 2: what decides the cost of ds_add_f32 on gfx950 -- distinct addresses per instruction or active lanes?
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/micro/lds_atomics2.hip -o /tmp/lds_atomics2 && /tmp/lds_atomics2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
// MODE < 100: every lane active, address = (lane % MODE) * 24 + word  (MODE distinct addresses per instruction)
// MODE >= 100: D = MODE - 100 groups; the wave issues D instructions, each with the lanes of one group active on ONE address
template <int MODE>
__global__ __launch_bounds__(256) void k_lds(float *out, int iters) {
    __shared__ float tab[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) tab[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    uint32_t x = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    for (int i = 0; i < iters; ++i) {
        x = x * 1664525u + 1013904223u;
        const int word = i % 21;
        if (MODE < 100) {
            const int row = MODE == 64 ? lane : (int) ((x >> 10) % (uint32_t) MODE);
            atomicAdd(&tab[row * 24 + word], 1.f);
        } else {
            constexpr int D = MODE - 100;
            const int row = (int) ((x >> 10) % (uint32_t) D);
#pragma unroll
            for (int r = 0; r < D; ++r)
                if (row == r) atomicAdd(&tab[r * 24 + word], 1.f);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = tab[0] + tab[5];
}
template <int MODE> void run(const char *name, int per_iter) {
    float *out; hipMalloc(&out, 4096 * sizeof(float));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256 * 8, iters = 2048;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_lds<MODE>), dim3(blocks), dim3(256), 0, 0, out, iters);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (rep) printf("%-44s %7.3f ms  ~%6.1f LDS cycles per ITERATION per CU (%d instr)\n", name, ms, ms * 1e-3 * 2.4e9 / (8.0 * 4 * iters), per_iter);
    }
    hipFree(out);
}
int main() {
    run<1>("all lanes, 1 address", 1);
    run<2>("all lanes, 2 random addresses", 1);
    run<4>("all lanes, 4 random addresses", 1);
    run<12>("all lanes, 12 random addresses", 1);
    run<64>("all lanes, 64 distinct addresses", 1);
    run<102>("2 instr, each uniform address (exec = group)", 2);
    run<104>("4 instr, each uniform address", 4);
    run<112>("12 instr, each uniform address", 12);
    return 0;
}
