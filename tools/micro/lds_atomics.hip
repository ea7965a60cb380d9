// micro-benchmark: LDS atomic throughput on gfx950 -- ds_add_f32 against ds_add_u32, distinct / shared / few-row address
// patterns, full waves against a single active lane.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/micro/lds_atomics.hip -o /tmp/lds_atomics && /tmp/lds_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <class T, int MODE>
__global__ __launch_bounds__(256) void k_lds(T *out, int iters) {
    __shared__ T tab[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) tab[i] = T(0);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    uint32_t x = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    for (int i = 0; i < iters; ++i) {
        x = x * 1664525u + 1013904223u;
        int idx;
        if (MODE == 0) idx = threadIdx.x + ((i & 7) << 8);                  // all lanes distinct, conflict-free banks
        else if (MODE == 1) idx = (i * 7) & 4095;                           // every lane of the workgroup on ONE word
        else if (MODE == 2) idx = (((x >> 10) % 12) * 24) + (i % 21);       // 12 rows, same word of a random row (the gradient sink)
        else idx = (i * 7) & 4095;                                          // MODE 3: one word, but only lane 0 of each wave adds
        if (MODE != 3 || lane == 0) atomicAdd(&tab[idx], T(1));
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = tab[0] + tab[5];
}
template <class T, int MODE> void run(const char *name) {
    T *out; hipMalloc(&out, 4096 * sizeof(T));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256 * 8, iters = 2048;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_lds<T, MODE>), dim3(blocks), dim3(256), 0, 0, out, iters);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double wave_instr = (double) blocks * 4 * iters;
        // 256 CUs, 8 workgroups each -> per CU 8 * 4 * iters wave-instructions; LDS cycles per wave-instruction if the LDS were the only limit
        if (rep) printf("%-34s %7.3f ms  %6.1f G wave-instr/s  ~%5.1f LDS cycles per wave-instruction per CU\n", name, ms, wave_instr / ms / 1e6,
                        ms * 1e-3 * 2.4e9 / (8.0 * 4 * iters));
    }
    hipFree(out);
}
int main() {
    run<float, 0>("f32 distinct addresses");
    run<float, 1>("f32 one word, 64 lanes");
    run<float, 2>("f32 12 rows (gradient-sink pattern)");
    run<float, 3>("f32 one word, 1 lane per wave");
    run<unsigned, 0>("u32 distinct addresses");
    run<unsigned, 1>("u32 one word, 64 lanes");
    run<unsigned, 2>("u32 12 rows");
    run<unsigned, 3>("u32 one word, 1 lane per wave");
    return 0;
}
