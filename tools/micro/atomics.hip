// micro-benchmark: throughput of global float atomicAdd (device scope) under different address patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k_atomic(float *tab, uint32_t words, int per_thread, int mode) {
    uint32_t x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    for (int i = 0; i < per_thread; ++i) {
        x = x * 1664525u + 1013904223u;
        uint32_t idx;
        if (mode == 0) idx = (x >> 8) % words;                                  // random per lane
        else if (mode == 1) idx = ((blockIdx.x * 4 + (threadIdx.x >> 6)) * 977 + i * 31) % words;  // same address per wave
        else idx = (((x >> 8) % (words / 24)) * 24) + (threadIdx.x % 24);        // random rows, 24 consecutive words
        atomicAdd(tab + idx, 1.0f);
    }
}
int main() {
    float *tab; const uint32_t words = 120000 * 10;   // 4.8 MB
    hipMalloc(&tab, words * 4); hipMemset(tab, 0, words * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int mode = 0; mode < 3; ++mode) for (int rep = 0; rep < 2; ++rep) {
        const int blocks = 4096, per = 64;
        hipEventRecord(a);
        hipLaunchKernelGGL(k_atomic, dim3(blocks), dim3(256), 0, 0, tab, words, per, mode);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        double n = (double) blocks * 256 * per;
        if (rep) printf("mode %d: %.2f ms for %.0fM atomics = %.2f G atomics/s\n", mode, ms, n / 1e6, n / ms / 1e6);
    }
    return 0;
}
