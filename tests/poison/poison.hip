// poison.hip -- TEST infrastructure only (tests/test_rough_rev_order_gpu.py): fills what a kernel must never read before writing it --
// the scratch arena of the queue (spill slots) and the LDS of every CU -- with a chosen bit pattern, so that a result that depends on
// "what ran before in the process" can be pinned on one of them.  Not linked into the product.
#include <hip/hip_runtime.h>
#include <cstdint>

__global__ __launch_bounds__(256) void k_poison_scratch(uint32_t pattern, uint32_t *sink) {
    volatile uint32_t a[1024];                      // 4 KB of scratch per lane: more than any render kernel spills
    for (int i = 0; i < 1024; ++i) a[i] = pattern;
    uint32_t s = 0;
    for (int i = threadIdx.x & 7; i < 1024; i += 97) s ^= a[i];
    if (s == 0x12345u) sink[0] = s;
}
__global__ __launch_bounds__(1024) void k_poison_lds(uint32_t pattern, int words, uint32_t *sink) {
    extern __shared__ uint32_t lds[];
    for (int i = threadIdx.x; i < words; i += 1024) lds[i] = pattern;
    __syncthreads();
    if (lds[(threadIdx.x * 7) % words] == 0x12345u) sink[0] = 1;
}
// also the VGPRs a later wave inherits: a kernel that leaves `pattern` in many registers
__global__ __launch_bounds__(256) void k_poison_vgpr(float p, float *sink) {
    float r[200];
#pragma unroll
    for (int i = 0; i < 200; ++i) r[i] = p + (float) (threadIdx.x & 1) * 0.f;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 200; ++i) s += r[i] * (float) (i + 1);
    if (s == 12345.f) sink[0] = s;
}

extern "C" int poison_gpu(uint32_t pattern, int what) {       // what: bit 0 scratch, bit 1 LDS, bit 2 VGPRs
    static uint32_t *sink = nullptr;
    if (!sink && hipMalloc(&sink, 64) != hipSuccess) return 1;
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 2;
    const int cus = prop.multiProcessorCount;
    if (what & 1) hipLaunchKernelGGL(k_poison_scratch, dim3(cus * 32), dim3(256), 0, nullptr, pattern, sink);
    if (what & 2) {
        const int bytes = 160 * 1024 - 64;
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_poison_lds), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return 3;
        hipLaunchKernelGGL(k_poison_lds, dim3(cus * 4), dim3(1024), bytes, nullptr, pattern, bytes / 4, sink);
    }
    if (what & 4) { float p; __builtin_memcpy(&p, &pattern, 4); hipLaunchKernelGGL(k_poison_vgpr, dim3(cus * 32), dim3(256), 0, nullptr, p, reinterpret_cast<float *>(sink)); }
    if (hipGetLastError() != hipSuccess) return 4;
    return hipDeviceSynchronize() == hipSuccess ? 0 : 5;
}
