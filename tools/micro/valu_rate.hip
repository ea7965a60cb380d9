// micro-benchmark: VALU issue rate of gfx950 -- wave64 v_fma_f32 wave-instructions per cycle per SIMD at 1, 2, 4 and 8 resident
// waves per SIMD, with 16 independent accumulators per lane (no dependent-issue stalls) and with ONE (a dependent chain).
// Answers "how many cycles does a wave64 VALU instruction occupy its SIMD": 2 (SIMD-32, MI355X_MICROARCH.md) or 4 (SIMD-16, GCN).
// Also: v_pk_fma_f32 (two fp32 FMAs per lane per instruction) and a half-wave (lanes 0..31 active) stream.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define FMA(a) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(x), "v"(y))
#define PKFMA(a) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(x2), "v"(y2))

#define MOV(a) asm volatile("v_mov_b32 %0, %1" : "=v"(a) : "v"(x))
#define XOR(a) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a) : "v"(x))
// mode 0: 16 independent accumulators; 1: one dependent chain; 2: packed fp32 FMA, 8 independent pairs; 3: as 0 with lanes 32..63 off;
// 4: v_xor_b32 x16 independent (integer, two operands); 5: v_mov_b32 x16
__global__ __launch_bounds__(256) void k_valu(float *out, int iters, int mode, unsigned long long *cycles) {
    float x = 1.0f + 1e-7f * threadIdx.x, y = 1e-9f;
    float a0 = 0, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7, a8 = 8, a9 = 9, a10 = 10, a11 = 11, a12 = 12, a13 = 13, a14 = 14, a15 = 15;
    typedef float float2_ __attribute__((ext_vector_type(2)));
    float2_ x2 = {x, x}, y2 = {y, y}, p0 = {0, 1}, p1 = {2, 3}, p2 = {4, 5}, p3 = {6, 7}, p4 = {8, 9}, p5 = {10, 11}, p6 = {12, 13}, p7 = {14, 15};
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    if (mode == 0 || (mode == 3 && (threadIdx.x & 63) < 32)) {
        for (int i = 0; i < iters; ++i) {
            FMA(a0); FMA(a1); FMA(a2); FMA(a3); FMA(a4); FMA(a5); FMA(a6); FMA(a7); FMA(a8); FMA(a9); FMA(a10); FMA(a11); FMA(a12); FMA(a13); FMA(a14); FMA(a15);
        }
    } else if (mode == 1) {
        for (int i = 0; i < iters; ++i) {
            FMA(a0); FMA(a0); FMA(a0); FMA(a0); FMA(a0); FMA(a0); FMA(a0); FMA(a0); FMA(a0); FMA(a0); FMA(a0); FMA(a0); FMA(a0); FMA(a0); FMA(a0); FMA(a0);
        }
    } else if (mode == 2) {
        for (int i = 0; i < iters; ++i) {
            PKFMA(p0); PKFMA(p1); PKFMA(p2); PKFMA(p3); PKFMA(p4); PKFMA(p5); PKFMA(p6); PKFMA(p7); PKFMA(p0); PKFMA(p1); PKFMA(p2); PKFMA(p3); PKFMA(p4); PKFMA(p5); PKFMA(p6); PKFMA(p7);
        }
    } else if (mode == 4) {
        for (int i = 0; i < iters; ++i) {
            XOR(a0); XOR(a1); XOR(a2); XOR(a3); XOR(a4); XOR(a5); XOR(a6); XOR(a7); XOR(a8); XOR(a9); XOR(a10); XOR(a11); XOR(a12); XOR(a13); XOR(a14); XOR(a15);
        }
    } else if (mode == 5) {
        for (int i = 0; i < iters; ++i) {
            MOV(a0); MOV(a1); MOV(a2); MOV(a3); MOV(a4); MOV(a5); MOV(a6); MOV(a7); MOV(a8); MOV(a9); MOV(a10); MOV(a11); MOV(a12); MOV(a13); MOV(a14); MOV(a15);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + a8 + a9 + a10 + a11 + a12 + a13 + a14 + a15 + p0.x + p1.y + p2.x + p3.y + p4.x +
                                                 p5.y + p6.x + p7.y;
    // the LAST workgroup's first wave: it runs while the chip is fully loaded (wave 0 of block 0 is the oldest wave of its SIMD and finishes early)
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x - 1) { cycles[0] = t1 - t0; cycles[1] = r1 - r0; }
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount, simds = cus * 4;
    printf("device: %s, %d CUs, clockRate %.0f MHz\n", prop.name, cus, prop.clockRate / 1e3);
    float *out; hipMalloc(&out, (size_t) cus * 8 * 256 * 4 * 4);
    unsigned long long *cyc; hipMalloc(&cyc, 16);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const char *names[6] = {"v_fma_f32 x16 independent", "v_fma_f32 dependent chain", "v_pk_fma_f32 x8 independent", "v_fma_f32 x16, lanes 32..63 inactive",
                            "v_xor_b32 x16 independent", "v_mov_b32 x16"};
    const int iters = 400000;
    // warm-up: ~2 s of full-chip FMA so that the clocks are where a sustained render kernel runs them (a cold GPU measured 30 % low)
    for (int i = 0; i < 40; ++i) hipLaunchKernelGGL(k_valu, dim3(cus * 8), dim3(256), 0, 0, out, iters, 0, cyc);
    hipDeviceSynchronize();
    printf("%-40s %6s %10s %12s %22s %30s %22s\n", "stream", "w/SIMD", "ms", "G winst/s", "winst/cyc/SIMD@2.4GHz", "shader MHz (s_memtime/s_memrealtime)", "winst/cyc/SIMD@measured");
    for (int mode = 0; mode < 6; ++mode)
        for (int w = 1; w <= 8; w *= 2) {
            // 256-thread workgroups = 4 waves = one per SIMD of a CU; w workgroups per CU -> w waves per SIMD
            const int blocks = cus * w;
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(a);
                hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(256), 0, 0, out, iters, mode, cyc);
                hipEventRecord(b); hipEventSynchronize(b);
                hipEventElapsedTime(&ms, a, b);
            }
            unsigned long long c[2] = {0, 0}; hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
            const double winst = (double) blocks * 4 * iters * 16;
            const double rate = winst / (ms * 1e-3);
            // s_memtime counts shader-clock cycles, s_memrealtime a constant 100 MHz: their ratio over one late wave = the clock the loaded chip runs at
            const double mhz = c[1] ? (double) c[0] / ((double) c[1] / 100.0) : 0.0;
            printf("%-40s %6d %10.3f %12.1f %22.3f %30.0f %22.3f\n", names[mode], w, ms, rate / 1e9, rate / 2.4e9 / simds, mhz, mhz > 0 ? rate / (mhz * 1e6) / simds : 0.0);
        }
    printf("peak if a wave64 VALU op takes 2 cycles: %.1f G winst/s; if 4 cycles: %.1f\n", simds * 2.4 / 2, simds * 2.4 / 4);
    return 0;
}
