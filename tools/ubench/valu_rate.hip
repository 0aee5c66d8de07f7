// Developer micro-benchmark: VALU throughput per SIMD for f32 instruction forms at 1/2/4 waves per SIMD.
// Long kernels (tens of ms) after a warm-up so the clock has settled; cycles are derived from the measured
// shader clock (s_memtime / s_memrealtime).   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, int iters, unsigned long long* clk) {
    float a[16];
    v2 p[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x * 1e-3f + i; p[i] = v2{a[i], a[i] + 1.f}; }
    float m = 1.0001f + threadIdx.x * 1e-9f, c = 0.5f + threadIdx.x * 1e-9f;
    v2 pm = {m, 0.9999f}, pc = {c, 0.25f};
    const float sm = 1.0001f, sc = 0.5f;   // wave-uniform -> SGPR operands
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pm), "v"(pc));
            if (MODE == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (MODE == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
            if (MODE == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pm));
            if (MODE == 5) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (MODE == 6) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(sm), "v"(c));     // 2 VGPR + SGPR
            if (MODE == 7) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));          // acc in place
            if (MODE == 8) asm volatile("v_mul_f32 %0, 0x3f3504f3, %0" : "+v"(a[i]));                    // literal
            if (MODE == 9) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "s"(sc));                  // 1 VGPR + SGPR
            if (MODE == 10) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(sm), "s"(sm));    // 1 VGPR
            if (MODE == 11) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(c));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i] + p[i][0] + p[i][1];
    extern __shared__ float dyn[];
    if (s == 123.456f) dyn[threadIdx.x] = s;
    out[(blockIdx.x & 255) * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, float* d, unsigned long long* dclk, int waves_per_simd) {
    const int threads = 64 * 4 * waves_per_simd;   // one block per CU
    const int iters = 100000;
    const int lds = 100 * 1024;   // forces one block per CU
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), lds, 0, d, iters, dclk);   // warm-up (clock ramp)
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * 8), dim3(threads), lds, 0, d, iters, dclk);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2];
    hipMemcpy(h, dclk, sizeof h, hipMemcpyDeviceToHost);
    const double instr_per_simd = (double)iters * 16 * waves_per_simd * 8;
    printf("%-22s waves/SIMD=%d  %7.2f memtime ticks, %6.3f ns wall per wave-instruction per SIMD  (memtime %.2f GHz, %.2f ms)\n", name,
           waves_per_simd, (double)h[0] * 8 / instr_per_simd, ms * 1e6 / instr_per_simd, (double)h[0] / ((double)h[1] * 10.0), ms);
}

int main() {
    float* d; hipMalloc(&d, 256 * 1024 * 4);
    unsigned long long* dclk; hipMalloc(&dclk, 16);
    for (int w : {1, 2, 3, 4}) {
        run<2>("v_add_f32 v,v", d, dclk, w);
        run<9>("v_add_f32 v,s", d, dclk, w);
        run<5>("v_mul_f32 v,v", d, dclk, w);
        run<8>("v_mul_f32 lit,v", d, dclk, w);
        run<0>("v_fma_f32 v,v,v", d, dclk, w);
        run<7>("v_fmac_f32 v,v", d, dclk, w);
        run<6>("v_fma_f32 v,s,v", d, dclk, w);
        run<10>("v_fma_f32 v,s,s", d, dclk, w);
        run<11>("v_mov_b32", d, dclk, w);
        run<3>("v_pk_add_f32", d, dclk, w);
        run<4>("v_pk_mul_f32", d, dclk, w);
        run<1>("v_pk_fma_f32", d, dclk, w);
    }
    return 0;
}
