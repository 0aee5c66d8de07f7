// Developer micro-benchmark: run ONE kind of work for ~8 s so rocm-smi can sample package power and sclk.
//   power_modes <mode> [seconds] [K]   mode: fma | add | read | read8 | read_fx | read_lin8 | lds | fma_read (K FMAs per 16-byte load, default 6)
// hipcc --offload-arch=gfx950 -O3 -o power_modes power_modes.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <chrono>

typedef float v2f __attribute__((ext_vector_type(2)));
// MODE: 0 v_add_f32, 1 v_fma_f32, 2 v_mul_f32, 3 v_pk_add_f32, 4 v_pk_fma_f32, 5 v_pk_mul_f32, 6 v_mov_b32
template <int MODE>
__global__ __launch_bounds__(512) void valu_k(float* out, int iters) {
    float a[16];
    v2f p[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x * 1e-3f + i; p[i] = v2f{a[i], a[i] + 1.f}; }
    float m = 1.0001f + threadIdx.x * 1e-9f, c = 0.5f + threadIdx.x * 1e-9f;
    v2f pm = {m, 0.9999f}, pc = {c, 0.25f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (MODE == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (MODE == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                if (MODE == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (MODE == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
                if (MODE == 4) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pm), "v"(pc));
                if (MODE == 5) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pm));
                if (MODE == 6) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(c));
            }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i] + p[i][0] + p[i][1];
    extern __shared__ float dyn[];
    if (s == 123.456f) dyn[threadIdx.x] = s;
    out[(blockIdx.x & 255) * blockDim.x + threadIdx.x] = s;
}

// streaming read: each block walks a contiguous slab with 16-byte loads, sums into a register
__global__ __launch_bounds__(512) void read_k(const float4* __restrict__ in, float* out, size_t n_vec, int fma_per_load) {
    float4 acc = {0, 0, 0, 0};
    const size_t per_block = n_vec / gridDim.x;
    const float4* p = in + (size_t)blockIdx.x * per_block;
    float m = 1.0001f, c = 0.5f;
    for (size_t i = threadIdx.x; i < per_block; i += 512 * 4) {
        float4 v0 = p[i], v1 = p[i + 512], v2 = p[i + 1024], v3 = p[i + 1536];
        acc.x += v0.x + v1.x + v2.x + v3.x; acc.y += v0.y + v1.y + v2.y + v3.y;
        acc.z += v0.z + v1.z + v2.z + v3.z; acc.w += v0.w + v1.w + v2.w + v3.w;
        for (int f = 0; f < fma_per_load; ++f) {
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc.x) : "v"(m), "v"(c));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc.y) : "v"(m), "v"(c));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc.z) : "v"(m), "v"(c));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc.w) : "v"(m), "v"(c));
        }
    }
    out[(blockIdx.x & 255) * 512 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

__global__ __launch_bounds__(512) void read8_k(const float2* __restrict__ in, float* out, size_t n_vec) {
    float2 acc = {0, 0};
    const size_t per_block = n_vec / gridDim.x;
    const float2* p = in + (size_t)blockIdx.x * per_block;
    for (size_t i = threadIdx.x; i < per_block; i += 512 * 8) {
        float2 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = p[i + 512 * k];
#pragma unroll
        for (int k = 0; k < 8; ++k) { acc.x += v[k].x; acc.y += v[k].y; }
    }
    out[(blockIdx.x & 255) * 512 + threadIdx.x] = acc.x + acc.y;
}

// the fused F+X kernel's read pattern without its arithmetic: one persistent block per CU walks chunk pairs b, b + 256, ...
// (4 MiB each: two 2 MiB streams); per step (one 4096-sample frame of both streams, 64 KiB) thread (ant, j) loads its 16
// branch samples with 8-byte loads, 512 contiguous bytes per wave-load, frame after frame
__global__ __launch_bounds__(512) void read_fx_k(const float2* __restrict__ in, float* out, int n_chunks) {
    const int ant = threadIdx.x >> 8, j = threadIdx.x & 255;
    float2 acc = {0, 0};
    for (int c = blockIdx.x; c < n_chunks; c += gridDim.x) {
        const float2* base = in + (size_t)c * 2 * 262144 + (size_t)ant * 262144 + (255 - j);
        for (int i = 0; i < 64; ++i) {
            float2 v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = base[(size_t)i * 4096 + 256 * (15 - r)];
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc.x += v[r].x; acc.y += v[r].y; }
        }
    }
    out[(blockIdx.x & 255) * 512 + threadIdx.x] = acc.x + acc.y;
}

// the same bytes as one linear walk: block b reads the b-th 1/256 of the buffer front to back, 8-byte loads
__global__ __launch_bounds__(512) void read_lin8_k(const float2* __restrict__ in, float* out, size_t n_vec) {
    float2 acc = {0, 0};
    const size_t per_block = n_vec / gridDim.x;
    const float2* p = in + (size_t)blockIdx.x * per_block;
    for (size_t i = threadIdx.x; i < per_block; i += 512 * 16) {
        float2 v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = p[i + 512 * k];
#pragma unroll
        for (int k = 0; k < 16; ++k) { acc.x += v[k].x; acc.y += v[k].y; }
    }
    out[(blockIdx.x & 255) * 512 + threadIdx.x] = acc.x + acc.y;
}

__global__ __launch_bounds__(512) void sleep_k(float* out, int iters, int mode) {
    float s = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        if (mode == 0) __builtin_amdgcn_s_sleep(32);
        else asm volatile("s_nop 7\ns_nop 7\ns_nop 7\ns_nop 7");
    }
    out[(blockIdx.x & 255) * 512 + threadIdx.x] = s;
}

__global__ __launch_bounds__(512) void lds_k(float* out, int iters) {
    __shared__ float2 buf[512 * 17];
    float2 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = float2{(float)threadIdx.x, (float)i};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) buf[threadIdx.x * 17 + i] = v[i];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = buf[((threadIdx.x + 64) & 511) * 17 + i];
        __syncthreads();
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i].x + v[i].y;
    out[(blockIdx.x & 255) * 512 + threadIdx.x] = s;
}

constexpr int kPin = 100 * 1024;   // dynamic LDS that forces one block per CU

template <int M>
static void pin() { hipFuncSetAttribute((const void*)valu_k<M>, hipFuncAttributeMaxDynamicSharedMemorySize, kPin); }

int main(int argc, char** argv) {
    pin<0>(); pin<1>(); pin<2>(); pin<3>(); pin<4>(); pin<5>(); pin<6>();
    hipFuncSetAttribute((const void*)read_fx_k, hipFuncAttributeMaxDynamicSharedMemorySize, kPin);
    hipFuncSetAttribute((const void*)read_lin8_k, hipFuncAttributeMaxDynamicSharedMemorySize, kPin);
    const char* mode = argc > 1 ? argv[1] : "fma";
    const double seconds = argc > 2 ? atof(argv[2]) : 8.0;
    float* out; hipMalloc(&out, 256 * 512 * 4);
    const size_t n_vec = (size_t)8 << 30 >> 4;            // 8 GiB of float4
    float4* in = nullptr;
    if (strstr(mode, "read")) { hipMalloc(&in, n_vec * 16); hipMemset(in, 0, n_vec * 16); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto t_start = std::chrono::steady_clock::now();
    double total_ms = 0; long launches = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() < seconds) {
        hipEventRecord(e0, 0);
        if (!strcmp(mode, "fma")) hipLaunchKernelGGL(valu_k<1>, dim3(256), dim3(512), kPin, 0, out, 200000);
        else if (!strcmp(mode, "add")) hipLaunchKernelGGL(valu_k<0>, dim3(256), dim3(512), kPin, 0, out, 200000);
        else if (!strcmp(mode, "mul")) hipLaunchKernelGGL(valu_k<2>, dim3(256), dim3(512), kPin, 0, out, 200000);
        else if (!strcmp(mode, "pk_add")) hipLaunchKernelGGL(valu_k<3>, dim3(256), dim3(512), kPin, 0, out, 200000);
        else if (!strcmp(mode, "pk_fma")) hipLaunchKernelGGL(valu_k<4>, dim3(256), dim3(512), kPin, 0, out, 200000);
        else if (!strcmp(mode, "pk_mul")) hipLaunchKernelGGL(valu_k<5>, dim3(256), dim3(512), kPin, 0, out, 200000);
        else if (!strcmp(mode, "mov")) hipLaunchKernelGGL(valu_k<6>, dim3(256), dim3(512), kPin, 0, out, 200000);
        else if (!strcmp(mode, "read")) hipLaunchKernelGGL(read_k, dim3(2048), dim3(512), 0, 0, in, out, n_vec, 0);
        else if (!strcmp(mode, "read8")) hipLaunchKernelGGL(read8_k, dim3(2048), dim3(512), 0, 0, (const float2*)in, out, n_vec * 2);
        else if (!strcmp(mode, "read_fx")) hipLaunchKernelGGL(read_fx_k, dim3(256), dim3(512), kPin, 0, (const float2*)in, out, 2048);
        else if (!strcmp(mode, "read_lin8")) hipLaunchKernelGGL(read_lin8_k, dim3(256), dim3(512), kPin, 0, (const float2*)in, out, n_vec * 2);
        else if (!strcmp(mode, "fma_read")) hipLaunchKernelGGL(read_k, dim3(2048), dim3(512), 0, 0, in, out, n_vec, argc > 3 ? atoi(argv[3]) : 6);
        else if (!strcmp(mode, "lds")) hipLaunchKernelGGL(lds_k, dim3(256), dim3(512), 0, 0, out, 100000);
        else if (!strcmp(mode, "sleep")) hipLaunchKernelGGL(sleep_k, dim3(256), dim3(512), 0, 0, out, 100000, 0);
        else if (!strcmp(mode, "nop")) hipLaunchKernelGGL(sleep_k, dim3(256), dim3(512), 0, 0, out, 1000000, 1);
        else { printf("unknown mode\n"); return 1; }
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        total_ms += ms; ++launches;
    }
    const double ms = total_ms / launches;
    if (!strcmp(mode, "fma") || !strcmp(mode, "add") || !strcmp(mode, "mul") || !strncmp(mode, "pk_", 3) || !strcmp(mode, "mov"))
        printf("%s: %.2f ms per launch, %.3f ns per wave-instruction per SIMD\n", mode, ms, ms * 1e6 / (200000.0 * 64 * 2));
    else if (strstr(mode, "read"))
        printf("%s: %.2f ms per launch, %.1f GB/s\n", mode, ms, n_vec * 16 / ms / 1e6);
    else if (!strcmp(mode, "sleep") || !strcmp(mode, "nop"))
        printf("%s: %.2f ms per launch\n", mode, ms);
    else
        printf("%s: %.2f ms per launch, %.2f ns per (16 b64 writes + 16 b64 reads + 2 barriers) per WG\n", mode, ms, ms * 1e6 / 100000.0);
    return 0;
}
