// Developer micro-benchmark: does a buffer written by one kernel come back faster than HBM when the next kernel
// reads it (MI355X memory-side Infinity Cache, 256 MB)?  hipcc --offload-arch=gfx950 -O3 -o mall_probe mall_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void wr(float4* p, size_t n, float v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = float4{v, v, v, v};
}
__global__ __launch_bounds__(256) void rd(const float4* p, size_t n, float* out) {
    float4 a = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float4 v = p[i];
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    if (a.x + a.y + a.z + a.w == 123.f) out[0] = 1.f;
}

int main() {
    float4* buf; float* out;
    const size_t max_bytes = (size_t)4 << 30;
    hipMalloc(&buf, max_bytes); hipMalloc(&out, 4);
    hipEvent_t e0, e1, e2; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
    for (size_t mb : {16, 32, 64, 128, 192, 256, 512, 1024, 4096}) {
        const size_t n = mb * (1 << 20) / 16;
        double tw = 0, tr = 0;
        const int reps = 20;
        for (int r = 0; r < reps + 2; ++r) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(wr, dim3(2048), dim3(256), 0, 0, buf, n, (float)r);
            hipEventRecord(e1, 0);
            hipLaunchKernelGGL(rd, dim3(2048), dim3(256), 0, 0, buf, n, out);
            hipEventRecord(e2, 0);
            hipEventSynchronize(e2);
            float a, b; hipEventElapsedTime(&a, e0, e1); hipEventElapsedTime(&b, e1, e2);
            if (r >= 2) { tw += a; tr += b; }
        }
        printf("%5zu MB: write %8.1f GB/s   read-after-write %8.1f GB/s\n", mb, mb / 1024.0 / (tw / reps) * 1e3,
               mb / 1024.0 / (tr / reps) * 1e3);
    }
    return 0;
}
