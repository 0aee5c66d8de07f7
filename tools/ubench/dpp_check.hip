// Developer check: what v_mov_b32_dpp wave_shr:1 delivers on gfx950 (lane i should read lane i - 1 across rows)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    int v = threadIdx.x + 100;
    int s1 = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xF, 0xF, false);
    int s2 = __builtin_amdgcn_update_dpp(-1, s1, 0x138, 0xF, 0xF, false);
    out[threadIdx.x] = s1;
    out[64 + threadIdx.x] = s2;
}
int main() {
    int* d; hipMalloc(&d, 128 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    int h[128]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int i = 0; i < 64; ++i) printf("%d:%d/%d ", i, h[i], h[64 + i]);
    printf("\n");
    return 0;
}
