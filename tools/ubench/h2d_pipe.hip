// Developer micro-benchmark: what slows a double-buffered pinned H2D stream (64 MiB batches, host waits one batch
// behind)?  hipcc --offload-arch=gfx950 -O3 -o h2d_pipe h2d_pipe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>

__global__ void touch(const float* in, float* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] + 1.f;
}

int main() {
    const size_t bytes = 64u << 20, obytes = 512u << 10;
    void *h[2], *d[2], *ho[2], *dd[2];
    for (int b = 0; b < 2; ++b) {
        hipHostMalloc(&h[b], bytes, hipHostMallocDefault); hipMalloc(&d[b], bytes);
        hipHostMalloc(&ho[b], obytes, hipHostMallocDefault); hipMalloc(&dd[b], obytes);
        memset(h[b], 1, bytes);
    }
    hipStream_t s_in, s_c, s_out;
    hipStreamCreateWithFlags(&s_in, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&s_c, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&s_out, hipStreamNonBlocking);
    hipEvent_t ev_in[2], ev_c[2], ev_done[2];
    for (int b = 0; b < 2; ++b) {
        hipEventCreateWithFlags(&ev_in[b], hipEventDisableTiming);
        hipEventCreateWithFlags(&ev_c[b], hipEventDisableTiming);
        hipEventCreateWithFlags(&ev_done[b], hipEventDisableTiming);
    }
    const char* names[] = {"warm-up", "H2D only", "+ kernel on the NULL stream", "+ kernel on a non-blocking stream",
                           "+ kernel (non-blocking) + D2H on a third stream", "+ kernel (non-blocking) + D2H on the kernel's stream",
                           "+ kernel (NULL) + D2H on a third stream", "+ kernel (NULL) + D2H on the NULL stream"};
    const int n = 24;
    for (int mode = 0; mode < 8; ++mode) {
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < n; ++k) {
            const int b = k & 1;
            hipMemcpyAsync(d[b], h[b], bytes, hipMemcpyHostToDevice, s_in);
            hipEventRecord(ev_in[b], s_in);
            hipEvent_t last = ev_in[b];
            if (mode >= 2) {
                hipStream_t cs = (mode == 2 || mode >= 6) ? (hipStream_t)0 : s_c;
                hipStreamWaitEvent(cs, ev_in[b], 0);
                hipLaunchKernelGGL(touch, dim3(512), dim3(256), 0, cs, (const float*)d[b], (float*)dd[b], (int)(obytes / 4));
                if (mode == 4 || mode == 6) {
                    hipEventRecord(ev_c[b], cs);
                    hipStreamWaitEvent(s_out, ev_c[b], 0);
                    hipMemcpyAsync(ho[b], dd[b], obytes, hipMemcpyDeviceToHost, s_out);
                    hipEventRecord(ev_done[b], s_out);
                } else {
                    if (mode == 5 || mode == 7) hipMemcpyAsync(ho[b], dd[b], obytes, hipMemcpyDeviceToHost, cs);
                    hipEventRecord(ev_done[b], cs);
                }
                last = ev_done[b];
            }
            (void)last;
            if (k > 0) hipEventSynchronize(mode >= 2 ? ev_done[(k - 1) & 1] : ev_in[(k - 1) & 1]);
        }
        hipDeviceSynchronize();
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("%-55s %6.2f GB/s\n", names[mode], n * (double)bytes / dt / 1e9);
    }
    return 0;
}
