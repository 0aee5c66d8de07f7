// Probe: what a read-1 / write-1 pass can reach on MI355X depending on access width and walk order -- the ceiling for
// the PFB pre-filter / 8192-split passes (k_prepass.h), which stream a copy-shaped pass at ~4.7 TB/s.
//   lin8 / lin16     grid-stride copy, 8 / 16 bytes per lane
//   col8 / col16     the pre-pass walk: a 256-thread workgroup owns a 2 KiB (8 B/lane) or 4 KiB (16 B/lane) wide column
//                    of a [frames][frame_bytes] stream and walks down the frames, TP loads in flight then TP stores
// Build: hipcc -O3 --offload-arch=gfx950 copy_shapes.hip -o copy_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

template <typename T>
__global__ __launch_bounds__(256) void lin_copy(const T* __restrict__ x, T* __restrict__ y, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = x[i];
}

// stream s: frames of frame_elems T each; grid (frame_elems / 256, streams)
template <typename T, int TP>
__global__ __launch_bounds__(256) void col_copy(const T* __restrict__ x, T* __restrict__ y, int frame_elems, int frames) {
    const size_t base = (size_t)blockIdx.y * frames * frame_elems + blockIdx.x * 256 + threadIdx.x;
    for (int i0 = 0; i0 < frames; i0 += TP) {
        T v[TP];
#pragma unroll
        for (int k = 0; k < TP; ++k) v[k] = x[base + (size_t)(i0 + k) * frame_elems];
#pragma unroll
        for (int k = 0; k < TP; ++k) y[base + (size_t)(i0 + k) * frame_elems] = v[k];
    }
}

template <typename F>
static float time_ms(F launch, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    launch();
    launch();
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0));
        launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    const size_t bytes = 4ull << 30;      // 4 GiB in, 4 GiB out
    void *x, *y;
    CK(hipMalloc(&x, bytes));
    CK(hipMalloc(&y, bytes));
    CK(hipMemset(x, 1, bytes));
    CK(hipMemset(y, 0, bytes));
    const double gb = 2.0 * bytes / 1e9;
    auto report = [&](const char* name, float ms) { printf("%-34s %7.3f ms  %5.2f TB/s (read + write)\n", name, ms, gb / ms); };
    for (int blocks : {2048, 8192, 65536}) {
        char name[64];
        snprintf(name, sizeof name, "lin8  grid %d", blocks);
        report(name, time_ms([&] { lin_copy<float2><<<blocks, 256>>>((const float2*)x, (float2*)y, bytes / 8); }, 5));
        snprintf(name, sizeof name, "lin16 grid %d", blocks);
        report(name, time_ms([&] { lin_copy<float4><<<blocks, 256>>>((const float4*)x, (float4*)y, bytes / 16); }, 5));
    }
    report("hipMemcpyDtoD", time_ms([&] { CK(hipMemcpyAsync(y, x, bytes, hipMemcpyDeviceToDevice, 0)); }, 5));
    // streams of 2 MiB (the headline's 262144 complex64), frame = nchan * 8 bytes
    for (int nchan : {512, 4096}) {
        const int frames = 262144 / nchan;
        const int streams = (int)(bytes / (2u << 20));
        char name[64];
        snprintf(name, sizeof name, "col8  nchan %d TP 8", nchan);
        report(name, time_ms([&] { col_copy<float2, 8><<<dim3(nchan / 256, streams), 256>>>((const float2*)x, (float2*)y, nchan, frames); }, 5));
        snprintf(name, sizeof name, "col8  nchan %d TP 32", nchan);
        report(name, time_ms([&] { col_copy<float2, 32><<<dim3(nchan / 256, streams), 256>>>((const float2*)x, (float2*)y, nchan, frames); }, 5));
        snprintf(name, sizeof name, "col16 nchan %d TP 8", nchan);
        report(name, time_ms([&] { col_copy<float4, 8><<<dim3(nchan / 512, streams), 256>>>((const float4*)x, (float4*)y, nchan / 2, frames); }, 5));
        snprintf(name, sizeof name, "col16 nchan %d TP 16", nchan);
        report(name, time_ms([&] { col_copy<float4, 16><<<dim3(nchan / 512, streams), 256>>>((const float4*)x, (float4*)y, nchan / 2, frames); }, 5));
    }
    return 0;
}
