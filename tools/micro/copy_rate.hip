// Developer aid: what the memory system moves for the access patterns of the three-pass routes -- a grid-stride copy (16-byte loads,
// 16-byte stores) and a read-only stream, each with the default cache policy and with the nontemporal hint.
//   hipcc --offload-arch=gfx950 -O3 -o build/copy_rate tools/micro/copy_rate.hip && build/copy_rate
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));

template <bool NTL, bool NTS>
__global__ __launch_bounds__(256) void copy_kernel(const v4f* __restrict__ src, v4f* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const v4f v = NTL ? __builtin_nontemporal_load(src + i) : src[i];
        if (NTS) __builtin_nontemporal_store(v, dst + i);
        else dst[i] = v;
    }
}

template <bool NTL>
__global__ __launch_bounds__(256) void read_kernel(const v4f* __restrict__ src, float* __restrict__ out, size_t n) {
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        acc += NTL ? __builtin_nontemporal_load(src + i) : src[i];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = 1.f;      // (never true: keeps the loads)
}

template <class F>
static double time_ms(F launch) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    launch();
    hipDeviceSynchronize();
    std::vector<float> ms;
    for (int r = 0; r < 7; ++r) {
        hipEventRecord(a);
        launch();
        hipEventRecord(b);
        hipEventSynchronize(b);
        float t = 0.f;
        hipEventElapsedTime(&t, a, b);
        ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
}

// `copy_rate three [bytes]`: what three passes over `bytes` (default 8 589 934 592 = BASELINE configs[4]'s samples: 512 chunks x 8 antennas x 2^18 x 8 B)
// cost the memory system at its best -- a copy (read once, write once) and a read of what was written, with the cache policies of the
// shipped route (nontemporal loads and stores) and with the default ones: the ceiling an F pass + X pass through HBM can reach, as a
// fraction of 8 TB/s over the ALGORITHMIC bytes (the samples alone).
static int three_pass(size_t bytes) {
    const size_t n = bytes / sizeof(v4f);
    v4f *src, *dst;
    float* out;
    if (hipMalloc(&src, bytes) != hipSuccess || hipMalloc(&dst, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
    hipMemset(src, 1, bytes);
    hipMemset(dst, 0, bytes);
    const int grid = 256 * 16;
    const double gb = (double)bytes / 1e9;
    const double c_nt = time_ms([&] { hipLaunchKernelGGL((copy_kernel<true, true>), dim3(grid), dim3(256), 0, 0, src, dst, n); });
    const double r_nt = time_ms([&] { hipLaunchKernelGGL((read_kernel<true>), dim3(grid), dim3(256), 0, 0, dst, out, n); });
    const double c_df = time_ms([&] { hipLaunchKernelGGL((copy_kernel<false, false>), dim3(grid), dim3(256), 0, 0, src, dst, n); });
    const double r_df = time_ms([&] { hipLaunchKernelGGL((read_kernel<false>), dim3(grid), dim3(256), 0, 0, dst, out, n); });
    // back to back, as the route runs them
    const double both_nt = time_ms([&] {
        hipLaunchKernelGGL((copy_kernel<true, true>), dim3(grid), dim3(256), 0, 0, src, dst, n);
        hipLaunchKernelGGL((read_kernel<true>), dim3(grid), dim3(256), 0, 0, dst, out, n);
    });
    std::printf("{\"bytes\": %zu, \"copy_nt_ms\": %.3f, \"read_nt_ms\": %.3f, \"copy_default_ms\": %.3f, \"read_default_ms\": %.3f, \"copy_then_read_nt_ms\": %.3f, "
                "\"three_pass_ceiling_of_8TBs_nt\": %.4f, \"three_pass_ceiling_of_8TBs_default\": %.4f}\n",
                bytes, c_nt, r_nt, c_df, r_df, both_nt, gb / both_nt * 1e3 / 8000.0, gb / (c_df + r_df) * 1e3 / 8000.0);
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && std::string(argv[1]) == "read16") {      // one read-only pass of 4 GiB with 16-byte loads, for `rocprofv3 --pmc FETCH_SIZE` (calibrates the gfx950 x 2)
        const size_t bytes = (size_t)4 << 30, n = bytes / sizeof(v4f);
        v4f* src;
        float* out;
        if (hipMalloc(&src, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
        hipMemset(src, 1, bytes);
        const bool nt = argc > 2 && std::string(argv[2]) == "nt";
        for (int r = 0; r < 5; ++r) {
            if (nt) hipLaunchKernelGGL((read_kernel<true>), dim3(256 * 16), dim3(256), 0, 0, src, out, n);
            else hipLaunchKernelGGL((read_kernel<false>), dim3(256 * 16), dim3(256), 0, 0, src, out, n);
        }
        hipDeviceSynchronize();
        std::printf("read16 %s: 5 launches of %zu bytes\n", nt ? "nt" : "default", bytes);
        return 0;
    }
    if (argc > 1 && std::string(argv[1]) == "three") return three_pass(argc > 2 ? (size_t)std::atoll(argv[2]) : (size_t)8589934592ull);
    const size_t bytes = (size_t)4 << 30;      // 4 GiB in, 4 GiB out
    const size_t n = bytes / sizeof(v4f);
    v4f *src, *dst;
    float* out;
    if (hipMalloc(&src, bytes) != hipSuccess || hipMalloc(&dst, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
    hipMemset(src, 1, bytes);
    hipMemset(dst, 0, bytes);
    const int grid = 256 * 16;
    const double gb = (double)bytes / 1e9;
    double t;
    t = time_ms([&] { hipLaunchKernelGGL((copy_kernel<false, false>), dim3(grid), dim3(256), 0, 0, src, dst, n); });
    std::printf("copy  default loads, default stores : %.3f ms  %.0f GB/s in + out\n", t, 2 * gb / t * 1e3);
    t = time_ms([&] { hipLaunchKernelGGL((copy_kernel<false, true>), dim3(grid), dim3(256), 0, 0, src, dst, n); });
    std::printf("copy  default loads, nt stores      : %.3f ms  %.0f GB/s in + out\n", t, 2 * gb / t * 1e3);
    t = time_ms([&] { hipLaunchKernelGGL((copy_kernel<true, false>), dim3(grid), dim3(256), 0, 0, src, dst, n); });
    std::printf("copy  nt loads, default stores      : %.3f ms  %.0f GB/s in + out\n", t, 2 * gb / t * 1e3);
    t = time_ms([&] { hipLaunchKernelGGL((copy_kernel<true, true>), dim3(grid), dim3(256), 0, 0, src, dst, n); });
    std::printf("copy  nt loads, nt stores           : %.3f ms  %.0f GB/s in + out\n", t, 2 * gb / t * 1e3);
    t = time_ms([&] { hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, 0); });
    std::printf("hipMemcpyDtoD                       : %.3f ms  %.0f GB/s in + out\n", t, 2 * gb / t * 1e3);
    t = time_ms([&] { hipLaunchKernelGGL((read_kernel<false>), dim3(grid), dim3(256), 0, 0, src, out, n); });
    std::printf("read  default loads                 : %.3f ms  %.0f GB/s\n", t, gb / t * 1e3);
    t = time_ms([&] { hipLaunchKernelGGL((read_kernel<true>), dim3(grid), dim3(256), 0, 0, src, out, n); });
    std::printf("read  nt loads                      : %.3f ms  %.0f GB/s\n", t, gb / t * 1e3);
    return 0;
}
