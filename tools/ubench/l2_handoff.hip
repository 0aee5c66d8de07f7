// Probe: can the 32 workgroups of one XCD hand 64 KiB per workgroup and step to each other through that XCD's L2
// while each also streams 64 KiB of fresh input from HBM?  (would a same-XCD corner turn take the
// spectra of the multi-antenna route off the HBM round trip?)
//
// One persistent 512-thread workgroup per CU.  A workgroup reads its XCC id from the hardware register and takes a
// slot in that XCD's cluster (so same-XCD membership holds by construction, not by assuming a dispatch order).
// Per step: [stream 64 KiB of input] -> [store 64 KiB to its slot of the cluster's slab, plain stores] ->
// vmcnt(0), barrier, one agent-scope add to the cluster counter -> poll the counter (sc1 loads, bounded) ->
// [read the transposed 64 KiB: 2 KiB from every slot of the slab, sc1 loads (L1 bypassed, L2 served)].
// Modes (bit mask): 1 = input stream, 2 = slab stores, 4 = slab reads, 8 = slab reads from a far copy instead
// (every step a fresh region of a large buffer: the HBM round trip for comparison).
// Prints ms per step set and the checksum test; run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE for the traffic.
//
// Build: hipcc -O3 --offload-arch=gfx950 l2_handoff.hip -o l2_handoff
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kThreads = 512;
constexpr int kMaxCluster = 64;
constexpr long kSpinLimit = 4000000;

struct Ctl {
    unsigned arrivals[8];            // slot allocation per XCD
    unsigned pad0[24];
    unsigned counter[8][32];         // one line per XCD: counter[x][0]
    unsigned failed;
};

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

typedef unsigned v4u32 __attribute__((ext_vector_type(4)));
constexpr int kAuxSc1 = 16;          // gfx940+: cache-policy bit 4 = sc1 (served by L2, L1 bypassed)

template <int J, int IN_AUX>          // IN_AUX: cache policy of the input loads (0 plain, 2 nt, 16 sc1, 18 sc1 nt); float4 per thread and step: 8 -> 64 KiB per workgroup and step, 4 -> 32 KiB, 2 -> 16 KiB
__global__ __launch_bounds__(kThreads) void handoff_kernel(const float4* __restrict__ input, float4* slab, const float4* far,
                                                           Ctl* ctl, float* out, int steps, int mode, int members,
                                                           unsigned epoch_base) {
    __shared__ unsigned s_slot, s_xcc, s_dead;
    __shared__ float s_pad[24 * 1024];                        // 96 KiB: one workgroup per CU
    constexpr int kSlotF4 = J * kThreads, kSlotBytes = kSlotF4 * 16;
    const int tid = threadIdx.x;
    if (tid == 0) {
        const unsigned x = xcc_id();
        s_xcc = x;
        s_dead = 0;
        s_slot = atomicAdd(&ctl->arrivals[x], 1u) % (unsigned)members;
    }
    s_pad[tid] = 0.f;
    __syncthreads();
    const unsigned xcc = s_xcc, slot = s_slot;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // slab: [xcc][buf 2][members][kSlotF4]
    float4* my_slab = slab + (size_t)xcc * 2 * members * kSlotF4;
    const float4* in_wg = input + (size_t)blockIdx.x * (size_t)steps * kSlotF4;
    const int per_slot_f4 = kSlotF4 / members;               // 2 KiB = 128 float4 when members = 32

    for (int s = 0; s < steps; ++s) {
        float4 v[J];
        #pragma unroll
        for (int j = 0; j < J; ++j) v[j] = make_float4((float)(s + 1), (float)slot, (float)j, (float)tid);
        if (mode & 1) {
            __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(in_wg + (size_t)s * kSlotF4), 0,
                                                                           kSlotBytes, 0x00020000);
            v4u32 t[J];
            #pragma unroll
            for (int j = 0; j < J; ++j) t[j] = __builtin_amdgcn_raw_buffer_load_b128(ri, (unsigned)(j * kThreads + tid) * 16u, 0, IN_AUX);
            #pragma unroll
            for (int j = 0; j < J; ++j) {
                v[j].x += __uint_as_float(t[j].x); v[j].y += __uint_as_float(t[j].y);
                v[j].z += __uint_as_float(t[j].z); v[j].w += __uint_as_float(t[j].w);
            }
        }
        float4* dst = my_slab + ((size_t)(s & 1) * members + slot) * kSlotF4;
        if (mode & 2) {
            #pragma unroll
            for (int j = 0; j < J; ++j) dst[j * kThreads + tid] = v[j];
        } else {
            #pragma unroll
            for (int j = 0; j < J; ++j) { acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w; }
        }
        if (mode & (4 | 8)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0 && !s_dead) {
                __hip_atomic_fetch_add(&ctl->counter[xcc][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned want = epoch_base + (unsigned)members * (unsigned)(s + 1);
                long spin = 0;
                for (;;) {
                    unsigned c;
                    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(c) : "v"(&ctl->counter[xcc][0]) : "memory");
                    if ((int)(c - want) >= 0) break;
                    if (++spin > kSpinLimit) { ctl->failed = 1; s_dead = 1; break; }
                    if ((spin & 1023) == 0 && __hip_atomic_load(&ctl->failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { s_dead = 1; break; }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            __syncthreads();
            // transposed read: piece `slot` of every member's 64 KiB
            const float4* src_base = (mode & 8) ? far + ((size_t)blockIdx.x * steps + s) * kSlotF4
                                                : my_slab + (size_t)(s & 1) * members * kSlotF4;
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(src_base), 0,
                                                                           members * kSlotBytes, 0x00020000);
            v4u32 t[J];
            #pragma unroll
            for (int j = 0; j < J; ++j) {
                const int e = j * kThreads + tid;                 // 0 .. 4095 float4 of my transposed 64 KiB
                const int m = e / per_slot_f4, o = e % per_slot_f4;
                const unsigned off = (mode & 8) ? (unsigned)e * 16u : ((unsigned)m * kSlotF4 + slot * per_slot_f4 + o) * 16u;
                t[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, kAuxSc1);
            }
            #pragma unroll
            for (int j = 0; j < J; ++j) {
                const int m = (j * kThreads + tid) / per_slot_f4;
                const float x = __uint_as_float(t[j].x), y = __uint_as_float(t[j].y);
                // check: x = s+1 (+input, which is zero), y = writer's slot
                if (!(mode & 8) && (mode & 2) && (x != (float)(s + 1) || y != (float)m)) ctl->failed = 2;
                acc.x += x; acc.y += y; acc.z += __uint_as_float(t[j].z); acc.w += __uint_as_float(t[j].w);
            }
        }
    }
    out[(size_t)blockIdx.x * kThreads + tid] = acc.x + acc.y + acc.z + acc.w + s_pad[tid];
}

template <int J, int IN_AUX>
void run(int steps, int reps, int cus, int only_mode) {
    constexpr int kSlotBytes = J * kThreads * 16, kSlotF4 = J * kThreads;
    const int members = cus / 8;
    printf("== %d KiB per workgroup and step; slab (two buffers) %d KiB per XCD; input load policy bits %d\n", kSlotBytes / 1024, 2 * members * kSlotBytes / 1024, IN_AUX);
    const size_t in_f4 = (size_t)cus * steps * kSlotF4;
    float4 *input, *slab, *far;
    Ctl* ctl;
    float* out;
    CK(hipMalloc(&input, in_f4 * 16));
    CK(hipMalloc(&far, in_f4 * 16));
    CK(hipMalloc(&slab, (size_t)8 * 2 * members * kSlotBytes));
    CK(hipMalloc(&ctl, sizeof(Ctl)));
    CK(hipMalloc(&out, (size_t)cus * kThreads * 4));
    CK(hipMemset(input, 0, in_f4 * 16));
    CK(hipMemset(far, 0, in_f4 * 16));
    CK(hipMemset(slab, 0, (size_t)8 * 2 * members * kSlotBytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int modes[] = {1, 2, 3, 1 | 2 | 4, 2 | 4, 1 | 2 | 8, 1 | 8};
    const char* names[] = {"input stream only", "slab stores only", "input + slab stores", "input + slab stores + same-XCD reads (L2 hand-off)",
                           "slab stores + same-XCD reads, no input", "input + slab stores + far reads (round trip through HBM)", "input + far reads (2x read stream)"};
    for (size_t k = 0; k < sizeof modes / sizeof *modes; ++k) {
        if (only_mode >= 0 && modes[k] != only_mode) continue;
        float best = 1e30f;
        int failed = 0;
        for (int r = 0; r < reps; ++r) {
            CK(hipMemset(ctl, 0, sizeof(Ctl)));
            CK(hipEventRecord(e0));
            handoff_kernel<J, IN_AUX><<<dim3(cus), dim3(kThreads), 0, 0>>>(input, slab, far, ctl, out, steps, modes[k], members, 0u);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
            Ctl h;
            CK(hipMemcpy(&h, ctl, sizeof h, hipMemcpyDeviceToHost));
            failed |= (int)h.failed;
            if (r == 0 && k == 0) {
                printf("   arrivals per XCD:");
                for (int x = 0; x < 8; ++x) printf(" %u", h.arrivals[x]);
                printf("\n");
            }
        }
        const double bytes = (double)cus * steps * kSlotBytes;
        printf("mode %2d %-58s %8.3f ms  %6.2f us/step  %5.2f TB/s per stream  %s\n", modes[k], names[k], best,
               best * 1e3 / steps, bytes / (best * 1e-3) / 1e12, failed ? (failed == 1 ? "SPIN LIMIT" : "STALE DATA") : "ok");
    }
    CK(hipFree(input)); CK(hipFree(far)); CK(hipFree(slab)); CK(hipFree(ctl)); CK(hipFree(out));
}

int main(int argc, char** argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 2000;
    const int reps = argc > 2 ? atoi(argv[2]) : 5;
    const int kib = argc > 3 ? atoi(argv[3]) : 0;          // 0: all sizes
    const int only_mode = argc > 4 ? atoi(argv[4]) : -1;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("CUs %d, clusters of %d per XCD, %d steps\n", cus, cus / 8, steps);
    const int aux = argc > 5 ? atoi(argv[5]) : 0;
    if (aux == 0) {
        if (kib == 0 || kib == 64) run<8, 0>(steps, reps, cus, only_mode);
        if (kib == 0 || kib == 32) run<4, 0>(steps, reps, cus, only_mode);
        if (kib == 0 || kib == 16) run<2, 0>(steps, reps, cus, only_mode);
    } else if (aux == 2) {
        if (kib == 0 || kib == 64) run<8, 2>(steps, reps, cus, only_mode);
        if (kib == 0 || kib == 32) run<4, 2>(steps, reps, cus, only_mode);
        if (kib == 0 || kib == 16) run<2, 2>(steps, reps, cus, only_mode);
    } else if (aux == 18) {
        if (kib == 0 || kib == 64) run<8, 18>(steps, reps, cus, only_mode);
        if (kib == 0 || kib == 32) run<4, 18>(steps, reps, cus, only_mode);
        if (kib == 0 || kib == 16) run<2, 18>(steps, reps, cus, only_mode);
    }
    return 0;
}
