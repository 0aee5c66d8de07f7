// Minimal reproducer: gfx950 (MI355X) needs wait states between a buffer store of more than 8 bytes and a VALU write of
// its data registers ALSO when the store carries an SGPR offset.  LLVM's GCNHazardRecognizer (ROCm 7.2) pads that hazard
// only for stores without an soffset register ("this hazard only exists if the instruction is not using a register in the
// soffset field"), so compiled code can contain
//     buffer_store_dwordx4 v[0:3], v40, s[8:11], s0 offen
//     v_mov_b64_e32        v[0:1], s[24:25]
// back to back -- found in this repository's pre-filter kernel (profiles/r02/round2_experiments.md), where the store then
// wrote the *moved* value whenever its issue was delayed by a busy memory pipeline.
//
// Each thread stores a 16-byte tag {block, thread, iteration, 0x600D} per iteration with hand-placed instructions and
// overwrites the first data register right behind the store; three variants of the two instructions:
//   mode 0  SGPR soffset, no wait state          (what the compiler emits)    -> corrupted records expected on gfx950
//   mode 1  SGPR soffset, s_nop 1 in between     (what the hazard needs)      -> clean
//   mode 2  offset folded into the VGPR, soffset = 0, no wait state by hand   -> clean only if the hardware hazard is
//           indeed tied to the soffset field; the compiler pads this form itself
// A second stream of plain stores from the same waves keeps the memory pipeline busy.
//   hipcc --offload-arch=gfx950 -O2 -o store_hazard store_hazard.hip && ./store_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void hazard_kernel(unsigned* __restrict__ out, unsigned* __restrict__ noise, int iters) {
    const unsigned tid = threadIdx.x, b = blockIdx.x;
    // raw buffer resource over this block's records: base, stride 0, num_records in bytes, dst_sel/format word of a raw buffer
    const unsigned long long base = (unsigned long long)(out + (size_t)b * iters * 256 * 4);
    v4i rs;
    rs.x = (int)(unsigned)(base & 0xffffffffull);
    rs.y = (int)(unsigned)(base >> 32);
    rs.z = iters * 256 * 16;
    rs.w = 0x00020000;
    unsigned* my_noise = noise + ((size_t)b * 256 + tid) * 64;
    for (int it = 0; it < iters; ++it) {
        const unsigned voff = tid * 16u, soff = (unsigned)it * 256u * 16u;
        for (int k = 0; k < 64; ++k) my_noise[k] = it + k;          // keep the vector-memory pipeline busy
        if (MODE == 0)
            asm volatile(
                "v_mov_b32 v20, %[a]\n\tv_mov_b32 v21, %[t]\n\tv_mov_b32 v22, %[i]\n\tv_mov_b32 v23, 0x600d\n\t"
                "s_nop 4\n\t"
                "buffer_store_dwordx4 v[20:23], %[vo], %[rs], %[so] offen\n\t"
                "v_mov_b32 v20, 0xbad\n\t"
                :: [a] "v"(b), [t] "v"(tid), [i] "v"((unsigned)it), [vo] "v"(voff), [rs] "s"(rs), [so] "s"(soff)
                : "v20", "v21", "v22", "v23", "memory");
        else if (MODE == 1)
            asm volatile(
                "v_mov_b32 v20, %[a]\n\tv_mov_b32 v21, %[t]\n\tv_mov_b32 v22, %[i]\n\tv_mov_b32 v23, 0x600d\n\t"
                "s_nop 4\n\t"
                "buffer_store_dwordx4 v[20:23], %[vo], %[rs], %[so] offen\n\t"
                "s_nop 1\n\t"
                "v_mov_b32 v20, 0xbad\n\t"
                :: [a] "v"(b), [t] "v"(tid), [i] "v"((unsigned)it), [vo] "v"(voff), [rs] "s"(rs), [so] "s"(soff)
                : "v20", "v21", "v22", "v23", "memory");
        else if (MODE == 2)
            asm volatile(
                "v_mov_b32 v20, %[a]\n\tv_mov_b32 v21, %[t]\n\tv_mov_b32 v22, %[i]\n\tv_mov_b32 v23, 0x600d\n\t"
                "s_nop 4\n\t"
                "buffer_store_dwordx4 v[20:23], %[vo], %[rs], 0 offen\n\t"
                "v_mov_b32 v20, 0xbad\n\t"
                :: [a] "v"(b), [t] "v"(tid), [i] "v"((unsigned)it), [vo] "v"(voff + soff), [rs] "s"(rs)
                : "v20", "v21", "v22", "v23", "memory");
        else if (MODE == 3)      // as mode 0, the overwrite a 64-bit move from SGPRs (the instruction pair found in the compiled kernel)
            asm volatile(
                "v_mov_b32 v20, %[a]\n\tv_mov_b32 v21, %[t]\n\tv_mov_b32 v22, %[i]\n\tv_mov_b32 v23, 0x600d\n\t"
                "s_nop 4\n\t"
                "buffer_store_dwordx4 v[20:23], %[vo], %[rs], %[so] offen\n\t"
                "v_mov_b64 v[20:21], %[bad]\n\t"
                :: [a] "v"(b), [t] "v"(tid), [i] "v"((unsigned)it), [vo] "v"(voff), [rs] "s"(rs), [so] "s"(soff),
                   [bad] "s"(0x00000bad00000badull)
                : "v20", "v21", "v22", "v23", "memory");
        else                     // as mode 3, behind eight loads in flight on the same wave (a store that has to queue)
            asm volatile(
                "v_mov_b32 v20, %[a]\n\tv_mov_b32 v21, %[t]\n\tv_mov_b32 v22, %[i]\n\tv_mov_b32 v23, 0x600d\n\t"
                "buffer_load_dwordx4 v[24:27], %[vo], %[rs], 0 offen\n\t"
                "buffer_load_dwordx4 v[28:31], %[vo], %[rs], 0 offen offset:16\n\t"
                "buffer_load_dwordx4 v[32:35], %[vo], %[rs], 0 offen offset:32\n\t"
                "buffer_load_dwordx4 v[36:39], %[vo], %[rs], 0 offen offset:48\n\t"
                "buffer_load_dwordx4 v[40:43], %[vo], %[rs], 0 offen offset:64\n\t"
                "buffer_load_dwordx4 v[44:47], %[vo], %[rs], 0 offen offset:80\n\t"
                "buffer_load_dwordx4 v[48:51], %[vo], %[rs], 0 offen offset:96\n\t"
                "buffer_load_dwordx4 v[52:55], %[vo], %[rs], 0 offen offset:112\n\t"
                "buffer_store_dwordx4 v[20:23], %[vo], %[rs], %[so] offen\n\t"
                "v_mov_b64 v[20:21], %[bad]\n\t"
                "s_waitcnt vmcnt(0)\n\t"
                :: [a] "v"(b), [t] "v"(tid), [i] "v"((unsigned)it), [vo] "v"(voff), [rs] "s"(rs), [so] "s"(soff),
                   [bad] "s"(0x00000bad00000badull)
                : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35",
                  "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51",
                  "v52", "v53", "v54", "v55", "memory");
    }
}

template <int MODE>
static long run(int blocks, int iters) {
    const size_t n = (size_t)blocks * iters * 256 * 4;
    unsigned *d_out = nullptr, *d_noise = nullptr;
    hipMalloc(&d_out, n * 4);
    hipMalloc(&d_noise, (size_t)blocks * 256 * 64 * 4);
    hipMemset(d_out, 0, n * 4);
    hipLaunchKernelGGL(hazard_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, d_noise, iters);
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    std::vector<unsigned> h(n);
    hipMemcpy(h.data(), d_out, n * 4, hipMemcpyDeviceToHost);
    long bad = 0, moved = 0;
    for (int b = 0; b < blocks; ++b)
        for (int it = 0; it < iters; ++it)
            for (int t = 0; t < 256; ++t) {
                const unsigned* r = &h[(((size_t)b * iters + it) * 256 + t) * 4];
                if (r[0] != (unsigned)b || r[1] != (unsigned)t || r[2] != (unsigned)it || r[3] != 0x600d) ++bad;
                if (r[0] == 0xbad) ++moved;
            }
    hipFree(d_out);
    hipFree(d_noise);
    printf("mode %d: %ld of %zu records wrong, %ld of them carry the value written BEHIND the store\n", MODE, bad,
           (size_t)blocks * iters * 256, moved);
    return bad;
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int blocks = prop.multiProcessorCount * 8, iters = 64;
    printf("%s, %d CUs, %d blocks x 256 threads x %d iterations\n", prop.gcnArchName, prop.multiProcessorCount, blocks, iters);
    const long b0 = run<0>(blocks, iters), b1 = run<1>(blocks, iters), b2 = run<2>(blocks, iters);
    const long b3 = run<3>(blocks, iters), b4 = run<4>(blocks, iters);
    printf("verdict: %s\n", ((b0 > 0 || b3 > 0 || b4 > 0) && b1 == 0) ? "hazard reproduced: SGPR-offset store needs the wait states too"
                                                                     : "SGPR-offset forms clean in this run");
    printf("         the hazard itself (mode 2, no soffset register, hand-written without the wait states): %s\n",
           b2 > 0 ? "reproduced" : "not reproduced");
    return 0;
}
