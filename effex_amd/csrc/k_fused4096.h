// Part of libfxcorr's single translation unit: included by fxcorr.hip (not a stand-alone header).
#pragma once

namespace {

// ------------------------------------------------------------------------------------------
// fused 2-antenna, nchan = 4096, ntaps = 4 kernel (phases in fx_fused4096.h)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// vdst keeps its low half and receives src's low half in its high half; src gets the two high halves
__device__ __forceinline__ void permlane32_swap(cf& a, cf& b) {
    auto rx = __builtin_amdgcn_permlane32_swap(__float_as_uint(a.x), __float_as_uint(b.x), false, false);
    auto ry = __builtin_amdgcn_permlane32_swap(__float_as_uint(a.y), __float_as_uint(b.y), false, false);
    a = fxc::mk(__uint_as_float(rx[0]), __uint_as_float(ry[0]));
    b = fxc::mk(__uint_as_float(rx[1]), __uint_as_float(ry[1]));
}

// this thread's 16 branch samples of frame i: element (255 - j) + 256 (15 - r); loads r = R0 .. R0+CNT-1.
// Buffer loads: one VGPR byte offset per thread, everything that varies with chunk / frame / r is scalar.
typedef unsigned v2u32 __attribute__((ext_vector_type(2)));
typedef unsigned v4u32 __attribute__((ext_vector_type(4)));
#ifndef FXC_LOAD_AUX_U8
#define FXC_LOAD_AUX_U8 2      // ... of the byte-pair loads of the uint8 ingest: nt, 4 096 chunk pairs 3.61 - 3.65 -> 3.45 - 3.55 ms
#endif
#ifndef FXC_LOAD_AUX
#define FXC_LOAD_AUX 2   // cache policy of the IQ stream loads: bit 0 sc0, bit 1 nt, bit 4 sc1.  nt (the samples are read once): the
                         // headline kernel 8.57 - 8.66 -> 8.47 - 8.49 ms, the F-only variant of 8 antennas - 1.6 % (profiles/r05/experiments.md 10)
#endif
template <int R0, int CNT>
__device__ __forceinline__ void load_frame_part(cf (&xr)[16], const cf* chunk_base, unsigned chunk_bytes, unsigned voff,
                                                int64_t i) {
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<cf*>(chunk_base), 0, (int)chunk_bytes,
                                                                   0x00020000);
    const unsigned soff = (unsigned)(i * fxc::fused::kN * (int64_t)sizeof(cf));
#pragma unroll
    for (int r = R0; r < R0 + CNT; ++r) {
        const v2u32 d = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff + (unsigned)(256 * (15 - r) * sizeof(cf)), FXC_LOAD_AUX);
        xr[r] = fxc::mk(__uint_as_float(d[0]), __uint_as_float(d[1]));
    }
}

// uint8 ingest (RTL-SDR interleaved I,Q bytes): the same 16 branches as raw byte pairs, one
// 16-bit load each -- a quarter of the complex64 stream's HBM bytes
template <int R0, int CNT>
__device__ __forceinline__ void load_frame_part_u8(cf (&xr)[16], const unsigned short* chunk_base,
                                                   unsigned chunk_bytes, unsigned voff, int64_t i) {
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(chunk_base), 0,
                                                                   (int)chunk_bytes, 0x00020000);
    const unsigned soff = (unsigned)(i * fxc::fused::kN * (int64_t)sizeof(unsigned short));
#pragma unroll
    for (int r = R0; r < R0 + CNT; ++r)   // the byte pair waits in the slot's own register (bit pattern in .x)
        xr[r].x = __uint_as_float(
            (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rs, voff, soff + (unsigned)(256 * (15 - r) * sizeof(unsigned short)), FXC_LOAD_AUX_U8));
}

// byte pair -> complex64: (b - 127.5) / 127.5 minus the chunk's mean = b / 127.5 + off, off = -mean_byte / 127.5
// (pyrtlsdr's conversion behind effex.py:652 and the DC removal of effex.py:394-395 in one fused multiply-add)
// (in place: the raw pair sits in the slot's .x register, see load_frame_part_u8)
__device__ __forceinline__ void convert_frame_u8(cf (&h)[16], cf off) {
    const float k = 1.0f / 127.5f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const unsigned raw = __float_as_uint(h[r].x);
        h[r] = fxc::mk(fmaf((float)(raw & 0xFFu), k, off.x), fmaf((float)((raw >> 8) & 0xFFu), k, off.y));
    }
}

// FXC_ABL: developer-only timing ablations (wrong results by design; the shipped build has FXC_ABL == 0):
//   1 no barrier B0, 2 no barrier B1, 4 no exchange-1 LDS traffic, 8 no exchange-2 LDS traffic, 16 no IQ loads.
// FXC_STAMPS: diagnostic build with s_memtime stamps between the phases, summed per wave in scalar
// registers and printed by fxc_kernel_time().  profiles/r01/ablation_and_stamps.md has the readings.
#ifndef FXC_ABL
#define FXC_ABL 0
#endif
#ifndef FXC_STAMPS
#define FXC_STAMPS 0
#endif
constexpr int kStampSegs = 12;
constexpr int kDckLdsBytes = 64;       // DCK launches: the waves' byte sums, behind the carve
#if FXC_STAMPS
#define FXC_STAMP(k)                                                              \
    do {                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                        \
        const unsigned long long t_now__ = __builtin_amdgcn_s_memtime();          \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                       \
        seg[k] += t_now__ - t_prev;                                               \
        t_prev = t_now__;                                                         \
        __builtin_amdgcn_sched_barrier(0);                                        \
    } while (0)
#else
#define FXC_STAMP(k)
#endif
#if (FXC_ABL & 16)
#define FXC_PREFETCH(R0) ((void)0)
#else
#define FXC_PREFETCH(R0)                                                                                        \
    do {                                                                                                        \
        FXC_SCHED_FENCE();                                                                                      \
        if (U8)                                                                                                 \
            load_frame_part_u8<R0, 4>(nx, reinterpret_cast<const unsigned short*>(x) + (int64_t)pc * 2 * num_samp, \
                                      chunk_bytes, voff, nframe);                                               \
        else                                                                                                    \
            load_frame_part<R0, 4>(nx, nbase, chunk_bytes, voff, nframe);                                       \
        FXC_SCHED_FENCE();                                                                                      \
    } while (0)
#endif

// One spectrum of both antennas: frame i of chunk c sits in ring slot PH.  All control flow is
// wave-uniform and none of it guards a *definition* of ring registers (the prefetch is unconditional),
// which keeps the register allocator from doubling live ranges at merge points.
// SPEC_OUT: the multi-antenna variant -- the pair of streams is only channelised and both spectra go to
// HBM for xengine_kernel (rows_raw then is the spectra buffer [chunk][i][stream][specpos]).
// uint8 ingest state: this chunk's conversion offsets
struct U8State {
    cf off;
    bool have_next;      // DCK: the byte sums of the chunk that starts now are in LDS (this workgroup channelised a chunk before it)
    bool rounds;         // this part of the walk is the round-robin one (whole chunks, the next one gridDim.x further on)
};

// DC removal inside the kernel (template flag DCK, uint8 ingest only): while a workgroup channelises chunk c of its
// round-robin share it also sums the bytes of its next chunk, c + gridDim.x, so that chunk's conversion offset exists
// when its first frame is converted and the pre-pass over the bytes (dc_sum_u8_stream_kernel: 1.7 of 9.7 ms per 10 000
// chunk pairs) only has to cover each workgroup's first chunk and the tail.  The kernel has one VGPR to spare, so
// nothing of this lives in registers across a step: every step each thread fetches 32 bytes of its own antenna's stream
// straight into LDS (global_load_lds_dwordx4), adds their v_dot4 byte sums to its two counters in LDS a step later,
// and the chunk-start branch -- where the
// ring's history registers are dead -- turns the 256 counters of an antenna into the offset.  Exact integer sums and
// the float64 formula of dc_offsets_u8_kernel: bit-identical to the pre-pass.  Straight-line code except in that
// branch: a workgroup without a next chunk sums its current one again and nobody reads the result.
typedef unsigned v4u32_t __attribute__((ext_vector_type(4)));
constexpr int kDckStageVecs = 2 * fxc::fused::kThreads;     // two 16-byte vectors per thread and step

__device__ __forceinline__ void dck_fetch(const unsigned char* pair_base, int64_t num_samp, int64_t i, int tid, v4u32_t* stage) {
    // vectors i * 512 + j and i * 512 + 256 + j of stream (tid >> 8): num_samp is a whole number of 4096-sample frames.
    // The two fetches are written in assembly so that the compiler does not know of them.  If it does, it guards every LDS
    // access that might touch bytes in flight -- without alias information for the dynamic LDS array that is every
    // exchange store of the step -- with s_waitcnt vmcnt(0), which also waits for every IQ load issued since; and it hoists
    // the loop-invariant lane address out of the frame loop, spills it, and reloads it behind another vmcnt(0).  Unknown
    // loads in the queue only make the compiler's own vmcnt(N) waits stricter, never weaker (loads return in order).
    // LDS address of a lane's 16 bytes: M0 + lane * 16.
    typedef __attribute__((address_space(3))) v4u32_t* lptr_t;
    unsigned t = (unsigned)tid;
    asm volatile("" : "+v"(t));           // the lane offsets are formed here and now: a handful of instructions, no live range
    const unsigned lane_off = ((t >> 8) & 1u) * (unsigned)(2 * num_samp) + (t & 255u) * 16u;
    const unsigned char* base = pair_base + i * (fxc::fused::kThreads * 16);      // uniform
    const unsigned lds_wave = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lptr_t)(stage + (t >> 6) * 128));   // this wave's 2 x 64 vectors
    unsigned m0_saved;
    asm volatile(
        "s_mov_b32 %[sv], m0\n\t"
        "s_mov_b32 m0, %[l0]\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %[o0], %[base]\n\t"
        "s_add_u32 m0, %[l0], 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %[o1], %[base]\n\t"
        "s_mov_b32 m0, %[sv]\n\t"
        : [sv] "=&s"(m0_saved)
        : [l0] "s"(lds_wave), [o0] "v"(lane_off), [o1] "v"(lane_off + 4096u), [base] "s"(base)
        : "memory", "scc");
}

// The fetches of a step are summed a step later, just before the next ones go out (a whole step for them to land: summed
// at the end of their own step, 45 % of a step after their issue, the kernel waited for HBM).  Between the two points the
// wave issues the IQ loads FXC_PREFETCH(8), (12) of the old step and (0), (4) of the new one -- sixteen, and at a row
// end eight stores more: once at most sixteen vector-memory operations are outstanding the fetches have landed (loads
// return in order).  The last step of a chunk sums its own fetches as well, behind the four loads of FXC_PREFETCH(12).
__device__ __forceinline__ void dck_wait_prev() { asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); }
__device__ __forceinline__ void dck_wait_own() { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }

// this thread's two landing slots cleared: the next unconditional dck_add then adds nothing
__device__ __forceinline__ void dck_clear(v4u32_t* stage, int tid) {
    unsigned z0;
    asm volatile("" : "+v"(tid));
    asm volatile("v_mov_b32 %0, 0" : "=v"(z0));      // made here: the compiler otherwise keeps a zero quad in scratch for this
    const v4u32_t z = {z0, z0, z0, z0};
    stage[(tid >> 6) * 128 + (tid & 63)] = z;
    stage[(tid >> 6) * 128 + 64 + (tid & 63)] = z;
}

__device__ __forceinline__ void dck_add(const v4u32_t* stage, unsigned* acc, int tid) {
    asm volatile("" : "+v"(tid));         // the two LDS addresses are formed here, every step: as loop invariants they are spilled
    unsigned si = 0u, sq = 0u;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const v4u32_t w = stage[(tid >> 6) * 128 + h * 64 + (tid & 63)];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            si = __builtin_amdgcn_udot4(w[k], 0x00010001u, si, false);     // bytes 0, 2: I
            sq = __builtin_amdgcn_udot4(w[k], 0x01000100u, sq, false);     // bytes 1, 3: Q
        }
    }
    acc[2 * tid] += si;
    acc[2 * tid + 1] += sq;
}

// chunk start: the 256 threads' counters of this thread's antenna -> its conversion offset; counters cleared
__device__ __forceinline__ cf dck_offset(unsigned* acc, unsigned* red, int tid, int64_t num_samp) {
    // (lane addresses and the zero formed here, in the chunk-start branch: hoisted out of the frame loop they live in scratch)
    asm volatile("" : "+v"(tid));
    unsigned zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
    unsigned v[2] = {acc[2 * tid], acc[2 * tid + 1]};
    acc[2 * tid] = zero;
    acc[2 * tid + 1] = zero;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v[k] += __shfl_xor(v[k], d, 64);
    }
    __syncthreads();                       // every wave is done with the previous chunk's `red`
    if ((tid & 63) == 0) {
        red[(tid >> 6) * 2] = v[0];
        red[(tid >> 6) * 2 + 1] = v[1];
    }
    __syncthreads();
    const int ant = (tid >> 8) & 1;        // waves 0-3 summed antenna 0, waves 4-7 antenna 1
    unsigned long long ti = 0, tq = 0;     // exact: the launcher keeps num_samp <= 2^26, so a wave's 32-bit sums cannot wrap
#pragma unroll
    for (int wv = 0; wv < 4; ++wv) {
        ti += red[(4 * ant + wv) * 2];
        tq += red[(4 * ant + wv) * 2 + 1];
    }
    asm volatile("" : "+s"(num_samp));     // converted here, in the chunk-start branch: hoisted, the double lives in scratch
    const double mr = (double)ti / (double)num_samp, mi = (double)tq / (double)num_samp;
    return fxc::mk((float)(-mr / 127.5), (float)(-mi / 127.5));
}

#ifndef FXC_SPEC_STORE_AUX
#define FXC_SPEC_STORE_AUX 2      // cache policy of the F-only spectra stores: nt (written once, read by the X pass: 8 antennas 2.35 -> 2.19 ms)
#endif
template <int PH, bool SPEC_OUT, bool U8, bool DCK>
__device__ __forceinline__ void fused_step(fxc::fused::State& s, U8State& u8, const cf* dc, const f4* win, cf* region,
                                           const cf* tw2, int tid, const cf* x, int64_t num_samp, unsigned chunk_bytes,
                                           unsigned voff, fxc::fused::RangeWalk& pos, cf* rows_raw, int hp, v4u32_t* dck_stage,
                                           unsigned* dck_acc, unsigned* dck_red,
                                           unsigned long long (&seg)[kStampSegs], unsigned long long& t_prev) {
    using namespace fxc::fused;
    const int64_t c = pos.c, i = pos.i, n_pts = pos.n_pts;   // (the walk itself is 32-bit: scalar registers are scarce)
    FXC_STAMP(0);    // loop overhead and the (rare) row store since the previous step's last stamp
    if (i == 0) {    // zero PFB history at the start of every chunk
        asm volatile("" ::: "memory");   // keep this a (rarely taken) uniform branch, not 96 v_cndmask per frame
        state_reset_history<PH>(s);
        if (U8) {
            if (DCK && u8.have_next) u8.off = dck_offset(dck_acc, dck_red, tid, num_samp);
            else u8.off = dc[c * 2 + __builtin_amdgcn_readfirstlane((tid >> 8) & 1)];   // the antenna is a wave's: a scalar load
        }
    }
    // DCK: the chunk whose bytes this step helps to sum: this workgroup's next one, or (none left in this part: the sums
    // are never read) the current one again
    const bool dck_next = DCK && u8.rounds && pos.left > n_pts - i;
    const unsigned char* dck_base = reinterpret_cast<const unsigned char*>(x) + (int64_t)(c + (dck_next ? 1 + pos.seg_jump : 0)) * 4 * num_samp;
    if (U8) convert_frame_u8(s.h[PH], u8.off);   // the byte pairs fetched a step ago become the samples of slot PH
    cf v[16];
    phase1_fir<PH>(s, win, tid, v);      // first use of this frame: waits for its loads (issued a step ago)
    FXC_STAMP(2);
    // The oldest ring slot is dead now: refill it with the next frame of this workgroup's range (next frame of
    // the chunk, or frame 0 of the next chunk; at the very end the current frame again, never used).  The 16
    // loads go out in four groups spread over the step: eight waves bursting 16 loads each at the same point
    // stall in the in-order vector-memory issue (measured -7 %).
    int pc, nframe;
    range_walk_prefetch(pos, pc, nframe);
    const cf* nbase = x + (int64_t)pc * 2 * num_samp;
    cf (&nx)[16] = s.h[(PH + 1) & 3];
    FXC_PREFETCH(0);
    fxc::dft16_a(v);
    FXC_PREFETCH(4);
    FXC_STAMP(3);
#if !(FXC_ABL & 1)
    __syncthreads();   // B0: every wave has finished reading the previous spectrum's exchange rows
#endif
    FXC_STAMP(4);
#if !(FXC_ABL & 4)
    // second half of the radix-16 with the twiddle w4096^(j k1) and the exchange-1 store of every output as it forms:
    // the stores are bound by the LDS write path, the butterflies and twiddles run in its shadow (B0 in front of the
    // whole radix-16 instead: +7 %; exchange 2 streamed the same way: spills, +6 %)
    phase1_finish_store(s, v, region, tid);
#endif
    FXC_STAMP(5);
#if !(FXC_ABL & 2)
    __syncthreads();   // B1: exchange-1 rows complete
#endif
    if (DCK) {      // straight-line: what the previous step fetched (nothing at a chunk's first step: slots cleared), then fetch
        dck_wait_prev();
        dck_add(dck_stage, dck_acc, tid);
        dck_fetch(dck_base, num_samp, i, tid, dck_stage);
    }
    FXC_STAMP(6);
#if !(FXC_ABL & 4)
    phase2_load(region, tid, v);
#endif
    FXC_PREFETCH(8);
    fxc::dft16(v);
    FXC_STAMP(7);
    phase2_twiddle(v, tw2, tid);
    FXC_STAMP(8);
#if !(FXC_ABL & 8)
    wave_sync();       // exchange 2 is a 16x16 transpose inside each 16-lane group: no s_barrier
    phase2_store(v, region, tid);
    wave_sync();
#endif
    FXC_PREFETCH(12);
#if !(FXC_ABL & 8)
    phase3_load(region, tid, v);
#endif
    FXC_STAMP(9);
    fxc::dft16(v);
    if (SPEC_OUT) {
        // stream = 2 * (virtual chunk) + antenna; for a fixed q2 a half-wave stores 256 contiguous bytes.  Buffer
        // stores from the row of antenna 0 of this frame: one VGPR byte offset per thread (antenna 1's row is n_pts
        // rows further on), scalar offsets for q2 -- no per-store address arithmetic on the vector unit
        // rows [chunk][frame][antenna], hp = stream pairs per chunk (c counts pairs): both antennas of the pair side by
        // side, all antennas of a frame in one block for the X-engine
        const unsigned cu = (unsigned)c;
        const unsigned cr = hp == 1 ? cu : (hp == 2 ? cu >> 1 : (hp == 4 ? cu >> 2 : (unsigned)(((unsigned long long)cu * 0xAAAAAAABull) >> 33)));
        const int64_t row0 = ((int64_t)cr * n_pts + i) * (2 * hp) + 2 * (cu - cr * hp);
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(rows_raw + row0 * kN, 0, (int)(2 * kN * (int64_t)sizeof(cf)), 0x00020000);
        // the lane's bins 2m, 2m + 1 go out together (fx_fused4096.h::specpos): eight 16-byte stores, 512 contiguous
        // bytes per half-wave.  The offset of m stays in the VGPR: this library keeps wide buffer stores free of scalar
        // offsets (k_prepass.h, tests/test_isa_hazards.py)
        const unsigned voff0 = (unsigned)(((tid >> 5) & 1) * kN + specpos(lane_specpos(tid), 0)) * (unsigned)sizeof(cf);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            v4u32 d = {__float_as_uint(v[2 * m].x), __float_as_uint(v[2 * m].y), __float_as_uint(v[2 * m + 1].x),
                       __float_as_uint(v[2 * m + 1].y)};
            __builtin_amdgcn_raw_buffer_store_b128(d, rs, voff0 + (unsigned)(m * 512 * sizeof(cf)), 0, FXC_SPEC_STORE_AUX);
        }
    } else {
        // lanes 0-31 hold antenna 0, lanes 32-63 antenna 1 of the same bins: after the swap each lane has both
        // antennas for 8 of them (lanes 0-31: q2 = 0..7, lanes 32-63: q2 = 8..15) -- effex.py:520 without rot
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            cf a = v[q], b = v[q + 8];
            permlane32_swap(a, b);
            xacc(s, q, a, b);
        }
        FXC_STAMP(10);
    }
    if (DCK) {
        if (i + 1 == n_pts) {             // the chunk's last fetches are summed here, for the chunk-start branch of the next step
            dck_wait_own();
            dck_add(dck_stage, dck_acc, tid);
            dck_clear(dck_stage, tid);
        }
        u8.have_next = dck_next;          // (read by the next chunk's first step only)
    }
    // a raw row ends with the last frame of every `unit`-th chunk, of the last chunk and of this workgroup's
    // range: store this lane's 8 bins (fire and forget)
    const bool row_ends = !SPEC_OUT && range_walk_row_ends(pos);
    if (row_ends) {
        if (DCK) {
            // the same stores through a buffer descriptor of the row: a 64-bit lane pointer kept for this rare branch is
            // a loop invariant the DCK variant has no register for (it was spilled and reloaded here)
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(rows_raw + (int64_t)pos.row * kN, 0, (int)(kN * (int64_t)sizeof(cf)), 0x00020000);
            unsigned t = (unsigned)tid;
            asm volatile("" : "+v"(t));
#pragma unroll
            for (int q = 0; q < kAccPerThread; ++q) {
                const v2u32 d = {__float_as_uint(s.acc[q].x), __float_as_uint(s.acc[q].y)};
                __builtin_amdgcn_raw_buffer_store_b64(d, rs, t * (unsigned)sizeof(cf), (unsigned)(q * kThreads * sizeof(cf)), 0);
                s.acc[q] = fxc::mk(0.f, 0.f);
            }
        } else {
            cf* row = rows_raw + (int64_t)pos.row * kN + tid;
#pragma unroll
            for (int q = 0; q < kAccPerThread; ++q) {
                row[q * kThreads] = s.acc[q];
                s.acc[q] = fxc::mk(0.f, 0.f);
            }
        }
    }
    range_walk_advance(pos, row_ends);
}

// Work split and raw-row layout: fx_fused4096.h::RangeWalk (whole chunks dealt round-robin, then the last
// n_chunks % (workgroups * seg) chunks as equal frame ranges).
// SPEC_OUT == false: rows_raw = range_rows() raw rows, float32, slot order (fx_fused4096.h::slot_of_bin);
// rows_are_chunks: row c = chunk c (+ leading-part rows for tail chunks shared by several workgroups), else rows of
// `unit` chunks whose total is the integration.
// SPEC_OUT == true: rows_raw[(chunk * n_pts + i) * n_ant + stream][specpos] = the spectra themselves (`unit` = n_ant / 2).  A "chunk" here is a pair
// of consecutive antenna streams, so an even number of antennas [n_chunks][A][S] is simply n_chunks * A/2 pairs.
// U8: x points at interleaved uint8 I,Q ([chunk][antenna][num_samp] byte pairs) and dc[chunk * 2 + antenna] holds the
// conversion offsets (-mean_byte / 127.5, or -1 without DC removal) of each stream; DCK: of the first chunk of every
// workgroup's round-robin share and of the tail chunks only, the kernel sums the others itself.  stamps: diagnostic
// builds only.
template <bool SPEC_OUT, bool U8 = false, bool DCK = false>
__global__ __launch_bounds__(fxc::fused::kThreads, 2) void fx_fused4096_kernel(
    const cf* __restrict__ x, int64_t num_samp, int64_t n_pts, int64_t n_chunks, const f4* __restrict__ win_g,
    const cf* __restrict__ tw1_g, const cf* __restrict__ tw2_g, cf* __restrict__ rows_raw,
    unsigned long long* __restrict__ stamps, const cf* __restrict__ dc, int seg, int unit, int rows_are_chunks) {
    using namespace fxc::fused;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f4* win = reinterpret_cast<f4*>(smem + kLdsWin);
    cf* region = reinterpret_cast<cf*>(smem + kLdsRegion);
    cf* tw2 = reinterpret_cast<cf*>(smem + kLdsTw2);

    const int tid = threadIdx.x;
    const int ant = tid >> 8, j = tid & 255;
    for (int idx = tid; idx < kN; idx += kThreads) win[idx] = win_g[idx];
    if (tid < 256) tw2[tid] = tw2_g[tid];
    State s;
    state_load_twiddles(s, tw1_g, tid);
#pragma unroll
    for (int q = 0; q < kAccPerThread; ++q) s.acc[q] = fxc::mk(0.f, 0.f);
    __syncthreads();

    constexpr int64_t kSampleBytes = U8 ? sizeof(unsigned short) : sizeof(cf);
    const unsigned voff = (unsigned)((ant * num_samp + (255 - j)) * kSampleBytes);
    const unsigned chunk_bytes = (unsigned)(2 * num_samp * kSampleBytes);
    static_assert(!DCK || (U8 && !SPEC_OUT), "in-kernel DC removal: uint8 ingest only");
    U8State u8;
    u8.off = fxc::mk(0.f, 0.f);
    u8.have_next = u8.rounds = false;
    // DCK: landing area of the byte fetches and the threads' counters -- LDS objects of their own, so that the compiler
    // can tell the kernel's other LDS reads from reads of bytes still in flight -- and 64 bytes behind the carve
    __shared__ v4u32_t dck_stage_mem[DCK ? kDckStageVecs : 1];
    __shared__ unsigned dck_acc_mem[DCK ? 2 * kThreads : 1];
    v4u32_t* dck_stage = dck_stage_mem;
    unsigned* dck_acc = dck_acc_mem;
    unsigned* dck_red = reinterpret_cast<unsigned*>(smem + kLdsBytes);
    if (DCK) {
        dck_acc[2 * tid] = dck_acc[2 * tid + 1] = 0u;
        dck_clear(dck_stage, tid);
    }
    unsigned long long seg_t[kStampSegs] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_prev = 0;
    int64_t frames_done = 0;
#if FXC_STAMPS
    t_prev = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll 1
    for (int part = 0; part < 2; ++part) {   // 0: whole chunks, round-robin; 1: this workgroup's range of the tail
        const int walk_unit = SPEC_OUT ? 1 : unit;      // (SPEC_OUT: `unit` carries the stream pairs per chunk)
        int np_walk = (int)n_pts, seg_walk = seg, unit_walk = walk_unit;
        // DCK: the divisions of the split are done per part -- their reciprocals, hoisted over both parts, lived in scratch
        if (DCK) asm volatile("" : "+s"(np_walk), "+s"(seg_walk), "+s"(unit_walk));
        RangeWalk pos = part == 0 ? range_walk_rounds(blockIdx.x, gridDim.x, (int)n_chunks, np_walk, seg_walk, unit_walk, rows_are_chunks != 0)
                                  : range_walk_tail(blockIdx.x, gridDim.x, (int)n_chunks, np_walk, seg_walk, unit_walk, rows_are_chunks != 0);
        const int total = pos.left;
        if (!SPEC_OUT && part == 1 && (!pos.lead || total == 0)) {   // no leading part: its row reads as zeros
            const RangeSplit sp = range_split(gridDim.x, (int)n_chunks, seg, walk_unit, rows_are_chunks != 0);
            int t_lead = tid;
            if (DCK) asm volatile("" : "+v"(t_lead));     // (formed here: as an invariant of the part loop the pointer is spilled)
            cf* lead_row = rows_raw + (int64_t)(sp.rows_rounds + sp.n_tail + blockIdx.x) * kN + t_lead;
#pragma unroll
            for (int q = 0; q < kAccPerThread; ++q) lead_row[q * kThreads] = fxc::mk(0.f, 0.f);
        }
        if (total == 0) continue;
        if (U8) u8.off = dc[(int64_t)pos.c * 2 + __builtin_amdgcn_readfirstlane(ant)];
        u8.have_next = false;      // a part's first chunk and every tail chunk: offsets from dc[]
        u8.rounds = part == 0;
        // ring prologue: frame i -> slot 0, its history i-1, i-2, i-3 -> slots 3, 2, 1 (zeros before the chunk start)
        const cf* cbase = reinterpret_cast<const cf*>(reinterpret_cast<const char*>(x) + (int64_t)pos.c * 2 * num_samp * kSampleBytes);
#pragma unroll
        for (int d = 1; d < 4; ++d) {
            if (pos.i - d >= 0) {
                if (U8) {
                    load_frame_part_u8<0, 16>(s.h[4 - d], reinterpret_cast<const unsigned short*>(cbase), chunk_bytes, voff, pos.i - d);
                    convert_frame_u8(s.h[4 - d], u8.off);
                } else {
                    load_frame_part<0, 16>(s.h[4 - d], cbase, chunk_bytes, voff, pos.i - d);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) s.h[4 - d][r] = fxc::mk(0.f, 0.f);
            }
        }
        if (U8)
            load_frame_part_u8<0, 16>(s.h[0], reinterpret_cast<const unsigned short*>(cbase), chunk_bytes, voff, pos.i);
        else
            load_frame_part<0, 16>(s.h[0], cbase, chunk_bytes, voff, pos.i);
        // frame g of the part sits in ring slot g & 3: unrolled by four so the ring rotates by register renaming
#pragma unroll 1
        for (int g = 0; g < total; g += 4) {
            fused_step<0, SPEC_OUT, U8, DCK>(s, u8, dc, win, region, tw2, tid, x, num_samp, chunk_bytes, voff, pos, rows_raw, SPEC_OUT ? unit : 0, dck_stage, dck_acc, dck_red, seg_t, t_prev);
            if (g + 1 < total)
                fused_step<1, SPEC_OUT, U8, DCK>(s, u8, dc, win, region, tw2, tid, x, num_samp, chunk_bytes, voff, pos, rows_raw, SPEC_OUT ? unit : 0, dck_stage, dck_acc, dck_red, seg_t, t_prev);
            if (g + 2 < total)
                fused_step<2, SPEC_OUT, U8, DCK>(s, u8, dc, win, region, tw2, tid, x, num_samp, chunk_bytes, voff, pos, rows_raw, SPEC_OUT ? unit : 0, dck_stage, dck_acc, dck_red, seg_t, t_prev);
            if (g + 3 < total)
                fused_step<3, SPEC_OUT, U8, DCK>(s, u8, dc, win, region, tw2, tid, x, num_samp, chunk_bytes, voff, pos, rows_raw, SPEC_OUT ? unit : 0, dck_stage, dck_acc, dck_red, seg_t, t_prev);
        }
        frames_done += total;
    }
#if FXC_STAMPS
    if (stamps && (tid & 63) == 0) {
        unsigned long long* dst = stamps + ((int64_t)blockIdx.x * (kThreads / 64) + (tid >> 6)) * kStampSegs;
        for (int k = 0; k < kStampSegs; ++k) dst[k] = seg_t[k];
        dst[kStampSegs - 1] = (unsigned long long)frames_done;
    }
#else
    (void)stamps;
    (void)frames_done;
#endif
}

}  // namespace
