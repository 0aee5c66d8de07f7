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
#ifndef FXC_LOAD_AUX
#define FXC_LOAD_AUX 0   // cache policy of the IQ stream loads: bit 0 sc0, bit 1 nt, bit 4 sc1
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
            (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rs, voff, soff + (unsigned)(256 * (15 - r) * sizeof(unsigned short)), 0));
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
};

template <int PH, bool SPEC_OUT, bool U8>
__device__ __forceinline__ void fused_step(fxc::fused::State& s, U8State& u8, const cf* dc, const f4* win, cf* region,
                                           const cf* tw2, int tid, const cf* x, int64_t num_samp, unsigned chunk_bytes,
                                           unsigned voff, fxc::fused::RangeWalk& pos, cf* rows_raw, int hp,
                                           unsigned long long (&seg)[kStampSegs], unsigned long long& t_prev) {
    using namespace fxc::fused;
    const int64_t c = pos.c, i = pos.i, n_pts = pos.n_pts;   // (the walk itself is 32-bit: scalar registers are scarce)
    FXC_STAMP(0);    // loop overhead and the (rare) row store since the previous step's last stamp
    if (i == 0) {    // zero PFB history at the start of every chunk
        asm volatile("" ::: "memory");   // keep this a (rarely taken) uniform branch, not 96 v_cndmask per frame
        state_reset_history<PH>(s);
        if (U8) u8.off = dc[c * 2 + ((tid >> 8) & 1)];
    }
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
        // bytes per half-wave.  The offset of m stays in the VGPR: gfx950 needs wait states between a store of more
        // than 8 bytes with an SGPR offset and a VALU write of its data registers that the compiler omits
        // (tests/test_isa_hazards.py)
        const unsigned voff0 = (unsigned)(((tid >> 5) & 1) * kN + specpos(lane_specpos(tid), 0)) * (unsigned)sizeof(cf);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            v4u32 d = {__float_as_uint(v[2 * m].x), __float_as_uint(v[2 * m].y), __float_as_uint(v[2 * m + 1].x),
                       __float_as_uint(v[2 * m + 1].y)};
            __builtin_amdgcn_raw_buffer_store_b128(d, rs, voff0 + (unsigned)(m * 512 * sizeof(cf)), 0, 0);
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
    // a raw row ends with the last frame of every `unit`-th chunk, of the last chunk and of this workgroup's
    // range: store this lane's 8 bins (fire and forget)
    const bool row_ends = !SPEC_OUT && range_walk_row_ends(pos);
    if (row_ends) {
        cf* row = rows_raw + (int64_t)pos.row * kN + tid;
#pragma unroll
        for (int q = 0; q < kAccPerThread; ++q) {
            row[q * kThreads] = s.acc[q];
            s.acc[q] = fxc::mk(0.f, 0.f);
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
// conversion offsets (-mean_byte / 127.5, or -1 without DC removal) of each stream.  stamps: diagnostic builds only.
template <bool SPEC_OUT, bool U8 = false>
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
    U8State u8;
    u8.off = fxc::mk(0.f, 0.f);
    unsigned long long seg_t[kStampSegs] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_prev = 0;
    int64_t frames_done = 0;
#if FXC_STAMPS
    t_prev = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll 1
    for (int part = 0; part < 2; ++part) {   // 0: whole chunks, round-robin; 1: this workgroup's range of the tail
        const int walk_unit = SPEC_OUT ? 1 : unit;      // (SPEC_OUT: `unit` carries the stream pairs per chunk)
        RangeWalk pos = part == 0 ? range_walk_rounds(blockIdx.x, gridDim.x, (int)n_chunks, (int)n_pts, seg, walk_unit, rows_are_chunks != 0)
                                  : range_walk_tail(blockIdx.x, gridDim.x, (int)n_chunks, (int)n_pts, seg, walk_unit, rows_are_chunks != 0);
        const int total = pos.left;
        if (!SPEC_OUT && part == 1 && (!pos.lead || total == 0)) {   // no leading part: its row reads as zeros
            const RangeSplit sp = range_split(gridDim.x, (int)n_chunks, seg, walk_unit, rows_are_chunks != 0);
            cf* lead_row = rows_raw + (int64_t)(sp.rows_rounds + sp.n_tail + blockIdx.x) * kN + tid;
#pragma unroll
            for (int q = 0; q < kAccPerThread; ++q) lead_row[q * kThreads] = fxc::mk(0.f, 0.f);
        }
        if (total == 0) continue;
        if (U8) u8.off = dc[(int64_t)pos.c * 2 + ant];
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
            fused_step<0, SPEC_OUT, U8>(s, u8, dc, win, region, tw2, tid, x, num_samp, chunk_bytes, voff, pos, rows_raw, SPEC_OUT ? unit : 0, seg_t, t_prev);
            if (g + 1 < total)
                fused_step<1, SPEC_OUT, U8>(s, u8, dc, win, region, tw2, tid, x, num_samp, chunk_bytes, voff, pos, rows_raw, SPEC_OUT ? unit : 0, seg_t, t_prev);
            if (g + 2 < total)
                fused_step<2, SPEC_OUT, U8>(s, u8, dc, win, region, tw2, tid, x, num_samp, chunk_bytes, voff, pos, rows_raw, SPEC_OUT ? unit : 0, seg_t, t_prev);
            if (g + 3 < total)
                fused_step<3, SPEC_OUT, U8>(s, u8, dc, win, region, tw2, tid, x, num_samp, chunk_bytes, voff, pos, rows_raw, SPEC_OUT ? unit : 0, seg_t, t_prev);
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
