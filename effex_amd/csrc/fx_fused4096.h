// fx_fused4096.h — per-thread phases of the fused 2-antenna F+X kernel for nchan = 4096, ntaps = 4.
//
// One 512-thread workgroup (8 waves, one per CU) walks whole chunks; per spectrum i it
//   phase 1  (thread = antenna h, branch set {j + 256 r}):  4-tap PFB FIR over a ring of four frames
//            held in VGPRs (every IQ sample is fetched from HBM exactly once; the frame loop is
//            unrolled by four so the ring rotates by renaming, not by moves),
//            radix-16 over r, twiddle w4096^(j*k1), exchange 1 through LDS        [s_barrier]
//   phase 2  (lane = antenna, k1, j0):  radix-16 over j1, twiddle w256^(j0*q1), exchange 2 —
//            a 16x16 transpose that stays inside one wave (no s_barrier)
//   phase 3  (lane = antenna, k1, q1):  radix-16 over j0 -> bins k = k1 + 16 q1 + 256 q2;
//            v_permlane32_swap pairs antenna 0 (lanes 0-31) with antenna 1 (lanes 32-63) so every
//            lane multiplies-and-accumulates 8 bins of  spec0 * conj(spec1).
// FFT decomposition (kernel exp(+2 pi i m k / 4096)):
//   m = j + 256 r,  j = j0 + 16 j1,  k = k1 + 16 q1 + 256 q2
//   w^(mk) = w16^(r k1) * w4096^(j k1) * w16^(j1 q1) * w256^(j0 q1) * w16^(j0 q2)
//
// The same source is compiled by g++ in tests/emul (host emulation of the index logic; test
// infrastructure only — the shipped library contains only the device build).
#pragma once
#include "fx_math.h"

#if defined(__HIP_DEVICE_COMPILE__)
#define FXC_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define FXC_SCHED_FENCE() ((void)0)
#endif

namespace fxc {
namespace fused {

// One 8-byte LDS read that the compiler will not pair into ds_read2_b64 (half the LDS rate of
// ds_read_b64 on gfx950, MI355X_MICROARCH.md §LDS).
FX_HD cf lds_load(const cf* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const volatile __attribute__((address_space(3))) unsigned long long* lds_u64_ptr;
    const unsigned long long u = *(lds_u64_ptr)(p);
    cf r;
    r.x = __uint_as_float((unsigned)(u & 0xffffffffull));
    r.y = __uint_as_float((unsigned)(u >> 32));
    return r;
#else
    return *p;
#endif
}

constexpr int kN = 4096;
constexpr int kT = 4;
constexpr int kThreads = 512;
constexpr int kRowPitch = 272;              // cf per k1 row; == 16 (mod 32) keeps ds_read_b64 conflict-free
constexpr int kRegion = 16 * kRowPitch;     // cf per antenna
constexpr int kAccPerThread = 8;
#ifndef FXC_FIR_GROUP
#define FXC_FIR_GROUP 4
#endif
constexpr int kFirGroup = FXC_FIR_GROUP;    // branches per software-pipeline group of the FIR's window reads

// LDS carve (bytes); every offset is a multiple of 16
constexpr int kLdsWin = 0;                                   // f4[4096]   window, [r*256 + j] = h[t*N + j + 256 r], t = x,y,z,w
constexpr int kLdsRegion = kLdsWin + kN * 16;                // cf[2][kRegion]
constexpr int kLdsTw2 = kLdsRegion + 2 * kRegion * 8;        // cf[256]    w256^(j0*q1) at [q1*16 + j0]
constexpr int kLdsBytes = kLdsTw2 + 256 * 8;

struct State {
    // ring of four frames of this thread's 16 branch samples: slot PH holds the frame being
    // channelised, the other three the PFB history (and, once dead, the prefetch of the next frame)
    cf h[4][16];
    cf tw1[16];                  // w4096^(j*k1), k1 = 1..15 ([0] unused): 15 twiddles in 30 VGPRs beat 6 stored
                                 // powers + 9 extra complex multiplies (the kernel is limited by energy per spectrum)
    cf acc[kAccPerThread];       // sum_i spec0*conj(spec1) for this lane's 8 bins
};

// zero PFB history at the start of every chunk; the chunk's frame 0 sits in slot PH
template <int PH>
FX_HD void state_reset_history(State& s) {
#pragma unroll
    for (int d = 1; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) s.h[(PH + d) & 3][r] = mk(0.f, 0.f);
}

FX_HD void state_reset_all(State& s) {
    state_reset_history<0>(s);
#pragma unroll
    for (int q = 0; q < kAccPerThread; ++q) s.acc[q] = mk(0.f, 0.f);
}

// Work split of one launch (fxcorr.hip::fx_fused4096_kernel) over g workgroups, in two parts:
//   rounds  the first n_full = rounds * g * seg chunks go out whole, in segments of `seg` consecutive chunks dealt
//           round-robin (workgroup b: segments b, b + g, b + 2g, ...): at any moment the g workgroups stream from one
//           compact window of g * seg chunks and start their chunks together (measured on 10 000 chunk pairs:
//           seg = 1 8.82 ms, seg = 39 8.94 ms, everything as frame ranges 8.92 ms);
//   tail    the remaining n_chunks - n_full chunks form one sequence of frames and workgroup b takes the contiguous
//           range [range_begin(b), range_begin(b + 1)) of it, chunk boundaries or not, so that every workgroup gets
//           the same number of frames (+-1) whatever n_chunks % g is.  A range that starts inside a chunk reloads the
//           (up to) three frames of PFB history before it.
// Raw rows (float32 sums of spectrum products): a row collects `unit` chunks finished in a row by one workgroup.
//   rows_are_chunks (unit = 1): row c = chunk c.  A tail chunk shared by several workgroups: the one that owns its
//   first frame writes row c, every other one its share into its leading-part row n_chunks + b (zeros if none).
//   otherwise (integration only): rounds part row b + g * j for workgroup b's j-th row, then one row per tail chunk
//   and the g leading-part rows; the sum of all range_rows() rows is the integration.
struct RangeWalk {
    int c, i;            // chunk and frame of the spectrum being computed
    int left;            // frames of this part still to do, this one included
    int row;             // raw row the sums in progress go to
    int n_pts;
    int seg, seg_left;   // chunks per segment; chunks of the current segment still to finish, this one included
    int seg_jump;        // chunks skipped at a segment end: (g - 1) * seg
    int unit, in_row;    // whole chunks per row; chunks finished in the row in progress
    int row_step;        // rows that are not chunks: next row = row + row_step
    int row_off;         // rows that are chunks: row = c + row_off
    bool rows_are_chunks;
    bool lead;                 // tail: the range starts inside a chunk
};

struct RangeSplit {
    int rounds, n_full, n_tail, rows_rounds, n_rows;   // n_rows: rows in all, leading-part rows included
};

FX_HD RangeSplit range_split(int g, int n_chunks, int seg, int unit, bool rows_are_chunks) {
    RangeSplit r;
    r.rounds = n_chunks / (g * seg);
    r.n_full = r.rounds * g * seg;
    r.n_tail = n_chunks - r.n_full;
    r.rows_rounds = rows_are_chunks ? r.n_full : g * ((r.rounds * seg + unit - 1) / unit);
    r.n_rows = r.rows_rounds + r.n_tail + g;
    return r;
}

FX_HD RangeWalk range_walk_rounds(int b, int g, int n_chunks, int n_pts, int seg,
                                   int unit, bool rows_are_chunks) {
    const RangeSplit sp = range_split(g, n_chunks, seg, unit, rows_are_chunks);
    RangeWalk w;
    w.c = b * seg;
    w.i = 0;
    w.left = sp.rounds * seg * n_pts;
    w.n_pts = n_pts;
    w.seg = w.seg_left = seg;
    w.seg_jump = (g - 1) * seg;
    w.unit = rows_are_chunks ? 1 : unit;
    w.in_row = 0;
    w.rows_are_chunks = rows_are_chunks;
    w.row_off = 0;
    w.row_step = g;
    w.row = rows_are_chunks ? w.c : b;
    w.lead = false;
    return w;
}

FX_HD RangeWalk range_walk_tail(int b, int g, int n_chunks, int n_pts, int seg,
                                 int unit, bool rows_are_chunks) {
    const RangeSplit sp = range_split(g, n_chunks, seg, unit, rows_are_chunks);
    RangeWalk w;
    const int n_frames = sp.n_tail * n_pts;
    const int f0 = range_begin(b, n_frames, g);
    w.left = range_begin(b + 1, n_frames, g) - f0;
    w.c = sp.n_full + f0 / n_pts;
    w.i = f0 % n_pts;
    w.n_pts = n_pts;
    w.seg = 1;
    w.seg_left = 1;
    w.seg_jump = 0;
    w.unit = 1;
    w.in_row = 0;
    w.rows_are_chunks = true;
    w.row_off = sp.rows_rounds - sp.n_full;
    w.row_step = 0;
    w.lead = w.i != 0;
    w.row = w.lead ? sp.rows_rounds + sp.n_tail + b : w.c + w.row_off;
    return w;
}

// the spectrum of (c, i) is the last one of the row in progress: last frame of the part, or of the row's last chunk
FX_HD bool range_walk_row_ends(const RangeWalk& w) {
    return w.left == 1 || (w.i + 1 == w.n_pts && w.in_row + 1 == w.unit);
}

// the frame to fetch while (c, i) is computed: the next one of the part (next frame of the chunk or frame 0 of the
// next chunk); at the very end of the part the current frame again (never used)
FX_HD void range_walk_prefetch(const RangeWalk& w, int& pc, int& pi) {
    pc = w.c;
    pi = w.i;
    if (w.left > 1 && ++pi == w.n_pts) {
        pi = 0;
        pc += 1 + (w.seg_left == 1 ? w.seg_jump : 0);
    }
}

// move on to the next frame of the part; call after the row store when range_walk_row_ends()
FX_HD void range_walk_advance(RangeWalk& w, bool row_ended) {
    if (++w.i == w.n_pts) {
        w.i = 0;
        w.c += 1;
        if (--w.seg_left == 0) {
            w.c += w.seg_jump;
            w.seg_left = w.seg;
        }
        w.in_row += 1;
    }
    w.left -= 1;
    if (row_ended) {
        w.in_row = 0;
        w.row = w.rows_are_chunks ? w.c + w.row_off : w.row + w.row_step;
    }
}

// element offset (in cf) inside one frame of the sample this thread feeds to branch j + 256 r
FX_HD int sample_offset(int j, int r) { return (kN - 1) - j - 256 * r; }

// phase 1a for a frame in ring slot PH: 4-tap FIR (taps t = 0..3 on frames i, i-1, i-2, i-3, summed
// in that order); result left in v[r]
template <int PH>
FX_HD void phase1_fir(const State& s, const f4* win, int tid, cf (&v)[16]) {
    const int j = tid & 255;
    const cf (&x0)[16] = s.h[PH];
    const cf (&x1)[16] = s.h[(PH + 3) & 3];
    const cf (&x2)[16] = s.h[(PH + 2) & 3];
    const cf (&x3)[16] = s.h[(PH + 1) & 3];
    // software-pipelined in groups of kFirGroup branches: the window quads of group g + 1 are requested from LDS
    // before group g is computed, so only the first ds_read latency is exposed (2 * kFirGroup * 4 VGPRs of quads in
    // flight at the kernel's point of highest register pressure)
    constexpr int G = kFirGroup, NG = 16 / G;
    f4 w[2][G];
#pragma unroll
    for (int q = 0; q < G; ++q) w[0][q] = win[q * 256 + j];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (g < NG - 1) {
#pragma unroll
            for (int q = 0; q < G; ++q) w[(g + 1) & 1][q] = win[(G * (g + 1) + q) * 256 + j];
        }
        FXC_SCHED_FENCE();
#pragma unroll
        for (int q = 0; q < G; ++q) {
            const int r = G * g + q;
            const f4 t = w[g & 1][q];
            cf a = cscale(x0[r], t.x);
            a = cfma(t.y, x1[r], a);
            a = cfma(t.z, x2[r], a);
            v[r] = cfma(t.w, x3[r], a);
        }
        FXC_SCHED_FENCE();
    }
}

// phase 1b: the second half of the radix-16 over r (dft16_b) with each output twiddled by w4096^(j*k1) and stored to
// exchange 1 as soon as it exists -- the stores are bound by the LDS write path (64 KiB at ~85 B/clk per CU), and the
// 72 + 60 vector instructions of the butterflies and twiddles run in its shadow instead of in front of barrier B0.
// Call after dft16_a(v).
FX_HD void phase1_finish_store(const State& s, cf (&v)[16], cf* region, int tid) {
    cf* mine = region + (tid >> 8) * kRegion + (tid & 255);
    dft16_b_stream(v, [&](int k1, cf val) {
        if (k1 > 0) val = cmul(val, s.tw1[k1]);
        mine[k1 * kRowPitch] = val;
    });
}

// load this thread's twiddles from the [16][256] table w4096^(j*k1)
FX_HD void state_load_twiddles(State& s, const cf* tw1_table, int tid) {
    const int j = tid & 255;
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) s.tw1[k1] = tw1_table[k1 * 256 + j];
}

// this lane's row for phases 2 and 3
FX_HD cf* lane_row(cf* region, int tid) {
    const int l = tid & 63, wave = tid >> 6;
    const int ant = l >> 5, k1 = 2 * wave + ((l >> 4) & 1);
    return region + ant * kRegion + k1 * kRowPitch;
}

FX_HD void phase2_load(cf* region, int tid, cf (&v)[16]) {
    const cf* row = lane_row(region, tid);
    const int j0 = tid & 15;
#pragma unroll
    for (int j1 = 0; j1 < 16; ++j1) v[j1] = lds_load(row + j0 + 16 * j1);
}

FX_HD void phase2_twiddle(cf (&v)[16], const cf* tw2, int tid) {
    const int j0 = tid & 15;
    // same pipelining for the w256^(j0*q1) table: group g + 1 is in flight while group g multiplies
    cf t[2][4];
#pragma unroll
    for (int q = 1; q < 4; ++q) t[0][q] = lds_load(tw2 + q * 16 + j0);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (g < 3) {
#pragma unroll
            for (int q = 0; q < 4; ++q) t[(g + 1) & 1][q] = lds_load(tw2 + (4 * (g + 1) + q) * 16 + j0);
        }
        FXC_SCHED_FENCE();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int q1 = 4 * g + q;
            if (q1 > 0) v[q1] = cmul(v[q1], t[g & 1][q]);
        }
        FXC_SCHED_FENCE();
    }
}

FX_HD void phase2_store(const cf (&v)[16], cf* region, int tid) {
    cf* row = lane_row(region, tid);
    const int j0 = tid & 15;
#pragma unroll
    for (int q1 = 0; q1 < 16; ++q1) row[q1 * 17 + j0] = v[q1];
}

FX_HD void phase3_load(cf* region, int tid, cf (&v)[16]) {
    const cf* row = lane_row(region, tid);
    const int q1 = tid & 15;
#pragma unroll
    for (int j0 = 0; j0 < 16; ++j0) v[j0] = lds_load(row + q1 * 17 + j0);
}

// X-stage on paired data: a = antenna 0, b = antenna 1 for this lane's bin q (lanes 0-31) or
// q + 8 (lanes 32-63) — effex/effex.py:520 without rot (applied once at finalize)
FX_HD void xacc(State& s, int q, cf a, cf b) { s.acc[q] = cmulc_acc(s.acc[q], a, b); }

// natural bin index of accumulator q of thread tid
FX_HD int bin_of(int tid, int q) {
    const int l = tid & 63, wave = tid >> 6;
    const int k1 = 2 * wave + ((l >> 4) & 1), q1 = l & 15, q2 = q + 8 * (l >> 5);
    return k1 + 16 * q1 + 256 * q2;
}

// inverse: where bin k lives in the [q*512 + tid] slot order the kernel stores partial sums in
FX_HD int slot_of_bin(int k) {
    const int k1 = k & 15, q1 = (k >> 4) & 15, q2 = k >> 8;
    const int l = (q2 >> 3) * 32 + (k1 & 1) * 16 + q1;
    const int tid = (k1 >> 1) * 64 + l;
    return (q2 & 7) * kThreads + tid;
}

// this lane's index L among the 256 (k1, q1) combinations of phase 3
FX_HD int lane_specpos(int tid) {
    const int l = tid & 63, wave = tid >> 6;
    return (2 * wave + ((l >> 4) & 1)) * 16 + (l & 15);
}

// position inside a spectrum row written by the F-only variant of the kernel (multi-antenna path) of the lane's bin
// q2: the lane's bins 2m and 2m + 1 sit side by side, so one 16-byte store takes both and the 32 lanes of a half-wave
// fill 512 consecutive bytes
FX_HD int specpos(int lane_l, int q2) { return (q2 >> 1) * 512 + 2 * lane_l + (q2 & 1); }
FX_HD int specpos_of_bin(int k) { return specpos((k & 15) * 16 + ((k >> 4) & 15), k >> 8); }

}  // namespace fused
}  // namespace fxc
