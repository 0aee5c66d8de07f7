// fx_spec.h -- the F+X kernel for two antennas at a channel count that is not a power of two (`--resolution` is a free
// integer in the reference, effex/effex.py:733-739), SPECIALISED for one channel count at run time: libfxcorr hands
// this file to hiprtc with the shape as -D options (h_rtc.h), the way FFT libraries build their kernels.  Same
// arithmetic as the any-shape kernel of k_generic.h (polyphase FIR of SURVEY.md 2.3, the Stockham autosort stages of
// fx_mixed.h with the inter-stage twiddles of the same float64-rounded table, s0 conj(s1) summed over the run of frames), but with every
// stride, stage and trip count a compile-time constant, the roots of unity of the butterflies literals, and composite radices run in registers:
//
//   * every sample is fetched ONCE: a thread keeps the last T - 1 frames' samples of its own points in registers (the
//     ring), frame f + 1 is in flight while frame f goes through its stages; the window taps of those points and the
//     inter-stage twiddles of the thread's butterflies are loaded once per launch and live in registers too;
//   * the FIR feeds the first butterfly straight from registers (a thread's points ARE the inputs of its stage-0
//     butterflies: m = b + r N/R0), and the last butterfly feeds the X stage straight from registers (both antennas'
//     outputs for the same bins are in the same thread): S - 1 trips through LDS instead of S + 1, S - 1 barriers;
//   * LDS addresses are one base register plus immediates; the only index arithmetic left inside the frame loop is the
//     frame pointer's increment.
//
// Options (all required): FXM_N channels, FXM_T taps (1..4), FXM_TPR threads per slot (a slot = the threads that carry
// one frame pair), FXM_SLOTS slots per workgroup (each with its own contiguous run of frames), FXM_NST stages,
// FXM_RADICES their radices (comma list; product = N; primes up to 23, 4, and composites of them -- 6, 8, 9, 10, 12, 15, 16, 20,
// 25 ... -- which run as Good-Thomas / Cooley-Tukey butterflies in registers: one trip through LDS instead of two or three),
// FXM_U8 (1: samples are the receivers' interleaved unsigned bytes, converted as pyrtlsdr does behind effex.py:652 with the
// per-stream offset of k_conditioning.h).
// Optional (h_rtc.h::spec_layout chooses them): FXM_GROUPS rows per work item of each stage (comma list; a stage's items are
// (butterfly, group of rows) pairs dealt over the slot's threads, so that a stage with few butterflies -- a large radix -- still
// fills the lanes), FXM_PLANE0 / FXM_PADS where the stages put their outputs in LDS (stores and loads free of bank conflicts),
// FXM_TWFULL the largest radix whose R - 1 twiddles all stay in registers (beyond: the first, and its powers every step).
//
// Compiles as device code under hiprtc / hipcc and as plain C++ under g++: tests/emul/emul_spec.cpp runs the same body
// with one host thread per GPU thread and a real barrier (test infrastructure only).
#pragma once

#include "fx_mixed.h"

#ifndef FXM_FONLY
#define FXM_FONLY 0      // 1: the F stage alone -- a slot's two rows are two STREAMS (2 pair, 2 pair + 1) and the last butterfly's outputs are
                         // the spectra, stored in natural order; replaces cusignal's channelize_poly + .T (effex.py:553) at any channel count
#endif
// FXM_XM 1 (with FXM_FONLY 1, FXM_ROWS 1): the second pass of two antennas above 4096 channels (a thread cannot hold sixteen points of two
// antennas): antenna 1 through the stages, and instead of storing its spectra the last butterfly loads antenna 0's values of the same bins
// (written by an F-only launch before: Args::spec0, [stream][frame][N], natural order), multiplies and adds to the thread's sums -- the X stage
// of effex.py:516-521 with antenna 0's spectra through HBM once (2 x the algorithmic bytes; spectra of both + xmul_kernel: 3 x)
#ifndef FXM_XM
#define FXM_XM 0
#endif
#ifndef FXM_U
#define FXM_U 1          // frames a slot carries through every step together (2: half the barriers per frame, twice the work in flight)
#endif
#if !defined(FXM_N) || !defined(FXM_T) || !defined(FXM_TPR) || !defined(FXM_SLOTS) || !defined(FXM_NST) || !defined(FXM_RADICES) || !defined(FXM_U8)
#error "fx_spec.h is compiled per shape: -DFXM_N= -DFXM_T= -DFXM_TPR= -DFXM_SLOTS= -DFXM_NST= -DFXM_RADICES= -DFXM_U8="
#endif

// FXM_ABL: developer-only timing ablations (WRONG RESULTS by design; plans are built with 0 unless FXC_RTC_ABL says otherwise):
//   1 no barriers   2 stage outputs not stored to LDS   4 butterflies skipped (inputs passed through)   8 samples not loaded
//   16 every load reads the chunk's first frames (cache hits: the memory system out of the picture, same instruction stream)
#ifndef FXM_ABL
#define FXM_ABL 0
#endif
// FXM_LEAN 1: the build for frames of more than 2048 channels (a thread carries up to eight points of each antenna: 128
// registers of ring) and for shapes with a prime factor of 17 ... 23 (the butterfly's registers) -- nothing but the ring and the sums stays in registers from step to step.  The window taps come from L2
// every step (Args::h4: the four taps of a point in one 16-byte load); of a butterfly's twiddles only the first is fetched
// (Args::tw1, a table by stage, item and thread -- consecutive lanes read consecutive entries -- requested at the top of the
// step, before the next frames' samples: Body::step_lean), the others are its powers (a few complex multiplies a stage, shared by the rows of the step); the output offsets
// are recomputed.
#ifndef FXM_LEAN
#define FXM_LEAN 0
#endif
// FXM_ROWS 1 (F only): a workgroup carries ONE stream instead of a pair -- a thread then has room for sixteen points of it (128
// registers of ring): the F stage of 4097 ... 8192 channels.
#ifndef FXM_ROWS
#define FXM_ROWS 2
#endif
#ifndef FXM_PLANE0
#define FXM_PLANE0 0     // > 0: the first stage stores output q of butterfly b at q PLANE0 + b (consecutive lanes, consecutive addresses)
#endif
#ifndef FXM_TWFULL
#define FXM_TWFULL 13
#endif
#ifndef FXM_OOB_ZERO
#define FXM_OOB_ZERO 1   // complex64 samples that do not exist are read as the zeros a buffer load returns beyond its records (0: a branch around the load)
#endif

namespace fxm {

using fxc::cf;
using fxc::pk;
using fxc::pk2;
using fxc::pk_cmul;
using fxc::pk_fma;
using fxc::pk_splat;
using fxc::unpk;

constexpr int N = FXM_N;
constexpr int T = FXM_T;
constexpr int TPR = FXM_TPR;
constexpr int SLOTS = FXM_SLOTS;
constexpr int S = FXM_NST;
constexpr int kRadix[S] = {FXM_RADICES};
constexpr bool U8 = FXM_U8 != 0;
constexpr int U = FXM_U;
constexpr bool FONLY = FXM_FONLY != 0;
constexpr bool XM = FXM_XM != 0;                 // the F stage of ONE stream whose spectra are multiplied by another stream's (Args::spec0) and summed
constexpr bool LEAN = FXM_LEAN != 0;
constexpr int NA = FXM_ROWS;                     // streams a workgroup carries through a step: the two antennas, or (F only) one or two streams
constexpr int THREADS = TPR * SLOTS;
// The ring: the frames a step needs -- its own U and the T - 1 before them -- in NS = T + U - 1 slots, frame g of a run in
// slot g mod NS; the next step's U frames land in the U slots the FIR has just finished with.  The slot pattern repeats after
// UNR steps, and the step loop is unrolled that far so that every slot index is a constant.
constexpr int NS = T + U - 1;
constexpr int gcd_of(int a, int b) { return b == 0 ? a : gcd_of(b, a % b); }
constexpr int UNR = NS / gcd_of(NS, U);
constexpr int ROWS = NA * U;                     // rows a step carries: (frame u, antenna a) -> u * NA + a

constexpr int ns_of(int s) {
    int v = 1;
    for (int i = 0; i < s; ++i) v *= kRadix[i];
    return v;
}
constexpr int nb_of(int s) { return N / kRadix[s]; }
// Work items: stage 0's are its butterflies (a thread runs the FIR of its butterflies' points for every row); a later stage's are
// (butterfly b, group g of grp(s) rows) pairs, item i = g nb + b, dealt over the slot's threads as i = lt + j TPR.  The last stage of
// the F + X build keeps both antennas of a frame in one item (the X product is formed in the thread's registers).
#ifdef FXM_GROUPS
constexpr int kGroupIn[S] = {FXM_GROUPS};
constexpr int grp(int s) { return (s == 0 || kGroupIn[s] <= 0) ? ROWS : kGroupIn[s]; }
#else
constexpr int grp(int) { return ROWS; }
#endif
#ifdef FXM_PADS
constexpr int kPadIn[S] = {FXM_PADS};
constexpr int pad_of(int s) { return kPadIn[s]; }
#else
constexpr int pad_of(int) { return 0; }
#endif
constexpr int ngrp(int s) { return ROWS / grp(s); }
constexpr int items_of(int s) { return nb_of(s) * ngrp(s); }
constexpr int j_of(int s) { return (items_of(s) + TPR - 1) / TPR; }     // items of stage s per thread
constexpr bool full_of(int s) { return j_of(s) * TPR == items_of(s); }  // ... and every thread has all of them
constexpr int tw_per(int s) { return (LEAN || kRadix[s] > FXM_TWFULL) ? 1 : kRadix[s] - 1; }   // twiddle registers of one item of stage s
constexpr int tw_base(int s) {                                          // first twiddle register of stage s (s >= 1)
    int c = 0;
    for (int i = 1; i < s; ++i) c += j_of(i) * tw_per(i);
    return c;
}
constexpr int ob_base(int s) {                                          // first offset register of stage s (1 <= s <= S-1)
    int c = 0;
    for (int i = 1; i < s; ++i) c += j_of(i);
    return c;
}
constexpr int R0 = kRadix[0], J0 = j_of(0), PTS = R0 * J0;              // a thread's points: m = lt + j TPR + r N/R0
constexpr int RL = kRadix[S - 1], JL = j_of(S - 1);
constexpr int TWC = LEAN ? 0 : tw_base(S), OBC = LEAN ? 0 : ob_base(S > 1 ? S - 1 : 1), IBC = LEAN ? 0 : ob_base(S);
constexpr int TW1C = tw_base(S);                 // LEAN: rows of Args::tw1, [TW1C][TPR]
constexpr bool SWAP = S >= 2 && S % 2 == 0;      // the last stage reads the buffer the next frame's first stage writes: alternate them

// Where a stage's outputs stand in LDS.  The buffer stage s writes (stage s + 1 reads it) holds the Stockham order x = (b - k) R + k +
// q ns of fx_mixed.h in blocks of ns_of(s + 1) elements -- the outputs of the ns butterflies that share b - k -- with pad_of(s) unused
// elements behind each block: a lane group's runs of ns consecutive elements then fall on different banks.  The first stage's buffer
// may instead be PLANES: output q of butterfly b at q P0 + b -- a wave's stores of one q are consecutive, and the R0 runs a wave of the
// second stage reads (its butterflies' inputs r are nb / R0 apart within one plane) tile the banks when P0 is chosen for it.  Either way a
// thread's addresses are one base register plus immediates.
constexpr int P0 = FXM_PLANE0;
constexpr int blk_of(int s) { return ns_of(s + 1); }
constexpr int len_of(int s) { return (s == 0 && P0 > 0) ? R0 * P0 : N + (N / blk_of(s)) * pad_of(s); }
constexpr int row_stride() {
    int m = N;
    for (int s = 0; s + 1 < S; ++s) m = len_of(s) > m ? len_of(s) : m;
    return m;
}
constexpr int RS = row_stride();                 // one row of a stage buffer; a slot's LDS: [buffer X | Y][row][RS]
constexpr int rd_stride(int s) {                 // between inputs r and r + 1 of a butterfly of stage s >= 1 (blk_of(s - 1) and R0 divide nb_of(s))
    return (s == 1 && P0 > 0) ? nb_of(s) / R0 : nb_of(s) + (nb_of(s) / blk_of(s - 1)) * pad_of(s - 1);
}
constexpr int rd_pos(int s, int b) {             // ... and where input 0 of its butterfly b stands
    return (s == 1 && P0 > 0) ? (b % R0) * P0 + b / R0 : b + (b / blk_of(s - 1)) * pad_of(s - 1);
}
constexpr int wr_pos(int s, int b) {             // where butterfly b of stage s (1 <= s <= S - 2) puts its output 0; output q stands q ns_of(s) further
    return (b / ns_of(s)) * (blk_of(s) + pad_of(s)) + b % ns_of(s);
}
constexpr int item_bfly_of(int s, int i) { return i >= items_of(s) ? 0 : i % nb_of(s); }      // the butterfly of item i of stage s (0 where there is none)
constexpr bool plain_reads(int s) { return ngrp(s) == 1 && !(s == 1 && P0 > 0) && pad_of(s - 1) == 0; }     // item j's input 0 stands at lt + j TPR
constexpr int LDS_PER_SLOT = S >= 2 ? 2 * ROWS * RS : 0; // complex64 elements

static_assert(ns_of(S) == N, "the radices multiply to N");
static_assert(T >= 1 && T <= 4, "one to four taps (the ring lives in registers)");
static_assert(U == 1 || U == 2, "one or two frames per step");
static_assert(NA == 2 || (NA == 1 && FONLY), "one stream per workgroup: the F stage alone");
static_assert(!(FONLY && U8), "the byte ingest is the two-antenna kernel's");
static_assert(!XM || (FONLY && NA == 1), "the second pass carries one stream per workgroup");
constexpr bool SUMS = !FONLY || XM;              // the thread keeps sums of products (else the last butterfly stores spectra)
static_assert(SLOTS >= 1 && (SLOTS == 1 || TPR % 64 == 0 || 64 % TPR == 0), "slots do not straddle waves");
static_assert(FONLY || S == 1 || grp(S - 1) % NA == 0, "the last stage's items hold both antennas of a frame");
static_assert(P0 == 0 || (S >= 2 && P0 >= nb_of(0)), "planes hold the first stage's butterflies");

// per-thread state, all of it registers once the loops below are unrolled
struct Thread {
    pk2 ring[NA][PTS][NS];            // frame g's samples of the thread's points in slot g mod NS (g counted from the run's first frame)
    pk2 tw[TWC > 0 ? TWC : 1];       // twiddles of the thread's items in stages 1 .. S-1
    int ob[OBC > 0 ? OBC : 1];       // where the items of stages 1 .. S-2 put their outputs
    int ib[IBC > 0 ? IBC : 1];       // where the items of stages 1 .. S-1 find their inputs (kept only where it is not lt + j TPR)
    pk2 xacc[SUMS ? JL * RL : 1];               // sum over the run of s0 conj(s1) at the bins the thread's last items produce
};

// Complex arithmetic on register pairs with the operand modifiers of the packed instructions (op_sel picks the half of a
// 64-bit operand each lane reads, neg_lo / neg_hi negate it): a complex multiply is two instructions, a +- i d one, with no
// swizzle moves or sign flips in between (the compiler emits a v_mov / v_xor pair for each of those -- a sixth of this
// kernel's vector instructions).  Host build: the same values in plain C.
// a w in two halves: callers run the first halves of several products, then the second halves (a packed instruction that
// reads the result of the one just before it costs a wait state)
FX_HD pk2 cmul_lo(pk2 a, pk2 w) {           // a.x (w.x, w.y)
#if defined(__HIP_DEVICE_COMPILE__)
    pk2 t;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(w));
    return t;
#else
    return pk_splat(a[0]) * w;
#endif
}
FX_HD pk2 cmul_hi(pk2 a, pk2 w, pk2 t) {    // t + a.y (-w.y, w.x)
#if defined(__HIP_DEVICE_COMPILE__)
    pk2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
#else
    return pk_fma(pk_splat(a[1]), fxc::pk_muli(w), t);
#endif
}
FX_HD pk2 add_i(pk2 a, pk2 d) {             // a + i d
#if defined(__HIP_DEVICE_COMPILE__)
    pk2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(d));
    return r;
#else
    return pk2{a[0] - d[1], a[1] + d[0]};
#endif
}
FX_HD pk2 sub_i(pk2 a, pk2 d) {             // a - i d
#if defined(__HIP_DEVICE_COMPILE__)
    pk2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(d));
    return r;
#else
    return pk2{a[0] + d[1], a[1] - d[0]};
#endif
}
// acc + a conj(b), likewise in two halves
FX_HD pk2 x_acc_lo(pk2 acc, pk2 a, pk2 b) { // acc + b.x (a.x, a.y)
#if defined(__HIP_DEVICE_COMPILE__)
    pk2 t;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(t) : "v"(a), "v"(b), "v"(acc));
    return t;
#else
    return pk_fma(pk_splat(b[0]), a, acc);
#endif
}
FX_HD pk2 x_acc_hi(pk2 t, pk2 a, pk2 b) {   // t + b.y (a.y, -a.x)
#if defined(__HIP_DEVICE_COMPILE__)
    pk2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
    return r;
#else
    const pk2 ar = {a[1], -a[0]};
    return pk_fma(pk_splat(b[1]), ar, t);
#endif
}

// The roots of unity the butterflies need, as literals: cos and sin of 2 pi m / R evaluated at compile time in float64 (octant
// reduction on the integers, Taylor series below pi / 4) and rounded once to float32 -- what the table Args::tw holds for the same
// angle, without a scalar load (and its wait on the counter the LDS operations share) in the frame loop.
constexpr double kHalfPi = 1.57079632679489661923132169163975144;
constexpr double c_sin_small(double x) {
    const double x2 = x * x;
    double term = x, sum = x;
    for (int n = 1; n <= 12; ++n) {
        term *= -x2 / (double)((2 * n) * (2 * n + 1));
        sum += term;
    }
    return sum;
}
constexpr double c_cos_small(double x) {
    const double x2 = x * x;
    double term = 1.0, sum = 1.0;
    for (int n = 1; n <= 12; ++n) {
        term *= -x2 / (double)((2 * n - 1) * (2 * n));
        sum += term;
    }
    return sum;
}
struct RootCS {
    float c, s;
};
constexpr RootCS c_root(int m, int R) {          // exp(+2 pi i m / R)
    m %= R;
    if (m < 0) m += R;
    const int q = (4 * m) / R, mm = 4 * m - q * R;      // the angle is (pi / 2) (q + mm / R)
    double c = 1.0, s = 0.0;
    if (2 * mm <= R) {
        const double th = kHalfPi * (double)mm / (double)R;
        c = c_cos_small(th);
        s = c_sin_small(th);
    } else {
        const double th = kHalfPi * (double)(R - mm) / (double)R;
        c = c_sin_small(th);
        s = c_cos_small(th);
    }
    switch (q) {
        case 0: return RootCS{(float)c, (float)s};
        case 1: return RootCS{(float)-s, (float)c};
        case 2: return RootCS{(float)-c, (float)-s};
        default: return RootCS{(float)s, (float)-c};
    }
}
template <int R>
struct RootTab {
    float c[R], s[R];
};
template <int R>
constexpr RootTab<R> make_roots() {
    RootTab<R> t{};
    for (int m = 0; m < R; ++m) {
        const RootCS r = c_root(m, R);
        t.c[m] = r.c;
        t.s[m] = r.s;
    }
    return t;
}
template <int R>
struct RootsOf {
    static constexpr RootTab<R> tab = make_roots<R>();
};

// a times the literal (c, s): two instructions on the device, the literal in a scalar register pair
FX_HD pk2 cmul_k(pk2 a, float c, float s) {
#if defined(__HIP_DEVICE_COMPILE__)
    const pk2 k = {c, s};
    pk2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "s"(k));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(a), "s"(k), "v"(t));
    return r;
#else
    return pk2{a[0] * c - a[1] * s, a[0] * s + a[1] * c};
#endif
}

// How a composite radix R = A B runs in registers.  split_a(R) = A (0: R is 2, 4 or an odd prime -- a butterfly of its own):
// coprime A, B (6, 10, 12, 15, 20 ...): Good-Thomas -- input (B n1 + A n2) mod R, output the k with k = k1 mod A, k = k2 mod B, no
// twiddles in between; a prime power (8, 9, 16, 25 ...): Cooley-Tukey -- input n1 + A n2, the B-point transforms over n2, the
// literal twiddles w_R^(n1 k2), the A-point transforms over n1, output B k1 + k2.
constexpr int split_a(int R) {
    if (R == 2 || R == 4) return 0;
    int p2 = 1;
    while (R % (2 * p2) == 0) p2 *= 2;
    if (p2 > 1 && p2 < R) return p2;
    if (p2 == R) return 4;                        // 8 = 4 x 2, 16 = 4 x 4, 32 = 4 x 8
    int p = 3;
    while (R % p != 0) p += 2;
    if (p == R) return 0;
    int pk_ = p;
    while (R % (pk_ * p) == 0) pk_ *= p;
    if (pk_ < R) return pk_;                      // 15 = 3 x 5, 45 = 9 x 5
    int a = p;                                    // R = p^k: p^(k / 2)
    while (a * a * p * p <= R) a *= p;
    return a;
}
constexpr int inv_mod(int a, int m) {            // a^-1 mod m (coprime; 0 for m = 1)
    for (int x = 1; x < m; ++x)
        if ((a * x) % m == 1) return x;
    return 0;
}

// R-point DFT, kernel exp(+2 pi i q r / R), of v[0..R) into o[0..R) (registers): fx_mixed.h's dft_store with the +- i
// rotations folded into the additions.  Odd R: outputs q and R - q are P +- i Q, P = v0 + sum a_r cos, Q = sum d_r sin with
// a_r = v_r + v_{R-r}, d_r = v_r - v_{R-r}.
template <int R>
FX_HD void dft_regs(pk2 (&v)[R], pk2 (&o)[R]) {
#if (FXM_ABL & 4)
#pragma unroll
    for (int q = 0; q < R; ++q) o[q] = v[q];
    return;
#endif
    constexpr int A = split_a(R);
    if constexpr (R == 2) {
        o[0] = v[0] + v[1];
        o[1] = v[0] - v[1];
    } else if constexpr (R == 4) {
        const pk2 t0 = v[0] + v[2], t1 = v[0] - v[2], t2 = v[1] + v[3], t3 = v[1] - v[3];
        o[0] = t0 + t2;
        o[1] = add_i(t1, t3);
        o[2] = t0 - t2;
        o[3] = sub_i(t1, t3);
    } else if constexpr (A == 0) {
        static_assert(R % 2 == 1, "odd radix");
        constexpr int H = (R - 1) / 2;
        pk2 sum = v[0];
#pragma unroll
        for (int r = 1; r <= H; ++r) {
            const pk2 a = v[r] + v[R - r], d = v[r] - v[R - r];
            v[r] = a;
            v[R - r] = d;
            sum = sum + a;
        }
        o[0] = sum;
        pk2 pacc[H + 1], qacc[H + 1];
#pragma unroll
        for (int r = 1; r <= H; ++r)                           // (r outside: consecutive instructions belong to different outputs)
#pragma unroll
            for (int q = 1; q <= H; ++q) {
                const int m = (q * r) % R;                     // cos(2 pi m / R), sin(2 pi m / R)
                const float cs = RootsOf<R>::tab.c[m], sn = RootsOf<R>::tab.s[m];
                if (r == 1) {
                    pacc[q] = pk_fma(pk_splat(cs), v[r], v[0]);
                    qacc[q] = pk_splat(sn) * v[R - r];
                } else {
                    pacc[q] = pk_fma(pk_splat(cs), v[r], pacc[q]);
                    qacc[q] = pk_fma(pk_splat(sn), v[R - r], qacc[q]);
                }
            }
#pragma unroll
        for (int q = 1; q <= H; ++q) {
            o[q] = add_i(pacc[q], qacc[q]);
            o[R - q] = sub_i(pacc[q], qacc[q]);
        }
    } else {
        constexpr int B = R / A;
        pk2 y[R];
        if constexpr (gcd_of(A, B) == 1) {
#pragma unroll
            for (int n2 = 0; n2 < B; ++n2) {                   // A-point transforms over n1: y[k1 B + n2]
                pk2 t[A], u[A];
#pragma unroll
                for (int n1 = 0; n1 < A; ++n1) t[n1] = v[(B * n1 + A * n2) % R];
                dft_regs<A>(t, u);
#pragma unroll
                for (int k1 = 0; k1 < A; ++k1) y[k1 * B + n2] = u[k1];
            }
            constexpr int ea = B * inv_mod(B % A, A), eb = A * inv_mod(A % B, B);      // k = k1 ea + k2 eb mod R
#pragma unroll
            for (int k1 = 0; k1 < A; ++k1) {                   // B-point transforms over n2
                pk2 t[B], u[B];
#pragma unroll
                for (int n2 = 0; n2 < B; ++n2) t[n2] = y[k1 * B + n2];
                dft_regs<B>(t, u);
#pragma unroll
                for (int k2 = 0; k2 < B; ++k2) o[(k1 * ea + k2 * eb) % R] = u[k2];
            }
        } else {
#pragma unroll
            for (int n1 = 0; n1 < A; ++n1) {                   // B-point transforms over n2, then the twiddles w_R^(n1 k2): y[n1 B + k2]
                pk2 t[B], u[B];
#pragma unroll
                for (int n2 = 0; n2 < B; ++n2) t[n2] = v[n1 + A * n2];
                dft_regs<B>(t, u);
#pragma unroll
                for (int k2 = 0; k2 < B; ++k2) {
                    const int m = (n1 * k2) % R;
                    y[n1 * B + k2] = m == 0 ? u[k2] : cmul_k(u[k2], RootsOf<R>::tab.c[m], RootsOf<R>::tab.s[m]);
                }
            }
#pragma unroll
            for (int k2 = 0; k2 < B; ++k2) {                   // A-point transforms over n1
                pk2 t[A], u[A];
#pragma unroll
                for (int n1 = 0; n1 < A; ++n1) t[n1] = y[n1 * B + k2];
                dft_regs<A>(t, u);
#pragma unroll
                for (int k1 = 0; k1 < A; ++k1) o[B * k1 + k2] = u[k1];
            }
        }
    }
}
template <int R>
FX_HD void dft_to(pk2 (&v)[R], cf* d, int ds) {      // ... into d[q * ds]
    pk2 o[R];
    dft_regs<R>(v, o);
#if (FXM_ABL & 2)
    if (ds >= 0) return;          // (never true at run time as far as the compiler knows: ds is data)
#endif
#pragma unroll
    for (int q = 0; q < R; ++q) d[q * ds] = unpk(o[q]);
}

// what the launch hands every thread
struct Args {
    const void* x;            // [chunk][2][num_samp] complex64, or interleaved unsigned bytes (U8)
    const float* h;           // [T][N]
    cf* out;                  // raw[split][chunk][N], split = workgroup split * SLOTS + slot
    const cf* tw;             // [N] exp(+2 pi i n / N)
    const cf* dc_u8;          // U8: conversion offsets [chunk][2]
    long long num_samp, n_pts, n_chunks;   // F only: n_chunks = the number of STREAMS (a workgroup takes a pair of them)
    int wg_splits;
    int ant;                  // F only: spectra as out[stream / ant][frame][stream % ant][N] (1: [stream][frame][N])
    const float* h4;          // LEAN: the taps by point, [N][4] (zeros beyond T)
    const cf* tw1;            // LEAN: the first twiddle of butterfly j of stage s at thread lt: [tw_base(s) + j][TPR] = tw[(b mod ns) nb / ns],
                              // b = lt + j TPR (0 where the thread has no such butterfly)
    long long stride;         // samples from one stream of this launch to the next (0: num_samp -- the streams stand back to back; 2 num_samp: one
                              // antenna of every chunk pair)
    const cf* spec0;          // XM: the other antenna's spectra, [stream][frame][N]
};

// The body of one GPU thread.  Ctx: tid(), bid(), lds() (the workgroup's LDS as cf*), sync() (all threads of the
// workgroup -- or, for slots inside one wave, just program order).
template <class Ctx>
struct Body {
    Ctx& cx;
    const Args ar;            // (a copy: a reference would pin the kernel argument block to the stack)
    Thread th;
    pk2 hw2[LEAN ? 1 : (T * PTS + 1) / 2];   // the window taps of the thread's points, [t][p], two to a register pair (an array of
                              // plain floats stayed on the stack under clang 20)
    int lt, slot;
    cf *bx, *by;              // this slot's two buffers (S >= 2)
    const cf* xs[2];          // this chunk's two streams (complex64)
    const unsigned short* xb[2];
    pk2 off8[2];
    bool row_ok[2];           // F only: the row's stream exists (the last pair of an odd number of streams has one)
    cf* row_out[2];           // F only: where the row's stream puts its first frame's spectrum
    const cf* spec0_row;      // XM: the other antenna's spectrum of this stream's first frame
#if defined(__HIP_DEVICE_COMPILE__)
    __amdgpu_buffer_rsrc_t rsrc[2], rsrc_h, rsrc_t;
#endif

    FX_HD Body(Ctx& c, const Args& a) : cx(c), ar(a) {}

    FX_HD static bool has_item(int s, int j, int lt_) { return full_of(s) || j + 1 < j_of(s) || lt_ + j * TPR < items_of(s); }

    // item i of stage s >= 1 (i = g nb + b; any valid index for a thread without it): its butterfly, and where its rows' inputs and
    // outputs stand in the stage buffers (row a of the item: a RS further)
    template <int s>
    FX_HD static int item_bfly(int i) {
        constexpr int nb = nb_of(s);
        if (i >= items_of(s)) i = 0;
        return ngrp(s) > 1 ? i % nb : i;
    }
    template <int s>
    FX_HD static int item_in(int i) {
        constexpr int nb = nb_of(s);
        if (i >= items_of(s)) i = 0;
        const int g = ngrp(s) > 1 ? i / nb : 0, b = i - g * nb;
        return g * grp(s) * RS + rd_pos(s, b);
    }
    template <int s>
    FX_HD static int item_out(int i) {
        constexpr int nb = nb_of(s);
        if (i >= items_of(s)) i = 0;
        const int g = ngrp(s) > 1 ? i / nb : 0, b = i - g * nb;
        return g * grp(s) * RS + wr_pos(s, b);
    }

    // ---- once per launch: taps, twiddles, offsets
    FX_HD void init() {
        lt = cx.tid() % TPR;
        slot = cx.tid() / TPR;
        bx = cx.lds() + slot * LDS_PER_SLOT;
        by = bx + ROWS * RS;
        if constexpr (!LEAN) {
#pragma unroll
            for (int j = 0; j < J0; ++j)
#pragma unroll
                for (int r = 0; r < R0; ++r) {
                    const bool ok = has_item(0, j, lt);
                    const int m = ok ? lt + j * TPR + r * nb_of(0) : 0;      // (an unconditional load and a select: a branch here kept the taps on the stack)
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        const float w = ar.h[t * N + m];
                        hw2[(t * PTS + j * R0 + r) / 2][(t * PTS + j * R0 + r) % 2] = ok ? w : 0.f;
                    }
                }
        }
        if constexpr (!LEAN) init_stage<1>();
#pragma unroll
        for (int i = 0; i < (SUMS ? JL * RL : 1); ++i) th.xacc[i] = pk_splat(0.f);
    }
    template <int s>
    FX_HD void init_stage() {
        if constexpr (s < S) {
            constexpr int nb = nb_of(s), ns = ns_of(s), tmul = nb / ns;
#pragma unroll
            for (int j = 0; j < j_of(s); ++j) {
                const int i = lt + j * TPR;
                const int k = item_bfly<s>(i) % ns;
#pragma unroll
                for (int r = 1; r <= tw_per(s); ++r) th.tw[tw_base(s) + j * tw_per(s) + r - 1] = pk(ar.tw[r * k * tmul]);
                if constexpr (s < S - 1) th.ob[ob_base(s) + j] = item_out<s>(i);
                if constexpr (!plain_reads(s)) th.ib[ob_base(s) + j] = item_in<s>(i);
            }
            init_stage<s + 1>();
        }
    }
    // where item j of stage s finds its inputs / puts its outputs: registers, or (LEAN) recomputed every step -- hoisted out of the
    // step loop these offsets are registers again, hence the asm
    template <int s>
    FX_HD int in_of(int j) const {
        if constexpr (plain_reads(s)) {
            return lt + j * TPR;
        } else if constexpr (LEAN) {
            int i = lt + j * TPR;
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" : "+v"(i));
#endif
            return item_in<s>(i);
        } else {
            return th.ib[ob_base(s) + j];
        }
    }
    template <int s>
    FX_HD int out_of(int j) const {
        if constexpr (LEAN) {
            int i = lt + j * TPR;
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" : "+v"(i));
#endif
            return item_out<s>(i);
        } else {
            return th.ob[ob_base(s) + j];
        }
    }

    // ---- samples of frame f (relative to the chunk) into ring slot P.  Device: buffer loads -- one VGPR byte offset per
    // thread (it carries the frame: slots inside one wave are at different frames), the point's place in the frame is an
    // immediate, the antenna a buffer descriptor of its own.
    static constexpr int kElem = U8 ? 2 : 8;                  // bytes per sample
    static constexpr int kLoadAux = 0;      // cache policy of the sample loads: default (nontemporal measured neutral to slower here: profiles/r05/experiments.md 10)
    template <int P>
    FX_HD void load_frame(long long f, bool valid) {
#pragma unroll
        for (int j = 0; j < J0; ++j) load_points<P>(f, valid, j);
    }
    template <int P>
    FX_HD void load_points(long long f, bool valid, int j) {       // ... the points of first-stage butterfly j
#if (FXM_ABL & 16)
        f &= 3;
#endif
#if defined(__HIP_DEVICE_COMPILE__)
        typedef unsigned v2u32 __attribute__((ext_vector_type(2)));
        // butterfly j's points sit at N-1-m, m = lt + j TPR + r N/R0: one VGPR offset per j, counted from the lowest address of
        // its points (r = R0 - 1), so that it is non-negative for every thread that has the butterfly and the rest is an immediate
        const unsigned voff = (unsigned)((int)(f * N) + (N - 1 - lt - j * TPR - (R0 - 1) * nb_of(0))) * (unsigned)kElem;      // a chunk is below 2 GiB
#endif
#pragma unroll
        for (int a = 0; a < NA; ++a)
            {
#pragma unroll
                for (int r = 0; r < R0; ++r) {
                    const int m = lt + j * TPR + r * nb_of(0);
                    const bool ok = valid && has_item(0, j, lt) && !(FXM_ABL & 8) && (!FONLY || row_ok[a]);
                    pk2 v = pk_splat(0.f);
#if defined(__HIP_DEVICE_COMPILE__)
                    if constexpr (!U8 && FXM_OOB_ZERO) {
                        // a sample that does not exist (a frame outside the run, a lane without the butterfly, a stream that is not there) is
                        // asked for beyond the buffer's records: the hardware answers zero -- no branch, no zeroed register to fall back on
                        const unsigned cst = (unsigned)((R0 - 1 - r) * nb_of(0)) * (unsigned)kElem;
                        const v2u32 d = __builtin_amdgcn_raw_buffer_load_b64(rsrc[a], ok ? voff : 0xC0000000u, cst, kLoadAux);
                        th.ring[a][j * R0 + r][P] = pk2{__uint_as_float(d[0]), __uint_as_float(d[1])};
                        continue;
                    }
#endif
                    if (ok) {
#if defined(__HIP_DEVICE_COMPILE__)
                        const unsigned cst = (unsigned)((R0 - 1 - r) * nb_of(0)) * (unsigned)kElem;
                        if constexpr (U8) {
                            const unsigned raw = (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rsrc[a], voff, cst, kLoadAux);
                            const pk2 bytes = {(float)(raw & 0xFFu), (float)(raw >> 8)};
                            v = pk_fma(bytes, pk_splat(1.0f / 127.5f), off8[a]);
                        } else {
                            const v2u32 d = __builtin_amdgcn_raw_buffer_load_b64(rsrc[a], voff, cst, kLoadAux);
                            v = pk2{__uint_as_float(d[0]), __uint_as_float(d[1])};
                        }
#else
                        const long long at = f * N + (N - 1 - m);
                        if constexpr (U8) {
                            const unsigned raw = xb[a][at];
                            const pk2 bytes = {(float)(raw & 0xFFu), (float)(raw >> 8)};
                            v = pk_fma(bytes, pk_splat(1.0f / 127.5f), off8[a]);
                        } else {
                            v = pk(xs[a][at]);
                        }
#endif
                    }
                    th.ring[a][j * R0 + r][P] = v;
                }
            }
    }

    // LEAN: the first twiddles of the thread's items of stage s, from the table
    template <int s>
    FX_HD void load_tw1(pk2 (&w1)[j_of(s)]) {
#pragma unroll
        for (int j = 0; j < j_of(s); ++j) {
#if defined(__HIP_DEVICE_COMPILE__)
            typedef unsigned v2u32 __attribute__((ext_vector_type(2)));
            const v2u32 d = __builtin_amdgcn_raw_buffer_load_b64(rsrc_t, (unsigned)lt * 8u, (unsigned)((tw_base(s) + j) * TPR) * 8u, 0);
            w1[j] = pk2{__uint_as_float(d[0]), __uint_as_float(d[1])};
#else
            w1[j] = pk(ar.tw1[(tw_base(s) + j) * TPR + lt]);
#endif
        }
    }
    // The twiddles of item j of stage s.  Kept whole (tw_per = R - 1): w[r] = w^r from the registers.  Else only w^1 is kept (LEAN: it
    // comes from the table) and the others are its powers, formed every step: for a prime radix all of them, w^r = w^(r/2) w^(r - r/2) (the
    // shortest chains); for a composite radix R = A B only the base powers w^a (a < A, in w[a]) and w^(A c) (c < B, in wc[c]) -- the
    // butterfly forms w^(a + A c) = w^a w^(A c) where it needs it, so that no more than A + B of them are ever live.
    template <int s>
    static constexpr bool whole_tw() { return tw_per(s) == kRadix[s] - 1 && !LEAN; }
    template <int s>
    static constexpr int base_a() { return split_a(kRadix[s]) > 0 ? split_a(kRadix[s]) : kRadix[s]; }
    template <int s>
    FX_HD void stage_tw(int j, pk2 (&w)[kRadix[s]], pk2 (&wc)[kRadix[s] / base_a<s>()], const pk2* w1) {
        constexpr int R = kRadix[s], A = base_a<s>(), B = R / A;
        if constexpr (whole_tw<s>()) {
#pragma unroll
            for (int r = 1; r < R; ++r) w[r] = th.tw[tw_base(s) + j * (R - 1) + r - 1];
        } else {
            w[1] = LEAN ? w1[j] : th.tw[tw_base(s) + j];
#pragma unroll
            for (int r = 2; r < A; ++r) {
                const pk2 a = w[r / 2], b = w[r - r / 2];
                w[r] = cmul_hi(a, b, cmul_lo(a, b));
            }
            if constexpr (B > 1) {
                const pk2 a = w[A / 2], b = w[A - A / 2];
                wc[1] = cmul_hi(a, b, cmul_lo(a, b));
#pragma unroll
                for (int c = 2; c < B; ++c) {
                    const pk2 x = wc[c / 2], y = wc[c - c / 2];
                    wc[c] = cmul_hi(x, y, cmul_lo(x, y));
                }
            }
        }
    }
    // w^n for input n of a composite butterfly of stage s (n >= 1)
    template <int s>
    FX_HD pk2 tw_at(int n, const pk2 (&w)[kRadix[s]], const pk2 (&wc)[kRadix[s] / base_a<s>()]) {
        constexpr int A = base_a<s>();
        if constexpr (whole_tw<s>()) {
            return w[n];
        } else {
            const int a = n % A, c = n / A;
            if (c == 0) return w[a];
            if (a == 0) return wc[c];
            return cmul_hi(w[a], wc[c], cmul_lo(w[a], wc[c]));
        }
    }

    // One butterfly of stage s >= 1 for one row: inputs p[r rs] (LDS), times the twiddles, through the R-point transform into o.  A
    // composite radix runs its first level group by group (load A or B inputs, twiddle, transform), so that the inputs of the next group
    // need not be live while this one is in flight.
    template <int s>
    FX_HD void stage_bfly(const cf* p, const pk2 (&w)[kRadix[s]], const pk2 (&wc)[kRadix[s] / base_a<s>()], pk2 (&o)[kRadix[s]]) {
        constexpr int R = kRadix[s], rs = rd_stride(s), A = split_a(R);
        if constexpr (A == 0) {
            pk2 v[R], t[R];
#pragma unroll
            for (int r = 0; r < R; ++r) v[r] = pk(p[r * rs]);
#pragma unroll
            for (int r = 1; r < R; ++r) t[r] = cmul_lo(v[r], w[r]);
#pragma unroll
            for (int r = 1; r < R; ++r) v[r] = cmul_hi(v[r], w[r], t[r]);
            dft_regs<R>(v, o);
        } else {
            constexpr int B = R / A;
            constexpr bool PFA = gcd_of(A, B) == 1;
            constexpr int GN = PFA ? B : A, GL = PFA ? A : B;       // groups of the first level, inputs of each
            pk2 y[R];
#pragma unroll
            for (int g = 0; g < GN; ++g) {
                pk2 v[GL], t[GL], u[GL];
#pragma unroll
                for (int i = 0; i < GL; ++i) v[i] = pk(p[(PFA ? (B * i + A * g) % R : g + A * i) * rs]);
#pragma unroll
                for (int i = 0; i < GL; ++i) {
                    const int n = PFA ? (B * i + A * g) % R : g + A * i;
                    if (n > 0) {
                        const pk2 wn = tw_at<s>(n, w, wc);
                        t[i] = cmul_lo(v[i], wn);
                        v[i] = cmul_hi(v[i], wn, t[i]);
                    }
                }
                dft_regs<GL>(v, u);
#pragma unroll
                for (int k = 0; k < GL; ++k) {
                    if constexpr (PFA) {
                        y[k * B + g] = u[k];
                    } else {
                        const int m = (g * k) % R;
                        y[g * B + k] = m == 0 ? u[k] : cmul_k(u[k], RootsOf<R>::tab.c[m], RootsOf<R>::tab.s[m]);
                    }
                }
            }
            if constexpr (PFA) {
                constexpr int ea = B * inv_mod(B % A, A), eb = A * inv_mod(A % B, B);
#pragma unroll
                for (int k1 = 0; k1 < A; ++k1) {
                    pk2 t[B], u[B];
#pragma unroll
                    for (int n2 = 0; n2 < B; ++n2) t[n2] = y[k1 * B + n2];
                    dft_regs<B>(t, u);
#pragma unroll
                    for (int k2 = 0; k2 < B; ++k2) o[(k1 * ea + k2 * eb) % R] = u[k2];
                }
            } else {
#pragma unroll
                for (int k2 = 0; k2 < B; ++k2) {
                    pk2 t[A], u[A];
#pragma unroll
                    for (int n1 = 0; n1 < A; ++n1) t[n1] = y[n1 * B + k2];
                    dft_regs<A>(t, u);
#pragma unroll
                    for (int k1 = 0; k1 < A; ++k1) o[B * k1 + k2] = u[k1];
                }
            }
        }
    }

    // ---- one middle stage: LDS -> LDS
    template <int s>
    FX_HD void mid_stage(const cf* src, cf* dst, const pk2* w1 = nullptr) {
        constexpr int R = kRadix[s], ns = ns_of(s);
#pragma unroll
        for (int j = 0; j < j_of(s); ++j) {
            if (has_item(s, j, lt)) {
                pk2 w[R], wc[R / base_a<s>()];
                stage_tw<s>(j, w, wc, w1);
                const int ib = in_of<s>(j), ob = out_of<s>(j);
#pragma unroll
                for (int a = 0; a < grp(s); ++a) {       // (every row of the item: the same indices and twiddles serve them all)
                    pk2 o[R];
                    stage_bfly<s>(src + a * RS + ib, w, wc, o);
#if (FXM_ABL & 2)
                    if (ns >= 0 && ar.wg_splits >= 0) continue;          // (never false at run time as far as the compiler knows)
#endif
#pragma unroll
                    for (int q = 0; q < R; ++q) dst[a * RS + ob + q * ns] = unpk(o[q]);
                }
            }
        }
    }

    // ---- the last stage (s = S - 1 >= 1): LDS -> registers -> X.  f: the step's first frame; frames from f_end on do not exist for
    // this slot (their rows hold the transform of zeros: nothing is added, nothing stored)
    FX_HD void last_stage(const cf* src, long long f, long long f_end, const pk2* w1 = nullptr) {
        constexpr int s = S - 1, R = RL, G = grp(s), FR = G / NA > 0 ? G / NA : 1;
#pragma unroll
        for (int j = 0; j < JL; ++j) {
            if (!has_item(s, j, lt)) continue;
            pk2 w[R], wc[R / base_a<s>()];
            stage_tw<s>(j, w, wc, w1);
            const int ib = in_of<s>(j);
            // the item's rows: g G .. g G + G - 1 of the step's (frame u, antenna a) -> u NA + a
            int g = 0;
            if constexpr (ngrp(s) > 1) g = (lt + j * TPR) / nb_of(s);
            const int bfly = ngrp(s) > 1 ? lt + j * TPR - g * nb_of(s) : lt + j * TPR;
            if constexpr (G % NA == 0) {
#pragma unroll
                for (int u = 0; u < FR; ++u) {
                    const long long frame = f + g * FR + u;
                    if (frame < f_end) {
                        pk2 o[NA][R];
#pragma unroll
                        for (int a = 0; a < NA; ++a) stage_bfly<s>(src + (u * NA + a) * RS + ib, w, wc, o[a]);
                        emit<R>(o, j, bfly, frame);
                    }
                }
            } else {
                // F only, one row per item (G = 1 of NA = 2): row g = (frame u, stream a)
                static_assert(G % NA == 0 || (FONLY && G == 1), "items of single rows: the F stage alone");
                const int a = g % NA;
                const long long frame = f + g / NA;
                if (frame < f_end && (a ? row_ok[NA - 1] : row_ok[0])) {
                    pk2 o[R];
                    stage_bfly<s>(src + ib, w, wc, o);
                    cf* d = (a ? row_out[NA - 1] : row_out[0]) + frame * (long long)ar.ant * N + bfly;
#pragma unroll
                    for (int q = 0; q < R; ++q) fxc::st_store(d + q * (N / R), unpk(o[q]));
                }
            }
        }
    }

    // the two rows' spectra of one butterfly of the last stage: X-multiplied into the thread's sums, or (F only) stored -- output q of
    // butterfly b is bin b + q N/R there, so the lanes of a wave write R runs of consecutive bins
    template <int R>
    FX_HD void emit(pk2 (&o)[NA][R], int j, int bfly, long long frame) {
        if constexpr (XM) {
            // antenna 0's values of the same bins (its F-only launch wrote them in natural order: a wave reads R runs of consecutive bins)
            const cf* d = spec0_row + frame * (long long)N + bfly;
            pk2 s0[R];
#pragma unroll
            for (int q = 0; q < R; ++q) s0[q] = pk(d[q * (N / R)]);
#pragma unroll
            for (int q = 0; q < R; ++q) th.xacc[j * R + q] = x_acc_lo(th.xacc[j * R + q], s0[q], o[0][q]);
#pragma unroll
            for (int q = 0; q < R; ++q) th.xacc[j * R + q] = x_acc_hi(th.xacc[j * R + q], s0[q], o[0][q]);
        } else if constexpr (FONLY) {
#pragma unroll
            for (int a = 0; a < NA; ++a)
                if (row_ok[a]) {
                    cf* d = row_out[a] + frame * (long long)ar.ant * N + bfly;
#pragma unroll
                    for (int q = 0; q < R; ++q) fxc::st_store(d + q * (N / R), unpk(o[a][q]));
                }
        } else {
#pragma unroll
            for (int q = 0; q < R; ++q) th.xacc[j * R + q] = x_acc_lo(th.xacc[j * R + q], o[0][q], o[NA - 1][q]);
#pragma unroll
            for (int q = 0; q < R; ++q) th.xacc[j * R + q] = x_acc_hi(th.xacc[j * R + q], o[0][q], o[NA - 1][q]);
        }
    }

    template <int s>
    FX_HD void mid_stages(cf* rd, cf* wr) {       // stage s reads rd, writes wr; a barrier behind each
        if constexpr (s < S - 1) {
            mid_stage<s>(rd, wr);
            cx.sync();
            mid_stages<s + 1>(wr, rd);
        }
    }

    // the first stage's butterfly j of one row, out of registers into the slot's buffer (planes, or blocks of R0 with their padding)
    FX_HD void first_to_lds(pk2 (&v)[R0], int row, int j) {
        const int b = lt + j * TPR;
        if constexpr (P0 > 0)
            dft_to<R0>(v, bx + row * RS + b, P0);
        else
            dft_to<R0>(v, bx + row * RS + b * (R0 + pad_of(0)), 1);      // ns = 1: o = b R0
    }

    // the FIR of the step at ring slot P (frames f ...) into sums, and the next frames' samples requested into the slots it leaves
    template <int P>
    FX_HD void fir_and_load(long long f, long long f_end, pk2 (&acc)[U][NA][PTS]) {
#pragma unroll
        for (int t = 0; t < T; ++t)                     // (tap outside: consecutive instructions belong to different points)
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int a = 0; a < NA; ++a)
#pragma unroll
                    for (int p = 0; p < PTS; ++p) {
                        const pk2 w = pk_splat(hw2[LEAN ? 0 : (t * PTS + p) / 2][(t * PTS + p) % 2]);
                        const pk2 x = th.ring[a][p][(P + u - t + NS) % NS];
                        acc[u][a][p] = t == 0 ? w * x : pk_fma(w, x, acc[u][a][p]);
                    }
        // the U oldest slots are free now: the next step's frames go there, in flight through the stages below
        load_frame<(P + U) % NS>(f + U, f + U < f_end);
        if constexpr (U == 2) load_frame<(P + U + 1) % NS>(f + U + 1, f + U + 1 < f_end);
    }
    // ---- one step: the FIR of its U frames out of the ring, the first butterfly, the stages, X.  P = the ring slot of the
    // step's first frame f.  Frame f + u exists for this slot while f + u < f_end (the slots of a workgroup take the same number of
    // steps); frames from f_end on are not loaded
    template <int P>
    FX_HD void step(long long f, long long f_end) {
        if constexpr (LEAN) {
            step_lean<P>(f, f_end);
            return;
        }
        pk2 acc[U][NA][PTS];
        fir_and_load<P>(f, f_end, acc);
        if constexpr (S == 1) {
#pragma unroll
            for (int j = 0; j < J0; ++j)
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (f + u < f_end && has_item(0, j, lt)) {
                        pk2 o[NA][R0];
#pragma unroll
                        for (int a = 0; a < NA; ++a) {
                            pk2 v[R0];
#pragma unroll
                            for (int r = 0; r < R0; ++r) v[r] = acc[u][a][j * R0 + r];
                            dft_regs<R0>(v, o[a]);
                        }
                        emit<R0>(o, j, lt + j * TPR, f + u);
                    }
        } else {
#pragma unroll
            for (int j = 0; j < J0; ++j)
                if (has_item(0, j, lt)) {
#pragma unroll
                    for (int u = 0; u < U; ++u)
#pragma unroll
                        for (int a = 0; a < NA; ++a) {
                            pk2 v[R0];
#pragma unroll
                            for (int r = 0; r < R0; ++r) v[r] = acc[u][a][j * R0 + r];
                            first_to_lds(v, u * NA + a, j);
                        }
                }
            cx.sync();
            mid_stages<1>(bx, by);
            last_stage((S % 2 == 0) ? bx : by, f, f_end);      // stage S-2 wrote X when S is even
            if constexpr (SWAP) {
                cf* t = bx;
                bx = by;
                by = t;
            }
        }
    }

    // LEAN: the first twiddles of every item of every stage, [tw_base(s) + j], requested at the top of the step -- BEFORE the next frames'
    // samples: vector-memory loads return in order, so a wait for a table entry requested behind the samples would wait for the samples
    // (an HBM round trip in every step; FXM_LEAN_TW_EARLY 0: one stage ahead, as in round 5)
#ifndef FXM_LEAN_TW_EARLY
#define FXM_LEAN_TW_EARLY 1
#endif
    static constexpr bool TW_EARLY = FXM_LEAN_TW_EARLY != 0 && TW1C <= 12;      // (3^8 channels: 35 entries -- 70 registers -- spilled; those keep one stage ahead)
    template <int s>
    FX_HD void load_tw1_all(pk2 (&all)[TW1C > 0 ? TW1C : 1]) {
        if constexpr (s < S) {
            pk2 w[j_of(s)];
            load_tw1<s>(w);
#pragma unroll
            for (int j = 0; j < j_of(s); ++j) all[tw_base(s) + j] = w[j];
            load_tw1_all<s + 1>(all);
        }
    }
    // LEAN: stage s reads rd and writes wr with the first twiddles w1 (one stage ahead: the next stage's are requested before; early: all of
    // them are there already); the last stage ends in X
    template <int s>
    FX_HD void lean_stages(cf* rd, cf* wr, const pk2* w1, long long f, long long f_end) {
        if constexpr (s < S - 1) {
            if constexpr (TW_EARLY) {
                mid_stage<s>(rd, wr, w1 + tw_base(s));
                cx.sync();
                lean_stages<s + 1>(wr, rd, w1, f, f_end);
            } else {
                pk2 nxt[j_of(s + 1)];
                load_tw1<s + 1>(nxt);
                mid_stage<s>(rd, wr, w1);
                cx.sync();
                lean_stages<s + 1>(wr, rd, nxt, f, f_end);
            }
        } else {
            last_stage(rd, f, f_end, TW_EARLY ? w1 + tw_base(s) : w1);
        }
    }

    // LEAN (S >= 2): one first-stage butterfly at a time -- its points' taps from L2, their FIR, the butterfly into LDS -- so that only R0
    // points' taps and sums are live at once; the next frames' samples go into the ring slots the FIR has left (early: behind all of them)
    template <int P>
    FX_HD void step_lean(long long f, long long f_end) {
        static_assert(!LEAN || S >= 2, "the lean build needs a first stage into LDS (h_rtc.h::spec_shape)");
        pk2 w1[TW_EARLY ? (TW1C > 0 ? TW1C : 1) : j_of(1)];
        if constexpr (TW_EARLY) {
            load_tw1_all<1>(w1);
        } else {
            pk2 first[j_of(1)];
            load_tw1<1>(first);
#pragma unroll
            for (int j = 0; j < j_of(1); ++j) w1[j] = first[j];
        }
#pragma unroll
        for (int j = 0; j < J0; ++j) {
            float hq[R0][4];
#pragma unroll
            for (int r = 0; r < R0; ++r) fetch_taps(hq[r], j, r);
            pk2 acc[U][NA][R0];
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int a = 0; a < NA; ++a)
#pragma unroll
                        for (int r = 0; r < R0; ++r) {
                            const pk2 x = th.ring[a][j * R0 + r][(P + u - t + NS) % NS];
                            acc[u][a][r] = t == 0 ? pk_splat(hq[r][t]) * x : pk_fma(pk_splat(hq[r][t]), x, acc[u][a][r]);
                        }
            if constexpr (!TW_EARLY || J0 > 2) {      // (many first butterflies a thread: the samples of each behind its FIR, as the ring frees)
                load_points<(P + U) % NS>(f + U, f + U < f_end, j);
                if constexpr (U == 2) load_points<(P + U + 1) % NS>(f + U + 1, f + U + 1 < f_end, j);
            }
            if (has_item(0, j, lt)) {
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int a = 0; a < NA; ++a) first_to_lds(acc[u][a], u * NA + a, j);
            }
        }
        if constexpr (TW_EARLY && J0 <= 2) {      // every table entry of the step has been requested: now the samples
            load_frame<(P + U) % NS>(f + U, f + U < f_end);
            if constexpr (U == 2) load_frame<(P + U + 1) % NS>(f + U + 1, f + U + 1 < f_end);
        }
        cx.sync();
        lean_stages<1>(bx, by, w1, f, f_end);
        if constexpr (SWAP) {
            cf* t = bx;
            bx = by;
            by = t;
        }
    }
    // LEAN: the (up to four) taps of point (j, r) of the thread from the table of quads
    FX_HD void fetch_taps(float (&q4)[4], int j, int r) {
#if defined(__HIP_DEVICE_COMPILE__)
        typedef unsigned v4u32 __attribute__((ext_vector_type(4)));
        const v4u32 q = __builtin_amdgcn_raw_buffer_load_b128(rsrc_h, (unsigned)(lt + j * TPR) * 16u, (unsigned)(r * nb_of(0)) * 16u, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) q4[t] = __uint_as_float(q[t]);      // (past the table: zeros, and so are those lanes' samples)
#else
        const int m = has_item(0, j, lt) ? lt + j * TPR + r * nb_of(0) : 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) q4[t] = ar.h4[4 * m + t];
#endif
    }

    // UNR steps per trip: step k of a trip starts at ring slot (k U) mod NS
    template <int K>
    FX_HD void steps(long long f, long long f_end, long long i, long long n_steps) {
        if constexpr (K < UNR) {
            if (i + K < n_steps) {          // uniform over the workgroup
                step<(K * U) % NS>(f + K * U, f_end);
                steps<K + 1>(f, f_end, i, n_steps);
            }
        }
    }

    // frames f0 - (T - 1) .. f0 + U - 1 into their slots: frame g in slot (g - f0) mod NS
    template <int K>
    FX_HD void preload(long long f0, long long f_end) {
        if constexpr (K < NS) {
            const long long g = f0 - (T - 1) + K;
            load_frame<(K - (T - 1) + NS) % NS>(g, g >= 0 && g < f_end);
            preload<K + 1>(f0, f_end);
        }
    }

    FX_HD void run() {
        const long long chunk = cx.bid() / ar.wg_splits;
        const int sp = (int)(cx.bid() % ar.wg_splits);
        init();
        const int e = sp * SLOTS + slot, E = ar.wg_splits * SLOTS;
        const long long f0 = (long long)e * ar.n_pts / E, f1 = ((long long)e + 1) * ar.n_pts / E;
        const long long n_steps = ((ar.n_pts + E - 1) / E + U - 1) / U;       // steps of the longest run of any slot: uniform
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            const long long st = NA * chunk + a;                    // F only: the row's stream; `chunk` counts groups of NA streams
            row_ok[a] = !FONLY || st < ar.n_chunks;
            const long long oc = FONLY && row_ok[a] ? st / ar.ant : 0;
            row_out[a] = FONLY && !XM && row_ok[a] ? ar.out + ((oc * ar.n_pts) * ar.ant + (st - oc * ar.ant)) * N : nullptr;
            if constexpr (XM) spec0_row = ar.spec0 + (row_ok[a] ? st : 0) * ar.n_pts * N;
            const long long stride = ar.stride > 0 ? ar.stride : ar.num_samp;
            xs[a] = reinterpret_cast<const cf*>(ar.x) + st * stride;
            xb[a] = reinterpret_cast<const unsigned short*>(ar.x) + st * stride;
            off8[a] = U8 ? pk(ar.dc_u8[st]) : pk_splat(0.f);
#if defined(__HIP_DEVICE_COMPILE__)
            rsrc[a] = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(ar.x)) + (row_ok[a] ? st : 0) * stride * kElem, 0,
                                                        (int)(ar.num_samp * kElem), 0x00020000);
#endif
        }
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (LEAN) {
            rsrc_h = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ar.h4), 0, N * 16, 0x00020000);
            rsrc_t = __builtin_amdgcn_make_buffer_rsrc(const_cast<cf*>(ar.tw1), 0, TW1C * TPR * 8, 0x00020000);
        }
#endif
        // zero history in front of the chunk; a run that starts inside it re-reads T - 1 frames
        preload<0>(f0, f1);
        for (long long i = 0; i < n_steps; i += UNR) steps<0>(f0 + i * U, f1, i, n_steps);
        if constexpr (SUMS) {
            // the thread's bins: the last stage's butterfly b puts output q at b + q N/RL (k = b there: ns = N/RL)
            cf* o = ar.out + ((long long)e * ar.n_chunks + chunk) * N;
            constexpr int s = S - 1, nb = nb_of(S - 1);
            if constexpr (S >= 2 && ngrp(s) > 1) {
                // the frames of a step went to different threads: their sums meet in LDS (row g of the slot's buffer), the thread that
                // had group 0 of a butterfly adds them up in group order
                cf* red = cx.lds() + slot * LDS_PER_SLOT;
                cx.sync();
#pragma unroll
                for (int j = 0; j < JL; ++j)
                    if (has_item(s, j, lt)) {
                        const int i = lt + j * TPR, g = i / nb, b = i - g * nb;
#pragma unroll
                        for (int q = 0; q < RL; ++q) red[g * N + b + q * nb] = unpk(th.xacc[j * RL + q]);
                    }
                cx.sync();
#pragma unroll
                for (int j = 0; j < JL; ++j)
                    if (has_item(s, j, lt) && lt + j * TPR < nb) {
                        const int b = lt + j * TPR;
#pragma unroll
                        for (int q = 0; q < RL; ++q) {
                            pk2 sum = pk(red[b + q * nb]);
#pragma unroll
                            for (int g = 1; g < ngrp(s); ++g) sum = sum + pk(red[g * N + b + q * nb]);
                            o[b + q * nb] = unpk(sum);
                        }
                    }
            } else {
#pragma unroll
                for (int j = 0; j < JL; ++j)
                    if (has_item(s, j, lt))
#pragma unroll
                        for (int q = 0; q < RL; ++q) o[lt + j * TPR + q * nb] = unpk(th.xacc[j * RL + q]);
            }
        }
    }
};

}  // namespace fxm

#if defined(__HIPCC__)
namespace fxm {
struct DeviceCtx {
    cf* lds_;
    __device__ __forceinline__ int tid() const { return (int)threadIdx.x; }
    __device__ __forceinline__ long long bid() const { return (long long)blockIdx.x; }
    __device__ __forceinline__ cf* lds() const { return lds_; }
    __device__ __forceinline__ void sync() const {
#if (FXM_ABL & 1)
        return;
#endif
        if constexpr (TPR <= 64)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // a slot is (part of) one wave: LDS operations of a wave complete in order
        else
            __syncthreads();
    }
};
}  // namespace fxm

#ifndef FXM_WAVES
#define FXM_WAVES 1      // waves per SIMD the register allocation is held to (h_rtc.h tries the higher occupancy first)
#endif
extern "C" __global__ __launch_bounds__(FXM_TPR * FXM_SLOTS, FXM_WAVES) void fxm_fx2_kernel(const fxm::Args args) {
    // static: the size is a compile-time constant here, and a module function needs no attribute to go past 64 KiB this way
    __shared__ __attribute__((aligned(16))) fxm::cf fxm_smem[fxm::SLOTS * fxm::LDS_PER_SLOT > 0 ? fxm::SLOTS * fxm::LDS_PER_SLOT : 1];
    fxm::DeviceCtx cx{fxm_smem};
    fxm::Body<fxm::DeviceCtx> body(cx, args);
    body.run();
}
#endif
