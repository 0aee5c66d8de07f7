// fx_mixed.h — the mixed-radix Stockham FFT of the generic path (any channel count that is not a power of two:
// `--resolution` is a free integer in the reference, effex/effex.py:733-739).  Compiles as device code under hipcc and as
// plain C++ under g++ (tests/emul runs the same stages thread by thread on the host — test infrastructure only).
//
//   X[k] = sum_n v[n] exp(+2 pi i n k / N),  N = r_0 r_1 ... r_{S-1}
//
// Stage j (radix R = r_j, Ns = r_0 ... r_{j-1}) is the autosort step of Stockham's FFT: butterfly b in [0, N/R) takes
// the R inputs src[b + r N/R], turns input r by w_N^(r k N/(Ns R)) with k = b mod Ns, runs an R-point DFT over them and
// writes output q to dst[(b - k) R + k + q Ns].  After the last stage the spectrum stands in natural order: no
// bit/digit reversal anywhere.  Every twiddle is an exact index into one table tw[n] = exp(+2 pi i n / N), n < N,
// rounded once from float64 (the same table the direct DFT used).
#pragma once

#include "fx_math.h"

namespace fxc {

constexpr int kMixedMaxStages = 16;      // 2^14 = 16384 is the largest N, and fours are taken before twos

struct MixedPlan {
    int n_stages;
    int radix[kMixedMaxStages];
};

// fours first, one two if it is left, then the odd primes ascending (the stage order does not change the result)
inline MixedPlan mixed_factor(int n) {
    MixedPlan mp;
    mp.n_stages = 0;
    for (int s = 0; s < kMixedMaxStages; ++s) mp.radix[s] = 1;
    while (n % 4 == 0) {
        mp.radix[mp.n_stages++] = 4;
        n /= 4;
    }
    if (n % 2 == 0) {
        mp.radix[mp.n_stages++] = 2;
        n /= 2;
    }
    for (int p = 3; n > 1; p += 2) {
        if ((long long)p * p > n) p = n;        // what is left is prime
        while (n % p == 0) {
            if (mp.n_stages == kMixedMaxStages) {   // cannot happen for n <= 16384 (3^8 = 6561 is 8 stages)
                mp.n_stages = -1;
                return mp;
            }
            mp.radix[mp.n_stages++] = p;
            n /= p;
        }
    }
    return mp;
}

// Two floats in one register pair: gfx950 runs v_pk_add/mul/fma_f32 at the rate of the scalar forms, so complex
// arithmetic written on pairs costs half the issue slots (the stages are issue-bound, not LDS- or HBM-bound).
typedef float pk2 __attribute__((vector_size(8)));
FX_HD pk2 pk(cf a) { pk2 r = {a.x, a.y}; return r; }
FX_HD cf unpk(pk2 a) { return mk(a[0], a[1]); }
FX_HD pk2 pk_splat(float s) { pk2 r = {s, s}; return r; }
FX_HD pk2 pk_muli(pk2 a) { pk2 r = {-a[1], a[0]}; return r; }     // times +i
FX_HD pk2 pk_fma(pk2 a, pk2 b, pk2 c) {
#if defined(__clang__)
    return __builtin_elementwise_fma(a, b, c);
#else
    return a * b + c;
#endif
}
// a * w = a.x (w.x, w.y) + a.y (-w.y, w.x)
FX_HD pk2 pk_cmul(pk2 a, pk2 w) { return pk_fma(pk_splat(a[1]), pk_muli(w), pk_splat(a[0]) * w); }

// the R-th roots of unity the odd butterflies need: w[m] = exp(+2 pi i m / R), m = 1 .. (R-1)/2, at stride N/R in the table
template <int R>
struct Roots {
    pk2 w[(R - 1) / 2 + 1];
};
template <int R>
FX_HD Roots<R> load_roots(const cf* tw, int root_stride) {
    Roots<R> rt;
    rt.w[0] = pk_splat(1.f);
    if constexpr (R != 2 && R != 4) {
#pragma unroll
        for (int m = 1; m <= (R - 1) / 2; ++m) rt.w[m] = pk(tw[m * root_stride]);
    }
    return rt;
}

// R-point DFT, kernel exp(+2 pi i q r / R), of v[0..R) straight into d[q * ds].  Odd R: inputs r and R - r enter as a sum
// and an i-times-difference, outputs q and R - q leave as P +- Q with P = v0 + sum a_r cos(2 pi q r / R),
// Q = sum b_r sin(2 pi q r / R) -- half the multiplies of the plain sum.
template <int R>
FX_HD void dft_store(pk2 (&v)[R], const Roots<R>& rt, cf* d, int ds) {
    if constexpr (R == 2) {
        d[0] = unpk(v[0] + v[1]);
        d[ds] = unpk(v[0] - v[1]);
    } else if constexpr (R == 4) {
        const pk2 t0 = v[0] + v[2], t1 = v[0] - v[2], t2 = v[1] + v[3], t3 = pk_muli(v[1] - v[3]);
        d[0] = unpk(t0 + t2);
        d[ds] = unpk(t1 + t3);
        d[2 * ds] = unpk(t0 - t2);
        d[3 * ds] = unpk(t1 - t3);
    } else {
        static_assert(R % 2 == 1, "odd radix");
        constexpr int H = (R - 1) / 2;
        pk2 sum = v[0];
#pragma unroll
        for (int r = 1; r <= H; ++r) {
            const pk2 a = v[r] + v[R - r], b = pk_muli(v[r] - v[R - r]);
            v[r] = a;
            v[R - r] = b;
            sum = sum + a;
        }
        d[0] = unpk(sum);
#pragma unroll
        for (int q = 1; q <= H; ++q) {
            pk2 pacc = v[0], qacc = {0.f, 0.f};
#pragma unroll
            for (int r = 1; r <= H; ++r) {
                const int m = (q * r) % R;                     // cos(2 pi m / R), sin(2 pi m / R) from the half table
                const pk2 ww = rt.w[m <= H ? m : R - m];
                pacc = pk_fma(pk_splat(ww[0]), v[r], pacc);
                const pk2 sb = pk_splat(ww[1]) * v[R - r];
                if (r == 1)
                    qacc = m <= H ? sb : -sb;
                else
                    qacc = m <= H ? qacc + sb : qacc - sb;
            }
            d[q * ds] = unpk(pacc + qacc);
            d[(R - q) * ds] = unpk(pacc - qacc);
        }
    }
}

// a / d for 0 <= a < 2^15 and 1 <= d < 2^15 with inv = 1.0f / d: (a + 1/2) / d is at least 1/(2d) away from an integer, and
// the rounding of the product is below a / d * 2^-22 <= 2^-7 / d -- the truncation cannot land on the wrong side
FX_HD int small_div(int a, float inv) { return (int)(((float)a + 0.5f) * inv); }

// one stage, register butterflies: "thread" lt of tpr walks the butterflies lt, lt + tpr, ... of U rows that stand
// row_stride elements apart in src and in dst -- one index computation and one set of twiddles for all of them
template <int R, int U>
FX_HD void mixed_stage_reg(const cf* src, cf* dst, int row_stride, const cf* tw, int n, int ns, int lt, int tpr) {
    const int nb = n / R;
    const int tmul = nb / ns;          // N / (Ns R)
    const float inv_ns = 1.0f / (float)ns;
    const Roots<R> rt = load_roots<R>(tw, nb);
    for (int b = lt; b < nb; b += tpr) {
        const int k = b - small_div(b, inv_ns) * ns;
        const int e = k * tmul;        // < N/R, so r e < N for every r < R: the index never wraps
        const int o = (b - k) * R + k;
        if constexpr (R >= 11 && U > 1) {       // 11, 13 with several rows: the twiddles fetched again per row (kept, they spill)
#pragma unroll 1
            for (int u = 0; u < U; ++u) {
                pk2 v[R];
                v[0] = pk(src[u * row_stride + b]);
#pragma unroll
                for (int r = 1; r < R; ++r) v[r] = pk_cmul(pk(src[u * row_stride + b + r * nb]), pk(tw[r * e]));
                dft_store<R>(v, rt, dst + u * row_stride + o, ns);
            }
        } else {
            pk2 w[R];
#pragma unroll
            for (int r = 1; r < R; ++r) w[r] = pk(tw[r * e]);
#pragma unroll R >= 7 ? 1 : U                  // the big butterflies one row at a time: U of them at once spill
            for (int u = 0; u < U; ++u) {
                pk2 v[R];
                v[0] = pk(src[u * row_stride + b]);
#pragma unroll
                for (int r = 1; r < R; ++r) v[r] = pk_cmul(pk(src[u * row_stride + b + r * nb]), w[r]);
                dft_store<R>(v, rt, dst + u * row_stride + o, ns);
            }
        }
    }
}

// one stage, any radix: item (b, q) = one output of one butterfly; the R inputs are read from the row each time
template <int U>
FX_HD void mixed_stage_any(const cf* src, cf* dst, int row_stride, const cf* tw, int n, int radix, int ns, int lt, int tpr) {
    const int nb = n / radix;
    const int tmul = nb / ns;
    const float inv_ns = 1.0f / (float)ns, inv_nb = 1.0f / (float)nb;
    for (int item = lt; item < n; item += tpr) {
        const int q = small_div(item, inv_nb), b = item - q * nb;
        const int k = b - small_div(b, inv_ns) * ns;
        const int e = k * tmul + q * nb;     // < N/R + (R-1) N/R = N
        pk2 acc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = pk(src[u * row_stride + b]);
        int idx = e;
        for (int r = 1; r < radix; ++r) {
            const pk2 w = pk(tw[idx]);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const pk2 a = pk(src[u * row_stride + b + r * nb]);
                acc[u] = pk_fma(pk_splat(a[1]), pk_muli(w), pk_fma(pk_splat(a[0]), w, acc[u]));
            }
            idx += e;
            if (idx >= n) idx -= n;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) dst[u * row_stride + (b - k) * radix + k + q * ns] = unpk(acc[u]);
    }
}

// BIG_REG: the register butterflies for 11 and 13 also with U > 1 (twiddles fetched again per row).  They cost ~8 VGPRs on the
// whole kernel -- the two-antenna kernel has them to spare (still four workgroups per CU; 1001 channels 5.34 -> 3.44 ms), the
// F-only kernel with two frames per slot would drop from five to four (360 channels 1.60 -> 1.74 ms) and goes without: the
// host gives channel counts with these factors one frame per slot there (mixed_rows_per_slot_cap).
template <int U, bool BIG_REG = (U == 1)>
FX_HD void mixed_stage(const cf* src, cf* dst, int row_stride, const cf* tw, int n, int radix, int ns, int lt, int tpr) {
    switch (radix) {
        case 2: mixed_stage_reg<2, U>(src, dst, row_stride, tw, n, ns, lt, tpr); break;
        case 3: mixed_stage_reg<3, U>(src, dst, row_stride, tw, n, ns, lt, tpr); break;
        case 4: mixed_stage_reg<4, U>(src, dst, row_stride, tw, n, ns, lt, tpr); break;
        case 5: mixed_stage_reg<5, U>(src, dst, row_stride, tw, n, ns, lt, tpr); break;
        case 7: mixed_stage_reg<7, U>(src, dst, row_stride, tw, n, ns, lt, tpr); break;
        case 11:
        case 13:
            if constexpr (BIG_REG) {
                if (radix == 11)
                    mixed_stage_reg<11, U>(src, dst, row_stride, tw, n, ns, lt, tpr);
                else
                    mixed_stage_reg<13, U>(src, dst, row_stride, tw, n, ns, lt, tpr);
                break;
            }
            [[fallthrough]];
        default: mixed_stage_any<U>(src, dst, row_stride, tw, n, radix, ns, lt, tpr); break;
    }
}

// most frames per slot the F-only kernel should take for this factorisation (see mixed_stage)
inline int mixed_rows_per_slot_cap(const MixedPlan& mp) {
    for (int s = 0; s < mp.n_stages; ++s)
        if (mp.radix[s] == 11 || mp.radix[s] == 13) return 1;
    return 2;
}

// threads that share one row: the power of two at or above N/4 within [4, 256] (one wave up to 384 channels with two antennas), and 512 or 1024
// of them only beyond `wide_from` = 1320 channels (the workgroup has max(256, that) threads)
inline int mixed_threads_per_row(int n, int cap = 1024, bool fused_x = false, int wide_from = 1320) {
    int t = 4;
    while (t < 256 && t < cap && t * 4 < n) t <<= 1;
    while (n > wide_from && t < cap && t * 4 < n) t <<= 1;
    // 257 .. 384 channels, two antennas: a wave per row (no workgroup barrier) beats 128 part-used threads (300 channels 3.44 ->
    // 2.42 ms); F only it is the other way round (360 channels 1.49 -> 1.65 ms)
    if (fused_x && t == 128 && n <= 384) t = 64;
    if (fused_x && t == 256 && n <= 640) t = 128;    // 513 .. 640: two slots of 128 (600 channels 3.78 -> 3.51 ms; from 700 on no more)
    return t;
}

}  // namespace fxc
