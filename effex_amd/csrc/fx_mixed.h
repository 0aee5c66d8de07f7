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
constexpr int kMixedMaxRegRadix = 13;    // primes up to here are register butterflies; larger ones walk the LDS row

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

// R-point DFT, kernel exp(+2 pi i q r / R), in place; `root` = tw + the stride N/R of the R-th roots in the table
template <int R>
FXC_HD void dft_reg(cf (&v)[R], const cf* root, int root_stride) {
    if constexpr (R == 2) {
        const cf a = v[0], b = v[1];
        v[0] = cadd(a, b);
        v[1] = csub(a, b);
    } else if constexpr (R == 4) {
        dft4(v[0], v[1], v[2], v[3]);
    } else {
        cf w[R];
#pragma unroll
        for (int m = 1; m < R; ++m) w[m] = root[m * root_stride];
        cf out[R];
#pragma unroll
        for (int q = 0; q < R; ++q) {
            cf acc = v[0];
#pragma unroll
            for (int r = 1; r < R; ++r) {
                if ((q * r) % R == 0) {
                    acc = cadd(acc, v[r]);
                } else {
                    const cf ww = w[(q * r) % R];
                    acc = mk(__builtin_fmaf(-v[r].y, ww.y, __builtin_fmaf(v[r].x, ww.x, acc.x)),
                             __builtin_fmaf(v[r].y, ww.x, __builtin_fmaf(v[r].x, ww.y, acc.y)));
                }
            }
            out[q] = acc;
        }
#pragma unroll
        for (int q = 0; q < R; ++q) v[q] = out[q];
    }
}

// one stage, register butterflies: "thread" lt of tpr walks the butterflies lt, lt + tpr, ...
template <int R>
FXC_HD void mixed_stage_reg(const cf* src, cf* dst, const cf* tw, int n, int ns, int lt, int tpr) {
    const int nb = n / R;
    const int tmul = nb / ns;          // N / (Ns R)
    for (int b = lt; b < nb; b += tpr) {
        const int k = b % ns;
        const int e = k * tmul;        // < N/R, so r e < N for every r < R: the index never wraps
        cf v[R];
        v[0] = src[b];
#pragma unroll
        for (int r = 1; r < R; ++r) v[r] = cmul(src[b + r * nb], tw[r * e]);
        dft_reg<R>(v, tw, nb);
        cf* d = dst + (b - k) * R + k;
#pragma unroll
        for (int q = 0; q < R; ++q) d[q * ns] = v[q];
    }
}

// one stage, any radix: item (b, q) = one output of one butterfly; the R inputs are read from the row each time
FXC_HD void mixed_stage_any(const cf* src, cf* dst, const cf* tw, int n, int radix, int ns, int lt, int tpr) {
    const int nb = n / radix;
    const int tmul = nb / ns;
    for (int item = lt; item < n; item += tpr) {
        const int q = item / nb, b = item - q * nb;
        const int k = b % ns;
        const int e = k * tmul + q * nb;     // < N/R + (R-1) N/R = N
        cf acc = src[b];
        int idx = e;
        for (int r = 1; r < radix; ++r) {
            const cf a = src[b + r * nb], w = tw[idx];
            acc = mk(__builtin_fmaf(-a.y, w.y, __builtin_fmaf(a.x, w.x, acc.x)),
                     __builtin_fmaf(a.y, w.x, __builtin_fmaf(a.x, w.y, acc.y)));
            idx += e;
            if (idx >= n) idx -= n;
        }
        dst[(b - k) * radix + k + q * ns] = acc;
    }
}

FXC_HD void mixed_stage(const cf* src, cf* dst, const cf* tw, int n, int radix, int ns, int lt, int tpr) {
    switch (radix) {
        case 2: mixed_stage_reg<2>(src, dst, tw, n, ns, lt, tpr); break;
        case 3: mixed_stage_reg<3>(src, dst, tw, n, ns, lt, tpr); break;
        case 4: mixed_stage_reg<4>(src, dst, tw, n, ns, lt, tpr); break;
        case 5: mixed_stage_reg<5>(src, dst, tw, n, ns, lt, tpr); break;
        case 7: mixed_stage_reg<7>(src, dst, tw, n, ns, lt, tpr); break;
        case 11: mixed_stage_reg<11>(src, dst, tw, n, ns, lt, tpr); break;
        case 13: mixed_stage_reg<13>(src, dst, tw, n, ns, lt, tpr); break;
        default: mixed_stage_any(src, dst, tw, n, radix, ns, lt, tpr); break;
    }
}

// threads of the 256-thread workgroup that share one row: the power of two at or above N/4, within [4, 256]
inline int mixed_threads_per_row(int n) {
    int t = 4;
    while (t < 256 && t * 4 < n) t <<= 1;
    return t;
}

}  // namespace fxc
