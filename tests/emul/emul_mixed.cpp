// Host emulation of the generic path's mixed-radix FFT (effex_amd/csrc/fx_mixed.h) — TEST INFRASTRUCTURE ONLY.
// Runs the `tpr` "threads" that share a row stage by stage (a barrier = the end of the loop over threads), so the
// Stockham indexing, the twiddle indices and the factorisation can be checked against numpy without a GPU.
#include <cmath>
#include <cstdint>
#include <vector>

#include "../../effex_amd/csrc/fx_mixed.h"

using namespace fxc;

extern "C" int emul_mixed_fft(const float* in, float* out, int n, int tpr, int* radices_out) {
    const double two_pi = 6.283185307179586476925286766559;
    std::vector<cf> tw((size_t)n);
    for (int j = 0; j < n; ++j) {
        const double ph = two_pi * (double)j / (double)n;
        tw[j] = mk((float)std::cos(ph), (float)std::sin(ph));
    }
    const MixedPlan mp = mixed_factor(n);
    if (mp.n_stages < 0) return -1;
    long long prod = 1;
    for (int s = 0; s < mp.n_stages; ++s) {
        prod *= mp.radix[s];
        if (radices_out) radices_out[s] = mp.radix[s];
    }
    if (prod != n) return -2;
    std::vector<cf> a((size_t)n), b((size_t)n);
    for (int j = 0; j < n; ++j) a[j] = mk(in[2 * j], in[2 * j + 1]);
    cf *src = a.data(), *dst = b.data();
    int ns = 1;
    for (int s = 0; s < mp.n_stages; ++s) {
        for (int lt = 0; lt < tpr; ++lt) mixed_stage<1>(src, dst, 0, tw.data(), n, mp.radix[s], ns, lt, tpr);
        ns *= mp.radix[s];
        cf* t = src;
        src = dst;
        dst = t;
    }
    for (int j = 0; j < n; ++j) {
        out[2 * j] = src[j].x;
        out[2 * j + 1] = src[j].y;
    }
    return mp.n_stages;
}

// two rows a slot carries together (U = 2, the register butterflies for 11 and 13 included): rows row_stride apart, as in LDS
extern "C" int emul_mixed_fft_two_rows(const float* in, float* out, int n, int tpr) {
    const double two_pi = 6.283185307179586476925286766559;
    std::vector<cf> tw((size_t)n);
    for (int j = 0; j < n; ++j) {
        const double ph = two_pi * (double)j / (double)n;
        tw[j] = mk((float)std::cos(ph), (float)std::sin(ph));
    }
    const MixedPlan mp = mixed_factor(n);
    if (mp.n_stages < 0) return -1;
    const int row_stride = 2 * n;                       // [row][a | b][n]
    std::vector<cf> rows((size_t)2 * row_stride);
    for (int u = 0; u < 2; ++u)
        for (int j = 0; j < n; ++j) rows[(size_t)u * row_stride + j] = mk(in[2 * (u * n + j)], in[2 * (u * n + j) + 1]);
    int so = 0, ns = 1;
    for (int s = 0; s < mp.n_stages; ++s) {
        for (int lt = 0; lt < tpr; ++lt)
            mixed_stage<2, true>(rows.data() + so, rows.data() + (n - so), row_stride, tw.data(), n, mp.radix[s], ns, lt, tpr);
        ns *= mp.radix[s];
        so = n - so;
    }
    for (int u = 0; u < 2; ++u)
        for (int j = 0; j < n; ++j) {
            out[2 * (u * n + j)] = rows[(size_t)u * row_stride + so + j].x;
            out[2 * (u * n + j) + 1] = rows[(size_t)u * row_stride + so + j].y;
        }
    return mp.n_stages;
}

extern "C" int emul_mixed_threads_per_row(int n) { return mixed_threads_per_row(n); }
