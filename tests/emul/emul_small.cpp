// Host emulation of the wave-local kernel's phases (effex_amd/csrc/fx_small.h) — TEST INFRASTRUCTURE ONLY.
// Runs the 2 P lanes of one work item (P per antenna) phase by phase over all frames of one chunk pair, so the
// decomposition, the exchange rows and the bin mapping can be checked against the oracle without a GPU.
#include <cmath>
#include <cstdint>
#include <vector>

#include "../../effex_amd/csrc/fx_small.h"

using namespace fxc;

template <int P>
static int run(const float* x, int64_t num_samp, int ntaps, const double* window, double* out_sum) {
    using G = small::Geo<P>;
    constexpr int N = G::N;
    if (ntaps > 4) return -3;
    const int64_t n_pts = num_samp / N;
    const double two_pi = 6.283185307179586476925286766559;
    std::vector<f4> win4(N);
    for (int r = 0; r < 16; ++r)
        for (int u = 0; u < P; ++u) {
            const int m = u + P * r;
            f4 w;
            w.x = (float)window[m];
            w.y = ntaps > 1 ? (float)window[(size_t)1 * N + m] : 0.f;
            w.z = ntaps > 2 ? (float)window[(size_t)2 * N + m] : 0.f;
            w.w = ntaps > 3 ? (float)window[(size_t)3 * N + m] : 0.f;
            win4[(size_t)r * P + u] = w;
        }
    std::vector<cf> tw(N);
    for (int u = 0; u < P; ++u)
        for (int k1 = 0; k1 < 16; ++k1) {
            const double ph = two_pi * (double)((u * k1) % N) / (double)N;
            tw[(size_t)u * 16 + k1] = mk((float)std::cos(ph), (float)std::sin(ph));
        }
    const int L = 2 * P;                                   // lanes of the item: antenna 0 first
    std::vector<cf> ring((size_t)L * 64, mk(0.f, 0.f)), vbuf((size_t)L * 16), acc((size_t)L * 8, mk(0.f, 0.f));
    std::vector<cf> rows((size_t)2 * G::kGroup);
    const cf* xc = reinterpret_cast<const cf*>(x);
    auto V = [&](int l) -> cf(&)[16] { return *reinterpret_cast<cf(*)[16]>(&vbuf[(size_t)l * 16]); };
    for (int64_t i = 0; i < n_pts; ++i) {
        for (int l = 0; l < L; ++l) {
            const int ant = l / P, u = l % P;
            cf(&h)[4][16] = *reinterpret_cast<cf(*)[4][16]>(&ring[(size_t)l * 64]);
            for (int r = 0; r < 16; ++r) h[i & 3][r] = xc[ant * num_samp + i * N + G::sample_offset(u, r)];
            switch (i & 3) {
                case 0: G::template fir_ring<0>(h, win4.data(), u, V(l)); break;
                case 1: G::template fir_ring<1>(h, win4.data(), u, V(l)); break;
                case 2: G::template fir_ring<2>(h, win4.data(), u, V(l)); break;
                default: G::template fir_ring<3>(h, win4.data(), u, V(l)); break;
            }
            dft16(V(l));
            if (P > 1) G::twiddle(V(l), tw.data(), u);
        }
        if (P > 1) {
            for (int l = 0; l < L; ++l) G::store(V(l), rows.data() + (l / P) * G::kGroup, l % P);
            for (int l = 0; l < L; ++l) {
                G::load(rows.data() + (l / P) * G::kGroup, l % P, V(l));
                G::transforms(V(l));
            }
        }
        for (int u = 0; u < P; ++u) {                     // v_permlane32_swap: lane u gets values 0-7, lane u + 32 values 8-15
            const int lo = u, hi = P + u;
            for (int q = 0; q < 8; ++q) {
                acc[(size_t)lo * 8 + q] = cadd(acc[(size_t)lo * 8 + q], cmulc(V(lo)[q], V(hi)[q]));
                acc[(size_t)hi * 8 + q] = cadd(acc[(size_t)hi * 8 + q], cmulc(V(lo)[q + 8], V(hi)[q + 8]));
            }
        }
    }
    std::vector<int> seen(N, 0);
    for (int l = 0; l < L; ++l)
        for (int q = 0; q < 8; ++q) {
            const int k = G::bin_of(l % P, q + 8 * (l / P));
            if (k < 0 || k >= N || seen[k]++) return -1;
            out_sum[2 * k] = acc[(size_t)l * 8 + q].x;
            out_sum[2 * k + 1] = acc[(size_t)l * 8 + q].y;
        }
    return 0;
}

extern "C" int emul_small(const float* x, int64_t num_samp, int nchan, int ntaps, const double* window, double* out_sum) {
    switch (nchan) {
        case 16: return run<1>(x, num_samp, ntaps, window, out_sum);
        case 32: return run<2>(x, num_samp, ntaps, window, out_sum);
        case 64: return run<4>(x, num_samp, ntaps, window, out_sum);
        case 128: return run<8>(x, num_samp, ntaps, window, out_sum);
        case 256: return run<16>(x, num_samp, ntaps, window, out_sum);
        default: return -2;
    }
}
