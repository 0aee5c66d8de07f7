// Host emulation of the tiled fused kernel's phases (effex_amd/csrc/fx_tiled.h) — TEST INFRASTRUCTURE ONLY.
// Runs the nchan/8 "threads" of one workgroup phase by phase (a barrier = the end of a loop over
// threads; the wave-local transpose is emulated per wave), so the FFT decomposition, the padded LDS
// layout and the bin mapping can be checked against the oracle without a GPU.
#include <cmath>
#include <cstdint>
#include <vector>

#include "../../effex_amd/csrc/fx_tiled.h"

using namespace fxc;

template <class G>
static int run(const float* x, int64_t num_samp, int ntaps, const double* window, double* out_sum, int ring) {
    constexpr int N = G::N, P = G::P, TH = G::kThreads;
    const int64_t n_pts = num_samp / N;
    const double two_pi = 6.283185307179586476925286766559;
    std::vector<float> win((size_t)ntaps * N);
    for (size_t k = 0; k < win.size(); ++k) win[k] = (float)window[k];
    std::vector<cf> tw0(16 * P), twA(16 * 256), tw16(256);
    for (int r = 0; r < 16; ++r)
        for (int u = 0; u < P; ++u) {
            const int g = r % G::G, k = r / G::G;
            const double ph = two_pi * (double)(((int64_t)(u + P * g) * k) % N) / (double)N;
            tw0[r * P + u] = mk((float)std::cos(ph), (float)std::sin(ph));
        }
    for (int k = 0; k < 16; ++k)
        for (int n = 0; n < 256; ++n) {
            const double ph = two_pi * (double)((n * k) % 4096) / 4096.0;
            twA[k * 256 + n] = mk((float)std::cos(ph), (float)std::sin(ph));
        }
    for (int k = 0; k < 16; ++k)
        for (int n = 0; n < 16; ++n) {
            const double ph = two_pi * (double)(n * k) / 256.0;
            tw16[k * 16 + n] = mk((float)std::cos(ph), (float)std::sin(ph));
        }
    std::vector<cf> region(2 * G::kRegion);
    std::vector<cf> vbuf((size_t)TH * 16), t0((size_t)TH * 16), tA((size_t)TH * 16), acc((size_t)TH * 8, mk(0.f, 0.f));
    // ring variant (ntaps <= 4): window quads [r P + u] and a four-frame ring per thread, as the kernel keeps them
    std::vector<f4> win4(N);
    std::vector<cf> ringbuf(ring ? (size_t)TH * 64 : 0, mk(0.f, 0.f));
    if (ring) {
        if (ntaps > 4) return -3;
        for (int r = 0; r < 16; ++r)
            for (int u = 0; u < P; ++u) {
                const int m = u + P * r;
                f4 w;
                w.x = win[m];
                w.y = ntaps > 1 ? win[(size_t)1 * N + m] : 0.f;
                w.z = ntaps > 2 ? win[(size_t)2 * N + m] : 0.f;
                w.w = ntaps > 3 ? win[(size_t)3 * N + m] : 0.f;
                win4[(size_t)r * P + u] = w;
            }
    }
    const cf* xc = reinterpret_cast<const cf*>(x);
    auto V = [&](int tid) -> cf(&)[16] { return *reinterpret_cast<cf(*)[16]>(&vbuf[(size_t)tid * 16]); };
    for (int tid = 0; tid < TH; ++tid) {
        G::load_tw0(*reinterpret_cast<cf(*)[16]>(&t0[(size_t)tid * 16]), tw0.data(), G::u_of(tid));
        G::load_twA(*reinterpret_cast<cf(*)[16]>(&tA[(size_t)tid * 16]), twA.data(), G::u_of(tid));
    }
    for (int64_t i = 0; i < n_pts; ++i) {
        for (int tid = 0; tid < TH; ++tid) {
            const int u = G::u_of(tid), ant = G::ant_of(tid);
            if (ring) {
                cf(&h)[4][16] = *reinterpret_cast<cf(*)[4][16]>(&ringbuf[(size_t)tid * 64]);
                for (int r = 0; r < 16; ++r) h[i & 3][r] = xc[ant * num_samp + i * N + G::sample_offset(u, r)];
                switch (i & 3) {
                    case 0: G::template fir_ring<0>(h, win4.data(), u, V(tid)); break;
                    case 1: G::template fir_ring<1>(h, win4.data(), u, V(tid)); break;
                    case 2: G::template fir_ring<2>(h, win4.data(), u, V(tid)); break;
                    default: G::template fir_ring<3>(h, win4.data(), u, V(tid)); break;
                }
            } else {
                G::fir(xc + ant * num_samp, win.data(), u, i, ntaps, V(tid));
            }
            if (G::R0 > 1) G::prestage(V(tid), *reinterpret_cast<cf(*)[16]>(&t0[(size_t)tid * 16]));
        }
        if (G::A3) {
            if (G::R0 > 1) {
                for (int tid = 0; tid < TH; ++tid) G::store0(V(tid), region.data() + G::ant_of(tid) * G::kRegion, G::u_of(tid));
                for (int tid = 0; tid < TH; ++tid) G::loadA(region.data() + G::ant_of(tid) * G::kRegion, G::u_of(tid), V(tid));
            }
            for (int tid = 0; tid < TH; ++tid) dft16(V(tid));
            for (int tid = 0; tid < TH; ++tid)
                G::twiddleA_store(V(tid), *reinterpret_cast<cf(*)[16]>(&tA[(size_t)tid * 16]),
                                  region.data() + G::ant_of(tid) * G::kRegion, G::u_of(tid));
        } else {
            for (int tid = 0; tid < TH; ++tid) G::store0(V(tid), region.data() + G::ant_of(tid) * G::kRegion, G::u_of(tid));
        }
        for (int wave = 0; wave < TH / 64; ++wave) {
            for (int l = 0; l < 64; ++l) {
                const int tid = wave * 64 + l;
                G::loadB(region.data() + G::ant_of(tid) * G::kRegion, G::u_of(tid), V(tid));
                dft16(V(tid));
                G::twiddleB(V(tid), tw16.data(), G::u_of(tid));
            }
            for (int l = 0; l < 64; ++l) {
                const int tid = wave * 64 + l;
                G::storeT(V(tid), region.data() + G::ant_of(tid) * G::kRegion, G::u_of(tid));
            }
            for (int l = 0; l < 64; ++l) {
                const int tid = wave * 64 + l;
                G::loadC(region.data() + G::ant_of(tid) * G::kRegion, G::u_of(tid), V(tid));
                dft16(V(tid));
            }
            for (int l = 0; l < 32; ++l) {
                const int lo = wave * 64 + l, hi = lo + 32;
                for (int q = 0; q < 8; ++q) {
                    acc[(size_t)lo * 8 + q] = cadd(acc[(size_t)lo * 8 + q], cmulc(V(lo)[q], V(hi)[q]));
                    acc[(size_t)hi * 8 + q] = cadd(acc[(size_t)hi * 8 + q], cmulc(V(lo)[q + 8], V(hi)[q + 8]));
                }
            }
        }
    }
    std::vector<int> seen(N, 0);
    for (int tid = 0; tid < TH; ++tid)
        for (int q = 0; q < 8; ++q) {
            const int k = G::bin_of(G::u_of(tid), q + 8 * G::ant_of(tid));
            if (k < 0 || k >= N || seen[k]++) return -1;
            out_sum[2 * k] = acc[(size_t)tid * 8 + q].x;
            out_sum[2 * k + 1] = acc[(size_t)tid * 8 + q].y;
        }
    return 0;
}

extern "C" int emul_tiled(const float* x, int64_t num_samp, int nchan, int ntaps, const double* window, double* out_sum,
                          int ring) {
    switch (nchan) {
        case 512: return run<tiled::Geo<2, false>>(x, num_samp, ntaps, window, out_sum, ring);
        case 1024: return run<tiled::Geo<4, false>>(x, num_samp, ntaps, window, out_sum, ring);
        case 2048: return run<tiled::Geo<8, false>>(x, num_samp, ntaps, window, out_sum, ring);
        case 4096: return run<tiled::Geo<1, true>>(x, num_samp, ntaps, window, out_sum, ring);
        case 8192: return run<tiled::Geo<2, true>>(x, num_samp, ntaps, window, out_sum, ring);
        default: return -2;
    }
}
