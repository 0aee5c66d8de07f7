// fx_tiled.h — per-thread phases of the fused 2-antenna F+X kernel for the other channel counts:
// nchan = R0 * 16^a in {512, 1024, 2048 (a = 2, R0 = 2/4/8), 4096 (a = 3, R0 = 1), 8192 (a = 3, R0 = 2)},
// any ntaps (the reference fixes ntaps = 4 but lets --nfft vary, effex/effex.py:115,778).
//
// One workgroup of nchan/8 threads = nchan/16 butterflies per antenna; lanes 0-31 of every wave carry
// antenna 0, lanes 32-63 antenna 1 of the same butterfly index u.  Decimation in frequency, in place:
//   FIR       thread u owns the 16 branches m = u + P r (P = nchan/16) and applies the window: v[r]
//   pre-stage R0-point DFTs over q of v[g + G q] (G = 16/R0), twiddle wN^((u + P g) k)       [registers]
//   stage A   (a = 3) radix-16 with stride 256, twiddle w4096^(n' k)                         [LDS, s_barrier]
//   stage B   radix-16 with stride 16, twiddle w256^(n' k)                                   [LDS, s_barrier]
//   stage C   radix-16 with stride 1 after a 16x16 transpose that stays inside the wave      [LDS, no barrier]
// Bin k = k0 + R0 (kA + 16 (k1 + 16 k2)) (a = 3) or k0 + R0 (k1 + 16 k2) (a = 2) ends at position
// 16 u + k2 with u = (k0, kA, k1) resp. (k0, k1); v_permlane32_swap then pairs the antennas as in
// fx_fused4096.h.  Exchange layout: position p sits at p + 16 (p >> 8) (every 256 positions padded to 272
// = 16 * 17), which makes the stride-256 and stride-16 accesses and the 17-pitch transpose conflict-free
// and keeps the transpose of a 16-lane group inside the 272 slots that group alone reads in stage B.
//
// Kernels built from these phases (fxcorr.hip): fx_tiled_ring_kernel (ntaps <= 4, nchan <= 4096: four frames in a
// VGPR ring + window quads in LDS, fir_ring), fx_tiled_kernel (any ntaps, nchan 8192: the FIR re-reads its history,
// two frames per pass), each as F+X (v_permlane32_swap X-stage), F-only (natural-order spectra through a
// transposition in the exchange region) and, for the ring kernel, uint8-ingest variants.
//
// The same source is compiled by g++ in tests/emul (host emulation; test infrastructure only).
#pragma once
#include "fx_fused4096.h"

namespace fxc {
namespace tiled {

using fxc::fused::lds_load;

FX_HD void dft2(cf& a, cf& b) {
    const cf t = csub(a, b);
    a = cadd(a, b);
    b = t;
}

// 8-point DFT, kernel exp(+2 pi i n k / 8), natural order in and out
FX_HD void dft8(cf& v0, cf& v1, cf& v2, cf& v3, cf& v4, cf& v5, cf& v6, cf& v7) {
    const float R2 = 0.70710678118654752440f;
    dft4(v0, v2, v4, v6);   // E[0..3] in v0, v2, v4, v6
    dft4(v1, v3, v5, v7);   // O[0..3] in v1, v3, v5, v7
    const cf o1 = mk((v3.x - v3.y) * R2, (v3.x + v3.y) * R2);      // w8^1 O[1]
    const cf o2 = muli(v5);                                        // w8^2 O[2]
    const cf o3 = mk((-v7.x - v7.y) * R2, (v7.x - v7.y) * R2);     // w8^3 O[3]
    const cf e0 = v0, e1 = v2, e2 = v4, e3 = v6, o0 = v1;
    v0 = cadd(e0, o0);
    v1 = cadd(e1, o1);
    v2 = cadd(e2, o2);
    v3 = cadd(e3, o3);
    v4 = csub(e0, o0);
    v5 = csub(e1, o1);
    v6 = csub(e2, o2);
    v7 = csub(e3, o3);
}

template <int R0_, bool A3_>
struct Geo {
    static constexpr int R0 = R0_;
    static constexpr bool A3 = A3_;
    static constexpr int N = R0 * (A3 ? 4096 : 256);
    static constexpr int P = N / 16;              // butterflies (threads) per antenna
    static constexpr int kThreads = 2 * P;
    static constexpr int G = 16 / R0;             // pre-stage groups per thread
    static constexpr int kRegion = N + N / 16;    // cf per antenna in the padded exchange layout
    static constexpr int kLdsRegion = 0;                          // cf[2][kRegion]
    static constexpr int kLdsTw16 = 2 * kRegion * 8;              // cf[256]  w256^(n' k) at [k*16 + n']
    static constexpr int kLdsBytes = kLdsTw16 + 256 * 8;
    static constexpr int kLdsWin = kLdsBytes;                     // ring variant only: f4[N]
    static constexpr int kLdsBytesRing = kLdsWin + N * 16;
    static constexpr int kAccPerThread = 8;

    static FX_HD int ant_of(int tid) { return (tid >> 5) & 1; }
    static FX_HD int u_of(int tid) { return (tid >> 6) * 32 + (tid & 31); }
    static FX_HD int pad16(int p) { return p + ((p >> 8) << 4); }

    // PFB FIR for frame i of one antenna stream xa (chunk start), taps t = 0 .. min(ntaps-1, i): zero history
    // per chunk; v[r] = sum_t h[t N + m] x[(i - t) N + N - 1 - m], m = u + P r   
    static FX_HD void fir(const cf* xa, const float* win, int u, int64_t i, int ntaps, cf (&v)[16]) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = mk(0.f, 0.f);
        const int tmax = (int64_t)(ntaps - 1) < i ? ntaps - 1 : (int)i;
        for (int t = 0; t <= tmax; ++t) {
            const cf* bx = xa + (i - t) * N + (N - 1 - u);
            const float* bh = win + (int64_t)t * N + u;
            cf xv[16];
            float hv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                xv[r] = bx[-P * r];
                hv[r] = bh[P * r];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = cfma(hv[r], xv[r], v[r]);
        }
    }

    // ntaps <= 4 variant: the four frames the FIR needs live in a VGPR ring (slot PH = frame i, (PH+3)&3 = i-1,
    // ...), the window in LDS as f4 quads [r P + u] = h[t N + u + P r], t = x, y, z, w (zero beyond ntaps)
    template <int PH>
    static FX_HD void fir_ring(const cf (&h)[4][16], const f4* win, int u, cf (&v)[16]) {
        const cf (&x0)[16] = h[PH];
        const cf (&x1)[16] = h[(PH + 3) & 3];
        const cf (&x2)[16] = h[(PH + 2) & 3];
        const cf (&x3)[16] = h[(PH + 1) & 3];
        f4 w[2][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) w[0][q] = win[q * P + u];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g < 3) {
#pragma unroll
                for (int q = 0; q < 4; ++q) w[(g + 1) & 1][q] = win[(4 * (g + 1) + q) * P + u];
            }
            FXC_SCHED_FENCE();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = 4 * g + q;
                const f4 t = w[g & 1][q];
                cf a = cscale(x0[r], t.x);
                a = cfma(t.y, x1[r], a);
                a = cfma(t.z, x2[r], a);
                v[r] = cfma(t.w, x3[r], a);
            }
            FXC_SCHED_FENCE();
        }
    }
    // element offset inside one frame of the sample feeding branch u + P r
    static FX_HD int sample_offset(int u, int r) { return (N - 1) - u - P * r; }

    // this thread's pre-stage twiddles from the [16][P] table wN^((u + P g) k) at slot g + G k
    static FX_HD void load_tw0(cf (&tw0)[16], const cf* table, int u) {
#pragma unroll
        for (int r = 0; r < 16; ++r) tw0[r] = table[r * P + u];
    }
    // stage-A twiddles from the [16][256] table w4096^(n' k)
    static FX_HD void load_twA(cf (&twA)[16], const cf* table, int u) {
#pragma unroll
        for (int k = 0; k < 16; ++k) twA[k] = table[k * 256 + (u & 255)];
    }

    static FX_HD void prestage(cf (&v)[16], const cf (&tw0)[16]) {
        if (R0 == 2) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                dft2(v[g], v[g + 8]);
                v[g + 8] = cmul(v[g + 8], tw0[g + 8]);
            }
        } else if (R0 == 4) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                dft4(v[g], v[g + 4], v[g + 8], v[g + 12]);
#pragma unroll
                for (int k = 1; k < 4; ++k) v[g + 4 * k] = cmul(v[g + 4 * k], tw0[g + 4 * k]);
            }
        } else if (R0 == 8) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                dft8(v[g], v[g + 2], v[g + 4], v[g + 6], v[g + 8], v[g + 10], v[g + 12], v[g + 14]);
#pragma unroll
                for (int k = 1; k < 8; ++k) v[g + 2 * k] = cmul(v[g + 2 * k], tw0[g + 2 * k]);
            }
        }
    }

    // exchange after the pre-stage: slot r' goes to position u + P r'
    static FX_HD void store0(const cf (&v)[16], cf* reg, int u) {
#pragma unroll
        for (int r = 0; r < 16; ++r) reg[pad16(u + P * r)] = v[r];
    }

    // stage A (a = 3): butterfly u = (block, n'), positions block*4096 + n' + 256 q
    static FX_HD void loadA(const cf* reg, int u, cf (&v)[16]) {
        const cf* b = reg + (u >> 8) * 4352 + (u & 255);
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = lds_load(b + 272 * q);
    }
    static FX_HD void twiddleA_store(cf (&v)[16], const cf (&twA)[16], cf* reg, int u) {
        cf* b = reg + (u >> 8) * 4352 + (u & 255);
        b[0] = v[0];
#pragma unroll
        for (int k = 1; k < 16; ++k) {
            v[k] = cmul(v[k], twA[k]);
            b[272 * k] = v[k];
        }
    }

    // stage B: butterfly u = (b16, n'), positions b16*256 + n' + 16 q
    static FX_HD void loadB(const cf* reg, int u, cf (&v)[16]) {
        const cf* b = reg + (u >> 4) * 272 + (u & 15);
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = lds_load(b + 16 * q);
    }
    static FX_HD void twiddleB(cf (&v)[16], const cf* tw16, int u) {
        const int n = u & 15;
        cf t[16];
#pragma unroll
        for (int k = 1; k < 16; ++k) t[k] = lds_load(tw16 + k * 16 + n);
#pragma unroll
        for (int k = 1; k < 16; ++k) v[k] = cmul(v[k], t[k]);
    }
    // 16x16 transpose inside the 16-lane group: out k of lane n' -> slot [k][n'] of the group's 272
    static FX_HD void storeT(const cf (&v)[16], cf* reg, int u) {
        cf* b = reg + (u >> 4) * 272 + (u & 15);
#pragma unroll
        for (int k = 0; k < 16; ++k) b[17 * k] = v[k];
    }
    // stage C: butterfly u reads positions 16 u + q
    static FX_HD void loadC(const cf* reg, int u, cf (&v)[16]) {
        const cf* b = reg + (u >> 4) * 272 + (u & 15) * 17;
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = lds_load(b + q);
    }

    // natural bin of output k2 of stage C in butterfly u
    static FX_HD int bin_of(int u, int k2) {
        if (A3) return (u >> 8) + R0 * (((u >> 4) & 15) + 16 * ((u & 15) + 16 * k2));
        return (u >> 4) + R0 * ((u & 15) + 16 * k2);
    }
};

}  // namespace tiled
}  // namespace fxc
