// fx_small.h — per-lane phases of the fused 2-antenna F+X kernel for the small channel counts nchan = 16 P,
// P in {1, 2, 4, 8, 16} (k_small.h has the kernel and the description of the work split).
//
// P adjacent lanes make one transform; lane u owns the 16 branches m = u + P r.  Decimation in frequency, bin k = k1 + 16 k2:
//   fir_ring     v[r] = sum_t h[t N + m] x[(i - t) N + N - 1 - m] from a four-frame VGPR ring, window quads [r P + u]
//   dft16        Y[u][k1] = sum_r v[r] w16^(r k1)                                               [registers]
//   twiddle      Y[u][k1] *= wN^(u k1), table [u][k1]
//   store/load   transposition inside the P lanes through rows of 17 P slots (f = k1 P + u at f + (f >> 4)): lane j
//                takes the 16 / P values k1 = j 16/P + t with all their u                        [LDS, no barrier]
//   transforms   16 / P transforms of P points over u: v[t P + k2] = bin (j 16/P + t) + 16 k2    [registers]
//
// The same source is compiled by g++ in tests/emul (host emulation; test infrastructure only).
#pragma once
#include "fx_tiled.h"

namespace fxc {
namespace small {

// a window quad, read where it is used: the quads are the same for every frame of a lane, and hoisted out of the frame
// loop they would take 64 VGPRs the ring needs
FX_HD f4 quad_load(const f4* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef float v4f __attribute__((ext_vector_type(4)));
    typedef const volatile __attribute__((address_space(3))) v4f* lds_v4f_ptr;
    const v4f q = *(lds_v4f_ptr)(p);
    f4 r;
    r.x = q[0];
    r.y = q[1];
    r.z = q[2];
    r.w = q[3];
    return r;
#else
    return *p;
#endif
}

template <int P_>
struct Geo {
    static constexpr int P = P_;
    static constexpr int N = 16 * P;
    static constexpr int kSub = 32 / P;                 // work items per antenna half of a wave
    static constexpr int kWaves = 4;
    static constexpr int kThreads = 64 * kWaves;
    static constexpr int kItemsPerWg = kWaves * kSub;
    static constexpr int kGroup = 17 * P;               // cf per item and antenna in the exchange rows
    static constexpr int kXchgPerWave = (64 / P) * kGroup;

    // element offset inside one frame of the sample feeding branch u + P r
    static FX_HD int sample_offset(int u, int r) { return (N - 1) - u - P * r; }

    // frame i sits in ring slot PH, (PH + 3) & 3 holds i - 1, ...; window quads [r P + u] = h[t N + u + P r], t = x, y, z, w
    template <int PH>
    static FX_HD void fir_ring(const cf (&h)[4][16], const f4* win, int u, cf (&v)[16]) {
        const cf (&x0)[16] = h[PH];
        const cf (&x1)[16] = h[(PH + 3) & 3];
        const cf (&x2)[16] = h[(PH + 2) & 3];
        const cf (&x3)[16] = h[(PH + 1) & 3];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const f4 t = quad_load(win + r * P + u);
            cf a = cscale(x0[r], t.x);
            a = cfma(t.y, x1[r], a);
            a = cfma(t.z, x2[r], a);
            v[r] = cfma(t.w, x3[r], a);
        }
    }

    static FX_HD void twiddle(cf (&v)[16], const cf* tw, int u) {
        cf t[16];
#pragma unroll
        for (int k = 1; k < 16; ++k) t[k] = fxc::fused::lds_load(tw + u * 16 + k);       // wN^(u k)
#pragma unroll
        for (int k = 1; k < 16; ++k) v[k] = cmul(v[k], t[k]);
    }

    // grp: the rows of this item and antenna
    static FX_HD void store(const cf (&v)[16], cf* grp, int u) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int f = k * P + u;
            grp[f + (f >> 4)] = v[k];
        }
    }
    static FX_HD void load(const cf* grp, int u, cf (&v)[16]) {
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = fxc::fused::lds_load(grp + 17 * u + q);
    }

    static FX_HD void transforms(cf (&v)[16]) {
        if (P == 16) {
            dft16(v);
        } else if (P == 8) {
            tiled::dft8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
            tiled::dft8(v[8], v[9], v[10], v[11], v[12], v[13], v[14], v[15]);
        } else if (P == 4) {
#pragma unroll
            for (int t = 0; t < 4; ++t) dft4(v[4 * t], v[4 * t + 1], v[4 * t + 2], v[4 * t + 3]);
        } else if (P == 2) {
#pragma unroll
            for (int t = 0; t < 8; ++t) tiled::dft2(v[2 * t], v[2 * t + 1]);
        }
    }

    // natural bin of value idx (0 .. 15) of lane u after `transforms`
    static FX_HD int bin_of(int u, int idx) { return u * (16 / P) + idx / P + 16 * (idx % P); }
};

}  // namespace small
}  // namespace fxc
