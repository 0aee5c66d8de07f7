// fx_math.h — complex helpers and the in-register radix-16 butterfly used by the FFT kernels.
// Compiles as device code under hipcc and as plain C++ under g++ (tests/emul builds the same
// source for a host emulation of the fused kernel's index logic — test infrastructure only).
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define FXC_HD __host__ __device__ __forceinline__
#define FXC_D __device__ __forceinline__
#else
#define FXC_HD inline
#define FXC_D inline
#endif

namespace fxc {

struct __attribute__((aligned(8))) cf {
    float x, y;
};

struct __attribute__((aligned(16))) f4 {
    float x, y, z, w;
};

struct __attribute__((aligned(16))) cd {
    double x, y;
};

// Frame-range work split of the fused kernel: n items over g workgroups, workgroup b owns [range_begin(b),
// range_begin(b + 1)); range_owner(f) is the workgroup whose range holds item f.
FXC_HD long long range_begin(long long b, long long n, long long g) { return b * n / g; }
FXC_HD long long range_owner(long long f, long long n, long long g) { return ((f + 1) * g + n - 1) / n - 1; }

FXC_HD cf mk(float x, float y) { cf r; r.x = x; r.y = y; return r; }
FXC_HD cf cadd(cf a, cf b) { return mk(a.x + b.x, a.y + b.y); }
FXC_HD cf csub(cf a, cf b) { return mk(a.x - b.x, a.y - b.y); }
FXC_HD cf cmul(cf a, cf b) { return mk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
// a * conj(b)
FXC_HD cf cmulc(cf a, cf b) { return mk(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }
FXC_HD cf cscale(cf a, float s) { return mk(a.x * s, a.y * s); }
// multiply by +i
FXC_HD cf muli(cf a) { return mk(-a.y, a.x); }
// a + s*b with real s
FXC_HD cf cfma(float s, cf b, cf a) { return mk(a.x + s * b.x, a.y + s * b.y); }

// 4-point DFT with kernel exp(+2*pi*i*n*k/4) (the channeliser's sign), in place.
FXC_HD void dft4(cf& a, cf& b, cf& c, cf& d) {
    cf t0 = cadd(a, c), t1 = csub(a, c), t2 = cadd(b, d), t3 = muli(csub(b, d));
    a = cadd(t0, t2);
    b = cadd(t1, t3);
    c = csub(t0, t2);
    d = csub(t1, t3);
}

// 16-point DFT, kernel exp(+2*pi*i*n*k/16), natural order in and out:
//   n = 4*n1 + n0, k = c + 4*d:  w16^(nk) = w4^(n1 c) * w16^(n0 c) * w4^(n0 d)
// stage A: four DFT4 over n1 and the internal twiddles; stage B: four DFT4 over n0 + reorder
FXC_HD void dft16_a(cf (&v)[16]) {
    const float C1 = 0.92387953251128673848f;  // cos(pi/8)
    const float S1 = 0.38268343236508978178f;  // sin(pi/8)
    const float R2 = 0.70710678118654752440f;  // sqrt(1/2)
#pragma unroll
    for (int n0 = 0; n0 < 4; ++n0) dft4(v[n0], v[4 + n0], v[8 + n0], v[12 + n0]);
    // now v[4*c + n0] = Z[n0][c]; twiddle by w16^(n0*c)
    v[5] = cmul(v[5], mk(C1, S1));                      // e = 1
    v[6] = mk((v[6].x - v[6].y) * R2, (v[6].x + v[6].y) * R2);    // e = 2
    v[7] = cmul(v[7], mk(S1, C1));                      // e = 3
    v[9] = mk((v[9].x - v[9].y) * R2, (v[9].x + v[9].y) * R2);    // e = 2
    v[10] = muli(v[10]);                                // e = 4
    v[11] = mk((-v[11].x - v[11].y) * R2, (v[11].x - v[11].y) * R2);  // e = 6
    v[13] = cmul(v[13], mk(S1, C1));                    // e = 3
    v[14] = mk((-v[14].x - v[14].y) * R2, (v[14].x - v[14].y) * R2);  // e = 6
    v[15] = cmul(v[15], mk(-C1, -S1));                  // e = 9
}

FXC_HD void dft16_b(cf (&v)[16]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) dft4(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
    // now v[4*c + d] = Y[c + 4*d]; transpose the 4x4 index to natural order
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int d = c + 1; d < 4; ++d) {
            cf t = v[4 * c + d];
            v[4 * c + d] = v[4 * d + c];
            v[4 * d + c] = t;
        }
}

FXC_HD void dft16(cf (&v)[16]) {
    dft16_a(v);
    dft16_b(v);
}

}  // namespace fxc
