// fx_math.h — complex helpers and the in-register radix-16 butterfly used by the FFT kernels.
// Compiles as device code under hipcc and as plain C++ under g++ (tests/emul builds the same
// source for a host emulation of the fused kernel's index logic — test infrastructure only).
#pragma once

#if defined(__HIPCC__)
#if !defined(__HIPCC_RTC__)      // (hiprtc brings the runtime's declarations itself: fx_spec.h is compiled through it)
#include <hip/hip_runtime.h>
#endif
#define FX_HD __host__ __device__ __forceinline__
#define FX_D __device__ __forceinline__
#else
#define FX_HD inline
#define FX_D inline
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
FX_HD int range_begin(int b, int n, int g) { return (int)((long long)b * n / g); }
FX_HD int range_owner(int f, int n, int g) { return (int)((((long long)f + 1) * g + n - 1) / n - 1); }

FX_HD cf mk(float x, float y) { cf r; r.x = x; r.y = y; return r; }

// Streaming accesses -- data a kernel touches once (spectra on their way from an F pass to an X pass): the nontemporal hint keeps
// them from displacing what the caches are for.  Host build: plain accesses.
FX_HD cf nt_load(const cf* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef float nt_v2f __attribute__((ext_vector_type(2)));
    const nt_v2f v = __builtin_nontemporal_load(reinterpret_cast<const nt_v2f*>(p));
    cf r;
    r.x = v[0];
    r.y = v[1];
    return r;
#else
    return *p;
#endif
}
FX_HD void nt_store(cf* p, cf v);
// (the spectra of the any-shape and per-channel-count F passes and the two-pass 8192 route keep the default policy: nontemporal there
// measured neutral to 4 % slower, profiles/r05/experiments.md 10)
FX_HD cf st_load(const cf* p) { return *p; }
FX_HD void st_store(cf* p, cf v) { *p = v; }
FX_HD void nt_store(cf* p, cf v) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef float nt_v2f __attribute__((ext_vector_type(2)));
    nt_v2f w;
    w[0] = v.x;
    w[1] = v.y;
    __builtin_nontemporal_store(w, reinterpret_cast<nt_v2f*>(p));
#else
    *p = v;
#endif
}
FX_HD cf cadd(cf a, cf b) { return mk(a.x + b.x, a.y + b.y); }
FX_HD cf csub(cf a, cf b) { return mk(a.x - b.x, a.y - b.y); }
FX_HD cf cmul(cf a, cf b) { return mk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
// a * conj(b)
FX_HD cf cmulc(cf a, cf b) { return mk(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }
// acc + a * conj(b) as four fused multiply-adds (two per component, no separate multiply or add)
FX_HD cf cmulc_acc(cf acc, cf a, cf b) {
    return mk(__builtin_fmaf(a.y, b.y, __builtin_fmaf(a.x, b.x, acc.x)), __builtin_fmaf(-a.x, b.y, __builtin_fmaf(a.y, b.x, acc.y)));
}
FX_HD cf cscale(cf a, float s) { return mk(a.x * s, a.y * s); }
// multiply by +i
FX_HD cf muli(cf a) { return mk(-a.y, a.x); }
// a + s*b with real s
FX_HD cf cfma(float s, cf b, cf a) { return mk(a.x + s * b.x, a.y + s * b.y); }

// 4-point DFT with kernel exp(+2*pi*i*n*k/4) (the channeliser's sign), in place.
FX_HD void dft4(cf& a, cf& b, cf& c, cf& d) {
    cf t0 = cadd(a, c), t1 = csub(a, c), t2 = cadd(b, d), t3 = muli(csub(b, d));
    a = cadd(t0, t2);
    b = cadd(t1, t3);
    c = csub(t0, t2);
    d = csub(t1, t3);
}

// 16-point DFT, kernel exp(+2*pi*i*n*k/16), natural order in and out:
//   n = 4*n1 + n0, k = c + 4*d:  w16^(nk) = w4^(n1 c) * w16^(n0 c) * w4^(n0 d)
// stage A: four DFT4 over n1 and the internal twiddles; stage B: four DFT4 over n0 + reorder.
// No internal twiddle costs a multiply of its own: stage A leaves them as un-scaled rotations and stage B folds the
// common scale into the fused multiply-adds of its butterflies (144 instructions in all = the 144 additions of a
// split-radix DFT-16, with its 24 multiplications riding along):
//   w16^2, w16^6 = sqrt(1/2) (+-1 +- i):   u = (x -+ y, x +- y), scale R2
//   w16^1, w16^3, w16^9 = C1 (1 + i t), C1 (t + i), -C1 (1 + i t) with C1 = cos(pi/8), t = tan(pi/8): two FMAs
//   each for u, scale C1 -- and the two twiddled inputs of one butterfly share that scale.
FX_HD void dft16_a(cf (&v)[16]) {
    const float T1 = 0.41421356237309504880f;  // tan(pi/8)
#pragma unroll
    for (int n0 = 0; n0 < 4; ++n0) dft4(v[n0], v[4 + n0], v[8 + n0], v[12 + n0]);
    // now v[4*c + n0] = Z[n0][c]; twiddle by w16^(n0*c), scales left to stage B
    v[5] = mk(__builtin_fmaf(-T1, v[5].y, v[5].x), __builtin_fmaf(T1, v[5].x, v[5].y));        // e = 1: (1 + i t) / C1
    v[6] = mk(v[6].x - v[6].y, v[6].x + v[6].y);                                               // e = 2, times sqrt 2
    v[7] = mk(__builtin_fmaf(T1, v[7].x, -v[7].y), __builtin_fmaf(T1, v[7].y, v[7].x));        // e = 3: (t + i) / C1
    v[9] = mk(v[9].x - v[9].y, v[9].x + v[9].y);                                               // e = 2, times sqrt 2
    v[10] = muli(v[10]);                                                                       // e = 4
    v[11] = mk(-v[11].x - v[11].y, v[11].x - v[11].y);                                         // e = 6, times sqrt 2
    v[13] = mk(__builtin_fmaf(T1, v[13].x, -v[13].y), __builtin_fmaf(T1, v[13].y, v[13].x));   // e = 3: (t + i) / C1
    v[14] = mk(-v[14].x - v[14].y, v[14].x - v[14].y);                                         // e = 6, times sqrt 2
    v[15] = mk(__builtin_fmaf(T1, v[15].y, -v[15].x), __builtin_fmaf(-T1, v[15].x, -v[15].y)); // e = 9: -(1 + i t) / C1
}

// dft4 whose third input arrives as c / R2 and whose second and fourth as b / C1, d / C1
FX_HD void dft4_scaled_c_bd(cf& a, cf& b, cf& c, cf& d) {
    const float R2 = 0.70710678118654752440f;  // sqrt(1/2)
    const float C1 = 0.92387953251128673848f;  // cos(pi/8)
    cf t0 = mk(__builtin_fmaf(R2, c.x, a.x), __builtin_fmaf(R2, c.y, a.y));
    cf t1 = mk(__builtin_fmaf(-R2, c.x, a.x), __builtin_fmaf(-R2, c.y, a.y));
    cf t2 = cadd(b, d), t3 = muli(csub(b, d));
    a = mk(__builtin_fmaf(C1, t2.x, t0.x), __builtin_fmaf(C1, t2.y, t0.y));
    b = mk(__builtin_fmaf(C1, t3.x, t1.x), __builtin_fmaf(C1, t3.y, t1.y));
    c = mk(__builtin_fmaf(-C1, t2.x, t0.x), __builtin_fmaf(-C1, t2.y, t0.y));
    d = mk(__builtin_fmaf(-C1, t3.x, t1.x), __builtin_fmaf(-C1, t3.y, t1.y));
}
// ... second and fourth inputs arrive as b / R2, d / R2: t2 = R2 (b + d), t3 = i R2 (b - d), folded into the outputs
FX_HD void dft4_bd_scaled(cf& a, cf& b, cf& c, cf& d) {
    const float R2 = 0.70710678118654752440f;
    cf t0 = cadd(a, c), t1 = csub(a, c), t2 = cadd(b, d), t3 = muli(csub(b, d));
    a = mk(__builtin_fmaf(R2, t2.x, t0.x), __builtin_fmaf(R2, t2.y, t0.y));
    b = mk(__builtin_fmaf(R2, t3.x, t1.x), __builtin_fmaf(R2, t3.y, t1.y));
    c = mk(__builtin_fmaf(-R2, t2.x, t0.x), __builtin_fmaf(-R2, t2.y, t0.y));
    d = mk(__builtin_fmaf(-R2, t3.x, t1.x), __builtin_fmaf(-R2, t3.y, t1.y));
}

FX_HD void dft16_b(cf (&v)[16]) {
    // after stage A: v[5], v[7], v[13], v[15] lack the factor C1, v[6], v[9], v[11], v[14] the factor R2
    dft4(v[0], v[1], v[2], v[3]);
    dft4_scaled_c_bd(v[4], v[5], v[6], v[7]);
    dft4_bd_scaled(v[8], v[9], v[10], v[11]);
    dft4_scaled_c_bd(v[12], v[13], v[14], v[15]);
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

// Stage B with every butterfly's four outputs handed to sink(k, Y[k]) (k = c + 4 d) as soon as they exist, so that
// a caller can let its stores trickle out between the remaining butterflies instead of after all of them
template <class Sink>
FX_HD void dft16_b_stream(cf (&v)[16], Sink&& sink) {
    dft4(v[0], v[1], v[2], v[3]);
#pragma unroll
    for (int d = 0; d < 4; ++d) sink(4 * d, v[d]);
    dft4_scaled_c_bd(v[4], v[5], v[6], v[7]);
#pragma unroll
    for (int d = 0; d < 4; ++d) sink(1 + 4 * d, v[4 + d]);
    dft4_bd_scaled(v[8], v[9], v[10], v[11]);
#pragma unroll
    for (int d = 0; d < 4; ++d) sink(2 + 4 * d, v[8 + d]);
    dft4_scaled_c_bd(v[12], v[13], v[14], v[15]);
#pragma unroll
    for (int d = 0; d < 4; ++d) sink(3 + 4 * d, v[12 + d]);
}

FX_HD void dft16(cf (&v)[16]) {
    dft16_a(v);
    dft16_b(v);
}

}  // namespace fxc
