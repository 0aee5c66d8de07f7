// Part of libfxcorr's single translation unit: included by fxcorr.hip (not a stand-alone header).
#pragma once

namespace {

// ------------------------------------------------------------------------------------------
// delay calibration — effex/effex.py:583-627: zero-padded FFT cross-correlation,
// arg-max of |xcorr|, 3-point log-Gaussian peak.  Runs once per calibration: the FFT is a global-memory
// Stockham autosort with radix-16 passes (the DFT-16 of fx_math.h in registers, twiddles formed in float64)
// and one radix-8 / 4 / 2 pass for what is left of log2 L -- 5 passes for the reference's 2 x 262144 points
// (19 with the radix-2 passes of rounds 1-2); the linear correlation is the same for any padded length
// L >= 2n, so L is the next power of two and lags are re-indexed to the reference's 2n layout.
// ------------------------------------------------------------------------------------------
__global__ void delay_pad_kernel(const cf* __restrict__ x, cf* __restrict__ out, int64_t n, int64_t len) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < len; idx += stride)
        out[idx] = idx < n ? x[idx] : fxc::mk(0.f, 0.f);
}

// one radix-R Stockham stage (R = 16, 8, 4, 2): thread j of len / R takes in[j + q len / R], q = 0 .. R-1, multiplies by
// exp(sign 2 pi i q k / (p R)) with k = j mod p (p = the product of the radices of the stages before), transforms, and
// writes Y_q to out[(j - k) R + k + q p]: natural order in, natural order out after the last stage.  The butterflies have
// the kernel exp(+2 pi i ..) (fx_math.h, fx_tiled.h); the forward transform (sign = -1) runs them on conjugated data.
template <int R>
__global__ void stockham_stage_kernel(const cf* __restrict__ in, cf* __restrict__ out, int64_t len, int64_t p, double sign) {
    const int64_t sub = len / R;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < sub; j += stride) {
        const int64_t k = j & (p - 1);
        cf v[16];
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const cf u = in[j + q * sub];
            double sn = 0.0, cs = 1.0;
            if (q > 0) sincospi(2.0 * (double)q * (double)k / ((double)p * (double)R), &sn, &cs);
            // u exp(sign i angle) in float64, conjugated for the forward transform
            const double re = (double)u.x * cs - sign * (double)u.y * sn;
            const double im = sign * (double)u.x * sn + (double)u.y * cs;
            v[q] = fxc::mk((float)re, (float)(sign < 0.0 ? -im : im));
        }
        if (R == 16) {
            fxc::dft16(v);
        } else if (R == 8) {
            fxc::tiled::dft8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
        } else if (R == 4) {
            fxc::dft4(v[0], v[1], v[2], v[3]);
        } else {
            fxc::tiled::dft2(v[0], v[1]);
        }
        const int64_t base = (j - k) * R + k;
#pragma unroll
        for (int q = 0; q < R; ++q) out[base + q * p] = fxc::mk(v[q].x, sign < 0.0 ? -v[q].y : v[q].y);
    }
}

__global__ void mul_conj_kernel(cf* __restrict__ a, const cf* __restrict__ b, int64_t len) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < len; idx += stride)
        a[idx] = fxc::cmulc(a[idx], b[idx]);
}

// arg-max of |r| over the reference's index i = 0..2n-1 (lag i - n, stored at (i - n) mod len); first maximum
// wins like numpy.argmax.  best[0] = packed (|r|^2 as ordered bits << 32 | ~i) maximised with atomicMax.
__global__ void delay_argmax_kernel(const cf* __restrict__ r, unsigned long long* __restrict__ best, int64_t n,
                                    int64_t len) {
    unsigned long long loc = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * n; i += stride) {
        const int64_t pos = (i - n + len) & (len - 1);
        const cf v = r[pos];
        const float m = v.x * v.x + v.y * v.y;
        const unsigned long long key = ((unsigned long long)__float_as_uint(m) << 32) | (0xFFFFFFFFull - (unsigned)i);
        loc = key > loc ? key : loc;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_down(loc, off);
        loc = o > loc ? o : loc;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(best, loc);
}

// out3 = r at reference indices imax-1 (python wrap for -1), imax, imax+1
__global__ void delay_fetch_kernel(const cf* __restrict__ r, const unsigned long long* __restrict__ best,
                                   cf* __restrict__ out3, int64_t n, int64_t len) {
    const int64_t imax = (int64_t)(0xFFFFFFFFull - (best[0] & 0xFFFFFFFFull));
    const int d = threadIdx.x;
    if (d < 3) {
        int64_t i = imax - 1 + d;
        if (i < 0) i += 2 * n;
        if (i >= 2 * n) i = imax;   // flagged on the host (the reference raises IndexError there)
        out3[d] = r[(i - n + len) & (len - 1)];
    }
}

}  // namespace
