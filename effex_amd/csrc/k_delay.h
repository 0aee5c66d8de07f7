// Part of libfxcorr's single translation unit: included by fxcorr.hip (not a stand-alone header).
#pragma once

namespace {

// ------------------------------------------------------------------------------------------
// delay calibration — effex/effex.py:583-627: zero-padded FFT cross-correlation,
// arg-max of |xcorr|, 3-point log-Gaussian peak.  Runs once per calibration, so the FFT is a plain
// global-memory radix-2 Stockham (log2 L passes); the linear correlation is the same for any padded
// length L >= 2n, so L is the next power of two and lags are re-indexed to the reference's 2n layout.
// ------------------------------------------------------------------------------------------
__global__ void delay_pad_kernel(const cf* __restrict__ x, cf* __restrict__ out, int64_t n, int64_t len) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < len; idx += stride)
        out[idx] = idx < n ? x[idx] : fxc::mk(0.f, 0.f);
}

// one radix-2 Stockham stage: natural order in, natural order out after log2(len) stages
__global__ void stockham_stage_kernel(const cf* __restrict__ in, cf* __restrict__ out, int64_t half_len, int64_t p,
                                      double sign) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < half_len; j += stride) {
        const int64_t k = j & (p - 1);
        double sn, cs;
        sincospi(sign * (double)k / (double)p, &sn, &cs);
        const cf u0 = in[j], u1 = in[j + half_len];
        const float tr = (float)((double)u1.x * cs - (double)u1.y * sn);
        const float ti = (float)((double)u1.x * sn + (double)u1.y * cs);
        const int64_t jj = ((j - k) << 1) + k;
        out[jj] = fxc::mk(u0.x + tr, u0.y + ti);
        out[jj + p] = fxc::mk(u0.x - tr, u0.y - ti);
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
