// Part of libfxcorr's single translation unit: included by fxcorr.hip (not a stand-alone header).
#pragma once

namespace {

// ------------------------------------------------------------------------------------------
// input conditioning: RTL-SDR uint8 IQ -> complex64 and per-chunk DC removal
//   reference: pyrtlsdr packed-bytes-to-samples (byte - 127.5) / 127.5 [third party], and
//   effex/effex.py:394-395  x = (x.real - x.real.mean()) + 1j * (x.imag - x.imag.mean())  per chunk, per antenna
// ------------------------------------------------------------------------------------------
// sums[stream] = {sum re, sum im} in float64; one workgroup per (stream, slice), fixed-order two-level sum
__global__ __launch_bounds__(256) void dc_sum_c64_kernel(const cf* __restrict__ x, double* __restrict__ part,
                                                        int64_t num_samp, int n_slices) {
    __shared__ double red[256];
    const int64_t s = blockIdx.y;
    const int slice = blockIdx.x;
    const int64_t per = (num_samp + n_slices - 1) / n_slices;
    const int64_t lo = slice * per, hi = (lo + per < num_samp) ? lo + per : num_samp;
    double ar = 0.0, ai = 0.0;
    const cf* __restrict__ xs = x + s * num_samp;
    int64_t n = lo + threadIdx.x;
    for (; n + 3 * (int64_t)blockDim.x < hi; n += 4 * (int64_t)blockDim.x) {       // four loads in flight, added in index order
        cf v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = xs[n + q * (int64_t)blockDim.x];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            ar += v[q].x;
            ai += v[q].y;
        }
    }
    for (; n < hi; n += blockDim.x) {
        const cf v = xs[n];
        ar += v.x;
        ai += v.y;
    }
    ar = block_sum(ar, red);
    ai = block_sum(ai, red);
    if (threadIdx.x == 0) {
        part[(s * n_slices + slice) * 2] = ar;
        part[(s * n_slices + slice) * 2 + 1] = ai;
    }
}

// uint8 ingest (fused and through the conversion pass): exact byte sums of streams, 16-byte loads (8 samples per lane).  n_slices == 1: one workgroup per
// stream, part[s * 2] = sum of I bytes, part[s * 2 + 1] = sum of Q bytes.  n_slices > 1 (calls of a few streams -- the
// reference hands over one chunk pair at a time, effex.py:490-494 -- would leave a stream of 512 KiB to one workgroup):
// workgroup w sums slice w % n_slices of stream w / n_slices into part[(s n_slices + slice) * 2]; the sums are integers,
// so dc_offsets_u8_kernel's total over the slices is the same number whatever the split.
__global__ __launch_bounds__(256) void dc_sum_u8_stream_kernel(const unsigned char* __restrict__ x, double* __restrict__ part,
                                                              int64_t num_samp, int64_t n_streams, int n_slices) {
    __shared__ double red[256];
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    for (int64_t w = blockIdx.x; w < n_streams * n_slices; w += gridDim.x) {
        const int64_t s = w / n_slices;
        const int slice = (int)(w - s * n_slices);
        const unsigned char* base = x + s * num_samp * 2;
        // align to 16 bytes: head and tail bytes one sample at a time (slice 0)
        const int64_t head = (int64_t)(((16 - (reinterpret_cast<uintptr_t>(base) & 15)) & 15) / 2);
        const int64_t h = head < num_samp ? head : num_samp;
        const int64_t n_vec = (num_samp - h) / 8;
        const int64_t v_lo = n_vec * slice / n_slices, v_hi = n_vec * (slice + 1) / n_slices;
        const v4u* vp = reinterpret_cast<const v4u*>(base + h * 2);
        unsigned long long ar = 0, ai = 0;
        for (int64_t n = v_lo + threadIdx.x; n < v_hi; n += 256) {
            const v4u w4 = vp[n];
            unsigned si = 0, sq = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                si = __builtin_amdgcn_sad_u8(w4[k] & 0x00FF00FFu, 0u, si);
                sq = __builtin_amdgcn_sad_u8((w4[k] >> 8) & 0x00FF00FFu, 0u, sq);
            }
            ar += si;
            ai += sq;
        }
        if (slice == 0) {
            const unsigned short* sp = reinterpret_cast<const unsigned short*>(base);
            for (int64_t n = threadIdx.x; n < h; n += 256) {
                ar += sp[n] & 0xFF;
                ai += sp[n] >> 8;
            }
            for (int64_t n = h + n_vec * 8 + threadIdx.x; n < num_samp; n += 256) {
                ar += sp[n] & 0xFF;
                ai += sp[n] >> 8;
            }
        }
        const double sr = block_sum((double)ar, red);
        const double si2 = block_sum((double)ai, red);
        if (threadIdx.x == 0) {
            part[w * 2] = sr;
            part[w * 2 + 1] = si2;
        }
    }
}

// the same sums of a complex128 stream (the reference's own sample type, effex.py:109-110, 394-395)
__global__ __launch_bounds__(256) void dc_sum_c128_kernel(const cd* __restrict__ x, double* __restrict__ part,
                                                         int64_t num_samp, int n_slices) {
    __shared__ double red[256];
    const int64_t s = blockIdx.y;
    const int slice = blockIdx.x;
    const int64_t per = (num_samp + n_slices - 1) / n_slices;
    const int64_t lo = slice * per, hi = (lo + per < num_samp) ? lo + per : num_samp;
    double ar = 0.0, ai = 0.0;
    for (int64_t n = lo + threadIdx.x; n < hi; n += blockDim.x) {
        const cd v = x[s * num_samp + n];
        ar += v.x;
        ai += v.y;
    }
    ar = block_sum(ar, red);
    ai = block_sum(ai, red);
    if (threadIdx.x == 0) {
        part[(s * n_slices + slice) * 2] = ar;
        part[(s * n_slices + slice) * 2 + 1] = ai;
    }
}

// mean of stream s from its slice sums (at most 256 of them): one partial per thread, summed by the workgroup's fixed tree
__device__ __forceinline__ void dc_mean(const double* __restrict__ part, int64_t s, int n_slices, int64_t num_samp,
                                        double* mr, double* mi, double* red) {
    const int k = threadIdx.x;
    double ar = k < n_slices ? part[(s * n_slices + k) * 2] : 0.0;
    double ai = k < n_slices ? part[(s * n_slices + k) * 2 + 1] : 0.0;
    ar = block_sum(ar, red);
    ai = block_sum(ai, red);
    *mr = ar / (double)num_samp;
    *mi = ai / (double)num_samp;
}

// out = x - mean (complex64, in place or out of place): workgroup (slice, stream) of the same grid as the sum kernels,
// so the mean is formed once per workgroup and no thread divides an index by num_samp.  remove_dc == 0: a copy.
__global__ __launch_bounds__(256) void dc_apply_c64_kernel(const cf* __restrict__ x, cf* __restrict__ out,
                                                          const double* __restrict__ part, int64_t num_samp, int n_slices,
                                                          int part_slices, int remove_dc) {
    __shared__ double red[256];
    const int64_t s = blockIdx.y;
    double mr = 0.0, mi = 0.0;
    if (remove_dc) dc_mean(part, s, part_slices, num_samp, &mr, &mi, red);
    const int64_t per = (num_samp + n_slices - 1) / n_slices;
    const int64_t lo = (int64_t)blockIdx.x * per, hi = (lo + per < num_samp) ? lo + per : num_samp;
    const cf* __restrict__ xs = x + s * num_samp;
    cf* __restrict__ os = out + s * num_samp;
    int64_t n = lo + threadIdx.x;
    for (; n + 3 * (int64_t)blockDim.x < hi; n += 4 * (int64_t)blockDim.x) {       // four loads in flight
        cf v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = xs[n + q * (int64_t)blockDim.x];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            os[n + q * (int64_t)blockDim.x] = fxc::mk((float)((double)v[q].x - mr), (float)((double)v[q].y - mi));
    }
    for (; n < hi; n += blockDim.x) {
        const cf v = xs[n];
        os[n] = fxc::mk((float)((double)v.x - mr), (float)((double)v.y - mi));
    }
}

// complex128 -> complex64 with the mean removed in float64 first: (float)(x - mean), rounded once -- what the reference's
// host line effex.py:394-395 followed by a narrowing copy gives.  remove_dc == 0: the narrowing alone.
__global__ __launch_bounds__(256) void narrow_c128_kernel(const cd* __restrict__ x, cf* __restrict__ out,
                                                         const double* __restrict__ part, int64_t num_samp, int n_slices,
                                                         int part_slices, int remove_dc) {
    __shared__ double red[256];
    const int64_t s = blockIdx.y;
    double mr = 0.0, mi = 0.0;
    if (remove_dc) dc_mean(part, s, part_slices, num_samp, &mr, &mi, red);
    const int64_t per = (num_samp + n_slices - 1) / n_slices;
    const int64_t lo = (int64_t)blockIdx.x * per, hi = (lo + per < num_samp) ? lo + per : num_samp;
    for (int64_t n = lo + threadIdx.x; n < hi; n += blockDim.x) {
        const cd v = x[s * num_samp + n];
        out[s * num_samp + n] = fxc::mk((float)(v.x - mr), (float)(v.y - mi));
    }
}

// out = (byte - 127.5) / 127.5 [- mean]; remove_dc == 0 keeps the mean.  Workgroup (x, y): slice x of gridDim.x of the streams
// y, y + gridDim.y, ...; the stream's mean once per workgroup (the byte sums are exact integers: any slicing gives this number)
__global__ __launch_bounds__(256) void convert_u8_kernel(const unsigned char* __restrict__ x, cf* __restrict__ out,
                                                        const double* __restrict__ part, int64_t num_samp, int n_slices,
                                                        int64_t n_streams, int remove_dc) {
    __shared__ double mean[2];
    for (int64_t s = blockIdx.y; s < n_streams; s += gridDim.y) {
        if (threadIdx.x == 0) {
            double mr = 127.5, mi = 127.5;      // without DC removal only the format offset is subtracted
            if (remove_dc) {
                mr = mi = 0.0;
                for (int k = 0; k < n_slices; ++k) {
                    mr += part[(s * n_slices + k) * 2];
                    mi += part[(s * n_slices + k) * 2 + 1];
                }
                mr /= (double)num_samp;
                mi /= (double)num_samp;
            }
            mean[0] = mr;
            mean[1] = mi;
        }
        __syncthreads();
        const double mr = mean[0], mi = mean[1];
        const unsigned short* in = reinterpret_cast<const unsigned short*>(x) + s * num_samp;
        cf* o = out + s * num_samp;
        for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < num_samp; n += (int64_t)gridDim.x * 256) {
            const unsigned short v = in[n];
            // ((b - 127.5) - (mean_b - 127.5)) / 127.5 = (b - mean_b) / 127.5, formed in float64, rounded once
            o[n] = fxc::mk((float)(((double)(v & 0xFF) - mr) / 127.5), (float)(((double)(v >> 8) - mi) / 127.5));
        }
        __syncthreads();
    }
}

// conversion offsets of the fused uint8 ingest: dc[s] = -mean_byte / 127.5 per component (float64, rounded once), or
// -1 when the mean is kept (only the format offset 127.5 is removed)
__global__ void dc_offsets_u8_kernel(const double* __restrict__ part, cf* __restrict__ dc, int64_t n_streams, int n_slices,
                                     int64_t num_samp, int remove_dc) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_streams) return;
    double mr = 127.5, mi = 127.5;
    if (remove_dc) {
        mr = mi = 0.0;
        for (int k = 0; k < n_slices; ++k) {
            mr += part[(s * n_slices + k) * 2];
            mi += part[(s * n_slices + k) * 2 + 1];
        }
        mr /= (double)num_samp;
        mi /= (double)num_samp;
    }
    dc[s] = fxc::mk((float)(-mr / 127.5), (float)(-mi / 127.5));
}

}  // namespace
