// Part of libfxcorr's single translation unit: included by fxcorr.hip (not a stand-alone header).
#pragma once

namespace {

// ------------------------------------------------------------------------------------------
// continuum streaming limit: nchan == 1, 2 antennas (BASELINE config 3(i))
// The PFB degenerates to a T-tap FIR y_a[n] = sum_t h[t] x_a[n - t] (zero history per chunk), the FFT is
// the identity and X is sum_n y_0[n] conj(y_1[n]).  One workgroup takes kStreamBlock consecutive
// samples of both streams (+ T-1 of halo) through LDS; raw[block][chunk] = its partial sum (float32),
// summed over blocks in float64 by the finishing kernels.  16 B of HBM per sample, ~40 flop.
// ------------------------------------------------------------------------------------------
constexpr int kStreamBlock = 2048;
struct StreamTaps {
    float h[kMaxTaps];
};

__global__ __launch_bounds__(256) void stream1_kernel(const cf* __restrict__ x, cf* __restrict__ raw, int64_t num_samp,
                                                     int ntaps, StreamTaps taps, int64_t n_chunks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cf* buf = reinterpret_cast<cf*>(smem);            // [2][kStreamBlock + ntaps - 1]
    __shared__ cf red[256];
    const int span = kStreamBlock + ntaps - 1;
    const int64_t blk = blockIdx.x, c = blockIdx.y;
    const int64_t n0 = blk * kStreamBlock;
    for (int a = 0; a < 2; ++a) {
        const cf* xs = x + (c * 2 + a) * num_samp;
        for (int idx = threadIdx.x; idx < span; idx += blockDim.x) {
            const int64_t n = n0 - (ntaps - 1) + idx;
            buf[a * span + idx] = (n >= 0 && n < num_samp) ? xs[n] : fxc::mk(0.f, 0.f);
        }
    }
    __syncthreads();
    float ar = 0.f, ai = 0.f;
    for (int q = 0; q < kStreamBlock / 256; ++q) {
        const int m = q * 256 + threadIdx.x;          // output n0 + m sits at buf[m + ntaps - 1]
        if (n0 + m < num_samp) {
            float y0r = 0.f, y0i = 0.f, y1r = 0.f, y1i = 0.f;
            for (int t = 0; t < ntaps; ++t) {
                const float w = taps.h[t];
                const cf u = buf[m + ntaps - 1 - t], z = buf[span + m + ntaps - 1 - t];
                y0r = fmaf(w, u.x, y0r);
                y0i = fmaf(w, u.y, y0i);
                y1r = fmaf(w, z.x, y1r);
                y1i = fmaf(w, z.y, y1i);
            }
            ar += y0r * y1r + y0i * y1i;
            ai += y0i * y1r - y0r * y1i;
        }
    }
    red[threadIdx.x] = fxc::mk(ar, ai);
    __syncthreads();
    for (int sft = 128; sft > 0; sft >>= 1) {
        if ((int)threadIdx.x < sft) red[threadIdx.x] = fxc::cadd(red[threadIdx.x], red[threadIdx.x + sft]);
        __syncthreads();
    }
    if (threadIdx.x == 0) raw[blk * n_chunks + c] = red[0];
}

// ntaps <= 4, even num_samp: no LDS.  A thread takes sample pairs (2m, 2m+1) of both streams with three
// aligned 16-byte loads each ([2m-4, 2m-3], [2m-2, 2m-1], [2m, 2m+1]; the two halo loads hit L1 / the
// neighbouring lanes' lines, HBM sees every sample once) and walks its workgroup's contiguous slice of the
// chunk with a stride of 256 pairs.  raw[block][chunk] = partial sum.
constexpr int kStream4Blocks = 16;   // workgroups per chunk
typedef float v4f32 __attribute__((ext_vector_type(4)));

// ntaps <= 4, even num_samp: a lane takes one sample pair of both streams with an aligned 16-byte load; the two earlier pairs the FIR
// needs come from the neighbouring lanes (v_mov_b32_dpp wave_shr:1).  A wave walks a CONTIGUOUS run of its workgroup's pairs, 64 new
// pairs a trip: what lanes 0 and 1 need from before the trip are lanes 62 and 63 of the wave's own previous trip (v_readlane, shifted in
// through the DPP move's `old` operand), so every pair is loaded exactly once -- in round 5 a wave covered 62 new pairs plus two halo
// lanes, and with nontemporal loads the halo's lines came from HBM a second time (FETCH_SIZE read 1.089 x the algorithmic bytes:
// profiles/r05/summary_stream1.json; now profiles/r06/summary_stream1.json).
__device__ __forceinline__ float lane_shr1(float v, float first) {      // lane l <- v of lane l - 1; lane 0 <- first
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(first), __float_as_int(v), 0x138, 0xF, 0xF, false));
}
__device__ __forceinline__ v4f32 wave_shr1(v4f32 v, v4f32 first) {
    const float x = lane_shr1(v.x, first.x), y = lane_shr1(v.y, first.y), z = lane_shr1(v.z, first.z), w = lane_shr1(v.w, first.w);
    v4f32 r = {x, y, z, w};
    return r;
}
__device__ __forceinline__ v4f32 wave_lane(v4f32 v, int lane) {            // lane `lane`'s value in every lane (a scalar register)
    v4f32 r = {__int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.x), lane)), __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.y), lane)),
               __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.z), lane)), __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.w), lane))};
    return r;
}

__global__ __launch_bounds__(256) void stream1_t4_kernel(const cf* __restrict__ x, cf* __restrict__ raw, int64_t num_samp,
                                                            float h0, float h1, float h2, float h3, int64_t n_chunks) {
    __shared__ cf red[256];
    const int64_t c = blockIdx.y;
    const int64_t pairs = num_samp / 2;
    const int64_t per_blk = (pairs + gridDim.x - 1) / gridDim.x;
    const int64_t p0 = (int64_t)blockIdx.x * per_blk;
    const int64_t p1 = (p0 + per_blk < pairs) ? p0 + per_blk : pairs;
    const v4f32* s0 = reinterpret_cast<const v4f32*>(x + (c * 2 + 0) * num_samp);
    const v4f32* s1 = reinterpret_cast<const v4f32*>(x + (c * 2 + 1) * num_samp);
    const v4f32 zero = {0.f, 0.f, 0.f, 0.f};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // the wave's run of the workgroup's pairs (wave-uniform bounds)
    const int64_t per_wave = (((p1 > p0 ? p1 - p0 : 0) + 3) / 4 + 63) / 64 * 64;
    const int64_t q0 = p0 + wave * per_wave < p1 ? p0 + wave * per_wave : p1;
    const int64_t q1 = q0 + per_wave < p1 ? q0 + per_wave : p1;
    // the two pairs in front of the run (zeros in front of the chunk): what lanes 62 and 63 of a trip before the first would have held
    v4f32 a62 = q0 >= 2 && q0 < q1 ? s0[q0 - 2] : zero, a63 = q0 >= 1 && q0 < q1 ? s0[q0 - 1] : zero;
    v4f32 b62 = q0 >= 2 && q0 < q1 ? s1[q0 - 2] : zero, b63 = q0 >= 1 && q0 < q1 ? s1[q0 - 1] : zero;
    float ar = 0.f, ai = 0.f;
    for (int64_t base = q0; base < q1; base += 64) {      // wave-uniform trip count
        const int64_t m = base + lane;
        const bool in_range = m < q1;
        const v4f32 a2 = in_range ? __builtin_nontemporal_load(s0 + m) : zero, b2 = in_range ? __builtin_nontemporal_load(s1 + m) : zero;
        const v4f32 a1 = wave_shr1(a2, a63), b1 = wave_shr1(b2, b63);
        const v4f32 a0 = wave_shr1(a1, a62), b0 = wave_shr1(b1, b62);
        a62 = wave_lane(a2, 62);
        a63 = wave_lane(a2, 63);
        b62 = wave_lane(b2, 62);
        b63 = wave_lane(b2, 63);
        const float y0er = h0 * a2[0] + h1 * a1[2] + h2 * a1[0] + h3 * a0[2];
        const float y0ei = h0 * a2[1] + h1 * a1[3] + h2 * a1[1] + h3 * a0[3];
        const float y0or = h0 * a2[2] + h1 * a2[0] + h2 * a1[2] + h3 * a1[0];
        const float y0oi = h0 * a2[3] + h1 * a2[1] + h2 * a1[3] + h3 * a1[1];
        const float y1er = h0 * b2[0] + h1 * b1[2] + h2 * b1[0] + h3 * b0[2];
        const float y1ei = h0 * b2[1] + h1 * b1[3] + h2 * b1[1] + h3 * b0[3];
        const float y1or = h0 * b2[2] + h1 * b2[0] + h2 * b1[2] + h3 * b1[0];
        const float y1oi = h0 * b2[3] + h1 * b2[1] + h2 * b1[3] + h3 * b1[1];
        if (in_range) {
            ar += y0er * y1er + y0ei * y1ei + y0or * y1or + y0oi * y1oi;
            ai += y0ei * y1er - y0er * y1ei + y0oi * y1or - y0or * y1oi;
        }
    }
    red[threadIdx.x] = fxc::mk(ar, ai);
    __syncthreads();
    for (int sft = 128; sft > 0; sft >>= 1) {
        if ((int)threadIdx.x < sft) red[threadIdx.x] = fxc::cadd(red[threadIdx.x], red[threadIdx.x + sft]);
        __syncthreads();
    }
    if (threadIdx.x == 0) raw[(int64_t)blockIdx.x * n_chunks + c] = red[0];
}

// acc[0] += sum of all partials (float64, fixed order)
__global__ __launch_bounds__(256) void stream1_acc_kernel(const cf* __restrict__ raw, cd* __restrict__ acc, int64_t n) {
    __shared__ double red[256];
    double ar = 0.0, ai = 0.0;
    for (int64_t idx = threadIdx.x; idx < n; idx += blockDim.x) {
        ar += raw[idx].x;
        ai += raw[idx].y;
    }
    ar = block_sum(ar, red);
    ai = block_sum(ai, red);
    if (threadIdx.x == 0) {
        acc[0].x += ar;
        acc[0].y += ai;
    }
}

}  // namespace
