// Part of libfxcorr's single translation unit: included by fxcorr.hip (not a stand-alone header).
#pragma once

namespace {

// ------------------------------------------------------------------------------------------
// PFB pre-filter for ntaps > 4 on the tiled channel counts (the reference's own test shape is taps = 32,
// branches 2048 / 4096: /root/reference/tests/test_effex.py:62-66).  A register ring of four frames does not
// stretch to 32 taps, and re-reading the history per spectrum costs (ntaps + 1) / 2 times the stream.  The FIR half of
// channelize_poly (effex.py:553) works branch by branch, so it is applied in place of the samples first:
//     y[i N + n] = sum_{t < T, i - t >= 0} h[t N + (N - 1 - n)] x[(i - t) N + n]
// after which the tiled kernels run with a single unit tap on y (their branch m reads position N - 1 - m: exactly the
// filtered branch).  A thread owns one sample position of one stream and walks its frames in blocks of TP: the block
// in flight and the one before it sit in registers (2 TP complex), every sample is loaded once and every tap is
// applied from registers.  HBM: stream in + stream out, then stream in again for the FFT/X kernel: 3 x algorithmic,
// whatever ntaps is.
// ------------------------------------------------------------------------------------------
typedef unsigned v4u32 __attribute__((ext_vector_type(4)));

// outputs i0 .. i0 + TP - 1 from the block in flight (xn) and the one before it (xo), stored as they are formed.
// W = adjacent sample positions per thread (1: 8-byte, 2: 16-byte accesses)
// cache policy of the pre-filter pass's sample loads and of its stores of the filtered frames (each read once by the pass behind it):
// 0 default, 2 nontemporal (2048 channels / 32 taps: 2.60 - 2.62 -> 2.53 - 2.55 ms)
#ifndef FXC_PRE_LD_AUX
#define FXC_PRE_LD_AUX 2
#endif
#ifndef FXC_PRE_ST_AUX
#define FXC_PRE_ST_AUX 2
#endif
template <int TP, int W>
__device__ __forceinline__ void prefilter_fir_store(const cf (&xo)[TP][W], const cf (&xn)[TP][W], const float (&hc)[TP][W],
                                                    __amdgpu_buffer_rsrc_t rs, unsigned voff, int64_t i0, int64_t i_end,
                                                    unsigned frame_bytes) {
    const bool full = i0 + TP <= i_end;    // wave-uniform
#pragma unroll
    for (int k = 0; k < TP; ++k) {
        float ar[W], ai[W];
#pragma unroll
        for (int w = 0; w < W; ++w) {
            ar[w] = ai[w] = 0.f;
#pragma unroll
            for (int t = 0; t < TP; ++t) {
                const cf v = (k - t >= 0) ? xn[(k - t) >= 0 ? k - t : 0][w] : xo[(TP + k - t) < TP ? TP + k - t : 0][w];
                ar[w] = fmaf(hc[t][w], v.x, ar[w]);
                ai[w] = fmaf(hc[t][w], v.y, ai[w]);
            }
        }
        if (full || i0 + k < i_end) {
            const unsigned soff = (unsigned)(i0 + k) * frame_bytes;
            if constexpr (W == 2) {
                // The frame offset rides in the VGPR offset here, not in the scalar one.  With the scalar offset the compiled
                // kernel had `buffer_store_dwordx4 v[0:3], ..., s0 offen` directly followed by `v_mov_b64 v[0:1], ...` (LLVM
                // pads a VALU write of a wide store's data registers only for stores without an soffset register) and stored
                // wrong frames 7 / 13 in the streams of workgroups >= 256, deterministically; in this form the compiler emits
                // the `s_nop 1` and every shape is right.  The instruction pair alone does not fail
                // (tools/ubench/store_hazard.hip, profiles/r03/gfx950_store_hazard.md), so what exactly went wrong in that
                // schedule is not known; the rule stays (tests/test_isa_hazards.py).
                v4u32 d = {__float_as_uint(ar[0]), __float_as_uint(ai[0]), __float_as_uint(ar[1]), __float_as_uint(ai[1])};
                __builtin_amdgcn_raw_buffer_store_b128(d, rs, voff + soff, 0, FXC_PRE_ST_AUX);
            } else {
                v2u32 d = {__float_as_uint(ar[0]), __float_as_uint(ai[0])};
                __builtin_amdgcn_raw_buffer_store_b64(d, rs, voff, soff, FXC_PRE_ST_AUX);
            }
        }
    }
}

// frames i0 .. i0 + TP - 1 of this thread's sample positions: buffer loads, one VGPR byte offset, scalar frame offsets.
// Frames past the stream's last one are clamped to it (a later frame never feeds an earlier output, and outputs past
// the end are not stored); frames before its first one read as zeros.
template <int TP, int W>
__device__ __forceinline__ void prefilter_load(cf (&xr)[TP][W], __amdgpu_buffer_rsrc_t rs, unsigned voff, int64_t i0, int64_t n_pts,
                                               unsigned frame_bytes) {
#pragma unroll
    for (int k = 0; k < TP; ++k) {
        const int64_t i = i0 + k;
        const int64_t ic = i < 0 ? 0 : (i < n_pts ? i : n_pts - 1);
        const bool zero = i < 0;   // wave-uniform
        if constexpr (W == 2) {
            const v4u32 d = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (unsigned)ic * frame_bytes, FXC_PRE_LD_AUX);
            xr[k][0] = fxc::mk(zero ? 0.f : __uint_as_float(d[0]), zero ? 0.f : __uint_as_float(d[1]));
            xr[k][1] = fxc::mk(zero ? 0.f : __uint_as_float(d[2]), zero ? 0.f : __uint_as_float(d[3]));
        } else {
            const v2u32 d = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, (unsigned)ic * frame_bytes, FXC_PRE_LD_AUX);
            xr[k][0] = fxc::mk(zero ? 0.f : __uint_as_float(d[0]), zero ? 0.f : __uint_as_float(d[1]));
        }
    }
}

// hcoef[t][n] = h[t N + (N - 1 - n)] for t < ntaps, zero rows up to TP; grid (N / (256 W), streams, frame splits)
// (asking for 3 waves per SIMD at TP = 32 makes the compiler spill and the pass 3 % slower: measured)
// PACK (channel counts below 256 W: 16 .. 256, in front of the wave-local kernels of k_small.h): a workgroup takes
// streams_per_block = 256 W / nchan consecutive streams side by side, grid (1, ceil(streams / streams_per_block), frame
// splits); the buffer descriptors span those streams and a lane's stream rides in its byte offset, so the descriptors stay
// wave-uniform (a per-lane stream index in the descriptor costs a readfirstlane loop around every access: +27 % on the
// 32-tap pass when the unpacked kernel was written that way).  The launcher packs only while the span fits 2^31 bytes.
template <int TP, int W, bool PACK = false>
__global__ __launch_bounds__(256) void pfb_prefilter_kernel(const cf* __restrict__ x, cf* __restrict__ y,
                                                           const float* __restrict__ hcoef, int64_t num_samp, int nchan,
                                                           int64_t n_pts, int64_t per_split, int streams_per_block,
                                                           int64_t n_streams) {
    const int lin = (blockIdx.x * 256 + threadIdx.x) * W;    // first of this thread's W adjacent positions
    const int n = PACK ? lin % nchan : lin;
    const int sl = PACK ? lin / nchan : 0;                   // this lane's stream among the workgroup's
    const int64_t s = PACK ? (int64_t)blockIdx.y * streams_per_block : (int64_t)blockIdx.y;      // (first) stream: uniform
    const int64_t span = PACK ? (n_streams - s < streams_per_block ? n_streams - s : streams_per_block) : 1;
    if (PACK && sl >= span) return;
    const int64_t i_begin = (int64_t)blockIdx.z * per_split;
    const int64_t i_end = (i_begin + per_split < n_pts) ? i_begin + per_split : n_pts;
    if (i_begin >= i_end) return;
    float hc[TP][W];
#pragma unroll
    for (int t = 0; t < TP; ++t)
#pragma unroll
        for (int w = 0; w < W; ++w) hc[t][w] = hcoef[(int64_t)t * nchan + n + w];
    const unsigned stream_bytes = (unsigned)(num_samp * (int64_t)sizeof(cf));       // num_samp <= 2^27
    const unsigned span_bytes = stream_bytes * (unsigned)span;                        // PACK: <= 2^31 (the launcher)
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<cf*>(x + s * num_samp), 0, (int)span_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(y + s * num_samp, 0, (int)span_bytes, 0x00020000);
    const unsigned voff = (unsigned)sl * stream_bytes + (unsigned)n * (unsigned)sizeof(cf);
    const unsigned frame_bytes = (unsigned)nchan * (unsigned)sizeof(cf);
    cf xa[TP][W], xb[TP][W];
    prefilter_load<TP, W>(xa, rx, voff, i_begin - TP, n_pts, frame_bytes);     // history (zeros before the stream's start)
    for (int64_t i0 = i_begin; i0 < i_end; i0 += 2 * TP) {                     // two blocks per trip: the pair swaps roles, no copies
        prefilter_load<TP, W>(xb, rx, voff, i0, n_pts, frame_bytes);
        prefilter_fir_store<TP, W>(xa, xb, hc, ry, voff, i0, i_end, frame_bytes);
        if (i0 + TP >= i_end) break;
        prefilter_load<TP, W>(xa, rx, voff, i0 + TP, n_pts, frame_bytes);
        prefilter_fir_store<TP, W>(xb, xa, hc, ry, voff, i0 + TP, i_end, frame_bytes);
    }
}

// ------------------------------------------------------------------------------------------
// nchan = 8192 as two 4096-channel problems.  A frame ring does not fit 8192 channels (128 VGPRs per thread at 1024
// threads, or 128 KiB of window + 136 KiB of exchange rows in LDS at 512), and the plain tiled kernel re-reads its
// history (0.145 of the HBM roofline).  Decimation in frequency splits the transform of the FIR output v[m]:
//     spec[2k']     = sum_{m < 4096} (v[m] + v[m + 4096])            w4096^(m k')
//     spec[2k' + 1] = sum_{m < 4096} (v[m] - v[m + 4096]) w8192^m    w4096^(m k')
// so the pre-filter pass (above) is extended: a thread owns the sample positions n' and n' + 4096 of a stream, forms
// both FIR outputs y_lo, y_hi per frame from registers, and writes a = y_hi + y_lo and b = (y_hi - y_lo) w8192^(4095 - n')
// at position n' of two half-size streams.  The headline kernel then runs on those as 2 n_chunks chunk pairs with a
// single unit tap -- pair 2c gives the even bins of chunk c, pair 2c + 1 the odd ones (raw layout 3).
// y = [chunk][even | odd][antenna][n_pts * 4096].  HBM: stream in + out, then in again: 3 x algorithmic.
// ------------------------------------------------------------------------------------------
// W = adjacent positions n', n' + 1 per thread (2: 16-byte accesses; the stores keep their frame offset in the VGPR,
// see prefilter_fir_store)
template <int TP, int W>
__global__ __launch_bounds__(256) void pfb_split8192_kernel(const cf* __restrict__ x, cf* __restrict__ y,
                                                           const float* __restrict__ hcoef, const cf* __restrict__ tw,
                                                           int64_t num_samp, int64_t n_pts, int64_t per_split) {
    constexpr int kHalf = 4096, kFull = 8192;
    const int n = (blockIdx.x * 256 + threadIdx.x) * W;    // first position inside the half frame
    const int64_t s = blockIdx.y;                           // stream = chunk * 2 + antenna
    const int64_t i_begin = (int64_t)blockIdx.z * per_split;
    const int64_t i_end = (i_begin + per_split < n_pts) ? i_begin + per_split : n_pts;
    if (i_begin >= i_end) return;
    float hc[TP][2][W];
    cf w[W];
#pragma unroll
    for (int q = 0; q < W; ++q) {
#pragma unroll
        for (int t = 0; t < TP; ++t) {
            hc[t][0][q] = hcoef[(int64_t)t * kFull + n + q];
            hc[t][1][q] = hcoef[(int64_t)t * kFull + n + q + kHalf];
        }
        w[q] = tw[n + q];
    }
    const int64_t half_samp = n_pts * kHalf;
    const int64_t c = s >> 1, a = s & 1;
    const unsigned in_bytes = (unsigned)(num_samp * (int64_t)sizeof(cf)), out_bytes = (unsigned)(half_samp * (int64_t)sizeof(cf));
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<cf*>(x + s * num_samp), 0, (int)in_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(y + ((c * 2 + 0) * 2 + a) * half_samp, 0, (int)out_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(y + ((c * 2 + 1) * 2 + a) * half_samp, 0, (int)out_bytes, 0x00020000);
    const unsigned voff = (unsigned)n * (unsigned)sizeof(cf);
    const unsigned in_frame = kFull * (unsigned)sizeof(cf), out_frame = kHalf * (unsigned)sizeof(cf), hi = kHalf * (unsigned)sizeof(cf);
    cf xa[TP][2][W], xb[TP][2][W];
    auto load = [&](cf (&xr)[TP][2][W], int64_t i0) {
#pragma unroll
        for (int k = 0; k < TP; ++k) {
            const int64_t i = i0 + k;
            const int64_t ic = i < 0 ? 0 : (i < n_pts ? i : n_pts - 1);      // see prefilter_load
            const bool zero = i < 0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const unsigned soff = (unsigned)ic * in_frame + (h ? hi : 0u);
                if constexpr (W == 2) {
                    const v4u32 d = __builtin_amdgcn_raw_buffer_load_b128(rx, voff, soff, 0);
                    xr[k][h][0] = fxc::mk(zero ? 0.f : __uint_as_float(d[0]), zero ? 0.f : __uint_as_float(d[1]));
                    xr[k][h][1] = fxc::mk(zero ? 0.f : __uint_as_float(d[2]), zero ? 0.f : __uint_as_float(d[3]));
                } else {
                    const v2u32 d = __builtin_amdgcn_raw_buffer_load_b64(rx, voff, soff, 0);
                    xr[k][h][0] = fxc::mk(zero ? 0.f : __uint_as_float(d[0]), zero ? 0.f : __uint_as_float(d[1]));
                }
            }
        }
    };
    auto fir_store = [&](const cf (&xo)[TP][2][W], const cf (&xn)[TP][2][W], int64_t i0) {
        const bool full = i0 + TP <= i_end;
#pragma unroll
        for (int k = 0; k < TP; ++k) {
            cf ea[W], eb[W];
#pragma unroll
            for (int q = 0; q < W; ++q) {
                cf yv[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float ar = 0.f, ai = 0.f;
#pragma unroll
                    for (int t = 0; t < TP; ++t) {
                        const cf v = (k - t >= 0) ? xn[(k - t) >= 0 ? k - t : 0][h][q] : xo[(TP + k - t) < TP ? TP + k - t : 0][h][q];
                        ar = fmaf(hc[t][h][q], v.x, ar);
                        ai = fmaf(hc[t][h][q], v.y, ai);
                    }
                    yv[h] = fxc::mk(ar, ai);
                }
                ea[q] = fxc::cadd(yv[1], yv[0]);
                eb[q] = fxc::cmul(fxc::csub(yv[1], yv[0]), w[q]);
            }
            if (full || i0 + k < i_end) {
                const unsigned soff = (unsigned)(i0 + k) * out_frame;
                if constexpr (W == 2) {
                    v4u32 da = {__float_as_uint(ea[0].x), __float_as_uint(ea[0].y), __float_as_uint(ea[1].x), __float_as_uint(ea[1].y)};
                    v4u32 db = {__float_as_uint(eb[0].x), __float_as_uint(eb[0].y), __float_as_uint(eb[1].x), __float_as_uint(eb[1].y)};
                    __builtin_amdgcn_raw_buffer_store_b128(da, ra, voff + soff, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(db, rb, voff + soff, 0, 0);
                } else {
                    v2u32 da = {__float_as_uint(ea[0].x), __float_as_uint(ea[0].y)}, db = {__float_as_uint(eb[0].x), __float_as_uint(eb[0].y)};
                    __builtin_amdgcn_raw_buffer_store_b64(da, ra, voff, soff, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(db, rb, voff, soff, 0);
                }
            }
        }
    };
    load(xa, i_begin - TP);
    for (int64_t i0 = i_begin; i0 < i_end; i0 += 2 * TP) {
        load(xb, i0);
        fir_store(xa, xb, i0);
        if (i0 + TP >= i_end) break;
        load(xa, i0 + TP);
        fir_store(xb, xa, i0 + TP);
    }
}

// acc[k] += the leading-part rows of the split launch that belong to bin k's half (even bins: fused chunks 2c, odd:
// 2c + 1); the chunk rows themselves go through fold_partial_kernel / fold_finish_kernel in layout 3
__global__ __launch_bounds__(256) void split_lead_acc_kernel(const cf* __restrict__ raw, cd* __restrict__ acc, LeadRows lr) {
    // the fused chunk workgroup b's range starts in (its leading part, if any, belongs to that chunk): parity per row,
    // worked out once per block instead of one 64-bit division per row and thread
    __shared__ unsigned char odd[1024];
    for (int b = threadIdx.x; b < lr.grid; b += blockDim.x)
        odd[b] = (unsigned char)((lr.first_chunk + fxc::range_begin(b, (int)lr.n_frames, lr.grid) / lr.n_pts) & 1);
    __syncthreads();
    const int k = blockIdx.x * blockDim.x + threadIdx.x;    // 0 .. 8191
    const int64_t slot = fxc::fused::slot_of_bin(k >> 1);
    const unsigned char mine = (unsigned char)(k & 1);
    double ar = 0.0, ai = 0.0;
#pragma unroll 8
    for (int b = 0; b < lr.grid; ++b) {
        const cf r = raw[lr.offset + (int64_t)b * fxc::fused::kN + slot];
        if (odd[b] == mine) {
            ar += r.x;
            ai += r.y;
        }
    }
    cd v = acc[k];
    v.x += ar;
    v.y += ai;
    acc[k] = v;
}

}  // namespace
