// Part of libfxcorr's single translation unit: included by fxcorr.hip (not a stand-alone header).
#pragma once

namespace {

// ------------------------------------------------------------------------------------------
// generic path kernels (any ntaps <= 32, any n_ant, any nchan <= 16384)
// ------------------------------------------------------------------------------------------

// v[s][i][m] = sum_{t<T, i-t>=0} x[s][(i-t)N + N-1-m] * h[tN+m]      
__global__ void pfb_fir_kernel(const cf* __restrict__ x, const float* __restrict__ h, cf* __restrict__ v,
                               int64_t num_samp, int nchan, int ntaps, int64_t n_pts, int64_t total) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const int m = (int)(idx % nchan);
        const int64_t si = idx / nchan;
        const int64_t i = si % n_pts;
        const int64_t s = si / n_pts;
        const cf* xs = x + s * num_samp + (nchan - 1 - m);
        float ar = 0.f, ai = 0.f;
        const int tmax = (i + 1 < (int64_t)ntaps) ? (int)(i + 1) : ntaps;
        for (int t = 0; t < tmax; ++t) {
            const cf xv = xs[(i - t) * nchan];
            const float w = h[(int64_t)t * nchan + m];
            ar = fmaf(w, xv.x, ar);
            ai = fmaf(w, xv.y, ai);
        }
        v[idx] = fxc::mk(ar, ai);
    }
}

__device__ __forceinline__ unsigned bitrev(unsigned v, int bits) { return __brev(v) >> (32 - bits); }

// in-place spec[k] = sum_m v[m] exp(+2 pi i k m / N) for each row; N = 2^lg2n <= 16384.  One workgroup per row, or
// 512 / N rows per workgroup when N < 512 (N/2 threads per row, each row in its own LDS slice)
__global__ __launch_bounds__(256) void fft_pow2_kernel(cf* __restrict__ data, const cf* __restrict__ tw /* [N/2] */,
                                                      int nchan, int lg2n, int64_t n_rows) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int half_n = nchan >> 1;
    const int rpw = (nchan < 512 && nchan >= 2) ? 512 / nchan : 1;      // rows per workgroup
    const int tpr = rpw > 1 ? half_n : (int)blockDim.x;                 // threads per row
    const int sub = rpw > 1 ? (int)threadIdx.x / tpr : 0;
    const int lt = rpw > 1 ? (int)threadIdx.x % tpr : (int)threadIdx.x;
    cf* buf = reinterpret_cast<cf*>(smem) + (int64_t)sub * nchan;
    for (int64_t rb = (int64_t)blockIdx.x * rpw; rb < n_rows; rb += (int64_t)gridDim.x * rpw) {
        const bool active = rb + sub < n_rows;
        cf* d = data + (rb + sub) * nchan;
        if (active)
            for (int n = lt; n < nchan; n += tpr) buf[bitrev((unsigned)n, lg2n)] = d[n];
        __syncthreads();
        for (int s = 0; s < lg2n; ++s) {
            const int half = 1 << s;
            const int tstep = nchan >> (s + 1);
            if (active)
                for (int b = lt; b < half_n; b += tpr) {
                    const int pos = b & (half - 1);
                    const int i0 = ((b >> s) << (s + 1)) + pos;
                    const cf w = tw[pos * tstep];
                    const cf a = buf[i0];
                    const cf t = fxc::cmul(buf[i0 + half], w);
                    buf[i0] = fxc::cadd(a, t);
                    buf[i0 + half] = fxc::csub(a, t);
                }
            __syncthreads();
        }
        if (active)
            for (int n = lt; n < nchan; n += tpr) d[n] = buf[n];
        __syncthreads();
    }
}

#if FXC_DEV_KERNELS
// same transform for any nchan <= 16384 (O(N^2) per row); tw = [N] table, exact index arithmetic.  Not in the shipped library:
// the tests' and the soak's independent reference for channel counts that are not a power of two (libfxcorr_dev.so)
__global__ __launch_bounds__(256) void dft_any_kernel(cf* __restrict__ data, const cf* __restrict__ tw, int nchan,
                                                     int64_t n_rows) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cf* buf = reinterpret_cast<cf*>(smem);
    for (int64_t row = blockIdx.x; row < n_rows; row += gridDim.x) {
        cf* d = data + row * nchan;
        for (int n = threadIdx.x; n < nchan; n += blockDim.x) buf[n] = d[n];
        __syncthreads();
        for (int k = threadIdx.x; k < nchan; k += blockDim.x) {
            float ar = 0.f, ai = 0.f;
            int idx = 0;
            for (int m = 0; m < nchan; ++m) {
                const cf w = tw[idx];
                const cf a = buf[m];
                ar += a.x * w.x - a.y * w.y;
                ai += a.x * w.y + a.y * w.x;
                idx += k;
                if (idx >= nchan) idx -= nchan;
            }
            d[k] = fxc::mk(ar, ai);
        }
        __syncthreads();
    }
}
#endif

// FIR + FFT for any nchan whose two LDS rows fit (kMixedMaxN): the polyphase FIR runs on the way into LDS, the mixed-radix
// Stockham stages of fx_mixed.h ping-pong between the two rows, and the natural-order spectrum goes out once.
//   workgroup = rpw slots of tpr threads (rpw = 256 / tpr when nchan is small; up to 1024 threads share a slot when it is
//   large); a slot carries U rows through every step together: one index computation, one set of twiddles and one barrier
//   serve U rows.
//   XF = false (F only): the U rows are consecutive frames of one stream -- the FIR's taps of neighbouring frames are the
//   same loads (U + 3 per four taps, not 4 U); a workgroup owns a contiguous run of frame groups of one stream after
//   another (tap re-reads across groups hit its L2); the natural-order spectra go to `out` = spec[stream][frame][nchan].
//   XF = true (two antennas, F + X): the U = 2 rows are the two antennas' spectra of one frame; their product
//   s0 conj(s1) adds up in registers over the workgroup's run of frames of chunk c = blockIdx / n_splits, and the run's
//   sum goes to `out` = raw[split][chunk][nchan] -- xmul_kernel's layout, with no spectrum ever written.
// TWL: the twiddle table sits in LDS in front of the rows.  The kernel is bound by instruction issue (index arithmetic and
// LDS traffic of five-odd short stages) -- hence the packed-pair arithmetic and the shared indices -- and, with four taps, by
// the fabric traffic of the taps' re-reads a few per cent behind it.
constexpr int kMixedXPoints = 8;    // bins per thread of the XF accumulator: nchan <= 8 tpr (mixed_threads_per_row up to 4096)
//   BLU = true (a prime factor too large for a butterfly, F only, U = 1): Bluestein's chirp-z form of the same transform.
//   With c[n] = exp(+i pi n^2 / N):  X[k] = c[k] sum_n (v[n] c[n]) conj(c[k - n])  -- a convolution, done as a cyclic one
//   of length nfft >= 2 N - 1 (7-smooth, chosen by the host) in the row: u = v c zero-padded, Z = FFT(conj(FFT(u) D)),
//   X[k] = c[k] conj(Z[k]), D = FFT(conj(c) wrapped) / nfft from the host (float64).  Both FFTs are the stages of mp (tw for nfft).
// what the variants need beyond the common arguments
struct MixedExtras {
    int wave_local;         // slots of one wave synchronise without the workgroup barrier
    int ant;                // F only: spectra as out[chunk][frame][ant][nchan] for stream = chunk * ant + a (1: [stream][frame][nchan])
    int nfft;               // == nchan when BLU is false
    const cf* chirp;        // [nchan]
    const cf* d;            // [nfft]
    const cf* dc_u8;        // U8: conversion offsets [stream]
};
//   BIG = true (10240 < nchan <= 16384: one row is all the LDS holds; F only, U = 1, 1024 threads): the stages go back and
//   forth between the LDS row and the frame's own row of `out` (L2-resident; __syncthreads() orders global memory within the
//   workgroup), and the spectrum ends in `out` either way.
//   U8 = true (with XF): x is the receivers' interleaved unsigned bytes [chunk][2][num_samp][2] and blu.dc_u8 the conversion
//   offsets per stream (k_conditioning.h: sample = byte / 127.5 + offset, one FMA -- pyrtlsdr's conversion behind effex.py:652
//   and the DC removal of effex.py:394-395); a quarter of the bytes of complex64 through the fabric, and no conversion pass.
template <bool TWL, int U, bool XF, bool BLU = false, bool BIG = false, bool U8 = false>
__global__ __launch_bounds__(1024) void pfb_fft_mixed_kernel(const cf* __restrict__ x, const float* __restrict__ h,
                                                            cf* __restrict__ out, const cf* __restrict__ tw_table,
                                                            const fxc::MixedPlan mp, int64_t num_samp, int nchan, int ntaps,
                                                            int64_t n_pts, int64_t n_streams, int tpr, int n_splits,
                                                            const MixedExtras blu) {
    static_assert(!XF || U == 2, "the fused X stage pairs two antennas");
    static_assert(!BLU || (U == 1 && !XF), "the chirp-z rows go one at a time, F only");
    static_assert(!BIG || (U == 1 && !XF && !BLU && !TWL), "one LDS row: plain F stage, twiddles from the table");
    static_assert(!U8 || XF, "the byte ingest is the two-antenna kernel's");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nfft = BLU ? blu.nfft : nchan;
    const int rpw = (int)blockDim.x / tpr;
    const int sub = (int)threadIdx.x / tpr, lt = (int)threadIdx.x % tpr;
    const int fpg = XF ? rpw : rpw * U;                             // frames per group
    const int64_t gps = (n_pts + fpg - 1) / fpg;                    // groups per stream
    cf* tw_lds = reinterpret_cast<cf*>(smem);
    if (TWL)
        for (int n = threadIdx.x; n < nfft; n += blockDim.x) tw_lds[n] = tw_table[n];      // the first barrier below covers it
    const cf* tw = TWL ? tw_lds : tw_table;
    // a slot of up to 64 threads is one wave (tpr is a power of two): its steps need no workgroup barrier -- LDS operations of
    // a wave complete in order -- only the compiler kept from moving them across; the waves of the workgroup then drift freely
    const bool wave_local = blu.wave_local && tpr <= 64;
    auto slot_sync = [wave_local]() {
        if (wave_local)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        else
            __syncthreads();
    };
    if (TWL && wave_local) __syncthreads();         // the table is the whole workgroup's work
    int64_t g0, g1, s, gi;          // XF: s = chunk, the run stays inside it; else s = stream, the run walks on
    if (XF) {
        const int sp = (int)(blockIdx.x % (unsigned)n_splits);
        s = blockIdx.x / (unsigned)n_splits;
        g0 = (int64_t)sp * gps / n_splits;
        g1 = ((int64_t)sp + 1) * gps / n_splits;
        gi = g0;
    } else {
        const int64_t n_groups = n_streams * gps;
        g0 = (int64_t)blockIdx.x * n_groups / gridDim.x;
        g1 = ((int64_t)blockIdx.x + 1) * n_groups / gridDim.x;
        s = g0 / gps;
        gi = g0 - s * gps;
    }
    fxc::pk2 xacc[XF ? kMixedXPoints : 1];
#pragma unroll
    for (int q = 0; q < (XF ? kMixedXPoints : 1); ++q) xacc[q] = fxc::pk_splat(0.f);
    for (int64_t g = g0; g < g1; ++g) {
        // per-lane and per-shape index math stays inside the group loop: hoisted (the compiler would hoist every radix's
        // share of it), it is a hundred live registers and spills
        int lt_g = lt, sub_g = sub, nch = nchan, tpr_g = tpr, nf = nfft;
        asm volatile("" : "+v"(lt_g), "+v"(sub_g), "+s"(nch), "+s"(tpr_g), "+s"(nf));
        if (!BLU) nf = nch;
        cf* rows = reinterpret_cast<cf*>(smem) + (TWL ? nf : 0) + sub_g * U * (BIG ? 1 : 2) * nf;       // [u][a|b][nfft]
        const int row_stride = 2 * nf;
        // ---- FIR: v[f][m] = sum_t h[t][m] x[(f - t) nchan + nchan - 1 - m]
        const int64_t gf = gi * fpg;                                            // the group's first frame
        const int64_t frame_lo = gf > ntaps - 1 ? gf - (ntaps - 1) : 0;         // the earliest frame any tap reaches
        const int f_end = (int)std::min<int64_t>(n_pts - frame_lo, 1 << 30);    // frames that exist, relative to frame_lo
        if constexpr (XF) {
            const cf* xg0 = x + (2 * s) * num_samp + frame_lo * nch;             // uniform bases; lane offsets stay small
            const cf* xg1 = xg0 + num_samp;
            const unsigned short* xb0 = reinterpret_cast<const unsigned short*>(x) + (2 * s) * num_samp + frame_lo * nch;
            const unsigned short* xb1 = xb0 + num_samp;
            const float k8 = 1.0f / 127.5f;
            fxc::pk2 off0 = fxc::pk_splat(0.f), off1 = off0;
            if constexpr (U8) {
                off0 = fxc::pk(blu.dc_u8[2 * s]);
                off1 = fxc::pk(blu.dc_u8[2 * s + 1]);
            }
            const int f0 = (int)(gf - frame_lo) + sub_g;
            for (int m = lt_g; m < nch; m += tpr_g) {
                fxc::pk2 acc0 = fxc::pk_splat(0.f), acc1 = acc0;
                for (int t0 = 0; t0 < ntaps; t0 += 4) {
                    fxc::pk2 xv0[4], xv1[4];
                    float hw[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int f = f0 - t0 - 3 + j;                          // frame f0 - (t0 + v) at j = 3 - v
                        const bool ok = f >= 0 && f < f_end;
                        const unsigned off = ok ? (unsigned)(f * nch + nch - 1 - m) : 0u;
                        fxc::pk2 l0, l1;
                        if constexpr (U8) {
                            const unsigned r0 = xb0[off], r1 = xb1[off];
                            const fxc::pk2 b0 = {(float)(r0 & 0xFFu), (float)(r0 >> 8)}, b1 = {(float)(r1 & 0xFFu), (float)(r1 >> 8)};
                            l0 = fxc::pk_fma(b0, fxc::pk_splat(k8), off0);
                            l1 = fxc::pk_fma(b1, fxc::pk_splat(k8), off1);
                        } else {
                            l0 = fxc::pk(xg0[off]);
                            l1 = fxc::pk(xg1[off]);
                        }
                        xv0[j] = ok ? l0 : fxc::pk_splat(0.f);
                        xv1[j] = ok ? l1 : fxc::pk_splat(0.f);
                    }
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const bool ok = t0 + v < ntaps;
                        const float ld = h[(unsigned)(ok ? (t0 + v) * nch + m : m)];
                        hw[v] = ok ? ld : 0.f;
                    }
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        acc0 = fxc::pk_fma(fxc::pk_splat(hw[v]), xv0[3 - v], acc0);
                        acc1 = fxc::pk_fma(fxc::pk_splat(hw[v]), xv1[3 - v], acc1);
                    }
                }
                rows[m] = fxc::unpk(acc0);
                rows[row_stride + m] = fxc::unpk(acc1);
            }
        } else {
            const cf* xg = x + s * num_samp + frame_lo * nch;                   // uniform base; lane offsets stay small
            const int f0 = (int)(gf - frame_lo) + sub_g * U;                    // frames f0 .. f0 + U - 1, relative to frame_lo
            for (int m = lt_g; m < nch; m += tpr_g) {
                fxc::pk2 acc[U];
#pragma unroll
                for (int u = 0; u < U; ++u) acc[u] = fxc::pk_splat(0.f);
                for (int t0 = 0; t0 < ntaps; t0 += 4) {
                    fxc::pk2 xv[U + 3];
                    float hw[4];
#pragma unroll
                    for (int j = 0; j < U + 3; ++j) {
                        const int f = f0 - t0 - 3 + j;                          // frame f0 + u - (t0 + v) at j = u - v + 3
                        const bool ok = f >= 0 && f < f_end;
                        const unsigned off = ok ? (unsigned)(f * nch + nch - 1 - m) : 0u;
                        const fxc::pk2 ld = fxc::pk(xg[off]);
                        xv[j] = ok ? ld : fxc::pk_splat(0.f);
                    }
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const bool ok = t0 + v < ntaps;
                        const float ld = h[(unsigned)(ok ? (t0 + v) * nch + m : m)];
                        hw[v] = ok ? ld : 0.f;
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u)
#pragma unroll
                        for (int v = 0; v < 4; ++v) acc[u] = fxc::pk_fma(fxc::pk_splat(hw[v]), xv[u - v + 3], acc[u]);
                }
                if constexpr (BLU) acc[0] = fxc::pk_cmul(acc[0], fxc::pk(blu.chirp[m]));
#pragma unroll
                for (int u = 0; u < U; ++u) rows[u * row_stride + m] = fxc::unpk(acc[u]);
            }
            if constexpr (BLU)
                for (int m = nch + lt_g; m < nf; m += tpr_g) rows[m] = fxc::mk(0.f, 0.f);
        }
        slot_sync();
        int so = 0;
        if constexpr (BIG) {
            const int64_t oc = s / blu.ant;                                    // fpg = 1: the group is one frame
            cf* orow = out + (((oc * n_pts + gf) * blu.ant) + (s - oc * blu.ant)) * nch;
            int ns = 1;
            for (int st = 0; st < mp.n_stages; ++st) {
                const int radix = mp.radix[st];
                int nst = nf;
                asm volatile("" : "+s"(nst));
                if ((st & 1) == 0)
                    fxc::mixed_stage<1>(rows, orow, 0, tw, nst, radix, ns, lt_g, tpr_g);
                else
                    fxc::mixed_stage<1>(orow, rows, 0, tw, nst, radix, ns, lt_g, tpr_g);
                slot_sync();
                ns *= radix;
            }
            if ((mp.n_stages & 1) == 0)
                for (int n = lt_g; n < nch; n += tpr_g) orow[n] = rows[n];
        }
        for (int pass = 0; pass < (BIG ? 0 : (BLU ? 2 : 1)); ++pass) {
            int ns = 1;
            for (int st = 0; st < mp.n_stages; ++st) {
                const int radix = mp.radix[st];
                int nst = nf;
                asm volatile("" : "+s"(nst));                                   // ... and inside the stage loop
                // (the register butterflies for 11 / 13 with two rows: not with the twiddles in L2 too -- 64-bit addresses, spills)
                fxc::mixed_stage<U, (XF && TWL) || U == 1>(rows + so, rows + (nst - so), row_stride, tw, nst, radix, ns, lt_g, tpr_g);
                slot_sync();
                ns *= radix;
                so = nst - so;
            }
            if (BLU && pass == 0) {
                for (int k = lt_g; k < nf; k += tpr_g) {
                    const fxc::pk2 t = fxc::pk_cmul(fxc::pk(rows[so + k]), fxc::pk(blu.d[k]));
                    rows[so + k] = fxc::mk(t[0], -t[1]);
                }
                slot_sync();
            }
        }
        if constexpr (XF) {
            if (gf + sub_g < n_pts) {
#pragma unroll
                for (int q = 0; q < kMixedXPoints; ++q) {
                    const int m = lt_g + q * tpr_g;
                    if (m < nch) {
                        const fxc::pk2 a = fxc::pk(rows[so + m]), b = fxc::pk(rows[row_stride + so + m]);
                        const fxc::pk2 ar = {a[1], -a[0]};                      // a conj(b) = b.x (a.x, a.y) + b.y (a.y, -a.x)
                        xacc[q] = fxc::pk_fma(fxc::pk_splat(b[1]), ar, fxc::pk_fma(fxc::pk_splat(b[0]), a, xacc[q]));
                    }
                }
            }
        } else if constexpr (!BIG) {
            const int64_t oc = s / blu.ant;                                     // chunk and antenna of stream s
            cf* o = out + (((oc * n_pts + gf) * blu.ant) + (s - oc * blu.ant)) * nch;   // uniform base of the group's rows
            const int fstride = blu.ant * nch;                                  // from one frame's row to the next
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int fr = sub_g * U + u;
                if (gf + fr < n_pts)
                    for (int n = lt_g; n < nch; n += tpr_g) {
                        cf v = rows[u * row_stride + so + n];
                        if constexpr (BLU) v = fxc::unpk(fxc::pk_cmul(fxc::pk(fxc::mk(v.x, -v.y)), fxc::pk(blu.chirp[n])));
                        o[(unsigned)(fr * fstride + n)] = v;
                    }
            }
        }
        slot_sync();
        if (++gi == gps && !XF) {
            gi = 0;
            ++s;
        }
    }
    if constexpr (XF) {
        // the slots of a workgroup (rpw > 1) hold sums over different frames of the same bins: add them up through LDS
        cf* red = reinterpret_cast<cf*>(smem) + (TWL ? nchan : 0);              // [rpw][nchan], free after the last barrier
        if (rpw > 1) {
            __syncthreads();                        // every slot is through with its rows (wave-local slots drift apart)
#pragma unroll
            for (int q = 0; q < kMixedXPoints; ++q) {
                const int m = lt + q * tpr;
                if (m < nchan) red[sub * nchan + m] = fxc::unpk(xacc[q]);
            }
            __syncthreads();
        }
        const int sp = (int)(blockIdx.x % (unsigned)n_splits);
        cf* o = out + ((int64_t)sp * n_streams + s) * nchan;                    // n_streams = chunks here
        if (sub == 0) {
#pragma unroll
            for (int q = 0; q < kMixedXPoints; ++q) {
                const int m = lt + q * tpr;
                if (m < nchan) {
                    fxc::pk2 t = xacc[q];
                    for (int r = 1; r < rpw; ++r) t = t + fxc::pk(red[r * nchan + m]);
                    o[m] = fxc::unpk(t);
                }
            }
        }
    }
}

// raw[split][c][p][k] = sum_{i in split} spec[c][a][i][k] * conj(spec[c][b][i][k]); block = kx x iy threads
__global__ __launch_bounds__(256) void xmul_kernel(const cf* __restrict__ spec, cf* __restrict__ raw, int n_ant,
                                                  int n_base, int nchan, int64_t n_pts, int kx, int n_splits,
                                                  int64_t n_chunks) {
    __shared__ cf red[256];
    const int iy = 256 / kx;
    const int tk = threadIdx.x % kx, ti = threadIdx.x / kx;
    const int kblocks = (nchan + kx - 1) / kx;
    const int64_t total = n_chunks * n_base * kblocks * n_splits;
    for (int64_t wid = blockIdx.x; wid < total; wid += gridDim.x) {
        const int split = (int)(wid % n_splits);
        int64_t rest = wid / n_splits;
        const int kb = (int)(rest % kblocks);
        rest /= kblocks;
        const int p = (int)(rest % n_base);
        const int64_t c = rest / n_base;
        // baseline p -> (a, b), ordered (0,1),(0,2),...,(A-2,A-1)
        int a = 0, q = p;
        while (q >= n_ant - 1 - a) { q -= n_ant - 1 - a; ++a; }
        const int b = a + 1 + q;
        const int k = kb * kx + tk;
        float ar = 0.f, ai = 0.f;
        if (k < nchan) {
            const cf* sa = spec + ((c * n_ant + a) * n_pts) * nchan + k;
            const cf* sb = spec + ((c * n_ant + b) * n_pts) * nchan + k;
            for (int64_t i = (int64_t)split * iy + ti; i < n_pts; i += (int64_t)iy * n_splits) {
                const cf u = fxc::st_load(sa + i * nchan), w = fxc::st_load(sb + i * nchan);
                ar += u.x * w.x + u.y * w.y;
                ai += u.y * w.x - u.x * w.y;
            }
        }
        red[threadIdx.x] = fxc::mk(ar, ai);
        __syncthreads();
        if (ti == 0 && k < nchan) {
            for (int j = 1; j < iy; ++j) {
                ar += red[j * kx + tk].x;
                ai += red[j * kx + tk].y;
            }
            raw[(((int64_t)split * n_chunks + c) * n_base + p) * nchan + k] = fxc::mk(ar, ai);
        }
        __syncthreads();
    }
}

}  // namespace
