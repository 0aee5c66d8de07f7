// fxcorr.hip — MI355X (gfx950) F/X hot path: kernels + the C ABI of include/fxcorr.h.
//
// Replaces, for effex's hot path:
//   cusignal.filtering.channelize_poly FIR half   effex/effex.py:553   -> pfb_fir_kernel / fused phase 1
//   cusignal channelize_poly FFT half + conj      effex/effex.py:553   -> fft_pow2_kernel / dft_any_kernel / fused phases 1-3
//   f0 * conj(f1 * rot), mean(axis=0), fftshift   effex/effex.py:516-521 -> xmul_kernel / fused X + finish kernels
//   continuum tail mean_k / bandwidth             effex/effex.py:523-524 -> continuum kernels
// and the steps either side of the path:
//   per-chunk DC removal, uint8 -> complex        effex/effex.py:394-395, :652 -> dc_* / convert_u8 kernels
//   delay calibration                             effex/effex.py:583-627 -> delay_* / stockham_stage kernels
//   per-chunk blocking copies                     effex/effex.py:391-392, 508-509, 693 -> fxc_pipe_* (host side)
// Paths: fused (nchan 4096, ntaps 4; 2 antennas in one kernel, 4/6/8 via F-only + X-engine), tiled (2 antennas,
// nchan 512..8192, any ntaps: the fused design generalised, fx_tiled.h), stream (nchan 1), generic (everything else).  Written for gfx950 only: wave64, 160 KiB LDS, v_permlane32_swap, buffer loads.
// No CPU fallback.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>   // types and prototypes only: the library itself is bound at run time (rccl_api)

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/fxcorr.h"
#include "fx_fused4096.h"
#include "fx_tiled.h"
#include "fx_math.h"

using fxc::cd;
using fxc::cf;
using fxc::f4;

namespace {

constexpr double kTwoPi = 6.283185307179586476925286766559;
constexpr int kMaxTaps = 32;         // cusignal ships 8x8 / 16x16 / 32x32 channeliser kernels only
constexpr int kMaxLdsFftN = 16384;   // 128 KiB of complex64 in LDS
constexpr int64_t kWorkspaceTarget = 8ll << 30;   // upper bound of the lazily grown workspace (288 GB of HBM per GPU)

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

// same transform for any nchan <= 16384 (O(N^2) per row); tw = [N] table, exact index arithmetic
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
                const cf u = sa[i * nchan], w = sb[i * nchan];
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

// ------------------------------------------------------------------------------------------
// finishing kernels (shared by all paths); `slots` = layout of the raw sums inside a row:
//   0 natural bin order, 1 the 2-antenna fused kernel's slot order, 2 the F-only kernel's spectrum order
// ------------------------------------------------------------------------------------------
//   3 the 8192-channel split (§ pfb_split8192_kernel): two 4096-rows of the fused kernel side by side, even bins in
//     the first, odd bins in the second
__device__ __forceinline__ int64_t raw_index(int k, int slots) {
    if (slots == 3) return (int64_t)(k & 1) * fxc::fused::kN + fxc::fused::slot_of_bin(k >> 1);
    return slots == 1 ? fxc::fused::slot_of_bin(k) : (slots == 2 ? fxc::fused::specpos_of_bin(k) : k);
}

// The 2-antenna fused kernel splits the last chunks of a launch (its tail) over workgroups without regard to chunk
// boundaries (fx_fused4096.h::RangeWalk): row c of such a chunk lacks the frames that later workgroups took over, which
// sit in those workgroups' leading-part rows raw[offset + b * nchan ...].  n_frames == 0: every row is complete.
struct LeadRows {
    int64_t first_chunk, n_frames, n_pts, offset;   // the tail: chunks from first_chunk on, n_frames frames in all
    int grid;
};

// slots == 3: row `row` of the caller is the pair of fused-kernel chunks 2 row (even bins) and 2 row + 1 (odd bins)
__device__ __forceinline__ void add_lead_rows(const cf* __restrict__ raw, const LeadRows& lr, int64_t row, int nchan,
                                              int k, int slots, float& ar, float& ai) {
    if (lr.n_frames == 0) return;
    const int64_t vrow = slots == 3 ? 2 * row + (k & 1) : row;
    if (vrow < lr.first_chunk) return;
    const int row_len = slots == 3 ? fxc::fused::kN : nchan;
    const int64_t ridx = slots == 3 ? fxc::fused::slot_of_bin(k >> 1) : raw_index(k, slots);
    const int64_t t = vrow - lr.first_chunk;   // chunk of the tail (fx_fused4096.h::range_walk_tail)
    const int64_t b_lo = fxc::range_owner(t * lr.n_pts, lr.n_frames, lr.grid);
    const int64_t b_hi = fxc::range_owner((t + 1) * lr.n_pts - 1, lr.n_frames, lr.grid);
    for (int64_t b = b_lo + 1; b <= b_hi; ++b) {   // the workgroups that start strictly inside that chunk
        const cf r = raw[lr.offset + b * row_len + ridx];
        ar += r.x;
        ai += r.y;
    }
}

// SPECTRUM rows: out[c][p][(k + N/2) % N] = (sum_split raw) * conj(rot[k]) / n_pts   (effex.py:520-521)
__global__ void rows_spectrum_kernel(const cf* __restrict__ raw, cf* __restrict__ out, const cd* __restrict__ rot,
                                     int nchan, int64_t rows, int n_splits, int64_t split_stride, float inv_pts,
                                     int slots, LeadRows lead) {
    const int64_t total = rows * nchan;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const int k = (int)(idx % nchan);
        const int64_t row = idx / nchan;
        float ar = 0.f, ai = 0.f;
        for (int s = 0; s < n_splits; ++s) {
            const cf r = raw[s * split_stride + row * nchan + raw_index(k, slots)];
            ar += r.x;
            ai += r.y;
        }
        add_lead_rows(raw, lead, row, nchan, k, slots, ar, ai);
        const float cr = (float)rot[k].x, ci = (float)rot[k].y;
        // (ar + i ai) * (cr - i ci)
        const float orr = (ar * cr + ai * ci) * inv_pts;
        const float oi = (ai * cr - ar * ci) * inv_pts;
        int ks = k + nchan / 2;
        if (ks >= nchan) ks -= nchan;
        out[row * nchan + ks] = fxc::mk(orr, oi);
    }
}

__device__ __forceinline__ double block_sum(double v, double* red) {
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}

// CONTINUUM rows: out[row] = mean_k( raw * conj(rot) / n_pts ) / bandwidth   (effex.py:523-524); one WG per row
__global__ __launch_bounds__(256) void rows_continuum_kernel(const cf* __restrict__ raw, cd* __restrict__ out,
                                                            const cd* __restrict__ rot, int nchan, int64_t rows,
                                                            int n_splits, int64_t split_stride, double scale,
                                                            int slots, LeadRows lead) {
    __shared__ double red[256];
    for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
        double ar = 0.0, ai = 0.0;
        for (int k = threadIdx.x; k < nchan; k += blockDim.x) {
            double xr = 0.0, xi = 0.0;
            for (int s = 0; s < n_splits; ++s) {
                const cf r = raw[s * split_stride + row * nchan + raw_index(k, slots)];
                xr += r.x;
                xi += r.y;
            }
            float lr_re = 0.f, lr_im = 0.f;
            add_lead_rows(raw, lead, row, nchan, k, slots, lr_re, lr_im);
            xr += lr_re;
            xi += lr_im;
            const cd w = rot[k];
            ar += xr * w.x + xi * w.y;
            ai += xi * w.x - xr * w.y;
        }
        ar = block_sum(ar, red);
        ai = block_sum(ai, red);
        if (threadIdx.x == 0) {
            cd o;
            o.x = ar * scale;
            o.y = ai * scale;
            out[row] = o;
        }
    }
}

// accumulate: acc[p][k] += sum_split sum_c raw[split][c][p][raw_index(k)]   (fixed order -> reproducible)
__global__ void acc_add_kernel(const cf* __restrict__ raw, cd* __restrict__ acc, int nchan, int n_base,
                               int64_t n_chunks, int n_splits, int slots) {
    const int64_t per_chunk = (int64_t)n_base * nchan;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < per_chunk; idx += stride) {
        const int k = (int)(idx % nchan);
        const int64_t src = (idx / nchan) * nchan + raw_index(k, slots);
        double ar = 0.0, ai = 0.0;
        for (int64_t sc = 0; sc < n_chunks * n_splits; ++sc) {
            const cf r = raw[sc * per_chunk + src];
            ar += r.x;
            ai += r.y;
        }
        cd a = acc[idx];
        a.x += ar;
        a.y += ai;
        acc[idx] = a;
    }
}

// accumulate over many chunks, stage 1: part[split][slot] = sum over this split's rows of the kernels' raw
// float32 rows (slot order, coalesced); fixed order -> bit-reproducible.  Latency-bound (a thread walks its rows one
// load after the other), so the launch uses as many splits as leave each a handful of rows (fused_reduce_splits)
__global__ __launch_bounds__(256) void fused_reduce1_kernel(const cf* __restrict__ raw, cd* __restrict__ part, int nchan,
                                                           int64_t n_rows, int n_splits) {
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    const int split = blockIdx.y;
    if (slot >= nchan) return;
    double ar = 0.0, ai = 0.0;
    for (int64_t c = split; c < n_rows; c += n_splits) {
        const cf r = raw[c * nchan + slot];
        ar += r.x;
        ai += r.y;
    }
    cd o;
    o.x = ar;
    o.y = ai;
    part[(int64_t)split * nchan + slot] = o;
}

// stage 2: acc[bin(slot)] += sum_split part[split][slot]; 16 slots x 16 threads per workgroup, each thread sums every
// 16th split (reads in slot order: coalesced; only the 64 KiB of accumulator updates are scattered by the slot -> bin
// permutation) and the 16 sub-sums are combined in a fixed order
__global__ __launch_bounds__(256) void fused_reduce2_kernel(const cd* __restrict__ part, cd* __restrict__ acc, int nchan,
                                                           int n_splits, int slots) {
    __shared__ cd sub[16][17];
    const int kl = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int slot = blockIdx.x * 16 + kl;
    double ar = 0.0, ai = 0.0;
    if (slot < nchan) {
        for (int s = sl; s < n_splits; s += 16) {
            const cd v = part[(int64_t)s * nchan + slot];
            ar += v.x;
            ai += v.y;
        }
    }
    sub[sl][kl].x = ar;
    sub[sl][kl].y = ai;
    __syncthreads();
    if (sl == 0 && slot < nchan) {
        // slots == 1: the fused kernel's order, slot = q * 512 + tid (fx_fused4096.h::bin_of); 0: natural order
        // 3: the 8192-channel split, [even | odd] halves each in the fused kernel's order
        int k = slot;
        if (slots == 1) k = fxc::fused::bin_of(slot % fxc::fused::kThreads, slot / fxc::fused::kThreads);
        if (slots == 3) {
            const int sl = slot % fxc::fused::kN;
            k = 2 * fxc::fused::bin_of(sl % fxc::fused::kThreads, sl / fxc::fused::kThreads) + slot / fxc::fused::kN;
        }
        cd a = acc[k];
        for (int j = 0; j < 16; ++j) {
            a.x += sub[j][kl].x;
            a.y += sub[j][kl].y;
        }
        acc[k] = a;
    }
}

// multi-antenna X-engine on the F-only kernel's spectra: spec[(c*A + a)*P + i][pos]; one thread per (chunk group,
// pos) keeps all A(A-1)/2 accumulators in registers over the spectra of `cg` consecutive chunks (cg = 1: one raw
// row per chunk; the integration takes float32 sums of up to 256 spectra, like the 2-antenna kernel's rows) and
// reads every spectrum sample exactly once; raw[group][p][pos], baselines ordered (0,1),(0,2)..(A-2,A-1) --
// effex.py:520 for A > 2
#ifndef FXC_XENGINE_UNROLL
#define FXC_XENGINE_UNROLL 2
#endif
constexpr int kXU = FXC_XENGINE_UNROLL;
template <int A>
__global__ __launch_bounds__(256) void xengine_kernel(const cf* __restrict__ spec, cf* __restrict__ raw, int64_t n_pts,
                                                     int nchan, int64_t n_chunks, int cg) {
    constexpr int NB = A * (A - 1) / 2;
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t grp = blockIdx.y;
    float ar[NB], ai[NB];
#pragma unroll
    for (int p = 0; p < NB; ++p) ar[p] = ai[p] = 0.f;
    const int64_t c_end = (grp + 1) * cg < n_chunks ? (grp + 1) * cg : n_chunks;
    for (int64_t c = grp * cg; c < c_end; ++c) {
        const cf* base = spec + (c * A * n_pts) * nchan + pos;
        // kXU spectra per trip: kXU * A independent 8-byte loads in flight before the multiply-accumulates
        int64_t i = 0;
        for (; i + kXU <= n_pts; i += kXU) {
            cf z[kXU][A];
#pragma unroll
            for (int u = 0; u < kXU; ++u)
#pragma unroll
                for (int a = 0; a < A; ++a) z[u][a] = base[((int64_t)a * n_pts + i + u) * nchan];
#pragma unroll
            for (int u = 0; u < kXU; ++u) {
                int p = 0;
#pragma unroll
                for (int a = 0; a < A; ++a)
#pragma unroll
                    for (int b = a + 1; b < A; ++b, ++p) {
                        ar[p] += z[u][a].x * z[u][b].x + z[u][a].y * z[u][b].y;
                        ai[p] += z[u][a].y * z[u][b].x - z[u][a].x * z[u][b].y;
                    }
            }
        }
        for (; i < n_pts; ++i) {
            cf z[A];
#pragma unroll
            for (int a = 0; a < A; ++a) z[a] = base[((int64_t)a * n_pts + i) * nchan];
            int p = 0;
#pragma unroll
            for (int a = 0; a < A; ++a)
#pragma unroll
                for (int b = a + 1; b < A; ++b, ++p) {
                    ar[p] += z[a].x * z[b].x + z[a].y * z[b].y;
                    ai[p] += z[a].y * z[b].x - z[a].x * z[b].y;
                }
        }
    }
#pragma unroll
    for (int p = 0; p < NB; ++p) raw[(grp * NB + p) * nchan + pos] = fxc::mk(ar[p], ai[p]);
}

// sums = [n_base*nchan] raw sums + [1] {count, 0}
__global__ void export_kernel(const cd* __restrict__ acc, cd* __restrict__ sums, int64_t n, double count) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx <= n; idx += stride) {
        cd v;
        if (idx < n) {
            v = acc[idx];
        } else {
            v.x = count;
            v.y = 0.0;
        }
        sums[idx] = v;
    }
}

// out[p][(k + N/2) % N] = sums[p][k] * conj(rot[k]) / count      (effex.py:520-521, integrated)
__global__ void finalize_spectrum_kernel(const cd* __restrict__ sums, cd* __restrict__ out, const cd* __restrict__ rot,
                                         int nchan, int n_base) {
    const int64_t n = (int64_t)n_base * nchan;
    const double inv = 1.0 / sums[n].x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += stride) {
        const int k = (int)(idx % nchan);
        const int64_t p = idx / nchan;
        const cd a = sums[idx], w = rot[k];
        cd o;
        o.x = (a.x * w.x + a.y * w.y) * inv;
        o.y = (a.y * w.x - a.x * w.y) * inv;
        int ks = k + nchan / 2;
        if (ks >= nchan) ks -= nchan;
        out[p * nchan + ks] = o;
    }
}

__global__ __launch_bounds__(256) void finalize_continuum_kernel(const cd* __restrict__ sums, cd* __restrict__ out,
                                                                const cd* __restrict__ rot, int nchan, int n_base,
                                                                double inv_bw) {
    __shared__ double red[256];
    const int64_t n = (int64_t)n_base * nchan;
    const double scale = inv_bw / (sums[n].x * (double)nchan);
    for (int p = blockIdx.x; p < n_base; p += gridDim.x) {
        double ar = 0.0, ai = 0.0;
        for (int k = threadIdx.x; k < nchan; k += blockDim.x) {
            const cd a = sums[(int64_t)p * nchan + k], w = rot[k];
            ar += a.x * w.x + a.y * w.y;
            ai += a.y * w.x - a.x * w.y;
        }
        ar = block_sum(ar, red);
        ai = block_sum(ai, red);
        if (threadIdx.x == 0) {
            cd o;
            o.x = ar * scale;
            o.y = ai * scale;
            out[p] = o;
        }
    }
}

// ------------------------------------------------------------------------------------------
// fused 2-antenna, nchan = 4096, ntaps = 4 kernel (phases in fx_fused4096.h)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// vdst keeps its low half and receives src's low half in its high half; src gets the two high halves
__device__ __forceinline__ void permlane32_swap(cf& a, cf& b) {
    auto rx = __builtin_amdgcn_permlane32_swap(__float_as_uint(a.x), __float_as_uint(b.x), false, false);
    auto ry = __builtin_amdgcn_permlane32_swap(__float_as_uint(a.y), __float_as_uint(b.y), false, false);
    a = fxc::mk(__uint_as_float(rx[0]), __uint_as_float(ry[0]));
    b = fxc::mk(__uint_as_float(rx[1]), __uint_as_float(ry[1]));
}

// this thread's 16 branch samples of frame i: element (255 - j) + 256 (15 - r); loads r = R0 .. R0+CNT-1.
// Buffer loads: one VGPR byte offset per thread, everything that varies with chunk / frame / r is scalar.
typedef unsigned v2u32 __attribute__((ext_vector_type(2)));
#ifndef FXC_LOAD_AUX
#define FXC_LOAD_AUX 0   // cache policy of the IQ stream loads: bit 0 sc0, bit 1 nt, bit 4 sc1
#endif
template <int R0, int CNT>
__device__ __forceinline__ void load_frame_part(cf (&xr)[16], const cf* chunk_base, unsigned chunk_bytes, unsigned voff,
                                                int64_t i) {
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<cf*>(chunk_base), 0, (int)chunk_bytes,
                                                                   0x00020000);
    const unsigned soff = (unsigned)(i * fxc::fused::kN * (int64_t)sizeof(cf));
#pragma unroll
    for (int r = R0; r < R0 + CNT; ++r) {
        const v2u32 d = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff + (unsigned)(256 * (15 - r) * sizeof(cf)), FXC_LOAD_AUX);
        xr[r] = fxc::mk(__uint_as_float(d[0]), __uint_as_float(d[1]));
    }
}

// uint8 ingest (RTL-SDR interleaved I,Q bytes): the same 16 branches as raw byte pairs, one
// 16-bit load each -- a quarter of the complex64 stream's HBM bytes
template <int R0, int CNT>
__device__ __forceinline__ void load_frame_part_u8(cf (&xr)[16], const unsigned short* chunk_base,
                                                   unsigned chunk_bytes, unsigned voff, int64_t i) {
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(chunk_base), 0,
                                                                   (int)chunk_bytes, 0x00020000);
    const unsigned soff = (unsigned)(i * fxc::fused::kN * (int64_t)sizeof(unsigned short));
#pragma unroll
    for (int r = R0; r < R0 + CNT; ++r)   // the byte pair waits in the slot's own register (bit pattern in .x)
        xr[r].x = __uint_as_float(
            (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rs, voff, soff + (unsigned)(256 * (15 - r) * sizeof(unsigned short)), 0));
}

// byte pair -> complex64: (b - 127.5) / 127.5 minus the chunk's mean = b / 127.5 + off, off = -mean_byte / 127.5
// (pyrtlsdr's conversion behind effex.py:652 and the DC removal of effex.py:394-395 in one fused multiply-add)
// (in place: the raw pair sits in the slot's .x register, see load_frame_part_u8)
__device__ __forceinline__ void convert_frame_u8(cf (&h)[16], cf off) {
    const float k = 1.0f / 127.5f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const unsigned raw = __float_as_uint(h[r].x);
        h[r] = fxc::mk(fmaf((float)(raw & 0xFFu), k, off.x), fmaf((float)((raw >> 8) & 0xFFu), k, off.y));
    }
}

// FXC_ABL: developer-only timing ablations (wrong results by design; the shipped build has FXC_ABL == 0):
//   1 no barrier B0, 2 no barrier B1, 4 no exchange-1 LDS traffic, 8 no exchange-2 LDS traffic, 16 no IQ loads.
// FXC_STAMPS: diagnostic build with s_memtime stamps between the phases, summed per wave in scalar
// registers and printed by fxc_kernel_time().  profiles/r01/ablation_and_stamps.md has the readings.
#ifndef FXC_ABL
#define FXC_ABL 0
#endif
#ifndef FXC_STAMPS
#define FXC_STAMPS 0
#endif
constexpr int kStampSegs = 12;
#if FXC_STAMPS
#define FXC_STAMP(k)                                                              \
    do {                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                        \
        const unsigned long long t_now__ = __builtin_amdgcn_s_memtime();          \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                       \
        seg[k] += t_now__ - t_prev;                                               \
        t_prev = t_now__;                                                         \
        __builtin_amdgcn_sched_barrier(0);                                        \
    } while (0)
#else
#define FXC_STAMP(k)
#endif
#if (FXC_ABL & 16)
#define FXC_PREFETCH(R0) ((void)0)
#else
#define FXC_PREFETCH(R0)                                                                                        \
    do {                                                                                                        \
        FXC_SCHED_FENCE();                                                                                      \
        if (U8)                                                                                                 \
            load_frame_part_u8<R0, 4>(nx, reinterpret_cast<const unsigned short*>(x) + (int64_t)pc * 2 * num_samp, \
                                      chunk_bytes, voff, nframe);                                               \
        else                                                                                                    \
            load_frame_part<R0, 4>(nx, nbase, chunk_bytes, voff, nframe);                                       \
        FXC_SCHED_FENCE();                                                                                      \
    } while (0)
#endif

// One spectrum of both antennas: frame i of chunk c sits in ring slot PH.  All control flow is
// wave-uniform and none of it guards a *definition* of ring registers (the prefetch is unconditional),
// which keeps the register allocator from doubling live ranges at merge points.
// SPEC_OUT: the multi-antenna variant -- the pair of streams is only channelised and both spectra go to
// HBM for xengine_kernel (rows_raw then is the spectra buffer [stream][i][specpos]).
// uint8 ingest state: this chunk's conversion offsets
struct U8State {
    cf off;
};

template <int PH, bool SPEC_OUT, bool U8>
__device__ __forceinline__ void fused_step(fxc::fused::State& s, U8State& u8, const cf* dc, const f4* win, cf* region,
                                           const cf* tw2, int tid, const cf* x, int64_t num_samp, unsigned chunk_bytes,
                                           unsigned voff, fxc::fused::RangeWalk& pos, cf* rows_raw,
                                           unsigned long long (&seg)[kStampSegs], unsigned long long& t_prev) {
    using namespace fxc::fused;
    const int64_t c = pos.c, i = pos.i, n_pts = pos.n_pts;   // (the walk itself is 32-bit: scalar registers are scarce)
    FXC_STAMP(0);    // loop overhead and the (rare) row store since the previous step's last stamp
    if (i == 0) {    // zero PFB history at the start of every chunk
        asm volatile("" ::: "memory");   // keep this a (rarely taken) uniform branch, not 96 v_cndmask per frame
        state_reset_history<PH>(s);
        if (U8) u8.off = dc[c * 2 + ((tid >> 8) & 1)];
    }
    if (U8) convert_frame_u8(s.h[PH], u8.off);   // the byte pairs fetched a step ago become the samples of slot PH
    cf v[16];
    phase1_fir<PH>(s, win, tid, v);      // first use of this frame: waits for its loads (issued a step ago)
    FXC_STAMP(2);
    // The oldest ring slot is dead now: refill it with the next frame of this workgroup's range (next frame of
    // the chunk, or frame 0 of the next chunk; at the very end the current frame again, never used).  The 16
    // loads go out in four groups spread over the step: eight waves bursting 16 loads each at the same point
    // stall in the in-order vector-memory issue (measured -7 %).
    int pc, nframe;
    range_walk_prefetch(pos, pc, nframe);
    const cf* nbase = x + (int64_t)pc * 2 * num_samp;
    cf (&nx)[16] = s.h[(PH + 1) & 3];
    FXC_PREFETCH(0);
    fxc::dft16_a(v);
    FXC_PREFETCH(4);
    FXC_STAMP(3);
#if !(FXC_ABL & 1)
    __syncthreads();   // B0: every wave has finished reading the previous spectrum's exchange rows
#endif
    FXC_STAMP(4);
#if !(FXC_ABL & 4)
    // second half of the radix-16 with the twiddle w4096^(j k1) and the exchange-1 store of every output as it forms:
    // the stores are bound by the LDS write path, the butterflies and twiddles run in its shadow (B0 in front of the
    // whole radix-16 instead: +7 %; exchange 2 streamed the same way: spills, +6 %)
    phase1_finish_store(s, v, region, tid);
#endif
    FXC_STAMP(5);
#if !(FXC_ABL & 2)
    __syncthreads();   // B1: exchange-1 rows complete
#endif
    FXC_STAMP(6);
#if !(FXC_ABL & 4)
    phase2_load(region, tid, v);
#endif
    FXC_PREFETCH(8);
    fxc::dft16(v);
    FXC_STAMP(7);
    phase2_twiddle(v, tw2, tid);
    FXC_STAMP(8);
#if !(FXC_ABL & 8)
    wave_sync();       // exchange 2 is a 16x16 transpose inside each 16-lane group: no s_barrier
    phase2_store(v, region, tid);
    wave_sync();
#endif
    FXC_PREFETCH(12);
#if !(FXC_ABL & 8)
    phase3_load(region, tid, v);
#endif
    FXC_STAMP(9);
    fxc::dft16(v);
    if (SPEC_OUT) {
        // stream = 2 * (virtual chunk) + antenna; for a fixed q2 a half-wave stores 256 contiguous bytes.  Buffer
        // stores from the row of antenna 0 of this frame: one VGPR byte offset per thread (antenna 1's row is n_pts
        // rows further on), scalar offsets for q2 -- no per-store address arithmetic on the vector unit
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(rows_raw + (c * 2 * n_pts + i) * kN, 0,
                                                                       (int)((n_pts + 1) * kN * (int64_t)sizeof(cf)), 0x00020000);
        const unsigned soff0 = (unsigned)(((tid >> 5) & 1) * n_pts * kN + lane_specpos(tid)) * (unsigned)sizeof(cf);
#pragma unroll
        for (int q2 = 0; q2 < 16; ++q2) {
            v2u32 d = {__float_as_uint(v[q2].x), __float_as_uint(v[q2].y)};
            __builtin_amdgcn_raw_buffer_store_b64(d, rs, soff0, (unsigned)(q2 * 256 * sizeof(cf)), 0);
        }
    } else {
        // lanes 0-31 hold antenna 0, lanes 32-63 antenna 1 of the same bins: after the swap each lane has both
        // antennas for 8 of them (lanes 0-31: q2 = 0..7, lanes 32-63: q2 = 8..15) -- effex.py:520 without rot
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            cf a = v[q], b = v[q + 8];
            permlane32_swap(a, b);
            xacc(s, q, a, b);
        }
        FXC_STAMP(10);
    }
    // a raw row ends with the last frame of every `unit`-th chunk, of the last chunk and of this workgroup's
    // range: store this lane's 8 bins (fire and forget)
    const bool row_ends = !SPEC_OUT && range_walk_row_ends(pos);
    if (row_ends) {
        cf* row = rows_raw + (int64_t)pos.row * kN + tid;
#pragma unroll
        for (int q = 0; q < kAccPerThread; ++q) {
            row[q * kThreads] = s.acc[q];
            s.acc[q] = fxc::mk(0.f, 0.f);
        }
    }
    range_walk_advance(pos, row_ends);
}

// Work split and raw-row layout: fx_fused4096.h::RangeWalk (whole chunks dealt round-robin, then the last
// n_chunks % (workgroups * seg) chunks as equal frame ranges).
// SPEC_OUT == false: rows_raw = range_rows() raw rows, float32, slot order (fx_fused4096.h::slot_of_bin);
// rows_are_chunks: row c = chunk c (+ leading-part rows for tail chunks shared by several workgroups), else rows of
// `unit` chunks whose total is the integration.
// SPEC_OUT == true: rows_raw[(2c + ant) * n_pts + i][specpos] = the spectra themselves.  A "chunk" here is a pair
// of consecutive antenna streams, so an even number of antennas [n_chunks][A][S] is simply n_chunks * A/2 pairs.
// U8: x points at interleaved uint8 I,Q ([chunk][antenna][num_samp] byte pairs) and dc[chunk * 2 + antenna] holds the
// conversion offsets (-mean_byte / 127.5, or -1 without DC removal) of each stream.  stamps: diagnostic builds only.
template <bool SPEC_OUT, bool U8 = false>
__global__ __launch_bounds__(fxc::fused::kThreads, 2) void fx_fused4096_kernel(
    const cf* __restrict__ x, int64_t num_samp, int64_t n_pts, int64_t n_chunks, const f4* __restrict__ win_g,
    const cf* __restrict__ tw1_g, const cf* __restrict__ tw2_g, cf* __restrict__ rows_raw,
    unsigned long long* __restrict__ stamps, const cf* __restrict__ dc, int seg, int unit, int rows_are_chunks) {
    using namespace fxc::fused;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f4* win = reinterpret_cast<f4*>(smem + kLdsWin);
    cf* region = reinterpret_cast<cf*>(smem + kLdsRegion);
    cf* tw2 = reinterpret_cast<cf*>(smem + kLdsTw2);

    const int tid = threadIdx.x;
    const int ant = tid >> 8, j = tid & 255;
    for (int idx = tid; idx < kN; idx += kThreads) win[idx] = win_g[idx];
    if (tid < 256) tw2[tid] = tw2_g[tid];
    State s;
    state_load_twiddles(s, tw1_g, tid);
#pragma unroll
    for (int q = 0; q < kAccPerThread; ++q) s.acc[q] = fxc::mk(0.f, 0.f);
    __syncthreads();

    constexpr int64_t kSampleBytes = U8 ? sizeof(unsigned short) : sizeof(cf);
    const unsigned voff = (unsigned)((ant * num_samp + (255 - j)) * kSampleBytes);
    const unsigned chunk_bytes = (unsigned)(2 * num_samp * kSampleBytes);
    U8State u8;
    u8.off = fxc::mk(0.f, 0.f);
    unsigned long long seg_t[kStampSegs] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_prev = 0;
    int64_t frames_done = 0;
#if FXC_STAMPS
    t_prev = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll 1
    for (int part = 0; part < 2; ++part) {   // 0: whole chunks, round-robin; 1: this workgroup's range of the tail
        RangeWalk pos = part == 0 ? range_walk_rounds(blockIdx.x, gridDim.x, (int)n_chunks, (int)n_pts, seg, unit, rows_are_chunks != 0)
                                  : range_walk_tail(blockIdx.x, gridDim.x, (int)n_chunks, (int)n_pts, seg, unit, rows_are_chunks != 0);
        const int total = pos.left;
        if (!SPEC_OUT && part == 1 && (!pos.lead || total == 0)) {   // no leading part: its row reads as zeros
            const RangeSplit sp = range_split(gridDim.x, (int)n_chunks, seg, unit, rows_are_chunks != 0);
            cf* lead_row = rows_raw + (int64_t)(sp.rows_rounds + sp.n_tail + blockIdx.x) * kN + tid;
#pragma unroll
            for (int q = 0; q < kAccPerThread; ++q) lead_row[q * kThreads] = fxc::mk(0.f, 0.f);
        }
        if (total == 0) continue;
        if (U8) u8.off = dc[(int64_t)pos.c * 2 + ant];
        // ring prologue: frame i -> slot 0, its history i-1, i-2, i-3 -> slots 3, 2, 1 (zeros before the chunk start)
        const cf* cbase = reinterpret_cast<const cf*>(reinterpret_cast<const char*>(x) + (int64_t)pos.c * 2 * num_samp * kSampleBytes);
#pragma unroll
        for (int d = 1; d < 4; ++d) {
            if (pos.i - d >= 0) {
                if (U8) {
                    load_frame_part_u8<0, 16>(s.h[4 - d], reinterpret_cast<const unsigned short*>(cbase), chunk_bytes, voff, pos.i - d);
                    convert_frame_u8(s.h[4 - d], u8.off);
                } else {
                    load_frame_part<0, 16>(s.h[4 - d], cbase, chunk_bytes, voff, pos.i - d);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) s.h[4 - d][r] = fxc::mk(0.f, 0.f);
            }
        }
        if (U8)
            load_frame_part_u8<0, 16>(s.h[0], reinterpret_cast<const unsigned short*>(cbase), chunk_bytes, voff, pos.i);
        else
            load_frame_part<0, 16>(s.h[0], cbase, chunk_bytes, voff, pos.i);
        // frame g of the part sits in ring slot g & 3: unrolled by four so the ring rotates by register renaming
#pragma unroll 1
        for (int g = 0; g < total; g += 4) {
            fused_step<0, SPEC_OUT, U8>(s, u8, dc, win, region, tw2, tid, x, num_samp, chunk_bytes, voff, pos, rows_raw, seg_t, t_prev);
            if (g + 1 < total)
                fused_step<1, SPEC_OUT, U8>(s, u8, dc, win, region, tw2, tid, x, num_samp, chunk_bytes, voff, pos, rows_raw, seg_t, t_prev);
            if (g + 2 < total)
                fused_step<2, SPEC_OUT, U8>(s, u8, dc, win, region, tw2, tid, x, num_samp, chunk_bytes, voff, pos, rows_raw, seg_t, t_prev);
            if (g + 3 < total)
                fused_step<3, SPEC_OUT, U8>(s, u8, dc, win, region, tw2, tid, x, num_samp, chunk_bytes, voff, pos, rows_raw, seg_t, t_prev);
        }
        frames_done += total;
    }
#if FXC_STAMPS
    if (stamps && (tid & 63) == 0) {
        unsigned long long* dst = stamps + ((int64_t)blockIdx.x * (kThreads / 64) + (tid >> 6)) * kStampSegs;
        for (int k = 0; k < kStampSegs; ++k) dst[k] = seg_t[k];
        dst[kStampSegs - 1] = (unsigned long long)frames_done;
    }
#else
    (void)stamps;
    (void)frames_done;
#endif
}

// ------------------------------------------------------------------------------------------
// tiled fused 2-antenna kernel for nchan in {512, 1024, 2048, 4096, 8192}, any ntaps (phases in fx_tiled.h)
// ------------------------------------------------------------------------------------------
// PFB FIR of frame i for butterfly u: buffer loads, one VGPR byte offset per thread (xoff into the chunk's
// stream pair, hoff into the window), everything that varies with frame / tap / branch is scalar
template <class G>
__device__ __forceinline__ void tiled_fir(cf (&v)[16], const cf* chunk_base, unsigned chunk_bytes, unsigned xoff,
                                          const float* win, unsigned win_bytes, unsigned hoff, int64_t i, int ntaps) {
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<cf*>(chunk_base), 0, (int)chunk_bytes,
                                                                   0x00020000);
    __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(win), 0, (int)win_bytes, 0x00020000);
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = fxc::mk(0.f, 0.f);
    const int tmax = (int64_t)(ntaps - 1) < i ? ntaps - 1 : (int)i;
    for (int t = 0; t <= tmax; ++t) {
        const unsigned sx = (unsigned)((i - t) * G::N * (int64_t)sizeof(cf));
        const unsigned sh = (unsigned)(t * G::N * (int)sizeof(float));
        cf xv[16];
        float hv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const v2u32 d = __builtin_amdgcn_raw_buffer_load_b64(rx, xoff, sx + (unsigned)(G::P * (15 - r) * sizeof(cf)), 0);
            xv[r] = fxc::mk(__uint_as_float(d[0]), __uint_as_float(d[1]));
            hv[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rh, hoff, sh + (unsigned)(G::P * r * sizeof(float)), 0));
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = fxc::cfma(hv[r], xv[r], v[r]);
    }
}

// Two frames per pass over the PFB history: v0 = FIR of frame i, v1 = FIR of frame i + 1 (computed only if
// two == true).  At tap t the pass holds x[i + 1 - t] and x[i - t]; the next tap re-uses the older one and
// loads one new frame, and every window coefficient is loaded once for both outputs: (ntaps + 1) frame loads
// and ntaps window loads per two spectra instead of 2 ntaps of each.
template <class G>
__device__ __forceinline__ void tiled_fir2(cf (&v0)[16], cf (&v1)[16], bool two, const cf* chunk_base,
                                           unsigned chunk_bytes, unsigned xoff, const float* win, unsigned win_bytes,
                                           unsigned hoff, int64_t i, int ntaps) {
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<cf*>(chunk_base), 0, (int)chunk_bytes,
                                                                   0x00020000);
    __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(win), 0, (int)win_bytes, 0x00020000);
    auto load_frame = [&](cf (&dst)[16], int64_t frame, bool present) {
        if (present) {
            const unsigned sx = (unsigned)(frame * G::N * (int64_t)sizeof(cf));
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const v2u32 d = __builtin_amdgcn_raw_buffer_load_b64(rx, xoff, sx + (unsigned)(G::P * (15 - r) * sizeof(cf)), 0);
                dst[r] = fxc::mk(__uint_as_float(d[0]), __uint_as_float(d[1]));
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[r] = fxc::mk(0.f, 0.f);
        }
    };
    auto tap = [&](int t, const cf (&xa)[16], const cf (&xb)[16]) {   // xa = x[i + 1 - t], xb = x[i - t]
        const unsigned sh = (unsigned)(t * G::N * (int)sizeof(float));
        float hv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r)
            hv[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rh, hoff, sh + (unsigned)(G::P * r * sizeof(float)), 0));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            v1[r] = fxc::cfma(hv[r], xa[r], v1[r]);
            v0[r] = fxc::cfma(hv[r], xb[r], v0[r]);
        }
    };
#pragma unroll
    for (int r = 0; r < 16; ++r) v0[r] = v1[r] = fxc::mk(0.f, 0.f);
    cf xw[2][16];
    load_frame(xw[0], i + 1, two);
    load_frame(xw[1], i, true);
    // taps in pairs so that the two-frame window rotates by renaming; a frame before the chunk start is zero
    for (int t = 0; t < ntaps; t += 2) {
        tap(t, xw[0], xw[1]);
        load_frame(xw[0], i - t - 1, i - t - 1 >= 0);       // x[i - (t + 1)]: the older frame of tap t + 1
        if (t + 1 < ntaps) {
            tap(t + 1, xw[1], xw[0]);
            load_frame(xw[1], i - t - 2, i - t - 2 >= 0);
        }
    }
}

// F-only tail of a tiled step: the two spectra of frame i leave in natural bin order.  Stage C leaves bin
// bin_of(u, k2) in v[k2]; the exchange region serves as a transposition buffer (bin k at k + (k >> 4): the
// 16 lanes of a group write 17 or R0 + 1/16 slots apart, conflict-free) and the rows go out 256 B per half-wave.
// valid: this lane's stream exists (an odd stream count leaves the last pair half empty).
template <class G>
__device__ __forceinline__ void tiled_store_spectrum(const cf (&v)[16], cf* reg, int u, cf* out_row, bool valid) {
    __syncthreads();   // every wave holds its stage-C outputs in registers: the rows can be overwritten
    // bin_of(u, k2) = b0 + C k2 with C a multiple of 16, and P is one too: both index maps are one base + constants
    constexpr int C = (G::A3 ? 256 : 16) * G::R0;
    const int b0 = G::bin_of(u, 0);
    cf* wr = reg + b0 + (b0 >> 4);
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) wr[(C + C / 16) * k2] = v[k2];
    __syncthreads();
    if (valid) {
        const cf* rd = reg + u + (u >> 4);
        cf* dst = out_row + u;
#pragma unroll
        for (int n = 0; n < 16; ++n) dst[G::P * n] = fxc::fused::lds_load(rd + (G::P + G::P / 16) * n);
    }
}

// raw[(split * n_chunks + c) * N + k] = sum over the split's frames of spec0[i,k] * conj(spec1[i,k]), natural
// bin order, float32.  Work item = (split, chunk); a split is a contiguous range of a chunk's frames (the
// FIR reads its history from memory, so ranges are independent).
// SPEC: F-only -- a "chunk" is a pair of consecutive streams (n_streams of them in all), raw is the spectra
// buffer [stream][i][k] and n_chunks the number of pairs.
template <class G, bool SPEC>
__global__ __launch_bounds__(G::kThreads) void fx_tiled_kernel(const cf* __restrict__ x, int64_t num_samp, int64_t n_pts,
                                                               int64_t n_chunks, int n_splits, int ntaps,
                                                               const float* __restrict__ win, const cf* __restrict__ tw0_g,
                                                               const cf* __restrict__ twA_g, const cf* __restrict__ tw16_g,
                                                               cf* __restrict__ raw, int64_t n_streams) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cf* region = reinterpret_cast<cf*>(smem + G::kLdsRegion);
    cf* tw16 = reinterpret_cast<cf*>(smem + G::kLdsTw16);
    const int tid = threadIdx.x;
    const int u = G::u_of(tid), ant = G::ant_of(tid);
    for (int idx = tid; idx < 256; idx += G::kThreads) tw16[idx] = tw16_g[idx];
    cf tw0[16], twA[16];
    if (G::R0 > 1) G::load_tw0(tw0, tw0_g, u);
    if (G::A3) G::load_twA(twA, twA_g, u);
    __syncthreads();
    cf* reg = region + ant * G::kRegion;
    const unsigned win_bytes = (unsigned)(ntaps * G::N * (int)sizeof(float));
    const unsigned hoff = (unsigned)(u * (int)sizeof(float));
    const int64_t per = (n_pts + n_splits - 1) / n_splits;
    for (int64_t w = blockIdx.x; w < n_chunks * n_splits; w += gridDim.x) {
        const int64_t c = w % n_chunks, split = w / n_chunks;
        const int64_t i0 = split * per, i1 = (i0 + per < n_pts) ? i0 + per : n_pts;
        const cf* chunk_base = x + c * 2 * num_samp;
        // F-only with an odd stream count: the missing second stream of the last pair re-reads the first
        const bool valid = !SPEC || (2 * c + ant) < n_streams;
        const int ant_ld = valid ? ant : 0;
        const unsigned chunk_bytes = (unsigned)((SPEC && 2 * c + 1 >= n_streams ? 1 : 2) * num_samp * (int64_t)sizeof(cf));
        const unsigned xoff = (unsigned)((ant_ld * num_samp + (G::P - 1 - u)) * (int64_t)sizeof(cf));
        cf acc[G::kAccPerThread];
#pragma unroll
        for (int q = 0; q < G::kAccPerThread; ++q) acc[q] = fxc::mk(0.f, 0.f);
        // everything after the FIR for one frame
        auto finish = [&](cf (&v)[16], int64_t i) {
            if (G::R0 > 1) G::prestage(v, tw0);
            if (G::A3) {
                if (G::R0 > 1) {
                    __syncthreads();   // every wave has finished reading the previous spectrum's exchange rows
                    G::store0(v, reg, u);
                    __syncthreads();
                    G::loadA(reg, u, v);
                }
                fxc::dft16(v);
                __syncthreads();
                G::twiddleA_store(v, twA, reg, u);
                __syncthreads();
            } else {
                __syncthreads();
                G::store0(v, reg, u);
                __syncthreads();
            }
            G::loadB(reg, u, v);
            fxc::dft16(v);
            G::twiddleB(v, tw16, u);
            wave_sync();       // the 16x16 transpose stays inside each 16-lane group: no s_barrier
            G::storeT(v, reg, u);
            wave_sync();
            G::loadC(reg, u, v);
            fxc::dft16(v);
            if (SPEC) {
                tiled_store_spectrum<G>(v, reg, u, raw + ((2 * c + ant) * n_pts + i) * G::N, valid);
                return;
            }
            // lanes 0-31 hold antenna 0, lanes 32-63 antenna 1 of the same bins (see fused_step)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                cf a = v[q], b = v[q + 8];
                permlane32_swap(a, b);
                acc[q] = fxc::cadd(acc[q], fxc::cmulc(a, b));
            }
        };
        if (G::kThreads <= 512) {   // two frames per pass over the history (the 1024-thread geometry has no registers for it)
            for (int64_t i = i0; i < i1; i += 2) {
                cf v0[16], v1[16];
                const bool two = i + 1 < i1;
                tiled_fir2<G>(v0, v1, two, chunk_base, chunk_bytes, xoff, win, win_bytes, hoff, i, ntaps);
                finish(v0, i);
                if (two) finish(v1, i + 1);
            }
        } else {
            for (int64_t i = i0; i < i1; ++i) {
                cf v[16];
                tiled_fir<G>(v, chunk_base, chunk_bytes, xoff, win, win_bytes, hoff, i, ntaps);
                finish(v, i);
            }
        }
        if (SPEC) continue;
        cf* row = raw + (split * n_chunks + c) * G::N;
#pragma unroll
        for (int q = 0; q < G::kAccPerThread; ++q) row[G::bin_of(u, q + 8 * ant)] = acc[q];
    }
}

// ntaps <= 4, nchan <= 2048 variant of the tiled kernel: every IQ sample is fetched once into a VGPR ring of
// four frames (as in fx_fused4096_kernel) and the window sits in LDS.
template <class G, int R0, int CNT>
__device__ __forceinline__ void tiled_load_part(cf (&xr)[16], const cf* chunk_base, unsigned chunk_bytes, unsigned xoff,
                                                int64_t frame) {
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<cf*>(chunk_base), 0, (int)chunk_bytes,
                                                                   0x00020000);
    const unsigned soff = (unsigned)(frame * G::N * (int64_t)sizeof(cf));
#pragma unroll
    for (int r = R0; r < R0 + CNT; ++r) {
        const v2u32 d = __builtin_amdgcn_raw_buffer_load_b64(rs, xoff, soff + (unsigned)(G::P * (15 - r) * sizeof(cf)), 0);
        xr[r] = fxc::mk(__uint_as_float(d[0]), __uint_as_float(d[1]));
    }
}

// uint8 ingest (see load_frame_part_u8): chunk_base then points at byte pairs
template <class G, int R0, int CNT>
__device__ __forceinline__ void tiled_load_part_u8(cf (&xr)[16], const cf* chunk_base, unsigned chunk_bytes,
                                                   unsigned xoff, int64_t frame) {
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<cf*>(chunk_base), 0, (int)chunk_bytes,
                                                                   0x00020000);
    const unsigned soff = (unsigned)(frame * G::N * (int64_t)sizeof(unsigned short));
#pragma unroll
    for (int r = R0; r < R0 + CNT; ++r)
        xr[r].x = __uint_as_float(
            (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rs, xoff, soff + (unsigned)(G::P * (15 - r) * sizeof(unsigned short)), 0));
}

template <class G>
struct TiledRing {
    cf h[4][16];
    cf tw0[16];   // pre-stage twiddles (R0 > 1)
    cf twA[16];   // stage-A twiddles (nchan 4096)
    cf acc[G::kAccPerThread];
    U8State u8;   // uint8 ingest only
};

#define FXC_TILED_PREFETCH(R0)                                                                  \
    do {                                                                                        \
        FXC_SCHED_FENCE();                                                                      \
        if (U8)                                                                                 \
            tiled_load_part_u8<G, R0, 4>(nx, chunk_base, chunk_bytes, xoff, nframe);            \
        else                                                                                    \
            tiled_load_part<G, R0, 4>(nx, chunk_base, chunk_bytes, xoff, nframe);               \
        FXC_SCHED_FENCE();                                                                      \
    } while (0)

// one spectrum of both antennas; frame i sits in ring slot PH, i1 = end of this work item's frame range
template <class G, int PH, bool SPEC, bool U8>
__device__ __forceinline__ void tiled_ring_step(TiledRing<G>& s, const f4* win, cf* reg, const cf* tw16, int u,
                                                const cf* chunk_base, unsigned chunk_bytes, unsigned xoff, int64_t i,
                                                int64_t i1, cf* out_row, bool valid) {
    if (U8) convert_frame_u8(s.h[PH], s.u8.off);   // the byte pairs fetched a step ago become the samples of slot PH
    cf v[16];
    G::template fir_ring<PH>(s.h, win, u, v);
    // the oldest slot is dead: refill it with the next frame of the range (the current one again at the end,
    // never used) -- unconditional so that no branch guards a definition of ring registers
    const int64_t nframe = (i + 1 < i1) ? i + 1 : i;
    cf (&nx)[16] = s.h[(PH + 1) & 3];
    FXC_TILED_PREFETCH(0);
    if (G::A3) {
        static_assert(!(G::A3 && G::R0 > 1), "ring variant: nchan <= 4096");
        fxc::dft16(v);
        FXC_TILED_PREFETCH(4);
        __syncthreads();   // every wave has finished reading the previous spectrum's exchange rows
        G::twiddleA_store(v, s.twA, reg, u);
        __syncthreads();
    } else {
        G::prestage(v, s.tw0);
        FXC_TILED_PREFETCH(4);
        __syncthreads();
        G::store0(v, reg, u);
        __syncthreads();
    }
    G::loadB(reg, u, v);
    FXC_TILED_PREFETCH(8);
    fxc::dft16(v);
    G::twiddleB(v, tw16, u);
    wave_sync();
    G::storeT(v, reg, u);
    wave_sync();
    FXC_TILED_PREFETCH(12);
    G::loadC(reg, u, v);
    fxc::dft16(v);
    if (SPEC) {
        tiled_store_spectrum<G>(v, reg, u, out_row + i * G::N, valid);
        return;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        cf a = v[q], b = v[q + 8];
        permlane32_swap(a, b);
        s.acc[q] = fxc::cadd(s.acc[q], fxc::cmulc(a, b));
    }
}

template <class G, bool SPEC, bool U8 = false>
__global__ __launch_bounds__(G::kThreads, 2) void fx_tiled_ring_kernel(const cf* __restrict__ x, int64_t num_samp,
                                                                      int64_t n_pts, int64_t n_chunks, int n_splits,
                                                                      const f4* __restrict__ win_g,
                                                                      const cf* __restrict__ tw0_g,
                                                                      const cf* __restrict__ twA_g,
                                                                      const cf* __restrict__ tw16_g, cf* __restrict__ raw,
                                                                      int64_t n_streams, const cf* __restrict__ dc) {
    static_assert(!(SPEC && U8), "uint8 ingest: F+X only");
    constexpr int64_t kSampleBytes = U8 ? sizeof(unsigned short) : sizeof(cf);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cf* region = reinterpret_cast<cf*>(smem + G::kLdsRegion);
    cf* tw16 = reinterpret_cast<cf*>(smem + G::kLdsTw16);
    f4* win = reinterpret_cast<f4*>(smem + G::kLdsWin);
    const int tid = threadIdx.x;
    const int u = G::u_of(tid), ant = G::ant_of(tid);
    for (int idx = tid; idx < 256; idx += G::kThreads) tw16[idx] = tw16_g[idx];
    for (int idx = tid; idx < G::N; idx += G::kThreads) win[idx] = win_g[idx];
    TiledRing<G> s;
    if (G::R0 > 1) G::load_tw0(s.tw0, tw0_g, u);
    if (G::A3) G::load_twA(s.twA, twA_g, u);
    __syncthreads();
    cf* reg = region + ant * G::kRegion;
    const int64_t per = (n_pts + n_splits - 1) / n_splits;
    for (int64_t w = blockIdx.x; w < n_chunks * n_splits; w += gridDim.x) {
        const int64_t c = w % n_chunks, split = w / n_chunks;
        const int64_t i0 = split * per, i1 = (i0 + per < n_pts) ? i0 + per : n_pts;
        const cf* chunk_base = reinterpret_cast<const cf*>(reinterpret_cast<const char*>(x) + c * 2 * num_samp * kSampleBytes);
        const bool valid = !SPEC || (2 * c + ant) < n_streams;   // see fx_tiled_kernel
        const int ant_ld = valid ? ant : 0;
        const unsigned chunk_bytes = (unsigned)((SPEC && 2 * c + 1 >= n_streams ? 1 : 2) * num_samp * kSampleBytes);
        const unsigned xoff = (unsigned)((ant_ld * num_samp + (G::P - 1 - u)) * kSampleBytes);
        if (U8) s.u8.off = dc[c * 2 + ant];
        cf* out_row = SPEC ? raw + (2 * c + ant) * n_pts * G::N : nullptr;
#pragma unroll
        for (int q = 0; q < G::kAccPerThread; ++q) s.acc[q] = fxc::mk(0.f, 0.f);
        // ring prologue: frame i0 -> slot 0, its history i0-1, i0-2, i0-3 -> slots 3, 2, 1 (zero before the chunk)
#pragma unroll
        for (int d = 1; d < 4; ++d) {
            if (i0 - d >= 0 && i0 < i1) {   // (an empty range at the end of a chunk loads nothing)
                if (U8) {
                    tiled_load_part_u8<G, 0, 16>(s.h[4 - d], chunk_base, chunk_bytes, xoff, i0 - d);
                    convert_frame_u8(s.h[4 - d], s.u8.off);
                } else {
                    tiled_load_part<G, 0, 16>(s.h[4 - d], chunk_base, chunk_bytes, xoff, i0 - d);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) s.h[4 - d][r] = fxc::mk(0.f, 0.f);
            }
        }
        if (i0 < i1) {
            if (U8)
                tiled_load_part_u8<G, 0, 16>(s.h[0], chunk_base, chunk_bytes, xoff, i0);
            else
                tiled_load_part<G, 0, 16>(s.h[0], chunk_base, chunk_bytes, xoff, i0);
        }
        for (int64_t i = i0; i < i1; i += 4) {
            tiled_ring_step<G, 0, SPEC, U8>(s, win, reg, tw16, u, chunk_base, chunk_bytes, xoff, i, i1, out_row, valid);
            if (i + 1 < i1)
                tiled_ring_step<G, 1, SPEC, U8>(s, win, reg, tw16, u, chunk_base, chunk_bytes, xoff, i + 1, i1, out_row, valid);
            if (i + 2 < i1)
                tiled_ring_step<G, 2, SPEC, U8>(s, win, reg, tw16, u, chunk_base, chunk_bytes, xoff, i + 2, i1, out_row, valid);
            if (i + 3 < i1)
                tiled_ring_step<G, 3, SPEC, U8>(s, win, reg, tw16, u, chunk_base, chunk_bytes, xoff, i + 3, i1, out_row, valid);
        }
        if (SPEC) continue;
        cf* row = raw + (split * n_chunks + c) * G::N;
#pragma unroll
        for (int q = 0; q < G::kAccPerThread; ++q) row[G::bin_of(u, q + 8 * ant)] = s.acc[q];
    }
}

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
// outputs i0 .. i0 + TP - 1 from the block in flight (xn) and the one before it (xo), stored as they are formed
template <int TP>
__device__ __forceinline__ void prefilter_fir_store(const cf (&xo)[TP], const cf (&xn)[TP], const float (&hc)[TP],
                                                    __amdgpu_buffer_rsrc_t rs, unsigned voff, int64_t i0, int64_t i_end,
                                                    unsigned frame_bytes) {
    const bool full = i0 + TP <= i_end;    // wave-uniform
#pragma unroll
    for (int k = 0; k < TP; ++k) {
        float ar = 0.f, ai = 0.f;
#pragma unroll
        for (int t = 0; t < TP; ++t) {
            const cf v = (k - t >= 0) ? xn[(k - t) >= 0 ? k - t : 0] : xo[(TP + k - t) < TP ? TP + k - t : 0];
            ar = fmaf(hc[t], v.x, ar);
            ai = fmaf(hc[t], v.y, ai);
        }
        if (full || i0 + k < i_end) {
            v2u32 d = {__float_as_uint(ar), __float_as_uint(ai)};
            __builtin_amdgcn_raw_buffer_store_b64(d, rs, voff, (unsigned)(i0 + k) * frame_bytes, 0);
        }
    }
}

// frames i0 .. i0 + TP - 1 of this thread's sample position: buffer loads, one VGPR byte offset, scalar frame offsets.
// Frames past the stream's last one are clamped to it (a later frame never feeds an earlier output, and outputs past
// the end are not stored); frames before its first one read as zeros.
template <int TP>
__device__ __forceinline__ void prefilter_load(cf (&xr)[TP], __amdgpu_buffer_rsrc_t rs, unsigned voff, int64_t i0, int64_t n_pts,
                                               unsigned frame_bytes) {
#pragma unroll
    for (int k = 0; k < TP; ++k) {
        const int64_t i = i0 + k;
        const int64_t ic = i < 0 ? 0 : (i < n_pts ? i : n_pts - 1);
        const v2u32 d = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, (unsigned)ic * frame_bytes, 0);
        const bool zero = i < 0;   // wave-uniform
        xr[k] = fxc::mk(zero ? 0.f : __uint_as_float(d[0]), zero ? 0.f : __uint_as_float(d[1]));
    }
}

// hcoef[t][n] = h[t N + (N - 1 - n)] for t < ntaps, zero rows up to TP; grid (N / 256, streams, frame splits)
// (asking for 3 waves per SIMD at TP = 32 makes the compiler spill and the pass 3 % slower: measured)
template <int TP>
__global__ __launch_bounds__(256) void pfb_prefilter_kernel(const cf* __restrict__ x, cf* __restrict__ y,
                                                           const float* __restrict__ hcoef, int64_t num_samp, int nchan,
                                                           int64_t n_pts, int64_t per_split) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int64_t s = blockIdx.y;
    const int64_t i_begin = (int64_t)blockIdx.z * per_split;
    const int64_t i_end = (i_begin + per_split < n_pts) ? i_begin + per_split : n_pts;
    if (i_begin >= i_end) return;
    float hc[TP];
#pragma unroll
    for (int t = 0; t < TP; ++t) hc[t] = hcoef[(int64_t)t * nchan + n];
    const unsigned stream_bytes = (unsigned)(num_samp * (int64_t)sizeof(cf));       // num_samp <= 2^27
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<cf*>(x + s * num_samp), 0, (int)stream_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(y + s * num_samp, 0, (int)stream_bytes, 0x00020000);
    const unsigned voff = (unsigned)n * (unsigned)sizeof(cf);
    const unsigned frame_bytes = (unsigned)nchan * (unsigned)sizeof(cf);
    cf xa[TP], xb[TP];
    prefilter_load<TP>(xa, rx, voff, i_begin - TP, n_pts, frame_bytes);     // history (zeros before the stream's start)
    for (int64_t i0 = i_begin; i0 < i_end; i0 += 2 * TP) {                  // two blocks per trip: the pair swaps roles, no copies
        prefilter_load<TP>(xb, rx, voff, i0, n_pts, frame_bytes);
        prefilter_fir_store<TP>(xa, xb, hc, ry, voff, i0, i_end, frame_bytes);
        if (i0 + TP >= i_end) break;
        prefilter_load<TP>(xa, rx, voff, i0 + TP, n_pts, frame_bytes);
        prefilter_fir_store<TP>(xb, xa, hc, ry, voff, i0 + TP, i_end, frame_bytes);
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
template <int TP>
__global__ __launch_bounds__(256) void pfb_split8192_kernel(const cf* __restrict__ x, cf* __restrict__ y,
                                                           const float* __restrict__ hcoef, const cf* __restrict__ tw,
                                                           int64_t num_samp, int64_t n_pts, int64_t per_split) {
    constexpr int kHalf = 4096, kFull = 8192;
    const int n = blockIdx.x * 256 + threadIdx.x;          // position inside the half frame
    const int64_t s = blockIdx.y;                           // stream = chunk * 2 + antenna
    const int64_t i_begin = (int64_t)blockIdx.z * per_split;
    const int64_t i_end = (i_begin + per_split < n_pts) ? i_begin + per_split : n_pts;
    if (i_begin >= i_end) return;
    float hc[TP][2];
#pragma unroll
    for (int t = 0; t < TP; ++t) {
        hc[t][0] = hcoef[(int64_t)t * kFull + n];
        hc[t][1] = hcoef[(int64_t)t * kFull + n + kHalf];
    }
    const cf w = tw[n];
    const int64_t half_samp = n_pts * kHalf;
    const int64_t c = s >> 1, a = s & 1;
    const unsigned in_bytes = (unsigned)(num_samp * (int64_t)sizeof(cf)), out_bytes = (unsigned)(half_samp * (int64_t)sizeof(cf));
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<cf*>(x + s * num_samp), 0, (int)in_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(y + ((c * 2 + 0) * 2 + a) * half_samp, 0, (int)out_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(y + ((c * 2 + 1) * 2 + a) * half_samp, 0, (int)out_bytes, 0x00020000);
    const unsigned voff = (unsigned)n * (unsigned)sizeof(cf);
    const unsigned in_frame = kFull * (unsigned)sizeof(cf), out_frame = kHalf * (unsigned)sizeof(cf), hi = kHalf * (unsigned)sizeof(cf);
    cf xa[TP][2], xb[TP][2];
    auto load = [&](cf (&xr)[TP][2], int64_t i0) {
#pragma unroll
        for (int k = 0; k < TP; ++k) {
            const int64_t i = i0 + k;
            const int64_t ic = i < 0 ? 0 : (i < n_pts ? i : n_pts - 1);      // see prefilter_load
            const bool zero = i < 0;
            const v2u32 d0 = __builtin_amdgcn_raw_buffer_load_b64(rx, voff, (unsigned)ic * in_frame, 0);
            const v2u32 d1 = __builtin_amdgcn_raw_buffer_load_b64(rx, voff, (unsigned)ic * in_frame + hi, 0);
            xr[k][0] = fxc::mk(zero ? 0.f : __uint_as_float(d0[0]), zero ? 0.f : __uint_as_float(d0[1]));
            xr[k][1] = fxc::mk(zero ? 0.f : __uint_as_float(d1[0]), zero ? 0.f : __uint_as_float(d1[1]));
        }
    };
    auto fir_store = [&](const cf (&xo)[TP][2], const cf (&xn)[TP][2], int64_t i0) {
        const bool full = i0 + TP <= i_end;
#pragma unroll
        for (int k = 0; k < TP; ++k) {
            cf yv[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float ar = 0.f, ai = 0.f;
#pragma unroll
                for (int t = 0; t < TP; ++t) {
                    const cf v = (k - t >= 0) ? xn[(k - t) >= 0 ? k - t : 0][h] : xo[(TP + k - t) < TP ? TP + k - t : 0][h];
                    ar = fmaf(hc[t][h], v.x, ar);
                    ai = fmaf(hc[t][h], v.y, ai);
                }
                yv[h] = fxc::mk(ar, ai);
            }
            if (full || i0 + k < i_end) {
                const cf ea = fxc::cadd(yv[1], yv[0]), eb = fxc::cmul(fxc::csub(yv[1], yv[0]), w);
                const unsigned soff = (unsigned)(i0 + k) * out_frame;
                v2u32 da = {__float_as_uint(ea.x), __float_as_uint(ea.y)}, db = {__float_as_uint(eb.x), __float_as_uint(eb.y)};
                __builtin_amdgcn_raw_buffer_store_b64(da, ra, voff, soff, 0);
                __builtin_amdgcn_raw_buffer_store_b64(db, rb, voff, soff, 0);
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
// 2c + 1); the chunk rows themselves go through fused_reduce1/2_kernel in layout 3
__global__ __launch_bounds__(256) void split_lead_acc_kernel(const cf* __restrict__ raw, cd* __restrict__ acc, LeadRows lr) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;    // 0 .. 8191
    const int64_t slot = fxc::fused::slot_of_bin(k >> 1);
    double ar = 0.0, ai = 0.0;
    for (int b = 0; b < lr.grid; ++b) {
        // the fused chunk workgroup b's range starts in (its leading part, if any, belongs to that chunk)
        const int64_t vc = lr.first_chunk + fxc::range_begin(b, (int)lr.n_frames, lr.grid) / lr.n_pts;
        if ((vc & 1) != (k & 1)) continue;
        const cf r = raw[lr.offset + (int64_t)b * fxc::fused::kN + slot];
        ar += r.x;
        ai += r.y;
    }
    cd v = acc[k];
    v.x += ar;
    v.y += ai;
    acc[k] = v;
}

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

// ntaps <= 4, even num_samp: a lane takes one sample pair of both streams with an aligned 16-byte load; the two
// earlier pairs the FIR needs come from the neighbouring lanes (v_mov_b32_dpp wave_shr:1), so a wave covers 62 new
// pairs plus 2 halo lanes and every pair is loaded exactly once per wave (the halo from L1 instead cost 5 %)
__device__ __forceinline__ float lane_shr1(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xF, 0xF, false));
}
__device__ __forceinline__ v4f32 wave_shr1(v4f32 v) {
    const float x = lane_shr1(v.x), y = lane_shr1(v.y), z = lane_shr1(v.z), w = lane_shr1(v.w);
    v4f32 r = {x, y, z, w};
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
    float ar = 0.f, ai = 0.f;
    for (int64_t base = p0 + wave * 62; base < p1; base += 4 * 62) {      // wave-uniform trip count
        const int64_t m = base + lane - 2;
        const bool in_range = m >= 0 && m < pairs;
        const v4f32 a2 = in_range ? s0[m] : zero, b2 = in_range ? s1[m] : zero;
        const v4f32 a1 = wave_shr1(a2), b1 = wave_shr1(b2);
        const v4f32 a0 = wave_shr1(a1), b0 = wave_shr1(b1);
        const float y0er = h0 * a2[0] + h1 * a1[2] + h2 * a1[0] + h3 * a0[2];
        const float y0ei = h0 * a2[1] + h1 * a1[3] + h2 * a1[1] + h3 * a0[3];
        const float y0or = h0 * a2[2] + h1 * a2[0] + h2 * a1[2] + h3 * a1[0];
        const float y0oi = h0 * a2[3] + h1 * a2[1] + h2 * a1[3] + h3 * a1[1];
        const float y1er = h0 * b2[0] + h1 * b1[2] + h2 * b1[0] + h3 * b0[2];
        const float y1ei = h0 * b2[1] + h1 * b1[3] + h2 * b1[1] + h3 * b0[3];
        const float y1or = h0 * b2[2] + h1 * b2[0] + h2 * b1[2] + h3 * b1[0];
        const float y1oi = h0 * b2[3] + h1 * b2[1] + h2 * b1[3] + h3 * b1[1];
        if (lane >= 2 && m < p1) {
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
    for (int64_t n = lo + threadIdx.x; n < hi; n += blockDim.x) {
        const cf v = x[s * num_samp + n];
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

__global__ __launch_bounds__(256) void dc_sum_u8_kernel(const unsigned char* __restrict__ x, double* __restrict__ part,
                                                       int64_t num_samp, int n_slices) {
    __shared__ double red[256];
    const int64_t s = blockIdx.y;
    const int slice = blockIdx.x;
    const int64_t per = (num_samp + n_slices - 1) / n_slices;
    const int64_t lo = slice * per, hi = (lo + per < num_samp) ? lo + per : num_samp;
    unsigned long long ar = 0, ai = 0;      // byte sums are exact
    for (int64_t n = lo + threadIdx.x; n < hi; n += blockDim.x) {
        const unsigned short v = reinterpret_cast<const unsigned short*>(x)[s * num_samp + n];
        ar += v & 0xFF;
        ai += v >> 8;
    }
    const double sr = block_sum((double)ar, red);
    const double si = block_sum((double)ai, red);
    if (threadIdx.x == 0) {
        part[(s * n_slices + slice) * 2] = sr;
        part[(s * n_slices + slice) * 2 + 1] = si;
    }
}

// fused uint8 ingest: exact byte sums of whole streams, one workgroup per stream, 16-byte loads (8 samples per lane);
// part[s * 2] = sum of I bytes, part[s * 2 + 1] = sum of Q bytes  (the n_slices = 1 layout of dc_sum_u8_kernel)
__global__ __launch_bounds__(256) void dc_sum_u8_stream_kernel(const unsigned char* __restrict__ x, double* __restrict__ part,
                                                              int64_t num_samp, int64_t n_streams) {
    __shared__ double red[256];
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    for (int64_t s = blockIdx.x; s < n_streams; s += gridDim.x) {
        const unsigned char* base = x + s * num_samp * 2;
        // align to 16 bytes: head and tail bytes one sample at a time
        const int64_t head = (int64_t)(((16 - (reinterpret_cast<uintptr_t>(base) & 15)) & 15) / 2);
        const int64_t h = head < num_samp ? head : num_samp;
        const int64_t n_vec = (num_samp - h) / 8;
        const v4u* vp = reinterpret_cast<const v4u*>(base + h * 2);
        unsigned long long ar = 0, ai = 0;
        for (int64_t n = threadIdx.x; n < n_vec; n += 256) {
            const v4u w = vp[n];
            unsigned si = 0, sq = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                si = __builtin_amdgcn_sad_u8(w[k] & 0x00FF00FFu, 0u, si);
                sq = __builtin_amdgcn_sad_u8((w[k] >> 8) & 0x00FF00FFu, 0u, sq);
            }
            ar += si;
            ai += sq;
        }
        const unsigned short* sp = reinterpret_cast<const unsigned short*>(base);
        for (int64_t n = threadIdx.x; n < h; n += 256) {
            ar += sp[n] & 0xFF;
            ai += sp[n] >> 8;
        }
        for (int64_t n = h + n_vec * 8 + threadIdx.x; n < num_samp; n += 256) {
            ar += sp[n] & 0xFF;
            ai += sp[n] >> 8;
        }
        const double sr = block_sum((double)ar, red);
        const double si2 = block_sum((double)ai, red);
        if (threadIdx.x == 0) {
            part[s * 2] = sr;
            part[s * 2 + 1] = si2;
        }
    }
}

// out = x - mean (complex64 in place or out of place)
__global__ void dc_apply_c64_kernel(const cf* __restrict__ x, cf* __restrict__ out, const double* __restrict__ part,
                                    int64_t num_samp, int n_slices, int64_t total) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const int64_t s = idx / num_samp;
        double mr = 0.0, mi = 0.0;
        for (int k = 0; k < n_slices; ++k) {
            mr += part[(s * n_slices + k) * 2];
            mi += part[(s * n_slices + k) * 2 + 1];
        }
        const cf v = x[idx];
        out[idx] = fxc::mk((float)((double)v.x - mr / (double)num_samp), (float)((double)v.y - mi / (double)num_samp));
    }
}

// out = (byte - 127.5) / 127.5 [- mean]; remove_dc == 0 keeps the mean
__global__ void convert_u8_kernel(const unsigned char* __restrict__ x, cf* __restrict__ out, const double* __restrict__ part,
                                  int64_t num_samp, int n_slices, int64_t total, int remove_dc) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const int64_t s = idx / num_samp;
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
        const unsigned short v = reinterpret_cast<const unsigned short*>(x)[idx];
        // ((b - 127.5) - (mean_b - 127.5)) / 127.5 = (b - mean_b) / 127.5, formed in float64, rounded once
        out[idx] = fxc::mk((float)(((double)(v & 0xFF) - mr) / 127.5), (float)(((double)(v >> 8) - mi) / 127.5));
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

// ------------------------------------------------------------------------------------------
// synthetic IQ (effex_amd/synth.py, bit for bit)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void synth_kernel(cf* __restrict__ x, uint64_t seed, int64_t first_chunk, int64_t n_chunks, int n_ant,
                             int64_t num_samp, const int* __restrict__ delays, const cf* __restrict__ tone,
                             int tone_period, const float* __restrict__ lut) {
    const int64_t total = n_chunks * n_ant * num_samp;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const uint64_t key_seed = seed * 0x8CB92BA72F3D8DD7ull;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const int64_t n = idx % num_samp;
        const int64_t ca = idx / num_samp;
        const int a = (int)(ca % n_ant);
        const int64_t c = ca / n_ant;
        const uint64_t g = (uint64_t)(1 << 20) + (uint64_t)((first_chunk + c) * num_samp) + (uint64_t)n;
        const uint64_t gd = g - (uint64_t)delays[a];
        const uint64_t hs = mix64(key_seed + gd);   // stream 0 = sky
        const uint64_t hr = mix64(key_seed + (uint64_t)(a + 1) * 0xD1B54A32D192ED03ull + g);
        const cf t = tone[(int)(gd % (uint64_t)tone_period)];
        const float s_re = lut[hs & 0xFF], s_im = lut[(hs >> 8) & 0xFF];
        const float r_re = lut[hr & 0xFF], r_im = lut[(hr >> 8) & 0xFF];
        // (s + 0.5 r) + t with one rounding per step; 0.5*r is exact
        const float re = __fadd_rn(__fadd_rn(s_re, 0.5f * r_re), t.x);
        const float im = __fadd_rn(__fadd_rn(s_im, 0.5f * r_im), t.y);
        x[idx] = fxc::mk(re, im);
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
thread_local std::string g_lib_error;

}  // namespace


// ------------------------------------------------------------------------------------------
// host-fed front end: double-buffered pinned staging, H2D / compute / D2H on three
// streams chained by events, so batch k+1 crosses PCIe while batch k is on the CUs.  Replaces the
// reference's per-chunk blocking copies (effex/effex.py:391-392, 508-509, 693).
// ------------------------------------------------------------------------------------------
struct fxc_pipe_slot {
    void* h_in = nullptr;    // pinned
    void* h_out = nullptr;   // pinned
    void* d_in = nullptr;
    void* d_out = nullptr;
    hipEvent_t ev_in = nullptr, ev_compute = nullptr, ev_out = nullptr;
    bool busy = false;
};

struct fxc_plan {
    int device = 0, cu_count = 0;
    int n_ant = 0, n_base = 0, nchan = 0, ntaps = 0;
    int64_t num_samp = 0, n_pts = 0;
    int path = FXC_PATH_GENERIC;
    bool pow2 = false;
    int lg2n = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev_order = nullptr;   // orders the old stream's work before the new one's (fxc_set_stream)
    int live_pipes = 0;              // fxc_pipe objects that hold a pointer to this plan
    // device tables
    float* d_win = nullptr;        // [ntaps*nchan] float (generic)
    cf* d_tw = nullptr;            // generic FFT twiddles
    cd* d_rot = nullptr;           // [nchan]
    f4* d_win4 = nullptr;          // fused
    cf* d_tw1 = nullptr;
    cf* d_tw2 = nullptr;
    cf* d_tw0 = nullptr;           // tiled: pre-stage twiddles [16][nchan/16]
    int tiled_grid_max = 0, tiled_grid_max_f = 0;   // resident workgroups of the F+X / F-only tiled kernels
    bool tiled_f = false;          // the F-only tiled kernel serves fxc_channelize
    bool tiled_ring = false;       // ntaps <= 4 and nchan <= 4096: VGPR frame ring + window in LDS
    bool prefilter = false;        // ntaps > 4: pfb_prefilter_kernel first, then the tiled kernels with one unit tap
    int pre_tp = 0;                // its register block: 8, 16 or 32 frames
    float* d_hpre = nullptr;       // [pre_tp][nchan] reversed polyphase coefficients
    float* d_ones = nullptr;       // [nchan] unit window of the plain tiled kernel behind the pre-filter
    void* d_pre = nullptr;         // pre-filtered streams of one pass
    size_t pre_bytes = 0;
    bool split8192 = false;        // nchan 8192, 2 antennas: pfb_split8192_kernel + the 4096-channel fused kernel
    cf* d_tw8192 = nullptr;        // [4096] w8192^(4095 - n')
    unsigned long long* d_stamps = nullptr;   // diagnostic builds only
    int fused_grid_max = 0;
    int64_t fused_seg = 1;         // chunks per round-robin segment of the fused kernel (fx_fused4096.h::RangeWalk)
    cd* d_acc = nullptr;           // [n_base*nchan]
    cd* d_sums = nullptr;          // [n_base*nchan + 1]
    cd* d_out = nullptr;           // finalize staging [n_base*nchan]
    cd* h_out = nullptr;           // its pinned host mirror: a D2H copy into pageable memory costs ~30 us of staging
    double spectra_count = 0.0;
    // workspace (grown on demand)
    void* d_ws = nullptr;
    int64_t ws_bytes = 0;
    void* d_stage[3] = {nullptr, nullptr, nullptr};   // host-buffer calls: device copies of x and out; uint8 calls on
    size_t stage_bytes[3] = {0, 0, 0};                // plans without the fused ingest: the converted samples
    void* d_dc = nullptr;                            // uint8 ingest: byte sums + conversion offsets per stream
    size_t dc_bytes = 0;
    // timing
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;
    bool profiling = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> kev;
    double kernel_ms = 0.0;
    int64_t kernel_launches = 0;
    int stamp_grid = 0;
    StreamTaps taps;               // nchan == 1: the FIR taps by value
    mutable std::string error;
};

struct fxc_pipe {
    fxc_plan* plan = nullptr;
    int64_t chunks = 0;
    int depth = 0, mode = FXC_MODE_SPECTRUM;
    double bandwidth = 1.0;
    size_t in_bytes = 0, out_bytes = 0;
    bool counted = false;      // registered in plan->live_pipes
    bool u8 = false;           // batches are RTL-SDR byte pairs (fxc_pipe_create_u8)
    int remove_dc = 0;
    hipStream_t s_in = nullptr, s_out = nullptr;
    std::vector<fxc_pipe_slot> slots;
    int64_t pushed = 0, popped = 0;
};

namespace {

int fail(const fxc_plan* p, int status, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (p)
        p->error = buf;
    else
        g_lib_error = buf;
    return status;
}

// Every ABI entry runs on the plan's device and leaves the caller's current device as it found it (a process
// that drives several GPUs, or torch's own notion of the current device, must not see it change).
struct DeviceGuard {
    int prev = -1;
    bool changed = false, ok = true;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) {
            ok = hipSetDevice(device) == hipSuccess;
            changed = ok && prev >= 0;
        }
    }
    ~DeviceGuard() {
        if (changed) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define FXC_DEVICE(p, device)                                                                  \
    DeviceGuard device_guard__(device);                                                        \
    if (!device_guard__.ok) return fail(p, FXC_ERR_HIP, "hipSetDevice(%d) failed", (int)(device))

#define FXC_HIP(p, call)                                                                                       \
    do {                                                                                                       \
        hipError_t e__ = (call);                                                                               \
        if (e__ != hipSuccess)                                                                                 \
            return fail(p, e__ == hipErrorOutOfMemory ? FXC_ERR_NOMEM : FXC_ERR_HIP, "%s failed: %s", #call,   \
                        hipGetErrorString(e__));                                                               \
    } while (0)

int grid_for(int64_t work_items, int block, int cu_count) {
    int64_t g = (work_items + block - 1) / block;
    const int64_t cap = (int64_t)cu_count * 8;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

int ensure_ws(fxc_plan* p, int64_t bytes) {
    if (bytes <= p->ws_bytes) return FXC_OK;
    if (p->d_ws) {
        FXC_HIP(p, hipStreamSynchronize(p->stream));
        FXC_HIP(p, hipFree(p->d_ws));
        p->d_ws = nullptr;
        p->ws_bytes = 0;
    }
    FXC_HIP(p, hipMalloc(&p->d_ws, (size_t)bytes));
    p->ws_bytes = bytes;
    return FXC_OK;
}

// grow-only device buffer owned by the plan (staging, conversion offsets)
int grow(fxc_plan* p, void** buf, size_t* have, size_t want) {
    if (want <= *have) return FXC_OK;
    FXC_HIP(p, hipStreamSynchronize(p->stream));
    if (*buf) (void)hipFree(*buf);
    *buf = nullptr;
    *have = 0;
    const hipError_t e = hipMalloc(buf, want);
    if (e != hipSuccess) return fail(p, FXC_ERR_NOMEM, "allocation of %zu bytes failed: %s", want, hipGetErrorString(e));
    *have = want;
    return FXC_OK;
}

struct KernelTimer {
    fxc_plan* p;
    hipEvent_t a = nullptr, b = nullptr;
    explicit KernelTimer(fxc_plan* plan) : p(plan) {
        if (p->profiling && hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess)
            (void)hipEventRecord(a, p->stream);
        else
            a = b = nullptr;
    }
    void stop() {
        if (a) {
            (void)hipEventRecord(b, p->stream);
            p->kev.emplace_back(a, b);
            a = b = nullptr;
        }
    }
};

int drain_kernel_events(fxc_plan* p) {
    for (auto& e : p->kev) {
        FXC_HIP(p, hipEventSynchronize(e.second));
        float ms = 0.f;
        FXC_HIP(p, hipEventElapsedTime(&ms, e.first, e.second));
        p->kernel_ms += ms;
        p->kernel_launches += 1;
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    p->kev.clear();
    return FXC_OK;
}

int launch_fused(fxc_plan* p, const cf* x, int64_t n_pairs, cf* out, bool spec_out, const cf* dc_u8 = nullptr,
                 int64_t unit = 1, bool rows_are_chunks = true, int64_t num_samp = 0);

// F-stage of `n_streams` streams: x -> spec (both device, natural bin order)
int tiled_channelize(fxc_plan* p, const cf* x, cf* spec, int64_t n_streams);

int run_channelize(fxc_plan* p, const cf* x, cf* spec, int64_t n_streams) {
    if (n_streams == 0 || p->n_pts == 0) return FXC_OK;
    // the F-only tiled kernel writes natural-order spectra straight to `spec` (no workspace): any caller may use it
    if (p->tiled_f) return tiled_channelize(p, x, spec, n_streams);
    const int64_t total = n_streams * p->n_pts * p->nchan;
    hipLaunchKernelGGL(pfb_fir_kernel, dim3(grid_for(total, 256, p->cu_count)), dim3(256), 0, p->stream, x, p->d_win,
                       spec, p->num_samp, p->nchan, p->ntaps, p->n_pts, total);
    const int64_t rows = n_streams * p->n_pts;
    if (p->nchan > 1) {
        const int grid = (int)std::min<int64_t>(rows, (int64_t)p->cu_count * 4);
        const size_t lds = (size_t)std::max(p->nchan, p->pow2 ? 512 : 0) * sizeof(cf);   // small N: 512 / N rows per workgroup
        if (p->pow2)
            hipLaunchKernelGGL(fft_pow2_kernel, dim3(grid), dim3(256), lds, p->stream, spec, p->d_tw, p->nchan,
                               p->lg2n, rows);
        else
            hipLaunchKernelGGL(dft_any_kernel, dim3(grid), dim3(256), lds, p->stream, spec, p->d_tw, p->nchan, rows);
    }
    FXC_HIP(p, hipGetLastError());
    return FXC_OK;
}

struct XGeom {
    int kx, n_splits;
};

XGeom x_geometry(const fxc_plan* p) {
    XGeom g;
    g.kx = 1;
    while (g.kx < 256 && g.kx < p->nchan) g.kx <<= 1;
    const int iy = 256 / g.kx;
    int64_t splits = (p->n_pts + (int64_t)iy * 64 - 1) / ((int64_t)iy * 64);
    if (splits < 1) splits = 1;
    if (splits > 256) splits = 256;
    g.n_splits = (int)splits;
    return g;
}

// chunks per pass on the generic path so that spectra + raw sums fit the workspace target
int64_t generic_chunks_per_pass(const fxc_plan* p, int64_t n_chunks, const XGeom& g, int64_t* spec_bytes,
                                int64_t* raw_bytes) {
    const int64_t spec_per_chunk = (int64_t)p->n_ant * p->n_pts * p->nchan * (int64_t)sizeof(cf);
    const int64_t raw_per_chunk = (int64_t)g.n_splits * p->n_base * p->nchan * (int64_t)sizeof(cf);
    int64_t cb = kWorkspaceTarget / std::max<int64_t>(1, spec_per_chunk + raw_per_chunk);
    if (cb < 1) cb = 1;
    if (cb > n_chunks) cb = n_chunks;
    *spec_bytes = (cb * spec_per_chunk + 255) / 256 * 256;
    *raw_bytes = cb * raw_per_chunk;
    return cb;
}

constexpr int kFusedReduceSplits = 256;   // most splits of the two-stage reduce over raw rows (size of `part`)
// splits that leave each thread of stage 1 about four rows to walk
int fused_reduce_splits(int64_t n_rows) { return (int)std::max<int64_t>(16, std::min<int64_t>(kFusedReduceSplits, n_rows / 4)); }

// workgroups of a fused launch over n_pairs chunk pairs: one per CU; a launch with fewer chunks than that is all
// tail (frame ranges), on fewer workgroups when a range would be under four frames (each reloads up to three
// frames of history)
int fused_grid(const fxc_plan* p, int64_t n_pairs) {
    if (n_pairs >= p->fused_grid_max) return p->fused_grid_max;
    const int64_t frames = n_pairs * p->n_pts;
    return (int)std::max<int64_t>(1, std::min<int64_t>(frames / 4, p->fused_grid_max));
}

// chunks per raw row when only the integration is wanted: float32 sums of up to 256 spectra
int64_t fused_unit(const fxc_plan* p) { return std::max<int64_t>(1, std::min<int64_t>(256 / std::max<int64_t>(1, p->n_pts), 64)); }

// raw rows a 2-antenna fused launch over nc chunks writes (leading-part rows included)
int64_t fused_rows(const fxc_plan* p, int64_t nc, int64_t unit, bool rows_are_chunks) {
    return fxc::fused::range_split(fused_grid(p, nc), (int)nc, (int)p->fused_seg, (int)unit, rows_are_chunks).n_rows;
}

LeadRows fused_lead(const fxc_plan* p, int64_t nc) {
    const fxc::fused::RangeSplit sp = fxc::fused::range_split(fused_grid(p, nc), (int)nc, (int)p->fused_seg, 1, true);
    LeadRows lr;
    lr.first_chunk = sp.n_full;
    lr.n_frames = sp.n_tail * p->n_pts;
    lr.n_pts = p->n_pts;
    lr.offset = nc * (int64_t)fxc::fused::kN;
    lr.grid = fused_grid(p, nc);
    return lr;
}
const LeadRows kNoLead = {0, 0, 0, 0, 0};

// n_pairs = pairs of consecutive antenna streams to channelise; spec_out: write spectra instead of X sums
// dc_u8 != nullptr: x is the uint8 I,Q stream and dc_u8 its per-stream conversion offsets (2 antennas, X fused in)
// unit / rows_are_chunks: the raw-row layout (fx_fused4096.h::RangeWalk)
int launch_fused(fxc_plan* p, const cf* x, int64_t n_pairs, cf* out, bool spec_out, const cf* dc_u8, int64_t unit,
                 bool rows_are_chunks, int64_t num_samp) {
    using namespace fxc::fused;
    if (num_samp == 0) num_samp = p->num_samp;      // (the 8192-channel split runs on half-size streams)
    const int grid = fused_grid(p, n_pairs);
    const int seg = (int)p->fused_seg;
    if (n_pairs * p->n_pts >= (1ll << 31)) return fail(p, FXC_ERR_ARG, "more than 2^31 frames in one launch");
    unsigned long long* stamps = nullptr;
#if FXC_STAMPS
    if (!p->d_stamps) FXC_HIP(p, hipMalloc(&p->d_stamps, (size_t)p->fused_grid_max * 8 * kStampSegs * 8));
    FXC_HIP(p, hipMemsetAsync(p->d_stamps, 0, (size_t)p->fused_grid_max * 8 * kStampSegs * 8, p->stream));
    stamps = p->d_stamps;
    p->stamp_grid = grid;
#endif
    KernelTimer kt(p);
    if (dc_u8)
        hipLaunchKernelGGL((fx_fused4096_kernel<false, true>), dim3(grid), dim3(kThreads), kLdsBytes, p->stream, x,
                           num_samp, p->n_pts, n_pairs, p->d_win4, p->d_tw1, p->d_tw2, out, stamps, dc_u8, seg, (int)unit,
                           rows_are_chunks ? 1 : 0);
    else if (spec_out)
        hipLaunchKernelGGL((fx_fused4096_kernel<true, false>), dim3(grid), dim3(kThreads), kLdsBytes, p->stream, x,
                           num_samp, p->n_pts, n_pairs, p->d_win4, p->d_tw1, p->d_tw2, out, stamps, (const cf*)nullptr,
                           seg, 1, 1);
    else
        hipLaunchKernelGGL((fx_fused4096_kernel<false, false>), dim3(grid), dim3(kThreads), kLdsBytes, p->stream, x,
                           num_samp, p->n_pts, n_pairs, p->d_win4, p->d_tw1, p->d_tw2, out, stamps, (const cf*)nullptr,
                           seg, (int)unit, rows_are_chunks ? 1 : 0);
    kt.stop();
    FXC_HIP(p, hipGetLastError());
    return FXC_OK;
}

// layout of the raw per-chunk sums the fused paths produce (see raw_index)
int fused_layout(const fxc_plan* p) { return p->n_ant == 2 ? 1 : (p->path == FXC_PATH_FUSED ? 2 : 0); }

// chunks per pass on the fused paths: 2 antennas only need the raw rows; more antennas also the spectra
int64_t fused_chunks_per_pass(const fxc_plan* p, int64_t n_chunks, int64_t* spec_bytes, int64_t* raw_bytes) {
    const int64_t raw_per_chunk = (int64_t)p->n_base * p->nchan * (int64_t)sizeof(cf);
    const int64_t spec_per_chunk = p->n_ant == 2 ? 0 : (int64_t)p->n_ant * p->n_pts * p->nchan * (int64_t)sizeof(cf);
    int64_t cb = kWorkspaceTarget / (raw_per_chunk + spec_per_chunk);
    if (cb < 1) cb = 1;
    if (cb > n_chunks) cb = n_chunks;
    if (p->n_ant > 2 && cb > 65535) cb = 65535;   // xengine_kernel carries the chunk in grid.y
    *spec_bytes = (cb * spec_per_chunk + 255) / 256 * 256;
    // 2 antennas: one leading-part row per workgroup after the chunk rows (fx_fused4096_kernel)
    *raw_bytes = ((cb + (p->n_ant == 2 ? p->fused_grid_max : 0)) * raw_per_chunk + 255) / 256 * 256;
    return cb;
}

// raw[c][p][layout] for nc chunks starting at x; spec = scratch for the multi-antenna path.  2 antennas: rows of
// `unit` chunks + leading-part rows (fused_rows() of them in all)
int fused_raw_sums(fxc_plan* p, const cf* x, int64_t nc, cf* spec, cf* raw, const cf* dc_u8 = nullptr, int64_t unit = 1,
                   bool rows_are_chunks = true) {
    using namespace fxc::fused;
    if (p->n_ant == 2) return launch_fused(p, x, nc, raw, false, dc_u8, unit, rows_are_chunks);
    // 4 / 6 / 8 antennas: spectra to HBM (the F-only fused kernel in its own spectrum order at nchan 4096 / ntaps 4,
    // the F-only tiled kernel in natural order otherwise), then the register-resident X-engine.  unit = chunks per
    // raw row here too: ceil(nc / unit) rows come out
    int rc = p->path == FXC_PATH_FUSED ? launch_fused(p, x, nc * (p->n_ant / 2), spec, true)
                                       : tiled_channelize(p, x, spec, nc * p->n_ant);
    if (rc) return rc;
    const int cg = (int)unit;
    const dim3 grid(p->nchan / 256, (unsigned)((nc + cg - 1) / cg));
    switch (p->n_ant) {
        case 3: hipLaunchKernelGGL(xengine_kernel<3>, grid, dim3(256), 0, p->stream, spec, raw, p->n_pts, p->nchan, nc, cg); break;
        case 4: hipLaunchKernelGGL(xengine_kernel<4>, grid, dim3(256), 0, p->stream, spec, raw, p->n_pts, p->nchan, nc, cg); break;
        case 5: hipLaunchKernelGGL(xengine_kernel<5>, grid, dim3(256), 0, p->stream, spec, raw, p->n_pts, p->nchan, nc, cg); break;
        case 6: hipLaunchKernelGGL(xengine_kernel<6>, grid, dim3(256), 0, p->stream, spec, raw, p->n_pts, p->nchan, nc, cg); break;
        case 7: hipLaunchKernelGGL(xengine_kernel<7>, grid, dim3(256), 0, p->stream, spec, raw, p->n_pts, p->nchan, nc, cg); break;
        case 8: hipLaunchKernelGGL(xengine_kernel<8>, grid, dim3(256), 0, p->stream, spec, raw, p->n_pts, p->nchan, nc, cg); break;
        default: return fail(p, FXC_ERR_UNSUPPORTED, "no X-engine instantiation for n_ant=%d", p->n_ant);
    }
    FXC_HIP(p, hipGetLastError());
    return FXC_OK;
}

// ---- tiled path -------------------------------------------------------------------------------------
template <class G, bool SPEC>
const void* tiled_fn(const fxc_plan* p, int* lds) {
    *lds = G::kLdsBytes;
    if constexpr (G::N <= 4096) {
        if (p->tiled_ring) {
            *lds = G::kLdsBytesRing;
            return reinterpret_cast<const void*>(&fx_tiled_ring_kernel<G, SPEC, false>);
        }
    }
    return reinterpret_cast<const void*>(&fx_tiled_kernel<G, SPEC>);
}

template <class G>
int tiled_setup(fxc_plan* p) {
    for (int spec = 0; spec < 2; ++spec) {
        int lds = 0;
        const void* fn = spec ? tiled_fn<G, true>(p, &lds) : tiled_fn<G, false>(p, &lds);
        FXC_HIP(p, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        int per_cu = 0;
        FXC_HIP(p, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, G::kThreads, lds));
        if (per_cu < 1) return fail(p, FXC_ERR_HIP, "tiled kernel for nchan=%d does not fit a CU", G::N);
        (spec ? p->tiled_grid_max_f : p->tiled_grid_max) = per_cu * p->cu_count;
    }
    if constexpr (G::N <= 4096) {
        if (p->tiled_ring)   // the uint8-ingest variant shares the F+X variant's launch geometry
            FXC_HIP(p, hipFuncSetAttribute(reinterpret_cast<const void*>(&fx_tiled_ring_kernel<G, false, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, G::kLdsBytesRing));
    }
    return FXC_OK;
}

// SPEC: x = n_streams consecutive streams, nc = pairs of them, raw = spectra [stream][i][k]
template <class G, bool SPEC>
void tiled_launch(fxc_plan* p, const cf* x, int64_t nc, int n_splits, cf* raw, int64_t n_streams, const cf* dc_u8 = nullptr) {
    const int grid = (int)std::min<int64_t>(nc * n_splits, SPEC ? p->tiled_grid_max_f : p->tiled_grid_max);
    if constexpr (G::N <= 4096) {
        if (p->tiled_ring) {
            if constexpr (!SPEC) {
                if (dc_u8) {   // uint8 ingest: x is the byte stream
                    hipLaunchKernelGGL((fx_tiled_ring_kernel<G, false, true>), dim3(grid), dim3(G::kThreads), G::kLdsBytesRing,
                                       p->stream, x, p->num_samp, p->n_pts, nc, n_splits, p->d_win4, p->d_tw0, p->d_tw1,
                                       p->d_tw2, raw, n_streams, dc_u8);
                    return;
                }
            }
            hipLaunchKernelGGL((fx_tiled_ring_kernel<G, SPEC, false>), dim3(grid), dim3(G::kThreads), G::kLdsBytesRing,
                               p->stream, x, p->num_samp, p->n_pts, nc, n_splits, p->d_win4, p->d_tw0, p->d_tw1, p->d_tw2, raw,
                               n_streams, (const cf*)nullptr);
            return;
        }
    }
    hipLaunchKernelGGL((fx_tiled_kernel<G, SPEC>), dim3(grid), dim3(G::kThreads), G::kLdsBytes, p->stream, x, p->num_samp,
                       p->n_pts, nc, n_splits, p->prefilter ? 1 : p->ntaps, p->prefilter ? p->d_ones : p->d_win, p->d_tw0,
                       p->d_tw1, p->d_tw2, raw, n_streams);
}

#define FXC_TILED_DISPATCH(p, CALL)                                                   \
    switch ((p)->nchan) {                                                             \
        case 512: { using G = fxc::tiled::Geo<2, false>; CALL; } break;               \
        case 1024: { using G = fxc::tiled::Geo<4, false>; CALL; } break;              \
        case 2048: { using G = fxc::tiled::Geo<8, false>; CALL; } break;              \
        case 4096: { using G = fxc::tiled::Geo<1, true>; CALL; } break;               \
        case 8192: { using G = fxc::tiled::Geo<2, true>; CALL; } break;               \
        default: return fail(p, FXC_ERR_UNSUPPORTED, "no tiled kernel for nchan=%d", (p)->nchan); \
    }

// this call goes through the tiled kernels.  (Round 1 also sent few-chunk calls on the headline shape here to split a
// chunk's frames over workgroups; the fused kernel's frame ranges do that themselves now, faster: one reference-sized
// call 25 us against 56.)
bool use_tiled(const fxc_plan* p, int64_t) { return p->path == FXC_PATH_TILED; }

bool tiled_nchan(int n) { return n == 512 || n == 1024 || n == 2048 || n == 4096 || n == 8192; }

// frame ranges per chunk so that a launch has at least ~2 work items per resident workgroup
int tiled_splits(const fxc_plan* p, int64_t n_chunks, bool f_only = false) {
    const int64_t cap = f_only ? p->tiled_grid_max_f : p->tiled_grid_max;
    const int64_t want = (2 * cap + n_chunks - 1) / n_chunks;
    const int64_t most = std::max<int64_t>(1, p->n_pts / 8);
    return (int)std::max<int64_t>(1, std::min<int64_t>(std::min(want, most), 256));
}

// streams the pre-filter handles per pass (its output stays within the workspace target)
int64_t prefilter_streams_per_pass(const fxc_plan* p) {
    if (!p->prefilter) return INT64_MAX;
    int64_t n = kWorkspaceTarget / (p->num_samp * (int64_t)sizeof(cf));
    n = std::min<int64_t>(n, 65534) & ~(int64_t)1;      // grid.y carries the stream; whole pairs
    return std::max<int64_t>(2, n);
}

// y = pre-filtered copy of n_streams streams (plan buffer, grown on demand)
int tiled_prefilter(fxc_plan* p, const cf* x, int64_t n_streams, const cf** y_out) {
    const int rg = grow(p, &p->d_pre, &p->pre_bytes, (size_t)n_streams * p->num_samp * sizeof(cf));
    if (rg) return rg;
    cf* y = static_cast<cf*>(p->d_pre);
    const int tp = p->pre_tp;
    // frame splits so that a few-stream call still fills the chip; each split reloads one block of history
    const int64_t blocks = (int64_t)(p->nchan / 256) * n_streams;
    int64_t fs = std::max<int64_t>(1, (2 * (int64_t)p->cu_count + blocks - 1) / blocks);
    fs = std::min<int64_t>(fs, std::max<int64_t>(1, p->n_pts / (4 * tp)));
    const int64_t per = ((p->n_pts + fs - 1) / fs + 2 * tp - 1) / (2 * tp) * (2 * tp);
    const dim3 grid((unsigned)(p->nchan / 256), (unsigned)n_streams, (unsigned)((p->n_pts + per - 1) / per));
    if (tp == 8)
        hipLaunchKernelGGL(pfb_prefilter_kernel<8>, grid, dim3(256), 0, p->stream, x, y, p->d_hpre, p->num_samp, p->nchan, p->n_pts, per);
    else if (tp == 16)
        hipLaunchKernelGGL(pfb_prefilter_kernel<16>, grid, dim3(256), 0, p->stream, x, y, p->d_hpre, p->num_samp, p->nchan, p->n_pts, per);
    else
        hipLaunchKernelGGL(pfb_prefilter_kernel<32>, grid, dim3(256), 0, p->stream, x, y, p->d_hpre, p->num_samp, p->nchan, p->n_pts, per);
    FXC_HIP(p, hipGetLastError());
    *y_out = y;
    return FXC_OK;
}

// raw[split][c][k] (natural bin order) for nc chunks starting at x (nc * 2 <= prefilter_streams_per_pass())
int tiled_raw_sums(fxc_plan* p, const cf* x, int64_t nc, int n_splits, cf* raw, const cf* dc_u8 = nullptr) {
    KernelTimer kt(p);
    if (p->prefilter && !dc_u8) {
        const int rc = tiled_prefilter(p, x, 2 * nc, &x);
        if (rc) return rc;
    }
    FXC_TILED_DISPATCH(p, (tiled_launch<G, false>(p, x, nc, n_splits, raw, 2 * nc, dc_u8)));
    kt.stop();
    FXC_HIP(p, hipGetLastError());
    return FXC_OK;
}

// F-stage only: n_streams consecutive streams -> spec[stream][i][k], pairs of streams per work item
int tiled_channelize(fxc_plan* p, const cf* x, cf* spec, int64_t n_streams) {
    KernelTimer kt(p);
    const int64_t per_pass = prefilter_streams_per_pass(p);
    for (int64_t s0 = 0; s0 < n_streams; s0 += per_pass) {
        const int64_t ns = std::min(per_pass, n_streams - s0);
        const cf* xs = x + s0 * p->num_samp;
        if (p->prefilter) {
            const int rc = tiled_prefilter(p, xs, ns, &xs);
            if (rc) return rc;
        }
        const int64_t pairs = (ns + 1) / 2;
        const int n_splits = tiled_splits(p, pairs, true);
        FXC_TILED_DISPATCH(p, (tiled_launch<G, true>(p, xs, pairs, n_splits, spec + s0 * p->n_pts * p->nchan, ns)));
    }
    kt.stop();
    FXC_HIP(p, hipGetLastError());
    return FXC_OK;
}

// ---- nchan 8192 as two 4096-channel problems (pfb_split8192_kernel) ---------------------------------
int64_t split_chunks_per_pass(const fxc_plan* p, int64_t n_chunks) {
    const int64_t per_chunk = 4 * p->n_pts * 4096 * (int64_t)sizeof(cf) + 2 * 4096 * (int64_t)sizeof(cf);   // y + two raw rows
    return std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(n_chunks, 32767), kWorkspaceTarget / per_chunk));
}

// raw = the fused kernel's rows over 2 nc chunk pairs (+ leading-part rows) for nc chunks of 8192-channel input
int split_raw_sums(fxc_plan* p, const cf* x, int64_t nc, cf* raw) {
    const int64_t half_samp = p->n_pts * 4096;
    int rc = grow(p, &p->d_pre, &p->pre_bytes, (size_t)nc * 4 * half_samp * sizeof(cf));
    if (rc) return rc;
    cf* y = static_cast<cf*>(p->d_pre);
    const int tp = p->pre_tp;
    const int64_t blocks = 16 * 2 * nc;
    int64_t fs = std::max<int64_t>(1, (2 * (int64_t)p->cu_count + blocks - 1) / blocks);
    fs = std::min<int64_t>(fs, std::max<int64_t>(1, p->n_pts / (4 * tp)));
    const int64_t per = ((p->n_pts + fs - 1) / fs + 2 * tp - 1) / (2 * tp) * (2 * tp);
    const dim3 grid(16, (unsigned)(2 * nc), (unsigned)((p->n_pts + per - 1) / per));
    KernelTimer kt(p);
    if (tp == 4)
        hipLaunchKernelGGL(pfb_split8192_kernel<4>, grid, dim3(256), 0, p->stream, x, y, p->d_hpre, p->d_tw8192, p->num_samp, p->n_pts, per);
    else if (tp == 8)
        hipLaunchKernelGGL(pfb_split8192_kernel<8>, grid, dim3(256), 0, p->stream, x, y, p->d_hpre, p->d_tw8192, p->num_samp, p->n_pts, per);
    else
        hipLaunchKernelGGL(pfb_split8192_kernel<16>, grid, dim3(256), 0, p->stream, x, y, p->d_hpre, p->d_tw8192, p->num_samp, p->n_pts, per);
    FXC_HIP(p, hipGetLastError());
    kt.stop();
    return launch_fused(p, y, 2 * nc, raw, false, nullptr, 1, true, half_samp);
}

// nchan == 1 streaming path: raw[block][chunk] partial sums for nc chunks
bool stream_is_t4(const fxc_plan* p) { return p->ntaps <= 4 && (p->num_samp % 2) == 0; }

int64_t stream_blocks(const fxc_plan* p) {
    return stream_is_t4(p) ? kStream4Blocks : (p->num_samp + kStreamBlock - 1) / kStreamBlock;
}

int stream_raw_sums(fxc_plan* p, const cf* x, int64_t nc, cf* raw) {
    const int blocks = (int)stream_blocks(p);
    KernelTimer kt(p);
    if (stream_is_t4(p)) {
        // 16-byte loads need 16-byte aligned streams: x from hipMalloc / torch is, and num_samp is even
        hipLaunchKernelGGL(stream1_t4_kernel, dim3(blocks, (unsigned)nc), dim3(256), 0, p->stream, x, raw, p->num_samp,
                           p->taps.h[0], p->taps.h[1], p->taps.h[2], p->taps.h[3], nc);
    } else {
        const size_t lds = (size_t)2 * (kStreamBlock + p->ntaps - 1) * sizeof(cf);
        hipLaunchKernelGGL(stream1_kernel, dim3(blocks, (unsigned)nc), dim3(256), lds, p->stream, x, raw, p->num_samp,
                           p->ntaps, p->taps, nc);
    }
    kt.stop();
    FXC_HIP(p, hipGetLastError());
    return FXC_OK;
}

// device-resident implementation of fx_accumulate
// dc_u8 != nullptr (fused 2-antenna plans only): x is the uint8 I,Q stream [n_chunks][2][num_samp][2] and dc_u8 its
// per-stream conversion offsets
int fx_accumulate_dev(fxc_plan* p, const cf* x, int64_t n_chunks, const cf* dc_u8 = nullptr) {
    if (n_chunks == 0) return FXC_OK;
    if (p->path == FXC_PATH_STREAM) {
        const int64_t blocks = stream_blocks(p);
        const int64_t cb = std::min<int64_t>(n_chunks, 65535);
        int rc = ensure_ws(p, cb * blocks * (int64_t)sizeof(cf));
        if (rc) return rc;
        cf* raw = reinterpret_cast<cf*>(p->d_ws);
        for (int64_t c0 = 0; c0 < n_chunks; c0 += cb) {
            const int64_t nc = std::min(cb, n_chunks - c0);
            rc = stream_raw_sums(p, x + c0 * 2 * p->num_samp, nc, raw);
            if (rc) return rc;
            hipLaunchKernelGGL(stream1_acc_kernel, dim3(1), dim3(256), 0, p->stream, raw, p->d_acc, nc * blocks);
            FXC_HIP(p, hipGetLastError());
        }
    } else if ((p->path == FXC_PATH_FUSED && (dc_u8 || !use_tiled(p, n_chunks))) || (p->path == FXC_PATH_TILED && p->n_ant > 2)) {
        using namespace fxc::fused;
        const int64_t in_bytes = (int64_t)p->n_ant * p->num_samp * (dc_u8 ? 2 : (int64_t)sizeof(cf));   // per chunk
        int64_t spec_bytes, raw_bytes;
        const int64_t cb = fused_chunks_per_pass(p, n_chunks, &spec_bytes, &raw_bytes);
        const int64_t part_bytes = (int64_t)kFusedReduceSplits * kN * (int64_t)sizeof(cd);
        int rc = ensure_ws(p, spec_bytes + raw_bytes + part_bytes);
        if (rc) return rc;
        cf* spec = reinterpret_cast<cf*>(p->d_ws);
        cf* raw = reinterpret_cast<cf*>(static_cast<char*>(p->d_ws) + spec_bytes);
        cd* part = reinterpret_cast<cd*>(static_cast<char*>(p->d_ws) + spec_bytes + raw_bytes);
        for (int64_t c0 = 0; c0 < n_chunks; c0 += cb) {
            const int64_t nc = std::min(cb, n_chunks - c0);
            const int64_t unit = fused_unit(p);
            rc = fused_raw_sums(p, reinterpret_cast<const cf*>(reinterpret_cast<const char*>(x) + c0 * in_bytes), nc, spec, raw,
                                dc_u8 ? dc_u8 + c0 * 2 : nullptr, unit, false);
            if (rc) return rc;
            if (p->n_ant == 2) {   // one baseline: two-stage reduce over all the raw rows (leading parts included)
                const int64_t n_rows = fused_rows(p, nc, unit, false);
                const int splits = fused_reduce_splits(n_rows);
                hipLaunchKernelGGL(fused_reduce1_kernel, dim3(kN / 256, splits), dim3(256), 0, p->stream, raw, part, kN,
                                   n_rows, splits);
                hipLaunchKernelGGL(fused_reduce2_kernel, dim3(kN / 16), dim3(256), 0, p->stream, part, p->d_acc, kN, splits,
                                   fused_layout(p));
            } else {   // raw rows of `unit` chunks each
                const int64_t per_chunk = (int64_t)p->n_base * p->nchan;
                hipLaunchKernelGGL(acc_add_kernel, dim3(grid_for(per_chunk, 256, p->cu_count)), dim3(256), 0, p->stream,
                                   raw, p->d_acc, p->nchan, p->n_base, (nc + unit - 1) / unit, 1, fused_layout(p));
            }
            FXC_HIP(p, hipGetLastError());
        }
    } else if (p->split8192 && !dc_u8) {
        const int N = p->nchan;
        const int64_t cb = split_chunks_per_pass(p, n_chunks);
        const int64_t row_bytes = (int64_t)fxc::fused::kN * (int64_t)sizeof(cf);
        const int64_t raw_bytes = ((2 * cb + p->fused_grid_max) * row_bytes + 255) / 256 * 256;
        const int64_t part_bytes = (int64_t)kFusedReduceSplits * N * (int64_t)sizeof(cd);
        int rc = ensure_ws(p, raw_bytes + part_bytes);
        if (rc) return rc;
        cf* raw = reinterpret_cast<cf*>(p->d_ws);
        cd* part = reinterpret_cast<cd*>(static_cast<char*>(p->d_ws) + raw_bytes);
        for (int64_t c0 = 0; c0 < n_chunks; c0 += cb) {
            const int64_t nc = std::min(cb, n_chunks - c0);
            rc = split_raw_sums(p, x + c0 * 2 * p->num_samp, nc, raw);
            if (rc) return rc;
            // the nc pairs of 4096-rows are nc rows of 8192 in layout 3; the leading-part rows are added by parity
            const int splits = fused_reduce_splits(nc);
            hipLaunchKernelGGL(fused_reduce1_kernel, dim3(N / 256, splits), dim3(256), 0, p->stream, raw, part, N, nc, splits);
            hipLaunchKernelGGL(fused_reduce2_kernel, dim3(N / 16), dim3(256), 0, p->stream, part, p->d_acc, N, splits, 3);
            hipLaunchKernelGGL(split_lead_acc_kernel, dim3(N / 256), dim3(256), 0, p->stream, raw, p->d_acc, fused_lead(p, 2 * nc));
            FXC_HIP(p, hipGetLastError());
        }
    } else if (use_tiled(p, n_chunks)) {
        const int N = p->nchan;
        const int64_t in_bytes = (int64_t)2 * p->num_samp * (dc_u8 ? 2 : (int64_t)sizeof(cf));   // per chunk
        const int n_splits = tiled_splits(p, n_chunks);
        const int64_t row_bytes = (int64_t)N * (int64_t)sizeof(cf);
        const int64_t cb = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(n_chunks, prefilter_streams_per_pass(p) / 2),
                                                                  kWorkspaceTarget / (row_bytes * n_splits)));
        const int64_t raw_bytes = (cb * n_splits * row_bytes + 255) / 256 * 256;
        const int64_t part_bytes = (int64_t)kFusedReduceSplits * N * (int64_t)sizeof(cd);
        int rc = ensure_ws(p, raw_bytes + part_bytes);
        if (rc) return rc;
        cf* raw = reinterpret_cast<cf*>(p->d_ws);
        cd* part = reinterpret_cast<cd*>(static_cast<char*>(p->d_ws) + raw_bytes);
        const int kb = (N + 255) / 256;
        for (int64_t c0 = 0; c0 < n_chunks; c0 += cb) {
            const int64_t nc = std::min(cb, n_chunks - c0);
            rc = tiled_raw_sums(p, reinterpret_cast<const cf*>(reinterpret_cast<const char*>(x) + c0 * in_bytes), nc, n_splits,
                                raw, dc_u8 ? dc_u8 + c0 * 2 : nullptr);
            if (rc) return rc;
            const int splits = fused_reduce_splits(nc * n_splits);
            hipLaunchKernelGGL(fused_reduce1_kernel, dim3(kb, splits), dim3(256), 0, p->stream, raw, part, N, nc * n_splits,
                               splits);
            hipLaunchKernelGGL(fused_reduce2_kernel, dim3((N + 15) / 16), dim3(256), 0, p->stream, part, p->d_acc, N, splits, 0);
            FXC_HIP(p, hipGetLastError());
        }
    } else {
        const XGeom g = x_geometry(p);
        int64_t spec_bytes, raw_bytes;
        const int64_t cb = generic_chunks_per_pass(p, n_chunks, g, &spec_bytes, &raw_bytes);
        int rc = ensure_ws(p, spec_bytes + raw_bytes);
        if (rc) return rc;
        cf* spec = reinterpret_cast<cf*>(p->d_ws);
        cf* raw = reinterpret_cast<cf*>(static_cast<char*>(p->d_ws) + spec_bytes);
        for (int64_t c0 = 0; c0 < n_chunks; c0 += cb) {
            const int64_t nc = std::min(cb, n_chunks - c0);
            KernelTimer kt(p);
            rc = run_channelize(p, x + c0 * p->n_ant * p->num_samp, spec, nc * p->n_ant);
            if (rc) return rc;
            const int kblocks = (p->nchan + g.kx - 1) / g.kx;
            const int64_t wgs = nc * p->n_base * kblocks * g.n_splits;
            hipLaunchKernelGGL(xmul_kernel, dim3((int)std::min<int64_t>(wgs, (int64_t)p->cu_count * 8)), dim3(256), 0,
                               p->stream, spec, raw, p->n_ant, p->n_base, p->nchan, p->n_pts, g.kx, g.n_splits, nc);
            const int64_t per_chunk = (int64_t)p->n_base * p->nchan;
            hipLaunchKernelGGL(acc_add_kernel, dim3(grid_for(per_chunk, 256, p->cu_count)), dim3(256), 0, p->stream, raw,
                               p->d_acc, p->nchan, p->n_base, nc, g.n_splits, 0);
            kt.stop();
            FXC_HIP(p, hipGetLastError());
        }
    }
    p->spectra_count += (double)n_chunks * (double)p->n_pts;
    return FXC_OK;
}

// device-resident implementation of fx_rows; out = cf[n_chunks][n_base][nchan] or cd[n_chunks][n_base]
int fx_rows_dev(fxc_plan* p, const cf* x, void* out, int64_t n_chunks, int mode, double bandwidth,
                const cf* dc_u8 = nullptr) {
    if (n_chunks == 0) return FXC_OK;
    const float inv_pts = (float)(1.0 / (double)p->n_pts);
    const double cscale = 1.0 / ((double)p->n_pts * (double)p->nchan * bandwidth);
    if (p->path == FXC_PATH_STREAM) {
        const int nb = (int)stream_blocks(p);
        const int64_t cb = std::min<int64_t>(n_chunks, 65535);
        int rc = ensure_ws(p, cb * nb * (int64_t)sizeof(cf));
        if (rc) return rc;
        cf* raw = reinterpret_cast<cf*>(p->d_ws);
        for (int64_t c0 = 0; c0 < n_chunks; c0 += cb) {
            const int64_t nc = std::min(cb, n_chunks - c0);
            rc = stream_raw_sums(p, x + c0 * 2 * p->num_samp, nc, raw);
            if (rc) return rc;
            // raw[block][chunk]: the blocks play the role of the generic path's splits (nchan = n_base = 1)
            if (mode == FXC_MODE_SPECTRUM)
                hipLaunchKernelGGL(rows_spectrum_kernel, dim3(grid_for(nc, 256, p->cu_count)), dim3(256), 0, p->stream, raw,
                                   static_cast<cf*>(out) + c0, p->d_rot, 1, nc, nb, nc, inv_pts, 0, kNoLead);
            else
                hipLaunchKernelGGL(rows_continuum_kernel, dim3((int)std::min<int64_t>(nc, (int64_t)p->cu_count * 8)),
                                   dim3(256), 0, p->stream, raw, static_cast<cd*>(out) + c0, p->d_rot, 1, nc, nb, nc,
                                   cscale, 0, kNoLead);
            FXC_HIP(p, hipGetLastError());
        }
        return FXC_OK;
    }
    if ((p->path == FXC_PATH_FUSED && (dc_u8 || !use_tiled(p, n_chunks))) || (p->path == FXC_PATH_TILED && p->n_ant > 2)) {
        const int64_t in_bytes = (int64_t)p->n_ant * p->num_samp * (dc_u8 ? 2 : (int64_t)sizeof(cf));   // per chunk
        int64_t spec_bytes, raw_bytes;
        const int64_t cb = fused_chunks_per_pass(p, n_chunks, &spec_bytes, &raw_bytes);
        int rc = ensure_ws(p, spec_bytes + raw_bytes);
        if (rc) return rc;
        cf* spec = reinterpret_cast<cf*>(p->d_ws);
        cf* raw = reinterpret_cast<cf*>(static_cast<char*>(p->d_ws) + spec_bytes);
        for (int64_t c0 = 0; c0 < n_chunks; c0 += cb) {
            const int64_t nc = std::min(cb, n_chunks - c0);
            rc = fused_raw_sums(p, reinterpret_cast<const cf*>(reinterpret_cast<const char*>(x) + c0 * in_bytes), nc, spec, raw,
                                dc_u8 ? dc_u8 + c0 * 2 : nullptr);
            if (rc) return rc;
            const int64_t rows = nc * p->n_base;
            const LeadRows lead = p->n_ant == 2 ? fused_lead(p, nc) : kNoLead;
            if (mode == FXC_MODE_SPECTRUM)
                hipLaunchKernelGGL(rows_spectrum_kernel, dim3(grid_for(rows * p->nchan, 256, p->cu_count)), dim3(256), 0,
                                   p->stream, raw, static_cast<cf*>(out) + c0 * p->n_base * p->nchan, p->d_rot, p->nchan,
                                   rows, 1, (int64_t)0, inv_pts, fused_layout(p), lead);
            else
                hipLaunchKernelGGL(rows_continuum_kernel, dim3((int)std::min<int64_t>(rows, (int64_t)p->cu_count * 8)),
                                   dim3(256), 0, p->stream, raw, static_cast<cd*>(out) + c0 * p->n_base, p->d_rot,
                                   p->nchan, rows, 1, (int64_t)0, cscale, fused_layout(p), lead);
            FXC_HIP(p, hipGetLastError());
        }
        return FXC_OK;
    }
    if (p->split8192 && !dc_u8) {
        const int N = p->nchan;
        const int64_t cb = split_chunks_per_pass(p, n_chunks);
        int rc = ensure_ws(p, (2 * cb + p->fused_grid_max) * (int64_t)fxc::fused::kN * (int64_t)sizeof(cf));
        if (rc) return rc;
        cf* raw = reinterpret_cast<cf*>(p->d_ws);
        for (int64_t c0 = 0; c0 < n_chunks; c0 += cb) {
            const int64_t nc = std::min(cb, n_chunks - c0);
            rc = split_raw_sums(p, x + c0 * 2 * p->num_samp, nc, raw);
            if (rc) return rc;
            const LeadRows lead = fused_lead(p, 2 * nc);
            if (mode == FXC_MODE_SPECTRUM)
                hipLaunchKernelGGL(rows_spectrum_kernel, dim3(grid_for(nc * N, 256, p->cu_count)), dim3(256), 0, p->stream,
                                   raw, static_cast<cf*>(out) + c0 * N, p->d_rot, N, nc, 1, (int64_t)0, inv_pts, 3, lead);
            else
                hipLaunchKernelGGL(rows_continuum_kernel, dim3((int)std::min<int64_t>(nc, (int64_t)p->cu_count * 8)),
                                   dim3(256), 0, p->stream, raw, static_cast<cd*>(out) + c0, p->d_rot, N, nc, 1, (int64_t)0,
                                   cscale, 3, lead);
            FXC_HIP(p, hipGetLastError());
        }
        return FXC_OK;
    }
    if (use_tiled(p, n_chunks)) {
        const int N = p->nchan;
        const int64_t in_bytes = (int64_t)2 * p->num_samp * (dc_u8 ? 2 : (int64_t)sizeof(cf));   // per chunk
        const int n_splits = tiled_splits(p, n_chunks);
        const int64_t row_bytes = (int64_t)N * (int64_t)sizeof(cf);
        const int64_t cb = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(n_chunks, prefilter_streams_per_pass(p) / 2),
                                                                  kWorkspaceTarget / (row_bytes * n_splits)));
        int rc = ensure_ws(p, cb * n_splits * row_bytes);
        if (rc) return rc;
        cf* raw = reinterpret_cast<cf*>(p->d_ws);
        for (int64_t c0 = 0; c0 < n_chunks; c0 += cb) {
            const int64_t nc = std::min(cb, n_chunks - c0);
            rc = tiled_raw_sums(p, reinterpret_cast<const cf*>(reinterpret_cast<const char*>(x) + c0 * in_bytes), nc, n_splits,
                                raw, dc_u8 ? dc_u8 + c0 * 2 : nullptr);
            if (rc) return rc;
            if (mode == FXC_MODE_SPECTRUM)
                hipLaunchKernelGGL(rows_spectrum_kernel, dim3(grid_for(nc * N, 256, p->cu_count)), dim3(256), 0, p->stream,
                                   raw, static_cast<cf*>(out) + c0 * N, p->d_rot, N, nc, n_splits, nc * N, inv_pts, 0, kNoLead);
            else
                hipLaunchKernelGGL(rows_continuum_kernel, dim3((int)std::min<int64_t>(nc, (int64_t)p->cu_count * 8)),
                                   dim3(256), 0, p->stream, raw, static_cast<cd*>(out) + c0, p->d_rot, N, nc, n_splits,
                                   nc * N, cscale, 0, kNoLead);
            FXC_HIP(p, hipGetLastError());
        }
        return FXC_OK;
    }
    const XGeom g = x_geometry(p);
    int64_t spec_bytes, raw_bytes;
    const int64_t cb = generic_chunks_per_pass(p, n_chunks, g, &spec_bytes, &raw_bytes);
    int rc = ensure_ws(p, spec_bytes + raw_bytes);
    if (rc) return rc;
    cf* spec = reinterpret_cast<cf*>(p->d_ws);
    cf* raw = reinterpret_cast<cf*>(static_cast<char*>(p->d_ws) + spec_bytes);
    for (int64_t c0 = 0; c0 < n_chunks; c0 += cb) {
        const int64_t nc = std::min(cb, n_chunks - c0);
        KernelTimer kt(p);
        rc = run_channelize(p, x + c0 * p->n_ant * p->num_samp, spec, nc * p->n_ant);
        if (rc) return rc;
        const int kblocks = (p->nchan + g.kx - 1) / g.kx;
        const int64_t wgs = nc * p->n_base * kblocks * g.n_splits;
        hipLaunchKernelGGL(xmul_kernel, dim3((int)std::min<int64_t>(wgs, (int64_t)p->cu_count * 8)), dim3(256), 0,
                           p->stream, spec, raw, p->n_ant, p->n_base, p->nchan, p->n_pts, g.kx, g.n_splits, nc);
        kt.stop();
        const int64_t rows = nc * p->n_base;
        const int64_t split_stride = rows * p->nchan;
        if (mode == FXC_MODE_SPECTRUM)
            hipLaunchKernelGGL(rows_spectrum_kernel, dim3(grid_for(rows * p->nchan, 256, p->cu_count)), dim3(256), 0,
                               p->stream, raw, static_cast<cf*>(out) + c0 * p->n_base * p->nchan, p->d_rot, p->nchan,
                               rows, g.n_splits, split_stride, inv_pts, 0, kNoLead);
        else
            hipLaunchKernelGGL(rows_continuum_kernel, dim3((int)std::min<int64_t>(rows, (int64_t)p->cu_count * 8)),
                               dim3(256), 0, p->stream, raw, static_cast<cd*>(out) + c0 * p->n_base, p->d_rot, p->nchan,
                               rows, g.n_splits, split_stride, cscale, 0, kNoLead);
        FXC_HIP(p, hipGetLastError());
    }
    return FXC_OK;
}

// host-buffer helper: stage in, run, stage out (synchronous)
template <class Fn>
int with_host_staging(fxc_plan* p, const void* x, size_t x_bytes, void* out, size_t out_bytes, Fn fn) {
    // staging buffers live in the plan and only grow: the reference calls once per chunk pair (effex.py:490-494),
    // and a hipMalloc / hipFree pair per call costs more than the copy of one chunk
    const size_t want[2] = {x_bytes ? x_bytes : 1, out_bytes};
    for (int k = 0; k < 2; ++k) {
        const int rg = grow(p, &p->d_stage[k], &p->stage_bytes[k], want[k]);
        if (rg) return rg;
    }
    void* dx = p->d_stage[0];
    void* dout = out_bytes ? p->d_stage[1] : nullptr;
    int rc = FXC_OK;
    hipError_t e = hipMemcpyAsync(dx, x, x_bytes, hipMemcpyHostToDevice, p->stream);
    if (e == hipSuccess) rc = fn(static_cast<const cf*>(dx), dout);
    if (e == hipSuccess && rc == FXC_OK && out_bytes)
        e = hipMemcpyAsync(out, dout, out_bytes, hipMemcpyDeviceToHost, p->stream);
    hipError_t e2 = hipStreamSynchronize(p->stream);
    if (rc != FXC_OK) return rc;
    if (e != hipSuccess) return fail(p, FXC_ERR_HIP, "host staging copy failed: %s", hipGetErrorString(e));
    if (e2 != hipSuccess) return fail(p, FXC_ERR_HIP, "stream sync failed: %s", hipGetErrorString(e2));
    return FXC_OK;
}


}  // namespace

// ------------------------------------------------------------------------------------------
// C ABI (include/fxcorr.h)
// ------------------------------------------------------------------------------------------
extern "C" {

int fxc_version(void) { return FXC_VERSION; }

const char* fxc_status_string(int status) {
    switch (status) {
        case FXC_OK: return "ok";
        case FXC_ERR_ARG: return "invalid argument";
        case FXC_ERR_UNSUPPORTED: return "unsupported configuration";
        case FXC_ERR_HIP: return "HIP runtime error";
        case FXC_ERR_NOMEM: return "out of device memory";
        case FXC_ERR_NODEVICE: return "no HIP device";
        case FXC_ERR_STATE: return "invalid call sequence";
        case FXC_ERR_COMM: return "RCCL unavailable or a collective failed";
        default: return "unknown status";
    }
}

int fxc_device_count(int* count) {
    if (!count) return fail(nullptr, FXC_ERR_ARG, "count is NULL");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *count = n;
    return FXC_OK;
}

const char* fxc_last_error(const fxc_plan* plan) { return plan ? plan->error.c_str() : g_lib_error.c_str(); }

int fxc_plan_destroy(fxc_plan* p) {
    if (!p) return FXC_OK;
    if (p->live_pipes > 0)
        return fail(p, FXC_ERR_STATE, "%d pipe(s) still use this plan: destroy them first", p->live_pipes);
    DeviceGuard device_guard__(p->device);
    (void)hipStreamSynchronize(p->stream);
    for (auto& e : p->kev) {
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    void* bufs[] = {p->d_win, p->d_tw, p->d_rot, p->d_win4, p->d_tw1, p->d_tw2, p->d_tw0, p->d_stamps,
                    p->d_acc, p->d_sums, p->d_out, p->d_ws, p->d_stage[0], p->d_stage[1], p->d_stage[2], p->d_dc, p->d_hpre,
                    p->d_ones, p->d_pre, p->d_tw8192};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    if (p->ev_t0) (void)hipEventDestroy(p->ev_t0);
    if (p->ev_t1) (void)hipEventDestroy(p->ev_t1);
    if (p->ev_order) (void)hipEventDestroy(p->ev_order);
    if (p->h_out) (void)hipHostFree(p->h_out);
    if (p->own_stream && p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
    return FXC_OK;
}

static int plan_build(fxc_plan* p, const double* window, int force_path) {
    hipDeviceProp_t prop;
    FXC_HIP(p, hipGetDeviceProperties(&prop, p->device));
    p->cu_count = prop.multiProcessorCount;
    if (p->own_stream) FXC_HIP(p, hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
    FXC_HIP(p, hipEventCreate(&p->ev_t0));
    FXC_HIP(p, hipEventCreate(&p->ev_t1));
    FXC_HIP(p, hipEventCreateWithFlags(&p->ev_order, hipEventDisableTiming));

    const int N = p->nchan, T = p->ntaps;
    // the fused kernel channelises pairs of antenna streams: 2 antennas (X fused in) or 4 / 6 / 8 (F-only +
    // xengine_kernel); num_samp is bounded by the 32-bit buffer-descriptor range of one stream pair
    const bool fused_shape = ((p->n_ant == 2 || p->n_ant == 4 || p->n_ant == 6 || p->n_ant == 8) && N == fxc::fused::kN &&
                              T == fxc::fused::kT && p->num_samp <= (1ll << 27));
    if (force_path == FXC_PATH_FUSED && !fused_shape)
        return fail(p, FXC_ERR_UNSUPPORTED, "no fused kernel for n_ant=%d nchan=%d ntaps=%d", p->n_ant, N, T);
    const bool stream_shape = (p->n_ant == 2 && N == 1);
    if (force_path == FXC_PATH_STREAM && !stream_shape)
        return fail(p, FXC_ERR_UNSUPPORTED, "the streaming kernel needs n_ant=2, nchan=1");
    // 2 antennas: X fused into the tiled kernel; 3 .. 8: F-only tiled kernel (an odd stream count leaves the last pair
    // half empty) + X-engine
    const bool tiled_shape = (p->n_ant >= 2 && p->n_ant <= 8 && tiled_nchan(N) && p->num_samp <= (1ll << 27));
    if (force_path == FXC_PATH_TILED && !tiled_shape)
        return fail(p, FXC_ERR_UNSUPPORTED, "no tiled kernel for n_ant=%d nchan=%d", p->n_ant, N);
    p->path = FXC_PATH_GENERIC;
    if (tiled_shape && (force_path == -1 || force_path == FXC_PATH_TILED)) p->path = FXC_PATH_TILED;
    if (fused_shape && (force_path == -1 || force_path == FXC_PATH_FUSED)) p->path = FXC_PATH_FUSED;
    if (stream_shape && (force_path == -1 || force_path == FXC_PATH_STREAM)) p->path = FXC_PATH_STREAM;
    for (int t = 0; t < kMaxTaps; ++t) p->taps.h[t] = t < T ? (float)window[t] : 0.f;

    // window: float32 copy of the float64 design (both layouts)
    std::vector<float> wf((size_t)T * N);
    for (size_t n = 0; n < wf.size(); ++n) wf[n] = (float)window[n];
    FXC_HIP(p, hipMalloc(&p->d_win, wf.size() * sizeof(float)));
    FXC_HIP(p, hipMemcpy(p->d_win, wf.data(), wf.size() * sizeof(float), hipMemcpyHostToDevice));

    // generic FFT twiddles exp(+2 pi i j / N): [N/2] for the radix-2 kernel, [N] for the direct DFT
    if (N > 1) {
        const int cnt = p->pow2 ? N / 2 : N;
        std::vector<cf> tw((size_t)cnt);
        for (int jx = 0; jx < cnt; ++jx) {
            const double ph = kTwoPi * (double)jx / (double)N;
            tw[jx] = fxc::mk((float)std::cos(ph), (float)std::sin(ph));
        }
        FXC_HIP(p, hipMalloc(&p->d_tw, tw.size() * sizeof(cf)));
        FXC_HIP(p, hipMemcpy(p->d_tw, tw.data(), tw.size() * sizeof(cf), hipMemcpyHostToDevice));
    }

    std::vector<cd> rot((size_t)N);
    for (auto& r : rot) {
        r.x = 1.0;
        r.y = 0.0;
    }
    FXC_HIP(p, hipMalloc(&p->d_rot, rot.size() * sizeof(cd)));
    FXC_HIP(p, hipMemcpy(p->d_rot, rot.data(), rot.size() * sizeof(cd), hipMemcpyHostToDevice));

    const size_t acc_n = (size_t)p->n_base * N;
    FXC_HIP(p, hipMalloc(&p->d_acc, acc_n * sizeof(cd)));
    FXC_HIP(p, hipMemset(p->d_acc, 0, acc_n * sizeof(cd)));
    FXC_HIP(p, hipMalloc(&p->d_sums, (acc_n + 1) * sizeof(cd)));
    FXC_HIP(p, hipMalloc(&p->d_out, acc_n * sizeof(cd)));
    FXC_HIP(p, hipHostMalloc(reinterpret_cast<void**>(&p->h_out), acc_n * sizeof(cd), hipHostMallocDefault));

    if (p->path == FXC_PATH_FUSED) {
        using namespace fxc::fused;
        std::vector<f4> w4((size_t)kN);
        for (int r = 0; r < 16; ++r)
            for (int jx = 0; jx < 256; ++jx) {
                const int m = jx + 256 * r;
                f4 w;
                w.x = wf[0 * kN + m];
                w.y = wf[1 * kN + m];
                w.z = wf[2 * kN + m];
                w.w = wf[3 * kN + m];
                w4[r * 256 + jx] = w;
            }
        std::vector<cf> tw1((size_t)16 * 256), tw2((size_t)256);
        for (int k1 = 0; k1 < 16; ++k1)
            for (int jx = 0; jx < 256; ++jx) {
                const double ph = kTwoPi * (double)((jx * k1) % kN) / (double)kN;
                tw1[k1 * 256 + jx] = fxc::mk((float)std::cos(ph), (float)std::sin(ph));
            }
        for (int q1 = 0; q1 < 16; ++q1)
            for (int j0 = 0; j0 < 16; ++j0) {
                const double ph = kTwoPi * (double)(j0 * q1) / 256.0;
                tw2[q1 * 16 + j0] = fxc::mk((float)std::cos(ph), (float)std::sin(ph));
            }
        FXC_HIP(p, hipMalloc(&p->d_win4, w4.size() * sizeof(f4)));
        FXC_HIP(p, hipMemcpy(p->d_win4, w4.data(), w4.size() * sizeof(f4), hipMemcpyHostToDevice));
        FXC_HIP(p, hipMalloc(&p->d_tw1, tw1.size() * sizeof(cf)));
        FXC_HIP(p, hipMemcpy(p->d_tw1, tw1.data(), tw1.size() * sizeof(cf), hipMemcpyHostToDevice));
        FXC_HIP(p, hipMalloc(&p->d_tw2, tw2.size() * sizeof(cf)));
        FXC_HIP(p, hipMemcpy(p->d_tw2, tw2.data(), tw2.size() * sizeof(cf), hipMemcpyHostToDevice));
        p->fused_grid_max = p->cu_count;   // one 512-thread workgroup (136 KiB LDS) per CU
        if (const char* e = std::getenv("FXC_FUSED_SEG")) p->fused_seg = std::max<int64_t>(1, std::atoll(e));   // developer knob
        FXC_HIP(p, hipFuncSetAttribute(reinterpret_cast<const void*>(&fx_fused4096_kernel<false, false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
        FXC_HIP(p, hipFuncSetAttribute(reinterpret_cast<const void*>(&fx_fused4096_kernel<true, false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
        FXC_HIP(p, hipFuncSetAttribute(reinterpret_cast<const void*>(&fx_fused4096_kernel<false, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
    }
    p->tiled_f = (tiled_nchan(N) && p->num_samp <= (1ll << 27) && force_path != FXC_PATH_GENERIC);
    if (p->path == FXC_PATH_TILED || p->tiled_f) {
        // pre-stage twiddles wN^((u + P g) k) at [g + G k][u]; stage tables as on the fused path
        const int P = N / 16, R0 = N >= 4096 ? N / 4096 : N / 256, G = 16 / R0;
        std::vector<cf> tw0((size_t)16 * P);
        for (int r = 0; r < 16; ++r)
            for (int u = 0; u < P; ++u) {
                const int g = r % G, k = r / G;
                const double ph = kTwoPi * (double)(((int64_t)(u + P * g) * k) % N) / (double)N;
                tw0[(size_t)r * P + u] = fxc::mk((float)std::cos(ph), (float)std::sin(ph));
            }
        FXC_HIP(p, hipMalloc(&p->d_tw0, tw0.size() * sizeof(cf)));
        FXC_HIP(p, hipMemcpy(p->d_tw0, tw0.data(), tw0.size() * sizeof(cf), hipMemcpyHostToDevice));
        if (!p->d_tw1) {
            std::vector<cf> tw1((size_t)16 * 256), tw2((size_t)256);
            for (int k1 = 0; k1 < 16; ++k1)
                for (int jx = 0; jx < 256; ++jx) {
                    const double ph = kTwoPi * (double)((jx * k1) % 4096) / 4096.0;
                    tw1[k1 * 256 + jx] = fxc::mk((float)std::cos(ph), (float)std::sin(ph));
                }
            for (int q1 = 0; q1 < 16; ++q1)
                for (int j0 = 0; j0 < 16; ++j0) {
                    const double ph = kTwoPi * (double)(j0 * q1) / 256.0;
                    tw2[q1 * 16 + j0] = fxc::mk((float)std::cos(ph), (float)std::sin(ph));
                }
            FXC_HIP(p, hipMalloc(&p->d_tw1, tw1.size() * sizeof(cf)));
            FXC_HIP(p, hipMemcpy(p->d_tw1, tw1.data(), tw1.size() * sizeof(cf), hipMemcpyHostToDevice));
            FXC_HIP(p, hipMalloc(&p->d_tw2, tw2.size() * sizeof(cf)));
            FXC_HIP(p, hipMemcpy(p->d_tw2, tw2.data(), tw2.size() * sizeof(cf), hipMemcpyHostToDevice));
        }
        // more than four taps (or FXC_PREFILTER=1, a developer knob to compare at <= 4): the FIR runs as its own pass
        const char* pre_env = std::getenv("FXC_PREFILTER");
        p->prefilter = (T > 4 || (pre_env && std::atoi(pre_env) == 1));
        if (p->prefilter) {
            p->pre_tp = T <= 8 ? 8 : (T <= 16 ? 16 : 32);
            std::vector<float> hp((size_t)p->pre_tp * N, 0.f), ones((size_t)N, 1.f);
            for (int t = 0; t < T; ++t)
                for (int n = 0; n < N; ++n) hp[(size_t)t * N + n] = wf[(size_t)t * N + (N - 1 - n)];
            FXC_HIP(p, hipMalloc(&p->d_hpre, hp.size() * sizeof(float)));
            FXC_HIP(p, hipMemcpy(p->d_hpre, hp.data(), hp.size() * sizeof(float), hipMemcpyHostToDevice));
            FXC_HIP(p, hipMalloc(&p->d_ones, ones.size() * sizeof(float)));
            FXC_HIP(p, hipMemcpy(p->d_ones, ones.data(), ones.size() * sizeof(float), hipMemcpyHostToDevice));
        }
        // nchan 8192, two antennas, up to 16 taps: the split into two 4096-channel problems (FXC_SPLIT8192=0: off)
        const char* split_env = std::getenv("FXC_SPLIT8192");
        p->split8192 = (N == 8192 && p->n_ant == 2 && T <= 16 && p->path == FXC_PATH_TILED && p->num_samp <= (1ll << 27) &&
                        !(split_env && std::atoi(split_env) == 0));
        if (p->split8192) {
            p->prefilter = false;
            p->pre_tp = T <= 4 ? 4 : (T <= 8 ? 8 : 16);
            std::vector<float> hp((size_t)p->pre_tp * N, 0.f);
            for (int t = 0; t < T; ++t)
                for (int n = 0; n < N; ++n) hp[(size_t)t * N + n] = wf[(size_t)t * N + (N - 1 - n)];
            if (p->d_hpre) (void)hipFree(p->d_hpre);
            FXC_HIP(p, hipMalloc(&p->d_hpre, hp.size() * sizeof(float)));
            FXC_HIP(p, hipMemcpy(p->d_hpre, hp.data(), hp.size() * sizeof(float), hipMemcpyHostToDevice));
            std::vector<cf> tw((size_t)4096);
            for (int n = 0; n < 4096; ++n) {
                const double ph = kTwoPi * (double)(4095 - n) / 8192.0;
                tw[n] = fxc::mk((float)std::cos(ph), (float)std::sin(ph));
            }
            FXC_HIP(p, hipMalloc(&p->d_tw8192, tw.size() * sizeof(cf)));
            FXC_HIP(p, hipMemcpy(p->d_tw8192, tw.data(), tw.size() * sizeof(cf), hipMemcpyHostToDevice));
            std::vector<f4> unit_taps((size_t)fxc::fused::kN);
            for (auto& q : unit_taps) {
                q.x = 1.f;
                q.y = q.z = q.w = 0.f;
            }
            FXC_HIP(p, hipMalloc(&p->d_win4, unit_taps.size() * sizeof(f4)));
            FXC_HIP(p, hipMemcpy(p->d_win4, unit_taps.data(), unit_taps.size() * sizeof(f4), hipMemcpyHostToDevice));
            p->fused_grid_max = p->cu_count;
            FXC_HIP(p, hipFuncSetAttribute(reinterpret_cast<const void*>(&fx_fused4096_kernel<false, false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, fxc::fused::kLdsBytes));
        }
        p->tiled_ring = ((T <= 4 || p->prefilter) && N <= 4096);
        if (p->tiled_ring && (!p->d_win4 || p->prefilter)) {
            std::vector<f4> w4((size_t)N);
            for (int r = 0; r < 16; ++r)
                for (int u = 0; u < P; ++u) {
                    const int m = u + P * r;
                    f4 w;
                    w.x = p->prefilter ? 1.f : wf[m];      // behind the pre-filter: one unit tap
                    w.y = (T > 1 && !p->prefilter) ? wf[(size_t)1 * N + m] : 0.f;
                    w.z = (T > 2 && !p->prefilter) ? wf[(size_t)2 * N + m] : 0.f;
                    w.w = (T > 3 && !p->prefilter) ? wf[(size_t)3 * N + m] : 0.f;
                    w4[(size_t)r * P + u] = w;
                }
            if (p->d_win4) {
                (void)hipFree(p->d_win4);
                p->d_win4 = nullptr;
            }
            FXC_HIP(p, hipMalloc(&p->d_win4, w4.size() * sizeof(f4)));
            FXC_HIP(p, hipMemcpy(p->d_win4, w4.data(), w4.size() * sizeof(f4), hipMemcpyHostToDevice));
        }
        int rc = FXC_OK;
        FXC_TILED_DISPATCH(p, rc = tiled_setup<G>(p));
        if (rc) return rc;
    }
    if (N > 1) {
        const int lds = N * (int)sizeof(cf);
        if (p->pow2)
            FXC_HIP(p, hipFuncSetAttribute(reinterpret_cast<const void*>(&fft_pow2_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        else
            FXC_HIP(p, hipFuncSetAttribute(reinterpret_cast<const void*>(&dft_any_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    }
    return FXC_OK;
}

int fxc_plan_create(fxc_plan** out, int device, int n_ant, int nchan, int ntaps, int64_t num_samp,
                    const double* window, void* stream, int force_path) {
    if (!out) return fail(nullptr, FXC_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (!window) return fail(nullptr, FXC_ERR_ARG, "window is NULL");
    if (n_ant < 1 || n_ant > 64) return fail(nullptr, FXC_ERR_ARG, "n_ant=%d out of range [1,64]", n_ant);
    if (nchan < 1) return fail(nullptr, FXC_ERR_ARG, "nchan=%d must be >= 1", nchan);
    if (ntaps < 1) return fail(nullptr, FXC_ERR_ARG, "ntaps=%d must be >= 1", ntaps);
    if (ntaps > kMaxTaps)
        return fail(nullptr, FXC_ERR_UNSUPPORTED, "Number of taps (%d) must be less than (32).", ntaps);
    if (nchan > kMaxLdsFftN)
        return fail(nullptr, FXC_ERR_UNSUPPORTED, "nchan=%d exceeds the in-LDS FFT limit %d", nchan, kMaxLdsFftN);
    if (num_samp < nchan)
        return fail(nullptr, FXC_ERR_ARG, "num_samp=%lld shorter than one frame of nchan=%d", (long long)num_samp,
                    nchan);
    if (force_path < -1 || force_path > FXC_PATH_TILED) return fail(nullptr, FXC_ERR_ARG, "bad force_path");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(nullptr, FXC_ERR_NODEVICE, "no HIP device available (this library has no CPU backend)");
    if (device < 0 || device >= ndev) return fail(nullptr, FXC_ERR_ARG, "device %d out of range [0,%d)", device, ndev);
    FXC_DEVICE(nullptr, device);

    fxc_plan* p = new (std::nothrow) fxc_plan();
    if (!p) return fail(nullptr, FXC_ERR_NOMEM, "host allocation failed");
    p->device = device;
    p->n_ant = n_ant;
    p->n_base = n_ant * (n_ant - 1) / 2;
    p->nchan = nchan;
    p->ntaps = ntaps;
    p->num_samp = num_samp;
    p->n_pts = num_samp / nchan;
    p->pow2 = (nchan & (nchan - 1)) == 0;
    p->lg2n = 0;
    while ((1 << p->lg2n) < nchan) ++p->lg2n;
    p->own_stream = (stream == FXC_STREAM_OWNED);
    p->stream = p->own_stream ? nullptr : static_cast<hipStream_t>(stream);
    const int rc = plan_build(p, window, force_path);
    if (rc != FXC_OK) {
        g_lib_error = p->error;
        fxc_plan_destroy(p);
        return rc;
    }
    *out = p;
    return FXC_OK;
}

int fxc_plan_get_info(const fxc_plan* p, fxc_info* info) {
    if (!p || !info) return fail(p, FXC_ERR_ARG, "NULL argument");
    std::memset(info, 0, sizeof *info);
    info->n_ant = p->n_ant;
    info->n_baselines = p->n_base;
    info->nchan = p->nchan;
    info->ntaps = p->ntaps;
    info->num_samp = p->num_samp;
    info->n_pts = p->n_pts;
    info->path = p->path;
    if (p->path == FXC_PATH_FUSED) {
        info->grid = p->fused_grid_max;
        info->block = fxc::fused::kThreads;
        info->lds_bytes = fxc::fused::kLdsBytes;
    } else if (p->path == FXC_PATH_TILED) {
        info->grid = p->tiled_grid_max;
        info->block = p->nchan / 8;
        info->lds_bytes = 2 * (p->nchan + p->nchan / 16) * (int)sizeof(cf) + 256 * (int)sizeof(cf) +
                          (p->tiled_ring ? p->nchan * (int)sizeof(f4) : 0);
    } else if (p->path == FXC_PATH_STREAM) {
        info->grid = (int)stream_blocks(p);
        info->block = 256;
        info->lds_bytes = 2 * (kStreamBlock + p->ntaps - 1) * (int)sizeof(cf);
    } else {
        info->grid = p->cu_count * 4;
        info->block = 256;
        info->lds_bytes = p->nchan > 1 ? p->nchan * (int)sizeof(cf) : 0;
    }
    info->device = p->device;
    info->cu_count = p->cu_count;
    info->workspace_bytes = p->ws_bytes;
    return FXC_OK;
}

int fxc_set_rot(fxc_plan* p, const double* rot_re_im) {
    if (!p || !rot_re_im) return fail(p, FXC_ERR_ARG, "NULL argument");
    FXC_DEVICE(p, p->device);
    // ordered after any queued finish kernel that still reads the old table
    FXC_HIP(p, hipStreamSynchronize(p->stream));
    FXC_HIP(p, hipMemcpy(p->d_rot, rot_re_im, (size_t)p->nchan * sizeof(cd), hipMemcpyHostToDevice));
    return FXC_OK;
}

int fxc_set_stream(fxc_plan* p, void* stream) {
    if (!p) return fail(p, FXC_ERR_ARG, "NULL plan");
    if (p->own_stream) return fail(p, FXC_ERR_STATE, "the plan owns its stream (FXC_STREAM_OWNED)");
    hipStream_t next = static_cast<hipStream_t>(stream);
    if (next == p->stream) return FXC_OK;
    FXC_DEVICE(p, p->device);
    // the plan's workspace, accumulator and tables are shared by everything it launches: what is queued on the old
    // stream completes before anything on the new one starts (device-side dependency, no host wait)
    FXC_HIP(p, hipEventRecord(p->ev_order, p->stream));
    FXC_HIP(p, hipStreamWaitEvent(next, p->ev_order, 0));
    p->stream = next;
    return FXC_OK;
}

// ------------------------------------------------------------------------------------------
// multi-GPU reduce: one RCCL sum of the exported accumulators over xGMI, enqueued on the plan's
// stream.  librccl is bound at run time (the copy the process already has -- PyTorch ships one -- else ROCm's), so
// single-GPU users need no RCCL at all.
// ------------------------------------------------------------------------------------------
namespace {

struct RcclApi {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclReduce) reduce = nullptr;
    decltype(&ncclAllReduce) all_reduce = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    std::string error;
};

RcclApi* rccl_api() {
    static RcclApi api;
    static bool tried = false;
    if (tried) return &api;
    tried = true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names)   // a copy that is already mapped wins: one RCCL per process
        if ((api.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    for (size_t k = 0; !api.handle && k < sizeof names / sizeof *names; ++k) api.handle = dlopen(names[k], RTLD_NOW | RTLD_GLOBAL);
    if (!api.handle) {
        const char* why = dlerror();
        api.error = std::string("librccl not found: ") + (why ? why : "dlopen failed");
        return &api;
    }
    bool ok = true;
    auto bind = [&](const char* sym) {
        void* f = dlsym(api.handle, sym);
        if (!f) {
            ok = false;
            api.error = std::string("librccl lacks ") + sym;
        }
        return f;
    };
    api.get_unique_id = reinterpret_cast<decltype(api.get_unique_id)>(bind("ncclGetUniqueId"));
    api.comm_init_rank = reinterpret_cast<decltype(api.comm_init_rank)>(bind("ncclCommInitRank"));
    api.comm_destroy = reinterpret_cast<decltype(api.comm_destroy)>(bind("ncclCommDestroy"));
    api.reduce = reinterpret_cast<decltype(api.reduce)>(bind("ncclReduce"));
    api.all_reduce = reinterpret_cast<decltype(api.all_reduce)>(bind("ncclAllReduce"));
    api.error_string = reinterpret_cast<decltype(api.error_string)>(bind("ncclGetErrorString"));
    if (!ok) {
        api.handle = nullptr;
    }
    return &api;
}

int rccl_fail(const fxc_plan* p, const RcclApi* api, const char* what, ncclResult_t r) {
    return fail(p, FXC_ERR_COMM, "%s failed: %s", what, api->error_string ? api->error_string(r) : "RCCL error");
}

}  // namespace

int fxc_comm_unique_id(void* id_out) {
    if (!id_out) return fail(nullptr, FXC_ERR_ARG, "id_out is NULL");
    static_assert(sizeof(ncclUniqueId) == FXC_COMM_ID_BYTES, "FXC_COMM_ID_BYTES must match ncclUniqueId");
    RcclApi* api = rccl_api();
    if (!api->handle) return fail(nullptr, FXC_ERR_COMM, "%s", api->error.c_str());
    ncclUniqueId id;
    const ncclResult_t r = api->get_unique_id(&id);
    if (r != ncclSuccess) return rccl_fail(nullptr, api, "ncclGetUniqueId", r);
    std::memcpy(id_out, &id, sizeof id);
    return FXC_OK;
}

int fxc_comm_create(void** rccl_comm_out, int device, int rank, int world_size, const void* id) {
    if (!rccl_comm_out || !id) return fail(nullptr, FXC_ERR_ARG, "NULL argument");
    *rccl_comm_out = nullptr;
    if (world_size < 1 || rank < 0 || rank >= world_size) return fail(nullptr, FXC_ERR_ARG, "rank %d outside world of %d", rank, world_size);
    RcclApi* api = rccl_api();
    if (!api->handle) return fail(nullptr, FXC_ERR_COMM, "%s", api->error.c_str());
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(nullptr, FXC_ERR_NODEVICE, "no HIP device available");
    if (device < 0 || device >= ndev) return fail(nullptr, FXC_ERR_ARG, "device %d out of range [0,%d)", device, ndev);
    FXC_DEVICE(nullptr, device);
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof uid);
    ncclComm_t comm = nullptr;
    const ncclResult_t r = api->comm_init_rank(&comm, world_size, uid, rank);
    if (r != ncclSuccess) return rccl_fail(nullptr, api, "ncclCommInitRank", r);
    *rccl_comm_out = comm;
    return FXC_OK;
}

int fxc_comm_destroy(void* rccl_comm) {
    if (!rccl_comm) return FXC_OK;
    RcclApi* api = rccl_api();
    if (!api->handle) return fail(nullptr, FXC_ERR_COMM, "%s", api->error.c_str());
    const ncclResult_t r = api->comm_destroy(static_cast<ncclComm_t>(rccl_comm));
    if (r != ncclSuccess) return rccl_fail(nullptr, api, "ncclCommDestroy", r);
    return FXC_OK;
}

int fxc_reduce(fxc_plan* p, void* rccl_comm, int root) {
    if (!p) return fail(p, FXC_ERR_ARG, "NULL plan");
    FXC_DEVICE(p, p->device);
    int rc = fxc_acc_export(p, p->d_sums);
    if (rc) return rc;
    if (!rccl_comm) return FXC_OK;          // single rank: the exported sums are the reduced sums
    RcclApi* api = rccl_api();
    if (!api->handle) return fail(p, FXC_ERR_COMM, "%s", api->error.c_str());
    // raw float64 sums + the spectra count, in place, ordered on the plan's stream behind the export
    const size_t count = 2 * ((size_t)p->n_base * p->nchan + 1);
    ncclComm_t comm = static_cast<ncclComm_t>(rccl_comm);
    const ncclResult_t r = root < 0 ? api->all_reduce(p->d_sums, p->d_sums, count, ncclFloat64, ncclSum, comm, p->stream)
                                    : api->reduce(p->d_sums, p->d_sums, count, ncclFloat64, ncclSum, root, comm, p->stream);
    if (r != ncclSuccess) return rccl_fail(p, api, root < 0 ? "ncclAllReduce" : "ncclReduce", r);
    return FXC_OK;
}

int fxc_sync(fxc_plan* p) {
    if (!p) return fail(p, FXC_ERR_ARG, "NULL plan");
    FXC_DEVICE(p, p->device);
    FXC_HIP(p, hipStreamSynchronize(p->stream));
    return FXC_OK;
}

int fxc_channelize(fxc_plan* p, const void* x, void* out, int64_t n_streams, int mem_kind) {
    if (!p) return fail(p, FXC_ERR_ARG, "NULL plan");
    if (n_streams < 0) return fail(p, FXC_ERR_ARG, "n_streams < 0");
    if (n_streams == 0) return FXC_OK;
    if (!x || !out) return fail(p, FXC_ERR_ARG, "NULL buffer");
    FXC_DEVICE(p, p->device);
    if (mem_kind == FXC_MEM_DEVICE)
        return run_channelize(p, static_cast<const cf*>(x), static_cast<cf*>(out), n_streams);
    if (mem_kind != FXC_MEM_HOST) return fail(p, FXC_ERR_ARG, "bad mem_kind %d", mem_kind);
    const size_t xb = (size_t)n_streams * p->num_samp * sizeof(cf);
    const size_t ob = (size_t)n_streams * p->n_pts * p->nchan * sizeof(cf);
    return with_host_staging(p, x, xb, out, ob, [&](const cf* dx, void* dout) {
        return run_channelize(p, dx, static_cast<cf*>(dout), n_streams);
    });
}

int fxc_fx_accumulate(fxc_plan* p, const void* x, int64_t n_chunks, int mem_kind) {
    if (!p) return fail(p, FXC_ERR_ARG, "NULL plan");
    if (n_chunks < 0) return fail(p, FXC_ERR_ARG, "n_chunks < 0");
    if (p->n_ant < 2) return fail(p, FXC_ERR_ARG, "cross-correlation needs n_ant >= 2");
    if (n_chunks == 0) return FXC_OK;
    if (!x) return fail(p, FXC_ERR_ARG, "NULL buffer");
    FXC_DEVICE(p, p->device);
    if (mem_kind == FXC_MEM_DEVICE) return fx_accumulate_dev(p, static_cast<const cf*>(x), n_chunks);
    if (mem_kind != FXC_MEM_HOST) return fail(p, FXC_ERR_ARG, "bad mem_kind %d", mem_kind);
    const size_t xb = (size_t)n_chunks * p->n_ant * p->num_samp * sizeof(cf);
    return with_host_staging(p, x, xb, nullptr, 0,
                             [&](const cf* dx, void*) { return fx_accumulate_dev(p, dx, n_chunks); });
}

int fxc_fx_rows(fxc_plan* p, const void* x, void* out, int64_t n_chunks, int mem_kind, int mode, double bandwidth) {
    if (!p) return fail(p, FXC_ERR_ARG, "NULL plan");
    if (n_chunks < 0) return fail(p, FXC_ERR_ARG, "n_chunks < 0");
    if (p->n_ant < 2) return fail(p, FXC_ERR_ARG, "cross-correlation needs n_ant >= 2");
    if (mode != FXC_MODE_SPECTRUM && mode != FXC_MODE_CONTINUUM) return fail(p, FXC_ERR_ARG, "bad mode %d", mode);
    if (mode == FXC_MODE_CONTINUUM && !(bandwidth > 0.0)) return fail(p, FXC_ERR_ARG, "bandwidth must be > 0");
    if (n_chunks == 0) return FXC_OK;
    if (!x || !out) return fail(p, FXC_ERR_ARG, "NULL buffer");
    FXC_DEVICE(p, p->device);
    if (mem_kind == FXC_MEM_DEVICE) return fx_rows_dev(p, static_cast<const cf*>(x), out, n_chunks, mode, bandwidth);
    if (mem_kind != FXC_MEM_HOST) return fail(p, FXC_ERR_ARG, "bad mem_kind %d", mem_kind);
    const size_t xb = (size_t)n_chunks * p->n_ant * p->num_samp * sizeof(cf);
    const size_t ob = mode == FXC_MODE_SPECTRUM ? (size_t)n_chunks * p->n_base * p->nchan * sizeof(cf)
                                                : (size_t)n_chunks * p->n_base * sizeof(cd);
    return with_host_staging(p, x, xb, out, ob, [&](const cf* dx, void* dout) {
        return fx_rows_dev(p, dx, dout, n_chunks, mode, bandwidth);
    });
}

int fxc_acc_reset(fxc_plan* p) {
    if (!p) return fail(p, FXC_ERR_ARG, "NULL plan");
    FXC_DEVICE(p, p->device);
    FXC_HIP(p, hipMemsetAsync(p->d_acc, 0, (size_t)p->n_base * p->nchan * sizeof(cd), p->stream));
    p->spectra_count = 0.0;
    return FXC_OK;
}

int fxc_acc_export(fxc_plan* p, void* sums_dev) {
    if (!p || !sums_dev) return fail(p, FXC_ERR_ARG, "NULL argument");
    FXC_DEVICE(p, p->device);
    const int64_t n = (int64_t)p->n_base * p->nchan;
    hipLaunchKernelGGL(export_kernel, dim3(grid_for(n + 1, 256, p->cu_count)), dim3(256), 0, p->stream, p->d_acc,
                       static_cast<cd*>(sums_dev), n, p->spectra_count);
    FXC_HIP(p, hipGetLastError());
    return FXC_OK;
}

int fxc_finalize_sums(fxc_plan* p, const void* sums_dev, void* out_host, int mode, double bandwidth) {
    if (!p || !out_host) return fail(p, FXC_ERR_ARG, "NULL argument");
    if (!sums_dev) sums_dev = p->d_sums;     // what fxc_reduce left in the plan
    if (mode != FXC_MODE_SPECTRUM && mode != FXC_MODE_CONTINUUM) return fail(p, FXC_ERR_ARG, "bad mode %d", mode);
    if (mode == FXC_MODE_CONTINUUM && !(bandwidth > 0.0)) return fail(p, FXC_ERR_ARG, "bandwidth must be > 0");
    FXC_DEVICE(p, p->device);
    const cd* sums = static_cast<const cd*>(sums_dev);
    size_t out_bytes;
    if (mode == FXC_MODE_SPECTRUM) {
        const int64_t n = (int64_t)p->n_base * p->nchan;
        hipLaunchKernelGGL(finalize_spectrum_kernel, dim3(grid_for(n, 256, p->cu_count)), dim3(256), 0, p->stream, sums,
                           p->d_out, p->d_rot, p->nchan, p->n_base);
        out_bytes = (size_t)n * sizeof(cd);
    } else {
        hipLaunchKernelGGL(finalize_continuum_kernel, dim3(p->n_base), dim3(256), 0, p->stream, sums, p->d_out,
                           p->d_rot, p->nchan, p->n_base, 1.0 / bandwidth);
        out_bytes = (size_t)p->n_base * sizeof(cd);
    }
    FXC_HIP(p, hipGetLastError());
    FXC_HIP(p, hipMemcpyAsync(p->h_out, p->d_out, out_bytes, hipMemcpyDeviceToHost, p->stream));
    FXC_HIP(p, hipStreamSynchronize(p->stream));
    std::memcpy(out_host, p->h_out, out_bytes);
    return FXC_OK;
}

int fxc_finalize(fxc_plan* p, void* out_host, int mode, double bandwidth, int reset) {
    if (!p || !out_host) return fail(p, FXC_ERR_ARG, "NULL argument");
    if (!(p->spectra_count > 0.0)) return fail(p, FXC_ERR_STATE, "nothing accumulated");
    int rc = fxc_acc_export(p, p->d_sums);
    if (rc) return rc;
    rc = fxc_finalize_sums(p, p->d_sums, out_host, mode, bandwidth);
    if (rc) return rc;
    if (reset) return fxc_acc_reset(p);
    return FXC_OK;
}

static int conditioning_common(fxc_plan* p, int64_t n_streams, const void* x, void* out) {
    if (!p) return fail(p, FXC_ERR_ARG, "NULL plan");
    if (n_streams < 0) return fail(p, FXC_ERR_ARG, "n_streams < 0");
    if (n_streams > 65535) return fail(p, FXC_ERR_ARG, "at most 65535 streams per call");
    if (n_streams > 0 && (!x || !out)) return fail(p, FXC_ERR_ARG, "NULL buffer");
    return FXC_OK;
}

int fxc_remove_dc(fxc_plan* p, const void* x_dev, void* out_dev, int64_t n_streams) {
    int rc = conditioning_common(p, n_streams, x_dev, out_dev);
    if (rc || n_streams == 0) return rc;
    FXC_DEVICE(p, p->device);
    const int n_slices = 32;
    rc = ensure_ws(p, n_streams * n_slices * 2 * (int64_t)sizeof(double));
    if (rc) return rc;
    double* part = static_cast<double*>(p->d_ws);
    const int64_t total = n_streams * p->num_samp;
    hipLaunchKernelGGL(dc_sum_c64_kernel, dim3(n_slices, (unsigned)n_streams), dim3(256), 0, p->stream,
                       static_cast<const cf*>(x_dev), part, p->num_samp, n_slices);
    hipLaunchKernelGGL(dc_apply_c64_kernel, dim3(grid_for(total, 256, p->cu_count)), dim3(256), 0, p->stream,
                       static_cast<const cf*>(x_dev), static_cast<cf*>(out_dev), part, p->num_samp, n_slices, total);
    FXC_HIP(p, hipGetLastError());
    return FXC_OK;
}

int fxc_convert_u8(fxc_plan* p, const void* iq_u8_dev, void* out_dev, int64_t n_streams, int remove_dc) {
    int rc = conditioning_common(p, n_streams, iq_u8_dev, out_dev);
    if (rc || n_streams == 0) return rc;
    FXC_DEVICE(p, p->device);
    const int n_slices = 32;
    rc = ensure_ws(p, n_streams * n_slices * 2 * (int64_t)sizeof(double));
    if (rc) return rc;
    double* part = static_cast<double*>(p->d_ws);
    const int64_t total = n_streams * p->num_samp;
    if (remove_dc)
        hipLaunchKernelGGL(dc_sum_u8_kernel, dim3(n_slices, (unsigned)n_streams), dim3(256), 0, p->stream,
                           static_cast<const unsigned char*>(iq_u8_dev), part, p->num_samp, n_slices);
    hipLaunchKernelGGL(convert_u8_kernel, dim3(grid_for(total, 256, p->cu_count)), dim3(256), 0, p->stream,
                       static_cast<const unsigned char*>(iq_u8_dev), static_cast<cf*>(out_dev), part, p->num_samp,
                       n_slices, total, remove_dc ? 1 : 0);
    FXC_HIP(p, hipGetLastError());
    return FXC_OK;
}

namespace {

// uint8 I,Q in: fused plans (2 antennas, nchan 4096, ntaps 4) read the bytes in the F+X kernel itself; every other plan
// converts into a complex64 staging buffer first.  rows: fxc_fx_rows semantics (out != nullptr) or accumulate.
int fx_u8_dev(fxc_plan* p, const unsigned char* x8, void* out, int64_t n_chunks, int mode, double bandwidth, int remove_dc,
              bool rows) {
    constexpr int kSlices = 32;
    const size_t row_elems = mode == FXC_MODE_SPECTRUM ? (size_t)p->n_base * p->nchan * sizeof(cf) : (size_t)p->n_base * sizeof(cd);
    // chunks per pass: dc_sum_u8_kernel carries the stream index in grid.y (<= 65535 streams), and plans without the
    // fused ingest convert a pass into a complex64 staging buffer that stays within the workspace target
    const bool fused_in = p->n_ant == 2 && (p->path == FXC_PATH_FUSED || (p->path == FXC_PATH_TILED && p->tiled_ring && !p->prefilter));
    int64_t per_pass = std::min<int64_t>(16384, 65535 / p->n_ant);
    if (!fused_in) per_pass = std::min<int64_t>(per_pass, kWorkspaceTarget / ((int64_t)p->n_ant * p->num_samp * (int64_t)sizeof(cf)));
    per_pass = std::max<int64_t>(1, per_pass);
    for (int64_t c0 = 0; c0 < n_chunks; c0 += per_pass) {
        const int64_t nc = std::min<int64_t>(per_pass, n_chunks - c0);
        const int64_t n_streams = nc * p->n_ant;
        const unsigned char* xb = x8 + c0 * p->n_ant * p->num_samp * 2;
        void* ob = rows ? static_cast<char*>(out) + (size_t)c0 * row_elems : nullptr;
        const size_t part_bytes = (size_t)n_streams * kSlices * 2 * sizeof(double);
        int rc = grow(p, &p->d_dc, &p->dc_bytes, part_bytes + (size_t)n_streams * sizeof(cf));
        if (rc) return rc;
        double* part = static_cast<double*>(p->d_dc);
        cf* dc = reinterpret_cast<cf*>(static_cast<char*>(p->d_dc) + part_bytes);
        const bool fused_ingest = fused_in;
        if (remove_dc && fused_ingest)
            hipLaunchKernelGGL(dc_sum_u8_stream_kernel, dim3((unsigned)std::min<int64_t>(n_streams, (int64_t)p->cu_count * 16)),
                               dim3(256), 0, p->stream, xb, part, p->num_samp, n_streams);
        else if (remove_dc)
            hipLaunchKernelGGL(dc_sum_u8_kernel, dim3(kSlices, (unsigned)n_streams), dim3(256), 0, p->stream, xb, part,
                               p->num_samp, kSlices);
        if (fused_ingest) {
            hipLaunchKernelGGL(dc_offsets_u8_kernel, dim3((unsigned)((n_streams + 255) / 256)), dim3(256), 0, p->stream, part,
                               dc, n_streams, 1, p->num_samp, remove_dc ? 1 : 0);
            FXC_HIP(p, hipGetLastError());
            rc = rows ? fx_rows_dev(p, reinterpret_cast<const cf*>(xb), ob, nc, mode, bandwidth, dc)
                      : fx_accumulate_dev(p, reinterpret_cast<const cf*>(xb), nc, dc);
        } else {
            const int64_t total = n_streams * p->num_samp;
            rc = grow(p, &p->d_stage[2], &p->stage_bytes[2], (size_t)total * sizeof(cf));
            if (rc) return rc;
            cf* xc = static_cast<cf*>(p->d_stage[2]);
            hipLaunchKernelGGL(convert_u8_kernel, dim3(grid_for(total, 256, p->cu_count)), dim3(256), 0, p->stream, xb, xc,
                               part, p->num_samp, kSlices, total, remove_dc ? 1 : 0);
            FXC_HIP(p, hipGetLastError());
            rc = rows ? fx_rows_dev(p, xc, ob, nc, mode, bandwidth) : fx_accumulate_dev(p, xc, nc);
        }
        if (rc) return rc;
    }
    return FXC_OK;
}

int fx_u8_entry(fxc_plan* p, const void* iq_u8, void* out, int64_t n_chunks, int mem_kind, int mode, double bandwidth,
                int remove_dc, bool rows) {
    if (!p) return fail(p, FXC_ERR_ARG, "NULL plan");
    if (p->n_ant < 2) return fail(p, FXC_ERR_ARG, "cross-correlation needs n_ant >= 2");
    if (n_chunks < 0) return fail(p, FXC_ERR_ARG, "n_chunks < 0");
    if (rows && mode != FXC_MODE_SPECTRUM && mode != FXC_MODE_CONTINUUM) return fail(p, FXC_ERR_ARG, "bad mode %d", mode);
    if (rows && mode == FXC_MODE_CONTINUUM && !(bandwidth > 0.0)) return fail(p, FXC_ERR_ARG, "bandwidth must be > 0");
    if (n_chunks == 0) return FXC_OK;
    if (!iq_u8 || (rows && !out)) return fail(p, FXC_ERR_ARG, "NULL buffer");
    FXC_DEVICE(p, p->device);
    if (mem_kind == FXC_MEM_DEVICE)
        return fx_u8_dev(p, static_cast<const unsigned char*>(iq_u8), out, n_chunks, mode, bandwidth, remove_dc, rows);
    if (mem_kind != FXC_MEM_HOST) return fail(p, FXC_ERR_ARG, "bad mem_kind %d", mem_kind);
    const size_t xb = (size_t)n_chunks * p->n_ant * p->num_samp * 2;
    const size_t ob = !rows ? 0
                            : (mode == FXC_MODE_SPECTRUM ? (size_t)n_chunks * p->n_base * p->nchan * sizeof(cf)
                                                         : (size_t)n_chunks * p->n_base * sizeof(cd));
    return with_host_staging(p, iq_u8, xb, out, ob, [&](const cf* dx, void* dout) {
        return fx_u8_dev(p, reinterpret_cast<const unsigned char*>(dx), dout, n_chunks, mode, bandwidth, remove_dc, rows);
    });
}

}  // namespace

int fxc_fx_rows_u8(fxc_plan* p, const void* iq_u8, void* out, int64_t n_chunks, int mem_kind, int mode, double bandwidth,
                   int remove_dc) {
    return fx_u8_entry(p, iq_u8, out, n_chunks, mem_kind, mode, bandwidth, remove_dc, true);
}

int fxc_fx_accumulate_u8(fxc_plan* p, const void* iq_u8, int64_t n_chunks, int mem_kind, int remove_dc) {
    return fx_u8_entry(p, iq_u8, nullptr, n_chunks, mem_kind, FXC_MODE_SPECTRUM, 1.0, remove_dc, false);
}

int fxc_estimate_delay(fxc_plan* p, const void* iq0, const void* iq1, int64_t n, int mem_kind, double rate,
                       double* delay_s) {
    if (!p || !iq0 || !iq1 || !delay_s) return fail(p, FXC_ERR_ARG, "NULL argument");
    if (n < 2 || n > (1ll << 28)) return fail(p, FXC_ERR_ARG, "n=%lld out of range", (long long)n);
    if (!(rate > 0.0)) return fail(p, FXC_ERR_ARG, "rate must be > 0");
    if (mem_kind != FXC_MEM_HOST && mem_kind != FXC_MEM_DEVICE) return fail(p, FXC_ERR_ARG, "bad mem_kind %d", mem_kind);
    FXC_DEVICE(p, p->device);
    int lg = 1;
    while ((1ll << lg) < 2 * n) ++lg;
    const int64_t len = 1ll << lg;
    // workspace: 4 transform buffers + staging for host inputs + result words
    const int64_t buf_bytes = len * (int64_t)sizeof(cf);
    const int64_t stage_bytes = mem_kind == FXC_MEM_HOST ? 2 * n * (int64_t)sizeof(cf) : 0;
    int rc = ensure_ws(p, 4 * buf_bytes + stage_bytes + 256);
    if (rc) return rc;
    char* ws = static_cast<char*>(p->d_ws);
    cf* a[2] = {reinterpret_cast<cf*>(ws), reinterpret_cast<cf*>(ws + buf_bytes)};
    cf* b[2] = {reinterpret_cast<cf*>(ws + 2 * buf_bytes), reinterpret_cast<cf*>(ws + 3 * buf_bytes)};
    const cf *x0 = static_cast<const cf*>(iq0), *x1 = static_cast<const cf*>(iq1);
    if (mem_kind == FXC_MEM_HOST) {
        cf* st = reinterpret_cast<cf*>(ws + 4 * buf_bytes);
        FXC_HIP(p, hipMemcpyAsync(st, iq0, (size_t)n * sizeof(cf), hipMemcpyHostToDevice, p->stream));
        FXC_HIP(p, hipMemcpyAsync(st + n, iq1, (size_t)n * sizeof(cf), hipMemcpyHostToDevice, p->stream));
        x0 = st;
        x1 = st + n;
    }
    unsigned long long* best = reinterpret_cast<unsigned long long*>(ws + 4 * buf_bytes + stage_bytes);
    cf* out3 = reinterpret_cast<cf*>(ws + 4 * buf_bytes + stage_bytes + 16);
    const int g_len = grid_for(len, 256, p->cu_count), g_half = grid_for(len / 2, 256, p->cu_count);
    hipLaunchKernelGGL(delay_pad_kernel, dim3(g_len), dim3(256), 0, p->stream, x0, a[0], n, len);
    hipLaunchKernelGGL(delay_pad_kernel, dim3(g_len), dim3(256), 0, p->stream, x1, b[0], n, len);
    int cur = 0;
    for (int s = 0; s < lg; ++s, cur ^= 1) {   // forward transforms, kernel exp(-2 pi i ...) like cp.fft.fft
        hipLaunchKernelGGL(stockham_stage_kernel, dim3(g_half), dim3(256), 0, p->stream, a[cur], a[cur ^ 1], len / 2,
                           1ll << s, -1.0);
        hipLaunchKernelGGL(stockham_stage_kernel, dim3(g_half), dim3(256), 0, p->stream, b[cur], b[cur ^ 1], len / 2,
                           1ll << s, -1.0);
    }
    hipLaunchKernelGGL(mul_conj_kernel, dim3(g_len), dim3(256), 0, p->stream, a[cur], b[cur], len);   // f0 * conj(f1)
    for (int s = 0; s < lg; ++s, cur ^= 1)     // inverse transform (un-normalised: the peak fit is scale free)
        hipLaunchKernelGGL(stockham_stage_kernel, dim3(g_half), dim3(256), 0, p->stream, a[cur], a[cur ^ 1], len / 2,
                           1ll << s, 1.0);
    FXC_HIP(p, hipMemsetAsync(best, 0, 8, p->stream));
    hipLaunchKernelGGL(delay_argmax_kernel, dim3(grid_for(2 * n, 256, p->cu_count)), dim3(256), 0, p->stream, a[cur], best,
                       n, len);
    hipLaunchKernelGGL(delay_fetch_kernel, dim3(1), dim3(64), 0, p->stream, a[cur], best, out3, n, len);
    FXC_HIP(p, hipGetLastError());
    unsigned long long h_best = 0;
    cf h3[3];
    FXC_HIP(p, hipMemcpyAsync(&h_best, best, 8, hipMemcpyDeviceToHost, p->stream));
    FXC_HIP(p, hipMemcpyAsync(h3, out3, sizeof h3, hipMemcpyDeviceToHost, p->stream));
    FXC_HIP(p, hipStreamSynchronize(p->stream));
    const int64_t imax = (int64_t)(0xFFFFFFFFull - (h_best & 0xFFFFFFFFull));
    if (imax + 1 >= 2 * n)
        return fail(p, FXC_ERR_STATE, "correlation peak at the last lag (the reference raises IndexError here)");
    // effex.py:619-625
    const double xprev = std::hypot((double)h3[0].x, (double)h3[0].y);
    const double xbest = std::hypot((double)h3[1].x, (double)h3[1].y);
    const double xnext = std::hypot((double)h3[2].x, (double)h3[2].y);
    const double delta = 0.5 * (std::log(xprev) - std::log(xnext)) /
                         (std::log(xprev) - 2.0 * std::log(xbest) + std::log(xnext));
    *delay_s = ((double)n - ((double)imax + delta)) / rate;
    return FXC_OK;
}

int fxc_pipe_destroy(fxc_pipe* q) {
    if (!q) return FXC_OK;
    DeviceGuard device_guard__(q->plan->device);
    if (q->counted) q->plan->live_pipes -= 1;
    (void)hipStreamSynchronize(q->plan->stream);
    if (q->s_in) (void)hipStreamSynchronize(q->s_in);
    if (q->s_out) (void)hipStreamSynchronize(q->s_out);
    for (auto& sl : q->slots) {
        if (sl.h_in) (void)hipHostFree(sl.h_in);
        if (sl.h_out) (void)hipHostFree(sl.h_out);
        if (sl.d_in) (void)hipFree(sl.d_in);
        if (sl.d_out) (void)hipFree(sl.d_out);
        if (sl.ev_in) (void)hipEventDestroy(sl.ev_in);
        if (sl.ev_compute) (void)hipEventDestroy(sl.ev_compute);
        if (sl.ev_out) (void)hipEventDestroy(sl.ev_out);
    }
    if (q->s_in) (void)hipStreamDestroy(q->s_in);
    if (q->s_out) (void)hipStreamDestroy(q->s_out);
    delete q;
    return FXC_OK;
}

static int pipe_create(fxc_pipe** out, fxc_plan* p, int64_t chunks_per_batch, int depth, int mode, double bandwidth, bool u8,
                       int remove_dc);

int fxc_pipe_create(fxc_pipe** out, fxc_plan* p, int64_t chunks_per_batch, int depth, int mode, double bandwidth) {
    return pipe_create(out, p, chunks_per_batch, depth, mode, bandwidth, false, 0);
}

int fxc_pipe_create_u8(fxc_pipe** out, fxc_plan* p, int64_t chunks_per_batch, int depth, int mode, double bandwidth,
                       int remove_dc) {
    return pipe_create(out, p, chunks_per_batch, depth, mode, bandwidth, true, remove_dc);
}

static int pipe_create(fxc_pipe** out, fxc_plan* p, int64_t chunks_per_batch, int depth, int mode, double bandwidth, bool u8,
                       int remove_dc) {
    if (!out || !p) return fail(p, FXC_ERR_ARG, "NULL argument");
    *out = nullptr;
    if (chunks_per_batch < 1 || depth < 1 || depth > 16) return fail(p, FXC_ERR_ARG, "bad batch size or depth");
    if (p->n_ant < 2) return fail(p, FXC_ERR_ARG, "cross-correlation needs n_ant >= 2");
    if (mode != FXC_MODE_SPECTRUM && mode != FXC_MODE_CONTINUUM) return fail(p, FXC_ERR_ARG, "bad mode %d", mode);
    if (mode == FXC_MODE_CONTINUUM && !(bandwidth > 0.0)) return fail(p, FXC_ERR_ARG, "bandwidth must be > 0");
    FXC_DEVICE(p, p->device);
    fxc_pipe* q = new (std::nothrow) fxc_pipe();
    if (!q) return fail(p, FXC_ERR_NOMEM, "host allocation failed");
    q->plan = p;
    q->chunks = chunks_per_batch;
    q->depth = depth;
    q->mode = mode;
    q->bandwidth = bandwidth;
    q->u8 = u8;
    q->remove_dc = remove_dc;
    q->in_bytes = (size_t)chunks_per_batch * p->n_ant * p->num_samp * (u8 ? 2 : sizeof(cf));
    q->out_bytes = mode == FXC_MODE_SPECTRUM ? (size_t)chunks_per_batch * p->n_base * p->nchan * sizeof(cf)
                                             : (size_t)chunks_per_batch * p->n_base * sizeof(cd);
    q->slots.resize((size_t)depth);
    hipError_t e = hipStreamCreateWithFlags(&q->s_in, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&q->s_out, hipStreamNonBlocking);
    for (auto& sl : q->slots) {
        if (e == hipSuccess) e = hipHostMalloc(&sl.h_in, q->in_bytes, hipHostMallocDefault);
        if (e == hipSuccess) e = hipHostMalloc(&sl.h_out, q->out_bytes, hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc(&sl.d_in, q->in_bytes);
        if (e == hipSuccess) e = hipMalloc(&sl.d_out, q->out_bytes);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.ev_in, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.ev_compute, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.ev_out, hipEventDisableTiming);
    }
    if (e != hipSuccess) {
        fxc_pipe_destroy(q);
        return fail(p, e == hipErrorOutOfMemory ? FXC_ERR_NOMEM : FXC_ERR_HIP, "pipeline setup failed: %s",
                    hipGetErrorString(e));
    }
    p->live_pipes += 1;
    q->counted = true;
    *out = q;
    return FXC_OK;
}

int fxc_pipe_in_flight(const fxc_pipe* q) { return q ? (int)(q->pushed - q->popped) : 0; }

int fxc_pipe_acquire(fxc_pipe* q, void** in_host) {
    if (!q || !in_host) return fail(q ? q->plan : nullptr, FXC_ERR_ARG, "NULL argument");
    if (q->pushed - q->popped >= q->depth)
        return fail(q->plan, FXC_ERR_STATE, "all %d slots in flight: pop first", q->depth);
    *in_host = q->slots[(size_t)(q->pushed % q->depth)].h_in;      // popped, hence idle
    return FXC_OK;
}

int fxc_pipe_submit(fxc_pipe* q) {
    if (!q) return fail(nullptr, FXC_ERR_ARG, "NULL pipe");
    fxc_plan* p = q->plan;
    if (q->pushed - q->popped >= q->depth) return fail(p, FXC_ERR_STATE, "all %d slots in flight: pop first", q->depth);
    FXC_DEVICE(p, p->device);
    fxc_pipe_slot& sl = q->slots[(size_t)(q->pushed % q->depth)];
    FXC_HIP(p, hipMemcpyAsync(sl.d_in, sl.h_in, q->in_bytes, hipMemcpyHostToDevice, q->s_in));
    FXC_HIP(p, hipEventRecord(sl.ev_in, q->s_in));
    FXC_HIP(p, hipStreamWaitEvent(p->stream, sl.ev_in, 0));
    int rc = q->u8 ? fx_u8_dev(p, static_cast<const unsigned char*>(sl.d_in), sl.d_out, q->chunks, q->mode, q->bandwidth,
                               q->remove_dc, true)
                   : fx_rows_dev(p, static_cast<const cf*>(sl.d_in), sl.d_out, q->chunks, q->mode, q->bandwidth);
    if (rc) return rc;
    FXC_HIP(p, hipEventRecord(sl.ev_compute, p->stream));
    FXC_HIP(p, hipStreamWaitEvent(q->s_out, sl.ev_compute, 0));
    FXC_HIP(p, hipMemcpyAsync(sl.h_out, sl.d_out, q->out_bytes, hipMemcpyDeviceToHost, q->s_out));
    FXC_HIP(p, hipEventRecord(sl.ev_out, q->s_out));
    sl.busy = true;
    q->pushed += 1;
    return FXC_OK;
}

int fxc_pipe_push(fxc_pipe* q, const void* x_host) {
    if (!q || !x_host) return fail(q ? q->plan : nullptr, FXC_ERR_ARG, "NULL argument");
    void* dst = nullptr;
    int rc = fxc_pipe_acquire(q, &dst);
    if (rc) return rc;
    std::memcpy(dst, x_host, q->in_bytes);
    return fxc_pipe_submit(q);
}

int fxc_pipe_pop(fxc_pipe* q, void* out_host) {
    if (!q || !out_host) return fail(q ? q->plan : nullptr, FXC_ERR_ARG, "NULL argument");
    fxc_plan* p = q->plan;
    if (q->pushed == q->popped) return fail(p, FXC_ERR_STATE, "nothing in flight");
    FXC_DEVICE(p, p->device);
    fxc_pipe_slot& sl = q->slots[(size_t)(q->popped % q->depth)];
    FXC_HIP(p, hipEventSynchronize(sl.ev_out));
    std::memcpy(out_host, sl.h_out, q->out_bytes);
    sl.busy = false;
    q->popped += 1;
    return FXC_OK;
}

int fxc_timer_start(fxc_plan* p) {
    if (!p) return fail(p, FXC_ERR_ARG, "NULL plan");
    FXC_DEVICE(p, p->device);
    FXC_HIP(p, hipEventRecord(p->ev_t0, p->stream));
    return FXC_OK;
}

int fxc_timer_stop(fxc_plan* p, double* elapsed_ms) {
    if (!p || !elapsed_ms) return fail(p, FXC_ERR_ARG, "NULL argument");
    FXC_DEVICE(p, p->device);
    FXC_HIP(p, hipEventRecord(p->ev_t1, p->stream));
    FXC_HIP(p, hipEventSynchronize(p->ev_t1));
    float ms = 0.f;
    FXC_HIP(p, hipEventElapsedTime(&ms, p->ev_t0, p->ev_t1));
    *elapsed_ms = ms;
    return FXC_OK;
}

int fxc_kernel_profiling(fxc_plan* p, int enable) {
    if (!p) return fail(p, FXC_ERR_ARG, "NULL plan");
    p->profiling = enable != 0;
    return FXC_OK;
}

int fxc_kernel_time(fxc_plan* p, double* total_ms, int64_t* launches, int reset) {
    if (!p || !total_ms || !launches) return fail(p, FXC_ERR_ARG, "NULL argument");
    FXC_DEVICE(p, p->device);
    int rc = drain_kernel_events(p);
    if (rc) return rc;
#if FXC_STAMPS
    if (p->stamp_grid > 0 && p->d_stamps) {
        const int nw = p->stamp_grid * 8;
        std::vector<unsigned long long> h((size_t)nw * kStampSegs);
        FXC_HIP(p, hipStreamSynchronize(p->stream));
        FXC_HIP(p, hipMemcpy(h.data(), p->d_stamps, h.size() * 8, hipMemcpyDeviceToHost));
        double sum[kStampSegs] = {0};
        double steps = 0;
        for (int w = 0; w < nw; ++w) {
            for (int k = 0; k < kStampSegs - 1; ++k) sum[k] += (double)h[(size_t)w * kStampSegs + k];
            steps += (double)h[(size_t)w * kStampSegs + kStampSegs - 1];
        }
        double tot = 0;
        for (int k = 0; k < kStampSegs - 1; ++k) tot += sum[k];
        fprintf(stderr, "[fxc stamps] cycles per step per wave (s_memtime ticks), total %.0f:", tot / steps);
        for (int k = 0; k < kStampSegs - 1; ++k) fprintf(stderr, " s%d=%.0f", k, sum[k] / steps);
        fprintf(stderr, "\n");
    }
#endif
    *total_ms = p->kernel_ms;
    *launches = p->kernel_launches;
    if (reset) {
        p->kernel_ms = 0.0;
        p->kernel_launches = 0;
    }
    return FXC_OK;
}

int fxc_synth_fill(int device, void* stream, void* x_dev, uint64_t seed, int64_t first_chunk, int64_t n_chunks,
                   int n_ant, int64_t num_samp, const int32_t* delays, const float* tone_re_im, int tone_period) {
    if (!x_dev || !delays || !tone_re_im) return fail(nullptr, FXC_ERR_ARG, "NULL argument");
    if (n_chunks < 0 || n_ant < 1 || num_samp < 1 || tone_period < 1) return fail(nullptr, FXC_ERR_ARG, "bad size");
    if (n_chunks == 0) return FXC_OK;
    const fxc_plan* p = nullptr;
    FXC_DEVICE(p, device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    float lut[256];
    for (int b = 0; b < 256; ++b) lut[b] = ((float)b - 127.5f) / 127.5f;
    int* d_delays = nullptr;
    cf* d_tone = nullptr;
    float* d_lut = nullptr;
    FXC_HIP(p, hipMalloc(&d_delays, (size_t)n_ant * sizeof(int)));
    FXC_HIP(p, hipMalloc(&d_tone, (size_t)tone_period * sizeof(cf)));
    FXC_HIP(p, hipMalloc(&d_lut, sizeof lut));
    FXC_HIP(p, hipMemcpy(d_delays, delays, (size_t)n_ant * sizeof(int), hipMemcpyHostToDevice));
    FXC_HIP(p, hipMemcpy(d_tone, tone_re_im, (size_t)tone_period * sizeof(cf), hipMemcpyHostToDevice));
    FXC_HIP(p, hipMemcpy(d_lut, lut, sizeof lut, hipMemcpyHostToDevice));
    const int64_t total = n_chunks * n_ant * num_samp;
    int64_t grid = (total + 255) / 256;
    if (grid > 256 * 16) grid = 256 * 16;
    hipLaunchKernelGGL(synth_kernel, dim3((int)grid), dim3(256), 0, st, static_cast<cf*>(x_dev), seed, first_chunk,
                       n_chunks, n_ant, num_samp, d_delays, d_tone, tone_period, d_lut);
    hipError_t e = hipGetLastError();
    hipError_t e2 = hipStreamSynchronize(st);
    (void)hipFree(d_delays);
    (void)hipFree(d_tone);
    (void)hipFree(d_lut);
    if (e != hipSuccess) return fail(nullptr, FXC_ERR_HIP, "synth launch failed: %s", hipGetErrorString(e));
    if (e2 != hipSuccess) return fail(nullptr, FXC_ERR_HIP, "synth sync failed: %s", hipGetErrorString(e2));
    return FXC_OK;
}

}  // extern "C"
