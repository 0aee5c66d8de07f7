// Part of libfxcorr's single translation unit: included by fxcorr.hip (not a stand-alone header).
#pragma once

namespace {

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
    // the workgroups that start strictly inside that chunk (up to grid - 1 of them when one chunk pair is the whole call,
    // effex.py:490-494): four loads in flight, added in the order of the plain loop
    const cf* __restrict__ src = raw + lr.offset + ridx;
    int64_t b = b_lo + 1;
    for (; b + 3 <= b_hi; b += 4) {
        const cf r0 = src[b * row_len], r1 = src[(b + 1) * row_len], r2 = src[(b + 2) * row_len], r3 = src[(b + 3) * row_len];
        ar = ((ar + r0.x) + r1.x) + r2.x + r3.x;
        ai = ((ai + r0.y) + r1.y) + r2.y + r3.y;
    }
    for (; b <= b_hi; ++b) {
        const cf r = src[b * row_len];
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
        // sixteen loads in flight, added in the order of the splits (one at a time a bin of a single chunk with 256 rows -- few
        // channels, many slots -- waited out 256 trips to L2: 0.1 ms)
        const cf* src = raw + row * nchan + raw_index(k, slots);
        int s = 0;
        for (; s + 16 <= n_splits; s += 16) {
            cf r[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) r[q] = src[(s + q) * split_stride];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                ar += r[q].x;
                ai += r[q].y;
            }
        }
        for (; s < n_splits; ++s) {
            const cf r = src[s * split_stride];
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

// sum over the splits of one bin's raw values, in split order, sixteen loads in flight (see rows_spectrum_kernel)
__device__ __forceinline__ void sum_splits(const cf* src, int n_splits, int64_t split_stride, double& xr, double& xi) {
    int s = 0;
    for (; s + 16 <= n_splits; s += 16) {
        cf r[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) r[q] = src[(s + q) * split_stride];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            xr += r[q].x;
            xi += r[q].y;
        }
    }
    for (; s < n_splits; ++s) {
        const cf r = src[s * split_stride];
        xr += r.x;
        xi += r.y;
    }
}

// CONTINUUM rows: out[row] = mean_k( raw * conj(rot) / n_pts ) / bandwidth   (effex.py:523-524); one WG per row, of
// kContinuumThreads threads: a reference-sized call is a single row whose frames the F+X kernel spread over the whole grid,
// so each bin gathers up to grid - 1 leading-part rows -- 72 us with 256 threads, the largest item of that call
constexpr int kContinuumThreads = 1024;
inline int continuum_threads(int nchan) { return nchan >= kContinuumThreads ? kContinuumThreads : 256; }
__global__ __launch_bounds__(kContinuumThreads) void rows_continuum_kernel(const cf* __restrict__ raw, cd* __restrict__ out,
                                                            const cd* __restrict__ rot, int nchan, int64_t rows,
                                                            int n_splits, int64_t split_stride, double scale,
                                                            int slots, LeadRows lead) {
    __shared__ double red[kContinuumThreads];
    for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
        double ar = 0.0, ai = 0.0;
        for (int k = threadIdx.x; k < nchan; k += blockDim.x) {
            double xr = 0.0, xi = 0.0;
            sum_splits(raw + row * nchan + raw_index(k, slots), n_splits, split_stride, xr, xi);
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

// The same for a call of few rows (the reference's own call is ONE: effex.py:490-494): a row's bins are cut into `slices`
// workgroups (grid = slices x rows) that leave float64 partial sums, and rows_continuum_fin_kernel adds them in slice
// order -- one workgroup per row gathered a chunk pair's up to 255 leading-part rows for all 4096 bins in 33 us, the largest
// item of that call.
__global__ __launch_bounds__(256) void rows_continuum_part_kernel(const cf* __restrict__ raw, cd* __restrict__ part,
                                                                 const cd* __restrict__ rot, int nchan, int64_t rows, int n_splits,
                                                                 int64_t split_stride, int slots, LeadRows lead, int slices) {
    __shared__ double red[256];
    const int64_t row = blockIdx.y;
    const int per = (nchan + slices - 1) / slices;
    const int k_lo = blockIdx.x * per, k_hi = k_lo + per < nchan ? k_lo + per : nchan;
    double ar = 0.0, ai = 0.0;
    for (int k = k_lo + threadIdx.x; k < k_hi; k += blockDim.x) {
        double xr = 0.0, xi = 0.0;
        sum_splits(raw + row * nchan + raw_index(k, slots), n_splits, split_stride, xr, xi);
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
        o.x = ar;
        o.y = ai;
        part[row * slices + blockIdx.x] = o;
    }
}

__global__ void rows_continuum_fin_kernel(const cd* __restrict__ part, cd* __restrict__ out, int64_t rows, int slices, double scale) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    double ar = 0.0, ai = 0.0;
    for (int s = 0; s < slices; ++s) {
        ar += part[row * slices + s].x;
        ai += part[row * slices + s].y;
    }
    cd o;
    o.x = ar * scale;
    o.y = ai * scale;
    out[row] = o;
}

// What the kernel that finishes an integration does with an accumulator element besides updating it: export it for the
// cross-rank reduce, finalise it, clear it -- in the same launch instead of export + finalize + device-to-host copy +
// memset (round 2: four commands and 50 us of gaps per integration).
struct FoldFinish {
    cd* sums;         // != nullptr: sums[idx] = acc[idx] (raw float64 sums) and sums[n] = {count, 0} -- fxc_acc_export
    cd* out;          // != nullptr: out[p][(k + N/2) % N] = acc * conj(rot[k]) / count (effex.py:520-521, integrated);
                      //             device memory or host memory mapped into the device (the plan's pinned result
                      //             slots).  Every kernel that fills it does so with consecutive lanes on consecutive
                      //             bins: single 16-byte stores scattered over mapped host memory cost ~45 ns each
                      //             (180 us per 4096-bin spectrum, measured), wave-wide runs go out at the copy rate
    const cd* rot;
    double count;     // spectra accumulated so far
    int reset;        // clear the accumulator afterwards
};

__device__ __forceinline__ void finish_element(cd a, cd* __restrict__ acc, int64_t idx, int k, int nchan, int64_t n,
                                               const FoldFinish& fin) {
    if (fin.sums) {
        fin.sums[idx] = a;
        if (idx == 0) {
            cd c;
            c.x = fin.count;
            c.y = 0.0;
            fin.sums[n] = c;
        }
    }
    if (fin.out) {
        const double inv = 1.0 / fin.count;
        const cd w = fin.rot[k];
        cd o;
        o.x = (a.x * w.x + a.y * w.y) * inv;
        o.y = (a.y * w.x - a.x * w.y) * inv;
        int ks = k + nchan / 2;
        if (ks >= nchan) ks -= nchan;
        fin.out[idx - k + ks] = o;
    }
    if (fin.reset) a.x = a.y = 0.0;
    acc[idx] = a;
}

// Integration of the kernels' raw float32 rows (row = [n_base][nchan] sums of spectrum products, each baseline in the
// layout `slots`): acc[p][bin] += sum over all rows, in float64 and in a fixed order (bit-reproducible), then FoldFinish.  Two launches back to back, no fences, no atomics (a single
// kernel with a last-workgroup-done ticket per column was tried first: its agent-scope release / acquire fences write
// back and invalidate the L2s once per workgroup and took 190 us):
//   fold_partial_kernel  workgroup (column of 256 slots, split): thread (slot, phase) walks the rows
//                        split * 4 + phase + 4 * n_splits * j (coalesced 8-byte loads, four in flight); the four phases
//                        are combined through LDS in phase order -> part[split][slot]
//   fold_finish_kernel   thread (bin, phase): sums every fourth of the n_rows rows it is given at raw_index(bin) -- the
//                        n_splits partials, or the raw rows themselves when they are few (one launch then) -- phases
//                        combined in order; acc[bin] += that; FoldFinish with consecutive lanes on consecutive bins
// `slots`: layout of a row (raw_index): 0 natural order, 1 the fused kernel's slot order, 3 the 8192-channel split
constexpr int kFoldPhases = 4;
constexpr int kFoldMaxSplits = 32;
__global__ __launch_bounds__(1024) void fold_partial_kernel(const cf* __restrict__ raw, cd* __restrict__ part, int64_t nchan,
                                                           int64_t n_rows, int n_splits) {   // (nchan: the row length)
    __shared__ cd sub[kFoldPhases][256];
    const int kl = threadIdx.x & 255, ph = threadIdx.x >> 8;
    const int64_t slot = (int64_t)blockIdx.x * 256 + kl;
    const int split = blockIdx.y;
    const bool live = slot < nchan;
    double ar = 0.0, ai = 0.0;
    if (live) {
        const int64_t step = (int64_t)n_splits * kFoldPhases;
        const cf* col = raw + slot;
        int64_t c = (int64_t)split * kFoldPhases + ph;
        for (; c + 3 * step < n_rows; c += 4 * step) {
            const cf r0 = col[c * nchan], r1 = col[(c + step) * nchan], r2 = col[(c + 2 * step) * nchan],
                     r3 = col[(c + 3 * step) * nchan];
            ar += r0.x;
            ai += r0.y;
            ar += r1.x;
            ai += r1.y;
            ar += r2.x;
            ai += r2.y;
            ar += r3.x;
            ai += r3.y;
        }
        for (; c < n_rows; c += step) {
            const cf r = col[c * nchan];
            ar += r.x;
            ai += r.y;
        }
    }
    sub[ph][kl].x = ar;
    sub[ph][kl].y = ai;
    __syncthreads();
    if (ph == 0 && live) {
        cd v = sub[0][kl];
#pragma unroll
        for (int q = 1; q < kFoldPhases; ++q) {
            v.x += sub[q][kl].x;
            v.y += sub[q][kl].y;
        }
        part[(int64_t)split * nchan + slot] = v;
    }
}

template <class RowT>
__global__ __launch_bounds__(1024) void fold_finish_kernel(const RowT* __restrict__ rows, int64_t n_rows, cd* __restrict__ acc,
                                                          int nchan, int n_base, int slots, FoldFinish fin) {
    __shared__ cd sub[kFoldPhases][256];
    const int kl = threadIdx.x & 255, ph = threadIdx.x >> 8;
    const int64_t n = (int64_t)n_base * nchan;
    const int64_t idx = (int64_t)blockIdx.x * 256 + kl;
    const bool live = idx < n;
    const int k = (int)(idx % nchan);
    double ar = 0.0, ai = 0.0;
    if (live) {
        const RowT* col = rows + (idx - k) + raw_index(k, slots);
        int64_t s = ph;
        for (; s + 3 * kFoldPhases < n_rows; s += 4 * kFoldPhases) {      // four loads in flight; summed in row order
            const RowT v0 = col[s * n], v1 = col[(s + kFoldPhases) * n], v2 = col[(s + 2 * kFoldPhases) * n],
                       v3 = col[(s + 3 * kFoldPhases) * n];
            ar += v0.x;
            ai += v0.y;
            ar += v1.x;
            ai += v1.y;
            ar += v2.x;
            ai += v2.y;
            ar += v3.x;
            ai += v3.y;
        }
        for (; s < n_rows; s += kFoldPhases) {
            const RowT v = col[s * n];
            ar += v.x;
            ai += v.y;
        }
    }
    sub[ph][kl].x = ar;
    sub[ph][kl].y = ai;
    __syncthreads();
    if (ph != 0 || !live) return;
    cd a = acc[idx];
#pragma unroll
    for (int q = 0; q < kFoldPhases; ++q) {
        a.x += sub[q][kl].x;
        a.y += sub[q][kl].y;
    }
    finish_element(a, acc, idx, k, nchan, n, fin);
}

// FoldFinish alone, on the accumulator as it stands (paths that update it themselves, or nothing pending)
__global__ void acc_finish_kernel(cd* __restrict__ acc, int nchan, int n_base, FoldFinish fin) {
    const int64_t n = (int64_t)n_base * nchan;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += stride)
        finish_element(acc[idx], acc, idx, (int)(idx % nchan), nchan, n, fin);
}

// multi-antenna X-engine on the F-only kernels' spectra: spectrum (chunk c, frame i, antenna a) is row
// (c * n_pts + i) * A + a of `spec` (rows of nchan samples) -- the A rows a thread needs for one frame lie inside one block
// of A rows (as [stream][frame] rows, 2 MiB apart, the kernel read 8 % slower: 8 antennas 1.64 -> 1.51 ms per 512 chunks).
// One wave per workgroup; a thread keeps all A(A-1)/2 accumulators of one position in registers over the spectra of `cg`
// consecutive chunks (cg = 1: one raw row per chunk; integrations take float32 sums of up to kRowSpectra spectra, like the
// 2-antenna kernel's rows) and reads every spectrum sample exactly once; raw[group][p][pos], baselines ordered
// (0,1),(0,2)..(A-2,A-1) -- effex.py:520 for A > 2.
// Measured and dropped (profiles/r03/experiments.md): two positions per thread with 16-byte loads (+10 %), 1 or 4
// spectra per trip instead of 2 (+-0.5 %).
// A group of chunks whose frames are too many for one float32 sum (more than kRowSpectra of them: one chunk per group, cg = 1)
// is cut into n_ranges frame ranges, each a raw row of its own: blockIdx.y = group * n_ranges + range, raw row =
// range * n_groups + group -- range-major, so that the rows kernels read the ranges as their splits (k_finish.h top).
struct XRange {
    int64_t grp, row, i0, i1;
};
__device__ __forceinline__ XRange x_range(int64_t n_pts, int64_t n_chunks, int cg, int n_ranges) {
    XRange r;
    const int64_t n_groups = (n_chunks + cg - 1) / cg;
    r.grp = blockIdx.y / n_ranges;
    const int64_t rg = blockIdx.y - r.grp * n_ranges;
    const int64_t per = (n_pts + n_ranges - 1) / n_ranges;
    r.i0 = rg * per < n_pts ? rg * per : n_pts;
    r.i1 = r.i0 + per < n_pts ? r.i0 + per : n_pts;
    r.row = rg * n_groups + r.grp;
    return r;
}

#ifndef FXC_NT_X
#define FXC_NT_X 1      // the X-engines read every spectrum once: nontemporal loads (- 1 %)
#endif
#if FXC_NT_X
#define FXC_X_LOAD(p) fxc::nt_load(p)
#else
#define FXC_X_LOAD(p) (*(p))
#endif
constexpr int kXU = 2;           // spectra per trip: kXU * A independent 8-byte loads in flight before the multiply-accumulates
constexpr int kXThreads = 64;
template <int A>
__global__ __launch_bounds__(kXThreads) void xengine_kernel(const cf* __restrict__ spec, cf* __restrict__ raw, int64_t n_pts,
                                                           int nchan, int64_t n_chunks, int cg, int n_ranges) {
    constexpr int NB = A * (A - 1) / 2;
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= nchan) return;          // 16 and 32 channels: part of a wave
    const XRange xr = x_range(n_pts, n_chunks, cg, n_ranges);
    const int64_t grp = xr.grp;
    float ar[NB], ai[NB];
#pragma unroll
    for (int p = 0; p < NB; ++p) ar[p] = ai[p] = 0.f;
    const int64_t c_end = (grp + 1) * cg < n_chunks ? (grp + 1) * cg : n_chunks;
    for (int64_t c = grp * cg; c < c_end; ++c) {
        const cf* base = spec + (c * A * n_pts) * nchan + pos;
        int64_t i = xr.i0;
        for (; i + kXU <= xr.i1; i += kXU) {
            cf z[kXU][A];
#pragma unroll
            for (int u = 0; u < kXU; ++u)
#pragma unroll
                for (int a = 0; a < A; ++a) z[u][a] = FXC_X_LOAD(base + ((i + u) * A + a) * nchan);
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
        for (; i < xr.i1; ++i) {
            cf z[A];
#pragma unroll
            for (int a = 0; a < A; ++a) z[a] = FXC_X_LOAD(base + (i * A + a) * nchan);
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
    for (int p = 0; p < NB; ++p) raw[(xr.row * NB + p) * nchan + pos] = fxc::mk(ar[p], ai[p]);
}

// More than 8 antennas: the same X-engine over blocks of kXB antennas.  A workgroup (one wave, as above) takes a column of
// 64 positions, a group of chunks and one pair of antenna blocks (I <= J, blockIdx.z): the 64 products of block I with
// block J, or the 28 of a block with itself, accumulate in registers; every spectrum row is read (A / kXB + 1) / 2 times
// on average.  Antennas past the last one of a partial block are read as the last one and their products dropped.
// raw[group][p][pos] with the baselines in the order of xengine_kernel: p(a, b) = a A - a (a + 1) / 2 + b - a - 1.
constexpr int kXB = 8;
__global__ __launch_bounds__(kXThreads) void xengine_block_kernel(const cf* __restrict__ spec, cf* __restrict__ raw,
                                                                 int64_t n_pts, int nchan, int64_t n_chunks, int cg, int A,
                                                                 int n_ranges) {
    const int G = (A + kXB - 1) / kXB;
    int bi = 0, rem = (int)blockIdx.z;
    while (rem >= G - bi) {
        rem -= G - bi;
        ++bi;
    }
    const int bj = bi + rem;
    const bool diag = bi == bj;
    const int a0 = bi * kXB, b0 = bj * kXB;
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= nchan) return;
    const XRange xr = x_range(n_pts, n_chunks, cg, n_ranges);
    const int64_t grp = xr.grp;
    int ra[kXB], rb[kXB];          // row offsets of the blocks' antennas inside a frame's A rows
#pragma unroll
    for (int k = 0; k < kXB; ++k) {
        ra[k] = (a0 + k < A ? a0 + k : A - 1) * nchan;
        rb[k] = (b0 + k < A ? b0 + k : A - 1) * nchan;
    }
    float ar[kXB * kXB], ai[kXB * kXB];
#pragma unroll
    for (int p = 0; p < kXB * kXB; ++p) ar[p] = ai[p] = 0.f;
    const int64_t c_end = (grp + 1) * cg < n_chunks ? (grp + 1) * cg : n_chunks;
    for (int64_t c = grp * cg; c < c_end; ++c) {
        const cf* base = spec + (c * A * n_pts) * nchan + pos;
        for (int64_t i = xr.i0; i < xr.i1; ++i) {
            const cf* frame = base + i * A * nchan;
            cf za[kXB], zb[kXB];
#pragma unroll
            for (int k = 0; k < kXB; ++k) za[k] = frame[ra[k]];
            if (diag) {
#pragma unroll
                for (int k = 0; k < kXB; ++k) zb[k] = za[k];
            } else {
#pragma unroll
                for (int k = 0; k < kXB; ++k) zb[k] = frame[rb[k]];
            }
#pragma unroll
            for (int ka = 0; ka < kXB; ++ka)
#pragma unroll
                for (int kb = 0; kb < kXB; ++kb) {
                    ar[ka * kXB + kb] += za[ka].x * zb[kb].x + za[ka].y * zb[kb].y;
                    ai[ka * kXB + kb] += za[ka].y * zb[kb].x - za[ka].x * zb[kb].y;
                }
        }
    }
    const int64_t n_base = (int64_t)A * (A - 1) / 2;
#pragma unroll
    for (int ka = 0; ka < kXB; ++ka)
#pragma unroll
        for (int kb = 0; kb < kXB; ++kb) {
            const int a = a0 + ka, b = b0 + kb;
            if (a < b && b < A) {
                const int64_t p = (int64_t)a * A - (int64_t)a * (a + 1) / 2 + (b - a - 1);
                raw[(xr.row * n_base + p) * nchan + pos] = fxc::mk(ar[ka * kXB + kb], ai[ka * kXB + kb]);
            }
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

}  // namespace
