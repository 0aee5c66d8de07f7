// Part of libfxcorr's single translation unit: included by fxcorr.hip (not a stand-alone header).
#pragma once

namespace {

// ------------------------------------------------------------------------------------------
// fused 2-antenna F+X kernel (and its F-only variant) for the small channel counts nchan = 16 P, P in {1, 2, 4, 8, 16}
// (16 ... 256), ntaps <= 4
// (--nfft is a free integer in the reference, effex/effex.py:733-739; its ntaps is 4, effex.py:115).
//
// The design of k_tiled.h's ring kernel with the workgroup taken out: a transform this small fits P lanes, so nothing
// crosses a wave and there is no s_barrier after the set-up.  Lanes 0-31 of a wave carry antenna 0, lanes 32-63 antenna 1;
// each half holds 32 / P work items -- (chunk, range of frames), as in the tiled kernels -- of P adjacent lanes.  Lane u of
// an item owns the 16 branches m = u + P r (sample N - 1 - m of a frame), keeps four frames of them in a VGPR ring (every
// sample is fetched once) and the window as quads in LDS.  Decimation in frequency, bin k = k1 + 16 k2:
//   radix-16 over r in registers                      Y[u][k1] = sum_r v[r] w16^(r k1)
//   twiddle wN^(u k1)                                 [registers]
//   transposition inside the P lanes of the item      lane j takes the 16 / P values k1 = j 16/P + t, all u   [LDS, no barrier]
//   P-point DFTs over u in registers                  v[t P + k2] = bin (j 16/P + t) + 16 k2
//   v_permlane32_swap pairs the antennas as in fx_fused4096.h; 8 accumulators per lane.
// Raw rows in natural bin order: raw[(split * n_chunks + c) * N + k], the layout of fx_tiled_kernel (h_run.h folds them).
// The per-lane phases are in fx_small.h (tests/emul runs them on the host).
// ------------------------------------------------------------------------------------------
// CNT branches of a frame into ring registers; U8: the stream is byte pairs (uint8 I, Q), a pair goes into .x as it is
// and convert_frame_u8 turns the slot into samples when its frame comes up (k_fused4096.h)
// (default cache policy on the sample loads: nontemporal measured 2.4 x SLOWER at 16 channels and no gain at 256, profiles/r05/experiments.md 10)
template <class G, int CNT, bool U8>
__device__ __forceinline__ void small_load(cf (&xr)[16], const cf* __restrict__ frame, int r0) {
    if (U8) {
        const unsigned short* f8 = reinterpret_cast<const unsigned short*>(frame);
#pragma unroll
        for (int r = r0; r < r0 + CNT; ++r) xr[r].x = __uint_as_float((unsigned)f8[G::P * (15 - r)]);
    } else {
#pragma unroll
        for (int r = r0; r < r0 + CNT; ++r) xr[r] = frame[G::P * (15 - r)];
    }
}

// frame f of the stream whose branch-u sample of frame 0 is at px (a pointer to samples, or to byte pairs for U8)
template <class G, bool U8>
__device__ __forceinline__ const cf* small_frame(const cf* px, int64_t f) {
    if (U8) return reinterpret_cast<const cf*>(reinterpret_cast<const unsigned short*>(px) + f * G::N);
    return px + f * G::N;
}

template <class G>
struct SmallRing {
    cf h[4][16];
    cf acc[8];
};

// one spectrum of both antennas for every item of the wave; the item's frame i sits in ring slot PH; `next` = the frame to
// fetch into the slot that becomes free (clamped into the chunk by the caller), `active`: frame i belongs to the item's range
// SPEC (F-only): the spectrum of frame i goes to out_row in natural bin order instead of into the X-stage
template <class G, int PH, bool U8, bool SPEC>
__device__ __forceinline__ void small_ring_step(SmallRing<G>& s, const f4* __restrict__ win, const cf* __restrict__ tw,
                                                cf* __restrict__ grp, int u, const cf* __restrict__ next, bool active, cf off,
                                                cf* __restrict__ out_row) {
    constexpr int P = G::P;
    if (U8) convert_frame_u8(s.h[PH], off);   // the byte pairs fetched a step ago become the samples of slot PH
    cf v[16];
    G::template fir_ring<PH>(s.h, win, u, v);
    // the oldest slot is dead: refill it (unconditionally: no branch guards a definition of ring registers)
    cf (&nx)[16] = s.h[(PH + 1) & 3];
    small_load<G, 8, U8>(nx, next, 0);
    fxc::dft16(v);
    small_load<G, 8, U8>(nx, next, 8);
    if (P > 1) {
        G::twiddle(v, tw, u);
        wave_sync();
        G::store(v, grp, u);
        wave_sync();
        G::load(grp, u, v);
        G::transforms(v);
    }
    if (SPEC) {
        if (active) {
#pragma unroll
            for (int idx = 0; idx < 16; ++idx) out_row[G::bin_of(u, idx)] = v[idx];
        }
        return;
    }
    // lanes 0-31 hold antenna 0, lanes 32-63 antenna 1 of the same item and bins (see fused_step)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        cf a = v[q], b = v[q + 8];
        permlane32_swap(a, b);
        const cf prod = fxc::cmulc(a, b);
        if (active) s.acc[q] = fxc::cadd(s.acc[q], prod);
    }
}

// U8: x is the receivers' bytes (uint8 I,Q pairs), dc[c * 2 + ant] the conversion offset of a stream (k_conditioning.h).
// SPEC: F-only -- a "chunk" is a pair of consecutive streams (n_streams of them in all: an odd count leaves the last pair
// half empty), raw is the spectra buffer with rows placed by spec_row (k_tiled.h), n_chunks the number of pairs.
template <int P, bool U8 = false, bool SPEC = false>
__global__ __launch_bounds__(256, 2) void fx_small_ring_kernel(const cf* __restrict__ x, int64_t num_samp, int64_t n_pts,
                                                              int64_t n_chunks, int n_splits, const f4* __restrict__ win_g,
                                                              const cf* __restrict__ tw_g, cf* __restrict__ raw,
                                                              const cf* __restrict__ dc, int64_t n_streams, int spec_a,
                                                              int64_t s_base) {
    static_assert(!(SPEC && U8), "uint8 ingest: F+X only");
    using G = fxc::small::Geo<P>;
    __shared__ f4 win[G::N];
    __shared__ cf tw[G::N];
    __shared__ cf xchg[G::kWaves * G::kXchgPerWave];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, ant = lane >> 5, l32 = lane & 31;
    const int sub = l32 / P;
    const int u = l32 % P;
    for (int idx = tid; idx < G::N; idx += G::kThreads) {
        win[idx] = win_g[idx];
        tw[idx] = tw_g[idx];
    }
    SmallRing<G> s;
    __syncthreads();
    cf* grp = xchg + wave * G::kXchgPerWave + (lane / P) * G::kGroup;
    const int per = (int)((n_pts + n_splits - 1) / n_splits);      // frame counters are 32-bit: n_pts < 2^31 (small_setup)
    const int n_pts_i = (int)n_pts;
    const int64_t total = n_chunks * n_splits;
    const int64_t stride = (int64_t)gridDim.x * G::kItemsPerWg;
    for (int64_t w0 = ((int64_t)blockIdx.x * G::kWaves + wave) * G::kSub; w0 < total; w0 += stride) {
        // this lane's item; the items of a wave beyond the last one run on the last one's samples and store nothing
        int sub_l = sub, ant_l = ant;
        // (F-only variant at 16 channels, one item per lane: widened to 64 bits and added to the kernel's pointers here, per
        // item -- as invariants of the item loop those lane values were kept in scratch)
        if (SPEC && P == 1) asm volatile("" : "+v"(sub_l), "+v"(ant_l));
        const int64_t w = w0 + sub_l;
        const bool live = w < total;
        const int64_t wc = live ? w : total - 1;
        // consecutive items = consecutive frame ranges of one chunk: the lanes of a wave then read from a few streams (2 MiB
        // apart) instead of from up to 64 -- at 16 and 32 channels, a lane or two per item, address translation for 64
        // streams per load instruction held the kernel at 0.35 / 0.45 of 8 TB/s (now 0.72 / 0.73; eight 16-byte loads per
        // frame instead of sixteen 8-byte ones, measured before and after this change: no difference either time)
        const int64_t c = wc / n_splits, split = wc - c * n_splits;
        const int i0 = (int)split * per;
        const int i1 = !live ? i0 : ((i0 + per < n_pts_i) ? i0 + per : n_pts_i);
        // SPEC with an odd stream count: the missing second stream of the last pair re-reads the first and stores nothing
        const bool valid = !SPEC || (2 * c + ant_l) < n_streams;
        const int ant_ld = valid ? ant_l : 0;
        // branch u of frame 0 (U8: the same element count in byte pairs)
        const cf* px = U8 ? reinterpret_cast<const cf*>(reinterpret_cast<const unsigned short*>(x) + (c * 2 + ant_ld) * num_samp + (P - 1 - u))
                          : x + (c * 2 + ant_ld) * num_samp + (P - 1 - u);
        cf* out_base = SPEC ? raw + spec_row(s_base + 2 * c + ant_l, 0, n_pts, spec_a) * G::N : nullptr;
        const int64_t out_step = (int64_t)(spec_a > 0 ? spec_a : 1) * G::N;
        const cf off = U8 ? dc[c * 2 + ant_l] : fxc::mk(0.f, 0.f);
#pragma unroll
        for (int q = 0; q < 8; ++q) s.acc[q] = fxc::mk(0.f, 0.f);
        // ring prologue: frame i0 -> slot 0, its history i0-1, i0-2, i0-3 -> slots 3, 2, 1 (zero before the chunk)
#pragma unroll
        for (int d = 1; d < 4; ++d) {
            const int f = i0 - d;
            const bool have = f >= 0 && i0 < i1;
            small_load<G, 16, U8>(s.h[4 - d], small_frame<G, U8>(px, have ? f : 0), 0);
            if (U8) convert_frame_u8(s.h[4 - d], off);
            if (!have) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s.h[4 - d][r] = fxc::mk(0.f, 0.f);
            }
        }
        small_load<G, 16, U8>(s.h[0], small_frame<G, U8>(px, i0 < n_pts_i ? i0 : n_pts_i - 1), 0);
        for (int st = 0; st < per; st += 4) {
#define FXC_SMALL_STEP(PH)                                                                                     \
    {                                                                                                          \
        const int i = i0 + st + PH;                                                                            \
        const int nf = i + 1 < n_pts_i ? i + 1 : n_pts_i - 1;                                                  \
        small_ring_step<G, PH, U8, SPEC>(s, win, tw, grp, u, small_frame<G, U8>(px, nf), valid && i < i1, off, \
                                         SPEC ? out_base + i * out_step : nullptr);                            \
    }
            FXC_SMALL_STEP(0)
            if (st + 1 < per) FXC_SMALL_STEP(1)
            if (st + 2 < per) FXC_SMALL_STEP(2)
            if (st + 3 < per) FXC_SMALL_STEP(3)
#undef FXC_SMALL_STEP
        }
        if (!SPEC && live) {
            cf* row = raw + (split * n_chunks + c) * G::N;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                row[G::bin_of(u, q + 8 * ant)] = s.acc[q];
            }
        }
    }
}

}  // namespace
