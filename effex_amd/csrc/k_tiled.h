// Part of libfxcorr's single translation unit: included by fxcorr.hip (not a stand-alone header).
#pragma once

namespace {

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
        for (int n = 0; n < 16; ++n) {
            const cf o = fxc::fused::lds_load(rd + (G::P + G::P / 16) * n);
            fxc::st_store(dst + G::P * n, o);
        }
    }
}

// raw[(split * n_chunks + c) * N + k] = sum over the split's frames of spec0[i,k] * conj(spec1[i,k]), natural
// bin order, float32.  Work item = (split, chunk); a split is a contiguous range of a chunk's frames (the
// FIR reads its history from memory, so ranges are independent).
// row of (stream s, frame i) in the spectra the F-only kernels write: spec_a == 0: [stream][frame] (fxc_channelize's
// output); spec_a = antennas per chunk: [chunk][frame][antenna] -- the rows an X-engine thread needs for one frame side
// by side (k_finish.h::xengine_kernel)
__device__ __forceinline__ int64_t spec_row(int64_t s, int64_t i, int64_t n_pts, int spec_a) {
    if (spec_a <= 0) return s * n_pts + i;
    const int64_t chunk = s / spec_a;
    return (chunk * n_pts + i) * spec_a + (s - chunk * spec_a);
}

// SPEC: F-only -- a "chunk" is a pair of consecutive streams (n_streams of them in all), raw is the spectra
// buffer [stream][i][k] and n_chunks the number of pairs.
template <class G, bool SPEC>
__global__ __launch_bounds__(G::kThreads) void fx_tiled_kernel(const cf* __restrict__ x, int64_t num_samp, int64_t n_pts,
                                                               int64_t n_chunks, int n_splits, int ntaps,
                                                               const float* __restrict__ win, const cf* __restrict__ tw0_g,
                                                               const cf* __restrict__ twA_g, const cf* __restrict__ tw16_g,
                                                               cf* __restrict__ raw, int64_t n_streams, int spec_a,
                                                               int64_t s_base) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cf* region = reinterpret_cast<cf*>(smem + G::kLdsRegion);
    cf* tw16 = reinterpret_cast<cf*>(smem + G::kLdsTw16);
    const int tid = threadIdx.x;
    const int u = G::u_of(tid), ant = G::ant_of(tid);
    for (int idx = tid; idx < 256; idx += G::kThreads) tw16[idx] = tw16_g[idx];
    // The 1024-thread geometry (nchan 8192) has 128 VGPRs per thread: the 8 + 15 twiddles of the pre-stage and of stage A
    // do not fit next to the FIR's 16 samples and 16 coefficients in flight (held, they cost 168 bytes of scratch per lane
    // and their reloads every frame).  There they are fetched again from their tables (96 KiB, L2) where they are used, once
    // the FIR's registers are free: 23 more 8-byte loads per frame next to the FIR's 32 per tap.
    constexpr bool kReloadTw = G::kThreads > 512;
    cf tw0[16], twA[16];
    if (G::R0 > 1 && !kReloadTw) G::load_tw0(tw0, tw0_g, u);
    if (G::A3 && !kReloadTw) G::load_twA(twA, twA_g, u);
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
        const int pair_streams = SPEC ? min(2, (int)n_streams - 2 * (int)c) : 2;      // a scalar minimum: see fx_tiled_ring_kernel
        const unsigned chunk_bytes = (unsigned)(pair_streams * num_samp * (int64_t)sizeof(cf));
        const unsigned xoff = (unsigned)((ant_ld * num_samp + (G::P - 1 - u)) * (int64_t)sizeof(cf));
        cf acc[G::kAccPerThread];
#pragma unroll
        for (int q = 0; q < G::kAccPerThread; ++q) acc[q] = fxc::mk(0.f, 0.f);
        // everything after the FIR for one frame
        auto finish = [&](cf (&v)[16], int64_t i) {
            if (G::R0 > 1) {
                if (kReloadTw) {
                    // a fresh pointer per frame (the loads stay here), typed global: behind the asm a plain pointer has lost its
                    // address space and loads through it are flat -- they count as LDS operations too, and every LDS wait of the
                    // frame would wait for them
                    typedef const __attribute__((address_space(1))) unsigned long long* gu64_t;
                    gu64_t t0 = (gu64_t)tw0_g;
                    asm volatile("" : "+s"(t0));
#pragma unroll
                    for (int r = 0; r < 16; ++r) {                                // = G::load_tw0
                        const unsigned long long w = t0[r * G::P + u];
                        tw0[r] = fxc::mk(__uint_as_float((unsigned)w), __uint_as_float((unsigned)(w >> 32)));
                    }
                }
                G::prestage(v, tw0);
            }
            if (G::A3) {
                if (G::R0 > 1) {
                    __syncthreads();   // every wave has finished reading the previous spectrum's exchange rows
                    G::store0(v, reg, u);
                    __syncthreads();
                    G::loadA(reg, u, v);
                }
                if (kReloadTw) {
                    typedef const __attribute__((address_space(1))) unsigned long long* gu64_t;
                    gu64_t tA = (gu64_t)twA_g;
                    asm volatile("" : "+s"(tA));
#pragma unroll
                    for (int k = 0; k < 16; ++k) {                                // = G::load_twA
                        const unsigned long long w = tA[k * 256 + (u & 255)];
                        twA[k] = fxc::mk(__uint_as_float((unsigned)w), __uint_as_float((unsigned)(w >> 32)));
                    }
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
                tiled_store_spectrum<G>(v, reg, u, raw + spec_row(s_base + 2 * c + ant, i, n_pts, spec_a) * G::N, valid);
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
// AUX: cache policy of the loads.  The F + X ring kernels' steady-state loads are nontemporal (FXC_TILED_RING_AUX: 512 / 1024 / 2048
// channels - 1 ... 2 %); the 8192-channel ring kernels' are not (+ 4 % with it), nor the F-only variant's (profiles/r05/experiments.md 10)
#ifndef FXC_TILED_RING_AUX
#define FXC_TILED_RING_AUX 2
#endif
template <class G, int R0, int CNT, int AUX = 0>
__device__ __forceinline__ void tiled_load_part(cf (&xr)[16], const cf* chunk_base, unsigned chunk_bytes, unsigned xoff,
                                                int64_t frame) {
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<cf*>(chunk_base), 0, (int)chunk_bytes,
                                                                   0x00020000);
    const unsigned soff = (unsigned)(frame * G::N * (int64_t)sizeof(cf));
#pragma unroll
    for (int r = R0; r < R0 + CNT; ++r) {
        const v2u32 d = __builtin_amdgcn_raw_buffer_load_b64(rs, xoff, soff + (unsigned)(G::P * (15 - r) * sizeof(cf)), AUX);
        xr[r] = fxc::mk(__uint_as_float(d[0]), __uint_as_float(d[1]));
    }
}

// uint8 ingest (see load_frame_part_u8): chunk_base then points at byte pairs
template <class G, int R0, int CNT, int AUX = 0>
__device__ __forceinline__ void tiled_load_part_u8(cf (&xr)[16], const cf* chunk_base, unsigned chunk_bytes,
                                                   unsigned xoff, int64_t frame) {
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<cf*>(chunk_base), 0, (int)chunk_bytes,
                                                                   0x00020000);
    const unsigned soff = (unsigned)(frame * G::N * (int64_t)sizeof(unsigned short));
#pragma unroll
    for (int r = R0; r < R0 + CNT; ++r)
        xr[r].x = __uint_as_float(
            (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rs, xoff, soff + (unsigned)(G::P * (15 - r) * sizeof(unsigned short)), AUX));
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
            tiled_load_part_u8<G, R0, 4, SPEC ? 0 : FXC_TILED_RING_AUX>(nx, chunk_base, chunk_bytes, xoff, nframe); \
        else                                                                                    \
            tiled_load_part<G, R0, 4, SPEC ? 0 : FXC_TILED_RING_AUX>(nx, chunk_base, chunk_bytes, xoff, nframe); \
        FXC_SCHED_FENCE();                                                                      \
    } while (0)

// one spectrum of both antennas; frame i sits in ring slot PH, i1 = end of this work item's frame range
template <class G, int PH, bool SPEC, bool U8>
__device__ __forceinline__ void tiled_ring_step(TiledRing<G>& s, const f4* win, cf* reg, const cf* tw16, int u,
                                                const cf* chunk_base, unsigned chunk_bytes, unsigned xoff, int64_t i,
                                                int64_t i1, cf* out_row, int64_t out_step, bool valid) {
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
        tiled_store_spectrum<G>(v, reg, u, out_row + i * out_step, valid);
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
                                                                      int64_t n_streams, const cf* __restrict__ dc, int spec_a,
                                                                      int64_t s_base) {
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
        // streams of this pair that exist, as a scalar minimum.  Written as `2 c + 1 >= n_streams ? 1 : 2` the compiler tied the
        // test to the per-lane `valid` above, formed the descriptor's size word with a v_cndmask and put every one of the F-only
        // kernel's buffer loads into a readfirstlane loop (520 v_readfirstlane in the kernel; the F+X variant: 6).
        const int pair_streams = SPEC ? min(2, (int)n_streams - 2 * (int)c) : 2;
        const unsigned chunk_bytes = (unsigned)(pair_streams * num_samp * kSampleBytes);
        const unsigned xoff = (unsigned)((ant_ld * num_samp + (G::P - 1 - u)) * kSampleBytes);
        if (U8) s.u8.off = dc[c * 2 + ant];
        // SPEC: row of this stream's frame 0 and the rows from frame to frame (spec_row)
        cf* out_row = SPEC ? raw + spec_row(s_base + 2 * c + ant, 0, n_pts, spec_a) * G::N : nullptr;
        const int64_t out_step = (int64_t)(spec_a > 0 ? spec_a : 1) * G::N;
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
            tiled_ring_step<G, 0, SPEC, U8>(s, win, reg, tw16, u, chunk_base, chunk_bytes, xoff, i, i1, out_row, out_step, valid);
            if (i + 1 < i1)
                tiled_ring_step<G, 1, SPEC, U8>(s, win, reg, tw16, u, chunk_base, chunk_bytes, xoff, i + 1, i1, out_row, out_step, valid);
            if (i + 2 < i1)
                tiled_ring_step<G, 2, SPEC, U8>(s, win, reg, tw16, u, chunk_base, chunk_bytes, xoff, i + 2, i1, out_row, out_step, valid);
            if (i + 3 < i1)
                tiled_ring_step<G, 3, SPEC, U8>(s, win, reg, tw16, u, chunk_base, chunk_bytes, xoff, i + 3, i1, out_row, out_step, valid);
        }
        if (SPEC) continue;
        cf* row = raw + (split * n_chunks + c) * G::N;
#pragma unroll
        for (int q = 0; q < G::kAccPerThread; ++q) row[G::bin_of(u, q + 8 * ant)] = s.acc[q];
    }
}


// ------------------------------------------------------------------------------------------
// nchan 8192, ntaps <= 4, F only (fxc_channelize, the drop-in's _spectrometer_poly at 8192 branches -- effex.py:530-555):
// ONE stream per workgroup of 512 threads (16 branches each), so that the four frames the FIR needs fit a VGPR ring (128
// registers; the pair kernel above has 1024 threads of 128 registers and re-reads 2.5 frames per frame) and every sample is
// fetched once.  Beside the exchange rows the LDS holds the stage-A twiddles and 7 / 16 of the window quads (158 KiB in all);
// the other window quads come from L2 every frame, the pre-stage twiddle is a register pair times a constant.  Phases and exchange layout: fx_tiled.h, geometry Geo<2, true> with u = the thread.
// ------------------------------------------------------------------------------------------
using G8192 = fxc::tiled::Geo<2, true>;
constexpr int kF8192Threads = G8192::P;                                          // 512
// exchange rows of one stream + w256 table + the stage-A table + the window quads of the first kF8192WinLds branch groups (the
// other 16 - kF8192WinLds groups -- the window is 128 KiB -- come from L2 every frame): 158 KiB
// (the two-pass route's private spectra -- pass 1's stores, pass 2's loads -- keep the default cache policy: nontemporal measured neutral
// to slightly worse, profiles/r05/experiments.md 10)
constexpr int kF8192WinLds = 7;

constexpr int kF8192LdsCf = G8192::kRegion + 256 + 16 * 256 + kF8192WinLds * G8192::P * 2;

// PRIV: the spectrum leaves the registers as it is -- output k2 of thread u at [k2 P + (P - 1 - u)] of the frame's row (bin_of(u, k2)
// there; the offset register of the sample loads serves): a layout private to the two-pass route, coalesced without the transposition through LDS and its two barriers.  XM: instead of
// storing the spectrum, multiply the OTHER antenna's spectrum of the same frame (in0_row: PRIV layout, written by a launch of the
// <false, true> kernel) by its conjugate and add to acc[k2] = the sum at bin_of(u, k2)
template <int PH, bool XM, bool PRIV, int kX8192Early, bool U8>
__device__ __forceinline__ void f8192_step(cf (&h)[4][16], const f4* __restrict__ win_g, const f4* win_l, cf wu,
                                           const cf* twA_l, cf* reg, const cf* tw16, int u, const cf* stream_base,
                                           unsigned stream_bytes, unsigned xoff, int64_t i, int64_t i1, cf* out_row, int64_t out_step,
                                           const cf* in0_row, cf (&acc)[XM ? 16 : 1], cf off8) {
    using G = G8192;
    if (U8) convert_frame_u8(h[PH], off8);      // the byte pairs fetched a step ago become the samples of slot PH
    const unsigned poff = U8 ? xoff * 4u : xoff;      // (P - 1 - u) complex64 elements, in bytes: the PRIV layout's offset
    cf v[16];
    {   // FIR out of the ring, window quads [r P + u] = taps 0 .. 3 of branch u + P r: the first kF8192WinLds groups from LDS, the
        // rest straight from their table (L2), four at a time
        const cf (&x0)[16] = h[PH];
        const cf (&x1)[16] = h[(PH + 3) & 3];
        const cf (&x2)[16] = h[(PH + 2) & 3];
        const cf (&x3)[16] = h[(PH + 1) & 3];
        __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<f4*>(win_g), 0, (int)(G::N * sizeof(f4)), 0x00020000);
        const unsigned woff = (unsigned)(u * (int)sizeof(f4));
        constexpr int GQ = XM ? 2 : 4;      // (XM: the sums take 32 registers: two quads in flight instead of four)
#pragma unroll
        for (int g = 0; g < 16 / GQ; ++g) {
            FXC_SCHED_FENCE();
            f4 w[GQ];
#pragma unroll
            for (int q = 0; q < GQ; ++q) {
                const int r = GQ * g + q;
                if (r < kF8192WinLds) {
                    w[q] = win_l[r * G::P + u];
                } else {
                    const v4u32 d = __builtin_amdgcn_raw_buffer_load_b128(rw, woff, (unsigned)(r * G::P * (int)sizeof(f4)), 0);
                    w[q].x = __uint_as_float(d[0]);
                    w[q].y = __uint_as_float(d[1]);
                    w[q].z = __uint_as_float(d[2]);
                    w[q].w = __uint_as_float(d[3]);
                }
            }
#pragma unroll
            for (int q = 0; q < GQ; ++q) {
                const int r = GQ * g + q;
                cf a = fxc::cscale(x0[r], w[q].x);
                a = fxc::cfma(w[q].y, x1[r], a);
                a = fxc::cfma(w[q].z, x2[r], a);
                v[r] = fxc::cfma(w[q].w, x3[r], a);
            }
        }
    }
    // the oldest slot is dead: the next frame of the run goes there, in flight through the stages below (the current one again
    // at the end of the run: never used, and no branch guards a definition of ring registers)
    FXC_SCHED_FENCE();
    if (U8)
        tiled_load_part_u8<G, 0, 16>(h[(PH + 1) & 3], stream_base, stream_bytes, xoff, (i + 1 < i1) ? i + 1 : i);
    else
        tiled_load_part<G, 0, 16>(h[(PH + 1) & 3], stream_base, stream_bytes, xoff, (i + 1 < i1) ? i + 1 : i);
    FXC_SCHED_FENCE();
    {   // pre-stage (R0 = 2): slots g and g + 8, twiddle w8192^(u + 512 g) = w8192^u w16^g on the second: the thread's own factor
        // (a register pair) times a constant
        constexpr float kC[8] = {1.f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f,
                                 0.f, -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f};
        constexpr float kS[8] = {0.f, 0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f,
                                 1.f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f};
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            fxc::tiled::dft2(v[g], v[g + 8]);
            v[g + 8] = fxc::cmul(fxc::cmul(v[g + 8], wu), fxc::mk(kC[g], kS[g]));
        }
        FXC_SCHED_FENCE();
    }
    __syncthreads();   // every wave has finished reading the previous spectrum's exchange rows
    G::store0(v, reg, u);
    __syncthreads();
    G::loadA(reg, u, v);
    fxc::dft16(v);
    __syncthreads();
    {   // = G::twiddleA_store with the stage-A twiddles w4096^(n' k) read from their LDS copy five at a time
        cf* b = reg + (u >> 8) * 4352 + (u & 255);
        b[0] = v[0];
#pragma unroll
        for (int k0 = 1; k0 < 16; k0 += 5) {
            FXC_SCHED_FENCE();
            cf w[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) w[k] = fxc::fused::lds_load(twA_l + (k0 + k) * 256 + (u & 255));
#pragma unroll
            for (int k = 0; k < 5; ++k) b[272 * (k0 + k)] = fxc::cmul(v[k0 + k], w[k]);
            FXC_SCHED_FENCE();
        }
    }
    __syncthreads();
    G::loadB(reg, u, v);
    fxc::dft16(v);
    G::twiddleB(v, tw16, u);
    wave_sync();       // the 16x16 transpose stays inside each 16-lane group: no s_barrier
    G::storeT(v, reg, u);
    wave_sync();
    if constexpr (XM) {
        // antenna 0's outputs of the same (thread, k2): kX8192Early of them requested before this antenna's last butterfly (no room
        // for all sixteen beside the ring, the sums and the butterfly's own points), the rest behind it
        cf s0[16];
        __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<cf*>(in0_row + i * (int64_t)G::N), 0, (int)(G::N * sizeof(cf)), 0x00020000);
#pragma unroll
        for (int n = 0; n < kX8192Early; ++n) {
            const v2u32 d = __builtin_amdgcn_raw_buffer_load_b64(r0, poff, (unsigned)(G::P * n * (int)sizeof(cf)), 0);
            s0[n] = fxc::mk(__uint_as_float(d[0]), __uint_as_float(d[1]));
        }
        G::loadC(reg, u, v);
        fxc::dft16(v);
        FXC_SCHED_FENCE();
#pragma unroll
        for (int n = kX8192Early; n < 16; ++n) {
            const v2u32 d = __builtin_amdgcn_raw_buffer_load_b64(r0, poff, (unsigned)(G::P * n * (int)sizeof(cf)), 0);
            s0[n] = fxc::mk(__uint_as_float(d[0]), __uint_as_float(d[1]));
        }
#pragma unroll
        for (int n = 0; n < 16; ++n) acc[n] = fxc::cadd(acc[n], fxc::cmulc(s0[n], v[n]));
    } else {
        G::loadC(reg, u, v);
        fxc::dft16(v);
        if constexpr (PRIV) {
            __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(out_row + i * out_step, 0, (int)(G::N * sizeof(cf)), 0x00020000);
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                v2u32 d;
                d[0] = __float_as_uint(v[n].x);
                d[1] = __float_as_uint(v[n].y);
                __builtin_amdgcn_raw_buffer_store_b64(d, rr, poff, (unsigned)(G::P * n * (int)sizeof(cf)), 0);
            }
        } else {
            tiled_store_spectrum<G>(v, reg, u, out_row + i * out_step, true);
        }
    }
}

// stream_stride: samples from one stream's start to the next's (num_samp; 2 num_samp for one antenna of chunk pairs).  XM: the
// streams are antenna 1 of n_streams chunk pairs, in0 = antenna 0's spectra [chunk][frame][N] in the PRIV layout (a launch of the <false, true> kernel), spec =
// raw[split][chunk][N], the sums over the split's frames of spec0 conj(spec1)
// EARLY: how many of antenna 0's sixteen values a thread requests before its last butterfly (8 / 12 / 16 measured alike: 1.93 - 1.96 ms)
// U8: the streams are interleaved unsigned bytes (I, Q), converted as x / 127.5 + dc[s * 2 + dc_ant] on their way into the ring
template <bool XM, bool PRIV, bool U8 = false, int EARLY = 12>
__global__ __launch_bounds__(kF8192Threads) void f8192_ring_kernel(const cf* __restrict__ x, int64_t num_samp, int64_t n_pts,
                                                                   int64_t n_streams, int n_splits, const f4* __restrict__ win_g,
                                                                   const cf* __restrict__ tw0_g, const cf* __restrict__ twA_g,
                                                                   const cf* __restrict__ tw16_g, cf* __restrict__ spec, int spec_a,
                                                                   int64_t s_base, int64_t stream_stride, const cf* __restrict__ in0,
                                                                   const cf* __restrict__ dc, int dc_ant) {
    using G = G8192;
    constexpr int64_t kSampleBytes = U8 ? sizeof(unsigned short) : sizeof(cf);
    __shared__ __attribute__((aligned(16))) cf smem[kF8192LdsCf];
    cf* reg = smem;
    cf* tw16 = smem + G::kRegion;
    cf* twA_l = tw16 + 256;                       // [16][256]
    f4* win_l = reinterpret_cast<f4*>(twA_l + 16 * 256);      // window quads of branch groups 0 .. kF8192WinLds - 1
    const int u = threadIdx.x;
    for (int idx = u; idx < 256; idx += kF8192Threads) tw16[idx] = tw16_g[idx];
    for (int idx = u; idx < 16 * 256; idx += kF8192Threads) twA_l[idx] = twA_g[idx];
    for (int idx = u; idx < kF8192WinLds * G::P; idx += kF8192Threads) win_l[idx] = win_g[idx];
    const cf wu = tw0_g[8 * G::P + u];            // w8192^u (row g = 0 of the second half of the [16][P] pre-stage table)
    __syncthreads();
    // (uniform, and said so: the 64-bit division runs on the vector unit, and left there the run's bounds and the conditions on them
    // were vector registers;
    // 32-bit: the scalar unit compares no 64-bit integers for order)
    const int npts = (int)n_pts;
    const int per = __builtin_amdgcn_readfirstlane((npts + n_splits - 1) / n_splits);
    const unsigned stream_bytes = (unsigned)(num_samp * kSampleBytes);
    const unsigned xoff = (unsigned)((G::P - 1 - u) * (int)kSampleBytes);
    // grid: (streams, splits) -- no division on the way to a work item
    for (int64_t s = blockIdx.x; s < n_streams; s += gridDim.x) {
        const int split = (int)blockIdx.y;
        const int i0 = split * per, i1 = (i0 + per < npts) ? i0 + per : npts;
        const cf* stream_base = reinterpret_cast<const cf*>(reinterpret_cast<const char*>(x) + s * stream_stride * kSampleBytes);
        const cf off8 = U8 ? dc[s * 2 + dc_ant] : fxc::mk(0.f, 0.f);
        cf* out_row = XM ? nullptr : spec + spec_row(s_base + s, 0, n_pts, spec_a) * G::N;
        const int64_t out_step = (int64_t)(spec_a > 0 ? spec_a : 1) * G::N;
        const cf* in0_row = XM ? in0 + s * n_pts * G::N : nullptr;
        cf acc[XM ? 16 : 1];
#pragma unroll
        for (int n = 0; n < (XM ? 16 : 1); ++n) acc[n] = fxc::mk(0.f, 0.f);
        cf h[4][16];
        // ring prologue: frame i0 -> slot 0, its history i0-1, i0-2, i0-3 -> slots 3, 2, 1 (zero before the stream's start)
#pragma unroll
        for (int d = 1; d < 4; ++d) {
            if (i0 - d >= 0 && i0 < i1) {
                if (U8) {
                    tiled_load_part_u8<G, 0, 16>(h[4 - d], stream_base, stream_bytes, xoff, i0 - d);
                    convert_frame_u8(h[4 - d], off8);
                } else {
                    tiled_load_part<G, 0, 16>(h[4 - d], stream_base, stream_bytes, xoff, i0 - d);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) h[4 - d][r] = fxc::mk(0.f, 0.f);
            }
        }
        if (i0 < i1) {
            if (U8)
                tiled_load_part_u8<G, 0, 16>(h[0], stream_base, stream_bytes, xoff, i0);      // (converted by its step)
            else
                tiled_load_part<G, 0, 16>(h[0], stream_base, stream_bytes, xoff, i0);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) h[0][r] = fxc::mk(0.f, 0.f);
        }
        for (int i = i0; i < i1; i += 4) {
            f8192_step<0, XM, PRIV, EARLY, U8>(h, win_g, win_l, wu, twA_l, reg, tw16, u, stream_base, stream_bytes, xoff, i, i1, out_row, out_step, in0_row, acc, off8);
            if (i + 1 < i1) f8192_step<1, XM, PRIV, EARLY, U8>(h, win_g, win_l, wu, twA_l, reg, tw16, u, stream_base, stream_bytes, xoff, i + 1, i1, out_row, out_step, in0_row, acc, off8);
            if (i + 2 < i1) f8192_step<2, XM, PRIV, EARLY, U8>(h, win_g, win_l, wu, twA_l, reg, tw16, u, stream_base, stream_bytes, xoff, i + 2, i1, out_row, out_step, in0_row, acc, off8);
            if (i + 3 < i1) f8192_step<3, XM, PRIV, EARLY, U8>(h, win_g, win_l, wu, twA_l, reg, tw16, u, stream_base, stream_bytes, xoff, i + 3, i1, out_row, out_step, in0_row, acc, off8);
        }
        if constexpr (XM) {
            // the sums sit at bin_of(u, k2): through the exchange rows into natural order, once per run (tiled_store_spectrum's
            // transposition; buffer stores -- the row's 64-bit address per thread would be a register pair live across the run)
            __syncthreads();
            constexpr int C = 256 * G::R0;
            // u again, from a register the run keeps anyway (the sample loads' offset), behind a barrier for the optimiser: the two
            // LDS addresses below -- or u itself -- hoisted out of the run were spilled
            unsigned xo = xoff;
            asm volatile("" : "+v"(xo));
            const int uu = (int)(G::P - 1) - (int)(xo >> (U8 ? 1 : 3));
            const int b0 = G::bin_of(uu, 0);
            cf* wr = reg + b0 + (b0 >> 4);
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2) wr[(C + C / 16) * k2] = acc[k2];
            __syncthreads();
            const cf* rd = reg + uu + (uu >> 4);
            __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(spec + (split * n_streams + s) * G::N, 0, (int)(G::N * sizeof(cf)), 0x00020000);
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                const cf a = fxc::fused::lds_load(rd + (G::P + G::P / 16) * n);
                v2u32 d;
                d[0] = __float_as_uint(a.x);
                d[1] = __float_as_uint(a.y);
                __builtin_amdgcn_raw_buffer_store_b64(d, rr, (unsigned)(u * (int)sizeof(cf)), (unsigned)(G::P * n * (int)sizeof(cf)), 0);
            }
        }
    }
}

}  // namespace
