// Part of libfxcorr's single translation unit: included by fxcorr.hip (not a stand-alone header).
#pragma once

namespace {

// ------------------------------------------------------------------------------------------
// X-engine for 9 .. 64 antennas on the matrix cores (f32-input MFMA) -- effex.py:520 generalised to A antennas:
// for every bin k the visibility matrix V_k[a][b] = sum_i z[i][a][k] conj(z[i][b][k]) over the frames i of a raw row is a
// Hermitian rank-F update, i.e. per bin one GEMM  Z_k Z_k^H  with the antennas as M and N and the frames as K.
//
// Why the matrix cores (same FLOP rate as the vector ALUs on gfx950, MI355X_MICROARCH.md): an A x A accumulator tile of ONE
// bin is spread over the 64 lanes of a wave (a 16 x 16 tile: 4 registers per lane), where the vector kernels keep the
// accumulators of 64 bins x one block of 8 x 8 antennas per wave -- 128 registers per lane -- and so have to re-read every
// spectrum row once per pair of antenna blocks ((A / 8 + 1) / 2 times: xengine_block_kernel, 3.5 x / 4.5 x / 6.5 x the
// algorithmic traffic at 16 / 32 / 64 antennas).  Here a workgroup owns a column of kCH = 16 bins (128-byte row segments)
// with ALL antennas and reads every spectrum sample once.
//
// Work: workgroup = (column of 16 bins, raw row = group of cg chunks x frame range, as x_range of xengine_kernel).
//   load     tiles of kFT frames x A antennas x 16 bins go HBM -> registers (16-byte loads, issued two tiles ahead: 64 KiB in
//            flight per workgroup) -> one of two LDS tiles, rows of 17 complex (136 B: 16 bins + 1 pad);
//            one s_barrier per tile; a frame's address is wave-uniform (two counters walk chunk and frame, no division)
//   multiply wave w owns kCPW of the 16 bins.  v_mfma_f32_16x16x4_f32, K = (Re, Im) of 2 frames:
//              A operand of antenna tile t (16 antennas): lane l holds part (l >> 4) & 1 of frame (l >> 5) of antenna l & 15
//              B operand = the A operand of the other tile:      sum_K = Re_a Re_b + Im_a Im_b          -> Re V[a][b]
//              B' = the other part, negated where it is Im:      sum_K = Im_a Re_b - Re_a Im_b          -> Im V[a][b]
//            (lane maps: cdna_hip_programming.md 'FP32-input MFMA': A[l & 15][k = l >> 4], B[k = l >> 4][l & 15],
//            D col = l & 15, row = 4 (l >> 4) + reg).  Tile pairs ti <= tj only: T (T + 1) / 2 of the T^2 tiles.
//            LDS reads: one ds_read_b32 per operand, bank = 2 m + part (+ 2 bin): conflict-free for each half-wave.
//   store    at the row's end the tiles go through LDS ([a][b][bin], pitch 17) and leave as 128-byte runs:
//            raw[(row * n_base + p(a, b)) * nchan + bin], a < b < A, baselines ordered as in xengine_kernel.
// Antennas past A up to 16 T are zero rows in LDS.  Frames past the end of a row are zero rows too.
// Float32 sums: a raw row is at most kRowSpectra spectra (the launcher's cg / n_ranges), one fmaf chain per element.
// ------------------------------------------------------------------------------------------
typedef float v4f32 __attribute__((ext_vector_type(4)));

template <int T>
struct XMfmaGeo {
    static constexpr int kT = T;                         // antenna tiles of 16
    static constexpr int kAP = 16 * T;                   // antennas, padded
#ifndef FXC_XMFMA_W1
#define FXC_XMFMA_W1 4
#define FXC_XMFMA_W2 4
#define FXC_XMFMA_W3 8
#define FXC_XMFMA_W4 8
#endif
    static constexpr int kWaves = T == 1 ? FXC_XMFMA_W1 : (T == 2 ? FXC_XMFMA_W2 : (T == 3 ? FXC_XMFMA_W3 : FXC_XMFMA_W4));
    static constexpr int kThreads = 64 * kWaves;
#ifndef FXC_XMFMA_CH34
#define FXC_XMFMA_CH34 16
#endif
    static constexpr int kCH = T <= 2 ? 16 : FXC_XMFMA_CH34;   // bins per workgroup: 128-byte row segments
    static constexpr int kSegLanes = kCH / 2;             // lanes per row segment (16 bytes each)
    static constexpr int kCPW = kCH / kWaves;            // bins per wave: 4 (T <= 2) or 2
    static constexpr int kPairs = T * (T + 1) / 2;
    static constexpr int kRowsPerPass = kThreads / kSegLanes;   // rows per load instruction
    static constexpr int kFPPmax = kRowsPerPass / kAP;   // whole frames per pass (48 antennas: rows idle)
    static constexpr int kFPP = kFPPmax >= 4 ? 4 : (kFPPmax >= 2 ? 2 : 1);
#ifndef FXC_XMFMA_FT1
#define FXC_XMFMA_FT1 8
#define FXC_XMFMA_FT2 4
#endif
    // frames per tile (even); two tiles in LDS.  Small tiles: 34 KiB of LDS per workgroup, so that three or four workgroups
    // share a CU (the waves wait a lot -- LDS operands, parked loads, the barrier -- and only other waves fill the matrix pipe)
    static constexpr int kFT = T == 1 ? FXC_XMFMA_FT1 : (T == 2 ? FXC_XMFMA_FT2 : 4);
    static_assert(kFT % kFPP == 0 && kFT % 4 == 0, "whole passes per tile, two halves of whole frame pairs");
    static constexpr int kPasses = kFT / kFPP;           // loads per thread and tile
    static constexpr int kRows = kFT * kAP;              // LDS rows of a tile
    static constexpr int kPitch = kCH + 1;               // complex per LDS row
    static constexpr int kTileCf = kRows * kPitch;       // complex per tile buffer
    static_assert(kFPP * kAP <= kRowsPerPass, "a pass covers whole frames");
    static constexpr int kLdsBytes = (2 * kTileCf > 256 * kPitch ? 2 * kTileCf : 256 * kPitch) * (int)sizeof(cf);   // two tiles, or the epilogue's [16][16][bin]
};

#ifndef FXC_XMFMA_LB34
#define FXC_XMFMA_LB34 2
#endif
template <int T>
__global__ __launch_bounds__(XMfmaGeo<T>::kThreads, (T == 1 ? 4 : (T == 2 ? 3 : FXC_XMFMA_LB34))) void xengine_mfma_kernel(const cf* __restrict__ spec, cf* __restrict__ raw,
                                                                               int64_t n_pts, int nchan, int64_t n_chunks, int cg,
                                                                               int A, int n_ranges, int abl) {
    using G = XMfmaGeo<T>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cf* tiles = reinterpret_cast<cf*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col0 = blockIdx.x * G::kCH;
    const XRange xr = x_range(n_pts, n_chunks, cg, n_ranges);
    const int64_t c0 = xr.grp * cg;
    const int64_t c_end = (xr.grp + 1) * cg < n_chunks ? (xr.grp + 1) * cg : n_chunks;
    const int nfr = (int)(xr.i1 - xr.i0);                      // frames per chunk in this row
    const int nf = nfr * (int)(c_end - c0);                    // frames of the row
    const int n_tiles = (nf + G::kFT - 1) / G::kFT;

    // ---- load side: thread -> (frame of the pass, antenna, 16-byte piece of the 128-byte segment).  Everything about a frame
    // is wave-uniform (scalar registers): the walk over (chunk, frame in chunk) is two counters, no division.
    const int lrow = tid / G::kSegLanes, lsub = tid % G::kSegLanes;
    const int la = lrow % G::kAP, lf = lrow / G::kAP;          // lf < kFPP for the rows that load
    const bool loads = lf < G::kFPP && la < A;
    // complex, inside a frame's A rows of nchan bins.  (Measured and dropped: spectra as [bin >> 4][antenna][bin & 15], a
    // column's A segments of 128 bytes side by side -- the same kernel time: the 128-byte segments are not what holds it.)
    const unsigned voff = (unsigned)(la * nchan + col0 + 2 * lsub);
    const int64_t frame_cf = (int64_t)A * nchan;               // complex per frame (all antennas)
    const cf* walk_base = spec + (c0 * n_pts + xr.i0) * frame_cf;            // frame 0 of the row's first chunk
    const int64_t chunk_skip = (n_pts - nfr) * frame_cf;       // from the last frame of a chunk's range to the next chunk's first
    int walk_q = 0, walk_i = 0;                                // next frame to fetch: index in the row, index in its chunk
    // two tiles in flight per workgroup (Little's law: one tile of 32 KiB per workgroup and 4 us of loaded HBM latency held
    // the kernel at 4 TB/s): tile k is fetched into stage[k & 1] at the end of iteration k - 3 and parked in iteration k - 1
    v4f32 stage[2][G::kPasses];
    typedef const __attribute__((address_space(1))) v4f32* gptr_t;      // global, not flat: a flat load also counts as an LDS
                                                                        // operation, and every wait for LDS operands would wait for HBM
    auto fetch = [&](v4f32 (&st)[G::kPasses]) {                // the next kFT frames -> registers (zeros where there is nothing)
        // a tile that stays inside its chunk -- all but one in nfr / kFT -- takes its frames at fixed strides
        if (walk_i + G::kFT < nfr && walk_q + G::kFT <= nf) {
            const cf* const tile_base = walk_base + voff;
            walk_q += G::kFT;
            walk_i += G::kFT;
            walk_base += G::kFT * frame_cf;
#pragma unroll
            for (int p = 0; p < G::kPasses; ++p) {
                v4f32 v = {0.f, 0.f, 0.f, 0.f};
                if (loads) v = *(gptr_t)(tile_base + (p * G::kFPP + lf) * frame_cf);
                st[p] = v;
            }
            return;
        }
#pragma unroll
        for (int p = 0; p < G::kPasses; ++p) {
            const cf* fb[G::kFPP];
            bool have[G::kFPP];
#pragma unroll
            for (int f = 0; f < G::kFPP; ++f) {                // uniform
                fb[f] = walk_base;
                have[f] = walk_q < nf;
                ++walk_q;
                walk_base += frame_cf;
                if (++walk_i == nfr) {
                    walk_i = 0;
                    walk_base += chunk_skip;
                }
            }
            const cf* src = fb[0];
            bool ok = have[0];
#pragma unroll
            for (int f = 1; f < G::kFPP; ++f)
                if (lf == f) {
                    src = fb[f];
                    ok = have[f];
                }
            ok = ok && loads;
            v4f32 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *(gptr_t)(src + voff);
            st[p] = v;
        }
    };
    auto park = [&](const v4f32 (&st)[G::kPasses], cf* tile) {  // registers -> LDS (idle rows of a pass write nothing)
        if (lf < G::kFPP) {
#pragma unroll
            for (int p = 0; p < G::kPasses; ++p) {
                cf* dst = tile + ((p * G::kFPP + lf) * G::kAP + la) * G::kPitch + 2 * lsub;
                dst[0] = fxc::mk(st[p][0], st[p][1]);
                dst[1] = fxc::mk(st[p][2], st[p][3]);
            }
        }
    };

    // ---- multiply side: lane -> (antenna m of a tile, k = (frame of the pair, part))
    const int m = lane & 15, kk = lane >> 4, kf = kk >> 1, part = kk & 1;
    const int op_off = ((kf * G::kAP + m) * G::kPitch) * 2 + (wave * G::kCPW) * 2;     // floats into a tile
    const unsigned flip = part == 0 ? 0x80000000u : 0u;        // B' = (-Im, Re): the Im that lands on an even k is negated
    v4f32 cre[G::kCPW][G::kPairs], cim[G::kCPW][G::kPairs];
#pragma unroll
    for (int cc = 0; cc < G::kCPW; ++cc)
#pragma unroll
        for (int pr = 0; pr < G::kPairs; ++pr) cre[cc][pr] = cim[cc][pr] = v4f32{0.f, 0.f, 0.f, 0.f};
    // (reading the operands of half a tile in one go, ahead of its multiplies, was measured: 16 .. 32 more registers, one
    // workgroup fewer per CU at 32 antennas, 6 .. 13 % slower: the other waves of the SIMD hide the LDS latency better)
    auto multiply = [&](const cf* tile, int fp_lo, int fp_hi) {
        const float* base = reinterpret_cast<const float*>(tile) + op_off;
#pragma unroll
        for (int fp = fp_lo; fp < fp_hi; ++fp) {
#pragma unroll
            for (int cc = 0; cc < G::kCPW; ++cc) {
                float a_op[T], b2_op[T];
#pragma unroll
                for (int tt = 0; tt < T; ++tt) {
                    const float* q = base + (fp * 2 * G::kAP + tt * 16) * G::kPitch * 2 + cc * 2;
                    a_op[tt] = q[part];
                    b2_op[tt] = __uint_as_float(__float_as_uint(q[1 - part]) ^ flip);
                }
                int pr = 0;
#pragma unroll
                for (int ti = 0; ti < T; ++ti)
#pragma unroll
                    for (int tj = ti; tj < T; ++tj, ++pr) {
                        cre[cc][pr] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_op[ti], a_op[tj], cre[cc][pr], 0, 0, 0);
                        cim[cc][pr] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_op[ti], b2_op[tj], cim[cc][pr], 0, 0, 0);
                    }
            }
        }
    };

    // (padding rows -- antennas A .. kAP - 1 -- and frames past the row's end are parked as zeros with every tile)
    cf* const buf0 = tiles;
    cf* const buf1 = tiles + G::kTileCf;
    fetch(stage[0]);                                           // tile 0
    park(stage[0], buf0);
    fetch(stage[1]);                                           // tile 1
    fetch(stage[0]);                                           // tile 2
    __syncthreads();
    // iteration t: multiply tile t (in buf[t & 1]), park tile t + 1 half-way (its loads were issued two tiles ago), one
    // barrier, fetch tile t + 3.  Unrolled by two so that the staging registers are named statically.  Tile k waits in
    // stage[k & 1]; fetches past the last tile load nothing (zeros).
    // Timing ablations (developer knob FXC_XMFMA_ABL, wrong results): bits 2 no multiplies, 4 no fetches, 8 no park / barrier.
    const bool abl_mul = abl & 2, abl_fetch = abl & 4, abl_park = abl & 8;
#define FXC_XMFMA_STEP(CUR, NXT, ST)                             \
    if (!abl_mul) multiply(CUR, 0, G::kFT / 4);                  \
    if (!abl_park) park(ST, NXT);                                \
    if (!abl_mul) multiply(CUR, G::kFT / 4, G::kFT / 2);         \
    if (!abl_park) __syncthreads();                              \
    if (!abl_fetch) fetch(ST);
    for (int t = 0; t < n_tiles; t += 2) {
        FXC_XMFMA_STEP(buf0, buf1, stage[1])
        if (t + 1 >= n_tiles) break;
        FXC_XMFMA_STEP(buf1, buf0, stage[0])
    }
#undef FXC_XMFMA_STEP

    // epilogue: tile pair by tile pair through LDS as [a][b][bin]; rows of 16 bins leave as 128-byte runs
    const int64_t n_base = (int64_t)A * (A - 1) / 2;
    int pr = 0;
#pragma unroll
    for (int ti = 0; ti < T; ++ti)
#pragma unroll
        for (int tj = ti; tj < T; ++tj, ++pr) {
            if (16 * tj >= A) continue;                        // (uniform) nothing but padding in this tile pair
            __syncthreads();
            cf* tile = tiles;
#pragma unroll
            for (int cc = 0; cc < G::kCPW; ++cc)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int e = (4 * (lane >> 4) + j) * 16 + (lane & 15);          // D row (antenna a), column (antenna b)
                    tile[e * G::kPitch + wave * G::kCPW + cc] = fxc::mk(cre[cc][pr][j], cim[cc][pr][j]);
                }
            __syncthreads();
            for (int e = lrow; e < 256; e += G::kRowsPerPass) {
                const int a = 16 * ti + (e >> 4), b = 16 * tj + (e & 15);
                if (a < b && b < A) {
                    const int64_t p = (int64_t)a * A - (int64_t)a * (a + 1) / 2 + (b - a - 1);
                    const cf* src = tile + e * G::kPitch + 2 * lsub;
                    const v4f32 v = {src[0].x, src[0].y, src[1].x, src[1].y};
                    *reinterpret_cast<v4f32*>(raw + (xr.row * n_base + p) * (int64_t)nchan + col0 + 2 * lsub) = v;
                }
            }
        }
}

}  // namespace
