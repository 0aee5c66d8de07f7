// Host emulation of the fused kernel's work split — TEST INFRASTRUCTURE ONLY.
// Walks every workgroup's two parts (whole chunks dealt round-robin in segments, then its frame range of the tail)
// with the kernel's own RangeWalk (effex_amd/csrc/fx_fused4096.h), keeps the frame ring as frame ids, and rebuilds
// per-chunk and total sums the way the finishing kernels do (fxcorr.hip::add_lead_rows, fused_reduce1_kernel).  The
// "spectrum product" of frame (c, i) is the weight w(c, i) passed in, so every frame must be visited exactly once
// and land in the right row.
#include <cmath>
#include <cstdint>
#include <vector>

#include "../../effex_amd/csrc/fx_fused4096.h"

using namespace fxc;
using namespace fxc::fused;

// returns 0, or a negative code for: -1 ring history wrong, -2 row written twice, -3 row never written,
// -4 sums left in registers at the end of a part, -5 frame visited twice or never
extern "C" int emul_fused_schedule(int n_chunks, int n_pts, int grid, int seg, int unit, int rows_are_chunks,
                                   const double* weight, double* per_chunk /* [n_chunks], rows_are_chunks only */,
                                   double* total, int64_t* frames_min, int64_t* frames_max, int64_t* n_rows_out) {
    const RangeSplit sp = range_split(grid, n_chunks, seg, unit, rows_are_chunks != 0);
    const double unset = std::nan("");
    std::vector<double> rows((size_t)sp.n_rows, unset);
    std::vector<int> visits((size_t)n_chunks * n_pts, 0);
    *frames_min = INT64_MAX;
    *frames_max = 0;
    *n_rows_out = sp.n_rows;
    for (int b = 0; b < grid; ++b) {
        int64_t frames_b = 0;
        for (int part = 0; part < 2; ++part) {
            RangeWalk pos = part == 0 ? range_walk_rounds(b, grid, n_chunks, n_pts, seg, unit, rows_are_chunks != 0)
                                      : range_walk_tail(b, grid, n_chunks, n_pts, seg, unit, rows_are_chunks != 0);
            const int total_frames = pos.left;
            frames_b += total_frames;
            if (part == 1 && (!pos.lead || total_frames == 0)) {
                const size_t lead_row = (size_t)(sp.rows_rounds + sp.n_tail + b);
                if (!std::isnan(rows[lead_row])) return -2;
                rows[lead_row] = 0.0;
            }
            if (total_frames == 0) continue;
            // ring of frame ids: (chunk << 32 | frame) or -1 for zeros; prologue as in the kernel
            int64_t ring[4];
            for (int d = 1; d < 4; ++d) ring[4 - d] = pos.i - d >= 0 ? (((int64_t)pos.c << 32) | (pos.i - d)) : -1;
            ring[0] = ((int64_t)pos.c << 32) | pos.i;
            double acc = 0.0;
            for (int g = 0; g < total_frames; ++g) {
                const int ph = g & 3;
                const int64_t c = pos.c, i = pos.i;
                if (c < 0 || c >= n_chunks || i < 0 || i >= n_pts) return -5;
                if (i == 0)
                    for (int d = 1; d < 4; ++d) ring[(ph + d) & 3] = -1;      // state_reset_history
                if (ring[ph] != ((c << 32) | i)) return -1;
                for (int t = 1; t < 4; ++t) {                                   // tap t reads slot (ph + 4 - t) & 3
                    const int64_t want = i - t >= 0 ? ((c << 32) | (i - t)) : -1;
                    if (ring[(ph + 4 - t) & 3] != want) return -1;
                }
                int pc, pi;
                range_walk_prefetch(pos, pc, pi);
                ring[(ph + 1) & 3] = ((int64_t)pc << 32) | pi;
                visits[(size_t)(c * n_pts + i)] += 1;
                acc += weight[c * n_pts + i];
                const bool row_ends = range_walk_row_ends(pos);
                if (row_ends) {
                    if (pos.row < 0 || pos.row >= sp.n_rows) return -3;
                    if (!std::isnan(rows[(size_t)pos.row])) return -2;
                    rows[(size_t)pos.row] = acc;
                    acc = 0.0;
                }
                range_walk_advance(pos, row_ends);
            }
            if (acc != 0.0) return -4;
        }
        if (frames_b < *frames_min) *frames_min = frames_b;
        if (frames_b > *frames_max) *frames_max = frames_b;
    }
    for (int v : visits)
        if (v != 1) return -5;
    double t = 0.0;
    for (double r : rows) {
        if (std::isnan(r)) return -3;
        t += r;
    }
    *total = t;
    if (rows_are_chunks && per_chunk) {
        const int n_frames = sp.n_tail * n_pts;
        for (int c = 0; c < n_chunks; ++c) {
            double v = rows[(size_t)c];
            if (c >= sp.n_full && n_frames > 0) {           // add_lead_rows
                const int tc = c - sp.n_full;
                const int b_lo = range_owner(tc * n_pts, n_frames, grid);
                const int b_hi = range_owner((tc + 1) * n_pts - 1, n_frames, grid);
                for (int b = b_lo + 1; b <= b_hi; ++b) v += rows[(size_t)(n_chunks + b)];
            }
            per_chunk[c] = v;
        }
    }
    return 0;
}
