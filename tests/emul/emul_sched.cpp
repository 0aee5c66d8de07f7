// Host emulation of the fused kernel's work split — TEST INFRASTRUCTURE ONLY.
// Walks every workgroup's frame range with the kernel's own RangeWalk (effex_amd/csrc/fx_fused4096.h), keeps the
// frame ring as frame ids, and rebuilds per-chunk and total sums the way the finishing kernels do
// (fxcorr.hip::add_lead_rows, fused_reduce1_kernel).  The "spectrum product" of frame (c, i) is the weight
// w(c, i) passed in, so every frame must be visited exactly once and land in the right row.
#include <cmath>
#include <cstdint>
#include <vector>

#include "../../effex_amd/csrc/fx_fused4096.h"

using namespace fxc;
using namespace fxc::fused;

// returns 0, or a negative code for: -1 ring history wrong, -2 row written twice, -3 row read before written
extern "C" int emul_fused_schedule(int64_t n_chunks, int64_t n_pts, int grid, int64_t unit, const double* weight,
                                   double* per_chunk /* [n_chunks], unit == 1 only */, double* total,
                                   int64_t* frames_min, int64_t* frames_max) {
    const int64_t n_rows = (n_chunks + unit - 1) / unit;
    const double unset = std::nan("");
    std::vector<double> rows((size_t)(n_rows + grid), unset);
    *frames_min = INT64_MAX;
    *frames_max = 0;
    for (int b = 0; b < grid; ++b) {
        RangeWalk pos = range_walk_init(b, grid, n_chunks, n_pts, unit);
        const int64_t total_frames = pos.left;
        if (total_frames < *frames_min) *frames_min = total_frames;
        if (total_frames > *frames_max) *frames_max = total_frames;
        if (!pos.lead || total_frames == 0) {
            if (!std::isnan(rows[(size_t)(pos.n_rows + b)])) return -2;
            rows[(size_t)(pos.n_rows + b)] = 0.0;
        }
        if (total_frames == 0) continue;
        // ring of frame ids: (chunk << 32 | frame) or -1 for zeros; prologue as in the kernel
        int64_t ring[4];
        for (int d = 1; d < 4; ++d) ring[4 - d] = pos.i - d >= 0 ? ((pos.c << 32) | (pos.i - d)) : -1;
        ring[0] = (pos.c << 32) | pos.i;
        double acc = 0.0;
        for (int64_t g = 0; g < total_frames; ++g) {
            const int ph = (int)(g & 3);
            const int64_t c = pos.c, i = pos.i;
            if (i == 0)
                for (int d = 1; d < 4; ++d) ring[(ph + d) & 3] = -1;      // state_reset_history
            if (ring[ph] != ((c << 32) | i)) return -1;
            for (int t = 1; t < 4; ++t) {                                   // tap t reads slot (ph + 4 - t) & 3
                const int64_t want = i - t >= 0 ? ((c << 32) | (i - t)) : -1;
                if (ring[(ph + 4 - t) & 3] != want) return -1;
            }
            long long pc, pi;
            range_walk_prefetch(pos, pc, pi);
            ring[(ph + 1) & 3] = ((int64_t)pc << 32) | pi;
            acc += weight[c * n_pts + i];
            const bool row_ends = range_walk_row_ends(pos);
            if (row_ends) {
                if (!std::isnan(rows[(size_t)pos.row])) return -2;
                rows[(size_t)pos.row] = acc;
                acc = 0.0;
            }
            range_walk_advance(pos, row_ends);
        }
        if (acc != 0.0) return -4;     // sums left in registers at the end of the range
    }
    double t = 0.0;
    for (double r : rows) {
        if (std::isnan(r)) return -3;
        t += r;
    }
    *total = t;
    if (unit == 1 && per_chunk) {
        const int64_t n_frames = n_chunks * n_pts;
        for (int64_t c = 0; c < n_chunks; ++c) {
            double v = rows[(size_t)c];
            const int64_t b_lo = range_owner(c * n_pts, n_frames, grid);
            const int64_t b_hi = range_owner((c + 1) * n_pts - 1, n_frames, grid);
            for (int64_t b = b_lo + 1; b <= b_hi; ++b) v += rows[(size_t)(n_rows + b)];
            per_chunk[c] = v;
        }
    }
    return 0;
}
