// Part of libfxcorr's single translation unit: included by fxcorr.hip (not a stand-alone header).
#pragma once

namespace {

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
        const int64_t part_bytes = fold_part_bytes(p);
        int rc = ensure_ws(p, spec_bytes + raw_bytes + part_bytes);
        if (rc) return rc;
        cf* spec = reinterpret_cast<cf*>(p->d_ws);
        cf* raw = reinterpret_cast<cf*>(static_cast<char*>(p->d_ws) + spec_bytes);
        cd* part = reinterpret_cast<cd*>(static_cast<char*>(p->d_ws) + spec_bytes + raw_bytes);
        for (int64_t c0 = 0; c0 < n_chunks; c0 += cb) {
            const int64_t nc = std::min(cb, n_chunks - c0);
            // chunks per raw row: 2 antennas, rows of up to kRowSpectra spectra; more, the X-engine's chunk groups
            const int64_t unit = p->n_ant == 2 ? fused_unit(p) : xengine_group(p, nc, fused_unit(p));
            rc = fused_raw_sums(p, reinterpret_cast<const cf*>(reinterpret_cast<const char*>(x) + c0 * in_bytes), nc, spec, raw,
                                dc_u8 ? dc_u8 + c0 * 2 : nullptr, unit, false, dc_u8 && p->u8_dck);
            if (rc) return rc;
            // 2 antennas: all the raw rows, leading parts included; more: one row [n_base][nchan] per chunk group
            const int64_t n_rows = p->n_ant == 2 ? fused_rows(p, nc, unit, false) : (nc + unit - 1) / unit * x_ranges(p, unit);
            rc = fold_or_defer(p, raw, part, n_rows, fused_layout(p), c0 + nc >= n_chunks);
            if (rc) return rc;
        }
    } else if (p->split8192 && !dc_u8) {
        const int N = p->nchan;
        const int64_t cb = split_chunks_per_pass(p, n_chunks);
        const int64_t row_bytes = (int64_t)fxc::fused::kN * (int64_t)sizeof(cf);
        const int64_t raw_bytes = ((2 * cb + p->fused_grid_max) * row_bytes + 255) / 256 * 256;
        const int64_t part_bytes = fold_part_bytes(p);
        int rc = ensure_ws(p, raw_bytes + part_bytes);
        if (rc) return rc;
        cf* raw = reinterpret_cast<cf*>(p->d_ws);
        cd* part = reinterpret_cast<cd*>(static_cast<char*>(p->d_ws) + raw_bytes);
        for (int64_t c0 = 0; c0 < n_chunks; c0 += cb) {
            const int64_t nc = std::min(cb, n_chunks - c0);
            rc = split_raw_sums(p, x + c0 * 2 * p->num_samp, nc, raw);
            if (rc) return rc;
            // the nc pairs of 4096-rows are nc rows of 8192 in layout 3; the leading-part rows are added by parity
            rc = fold_rows(p, raw, part, nc, 3, kNoFinish);
            if (rc) return rc;
            hipLaunchKernelGGL(split_lead_acc_kernel, dim3(N / 256), dim3(256), 0, p->stream, raw, p->d_acc, fused_lead(p, 2 * nc));
            FXC_HIP(p, hipGetLastError());
        }
    } else if (use_tiled(p, n_chunks)) {
        const int N = p->nchan;
        const int64_t in_bytes = (int64_t)2 * p->num_samp * (dc_u8 ? 2 : (int64_t)sizeof(cf));   // per chunk
        const int n_splits = tiled_splits(p, n_chunks);
        const int64_t row_bytes = (int64_t)N * (int64_t)sizeof(cf);
        const int64_t cb = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(n_chunks, tiled_streams_per_pass(p) / 2),
                                                                  ws_target() / (row_bytes * n_splits)));
        const int64_t raw_bytes = (cb * n_splits * row_bytes + 255) / 256 * 256;
        const int64_t part_bytes = fold_part_bytes(p);
        int rc = ensure_ws(p, raw_bytes + part_bytes);
        if (rc) return rc;
        cf* raw = reinterpret_cast<cf*>(p->d_ws);
        cd* part = reinterpret_cast<cd*>(static_cast<char*>(p->d_ws) + raw_bytes);
        for (int64_t c0 = 0; c0 < n_chunks; c0 += cb) {
            const int64_t nc = std::min(cb, n_chunks - c0);
            rc = tiled_raw_sums(p, reinterpret_cast<const cf*>(reinterpret_cast<const char*>(x) + c0 * in_bytes), nc, n_splits,
                                raw, dc_u8 ? dc_u8 + c0 * 2 : nullptr);
            if (rc) return rc;
            rc = fold_or_defer(p, raw, part, nc * n_splits, 0, c0 + nc >= n_chunks);
            if (rc) return rc;
        }
    } else {
        const bool xf = mixed_one_pass(p, dc_u8 != nullptr);
        const bool xm = !xf && two_pass_xm(p, dc_u8 != nullptr);
        const XGeom g = x_geometry(p, n_chunks, xf, xm);
        int64_t spec_bytes, raw_bytes;
        const int64_t cb = generic_chunks_per_pass(p, n_chunks, g, &spec_bytes, &raw_bytes, xf, xm);
        raw_bytes = (raw_bytes + 255) / 256 * 256;
        int rc = ensure_ws(p, spec_bytes + raw_bytes + fold_part_bytes(p));
        if (rc) return rc;
        cf* spec = reinterpret_cast<cf*>(p->d_ws);
        cf* raw = reinterpret_cast<cf*>(static_cast<char*>(p->d_ws) + spec_bytes);
        cd* part = reinterpret_cast<cd*>(static_cast<char*>(p->d_ws) + spec_bytes + raw_bytes);
        for (int64_t c0 = 0; c0 < n_chunks; c0 += cb) {
            const int64_t nc = std::min(cb, n_chunks - c0);
            KernelTimer kt(p);
            if (xf) {
                // (bytes in: two per sample, and the offsets of this pass's streams)
                rc = dc_u8 ? mixed_fx_raw_sums(p, reinterpret_cast<const cf*>(reinterpret_cast<const char*>(x) + c0 * 2 * p->num_samp * 2),
                                               nc, g.n_splits, raw, dc_u8 + c0 * 2)
                           : mixed_fx_raw_sums(p, x + c0 * p->n_ant * p->num_samp, nc, g.n_splits, raw);
                if (rc) return rc;
            } else if (xm) {
                rc = two_pass_raw_sums(p, x + c0 * p->n_ant * p->num_samp, nc, g.n_splits, spec, raw);
                if (rc) return rc;
            } else if (p->mixed_xeng) {
                // 3 .. 64 antennas off the powers of two: spectra antenna-interleaved, then the X-engines of the tiled paths
                rc = run_channelize(p, x + c0 * p->n_ant * p->num_samp, spec, nc * p->n_ant, p->n_ant);
                if (rc) return rc;
                rc = launch_xengine(p, spec, raw, nc, 1, g.n_splits);
                if (rc) return rc;
            } else {
                rc = run_channelize(p, x + c0 * p->n_ant * p->num_samp, spec, nc * p->n_ant);
                if (rc) return rc;
                const int kblocks = (p->nchan + g.kx - 1) / g.kx;
                const int64_t wgs = nc * p->n_base * kblocks * g.n_splits;
                hipLaunchKernelGGL(xmul_kernel, dim3((int)std::min<int64_t>(wgs, (int64_t)p->cu_count * 8)), dim3(256), 0,
                                   p->stream, spec, raw, p->n_ant, p->n_base, p->nchan, p->n_pts, g.kx, g.n_splits, nc);
            }
            kt.stop();
            FXC_HIP(p, hipGetLastError());
            // raw[split][chunk] = nc * n_splits rows of [n_base][nchan], natural bin order
            rc = fold_or_defer(p, raw, part, nc * g.n_splits, 0, c0 + nc >= n_chunks);
            if (rc) return rc;
        }
    }
    p->spectra_count += (double)n_chunks * (double)p->n_pts;
    return FXC_OK;
}

// CONTINUUM rows: one workgroup per row when there are rows enough to fill the chip, else bin slices + a second small kernel
int launch_rows_continuum(fxc_plan* p, const cf* raw, cd* out, int nchan, int64_t rows, int n_splits, int64_t split_stride,
                          double scale, int slots, LeadRows lead) {
    const int slices = (int)std::min<int64_t>(32, nchan / 128);
    if (slices >= 2 && rows * 2 <= p->cu_count && rows <= 65535) {
        const int rg = grow(p, &p->d_rowpart, &p->rowpart_bytes, (size_t)rows * slices * sizeof(cd));
        if (rg) return rg;
        cd* part = static_cast<cd*>(p->d_rowpart);
        hipLaunchKernelGGL(rows_continuum_part_kernel, dim3(slices, (unsigned)rows), dim3(256), 0, p->stream, raw, part, p->d_rot, nchan,
                           rows, n_splits, split_stride, slots, lead, slices);
        hipLaunchKernelGGL(rows_continuum_fin_kernel, dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, p->stream, part, out, rows, slices,
                           scale);
    } else {
        hipLaunchKernelGGL(rows_continuum_kernel, dim3((int)std::min<int64_t>(rows, (int64_t)p->cu_count * 8)),
                           dim3(continuum_threads(nchan)), 0, p->stream, raw, out, p->d_rot, nchan, rows, n_splits, split_stride, scale,
                           slots, lead);
    }
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
            else {
                rc = launch_rows_continuum(p, raw, static_cast<cd*>(out) + c0, 1, nc, nb, nc, cscale, 0, kNoLead);
                if (rc) return rc;
            }
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
                                dc_u8 ? dc_u8 + c0 * 2 : nullptr, 1, true, dc_u8 && p->u8_dck);
            if (rc) return rc;
            const int64_t rows = nc * p->n_base;
            const LeadRows lead = p->n_ant == 2 ? fused_lead(p, nc) : kNoLead;
            // 3 and more antennas: the frame ranges of a chunk are the rows kernels' splits (range-major raw rows)
            const int xr = x_ranges(p, 1);
            const int64_t xr_stride = rows * p->nchan;
            if (mode == FXC_MODE_SPECTRUM)
                hipLaunchKernelGGL(rows_spectrum_kernel, dim3(grid_for(rows * p->nchan, 256, p->cu_count)), dim3(256), 0,
                                   p->stream, raw, static_cast<cf*>(out) + c0 * p->n_base * p->nchan, p->d_rot, p->nchan,
                                   rows, xr, xr_stride, inv_pts, fused_layout(p), lead);
            else {
                rc = launch_rows_continuum(p, raw, static_cast<cd*>(out) + c0 * p->n_base, p->nchan, rows, xr, xr_stride, cscale, fused_layout(p), lead);
                if (rc) return rc;
            }
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
            else {
                rc = launch_rows_continuum(p, raw, static_cast<cd*>(out) + c0, N, nc, 1, (int64_t)0, cscale, 3, lead);
                if (rc) return rc;
            }
            FXC_HIP(p, hipGetLastError());
        }
        return FXC_OK;
    }
    if (use_tiled(p, n_chunks)) {
        const int N = p->nchan;
        const int64_t in_bytes = (int64_t)2 * p->num_samp * (dc_u8 ? 2 : (int64_t)sizeof(cf));   // per chunk
        const int n_splits = tiled_splits(p, n_chunks);
        const int64_t row_bytes = (int64_t)N * (int64_t)sizeof(cf);
        const int64_t cb = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(n_chunks, tiled_streams_per_pass(p) / 2),
                                                                  ws_target() / (row_bytes * n_splits)));
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
            else {
                rc = launch_rows_continuum(p, raw, static_cast<cd*>(out) + c0, N, nc, n_splits, nc * N, cscale, 0, kNoLead);
                if (rc) return rc;
            }
            FXC_HIP(p, hipGetLastError());
        }
        return FXC_OK;
    }
    const bool xf = mixed_one_pass(p, dc_u8 != nullptr);
    const bool xm = !xf && two_pass_xm(p, dc_u8 != nullptr);
    const XGeom g = x_geometry(p, n_chunks, xf, xm);
    int64_t spec_bytes, raw_bytes;
    const int64_t cb = generic_chunks_per_pass(p, n_chunks, g, &spec_bytes, &raw_bytes, xf, xm);
    int rc = ensure_ws(p, spec_bytes + raw_bytes);
    if (rc) return rc;
    cf* spec = reinterpret_cast<cf*>(p->d_ws);
    cf* raw = reinterpret_cast<cf*>(static_cast<char*>(p->d_ws) + spec_bytes);
    for (int64_t c0 = 0; c0 < n_chunks; c0 += cb) {
        const int64_t nc = std::min(cb, n_chunks - c0);
        KernelTimer kt(p);
        if (xf) {
            rc = dc_u8 ? mixed_fx_raw_sums(p, reinterpret_cast<const cf*>(reinterpret_cast<const char*>(x) + c0 * 2 * p->num_samp * 2),
                                           nc, g.n_splits, raw, dc_u8 + c0 * 2)
                       : mixed_fx_raw_sums(p, x + c0 * p->n_ant * p->num_samp, nc, g.n_splits, raw);
            if (rc) return rc;
        } else if (xm) {
            rc = two_pass_raw_sums(p, x + c0 * p->n_ant * p->num_samp, nc, g.n_splits, spec, raw);
            if (rc) return rc;
        } else if (p->mixed_xeng) {
            rc = run_channelize(p, x + c0 * p->n_ant * p->num_samp, spec, nc * p->n_ant, p->n_ant);
            if (rc) return rc;
            rc = launch_xengine(p, spec, raw, nc, 1, g.n_splits);
            if (rc) return rc;
        } else {
            rc = run_channelize(p, x + c0 * p->n_ant * p->num_samp, spec, nc * p->n_ant);
            if (rc) return rc;
            const int kblocks = (p->nchan + g.kx - 1) / g.kx;
            const int64_t wgs = nc * p->n_base * kblocks * g.n_splits;
            hipLaunchKernelGGL(xmul_kernel, dim3((int)std::min<int64_t>(wgs, (int64_t)p->cu_count * 8)), dim3(256), 0,
                               p->stream, spec, raw, p->n_ant, p->n_base, p->nchan, p->n_pts, g.kx, g.n_splits, nc);
        }
        kt.stop();
        const int64_t rows = nc * p->n_base;
        const int64_t split_stride = rows * p->nchan;
        if (mode == FXC_MODE_SPECTRUM)
            hipLaunchKernelGGL(rows_spectrum_kernel, dim3(grid_for(rows * p->nchan, 256, p->cu_count)), dim3(256), 0,
                               p->stream, raw, static_cast<cf*>(out) + c0 * p->n_base * p->nchan, p->d_rot, p->nchan,
                               rows, g.n_splits, split_stride, inv_pts, 0, kNoLead);
        else {
            rc = launch_rows_continuum(p, raw, static_cast<cd*>(out) + c0 * p->n_base, p->nchan, rows, g.n_splits, split_stride, cscale, 0, kNoLead);
            if (rc) return rc;
        }
        FXC_HIP(p, hipGetLastError());
    }
    return FXC_OK;
}

// host-buffer helper: stage in, run, stage out (synchronous)
// Pageable buffers go through the runtime's bounce buffers inside hipMemcpyAsync; buffers from fxc_host_alloc are pinned, so
// the same call is one DMA -- and an `out` inside such a block is handed to fn as the device's mapping of it: the finishing
// kernel writes the rows over PCIe itself (32 KiB per chunk pair) and no copy back is queued.
template <class Fn>
int with_host_staging(fxc_plan* p, const void* x, size_t x_bytes, void* out, size_t out_bytes, Fn fn) {
    // staging buffers live in the plan and only grow: the reference calls once per chunk pair (effex.py:490-494),
    // and a hipMalloc / hipFree pair per call costs more than the copy of one chunk
    void* out_mapped = out_bytes ? pinned_device_ptr(out, out_bytes) : nullptr;
    const size_t want[2] = {x_bytes ? x_bytes : 1, out_mapped ? 0 : out_bytes};
    for (int k = 0; k < 2; ++k) {
        const int rg = grow(p, &p->d_stage[k], &p->stage_bytes[k], want[k]);
        if (rg) return rg;
    }
    void* dx = p->d_stage[0];
    void* dout = out_bytes ? (out_mapped ? out_mapped : p->d_stage[1]) : nullptr;
    int rc = FXC_OK;
    // (a kernel fetching the pinned block over PCIe itself instead of the copy engine was measured: 0.149 against 0.133 ms per
    // reference-sized call, profiles/r04/experiments.md)
    hipError_t e = hipMemcpyAsync(dx, x, x_bytes, hipMemcpyHostToDevice, p->stream);
    if (e == hipSuccess) rc = fn(static_cast<const cf*>(dx), dout);
    if (e == hipSuccess && rc == FXC_OK && out_bytes && !out_mapped)
        e = hipMemcpyAsync(out, dout, out_bytes, hipMemcpyDeviceToHost, p->stream);
    hipError_t e2 = hipStreamSynchronize(p->stream);
    if (rc != FXC_OK) return rc;
    if (e != hipSuccess) return fail(p, FXC_ERR_HIP, "host staging copy failed: %s", hipGetErrorString(e));
    if (e2 != hipSuccess) return fail(p, FXC_ERR_HIP, "stream sync failed: %s", hipGetErrorString(e2));
    return FXC_OK;
}

}  // namespace
