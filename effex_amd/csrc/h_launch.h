// Part of libfxcorr's single translation unit: included by fxcorr.hip (not a stand-alone header).
#pragma once

// per-path launchers: which kernels a pass over a batch of chunks takes, and how the workspace is cut into passes

namespace {

int launch_fused(fxc_plan* p, const cf* x, int64_t n_pairs, cf* out, bool spec_out, const cf* dc_u8 = nullptr,
                 int64_t unit = 1, bool rows_are_chunks = true, int64_t num_samp = 0, bool dck = false);

// F-stage of `n_streams` streams: x -> spec (both device, natural bin order)
int tiled_channelize(fxc_plan* p, const cf* x, cf* spec, int64_t n_streams, int spec_a = 0);
int64_t spec_wg_splits(const fxc_plan* p, const SpecKernel* k, int64_t n_groups, bool sums);

// Slots of one wave (tpr <= 64) go without the workgroup barrier between their steps: measured on one box, two antennas
// 7 - 16 % faster at 8 ... 250 channels, F only 5 - 7 % faster at 96 ... 250 but 17 % slower at 12 (slots of 4 threads) -- so F
// only from 16 threads per slot.  Developer knob: FXC_MIXED_WAVELOCAL=0 keeps the barrier everywhere.
int mixed_wave_local(const fxc_plan* p, bool fused_x) {
    static const int v = FXC_DEV_ENV_INT("FXC_MIXED_WAVELOCAL", 1);
    return v && (fused_x || p->mixed_tpr >= 16);
}

// the F stage built for exactly this channel count (fx_spec.h, FXM_FONLY), compiled (or found) on first use; nullptr: the shape has none
const SpecKernel* spec_f_kernel(fxc_plan* p) {
    if (!p->spec_f_tried) {
        p->spec_f_tried = true;
        if (p->mixed && p->ntaps <= 4 && p->num_samp < (1ll << 28) && p->rtc &&
            !spec_first_radices(p->nchan, p->ntaps, spec_rows(p->nchan, kSpecFOnly)).empty()) {
            const SpecKernel* k = spec_kernel(p->device, p->nchan, p->ntaps, kSpecFOnly);
            p->spec_f = k->fn ? k : nullptr;
        }
    }
    return p->spec_f;
}

int run_channelize(fxc_plan* p, const cf* x, cf* spec, int64_t n_streams, int ant = 1) {
    if (n_streams == 0 || p->n_pts == 0) return FXC_OK;
    // the F-only tiled kernel writes natural-order spectra straight to `spec` (no workspace): any caller may use it
    if (p->tiled_f) return tiled_channelize(p, x, spec, n_streams);
    if (p->mixed && p->mixed_blu) {
        // a large prime factor: chirp-z rows of nfft = 2^j >= 2 nchan - 1, one row per workgroup pass
        const int threads = std::max(256, p->mixed_tpr);
        const int rpw = threads / p->mixed_tpr;
        const bool twl = p->blu_nfft <= 4096;
        const size_t lds = ((size_t)rpw * 2 + (twl ? 1 : 0)) * p->blu_nfft * sizeof(cf);
        const int64_t n_groups = n_streams * ((p->n_pts + rpw - 1) / rpw);
        const int64_t run = std::max<int64_t>(1, std::min<int64_t>(16, n_groups / ((int64_t)p->cu_count * 8)));
        const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((n_groups + run - 1) / run, 1 << 30));
        const MixedExtras blu = {mixed_wave_local(p, false), ant, p->blu_nfft, p->d_chirp, p->d_blud, nullptr};
        if (twl)
            hipLaunchKernelGGL((pfb_fft_mixed_kernel<true, 1, false, true>), dim3(grid), dim3(threads), lds, p->stream, x, p->d_win,
                               spec, p->d_tw, p->mixed_plan, p->num_samp, p->nchan, p->ntaps, p->n_pts, n_streams, p->mixed_tpr, 1, blu);
        else
            hipLaunchKernelGGL((pfb_fft_mixed_kernel<false, 1, false, true>), dim3(grid), dim3(threads), lds, p->stream, x, p->d_win,
                               spec, p->d_tw, p->mixed_plan, p->num_samp, p->nchan, p->ntaps, p->n_pts, n_streams, p->mixed_tpr, 1, blu);
        FXC_HIP(p, hipGetLastError());
        return FXC_OK;
    }
    if (p->mixed && p->ntaps <= 4 && p->num_samp < (1ll << 28) && p->rtc) {
        // the F stage built for exactly this channel count (fx_spec.h, FXM_FONLY: a workgroup carries two streams through the
        // stages, the last butterfly stores the spectra); compiled on first use
        if (const SpecKernel* k = spec_f_kernel(p)) {
            const int64_t pairs = (n_streams + k->shape.rows - 1) / k->shape.rows;      // (groups of streams: a workgroup's rows)
            const int64_t ws = spec_wg_splits(p, k, pairs, false);
            if (pairs * ws > (1ll << 30)) return fail(p, FXC_ERR_ARG, "too many streams for one launch");
            SpecArgs a = {x, p->d_win, spec, p->d_tw, nullptr, (long long)p->num_samp, (long long)p->n_pts, (long long)n_streams, (int)ws, ant,
                          reinterpret_cast<const float*>(p->d_win4), k->d_tw1, 0, nullptr};
            void* params[] = {&a};
            FXC_HIP(p, hipModuleLaunchKernel(k->fn, (unsigned)(pairs * ws), 1, 1, (unsigned)k->shape.threads(), 1, 1, 0, p->stream, params, nullptr));
            return FXC_OK;
        }
    }
    if (p->mixed && p->nchan > kMixedMaxN) {
        // one LDS row, the other in the output (pfb_fft_mixed_kernel, BIG): one frame per workgroup pass, 1024 threads
        const int64_t n_groups = n_streams * p->n_pts;
        const int64_t run = std::max<int64_t>(1, std::min<int64_t>(16, n_groups / ((int64_t)p->cu_count * 8)));
        const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((n_groups + run - 1) / run, 1 << 30));
        hipLaunchKernelGGL((pfb_fft_mixed_kernel<false, 1, false, false, true>), dim3(grid), dim3(1024), (size_t)p->nchan * sizeof(cf),
                           p->stream, x, p->d_win, spec, p->d_tw, p->mixed_plan, p->num_samp, p->nchan, p->ntaps, p->n_pts, n_streams,
                           1024, 1, MixedExtras{0, ant, p->nchan, nullptr, nullptr, nullptr});
        FXC_HIP(p, hipGetLastError());
        return FXC_OK;
    }
    if (p->mixed) {
        const int threads = std::max(256, p->mixed_tpr);
        const int rpw = threads / p->mixed_tpr;
        const MixedExtras no_blu = {mixed_wave_local(p, false), ant, p->nchan, nullptr, nullptr, nullptr};
        static const int tw_knob = FXC_DEV_ENV_INT("FXC_MIXED_TWLDS", 1), u_knob = FXC_DEV_ENV_INT("FXC_MIXED_U", 0);
        // U = 2 frames per slot where the measurements favour it (tools/bench_channelize.py, r04 experiments.md §7): up to 1280
        // channels (three or more 256-thread workgroups still fit a CU's LDS) and, with 512 or 1024 threads per row, from 1321 to
        // 4096 (1440 ... 2000 channels: 10 - 25 % faster than one frame per slot on 256 threads)
        const size_t row_bytes = (size_t)p->nchan * sizeof(cf);
        int u = (p->nchan <= 1280 || (p->mixed_tpr >= 512 && p->nchan <= 4096)) ? 2 : 1;
        u = std::min(u, fxc::mixed_rows_per_slot_cap(p->mixed_plan));     // 11 / 13: register butterflies one frame at a time here
        if (u_knob == 1 || u_knob == 2) u = u_knob;
        const bool twl = tw_knob && (1 + (size_t)rpw * 2 * u) * row_bytes <= (size_t)(160 * 1024);
        if (!twl) u = 1;
        const size_t lds = ((size_t)rpw * 2 * u + (twl ? 1 : 0)) * row_bytes;
        const int fpg = rpw * u;
        const int64_t n_groups = n_streams * ((p->n_pts + fpg - 1) / fpg);
        // runs of up to 16 frame groups per workgroup (the FIR's re-reads of a frame stay in one L2), many more workgroups than
        // fit at once when there are rows for it (no second, part-filled round of resident workgroups)
        static const int run_knob = FXC_DEV_ENV_INT("FXC_MIXED_RUN", 16);
        const int64_t run = std::max<int64_t>(1, std::min<int64_t>(run_knob, n_groups / ((int64_t)p->cu_count * 8)));
        const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((n_groups + run - 1) / run, 1 << 30));
#define FXC_MIXED_LAUNCH(TWL, UU)                                                                                                     \
    hipLaunchKernelGGL((pfb_fft_mixed_kernel<TWL, UU, false>), dim3(grid), dim3(threads), lds, p->stream, x, p->d_win, spec, p->d_tw, \
                       p->mixed_plan, p->num_samp, p->nchan, p->ntaps, p->n_pts, n_streams, p->mixed_tpr, 1, no_blu)
        if (!twl)
            FXC_MIXED_LAUNCH(false, 1);
        else if (u == 2)
            FXC_MIXED_LAUNCH(true, 2);
        else
            FXC_MIXED_LAUNCH(true, 1);
#undef FXC_MIXED_LAUNCH
        FXC_HIP(p, hipGetLastError());
        return FXC_OK;
    }
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
#if FXC_DEV_KERNELS
            hipLaunchKernelGGL(dft_any_kernel, dim3(grid), dim3(256), lds, p->stream, spec, p->d_tw, p->nchan, rows);
#else
            return fail(p, FXC_ERR_UNSUPPORTED, "no FFT kernel for %d channels in this build", p->nchan);      // (every such count takes the mixed-radix kernel)
#endif
    }
    FXC_HIP(p, hipGetLastError());
    return FXC_OK;
}

// Two antennas above 4096 channels off the powers of two (complex64 samples): two passes of the kernels built for the channel count --
// antenna 0 of every chunk pair through the F stage into `spec` [chunk][frame][nchan], then antenna 1 through the same stages with the
// last butterfly multiplying by antenna 0's values and adding to the thread's sums (fx_spec.h, FXM_XM): 2 x the algorithmic bytes, where
// the spectra of both antennas + xmul_kernel move 3 x.  raw[split][chunk][nchan] out, as mixed_fx_raw_sums.
bool two_pass_xm(fxc_plan* p, bool bytes_in) {
    if (bytes_in || p->n_ant != 2 || !p->mixed || p->mixed_blu || p->nchan <= 4096 || p->mixed_xeng) return false;
    if (!spec_f_kernel(p) || p->spec_f->shape.rows != 1) return false;
    if (!FXC_DEV_ENV_INT("FXC_XM", 1)) return false;
    if (!p->spec_xm_tried) {
        p->spec_xm_tried = true;
        const SpecKernel* k = spec_kernel(p->device, p->nchan, p->ntaps, kSpecXM);
        p->spec_xm = k->fn ? k : nullptr;
        if (!k->fn && env_int("FXC_RTC_VERBOSE", 0))
            std::fprintf(stderr, "libfxcorr: %d channels: no second-pass kernel (%s): both antennas' spectra + xmul_kernel\n", p->nchan, k->error.c_str());
    }
    return p->spec_xm != nullptr;
}
int two_pass_raw_sums(fxc_plan* p, const cf* x, int64_t n_chunks, int n_splits, cf* spec, cf* raw) {
    const SpecKernel* kf = p->spec_f;
    const SpecKernel* kx = p->spec_xm;
    if (n_splits % kx->shape.slots) return fail(p, FXC_ERR_STATE, "row splits and the second-pass kernel's slots disagree");
    const int64_t wsf = spec_wg_splits(p, kf, n_chunks, false);
    const int wsx = n_splits / kx->shape.slots;
    if (n_chunks * std::max<int64_t>(wsf, wsx) > (1ll << 30)) return fail(p, FXC_ERR_ARG, "too many chunks for one launch");
    SpecArgs a0 = {x, p->d_win, spec, p->d_tw, nullptr, (long long)p->num_samp, (long long)p->n_pts, (long long)n_chunks, (int)wsf, 1,
                   reinterpret_cast<const float*>(p->d_win4), kf->d_tw1, 2 * (long long)p->num_samp, nullptr};
    void* params0[] = {&a0};
    FXC_HIP(p, hipModuleLaunchKernel(kf->fn, (unsigned)(n_chunks * wsf), 1, 1, (unsigned)kf->shape.threads(), 1, 1, 0, p->stream, params0, nullptr));
    SpecArgs a1 = {x + p->num_samp, p->d_win, raw, p->d_tw, nullptr, (long long)p->num_samp, (long long)p->n_pts, (long long)n_chunks, wsx, 1,
                   reinterpret_cast<const float*>(p->d_win4), kx->d_tw1, 2 * (long long)p->num_samp, spec};
    void* params1[] = {&a1};
    FXC_HIP(p, hipModuleLaunchKernel(kx->fn, (unsigned)(n_chunks * wsx), 1, 1, (unsigned)kx->shape.threads(), 1, 1, 0, p->stream, params1, nullptr));
    return FXC_OK;
}

// two antennas on the mixed-radix kernel: F and X in one pass, raw[split][chunk][nchan] out (xmul_kernel's layout)
// (dc_u8 != nullptr: x is the uint8 I,Q stream of these chunks and dc_u8 its conversion offsets per stream)
int mixed_fx_raw_sums(fxc_plan* p, const cf* x, int64_t n_chunks, int n_splits, cf* raw, const cf* dc_u8 = nullptr) {
    if (p->spec) {
        // the build of fx_spec.h made for this channel count (h_rtc.h); its byte-ingest twin is compiled on first use
        const SpecKernel* k = p->spec;
        if (dc_u8) {
            if (!p->spec_u8_tried) {
                p->spec_u8_tried = true;
                const SpecKernel* k8 = spec_kernel(p->device, p->nchan, p->ntaps, kSpecU8);
                p->spec_u8 = k8->fn ? k8 : nullptr;
            }
            k = p->spec_u8;
        }
        if (k && n_splits % k->shape.slots == 0 && n_splits / k->shape.slots >= 1) {
            const int wg_splits = n_splits / k->shape.slots;
            const int64_t grid = n_chunks * wg_splits;
            if (grid > (1ll << 30)) return fail(p, FXC_ERR_ARG, "too many chunks for one launch");
            SpecArgs a = {x, p->d_win, raw, p->d_tw, dc_u8, (long long)p->num_samp, (long long)p->n_pts, (long long)n_chunks, wg_splits, 1,
                          reinterpret_cast<const float*>(p->d_win4), k->d_tw1, 0, nullptr};
            void* params[] = {&a};
            FXC_HIP(p, hipModuleLaunchKernel(k->fn, (unsigned)grid, 1, 1, (unsigned)k->shape.threads(), 1, 1, 0, p->stream, params, nullptr));
            return FXC_OK;
        }
        if (env_int("FXC_RTC_VERBOSE", 0))      // (results stay correct: the any-shape kernel below takes the call)
            std::fprintf(stderr, "libfxcorr: %d channels: the %s build of the kernel per channel count does not take this call (%s), the any-shape kernel does\n",
                         p->nchan, dc_u8 ? "byte-ingest" : "complex64", !k ? "no such build" : "its slots per workgroup do not divide the call's row splits");
    }
    const int threads = std::max(256, p->mixed_tpr);
    const int rpw = threads / p->mixed_tpr;
    const size_t lds = ((size_t)rpw * 4 + (p->mixed_xf_twl ? 1 : 0)) * p->nchan * sizeof(cf);
    const int64_t grid = n_chunks * n_splits;
    if (grid > (1ll << 30)) return fail(p, FXC_ERR_ARG, "too many chunks for one launch");
    const MixedExtras ex = {mixed_wave_local(p, true), 1, p->nchan, nullptr, nullptr, dc_u8};
#define FXC_MIXED_XF_LAUNCH(TWL, BYTES)                                                                                            \
    hipLaunchKernelGGL((pfb_fft_mixed_kernel<TWL, 2, true, false, false, BYTES>), dim3((unsigned)grid), dim3(threads), lds, p->stream, \
                       x, p->d_win, raw, p->d_tw, p->mixed_plan, p->num_samp, p->nchan, p->ntaps, p->n_pts, n_chunks, p->mixed_tpr,    \
                       n_splits, ex)
    if (p->mixed_xf_twl) {
        if (dc_u8)
            FXC_MIXED_XF_LAUNCH(true, true);
        else
            FXC_MIXED_XF_LAUNCH(true, false);
    } else {
        if (dc_u8)
            FXC_MIXED_XF_LAUNCH(false, true);
        else
            FXC_MIXED_XF_LAUNCH(false, false);
    }
#undef FXC_MIXED_XF_LAUNCH
    FXC_HIP(p, hipGetLastError());
    return FXC_OK;
}

// float32 sums of up to kRowSpectra spectra per raw row (the rows themselves are summed in float64).  Measured on 10 000
// frames against the float64 mean of the per-frame rows (profiles/r02/round2_experiments.md): 256 spectra per row 5.6e-9 of
// max|vis|, 1 024: 2.5e-8, 4 096: 6.3e-8 -- against a tolerance of 1e-5; 1 024 leaves a quarter of the rows to fold (34 MB
// instead of 92 MB per 10 000 frames)
constexpr int64_t kRowSpectra = 1024;

struct XGeom {
    int kx, n_splits;
};

// Workgroup splits of a chunk's (or stream pair's) frames for a specialised kernel: workgroups = groups x splits, every slot of a
// workgroup with a run of its own.  A run re-reads ntaps - 1 frames of history and should be 16 frames at least; with the X stage
// in the kernel a run is a float32 sum of at most kRowSpectra spectra; among the splits that allow, the one that fills the device's
// resident workgroups in whole rounds best
int64_t spec_wg_splits(const fxc_plan* p, const SpecKernel* k, int64_t n_groups, bool sums) {
    const SpecShape& sh = k->shape;
    const int64_t cap = (int64_t)p->cu_count * k->wgs_per_cu;
    const int64_t ws_lo = sums ? std::max<int64_t>(1, (p->n_pts + sh.slots * kRowSpectra - 1) / (sh.slots * kRowSpectra)) : 1;
    // runs of sixteen frames at least (a run re-reads ntaps - 1 frames of history and loads its taps and twiddles) -- unless that
    // leaves most of the device idle: a call over one chunk pair (the drop-in's _run_task, effex.py:490-494) is about latency, and
    // takes runs of two
    const int64_t few = std::max<int64_t>(n_groups, 1) * std::max<int64_t>(1, p->n_pts / (16 * (int64_t)sh.slots)) < cap;
    const int64_t min_run = few ? 2 : 16;
    // (... and of at most 256 rows per chunk: what reads the rows of a single chunk -- rows_spectrum_kernel -- adds them up one thread
    // per bin; 12 channels, 64 slots a workgroup: 4 096 rows and 1.1 ms per _run_task with 64 splits)
    const int64_t row_cap = few ? std::max<int64_t>(1, 256 / sh.slots) : 64;
    const int64_t ws_hi = std::max<int64_t>(ws_lo, std::min<int64_t>(std::min<int64_t>(64, row_cap), p->n_pts / (min_run * (int64_t)sh.slots)));
    int64_t best = ws_lo;
    double best_cost = 1e300;
    for (int64_t ws = ws_lo; ws <= ws_hi; ++ws) {
        const int64_t wgs = std::max<int64_t>(n_groups, 1) * ws;
        const double rounds = (double)((wgs + cap - 1) / cap) * (double)cap / (double)wgs;      // >= 1: idle share of the last round
        const double run = std::max(1.0, (double)p->n_pts / (double)(ws * sh.slots));
        const double cost = rounds * (1.0 + (double)(p->ntaps - 1) / run);
        if (cost < best_cost - 1e-9) {
            best_cost = cost;
            best = ws;
        }
    }
    return best;
}

// xf: this call takes the mixed-radix kernel that does F and X in one pass (mixed_xf plans; h_run.h::mixed_one_pass)
XGeom x_geometry(const fxc_plan* p, int64_t n_chunks, bool xf, bool xm = false) {
    XGeom g;
    if (xm) {      // two passes of the kernels built for the channel count (two_pass_raw_sums): the second one's rows
        g.kx = 1;
        g.n_splits = (int)(spec_wg_splits(p, p->spec_xm, n_chunks, true) * p->spec_xm->shape.slots);
        return g;
    }
    if (xf && p->spec) {
        g.kx = 1;
        g.n_splits = (int)(spec_wg_splits(p, p->spec, n_chunks, true) * p->spec->shape.slots);
        return g;
    }
    if (xf) {
        // workgroups = chunks x splits: eight per CU when the frames allow it, runs of four frame groups at least
        const int rpw = std::max(256, p->mixed_tpr) / p->mixed_tpr;
        const int64_t gps = (p->n_pts + rpw - 1) / rpw;
        const int64_t want = ((int64_t)p->cu_count * 8 + n_chunks - 1) / std::max<int64_t>(n_chunks, 1);
        g.kx = 1;
        g.n_splits = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(want, 256), gps / 4));
        // a slot sums its run of frame groups -- one spectrum per group -- in float32 registers: no run longer than the
        // kRowSpectra spectra every other path limits a float32 row to (few channels with long chunks: nchan 12,
        // num_samp 2^22 would otherwise put 5 400 terms into one float32 sum)
        g.n_splits = (int)std::min<int64_t>(std::max<int64_t>(g.n_splits, (gps + kRowSpectra - 1) / kRowSpectra), 1 << 16);
        return g;
    }
    g.kx = 1;
    while (g.kx < 256 && g.kx < p->nchan) g.kx <<= 1;
    const int iy = 256 / g.kx;
    int64_t splits = (p->n_pts + (int64_t)iy * 64 - 1) / ((int64_t)iy * 64);
    if (splits < 1) splits = 1;
    if (splits > 256) splits = 256;
    g.n_splits = (int)splits;
    return g;
}

// F and X in one pass for this call?  mixed_xf plans above 4096 channels that have the F stage built for their channel count
// (fx_spec.h, one stream per workgroup) take it for complex64 input -- F pass + xmul_kernel: 4500 channels 4.39 -> 3.43 ms -- and keep
// the one-pass kernel for the receivers' bytes, which it converts itself (3.5 ms against 5.0 through the conversion pass)
bool mixed_one_pass(const fxc_plan* p, bool bytes_in) { return p->mixed_xf && (bytes_in || !p->xf_bytes_only); }

// chunks per pass on the generic path so that spectra + raw sums fit the workspace target
int64_t generic_chunks_per_pass(const fxc_plan* p, int64_t n_chunks, const XGeom& g, int64_t* spec_bytes,
                                int64_t* raw_bytes, bool xf, bool xm = false) {
    const int64_t spec_per_chunk = xf ? 0 : (int64_t)(xm ? 1 : p->n_ant) * p->n_pts * p->nchan * (int64_t)sizeof(cf);      // (xm: antenna 0's spectra only)
    const int64_t raw_per_chunk = (int64_t)g.n_splits * p->n_base * p->nchan * (int64_t)sizeof(cf);
    int64_t cb = ws_target() / std::max<int64_t>(1, spec_per_chunk + raw_per_chunk);
    if (cb < 1) cb = 1;
    if (cb > n_chunks) cb = n_chunks;
    if (p->mixed_xeng && cb > 65535 / g.n_splits) cb = std::max<int64_t>(1, 65535 / g.n_splits);   // the X-engines carry chunk and range in grid.y
    *spec_bytes = (cb * spec_per_chunk + 255) / 256 * 256;
    *raw_bytes = cb * raw_per_chunk;
    return cb;
}

// fold_partial_kernel: splits that leave each of its threads about eight rows to walk, and the partials within 64 MiB
// (long rows -- many baselines -- have the parallelism in the row itself)
int fold_max_splits(int64_t row_len) { return (int)std::max<int64_t>(1, std::min<int64_t>(kFoldMaxSplits, (4ll << 20) / row_len)); }
int fold_splits(int64_t n_rows, int64_t row_len) {
    return (int)std::max<int64_t>(1, std::min<int64_t>(fold_max_splits(row_len), n_rows / (8 * kFoldPhases)));
}
int64_t fold_part_bytes(const fxc_plan* p) { return (int64_t)fold_max_splits((int64_t)p->n_base * p->nchan) * p->n_base * p->nchan * (int64_t)sizeof(cd); }

const FoldFinish kNoFinish = {nullptr, nullptr, nullptr, 0.0, 0};

// acc[p][bin] += sum of the raw rows [n_base][nchan], and `fin` for every element: two launches, one when the rows are few
// `done`: an event to complete with the last kernel (it rides on that dispatch: no packet of its own in the stream)
int fold_rows(fxc_plan* p, const cf* raw, cd* part, int64_t n_rows, int layout, const FoldFinish& fin, hipEvent_t done = nullptr) {
    const int64_t row_len = (int64_t)p->n_base * p->nchan;
    const unsigned cols = (unsigned)((row_len + 255) / 256);
    const int splits = fold_splits(n_rows, row_len);
    if (splits == 1) {
        hipExtLaunchKernelGGL(fold_finish_kernel<cf>, dim3(cols), dim3(256 * kFoldPhases), 0, p->stream, nullptr, done, 0, raw,
                              n_rows, p->d_acc, p->nchan, p->n_base, layout, fin);
    } else {
        hipLaunchKernelGGL(fold_partial_kernel, dim3(cols, splits), dim3(256 * kFoldPhases), 0, p->stream, raw, part, row_len,
                           n_rows, splits);
        // the partials are in the rows' own layout
        hipExtLaunchKernelGGL(fold_finish_kernel<cd>, dim3(cols), dim3(256 * kFoldPhases), 0, p->stream, nullptr, done, 0,
                              (const cd*)part, (int64_t)splits, p->d_acc, p->nchan, p->n_base, layout, fin);
    }
    FXC_HIP(p, hipGetLastError());
    return FXC_OK;
}

// the fold of the last fx_accumulate pass, if it is still pending, with `fin` riding along; nothing pending: `fin`
// alone on the accumulator as it stands
int flush_pending(fxc_plan* p, const FoldFinish* fin, hipEvent_t done) {
    if (p->pend.valid) {
        p->pend.valid = false;
        return fold_rows(p, p->pend.raw, p->pend.part, p->pend.n_rows, p->pend.layout, fin ? *fin : kNoFinish, done);
    }
    if (fin) {
        const int64_t n = (int64_t)p->n_base * p->nchan;
        hipExtLaunchKernelGGL(acc_finish_kernel, dim3(grid_for(n, 256, p->cu_count)), dim3(256), 0, p->stream, nullptr, done, 0,
                              p->d_acc, p->nchan, p->n_base, *fin);
        FXC_HIP(p, hipGetLastError());
    } else if (done) {
        FXC_HIP(p, hipEventRecord(done, p->stream));
    }
    return FXC_OK;
}

// the rows of an fx_accumulate pass: folded right away, or left pending when the pass is the call's last one
int fold_or_defer(fxc_plan* p, const cf* raw, cd* part, int64_t n_rows, int layout, bool last_pass) {
    if (!last_pass) return fold_rows(p, raw, part, n_rows, layout, kNoFinish);
    p->pend.valid = true;
    p->pend.raw = raw;
    p->pend.part = part;
    p->pend.n_rows = n_rows;
    p->pend.layout = layout;
    return FXC_OK;
}

// workgroups of a fused launch over n_pairs chunk pairs: one per CU; a launch with fewer chunks than that is all
// tail (frame ranges), on fewer workgroups when a range would be under four frames (each reloads up to three
// frames of history)
int fused_grid(const fxc_plan* p, int64_t n_pairs) {
    if (n_pairs >= p->fused_grid_max) return p->fused_grid_max;
    const int64_t frames = n_pairs * p->n_pts;
    return (int)std::max<int64_t>(1, std::min<int64_t>(frames / 4, p->fused_grid_max));
}

// chunks per raw row when only the integration is wanted: float32 sums of up to kRowSpectra spectra
int64_t fused_unit(const fxc_plan* p) { return std::max<int64_t>(1, std::min<int64_t>(kRowSpectra / std::max<int64_t>(1, p->n_pts), 64)); }

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
                 bool rows_are_chunks, int64_t num_samp, bool dck) {
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
    // kernel profiling (bench.py): the two events ride on the dispatch itself (hipExtLaunchKernelGGL: start and stop
    // time of this kernel, no barrier packets of their own in the stream -- events recorded around the launch cost 6 + 11 us
    // of stream time per launch, profiles/r03/experiments.md)
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    if (p->profiling && (hipEventCreate(&ev_a) != hipSuccess || hipEventCreate(&ev_b) != hipSuccess)) ev_a = ev_b = nullptr;
#define FXC_FUSED_LAUNCH(KERNEL, LDS, ...) \
    hipExtLaunchKernelGGL(KERNEL, dim3(grid), dim3(kThreads), LDS, p->stream, ev_a, ev_b, 0, __VA_ARGS__)
    if (dc_u8 && dck)
        FXC_FUSED_LAUNCH((fx_fused4096_kernel<false, true, true>), kLdsBytes + kDckLdsBytes, x, num_samp, p->n_pts, n_pairs,
                         p->d_win4, p->d_tw1, p->d_tw2, out, stamps, dc_u8, seg, (int)unit, rows_are_chunks ? 1 : 0);
    else if (dc_u8)
        FXC_FUSED_LAUNCH((fx_fused4096_kernel<false, true>), kLdsBytes, x, num_samp, p->n_pts, n_pairs, p->d_win4, p->d_tw1,
                         p->d_tw2, out, stamps, dc_u8, seg, (int)unit, rows_are_chunks ? 1 : 0);
    else if (spec_out)      // (`unit` carries the stream pairs per chunk here)
        FXC_FUSED_LAUNCH((fx_fused4096_kernel<true, false>), kLdsBytes, x, num_samp, p->n_pts, n_pairs, p->d_win4, p->d_tw1,
                         p->d_tw2, out, stamps, (const cf*)nullptr, seg, p->n_ant / 2, 1);
    else
        FXC_FUSED_LAUNCH((fx_fused4096_kernel<false, false>), kLdsBytes, x, num_samp, p->n_pts, n_pairs, p->d_win4, p->d_tw1,
                         p->d_tw2, out, stamps, (const cf*)nullptr, seg, (int)unit, rows_are_chunks ? 1 : 0);
#undef FXC_FUSED_LAUNCH
    if (ev_a) p->kev.emplace_back(ev_a, ev_b);
    FXC_HIP(p, hipGetLastError());
    return FXC_OK;
}

// chunks per X-engine workgroup (= per raw row) for an integration over nc chunks.  The kernel's one-wave workgroups
// all do the same work, so the launch is cut into exactly as many as are resident at once (p->x_resident, from the
// occupancy API: one round, no tail) unless a float32 row (fused_unit) allows fewer chunks than that: then many rounds
// antenna tiles of 16 of the matrix-core X-engine: XT = 1 .. 4
#define FXC_XMFMA_DISPATCH(p, ...)                       \
    do {                                                 \
        switch (((p)->n_ant + 15) / 16) {                \
            case 1: { constexpr int XT = 1; __VA_ARGS__; } break; \
            case 2: { constexpr int XT = 2; __VA_ARGS__; } break; \
            case 3: { constexpr int XT = 3; __VA_ARGS__; } break; \
            default: { constexpr int XT = 4; __VA_ARGS__; } break; \
        }                                                \
    } while (0)

int64_t xengine_group(const fxc_plan* p, int64_t nc, int64_t unit) {
    // more than 8 antennas: a column of 16 bins with all antennas per workgroup (xengine_mfma_kernel), or -- the vector
    // kernel it replaced -- a column of 64 shared by the G (G + 1) / 2 pairs of antenna blocks (xengine_block_kernel)
    const int64_t gb = (p->n_ant + kXB - 1) / kXB;
    int x_ch = 16;
    if (p->x_mfma) FXC_XMFMA_DISPATCH(p, x_ch = XMfmaGeo<XT>::kCH);
    const int64_t cols = p->x_mfma ? std::max<int64_t>(1, p->nchan / x_ch)
                                   : std::max<int64_t>(1, p->nchan / kXThreads) * (p->n_ant > kXB ? gb * (gb + 1) / 2 : 1);
    const int64_t groups = std::max<int64_t>(1, p->x_resident / cols);
    return std::max<int64_t>(1, std::min<int64_t>(unit, (nc + groups - 1) / groups));
}

// frame ranges per chunk group of the X-engines (k_finish.h::x_range): groups of one chunk with more than kRowSpectra frames
int x_ranges(const fxc_plan* p, int64_t unit) {
    if (p->n_ant <= 2 || unit > 1) return 1;
    return (int)std::min<int64_t>((p->n_pts + kRowSpectra - 1) / kRowSpectra, 4096);
}

// layout of the raw per-chunk sums the fused paths produce (see raw_index)
int fused_layout(const fxc_plan* p) { return p->n_ant == 2 ? 1 : (p->path == FXC_PATH_FUSED ? 2 : 0); }

// chunks per pass on the fused paths: 2 antennas only need the raw rows; more antennas also the spectra
int64_t fused_chunks_per_pass(const fxc_plan* p, int64_t n_chunks, int64_t* spec_bytes, int64_t* raw_bytes) {
    // (3 and more antennas: one raw row per chunk and frame range at most, x_ranges)
    const int64_t xr = x_ranges(p, 1);
    const int64_t raw_per_chunk = (int64_t)p->n_base * p->nchan * (int64_t)sizeof(cf) * xr;
    const int64_t spec_per_chunk = p->n_ant == 2 ? 0 : (int64_t)p->n_ant * p->n_pts * p->nchan * (int64_t)sizeof(cf);
    int64_t cb = ws_target() / (raw_per_chunk + spec_per_chunk);
    if (cb < 1) cb = 1;
    if (cb > n_chunks) cb = n_chunks;
    if (p->n_ant > 2 && cb > 65535 / xr) cb = std::max<int64_t>(1, 65535 / xr);   // the X-engines carry group and range in grid.y
    *spec_bytes = (cb * spec_per_chunk + 255) / 256 * 256;
    // 2 antennas: one leading-part row per workgroup after the chunk rows (fx_fused4096_kernel)
    *raw_bytes = ((cb + (p->n_ant == 2 ? p->fused_grid_max : 0)) * raw_per_chunk + 255) / 256 * 256;
    return cb;
}

int launch_xengine(fxc_plan* p, const cf* spec, cf* raw, int64_t nc, int cg, int xr);

// raw[c][p][layout] for nc chunks starting at x; spec = scratch for the multi-antenna path.  2 antennas: rows of
// `unit` chunks + leading-part rows (fused_rows() of them in all)
int fused_raw_sums(fxc_plan* p, const cf* x, int64_t nc, cf* spec, cf* raw, const cf* dc_u8 = nullptr, int64_t unit = 1,
                   bool rows_are_chunks = true, bool dck = false) {
    using namespace fxc::fused;
    if (p->n_ant == 2) return launch_fused(p, x, nc, raw, false, dc_u8, unit, rows_are_chunks, 0, dck);
    // 3 .. 64 antennas: spectra to HBM as [chunk][frame][antenna] rows (the F-only fused kernel in its own position order
    // at nchan 4096 / ntaps 4, the F-only tiled kernel in natural order otherwise), then the register-resident X-engine
    // (over blocks of 8 antennas beyond 8).
    // unit = chunks per raw row here too: ceil(nc / unit) rows come out
    int rc = p->path == FXC_PATH_FUSED ? launch_fused(p, x, nc * (p->n_ant / 2), spec, true)
                                       : tiled_channelize(p, x, spec, nc * p->n_ant, p->n_ant);
    if (rc) return rc;
    return launch_xengine(p, spec, raw, nc, (int)unit, x_ranges(p, unit));
}

// spec[chunk][frame][antenna][nchan] -> raw[range][group][baseline][nchan] (natural bin order): 3 .. 8 antennas in registers,
// more on the matrix cores (plans whose nchan the tiles divide) or over blocks of 8
int launch_xengine(fxc_plan* p, const cf* spec, cf* raw, int64_t nc, int cg, int xr) {
    int x_ch = 16;
    if (p->x_mfma) FXC_XMFMA_DISPATCH(p, x_ch = XMfmaGeo<XT>::kCH);
    const dim3 grid((p->nchan + kXThreads - 1) / kXThreads, (unsigned)(((nc + cg - 1) / cg) * xr));
#define FXC_X_LAUNCH(A) \
    hipLaunchKernelGGL(xengine_kernel<A>, grid, dim3(kXThreads), 0, p->stream, spec, raw, p->n_pts, p->nchan, nc, cg, xr)
    switch (p->n_ant) {
        case 3: FXC_X_LAUNCH(3); break;
        case 4: FXC_X_LAUNCH(4); break;
        case 5: FXC_X_LAUNCH(5); break;
        case 6: FXC_X_LAUNCH(6); break;
        case 7: FXC_X_LAUNCH(7); break;
        case 8: FXC_X_LAUNCH(8); break;
        default: if (p->x_mfma && p->nchan % x_ch == 0) {       // (the matrix-core tiles take whole columns of x_ch bins)
            FXC_XMFMA_DISPATCH(p, hipLaunchKernelGGL(xengine_mfma_kernel<XT>, dim3((unsigned)(p->nchan / XMfmaGeo<XT>::kCH), grid.y),
                                                     dim3(XMfmaGeo<XT>::kThreads), XMfmaGeo<XT>::kLdsBytes, p->stream, spec, raw,
                                                     p->n_pts, p->nchan, nc, cg, p->n_ant, xr, FXC_DEV_ENV_INT("FXC_XMFMA_ABL", 0)));
        } else {
            const unsigned gb = (unsigned)((p->n_ant + kXB - 1) / kXB);
            hipLaunchKernelGGL(xengine_block_kernel, dim3(grid.x, grid.y, gb * (gb + 1) / 2), dim3(kXThreads), 0, p->stream, spec,
                               raw, p->n_pts, p->nchan, nc, cg, p->n_ant, xr);
        } break;
    }
#undef FXC_X_LAUNCH
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
void tiled_launch(fxc_plan* p, const cf* x, int64_t nc, int n_splits, cf* raw, int64_t n_streams, const cf* dc_u8 = nullptr,
                  int spec_a = 0, int64_t s_base = 0) {
    const int grid = (int)std::min<int64_t>(nc * n_splits, SPEC ? p->tiled_grid_max_f : p->tiled_grid_max);
    if constexpr (G::N <= 4096) {
        if (p->tiled_ring) {
            if constexpr (!SPEC) {
                if (dc_u8) {   // uint8 ingest: x is the byte stream
                    hipLaunchKernelGGL((fx_tiled_ring_kernel<G, false, true>), dim3(grid), dim3(G::kThreads), G::kLdsBytesRing,
                                       p->stream, x, p->num_samp, p->n_pts, nc, n_splits, p->d_win4, p->d_tw0, p->d_tw1,
                                       p->d_tw2, raw, n_streams, dc_u8, 0, (int64_t)0);
                    return;
                }
            }
            hipLaunchKernelGGL((fx_tiled_ring_kernel<G, SPEC, false>), dim3(grid), dim3(G::kThreads), G::kLdsBytesRing,
                               p->stream, x, p->num_samp, p->n_pts, nc, n_splits, p->d_win4, p->d_tw0, p->d_tw1, p->d_tw2, raw,
                               n_streams, (const cf*)nullptr, spec_a, s_base);
            return;
        }
    }
    hipLaunchKernelGGL((fx_tiled_kernel<G, SPEC>), dim3(grid), dim3(G::kThreads), G::kLdsBytes, p->stream, x, p->num_samp,
                       p->n_pts, nc, n_splits, p->prefilter ? 1 : p->ntaps, p->prefilter ? p->d_ones : p->d_win, p->d_tw0,
                       p->d_tw1, p->d_tw2, raw, n_streams, spec_a, s_base);
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
bool small_nchan(int n) { return n == 16 || n == 32 || n == 64 || n == 128 || n == 256; }

// ---- tiled path, 16 .. 256 channels (k_small.h) -----------------------------------------------------
#define FXC_SMALL_DISPATCH(p, CALL)                                                   \
    switch ((p)->nchan) {                                                             \
        case 16: { constexpr int P = 1; CALL; } break;                                \
        case 32: { constexpr int P = 2; CALL; } break;                                \
        case 64: { constexpr int P = 4; CALL; } break;                                \
        case 128: { constexpr int P = 8; CALL; } break;                               \
        case 256: { constexpr int P = 16; CALL; } break;                              \
        default: return fail(p, FXC_ERR_UNSUPPORTED, "no small-transform kernel for nchan=%d", (p)->nchan); \
    }

int small_setup(fxc_plan* p) {
    int per_cu = 0;
    FXC_SMALL_DISPATCH(p, FXC_HIP(p, hipOccupancyMaxActiveBlocksPerMultiprocessor(
                              &per_cu, reinterpret_cast<const void*>(&fx_small_ring_kernel<P, false, false>), 256, 0)));
    if (per_cu < 1) return fail(p, FXC_ERR_HIP, "small-transform kernel for nchan=%d does not fit a CU", p->nchan);
    p->small_wgs = per_cu * p->cu_count;
    // work items resident at once (tiled_splits); the F-only variant has the same launch geometry
    p->tiled_grid_max = p->tiled_grid_max_f = p->small_wgs * 4 * (32 / (p->nchan / 16));
    return FXC_OK;
}

int small_grid(const fxc_plan* p, int64_t items) {
    const int64_t items_per_wg = 4 * (32 / (p->nchan / 16));
    return (int)std::min<int64_t>((items + items_per_wg - 1) / items_per_wg, p->small_wgs);
}

// dc_u8 != nullptr: x is the byte stream, dc_u8 the streams' conversion offsets
int small_launch(fxc_plan* p, const cf* x, int64_t nc, int n_splits, cf* raw, const cf* dc_u8) {
    const int grid = small_grid(p, nc * n_splits);
    if (dc_u8) {
        FXC_SMALL_DISPATCH(p, hipLaunchKernelGGL((fx_small_ring_kernel<P, true, false>), dim3(grid), dim3(256), 0, p->stream, x,
                                                 p->num_samp, p->n_pts, nc, n_splits, p->d_win4, p->d_tw_small, raw, dc_u8,
                                                 2 * nc, 0, (int64_t)0));
    } else {
        FXC_SMALL_DISPATCH(p, hipLaunchKernelGGL((fx_small_ring_kernel<P, false, false>), dim3(grid), dim3(256), 0, p->stream, x,
                                                 p->num_samp, p->n_pts, nc, n_splits, p->d_win4, p->d_tw_small, raw,
                                                 (const cf*)nullptr, 2 * nc, 0, (int64_t)0));
    }
    return FXC_OK;
}

int tiled_splits(const fxc_plan* p, int64_t n_chunks, bool f_only = false);

// F-stage only (see tiled_channelize): n_streams consecutive streams -> natural-order spectra, pairs of streams per item
int tiled_splits(const fxc_plan* p, int64_t n_chunks, bool f_only);
int64_t tiled_streams_per_pass(const fxc_plan* p);
int tiled_prefilter(fxc_plan* p, const cf* x, int64_t n_streams, const cf** y_out);

int small_channelize(fxc_plan* p, const cf* x, cf* spec, int64_t n_streams, int spec_a) {
    KernelTimer kt(p);
    const int64_t per_pass = tiled_streams_per_pass(p);      // everything at once unless the pre-filter bounds a pass
    for (int64_t s0 = 0; s0 < n_streams; s0 += per_pass) {
        const int64_t ns = std::min(per_pass, n_streams - s0);
        const cf* xs = x + s0 * p->num_samp;
        if (p->prefilter) {
            const int rc = tiled_prefilter(p, xs, ns, &xs);
            if (rc) return rc;
        }
        const int64_t pairs = (ns + 1) / 2;
        const int n_splits = tiled_splits(p, pairs, true);
        const int grid = small_grid(p, pairs * n_splits);
        // spec_a == 0: this pass's streams start at row s0 * n_pts; by frame: rows are placed from the global stream index
        FXC_SMALL_DISPATCH(p, hipLaunchKernelGGL((fx_small_ring_kernel<P, false, true>), dim3(grid), dim3(256), 0, p->stream, xs,
                                                 p->num_samp, p->n_pts, pairs, n_splits, p->d_win4, p->d_tw_small,
                                                 spec_a ? spec : spec + s0 * p->n_pts * p->nchan, (const cf*)nullptr, ns, spec_a,
                                                 spec_a ? s0 : (int64_t)0));
    }
    kt.stop();
    FXC_HIP(p, hipGetLastError());
    return FXC_OK;
}

// frame ranges per chunk so that a launch has at least ~2 work items per resident workgroup
int tiled_splits(const fxc_plan* p, int64_t n_chunks, bool f_only) {
    const int64_t cap = f_only ? p->tiled_grid_max_f : p->tiled_grid_max;
    const int64_t want = (2 * cap + n_chunks - 1) / n_chunks;
    // runs of eight frames at least -- the two-pass 8192-channel route (one workgroup per CU, 8 us a frame) with a handful of chunks:
    // of one (latency: see spec_wg_splits)
    const bool few = p->x8192 && n_chunks * std::max<int64_t>(1, p->n_pts / 8) < p->cu_count;
    const int64_t most = std::max<int64_t>(1, p->n_pts / (few ? 1 : 8));
    const int64_t fill = std::max<int64_t>(1, std::min<int64_t>(std::min(want, most), 256));
    if (f_only) return (int)fill;
    // F+X: a work item sums its frames in float32, like the headline kernel's rows at most kRowSpectra of them (small
    // channel counts and long chunks have thousands of frames per chunk)
    const int64_t exact = std::min<int64_t>((p->n_pts + kRowSpectra - 1) / kRowSpectra, 65536);
    return (int)std::max(fill, exact);
}

// streams of the tiled path per pass: what a pass keeps beside the raw rows -- the pre-filter's output, or antenna 0's spectra of the
// two-pass 8192-channel route -- stays within the workspace target
int64_t tiled_streams_per_pass(const fxc_plan* p) {
    if (p->x8192) {       // 8192 channels in two passes: antenna 0's spectra of a pass stay within the workspace target
        const int64_t chunks = std::max<int64_t>(1, ws_target() / (p->n_pts * (int64_t)p->nchan * (int64_t)sizeof(cf)));
        return 2 * std::min<int64_t>(chunks, 1 << 20);
    }
    if (!p->prefilter) return INT64_MAX;
    int64_t n = ws_target() / (p->num_samp * (int64_t)sizeof(cf));
    n = std::min<int64_t>(n, 65534) & ~(int64_t)1;      // grid.y carries the stream (or a row of them); whole pairs
    return std::max<int64_t>(2, n);
}

// y = pre-filtered copy of n_streams streams (plan buffer, grown on demand)
int tiled_prefilter(fxc_plan* p, const cf* x, int64_t n_streams, const cf** y_out) {
    const int rg = grow(p, &p->d_pre, &p->pre_bytes, (size_t)n_streams * p->num_samp * sizeof(cf));
    if (rg) return rg;
    cf* y = static_cast<cf*>(p->d_pre);
    const int tp = p->pre_tp;
    // two adjacent positions per thread (16-byte accesses) for the 8-frame block (1024 channels / 8 taps: -6 %; the 16-frame
    // block would need 218 VGPRs: -2 % at 512 channels, +3 % at 2048; the 32-frame block has no registers to spare);
    // streams of odd length are not 16-byte aligned one after the other.  FXC_PRE_W=1: developer knob, 8-byte accesses
    static const bool narrow = FXC_DEV_ENV_INT("FXC_PRE_W", 0) == 1;
    const int w = (!narrow && tp == 8 && (p->num_samp % 2) == 0 && (reinterpret_cast<uintptr_t>(x) % 16) == 0) ? 2 : 1;
    // channel counts below 256 w: several streams side by side in a workgroup, while their span fits a buffer descriptor
    const int64_t stream_bytes = p->num_samp * (int64_t)sizeof(cf);
    const int spb = (int)std::max<int64_t>(1, std::min<int64_t>(256 * w / p->nchan, (1ll << 31) / stream_bytes));
    const bool pack = p->nchan < 256 * w;
    const int64_t rows = (n_streams + spb - 1) / spb;
    // frame splits so that a few-stream call still fills the chip; each split reloads one block of history
    const int64_t blocks = std::max<int64_t>(1, p->nchan / (256 * w)) * rows;
    int64_t fs = std::max<int64_t>(1, (2 * (int64_t)p->cu_count + blocks - 1) / blocks);
    fs = std::min<int64_t>(fs, std::max<int64_t>(1, p->n_pts / (4 * tp)));
    const int64_t per = ((p->n_pts + fs - 1) / fs + 2 * tp - 1) / (2 * tp) * (2 * tp);
    const dim3 grid((unsigned)std::max(1, p->nchan / (256 * w)), (unsigned)rows, (unsigned)((p->n_pts + per - 1) / per));
#define FXC_PRE_LAUNCH(TP, W)                                                                                                    \
    do {                                                                                                                         \
        if (pack)                                                                                                                \
            hipLaunchKernelGGL((pfb_prefilter_kernel<TP, W, true>), grid, dim3(256), 0, p->stream, x, y, p->d_hpre, p->num_samp,    \
                               p->nchan, p->n_pts, per, spb, n_streams);                                                         \
        else                                                                                                                     \
            hipLaunchKernelGGL((pfb_prefilter_kernel<TP, W, false>), grid, dim3(256), 0, p->stream, x, y, p->d_hpre, p->num_samp,   \
                               p->nchan, p->n_pts, per, spb, n_streams);                                                         \
    } while (0)
    if (tp == 8) {
        if (w == 2) FXC_PRE_LAUNCH(8, 2);
        else FXC_PRE_LAUNCH(8, 1);
    } else if (tp == 16) {
        FXC_PRE_LAUNCH(16, 1);
    } else {
        FXC_PRE_LAUNCH(32, 1);
    }
#undef FXC_PRE_LAUNCH
    FXC_HIP(p, hipGetLastError());
    *y_out = y;
    return FXC_OK;
}

// raw[split][c][k] (natural bin order) for nc chunks starting at x (nc * 2 <= tiled_streams_per_pass())
int tiled_raw_sums(fxc_plan* p, const cf* x, int64_t nc, int n_splits, cf* raw, const cf* dc_u8 = nullptr) {
    KernelTimer kt(p);
    if (p->small) {
        if (p->prefilter && !dc_u8) {       // more than four taps: the FIR as its own pass, the wave-local kernel with one unit tap
            const int rp = tiled_prefilter(p, x, 2 * nc, &x);
            if (rp) return rp;
        }
        const int rc = small_launch(p, x, nc, n_splits, raw, dc_u8);
        if (rc) return rc;
        kt.stop();
        FXC_HIP(p, hipGetLastError());
        return FXC_OK;
    }
    if (p->x8192 && p->num_samp < (1ll << 28)) {
        // 8192 channels, two antennas, up to four taps, in TWO passes (2 x the algorithmic bytes; the split into two 4096-channel
        // problems and the pair kernel move 3 x): f8192_ring_kernel writes antenna 0's spectra (in register order: a layout private
        // to this route), its XM form runs antenna 1 through the same stages and multiplies by them as it goes --
        // raw[split][chunk][N] like the tiled kernels'.  Bytes in (dc_u8: the conversion offsets [chunk][2]): converted on their way
        // into the ring
        int rc = grow(p, &p->d_pre, &p->pre_bytes, (size_t)nc * p->n_pts * p->nchan * sizeof(cf));
        if (rc) return rc;
        cf* s0 = static_cast<cf*>(p->d_pre);
        const int64_t want = (2 * (int64_t)p->cu_count + nc - 1) / nc;
        const bool few = nc * std::max<int64_t>(1, p->n_pts / 8) < p->cu_count;
        const int f_splits = (int)std::max<int64_t>(1, std::min<int64_t>(want, p->n_pts / (few ? 1 : 8)));
        const dim3 grid_f((unsigned)std::min<int64_t>(nc, std::max<int64_t>(1, (int64_t)p->cu_count * 8 / f_splits)), (unsigned)f_splits);
        const dim3 grid_x((unsigned)std::min<int64_t>(nc, std::max<int64_t>(1, (int64_t)p->cu_count * 8 / n_splits)), (unsigned)n_splits);
        // antenna 1's stream of the first chunk: num_samp samples (of 8 bytes, or of 2) behind antenna 0's
        const cf* x1 = reinterpret_cast<const cf*>(reinterpret_cast<const char*>(x) + p->num_samp * (dc_u8 ? 2 : (int64_t)sizeof(cf)));
#define FXC_X8192_LAUNCH(U8)                                                                                                                  \
    do {                                                                                                                                      \
        hipLaunchKernelGGL((f8192_ring_kernel<false, true, U8>), grid_f, dim3(kF8192Threads), 0, p->stream, x, p->num_samp, p->n_pts, nc,        \
                           f_splits, p->d_win4, p->d_tw0, p->d_tw1, p->d_tw2, s0, 0, (int64_t)0, 2 * p->num_samp, (const cf*)nullptr, dc_u8, 0); \
        hipLaunchKernelGGL((f8192_ring_kernel<true, true, U8>), grid_x, dim3(kF8192Threads), 0, p->stream, x1, p->num_samp, p->n_pts, nc,        \
                           n_splits, p->d_win4, p->d_tw0, p->d_tw1, p->d_tw2, raw, 0, (int64_t)0, 2 * p->num_samp, (const cf*)s0, dc_u8, 1);     \
    } while (0)
        if (dc_u8) FXC_X8192_LAUNCH(true);
        else FXC_X8192_LAUNCH(false);
#undef FXC_X8192_LAUNCH
        kt.stop();
        FXC_HIP(p, hipGetLastError());
        return FXC_OK;
    }
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
int tiled_channelize(fxc_plan* p, const cf* x, cf* spec, int64_t n_streams, int spec_a) {
    if (p->small_f) return small_channelize(p, x, spec, n_streams, spec_a);
    if (p->f8192 && p->num_samp < (1ll << 28)) {
        // one stream per workgroup; runs of at least eight frames (a run re-reads three frames of history), enough workgroups
        // for two rounds of the CUs when the streams are few
        KernelTimer kt8(p);
        const int64_t want = (2 * (int64_t)p->cu_count + n_streams - 1) / n_streams;
        const bool few = n_streams * std::max<int64_t>(1, p->n_pts / 8) < p->cu_count;      // (one stream: _spectrometer_poly itself -- latency)
        const int n_splits = (int)std::max<int64_t>(1, std::min<int64_t>(want, p->n_pts / (few ? 1 : 8)));
        const int grid_x = (int)std::min<int64_t>(n_streams, std::max<int64_t>(1, (int64_t)p->cu_count * 8 / n_splits));
        const dim3 grid((unsigned)grid_x, (unsigned)n_splits);
        hipLaunchKernelGGL((f8192_ring_kernel<false, false>), grid, dim3(kF8192Threads), 0, p->stream, x, p->num_samp, p->n_pts, n_streams, n_splits,
                           p->d_win4, p->d_tw0, p->d_tw1, p->d_tw2, spec, spec_a, (int64_t)0, p->num_samp, (const cf*)nullptr, (const cf*)nullptr, 0);
        kt8.stop();
        FXC_HIP(p, hipGetLastError());
        return FXC_OK;
    }
    KernelTimer kt(p);
    const int64_t per_pass = tiled_streams_per_pass(p);
    for (int64_t s0 = 0; s0 < n_streams; s0 += per_pass) {
        const int64_t ns = std::min(per_pass, n_streams - s0);
        const cf* xs = x + s0 * p->num_samp;
        if (p->prefilter) {
            const int rc = tiled_prefilter(p, xs, ns, &xs);
            if (rc) return rc;
        }
        const int64_t pairs = (ns + 1) / 2;
        const int n_splits = tiled_splits(p, pairs, true);
        // spec_a == 0: this pass's streams start at row s0 * n_pts; by frame: the kernel places rows from the global stream index
        FXC_TILED_DISPATCH(p, (tiled_launch<G, true>(p, xs, pairs, n_splits, spec_a ? spec : spec + s0 * p->n_pts * p->nchan, ns,
                                                     nullptr, spec_a, spec_a ? s0 : 0)));
    }
    kt.stop();
    FXC_HIP(p, hipGetLastError());
    return FXC_OK;
}

// ---- nchan 8192 as two 4096-channel problems (pfb_split8192_kernel) ---------------------------------
int64_t split_chunks_per_pass(const fxc_plan* p, int64_t n_chunks) {
    const int64_t per_chunk = 4 * p->n_pts * 4096 * (int64_t)sizeof(cf) + 2 * 4096 * (int64_t)sizeof(cf);   // y + two raw rows
    return std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(n_chunks, 32767), ws_target() / per_chunk));
}

// raw = the fused kernel's rows over 2 nc chunk pairs (+ leading-part rows) for nc chunks of 8192-channel input
int split_raw_sums(fxc_plan* p, const cf* x, int64_t nc, cf* raw) {
    const int64_t half_samp = p->n_pts * 4096;
    int rc = grow(p, &p->d_pre, &p->pre_bytes, (size_t)nc * 4 * half_samp * sizeof(cf));
    if (rc) return rc;
    cf* y = static_cast<cf*>(p->d_pre);
    const int tp = p->pre_tp;
    // two adjacent positions per thread (16-byte accesses) for the 4- and 8-frame blocks (the 16-frame one has no registers left)
    static const bool narrow = FXC_DEV_ENV_INT("FXC_PRE_W", 0) == 1;
    const int w = (!narrow && tp <= 8 && (p->num_samp % 2) == 0 && (reinterpret_cast<uintptr_t>(x) % 16) == 0) ? 2 : 1;
    const int64_t blocks = (16 / w) * 2 * nc;
    int64_t fs = std::max<int64_t>(1, (2 * (int64_t)p->cu_count + blocks - 1) / blocks);
    fs = std::min<int64_t>(fs, std::max<int64_t>(1, p->n_pts / (4 * tp)));
    const int64_t per = ((p->n_pts + fs - 1) / fs + 2 * tp - 1) / (2 * tp) * (2 * tp);
    const dim3 grid((unsigned)(16 / w), (unsigned)(2 * nc), (unsigned)((p->n_pts + per - 1) / per));
    KernelTimer kt(p);
#define FXC_SPLIT_LAUNCH(TP, W)                                                                                             \
    hipLaunchKernelGGL((pfb_split8192_kernel<TP, W>), grid, dim3(256), 0, p->stream, x, y, p->d_hpre, p->d_tw8192, p->num_samp, \
                       p->n_pts, per)
    if (tp == 4) {
        if (w == 2) FXC_SPLIT_LAUNCH(4, 2);
        else FXC_SPLIT_LAUNCH(4, 1);
    } else if (tp == 8) {
        if (w == 2) FXC_SPLIT_LAUNCH(8, 2);
        else FXC_SPLIT_LAUNCH(8, 1);
    } else {
        FXC_SPLIT_LAUNCH(16, 1);
    }
#undef FXC_SPLIT_LAUNCH
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

}  // namespace
