/*
 * fxcorr.h — C ABI of the MI355X-native F/X hot path (libfxcorr.so).
 *
 * The reference (evanmayer/effex) is pure Python and has no FFI of its own; its seam is the
 * method surface of `Correlator` (effex/effex.py).  This ABI sits *underneath* that surface and
 * replaces the third-party GPU calls the reference makes on its hot path.  Each entry point
 * names the reference lines it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - every function returns FXC_OK (0) or a negative fxc_status; no exception crosses the ABI;
 *     fxc_last_error(plan) returns a NUL-terminated message owned by the plan (or by the library
 *     for plan == NULL), valid until the next failing call on that plan.
 *   - "complex64" = interleaved float re,im (8 B); "complex128" = interleaved double (16 B).
 *   - all device work is issued asynchronously on the plan's HIP stream; the caller owns every
 *     x/out buffer and keeps it alive until fxc_sync() / a finalize call returns.
 *   - a plan is thread-compatible, not thread-safe (one caller thread per plan — the reference
 *     drives the path from one thread, effex/effex.py:326-417).
 *   - there is NO CPU backend: without a HIP device fxc_plan_create fails with FXC_ERR_NODEVICE.
 *
 * Environment.  The library reads FOUR variables and no others (`strings libfxcorr.so | grep '^FXC_'` lists exactly these):
 *   FXC_RTC          0: plans keep the any-shape kernels for channel counts that are not a power of two instead of building the kernel
 *                    for the channel count (fxc_info.specialised); read when a plan is made.  Default 1.
 *   FXC_RTC_CACHE    directory of the code objects built at run time (default $XDG_CACHE_HOME/fxcorr, else ~/.cache/fxcorr; "0" or
 *                    empty: no files).  The pre-built code objects that ship beside the library are looked up first.
 *   FXC_RTC_VERBOSE  1: one line on stderr per kernel built or loaded for a channel count (stage list, registers, where it came from).
 *   FXC_WS_MB        upper bound of the lazily grown device workspace in MiB (default 12288); calls over more chunks run in passes.
 * Results do not depend on any of them beyond rounding (FXC_RTC chooses between two kernels of the same arithmetic family).  Route and
 * tuning knobs for A/B measurements exist only in the developer build (libfxcorr_dev.so, fxc_dev_kernels() == 1), which tests and
 * tools load explicitly.
 */
#ifndef FXCORR_H
#define FXCORR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FXC_VERSION 106 /* 0.1.1: fxc_info.spec_source, spec_seconds */

typedef struct fxc_plan fxc_plan; /* opaque; one per (device, configuration) */
typedef struct fxc_pipe fxc_pipe; /* opaque; host-fed double-buffered front end on a plan */

#define FXC_STREAM_OWNED ((void*)(intptr_t)-1)

enum fxc_status {
    FXC_OK = 0,
    FXC_ERR_ARG = -1,         /* bad argument (NULL, non-positive size, ...)                     */
    FXC_ERR_UNSUPPORTED = -2, /* ntaps > 32 (cusignal raises NotImplementedError), nchan too big */
    FXC_ERR_HIP = -3,         /* a HIP runtime call failed; message has hipGetErrorString        */
    FXC_ERR_NOMEM = -4,
    FXC_ERR_NODEVICE = -5,
    FXC_ERR_STATE = -6,       /* call sequence error (e.g. finalize with nothing accumulated)    */
    FXC_ERR_COMM = -7         /* librccl could not be bound, or an RCCL call failed              */
};

/* FXC_MEM_DEVICE_TO_PINNED (fxc_fx_rows, _u8, _iq only): x is device memory, `out` lies in memory from fxc_host_alloc -- the
 * finishing kernel writes the rows there across PCIe (no copy, no wait: the call is asynchronous on the plan's stream like any
 * device call; the rows are complete when the stream reaches that point, fxc_sync or an event of the caller's).  For the
 * time-series product of device-resident samples (one row per chunk pair, effex.py:402-410, 687-696). */
enum fxc_mem_kind { FXC_MEM_HOST = 0, FXC_MEM_DEVICE = 1, FXC_MEM_DEVICE_TO_PINNED = 2 };
enum fxc_mode { FXC_MODE_SPECTRUM = 0, FXC_MODE_CONTINUUM = 1 }; /* TEST == CONTINUUM arithmetic */
/* sample formats of the fxc_*_iq entry points: complex64 (what the path computes in), the receivers' interleaved
 * unsigned 8-bit I,Q (pyrtlsdr's packed bytes, effex.py:652), and complex128 (the reference's own sample type,
 * effex.py:109-110: narrowed to complex64 on the device, after the DC removal when that is asked for) */
enum fxc_iq_format { FXC_IQ_C64 = 0, FXC_IQ_U8 = 1, FXC_IQ_C128 = 2 };
enum fxc_path {
    FXC_PATH_GENERIC = 0, /* any shape (nchan <= 16384, ntaps <= 32, 2..64 antennas).  Channel counts that are not a
                             power of two (effex.py:733-739: --resolution is a free integer) and nchan 4 / 8 / 16384 when the
                             path is chosen automatically: FIR + mixed-radix Stockham FFT (chirp-z for large prime factors)
                             in one kernel, which with 2 antennas also multiplies and integrates; otherwise FIR / radix-2
                             FFT / X kernels through a workspace (the tests' independent reference when forced)      */
    FXC_PATH_FUSED = 1,   /* nchan 4096, ntaps 4, 2 antennas (one kernel) or 4/6/8 (F-only kernel + X-engine) */
    FXC_PATH_STREAM = 2,  /* nchan 1, 2 antennas: the continuum streaming limit                             */
    FXC_PATH_TILED = 3    /* nchan 512/1024/2048/4096/8192, any ntaps: 2 antennas in one fused F+X kernel, 3..64
                             via its F-only variant + X-engine; nchan 16/32/64/128/256, ntaps <= 4: the same
                             design inside one wave (2 antennas, or 3..64 via its F-only variant)           */
};

typedef struct fxc_info {
    int32_t n_ant, n_baselines, nchan, ntaps;
    int64_t num_samp, n_pts;   /* n_pts = num_samp / nchan spectra per chunk (effex.py:553)     */
    int32_t path;              /* fxc_path actually used by fxc_fx_* for this configuration      */
    int32_t grid, block;       /* launch geometry of the dominant kernel                         */
    int32_t lds_bytes;         /* dynamic LDS of the dominant kernel                             */
    int32_t device, cu_count;
    int64_t workspace_bytes;
    int32_t specialised;       /* bit 0: the F+X kernel was compiled for exactly this channel count when the plan was made
                                  (two antennas, a channel count that is not a power of two, up to four taps); bit 1: so was
                                  the F stage alone (built at the first fxc_channelize / multi-antenna call); bit 2: and the second
                                  pass of two antennas above 4096 channels (antenna 1's F stage multiplied with antenna 0's spectra) */
    int32_t spec_vgprs;        /* its vector registers per lane                                                          */
    int32_t spec_source;       /* where its code object came from: 0 none, 1 built by hiprtc when the plan was made, 2 the cache of
                                  earlier builds (FXC_RTC_CACHE), 3 pre-built beside the library (rtc_prebuilt/, made at build time
                                  for a stated list of channel counts: effex_amd/build.py::PREBUILT)                     */
    float   spec_seconds;      /* what getting it took when the plan was made (hiprtc: seconds; a file: milliseconds)      */
} fxc_info;

int         fxc_version(void);
int         fxc_dev_kernels(void); /* 0: the shipped library; 1: the developer build with the reference / A-B kernels as well */
int         fxc_device_count(int* count);
const char* fxc_status_string(int status);

/* Plan = the configuration Correlator.__init__ fixes once (effex.py:109-127): antenna count,
 * nbins, ntaps, num_samp and the PFB window (float64 design, used as float32 on the device).
 * `window` is host memory, [ntaps*nchan] doubles, copied.  `stream` is the hipStream_t to issue
 * work on: the caller's stream (e.g. torch.cuda.current_stream().cuda_stream, so the plan's kernels
 * are ordered with the caller's own work on that stream), NULL for HIP's default (null) stream, or
 * FXC_STREAM_OWNED for a private non-blocking stream the caller then orders against with fxc_sync().
 * force_path: -1 = choose automatically, else an fxc_path (FUSED fails if the shape has none). */
int fxc_plan_create(fxc_plan** out, int device, int n_ant, int nchan, int ntaps, int64_t num_samp,
                    const double* window, void* stream, int force_path);
int fxc_plan_destroy(fxc_plan* plan); /* FXC_ERR_STATE while an fxc_pipe still uses the plan */
/* Move the plan to another hipStream_t (e.g. when the caller's current stream changes, `with torch.cuda.stream(s)`):
 * everything already queued on the old stream is ordered before what follows on the new one by an event, no host
 * wait.  Not for plans that own their stream. */
int fxc_set_stream(fxc_plan* plan, void* stream);
int fxc_plan_get_info(const fxc_plan* plan, fxc_info* info);
/* Diagnostic: build the specialised kernel (fxc_info.specialised) for `nchan` channels and `ntaps` taps -- variant 0: F+X from
 * complex64 samples, 1: F+X from the receivers' bytes, 2: the F stage alone (fxc_channelize; the F pass of 3 and more
 * antennas), 3: the second pass of two antennas above 4096 channels -- for the device architecture `arch` ("gfx950"; NULL: the current
 * device's), without a device and without a plan: the library's embedded kernel source through hiprtc.  FXC_OK and a one-line
 * description in `report` (may be NULL: "nchan= ntaps= tpr= slots= frames_per_step= stages= lds_bytes= code_bytes= vgprs= scratch=
 * resident= lean= rows= groups= pads= plane0= twfull= waves=" -- threads per frame, frames side by side in a workgroup, frames per step,
 * the stage radices in order, whether taps and twiddles come from tables (above 2048 channels), streams per workgroup, rows per work
 * item of each stage, the LDS layout of the stage buffers, the largest radix whose twiddles all stay in registers, waves per SIMD the
 * registers were held to); FXC_ERR_UNSUPPORTED when the shape has no such kernel (plans of that shape run the
 * any-shape kernel); FXC_ERR_HIP with the compiler's log in fxc_last_error(NULL) when the build fails.
 * (The reference takes any integer --resolution, effex.py:733-739; this is where the build meets that freedom.) */
int fxc_spec_probe(int nchan, int ntaps, int variant, const char* arch, char* report, int report_bytes);
const char* fxc_last_error(const fxc_plan* plan);

/* rot[k] = exp(+2*pi*i*f_k*tau), natural bin order — effex.py:516,519.  Formed by the caller in
 * float64 (phase ~ 9e3 rad), complex128[nchan] host memory, copied.  Default: all ones. */
int fxc_set_rot(fxc_plan* plan, const double* rot_re_im);

/* F-stage only — replaces cusignal.filtering.channelize_poly + .T at effex.py:553 (and the
 * complex128 copy at :551).  x = [n_streams][num_samp] complex64, out = [n_streams][n_pts][nchan]
 * complex64, natural (un-shifted) bin order; trailing num_samp mod nchan samples ignored; zero
 * PFB history at the start of every stream. */
int fxc_channelize(fxc_plan* plan, const void* x, void* out, int64_t n_streams, int mem_kind);

/* F+X, integrate — replaces effex.py:508-521 for a batch of chunks: x = [n_chunks][n_ant][num_samp]
 * complex64.  Adds sum_chunks sum_i spec_a[i,k]*conj(spec_b[i,k]) (natural bin order, no rot, no
 * scale) into the plan's float64 accumulator [n_baselines][nchan], baselines ordered
 * (0,1),(0,2)..(A-2,A-1), and n_chunks*n_pts into its spectra counter. */
int fxc_fx_accumulate(fxc_plan* plan, const void* x, int64_t n_chunks, int mem_kind);

/* F+X, one visibility row per chunk — the reference's literal _run_task() output (effex.py:490-527):
 *   SPECTRUM : out = [n_chunks][n_baselines][nchan] complex64 = fftshift(mean_i(f_a*conj(f_b*rot)))
 *   CONTINUUM: out = [n_chunks][n_baselines] complex128      = mean_k(that) / bandwidth  (:523-524)
 * `out` has the same mem_kind as `x` (or is pinned host memory: FXC_MEM_DEVICE_TO_PINNED). */
int fxc_fx_rows(fxc_plan* plan, const void* x, void* out, int64_t n_chunks, int mem_kind, int mode,
                double bandwidth);

/* Accumulator access for the multi-GPU reduce.  fxc_acc_export writes
 * [n_baselines*nchan] complex128 raw sums followed by one complex128 whose real part is the
 * spectra count into device memory `sums_dev` (n_baselines*nchan + 1 complex128); the host sums
 * those buffers across ranks (torch.distributed / RCCL all-reduce) and hands the result to
 * fxc_finalize_sums on the root. */
int fxc_acc_reset(fxc_plan* plan);
int fxc_acc_export(fxc_plan* plan, void* sums_dev);
/* sums_dev == NULL: the plan's own copy, as fxc_reduce leaves it (FXC_ERR_STATE if fxc_reduce has not run) */
int fxc_finalize_sums(fxc_plan* plan, const void* sums_dev, void* out_host, int mode, double bandwidth);

/* The reduce itself: export the plan's accumulator and sum it over the ranks of `rccl_comm`
 * (made by fxc_comm_create; NULL = single rank) with one ncclReduce to `root` (root < 0: ncclAllReduce), float64, in place, on
 * the plan's stream -- no host synchronisation between the F+X kernels, the collective and fxc_finalize_sums(plan,
 * NULL, ...) on the root.  64 KiB for two antennas; latency-bound over xGMI.
 * fxc_comm_*: the communicator for it.  Rank 0 calls fxc_comm_unique_id and hands the FXC_COMM_ID_BYTES bytes to
 * every rank by any channel (bench.py: torch.distributed broadcast); every rank then calls fxc_comm_create (blocking,
 * collective: ncclCommInitRank on `device`).  librccl is bound at run time; FXC_ERR_COMM if it cannot be.
 * fxc_reduce refuses (FXC_ERR_ARG, nothing queued) a communicator made on another device than the plan's, and a root
 * outside its world: a collective entered on the wrong device leaves the other ranks waiting in theirs. */
#define FXC_COMM_ID_BYTES 128
int fxc_comm_unique_id(void* id_out);
int fxc_comm_create(void** rccl_comm_out, int device, int rank, int world_size, const void* id);
int fxc_comm_destroy(void* rccl_comm);
int fxc_reduce(fxc_plan* plan, void* rccl_comm, int root);

/* What a communicator says about itself, so that a multi-GPU result can carry the proof of the ranks RCCL saw (the
 * reference has no counterpart: its chunks are merely independent, effex.py:391-410).  *_seen are asked of the live
 * ncclComm_t (ncclCommCount / ncclCommUserRank / ncclCommCuDevice; -1 where the bound RCCL lacks the query), *_given are
 * the arguments fxc_comm_create was called with; rccl_version = ncclGetVersion (e.g. 22105); async_error =
 * ncclCommGetAsyncError (0 = ncclSuccess); reduces = collectives fxc_reduce has queued on this communicator.
 * fxc_comm_probe (collective, blocking): one ncclAllReduce in which every rank contributes 1.0 -- *ranks_summed is the
 * number of ranks RCCL itself added up; FXC_ERR_COMM if their rank numbers do not add up to 1 + ... + n.
 * fxc_rccl_version: the version and (path_out, may be NULL) the file name of the librccl that was bound at run time. */
typedef struct fxc_comm_desc {
    int32_t ranks_seen, rank_seen, device_seen;
    int32_t world_given, rank_given, device_given;
    int32_t rccl_version, async_error;
    int64_t reduces;
} fxc_comm_desc;
int fxc_comm_info(void* rccl_comm, fxc_comm_desc* info);
int fxc_comm_probe(void* rccl_comm, int64_t* ranks_summed);
int fxc_rccl_version(int* version, char* path_out, int path_bytes);

/* Single-GPU finalize: mean over everything accumulated, times conj(rot), fftshift; D2H.
 *   SPECTRUM : out_host = [n_baselines][nchan] complex128;  CONTINUUM: [n_baselines] complex128.
 * Waits for the result.  reset != 0 clears the accumulator afterwards.  = fxc_finalize_async + fxc_finalize_wait. */
int fxc_finalize(fxc_plan* plan, void* out_host, int mode, double bandwidth, int reset);

/* The same without the wait: the finalize is queued on the plan's stream -- on the 2-antenna fast paths as part of
 * the kernel that folds the last fx_accumulate call's partial sums into the accumulator: fold, mean, conj(rot),
 * fftshift, reset and the write into pinned host memory are one launch -- and the call returns.  The caller may queue
 * the next integration (fxc_fx_accumulate ...) before it collects the result with fxc_finalize_wait, which blocks on
 * that result's event only.  Up to two results may be outstanding (FXC_ERR_STATE beyond that, and from the blocking
 * finalize calls while any is); they are collected in the order they were queued.
 * fxc_finalize_sums_async: the multi-GPU form (sums as for fxc_finalize_sums; the accumulator is not touched). */
int fxc_finalize_async(fxc_plan* plan, int mode, double bandwidth, int reset);
/* fxc_finalize_async with the destination named up front: the result is delivered into out_host (same layout as
 * fxc_finalize) by the device -- written by the finishing kernel itself when it is small and out_host lies in
 * fxc_host_alloc memory, by the side-stream copy when it is large (28 baselines and more: a direct DMA into pinned memory) --
 * so that fxc_finalize_wait(plan, out_host or NULL) only waits: with 496 baselines of 4 096 bins the copy out of the plan's
 * slot is 32 MB of host memcpy per integration, more than the integration's kernels take.  The buffer must stay valid until
 * the wait returns. */
int fxc_finalize_async_to(fxc_plan* plan, void* out_host, int mode, double bandwidth, int reset);
int fxc_finalize_sums_async(fxc_plan* plan, const void* sums_dev, int mode, double bandwidth);
int fxc_finalize_wait(fxc_plan* plan, void* out_host);
int fxc_finalize_pending(const fxc_plan* plan); /* results queued and not yet collected */

int fxc_sync(fxc_plan* plan);

/* Input conditioning, device resident (both are steps the reference runs on the host
 * just before the path).  Streams are [n_streams][num_samp] with the plan's num_samp; n_streams <= 65535.
 *   fxc_remove_dc : out = x - mean(x) per stream, real and imaginary parts separately — effex.py:394-395
 *                   (complex64 in, complex64 out; out may alias x; means formed in float64).
 *   fxc_convert_u8: RTL-SDR interleaved unsigned 8-bit I,Q -> complex64 (byte - 127.5) / 127.5, what
 *                   pyrtlsdr does for sdr.stream(format='samples') (effex.py:652); with remove_dc != 0 the
 *                   per-stream mean is removed in the same pass from exact integer byte sums. */
int fxc_remove_dc(fxc_plan* plan, const void* x_dev, void* out_dev, int64_t n_streams);
int fxc_convert_u8(fxc_plan* plan, const void* iq_u8_dev, void* out_dev, int64_t n_streams, int remove_dc);

/* F+X straight from the RTL-SDR byte stream: iq_u8 = [n_chunks][n_ant][num_samp] interleaved unsigned 8-bit I,Q
 * (2 bytes per sample), converted as fxc_convert_u8 does (remove_dc != 0: per-stream mean removed, effex.py:394-395)
 * and then processed exactly like fxc_fx_rows / fxc_fx_accumulate.  On fused plans (2 antennas, nchan 4096, ntaps 4)
 * the F+X kernel loads the bytes itself -- a quarter of the complex64 stream's HBM traffic, no intermediate copy;
 * other plans convert into a staging buffer first.  Replaces, for byte sources, the chain pyrtlsdr conversion
 * (effex.py:652) -> host DC removal (effex.py:394-395) -> cp.array copies (effex.py:508-509) -> _pfb_xcorr. */
int fxc_fx_rows_u8(fxc_plan* plan, const void* iq_u8, void* out, int64_t n_chunks, int mem_kind, int mode,
                   double bandwidth, int remove_dc);
int fxc_fx_accumulate_u8(fxc_plan* plan, const void* iq_u8, int64_t n_chunks, int mem_kind, int remove_dc);

/* The same two calls for any sample format, with the per-chunk DC removal of effex.py:394-395 on the device:
 * x = [n_chunks][n_ant][num_samp] samples of `iq_format` (fxc_iq_format); remove_dc != 0 subtracts, per chunk and antenna,
 * the mean of the real and of the imaginary parts (float64 sums) before the path.  FXC_IQ_U8 = fxc_fx_rows_u8;
 * FXC_IQ_C64 with remove_dc == 0 = fxc_fx_rows.  complex64 / complex128 with DC removal run the sums and the subtraction
 * (complex128: subtraction in float64, then one rounding to complex64) as a pre-pass -- in place on the library's own
 * staging copy for host buffers, into a staging buffer for device buffers (the caller's samples are never written).
 * Replaces the host lines effex.py:394-395 (+ the narrowing copy of a complex128 source) in front of _pfb_xcorr. */
int fxc_fx_rows_iq(fxc_plan* plan, const void* x, void* out, int64_t n_chunks, int mem_kind, int mode, double bandwidth,
                   int iq_format, int remove_dc);
int fxc_fx_accumulate_iq(fxc_plan* plan, const void* x, int64_t n_chunks, int mem_kind, int iq_format, int remove_dc);

/* Pinned host memory for FXC_MEM_HOST buffers -- the counterpart of the reference's mapped pinned staging buffers
 * (cusignal.get_shared_mem, effex.py:109-110).  Buffers from fxc_host_alloc cross PCIe by direct DMA (pageable memory goes
 * through the runtime's bounce buffers at about half the rate), and an `out` buffer inside such an allocation is written by
 * the finishing kernel itself through the device's mapping of it: no copy back.  Any host pointer is still accepted
 * everywhere; these only make it fast.  fxc_host_free(NULL) is a no-op; FXC_ERR_ARG for a pointer fxc_host_alloc did not
 * return.  Process-wide, thread-safe; the memory is usable with every device. */
int fxc_host_alloc(void** out, int64_t bytes);
int fxc_host_free(void* ptr);

/* Delay calibration — replaces Correlator._estimate_delay_gaussian, effex.py:583-627:
 * zero-pad both streams, FFT, f0*conj(f1), inverse FFT, arg-max of |xcorr|, 3-point log-Gaussian peak;
 * *delay_s = (n - (imax + delta)) / rate.  iq0, iq1: n complex64 samples each (host or device), any n.
 * Uses the plan's device, stream and workspace; synchronises. */
int fxc_estimate_delay(fxc_plan* plan, const void* iq0, const void* iq1, int64_t n, int mem_kind, double rate,
                       double* delay_s);

/* Host-fed front end: replaces the reference's blocking per-chunk copies
 * (effex.py:391-392, 508-509, 693).  A pipe owns `depth` slots of pinned host staging + device buffers.
 * fxc_pipe_acquire hands the producer the pinned input buffer of the next free slot
 * ([chunks_per_batch][n_ant][num_samp] complex64) to fill in place; fxc_pipe_submit queues
 * H2D -> fxc_fx_rows -> D2H on three streams chained by events and returns; fxc_pipe_push = acquire + memcpy
 * from any host memory + submit; fxc_pipe_pop waits for the oldest batch and copies its rows out (layout as
 * fxc_fx_rows).  With depth >= 2 batch k+1 crosses PCIe while batch k computes.  acquire/submit/push fail with
 * FXC_ERR_STATE when `depth` batches are in flight, pop when none is.
 * The pipe uses the plan's stream and workspace: do not interleave other fxc_fx_* calls on the plan. */
int fxc_pipe_create(fxc_pipe** out, fxc_plan* plan, int64_t chunks_per_batch, int depth, int mode, double bandwidth);
/* the same pipe fed with RTL-SDR bytes: batches are [chunks_per_batch][n_ant][num_samp] interleaved uint8 I,Q and go
 * through fxc_fx_rows_u8 (a quarter of the PCIe traffic of complex64 samples) */
int fxc_pipe_create_u8(fxc_pipe** out, fxc_plan* plan, int64_t chunks_per_batch, int depth, int mode, double bandwidth,
                       int remove_dc);
/* any sample format (fxc_iq_format), batches through fxc_fx_rows_iq: complex64 / complex128 recordings with the DC
 * removal of effex.py:394-395 on the device instead of a host pass per chunk */
int fxc_pipe_create_iq(fxc_pipe** out, fxc_plan* plan, int64_t chunks_per_batch, int depth, int mode, double bandwidth,
                       int iq_format, int remove_dc);
int fxc_pipe_acquire(fxc_pipe* pipe, void** in_host);
int fxc_pipe_submit(fxc_pipe* pipe);
int fxc_pipe_push(fxc_pipe* pipe, const void* x_host);
int fxc_pipe_pop(fxc_pipe* pipe, void* out_host);
int fxc_pipe_in_flight(const fxc_pipe* pipe);
int fxc_pipe_destroy(fxc_pipe* pipe);

/* Measurement hooks (bench.py): HIP events on the plan's stream.  fxc_timer_* bracket a region;
 * with kernel profiling on, every launch of the dominant kernel is bracketed by its own event
 * pair and fxc_kernel_time returns the summed duration and launch count since the last reset. */
int fxc_timer_start(fxc_plan* plan);
int fxc_timer_stop(fxc_plan* plan, double* elapsed_ms);
int fxc_kernel_profiling(fxc_plan* plan, int enable);
int fxc_kernel_time(fxc_plan* plan, double* total_ms, int64_t* launches, int reset);

/* Deterministic synthetic IQ straight into HBM (same arithmetic as effex_amd/synth.py, bit for
 * bit): x_dev = [n_chunks][n_ant][num_samp] complex64.  delays = n_ant ints (host), tone =
 * complex64[tone_period] (host). */
int fxc_synth_fill(int device, void* stream, void* x_dev, uint64_t seed, int64_t first_chunk,
                   int64_t n_chunks, int n_ant, int64_t num_samp, const int32_t* delays,
                   const float* tone_re_im, int tone_period);

#ifdef __cplusplus
}
#endif
#endif /* FXCORR_H */
