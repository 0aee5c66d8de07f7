// fxcorr.hip — MI355X (gfx950) F/X hot path: the C ABI of include/fxcorr.h over the kernels.
//
// One translation unit.  This file holds the ABI entry points; it includes
//   fx_math.h, fx_fused4096.h, fx_tiled.h, fx_small.h, fx_mixed.h   index maps, butterflies and kernel phases (also compiled by
//                                           g++ for the host emulation under tests/emul)
//   k_generic.h k_finish.h k_fused4096.h k_tiled.h k_small.h k_prepass.h k_stream.h k_conditioning.h k_delay.h k_synth.h
//                                           the __global__ kernels, one file per path / step
//   h_plan.h h_launch.h h_run.h h_rccl.h    fxc_plan, the per-path launchers and workspace passes, the device-resident
//                                           fx_accumulate / fx_rows, the run-time binding of librccl
//
// Replaces, for effex's hot path:
//   cusignal.filtering.channelize_poly FIR half   effex/effex.py:553   -> pfb_fir_kernel / pfb_fft_mixed_kernel / fused phase 1
//   cusignal channelize_poly FFT half + conj      effex/effex.py:553   -> pfb_fft_mixed_kernel (any channel count: mixed radix,
//                                                                          chirp-z) / fft_pow2_kernel / dft_any_kernel / fused phases 1-3
//   f0 * conj(f1 * rot), mean(axis=0), fftshift   effex/effex.py:516-521 -> xmul_kernel / fused X + finish kernels
//   continuum tail mean_k / bandwidth             effex/effex.py:523-524 -> continuum kernels
// and the steps either side of the path:
//   per-chunk DC removal, uint8 -> complex        effex/effex.py:394-395, :652 -> dc_* / convert_u8 kernels
//   delay calibration                             effex/effex.py:583-627 -> delay_* / stockham_stage kernels
//   per-chunk blocking copies                     effex/effex.py:391-392, 508-509, 693 -> fxc_pipe_* (host side)
// Paths: fused (nchan 4096, ntaps 4; 2 antennas in one kernel, 4/6/8 via F-only + X-engine), tiled (2 antennas,
// nchan 512..8192, any ntaps: the fused design generalised, fx_tiled.h; nchan 16..256, ntaps <= 4: the same inside one
// wave, k_small.h), stream (nchan 1), generic (everything else: channel counts that are not a power of two on the mixed-radix
// kernel of k_generic.h / fx_mixed.h -- with two antennas F and X in one pass --, the rest on plain FIR / FFT / X kernels).  Written for gfx950 only: wave64, 160 KiB LDS, v_permlane32_swap, buffer loads.
// No CPU fallback.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>
// RCCL: types and prototypes only -- the library itself is bound at run time (rccl_api), so a ROCm install without the
// rccl headers still builds libfxcorr (single-GPU users need no RCCL at all)
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclFloat64 = 8 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
ncclResult_t ncclGetUniqueId(ncclUniqueId* uniqueId);
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId commId, int rank);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
const char* ncclGetErrorString(ncclResult_t result);
ncclResult_t ncclReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, int root,
                        ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op,
                           ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclCommCount(const ncclComm_t comm, int* count);
ncclResult_t ncclCommUserRank(const ncclComm_t comm, int* rank);
ncclResult_t ncclCommCuDevice(const ncclComm_t comm, int* device);
ncclResult_t ncclGetVersion(int* version);
ncclResult_t ncclCommGetAsyncError(ncclComm_t comm, ncclResult_t* asyncError);
}
#endif

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/fxcorr.h"

// FXC_DEV_KERNELS=1 (libfxcorr_dev.so, built by effex_amd/build.py for the tests and tools only): also the kernels and knobs
// that exist to check or time the shipped ones against -- the direct O(N^2) DFT for channel counts that are not a power of two
// (FXC_GENERIC_FFT=radix2) and the vector X-engine where the matrix-core one serves (FXC_XENGINE=block)
#ifndef FXC_DEV_KERNELS
#define FXC_DEV_KERNELS 0
#endif
namespace {
// an integer from the environment.  The shipped library reads FOUR variables, all documented in include/fxcorr.h: FXC_RTC, FXC_RTC_CACHE,
// FXC_RTC_VERBOSE, FXC_WS_MB.  Every other route / tuning knob exists in the developer library only (libfxcorr_dev.so, -DFXC_DEV_KERNELS=1:
// tests, tools/): FXC_DEV_ENV* below fold to their defaults in the shipped build, strings and all.
int env_int(const char* name, int dflt) {
    const char* e = std::getenv(name);
    return e ? std::atoi(e) : dflt;
}
#define FXC_DEV_ENV(name) (FXC_DEV_KERNELS ? std::getenv(name) : static_cast<const char*>(nullptr))
#define FXC_DEV_ENV_INT(name, dflt) (FXC_DEV_KERNELS ? env_int(name, dflt) : (dflt))
}  // namespace
#include "fx_fused4096.h"
#include "fx_tiled.h"
#include "fx_small.h"
#include "fx_mixed.h"
#include "fx_math.h"

using fxc::cd;
using fxc::cf;
using fxc::f4;

namespace {

constexpr double kTwoPi = 6.283185307179586476925286766559;
constexpr int kMaxTaps = 32;         // cusignal ships 8x8 / 16x16 / 32x32 channeliser kernels only
constexpr int kMaxXAnt = 64;         // antennas the F-only + X-engine route takes (fxc_plan_create's own limit)
constexpr int kMaxLdsFftN = 16384;   // 128 KiB of complex64 in LDS
constexpr int kBluPrimePerRatio = 45;  // prime factors beyond 45 nfft / N: the chirp-z form (see pfb_fft_mixed_kernel, BLU)
constexpr int kBluMaxNfft = 10240;    // two chirp-z rows in the 160 KiB of LDS: up to 5120 channels
constexpr int kMixedMaxN = 10240;    // two rows of complex64 in the 160 KiB of LDS (pfb_fft_mixed_kernel)
size_t res_direct_bytes() {      // finalize results up to this size are written to host memory by the kernel (FXC_RES_DIRECT: developer knob, bytes)
    static const size_t v = [] { const char* e = FXC_DEV_ENV("FXC_RES_DIRECT"); return e ? (size_t)std::atoll(e) : (size_t)(256 << 10); }();
    return v;
}
// upper bound of the lazily grown workspace (288 GB of HBM per GPU): a call over more chunks than fit runs in passes.
// FXC_WS_MB: developer / test knob, the bound in MiB (tests/test_gpu_finish.py forces many passes with it)
int64_t ws_target() {
    static const int64_t v = [] {
        const char* e = std::getenv("FXC_WS_MB");
        const long long mb = e ? std::atoll(e) : 0;
        return mb > 0 ? (int64_t)mb << 20 : (int64_t)12 << 30;
    }();
    return v;
}

}  // namespace

#include "k_generic.h"
#include "k_finish.h"
#include "k_xmfma.h"
#include "k_fused4096.h"
#include "k_tiled.h"
#include "k_small.h"
#include "k_prepass.h"
#include "k_stream.h"
#include "k_conditioning.h"
#include "k_delay.h"
#include "k_synth.h"
#include "h_plan.h"
#include "h_rtc.h"
#include "h_launch.h"
#include "h_run.h"
#include "h_rccl.h"


// ------------------------------------------------------------------------------------------
// C ABI (include/fxcorr.h)
// ------------------------------------------------------------------------------------------
extern "C" {

int fxc_version(void) { return FXC_VERSION; }
int fxc_dev_kernels(void) { return FXC_DEV_KERNELS; }

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
    if (p->s_copy) (void)hipStreamSynchronize(p->s_copy);   // an uncollected large result may still be on its way to h_res[]
    for (auto& e : p->kev) {
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    void* bufs[] = {p->d_win, p->d_tw, p->d_rot, p->d_win4, p->d_tw1, p->d_tw2, p->d_tw0, p->d_tw_small, p->d_stamps,
                    p->d_acc, p->d_sums, p->d_cont, p->d_rowpart, p->d_ws, p->d_stage[0], p->d_stage[1], p->d_stage[2], p->d_dc, p->d_hpre,
                    p->d_ones, p->d_pre, p->d_tw8192, p->d_chirp, p->d_blud};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    if (p->ev_t0) (void)hipEventDestroy(p->ev_t0);
    if (p->ev_t1) (void)hipEventDestroy(p->ev_t1);
    if (p->ev_order) (void)hipEventDestroy(p->ev_order);
    for (int k = 0; k < fxc_plan::kResSlots; ++k) {
        if (p->ev_res[k]) (void)hipEventDestroy(p->ev_res[k]);
        if (p->h_res[k]) (void)hipHostFree(p->h_res[k]);
        if (p->d_res_big[k]) (void)hipFree(p->d_res_big[k]);
    }
    if (p->s_copy) {
        (void)hipStreamSynchronize(p->s_copy);
        (void)hipStreamDestroy(p->s_copy);
    }
    if (p->ev_fin) (void)hipEventDestroy(p->ev_fin);
    if (p->own_stream && p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
    return FXC_OK;
}

// float64 transform of any length on the host, kernel exp(+2 pi i j k / n): decimation in time over the smallest prime factor,
// a direct sum for a prime length (the chirp-z table: 7-smooth lengths up to 8192, built once per plan)
static std::vector<cd> host_dft(const std::vector<cd>& x) {
    const int n = (int)x.size();
    if (n == 1) return x;
    int p1 = n;
    for (int q = 2; q * q <= n; ++q)
        if (n % q == 0) {
            p1 = q;
            break;
        }
    std::vector<cd> out((size_t)n);
    if (p1 == n) {
        for (int k = 0; k < n; ++k) {
            double ar = 0.0, ai = 0.0;
            for (int j = 0; j < n; ++j) {
                const double ph = kTwoPi * (double)(((int64_t)j * k) % n) / (double)n;
                const double wr = std::cos(ph), wi = std::sin(ph);
                ar += x[j].x * wr - x[j].y * wi;
                ai += x[j].x * wi + x[j].y * wr;
            }
            out[k].x = ar;
            out[k].y = ai;
        }
        return out;
    }
    const int m = n / p1;                                   // x[p1 j + r] -> p1 transforms of m points
    std::vector<std::vector<cd>> sub((size_t)p1);
    for (int r = 0; r < p1; ++r) {
        std::vector<cd> part((size_t)m);
        for (int j = 0; j < m; ++j) part[j] = x[(size_t)p1 * j + r];
        sub[r] = host_dft(part);
    }
    for (int k = 0; k < n; ++k) {
        double ar = 0.0, ai = 0.0;
        for (int r = 0; r < p1; ++r) {
            const double ph = kTwoPi * (double)(((int64_t)r * k) % n) / (double)n;
            const double wr = std::cos(ph), wi = std::sin(ph);
            const cd v = sub[r][k % m];
            ar += v.x * wr - v.y * wi;
            ai += v.x * wi + v.y * wr;
        }
        out[k].x = ar;
        out[k].y = ai;
    }
    return out;
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
    // 16 .. 256 channels, up to four taps: the wave-local variant of the tiled design (k_small.h) -- 2 antennas in one
    // F+X kernel, 3 .. 64 through its F-only variant + X-engine
    // (32-bit frame counters in the kernel; more than four taps: behind the pre-filter pass, whose buffer descriptors bound the stream)
    const bool small_n = small_nchan(N) && p->num_samp / N < (1ll << 31) && (T <= 4 || p->num_samp <= (1ll << 27));
    const bool small_shape = small_n && p->n_ant >= 2 && p->n_ant <= kMaxXAnt;
    const bool tiled_shape = small_shape || (p->n_ant >= 2 && p->n_ant <= kMaxXAnt && tiled_nchan(N) && p->num_samp <= (1ll << 27));
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

    // generic F stage: every channel count that is not a power of two (and fits two LDS rows) takes the mixed-radix kernel;
    // FXC_GENERIC_FFT=mixed|radix2 moves the powers of two onto it / everything off it (developer knob)
    {
        const char* gf = FXC_DEV_ENV("FXC_GENERIC_FFT");
        const bool force_mixed = gf && !std::strcmp(gf, "mixed");
        // (off the powers of two "radix2" means the direct DFT, a kernel only the developer build has)
        const bool force_old = gf && !std::strcmp(gf, "radix2") && (p->pow2 || FXC_DEV_KERNELS);
        // powers of two on the automatic path that no tuned kernel takes -- 4, 8 and 16384 channels -- ride along; a forced
        // generic path keeps the radix-2 kernels (the tests' independent reference)
        const bool auto_pow2 = force_path == -1 && (N == 4 || N == 8 || N == 16384);
        p->mixed = N > 1 && N <= kMaxLdsFftN && !force_old && (!p->pow2 || force_mixed || auto_pow2);
        if (p->mixed) {
            p->mixed_plan = fxc::mixed_factor(N);
            // (512 threads per row beyond 1320 channels, where three 256-thread workgroups stop fitting a CU's LDS: two antennas
            // 1350 ... 2000 channels 18 - 20 % faster, F only with two frames per slot 10 - 25 %; 1120 ... 1300 slower)
            p->mixed_tpr = fxc::mixed_threads_per_row(N, FXC_DEV_ENV_INT("FXC_MIXED_TPR", 1024), p->n_ant == 2, FXC_DEV_ENV_INT("FXC_MIXED_WIDE_FROM", 1320));
            if (p->mixed_plan.n_stages < 0) p->mixed = false;
        }
        if (p->mixed) {
            // a large prime factor costs more as an O(N p) stage than the whole transform as a chirp-z convolution
            int pmax = 1;
            for (int st = 0; st < p->mixed_plan.n_stages; ++st) pmax = std::max(pmax, p->mixed_plan.radix[st]);
            // the convolution length: any 7-smooth number from 2 N - 1 up to the next power of two -- the one whose stages cost
            // least (points x a weight per stage from the measured stage times), not the power of two itself (2049 channels:
            // 4116 = 4 3 7 7 7 points instead of 8192)
            int m_pow2 = 1;
            while (m_pow2 < 2 * N - 1) m_pow2 <<= 1;
            int m = m_pow2;
            if (FXC_DEV_ENV_INT("FXC_BLU_SMOOTH", 1)) {
                double best = 1e300;
                for (int cand = 2 * N - 1; cand <= std::min(m_pow2, kBluMaxNfft); ++cand) {
                    int rest = cand;
                    double w = 0.0;
                    for (int q : {4, 2, 3, 5, 7}) {
                        const double wq = q == 4 ? 1.0 : q == 2 ? 0.8 : q == 3 ? 1.0 : q == 5 ? 1.5 : 2.0;
                        while (rest % q == 0) {
                            rest /= q;
                            w += wq;
                        }
                    }
                    if (rest == 1 && w * cand < best) {
                        best = w * cand;
                        m = cand;
                    }
                }
            }
            // measured (profiles/r04/experiments.md §7): the stage costs ~p, the chirp-z rows ~nfft / N; they cross near p = 45 nfft / N
            const int p_min = FXC_DEV_ENV_INT("FXC_BLU_MIN_PRIME", (int)((int64_t)kBluPrimePerRatio * m / N));
            if (pmax > p_min && m <= kBluMaxNfft) {
                p->mixed_blu = true;
                p->blu_nfft = m;
                p->mixed_plan = fxc::mixed_factor(m);
                p->mixed_tpr = fxc::mixed_threads_per_row(m, 1024);
            }
        }
        p->rtc = env_int("FXC_RTC", 1) != 0;      // (read once, when the plan is made)
        if (p->mixed && p->rtc && N <= 8192 && T <= 4 && !p->d_win4) {
            // the lean builds of fx_spec.h (above 2048 channels, or with a prime factor of 17 ... 23) read a point's taps as one quad from
            // L2: [N][4], zeros beyond T
            std::vector<f4> w4((size_t)N);
            for (int m = 0; m < N; ++m) {
                f4 w;
                w.x = wf[m];
                w.y = T > 1 ? wf[(size_t)1 * N + m] : 0.f;
                w.z = T > 2 ? wf[(size_t)2 * N + m] : 0.f;
                w.w = T > 3 ? wf[(size_t)3 * N + m] : 0.f;
                w4[(size_t)m] = w;
            }
            FXC_HIP(p, hipMalloc(&p->d_win4, w4.size() * sizeof(f4)));
            FXC_HIP(p, hipMemcpy(p->d_win4, w4.data(), w4.size() * sizeof(f4), hipMemcpyHostToDevice));
        }
        p->mixed_xeng = p->mixed && p->n_ant >= 3 && FXC_DEV_ENV_INT("FXC_MIXED_XENGINE", 1);
        if (p->mixed && !p->mixed_blu && p->n_ant == 2 && FXC_DEV_ENV_INT("FXC_MIXED_XF", 1)) {
            const size_t rpw = (size_t)(std::max(256, p->mixed_tpr) / p->mixed_tpr);
            // four rows (two antennas x ping-pong) must fit the LDS, with the twiddle table beside them (up to 4096 channels) or
            // without (up to 5120)
            p->mixed_xf = rpw * 4 * (size_t)N * sizeof(cf) <= (size_t)(160 * 1024) && N <= kMixedXPoints * p->mixed_tpr;
            p->mixed_xf_twl = (rpw * 4 + 1) * (size_t)N * sizeof(cf) <= (size_t)(160 * 1024) && FXC_DEV_ENV_INT("FXC_MIXED_TWLDS", 1);
            // (without the table in LDS that kernel has no register butterflies for 11 / 13: such channel counts go through the
            // F-only kernel, which has, and xmul_kernel)
            if (!p->mixed_xf_twl && fxc::mixed_rows_per_slot_cap(p->mixed_plan) == 1) p->mixed_xf = false;
            // the kernel built for exactly this channel count (fx_spec.h through hiprtc, h_rtc.h): every sample fetched once,
            // strides and trip counts compile-time constants.  FXC_RTC=0 keeps the any-shape kernel (developer knob, and what
            // a box without hiprtc runs)
            if (p->mixed_xf && p->rtc && p->num_samp < (1ll << 28)) {       // (32-bit byte offsets inside a chunk)
                if (!spec_first_radices(N, T).empty()) {
                    const SpecKernel* k = spec_kernel(p->device, N, T, kSpecC64);
                    if (k->fn)
                        p->spec = k;
                    else if (env_int("FXC_RTC_VERBOSE", 0))
                        std::fprintf(stderr, "libfxcorr: no specialised kernel for %d channels: %s\n", N, k->error.c_str());
                }
            }
            // above 4096 channels there is no such kernel (sixteen points of two antennas: 256 registers of ring), but the F stage
            // alone has one: complex64 input takes it, and xmul_kernel (h_launch.h::mixed_one_pass)
            if (p->mixed_xf && !p->spec && N > 4096 && T <= 4 && p->rtc && p->num_samp < (1ll << 28) && FXC_DEV_ENV_INT("FXC_MIXED_XF_BYTES_ONLY", 1) &&
                !spec_first_radices(N, T, spec_rows(N, kSpecFOnly)).empty()) {
                const SpecKernel* k = spec_kernel(p->device, N, T, kSpecFOnly);
                p->spec_f_tried = true;
                p->spec_f = k->fn ? k : nullptr;
                p->xf_bytes_only = p->spec_f != nullptr;
            }
        }
    }
    // generic FFT twiddles exp(+2 pi i j / N): [N/2] for the radix-2 kernel, [N] for the mixed-radix kernel and the direct DFT
    if (N > 1) {
        const int tn = p->mixed_blu ? p->blu_nfft : N;         // the chirp-z rows transform blu_nfft points
        const int cnt = (p->pow2 && !p->mixed) ? N / 2 : tn;
        std::vector<cf> tw((size_t)cnt);
        for (int jx = 0; jx < cnt; ++jx) {
            const double ph = kTwoPi * (double)jx / (double)tn;
            tw[jx] = fxc::mk((float)std::cos(ph), (float)std::sin(ph));
        }
        FXC_HIP(p, hipMalloc(&p->d_tw, tw.size() * sizeof(cf)));
        FXC_HIP(p, hipMemcpy(p->d_tw, tw.data(), tw.size() * sizeof(cf), hipMemcpyHostToDevice));
    }

    if (p->mixed_blu) {
        // chirp c[n] = exp(+i pi n^2 / N) with n^2 reduced mod 2N (exact), and D = FFT_M(d) / M for d[m] = d[M - m] = conj(c[m]),
        // m < N, zero elsewhere; kernel exp(+2 pi i j k / M), float64 radix-2 on the host
        const int M = p->blu_nfft;
        std::vector<cd> c((size_t)N);
        for (int n = 0; n < N; ++n) {
            const double ph = kTwoPi / 2.0 * (double)(((int64_t)n * n) % (2 * (int64_t)N)) / (double)N;
            c[n].x = std::cos(ph);
            c[n].y = std::sin(ph);
        }
        std::vector<cd> d((size_t)M);
        for (auto& v : d) v.x = v.y = 0.0;
        for (int m = 0; m < N; ++m) {
            d[m].x = c[m].x;
            d[m].y = -c[m].y;
            if (m) d[M - m] = d[m];
        }
        d = host_dft(d);
        std::vector<cf> cfl((size_t)N), dfl((size_t)M);
        for (int n = 0; n < N; ++n) cfl[n] = fxc::mk((float)c[n].x, (float)c[n].y);
        for (int k = 0; k < M; ++k) dfl[k] = fxc::mk((float)(d[k].x / M), (float)(d[k].y / M));
        FXC_HIP(p, hipMalloc(&p->d_chirp, cfl.size() * sizeof(cf)));
        FXC_HIP(p, hipMemcpy(p->d_chirp, cfl.data(), cfl.size() * sizeof(cf), hipMemcpyHostToDevice));
        FXC_HIP(p, hipMalloc(&p->d_blud, dfl.size() * sizeof(cf)));
        FXC_HIP(p, hipMemcpy(p->d_blud, dfl.data(), dfl.size() * sizeof(cf), hipMemcpyHostToDevice));
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
    FXC_HIP(p, hipMemset(p->d_sums, 0, (acc_n + 1) * sizeof(cd)));
    // finalize results: pinned host memory the finishing kernels write through the device's mapping of it (coherent,
    // so the host sees the bytes once the slot's event has completed)
    for (int k = 0; k < fxc_plan::kResSlots; ++k) {
        FXC_HIP(p, hipHostMalloc(reinterpret_cast<void**>(&p->h_res[k]), std::max<size_t>(acc_n, 16) * sizeof(cd),
                                 hipHostMallocMapped | hipHostMallocCoherent));
        FXC_HIP(p, hipHostGetDevicePointer(reinterpret_cast<void**>(&p->d_res[k]), p->h_res[k], 0));
        FXC_HIP(p, hipEventCreateWithFlags(&p->ev_res[k], hipEventReleaseToSystem));
    }

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
        if (const char* e = FXC_DEV_ENV("FXC_FUSED_SEG")) p->fused_seg = std::max<int64_t>(1, std::atoll(e));   // developer knob
        FXC_HIP(p, hipFuncSetAttribute(reinterpret_cast<const void*>(&fx_fused4096_kernel<false, false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
        FXC_HIP(p, hipFuncSetAttribute(reinterpret_cast<const void*>(&fx_fused4096_kernel<true, false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
        FXC_HIP(p, hipFuncSetAttribute(reinterpret_cast<const void*>(&fx_fused4096_kernel<false, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
        FXC_HIP(p, hipFuncSetAttribute(reinterpret_cast<const void*>(&fx_fused4096_kernel<false, true, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes + kDckLdsBytes));
    }
    p->small = small_shape && p->n_ant == 2 && p->path == FXC_PATH_TILED;
    p->small_f = small_n && force_path != FXC_PATH_GENERIC;
    if (p->small || p->small_f) {
        const int P = N / 16;
        // more than four taps: pfb_prefilter_kernel applies the FIR first (k_prepass.h), the wave-local kernel then runs with
        // one unit tap -- the route the tiled channel counts take (FXC_PREFILTER=1: developer knob, the same at <= 4 taps)
        const char* pre_env_s = FXC_DEV_ENV("FXC_PREFILTER");
        p->prefilter = (T > 4 || (pre_env_s && std::atoi(pre_env_s) == 1));
        if (p->prefilter) {
            p->pre_tp = T <= 8 ? 8 : (T <= 16 ? 16 : 32);
            std::vector<float> hp((size_t)p->pre_tp * N, 0.f);
            for (int t = 0; t < T; ++t)
                for (int n = 0; n < N; ++n) hp[(size_t)t * N + n] = wf[(size_t)t * N + (N - 1 - n)];
            FXC_HIP(p, hipMalloc(&p->d_hpre, hp.size() * sizeof(float)));
            FXC_HIP(p, hipMemcpy(p->d_hpre, hp.data(), hp.size() * sizeof(float), hipMemcpyHostToDevice));
        }
        std::vector<f4> w4((size_t)N);          // window quads [r P + u] = h[t N + u + P r], t = x, y, z, w (zero beyond ntaps)
        for (int r = 0; r < 16; ++r)
            for (int u = 0; u < P; ++u) {
                const int m = u + P * r;
                f4 w;
                w.x = p->prefilter ? 1.f : wf[m];
                w.y = (T > 1 && !p->prefilter) ? wf[(size_t)1 * N + m] : 0.f;
                w.z = (T > 2 && !p->prefilter) ? wf[(size_t)2 * N + m] : 0.f;
                w.w = (T > 3 && !p->prefilter) ? wf[(size_t)3 * N + m] : 0.f;
                w4[(size_t)r * P + u] = w;
            }
        FXC_HIP(p, hipMalloc(&p->d_win4, w4.size() * sizeof(f4)));
        FXC_HIP(p, hipMemcpy(p->d_win4, w4.data(), w4.size() * sizeof(f4), hipMemcpyHostToDevice));
        std::vector<cf> tw((size_t)N);
        for (int u = 0; u < P; ++u)
            for (int k1 = 0; k1 < 16; ++k1) {
                const double ph = kTwoPi * (double)((u * k1) % N) / (double)N;
                tw[(size_t)u * 16 + k1] = fxc::mk((float)std::cos(ph), (float)std::sin(ph));
            }
        FXC_HIP(p, hipMalloc(&p->d_tw_small, tw.size() * sizeof(cf)));
        FXC_HIP(p, hipMemcpy(p->d_tw_small, tw.data(), tw.size() * sizeof(cf), hipMemcpyHostToDevice));
        const int rc = small_setup(p);
        if (rc) return rc;
    }
    p->tiled_f = p->small_f || (tiled_nchan(N) && p->num_samp <= (1ll << 27) && force_path != FXC_PATH_GENERIC);
    if (!small_nchan(N) && (p->path == FXC_PATH_TILED || p->tiled_f)) {
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
        const char* pre_env = FXC_DEV_ENV("FXC_PREFILTER");
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
        const char* split_env = FXC_DEV_ENV("FXC_SPLIT8192");
        // ... up to four taps: two passes instead (f8192_ring_kernel and its XM form, h_launch.h::tiled_raw_sums; FXC_X8192=0: off)
        const bool want_x8192 = N == 8192 && p->n_ant == 2 && T <= 4 && p->path == FXC_PATH_TILED && p->num_samp < (1ll << 28) &&
                                FXC_DEV_ENV_INT("FXC_X8192", 1) && FXC_DEV_ENV_INT("FXC_F8192", 1);
        p->split8192 = (N == 8192 && p->n_ant == 2 && T <= 16 && p->path == FXC_PATH_TILED && p->num_samp <= (1ll << 27) &&
                        !(split_env && std::atoi(split_env) == 0) && !want_x8192);
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
        // 8192 channels, up to four taps: the F stage alone has a ring kernel of its own (k_tiled.h::f8192_ring_kernel), fed with the
        // same window quads from L2.  FXC_F8192=0: the pair kernel (developer knob)
        p->f8192 = N == 8192 && T <= 4 && !p->prefilter && FXC_DEV_ENV_INT("FXC_F8192", 1);
        p->x8192 = want_x8192 && p->f8192;
        if ((p->tiled_ring || p->f8192) && (!p->d_win4 || p->prefilter)) {
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
    if (p->n_ant >= 3 && p->n_ant <= kMaxXAnt) {
        const void* xfn = nullptr;
        switch (p->n_ant) {
            case 3: xfn = reinterpret_cast<const void*>(&xengine_kernel<3>); break;
            case 4: xfn = reinterpret_cast<const void*>(&xengine_kernel<4>); break;
            case 5: xfn = reinterpret_cast<const void*>(&xengine_kernel<5>); break;
            case 6: xfn = reinterpret_cast<const void*>(&xengine_kernel<6>); break;
            case 7: xfn = reinterpret_cast<const void*>(&xengine_kernel<7>); break;
            case 8: xfn = reinterpret_cast<const void*>(&xengine_kernel<8>); break;
            default: xfn = reinterpret_cast<const void*>(&xengine_block_kernel); break;
        }
        // more than 8 antennas: the matrix-core X-engine (k_xmfma.h) unless FXC_XENGINE=block (developer knob: the vector
        // kernel over blocks of 8 antennas it replaced)
        const char* xe = FXC_DEV_ENV("FXC_XENGINE");
        p->x_mfma = p->n_ant > kXB && !(FXC_DEV_KERNELS && xe && std::string(xe) == "block");
        if (p->x_mfma) {
            int per_cu = 0, threads = 0, lds = 0;
            FXC_XMFMA_DISPATCH(p, {
                xfn = reinterpret_cast<const void*>(&xengine_mfma_kernel<XT>);
                threads = XMfmaGeo<XT>::kThreads;
                lds = XMfmaGeo<XT>::kLdsBytes;
            });
            FXC_HIP(p, hipFuncSetAttribute(xfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            FXC_HIP(p, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, xfn, threads, lds));
            p->x_resident = (int64_t)std::max(per_cu, 1) * p->cu_count;
        } else {
        // one-wave workgroups resident per CU: the occupancy API, bounded by the register file (512 VGPRs per SIMD lane in
        // granules of 8, at most 8 waves per SIMD) -- the API has been seen one block per CU high (MI355X_MICROARCH.md),
        // and a launch sized one wave per CU too large would run a second round for that sliver
        int per_cu = 0;
        FXC_HIP(p, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, xfn, kXThreads, 0));
        hipFuncAttributes fa;
        FXC_HIP(p, hipFuncGetAttributes(&fa, xfn));
        const int regs = std::max(8, (fa.numRegs + 7) / 8 * 8);
        per_cu = std::min(per_cu, 4 * std::min(8, 512 / regs));
        p->x_resident = (int64_t)std::max(per_cu, 1) * p->cu_count;
        }
    }
    if (p->mixed) {
        const int lds_max = 160 * 1024;
        FXC_HIP(p, hipFuncSetAttribute(reinterpret_cast<const void*>(&pfb_fft_mixed_kernel<true, 1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
        FXC_HIP(p, hipFuncSetAttribute(reinterpret_cast<const void*>(&pfb_fft_mixed_kernel<true, 2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
        FXC_HIP(p, hipFuncSetAttribute(reinterpret_cast<const void*>(&pfb_fft_mixed_kernel<false, 1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
        FXC_HIP(p, hipFuncSetAttribute(reinterpret_cast<const void*>(&pfb_fft_mixed_kernel<true, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
        FXC_HIP(p, hipFuncSetAttribute(reinterpret_cast<const void*>(&pfb_fft_mixed_kernel<true, 2, true, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
        FXC_HIP(p, hipFuncSetAttribute(reinterpret_cast<const void*>(&pfb_fft_mixed_kernel<false, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
        FXC_HIP(p, hipFuncSetAttribute(reinterpret_cast<const void*>(&pfb_fft_mixed_kernel<false, 2, true, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
        FXC_HIP(p, hipFuncSetAttribute(reinterpret_cast<const void*>(&pfb_fft_mixed_kernel<false, 1, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
        FXC_HIP(p, hipFuncSetAttribute(reinterpret_cast<const void*>(&pfb_fft_mixed_kernel<true, 1, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
        FXC_HIP(p, hipFuncSetAttribute(reinterpret_cast<const void*>(&pfb_fft_mixed_kernel<false, 1, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
    } else if (N > 1) {
        const int lds = N * (int)sizeof(cf);
        if (p->pow2)
            FXC_HIP(p, hipFuncSetAttribute(reinterpret_cast<const void*>(&fft_pow2_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds));
#if FXC_DEV_KERNELS
        else
            FXC_HIP(p, hipFuncSetAttribute(reinterpret_cast<const void*>(&dft_any_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds));
#endif
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
    } else if (p->path == FXC_PATH_TILED && small_nchan(p->nchan)) {
        info->grid = p->small_wgs;
        info->block = 256;
        info->lds_bytes = p->nchan * (int)sizeof(f4) + 4 * 1088 * (int)sizeof(cf);
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
    if (p->x8192) {       // 8192 channels, two antennas, two passes: f8192_ring_kernel and its XM form
        info->grid = p->cu_count * 8;
        info->block = kF8192Threads;
        info->lds_bytes = kF8192LdsCf * (int)sizeof(cf);
    }
    if (p->spec_f) info->specialised |= 2;
    if (p->spec_xm) info->specialised |= 4;
    if (p->spec) {
        info->specialised |= 1;
        info->spec_vgprs = p->spec->vgprs;
        info->spec_source = p->spec->source;
        info->spec_seconds = (float)p->spec->seconds;
        info->grid = p->cu_count * p->spec->wgs_per_cu;
        info->block = p->spec->shape.threads();
        info->lds_bytes = (int)p->spec->shape.lds_bytes();
    } else if (p->spec_f) {      // (no one-pass build: where the F-only build -- and the second pass behind it -- came from, what both took)
        info->spec_vgprs = p->spec_f->vgprs;
        info->spec_source = p->spec_f->source;
        info->spec_seconds = (float)(p->spec_f->seconds + (p->spec_xm ? p->spec_xm->seconds : 0.0));
    }
    info->device = p->device;
    info->cu_count = p->cu_count;
    info->workspace_bytes = p->ws_bytes;
    return FXC_OK;
}

int fxc_spec_probe(int nchan, int ntaps, int variant, const char* arch, char* report, int report_bytes) {
    if (variant < 0 || variant > 3) return fail(nullptr, FXC_ERR_ARG, "variant %d: 0 complex64 F+X, 1 bytes F+X, 2 F only, 3 second pass (F x another stream's spectra)", variant);
    if (report && report_bytes > 0) report[0] = 0;
    if (spec_first_radices(nchan, ntaps, spec_rows(nchan, variant)).empty())
        return fail(nullptr, FXC_ERR_UNSUPPORTED, "no specialised kernel for %d channels, %d taps", nchan, ntaps);
    std::string target = arch ? arch : "";
    if (target.empty()) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
            return fail(nullptr, FXC_ERR_NODEVICE, "no HIP device to take the architecture from");
        target = spec_arch(prop.gcnArchName);
    }
    const SpecBuild b = spec_search(nchan, ntaps, variant, target.c_str());
    if (b.image.empty()) return fail(nullptr, b.scratch ? FXC_ERR_UNSUPPORTED : FXC_ERR_HIP, "%s", b.error.c_str());
    if (report && report_bytes > 0) {
        const SpecShape& sh = b.shape;
        std::snprintf(report, (size_t)report_bytes,
                      "nchan=%d ntaps=%d tpr=%d slots=%d frames_per_step=%d stages=%s lds_bytes=%zu code_bytes=%zu vgprs=%lld scratch=%lld resident=%d lean=%d rows=%d "
                      "groups=%s pads=%s plane0=%d twfull=%d waves=%d source=%s",
                      sh.n, sh.taps, sh.tpr, sh.slots, sh.u, sh.list(sh.radix).c_str(), sh.lds_bytes(), b.image.size(), b.vgprs, b.scratch, b.resident,
                      (int)sh.lean, sh.rows, sh.list(sh.grp).c_str(), sh.list(sh.pad).c_str(), sh.plane0, sh.twfull, sh.waves,
                      b.source == kSpecPrebuilt ? "prebuilt" : b.source == kSpecCached ? "cache" : "built");
    }
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
    fxc_comm* c = new (std::nothrow) fxc_comm();
    if (!c) {
        (void)api->comm_destroy(comm);
        return fail(nullptr, FXC_ERR_NOMEM, "host allocation failed");
    }
    c->comm = comm;
    c->device = device;
    c->rank = rank;
    c->world_size = world_size;
    *rccl_comm_out = c;
    return FXC_OK;
}

int fxc_comm_destroy(void* rccl_comm) {
    if (!rccl_comm) return FXC_OK;
    fxc_comm* c = static_cast<fxc_comm*>(rccl_comm);
    if (c->magic != fxc_comm::kMagic) return fail(nullptr, FXC_ERR_ARG, "not a communicator made by fxc_comm_create");
    RcclApi* api = rccl_api();
    if (!api->handle) return fail(nullptr, FXC_ERR_COMM, "%s", api->error.c_str());
    DeviceGuard device_guard__(c->device);
    const ncclResult_t r = api->comm_destroy(c->comm);
    c->magic = 0;
    delete c;
    if (r != ncclSuccess) return rccl_fail(nullptr, api, "ncclCommDestroy", r);
    return FXC_OK;
}

int fxc_rccl_version(int* version, char* path_out, int path_bytes) {
    if (!version) return fail(nullptr, FXC_ERR_ARG, "version is NULL");
    *version = 0;
    if (path_out && path_bytes > 0) path_out[0] = 0;
    RcclApi* api = rccl_api();
    if (!api->handle) return fail(nullptr, FXC_ERR_COMM, "%s", api->error.c_str());
    if (api->get_version) {
        const ncclResult_t r = api->get_version(version);
        if (r != ncclSuccess) return rccl_fail(nullptr, api, "ncclGetVersion", r);
    }
    if (path_out && path_bytes > 0) std::snprintf(path_out, (size_t)path_bytes, "%s", api->path.c_str());
    return FXC_OK;
}

int fxc_comm_info(void* rccl_comm, fxc_comm_desc* info) {
    if (!rccl_comm || !info) return fail(nullptr, FXC_ERR_ARG, "NULL argument");
    fxc_comm* c = static_cast<fxc_comm*>(rccl_comm);
    if (c->magic != fxc_comm::kMagic) return fail(nullptr, FXC_ERR_ARG, "not a communicator made by fxc_comm_create");
    RcclApi* api = rccl_api();
    if (!api->handle) return fail(nullptr, FXC_ERR_COMM, "%s", api->error.c_str());
    std::memset(info, 0, sizeof *info);
    info->ranks_seen = info->rank_seen = info->device_seen = info->async_error = -1;   // -1: this RCCL has no such query
    info->world_given = c->world_size;
    info->rank_given = c->rank;
    info->device_given = c->device;
    info->reduces = c->reduces;
    // asked of the live ncclComm_t, not echoed from the arguments of fxc_comm_create
    ncclResult_t r = ncclSuccess;
    int v = 0;
    if (api->comm_count) {
        if ((r = api->comm_count(c->comm, &v)) != ncclSuccess) return rccl_fail(nullptr, api, "ncclCommCount", r);
        info->ranks_seen = v;
    }
    if (api->comm_user_rank) {
        if ((r = api->comm_user_rank(c->comm, &v)) != ncclSuccess) return rccl_fail(nullptr, api, "ncclCommUserRank", r);
        info->rank_seen = v;
    }
    if (api->comm_cu_device) {
        if ((r = api->comm_cu_device(c->comm, &v)) != ncclSuccess) return rccl_fail(nullptr, api, "ncclCommCuDevice", r);
        info->device_seen = v;
    }
    if (api->get_version && api->get_version(&v) == ncclSuccess) info->rccl_version = v;
    if (api->comm_async_error) {
        ncclResult_t async = ncclSuccess;
        if (api->comm_async_error(c->comm, &async) == ncclSuccess) info->async_error = (int)async;
    }
    return FXC_OK;
}

int fxc_comm_probe(void* rccl_comm, int64_t* ranks_summed) {
    if (!rccl_comm || !ranks_summed) return fail(nullptr, FXC_ERR_ARG, "NULL argument");
    *ranks_summed = 0;
    fxc_comm* c = static_cast<fxc_comm*>(rccl_comm);
    if (c->magic != fxc_comm::kMagic) return fail(nullptr, FXC_ERR_ARG, "not a communicator made by fxc_comm_create");
    RcclApi* api = rccl_api();
    if (!api->handle) return fail(nullptr, FXC_ERR_COMM, "%s", api->error.c_str());
    FXC_DEVICE(nullptr, c->device);
    // every rank puts in 1.0 (and its rank + 1 in a second slot): what comes back was added up by RCCL itself
    double* d = nullptr;
    hipStream_t s = nullptr;
    FXC_HIP(nullptr, hipMalloc(reinterpret_cast<void**>(&d), 2 * sizeof(double)));
    int rc = FXC_OK;
    double h[2] = {1.0, (double)(c->rank + 1)};
    hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMemcpyAsync(d, h, sizeof h, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        const ncclResult_t r = api->all_reduce(d, d, 2, ncclFloat64, ncclSum, c->comm, s);
        if (r != ncclSuccess) rc = rccl_fail(nullptr, api, "ncclAllReduce", r);
    }
    if (e == hipSuccess && rc == FXC_OK) e = hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && rc == FXC_OK) e = hipStreamSynchronize(s);
    if (s) (void)hipStreamDestroy(s);
    (void)hipFree(d);
    if (rc) return rc;
    if (e != hipSuccess) return fail(nullptr, FXC_ERR_HIP, "fxc_comm_probe: %s", hipGetErrorString(e));
    const double n = h[0];
    if (h[1] != n * (n + 1) / 2) return fail(nullptr, FXC_ERR_COMM, "fxc_comm_probe: %g ranks answered but their ranks sum to %g", n, h[1]);
    *ranks_summed = (int64_t)n;
    return FXC_OK;
}

int fxc_reduce(fxc_plan* p, void* rccl_comm, int root) {
    if (!p) return fail(p, FXC_ERR_ARG, "NULL plan");
    fxc_comm* c = static_cast<fxc_comm*>(rccl_comm);
    if (c) {
        // checked before anything is queued: a collective on the wrong device or with a root no rank has would leave
        // the other ranks waiting in theirs
        if (c->magic != fxc_comm::kMagic) return fail(p, FXC_ERR_ARG, "not a communicator made by fxc_comm_create");
        if (c->device != p->device)
            return fail(p, FXC_ERR_ARG, "the communicator was made on device %d, the plan is on device %d", c->device, p->device);
        if (root >= c->world_size) return fail(p, FXC_ERR_ARG, "root %d outside the communicator's world of %d", root, c->world_size);
    }
    FXC_DEVICE(p, p->device);
    int rc = fxc_acc_export(p, p->d_sums);
    if (rc) return rc;
    p->sums_valid = true;
    if (!c) return FXC_OK;                  // single rank: the exported sums are the reduced sums
    RcclApi* api = rccl_api();
    if (!api->handle) return fail(p, FXC_ERR_COMM, "%s", api->error.c_str());
    // raw float64 sums + the spectra count, in place, ordered on the plan's stream behind the export
    const size_t count = 2 * ((size_t)p->n_base * p->nchan + 1);
    const ncclResult_t r = root < 0 ? api->all_reduce(p->d_sums, p->d_sums, count, ncclFloat64, ncclSum, c->comm, p->stream)
                                    : api->reduce(p->d_sums, p->d_sums, count, ncclFloat64, ncclSum, root, c->comm, p->stream);
    if (r != ncclSuccess) return rccl_fail(p, api, root < 0 ? "ncclAllReduce" : "ncclReduce", r);
    ++c->reduces;
    return FXC_OK;
}

int fxc_sync(fxc_plan* p) {
    if (!p) return fail(p, FXC_ERR_ARG, "NULL plan");
    FXC_DEVICE(p, p->device);
    const int rf = flush_pending(p);        // the fold of the last fx_accumulate pass belongs to "everything queued"
    if (rf) return rf;
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

namespace {
// FXC_MEM_DEVICE_TO_PINNED: the rows of device-resident samples go straight into fxc_host_alloc memory -- the finishing
// kernel writes them across PCIe through the device's mapping of the block, nothing is copied and nothing waits
int pinned_rows_out(fxc_plan* p, void** out, int* mem_kind, int64_t n_chunks, int mode) {
    if (*mem_kind != FXC_MEM_DEVICE_TO_PINNED) return FXC_OK;
    if (!p || !*out || n_chunks <= 0) {
        *mem_kind = FXC_MEM_DEVICE;         // (the entry's own argument checks answer)
        return FXC_OK;
    }
    const size_t ob = mode == FXC_MODE_SPECTRUM ? (size_t)n_chunks * p->n_base * p->nchan * sizeof(cf) : (size_t)n_chunks * p->n_base * sizeof(cd);
    void* d = pinned_device_ptr(*out, ob);
    if (!d) return fail(p, FXC_ERR_ARG, "`out` of FXC_MEM_DEVICE_TO_PINNED (%zu bytes) is not inside memory from fxc_host_alloc", ob);
    *out = d;
    *mem_kind = FXC_MEM_DEVICE;
    return FXC_OK;
}
}  // namespace

int fxc_fx_rows(fxc_plan* p, const void* x, void* out, int64_t n_chunks, int mem_kind, int mode, double bandwidth) {
    if (const int rp = pinned_rows_out(p, &out, &mem_kind, n_chunks, mode)) return rp;
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
    p->pend.valid = false;                  // rows not folded yet are simply dropped
    FXC_HIP(p, hipMemsetAsync(p->d_acc, 0, (size_t)p->n_base * p->nchan * sizeof(cd), p->stream));
    p->spectra_count = 0.0;
    return FXC_OK;
}

int fxc_acc_export(fxc_plan* p, void* sums_dev) {
    if (!p || !sums_dev) return fail(p, FXC_ERR_ARG, "NULL argument");
    FXC_DEVICE(p, p->device);
    const FoldFinish fin = {static_cast<cd*>(sums_dev), nullptr, p->d_rot, p->spectra_count, 0};
    return flush_pending(p, &fin);
}

namespace {

// Queue the finalize of `sums_src` (exported, possibly cross-rank reduced sums) or, sums_src == nullptr, of the plan's
// accumulator -- then together with the fold of the rows still pending and with the reset, in one kernel -- into the
// next result slot; the slot's event marks the host copy complete.
// user_out != nullptr (fxc_finalize_async_to): the result goes to that host buffer instead of the plan's pinned slot -- by
// the finishing kernel itself when it is small and lies in fxc_host_alloc memory, by the side-stream copy when it is large
// (a direct DMA for pinned memory) -- and fxc_finalize_wait copies nothing.
int finalize_enqueue(fxc_plan* p, const cd* sums_src, int mode, double bandwidth, int reset, void* user_out = nullptr) {
    if (mode != FXC_MODE_SPECTRUM && mode != FXC_MODE_CONTINUUM) return fail(p, FXC_ERR_ARG, "bad mode %d", mode);
    if (mode == FXC_MODE_CONTINUUM && !(bandwidth > 0.0)) return fail(p, FXC_ERR_ARG, "bandwidth must be > 0");
    if (p->res_head - p->res_tail >= fxc_plan::kResSlots)
        return fail(p, FXC_ERR_STATE, "%d finalize results outstanding: collect one with fxc_finalize_wait first",
                    fxc_plan::kResSlots);
    if (!sums_src && !(p->spectra_count > 0.0)) return fail(p, FXC_ERR_STATE, "nothing accumulated");
    const int slot = (int)(p->res_head % fxc_plan::kResSlots);
    const int64_t n = (int64_t)p->n_base * p->nchan;
    const size_t bytes = mode == FXC_MODE_SPECTRUM ? (size_t)n * sizeof(cd) : (size_t)p->n_base * sizeof(cd);
    // small results are written into the pinned slot by the finishing kernel itself; large ones go through device
    // memory and a copy on a side stream, off the F+X stream's critical path
    const bool big = bytes > res_direct_bytes();
    if (big && !p->s_copy) {
        FXC_HIP(p, hipStreamCreateWithFlags(&p->s_copy, hipStreamNonBlocking));
        FXC_HIP(p, hipEventCreateWithFlags(&p->ev_fin, hipEventDisableTiming));
        for (int k = 0; k < fxc_plan::kResSlots; ++k) FXC_HIP(p, hipMalloc(&p->d_res_big[k], (size_t)n * sizeof(cd)));
    }
    cd* out = big ? p->d_res_big[slot] : p->d_res[slot];
    cd* const user_mapped = (user_out && !big) ? static_cast<cd*>(pinned_device_ptr(user_out, bytes)) : nullptr;
    if (user_mapped) out = user_mapped;
    // where fxc_finalize_wait finds the bytes: the caller's buffer (nothing to copy) or the plan's slot
    p->res_user[slot] = (user_out && (big || user_mapped)) ? user_out : nullptr;
    p->res_dst[slot] = user_out;
    if (!sums_src) {
        // SPECTRUM: one kernel.  CONTINUUM needs the mean over the bins of the finished accumulator: export, then reduce
        FoldFinish fin = {nullptr, out, p->d_rot, p->spectra_count, reset ? 1 : 0};
        if (mode == FXC_MODE_CONTINUUM) {
            // into a buffer of its own: d_sums may hold reduced sums (fxc_reduce) that fxc_finalize_sums(plan, NULL) has yet
            // to read, and only fxc_reduce makes that copy valid
            if (!p->d_cont) FXC_HIP(p, hipMalloc(&p->d_cont, ((size_t)n + 1) * sizeof(cd)));
            fin.sums = p->d_cont;
            fin.out = nullptr;
            sums_src = p->d_cont;
        }
        // the slot's event rides on the last kernel's own completion (hipExtLaunchKernelGGL): an event recorded behind it is
        // a packet of its own in the stream, and the next F+X kernel starts 11 us later for it
        const int rc = flush_pending(p, &fin, (mode == FXC_MODE_SPECTRUM && !big) ? p->ev_res[slot] : nullptr);
        if (rc) return rc;
        if (reset) p->spectra_count = 0.0;
    } else if (mode == FXC_MODE_SPECTRUM) {
        const int rc = flush_pending(p);
        if (rc) return rc;
        hipExtLaunchKernelGGL(finalize_spectrum_kernel, dim3(grid_for(n, 256, p->cu_count)), dim3(256), 0, p->stream, nullptr,
                              big ? nullptr : p->ev_res[slot], 0, sums_src, out, p->d_rot, p->nchan, p->n_base);
    }
    if (mode == FXC_MODE_CONTINUUM)
        hipExtLaunchKernelGGL(finalize_continuum_kernel, dim3(p->n_base), dim3(256), 0, p->stream, nullptr,
                              big ? nullptr : p->ev_res[slot], 0, sums_src, out, p->d_rot, p->nchan, p->n_base, 1.0 / bandwidth);
    FXC_HIP(p, hipGetLastError());
    if (big) {
        FXC_HIP(p, hipEventRecord(p->ev_fin, p->stream));
        FXC_HIP(p, hipStreamWaitEvent(p->s_copy, p->ev_fin, 0));
        FXC_HIP(p, hipMemcpyAsync(user_out ? user_out : static_cast<void*>(p->h_res[slot]), out, bytes, hipMemcpyDeviceToHost, p->s_copy));
        FXC_HIP(p, hipEventRecord(p->ev_res[slot], p->s_copy));
    }
    p->res_bytes[slot] = bytes;
    p->res_head += 1;
    return FXC_OK;
}

}  // namespace

int fxc_finalize_async(fxc_plan* p, int mode, double bandwidth, int reset) {
    if (!p) return fail(p, FXC_ERR_ARG, "NULL plan");
    FXC_DEVICE(p, p->device);
    return finalize_enqueue(p, nullptr, mode, bandwidth, reset);
}

int fxc_finalize_async_to(fxc_plan* p, void* out_host, int mode, double bandwidth, int reset) {
    if (!p || !out_host) return fail(p, FXC_ERR_ARG, "NULL argument");
    FXC_DEVICE(p, p->device);
    return finalize_enqueue(p, nullptr, mode, bandwidth, reset, out_host);
}

int fxc_finalize_sums_async(fxc_plan* p, const void* sums_dev, int mode, double bandwidth) {
    if (!p) return fail(p, FXC_ERR_ARG, "NULL plan");
    if (!sums_dev) {                         // what fxc_reduce left in the plan
        if (!p->sums_valid) return fail(p, FXC_ERR_STATE, "no reduced sums in the plan: call fxc_reduce first");
        sums_dev = p->d_sums;
    }
    FXC_DEVICE(p, p->device);
    return finalize_enqueue(p, static_cast<const cd*>(sums_dev), mode, bandwidth, 0);
}

int fxc_finalize_wait(fxc_plan* p, void* out_host) {
    if (!p) return fail(p, FXC_ERR_ARG, "NULL plan");
    if (p->res_head == p->res_tail) return fail(p, FXC_ERR_STATE, "no finalize result outstanding");
    const int slot = (int)(p->res_tail % fxc_plan::kResSlots);
    void* const dst = p->res_dst[slot];       // fxc_finalize_async_to: the buffer named when the result was queued
    if (dst ? (out_host && out_host != dst) : !out_host)
        return fail(p, FXC_ERR_ARG, dst ? "this result was queued with fxc_finalize_async_to: pass that buffer or NULL" : "out_host is NULL");
    FXC_DEVICE(p, p->device);
    FXC_HIP(p, hipEventSynchronize(p->ev_res[slot]));
    if (!p->res_user[slot]) std::memcpy(dst ? dst : out_host, p->h_res[slot], p->res_bytes[slot]);
    p->res_tail += 1;
    return FXC_OK;
}

int fxc_finalize_pending(const fxc_plan* p) { return p ? (int)(p->res_head - p->res_tail) : 0; }

int fxc_finalize_sums(fxc_plan* p, const void* sums_dev, void* out_host, int mode, double bandwidth) {
    if (!p || !out_host) return fail(p, FXC_ERR_ARG, "NULL argument");
    if (p->res_head != p->res_tail) return fail(p, FXC_ERR_STATE, "asynchronous finalize results outstanding");
    const int rc = fxc_finalize_sums_async(p, sums_dev, mode, bandwidth);
    if (rc) return rc;
    return fxc_finalize_wait(p, out_host);
}

int fxc_finalize(fxc_plan* p, void* out_host, int mode, double bandwidth, int reset) {
    if (!p || !out_host) return fail(p, FXC_ERR_ARG, "NULL argument");
    if (p->res_head != p->res_tail) return fail(p, FXC_ERR_STATE, "asynchronous finalize results outstanding");
    const int rc = fxc_finalize_async(p, mode, bandwidth, reset);
    if (rc) return rc;
    return fxc_finalize_wait(p, out_host);
}

// workgroups per stream of the subtract / narrow pass: enough to fill the chip when a call has few streams (one chunk
// pair: the reference's own call), a handful when it has thousands
static int cond_slices(const fxc_plan* p, int64_t n_streams) {
    return (int)std::max<int64_t>(1, std::min<int64_t>(256, ((int64_t)p->cu_count * 8 + n_streams - 1) / n_streams));
}
// slice sums per stream (each a workgroup): 32 for batches, up to 256 when a call has a handful of streams (one chunk pair:
// 64 workgroups of 32 sequential loads each took 7 us for 4 MiB)
static int sum_slices(const fxc_plan* p, int64_t n_streams) {
    return (int)std::max<int64_t>(32, std::min<int64_t>(256, (int64_t)p->cu_count * 4 / n_streams));
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
    hipLaunchKernelGGL(dc_sum_c64_kernel, dim3(n_slices, (unsigned)n_streams), dim3(256), 0, p->stream,
                       static_cast<const cf*>(x_dev), part, p->num_samp, n_slices);
    hipLaunchKernelGGL(dc_apply_c64_kernel, dim3(cond_slices(p, n_streams), (unsigned)n_streams), dim3(256), 0, p->stream,
                       static_cast<const cf*>(x_dev), static_cast<cf*>(out_dev), part, p->num_samp, cond_slices(p, n_streams),
                       n_slices, 1);
    FXC_HIP(p, hipGetLastError());
    return FXC_OK;
}

namespace {
// bytes -> complex64 of n_streams streams: a few thousand workgroups, each on one slice of a stream at a time
void launch_convert_u8(fxc_plan* p, const unsigned char* x8, cf* out, const double* part, int n_slices, int64_t n_streams, int remove_dc) {
    const unsigned gy = (unsigned)std::min<int64_t>(n_streams, 65535);
    const int64_t per_stream = std::max<int64_t>(1, ((int64_t)p->cu_count * 16 + gy - 1) / gy);
    const unsigned gx = (unsigned)std::max<int64_t>(1, std::min<int64_t>(per_stream, (p->num_samp + 1023) / 1024));
    hipLaunchKernelGGL(convert_u8_kernel, dim3(gx, gy), dim3(256), 0, p->stream, x8, out, part, p->num_samp, n_slices, n_streams,
                       remove_dc ? 1 : 0);
}
}  // namespace

int fxc_convert_u8(fxc_plan* p, const void* iq_u8_dev, void* out_dev, int64_t n_streams, int remove_dc) {
    int rc = conditioning_common(p, n_streams, iq_u8_dev, out_dev);
    if (rc || n_streams == 0) return rc;
    FXC_DEVICE(p, p->device);
    const int n_slices = 32;
    rc = ensure_ws(p, n_streams * n_slices * 2 * (int64_t)sizeof(double));
    if (rc) return rc;
    double* part = static_cast<double*>(p->d_ws);
    if (remove_dc)
        hipLaunchKernelGGL(dc_sum_u8_stream_kernel, dim3((unsigned)std::min<int64_t>(n_streams * n_slices, (int64_t)p->cu_count * 16)),
                           dim3(256), 0, p->stream, static_cast<const unsigned char*>(iq_u8_dev), part, p->num_samp, n_streams, n_slices);
    launch_convert_u8(p, static_cast<const unsigned char*>(iq_u8_dev), static_cast<cf*>(out_dev), part, n_slices, n_streams, remove_dc);
    FXC_HIP(p, hipGetLastError());
    return FXC_OK;
}

namespace {

// uint8 I,Q in: fused plans (2 antennas: nchan 4096 / ntaps 4, the tiled ring and wave-local kernels, the mixed-radix F + X
// kernel) read the bytes in the F+X kernel itself; every other plan
// converts into a complex64 staging buffer first.  rows: fxc_fx_rows semantics (out != nullptr) or accumulate.
int fx_u8_dev(fxc_plan* p, const unsigned char* x8, void* out, int64_t n_chunks, int mode, double bandwidth, int remove_dc,
              bool rows) {
    constexpr int kSlices = 32;
    const size_t row_elems = mode == FXC_MODE_SPECTRUM ? (size_t)p->n_base * p->nchan * sizeof(cf) : (size_t)p->n_base * sizeof(cd);
    // chunks per pass: at most 65535 streams (a grid dimension of the conditioning kernels), and plans without the
    // fused ingest convert a pass into a complex64 staging buffer that stays within the workspace target
    const bool fused_in = p->n_ant == 2 && !p->prefilter && (p->path == FXC_PATH_FUSED || (p->path == FXC_PATH_TILED && (p->tiled_ring || p->small || p->x8192)) ||
                                                              (p->path == FXC_PATH_GENERIC && p->mixed_xf && FXC_DEV_ENV_INT("FXC_MIXED_U8", 1)));
    int64_t per_pass = std::min<int64_t>(16384, 65535 / p->n_ant);
    if (!fused_in) per_pass = std::min<int64_t>(per_pass, ws_target() / ((int64_t)p->n_ant * p->num_samp * (int64_t)sizeof(cf)));
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
        // The fused 4096-channel kernel can sum the bytes of a workgroup's next chunk while it channelises the current one
        // (k_fused4096.h, DCK): the pre-pass then only covers the first chunk of every workgroup's round-robin share and
        // the tail chunks -- 272 of 10 000 chunk pairs.  Needs whole frames (num_samp % 4096 == 0), 16-byte aligned
        // streams, the default work split, one launch for the pass and at least two rounds of chunks.
        int64_t spec_b, raw_b;
        const int64_t g = p->fused_grid_max;
        // (num_samp <= 2^26: a wave's byte sums are reduced in 32 bits, 16384 frames x 4080 x 64 lanes < 2^32)
        const bool dck = remove_dc && fused_ingest && p->path == FXC_PATH_FUSED && p->fused_seg == 1 &&
                         (p->num_samp % fxc::fused::kN) == 0 && p->num_samp <= (1ll << 26) &&
                         (reinterpret_cast<uintptr_t>(xb) % 16) == 0 &&
                         fused_chunks_per_pass(p, nc, &spec_b, &raw_b) >= nc && nc >= 2 * g;
        if (remove_dc && fused_ingest) {
            const int64_t n_full = dck ? nc / g * g : nc;
            // chunk ranges the pre-pass sums: everything, or [0, g) and [n_full, nc)
            const int64_t lo[2] = {0, n_full}, hi[2] = {dck ? g : nc, dck ? nc : n_full};
            // a call of a few streams (one chunk pair: the reference's own call) is cut into slices to fill the chip
            const int sl = dck ? 1 : (int)std::max<int64_t>(1, std::min<int64_t>(kSlices, (int64_t)p->cu_count * 2 / n_streams));
            for (int r = 0; r < 2; ++r) {
                const int64_t ns = (hi[r] - lo[r]) * p->n_ant;
                if (ns <= 0) continue;
                hipLaunchKernelGGL(dc_sum_u8_stream_kernel, dim3((unsigned)std::min<int64_t>(ns * sl, (int64_t)p->cu_count * 16)),
                                   dim3(256), 0, p->stream, xb + lo[r] * p->n_ant * p->num_samp * 2,
                                   part + lo[r] * p->n_ant * 2 * sl, p->num_samp, ns, sl);
                hipLaunchKernelGGL(dc_offsets_u8_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, p->stream,
                                   part + lo[r] * p->n_ant * 2 * sl, dc + lo[r] * p->n_ant, ns, sl, p->num_samp, 1);
            }
        } else if (remove_dc) {
            hipLaunchKernelGGL(dc_sum_u8_stream_kernel, dim3((unsigned)std::min<int64_t>(n_streams * kSlices, (int64_t)p->cu_count * 16)),
                               dim3(256), 0, p->stream, xb, part, p->num_samp, n_streams, kSlices);
        }
        if (fused_ingest) {
            if (!remove_dc)
                hipLaunchKernelGGL(dc_offsets_u8_kernel, dim3((unsigned)((n_streams + 255) / 256)), dim3(256), 0, p->stream, part,
                                   dc, n_streams, 1, p->num_samp, 0);
            FXC_HIP(p, hipGetLastError());
            p->u8_dck = dck;
            rc = rows ? fx_rows_dev(p, reinterpret_cast<const cf*>(xb), ob, nc, mode, bandwidth, dc)
                      : fx_accumulate_dev(p, reinterpret_cast<const cf*>(xb), nc, dc);
            p->u8_dck = false;
        } else {
            const int64_t total = n_streams * p->num_samp;
            rc = grow(p, &p->d_stage[2], &p->stage_bytes[2], (size_t)total * sizeof(cf));
            if (rc) return rc;
            cf* xc = static_cast<cf*>(p->d_stage[2]);
            launch_convert_u8(p, xb, xc, part, kSlices, n_streams, remove_dc);
            FXC_HIP(p, hipGetLastError());
            rc = rows ? fx_rows_dev(p, xc, ob, nc, mode, bandwidth) : fx_accumulate_dev(p, xc, nc);
        }
        if (rc) return rc;
    }
    return FXC_OK;
}

int fx_u8_entry(fxc_plan* p, const void* iq_u8, void* out, int64_t n_chunks, int mem_kind, int mode, double bandwidth,
                int remove_dc, bool rows) {
    if (rows)
        if (const int rp = pinned_rows_out(p, &out, &mem_kind, n_chunks, mode)) return rp;
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

namespace {

// complex64 with DC removal, or complex128 (narrowed on the device, after the DC removal when asked for): sums, then
// subtract / narrow into a complex64 staging buffer -- or in place when x is the library's own complex64 staging copy of a
// host buffer (x_is_scratch) -- then the plan's usual kernels.  The caller's device buffers are never written.
int fx_cond_dev(fxc_plan* p, const void* x, void* out, int64_t n_chunks, int mode, double bandwidth, int fmt, int remove_dc,
                bool rows, bool x_is_scratch) {
    const size_t in_elem = fmt == FXC_IQ_C128 ? sizeof(cd) : sizeof(cf);
    const size_t row_bytes = mode == FXC_MODE_SPECTRUM ? (size_t)p->n_base * p->nchan * sizeof(cf) : (size_t)p->n_base * sizeof(cd);
    const bool in_place = x_is_scratch && fmt == FXC_IQ_C64;
    // streams per pass: the stream index rides in grid.y, and the staging buffer stays within the workspace target
    int64_t per_pass = 65535 / p->n_ant;
    if (!in_place) per_pass = std::min<int64_t>(per_pass, ws_target() / ((int64_t)p->n_ant * p->num_samp * (int64_t)sizeof(cf)));
    per_pass = std::max<int64_t>(1, std::min<int64_t>(per_pass, n_chunks));
    for (int64_t c0 = 0; c0 < n_chunks; c0 += per_pass) {
        const int64_t nc = std::min<int64_t>(per_pass, n_chunks - c0);
        const int64_t n_streams = nc * p->n_ant;
        const char* xb = static_cast<const char*>(x) + (size_t)c0 * p->n_ant * p->num_samp * in_elem;
        void* ob = rows ? static_cast<char*>(out) + (size_t)c0 * row_bytes : nullptr;
        const int kSlices = sum_slices(p, n_streams);
        int rc = grow(p, &p->d_dc, &p->dc_bytes, (size_t)n_streams * kSlices * 2 * sizeof(double));
        if (rc) return rc;
        double* part = static_cast<double*>(p->d_dc);
        cf* xc = in_place ? reinterpret_cast<cf*>(const_cast<char*>(xb)) : nullptr;
        if (!in_place) {
            rc = grow(p, &p->d_stage[2], &p->stage_bytes[2], (size_t)n_streams * p->num_samp * sizeof(cf));
            if (rc) return rc;
            xc = static_cast<cf*>(p->d_stage[2]);
        }
        const dim3 sum_grid(kSlices, (unsigned)n_streams), app_grid(cond_slices(p, n_streams), (unsigned)n_streams);
        if (fmt == FXC_IQ_C128) {
            if (remove_dc)
                hipLaunchKernelGGL(dc_sum_c128_kernel, sum_grid, dim3(256), 0, p->stream, reinterpret_cast<const cd*>(xb), part,
                                   p->num_samp, kSlices);
            hipLaunchKernelGGL(narrow_c128_kernel, app_grid, dim3(256), 0, p->stream, reinterpret_cast<const cd*>(xb), xc, part,
                               p->num_samp, (int)app_grid.x, kSlices, remove_dc ? 1 : 0);
        } else {
            hipLaunchKernelGGL(dc_sum_c64_kernel, sum_grid, dim3(256), 0, p->stream, reinterpret_cast<const cf*>(xb), part,
                               p->num_samp, kSlices);
            hipLaunchKernelGGL(dc_apply_c64_kernel, app_grid, dim3(256), 0, p->stream, reinterpret_cast<const cf*>(xb), xc, part,
                               p->num_samp, (int)app_grid.x, kSlices, 1);
        }
        FXC_HIP(p, hipGetLastError());
        rc = rows ? fx_rows_dev(p, xc, ob, nc, mode, bandwidth) : fx_accumulate_dev(p, xc, nc);
        if (rc) return rc;
    }
    return FXC_OK;
}

int fx_iq_entry(fxc_plan* p, const void* x, void* out, int64_t n_chunks, int mem_kind, int mode, double bandwidth, int fmt,
                int remove_dc, bool rows) {
    if (fmt == FXC_IQ_U8) return fx_u8_entry(p, x, out, n_chunks, mem_kind, mode, bandwidth, remove_dc, rows);
    if (fmt != FXC_IQ_C64 && fmt != FXC_IQ_C128) return fail(p, FXC_ERR_ARG, "bad iq_format %d", fmt);
    if (fmt == FXC_IQ_C64 && !remove_dc)
        return rows ? fxc_fx_rows(p, x, out, n_chunks, mem_kind, mode, bandwidth) : fxc_fx_accumulate(p, x, n_chunks, mem_kind);
    if (rows)
        if (const int rp = pinned_rows_out(p, &out, &mem_kind, n_chunks, mode)) return rp;
    if (!p) return fail(p, FXC_ERR_ARG, "NULL plan");
    if (p->n_ant < 2) return fail(p, FXC_ERR_ARG, "cross-correlation needs n_ant >= 2");
    if (n_chunks < 0) return fail(p, FXC_ERR_ARG, "n_chunks < 0");
    if (rows && mode != FXC_MODE_SPECTRUM && mode != FXC_MODE_CONTINUUM) return fail(p, FXC_ERR_ARG, "bad mode %d", mode);
    if (rows && mode == FXC_MODE_CONTINUUM && !(bandwidth > 0.0)) return fail(p, FXC_ERR_ARG, "bandwidth must be > 0");
    if (n_chunks == 0) return FXC_OK;
    if (!x || (rows && !out)) return fail(p, FXC_ERR_ARG, "NULL buffer");
    FXC_DEVICE(p, p->device);
    if (mem_kind == FXC_MEM_DEVICE) return fx_cond_dev(p, x, out, n_chunks, mode, bandwidth, fmt, remove_dc, rows, false);
    if (mem_kind != FXC_MEM_HOST) return fail(p, FXC_ERR_ARG, "bad mem_kind %d", mem_kind);
    const size_t xb = (size_t)n_chunks * p->n_ant * p->num_samp * (fmt == FXC_IQ_C128 ? sizeof(cd) : sizeof(cf));
    const size_t ob = !rows ? 0
                            : (mode == FXC_MODE_SPECTRUM ? (size_t)n_chunks * p->n_base * p->nchan * sizeof(cf)
                                                         : (size_t)n_chunks * p->n_base * sizeof(cd));
    return with_host_staging(p, x, xb, out, ob, [&](const cf* dx, void* dout) {
        return fx_cond_dev(p, dx, dout, n_chunks, mode, bandwidth, fmt, remove_dc, rows, true);
    });
}

}  // namespace

int fxc_fx_rows_iq(fxc_plan* p, const void* x, void* out, int64_t n_chunks, int mem_kind, int mode, double bandwidth,
                   int iq_format, int remove_dc) {
    return fx_iq_entry(p, x, out, n_chunks, mem_kind, mode, bandwidth, iq_format, remove_dc, true);
}

int fxc_fx_accumulate_iq(fxc_plan* p, const void* x, int64_t n_chunks, int mem_kind, int iq_format, int remove_dc) {
    return fx_iq_entry(p, x, nullptr, n_chunks, mem_kind, FXC_MODE_SPECTRUM, 1.0, iq_format, remove_dc, false);
}

int fxc_host_alloc(void** out, int64_t bytes) {
    if (!out) return fail(nullptr, FXC_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (bytes <= 0) return fail(nullptr, FXC_ERR_ARG, "bytes must be > 0");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(nullptr, FXC_ERR_NODEVICE, "no HIP device available (pinned memory needs the HIP runtime)");
    void* h = nullptr;
    const hipError_t e = hipHostMalloc(&h, (size_t)bytes, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent);
    if (e != hipSuccess) return fail(nullptr, FXC_ERR_NOMEM, "hipHostMalloc of %lld bytes failed: %s", (long long)bytes, hipGetErrorString(e));
    void* d = nullptr;
    if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess || !d) {
        (void)hipHostFree(h);
        return fail(nullptr, FXC_ERR_HIP, "hipHostGetDevicePointer failed for a fresh pinned block");
    }
    {
        std::lock_guard<std::mutex> lock(g_pinned_mutex);
        g_pinned.push_back({static_cast<char*>(h), static_cast<char*>(d), (size_t)bytes});
    }
    *out = h;
    return FXC_OK;
}

int fxc_host_free(void* ptr) {
    if (!ptr) return FXC_OK;
    {
        std::lock_guard<std::mutex> lock(g_pinned_mutex);
        size_t k = 0;
        while (k < g_pinned.size() && g_pinned[k].host != ptr) ++k;
        if (k == g_pinned.size()) return fail(nullptr, FXC_ERR_ARG, "not a pointer fxc_host_alloc returned");
        g_pinned.erase(g_pinned.begin() + (long)k);
    }
    // hipHostFree waits for the device: nothing queued can still touch the block when it goes
    const hipError_t e = hipHostFree(ptr);
    if (e != hipSuccess) return fail(nullptr, FXC_ERR_HIP, "hipHostFree failed: %s", hipGetErrorString(e));
    return FXC_OK;
}

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
    const int g_len = grid_for(len, 256, p->cu_count);
    hipLaunchKernelGGL(delay_pad_kernel, dim3(g_len), dim3(256), 0, p->stream, x0, a[0], n, len);
    hipLaunchKernelGGL(delay_pad_kernel, dim3(g_len), dim3(256), 0, p->stream, x1, b[0], n, len);
    // one transform = radix-16 passes, then one radix-8 / 4 / 2 pass for the remaining bits of lg
    auto transform = [&](cf* (&buf)[2], int& cur, double sign) {
        int64_t pp = 1;
        for (int bits = lg; bits > 0;) {
            const int r = bits >= 4 ? 4 : bits;
            const int grid = grid_for(len >> r, 256, p->cu_count);
            if (r == 4) hipLaunchKernelGGL(stockham_stage_kernel<16>, dim3(grid), dim3(256), 0, p->stream, buf[cur], buf[cur ^ 1], len, pp, sign);
            else if (r == 3) hipLaunchKernelGGL(stockham_stage_kernel<8>, dim3(grid), dim3(256), 0, p->stream, buf[cur], buf[cur ^ 1], len, pp, sign);
            else if (r == 2) hipLaunchKernelGGL(stockham_stage_kernel<4>, dim3(grid), dim3(256), 0, p->stream, buf[cur], buf[cur ^ 1], len, pp, sign);
            else hipLaunchKernelGGL(stockham_stage_kernel<2>, dim3(grid), dim3(256), 0, p->stream, buf[cur], buf[cur ^ 1], len, pp, sign);
            pp <<= r;
            bits -= r;
            cur ^= 1;
        }
    };
    int cur = 0, cur_b = 0;
    transform(a, cur, -1.0);       // forward transforms, kernel exp(-2 pi i ...) like cp.fft.fft
    transform(b, cur_b, -1.0);
    hipLaunchKernelGGL(mul_conj_kernel, dim3(g_len), dim3(256), 0, p->stream, a[cur], b[cur_b], len);   // f0 * conj(f1)
    transform(a, cur, 1.0);        // inverse transform (un-normalised: the peak fit is scale free)
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

int fxc_pipe_create(fxc_pipe** out, fxc_plan* p, int64_t chunks_per_batch, int depth, int mode, double bandwidth) {
    return fxc_pipe_create_iq(out, p, chunks_per_batch, depth, mode, bandwidth, FXC_IQ_C64, 0);
}

int fxc_pipe_create_u8(fxc_pipe** out, fxc_plan* p, int64_t chunks_per_batch, int depth, int mode, double bandwidth,
                       int remove_dc) {
    return fxc_pipe_create_iq(out, p, chunks_per_batch, depth, mode, bandwidth, FXC_IQ_U8, remove_dc);
}

int fxc_pipe_create_iq(fxc_pipe** out, fxc_plan* p, int64_t chunks_per_batch, int depth, int mode, double bandwidth, int fmt,
                       int remove_dc) {
    if (!out || !p) return fail(p, FXC_ERR_ARG, "NULL argument");
    *out = nullptr;
    if (chunks_per_batch < 1 || depth < 1 || depth > 16) return fail(p, FXC_ERR_ARG, "bad batch size or depth");
    if (p->n_ant < 2) return fail(p, FXC_ERR_ARG, "cross-correlation needs n_ant >= 2");
    if (mode != FXC_MODE_SPECTRUM && mode != FXC_MODE_CONTINUUM) return fail(p, FXC_ERR_ARG, "bad mode %d", mode);
    if (mode == FXC_MODE_CONTINUUM && !(bandwidth > 0.0)) return fail(p, FXC_ERR_ARG, "bandwidth must be > 0");
    if (fmt != FXC_IQ_C64 && fmt != FXC_IQ_U8 && fmt != FXC_IQ_C128) return fail(p, FXC_ERR_ARG, "bad iq_format %d", fmt);
    FXC_DEVICE(p, p->device);
    fxc_pipe* q = new (std::nothrow) fxc_pipe();
    if (!q) return fail(p, FXC_ERR_NOMEM, "host allocation failed");
    q->plan = p;
    q->chunks = chunks_per_batch;
    q->depth = depth;
    q->mode = mode;
    q->bandwidth = bandwidth;
    q->fmt = fmt;
    q->remove_dc = remove_dc;
    q->in_bytes = (size_t)chunks_per_batch * p->n_ant * p->num_samp * (fmt == FXC_IQ_U8 ? 2 : (fmt == FXC_IQ_C128 ? sizeof(cd) : sizeof(cf)));
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
    // the slot's device copy is the pipe's own: complex64 batches are de-meaned in place
    int rc = q->fmt == FXC_IQ_U8 ? fx_u8_dev(p, static_cast<const unsigned char*>(sl.d_in), sl.d_out, q->chunks, q->mode,
                                             q->bandwidth, q->remove_dc, true)
             : (q->fmt == FXC_IQ_C128 || q->remove_dc)
                 ? fx_cond_dev(p, sl.d_in, sl.d_out, q->chunks, q->mode, q->bandwidth, q->fmt, q->remove_dc, true, true)
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
