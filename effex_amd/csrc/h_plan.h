// Part of libfxcorr's single translation unit: included by fxcorr.hip (not a stand-alone header).
#pragma once

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
namespace {

thread_local std::string g_lib_error;


// Pinned host allocations handed out by fxc_host_alloc (process-wide).  with_host_staging looks a caller's pointer up
// here: an output buffer inside one of them is written by the finishing kernel through `dev` (no copy back).
struct PinnedBlock {
    char* host;
    char* dev;       // the same memory as the devices see it
    size_t bytes;
};
std::mutex g_pinned_mutex;
std::vector<PinnedBlock> g_pinned;

// device address of [ptr, ptr + bytes) if it lies inside one fxc_host_alloc block, else nullptr
void* pinned_device_ptr(const void* ptr, size_t bytes) {
    const char* q = static_cast<const char*>(ptr);
    std::lock_guard<std::mutex> lock(g_pinned_mutex);
    for (const PinnedBlock& b : g_pinned)
        if (q >= b.host && q + bytes <= b.host + b.bytes) return b.dev + (q - b.host);
    return nullptr;
}

}  // namespace


// ------------------------------------------------------------------------------------------
// host-fed front end: double-buffered pinned staging, H2D / compute / D2H on three
// streams chained by events, so batch k+1 crosses PCIe while batch k is on the CUs.  Replaces the
// reference's per-chunk blocking copies (effex/effex.py:391-392, 508-509, 693).
// ------------------------------------------------------------------------------------------
struct fxc_pipe_slot {
    void* h_in = nullptr;    // pinned
    void* h_out = nullptr;   // pinned
    void* d_in = nullptr;
    void* d_out = nullptr;
    hipEvent_t ev_in = nullptr, ev_compute = nullptr, ev_out = nullptr;
    bool busy = false;
};

namespace {
struct SpecKernel;      // h_rtc.h
}

struct fxc_plan {
    int device = 0, cu_count = 0;
    int n_ant = 0, n_base = 0, nchan = 0, ntaps = 0;
    int64_t num_samp = 0, n_pts = 0;
    int path = FXC_PATH_GENERIC;
    bool pow2 = false;
    int lg2n = 0;
    bool mixed = false;            // generic F stage = pfb_fft_mixed_kernel (FIR + mixed-radix FFT in one pass)
    fxc::MixedPlan mixed_plan{};
    int mixed_tpr = 256;           // threads per row
    bool mixed_blu = false;        // a large prime factor (plan_build): chirp-z rows of blu_nfft points (F only)
    int blu_nfft = 0;
    cf* d_chirp = nullptr;         // [nchan] exp(+i pi n^2 / nchan)
    cf* d_blud = nullptr;          // [blu_nfft] FFT of the wrapped conjugate chirp / blu_nfft
    bool mixed_xeng = false;       // 3 .. 64 antennas: F-only mixed kernel (antenna-interleaved spectra) + the X-engines
    bool mixed_xf_twl = true;      // ... with the twiddle table in LDS (up to 4096 channels; from L2 up to 5120)
    bool mixed_xf = false;         // two antennas: the same kernel multiplies and integrates too (no spectra in HBM)
    bool xf_bytes_only = false;    // ... for the receivers' bytes only: complex64 input takes the F stage built for the channel count + xmul_kernel
    // ... and, where the shape allows, in the build of fx_spec.h made for exactly this channel count (h_rtc.h); spec_u8: its
    // byte-ingest twin, compiled when bytes first arrive
    const SpecKernel* spec = nullptr;
    const SpecKernel* spec_u8 = nullptr;
    bool spec_u8_tried = false;
    const SpecKernel* spec_f = nullptr;      // the F stage alone (fxc_channelize, 3 + antennas), built on first use
    bool spec_f_tried = false;
    const SpecKernel* spec_xm = nullptr;     // two antennas above 4096 channels: antenna 1's F stage multiplied into sums with antenna 0's spectra (built on first use)
    bool spec_xm_tried = false;
    bool rtc = true;                         // FXC_RTC as it stood when the plan was made (developer knob: 0 keeps the any-shape kernels)
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev_order = nullptr;   // orders the old stream's work before the new one's (fxc_set_stream)
    int live_pipes = 0;              // fxc_pipe objects that hold a pointer to this plan
    // device tables
    float* d_win = nullptr;        // [ntaps*nchan] float (generic)
    cf* d_tw = nullptr;            // generic FFT twiddles
    cd* d_rot = nullptr;           // [nchan]
    f4* d_win4 = nullptr;          // fused
    cf* d_tw1 = nullptr;
    cf* d_tw2 = nullptr;
    cf* d_tw0 = nullptr;           // tiled: pre-stage twiddles [16][nchan/16]
    int tiled_grid_max = 0, tiled_grid_max_f = 0;   // resident workgroups of the F+X / F-only tiled kernels
    bool tiled_f = false;          // the F-only tiled kernel serves fxc_channelize
    bool small = false;            // tiled path, 2 antennas, nchan 16 .. 256: fx_small_ring_kernel (k_small.h);
                                   // tiled_grid_max counts the work items resident at once, small_wgs the workgroups
    bool small_f = false;          // its F-only variant serves fxc_channelize and the 3 .. 8 antenna route (with tiled_f)
    int small_wgs = 0;
    cf* d_tw_small = nullptr;      // [nchan/16][16] wN^(u k1)
    bool tiled_ring = false;       // ntaps <= 4 and nchan <= 4096: VGPR frame ring + window in LDS
    bool prefilter = false;        // ntaps > 4: pfb_prefilter_kernel first, then the tiled kernels with one unit tap
    int pre_tp = 0;                // its register block: 8, 16 or 32 frames
    float* d_hpre = nullptr;       // [pre_tp][nchan] reversed polyphase coefficients
    float* d_ones = nullptr;       // [nchan] unit window of the plain tiled kernel behind the pre-filter
    void* d_pre = nullptr;         // pre-filtered streams of one pass
    size_t pre_bytes = 0;
    bool f8192 = false;            // nchan 8192, ntaps <= 4: fxc_channelize on f8192_ring_kernel (one stream per workgroup, VGPR ring)
    bool x8192 = false;            // nchan 8192, 2 antennas, ntaps <= 4: two passes (f8192_ring_kernel, then its XM form); FXC_X8192=0: off
    bool split8192 = false;        // nchan 8192, 2 antennas: pfb_split8192_kernel + the 4096-channel fused kernel
    cf* d_tw8192 = nullptr;        // [4096] w8192^(4095 - n')
    unsigned long long* d_stamps = nullptr;   // diagnostic builds only
    int fused_grid_max = 0;
    int64_t x_resident = 0;        // workgroups of the X-engine kernel the device holds at once
    bool x_mfma = false;           // more than 8 antennas: xengine_mfma_kernel (k_xmfma.h)
    int64_t fused_seg = 1;         // chunks per round-robin segment of the fused kernel (fx_fused4096.h::RangeWalk)
    cd* d_acc = nullptr;           // [n_base*nchan]
    cd* d_sums = nullptr;          // [n_base*nchan + 1]
    cd* d_cont = nullptr;          // [n_base*nchan + 1] the export a CONTINUUM finalize of the accumulator reduces (lazy)
    bool sums_valid = false;       // d_sums holds exported sums (fxc_reduce): fxc_finalize_sums(plan, NULL, ...) may read it
    // raw rows of the last fx_accumulate pass whose fold into the accumulator is still to be launched: it is launched
    // by whatever needs the accumulator or the workspace next (flush_pending) -- by a finalize together with the
    // export / finalize / reset of every element, in the same kernel
    struct Pending {
        bool valid = false;
        const cf* raw = nullptr;
        cd* part = nullptr;
        int64_t n_rows = 0;
        int layout = 0;
    } pend;
    // finalize results: kResSlots pinned host buffers mapped into the device (the finishing kernel writes the
    // visibilities straight into them: no staging copy), each with the event that marks it complete
    static constexpr int kResSlots = 2;
    cd* h_res[kResSlots] = {nullptr, nullptr};
    cd* d_res[kResSlots] = {nullptr, nullptr};     // the same memory as the device sees it
    hipEvent_t ev_res[kResSlots] = {nullptr, nullptr};
    size_t res_bytes[kResSlots] = {0, 0};
    void* res_dst[kResSlots] = {nullptr, nullptr};    // fxc_finalize_async_to: the caller's buffer for this result
    void* res_user[kResSlots] = {nullptr, nullptr};   // ... and whether the device delivers into it (else via the slot + a copy)
    int64_t res_head = 0, res_tail = 0;            // results queued / collected
    // results too large to be worth a kernel's time on PCIe (28 baselines: 1.8 MB, 35 us inside the finishing kernel):
    // the kernel writes device memory and a copy on a side stream carries it to the slot while the next F+X runs
    cd* d_res_big[kResSlots] = {nullptr, nullptr};
    hipStream_t s_copy = nullptr;
    hipEvent_t ev_fin = nullptr;
    double spectra_count = 0.0;
    // workspace (grown on demand)
    void* d_ws = nullptr;
    int64_t ws_bytes = 0;
    void* d_stage[3] = {nullptr, nullptr, nullptr};   // host-buffer calls: device copies of x and out; uint8 calls on
    size_t stage_bytes[3] = {0, 0, 0};                // plans without the fused ingest: the converted samples
    void* d_rowpart = nullptr;                       // continuum rows of a few-row call: float64 partial sums per bin slice
    size_t rowpart_bytes = 0;
    void* d_dc = nullptr;                            // uint8 ingest: byte sums + conversion offsets per stream
    bool u8_dck = false;                             // this uint8 call: the fused kernel sums its later chunks' bytes itself
    size_t dc_bytes = 0;
    // timing
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;
    bool profiling = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> kev;
    double kernel_ms = 0.0;
    int64_t kernel_launches = 0;
    int stamp_grid = 0;
    StreamTaps taps;               // nchan == 1: the FIR taps by value
    mutable std::string error;
};

struct fxc_pipe {
    fxc_plan* plan = nullptr;
    int64_t chunks = 0;
    int depth = 0, mode = FXC_MODE_SPECTRUM;
    double bandwidth = 1.0;
    size_t in_bytes = 0, out_bytes = 0;
    bool counted = false;      // registered in plan->live_pipes
    int fmt = FXC_IQ_C64;      // sample format of the batches (fxc_iq_format)
    int remove_dc = 0;
    hipStream_t s_in = nullptr, s_out = nullptr;
    std::vector<fxc_pipe_slot> slots;
    int64_t pushed = 0, popped = 0;
};

namespace {


int fail(const fxc_plan* p, int status, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (p)
        p->error = buf;
    else
        g_lib_error = buf;
    return status;
}

// Every ABI entry runs on the plan's device and leaves the caller's current device as it found it (a process
// that drives several GPUs, or torch's own notion of the current device, must not see it change).
struct DeviceGuard {
    int prev = -1;
    bool changed = false, ok = true;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) {
            ok = hipSetDevice(device) == hipSuccess;
            changed = ok && prev >= 0;
        }
    }
    ~DeviceGuard() {
        if (changed) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define FXC_DEVICE(p, device)                                                                  \
    DeviceGuard device_guard__(device);                                                        \
    if (!device_guard__.ok) return fail(p, FXC_ERR_HIP, "hipSetDevice(%d) failed", (int)(device))

#define FXC_HIP(p, call)                                                                                       \
    do {                                                                                                       \
        hipError_t e__ = (call);                                                                               \
        if (e__ != hipSuccess)                                                                                 \
            return fail(p, e__ == hipErrorOutOfMemory ? FXC_ERR_NOMEM : FXC_ERR_HIP, "%s failed: %s", #call,   \
                        hipGetErrorString(e__));                                                               \
    } while (0)

int grid_for(int64_t work_items, int block, int cu_count) {
    int64_t g = (work_items + block - 1) / block;
    const int64_t cap = (int64_t)cu_count * 8;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

// launches the fold of the pending raw rows (h_launch.h); every user of the workspace or the accumulator calls it first
int flush_pending(fxc_plan* p, const FoldFinish* fin = nullptr, hipEvent_t done = nullptr);

int ensure_ws(fxc_plan* p, int64_t bytes) {
    const int rf = flush_pending(p);        // the pending rows (and their partials) live in the workspace
    if (rf) return rf;
    if (bytes <= p->ws_bytes) return FXC_OK;
    if (p->d_ws) {
        FXC_HIP(p, hipStreamSynchronize(p->stream));
        FXC_HIP(p, hipFree(p->d_ws));
        p->d_ws = nullptr;
        p->ws_bytes = 0;
    }
    FXC_HIP(p, hipMalloc(&p->d_ws, (size_t)bytes));
    p->ws_bytes = bytes;
    return FXC_OK;
}

// grow-only device buffer owned by the plan (staging, conversion offsets)
int grow(fxc_plan* p, void** buf, size_t* have, size_t want) {
    if (want <= *have) return FXC_OK;
    FXC_HIP(p, hipStreamSynchronize(p->stream));
    if (*buf) (void)hipFree(*buf);
    *buf = nullptr;
    *have = 0;
    const hipError_t e = hipMalloc(buf, want);
    if (e != hipSuccess) return fail(p, FXC_ERR_NOMEM, "allocation of %zu bytes failed: %s", want, hipGetErrorString(e));
    *have = want;
    return FXC_OK;
}

struct KernelTimer {
    fxc_plan* p;
    hipEvent_t a = nullptr, b = nullptr;
    explicit KernelTimer(fxc_plan* plan) : p(plan) {
        if (p->profiling && hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess)
            (void)hipEventRecord(a, p->stream);
        else
            a = b = nullptr;
    }
    void stop() {
        if (a) {
            (void)hipEventRecord(b, p->stream);
            p->kev.emplace_back(a, b);
            a = b = nullptr;
        }
    }
};

int drain_kernel_events(fxc_plan* p) {
    for (auto& e : p->kev) {
        FXC_HIP(p, hipEventSynchronize(e.second));
        float ms = 0.f;
        FXC_HIP(p, hipEventElapsedTime(&ms, e.first, e.second));
        p->kernel_ms += ms;
        p->kernel_launches += 1;
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    p->kev.clear();
    return FXC_OK;
}
}  // namespace
