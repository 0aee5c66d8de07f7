// Test infrastructure only: the specialised F+X kernel of effex_amd/csrc/fx_spec.h on the host -- the same Body<> the
// device runs, one host thread per GPU thread of a workgroup, a real barrier for its sync() -- so that its index logic
// (thread -> points, ring slots, stage buffers, slots' frame runs, output bins) is checked against the oracle here, without a
// GPU and under ASan / UBSan.  Built per shape by tests/test_emul.py with the same -DFXM_* options hiprtc gets.
#include <pthread.h>

#include <cstring>
#include <thread>
#include <vector>

#include "../../effex_amd/csrc/fx_spec.h"

namespace {

struct HostCtx {
    int tid_;
    long long bid_;
    fxm::cf* lds_;
    pthread_barrier_t* bar_;
    int tid() const { return tid_; }
    long long bid() const { return bid_; }
    fxm::cf* lds() const { return lds_; }
    void sync() const { pthread_barrier_wait(bar_); }      // (the device needs less inside one wave; more is always safe)
};

}  // namespace

extern "C" {

int emul_spec_threads() { return fxm::THREADS; }
int emul_spec_slots() { return fxm::SLOTS; }

// raw[split][chunk][N] for n_chunks chunks, split = workgroup split * SLOTS + slot
int emul_spec_fonly() { return fxm::FONLY ? 1 : 0; }

// F only (FXM_FONLY): n_chunks = the number of streams, `out` = spec[stream / ant][frame][stream % ant][N]
int emul_spec_xm() { return fxm::XM ? 1 : 0; }

int emul_spec_run2(const void* x, const float* h, void* out, const void* tw, const void* dc_u8, long long num_samp, long long n_pts,
                   long long n_chunks, int wg_splits, int ant, long long stride, const void* spec0);

int emul_spec_run(const void* x, const float* h, void* out, const void* tw, const void* dc_u8, long long num_samp, long long n_pts,
                  long long n_chunks, int wg_splits, int ant) {
    return emul_spec_run2(x, h, out, tw, dc_u8, num_samp, n_pts, n_chunks, wg_splits, ant, 0, nullptr);
}

// ... with the streams `stride` samples apart (0: back to back) and, for the XM build, the other antenna's spectra [stream][frame][N]
int emul_spec_run2(const void* x, const float* h, void* out, const void* tw, const void* dc_u8, long long num_samp, long long n_pts,
                   long long n_chunks, int wg_splits, int ant, long long stride, const void* spec0) {
    // the lean build's tables, from fx_spec.h's own definitions of them (the library builds them from the shape: h_rtc.h)
    std::vector<float> h4((size_t)fxm::N * 4, 0.f);
    for (int m = 0; m < fxm::N; ++m)
        for (int t = 0; t < fxm::T; ++t) h4[(size_t)4 * m + t] = h[(size_t)t * fxm::N + m];
    std::vector<fxm::cf> tw1((size_t)(fxm::TW1C > 0 ? fxm::TW1C : 1) * fxm::TPR);
    {
        const fxm::cf* twc = static_cast<const fxm::cf*>(tw);
        int row = 0;
        for (int s = 1; s < fxm::S; ++s) {
            const int nb = fxm::nb_of(s), ns = fxm::ns_of(s);
            for (int j = 0; j < fxm::j_of(s); ++j, ++row)
                for (int lt = 0; lt < fxm::TPR; ++lt) {
                    const int b = fxm::item_bfly_of(s, lt + j * fxm::TPR);
                    if (fxm::LEAN) tw1[(size_t)row * fxm::TPR + lt] = twc[(b % ns) * (nb / ns)];
                }
        }
    }
    const fxm::Args args = {x, h, static_cast<fxm::cf*>(out), static_cast<const fxm::cf*>(tw), static_cast<const fxm::cf*>(dc_u8),
                            num_samp, n_pts, n_chunks, wg_splits, ant, h4.data(), tw1.data(), stride, static_cast<const fxm::cf*>(spec0)};
    const long long groups = fxm::FONLY ? (n_chunks + fxm::NA - 1) / fxm::NA : n_chunks;      // workgroups per split: chunk pairs, or groups of NA streams
    for (long long bid = 0; bid < groups * wg_splits; ++bid) {
        std::vector<fxm::cf> lds((size_t)fxm::SLOTS * fxm::LDS_PER_SLOT + 1);
        std::memset(lds.data(), 0xFF, lds.size() * sizeof(fxm::cf));          // NaNs: nothing may be read before it is written
        pthread_barrier_t bar;
        pthread_barrier_init(&bar, nullptr, fxm::THREADS);
        std::vector<std::thread> threads;
        for (int t = 0; t < fxm::THREADS; ++t)
            threads.emplace_back([&, t] {
                HostCtx cx{t, bid, lds.data(), &bar};
                fxm::Body<HostCtx> body(cx, args);
                body.run();
            });
        for (std::thread& th : threads) th.join();
        pthread_barrier_destroy(&bar);
    }
    return 0;
}
}
