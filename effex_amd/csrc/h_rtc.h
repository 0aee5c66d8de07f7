// Part of libfxcorr's single translation unit: included by fxcorr.hip (not a stand-alone header).
#pragma once

// ------------------------------------------------------------------------------------------
// Kernels specialised at run time (fx_spec.h): for a channel count that is not a power of two the two-antenna F+X kernel is
// compiled for exactly that count when the plan is made -- hiprtc, bound with dlopen like RCCL; the sources are the
// library's own headers, embedded in it at build time (.incbin below), so nothing is read from disk at run time.  One
// compile per (device architecture, shape) and process; a plan whose shape cannot be specialised (or a box without
// hiprtc) keeps the any-shape kernel of k_generic.h -- still a HIP kernel on the device, never a CPU path.
// ------------------------------------------------------------------------------------------

#if !defined(__HIP_DEVICE_COMPILE__)
// the kernel's sources as they stand in this directory when the library is built (NUL-terminated)
__asm__(
    ".pushsection .rodata\n"
    ".hidden fxc_src_fx_spec_h\n.global fxc_src_fx_spec_h\nfxc_src_fx_spec_h:\n.incbin \"fx_spec.h\"\n.byte 0\n"
    ".hidden fxc_src_fx_mixed_h\n.global fxc_src_fx_mixed_h\nfxc_src_fx_mixed_h:\n.incbin \"fx_mixed.h\"\n.byte 0\n"
    ".hidden fxc_src_fx_math_h\n.global fxc_src_fx_math_h\nfxc_src_fx_math_h:\n.incbin \"fx_math.h\"\n.byte 0\n"
    ".popsection\n");
#endif
extern "C" {
extern const char fxc_src_fx_spec_h[];
extern const char fxc_src_fx_mixed_h[];
extern const char fxc_src_fx_math_h[];
}

#include "spec_tuned.h"

namespace {

// timing ablations of fx_spec.h (FXM_ABL: wrong results by design) exist in the developer library only
int spec_ablation() { return FXC_DEV_ENV_INT("FXC_RTC_ABL", 0); }

// hiprtc's C API, the handful of calls used here (its header is not needed: plain C types)
typedef struct _hiprtcProgram* rtc_program;
struct RtcApi {
    void* handle = nullptr;
    int (*create)(rtc_program*, const char*, const char*, int, const char* const*, const char* const*) = nullptr;
    int (*compile)(rtc_program, int, const char* const*) = nullptr;
    int (*log_size)(rtc_program, size_t*) = nullptr;
    int (*log)(rtc_program, char*) = nullptr;
    int (*code_size)(rtc_program, size_t*) = nullptr;
    int (*code)(rtc_program, char*) = nullptr;
    int (*destroy)(rtc_program*) = nullptr;
    const char* (*error_string)(int) = nullptr;
    int (*version)(int*, int*) = nullptr;      // optional
    std::string error;
};

RtcApi* rtc_api() {
    static RtcApi* api = [] {
        RtcApi* a = new RtcApi;
        const char* names[] = {"libhiprtc.so.7", "libhiprtc.so", "/opt/rocm/lib/libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"};
        for (const char* n : names)   // a copy the process already has wins (PyTorch ships one)
            if ((a->handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
        for (size_t k = 0; !a->handle && k < sizeof names / sizeof *names; ++k) a->handle = dlopen(names[k], RTLD_NOW | RTLD_GLOBAL);
        if (!a->handle) {
            const char* why = dlerror();
            a->error = std::string("libhiprtc not found: ") + (why ? why : "dlopen failed");
            return a;
        }
        bool ok = true;
        auto bind = [&](const char* sym) {
            void* f = dlsym(a->handle, sym);
            if (!f) {
                ok = false;
                a->error = std::string("libhiprtc lacks ") + sym;
            }
            return f;
        };
        a->create = reinterpret_cast<decltype(a->create)>(bind("hiprtcCreateProgram"));
        a->compile = reinterpret_cast<decltype(a->compile)>(bind("hiprtcCompileProgram"));
        a->log_size = reinterpret_cast<decltype(a->log_size)>(bind("hiprtcGetProgramLogSize"));
        a->log = reinterpret_cast<decltype(a->log)>(bind("hiprtcGetProgramLog"));
        a->code_size = reinterpret_cast<decltype(a->code_size)>(bind("hiprtcGetCodeSize"));
        a->code = reinterpret_cast<decltype(a->code)>(bind("hiprtcGetCode"));
        a->destroy = reinterpret_cast<decltype(a->destroy)>(bind("hiprtcDestroyProgram"));
        a->error_string = reinterpret_cast<decltype(a->error_string)>(bind("hiprtcGetErrorString"));
        if (ok) a->version = reinterpret_cast<decltype(a->version)>(dlsym(a->handle, "hiprtcVersion"));
        if (!ok) a->handle = nullptr;
        return a;
    }();
    return api;
}

// where a build came from (fxc_info.spec_source)
enum SpecSource { kSpecNone = 0, kSpecBuilt = 1, kSpecCached = 2, kSpecPrebuilt = 3 };

// how fx_spec.h is cut for one channel count: stage order, threads per slot, slots per workgroup
// F + X from complex64 / from the receivers' bytes, the F stage alone, the F stage of one stream multiplied into sums with another's spectra
// (FXM_XM: the second pass of two antennas above 4096 channels)
enum SpecVariant { kSpecC64 = 0, kSpecU8 = 1, kSpecFOnly = 2, kSpecXM = 3 };

constexpr int kSpecLeanAbove = 2048;

// developer knobs of the search below exist in the developer library only (libfxcorr_dev.so): the shipped library's choices do not
// depend on the process environment
#define dev_env_int(name, dflt) FXC_DEV_ENV_INT(name, dflt)
inline int dev_env_list_parse(const char* name, int* out, int cap) {      // "4,25,10" -> out[], the count (0: unset)
    const char* e = name;
    int n = 0;
    while (e && *e && n < cap) {
        out[n++] = std::atoi(e);
        while (*e && *e != ',') ++e;
        if (*e == ',') ++e;
    }
    return n;
}
#define dev_env_list(name, out, cap) dev_env_list_parse(FXC_DEV_ENV(name), out, cap)

inline int spec_lean_above() { return dev_env_int("FXC_RTC_LEAN_ABOVE", kSpecLeanAbove); }

struct SpecShape {
    bool ok = false;
    int n = 0, taps = 0, n_stages = 0, radix[fxc::kMixedMaxStages] = {0}, tpr = 0, slots = 0;
    int u = 1;                   // frames a slot carries through a step together
    int rows = 2;                // streams a workgroup carries: the two antennas / a pair of streams, or (F only, above 4096 channels) one
    bool lean = false;           // FXM_LEAN: taps and first twiddles from L2 tables, nothing but the ring and the sums kept in registers
    // fx_spec.h's work items and LDS layout (spec_layout): rows per item of each stage (0: all the step's rows), the plane stride of the
    // first stage's outputs (0: blocks), the padding behind each block of the buffer a stage writes, the largest radix whose twiddles
    // all stay in registers, the waves per SIMD the register allocation is held to
    int grp[fxc::kMixedMaxStages] = {0}, pad[fxc::kMixedMaxStages] = {0}, plane0 = 0, twfull = 13, waves = 1;
    int threads() const { return tpr * slots; }
    int n_rows() const { return rows * u; }
    int ns_of(int s) const {
        int v = 1;
        for (int i = 0; i < s; ++i) v *= radix[i];
        return v;
    }
    int nb_of(int s) const { return n / radix[s]; }
    int grp_of(int s) const { return (s == 0 || grp[s] <= 0) ? n_rows() : grp[s]; }
    int items_of(int s) const { return nb_of(s) * (n_rows() / grp_of(s)); }
    int j_of(int s) const { return (items_of(s) + tpr - 1) / tpr; }
    int item_bfly(int s, int i) const { return i >= items_of(s) ? 0 : i % nb_of(s); }
    int row_stride() const {     // fx_spec.h's RS
        int m = n;
        for (int s = 0; s + 1 < n_stages; ++s) {
            const int len = (s == 0 && plane0 > 0) ? radix[0] * plane0 : n + (n / ns_of(s + 1)) * pad[s];
            m = std::max(m, len);
        }
        return m;
    }
    size_t lds_bytes() const { return n_stages >= 2 ? (size_t)slots * 2 * n_rows() * row_stride() * sizeof(cf) : 0; }
    std::string list(const int* v) const {
        std::string t;
        for (int i = 0; i < n_stages; ++i) t += (i ? "," : "") + std::to_string(v[i]);
        return t;
    }
};

// a radix fx_spec.h has a butterfly for: 2, 4, the odd primes up to 23, and products of those up to 32 (run in registers as
// Good-Thomas / Cooley-Tukey butterflies)
inline bool spec_radix_ok(int r, bool* big = nullptr) {
    if (r < 2 || r > 32) return false;
    int m = r;
    for (int p : {2, 3, 5, 7, 11, 13, 17, 19, 23})
        while (m % p == 0) {
            m /= p;
            if (p > 13 && big) *big = true;
        }
    return m == 1;
}

// Eligible: two antennas, up to four taps (the frame ring lives in registers), every prime factor has a register butterfly
// (2, 3, 4, 5, 7, 11, 13), a thread's points (first radix x its first-stage butterflies) fit the ring (<= 8), the slots'
// rows fit the LDS.  `first`: the radix of the first stage (0: fx_mixed.h's order -- fours, a two, the odd primes ascending);
// it sets the threads per slot (N / first butterflies), the points a thread keeps in its ring (first of them) and the LDS
// bank pattern of the first stage's stores (an odd stride is conflict-free).  `u`: frames per step.  Above 2048 channels: the
// lean build (fx_spec.h, FXM_LEAN), up to 512 threads a frame with up to two first-stage butterflies each.
SpecShape spec_shape_of(int n, int taps, const int* radix, int n_stages, int u = 1, int rows = 2) {
    SpecShape s;
    if (n < 2 || n > 8192 || taps < 1 || taps > 4 || n_stages < 1 || n_stages > fxc::kMixedMaxStages) return s;
    bool big = false;
    int prod = 1;
    for (int i = 0; i < n_stages; ++i) {
        if (!spec_radix_ok(radix[i], &big)) return s;      // (a register butterfly of 17 ... 23 points: only beside the lean build's few persistent registers)
        s.radix[i] = radix[i];
        prod *= radix[i];
    }
    if (prod != n) return s;
    s.n = n;
    s.taps = taps;
    s.n_stages = n_stages;
    const int nb0 = n / s.radix[0];
    int j0 = 1;
    if (nb0 <= 64) {
        s.tpr = 1;
        while (s.tpr < nb0) s.tpr <<= 1;
    } else {
        // threads a frame at most: 1024 up to 2048 channels (spec_first_radices keeps the shapes of up to 512), 512 above, 256 with a
        // butterfly of 17 ... 23 points (one wave a SIMD, see spec_first_radices).  FXC_RTC_TPR_MAX: developer knob
        const int tmax = big ? 256 : dev_env_int("FXC_RTC_TPR_MAX", n > spec_lean_above() ? 512 : 1024);
        j0 = (nb0 + tmax - 1) / tmax;
        s.tpr = ((nb0 + j0 - 1) / j0 + 63) / 64 * 64;
    }
    s.slots = std::max(1, 256 / s.tpr);
    if (s.radix[0] * j0 > (rows == 1 ? 16 : 8)) return s;      // (the ring: 128 registers at most)
    s.rows = rows;
    s.u = u;
    s.lean = n > spec_lean_above() || big;
    if (big && !dev_env_int("FXC_RTC_BIG_PRIMES", 1)) return s;
    if (s.lean && s.n_stages < 2) return s;
    if (u != 1 && (u != 2 || s.n_stages < 2)) return s;      // (whether two frames' rows cost a resident workgroup: spec_search)
    if (s.lds_bytes() > (size_t)(160 * 1024)) return s;
    s.ok = true;
    return s;
}

SpecShape spec_shape(int n, int taps, int first = 0, int u = 1, int rows = 2) {
    const fxc::MixedPlan mp = fxc::mixed_factor(n);
    if (mp.n_stages < 1) return SpecShape();
    int radix[fxc::kMixedMaxStages];
    for (int i = 0; i < mp.n_stages; ++i) radix[i] = mp.radix[i];
    if (first) {
        int at = -1;
        for (int i = 0; i < mp.n_stages && at < 0; ++i)
            if (radix[i] == first) at = i;
        if (at < 0) return SpecShape();
        for (int i = at; i > 0; --i) radix[i] = radix[i - 1];      // (the others keep their order)
        radix[0] = first;
    }
    return spec_shape_of(n, taps, radix, mp.n_stages, u, rows);
}

// ---- fx_spec.h's work items and LDS layout for a shape (what FXM_GROUPS / FXM_PLANE0 / FXM_PADS say)
//
// LDS banking of the 8-byte accesses (MI355X_MICROARCH.md, LDS): a ds_write_b64 -- and each half of a ds_write2_b64 / ds_read2_b64 -- is
// served in four groups of 16 consecutive lanes, bank = dword address mod 32; a ds_read_b64 in two groups of 32 lanes, bank = dword address
// mod 64; each further distinct dword on a busy bank of a group costs a cycle.  Extra cycles of one wave instruction whose lane l touches element elem[l] (-1: lane idle):
inline int spec_lds_extra(const int* elem, int group, int banks) {
    int extra = 0;
    for (int g0 = 0; g0 < 64; g0 += group) {
        int seen[64][8], cnt[64] = {0};
        int worst = 1;
        for (int l = g0; l < g0 + group; ++l) {
            if (elem[l] < 0) continue;
            for (int d = 0; d < 2; ++d) {
                const int dw = 2 * elem[l] + d, bk = dw % banks;
                bool dup = false;
                for (int k = 0; k < cnt[bk] && k < 8; ++k) dup = dup || seen[bk][k] == dw;
                if (!dup) {
                    if (cnt[bk] < 8) seen[bk][cnt[bk]] = dw;
                    ++cnt[bk];
                    worst = std::max(worst, cnt[bk]);
                }
            }
        }
        extra += worst - 1;
    }
    return extra;
}
// ... of all the stores of stage s and all the loads of stage s + 1 of one step of one row group, with the buffer between them laid out
// as (plane0, pad) say
inline long spec_buffer_conflicts(const SpecShape& sh, int s, int plane0, int pad) {
    const int R = sh.radix[s], ns = sh.ns_of(s), nb = sh.nb_of(s), blk = ns * R;
    long extra = 0;
    int elem[64];
    for (int j = 0; j < sh.j_of(s); ++j)
        for (int w0 = 0; w0 < sh.tpr; w0 += 64)
            for (int q = 0; q < R; ++q) {
                bool any = false;
                for (int l = 0; l < 64; ++l) {
                    const int i = w0 + l + j * sh.tpr;
                    elem[l] = -1;
                    if (w0 + l >= sh.tpr || i >= sh.items_of(s)) continue;
                    const int b = i % nb;
                    elem[l] = s == 0 ? (plane0 > 0 ? q * plane0 + b : b * (R + pad) + q) : (b / ns) * (blk + pad) + b % ns + q * ns;
                    any = true;
                }
                if (any) extra += 4 * spec_lds_extra(elem, 16, 32);
            }
    const int s1 = s + 1, R1 = sh.radix[s1], nb1 = sh.nb_of(s1);
    const int rs = (s == 0 && plane0 > 0) ? nb1 / sh.radix[0] : nb1 + (nb1 / blk) * pad;
    for (int j = 0; j < sh.j_of(s1); ++j)
        for (int w0 = 0; w0 < sh.tpr; w0 += 64)
            for (int r = 0; r < R1; ++r) {
                bool any = false;
                for (int l = 0; l < 64; ++l) {
                    const int i = w0 + l + j * sh.tpr;
                    elem[l] = -1;
                    if (w0 + l >= sh.tpr || i >= sh.items_of(s1)) continue;
                    const int b = i % nb1;
                    elem[l] = ((s == 0 && plane0 > 0) ? (b % sh.radix[0]) * plane0 + b / sh.radix[0] : b + (b / blk) * pad) + r * rs;
                    any = true;
                }
                // (the compiler pairs most of a butterfly's loads into ds_read2_b64 -- served like the stores, 16 lanes over 32 banks -- and leaves
                // about one in four a ds_read_b64: 32 lanes over 64 banks; profiles/r06/experiments.md 3)
                if (any) extra += 3 * spec_lds_extra(elem, 16, 32) + spec_lds_extra(elem, 32, 64);
            }
    return extra;
}

// Vector instructions of one R-point butterfly with its R - 1 twiddle multiplies (fx_spec.h::stage_bfly; packed instructions), roughly
inline int spec_bfly_cost(int r) {
    switch (r) {
        case 2: return 2 + 2;
        case 3: return 8 + 4;
        case 4: return 8 + 6;
        case 5: return 18 + 8;
        case 7: return 34 + 12;
        default: break;
    }
    for (int a : {4, 2, 3, 5, 7, 11, 13})
        if (r % a == 0 && r > a) {
            const int b = r / a;
            int g = a, h = b;
            while (h) {
                const int t = g % h;
                g = h;
                h = t;
            }
            // b butterflies of a points, a of b points (their own twiddle terms taken off again), the literals between them unless coprime
            return b * (spec_bfly_cost(a) - 2 * (a - 1)) + a * (spec_bfly_cost(b) - 2 * (b - 1)) + (g == 1 ? 0 : 2 * (a - 1) * (b - 1)) + 2 * (r - 1);
        }
    return (r - 1) * (r - 1) / 2 + 4 * r;      // odd primes 11 ... 23
}

// Rows per item of every stage after the first: the divisor of the step's rows (a multiple of the antennas in the F + X build's last
// stage) that costs a SIMD the fewest issue slots -- whole rounds of four waves x the item's butterflies -- and, among equals, the most rows
// (fewer twiddle and offset registers).  Then where the stages' outputs stand (spec_buffer_conflicts), kept within the LDS the
// unpadded rows already allowed a CU's resident workgroups.
inline void spec_layout(SpecShape& sh, bool fonly, bool groups_only = false) {
    const int rows = sh.n_rows();
    int forced[fxc::kMixedMaxStages];
    const int nf = dev_env_list("FXC_RTC_GROUPS", forced, fxc::kMixedMaxStages);
    for (int s = 1; s < sh.n_stages; ++s) {
        const bool last = s == sh.n_stages - 1;
        long best = -1;
        for (int g = rows; g >= 1; --g) {
            if (rows % g) continue;
            if (last && !fonly && g % sh.rows) continue;
            const int items = sh.nb_of(s) * (rows / g);
            long slots = 0;                                   // issue slots of the busiest SIMD: its waves' items, one after the other
            for (int i0 = 0; i0 < items; i0 += sh.tpr) {
                const int waves = (std::min(items - i0, sh.tpr) + 63) / 64;
                slots += (waves + 3) / 4;
            }
            const long cost = slots * ((long)g * spec_bfly_cost(sh.radix[s]) + (sh.radix[s] > sh.twfull ? 2 * (sh.radix[s] - 2) : 0));
            if (best < 0 || cost < best) {
                best = cost;
                sh.grp[s] = g;
            }
        }
        if (s < nf && forced[s] > 0 && rows % forced[s] == 0) sh.grp[s] = forced[s];
    }
    if (sh.n_stages < 2 || groups_only) return;      // (groups_only: ranking thousands of stage lists needs the items, not the bank patterns)
    const size_t base_lds = sh.lds_bytes();
    auto fits = [&](const SpecShape& t) {
        const size_t b = t.lds_bytes();
        return b <= (size_t)(160 * 1024) && (size_t)(160 * 1024) / b >= std::min<size_t>((size_t)(160 * 1024) / base_lds, 2);
    };
    int fpads[fxc::kMixedMaxStages];
    const int npf = dev_env_list("FXC_RTC_PADS", fpads, fxc::kMixedMaxStages);
    const int fplane = dev_env_int("FXC_RTC_PLANE0", -1);
    if (!dev_env_int("FXC_RTC_LAYOUT", 1)) return;
    for (int s = 0; s + 1 < sh.n_stages; ++s) {
        long best = spec_buffer_conflicts(sh, s, 0, 0);
        int best_plane = 0, best_pad = 0;
        for (int pad = 1; pad < 16 && best > 0; ++pad) {
            SpecShape t = sh;
            t.pad[s] = pad;
            if (!fits(t)) break;
            const long c = spec_buffer_conflicts(sh, s, 0, pad);
            if (c < best) best = c, best_pad = pad;
        }
        if (s == 0 && sh.radix[0] % 2 == 0)
            for (int pl = sh.nb_of(0); pl < sh.nb_of(0) + 64 && best > 0; ++pl) {
                SpecShape t = sh;
                t.plane0 = pl;
                if (!fits(t)) break;
                const long c = spec_buffer_conflicts(sh, 0, pl, 0);
                if (c < best) best = c, best_plane = pl, best_pad = 0;
            }
        if (env_int("FXC_RTC_VERBOSE", 0) > 2)
            std::fprintf(stderr, "libfxcorr: layout of the buffer behind stage %d of %s: %ld extra LDS cycles plain, %ld with pad %d / plane %d\n", s,
                         sh.list(sh.radix).c_str(), spec_buffer_conflicts(sh, s, 0, 0), best, best_pad, best_plane);
        if (s == 0) sh.plane0 = fplane >= 0 ? fplane : best_plane;
        sh.pad[s] = s < npf ? fpads[s] : (s == 0 && sh.plane0 > 0 ? 0 : best_pad);
    }
}

// The first-stage radices worth building for n channels, best guess first.  Measured (tools/sweep_spec_r0.sh, 14 channel counts,
// profiles/r05/experiments.md): workgroups of whole multiples of 256 threads -- a wave on every SIMD -- win (720 channels: 3 first,
// 240 butterflies on 256 threads, 1.72 ms; 4 first, 180 on 192 threads, 2.11 ms), 2 first loses wherever there is a choice
// (N / 2 threads per frame leave the later stages a few butterflies each: 600 channels 3.4 ms against 1.75), then the fuller
// first stage, then the smaller radix (fewer ring registers).
// streams per workgroup of the build for (n, variant): F only above 4096 channels carries one (sixteen points a thread)
inline int spec_rows(int n, int variant) {
    return variant == kSpecXM || (variant == kSpecFOnly && n > dev_env_int("FXC_RTC_ROWS1_ABOVE", 4096)) ? 1 : 2;      // (developer knob)
}

std::vector<int> spec_first_radices(int n, int taps, int rows = 2) {
    struct Cand {
        int r, unbalanced, is_two, waste;
    };
    std::vector<Cand> c;
    const SpecShape base = spec_shape(n, taps, 0, 1, rows);
    if (!base.n) return {};
    for (int r : {3, 4, 5, 7, 2, 11, 13, 17, 19, 23}) {
        const SpecShape s = spec_shape(n, taps, r, 1, rows);
        if (!s.ok || s.threads() > 512) continue;      // (more than 512 threads leave under 256 registers a thread: the ring does not fit)
        bool big = false;
        for (int i = 0; i < s.n_stages; ++i) big = big || s.radix[i] > 13;
        // (a butterfly of 17 ... 23 points takes 220 - 400 registers with the ring: one wave a SIMD, so 256 threads at most -- 1700 / 1900 /
        // 2040 channels on 512 threads spilled 150 - 240 B after 7 s of compiling)
        if (big && s.threads() > 256) continue;
        const int nb0 = n / r, j0 = (nb0 + s.tpr - 1) / s.tpr;
        const int waste = 20 - 20 * nb0 / (s.tpr * j0);                 // idle lanes of the first stage, in twentieths
        c.push_back({r, s.threads() % 256 != 0, r == 2, waste});
    }
    std::stable_sort(c.begin(), c.end(), [](const Cand& a, const Cand& b) {
        if (a.unbalanced != b.unbalanced) return a.unbalanced < b.unbalanced;
        if (a.is_two != b.is_two) return a.is_two < b.is_two;
        if (a.waste != b.waste) return a.waste < b.waste;
        return a.r < b.r;
    });
    std::vector<int> out;
    for (const Cand& k : c) out.push_back(k.r);
    return out;
}

struct SpecKernel {
    hipModule_t module = nullptr;
    hipFunction_t fn = nullptr;
    SpecShape shape;
    int wgs_per_cu = 1;          // resident workgroups per CU (occupancy query)
    int vgprs = 0;
    int source = kSpecNone;      // where the code object came from (SpecSource), and the seconds the whole search for it took
    double seconds = 0;
    cf* d_tw1 = nullptr;         // lean builds: the first twiddles by stage, butterfly and thread (fx_spec.h, Args::tw1)
    std::string error;           // why there is none (fn == nullptr)
};

// the argument block of fxm_fx2_kernel (fx_spec.h::Args), field for field
struct SpecArgs {
    const void* x;
    const float* h;
    cf* out;
    const cf* tw;
    const cf* dc_u8;
    long long num_samp, n_pts, n_chunks;
    int wg_splits;
    int ant;
    const float* h4;
    const cf* tw1;
    long long stride;
    const cf* spec0;
};

// Args::tw1 of a lean build: row tw_base(s) + j (stages 1 .. S-1, the thread's j-th butterfly) holds, at thread lt, the twiddle
// exp(+2 pi i (b mod ns) (nb / ns) / N) of butterfly b = lt + j tpr (b = 0 where the thread has none) -- the values of the
// plan's own table (fxcorr.hip: float32 of the float64 cosine and sine)
std::vector<cf> spec_tw1_table(const SpecShape& sh) {
    std::vector<cf> t;
    int ns = sh.radix[0];
    for (int s = 1; s < sh.n_stages; ++s) {
        const int nb = sh.nb_of(s), jn = sh.j_of(s), tmul = nb / ns;
        for (int j = 0; j < jn; ++j)
            for (int lt = 0; lt < sh.tpr; ++lt) {
                const int b = sh.item_bfly(s, lt + j * sh.tpr);
                const double ph = 6.283185307179586476925286766559 * (double)((b % ns) * tmul) / (double)sh.n;
                t.push_back(fxc::mk((float)std::cos(ph), (float)std::sin(ph)));
            }
        ns *= sh.radix[s];
    }
    return t;
}

std::mutex g_spec_mutex;
std::map<std::string, SpecKernel*> g_spec_cache;     // (device, shape) -> kernel; entries live as long as the process

// An integer of the kernel's metadata note in a code object (msgpack: the key as a string, then the value): -1 if absent.
// (hipFuncGetAttribute's LOCAL_SIZE_BYTES does not report the scratch of a module function here: it read 80 for a kernel
// whose .private_segment_fixed_size is 0.)
long long code_object_int(const std::vector<char>& image, const char* key) {
    const size_t klen = std::strlen(key);
    for (size_t i = 0; i + klen + 1 < image.size(); ++i) {
        if (std::memcmp(image.data() + i, key, klen) != 0) continue;
        const unsigned char* v = reinterpret_cast<const unsigned char*>(image.data()) + i + klen;
        const size_t left = image.size() - i - klen;
        if (v[0] <= 0x7f) return v[0];
        if (v[0] == 0xcc && left >= 2) return v[1];
        if (v[0] == 0xcd && left >= 3) return ((long long)v[1] << 8) | v[2];
        if (v[0] == 0xce && left >= 5) return ((long long)v[1] << 24) | ((long long)v[2] << 16) | ((long long)v[3] << 8) | v[4];
    }
    return -1;
}

// the builds of fx_spec.h: F+X from complex64 samples, F+X from the receivers' bytes, the F stage alone

// Code objects on disk: a build is keyed by everything that goes into it -- the three sources as embedded, the options, the
// architecture, the compiler's version -- and kept under $FXC_RTC_CACHE (default $XDG_CACHE_HOME/fxcorr or ~/.cache/fxcorr; "0"
// turns it off), so that only the first process on a machine pays the second or so a shape costs.  Written to a temporary name
// and renamed: concurrent ranks may race for the same key and all end up with a whole file.
std::string spec_cache_dir() {
    const char* e = std::getenv("FXC_RTC_CACHE");
    if (e) return (e[0] == 0 || std::strcmp(e, "0") == 0) ? std::string() : std::string(e);
    const char* x = std::getenv("XDG_CACHE_HOME");
    if (x && x[0]) return std::string(x) + "/fxcorr";
    const char* h = std::getenv("HOME");
    return (h && h[0]) ? std::string(h) + "/.cache/fxcorr" : std::string();
}

// Code objects that ship beside the library: <directory of libfxcorr>/rtc_prebuilt/<key>.co, made at build time (effex_amd/build.py) for a
// stated list of channel counts.  Their key is the kernel's sources, the options and the architecture -- NOT the compiler, so that a
// process with another hiprtc loaded (PyTorch ships clang 20, ROCm 7.2 clang 22) finds them too; they are looked up before hiprtc is
// even bound.  Read-only for the shipped library; the developer library fills FXC_RTC_PREBUILD_DIR when that is set.
std::string spec_prebuilt_dir() {
    Dl_info where;
    if (!dladdr(reinterpret_cast<const void*>(fxc_src_fx_spec_h), &where) || !where.dli_fname) return std::string();
    std::string path = where.dli_fname;
    const size_t slash = path.rfind('/');
    return (slash == std::string::npos ? std::string(".") : path.substr(0, slash)) + "/rtc_prebuilt";
}

std::string spec_cache_key(const std::vector<std::string>& opts, RtcApi* api) {
    unsigned long long a = 1469598103934665603ull, b = 0x9E3779B97F4A7C15ull;      // two FNV-1a style sums over the same bytes
    auto eat = [&](const char* p, size_t n) {
        for (size_t i = 0; i < n; ++i) {
            a = (a ^ (unsigned char)p[i]) * 1099511628211ull;
            b = (b + (unsigned char)p[i] + 1) * 0xD6E8FEB86659FD93ull;
            b ^= b >> 29;
        }
    };
    for (const char* src : {fxc_src_fx_spec_h, fxc_src_fx_mixed_h, fxc_src_fx_math_h}) eat(src, std::strlen(src) + 1);
    for (const std::string& o : opts) eat(o.c_str(), o.size() + 1);
    if (api) {                               // (nullptr: the key of a pre-built code object -- any compiler's)
        int major = 0, minor = 0;
        if (api->version) (void)api->version(&major, &minor);
        const std::string v = "hiprtc " + std::to_string(major) + "." + std::to_string(minor);
        eat(v.c_str(), v.size());
        Dl_info where;                       // (two hiprtc builds of one version number: PyTorch's and ROCm's differ in path)
        if (api->compile && dladdr(reinterpret_cast<void*>(api->compile), &where) && where.dli_fname) eat(where.dli_fname, std::strlen(where.dli_fname));
    }
    char hex[40];
    std::snprintf(hex, sizeof hex, "%016llx%016llx", a, b);
    return hex;
}

bool spec_cache_load(const std::string& path, std::vector<char>& image) {
    FILE* fh = std::fopen(path.c_str(), "rb");
    if (!fh) return false;
    std::fseek(fh, 0, SEEK_END);
    const long n = std::ftell(fh);
    std::fseek(fh, 0, SEEK_SET);
    bool ok = n > 64;
    if (ok) {
        image.resize((size_t)n);
        ok = std::fread(image.data(), 1, (size_t)n, fh) == (size_t)n && std::memcmp(image.data(), "\177ELF", 4) == 0;
    }
    std::fclose(fh);
    if (!ok) image.clear();
    return ok;
}

void spec_cache_store(const std::string& dir, const std::string& path, const std::vector<char>& image) {
    std::string made;
    for (size_t i = 1; i <= dir.size(); ++i)        // mkdir -p
        if (i == dir.size() || dir[i] == '/') {
            made = dir.substr(0, i);
            (void)mkdir(made.c_str(), 0755);
        }
    const std::string tmp = path + ".tmp" + std::to_string((long long)getpid());
    FILE* fh = std::fopen(tmp.c_str(), "wb");
    if (!fh) return;
    const bool ok = std::fwrite(image.data(), 1, image.size(), fh) == image.size();
    std::fclose(fh);
    if (!ok || std::rename(tmp.c_str(), path.c_str()) != 0) (void)std::remove(tmp.c_str());
}

// The target the kernels are built for: the device's architecture without its feature suffix ("gfx950:sramecc+:xnack-" -> "gfx950"),
// like the library itself (--offload-arch=gfx950) -- so a code object built where there is no device (fxc_spec_probe, the CPU test
// suite, a build step) is the one a plan on the device looks up in the cache
std::string spec_arch(const char* gcn_arch_name) {
    std::string a = gcn_arch_name ? gcn_arch_name : "";
    const size_t colon = a.find(':');
    return colon == std::string::npos ? a : a.substr(0, colon);
}

// fx_spec.h for one shape -> a code object for `arch` (e.g. "gfx950:sramecc+:xnack-"); needs no device
bool spec_compile(const SpecShape& shape, int variant, const char* arch, std::vector<char>& image, std::string& error, int* source = nullptr,
                  double* seconds = nullptr) {
    const auto t0 = std::chrono::steady_clock::now();
    auto done = [&](int src) {
        if (source) *source = src;
        if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return true;
    };
    std::string radices;
    for (int i = 0; i < shape.n_stages; ++i) radices += (i ? "," : "") + std::to_string(shape.radix[i]);
    std::vector<std::string> opts = {std::string("--offload-arch=") + arch, "-O3", "-std=c++17", "-fno-slp-vectorize",
                                     "-DFXM_N=" + std::to_string(shape.n), "-DFXM_T=" + std::to_string(shape.taps),
                                     "-DFXM_TPR=" + std::to_string(shape.tpr), "-DFXM_SLOTS=" + std::to_string(shape.slots),
                                     "-DFXM_NST=" + std::to_string(shape.n_stages), "-DFXM_RADICES=" + radices,
                                     "-DFXM_U8=" + std::to_string((int)(variant == kSpecU8)),
                                     "-DFXM_FONLY=" + std::to_string((int)(variant == kSpecFOnly || variant == kSpecXM)), "-DFXM_XM=" + std::to_string((int)(variant == kSpecXM)),
                                     "-DFXM_U=" + std::to_string(shape.u), "-DFXM_LEAN=" + std::to_string((int)shape.lean), "-DFXM_ROWS=" + std::to_string(shape.rows),
                                     "-DFXM_GROUPS=" + shape.list(shape.grp), "-DFXM_PADS=" + shape.list(shape.pad), "-DFXM_PLANE0=" + std::to_string(shape.plane0),
                                     "-DFXM_TWFULL=" + std::to_string(shape.twfull), "-DFXM_WAVES=" + std::to_string(shape.waves),
                                     "-DFXM_LEAN_TW_EARLY=" + std::to_string(dev_env_int("FXC_RTC_TW_EARLY", 1)),
                                     "-DFXM_OOB_ZERO=" + std::to_string(dev_env_int("FXC_RTC_OOB_ZERO", 1)),
                                     "-DFXM_ABL=" + std::to_string(spec_ablation())};      // (timing ablations: wrong results, developer library only)
    const std::string pre_name = spec_cache_key(opts, nullptr) + ".co", pre_dir = spec_prebuilt_dir();
    if (!pre_dir.empty() && !spec_ablation() && spec_cache_load(pre_dir + "/" + pre_name, image)) return done(kSpecPrebuilt);
    RtcApi* api = rtc_api();
    if (!api->handle) {
        error = api->error;
        return false;
    }
    const std::string dir = spec_cache_dir();
    const std::string cached = dir.empty() ? std::string() : dir + "/" + spec_cache_key(opts, api) + ".co";
    if (!cached.empty() && spec_cache_load(cached, image)) {
        if (const char* fill = FXC_DEV_ENV("FXC_RTC_PREBUILD_DIR")) spec_cache_store(fill, std::string(fill) + "/" + pre_name, image);
        return done(kSpecCached);
    }
    std::vector<const char*> optv;
    for (const std::string& o : opts) optv.push_back(o.c_str());
    const char* headers[] = {fxc_src_fx_spec_h, fxc_src_fx_mixed_h, fxc_src_fx_math_h};
    const char* names[] = {"fx_spec.h", "fx_mixed.h", "fx_math.h"};
    rtc_program prog = nullptr;
    int r = api->create(&prog, "#include \"fx_spec.h\"\n", "fxm_fx2.hip", 3, headers, names);
    if (r != 0) {
        error = std::string("hiprtcCreateProgram: ") + api->error_string(r);
        return false;
    }
    r = api->compile(prog, (int)optv.size(), optv.data());
    if (r != 0) {
        size_t n = 0;
        std::string log;
        if (api->log_size(prog, &n) == 0 && n > 1) {
            log.resize(n);
            api->log(prog, &log[0]);
        }
        error = std::string("hiprtcCompileProgram: ") + api->error_string(r) + ": " + log.substr(0, 1500);
        api->destroy(&prog);
        return false;
    }
    size_t bytes = 0;
    api->code_size(prog, &bytes);
    image.resize(bytes);
    api->code(prog, image.data());
    api->destroy(&prog);
    if (!cached.empty()) spec_cache_store(dir, cached, image);
    if (const char* fill = FXC_DEV_ENV("FXC_RTC_PREBUILD_DIR")) spec_cache_store(fill, std::string(fill) + "/" + pre_name, image);
    return done(kSpecBuilt);
}

// workgroups of this build a CU holds: registers (512 per lane and SIMD, granule 8), LDS (160 KiB), 32 waves
int spec_resident(const SpecShape& sh, long long vgprs) {
    const int alloc = (int)((vgprs + 7) / 8 * 8);
    const int waves_simd = std::min(8, 512 / std::max(alloc, 8));
    const int wg_waves_simd = (sh.threads() / 64 + 3) / 4;                    // waves of one workgroup on its fullest SIMD
    int wgs = waves_simd / std::max(wg_waves_simd, 1);
    if (sh.lds_bytes() > 0) wgs = std::min<long long>(wgs, (long long)(160 * 1024) / (long long)sh.lds_bytes());
    return std::min(wgs, 32 * 64 / std::max(sh.threads(), 64));
}

// one build: compile `shape`, read its registers / scratch from the code object.  image empty: the compile failed (error says why)
struct SpecBuild {
    SpecShape shape;
    std::vector<char> image;
    long long vgprs = 0, scratch = 0;
    int resident = 0;
    int source = kSpecNone;      // SpecSource of this build's code object
    double seconds = 0;          // what getting it took (compiling, or reading the file)
    std::string error;
};
SpecBuild spec_build(const SpecShape& shape, int variant, const char* arch) {
    SpecBuild b;
    b.shape = shape;
    if (!spec_compile(shape, variant, arch, b.image, b.error, &b.source, &b.seconds)) {
        b.image.clear();
        return b;
    }
    b.vgprs = code_object_int(b.image, ".vgpr_count");
    b.scratch = code_object_int(b.image, ".private_segment_fixed_size");
    b.resident = b.scratch == 0 ? spec_resident(shape, b.vgprs) : 0;
    return b;
}

inline int spec_regs_est(SpecShape sh, bool fonly);

// one candidate: its work items and layout chosen (spec_layout), built for two waves per SIMD first where the workgroup is small
// enough for two of them on a CU (a build that spills there is built again for one)
SpecBuild spec_build_laid_out(SpecShape sh, int variant, const char* arch) {
    spec_layout(sh, variant == kSpecFOnly);
    if (dev_env_int("FXC_RTC_TWFULL", 0) > 0) sh.twfull = dev_env_int("FXC_RTC_TWFULL", 0);
    const int force_waves = dev_env_int("FXC_RTC_WAVES", 0);
    if (force_waves > 0) {
        sh.waves = force_waves;
        return spec_build(sh, variant, arch);
    }
    if (sh.threads() <= 256 && sh.n_stages >= 2) {
        sh.waves = 2;
        SpecBuild b = spec_build(sh, variant, arch);
        if (!b.image.empty() && b.scratch == 0) return b;
    }
    sh.waves = 1;
    SpecBuild b = spec_build(sh, variant, arch);
    if (env_int("FXC_RTC_VERBOSE", 0) > 1)
        std::fprintf(stderr, "libfxcorr: built stages=%s u=%d lean=%d: vgprs=%lld (estimated %d) scratch=%lld resident=%d %s\n", sh.list(sh.radix).c_str(), sh.u,
                     (int)sh.lean, b.vgprs, spec_regs_est(sh, variant == kSpecFOnly), b.scratch, b.resident, b.error.c_str());
    return b;
}

// Vector registers a build of this shape needs, roughly: what stays from step to step (the ring, the taps, the twiddles and offsets of the
// thread's items, the sums) plus the busiest stage's butterfly in flight.  Checked against the compiler on the shapes of
// profiles/r06/sweep_lists.md (within 10 %); a candidate estimated beyond the 256 a thread of two resident waves per SIMD has is not
// built (the compiler would spill it, or halve the resident workgroups: [4,10,25] at 1000 channels: 348 registers, 2.33 ms against 1.52).
inline int spec_regs_est(SpecShape sh, bool fonly) {
    spec_layout(sh, fonly, true);
    const int S = sh.n_stages, r0 = sh.radix[0];
    const int j0 = (sh.nb_of(0) + sh.tpr - 1) / sh.tpr, pts = r0 * j0, ns_ring = sh.taps + sh.u - 1;
    int keep = pts * sh.rows * ns_ring * 2 + (sh.lean ? 0 : sh.taps * pts) + 24;      // (+ addresses, loop state)
    int busy = sh.u * sh.rows * (sh.lean ? r0 : pts) * 2 + 2 * r0 + 8;                  // the FIR's sums, one first butterfly
    for (int s = 1; s < S; ++s) {
        const int R = sh.radix[s], j = sh.j_of(s);
        const bool whole = !sh.lean && R <= sh.twfull;
        keep += j * ((whole ? 2 * (R - 1) : (sh.lean ? 0 : 2)) + (sh.lean ? 0 : 2));
        int a = 0;
        for (int c : {4, 2, 3, 5, 7})
            if (!a && R % c == 0 && R > c && R != 4) a = c;
        const int sub = a ? std::max(a, R / a) : R;
        int t = 2 * R + 4 * sub + (whole ? 0 : 2 * (a ? a + R / a : R)) + 8;
        if (s == S - 1 && !fonly) {
            keep += j * 2 * R;                                                         // the sums
            t += 2 * R;                                                                // the other antenna's outputs
        }
        busy = std::max(busy, t);
    }
    if (sh.lean && sh.u == 2) keep += 24;       // (measured: the lean build's two-frame steps need more than their rows' share)
    if (sh.lean && r0 >= 8) keep += 32;         // (eight points through one butterfly beside the 128 registers of ring: 4000 channels spilled 52 - 100 B)
    return keep + busy;
}

// What one step of a shape costs a CU, in cycles per frame, as far as a static count can tell: the vector instructions of the busiest
// SIMD (a packed instruction about six cycles: tools/ubench/valu_rate.hip), the LDS's cycles for every wave's loads (two each) and
// stores (six each: MI355X_MICROARCH.md), and a stall per trip through LDS (the barrier and the first load's latency).  It ranks the
// candidates; the builds' registers decide (spec_search).
inline double spec_cost(SpecShape sh, bool fonly) {
    spec_layout(sh, fonly, true);
    const int rows = sh.n_rows(), S = sh.n_stages;
    auto rounds_of = [&](int items) {
        long r = 0;
        for (int i0 = 0; i0 < items; i0 += sh.tpr) r += ((std::min(items - i0, sh.tpr) + 63) / 64 + 3) / 4;
        return r;
    };
    auto waves_of = [&](int items) { return (long)(items + 63) / 64; };
    double valu = 0, lds = 0;
    {   // the first stage: the FIR of a thread's points, their butterflies, the stores
        const int nb0 = sh.nb_of(0), r0 = sh.radix[0];
        const long rounds = rounds_of(nb0), waves = waves_of(nb0);
        valu += (double)rounds * rows * (r0 * sh.taps + spec_bfly_cost(r0) - 2 * (r0 - 1));
        if (S >= 2) lds += (double)waves * rows * r0 * 6;
    }
    for (int s = 1; s < S; ++s) {
        const int R = sh.radix[s], G = sh.grp_of(s), items = sh.items_of(s);
        const long rounds = rounds_of(items), waves = waves_of(items);
        valu += (double)rounds * (G * spec_bfly_cost(R) + ((sh.lean || R > sh.twfull) ? 2 * (R - 2) : 0));
        lds += (double)waves * G * R * 2;
        if (s < S - 1) lds += (double)waves * G * R * 6;
        else if (!fonly) valu += (double)rounds * (G / sh.rows) * 2 * R;
    }
    double cost = (6.0 * valu + lds + 700.0 * (S - 1)) / sh.u;
    if (sh.threads() % 256 != 0 && sh.threads() > 64) cost *= 1.2;      // (a SIMD without a wave of its own: 720 channels on 192 threads 2.11 ms, on 256: 1.72)
    return cost;
}

// The stage lists worth building for n channels, best guess first.  Candidates: every ordered way to write n as a product of radices
// fx_spec.h has butterflies for -- primes, 4, and the composites that run in registers (6, 8, 9, 10, 16, 20, 25, 32) -- with at most
// as many stages as the prime-factor order, a last radix of at most 10 (16 for the F stage alone: the last butterfly's outputs of
// both antennas and the sums live in registers together) and at most one radix beyond 16; ranked by spec_cost.  The prime-factor order
// of round 5 (spec_first_radices) follows as the fallback.  FXC_RTC_RADICES (developer knob): exactly this list; FXC_RTC_PICK: only the
// k-th ranked candidate.
inline void spec_enum_lists(int n_left, int max_stages, bool fonly, std::vector<int>& cur, std::vector<std::vector<int>>& out) {
    if (out.size() >= 20000) return;
    if (n_left == 1) {
        if (!cur.empty() && cur.back() <= (fonly ? 16 : 10) ) out.push_back(cur);
        return;
    }
    if ((int)cur.size() >= max_stages) return;
    int n_big = 0;
    for (int r : cur) n_big += r > 16;
    for (int r : {2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 16, 17, 19, 20, 23, 25, 32}) {
        if (n_left % r || (r > 16 && n_big)) continue;
        if (cur.empty() && (r > 8 || r == 2) && r != 11 && r != 13 && n_left != 2 * 2 && n_left > 2) continue;      // (the first radix: the ring holds eight points a thread; 2 first loses wherever there is a choice)
        cur.push_back(r);
        spec_enum_lists(n_left / r, max_stages, fonly, cur, out);
        cur.pop_back();
    }
}

// the measured choice for (n, variant) (spec_tuned.h; four taps), or nullptr
inline const SpecTuned* spec_tuned_entry(int n, int taps, int variant) {
    if (taps != 4 || !dev_env_int("FXC_RTC_TUNED", 1)) return nullptr;
    const SpecTuned* table = variant == kSpecFOnly ? kSpecTunedF : (variant == kSpecXM ? kSpecTunedXM : kSpecTuned);
    for (const SpecTuned* t = table; t->n > 0; ++t)
        if (t->n == n) return t;
    return nullptr;
}

std::vector<std::vector<int>> spec_stage_lists(int n, int taps, int rows, int variant, int* n_chosen = nullptr) {
    const bool fonly = variant == kSpecFOnly;
    std::vector<std::vector<int>> out;
    if (n_chosen) *n_chosen = 0;      // how many leading entries are measured / ranked choices (the prime-factor orders follow)
    int forced[fxc::kMixedMaxStages];
    const int nf = dev_env_list("FXC_RTC_RADICES", forced, fxc::kMixedMaxStages);
    if (nf > 0) {
        out.emplace_back(forced, forced + nf);
        return out;
    }
    std::vector<int> firsts = spec_first_radices(n, taps, rows);
    if (const int want = dev_env_int("FXC_RTC_R0", 0)) firsts.assign(1, want);
    std::vector<std::vector<int>> legacy;
    for (int r : firsts) {
        const SpecShape s = spec_shape(n, taps, r, 1, rows);
        if (s.ok) legacy.emplace_back(s.radix, s.radix + s.n_stages);
    }
    if (legacy.empty()) return out;
    if (dev_env_int("FXC_RTC_COMPOSITE", 1) && dev_env_int("FXC_RTC_PICK", -1) < 0)
        if (const SpecTuned* t = spec_tuned_entry(n, taps, variant)) {      // a measured choice for this channel count (spec_tuned.h): first
            const SpecShape probe = spec_shape_of(n, taps, t->radix, t->n_stages, 1, rows);
            if (probe.ok) out.emplace_back(t->radix, t->radix + t->n_stages);
        }
    if (dev_env_int("FXC_RTC_COMPOSITE", 1) && out.empty()) {      // (a measured choice needs no ranking behind it: the prime-factor orders are its fallback)
        std::vector<std::vector<int>> all;
        std::vector<int> cur;
        spec_enum_lists(n, (int)legacy[0].size(), fonly, cur, all);
        std::vector<std::pair<double, int>> ranked;
        for (size_t i = 0; i < all.size(); ++i) {
            const SpecShape one = spec_shape_of(n, taps, all[i].data(), (int)all[i].size(), 1, rows);
            if (!one.ok || one.threads() > 512) continue;
            bool big = false;
            for (int r : all[i]) (void)spec_radix_ok(r, &big);
            if (big && one.threads() > 256) continue;
            const SpecShape two = spec_shape_of(n, taps, all[i].data(), (int)all[i].size(), 2, rows);
            const int budget = one.threads() <= 256 ? 256 : 512 * 256 / one.threads();
            if (spec_regs_est(one, fonly) > budget) continue;
            const double c = std::min(spec_cost(one, fonly), two.ok && spec_regs_est(two, fonly) <= budget ? spec_cost(two, fonly) : 1e300);
            ranked.emplace_back(c, (int)i);
        }
        std::sort(ranked.begin(), ranked.end());
        const int pick = dev_env_int("FXC_RTC_PICK", -1);
        if (pick >= 0) {
            if (pick < (int)ranked.size()) out.push_back(all[ranked[pick].second]);
            return out;
        }
        for (size_t k = 0; k < ranked.size() && k < 2; ++k)
            if (std::find(out.begin(), out.end(), all[ranked[k].second]) == out.end()) out.push_back(all[ranked[k].second]);
    }
    if (n_chosen) *n_chosen = (int)out.size();
    for (const std::vector<int>& l : legacy)
        if (std::find(out.begin(), out.end(), l) == out.end()) out.push_back(l);
    return out;
}

// The build for (n, taps): the first candidate order that keeps two workgroups on a CU (else the best seen), with two frames per
// step when that costs no resident workgroup (4 - 10 % where it fits: 1000 channels 206 -> 256 registers, 1.87 -> 1.73 ms; 96
// channels with 3 first 150 -> 192 registers, three workgroups -> two, 1.25 -> 1.33 ms: one frame there).  Developer knobs:
// FXC_RTC_R0 / FXC_RTC_U force the first radix / the frames per step.
SpecBuild spec_search(int n, int taps, int variant, const char* arch) {
    SpecBuild best;
    best.error = "no specialised kernel for this channel count";
    const int rows = spec_rows(n, variant);
    const int knob_u = dev_env_int("FXC_RTC_U", 0);
    int tried = 0, n_chosen = 0, index = -1;
    const std::vector<std::vector<int>> lists = spec_stage_lists(n, taps, rows, variant, &n_chosen);
    // Measured (profiles/r06/tune_spec.md): a ranked list that ends up with ONE frame per step where the prime-factor order carries TWO loses
    // to it (0.75 - 0.98 x over seven channel counts); with as many frames it wins (1.00 - 1.31 x).  Such a build is set aside until the
    // prime-factor order has shown what it gets.
    SpecBuild aside;
    for (const std::vector<int>& list : lists) {
        ++index;
        if (tried == 3 + !aside.image.empty()) break;                       // (a compile is a second or two: three lists at most, one more behind a build set aside)
        const SpecShape one = spec_shape_of(n, taps, list.data(), (int)list.size(), 1, rows);
        if (!one.ok) continue;
        ++tried;
        const SpecShape two = spec_shape_of(n, taps, list.data(), (int)list.size(), 2, rows);
        int force_u = knob_u;
        bool tuned = false;
        if (!force_u)      // a measured choice (spec_tuned.h) names its frames per step too
            if (const SpecTuned* t = spec_tuned_entry(n, taps, variant))
                if (t->n_stages == (int)list.size() && std::equal(list.begin(), list.end(), t->radix)) force_u = t->u, tuned = true;
        SpecBuild b = (force_u == 2 && two.ok) ? SpecBuild() : spec_build_laid_out(one, variant, arch);
        if (two.ok && force_u != 1) {
            SpecBuild b2 = spec_build_laid_out(two, variant, arch);
            if (b2.resident >= 1 && (b2.resident >= b.resident || force_u == 2)) b = std::move(b2);
            else if (b.image.empty()) b = spec_build_laid_out(one, variant, arch);      // (two frames were asked for and do not fit)
        }
        if (b.image.empty() || b.resident < 1) {
            if (best.image.empty() && !b.error.empty()) best.error = b.error;
            else if (best.image.empty() && b.scratch) best.error = "the specialised kernel spills (" + std::to_string(b.scratch) + " B of scratch per lane)";
            continue;
        }
        if (index < n_chosen && !tuned && !knob_u && b.shape.u == 1 && two.ok && n_chosen < (int)lists.size()) {
            if (aside.image.empty()) aside = std::move(b);
            continue;
        }
        if (tuned) {                                                   // a measured choice: taken as it is
            best = std::move(b);
            break;
        }
        if (best.image.empty() || b.resident * b.shape.threads() > best.resident * best.shape.threads()) best = std::move(b);
        if (best.resident * best.shape.threads() >= 512) break;       // two workgroups of 256 (or one of 512 and more): good enough
    }
    if (!aside.image.empty() && (best.image.empty() || best.shape.u == 1)) best = std::move(aside);      // (the prime-factor order carries one frame too)
    return best;
}

// every developer knob that reaches a build, as part of the in-process cache's key (empty in the shipped library)
std::string spec_knob_key() {
    std::string k;
#if FXC_DEV_KERNELS
    for (const char* name : {"FXC_RTC_ABL", "FXC_RTC_R0", "FXC_RTC_U", "FXC_RTC_LEAN_ABOVE", "FXC_RTC_TPR_MAX", "FXC_RTC_ROWS1_ABOVE",
                             "FXC_RTC_BIG_PRIMES", "FXC_RTC_RADICES", "FXC_RTC_GROUPS", "FXC_RTC_PADS", "FXC_RTC_PLANE0", "FXC_RTC_LAYOUT", "FXC_RTC_TWFULL",
                             "FXC_RTC_WAVES", "FXC_RTC_COMPOSITE", "FXC_RTC_PICK", "FXC_RTC_TUNED", "FXC_RTC_TW_EARLY", "FXC_RTC_OOB_ZERO"}) {
        const char* e = std::getenv(name);
        k += std::string(e ? e : "") + ";";
    }
#endif
    return k;
}

// compile (or find) the kernel for n channels on `device`; never nullptr -- a failed build is cached with its reason
const SpecKernel* spec_kernel(int device, int n, int taps, int variant) {
    char key[256];
    std::snprintf(key, sizeof key, "d%d n%d t%d v%d %s", device, n, taps, variant, spec_knob_key().c_str());
    std::lock_guard<std::mutex> lock(g_spec_mutex);
    auto it = g_spec_cache.find(key);
    if (it != g_spec_cache.end()) return it->second;
    SpecKernel* k = new SpecKernel;
    g_spec_cache[key] = k;
    hipDeviceProp_t prop;
    // (a failure that may not repeat -- the device query, the module load, device memory -- is not remembered: the next plan tries again)
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) {
        k->error = "hipGetDeviceProperties failed";
        g_spec_cache.erase(key);
        return k;
    }
    const auto t0 = std::chrono::steady_clock::now();
    SpecBuild b = spec_search(n, taps, variant, spec_arch(prop.gcnArchName).c_str());
    k->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    k->source = b.source;
    if (b.image.empty()) {
        k->error = b.error;
        if (b.error.find("libhiprtc") != std::string::npos) g_spec_cache.erase(key);      // (no compiler in reach yet: not a property of the shape)
        return k;
    }
    k->shape = b.shape;
    k->vgprs = (int)b.vgprs;
    if (env_int("FXC_RTC_VERBOSE", 0))
        std::fprintf(stderr, "libfxcorr: %d channels, %d taps, build %d: stages=%s groups=%s pads=%s plane0=%d frames_per_step=%d threads=%d vgprs=%lld resident=%d lds=%zu source=%s (%.2f s)\n", n,
                     taps, variant, b.shape.list(b.shape.radix).c_str(), b.shape.list(b.shape.grp).c_str(), b.shape.list(b.shape.pad).c_str(), b.shape.plane0,
                     b.shape.u, b.shape.threads(), b.vgprs, b.resident, b.shape.lds_bytes(),
                     b.source == kSpecPrebuilt ? "prebuilt" : b.source == kSpecCached ? "cache" : "built", k->seconds);
    DeviceGuard guard(device);
    hipError_t e = hipModuleLoadData(&k->module, b.image.data());
    if (e == hipSuccess) e = hipModuleGetFunction(&k->fn, k->module, "fxm_fx2_kernel");
    if (e != hipSuccess) {
        k->error = std::string("loading the compiled kernel: ") + hipGetErrorString(e);
        k->fn = nullptr;
        g_spec_cache.erase(key);
        return k;
    }
    if (k->shape.lean) {
        const std::vector<cf> t = spec_tw1_table(k->shape);
        if (hipMalloc(&k->d_tw1, t.size() * sizeof(cf)) != hipSuccess ||
            hipMemcpy(k->d_tw1, t.data(), t.size() * sizeof(cf), hipMemcpyHostToDevice) != hipSuccess) {
            k->error = "the twiddle table of the lean build: out of device memory";
            k->fn = nullptr;
            g_spec_cache.erase(key);
            return k;
        }
    }
    int blocks = 0;
    k->wgs_per_cu = std::max(1, b.resident);
    if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, k->fn, k->shape.threads(), 0) == hipSuccess && blocks > 0)
        k->wgs_per_cu = blocks;
    return k;
}

}  // namespace
