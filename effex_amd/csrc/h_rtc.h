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

namespace {

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
        if (!ok) a->handle = nullptr;
        return a;
    }();
    return api;
}

// how fx_spec.h is cut for one channel count: stage order, threads per slot, slots per workgroup
struct SpecShape {
    bool ok = false;
    int n = 0, taps = 0, n_stages = 0, radix[fxc::kMixedMaxStages] = {0}, tpr = 0, slots = 0;
    int threads() const { return tpr * slots; }
    size_t lds_bytes() const { return n_stages >= 2 ? (size_t)slots * 4 * n * sizeof(cf) : 0; }
};

// Eligible: two antennas, up to four taps (the frame ring lives in registers), every prime factor has a register butterfly
// (2, 3, 4, 5, 7, 11, 13), a thread's points (first radix x its first-stage butterflies) fit the ring (<= 8), the slots'
// rows fit the LDS.  Stage order as fx_mixed.h's: fours, a two, the odd primes ascending.
SpecShape spec_shape(int n, int taps) {
    SpecShape s;
    if (n < 2 || n > 8192 || taps < 1 || taps > 4) return s;
    const fxc::MixedPlan mp = fxc::mixed_factor(n);
    if (mp.n_stages < 1) return s;
    for (int i = 0; i < mp.n_stages; ++i) {
        const int r = mp.radix[i];
        if (!(r == 2 || r == 3 || r == 4 || r == 5 || r == 7 || r == 11 || r == 13)) return s;
        s.radix[i] = r;
    }
    s.n = n;
    s.taps = taps;
    s.n_stages = mp.n_stages;
    const int nb0 = n / s.radix[0];
    int j0 = 1;
    if (nb0 <= 64) {
        s.tpr = 1;
        while (s.tpr < nb0) s.tpr <<= 1;
    } else {
        j0 = (nb0 + 1023) / 1024;
        s.tpr = ((nb0 + j0 - 1) / j0 + 63) / 64 * 64;
    }
    s.slots = std::max(1, 256 / s.tpr);
    if (s.radix[0] * j0 > 8) return s;
    if (s.lds_bytes() > (size_t)(160 * 1024)) return s;
    s.ok = true;
    return s;
}

struct SpecKernel {
    hipModule_t module = nullptr;
    hipFunction_t fn = nullptr;
    SpecShape shape;
    int wgs_per_cu = 1;          // resident workgroups per CU (occupancy query)
    int vgprs = 0;
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
};

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

// fx_spec.h for one shape -> a code object for `arch` (e.g. "gfx950:sramecc+:xnack-"); needs no device
bool spec_compile(const SpecShape& shape, bool u8, const char* arch, std::vector<char>& image, std::string& error) {
    RtcApi* api = rtc_api();
    if (!api->handle) {
        error = api->error;
        return false;
    }
    std::string radices;
    for (int i = 0; i < shape.n_stages; ++i) radices += (i ? "," : "") + std::to_string(shape.radix[i]);
    std::vector<std::string> opts = {std::string("--offload-arch=") + arch, "-O3", "-std=c++17", "-fno-slp-vectorize",
                                     "-DFXM_N=" + std::to_string(shape.n), "-DFXM_T=" + std::to_string(shape.taps),
                                     "-DFXM_TPR=" + std::to_string(shape.tpr), "-DFXM_SLOTS=" + std::to_string(shape.slots),
                                     "-DFXM_NST=" + std::to_string(shape.n_stages), "-DFXM_RADICES=" + radices,
                                     "-DFXM_U8=" + std::to_string((int)u8)};
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
    return true;
}

// compile (or find) the kernel for `shape` on `device`; never nullptr -- a failed build is cached with its reason
const SpecKernel* spec_kernel(int device, const SpecShape& shape, bool u8) {
    char key[256];
    std::string radices;
    for (int i = 0; i < shape.n_stages; ++i) radices += (i ? "," : "") + std::to_string(shape.radix[i]);
    std::snprintf(key, sizeof key, "d%d n%d t%d tpr%d s%d r%s u%d", device, shape.n, shape.taps, shape.tpr, shape.slots, radices.c_str(), (int)u8);
    std::lock_guard<std::mutex> lock(g_spec_mutex);
    auto it = g_spec_cache.find(key);
    if (it != g_spec_cache.end()) return it->second;
    SpecKernel* k = new SpecKernel;
    k->shape = shape;
    g_spec_cache[key] = k;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) {
        k->error = "hipGetDeviceProperties failed";
        return k;
    }
    std::vector<char> image;
    if (!spec_compile(shape, u8, prop.gcnArchName, image, k->error)) return k;
    DeviceGuard guard(device);
    hipError_t e = hipModuleLoadData(&k->module, image.data());
    if (e == hipSuccess) e = hipModuleGetFunction(&k->fn, k->module, "fxm_fx2_kernel");
    if (e != hipSuccess) {
        k->error = std::string("loading the compiled kernel: ") + hipGetErrorString(e);
        k->fn = nullptr;
        return k;
    }
    int blocks = 0;
    const long long scratch = code_object_int(image, ".private_segment_fixed_size");
    k->vgprs = (int)code_object_int(image, ".vgpr_count");
    if (scratch != 0 && !env_int("FXC_RTC_ALLOW_SPILLS", 0)) {      // a shape whose registers spill: the any-shape kernel is the better one
        k->error = "the specialised kernel spills (" + std::to_string(scratch) + " B of scratch per lane)";
        k->fn = nullptr;
        return k;
    }
    if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, k->fn, shape.threads(), 0) == hipSuccess && blocks > 0)
        k->wgs_per_cu = blocks;
    return k;
}

}  // namespace
