// Part of libfxcorr's single translation unit: included by fxcorr.hip (not a stand-alone header).
#pragma once

// ------------------------------------------------------------------------------------------
// multi-GPU reduce: one RCCL sum of the exported accumulators over xGMI, enqueued on the plan's
// stream.  librccl is bound at run time (the copy the process already has -- PyTorch ships one -- else ROCm's), so
// single-GPU users need no RCCL at all.
// ------------------------------------------------------------------------------------------
// what fxc_comm_create hands out: the ncclComm_t and the device it lives on (fxc_reduce checks it against the plan's)
struct fxc_comm {
    static constexpr unsigned kMagic = 0x46584343u;   // "FXCC"
    unsigned magic = kMagic;
    ncclComm_t comm = nullptr;
    int device = -1, rank = 0, world_size = 1;
    long long reduces = 0;          // collectives fxc_reduce has queued on it
};

namespace {

struct RcclApi {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclReduce) reduce = nullptr;
    decltype(&ncclAllReduce) all_reduce = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    // what the communicator says about itself (fxc_comm_info); optional: an RCCL without them still reduces
    decltype(&ncclCommCount) comm_count = nullptr;
    decltype(&ncclCommUserRank) comm_user_rank = nullptr;
    decltype(&ncclCommCuDevice) comm_cu_device = nullptr;
    decltype(&ncclGetVersion) get_version = nullptr;
    decltype(&ncclCommGetAsyncError) comm_async_error = nullptr;
    std::string error, path;
};

void rccl_bind(RcclApi& api) {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names)   // a copy that is already mapped wins: one RCCL per process
        if ((api.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    for (size_t k = 0; !api.handle && k < sizeof names / sizeof *names; ++k) api.handle = dlopen(names[k], RTLD_NOW | RTLD_GLOBAL);
    if (!api.handle) {
        const char* why = dlerror();
        api.error = std::string("librccl not found: ") + (why ? why : "dlopen failed");
        return;
    }
    bool ok = true;
    auto bind = [&](const char* sym) {
        void* f = dlsym(api.handle, sym);
        if (!f) {
            ok = false;
            api.error = std::string("librccl lacks ") + sym;
        }
        return f;
    };
    api.get_unique_id = reinterpret_cast<decltype(api.get_unique_id)>(bind("ncclGetUniqueId"));
    api.comm_init_rank = reinterpret_cast<decltype(api.comm_init_rank)>(bind("ncclCommInitRank"));
    api.comm_destroy = reinterpret_cast<decltype(api.comm_destroy)>(bind("ncclCommDestroy"));
    api.reduce = reinterpret_cast<decltype(api.reduce)>(bind("ncclReduce"));
    api.all_reduce = reinterpret_cast<decltype(api.all_reduce)>(bind("ncclAllReduce"));
    api.error_string = reinterpret_cast<decltype(api.error_string)>(bind("ncclGetErrorString"));
    if (!ok) {
        api.handle = nullptr;
        return;
    }
    api.comm_count = reinterpret_cast<decltype(api.comm_count)>(dlsym(api.handle, "ncclCommCount"));
    api.comm_user_rank = reinterpret_cast<decltype(api.comm_user_rank)>(dlsym(api.handle, "ncclCommUserRank"));
    api.comm_cu_device = reinterpret_cast<decltype(api.comm_cu_device)>(dlsym(api.handle, "ncclCommCuDevice"));
    api.get_version = reinterpret_cast<decltype(api.get_version)>(dlsym(api.handle, "ncclGetVersion"));
    api.comm_async_error = reinterpret_cast<decltype(api.comm_async_error)>(dlsym(api.handle, "ncclCommGetAsyncError"));
    Dl_info where;                           // which copy was bound: the line of an N > 1 run names it
    if (api.reduce && dladdr(reinterpret_cast<void*>(api.reduce), &where) && where.dli_fname) api.path = where.dli_fname;
}

// bound once per process, on first use (a function-local static: initialised exactly once even with several caller threads)
RcclApi* rccl_api() {
    static RcclApi* api = [] {
        RcclApi* a = new RcclApi;
        rccl_bind(*a);
        return a;
    }();
    return api;
}

int rccl_fail(const fxc_plan* p, const RcclApi* api, const char* what, ncclResult_t r) {
    return fail(p, FXC_ERR_COMM, "%s failed: %s", what, api->error_string ? api->error_string(r) : "RCCL error");
}

}  // namespace
