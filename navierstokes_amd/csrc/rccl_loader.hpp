// rccl_loader.hpp — RCCL resolved at run time (dlopen), shared by the library and tools/comm_timing.hip.
// No link-time dependency: the library must load (and plan partitions) on machines
// without RCCL.  dlopen picks up the copy already in the process (torch's) if any.
#pragma once
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cstdlib>
#include <mutex>
#include <string>

namespace mi355 {

constexpr int kCommIdBytes = 128;
struct IdByValue { // ncclUniqueId, passed BY VALUE to ncclCommInitRank
    char internal[kCommIdBytes];
};

struct Rccl {
    bool tried = false, ok = false;
    std::string why;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, IdByValue, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*AllToAllv)(const void*, const size_t*, const size_t*, void*, const size_t*, const size_t*, int, void*, hipStream_t) = nullptr; // optional
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr; // optional (the wide-halo form of the step)
    int (*CommCount)(void*, int*) = nullptr;    // optional: what the communicator itself says its size is
    int (*CommUserRank)(void*, int*) = nullptr; // optional
};
constexpr int kNcclDouble = 8; // ncclFloat64 (rccl.h)

inline Rccl& rccl_state()
{
    static Rccl r;
    return r;
}

// thread-safe; the result is cached
inline bool rccl_load()
{
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    Rccl& R = rccl_state();
    if (R.tried) return R.ok;
    R.tried = true;
    void* h = nullptr;
    // MI355_RCCL_LIBRARY: a specific RCCL build (or the tests' in-process stand-in, tests/fake_rccl)
    if (const char* e = getenv("MI355_RCCL_LIBRARY")) {
        if (!(h = dlopen(e, RTLD_NOW | RTLD_LOCAL))) {
            R.why = std::string("dlopen(") + e + "): " + (dlerror() ? dlerror() : "failed");
            return false;
        }
    }
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* nm : names)
        if (h || (h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) {
        R.why = std::string("dlopen(librccl): ") + (dlerror() ? dlerror() : "not found");
        return false;
    }
    auto sym = [&](const char* nm) -> void* {
        void* p = dlsym(h, nm);
        if (!p) R.why = std::string("dlsym ") + nm + " failed";
        return p;
    };
    R.GetUniqueId = (int (*)(void*))sym("ncclGetUniqueId");
    R.CommInitRank = (int (*)(void**, int, IdByValue, int))sym("ncclCommInitRank");
    R.CommDestroy = (int (*)(void*))sym("ncclCommDestroy");
    R.GroupStart = (int (*)())sym("ncclGroupStart");
    R.GroupEnd = (int (*)())sym("ncclGroupEnd");
    R.Send = (int (*)(const void*, size_t, int, int, void*, hipStream_t))sym("ncclSend");
    R.Recv = (int (*)(void*, size_t, int, int, void*, hipStream_t))sym("ncclRecv");
    R.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
    R.AllToAllv = (int (*)(const void*, const size_t*, const size_t*, void*, const size_t*, const size_t*, int, void*, hipStream_t))dlsym(h, "ncclAllToAllv");
    R.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(h, "ncclAllGather");
    R.CommCount = (int (*)(void*, int*))dlsym(h, "ncclCommCount");
    R.CommUserRank = (int (*)(void*, int*))dlsym(h, "ncclCommUserRank");
    R.ok = R.GetUniqueId && R.CommInitRank && R.CommDestroy && R.GroupStart && R.GroupEnd && R.Send && R.Recv && R.GetErrorString;
    return R.ok;
}

} // namespace mi355
