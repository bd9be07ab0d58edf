// RCCL bound at first use with dlopen: the single-GPU path must not depend on librccl being loadable.
// Only the handful of entry points the backend needs (the 16-byte {SUM, COUNT} all-reduce of BASELINE
// config 5) are resolved; constants are RCCL's (rccl.h): ncclInt64 == 4, ncclFloat64 == 8, ncclSum == 0.
#pragma once

#include <dlfcn.h>

#include <array>
#include <mutex>

#include "runtime.hpp"

namespace rvh {

constexpr int kNcclInt64 = 4, kNcclFloat64 = 8, kNcclSum = 0;

struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, /* ncclUniqueId by value */ std::array<char, RV_COMM_ID_BYTES>, int) = nullptr;
    int (*CommInitAll)(void **, int, const int *) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*CommAbort)(void *) = nullptr;  // optional: a communicator whose collective cannot complete is aborted, not destroyed
    int (*CommCount)(const void *, int *) = nullptr;  // optional: ranks of a communicator (reported by the bench line)
    const char *(*GetErrorString)(int) = nullptr;
};
inline Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
        }
        if (!r.lib) return;
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.lib, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.lib, "ncclCommInitRank"));
        r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(r.lib, "ncclCommInitAll"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(r.lib, "ncclAllReduce"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(r.lib, "ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(r.lib, "ncclGroupEnd"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
        r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(dlsym(r.lib, "ncclCommAbort"));
        r.CommCount = reinterpret_cast<decltype(r.CommCount)>(dlsym(r.lib, "ncclCommCount"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
    });
    require(r.lib && r.GetUniqueId && r.CommInitRank && r.CommInitAll && r.AllReduce && r.GroupStart && r.GroupEnd && r.CommDestroy,
            RV_ERR_DEVICE, "librccl.so could not be loaded: multi-GPU aggregates need RCCL");
    return r;
}
inline void rccl_check(int rc, const char *what) {
    if (rc != 0) throw Error(RV_ERR_DEVICE, fmt("%s failed: %s", what, rccl().GetErrorString ? rccl().GetErrorString(rc) : "?"));
}

}  // namespace rvh
