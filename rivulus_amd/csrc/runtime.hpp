// Host runtime behind the C ABI: context (device + stream), pooled HBM allocator,
// device column handles, error plumbing.  C++17, compiled with hipcc.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rivulus_gpu.h"

namespace rvh {

struct Error : std::runtime_error {
    rv_status status;
    Error(rv_status s, const std::string &m) : std::runtime_error(m), status(s) {}
};

inline std::string fmt(const char *f, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, f);
    vsnprintf(buf, sizeof buf, f, ap);
    va_end(ap);
    return buf;
}

#define RV_HIP(expr)                                                                                  \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess)                                                                         \
            throw rvh::Error(e_ == hipErrorOutOfMemory ? RV_ERR_OOM : RV_ERR_DEVICE,                  \
                             rvh::fmt("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                                      __LINE__));                                                     \
    } while (0)

inline void require(bool cond, rv_status s, const std::string &msg) {
    if (!cond) throw Error(s, msg);
}

// thread-local text behind rv_last_error() (defined in core.hip; shared by every unit of the library)
std::string &last_error();

// body of every extern "C" entry point: no exception crosses the ABI
template <class F>
rv_status guarded(F f) {
    try {
        f();
        return RV_OK;
    } catch (const Error &e) {
        last_error() = e.what();
        return e.status;
    } catch (const std::bad_alloc &) {
        last_error() = "host allocation failed";
        return RV_ERR_OOM;
    } catch (const std::exception &e) {
        last_error() = e.what();
        return RV_ERR_INTERNAL;
    }
}

// Size-bucketed free list of HBM blocks.  Query outputs are sized for the worst case
// (every row survives), so blocks are large and few: reuse beats hipMalloc/hipFree
// round trips inside a streaming pipeline.  Outlives the context through shared_ptr.
class Pool {
  public:
    explicit Pool(int device) : device_(device) {}
    ~Pool() { release_all(); }
    void *alloc(size_t bytes, size_t *got) {
        bytes = (bytes + 255) & ~size_t(255);
        if (bytes == 0) bytes = 256;
        {
            std::lock_guard<std::mutex> g(mu_);
            auto it = free_.lower_bound(bytes);
            if (it != free_.end() && it->first <= bytes + bytes / 4 + (1u << 20)) {
                void *p = it->second;
                *got = it->first;
                free_.erase(it);
                return p;
            }
        }
        void *p = nullptr;
        static const bool trace = getenv("RV_TRACE_POOL") != nullptr;  // diagnostic
        if (trace) fprintf(stderr, "[pool] hipMalloc %zu bytes\n", bytes);
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            release_all();
            e = hipMalloc(&p, bytes);
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            throw Error(RV_ERR_OOM, fmt("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e)));
        }
        *got = bytes;
        return p;
    }
    void give_back(void *p, size_t bytes) {
        std::lock_guard<std::mutex> g(mu_);
        free_.emplace(bytes, p);
    }
    void release_all() {
        std::lock_guard<std::mutex> g(mu_);
        for (auto &kv : free_) (void)hipFree(kv.second);
        free_.clear();
    }

  private:
    int device_;
    std::mutex mu_;
    std::multimap<size_t, void *> free_;
};

struct DevBuf {
    void *ptr = nullptr;
    size_t bytes = 0;  // readable bytes (pool blocks: the whole block)
    uint64_t id = 0;   // pool blocks: unique per allocation (the pool hands a freed table's address to the next one: what a predicate's
                       // remembered selectivity is keyed on must not be inherited with it); 0: caller-owned memory, keyed by address
    std::shared_ptr<Pool> pool;  // null: caller-owned memory (rv_wrap)
    ~DevBuf() {
        if (pool && ptr) pool->give_back(ptr, bytes);
    }
};
using DevBufRef = std::shared_ptr<DevBuf>;

}  // namespace rvh

// Device-resident PrimitiveArray / BooleanArray (reference primitive.rs:20-28, boolean.rs:9-16).
struct rv_dcolumn {
    rv_dtype dtype = RV_NULL;
    rvh::DevBufRef values;
    rvh::DevBufRef validity;  // null: no null bitmap
    rvh::DevBufRef offsets;   // RV_STRING: int32 offsets (values = the UTF-8 bytes)
    uint64_t data_bytes = 0;  // RV_STRING: bytes in `values`
    uint64_t offset = 0;
    uint64_t length = 0;
    int64_t null_count = -1;  // -1 unknown
};

struct rv_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipStream_t copy_stream = nullptr;  // chunk uploads of rv_filter_project_host (created on first use)
    hipEvent_t ev_up[2] = {nullptr, nullptr};
    hipEvent_t ev_main = nullptr;  // main stream -> copy stream ordering
    std::shared_ptr<rvh::Pool> pool;
    hipDeviceProp_t props{};
    // control block + look-back descriptors (one allocation, one memset per launch)
    void *d_ctrl = nullptr;
    size_t ctrl_bytes = 0;
    void *h_ctrl = nullptr;  // pinned mirror of the first kCtrlBytes
    // striped counters (device_common.hpp, striped_add): kStripeSlots x kStripes words in separate 128-byte lines, all zero
    // between queries; stripe_mask = slots a kernel of the current query was pointed at (folded by fetch_ctrl)
    void *d_stripes = nullptr;
    uint32_t stripe_mask = 0;
    // kernel-only timing (option profile_kernels)
    hipEvent_t evk0 = nullptr, evk1 = nullptr;
    double kernel_ms = 0.0;
    uint64_t kernel_launches = 0;
    std::string last_kernel;  // instantiation of the last hot-path launch (rv_ctx_last_kernel)
    // options
    int64_t opt_profile = 0;
    int64_t opt_rows_per_lane = 0;  // 0 = default per column count
    int64_t opt_vec = 0;            // 0 auto, 1 force 8-byte loads, 2 force 16-byte loads
    int64_t opt_cap_rows = 0;       // 0 = as many as LDS allows
    int64_t opt_wgs_per_cu = 0;     // 0 = occupancy query
    double last_redo_fraction = 0.0;  // share of tiles the last fused launch left to the redo kernel
    int64_t opt_stamp = 0;          // diagnostic: run the FF_STAMP instantiation
    int64_t opt_direct = 0;         // 0 auto (dense selections of plain value columns), 1 whenever eligible, -1 never: the direct kernel
    int64_t opt_direct_r = 0;       // diagnostic: rows per lane of the direct kernel (0 = the first listed instantiation)
    int64_t opt_direct_waves = 0;   // diagnostic: waves per workgroup of the direct kernel (0 = the first listed instantiation)
    int64_t opt_roomy = 0;          // diagnostic: 1 = size the LDS slots as for a dense selection (144 KiB, two stages)
    int64_t opt_depth = 0;          // 0 auto, 1 / 2: iterations between a tile's aggregate and its write-out
    int64_t opt_agg_grid = 0;       // filter + aggregate: workgroups per CU striding over the tiles (0 = a constant 8192 workgroups, -1 = one workgroup per tile)
    int64_t opt_spin_limit = 0;     // polls before a look-back / the scanner gives up on a missing descriptor (0 = kSpinLimit)
    int64_t opt_debug = 0;          // diagnostic ablations (results are wrong): 1 no output stores, 2 no look-back
    int64_t opt_out_sizing = 0;     // 0: from the predicate's known selectivity (tables of 2^25 rows and more), else for every row; -1: always for every row; 1: last selectivity x 1.5 + 1 %; k >= 2: k rows per million
    double last_selectivity = -1.0; // survivors / rows of the last fused launch (-1: none yet)
    // selectivity the last passes of the last few PREDICATES had (signature: terms, literals, null policy, expression):
    // what a fused launch is sized from -- a stream's windows share their predicate; an unrelated query does not inherit it
    struct SeenPredicate {
        uint64_t signature = 0;
        double selectivity = -1.0;
        // How unevenly the survivors are spread (a table sorted or clustered on the predicate's column): the share of a staged pass's
        // wave ranges that outgrew their LDS slot the last time one ran (and the selectivity it had then), and the strided sample's
        // histogram of 1024-row blocks by the share of their rows that survive (16 buckets; empty: never sampled).
        double redo_fraction = -1.0, redo_at = -1.0, redo_ratio = 0.0;  // redo_ratio: the slot's share of a wave's rows in that pass
        float hist[16] = {};
        bool have_hist = false;
        // ... and its profile: the survivors (of 1024 rows) of every sampled block in table order, `profile_stride` rows apart
        uint16_t profile[1024] = {};
        uint64_t profile_stride = 0, profile_rows = 0;  // 0: no profile
    };
    SeenPredicate seen[8];
    unsigned seen_next = 0;
    double seen_selectivity(uint64_t signature) const {
        for (const SeenPredicate &q : seen)
            if (q.signature == signature && q.selectivity >= 0.0) return q.selectivity;
        return -1.0;
    }
    SeenPredicate *seen_entry(uint64_t signature) {
        for (SeenPredicate &q : seen)
            if (q.signature == signature && q.selectivity >= 0.0) return &q;
        return nullptr;
    }
    SeenPredicate *remember_selectivity(uint64_t signature, double selectivity) {
        if (SeenPredicate *q = seen_entry(signature)) {
            q->selectivity = selectivity;
            return q;
        }
        SeenPredicate &q = seen[seen_next++ % 8];
        q = SeenPredicate{};
        q.signature = signature;
        q.selectivity = selectivity;
        return &q;
    }
    // the strided selectivity sample of a predicate the context has not seen (agg_kernel.hpp, sample_count_kernel)
    unsigned long long *d_sample = nullptr;           // [kSampleWords], zero between samples
    volatile unsigned long long *h_sample = nullptr;  // [kSampleWords] pinned: {survivors, sequence, packed histogram}
    float last_sample_hist[16] = {};                  // of the last sample taken: share of the blocks per bucket of 64 surviving rows
    uint16_t last_sample_profile[1024] = {};          // ... and every block's survivors, in table order
    uint64_t last_sample_stride = 0, last_sample_rows = 0;
    unsigned long long sample_seq = 0;
    uint64_t samples_taken = 0;
    uint64_t last_rows_out = 0, last_rows_in = 0;  // of the last fused pass
    int64_t opt_speculative_batches = 0;   // rv_filter_project_batches: -1 = never launch before the handles are validated
    uint64_t speculative_batch_passes = 0; // windows whose pass ran while the handle walk validated them
    int64_t opt_str_tiles_from = 0; // String columns of a filter in source-tile order from this expected selectivity (percent) on; 0: 50 %; -1: never; 1: always
    int64_t opt_bool_cap = 0;       // k > 0: Boolean columns compacted behind the pass get output bitmaps of at most k rows (tests of the fallback)
    int64_t opt_groups_by_ranges = 0; // later column groups of a wide projection at the first pass's wave offsets: 0 = up to 55 % of the rows surviving (plain columns the predicate does not read are left to it from 25 % down, nullable ones always), 1 = always, -1 = never (passes of their own)
    int64_t opt_sample = 0;         // 0: sample unseen predicates over tables of >= 2^25 rows; -1: never; k > 0: from k rows on
    int64_t opt_segments = 0;       // 0: a table of 2^28 rows and more whose survivors sit in a few long stretches (sorted on the predicate's column) is filtered stretch by stretch; k > 0: from k rows on; -1: never
    uint64_t segmented_passes = 0;  // queries that ran that way
    uint64_t segment_fallbacks = 0; // ... that started that way and ran as one pass after all (more survivors than the profile promised)
    int64_t opt_skew = 0;           // 0: a selection whose survivors come in runs takes the direct kernel where the redo kernel would cost more; -1: never
    uint64_t fused_rows_scanned = 0;    // input rows of every fused filter launch so far (a Limit that is pushed down shows here)
    uint64_t batch_counts_in_pass = 0;  // launches whose per-batch survivor counts came out of the pass itself (BatchReq)
    uint64_t overflow_reruns = 0;   // launches re-run because the speculative outputs were too small
    int64_t opt_inject_failure = 0; // fault injection: this many upcoming query calls fail with RV_ERR_DEVICE before launching
    bool undrained = false;         // a group's failed collective may still sit on this context's stream (group.hip, drop_comms): destroy waits
                                    // for it a bounded time and leaks the context's device memory rather than wait without end
    int64_t opt_bools_in_pass = 0;  // 1: projected Boolean columns are compacted inside the fused pass (lane-form PEXT)
    unsigned long long last_stamps[32] = {};
    // per (kernel, dynamic LDS bytes): resident workgroups per CU; per kernel: largest LDS size enabled so far
    std::map<std::pair<const void *, size_t>, int> occupancy;
    std::map<const void *, size_t> lds_enabled;
    // control blocks of fused launches (one per launch in flight; recycled): device block + pinned mirror + event
    struct LaunchCtrl {
        void *dev = nullptr;
        size_t bytes = 0;
        void *host = nullptr;
        hipEvent_t ev = nullptr;
        hipEvent_t tk0 = nullptr, tk1 = nullptr;  // option profile_kernels: around THIS launch's kernel(s) (several launches may be in flight)
        // A free block is zeroed for its next user by the pass that runs before that user's (fused_kernel.hpp: the idle waves of the
        // scanner's workgroup), instead of by a memset queued between two passes: `clean` = bytes from the start that are zero for any
        // work queued on the context's stream from now on; `dirty` = bytes the block's last use may have written.
        size_t clean = 0, dirty = 0;
    };
    std::vector<LaunchCtrl> ctrl_free;
    // pinned host staging for host <-> device transfers of a few megabytes (per-batch tables of
    // rv_filter_project_batches): an asynchronous copy from / to PAGEABLE memory of that size makes the runtime pin
    // and unpin the pages on the fly, which showed up as ~20 ms stalls in the following synchronisation
    void *h_stage = nullptr;
    size_t stage_bytes = 0;
    void *stage(size_t bytes) {
        if (bytes > stage_bytes) {
            if (h_stage) (void)hipHostFree(h_stage);
            h_stage = nullptr;
            stage_bytes = 0;
            const size_t want = std::max(bytes + bytes / 2, static_cast<size_t>(1) << 20);
            if (hipHostMalloc(&h_stage, want, hipHostMallocDefault) != hipSuccess) {
                (void)hipGetLastError();
                throw rvh::Error(RV_ERR_OOM, rvh::fmt("cannot pin %zu bytes of host staging", want));
            }
            stage_bytes = want;
        }
        return h_stage;
    }
};
