// rv_group_*: the single-process multi-GPU driver (include/rivulus_gpu.h, SURVEY.md sections 8b / 8e).
//
// A group is N contexts (one per listed device) and N host worker threads.  A table is sharded by row range;
// filter / project needs no collective: every device runs the same single-pass kernels on its shard, the host
// prefix-sums the N survivor counts and every device copies its output into its slice of ONE pinned host buffer
// per column, in rank order == row order -- the gather the reference does with collect_stream_batches ->
// RecordBatch::concat (src/physical_plan/streaming.rs:343-352, src/execution/record_batch.rs:245-342), with the
// shards as the batches.  The only exchange is the final scalar of a global aggregate: one RCCL all-reduce of
// 16 bytes (BASELINE configs[4]).
//
// Written over the public C ABI on purpose: a shard is an ordinary rv_dcolumn of an ordinary rv_ctx, so
// whatever a single context can filter, the group can.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <thread>

#include "rccl_loader.hpp"

using namespace rvh;

namespace {

// One worker thread per rank: HIP device selection is per thread, and N devices are only busy together when N
// host threads launch and wait on them.  A query is ~1 ms of device time: both sides of the hand-over spin for a
// while before they sleep on the condition variable (a futex wake-up is 30-60 us, several per cent of a step).
class Worker {
  public:
    Worker() : th_([this] { loop(); }) {}
    ~Worker() {
        {
            std::lock_guard<std::mutex> g(mu_);
            stop_ = true;
            posted_.store(true, std::memory_order_release);
        }
        cv_.notify_all();
        th_.join();
    }
    void post(std::function<void()> job) {
        {
            std::lock_guard<std::mutex> g(mu_);
            job_ = std::move(job);
            busy_ = true;
            error_ = nullptr;
            done_.store(false, std::memory_order_relaxed);
            posted_.store(true, std::memory_order_release);
        }
        cv_.notify_all();
    }
    void wait() {  // rethrows what the job threw
        spin_until(done_, kWaitSpinUs);
        std::unique_lock<std::mutex> g(mu_);
        cv_.wait(g, [this] { return !busy_; });
        if (error_) {
            auto e = error_;
            error_ = nullptr;
            std::rethrow_exception(e);
        }
    }

  private:
    static constexpr int kIdleSpinUs = 300, kWaitSpinUs = 20000;
    static void spin_until(const std::atomic<bool> &flag, int budget_us) {
        const auto t0 = std::chrono::steady_clock::now();
        while (!flag.load(std::memory_order_acquire)) {
            for (int i = 0; i < 64; ++i) __builtin_ia32_pause();
            if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(budget_us)) return;
        }
    }
    void loop() {
        for (;;) {
            std::function<void()> job;
            spin_until(posted_, kIdleSpinUs);
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [this] { return stop_ || (busy_ && job_); });
                if (stop_) return;
                job = std::move(job_);
                job_ = nullptr;
                posted_.store(false, std::memory_order_relaxed);
            }
            std::exception_ptr err;
            try {
                job();
            } catch (...) {
                err = std::current_exception();
            }
            {
                std::lock_guard<std::mutex> g(mu_);
                error_ = err;
                busy_ = false;
                done_.store(true, std::memory_order_release);
            }
            cv_.notify_all();
        }
    }
    std::mutex mu_;
    std::condition_variable cv_;
    std::function<void()> job_;
    bool busy_ = false, stop_ = false;
    std::atomic<bool> posted_{false}, done_{true};
    std::exception_ptr error_;
    std::thread th_;
};

// status of a C-ABI call made on a worker thread -> exception carrying that thread's message
void ck(rv_status s) {
    if (s != RV_OK) throw Error(s, rv_last_error());
}

double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// dst bits [pos, pos + n) |= src bits [0, n) (LSB-first, bitmap.rs:61-68); dst zero-initialised there
void or_bits(uint8_t *dst, uint64_t pos, const uint8_t *src, uint64_t n) {
    if (n == 0) return;
    const unsigned sh = static_cast<unsigned>(pos & 7);
    uint8_t *d = dst + (pos >> 3);
    const uint64_t nbytes = (n + 7) / 8;
    const uint8_t tail = (n & 7) ? static_cast<uint8_t>((1u << (n & 7)) - 1) : 0xFF;
    if (sh == 0) {
        if (nbytes > 1) std::memcpy(d, src, nbytes - 1);  // whole bytes owned by this range (its first byte is byte-aligned)
        d[nbytes - 1] |= static_cast<uint8_t>(src[nbytes - 1] & tail);
        return;
    }
    for (uint64_t i = 0; i < nbytes; ++i) {
        const uint8_t b = i + 1 == nbytes ? static_cast<uint8_t>(src[i] & tail) : src[i];
        d[i] |= static_cast<uint8_t>(b << sh);
        const uint8_t hi = static_cast<uint8_t>(b >> (8 - sh));
        if (hi) d[i + 1] |= hi;  // never past the buffer: hi != 0 only if bits of the range live there
    }
}
// dst bits [pos, pos + n) = 1
void set_bits(uint8_t *dst, uint64_t pos, uint64_t n) {
    for (; n && (pos & 7); ++pos, --n) dst[pos >> 3] |= static_cast<uint8_t>(1u << (pos & 7));
    if (n >= 8) std::memset(dst + (pos >> 3), 0xFF, n / 8);
    pos += n & ~uint64_t(7);
    for (n &= 7; n; ++pos, --n) dst[pos >> 3] |= static_cast<uint8_t>(1u << (pos & 7));
}

}  // namespace

// Pinned host blocks of a group's gathers.  Outlives the group through the blocks still handed out: after close() a
// returned block is freed at once instead of being kept for reuse.
struct PinnedPool {
    std::mutex mu;
    std::multimap<size_t, void *> free_list;
    bool closed = false;

    void *get(size_t bytes, size_t *got) {
        bytes = std::max<size_t>((bytes + 4095) & ~size_t(4095), 4096);
        {
            std::lock_guard<std::mutex> g(mu);
            auto it = free_list.lower_bound(bytes);
            if (it != free_list.end() && it->first <= bytes + bytes / 2 + (1u << 20)) {
                void *p = it->second;
                *got = it->first;
                free_list.erase(it);
                return p;
            }
        }
        void *p = nullptr;
        if (hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) {
            (void)hipGetLastError();
            release();
            if (hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) {
                (void)hipGetLastError();
                throw Error(RV_ERR_OOM, fmt("cannot pin %zu bytes of host memory for the gather", bytes));
            }
        }
        *got = bytes;
        return p;
    }
    void give_back(void *p, size_t bytes) {
        std::lock_guard<std::mutex> g(mu);
        if (closed) (void)hipHostFree(p);
        else free_list.emplace(bytes, p);
    }
    void release() {
        std::lock_guard<std::mutex> g(mu);
        for (auto &kv : free_list) (void)hipHostFree(kv.second);
        free_list.clear();
    }
    void close() {
        {
            std::lock_guard<std::mutex> g(mu);
            closed = true;
        }
        release();
    }
    ~PinnedPool() { release(); }
};

struct rv_group {
    std::vector<int> devices;
    std::vector<rv_ctx *> ctx;
    std::vector<std::unique_ptr<Worker>> workers;
    bool distinct = true;  // no device listed twice: RCCL can form the communicator
    // RCCL, made on the first aggregate (ncclCommInitAll); both vectors are committed together, complete or not at all
    std::vector<void *> comms;
    std::vector<void *> d_red;  // per rank: 2 x int64 + 1 x double on the device
    void *h_red = nullptr;      // pinned: 32 bytes per rank, the host side of the all-reduce payload
    // counters behind rv_group_stat
    int64_t rccl_ranks = 0, allreduce_calls = 0, last_agg_filter_us = 0, last_allreduce_us = 0, comm_aborts = 0;
    // pinned host blocks, reused from query to query (pinning gigabytes costs far more than filtering them).  Shared with
    // the results that hold blocks: an rv_gather may be freed after its group (a garbage-collected binding does that)
    std::shared_ptr<PinnedPool> pinned_pool = std::make_shared<PinnedPool>();
    // set when a failed collective could not be drained from the contexts' streams (drop_comms): work that targets the group's
    // reduction buffers may still be queued, so the buffers are leaked and every later call is refused
    bool broken = false;
    // run f(rank) on every worker at once; the first failure is rethrown after ALL have finished
    template <class F>
    void parallel(F f) {
        if (broken) throw rvh::Error(RV_ERR_DEVICE, "this group is unusable: a failed collective could not be drained from its streams (create a new group)");
        for (size_t r = 0; r < workers.size(); ++r) workers[r]->post([f, r] { f(static_cast<uint32_t>(r)); });
        std::exception_ptr first;
        for (auto &w : workers) {
            try {
                w->wait();
            } catch (...) {
                if (!first) first = std::current_exception();
            }
        }
        if (first) std::rethrow_exception(first);
    }
};

namespace {
struct PinnedBlock {
    std::shared_ptr<PinnedPool> pool;
    void *ptr = nullptr;
    size_t bytes = 0;
    PinnedBlock() = default;
    PinnedBlock(rv_group *grp, size_t want) : pool(grp->pinned_pool) { ptr = pool->get(want, &bytes); }
    PinnedBlock(PinnedBlock &&o) noexcept : pool(std::move(o.pool)), ptr(o.ptr), bytes(o.bytes) { o.ptr = nullptr; }
    PinnedBlock &operator=(PinnedBlock &&o) noexcept {
        reset();
        pool = std::move(o.pool), ptr = o.ptr, bytes = o.bytes;
        o.ptr = nullptr;
        return *this;
    }
    PinnedBlock(const PinnedBlock &) = delete;
    PinnedBlock &operator=(const PinnedBlock &) = delete;
    ~PinnedBlock() { reset(); }
    void reset() {
        if (ptr) pool->give_back(ptr, bytes);
        ptr = nullptr;
    }
};
}  // namespace

namespace {
// the calling thread's current HIP device, restored on scope exit (the group's calls visit every device)
struct DeviceGuard {
    int saved = -1;
    DeviceGuard() {
        if (hipGetDevice(&saved) != hipSuccess) {
            (void)hipGetLastError();
            saved = -1;
        }
    }
    ~DeviceGuard() {
        if (saved >= 0) (void)hipSetDevice(saved);
    }
};

// Communicators and their device buffers leave the group together.  abort: a collective may still be queued on them
// (a rank failed or timed out) -- ncclCommAbort tears that down, ncclCommDestroy would wait for it.
void drop_comms(rv_group *g, bool abort) noexcept {
    std::vector<void *> comms, d_red;
    comms.swap(g->comms);
    d_red.swap(g->d_red);
    if (comms.empty() && d_red.empty() && !g->h_red) return;
    DeviceGuard guard;
    Rccl *r = nullptr;
    try {
        r = &rccl();
    } catch (...) {  // the library went away: nothing to call
    }
    const bool grouped = abort && r && r->GroupStart && r->GroupEnd && comms.size() > 1;  // abort all ranks as one operation where possible
    if (grouped) (void)r->GroupStart();
    for (size_t i = 0; i < comms.size(); ++i) {
        if (!comms[i] || !r) continue;
        (void)hipSetDevice(g->devices[i]);
        if (abort && r->CommAbort) (void)r->CommAbort(comms[i]);
        else (void)r->CommDestroy(comms[i]);
    }
    if (grouped) (void)r->GroupEnd();
    // After an abort the copies into h_red (and possibly the collective's kernels) may still be queued on the contexts' own
    // streams: wait for them, bounded, BEFORE the buffers go -- hipFree synchronises the device, so freeing under a kernel the
    // abort did not end would turn the bounded wait of the caller into an unbounded one, and freeing under a queued copy would
    // let it write into released pinned memory.  Streams that do not drain in time: the buffers are leaked, the group is marked
    // unusable.
    bool drained = true;
    if (abort) {
        const double t0 = now_ms();
        for (size_t i = 0; i < g->ctx.size() && i < g->devices.size(); ++i) {
            if (!g->ctx[i]) continue;
            (void)hipSetDevice(g->devices[i]);
            hipStream_t s = static_cast<hipStream_t>(rv_ctx_stream(g->ctx[i]));
            for (;;) {
                const hipError_t q = hipStreamQuery(s);
                if (q != hipErrorNotReady) break;  // drained, or an error that a later call on the context reports
                if (now_ms() - t0 > 2000.0) {
                    drained = false;
                    break;
                }
                std::this_thread::sleep_for(std::chrono::microseconds(100));
            }
        }
    }
    if (drained) {
        for (size_t i = 0; i < d_red.size(); ++i) {
            if (!d_red[i]) continue;
            (void)hipSetDevice(g->devices[i]);
            (void)hipFree(d_red[i]);
        }
        if (g->h_red) (void)hipHostFree(g->h_red);
    } else {
        g->broken = true;
        for (rv_ctx *c : g->ctx)
            if (c) c->undrained = true;  // rv_ctx_destroy bounds its wait too (rv_group_destroy)
    }
    g->h_red = nullptr;
    g->rccl_ranks = 0;
    if (abort) g->comm_aborts += 1;
    (void)hipGetLastError();
}

// ncclCommInitAll: one communicator per device, this process owns them all.  Built in locals and committed to the
// group only when every step succeeded; the caller's current device is left as it was.
void ensure_comms(rv_group *g) {
    if (!g->comms.empty()) return;
    const uint32_t n = static_cast<uint32_t>(g->devices.size());
    DeviceGuard guard;
    std::vector<void *> comms(n, nullptr), d_red(n, nullptr);
    void *h_red = nullptr;
    Rccl &r = rccl();
    rccl_check(r.CommInitAll(comms.data(), static_cast<int>(n), g->devices.data()), "ncclCommInitAll");
    try {
        for (uint32_t i = 0; i < n; ++i) {
            RV_HIP(hipSetDevice(g->devices[i]));
            RV_HIP(hipMalloc(&d_red[i], 32));
        }
        RV_HIP(hipHostMalloc(&h_red, static_cast<size_t>(n) * 32, hipHostMallocPortable));
        int ranks = static_cast<int>(n);
        if (r.CommCount) rccl_check(r.CommCount(comms[0], &ranks), "ncclCommCount");
        require(ranks == static_cast<int>(n), RV_ERR_DEVICE, fmt("RCCL formed a communicator of %d ranks for %u devices", ranks, n));
        g->rccl_ranks = ranks;
    } catch (...) {
        for (uint32_t i = 0; i < n; ++i) {
            (void)hipSetDevice(g->devices[i]);
            if (comms[i]) (void)r.CommDestroy(comms[i]);
            if (d_red[i]) (void)hipFree(d_red[i]);
        }
        if (h_red) (void)hipHostFree(h_red);
        throw;
    }
    g->comms = std::move(comms);
    g->d_red = std::move(d_red);
    g->h_red = h_red;
}

struct AggPartialHost {
    int64_t si = 0;
    double sf = 0.0;
    uint64_t cnt = 0;
};

// The final scalar of BASELINE configs[4]: ncclAllReduce(count = 2, ncclInt64, ncclSum) over xGMI (+ 1 x ncclFloat64
// for a Float64 SUM), every rank's result read back and compared.  Issued for ALL ranks by the calling thread inside
// ONE ncclGroupStart / ncclGroupEnd -- the documented way for one thread to drive several devices -- and only after
// every rank's partial exists: no rank can be left alone inside the collective by another one's failure.  The wait is
// bounded (RV_GROUP_TIMEOUT_MS, default 120 s: the first collective also builds the transport); on any failure the
// communicators are aborted and re-made by the next call.
void allreduce_partials(rv_group *g, std::vector<AggPartialHost> &part) {
    const uint32_t n = static_cast<uint32_t>(g->devices.size());
    ensure_comms(g);
    DeviceGuard guard;
    Rccl &r = rccl();
    int64_t *h = static_cast<int64_t *>(g->h_red);
    std::vector<hipStream_t> streams(n);
    try {
        for (uint32_t i = 0; i < n; ++i) {
            streams[i] = static_cast<hipStream_t>(rv_ctx_stream(g->ctx[i]));
            int64_t *hi = h + 4 * i;
            hi[0] = part[i].si;
            hi[1] = static_cast<int64_t>(part[i].cnt);
            std::memcpy(&hi[2], &part[i].sf, 8);
            hi[3] = 0;
            RV_HIP(hipSetDevice(g->devices[i]));
            RV_HIP(hipMemcpyAsync(g->d_red[i], hi, 32, hipMemcpyHostToDevice, streams[i]));
        }
        rccl_check(r.GroupStart(), "ncclGroupStart");
        int rc = 0;
        for (uint32_t i = 0; i < n && rc == 0; ++i) {
            char *d = static_cast<char *>(g->d_red[i]);
            rc = r.AllReduce(d, d, 2, kNcclInt64, kNcclSum, g->comms[i], streams[i]);
            if (rc == 0) rc = r.AllReduce(d + 16, d + 16, 1, kNcclFloat64, kNcclSum, g->comms[i], streams[i]);
        }
        const int rc_end = r.GroupEnd();  // always closed, also after a failed enqueue
        rccl_check(rc, "ncclAllReduce");
        rccl_check(rc_end, "ncclGroupEnd");
        g->allreduce_calls += 1;
        for (uint32_t i = 0; i < n; ++i) {
            RV_HIP(hipSetDevice(g->devices[i]));
            RV_HIP(hipMemcpyAsync(h + 4 * i, g->d_red[i], 32, hipMemcpyDeviceToHost, streams[i]));
        }
        static const long timeout_ms = [] {
            const char *e = getenv("RV_GROUP_TIMEOUT_MS");
            const long v = e ? std::strtol(e, nullptr, 10) : 0;
            return v > 0 ? v : 120000L;
        }();
        const double t0 = now_ms();
        for (uint32_t i = 0; i < n; ++i) {
            for (unsigned spins = 0;; ++spins) {
                const hipError_t q = hipStreamQuery(streams[i]);
                if (q == hipSuccess) break;
                if (q != hipErrorNotReady) RV_HIP(q);
                require(now_ms() - t0 < static_cast<double>(timeout_ms), RV_ERR_DEVICE,
                        fmt("the all-reduce did not complete on rank %u within %ld ms (RV_GROUP_TIMEOUT_MS)", i, timeout_ms));
                if (spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
            }
        }
    } catch (...) {
        drop_comms(g, true);
        throw;
    }
    for (uint32_t i = 0; i < n; ++i) {
        const int64_t *hi = h + 4 * i;
        part[i].si = hi[0];
        part[i].cnt = static_cast<uint64_t>(hi[1]);
        std::memcpy(&part[i].sf, &hi[2], 8);
    }
}
}  // namespace

struct rv_gather {
    uint64_t rows = 0;
    struct Col {
        rv_dtype dtype = RV_NULL;
        PinnedBlock values, validity, offsets;
        bool has_validity = false;
        uint64_t data_bytes = 0;
        int64_t null_count = 0;
    };
    std::vector<Col> cols;
    std::vector<uint64_t> rank_rows;
    double filter_ms = 0.0, gather_ms = 0.0;
};

extern "C" {

rv_status rv_group_create(const int *devices, uint32_t n, rv_group **out) {
    return guarded([&] {
        require(devices && out && n >= 1 && n <= 64, RV_ERR_INVALID_ARG, "rv_group_create: 1..64 devices");
        auto g = std::make_unique<rv_group>();
        g->devices.assign(devices, devices + n);
        for (uint32_t i = 0; i < n; ++i)
            for (uint32_t j = 0; j < i; ++j) g->distinct = g->distinct && devices[i] != devices[j];
        try {
            for (uint32_t r = 0; r < n; ++r) {
                rv_ctx *c = nullptr;
                ck(rv_ctx_create(devices[r], &c));
                g->ctx.push_back(c);
            }
            for (uint32_t r = 0; r < n; ++r) g->workers.push_back(std::make_unique<Worker>());
        } catch (...) {
            g->workers.clear();
            for (auto *c : g->ctx) rv_ctx_destroy(c);
            throw;
        }
        *out = g.release();
    });
}

rv_status rv_group_destroy(rv_group *group) {
    return guarded([&] {
        if (!group) return;
        DeviceGuard guard;
        group->workers.clear();  // joins
        drop_comms(group, false);  // never throws: the contexts below are released whatever RCCL says
        for (auto *c : group->ctx) rv_ctx_destroy(c);
        group->pinned_pool->close();  // blocks still held by live rv_gather results are freed when those go
        delete group;
    });
}

uint32_t rv_group_size(const rv_group *group) { return group ? static_cast<uint32_t>(group->ctx.size()) : 0; }
rv_ctx *rv_group_ctx(rv_group *group, uint32_t rank) { return (group && rank < group->ctx.size()) ? group->ctx[rank] : nullptr; }

rv_status rv_group_generate(rv_group *group, const rv_synth_spec *spec, rv_dcolumn **shards) {
    return guarded([&] {
        require(group && spec && shards, RV_ERR_INVALID_ARG, "rv_group_generate: NULL argument");
        const uint32_t n = rv_group_size(group);
        for (uint32_t r = 0; r < n; ++r) shards[r] = nullptr;
        try {
            group->parallel([&](uint32_t r) {
                uint64_t b = 0, e = 0;
                ck(rv_shard_range(spec->length, n, r, &b, &e));
                rv_synth_spec s = *spec;
                s.first_row = spec->first_row + b;  // global row index: the shards agree with the unsharded column
                s.length = e - b;
                if (!s.table_rows) s.table_rows = spec->first_row + spec->length;  // sorted patterns: the whole table's rows, whatever the shard
                ck(rv_generate(group->ctx[r], &s, &shards[r]));
                ck(rv_ctx_synchronize(group->ctx[r]));
            });
        } catch (...) {
            for (uint32_t r = 0; r < n; ++r) {
                if (shards[r]) rv_free(group->ctx[r], shards[r]);
                shards[r] = nullptr;
            }
            throw;
        }
    });
}

rv_status rv_group_upload(rv_group *group, const rv_column *host, rv_dcolumn **shards) {
    return guarded([&] {
        require(group && host && shards, RV_ERR_INVALID_ARG, "rv_group_upload: NULL argument");
        const uint32_t n = rv_group_size(group);
        for (uint32_t r = 0; r < n; ++r) shards[r] = nullptr;
        try {
            group->parallel([&](uint32_t r) {
                uint64_t b = 0, e = 0;
                ck(rv_shard_range(host->length, n, r, &b, &e));
                // a view of rows [b, e) whose buffers start at a byte boundary of the bitmaps, so that rv_upload
                // copies this rank's range only (it copies elements [0, offset + length) of what it is given)
                const uint64_t first = host->offset + b, back = first & 7, lead = first - back;
                rv_column v = *host;
                v.offset = back;
                v.length = e - b;
                std::vector<int32_t> rebased;
                if (host->validity) v.validity = host->validity + lead / 8;
                if (host->dtype == RV_INT64 || host->dtype == RV_FLOAT64) {
                    if (host->values) v.values = static_cast<const uint64_t *>(host->values) + lead;
                } else if (host->dtype == RV_BOOLEAN) {
                    if (host->values) v.values = static_cast<const uint8_t *>(host->values) + lead / 8;
                } else if (host->dtype == RV_STRING) {
                    require(host->offsets != nullptr, RV_ERR_INVALID_ARG, "rv_group_upload: offsets is NULL");
                    const int32_t *o = host->offsets + lead;
                    const uint64_t cnt = back + (e - b);
                    const int32_t b0 = o[0], b1 = o[cnt];
                    require(b0 >= 0 && b1 >= b0 && static_cast<uint64_t>(b1) <= host->data_bytes, RV_ERR_INVALID_ARG, "Offset out of bounds");
                    rebased.resize(cnt + 1);
                    for (uint64_t i = 0; i <= cnt; ++i) rebased[i] = o[i] - b0;
                    v.offsets = rebased.data();
                    v.values = static_cast<const uint8_t *>(host->values) + b0;
                    v.data_bytes = static_cast<uint64_t>(b1 - b0);
                }
                ck(rv_upload(group->ctx[r], &v, &shards[r]));
            });
        } catch (...) {
            for (uint32_t r = 0; r < n; ++r) {
                if (shards[r]) rv_free(group->ctx[r], shards[r]);
                shards[r] = nullptr;
            }
            throw;
        }
    });
}

rv_status rv_group_free(rv_group *group, rv_dcolumn **shards) {
    return guarded([&] {
        require(group && shards, RV_ERR_INVALID_ARG, "rv_group_free: NULL argument");
        for (uint32_t r = 0; r < rv_group_size(group); ++r) {
            if (shards[r]) rv_free(group->ctx[r], shards[r]);
            shards[r] = nullptr;
        }
    });
}

}  // extern "C"

namespace {
// phase 1 of BASELINE configs[3]: every device filters + compacts its shard, all devices at once; the outputs stay
// in HBM (outs[r * nproj + j]).  A rank that fails frees nothing of the others': the caller drops all of it.
void group_filter_resident(rv_group *group, const rv_dcolumn *const *shards, uint32_t ncols, const rv_predicate *pred, const uint32_t *proj,
                           uint32_t nproj, rv_dcolumn **outs, uint64_t *rank_rows, double *filter_ms) {
    const uint32_t n = rv_group_size(group);
    std::vector<double> ms(n, 0.0);
    group->parallel([&](uint32_t r) {
        const double a = now_ms();
        std::vector<rv_dcolumn *> none(1, nullptr);
        ck(rv_filter_project(group->ctx[r], shards + static_cast<size_t>(r) * ncols, ncols, pred, proj, nproj,
                             nproj ? outs + static_cast<size_t>(r) * nproj : none.data(), &rank_rows[r], nullptr));
        ms[r] = now_ms() - a;
    });
    if (filter_ms) *filter_ms = *std::max_element(ms.begin(), ms.end());
}

// phase 2: prefix sum of the N survivor counts, one pinned buffer per output column, every device copies its output
// into its slice (rank order == row order).  No collective: a rank that fails is reported after all have finished.
void group_gather(rv_group *group, const rv_dcolumn *const *outs, uint32_t nproj, rv_gather *res) {
    const uint32_t n = rv_group_size(group);
    const double t1 = now_ms();
    std::vector<std::vector<rv_column_info>> infos(n, std::vector<rv_column_info>(nproj));
    res->rank_rows.assign(n, 0);
    res->cols.resize(nproj);
    for (size_t i = 0; i < static_cast<size_t>(n) * nproj; ++i)
        require(outs[i] != nullptr, RV_ERR_INVALID_ARG, fmt("rv_group_gather: output %zu of rank %zu is NULL", i % nproj, i / nproj));
    group->parallel([&](uint32_t r) {  // String columns: two 4-byte reads on the rank's stream
        for (uint32_t j = 0; j < nproj; ++j) ck(rv_column_info_get(group->ctx[r], outs[static_cast<size_t>(r) * nproj + j], &infos[r][j]));
    });
    for (uint32_t r = 0; r < n; ++r)
        for (uint32_t j = 0; j < nproj; ++j) {
            if (j == 0) res->rank_rows[r] = infos[r][j].length;
            require(infos[r][j].length == res->rank_rows[r], RV_ERR_LENGTH_MISMATCH, "rv_group_gather: output columns of one rank differ in length");
            require(infos[r][j].dtype == infos[0][j].dtype, RV_ERR_TYPE_MISMATCH, "All batches must have the same schema");  // record_batch.rs:252-254
        }
    std::vector<uint64_t> prefix(n + 1, 0);
    for (uint32_t r = 0; r < n; ++r) prefix[r + 1] = prefix[r] + res->rank_rows[r];
    const uint64_t total = prefix[n];
    res->rows = total;
    std::vector<std::vector<uint64_t>> byte_prefix(nproj, std::vector<uint64_t>(n + 1, 0));
    for (uint32_t j = 0; j < nproj; ++j) {
        rv_gather::Col &c = res->cols[j];
        c.dtype = infos[0][j].dtype;
        for (uint32_t r = 0; r < n; ++r) {
            c.has_validity = c.has_validity || infos[r][j].has_validity != 0;
            byte_prefix[j][r + 1] = byte_prefix[j][r] + infos[r][j].data_bytes;
        }
        const size_t bits = static_cast<size_t>((total + 63) / 64) * 8 + 8;
        switch (c.dtype) {
            case RV_INT64:
            case RV_FLOAT64: c.values = PinnedBlock(group, std::max<size_t>(total * 8, 8)); break;
            case RV_BOOLEAN:
                c.values = PinnedBlock(group, bits);
                std::memset(c.values.ptr, 0, bits);
                break;
            case RV_STRING:
                c.data_bytes = byte_prefix[j][n];
                require(c.data_bytes <= 0x7FFFFFFFull, RV_ERR_UNSUPPORTED, "StringArray data larger than 2 GiB (int32 offsets, string.rs:11)");
                c.values = PinnedBlock(group, std::max<size_t>(c.data_bytes, 8));
                c.offsets = PinnedBlock(group, (total + 1) * 4);
                break;
            default: break;  // NullArray: a length
        }
        if (c.has_validity) {
            c.validity = PinnedBlock(group, bits);
            std::memset(c.validity.ptr, 0, bits);
        }
        if (c.dtype == RV_NULL) c.null_count = static_cast<int64_t>(total);
    }

    // 8-byte values and String bytes land in place; bit buffers go through a per-rank pinned block and are merged
    // below (neighbouring ranks share bytes at arbitrary bit offsets)
    struct BitPart {
        PinnedBlock values, validity;
        int has_validity = 0;
    };
    std::vector<std::vector<BitPart>> parts(n);
    std::vector<std::vector<int64_t>> nulls(n, std::vector<int64_t>(nproj, 0));
    for (uint32_t r = 0; r < n; ++r) parts[r].resize(nproj);
    group->parallel([&](uint32_t r) {
        const uint64_t rows = res->rank_rows[r];
        for (uint32_t j = 0; j < nproj; ++j) {
            rv_gather::Col &c = res->cols[j];
            const rv_dcolumn *o = outs[static_cast<size_t>(r) * nproj + j];
            BitPart &bp = parts[r][j];
            const size_t bit_bytes = static_cast<size_t>((rows + 7) / 8) + 8;
            if (infos[r][j].has_validity) bp.validity = PinnedBlock(group, bit_bytes);
            uint8_t *vtmp = static_cast<uint8_t *>(bp.validity.ptr);
            if (c.dtype == RV_INT64 || c.dtype == RV_FLOAT64) {
                ck(rv_download(group->ctx[r], o, rows ? static_cast<uint64_t *>(c.values.ptr) + prefix[r] : nullptr, vtmp, &bp.has_validity));
            } else if (c.dtype == RV_BOOLEAN) {
                bp.values = PinnedBlock(group, bit_bytes);
                ck(rv_download(group->ctx[r], o, bp.values.ptr, vtmp, &bp.has_validity));
            } else if (c.dtype == RV_STRING) {
                // offsets of the rank's elements arrive rebased to 0 in place, then move to the rank's byte range
                int32_t *offs = static_cast<int32_t *>(c.offsets.ptr) + prefix[r];
                PinnedBlock tmp(group, (rows + 1) * 4);
                ck(rv_download_string(group->ctx[r], o, static_cast<int32_t *>(tmp.ptr), static_cast<uint8_t *>(c.values.ptr) + byte_prefix[j][r], vtmp,
                                      &bp.has_validity));
                const int32_t base = static_cast<int32_t>(byte_prefix[j][r]);
                const int32_t *src = static_cast<const int32_t *>(tmp.ptr);
                for (uint64_t i = 0; i < rows; ++i) offs[i] = src[i] + base;  // element i starts here; the end is the next start
            }
            if (infos[r][j].has_validity) {
                uint64_t nc = 0;
                ck(rv_null_count(group->ctx[r], o, &nc));
                nulls[r][j] = static_cast<int64_t>(nc);
            }
        }
    });
    for (uint32_t j = 0; j < nproj; ++j) {
        rv_gather::Col &c = res->cols[j];
        if (c.dtype == RV_STRING) static_cast<int32_t *>(c.offsets.ptr)[total] = static_cast<int32_t>(c.data_bytes);
        for (uint32_t r = 0; r < n; ++r) {
            const uint64_t rows = res->rank_rows[r];
            if (c.dtype == RV_BOOLEAN) or_bits(static_cast<uint8_t *>(c.values.ptr), prefix[r], static_cast<const uint8_t *>(parts[r][j].values.ptr), rows);
            if (c.has_validity) {
                // concat_arrays re-appends every element (record_batch.rs:277-342): a part without a bitmap is all valid
                if (infos[r][j].has_validity) or_bits(static_cast<uint8_t *>(c.validity.ptr), prefix[r], static_cast<const uint8_t *>(parts[r][j].validity.ptr), rows);
                else set_bits(static_cast<uint8_t *>(c.validity.ptr), prefix[r], rows);
                c.null_count += nulls[r][j];
            }
        }
    }
    res->gather_ms = now_ms() - t1;
}

void check_group_query(rv_group *group, const rv_dcolumn *const *shards, uint32_t ncols, const rv_predicate *pred, const char *who) {
    require(group && shards && pred && pred->terms, RV_ERR_INVALID_ARG, fmt("%s: NULL argument", who));
    require(ncols >= 1, RV_ERR_INVALID_ARG, fmt("%s: no columns", who));
    const uint32_t n = rv_group_size(group);
    for (uint32_t i = 0; i < n * ncols; ++i) require(shards[i] != nullptr, RV_ERR_INVALID_ARG, fmt("shard %u is NULL", i));
}
}  // namespace

extern "C" {

rv_status rv_group_filter_project_resident(rv_group *group, const rv_dcolumn *const *shards, uint32_t ncols, const rv_predicate *pred,
                                           const uint32_t *proj, uint32_t nproj, rv_dcolumn **outs, uint64_t *rank_rows, uint64_t *out_rows) {
    return guarded([&] {
        check_group_query(group, shards, ncols, pred, "rv_group_filter_project_resident");
        require((proj && outs) || nproj == 0, RV_ERR_INVALID_ARG, "rv_group_filter_project_resident: NULL argument");
        const uint32_t n = rv_group_size(group);
        for (size_t i = 0; i < static_cast<size_t>(n) * nproj; ++i) outs[i] = nullptr;
        std::vector<uint64_t> rows(n, 0);
        try {
            group_filter_resident(group, shards, ncols, pred, proj, nproj, outs, rows.data(), nullptr);
        } catch (...) {
            for (uint32_t r = 0; r < n; ++r)
                for (uint32_t j = 0; j < nproj; ++j) {
                    rv_dcolumn *&d = outs[static_cast<size_t>(r) * nproj + j];
                    if (d) rv_free(group->ctx[r], d);
                    d = nullptr;
                }
            throw;
        }
        uint64_t total = 0;
        for (uint32_t r = 0; r < n; ++r) {
            total += rows[r];
            if (rank_rows) rank_rows[r] = rows[r];
        }
        if (out_rows) *out_rows = total;
    });
}

rv_status rv_group_gather(rv_group *group, const rv_dcolumn *const *outs, uint32_t nproj, rv_gather **out) {
    return guarded([&] {
        require(group && out && (outs || nproj == 0), RV_ERR_INVALID_ARG, "rv_group_gather: NULL argument");
        auto res = std::make_unique<rv_gather>();
        group_gather(group, outs, nproj, res.get());
        *out = res.release();
    });
}

rv_status rv_group_filter_project(rv_group *group, const rv_dcolumn *const *shards, uint32_t ncols, const rv_predicate *pred,
                                  const uint32_t *proj, uint32_t nproj, rv_gather **out, uint64_t *out_rows) {
    return guarded([&] {
        check_group_query(group, shards, ncols, pred, "rv_group_filter_project");
        require(out && (proj || nproj == 0), RV_ERR_INVALID_ARG, "rv_group_filter_project: NULL argument");
        const uint32_t n = rv_group_size(group);
        auto res = std::make_unique<rv_gather>();
        std::vector<rv_dcolumn *> outs(static_cast<size_t>(n) * std::max<uint32_t>(nproj, 1), nullptr);
        std::vector<uint64_t> rows(n, 0);
        auto drop_outs = [&] {
            for (uint32_t r = 0; r < n; ++r)
                for (uint32_t j = 0; j < nproj; ++j) {
                    rv_dcolumn *&d = outs[static_cast<size_t>(r) * nproj + j];
                    if (d) rv_free(group->ctx[r], d);
                    d = nullptr;
                }
        };
        try {
            double filter_ms = 0.0;
            group_filter_resident(group, shards, ncols, pred, proj, nproj, outs.data(), rows.data(), &filter_ms);
            group_gather(group, outs.data(), nproj, res.get());
            res->filter_ms = filter_ms;
            if (nproj == 0) {  // no column to read the counts from
                res->rank_rows = rows;
                res->rows = 0;
                for (uint64_t v : rows) res->rows += v;
            }
            drop_outs();
        } catch (...) {
            drop_outs();
            throw;
        }
        if (out_rows) *out_rows = res->rows;
        *out = res.release();
    });
}

// StreamingPhysicalPlan::collect() over a HOST table on N devices (streaming.rs:71-133: dataframe_to_batches :135-233 -> the pull loop ->
// collect_stream_batches -> concat, :343-352): the table is cut into N row ranges (rv_shard_range: 64-row boundaries), every range is
// streamed through ITS device's own double-buffered chunk pipeline (rv_filter_project_host: upload of chunk k + 1 on a second stream
// while chunk k is filtered) -- all devices at once, each over its own PCIe link, where one context is held to one link's 55 GB/s --
// and the survivors are gathered in rank order == row order.  No data-path collective.
rv_status rv_group_filter_project_host(rv_group *group, const rv_column *host_cols, uint32_t ncols, const rv_predicate *pred, const uint32_t *proj,
                                       uint32_t nproj, uint64_t chunk_rows, rv_gather **out, uint64_t *out_rows, double *rank_upload_gbs) {
    return guarded([&] {
        require(group && host_cols && pred && pred->terms && out && (proj || nproj == 0), RV_ERR_INVALID_ARG, "rv_group_filter_project_host: NULL argument");
        require(ncols >= 1, RV_ERR_INVALID_ARG, "rv_group_filter_project_host: no columns");
        const uint32_t n = rv_group_size(group);
        const uint64_t rows_all = host_cols[0].length;
        for (uint32_t c = 0; c < ncols; ++c) require(host_cols[c].length == rows_all, RV_ERR_LENGTH_MISMATCH, "All columns must have the same length");  // record_batch.rs:31-38
        auto res = std::make_unique<rv_gather>();
        std::vector<rv_dcolumn *> outs(static_cast<size_t>(n) * std::max<uint32_t>(nproj, 1), nullptr);
        std::vector<uint64_t> rows(n, 0);
        std::vector<double> ms(n, 0.0), bytes(n, 0.0);
        auto drop_outs = [&] {
            for (uint32_t r = 0; r < n; ++r)
                for (uint32_t j = 0; j < nproj; ++j) {
                    rv_dcolumn *&d = outs[static_cast<size_t>(r) * nproj + j];
                    if (d) rv_free(group->ctx[r], d);
                    d = nullptr;
                }
        };
        try {
            group->parallel([&](uint32_t r) {
                uint64_t b = 0, e = 0;
                ck(rv_shard_range(rows_all, n, r, &b, &e));
                // the rank's rows as views of the caller's arrays: the element offset moves, nothing is copied on the host
                std::vector<rv_column> mine(host_cols, host_cols + ncols);
                for (rv_column &c : mine) {
                    c.offset += b;
                    c.length = e - b;
                    if (c.dtype == RV_STRING && c.offsets) bytes[r] += static_cast<double>(c.offsets[c.offset + c.length] - c.offsets[c.offset]) + 4.0 * static_cast<double>(c.length);
                    else bytes[r] += (c.dtype == RV_BOOLEAN ? 0.125 : 8.0) * static_cast<double>(c.length);
                    if (c.validity) bytes[r] += 0.125 * static_cast<double>(c.length);
                }
                std::vector<rv_dcolumn *> none(1, nullptr);
                const double a = now_ms();
                ck(rv_filter_project_host(group->ctx[r], mine.data(), ncols, pred, proj, nproj, chunk_rows, nproj ? outs.data() + static_cast<size_t>(r) * nproj : none.data(), &rows[r]));
                ck(rv_ctx_synchronize(group->ctx[r]));
                ms[r] = now_ms() - a;
            });
            group_gather(group, outs.data(), nproj, res.get());
            res->filter_ms = *std::max_element(ms.begin(), ms.end());
            if (nproj == 0) {  // no column to read the counts from
                res->rank_rows = rows;
                res->rows = 0;
                for (uint64_t v : rows) res->rows += v;
            }
            drop_outs();
        } catch (...) {
            drop_outs();
            throw;
        }
        if (rank_upload_gbs)
            for (uint32_t r = 0; r < n; ++r) rank_upload_gbs[r] = ms[r] > 0.0 ? bytes[r] / (ms[r] * 1e-3) / 1e9 : 0.0;
        if (out_rows) *out_rows = res->rows;
        *out = res.release();
    });
}

rv_status rv_gather_column(const rv_gather *result, uint32_t j, rv_column *view, int64_t *null_count) {
    return guarded([&] {
        require(result && view && j < result->cols.size(), RV_ERR_INVALID_ARG, "rv_gather_column: bad arguments");
        const rv_gather::Col &c = result->cols[j];
        *view = rv_column{};
        view->dtype = c.dtype;
        view->values = c.values.ptr;
        view->validity = c.has_validity ? static_cast<const uint8_t *>(c.validity.ptr) : nullptr;
        view->offset = 0;
        view->length = result->rows;
        view->offsets = static_cast<const int32_t *>(c.offsets.ptr);
        view->data_bytes = c.data_bytes;
        if (null_count) *null_count = c.null_count;
    });
}

rv_status rv_gather_stats(const rv_gather *result, uint64_t *rank_rows, double *filter_ms, double *gather_ms) {
    return guarded([&] {
        require(result != nullptr, RV_ERR_INVALID_ARG, "rv_gather_stats: result is NULL");
        if (rank_rows) std::copy(result->rank_rows.begin(), result->rank_rows.end(), rank_rows);
        if (filter_ms) *filter_ms = result->filter_ms;
        if (gather_ms) *gather_ms = result->gather_ms;
    });
}

rv_status rv_gather_free(rv_gather *result) {
    return guarded([&] { delete result; });
}

rv_status rv_group_filter_agg(rv_group *group, const rv_dcolumn *const *shards, uint32_t ncols, const rv_predicate *pred,
                              uint32_t agg_col, int64_t *sum_i, double *sum_f, uint64_t *count) {
    return guarded([&] {
        check_group_query(group, shards, ncols, pred, "rv_group_filter_agg");
        require(agg_col < ncols, RV_ERR_INVALID_ARG, "rv_group_filter_agg: bad column index");
        const uint32_t n = rv_group_size(group);
        std::vector<AggPartialHost> part(n);
        // ---- phase 1: every rank's partial, all devices at once.  parallel() returns when ALL ranks have finished and
        //      rethrows the first failure: a rank that failed keeps every rank out of the collective ----------------------
        const double t0 = now_ms();
        group->parallel([&](uint32_t r) {
            AggPartialHost &p = part[r];
            ck(rv_filter_agg(group->ctx[r], shards + static_cast<size_t>(r) * ncols, ncols, pred, agg_col, &p.si, &p.sf, &p.cnt));
        });
        const double t1 = now_ms();
        group->last_agg_filter_us = static_cast<int64_t>((t1 - t0) * 1e3);
        // ---- phase 2: the final scalar ---------------------------------------------------------------------------------
        AggPartialHost total;
        if (group->distinct) {
            allreduce_partials(group, part);
            total = part[0];
            // every rank must hold the same 16 bytes; the Float64 sum is rank 0's (RCCL reduces once and hands the result round, so
            // the ranks agree bit for bit in practice, but only the integers are part of the contract)
            for (uint32_t r = 1; r < n; ++r)
                require(part[r].si == total.si && part[r].cnt == total.cnt, RV_ERR_INTERNAL, "all-reduce left different values on different ranks");
        } else {
            // a device listed twice: RCCL refuses such a communicator; the partials are summed here, rank order
            for (uint32_t r = 0; r < n; ++r) {
                total.si = static_cast<int64_t>(static_cast<uint64_t>(total.si) + static_cast<uint64_t>(part[r].si));
                total.sf += part[r].sf;
                total.cnt += part[r].cnt;
            }
        }
        group->last_allreduce_us = static_cast<int64_t>((now_ms() - t1) * 1e3);
        if (sum_i) *sum_i = total.si;
        if (sum_f) *sum_f = total.sf;
        if (count) *count = total.cnt;
    });
}

rv_status rv_group_stat(rv_group *group, const char *key, int64_t *value) {
    return guarded([&] {
        require(group && key && value, RV_ERR_INVALID_ARG, "rv_group_stat: NULL argument");
        const std::string k(key);
        if (k == "rccl_ranks") *value = group->rccl_ranks;
        else if (k == "allreduce_calls") *value = group->allreduce_calls;
        else if (k == "comm_aborts") *value = group->comm_aborts;
        else if (k == "distinct_devices") *value = group->distinct ? 1 : 0;
        else if (k == "last_agg_filter_us") *value = group->last_agg_filter_us;
        else if (k == "last_allreduce_us") *value = group->last_allreduce_us;
        else throw Error(RV_ERR_INVALID_ARG, "unknown group statistic '" + k + "'");
    });
}

rv_status rv_host_register(rv_ctx *ctx, void *ptr, size_t bytes) {
    return guarded([&] {
        require(ctx && ptr && bytes, RV_ERR_INVALID_ARG, "rv_host_register: NULL argument");
        RV_HIP(hipSetDevice(ctx->device));
        if (hipHostRegister(ptr, bytes, hipHostRegisterPortable) != hipSuccess) {
            (void)hipGetLastError();
            throw Error(RV_ERR_OOM, fmt("rv_host_register: cannot pin %zu bytes", bytes));
        }
    });
}
rv_status rv_host_unregister(rv_ctx *ctx, void *ptr) {
    return guarded([&] {
        require(ctx && ptr, RV_ERR_INVALID_ARG, "rv_host_unregister: NULL argument");
        RV_HIP(hipHostUnregister(ptr));
    });
}

}  // extern "C"
