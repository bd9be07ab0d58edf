// Context, options, kernel timers, and the helpers every launch uses: control block, striped counters, term lowering.
// One unit of the backend library behind include/rivulus_gpu.h (gfx950 only; compiled with hipcc).  Shared helpers and the
// functions the units call across each other are declared in launch.hpp (namespace rvl).
#include "launch.hpp"

using namespace rvh;
using namespace rvl;

namespace {
thread_local std::string g_last_error_text;
}  // namespace
std::string &rvh::last_error() { return g_last_error_text; }

namespace rvl {

size_t elem_bytes(rv_dtype t, uint64_t n) {
    switch (t) {
        case RV_INT64:
        case RV_FLOAT64: return static_cast<size_t>(n) * 8;
        case RV_BOOLEAN: return static_cast<size_t>((n + 63) / 64) * 8;  // whole words
        default: return 0;
    }
}
size_t bitmap_words_bytes(uint64_t n) { return static_cast<size_t>((n + 63) / 64) * 8; }

DevBufRef pool_alloc(rv_ctx *ctx, size_t bytes) {
    auto b = std::make_shared<DevBuf>();
    size_t got = 0;
    b->ptr = ctx->pool->alloc(bytes, &got);
    b->bytes = got;
    b->pool = ctx->pool;
    static std::atomic<uint64_t> next_id{1};
    b->id = next_id.fetch_add(1, std::memory_order_relaxed);
    return b;
}

void set_device(rv_ctx *ctx) { RV_HIP(hipSetDevice(ctx->device)); }
// option "inject_failure": a query entry point fails before it launches anything (rv_group_* failure handling)
void maybe_injected_failure(rv_ctx *ctx) {
    if (ctx->opt_inject_failure > 0) {
        ctx->opt_inject_failure -= 1;
        throw Error(RV_ERR_DEVICE, "injected failure (option inject_failure)");
    }
}

// control block + `ntiles` look-back descriptors, zeroed on the stream
// layout: [Ctrl | look-back descriptors ntiles x 8 B | redo list ntiles x 16 B]; the first two are zeroed
Ctrl *prepare_ctrl(rv_ctx *ctx, size_t ntiles) {
    const size_t zeroed = kCtrlBytes + ntiles * 8;
    const size_t need = zeroed + ntiles * 16;
    if (need > ctx->ctrl_bytes) {
        if (ctx->d_ctrl) {
            RV_HIP(hipStreamSynchronize(ctx->stream));
            RV_HIP(hipFree(ctx->d_ctrl));
            ctx->d_ctrl = nullptr;
            ctx->ctrl_bytes = 0;
        }
        const size_t cap = std::max(need + need / 2, static_cast<size_t>(1) << 20);
        RV_HIP(hipMalloc(&ctx->d_ctrl, cap));
        ctx->ctrl_bytes = cap;
    }
    RV_HIP(hipMemsetAsync(ctx->d_ctrl, 0, (zeroed + 15) & ~size_t(15), ctx->stream));
    if (ctx->stripe_mask) {  // a query that failed between its kernels and fetch_ctrl left stripes behind
        RV_HIP(hipMemsetAsync(ctx->d_stripes, 0, kStripeBytes, ctx->stream));
        ctx->stripe_mask = 0;
    }
    return static_cast<Ctrl *>(ctx->d_ctrl);
}
// the stripes of a counter of ctx->d_ctrl (one of its first kStripeSlots words): what a kernel whose waves all add to that
// counter is handed instead of the word itself.  fetch_ctrl folds them into the word.
unsigned long long *striped(rv_ctx *ctx, const unsigned long long *ctrl_word) {
    const size_t slot = static_cast<size_t>(reinterpret_cast<const char *>(ctrl_word) - static_cast<const char *>(ctx->d_ctrl)) / 8;
    require(slot < static_cast<size_t>(rvk::kStripeSlots), RV_ERR_INTERNAL, "striped counter outside the head of the control block");
    ctx->stripe_mask |= 1u << slot;
    return static_cast<unsigned long long *>(ctx->d_stripes) + slot * rvk::kStripeSlotWords;
}
const Ctrl *fetch_ctrl(rv_ctx *ctx) {
    if (ctx->stripe_mask) {
        hipLaunchKernelGGL(rvk::fold_stripes_kernel, dim3(1), dim3(64), 0, ctx->stream, static_cast<unsigned long long *>(ctx->d_stripes),
                           static_cast<unsigned long long *>(ctx->d_ctrl), ctx->stripe_mask);
        RV_HIP(hipGetLastError());
        ctx->stripe_mask = 0;
    }
    RV_HIP(hipMemcpyAsync(ctx->h_ctrl, ctx->d_ctrl, kCtrlBytes, hipMemcpyDeviceToHost, ctx->stream));
    RV_HIP(hipStreamSynchronize(ctx->stream));
    return static_cast<const Ctrl *>(ctx->h_ctrl);
}
// Control block of ONE fused launch (several may be in flight: rv_filter_project_begin): same layout as above,
// own device memory, own pinned mirror, own event.  Zeroed on the stream.
// `nranges`: wave ranges of the launch (tiles x waves) for the redo list behind the descriptors, zeroed with them; 0: none
rv_ctx::LaunchCtrl acquire_launch_ctrl(rv_ctx *ctx, size_t ntiles, size_t nranges) {
    const size_t zeroed = kCtrlBytes + (ntiles + nranges) * 8, need = zeroed;
    rv_ctx::LaunchCtrl c;
    // a block an earlier pass has zeroed for this use (LaunchCtrl::clean): no memset between two passes
    for (size_t i = 0; i < ctx->ctrl_free.size(); ++i)
        if (ctx->ctrl_free[i].bytes >= need && ctx->ctrl_free[i].clean >= zeroed) {
            c = ctx->ctrl_free[i];
            ctx->ctrl_free.erase(ctx->ctrl_free.begin() + static_cast<long>(i));
            c.clean = 0;
            c.dirty = zeroed;
            return c;
        }
    for (size_t i = 0; i < ctx->ctrl_free.size(); ++i)
        if (ctx->ctrl_free[i].bytes >= need) {
            c = ctx->ctrl_free[i];
            ctx->ctrl_free.erase(ctx->ctrl_free.begin() + static_cast<long>(i));
            break;
        }
    if (!c.dev) {
        if (!ctx->ctrl_free.empty()) {  // recycle the host side of a block that is too small
            c = ctx->ctrl_free.back();
            ctx->ctrl_free.pop_back();
            RV_HIP(hipStreamSynchronize(ctx->stream));
            RV_HIP(hipFree(c.dev));
            c.dev = nullptr;
        } else {
            RV_HIP(hipHostMalloc(&c.host, kCtrlBytes, hipHostMallocDefault));
            RV_HIP(hipEventCreateWithFlags(&c.ev, hipEventDisableTiming));
        }
        c.bytes = std::max(need + need / 2, static_cast<size_t>(1) << 16);
        RV_HIP(hipMalloc(&c.dev, c.bytes));
    }
    RV_HIP(hipMemsetAsync(c.dev, 0, (zeroed + 15) & ~size_t(15), ctx->stream));
    c.clean = 0;
    c.dirty = zeroed;
    return c;
}
void release_launch_ctrl(rv_ctx *ctx, const rv_ctx::LaunchCtrl &c) { ctx->ctrl_free.push_back(c); }
// A free block for the pass about to be launched to zero (its idle waves do it while it runs): one whose last use dirtied `dirty` bytes
// and nobody has cleaned since.  None free: a spare of `like`'s size is made once, so that one launch at a time alternates between two
// blocks.  Returns the index into ctx->ctrl_free, or -1; the caller marks it clean once the launch is queued.
int block_to_zero(rv_ctx *ctx, const rv_ctx::LaunchCtrl &like) {
    constexpr size_t kMost = size_t(64) << 20;
    for (size_t i = 0; i < ctx->ctrl_free.size(); ++i)
        if (ctx->ctrl_free[i].clean == 0 && ctx->ctrl_free[i].dirty > 0 && ctx->ctrl_free[i].dirty <= kMost) return static_cast<int>(i);
    bool any_clean = false;
    for (auto &f : ctx->ctrl_free) any_clean = any_clean || f.clean > 0;
    if (any_clean || ctx->ctrl_free.size() >= 3 || like.bytes > kMost) return -1;
    rv_ctx::LaunchCtrl c;
    if (hipHostMalloc(&c.host, kCtrlBytes, hipHostMallocDefault) != hipSuccess || hipEventCreateWithFlags(&c.ev, hipEventDisableTiming) != hipSuccess ||
        hipMalloc(&c.dev, like.bytes) != hipSuccess) {
        (void)hipGetLastError();
        if (c.host) (void)hipHostFree(c.host);
        if (c.ev) (void)hipEventDestroy(c.ev);
        return -1;
    }
    c.bytes = like.bytes;
    c.dirty = like.dirty;  // (never used: everything of it that the next user needs zero is zeroed by this launch)
    ctx->ctrl_free.push_back(c);
    return static_cast<int>(ctx->ctrl_free.size()) - 1;
}

rvk::DevCol dev_view(const rv_dcolumn *c) {
    rvk::DevCol d{};
    d.values = c->values ? c->values->ptr : nullptr;
    d.validity = c->validity ? static_cast<const uint8_t *>(c->validity->ptr) : nullptr;
    d.offset = c->offset;
    d.values_bytes = c->values ? c->values->bytes : 0;
    d.validity_bytes = c->validity ? c->validity->bytes : 0;
    d.dtype = static_cast<int32_t>(c->dtype);
    return d;
}

bool is_value_type(rv_dtype t) { return t == RV_INT64 || t == RV_FLOAT64; }

// StringArray::validate_utf8's offsets walk (string.rs:126-147) for a host array about to be copied to the
// device: entries [first, first + count] must start >= 0, never decrease and end inside the data buffer --
// the device kernels read data + offsets[i] unchecked.  (UTF-8 validity itself is not re-checked: the bytes are
// only ever moved and compared, never decoded.)
void check_string_offsets(const int32_t *offsets, uint64_t first, uint64_t count, uint64_t data_bytes) {
    const int32_t *o = offsets + first;
    bool ok = o[0] >= 0;
    for (uint64_t i = 0; i < count && ok; ++i) ok = o[i + 1] >= o[i];
    require(ok && static_cast<uint64_t>(o[count]) <= data_bytes, RV_ERR_INVALID_ARG, "Offset out of bounds");  // string.rs:137-139
}

// `Column <op> Literal` -> device term.  Folds the AnyValue truth table of the reference
// (series.rs:87-117 as used by plan.rs:112-130) for null cells, null literals and
// cross-type compares into {code, const_v, null_v}.
rvk::DevTerm lower_term(const rv_term &t, rv_dtype col_type, rv_null_policy policy, uint32_t slot) {
    rvk::DevTerm d{};
    require(t.op >= RV_EQ && t.op <= RV_IS_TRUE, RV_ERR_INVALID_ARG, "unknown compare operator");
    const bool is_bool = col_type == RV_BOOLEAN;
    if (t.op == RV_IS_TRUE) {
        require(col_type == RV_BOOLEAN, RV_ERR_TYPE_MISMATCH, "Predicate must be a BooleanArray");
        d.set(slot, rvk::TC_BOOL, t.op, true, false, false);
        return d;
    }
    const bool lit_null = t.lit_type == RV_NULL;
    const bool least = policy == RV_NULL_IS_LEAST;
    bool null_v, const_v = false;
    int code;
    if (lit_null) null_v = least && (t.op == RV_EQ || t.op == RV_LE || t.op == RV_GE);
    else null_v = least && (t.op == RV_LT || t.op == RV_LE || t.op == RV_NE);
    if (lit_null) {
        code = rvk::TC_CONST;
        const_v = (t.op == RV_GT || t.op == RV_GE || t.op == RV_NE);  // any value > Null
    } else if (t.lit_type != col_type) {
        code = rvk::TC_CONST;
        const_v = (t.op == RV_NE);  // cross-type partial_cmp == None
    } else if (col_type == RV_INT64) {
        code = rvk::TC_I64 + t.op;
        d.lit = t.lit.i;
    } else if (col_type == RV_FLOAT64) {
        code = rvk::TC_F64 + t.op;
        std::memcpy(&d.lit, &t.lit.f, 8);
    } else {
        code = rvk::TC_BOOL;
        d.lit = t.lit.i != 0;
    }
    d.set(slot, code, t.op, is_bool, const_v, null_v);
    return d;
}

int grid_for_words(rv_ctx *ctx, uint64_t items, int block) {
    const uint64_t want = (items + block - 1) / block;
    const uint64_t cap = static_cast<uint64_t>(ctx->props.multiProcessorCount) * 8;
    return static_cast<int>(std::max<uint64_t>(1, std::min(want, cap)));
}

}  // namespace rvl

extern "C" {

uint32_t rv_abi_version(void) { return RV_ABI_VERSION; }
const char *rv_last_error(void) { return last_error().c_str(); }
const char *rv_status_name(rv_status s) {
    switch (s) {
        case RV_OK: return "RV_OK";
        case RV_ERR_INVALID_ARG: return "RV_ERR_INVALID_ARG";
        case RV_ERR_LENGTH_MISMATCH: return "RV_ERR_LENGTH_MISMATCH";
        case RV_ERR_TYPE_MISMATCH: return "RV_ERR_TYPE_MISMATCH";
        case RV_ERR_OUT_OF_BOUNDS: return "RV_ERR_OUT_OF_BOUNDS";
        case RV_ERR_UNSUPPORTED: return "RV_ERR_UNSUPPORTED";
        case RV_ERR_DEVICE: return "RV_ERR_DEVICE";
        case RV_ERR_OOM: return "RV_ERR_OOM";
        case RV_ERR_INTERNAL: return "RV_ERR_INTERNAL";
    }
    return "RV_ERR_?";
}

int rv_device_count(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return count;
}

rv_status rv_ctx_create(int device, rv_ctx **out) {
    return guarded([&] {
        require(out != nullptr, RV_ERR_INVALID_ARG, "rv_ctx_create: out is NULL");
        int count = 0;
        hipError_t e = hipGetDeviceCount(&count);
        if (e != hipSuccess || count == 0) {
            (void)hipGetLastError();
            throw Error(RV_ERR_DEVICE, "no HIP device available: the MI355X backend has no CPU fallback");
        }
        require(device >= 0 && device < count, RV_ERR_INVALID_ARG, fmt("device %d out of range (%d present)", device, count));
        auto ctx = std::make_unique<rv_ctx>();
        ctx->device = device;
        RV_HIP(hipSetDevice(device));
        RV_HIP(hipGetDeviceProperties(&ctx->props, device));
        require(std::string(ctx->props.gcnArchName).rfind("gfx950", 0) == 0, RV_ERR_DEVICE,
                fmt("device %d is %s; this library carries gfx950 (MI355X) code objects only", device, ctx->props.gcnArchName));
        RV_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        RV_HIP(hipEventCreate(&ctx->ev0));
        RV_HIP(hipEventCreate(&ctx->ev1));
        RV_HIP(hipEventCreate(&ctx->evk0));
        RV_HIP(hipEventCreate(&ctx->evk1));
        RV_HIP(hipHostMalloc(&ctx->h_ctrl, kCtrlBytes, hipHostMallocDefault));
        RV_HIP(hipMalloc(&ctx->d_stripes, kStripeBytes));
        RV_HIP(hipMemset(ctx->d_stripes, 0, kStripeBytes));
        ctx->pool = std::make_shared<Pool>(device);
        // diagnostics: RV_OPTIONS="vec=1,rows_per_lane=4112" presets rv_ctx_set_option keys for tools that cannot call it (bench.py under rocprofv3)
        if (const char *env = getenv("RV_OPTIONS")) {
            std::string all(env);
            size_t at = 0;
            while (at < all.size()) {
                const size_t comma = std::min(all.find(',', at), all.size()), eq = all.find('=', at);
                if (eq != std::string::npos && eq < comma) {
                    const rv_status st = rv_ctx_set_option(ctx.get(), all.substr(at, eq - at).c_str(), std::strtoll(all.c_str() + eq + 1, nullptr, 0));
                    if (st != RV_OK) throw Error(st, "RV_OPTIONS: " + last_error());
                }
                at = comma + 1;
            }
        }
        *out = ctx.release();
    });
}

rv_status rv_ctx_destroy(rv_ctx *ctx) {
    return guarded([&] {
        if (!ctx) return;
        set_device(ctx);
        if (ctx->undrained) {
            // work a failed collective left on the stream: wait for it a bounded time; if it does not end, nothing of the context is
            // freed (hipFree and hipStreamDestroy would wait for it without a bound) -- leaked, like the group's reduction buffers
            const auto t0 = std::chrono::steady_clock::now();
            while (hipStreamQuery(ctx->stream) == hipErrorNotReady) {
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
                    (void)hipGetLastError();
                    return;
                }
                std::this_thread::sleep_for(std::chrono::microseconds(200));
            }
            (void)hipGetLastError();
        } else {
            (void)hipStreamSynchronize(ctx->stream);
        }
        if (ctx->d_ctrl) (void)hipFree(ctx->d_ctrl);
        if (ctx->d_stripes) (void)hipFree(ctx->d_stripes);
        if (ctx->d_sample) (void)hipFree(ctx->d_sample);
        if (ctx->h_sample) (void)hipHostFree(const_cast<unsigned long long *>(ctx->h_sample));
        if (ctx->h_ctrl) (void)hipHostFree(ctx->h_ctrl);
        if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
        (void)hipEventDestroy(ctx->ev0);
        (void)hipEventDestroy(ctx->ev1);
        (void)hipEventDestroy(ctx->evk0);
        (void)hipEventDestroy(ctx->evk1);
        for (auto &c : ctx->ctrl_free) {
            (void)hipFree(c.dev);
            (void)hipHostFree(c.host);
            (void)hipEventDestroy(c.ev);
            if (c.tk0) (void)hipEventDestroy(c.tk0);
            if (c.tk1) (void)hipEventDestroy(c.tk1);
        }
        (void)hipStreamDestroy(ctx->stream);
        if (ctx->copy_stream) {
            (void)hipStreamDestroy(ctx->copy_stream);
            (void)hipEventDestroy(ctx->ev_up[0]);
            (void)hipEventDestroy(ctx->ev_up[1]);
            (void)hipEventDestroy(ctx->ev_main);
        }
        ctx->pool->release_all();
        delete ctx;
    });
}

rv_status rv_ctx_synchronize(rv_ctx *ctx) {
    return guarded([&] {
        require(ctx != nullptr, RV_ERR_INVALID_ARG, "ctx is NULL");
        RV_HIP(hipStreamSynchronize(ctx->stream));
    });
}

void *rv_ctx_stream(rv_ctx *ctx) { return ctx ? static_cast<void *>(ctx->stream) : nullptr; }

rv_status rv_ctx_device_info(rv_ctx *ctx, int *compute_units, uint64_t *hbm_bytes, char *name, size_t name_len) {
    return guarded([&] {
        require(ctx != nullptr, RV_ERR_INVALID_ARG, "ctx is NULL");
        if (compute_units) *compute_units = ctx->props.multiProcessorCount;
        if (hbm_bytes) *hbm_bytes = ctx->props.totalGlobalMem;
        if (name && name_len) snprintf(name, name_len, "%s (%s)", ctx->props.name, ctx->props.gcnArchName);
    });
}

rv_status rv_ctx_set_option(rv_ctx *ctx, const char *key, int64_t value) {
    return guarded([&] {
        require(ctx && key, RV_ERR_INVALID_ARG, "ctx/key is NULL");
        const std::string k(key);
        if (k == "profile_kernels") ctx->opt_profile = value;
        else if (k == "rows_per_lane") ctx->opt_rows_per_lane = value;
        else if (k == "vec") ctx->opt_vec = value;
        else if (k == "cap_rows") ctx->opt_cap_rows = value;
        else if (k == "wgs_per_cu") ctx->opt_wgs_per_cu = value;
        else if (k == "stamp") ctx->opt_stamp = value;
        else if (k == "debug") ctx->opt_debug = value;
        else if (k == "depth") ctx->opt_depth = value;
        else if (k == "roomy") ctx->opt_roomy = value;
        else if (k == "direct") ctx->opt_direct = value;
        else if (k == "direct_r") ctx->opt_direct_r = value;
        else if (k == "sample") ctx->opt_sample = value;
        else if (k == "skew") ctx->opt_skew = value;
        else if (k == "segments") ctx->opt_segments = value;
        else if (k == "str_tiles_from") ctx->opt_str_tiles_from = value;
        else if (k == "groups_by_ranges") ctx->opt_groups_by_ranges = value;
        else if (k == "bool_cap") ctx->opt_bool_cap = value;
        else if (k == "speculative_batches") ctx->opt_speculative_batches = value;
        else if (k == "direct_waves") ctx->opt_direct_waves = value;
        else if (k == "spin_limit") ctx->opt_spin_limit = value;
        else if (k == "agg_grid") ctx->opt_agg_grid = value;
        else if (k == "bools_in_pass") ctx->opt_bools_in_pass = value;
        else if (k == "inject_failure") ctx->opt_inject_failure = value;
        else if (k == "out_sizing") {
            require(value >= -1 && value <= 1000000, RV_ERR_INVALID_ARG, "out_sizing: -1, 0, 1 or a bound in rows per million");
            ctx->opt_out_sizing = value;
        }
        else throw Error(RV_ERR_INVALID_ARG, "unknown option '" + k + "'");
    });
}

rv_status rv_ctx_get_option(rv_ctx *ctx, const char *key, int64_t *value) {
    return guarded([&] {
        require(ctx && key && value, RV_ERR_INVALID_ARG, "ctx/key/value is NULL");
        const std::string k(key);
        if (k == "profile_kernels") *value = ctx->opt_profile;
        else if (k == "rows_per_lane") *value = ctx->opt_rows_per_lane;
        else if (k == "vec") *value = ctx->opt_vec;
        else if (k == "cap_rows") *value = ctx->opt_cap_rows;
        else if (k == "wgs_per_cu") *value = ctx->opt_wgs_per_cu;
        else if (k == "depth") *value = ctx->opt_depth;
        else if (k == "roomy") *value = ctx->opt_roomy;
        else if (k == "direct") *value = ctx->opt_direct;
        else if (k == "direct_r") *value = ctx->opt_direct_r;
        else if (k == "sample") *value = ctx->opt_sample;
        else if (k == "skew") *value = ctx->opt_skew;
        else if (k == "segments") *value = ctx->opt_segments;
        else if (k == "segmented_passes") *value = static_cast<int64_t>(ctx->segmented_passes);
        else if (k == "segment_fallbacks") *value = static_cast<int64_t>(ctx->segment_fallbacks);
        else if (k == "str_tiles_from") *value = ctx->opt_str_tiles_from;
        else if (k == "speculative_batches") *value = ctx->opt_speculative_batches;
        else if (k == "speculative_batch_passes") *value = static_cast<int64_t>(ctx->speculative_batch_passes);
        else if (k == "samples_taken") *value = static_cast<int64_t>(ctx->samples_taken);
        else if (k == "last_rows_out") *value = static_cast<int64_t>(ctx->last_rows_out);  // survivors of the last fused pass (read-only)
        else if (k == "last_rows_in") *value = static_cast<int64_t>(ctx->last_rows_in);
        else if (k == "hbm_free_bytes") {  // what the device reports free right now (pooled blocks of this context count as used)
            size_t fr = 0, tot = 0;
            set_device(ctx);
            RV_HIP(hipMemGetInfo(&fr, &tot));
            *value = static_cast<int64_t>(fr);
        }
        else if (k == "direct_waves") *value = ctx->opt_direct_waves;
        else if (k == "spin_limit") *value = ctx->opt_spin_limit;
        else if (k == "agg_grid") *value = ctx->opt_agg_grid;
        else if (k == "bools_in_pass") *value = ctx->opt_bools_in_pass;
        else if (k == "inject_failure") *value = ctx->opt_inject_failure;
        else if (k == "out_sizing") *value = ctx->opt_out_sizing;
        else if (k == "overflow_reruns") *value = static_cast<int64_t>(ctx->overflow_reruns);  // read-only counter
        else if (k == "batch_counts_in_pass") *value = static_cast<int64_t>(ctx->batch_counts_in_pass);  // read-only counter
        else if (k == "fused_rows_scanned") *value = static_cast<int64_t>(ctx->fused_rows_scanned);  // read-only counter
        else if (k == "last_redo_ppm") *value = static_cast<int64_t>(ctx->last_redo_fraction * 1e6);  // wave ranges per million left to the redo kernel
        else if (k == "last_selectivity_ppm") *value = ctx->last_selectivity < 0 ? -1 : static_cast<int64_t>(ctx->last_selectivity * 1e6);
        else throw Error(RV_ERR_INVALID_ARG, "unknown option '" + k + "'");
    });
}

rv_status rv_ctx_kernel_stats(rv_ctx *ctx, double *total_ms, uint64_t *launches, int reset) {
    return guarded([&] {
        require(ctx != nullptr, RV_ERR_INVALID_ARG, "ctx is NULL");
        if (total_ms) *total_ms = ctx->kernel_ms;
        if (launches) *launches = ctx->kernel_launches;
        if (reset) {
            ctx->kernel_ms = 0.0;
            ctx->kernel_launches = 0;
        }
    });
}

rv_status rv_ctx_last_kernel(rv_ctx *ctx, char *name, size_t name_len) {
    return guarded([&] {
        require(ctx && name && name_len, RV_ERR_INVALID_ARG, "rv_ctx_last_kernel: NULL argument");
        snprintf(name, name_len, "%s", ctx->last_kernel.c_str());
    });
}

rv_status rv_timer_start(rv_ctx *ctx) {
    return guarded([&] {
        require(ctx != nullptr, RV_ERR_INVALID_ARG, "ctx is NULL");
        RV_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    });
}
rv_status rv_timer_stop(rv_ctx *ctx, float *elapsed_ms) {
    return guarded([&] {
        require(ctx && elapsed_ms, RV_ERR_INVALID_ARG, "ctx/elapsed_ms is NULL");
        RV_HIP(hipEventRecord(ctx->ev1, ctx->stream));
        RV_HIP(hipEventSynchronize(ctx->ev1));
        RV_HIP(hipEventElapsedTime(elapsed_ms, ctx->ev0, ctx->ev1));
    });
}

}  // extern "C"
