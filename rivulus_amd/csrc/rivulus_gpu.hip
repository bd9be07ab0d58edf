// C ABI of the MI355X backend (include/rivulus_gpu.h): context, arrays, and the
// launch logic of every kernel.  gfx950 only; compiled with hipcc.
//
// Sections, in file order (search for the `// ----` banners):
//   control block, striped counters, kernel variant selection        helpers every launch uses
//   fused launch                                                     fused_begin / fused_finish: shape -> instantiation, LDS and
//                                                                    output sizing, launch, read-back, redo and overflow re-run
//   rv_ctx_*                                                         context, options, kernel timers
//   StringArray on the device                                        scans, gather (take / selection form), concat
//   predicate normalisation                                          rv_predicate -> AND list / CNF literals / composed BooleanArray
//   arrays                                                           rv_upload .. rv_download
//   predicate, BooleanArray logic                                    rv_eval_predicate, rv_compare*, rv_boolean_*
//   RecordBatch kernels                                              filter_by_groups, filter_query, rv_filter_project*
//   many RecordBatches, one launch                                   batch_counts, rv_filter_project_batches / _chunked, rv_filter
//   take / concat                                                    rv_take*, rv_selection_indices, rv_concat
//   host-resident table                                              rv_host_*, rv_filter_project_host (second stream)
//   filter + aggregate, multi-GPU                                    rv_filter_agg, rv_shard_range, rv_comm_*   (rv_group_*: group.hip)
#include <dlfcn.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <functional>
#include <thread>

#include "aux_kernels.hpp"
#include "fused_table.hpp"
#include "string_kernels.hpp"
#include "runtime.hpp"
#include "rccl_loader.hpp"

using namespace rvh;

namespace {

thread_local std::string g_last_error;
}  // namespace
std::string &rvh::last_error() { return g_last_error; }
namespace {

constexpr size_t kCtrlBytes = 512;
constexpr size_t kStripeBytes = static_cast<size_t>(rvk::kStripeSlots) * rvk::kStripeSlotWords * 8;
struct Ctrl {  // mirrors the first kCtrlBytes of rv_ctx::d_ctrl
    uint32_t ticket;
    uint32_t err;
    unsigned long long out_count;
    unsigned long long valid_pop[8];
    unsigned long long pops[3];
    unsigned long long pad0[3];
    rvk::AggPartial agg;
    unsigned long long stamps[32];  // [0,8) wave 0, [8,16) wave 1 phase sums; [16,28) sub-phase marks, [28,32) scanner / fallback counts (FF_STAMP builds)
    uint32_t redo_count;
    uint32_t overflow;  // survivors did not fit the speculatively sized outputs
};
static_assert(sizeof(Ctrl) <= kCtrlBytes, "ctrl block");
static_assert(offsetof(Ctrl, agg) == 128, "ctrl layout");

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
rv_ctx::LaunchCtrl acquire_launch_ctrl(rv_ctx *ctx, size_t ntiles) {
    const size_t zeroed = kCtrlBytes + ntiles * 8, need = zeroed + ntiles * 16;
    rv_ctx::LaunchCtrl c;
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
    return c;
}
void release_launch_ctrl(rv_ctx *ctx, const rv_ctx::LaunchCtrl &c) { ctx->ctrl_free.push_back(c); }

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
rvk::DevTerm lower_term(const rv_term &t, rv_dtype col_type, rv_null_policy policy, uint32_t slot = 0) {
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

// ---------------------------------------------------------------------------------------
// fused launch
// ---------------------------------------------------------------------------------------
// Smallest instantiation whose feature flags cover `need`; for one-column lean/validity
// launches the geometry can be steered with rv_ctx_set_option("rows_per_lane", R | waves << 8).
const rvk::FusedEntry *find_fused(rv_ctx *ctx, int ncols, int vec, int need) {
    const rvk::FusedEntry *best = nullptr;
    auto scan = [&](const rvk::FusedEntry *t, size_t n) {
        for (size_t i = 0; i < n; ++i) {
            const rvk::FusedEntry &e = t[i];
            constexpr int kShape = rvk::FF_ONE_I64 | rvk::FF_ONE_F64 | rvk::FF_STAMP | rvk::FF_PROJALL | rvk::FF_NONULL | rvk::FF_EXPR;  // must match exactly
            // the generic FF_PROJALL instantiations write the selection bitmap on request (fused_kernel.hpp, kSel)
            const int has = e.flags | (((e.flags & rvk::FF_PROJALL) && !(e.flags & (rvk::FF_ONE_I64 | rvk::FF_ONE_F64))) ? rvk::FF_SEL : 0);
            if (e.ncols != ncols || (ncols > 0 && e.vec != vec) || (has & need) != need) continue;
            if ((e.flags & kShape) != (need & kShape)) continue;
            bool wanted = false;
            if (ctx->opt_rows_per_lane <= 0 && ctx->last_redo_fraction > 0.05 && ncols == 1)
                wanted = e.r == 8 && e.waves == 16;  // dense data last time: 512-row slots hold every row of a wave
            if (ctx->opt_rows_per_lane > 0) {
                const int want_r = static_cast<int>(ctx->opt_rows_per_lane & 0xFF);
                const int want_w = static_cast<int>((ctx->opt_rows_per_lane >> 8) & 0xFF);
                wanted = e.r == want_r && (want_w == 0 || e.waves == want_w);
            }
            if (!best || __builtin_popcount(e.flags) < __builtin_popcount(best->flags) ||
                (wanted && e.flags == best->flags))
                best = &e;
        }
    };
    size_t n = 0;
    const rvk::FusedEntry *t;
    // first match wins among equals, so list the preferred default geometry first in each table
    for (int pass = 0; pass < 2 && !best; ++pass) {
        t = rvk::fused_entries_lean1(&n), scan(t, n);
        t = rvk::fused_entries_valid1(&n), scan(t, n);
        t = rvk::fused_entries_multi(&n), scan(t, n);
        t = rvk::fused_entries_bool(&n), scan(t, n);
        t = rvk::fused_entries_full(&n), scan(t, n);
        t = rvk::fused_entries_expr(&n), scan(t, n);
        vec = 1;  // every feature set exists with 8-byte loads
    }
    return best;
}
// `prefer`: shape flags worth having when an instantiation exists (FF_PROJALL)
const rvk::FusedEntry &pick_fused(rv_ctx *ctx, int ncols, int vec, int need, int prefer = 0) {
    // the refinements the launch qualifies for, dropped one by one (FF_NONULL first) until an instantiation exists
    const rvk::FusedEntry *best = nullptr;
    for (const int pf : {prefer, prefer & ~rvk::FF_NONULL}) {
        if (best || !pf) continue;
        best = find_fused(ctx, ncols, vec, need | pf);
        if (best && (best->flags & ~(need | pf)) != 0) best = nullptr;  // not at the price of features the launch does not need
    }
    if (!best) best = find_fused(ctx, ncols, vec, need);
    require(best != nullptr, RV_ERR_INTERNAL, fmt("no fused kernel variant for %d columns, flags %d", ncols, need));
    return *best;
}

// A predicate with OR / NOT, lowered for the kernels: `terms` handed along with it is the literal list of a
// conjunctive normal form (a user term may appear several times); see normalize_predicate.
struct ExprInfo {
    std::vector<uint8_t> negate, group_end;  // per literal
    bool negate_result = false;              // the list is the CNF of NOT(expression)
    bool strict = false;                     // RV_NULL_DROPS: a null in any column below drops the row
    std::vector<uint32_t> strict_cols;       // batch column indices the expression reads
};

struct OutCol {
    rv_dcolumn *col = nullptr;
    int value_slot = -1;           // value column slot, or -1
    int xs_values = -1, xs_valid = -1;  // bit stream indices (Boolean columns)
};

// Per-batch survivor counts of a pass over many RecordBatches of equal length (seam S1: rv_filter_project_chunked /
// _batches), asked of the pass itself.  The pass that evaluates the predicate fills `counts` when a batch is a whole
// number of its wave ranges (FusedParams::wave_counts) and says so; otherwise the caller counts the selection bitmap.
struct BatchReq {
    uint64_t chunk_rows = 0, nb = 0;
    unsigned long long *counts = nullptr;  // device-visible: pinned host memory (the caller's array or the staging block)
    bool sel_optional = false;             // the selection bitmap is wanted only for counting: skip it when counted here
    bool counted = false;                  // out
};

// Output row of the first survivor of every wave range of the pass (FusedParams::wave_offsets), asked for by a caller
// that compacts bit-packed columns by the selection bitmap after the pass: `offsets` stays empty when the geometry's
// ranges do not tile a 4096-row step of the compaction kernel (12 rows per lane), or when nothing was launched.
struct RangeOffsets {
    DevBufRef offsets;
    uint32_t range_rows = 0;
    uint64_t out_capacity = 0;
};

// One single-pass launch in flight: everything fused_finish needs once the kernel has run.
struct FusedLaunch {
    rvk::FusedParams p{};
    std::vector<OutCol> outs;
    rv_ctx::LaunchCtrl ctrl;
    int need = 0, nvals = 0, nxs = 0;
    size_t stage_row_bytes = 0;
    uint64_t tile_rows = 0;
    bool launched = false;  // false: empty input, nothing to wait for
    bool timed = false;     // kernel events recorded (option profile_kernels)
    // for a re-run after an output overflow (speculative sizing)
    void (*fn)(const rvk::FusedParams) = nullptr;
    uint32_t grid = 0, block = 0;
    size_t lds = 0;
    uint64_t n = 0;
    std::vector<rv_dtype> out_dtypes;  // dtype of every projected source column
};
uint64_t fused_finish(rv_ctx *ctx, FusedLaunch &L);

// rows the outputs of a pass over n rows are sized for (option "out_sizing")
uint64_t output_capacity(rv_ctx *ctx, uint64_t n) {
    if (ctx->opt_out_sizing == 1 && ctx->last_selectivity >= 0.0)
        return std::min<uint64_t>(n, static_cast<uint64_t>(static_cast<double>(n) * (ctx->last_selectivity * 1.5 + 0.01)) + 1024);
    if (ctx->opt_out_sizing >= 2)
        return std::min<uint64_t>(n, static_cast<uint64_t>(static_cast<double>(n) * static_cast<double>(ctx->opt_out_sizing) * 1e-6) + 1024);
    return n;
}

// One single-pass launch: predicate over `cols`, compaction of the columns in proj; queued on the context's
// stream, not waited for.  out[] / sel_out receive the output handles at once (their length is set by
// fused_finish).
void fused_begin(rv_ctx *ctx, const rv_dcolumn *const *cols, uint32_t ncols, const rv_term *terms,
                 uint32_t nterms, rv_null_policy policy, const uint32_t *proj, uint32_t nproj,
                 rv_dcolumn **out, rv_dcolumn **sel_out, FusedLaunch &L, const ExprInfo *ex = nullptr, BatchReq *req = nullptr,
                 RangeOffsets *ranges = nullptr) {
    require(nterms >= 1 && nterms <= static_cast<uint32_t>(rvk::kMaxTerms), RV_ERR_UNSUPPORTED,
            fmt("predicate needs 1..%d terms, got %u", rvk::kMaxTerms, nterms));
    const uint64_t n = ncols ? cols[0]->length : 0;

    rvk::FusedParams &p = L.p;
    p = rvk::FusedParams{};
    p.in.n = n;
    p.in.nterms = static_cast<int32_t>(nterms);
    std::vector<int> value_slot(ncols, -1), bool_slot(ncols, -1);
    int nvals = 0, nbools = 0;
    auto slot_of_value = [&](uint32_t c) {
        if (value_slot[c] < 0) {
            require(nvals < rvk::kMaxValueCols, RV_ERR_UNSUPPORTED, "too many 8-byte columns for one pass");
            value_slot[c] = nvals;
            p.in.cols[nvals++] = dev_view(cols[c]);
        }
        return value_slot[c];
    };
    for (uint32_t t = 0; t < nterms; ++t) {
        const uint32_t c = terms[t].column;
        require(c < ncols, RV_ERR_INVALID_ARG, fmt("term %u references column %u of %u", t, c, ncols));
        const rv_dtype ct = cols[c]->dtype;
        require(is_value_type(ct) || ct == RV_BOOLEAN, RV_ERR_UNSUPPORTED,
                "predicate columns must be Int64, Float64 or Boolean on the device path");
        uint32_t slot;
        if (ct == RV_BOOLEAN) {
            if (bool_slot[c] < 0) {
                require(nbools < rvk::kMaxBoolCols, RV_ERR_UNSUPPORTED, "too many Boolean predicate columns for one pass");
                bool_slot[c] = nbools;
                p.in.bcols[nbools++] = dev_view(cols[c]);
            }
            slot = static_cast<uint32_t>(bool_slot[c]);
        } else {
            slot = static_cast<uint32_t>(slot_of_value(c));
        }
        p.in.terms[t] = lower_term(terms[t], ct, policy, slot);
        if (ex) p.in.terms[t].set_literal(ex->negate[t] != 0, ex->group_end[t] != 0);
    }

    // a nullable column tested by a term that drops its null rows has no null among the survivors: its output
    // needs no bitmap (and the builder would drop it anyway, primitive.rs:179-185)
    std::vector<char> never_null(ncols, 0);
    if (!ex) {
        for (uint32_t t = 0; t < nterms; ++t)
            if (is_value_type(cols[terms[t].column]->dtype) && !p.in.terms[t].null_v()) never_null[terms[t].column] = 1;
    } else {
        // OR / NOT: only strict propagation (RV_NULL_DROPS) guarantees it, and then for every column the expression reads
        p.in.expr_mode = 1;
        p.in.negate_result = ex->negate_result ? 1 : 0;
        if (ex->strict) {
            for (uint32_t c : ex->strict_cols) {
                require(c < ncols, RV_ERR_INTERNAL, "strict column out of range");
                if (is_value_type(cols[c]->dtype)) {
                    never_null[c] = 1;
                    if (cols[c]->validity) p.in.strict_values |= 1u << slot_of_value(c);
                } else if (cols[c]->dtype == RV_BOOLEAN && cols[c]->validity) {
                    if (bool_slot[c] < 0) {  // its literals were simplified away: still read for its nulls
                        require(nbools < rvk::kMaxBoolCols, RV_ERR_UNSUPPORTED, "too many Boolean predicate columns for one pass");
                        bool_slot[c] = nbools;
                        p.in.bcols[nbools++] = dev_view(cols[c]);
                    }
                    p.in.strict_bools |= 1u << bool_slot[c];
                }
            }
        }
    }

    // Output capacity.  Default: every row may survive (no second pass, 2x the input in HBM).  Option "out_sizing":
    // 1 = the context's last observed selectivity x 1.5 + 1 % (a stream of similar batches), k >= 2 = a caller-given bound
    // of k rows per million.  A launch that overflows its outputs still counts exactly; fused_finish then re-runs it
    // with buffers of the exact size (record_batch.rs:131-178 never over-allocates either: the builders grow).
    const uint64_t cap_out = output_capacity(ctx, n);
    p.out_capacity = cap_out;
    ctx->fused_rows_scanned += n;
    L.n = n;
    L.out_dtypes.clear();
    // outputs
    std::vector<OutCol> &outs = L.outs;
    outs.assign(nproj, OutCol{});
    size_t stage_row_bytes = 0;
    int nxs = 0;
    for (uint32_t j = 0; j < nproj; ++j) {
        const uint32_t c = proj[j];
        require(c < ncols, RV_ERR_INVALID_ARG, fmt("projection %u references column %u of %u", j, c, ncols));
        const rv_dcolumn *src = cols[c];
        auto *o = new rv_dcolumn();
        outs[j].col = o;
        out[j] = o;
        o->dtype = src->dtype;
        if (is_value_type(src->dtype)) {
            // the same source column projected twice shares nothing: give it its own slot view
            int slot = value_slot[c];
            if (slot >= 0 && p.out_values[slot]) {  // already projected once: duplicate slot
                require(nvals < rvk::kMaxValueCols, RV_ERR_UNSUPPORTED, "too many 8-byte columns for one pass");
                slot = nvals;
                p.in.cols[nvals++] = dev_view(src);
            } else {
                slot = slot_of_value(c);
            }
            outs[j].value_slot = slot;
            o->values = pool_alloc(ctx, std::max<size_t>(elem_bytes(src->dtype, cap_out), 8));
            p.out_values[slot] = static_cast<uint64_t *>(o->values->ptr);
            stage_row_bytes += 8;
            if (src->validity && !never_null[c]) {
                o->validity = pool_alloc(ctx, std::max<size_t>(bitmap_words_bytes(cap_out) + 8, 16));
                RV_HIP(hipMemsetAsync(o->validity->ptr, 0, std::max<size_t>(bitmap_words_bytes(cap_out) + 8, 16), ctx->stream));
                p.out_validity[slot] = static_cast<uint64_t *>(o->validity->ptr);
                stage_row_bytes += 1;
            }
        } else if (src->dtype == RV_BOOLEAN) {
            require(nxs + (src->validity ? 2 : 1) <= rvk::kMaxBitStreams, RV_ERR_UNSUPPORTED,
                    "too many Boolean columns for one pass");
            const size_t wb = std::max<size_t>(bitmap_words_bytes(cap_out) + 8, 16);
            o->values = pool_alloc(ctx, wb);
            RV_HIP(hipMemsetAsync(o->values->ptr, 0, wb, ctx->stream));
            rvk::BitStream bs{};
            bs.src = static_cast<const uint8_t *>(src->values->ptr);
            bs.src_bytes = src->values->bytes;
            bs.mask = src->validity ? static_cast<const uint8_t *>(src->validity->ptr) : nullptr;
            bs.mask_bytes = src->validity ? src->validity->bytes : 0;
            bs.offset = src->offset;
            bs.out = static_cast<uint64_t *>(o->values->ptr);
            outs[j].xs_values = nxs;
            p.xs[nxs++] = bs;
            stage_row_bytes += 1;
            if (src->validity) {
                o->validity = pool_alloc(ctx, wb);
                RV_HIP(hipMemsetAsync(o->validity->ptr, 0, wb, ctx->stream));
                rvk::BitStream vs{};
                vs.src = static_cast<const uint8_t *>(src->validity->ptr);
                vs.src_bytes = src->validity->bytes;
                vs.offset = src->offset;
                vs.out = static_cast<uint64_t *>(o->validity->ptr);
                outs[j].xs_valid = nxs;
                p.xs[nxs++] = vs;
                stage_row_bytes += 1;
            }
        } else {
            throw Error(RV_ERR_UNSUPPORTED, "only Int64, Float64 and Boolean columns are compacted on the device path");
        }
    }
    p.nxs = nxs;

    rv_dcolumn *sel = nullptr;
    auto make_selection = [&] {
        sel = new rv_dcolumn();
        *sel_out = sel;
        sel->dtype = RV_BOOLEAN;
        sel->length = n;
        sel->null_count = 0;
        sel->values = pool_alloc(ctx, std::max<size_t>(bitmap_words_bytes(n), 8));
        p.out_selection = static_cast<uint64_t *>(sel->values->ptr);
    };
    // a selection bitmap wanted only for per-batch counts is decided below, once the geometry is known
    const bool sel_deferred = sel_out && req && req->sel_optional && n > 0;
    if (sel_out && !sel_deferred) make_selection();
    if (sel_deferred) *sel_out = nullptr;
    if (n == 0) {
        for (auto &o : outs) {
            o.col->length = 0;
            o.col->null_count = 0;
            o.col->validity.reset();
        }
        L.launched = false;
        return;
    }

    // 16-byte loads need every loaded 8-byte column to start 16-byte aligned
    int vec = ctx->opt_vec == 1 ? 1 : (ctx->opt_vec == 2 ? 2 : (nvals <= 3 ? 2 : 1));
    for (int s = 0; s < nvals; ++s) {
        const uintptr_t a = reinterpret_cast<uintptr_t>(p.in.cols[s].values) + p.in.cols[s].offset * 8;
        if (a & 15) vec = 1;
    }
    int need = 0;
    for (int s = 0; s < nvals; ++s)
        if (p.in.cols[s].validity) need |= rvk::FF_VALIDITY;
    if (nbools) need |= rvk::FF_BOOL;
    // shapes that read bit buffers (null bitmaps, Boolean columns) or evaluate an expression run in lane form with 8-byte
    // loads: measured faster than 16-byte loads + per-slot mask arrays on every such shape (profiles/README.md)
    if (ctx->opt_vec == 0 && (need || ex)) vec = 1;
    if (nxs) need |= rvk::FF_XS;
    if (p.out_selection) need |= rvk::FF_SEL;
    if (ex) need |= rvk::FF_EXPR;
    // predicate shape: one compare term on the only loaded column, no nulls -> single-pass fast path
    if (!ex && (need & ~rvk::FF_SEL) == 0 && nvals == 1 && nterms == 1 && !p.in.terms[0].is_bool() && p.in.terms[0].code() != rvk::TC_CONST)
        need |= p.in.terms[0].is_float() ? rvk::FF_ONE_F64 : rvk::FF_ONE_I64;
    // diagnostics (per-phase stamps, ablations) exist in the FF_STAMP instantiations only; "debug" implies them
    if ((ctx->opt_stamp || ctx->opt_debug) && (need == rvk::FF_ONE_I64 || (nvals == 2 && need == rvk::FF_VALIDITY))) need |= rvk::FF_STAMP;
    // every loaded column projected, output bitmap exactly where there is an input bitmap?
    // ... or no output bitmap at all (FF_NONULL: every nullable column is tested by a null-dropping term)
    bool all_proj = nvals > 0 && !(need & (rvk::FF_ONE_I64 | rvk::FF_ONE_F64));
    for (int s = 0; s < nvals; ++s) all_proj = all_proj && p.out_values[s];
    bool mirror = all_proj, none = all_proj;
    for (int s = 0; s < nvals; ++s) {
        mirror = mirror && ((p.out_validity[s] != nullptr) == (p.in.cols[s].validity != nullptr));
        none = none && p.out_validity[s] == nullptr;
    }
    // not every loaded column projected, but no output bitmap anywhere: the staging needs no validity select either
    bool no_out_validity = nvals > 0 && (need & rvk::FF_VALIDITY) && !(need & (rvk::FF_ONE_I64 | rvk::FF_ONE_F64));
    for (int s2 = 0; s2 < nvals; ++s2) no_out_validity = no_out_validity && p.out_validity[s2] == nullptr;
    const int prefer = mirror ? rvk::FF_PROJALL
                              : ((none && (need & rvk::FF_VALIDITY)) ? (rvk::FF_PROJALL | rvk::FF_NONULL) : ((no_out_validity && !all_proj) ? rvk::FF_NONULL : 0));
    const rvk::FusedEntry *chosen = &pick_fused(ctx, nvals, vec, need, prefer);
    // per-batch counts out of the pass: a batch must be a whole number of the geometry's wave ranges
    auto counts_here = [&](const rvk::FusedEntry &g) { return req && req->counts && req->chunk_rows % (64u * static_cast<uint64_t>(g.r)) == 0; };
    if (sel_deferred && !counts_here(*chosen)) {  // the caller will count the selection bitmap instead: materialise it after all
        make_selection();
        need |= rvk::FF_SEL;
        chosen = &pick_fused(ctx, nvals, vec, need, prefer);
    }
    const rvk::FusedEntry &e = *chosen;
    if (e.flags & rvk::FF_PROJALL)  // the kernel stages a validity byte for every column when any has a bitmap
        stage_row_bytes = static_cast<size_t>(nvals) * (((e.flags & rvk::FF_VALIDITY) && !(e.flags & rvk::FF_NONULL)) ? 9 : 8) + static_cast<size_t>(nxs);
    const uint64_t tile_rows = static_cast<uint64_t>(e.waves) * 64 * e.r;
    const uint64_t ntiles64 = (n + tile_rows - 1) / tile_rows;
    require(ntiles64 < (1ull << 31), RV_ERR_UNSUPPORTED, "batch too large for one launch");
    p.ntiles = static_cast<uint32_t>(ntiles64);

    // LDS: every wave owns two slots (double buffered for the deferred look-back) of cap rows.
    // One 1024-thread workgroup per CU may use most of the 160 KiB; 512-thread variants keep to
    // half so that two workgroups fit.
    const uint32_t rows_per_wave = 64u * static_cast<uint32_t>(e.r);
    // two workgroups per CU; after a dense launch one workgroup with slots that hold every row of a wave
    const bool dense_mode = ctx->opt_rows_per_lane <= 0 && ctx->last_redo_fraction > 0.05 && nvals == 1;
    // 16 waves x 4 per SIMD is one workgroup per CU (128 VGPRs each): it may use most of the LDS
    const size_t budget = (dense_mode || e.waves >= 16) ? 144 * 1024 : 72 * 1024;
    // Three stages (write-out two iterations after the aggregate went out, so the scanner's prefix is
    // there when it is needed) when a slot still holds 3/16 of a wave's rows; two otherwise.
    auto cap_for = [&](size_t stages) -> uint32_t {
        if (!stage_row_bytes) return rows_per_wave;
        return static_cast<uint32_t>(std::min<uint64_t>(rows_per_wave, (budget / (stages * e.waves * stage_row_bytes)) & ~size_t(63)));
    };
    size_t stages = 3;
    if (ctx->opt_depth == 1 || (ctx->opt_depth == 0 && (dense_mode || cap_for(3) * 16 < rows_per_wave * 3))) stages = 2;
    uint32_t cap = cap_for(stages);
    if (ctx->opt_cap_rows > 0) cap = static_cast<uint32_t>(std::min<int64_t>(cap, std::max<int64_t>(64, ctx->opt_cap_rows & ~int64_t(63))));
    cap = std::max<uint32_t>(cap, 64u * static_cast<uint32_t>(e.vec));  // a slot holds at least one chunk
    require(cap >= 64, RV_ERR_INTERNAL, "LDS stage too small");
    // bit streams of a lane-form launch are staged as R + 1 words of bits, whatever the slot's row capacity
    auto lds_for = [&](size_t st, uint32_t rows) {
        const size_t xs_words = (e.vec == 1 && rows < (static_cast<uint32_t>(e.r) + 2) * 8u) ? static_cast<size_t>(nxs) * ((e.r + 2) * 8 - rows) : 0;
        const size_t slot = (static_cast<size_t>(rows) * stage_row_bytes + xs_words + 15) & ~size_t(15);
        return rvk::kLdsHeader + st * e.waves * slot + static_cast<size_t>(e.waves) * rvk::kLdsDumpBytes;
    };
    // the minimum slot of a wide row (several columns with validity bytes) times three stages can pass the CU's 160 KiB
    // (a forced "depth" = 2 on such a shape): two stages then
    constexpr size_t kLdsPerCu = 160 * 1024;
    if (stages == 3 && lds_for(3, cap) > kLdsPerCu) stages = 2;
    require(lds_for(stages, cap) <= kLdsPerCu, RV_ERR_UNSUPPORTED,
            fmt("fused pass: %zu bytes of LDS for %d columns at %u rows per slot", lds_for(stages, cap), nvals, cap));
    p.cap_rows = cap;
    p.depth = static_cast<int32_t>(stages) - 1;
    const size_t lds = lds_for(stages, cap);

    L.ctrl = acquire_launch_ctrl(ctx, p.ntiles);
    Ctrl *ctrl = static_cast<Ctrl *>(L.ctrl.dev);
    p.state = reinterpret_cast<uint64_t *>(static_cast<unsigned char *>(L.ctrl.dev) + kCtrlBytes);
    p.ticket = &ctrl->ticket;
    p.err = &ctrl->err;
    p.out_count = &ctrl->out_count;
    p.out_valid_pop = ctrl->valid_pop;
    p.stamps = ctrl->stamps;
    p.debug = static_cast<int32_t>(ctx->opt_debug);
    p.spin_limit = ctx->opt_spin_limit > 0 ? static_cast<uint32_t>(ctx->opt_spin_limit) : rvk::kSpinLimit;
    p.redo_count = &ctrl->redo_count;
    p.redo = reinterpret_cast<unsigned long long *>(static_cast<unsigned char *>(L.ctrl.dev) + kCtrlBytes + static_cast<size_t>(p.ntiles) * 8);

    // both calls cost several microseconds: once per (kernel, LDS size) and context
    const void *fn = reinterpret_cast<const void *>(e.fn);
    size_t &enabled = ctx->lds_enabled[fn];
    if (lds > enabled) {
        RV_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
        enabled = lds;
    }
    // persistent grid: as many workgroups as the device keeps resident (tiles are handed out
    // by the ticket counter, so residency is a speed matter only, never correctness)
    auto occ = ctx->occupancy.find({fn, lds});
    if (occ == ctx->occupancy.end()) {
        int q = 0;
        RV_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, fn, e.waves * 64, lds));
        occ = ctx->occupancy.emplace(std::make_pair(fn, lds), std::max(1, q)).first;
    }
    int per_cu = occ->second;
    if (ctx->opt_wgs_per_cu > 0) per_cu = static_cast<int>(ctx->opt_wgs_per_cu);
    // + 1: workgroup 0 is the scanner (lookback.hpp, scanner_wave)
    const uint32_t grid = 1 + static_cast<uint32_t>(std::min<uint64_t>(p.ntiles, static_cast<uint64_t>(ctx->props.multiProcessorCount) * per_cu - 1));
    p.overflow = &ctrl->overflow;
    if (ranges) {
        ranges->range_rows = 64u * static_cast<uint32_t>(e.r);
        ranges->out_capacity = cap_out;
        if (4096u % ranges->range_rows == 0) {
            ranges->offsets = pool_alloc(ctx, static_cast<size_t>(p.ntiles) * e.waves * 8 + 16);
            p.wave_offsets = static_cast<uint64_t *>(ranges->offsets->ptr);
        }
    }
    DevBufRef wave_counts;
    if (counts_here(e)) {
        wave_counts = pool_alloc(ctx, static_cast<size_t>(p.ntiles) * e.waves * 4 + 16);
        p.wave_counts = static_cast<uint32_t *>(wave_counts->ptr);
    }
    L.fn = e.fn;
    L.grid = grid;
    L.block = static_cast<uint32_t>(e.waves * 64);
    L.lds = lds;
    L.timed = ctx->opt_profile != 0;
    ctx->last_kernel = fmt("fused_filter_compact<%d,%d,%d,%d,%d>", e.ncols, e.r, e.vec, e.waves, e.flags);
    if (L.timed) RV_HIP(hipEventRecord(ctx->evk0, ctx->stream));
    hipLaunchKernelGGL(e.fn, dim3(grid), dim3(e.waves * 64), lds, ctx->stream, p);
    RV_HIP(hipGetLastError());
    if (L.timed) RV_HIP(hipEventRecord(ctx->evk1, ctx->stream));
    if (p.wave_counts) {
        // wave counts -> the caller's per-batch array (pinned host memory, written by the device: no read-back to queue);
        // the scratch goes back to the pool at scope end, every later user runs on this stream
        const uint64_t per_batch = req->chunk_rows / (64u * static_cast<uint64_t>(e.r)), nwaves = static_cast<uint64_t>(p.ntiles) * e.waves;
        const uint64_t threads = per_batch < 32 ? req->nb : (per_batch < 4096 ? req->nb * 64 : req->nb * 256);
        const dim3 cgrid(static_cast<uint32_t>(std::max<uint64_t>(1, std::min<uint64_t>((threads + 255) / 256, static_cast<uint64_t>(ctx->props.multiProcessorCount) * 8))));
        hipLaunchKernelGGL(rvk::batch_counts_from_waves, cgrid, dim3(256), 0, ctx->stream, static_cast<const uint32_t *>(p.wave_counts), nwaves, per_batch, req->nb, req->counts);
        RV_HIP(hipGetLastError());
        req->counted = true;
        ctx->batch_counts_in_pass += 1;
        p.wave_counts = nullptr;  // a re-run after an output overflow does not count again (the first pass's counts are exact)
    }
    RV_HIP(hipMemcpyAsync(L.ctrl.host, L.ctrl.dev, kCtrlBytes, hipMemcpyDeviceToHost, ctx->stream));
    RV_HIP(hipEventRecord(L.ctrl.ev, ctx->stream));
    L.launched = true;
    L.need = need;
    L.nvals = nvals;
    L.nxs = nxs;
    L.stage_row_bytes = stage_row_bytes;
    L.tile_rows = tile_rows;
}

// Waits for the launch, runs the redo kernel when tiles were dense, fixes the output lengths / null counts.
uint64_t fused_finish(rv_ctx *ctx, FusedLaunch &L) {
    if (!L.launched) return 0;
    rvk::FusedParams &p = L.p;
    std::vector<OutCol> &outs = L.outs;
    const int need = L.need, nvals = L.nvals, nxs = L.nxs;
    const size_t stage_row_bytes = L.stage_row_bytes;
    const uint64_t tile_rows = L.tile_rows;
    struct Release {
        rv_ctx *ctx;
        FusedLaunch &L;
        ~Release() {
            release_launch_ctrl(ctx, L.ctrl);
            L.launched = false;
        }
    } release{ctx, L};
    RV_HIP(hipEventSynchronize(L.ctrl.ev));
    const Ctrl *h = static_cast<const Ctrl *>(L.ctrl.host);
    if (L.timed) {
        float ms = 0.f;
        RV_HIP(hipEventElapsedTime(&ms, ctx->evk0, ctx->evk1));
        ctx->kernel_ms += ms;
        ctx->kernel_launches += 1;
    }
    require(h->err == 0, RV_ERR_DEVICE, "fused kernel: look-back spin limit reached (device fault or lost workgroup)");
    ctx->last_selectivity = L.n ? static_cast<double>(h->out_count) / static_cast<double>(L.n) : 0.0;
    if (h->overflow || h->out_count > p.out_capacity) {
        // speculative output sizing guessed too low: the count is exact, so give every output exactly that many rows
        // and run the pass once more (same kernel, same geometry, fresh descriptors)
        const uint64_t exact = h->out_count;
        p.out_capacity = exact;
        for (auto &o : outs) {
            if (o.value_slot >= 0) {
                o.col->values = pool_alloc(ctx, std::max<size_t>(exact * 8, 8));
                p.out_values[o.value_slot] = static_cast<uint64_t *>(o.col->values->ptr);
                if (o.col->validity) {
                    const size_t wb = std::max<size_t>(bitmap_words_bytes(exact) + 8, 16);
                    o.col->validity = pool_alloc(ctx, wb);
                    RV_HIP(hipMemsetAsync(o.col->validity->ptr, 0, wb, ctx->stream));
                    p.out_validity[o.value_slot] = static_cast<uint64_t *>(o.col->validity->ptr);
                }
            }
            if (o.xs_values >= 0) {
                const size_t wb = std::max<size_t>(bitmap_words_bytes(exact) + 8, 16);
                o.col->values = pool_alloc(ctx, wb);
                RV_HIP(hipMemsetAsync(o.col->values->ptr, 0, wb, ctx->stream));
                p.xs[o.xs_values].out = static_cast<uint64_t *>(o.col->values->ptr);
                if (o.xs_valid >= 0) {
                    o.col->validity = pool_alloc(ctx, wb);
                    RV_HIP(hipMemsetAsync(o.col->validity->ptr, 0, wb, ctx->stream));
                    p.xs[o.xs_valid].out = static_cast<uint64_t *>(o.col->validity->ptr);
                }
            }
        }
        const size_t zeroed = kCtrlBytes + static_cast<size_t>(p.ntiles) * 8;
        RV_HIP(hipMemsetAsync(L.ctrl.dev, 0, (zeroed + 15) & ~size_t(15), ctx->stream));
        hipLaunchKernelGGL(L.fn, dim3(L.grid), dim3(L.block), L.lds, ctx->stream, p);
        RV_HIP(hipGetLastError());
        RV_HIP(hipMemcpyAsync(L.ctrl.host, L.ctrl.dev, kCtrlBytes, hipMemcpyDeviceToHost, ctx->stream));
        RV_HIP(hipStreamSynchronize(ctx->stream));
        require(h->err == 0 && h->overflow == 0 && h->out_count == exact, RV_ERR_INTERNAL, "re-run after an output overflow disagrees with the first pass");
        ctx->overflow_reruns += 1;
    }
    ctx->last_redo_fraction = static_cast<double>(h->redo_count) / static_cast<double>(p.ntiles);
    if (h->redo_count > 0 && (stage_row_bytes || nxs)) {
        // dense tiles: re-read them with the generic kernel at their reserved output offsets
        const rvk::RedoFn redo = rvk::redo_kernel(nvals);
        require(redo != nullptr, RV_ERR_INTERNAL, "no redo kernel variant");
        const size_t redo_lds = rvk::kLdsHeader + 2048 * stage_row_bytes;
        RV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(redo), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(redo_lds)));
        const uint32_t rgrid = std::min<uint32_t>(h->redo_count, static_cast<uint32_t>(ctx->props.multiProcessorCount) * 2);
        hipLaunchKernelGGL(redo, dim3(rgrid), dim3(1024), redo_lds, ctx->stream, p, static_cast<uint32_t>(tile_rows));
        RV_HIP(hipGetLastError());
        RV_HIP(hipMemcpyAsync(L.ctrl.host, L.ctrl.dev, kCtrlBytes, hipMemcpyDeviceToHost, ctx->stream));
        RV_HIP(hipStreamSynchronize(ctx->stream));
    }
    const uint64_t rows = h->out_count;
    if (ctx->opt_debug & 4)
        fprintf(stderr, "[scan] tiles %llu | scanner polls %llu, tiles scanned %llu, empty polls %llu | fallback look-backs %llu\n",
                static_cast<unsigned long long>(p.ntiles), h->stamps[28], h->stamps[29], h->stamps[30], h->stamps[31]);
    if ((need & rvk::FF_STAMP) && ctx->opt_stamp) {
        std::memcpy(ctx->last_stamps, h->stamps, sizeof(h->stamps));
        for (int w = 0; w < 2; ++w) {
            const unsigned long long *q = h->stamps + 8 * w;
            const double t = static_cast<double>(std::max<unsigned long long>(1, q[5]));
            if (w == 1) {
                fprintf(stderr, "[stamp] wave1 cycles/tile: of eval: load wait %.0f, stage %.0f\n", q[6] / t, q[7] / t);
            }
            fprintf(stderr, "[stamp] wave%d cycles/tile: eval(+ticket,+load wait) %.0f | scatter+prefetch %.0f | lookback %.0f | barrierB %.0f | flush %.0f | tiles %llu | polls/tile %.2f windows/tile %.2f\n",
                    w, q[0] / t, q[1] / t, q[2] / t, q[3] / t, q[4] / t, q[5], q[6] / t, q[7] / t);
        }
    }
    for (auto &o : outs) {
        o.col->length = rows;
        o.col->offset = 0;
        long long valid_pop = -1;
        if (o.value_slot >= 0 && o.col->validity) valid_pop = static_cast<long long>(h->valid_pop[o.value_slot]);
        if (o.xs_valid >= 0) valid_pop = static_cast<long long>(h->valid_pop[rvk::kMaxValueCols + o.xs_valid]);
        if (valid_pop < 0) {
            o.col->null_count = 0;
        } else {
            o.col->null_count = static_cast<int64_t>(rows) - valid_pop;
            if (o.col->null_count == 0) o.col->validity.reset();  // builder drops it (primitive.rs:179-185)
        }
    }
    return rows;
}

// begin + finish: the synchronous form
// `after_launch` (optional) runs between the two halves, with the selection bitmap the pass is writing: work queued there
// (the String gather of a filter) follows the pass on the stream without the host having waited for anything.
using AfterLaunch = std::function<void(const rv_dcolumn *sel)>;
uint64_t run_fused_pass(rv_ctx *ctx, const rv_dcolumn *const *cols, uint32_t ncols, const rv_term *terms,
                        uint32_t nterms, rv_null_policy policy, const uint32_t *proj, uint32_t nproj,
                        rv_dcolumn **out, rv_dcolumn **sel_out, const ExprInfo *ex = nullptr, BatchReq *req = nullptr,
                        const AfterLaunch *after_launch = nullptr, RangeOffsets *ranges = nullptr) {
    FusedLaunch L;
    fused_begin(ctx, cols, ncols, terms, nterms, policy, proj, nproj, out, sel_out, L, ex, req, ranges);
    if (after_launch && *after_launch) {
        try {
            (*after_launch)(sel_out ? *sel_out : nullptr);
        } catch (...) {
            if (L.launched) {  // the pass may still be running: drain before its buffers go
                (void)hipStreamSynchronize(ctx->stream);
                release_launch_ctrl(ctx, L.ctrl);
                L.launched = false;
            }
            throw;
        }
    }
    return fused_finish(ctx, L);
}

}  // namespace

// =========================================================================================
extern "C" {

uint32_t rv_abi_version(void) { return RV_ABI_VERSION; }
const char *rv_last_error(void) { return g_last_error.c_str(); }
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
        (void)hipStreamSynchronize(ctx->stream);
        if (ctx->d_ctrl) (void)hipFree(ctx->d_ctrl);
        if (ctx->d_stripes) (void)hipFree(ctx->d_stripes);
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
        else if (k == "spin_limit") ctx->opt_spin_limit = value;
        else if (k == "agg_grid") ctx->opt_agg_grid = value;
        else if (k == "bools_in_pass") ctx->opt_bools_in_pass = value;
        else if (k == "inject_failure") ctx->opt_inject_failure = value;
        else if (k == "out_sizing") {
            require(value >= 0 && value <= 1000000, RV_ERR_INVALID_ARG, "out_sizing: 0, 1 or a bound in rows per million");
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
        else if (k == "spin_limit") *value = ctx->opt_spin_limit;
        else if (k == "agg_grid") *value = ctx->opt_agg_grid;
        else if (k == "bools_in_pass") *value = ctx->opt_bools_in_pass;
        else if (k == "inject_failure") *value = ctx->opt_inject_failure;
        else if (k == "out_sizing") *value = ctx->opt_out_sizing;
        else if (k == "overflow_reruns") *value = static_cast<int64_t>(ctx->overflow_reruns);  // read-only counter
        else if (k == "batch_counts_in_pass") *value = static_cast<int64_t>(ctx->batch_counts_in_pass);  // read-only counter
        else if (k == "fused_rows_scanned") *value = static_cast<int64_t>(ctx->fused_rows_scanned);  // read-only counter
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

// ---- StringArray on the device (string_kernels.hpp) ---------------------------------------------------
namespace {
// exclusive scan of n counts -> (n + 1) uint64 prefixes.  The counts are uint32 values or (pop) the popcounts of 64-bit
// words read in place.  want_total: wait for the result and return the total (else 0, nothing is waited for).
uint64_t device_exclusive_scan(rv_ctx *ctx, const void *counts, uint64_t n, DevBufRef &excl, bool pop = false, bool want_total = true) {
    excl = pool_alloc(ctx, (n + 1) * 8 + 16);
    if (n == 0) {
        RV_HIP(hipMemsetAsync(excl->ptr, 0, 8, ctx->stream));
        return 0;
    }
    const uint64_t nblocks = (n + rvk::kScanBlock - 1) / rvk::kScanBlock;
    DevBufRef sums = pool_alloc(ctx, nblocks * 8 + 16);
    Ctrl *ctrl = prepare_ctrl(ctx, 0);
    const dim3 grid(static_cast<uint32_t>(nblocks)), block(rvk::kScanThreads);
    if (pop) hipLaunchKernelGGL(rvk::scan_block_sums<true>, grid, block, 0, ctx->stream, counts, n, static_cast<uint64_t *>(sums->ptr));
    else hipLaunchKernelGGL(rvk::scan_block_sums<false>, grid, block, 0, ctx->stream, counts, n, static_cast<uint64_t *>(sums->ptr));
    hipLaunchKernelGGL(rvk::scan_sums_inplace, dim3(1), dim3(1024), 0, ctx->stream, static_cast<uint64_t *>(sums->ptr), nblocks,
                       &ctrl->pops[0]);
    if (pop) hipLaunchKernelGGL(rvk::scan_apply<true>, grid, block, 0, ctx->stream, counts, n, static_cast<const uint64_t *>(sums->ptr), static_cast<uint64_t *>(excl->ptr));
    else hipLaunchKernelGGL(rvk::scan_apply<false>, grid, block, 0, ctx->stream, counts, n, static_cast<const uint64_t *>(sums->ptr), static_cast<uint64_t *>(excl->ptr));
    RV_HIP(hipGetLastError());
    // `sums` goes back to the pool here; every later user runs on this stream, after the kernels that read it
    if (!want_total) return 0;
    const Ctrl *h = fetch_ctrl(ctx);
    return h->pops[0];
}

// exclusive survivor counts per 64-row word of a selection bitmap (BooleanArray without validity, offset 0): a scan over
// the popcounts of the words, read in place; nothing is waited for (the survivor count is the fused pass's)
DevBufRef selection_prefix(rv_ctx *ctx, const rv_dcolumn *sel, uint64_t rows) {
    (void)rows;
    const uint64_t nwords = (sel->length + 63) / 64;
    DevBufRef excl;
    device_exclusive_scan(ctx, sel->values->ptr, nwords, excl, true, false);
    return excl;
}
// ... -> ascending row indices
DevBufRef selection_to_indices(rv_ctx *ctx, const rv_dcolumn *sel, uint64_t rows, const DevBufRef &excl) {
    const uint64_t nwords = (sel->length + 63) / 64;
    DevBufRef indices = pool_alloc(ctx, std::max<size_t>(rows * 8, 16));
    if (rows == 0 || nwords == 0) return indices;
    hipLaunchKernelGGL(rvk::sel_expand_indices, dim3(static_cast<uint32_t>((nwords + 255) / 256)), dim3(256), 0, ctx->stream,
                       static_cast<const uint64_t *>(sel->values->ptr), nwords, static_cast<const uint64_t *>(excl->ptr),
                       static_cast<uint64_t *>(indices->ptr));
    RV_HIP(hipGetLastError());
    return indices;
}
// ... -> a Boolean column compacted by it (values under their validity, and the validity itself)
rv_dcolumn *compact_boolean(rv_ctx *ctx, const rv_dcolumn *src, const rv_dcolumn *sel, uint64_t rows, const DevBufRef &excl) {
    auto o = std::make_unique<rv_dcolumn>();
    o->dtype = RV_BOOLEAN;
    o->length = rows;
    const size_t wb = std::max<size_t>(bitmap_words_bytes(rows) + 8, 16);
    o->values = pool_alloc(ctx, wb);
    RV_HIP(hipMemsetAsync(o->values->ptr, 0, wb, ctx->stream));
    if (src->validity) {
        o->validity = pool_alloc(ctx, wb);
        RV_HIP(hipMemsetAsync(o->validity->ptr, 0, wb, ctx->stream));
    }
    const uint64_t nwords = (sel->length + 63) / 64;
    if (rows == 0 || nwords == 0) {
        o->validity.reset();
        o->null_count = 0;
        return o.release();
    }
    Ctrl *ctrl = prepare_ctrl(ctx, 0);
    rvk::BitsCompact b{};
    b.sel = static_cast<const uint64_t *>(sel->values->ptr);
    b.nwords = nwords;
    b.offset = src->offset;
    b.excl = static_cast<const uint64_t *>(excl->ptr);
    const dim3 grid(static_cast<uint32_t>(std::min<uint64_t>((nwords + 255) / 256, static_cast<uint64_t>(ctx->props.multiProcessorCount) * 8)));
    // values: false under a null (BooleanArray::new, boolean.rs:29-32); the validity rides in the same launch
    b.src = static_cast<const uint8_t *>(src->values->ptr);
    b.src_bytes = src->values->bytes;
    b.mask = src->validity ? static_cast<const uint8_t *>(src->validity->ptr) : nullptr;
    b.mask_bytes = src->validity ? src->validity->bytes : 0;
    b.out = static_cast<uint64_t *>(o->values->ptr);
    b.pop = striped(ctx, &ctrl->pops[0]);
    if (src->validity) {
        b.src2 = static_cast<const uint8_t *>(src->validity->ptr);
        b.src2_bytes = src->validity->bytes;
        b.out2 = static_cast<uint64_t *>(o->validity->ptr);
        b.pop2 = striped(ctx, &ctrl->pops[1]);
    }
    hipLaunchKernelGGL(rvk::bits_compact_kernel, grid, dim3(256), 0, ctx->stream, b);
    RV_HIP(hipGetLastError());
    const Ctrl *h = fetch_ctrl(ctx);
    o->null_count = src->validity ? static_cast<int64_t>(rows - h->pops[1]) : 0;
    if (o->null_count == 0) o->validity.reset();  // BooleanArrayBuilder::finish (boolean.rs:282-286)
    return o.release();
}

// Shared tail of the two String gathers.  `lengths` / `starts` of the n output elements are on their way (queued on the
// stream, control block prepared by the caller); here: sums of the lengths per block of kStrBlock elements -> sums per
// group of kStrGroup blocks -> the scan of those
// -> ONE read of the control block (total bytes, surviving valid elements, bounds error) -> the byte copy, whose
// workgroups scan the lengths inside their block themselves (no per-element prefix array is written or read).
void finish_string_gather(rv_ctx *ctx, rv_dcolumn *o, rvk::StrGather &g, uint64_t n, Ctrl *ctrl) {
    const uint64_t nblocks = (n + rvk::kStrBlock - 1) / rvk::kStrBlock;
    DevBufRef sums = pool_alloc(ctx, nblocks * 8 + 16);
    hipLaunchKernelGGL(rvk::str_block_sums, dim3(static_cast<uint32_t>((nblocks + 3) / 4)), dim3(256), 0, ctx->stream,
                       static_cast<const uint32_t *>(g.lengths), n, static_cast<uint64_t *>(sums->ptr));
    const uint64_t ngroups = (nblocks + rvk::kStrGroup - 1) / rvk::kStrGroup;
    DevBufRef groups = pool_alloc(ctx, ngroups * 8 + 16);
    hipLaunchKernelGGL(rvk::str_group_sums, dim3(static_cast<uint32_t>((ngroups + 3) / 4)), dim3(256), 0, ctx->stream,
                       static_cast<const uint64_t *>(sums->ptr), nblocks, static_cast<uint64_t *>(groups->ptr));
    hipLaunchKernelGGL(rvk::scan_sums_inplace, dim3(1), dim3(1024), 0, ctx->stream, static_cast<uint64_t *>(groups->ptr), ngroups, &ctrl->pops[0]);
    RV_HIP(hipGetLastError());
    const Ctrl *h = fetch_ctrl(ctx);
    require(h->err == 0, RV_ERR_OUT_OF_BOUNDS, "string gather: index out of bounds");
    const uint64_t valid = o->validity ? h->valid_pop[0] : n, total = h->pops[0];
    require(total <= 0x7FFFFFFFull, RV_ERR_UNSUPPORTED, "StringArray data larger than 2 GiB (int32 offsets, string.rs:11)");
    o->values = pool_alloc(ctx, std::max<size_t>(total + 8, 16));  // + 8: gathers read aligned words
    o->data_bytes = total;
    g.block_sums = static_cast<const uint64_t *>(sums->ptr);
    g.group_base = static_cast<const uint64_t *>(groups->ptr);
    g.total_bytes = total;
    g.out_offsets = static_cast<int32_t *>(o->offsets->ptr);
    g.out_data = static_cast<uint8_t *>(o->values->ptr);
    hipLaunchKernelGGL(rvk::str_gather_copy, dim3(static_cast<uint32_t>(nblocks)), dim3(rvk::kStrBlock), 0, ctx->stream, g);
    RV_HIP(hipGetLastError());
    RV_HIP(hipStreamSynchronize(ctx->stream));  // lengths / starts / sums go back to the pool
    o->null_count = static_cast<int64_t>(n - valid);
    if (o->null_count == 0) o->validity.reset();  // builder drops the bitmap (string.rs:41-45)
}

std::unique_ptr<rv_dcolumn> empty_string_gather(rv_ctx *ctx, uint64_t n) {
    auto o = std::make_unique<rv_dcolumn>();
    o->dtype = RV_STRING;
    o->length = n;
    o->offsets = pool_alloc(ctx, (n + 1) * 4 + 16);
    if (n == 0) {
        RV_HIP(hipMemsetAsync(o->offsets->ptr, 0, 4, ctx->stream));
        o->values = pool_alloc(ctx, 16);
        o->null_count = 0;
    }
    return o;
}

// take() of a StringArray (record_batch.rs:163-170 -> StringArray::new, string.rs:19-57)
rv_dcolumn *gather_strings(rv_ctx *ctx, const rv_dcolumn *src, const uint64_t *d_indices, uint64_t n) {
    auto o = empty_string_gather(ctx, n);
    if (n == 0) return o.release();
    DevBufRef lengths = pool_alloc(ctx, n * 4 + 16), starts = pool_alloc(ctx, n * 4 + 16);
    if (src->validity) o->validity = pool_alloc(ctx, std::max<size_t>(bitmap_words_bytes(n), 16));
    Ctrl *ctrl = prepare_ctrl(ctx, 0);
    rvk::StrGather g{};
    g.offsets = static_cast<const int32_t *>(src->offsets->ptr);
    g.data = static_cast<const uint8_t *>(src->values->ptr);
    g.validity = src->validity ? static_cast<const uint8_t *>(src->validity->ptr) : nullptr;
    g.validity_bytes = src->validity ? src->validity->bytes : 0;
    g.offset = src->offset;
    g.src_length = src->length;
    g.indices = d_indices;
    g.n = n;
    g.lengths = static_cast<uint32_t *>(lengths->ptr);
    g.starts = static_cast<int32_t *>(starts->ptr);
    g.out_validity = o->validity ? static_cast<uint64_t *>(o->validity->ptr) : nullptr;
    g.valid_pop = striped(ctx, &ctrl->valid_pop[0]);
    g.err = &ctrl->err;
    hipLaunchKernelGGL(rvk::str_gather_lengths, dim3(static_cast<uint32_t>((n + 255) / 256)), dim3(256), 0, ctx->stream, g);
    RV_HIP(hipGetLastError());
    finish_string_gather(ctx, o.get(), g, n, ctrl);
    return o.release();
}

// filter() of a StringArray: the same array as gather_strings over the ascending indices of `sel`'s set bits
// (record_batch.rs:131-178), built without the index list: (start, length) of the survivors straight from the selection
// words (sel_str_lengths), the surviving validity bits by the bitmap compaction Boolean columns use.
rv_dcolumn *gather_strings_selected(rv_ctx *ctx, const rv_dcolumn *src, const rv_dcolumn *sel, uint64_t rows, const DevBufRef &excl) {
    auto o = empty_string_gather(ctx, rows);
    if (rows == 0) return o.release();
    const uint64_t nwords = (sel->length + 63) / 64;
    DevBufRef lengths = pool_alloc(ctx, rows * 4 + 16), starts = pool_alloc(ctx, rows * 4 + 16);
    Ctrl *ctrl = prepare_ctrl(ctx, 0);
    const uint32_t wgs = static_cast<uint32_t>((nwords + 255) / 256);  // a wave per 64 selection words
    if (src->validity) {
        const size_t wb = std::max<size_t>(bitmap_words_bytes(rows) + 8, 16);
        o->validity = pool_alloc(ctx, wb);
        RV_HIP(hipMemsetAsync(o->validity->ptr, 0, wb, ctx->stream));
        rvk::BitsCompact b{};
        b.sel = static_cast<const uint64_t *>(sel->values->ptr);
        b.nwords = nwords;
        b.offset = src->offset;
        b.excl = static_cast<const uint64_t *>(excl->ptr);
        b.src = static_cast<const uint8_t *>(src->validity->ptr);
        b.src_bytes = src->validity->bytes;
        b.out = static_cast<uint64_t *>(o->validity->ptr);
        b.pop = striped(ctx, &ctrl->valid_pop[0]);
        const dim3 grid(static_cast<uint32_t>(std::min<uint64_t>((nwords + 255) / 256, static_cast<uint64_t>(ctx->props.multiProcessorCount) * 8)));
        hipLaunchKernelGGL(rvk::bits_compact_kernel, grid, dim3(256), 0, ctx->stream, b);
    }
    rvk::SelStr q{};
    q.sel = static_cast<const uint64_t *>(sel->values->ptr);
    q.nwords = nwords;
    q.excl = static_cast<const uint64_t *>(excl->ptr);
    q.offsets = static_cast<const int32_t *>(src->offsets->ptr);
    q.validity = src->validity ? static_cast<const uint8_t *>(src->validity->ptr) : nullptr;
    q.offset = src->offset;
    q.length = src->length;
    q.cap_rows = rows;
    q.lengths = static_cast<uint32_t *>(lengths->ptr);
    q.starts = static_cast<int32_t *>(starts->ptr);
    hipLaunchKernelGGL(rvk::sel_str_lengths, dim3(wgs), dim3(256), 0, ctx->stream, q);
    RV_HIP(hipGetLastError());
    rvk::StrGather g{};
    g.data = static_cast<const uint8_t *>(src->values->ptr);
    g.n = rows;
    g.lengths = q.lengths;
    g.starts = q.starts;
    finish_string_gather(ctx, o.get(), g, rows, ctrl);
    return o.release();
}

// filter() of a StringArray (record_batch.rs:131-178 -> string.rs:19-57) in three launches behind the fused pass, with no
// host round trip of its own:
//   str_sel_queue   sel_str_lengths, queued while the pass is still writing the selection bitmap: (start, length) of the
//                   survivors at the pass's wave offsets (no scan over the bitmap) + the byte sums per block of 256
//                   elements (atomics); a nullable column's validity bits are compacted next to it (bits_compact_kernel);
//   str_sel_copy    once the pass has told the host the survivor count (the wait the pass needs anyway; the lengths
//                   launch runs meanwhile): str_sums_scan (group sums + scan + total, one workgroup) and
//                   str_gather_copy.  The output bytes are sized by the source's bytes: nothing to read back first;
//   str_sel_result  after the call's one fetch of the control block: total bytes and surviving valid elements.
struct StrSelLaunch {
    std::unique_ptr<rv_dcolumn> col;
    const rv_dcolumn *src = nullptr;
    DevBufRef lengths, starts, block_sums;
    unsigned long long *group_sums = nullptr;  // inside block_sums' buffer
    uint64_t cap_rows = 0;
    Ctrl *ctrl = nullptr;
    int slot = 0;  // valid_pop[slot]: surviving valid elements; pops[0]: total bytes
    bool queued = false;
};
bool str_sel_eligible(const rv_dcolumn *sel, const RangeOffsets &ranges) { return sel != nullptr && sel->length > 0 && ranges.offsets != nullptr; }
void str_sel_queue(rv_ctx *ctx, const rv_dcolumn *src, const rv_dcolumn *sel, const RangeOffsets &ranges, Ctrl *ctrl, int slot, StrSelLaunch &L) {
    const uint64_t nwords = (sel->length + 63) / 64, cap = ranges.out_capacity;
    L.src = src;
    L.ctrl = ctrl;
    L.slot = slot;
    L.cap_rows = cap;
    L.col = std::make_unique<rv_dcolumn>();
    rv_dcolumn *o = L.col.get();
    o->dtype = RV_STRING;
    o->offsets = pool_alloc(ctx, (cap + 1) * 4 + 16);
    L.lengths = pool_alloc(ctx, cap * 4 + 16);
    L.starts = pool_alloc(ctx, cap * 4 + 16);
    const uint64_t max_blocks = (cap + rvk::kStrBlock - 1) / rvk::kStrBlock;
    const uint64_t max_groups = (max_blocks + rvk::kStrGroup - 1) / rvk::kStrGroup;
    L.block_sums = pool_alloc(ctx, (max_blocks + max_groups) * 8 + 32);  // [block sums | group sums], one memset
    RV_HIP(hipMemsetAsync(L.block_sums->ptr, 0, (max_blocks + max_groups) * 8 + 32, ctx->stream));
    L.group_sums = static_cast<unsigned long long *>(L.block_sums->ptr) + max_blocks + 1;
    if (src->validity) {  // the output bitmap: the source's, compacted by the same selection at the same offsets
        const size_t wb = std::max<size_t>(bitmap_words_bytes(cap) + 8, 16);
        o->validity = pool_alloc(ctx, wb);
        RV_HIP(hipMemsetAsync(o->validity->ptr, 0, wb, ctx->stream));
        rvk::BitsCompact b{};
        b.sel = static_cast<const uint64_t *>(sel->values->ptr);
        b.nwords = nwords;
        b.offset = src->offset;
        b.range_offsets = static_cast<const uint64_t *>(ranges.offsets->ptr);
        b.range_rows = ranges.range_rows;
        b.out_capacity = cap;
        b.src = static_cast<const uint8_t *>(src->validity->ptr);
        b.src_bytes = src->validity->bytes;
        b.out = static_cast<uint64_t *>(o->validity->ptr);
        b.pop = striped(ctx, &ctrl->valid_pop[slot]);
        const dim3 grid(static_cast<uint32_t>(std::min<uint64_t>((nwords + 255) / 256, static_cast<uint64_t>(ctx->props.multiProcessorCount) * 8)));
        hipLaunchKernelGGL(rvk::bits_compact_kernel, grid, dim3(256), 0, ctx->stream, b);
        RV_HIP(hipGetLastError());
    }
    rvk::SelStr q{};
    q.sel = static_cast<const uint64_t *>(sel->values->ptr);
    q.nwords = nwords;
    q.range_offsets = static_cast<const uint64_t *>(ranges.offsets->ptr);
    q.range_rows = ranges.range_rows;
    q.block_sums = static_cast<unsigned long long *>(L.block_sums->ptr);
    q.group_sums = L.group_sums;
    q.cap_rows = cap;
    q.offsets = static_cast<const int32_t *>(src->offsets->ptr);
    q.validity = src->validity ? static_cast<const uint8_t *>(src->validity->ptr) : nullptr;
    q.offset = src->offset;
    q.length = src->length;
    q.lengths = static_cast<uint32_t *>(L.lengths->ptr);
    q.starts = static_cast<int32_t *>(L.starts->ptr);
    hipLaunchKernelGGL(rvk::sel_str_lengths, dim3(static_cast<uint32_t>((nwords + 255) / 256)), dim3(256), 0, ctx->stream, q);
    RV_HIP(hipGetLastError());
    L.queued = true;
}
// rows: the pass's survivor count (<= cap_rows: the caller takes the other path after an overflow re-run)
void str_sel_copy(rv_ctx *ctx, StrSelLaunch &L, uint64_t rows) {
    rv_dcolumn *o = L.col.get();
    o->length = rows;
    const uint64_t cap_bytes = std::min<uint64_t>(L.src->data_bytes, 0x7FFFFFFFull);  // the survivors' bytes are among the source's
    o->values = pool_alloc(ctx, std::max<size_t>(cap_bytes + 8, 16));
    if (rows == 0) {
        RV_HIP(hipMemsetAsync(o->offsets->ptr, 0, 4, ctx->stream));
        return;
    }
    const uint64_t nblocks = (rows + rvk::kStrBlock - 1) / rvk::kStrBlock, ngroups = (nblocks + rvk::kStrGroup - 1) / rvk::kStrGroup;
    hipLaunchKernelGGL(rvk::str_sums_scan, dim3(1), dim3(1024), 0, ctx->stream, reinterpret_cast<uint64_t *>(L.group_sums), ngroups, &L.ctrl->pops[0],
                       static_cast<int32_t *>(o->offsets->ptr), rows);
    rvk::StrGather g{};
    g.data = static_cast<const uint8_t *>(L.src->values->ptr);
    g.n = rows;
    g.lengths = static_cast<uint32_t *>(L.lengths->ptr);
    g.starts = static_cast<int32_t *>(L.starts->ptr);
    g.block_sums = static_cast<const uint64_t *>(L.block_sums->ptr);
    g.group_base = reinterpret_cast<const uint64_t *>(L.group_sums);
    g.total_bytes = ~0ull;  // out_offsets[rows] is str_sums_scan's
    g.out_offsets = static_cast<int32_t *>(o->offsets->ptr);
    g.out_data = static_cast<uint8_t *>(o->values->ptr);
    hipLaunchKernelGGL(rvk::str_gather_copy, dim3(static_cast<uint32_t>(nblocks)), dim3(rvk::kStrBlock), 0, ctx->stream, g);
    RV_HIP(hipGetLastError());
}
rv_dcolumn *str_sel_result(StrSelLaunch &L, uint64_t rows, const Ctrl &fetched) {
    rv_dcolumn *o = L.col.get();
    const uint64_t total = rows ? fetched.pops[0] : 0;
    require(total <= 0x7FFFFFFFull, RV_ERR_UNSUPPORTED, "StringArray data larger than 2 GiB (int32 offsets, string.rs:11)");
    o->data_bytes = total;
    const uint64_t valid = o->validity ? fetched.valid_pop[L.slot] : rows;
    o->null_count = static_cast<int64_t>(rows - valid);
    if (o->null_count == 0) o->validity.reset();  // builder drops the bitmap (string.rs:41-45)
    return L.col.release();
}

// filter() of a BooleanArray queued right behind the fused pass: bits_compact_kernel finds every wave's output position in
// the pass's wave offsets (RangeOffsets) instead of a scan over the selection bitmap -- one launch, nothing waited for.
// Counter `slot` of the shared control block (valid_pop[slot]) receives the surviving validity bits.
struct BoolCompactLaunch {
    std::unique_ptr<rv_dcolumn> col;
    int slot = -1;
    bool launched = false;
};
void bool_compact_queue(rv_ctx *ctx, const rv_dcolumn *src, const rv_dcolumn *sel, const RangeOffsets &ranges, Ctrl *ctrl, int slot, BoolCompactLaunch &L) {
    auto o = std::make_unique<rv_dcolumn>();
    o->dtype = RV_BOOLEAN;
    const uint64_t cap = ranges.out_capacity, nwords = (sel->length + 63) / 64;
    const size_t wb = std::max<size_t>(bitmap_words_bytes(cap) + 8, 16);
    o->values = pool_alloc(ctx, wb);
    RV_HIP(hipMemsetAsync(o->values->ptr, 0, wb, ctx->stream));
    if (src->validity) {
        o->validity = pool_alloc(ctx, wb);
        RV_HIP(hipMemsetAsync(o->validity->ptr, 0, wb, ctx->stream));
    }
    L.slot = slot;
    if (nwords && ranges.offsets) {
        rvk::BitsCompact b{};
        b.sel = static_cast<const uint64_t *>(sel->values->ptr);
        b.nwords = nwords;
        b.offset = src->offset;
        b.range_offsets = static_cast<const uint64_t *>(ranges.offsets->ptr);
        b.range_rows = ranges.range_rows;
        b.out_capacity = cap;
        // values: false under a null (BooleanArray::new, boolean.rs:29-32); the validity rides in the same launch
        b.src = static_cast<const uint8_t *>(src->values->ptr);
        b.src_bytes = src->values->bytes;
        b.mask = src->validity ? static_cast<const uint8_t *>(src->validity->ptr) : nullptr;
        b.mask_bytes = src->validity ? src->validity->bytes : 0;
        b.out = static_cast<uint64_t *>(o->values->ptr);
        b.pop = striped(ctx, &ctrl->pops[2]);  // set value bits: not needed by anyone, one shared counter
        if (src->validity) {
            b.src2 = static_cast<const uint8_t *>(src->validity->ptr);
            b.src2_bytes = src->validity->bytes;
            b.out2 = static_cast<uint64_t *>(o->validity->ptr);
            b.pop2 = striped(ctx, &ctrl->valid_pop[slot]);
        }
        const dim3 grid(static_cast<uint32_t>(std::min<uint64_t>((nwords + 255) / 256, static_cast<uint64_t>(ctx->props.multiProcessorCount) * 8)));
        hipLaunchKernelGGL(rvk::bits_compact_kernel, grid, dim3(256), 0, ctx->stream, b);
        RV_HIP(hipGetLastError());
        L.launched = true;
    }
    L.col = std::move(o);
}
rv_dcolumn *bool_compact_result(BoolCompactLaunch &L, uint64_t rows, const Ctrl &fetched) {
    rv_dcolumn *o = L.col.get();
    o->length = rows;
    o->null_count = o->validity ? static_cast<int64_t>(rows - fetched.valid_pop[L.slot]) : 0;
    if (o->null_count == 0) o->validity.reset();  // BooleanArrayBuilder::finish (boolean.rs:282-286)
    return L.col.release();
}

// concat_arrays, string branch (record_batch.rs:277-342).  Parts are StringArrays as the reference builds
// them: a null element spans no bytes, so a part's logical bytes are one contiguous range.
rv_dcolumn *concat_strings(rv_ctx *ctx, const rv_dcolumn *const *parts, uint32_t nparts) {
    std::vector<rvk::StrPart> hp(nparts);
    std::vector<uint64_t> starts(nparts + 1, 0);
    bool any_validity = false;
    for (uint32_t i = 0; i < nparts; ++i) {
        require(parts[i] && parts[i]->dtype == RV_STRING, RV_ERR_TYPE_MISMATCH, "All batches must have the same schema");
        hp[i].offsets = static_cast<const int32_t *>(parts[i]->offsets->ptr);
        hp[i].data = static_cast<const uint8_t *>(parts[i]->values->ptr);
        hp[i].validity = parts[i]->validity ? static_cast<const uint8_t *>(parts[i]->validity->ptr) : nullptr;
        hp[i].offset = parts[i]->offset;
        hp[i].length = parts[i]->length;
        any_validity |= parts[i]->validity != nullptr;
        starts[i + 1] = starts[i] + parts[i]->length;
    }
    const uint64_t n = starts[nparts];
    DevBufRef d_parts = pool_alloc(ctx, nparts * sizeof(rvk::StrPart) + 16);
    DevBufRef d_ranges = pool_alloc(ctx, nparts * 16 + 16);
    RV_HIP(hipMemcpyAsync(d_parts->ptr, hp.data(), nparts * sizeof(rvk::StrPart), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(rvk::str_part_ranges, dim3((nparts + 63) / 64), dim3(64), 0, ctx->stream, static_cast<const rvk::StrPart *>(d_parts->ptr),
                       nparts, static_cast<int64_t *>(d_ranges->ptr));
    RV_HIP(hipGetLastError());
    std::vector<int64_t> ranges(2 * static_cast<size_t>(nparts));
    RV_HIP(hipMemcpyAsync(ranges.data(), d_ranges->ptr, ranges.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    RV_HIP(hipStreamSynchronize(ctx->stream));
    std::vector<int64_t> byte_start(nparts + 1, 0);
    for (uint32_t i = 0; i < nparts; ++i) byte_start[i + 1] = byte_start[i] + (ranges[2 * i + 1] - ranges[2 * i]);
    const uint64_t total = static_cast<uint64_t>(byte_start[nparts]);
    require(total <= 0x7FFFFFFFull, RV_ERR_UNSUPPORTED, "StringArray data larger than 2 GiB (int32 offsets, string.rs:11)");
    auto o = std::make_unique<rv_dcolumn>();
    o->dtype = RV_STRING;
    o->length = n;
    o->data_bytes = total;
    o->values = pool_alloc(ctx, std::max<size_t>(total + 8, 16));
    o->offsets = pool_alloc(ctx, (n + 1) * 4 + 16);
    if (any_validity) o->validity = pool_alloc(ctx, std::max<size_t>(bitmap_words_bytes(n), 16));
    for (uint32_t i = 0; i < nparts; ++i) {
        const size_t len = static_cast<size_t>(ranges[2 * i + 1] - ranges[2 * i]);
        if (len) RV_HIP(hipMemcpyAsync(static_cast<char *>(o->values->ptr) + byte_start[i], hp[i].data + ranges[2 * i], len, hipMemcpyDeviceToDevice, ctx->stream));
    }
    DevBufRef d_starts = pool_alloc(ctx, (nparts + 1) * 8 + 16), d_bytes = pool_alloc(ctx, (nparts + 1) * 8 + 16);
    RV_HIP(hipMemcpyAsync(d_starts->ptr, starts.data(), (nparts + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    RV_HIP(hipMemcpyAsync(d_bytes->ptr, byte_start.data(), (nparts + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    Ctrl *ctrl = prepare_ctrl(ctx, 0);
    rvk::StrConcat c{};
    c.parts = static_cast<const rvk::StrPart *>(d_parts->ptr);
    c.part_start = static_cast<const uint64_t *>(d_starts->ptr);
    c.byte_start = static_cast<const int64_t *>(d_bytes->ptr);
    c.ranges = static_cast<const int64_t *>(d_ranges->ptr);
    c.nparts = nparts;
    c.n = n;
    c.out_offsets = static_cast<int32_t *>(o->offsets->ptr);
    c.out_validity = o->validity ? static_cast<uint64_t *>(o->validity->ptr) : nullptr;
    c.valid_pop = striped(ctx, &ctrl->valid_pop[0]);
    if (n) {
        hipLaunchKernelGGL(rvk::str_concat_offsets, dim3(static_cast<uint32_t>((n + 255) / 256)), dim3(256), 0, ctx->stream, c);
        RV_HIP(hipGetLastError());
    } else {
        RV_HIP(hipMemsetAsync(o->offsets->ptr, 0, 4, ctx->stream));
    }
    const Ctrl *h = fetch_ctrl(ctx);
    o->null_count = o->validity ? static_cast<int64_t>(n - h->valid_pop[0]) : 0;
    if (o->null_count == 0) o->validity.reset();
    return o.release();
}
// `StringColumn <op> Literal`: evaluated into a truth bitmap up front (null policy folded in exactly as
// lower_term does for the fixed-width types); the term then reads that bitmap with RV_IS_TRUE.
rv_dcolumn *string_term_mask(rv_ctx *ctx, const rv_dcolumn *col, const rv_term &t, rv_null_policy policy) {
    require(t.op >= RV_EQ && t.op <= RV_GE, RV_ERR_INVALID_ARG, "unknown compare operator on a String column");
    const uint64_t n = col->length;
    auto m = std::make_unique<rv_dcolumn>();
    m->dtype = RV_BOOLEAN;
    m->length = n;
    m->null_count = 0;
    m->values = pool_alloc(ctx, std::max<size_t>(bitmap_words_bytes(n) + 8, 16));
    RV_HIP(hipMemsetAsync(m->values->ptr, 0, std::max<size_t>(bitmap_words_bytes(n) + 8, 16), ctx->stream));
    if (n == 0) return m.release();
    const bool lit_null = t.lit_type == RV_NULL, least = policy == RV_NULL_IS_LEAST;
    rvk::StrCompare p{};
    p.offsets = static_cast<const int32_t *>(col->offsets->ptr);
    p.data = static_cast<const uint8_t *>(col->values->ptr);
    p.validity = col->validity ? static_cast<const uint8_t *>(col->validity->ptr) : nullptr;
    p.offset = col->offset;
    p.n = n;
    p.out_words = static_cast<uint64_t *>(m->values->ptr);
    // same folding as lower_term (series.rs:100-117): Null == Null, Null < everything, cross-type -> None
    if (lit_null) p.null_v = least && (t.op == RV_EQ || t.op == RV_LE || t.op == RV_GE);
    else p.null_v = least && (t.op == RV_LT || t.op == RV_LE || t.op == RV_NE);
    DevBufRef lit;
    if (lit_null) {
        p.op = -1;
        p.const_v = (t.op == RV_GT || t.op == RV_GE || t.op == RV_NE);
    } else if (t.lit_type != RV_STRING) {
        p.op = -1;
        p.const_v = (t.op == RV_NE);
    } else {
        require(t.lit.s.ptr || t.lit.s.len == 0, RV_ERR_INVALID_ARG, "String literal is NULL");
        require(t.lit.s.len <= 0x7FFFFFFFull, RV_ERR_INVALID_ARG, "String literal too long");
        p.op = static_cast<int32_t>(t.op);
        p.lit_len = static_cast<uint32_t>(t.lit.s.len);
        lit = pool_alloc(ctx, std::max<size_t>(p.lit_len, 16));
        if (p.lit_len) RV_HIP(hipMemcpyAsync(lit->ptr, t.lit.s.ptr, p.lit_len, hipMemcpyHostToDevice, ctx->stream));
        p.lit = static_cast<const uint8_t *>(lit->ptr);
    }
    hipLaunchKernelGGL(rvk::str_compare_mask, dim3(static_cast<uint32_t>((n + 255) / 256)), dim3(256), 0, ctx->stream, p);
    RV_HIP(hipGetLastError());
    RV_HIP(hipStreamSynchronize(ctx->stream));  // the literal is borrowed for the call; its buffer goes back to the pool
    return m.release();
}

// ---- predicate normalisation -------------------------------------------------------------------------------
// What the kernels take is a flat literal list (+ ExprInfo when there is an OR / NOT).  normalize_predicate turns an
// rv_predicate into that:
//   * rv_predicate::expr (postfix AND / OR / NOT over the terms) -> conjunctive normal form of the expression or of
//     its negation, whichever is smaller (a disjunction of conjunctions is small as the negation of one); a pure AND of
//     terms drops back to the plain term list (the tuned kernels of BASELINE configs 2 and 3);
//   * terms on String columns -> RV_IS_TRUE terms on freshly evaluated truth bitmaps appended to the column list
//     (with the column's validity attached when nulls propagate strictly through an expression);
//   * AND-only predicates over more Boolean / String columns than one pass reads (kMaxBoolCols): all Boolean terms
//     folded into ONE truth bitmap (bool_fold_kernel);
//   * what still does not fit one pass (more than kMaxTerms literals, too many predicate columns) is COMPOSED the way
//     the reference composes it: every term a BooleanArray (rv_compare_term / the eager mask), AND / OR / NOT the
//     BooleanArray operators (boolop_kernel, boolean.rs:120-165), and the result one RV_IS_TRUE term.
struct Normalized {
    std::vector<const rv_dcolumn *> cols;
    std::vector<rv_term> terms;
    std::vector<std::unique_ptr<rv_dcolumn>> masks;  // temporaries the rewritten terms read
    ExprInfo ex;
    bool has_ex = false;
    const ExprInfo *expr() const { return has_ex ? &ex : nullptr; }
};

struct ExprNode {
    int kind;  // 0 term, 1 and, 2 or, 3 not
    int a, b;
    uint32_t term;
};
struct Literal {
    uint32_t term;
    bool neg;
    bool operator==(const Literal &o) const { return term == o.term && neg == o.neg; }
};
using Clause = std::vector<Literal>;

// postfix program -> tree (nodes in evaluation order, the root last)
std::vector<ExprNode> parse_expression(const uint8_t *expr, uint32_t n_expr, uint32_t n_terms) {
    require(n_expr >= 1 && n_expr <= 255, RV_ERR_INVALID_ARG, "predicate expression: 1..255 postfix entries");
    std::vector<ExprNode> nodes;
    std::vector<int> stack;
    for (uint32_t i = 0; i < n_expr; ++i) {
        const uint8_t op = expr[i];
        if (op < 0x80) {
            require(op < n_terms, RV_ERR_INVALID_ARG, fmt("predicate expression: entry %u pushes term %u of %u", i, op, n_terms));
            nodes.push_back(ExprNode{0, -1, -1, op});
        } else if (op == RV_EXPR_NOT) {
            require(!stack.empty(), RV_ERR_INVALID_ARG, fmt("predicate expression: NOT at entry %u has no operand", i));
            const int a = stack.back();
            stack.pop_back();
            nodes.push_back(ExprNode{3, a, -1, 0});
        } else {
            require(op == RV_EXPR_AND || op == RV_EXPR_OR, RV_ERR_INVALID_ARG, fmt("predicate expression: unknown entry 0x%02x", op));
            require(stack.size() >= 2, RV_ERR_INVALID_ARG, fmt("predicate expression: operator at entry %u has fewer than two operands", i));
            const int b = stack.back();
            stack.pop_back();
            const int a = stack.back();
            stack.pop_back();
            nodes.push_back(ExprNode{op == RV_EXPR_AND ? 1 : 2, a, b, 0});
        }
        stack.push_back(static_cast<int>(nodes.size()) - 1);
    }
    require(stack.size() == 1, RV_ERR_INVALID_ARG, "predicate expression must leave exactly one value");
    return nodes;
}

// conjunctive normal form of node `i` (negated when neg); false when it outgrows `cap` literals
bool cnf_of(const std::vector<ExprNode> &nodes, int i, bool neg, size_t cap, std::vector<Clause> &out) {
    const ExprNode &n = nodes[i];
    if (n.kind == 0) {
        out.push_back(Clause{Literal{n.term, neg}});
        return true;
    }
    if (n.kind == 3) return cnf_of(nodes, n.a, !neg, cap, out);
    std::vector<Clause> A, B;
    if (!cnf_of(nodes, n.a, neg, cap, A) || !cnf_of(nodes, n.b, neg, cap, B)) return false;
    const bool conj = (n.kind == 1) != neg;  // De Morgan: NOT(a AND b) = NOT a OR NOT b
    if (conj) {
        out = std::move(A);
        for (auto &c : B)
            if (std::find(out.begin(), out.end(), c) == out.end()) out.push_back(std::move(c));
    } else {  // OR distributes over the clauses of both sides
        size_t lits = 0;
        for (const Clause &ca : A)
            for (const Clause &cb : B) {
                Clause c = ca;
                bool tautology = false;
                for (const Literal &l : cb) {
                    if (std::find(c.begin(), c.end(), Literal{l.term, !l.neg}) != c.end()) tautology = true;
                    if (std::find(c.begin(), c.end(), l) == c.end()) c.push_back(l);
                }
                if (tautology || std::find(out.begin(), out.end(), c) != out.end()) continue;  // t OR NOT t
                lits += c.size();
                if (lits > 4 * cap) return false;
                out.push_back(std::move(c));
            }
    }
    size_t lits = 0;
    for (auto &c : out) lits += c.size();
    return lits <= 4 * cap;  // generous while composing; the caller applies the real cap to the final form
}
size_t literal_count(const std::vector<Clause> &f) {
    size_t n = 0;
    for (auto &c : f) n += c.size();
    return n;
}

}  // namespace
extern "C" {
static void bool_op(rv_ctx *ctx, int kind, const rv_dcolumn *a, const rv_dcolumn *b, rv_dcolumn **out);  // defined with rv_boolean_*
}
namespace {

// copy of a column's validity bits re-based to bit 0 (attached to a String truth bitmap when nulls propagate strictly)
DevBufRef rebased_validity(rv_ctx *ctx, const rv_dcolumn *col) {
    const uint64_t n = col->length;
    DevBufRef v = pool_alloc(ctx, std::max<size_t>(bitmap_words_bytes(n) + 8, 16));
    if (n) {
        hipLaunchKernelGGL(rvk::copy_bits_kernel, dim3(grid_for_words(ctx, (n + 63) / 64, 256)), dim3(256), 0, ctx->stream,
                           static_cast<const uint8_t *>(col->validity->ptr), static_cast<uint64_t>(col->validity->bytes), col->offset, n,
                           static_cast<uint64_t *>(v->ptr));
        RV_HIP(hipGetLastError());
    }
    return v;
}

// The reference's own composition, on the device: every term a BooleanArray, the expression the BooleanArray
// operators, the result the predicate of RecordBatch::filter.  Used when one pass cannot hold the predicate.
std::unique_ptr<rv_dcolumn> compose_predicate(rv_ctx *ctx, const rv_dcolumn *const *cols, uint32_t ncols, const rv_term *terms,
                                              rv_null_policy policy, const std::vector<ExprNode> &nodes) {
    std::vector<std::unique_ptr<rv_dcolumn>> val(nodes.size());
    auto term_array = [&](const rv_term &t) -> std::unique_ptr<rv_dcolumn> {
        require(t.column < ncols, RV_ERR_INVALID_ARG, fmt("term references column %u of %u", t.column, ncols));
        const rv_dcolumn *col = cols[t.column];
        rv_dcolumn *o = nullptr;
        if (policy == RV_NULL_DROPS) {  // nullable: null where the cell is null
            const rv_status st = rv_compare_term(ctx, col, &t, &o);
            if (st != RV_OK) throw Error(st, last_error());
        } else if (col->dtype == RV_STRING) {  // eager mask (plan.rs:112-130): a definite bool per row
            o = string_term_mask(ctx, col, t, policy);
        } else {
            rv_term one = t;
            one.column = 0;
            rv_dcolumn *none = nullptr;
            run_fused_pass(ctx, &col, 1, &one, 1, policy, nullptr, 0, &none, &o);
        }
        return std::unique_ptr<rv_dcolumn>(o);
    };
    for (size_t i = 0; i < nodes.size(); ++i) {
        const ExprNode &n = nodes[i];
        rv_dcolumn *o = nullptr;
        if (n.kind == 0) {
            val[i] = term_array(terms[n.term]);
            continue;
        }
        if (n.kind == 3) bool_op(ctx, 2, val[n.a].get(), nullptr, &o);
        else bool_op(ctx, n.kind == 1 ? 0 : 1, val[n.a].get(), val[n.b].get(), &o);
        val[i].reset(o);
        // operands may be shared by several parents in principle; a postfix program uses each value once
        val[n.a].reset();
        if (n.kind != 3) val[n.b].reset();
    }
    return std::move(val.back());
}

void normalize_predicate(rv_ctx *ctx, const rv_dcolumn *const *cols, uint32_t ncols, const rv_predicate *pred, Normalized &out) {
    const rv_term *terms = pred->terms;
    const uint32_t nterms = pred->n_terms;
    const rv_null_policy policy = pred->nulls;
    require(nterms >= 1 && nterms <= static_cast<uint32_t>(rvk::kMaxTerms), RV_ERR_UNSUPPORTED,
            fmt("predicate needs 1..%d terms, got %u", rvk::kMaxTerms, nterms));
    for (uint32_t t = 0; t < nterms; ++t) {
        require(terms[t].column < ncols, RV_ERR_INVALID_ARG, fmt("term %u references column %u of %u", t, terms[t].column, ncols));
        const rv_dtype dt = cols[terms[t].column]->dtype;
        require(is_value_type(dt) || dt == RV_BOOLEAN || dt == RV_STRING, RV_ERR_UNSUPPORTED,
                "predicate columns must be Int64, Float64, Boolean or String on the device path");
    }
    out.cols.assign(cols, cols + ncols);

    // ---- the expression: plain AND, conjunctive normal form, or too large for one pass -------------------------
    std::vector<ExprNode> nodes;
    std::vector<Clause> form;  // empty when the predicate is the AND of `and_terms`
    std::vector<uint32_t> and_terms;
    bool negate_result = false, compose = false;
    if (pred->expr == nullptr) {
        for (uint32_t t = 0; t < nterms; ++t) and_terms.push_back(t);
    } else {
        nodes = parse_expression(pred->expr, pred->n_expr, nterms);
        std::vector<Clause> pos, negf;
        const size_t cap = static_cast<size_t>(rvk::kMaxTerms);
        const bool okp = cnf_of(nodes, static_cast<int>(nodes.size()) - 1, false, cap, pos) && literal_count(pos) <= cap;
        const bool okn = cnf_of(nodes, static_cast<int>(nodes.size()) - 1, true, cap, negf) && literal_count(negf) <= cap;
        bool pure_and = okp && !pos.empty();
        for (auto &c : pos) pure_and = pure_and && c.size() == 1 && !c[0].neg;
        // Simplification may have dropped every literal of a column (`a AND (NOT a OR b OR NOT b)` is `a`), but under
        // RV_NULL_DROPS the nulls of every column the expression READS still drop the row (BooleanArray::and / or / not
        // propagate them, boolean.rs:120-165): the plain AND only stands in when its terms cover those columns.
        if (pure_and && policy == RV_NULL_DROPS)
            for (const ExprNode &n : nodes) {
                if (n.kind != 0 || !cols[terms[n.term].column]->validity) continue;
                bool covered = false;
                for (auto &c : pos) covered = covered || terms[c[0].term].column == terms[n.term].column;
                pure_and = pure_and && covered;
            }
        if (pure_and) {
            for (auto &c : pos) and_terms.push_back(c[0].term);
        } else if (okp && (!okn || literal_count(pos) <= literal_count(negf))) {
            form = std::move(pos);
        } else if (okn) {
            form = std::move(negf);
            negate_result = true;
        } else {
            compose = true;
        }
        if (!pure_and && !compose && form.empty()) {
            // every clause was a tautology: the expression is constantly true (false when negated) wherever it is
            // not null -- one always-true literal keeps the kernels' term list non-empty
            const uint32_t any = nodes.front().term;  // a postfix program starts with a term
            form.push_back(Clause{Literal{any, false}, Literal{any, true}});
        }
    }
    const bool is_expr = !form.empty();
    // the columns the ORIGINAL expression reads (simplification may have dropped literals, never null propagation)
    std::vector<uint32_t> read_cols;
    if (is_expr || compose)
        for (const ExprNode &n : nodes)
            if (n.kind == 0 && std::find(read_cols.begin(), read_cols.end(), terms[n.term].column) == read_cols.end())
                read_cols.push_back(terms[n.term].column);

    // ---- column budget of one pass ---------------------------------------------------------------------------
    std::vector<uint32_t> used_terms = and_terms;
    if (is_expr)
        for (auto &c : form)
            for (auto &l : c)
                if (std::find(used_terms.begin(), used_terms.end(), l.term) == used_terms.end()) used_terms.push_back(l.term);
    std::vector<uint32_t> value_cols, bool_cols;
    size_t string_terms = 0;
    auto note = [](std::vector<uint32_t> &v, uint32_t c) {
        if (std::find(v.begin(), v.end(), c) == v.end()) v.push_back(c);
    };
    for (uint32_t t : used_terms) {
        const uint32_t c = terms[t].column;
        if (is_value_type(cols[c]->dtype)) note(value_cols, c);
        else if (cols[c]->dtype == RV_BOOLEAN) note(bool_cols, c);
        else ++string_terms;  // every String term gets its own truth bitmap
    }
    if (is_expr && pred->nulls == RV_NULL_DROPS)
        for (uint32_t c : read_cols)
            if (cols[c]->dtype == RV_BOOLEAN && cols[c]->validity) note(bool_cols, c);
    const bool too_many_values = value_cols.size() > static_cast<size_t>(rvk::kMaxValueCols);
    size_t string_nulls_only = 0;  // nullable String columns read for their nulls only: one Boolean slot each
    if (is_expr && pred->nulls == RV_NULL_DROPS)
        for (uint32_t c : read_cols) {
            if (cols[c]->dtype != RV_STRING || !cols[c]->validity) continue;
            bool used = false;
            for (uint32_t t : used_terms) used = used || terms[t].column == c;
            if (!used) ++string_nulls_only;
        }
    const bool too_many_bools = bool_cols.size() + string_terms + string_nulls_only > static_cast<size_t>(rvk::kMaxBoolCols);
    if (!compose && (too_many_values || (is_expr && too_many_bools))) {
        compose = true;
        if (nodes.empty()) {  // the AND of the terms as a tree
            for (uint32_t t = 0; t < nterms; ++t) {
                nodes.push_back(ExprNode{0, -1, -1, t});
                if (t) {
                    const int b = static_cast<int>(nodes.size()) - 1, a = t == 1 ? 0 : b - 1;
                    nodes.push_back(ExprNode{1, a, b, 0});
                }
            }
        }
    }
    if (compose) {
        out.masks.emplace_back(compose_predicate(ctx, cols, ncols, terms, policy, nodes));
        rv_term r{};
        r.column = static_cast<uint32_t>(out.cols.size());
        r.op = RV_IS_TRUE;
        out.cols.push_back(out.masks.back().get());
        out.terms.assign(1, r);
        return;
    }

    // ---- String terms -> truth bitmaps ---------------------------------------------------------------------------
    const bool strict = is_expr && policy == RV_NULL_DROPS;
    std::vector<rv_term> rewritten(terms, terms + nterms);
    for (uint32_t t : used_terms) {
        const rv_dcolumn *c = cols[terms[t].column];
        if (c->dtype != RV_STRING) continue;
        out.masks.emplace_back(string_term_mask(ctx, c, terms[t], policy));
        if (strict && c->validity) {  // its nulls have to drop the row even under a NOT: a nullable BooleanArray
            out.masks.back()->validity = rebased_validity(ctx, c);
            out.masks.back()->null_count = -1;
        }
        rv_term r{};
        r.column = static_cast<uint32_t>(out.cols.size());
        r.op = RV_IS_TRUE;
        out.cols.push_back(out.masks.back().get());
        rewritten[t] = r;
    }
    if (is_expr) {
        for (auto &c : form)
            for (size_t i = 0; i < c.size(); ++i) {
                out.terms.push_back(rewritten[c[i].term]);
                out.ex.negate.push_back(c[i].neg);
                out.ex.group_end.push_back(i + 1 == c.size());
            }
        out.ex.negate_result = negate_result;
        out.ex.strict = strict;
        if (strict)
            for (uint32_t c : read_cols) out.ex.strict_cols.push_back(cols[c]->dtype == RV_STRING ? UINT32_MAX : c);
        // a String column's nulls travel with its truth bitmap (validity attached above); a String column whose
        // literals were all simplified away still drops its null rows: a Boolean stand-in that is only its validity
        if (strict) {
            std::vector<uint32_t> covered;
            for (uint32_t t : used_terms)
                if (cols[terms[t].column]->dtype == RV_STRING && cols[terms[t].column]->validity) {
                    out.ex.strict_cols.push_back(rewritten[t].column);
                    covered.push_back(terms[t].column);
                }
            for (uint32_t c : read_cols) {
                if (cols[c]->dtype != RV_STRING || !cols[c]->validity || std::find(covered.begin(), covered.end(), c) != covered.end()) continue;
                auto m = std::make_unique<rv_dcolumn>();
                m->dtype = RV_BOOLEAN;
                m->length = cols[c]->length;
                m->validity = rebased_validity(ctx, cols[c]);
                m->values = m->validity;
                out.masks.emplace_back(std::move(m));
                out.ex.strict_cols.push_back(static_cast<uint32_t>(out.cols.size()));
                out.cols.push_back(out.masks.back().get());
            }
        }
        out.ex.strict_cols.erase(std::remove(out.ex.strict_cols.begin(), out.ex.strict_cols.end(), UINT32_MAX), out.ex.strict_cols.end());
        out.has_ex = true;
        return;
    }
    for (uint32_t t : and_terms) out.terms.push_back(rewritten[t]);
    if (!too_many_bools) return;

    // ---- AND only, more Boolean / String predicate columns than one pass reads: fold them into one truth bitmap ----
    const uint64_t n = cols[0]->length;
    rvk::BoolFold f{};
    std::vector<rv_term> kept;
    for (const rv_term &t : out.terms) {
        const rv_dcolumn *c = out.cols[t.column];
        if (c->dtype != RV_BOOLEAN) {
            kept.push_back(t);
            continue;
        }
        require(f.nterms < rvk::kMaxTerms, RV_ERR_UNSUPPORTED, "too many predicate terms");
        f.cols[f.nterms] = dev_view(c);
        f.terms[f.nterms] = lower_term(t, RV_BOOLEAN, policy, static_cast<uint32_t>(f.nterms));
        ++f.nterms;
    }
    auto m = std::make_unique<rv_dcolumn>();
    m->dtype = RV_BOOLEAN;
    m->length = n;
    m->null_count = 0;
    m->values = pool_alloc(ctx, std::max<size_t>(bitmap_words_bytes(n) + 8, 16));
    RV_HIP(hipMemsetAsync(m->values->ptr, 0, std::max<size_t>(bitmap_words_bytes(n) + 8, 16), ctx->stream));
    f.n = n;
    f.out_words = static_cast<uint64_t *>(m->values->ptr);
    if (n) {
        hipLaunchKernelGGL(rvk::bool_fold_kernel, dim3(static_cast<uint32_t>(((n + 63) / 64 + 255) / 256)), dim3(256), 0, ctx->stream, f);
        RV_HIP(hipGetLastError());
    }
    rv_term r{};
    r.column = static_cast<uint32_t>(out.cols.size());
    r.op = RV_IS_TRUE;
    out.masks.emplace_back(std::move(m));
    out.cols.push_back(out.masks.back().get());
    kept.push_back(r);
    out.terms = std::move(kept);
}
}  // namespace

extern "C" {

// ---- arrays --------------------------------------------------------------------------------
rv_status rv_upload(rv_ctx *ctx, const rv_column *host, rv_dcolumn **out) {
    return guarded([&] {
        require(ctx && host && out, RV_ERR_INVALID_ARG, "rv_upload: NULL argument");
        require(is_value_type(host->dtype) || host->dtype == RV_BOOLEAN || host->dtype == RV_STRING || host->dtype == RV_NULL,
                RV_ERR_UNSUPPORTED, "rv_upload: unknown array type");
        set_device(ctx);
        const uint64_t total = host->offset + host->length;
        auto col = std::make_unique<rv_dcolumn>();
        col->dtype = host->dtype;
        col->offset = host->offset;
        col->length = host->length;
        if (host->dtype == RV_NULL) {  // NullArray (null.rs:5-66): a length, no buffers, every element null
            col->null_count = static_cast<int64_t>(host->length);
            *out = col.release();
            return;
        }
        auto put = [&](const void *src, size_t src_bytes, size_t padded) {
            DevBufRef b = pool_alloc(ctx, std::max<size_t>(padded, 16));
            if (padded > src_bytes) RV_HIP(hipMemsetAsync(static_cast<char *>(b->ptr) + (src_bytes & ~size_t(7)), 0,
                                                          std::max<size_t>(padded, 16) - (src_bytes & ~size_t(7)), ctx->stream));
            if (src_bytes) RV_HIP(hipMemcpyAsync(b->ptr, src, src_bytes, hipMemcpyHostToDevice, ctx->stream));
            return b;
        };
        if (host->dtype == RV_STRING) {
            require(host->offsets != nullptr, RV_ERR_INVALID_ARG, "rv_upload: offsets is NULL");
            require(host->values || host->data_bytes == 0, RV_ERR_INVALID_ARG, "rv_upload: values is NULL");
            check_string_offsets(host->offsets, 0, total, host->data_bytes);
            col->offsets = put(host->offsets, static_cast<size_t>(total + 1) * 4, static_cast<size_t>(total + 1) * 4 + 8);
            col->values = put(host->values, static_cast<size_t>(host->data_bytes), static_cast<size_t>(host->data_bytes) + 8);
            col->data_bytes = host->data_bytes;
        } else if (host->dtype == RV_BOOLEAN) {
            require(host->values || total == 0, RV_ERR_INVALID_ARG, "rv_upload: values is NULL");
            col->values = put(host->values, static_cast<size_t>((total + 7) / 8), bitmap_words_bytes(total) + 8);
        } else {
            require(host->values || total == 0, RV_ERR_INVALID_ARG, "rv_upload: values is NULL");
            col->values = put(host->values, static_cast<size_t>(total) * 8, static_cast<size_t>(total) * 8);
        }
        if (host->validity) col->validity = put(host->validity, static_cast<size_t>((total + 7) / 8), bitmap_words_bytes(total) + 8);
        else col->null_count = 0;
        RV_HIP(hipStreamSynchronize(ctx->stream));  // host pointers are borrowed for the call only
        *out = col.release();
    });
}

rv_status rv_wrap(rv_ctx *ctx, const rv_column *device, rv_dcolumn **out) {
    return guarded([&] {
        require(ctx && device && out, RV_ERR_INVALID_ARG, "rv_wrap: NULL argument");
        require(is_value_type(device->dtype) || device->dtype == RV_BOOLEAN, RV_ERR_UNSUPPORTED, "rv_wrap: unsupported dtype");
        require((reinterpret_cast<uintptr_t>(device->values) & 7) == 0 && (reinterpret_cast<uintptr_t>(device->validity) & 7) == 0,
                RV_ERR_INVALID_ARG, "rv_wrap: buffers must be 8-byte aligned");
        const uint64_t total = device->offset + device->length;
        auto col = std::make_unique<rv_dcolumn>();
        col->dtype = device->dtype;
        col->offset = device->offset;
        col->length = device->length;
        col->values = std::make_shared<DevBuf>();
        col->values->ptr = const_cast<void *>(device->values);
        col->values->bytes = device->dtype == RV_BOOLEAN ? static_cast<size_t>((total + 7) / 8) : static_cast<size_t>(total) * 8;
        if (device->validity) {
            col->validity = std::make_shared<DevBuf>();
            col->validity->ptr = const_cast<uint8_t *>(device->validity);
            col->validity->bytes = static_cast<size_t>((total + 7) / 8);
        } else {
            col->null_count = 0;
        }
        *out = col.release();
    });
}

rv_status rv_generate(rv_ctx *ctx, const rv_synth_spec *spec, rv_dcolumn **out) {
    return guarded([&] {
        require(ctx && spec && out, RV_ERR_INVALID_ARG, "rv_generate: NULL argument");
        require(is_value_type(spec->dtype) || spec->dtype == RV_BOOLEAN, RV_ERR_UNSUPPORTED, "rv_generate: unsupported dtype");
        require(spec->dtype != RV_INT64 || spec->modulus > 0, RV_ERR_INVALID_ARG, "rv_generate: modulus must be > 0");
        set_device(ctx);
        auto col = std::make_unique<rv_dcolumn>();
        col->dtype = spec->dtype;
        col->length = spec->length;
        col->values = pool_alloc(ctx, std::max<size_t>(elem_bytes(spec->dtype, spec->length), 16));
        if (spec->with_validity) col->validity = pool_alloc(ctx, std::max<size_t>(bitmap_words_bytes(spec->length), 16));
        else col->null_count = 0;
        if (spec->length) {
            rvk::GenParams g{};
            g.values = static_cast<uint64_t *>(col->values->ptr);
            g.validity = col->validity ? static_cast<uint64_t *>(col->validity->ptr) : nullptr;
            g.seed = spec->seed;
            g.first_row = spec->first_row;
            g.length = spec->length;
            g.modulus = spec->modulus;
            g.validity_seed = spec->validity_seed;
            g.true_percent = spec->true_percent;
            g.null_percent = spec->null_percent;
            g.dtype = static_cast<int32_t>(spec->dtype);
            hipLaunchKernelGGL(rvk::generate_kernel, dim3(grid_for_words(ctx, spec->length, 256)), dim3(256), 0, ctx->stream, g);
            RV_HIP(hipGetLastError());
        }
        *out = col.release();
    });
}

rv_status rv_free(rv_ctx *ctx, rv_dcolumn *col) {
    return guarded([&] {
        (void)ctx;
        // No synchronisation: the buffers go back to the context's pool and every later user runs on the context's
        // stream (or waits for it: the chunk uploads of rv_filter_project_host), i.e. after the work queued so far.
        delete col;
    });
}

rv_status rv_slice(rv_ctx *ctx, const rv_dcolumn *col, uint64_t offset, uint64_t length, rv_dcolumn **out) {
    return guarded([&] {
        require(ctx && col && out, RV_ERR_INVALID_ARG, "rv_slice: NULL argument");
        require(offset + length <= col->length, RV_ERR_OUT_OF_BOUNDS, "Slice out of bounds");  // boolean.rs:209
        auto s = std::make_unique<rv_dcolumn>(*col);
        s->offset = col->offset + offset;
        s->length = length;
        s->null_count = col->dtype == RV_NULL ? static_cast<int64_t>(length) : (col->validity ? -1 : 0);
        *out = s.release();
    });
}

rv_status rv_slice_known(rv_ctx *ctx, const rv_dcolumn *col, uint64_t offset, uint64_t length, int64_t null_count, rv_dcolumn **out) {
    return guarded([&] {
        require(ctx && col && out, RV_ERR_INVALID_ARG, "rv_slice_known: NULL argument");
        require(offset + length <= col->length, RV_ERR_OUT_OF_BOUNDS, "Slice out of bounds");
        require(null_count >= 0 && static_cast<uint64_t>(null_count) <= length, RV_ERR_INVALID_ARG, "rv_slice_known: null count out of range");
        auto s = std::make_unique<rv_dcolumn>(*col);
        s->offset = col->offset + offset;
        s->length = length;
        if (col->dtype == RV_NULL) {
            s->null_count = static_cast<int64_t>(length);
        } else {
            s->null_count = col->validity ? null_count : 0;
            if (s->null_count == 0) s->validity.reset();  // the builder drops a bitmap without nulls (primitive.rs:179-185)
        }
        *out = s.release();
    });
}

rv_status rv_fill_nulls(rv_ctx *ctx, const rv_dcolumn *col, rv_dcolumn **out) {
    return guarded([&] {
        require(ctx && col && out, RV_ERR_INVALID_ARG, "rv_fill_nulls: NULL argument");
        if (col->dtype == RV_STRING || col->dtype == RV_NULL || !col->validity) {  // nothing to fill: a shared view
            *out = new rv_dcolumn(*col);
            return;
        }
        set_device(ctx);
        const uint64_t n = col->length;
        auto o = std::make_unique<rv_dcolumn>();
        o->dtype = col->dtype;
        o->length = n;
        o->null_count = 0;
        o->values = pool_alloc(ctx, std::max<size_t>(elem_bytes(col->dtype, n) + 8, 16));
        rvk::FillNullsParams p{};
        p.col = dev_view(col);
        p.n = n;
        p.out = static_cast<uint64_t *>(o->values->ptr);
        if (n) {
            const uint64_t items = col->dtype == RV_BOOLEAN ? (n + 63) / 64 : n;
            hipLaunchKernelGGL(rvk::fill_nulls_kernel, dim3(static_cast<uint32_t>((items + 255) / 256)), dim3(256), 0, ctx->stream, p);
            RV_HIP(hipGetLastError());
        }
        *out = o.release();
    });
}

rv_status rv_null_count(rv_ctx *ctx, const rv_dcolumn *col, uint64_t *out) {
    return guarded([&] {
        require(ctx && col && out, RV_ERR_INVALID_ARG, "rv_null_count: NULL argument");
        if (col->dtype == RV_NULL) {
            *out = col->length;
            return;
        }
        if (col->null_count >= 0) {
            *out = static_cast<uint64_t>(col->null_count);
            return;
        }
        set_device(ctx);
        Ctrl *ctrl = prepare_ctrl(ctx, 0);
        rvk::PopParams p{};
        p.values = nullptr;
        p.validity = static_cast<const uint8_t *>(col->validity->ptr);
        p.validity_bytes = col->validity->bytes;
        p.offset = col->offset;
        p.n = col->length;
        p.out = striped(ctx, &ctrl->pops[0]), (void)striped(ctx, &ctrl->pops[1]), (void)striped(ctx, &ctrl->pops[2]);
        if (col->length) {
            hipLaunchKernelGGL(rvk::popcount_kernel, dim3(grid_for_words(ctx, (col->length + 63) / 64, 256)), dim3(256), 0, ctx->stream, p);
            RV_HIP(hipGetLastError());
        }
        const Ctrl *h = fetch_ctrl(ctx);
        const_cast<rv_dcolumn *>(col)->null_count = static_cast<int64_t>(col->length - h->pops[2]);
        *out = static_cast<uint64_t>(col->null_count);
    });
}

rv_status rv_column_info_get(rv_ctx *ctx, const rv_dcolumn *col, rv_column_info *out) {
    return guarded([&] {
        require(ctx && col && out, RV_ERR_INVALID_ARG, "rv_column_info_get: NULL argument");
        out->dtype = col->dtype;
        out->length = col->length;
        out->offset = col->offset;
        out->has_validity = col->validity ? 1 : 0;
        out->null_count = col->null_count;
        out->data_bytes = 0;
        if (col->dtype == RV_STRING) {  // bytes of the logical elements: offsets[offset + length] - offsets[offset]
            set_device(ctx);
            int32_t ends[2] = {0, 0};
            const int32_t *o = static_cast<const int32_t *>(col->offsets->ptr);
            RV_HIP(hipMemcpyAsync(&ends[0], o + col->offset, 4, hipMemcpyDeviceToHost, ctx->stream));
            RV_HIP(hipMemcpyAsync(&ends[1], o + col->offset + col->length, 4, hipMemcpyDeviceToHost, ctx->stream));
            RV_HIP(hipStreamSynchronize(ctx->stream));
            out->data_bytes = static_cast<uint64_t>(ends[1] - ends[0]);
        }
    });
}

rv_status rv_download_string(rv_ctx *ctx, const rv_dcolumn *col, int32_t *offsets, uint8_t *data, uint8_t *validity, int *has_validity) {
    return guarded([&] {
        require(ctx && col && offsets, RV_ERR_INVALID_ARG, "rv_download_string: NULL argument");
        require(col->dtype == RV_STRING, RV_ERR_TYPE_MISMATCH, "rv_download_string: not a StringArray");
        set_device(ctx);
        const uint64_t n = col->length;
        RV_HIP(hipMemcpyAsync(offsets, static_cast<const int32_t *>(col->offsets->ptr) + col->offset, (n + 1) * 4, hipMemcpyDeviceToHost, ctx->stream));
        RV_HIP(hipStreamSynchronize(ctx->stream));
        const int32_t first = offsets[0];
        const size_t bytes = static_cast<size_t>(offsets[n] - first);
        for (uint64_t i = 0; i <= n; ++i) offsets[i] -= first;
        if (bytes) {
            require(data != nullptr, RV_ERR_INVALID_ARG, "rv_download_string: data is NULL");
            RV_HIP(hipMemcpyAsync(data, static_cast<const uint8_t *>(col->values->ptr) + first, bytes, hipMemcpyDeviceToHost, ctx->stream));
            RV_HIP(hipStreamSynchronize(ctx->stream));
        }
        if (has_validity) *has_validity = col->validity ? 1 : 0;
        if (validity && col->validity && n) {
            DevBufRef tmp = pool_alloc(ctx, bitmap_words_bytes(n));
            hipLaunchKernelGGL(rvk::copy_bits_kernel, dim3(grid_for_words(ctx, (n + 63) / 64, 256)), dim3(256), 0, ctx->stream,
                               static_cast<const uint8_t *>(col->validity->ptr), static_cast<uint64_t>(col->validity->bytes), col->offset, n,
                               static_cast<uint64_t *>(tmp->ptr));
            RV_HIP(hipGetLastError());
            RV_HIP(hipMemcpyAsync(validity, tmp->ptr, static_cast<size_t>((n + 7) / 8), hipMemcpyDeviceToHost, ctx->stream));
            RV_HIP(hipStreamSynchronize(ctx->stream));
        }
    });
}

rv_status rv_device_ptrs(rv_ctx *ctx, const rv_dcolumn *col, rv_column *out) {
    return guarded([&] {
        require(ctx && col && out, RV_ERR_INVALID_ARG, "rv_device_ptrs: NULL argument");
        out->dtype = col->dtype;
        out->values = col->values ? col->values->ptr : nullptr;
        out->validity = col->validity ? static_cast<const uint8_t *>(col->validity->ptr) : nullptr;
        out->offset = col->offset;
        out->length = col->length;
    });
}

rv_status rv_download(rv_ctx *ctx, const rv_dcolumn *col, void *values, uint8_t *validity, int *has_validity) {
    return guarded([&] {
        require(ctx && col, RV_ERR_INVALID_ARG, "rv_download: NULL argument");
        require(col->dtype != RV_STRING, RV_ERR_TYPE_MISMATCH, "rv_download: StringArray needs rv_download_string");
        if (col->dtype == RV_NULL) {  // nothing to copy: length and null count say it all
            if (has_validity) *has_validity = 0;
            return;
        }
        set_device(ctx);
        if (has_validity) *has_validity = col->validity ? 1 : 0;
        const uint64_t n = col->length;
        DevBufRef tmp;
        auto download_bits = [&](const DevBufRef &src, uint8_t *dst) {
            if (n == 0) return;
            if (!tmp) tmp = pool_alloc(ctx, bitmap_words_bytes(n));
            hipLaunchKernelGGL(rvk::copy_bits_kernel, dim3(grid_for_words(ctx, (n + 63) / 64, 256)), dim3(256), 0, ctx->stream,
                               static_cast<const uint8_t *>(src->ptr), static_cast<uint64_t>(src->bytes), col->offset, n,
                               static_cast<uint64_t *>(tmp->ptr));
            RV_HIP(hipGetLastError());
            RV_HIP(hipMemcpyAsync(dst, tmp->ptr, static_cast<size_t>((n + 7) / 8), hipMemcpyDeviceToHost, ctx->stream));
            RV_HIP(hipStreamSynchronize(ctx->stream));
        };
        if (values && n) {
            if (col->dtype == RV_BOOLEAN) download_bits(col->values, static_cast<uint8_t *>(values));
            else {
                RV_HIP(hipMemcpyAsync(values, static_cast<const char *>(col->values->ptr) + col->offset * 8, static_cast<size_t>(n) * 8,
                                      hipMemcpyDeviceToHost, ctx->stream));
                RV_HIP(hipStreamSynchronize(ctx->stream));
            }
        }
        if (validity && col->validity) download_bits(col->validity, validity);
    });
}

// ---- predicate ---------------------------------------------------------------------------------
static void check_batch(const rv_dcolumn *const *cols, uint32_t ncols) {
    require(cols != nullptr || ncols == 0, RV_ERR_INVALID_ARG, "cols is NULL");
    for (uint32_t i = 0; i < ncols; ++i) {
        require(cols[i] != nullptr, RV_ERR_INVALID_ARG, fmt("column %u is NULL", i));
        // RecordBatch::try_new (record_batch.rs:31-40)
        require(cols[i]->length == cols[0]->length, RV_ERR_LENGTH_MISMATCH,
                fmt("Column %u has length %llu but expected %llu", i, static_cast<unsigned long long>(cols[i]->length),
                    static_cast<unsigned long long>(cols[0]->length)));
    }
}

rv_status rv_eval_predicate(rv_ctx *ctx, const rv_dcolumn *const *cols, uint32_t ncols, const rv_predicate *pred,
                            rv_dcolumn **out_selection, uint64_t *out_count) {
    return guarded([&] {
        require(ctx && pred && pred->terms, RV_ERR_INVALID_ARG, "rv_eval_predicate: NULL argument");
        require(ncols >= 1, RV_ERR_INVALID_ARG, "rv_eval_predicate: no columns");
        check_batch(cols, ncols);
        set_device(ctx);
        rv_dcolumn *sel = nullptr;
        Normalized nz;
        normalize_predicate(ctx, cols, ncols, pred, nz);
        const uint64_t rows = run_fused_pass(ctx, nz.cols.data(), static_cast<uint32_t>(nz.cols.size()), nz.terms.data(),
                                             static_cast<uint32_t>(nz.terms.size()), pred->nulls, nullptr, 0, nullptr,
                                             out_selection ? &sel : nullptr, nz.expr());
        if (out_selection) *out_selection = sel;
        if (out_count) *out_count = rows;
    });
}

rv_status rv_compare(rv_ctx *ctx, const rv_dcolumn *col, rv_cmp op, rv_dtype lit_type, int64_t lit_i, double lit_f,
                     rv_dcolumn **out_bool) {
    return guarded([&] {
        require(ctx && col && out_bool, RV_ERR_INVALID_ARG, "rv_compare: NULL argument");
        require(is_value_type(col->dtype) || col->dtype == RV_BOOLEAN, RV_ERR_UNSUPPORTED, "rv_compare: unsupported dtype");
        set_device(ctx);
        rv_term t{};
        t.op = op;
        t.lit_type = lit_type;
        if (lit_type == RV_FLOAT64) t.lit.f = lit_f;
        else t.lit.i = lit_i;
        rvk::CompareParams p{};
        p.col = dev_view(col);
        p.term = lower_term(t, col->dtype, RV_NULL_DROPS);
        p.n = col->length;
        auto o = std::make_unique<rv_dcolumn>();
        o->dtype = RV_BOOLEAN;
        o->length = col->length;
        const size_t wb = std::max<size_t>(bitmap_words_bytes(col->length), 16);
        o->values = pool_alloc(ctx, wb);
        if (col->validity) o->validity = pool_alloc(ctx, wb);
        Ctrl *ctrl = prepare_ctrl(ctx, 0);
        p.out_values = static_cast<uint64_t *>(o->values->ptr);
        p.out_validity = o->validity ? static_cast<uint64_t *>(o->validity->ptr) : nullptr;
        p.out_valid_pop = striped(ctx, &ctrl->valid_pop[0]);
        if (col->length) {
            hipLaunchKernelGGL(rvk::compare_kernel, dim3(grid_for_words(ctx, col->length, 256)), dim3(256), 0, ctx->stream, p);
            RV_HIP(hipGetLastError());
        }
        const Ctrl *h = fetch_ctrl(ctx);
        o->null_count = o->validity ? static_cast<int64_t>(col->length - h->valid_pop[0]) : 0;
        if (o->null_count == 0) o->validity.reset();  // BooleanArrayBuilder::finish (boolean.rs:282-286)
        *out_bool = o.release();
    });
}

rv_status rv_compare_term(rv_ctx *ctx, const rv_dcolumn *col, const rv_term *term, rv_dcolumn **out_bool) {
    if (col && term && col->dtype != RV_STRING)
        return rv_compare(ctx, col, term->op, term->lit_type, term->lit.i, term->lit_type == RV_FLOAT64 ? term->lit.f : 0.0, out_bool);
    return guarded([&] {
        require(ctx && col && term && out_bool, RV_ERR_INVALID_ARG, "rv_compare_term: NULL argument");
        set_device(ctx);
        // values: the truth of every valid cell, false under a null (BooleanArray::new, boolean.rs:29-32);
        // validity: the column's own bitmap re-based to bit 0
        std::unique_ptr<rv_dcolumn> o(string_term_mask(ctx, col, *term, RV_NULL_DROPS));
        const uint64_t n = col->length;
        if (col->validity && n) {
            o->validity = pool_alloc(ctx, std::max<size_t>(bitmap_words_bytes(n), 16));
            hipLaunchKernelGGL(rvk::copy_bits_kernel, dim3(grid_for_words(ctx, (n + 63) / 64, 256)), dim3(256), 0, ctx->stream,
                               static_cast<const uint8_t *>(col->validity->ptr), static_cast<uint64_t>(col->validity->bytes), col->offset, n,
                               static_cast<uint64_t *>(o->validity->ptr));
            RV_HIP(hipGetLastError());
            o->null_count = -1;
            uint64_t nulls = 0;
            const rv_status st = rv_null_count(ctx, o.get(), &nulls);
            if (st != RV_OK) throw Error(st, g_last_error);
            if (nulls == 0) o->validity.reset();  // BooleanArrayBuilder::finish (boolean.rs:282-286)
        }
        *out_bool = o.release();
    });
}

// ---- BooleanArray logic ---------------------------------------------------------------------------
static void bool_op(rv_ctx *ctx, int kind, const rv_dcolumn *a, const rv_dcolumn *b, rv_dcolumn **out) {
    require(ctx && a && out && (kind == 2 || b), RV_ERR_INVALID_ARG, "boolean op: NULL argument");
    require(a->dtype == RV_BOOLEAN && (kind == 2 || b->dtype == RV_BOOLEAN), RV_ERR_TYPE_MISMATCH, "boolean op: operands must be BooleanArray");
    if (kind != 2) require(a->length == b->length, RV_ERR_LENGTH_MISMATCH, "Array lengths must match for logical operations");  // boolean.rs:121-123
    set_device(ctx);
    rvk::BoolOpParams p{};
    p.a = dev_view(a);
    if (kind != 2) p.b = dev_view(b);
    p.n = a->length;
    p.kind = kind;
    auto o = std::make_unique<rv_dcolumn>();
    o->dtype = RV_BOOLEAN;
    o->length = a->length;
    const size_t wb = std::max<size_t>(bitmap_words_bytes(a->length), 16);
    o->values = pool_alloc(ctx, wb);
    const bool any_validity = a->validity || (kind != 2 && b->validity);
    if (any_validity) o->validity = pool_alloc(ctx, wb);
    Ctrl *ctrl = prepare_ctrl(ctx, 0);
    p.out_values = static_cast<uint64_t *>(o->values->ptr);
    p.out_validity = o->validity ? static_cast<uint64_t *>(o->validity->ptr) : nullptr;
    p.out_valid_pop = striped(ctx, &ctrl->valid_pop[0]);
    if (a->length) {
        hipLaunchKernelGGL(rvk::boolop_kernel, dim3(grid_for_words(ctx, (a->length + 63) / 64, 256)), dim3(256), 0, ctx->stream, p);
        RV_HIP(hipGetLastError());
    }
    const Ctrl *h = fetch_ctrl(ctx);
    o->null_count = o->validity ? static_cast<int64_t>(a->length - h->valid_pop[0]) : 0;
    if (o->null_count == 0) o->validity.reset();
    *out = o.release();
}
rv_status rv_boolean_and(rv_ctx *ctx, const rv_dcolumn *a, const rv_dcolumn *b, rv_dcolumn **out) {
    return guarded([&] { bool_op(ctx, 0, a, b, out); });
}
rv_status rv_boolean_or(rv_ctx *ctx, const rv_dcolumn *a, const rv_dcolumn *b, rv_dcolumn **out) {
    return guarded([&] { bool_op(ctx, 1, a, b, out); });
}
rv_status rv_boolean_not(rv_ctx *ctx, const rv_dcolumn *a, rv_dcolumn **out) {
    return guarded([&] { bool_op(ctx, 2, a, nullptr, out); });
}
rv_status rv_boolean_count(rv_ctx *ctx, const rv_dcolumn *a, uint64_t *count_true, uint64_t *count_false) {
    return guarded([&] {
        require(ctx && a, RV_ERR_INVALID_ARG, "rv_boolean_count: NULL argument");
        require(a->dtype == RV_BOOLEAN, RV_ERR_TYPE_MISMATCH, "rv_boolean_count: not a BooleanArray");
        set_device(ctx);
        Ctrl *ctrl = prepare_ctrl(ctx, 0);
        rvk::PopParams p{};
        p.values = static_cast<const uint8_t *>(a->values->ptr);
        p.values_bytes = a->values->bytes;
        p.validity = a->validity ? static_cast<const uint8_t *>(a->validity->ptr) : nullptr;
        p.validity_bytes = a->validity ? a->validity->bytes : 0;
        p.offset = a->offset;
        p.n = a->length;
        p.out = striped(ctx, &ctrl->pops[0]), (void)striped(ctx, &ctrl->pops[1]), (void)striped(ctx, &ctrl->pops[2]);
        if (a->length) {
            hipLaunchKernelGGL(rvk::popcount_kernel, dim3(grid_for_words(ctx, (a->length + 63) / 64, 256)), dim3(256), 0, ctx->stream, p);
            RV_HIP(hipGetLastError());
        }
        const Ctrl *h = fetch_ctrl(ctx);
        if (count_true) *count_true = h->pops[0];
        if (count_false) *count_false = h->pops[1];
    });
}

// ---- RecordBatch kernels ---------------------------------------------------------------------------
// Columns are compacted in groups that fit one single-pass launch (<= 4 eight-byte columns
// and <= 4 bit streams each); every group re-reads the predicate bitmap only (1 bit/row).
// `terms` is a normalised term list (normalize_predicate): no String columns, at most kMaxBoolCols Boolean ones.
static uint64_t filter_by_groups(rv_ctx *ctx, const rv_dcolumn *const *cols, uint32_t ncols, const rv_term *terms,
                                 uint32_t nterms, rv_null_policy policy, const uint32_t *proj, uint32_t nproj,
                                 rv_dcolumn **out, rv_dcolumn **out_selection, const ExprInfo *ex = nullptr, BatchReq *req = nullptr,
                                 const AfterLaunch *after_launch = nullptr, RangeOffsets *ranges = nullptr) {
    // String, Boolean and Null projections are produced AFTER the fused pass, from the selection bitmap it
    // materialises: strings gathered by the surviving row indices, Boolean columns compacted bit-wise
    // (bits_compact_kernel; as byte-staged streams inside the fused pass they made it ~2.4x slower), Null
    // columns are just a length.  The 8-byte columns go through the fused pass.
    // ... unless the launch runs in lane form (8-byte loads), where a Boolean column rides along as a bit stream: a software
    // PEXT per 64-row word inside the pass (fused_kernel.hpp), up to kMaxBitStreams streams (values + validity each).
    // Measured (profiles/README.md): the per-lane PEXT costs the issue-bound pass more than the separate bit-compaction
    // kernel costs in traffic, so it is off by default (option "bools_in_pass" = 1 turns it on).
    int bool_streams = 0;
    bool bools_in_pass = ctx->opt_bools_in_pass != 0 && ctx->opt_vec != 2 && ex == nullptr;
    {
        std::vector<char> seen(ncols, 0);
        for (uint32_t j = 0; j < nproj && bools_in_pass; ++j) {
            const uint32_t c = proj[j];
            if (c >= ncols) break;
            if (cols[c]->dtype == RV_BOOLEAN) bool_streams += cols[c]->validity ? 2 : 1;
            else if (cols[c]->dtype == RV_STRING || cols[c]->dtype == RV_NULL) bools_in_pass = false;  // a selection bitmap is made anyway
        }
        bools_in_pass = bools_in_pass && bool_streams > 0 && bool_streams <= rvk::kMaxBitStreams;
    }
    auto post_pass = [&](uint32_t c) {
        return cols[c]->dtype == RV_STRING || cols[c]->dtype == RV_NULL || (cols[c]->dtype == RV_BOOLEAN && !bools_in_pass);
    };
    bool any_post = false;
    for (uint32_t j = 0; j < nproj; ++j) {
        require(proj[j] < ncols, RV_ERR_INVALID_ARG, fmt("projection %u references column %u of %u", j, proj[j], ncols));
        any_post |= post_pass(proj[j]);
    }
    if (any_post) {
        std::vector<uint32_t> fixed, fixed_pos;
        for (uint32_t j = 0; j < nproj; ++j)
            if (!post_pass(proj[j])) {
                fixed.push_back(proj[j]);
                fixed_pos.push_back(j);
            }
        std::vector<rv_dcolumn *> fo(fixed.size() ? fixed.size() : 1, nullptr);
        rv_dcolumn *sel = nullptr;
        uint64_t rows = 0;
        // String and Boolean columns are produced from the selection bitmap by launches queued right behind the fused pass,
        // while it is still writing that bitmap, at the pass's wave offsets (no scan over the bitmap): the lengths pass of the
        // first String column, bits_compact_kernel for up to 6 Boolean columns.  The host waits for the pass (it sizes the
        // copy launches by the survivor count) while those run, and reads the shared control block once, at the end.
        StrSelLaunch first_str;
        int first_str_j = -1;
        std::vector<uint32_t> bool_js;
        // A Boolean column that the predicate itself requires to be true (`b is true` in a plain AND: the one filter form the
        // reference's streaming planner accepts, streaming_planner.rs:139) is all true and never null among the survivors
        // (record_batch.rs:237 keeps Some(true) only): its output is rows ones, nothing to read or compact.
        std::vector<char> all_true(nproj, 0);
        for (uint32_t j = 0; j < nproj; ++j) {
            if (cols[proj[j]]->dtype == RV_STRING && first_str_j < 0) first_str_j = static_cast<int>(j);
            if (cols[proj[j]]->dtype != RV_BOOLEAN) continue;
            for (uint32_t t = 0; t < nterms && !ex; ++t)
                if (terms[t].column == proj[j] && terms[t].op == RV_IS_TRUE) all_true[j] = 1;
            if (!all_true[j]) bool_js.push_back(j);
        }
        RangeOffsets wave_ranges;
        const bool want_bools = !bool_js.empty() && bool_js.size() <= 6;
        const bool want_ranges = want_bools || first_str_j >= 0;
        std::vector<BoolCompactLaunch> bool_launches(want_bools ? bool_js.size() : 0);
        bool bools_queued = false;
        const AfterLaunch queue_post = [&](const rv_dcolumn *s) {
            if (!str_sel_eligible(s, wave_ranges)) return;  // an empty table, or a geometry whose ranges do not tile 4096 rows
            Ctrl *ctrl = prepare_ctrl(ctx, 0);
            if (first_str_j >= 0) str_sel_queue(ctx, cols[proj[first_str_j]], s, wave_ranges, ctrl, 0, first_str);
            if (want_bools) {
                for (size_t k = 0; k < bool_js.size(); ++k)
                    bool_compact_queue(ctx, cols[proj[bool_js[k]]], s, wave_ranges, ctrl, 1 + static_cast<int>(k), bool_launches[k]);
                bools_queued = true;
            }
        };
        try {
            if (req && (first_str_j >= 0 || !bool_js.empty())) req->sel_optional = false;  // columns produced from the selection bitmap
            const uint64_t reruns_before = ctx->overflow_reruns;
            // the selection bitmap: for the String / Boolean columns compacted by it (a NullArray or an all-true column is a length)
            const bool need_sel = out_selection != nullptr || first_str_j >= 0 || !bool_js.empty();
            rows = filter_by_groups(ctx, cols, ncols, terms, nterms, policy, fixed.data(), static_cast<uint32_t>(fixed.size()), fo.data(),
                                    need_sel ? &sel : nullptr, ex, req, &queue_post, want_ranges ? &wave_ranges : nullptr);
            for (size_t k = 0; k < fixed.size(); ++k) {
                out[fixed_pos[k]] = fo[k];
                fo[k] = nullptr;
            }
            if (first_str.queued || bools_queued) {
                // outputs sized by a bound that the pass overflowed (it was re-run with exact sizes): what was queued with
                // the same bound is dropped and the columns take the scan path below
                const bool usable = ctx->overflow_reruns == reruns_before;
                if (first_str.queued && usable) str_sel_copy(ctx, first_str, rows);
                const Ctrl fetched = *fetch_ctrl(ctx);  // one read-back for everything queued behind the pass
                if (first_str.queued && usable) out[first_str_j] = str_sel_result(first_str, rows, fetched);
                if (bools_queued && usable)
                    for (size_t k = 0; k < bool_js.size(); ++k) out[bool_js[k]] = bool_compact_result(bool_launches[k], rows, fetched);
            }
            DevBufRef excl;  // survivor prefix per selection word: the paths that could not be queued behind the pass
            auto need_excl = [&]() -> const DevBufRef & {
                if (!excl) excl = selection_prefix(ctx, sel, rows);
                return excl;
            };
            for (uint32_t j = 0; j < nproj; ++j) {
                const rv_dcolumn *src = cols[proj[j]];
                if (out[j]) continue;  // fixed-width columns, and what was queued behind the pass
                if (src->dtype == RV_STRING) {
                    if (str_sel_eligible(sel, wave_ranges) && rows <= wave_ranges.out_capacity) {  // further String columns: the same launches, one after the other
                        StrSelLaunch L;
                        str_sel_queue(ctx, src, sel, wave_ranges, prepare_ctrl(ctx, 0), 0, L);
                        str_sel_copy(ctx, L, rows);
                        const Ctrl fetched = *fetch_ctrl(ctx);
                        out[j] = str_sel_result(L, rows, fetched);
                    } else {
                        out[j] = gather_strings_selected(ctx, src, sel, rows, need_excl());
                    }
                } else if (src->dtype == RV_BOOLEAN && all_true[j]) {
                    auto o = std::make_unique<rv_dcolumn>();
                    o->dtype = RV_BOOLEAN;
                    o->length = rows;
                    o->null_count = 0;
                    const size_t wb = std::max<size_t>(bitmap_words_bytes(rows) + 8, 16);
                    o->values = pool_alloc(ctx, wb);
                    RV_HIP(hipMemsetAsync(o->values->ptr, 0, wb, ctx->stream));  // tail bits zero (bitmap.rs:178-188)
                    if (rows / 8) RV_HIP(hipMemsetAsync(o->values->ptr, 0xFF, rows / 8, ctx->stream));
                    if (rows % 8) RV_HIP(hipMemsetAsync(static_cast<char *>(o->values->ptr) + rows / 8, (1 << (rows % 8)) - 1, 1, ctx->stream));
                    out[j] = o.release();
                } else if (src->dtype == RV_BOOLEAN) {
                    out[j] = compact_boolean(ctx, src, sel, rows, need_excl());
                } else if (src->dtype == RV_NULL) {
                    auto o = std::make_unique<rv_dcolumn>();
                    o->dtype = RV_NULL;
                    o->length = rows;
                    o->null_count = static_cast<int64_t>(rows);
                    out[j] = o.release();
                }
            }
            RV_HIP(hipStreamSynchronize(ctx->stream));  // excl goes back to the pool
        } catch (...) {
            for (auto *d : fo) delete d;
            for (uint32_t j = 0; j < nproj; ++j) {
                delete out[j];
                out[j] = nullptr;
            }
            delete sel;
            throw;
        }
        if (out_selection) *out_selection = sel;
        else delete sel;
        return rows;
    }
    // how much of the budget do the predicate columns take?
    std::vector<char> pred_value(ncols, 0);
    int pred_vals = 0;
    for (uint32_t t = 0; t < nterms; ++t) {
        const uint32_t c = terms[t].column;
        require(c < ncols, RV_ERR_INVALID_ARG, fmt("term %u references column %u of %u", t, c, ncols));
        if (is_value_type(cols[c]->dtype) && !pred_value[c]) {
            pred_value[c] = 1;
            ++pred_vals;
        }
    }
    if (ex)  // columns read for their nulls only (their literals were simplified away) are loaded as well
        for (uint32_t c : ex->strict_cols)
            if (c < ncols && is_value_type(cols[c]->dtype) && cols[c]->validity && !pred_value[c]) {
                pred_value[c] = 1;
                ++pred_vals;
            }
    // greedy grouping of the projection list
    std::vector<std::vector<uint32_t>> groups(1);
    std::vector<std::vector<uint32_t>> group_pos(1);
    auto cost_of = [&](const std::vector<uint32_t> &g, bool with_pred, int &vals, int &bits) {
        vals = with_pred ? pred_vals : 0;
        bits = 0;
        std::vector<char> seen(ncols, 0);
        for (uint32_t c : g) {
            if (is_value_type(cols[c]->dtype)) {
                if (!(with_pred && pred_value[c] && !seen[c])) ++vals;
                seen[c] = 1;
            } else {
                bits += cols[c]->validity ? 2 : 1;
            }
        }
    };
    for (uint32_t j = 0; j < nproj; ++j) {
        require(proj[j] < ncols, RV_ERR_INVALID_ARG, fmt("projection %u references column %u of %u", j, proj[j], ncols));
        auto trial = groups.back();
        trial.push_back(proj[j]);
        int vals, bits;
        cost_of(trial, groups.size() == 1, vals, bits);
        if (vals > rvk::kMaxValueCols || bits > rvk::kMaxBitStreams) {
            groups.emplace_back();
            group_pos.emplace_back();
        }
        groups.back().push_back(proj[j]);
        group_pos.back().push_back(j);
    }
    const bool multi = groups.size() > 1;
    rv_dcolumn *sel = nullptr;
    std::vector<rv_dcolumn *> tmp(nproj ? nproj : 1, nullptr);
    uint64_t rows = 0;
    try {
        if (req && multi) req->sel_optional = false;  // later groups read the selection bitmap
        rows = run_fused_pass(ctx, cols, ncols, terms, nterms, policy, groups[0].data(), static_cast<uint32_t>(groups[0].size()),
                              tmp.data(), (multi || out_selection) ? &sel : nullptr, ex, req, after_launch, ranges);
        for (size_t k = 0; k < groups[0].size(); ++k) out[group_pos[0][k]] = tmp[k];
        for (size_t g = 1; g < groups.size(); ++g) {
            // later groups: predicate == the materialised selection bitmap
            std::vector<const rv_dcolumn *> gc;
            std::vector<uint32_t> gp;
            for (uint32_t c : groups[g]) {
                gp.push_back(static_cast<uint32_t>(gc.size()));
                gc.push_back(cols[c]);
            }
            rv_term st{};
            st.column = static_cast<uint32_t>(gc.size());
            st.op = RV_IS_TRUE;
            gc.push_back(sel);
            std::vector<rv_dcolumn *> gout(gp.size(), nullptr);
            const uint64_t r2 = run_fused_pass(ctx, gc.data(), static_cast<uint32_t>(gc.size()), &st, 1, RV_NULL_DROPS, gp.data(),
                                               static_cast<uint32_t>(gp.size()), gout.data(), nullptr);
            for (size_t k = 0; k < gout.size(); ++k) out[group_pos[g][k]] = gout[k];
            require(r2 == rows, RV_ERR_INTERNAL, "group passes disagree on the number of surviving rows");
        }
    } catch (...) {
        for (uint32_t j = 0; j < nproj; ++j) {
            delete out[j];
            out[j] = nullptr;
        }
        delete sel;
        throw;
    }
    if (out_selection) *out_selection = sel;
    else delete sel;
    return rows;
}

// rv_predicate -> normalised term list -> column groups
static uint64_t filter_query(rv_ctx *ctx, const rv_dcolumn *const *cols, uint32_t ncols, const rv_predicate *pred, const uint32_t *proj,
                             uint32_t nproj, rv_dcolumn **out, rv_dcolumn **out_selection, BatchReq *req = nullptr) {
    Normalized nz;
    normalize_predicate(ctx, cols, ncols, pred, nz);
    return filter_by_groups(ctx, nz.cols.data(), static_cast<uint32_t>(nz.cols.size()), nz.terms.data(), static_cast<uint32_t>(nz.terms.size()),
                            pred->nulls, proj, nproj, out, out_selection, nz.expr(), req);
}

rv_status rv_filter_project(rv_ctx *ctx, const rv_dcolumn *const *cols, uint32_t ncols, const rv_predicate *pred,
                            const uint32_t *proj, uint32_t nproj, rv_dcolumn **out, uint64_t *out_rows,
                            rv_dcolumn **out_selection) {
    return guarded([&] {
        require(ctx && pred && pred->terms && (out || nproj == 0) && (proj || nproj == 0), RV_ERR_INVALID_ARG,
                "rv_filter_project: NULL argument");
        require(ncols >= 1, RV_ERR_INVALID_ARG, "rv_filter_project: no columns");
        check_batch(cols, ncols);
        set_device(ctx);
        for (uint32_t j = 0; j < nproj; ++j) out[j] = nullptr;
        maybe_injected_failure(ctx);
        const uint64_t rows = filter_query(ctx, cols, ncols, pred, proj, nproj, out, out_selection);
        if (out_rows) *out_rows = rows;
    });
}

// ---- the fused pass in two halves (rv_filter_project_begin / _finish) ------------------------------------
}  // extern "C"

struct rv_pending {
    FusedLaunch launch;                 // valid when !done
    std::vector<rv_dcolumn *> outs;     // output handles (owned until finish hands them over)
    uint64_t rows = 0;
    bool done = false;                  // completed inside begin (several passes)
};

namespace {
// does the query fit ONE fused pass (no String column involved, column budget of a single launch)?
bool single_pass_shape(const rv_dcolumn *const *cols, uint32_t ncols, const rv_term *terms, uint32_t nterms, const uint32_t *proj,
                       uint32_t nproj) {
    std::vector<char> val(ncols, 0), bl(ncols, 0);
    int nvals = 0, nbools = 0, nbits = 0;
    for (uint32_t t = 0; t < nterms; ++t) {
        const uint32_t c = terms[t].column;
        if (c >= ncols) return false;
        const rv_dtype dt = cols[c]->dtype;
        if (is_value_type(dt)) {
            if (!val[c]) val[c] = 1, ++nvals;
        } else if (dt == RV_BOOLEAN) {
            if (!bl[c]) bl[c] = 1, ++nbools;
        } else {
            return false;
        }
    }
    std::vector<char> projected(ncols, 0);
    for (uint32_t j = 0; j < nproj; ++j) {
        const uint32_t c = proj[j];
        if (c >= ncols) return false;
        const rv_dtype dt = cols[c]->dtype;
        if (is_value_type(dt)) {
            if (!val[c] || projected[c]) ++nvals;  // a column projected twice takes a second slot
            val[c] = projected[c] = 1;
        } else {
            return false;  // Boolean / String / Null projections are produced after the pass
        }
    }
    return nvals <= rvk::kMaxValueCols && nbools <= rvk::kMaxBoolCols && nbits <= rvk::kMaxBitStreams && nterms <= static_cast<uint32_t>(rvk::kMaxTerms);
}
}  // namespace

extern "C" {

rv_status rv_filter_project_begin(rv_ctx *ctx, const rv_dcolumn *const *cols, uint32_t ncols, const rv_predicate *pred,
                                  const uint32_t *proj, uint32_t nproj, rv_pending **out_pending) {
    return guarded([&] {
        require(ctx && pred && pred->terms && out_pending && (proj || nproj == 0), RV_ERR_INVALID_ARG, "rv_filter_project_begin: NULL argument");
        require(ncols >= 1, RV_ERR_INVALID_ARG, "rv_filter_project_begin: no columns");
        check_batch(cols, ncols);
        set_device(ctx);
        auto pend = std::make_unique<rv_pending>();
        pend->outs.assign(nproj ? nproj : 1, nullptr);
        try {
            if (!pred->expr && single_pass_shape(cols, ncols, pred->terms, pred->n_terms, proj, nproj) && !ctx->opt_profile) {
                fused_begin(ctx, cols, ncols, pred->terms, pred->n_terms, pred->nulls, proj, nproj, pend->outs.data(), nullptr, pend->launch);
            } else {
                pend->rows = filter_query(ctx, cols, ncols, pred, proj, nproj, pend->outs.data(), nullptr);
                pend->done = true;
            }
        } catch (...) {
            for (auto *d : pend->outs) delete d;
            throw;
        }
        pend->outs.resize(nproj);
        *out_pending = pend.release();
    });
}

rv_status rv_filter_project_finish(rv_ctx *ctx, rv_pending *pending, rv_dcolumn **out, uint64_t *out_rows) {
    return guarded([&] {
        require(ctx && pending, RV_ERR_INVALID_ARG, "rv_filter_project_finish: NULL argument");
        std::unique_ptr<rv_pending> pend(pending);
        set_device(ctx);
        try {
            require(out || pend->outs.empty(), RV_ERR_INVALID_ARG, "rv_filter_project_finish: out is NULL");
            if (!pend->done) pend->rows = fused_finish(ctx, pend->launch);
        } catch (...) {
            if (!pend->done && pend->launch.launched) {  // the launch may still be running: drain before the buffers go
                (void)hipStreamSynchronize(ctx->stream);
                release_launch_ctrl(ctx, pend->launch.ctrl);
            }
            for (auto *d : pend->outs) delete d;
            throw;
        }
        for (size_t j = 0; j < pend->outs.size(); ++j) out[j] = pend->outs[j];
        if (out_rows) *out_rows = pend->rows;
    });
}

// ---- many RecordBatches, one launch (seam S1 at the reference's batch size) ------------------------------------
}  // extern "C"

namespace {
// Where the pass may drop the per-batch survivor counts: the caller's own array when the device can write it (memory from
// rv_host_alloc / rv_host_register: the counts then cross PCIe once, written by the kernel, and the host touches nothing),
// else the context's pinned staging block, copied out by finish_batch_req.
BatchReq make_batch_req(rv_ctx *ctx, uint64_t chunk_rows, uint64_t nb, uint64_t *out_rows) {
    BatchReq req;
    if (!chunk_rows || nb < 2 || !out_rows) return req;
    req.chunk_rows = chunk_rows;
    req.nb = nb;
    req.sel_optional = true;
    hipPointerAttribute_t attr{};
    if (hipPointerGetAttributes(&attr, out_rows) == hipSuccess && attr.type == hipMemoryTypeHost && attr.devicePointer) {
        req.counts = static_cast<unsigned long long *>(attr.devicePointer);
    } else {
        (void)hipGetLastError();  // an ordinary (pageable) pointer is reported as an error by some runtimes
        req.counts = static_cast<unsigned long long *>(ctx->stage(nb * 8));
    }
    return req;
}
// after the pass has been waited for: the counts are in place, or move from the staging block to the caller's array
void finish_batch_req(const BatchReq &req, uint64_t *out_rows) {
    if (static_cast<const void *>(req.counts) != static_cast<const void *>(out_rows)) {
        hipPointerAttribute_t attr{};
        const bool direct = hipPointerGetAttributes(&attr, out_rows) == hipSuccess && attr.type == hipMemoryTypeHost && attr.devicePointer == req.counts;
        if (!direct) {
            (void)hipGetLastError();
            std::memcpy(out_rows, req.counts, req.nb * 8);
        }
    }
}

// Per-batch bookkeeping of a pass that ran over several RecordBatches at once: the survivor count of every input batch out
// of the selection bitmap, the null count of every output batch out of the compacted validity bitmaps.
//   bounds        [nb + 1] first input row of every batch (general form), or empty with
//   uniform_rows  > 0: batch k is rows [k * uniform_rows, min((k + 1) * uniform_rows, sel->length)) -- no table to build or upload
// `sel` == nullptr: out_rows already holds the survivor counts (they came out of the pass itself, BatchReq); only the null
// counts are taken here.
void batch_counts(rv_ctx *ctx, const rv_dcolumn *sel, uint64_t rows, const std::vector<uint64_t> &bounds, uint64_t uniform_rows, size_t nb,
                  rv_dcolumn *const *out, uint32_t nproj, uint64_t *out_rows, int64_t *out_nulls) {
    DevBufRef d_bounds = pool_alloc(ctx, (nb + 1) * 8), d_counts = pool_alloc(ctx, nb * 8);
    std::vector<rvk::SegItem> items;
    DevBufRef d_items;
    // set bits of `words` per range of `b` -> dst (host), through segment_popcount_kernel
    auto segment_counts = [&](const uint64_t *words, const std::vector<uint64_t> &b, uint64_t *dst) {
        items.clear();
        uint64_t all_words = 0;
        for (size_t k = 0; k < nb; ++k)
            if (b[k + 1] > b[k]) all_words += ((b[k + 1] - 1) >> 6) - (b[k] >> 6) + 1;
        const uint64_t chunk_words = std::max<uint64_t>(rvk::kSegChunkWords, (all_words / (static_cast<uint64_t>(ctx->props.multiProcessorCount) * 8) + 63) & ~63ull);
        for (size_t k = 0; k < nb; ++k) {
            if (b[k + 1] <= b[k]) continue;
            const uint64_t nwords = ((b[k + 1] - 1) >> 6) - (b[k] >> 6) + 1;
            for (uint64_t c = 0; c * chunk_words < nwords; ++c) items.push_back(rvk::SegItem{static_cast<uint32_t>(k), static_cast<uint32_t>(c)});
        }
        RV_HIP(hipMemsetAsync(d_counts->ptr, 0, nb * 8, ctx->stream));
        // tables go through pinned staging: [bounds | items] in, [counts] out
        const size_t bb = (nb + 1) * 8, ib = items.size() * sizeof(rvk::SegItem);
        char *hs = static_cast<char *>(ctx->stage(std::max(bb + ib, nb * 8)));
        if (!items.empty()) {
            if (!d_items || d_items->bytes < ib) d_items = pool_alloc(ctx, ib);
            std::memcpy(hs, b.data(), bb);
            std::memcpy(hs + bb, items.data(), ib);
            RV_HIP(hipMemcpyAsync(d_bounds->ptr, hs, bb, hipMemcpyHostToDevice, ctx->stream));
            RV_HIP(hipMemcpyAsync(d_items->ptr, hs + bb, ib, hipMemcpyHostToDevice, ctx->stream));
            const dim3 grid(static_cast<uint32_t>(std::min<uint64_t>((items.size() + 3) / 4, static_cast<uint64_t>(ctx->props.multiProcessorCount) * 16)));
            hipLaunchKernelGGL(rvk::segment_popcount_kernel, grid, dim3(256), 0, ctx->stream, words, static_cast<const uint64_t *>(d_bounds->ptr),
                               static_cast<const rvk::SegItem *>(d_items->ptr), static_cast<uint64_t>(items.size()), chunk_words,
                               static_cast<unsigned long long *>(d_counts->ptr));
            RV_HIP(hipGetLastError());
        }
        RV_HIP(hipMemcpyAsync(hs, d_counts->ptr, nb * 8, hipMemcpyDeviceToHost, ctx->stream));  // stream order: after the uploads read hs
        RV_HIP(hipStreamSynchronize(ctx->stream));
        std::memcpy(dst, hs, nb * 8);
    };
    // ... per range of equal length: no tables (uniform_segment_popcount_kernel)
    auto uniform_counts = [&](const uint64_t *words, uint64_t n_bits, uint64_t *dst) {
        const dim3 grid(static_cast<uint32_t>(std::min<uint64_t>((nb + 3) / 4, static_cast<uint64_t>(ctx->props.multiProcessorCount) * 16)));
        hipLaunchKernelGGL(rvk::uniform_segment_popcount_kernel, grid, dim3(256), 0, ctx->stream, words, n_bits, uniform_rows, static_cast<uint64_t>(nb),
                           static_cast<unsigned long long *>(d_counts->ptr));
        RV_HIP(hipGetLastError());
        char *hs = static_cast<char *>(ctx->stage(nb * 8));
        RV_HIP(hipMemcpyAsync(hs, d_counts->ptr, nb * 8, hipMemcpyDeviceToHost, ctx->stream));
        RV_HIP(hipStreamSynchronize(ctx->stream));
        std::memcpy(dst, hs, nb * 8);
    };
    std::vector<uint64_t> made;  // explicit boundaries of long uniform ranges (few of them)
    const std::vector<uint64_t> *in_bounds = &bounds;
    if (sel) {
        if (uniform_rows && uniform_rows <= rvk::kSegChunkWords * 64) {
            uniform_counts(static_cast<const uint64_t *>(sel->values->ptr), sel->length, out_rows);
        } else {
            if (uniform_rows) {
                made.resize(nb + 1);
                for (size_t k = 0; k <= nb; ++k) made[k] = std::min<uint64_t>(sel->length, static_cast<uint64_t>(k) * uniform_rows);
                in_bounds = &made;
            }
            segment_counts(static_cast<const uint64_t *>(sel->values->ptr), *in_bounds, out_rows);
        }
        uint64_t sum = 0;
        for (size_t b = 0; b < nb; ++b) sum += out_rows[b];
        require(sum == rows, RV_ERR_INTERNAL, "per-batch survivor counts do not add up");
    }
    if (!out_nulls) return;
    bool any_validity = false;
    for (uint32_t j = 0; j < nproj; ++j) any_validity = any_validity || (out[j]->dtype != RV_NULL && out[j]->validity);
    if (!any_validity) {  // no projected column kept a null (config 3: every nullable column is tested): nothing to read
        for (size_t b = 0; b < nb; ++b)
            for (uint32_t j = 0; j < nproj; ++j) out_nulls[b * nproj + j] = out[j]->dtype == RV_NULL ? static_cast<int64_t>(out_rows[b]) : 0;
        return;
    }
    // null count of every output batch: the same segmented count over the compacted validity, at the output boundaries
    std::vector<uint64_t> obounds;
    std::vector<uint64_t> valid(nb);
    for (uint32_t j = 0; j < nproj; ++j) {
        const rv_dcolumn *o = out[j];
        if (o->dtype == RV_NULL) {
            for (size_t b = 0; b < nb; ++b) out_nulls[b * nproj + j] = static_cast<int64_t>(out_rows[b]);
            continue;
        }
        if (!o->validity) {
            for (size_t b = 0; b < nb; ++b) out_nulls[b * nproj + j] = 0;
            continue;
        }
        if (obounds.empty()) {
            obounds.assign(nb + 1, 0);
            for (size_t b = 0; b < nb; ++b) obounds[b + 1] = obounds[b] + out_rows[b];
        }
        segment_counts(static_cast<const uint64_t *>(o->validity->ptr), obounds, valid.data());
        for (size_t b = 0; b < nb; ++b) out_nulls[b * nproj + j] = static_cast<int64_t>(out_rows[b] - valid[b]);
    }
}
}  // namespace

extern "C" {

rv_status rv_filter_project_batches(rv_ctx *ctx, const rv_dcolumn *const *cols, uint32_t nbatches, uint32_t ncols, const rv_predicate *pred,
                                    const uint32_t *proj, uint32_t nproj, rv_dcolumn **out, uint64_t *out_rows, int64_t *out_nulls,
                                    uint64_t *out_total) {
    return guarded([&] {
        require(ctx && cols && pred && pred->terms && (out || nproj == 0) && (proj || nproj == 0) && out_rows, RV_ERR_INVALID_ARG,
                "rv_filter_project_batches: NULL argument");
        require(nbatches >= 1 && ncols >= 1, RV_ERR_INVALID_ARG, "rv_filter_project_batches: no batches / no columns");
        set_device(ctx);
        for (uint32_t j = 0; j < nproj; ++j) out[j] = nullptr;
        maybe_injected_failure(ctx);
        static const bool trace = getenv("RV_TRACE_BATCHES") != nullptr;  // diagnostic: phase times on stderr
        auto tnow = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double tt0 = tnow();
        // ---- coalesce: runs of batches that are adjacent zero-copy slices of the same buffers (what dataframe_to_batches
        //      and RecordBatch::slice hand out, streaming.rs:135-233) are ONE batch as they lie in HBM.  One walk over the
        //      K x ncols handles (each a separate heap object: prefetched a few batches ahead, or the walk is one cache
        //      miss per handle and caps 1024-row batches at ~6e9 rows/s) ----------------------------------------------------
        struct Run {
            uint32_t first, count;
            uint64_t rows;
        };
        std::vector<Run> runs;
        std::vector<uint64_t> bounds(static_cast<size_t>(nbatches) + 1, 0);
        // The walk is one dependent cache miss per handle; past a few thousand batches it is split over host threads
        // (each validates its range and notes length + adjacency to the batch before; the runs are then one linear pass).
        // An error is reported for the FIRST offending batch, as by the sequential walk.
        const size_t nhandles = static_cast<size_t>(nbatches) * ncols, ahead = 8 * static_cast<size_t>(ncols);
        std::vector<uint64_t> lens(nbatches);
        std::vector<uint8_t> adj(nbatches, 0);
        struct WalkError {
            uint32_t batch = UINT32_MAX;
            rv_status status = RV_OK;
            std::string text;
        };
        auto walk = [&](uint32_t b0, uint32_t b1, WalkError &err) {
            for (size_t i = static_cast<size_t>(b0) * ncols; i < std::min(nhandles, static_cast<size_t>(b0) * ncols + ahead); ++i) __builtin_prefetch(cols[i]);
            for (uint32_t b = b0; b < b1; ++b) {
                const rv_dcolumn *const *cur = cols + static_cast<size_t>(b) * ncols;
                auto fail = [&](rv_status st, std::string text) {
                    err.batch = b;
                    err.status = st;
                    err.text = std::move(text);
                };
                for (uint32_t c = 0; c < ncols; ++c) {
                    const size_t i = static_cast<size_t>(b) * ncols + c;
                    if (i + ahead < nhandles) __builtin_prefetch(cols[i + ahead]);
                    if (cur[c] == nullptr) return fail(RV_ERR_INVALID_ARG, "rv_filter_project_batches: a column handle is NULL");
                }
                const uint64_t len = cur[0]->length;
                bool adjacent = b > 0;
                const rv_dcolumn *const *prev = b ? cur - ncols : cur;
                for (uint32_t c = 0; c < ncols; ++c) {
                    // RecordBatch::try_new (record_batch.rs:31-40); every batch of one stream has the stream's schema (stream.rs:58-114)
                    if (cur[c]->length != len)
                        return fail(RV_ERR_LENGTH_MISMATCH, fmt("Column %u has length %llu but expected %llu", c, static_cast<unsigned long long>(cur[c]->length),
                                                                static_cast<unsigned long long>(len)));
                    if (cur[c]->dtype != cols[c]->dtype) return fail(RV_ERR_TYPE_MISMATCH, "All batches must have the same schema");  // record_batch.rs:252-254
                    if (adjacent && prev[c] == nullptr) adjacent = false;  // the NULL is the previous batch's error to report
                    adjacent = adjacent && cur[c]->values == prev[c]->values && cur[c]->validity == prev[c]->validity && cur[c]->offsets == prev[c]->offsets &&
                               cur[c]->offset == prev[c]->offset + prev[c]->length;
                }
                lens[b] = len;
                adj[b] = adjacent ? 1 : 0;
            }
        };
        for (uint32_t c = 0; c < ncols; ++c)  // every range compares its dtypes with batch 0's
            require(cols[c] != nullptr, RV_ERR_INVALID_ARG, "rv_filter_project_batches: a column handle is NULL");
        const uint32_t nthreads = nbatches >= 16384 ? std::min<uint32_t>(8, std::max<uint32_t>(1, std::thread::hardware_concurrency())) : 1;
        std::vector<WalkError> errors(nthreads);
        if (nthreads == 1) {
            walk(0, nbatches, errors[0]);
        } else {
            std::vector<std::thread> pool;
            const uint32_t per = (nbatches + nthreads - 1) / nthreads;
            for (uint32_t t = 0; t < nthreads; ++t) pool.emplace_back([&, t] { walk(std::min(nbatches, t * per), std::min(nbatches, (t + 1) * per), errors[t]); });
            for (auto &th : pool) th.join();
        }
        const WalkError *first_error = nullptr;
        for (auto &e : errors)
            if (e.batch != UINT32_MAX && (!first_error || e.batch < first_error->batch)) first_error = &e;
        if (first_error) throw Error(first_error->status, first_error->text);
        for (uint32_t b = 0; b < nbatches; ++b) {
            bounds[b + 1] = bounds[b] + lens[b];
            if (adj[b]) {
                runs.back().count += 1;
                runs.back().rows += lens[b];
            } else {
                runs.push_back(Run{b, 1, lens[b]});
            }
        }
        std::vector<std::unique_ptr<rv_dcolumn>> owned;
        std::vector<const rv_dcolumn *> whole(ncols);
        for (uint32_t c = 0; c < ncols; ++c) {
            std::vector<const rv_dcolumn *> parts;
            for (const Run &r : runs) {
                const rv_dcolumn *first = cols[static_cast<size_t>(r.first) * ncols + c];
                if (r.count == 1) {
                    parts.push_back(first);
                    continue;
                }
                auto v = std::make_unique<rv_dcolumn>(*first);  // the run as one zero-copy view
                v->length = r.rows;
                v->null_count = first->dtype == RV_NULL ? static_cast<int64_t>(r.rows) : (first->validity ? -1 : 0);
                parts.push_back(v.get());
                owned.emplace_back(std::move(v));
            }
            if (parts.size() == 1) {
                whole[c] = parts[0];
            } else {  // separately allocated batches: one device concat (concat_arrays, record_batch.rs:277-342) in front of the pass
                rv_dcolumn *joined = nullptr;
                const rv_status st = rv_concat(ctx, parts.data(), static_cast<uint32_t>(parts.size()), &joined);
                if (st != RV_OK) throw Error(st, last_error());
                owned.emplace_back(joined);
                whole[c] = joined;
            }
        }
        // ---- one pass over everything; the selection bitmap tells which batch every survivor came from --------------------
        rv_dcolumn *sel = nullptr;
        const double tt1 = tnow();
        // batches of one size (the last one may be shorter) that lie back to back: no boundary table needed, and the pass
        // itself can count the survivors per batch
        uint64_t uniform = bounds[1];
        for (uint32_t b = 1; b < nbatches && uniform; ++b) {
            const uint64_t len = bounds[b + 1] - bounds[b];
            if (len != uniform && !(b + 1 == nbatches && len < uniform)) uniform = 0;
        }
        BatchReq req = make_batch_req(ctx, uniform, nbatches, out_rows);
        const uint64_t rows = filter_query(ctx, whole.data(), ncols, pred, proj, nproj, out, nbatches > 1 ? &sel : nullptr,
                                           (nbatches > 1 && uniform) ? &req : nullptr);
        const double tt2 = tnow();
        std::unique_ptr<rv_dcolumn> sel_owner(sel);
        struct Trace {
            bool on;
            double a, b, c;
            ~Trace() {
                const double d = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
                if (on) fprintf(stderr, "[batches] walk %.2f ms | pass %.2f ms | counts %.2f ms\n", b - a, c - b, d - c);
            }
        } tr{trace, tt0, tt1, tt2};
        try {
            if (out_total) *out_total = rows;
            if (nbatches == 1) {
                out_rows[0] = rows;
                if (out_nulls)
                    for (uint32_t j = 0; j < nproj; ++j) out_nulls[j] = out[j]->dtype == RV_NULL ? static_cast<int64_t>(rows) : std::max<int64_t>(0, out[j]->null_count);
                return;
            }
            require(req.counted || sel != nullptr, RV_ERR_INTERNAL, "per-batch counts: neither counted in the pass nor a selection bitmap to count");
                if (req.counted) finish_batch_req(req, out_rows);
            batch_counts(ctx, req.counted ? nullptr : sel, rows, bounds, uniform, nbatches, out, nproj, out_rows, out_nulls);
        } catch (...) {
            for (uint32_t j = 0; j < nproj; ++j) {
                delete out[j];
                out[j] = nullptr;
            }
            throw;
        }
    });
}

rv_status rv_filter_project_chunked(rv_ctx *ctx, const rv_dcolumn *const *cols, uint32_t ncols, uint64_t chunk_rows, const rv_predicate *pred,
                                    const uint32_t *proj, uint32_t nproj, rv_dcolumn **out, uint64_t *out_rows, uint64_t nchunks,
                                    int64_t *out_nulls, uint64_t *out_total) {
    return guarded([&] {
        require(ctx && cols && pred && pred->terms && (out || nproj == 0) && (proj || nproj == 0), RV_ERR_INVALID_ARG,
                "rv_filter_project_chunked: NULL argument");
        require(ncols >= 1 && chunk_rows >= 1, RV_ERR_INVALID_ARG, "rv_filter_project_chunked: no columns / chunk_rows is 0");
        check_batch(cols, ncols);
        set_device(ctx);
        for (uint32_t j = 0; j < nproj; ++j) out[j] = nullptr;
        const uint64_t n = cols[0]->length;
        // dataframe_to_batches: ceil(n / chunk_rows) batches, none for an empty frame (streaming.rs:135-233)
        const uint64_t nb = (n + chunk_rows - 1) / chunk_rows;
        require(nb <= nchunks && (out_rows || nb == 0), RV_ERR_INVALID_ARG,
                fmt("rv_filter_project_chunked: %llu chunks, room for %llu", static_cast<unsigned long long>(nb), static_cast<unsigned long long>(nchunks)));
        maybe_injected_failure(ctx);
        rv_dcolumn *sel = nullptr;
        BatchReq req = make_batch_req(ctx, chunk_rows, nb, out_rows);
        const uint64_t rows = filter_query(ctx, cols, ncols, pred, proj, nproj, out, nb > 1 ? &sel : nullptr, nb > 1 ? &req : nullptr);
        std::unique_ptr<rv_dcolumn> sel_owner(sel);
        try {
            if (out_total) *out_total = rows;
            if (nb == 1) {
                out_rows[0] = rows;
                if (out_nulls)
                    for (uint32_t j = 0; j < nproj; ++j) out_nulls[j] = out[j]->dtype == RV_NULL ? static_cast<int64_t>(rows) : std::max<int64_t>(0, out[j]->null_count);
            } else if (nb > 1) {
                require(req.counted || sel != nullptr, RV_ERR_INTERNAL, "per-batch counts: neither counted in the pass nor a selection bitmap to count");
                if (req.counted) finish_batch_req(req, out_rows);
                batch_counts(ctx, req.counted ? nullptr : sel, rows, {}, chunk_rows, static_cast<size_t>(nb), out, nproj, out_rows, out_nulls);
            }
        } catch (...) {
            for (uint32_t j = 0; j < nproj; ++j) {
                delete out[j];
                out[j] = nullptr;
            }
            throw;
        }
    });
}

rv_status rv_filter(rv_ctx *ctx, const rv_dcolumn *const *cols, uint32_t ncols, const rv_dcolumn *predicate, rv_dcolumn **out,
                    uint64_t *out_rows) {
    return guarded([&] {
        require(ctx && predicate && (out || ncols == 0), RV_ERR_INVALID_ARG, "rv_filter: NULL argument");
        check_batch(cols, ncols);
        const uint64_t batch_rows = ncols ? cols[0]->length : 0;
        // record_batch.rs:222-233
        require(predicate->length == batch_rows, RV_ERR_LENGTH_MISMATCH,
                fmt("Predicate length %llu doesn't match batch length %llu", static_cast<unsigned long long>(predicate->length),
                    static_cast<unsigned long long>(batch_rows)));
        require(predicate->dtype == RV_BOOLEAN, RV_ERR_TYPE_MISMATCH, "Predicate must be a BooleanArray");
        set_device(ctx);
        std::vector<const rv_dcolumn *> all(cols, cols + ncols);
        all.push_back(predicate);
        std::vector<uint32_t> proj(ncols);
        for (uint32_t i = 0; i < ncols; ++i) {
            proj[i] = i;
            out[i] = nullptr;
        }
        rv_term t{};
        t.column = ncols;
        t.op = RV_IS_TRUE;
        const uint64_t rows = filter_by_groups(ctx, all.data(), ncols + 1, &t, 1, RV_NULL_DROPS, proj.data(), ncols, out, nullptr);
        if (out_rows) *out_rows = rows;
    });
}

}  // extern "C"

namespace {
// RecordBatch::take (record_batch.rs:108-178) with the index list in HBM: bounds pre-pass as a device reduction, then one
// gather per column.  `d_idx` holds n_indices 8-byte indices.
void take_on_device(rv_ctx *ctx, const rv_dcolumn *const *cols, uint32_t ncols, const uint64_t *d_idx, uint64_t n_indices, rv_dcolumn **out) {
    const uint64_t rows = ncols ? cols[0]->length : 0;
    for (uint32_t c = 0; c < ncols; ++c) {
        require(is_value_type(cols[c]->dtype) || cols[c]->dtype == RV_BOOLEAN || cols[c]->dtype == RV_STRING || cols[c]->dtype == RV_NULL,
                RV_ERR_UNSUPPORTED, "rv_take: unsupported dtype");
        out[c] = nullptr;
    }
    if (n_indices) {  // record_batch.rs:109-116: the FIRST index that is out of bounds, with the reference's text
        Ctrl *ctrl = prepare_ctrl(ctx, 0);
        RV_HIP(hipMemsetAsync(&ctrl->pops[0], 0xFF, 8, ctx->stream));
        hipLaunchKernelGGL(rvk::take_bounds_kernel, dim3(grid_for_words(ctx, n_indices, 256)), dim3(256), 0, ctx->stream, d_idx, n_indices, rows, &ctrl->pops[0]);
        RV_HIP(hipGetLastError());
        const Ctrl *h = fetch_ctrl(ctx);
        if (h->pops[0] != ~0ull) {
            uint64_t bad = 0;
            RV_HIP(hipMemcpyAsync(&bad, d_idx + h->pops[0], 8, hipMemcpyDeviceToHost, ctx->stream));
            RV_HIP(hipStreamSynchronize(ctx->stream));
            throw Error(RV_ERR_OUT_OF_BOUNDS, fmt("Index %llu out of bounds for %llu rows", static_cast<unsigned long long>(bad), static_cast<unsigned long long>(rows)));
        }
    }
    try {
        for (uint32_t c = 0; c < ncols; ++c) {
            if (cols[c]->dtype == RV_STRING) {
                out[c] = gather_strings(ctx, cols[c], d_idx, n_indices);
                continue;
            }
            if (cols[c]->dtype == RV_NULL) {  // record_batch.rs:176: NullArray::new(indices.len())
                auto o = std::make_unique<rv_dcolumn>();
                o->dtype = RV_NULL;
                o->length = n_indices;
                o->null_count = static_cast<int64_t>(n_indices);
                out[c] = o.release();
                continue;
            }
            auto o = std::make_unique<rv_dcolumn>();
            o->dtype = cols[c]->dtype;
            o->length = n_indices;
            o->values = pool_alloc(ctx, std::max<size_t>(elem_bytes(cols[c]->dtype, n_indices), 16));
            if (cols[c]->validity) o->validity = pool_alloc(ctx, std::max<size_t>(bitmap_words_bytes(n_indices), 16));
            Ctrl *ctrl = prepare_ctrl(ctx, 0);
            rvk::TakeParams p{};
            p.col = dev_view(cols[c]);
            p.indices = d_idx;
            p.out_values = static_cast<uint64_t *>(o->values->ptr);
            p.out_validity = o->validity ? static_cast<uint64_t *>(o->validity->ptr) : nullptr;
            p.out_valid_pop = striped(ctx, &ctrl->valid_pop[0]);
            p.n = n_indices;
            if (n_indices) {
                hipLaunchKernelGGL(rvk::take_kernel, dim3(grid_for_words(ctx, n_indices, 256)), dim3(256), 0, ctx->stream, p);
                RV_HIP(hipGetLastError());
            }
            const Ctrl *h = fetch_ctrl(ctx);
            o->null_count = o->validity ? static_cast<int64_t>(n_indices - h->valid_pop[0]) : 0;
            if (o->null_count == 0) o->validity.reset();
            out[c] = o.release();
        }
    } catch (...) {
        (void)hipStreamSynchronize(ctx->stream);
        for (uint32_t c = 0; c < ncols; ++c) {
            delete out[c];
            out[c] = nullptr;
        }
        throw;
    }
}
}  // namespace

extern "C" {

rv_status rv_take(rv_ctx *ctx, const rv_dcolumn *const *cols, uint32_t ncols, const uint64_t *indices, uint64_t n_indices,
                  rv_dcolumn **out) {
    return guarded([&] {
        require(ctx && (out || ncols == 0) && (indices || n_indices == 0), RV_ERR_INVALID_ARG, "rv_take: NULL argument");
        check_batch(cols, ncols);
        set_device(ctx);
        DevBufRef d_idx = pool_alloc(ctx, std::max<size_t>(n_indices * 8, 16));
        // the caller's (pageable) list goes up through pinned staging, 8 MiB at a time; it is borrowed for the call only
        constexpr uint64_t kChunk = 1u << 20;
        for (uint64_t at = 0; at < n_indices; at += kChunk) {
            const uint64_t m = std::min<uint64_t>(kChunk, n_indices - at);
            void *hs = ctx->stage(m * 8);
            std::memcpy(hs, indices + at, m * 8);
            RV_HIP(hipMemcpyAsync(static_cast<uint64_t *>(d_idx->ptr) + at, hs, m * 8, hipMemcpyHostToDevice, ctx->stream));
            RV_HIP(hipStreamSynchronize(ctx->stream));
        }
        take_on_device(ctx, cols, ncols, static_cast<const uint64_t *>(d_idx->ptr), n_indices, out);
        RV_HIP(hipStreamSynchronize(ctx->stream));  // d_idx goes back to the pool
    });
}

rv_status rv_take_device(rv_ctx *ctx, const rv_dcolumn *const *cols, uint32_t ncols, const rv_dcolumn *indices, rv_dcolumn **out) {
    return guarded([&] {
        require(ctx && indices && (out || ncols == 0), RV_ERR_INVALID_ARG, "rv_take_device: NULL argument");
        require(indices->dtype == RV_INT64, RV_ERR_TYPE_MISMATCH, "rv_take_device: the index list must be an Int64 array");
        check_batch(cols, ncols);
        set_device(ctx);
        uint64_t nulls = 0;
        const rv_status st = rv_null_count(ctx, indices, &nulls);
        if (st != RV_OK) throw Error(st, last_error());
        require(nulls == 0, RV_ERR_INVALID_ARG, "rv_take_device: the index list must not contain nulls");
        // a negative index reads as an enormous unsigned one and fails the bounds pre-pass like any other
        take_on_device(ctx, cols, ncols, static_cast<const uint64_t *>(indices->values->ptr) + indices->offset, indices->length, out);
        RV_HIP(hipStreamSynchronize(ctx->stream));
    });
}

rv_status rv_selection_indices(rv_ctx *ctx, const rv_dcolumn *selection, rv_dcolumn **out_indices) {
    return guarded([&] {
        require(ctx && selection && out_indices, RV_ERR_INVALID_ARG, "rv_selection_indices: NULL argument");
        require(selection->dtype == RV_BOOLEAN, RV_ERR_TYPE_MISMATCH, "Predicate must be a BooleanArray");  // record_batch.rs:230-233
        set_device(ctx);
        // rows with Some(true) (record_batch.rs:235-240): values under their validity, re-based to bit 0
        const uint64_t n = selection->length;
        std::unique_ptr<rv_dcolumn> flat;
        const rv_dcolumn *sel = selection;
        {  // always re-based: bits past the last row of a caller's buffer may hold anything
            if (selection->validity) {
                rv_dcolumn *f = nullptr;
                const rv_status st = rv_fill_nulls(ctx, selection, &f);
                if (st != RV_OK) throw Error(st, last_error());
                flat.reset(f);
            } else {
                auto o = std::make_unique<rv_dcolumn>();
                o->dtype = RV_BOOLEAN;
                o->length = n;
                o->null_count = 0;
                o->values = pool_alloc(ctx, std::max<size_t>(bitmap_words_bytes(n) + 8, 16));
                if (n) {
                    hipLaunchKernelGGL(rvk::copy_bits_kernel, dim3(grid_for_words(ctx, (n + 63) / 64, 256)), dim3(256), 0, ctx->stream,
                                       static_cast<const uint8_t *>(selection->values->ptr), static_cast<uint64_t>(selection->values->bytes), selection->offset, n,
                                       static_cast<uint64_t *>(o->values->ptr));
                    RV_HIP(hipGetLastError());
                }
                flat = std::move(o);
            }
            sel = flat.get();
        }
        uint64_t t = 0, f = 0;
        {
            const rv_status st = rv_boolean_count(ctx, sel, &t, &f);
            if (st != RV_OK) throw Error(st, last_error());
        }
        DevBufRef excl = selection_prefix(ctx, sel, t);
        DevBufRef idx = selection_to_indices(ctx, sel, t, excl);
        auto o = std::make_unique<rv_dcolumn>();
        o->dtype = RV_INT64;
        o->length = t;
        o->null_count = 0;
        o->values = idx;
        RV_HIP(hipStreamSynchronize(ctx->stream));  // excl goes back to the pool
        *out_indices = o.release();
    });
}

rv_status rv_concat(rv_ctx *ctx, const rv_dcolumn *const *parts, uint32_t nparts, rv_dcolumn **out) {
    return guarded([&] {
        require(ctx && parts && out, RV_ERR_INVALID_ARG, "rv_concat: NULL argument");
        require(nparts >= 1, RV_ERR_INVALID_ARG, "Cannot concatenate empty array list");  // record_batch.rs:280-282
        set_device(ctx);
        const rv_dtype dt = parts[0]->dtype;
        if (dt == RV_STRING) {
            *out = concat_strings(ctx, parts, nparts);
            return;
        }
        if (dt == RV_NULL) {
            auto o = std::make_unique<rv_dcolumn>();
            o->dtype = RV_NULL;
            for (uint32_t i = 0; i < nparts; ++i) {
                require(parts[i] && parts[i]->dtype == RV_NULL, RV_ERR_TYPE_MISMATCH, "All batches must have the same schema");
                o->length += parts[i]->length;
            }
            o->null_count = static_cast<int64_t>(o->length);
            *out = o.release();
            return;
        }
        require(is_value_type(dt) || dt == RV_BOOLEAN, RV_ERR_UNSUPPORTED, "rv_concat: unsupported dtype");
        std::vector<rvk::ConcatPart> hp(nparts);
        std::vector<uint64_t> starts(nparts + 1, 0);
        bool any_validity = false;
        for (uint32_t i = 0; i < nparts; ++i) {
            require(parts[i] && parts[i]->dtype == dt, RV_ERR_TYPE_MISMATCH, "All batches must have the same schema");  // :252-254
            hp[i].values = parts[i]->values->ptr;
            hp[i].validity = parts[i]->validity ? static_cast<const uint8_t *>(parts[i]->validity->ptr) : nullptr;
            hp[i].offset = parts[i]->offset;
            any_validity |= parts[i]->validity != nullptr;
            starts[i + 1] = starts[i] + parts[i]->length;
        }
        const uint64_t n = starts[nparts];
        auto o = std::make_unique<rv_dcolumn>();
        o->dtype = dt;
        o->length = n;
        o->values = pool_alloc(ctx, std::max<size_t>(elem_bytes(dt, n), 16));
        if (any_validity) o->validity = pool_alloc(ctx, std::max<size_t>(bitmap_words_bytes(n), 16));
        DevBufRef d_parts = pool_alloc(ctx, nparts * sizeof(rvk::ConcatPart));
        DevBufRef d_starts = pool_alloc(ctx, (nparts + 1) * 8);
        RV_HIP(hipMemcpyAsync(d_parts->ptr, hp.data(), nparts * sizeof(rvk::ConcatPart), hipMemcpyHostToDevice, ctx->stream));
        RV_HIP(hipMemcpyAsync(d_starts->ptr, starts.data(), (nparts + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        Ctrl *ctrl = prepare_ctrl(ctx, 0);
        rvk::ConcatParams p{};
        p.parts = static_cast<const rvk::ConcatPart *>(d_parts->ptr);
        p.part_start = static_cast<const uint64_t *>(d_starts->ptr);
        p.out_values = static_cast<uint64_t *>(o->values->ptr);
        p.out_validity = o->validity ? static_cast<uint64_t *>(o->validity->ptr) : nullptr;
        p.out_valid_pop = striped(ctx, &ctrl->valid_pop[0]);
        p.n = n;
        p.nparts = nparts;
        p.dtype = static_cast<int32_t>(dt);
        if (n) {
            hipLaunchKernelGGL(rvk::concat_kernel, dim3(grid_for_words(ctx, n, 256)), dim3(256), 0, ctx->stream, p);
            RV_HIP(hipGetLastError());
        }
        const Ctrl *h = fetch_ctrl(ctx);  // also keeps hp/starts alive until the copies are done
        o->null_count = o->validity ? static_cast<int64_t>(n - h->valid_pop[0]) : 0;
        if (o->null_count == 0) o->validity.reset();
        *out = o.release();
    });
}

// ---- host-resident table: chunked upload overlapped with the fused pass -------------------------------
rv_status rv_host_alloc(rv_ctx *ctx, size_t bytes, void **out) {
    return guarded([&] {
        require(ctx && out, RV_ERR_INVALID_ARG, "rv_host_alloc: NULL argument");
        set_device(ctx);
        *out = nullptr;
        if (hipHostMalloc(out, std::max<size_t>(bytes, 8), hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            throw Error(RV_ERR_OOM, fmt("rv_host_alloc: cannot pin %zu bytes", bytes));
        }
    });
}
rv_status rv_host_free(rv_ctx *ctx, void *ptr) {
    return guarded([&] {
        require(ctx, RV_ERR_INVALID_ARG, "rv_host_free: NULL context");
        if (ptr) RV_HIP(hipHostFree(ptr));
    });
}

}  // extern "C"

namespace {
// rows [r0, r0 + len) of a host array -> device column, copies queued on `s` (not waited for)
// `keep`: host staging that must outlive the queued copies (rebased String offsets)
std::unique_ptr<rv_dcolumn> upload_chunk(rv_ctx *ctx, const rv_column &h, uint64_t r0, uint64_t len, hipStream_t s,
                                         std::vector<std::shared_ptr<std::vector<int32_t>>> &keep) {
    auto col = std::make_unique<rv_dcolumn>();
    col->dtype = h.dtype;
    col->length = len;
    const uint64_t first = h.offset + r0;  // absolute element index of the chunk's first row
    auto put_bits = [&](const void *src) {
        // whole bytes that cover bits [first, first + len); the view keeps the sub-byte offset
        const size_t b0 = static_cast<size_t>(first >> 3), nbytes = static_cast<size_t>(((first & 7) + len + 7) >> 3);
        const size_t padded = bitmap_words_bytes((first & 7) + len) + 8;
        DevBufRef b = pool_alloc(ctx, std::max<size_t>(padded, 16));
        RV_HIP(hipMemsetAsync(static_cast<char *>(b->ptr) + (nbytes & ~size_t(7)), 0, std::max<size_t>(padded, 16) - (nbytes & ~size_t(7)), s));
        if (nbytes) RV_HIP(hipMemcpyAsync(b->ptr, static_cast<const uint8_t *>(src) + b0, nbytes, hipMemcpyHostToDevice, s));
        return b;
    };
    if (h.dtype == RV_STRING) {
        // elements [first - back, first + len): offsets rebased to the first byte of the range (string.rs:9-15)
        const uint64_t back = h.validity ? (first & 7) : 0;
        const int32_t *o = h.offsets + (first - back);
        const int32_t b0 = o[0], b1 = o[len + back];
        auto rebased = std::make_shared<std::vector<int32_t>>(len + back + 1);
        for (uint64_t i = 0; i <= len + back; ++i) (*rebased)[i] = o[i] - b0;
        keep.push_back(rebased);
        col->offsets = pool_alloc(ctx, (len + back + 1) * 4 + 16);
        RV_HIP(hipMemcpyAsync(col->offsets->ptr, rebased->data(), (len + back + 1) * 4, hipMemcpyHostToDevice, s));
        const size_t bytes = static_cast<size_t>(b1 - b0);
        col->values = pool_alloc(ctx, std::max<size_t>(bytes + 8, 16));
        if (bytes) RV_HIP(hipMemcpyAsync(col->values->ptr, static_cast<const uint8_t *>(h.values) + b0, bytes, hipMemcpyHostToDevice, s));
        col->data_bytes = bytes;
        col->offset = back;
    } else if (h.dtype == RV_BOOLEAN) {
        col->values = put_bits(h.values);
        col->offset = first & 7;
    } else {
        // a column has ONE offset for its values and its bitmap (primitive.rs:20-28): the bitmap is copied
        // from a byte boundary, so the values start the same `back` elements early (first >= back)
        const uint64_t back = h.validity ? (first & 7) : 0;
        DevBufRef b = pool_alloc(ctx, std::max<size_t>(static_cast<size_t>(len + back) * 8, 16));
        if (len) RV_HIP(hipMemcpyAsync(b->ptr, static_cast<const uint64_t *>(h.values) + (first - back), static_cast<size_t>(len + back) * 8, hipMemcpyHostToDevice, s));
        col->values = b;
        col->offset = back;
    }
    if (h.validity) col->validity = put_bits(h.validity);
    else col->null_count = 0;
    return col;
}
}  // namespace

extern "C" {

rv_status rv_filter_project_host(rv_ctx *ctx, const rv_column *host_cols, uint32_t ncols, const rv_predicate *pred,
                                 const uint32_t *proj, uint32_t nproj, uint64_t chunk_rows, rv_dcolumn **out, uint64_t *out_rows) {
    return guarded([&] {
        require(ctx && host_cols && pred && pred->terms && (out || nproj == 0) && (proj || nproj == 0), RV_ERR_INVALID_ARG,
                "rv_filter_project_host: NULL argument");
        require(ncols >= 1, RV_ERR_INVALID_ARG, "rv_filter_project_host: no columns");
        const uint64_t n = host_cols[0].length;
        for (uint32_t c = 0; c < ncols; ++c) {
            require(is_value_type(host_cols[c].dtype) || host_cols[c].dtype == RV_BOOLEAN || host_cols[c].dtype == RV_STRING, RV_ERR_UNSUPPORTED,
                    "rv_filter_project_host: only Int64, Float64, Boolean and String arrays live on the device");
            require(host_cols[c].length == n, RV_ERR_LENGTH_MISMATCH, "All columns must have the same length");  // record_batch.rs:31-38
            if (host_cols[c].dtype == RV_STRING) {
                require(host_cols[c].offsets != nullptr, RV_ERR_INVALID_ARG, "rv_filter_project_host: offsets is NULL");
                // every chunk's byte range is cut out of these offsets: validate them once, before any copy is sized by them
                check_string_offsets(host_cols[c].offsets, host_cols[c].offset, n, host_cols[c].data_bytes);
            }
            else
                require(host_cols[c].values || host_cols[c].offset + n == 0, RV_ERR_INVALID_ARG, "rv_filter_project_host: values is NULL");
        }
        set_device(ctx);
        if (!ctx->copy_stream) {
            RV_HIP(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
            RV_HIP(hipEventCreateWithFlags(&ctx->ev_up[0], hipEventDisableTiming));
            RV_HIP(hipEventCreateWithFlags(&ctx->ev_up[1], hipEventDisableTiming));
            RV_HIP(hipEventCreateWithFlags(&ctx->ev_main, hipEventDisableTiming));
        }
        uint64_t chunk = chunk_rows ? chunk_rows : (1ull << 25);
        chunk = (chunk + 63) & ~63ull;
        const uint64_t nchunks = n ? (n + chunk - 1) / chunk : 1;
        for (uint32_t j = 0; j < nproj; ++j) out[j] = nullptr;

        struct Batch {
            std::vector<std::unique_ptr<rv_dcolumn>> cols;
            std::vector<std::shared_ptr<std::vector<int32_t>>> keep;  // host staging of the chunk's queued copies
        };
        auto issue = [&](uint64_t k) {
            // pool blocks handed to this chunk may still be read by work queued on the main stream
            RV_HIP(hipEventRecord(ctx->ev_main, ctx->stream));
            RV_HIP(hipStreamWaitEvent(ctx->copy_stream, ctx->ev_main, 0));
            Batch b;
            const uint64_t r0 = k * chunk, len = std::min(chunk, n - std::min(n, r0));
            for (uint32_t c = 0; c < ncols; ++c) b.cols.push_back(upload_chunk(ctx, host_cols[c], r0, len, ctx->copy_stream, b.keep));
            RV_HIP(hipEventRecord(ctx->ev_up[k & 1], ctx->copy_stream));
            return b;
        };
        std::vector<std::vector<rv_dcolumn *>> parts(nproj);
        auto drop_parts = [&] {
            for (auto &v : parts)
                for (auto *d : v) delete d;
            parts.assign(nproj, {});
        };
        uint64_t total = 0;
        try {
            Batch cur = issue(0);
            for (uint64_t k = 0; k < nchunks; ++k) {
                RV_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_up[k & 1], 0));
                Batch next;
                if (k + 1 < nchunks) next = issue(k + 1);  // flies while chunk k is filtered
                std::vector<const rv_dcolumn *> view;
                for (auto &c : cur.cols) view.push_back(c.get());
                std::vector<rv_dcolumn *> o(nproj, nullptr);
                total += filter_query(ctx, view.data(), ncols, pred, proj, nproj, o.data(), nullptr);
                for (uint32_t j = 0; j < nproj; ++j) parts[j].push_back(o[j]);
                cur = std::move(next);  // the chunk's inputs go back to the pool (its kernel has finished)
            }
            RV_HIP(hipStreamSynchronize(ctx->copy_stream));
            for (uint32_t j = 0; j < nproj; ++j) {
                if (parts[j].size() == 1) {
                    out[j] = parts[j][0];
                    parts[j].clear();
                } else {
                    rv_dcolumn *joined = nullptr;
                    std::vector<const rv_dcolumn *> cp(parts[j].begin(), parts[j].end());
                    const rv_status st = rv_concat(ctx, cp.data(), static_cast<uint32_t>(cp.size()), &joined);
                    if (st != RV_OK) throw Error(st, g_last_error);
                    out[j] = joined;
                }
            }
            drop_parts();
        } catch (...) {
            (void)hipStreamSynchronize(ctx->copy_stream);
            drop_parts();
            for (uint32_t j = 0; j < nproj; ++j) {
                delete out[j];
                out[j] = nullptr;
            }
            throw;
        }
        if (out_rows) *out_rows = total;
    });
}

// ---- filter + aggregate ---------------------------------------------------------------------------
rv_status rv_filter_agg(rv_ctx *ctx, const rv_dcolumn *const *cols, uint32_t ncols, const rv_predicate *pred, uint32_t agg_col,
                        int64_t *sum_i, double *sum_f, uint64_t *count) {
    return guarded([&] {
        require(ctx && pred && pred->terms, RV_ERR_INVALID_ARG, "rv_filter_agg: NULL argument");
        require(ncols >= 1 && agg_col < ncols, RV_ERR_INVALID_ARG, "rv_filter_agg: bad column index");
        check_batch(cols, ncols);
        require(is_value_type(cols[agg_col]->dtype), RV_ERR_UNSUPPORTED, "rv_filter_agg: SUM needs an Int64 or Float64 column");
        set_device(ctx);
        maybe_injected_failure(ctx);
        // String compares / many Boolean columns / OR and NOT: normalised first, as for the compaction path
        Normalized nz;
        normalize_predicate(ctx, cols, ncols, pred, nz);
        cols = nz.cols.data();
        ncols = static_cast<uint32_t>(nz.cols.size());
        rv_predicate folded = *pred;
        folded.terms = nz.terms.data();
        folded.n_terms = static_cast<uint32_t>(nz.terms.size());
        folded.expr = nullptr;
        folded.n_expr = 0;
        pred = &folded;
        const ExprInfo *ex = nz.expr();
        // more 8-byte columns than one pass reads (the aggregated column + the predicate's): the predicate is
        // evaluated into a selection bitmap first and the aggregate reads that
        std::unique_ptr<rv_dcolumn> sel_owner;
        std::vector<const rv_dcolumn *> two;
        rv_term sel_term{};
        {
            std::vector<char> seen(ncols, 0);
            int nv = 1;
            seen[agg_col] = 1;
            for (uint32_t t = 0; t < pred->n_terms; ++t) {
                const uint32_t c = pred->terms[t].column;
                require(c < ncols, RV_ERR_INVALID_ARG, "rv_filter_agg: term column out of range");
                if (is_value_type(cols[c]->dtype) && !seen[c]) seen[c] = 1, ++nv;
            }
            if (ex)
                for (uint32_t c : ex->strict_cols)
                    if (is_value_type(cols[c]->dtype) && cols[c]->validity && !seen[c]) seen[c] = 1, ++nv;
            if (nv > rvk::kMaxValueCols) {
                rv_dcolumn *sel = nullptr, *none = nullptr;
                filter_by_groups(ctx, cols, ncols, pred->terms, pred->n_terms, pred->nulls, nullptr, 0, &none, &sel, ex);
                sel_owner.reset(sel);
                two = {cols[agg_col], sel};
                cols = two.data();
                ncols = 2;
                agg_col = 0;
                sel_term.column = 1;
                sel_term.op = RV_IS_TRUE;
                folded.terms = &sel_term;
                folded.n_terms = 1;
                folded.nulls = RV_NULL_DROPS;
                ex = nullptr;
            }
        }
        const uint64_t n = cols[0]->length;
        rvk::AggParams p{};
        p.in.n = n;
        p.in.nterms = static_cast<int32_t>(pred->n_terms);
        p.agg_is_float = cols[agg_col]->dtype == RV_FLOAT64;
        std::vector<int> vslot(ncols, -1), bslot(ncols, -1);
        int nvals = 0, nbools = 0;
        vslot[agg_col] = nvals;
        p.in.cols[nvals++] = dev_view(cols[agg_col]);  // slot 0 == aggregated column
        for (uint32_t t = 0; t < pred->n_terms; ++t) {
            const uint32_t c = pred->terms[t].column;
            require(c < ncols, RV_ERR_INVALID_ARG, "rv_filter_agg: term column out of range");
            uint32_t slot;
            if (cols[c]->dtype == RV_BOOLEAN) {
                if (bslot[c] < 0) {
                    require(nbools < rvk::kMaxBoolCols, RV_ERR_UNSUPPORTED, "too many Boolean predicate columns");
                    bslot[c] = nbools;
                    p.in.bcols[nbools++] = dev_view(cols[c]);
                }
                slot = static_cast<uint32_t>(bslot[c]);
            } else {
                require(is_value_type(cols[c]->dtype), RV_ERR_UNSUPPORTED, "rv_filter_agg: unsupported predicate column type");
                if (vslot[c] < 0) {
                    require(nvals < rvk::kMaxValueCols, RV_ERR_UNSUPPORTED, "too many 8-byte columns");
                    vslot[c] = nvals;
                    p.in.cols[nvals++] = dev_view(cols[c]);
                }
                slot = static_cast<uint32_t>(vslot[c]);
            }
            p.in.terms[t] = lower_term(pred->terms[t], cols[c]->dtype, pred->nulls, slot);
            if (ex) p.in.terms[t].set_literal(ex->negate[t] != 0, ex->group_end[t] != 0);
        }
        if (ex) {
            p.in.expr_mode = 1;
            p.in.negate_result = ex->negate_result ? 1 : 0;
            if (ex->strict)
                for (uint32_t c : ex->strict_cols) {
                    if (is_value_type(cols[c]->dtype) && cols[c]->validity) {
                        if (vslot[c] < 0) {
                            require(nvals < rvk::kMaxValueCols, RV_ERR_UNSUPPORTED, "too many 8-byte columns");
                            vslot[c] = nvals;
                            p.in.cols[nvals++] = dev_view(cols[c]);
                        }
                        p.in.strict_values |= 1u << vslot[c];
                    } else if (cols[c]->dtype == RV_BOOLEAN && cols[c]->validity) {
                        if (bslot[c] < 0) {
                            require(nbools < rvk::kMaxBoolCols, RV_ERR_UNSUPPORTED, "too many Boolean predicate columns");
                            bslot[c] = nbools;
                            p.in.bcols[nbools++] = dev_view(cols[c]);
                        }
                        p.in.strict_bools |= 1u << bslot[c];
                    }
                }
        }
        if (n == 0) {
            if (sum_i) *sum_i = 0;
            if (sum_f) *sum_f = 0.0;
            if (count) *count = 0;
            return;
        }
        int vec = ctx->opt_vec == 1 ? 1 : (ctx->opt_vec == 2 ? 2 : (nvals <= 1 ? 2 : 1));
        for (int s = 0; s < nvals; ++s)
            if ((reinterpret_cast<uintptr_t>(p.in.cols[s].values) + p.in.cols[s].offset * 8) & 15) vec = 1;
        int need = nbools ? rvk::FF_BOOL : 0;
        for (int s = 0; s < nvals; ++s)
            if (p.in.cols[s].validity) need |= rvk::FF_VALIDITY;
        size_t nagg = 0;
        const rvk::AggEntry *table = rvk::agg_entries(&nagg), *e = nullptr;
        for (size_t i = 0; i < nagg; ++i)
            if (table[i].ncols == nvals && table[i].vec == vec && (table[i].flags & need) == need &&
                (!e || __builtin_popcount(table[i].flags) < __builtin_popcount(e->flags)))
                e = &table[i];
        require(e != nullptr, RV_ERR_INTERNAL, "no aggregate kernel variant");
        const uint64_t tile_rows = static_cast<uint64_t>(e->waves) * 64 * e->r;
        const uint64_t ntiles = (n + tile_rows - 1) / tile_rows;
        require(ntiles < (1ull << 31), RV_ERR_UNSUPPORTED, "batch too large for one launch");
        Ctrl *ctrl = prepare_ctrl(ctx, 0);
        // The kernel strides over the tiles.  32 workgroups per CU (four rounds of the eight a CU holds): 1e9 Int64 rows
        // 1.145 ms = 87.4 % of the HBM peak against 1.21 ms = 82.7 % with one workgroup per tile (244 k workgroups, each
        // fetching its kernel arguments before its first load) -- tools/agg_grid.py.  Option "agg_grid": k > 0 = k per CU,
        // -1 = one per tile.
        // The default grid is a CONSTANT (8192 workgroups = 32 per CU of an MI355X), not a multiple of the CU count: the order
        // of a Float64 sum's additions then depends on the row count alone -- the same bits on any part, in any partition mode.
        const uint64_t grid = ctx->opt_agg_grid < 0 ? ntiles
                                                    : std::min<uint64_t>(ntiles, ctx->opt_agg_grid > 0 ? static_cast<uint64_t>(ctx->opt_agg_grid) * static_cast<uint64_t>(ctx->props.multiProcessorCount)
                                                                                                       : 8192);
        DevBufRef partials = pool_alloc(ctx, grid * sizeof(rvk::AggPartial));
        p.partials = static_cast<rvk::AggPartial *>(partials->ptr);
        p.ntiles = static_cast<uint32_t>(ntiles);
        ctx->last_kernel = fmt("filter_agg_kernel<%d,%d,%d,%d,%d>", e->ncols, e->r, e->vec, e->waves, e->flags);
        if (ctx->opt_profile) RV_HIP(hipEventRecord(ctx->evk0, ctx->stream));
        hipLaunchKernelGGL(e->fn, dim3(static_cast<uint32_t>(grid)), dim3(e->waves * 64), 0, ctx->stream, p);
        RV_HIP(hipGetLastError());
        if (ctx->opt_profile) RV_HIP(hipEventRecord(ctx->evk1, ctx->stream));
        if (grid > 16384) {  // two levels: 1024-partial chunks first
            const uint32_t chunk = 1024, nchunks = static_cast<uint32_t>((grid + chunk - 1) / chunk);
            DevBufRef level1 = pool_alloc(ctx, static_cast<size_t>(nchunks) * sizeof(rvk::AggPartial));
            hipLaunchKernelGGL(rvk::agg_final_kernel<0>, dim3(nchunks), dim3(1024), 0, ctx->stream, p.partials, static_cast<uint32_t>(grid), chunk,
                               static_cast<rvk::AggPartial *>(level1->ptr));
            hipLaunchKernelGGL(rvk::agg_final_kernel<0>, dim3(1), dim3(1024), 0, ctx->stream, static_cast<const rvk::AggPartial *>(level1->ptr), nchunks, nchunks,
                               &ctrl->agg);
            RV_HIP(hipGetLastError());
            // level1 returns to the pool at scope end; later users run on this stream, after the fold
        } else {
            hipLaunchKernelGGL(rvk::agg_final_kernel<0>, dim3(1), dim3(1024), 0, ctx->stream, p.partials, static_cast<uint32_t>(grid),
                               static_cast<uint32_t>(grid), &ctrl->agg);
            RV_HIP(hipGetLastError());
        }
        const Ctrl *h = fetch_ctrl(ctx);
        if (ctx->opt_profile) {
            float ms = 0.f;
            RV_HIP(hipEventElapsedTime(&ms, ctx->evk0, ctx->evk1));
            ctx->kernel_ms += ms;
            ctx->kernel_launches += 1;
        }
        if (sum_i) *sum_i = h->agg.sum_i;
        if (sum_f) *sum_f = h->agg.sum_f;
        if (count) *count = h->agg.count;
    });
}

// ---- multi-GPU -----------------------------------------------------------------------------------------
rv_status rv_shard_range(uint64_t n_rows, uint32_t world, uint32_t rank, uint64_t *begin, uint64_t *end) {
    return guarded([&] {
        require(world >= 1 && rank < world && begin && end, RV_ERR_INVALID_ARG, "rv_shard_range: bad arguments");
        // ceil(N / world) rounded up to a multiple of 64 rows (one selection-bitmap word)
        uint64_t per = (n_rows + world - 1) / world;
        per = (per + 63) & ~uint64_t(63);
        *begin = std::min<uint64_t>(n_rows, per * rank);
        *end = std::min<uint64_t>(n_rows, per * (static_cast<uint64_t>(rank) + 1));
    });
}

}  // extern "C"

struct rv_comm {
    rv_ctx *ctx = nullptr;
    void *comm = nullptr;
    void *d_buf = nullptr;  // 2 x int64 on the device
};

extern "C" {

rv_status rv_comm_unique_id(uint8_t id[RV_COMM_ID_BYTES]) {
    return guarded([&] {
        require(id != nullptr, RV_ERR_INVALID_ARG, "id is NULL");
        rccl_check(rccl().GetUniqueId(id), "ncclGetUniqueId");
    });
}

rv_status rv_comm_create(rv_ctx *ctx, const uint8_t id[RV_COMM_ID_BYTES], uint32_t world, uint32_t rank, rv_comm **out) {
    return guarded([&] {
        require(ctx && id && out && rank < world, RV_ERR_INVALID_ARG, "rv_comm_create: bad arguments");
        set_device(ctx);
        auto c = std::make_unique<rv_comm>();
        c->ctx = ctx;
        std::array<char, RV_COMM_ID_BYTES> uid;
        std::memcpy(uid.data(), id, RV_COMM_ID_BYTES);
        rccl_check(rccl().CommInitRank(&c->comm, static_cast<int>(world), uid, static_cast<int>(rank)), "ncclCommInitRank");
        RV_HIP(hipMalloc(&c->d_buf, 16));
        *out = c.release();
    });
}

rv_status rv_comm_allreduce_sum_count(rv_comm *comm, int64_t *sum, uint64_t *count) {
    return guarded([&] {
        require(comm && sum && count, RV_ERR_INVALID_ARG, "rv_comm_allreduce_sum_count: NULL argument");
        set_device(comm->ctx);
        int64_t h[2] = {*sum, static_cast<int64_t>(*count)};
        hipStream_t s = comm->ctx->stream;
        RV_HIP(hipMemcpyAsync(comm->d_buf, h, 16, hipMemcpyHostToDevice, s));
        // ncclInt64 == 4, ncclSum == 0 (rccl.h)
        rccl_check(rccl().AllReduce(comm->d_buf, comm->d_buf, 2, 4, 0, comm->comm, s), "ncclAllReduce");
        RV_HIP(hipMemcpyAsync(h, comm->d_buf, 16, hipMemcpyDeviceToHost, s));
        RV_HIP(hipStreamSynchronize(s));
        *sum = h[0];
        *count = static_cast<uint64_t>(h[1]);
    });
}

rv_status rv_comm_destroy(rv_comm *comm) {
    return guarded([&] {
        if (!comm) return;
        set_device(comm->ctx);
        if (comm->comm) (void)rccl().CommDestroy(comm->comm);
        if (comm->d_buf) (void)hipFree(comm->d_buf);
        delete comm;
    });
}

}  // extern "C"
