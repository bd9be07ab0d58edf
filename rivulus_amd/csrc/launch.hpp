// Internal interface between the translation units of librivulus_gpu.so (core / fused_launch / strings / predicate / arrays /
// query / take_concat / host_table / aggregate .hip): the control block, the request / launch records that travel with a
// fused pass, and the functions the units call across each other.  Nothing here is part of the C ABI.
#pragma once

#include <algorithm>
#include <array>
#include <chrono>
#include <functional>
#include <thread>

#include "aux_kernels.hpp"
#include "fused_table.hpp"
#include "string_kernels.hpp"
#include "runtime.hpp"
#include "thresholds.hpp"

namespace rvl {
using namespace rvh;

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

// ---- core.hip ---------------------------------------------------------------------------------------------
size_t elem_bytes(rv_dtype t, uint64_t n);
size_t bitmap_words_bytes(uint64_t n);
// bytes of an output bitmap of n rows that kernels OR their edge words into: its words + one, rounded up to 256 bytes -- the runtime
// zeroes a size that is not a multiple of 16 bytes with TWO fill launches (the bulk and the remainder: 5 us more on the stream each time)
inline size_t zeroed_bitmap_bytes(uint64_t n) { return (bitmap_words_bytes(n) + 8 + 255) & ~static_cast<size_t>(255); }
DevBufRef pool_alloc(rv_ctx *ctx, size_t bytes);
void set_device(rv_ctx *ctx);
void maybe_injected_failure(rv_ctx *ctx);
Ctrl *prepare_ctrl(rv_ctx *ctx, size_t ntiles);
unsigned long long *striped(rv_ctx *ctx, const unsigned long long *ctrl_word);
const Ctrl *fetch_ctrl(rv_ctx *ctx);
rv_ctx::LaunchCtrl acquire_launch_ctrl(rv_ctx *ctx, size_t ntiles, size_t nranges = 0);
void release_launch_ctrl(rv_ctx *ctx, const rv_ctx::LaunchCtrl &c);
int block_to_zero(rv_ctx *ctx, const rv_ctx::LaunchCtrl &like);
rvk::DevCol dev_view(const rv_dcolumn *c);
bool is_value_type(rv_dtype t);
void check_string_offsets(const int32_t *offsets, uint64_t first, uint64_t count, uint64_t data_bytes);
rvk::DevTerm lower_term(const rv_term &t, rv_dtype col_type, rv_null_policy policy, uint32_t slot = 0);
int grid_for_words(rv_ctx *ctx, uint64_t items, int block);

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
    double expected_selectivity = -1.0;  // what the pass was sized by (the predicate's memory or the sample); < 0: unknown
};

// One single-pass launch in flight: everything fused_finish needs once the kernel has run.
struct FusedLaunch {
    rvk::FusedParams p{};
    std::vector<OutCol> outs;
    rv_ctx::LaunchCtrl ctrl;
    int need = 0, nvals = 0, nxs = 0;
    size_t stage_row_bytes = 0;
    uint64_t tile_rows = 0;
    uint32_t range_rows = 0;  // rows of a wave range (64 x rows per lane); nranges: ranges of the launch's redo list (0: no list)
    uint64_t nranges = 0;
    bool launched = false;  // false: empty input, nothing to wait for
    DevBufRef wave_counts;      // scratch of the per-batch counts kernel queued behind the pass, kept until the launch is finished
    bool redo_queued = false;  // the redo kernel was queued right behind the pass (ranges were expected to outgrow their slots)
    bool sample_only = false;  // fused_begin stops behind the selectivity it would size the launch by (expected_selectivity)
    double sampled = -1.0;
    bool timed = false;     // kernel events recorded (option profile_kernels)
    bool direct_stamp = false;  // a diagnostic instantiation of the direct kernel ran: print its phase sums
    // for a re-run after an output overflow (speculative sizing)
    void (*fn)(const rvk::FusedParams) = nullptr;
    uint32_t grid = 0, block = 0;
    size_t lds = 0;
    uint64_t n = 0;
    uint64_t signature = 0;  // of the predicate (rv_ctx::seen)
    std::vector<rv_dtype> out_dtypes;  // dtype of every projected source column
    // ---- a SEGMENT of a table filtered piece by piece (run_segmented_pass): the pass writes into buffers the caller owns, from row
    //      `place_base` on; it is sized by `place_selectivity` (the segment's own, out of the sample's profile) instead of the
    //      predicate's memory, leaves that memory alone, and an overflow of the shared outputs is the caller's to handle (`overflowed`)
    const std::vector<DevBufRef> *place = nullptr;  // one values buffer per projected column
    uint64_t place_base = 0, place_capacity = 0;    // rows
    double place_selectivity = -1.0;
    bool place_edge = false;  // a sparse stretch next to a dense one: the block the edge falls into leaves a few wave ranges to the redo kernel
    bool overflowed = false;
    uint32_t out_bias = 0;  // out: FusedParams::out_bias of the launch (place_base & 15)
};
struct SegmentOverflow {};  // thrown by run_segmented_pass's callee chain: the shared outputs were too small -- the caller falls back to one pass

// `after_launch` (optional) runs between the two halves, with the selection bitmap the pass is writing: work queued there
// (the String gather of a filter) follows the pass on the stream without the host having waited for anything.
using AfterLaunch = std::function<void(const rv_dcolumn *sel)>;

// ---- fused_launch.hip --------------------------------------------------------------------------------------
uint64_t output_capacity(rv_ctx *ctx, uint64_t n, double expected);
double expected_selectivity(rv_ctx *ctx, const rv_dcolumn *const *cols, uint32_t ncols, const rv_term *terms, uint32_t nterms, rv_null_policy policy, const ExprInfo *ex);
uint64_t predicate_signature(const rv_dcolumn *const *cols, uint32_t ncols, const rv_term *terms, uint32_t nterms, rv_null_policy policy, const ExprInfo *ex);
void fused_begin(rv_ctx *ctx, const rv_dcolumn *const *cols, uint32_t ncols, const rv_term *terms, uint32_t nterms, rv_null_policy policy,
                 const uint32_t *proj, uint32_t nproj, rv_dcolumn **out, rv_dcolumn **sel_out, FusedLaunch &L, const ExprInfo *ex = nullptr,
                 BatchReq *req = nullptr, RangeOffsets *ranges = nullptr);
uint64_t fused_finish(rv_ctx *ctx, FusedLaunch &L);
void abandon_launch(rv_ctx *ctx, FusedLaunch &L);  // a launch nobody will finish: drained, its control block released
uint64_t run_fused_pass(rv_ctx *ctx, const rv_dcolumn *const *cols, uint32_t ncols, const rv_term *terms, uint32_t nterms, rv_null_policy policy,
                        const uint32_t *proj, uint32_t nproj, rv_dcolumn **out, rv_dcolumn **sel_out, const ExprInfo *ex = nullptr, BatchReq *req = nullptr,
                        const AfterLaunch *after_launch = nullptr, RangeOffsets *ranges = nullptr);
// A table whose survivors sit in a few long stretches (sorted on the predicate's column: the sample's profile says so) filtered stretch
// by stretch, each with the kernel its own density asks for, all into one set of outputs.  False: not such a table / not such a query
// (nothing was launched: take the one-pass path).
bool run_segmented_pass(rv_ctx *ctx, const rv_dcolumn *const *cols, uint32_t ncols, const rv_term *terms, uint32_t nterms, rv_null_policy policy,
                        const uint32_t *proj, uint32_t nproj, rv_dcolumn **out, const ExprInfo *ex, uint64_t *rows_out);

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
    DevBufRef lengths, starts, block_sums, groups;
    uint64_t cap_rows = 0;
    Ctrl *ctrl = nullptr;
    int slot = 0;  // valid_pop[slot]: surviving valid elements; pops[0]: total bytes
    bool queued = false;
    bool tiles = false;  // source-tile order (dense selections): sel_str_tile_sums / sel_str_tile_copy, all queued by str_sel_queue
};

// filter() of a BooleanArray queued right behind the fused pass: bits_compact_kernel finds every wave's output position in
// the pass's wave offsets (RangeOffsets) instead of a scan over the selection bitmap -- one launch, nothing waited for.
// Counter `slot` of the shared control block (valid_pop[slot]) receives the surviving validity bits.
struct BoolCompactLaunch {
    std::unique_ptr<rv_dcolumn> col;
    uint64_t cap_rows = 0;  // rows the output bitmaps hold (the pass's capacity, or the expected survivors + 25 %)
    int slot = -1;
    bool launched = false;
};


// ---- strings.hip -------------------------------------------------------------------------------------------
// total_dst (device, zeroed by the caller): where the scan leaves its total instead of the context's control block -- nothing of the
// context is touched then, so scans of several windows may be in flight; want_total is ignored
uint64_t device_exclusive_scan(rv_ctx *ctx, const void *counts, uint64_t n, DevBufRef &excl, bool pop = false, bool want_total = true,
                               unsigned long long *total_dst = nullptr);
DevBufRef selection_prefix(rv_ctx *ctx, const rv_dcolumn *sel, uint64_t rows);
DevBufRef selection_to_indices(rv_ctx *ctx, const rv_dcolumn *sel, uint64_t rows, const DevBufRef &excl);
rv_dcolumn *compact_boolean(rv_ctx *ctx, const rv_dcolumn *src, const rv_dcolumn *sel, uint64_t rows, const DevBufRef &excl);
rv_dcolumn *gather_strings(rv_ctx *ctx, const rv_dcolumn *src, const uint64_t *d_indices, uint64_t n);
rv_dcolumn *gather_strings_selected(rv_ctx *ctx, const rv_dcolumn *src, const rv_dcolumn *sel, uint64_t rows, const DevBufRef &excl);
bool str_sel_eligible(const rv_dcolumn *sel, const RangeOffsets &ranges);
void str_sel_queue(rv_ctx *ctx, const rv_dcolumn *src, const rv_dcolumn *sel, const RangeOffsets &ranges, Ctrl *ctrl, int slot, StrSelLaunch &L);
void str_sel_copy(rv_ctx *ctx, StrSelLaunch &L, uint64_t rows);
rv_dcolumn *str_sel_result(StrSelLaunch &L, uint64_t rows, const Ctrl &fetched);
void bool_compact_queue(rv_ctx *ctx, const rv_dcolumn *src, const rv_dcolumn *sel, const RangeOffsets &ranges, Ctrl *ctrl, int slot, BoolCompactLaunch &L);
rv_dcolumn *bool_compact_result(BoolCompactLaunch &L, uint64_t rows, const Ctrl &fetched);
rv_dcolumn *concat_strings(rv_ctx *ctx, const rv_dcolumn *const *parts, uint32_t nparts);
rv_dcolumn *string_term_mask(rv_ctx *ctx, const rv_dcolumn *col, const rv_term &t, rv_null_policy policy);

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


// ---- predicate.hip -----------------------------------------------------------------------------------------
void normalize_predicate(rv_ctx *ctx, const rv_dcolumn *const *cols, uint32_t ncols, const rv_predicate *pred, Normalized &out);

// ---- arrays.hip --------------------------------------------------------------------------------------------
void check_batch(const rv_dcolumn *const *cols, uint32_t ncols);
void bool_op(rv_ctx *ctx, int kind, const rv_dcolumn *a, const rv_dcolumn *b, rv_dcolumn **out);

// ---- query.hip ---------------------------------------------------------------------------------------------
uint64_t filter_by_groups(rv_ctx *ctx, const rv_dcolumn *const *cols, uint32_t ncols, const rv_term *terms, uint32_t nterms, rv_null_policy policy,
                          const uint32_t *proj, uint32_t nproj, rv_dcolumn **out, rv_dcolumn **out_selection, const ExprInfo *ex = nullptr,
                          BatchReq *req = nullptr, const AfterLaunch *after_launch = nullptr, RangeOffsets *ranges = nullptr);
uint64_t filter_query(rv_ctx *ctx, const rv_dcolumn *const *cols, uint32_t ncols, const rv_predicate *pred, const uint32_t *proj, uint32_t nproj,
                      rv_dcolumn **out, rv_dcolumn **out_selection, BatchReq *req = nullptr);

}  // namespace rvl
