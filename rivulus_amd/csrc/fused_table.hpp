// Launch tables of the fused / aggregate kernel instantiations.  The instantiations are
// spread over several translation units (fused_nc*.hip) so hipcc can build them in parallel.
#pragma once

#include "agg_kernel.hpp"
#include "fused_kernel.hpp"

namespace rvk {

struct FusedEntry {
    int ncols, r, vec, waves, flags;
    void (*fn)(const FusedParams);
};
struct AggEntry {
    int ncols, r, vec, waves, flags;
    void (*fn)(const AggParams);
};
#define RV_FUSED(NC, R, V, W, F) ::rvk::FusedEntry{NC, R, V, W, F, &::rvk::fused_filter_compact<NC, R, V, W, F>}
#define RV_AGG(NC, R, V, W, F) ::rvk::AggEntry{NC, R, V, W, F, &::rvk::filter_agg_kernel<NC, R, V, W, F>}

// each returns a static array and its length
const FusedEntry *fused_entries_lean1(size_t *n);   // 1 column, no nulls: BASELINE config 2
const FusedEntry *fused_entries_valid1(size_t *n);  // 1 column with a null bitmap
const FusedEntry *fused_entries_multi(size_t *n);   // 2..4 columns
const FusedEntry *fused_entries_bool(size_t *n);    // Boolean-column predicate, 1..4 columns compacted
const FusedEntry *fused_entries_full(size_t *n);    // every feature (Boolean terms/columns, selection)
const FusedEntry *fused_entries_expr(size_t *n);    // OR / NOT expressions (conjunctive normal form)
const FusedEntry *fused_entries_roomy(size_t *n);   // 2..4 columns, slots that hold every row of a wave (dense selections)
// the direct (register-staged) kernel for dense selections (direct_kernel.hpp): np 8-byte columns the predicate reads, nq more
// that are only projected; flags: FF_VALIDITY / FF_BOOL of the predicate's inputs
struct DirectEntry {
    int np, nq, r, waves, flags;
    void (*fn)(const FusedParams);
};
const DirectEntry *direct_entries_a(size_t *n);  // fused_direct.hip
const DirectEntry *direct_entries_b(size_t *n);  // fused_direct2.hip
const AggEntry *agg_entries(size_t *n);
// the strided selectivity sample (agg_kernel.hpp) for 0..4 loaded 8-byte columns
using SampleFn = void (*)(const SampleParams);
SampleFn sample_kernel(int ncols);
// redo kernel (wave ranges whose survivors outgrew their LDS slot) for 0..4 loaded 8-byte columns
using RedoFn = void (*)(const FusedParams, uint32_t, uint64_t);
int redo_rows_per_lane(int ncols, uint32_t range_rows);  // 4, 8 or 16 rows per lane and step
RedoFn redo_kernel(int ncols, int rows_per_lane);

}  // namespace rvk
