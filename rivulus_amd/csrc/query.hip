// RecordBatch kernels and the query entry points: filter_by_groups, rv_filter_project*, many RecordBatches in one launch, rv_filter.
// One unit of the backend library behind include/rivulus_gpu.h (gfx950 only; compiled with hipcc).  Shared helpers and the
// functions the units call across each other are declared in launch.hpp (namespace rvl).
#include "launch.hpp"
#include "ranges_kernel.hpp"

using namespace rvh;
using namespace rvl;

namespace rvl {
// ---- RecordBatch kernels ---------------------------------------------------------------------------
// Columns are compacted in groups that fit one single-pass launch (<= 4 eight-byte columns
// and <= 4 bit streams each); every group re-reads the predicate bitmap only (1 bit/row).
// `terms` is a normalised term list (normalize_predicate): no String columns, at most kMaxBoolCols Boolean ones.
uint64_t filter_by_groups(rv_ctx *ctx, const rv_dcolumn *const *cols, uint32_t ncols, const rv_term *terms, uint32_t nterms,
                          rv_null_policy policy, const uint32_t *proj, uint32_t nproj, rv_dcolumn **out, rv_dcolumn **out_selection,
                          const ExprInfo *ex, BatchReq *req, const AfterLaunch *after_launch, RangeOffsets *ranges) {
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
                    const size_t wb = zeroed_bitmap_bytes(rows);
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
    // A sparse selection over a big table: the chained pass reads only what the PREDICATE reads (and compacts what of it is
    // projected, and whatever needs the pass: nullable and Boolean columns); every other plain 8-byte column is compacted afterwards at
    // the pass's wave offsets (ranges_kernel.hpp) -- that kernel runs at the read-only aggregate's rate (7.0 TB/s: nothing is shared
    // between its waves), the chained pass with four columns at 5.8.  Decided from what this predicate kept the last time it ran
    // over these buffers, or from the sample a first call over 2^25 rows and more takes anyway (expected_selectivity: the reference
    // has no warm-up call); should the pass then keep more than 55 % of the rows, or its offsets be unusable, the deferred groups run
    // as passes of their own, as before.
    bool defer_plain = false;
    const uint64_t n_rows = ncols ? cols[0]->length : 0;
    // (a window of RecordBatches -- `req` -- takes it as well since round 5: the pass still counts the survivors of every batch, the
    // null counts per output batch are taken from the compacted outputs whoever wrote them)
    if (ctx->opt_groups_by_ranges >= 0 && n_rows >= rvt::kRangesFromRows && nterms >= 1 && nterms <= static_cast<uint32_t>(rvk::kMaxTerms)) {
        const double kept = expected_selectivity(ctx, cols, ncols, terms, nterms, policy, ex);  // (a first call over a big table: the sample, now)
        defer_plain = kept >= 0.0 && kept <= rvt::kDeferPlainUpTo;  // (tools/wide_ab.py sweep: 10-15 % faster at 10 and 20 % kept, a wash from 30 % on)
    }
    if (ctx->opt_groups_by_ranges == 1) defer_plain = true;  // (tests: whatever the size and the selectivity)
    // NULLABLE columns the predicate does not read are left to it at every selectivity: the passes that carry columns with output
    // bitmaps are the weakest launches there are (direct kernel with bitmaps 3.5 TB/s; tools/wide_ab.py nullable always sweep, 2e8
    // rows, kept 30 / 50 / 84 %: three columns 1.59 / 1.77 / 2.05 -> 1.14 / 1.30 / 1.63 ms, nine 4.24 / 5.20 / 6.22 -> 3.71 / 4.33 / 5.08)
    const bool defer_nullable_always = ctx->opt_groups_by_ranges >= 0 && !after_launch && n_rows >= rvt::kRangesFromRows;
    // A predicate that is ONE Boolean column (`mask is true`: RecordBatch::filter, the reference's streaming filter) over a big table,
    // sparse or with nullable columns: no chained pass at all -- mask_select_kernel + a scan of its counts stand in for it.
    // A WINDOW of RecordBatches (`req`: rv_filter_project_chunked / _batches, the reference's streaming filter at its 1024-row batches,
    // stream.rs:136-158) takes it too when a batch is a whole number of 1024-row ranges: the counts mask_select_kernel leaves per 1024
    // rows ARE the per-batch survivor counts.  String / Boolean columns projected next to the value columns (`after_launch`, `ranges`)
    // are queued at the scan's offsets once the value columns are on their way.
    bool mask_path = false;
    if (ctx->opt_groups_by_ranges >= 0 && (!req || (req->counts && req->chunk_rows % 1024 == 0)) && !ex && nterms == 1 && terms[0].op == RV_IS_TRUE && policy == RV_NULL_DROPS &&
        terms[0].column < ncols && cols[terms[0].column]->dtype == RV_BOOLEAN && (n_rows >= rvt::kRangesFromRows || (ctx->opt_groups_by_ranges == 1 && n_rows > 0)) &&
        (nproj >= 1 || (after_launch && ranges))) {
        mask_path = true;
        bool any_plain = false;
        for (uint32_t j = 0; j < nproj && mask_path; ++j) {
            mask_path = proj[j] < ncols && is_value_type(cols[proj[j]]->dtype);
            any_plain = any_plain || (mask_path && !cols[proj[j]]->validity);
        }
        // plain columns of a dense selection are better off in the direct kernel's pass (known from the predicate's last run only)
        const double kept = ctx->seen_selectivity(predicate_signature(cols, ncols, terms, nterms, policy, ex));
        if (mask_path && any_plain && kept > rvt::kMaskPathPlainUpTo && ctx->opt_groups_by_ranges != 1) mask_path = false;
        if (mask_path) defer_plain = true;
    }
    std::vector<uint32_t> late, late_pos;
    for (uint32_t j = 0; j < nproj; ++j) {
        require(proj[j] < ncols, RV_ERR_INVALID_ARG, fmt("projection %u references column %u of %u", j, proj[j], ncols));
        const rv_dcolumn *pc = cols[proj[j]];
        if ((defer_plain || (defer_nullable_always && pc->validity)) && is_value_type(pc->dtype) && (!pc->validity || !after_launch || mask_path) && !pred_value[proj[j]]) {
            late.push_back(proj[j]);
            late_pos.push_back(j);
            continue;
        }
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
    // (the first group may be empty: the pass then only evaluates the predicate -- selection bitmap, wave offsets, survivor count -- as
    // rv_eval_predicate's does.  Not when the predicate reads no 8-byte column at all (RecordBatch::filter by a BooleanArray): a pass
    // over nothing but a bitmap still walks every tile through the chain, 0.60 ms per 5e8 rows -- as long as one that carries a
    // column; measured: b is true -> [x] 1.25 ms that way against 0.79)
    if (pred_vals == 0 && !mask_path && groups.size() == 1 && groups[0].empty() && !late.empty()) {
        groups[0].push_back(late.front());
        group_pos[0].push_back(late_pos.front());
        late.erase(late.begin());
        late_pos.erase(late_pos.begin());
    }
    for (size_t k = 0; k < late.size(); ++k) {
        if (k % rvk::kRangesMaxCols == 0) {
            groups.emplace_back();
            group_pos.emplace_back();
        }
        groups.back().push_back(late[k]);
        group_pos.back().push_back(late_pos[k]);
    }
    const bool multi = groups.size() > 1;
    rv_dcolumn *sel = nullptr;
    std::vector<rv_dcolumn *> tmp(nproj ? nproj : 1, nullptr);
    uint64_t rows = 0;
    try {
        if (req && multi) req->sel_optional = false;  // later groups read the selection bitmap
        RangeOffsets own_ranges;  // the first pass's wave offsets: the later groups are compacted at them (ranges_kernel.hpp)
        RangeOffsets *first_ranges = ranges ? ranges : (multi ? &own_ranges : nullptr);
        const uint64_t reruns_before = ctx->overflow_reruns;
        // kernels queued behind the pass for the deferred groups: the call returns when they have run (a caller on another stream --
        // rv_device_ptrs, rv_ctx_stream -- sees finished columns, a device fault surfaces in THIS call, and `sel` / the wave offsets go
        // back to the pool behind their last reader); with option profile_kernels their device time counts as the call's
        bool late_launched = false;
        auto before_late_launch = [&] {
            if (!late_launched && ctx->opt_profile) RV_HIP(hipEventRecord(ctx->evk0, ctx->stream));
            late_launched = true;
        };
        auto after_late_launches = [&] {
            if (!late_launched) return;
            if (ctx->opt_profile) RV_HIP(hipEventRecord(ctx->evk1, ctx->stream));
            RV_HIP(hipStreamSynchronize(ctx->stream));
            if (ctx->opt_profile) {
                float ms = 0.f;
                RV_HIP(hipEventElapsedTime(&ms, ctx->evk0, ctx->evk1));
                ctx->kernel_ms += ms;
            }
            late_launched = false;
        };
        bool mask_ran = false;
        uint64_t rows_assumed = 0;  // the mask path's outputs were sized before its survivor count was known: for this many rows
        if (mask_path && groups[0].empty()) {
            const rv_dcolumn *mask = cols[terms[0].column];
            const uint64_t nwords = (n_rows + 63) / 64, nranges = (n_rows + 1023) / 1024;
            auto s = std::make_unique<rv_dcolumn>();
            s->dtype = RV_BOOLEAN;
            s->length = n_rows;
            s->null_count = 0;
            s->values = pool_alloc(ctx, std::max<size_t>(bitmap_words_bytes(n_rows) + 8, 16));
            DevBufRef counts = pool_alloc(ctx, nranges * 4 + 16);
            rvk::MaskSelect q{};
            q.values = static_cast<const uint8_t *>(mask->values->ptr);
            q.values_bytes = mask->values->bytes;
            q.validity = mask->validity ? static_cast<const uint8_t *>(mask->validity->ptr) : nullptr;
            q.validity_bytes = mask->validity ? mask->validity->bytes : 0;
            q.offset = mask->offset;
            q.n = n_rows;
            q.sel = static_cast<uint64_t *>(s->values->ptr);
            q.counts = static_cast<uint32_t *>(counts->ptr);
            // the reference's 1024-row batches ARE the ranges: their counts go straight to the caller's (pinned) array, 8 bytes per batch
            q.batch_counts = (req && req->counts && req->chunk_rows == 1024) ? req->counts : nullptr;
            before_late_launch();  // (option profile_kernels: the mask path's kernels -- selection, scan, compaction -- are the call's device time)
            hipLaunchKernelGGL(rvk::mask_select_kernel, dim3(static_cast<uint32_t>((nwords + 255) / 256)), dim3(256), 0, ctx->stream, q);
            RV_HIP(hipGetLastError());
            if (req && req->counts && req->chunk_rows != 1024) {  // the survivors of every batch: sums of the counts per 1024 rows, written where the caller reads them
                const uint64_t per_batch = req->chunk_rows / 1024;
                const uint64_t threads = per_batch < 32 ? req->nb : (per_batch < 4096 ? req->nb * 64 : req->nb * 256);
                const dim3 cgrid(static_cast<uint32_t>(std::max<uint64_t>(1, std::min<uint64_t>((threads + 255) / 256, static_cast<uint64_t>(ctx->props.multiProcessorCount) * 8))));
                hipLaunchKernelGGL(rvk::batch_counts_from_waves, cgrid, dim3(256), 0, ctx->stream, static_cast<const uint32_t *>(q.counts), nranges, per_batch, req->nb, req->counts);
                RV_HIP(hipGetLastError());
            }
            if (req && req->counts) {
                req->counted = true;
                ctx->batch_counts_in_pass += 1;
            }
            own_ranges.range_rows = 1024;
            // The outputs are sized by the survivor count, which the scan delivers -- a host round trip with the device idle (20-30 us of
            // a 0.4 ms window).  A predicate the context has run over these buffers before (a stream's windows) sizes them from what it
            // kept then (x 1.2 + 2 % of the rows), queues the columns' compaction right behind the scan, and reads the count at the end;
            // more survivors than that: the compaction runs once more with outputs of the exact size.  Plain columns only (a nullable
            // one's null count shares the control block with the scan's total; String / Boolean columns are sized by the exact count).
            const double known = ctx->seen_selectivity(predicate_signature(cols, ncols, terms, nterms, policy, ex));
            bool all_plain = !after_launch && groups.size() > 1 && ctx->opt_out_sizing >= 0;
            for (size_t g = 1; g < groups.size(); ++g)
                for (uint32_t c : groups[g]) all_plain = all_plain && is_value_type(cols[c]->dtype) && !cols[c]->validity;
            if (all_plain && known >= 0.0 && known <= rvt::kMaskPathAssumeUpTo && n_rows >= rvt::kRangesFromRows) {
                rows_assumed = std::min<uint64_t>(n_rows, static_cast<uint64_t>(static_cast<double>(n_rows) * (known * rvt::kOutSizingFactor + rvt::kOutSizingSlack)) + 4096);
                device_exclusive_scan(ctx, counts->ptr, nranges, own_ranges.offsets, false, false);  // (queued; the total stays in the control block)
                rows = rows_assumed;
            } else {
                rows = device_exclusive_scan(ctx, counts->ptr, nranges, own_ranges.offsets, false, true);  // (waits: the outputs are sized by it)
            }
            own_ranges.out_capacity = rows;
            own_ranges.expected_selectivity = n_rows ? static_cast<double>(rows) / static_cast<double>(n_rows) : 0.0;
            first_ranges = &own_ranges;
            if (ranges) *ranges = own_ranges;  // the caller's String / Boolean columns are compacted at the same offsets
            mask_ran = true;
            sel = s.release();
            if (!rows_assumed) ctx->remember_selectivity(predicate_signature(cols, ncols, terms, nterms, policy, ex), own_ranges.expected_selectivity);
            ctx->last_kernel = "mask_select_kernel";
        } else {
            // (a table sorted on the predicate's column, a query of one plain pass: stretch by stretch, fused_launch.hip)
            const bool simple = !multi && !out_selection && !req && !after_launch && !first_ranges && !groups[0].empty();
            if (!simple || !run_segmented_pass(ctx, cols, ncols, terms, nterms, policy, groups[0].data(), static_cast<uint32_t>(groups[0].size()), tmp.data(), ex, &rows))
            rows = run_fused_pass(ctx, cols, ncols, terms, nterms, policy, groups[0].data(), static_cast<uint32_t>(groups[0].size()),
                                  tmp.data(), (multi || out_selection) ? &sel : nullptr, ex, req, after_launch, first_ranges);
        }
        for (size_t k = 0; k < groups[0].size(); ++k) out[group_pos[0][k]] = tmp[k];
        // (offsets of a pass that overflowed its outputs and was re-run are not the re-run's: those groups take the pass path)
        // (and up to 55 % of the rows surviving -- tools/wide_ab.py, nine columns of 2e8 rows, groups beyond the first: 2.47 against 2.88 ms
        // at 10 %, 3.71 / 3.83 at 50 %, 4.67 / 4.46 at 84 %: there the direct kernel's whole-line stores win; option groups_by_ranges = 1: always)
        const bool offsets_there = first_ranges && first_ranges->offsets && ctx->overflow_reruns == reruns_before && ctx->opt_groups_by_ranges >= 0 && sel && sel->length > 0;
        const bool sparse_enough = ctx->opt_groups_by_ranges == 1 || (sel && rows * rvt::kRangesSparseDen <= sel->length * rvt::kRangesSparseNum);
        // nullable columns among them: their validity bits by bits_compact_kernel at the same offsets, their null counts out of the
        // context's control block (eight counters per read-back) -- not next to a caller's own launches on that block (after_launch)
        struct LateNulls {
            rv_dcolumn *col;
            int slot;
        };
        std::vector<LateNulls> late_nulls;
        Ctrl *late_ctrl = nullptr;
        auto finish_late_nulls = [&]() {
            if (late_nulls.empty()) return;
            const Ctrl fetched = *fetch_ctrl(ctx);
            for (const LateNulls &q : late_nulls) {
                q.col->null_count = static_cast<int64_t>(rows - fetched.valid_pop[q.slot]);
                if (q.col->null_count == 0) q.col->validity.reset();  // the builder drops the bitmap when no null survived (primitive.rs:185-197)
            }
            late_nulls.clear();
            late_ctrl = nullptr;
        };
        auto late_groups = [&] {
        for (size_t g = 1; g < groups.size(); ++g) {
            bool plain = offsets_there && groups[g].size() <= static_cast<size_t>(rvk::kRangesMaxCols);
            bool any_nulls = false;
            for (uint32_t c : groups[g]) {
                plain = plain && is_value_type(cols[c]->dtype) && (!cols[c]->validity || mask_ran || !after_launch);
                any_nulls = any_nulls || cols[c]->validity != nullptr;
            }
            // a dense selection: plain columns go through the direct kernel (5 % ahead); NULLABLE ones stay here -- the pass by a
            // Boolean predicate has no dense geometry for columns that keep nulls, and fell back to the default one + fused_redo_tiles
            // (nine nullable columns of 2e8 rows at 84 %: 18.4 ms)
            if (plain && (sparse_enough || any_nulls)) {
                rvk::RangesCompact q{};
                q.sel = static_cast<const uint64_t *>(sel->values->ptr);
                q.nwords = (sel->length + 63) / 64;
                q.n = sel->length;
                q.range_offsets = static_cast<const uint64_t *>(first_ranges->offsets->ptr);
                q.range_rows = first_ranges->range_rows;
                q.out_capacity = rows;
                struct GroupNull {
                    const rv_dcolumn *src;
                    rv_dcolumn *col;
                    int slot;
                };
                std::vector<GroupNull> group_nulls;
                // (a group's nullable columns share one read-back of the control block: make room for all of them first)
                size_t group_nullable = 0;
                for (uint32_t c : groups[g]) group_nullable += cols[c]->validity && rows ? 1 : 0;
                if (late_nulls.size() + group_nullable > 8) finish_late_nulls();
                before_late_launch();
                for (size_t k = 0; k < groups[g].size(); ++k) {
                    const rv_dcolumn *src = cols[groups[g][k]];
                    auto o = std::make_unique<rv_dcolumn>();
                    o->dtype = src->dtype;
                    o->length = rows;
                    o->null_count = 0;
                    o->values = pool_alloc(ctx, std::max<size_t>(elem_bytes(src->dtype, rows), 8));
                    q.in[k] = static_cast<const char *>(src->values->ptr) + src->offset * 8;
                    q.out[k] = static_cast<uint64_t *>(o->values->ptr);
                    if (src->validity && rows) {
                        q.validity[k] = static_cast<const uint8_t *>(src->validity->ptr);
                        q.validity_bytes[k] = src->validity->bytes;
                        q.bit_offset[k] = src->offset;
                        if (!late_ctrl) late_ctrl = prepare_ctrl(ctx, 0);
                        const int slot = static_cast<int>(late_nulls.size());
                        const size_t wb = zeroed_bitmap_bytes(rows);
                        o->validity = pool_alloc(ctx, wb);
                        RV_HIP(hipMemsetAsync(o->validity->ptr, 0, wb, ctx->stream));
                        group_nulls.push_back(GroupNull{src, o.get(), slot});
                        late_nulls.push_back(LateNulls{o.get(), slot});
                    }
                    out[group_pos[g][k]] = o.release();
                }
                // the validity bits of the group's nullable columns, two columns per launch where they share their bit offset
                for (size_t a = 0; a < group_nulls.size();) {
                    const GroupNull &x0 = group_nulls[a];
                    const bool two = a + 1 < group_nulls.size() && group_nulls[a + 1].src->offset == x0.src->offset;
                    rvk::BitsCompact b{};
                    b.sel = q.sel;
                    b.nwords = q.nwords;
                    b.offset = x0.src->offset;
                    b.range_offsets = q.range_offsets;
                    b.range_rows = q.range_rows;
                    b.out_capacity = rows;
                    b.src = static_cast<const uint8_t *>(x0.src->validity->ptr);
                    b.src_bytes = x0.src->validity->bytes;
                    b.out = static_cast<uint64_t *>(x0.col->validity->ptr);
                    b.pop = striped(ctx, &late_ctrl->valid_pop[x0.slot]);
                    if (two) {
                        const GroupNull &x1 = group_nulls[a + 1];
                        b.src2 = static_cast<const uint8_t *>(x1.src->validity->ptr);
                        b.src2_bytes = x1.src->validity->bytes;
                        b.out2 = static_cast<uint64_t *>(x1.col->validity->ptr);
                        b.pop2 = striped(ctx, &late_ctrl->valid_pop[x1.slot]);
                    }
                    const dim3 bgrid(static_cast<uint32_t>(std::min<uint64_t>((q.nwords + 255) / 256, static_cast<uint64_t>(ctx->props.multiProcessorCount) * 8)));
                    hipLaunchKernelGGL(rvk::bits_compact_kernel, bgrid, dim3(256), 0, ctx->stream, b);
                    RV_HIP(hipGetLastError());
                    a += two ? 2 : 1;
                }
                if (rows) {
                    const dim3 grid(static_cast<uint32_t>((q.nwords + 63) / 64)), block(256);  // a wave per 16 words, four per workgroup
                    const int pick = static_cast<int>(groups[g].size()) * 2 + (any_nulls ? 1 : 0);
                    switch (pick) {
                        case 2: hipLaunchKernelGGL((rvk::compact_ranges_kernel<1, false>), grid, block, 0, ctx->stream, q); break;
                        case 3: hipLaunchKernelGGL((rvk::compact_ranges_kernel<1, true>), grid, block, 0, ctx->stream, q); break;
                        case 4: hipLaunchKernelGGL((rvk::compact_ranges_kernel<2, false>), grid, block, 0, ctx->stream, q); break;
                        case 5: hipLaunchKernelGGL((rvk::compact_ranges_kernel<2, true>), grid, block, 0, ctx->stream, q); break;
                        case 6: hipLaunchKernelGGL((rvk::compact_ranges_kernel<3, false>), grid, block, 0, ctx->stream, q); break;
                        case 7: hipLaunchKernelGGL((rvk::compact_ranges_kernel<3, true>), grid, block, 0, ctx->stream, q); break;
                        case 8: hipLaunchKernelGGL((rvk::compact_ranges_kernel<4, false>), grid, block, 0, ctx->stream, q); break;
                        default: hipLaunchKernelGGL((rvk::compact_ranges_kernel<4, true>), grid, block, 0, ctx->stream, q); break;
                    }
                    RV_HIP(hipGetLastError());
                    ctx->last_kernel = fmt("compact_ranges_kernel<%d>", static_cast<int>(groups[g].size()));
                }
                continue;
            }
            // later groups: predicate == the materialised selection bitmap
            after_late_launches();
            finish_late_nulls();
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
        };
        late_groups();
        if (rows_assumed) {  // the count the scan left in the control block: fits what the outputs were sized for, or the compaction runs again
            after_late_launches();
            const uint64_t counted = fetch_ctrl(ctx)->pops[0];
            const bool fits = counted <= rows_assumed;
            rows = counted;
            own_ranges.out_capacity = rows;
            for (size_t g = 1; g < groups.size(); ++g)
                for (size_t k = 0; k < groups[g].size(); ++k) {
                    rv_dcolumn *&o = out[group_pos[g][k]];
                    if (fits) {
                        if (o) o->length = rows;
                    } else {
                        delete o;
                        o = nullptr;
                    }
                }
            if (!fits) {
                ctx->overflow_reruns += 1;
                late_groups();
            }
            ctx->remember_selectivity(predicate_signature(cols, ncols, terms, nterms, policy, ex), n_rows ? static_cast<double>(rows) / static_cast<double>(n_rows) : 0.0);
        }
        {
            const bool waits = late_launched;
            after_late_launches();
            if (mask_ran && req && req->counted && !waits) RV_HIP(hipStreamSynchronize(ctx->stream));  // the per-batch counts are the caller's to read on return
        }
        finish_late_nulls();
        // the caller's launches behind "the pass" (String / Boolean columns at the scan's offsets): the value columns' null counts
        // have been read, so the context's one control block is theirs now
        if (mask_ran && after_launch && *after_launch) (*after_launch)(sel);
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
uint64_t filter_query(rv_ctx *ctx, const rv_dcolumn *const *cols, uint32_t ncols, const rv_predicate *pred, const uint32_t *proj,
                      uint32_t nproj, rv_dcolumn **out, rv_dcolumn **out_selection, BatchReq *req) {
    Normalized nz;
    normalize_predicate(ctx, cols, ncols, pred, nz);
    return filter_by_groups(ctx, nz.cols.data(), static_cast<uint32_t>(nz.cols.size()), nz.terms.data(), static_cast<uint32_t>(nz.terms.size()),
                            pred->nulls, proj, nproj, out, out_selection, nz.expr(), req);
}
}  // namespace rvl
extern "C" {

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

}  // extern "C"

namespace rvl {
struct BatchWalk;
struct MaskWindow;
}
struct rv_pending {
    FusedLaunch launch;                 // valid when !done
    std::vector<rv_dcolumn *> outs;     // output handles (owned until finish hands them over)
    uint64_t rows = 0;
    bool done = false;                  // completed inside begin (several passes)
    // ---- a WINDOW of RecordBatches (rv_filter_project_chunked_begin / _batches_begin -> rv_filter_project_window_finish) ----
    bool window = false;
    rvl::BatchReq req;                  // per-batch survivor counts out of the pass, written to the caller's pinned array
    rv_dcolumn *sel = nullptr;          // ... or the selection bitmap they are counted from at finish
    uint64_t nb = 0, chunk_rows = 0;
    uint64_t *out_rows = nullptr;
    // handle form: the walk that validates the assumed (regular) window runs on `walker` until finish; the call's arguments for the
    // ordinary path, should it not confirm the assumption (the caller keeps them alive until finish)
    std::unique_ptr<rvl::MaskWindow> mask;  // the window runs on the mask path (a Boolean-column predicate): mask_window_begin / _finish
    std::unique_ptr<rvl::BatchWalk> walk;
    std::thread walker;
    std::vector<std::unique_ptr<rv_dcolumn>> views;
    uint64_t assumed_total = 0;
    const rv_dcolumn *const *cols = nullptr;
    uint32_t nbatches = 0, ncols = 0, nproj = 0;
    const rv_predicate *pred = nullptr;
    const uint32_t *proj = nullptr;
    ~rv_pending();
};

namespace rvl {
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
}  // namespace rvl

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
            if (!pred->expr && single_pass_shape(cols, ncols, pred->terms, pred->n_terms, proj, nproj)) {
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
            if (!pend->done) abandon_launch(ctx, pend->launch);  // the launch may still be running: drain before the buffers go
            for (auto *d : pend->outs) delete d;
            throw;
        }
        for (size_t j = 0; j < pend->outs.size(); ++j) out[j] = pend->outs[j];
        if (out_rows) *out_rows = pend->rows;
    });
}

// ---- many RecordBatches, one launch (seam S1 at the reference's batch size) ------------------------------------
}  // extern "C"

namespace rvl {
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
    // ... the WHOLE array: a caller that registered only part of it (or points near the end of a pinned block) gets the staging
    // path, not a device write past the pinned range
    void *range_base = nullptr;
    size_t range_bytes = 0;
    if (hipPointerGetAttributes(&attr, out_rows) == hipSuccess && attr.type == hipMemoryTypeHost && attr.devicePointer &&
        hipMemGetAddressRange(reinterpret_cast<hipDeviceptr_t *>(&range_base), &range_bytes, attr.devicePointer) == hipSuccess &&
        static_cast<char *>(range_base) + range_bytes >= static_cast<char *>(attr.devicePointer) + nb * 8) {
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
    DevBufRef d_bounds, d_counts;  // taken when a count is actually run (none is, with the counts out of the pass and no nulls asked for)
    auto need_tables = [&] {
        if (!d_counts) d_bounds = pool_alloc(ctx, (nb + 1) * 8), d_counts = pool_alloc(ctx, nb * 8);
    };
    std::vector<rvk::SegItem> items;
    DevBufRef d_items;
    // set bits of `words` per range of `b` -> dst (host), through segment_popcount_kernel
    auto segment_counts = [&](const uint64_t *words, const std::vector<uint64_t> &b, uint64_t *dst) {
        need_tables();
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
        need_tables();
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
}  // namespace rvl

namespace rvl {
// The walk over the K x ncols handles of a window of RecordBatches (each handle a separate heap object: prefetched a few batches
// ahead, or the walk is one cache miss per handle and caps 1024-row batches at ~6e9 rows/s; past a few thousand batches it is split
// over host threads).  It validates every batch (RecordBatch::try_new, record_batch.rs:31-40; one schema per stream, stream.rs:58-114)
// -- an error is reported for the FIRST offending batch, as by a sequential walk -- and notes whether the window is REGULAR: every
// batch a zero-copy slice right behind the one before, all of batch 0's length but a shorter last one.  What a stream hands over
// almost always is, so the first walk only validates and sums; only if that did not hold is it repeated recording length + adjacency.
struct BatchWalk {
    struct Result {
        uint32_t batch = UINT32_MAX;  // the first offending batch of the range, if any
        rv_status status = RV_OK;
        std::string text;
        bool regular = true;
        uint64_t rows = 0;
    };
    const rv_dcolumn *const *cols;
    uint32_t nbatches, ncols;
    uint64_t len0;
    std::vector<Result> results;
    std::vector<uint64_t> lens;
    std::vector<uint8_t> adj;
    BatchWalk(const rv_dcolumn *const *cols_, uint32_t nbatches_, uint32_t ncols_) : cols(cols_), nbatches(nbatches_), ncols(ncols_), len0(cols_[0]->length) {}
    void walk(uint32_t b0, uint32_t b1, Result &res, bool record) {
        // (measured and not kept: 16 walk threads with both of a handle's cache lines prefetched 16 batches ahead -- 0.56 ms per window of
        // 262 144 batches against 0.39 with 8 threads, one line, 8 batches ahead: the box gives a process 16 hardware threads, and the pass's
        // own host thread and the Python caller want theirs)
        const size_t nhandles = static_cast<size_t>(nbatches) * ncols, ahead = 8 * static_cast<size_t>(ncols);
        auto fetch = [](const rv_dcolumn *h) { __builtin_prefetch(h); };
        for (size_t i = static_cast<size_t>(b0) * ncols; i < std::min(nhandles, static_cast<size_t>(b0) * ncols + ahead); ++i) fetch(cols[i]);
        bool regular = true;
        uint64_t rows = 0;
        for (uint32_t b = b0; b < b1; ++b) {
            const rv_dcolumn *const *cur = cols + static_cast<size_t>(b) * ncols;
            auto fail = [&](rv_status st, std::string text) {
                res.batch = b;
                res.status = st;
                res.text = std::move(text);
            };
            for (uint32_t c = 0; c < ncols; ++c) {
                const size_t i = static_cast<size_t>(b) * ncols + c;
                if (i + ahead < nhandles) fetch(cols[i + ahead]);
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
            regular = regular && (adjacent || b == 0) && (len == len0 || (b + 1 == nbatches && len < len0));
            rows += len;
            if (record) {
                lens[b] = len;
                adj[b] = adjacent ? 1 : 0;
            }
        }
        res.regular = regular;
        res.rows = rows;
    }
    void walk_all(bool record) {
        const uint32_t nthreads = nbatches >= rvt::kWalkThreadsFromBatches ? std::min<uint32_t>(rvt::kWalkThreads, std::max<uint32_t>(1, std::thread::hardware_concurrency())) : 1;
        results.assign(nthreads, Result{});
        if (record) {
            lens.resize(nbatches);
            adj.assign(nbatches, 0);
        }
        if (nthreads == 1) {
            walk(0, nbatches, results[0], record);
        } else {
            std::vector<std::thread> pool;
            const uint32_t per = (nbatches + nthreads - 1) / nthreads;
            for (uint32_t t = 0; t < nthreads; ++t)
                pool.emplace_back([this, t, per, record] { walk(std::min(nbatches, t * per), std::min(nbatches, (t + 1) * per), results[t], record); });
            for (auto &th : pool) th.join();
        }
    }
    const Result *first_error() const {
        const Result *first = nullptr;
        for (auto &r : results)
            if (r.batch != UINT32_MAX && (!first || r.batch < first->batch)) first = &r;
        return first;
    }
    bool regular() const {
        bool ok = len0 > 0;
        for (auto &r : results) ok = ok && r.regular;
        return ok;
    }
    uint64_t total_rows() const {
        uint64_t t = 0;
        for (auto &r : results) t += r.rows;
        return t;
    }
};

// The window AS IF it were regular, from its first and last batch alone: one zero-copy view per column over all of it, checked against
// the buffers (a wrong assumption reads garbage, never out of bounds).  False when the two batches do not look like a regular window's.
bool assume_regular_window(const rv_dcolumn *const *cols, uint32_t nbatches, uint32_t ncols, std::vector<std::unique_ptr<rv_dcolumn>> &views, uint64_t &total) {
    const uint64_t len0 = cols[0]->length;
    if (len0 == 0) return false;
    const rv_dcolumn *const *last = cols + static_cast<size_t>(nbatches - 1) * ncols;
    for (uint32_t c = 0; c < ncols; ++c)
        if (last[c] == nullptr || cols[c] == nullptr) return false;
    const uint64_t last_len = last[0]->length;
    if (last_len == 0 || last_len > len0) return false;
    total = static_cast<uint64_t>(nbatches - 1) * len0 + last_len;
    for (uint32_t c = 0; c < ncols; ++c) {
        const rv_dcolumn *a = cols[c], *z = last[c];
        bool ok = a->length == len0 && z->length == last_len && z->dtype == a->dtype && z->values == a->values && z->validity == a->validity && z->offsets == a->offsets &&
                  z->offset == a->offset + static_cast<uint64_t>(nbatches - 1) * len0;
        if (!ok) return false;
        const uint64_t end = a->offset + total;  // elements / bits the assumed view reaches
        if (is_value_type(a->dtype)) ok = a->values && a->values->bytes / 8 >= end;
        else if (a->dtype == RV_BOOLEAN) ok = a->values && a->values->bytes * 8 >= end;
        else if (a->dtype == RV_STRING) ok = a->offsets && a->offsets->bytes / 4 >= end + 1;
        else ok = a->dtype == RV_NULL;
        if (!ok || (a->validity && a->validity->bytes * 8 < end)) return false;
    }
    views.clear();
    for (uint32_t c = 0; c < ncols; ++c) {
        auto v = std::make_unique<rv_dcolumn>(*cols[c]);
        v->length = total;
        v->null_count = cols[c]->dtype == RV_NULL ? static_cast<int64_t>(total) : (cols[c]->validity ? -1 : 0);
        views.emplace_back(std::move(v));
    }
    return true;
}

// rv_filter_project_batches (throws): also the ordinary path of a pipelined window whose speculation the walk did not confirm
void filter_project_batches_sync(rv_ctx *ctx, const rv_dcolumn *const *cols, uint32_t nbatches, uint32_t ncols, const rv_predicate *pred, const uint32_t *proj,
                                 uint32_t nproj, rv_dcolumn **out, uint64_t *out_rows, int64_t *out_nulls, uint64_t *out_total) {
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
        //      and RecordBatch::slice hand out, streaming.rs:135-233) are ONE batch as they lie in HBM ----------------------
        struct Run {
            uint32_t first, count;
            uint64_t rows;
        };
        std::vector<Run> runs;
        std::vector<uint64_t> bounds;
        for (uint32_t c = 0; c < ncols; ++c)  // every range compares its dtypes with batch 0's
            require(cols[c] != nullptr, RV_ERR_INVALID_ARG, "rv_filter_project_batches: a column handle is NULL");
        BatchWalk walk(cols, nbatches, ncols);
        const uint64_t len0 = walk.len0;
        // ---- optimistic run: what a stream hands over is almost always regular (BatchWalk), and then everything the query needs is
        //      known from the first and the last batch -- the views, the row total, the batch length.  So the query RUNS on that
        //      assumption, on this thread, while a helper thread walks the K x ncols handles (0.3-0.4 ms for 262 144 batches of a
        //      column: the size of the pass itself) and validates it; an error or an irregular window found by the walk drops the
        //      speculative result and takes the ordinary path (the error is the first offending batch's, as always; the caller's
        //      per-batch arrays hold nothing defined after an error).  Any query shape: the chained pass, the mask path of the
        //      reference's streaming filter (a Boolean column), String columns riding along.
        bool speculative = false;
        uint64_t spec_total = 0, spec_rows = 0;
        rv_dcolumn *spec_sel = nullptr;
        BatchReq spec_req;
        std::vector<std::unique_ptr<rv_dcolumn>> spec_views;
        auto drop_speculative = [&] {
            (void)hipStreamSynchronize(ctx->stream);
            for (uint32_t j = 0; j < nproj; ++j) {
                delete out[j];
                out[j] = nullptr;
            }
            delete spec_sel;
            spec_sel = nullptr;
            speculative = false;
        };
        bool walked = false;
        if (nbatches >= rvt::kSpeculateFromBatches && ctx->opt_speculative_batches >= 0 && assume_regular_window(cols, nbatches, ncols, spec_views, spec_total)) {
            std::vector<const rv_dcolumn *> views(ncols);
            for (uint32_t c = 0; c < ncols; ++c) views[c] = spec_views[c].get();
            spec_req = make_batch_req(ctx, len0, nbatches, out_rows);
            std::thread walker([&] { walk.walk_all(false); });
            std::exception_ptr spec_error;
            try {
                spec_rows = filter_query(ctx, views.data(), ncols, pred, proj, nproj, out, &spec_sel, &spec_req);
                speculative = true;
            } catch (...) {
                spec_error = std::current_exception();
            }
            walker.join();
            walked = true;
            const bool confirmed = walk.first_error() == nullptr && walk.regular() && walk.total_rows() == spec_total;
            if (!confirmed) drop_speculative();  // (whatever the speculative run did or threw: the ordinary path reports)
            else if (spec_error) {
                drop_speculative();
                std::rethrow_exception(spec_error);
            }
        }
        if (!walked) walk.walk_all(false);
        if (const BatchWalk::Result *first_error = walk.first_error()) throw Error(first_error->status, first_error->text);
        const bool regular = walk.regular();
        const uint64_t total_rows = walk.total_rows();
        // batches of one size (the last one may be shorter): no boundary table needed, and the pass itself can count the
        // survivors per batch
        uint64_t uniform = 0;
        if (regular) {
            runs.push_back(Run{0, nbatches, total_rows});
            uniform = len0;
        } else {
            bounds.assign(static_cast<size_t>(nbatches) + 1, 0);
            walk.walk_all(true);
            const std::vector<uint64_t> &lens = walk.lens;
            const std::vector<uint8_t> &adj = walk.adj;
            for (uint32_t b = 0; b < nbatches; ++b) {
                bounds[b + 1] = bounds[b] + lens[b];
                if (adj[b]) {
                    runs.back().count += 1;
                    runs.back().rows += lens[b];
                } else {
                    runs.push_back(Run{b, 1, lens[b]});
                }
            }
            uniform = bounds[1];
            for (uint32_t b = 1; b < nbatches && uniform; ++b) {
                const uint64_t len = bounds[b + 1] - bounds[b];
                if (len != uniform && !(b + 1 == nbatches && len < uniform)) uniform = 0;
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
        BatchReq req;
        uint64_t rows = 0;
        if (speculative) {  // ran beside the walk, on what the walk has now confirmed
            req = spec_req;
            sel = spec_sel;
            rows = spec_rows;
            ctx->speculative_batch_passes += 1;
        } else {
            req = make_batch_req(ctx, uniform, nbatches, out_rows);
            rows = filter_query(ctx, whole.data(), ncols, pred, proj, nproj, out, nbatches > 1 ? &sel : nullptr, (nbatches > 1 && uniform) ? &req : nullptr);
        }
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
}
}  // namespace rvl

extern "C" {

rv_status rv_filter_project_batches(rv_ctx *ctx, const rv_dcolumn *const *cols, uint32_t nbatches, uint32_t ncols, const rv_predicate *pred,
                                    const uint32_t *proj, uint32_t nproj, rv_dcolumn **out, uint64_t *out_rows, int64_t *out_nulls,
                                    uint64_t *out_total) {
    return guarded([&] { filter_project_batches_sync(ctx, cols, nbatches, ncols, pred, proj, nproj, out, out_rows, out_nulls, out_total); });
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

}  // extern "C"

namespace rvl {
// A window of 1024-row-multiple RecordBatches filtered by a BOOLEAN column (the reference's streaming filter, stream.rs:136-158) with its
// whole work QUEUED: mask_select_kernel (selection words, counts per 1024 rows -- the per-batch counts, written where the caller reads
// them), the scan of the counts (offsets; the total into the window's own control block) and the compaction of the plain value columns
// at those offsets into outputs sized from what the predicate kept the last time.  finish reads the total: it fits, or the
// compaction runs once more with outputs of the exact size.  (filter_by_groups' mask path is the same work with the host waiting for
// the scan's total in the middle; this is its form for a stream operator that keeps two windows in flight.)
struct MaskWindow {
    rv_ctx::LaunchCtrl ctrl;  // pops[0] <- the scan's total
    bool launched = false;
    std::unique_ptr<rv_dcolumn> sel;
    DevBufRef counts, offsets;
    std::vector<const rv_dcolumn *> src;  // the projected source columns (the caller keeps them alive until finish)
    uint64_t n = 0, assumed = 0, signature = 0;
};
static void mask_window_compact(rv_ctx *ctx, const MaskWindow &w, uint64_t cap, rv_dcolumn *const *outs) {
    rvk::RangesCompact q{};
    q.sel = static_cast<const uint64_t *>(w.sel->values->ptr);
    q.nwords = (w.n + 63) / 64;
    q.n = w.n;
    q.range_offsets = static_cast<const uint64_t *>(w.offsets->ptr);
    q.range_rows = 1024;
    q.out_capacity = cap;
    const dim3 grid(static_cast<uint32_t>((q.nwords + 63) / 64)), block(256);
    for (size_t g0 = 0; g0 < w.src.size(); g0 += rvk::kRangesMaxCols) {
        const size_t k = std::min<size_t>(rvk::kRangesMaxCols, w.src.size() - g0);
        for (size_t c = 0; c < k; ++c) {
            q.in[c] = static_cast<const char *>(w.src[g0 + c]->values->ptr) + w.src[g0 + c]->offset * 8;
            q.out[c] = static_cast<uint64_t *>(outs[g0 + c]->values->ptr);
        }
        switch (k) {
            case 1: hipLaunchKernelGGL((rvk::compact_ranges_kernel<1, false>), grid, block, 0, ctx->stream, q); break;
            case 2: hipLaunchKernelGGL((rvk::compact_ranges_kernel<2, false>), grid, block, 0, ctx->stream, q); break;
            case 3: hipLaunchKernelGGL((rvk::compact_ranges_kernel<3, false>), grid, block, 0, ctx->stream, q); break;
            default: hipLaunchKernelGGL((rvk::compact_ranges_kernel<4, false>), grid, block, 0, ctx->stream, q); break;
        }
        RV_HIP(hipGetLastError());
    }
    ctx->last_kernel = fmt("compact_ranges_kernel<%d>", static_cast<int>(std::min<size_t>(rvk::kRangesMaxCols, w.src.size())));
}
// eligible: `b is true` alone over a Boolean column, RV_NULL_DROPS, plain value columns projected, a window of kRangesFromRows rows and
// more in batches of a multiple of 1024 rows, counts the device can write, and a selectivity the context remembers for these buffers
static bool mask_window_eligible(rv_ctx *ctx, const rv_dcolumn *const *cols, uint32_t ncols, const rv_predicate *pred, const uint32_t *proj, uint32_t nproj,
                                 uint64_t chunk_rows, const BatchReq &req, double &known, uint64_t &signature) {
    if (pred->expr || pred->n_terms != 1 || pred->terms[0].op != RV_IS_TRUE || pred->nulls != RV_NULL_DROPS || pred->terms[0].column >= ncols) return false;
    if (cols[pred->terms[0].column]->dtype != RV_BOOLEAN || (cols[0]->length < rvt::kRangesFromRows && ctx->opt_groups_by_ranges != 1) || chunk_rows % 1024 != 0 || nproj == 0) return false;
    if (ctx->opt_groups_by_ranges < 0 || ctx->opt_out_sizing < 0 || !req.counts || static_cast<const void *>(req.counts) == ctx->h_stage) return false;
    for (uint32_t j = 0; j < nproj; ++j)
        if (proj[j] >= ncols || !is_value_type(cols[proj[j]]->dtype) || cols[proj[j]]->validity) return false;
    rv_term t = pred->terms[0];
    t.lit.i = 0;
    signature = predicate_signature(cols, ncols, &t, 1, pred->nulls, nullptr);
    known = ctx->seen_selectivity(signature);
    return known >= 0.0 && known <= rvt::kMaskPathAssumeUpTo;
}
static void mask_window_begin(rv_ctx *ctx, const rv_dcolumn *const *cols, const rv_predicate *pred, const uint32_t *proj, uint32_t nproj, BatchReq &req,
                              double known, uint64_t signature, MaskWindow &w, rv_dcolumn **outs) {
    const rv_dcolumn *mask = cols[pred->terms[0].column];
    w.n = mask->length;
    w.signature = signature;
    w.assumed = std::min<uint64_t>(w.n, static_cast<uint64_t>(static_cast<double>(w.n) * (known * rvt::kOutSizingFactor + rvt::kOutSizingSlack)) + 4096);
    const uint64_t nwords = (w.n + 63) / 64, nranges = (w.n + 1023) / 1024;
    w.sel = std::make_unique<rv_dcolumn>();
    w.sel->dtype = RV_BOOLEAN;
    w.sel->length = w.n;
    w.sel->null_count = 0;
    w.sel->values = pool_alloc(ctx, std::max<size_t>(bitmap_words_bytes(w.n) + 8, 16));
    w.counts = pool_alloc(ctx, nranges * 4 + 16);
    w.ctrl = acquire_launch_ctrl(ctx, 0, 0);  // zeroed on the stream
    w.launched = true;
    rvk::MaskSelect q{};
    q.values = static_cast<const uint8_t *>(mask->values->ptr);
    q.values_bytes = mask->values->bytes;
    q.validity = mask->validity ? static_cast<const uint8_t *>(mask->validity->ptr) : nullptr;
    q.validity_bytes = mask->validity ? mask->validity->bytes : 0;
    q.offset = mask->offset;
    q.n = w.n;
    q.sel = static_cast<uint64_t *>(w.sel->values->ptr);
    q.counts = static_cast<uint32_t *>(w.counts->ptr);
    q.batch_counts = req.chunk_rows == 1024 ? req.counts : nullptr;
    hipLaunchKernelGGL(rvk::mask_select_kernel, dim3(static_cast<uint32_t>((nwords + 255) / 256)), dim3(256), 0, ctx->stream, q);
    RV_HIP(hipGetLastError());
    if (req.chunk_rows != 1024) {
        const uint64_t per_batch = req.chunk_rows / 1024;
        const uint64_t threads = per_batch < 32 ? req.nb : (per_batch < 4096 ? req.nb * 64 : req.nb * 256);
        const dim3 cgrid(static_cast<uint32_t>(std::max<uint64_t>(1, std::min<uint64_t>((threads + 255) / 256, static_cast<uint64_t>(ctx->props.multiProcessorCount) * 8))));
        hipLaunchKernelGGL(rvk::batch_counts_from_waves, cgrid, dim3(256), 0, ctx->stream, static_cast<const uint32_t *>(q.counts), nranges, per_batch, req.nb, req.counts);
        RV_HIP(hipGetLastError());
    }
    req.counted = true;
    ctx->batch_counts_in_pass += 1;
    device_exclusive_scan(ctx, w.counts->ptr, nranges, w.offsets, false, false, &static_cast<Ctrl *>(w.ctrl.dev)->pops[0]);
    w.src.clear();
    for (uint32_t j = 0; j < nproj; ++j) {
        const rv_dcolumn *src = cols[proj[j]];
        auto o = std::make_unique<rv_dcolumn>();
        o->dtype = src->dtype;
        o->length = w.assumed;
        o->null_count = 0;
        o->values = pool_alloc(ctx, std::max<size_t>(elem_bytes(src->dtype, w.assumed), 8));
        outs[j] = o.release();
        w.src.push_back(src);
    }
    mask_window_compact(ctx, w, w.assumed, outs);
    RV_HIP(hipMemcpyAsync(w.ctrl.host, w.ctrl.dev, kCtrlBytes, hipMemcpyDeviceToHost, ctx->stream));
    RV_HIP(hipEventRecord(w.ctrl.ev, ctx->stream));
}
static uint64_t mask_window_finish(rv_ctx *ctx, MaskWindow &w, rv_dcolumn **outs, uint32_t nproj) {
    RV_HIP(hipEventSynchronize(w.ctrl.ev));
    const uint64_t rows = static_cast<const Ctrl *>(w.ctrl.host)->pops[0];
    release_launch_ctrl(ctx, w.ctrl);
    w.launched = false;
    if (rows > w.assumed) {  // more survivors than the outputs were sized for: the compaction once more, exact
        for (uint32_t j = 0; j < nproj; ++j) outs[j]->values = pool_alloc(ctx, std::max<size_t>(elem_bytes(outs[j]->dtype, rows), 8));
        mask_window_compact(ctx, w, rows, outs);
        RV_HIP(hipStreamSynchronize(ctx->stream));
        ctx->overflow_reruns += 1;
    }
    for (uint32_t j = 0; j < nproj; ++j) outs[j]->length = rows;
    ctx->remember_selectivity(w.signature, w.n ? static_cast<double>(rows) / static_cast<double>(w.n) : 0.0);
    ctx->last_selectivity = w.n ? static_cast<double>(rows) / static_cast<double>(w.n) : 0.0;
    ctx->last_rows_out = rows, ctx->last_rows_in = w.n;
    return rows;
}
static void mask_window_abandon(rv_ctx *ctx, MaskWindow &w) {
    if (!w.launched) return;
    (void)hipStreamSynchronize(ctx->stream);
    release_launch_ctrl(ctx, w.ctrl);
    w.launched = false;
}
}  // namespace rvl

rv_pending::~rv_pending() {
    if (walker.joinable()) walker.join();
    delete sel;
}

namespace rvl {
// `x is true` terms carry no literal: what the caller left in that field stays out of the selectivity memory's key (predicate.hip)
static std::vector<rv_term> plain_terms(const rv_predicate *pred) {
    std::vector<rv_term> t(pred->terms, pred->terms + pred->n_terms);
    for (rv_term &q : t)
        if (q.op == RV_IS_TRUE) q.lit.i = 0;
    return t;
}
// A window whose pass can be QUEUED (begin) and waited for later (finish): one chained pass, batches the pass can count itself,
// counts the device can write where the caller reads them.  Everything else -- several passes, String columns, the mask path of a
// Boolean-column predicate (no chained pass at all: filter_by_groups) -- completes inside begin.
static bool window_can_be_queued(rv_ctx *ctx, const rv_dcolumn *const *cols, uint32_t ncols, const rv_predicate *pred, const uint32_t *proj, uint32_t nproj,
                                 uint64_t nb, uint64_t chunk_rows, const BatchReq &req) {
    if (nb < 2 || pred->expr || !single_pass_shape(cols, ncols, pred->terms, pred->n_terms, proj, nproj)) return false;
    if (!req.counts || static_cast<const void *>(req.counts) == ctx->h_stage) return false;  // counts through the shared staging block: one window at a time
    const bool mask_shape = pred->n_terms == 1 && pred->terms[0].op == RV_IS_TRUE && pred->nulls == RV_NULL_DROPS && pred->terms[0].column < ncols &&
                            cols[pred->terms[0].column]->dtype == RV_BOOLEAN && (cols[0]->length >= rvt::kRangesFromRows || ctx->opt_groups_by_ranges == 1) && chunk_rows % 1024 == 0 &&
                            ctx->opt_groups_by_ranges >= 0;
    return !mask_shape;
}
// the window's null counts per output batch (and, where the pass did not count them, its survivors per batch): finish's half
static void window_counts(rv_ctx *ctx, rv_pending &pend, int64_t *out_nulls) {
    rv_dcolumn *const *outs = pend.outs.data();
    const uint32_t nproj = pend.nproj;
    if (pend.nb == 1) {
        pend.out_rows[0] = pend.rows;
        if (out_nulls)
            for (uint32_t j = 0; j < nproj; ++j) out_nulls[j] = outs[j]->dtype == RV_NULL ? static_cast<int64_t>(pend.rows) : std::max<int64_t>(0, outs[j]->null_count);
        return;
    }
    if (pend.nb < 2) return;
    const bool counted = pend.done || pend.req.counted;
    require(counted || pend.sel != nullptr, RV_ERR_INTERNAL, "per-batch counts: neither counted in the pass nor a selection bitmap to count");
    if (!pend.done && pend.req.counted) finish_batch_req(pend.req, pend.out_rows);
    batch_counts(ctx, counted ? nullptr : pend.sel, pend.rows, {}, pend.chunk_rows, static_cast<size_t>(pend.nb), outs, nproj, pend.out_rows, out_nulls);
}
}  // namespace rvl

extern "C" {

rv_status rv_filter_project_chunked_begin(rv_ctx *ctx, const rv_dcolumn *const *cols, uint32_t ncols, uint64_t chunk_rows, const rv_predicate *pred,
                                          const uint32_t *proj, uint32_t nproj, uint64_t *out_rows, uint64_t nchunks, rv_pending **out_pending) {
    return guarded([&] {
        require(ctx && cols && pred && pred->terms && (proj || nproj == 0) && out_pending, RV_ERR_INVALID_ARG, "rv_filter_project_chunked_begin: NULL argument");
        require(ncols >= 1 && chunk_rows >= 1, RV_ERR_INVALID_ARG, "rv_filter_project_chunked_begin: no columns / chunk_rows is 0");
        check_batch(cols, ncols);
        set_device(ctx);
        const uint64_t n = cols[0]->length, nb = (n + chunk_rows - 1) / chunk_rows;
        require(nb <= nchunks && (out_rows || nb == 0), RV_ERR_INVALID_ARG,
                fmt("rv_filter_project_chunked_begin: %llu chunks, room for %llu", static_cast<unsigned long long>(nb), static_cast<unsigned long long>(nchunks)));
        maybe_injected_failure(ctx);
        auto pend = std::make_unique<rv_pending>();
        pend->window = true;
        pend->nb = nb, pend->chunk_rows = chunk_rows, pend->out_rows = out_rows, pend->nproj = nproj;
        pend->outs.assign(nproj ? nproj : 1, nullptr);
        try {
            pend->req = make_batch_req(ctx, chunk_rows, nb, out_rows);
            double known = -1.0;
            uint64_t signature = 0;
            if (nb >= 2 && mask_window_eligible(ctx, cols, ncols, pred, proj, nproj, chunk_rows, pend->req, known, signature)) {
                pend->mask = std::make_unique<MaskWindow>();
                mask_window_begin(ctx, cols, pred, proj, nproj, pend->req, known, signature, *pend->mask, pend->outs.data());
            } else if (window_can_be_queued(ctx, cols, ncols, pred, proj, nproj, nb, chunk_rows, pend->req)) {
                const std::vector<rv_term> terms = plain_terms(pred);
                fused_begin(ctx, cols, ncols, terms.data(), pred->n_terms, pred->nulls, proj, nproj, pend->outs.data(), &pend->sel, pend->launch, nullptr, &pend->req, nullptr);
            } else {
                rv_dcolumn *sel = nullptr;
                pend->rows = filter_query(ctx, cols, ncols, pred, proj, nproj, pend->outs.data(), nb > 1 ? &sel : nullptr, nb > 1 ? &pend->req : nullptr);
                pend->sel = sel;
                if (nb > 1) {  // the survivors per batch now (the caller's array is complete on return); the null counts at finish
                    require(pend->req.counted || sel != nullptr, RV_ERR_INTERNAL, "per-batch counts: neither counted in the pass nor a selection bitmap to count");
                    if (pend->req.counted) finish_batch_req(pend->req, out_rows);
                    batch_counts(ctx, pend->req.counted ? nullptr : sel, pend->rows, {}, chunk_rows, static_cast<size_t>(nb), pend->outs.data(), nproj, out_rows, nullptr);
                }
                pend->done = true;
            }
        } catch (...) {
            if (pend->mask) mask_window_abandon(ctx, *pend->mask);
            abandon_launch(ctx, pend->launch);
            for (auto *d : pend->outs) delete d;
            throw;
        }
        *out_pending = pend.release();
    });
}

rv_status rv_filter_project_batches_begin(rv_ctx *ctx, const rv_dcolumn *const *cols, uint32_t nbatches, uint32_t ncols, const rv_predicate *pred,
                                          const uint32_t *proj, uint32_t nproj, uint64_t *out_rows, rv_pending **out_pending) {
    return guarded([&] {
        require(ctx && cols && pred && pred->terms && (proj || nproj == 0) && out_rows && out_pending, RV_ERR_INVALID_ARG, "rv_filter_project_batches_begin: NULL argument");
        require(nbatches >= 1 && ncols >= 1, RV_ERR_INVALID_ARG, "rv_filter_project_batches_begin: no batches / no columns");
        set_device(ctx);
        for (uint32_t c = 0; c < ncols; ++c) require(cols[c] != nullptr, RV_ERR_INVALID_ARG, "rv_filter_project_batches: a column handle is NULL");
        maybe_injected_failure(ctx);
        auto pend = std::make_unique<rv_pending>();
        pend->window = true;
        pend->nb = nbatches, pend->chunk_rows = cols[0]->length, pend->out_rows = out_rows, pend->nproj = nproj;
        pend->cols = cols, pend->nbatches = nbatches, pend->ncols = ncols, pend->pred = pred, pend->proj = proj;
        pend->outs.assign(nproj ? nproj : 1, nullptr);
        try {
            bool queued = false;
            if (ctx->opt_speculative_batches >= 0 && assume_regular_window(cols, nbatches, ncols, pend->views, pend->assumed_total)) {
                std::vector<const rv_dcolumn *> views(ncols);
                for (uint32_t c = 0; c < ncols; ++c) views[c] = pend->views[c].get();
                pend->req = make_batch_req(ctx, pend->chunk_rows, nbatches, out_rows);
                double known = -1.0;
                uint64_t signature = 0;
                if (nbatches >= 2 && mask_window_eligible(ctx, views.data(), ncols, pred, proj, nproj, pend->chunk_rows, pend->req, known, signature)) {
                    // the reference's streaming filter (a Boolean column): the window's whole mask path is queued on the assumed window
                    pend->walk = std::make_unique<BatchWalk>(cols, nbatches, ncols);
                    BatchWalk *w = pend->walk.get();
                    pend->walker = std::thread([w] { w->walk_all(false); });
                    pend->mask = std::make_unique<MaskWindow>();
                    mask_window_begin(ctx, views.data(), pred, proj, nproj, pend->req, known, signature, *pend->mask, pend->outs.data());
                    queued = true;
                } else if (window_can_be_queued(ctx, views.data(), ncols, pred, proj, nproj, nbatches, pend->chunk_rows, pend->req)) {
                    // the pass runs on the ASSUMED window while the walk over the handles validates it on a helper thread: finish joins it
                    pend->walk = std::make_unique<BatchWalk>(cols, nbatches, ncols);
                    BatchWalk *w = pend->walk.get();
                    pend->walker = std::thread([w] { w->walk_all(false); });
                    const std::vector<rv_term> terms = plain_terms(pred);
                    fused_begin(ctx, views.data(), ncols, terms.data(), pred->n_terms, pred->nulls, proj, nproj, pend->outs.data(), &pend->sel, pend->launch, nullptr, &pend->req, nullptr);
                    queued = true;
                }
            }
            if (!queued) {  // completed here, the ordinary way (null counts at finish, from the outputs)
                uint64_t total = 0;
                filter_project_batches_sync(ctx, cols, nbatches, ncols, pred, proj, nproj, pend->outs.data(), out_rows, nullptr, &total);
                pend->rows = total;
                pend->done = true;
            }
        } catch (...) {
            if (pend->walker.joinable()) pend->walker.join();
            if (pend->mask) mask_window_abandon(ctx, *pend->mask);
            abandon_launch(ctx, pend->launch);
            for (auto *d : pend->outs) delete d;
            throw;
        }
        *out_pending = pend.release();
    });
}

rv_status rv_filter_project_window_finish(rv_ctx *ctx, rv_pending *pending, rv_dcolumn **out, int64_t *out_nulls, uint64_t *out_total) {
    return guarded([&] {
        require(ctx && pending, RV_ERR_INVALID_ARG, "rv_filter_project_window_finish: NULL argument");
        std::unique_ptr<rv_pending> pend(pending);
        require(pend->window, RV_ERR_INVALID_ARG, "rv_filter_project_window_finish: not a window's pending handle");
        set_device(ctx);
        try {
            require(out || pend->nproj == 0, RV_ERR_INVALID_ARG, "rv_filter_project_window_finish: out is NULL");
            if (pend->walker.joinable()) pend->walker.join();
            if (pend->walk && !(pend->walk->first_error() == nullptr && pend->walk->regular() && pend->walk->total_rows() == pend->assumed_total)) {
                // the walk did not confirm what the pass was launched on: its result is dropped, the ordinary path reports (or runs)
                if (pend->mask) {
                    mask_window_abandon(ctx, *pend->mask);
                    pend->mask.reset();
                }
                abandon_launch(ctx, pend->launch);
                for (auto *&d : pend->outs) {
                    delete d;
                    d = nullptr;
                }
                filter_project_batches_sync(ctx, pend->cols, pend->nbatches, pend->ncols, pend->pred, pend->proj, pend->nproj, out, pend->out_rows, out_nulls, out_total);
                return;
            }
            if (pend->mask) {
                pend->rows = mask_window_finish(ctx, *pend->mask, pend->outs.data(), pend->nproj);
                pend->done = true;  // (the per-batch counts are in place; the null counts below are zeros: plain columns)
                if (pend->walk) ctx->speculative_batch_passes += 1;
            } else if (!pend->done) {
                pend->rows = fused_finish(ctx, pend->launch);
                if (pend->walk) ctx->speculative_batch_passes += 1;
            }
            window_counts(ctx, *pend, out_nulls);
        } catch (...) {
            if (pend->mask) mask_window_abandon(ctx, *pend->mask);
            if (!pend->done) abandon_launch(ctx, pend->launch);
            for (auto *d : pend->outs) delete d;
            throw;
        }
        for (uint32_t j = 0; j < pend->nproj; ++j) out[j] = pend->outs[j];
        if (out_total) *out_total = pend->rows;
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
