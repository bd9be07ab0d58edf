// The fused filter + compact launch: shape -> instantiation, LDS and output sizing, launch, read-back, redo and overflow re-run.
// One unit of the backend library behind include/rivulus_gpu.h (gfx950 only; compiled with hipcc).  Shared helpers and the
// functions the units call across each other are declared in launch.hpp (namespace rvl).
#include "launch.hpp"

#include <cmath>

using namespace rvh;
using namespace rvl;

namespace rvl {

// ---------------------------------------------------------------------------------------
// fused launch
// ---------------------------------------------------------------------------------------
// Smallest instantiation whose feature flags cover `need`; for one-column lean/validity
// launches the geometry can be steered with rv_ctx_set_option("rows_per_lane", R | waves << 8).
const rvk::FusedEntry *find_fused(rv_ctx *ctx, int ncols, int vec, int need, int roomy = 0, int below_r = 1 << 30) {
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
            if (ctx->opt_rows_per_lane > 0) {
                const int want_r = static_cast<int>(ctx->opt_rows_per_lane & 0xFF);
                const int want_w = static_cast<int>((ctx->opt_rows_per_lane >> 8) & 0xFF);
                wanted = e.r == want_r && (want_w == 0 || e.waves == want_w);
            }
            // roomy 1: a 16-wave geometry with fewer rows per lane -- its LDS slots hold a larger share of a wave's
            // rows; roomy 2: the 8-wave geometries of fused_roomy.hip, whose slots hold every row of a wave
            if (roomy == 1 && ctx->opt_rows_per_lane <= 0 && ncols >= 1)  // ... among those with fewer than below_r rows per lane, the largest
                wanted = best && e.waves == 16 && e.r < below_r && (best->waves != 16 || best->r >= below_r || e.r > best->r);
            if (roomy == 2 && ctx->opt_rows_per_lane <= 0 && ncols >= 1) wanted = best && (e.waves < best->waves || (e.waves == best->waves && e.r < best->r));
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
        t = rvk::fused_entries_roomy(&n), scan(t, n);
        vec = 1;  // every feature set exists with 8-byte loads
    }
    return best;
}
// `prefer`: shape flags worth having when an instantiation exists (FF_PROJALL)
const rvk::FusedEntry &pick_fused(rv_ctx *ctx, int ncols, int vec, int need, int prefer = 0, int roomy = 0, int below_r = 1 << 30) {
    // the refinements the launch qualifies for, dropped one by one (FF_NONULL first) until an instantiation exists
    const rvk::FusedEntry *best = nullptr;
    for (const int pf : {prefer, prefer & ~rvk::FF_NONULL}) {
        if (best || !pf) continue;
        best = find_fused(ctx, ncols, vec, need | pf, roomy, below_r);
        if (best && (best->flags & ~(need | pf)) != 0) best = nullptr;  // not at the price of features the launch does not need
    }
    if (!best) best = find_fused(ctx, ncols, vec, need, roomy, below_r);
    require(best != nullptr, RV_ERR_INTERNAL, fmt("no fused kernel variant for %d columns, flags %d", ncols, need));
    return *best;
}

// Selectivity of the lowered predicate `in` over a strided sample of its rows; -1 when the sample could not be taken.
double sample_selectivity(rv_ctx *ctx, const rvk::ScanInputs &in, int nvals) {
    constexpr uint32_t kBlocks = 1024, kBlockRows = 1024;
    const rvk::SampleFn fn = rvk::sample_kernel(nvals);
    if (!fn || in.n < static_cast<uint64_t>(kBlocks) * kBlockRows) return -1.0;
    if (!ctx->d_sample) {
        void *d = nullptr, *h = nullptr;
        constexpr size_t kBytes = rvk::kSampleWords * 8;
        if (hipMalloc(&d, kBytes) != hipSuccess || hipHostMalloc(&h, kBytes, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            if (d) (void)hipFree(d);
            return -1.0;
        }
        RV_HIP(hipMemset(d, 0, kBytes));
        std::memset(h, 0, kBytes);
        ctx->d_sample = static_cast<unsigned long long *>(d);
        ctx->h_sample = static_cast<volatile unsigned long long *>(h);
    }
    rvk::SampleParams sp{};
    sp.in = in;
    sp.stride = (in.n / kBlocks) & ~uint64_t(63);
    sp.dev_words = ctx->d_sample;
    sp.host_words = ctx->h_sample;
    sp.sequence = ++ctx->sample_seq;
    hipLaunchKernelGGL(fn, dim3(kBlocks), dim3(256), 0, ctx->stream, sp);
    RV_HIP(hipGetLastError());
    // the last workgroup writes {survivors, sequence} into pinned memory: spin on it; after 2 ms fall back to draining the stream
    const auto t0 = std::chrono::steady_clock::now();
    while (ctx->h_sample[1] != sp.sequence) {
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) {
            RV_HIP(hipStreamSynchronize(ctx->stream));
            break;
        }
    }
    if (ctx->h_sample[1] != sp.sequence) return -1.0;
    ctx->samples_taken += 1;
    for (int b = 0; b < rvk::kSampleBuckets; ++b)
        ctx->last_sample_hist[b] = static_cast<float>((ctx->h_sample[2 + b / 4] >> (16 * (b % 4))) & 0xFFFF) / static_cast<float>(kBlocks);
    static_assert(kBlocks == rvk::kSampleBlocks, "sample geometry");
    for (uint32_t b = 0; b < kBlocks; ++b)
        ctx->last_sample_profile[b] = static_cast<uint16_t>((ctx->h_sample[2 + rvk::kSampleHistWords + b / 4] >> (16 * (b % 4))) & 0xFFFF);
    ctx->last_sample_stride = sp.stride;
    ctx->last_sample_rows = in.n;
    uint64_t sampled = 0;  // rows the blocks covered (the last ones may be cut by the table's end)
    for (uint32_t b = 0; b < kBlocks; ++b) {
        const uint64_t first = static_cast<uint64_t>(b) * sp.stride;
        sampled += first >= in.n ? 0 : std::min<uint64_t>(kBlockRows, in.n - first);
    }
    return sampled ? static_cast<double>(ctx->h_sample[0]) / static_cast<double>(sampled) : -1.0;
}

// the last sample's histogram and profile travel with the predicate's memory
static void keep_sample(rv_ctx *ctx, rv_ctx::SeenPredicate *q) {
    std::copy(ctx->last_sample_hist, ctx->last_sample_hist + 16, q->hist);
    q->have_hist = true;
    std::copy(ctx->last_sample_profile, ctx->last_sample_profile + 1024, q->profile);
    q->profile_stride = ctx->last_sample_stride;
    q->profile_rows = ctx->last_sample_rows;
}

// rows the outputs of a pass over n rows are sized for (option "out_sizing")
// `expected`: the selectivity this predicate is known to have over these buffers (its last pass, or the strided sample of a first
// call); < 0: unknown
uint64_t output_capacity(rv_ctx *ctx, uint64_t n, double expected) {
    if (ctx->opt_out_sizing == 1 && ctx->last_selectivity >= 0.0)
        return std::min<uint64_t>(n, static_cast<uint64_t>(static_cast<double>(n) * (ctx->last_selectivity * 1.5 + 0.01)) + 1024);
    if (ctx->opt_out_sizing >= 2)
        return std::min<uint64_t>(n, static_cast<uint64_t>(static_cast<double>(n) * static_cast<double>(ctx->opt_out_sizing) * 1e-6) + 1024);
    // Default: big tables (the ones whose first call is sampled: 2^25 rows, 256 MiB per output column) get outputs for what the predicate
    // is known to keep x 1.2 + 2 % of the rows (the sample's 1024 blocks of a table whose survivors come in runs of 1e5 rows are off by
    // 1 % of the rows, one sigma) -- BASELINE configs[1] then holds 8 + 1.3 GB instead of 8 + 8, configs[3] at G = 1 fits one GPU without
    // an option.  A pass that keeps more counts exactly and is re-run once with outputs of that size (fused_finish).  Small tables and
    // predicates nobody has seen keep outputs for every row: nothing to save, nothing to re-run.
    if (ctx->opt_out_sizing == 0 && expected >= 0.0 && n >= rvt::kOutSizingFromRows)
        return std::min<uint64_t>(n, static_cast<uint64_t>(static_cast<double>(n) * (expected * rvt::kOutSizingFactor + rvt::kOutSizingSlack)) + 4096);
    return n;
}

// Signature of a predicate: the key of the selectivity a launch is sized from (FNV-1a over what decides a row's fate).
uint64_t predicate_signature(const rv_dcolumn *const *cols, uint32_t ncols, const rv_term *terms, uint32_t nterms, rv_null_policy policy, const ExprInfo *ex) {
    uint64_t signature = 0xcbf29ce484222325ull;
    auto mix = [&](uint64_t v) {
        for (int b = 0; b < 8; ++b) signature = (signature ^ ((v >> (8 * b)) & 0xFF)) * 0x100000001b3ull;
    };
    mix(nterms);
    mix(static_cast<uint64_t>(policy));
    for (uint32_t t = 0; t < nterms; ++t) {
        const rv_dcolumn *tc = terms[t].column < ncols ? cols[terms[t].column] : nullptr;
        mix(tc ? static_cast<uint64_t>(tc->dtype) : ~0ull);
        mix(terms[t].column);
        mix(static_cast<uint64_t>(terms[t].op));
        mix(static_cast<uint64_t>(terms[t].lit_type));
        // WHICH data the term reads: the column's buffer (a stream's windows are slices of one table: offset and length stay out of
        // it), or, for a term rewritten to `mask is true`, what the mask stood for (predicate.hip, term_identity) -- so the same
        // predicate text over another table, or another String literal over the same one, is another predicate
        if (terms[t].op == RV_IS_TRUE && terms[t].lit.i != 0) mix(static_cast<uint64_t>(terms[t].lit.i));
        else {
            mix(tc && tc->values ? (tc->values->id ? tc->values->id : reinterpret_cast<uint64_t>(tc->values->ptr)) : 0ull);
            mix(terms[t].lit_type == RV_STRING || terms[t].lit_type == RV_NULL ? 0ull : static_cast<uint64_t>(terms[t].lit.i));
        }
    }
    if (ex) {
        mix(ex->negate_result ? 3 : 2);
        for (size_t t = 0; t < ex->negate.size(); ++t) mix(static_cast<uint64_t>(ex->negate[t]) | (static_cast<uint64_t>(ex->group_end[t]) << 8));
    }
    return signature;
}

// Wave ranges whose survivors outgrew their LDS slot (runs of survivors: clustered or sorted data): re-read, range by range, by
// the generic kernel at their reserved output offsets -- one wave per listed range, nothing shared (fused_kernel.hpp).
static void launch_redo(rv_ctx *ctx, FusedLaunch &L) {
    const int rr = L.nvals ? rvk::redo_rows_per_lane(L.nvals, L.range_rows) : 4;
    const rvk::RedoFn redo = rvk::redo_kernel(L.nvals, rr);
    require(redo != nullptr && L.nranges > 0, RV_ERR_INTERNAL, "no redo kernel variant");
    const size_t redo_lds = rvk::kLdsHeader + static_cast<size_t>(rvk::kRedoWaves) * (std::max(L.nvals, 1) + rvk::kMaxBitStreams) * 64 * rr;
    const uint64_t want = (L.nranges + rvk::kRedoWaves - 1) / rvk::kRedoWaves;
    const uint32_t rgrid = static_cast<uint32_t>(std::min<uint64_t>(want, static_cast<uint64_t>(ctx->props.multiProcessorCount) * 8));
    hipLaunchKernelGGL(redo, dim3(rgrid), dim3(rvk::kRedoWaves * 64), redo_lds, ctx->stream, L.p, L.range_rows, L.nranges);
    RV_HIP(hipGetLastError());
}

// One single-pass launch: predicate over `cols`, compaction of the columns in proj; queued on the context's
// stream, not waited for.  out[] / sel_out receive the output handles at once (their length is set by
// fused_finish).
void fused_begin(rv_ctx *ctx, const rv_dcolumn *const *cols, uint32_t ncols, const rv_term *terms,
                 uint32_t nterms, rv_null_policy policy, const uint32_t *proj, uint32_t nproj,
                 rv_dcolumn **out, rv_dcolumn **sel_out, FusedLaunch &L, const ExprInfo *ex, BatchReq *req,
                 RangeOffsets *ranges) {
    require(nterms >= 1 && nterms <= static_cast<uint32_t>(rvk::kMaxTerms), RV_ERR_UNSUPPORTED,
            fmt("predicate needs 1..%d terms, got %u", rvk::kMaxTerms, nterms));
    const uint64_t n = ncols ? cols[0]->length : 0;
    const uint64_t signature = predicate_signature(cols, ncols, terms, nterms, policy, ex);
    L.signature = signature;
    double seen = L.place ? L.place_selectivity : ctx->seen_selectivity(signature);  // (a segment: its own density, out of the profile)
    bool sampled_now = false;  // `seen` comes from a sample taken by this call: its histogram is in ctx->last_sample_hist

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
    const int npred = nvals;  // value slots [0, npred): the 8-byte columns the terms read (the direct kernel loads them first)

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

    // A predicate this context has not run over this data: its selectivity from a strided sample (agg_kernel.hpp,
    // sample_count_kernel; ~10 us of device time, one launch, the host spins on a pinned word), so that the FIRST launch is already
    // sized for it -- the reference's operators have no warm-up call (stream.rs:136-158), and a one-shot collect() is always
    // the first call.  Not for small tables: below 2^25 rows a pass is a few tens of microseconds, and a mis-sized one costs less
    // than the sample.
    if (seen < 0.0 && !L.place && ctx->opt_sample >= 0 && n >= (ctx->opt_sample > 0 ? static_cast<uint64_t>(ctx->opt_sample) : rvt::kSampleFromRows)) {
        const double s = sample_selectivity(ctx, p.in, nvals);
        if (s >= 0.0) {
            seen = s;
            sampled_now = true;
        }
    }
    if (L.sample_only) {  // expected_selectivity(): the caller wanted what this launch would have been sized by, nothing more
        L.sampled = seen;
        if (seen >= 0.0 && ctx->seen_selectivity(signature) < 0.0) {  // (the pass itself will not sample again)
            rv_ctx::SeenPredicate *q = ctx->remember_selectivity(signature, seen);
            if (sampled_now) keep_sample(ctx, q);
        }
        return;
    }

    if (sampled_now) {  // the sample's histogram travels with the predicate's memory (the pass will store its true selectivity)
        keep_sample(ctx, ctx->remember_selectivity(signature, seen));
    }

    // Output capacity (output_capacity above).  Option "out_sizing": 0 = sized from the predicate's known selectivity for big tables,
    // for every row otherwise; -1 = always for every row (no second pass, 2x the input in HBM); 1 = the context's last observed
    // selectivity x 1.5 + 1 % (a stream of similar batches), k >= 2 = a caller-given bound of k rows per million.  A launch that
    // overflows its outputs still counts exactly; fused_finish then re-runs it with buffers of the exact size
    // (record_batch.rs:131-178 never over-allocates either: the builders grow).
    // (a stretch appends at row place_base of the caller's buffers: pointers rounded down to a 128-byte line, the rest enters the
    // chain as the output row of its first survivor -- FusedParams::out_bias -- and every buffer and capacity below counts from there)
    const uint32_t bias = L.place ? static_cast<uint32_t>(L.place_base & 15) : 0u;
    const uint64_t cap_out = L.place ? (L.place_capacity > L.place_base ? L.place_capacity - L.place_base : 0) + bias : output_capacity(ctx, n, seen);
    p.out_capacity = cap_out;
    p.out_bias = bias;
    L.out_bias = bias;
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
            if (L.place) {  // the caller's buffer, from row place_base on
                o->values = (*L.place)[j];
                p.out_values[slot] = static_cast<uint64_t *>(o->values->ptr) + (L.place_base - bias);
            } else {
                o->values = pool_alloc(ctx, std::max<size_t>(elem_bytes(src->dtype, cap_out), 8));
                p.out_values[slot] = static_cast<uint64_t *>(o->values->ptr);
            }
            stage_row_bytes += 8;
            if (src->validity && !never_null[c]) {
                o->validity = pool_alloc(ctx, zeroed_bitmap_bytes(cap_out));
                RV_HIP(hipMemsetAsync(o->validity->ptr, 0, zeroed_bitmap_bytes(cap_out), ctx->stream));
                p.out_validity[slot] = static_cast<uint64_t *>(o->validity->ptr);
                stage_row_bytes += 1;
            }
        } else if (src->dtype == RV_BOOLEAN) {
            require(nxs + (src->validity ? 2 : 1) <= rvk::kMaxBitStreams, RV_ERR_UNSUPPORTED,
                    "too many Boolean columns for one pass");
            const size_t wb = zeroed_bitmap_bytes(cap_out);
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
    // Geometry: the instantiation, then the LDS slots (rows a wave can stage per tile).  A wave with more survivors than its
    // slot holds leaves its range to the redo kernel, which re-reads it -- so a selectivity the default geometry's slots would
    // not hold (the context's last pass WITH THIS PREDICATE says so) walks down:
    //   1. the 16-wave instantiations with fewer rows per lane (their three-stage slots hold a larger share of a wave's rows);
    //   2. the same list sized for a dense selection (one workgroup per CU, two stages: a two-stage launch pays for the exposed
    //      look-back, 0.4-0.7 ms per 1e9 rows, so it comes second);
    //   3. the instantiation with the fewest waves (fused_roomy.hip), whose slots hold EVERY row of a wave.
    // Decided from the selectivity, which does not depend on the geometry, so the choice does not flip from call to call (a rule
    // on the share of redone tiles did: the dense geometry redoes none).  x > lit -> [x, y, fn], 5e8 rows, ms per call at
    // 10 / 20 / 30 / 50 / 90 %: default geometry + redo kernel 2.4 / 8.1 / 8.7 / 9.8 / 12.4; walked down 2.4 / 3.7 / 3.8 / 6.5 / 6.9
    // (tools/roomy_ab.py; one column: tools/dense_one.py).
    const size_t stage_row_bytes_in = stage_row_bytes;
    const rvk::FusedEntry *chosen = nullptr;
    uint64_t tile_rows = 0;
    uint32_t cap = 0;
    size_t stages = 3, lds = 0;
    auto counts_here = [&](int rows_per_lane) { return req && req->counts && req->chunk_rows % (64u * static_cast<uint64_t>(rows_per_lane)) == 0; };
    int min_r = 0, below_r = 1 << 30;  // pick_fused's `roomy` level (0 default, 1 walk down the 16-wave geometries, 2 fewest waves)
    bool dense_sizing = false;
    // Most rows survive and the outputs are plain value columns: the direct kernel (direct_kernel.hpp), which keeps a tile's rows
    // in registers until its output offset is known
    const rvk::DirectEntry *direct = nullptr;
    bool plain = nvals >= 1 && nxs == 0 && ctx->opt_rows_per_lane <= 0 &&
                 ctx->opt_cap_rows == 0 && (ctx->opt_debug & ~int64_t(4)) == 0 && (p.in.strict_values >> npred) == 0;
    bool out_validity = false;  // a projected column keeps nulls among the survivors: the FF_OUTVALID instantiations
    for (int s = 0; s < nvals; ++s) out_validity = out_validity || p.out_validity[s] != nullptr;
    for (int s = npred; s < nvals; ++s) plain = plain && p.out_values[s] != nullptr;
    // the direct instantiation for this launch's columns and features, or none
    auto direct_candidate = [&]() -> const rvk::DirectEntry * {
        const rvk::DirectEntry *found = nullptr;
        int dflags = nbools ? (rvk::FF_VALIDITY | rvk::FF_BOOL) : 0;
        if (ctx->opt_stamp) dflags |= rvk::FF_STAMP;  // diagnostic instantiations (phase cycle sums), a few geometries only
        for (int s = 0; s < npred; ++s)
            if (p.in.cols[s].validity) dflags |= rvk::FF_VALIDITY;
        if (out_validity) dflags |= rvk::FF_VALIDITY | rvk::FF_OUTVALID;
        // the first listed instantiation that covers the inputs' features (listed leanest first) -- among those whose wave
        // ranges serve the caller's side outputs, when it asks for any: wave offsets need a range that tiles 4096 rows, per-batch
        // counts a batch that is a whole number of ranges
        const rvk::DirectEntry *fallback = nullptr;
        bool tall_tried = false;
        int nv_out = 0;  // columns with an output bitmap: a validity byte per row each in the LDS slot
        for (int s = 0; s < nvals; ++s) nv_out += p.out_validity[s] != nullptr;
        auto direct_lds = [&](const rvk::DirectEntry &g) { return static_cast<size_t>(g.waves) * 64 * g.r * (8 * static_cast<size_t>(nvals) + nv_out); };
        constexpr size_t kDirectLdsBudget = (160 * 1024) / 2 - 512;
        for (int t = 0; t < 2 && !found; ++t) {
            size_t cnt = 0;
            const rvk::DirectEntry *tab = t ? rvk::direct_entries_b(&cnt) : rvk::direct_entries_a(&cnt);
            for (size_t i = 0; i < cnt && !found; ++i) {
                const rvk::DirectEntry &g = tab[i];
                if (g.np != npred || g.nq != nvals - npred || (g.flags & dflags) != dflags || ((g.flags ^ dflags) & rvk::FF_STAMP) != 0) continue;
                if (ctx->opt_direct_r > 0 || ctx->opt_direct_waves > 0) {  // diagnostic: a named geometry or none
                    if ((ctx->opt_direct_r <= 0 || g.r == ctx->opt_direct_r) && (ctx->opt_direct_waves <= 0 || g.waves == ctx->opt_direct_waves)) found = &g;
                    continue;
                }
                if (g.waves != 8 || direct_lds(g) > kDirectLdsBudget) continue;  // two workgroups per CU have to fit
                const bool serves = (!ranges || 4096u % (64u * static_cast<uint32_t>(g.r)) == 0) && (!(req && req->counts) || counts_here(g.r));
                // one loaded column while fewer than kDirectTallBelow survive (a table of runs at 30-50 %, a selection just past
                // kDirectFromOneColumn): the 16-row geometry -- what the launch keeps in registers and LDS behind a tile's aggregate
                // covers the chain's latency only at the rate that storage / latency gives, and below ~70 % that, not HBM, is the bound
                if (serves && npred == 1 && nvals == 1 && !out_validity && seen >= 0.0 && seen < rvt::kDirectTallBelow && g.r != 16 && !tall_tried) {
                    tall_tried = true;
                    for (size_t k = i + 1; k < cnt; ++k)
                        if (tab[k].np == 1 && tab[k].nq == 0 && tab[k].r == 16 && tab[k].waves == 8 && tab[k].flags == g.flags && direct_lds(tab[k]) <= kDirectLdsBudget &&
                            (!(req && req->counts) || counts_here(16))) {
                            found = &tab[k];
                            break;
                        }
                    if (found) break;
                }
                if (serves) found = &g;
                else if (!fallback) fallback = &g;
            }
        }
        if (!found && ctx->opt_direct_r <= 0 && ctx->opt_direct_waves <= 0) found = fallback;
        return found;
    };
    {
        int projected = 0;
        for (int s = 0; s < nvals; ++s) projected += p.out_values[s] != nullptr;
        // measured crossovers against the staged geometries (tools/dense_sweep.py, profiles/r04_dense_sweep.txt): one loaded column
        // from 55 %; one column projected of several loaded from 60 % (the staged pass holds every survivor in its slots there);
        // two projected columns from 22 %, three or four from 15 % (their staged rows crowd the LDS slots early)
        // (columns that keep nulls carry a validity byte per row through the LDS slot: later, tools/dense_nullable.py -- two projected
        // columns from 35 %, three from 22 %)
        double dense_from = nvals == 1 ? rvt::kDirectFromOneColumn
                                       : (projected <= 1 ? rvt::kDirectFromOneProjectedOfSeveral : (projected == 2 ? rvt::kDirectFromTwoProjected : rvt::kDirectFromThreeProjected));
        if (out_validity && nvals > 1) dense_from = std::max(dense_from, projected <= 2 ? rvt::kDirectFromTwoProjectedNullable : rvt::kDirectFromThreeProjectedNullable);
        const bool dense = seen >= dense_from;
        if (plain && (ctx->opt_direct > 0 || (ctx->opt_direct == 0 && dense))) direct = direct_candidate();
    }
    auto size_direct = [&] {
        // per-batch counts out of the pass need a batch to be a whole number of the geometry's wave ranges: else the caller counts
        // the selection bitmap, which the kernel writes on the way
        if (sel_deferred && !sel && !counts_here(direct->r)) make_selection();
        tile_rows = static_cast<uint64_t>(direct->waves) * 64 * direct->r;
        const uint64_t ntiles64 = (n + tile_rows - 1) / tile_rows;
        require(ntiles64 < (1ull << 31) - 1, RV_ERR_UNSUPPORTED, "batch too large for one launch");
        p.ntiles = static_cast<uint32_t>(ntiles64);
        stages = 2;
        // one slot per wave: its rows of every loaded column (+ a validity byte per row and column when nulls can survive)
        int nv_out = 0;
        for (int s = 0; s < nvals; ++s) nv_out += p.out_validity[s] != nullptr;
        lds = static_cast<size_t>(direct->waves) * 64 * direct->r * (8 * static_cast<size_t>(nvals) + nv_out);
    };
    if (direct) size_direct();
    // Survivors that come in RUNS (a table sorted or clustered on the predicate's column: ids, timestamps): a wave of the staged
    // pass whose 64 R rows hold more of them than its LDS slot leaves its range to the redo kernel (fused_kernel.hpp,
    // fused_redo_waves) -- cheap while few ranges do, but past a point the direct kernel, whose rows wait in registers whatever
    // survives, is ahead.  The share of ranges to expect: what the last staged pass of this predicate over these buffers left
    // to the redo kernel, or -- on a first call -- the share of the sample's 1024-row blocks denser than the slot.
    auto redo_estimate = [&](uint32_t cap_rows, uint32_t rows_per_wave) -> double {
        const rv_ctx::SeenPredicate *q = L.place ? nullptr : ctx->seen_entry(signature);
        if (cap_rows >= rows_per_wave) return 0.0;
        const double ratio = static_cast<double>(cap_rows) / rows_per_wave, width = 1.0 / rvk::kSampleBuckets;
        // measured by a staged pass whose slots held at least this share of a wave's rows (roomier slots than it had: unknown -- independent
        // rows stop outgrowing them, runs do not)
        if (q && q->redo_fraction >= 0.0 && std::fabs(q->redo_at - seen) < rvt::kRedoMemorySelectivityBand && ratio <= q->redo_ratio + rvt::kRedoMemorySlotBand) return q->redo_fraction;
        const float *hist = sampled_now ? ctx->last_sample_hist : ((q && q->have_hist) ? q->hist : nullptr);
        if (!hist) return 0.0;
        double f = 0.0;
        for (int b = 0; b < rvk::kSampleBuckets; ++b) {
            const double lo = b * width, hi = lo + width;
            if (lo >= ratio) f += hist[b];
            else if (hi > ratio) f += hist[b] * (hi - ratio) / width;
        }
        return f;
    };
    double redo_expected = 0.0;
    while (!direct) {
        chosen = &pick_fused(ctx, nvals, vec, need, prefer, min_r, below_r);
        if (min_r == 1 && (chosen->waves != 16 || chosen->r >= below_r)) {  // no 16-wave geometry below that many rows per lane left
            if (!dense_sizing) {  // the walk again, sized for a dense selection
                dense_sizing = true;
                below_r = 1 << 30;
            } else {
                min_r = 2;
            }
            continue;
        }
        // per-batch counts out of the pass: a batch must be a whole number of the geometry's wave ranges
        if (sel_deferred && !sel && !counts_here(chosen->r)) {  // the caller will count the selection bitmap instead: materialise it after all
            make_selection();
            need |= rvk::FF_SEL;
            chosen = &pick_fused(ctx, nvals, vec, need, prefer, min_r, below_r);
        }
        const rvk::FusedEntry &e = *chosen;
        stage_row_bytes = stage_row_bytes_in;
        if (e.flags & rvk::FF_PROJALL)  // the kernel stages a validity byte for every column when any has a bitmap
            stage_row_bytes = static_cast<size_t>(nvals) * (((e.flags & rvk::FF_VALIDITY) && !(e.flags & rvk::FF_NONULL)) ? 9 : 8) + static_cast<size_t>(nxs);
        tile_rows = static_cast<uint64_t>(e.waves) * 64 * e.r;
        const uint64_t ntiles64 = (n + tile_rows - 1) / tile_rows;
        require(ntiles64 < (1ull << 31), RV_ERR_UNSUPPORTED, "batch too large for one launch");
        p.ntiles = static_cast<uint32_t>(ntiles64);

        // LDS: every wave owns two slots (double buffered for the deferred look-back) of cap rows.
        // One 1024-thread workgroup per CU may use most of the 160 KiB; 512-thread variants keep to
        // half so that two workgroups fit.
        const uint32_t rows_per_wave = 64u * static_cast<uint32_t>(e.r);
        // 16 waves x 4 per SIMD is one workgroup per CU (128 VGPRs each): it may use most of the LDS; smaller workgroups run
        // two per CU -- unless sized for a dense selection: one workgroup per CU whatever its size, two stages
        const bool roomy = dense_sizing || ctx->opt_roomy != 0;
        const size_t budget = (roomy || e.waves >= 16) ? 144 * 1024 : 72 * 1024;
        // Three stages (write-out two iterations after the aggregate went out, so the scanner's prefix is
        // there when it is needed) when a slot still holds 3/16 of a wave's rows; two otherwise.
        auto cap_for = [&](size_t st) -> uint32_t {
            if (!stage_row_bytes) return rows_per_wave;
            return static_cast<uint32_t>(std::min<uint64_t>(rows_per_wave, (budget / (st * e.waves * stage_row_bytes)) & ~size_t(63)));
        };
        stages = 3;
        if (ctx->opt_depth == 1 || (ctx->opt_depth == 0 && (roomy || cap_for(3) * 16 < rows_per_wave * 3))) stages = 2;
        cap = cap_for(stages);
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
        lds = lds_for(stages, cap);
        // expected survivors of a wave (+ 10 % and three standard deviations of a binomial) against the slot
        const double expect = seen * rows_per_wave;
        const bool crowded = min_r < 2 && ctx->opt_rows_per_lane <= 0 && nvals >= 1 && stage_row_bytes && cap < rows_per_wave &&
                             seen > 0.0 && expect * rvt::kCrowdedMargin + rvt::kCrowdedSigmas * std::sqrt(expect) > static_cast<double>(cap);
        if (!crowded) break;
        below_r = e.r;  // the next 16-wave geometry with fewer rows per lane
        min_r = 1;
    }
    if (!direct && chosen && seen >= 0.0 && stage_row_bytes) {
        redo_expected = redo_estimate(cap, 64u * static_cast<uint32_t>(chosen->r));
        // Per 1e9 rows of one column the redo kernel costs ~3.5 ms x the share of ranges it re-reads (tools/skew_sweep.py: sorted
        // 10 % 1.37 -> 1.68 ms, sorted 50 % 1.98 -> 3.95 ms); the direct kernel costs 0.8 ms more than the staged pass at 10 %
        // selectivity, 0.4 at 30 %, 0.25 at 50 % (profiles/r04_dense_sweep.txt).
        if (plain && ctx->opt_direct == 0 && ctx->opt_skew >= 0 &&
            redo_expected * rvt::kRedoMsPerShare > std::max(rvt::kDirectPenaltyFloor, rvt::kDirectPenaltyAt0 - rvt::kDirectPenaltySlope * seen)) {
            direct = direct_candidate();
            if (direct) {
                size_direct();
                redo_expected = 0.0;
            }
        }
    }
    // the launch geometry, whichever kernel was chosen
    struct Geometry {
        int ncols, r, vec, waves, flags;
        void (*fn)(const rvk::FusedParams);
    };
    const Geometry e = direct ? Geometry{nvals, direct->r, 1, direct->waves, direct->flags, direct->fn}
                              : Geometry{chosen->ncols, chosen->r, chosen->vec, chosen->waves, chosen->flags, chosen->fn};
    p.cap_rows = cap;
    p.depth = static_cast<int32_t>(stages) - 1;

    // the staged pass lists the wave ranges whose survivors outgrew their LDS slot (fused_kernel.hpp, fused_redo_waves): one
    // word per range behind the descriptors, zeroed with them; the direct kernel's rows never leave their registers: no list
    L.range_rows = 64u * static_cast<uint32_t>(e.r);
    L.nranges = direct ? 0 : static_cast<uint64_t>(p.ntiles) * e.waves;
    L.ctrl = acquire_launch_ctrl(ctx, p.ntiles, L.nranges);
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
    p.redo = L.nranges ? reinterpret_cast<unsigned long long *>(static_cast<unsigned char *>(L.ctrl.dev) + kCtrlBytes + static_cast<size_t>(p.ntiles) * 8) : nullptr;

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
        ranges->expected_selectivity = seen;
        if (4096u % ranges->range_rows == 0) {
            ranges->offsets = pool_alloc(ctx, static_cast<size_t>(p.ntiles) * e.waves * 8 + 16);
            p.wave_offsets = static_cast<uint64_t *>(ranges->offsets->ptr);
        }
    }
    if (counts_here(e.r)) {
        if (req->chunk_rows == 64u * static_cast<uint64_t>(e.r)) {
            // a batch IS a wave range (the reference's 1024-row batches at 16 rows per lane): the pass writes every batch's survivor
            // count where the caller reads it -- 128 bytes per tile over PCIe, spread over the whole pass; no scratch, no second kernel
            // (which took 39 us per 2^28-row window, serialised behind the pass whatever stream it ran on: profiles/r05_seam_*)
            p.batch_counts = req->counts;
            p.nbatch_counts = req->nb;
            req->counted = true;
            ctx->batch_counts_in_pass += 1;
        } else {
            L.wave_counts = pool_alloc(ctx, static_cast<size_t>(p.ntiles) * e.waves * 4 + 16);
            p.wave_counts = static_cast<uint32_t *>(L.wave_counts->ptr);
        }
    }
    L.direct_stamp = direct && (direct->flags & rvk::FF_STAMP);
    L.fn = e.fn;
    L.grid = grid;
    L.block = static_cast<uint32_t>(e.waves * 64);
    L.lds = lds;
    L.timed = ctx->opt_profile != 0;
    ctx->last_kernel = direct ? fmt("fused_direct_compact<%d,%d,%d,%d,%d>", direct->np, direct->nq, e.r, e.waves, e.flags) : fmt("fused_filter_compact<%d,%d,%d,%d,%d>", e.ncols, e.r, e.vec, e.waves, e.flags);
    // a free control block for this pass to zero while it runs (big launches: the zeroing hides in the pass), for a later launch
    int zero_at = -1;
    if (n >= rvt::kRangesFromRows && e.waves >= 2) {
        zero_at = block_to_zero(ctx, L.ctrl);
        if (zero_at >= 0) {
            const rv_ctx::LaunchCtrl &z = ctx->ctrl_free[static_cast<size_t>(zero_at)];
            p.zero_ptr = static_cast<rvk::rv_u32x4 *>(z.dev);
            p.zero_n16 = (std::min(z.dirty, z.bytes) + 15) / 16;
        }
    }
    if (L.timed) {  // the launch's own pair of events: launches of a stream's windows overlap (rv_filter_project_chunked_begin)
        if (!L.ctrl.tk0) {
            RV_HIP(hipEventCreate(&L.ctrl.tk0));
            RV_HIP(hipEventCreate(&L.ctrl.tk1));
        }
        RV_HIP(hipEventRecord(L.ctrl.tk0, ctx->stream));
    }
    hipLaunchKernelGGL(e.fn, dim3(grid), dim3(e.waves * 64), lds, ctx->stream, p);
    RV_HIP(hipGetLastError());
    if (zero_at >= 0) {  // queued: whatever follows on this stream finds the block zero
        rv_ctx::LaunchCtrl &z = ctx->ctrl_free[static_cast<size_t>(zero_at)];
        z.clean = std::min(z.dirty, z.bytes);
    }
    L.p.zero_ptr = nullptr;  // (a re-run after an output overflow zeroes nothing: the block may be in use by then)
    L.p.zero_n16 = 0;
    L.nvals = nvals;
    L.redo_queued = false;
    if (L.nranges && (redo_expected > 0.0 || L.place_edge) && (stage_row_bytes || nxs)) {
        // ranges are expected to outgrow their slots (they did the last time): the redo kernel follows the pass on the stream at
        // once instead of waiting for the host to read the count
        launch_redo(ctx, L);
        L.redo_queued = true;
    }
    if (L.timed) RV_HIP(hipEventRecord(L.ctrl.tk1, ctx->stream));
    if (p.wave_counts) {
        // wave counts -> the caller's per-batch array (pinned host memory, written by the device: no read-back to queue); the scratch is
        // kept until the launch is finished
        const uint64_t per_batch = req->chunk_rows / (64u * static_cast<uint64_t>(e.r)), nwaves = static_cast<uint64_t>(p.ntiles) * e.waves;
        const uint64_t threads = per_batch < 32 ? req->nb : (per_batch < 4096 ? req->nb * 64 : req->nb * 256);
        const dim3 cgrid(static_cast<uint32_t>(std::max<uint64_t>(1, std::min<uint64_t>((threads + 255) / 256, static_cast<uint64_t>(ctx->props.multiProcessorCount) * 8))));
        hipLaunchKernelGGL(rvk::batch_counts_from_waves, cgrid, dim3(256), 0, ctx->stream, static_cast<const uint32_t *>(p.wave_counts), nwaves, per_batch, req->nb, req->counts);
        RV_HIP(hipGetLastError());
        req->counted = true;
        ctx->batch_counts_in_pass += 1;
        p.wave_counts = nullptr;  // a re-run after an output overflow does not count again (the first pass's counts are exact)
    }
    L.p.batch_counts = nullptr;  // (likewise)
    RV_HIP(hipMemcpyAsync(L.ctrl.host, L.ctrl.dev, kCtrlBytes, hipMemcpyDeviceToHost, ctx->stream));
    RV_HIP(hipEventRecord(L.ctrl.ev, ctx->stream));
    L.launched = true;
    L.need = need;
    L.nvals = nvals;
    L.nxs = nxs;
    L.stage_row_bytes = stage_row_bytes;
    L.tile_rows = tile_rows;
}

// Waits for the launch, runs the redo kernel when wave ranges outgrew their slots (unless it was queued behind the pass), fixes the output lengths / null counts.
// What a pass over this predicate would be sized by: the selectivity it had the last time it ran over these buffers, or -- a predicate
// the context has not seen, over a big table -- the strided sample its first launch would take (taken here instead, once); < 0: unknown.
// For decisions that precede the launch (which columns the pass should carry at all: query.hip, filter_by_groups).
double expected_selectivity(rv_ctx *ctx, const rv_dcolumn *const *cols, uint32_t ncols, const rv_term *terms, uint32_t nterms, rv_null_policy policy,
                            const ExprInfo *ex) {
    const double known = ctx->seen_selectivity(predicate_signature(cols, ncols, terms, nterms, policy, ex));
    if (known >= 0.0) return known;
    FusedLaunch L;
    L.sample_only = true;
    fused_begin(ctx, cols, ncols, terms, nterms, policy, nullptr, 0, nullptr, nullptr, L, ex, nullptr, nullptr);
    return L.sampled;
}

uint64_t fused_finish(rv_ctx *ctx, FusedLaunch &L) {
    if (!L.launched) return 0;
    rvk::FusedParams &p = L.p;
    std::vector<OutCol> &outs = L.outs;
    const int need = L.need, nxs = L.nxs;
    const size_t stage_row_bytes = L.stage_row_bytes;
    struct Release {
        rv_ctx *ctx;
        FusedLaunch &L;
        ~Release() {
            L.wave_counts.reset();
            release_launch_ctrl(ctx, L.ctrl);
            L.launched = false;
        }
    } release{ctx, L};
    RV_HIP(hipEventSynchronize(L.ctrl.ev));
    L.wave_counts.reset();
    const Ctrl *h = static_cast<const Ctrl *>(L.ctrl.host);
    if (L.timed) {
        float ms = 0.f;
        RV_HIP(hipEventElapsedTime(&ms, L.ctrl.tk0, L.ctrl.tk1));
        ctx->kernel_ms += ms;
        ctx->kernel_launches += 1;
    }
    require(h->err == 0, RV_ERR_DEVICE, "fused kernel: look-back spin limit reached (device fault or lost workgroup)");
    ctx->last_selectivity = L.n ? static_cast<double>(h->out_count) / static_cast<double>(L.n) : 0.0;
    ctx->last_rows_out = h->out_count, ctx->last_rows_in = L.n;
    if (!L.place) ctx->remember_selectivity(L.signature, ctx->last_selectivity);
    bool rerun = false;
    if (L.place && (h->overflow || h->out_count > p.out_capacity)) {  // the shared outputs are the caller's: it falls back to one pass
        L.overflowed = true;
        throw SegmentOverflow{};
    }
    if (h->overflow || h->out_count > p.out_capacity) {
        rerun = true;
        // speculative output sizing guessed too low: the count is exact, so give every output exactly that many rows
        // and run the pass once more (same kernel, same geometry, fresh descriptors)
        const uint64_t exact = h->out_count;
        p.out_capacity = exact;
        for (auto &o : outs) {
            if (o.value_slot >= 0) {
                o.col->values = pool_alloc(ctx, std::max<size_t>(exact * 8, 8));
                p.out_values[o.value_slot] = static_cast<uint64_t *>(o.col->values->ptr);
                if (o.col->validity) {
                    const size_t wb = zeroed_bitmap_bytes(exact);
                    o.col->validity = pool_alloc(ctx, wb);
                    RV_HIP(hipMemsetAsync(o.col->validity->ptr, 0, wb, ctx->stream));
                    p.out_validity[o.value_slot] = static_cast<uint64_t *>(o.col->validity->ptr);
                }
            }
            if (o.xs_values >= 0) {
                const size_t wb = zeroed_bitmap_bytes(exact);
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
        const size_t zeroed = kCtrlBytes + (static_cast<size_t>(p.ntiles) + L.nranges) * 8;
        RV_HIP(hipMemsetAsync(L.ctrl.dev, 0, (zeroed + 15) & ~size_t(15), ctx->stream));
        hipLaunchKernelGGL(L.fn, dim3(L.grid), dim3(L.block), L.lds, ctx->stream, p);
        RV_HIP(hipGetLastError());
        RV_HIP(hipMemcpyAsync(L.ctrl.host, L.ctrl.dev, kCtrlBytes, hipMemcpyDeviceToHost, ctx->stream));
        RV_HIP(hipStreamSynchronize(ctx->stream));
        require(h->err == 0 && h->overflow == 0 && h->out_count == exact, RV_ERR_INTERNAL, "re-run after an output overflow disagrees with the first pass");
        ctx->overflow_reruns += 1;
    }
    ctx->last_redo_fraction = L.nranges ? static_cast<double>(h->redo_count) / static_cast<double>(L.nranges) : 0.0;
    if (L.nranges && !L.place) {  // a staged pass ran: what it left to the redo kernel sizes the predicate's next launch (fused_begin, redo_estimate)
        if (rv_ctx::SeenPredicate *q = ctx->seen_entry(L.signature)) {
            q->redo_fraction = ctx->last_redo_fraction, q->redo_at = ctx->last_selectivity;
            q->redo_ratio = L.range_rows ? static_cast<double>(p.cap_rows) / L.range_rows : 1.0;
        }
    }
    if (h->redo_count > 0 && (stage_row_bytes || nxs) && !(L.redo_queued && !rerun)) {
        if (L.timed) RV_HIP(hipEventRecord(ctx->evk0, ctx->stream));
        launch_redo(ctx, L);
        if (L.timed) RV_HIP(hipEventRecord(ctx->evk1, ctx->stream));
        RV_HIP(hipMemcpyAsync(L.ctrl.host, L.ctrl.dev, kCtrlBytes, hipMemcpyDeviceToHost, ctx->stream));
        RV_HIP(hipStreamSynchronize(ctx->stream));
        if (L.timed) {  // the redo kernel is part of the pass's device time
            float ms = 0.f;
            RV_HIP(hipEventElapsedTime(&ms, ctx->evk0, ctx->evk1));
            ctx->kernel_ms += ms;
        }
        require(h->overflow == 0, RV_ERR_INTERNAL, "redo kernel: outputs too small after the pass fitted them");
    }
    const uint64_t rows = h->out_count;
    if (ctx->opt_debug & 4)
        fprintf(stderr, "[scan] tiles %llu | scanner polls %llu, tiles scanned %llu, empty polls %llu | fallback look-backs %llu\n",
                static_cast<unsigned long long>(p.ntiles), h->stamps[28], h->stamps[29], h->stamps[30], h->stamps[31]);
    if ((ctx->opt_debug & 4) && L.nranges == 0)
        fprintf(stderr, "[scan] direct kernel: tiles whose predecessor's prefix was not there at the top of the iteration: looked again once %llu, twice %llu, three times %llu, four %llu\n",
                h->stamps[16], h->stamps[17], h->stamps[18], h->stamps[19]);
    if (L.direct_stamp) {  // diagnostic instantiations of the direct kernel: where an ordinary wave's cycles go, per tile
        const double t = static_cast<double>(p.ntiles);
        const unsigned long long *q = h->stamps;
        fprintf(stderr, "[stamp direct] cycles/tile (s_memtime, wave 1): load wait %.0f | predicate+count %.0f | barrier 1 %.0f | publish..offset (wave 0: %.0f) %.0f | barrier 2 %.0f | stores %.0f | barrier 3 %.0f | moves+load issue %.0f\n",
                q[0] / t, q[1] / t, q[2] / t, q[8] / t, q[3] / t, q[4] / t, q[5] / t, q[6] / t, q[7] / t);
    }
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
            // builder drops it (primitive.rs:179-185) -- a stretch keeps its own: the table's bitmap is put together from all of them
            if (o.col->null_count == 0 && !L.place) o.col->validity.reset();
        }
    }
    return rows;
}

void abandon_launch(rv_ctx *ctx, FusedLaunch &L) {
    if (!L.launched) return;
    (void)hipStreamSynchronize(ctx->stream);
    L.wave_counts.reset();
    release_launch_ctrl(ctx, L.ctrl);
    L.launched = false;
}

// ---- a table filtered stretch by stretch ------------------------------------------------------------------------------------------
// The reference filters whatever order its source has (plan.rs:112-147) and `col > lit` columns are ids and timestamps: SORTED.  The
// survivors of such a table sit in one or two long stretches -- which the strided sample shows as a profile of its 1024 blocks in table
// order (agg_kernel.hpp).  One launch sized for the table's global selectivity is wrong everywhere there: the staged pass's slots
// overflow where every row survives (the redo kernel re-reads those ranges), the direct kernel crawls at its chain's pace where none
// does.  So the table is cut at the profile's edges and every stretch gets the kernel ITS density asks for -- the staged pass where
// (nearly) nothing survives, the direct kernel where (nearly) everything does -- all writing into one set of outputs, each stretch
// from the row where the one before it ended (the host reads that count between two launches: ~30 us each).  No range is read twice.
namespace {
struct Stretch {
    uint64_t row0, row1;
    double density;
};
// at most kMost stretches, each at least kLeast blocks long and clearly sparse or clearly dense (thresholds.hpp); anything else --
// survivors in many runs, independent rows -- is not a table to cut
bool plan_stretches(const rv_ctx::SeenPredicate &q, uint64_t n, std::vector<Stretch> &out) {
    constexpr int kBlocks = 1024, kLeast = rvt::kStretchLeastBlocks, kMost = rvt::kStretchesMost;
    if (!q.profile_stride || q.profile_rows != n) return false;
    int cls[kBlocks];
    for (int b = 0; b < kBlocks; ++b) cls[b] = q.profile[b] <= rvt::kStretchSparseUpTo * 1024 ? 0 : (q.profile[b] >= rvt::kStretchDenseFrom * 1024 ? 2 : 1);
    // single blocks of another class inside a stretch (a run's edge falls into the block) take their neighbours' class
    for (int b = 1; b + 1 < kBlocks; ++b)
        if (cls[b] != cls[b - 1] && cls[b - 1] == cls[b + 1]) cls[b] = cls[b - 1];
    struct Run {
        int first, count, cls;
        uint64_t survivors;
    };
    std::vector<Run> runs;
    for (int b = 0; b < kBlocks; ++b) {
        if (runs.empty() || runs.back().cls != cls[b]) runs.push_back(Run{b, 0, cls[b], 0});
        runs.back().count += 1;
        runs.back().survivors += q.profile[b];
    }
    // an in-between block or two at an edge joins the stretch before it
    std::vector<Run> merged;
    for (const Run &r : runs) {
        if (r.cls == 1 && r.count <= 2 && !merged.empty()) {
            merged.back().count += r.count;
            merged.back().survivors += r.survivors;
        } else if (!merged.empty() && merged.back().cls == r.cls) {
            merged.back().count += r.count;
            merged.back().survivors += r.survivors;
        } else {
            merged.push_back(r);
        }
    }
    if (merged.size() < 2 || merged.size() > static_cast<size_t>(kMost)) return false;
    bool sparse = false, dense = false;
    for (const Run &r : merged) {
        if (r.cls == 1 || r.count < kLeast) return false;
        sparse = sparse || r.cls == 0;
        dense = dense || r.cls == 2;
    }
    if (!sparse || !dense) return false;
    out.clear();
    uint64_t at = 0;
    for (size_t k = 0; k < merged.size(); ++k) {
        const Run &r = merged[k];
        // the edge lies somewhere between two sampled blocks: cut at the start of the next stretch's first block, on a tile boundary
        uint64_t end = k + 1 == merged.size() ? n : std::min<uint64_t>(n, (static_cast<uint64_t>(r.first + r.count) * q.profile_stride) & ~uint64_t(16383));
        if (end <= at) return false;
        out.push_back(Stretch{at, end, static_cast<double>(r.survivors) / (1024.0 * r.count)});
        at = end;
    }
    return true;
}
}  // namespace

bool run_segmented_pass(rv_ctx *ctx, const rv_dcolumn *const *cols, uint32_t ncols, const rv_term *terms, uint32_t nterms, rv_null_policy policy,
                        const uint32_t *proj, uint32_t nproj, rv_dcolumn **out, const ExprInfo *ex, uint64_t *rows_out) {
    const uint64_t n = ncols ? cols[0]->length : 0;
    if (ctx->opt_segments < 0 || n < (ctx->opt_segments > 0 ? static_cast<uint64_t>(ctx->opt_segments) : rvt::kStretchFromRows) || nproj == 0 || ctx->opt_out_sizing < 0 || ctx->opt_rows_per_lane > 0 || ctx->opt_cap_rows > 0 || ctx->opt_debug) return false;
    for (uint32_t j = 0; j < nproj; ++j)
        if (proj[j] >= ncols || !is_value_type(cols[proj[j]]->dtype)) return false;  // 8-byte values land in place; a bitmap is put together afterwards
    const uint64_t signature = predicate_signature(cols, ncols, terms, nterms, policy, ex);
    // (a predicate the context has not seen: the strided sample its first launch would take, taken here -- once, the pass will not repeat it)
    if (!ctx->seen_entry(signature)) (void)expected_selectivity(ctx, cols, ncols, terms, nterms, policy, ex);
    const rv_ctx::SeenPredicate *q = ctx->seen_entry(signature);
    std::vector<Stretch> plan;
    if (!q || !plan_stretches(*q, n, plan)) return false;
    {  // what a pass over the whole table has counted since the sample was taken outranks the profile
        double promised = 0.0;
        for (const Stretch &s : plan) promised += s.density * static_cast<double>(s.row1 - s.row0);
        const double known = ctx->seen_selectivity(signature);
        if (known >= 0.0 && std::fabs(known - promised / static_cast<double>(n)) > rvt::kStretchKnownBand) return false;
    }
    // one set of outputs for all stretches: what the profile says survives x 1.2 + 2 % of the rows
    double expect = 0.0;
    for (const Stretch &s : plan) expect += s.density * static_cast<double>(s.row1 - s.row0);
    const uint64_t capacity = std::min<uint64_t>(n, static_cast<uint64_t>(expect * rvt::kOutSizingFactor + static_cast<double>(n) * rvt::kOutSizingSlack) + 4096);
    std::vector<DevBufRef> bufs;
    for (uint32_t j = 0; j < nproj; ++j) bufs.push_back(pool_alloc(ctx, std::max<size_t>(elem_bytes(cols[proj[j]]->dtype, capacity), 8)));
    uint64_t base = 0;
    std::string kernels;
    const uint64_t launches_before = ctx->kernel_launches;
    // a column that keeps nulls: every stretch writes a bitmap of its own (bit 0 = its first survivor), concatenated at the end
    struct Bits {
        DevBufRef words;
        uint64_t first, rows;  // bit of the first survivor (the stretch's out_bias), survivors
        int64_t nulls;
    };
    std::vector<std::vector<Bits>> bitmaps(nproj);
    try {
        for (const Stretch &s : plan) {
            std::vector<std::unique_ptr<rv_dcolumn>> views;
            std::vector<const rv_dcolumn *> vc;
            for (uint32_t c = 0; c < ncols; ++c) {
                auto v = std::make_unique<rv_dcolumn>(*cols[c]);
                v->offset = cols[c]->offset + s.row0;
                v->length = s.row1 - s.row0;
                v->null_count = cols[c]->dtype == RV_NULL ? static_cast<int64_t>(v->length) : (cols[c]->validity ? -1 : 0);  // (rv_slice)
                vc.push_back(v.get());
                views.emplace_back(std::move(v));
            }
            FusedLaunch L;
            L.place = &bufs;
            L.place_base = base;
            L.place_capacity = capacity;
            L.place_selectivity = s.density;
            L.place_edge = s.density < 0.5;  // (a plan alternates sparse and dense stretches: queue the redo kernel behind the pass, ~13 us,
                                             //  instead of a second launch after the host has read the count, ~45 us)
            std::vector<rv_dcolumn *> tmp(nproj, nullptr);
            struct Drop {  // the stretch's output handles are views of the shared buffers: dropped, whatever happens
                std::vector<rv_dcolumn *> &t;
                ~Drop() {
                    for (auto *d : t) delete d;
                }
            } drop{tmp};
            fused_begin(ctx, vc.data(), ncols, terms, nterms, policy, proj, nproj, tmp.data(), nullptr, L, ex, nullptr, nullptr);
            uint64_t kept = 0;
            try {
                kept = fused_finish(ctx, L);
            } catch (...) {
                abandon_launch(ctx, L);
                throw;
            }
            for (uint32_t j = 0; j < nproj; ++j)
                if (tmp[j] && tmp[j]->validity) bitmaps[j].push_back(Bits{tmp[j]->validity, L.out_bias, kept, tmp[j]->null_count});
            base += kept;
            kernels += (kernels.empty() ? "" : " + ") + ctx->last_kernel;
        }
    } catch (const SegmentOverflow &) {
        RV_HIP(hipStreamSynchronize(ctx->stream));
        if (rv_ctx::SeenPredicate *m = ctx->seen_entry(signature)) m->profile_stride = 0;  // a profile that lied is not asked again
        ctx->segment_fallbacks += 1;
        return false;  // more survivors than the profile promised: the one-pass path, which counts exactly and re-runs itself
    }
    std::vector<std::unique_ptr<rv_dcolumn>> made;
    for (uint32_t j = 0; j < nproj; ++j) {
        auto o = std::make_unique<rv_dcolumn>();
        o->dtype = cols[proj[j]]->dtype;
        o->values = bufs[j];
        o->length = base;
        o->null_count = 0;
        for (const Bits &b : bitmaps[j]) o->null_count += b.nulls;
        if (o->null_count > 0) {  // (record_batch.rs:277-342's bitmap concatenation, over the stretches' bitmaps as Boolean columns)
            std::vector<std::unique_ptr<rv_dcolumn>> parts;
            std::vector<const rv_dcolumn *> pp;
            for (const Bits &b : bitmaps[j]) {
                auto c = std::make_unique<rv_dcolumn>();
                c->dtype = RV_BOOLEAN, c->values = b.words, c->offset = b.first, c->length = b.rows, c->null_count = 0;
                pp.push_back(c.get());
                parts.emplace_back(std::move(c));
            }
            rv_dcolumn *merged = nullptr;
            require(bitmaps[j].size() == plan.size(), RV_ERR_INTERNAL, "a stretch lost its bitmap");
            if (rv_concat(ctx, pp.data(), static_cast<uint32_t>(pp.size()), &merged) != RV_OK) throw Error(RV_ERR_INTERNAL, "stretches: the bitmaps could not be put together");
            o->validity = merged->values;
            delete merged;
        }
        made.emplace_back(std::move(o));
    }
    for (uint32_t j = 0; j < nproj; ++j) out[j] = made[j].release();
    ctx->last_selectivity = n ? static_cast<double>(base) / static_cast<double>(n) : 0.0;
    ctx->last_rows_out = base, ctx->last_rows_in = n;
    ctx->remember_selectivity(signature, ctx->last_selectivity);
    ctx->last_kernel = "stretches: " + kernels;
    if (ctx->kernel_launches > launches_before) ctx->kernel_launches = launches_before + 1;  // (the stretches' device times add up to ONE pass over the table)
    ctx->segmented_passes += 1;
    *rows_out = base;
    return true;
}

// begin + finish: the synchronous form
uint64_t run_fused_pass(rv_ctx *ctx, const rv_dcolumn *const *cols, uint32_t ncols, const rv_term *terms,
                        uint32_t nterms, rv_null_policy policy, const uint32_t *proj, uint32_t nproj,
                        rv_dcolumn **out, rv_dcolumn **sel_out, const ExprInfo *ex, BatchReq *req,
                        const AfterLaunch *after_launch, RangeOffsets *ranges) {
    FusedLaunch L;
    fused_begin(ctx, cols, ncols, terms, nterms, policy, proj, nproj, out, sel_out, L, ex, req, ranges);
    if (after_launch && *after_launch) {
        try {
            (*after_launch)(sel_out ? *sel_out : nullptr);
        } catch (...) {
            abandon_launch(ctx, L);  // the pass may still be running: drain before its buffers go
            throw;
        }
    }
    return fused_finish(ctx, L);
}

}  // namespace rvl
