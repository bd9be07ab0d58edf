// Filter + SUM/COUNT (rv_filter_agg), row-range shards, one-rank-per-process RCCL communicator (rv_comm_*).
// One unit of the backend library behind include/rivulus_gpu.h (gfx950 only; compiled with hipcc).  Shared helpers and the
// functions the units call across each other are declared in launch.hpp (namespace rvl).
#include "launch.hpp"
#include "rccl_loader.hpp"

using namespace rvh;
using namespace rvl;

extern "C" {

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
