// RecordBatch::take / concat on the device, selection -> indices, pinned host memory.
// One unit of the backend library behind include/rivulus_gpu.h (gfx950 only; compiled with hipcc).  Shared helpers and the
// functions the units call across each other are declared in launch.hpp (namespace rvl).
#include "launch.hpp"

using namespace rvh;
using namespace rvl;

namespace rvl {
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
}  // namespace rvl

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
