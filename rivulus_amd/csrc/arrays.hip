// Arrays (rv_upload .. rv_download), rv_eval_predicate, rv_compare*, BooleanArray logic.
// One unit of the backend library behind include/rivulus_gpu.h (gfx950 only; compiled with hipcc).  Shared helpers and the
// functions the units call across each other are declared in launch.hpp (namespace rvl).
#include "launch.hpp"

using namespace rvh;
using namespace rvl;

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
        require(spec->pattern <= RV_SYNTH_SORTED_DESC, RV_ERR_INVALID_ARG, "rv_generate: unknown pattern");
        require(spec->pattern != RV_SYNTH_CLUSTERED || spec->run_rows > 0, RV_ERR_INVALID_ARG, "rv_generate: run_rows must be > 0");
        const bool sorted = spec->pattern == RV_SYNTH_SORTED_ASC || spec->pattern == RV_SYNTH_SORTED_DESC;
        const uint64_t table_rows = spec->table_rows ? spec->table_rows : spec->first_row + spec->length;
        require(!sorted || spec->first_row + spec->length <= table_rows, RV_ERR_INVALID_ARG, "rv_generate: rows past table_rows");
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
            g.pattern = spec->pattern;
            g.run_rows = spec->run_rows;
            g.table_rows = table_rows;
            g.step = (sorted && table_rows) ? ~0ull / table_rows : 0;
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

}  // extern "C"
namespace rvl {
// ---- predicate ---------------------------------------------------------------------------------
void check_batch(const rv_dcolumn *const *cols, uint32_t ncols) {
    require(cols != nullptr || ncols == 0, RV_ERR_INVALID_ARG, "cols is NULL");
    for (uint32_t i = 0; i < ncols; ++i) {
        require(cols[i] != nullptr, RV_ERR_INVALID_ARG, fmt("column %u is NULL", i));
        // RecordBatch::try_new (record_batch.rs:31-40)
        require(cols[i]->length == cols[0]->length, RV_ERR_LENGTH_MISMATCH,
                fmt("Column %u has length %llu but expected %llu", i, static_cast<unsigned long long>(cols[i]->length),
                    static_cast<unsigned long long>(cols[0]->length)));
    }
}
}  // namespace rvl
extern "C" {

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
            if (st != RV_OK) throw Error(st, last_error());
            if (nulls == 0) o->validity.reset();  // BooleanArrayBuilder::finish (boolean.rs:282-286)
        }
        *out_bool = o.release();
    });
}

}  // extern "C"
namespace rvl {
// ---- BooleanArray logic ---------------------------------------------------------------------------
void bool_op(rv_ctx *ctx, int kind, const rv_dcolumn *a, const rv_dcolumn *b, rv_dcolumn **out) {
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
}  // namespace rvl
extern "C" {

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

}  // extern "C"
