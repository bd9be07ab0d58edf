// Host-resident tables: chunked upload on a second stream overlapped with the fused pass (rv_filter_project_host).
// One unit of the backend library behind include/rivulus_gpu.h (gfx950 only; compiled with hipcc).  Shared helpers and the
// functions the units call across each other are declared in launch.hpp (namespace rvl).
#include "launch.hpp"

using namespace rvh;
using namespace rvl;

namespace rvl {
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
}  // namespace rvl

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
        maybe_injected_failure(ctx);
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
                    if (st != RV_OK) throw Error(st, last_error());
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

}  // extern "C"
