// StringArray and bit-packed columns on the device: scans, gathers (take / selection form), the launches queued behind the fused pass, concat, String compares.
// One unit of the backend library behind include/rivulus_gpu.h (gfx950 only; compiled with hipcc).  Shared helpers and the
// functions the units call across each other are declared in launch.hpp (namespace rvl).
#include "launch.hpp"

using namespace rvh;
using namespace rvl;

namespace rvl {

// exclusive scan of n counts -> (n + 1) uint64 prefixes.  The counts are uint32 values or (pop) the popcounts of 64-bit
// words read in place.  want_total: wait for the result and return the total (else 0, nothing is waited for).
uint64_t device_exclusive_scan(rv_ctx *ctx, const void *counts, uint64_t n, DevBufRef &excl, bool pop, bool want_total, unsigned long long *total_dst) {
    excl = pool_alloc(ctx, (n + 1) * 8 + 16);
    if (n == 0) {
        RV_HIP(hipMemsetAsync(excl->ptr, 0, 8, ctx->stream));
        return 0;
    }
    const uint64_t nblocks = (n + rvk::kScanBlock - 1) / rvk::kScanBlock;
    DevBufRef sums = pool_alloc(ctx, nblocks * 8 + 16);
    unsigned long long *total_at = total_dst ? total_dst : &prepare_ctrl(ctx, 0)->pops[0];
    const dim3 grid(static_cast<uint32_t>(nblocks)), block(rvk::kScanThreads);
    if (pop) hipLaunchKernelGGL(rvk::scan_block_sums<true>, grid, block, 0, ctx->stream, counts, n, static_cast<uint64_t *>(sums->ptr));
    else hipLaunchKernelGGL(rvk::scan_block_sums<false>, grid, block, 0, ctx->stream, counts, n, static_cast<uint64_t *>(sums->ptr));
    hipLaunchKernelGGL(rvk::scan_sums_inplace, dim3(1), dim3(1024), 0, ctx->stream, static_cast<uint64_t *>(sums->ptr), nblocks, total_at);
    if (pop) hipLaunchKernelGGL(rvk::scan_apply<true>, grid, block, 0, ctx->stream, counts, n, static_cast<const uint64_t *>(sums->ptr), static_cast<uint64_t *>(excl->ptr));
    else hipLaunchKernelGGL(rvk::scan_apply<false>, grid, block, 0, ctx->stream, counts, n, static_cast<const uint64_t *>(sums->ptr), static_cast<uint64_t *>(excl->ptr));
    RV_HIP(hipGetLastError());
    // `sums` goes back to the pool here; every later user runs on this stream, after the kernels that read it
    if (!want_total || total_dst) return 0;
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
    const size_t wb = zeroed_bitmap_bytes(rows);
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
static void launch_str_copy(rv_ctx *ctx, const rvk::StrGather &g, uint64_t nblocks) {
    hipLaunchKernelGGL(rvk::str_gather_copy<8192>, dim3(static_cast<uint32_t>(nblocks)), dim3(64), 0, ctx->stream, g);  // one wave per block of kStrBlock elements
}
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
    launch_str_copy(ctx, g, nblocks);
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
        const size_t wb = zeroed_bitmap_bytes(rows);
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
    // half the rows or more expected to survive (the selectivity the pass itself was sized by): source-tile order, without the
    // (start, length) lists -- 16 bytes of traffic per survivor less, 0.8 GB of offsets and one light launch more (tools/str_sweep.py:
    // 1.44 against 1.45 ms per 2e8 rows at 50 %, 1.64 / 1.83 at 70 %, 1.29 / 1.15 at 30 %)
    L.tiles = ctx->opt_str_tiles_from == 1 ||  // (diagnostic: always)
              (ctx->opt_str_tiles_from >= 0 && ranges.expected_selectivity >= (ctx->opt_str_tiles_from > 0 ? ctx->opt_str_tiles_from / 100.0 : rvt::kStrTilesFrom));
    if (src->validity) {  // the output bitmap: the source's, compacted by the same selection at the same offsets
        const size_t wb = zeroed_bitmap_bytes(cap);
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
    if (L.tiles) {  // the whole String side in source-tile order, queued here: nothing of it is sized by the survivor count
        const uint64_t ntiles = (nwords + 7) / 8, ngroups = (ntiles + rvk::kStrGroup - 1) / rvk::kStrGroup;
        const uint64_t cap_bytes = std::min<uint64_t>(src->data_bytes, 0x7FFFFFFFull);  // the survivors' bytes are among the source's
        o->values = pool_alloc(ctx, std::max<size_t>(cap_bytes + 8, 16));
        L.block_sums = pool_alloc(ctx, ntiles * 8 + 16);
        L.groups = pool_alloc(ctx, ngroups * 8 + 16);
        RV_HIP(hipMemsetAsync(o->offsets->ptr, 0, 4, ctx->stream));  // offsets[0] of an empty result (string.rs:13)
        rvk::SelStrTiles t{};
        t.sel = static_cast<const uint64_t *>(sel->values->ptr);
        t.nwords = nwords;
        t.range_offsets = static_cast<const uint64_t *>(ranges.offsets->ptr);
        t.range_rows = ranges.range_rows;
        t.offsets = static_cast<const int32_t *>(src->offsets->ptr);
        t.validity = src->validity ? static_cast<const uint8_t *>(src->validity->ptr) : nullptr;
        t.validity_bytes = src->validity ? src->validity->bytes : 0;
        t.offset = src->offset;
        t.length = src->length;
        t.data = static_cast<const uint8_t *>(src->values->ptr);
        t.tile_sums = static_cast<unsigned long long *>(L.block_sums->ptr);
        t.group_base = static_cast<const uint64_t *>(L.groups->ptr);
        t.cap_rows = cap;
        t.out_offsets = static_cast<int32_t *>(o->offsets->ptr);
        t.out_data = static_cast<uint8_t *>(o->values->ptr);
        hipLaunchKernelGGL(rvk::sel_str_tile_sums, dim3(static_cast<uint32_t>((nwords + 255) / 256)), dim3(256), 0, ctx->stream, t);
        const bool wide = ngroups > 64;  // many groups: their sums by many workgroups, the scan launch only scans
        if (wide)
            hipLaunchKernelGGL(rvk::str_group_sums, dim3(static_cast<uint32_t>((ngroups + 3) / 4)), dim3(256), 0, ctx->stream,
                               static_cast<const uint64_t *>(L.block_sums->ptr), ntiles, static_cast<uint64_t *>(L.groups->ptr));
        hipLaunchKernelGGL(rvk::str_sums_scan, dim3(1), dim3(1024), 0, ctx->stream, wide ? nullptr : static_cast<const unsigned long long *>(L.block_sums->ptr),
                           ntiles, static_cast<uint64_t *>(L.groups->ptr), &ctrl->pops[0], static_cast<int32_t *>(nullptr), uint64_t{0});
        // the window: the bytes an average tile is expected to keep + 15 % and 256 bytes (an outlier tile goes straight to the output);
        // it bounds the tiles resident on a CU
        const double avg = static_cast<double>(src->data_bytes) / static_cast<double>(src->length ? src->length : 1);
        const double keep = ranges.expected_selectivity > 0 ? std::min(1.0, ranges.expected_selectivity * 1.15) : 1.0;
        const double want = avg * rvk::kStrTile * keep + 256;
        const dim3 grid(static_cast<uint32_t>(ntiles)), wave(64);
        if (want <= 4096)
            hipLaunchKernelGGL(rvk::sel_str_tile_copy<4096>, grid, wave, 0, ctx->stream, t);
        else if (want <= 6144)
            hipLaunchKernelGGL(rvk::sel_str_tile_copy<6144>, grid, wave, 0, ctx->stream, t);
        else if (want <= 8192)
            hipLaunchKernelGGL(rvk::sel_str_tile_copy<8192>, grid, wave, 0, ctx->stream, t);
        else if (want <= 12288)
            hipLaunchKernelGGL(rvk::sel_str_tile_copy<12288>, grid, wave, 0, ctx->stream, t);
        else
            hipLaunchKernelGGL(rvk::sel_str_tile_copy<16384>, grid, wave, 0, ctx->stream, t);
        RV_HIP(hipGetLastError());
        L.queued = true;
        return;
    }
    L.lengths = pool_alloc(ctx, cap * 4 + 16);
    L.starts = pool_alloc(ctx, cap * 4 + 16);
    const uint64_t max_blocks = (cap + rvk::kStrBlock - 1) / rvk::kStrBlock;
    const size_t sums_bytes = (max_blocks * 8 + 16 + 255) & ~static_cast<size_t>(255);  // (one fill launch: launch.hpp, zeroed_bitmap_bytes)
    L.block_sums = pool_alloc(ctx, sums_bytes);
    RV_HIP(hipMemsetAsync(L.block_sums->ptr, 0, sums_bytes, ctx->stream));
    rvk::SelStr q{};
    q.sel = static_cast<const uint64_t *>(sel->values->ptr);
    q.nwords = nwords;
    q.range_offsets = static_cast<const uint64_t *>(ranges.offsets->ptr);
    q.range_rows = ranges.range_rows;
    q.block_sums = static_cast<unsigned long long *>(L.block_sums->ptr);
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
    if (L.tiles) return;  // everything was queued behind the pass
    const uint64_t cap_bytes = std::min<uint64_t>(L.src->data_bytes, 0x7FFFFFFFull);  // the survivors' bytes are among the source's
    o->values = pool_alloc(ctx, std::max<size_t>(cap_bytes + 8, 16));
    if (rows == 0) {
        RV_HIP(hipMemsetAsync(o->offsets->ptr, 0, 4, ctx->stream));
        return;
    }
    const uint64_t nblocks = (rows + rvk::kStrBlock - 1) / rvk::kStrBlock, ngroups = (nblocks + rvk::kStrGroup - 1) / rvk::kStrGroup;
    L.groups = pool_alloc(ctx, ngroups * 8 + 16);
    const bool wide = ngroups > 64;  // many groups: their sums by many workgroups, the scan launch only scans
    if (wide)
        hipLaunchKernelGGL(rvk::str_group_sums, dim3(static_cast<uint32_t>((ngroups + 3) / 4)), dim3(256), 0, ctx->stream,
                           static_cast<const uint64_t *>(L.block_sums->ptr), nblocks, static_cast<uint64_t *>(L.groups->ptr));
    hipLaunchKernelGGL(rvk::str_sums_scan, dim3(1), dim3(1024), 0, ctx->stream, wide ? nullptr : static_cast<const unsigned long long *>(L.block_sums->ptr),
                       nblocks, static_cast<uint64_t *>(L.groups->ptr), &L.ctrl->pops[0], static_cast<int32_t *>(o->offsets->ptr), rows);
    rvk::StrGather g{};
    g.data = static_cast<const uint8_t *>(L.src->values->ptr);
    g.n = rows;
    g.lengths = static_cast<uint32_t *>(L.lengths->ptr);
    g.starts = static_cast<int32_t *>(L.starts->ptr);
    g.block_sums = static_cast<const uint64_t *>(L.block_sums->ptr);
    g.group_base = static_cast<const uint64_t *>(L.groups->ptr);
    g.total_bytes = ~0ull;  // out_offsets[rows] is str_sums_scan's
    g.out_offsets = static_cast<int32_t *>(o->offsets->ptr);
    g.out_data = static_cast<uint8_t *>(o->values->ptr);
    launch_str_copy(ctx, g, nblocks);
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

void bool_compact_queue(rv_ctx *ctx, const rv_dcolumn *src, const rv_dcolumn *sel, const RangeOffsets &ranges, Ctrl *ctrl, int slot, BoolCompactLaunch &L) {
    auto o = std::make_unique<rv_dcolumn>();
    o->dtype = RV_BOOLEAN;
    const uint64_t nwords = (sel->length + 63) / 64;
    // The output bitmaps are zeroed (waves OR their edge words in): sized for every row that may survive that is 2 x 62 MB and 20 us of
    // fills per 5e8 rows between the pass and this launch.  With an expectation of the selectivity they are sized for it + 25 %; a
    // wave whose run would pass that writes nothing, and bool_compact_result hands the column to the scan path (compact_boolean).
    uint64_t cap = ranges.out_capacity;
    if (ranges.expected_selectivity >= 0)
        cap = std::min<uint64_t>(cap, static_cast<uint64_t>(ranges.expected_selectivity * rvt::kBoolCapFactor * static_cast<double>(sel->length)) + rvt::kBoolCapSlack);
    if (ctx->opt_bool_cap > 0) cap = std::min<uint64_t>(cap, static_cast<uint64_t>(ctx->opt_bool_cap));  // (tests: a bound the count passes)
    L.cap_rows = cap;
    const size_t wb = zeroed_bitmap_bytes(cap);
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
    if (rows > L.cap_rows) {  // more survivors than the bitmaps were sized for: nothing usable was written
        L.col.reset();
        return nullptr;
    }
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

}  // namespace rvl
