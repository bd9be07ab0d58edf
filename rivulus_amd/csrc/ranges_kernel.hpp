// Later column groups of a wide projection (the eager Filter keeps EVERY column, plan.rs:132-147): compacted by the selection
// bitmap AT THE FIRST PASS'S WAVE OFFSETS.  The first group's pass has left, for every range of its waves' rows, the output row
// of the range's first survivor (FusedParams::wave_offsets): a wave of this kernel finds where its 1024 rows go from that offset and
// the popcounts of its own selection words -- no look-back chain, no scanner wave, no descriptors, no persistent grid, nothing
// shared between waves -- and the launches of all later groups are queued back to back behind the first pass, with outputs of the
// exact size (round 3: every group was a pass of its own, chained, with a host round trip in between).
//
// One wave per 1024 rows: row set j (j = 0..15) is rows 64 j + lane, its selection word wave-uniform (a row's survival is bit `lane`
// of it, its rank a popcount of the bits below), so the control flow is scalar.  Survivors are staged by rank in an LDS slice of
// kStage slots per column and written out as whole coalesced runs whenever the next row set might not fit -- at 10 % selectivity
// once per wave and column, at 84 % every three to four row sets (>= 1.5 KiB per column and flush).  8-byte value columns; a
// nullable one's validity bits are compacted next to it by bits_compact_kernel at the same offsets (query.hip).
#pragma once
#include "device_common.hpp"
#include "scan_frontend.hpp"

namespace rvk {
constexpr int kRangesMaxCols = 4;
constexpr int kRangesStage = 256;  // staged survivors per wave and column (2 KiB): 4 waves x 4 columns = 32 KiB per workgroup
struct RangesCompact {
    const uint64_t *sel;  // selection words (bits past the last row zero)
    uint64_t nwords;
    uint64_t n;           // rows
    const uint64_t *range_offsets;
    uint32_t range_rows;  // divides 4096
    uint32_t pad;
    uint64_t out_capacity;  // rows the outputs hold: survivors past it are dropped (outputs sized before the survivor count was known)
    const void *in[kRangesMaxCols];  // first value of each column (offset applied)
    uint64_t *out[kRangesMaxCols];   // [rows]
    // a column with a null bitmap: a null survivor's slot holds 0 (PrimitiveArrayBuilder::append_null, primitive.rs:168-175); its
    // validity bits are compacted by bits_compact_kernel at the same wave offsets
    const uint8_t *validity[kRangesMaxCols];  // or nullptr
    uint64_t validity_bytes[kRangesMaxCols];
    uint64_t bit_offset[kRangesMaxCols];      // bit of row 0 in `validity`
};
// RecordBatch::filter by a BooleanArray (record_batch.rs:221-243; the reference's streaming filter, stream.rs:136-158): the predicate IS
// a bitmap, so the selection and the survivor count of every 1024 rows cost one pass over 2-3 bits per row -- no chained pass at all
// (one over nothing but bitmaps still walks every tile through the chain: 0.60 ms per 5e8 rows).  sel[w] = values & validity at the
// column's bit offset, tail bits zero; counts[r] = survivors of rows [1024 r, 1024 r + 1024).  A scan of the counts gives
// compact_ranges_kernel its offsets.
struct MaskSelect {
    const uint8_t *values;
    uint64_t values_bytes;
    const uint8_t *validity;  // or nullptr
    uint64_t validity_bytes;
    uint64_t offset;          // bit of row 0
    uint64_t n;               // rows
    uint64_t *sel;            // [ceil(n / 64)]
    uint32_t *counts;         // [ceil(n / 1024)]
    unsigned long long *batch_counts;  // nullptr, or [ceil(n / 1024)] in pinned host memory: the same counts where the caller of a window of
                                       // 1024-row RecordBatches reads them (its batches ARE the ranges)
};
static __global__ __launch_bounds__(256) void mask_select_kernel(const MaskSelect p) {
    const uint64_t w = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x, nwords = (p.n + 63) / 64;
    uint64_t m = 0;
    if (w < nwords) {
        m = load_bits64(p.values, p.offset + w * 64, p.values_bytes);
        if (p.validity) m &= load_bits64(p.validity, p.offset + w * 64, p.validity_bytes);
        if (p.n - w * 64 < 64) m &= (1ull << (p.n - w * 64)) - 1;
        p.sel[w] = m;
    }
    uint32_t c = static_cast<uint32_t>(__popcll(m));
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) c += static_cast<uint32_t>(__shfl_xor(static_cast<int>(c), d, 64));  // 16 words = 1024 rows
    if ((threadIdx.x & 15) == 0 && w < nwords) {
        p.counts[w / 16] = c;
        if (p.batch_counts) p.batch_counts[w / 16] = c;
    }
}

template <int NCOLS, bool NULLS>
static __global__ __launch_bounds__(256) void compact_ranges_kernel(const RangesCompact p) {
    constexpr int SETS = 16, HALF = 8;
    __shared__ uint64_t stage[4][NCOLS][kRangesStage];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t w0 = (static_cast<uint64_t>(blockIdx.x) * 4 + wave) * SETS;  // first selection word of the wave's rows
    if (w0 >= p.nwords) return;  // wave-uniform
    const uint64_t row0 = w0 * 64;
    // the selection words from the start of the wave's range of the first pass to the end of its rows: lane k holds word range_first + k
    const uint32_t range_words = p.range_rows / 64;
    const uint64_t range_first = range_words > SETS ? (w0 / range_words) * range_words : w0;
    const uint32_t before = static_cast<uint32_t>(w0 - range_first);  // 0 .. 48
    const uint64_t wq = range_first + lane;
    const uint64_t mine = (static_cast<uint32_t>(lane) < before + SETS && wq < p.nwords) ? p.sel[wq] : 0;
    const uint64_t range_at = p.range_offsets[row0 / p.range_rows];
    const uint32_t cntw = static_cast<uint32_t>(__popcll(mine));
    const uint32_t incl = wave_scan_u32(cntw), excl = incl - cntw;
    const uint32_t in_front = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(excl), static_cast<int>(before)));
    const uint32_t total = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(incl), static_cast<int>(before + SETS - 1))) - in_front;
    if (total == 0) return;  // wave-uniform
    const uint64_t P = range_at + in_front;  // output row of the wave's first survivor
    const uint64_t left = p.n - row0;
    const uint32_t nbytes = uniform32(static_cast<uint32_t>(left < 64 * SETS ? left : 64 * SETS) * 8u);
    // validity words of the wave's rows: lane k holds the aligned 64-bit word (first bit >> 6) + k of every nullable column's bitmap
    // (17 words cover the 16 unaligned windows), ONE vector load per column; a row set's window is cut out with two readlanes and a
    // scalar funnel shift (a load per row set and column -- 64 more memory instructions per wave -- halved the kernel's rate)
    uint64_t vwords[NCOLS];
    if constexpr (NULLS) {
#pragma unroll
        for (int c = 0; c < NCOLS; ++c) {
            vwords[c] = ~0ull;
            if (p.validity[c] && lane <= SETS) vwords[c] = load_word_safe(p.validity[c], ((p.bit_offset[c] + row0) >> 6) + lane, p.validity_bytes[c]);
        }
    }
    uint32_t filled = 0;   // survivors in the stage
    uint64_t flushed = 0;  // survivors written so far
    auto flush = [&]() {
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the staged values of every lane are in place
#pragma unroll
        for (int c = 0; c < NCOLS; ++c)
            for (uint32_t i = lane; i < filled; i += 64)
                if (P + flushed + i < p.out_capacity) __builtin_nontemporal_store(stage[wave][c][i], &p.out[c][P + flushed + i]);
        __builtin_amdgcn_wave_barrier();  // (LDS instructions of one wave execute in order: the next writes follow these reads)
        flushed += filled;
        filled = 0;
    };
#pragma unroll
    for (int h = 0; h < SETS / HALF; ++h) {
        uint64_t m[HALF];
        uint64_t v[NCOLS][HALF];
#pragma unroll
        for (int j = 0; j < HALF; ++j) {
            m[j] = readlane64(mine, static_cast<int>(before) + h * HALF + j);  // wave-uniform
            const bool taken = (m[j] >> lane) & 1;
#pragma unroll
            for (int c = 0; c < NCOLS; ++c) {
                v[c][j] = 0;
                if (taken) {  // only survivors request their value (every line is fetched once either way)
                    const uint64_t base = uniform64(reinterpret_cast<uint64_t>(p.in[c]) + row0 * 8);
                    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(base), 0, nbytes, 0x00020000);
                    const rv_u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rsrc, lane * 8, (h * HALF + j) * 512, kStreamPolicy);
                    v[c][j] = (static_cast<uint64_t>(t.y) << 32) | t.x;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < HALF; ++j) {
            const uint32_t cj = static_cast<uint32_t>(__popcll(m[j]));  // wave-uniform
            if (cj == 0) continue;
            if (filled + cj > kRangesStage) flush();
            if ((m[j] >> lane) & 1) {
                const uint32_t slot = filled + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m[j] >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m[j]), 0u));
#pragma unroll
                for (int c = 0; c < NCOLS; ++c) {
                    uint64_t x = v[c][j];
                    if constexpr (NULLS) {  // (here, not where the value is requested: a use right behind each load serialises the loads)
                        if (p.validity[c]) {  // wave-uniform
                            const uint32_t sh = static_cast<uint32_t>((p.bit_offset[c] + row0) & 63);
                            const uint64_t lo = readlane64(vwords[c], h * HALF + j), hi = readlane64(vwords[c], h * HALF + j + 1);
                            const uint64_t vw = sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
                            if (!((vw >> lane) & 1)) x = 0;
                        }
                    }
                    stage[wave][c][slot] = x;
                }
            }
            filled += cj;
        }
    }
    if (filled) flush();
}
}  // namespace rvk
