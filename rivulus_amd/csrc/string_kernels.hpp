// StringArray on the device: gather of variable-width elements.
// Layout = the reference's (string.rs:9-15): UTF-8 bytes + int32 offsets (element i of the
// buffer spans offsets[i] .. offsets[i+1]) + optional validity bitmap + element offset.
// take / filter re-build the array from the surviving elements (record_batch.rs:163-170 ->
// StringArray::new, string.rs:19-57): offsets restart at 0, a null contributes no bytes, the
// bitmap is kept only if a null survived.
//
// Kernels: a three-step exclusive scan over uint32 counts (block sums -> scan of the sums ->
// per-block scan), selection bitmap -> ascending row indices, element lengths, byte copy,
// and the offsets/validity part of concat.  All HBM-bound integer/byte work, no MFMA.
#pragma once

#include "device_common.hpp"

namespace rvk {

// ---- exclusive scan of uint32 counts into uint64 prefixes ---------------------------------------
constexpr int kScanThreads = 256;
constexpr int kScanItems = 8;
constexpr int kScanBlock = kScanThreads * kScanItems;  // 2048 counts per workgroup

// the counts are uint32 values, or (POP) the popcounts of 64-bit selection words read in place (no counts buffer)
template <bool POP>
__device__ __forceinline__ uint32_t scan_item(const void *in, uint64_t i) {
    if constexpr (POP) return static_cast<uint32_t>(__popcll(static_cast<const uint64_t *>(in)[i]));
    else return static_cast<const uint32_t *>(in)[i];
}
// sums[b] = sum of in[b*2048 .. )
template <bool POP>
static __global__ __launch_bounds__(kScanThreads) void scan_block_sums(const void *in, uint64_t n, uint64_t *sums) {
    __shared__ uint64_t s_wave[kScanThreads / 64];
    const uint64_t base = static_cast<uint64_t>(blockIdx.x) * kScanBlock + threadIdx.x * kScanItems;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k)
        if (base + k < n) acc += scan_item<POP>(in, base + k);
    acc = wave_sum64(acc);
    if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t t = 0;
        for (int w = 0; w < kScanThreads / 64; ++w) t += s_wave[w];
        sums[blockIdx.x] = t;
    }
}

// one workgroup: sums[0..nblocks) -> exclusive prefixes in place; *total = grand total.  Eight consecutive entries per
// thread and round (8192 per round): the rounds are a dependent chain of global round trips, so fewer, wider rounds.
constexpr int kSumsItems = 8;
static __global__ __launch_bounds__(1024) void scan_sums_inplace(uint64_t *sums, uint64_t nblocks, unsigned long long *total) {
    __shared__ uint64_t s_wave[16];
    __shared__ uint64_t s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint64_t base = 0; base < nblocks; base += 1024 * kSumsItems) {
        const uint64_t i0 = base + static_cast<uint64_t>(threadIdx.x) * kSumsItems;
        uint64_t x[kSumsItems], mine = 0;
#pragma unroll
        for (int k = 0; k < kSumsItems; ++k) {
            x[k] = i0 + k < nblocks ? sums[i0 + k] : 0;
            mine += x[k];
        }
        uint64_t incl = mine;  // inclusive scan inside the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t y = (static_cast<uint64_t>(__shfl_up(static_cast<uint32_t>(incl >> 32), d, 64)) << 32) |
                               __shfl_up(static_cast<uint32_t>(incl), d, 64);
            if (lane >= d) incl += y;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint64_t run = s_carry + incl - mine;
        for (int w = 0; w < wave; ++w) run += s_wave[w];
#pragma unroll
        for (int k = 0; k < kSumsItems; ++k) {
            if (i0 + k < nblocks) sums[i0 + k] = run;
            run += x[k];
        }
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = run;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = s_carry;
}

// out[i] = sums[block] + exclusive prefix inside the block; out[n] = total (written by the last block)
template <bool POP>
static __global__ __launch_bounds__(kScanThreads) void scan_apply(const void *in, uint64_t n, const uint64_t *sums, uint64_t *out) {
    __shared__ uint64_t s_wave[kScanThreads / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t base = static_cast<uint64_t>(blockIdx.x) * kScanBlock + threadIdx.x * kScanItems;
    uint32_t v[kScanItems];
    uint64_t mine = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        v[k] = base + k < n ? scan_item<POP>(in, base + k) : 0;
        mine += v[k];
    }
    uint64_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t y = (static_cast<uint64_t>(__shfl_up(static_cast<uint32_t>(incl >> 32), d, 64)) << 32) |
                           __shfl_up(static_cast<uint32_t>(incl), d, 64);
        if (lane >= d) incl += y;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint64_t run = sums[blockIdx.x] + incl - mine;
    for (int w = 0; w < wave; ++w) run += s_wave[w];
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        if (base + k < n) out[base + k] = run;
        run += v[k];
        if (base + k + 1 == n) out[n] = run;
    }
}

// ---- selection bitmap -> ascending row indices ------------------------------------------------------
// counts[w] = popcount of selection word w (bits past n are zero by construction)
static __global__ void sel_word_counts(const uint64_t *sel, uint64_t nwords, uint32_t *counts) {
    const uint64_t w = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (w < nwords) counts[w] = static_cast<uint32_t>(__popcll(sel[w]));
}
static __global__ void sel_expand_indices(const uint64_t *sel, uint64_t nwords, const uint64_t *excl, uint64_t *indices) {
    const uint64_t w = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (w >= nwords) return;
    uint64_t m = sel[w], o = excl[w];
    while (m) {
        indices[o++] = w * 64 + static_cast<uint64_t>(__builtin_ctzll(m));
        m &= m - 1;
    }
}

// ---- Boolean column compaction by the selection bitmap ------------------------------------------------
// out bits = the bits of (src [& mask]) at the set positions of sel, in order (a software PEXT per 64-row word,
// cheap because few rows survive), appended at the word's exclusive survivor count.  One lane per selection
// word, one wave per 64 words; the wave assembles its run in LDS and stores it with one coalesced pass
// (first / last output word shared with the neighbouring waves: atomicOr into the zero-filled buffer).
// Reads 2-3 bits per row: a Boolean column costs ~0.1 ms per 1e9 rows here instead of riding through the
// fused pass as a byte-staged stream.
struct BitsCompact {
    const uint64_t *sel;   // selection words (bits past n zero)
    uint64_t nwords;
    const uint8_t *src;    // source bit buffer
    uint64_t src_bytes;
    const uint8_t *mask;   // optional second buffer ANDed in (Boolean values under their validity), or nullptr
    uint64_t mask_bytes;
    uint64_t offset;       // bit offset of row 0 in src / mask
    const uint64_t *excl;  // [nwords + 1] exclusive survivor counts per selection word, or nullptr with
    // range_offsets: [ceil(rows / range_rows)] output row of the first survivor of every range of range_rows rows (the fused
    // pass's wave ranges, FusedParams::wave_offsets); range_rows divides 4096, so a wave's 64 * kBcWords words start a range
    const uint64_t *range_offsets;
    uint32_t range_rows;
    uint64_t out_capacity;  // rows the outputs hold (speculative sizing): a wave whose run passes it writes nothing
    uint64_t *out;         // zero-filled output bitmap
    unsigned long long *pop;  // += number of set output bits
    // optional second stream compacted by the same selection in the same launch (a Boolean column's validity next to its
    // values): src2 / out2 / pop2, same offset
    const uint8_t *src2;
    uint64_t src2_bytes;
    uint64_t *out2;
    unsigned long long *pop2;
};
// Two selection words per lane (a wave owns 128 words = 8192 rows per step): the loads and the two PEXT chains of a lane are
// independent, which is what this latency-bound kernel was short of (one word per lane: 0.13 ms per 5e8 rows for two
// streams, i.e. 1.4 TB/s of the 190 MB it reads).
constexpr int kBcWords = 2;
static __global__ __launch_bounds__(256) void bits_compact_kernel(const BitsCompact p) {
    constexpr int W = kBcWords, SPAN = 64 * W;
    __shared__ uint64_t s_out[4][SPAN + 2], s_out2[4][SPAN + 2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t ones = 0, ones2 = 0;  // set output bits seen by this lane; ONE atomic per wave at the end (grid-stride loop:
                                   // an atomic per 64 words on one address would cap the kernel at ~88 waves/us)
    const bool two = p.src2 != nullptr;
    const uint64_t nchunks = (p.nwords + 4 * SPAN - 1) / (4 * SPAN);
    for (uint64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        const uint64_t w0 = (chunk * 4 + wave) * SPAN;  // lane's words: w0 + k * 64 + lane, in row order k-major
        const bool live = w0 < p.nwords;                // wave-uniform
        uint64_t s[W], x[W], x2[W];
#pragma unroll
        for (int k = 0; k < W; ++k) {
            const uint64_t w = w0 + static_cast<uint64_t>(k) * 64 + lane;
            s[k] = (live && w < p.nwords) ? p.sel[w] : 0;
        }
        // every load of the step goes out before the first is used (the source words do not wait for the selection word:
        // at any selectivity worth compacting nearly every word has a survivor), the wave's output position with them
        const uint64_t base = !live ? 0 : (p.excl ? p.excl[w0] : p.range_offsets[w0 * 64 / p.range_rows]);  // first output bit of the wave
#pragma unroll
        for (int k = 0; k < W; ++k) {
            const uint64_t w = w0 + static_cast<uint64_t>(k) * 64 + lane;
            x[k] = x2[k] = 0;
            if (live && w < p.nwords) {
                x[k] = load_bits64(p.src, p.offset + w * 64, p.src_bytes);
                if (p.mask) x[k] &= load_bits64(p.mask, p.offset + w * 64, p.mask_bytes);
                if (two) x2[k] = load_bits64(p.src2, p.offset + w * 64, p.src2_bytes);
            }
        }
#pragma unroll
        for (int k = 0; k < W; ++k) x[k] &= s[k], x2[k] &= s[k];
        // PEXT(x, s), PEXT(x2, s) per word, as 32-bit HALVES: four independent chains per lane run side by side, every step
        // full-rate 32-bit ALU (64-bit shifts are quarter rate, and a 64-bit chain is as long as the word's set bits: the
        // kernel spent most of its 0.14 ms per 5e8 rows here), the halves joined by one 64-bit shift at the end
        uint64_t c[W], c2[W];
        uint32_t cnt[W];
        {
            constexpr int H = 2 * W;
            uint32_t m[H], xs[H], x2s[H], cs[H], c2s[H], j[H];
#pragma unroll
            for (int h = 0; h < H; ++h) {
                const int k = h >> 1, sh = (h & 1) * 32;
                m[h] = static_cast<uint32_t>(s[k] >> sh);
                xs[h] = static_cast<uint32_t>(x[k] >> sh);
                x2s[h] = static_cast<uint32_t>(x2[k] >> sh);
                cs[h] = c2s[h] = 0;
                j[h] = 0;
            }
            // The loop runs as long as the fullest half of the wave has set bits -- or, when most rows survive, as long as its emptiest
            // half has CLEAR bits: deleting the clear positions from (x & m), highest first, leaves the same PEXT.  One mode per
            // wave and step (wave-uniform: no divergence between the two loop bodies), the one whose longest chain is shorter:
            // at 84 % selectivity ~11 rounds instead of ~32 (0.20 -> 0.17 ms per 5e8 rows, two streams).
            uint32_t most_set = 0, most_clear = 0;
#pragma unroll
            for (int h = 0; h < H; ++h) {
                const uint32_t pc = static_cast<uint32_t>(__builtin_popcount(m[h]));
                most_set = pc > most_set ? pc : most_set;
                const uint32_t clear = m[h] ? 32u - pc : 0u;  // (a half without a survivor has nothing to delete from)
                most_clear = clear > most_clear ? clear : most_clear;
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                const uint32_t a = static_cast<uint32_t>(__shfl_xor(static_cast<int>(most_set), d, 64)), b = static_cast<uint32_t>(__shfl_xor(static_cast<int>(most_clear), d, 64));
                most_set = a > most_set ? a : most_set;
                most_clear = b > most_clear ? b : most_clear;
            }
            if (most_clear < most_set) {  // wave-uniform
                uint32_t z[H];
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    z[h] = m[h] ? ~m[h] : 0u;
                    cs[h] = xs[h], c2s[h] = x2s[h];  // (both already ANDed with the selection word)
                    j[h] = static_cast<uint32_t>(__builtin_popcount(m[h]));
                }
                bool any = most_clear != 0;
                while (any) {
                    any = false;
#pragma unroll
                    for (int h = 0; h < H; ++h) {
                        if (z[h]) {
                            const uint32_t i = 31u - static_cast<uint32_t>(__builtin_clz(z[h])), low = (1u << i) - 1u;
                            cs[h] = (cs[h] & low) | (((cs[h] >> 1) >> i) << i);
                            c2s[h] = (c2s[h] & low) | (((c2s[h] >> 1) >> i) << i);
                            z[h] &= low;  // every clear position above i is gone already
                        }
                        any = any || z[h] != 0;
                    }
                }
            } else {
                bool any = most_set != 0;
                while (any) {
                    any = false;
#pragma unroll
                    for (int h = 0; h < H; ++h) {
                        if (m[h]) {
                            const uint32_t i = static_cast<uint32_t>(__builtin_ctz(m[h]));
                            cs[h] |= ((xs[h] >> i) & 1u) << j[h];
                            c2s[h] |= ((x2s[h] >> i) & 1u) << j[h];
                            ++j[h];
                            m[h] &= m[h] - 1;
                        }
                        any = any || m[h] != 0;
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < W; ++k) {
                const uint32_t lo_bits = j[2 * k];  // set bits of the word's low half == length of its compacted low part
                cnt[k] = lo_bits + j[2 * k + 1];
                c[k] = static_cast<uint64_t>(cs[2 * k]) | (lo_bits < 64 ? static_cast<uint64_t>(cs[2 * k + 1]) << lo_bits : 0ull);
                c2[k] = static_cast<uint64_t>(c2s[2 * k]) | (lo_bits < 64 ? static_cast<uint64_t>(c2s[2 * k + 1]) << lo_bits : 0ull);
            }
        }
        // output position of every word: words are in row order k-major (word k*64 + lane), so an inclusive scan per k and
        // the totals of the k before it
        uint32_t pos[W], before = 0;
#pragma unroll
        for (int k = 0; k < W; ++k) {
            uint32_t incl = cnt[k];
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t y = __shfl_up(incl, d, 64);
                if (lane >= d) incl += y;
            }
            pos[k] = before + incl - cnt[k];
            before += __shfl(incl, 63, 64);
        }
        const uint32_t total = before;
        const uint32_t lead = static_cast<uint32_t>(base & 63);
        for (int q = lane; q < SPAN + 2; q += 64) s_out[wave][q] = 0, s_out2[wave][q] = 0;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < W; ++k)
            if (cnt[k]) {
                const uint32_t at = lead + pos[k], q = at >> 6, sh = at & 63;
                atomicOr(reinterpret_cast<unsigned long long *>(&s_out[wave][q]), static_cast<unsigned long long>(c[k] << sh));
                if (sh && (c[k] >> (64 - sh))) atomicOr(reinterpret_cast<unsigned long long *>(&s_out[wave][q + 1]), static_cast<unsigned long long>(c[k] >> (64 - sh)));
                if (two) {
                    atomicOr(reinterpret_cast<unsigned long long *>(&s_out2[wave][q]), static_cast<unsigned long long>(c2[k] << sh));
                    if (sh && (c2[k] >> (64 - sh))) atomicOr(reinterpret_cast<unsigned long long *>(&s_out2[wave][q + 1]), static_cast<unsigned long long>(c2[k] >> (64 - sh)));
                }
            }
        __syncthreads();
        const bool fits = p.excl != nullptr || base + total <= p.out_capacity;  // wave-uniform; the fused pass flags the overflow
        const uint32_t words = live && total && fits ? (lead + total + 63) >> 6 : 0;
        for (uint32_t q = lane; q < words; q += 64) {
            const bool edge = q == 0 || q + 1 == words;
            const uint64_t v = s_out[wave][q];
            uint64_t *dst = p.out + (base >> 6) + q;
            if (edge) {
                if (v) atomicOr(reinterpret_cast<unsigned long long *>(dst), static_cast<unsigned long long>(v));
            } else {
                *dst = v;
            }
            if (two) {
                const uint64_t v2 = s_out2[wave][q];
                uint64_t *dst2 = p.out2 + (base >> 6) + q;
                if (edge) {
                    if (v2) atomicOr(reinterpret_cast<unsigned long long *>(dst2), static_cast<unsigned long long>(v2));
                } else {
                    *dst2 = v2;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < W; ++k) {
            ones += static_cast<uint64_t>(__popcll(c[k]));
            ones2 += static_cast<uint64_t>(__popcll(c2[k]));
        }
        __syncthreads();  // s_out is reused by the next chunk
    }
    ones = wave_sum64(ones);
    if (lane == 0 && ones) striped_add(p.pop, static_cast<unsigned long long>(ones));
    if (two) {
        ones2 = wave_sum64(ones2);
        if (lane == 0 && ones2) striped_add(p.pop2, static_cast<unsigned long long>(ones2));
    }
}

// ---- gather ------------------------------------------------------------------------------------------
struct StrGather {
    const int32_t *offsets;   // source offsets buffer (absolute element index)
    const uint8_t *data;      // source bytes
    const uint8_t *validity;  // source bitmap or nullptr
    uint64_t validity_bytes;
    uint64_t offset;          // element offset of the source column
    uint64_t src_length;      // logical length of the source (index bound)
    const uint64_t *indices;  // [n] logical row indices
    uint64_t n;
    uint32_t *lengths;        // [n] out: byte length of each gathered element (0 under a null)
    int32_t *starts;          // [n] out: first source byte of each gathered element (the copy pass then reads neither the
                              //     index list nor the offsets array again: both are scattered, one 128-byte line per element)
    uint64_t *out_validity;   // [ceil(n/64)] words, or nullptr when the source has no bitmap
    unsigned long long *valid_pop;
    uint32_t *err;            // set when an index is out of bounds
    // second pass
    const uint64_t *block_sums;  // [ceil(n / kStrBlock)] bytes of each block of kStrBlock elements
    const uint64_t *group_base;  // [ceil(blocks / kStrGroup)] exclusive byte prefix of each group of kStrGroup blocks
    uint64_t total_bytes;        // grand total (= out_offsets[n])
    int32_t *out_offsets;        // [n+1]
    uint8_t *out_data;
};
constexpr int kStrBlock = 256;   // elements per workgroup of the copy pass = per entry of block_sums
constexpr int kStrGroup = 256;   // blocks per entry of group_base

static __global__ __launch_bounds__(256) void str_gather_lengths(const StrGather p) {
    const uint64_t j = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    bool valid = false;
    uint32_t len = 0;
    if (j < p.n) {
        const uint64_t idx = p.indices[j];
        if (idx >= p.src_length) {
            atomicExch(p.err, 1u);
        } else {
            const uint64_t e = p.offset + idx;
            valid = !p.validity || ((p.validity[e >> 3] >> (e & 7)) & 1);
            const int32_t b0 = p.offsets[e];
            if (valid) len = static_cast<uint32_t>(p.offsets[e + 1] - b0);
            p.starts[j] = b0;
        }
        p.lengths[j] = len;
    }
    if (p.out_validity) {  // kernel-uniform
        __shared__ uint32_t s_valid[4];
        const uint64_t word = ballot64(valid);  // 64 consecutive j per wave (blockDim is a multiple of 64)
        if ((threadIdx.x & 63) == 0) {
            if (j < p.n) p.out_validity[j >> 6] = word;
            s_valid[threadIdx.x >> 6] = static_cast<uint32_t>(__popcll(word));
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned long long t = static_cast<unsigned long long>(s_valid[0]) + s_valid[1] + s_valid[2] + s_valid[3];
            if (t) striped_add_wg(p.valid_pop, t);
        }
    }
}

// The same two arrays for the survivors of a selection bitmap, in ROW order, without an index list (the filter path).
// A wave owns 64 selection words (4096 rows): lane q fetches word q and its survivor prefix (two coalesced loads).  The
// offsets of the rows are read as full 16-byte-per-lane buffer loads -- a 10 % selection touches nearly every line of the
// offsets array anyway, and a wave-wide load costs the address unit as much as a load with six active lanes.  Chunks of
// 1024 rows: (start, length) of the chunk's survivors are packed in LDS at their rank and leave as coalesced stores
// (sparse per-row stores were the kernel's bound: one store instruction per row slot and array).  The output validity is
// the source bitmap compacted by the same selection (bits_compact_kernel).
struct SelStr {
    const uint64_t *sel;      // selection words (bits past the last row zero)
    uint64_t nwords;
    const uint64_t *excl;     // [nwords + 1] exclusive survivor counts per word, or nullptr with
    const uint64_t *range_offsets;  // output row of the first survivor of every range of range_rows rows (the fused pass's
    uint32_t range_rows;            // wave offsets, FusedParams::wave_offsets); range_rows divides 4096
    unsigned long long *block_sums;  // nullptr, or zeroed [ceil(survivors / kStrBlock)]: += bytes of every block of kStrBlock
                                     // elements (folds the str_block_sums pass into this one)
    uint64_t cap_rows;               // rows `lengths` / `starts` hold: a chunk that would pass it writes nothing
    const int32_t *offsets;
    const uint8_t *validity;  // or nullptr
    uint64_t offset;          // element offset of the source column
    uint64_t length;          // logical length of the source column (= rows of the selection)
    uint32_t *lengths;        // [survivors]
    int32_t *starts;          // [survivors]
};
static __global__ __launch_bounds__(256) void sel_str_lengths(const SelStr p) {
    __shared__ int32_t s_start[4][1024];
    __shared__ uint32_t s_len[4][1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t w0 = (static_cast<uint64_t>(blockIdx.x) * 4 + wave) * 64;
    if (w0 >= p.nwords) return;  // wave-uniform
    const uint64_t wq = w0 + lane < p.nwords ? w0 + lane : p.nwords;
    const uint64_t mine = wq < p.nwords ? p.sel[wq] : 0;
    uint64_t mine_at;  // output row of the word's first survivor; lanes past the last word hold the end of the wave's run
    if (p.excl) {
        mine_at = p.excl[wq];
    } else {  // the wave's 64 words start a range of the fused pass: its offset + the survivors of the words before
        const uint32_t cnt = static_cast<uint32_t>(__popcll(mine));
        mine_at = p.range_offsets[w0 * 64 / p.range_rows] + (wave_scan_u32(cnt) - cnt);  // six DPP adds (round 3: six trips through the LDS crossbar)
    }
    const uint64_t nonzero = ballot64(mine != 0);
    if (nonzero == 0) return;
    // offsets[e0 .. e0 + rows] of the wave's rows through a bounds-checked buffer view (reads past it return 0; only
    // rows without a selection bit can fall there)
    const uint64_t row0 = w0 * 64, left = p.length - row0;
    const uint32_t nbytes = uniform32(static_cast<uint32_t>((left < 4096 ? left : 4096) + 1) * 4u);
    const uint64_t base = uniform64(reinterpret_cast<uint64_t>(p.offsets + p.offset + row0));
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(base), 0, nbytes, 0x00020000);
    // every offset of the wave's 4096 rows is requested up front (16 x 16 bytes + 16 x 4 per lane): round 3 requested a chunk's, waited,
    // staged, wrote and only then requested the next chunk's -- four memory round trips in a row per wave
    rv_u32x4 qa[16];
    uint32_t nxa[16];
#pragma unroll
    for (int cg = 0; cg < 16; ++cg) {  // lane's rows of group cg: 256 cg + 4 lane + {0..3}
        qa[cg] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, cg * 1024, 0);
        nxa[cg] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, lane * 16 + 16, cg * 1024, 0);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {  // 16 words = 1024 rows per chunk
        if (((nonzero >> (16 * c)) & 0xFFFFull) == 0) continue;  // wave-uniform
        const uint64_t P = readlane64(mine_at, 16 * c);  // output position of the chunk's first survivor
        const uint64_t end_at = c < 3 ? readlane64(mine_at, 16 * c + 16)
                                      : readlane64(mine_at, 63) + static_cast<uint64_t>(__popcll(readlane64(mine, 63)));
        const uint32_t cnt = static_cast<uint32_t>(end_at - P);
        const rv_u32x4 *q = &qa[4 * c];
        const uint32_t *nx = &nxa[4 * c];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int word = c * 16 + g * 4 + (lane >> 4);  // the word holding this lane's four rows
            const uint64_t m = shfl64(mine, word);
            const uint32_t at = static_cast<uint32_t>(shfl64(mine_at, word) - P);
            const int bit0 = (lane & 15) * 4;
            const uint32_t four = static_cast<uint32_t>(m >> bit0) & 15u;
            if (four) {
                uint32_t pos = at + static_cast<uint32_t>(__popcll(m & ((1ull << bit0) - 1)));
                const uint32_t b[5] = {q[g].x, q[g].y, q[g].z, q[g].w, nx[g]};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if ((four >> r) & 1) {
                        bool valid = true;
                        if (p.validity) {
                            const uint64_t e = p.offset + row0 + static_cast<uint64_t>(c * 1024 + g * 256 + lane * 4 + r);
                            valid = (p.validity[e >> 3] >> (e & 7)) & 1;
                        }
                        s_start[wave][pos] = static_cast<int32_t>(b[r]);
                        s_len[wave][pos] = valid ? b[r + 1] - b[r] : 0u;
                        ++pos;
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (P + cnt <= p.cap_rows) {  // wave-uniform; beyond it the fused pass has flagged the overflow and the host re-runs
            for (uint32_t k = lane; k < cnt; k += 64) {
                p.starts[P + k] = s_start[wave][k];
                p.lengths[P + k] = s_len[wave][k];
            }
            if (p.block_sums && cnt) {  // bytes per block of kStrBlock output elements: one atomic per (chunk, block), <= 5 blocks
                for (uint64_t blk = P / kStrBlock; blk <= (P + cnt - 1) / kStrBlock; ++blk) {
                    const uint64_t lo = blk * kStrBlock > P ? blk * kStrBlock - P : 0;
                    const uint64_t hi = (blk + 1) * kStrBlock < P + cnt ? (blk + 1) * kStrBlock - P : cnt;
                    uint64_t acc = 0;
                    for (uint64_t k = lo + lane; k < hi; k += 64) acc += s_len[wave][k];
                    acc = wave_sum64(acc);
                    if (lane == 0 && acc) atomicAdd(&p.block_sums[blk], static_cast<unsigned long long>(acc));
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// block_sums (sel_str_lengths' atomics) -> group[g] = exclusive byte prefix of the groups of kStrGroup blocks; *total = all
// bytes, out_offsets[n] = all bytes.  ONE workgroup: str_group_sums + scan_sums_inplace in a single launch.  (Adding the group
// sums with a second atomic in the lengths pass put ~900 adds on each of a few hundred words: 0.21 -> 0.51 ms for that pass.)
static __global__ __launch_bounds__(1024) void str_sums_scan(const unsigned long long *block_sums, uint64_t nblocks, uint64_t *group, unsigned long long *total,
                                                             int32_t *out_offsets, uint64_t n) {
    __shared__ uint64_t s_wave[16];
    __shared__ uint64_t s_carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t ngroups = (nblocks + kStrGroup - 1) / kStrGroup;
    // block_sums == nullptr: `group` already holds the group sums (str_group_sums ran: many workgroups instead of this one --
    // with ~80 k block sums to read, the single workgroup spent 42 us on them)
    for (uint64_t g0 = static_cast<uint64_t>(wave) * 4; block_sums != nullptr && g0 < ngroups; g0 += 64) {  // a wave per four groups
        uint64_t acc[4] = {0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < kStrGroup / 64; ++k) {
                const uint64_t b = (g0 + u) * kStrGroup + static_cast<uint64_t>(k) * 64 + lane;
                if (b < nblocks) acc[u] += block_sums[b];
            }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint64_t t = wave_sum64(acc[u]);
            if (lane == 0 && g0 + u < ngroups) group[g0 + u] = t;
        }
    }
    __threadfence_block();
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (uint64_t base = 0; base < ngroups; base += 1024) {  // exclusive scan in place, 1024 groups per round
        const uint64_t i = base + threadIdx.x;
        const uint64_t mine = i < ngroups ? group[i] : 0;
        uint64_t incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t y = (static_cast<uint64_t>(__shfl_up(static_cast<uint32_t>(incl >> 32), d, 64)) << 32) |
                               __shfl_up(static_cast<uint32_t>(incl), d, 64);
            if (lane >= d) incl += y;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint64_t run = s_carry + incl - mine;
        for (int w = 0; w < wave; ++w) run += s_wave[w];
        if (i < ngroups) group[i] = run;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = run + mine;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        *total = s_carry;
        if (out_offsets) out_offsets[n] = static_cast<int32_t>(s_carry);  // (nullptr: the tile-order copy writes it)
    }
}

// sums[b] = bytes of elements [b * kStrBlock, (b + 1) * kStrBlock): one wave per block, 16 lengths per lane
static __global__ __launch_bounds__(256) void str_block_sums(const uint32_t *lengths, uint64_t n, uint64_t *sums) {
    const int lane = threadIdx.x & 63;
    const uint64_t b = static_cast<uint64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (b * kStrBlock >= n) return;  // wave-uniform
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < kStrBlock / 256; ++k) {
        const uint64_t first = b * kStrBlock + static_cast<uint64_t>(k) * 256 + static_cast<uint64_t>(lane) * 4;
        if (first + 4 <= n) {
            const uint4 q = *reinterpret_cast<const uint4 *>(lengths + first);  // pool blocks are 256-byte aligned
            acc += static_cast<uint64_t>(q.x) + q.y + q.z + q.w;
        } else {
            for (uint64_t i = first; i < n; ++i) acc += lengths[i];
        }
    }
    acc = wave_sum64(acc);
    if (lane == 0) sums[b] = acc;
}

// group[g] = sum of sums[g * kStrGroup .. ): one wave per group.  The scan then runs over blocks / 256 entries (one round
// of the single-workgroup scan whatever the size) and a copy workgroup adds the sums of the blocks before it in its group.
static __global__ __launch_bounds__(256) void str_group_sums(const uint64_t *sums, uint64_t nblocks, uint64_t *group) {
    const int lane = threadIdx.x & 63;
    const uint64_t g = static_cast<uint64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (g * kStrGroup >= nblocks) return;  // wave-uniform
    uint64_t acc = 0;
    for (int k = lane; k < kStrGroup; k += 64) {
        const uint64_t b = g * kStrGroup + static_cast<uint64_t>(k);
        if (b < nblocks) acc += sums[b];
    }
    acc = wave_sum64(acc);
    if (lane == 0) group[g] = acc;
}

// The low m (<= 8) bytes of val to byte `at` of a ZEROED window: OR-ed into the two aligned 8-byte words they touch (LDS atomics).
// No branches and no partial stores: what lies behind an element's last byte belongs to the next element, which another lane is
// writing in the same instruction -- round 4's first form stored a last chunk as 4 + 2 + 1 byte pieces under three divergent
// branches, and the kernel took the time of its instruction issue (800 VALU + 570 SALU instructions per 512 rows).
__device__ __forceinline__ void win_or(unsigned long long *win_words, uint32_t at, uint64_t val, uint32_t m) {
    if (m == 0) return;  // (not a formality: rows that do not survive all sit at the NEXT survivor's position, and atomics of many lanes
                         // on one word are served one by one -- at 10 % selectivity the tile copy took twice its time at 84 %)
    const uint64_t keep = m >= 8 ? ~0ull : ((1ull << (8 * m)) - 1);
    val &= keep;
    const uint32_t s = (at & 7) * 8;
    const uint64_t lo = val << s, hi = (val >> 1) >> (63 - s);  // (s == 0: nothing spills into the next word)
    atomicOr(&win_words[at >> 3], static_cast<unsigned long long>(lo));
    atomicOr(&win_words[(at >> 3) + 1], static_cast<unsigned long long>(hi));
}
// nwords8 8-byte words of zeros (+ the one behind them that a last element's spill may touch), 16 bytes per lane and store
__device__ __forceinline__ void win_zero(unsigned long long *win_words, uint32_t words, int lane) {
    for (uint32_t k = 2 * lane; k < words + 1; k += 128) {
        win_words[k] = 0;
        win_words[k + 1] = 0;
    }
}
// Window bytes [lead, nb) to the output at first_byte - lead (8-byte aligned): whole words by every lane, the partial first and
// last word -- shared with the neighbouring run -- byte by byte by lanes 0..7 and 8..15 (one predicated store each, no loops)
__device__ __forceinline__ void win_out(const unsigned long long *win_words, uint8_t *out_aligned, uint32_t lead, uint32_t nb, int lane) {
    const uint32_t nw = (nb + 7) >> 3;
    const bool head = lead != 0, tail = (nb & 7) != 0 && !(head && nw == 1);
    unsigned long long *out_words = reinterpret_cast<unsigned long long *>(out_aligned);
    for (uint32_t k = (head ? 1u : 0u) + lane; k + (tail ? 1u : 0u) < nw; k += 64) out_words[k] = win_words[k];
    if (head && lane < 8 && static_cast<uint32_t>(lane) >= lead && static_cast<uint32_t>(lane) < nb)
        out_aligned[lane] = static_cast<uint8_t>(win_words[0] >> (8 * lane));
    if (tail && lane >= 8 && lane < 16 && static_cast<uint32_t>(lane - 8) < (nb & 7))
        out_aligned[8ull * (nw - 1) + (lane - 8)] = static_cast<uint8_t>(win_words[nw - 1] >> (8 * (lane - 8)));
}
// the same straight to global memory (a run that does not fit the window)
__device__ __forceinline__ void copy_direct(uint8_t *dst, const uint8_t *src, uint32_t len) {
    struct __attribute__((packed)) U64 { uint64_t v; };
    struct __attribute__((packed)) U32 { uint32_t v; };
    struct __attribute__((packed)) U16 { uint16_t v; };
    uint32_t done = 0;
    for (; done + 8 <= len; done += 8) reinterpret_cast<U64 *>(dst + done)->v = reinterpret_cast<const U64 *>(src + done)->v;
    if (done < len) {
        uint64_t val = reinterpret_cast<const U64 *>(src + done)->v;  // <= 7 bytes past the element: inside the buffer's padding
        const uint32_t m = len - done;
        if (m & 4) {
            reinterpret_cast<U32 *>(dst + done)->v = static_cast<uint32_t>(val);
            val >>= 32, done += 4;
        }
        if (m & 2) {
            reinterpret_cast<U16 *>(dst + done)->v = static_cast<uint16_t>(val);
            val >>= 16, done += 2;
        }
        if (m & 1) dst[done] = static_cast<uint8_t>(val);
    }
}

// ONE WAVE per block of kStrBlock = 256 elements; lane l holds elements l, l + 64, l + 128, l + 192 of the block (neighbouring lanes
// load neighbouring lengths, starts and chunks and write neighbouring window bytes).  The 256 elements land next to each other in
// the output, so the wave assembles that run in a zeroed LDS window (win_or) and writes it as aligned 8-byte words (the run's first /
// last partial word byte by byte: win_out).  A run longer than the window (long strings) goes straight to the output (copy_direct).
// Round 3's kernel ran one element per lane in 256-thread workgroups: a block's life was a chain of four to six dependent memory
// round trips (length -> block sums -> start -> bytes, a trip per further chunk) with ONE load in flight per lane -- 0.33 ms per 2e7
// gathered elements and 1.9 ms per 1.7e8.  Round 4: the block sums are requested before the lengths and the first TWO chunks of
// every element together -- three round trips for strings up to 16 bytes (1.21 -> 0.96 ms per 1.7e8 elements: 4.9 GB of traffic at
// 5.1 TB/s, a third of it the lists of starts and lengths -- dense selections take the source-tile path below instead); elements
// interleaved across the lanes and the OR-composed window: 0.36 -> 0.30 ms per 2e7.  A 4 KiB window (32 resident blocks per CU
// instead of 18) changed nothing.
constexpr int kStrPerLane = kStrBlock / 64;
template <uint32_t kStrWindow>
static __global__ __launch_bounds__(64) void str_gather_copy(const StrGather p) {
    __shared__ unsigned long long win[kStrWindow / 8 + 2];
    struct __attribute__((packed)) U64 { uint64_t v; };
    const int lane = threadIdx.x;
    const uint64_t j0 = static_cast<uint64_t>(blockIdx.x) * kStrBlock + static_cast<uint64_t>(lane);  // lane l holds elements j0 + 64 i:
    if (static_cast<uint64_t>(blockIdx.x) * kStrBlock >= p.n) return;                                 // neighbouring lanes, neighbouring elements
    // bytes of the blocks before this one in its group: four block sums per lane, requested BEFORE the lengths and starts so that all
    // of them come back in one round trip (they were a trip of their own behind the first wait)
    const uint32_t in_group = blockIdx.x % kStrGroup;
    // (a vector load by lane 0, in the same queue as the loads below: as a scalar load it was issued once the lengths were back)
    const uint64_t group_base_v = lane == 0 ? __builtin_nontemporal_load(p.group_base + blockIdx.x / kStrGroup) : 0;
    uint32_t prior_of[kStrPerLane];
#pragma unroll
    for (int i = 0; i < kStrPerLane; ++i) {
        const uint32_t b = static_cast<uint32_t>(lane) * kStrPerLane + i;
        prior_of[i] = b < in_group ? static_cast<uint32_t>(p.block_sums[blockIdx.x - in_group + b]) : 0u;
    }
    uint32_t len[kStrPerLane];
    int32_t st[kStrPerLane];
#pragma unroll
    for (int i = 0; i < kStrPerLane; ++i) {
        const bool in = j0 + 64 * i < p.n;
        len[i] = in ? p.lengths[j0 + 64 * i] : 0;
        st[i] = in ? p.starts[j0 + 64 * i] : 0;
    }
    // the element's output byte = the block's base (scanned block sums) + an exclusive scan of the lengths inside the block.
    // 32-bit DPP adds: a StringArray's bytes are indexed by int32 offsets (string.rs:9-15), so every partial sum fits
    uint32_t rel[kStrPerLane];  // output byte of every element, relative to run0
    uint32_t run_bytes = 0;
#pragma unroll
    for (int i = 0; i < kStrPerLane; ++i) {
        const uint32_t incl = wave_scan_u32(len[i]);
        rel[i] = run_bytes + incl - len[i];
        run_bytes += static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(incl), 63));
    }
    const uint32_t prior = prior_of[0] + prior_of[1] + prior_of[2] + prior_of[3];
    const uint64_t run0 = readlane64(group_base_v, 0) + wave_sum_u32(prior);  // wave-uniform: first output byte of the block
#pragma unroll
    for (int i = 0; i < kStrPerLane; ++i)
        if (j0 + 64 * i < p.n) p.out_offsets[j0 + 64 * i] = static_cast<int32_t>(run0 + rel[i]);
    if (p.total_bytes != ~0ull && lane == 0 && static_cast<uint64_t>(blockIdx.x) * kStrBlock + kStrBlock >= p.n) p.out_offsets[p.n] = static_cast<int32_t>(p.total_bytes);  // ~0: str_sums_scan wrote it
    // the window starts at the 8-byte boundary below run0, so LDS word k == output word (run0 >> 3) + k
    const uint32_t lead = static_cast<uint32_t>(run0 & 7);
    if (run_bytes + lead > kStrWindow) {  // wave-uniform
        // (not through LDS: straight to the output in unaligned 8-byte pieces, the tail as 4 + 2 + 1 -- an outlier block among shorter
        // ones, or a column whose strings grow along its rows, must not fall off a cliff)
#pragma unroll
        for (int i = 0; i < kStrPerLane; ++i) copy_direct(p.out_data + run0 + rel[i], p.data + st[i], len[i]);
        return;
    }
    // The elements' bytes, eight at a time: an UNALIGNED 8-byte load, OR-ed into the zeroed window at the element's byte position
    // (win_or).  The first TWO chunks of all four elements are requested together (strings up to 16 bytes cost one memory round trip
    // per block, not one per chunk: the block's life is a chain of round trips, see above); longer ones loop from there.  A last,
    // partial chunk is loaded whole -- at most 7 bytes past the element: inside any data buffer (every String data buffer of the
    // library is a pool block with >= 8 bytes of padding; rv_wrap refuses Strings) -- and masked.
    uint64_t first[kStrPerLane], second[kStrPerLane];
    bool more8 = false, more16 = false;
#pragma unroll
    for (int i = 0; i < kStrPerLane; ++i) {
        first[i] = len[i] ? reinterpret_cast<const U64 *>(p.data + st[i])->v : 0;
        second[i] = len[i] > 8 ? reinterpret_cast<const U64 *>(p.data + st[i] + 8)->v : 0;
        more8 |= len[i] > 8;
        more16 |= len[i] > 16;
    }
    win_zero(win, (run_bytes + lead + 7) >> 3, lane);  // (only what the run covers, while the chunks are on their way)
#pragma unroll
    for (int i = 0; i < kStrPerLane; ++i) win_or(win, rel[i] + lead, first[i], len[i]);
    if (ballot64(more8) != 0) {
#pragma unroll
        for (int i = 0; i < kStrPerLane; ++i) win_or(win, rel[i] + lead + 8, second[i], len[i] > 8 ? len[i] - 8 : 0u);
    }
    if (ballot64(more16) != 0) {
#pragma unroll
        for (int i = 0; i < kStrPerLane; ++i)
            for (uint32_t done = 16; done < len[i]; done += 8)
                win_or(win, rel[i] + lead + done, reinterpret_cast<const U64 *>(p.data + st[i] + done)->v, len[i] - done);
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the LDS bytes of every lane are in place
    win_out(win, p.out_data + (run0 - lead), lead, run_bytes + lead, lane);
}

// ---- filter() of a StringArray in SOURCE-TILE order (dense selections) --------------------------------------------------------
// When most rows survive, the (start, length) lists of sel_str_lengths are the query's largest intermediate: 8 bytes written and
// read again per survivor (1.34 GB each way for 1.7e8 survivors) next to 1.6 GB of string bytes.  These two kernels go without:
//   sel_str_tile_sums   per tile of kStrTile source rows, the bytes of its survivors (selection words + offsets only);
//   (str_group_sums / str_sums_scan turn the tile sums into byte prefixes, as for the block sums of the other path)
//   sel_str_tile_copy   one wave per tile: selection words, offsets, the survivors' bytes and their output offsets in one go,
//                       the tile's output rows at the fused pass's wave offsets, its output bytes at the tile's byte prefix.
// Everything is sized by SOURCE rows, so all four launches are queued right behind the fused pass: the host never waits for the
// survivor count in between (the block-order path sizes its copy launch by it).
constexpr int kStrTile = 512;  // source rows per tile: eight selection words, two groups of 256 rows (four consecutive rows per lane)
struct SelStrTiles {
    const uint64_t *sel;      // selection words (bits past the last row zero)
    uint64_t nwords;
    const uint64_t *range_offsets;  // the fused pass's wave offsets: output row of the first survivor of every range_rows rows
    uint32_t range_rows;            // divides 4096
    uint32_t pad;
    const int32_t *offsets;
    const uint8_t *validity;  // or nullptr
    uint64_t validity_bytes;
    uint64_t offset;          // element offset of the source column
    uint64_t length;          // logical length of the source column (= rows of the selection)
    const uint8_t *data;
    unsigned long long *tile_sums;  // [ceil(length / kStrTile)] bytes of every tile's survivors (a null survivor has none)
    const uint64_t *group_base;     // [ceil(tiles / kStrGroup)] exclusive byte prefix of each group of kStrGroup tiles
    uint64_t cap_rows;              // rows out_offsets holds: a tile that would pass it writes nothing (the pass flagged the overflow)
    int32_t *out_offsets;           // [cap_rows + 1]
    uint8_t *out_data;
};
// validity bits of elements e .. e + 3 (four consecutive rows of a lane)
__device__ __forceinline__ uint32_t str_valid4(const uint8_t *validity, uint64_t validity_bytes, uint64_t e) {
    if (!validity) return 15u;
    const uint64_t by = e >> 3;
    const uint32_t sh = static_cast<uint32_t>(e & 7);
    uint32_t v = by < validity_bytes ? validity[by] : 0u;
    if (sh > 4 && by + 1 < validity_bytes) v |= static_cast<uint32_t>(validity[by + 1]) << 8;
    return (v >> sh) & 15u;
}
// One wave per 4096 rows = 8 tiles (64 selection words, a word per lane); every offset of the rows is requested up front.
static __global__ __launch_bounds__(256) void sel_str_tile_sums(const SelStrTiles p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t w0 = (static_cast<uint64_t>(blockIdx.x) * 4 + wave) * 64;
    if (w0 >= p.nwords) return;  // wave-uniform
    const uint64_t mine = w0 + lane < p.nwords ? p.sel[w0 + lane] : 0;
    const uint64_t ntiles = (p.nwords + 7) / 8, tile0 = w0 / 8;
    const uint64_t nonzero = ballot64(mine != 0);
    const uint64_t row0 = w0 * 64, left = p.length - row0;
    const uint32_t nbytes = uniform32(static_cast<uint32_t>((left < 4096 ? left : 4096) + 1) * 4u);
    const uint64_t base = uniform64(reinterpret_cast<uint64_t>(p.offsets + p.offset + row0));
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(base), 0, nbytes, 0x00020000);
    rv_u32x4 qa[16];
    uint32_t nxa[16];
    if (nonzero != 0) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {  // lane's rows of group g: 256 g + 4 lane + {0..3}
            qa[g] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, g * 1024, 0);
            nxa[g] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, lane * 16 + 16, g * 1024, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        if (tile0 + t >= ntiles) break;  // wave-uniform
        uint32_t acc = 0;
        if (((nonzero >> (8 * t)) & 0xFFull) != 0) {  // wave-uniform
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int g = 2 * t + h;
                const uint64_t m = shfl64(mine, g * 4 + (lane >> 4));  // the word holding this lane's four rows
                uint32_t four = static_cast<uint32_t>(m >> ((lane & 15) * 4)) & 15u;
                if (four) four &= str_valid4(p.validity, p.validity_bytes, p.offset + row0 + static_cast<uint64_t>(g * 256 + lane * 4));
                const uint32_t b[5] = {qa[g].x, qa[g].y, qa[g].z, qa[g].w, nxa[g]};
#pragma unroll
                for (int r = 0; r < 4; ++r) acc += ((four >> r) & 1) ? b[r + 1] - b[r] : 0u;
            }
            acc = wave_sum_u32(acc);
        }
        if (lane == 0) p.tile_sums[tile0 + t] = acc;
    }
}

// One wave per tile.  The tile's eight selection words are eight row sets of 64 rows, a row per lane: lane l of set j holds row
// 64 j + l, its offset and the next (two coalesced dword loads), its selection bit (bit l of word j: the word is wave-uniform), and
// copies it if it survives: the first two chunks of all eight rows are requested together, assembled in the LDS window at the
// survivors' byte positions inside the tile's run (a wave scan of the selected lengths per row set), and the run leaves as aligned
// 8-byte words.  Neighbouring lanes hold neighbouring rows, so their
// chunk loads, their window stores (a stride of one element: few LDS bank conflicts -- with four consecutive rows per lane the
// stride was ~32 bytes, eight banks for 64 lanes, and the kernel took the time of its window stores) and their output offsets
// (consecutive ranks) are as coalesced as the selection allows.
// What a tile's wave requests before anything depends on anything (one round trip).
struct TileFront {
    uint64_t mine;          // lane k: selection word range_first + k
    uint64_t group_base_v;  // lane 0: byte prefix of the tile's group of tiles
    uint64_t range_at;      // output row of the first survivor of the tile's range
    uint32_t prior_of[4];   // bytes of tiles before this one in its group (four per lane)
    uint32_t b0[kStrTile / 64], b1[kStrTile / 64];  // offsets of row 64 j + lane and of the next
    uint32_t before;        // selection words of the range in front of the tile: 0 .. 56
};
__device__ __forceinline__ void tile_fetch(const SelStrTiles &p, uint64_t tile, int lane, TileFront &f) {
    constexpr int SETS = kStrTile / 64;
    const uint64_t w0 = tile * SETS, row0 = tile * kStrTile;
    const uint32_t in_group = static_cast<uint32_t>(tile % kStrGroup);
    f.group_base_v = lane == 0 ? __builtin_nontemporal_load(p.group_base + tile / kStrGroup) : 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t b = static_cast<uint32_t>(lane) * 4 + i;
        f.prior_of[i] = b < in_group ? static_cast<uint32_t>(p.tile_sums[tile - in_group + b]) : 0u;
    }
    // the selection words from the start of the tile's range of the fused pass (its wave offset is the output row of the range's
    // first survivor) to the end of the tile
    const uint32_t range_words = p.range_rows / 64;
    const uint64_t range_first = range_words > SETS ? (w0 / range_words) * range_words : w0;
    f.before = static_cast<uint32_t>(w0 - range_first);
    const uint64_t wq = range_first + lane;
    f.mine = (static_cast<uint32_t>(lane) < f.before + SETS && wq < p.nwords) ? p.sel[wq] : 0;
    f.range_at = p.range_offsets[row0 / p.range_rows];
    // offsets of the tile's rows through a bounds-checked view (reads past it return 0: only rows without a selection bit fall there)
    const uint64_t left = p.length - row0;
    const uint32_t nbytes = uniform32(static_cast<uint32_t>((left < kStrTile ? left : kStrTile) + 1) * 4u);
    const uint64_t base = uniform64(reinterpret_cast<uint64_t>(p.offsets + p.offset + row0));
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(base), 0, nbytes, 0x00020000);
#pragma unroll
    for (int j = 0; j < SETS; ++j) {
        f.b0[j] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, lane * 4, j * 256, 0);
        f.b1[j] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, lane * 4 + 4, j * 256, 0);
    }
}
template <uint32_t kWindow>
__device__ __forceinline__ void tile_copy(const SelStrTiles &p, uint64_t tile, int lane, const TileFront &f, unsigned long long *win) {
    struct __attribute__((packed)) U64 { uint64_t v; };
    constexpr int SETS = kStrTile / 64;
    const uint64_t row0 = tile * kStrTile;
    const uint32_t before = f.before;
    const uint32_t cntw = static_cast<uint32_t>(__popcll(f.mine));
    const uint32_t incl_w = wave_scan_u32(cntw), excl_w = incl_w - cntw;
    const uint32_t in_front = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(excl_w), static_cast<int>(before)));  // survivors of the range before the tile
    const uint32_t cnt = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(incl_w), static_cast<int>(before + SETS - 1))) - in_front;
    if (cnt == 0) return;  // wave-uniform: nothing of this tile survives (its sum is 0)
    const uint64_t P = f.range_at + in_front;  // output row of the tile's first survivor
    uint32_t len[SETS], rank[SETS];
    bool taken[SETS];
#pragma unroll
    for (int j = 0; j < SETS; ++j) {
        const uint64_t m = readlane64(f.mine, static_cast<int>(before) + j);  // wave-uniform
        taken[j] = (m >> lane) & 1;
        rank[j] = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(excl_w), static_cast<int>(before) + j)) - in_front +
                  __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
        bool live = taken[j];
        if (live && p.validity) {
            const uint64_t e = p.offset + row0 + static_cast<uint64_t>(j * 64 + lane);
            live = (p.validity[e >> 3] >> (e & 7)) & 1;
        }
        len[j] = live ? f.b1[j] - f.b0[j] : 0u;
    }
    uint32_t rel[SETS];  // output byte of every row's element, relative to the tile's first
    uint32_t run_bytes = 0;
#pragma unroll
    for (int j = 0; j < SETS; ++j) {
        const uint32_t incl = wave_scan_u32(len[j]);
        rel[j] = run_bytes + incl - len[j];
        run_bytes += static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(incl), 63));
    }
    const uint32_t prior = f.prior_of[0] + f.prior_of[1] + f.prior_of[2] + f.prior_of[3];
    const uint64_t run0 = readlane64(f.group_base_v, 0) + wave_sum_u32(prior);  // first output byte of the tile
    if (P + cnt > p.cap_rows) return;  // wave-uniform; beyond it the fused pass has flagged the overflow and the host re-runs
    // out_offsets[P + cnt] = the byte behind the tile's run: the next tile with survivors writes the same value there, the last
    // one's is the column's offsets[rows]
#pragma unroll
    for (int j = 0; j < SETS; ++j)
        if (taken[j]) p.out_offsets[P + rank[j]] = static_cast<int32_t>(run0 + rel[j]);
    if (lane == 0) p.out_offsets[P + cnt] = static_cast<int32_t>(run0 + run_bytes);
    const uint32_t lead = static_cast<uint32_t>(run0 & 7);  // the window starts at the 8-byte boundary below the tile's first byte
    if (run_bytes + lead > kWindow) {  // wave-uniform: an outlier of long strings goes straight to the output
        // (one copy of the loop body, the row sets rotated through slot 0: unrolled eight times these rare paths cost the kernel 65
        // registers)
        uint32_t at[SETS], from[SETS], left[SETS];
#pragma unroll
        for (int j = 0; j < SETS; ++j) at[j] = rel[j], from[j] = f.b0[j], left[j] = len[j];
#pragma unroll 1
        for (int it = 0; it < SETS; ++it) {
            copy_direct(p.out_data + run0 + at[0], p.data + from[0], left[0]);
#pragma unroll
            for (int j = 0; j + 1 < SETS; ++j) at[j] = at[j + 1], from[j] = from[j + 1], left[j] = left[j + 1];
        }
        return;
    }
    // the first two chunks of every surviving element (strings up to 16 bytes: one round trip), through a 32-bit offset on the
    // (wave-uniform) data pointer
    uint64_t first[SETS], second[SETS];
#pragma unroll
    for (int j = 0; j < SETS; ++j) {
        first[j] = len[j] ? reinterpret_cast<const U64 *>(p.data + f.b0[j])->v : 0;
        second[j] = len[j] > 8 ? reinterpret_cast<const U64 *>(p.data + f.b0[j] + 8)->v : 0;
    }
    win_zero(win, (run_bytes + lead + 7) >> 3, lane);
    bool more16 = false, more8 = false;
#pragma unroll
    for (int j = 0; j < SETS; ++j) {
        win_or(win, rel[j] + lead, first[j], len[j]);
        more8 |= len[j] > 8;
        more16 |= len[j] > 16;
    }
    if (ballot64(more8) != 0) {
#pragma unroll
        for (int j = 0; j < SETS; ++j) win_or(win, rel[j] + lead + 8, second[j], len[j] > 8 ? len[j] - 8 : 0u);
    }
    if (ballot64(more16) != 0) {  // strings past 16 bytes: their further chunks, a round trip each
        uint32_t at[SETS], from[SETS], left[SETS];
#pragma unroll
        for (int j = 0; j < SETS; ++j) at[j] = rel[j] + lead, from[j] = f.b0[j], left[j] = len[j];
#pragma unroll 1
        for (int it = 0; it < SETS; ++it) {
            for (uint32_t done = 16; done < left[0]; done += 8)
                win_or(win, at[0] + done, reinterpret_cast<const U64 *>(p.data + from[0] + done)->v, left[0] - done);
#pragma unroll
            for (int j = 0; j + 1 < SETS; ++j) at[j] = at[j + 1], from[j] = from[j + 1], left[j] = left[j + 1];
        }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the LDS bytes of every lane are in place
    win_out(win, p.out_data + (run0 - lead), lead, run_bytes + lead, lane);
    __builtin_amdgcn_wave_barrier();  // (LDS instructions of one wave execute in order: the next tile's zeros follow these reads)
}
template <uint32_t kWindow>
static __global__ __launch_bounds__(64) void sel_str_tile_copy(const SelStrTiles p) {
    __shared__ unsigned long long win[kWindow / 8 + 2];
    const int lane = threadIdx.x;
    // One tile per wave.  (Measured and dropped: waves that loop over tiles and request the next tile's words and offsets before they
    // copy the current one -- 143 VGPRs, or 128 with ten spilled; 1.14 ms against 1.01 ms per 2e8 rows.)
    TileFront f;
    tile_fetch(p, blockIdx.x, lane, f);
    tile_copy<kWindow>(p, blockIdx.x, lane, f, win);
}

// ---- `StringColumn <op> Literal` -> truth bitmap (plan.rs:112-130 with series.rs:87-117 for String cells) --------
struct StrCompare {
    const int32_t *offsets;
    const uint8_t *data;
    const uint8_t *validity;  // or nullptr
    uint64_t offset, n;
    const uint8_t *lit;       // literal bytes (device)
    uint32_t lit_len;
    int32_t op;               // rv_cmp; -1: constant truth for valid cells (null / cross-type literal)
    int32_t const_v, null_v;  // truth of a valid cell when op == -1; truth of a null cell
    uint64_t *out_words;      // ceil(n/64) words, bits past n zero
};
static __global__ __launch_bounds__(256) void str_compare_mask(const StrCompare p) {
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    bool r = false;
    if (i < p.n) {
        const uint64_t e = p.offset + i;
        const bool valid = !p.validity || ((p.validity[e >> 3] >> (e & 7)) & 1);
        if (!valid) r = p.null_v != 0;
        else if (p.op < 0) r = p.const_v != 0;
        else {
            const uint8_t *a = p.data + p.offsets[e];
            const uint32_t la = static_cast<uint32_t>(p.offsets[e + 1] - p.offsets[e]), m = la < p.lit_len ? la : p.lit_len;
            int ord = 0;  // byte-wise lexicographic, shorter prefix first (str::cmp)
            for (uint32_t k = 0; k < m && ord == 0; ++k) ord = static_cast<int>(a[k]) - static_cast<int>(p.lit[k]);
            if (ord == 0) ord = la < p.lit_len ? -1 : (la > p.lit_len ? 1 : 0);
            switch (p.op) {
                case 0: r = ord == 0; break;  // RV_EQ
                case 1: r = ord != 0; break;  // RV_NE
                case 2: r = ord < 0; break;   // RV_LT
                case 3: r = ord > 0; break;   // RV_GT
                case 4: r = ord <= 0; break;  // RV_LE
                default: r = ord >= 0; break; // RV_GE
            }
        }
    }
    const uint64_t word = ballot64(r);
    if ((threadIdx.x & 63) == 0 && i < p.n) p.out_words[i >> 6] = word;
}

// ---- AND of several Boolean-column terms -> one truth bitmap ------------------------------------------
// The fused pass reads at most kMaxBoolCols Boolean predicate columns; a predicate over more of them (String
// compares count: each becomes a truth bitmap) is folded here first: one 64-row word per lane.
struct BoolFold {
    DevCol cols[kMaxTerms];
    DevTerm terms[kMaxTerms];  // lowered Boolean terms; terms[t] reads cols[t]
    int32_t nterms;
    int32_t pad;
    uint64_t n;
    uint64_t *out_words;  // ceil(n/64) words, bits past n zero
};
static __global__ __launch_bounds__(256) void bool_fold_kernel(const BoolFold p) {
    const uint64_t w = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (w * 64 >= p.n) return;
    uint64_t acc = p.n - w * 64 >= 64 ? ~0ull : low_mask(p.n - w * 64);
    for (int t = 0; t < p.nterms; ++t) {
        const DevCol c = p.cols[t];
        const uint64_t pos = c.offset + w * 64;
        const uint64_t V = load_bits64(static_cast<const uint8_t *>(c.values), pos, c.values_bytes);
        const uint64_t M = c.validity ? load_bits64(c.validity, pos, c.validity_bytes) : ~0ull;
        acc &= eval_bool_word(bool_coef(p.terms[t]), V, M);
    }
    p.out_words[w] = acc;
}

// ---- concat (record_batch.rs:277-342, string branch) ----------------------------------------------
struct StrPart {
    const int32_t *offsets;
    const uint8_t *data;
    const uint8_t *validity;  // or nullptr
    uint64_t offset;          // element offset
    uint64_t length;
};
// ranges[2p], ranges[2p+1] = first / one-past-last byte of part p's logical content
static __global__ void str_part_ranges(const StrPart *parts, uint32_t nparts, int64_t *ranges) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nparts) return;
    ranges[2 * p] = parts[p].offsets[parts[p].offset];
    ranges[2 * p + 1] = parts[p].offsets[parts[p].offset + parts[p].length];
}
struct StrConcat {
    const StrPart *parts;
    const uint64_t *part_start;  // [nparts+1] first output row of each part
    const int64_t *byte_start;   // [nparts+1] first output byte of each part
    const int64_t *ranges;       // from str_part_ranges
    uint32_t nparts;
    uint64_t n;
    int32_t *out_offsets;        // [n+1]
    uint64_t *out_validity;      // words or nullptr
    unsigned long long *valid_pop;
};
static __global__ __launch_bounds__(256) void str_concat_offsets(const StrConcat c) {
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    bool valid = false;
    if (i < c.n) {
        uint32_t lo = 0, hi = c.nparts;  // last part with part_start <= i
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (c.part_start[mid] <= i) lo = mid;
            else hi = mid;
        }
        const StrPart part = c.parts[lo];
        const uint64_t e = part.offset + (i - c.part_start[lo]);
        c.out_offsets[i] = static_cast<int32_t>(c.byte_start[lo] + (part.offsets[e] - c.ranges[2 * lo]));
        if (i + 1 == c.n) c.out_offsets[c.n] = static_cast<int32_t>(c.byte_start[c.nparts]);
        valid = !part.validity || ((part.validity[e >> 3] >> (e & 7)) & 1);
    }
    if (c.out_validity) {  // kernel-uniform
        __shared__ uint32_t s_valid[4];
        const uint64_t word = ballot64(valid);
        if ((threadIdx.x & 63) == 0) {
            if (i < c.n) c.out_validity[i >> 6] = word;
            s_valid[threadIdx.x >> 6] = static_cast<uint32_t>(__popcll(word));
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned long long t = static_cast<unsigned long long>(s_valid[0]) + s_valid[1] + s_valid[2] + s_valid[3];
            if (t) striped_add_wg(c.valid_pop, t);
        }
    }
}

}  // namespace rvk
