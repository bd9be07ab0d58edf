// Remaining gfx950 kernels of the backend: synthetic generator, compare -> nullable
// BooleanArray, BooleanArray logic (K3), bit-range copy, index gather (take), concat,
// and the masked SUM/COUNT reduction (K4).  All are HBM-bound streaming kernels: one
// lane per row (a wave covers one 64-row bitmap word), grid-stride over 64-row chunks.
#pragma once

#include "device_common.hpp"

namespace rvk {

__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

struct GenParams {
    uint64_t *values;    // int64/double elements or bit words
    uint64_t *validity;  // words, or nullptr
    uint64_t seed, first_row, length, modulus, validity_seed;
    uint32_t true_percent, null_percent;
    int32_t dtype;
    uint32_t pattern;          // rv_synth_pattern: 0 independent rows, 1 runs of run_rows equal cells, 2 / 3 ascending / descending
    uint64_t run_rows;         // pattern 1
    uint64_t step, table_rows; // patterns 2, 3: (2^64 - 1) / table_rows, table_rows
};

// rv_generate: bit-identical to orc_generate (the patterns of rivulus_gpu.h, rv_synth_spec)
static __global__ __launch_bounds__(256) void generate_kernel(const GenParams g) {
    const int lane = lane_id();
    const uint64_t nchunks = (g.length + 63) / 64;
    const uint64_t wave0 = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = (static_cast<uint64_t>(gridDim.x) * blockDim.x) >> 6;
    const bool sorted = g.pattern >= 2;
    for (uint64_t c = wave0; c < nchunks; c += nwaves) {
        const uint64_t i = c * 64 + lane;
        const bool in = i < g.length;
        const uint64_t row = g.first_row + i;
        uint64_t h;
        if (g.pattern == 0) h = splitmix64(g.seed + row);
        else if (g.pattern == 1) h = splitmix64(g.seed + row / g.run_rows);
        else h = (g.pattern == 2 ? row : g.table_rows - 1 - row) * g.step;  // a fraction of 2^64 that follows the row index
        if (g.dtype == DT_INT64) {
            if (in) g.values[i] = sorted ? __umul64hi(h, g.modulus) : h % g.modulus;
        } else if (g.dtype == DT_FLOAT64) {
            if (in) reinterpret_cast<double *>(g.values)[i] = static_cast<double>(h >> 11) * 0x1.0p-53;
        } else {
            const uint64_t pct = sorted ? __umul64hi(h, 100ull) : h % 100;
            const uint64_t m = ballot64(in && (pct < g.true_percent));
            if (lane == 0) g.values[c] = m;
        }
        if (g.validity) {
            const uint64_t m = ballot64(in && (splitmix64(g.validity_seed + row) % 100 >= g.null_percent));
            if (lane == 0) g.validity[c] = m;
        }
    }
}

// rv_compare: one term -> BooleanArray {values, validity}.  values bit = valid && cmp
// (value bit under a null is false, boolean.rs:275-278); validity = input validity re-based.
struct CompareParams {
    DevCol col;
    DevTerm term;      // null_v unused
    uint64_t *out_values;
    uint64_t *out_validity;  // nullptr when the input has no validity
    unsigned long long *out_valid_pop;
    uint64_t n;
};
static __global__ __launch_bounds__(256) void compare_kernel(const CompareParams p) {
    const int lane = lane_id();
    const uint64_t nchunks = (p.n + 63) / 64;
    const uint64_t wave0 = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = (static_cast<uint64_t>(gridDim.x) * blockDim.x) >> 6;
    unsigned long long pop = 0;
    for (uint64_t c = wave0; c < nchunks; c += nwaves) {
        const uint64_t i = c * 64 + lane;
        const uint64_t tail = low_mask(p.n - c * 64);
        uint64_t M = p.col.validity ? load_bits64(p.col.validity, p.col.offset + c * 64, p.col.validity_bytes) : ~0ull;
        M &= tail;
        uint64_t res;
        if (p.col.dtype == DT_BOOLEAN) {
            const uint64_t V = load_bits64(static_cast<const uint8_t *>(p.col.values), p.col.offset + c * 64, p.col.values_bytes);
            res = eval_bool_word(p.term, V, M) & M;  // null_v is 0 for compare terms
        } else {
            const uint64_t bits = i < p.n ? static_cast<const uint64_t *>(p.col.values)[p.col.offset + i] : 0;
            const bool r = eval_value_cell(p.term.code(), p.term.lit, p.term.const_v(), bits);
            res = ballot64(r) & M;
        }
        if (lane == 0) {
            p.out_values[c] = res;
            if (p.out_validity) {
                p.out_validity[c] = M;
                pop += static_cast<unsigned long long>(__popcll(M));
            }
        }
    }
    if (lane == 0 && pop) striped_add(p.out_valid_pop, pop);
}

// K3: BooleanArray and / or / not with strict null propagation (boolean.rs:120-165).
// One thread per 64-bit output word.  kind: 0 and, 1 or, 2 not.
struct BoolOpParams {
    DevCol a, b;
    uint64_t *out_values;
    uint64_t *out_validity;  // nullptr when neither input has validity
    unsigned long long *out_valid_pop;
    uint64_t n;
    int32_t kind;
};
static __global__ __launch_bounds__(256) void boolop_kernel(const BoolOpParams p) {
    const uint64_t nwords = (p.n + 63) / 64;
    unsigned long long pop = 0;
    for (uint64_t w = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; w < nwords;
         w += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        const uint64_t tail = low_mask(p.n - w * 64);
        const uint64_t pa = p.a.offset + w * 64;
        const uint64_t va = load_bits64(static_cast<const uint8_t *>(p.a.values), pa, p.a.values_bytes);
        uint64_t m = p.a.validity ? load_bits64(p.a.validity, pa, p.a.validity_bytes) : ~0ull;
        uint64_t r;
        if (p.kind == 2) r = ~va;
        else {
            const uint64_t pbp = p.b.offset + w * 64;
            const uint64_t vb = load_bits64(static_cast<const uint8_t *>(p.b.values), pbp, p.b.values_bytes);
            if (p.b.validity) m &= load_bits64(p.b.validity, pbp, p.b.validity_bytes);
            r = p.kind == 0 ? (va & vb) : (va | vb);
        }
        m &= tail;
        p.out_values[w] = r & m;  // false under null, tail bits zero
        if (p.out_validity) {
            p.out_validity[w] = m;
            pop += static_cast<unsigned long long>(__popcll(m));
        }
    }
    pop = wave_sum64(pop);
    if (lane_id() == 0 && pop) striped_add(p.out_valid_pop, pop);
}

// count_true / count_false / valid count (boolean.rs:167-180, primitive.rs:90-105).
// out[0] += popc(values & validity), out[1] += popc(~values & validity), out[2] += popc(validity)
struct PopParams {
    const uint8_t *values;  // may be nullptr (null_count of a primitive array)
    const uint8_t *validity;
    uint64_t values_bytes, validity_bytes, offset, n;
    unsigned long long *out;
};
static __global__ __launch_bounds__(256) void popcount_kernel(const PopParams p) {
    const uint64_t nwords = (p.n + 63) / 64;
    unsigned long long t = 0, f = 0, v = 0;
    for (uint64_t w = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; w < nwords;
         w += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        const uint64_t tail = low_mask(p.n - w * 64);
        const uint64_t pos = p.offset + w * 64;
        const uint64_t m = (p.validity ? load_bits64(p.validity, pos, p.validity_bytes) : ~0ull) & tail;
        const uint64_t x = p.values ? load_bits64(p.values, pos, p.values_bytes) : 0;
        t += __popcll(x & m);
        f += __popcll(~x & m);
        v += __popcll(m);
    }
    t = wave_sum64(t);
    f = wave_sum64(f);
    v = wave_sum64(v);
    if (lane_id() == 0) {
        if (t) striped_add(p.out, t);
        if (f) striped_add(p.out + kStripeSlotWords, f);
        if (v) striped_add(p.out + 2 * kStripeSlotWords, v);
    }
}

// control-block words [slot] += sum of the slot's stripes, for the slots in `mask`; the stripes are left zero
static __global__ __launch_bounds__(64) void fold_stripes_kernel(unsigned long long *stripes, unsigned long long *ctrl_words, uint32_t mask) {
    const int lane = lane_id();
    for (int slot = 0; slot < kStripeSlots; ++slot) {
        if (!((mask >> slot) & 1)) continue;
        unsigned long long *s = stripes + slot * kStripeSlotWords + (lane & (kStripes - 1)) * kStripeWords;
        unsigned long long v = 0;
        if (lane < kStripes) {
            v = *s;
            *s = 0;
        }
        v = wave_sum64(v);
        if (lane == 0 && v) ctrl_words[slot] += v;
    }
}

// counts[seg] += set bits of the LSB-first bitmap `words` inside the bit range [bounds[seg], bounds[seg + 1]).
// The ranges are cut into chunks of chunk_words (>= kSegChunkWords) words by the host (items[i] = {segment, chunk within it}:
// about eight items per CU when the ranges are long, because the atomics of one segment are served one at a time); one
// wave per chunk, one atomic per chunk: a million 1024-row ranges and four 64 Mi-row ranges both fill the device.
// Survivor count of every input batch out of the selection bitmap of a coalesced launch, null count of every
// output batch out of the compacted validity (rv_filter_project_batches).
constexpr uint64_t kSegChunkWords = 4096;
struct SegItem {
    uint32_t segment, chunk;
};
static __global__ __launch_bounds__(256) void segment_popcount_kernel(const uint64_t *words, const uint64_t *bounds, const SegItem *items,
                                                               uint64_t nitems, uint64_t chunk_words, unsigned long long *counts) {
    const int lane = lane_id();
    const uint64_t wave0 = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = (static_cast<uint64_t>(gridDim.x) * blockDim.x) >> 6;
    for (uint64_t i = wave0; i < nitems; i += nwaves) {
        const SegItem it = items[i];
        const uint64_t lo = bounds[it.segment], hi = bounds[it.segment + 1];
        if (hi <= lo) continue;
        const uint64_t w0 = lo >> 6, w1 = (hi - 1) >> 6;
        const uint64_t c0 = w0 + static_cast<uint64_t>(it.chunk) * chunk_words;
        const uint64_t c1 = c0 + chunk_words - 1 < w1 ? c0 + chunk_words - 1 : w1;
        uint64_t acc = 0;
        for (uint64_t w = c0 + lane; w <= c1; w += 64) {
            uint64_t x = words[w];
            if (w == w0) x &= ~low_mask(lo & 63);
            if (w == w1 && (hi & 63)) x &= low_mask(hi & 63);
            acc += static_cast<uint64_t>(__popcll(x));
        }
        acc = wave_sum64(acc);
        if (lane == 0 && acc) atomicAdd(&counts[it.segment], static_cast<unsigned long long>(acc));
    }
}

// The same for ranges of EQUAL length that tile [0, n_bits): counts[k] = set bits in [k * chunk_bits, min((k + 1) * chunk_bits, n_bits)).
// No tables, no atomics: one wave per range (the 1024-row RecordBatches of dataframe_to_batches, streaming.rs:135-233 --
// 16 words each; up to kSegChunkWords words per range, longer ranges take the general kernel).
static __global__ __launch_bounds__(256) void uniform_segment_popcount_kernel(const uint64_t *words, uint64_t n_bits, uint64_t chunk_bits, uint64_t nchunks,
                                                                       unsigned long long *counts) {
    const int lane = lane_id();
    const uint64_t wave0 = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = (static_cast<uint64_t>(gridDim.x) * blockDim.x) >> 6;
    for (uint64_t k = wave0; k < nchunks; k += nwaves) {
        const uint64_t lo = k * chunk_bits, hi = lo + chunk_bits < n_bits ? lo + chunk_bits : n_bits;
        uint64_t acc = 0;
        if (hi > lo) {
            const uint64_t w0 = lo >> 6, w1 = (hi - 1) >> 6;
            for (uint64_t w = w0 + lane; w <= w1; w += 64) {
                uint64_t x = words[w];
                if (w == w0) x &= ~low_mask(lo & 63);
                if (w == w1 && (hi & 63)) x &= low_mask(hi & 63);
                acc += static_cast<uint64_t>(__popcll(x));
            }
        }
        acc = wave_sum64(acc);
        if (lane == 0) counts[k] = acc;
    }
}

// Per-batch survivor counts out of the per-wave counts the fused pass leaves behind (FusedParams::wave_counts): batch b is
// `per_batch` consecutive wave ranges.  `counts` may be pinned host memory (the caller's own array): 8 bytes per batch
// cross PCIe once, written by the device, and no read-back is queued.  One thread per batch for short runs, one wave
// per batch for long ones.
static __global__ __launch_bounds__(256) void batch_counts_from_waves(const uint32_t *wave_counts, uint64_t nwaves, uint64_t per_batch, uint64_t nbatches,
                                                               unsigned long long *counts) {
    if (per_batch < 32) {  // one thread per batch
        for (uint64_t b = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; b < nbatches; b += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
            unsigned long long acc = 0;
            for (uint64_t k = b * per_batch; k < (b + 1) * per_batch && k < nwaves; ++k) acc += wave_counts[k];
            counts[b] = acc;
        }
        return;
    }
    const int lane = lane_id();
    if (per_batch < 4096) {  // one wave per batch
        const uint64_t wave0 = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
        const uint64_t nw = (static_cast<uint64_t>(gridDim.x) * blockDim.x) >> 6;
        for (uint64_t b = wave0; b < nbatches; b += nw) {
            uint64_t acc = 0;
            const uint64_t hi = (b + 1) * per_batch < nwaves ? (b + 1) * per_batch : nwaves;
            for (uint64_t k = b * per_batch + lane; k < hi; k += 64) acc += wave_counts[k];
            acc = wave_sum64(acc);
            if (lane == 0) counts[b] = acc;
        }
        return;
    }
    __shared__ uint64_t s_part[4];  // one workgroup per batch (few, long batches)
    for (uint64_t b = blockIdx.x; b < nbatches; b += gridDim.x) {
        uint64_t acc = 0;
        const uint64_t hi = (b + 1) * per_batch < nwaves ? (b + 1) * per_batch : nwaves;
        for (uint64_t k = b * per_batch + threadIdx.x; k < hi; k += blockDim.x) acc += wave_counts[k];
        acc = wave_sum64(acc);
        if (lane == 0) s_part[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) counts[b] = s_part[0] + s_part[1] + s_part[2] + s_part[3];
        __syncthreads();
    }
}

// bits [offset, offset+n) -> offset 0, tail bits zero (download of sliced bit buffers)
static __global__ __launch_bounds__(256) void copy_bits_kernel(const uint8_t *src, uint64_t src_bytes, uint64_t offset,
                                                        uint64_t n, uint64_t *out) {
    const uint64_t nwords = (n + 63) / 64;
    for (uint64_t w = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; w < nwords;
         w += static_cast<uint64_t>(gridDim.x) * blockDim.x)
        out[w] = load_bits64(src, offset + w * 64, src_bytes) & low_mask(n - w * 64);
}

// RecordBatch::take for one column (record_batch.rs:131-178): gather by index, null slots
// -> placeholder 0 / false, validity word per 64 output rows.
struct TakeParams {
    DevCol col;
    const uint64_t *indices;
    uint64_t *out_values;    // elements, or bit words for DT_BOOLEAN
    uint64_t *out_validity;  // nullptr when the input has no validity
    unsigned long long *out_valid_pop;
    uint64_t n;  // number of indices
};
static __global__ __launch_bounds__(256) void take_kernel(const TakeParams p) {
    const int lane = lane_id();
    const uint64_t nchunks = (p.n + 63) / 64;
    const uint64_t wave0 = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = (static_cast<uint64_t>(gridDim.x) * blockDim.x) >> 6;
    unsigned long long pop = 0;
    for (uint64_t c = wave0; c < nchunks; c += nwaves) {
        const uint64_t i = c * 64 + lane;
        const bool in = i < p.n;
        const uint64_t src = in ? p.indices[i] + p.col.offset : 0;
        bool valid = in;
        if (in && p.col.validity) valid = (p.col.validity[src >> 3] >> (src & 7)) & 1;
        if (p.col.dtype == DT_BOOLEAN) {
            const uint8_t *vals = static_cast<const uint8_t *>(p.col.values);
            const bool bit = in && valid && ((vals[src >> 3] >> (src & 7)) & 1);
            const uint64_t m = ballot64(bit);
            if (lane == 0) p.out_values[c] = m;
        } else if (in) {
            p.out_values[i] = valid ? static_cast<const uint64_t *>(p.col.values)[src] : 0;
        }
        if (p.out_validity) {
            const uint64_t m = ballot64(valid);
            if (lane == 0) {
                p.out_validity[c] = m;
                pop += static_cast<unsigned long long>(__popcll(m));
            }
        }
    }
    if (lane == 0 && pop) striped_add(p.out_valid_pop, pop);
}

// Bounds pre-pass of RecordBatch::take (record_batch.rs:109-116) for an index list that lives on the device:
// *first_bad = the smallest position whose index is >= rows (ULLONG_MAX when every index is in range), so that the
// host can report the FIRST offending index, as the reference's loop does.
static __global__ __launch_bounds__(256) void take_bounds_kernel(const uint64_t *indices, uint64_t n, uint64_t rows, unsigned long long *first_bad) {
    unsigned long long bad = ~0ull;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += static_cast<uint64_t>(gridDim.x) * blockDim.x)
        if (indices[i] >= rows && i < bad) bad = i;
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        const unsigned long long o = (static_cast<unsigned long long>(__shfl_xor(static_cast<uint32_t>(bad >> 32), s, 64)) << 32) |
                                     __shfl_xor(static_cast<uint32_t>(bad), s, 64);
        bad = o < bad ? o : bad;
    }
    if (lane_id() == 0 && bad != ~0ull) atomicMin(first_bad, bad);
}

// concat_arrays (record_batch.rs:277-342): every output row finds its part by binary
// search over the part start offsets; null slots -> placeholder.
struct ConcatPart {
    const void *values;
    const uint8_t *validity;
    uint64_t offset;
};
struct ConcatParams {
    const ConcatPart *parts;     // device array [nparts]
    const uint64_t *part_start;  // device array [nparts + 1], part_start[nparts] == n
    uint64_t *out_values;
    uint64_t *out_validity;  // nullptr when no part has validity
    unsigned long long *out_valid_pop;
    uint64_t n;
    uint32_t nparts;
    int32_t dtype;
};
static __global__ __launch_bounds__(256) void concat_kernel(const ConcatParams p) {
    const int lane = lane_id();
    const uint64_t nchunks = (p.n + 63) / 64;
    const uint64_t wave0 = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = (static_cast<uint64_t>(gridDim.x) * blockDim.x) >> 6;
    unsigned long long pop = 0;
    for (uint64_t c = wave0; c < nchunks; c += nwaves) {
        const uint64_t i = c * 64 + lane;
        const bool in = i < p.n;
        bool valid = in, bit = false;
        if (in) {
            uint32_t lo = 0, hi = p.nparts;  // last part with part_start <= i
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (p.part_start[mid] <= i) lo = mid;
                else hi = mid;
            }
            const ConcatPart part = p.parts[lo];
            const uint64_t src = part.offset + (i - p.part_start[lo]);
            if (part.validity) valid = (part.validity[src >> 3] >> (src & 7)) & 1;
            if (p.dtype == DT_BOOLEAN) bit = valid && ((static_cast<const uint8_t *>(part.values)[src >> 3] >> (src & 7)) & 1);
            else p.out_values[i] = valid ? static_cast<const uint64_t *>(part.values)[src] : 0;
        }
        if (p.dtype == DT_BOOLEAN) {
            const uint64_t m = ballot64(bit);
            if (lane == 0) p.out_values[c] = m;
        }
        if (p.out_validity) {
            const uint64_t m = ballot64(valid);
            if (lane == 0) {
                p.out_validity[c] = m;
                pop += static_cast<unsigned long long>(__popcll(m));
            }
        }
    }
    if (lane == 0 && pop) striped_add(p.out_valid_pop, pop);
}

// ---- dataframe_to_batches (streaming.rs:135-233): null cells become 0 / 0.0 / false, the bitmap is dropped ----------
struct FillNullsParams {
    DevCol col;
    uint64_t n;
    uint64_t *out;  // value types: n elements; Boolean: ceil(n/64) words with the tail bits zero
};
static __global__ __launch_bounds__(256) void fill_nulls_kernel(const FillNullsParams p) {
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (p.col.dtype == DT_BOOLEAN) {
        if (i * 64 >= p.n) return;
        const uint64_t pos = p.col.offset + i * 64;
        uint64_t w = load_bits64(static_cast<const uint8_t *>(p.col.values), pos, p.col.values_bytes);
        if (p.col.validity) w &= load_bits64(p.col.validity, pos, p.col.validity_bytes);
        const uint64_t left = p.n - i * 64;
        p.out[i] = left >= 64 ? w : (w & low_mask(left));
        return;
    }
    if (i >= p.n) return;
    const uint64_t e = p.col.offset + i;
    const bool valid = !p.col.validity || ((p.col.validity[e >> 3] >> (e & 7)) & 1);
    p.out[i] = valid ? static_cast<const uint64_t *>(p.col.values)[e] : 0;
}

}  // namespace rvk
