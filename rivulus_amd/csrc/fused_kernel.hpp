// K1+K2 fused: predicate scan -> selection bits -> order-preserving stream compaction,
// ONE pass over HBM (gfx950 / CDNA4, wave64).
//
// Replaces, in one launch, the reference's
//   eager mask loop + per-column clone      src/physical_plan/plan.rs:112-147
//   bool -> index scan                      src/execution/record_batch.rs:235-240
//   take_array gather through builders      src/execution/record_batch.rs:131-178
//
// Structure (per workgroup = one tile of WAVES*64*R rows):
//   1. draw a tile id from a ticket counter (ids follow start order, so a tile only ever
//      waits on tiles whose workgroups already run: no dispatch-order assumption);
//   2. every lane loads R rows of each 8-byte column (coalesced 8/16-byte loads, all
//      issued before the first use) and keeps them in registers;
//   3. compare terms -> per-row survive bits; __ballot + popcount give per-chunk counts;
//   4. decoupled look-back over 8-byte {status,value} descriptors (one relaxed
//      agent-scope atomic store/load each: the payload IS the flag, so no fence)
//      yields the tile's exclusive output offset;
//   5. survivors are staged in LDS at their in-tile rank (ballot + mbcnt prefix), then
//      written to HBM as whole coalesced runs; validity bits are staged as bytes and
//      packed to words, boundary words merged with atomicOr.
// Output order == input order (reference: ascending index list, record_batch.rs:235-240).
//
// Feature flags are template parameters so the lean variant (BASELINE config 2: one
// Int64 column, no nulls) carries no code or registers for the others.
#pragma once

#include "device_common.hpp"

namespace rvk {

constexpr int kMaxBitStreams = 4;  // extra bit streams compacted alongside (Boolean columns)

enum : int {
    FF_VALIDITY = 1,  // some loaded 8-byte column has a null bitmap
    FF_BOOL = 2,      // predicate has terms over bit-packed Boolean columns
    FF_XS = 4,        // Boolean columns are compacted (extra bit streams)
    FF_SEL = 8,       // the selection bitmap is materialised
    FF_ALL = 15
};

// A bit stream compacted with the rows: out bit = src bit (& mask bit).
struct BitStream {
    const uint8_t *src;
    const uint8_t *mask;  // nullptr: none.  Boolean values use mask = validity (boolean.rs:29-32)
    uint64_t *out;        // zero-initialised by the host (boundary words are OR-merged)
    uint64_t src_bytes;
    uint64_t mask_bytes;
    uint64_t offset;
};

// What the scan front end (loads + predicate) reads; shared by the fused compaction
// kernel and the masked-aggregate kernel.
struct ScanInputs {
    DevCol cols[kMaxValueCols];
    DevCol bcols[kMaxBoolCols];
    DevTerm terms[kMaxTerms];
    uint64_t n;  // rows
    int32_t nterms;
    int32_t pad;
};

struct FusedParams {
    ScanInputs in;
    BitStream xs[kMaxBitStreams];
    uint64_t *out_values[kMaxValueCols];    // nullptr: slot not projected
    uint64_t *out_validity[kMaxValueCols];  // nullptr: no validity produced; else zero-initialised
    unsigned long long *out_valid_pop;      // [kMaxValueCols + kMaxBitStreams] set-bit counts
    uint64_t *out_selection;                // nullptr: selection bitmap not materialised
    uint64_t *state;                        // [ntiles] look-back descriptors, zeroed per launch
    uint32_t *ticket;                       // zeroed per launch
    unsigned long long *out_count;          // total survivors
    uint32_t *err;                          // set when a bounded spin gives up
    uint32_t ntiles;
    uint32_t cap_rows;                      // LDS staging capacity in rows (per round)
    int32_t nxs;
    int32_t pad;
};

constexpr uint64_t kStAgg = 1ull << 62;  // tile aggregate available
constexpr uint64_t kStPfx = 2ull << 62;  // inclusive prefix available
constexpr uint64_t kStVal = (1ull << 62) - 1;
constexpr uint32_t kSpinLimit = 1u << 22;

__device__ __forceinline__ uint64_t ld_state(const uint64_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_state(uint64_t *p, uint64_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Decoupled look-back, executed by one full wave.  Returns the exclusive prefix of `tile`
// (wave-uniform) and publishes the tile's inclusive prefix.  Kept out of line: it runs
// once per tile and must not add to the register budget of the streaming code.
static __device__ __attribute__((noinline)) uint64_t lookback_exclusive(uint64_t *state, uint32_t tile, uint64_t aggregate,
                                                                 uint32_t *err) {
    const int lane = lane_id();
    if (tile == 0) {
        if (lane == 0) st_state(&state[0], kStPfx | aggregate);
        return 0;
    }
    if (lane == 0) st_state(&state[tile], kStAgg | aggregate);
    uint64_t excl = 0;
    int64_t base = static_cast<int64_t>(tile) - 1;
    uint32_t spins = 0;
    for (;;) {
        const int64_t idx = base - lane;
        const uint64_t s = idx >= 0 ? ld_state(&state[idx]) : kStPfx;  // before tile 0: prefix 0
        const uint32_t st = static_cast<uint32_t>(s >> 62);
        const uint64_t pm = ballot64(st == 2);
        const uint64_t im = ballot64(st == 0);
        const uint64_t nearest = pm & (0 - pm);             // lowest lane holding a prefix
        const uint64_t below = pm ? (nearest - 1) : ~0ull;  // lanes nearer than it
        if (im & below) {                                   // a needed descriptor is not there yet
            if (++spins > kSpinLimit) {
                if (lane == 0) atomicExch(err, 1u);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
            continue;
        }
        const uint64_t take = below | nearest;
        const uint64_t contrib = ((take >> lane) & 1) ? (s & kStVal) : 0;
        excl += wave_sum64(contrib);
        if (pm) break;
        base -= 64;
    }
    excl = uniform64(excl);
    if (lane == 0) st_state(&state[tile], kStPfx | ((excl + aggregate) & kStVal));
    return excl;
}

// R per-row bits for this lane out of per-chunk 64-bit words.  word_of(q) must be
// wave-uniform: the word covering rows [q*64, q*64+64) of the wave's row range.
template <int R, int VEC, class F>
__device__ __forceinline__ uint32_t gather_row_bits(F word_of, int lane) {
    uint32_t out = 0;
    if constexpr (VEC == 1) {
#pragma unroll
        for (int j = 0; j < R; ++j) out |= static_cast<uint32_t>((word_of(j) >> lane) & 1) << j;
    } else {
#pragma unroll
        for (int j = 0; j < R / 2; ++j) {
            const uint64_t wa = word_of(2 * j), wb = word_of(2 * j + 1);
            const uint64_t w = lane < 32 ? wa : wb;
            out |= static_cast<uint32_t>((w >> ((2 * lane) & 63)) & 3) << (2 * j);  // rows 2l, 2l+1
        }
    }
    return out;
}

template <int R, bool HV, class Cmp>
__device__ __forceinline__ uint32_t term_mask(const uint64_t (&v)[R], uint32_t vb, bool null_v, Cmp cmp) {
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < R; ++k) m |= static_cast<uint32_t>(cmp(v[k])) << k;
    if constexpr (HV) m = (m & vb) | (null_v ? ~vb : 0u);
    return m;
}

template <int R, bool HV>
__device__ __forceinline__ uint32_t eval_value_term(const DevTerm &t, const uint64_t (&v)[R], uint32_t vb) {
    const int64_t lit = t.lit;
    const double litf = __longlong_as_double(t.lit);
    const bool nv = t.null_v != 0;
    switch (t.code) {
        case TC_I64 + OP_EQ: return term_mask<R, HV>(v, vb, nv, [=](uint64_t b) { return static_cast<int64_t>(b) == lit; });
        case TC_I64 + OP_NE: return term_mask<R, HV>(v, vb, nv, [=](uint64_t b) { return static_cast<int64_t>(b) != lit; });
        case TC_I64 + OP_LT: return term_mask<R, HV>(v, vb, nv, [=](uint64_t b) { return static_cast<int64_t>(b) < lit; });
        case TC_I64 + OP_GT: return term_mask<R, HV>(v, vb, nv, [=](uint64_t b) { return static_cast<int64_t>(b) > lit; });
        case TC_I64 + OP_LE: return term_mask<R, HV>(v, vb, nv, [=](uint64_t b) { return static_cast<int64_t>(b) <= lit; });
        case TC_I64 + OP_GE: return term_mask<R, HV>(v, vb, nv, [=](uint64_t b) { return static_cast<int64_t>(b) >= lit; });
        case TC_F64 + OP_EQ: return term_mask<R, HV>(v, vb, nv, [=](uint64_t b) { return __longlong_as_double(b) == litf; });
        case TC_F64 + OP_NE: return term_mask<R, HV>(v, vb, nv, [=](uint64_t b) { return __longlong_as_double(b) != litf; });
        case TC_F64 + OP_LT: return term_mask<R, HV>(v, vb, nv, [=](uint64_t b) { return __longlong_as_double(b) < litf; });
        case TC_F64 + OP_GT: return term_mask<R, HV>(v, vb, nv, [=](uint64_t b) { return __longlong_as_double(b) > litf; });
        case TC_F64 + OP_LE: return term_mask<R, HV>(v, vb, nv, [=](uint64_t b) { return __longlong_as_double(b) <= litf; });
        case TC_F64 + OP_GE: return term_mask<R, HV>(v, vb, nv, [=](uint64_t b) { return __longlong_as_double(b) >= litf; });
        default: {
            const uint32_t cv = t.const_v ? ~0u : 0u;
            if constexpr (HV) return (cv & vb) | (nv ? ~vb : 0u);
            return cv;
        }
    }
}

// Scan front end: every lane loads R rows of each 8-byte column into registers (all loads
// issued before the first use), gathers validity bits and evaluates the AND-of-terms
// predicate.  pb bit k == row k of this lane survives.
template <int NCOLS, int R, int VEC, int FLAGS>
__device__ __forceinline__ void scan_rows(const ScanInputs &in, uint64_t wave_base, bool full, int lane,
                                          uint64_t (&v)[NCOLS > 0 ? NCOLS : 1][R],
                                          uint32_t (&vb)[NCOLS > 0 ? NCOLS : 1], uint32_t &pb) {
    constexpr uint32_t ALL = R == 32 ? 0xFFFFFFFFu : ((1u << R) - 1);
#pragma unroll
    for (int c = 0; c < NCOLS; ++c) {
        const uint64_t *src = static_cast<const uint64_t *>(in.cols[c].values) + in.cols[c].offset + wave_base;
        if constexpr (VEC == 1) {
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const uint32_t row = j * 64 + lane;
                v[c][j] = (full || wave_base + row < in.n) ? src[row] : 0;
            }
        } else {
#pragma unroll
            for (int j = 0; j < R / 2; ++j) {
                const uint32_t row = j * 128 + 2 * lane;
                if (full || wave_base + row + 1 < in.n) {
                    const ulonglong2 t = *reinterpret_cast<const ulonglong2 *>(src + row);
                    v[c][2 * j] = t.x;
                    v[c][2 * j + 1] = t.y;
                } else {
                    v[c][2 * j] = wave_base + row < in.n ? src[row] : 0;
                    v[c][2 * j + 1] = 0;
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < (NCOLS > 0 ? NCOLS : 1); ++c) vb[c] = ALL;
    if constexpr ((FLAGS & FF_VALIDITY) != 0) {
#pragma unroll
        for (int c = 0; c < NCOLS; ++c) {
            const uint8_t *val = in.cols[c].validity;
            if (val) {
                const uint64_t pos0 = in.cols[c].offset + wave_base, nb = in.cols[c].validity_bytes;
                vb[c] = gather_row_bits<R, VEC>([&](int q) { return load_bits64(val, pos0 + q * 64u, nb); }, lane);
            }
        }
    }
    pb = ALL;
    if (!full) {
        pb = 0;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const uint32_t row = VEC == 1 ? k * 64 + lane : (k / 2) * 128 + 2 * lane + (k & 1);
            pb |= static_cast<uint32_t>(wave_base + row < in.n) << k;
        }
    }
#pragma unroll
    for (int c = 0; c < NCOLS; ++c)
        for (int t = 0; t < in.nterms; ++t)
            if (!in.terms[t].is_bool && in.terms[t].slot == c)
                pb &= eval_value_term<R, (FLAGS & FF_VALIDITY) != 0>(in.terms[t], v[c], vb[c]);
    if constexpr ((FLAGS & FF_BOOL) != 0) {
        for (int t = 0; t < in.nterms; ++t) {
            if (!in.terms[t].is_bool) continue;
            const DevTerm term = in.terms[t];
            const DevCol col = in.bcols[term.slot];
            pb &= gather_row_bits<R, VEC>(
                [&](int q) {
                    const uint64_t pos = col.offset + wave_base + q * 64u;
                    const uint64_t V = load_bits64(static_cast<const uint8_t *>(col.values), pos, col.values_bytes);
                    const uint64_t M = col.validity ? load_bits64(col.validity, pos, col.validity_bytes) : ~0ull;
                    return eval_bool_word(term, V, M);
                },
                lane);
        }
    }
}

// in-tile rank of each surviving row of this lane; calls sink(k, rank - lo) for ranks in [lo, hi)
template <int R, int VEC, class Sink>
__device__ __forceinline__ void for_each_survivor(uint32_t pb, uint32_t wave_prefix, uint32_t lo, uint32_t hi,
                                                  Sink sink) {
    uint32_t running = wave_prefix;
    if constexpr (VEC == 1) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const bool p = (pb >> j) & 1;
            const uint64_t m = ballot64(p);
            const uint32_t rank = running + mbcnt(m);
            if (p && rank >= lo && rank < hi) sink(j, rank - lo);
            running += static_cast<uint32_t>(__popcll(m));
        }
    } else {
#pragma unroll
        for (int j = 0; j < R / 2; ++j) {
            const bool p0 = (pb >> (2 * j)) & 1, p1 = (pb >> (2 * j + 1)) & 1;
            const uint64_t m0 = ballot64(p0), m1 = ballot64(p1);
            const uint32_t r0 = running + mbcnt(m0) + mbcnt(m1);  // rows before row 2l of the chunk
            const uint32_t r1 = r0 + (p0 ? 1u : 0u);
            if (p0 && r0 >= lo && r0 < hi) sink(2 * j, r0 - lo);
            if (p1 && r1 >= lo && r1 < hi) sink(2 * j + 1, r1 - lo);
            running += static_cast<uint32_t>(__popcll(m0) + __popcll(m1));
        }
    }
}

// staged validity bytes [0,cnt) -> output bit range [g0, g0+cnt); adds the number of set
// bits to *s_pop (LDS).  Fully covered words are stored, boundary words OR-merged (the host
// zero-fills the buffer).  Cold relative to the value path: out of line.
static __device__ __attribute__((noinline)) void flush_bits(const uint8_t *stage, uint32_t cnt, uint64_t g0, uint64_t *out,
                                                     uint32_t *s_pop) {
    if (cnt == 0) return;
    const uint64_t w0 = g0 >> 6, w1 = (g0 + cnt - 1) >> 6;
    uint32_t pop = 0;
    for (uint64_t w = w0 + threadIdx.x; w <= w1; w += blockDim.x) {
        const uint64_t b_lo = (w << 6) > g0 ? (w << 6) : g0;
        const uint64_t b_hi = ((w + 1) << 6) < g0 + cnt ? ((w + 1) << 6) : g0 + cnt;
        uint64_t word = 0;
        for (uint64_t b = b_lo; b < b_hi; ++b) word |= static_cast<uint64_t>(stage[b - g0] & 1) << (b & 63);
        if (b_hi - b_lo == 64) out[w] = word;
        else if (word) atomicOr(reinterpret_cast<unsigned long long *>(&out[w]), static_cast<unsigned long long>(word));
        pop += static_cast<uint32_t>(__popcll(word));
    }
    if (pop) atomicAdd(s_pop, pop);
}

constexpr int kLdsHeader = 128;

template <int NCOLS, int R, int VEC, int WAVES, int FLAGS>
__global__ __launch_bounds__(WAVES * 64) void fused_filter_compact(const FusedParams p) {
    static_assert(VEC == 1 || (VEC == 2 && R % 2 == 0), "VEC");
    static_assert(WAVES <= 16, "WAVES");
    constexpr int NV = NCOLS > 0 ? NCOLS : 1;
    constexpr uint32_t ROWS_PER_WAVE = 64u * R;
    constexpr uint32_t TILE = ROWS_PER_WAVE * WAVES;
    constexpr bool kValidity = (FLAGS & FF_VALIDITY) != 0;
    constexpr bool kXs = (FLAGS & FF_XS) != 0;
    constexpr bool kSel = (FLAGS & FF_SEL) != 0;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *s_tile = reinterpret_cast<uint32_t *>(smem);
    uint64_t *s_excl = reinterpret_cast<uint64_t *>(smem + 8);
    uint32_t *s_pop = reinterpret_cast<uint32_t *>(smem + 16);   // [8]
    uint32_t *s_wtot = reinterpret_cast<uint32_t *>(smem + 48);  // [WAVES <= 16]

    const int lane = lane_id();
    const uint32_t wave = uniform32(threadIdx.x >> 6);

    // ---- 1. ticket ---------------------------------------------------------------------------
    if (threadIdx.x == 0) *s_tile = __hip_atomic_fetch_add(p.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if constexpr (kValidity || kXs)
        if (threadIdx.x < 8) s_pop[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t tile = uniform32(*s_tile);
    const uint64_t tile_base = static_cast<uint64_t>(tile) * TILE;
    const uint64_t wave_base = tile_base + static_cast<uint64_t>(wave) * ROWS_PER_WAVE;
    const bool full = tile_base + TILE <= p.in.n;

    // ---- 2./3. loads, validity bits, predicate -> survive bits ----------------------------------
    uint64_t v[NV][R];
    uint32_t vb[NV];
    uint32_t pb;
    scan_rows<NCOLS, R, VEC, FLAGS>(p.in, wave_base, full, lane, v, vb, pb);

    // extra bit streams (Boolean columns travelling with the rows)
    uint32_t xb[kMaxBitStreams];
    if constexpr (kXs) {
#pragma unroll
        for (int s = 0; s < kMaxBitStreams; ++s) {
            xb[s] = 0;
            if (s < p.nxs) {
                const BitStream bs = p.xs[s];
                xb[s] = gather_row_bits<R, VEC>(
                    [&](int q) {
                        const uint64_t pos = bs.offset + wave_base + q * 64u;
                        uint64_t w = load_bits64(bs.src, pos, bs.src_bytes);
                        if (bs.mask) w &= load_bits64(bs.mask, pos, bs.mask_bytes);
                        return w;
                    },
                    lane);
            }
        }
    }

    // ---- selection bitmap (optional) + per-wave survivor count ----------------------------------
    uint32_t wave_total = 0;
    if constexpr (VEC == 1) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const uint64_t m = ballot64((pb >> j) & 1);
            wave_total += static_cast<uint32_t>(__popcll(m));
            if constexpr (kSel)
                if (p.out_selection && lane == 0 && wave_base + j * 64u < p.in.n) p.out_selection[(wave_base >> 6) + j] = m;
        }
    } else {
#pragma unroll
        for (int j = 0; j < R / 2; ++j) {
            const uint64_t m0 = ballot64((pb >> (2 * j)) & 1), m1 = ballot64((pb >> (2 * j + 1)) & 1);
            wave_total += static_cast<uint32_t>(__popcll(m0) + __popcll(m1));
            if constexpr (kSel) {
                if (p.out_selection && lane < 16) {
                    // rows 8*lane .. 8*lane+7 of the 128-row chunk: interleave 4 even + 4 odd bits
                    const uint32_t e = static_cast<uint32_t>(m0 >> (4 * lane)) & 0xF, o = static_cast<uint32_t>(m1 >> (4 * lane)) & 0xF;
                    auto spread4 = [](uint32_t x) { return (x & 1) | ((x & 2) << 1) | ((x & 4) << 2) | ((x & 8) << 3); };
                    const uint64_t row0 = wave_base + j * 128u + lane * 8u;
                    if (row0 < p.in.n)
                        reinterpret_cast<uint8_t *>(p.out_selection)[row0 >> 3] = static_cast<uint8_t>(spread4(e) | (spread4(o) << 1));
                }
            }
        }
    }
    if (lane == 0) s_wtot[wave] = wave_total;
    __syncthreads();

    uint32_t wave_prefix = 0, tile_count = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
        const uint32_t t = s_wtot[w];
        wave_prefix += (static_cast<uint32_t>(w) < wave) ? t : 0;
        tile_count += t;
    }
    wave_prefix = uniform32(wave_prefix);
    tile_count = uniform32(tile_count);

    // ---- LDS staging carve (byte offsets from smem) -----------------------------------------------
    const uint32_t cap = p.cap_rows;
    uint32_t off_v[NV], off_b[NV], off_x[kMaxBitStreams];
    {
        uint32_t cur = kLdsHeader;
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            off_v[c] = cur;
            if (c < NCOLS && p.out_values[c]) cur += cap * 8;
        }
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            off_b[c] = cur;
            if constexpr (kValidity)
                if (c < NCOLS && p.out_validity[c]) cur += cap;
        }
#pragma unroll
        for (int s = 0; s < kMaxBitStreams; ++s) {
            off_x[s] = cur;
            if constexpr (kXs)
                if (s < p.nxs) cur += cap;
        }
    }

    const uint32_t nrounds = (tile_count + cap - 1) / cap;
    uint64_t excl = 0;
    for (uint32_t r = 0; r == 0 || r < nrounds; ++r) {
        const uint32_t lo = r * cap, hi = lo + cap;
        if (r > 0) __syncthreads();  // previous round's staging fully flushed
        // ---- 5a. scatter this round's survivors into LDS at their in-tile rank ----------------
        if (wave_total && wave_prefix < hi && wave_prefix + wave_total > lo) {
#pragma unroll
            for (int c = 0; c < NCOLS; ++c) {
                if (!p.out_values[c]) continue;
                uint64_t *sv = reinterpret_cast<uint64_t *>(smem + off_v[c]);
                if constexpr (kValidity) {
                    uint8_t *sb = smem + off_b[c];
                    const bool hv = p.out_validity[c] != nullptr;
                    for_each_survivor<R, VEC>(pb, wave_prefix, lo, hi, [&](int k, uint32_t pos) {
                        const bool valid = (vb[c] >> k) & 1;
                        sv[pos] = valid ? v[c][k] : 0;  // placeholder 0 / 0.0 (record_batch.rs:142-146)
                        if (hv) sb[pos] = valid;
                    });
                } else {
                    for_each_survivor<R, VEC>(pb, wave_prefix, lo, hi, [&](int k, uint32_t pos) { sv[pos] = v[c][k]; });
                }
            }
            if constexpr (kXs) {
#pragma unroll
                for (int s = 0; s < kMaxBitStreams; ++s)
                    if (s < p.nxs) {
                        uint8_t *sx = smem + off_x[s];
                        for_each_survivor<R, VEC>(pb, wave_prefix, lo, hi, [&](int k, uint32_t pos) { sx[pos] = (xb[s] >> k) & 1; });
                    }
            }
        }
        // ---- 4. look-back overlaps the other waves' scatter ----------------------------------
        if (r == 0 && wave == 0) {
            const uint64_t e = lookback_exclusive(p.state, tile, tile_count, p.err);
            if (lane == 0) {
                *s_excl = e;
                if (tile == p.ntiles - 1) *p.out_count = e + tile_count;
            }
        }
        __syncthreads();
        if (r == 0) excl = uniform64(*s_excl);
        if (r >= nrounds) break;  // tile without survivors
        // ---- 5b. flush: whole coalesced runs ------------------------------------------------------
        const uint32_t cnt = tile_count - lo < cap ? tile_count - lo : cap;
        const uint64_t g0 = excl + lo;
#pragma unroll
        for (int c = 0; c < NCOLS; ++c) {
            if (!p.out_values[c]) continue;
            uint64_t *dst = p.out_values[c] + g0;
            const uint64_t *sv = reinterpret_cast<const uint64_t *>(smem + off_v[c]);
            for (uint32_t k = threadIdx.x; k < cnt; k += WAVES * 64) dst[k] = sv[k];
            if constexpr (kValidity)
                if (p.out_validity[c]) flush_bits(smem + off_b[c], cnt, g0, p.out_validity[c], &s_pop[c]);
        }
        if constexpr (kXs) {
#pragma unroll
            for (int s = 0; s < kMaxBitStreams; ++s)
                if (s < p.nxs) flush_bits(smem + off_x[s], cnt, g0, p.xs[s].out, &s_pop[kMaxValueCols + s]);
        }
    }

    // ---- set-bit counts of the compacted bit streams (null counts on the host side) ---------------
    if constexpr (kValidity || kXs) {
        __syncthreads();
        if (threadIdx.x < kMaxValueCols + kMaxBitStreams && s_pop[threadIdx.x])
            atomicAdd(&p.out_valid_pop[threadIdx.x], static_cast<unsigned long long>(s_pop[threadIdx.x]));
    }
}

}  // namespace rvk
