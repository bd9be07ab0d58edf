// K1+K2 fused: predicate scan -> selection bits -> order-preserving stream compaction,
// ONE pass over HBM (gfx950 / CDNA4, wave64).
//
// Replaces, in one launch, the reference's
//   eager mask loop + per-column clone      src/physical_plan/plan.rs:112-147
//   bool -> index scan                      src/execution/record_batch.rs:235-240
//   take_array gather through builders      src/execution/record_batch.rs:131-178
//
// Structure (persistent workgroups; one tile = WAVES*64*R rows per loop iteration):
//   1. draw a tile id from a ticket counter (ids follow draw order, so a tile only ever
//      waits on tiles already held by a running workgroup: no dispatch-order assumption);
//      tickets are drawn three iterations ahead;
//   2. every lane loads R rows of each 8-byte column (coalesced 8/16-byte nontemporal buffer
//      loads, all issued before the first use, one tile ahead) and keeps them in registers;
//      the words of null bitmaps / Boolean columns travel with them (lane q = word q);
//   3. compare terms -> 64-bit WAVE masks per row slot (v_cmp is the ballot); validity, null
//      policy and the AND of terms are scalar instructions on those masks;
//   4. survivors are staged in the wave's private LDS slot at their in-wave rank (mbcnt);
//      one barrier exchanges the wave counts and the tile aggregate is published as an 8-byte
//      {status,value} descriptor (one relaxed agent-scope atomic store: the payload IS the flag);
//   5. the tile's output offset: a scanner wave (workgroup 0) turns aggregates into inclusive
//      prefixes; two iterations later the tile reads ONE descriptor (the classic decoupled
//      look-back is the fallback) and each wave writes its slot to HBM as one coalesced run;
//      validity bits are staged as bytes and packed to words, boundary words merged with atomicOr.
// Output order == input order (reference: ascending index list, record_batch.rs:235-240).
//
// Feature flags are template parameters so the lean variant (BASELINE config 2: one
// Int64 column, no nulls) carries no code or registers for the others.
#pragma once

#include <type_traits>

#include "device_common.hpp"

namespace rvk {

constexpr int kMaxBitStreams = 4;  // extra bit streams compacted alongside (Boolean columns)

enum : int {
    FF_VALIDITY = 1,  // some loaded 8-byte column has a null bitmap
    FF_BOOL = 2,      // predicate has terms over bit-packed Boolean columns
    FF_XS = 4,        // Boolean columns are compacted (extra bit streams)
    FF_SEL = 8,       // the selection bitmap is materialised
    FF_ALL = 15,
    FF_STAMP = 16     // diagnostic build only: per-phase s_memtime sums (never timed, never shipped to callers)
};
// Predicate shape known at compile time: exactly ONE compare term, on value slot 0, of that type,
// no null bitmap (BASELINE config 2: `x > lit`).  Predicate, rank and staging then run as one
// straight-line pass per row slot (no packed predicate bits, no second ballot pass).
enum : int { FF_ONE_I64 = 32, FF_ONE_F64 = 64 };
// Every loaded 8-byte column is projected, and has an output bitmap exactly when it has an input bitmap
// (`filter(...)` keeping the columns it tests -- BASELINE configs 2 and 3): no per-column checks in the
// staging loop.
enum : int { FF_PROJALL = 128 };
// With FF_PROJALL: no projected column can hold a null among the survivors (every nullable column is tested by
// a term that drops its nulls -- the streaming composition of BASELINE config 3), so no validity is staged or
// written; the bitmaps are still read for the predicate.
enum : int { FF_NONULL = 256 };
// The term list is a conjunctive normal form with negated literals (ScanInputs::expr_mode): generic shapes only.
enum : int { FF_EXPR = 512 };

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
    // Predicate expressions with OR / NOT (rv_predicate::expr), lowered by the host to conjunctive normal form:
    //   survive = live & strict validity & (AND over groups (OR over literals (negate ? ~term : term))) ^ negate_result
    // expr_mode 0: the plain AND of the terms (every launch of BASELINE configs 2 and 3).
    int32_t expr_mode;
    int32_t negate_result;   // the lowered form is the CNF of NOT(expression): complement the accumulated mask
    uint32_t strict_values;  // bit c: a row survives only where value slot c is valid (strict null propagation of
    uint32_t strict_bools;   // BooleanArray::and / or / not under RV_NULL_DROPS, boolean.rs:120-165); same for bcols
    uint32_t pad;
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
    uint32_t *overflow;                     // set when survivors did not fit out_capacity (speculative output sizing)
    uint64_t out_capacity;                  // rows the output buffers hold; writes past it are dropped and flagged
    uint32_t *redo_count;                   // tiles left to the redo kernel (a wave outgrew its slot)
    unsigned long long *redo;               // [ntiles][2]: {tile, exclusive output offset}
    unsigned long long *stamps;             // FF_STAMP builds: [8] cycle sums + tile count
    uint32_t ntiles;
    uint32_t cap_rows;                      // LDS staging capacity in rows (per round)
    int32_t nxs;
    int32_t debug;  // honoured by FF_STAMP (diagnostic) instantiations only: 1 = skip output stores, 2 = skip the look-back, 4 = count,
                    // 8 = no scanner wave, 16 = tile 1 never publishes its count (fault injection for the bounded spins)
    int32_t depth;  // 1: two slot stages, write out one iteration later; 2: three stages, two later
    uint32_t spin_limit;  // polls a look-back / the scanner waits for a missing descriptor before giving up (*err = 1)
};

constexpr uint64_t kStAgg = 1ull << 62;  // tile aggregate available
constexpr uint64_t kStPfx = 2ull << 62;  // inclusive prefix available
constexpr uint64_t kStVal = (1ull << 62) - 1;
constexpr uint32_t kSpinLimit = 1u << 22;  // default of FusedParams::spin_limit (seconds of polling)

__device__ __forceinline__ uint64_t ld_state(const uint64_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_state(uint64_t *p, uint64_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Decoupled look-back, executed by one full wave.  The tile's aggregate has already been
// published (publish_aggregate, as early as the count is known).  lookback_finish returns the
// exclusive prefix of `tile` (wave-uniform) and publishes the tile's inclusive prefix.
//
// One poll inspects kLookK * 64 predecessors (kLookK descriptors per lane, all loads in
// flight together).  The window has to cover every tile that can be in flight at once:
// with a 64-wide window the prefix front advances only 64 tiles per poll round trip
// (~2 us), which capped the whole kernel at ~27 tiles/us (profiles/README.md, r01a).
constexpr int kLookK = 8;

__device__ __forceinline__ void publish_aggregate(uint64_t *state, uint32_t tile, uint64_t aggregate) {
    st_state(&state[tile], (tile == 0 ? kStPfx : kStAgg) | aggregate);
}

// one poll: the kLookK * 64 descriptors in front of `base` (nearest first), all loads in flight
__device__ __forceinline__ void lookback_issue(const uint64_t *state, int64_t base, int lane, uint64_t (&s)[kLookK]) {
#pragma unroll
    for (int k = 0; k < kLookK; ++k) {
        const int64_t idx = base - (lane + 64 * k);
        s[k] = idx >= 0 ? ld_state(&state[idx]) : kStPfx;  // before tile 0: prefix 0
    }
}

// Look-back of `tile`: returns its exclusive prefix (wave-uniform) and publishes its inclusive
// prefix.  The FALLBACK of the output-offset lookup (the scanner normally has the prefix ready).
// OUT OF LINE on purpose: its 16 descriptor registers and the reduction temporaries then never
// overlap the streaming code's live ranges; inlined it cost ~30 VGPRs.
[[maybe_unused]] static __device__ __attribute__((noinline)) uint64_t lookback_exclusive(uint64_t *state, uint32_t tile, uint64_t aggregate,
                                                                 uint32_t *err, uint32_t spin_limit, unsigned long long *poll_stats) {
    const int lane = lane_id();
    if (tile == 0) return 0;
    uint64_t s[kLookK];
    uint32_t polls = 1, windows = 0;
    uint64_t excl = 0;
    int64_t base = static_cast<int64_t>(tile) - 1;
    uint32_t spins = 0;
    lookback_issue(state, base, lane, s);
    for (;;) {
        uint64_t contrib = 0;
        bool found = false, ready = true;
#pragma unroll
        for (int k = 0; k < kLookK; ++k) {
            if (found || !ready) continue;  // wave-uniform
            const uint32_t st = static_cast<uint32_t>(s[k] >> 62);
            const uint64_t pm = ballot64(st == 2);
            const uint64_t im = ballot64(st == 0);
            const uint64_t nearest = pm & (0 - pm);             // lowest lane holding a prefix
            const uint64_t below = pm ? (nearest - 1) : ~0ull;  // lanes nearer than it
            if (im & below) {                                   // a needed descriptor is not there yet
                ready = false;
            } else {
                const uint64_t take = below | nearest;
                contrib += ((take >> lane) & 1) ? (s[k] & kStVal) : 0;
                found = pm != 0;
            }
        }
        if (ready) {
            excl += wave_sum64(contrib);
            ++windows;
            if (found) break;
            base -= 64 * kLookK;
        } else {
            if (++spins > spin_limit) {
                if (lane == 0) atomicExch(err, 1u);
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        lookback_issue(state, base, lane, s);
        ++polls;
    }
    if (poll_stats && lane == 0) {
        atomicAdd(&poll_stats[0], static_cast<unsigned long long>(polls));
        atomicAdd(&poll_stats[1], static_cast<unsigned long long>(windows));
    }
    excl = uniform64(excl);
    if (lane == 0) st_state(&state[tile], kStPfx | ((excl + aggregate) & kStVal));
    return excl;
}

// Scanner: ONE wave (wave 0 of workgroup 0) walks the descriptor array in tile
// order, turns aggregates into inclusive prefixes and publishes them, 512 tiles per poll.  Every
// aggregate is then read once instead of ~512 times: with 512 tiles in flight, every tile polling
// the 512 descriptors in front of it put ~16 k line requests per generation on the one memory
// channel that holds the ~4 KiB live window of the array, and the look-back cost 0.3-0.5 ms of a
// 1.8 ms launch (profiles/README.md, ablation).  Compute workgroups now read ONE descriptor per
// tile (their predecessor's inclusive prefix) and fall back to lookback_exclusive only when it is
// not there yet -- so correctness never depends on the scanner being resident or keeping up.
// A descriptor that already holds a prefix (published by a fallback look-back) is adopted.
static __device__ __attribute__((noinline)) void scanner_wave(uint64_t *state, uint32_t ntiles, uint32_t *err, uint32_t spin_limit,
                                                              unsigned long long *stats) {
    const int lane = lane_id();
    uint64_t carry = 0;  // inclusive prefix of tile next-1
    uint32_t next = 0, idle = 0;
    while (next < ntiles) {
        uint64_t s[kLookK];
#pragma unroll
        for (int k = 0; k < kLookK; ++k) {
            const uint32_t idx = next + 64u * k + lane;  // ascending: position p = 64k + lane
            s[k] = idx < ntiles ? ld_state(&state[idx]) : 0;
        }
        uint32_t done = 0;
        bool stop = false;
#pragma unroll
        for (int k = 0; k < kLookK; ++k) {
            if (stop) continue;  // wave-uniform
            const uint32_t idx = next + 64u * k + lane;
            const uint32_t st = static_cast<uint32_t>(s[k] >> 62);
            const uint64_t in = ballot64(idx < ntiles);
            const uint64_t valid = ballot64(st != 0) & in;
            // leading run of published descriptors of this group of 64
            const uint64_t missing = ~valid & in;
            const uint32_t run = missing ? static_cast<uint32_t>(__builtin_ctzll(missing)) : static_cast<uint32_t>(__popcll(in));
            if (run) {
                const uint64_t runmask = low_mask(run);
                // adopt the last prefix already published inside the run, scan the aggregates after it
                const uint64_t pm = ballot64(st == 2) & runmask;
                const int last_p = pm ? 63 - __builtin_clzll(pm) : -1;
                uint64_t base = carry;
                if (last_p >= 0) base = uniform64(__shfl(s[k] & kStVal, last_p, 64));
                uint64_t x = (lane > last_p && ((runmask >> lane) & 1)) ? (s[k] & kStVal) : 0;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {  // inclusive wave scan
                    const uint64_t y = (static_cast<uint64_t>(__shfl_up(static_cast<uint32_t>(x >> 32), d, 64)) << 32) |
                                       __shfl_up(static_cast<uint32_t>(x), d, 64);
                    if (lane >= d) x += y;
                }
                const uint64_t incl = base + x;
                if (lane > last_p && ((runmask >> lane) & 1)) st_state(&state[idx], kStPfx | (incl & kStVal));
                carry = uniform64(__shfl(incl, static_cast<int>(run) - 1, 64));
                done += run;
            }
            if (run < 64) stop = true;
        }
        if (stats && lane == 0) {
            stats[0] += 1;  // polls
            stats[1] += done;
            if (!done) stats[2] += 1;
        }
        if (done) {
            next += done;
            idle = 0;
        } else {
            if (++idle > spin_limit) {
                if (lane == 0) atomicExch(err, 1u);
                return;
            }
            __builtin_amdgcn_s_sleep(8);
        }
    }
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
    const bool nv = t.null_v();
    switch (t.code()) {
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
            const uint32_t cv = t.const_v() ? ~0u : 0u;
            if constexpr (HV) return (cv & vb) | (nv ? ~vb : 0u);
            return cv;
        }
    }
}

// Scan front end, part 1: every lane issues the loads of R rows of each 8-byte column (all
// before any use) into registers.  Values are read from HBM exactly once.
// The loads go through a buffer descriptor built per wave from wave-uniform values: base =
// first row of the wave, num_records = bytes left in the column.  The hardware bounds check
// returns 0 for rows past the end, so the ragged last tile needs no branches, and addressing
// is one 32-bit lane offset plus immediates (guide T8/T20).
// Cache policy of the row stream (buffer-instruction aux bits, gfx940+: bit 1 = nt).  Every row is
// read once and every output row written once: nontemporal keeps them from displacing each other
// in L2 / MALL.  Measured on MI355X (tools/micro/mixbench.hip): read-only 7.14 TB/s with nt loads
// against 6.33 TB/s without; read 8 GB + write 0.8 GB in 1.47 ms (nt, nt) against 1.67 ms (nt, plain).
constexpr int kStreamPolicy = 2;

template <int NCOLS, int R, int VEC>
__device__ __forceinline__ void load_rows(const ScanInputs &in, uint64_t wave_base, int lane,
                                          uint64_t (&v)[NCOLS > 0 ? NCOLS : 1][R]) {
    constexpr uint32_t ROWS_PER_WAVE = 64u * R;
    const uint64_t left = in.n > wave_base ? in.n - wave_base : 0;
    const uint32_t nbytes = uniform32(static_cast<uint32_t>(left < ROWS_PER_WAVE ? left : ROWS_PER_WAVE) * 8u);
#pragma unroll
    for (int c = 0; c < NCOLS; ++c) {
        const uint64_t base = uniform64(reinterpret_cast<uint64_t>(in.cols[c].values) + (in.cols[c].offset + wave_base) * 8);
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(base), 0, nbytes, 0x00020000);
        if constexpr (VEC == 1) {
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const rv_u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rsrc, lane * 8, j * 512, kStreamPolicy);
                v[c][j] = (static_cast<uint64_t>(t.y) << 32) | t.x;
            }
        } else {
#pragma unroll
            for (int j = 0; j < R / 2; ++j) {
                const rv_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, j * 1024, kStreamPolicy);
                v[c][2 * j] = (static_cast<uint64_t>(t.y) << 32) | t.x;
                v[c][2 * j + 1] = (static_cast<uint64_t>(t.w) << 32) | t.z;
            }
        }
    }
}

// The validity words of a wave's rows travel with the row prefetch: lane k holds the aligned 64-bit
// word (first bit of the wave >> 6) + k of every column's bitmap, k <= R (R windows of 64 bits need
// R + 1 aligned words).  One vector load per column and tile, in flight with the rows; the windows
// are cut out later with readlane + scalar funnel shifts (validity_masks).
template <int NCOLS, int R>
__device__ __forceinline__ void load_validity_words(const ScanInputs &in, uint64_t wave_base, int lane,
                                                    uint64_t (&vw)[NCOLS > 0 ? NCOLS : 1]) {
#pragma unroll
    for (int c = 0; c < NCOLS; ++c) {
        vw[c] = ~0ull;
        const uint8_t *val = in.cols[c].validity;
        if (val && lane <= R) vw[c] = load_word_safe(val, ((in.cols[c].offset + wave_base) >> 6) + lane, in.cols[c].validity_bytes);
    }
}
// the same for any bit buffer (Boolean predicate columns, Boolean columns travelling with the rows): lane q <= R
// holds aligned word (first bit >> 6) + q
template <int R>
__device__ __forceinline__ uint64_t load_bit_words(const uint8_t *buf, uint64_t first_bit, uint64_t nbytes, int lane) {
    return (buf && lane <= R) ? load_word_safe(buf, (first_bit >> 6) + lane, nbytes) : ~0ull;
}
// Scan front end, part 2: validity bits and the AND-of-terms predicate over the loaded rows.
// pb bit k == row k of this lane survives.
template <int NCOLS, int R, int VEC, int FLAGS>
__device__ __forceinline__ void eval_rows(const ScanInputs &in, uint64_t wave_base, bool full, int lane,
                                          const uint64_t (&v)[NCOLS > 0 ? NCOLS : 1][R],
                                          uint32_t (&vb)[NCOLS > 0 ? NCOLS : 1], uint32_t &pb) {
    constexpr uint32_t ALL = R == 32 ? 0xFFFFFFFFu : ((1u << R) - 1);
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
    if (in.expr_mode) {
        // conjunctive normal form with negated literals; null rows of strictly propagating columns are masked up
        // front, so a literal's truth under a null never matters there
        uint32_t base = pb, acc = ALL, grp = 0;
#pragma unroll
        for (int c = 0; c < NCOLS; ++c)
            if ((in.strict_values >> c) & 1) base &= vb[c];
        for (int t = 0; t < in.nterms; ++t) {
            const DevTerm term = in.terms[t];
            uint32_t x = 0;
            if (!term.is_bool()) {
#pragma unroll
                for (int c = 0; c < NCOLS; ++c)
                    if (term.slot() == static_cast<uint32_t>(c)) x = eval_value_term<R, (FLAGS & FF_VALIDITY) != 0>(term, v[c], vb[c]);
            } else if constexpr ((FLAGS & FF_BOOL) != 0) {
                const DevCol col = in.bcols[term.slot()];
                x = gather_row_bits<R, VEC>(
                    [&](int q) {
                        const uint64_t pos = col.offset + wave_base + q * 64u;
                        const uint64_t V = load_bits64(static_cast<const uint8_t *>(col.values), pos, col.values_bytes);
                        const uint64_t M = col.validity ? load_bits64(col.validity, pos, col.validity_bytes) : ~0ull;
                        return eval_bool_word(term, V, M);
                    },
                    lane);
            }
            grp |= term.negate() ? ~x : x;
            if (term.group_end()) {
                acc &= grp;
                grp = 0;
            }
        }
        if constexpr ((FLAGS & FF_BOOL) != 0) {
#pragma unroll
            for (int c = 0; c < kMaxBoolCols; ++c)
                if (((in.strict_bools >> c) & 1) && in.bcols[c].validity) {
                    const DevCol col = in.bcols[c];
                    base &= gather_row_bits<R, VEC>([&](int q) { return load_bits64(col.validity, col.offset + wave_base + q * 64u, col.validity_bytes); }, lane);
                }
        }
        pb = base & (in.negate_result ? ~acc : acc);
        return;
    }
#pragma unroll
    for (int c = 0; c < NCOLS; ++c)
        for (int t = 0; t < in.nterms; ++t)
            if (!in.terms[t].is_bool() && in.terms[t].slot() == static_cast<uint32_t>(c))
                pb &= eval_value_term<R, (FLAGS & FF_VALIDITY) != 0>(in.terms[t], v[c], vb[c]);
    if constexpr ((FLAGS & FF_BOOL) != 0) {
        for (int t = 0; t < in.nterms; ++t) {
            if (!in.terms[t].is_bool()) continue;
            const DevTerm term = in.terms[t];
            const DevCol col = in.bcols[term.slot()];
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

template <int NCOLS, int R, int VEC, int FLAGS>
__device__ __forceinline__ void scan_rows(const ScanInputs &in, uint64_t wave_base, bool full, int lane,
                                          uint64_t (&v)[NCOLS > 0 ? NCOLS : 1][R],
                                          uint32_t (&vb)[NCOLS > 0 ? NCOLS : 1], uint32_t &pb) {
    load_rows<NCOLS, R, VEC>(in, wave_base, lane, v);
    eval_rows<NCOLS, R, VEC, FLAGS>(in, wave_base, full, lane, v, vb, pb);
}

// ---- mask-major front end of the fused kernel ---------------------------------------------------------
// Row slot k of a wave (the k-th row of every lane) is described by 64-bit WAVE masks, bit l =
// lane l: they live in SGPRs, v_cmp produces them for free, validity / null policy / AND of terms
// are scalar instructions, and inverse_ballot turns one back into the exec mask or a v_cndmask
// condition without a single VALU instruction.  The generic shapes are ISSUE bound, not HBM bound
// (a CU issues about one scalar and one vector instruction per cycle for all of its 16 waves;
// tools/stamp3.py), so the instruction count per row slot is what this code is written for.
// VEC == 1: slot k = rows [64k, 64k+64) of the wave, so a validity mask is simply the (unaligned)
// 64-bit window of the bitmap.  VEC == 2: lane l holds rows 2l, 2l+1 of the 128-row chunk j in
// slots 2j, 2j+1, and the window pair is split per lane.
__device__ __forceinline__ bool lane_of(uint64_t wave_mask) { return __builtin_amdgcn_inverse_ballot_w64(wave_mask); }

template <int R, int VEC, class F>
__device__ __forceinline__ void word_masks(F word_of, int lane, uint64_t (&M)[R]) {
    if constexpr (VEC == 1) {
#pragma unroll
        for (int k = 0; k < R; ++k) M[k] = uniform64(word_of(k));
    } else {
#pragma unroll
        for (int j = 0; j < R / 2; ++j) {
            const uint64_t wa = uniform64(word_of(2 * j)), wb = uniform64(word_of(2 * j + 1));
            const uint64_t w = lane < 32 ? wa : wb;
            const uint32_t two = static_cast<uint32_t>(w >> ((2 * lane) & 63)) & 3u;
            M[2 * j] = ballot64((two & 1u) != 0);
            M[2 * j + 1] = ballot64((two & 2u) != 0);
        }
    }
}
// lane `l` of acc <- the wave-uniform 64-bit value x
__device__ __forceinline__ uint64_t writelane64(uint64_t acc, uint64_t x, int l) { return lane_id() == l ? x : acc; }
// Selection bitmap of a wave's rows: the words are collected in lane registers (lane q = word q of the wave's
// range) and stored once per tile as one coalesced run, instead of one 8-byte store per row slot.
// VEC == 1: slot k IS word k.  VEC == 2: chunk j (slots 2j, 2j+1; lane l = rows 2l, 2l+1) gives words 2j, 2j+1;
// row r of the chunk sits in lane r >> 1, so lane t fetches the pair of lane (t >> 1) + 32 * half and ballots its bit.
template <int VEC>
__device__ __forceinline__ void sel_collect(uint64_t &acc, int slot, uint64_t m0, uint64_t m1, int lane) {
    if constexpr (VEC == 1) {
        (void)m1;
        acc = writelane64(acc, m0, slot);
    } else {
        const int two = (__builtin_amdgcn_inverse_ballot_w64(m0) ? 1 : 0) | (__builtin_amdgcn_inverse_ballot_w64(m1) ? 2 : 0);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int pair = __shfl(two, (lane >> 1) + 32 * half, 64);
            acc = writelane64(acc, ballot64(((pair >> (lane & 1)) & 1) != 0), slot + half);
        }
    }
}
template <int R>
__device__ __forceinline__ void sel_store(uint64_t acc, uint64_t *out, uint64_t wave_base, uint64_t n, int lane) {
    if (lane < R && wave_base + static_cast<uint64_t>(lane) * 64 < n) out[(wave_base >> 6) + lane] = acc;
}

// The words loaded by load_validity_words (lane q = aligned word q of the wave's range) -> windows:
// lane q = bits [64q, 64q + 64) of the range.  One funnel shift per lane and tile; a slot's validity
// mask is then two v_readlane away (validity_of) and never has to be kept in SGPRs.
__device__ __forceinline__ uint64_t validity_windows(uint64_t vw, uint32_t shift) {
    if (shift == 0) return vw;
    const uint64_t next = (static_cast<uint64_t>(__shfl_down(static_cast<uint32_t>(vw >> 32), 1, 64)) << 32) |
                          __shfl_down(static_cast<uint32_t>(vw), 1, 64);
    return (vw >> shift) | (next << (64 - shift));
}
// lanes of slot k whose row lies inside a wave range of `rem` rows
template <int VEC>
__device__ __forceinline__ uint64_t live_mask(int32_t rem, int k) {
    int32_t cnt;
    if constexpr (VEC == 1) cnt = rem - 64 * k;
    else cnt = (rem - 128 * (k / 2) + ((k & 1) ? 0 : 1)) >> 1;  // rows 2l (even slot) / 2l+1 (odd slot) below rem
    return cnt <= 0 ? 0ull : low_mask(static_cast<uint64_t>(cnt));
}
template <int R, class Cmp>
__device__ __forceinline__ void cmp_masks(const uint64_t (&v)[R], uint64_t (&C)[R], Cmp cmp) {
#pragma unroll
    for (int k = 0; k < R; ++k) C[k] = ballot64(cmp(v[k]));
}
// S[k] &= term(rows of slot k).  valid_of(k) = validity mask of slot k of the term's column, used when
// hv.  The AnyValue truth table is lowered on the host (predicate.rs): null rows take null_v, valid rows
// the compare (or const_v).  (C & V) | (null_v ? ~V : 0) is C & V or C | ~V: two scalar instructions.
// C[k] = lanes of slot k whose (valid) cell satisfies the compare of term t
template <int R>
__device__ __forceinline__ void compare_masks(const DevTerm &t, const uint64_t (&v)[R], uint64_t (&C)[R]) {
    const int64_t lit = t.lit;
    const double litf = __longlong_as_double(t.lit);
    switch (t.code()) {
        case TC_I64 + OP_EQ: cmp_masks<R>(v, C, [=](uint64_t b) { return static_cast<int64_t>(b) == lit; }); break;
        case TC_I64 + OP_NE: cmp_masks<R>(v, C, [=](uint64_t b) { return static_cast<int64_t>(b) != lit; }); break;
        case TC_I64 + OP_LT: cmp_masks<R>(v, C, [=](uint64_t b) { return static_cast<int64_t>(b) < lit; }); break;
        case TC_I64 + OP_GT: cmp_masks<R>(v, C, [=](uint64_t b) { return static_cast<int64_t>(b) > lit; }); break;
        case TC_I64 + OP_LE: cmp_masks<R>(v, C, [=](uint64_t b) { return static_cast<int64_t>(b) <= lit; }); break;
        case TC_I64 + OP_GE: cmp_masks<R>(v, C, [=](uint64_t b) { return static_cast<int64_t>(b) >= lit; }); break;
        case TC_F64 + OP_EQ: cmp_masks<R>(v, C, [=](uint64_t b) { return __longlong_as_double(b) == litf; }); break;
        case TC_F64 + OP_NE: cmp_masks<R>(v, C, [=](uint64_t b) { return __longlong_as_double(b) != litf; }); break;
        case TC_F64 + OP_LT: cmp_masks<R>(v, C, [=](uint64_t b) { return __longlong_as_double(b) < litf; }); break;
        case TC_F64 + OP_GT: cmp_masks<R>(v, C, [=](uint64_t b) { return __longlong_as_double(b) > litf; }); break;
        case TC_F64 + OP_LE: cmp_masks<R>(v, C, [=](uint64_t b) { return __longlong_as_double(b) <= litf; }); break;
        case TC_F64 + OP_GE: cmp_masks<R>(v, C, [=](uint64_t b) { return __longlong_as_double(b) >= litf; }); break;
        default: {
            const uint64_t cv = t.const_v() ? ~0ull : 0ull;
#pragma unroll
            for (int k = 0; k < R; ++k) C[k] = cv;
        }
    }
}
template <int R, class VM>
__device__ __forceinline__ void and_value_term(const DevTerm &t, const uint64_t (&v)[R], VM valid_of, bool hv, uint64_t (&S)[R]) {
    uint64_t C[R];
    compare_masks<R>(t, v, C);
    if (!hv) {
#pragma unroll
        for (int k = 0; k < R; ++k) S[k] &= C[k];
    } else if (t.null_v()) {
#pragma unroll
        for (int k = 0; k < R; ++k) S[k] &= C[k] | ~valid_of(k);
    } else {
#pragma unroll
        for (int k = 0; k < R; ++k) S[k] &= C[k] & valid_of(k);
    }
}

// X[k] = truth of term t on the rows of slot k, null rows at null_v (expression launches: the literal is then
// negated / ORed into its group by the caller)
template <int R, class VM>
__device__ __forceinline__ void value_term_truth(const DevTerm &t, const uint64_t (&v)[R], VM valid_of, bool hv, uint64_t (&X)[R]) {
    compare_masks<R>(t, v, X);
    if (hv) {
        if (t.null_v()) {
#pragma unroll
            for (int k = 0; k < R; ++k) X[k] |= ~valid_of(k);
        } else {
#pragma unroll
            for (int k = 0; k < R; ++k) X[k] &= valid_of(k);
        }
    }
}

// ---- lane form (VEC == 1) ----------------------------------------------------------------------------------
// A 64-bit VGPR value whose lane k holds the wave mask of row slot k ("mask vector").  With 8-byte loads slot k is rows
// [64k, 64k + 64) of the wave's range, i.e. the k-th 64-bit word of every bit buffer, so null bitmaps, Boolean columns
// and the selection bitmap ARE lane-form already (lane q = word q) and AND / OR / NOT of whole tiles are single VALU
// instructions.  Only compares produce per-slot scalar masks; they are dropped into their lane with v_writelane.  The
// generic shapes ran out of SGPRs with one 2R-SGPR array per mask set (hundreds of compiler spills to VGPR lanes, see
// profiles/README.md); in lane form a mask set costs two VGPRs.
template <int L>
__device__ __forceinline__ uint64_t set_lane64(uint64_t acc, uint64_t uniform_x) {
    // v_writelane_b32 vdst, ssrc (data), lane: ROCm 7.2's clang has no builtin for it.  The data comes out of a v_cmp
    // (VALU write of an SGPR pair read as DATA: no wait states needed); the lane is an inline constant (a second SGPR
    // operand would break the constant-bus limit of one).
    // gfx940+: a VALU read of an SGPR needs two wait states after the VALU that wrote it (the compiler inserts them for
    // its own instructions, not for operands of inline asm: without the s_nop a few rows per million were lost).
    uint32_t lo = static_cast<uint32_t>(acc), hi = static_cast<uint32_t>(acc >> 32);
    const uint32_t xlo = static_cast<uint32_t>(uniform_x), xhi = static_cast<uint32_t>(uniform_x >> 32);
    asm("s_nop 1\n\tv_writelane_b32 %0, %2, %4\n\tv_writelane_b32 %1, %3, %4" : "+v"(lo), "+v"(hi) : "s"(xlo), "s"(xhi), "n"(L));
    return (static_cast<uint64_t>(hi) << 32) | lo;
}
template <int K, int N, class F>
__device__ __forceinline__ void static_for(F f) {
    if constexpr (K < N) {
        f(std::integral_constant<int, K>{});
        static_for<K + 1, N>(f);
    }
}
#define RV_LANE_CMP(EXPR)                                        \
    static_for<0, R>([&](auto kc) {                              \
        constexpr int k = decltype(kc)::value;                   \
        const uint64_t b = v[k];                                 \
        acc = set_lane64<k>(acc, ballot64(EXPR));                \
    });                                                          \
    break;
// lane k of the result = lanes of slot k whose (valid) cell satisfies the compare of term t
template <int R>
__device__ __forceinline__ uint64_t compare_lanes(const DevTerm &t, const uint64_t (&v)[R]) {
    const int64_t lit = t.lit;
    const double litf = __longlong_as_double(t.lit);
    uint64_t acc = 0;
    switch (t.code()) {
        case TC_I64 + OP_EQ: RV_LANE_CMP(static_cast<int64_t>(b) == lit)
        case TC_I64 + OP_NE: RV_LANE_CMP(static_cast<int64_t>(b) != lit)
        case TC_I64 + OP_LT: RV_LANE_CMP(static_cast<int64_t>(b) < lit)
        case TC_I64 + OP_GT: RV_LANE_CMP(static_cast<int64_t>(b) > lit)
        case TC_I64 + OP_LE: RV_LANE_CMP(static_cast<int64_t>(b) <= lit)
        case TC_I64 + OP_GE: RV_LANE_CMP(static_cast<int64_t>(b) >= lit)
        case TC_F64 + OP_EQ: RV_LANE_CMP(__longlong_as_double(b) == litf)
        case TC_F64 + OP_NE: RV_LANE_CMP(__longlong_as_double(b) != litf)
        case TC_F64 + OP_LT: RV_LANE_CMP(__longlong_as_double(b) < litf)
        case TC_F64 + OP_GT: RV_LANE_CMP(__longlong_as_double(b) > litf)
        case TC_F64 + OP_LE: RV_LANE_CMP(__longlong_as_double(b) <= litf)
        case TC_F64 + OP_GE: RV_LANE_CMP(__longlong_as_double(b) >= litf)
        default: acc = t.const_v() ? ~0ull : 0ull;
    }
    return acc;
}
#undef RV_LANE_CMP

// in-wave rank of each surviving row of this lane (rows of a wave are ordered chunk by chunk,
// lane by lane); calls sink(k, rank - lo) for ranks in [lo, hi)
template <int R, int VEC, class Sink>
__device__ __forceinline__ void for_each_survivor(uint32_t pb, uint32_t lo, uint32_t hi, Sink sink) {
    uint32_t running = 0;
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

// The dynamic LDS block of the fused kernel.  Helpers address it by byte offset (generic
// pointers into LDS handed to out-of-line functions trip a gfx950 codegen bug in ROCm 7.2).
extern __shared__ __attribute__((aligned(16))) unsigned char rv_smem[];

// staged validity bytes [0,cnt) at LDS offset `stage_off` -> output bit range [g0, g0+cnt),
// executed by ONE wave; adds the number of set bits to the LDS counter at `pop_off`.  Fully
// covered words are stored, words shared with a neighbouring wave/tile are OR-merged (the
// host zero-fills the buffer).  Cold relative to the value path: out of line.
static __device__ __forceinline__ void flush_bits(uint32_t stage_off, uint32_t cnt, uint64_t g0, uint64_t *out,
                                                     uint32_t pop_off) {
    if (cnt == 0) return;
    const uint8_t *stage = rv_smem + stage_off;
    const int lane = static_cast<int>(opaque(static_cast<uint32_t>(lane_id())));
    const uint64_t w0 = g0 >> 6, w1 = (g0 + cnt - 1) >> 6;
    uint32_t pop = 0;
    // one output word per step: lane l owns bit l of the word, __ballot packs the 64 staged bytes
    for (uint64_t w = w0; w <= w1; ++w) {
        const uint64_t b = (w << 6) + lane;  // global bit position of this lane
        const bool in = b >= g0 && b < g0 + cnt;
        const uint64_t word = ballot64(in && (stage[in ? b - g0 : 0] & 1));
        const bool full = (w << 6) >= g0 && ((w + 1) << 6) <= g0 + cnt;
        if (lane == 0) {
            if (full) out[w] = word;
            else if (word) atomicOr(reinterpret_cast<unsigned long long *>(&out[w]), static_cast<unsigned long long>(word));
        }
        pop += static_cast<uint32_t>(__popcll(word));
    }
    if (lane == 0 && pop) atomicAdd(reinterpret_cast<uint32_t *>(rv_smem + pop_off), pop);
}

// The same for a bit stream that is staged as BITS (lane form, VEC == 1): LDS words [0, ceil(cnt / 64)] at `stage_off` hold
// the wave's compacted bits from bit 0 (zero above cnt) -> output bit range [g0, g0 + cnt).  One funnel shift per
// output word, every word of the run handled by its own lane (a wave's run is at most R + 1 words).
static __device__ __forceinline__ void flush_words(uint32_t stage_off, uint32_t cnt, uint64_t g0, uint64_t *out, uint32_t pop_off) {
    if (cnt == 0) return;
    const uint64_t *src = reinterpret_cast<const uint64_t *>(rv_smem + stage_off);
    const uint32_t lane = opaque(static_cast<uint32_t>(lane_id()));
    const uint32_t sh = static_cast<uint32_t>(g0 & 63), nsrc = (cnt + 63) >> 6, nw = (sh + cnt + 63) >> 6;
    uint32_t pop = 0;
    for (uint32_t i = lane; i < nw; i += 64) {
        const uint64_t cur = i < nsrc ? src[i] : 0, prev = i ? src[i - 1] : 0;
        const uint64_t val = sh ? (cur << sh) | (prev >> (64 - sh)) : cur;
        const bool first = i == 0 && sh != 0, last = i + 1 == nw && ((sh + cnt) & 63) != 0;
        uint64_t *dst = out + (g0 >> 6) + i;
        if (first || last) {  // shared with a neighbouring wave / tile: merged into the zero-filled buffer
            if (val) atomicOr(reinterpret_cast<unsigned long long *>(dst), static_cast<unsigned long long>(val));
        } else {
            *dst = val;
        }
        if (i < nsrc) pop += static_cast<uint32_t>(__popcll(cur));
    }
    pop = static_cast<uint32_t>(wave_sum64(pop));
    if (lane == 0 && pop) atomicAdd(reinterpret_cast<uint32_t *>(rv_smem + pop_off), pop);
}

constexpr int kLdsHeader = 128;
// per-wave dump area behind the slots: 64 x 8 bytes + 64 bytes.  The generic staging loop is branch-free:
// lanes without a survivor store to their own dump cell instead of being masked off (see stage_slot).
constexpr int kLdsDumpBytes = 576;

__device__ __forceinline__ unsigned long long stamp_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

// Persistent and software-pipelined.  A workgroup keeps drawing tiles until the ticket
// counter runs out.  Every wave owns a contiguous run of 64*R rows of the tile and PRIVATE
// LDS slots (one per stage), so the per-tile critical path is short:
//   wave:  wait rows -> predicate -> ballot/mbcnt ranks -> survivors into the own slot
//          -> issue the NEXT tile's row loads at once (the registers are dead)
//          -> one barrier to exchange the wave counts, publish the tile aggregate
//   later: each wave writes its own slot to out[tile offset + wave prefix ...] (coalesced run)
// and these latencies stay off it:
//   ticket        drawn three iterations ahead;
//   row loads     of the next tile are in flight across the barriers, the offset lookup and the flush;
//   output offset needed two iterations after the aggregate went out (p.depth == 2, three slot
//                 stages).  By then the scanner wave (workgroup 0) has published the prefix in
//                 front of the tile, and ONE descriptor load, issued at the top of the iteration,
//                 delivers it.  The full decoupled look-back is the fallback when the prefix is
//                 not there yet, so progress never depends on the scanner.
// 128 VGPRs at most (4 waves per SIMD): two 8-wave workgroups, or one 16-wave workgroup, per CU.
// A wave with more survivors than its slot holds (dense data) does not belong on this path: the
// tile still takes part in the scan (its output range is reserved), but its rows are left to the
// redo kernel (fused_redo_tiles), which re-reads that tile and writes it at the reserved offset.
template <int NCOLS, int R, int VEC, int WAVES, int FLAGS>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(4))) void fused_filter_compact(const FusedParams p) {
    static_assert(VEC == 1 || (VEC == 2 && R % 2 == 0), "VEC");
    static_assert(WAVES <= 16, "WAVES");
    constexpr int NV = NCOLS > 0 ? NCOLS : 1;
    constexpr uint32_t ROWS_PER_WAVE = 64u * R;
    constexpr uint32_t TILE = ROWS_PER_WAVE * WAVES;
    constexpr bool kValidity = (FLAGS & FF_VALIDITY) != 0;
    constexpr bool kXs = (FLAGS & FF_XS) != 0;
    // the FF_PROJALL generic instantiations can also materialise the selection bitmap (String / Boolean / Null
    // columns are produced from it after the pass): decided at run time there, three instructions per row slot
    constexpr bool kSel = (FLAGS & FF_SEL) != 0 || ((FLAGS & FF_PROJALL) != 0 && (FLAGS & (FF_ONE_I64 | FF_ONE_F64)) == 0);
    const bool want_sel = p.out_selection != nullptr;
    constexpr bool kStamp = (FLAGS & FF_STAMP) != 0;
    constexpr bool kOne = (FLAGS & (FF_ONE_I64 | FF_ONE_F64)) != 0;
    constexpr bool kAll = (FLAGS & FF_PROJALL) != 0;
    constexpr bool kNoNull = (FLAGS & FF_NONULL) != 0;
    constexpr bool kExpr = (FLAGS & FF_EXPR) != 0;
    static_assert(!kExpr || !kOne, "expressions run on the generic shapes");
    static_assert(!kNoNull || kAll, "FF_NONULL refines FF_PROJALL");
    static_assert(!kOne || (NCOLS == 1 && (FLAGS & (FF_VALIDITY | FF_BOOL | FF_XS)) == 0), "single-term fast path");
    // selector masks of term 0 (all ones / all zeros), fixed for the launch
    const DevTerm term0 = p.in.terms[0];
    const uint64_t SLT = term0.sel_lt() ? ~0ull : 0, SEQ = term0.sel_eq() ? ~0ull : 0, SGT = term0.sel_gt() ? ~0ull : 0,
                   SUN = term0.sel_un() ? ~0ull : 0;
    const int64_t lit0 = term0.lit;
    unsigned long long st_wait = 0, st_stage = 0, tm = 0;
    unsigned long long st_eval = 0, st_scatter = 0, st_look = 0, st_waitB = 0, st_flush = 0, st_tiles = 0, t0 = 0, t1 = 0;

    unsigned char *const smem = rv_smem;
    uint32_t *s_tick = reinterpret_cast<uint32_t *>(smem);       // [4] ring of tile ids, drawn three iterations ahead
    uint64_t *s_excl = reinterpret_cast<uint64_t *>(smem + 16);
    uint32_t *s_wtot = reinterpret_cast<uint32_t *>(smem + 56);  // [WAVES <= 16]
    // s_pop: 8 x uint32 at smem + 24 (set-bit counts of compacted bit streams)

    const int lane = lane_id();
    const uint32_t wave = uniform32(threadIdx.x >> 6);

    // workgroup 0 (dispatched first, so resident before any tile needs it) is the scanner: one wave,
    // the others leave at once
    if (blockIdx.x == 0) {
        // debug bit 3 (FF_STAMP builds): no scanner at all -- every tile takes the fallback look-back (a test of it)
        if (wave == 0 && !(kStamp && (p.debug & 8))) scanner_wave(p.state, p.ntiles, p.err, p.spin_limit, (kStamp && (p.debug & 4)) ? p.stamps + 28 : nullptr);
        return;
    }

    // ---- LDS carve: slot(stage, wave) of `cap` rows; byte offsets of the columns inside a slot ----
    const uint32_t cap = p.cap_rows;  // rows per wave slot
    uint32_t off_v[NV], off_b[NV], off_x[kMaxBitStreams];
    uint32_t slot_bytes;
    {
        uint32_t cur = 0;
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            off_v[c] = cur;
            if (c < NCOLS && (kAll || p.out_values[c])) cur += cap * 8;
        }
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            off_b[c] = cur;
            if constexpr (kValidity)
                if (c < NCOLS && ((kAll && !kNoNull) || (!kAll && p.out_validity[c]))) cur += cap;
        }
#pragma unroll
        for (int s = 0; s < kMaxBitStreams; ++s) {
            off_x[s] = cur;
            if constexpr (kXs)
                if (s < p.nxs) cur += (VEC == 1 && cap < (R + 2) * 8u) ? (R + 2) * 8u : cap;  // VEC == 1: R + 1 words of bits
        }
        slot_bytes = (cur + 15u) & ~15u;
    }
    auto slot_of = [&](uint32_t stage) { return kLdsHeader + (stage * WAVES + wave) * slot_bytes; };

    // this wave's slot rows [0, cnt) -> out[g0 ...): one coalesced run per column
    auto flush = [&](uint32_t sb, uint32_t cnt, uint64_t g0) {
        if (g0 + cnt > p.out_capacity) {  // wave-uniform; the counts stay exact, the host re-runs with buffers of that size
            if (lane == 0) *p.overflow = 1u;
            return;
        }
        const uint32_t fl = opaque(static_cast<uint32_t>(lane));  // device_common.hpp: keeps the addresses below out of scratch
#pragma unroll
        for (int c = 0; c < NCOLS; ++c) {
            if (!p.out_values[c]) continue;
            uint64_t *dst = p.out_values[c] + g0;
            const uint64_t *sv = reinterpret_cast<const uint64_t *>(smem + sb + off_v[c]);
            if (!(kStamp && (p.debug & 1)))
                for (uint32_t k = fl; k < cnt; k += 64) __builtin_nontemporal_store(sv[k], &dst[k]);
            if constexpr (kValidity && !kNoNull)
                if (p.out_validity[c]) flush_bits(sb + off_b[c], cnt, g0, p.out_validity[c], 24 + 4 * c);
        }
        if constexpr (kXs) {
#pragma unroll
            for (int s = 0; s < kMaxBitStreams; ++s)
                if (s < p.nxs) {
                    if constexpr (VEC == 1) flush_words(sb + off_x[s], cnt, g0, p.xs[s].out, 24 + 4 * (kMaxValueCols + s));
                    else flush_bits(sb + off_x[s], cnt, g0, p.xs[s].out, 24 + 4 * (kMaxValueCols + s));
                }
        }
    };

    // The predicate terms live in VGPR lanes for the whole launch (lane t = term t): the tile loop reads
    // them with v_readlane instead of scalar loads from the kernel arguments, whose latency would sit on
    // every tile's critical path.  Likewise one word of "is this column projected / has an output bitmap".
    uint32_t term_lo = 0, term_hi = 0, term_pk = 0;
    if (lane < p.in.nterms) {
        const DevTerm t = p.in.terms[lane];
        term_lo = static_cast<uint32_t>(t.lit);
        term_hi = static_cast<uint32_t>(static_cast<uint64_t>(t.lit) >> 32);
        term_pk = t.packed;
    }
    auto term_at = [&](int t) {
        DevTerm r;
        r.lit = static_cast<int64_t>((static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(term_hi), t))) << 32) |
                                     static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(term_lo), t)));
        r.packed = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(term_pk), t));
        r.pad = 0;
        return r;
    };
    uint32_t outflags = 0;  // bit c: column c projected; bit 8 + c: with an output bitmap
#pragma unroll
    for (int c = 0; c < NCOLS; ++c) outflags |= (p.out_values[c] ? 1u << c : 0u) | (p.out_validity[c] ? 0x100u << c : 0u);
    outflags = uniform32(outflags);
    const int nterms = p.in.nterms;

    // ---- prologue: three tickets, first loads -------------------------------------------------------
    if (threadIdx.x == 0) {
        s_tick[0] = __hip_atomic_fetch_add(p.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_tick[1] = __hip_atomic_fetch_add(p.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_tick[2] = __hip_atomic_fetch_add(p.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if constexpr (kValidity || kXs)
        if (threadIdx.x < 8) reinterpret_cast<uint32_t *>(smem + 24)[threadIdx.x] = 0;
    __syncthreads();
    uint32_t tile = uniform32(s_tick[0]);
    uint64_t v[NV][R];
    if (tile < p.ntiles)
        load_rows<NCOLS, R, VEC>(p.in, static_cast<uint64_t>(tile) * TILE + static_cast<uint64_t>(wave) * ROWS_PER_WAVE, lane, v);
    uint64_t vw[NV];  // validity words of the loaded rows (generic shapes with null bitmaps)
    // words of the Boolean predicate columns (values, validity) and of the Boolean columns travelling with
    // the rows (source, mask): fetched with the row prefetch, like vw
    constexpr bool kBool = (FLAGS & FF_BOOL) != 0;
    uint64_t bw[kBool ? 2 * kMaxBoolCols : 1], xw[kXs ? 2 * kMaxBitStreams : 1];
    auto prefetch_bits = [&](uint64_t first_row) {
        if constexpr (kValidity && !kOne) load_validity_words<NCOLS, R>(p.in, first_row, lane, vw);
        if constexpr (kBool) {
#pragma unroll
            for (int c = 0; c < kMaxBoolCols; ++c) {
                const DevCol col = p.in.bcols[c];
                bw[2 * c] = load_bit_words<R>(static_cast<const uint8_t *>(col.values), col.offset + first_row, col.values_bytes, lane);
                bw[2 * c + 1] = load_bit_words<R>(col.validity, col.offset + first_row, col.validity_bytes, lane);
            }
        }
        if constexpr (kXs) {
#pragma unroll
            for (int s2 = 0; s2 < kMaxBitStreams; ++s2) {
                const BitStream bs = p.xs[s2];
                const bool on = s2 < p.nxs;
                xw[2 * s2] = load_bit_words<R>(on ? bs.src : nullptr, bs.offset + first_row, bs.src_bytes, lane);
                xw[2 * s2 + 1] = load_bit_words<R>(on ? bs.mask : nullptr, bs.offset + first_row, bs.mask_bytes, lane);
            }
        }
    };
    if (tile < p.ntiles) prefetch_bits(static_cast<uint64_t>(tile) * TILE + static_cast<uint64_t>(wave) * ROWS_PER_WAVE);

    // Tiles whose survivors wait in their LDS slots for the output offset: `older` was staged two
    // iterations ago, `newer` one.  With p.depth == 2 (three slot stages) a tile is written out two
    // iterations after it published its aggregate, which gives the scanner a whole iteration to
    // publish the prefix in front of it; with p.depth == 1 (two stages, larger slots) one later.
    struct Pending {
        uint32_t tile, count, stage, wave_prefix, wave_total;
        bool dense, have;
    };
    Pending older{0, 0, 0, 0, 0, false, false}, newer{0, 0, 0, 0, 0, false, false};
    const bool deep = p.depth == 2;
    const uint32_t nstages = deep ? 3u : 2u;
    uint32_t cur_stage = 0;
    uint64_t prev_desc = 0;  // wave 0: descriptor of the tile in front of the retiring one, loaded early

    // wave 0 (holding no row data): output offset of the retiring tile, handed to the workgroup.
    // Fast path: the scanner has already published the predecessor's inclusive prefix.
    auto resolve = [&](const Pending &r, uint64_t prev) {  // prev: the descriptor in front of r.tile, wave-uniform
        uint64_t e;
        if (kStamp && (p.debug & 2)) e = static_cast<uint64_t>(r.tile) * 1024;
        else if (r.tile == 0) e = 0;
        else if ((prev >> 62) == 2) e = prev & kStVal;
        else {
            if (kStamp && (p.debug & 4) && lane == 0) atomicAdd(p.stamps + 31, 1ull);
            e = lookback_exclusive(p.state, r.tile, r.count, p.err, p.spin_limit, kStamp ? p.stamps + 6 : nullptr);
        }
        if (lane == 0) {
            *s_excl = e;
            if (r.tile == p.ntiles - 1) *p.out_count = e + r.count;
            if (r.dense) {  // leave the tile to the redo kernel, at its reserved output offset
                const uint32_t k = atomicAdd(p.redo_count, 1u);
                p.redo[2 * k] = r.tile;
                p.redo[2 * k + 1] = e;
            }
        }
    };

    for (uint32_t it = 0; tile < p.ntiles; ++it) {
        // ---- ticket for three iterations ahead: issued now, stored at the end of the iteration and read
        //      at the top of iteration it + 2 (barrier A of it + 1 lies in between) --------------------
        uint32_t ticket3 = 0;
        if (threadIdx.x == 0) ticket3 = __hip_atomic_fetch_add(p.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the pending tile's predecessor descriptor: one 8-byte load whose round trip runs under this
        // tile's predicate + staging
        const Pending ret = deep ? older : newer;  // the tile to write out in this iteration
        if (wave == 0 && ret.have && ret.tile != 0) prev_desc = ld_state(&p.state[ret.tile - 1]);
        const uint32_t next_tile = uniform32(s_tick[(it + 1) & 3]);  // stored two iterations ago
        const uint64_t tile_base = static_cast<uint64_t>(tile) * TILE;
        const uint64_t wave_base = tile_base + static_cast<uint64_t>(wave) * ROWS_PER_WAVE;
        const bool full = tile_base + TILE <= p.in.n;
        const uint64_t ntb = static_cast<uint64_t>(next_tile) * TILE;
        const bool more = next_tile < p.ntiles;

        uint32_t wave_total = 0;
        uint64_t selw = 0;  // selection words of this wave's rows (FF_SEL), lane q = word q
        const uint32_t sb = slot_of(cur_stage);
        if constexpr (kStamp) {
            t0 = stamp_now();
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the rows are here -- separates load wait from compute
            t1 = stamp_now();
            st_wait += t1 - t0;
        }
        if constexpr (kOne) {
            // ---- single-term fast path: compare -> lane masks -> rank -> own slot, row slot by row slot ----
            uint64_t *sv = reinterpret_cast<uint64_t *>(smem + sb + off_v[0]);
            const bool project = p.out_values[0] != nullptr;
            // rows of this wave inside the batch (only the last tile is ragged); compared as scalar - lane
            // part so that no per-row-slot index is kept in registers
            int32_t rem = 0;
            if (!full) {
                const int64_t left = static_cast<int64_t>(p.in.n) - static_cast<int64_t>(wave_base);
                rem = static_cast<int32_t>(left < 0 ? 0 : (left > (1 << 30) ? (1 << 30) : left));
            }
            auto row_mask = [&](uint64_t cell, int32_t srow, int32_t lrow) -> uint64_t {
                uint64_t m;
                if constexpr ((FLAGS & FF_ONE_F64) != 0) {
                    const double a = __longlong_as_double(cell), b = __longlong_as_double(lit0);
                    const uint64_t lt = ballot64(a < b), eq = ballot64(a == b), gt = ballot64(a > b);
                    m = (lt & SLT) | (eq & SEQ) | (gt & SGT) | (~(lt | eq | gt) & SUN);
                } else {
                    const uint64_t lt = ballot64(static_cast<int64_t>(cell) < lit0), eq = ballot64(static_cast<int64_t>(cell) == lit0);
                    m = (lt & SLT) | (eq & SEQ) | (~(lt | eq) & SGT);
                }
                if (!full) m &= ballot64(lrow < rem - srow);
                return m;
            };
            if constexpr (VEC == 1) {
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    const uint64_t m = row_mask(v[0][j], j * 64, lane);
                    const uint32_t rank = wave_total + mbcnt(m);
                    if (project && lane_of(m) && rank < cap) sv[rank] = v[0][j];
                    if constexpr (kSel) if (want_sel) sel_collect<1>(selw, j, m, 0, lane);
                    wave_total += static_cast<uint32_t>(__popcll(m));
                }
            } else {
#pragma unroll
                for (int j = 0; j < R / 2; ++j) {
                    const uint64_t m0 = row_mask(v[0][2 * j], j * 128, 2 * lane), m1 = row_mask(v[0][2 * j + 1], j * 128, 2 * lane + 1);
                    const bool p0 = lane_of(m0), p1 = lane_of(m1);
                    const uint32_t r0 = wave_total + mbcnt(m0) + mbcnt(m1), r1 = r0 + (p0 ? 1u : 0u);
                    if (project && p0 && r0 < cap) sv[r0] = v[0][2 * j];
                    if (project && p1 && r1 < cap) sv[r1] = v[0][2 * j + 1];
                    if constexpr (kSel) if (want_sel) sel_collect<2>(selw, 2 * j, m0, m1, lane);
                    wave_total += static_cast<uint32_t>(__popcll(m0) + __popcll(m1));
                }
            }
        } else {
            // ---- generic shape, mask-major: survive masks S[k] per row slot ------------------------------
            int32_t rem = 0;
            if (!full) {
                const int64_t left = static_cast<int64_t>(p.in.n) - static_cast<int64_t>(wave_base);
                rem = static_cast<int32_t>(left < 0 ? 0 : (left > (1 << 30) ? (1 << 30) : left));
            }
            uint64_t S[VEC == 2 ? R : 1];  // VEC == 2: survive masks per slot (SGPR pairs)
            uint64_t Sv = ~0ull;           // VEC == 1: the same in lane form (lane k = slot k)
            if constexpr (VEC == 2) {
#pragma unroll
                for (int k = 0; k < R; ++k) S[k] = full ? ~0ull : live_mask<VEC>(rem, k);
            } else if (!full) {
                const int32_t cnt = rem - 64 * lane;
                Sv = cnt <= 0 ? 0ull : low_mask(static_cast<uint64_t>(cnt > 64 ? 64 : cnt));
            }
            // validity: VEC == 1 reads a slot's mask out of the window lanes on demand; VEC == 2 has to
            // split window pairs per lane and keeps the masks
            uint64_t vwin[NV];
            bool hv[NV];
            uint64_t V2[VEC == 2 ? NV : 1][VEC == 2 ? R : 1];
#pragma unroll
            for (int c = 0; c < NV; ++c) {
                hv[c] = false;
                vwin[c] = ~0ull;
                if constexpr (kValidity) {
                    if (c < NCOLS && p.in.cols[c].validity) {
                        hv[c] = true;
                        vwin[c] = validity_windows(vw[c], uniform32(static_cast<uint32_t>((p.in.cols[c].offset + wave_base) & 63)));
                    }
                    if constexpr (VEC == 2) word_masks<R, 2>([&](int q) { return readlane64(vwin[c], q); }, lane, V2[c]);
                }
            }
            auto valid_mask = [&](int c, int k) -> uint64_t {
                if constexpr (!kValidity) return ~0ull;
                else if constexpr (VEC == 2) return V2[c][k];
                else return readlane64(vwin[c], k);
            };
            uint64_t Xv[kXs ? kMaxBitStreams : 1];  // VEC == 1: Boolean columns travelling with the rows, lane form
            if constexpr (VEC == 1) {
                // ---- lane form: every mask set is one 64-bit VGPR value --------------------------------------------------
                uint64_t bwin[kBool ? 2 * kMaxBoolCols : 1];
                if constexpr (kBool) {
#pragma unroll
                    for (int c = 0; c < kMaxBoolCols; ++c) {
                        const uint32_t sh = uniform32(static_cast<uint32_t>((p.in.bcols[c].offset + wave_base) & 63));
                        bwin[2 * c] = validity_windows(bw[2 * c], sh);
                        bwin[2 * c + 1] = validity_windows(bw[2 * c + 1], sh);
                    }
                }
                // truth of term `term` on every slot, null rows at null_v (the AnyValue table is lowered on the host)
                auto term_truth = [&](const DevTerm &term) -> uint64_t {
                    uint64_t X = 0;
                    if (!term.is_bool()) {
#pragma unroll
                        for (int c = 0; c < NCOLS; ++c)
                            if (term.slot() == static_cast<uint32_t>(c)) {
                                X = compare_lanes<R>(term, v[c]);
                                if (kValidity && hv[c]) X = term.null_v() ? (X | ~vwin[c]) : (X & vwin[c]);
                            }
                    } else if constexpr (kBool) {
                        const BoolCoef coef = bool_coef(term);
#pragma unroll
                        for (int c = 0; c < kMaxBoolCols; ++c)
                            if (term.slot() == static_cast<uint32_t>(c)) X = eval_bool_word(coef, bwin[2 * c], bwin[2 * c + 1]);
                    }
                    return X;
                };
                if constexpr (kExpr) {  // conjunctive normal form with negated literals (device_common.hpp, DevTerm)
                    uint64_t A = ~0ull, G = 0;
                    for (int t = 0; t < nterms; ++t) {
                        const DevTerm term = term_at(t);
                        const uint64_t X = term_truth(term);
                        G |= term.negate() ? ~X : X;
                        if (term.group_end()) {
                            A &= G;
                            G = 0;
                        }
                    }
                    // strict null propagation (BooleanArray::and / or / not, boolean.rs:120-165)
                    if constexpr (kValidity) {
#pragma unroll
                        for (int c = 0; c < NCOLS; ++c)
                            if (hv[c] && ((p.in.strict_values >> c) & 1)) Sv &= vwin[c];
                    }
                    if constexpr (kBool) {
#pragma unroll
                        for (int c = 0; c < kMaxBoolCols; ++c)
                            if (((p.in.strict_bools >> c) & 1) && p.in.bcols[c].validity) Sv &= bwin[2 * c + 1];
                    }
                    Sv &= p.in.negate_result ? ~A : A;
                } else {
                    for (int t = 0; t < nterms; ++t) Sv &= term_truth(term_at(t));
                }
                if constexpr (kXs) {
#pragma unroll
                    for (int s2 = 0; s2 < kMaxBitStreams; ++s2) {
                        Xv[s2] = 0;
                        if (s2 >= p.nxs) continue;
                        const uint32_t sh = uniform32(static_cast<uint32_t>((p.xs[s2].offset + wave_base) & 63));
                        Xv[s2] = validity_windows(xw[2 * s2], sh) & validity_windows(xw[2 * s2 + 1], sh);
                    }
                }
            } else
            if constexpr (kExpr) {
                // ---- OR / NOT: the literals of a conjunctive normal form, in order (device_common.hpp, DevTerm) ----
                uint64_t bwin[kBool ? 2 * kMaxBoolCols : 1];
                if constexpr (kBool) {
#pragma unroll
                    for (int c = 0; c < kMaxBoolCols; ++c) {
                        const uint32_t sh = uniform32(static_cast<uint32_t>((p.in.bcols[c].offset + wave_base) & 63));
                        bwin[2 * c] = validity_windows(bw[2 * c], sh);
                        bwin[2 * c + 1] = validity_windows(bw[2 * c + 1], sh);
                    }
                }
                uint64_t A[R], G[R];
#pragma unroll
                for (int k = 0; k < R; ++k) A[k] = ~0ull, G[k] = 0;
                for (int t = 0; t < nterms; ++t) {
                    const DevTerm term = term_at(t);
                    uint64_t X[R];
#pragma unroll
                    for (int k = 0; k < R; ++k) X[k] = 0;
                    if (!term.is_bool()) {
#pragma unroll
                        for (int c = 0; c < NCOLS; ++c)
                            if (term.slot() == static_cast<uint32_t>(c))
                                value_term_truth<R>(term, v[c], [&](int k) { return valid_mask(c, k); }, kValidity && hv[c], X);
                    } else if constexpr (kBool) {
                        const BoolCoef coef = bool_coef(term);
#pragma unroll
                        for (int c = 0; c < kMaxBoolCols; ++c)
                            if (term.slot() == static_cast<uint32_t>(c))
                                word_masks<R, VEC>([&](int q) { return eval_bool_word(coef, readlane64(bwin[2 * c], q), readlane64(bwin[2 * c + 1], q)); }, lane, X);
                    }
                    const uint64_t flip = term.negate() ? ~0ull : 0ull;
                    const bool close = term.group_end();
#pragma unroll
                    for (int k = 0; k < R; ++k) {
                        G[k] |= X[k] ^ flip;
                        if (close) {
                            A[k] &= G[k];
                            G[k] = 0;
                        }
                    }
                }
                // strict null propagation (BooleanArray::and / or / not, boolean.rs:120-165): rows where a cell the
                // expression reads is null do not survive, whatever the literals said there
                if constexpr (kValidity) {
#pragma unroll
                    for (int c = 0; c < NCOLS; ++c)
                        if (hv[c] && ((p.in.strict_values >> c) & 1)) {
#pragma unroll
                            for (int k = 0; k < R; ++k) S[k] &= valid_mask(c, k);
                        }
                }
                if constexpr (kBool) {
#pragma unroll
                    for (int c = 0; c < kMaxBoolCols; ++c)
                        if (((p.in.strict_bools >> c) & 1) && p.in.bcols[c].validity) {
                            uint64_t M[R];
                            word_masks<R, VEC>([&](int q) { return readlane64(bwin[2 * c + 1], q); }, lane, M);
#pragma unroll
                            for (int k = 0; k < R; ++k) S[k] &= M[k];
                        }
                }
                const uint64_t rflip = p.in.negate_result ? ~0ull : 0ull;
#pragma unroll
                for (int k = 0; k < R; ++k) S[k] &= A[k] ^ rflip;
            } else {
            for (int t = 0; t < nterms; ++t) {
                const DevTerm term = term_at(t);
                if (term.is_bool()) continue;
#pragma unroll
                for (int c = 0; c < NCOLS; ++c)
                    if (term.slot() == static_cast<uint32_t>(c))
                        and_value_term<R>(term, v[c], [&](int k) { return valid_mask(c, k); }, kValidity && hv[c], S);
            }
            }
            if constexpr (kBool && !kExpr && VEC == 2) {
                // windows of the prefetched words: lane q = bits [64q, 64q + 64) of the wave's range
                uint64_t bwin[2 * kMaxBoolCols];
#pragma unroll
                for (int c = 0; c < kMaxBoolCols; ++c) {
                    const uint32_t sh = uniform32(static_cast<uint32_t>((p.in.bcols[c].offset + wave_base) & 63));
                    bwin[2 * c] = validity_windows(bw[2 * c], sh);
                    bwin[2 * c + 1] = validity_windows(bw[2 * c + 1], sh);
                }
                for (int t = 0; t < nterms; ++t) {
                    const DevTerm term = term_at(t);
                    if (!term.is_bool()) continue;
                    uint64_t B[R];
                    const BoolCoef coef = bool_coef(term);
#pragma unroll
                    for (int c = 0; c < kMaxBoolCols; ++c) {
                        if (term.slot() != static_cast<uint32_t>(c)) continue;
                        word_masks<R, VEC>([&](int q) { return eval_bool_word(coef, readlane64(bwin[2 * c], q), readlane64(bwin[2 * c + 1], q)); }, lane, B);
#pragma unroll
                        for (int k = 0; k < R; ++k) S[k] &= B[k];
                    }
                }
            }
            // Boolean columns travelling with the rows
            uint64_t X[(kXs && VEC == 2) ? kMaxBitStreams : 1][(kXs && VEC == 2) ? R : 1];
            if constexpr (kXs && VEC == 2) {
#pragma unroll
                for (int s2 = 0; s2 < kMaxBitStreams; ++s2) {
                    if (s2 >= p.nxs) continue;
                    const uint32_t sh = uniform32(static_cast<uint32_t>((p.xs[s2].offset + wave_base) & 63));
                    const uint64_t ws = validity_windows(xw[2 * s2], sh), wm = validity_windows(xw[2 * s2 + 1], sh);
                    word_masks<R, VEC>([&](int q) { return readlane64(ws, q) & readlane64(wm, q); }, lane, X[s2]);
                }
            }

            // ---- rank + stage, slot by slot (slot order == row order) ---------------------------------
            if constexpr (kStamp) tm = stamp_now();
            // rows of slot k whose lanes are `m`, ranks `rank`: value (placeholder 0 under a null,
            // record_batch.rs:142-146), validity byte, bit streams -> this wave's LDS slot
            // Branch-free on purpose: with one basic block for all R slots the scheduler overlaps the
            // slots' dependency chains (v_readlane -> v_cndmask -> ds_write); with an exec-masked block per
            // slot every wave ran them one after the other and the loop was latency bound.
            const uint32_t dump = kLdsHeader + nstages * WAVES * slot_bytes + wave * kLdsDumpBytes;
            const uint32_t dump_v = dump + lane * 8u, dump_b = dump + 512u + lane;
            auto stage_slot = [&](int k, uint64_t m, uint32_t rank) {
                const bool keep = lane_of(m) && rank < cap;
#pragma unroll
                for (int c = 0; c < NCOLS; ++c) {
                    if constexpr (!kAll)
                        if (!(outflags & (1u << c))) continue;
                    const uint32_t av = keep ? sb + off_v[c] + rank * 8u : dump_v;
                    if constexpr (kValidity && !kNoNull) {
                        const bool valid = lane_of(valid_mask(c, k));
                        *reinterpret_cast<uint64_t *>(smem + av) = valid ? v[c][k] : 0;
                        if (kAll || (outflags & (0x100u << c))) smem[keep ? sb + off_b[c] + rank : dump_b] = valid;
                    } else {
                        *reinterpret_cast<uint64_t *>(smem + av) = v[c][k];
                    }
                }
                if constexpr (kXs && VEC == 2) {
#pragma unroll
                    for (int s2 = 0; s2 < kMaxBitStreams; ++s2)
                        if (s2 < p.nxs) smem[keep ? sb + off_x[s2] + rank : dump_b] = lane_of(X[s2][k]);
                }
            };
            if constexpr (kXs && VEC == 1) {
                // Boolean columns travelling with the rows, in lane form: lane q holds 64 rows of the column and their
                // survive mask, so the surviving bits are a software PEXT per lane (few rows survive), dropped at the
                // lane's bit position in the wave's run with LDS atomics -- 2 bits read per row, no per-row staging.
                const uint64_t selq = lane < R ? Sv : 0ull;
                const uint32_t cntq = static_cast<uint32_t>(__popcll(selq));
                uint32_t incl = cntq;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t y = __shfl_up(incl, d, 64);
                    if (lane >= d) incl += y;
                }
                const uint32_t pos = incl - cntq, wq = pos >> 6, sh = pos & 63;
#pragma unroll
                for (int s2 = 0; s2 < kMaxBitStreams; ++s2) {
                    if (s2 >= p.nxs) continue;
                    uint64_t *words = reinterpret_cast<uint64_t *>(smem + sb + off_x[s2]);
                    if (lane <= R) words[lane] = 0;  // R + 1 words: every bit a wave can stage
                    const uint64_t x = Xv[s2];
                    uint64_t c = 0, m = selq;
                    for (int j = 0; m; ++j, m &= m - 1) c |= ((x >> __builtin_ctzll(m)) & 1ull) << j;
                    if (c) {
                        atomicOr(reinterpret_cast<unsigned long long *>(&words[wq]), static_cast<unsigned long long>(c << sh));
                        if (sh && (c >> (64 - sh))) atomicOr(reinterpret_cast<unsigned long long *>(&words[wq + 1]), static_cast<unsigned long long>(c >> (64 - sh)));
                    }
                }
            }
            if constexpr (VEC == 1) {
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    const uint64_t m = readlane64(Sv, k);
                    stage_slot(k, m, wave_total + mbcnt(m));
                    wave_total += static_cast<uint32_t>(__popcll(m));
                }
                if constexpr (kSel) selw = Sv;  // slot k IS selection word k
            } else {
#pragma unroll
                for (int j = 0; j < R / 2; ++j) {
                    const uint64_t m0 = S[2 * j], m1 = S[2 * j + 1];
                    const uint32_t r0 = wave_total + mbcnt(m0) + mbcnt(m1);
                    stage_slot(2 * j, m0, r0);
                    stage_slot(2 * j + 1, m1, r0 + (lane_of(m0) ? 1u : 0u));
                    wave_total += static_cast<uint32_t>(__popcll(m0) + __popcll(m1));
                    if constexpr (kSel) if (want_sel) sel_collect<2>(selw, 2 * j, m0, m1, lane);
                }
            }
        }
        if constexpr (kStamp && !kOne) st_stage += stamp_now() - tm;
        if constexpr (kSel)
            if (want_sel) sel_store<R>(selw, p.out_selection, wave_base, p.in.n, lane);
        wave_total = uniform32(wave_total);

        // ---- stage the survivors in this wave's slot; free the registers; prefetch ------------------------
        const bool wave_dense = wave_total > cap;  // wave-uniform: the tile goes to the redo kernel
        // the rows are staged (or given up): their registers are free -> prefetch the next tile.  (Wave 0's
        // descriptor load for the offset lookup went out at the top of the iteration, ahead of these.)
        // the descriptor wave 0 loaded at the top of the iteration goes into scalar registers HERE, where every older
        // load has landed anyway (the compares needed the rows): read after the prefetch below it would wait for the
        // whole prefetch (vmcnt counts in order), and the other fifteen waves wait for wave 0 at barrier B
        const uint64_t prev_now = uniform64(prev_desc);
        if (more) {
            load_rows<NCOLS, R, VEC>(p.in, ntb + static_cast<uint64_t>(wave) * ROWS_PER_WAVE, lane, v);
            prefetch_bits(ntb + static_cast<uint64_t>(wave) * ROWS_PER_WAVE);
        }
        if (lane == 0) s_wtot[wave] = wave_total | (wave_dense ? 0x80000000u : 0u);
        if constexpr (kStamp) {
            t1 = stamp_now();
            st_eval += t1 - t0;
            t0 = t1;
        }
        __syncthreads();  // A: wave totals visible
        if constexpr (kStamp) {
            t1 = stamp_now();
            st_scatter += t1 - t0;
            t0 = t1;
        }

        uint32_t wave_prefix = 0, tile_count = 0;
        bool any_dense = false;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            const uint32_t t = s_wtot[w];
            wave_prefix += (static_cast<uint32_t>(w) < wave) ? (t & 0x7FFFFFFFu) : 0;
            tile_count += t & 0x7FFFFFFFu;
            any_dense |= (t >> 31) != 0;
        }
        wave_prefix = uniform32(wave_prefix);
        tile_count = uniform32(tile_count);
        // successors can sum this tile's count from here on
        if (threadIdx.x == 0 && !(kStamp && (p.debug & 16) && tile == 1)) publish_aggregate(p.state, tile, tile_count);

        // ---- write out one iteration later: the pending tile's offset, then its slots ----------------
        if (wave == 0 && ret.have) resolve(ret, prev_now);
        if constexpr (kStamp) {
            t1 = stamp_now();
            st_look += t1 - t0;
            t0 = t1;
        }
        __syncthreads();  // B: the pending tile's offset is known; s_wtot may be rewritten
        if constexpr (kStamp) {
            t1 = stamp_now();
            st_waitB += t1 - t0;
            t0 = t1;
        }
        if (ret.have && !ret.dense && ret.wave_total) flush(slot_of(ret.stage), ret.wave_total, uniform64(*s_excl) + ret.wave_prefix);
        older = newer;
        newer = Pending{tile, tile_count, cur_stage, wave_prefix, wave_total, any_dense, true};
        cur_stage = cur_stage + 1 == nstages ? 0 : cur_stage + 1;
        if constexpr (kStamp) {
            t1 = stamp_now();
            st_flush += t1 - t0;
            st_tiles += 1;
        }
        if (threadIdx.x == 0) s_tick[(it + 3) & 3] = ticket3;
        tile = next_tile;
    }

    // ---- epilogue: the tiles still staged -----------------------------------------------------------
    auto retire = [&](const Pending &r) {
        if (!r.have) return;  // workgroup-uniform
        if (wave == 0) {
            if (r.tile != 0) prev_desc = ld_state(&p.state[r.tile - 1]);
            resolve(r, uniform64(prev_desc));
        }
        __syncthreads();
        if (!r.dense && r.wave_total) flush(slot_of(r.stage), r.wave_total, uniform64(*s_excl) + r.wave_prefix);
        __syncthreads();  // s_excl may be rewritten
    };
    if (deep) retire(older);
    retire(newer);

    if constexpr (kStamp) {
        if (lane == 0 && (wave == 0 || wave == 1)) {  // wave 0 runs the look-back, wave 1 waits for it
            unsigned long long *d = p.stamps + (wave ? 8 : 0);
            atomicAdd(&d[0], st_eval);
            atomicAdd(&d[1], st_scatter);
            atomicAdd(&d[2], st_look);
            atomicAdd(&d[3], st_waitB);
            atomicAdd(&d[4], st_flush);
            atomicAdd(&d[5], st_tiles);
            if (wave == 1) {
                atomicAdd(&d[6], st_wait);
                atomicAdd(&d[7], st_stage);
            }
        }
    }

    // ---- set-bit counts of the compacted bit streams (null counts on the host side) ---------------
    if constexpr (kValidity || kXs) {
        __syncthreads();
        const uint32_t *s_pop = reinterpret_cast<const uint32_t *>(smem + 24);
        if (threadIdx.x < kMaxValueCols + kMaxBitStreams && s_pop[threadIdx.x])
            atomicAdd(&p.out_valid_pop[threadIdx.x], static_cast<unsigned long long>(s_pop[threadIdx.x]));
    }
}

// ---- redo kernel ----------------------------------------------------------------------------------
// Tiles in which some wave had more survivors than its LDS slot (dense data).  The main kernel has
// already counted them and reserved their output range; here one workgroup re-reads a tile in
// blocks of 2048 rows, ranks across the whole workgroup and writes block after block at the known
// offset.  Generic (every feature), simple, correct for any selectivity; costs one extra read of
// the tiles on the list.
template <int NCOLS>
__global__ __launch_bounds__(1024) void fused_redo_tiles(const FusedParams p, uint32_t tile_rows) {
    constexpr int NV = NCOLS > 0 ? NCOLS : 1;
    constexpr int RR = 2, WAVES = 16;
    constexpr uint32_t BLOCK = 64u * RR * WAVES;  // 2048 rows
    unsigned char *const smem = rv_smem;
    uint32_t *s_wtot = reinterpret_cast<uint32_t *>(smem + 56);
    const int lane = lane_id();
    const uint32_t wave = uniform32(threadIdx.x >> 6);
    if (threadIdx.x < 8) reinterpret_cast<uint32_t *>(smem + 24)[threadIdx.x] = 0;

    uint32_t off_v[NV], off_b[NV], off_x[kMaxBitStreams];
    {
        uint32_t cur = kLdsHeader;
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            off_v[c] = cur;
            if (c < NCOLS && p.out_values[c]) cur += BLOCK * 8;
        }
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            off_b[c] = cur;
            if (c < NCOLS && p.out_validity[c]) cur += BLOCK;
        }
#pragma unroll
        for (int s = 0; s < kMaxBitStreams; ++s) {
            off_x[s] = cur;
            if (s < p.nxs) cur += BLOCK;
        }
    }
    __syncthreads();
    const uint32_t nredo = *p.redo_count;
    for (uint32_t k = blockIdx.x; k < nredo; k += gridDim.x) {
        const uint64_t tile_base = p.redo[2 * k] * tile_rows;
        uint64_t g0 = p.redo[2 * k + 1];
        for (uint32_t sub = 0; sub < tile_rows; sub += BLOCK) {
            const uint64_t wave_base = tile_base + sub + static_cast<uint64_t>(wave) * (64u * RR);
            const bool full = tile_base + sub + BLOCK <= p.in.n;
            uint64_t v[NV][RR];
            uint32_t vb[NV], pb;
            scan_rows<NCOLS, RR, 1, FF_ALL>(p.in, wave_base, full, lane, v, vb, pb);
            uint32_t xb[kMaxBitStreams];
#pragma unroll
            for (int s = 0; s < kMaxBitStreams; ++s) {
                xb[s] = 0;
                if (s < p.nxs) {
                    const BitStream bs = p.xs[s];
                    xb[s] = gather_row_bits<RR, 1>(
                        [&](int q) {
                            const uint64_t pos = bs.offset + wave_base + q * 64u;
                            uint64_t w = load_bits64(bs.src, pos, bs.src_bytes);
                            if (bs.mask) w &= load_bits64(bs.mask, pos, bs.mask_bytes);
                            return w;
                        },
                        lane);
                }
            }
            uint32_t wave_total = 0;
#pragma unroll
            for (int j = 0; j < RR; ++j) wave_total += static_cast<uint32_t>(__popcll(ballot64((pb >> j) & 1)));
            if (lane == 0) s_wtot[wave] = wave_total;
            __syncthreads();
            uint32_t wave_prefix = 0, count = 0;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) {
                const uint32_t t = s_wtot[w];
                wave_prefix += (static_cast<uint32_t>(w) < wave) ? t : 0;
                count += t;
            }
            wave_prefix = uniform32(wave_prefix);
            count = uniform32(count);
            // stage at the in-block rank (a block's survivors always fit: BLOCK rows of stage)
#pragma unroll
            for (int c = 0; c < NCOLS; ++c) {
                if (!p.out_values[c]) continue;
                uint64_t *sv = reinterpret_cast<uint64_t *>(smem + off_v[c]);
                const bool hv = p.out_validity[c] != nullptr;
                uint32_t running = wave_prefix;
#pragma unroll
                for (int j = 0; j < RR; ++j) {
                    const bool pj = (pb >> j) & 1;
                    const uint64_t m = ballot64(pj);
                    const uint32_t rank = running + mbcnt(m);
                    if (pj) {
                        const bool valid = (vb[c] >> j) & 1;
                        sv[rank] = valid ? v[c][j] : 0;
                        if (hv) smem[off_b[c] + rank] = valid;
                    }
                    running += static_cast<uint32_t>(__popcll(m));
                }
            }
#pragma unroll
            for (int s = 0; s < kMaxBitStreams; ++s)
                if (s < p.nxs) {
                    uint32_t running = wave_prefix;
#pragma unroll
                    for (int j = 0; j < RR; ++j) {
                        const bool pj = (pb >> j) & 1;
                        const uint64_t m = ballot64(pj);
                        if (pj) smem[off_x[s] + running + mbcnt(m)] = (xb[s] >> j) & 1;
                        running += static_cast<uint32_t>(__popcll(m));
                    }
                }
            __syncthreads();
            const bool fits = g0 + count <= p.out_capacity;  // workgroup-uniform
            if (!fits && threadIdx.x == 0) *p.overflow = 1u;
#pragma unroll
            for (int c = 0; c < NCOLS; ++c) {
                if (!p.out_values[c] || !fits) continue;
                const uint64_t *sv = reinterpret_cast<const uint64_t *>(smem + off_v[c]);
                for (uint32_t i = threadIdx.x; i < count; i += 1024) p.out_values[c][g0 + i] = sv[i];
                if (p.out_validity[c] && wave == 0) flush_bits(off_b[c], count, g0, p.out_validity[c], 24 + 4 * c);
            }
#pragma unroll
            for (int s = 0; s < kMaxBitStreams; ++s)
                if (s < p.nxs && wave == 0 && fits) flush_bits(off_x[s], count, g0, p.xs[s].out, 24 + 4 * (kMaxValueCols + s));
            g0 += count;
            __syncthreads();  // stage and s_wtot are reused by the next block
        }
    }
    __syncthreads();
    const uint32_t *s_pop = reinterpret_cast<const uint32_t *>(smem + 24);
    if (threadIdx.x < kMaxValueCols + kMaxBitStreams && s_pop[threadIdx.x])
        atomicAdd(&p.out_valid_pop[threadIdx.x], static_cast<unsigned long long>(s_pop[threadIdx.x]));
}

}  // namespace rvk
