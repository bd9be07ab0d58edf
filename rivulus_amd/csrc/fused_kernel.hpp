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
//      tickets are drawn two iterations ahead;
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
    FF_ALL = 15,
    FF_STAMP = 16     // diagnostic build only: per-phase s_memtime sums (never timed, never shipped to callers)
};
enum : int { FF_ONE_I64 = 0, FF_ONE_F64 = 0 };  // predicate-shape specialisations: not used by this kernel version

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
    unsigned long long *stamps;             // FF_STAMP builds: [8] cycle sums + tile count
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

// Decoupled look-back, executed by one full wave.  The tile's aggregate has already been
// published (publish_aggregate, as early as the count is known).  lookback_finish returns the
// exclusive prefix of `tile` (wave-uniform) and publishes the tile's inclusive prefix.
//
// One poll inspects kLookK * 64 predecessors (kLookK descriptors per lane, all loads in
// flight together).  The window has to cover every tile that can be in flight at once:
// with a 64-wide window the prefix front advances only 64 tiles per poll round trip
// (~2 us), which capped the whole kernel at ~27 tiles/us (profiles/README.md, r01a).
// Inlined on purpose: an out-of-line call makes every value that lives across it sit in
// callee-saved VGPRs (every other block of 8 on gfx950), which doubled the kernel's footprint.
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

// Completes the look-back of `tile` (> 0).  `s` holds a poll issued earlier with
// base = tile - 1 (its round trip overlapped other work); further polls are issued here
// only if a needed descriptor was not there yet or the window held no prefix.
__device__ __forceinline__ uint64_t lookback_finish(uint64_t *state, uint32_t tile, uint64_t aggregate, uint32_t *err,
                                                    uint64_t (&s)[kLookK], unsigned long long *poll_stats = nullptr) {
    const int lane = lane_id();
    uint32_t polls = 1, windows = 0;
    uint64_t excl = 0;
    int64_t base = static_cast<int64_t>(tile) - 1;
    uint32_t spins = 0;
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
            if (++spins > kSpinLimit) {
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
typedef unsigned int rv_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int rv_u32x2 __attribute__((ext_vector_type(2)));

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
                const rv_u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rsrc, lane * 8, j * 512, 0);
                v[c][j] = (static_cast<uint64_t>(t.y) << 32) | t.x;
            }
        } else {
#pragma unroll
            for (int j = 0; j < R / 2; ++j) {
                const rv_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, j * 1024, 0);
                v[c][2 * j] = (static_cast<uint64_t>(t.y) << 32) | t.x;
                v[c][2 * j + 1] = (static_cast<uint64_t>(t.w) << 32) | t.z;
            }
        }
    }
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
static __device__ __attribute__((noinline)) void flush_bits(uint32_t stage_off, uint32_t cnt, uint64_t g0, uint64_t *out,
                                                     uint32_t pop_off) {
    if (cnt == 0) return;
    const uint8_t *stage = rv_smem + stage_off;
    const int lane = lane_id();
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

constexpr int kLdsHeader = 128;

__device__ __forceinline__ unsigned long long stamp_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

// Persistent and software-pipelined.  A workgroup keeps drawing tiles until the ticket
// counter runs out.  Every wave owns a contiguous run of 64*R rows of the tile and a PRIVATE
// LDS slot (two of them: double buffered), so the per-tile critical path is short:
//   wave:  wait rows -> predicate -> ballot/mbcnt ranks -> survivors into the own slot
//          -> issue the NEXT tile's row loads at once (the registers are dead)
//          -> one barrier to exchange the 16 wave counts
//   later: each wave writes its own slot to out[tile offset + wave prefix ...] (coalesced run)
// and three latencies never sit on it:
//   ticket      drawn three iterations ahead;
//   row loads   of the next tile are in flight across the barrier, the look-back and the flush;
//   look-back   DEFERRED by one iteration: wave 0 issues the descriptor poll at the top of the
//               next iteration and consumes it after that iteration's predicate + staging
//               (under streaming load one poll round trip is ~4.6 us: profiles/README.md).
// A wave with more survivors than its slot holds (dense data) keeps its rows in registers and
// the workgroup resolves that tile at once, in rounds (slow path, any selectivity is correct).
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
    constexpr bool kStamp = (FLAGS & FF_STAMP) != 0;
    unsigned long long st_eval = 0, st_scatter = 0, st_look = 0, st_waitB = 0, st_flush = 0, st_tiles = 0, t0 = 0, t1 = 0;

    unsigned char *const smem = rv_smem;
    uint32_t *s_tick = reinterpret_cast<uint32_t *>(smem);       // [4] ring of tile ids, drawn three iterations ahead
    uint64_t *s_excl = reinterpret_cast<uint64_t *>(smem + 16);
    uint32_t *s_wtot = reinterpret_cast<uint32_t *>(smem + 56);  // [WAVES <= 16]
    // s_pop: 8 x uint32 at smem + 24 (set-bit counts of compacted bit streams)

    const int lane = lane_id();
    const uint32_t wave = uniform32(threadIdx.x >> 6);

    // ---- LDS carve: slot(stage, wave) of `cap` rows; byte offsets of the columns inside a slot ----
    const uint32_t cap = p.cap_rows;  // rows per wave slot
    uint32_t off_v[NV], off_b[NV], off_x[kMaxBitStreams];
    uint32_t slot_bytes;
    {
        uint32_t cur = 0;
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
        slot_bytes = (cur + 15u) & ~15u;
    }
    auto slot_of = [&](uint32_t stage) { return kLdsHeader + (stage * WAVES + wave) * slot_bytes; };

    // this wave's survivors with in-wave rank in [lo, lo + cap) -> slot at byte offset `sb`
    auto scatter = [&](uint32_t sb, uint32_t pb, const uint32_t (&vb)[NV], const uint32_t (&xb)[kMaxBitStreams],
                       const uint64_t (&v)[NV][R], uint32_t lo) {
        const uint32_t hi = lo + cap;
#pragma unroll
        for (int c = 0; c < NCOLS; ++c) {
            if (!p.out_values[c]) continue;
            uint64_t *sv = reinterpret_cast<uint64_t *>(smem + sb + off_v[c]);
            if constexpr (kValidity) {
                uint8_t *sbits = smem + sb + off_b[c];
                const bool hv = p.out_validity[c] != nullptr;
                for_each_survivor<R, VEC>(pb, lo, hi, [&](int k, uint32_t pos) {
                    const bool valid = (vb[c] >> k) & 1;
                    sv[pos] = valid ? v[c][k] : 0;  // placeholder 0 / 0.0 (record_batch.rs:142-146)
                    if (hv) sbits[pos] = valid;
                });
            } else {
                for_each_survivor<R, VEC>(pb, lo, hi, [&](int k, uint32_t pos) { sv[pos] = v[c][k]; });
            }
        }
        if constexpr (kXs) {
#pragma unroll
            for (int s = 0; s < kMaxBitStreams; ++s)
                if (s < p.nxs) {
                    uint8_t *sx = smem + sb + off_x[s];
                    for_each_survivor<R, VEC>(pb, lo, hi, [&](int k, uint32_t pos) { sx[pos] = (xb[s] >> k) & 1; });
                }
        }
    };
    // this wave's slot rows [0, cnt) -> out[g0 ...): one coalesced run per column
    auto flush = [&](uint32_t sb, uint32_t cnt, uint64_t g0) {
#pragma unroll
        for (int c = 0; c < NCOLS; ++c) {
            if (!p.out_values[c]) continue;
            uint64_t *dst = p.out_values[c] + g0;
            const uint64_t *sv = reinterpret_cast<const uint64_t *>(smem + sb + off_v[c]);
            for (uint32_t k = lane; k < cnt; k += 64) dst[k] = sv[k];
            if constexpr (kValidity)
                if (p.out_validity[c]) flush_bits(sb + off_b[c], cnt, g0, p.out_validity[c], 24 + 4 * c);
        }
        if constexpr (kXs) {
#pragma unroll
            for (int s = 0; s < kMaxBitStreams; ++s)
                if (s < p.nxs) flush_bits(sb + off_x[s], cnt, g0, p.xs[s].out, 24 + 4 * (kMaxValueCols + s));
        }
    };

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

    // the tile whose survivors wait in the slots for their output offset
    bool have_prev = false;
    uint32_t prev_tile = 0, prev_count = 0, prev_stage = 0, cur_stage = 0;
    uint32_t prev_wave_prefix = 0, prev_wave_total = 0;
    uint64_t poll[kLookK];  // wave 0: descriptors of the pending tile's look-back, in flight

    // wave 0: finish the pending look-back, hand the offset to the workgroup
    auto resolve_prev = [&]() {
        uint64_t e = 0;
        if (prev_tile != 0) e = lookback_finish(p.state, prev_tile, prev_count, p.err, poll, kStamp ? p.stamps + 6 : nullptr);
        if (lane == 0) {
            *s_excl = e;
            if (prev_tile == p.ntiles - 1) *p.out_count = e + prev_count;
        }
    };

    for (uint32_t it = 0; tile < p.ntiles; ++it) {
        // ---- ticket for three iterations ahead: issued now, stored at the end of the iteration and read
        //      at the top of iteration it + 2 (barrier A of it + 1 lies in between) --------------------
        uint32_t ticket3 = 0;
        if (threadIdx.x == 0) ticket3 = __hip_atomic_fetch_add(p.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // ---- poll for the pending tile: its round trip runs under this tile's predicate + staging ----
        if (wave == 0 && have_prev && prev_tile != 0) lookback_issue(p.state, static_cast<int64_t>(prev_tile) - 1, lane, poll);

        const uint32_t next_tile = uniform32(s_tick[(it + 1) & 3]);  // stored two iterations ago
        const uint64_t tile_base = static_cast<uint64_t>(tile) * TILE;
        const uint64_t wave_base = tile_base + static_cast<uint64_t>(wave) * ROWS_PER_WAVE;
        const bool full = tile_base + TILE <= p.in.n;
        const uint64_t ntb = static_cast<uint64_t>(next_tile) * TILE;
        const bool more = next_tile < p.ntiles;

        // ---- validity bits, predicate -> survive bits ------------------------------------------------------
        uint32_t vb[NV];
        uint32_t pb;
        if constexpr (kStamp) t0 = stamp_now();
        eval_rows<NCOLS, R, VEC, FLAGS>(p.in, wave_base, full, lane, v, vb, pb);

        // extra bit streams (Boolean columns travelling with the rows)
        uint32_t xb[kMaxBitStreams];
#pragma unroll
        for (int s = 0; s < kMaxBitStreams; ++s) xb[s] = 0;
        if constexpr (kXs) {
#pragma unroll
            for (int s = 0; s < kMaxBitStreams; ++s) {
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

        // ---- selection bitmap (optional) + per-wave survivor count ----------------------------------------
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
        wave_total = uniform32(wave_total);

        // ---- stage the survivors in this wave's slot; free the registers; prefetch ------------------------
        const bool wave_dense = wave_total > cap;  // wave-uniform
        const uint32_t sb = slot_of(cur_stage);
        if (wave_total) scatter(sb, pb, vb, xb, v, 0);
        // wave 0 must consume its poll before queueing row loads behind it (results return in issue order)
        if (more && !wave_dense && wave != 0) load_rows<NCOLS, R, VEC>(p.in, ntb + static_cast<uint64_t>(wave) * ROWS_PER_WAVE, lane, v);
        if (lane == 0) s_wtot[wave] = wave_total;
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
            wave_prefix += (static_cast<uint32_t>(w) < wave) ? t : 0;
            tile_count += t;
            any_dense |= t > cap;
        }
        wave_prefix = uniform32(wave_prefix);
        tile_count = uniform32(tile_count);
        // successors can sum this tile's count from here on
        if (threadIdx.x == 0) publish_aggregate(p.state, tile, tile_count);

        if (!any_dense) {
            // ================= common case: write out one iteration later =================
            if (wave == 0) {
                if (have_prev) resolve_prev();
                if (more) load_rows<NCOLS, R, VEC>(p.in, ntb, lane, v);
            }
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
            if (have_prev && prev_wave_total) flush(slot_of(prev_stage), prev_wave_total, uniform64(*s_excl) + prev_wave_prefix);
            prev_tile = tile;
            prev_count = tile_count;
            prev_stage = cur_stage;
            prev_wave_prefix = wave_prefix;
            prev_wave_total = wave_total;
            cur_stage ^= 1;
            have_prev = true;
        } else {
            // ================= dense tile: some wave has more survivors than its slot holds =================
            if (have_prev) {  // drain the pending tile first
                if (wave == 0) resolve_prev();
                __syncthreads();
                if (prev_wave_total) flush(slot_of(prev_stage), prev_wave_total, uniform64(*s_excl) + prev_wave_prefix);
                have_prev = false;
                __syncthreads();  // s_excl is rewritten below
            }
            if (wave == 0) {
                uint64_t e = 0;
                if (tile != 0) {
                    lookback_issue(p.state, static_cast<int64_t>(tile) - 1, lane, poll);
                    e = lookback_finish(p.state, tile, tile_count, p.err, poll);
                }
                if (lane == 0) {
                    *s_excl = e;
                    if (tile == p.ntiles - 1) *p.out_count = e + tile_count;
                }
            }
            __syncthreads();
            const uint64_t g0 = uniform64(*s_excl) + wave_prefix;
            // every wave writes its own run; a dense wave goes round by round out of its registers
            for (uint32_t lo = 0; lo < wave_total; lo += cap) {
                if (lo) scatter(sb, pb, vb, xb, v, lo);  // round 0 was staged above
                flush(sb, wave_total - lo < cap ? wave_total - lo : cap, g0 + lo);
            }
            if (more && (wave_dense || wave == 0)) load_rows<NCOLS, R, VEC>(p.in, ntb + static_cast<uint64_t>(wave) * ROWS_PER_WAVE, lane, v);
        }
        if constexpr (kStamp) {
            t1 = stamp_now();
            st_flush += t1 - t0;
            st_tiles += 1;
        }
        if (threadIdx.x == 0) s_tick[(it + 3) & 3] = ticket3;
        tile = next_tile;
    }

    // ---- epilogue: the last staged tile ----------------------------------------------------------------
    if (have_prev) {
        if (wave == 0) {
            if (prev_tile != 0) lookback_issue(p.state, static_cast<int64_t>(prev_tile) - 1, lane, poll);
            resolve_prev();
        }
        __syncthreads();
        if (prev_wave_total) flush(slot_of(prev_stage), prev_wave_total, uniform64(*s_excl) + prev_wave_prefix);
    }

    if constexpr (kStamp) {
        if (lane == 0 && (wave == 0 || wave == 1)) {  // wave 0 runs the look-back, wave 1 waits for it
            unsigned long long *d = p.stamps + (wave ? 8 : 0);
            atomicAdd(&d[0], st_eval);
            atomicAdd(&d[1], st_scatter);
            atomicAdd(&d[2], st_look);
            atomicAdd(&d[3], st_waitB);
            atomicAdd(&d[4], st_flush);
            atomicAdd(&d[5], st_tiles);
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

}  // namespace rvk
