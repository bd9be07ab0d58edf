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

#include "lookback.hpp"
#include "scan_frontend.hpp"

namespace rvk {

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
    uint32_t *redo_count;                   // wave ranges left to the redo kernel (a wave outgrew its slot)
    unsigned long long *redo;               // [ntiles * WAVES], zeroed per launch: kRedoFlag | output row of the range's first survivor
    unsigned long long *stamps;             // FF_STAMP builds: [8] cycle sums + tile count
    uint32_t ntiles;
    uint32_t cap_rows;                      // LDS staging capacity in rows (per round)
    int32_t nxs;
    int32_t debug;  // honoured by FF_STAMP (diagnostic) instantiations only: 1 = skip output stores, 2 = skip the look-back, 4 = count,
                    // 8 = no scanner wave, 16 = tile 1 never publishes its count (fault injection for the bounded spins)
    int32_t depth;  // 1: two slot stages, write out one iteration later; 2: three stages, two later
    uint32_t spin_limit;  // polls a look-back / the scanner waits for a missing descriptor before giving up (*err = 1)
    // 0..15: the output row of the launch's FIRST survivor.  A launch that appends to rows other launches wrote (a stretch of a table
    // filtered stretch by stretch, fused_launch.hip) gets its output pointers rounded DOWN to a 128-byte line and the rest here, so
    // that "row index % 16 == 0" still means "line-aligned address" for the whole-line stores; out_capacity counts from the rounded
    // pointer, *out_count does not include it.
    uint32_t out_bias;
    // nullptr, or [ntiles * WAVES]: survivors of every wave's row range (64 * R rows), in row order.  When a RecordBatch of
    // the stream seam is a whole number of such ranges (the reference's 1024-row batches with R = 16: exactly one), the
    // per-batch survivor counts come out of the pass itself instead of a second read of a materialised selection bitmap.
    uint32_t *wave_counts;
    // nullptr, or the control block of a LATER launch: zero_n16 x 16 bytes, zeroed by the idle waves of the scanner's workgroup while the
    // pass runs -- the memset that would otherwise sit between this pass and the next one (5 us + 6 us of queue idle per launch)
    rv_u32x4 *zero_ptr;
    uint64_t zero_n16;
    // nullptr, or [nbatch_counts], PINNED HOST memory: the same counts as 64-bit words where the caller of a window of RecordBatches
    // reads them, when a batch IS a wave range (1024 rows at 16 rows per lane): one 128-byte write per tile over PCIe
    unsigned long long *batch_counts;
    uint64_t nbatch_counts;
    // nullptr, or [ntiles * WAVES]: output row of every wave range's first survivor (the tile's offset + the waves before
    // it).  Bit-packed columns are compacted after the pass by the selection bitmap (bits_compact_kernel); with these a
    // wave of that kernel finds its output position with one load instead of a scan over the whole bitmap before it.
    uint64_t *wave_offsets;
};

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

// the scanner workgroup's other waves: zero the control block a later launch will use (plain stores: that launch starts behind this one)
template <int WAVES>
__device__ __forceinline__ void zero_for_the_next_launch(rv_u32x4 *ptr, uint64_t n16, uint32_t wave) {
    if (ptr == nullptr) return;
    const rv_u32x4 z{0u, 0u, 0u, 0u};
    for (uint64_t i = static_cast<uint64_t>(wave - 1) * 64 + static_cast<uint64_t>(lane_id()); i < n16; i += static_cast<uint64_t>(WAVES - 1) * 64) ptr[i] = z;
}

constexpr int kLdsHeader = 256;
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
// A wave with more survivors than its slot holds (a run of survivors in clustered or sorted data) takes part in
// the scan like any other -- its output range is reserved -- but leaves ITS 64 R rows to the redo kernel
// (fused_redo_waves), which re-reads that wave range alone and writes it at the reserved offset: the tile's
// other waves flush as usual, and no wave of the redo kernel waits for another.
constexpr unsigned long long kRedoFlag = 1ull << 63;
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
    // FF_NONULL: no output column can hold a null (every nullable column read is tested by a null-dropping term, or nulls
    // propagate strictly through an expression): nothing of the validity is staged or written.  With FF_PROJALL every
    // loaded column is projected; without it the per-column projection checks stay in the staging loop.
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
    // The pending tile's output offset and the wave totals exist in TWO generations (tile loop iteration parity): a word
    // written before the one workgroup barrier of iteration i is read after it and rewritten in iteration i + 2, behind
    // the barrier of i + 1 -- so one barrier per tile is enough.
    auto s_excl_of = [&](uint32_t gen) { return reinterpret_cast<uint64_t *>(smem + (gen ? 128u : 16u)); };
    auto s_wtot_of = [&](uint32_t gen) { return reinterpret_cast<uint32_t *>(smem + (gen ? 136u : 56u)); };  // [WAVES <= 16] each
    // s_pop: 8 x uint32 at smem + 24 (set-bit counts of compacted bit streams)

    const int lane = lane_id();
    const uint32_t wave = uniform32(threadIdx.x >> 6);

    // workgroup 0 (dispatched first, so resident before any tile needs it) is the scanner: one wave,
    // the others leave at once
    if (blockIdx.x == 0) {
        // debug bit 3 (FF_STAMP builds): no scanner at all -- every tile takes the fallback look-back (a test of it)
        if (wave == 0 && !(kStamp && (p.debug & 8))) scanner_wave(p.state, p.ntiles, p.err, p.spin_limit, (kStamp && (p.debug & 4)) ? p.stamps + 28 : nullptr);
        else if (wave != 0) zero_for_the_next_launch<WAVES>(p.zero_ptr, p.zero_n16, wave);
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
    uint32_t *const s_redo = reinterpret_cast<uint32_t *>(smem + 200);  // wave ranges this workgroup has left to the redo kernel
    if (threadIdx.x == 0) *s_redo = 0;
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
        bool dense, have;  // dense: THIS WAVE's survivors outgrew its slot
    };
    Pending older{0, 0, 0, 0, 0, false, false}, newer{0, 0, 0, 0, 0, false, false};
    const bool deep = p.depth == 2;
    const uint32_t nstages = deep ? 3u : 2u;
    uint32_t cur_stage = 0;
    uint64_t prev_desc = 0;  // wave 0: descriptor of the tile in front of the retiring one, loaded early

    // wave 0 (holding no row data): output offset of the retiring tile, handed to the workgroup.
    // Fast path: the scanner has already published the predecessor's inclusive prefix.
    // The descriptor as two 32-bit scalars: the status test is then a 32-bit scalar compare.  As one 64-bit value the
    // compiler turns `status == 2` into a signed 64-bit compare against a constant it keeps in a VGPR pair -- which the
    // larger instantiations spill and reload right here, behind the prefetch.
    struct Desc {
        uint32_t lo, hi;
    };
    auto resolve = [&](const Pending &r, Desc prev, uint64_t *excl_out) {  // prev: the descriptor in front of r.tile, wave-uniform
        uint64_t e;
        if (kStamp && (p.debug & 2)) e = static_cast<uint64_t>(r.tile) * 1024;
        else if (r.tile == 0) e = p.out_bias;
        else if ((prev.hi >> 30) == 2u) e = (static_cast<uint64_t>(prev.hi & 0x3FFFFFFFu) << 32) | prev.lo;
        else {
            if (kStamp && (p.debug & 4) && lane == 0) atomicAdd(p.stamps + 31, 1ull);
            e = lookback_exclusive(p.state, r.tile, r.count, p.err, p.spin_limit, kStamp ? p.stamps + 6 : nullptr);
        }
        if (lane == 0) {
            *excl_out = e;
            if (r.tile == p.ntiles - 1) *p.out_count = e + r.count - p.out_bias;
        }
    };
    uint32_t n_redo = 0;  // wave ranges this wave has left to the redo kernel
    // this wave's part of a retiring tile: the staged run goes out, or -- the wave outgrew its slot -- the range is listed
    auto write_out = [&](const Pending &r, uint64_t excl) {
        if (!r.wave_total) return;
        if (!r.dense) return flush(slot_of(r.stage), r.wave_total, excl + r.wave_prefix);
        if (lane == 0) p.redo[static_cast<uint64_t>(r.tile) * WAVES + wave] = kRedoFlag | (excl + r.wave_prefix);
        n_redo += 1;
    };

    for (uint32_t it = 0; tile < p.ntiles; ++it) {
        // ---- ticket for three iterations ahead: issued now, stored at the end of the iteration and read
        //      at the top of iteration it + 2 (the barrier of it + 1 lies in between) --------------------
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
        // ---- wave 0: the output offset of the tile that is written out in this iteration.  Its descriptor load went out at
        //      the top of the iteration and has landed with the rows; resolved HERE, before the prefetch is issued, nothing
        //      waits behind the prefetch for it (vmcnt counts in order) and nobody waits for wave 0 at a second barrier.
        uint64_t *const s_excl = s_excl_of(it & 1);
        uint32_t *const s_wtot = s_wtot_of(it & 1);
        if (wave == 0 && ret.have)
            resolve(ret, Desc{uniform32(static_cast<uint32_t>(prev_desc)), uniform32(static_cast<uint32_t>(prev_desc >> 32))}, s_excl);
        if (more) {
            load_rows<NCOLS, R, VEC>(p.in, ntb + static_cast<uint64_t>(wave) * ROWS_PER_WAVE, lane, v);
            prefetch_bits(ntb + static_cast<uint64_t>(wave) * ROWS_PER_WAVE);
        }
        if (lane == 0) s_wtot[wave] = wave_total;
        if constexpr (kStamp) {
            t1 = stamp_now();
            st_eval += t1 - t0;
            t0 = t1;
        }
        __syncthreads();  // the tile's one barrier: wave totals and the pending tile's offset are visible
        if constexpr (kStamp) {
            t1 = stamp_now();
            st_scatter += t1 - t0;
            t0 = t1;
        }

        uint32_t wave_prefix = 0, tile_count = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            const uint32_t t = s_wtot[w];
            wave_prefix += (static_cast<uint32_t>(w) < wave) ? t : 0;
            tile_count += t;
        }
        wave_prefix = uniform32(wave_prefix);
        tile_count = uniform32(tile_count);
        // successors can sum this tile's count from here on
        if (threadIdx.x == 0 && !(kStamp && (p.debug & 16) && tile == 1)) publish_aggregate(p.state, tile, tile_count, p.out_bias);
        if (p.wave_counts != nullptr && threadIdx.x < WAVES)  // one 64-byte line per tile (batch counts of seam S1)
            p.wave_counts[static_cast<uint64_t>(tile) * WAVES + threadIdx.x] = s_wtot[threadIdx.x];
        if (p.batch_counts != nullptr && threadIdx.x < WAVES) {  // ... or straight to the caller's array: a batch is a wave range
            const uint64_t b = static_cast<uint64_t>(tile) * WAVES + threadIdx.x;
            if (b < p.nbatch_counts) p.batch_counts[b] = s_wtot[threadIdx.x];
        }
        if constexpr (kStamp) {
            t1 = stamp_now();
            st_look += t1 - t0;
            t0 = t1;
        }

        // ---- write out the pending tile's slots -------------------------------------------------------------
        if (p.wave_offsets != nullptr && ret.have && lane == 0) p.wave_offsets[static_cast<uint64_t>(ret.tile) * WAVES + wave] = *s_excl + ret.wave_prefix;
        if (ret.have) write_out(ret, uniform64(*s_excl));
        older = newer;
        newer = Pending{tile, tile_count, cur_stage, wave_prefix, wave_total, wave_dense, true};
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
    __syncthreads();  // the last iteration's flush has read its offset word: the epilogue reuses generation 0
    auto retire = [&](const Pending &r) {
        if (!r.have) return;  // workgroup-uniform
        if (wave == 0) {
            if (r.tile != 0) prev_desc = ld_state(&p.state[r.tile - 1]);
            resolve(r, Desc{uniform32(static_cast<uint32_t>(prev_desc)), uniform32(static_cast<uint32_t>(prev_desc >> 32))}, s_excl_of(0));
        }
        __syncthreads();
        if (p.wave_offsets != nullptr && lane == 0) p.wave_offsets[static_cast<uint64_t>(r.tile) * WAVES + wave] = *s_excl_of(0) + r.wave_prefix;
        write_out(r, uniform64(*s_excl_of(0)));
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

    // ---- wave ranges left to the redo kernel: one global add per workgroup that has any --------------
    if (__syncthreads_or(n_redo != 0)) {  // (also orders the retire() reads of generation 0 before the epilogue's reuse)
        if (lane == 0 && n_redo) atomicAdd(s_redo, n_redo);
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(p.redo_count, *s_redo);
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
// Wave ranges (64 R consecutive rows: what ONE wave of the pass held of a tile) whose survivors outgrew the wave's LDS
// slot -- a run of survivors in clustered or sorted data.  The pass has counted them and reserved their output range
// (FusedParams::redo: kRedoFlag | output row of the range's first survivor); here ONE WAVE re-reads such a range in steps
// of 64 RR rows (RR = 16, the whole range in flight at once, for one column; 8 for two; 4 beyond), ranks inside the wave and
// stores the survivors straight from its registers at offset + rank.  Nothing is
// shared between waves: no barrier, no look-back, no second pass over the tile's other ranges.  Generic (every feature)
// and correct for any selectivity; costs one extra read of the listed ranges.  Entry e of the list belongs to wave
// e mod (waves of the launch): a run of listed ranges (sorted data) spreads over the whole grid.
constexpr int kRedoWaves = 4;       // waves per workgroup
template <int NCOLS, int RR>
__global__ __launch_bounds__(kRedoWaves * 64) void fused_redo_waves(const FusedParams p, uint32_t range_rows, uint64_t nranges) {
    constexpr int NV = NCOLS > 0 ? NCOLS : 1;
    constexpr int kRedoStageBytes = 64 * RR;  // one staged byte per row of a step, per bit-packed output
    unsigned char *const smem = rv_smem;
    const int lane = lane_id();
    const uint32_t wave = uniform32(threadIdx.x >> 6);
    if (threadIdx.x < 8) reinterpret_cast<uint32_t *>(smem + 24)[threadIdx.x] = 0;
    __syncthreads();
    // this wave's staging bytes: validity of every value column, then the bit streams
    const uint32_t my = kLdsHeader + wave * static_cast<uint32_t>((NV + kMaxBitStreams) * kRedoStageBytes);
    auto off_b = [&](int c) { return my + static_cast<uint32_t>(c) * kRedoStageBytes; };
    auto off_x = [&](int s2) { return my + static_cast<uint32_t>(NV + s2) * kRedoStageBytes; };
    const uint64_t gw = static_cast<uint64_t>(blockIdx.x) * kRedoWaves + wave, nw = static_cast<uint64_t>(gridDim.x) * kRedoWaves;

    auto redo_range = [&](uint64_t range, uint64_t g0) {
        const uint64_t first = range * range_rows;
        for (uint32_t sub = 0; sub < range_rows; sub += 64u * RR) {
            const uint64_t wave_base = first + sub;
            if (wave_base >= p.in.n) break;
            const bool full = wave_base + 64u * RR <= p.in.n;
            uint64_t v[NV][RR];
            uint32_t vb[NV], pb;
            scan_rows<NCOLS, RR, 1, FF_ALL>(p.in, wave_base, full, lane, v, vb, pb);
            if (sub + 64u * RR > range_rows) {  // a range of 384 or 768 rows (6 / 12 rows per lane) ends inside its last step
#pragma unroll
                for (int j = 0; j < RR; ++j)
                    if (sub + static_cast<uint32_t>(j) * 64u + static_cast<uint32_t>(lane) >= range_rows) pb &= ~(1u << j);
            }
            uint32_t xb[kMaxBitStreams];
#pragma unroll
            for (int s2 = 0; s2 < kMaxBitStreams; ++s2) {
                xb[s2] = 0;
                if (s2 < p.nxs) {
                    const BitStream bs = p.xs[s2];
                    xb[s2] = gather_row_bits<RR, 1>(
                        [&](int q) {
                            const uint64_t pos = bs.offset + wave_base + q * 64u;
                            uint64_t w = load_bits64(bs.src, pos, bs.src_bytes);
                            if (bs.mask) w &= load_bits64(bs.mask, pos, bs.mask_bytes);
                            return w;
                        },
                        lane);
                }
            }
            uint32_t count = 0;
#pragma unroll
            for (int j = 0; j < RR; ++j) count += static_cast<uint32_t>(__popcll(ballot64((pb >> j) & 1)));
            count = uniform32(count);
            if (!count) continue;
            if (g0 + count > p.out_capacity) {  // the counts stay exact; the host re-runs with outputs of that size
                if (lane == 0) *p.overflow = 1u;
                g0 += count;
                continue;
            }
            uint32_t running = 0;
            // EVERY row of the step survives (the inside of a run: most steps of a sorted or clustered table's listed ranges): the
            // survivor of rank r is row r, so the values leave as WHOLE 128-byte lines -- store i covers the output rows
            // [A + 64 i, A + 64 i + 64) from the aligned row A = g0 & ~15, lane l taking rank 64 i + l - (g0 & 15) out of the lane that
            // holds it (one 64-bit shuffle per row) -- instead of 512-byte runs that start anywhere: the write traffic of a sorted table's
            // redo fell from 1.13x to the survivors' bytes (WRITE_SIZE, profiles/r05_sorted10_*)
            const bool all_rows = count == 64u * RR;  // wave-uniform
            if (all_rows) {
                const uint32_t head = static_cast<uint32_t>(g0) & 15u;
                const int src = (lane - static_cast<int>(head)) & 63;
#pragma unroll
                for (int c = 0; c < NCOLS; ++c) {
                    if (!p.out_values[c]) continue;  // wave-uniform
                    uint64_t *base = p.out_values[c] + (g0 - head);
                    uint64_t prev = 0;
#pragma unroll
                    for (int i = 0; i <= RR; ++i) {
                        uint64_t cur = 0;
                        if (i < RR) cur = shfl64(((vb[c] >> i) & 1) ? v[c][i] : 0ull, src);  // placeholder 0 under a null (record_batch.rs:142-146)
                        const bool low = static_cast<uint32_t>(lane) < head;  // these lanes take the tail of the row set before
                        const bool in = low ? (i > 0) : (i < RR);
                        if (in) __builtin_nontemporal_store(low ? prev : cur, &base[64 * i + lane]);
                        prev = cur;
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < RR; ++j) {
                const bool pj = (pb >> j) & 1;
                const uint64_t m = ballot64(pj);
                const uint32_t rank = running + mbcnt(m);
                if (pj) {
#pragma unroll
                    for (int c = 0; c < NCOLS; ++c) {
                        if (!p.out_values[c]) continue;
                        const bool valid = (vb[c] >> j) & 1;  // placeholder 0 under a null (record_batch.rs:142-146)
                        if (!all_rows) __builtin_nontemporal_store(valid ? v[c][j] : 0ull, &p.out_values[c][g0 + rank]);
                        if (p.out_validity[c]) smem[off_b(c) + rank] = valid;
                    }
#pragma unroll
                    for (int s2 = 0; s2 < kMaxBitStreams; ++s2)
                        if (s2 < p.nxs) smem[off_x(s2) + rank] = (xb[s2] >> j) & 1;
                }
                running += static_cast<uint32_t>(__popcll(m));
            }
            // the staged bytes were written by this wave's lanes and are packed by the same wave: in order on the LDS queue
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int c = 0; c < NCOLS; ++c)
                if (p.out_values[c] && p.out_validity[c]) flush_bits(off_b(c), count, g0, p.out_validity[c], 24 + 4 * c);
#pragma unroll
            for (int s2 = 0; s2 < kMaxBitStreams; ++s2)
                if (s2 < p.nxs) flush_bits(off_x(s2), count, g0, p.xs[s2].out, 24 + 4 * (kMaxValueCols + s2));
            __builtin_amdgcn_wave_barrier();  // the next step restages
            g0 += count;
        }
    };

    for (uint64_t k = 0; k * 64 * nw < nranges; ++k) {
        const uint64_t e = (k * 64 + static_cast<uint64_t>(lane)) * nw + gw;
        const uint64_t entry = e < nranges ? p.redo[e] : 0ull;
        uint64_t listed = ballot64((entry & kRedoFlag) != 0);
        while (listed) {
            const int b = __builtin_ctzll(listed);
            listed &= listed - 1;
            redo_range((k * 64 + static_cast<uint64_t>(b)) * nw + gw, uniform64(shfl64(entry, b)) & ~kRedoFlag);
        }
    }
    __syncthreads();
    const uint32_t *s_pop = reinterpret_cast<const uint32_t *>(smem + 24);
    if (threadIdx.x < kMaxValueCols + kMaxBitStreams && s_pop[threadIdx.x])
        atomicAdd(&p.out_valid_pop[threadIdx.x], static_cast<unsigned long long>(s_pop[threadIdx.x]));
}

}  // namespace rvk
