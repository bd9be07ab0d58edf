// Filter + compact for DENSE selections of 8-byte columns: a tile's rows wait IN REGISTERS for their output offset.
//
// The fused kernel (fused_kernel.hpp) stages a tile's survivors in LDS so that the write-out can wait, off the critical
// path, for the tile's output offset -- which caps a wave at the rows its LDS slots hold and, once most rows survive, leaves
// the pass on 2048- to 4096-row tiles.  A CU has 160 KiB of LDS but 512 KiB of vector registers: here the rows never leave
// the registers they were loaded into.  Persistent workgroups, software-pipelined over two tiles:
//
//   iteration i:   tile i (set C: its PREDICATE columns, requested at the end of iteration i-1) is evaluated and counted;
//                  one workgroup barrier; its aggregate is published;
//                  tile i-1 (set S, published one iteration ago) reads its predecessor's inclusive prefix -- the scanner
//                  wave has had a whole iteration to write it -- and every lane stores its survivors at offset + rank;
//                  C moves to S (register moves), the loads of tile i+1's predicate columns and of tile i's PAYLOAD columns
//                  (the ones the predicate does not read) go out
//
// so nothing on a tile's path waits for another workgroup (round 3's direct kernel published and then waited for its own
// prefix: the launch ran at the pace of that chain, 45 tiles per microsecond, whatever the tile held).  A workgroup's load
// latency is exposed once per iteration; the other workgroup(s) of the CU, out of phase, cover it.  With most rows
// surviving, consecutive lanes write consecutive output rows: the stores coalesce as the loads do.
//
// Same FusedParams, descriptors, scanner wave, look-back fallback and control block as the fused kernel (the host's overflow
// re-run and read-back do not know the difference); tile ids follow ticket order, so a tile only ever waits for tiles that
// are already held by a running workgroup.  Only for launches whose outputs are value columns without a validity bitmap:
// no bit streams, no side outputs (fused_begin decides; rivulus_gpu.h option "direct").  The selection bitmap can be written
// on the way (slot k of a wave IS word k).
#pragma once

#include "fused_kernel.hpp"

namespace rvk {

// R 8-byte loads per lane of columns [C0, C0 + N) of `in` (rows wave_base + 64 j + lane): see load_rows
template <int C0, int N, int R>
__device__ __forceinline__ void load_cols(const ScanInputs &in, uint64_t wave_base, int lane, uint64_t (&v)[N > 0 ? N : 1][R]) {
    constexpr uint32_t ROWS_PER_WAVE = 64u * R;
    const uint64_t left = in.n > wave_base ? in.n - wave_base : 0;
    const uint32_t nbytes = uniform32(static_cast<uint32_t>(left < ROWS_PER_WAVE ? left : ROWS_PER_WAVE) * 8u);
#pragma unroll
    for (int c = 0; c < N; ++c) {
        const uint64_t base = uniform64(reinterpret_cast<uint64_t>(in.cols[C0 + c].values) + (in.cols[C0 + c].offset + wave_base) * 8);
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(base), 0, nbytes, 0x00020000);
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const rv_u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rsrc, lane * 8, j * 512, kStreamPolicy);
            v[c][j] = (static_cast<uint64_t>(t.y) << 32) | t.x;
        }
    }
}

// Truth of one value term on every row slot of a wave, in lane form (lane k = wave mask of slot k; scan_frontend.hpp): the
// compare as selector algebra on the "<", "==", ">" (and unordered) masks the host folded the operator into (DevTerm::sel_*),
// so there is no switch over the twelve compares -- one integer and one IEEE loop.  A TC_CONST term has all four selectors equal.
template <int R>
__device__ __forceinline__ uint64_t term_lanes(const DevTerm &t, const uint64_t (&v)[R]) {
    const uint64_t SLT = t.sel_lt() ? ~0ull : 0, SEQ = t.sel_eq() ? ~0ull : 0, SGT = t.sel_gt() ? ~0ull : 0, SUN = t.sel_un() ? ~0ull : 0;
    uint64_t acc = 0;
    if (t.is_float()) {
        const double b = __longlong_as_double(t.lit);
        static_for<0, R>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            const double a = __longlong_as_double(v[k]);
            const uint64_t lt = ballot64(a < b), eq = ballot64(a == b), gt = ballot64(a > b);
            acc = set_lane64<k>(acc, (lt & SLT) | (eq & SEQ) | (gt & SGT) | (~(lt | eq | gt) & SUN));
        });
    } else {
        const int64_t b = t.lit;
        static_for<0, R>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            const uint64_t lt = ballot64(static_cast<int64_t>(v[k]) < b), eq = ballot64(static_cast<int64_t>(v[k]) == b);
            acc = set_lane64<k>(acc, (lt & SLT) | (eq & SEQ) | (~(lt | eq) & SGT));
        });
    }
    return acc;
}

// NP: 8-byte columns the predicate reads (value slots [0, NP): the host numbers them first), NQ: the other loaded columns
// (slots [NP, NP + NQ), all projected).  FLAGS: FF_VALIDITY (a predicate column has a null bitmap; no null survives),
// FF_BOOL (terms over bit-packed Boolean columns).  WPE: waves per SIMD the register budget is set for.
template <int NP, int NQ, int R, int WAVES, int FLAGS, int WPE>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(WPE))) void fused_direct_compact(const FusedParams p) {
    static_assert(NP >= 0 && NQ >= 0 && NP + NQ >= 1 && NP + NQ <= kMaxValueCols, "columns");
    static_assert(WAVES >= 1 && WAVES <= 16 && R <= 32, "geometry");
    constexpr int NPV = NP > 0 ? NP : 1, NQV = NQ > 0 ? NQ : 1;
    constexpr bool kValidity = (FLAGS & FF_VALIDITY) != 0, kBool = (FLAGS & FF_BOOL) != 0;
    constexpr bool kStamp = (FLAGS & FF_STAMP) != 0;  // diagnostic instantiation: s_memtime sums per phase (tools/dense_stamp.py)
    constexpr bool kOutV = (FLAGS & FF_OUTVALID) != 0;  // some projected column keeps its nulls: validity travels with the rows
    static_assert(!kOutV || kValidity, "FF_OUTVALID comes with FF_VALIDITY");
    unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0 = 0, t1 = 0;
    auto mark = [&](int i) {
        if constexpr (kStamp) {
            t1 = stamp_now();
            st[i] += t1 - t0;
            t0 = t1;
        }
    };
    constexpr uint32_t ROWS_PER_WAVE = 64u * R, TILE = ROWS_PER_WAVE * WAVES;
    __shared__ uint32_t s_tick;            // the next tile id (thread 0 draws it one iteration before its loads go out, see below)
    __shared__ uint32_t s_wtot[2][WAVES];  // wave totals and the retiring tile's offset, two generations by iteration parity
    __shared__ uint64_t s_excl[2];
    __shared__ uint32_t s_pop[kMaxValueCols];  // FF_OUTVALID: valid cells among the survivors, per column (null counts on the host side)
    const int lane = lane_id();
    const uint32_t wave = uniform32(threadIdx.x >> 6);
    if (blockIdx.x == 0) {  // the scanner: one wave, the others leave at once
        if (wave == 0) scanner_wave(p.state, p.ntiles, p.err, p.spin_limit, (p.debug & 4) ? p.stamps + 28 : nullptr);
        else zero_for_the_next_launch<WAVES>(p.zero_ptr, p.zero_n16, wave);
        return;
    }
    // the predicate terms live in VGPR lanes (lane t = term t), as in the fused kernel: no scalar loads on the tile path
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
    const int nterms = p.in.nterms;
    if (threadIdx.x == 0) s_tick = __hip_atomic_fetch_add(p.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (kOutV && threadIdx.x < kMaxValueCols) s_pop[threadIdx.x] = 0;
    __syncthreads();
    auto wave_base_of = [&](uint32_t tile) { return static_cast<uint64_t>(tile) * TILE + static_cast<uint64_t>(wave) * ROWS_PER_WAVE; };
    // C: the tile whose predicate columns are in flight / being counted.  S: the tile waiting for its offset.
    uint64_t cxp[NPV][R], sxp[NPV][R], sxq[NQV][R];
    uint64_t vw[NPV];                      // validity words of C's predicate columns, lane q = aligned word q of the wave's range
    uint64_t bw[kBool ? 2 * kMaxBoolCols : 1];  // words of the Boolean predicate columns (values, validity)
    uint64_t s_sv = 0;                     // S's survive masks, lane k = slot k
    uint32_t c_tile = uniform32(s_tick), s_tile = 0, s_count = 0, s_wave_prefix = 0, s_wave_total = 0;
    uint32_t l_tile = 0, l_count = 0, l_wave_prefix = 0, l_wave_total = 0;  // L: the tile whose survivors wait in this wave's LDS slot
    bool c_valid = c_tile < p.ntiles, s_valid = false, l_valid = false;
    // this wave's LDS slot: 64 R rows of every loaded column, column after column
    // (FF_OUTVALID: + a validity byte per row of every column that has an output bitmap, behind the values)
    constexpr uint32_t kColBytes = ROWS_PER_WAVE * 8u, kValBytes = kColBytes * (NP + NQ);
    uint32_t outmask = 0;  // bit q: value slot q is projected; bit 8 + q: with an output bitmap
#pragma unroll
    for (int q = 0; q < NP + NQ; ++q) outmask |= (p.out_values[q] ? 1u << q : 0u) | ((kOutV && p.out_validity[q]) ? 0x100u << q : 0u);
    outmask = uniform32(outmask);
    const uint32_t slot = wave * (kValBytes + static_cast<uint32_t>(__builtin_popcount(outmask >> 8)) * ROWS_PER_WAVE);
    // byte column of value slot q inside the slot's validity area: the columns with an output bitmap, in slot order
    auto vcol_of = [&](int q) { return slot + kValBytes + static_cast<uint32_t>(__builtin_popcount((outmask >> 8) & ((1u << q) - 1u))) * ROWS_PER_WAVE; };
    uint64_t s_vwin[NPV];  // FF_OUTVALID: validity windows (lane k = slot k) of S's predicate columns ...
    uint64_t s_qvw[NQV];   // ... and the validity words of S's payload columns, requested with their rows
    uint32_t pop[NP + NQ]; // valid cells this wave has written, per column
#pragma unroll
    for (int q = 0; q < NPV; ++q) s_vwin[q] = ~0ull;
#pragma unroll
    for (int q = 0; q < NQV; ++q) s_qvw[q] = ~0ull;
#pragma unroll
    for (int q = 0; q < NP + NQ; ++q) pop[q] = 0;
    auto request_c = [&](uint64_t base) {
        if constexpr (NP > 0) load_cols<0, NP, R>(p.in, base, lane, cxp);
        if constexpr (kValidity && NP > 0) load_validity_words<NP, R>(p.in, base, lane, vw);
        if constexpr (kBool) {
#pragma unroll
            for (int c = 0; c < kMaxBoolCols; ++c) {
                const DevCol col = p.in.bcols[c];
                bw[2 * c] = load_bit_words<R>(static_cast<const uint8_t *>(col.values), col.offset + base, col.values_bytes, lane);
                bw[2 * c + 1] = load_bit_words<R>(col.validity, col.offset + base, col.validity_bytes, lane);
            }
        }
    };
    // Two round trips of every iteration go to another agent's memory and come back in microseconds under the launch's own traffic:
    // the id of the tile after next (a fetch-add on the ticket counter) and the descriptor in front of L (the scanner's prefix).  Both
    // are requested at the END of the iteration before, AHEAD of C's rows -- loads return in order, so they are back when the first
    // row is -- and wave 0 takes them behind barrier 1.  (Round 4 drew the id where wave 0 looks L's offset up, between two barriers:
    // every wave of the workgroup waited ~2 us for it, a third of an iteration at 10 % kept; and requested the descriptor at the top
    // of the iteration, which the compiler turned into "behind the last of C's rows" -- the loop header copies them out of the load
    // registers.)  The unit is built with the atomic optimizer off (Makefile): it rewrites a one-lane fetch-add into add +
    // readfirstlane and waits where it stands.  A workgroup thus holds one unpublished id more, drawn an iteration before its loads go
    // out; it still only ever waits for LOWER ids, held by running workgroups.
    // Only wave 0 uses the descriptor, every wave requests it: with the load under `wave == 0` the compiler has to cover both paths
    // where C's rows are first used and waits for vmcnt(0) -- in wave 0 for this load's whole round trip.
    uint32_t ticket = 0;      // thread 0: the id drawn for the tile after C
    uint64_t prev_desc = 0;   // the descriptor in front of L
    if (threadIdx.x == 0) ticket = __hip_atomic_fetch_add(p.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_sched_barrier(0);
    if (c_valid) request_c(wave_base_of(c_tile));

    for (uint32_t it = 0; c_valid || s_valid || l_valid; ++it) {  // workgroup-uniform
        const uint32_t gen = it & 1;
        const uint64_t c_base = wave_base_of(c_tile);

        if constexpr (kStamp) {
            t0 = stamp_now();
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): C's rows are here -- separates the load wait from the predicate
            mark(0);
        }
        // ---- C: predicate in lane form (one 64-bit VGPR value, lane k = wave mask of row slot k), count ----------------
        uint64_t c_sv = 0;
        uint32_t wave_total = 0;
        uint64_t c_vwin[NPV];  // validity windows of C's predicate columns (lane k = the 64 rows of slot k), all ones without a bitmap
#pragma unroll
        for (int c = 0; c < NPV; ++c) c_vwin[c] = ~0ull;
        if (c_valid) {
            c_sv = ~0ull;
            if (static_cast<uint64_t>(c_tile) * TILE + TILE > p.in.n) {  // the ragged last tile
                const int64_t left = static_cast<int64_t>(p.in.n) - static_cast<int64_t>(c_base) - 64 * lane;
                c_sv = left <= 0 ? 0ull : low_mask(static_cast<uint64_t>(left > 64 ? 64 : left));
            }
            uint64_t(&vwin)[NPV] = c_vwin;
            bool hv[NPV];
#pragma unroll
            for (int c = 0; c < NPV; ++c) {
                hv[c] = false;
                if constexpr (kValidity && NP > 0)
                    if (p.in.cols[c].validity) {
                        hv[c] = true;
                        vwin[c] = validity_windows(vw[c], uniform32(static_cast<uint32_t>((p.in.cols[c].offset + c_base) & 63)));
                    }
            }
            uint64_t bwin[kBool ? 2 * kMaxBoolCols : 1];
            if constexpr (kBool) {
#pragma unroll
                for (int c = 0; c < kMaxBoolCols; ++c) {
                    const uint32_t sh = uniform32(static_cast<uint32_t>((p.in.bcols[c].offset + c_base) & 63));
                    bwin[2 * c] = validity_windows(bw[2 * c], sh);
                    bwin[2 * c + 1] = validity_windows(bw[2 * c + 1], sh);
                }
            }
            // truth of a term on every slot, null rows at null_v (the AnyValue table is lowered on the host)
            auto term_truth = [&](const DevTerm &term) -> uint64_t {
                uint64_t X = 0;
                if (!term.is_bool()) {
#pragma unroll
                    for (int c = 0; c < NP; ++c)
                        if (term.slot() == static_cast<uint32_t>(c)) {
                            X = term_lanes<R>(term, cxp[c]);
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
            if (p.in.expr_mode) {  // conjunctive normal form with negated literals (device_common.hpp, DevTerm)
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
                    for (int c = 0; c < NP; ++c)
                        if (hv[c] && ((p.in.strict_values >> c) & 1)) c_sv &= vwin[c];
                }
                if constexpr (kBool) {
#pragma unroll
                    for (int c = 0; c < kMaxBoolCols; ++c)
                        if (((p.in.strict_bools >> c) & 1) && p.in.bcols[c].validity) c_sv &= bwin[2 * c + 1];
                }
                c_sv &= p.in.negate_result ? ~A : A;
            } else {
                for (int t = 0; t < nterms; ++t) c_sv &= term_truth(term_at(t));
            }
            if (lane >= R) c_sv = 0;
            if (p.out_selection) sel_store<R>(c_sv, p.out_selection, c_base, p.in.n, lane);  // slot k IS selection word k
            wave_total = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(wave_scan_u32(static_cast<uint32_t>(__popcll(c_sv)))), 63));
        }
        wave_total = uniform32(wave_total);
        if (lane == 0) s_wtot[gen][wave] = wave_total;
        mark(1);
        __syncthreads();
        mark(2);
        uint32_t c_wave_prefix = 0, c_count = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            const uint32_t t = s_wtot[gen][w];
            c_wave_prefix += static_cast<uint32_t>(w) < wave ? t : 0;
            c_count += t;
        }
        c_wave_prefix = uniform32(c_wave_prefix);
        c_count = uniform32(c_count);
        // C's aggregate goes out BEFORE anything of this iteration can wait for another workgroup: a tile's publication depends on
        // its loads alone.  (Published after S's offset lookup, one workgroup's fallback look-back held back its own next aggregate,
        // hence every later tile's prefix, and four tiles in five ended up in the fallback themselves.)
        // wave 0: the descriptor requested at the top of the iteration, taken before the aggregate's store is issued (a wait for that
        // load placed behind the store waited for the store's round trip as well: vmcnt counts in order)
        uint32_t dhi = 0, dlo = 0;
        if (wave == 0) dhi = uniform32(static_cast<uint32_t>(prev_desc >> 32)), dlo = uniform32(static_cast<uint32_t>(prev_desc));
        if (threadIdx.x == 0) s_tick = ticket;  // (every wave has read the last id: they are past barrier 1)
        if (c_valid) {
            if (threadIdx.x == 0) publish_aggregate(p.state, c_tile, c_count, p.out_bias);
            if (p.wave_counts != nullptr && threadIdx.x < WAVES)  // batch counts of seam S1 (fused_kernel.hpp)
                p.wave_counts[static_cast<uint64_t>(c_tile) * WAVES + threadIdx.x] = s_wtot[gen][threadIdx.x];
            if (p.batch_counts != nullptr && threadIdx.x < WAVES) {
                const uint64_t b = static_cast<uint64_t>(c_tile) * WAVES + threadIdx.x;
                if (b < p.nbatch_counts) p.batch_counts[b] = s_wtot[gen][threadIdx.x];
            }
        }
        if (wave == 0 && l_valid) {  // the output offset of L: the scanner's prefix in front of it, or the look-back
            uint32_t hi = dhi, lo = dlo;
            for (int poll = 0; poll < 4 && l_tile != 0 && (hi >> 30) != 2u; ++poll) {  // not there at the top of the iteration: look again
                const uint64_t d = ld_state(&p.state[l_tile - 1]);
                hi = uniform32(static_cast<uint32_t>(d >> 32)), lo = uniform32(static_cast<uint32_t>(d));
                if ((p.debug & 4) && lane == 0) atomicAdd(p.stamps + 16 + poll, 1ull);  // diagnostic: tiles whose prefix took this many looks more
            }
            uint64_t e;
            if (l_tile == 0) e = p.out_bias;
            else if ((hi >> 30) == 2u) e = (static_cast<uint64_t>(hi & 0x3FFFFFFFu) << 32) | lo;
            else {
                if ((p.debug & 4) && lane == 0) atomicAdd(p.stamps + 31, 1ull);  // diagnostic: tiles that took the fallback
                e = lookback_exclusive(p.state, l_tile, l_count, p.err, p.spin_limit, nullptr);
            }
            if (lane == 0) {
                s_excl[gen] = e;
                if (l_tile == p.ntiles - 1) *p.out_count = e + l_count - p.out_bias;
            }
        }
        mark(3);
        __syncthreads();  // S's offset is visible
        mark(4);

        // ---- L: this wave's run leaves its LDS slot as whole 128-byte lines --------------------------------------------------
        // Output rows [w0, w0 + cnt) of every projected column; lane l of store i writes row (w0 & ~15) + 64 i + l, so every store
        // instruction covers four aligned lines (only the run's first and last line are shared with a neighbour).
        if (l_valid) {
            const uint64_t g0 = uniform64(s_excl[gen]);
            const uint64_t w0 = g0 + l_wave_prefix;  // output row of this wave's first survivor
            if (p.wave_offsets != nullptr && lane == 0) p.wave_offsets[static_cast<uint64_t>(l_tile) * WAVES + wave] = w0;
            const bool fits = g0 + l_count <= p.out_capacity;  // else: the counts stay exact, the host re-runs with buffers of that size
            if (!fits && threadIdx.x == 0) *p.overflow = 1u;
            const uint32_t head = static_cast<uint32_t>(w0) & 15u;
            const uint32_t span = fits ? head + l_wave_total : 0u;  // rows from the aligned start to the run's end
            const int32_t rank0 = static_cast<int32_t>(lane) - static_cast<int32_t>(head);
#pragma unroll
            for (int q = 0; q < NP + NQ; ++q) {
                if (!((outmask >> q) & 1)) continue;  // wave-uniform
                const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(p.out_values[q] + (w0 - head)), 0, span * 8u, 0x00020000);
                const uint32_t col = slot + q * kColBytes;
#pragma unroll
                for (int i = 0; i <= R; ++i) {
                    if (static_cast<uint32_t>(i) * 64u >= span) break;  // wave-uniform
                    const int32_t rank = rank0 + 64 * i;
                    if (rank >= 0) {  // rows past the run's end fall outside the descriptor and are dropped by the hardware
                        const uint64_t v = *reinterpret_cast<const uint64_t *>(rv_smem + col + static_cast<uint32_t>(rank) * 8u);
                        __builtin_amdgcn_raw_buffer_store_b64(rv_u32x2{static_cast<uint32_t>(v), static_cast<uint32_t>(v >> 32)}, rsrc, (lane + 64 * i) * 8, 0, kStreamPolicy);
                    }
                }
                if constexpr (kOutV) {
                    // the column's validity: the staged bytes [0, cnt) -> output bits [w0, w0 + cnt).  Lane l owns bit l of output word
                    // (w0 >> 6) + i in step i and the ballot packs 64 bytes; the steps are unrolled (their LDS reads overlap) and
                    // the words collected in lane i, so that ONE store writes the run -- its first and last word, shared with the
                    // neighbouring waves, OR-merged into the zero-filled bitmap (fused_kernel.hpp, flush_bits)
                    if (fits && ((outmask >> (8 + q)) & 1) && l_wave_total) {
                        const uint32_t vcol = vcol_of(q);
                        const uint32_t bhead = static_cast<uint32_t>(w0) & 63u;
                        const uint32_t bits = bhead + l_wave_total;  // from the first word's bit 0 to the run's end
                        uint64_t mine = 0;
                        uint32_t valid_here = 0;
#pragma unroll
                        for (int i = 0; i <= R; ++i) {
                            const int32_t r = static_cast<int32_t>(lane) + 64 * i - static_cast<int32_t>(bhead);  // rank of this lane's bit
                            const bool in = r >= 0 && r < static_cast<int32_t>(l_wave_total);
                            const uint64_t word = ballot64(in && (rv_smem[vcol + (in ? static_cast<uint32_t>(r) : 0u)] & 1));
                            mine = lane == i ? word : mine;
                            valid_here += static_cast<uint32_t>(__popcll(word));
                        }
                        pop[q] += valid_here;
                        const uint32_t nwords = (bits + 63u) >> 6;
                        if (static_cast<uint32_t>(lane) < nwords) {
                            uint64_t *dst = p.out_validity[q] + (w0 >> 6) + lane;
                            const bool shared = (lane == 0 && bhead != 0) || (static_cast<uint32_t>(lane) + 1 == nwords && (bits & 63u) != 0);
                            if (!shared) *dst = mine;
                            else if (mine) atomicOr(reinterpret_cast<unsigned long long *>(dst), static_cast<unsigned long long>(mine));
                        }
                    }
                }
            }
        }
        // ---- S: the survivors move from the registers into the slot, at their rank in the wave's run --------------------------
        if (s_valid) {
            uint64_t s_qwin[NQV];  // validity windows of S's payload columns out of the words requested with their rows
#pragma unroll
            for (int q = 0; q < NQV; ++q) {
                s_qwin[q] = ~0ull;
                if constexpr (kOutV && NQ > 0)
                    if ((outmask >> (8 + NP + q)) & 1)
                        s_qwin[q] = validity_windows(s_qvw[q], uniform32(static_cast<uint32_t>((p.in.cols[NP + q].offset + wave_base_of(s_tile)) & 63)));
            }
            uint32_t running = 0;
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const uint64_t m = readlane64(s_sv, k);
                const uint32_t at = slot + (running + mbcnt(m)) * 8u;
                if (lane_of(m)) {
                    const uint32_t vrank = (at - slot) >> 3;  // the row's rank in the run: its byte in a validity column (FF_OUTVALID)
#pragma unroll
                    for (int q = 0; q < NP; ++q) {
                        if (!((outmask >> q) & 1)) continue;
                        uint64_t v = sxp[q][k];
                        if constexpr (kOutV)
                            if ((outmask >> (8 + q)) & 1) {  // placeholder 0 under a null (record_batch.rs:142-146)
                                const bool valid = lane_of(readlane64(s_vwin[q], k));
                                v = valid ? v : 0;
                                rv_smem[vcol_of(q) + vrank] = valid;
                            }
                        *reinterpret_cast<uint64_t *>(rv_smem + at + q * kColBytes) = v;
                    }
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        uint64_t v = sxq[q][k];
                        if constexpr (kOutV)
                            if ((outmask >> (8 + NP + q)) & 1) {
                                const bool valid = lane_of(readlane64(s_qwin[q], k));
                                v = valid ? v : 0;
                                rv_smem[vcol_of(NP + q) + vrank] = valid;
                            }
                        *reinterpret_cast<uint64_t *>(rv_smem + at + (NP + q) * kColBytes) = v;
                    }
                }
                running += static_cast<uint32_t>(__popcll(m));
            }
        }

        mark(5);
        __syncthreads();
        mark(6);
        const uint32_t next_tile = uniform32(s_tick);
        // ---- S is in the slot (L), C becomes S; the next tile's predicate columns and this tile's payload columns are requested ----
        l_valid = s_valid, l_tile = s_tile, l_count = s_count, l_wave_prefix = s_wave_prefix, l_wave_total = s_wave_total;
        s_valid = c_valid, s_tile = c_tile, s_count = c_count, s_wave_prefix = c_wave_prefix, s_wave_total = wave_total, s_sv = c_sv;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
#pragma unroll
            for (int k = 0; k < R; ++k) sxp[q][k] = cxp[q][k];
            if constexpr (kOutV) s_vwin[q] = c_vwin[q];
        }
        c_tile = next_tile;
        c_valid = c_tile < p.ntiles;
        // the id after that and L's descriptor first (see above), then the rows
        ticket = 0, prev_desc = 0;
        if (threadIdx.x == 0) ticket = __hip_atomic_fetch_add(p.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (l_valid && l_tile != 0) prev_desc = ld_state(&p.state[l_tile - 1]);
        __builtin_amdgcn_sched_barrier(0);
        if (c_valid) request_c(wave_base_of(c_tile));
        if constexpr (NQ > 0)
            if (s_valid) {
                load_cols<NP, NQ, R>(p.in, c_base, lane, sxq);
                if constexpr (kOutV) {
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const DevCol col = p.in.cols[NP + q];
                        s_qvw[q] = ((outmask >> (8 + NP + q)) & 1) ? load_bit_words<R>(col.validity, col.offset + c_base, col.validity_bytes, lane) : ~0ull;
                    }
                }
            }
        mark(7);
    }
    if constexpr (kOutV) {  // valid cells written: one LDS add per wave and column, one global add per workgroup and column
#pragma unroll
        for (int q = 0; q < NP + NQ; ++q)
            if (lane == 0 && pop[q]) atomicAdd(&s_pop[q], pop[q]);
        __syncthreads();
        if (threadIdx.x < NP + NQ && s_pop[threadIdx.x]) atomicAdd(&p.out_valid_pop[threadIdx.x], static_cast<unsigned long long>(s_pop[threadIdx.x]));
    }
    if constexpr (kStamp) {
        if (lane == 0 && wave == 1) {  // wave 0 runs the offset lookup; wave 1 is an ordinary wave
#pragma unroll
            for (int i = 0; i < 8; ++i) atomicAdd(&p.stamps[i], st[i]);
        }
        if (lane == 0 && wave == 0) atomicAdd(&p.stamps[8], st[3]);  // wave 0's share: publish + offset lookup + ticket
    }
}

}  // namespace rvk
