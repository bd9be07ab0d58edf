// Scan front end shared by the fused compaction kernel (fused_kernel.hpp) and the masked-aggregate kernel
// (agg_kernel.hpp): feature flags, what a launch reads (ScanInputs), the coalesced row / bitmap loads, and the
// predicate on wave masks -- packed per-lane bits (eval_rows, the aggregate's form), per-slot wave masks in SGPR pairs
// (16-byte loads) and lane form (8-byte loads: one 64-bit VGPR whose lane k is the wave mask of row slot k).
// Reference semantics: plan.rs:112-130 (eager compare), series.rs:20-117 (AnyValue order), boolean.rs:120-165.
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
// No projected column can hold a null among the survivors (with or without FF_PROJALL) (every nullable column is tested by
// a term that drops its nulls -- the streaming composition of BASELINE config 3), so no validity is staged or
// written; the bitmaps are still read for the predicate.
enum : int { FF_NONULL = 256 };
// The term list is a conjunctive normal form with negated literals (ScanInputs::expr_mode): generic shapes only.
enum : int { FF_EXPR = 512 };
// direct kernel only (direct_kernel.hpp): some projected column keeps nulls among the survivors -- validity bits are compacted with
// the rows (a byte per survivor next to its value in the LDS slot, packed to words on the way out)
enum : int { FF_OUTVALID = 2048 };

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

}  // namespace rvk
