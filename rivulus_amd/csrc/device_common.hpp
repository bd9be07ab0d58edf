// Device-side building blocks shared by every gfx950 kernel of the backend:
// column views, bit-buffer access at arbitrary bit offsets, the compare-term
// evaluator and wave64 helpers.  CDNA4 only (wave = 64 lanes, hard-coded).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rvk {

constexpr int kWave = 64;
constexpr int kMaxValueCols = 4;  // 8-byte columns one fused launch keeps in registers
constexpr int kMaxBoolCols = 2;   // bit-packed predicate columns per launch
constexpr int kMaxTerms = 16;  // literals of one launch (an expression in conjunctive normal form may repeat a term)

enum : int { DT_NULL = 0, DT_BOOLEAN = 1, DT_INT64 = 2, DT_FLOAT64 = 3 };
enum : int { OP_EQ = 0, OP_NE = 1, OP_LT = 2, OP_GT = 3, OP_LE = 4, OP_GE = 5, OP_IS_TRUE = 6 };

// Device view of PrimitiveArray / BooleanArray (reference primitive.rs:20-28, boolean.rs:9-16).
struct DevCol {
    const void *values;       // int64/double elements, or LSB-first bits for DT_BOOLEAN
    const uint8_t *validity;  // nullptr: no null bitmap
    uint64_t offset;          // element offset == bit offset
    uint64_t values_bytes;    // readable bytes behind `values` (bit buffers: tail-safe loads)
    uint64_t validity_bytes;  // readable bytes behind `validity`
    int32_t dtype;
    int32_t pad;
};

// One lowered `Column <op> Literal` term.  The host resolves type mismatch / null
// literal / null policy (reference series.rs:87-117, plan.rs:112-130) into:
//   code      which compare runs on a valid cell
//   const_v   result on a valid cell when code == TC_CONST
//   null_v    result on a null cell (always 0 under RV_NULL_DROPS)
enum : int {
    TC_CONST = 0,
    TC_I64 = 1,   // + op (signed compare)
    TC_F64 = 7,   // + op (IEEE compare: NaN false except !=, -0.0 == 0.0)
    TC_BOOL = 13  // + op, bit-packed column (false < true); OP_IS_TRUE allowed
};
// All fields are dwords: terms are read with a wave-uniform runtime index from the kernel
// arguments, and only dword-sized uniform loads go through the scalar cache (SMEM).  Byte
// fields made hipcc emit global_load_ubyte + s_waitcnt vmcnt(0), draining every row load in
// flight each time a term was decoded.
struct DevTerm {
    int64_t lit;      // int64 value, double bit pattern, or 0/1
    uint32_t packed;  // slot[0:7] | code[8:15] | op[16:19] | is_bool[20] | const_v[21] | null_v[22]
                      // | sel_lt[23] sel_eq[24] sel_gt[25] sel_un[26] | is_float[27] | negate[28] | group_end[29]
    uint32_t pad;
    __host__ __device__ uint32_t slot() const { return packed & 0xFF; }
    __host__ __device__ int code() const { return static_cast<int>((packed >> 8) & 0xFF); }
    __host__ __device__ int op() const { return static_cast<int>((packed >> 16) & 0xF); }
    __host__ __device__ bool is_bool() const { return (packed >> 20) & 1; }
    __host__ __device__ bool const_v() const { return (packed >> 21) & 1; }
    __host__ __device__ bool null_v() const { return (packed >> 22) & 1; }
    // Mask form of a compare: with lt/eq/gt/un the lane masks of "cell < lit", "==", ">" and
    // "unordered" (NaN), the term is (lt & SLT) | (eq & SEQ) | (gt & SGT) | (un & SUN).
    __host__ __device__ bool sel_lt() const { return (packed >> 23) & 1; }
    __host__ __device__ bool sel_eq() const { return (packed >> 24) & 1; }
    __host__ __device__ bool sel_gt() const { return (packed >> 25) & 1; }
    __host__ __device__ bool sel_un() const { return (packed >> 26) & 1; }
    __host__ __device__ bool is_float() const { return (packed >> 27) & 1; }
    // Expression launches (ScanInputs::expr_mode): the term list is a conjunctive normal form -- consecutive
    // literals up to one with group_end are ORed (each negated first when `negate` is set), the groups are ANDed.
    __host__ __device__ bool negate() const { return (packed >> 28) & 1; }
    __host__ __device__ bool group_end() const { return (packed >> 29) & 1; }
    __host__ __device__ void set_literal(bool negate, bool group_end) {
        packed = (packed & ~(3u << 28)) | (static_cast<uint32_t>(negate) << 28) | (static_cast<uint32_t>(group_end) << 29);
    }
    __host__ __device__ void set(uint32_t slot, int code, int op, bool is_bool, bool const_v, bool null_v) {
        packed = (slot & 0xFF) | (static_cast<uint32_t>(code) << 8) | (static_cast<uint32_t>(op) << 16) |
                 (static_cast<uint32_t>(is_bool) << 20) | (static_cast<uint32_t>(const_v) << 21) |
                 (static_cast<uint32_t>(null_v) << 22);
        bool lt = false, eq = false, gt = false, un = false;
        if (code == TC_CONST) {
            lt = eq = gt = un = const_v;
        } else {
            switch (op) {
                case OP_EQ: eq = true; break;
                case OP_NE: lt = gt = un = true; break;  // NaN != x is true
                case OP_LT: lt = true; break;
                case OP_GT: gt = true; break;
                case OP_LE: lt = eq = true; break;
                case OP_GE: gt = eq = true; break;
                default: break;
            }
        }
        const bool flt = code >= TC_F64 && code < TC_BOOL;
        packed |= (static_cast<uint32_t>(lt) << 23) | (static_cast<uint32_t>(eq) << 24) | (static_cast<uint32_t>(gt) << 25) |
                  (static_cast<uint32_t>(un) << 26) | (static_cast<uint32_t>(flt) << 27);
    }
};

// ---- wave64 helpers ---------------------------------------------------------
__device__ __forceinline__ int lane_id() { return static_cast<int>(threadIdx.x) & 63; }
// number of set bits of `mask` strictly below this lane
__device__ __forceinline__ uint32_t mbcnt(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32),
                                     __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u));
}
__device__ __forceinline__ uint64_t ballot64(bool p) { return __ballot(p); }
__device__ __forceinline__ uint32_t uniform32(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint64_t uniform64(uint64_t v) {
    uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v));
    uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v >> 32));
    return (static_cast<uint64_t>(hi) << 32) | lo;
}
typedef unsigned int rv_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int rv_u32x2 __attribute__((ext_vector_type(2)));
// A copy of x the optimizer cannot see through.  The flush of a tile runs while the NEXT tile's loads are in flight;
// address arithmetic on the lane index that is hoisted out of the tile loop ends up spilled at 128 VGPRs, and a scratch
// reload waits for every older load of the wave (`s_waitcnt vmcnt(0)`): the prefetch it was meant to overlap.
// Recomputing `lane * 8 + base` from an opaque copy costs two instructions and keeps the loads in flight.
__device__ __forceinline__ uint32_t opaque(uint32_t x) {
    asm volatile("" : "+v"(x));
    return x;
}

// ---- counters of set bits etc. that many waves add to ------------------------------------------------------------
// Atomics on ONE address are served one at a time, 12 ns each: 8192 waves adding their partial count to the same word at
// the end of a kernel is a 100 us tail (tools/micro/atomic_tail.hip; 32 separate lines: 4.6 us).  A counter is therefore
// kStripes words, each in its own 128-byte line; a wave adds to the stripe of its index and the host folds the stripes
// into the control block's word before reading it (fold_stripes_kernel, which also leaves them zeroed).
constexpr int kStripes = 32;
constexpr int kStripeWords = 16;   // uint64 words between two stripes of a counter (128 bytes)
constexpr int kStripeSlots = 16;   // counters with stripes: the first 16 words of the control block
constexpr int kStripeSlotWords = kStripes * kStripeWords;  // words between the stripes of two consecutive counters
__device__ __forceinline__ void striped_add(unsigned long long *stripes, unsigned long long v) {
    const uint32_t w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    atomicAdd(stripes + kStripeWords * (w & (kStripes - 1)), v);
}
// the same, called by ONE thread with the sum of its workgroup
__device__ __forceinline__ void striped_add_wg(unsigned long long *stripes, unsigned long long v) {
    atomicAdd(stripes + kStripeWords * (blockIdx.x & (kStripes - 1)), v);
}
// any lane's copy of a 64-bit value (per-lane source index)
__device__ __forceinline__ uint64_t shfl64(uint64_t x, int src) {
    return (static_cast<uint64_t>(static_cast<uint32_t>(__shfl(static_cast<int>(x >> 32), src, 64))) << 32) |
           static_cast<uint32_t>(__shfl(static_cast<int>(x), src, 64));
}
// lane l's copy of a 64-bit value (l wave-uniform)
__device__ __forceinline__ uint64_t readlane64(uint64_t x, int l) {
    return (static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(x >> 32), l))) << 32) |
           static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(x), l));
}
__device__ __forceinline__ uint64_t wave_sum64(uint64_t v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        uint32_t lo = __shfl_xor(static_cast<uint32_t>(v), s, 64);
        uint32_t hi = __shfl_xor(static_cast<uint32_t>(v >> 32), s, 64);
        v += (static_cast<uint64_t>(hi) << 32) | lo;
    }
    return v;
}
// inclusive prefix sum over the 64 lanes of a wave: row_shr 1 / 2 / 4 / 8 inside the rows of 16, then row_bcast 15 and 31 carry
// the row totals over (gfx9 DPP controls; lanes without a source add 0)
__device__ __forceinline__ uint32_t wave_scan_u32(uint32_t v) {
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x111, 0xf, 0xf, false));
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x112, 0xf, 0xf, false));
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x114, 0xf, 0xf, false));
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x118, 0xf, 0xf, false));
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x142, 0xa, 0xf, false));
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x143, 0xc, 0xf, false));
    return v;
}

// sum over the 64 lanes of a wave by the same DPP steps (the total lands in lane 63)
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) { return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(wave_scan_u32(v)), 63)); }
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        uint64_t b = __double_as_longlong(v);
        uint32_t lo = __shfl_xor(static_cast<uint32_t>(b), s, 64);
        uint32_t hi = __shfl_xor(static_cast<uint32_t>(b >> 32), s, 64);
        v += __longlong_as_double((static_cast<uint64_t>(hi) << 32) | lo);
    }
    return v;
}

// ---- LSB-first bit buffers at arbitrary bit offsets (reference bitmap.rs:61-68) -------
// 64-bit word `w` of a byte buffer with `nbytes` readable bytes; bytes past the end read 0.
// `base` must be 8-byte aligned (hipMalloc memory is; rv_wrap checks).
// cold path: the last, partial word of a buffer whose size is not a multiple of 8
static __device__ __attribute__((noinline)) uint64_t load_word_tail(const uint8_t *base, uint64_t b0, uint64_t nbytes) {
    uint64_t r = 0;
    for (int k = 0; k < 8; ++k)
        if (b0 + k < nbytes) r |= static_cast<uint64_t>(base[b0 + k]) << (8 * k);
    return r;
}
__device__ __forceinline__ uint64_t load_word_safe(const uint8_t *base, uint64_t w, uint64_t nbytes) {
    const uint64_t b0 = w * 8;
    if (__builtin_expect(b0 + 8 <= nbytes, 1)) return *reinterpret_cast<const uint64_t *>(base + b0);
    return load_word_tail(base, b0, nbytes);
}
// 64 bits starting at absolute bit position `bitpos`
__device__ __forceinline__ uint64_t load_bits64(const uint8_t *base, uint64_t bitpos, uint64_t nbytes) {
    uint64_t w = bitpos >> 6;
    unsigned s = static_cast<unsigned>(bitpos & 63);
    uint64_t lo = load_word_safe(base, w, nbytes);
    if (s == 0) return lo;
    uint64_t hi = load_word_safe(base, w + 1, nbytes);
    return (lo >> s) | (hi << (64 - s));
}
// mask with the low `n` bits set, n in [0, 64]
__device__ __forceinline__ uint64_t low_mask(uint64_t n) { return n >= 64 ? ~0ull : ((1ull << n) - 1); }

// ---- compare-term evaluation ---------------------------------------------------
// value cell (8 bytes) of a valid row
__device__ __forceinline__ bool eval_value_cell(int code, int64_t lit, bool const_v, uint64_t bits) {
    switch (code) {
        case TC_I64 + OP_EQ: return static_cast<int64_t>(bits) == lit;
        case TC_I64 + OP_NE: return static_cast<int64_t>(bits) != lit;
        case TC_I64 + OP_LT: return static_cast<int64_t>(bits) < lit;
        case TC_I64 + OP_GT: return static_cast<int64_t>(bits) > lit;
        case TC_I64 + OP_LE: return static_cast<int64_t>(bits) <= lit;
        case TC_I64 + OP_GE: return static_cast<int64_t>(bits) >= lit;
        case TC_F64 + OP_EQ: return __longlong_as_double(bits) == __longlong_as_double(lit);
        case TC_F64 + OP_NE: return __longlong_as_double(bits) != __longlong_as_double(lit);
        case TC_F64 + OP_LT: return __longlong_as_double(bits) < __longlong_as_double(lit);
        case TC_F64 + OP_GT: return __longlong_as_double(bits) > __longlong_as_double(lit);
        case TC_F64 + OP_LE: return __longlong_as_double(bits) <= __longlong_as_double(lit);
        case TC_F64 + OP_GE: return __longlong_as_double(bits) >= __longlong_as_double(lit);
        default: return const_v;
    }
}
// 64 rows of a bit-packed column at once: V = value bits, M = validity bits
// the same as three all-ones / all-zeros coefficients, so that a loop over many words has no switch in it:
//   truth(V, M) = (M & ((V & a) | (~V & b))) | (~M & n)
struct BoolCoef {
    uint64_t a, b, n;
};
__device__ __forceinline__ BoolCoef bool_coef(const DevTerm &t) {
    const bool lit = t.lit != 0;
    bool a, b;  // truth of a valid `true` cell / of a valid `false` cell
    switch (t.code() == TC_CONST ? -1 : t.op()) {
        case OP_IS_TRUE: a = true, b = false; break;
        case OP_EQ: a = lit, b = !lit; break;
        case OP_NE: a = !lit, b = lit; break;
        case OP_LT: a = false, b = lit; break;   // false < true
        case OP_GT: a = !lit, b = false; break;
        case OP_LE: a = lit, b = true; break;
        case OP_GE: a = true, b = !lit; break;
        default: a = b = t.const_v(); break;
    }
    return BoolCoef{a ? ~0ull : 0ull, b ? ~0ull : 0ull, t.null_v() ? ~0ull : 0ull};
}
__device__ __forceinline__ uint64_t eval_bool_word(const BoolCoef &c, uint64_t V, uint64_t M) {
    return (M & ((V & c.a) | (~V & c.b))) | (~M & c.n);
}
__device__ __forceinline__ uint64_t eval_bool_word(const DevTerm &t, uint64_t V, uint64_t M) {
    uint64_t res;
    const bool b = t.lit != 0;
    switch (t.code() == TC_CONST ? -1 : t.op()) {
        case OP_IS_TRUE: res = V; break;
        case OP_EQ: res = b ? V : ~V; break;
        case OP_NE: res = b ? ~V : V; break;
        case OP_LT: res = b ? ~V : 0ull; break;
        case OP_GT: res = b ? 0ull : V; break;
        case OP_LE: res = b ? ~0ull : ~V; break;
        case OP_GE: res = b ? V : ~0ull; break;
        default: res = t.const_v() ? ~0ull : 0ull; break;
    }
    return (M & res) | (~M & (t.null_v() ? ~0ull : 0ull));
}

}  // namespace rvk
