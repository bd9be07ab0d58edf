// Output offsets of the fused compaction kernel: tile descriptors, the scanner wave that turns tile counts into
// prefixes, and the classic decoupled look-back as its fallback (fused_kernel.hpp, step 5).
#pragma once

#include "device_common.hpp"

namespace rvk {

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

// (`first`: the output row of the launch's first survivor -- FusedParams::out_bias -- enters the chain with tile 0's inclusive prefix)
__device__ __forceinline__ void publish_aggregate(uint64_t *state, uint32_t tile, uint64_t aggregate, uint32_t first) {
    st_state(&state[tile], tile == 0 ? (kStPfx | (aggregate + first)) : (kStAgg | aggregate));
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
static __device__ __attribute__((noinline)) void scanner_wave(uint64_t *state, uint32_t ntiles, uint32_t * /*err*/, uint32_t spin_limit,
                                                              unsigned long long *stats) {
    const int lane = lane_id();
    __builtin_amdgcn_s_setprio(3);  // every tile's write-out waits for this wave: ahead of the compute waves it shares a SIMD with
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
                // inclusive wave scan of the aggregates after it: a tile holds < 2^25 rows, 64 of them < 2^31 -- 32-bit DPP adds
                // (six VALU instructions; the shuffle form, twelve trips through the LDS crossbar, was most of a group's time)
                const uint32_t x = wave_scan_u32((lane > last_p && ((runmask >> lane) & 1)) ? static_cast<uint32_t>(s[k]) : 0u);
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
            // The scanner only gives up; it reports nothing.  Results never depend on it (a compute workgroup that misses
            // its prefix runs lookback_exclusive, whose own bounded spin reports a tile that is really lost), so a
            // small user-set "spin_limit" tripping here on launch skew must not fail a correct query.
            if (++idle > spin_limit) return;
            __builtin_amdgcn_s_sleep(8);
        }
    }
}

}  // namespace rvk
