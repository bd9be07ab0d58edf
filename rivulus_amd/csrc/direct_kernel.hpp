// Filter + compact for DENSE selections of 8-byte columns: survivors go from the registers straight to their output rows.
//
// The fused kernel (fused_kernel.hpp) stages a tile's survivors in LDS so that the tile's write-out can wait, off the critical
// path, for its output offset -- which caps a wave at the rows its LDS slots hold and, once most rows survive, leaves the
// pass working on 2048- to 4096-row tiles.  Here nothing is staged: a workgroup (eight waves, one tile of 512 R rows) loads
// its rows, evaluates the predicate, publishes the tile's count, WAITS for the prefix in front of it (the scanner wave of
// workgroup 0 normally has it within a few polls; the decoupled look-back of lookback.hpp is the fallback), and every lane
// stores its survivors at offset + rank: with most rows surviving, consecutive lanes write consecutive output rows, so the
// stores coalesce as the loads do.  The wait is covered by the other workgroups of the CU (two of them resident at 128 VGPRs, no LDS
// to speak of), not by a software pipeline.  Tiles are handed out by a ticket in launch order, so a tile only ever waits for
// tiles that are already running.
//
// Takes the same FusedParams, descriptors and control block as the fused kernel (the host's overflow re-run and read-back do
// not know the difference).  Only for launches whose outputs are value columns without a validity bitmap, no selection
// bitmap, no bit streams, no side outputs (fused_begin decides; rivulus_gpu.h option "direct").
#pragma once

#include "fused_kernel.hpp"

namespace rvk {

template <int NCOLS, int R>
__global__ __launch_bounds__(512) void fused_direct_compact(const FusedParams p) {
    static_assert(NCOLS >= 1 && NCOLS <= kMaxValueCols, "NCOLS");
    constexpr int WAVES = 8;
    constexpr uint32_t ROWS_PER_WAVE = 64u * R, TILE = ROWS_PER_WAVE * WAVES;
    __shared__ uint32_t s_tile, s_wtot[WAVES];
    __shared__ uint64_t s_excl;
    const int lane = lane_id();
    const uint32_t wave = uniform32(threadIdx.x >> 6);
    if (blockIdx.x == 0) {  // the scanner: one wave, the others leave at once
        if (wave == 0) scanner_wave(p.state, p.ntiles, p.err, p.spin_limit, nullptr);
        return;
    }
    if (threadIdx.x == 0) s_tile = __hip_atomic_fetch_add(p.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const uint32_t tile = uniform32(s_tile);
    if (tile >= p.ntiles) return;  // workgroup-uniform (the grid has one workgroup per tile)
    const uint64_t wave_base = static_cast<uint64_t>(tile) * TILE + static_cast<uint64_t>(wave) * ROWS_PER_WAVE;
    const bool full = wave_base + ROWS_PER_WAVE <= p.in.n;
    uint64_t v[NCOLS][R];
    uint32_t vb[NCOLS], pb;  // bit k: row k of this lane (row wave_base + 64 k + lane) is valid / survives
    scan_rows<NCOLS, R, 1, FF_VALIDITY | FF_BOOL>(p.in, wave_base, full, lane, v, vb, pb);
    uint32_t wave_total = 0;
#pragma unroll
    for (int k = 0; k < R; ++k) wave_total += static_cast<uint32_t>(__popcll(ballot64((pb >> k) & 1)));
    if (lane == 0) s_wtot[wave] = wave_total;
    __syncthreads();
    uint32_t wave_prefix = 0, count = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
        const uint32_t t = s_wtot[w];
        wave_prefix += static_cast<uint32_t>(w) < wave ? t : 0;
        count += t;
    }
    wave_prefix = uniform32(wave_prefix);
    count = uniform32(count);
    if (wave == 0) {
        if (lane == 0) publish_aggregate(p.state, tile, count);
        uint64_t excl = 0;
        if (tile > 0) {
            bool have = false;
            for (int poll = 0; poll < 48 && !have; ++poll) {  // the predecessor's inclusive prefix, from the scanner
                const uint64_t d = uniform64(ld_state(&p.state[tile - 1]));
                if ((d >> 62) == 2) {
                    excl = d & kStVal;
                    have = true;
                } else {
                    __builtin_amdgcn_s_sleep(4);
                }
            }
            if (!have) excl = lookback_exclusive(p.state, tile, count, p.err, p.spin_limit, nullptr);
        }
        if (lane == 0) {
            s_excl = excl;
            if (tile == p.ntiles - 1) *p.out_count = excl + count;
        }
    }
    __syncthreads();
    const uint64_t g0 = s_excl;
    if (g0 + count > p.out_capacity) {  // workgroup-uniform; the counts stay exact, the host re-runs with buffers of that size
        if (threadIdx.x == 0) *p.overflow = 1u;
        return;
    }
    uint64_t running = g0 + wave_prefix;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const bool keep = (pb >> k) & 1;
        const uint64_t m = ballot64(keep);
        const uint64_t at = running + mbcnt(m);
        if (keep) {
#pragma unroll
            for (int c = 0; c < NCOLS; ++c)
                if (p.out_values[c]) __builtin_nontemporal_store(((vb[c] >> k) & 1) ? v[c][k] : 0ull, &p.out_values[c][at]);  // placeholder 0 under a null (record_batch.rs:142-146)
        }
        running += static_cast<uint64_t>(__popcll(m));
    }
}

}  // namespace rvk
