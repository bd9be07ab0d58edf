// K4: filter + SUM/COUNT over survivors, one pass, no output column.
// The reference has no aggregate operator (parity unpinned);
// semantics: COUNT = surviving rows, SUM over surviving NON-NULL cells of value slot 0,
// Int64 wrapping two's complement (order independent => bit exact on any geometry),
// Float64 through a fixed reduction tree (reproducible for a fixed launch geometry).
// Stage 1 writes one {sum_i, sum_f, count} partial per tile; stage 2 folds the partials
// in index order with one workgroup.  No atomics, so results do not depend on timing.
#pragma once

#include "scan_frontend.hpp"

namespace rvk {

struct AggPartial {
    int64_t sum_i;
    double sum_f;
    uint64_t count;
    uint64_t pad;
};

struct AggParams {
    ScanInputs in;
    AggPartial *partials;  // [workgroups of the launch]
    int32_t agg_is_float;
    uint32_t ntiles;
};

template <int NCOLS, int R, int VEC, int WAVES, int FLAGS>
__global__ __launch_bounds__(WAVES * 64) void filter_agg_kernel(const AggParams p) {
    constexpr uint32_t ROWS_PER_WAVE = 64u * R;
    constexpr uint32_t TILE = ROWS_PER_WAVE * WAVES;
    constexpr int NV = NCOLS > 0 ? NCOLS : 1;
    __shared__ AggPartial s_part[WAVES];

    const int lane = lane_id();
    const uint32_t wave = uniform32(threadIdx.x >> 6);
    // grid-stride over the tiles: launched with one workgroup per tile this is one pass of the body; launched with a few
    // workgroups per CU every lane keeps its sums across its tiles (the order of a lane's additions depends on the grid
    // only, so a Float64 sum stays reproducible for a given launch geometry)
    uint64_t si = 0, cnt = 0;
    double sf = 0.0;
    for (uint32_t tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        const uint64_t tile_base = static_cast<uint64_t>(tile) * TILE;
        const uint64_t wave_base = tile_base + static_cast<uint64_t>(wave) * ROWS_PER_WAVE;
        const bool full = tile_base + TILE <= p.in.n;
        uint64_t v[NV][R];
        uint32_t vb[NV];
        uint32_t pb;
        scan_rows<NCOLS, R, VEC, FLAGS>(p.in, wave_base, full, lane, v, vb, pb);
        cnt += static_cast<uint64_t>(__popc(pb));
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const bool take = ((pb >> k) & 1) && ((vb[0] >> k) & 1);
            if (p.agg_is_float) sf += take ? __longlong_as_double(v[0][k]) : 0.0;
            else si += take ? v[0][k] : 0;
        }
    }
    si = wave_sum64(si);
    cnt = wave_sum64(cnt);
    sf = wave_sum_f64(sf);
    if (lane == 0) s_part[wave] = AggPartial{static_cast<int64_t>(si), sf, cnt, 0};
    __syncthreads();
    if (threadIdx.x == 0) {
        AggPartial t{0, 0.0, 0, 0};
        for (int w = 0; w < WAVES; ++w) {
            t.sum_i = static_cast<int64_t>(static_cast<uint64_t>(t.sum_i) + static_cast<uint64_t>(s_part[w].sum_i));
            t.sum_f += s_part[w].sum_f;
            t.count += s_part[w].count;
        }
        p.partials[blockIdx.x] = t;
    }
}

// The selectivity of a predicate nobody has run on this table yet, from a strided sample: workgroup j counts the survivors of the
// 1024 rows from row j * stride on (same scan front end, every feature), adds them to a device word, and the workgroup that
// arrives last hands the total to the host through pinned memory (value, then sequence number: the host spins on that word --
// no copy to queue, no stream to drain) and leaves the device words zero for the next sample.  1024 workgroups read 8 KiB per
// column each: ~10 us for a table of any size.  What a fused launch is sized from when the context has never seen its
// predicate (fused_launch.hip): the reference's operators have no warm-up call either (stream.rs:136-158).
// Besides the total, a histogram of the blocks by how many of their 1024 rows survive (16 buckets of 64 rows; a block in which every
// row survives counts in the last): whether the survivors are spread evenly or come in runs -- a table that is sorted or clustered on
// the predicate's column -- decides how many of a staged pass's waves would outgrow their LDS slot (fused_launch.hip).
constexpr int kSampleBuckets = 16;
constexpr int kSampleBlocks = 1024;
// {survivors, arrivals | sequence}, four 16-bit bucket counts per word, then the PROFILE: the survivors of every block in table order,
// four 16-bit counts per word -- where in the table the survivors are (a sorted table: none, then all), which the histogram cannot tell
constexpr int kSampleHistWords = kSampleBuckets / 4;
constexpr int kSampleWords = 2 + kSampleHistWords + kSampleBlocks / 4;
struct SampleParams {
    ScanInputs in;
    uint64_t stride;                 // rows between the starts of two sampled blocks
    unsigned long long *dev_words;   // [kSampleWords] {survivors, arrivals, packed histogram}, zero between samples
    volatile unsigned long long *host_words;  // [kSampleWords] pinned: {survivors, sequence, packed histogram}
    unsigned long long sequence;
};
template <int NCOLS>
__global__ __launch_bounds__(256) void sample_count_kernel(const SampleParams p) {
    constexpr int NV = NCOLS > 0 ? NCOLS : 1;
    constexpr int RS = 4;  // rows per lane: 4 waves x 64 x 4 = 1024 rows per workgroup
    const int lane = lane_id();
    const uint32_t wave = uniform32(threadIdx.x >> 6);
    const uint64_t wave_base = static_cast<uint64_t>(blockIdx.x) * p.stride + static_cast<uint64_t>(wave) * (64u * RS);
    uint64_t v[NV][RS];
    uint32_t vb[NV], pb = 0;
    if (wave_base < p.in.n) scan_rows<NCOLS, RS, 1, FF_VALIDITY | FF_BOOL>(p.in, wave_base, wave_base + 64u * RS <= p.in.n, lane, v, vb, pb);
    const uint64_t cnt = wave_sum64(static_cast<uint64_t>(__popc(pb)));
    __shared__ unsigned long long s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    if (lane == 0 && cnt) atomicAdd(&s_cnt, static_cast<unsigned long long>(cnt));
    __syncthreads();
    __shared__ int s_last;
    if (threadIdx.x == 0) {
        if (s_cnt) atomicAdd(&p.dev_words[0], s_cnt);
        {  // at most 1024 blocks: a bucket's count fits 16 bits
            const uint32_t bucket = s_cnt >= 1024 ? kSampleBuckets - 1 : static_cast<uint32_t>(s_cnt >> 6);
            atomicAdd(&p.dev_words[2 + bucket / 4], 1ull << (16 * (bucket % 4)));
            atomicAdd(&p.dev_words[2 + kSampleHistWords + blockIdx.x / 4], static_cast<unsigned long long>(s_cnt) << (16 * (blockIdx.x % 4)));
        }
        __threadfence();
        s_last = atomicAdd(&p.dev_words[1], 1ull) + 1 == gridDim.x;  // the last workgroup
    }
    __syncthreads();
    if (s_last) {  // (workgroup-uniform) histogram and profile first, every thread a few words; then the total, then the sequence number
        __threadfence();
        for (int w = 2 + static_cast<int>(threadIdx.x); w < kSampleWords; w += 256) p.host_words[w] = atomicExch(&p.dev_words[w], 0ull);
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned long long total = atomicExch(&p.dev_words[0], 0ull);
            atomicExch(&p.dev_words[1], 0ull);
            p.host_words[0] = total;
            __threadfence_system();
            p.host_words[1] = p.sequence;
        }
    }
}

// stage 2: fixed order (template only so the header can be shared by several units).  Workgroup b folds the partials
// [b * chunk, (b + 1) * chunk) into out[b]; a launch with one workgroup and chunk >= n is the final fold.  A 1e10-row
// shard leaves 2.4 M per-tile partials: folded by ONE workgroup that was 0.8 ms of a 13 ms step, so large inputs take two
// levels (the geometry of both depends on n only: results stay reproducible run to run).
template <int UNUSED>
__global__ __launch_bounds__(1024) void agg_final_kernel(const AggPartial *all, uint32_t total, uint32_t chunk, AggPartial *outs) {
    __shared__ AggPartial s[16];
    const uint64_t first = static_cast<uint64_t>(blockIdx.x) * chunk;
    const AggPartial *partials = all + first;
    const uint32_t n = first >= total ? 0u : (total - first < chunk ? static_cast<uint32_t>(total - first) : chunk);
    AggPartial *out = outs + blockIdx.x;
    uint64_t si = 0, cnt = 0;
    double sf = 0.0;
    for (uint32_t i = threadIdx.x; i < n; i += 1024) {
        si += static_cast<uint64_t>(partials[i].sum_i);
        sf += partials[i].sum_f;
        cnt += partials[i].count;
    }
    si = wave_sum64(si);
    cnt = wave_sum64(cnt);
    sf = wave_sum_f64(sf);
    if (lane_id() == 0) s[threadIdx.x >> 6] = AggPartial{static_cast<int64_t>(si), sf, cnt, 0};
    __syncthreads();
    if (threadIdx.x == 0) {
        AggPartial t{0, 0.0, 0, 0};
        for (int w = 0; w < 16; ++w) {
            t.sum_i = static_cast<int64_t>(static_cast<uint64_t>(t.sum_i) + static_cast<uint64_t>(s[w].sum_i));
            t.sum_f += s[w].sum_f;
            t.count += s[w].count;
        }
        *out = t;
    }
}

}  // namespace rvk
