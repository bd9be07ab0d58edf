// Microbenchmark: what does MI355X HBM deliver for "stream-read 8 GB, write a compacted 10 %"?
// Gives the ceiling the fused filter kernel can be compared with (tools/micro, not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -o mixbench mixbench.hip && ./mixbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int R = 32, WAVES = 8, TILE = 64 * R * WAVES;

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
// LP / SP: cache-policy bits of the buffer load / store (gfx940+: bit0 sc0, bit1 nt, bit4 sc1).
// keep == 0: read only; else each wave writes `keep` rows at (tile*WAVES + wave)*keep.
// BURST > 1: the stores of BURST tiles are issued back to back (survivors parked in LDS meanwhile).
template <int LP, int SP, int BURST>
__global__ __launch_bounds__(WAVES * 64) void k(const uint64_t *in, uint64_t *out, uint64_t *sink, uint32_t ntiles, uint32_t keep) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ uint64_t park[BURST > 1 ? BURST * WAVES * 256 : 1];
    uint64_t acc = 0;
    uint32_t it = 0;
    uint32_t tiles[BURST];
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x, ++it) {
        const uint64_t *src = in + (uint64_t)tile * TILE + wave * 64 * R;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint64_t *>(src), 0, 64 * R * 8, 0x00020000);
        uint64_t v[R];
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs, lane * 8, j * 512, LP);
            v[j] = ((uint64_t)t.y << 32) | t.x;
        }
#pragma unroll
        for (int j = 0; j < R; ++j) acc += v[j];
        if (keep) {
            if (BURST == 1) {
                uint64_t *dst = out + ((uint64_t)tile * WAVES + wave) * keep;
                const auto rd = __builtin_amdgcn_make_buffer_rsrc(dst, 0, keep * 8, 0x00020000);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (j * 64 < keep) {
                        u32x2 t; t.x = (unsigned)v[j]; t.y = (unsigned)(v[j] >> 32);
                        __builtin_amdgcn_raw_buffer_store_b64(t, rd, lane * 8, j * 512, SP);
                    }
                }
            } else {
                const uint32_t slot = it % BURST;
                uint64_t *pk = park + (slot * WAVES + wave) * 256;
#pragma unroll
                for (int j = 0; j < 4; ++j) pk[j * 64 + lane] = v[j];
#pragma unroll
                for (int q = 0; q < BURST; ++q) if (q == slot) tiles[q] = tile;
                if (slot == BURST - 1) {
#pragma unroll
                    for (int q = 0; q < BURST; ++q) {
                        uint64_t *dst = out + ((uint64_t)tiles[q] * WAVES + wave) * keep;
                        const auto rd = __builtin_amdgcn_make_buffer_rsrc(dst, 0, keep * 8, 0x00020000);
                        const uint64_t *pq = park + (q * WAVES + wave) * 256;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (j * 64 < keep) {
                                const uint64_t x = pq[j * 64 + lane];
                                u32x2 t; t.x = (unsigned)x; t.y = (unsigned)(x >> 32);
                                __builtin_amdgcn_raw_buffer_store_b64(t, rd, lane * 8, j * 512, SP);
                            }
                        }
                    }
                }
            }
        }
    }
    if (acc == 0x1234567) *sink = acc;
}

// write only: every wave stores 64*R rows per tile
template <int SP>
__global__ __launch_bounds__(WAVES * 64) void wr(uint64_t *out, uint32_t ntiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        uint64_t *dst = out + (uint64_t)tile * TILE + wave * 64 * R;
        const auto rd = __builtin_amdgcn_make_buffer_rsrc(dst, 0, 64 * R * 8, 0x00020000);
        u32x2 t; t.x = lane; t.y = tile;
#pragma unroll
        for (int j = 0; j < R; ++j) __builtin_amdgcn_raw_buffer_store_b64(t, rd, lane * 8, j * 512, SP);
    }
}

int main() {
    const uint64_t n = 1000000000ull;
    const uint32_t ntiles = n / TILE;
    uint64_t *in, *out, *sink;
    CK(hipMalloc(&in, n * 8));
    CK(hipMalloc(&out, n * 8 / 4));
    CK(hipMalloc(&sink, 8));
    CK(hipMemset(in, 1, n * 8));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto run = [&](const char *name, auto kern, uint32_t keep, int grid) {
        for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), 0, 0, in, out, sink, ntiles, keep);
        CK(hipEventRecord(a));
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), 0, 0, in, out, sink, ntiles, keep);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
        printf("%-40s keep=%4u grid=%5d: %.3f ms  read %.2f TB/s  total %.2f TB/s\n", name, keep, grid, ms, n * 8 / ms / 1e9,
               (n * 8 + (double)ntiles * WAVES * keep * 8) / ms / 1e9);
        fflush(stdout);
    };
    const int grid = 1024;
    run("read, policy 0", k<0, 0, 1>, 0, grid);
    run("read, nt", k<2, 0, 1>, 0, grid);
    run("read, sc0", k<1, 0, 1>, 0, grid);
    run("read, sc1", k<16, 0, 1>, 0, grid);
    run("read, sc1 nt", k<18, 0, 1>, 0, grid);
    run("read, sc0 sc1", k<17, 0, 1>, 0, grid);
    run("read, sc0 sc1 nt", k<19, 0, 1>, 0, grid);
    for (uint32_t keep : {205u, 20u}) {
        run("load nt, store 0", k<2, 0, 1>, keep, grid);
        run("load nt, store nt", k<2, 2, 1>, keep, grid);
        run("load nt, store sc0", k<2, 1, 1>, keep, grid);
        run("load nt, store sc1", k<2, 16, 1>, keep, grid);
        run("load nt, store sc0 sc1", k<2, 17, 1>, keep, grid);
        run("load nt, store sc1 nt", k<2, 18, 1>, keep, grid);
        run("load nt, store sc0 sc1 nt", k<2, 19, 1>, keep, grid);
        run("load nt, store sc0 nt", k<2, 3, 1>, keep, grid);
        run("load 0, store 0", k<0, 0, 1>, keep, grid);
        run("load 0, store nt", k<0, 2, 1>, keep, grid);
        run("burst 4: load nt, store nt", k<2, 2, 4>, keep, 512);
        run("burst 4: load nt, store 0", k<2, 0, 4>, keep, 512);
        run("burst 1: load nt, store nt, grid 512", k<2, 2, 1>, keep, 512);
    }
    {
        auto w = [&](const char *name, auto kern) {
            for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(1024), dim3(WAVES * 64), 0, 0, out, ntiles / 4);
            CK(hipEventRecord(a));
            for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(1024), dim3(WAVES * 64), 0, 0, out, ntiles / 4);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
            printf("%-40s 2 GB: %.3f ms  %.2f TB/s\n", name, ms, (double)(ntiles / 4) * TILE * 8 / ms / 1e9);
        };
        w("write only, policy 0", wr<0>);
        w("write only, nt", wr<2>);
        w("write only, sc1", wr<16>);
        w("write only, sc0 sc1", wr<17>);
    }
    return 0;
}
