// How long do N fire-and-forget atomicAdds on ONE address take (one per wave, at the end of a kernel)?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/atomic_tail tools/micro/atomic_tail.hip && /tmp/atomic_tail
// Variants: one address; 32 addresses in separate 128-byte lines; one atomic per workgroup (LDS reduce first).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void tail_kernel(unsigned long long *ctr, int mode) {
    __shared__ unsigned long long s[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long v = __popcll(__ballot(threadIdx.x & 1));
    if (mode == 0) {  // one atomic per wave, one address
        if (lane == 0) atomicAdd(ctr, v);
    } else if (mode == 1) {  // one per wave, 32 lines
        if (lane == 0) atomicAdd(ctr + 16 * ((blockIdx.x * 4 + wave) & 31), v);
    } else if (mode == 2) {  // one per workgroup, one address
        if (lane == 0) s[wave] = v;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(ctr, s[0] + s[1] + s[2] + s[3]);
    } else if (mode == 3) {  // no atomic: a plain store per workgroup
        if (lane == 0) s[wave] = v;
        __syncthreads();
        if (threadIdx.x == 0) ctr[64 + blockIdx.x] = s[0] + s[1] + s[2] + s[3];
    }
}

int main() {
    unsigned long long *d;
    CHECK(hipMalloc(&d, 1 << 20));
    CHECK(hipMemset(d, 0, 1 << 20));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    const char *names[] = {"per wave, one address", "per wave, 32 lines", "per workgroup, one address", "plain store per workgroup"};
    for (int grid : {512, 2048, 8192, 65536}) {
        for (int mode = 0; mode < 4; ++mode) {
            for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(tail_kernel, dim3(grid), dim3(256), 0, 0, d, mode);
            CHECK(hipEventRecord(a));
            const int reps = 20;
            for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(tail_kernel, dim3(grid), dim3(256), 0, 0, d, mode);
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            float ms;
            CHECK(hipEventElapsedTime(&ms, a, b));
            printf("grid %6d  %-28s %8.2f us per launch  (%d atomics)\n", grid, names[mode], 1e3 * ms / reps,
                   mode < 2 ? grid * 4 : (mode == 2 ? grid : 0));
        }
    }
    return 0;
}
