// Which CU does a workgroup run on?  s_getreg of HW_REG_HW_ID (gfx9: cu_id[11:8], sh_id[12], se_id[15:13]) and of
// HW_REG_XCC_ID (gfx940+).  Prints the distinct (xcc, se, sh, cu) keys of a 1024-workgroup launch and how many workgroups
// shared each -- the direct kernel keeps streaming workgroups off the scanner wave's CU by this key (direct_kernel.hpp).
//   hipcc --offload-arch=gfx950 -O2 -o hwid hwid.hip && ./hwid
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ void probe(unsigned *out) {
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));    // HW_REG_HW_ID, 32 bits
        const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));   // HW_REG_XCC_ID, bits [3:0]
        out[2 * blockIdx.x] = hw;
        out[2 * blockIdx.x + 1] = xcc;
    }
    for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(10);  // stay resident so that the launch spreads over the chip
}
int main() {
    const int n = 1024;
    unsigned *d;
    hipMalloc(&d, n * 8);
    hipLaunchKernelGGL(probe, dim3(n), dim3(512), 0, 0, d);
    std::vector<unsigned> h(2 * n);
    hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, int> keys;
    for (int i = 0; i < n; ++i) keys[(h[2 * i + 1] << 16) | (h[2 * i] & 0xFF00)]++;
    printf("%zu distinct (xcc, se, sh, cu) keys over %d workgroups\n", keys.size(), n);
    for (int i = 0; i < 12; ++i) printf("wg %d: hw_id %08x xcc %u -> cu %u sh %u se %u\n", i, h[2 * i], h[2 * i + 1], (h[2 * i] >> 8) & 15, (h[2 * i] >> 12) & 1, (h[2 * i] >> 13) & 7);
    int hist[16] = {0};
    for (auto &kv : keys) hist[kv.second < 15 ? kv.second : 15]++;
    for (int i = 1; i < 16; ++i) if (hist[i]) printf("%d keys hold %d workgroups\n", hist[i], i);
    return 0;
}
