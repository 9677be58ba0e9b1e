// Which XCD does a workgroup run on?  s_getreg_b32 HW_REG_XCC_ID against blockIdx.x % 8 (MI355X_MICROARCH.md: "blocks are
// dealt round-robin over the 8 XCDs ... read the id from HW_REG_XCC_ID").  hipcc --offload-arch=gfx950 -O2 tools/xcc_probe.hip -o tools/xcc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(int *out)
{
    // HW_REG_XCC_ID = 20; s_getreg_b32 simm16 = id | offset << 6 | (size - 1) << 11: bits [3:0]
    const unsigned raw = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));
    if (threadIdx.x == 0) out[blockIdx.x] = (int)raw;
}
int main()
{
    const int n = 4096;
    int *d = nullptr;
    if (hipMalloc(&d, n * sizeof(int)) != hipSuccess) return 1;
    for (int block : {64, 128}) {
        hipLaunchKernelGGL(probe, dim3(n), dim3(block), 0, 0, d);
        std::vector<int> h(n);
        if (hipMemcpy(h.data(), d, n * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return 2;
        int hist[16] = {0}, agree[8][8] = {{0}};
        for (int b = 0; b < n; b++) { hist[h[b] & 15]++; agree[b % 8][h[b] & 7]++; }
        printf("block %d: raw[0..3] = %#x %#x %#x %#x; xcc id (bits 3:0) histogram:", block, h[0], h[1], h[2], h[3]);
        for (int i = 0; i < 16; i++) printf(" %d", hist[i]);
        printf("\n  blockIdx %% 8 -> xcc id counts:\n");
        for (int i = 0; i < 8; i++) { printf("   %d:", i); for (int j = 0; j < 8; j++) printf(" %4d", agree[i][j]); printf("\n"); }
    }
    hipFree(d);
    return 0;
}
