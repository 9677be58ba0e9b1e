// Does hipExtAnyOrderLaunch let two kernels of one stream overlap on gfx950?  Kernel A (1 workgroup) spins ~300 us and
// stamps its end; kernel B stamps its start.  s_memrealtime is a device-wide 100 MHz clock.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void ka(unsigned long long *t) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < 30000ull) __builtin_amdgcn_s_sleep(10);
    if (threadIdx.x == 0) t[0] = __builtin_amdgcn_s_memrealtime();
}
__global__ void kb(unsigned long long *t) { if (threadIdx.x == 0 && blockIdx.x == 0) t[1] = __builtin_amdgcn_s_memrealtime(); }
int main() {
    unsigned long long *d, h[2];
    hipMalloc(&d, 16); hipStream_t s; hipStreamCreate(&s);
    for (int flags = 0; flags < 2; flags++) {
        hipMemset(d, 0, 16); hipDeviceSynchronize();
        hipLaunchKernelGGL(ka, dim3(1), dim3(64), 0, s, d);
        hipExtLaunchKernelGGL(kb, dim3(256), dim3(64), 0, s, nullptr, nullptr, flags, d);
        hipDeviceSynchronize();
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("flags=%d  B.start - A.end = %lld ticks (negative: B started while A was running)\n", flags, (long long)(h[1] - h[0]));
    }
    return 0;
}
