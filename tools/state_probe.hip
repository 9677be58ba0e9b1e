// Microbenchmark: per-env state traffic of the core kernel.  Each workgroup (= env) reads NF fields of U
// elements and writes them back.  A: fields are separate [B][U] arrays (what the build has), B: one
// [B][NF][U] block per env.  Same bytes, same instructions; only the addresses differ.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int NF = 16;

__global__ void __launch_bounds__(128) kstate(int *base, const int *perm, long long field_stride, long long env_stride, int U)
{
    const int e = perm[blockIdx.x];
    const int u = threadIdx.x;
    if (u >= U) return;
    int v[NF];
#pragma unroll
    for (int f = 0; f < NF; f++) v[f] = base[f * field_stride + e * env_stride + u];
    int s = 0;
#pragma unroll
    for (int f = 0; f < NF; f++) s += v[f];
#pragma unroll
    for (int f = 0; f < NF; f++) base[f * field_stride + e * env_stride + u] = v[f] + (s & 1);
}

__global__ void __launch_bounds__(128) kstate4(int4 *base, const int *perm, int U)
{
    const int e = perm[blockIdx.x];
    const int u = threadIdx.x;
    if (u >= U) return;
    int4 v[NF / 4];
#pragma unroll
    for (int f = 0; f < NF / 4; f++) v[f] = base[((size_t)e * (NF / 4) + f) * U + u];
    int s = 0;
#pragma unroll
    for (int f = 0; f < NF / 4; f++) s += v[f].x + v[f].y + v[f].z + v[f].w;
#pragma unroll
    for (int f = 0; f < NF / 4; f++) { v[f].x += s & 1; base[((size_t)e * (NF / 4) + f) * U + u] = v[f]; }
}

int main()
{
    const int U = 100, B = 4096, REP = 16;      // REP disjoint state sets so that nothing stays in L2 / MALL
    const size_t n = (size_t)REP * B * NF * U;
    int *buf, *perm;
    CK(hipMalloc(&buf, n * 4)); CK(hipMemset(buf, 0, n * 4));
    CK(hipMalloc(&perm, B * 4));
    std::vector<int> h(B);
    for (int b = 0; b < B; b++) h[b] = b;
    CK(hipMemcpy(perm, h.data(), B * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = 2.0 * B * NF * U * 4;
    for (int layout = 0; layout < 2; layout++) {
        const long long fs = layout == 0 ? (long long)B * U : U, es = layout == 0 ? U : (long long)NF * U;
        float best = 1e9f;
        for (int it = 0; it < 3 * REP; it++) {
            int *set = buf + (size_t)(it % REP) * B * NF * U;
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(kstate, dim3(B), dim3(128), 0, 0, set, perm, fs, es, U);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (it >= REP && ms < best) best = ms;
        }
        printf("%-34s %7.1f us  %5.2f TB/s\n", layout == 0 ? "A separate [B][U] arrays" : "B one [B][NF][U] block per env", best * 1e3, bytes / (best * 1e-3) / 1e12);
    }
    {
        float best = 1e9f;
        for (int it = 0; it < 3 * REP; it++) {
            int *set = buf + (size_t)(it % REP) * B * NF * U;
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(kstate4, dim3(B), dim3(128), 0, 0, (int4 *)set, perm, U);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (it >= REP && ms < best) best = ms;
        }
        printf("%-34s %7.1f us  %5.2f TB/s\n", "C [B][NF/4][U] of 16-byte pieces", best * 1e3, bytes / (best * 1e-3) / 1e12);
    }
    return 0;
}
