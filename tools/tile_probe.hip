// Microbenchmark (round 5): a workgroup that streams whole SE tiles one after the other -- the steady state of a persistent
// rollout at <= 2 waves per SIMD (BASELINE configs[1]) -- with the tile
//   D: RB-major [R][U] float32, lane = UE, one dword per RB and lane: R load instructions per lane and tile.  A wave has at most 63
//      vector memory instructions in flight (vmcnt is 6 bits), so a tile of 135 RBs cannot be in flight at once;
//   Q: RB-quad-major [R/4][U][4] float32, lane = UE, one dwordx4 per four RBs: R/4 = 34 load instructions per lane and tile.
// Prints us per tile step of the whole batch and TB/s.   hipcc --offload-arch=gfx950 -O3 tools/tile_probe.hip -o tools/tile_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

template <int WPE>
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
kD(const float *pool, const int *tile_of, float *out, int U, int R, int K, int members)
{
    const int u = (int)threadIdx.x < U ? (int)threadIdx.x : U - 1;
    if ((int)(threadIdx.x & ~63u) >= members) return;          // (a one-wave env: the second wave leaves)
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float q[136];
    const int voff = u * 4, rb = U * 4;
    auto issue = [&](int k) {
        const float *tile = pool + (size_t)(tile_of[blockIdx.x] + k) * U * R;
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(tile), 0, U * R * 4, 0x00020000);
#pragma unroll
        for (int r = 0; r < 135; r++) q[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, r * rb, 0));
    };
    issue(0);
#pragma unroll 1
    for (int k = 0; k < K; k++) {
        double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < 135; r++) s[r & 7] += (double)q[r];
        if (k + 1 < K) issue(k + 1);
#pragma unroll
        for (int j = 0; j < 8; j++) acc[j] += s[j];
    }
    double t = 0;
    for (int j = 0; j < 8; j++) t += acc[j];
    if ((int)threadIdx.x < U) out[(size_t)blockIdx.x * U + u] = (float)t;
}

template <int WPE>
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
kQ(const float *pool, const int *tile_of, float *out, int U, int R, int K, int members)
{
    const int u = (int)threadIdx.x < U ? (int)threadIdx.x : U - 1;
    if ((int)(threadIdx.x & ~63u) >= members) return;
    const int Rq = (R + 3) / 4;
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    v4f q[34];
    const int voff = u * 16, rb = U * 16;
    auto issue = [&](int k) {
        const float *tile = pool + (size_t)(tile_of[blockIdx.x] + k) * U * Rq * 4;
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(tile), 0, U * Rq * 16, 0x00020000);
#pragma unroll
        for (int r = 0; r < 34; r++) q[r] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, r * rb, 0));
    };
    issue(0);
#pragma unroll 1
    for (int k = 0; k < K; k++) {
        double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < 34; r++) {
            s[(4 * r) & 7] += (double)q[r].x; s[(4 * r + 1) & 7] += (double)q[r].y;
            s[(4 * r + 2) & 7] += (double)q[r].z; s[(4 * r + 3) & 7] += (double)q[r].w;
        }
        if (k + 1 < K) issue(k + 1);
#pragma unroll
        for (int j = 0; j < 8; j++) acc[j] += s[j];
    }
    double t = 0;
    for (int j = 0; j < 8; j++) t += acc[j];
    if ((int)threadIdx.x < U) out[(size_t)blockIdx.x * U + u] = (float)t;
}

// The big-batch regime (BASELINE configs[2]): 5 waves per SIMD, a rotating queue of NQ x 8 RBs per lane (the step kernel's SeStream).
//   DQ: RB-major, NQ groups of 8 dword loads;   QQ: RB-quad-major, NQ groups of 2 dwordx4 loads (the same registers)
template <int NQ>
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(5, 5)))
kDQ(const float *pool, const int *tile_of, float *out, int U, int R, int K)
{
    const int u = (int)threadIdx.x < U ? (int)threadIdx.x : U - 1;
    const int voff = u * 4, rb = U * 4, G = R / 8;
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 1
    for (int k = 0; k < K; k++) {
        const float *tile = pool + (size_t)(tile_of[blockIdx.x] + k) * U * R;
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(tile), 0, U * R * 4, 0x00020000);
        float q[NQ][8];
        auto ld = [&](float (&d)[8], int g) {
#pragma unroll
            for (int j = 0; j < 8; j++) d[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, (g * 8 + j) * rb, 0));
        };
#pragma unroll
        for (int d = 0; d < NQ; d++) ld(q[d], d);
#pragma unroll 1
        for (int g = 0; g < G; g += NQ) {
#pragma unroll
            for (int d = 0; d < NQ; d++) {
                if (g + d < G) {
#pragma unroll
                    for (int j = 0; j < 8; j++) acc[j] += (double)q[d][j];
                    if (g + d + NQ < G) ld(q[d], g + d + NQ);
                }
            }
        }
    }
    double t = 0;
    for (int j = 0; j < 8; j++) t += acc[j];
    if ((int)threadIdx.x < U) out[(size_t)blockIdx.x * U + u] = (float)t;
}

template <int NQ>
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(5, 5)))
kQQ(const float *pool, const int *tile_of, float *out, int U, int R, int K)
{
    const int u = (int)threadIdx.x < U ? (int)threadIdx.x : U - 1;
    const int Rq = (R + 3) / 4;
    const int voff = u * 16, rb = U * 16, G = R / 8;
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 1
    for (int k = 0; k < K; k++) {
        const float *tile = pool + (size_t)(tile_of[blockIdx.x] + k) * U * Rq * 4;
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(tile), 0, U * Rq * 16, 0x00020000);
        v4f q[NQ][2];
        auto ld = [&](v4f (&d)[2], int g) {
            d[0] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, (g * 2) * rb, 0));
            d[1] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, (g * 2 + 1) * rb, 0));
        };
#pragma unroll
        for (int d = 0; d < NQ; d++) ld(q[d], d);
#pragma unroll 1
        for (int g = 0; g < G; g += NQ) {
#pragma unroll
            for (int d = 0; d < NQ; d++) {
                if (g + d < G) {
                    acc[0] += (double)q[d][0].x; acc[1] += (double)q[d][0].y; acc[2] += (double)q[d][0].z; acc[3] += (double)q[d][0].w;
                    acc[4] += (double)q[d][1].x; acc[5] += (double)q[d][1].y; acc[6] += (double)q[d][1].z; acc[7] += (double)q[d][1].w;
                    if (g + d + NQ < G) ld(q[d], g + d + NQ);
                }
            }
        }
    }
    double t = 0;
    for (int j = 0; j < 8; j++) t += acc[j];
    if ((int)threadIdx.x < U) out[(size_t)blockIdx.x * U + u] = (float)t;
}

int main()
{
    const int U = 100, R = 135, T = 60000, K = 40;
    float *pool; int *tile_of; float *out;
    const size_t tile_floats = (size_t)U * 136;
    CK(hipMalloc(&pool, (size_t)T * tile_floats * 4));
    CK(hipMemset(pool, 0, (size_t)T * tile_floats * 4));
    CK(hipMalloc(&tile_of, 16384 * 4)); CK(hipMalloc(&out, (size_t)16384 * U * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int B : {1024, 2048, 4096}) {
        std::vector<int> h(B);
        auto run = [&](const char *name, auto launch) {
            float best = 1e9f;
            for (int it = 0; it < 6; it++) {
                for (int b = 0; b < B; b++) h[b] = (int)(((long long)b * 9973 + it * 7919) % (T - K - 1));
                hipMemcpy(tile_of, h.data(), B * 4, hipMemcpyHostToDevice);
                hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (it > 0 && ms < best) best = ms;
            }
            const double bytes = (double)B * U * R * 4 * K;
            printf("B %5d  %-44s %7.2f us per tile step  %5.2f TB/s\n", B, name, best * 1e3 / K, bytes / (best * 1e-3) / 1e12);
        };
        run("D dword, whole tile requested, 2 waves/SIMD", [&] { hipLaunchKernelGGL(kD<2>, dim3(B), dim3(128), 0, 0, pool, tile_of, out, U, R, K, 128); });
        run("Q dwordx4, whole tile requested, 2 waves/SIMD", [&] { hipLaunchKernelGGL(kQ<2>, dim3(B), dim3(128), 0, 0, pool, tile_of, out, U, R, K, 128); });
        run("Q dwordx4, 3 waves/SIMD", [&] { hipLaunchKernelGGL(kQ<3>, dim3(B), dim3(128), 0, 0, pool, tile_of, out, U, R, K, 128); });
        run("D dword, one-wave envs (64 lanes)", [&] { hipLaunchKernelGGL(kD<2>, dim3(B), dim3(128), 0, 0, pool, tile_of, out, U, R, K, 64); });
        run("Q dwordx4, one-wave envs (64 lanes)", [&] { hipLaunchKernelGGL(kQ<2>, dim3(B), dim3(128), 0, 0, pool, tile_of, out, U, R, K, 64); });
    }
    {
        const int B = 2560, K2 = 60;           // 5120 waves: what the chip holds at 5 waves per SIMD, all resident, each streaming K2 tiles
        std::vector<int> h(B);
        auto run = [&](const char *name, auto launch) {
            float best = 1e9f;
            for (int it = 0; it < 6; it++) {
                for (int b = 0; b < B; b++) h[b] = (int)(((long long)b * 9973 + it * 7919) % (T - K2 - 1));
                hipMemcpy(tile_of, h.data(), B * 4, hipMemcpyHostToDevice);
                hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (it > 0 && ms < best) best = ms;
            }
            const double bytes = (double)B * U * R * 4 * K2;
            printf("5 waves/SIMD, %d resident workgroups  %-30s %6.2f TB/s\n", B, name, bytes / (best * 1e-3) / 1e12);
        };
        run("dword queue 2 x 8", [&] { hipLaunchKernelGGL(kDQ<2>, dim3(B), dim3(128), 0, 0, pool, tile_of, out, U, R, K2); });
        run("dword queue 4 x 8", [&] { hipLaunchKernelGGL(kDQ<4>, dim3(B), dim3(128), 0, 0, pool, tile_of, out, U, R, K2); });
        run("quad  queue 2 x (2 x4)", [&] { hipLaunchKernelGGL(kQQ<2>, dim3(B), dim3(128), 0, 0, pool, tile_of, out, U, R, K2); });
        run("quad  queue 4 x (2 x4)", [&] { hipLaunchKernelGGL(kQQ<4>, dim3(B), dim3(128), 0, 0, pool, tile_of, out, U, R, K2); });
    }
    return 0;
}
