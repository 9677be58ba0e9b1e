// Microbenchmark: how fast can one MI355X stream [tile][R][U] float32 tiles with different load shapes?
//   A: lane = UE, one dword per row per lane (the env-step pattern), DEPTH rows in flight
//   B: lane = (UE quad, row mod 8), one dwordx4 per 8-row group per lane
//   C: flat float4 copy-reduce of the same bytes (upper bound)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int DEPTH>
__global__ void __launch_bounds__(128) kA(const float *pool, const int *tile_of, float *out, int U, int R)
{
    const float *tile = pool + (size_t)tile_of[blockIdx.x] * U * R;
    const int u = threadIdx.x < U ? threadIdx.x : U - 1;
    float acc = 0.f;
    float q[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; d++) q[d] = tile[(size_t)d * U + u];
    for (int r = 0; r < R; r += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            if (r + d < R) {
                acc += q[d];
                const int rn = r + d + DEPTH;
                if (rn < R) q[d] = tile[(size_t)rn * U + u];
            }
        }
    }
    if (threadIdx.x < U) out[(size_t)blockIdx.x * U + u] = acc;
}

template <int DEPTH>
__global__ void __launch_bounds__(256) kB(const float *pool, const int *tile_of, float *out, int U, int R)
{
    const float *tile = pool + (size_t)tile_of[blockIdx.x] * U * R;
    const int C = U / 4;
    int c = threadIdx.x >> 3; const int j = threadIdx.x & 7;
    const bool act = c < C; if (!act) c = C - 1;
    const int G = R / 8;
    float4 q[DEPTH];
    float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int d = 0; d < DEPTH; d++) if (d < G) q[d] = *(const float4 *)(tile + (size_t)(d * 8 + j) * U + 4 * c);
    for (int g = 0; g < G; g += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            if (g + d < G) {
                acc.x += q[d].x; acc.y += q[d].y; acc.z += q[d].z; acc.w += q[d].w;
                const int gn = g + d + DEPTH;
                if (gn < G) q[d] = *(const float4 *)(tile + (size_t)(gn * 8 + j) * U + 4 * c);
            }
        }
    }
    if (act) out[(size_t)blockIdx.x * U + 4 * c + (j & 3)] = acc.x + acc.y + acc.z + acc.w;
}

__global__ void __launch_bounds__(256) kC(const float4 *pool, const int *tile_of, float *out, int n4)
{
    const float4 *tile = pool + (size_t)tile_of[blockIdx.x] * n4;
    float acc = 0.f;
    for (int i = threadIdx.x; i < n4; i += 256 * 4) {
        float4 a = tile[i], b = i + 256 < n4 ? tile[i + 256] : make_float4(0, 0, 0, 0);
        float4 c = i + 512 < n4 ? tile[i + 512] : make_float4(0, 0, 0, 0), d = i + 768 < n4 ? tile[i + 768] : make_float4(0, 0, 0, 0);
        acc += a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w + c.x + c.y + c.z + c.w + d.x + d.y + d.z + d.w;
    }
    if (acc == 12345.f) out[blockIdx.x] = acc;
}


// D: LDS-DMA staging: groups of 8 rows (8*U floats, contiguous) are copied global -> LDS with 16-byte
// per-lane loads that bypass the VGPRs (global_load_lds_dwordx4); lane = UE then reads its 8 values of
// the group from LDS.  NB ring buffers, one barrier per group, refill lags one group.
template <int NB, int NI>
__global__ void __launch_bounds__(128) kD(const float *pool, const int *tile_of, float *out, int U, int R)
{
    extern __shared__ __align__(16) unsigned char lds[];
    const float *tile = pool + (size_t)tile_of[blockIdx.x] * U * R;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int u = tid < U ? tid : U - 1;
    const int CB = 32 * U;                    // bytes per 8-row group
    const int G = R / 8;
    auto issue = [&](int g) {
        const char *src = (const char *)tile + (size_t)g * CB;
        unsigned char *dst = lds + (g % NB) * CB;
#pragma unroll
        for (int k = 0; k < NI; k++) {
            const int off = k * 2048 + wave * 1024 + lane * 16;
            if (off < CB)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + off),
                                                 (__attribute__((address_space(3))) void *)(dst + k * 2048 + wave * 1024), 16, 0, 0);
        }
    };
#pragma unroll
    for (int g = 0; g < NB - 1; g++) if (g < G) issue(g);
    float acc = 0.f;
    for (int g = 0; g < G; g++) {
        if (g + NB - 1 <= G) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NB - 2) * NI) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (g + NB - 1 < G) issue(g + NB - 1);
        const float *b = (const float *)(lds + (g % NB) * CB);
#pragma unroll
        for (int j = 0; j < 8; j++) acc += b[j * U + u];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (tid < U) out[(size_t)blockIdx.x * U + u] = acc;
}


// E: the env-step form: buffer loads, scalar row offset, lane = UE, DEPTH rows in flight, f64 accumulate
template <int DEPTH>
__global__ void __launch_bounds__(256) kE(const float *pool, const int *tile_of, float *out, int U, int R, int pitch)
{
    const float *tile = pool + (size_t)tile_of[blockIdx.x] * pitch * R;
    const int u = threadIdx.x < U ? threadIdx.x : U - 1;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(tile), 0, pitch * R * 4, 0x00020000);
    const int voff = u * 4, rb = pitch * 4;
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float q[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; d++) q[d] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, d * rb, 0));
#pragma unroll 1
    for (int r = 0; r < R; r += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            if (r + d < R) {
                acc[d & 7] += (double)q[d];
                const int rn = r + d + DEPTH;
                if (rn < R) q[d] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, rn * rb, 0));
            }
        }
    }
    double t = 0; for (int j = 0; j < 8; j++) t += acc[j];
    if (threadIdx.x < U) out[(size_t)blockIdx.x * U + u] = (float)t;
}

// F: one wave covers whole rows with 8-byte loads (lane i = UEs 2i, 2i+1); the NW waves of the workgroup
// take the rows r = w (mod NW)
template <int DEPTH, int NW>
__global__ void __launch_bounds__(256) kF(const float *pool, const int *tile_of, float *out, int U, int R)
{
    const float *tile = pool + (size_t)tile_of[blockIdx.x] * U * R;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = lane * 2 < U ? lane : 0;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(tile), 0, U * R * 4, 0x00020000);
    const int voff = c * 8, rb = U * 4;
    double a0[4] = {0, 0, 0, 0}, a1[4] = {0, 0, 0, 0};
    float2 q[DEPTH];
    auto ld = [&](int k) { return __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, (k * NW + w) * rb, 0)); };
    const int n = (R - w + NW - 1) / NW;      // rows of this wave
#pragma unroll
    for (int d = 0; d < DEPTH; d++) if (d < n) q[d] = ld(d);
#pragma unroll 1
    for (int k = 0; k < n; k += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            if (k + d < n) {
                a0[d & 3] += (double)q[d].x; a1[d & 3] += (double)q[d].y;
                if (k + d + DEPTH < n) q[d] = ld(k + d + DEPTH);
            }
        }
    }
    const double t0 = (a0[0] + a0[1]) + (a0[2] + a0[3]), t1 = (a1[0] + a1[1]) + (a1[2] + a1[3]);
    if (lane * 2 < U) { out[(size_t)blockIdx.x * U + lane * 2] = (float)(t0 + w); out[(size_t)blockIdx.x * U + lane * 2 + 1] = (float)t1; }
}

// G: 16 bytes per lane along U (lane i = UEs 4i..4i+3), a wave-load covers 2.5 rows of 400 B: lanes 0..24 row k,
// 25..49 row k+1 (NW waves take alternating row pairs)
template <int DEPTH, int NW>
__global__ void __launch_bounds__(256) kG(const float *pool, const int *tile_of, float *out, int U, int R)
{
    const float *tile = pool + (size_t)tile_of[blockIdx.x] * U * R;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int C = U / 4;                       // float4 per row
    const int RPW = 64 / C;                    // rows per wave-load (2 for U = 100)
    const int sub = lane / C < RPW ? lane / C : RPW - 1, c = lane % C;
    const bool act = lane < RPW * C;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(tile), 0, U * R * 4, 0x00020000);
    const int rb = U * 4;
    const int voff = sub * rb + c * 16;
    double a[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) a[i][j] = 0;
    float4 q[DEPTH];
    const int n = (R / RPW - w + NW - 1) / NW;   // row groups of this wave
    auto ld = [&](int k) { return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, (k * NW + w) * RPW * rb, 0)); };
#pragma unroll
    for (int d = 0; d < DEPTH; d++) if (d < n) q[d] = ld(d);
#pragma unroll 1
    for (int k = 0; k < n; k += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            if (k + d < n) {
                a[d & 3][0] += (double)q[d].x; a[d & 3][1] += (double)q[d].y; a[d & 3][2] += (double)q[d].z; a[d & 3][3] += (double)q[d].w;
                if (k + d + DEPTH < n) q[d] = ld(k + d + DEPTH);
            }
        }
    }
    double t = 0;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) t += a[i][j];
    if (act) out[(size_t)blockIdx.x * U + (lane % U)] = (float)t;
}

int main()
{
    const int U = 100, R = 135, B = 4096, T = 40000;
    float *pool; int *tile_of; float *out;
    CK(hipMalloc(&pool, (size_t)T * U * R * 4));
    CK(hipMemset(pool, 0, (size_t)T * U * R * 4));
    CK(hipMalloc(&tile_of, B * 4)); CK(hipMalloc(&out, (size_t)B * U * 4));
    std::vector<int> h(B);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = (double)B * U * R * 4;
    auto run = [&](const char *name, auto launch) {
        float best = 1e9f;
        for (int it = 0; it < 12; it++) {
            for (int b = 0; b < B; b++) h[b] = (int)(((long long)b * 9973 + it * 7919) % T);
            hipMemcpy(tile_of, h.data(), B * 4, hipMemcpyHostToDevice);
            hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (it > 1 && ms < best) best = ms;
        }
        printf("%-28s %8.1f us  %6.2f TB/s\n", name, best * 1e3, bytes / (best * 1e-3) / 1e12);
    };
    run("A dword lane=UE depth 8", [&] { hipLaunchKernelGGL(kA<8>, dim3(B), dim3(128), 0, 0, pool, tile_of, out, U, R); });
    run("A dword lane=UE depth 16", [&] { hipLaunchKernelGGL(kA<16>, dim3(B), dim3(128), 0, 0, pool, tile_of, out, U, R); });
    run("A dword lane=UE depth 32", [&] { hipLaunchKernelGGL(kA<32>, dim3(B), dim3(128), 0, 0, pool, tile_of, out, U, R); });
    run("B dwordx4 quad x row8 d2", [&] { hipLaunchKernelGGL(kB<2>, dim3(B), dim3(256), 0, 0, pool, tile_of, out, U, R); });
    run("B dwordx4 quad x row8 d4", [&] { hipLaunchKernelGGL(kB<4>, dim3(B), dim3(256), 0, 0, pool, tile_of, out, U, R); });
    run("B dwordx4 quad x row8 d8", [&] { hipLaunchKernelGGL(kB<8>, dim3(B), dim3(256), 0, 0, pool, tile_of, out, U, R); });
    hipFuncSetAttribute((const void *)kD<4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void *)kD<6, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void *)kD<3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    run("D lds-dma ring 3", [&] { hipLaunchKernelGGL((kD<3, 2>), dim3(B), dim3(128), 3 * 32 * U, 0, pool, tile_of, out, U, R); });
    run("D lds-dma ring 4", [&] { hipLaunchKernelGGL((kD<4, 2>), dim3(B), dim3(128), 4 * 32 * U, 0, pool, tile_of, out, U, R); });
    run("D lds-dma ring 6", [&] { hipLaunchKernelGGL((kD<6, 2>), dim3(B), dim3(128), 6 * 32 * U, 0, pool, tile_of, out, U, R); });

    run("E buffer dword d16 (128 thr)", [&] { hipLaunchKernelGGL(kE<16>, dim3(B), dim3(128), 0, 0, pool, tile_of, out, U, R, U); });
    run("E buffer dword d24 (128 thr)", [&] { hipLaunchKernelGGL(kE<24>, dim3(B), dim3(128), 0, 0, pool, tile_of, out, U, R, U); });
    run("E buffer dword d48 (128 thr)", [&] { hipLaunchKernelGGL(kE<48>, dim3(B), dim3(128), 0, 0, pool, tile_of, out, U, R, U); });
    run("F b64 full rows 2 waves d12", [&] { hipLaunchKernelGGL((kF<12, 2>), dim3(B), dim3(128), 0, 0, pool, tile_of, out, U, R); });
    run("F b64 full rows 2 waves d24", [&] { hipLaunchKernelGGL((kF<24, 2>), dim3(B), dim3(128), 0, 0, pool, tile_of, out, U, R); });
    run("F b64 full rows 1 wave  d24", [&] { hipLaunchKernelGGL((kF<24, 1>), dim3(B), dim3(64), 0, 0, pool, tile_of, out, U, R); });
    run("F b64 full rows 4 waves d12", [&] { hipLaunchKernelGGL((kF<12, 4>), dim3(B), dim3(256), 0, 0, pool, tile_of, out, U, R); });
    run("G b128 2 rows/load 2 waves d8", [&] { hipLaunchKernelGGL((kG<8, 2>), dim3(B), dim3(128), 0, 0, pool, tile_of, out, U, R); });
    run("G b128 2 rows/load 2 waves d16", [&] { hipLaunchKernelGGL((kG<16, 2>), dim3(B), dim3(128), 0, 0, pool, tile_of, out, U, R); });
    run("G b128 2 rows/load 1 wave d16", [&] { hipLaunchKernelGGL((kG<16, 1>), dim3(B), dim3(64), 0, 0, pool, tile_of, out, U, R); });
    run("G b128 2 rows/load 4 waves d8", [&] { hipLaunchKernelGGL((kG<8, 4>), dim3(B), dim3(256), 0, 0, pool, tile_of, out, U, R); });
    run("C flat float4", [&] { hipLaunchKernelGGL(kC, dim3(B), dim3(256), 0, 0, (const float4 *)pool, tile_of, out, U * R / 4); });
    return 0;
}
