// Issue cost of the VALU instructions the step kernel is made of, on gfx950: each kernel runs ITER x 8 independent
// instructions of one kind per wave, 8 waves per SIMD on every SIMD; cycles per wave-instruction and SIMD = time * clock / count.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_probe.hip -o tools/valu_probe && tools/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 4096
#define OP8(asmstr) \
    for (int i = 0; i < ITER; i++) { \
        asm volatile(asmstr :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(f0), "v"(f1), "v"(i0), "v"(i1)); }
template <int K> __global__ void __launch_bounds__(256) probe(double *out, float seed)
{
    double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3;
    float f0 = seed, f1 = seed * 2;
    int i0 = (int)seed, i1 = i0 + 7;
    double r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0, r5 = 0, r6 = 0, r7 = 0;
    for (int i = 0; i < ITER; i++) {
        if (K == 0) asm volatile("v_add_f64 %0, %8, %9\n v_add_f64 %1, %9, %10\n v_add_f64 %2, %10, %11\n v_add_f64 %3, %8, %11\n v_add_f64 %4, %8, %9\n v_add_f64 %5, %9, %10\n v_add_f64 %6, %10, %11\n v_add_f64 %7, %8, %11"
                                 : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
        if (K == 1) asm volatile("v_fma_f64 %0, %8, %9, %10\n v_fma_f64 %1, %9, %10, %11\n v_fma_f64 %2, %10, %11, %8\n v_fma_f64 %3, %8, %11, %9\n v_fma_f64 %4, %8, %9, %10\n v_fma_f64 %5, %9, %10, %11\n v_fma_f64 %6, %10, %11, %8\n v_fma_f64 %7, %8, %11, %9"
                                 : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
        if (K == 2) asm volatile("v_cvt_f64_f32 %0, %8\n v_cvt_f64_f32 %1, %9\n v_cvt_f64_f32 %2, %8\n v_cvt_f64_f32 %3, %9\n v_cvt_f64_f32 %4, %8\n v_cvt_f64_f32 %5, %9\n v_cvt_f64_f32 %6, %8\n v_cvt_f64_f32 %7, %9"
                                 : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(f0), "v"(f1));
        if (K == 3) { int t0, t1, t2, t3, t4, t5, t6, t7;
            asm volatile("v_add_u32 %0, %8, %9\n v_add_u32 %1, %9, %8\n v_add_u32 %2, %8, %9\n v_add_u32 %3, %9, %8\n v_add_u32 %4, %8, %9\n v_add_u32 %5, %9, %8\n v_add_u32 %6, %8, %9\n v_add_u32 %7, %9, %8"
                         : "=v"(t0), "=v"(t1), "=v"(t2), "=v"(t3), "=v"(t4), "=v"(t5), "=v"(t6), "=v"(t7) : "v"(i0), "v"(i1));
            r0 += t0 + t1 + t2 + t3 + t4 + t5 + t6 + t7 == 12345 ? 1.0 : 0.0; }
        if (K == 4) asm volatile("v_mul_f64 %0, %8, %9\n v_mul_f64 %1, %9, %10\n v_mul_f64 %2, %10, %11\n v_mul_f64 %3, %8, %11\n v_mul_f64 %4, %8, %9\n v_mul_f64 %5, %9, %10\n v_mul_f64 %6, %10, %11\n v_mul_f64 %7, %8, %11"
                                 : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
        if (K == 5) asm volatile("v_rcp_f64 %0, %8\n v_rcp_f64 %1, %9\n v_rcp_f64 %2, %10\n v_rcp_f64 %3, %11\n v_rcp_f64 %4, %8\n v_rcp_f64 %5, %9\n v_rcp_f64 %6, %10\n v_rcp_f64 %7, %11"
                                 : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
        if (K == 6) { int t0, t1, t2, t3, t4, t5, t6, t7;
            asm volatile("v_cndmask_b32 %0, %8, %9, vcc\n v_cndmask_b32 %1, %9, %8, vcc\n v_cndmask_b32 %2, %8, %9, vcc\n v_cndmask_b32 %3, %9, %8, vcc\n v_cndmask_b32 %4, %8, %9, vcc\n v_cndmask_b32 %5, %9, %8, vcc\n v_cndmask_b32 %6, %8, %9, vcc\n v_cndmask_b32 %7, %9, %8, vcc"
                         : "=v"(t0), "=v"(t1), "=v"(t2), "=v"(t3), "=v"(t4), "=v"(t5), "=v"(t6), "=v"(t7) : "v"(i0), "v"(i1) : "vcc");
            r0 += t0 + t1 + t2 + t3 + t4 + t5 + t6 + t7 == 12345 ? 1.0 : 0.0; }
        if (K == 7) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cmp_lt_u32 vcc, %1, %0\n v_cmp_lt_u32 vcc, %0, %1\n v_cmp_lt_u32 vcc, %1, %0\n v_cmp_lt_u32 vcc, %0, %1\n v_cmp_lt_u32 vcc, %1, %0\n v_cmp_lt_u32 vcc, %0, %1\n v_cmp_lt_u32 vcc, %1, %0"
                                 :: "v"(i0), "v"(i1) : "vcc");
        if (K == 8) asm volatile("v_div_fmas_f64 %0, %8, %9, %10\n v_div_fixup_f64 %1, %9, %10, %11\n v_div_fmas_f64 %2, %10, %11, %8\n v_div_fixup_f64 %3, %8, %11, %9\n v_div_fmas_f64 %4, %8, %9, %10\n v_div_fixup_f64 %5, %9, %10, %11\n v_div_fmas_f64 %6, %10, %11, %8\n v_div_fixup_f64 %7, %8, %11, %9"
                                 : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc");
        if (K == 9) asm volatile("v_cvt_f64_i32 %0, %8\n v_cvt_f64_i32 %1, %9\n v_cvt_f64_i32 %2, %8\n v_cvt_f64_i32 %3, %9\n v_cvt_f64_i32 %4, %8\n v_cvt_f64_i32 %5, %9\n v_cvt_f64_i32 %6, %8\n v_cvt_f64_i32 %7, %9"
                                 : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(i0), "v"(i1));
        if (K == 10) asm volatile("v_cmp_lt_f64 vcc, %0, %1\n v_cmp_lt_f64 vcc, %1, %0\n v_cmp_lt_f64 vcc, %0, %1\n v_cmp_lt_f64 vcc, %1, %0\n v_cmp_lt_f64 vcc, %0, %1\n v_cmp_lt_f64 vcc, %1, %0\n v_cmp_lt_f64 vcc, %0, %1\n v_cmp_lt_f64 vcc, %1, %0"
                                 :: "v"(a0), "v"(a1) : "vcc");
        if (K == 11) { int t0, t1, t2, t3, t4, t5, t6, t7;
            asm volatile("v_readlane_b32 s20, %8, 3\n v_readlane_b32 s21, %9, 5\n v_readlane_b32 s22, %8, 7\n v_readlane_b32 s23, %9, 9\n v_writelane_b32 %0, s20, 1\n v_writelane_b32 %1, s21, 2\n v_writelane_b32 %2, s22, 3\n v_writelane_b32 %3, s23, 4"
                         : "=v"(t0), "=v"(t1), "=v"(t2), "=v"(t3) : "v"(i0), "v"(i1), "0"(i0), "1"(i1), "2"(i0), "3"(i1) : "s20", "s21", "s22", "s23");
            r0 += t0 + t1 + t2 + t3 == 12345 ? 1.0 : 0.0; }
    }
    if (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 == 0.123) out[0] = 1.0;
}
template <int K> void run(const char *name, double clock_ghz, int cus)
{
    double *out; hipMalloc(&out, 8);
    const int blocks = cus * 8;          // 8 blocks x 4 waves per CU = 8 waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<K><<<blocks, 256>>>(out, 1.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0); probe<K><<<blocks, 256>>>(out, 1.5f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)ITER * 8 * 8;      // instructions per SIMD (8 waves x 8 per iteration)
    printf("%-28s %7.3f ms  %5.2f cycles per wave-instruction (at %.1f GHz)\n", name, ms, ms * 1e-3 * clock_ghz * 1e9 / per_simd, clock_ghz);
    hipFree(out);
}
int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const double ghz = p.clockRate * 1e-6; const int cus = p.multiProcessorCount;
    printf("%s: %d CUs, %.2f GHz\n", p.name, cus, ghz);
    run<0>("v_add_f64", ghz, cus); run<1>("v_fma_f64", ghz, cus); run<4>("v_mul_f64", ghz, cus); run<2>("v_cvt_f64_f32", ghz, cus);
    run<9>("v_cvt_f64_i32", ghz, cus); run<5>("v_rcp_f64", ghz, cus); run<8>("v_div_fmas / fixup_f64", ghz, cus);
    run<10>("v_cmp_lt_f64", ghz, cus); run<3>("v_add_u32", ghz, cus); run<6>("v_cndmask_b32", ghz, cus); run<7>("v_cmp_lt_u32", ghz, cus);
    run<11>("v_readlane / v_writelane", ghz, cus);
    return 0;
}
