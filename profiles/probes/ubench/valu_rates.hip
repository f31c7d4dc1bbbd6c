// Measures issue cost (cycles per wave64 instruction per SIMD, all CUs saturated) of the
// instructions the scan kernel's march loop is made of.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rates tools/ubench/valu_rates.hip && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x

#define BENCH_KERNEL(NAME, ASM8, SETUP)                                                     \
    __global__ void NAME(double *out, int iters)                                            \
    {                                                                                       \
        double a0 = threadIdx.x * 1.5 + 3.25, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;        \
        double a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b = 1.000001;           \
        int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, i4 = i0 + 4, i5 = i0 + 5, i6 = i0 + 6, i7 = i0 + 7; \
        int ib = 12345;                                                                     \
        SETUP;                                                                              \
        for (int it = 0; it < iters; it++) {                                                \
            asm volatile(ASM8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), \
                                "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7)   \
                              : "v"(b), "v"(ib) : "vcc");                                  \
        }                                                                                   \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7; \
    }

// operands: %0..%7 doubles, %8..%15 ints, %16 double b, %17 int ib
BENCH_KERNEL(k_add_f64, "v_add_f64 %0, %0, %16\nv_add_f64 %1, %1, %16\nv_add_f64 %2, %2, %16\nv_add_f64 %3, %3, %16\nv_add_f64 %4, %4, %16\nv_add_f64 %5, %5, %16\nv_add_f64 %6, %6, %16\nv_add_f64 %7, %7, %16\n", )
BENCH_KERNEL(k_mul_f64, "v_mul_f64 %0, %0, %16\nv_mul_f64 %1, %1, %16\nv_mul_f64 %2, %2, %16\nv_mul_f64 %3, %3, %16\nv_mul_f64 %4, %4, %16\nv_mul_f64 %5, %5, %16\nv_mul_f64 %6, %6, %16\nv_mul_f64 %7, %7, %16\n", )
BENCH_KERNEL(k_fma_f64, "v_fma_f64 %0, %0, %16, %16\nv_fma_f64 %1, %1, %16, %16\nv_fma_f64 %2, %2, %16, %16\nv_fma_f64 %3, %3, %16, %16\nv_fma_f64 %4, %4, %16, %16\nv_fma_f64 %5, %5, %16, %16\nv_fma_f64 %6, %6, %16, %16\nv_fma_f64 %7, %7, %16, %16\n", )
BENCH_KERNEL(k_floor_f64, "v_floor_f64 %0, %0\nv_floor_f64 %1, %1\nv_floor_f64 %2, %2\nv_floor_f64 %3, %3\nv_floor_f64 %4, %4\nv_floor_f64 %5, %5\nv_floor_f64 %6, %6\nv_floor_f64 %7, %7\n", )
BENCH_KERNEL(k_cvt_i32_f64, "v_cvt_i32_f64 %8, %0\nv_cvt_i32_f64 %9, %1\nv_cvt_i32_f64 %10, %2\nv_cvt_i32_f64 %11, %3\nv_cvt_i32_f64 %12, %4\nv_cvt_i32_f64 %13, %5\nv_cvt_i32_f64 %14, %6\nv_cvt_i32_f64 %15, %7\n", )
BENCH_KERNEL(k_cvt_f64_i32, "v_cvt_f64_i32 %0, %8\nv_cvt_f64_i32 %1, %9\nv_cvt_f64_i32 %2, %10\nv_cvt_f64_i32 %3, %11\nv_cvt_f64_i32 %4, %12\nv_cvt_f64_i32 %5, %13\nv_cvt_f64_i32 %6, %14\nv_cvt_f64_i32 %7, %15\n", )
BENCH_KERNEL(k_cmp_f64, "v_cmp_lt_f64 vcc, %0, %16\nv_cmp_lt_f64 vcc, %1, %16\nv_cmp_lt_f64 vcc, %2, %16\nv_cmp_lt_f64 vcc, %3, %16\nv_cmp_lt_f64 vcc, %4, %16\nv_cmp_lt_f64 vcc, %5, %16\nv_cmp_lt_f64 vcc, %6, %16\nv_cmp_lt_f64 vcc, %7, %16\n", )
BENCH_KERNEL(k_cmp_u32, "v_cmp_lt_u32 vcc, %8, %17\nv_cmp_lt_u32 vcc, %9, %17\nv_cmp_lt_u32 vcc, %10, %17\nv_cmp_lt_u32 vcc, %11, %17\nv_cmp_lt_u32 vcc, %12, %17\nv_cmp_lt_u32 vcc, %13, %17\nv_cmp_lt_u32 vcc, %14, %17\nv_cmp_lt_u32 vcc, %15, %17\n", )
BENCH_KERNEL(k_add_u32, "v_add_u32 %8, %8, %17\nv_add_u32 %9, %9, %17\nv_add_u32 %10, %10, %17\nv_add_u32 %11, %11, %17\nv_add_u32 %12, %12, %17\nv_add_u32 %13, %13, %17\nv_add_u32 %14, %14, %17\nv_add_u32 %15, %15, %17\n", )
BENCH_KERNEL(k_cndmask, "v_cndmask_b32 %8, %8, %17, vcc\nv_cndmask_b32 %9, %9, %17, vcc\nv_cndmask_b32 %10, %10, %17, vcc\nv_cndmask_b32 %11, %11, %17, vcc\nv_cndmask_b32 %12, %12, %17, vcc\nv_cndmask_b32 %13, %13, %17, vcc\nv_cndmask_b32 %14, %14, %17, vcc\nv_cndmask_b32 %15, %15, %17, vcc\n", )
BENCH_KERNEL(k_cndmask_sgpr, "v_cndmask_b32_e64 %8, %8, %17, s[20:21]\nv_cndmask_b32_e64 %9, %9, %17, s[20:21]\nv_cndmask_b32_e64 %10, %10, %17, s[20:21]\nv_cndmask_b32_e64 %11, %11, %17, s[20:21]\nv_cndmask_b32_e64 %12, %12, %17, s[20:21]\nv_cndmask_b32_e64 %13, %13, %17, s[20:21]\nv_cndmask_b32_e64 %14, %14, %17, s[20:21]\nv_cndmask_b32_e64 %15, %15, %17, s[20:21]\n", )
BENCH_KERNEL(k_cmp_cndmask, "v_cmp_lt_u32 vcc, %8, %17\nv_cndmask_b32 %8, %8, %17, vcc\nv_cmp_lt_u32 vcc, %9, %17\nv_cndmask_b32 %9, %9, %17, vcc\nv_cmp_lt_u32 vcc, %10, %17\nv_cndmask_b32 %10, %10, %17, vcc\nv_cmp_lt_u32 vcc, %11, %17\nv_cndmask_b32 %11, %11, %17, vcc\n", )
BENCH_KERNEL(k_med3_i32, "v_med3_i32 %8, %8, -1, %17\nv_med3_i32 %9, %9, -1, %17\nv_med3_i32 %10, %10, -1, %17\nv_med3_i32 %11, %11, -1, %17\nv_med3_i32 %12, %12, -1, %17\nv_med3_i32 %13, %13, -1, %17\nv_med3_i32 %14, %14, -1, %17\nv_med3_i32 %15, %15, -1, %17\n", )
BENCH_KERNEL(k_mad_u64_u32, "v_mad_u64_u32 %0, vcc, %8, %17, %0\nv_mad_u64_u32 %1, vcc, %9, %17, %1\nv_mad_u64_u32 %2, vcc, %10, %17, %2\nv_mad_u64_u32 %3, vcc, %11, %17, %3\nv_mad_u64_u32 %4, vcc, %12, %17, %4\nv_mad_u64_u32 %5, vcc, %13, %17, %5\nv_mad_u64_u32 %6, vcc, %14, %17, %6\nv_mad_u64_u32 %7, vcc, %15, %17, %7\n", )
BENCH_KERNEL(k_mad_u32_u24, "v_mad_u32_u24 %8, %8, %17, %8\nv_mad_u32_u24 %9, %9, %17, %9\nv_mad_u32_u24 %10, %10, %17, %10\nv_mad_u32_u24 %11, %11, %17, %11\nv_mad_u32_u24 %12, %12, %17, %12\nv_mad_u32_u24 %13, %13, %17, %13\nv_mad_u32_u24 %14, %14, %17, %14\nv_mad_u32_u24 %15, %15, %17, %15\n", )
BENCH_KERNEL(k_mul_lo_u32, "v_mul_lo_u32 %8, %8, %17\nv_mul_lo_u32 %9, %9, %17\nv_mul_lo_u32 %10, %10, %17\nv_mul_lo_u32 %11, %11, %17\nv_mul_lo_u32 %12, %12, %17\nv_mul_lo_u32 %13, %13, %17\nv_mul_lo_u32 %14, %14, %17\nv_mul_lo_u32 %15, %15, %17\n", )
BENCH_KERNEL(k_lshl_add_u32, "v_lshl_add_u32 %8, %8, 3, %17\nv_lshl_add_u32 %9, %9, 3, %17\nv_lshl_add_u32 %10, %10, 3, %17\nv_lshl_add_u32 %11, %11, 3, %17\nv_lshl_add_u32 %12, %12, 3, %17\nv_lshl_add_u32 %13, %13, 3, %17\nv_lshl_add_u32 %14, %14, 3, %17\nv_lshl_add_u32 %15, %15, 3, %17\n", )
BENCH_KERNEL(k_cvt_f32_f64, "v_cvt_f32_f64 %8, %0\nv_cvt_f32_f64 %9, %1\nv_cvt_f32_f64 %10, %2\nv_cvt_f32_f64 %11, %3\nv_cvt_f32_f64 %12, %4\nv_cvt_f32_f64 %13, %5\nv_cvt_f32_f64 %14, %6\nv_cvt_f32_f64 %15, %7\n", )
BENCH_KERNEL(k_min_f64, "v_min_f64 %0, %0, %16\nv_min_f64 %1, %1, %16\nv_min_f64 %2, %2, %16\nv_min_f64 %3, %3, %16\nv_min_f64 %4, %4, %16\nv_min_f64 %5, %5, %16\nv_min_f64 %6, %6, %16\nv_min_f64 %7, %7, %16\n", )
BENCH_KERNEL(k_fract_f64, "v_fract_f64 %0, %0\nv_fract_f64 %1, %1\nv_fract_f64 %2, %2\nv_fract_f64 %3, %3\nv_fract_f64 %4, %4\nv_fract_f64 %5, %5\nv_fract_f64 %6, %6\nv_fract_f64 %7, %7\n", )
BENCH_KERNEL(k_ldexp_f64, "v_ldexp_f64 %0, %0, 4\nv_ldexp_f64 %1, %1, 4\nv_ldexp_f64 %2, %2, 4\nv_ldexp_f64 %3, %3, 4\nv_ldexp_f64 %4, %4, 4\nv_ldexp_f64 %5, %5, 4\nv_ldexp_f64 %6, %6, 4\nv_ldexp_f64 %7, %7, 4\n", )
BENCH_KERNEL(k_salu, "s_add_u32 s20, s20, 1\ns_add_u32 s21, s21, 1\ns_add_u32 s22, s22, 1\ns_add_u32 s23, s23, 1\ns_add_u32 s20, s20, 1\ns_add_u32 s21, s21, 1\ns_add_u32 s22, s22, 1\ns_add_u32 s23, s23, 1\n", )

typedef void (*kern_t)(double *, int);
struct Entry { const char *name; kern_t k; };

int main()
{
    Entry es[] = {{"v_add_f64", k_add_f64}, {"v_mul_f64", k_mul_f64}, {"v_fma_f64", k_fma_f64}, {"v_floor_f64", k_floor_f64},
                  {"v_fract_f64", k_fract_f64}, {"v_ldexp_f64", k_ldexp_f64}, {"v_min_f64", k_min_f64},
                  {"v_cvt_i32_f64", k_cvt_i32_f64}, {"v_cvt_f64_i32", k_cvt_f64_i32}, {"v_cvt_f32_f64", k_cvt_f32_f64},
                  {"v_cmp_lt_f64", k_cmp_f64}, {"v_cmp_lt_u32", k_cmp_u32}, {"v_add_u32", k_add_u32}, {"v_cndmask_b32", k_cndmask},
                  {"v_cndmask_sgprmask", k_cndmask_sgpr}, {"cmp+cndmask x4", k_cmp_cndmask}, {"v_med3_i32", k_med3_i32}, {"v_mad_u64_u32", k_mad_u64_u32}, {"v_mad_u32_u24", k_mad_u32_u24}, {"v_mul_lo_u32", k_mul_lo_u32},
                  {"v_lshl_add_u32", k_lshl_add_u32}, {"s_add_u32", k_salu}};
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, iters = 20000;
    double *out;
    hipMalloc(&out, sizeof(double) * cus * 8 * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("device %s, %d CUs, clock %d kHz\n", p.name, cus, p.clockRate);
    for (int wps : {4, 8}) {  // waves per SIMD
        printf("-- %d wave(s) per SIMD --\n", wps);
        for (auto &e : es) {
            dim3 grid(cus * wps), block(256); // 256 threads = 4 waves = one per SIMD
            hipLaunchKernelGGL(e.k, grid, block, 0, 0, out, 100);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(e.k, grid, block, 0, 0, out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            // per SIMD: wps waves x iters x 8 instructions
            const double instr_per_simd = (double)wps * iters * 8;
            const double ns_per = ms * 1e6 / instr_per_simd;
            printf("%-16s %7.3f ms  %6.2f ns/wave-instr/SIMD  = %5.2f cycles @2.4GHz\n", e.name, ms, ns_per, ns_per * 2.4);
        }
    }
    return 0;
}
