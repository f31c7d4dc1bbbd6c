// Does the SIMD skip the 16-lane passes of a wave64 VALU instruction whose EXEC bits are all zero?
// (If it did, a draining wave of the scan -- a few live rays -- would be cheaper with its live lanes packed into one quarter.)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/exec_mask_rates profiles/probes/ubench/exec_mask_rates.hip && /tmp/exec_mask_rates
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ void k(double *out, int iters, unsigned long long mask)
{
    double a0 = threadIdx.x * 1.5 + 3.25, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b = 1.000001;
    int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, ib = 12345;
    unsigned long long saved;
    asm volatile("s_mov_b64 %0, exec\n s_mov_b64 exec, %1" : "=s"(saved) : "s"(mask));
    for (int it = 0; it < iters; it++) {
        if (OP == 0)
            asm volatile("v_fma_f64 %0, %0, %8, %8\nv_fma_f64 %1, %1, %8, %8\nv_fma_f64 %2, %2, %8, %8\nv_fma_f64 %3, %3, %8, %8\n"
                         "v_fma_f64 %4, %4, %8, %8\nv_fma_f64 %5, %5, %8, %8\nv_fma_f64 %6, %6, %8, %8\nv_fma_f64 %7, %7, %8, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        else
            asm volatile("v_add_u32 %0, %0, %4\nv_add_u32 %1, %1, %4\nv_add_u32 %2, %2, %4\nv_add_u32 %3, %3, %4\n"
                         "v_add_u32 %0, %0, %4\nv_add_u32 %1, %1, %4\nv_add_u32 %2, %2, %4\nv_add_u32 %3, %3, %4\n"
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(ib));
    }
    asm volatile("s_mov_b64 exec, %0" :: "s"(saved));
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + i0 + i1 + i2 + i3;
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, iters = 20000, wps = 8;
    double *out;
    hipMalloc(&out, sizeof(double) * cus * wps * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    struct M { const char *name; unsigned long long m; } masks[] = {
        {"all 64 lanes", ~0ull}, {"lanes 0-31", 0xffffffffull}, {"lanes 0-15", 0xffffull}, {"lanes 16-31", 0xffff0000ull},
        {"lane 0", 1ull}, {"4 lanes, one per 16", 0x0001000100010001ull}, {"lanes 0-15 and 48-63", 0xffff00000000ffffull}};
    for (int op = 0; op < 2; op++)
        for (auto &mk : masks) {
            void (*kern)(double *, int, unsigned long long) = op == 0 ? k<0> : k<1>;
            hipLaunchKernelGGL(kern, dim3(cus * wps), dim3(256), 0, 0, out, 100, mk.m);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(cus * wps), dim3(256), 0, 0, out, iters, mk.m);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double ns_per = ms * 1e6 / ((double)wps * iters * 8);
            printf("%-10s EXEC = %-22s %7.3f ms  %5.2f cycles per wave-instruction per SIMD @2.4 GHz\n", op == 0 ? "v_fma_f64" : "v_add_u32", mk.name, ms, ns_per * 2.4);
        }
    return 0;
}
