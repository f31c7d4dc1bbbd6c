// Does hipExtAnyOrderLaunch let a kernel start before its predecessor in the same stream has finished (gfx950)?
// Kernel A spins ~200 us and stamps its end; kernel B (launched behind it) stamps its start.  B < A_end => overlap.
// Also: same two kernels on two streams (expected to overlap), for reference.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin_kernel(unsigned long long *out, unsigned long long ticks)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) { out[0] = t0; out[1] = wall_clock64(); }
}
__global__ void stamp_kernel(unsigned long long *out)
{
    if (threadIdx.x == 0) out[2] = wall_clock64();
}
int main()
{
    unsigned long long *d, h[3];
    hipMalloc(&d, 3 * sizeof(*d));
    hipStream_t s, s2;
    hipStreamCreate(&s);
    hipStreamCreate(&s2);
    for (int mode = 0; mode < 3; mode++) {
        for (int rep = 0; rep < 3; rep++) {
            hipMemset(d, 0, 3 * sizeof(*d));
            hipDeviceSynchronize();
            void *a1[2]; unsigned long long ticks = 20000; // 200 us at 100 MHz
            a1[0] = &d; a1[1] = &ticks;
            void *a2[1] = {&d};
            hipLaunchKernel((const void *)spin_kernel, dim3(1), dim3(64), a1, 0, s);
            if (mode == 0) hipLaunchKernel((const void *)stamp_kernel, dim3(1), dim3(64), a2, 0, s);
            else if (mode == 1) hipExtLaunchKernel((const void *)stamp_kernel, dim3(1), dim3(64), a2, 0, s, nullptr, nullptr, hipExtAnyOrderLaunch);
            else hipLaunchKernel((const void *)stamp_kernel, dim3(1), dim3(64), a2, 0, s2);
            hipDeviceSynchronize();
            hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            printf("%s: A ran %.1f us; B started %.1f us after A started (%s)\n",
                   mode == 0 ? "same stream, plain      " : mode == 1 ? "same stream, AnyOrder   " : "two streams             ",
                   (h[1] - h[0]) / 100.0, ((long long)h[2] - (long long)h[0]) / 100.0, h[2] < h[1] ? "OVERLAP" : "serial");
        }
    }
    return 0;
}
