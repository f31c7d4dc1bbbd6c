// gather_cost.hip's loop spends two runtime modulos per gather (dozens of VALU instructions), which floors every
// pattern at ~32 cycles whatever the L1 does.  This version keeps the dependent chain but only mask / add
// arithmetic (3 VALU per gather), so the number it prints is the memory pipeline's own cost of a 64-lane
// buffer_load_ushort gather at a given occupancy.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/gather_cost2 tools/ubench/gather_cost2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void gather(const uint16_t *tab, unsigned mask, int pattern, int iters, unsigned *out)
{
    const int lane = threadIdx.x & 63;
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(tab), 0, (int)(mask + 1u), 0x00020000);
    unsigned base = ((blockIdx.x * 4 + (threadIdx.x >> 6)) * 40503u) & mask & ~4095u;
    unsigned off;
    switch (pattern) {
    case 0: off = 0; break;                                   // all lanes one address
    case 1: off = lane * 2; break;                            // 1 line contiguous
    case 2: off = (lane >> 3) * 128 + (lane & 7) * 2; break;  // 8 lines x 8 lanes
    case 3: off = (lane >> 2) * 128 + (lane & 3) * 2; break;  // 16 lines, quad-aligned
    case 4: off = (lane & 15) * 128 + (lane >> 4) * 2; break; // 16 lines, lanes of a quad on 4 lines
    case 5: off = lane * 128; break;                          // 64 lines
    case 6: off = ((lane * 37u) & 63u) * 128; break;          // 64 lines permuted
    case 7: off = (lane & 31) * 128 + (lane >> 5) * 2; break; // 32 lines
    default: off = 0xffffffffu; break;                        // 8: out of range
    }
    unsigned acc = 0;
    for (int i = 0; i < iters; i++) {
        const unsigned o = pattern == 8 ? off : ((base + off) & mask);
        const unsigned v = (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rsrc, (int)o, 0, 0);
        acc += v;                       // dependent chain like the ray march
        base = base + 8192u + v;        // table is all zeros: v only carries the dependence
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main()
{
    const unsigned tab_bytes = 4u << 20; // 4 MiB: L2-resident
    uint16_t *tab;
    unsigned *out;
    hipMalloc(&tab, tab_bytes);
    hipMemset(tab, 0, tab_bytes);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, iters = 4000;
    hipMalloc(&out, sizeof(unsigned) * cus * 8 * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const char *names[] = {"1 address", "1 line contiguous", "8 lines x 8 lanes", "16 lines quad-aligned", "16 lines quad-scattered",
                           "64 lines", "64 lines permuted", "32 lines", "out of range"};
    for (int wps = 4; wps <= 8; wps += 4) {
        printf("-- %d waves per SIMD --\n", wps);
        for (int pat = 0; pat <= 8; pat++) {
            hipLaunchKernelGGL(gather, dim3(cus * wps), dim3(256), 0, 0, tab, tab_bytes - 1, pat, 50, out);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(gather, dim3(cus * wps), dim3(256), 0, 0, tab, tab_bytes - 1, pat, iters, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double per_cu = (double)wps * 4 * iters; // wave-gathers per CU
            printf("pattern %d %-26s %8.3f ms  %6.2f ns per wave-gather per CU\n", pat, names[pat], ms, ms * 1e6 / per_cu);
        }
    }
    return 0;
}
