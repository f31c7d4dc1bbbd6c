// How does the L1 (TCP) price a 64-lane buffer_load_ushort gather?  Patterns differ in
// how many distinct 128-B lines a wave touches and how lanes are grouped.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/gather_cost tools/ubench/gather_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void gather(const uint16_t *tab, unsigned tab_bytes, int pattern, int iters, unsigned *out)
{
    const int lane = threadIdx.x & 63;
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(tab), 0, (int)tab_bytes, 0x00020000);
    unsigned base = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 65536u % (tab_bytes / 2);
    unsigned off;
    switch (pattern) {
    case 0: off = 0; break;                                   // all lanes one address
    case 1: off = lane * 2; break;                            // 128 contiguous bytes: 1 line
    case 2: off = (lane >> 2) * 128 + (lane & 3) * 2; break;  // each quad its own line (16 lines)
    case 3: off = lane * 128; break;                          // every lane its own line (64 lines)
    case 4: off = (lane & 15) * 128 + (lane >> 4) * 2; break; // 16 lines, but lanes of a quad on 4 different lines
    case 5: off = (lane >> 3) * 128 + (lane & 7) * 2; break;  // 8 lines, 8 lanes each
    case 6: off = (lane & 7) * 128 + (lane >> 3) * 2; break;  // 8 lines, strided lanes
    case 7: off = (lane >> 4) * 128 + (lane & 15) * 2; break; // 4 lines
    case 8: off = 0xffffffffu; break;                         // out of range (no access)
    case 9: off = ((lane * 37u) & 63u) * 2; break;            // one 128-B line, scattered 2-B positions
    case 10: off = ((lane * 37u) & 31u) * 2; break;           // one 64-B half line, scattered
    case 11: off = ((lane * 37u) & 15u) * 2; break;           // one 32-B sector, scattered
    case 12: off = (lane >> 4) * 128 + ((lane * 37u) & 63u) * 2; break; // one line per 16-lane group, scattered inside
    case 13: off = (lane >> 4) * 128 + ((lane * 5u) & 15u) * 8; break;  // one line per group, 8-B apart positions
    default: off = (lane >> 4) * 128 + (lane & 1) * 64 + ((lane >> 1) & 7) * 2; break; // 14: two 64-B halves per group
    }
    unsigned acc = 0;
    for (int i = 0; i < iters; i++) {
        unsigned o = pattern == 8 ? off : (base * 2 + off + (unsigned)(acc & 1u) * 0u) % (tab_bytes - 8192);
        unsigned v = (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rsrc, (int)o, 0, 0);
        acc += v; // dependent chain like the ray march
        base = (base + 4096 + (v & 1u)) % (tab_bytes / 2 - 65536);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main()
{
    const unsigned tab_bytes = 8u << 20; // 8 MiB: L2-resident
    uint16_t *tab;
    unsigned *out;
    hipMalloc(&tab, tab_bytes);
    hipMemset(tab, 0, tab_bytes);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, wps = 6, iters = 4000;
    hipMalloc(&out, sizeof(unsigned) * cus * wps * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const char *names[] = {"1 address", "1 line contiguous", "16 lines, quad-aligned", "64 lines", "16 lines, quad-scattered",
                           "8 lines x 8 lanes", "8 lines strided lanes", "4 lines x 16 lanes", "out of range",
                           "1 line scattered", "1 half-line scattered", "1 sector scattered", "line/group scattered",
                           "line/group 8B apart", "2 halves per group"};
    for (int pat = 0; pat <= 14; pat++) {
        hipLaunchKernelGGL(gather, dim3(cus * wps), dim3(256), 0, 0, tab, tab_bytes, pat, 50, out);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(gather, dim3(cus * wps), dim3(256), 0, 0, tab, tab_bytes, pat, iters, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        // per CU: wps*4 waves x iters gathers
        const double per_cu = (double)wps * 4 * iters;
        printf("pattern %d %-28s %8.3f ms  %7.1f ns per wave-gather per CU (=%6.1f cyc @2.4GHz)\n", pat, names[pat], ms,
               ms * 1e6 / per_cu, ms * 1e6 / per_cu * 2.4);
    }
    return 0;
}
