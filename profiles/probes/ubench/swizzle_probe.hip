// swizzle_probe.hip -- does gfx950 still implement the GFX9 SWIZZLED structured buffer (descriptor bit 63, ELEMENT_SIZE,
// INDEX_STRIDE), which address does `buffer_load_ushort ... idxen offen` form from (index, offset), and which accesses does
// its range check answer with 0?  (Round 5: the scan kernel's march spends 6 of its 21 VALU instructions on clamping the
// cell (row, column) and forming the byte offset of an 8x8-cell block layout; a swizzled descriptor with ELEMENT_SIZE 16 B,
// INDEX_STRIDE 8 and index = row, offset = 2 * column forms exactly that layout in the address unit, and its range check
// -- index >= num_records or offset >= stride -- is the clamp.)
//   hipcc --offload-arch=gfx950 -O3 -o swizzle_probe swizzle_probe.hip && ./swizzle_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ inline unsigned load_u16_struct(u32x4 rsrc, unsigned index, unsigned offset)
{
    unsigned v;
    unsigned long long io = (unsigned long long)index | ((unsigned long long)offset << 32);
    asm volatile("s_nop 4\n\tbuffer_load_ushort %0, %1, %2, 0 idxen offen\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(io), "s"(rsrc) : "memory");
    return v;
}

__device__ inline u32x4 make_rsrc(const void *p, unsigned stride, unsigned num_records, unsigned w1_extra, unsigned w3)
{
    const unsigned long long a = (unsigned long long)p;
    u32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((unsigned)a);
    r.y = __builtin_amdgcn_readfirstlane(((unsigned)(a >> 32) & 0xffffu) | (stride << 16) | w1_extra);
    r.z = __builtin_amdgcn_readfirstlane(num_records);
    r.w = __builtin_amdgcn_readfirstlane(w3);
    return r;
}

// two tables with the low / high 16 bits of each cell's own element number: the pair says where a load landed
__global__ void probe(const uint16_t *lo, const uint16_t *hi, unsigned stride, unsigned num_records, unsigned w1_extra, unsigned w3,
                      const unsigned *idx, const unsigned *off, unsigned *out, int n)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const u32x4 rl = make_rsrc(lo, stride, num_records, w1_extra, w3), rh = make_rsrc(hi, stride, num_records, w1_extra, w3);
    if (t < n) {
        const unsigned a = load_u16_struct(rl, idx[t], off[t]), b = load_u16_struct(rh, idx[t], off[t]);
        out[t] = a | (b << 16);
    }
}

// rate: a dependent chain of gathers as in the march; identical footprints in every mode (the swizzled 8x8-block layout)
//   MODE 0: raw descriptor, `offen`, the swizzled byte offset formed by VALU instructions
//   MODE 1: linear structured descriptor (stride = one band of 8 rows), `idxen offen`: index = row >> 3, offset = the rest (VALU)
//   MODE 2: swizzled structured descriptor, `idxen offen`: index = row, offset = 2 * column
//   PAT 0: the wave's 64 lanes inside one 8x8 block (1 line); 1: each 16-lane group in its own block (4 lines);
//       2: each quad of lanes in its own block (16 lines); 3: every lane its own block (64 lines)
template <int MODE, int PAT>
__global__ void rate(const uint16_t *tab, unsigned bytes, unsigned stride, unsigned rows, int iters, unsigned *out)
{
    const int lane = threadIdx.x & 63;
    const unsigned wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    unsigned r0 = (wave * 40u) & 1016u, c0 = (wave * 72u) & 1016u; // block-aligned
    unsigned dr, dc;
    if (PAT == 0) { dr = lane >> 3; dc = lane & 7; }
    else if (PAT == 1) { dr = (lane >> 2) & 3; dc = (lane & 3) + 8 * (lane >> 4); }
    else if (PAT == 2) { dr = (lane & 1) + 8 * ((lane >> 4) & 3); dc = ((lane >> 1) & 1) + 8 * ((lane >> 2) & 3); }
    else { dr = 8 * (lane >> 3); dc = 8 * (lane & 7); }
    unsigned acc = 0;
    const u32x4 rl = make_rsrc(tab, stride * 8u, (rows + 7u) >> 3, 0u, 0x00020000u);
    const u32x4 rs = make_rsrc(tab, stride, rows, 0x80000000u, 0x00020000u);
    auto rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(tab), 0, (int)bytes, 0x00020000);
    for (int i = 0; i < iters; i++) {
        const unsigned r = (r0 + dr) & 1023u, c = (c0 + dc) & 1023u;
        unsigned v;
        if (MODE == 0) {
            const unsigned o = ((r >> 3) * stride + (c >> 1) * 4u) * 8u + (r & 7u) * 4u + (c & 1u) * 2u;
            v = (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rr, (int)o, 0, 0);
        } else if (MODE == 1) {
            v = load_u16_struct(rl, r >> 3, (c >> 1) * 32u + (r & 7u) * 4u + (c & 1u) * 2u);
        } else {
            v = load_u16_struct(rs, r, c * 2u);
        }
        acc += v; r0 = (r0 + 24u + v) & 1016u; c0 = (c0 + 40u + v) & 1016u; // table is all zeros: v only carries the dependence
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int MODE, int PAT>
static void time_rate(const uint16_t *tab, unsigned bytes, unsigned stride, unsigned rows, int cus, unsigned *dout, hipEvent_t e0, hipEvent_t e1)
{
    const int iters = 4000;
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((rate<MODE, PAT>), dim3(cus * 8), dim3(256), 0, 0, tab, bytes, stride, rows, iters, dout);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    static const char *mn[] = {"raw offen + VALU offset", "linear idxen offen", "swizzled idxen offen"};
    static const char *pn[] = {"1 line", "4 lines (one per 16-lane group)", "16 lines (one per quad)", "64 lines"};
    printf("rate %-26s %-34s %.3f ms  %.2f ns per wave-gather per CU\n", mn[MODE], pn[PAT], best, best * 1e6 / iters / 32.0); fflush(stdout);
}

int main(int argc, char **argv)
{
    const bool extreme = argc > 1; // also the queries that an absent range check would send gigabytes away (a fault)
    const unsigned W = 1024, H = 1024, stride = W * 2; // a 1024 x 1024 table of u16: 2 KiB rows
    const size_t cells = (size_t)W * H;
    std::vector<uint16_t> hlo(cells), hhi(cells);
    for (size_t i = 0; i < cells; i++) { hlo[i] = (uint16_t)i; hhi[i] = (uint16_t)(i >> 16); }
    uint16_t *lo, *hi;
    hipMalloc(&lo, cells * 2 + (8 << 20)); hipMalloc(&hi, cells * 2 + (8 << 20));
    hipMemset(lo, 0xee, cells * 2 + (8 << 20)); hipMemset(hi, 0xee, cells * 2 + (8 << 20));
    hipMemcpy(lo, hlo.data(), cells * 2, hipMemcpyHostToDevice);
    hipMemcpy(hi, hhi.data(), cells * 2, hipMemcpyHostToDevice);
    struct Q { unsigned i, o; const char *what; };
    // in-range queries first; then the ones only a range check keeps inside the allocation, mildest first (one launch
    // each, printed and flushed before the next: a fault names its query)
    std::vector<Q> qs = {
        {0, 0, "origin"}, {0, 2, "col 1"}, {0, 4, "col 2"}, {0, 14, "col 7"}, {0, 16, "col 8"}, {1, 0, "row 1"}, {7, 14, "row 7 col 7"},
        {8, 0, "row 8"}, {9, 18, "row 9 col 9"}, {100, 600, "row 100 col 300"}, {1023, 2046, "last cell"}, {5, 1, "odd offset 1"},
        {1024, 0, "index == num_records"}, {5, 2048, "offset == stride"}, {5, 2050, "offset == stride + 2"},
        {1023, 2048, "last row, offset == stride"}, {1030, 40, "index > num_records"}, {5, 6000, "offset ~ 3 strides"},
        {0xffffffffu, 0, "index -1"}, {5, 0xfffffffeu, "offset -2"}, {0x7fffffffu, 0, "index 2^31-1"}, {5, 0x7ffffffeu, "offset 2^31-2"},
    };
    const int n = (int)qs.size();
    unsigned *di, *dof, *dout;
    hipMalloc(&di, 4); hipMalloc(&dof, 4); hipMalloc(&dout, 4 << 20);
    struct Cfg { unsigned w1, w3; int nq; const char *name; };
    // word 3: data format 32 (bits 15-18 = 4) as the shipped kernel; INDEX_STRIDE bits 21-22 (0..3 = 8, 16, 32, 64).  GFX9 has
    // no ELEMENT_SIZE field (the swizzle's element is a dword); bits 19-20 are USER_VM_ENABLE / USER_VM_MODE there -- setting
    // them is a memory access fault (the first version of this probe did).
    const Cfg cfgs[] = {
        {0x80000000u, 0x00020000u | (0u << 21), extreme ? n : 18, "swizzled, index stride 8"},
        {0x80000000u, 0x00020000u | (1u << 21), 12, "swizzled, index stride 16"},
        {0x00000000u, 0x00020000u, 18, "linear structured (swizzle off)"},
    };
    for (const Cfg &c : cfgs) {
        printf("== %s: stride %u, num_records %u\n", c.name, stride, H);
        for (int k = 0; k < c.nq; k++) {
            const unsigned idx = qs[k].i, off = qs[k].o;
            hipMemcpy(di, &idx, 4, hipMemcpyHostToDevice);
            hipMemcpy(dof, &off, 4, hipMemcpyHostToDevice);
            printf("  %-28s idx %10u off %10u -> ", qs[k].what, idx, off); fflush(stdout);
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, lo, hi, stride, H, c.w1, c.w3, di, dof, dout, 1);
            unsigned res = 0;
            if (hipMemcpy(&res, dout, 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("launch failed\n"); return 1; }
            // linear: index * stride + offset; swizzled (dword elements, IS): ((idx / IS) * stride + (off / 4) * 4) * IS + (idx % IS) * 4 + off % 4
            const unsigned es = 4u, is = 8u << ((c.w3 >> 21) & 3u);
            unsigned long long lin = (unsigned long long)idx * stride + off;
            unsigned long long swz = ((unsigned long long)(idx / is) * stride + (unsigned long long)(off / es) * es) * is + (idx % is) * es + off % es;
            printf("element %10u   (linear model %llu, swizzle model %llu)\n", res, lin / 2, swz / 2); fflush(stdout);
        }
    }
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    hipMemset(lo, 0, cells * 2);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const unsigned bytes = (unsigned)(cells * 2);
#define ROW(P) time_rate<0, P>(lo, bytes, stride, H, cus, dout, e0, e1); time_rate<1, P>(lo, bytes, stride, H, cus, dout, e0, e1); time_rate<2, P>(lo, bytes, stride, H, cus, dout, e0, e1);
    ROW(0) ROW(1) ROW(2) ROW(3)
    return 0;
}
