// f110_mapgen.h -- the map pipeline on the GPU (map tooling): occupancy mask -> exact squared
// Euclidean distance transform -> rank-coded cell table + fp64 LUT + fp64 distance table, i.e. everything
// ScanSimulator2D.set_map (reference laser_models.py:383-427, get_dt :40-53) prepares, in ~1 ms instead of the
// host's ~0.3 s, so that a map can be swapped per episode (domain randomisation) without stalling the step path.
// Integer arithmetic throughout: d2 is exact, and resolution*sqrt(d2) (IEEE sqrt and multiply) reproduces
// scipy.ndimage.distance_transform_edt bit for bit like the host pipeline it replaces.
#pragma once
#include "f110_kernels.h"

#pragma clang fp contract(off)

namespace f110 {

constexpr unsigned EDT_NONE = 0xffffffffu; // column without any occupied cell

// Pass 1, one lane per column (coalesced across x): g = distance along the column to the nearest zero cell.
#if defined(F110_UNIT_MAPS)
static __global__ __launch_bounds__(256) void edt_columns_kernel(const uint8_t *mask, int H, int W, unsigned *g)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= W) return;
    unsigned d = EDT_NONE;
    for (int y = 0; y < H; y++) {
        const size_t i = (size_t)y * W + x;
        d = mask[i] ? (d == EDT_NONE ? EDT_NONE : d + 1) : 0u;
        g[i] = d;
    }
    d = EDT_NONE;
    for (int y = H - 1; y >= 0; y--) {
        const size_t i = (size_t)y * W + x;
        d = mask[i] ? (d == EDT_NONE ? EDT_NONE : d + 1) : 0u;
        const unsigned up = g[i];
        g[i] = d < up ? d : up;
    }
}
#endif

// Pass 2, one workgroup per row with the row's g^2 in LDS: d2(x) = min over x' of (x - x')^2 + g(x')^2, searched
// outwards from x and stopped as soon as dx^2 alone reaches the best value -- O(distance) LDS reads per cell,
// exact, no lower-envelope bookkeeping (the sequential formulation of the host code).
#if defined(F110_UNIT_MAPS)
static __global__ __launch_bounds__(256) void edt_rows_kernel(const unsigned *g, int H, int W, unsigned *d2, unsigned *max_d2)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    unsigned *g2 = reinterpret_cast<unsigned *>(s_raw);
    const int y = blockIdx.x;
    for (int x = threadIdx.x; x < W; x += blockDim.x) {
        const unsigned v = g[(size_t)y * W + x];
        g2[x] = v == EDT_NONE ? EDT_NONE : v * v; // v <= 32768: fits
    }
    __syncthreads();
    unsigned local_max = 0;
    for (int x = threadIdx.x; x < W; x += blockDim.x) {
        unsigned long long best = g2[x] == EDT_NONE ? ~0ull : g2[x];
        for (unsigned dx = 1; (unsigned long long)dx * dx < best && (x >= (int)dx || x + (int)dx < W); dx++) {
            const unsigned long long q = (unsigned long long)dx * dx;
            if (x >= (int)dx) {
                const unsigned v = g2[x - dx];
                if (v != EDT_NONE && q + v < best) best = q + v;
            }
            if (x + (int)dx < W) {
                const unsigned v = g2[x + dx];
                if (v != EDT_NONE && q + v < best) best = q + v;
            }
        }
        const unsigned out = (unsigned)best; // the caller guarantees an occupied cell exists and H, W <= 32768
        d2[(size_t)y * W + x] = out;
        local_max = out > local_max ? out : local_max;
    }
    if (max_d2) {
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned o = __shfl_down(local_max, off, 64);
            local_max = o > local_max ? o : local_max;
        }
        if ((threadIdx.x & 63) == 0) atomicMax(max_d2, local_max);
    }
}
#endif

// presence bitmap of the distinct d2 values
#if defined(F110_UNIT_MAPS)
static __global__ __launch_bounds__(256) void d2_mark_kernel(const unsigned *d2, size_t n, unsigned *bits)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned v = d2[i];
    const unsigned m = 1u << (v & 31);
    if (!(bits[v >> 5] & m)) atomicOr(&bits[v >> 5], m); // most values are already marked: read before the atomic
}
#endif

// exclusive prefix of per-word popcounts, three small kernels (1024 words per block)
constexpr int SCAN_BLOCK_WORDS = 1024;
#if defined(F110_UNIT_MAPS)
static __global__ __launch_bounds__(256) void rank_block_sums_kernel(const unsigned *bits, int n_words, unsigned *block_sums)
{
    __shared__ unsigned s_part[4];
    unsigned c = 0;
    for (int k = 0; k < 4; k++) {
        const int w = blockIdx.x * SCAN_BLOCK_WORDS + k * 256 + threadIdx.x;
        if (w < n_words) c += __popc(bits[w]);
    }
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = s_part[0] + s_part[1] + s_part[2] + s_part[3];
}
#endif

#if defined(F110_UNIT_MAPS)
static __global__ __launch_bounds__(64) void rank_scan_sums_kernel(unsigned *block_sums, int n_blocks, unsigned *total)
{
    // one wave, sequential over chunks of 64: n_blocks is small (words / 1024)
    unsigned run = 0;
    for (int base = 0; base < n_blocks; base += 64) {
        const int i = base + threadIdx.x;
        const unsigned v = i < n_blocks ? block_sums[i] : 0u;
        unsigned incl = v;
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned t = __shfl_up(incl, off, 64);
            if ((int)threadIdx.x >= off) incl += t;
        }
        if (i < n_blocks) block_sums[i] = run + incl - v;
        run += __shfl(incl, 63, 64);
    }
    if (threadIdx.x == 0) *total = run;
}
#endif

#if defined(F110_UNIT_MAPS)
static __global__ __launch_bounds__(256) void rank_word_prefix_kernel(const unsigned *bits, int n_words, const unsigned *block_sums,
                                                               unsigned *word_prefix)
{
    __shared__ unsigned s_wave[4];
    // thread t owns words base + 4t .. 4t+3 (consecutive, so the in-block order is the word order)
    const int w0 = blockIdx.x * SCAN_BLOCK_WORDS + threadIdx.x * 4;
    unsigned c[4], sum = 0;
    for (int k = 0; k < 4; k++) { c[k] = (w0 + k < n_words) ? __popc(bits[w0 + k]) : 0u; sum += c[k]; }
    unsigned incl = sum;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    unsigned base = block_sums[blockIdx.x];
    for (int w = 0; w < wave; w++) base += s_wave[w];
    unsigned run = base + incl - sum;
    for (int k = 0; k < 4; k++) {
        if (w0 + k < n_words) word_prefix[w0 + k] = run;
        run += c[k];
    }
}
#endif

__device__ inline unsigned d2_rank(const unsigned *bits, const unsigned *word_prefix, unsigned v)
{
    return word_prefix[v >> 5] + __popc(bits[v >> 5] & ((1u << (v & 31)) - 1u));
}

// cell codes (strip layout with border, see MapDev), second rank table and the fp64 distance table
#if defined(F110_UNIT_MAPS)
static __global__ __launch_bounds__(256) void map_encode_kernel(const unsigned *d2, int H, int W, int Hp, const unsigned *bits,
                                                         const unsigned *word_prefix, double res, uint16_t *cells,
                                                         uint16_t *cells_far, double *dt)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)H * W) return;
    const unsigned v = d2[i];
    const unsigned rank = d2_rank(bits, word_prefix, v);
    const size_t t = cell_elem((int)(i / W), (int)(i % W), Hp);
    cells[t] = (uint16_t)cell_code(rank);
    cells_far[t] = (uint16_t)(rank < CODE_ESC ? rank : CODE_ESC);
    dt[i] = res * sqrt((double)v);
}
#endif

#if defined(F110_UNIT_MAPS)
static __global__ __launch_bounds__(256) void map_fill_border_kernel(uint16_t *cells, uint16_t *cells_far, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    cells[i] = 0; // the border and the padding: code 0 = LDS slot 0 = dt[-1, -1]
    cells_far[i] = 0;
}
#endif

// lut[rank] = resolution * sqrt(d2) for every distinct d2 with rank < n_lut
#if defined(F110_UNIT_MAPS)
static __global__ __launch_bounds__(256) void map_lut_kernel(const unsigned *bits, int n_words, const unsigned *word_prefix, double res,
                                                      double *lut, unsigned n_lut)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    unsigned b = bits[w], rank = word_prefix[w];
    while (b && rank < n_lut) {
        const int bit = __ffs(b) - 1;
        b &= b - 1;
        lut[rank++] = res * sqrt((double)((unsigned)w * 32u + (unsigned)bit));
    }
}
#endif


// ---- track walls from a centre line (red_gym_amd/trackgen.py): pixel (ix, iy) is a wall (0) iff the
// distance of its centre to the polyline is within half_stroke of `offset`, free (1) otherwise.  The reference
// strokes the two shapely offset curves with matplotlib (random_trackgen.py:161-193); the curves at distance
// `offset` from a closed line are exactly that level set, drawn here without a geometry library.
// One lane per pixel, the segments staged through LDS in chunks; fp64, plain mul/add.
struct TrackSeg { double ax, ay, ex, ey, inv_l2; };
constexpr int TRACK_CHUNK = 256;

#if defined(F110_UNIT_MAPS)
static __global__ __launch_bounds__(256) void track_mask_kernel(const double *pts, int n_pts, int closed, int H, int W, double x0,
                                                         double y0, double pixel, double offset, double half_stroke,
                                                         uint8_t *mask)
{
    __shared__ TrackSeg s_seg[TRACK_CHUNK];
    const int ix = blockIdx.x * 16 + (threadIdx.x & 15), iy = blockIdx.y * 16 + (threadIdx.x >> 4);
    const double px = x0 + ((double)ix + 0.5) * pixel, py = y0 + ((double)iy + 0.5) * pixel;
    const int n_seg = closed ? n_pts : n_pts - 1;
    double best = __builtin_inf();
    for (int base = 0; base < n_seg; base += TRACK_CHUNK) {
        const int m = min(TRACK_CHUNK, n_seg - base);
        __syncthreads();
        if ((int)threadIdx.x < m) {
            const int i = base + threadIdx.x, j = (i + 1) % n_pts;
            TrackSeg sg;
            sg.ax = pts[2 * i]; sg.ay = pts[2 * i + 1];
            sg.ex = pts[2 * j] - sg.ax; sg.ey = pts[2 * j + 1] - sg.ay;
            const double l2 = sg.ex * sg.ex + sg.ey * sg.ey;
            sg.inv_l2 = l2 > 0.0 ? 1.0 / l2 : 0.0;
            s_seg[threadIdx.x] = sg;
        }
        __syncthreads();
        for (int k = 0; k < m; k++) {
            const TrackSeg sg = s_seg[k];
            const double rx = px - sg.ax, ry = py - sg.ay;
            double t = (rx * sg.ex + ry * sg.ey) * sg.inv_l2;
            t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
            const double dx = rx - t * sg.ex, dy = ry - t * sg.ey;
            const double d2 = dx * dx + dy * dy;
            best = d2 < best ? d2 : best;
        }
    }
    if (ix < W && iy < H) mask[(size_t)iy * W + ix] = fabs(sqrt(best) - offset) <= half_stroke ? 0 : 1;
}
#endif

} // namespace f110
