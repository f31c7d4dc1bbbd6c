// f110_bitmap.h -- scan -> bird's-eye bitmap rasteriser, the first consumer of
// F110Env.step's scans in both RL callers (reference weap_util/weap_util/lidar.py:4-154,
// src/SAL.py:76,119, examples/lidar_example.py:104-105).  The reference draws with OpenCV 4.11
// (fillPoly / polylines / line / rectangle, 8-bit, LINE_8); the kernel reproduces those results with
// order-independent formulations so that one workgroup can draw one image in parallel:
//   * every Bresenham pixel of a segment in closed form (minor offset after k major steps
//     = floor((2*dmin*k + dmaj - 1) / (2*dmaj)), which is what LineIterator's error recurrence yields);
//   * polygon fill as bit-plane parity counting: a pixel of row y is inside iff a crossing falls on it,
//     or an odd number of crossings lie strictly left of it (equivalent to pairing the sorted crossings
//     of FillEdgeCollection's active-edge list), crossings x(y) = x0 + (y - y0)*dx in 48.16 fixed point;
//   * all drawing is one colour, so outline, fill and markers are OR-ed into a 1-bit image in LDS and the
//     grey levels / channels are expanded only when the image is streamed out (the only HBM write).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace f110 {

enum { BM_FILL = 0, BM_POLYGON = 1, BM_RAYS = 2 };
constexpr int BM_THREADS = 256;
constexpr int BM_XY_SHIFT = 16;

struct BitmapArgs {
    const void *scans;       // [n, stride] f32 or f64
    int is_f64;
    long long stride;        // elements between consecutive scans
    int n;
    const int *idx;          // [T] beam subset (np.linspace(0, num_beams - 1, T, dtype=int))
    const double *cosv, *sinv; // [T] cos / sin of the drawing angles
    int T;
    int rows, cols, channels, mode, bg, draw, draw_center;
    double scale;
    unsigned char *out;      // [n, rows, cols(, channels)]
    int S;                   // words per bit-plane row (cols/32 rounded up, made odd)
};

struct LineRec { int x0, y0, dmaj, dmin; int sy, vert; };
struct EdgeRec { long long x, dx; int y0, ya; };

__host__ __device__ inline size_t bitmap_lds_bytes(int T, int rows, int S)
{
    // pts int2[T] | start int[T+1] | LineRec[T] | EdgeRec[T] | any[rows*S] | par[rows*S] | carry[rows]
    size_t b = (size_t)T * 8 + (size_t)(T + 1) * 4;
    b = (b + 7) & ~(size_t)7;
    b += (size_t)T * sizeof(LineRec);
    b = (b + 7) & ~(size_t)7;
    b += (size_t)T * sizeof(EdgeRec);
    b += (size_t)rows * S * 4 * 2 + (size_t)rows * 4;
    return b;
}

// cv::clipLine(Size2l, Point2l&, Point2l&), drawing.cpp
__device__ inline bool bm_clip_line(long long width, long long height, long long &x1, long long &y1, long long &x2, long long &y2)
{
    const long long right = width - 1, bottom = height - 1;
    int c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8;
    int c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8;
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
        long long a;
        if (c1 & 12) {
            a = c1 < 8 ? 0 : bottom;
            x1 += (long long)((double)(a - y1) * (double)(x2 - x1) / (double)(y2 - y1));
            y1 = a;
            c1 = (x1 < 0) + (x1 > right) * 2;
        }
        if (c2 & 12) {
            a = c2 < 8 ? 0 : bottom;
            x2 += (long long)((double)(a - y2) * (double)(x2 - x1) / (double)(y2 - y1));
            y2 = a;
            c2 = (x2 < 0) + (x2 > right) * 2;
        }
        if ((c1 & c2) == 0 && (c1 | c2) != 0) {
            if (c1) {
                a = c1 == 1 ? 0 : right;
                y1 += (long long)((double)(a - x1) * (double)(y2 - y1) / (double)(x2 - x1));
                x1 = a;
                c1 = 0;
            }
            if (c2) {
                a = c2 == 1 ? 0 : right;
                y2 += (long long)((double)(a - x2) * (double)(y2 - y1) / (double)(x2 - x1));
                x2 = a;
                c2 = 0;
            }
        }
    }
    return (c1 | c2) == 0;
}

// Line() + LineIterator::init (connectivity 8, leftToRight): the record of one segment; returns its pixel count
__device__ inline int bm_line_setup(int rows, int cols, int ax, int ay, int bx, int by, LineRec &r)
{
    long long x1 = ax, y1 = ay, x2 = bx, y2 = by;
    if ((unsigned long long)x1 >= (unsigned long long)cols || (unsigned long long)x2 >= (unsigned long long)cols ||
        (unsigned long long)y1 >= (unsigned long long)rows || (unsigned long long)y2 >= (unsigned long long)rows) {
        if (!bm_clip_line(cols, rows, x1, y1, x2, y2)) { r.dmaj = -1; return 0; }
    }
    int dx = (int)(x2 - x1), dy = (int)(y2 - y1);
    if (dx < 0) { dx = -dx; dy = -dy; x1 = x2; y1 = y2; }
    int sy = 1;
    if (dy < 0) { dy = -dy; sy = -1; }
    const int vert = dy > dx;
    r.x0 = (int)x1; r.y0 = (int)y1; r.sy = sy; r.vert = vert;
    r.dmaj = vert ? dy : dx;
    r.dmin = vert ? dx : dy;
    return r.dmaj + 1;
}

__device__ inline void bm_set(unsigned *plane, int S, int x, int y) { atomicOr(&plane[y * S + (x >> 5)], 1u << (x & 31)); }

__device__ inline void bm_line_pixel(const LineRec &r, int k, int &x, int &y)
{
    const int m = r.dmaj > 0 ? (2 * r.dmin * k + r.dmaj - 1) / (2 * r.dmaj) : 0;
    if (r.vert) { x = r.x0 + m; y = r.y0 + r.sy * k; }
    else { x = r.x0 + k; y = r.y0 + r.sy * m; }
}

// block-wide exclusive scan of one int per thread (BM_THREADS threads); returns the exclusive prefix, total in `total`
__device__ inline int bm_block_scan(int v, int &total, int *s_wave /*[4]*/)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    __syncthreads();
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < BM_THREADS / 64; w++) {
        const int t = s_wave[w];
        if (w < wave) base += t;
        tot += t;
    }
    total = tot;
    return base + incl - v;
}

// CollectPolyEdges (LINE_8, shift 0): the fill record of the edge pt0 -> pt1; returns the number of image rows it crosses
__device__ inline int bm_edge_setup(int rows, int cols, int p0x, int p0y, int p1x, int p1y, EdgeRec &e)
{
    e.ya = 0; e.y0 = 0; e.x = 0; e.dx = 0;
    if (p0y == p1y) return 0;
    const long long half = 1ll << (BM_XY_SHIFT - 1);
    long long c0x = (long long)p0x << BM_XY_SHIFT, c1x = (long long)p1x << BM_XY_SHIFT, c0y = p0y, c1y = p1y;
    if ((unsigned)p0x >= (unsigned)cols || (unsigned)p1x >= (unsigned)cols || (unsigned)p0y >= (unsigned)rows ||
        (unsigned)p1y >= (unsigned)rows) {
        long long tx0 = p0x, ty0 = p0y, tx1 = p1x, ty1 = p1y;
        bm_clip_line(cols, rows, tx0, ty0, tx1, ty1);
        if (ty0 != ty1) { c0y = ty0; c1y = ty1; c0x = tx0 << BM_XY_SHIFT; c1x = tx1 << BM_XY_SHIFT; }
    } else {
        c0x += half; c1x += half;
    }
    const long long dx = (c1x - c0x) / (c1y - c0y);
    int y0, y1;
    long long x;
    if (p0y < p1y) { y0 = p0y; y1 = p1y; x = c0x + ((long long)p0y - c0y) * dx; }
    else { y0 = p1y; y1 = p0y; x = c1x + ((long long)p1y - c1y) * dx; }
    e.x = x; e.dx = dx; e.y0 = y0;
    const int ya = y0 > 0 ? y0 : 0, yb = y1 < rows ? y1 : rows;
    e.ya = ya;
    return yb > ya ? yb - ya : 0;
}

// first index e in [0, n) with start[e + 1] > j  (start is an exclusive prefix with start[n] = total)
__device__ inline int bm_find(const int *start, int n, int j)
{
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (start[mid + 1] > j) hi = mid; else lo = mid + 1;
    }
    return lo;
}

__device__ inline unsigned bm_bit(const unsigned *plane, int S, int x, int y) { return (plane[y * S + (x >> 5)] >> (x & 31)) & 1u; }

__global__ __launch_bounds__(BM_THREADS) void bitmap_kernel(BitmapArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    __shared__ int s_wave[BM_THREADS / 64];
    const int T = a.T, rows = a.rows, cols = a.cols, S = a.S, tid = threadIdx.x;
    int2 *pts = reinterpret_cast<int2 *>(s_raw);
    int *start = reinterpret_cast<int *>(s_raw + (size_t)T * 8);
    size_t off = ((size_t)T * 8 + (size_t)(T + 1) * 4 + 7) & ~(size_t)7;
    LineRec *lines = reinterpret_cast<LineRec *>(s_raw + off);
    off = (off + (size_t)T * sizeof(LineRec) + 7) & ~(size_t)7;
    EdgeRec *edges = reinterpret_cast<EdgeRec *>(s_raw + off);
    off += (size_t)T * sizeof(EdgeRec);
    unsigned *anyp = reinterpret_cast<unsigned *>(s_raw + off);
    unsigned *parp = anyp + rows * S;
    unsigned *carry = parp + rows * S;

    const int img = blockIdx.x;
    if (img >= a.n) return;
    const int cx = rows / 2, cy = cols / 2; // lidar.py:75: center = (dims[0]//2, dims[1]//2), used as (x, y)

    for (int i = tid; i < rows * S * 2 + rows; i += BM_THREADS) anyp[i] = 0u;
    // lidar.py:70-81: points = rint(center + (scaling_factor * data) * {cos, sin}(angles)).astype(int)
    for (int k = tid; k < T; k += BM_THREADS) {
        const long long o = (long long)img * a.stride + a.idx[k];
        const double r = a.is_f64 ? static_cast<const double *>(a.scans)[o] : (double)static_cast<const float *>(a.scans)[o];
        const double d = a.scale * r;
        pts[k] = make_int2((int)(long long)__builtin_rint((double)cx + d * a.cosv[k]),
                           (int)(long long)__builtin_rint((double)cy + d * a.sinv[k]));
    }
    __syncthreads();

    // ---- segments: polygon outline (FILL, POLYGON) or centre -> point rays (RAYS)
    const int per = (T + BM_THREADS - 1) / BM_THREADS;
    {
        int cnt[8], local = 0;
        for (int q = 0; q < per; q++) {
            const int i = tid * per + q;
            cnt[q & 7] = 0;
            if (i < T) {
                const int2 p1 = pts[i];
                const int2 p0 = a.mode == BM_RAYS ? make_int2(cx, cy) : pts[i == 0 ? T - 1 : i - 1];
                LineRec r;
                cnt[q & 7] = bm_line_setup(rows, cols, p0.x, p0.y, p1.x, p1.y, r);
                lines[i] = r;
            }
            local += cnt[q & 7];
        }
        int total;
        int base = bm_block_scan(local, total, s_wave);
        for (int q = 0; q < per; q++) {
            const int i = tid * per + q;
            if (i < T) { start[i] = base; base += cnt[q & 7]; }
        }
        if (tid == 0) start[T] = total;
        __syncthreads();
        const int ipt = (total + BM_THREADS - 1) / BM_THREADS;
        int j = tid * ipt;
        const int jend = min(j + ipt, total);
        if (j < jend) {
            int e = bm_find(start, T, j);
            for (; j < jend; j++) {
                while (start[e + 1] <= j) e++;
                int x, y;
                bm_line_pixel(lines[e], j - start[e], x, y);
                bm_set(anyp, S, x, y);
            }
        }
        __syncthreads();
    }

    if (a.mode == BM_FILL) {
        // ---- crossings of every non-horizontal edge with every image row it spans
        int cnt[8], local = 0;
        for (int q = 0; q < per; q++) {
            const int i = tid * per + q;
            cnt[q & 7] = 0;
            if (i < T) {
                const int2 p1 = pts[i], p0 = pts[i == 0 ? T - 1 : i - 1];
                EdgeRec e;
                cnt[q & 7] = bm_edge_setup(rows, cols, p0.x, p0.y, p1.x, p1.y, e);
                edges[i] = e;
            }
            local += cnt[q & 7];
        }
        int total;
        int base = bm_block_scan(local, total, s_wave);
        for (int q = 0; q < per; q++) {
            const int i = tid * per + q;
            if (i < T) { start[i] = base; base += cnt[q & 7]; }
        }
        if (tid == 0) start[T] = total;
        __syncthreads();
        const int ipt = (total + BM_THREADS - 1) / BM_THREADS;
        int j = tid * ipt;
        const int jend = min(j + ipt, total);
        if (j < jend) {
            int e = bm_find(start, T, j);
            for (; j < jend; j++) {
                while (start[e + 1] <= j) e++;
                const EdgeRec ed = edges[e];
                const int y = ed.ya + (j - start[e]);
                const long long X = (ed.x + (long long)(y - ed.y0) * ed.dx) >> BM_XY_SHIFT;
                if (X < 0) atomicXor(&carry[y], 1u);
                else if (X < cols) {
                    atomicXor(&parp[y * S + (int)(X >> 5)], 1u << (X & 31));
                    bm_set(anyp, S, (int)X, y);
                }
            }
        }
        __syncthreads();
        // ---- inside = crossing on the pixel, or odd number of crossings strictly left of it
        for (int y = tid; y < rows; y += BM_THREADS) {
            unsigned c = carry[y] & 1u;
            for (int w = 0; w < S; w++) {
                unsigned p = parp[y * S + w];
                p ^= p << 1; p ^= p << 2; p ^= p << 4; p ^= p << 8; p ^= p << 16; // inclusive prefix parity
                const unsigned inside = (p << 1) ^ (0u - c);
                c ^= p >> 31;
                anyp[y * S + w] |= inside;
            }
        }
        __syncthreads();
    }

    if (a.mode == BM_RAYS) {
        // lidar.py:95: cv2.rectangle(p - 2, p + 2, filled) around every point
        for (int i = tid; i < T * 25; i += BM_THREADS) {
            const int2 p = pts[i / 25];
            const int x = p.x - 2 + (i % 25) % 5, y = p.y - 2 + (i % 25) / 5;
            if ((unsigned)x < (unsigned)cols && (unsigned)y < (unsigned)rows) bm_set(anyp, S, x, y);
        }
        __syncthreads();
    }
    if (a.draw_center && tid < 25) {
        // lidar.py:98-100: the centre marker, background colour in FILL mode
        const int x = cx - 2 + tid % 5, y = cy - 2 + tid / 5;
        if ((unsigned)x < (unsigned)cols && (unsigned)y < (unsigned)rows) {
            if (a.mode == BM_FILL) atomicAnd(&anyp[y * S + (x >> 5)], ~(1u << (x & 31)));
            else bm_set(anyp, S, x, y);
        }
    }
    __syncthreads();

    // ---- stream the image out: grey levels and channels are expanded here
    const int ch = a.channels;
    const size_t img_bytes = (size_t)rows * cols * ch;
    unsigned char *dst = a.out + (size_t)img * img_bytes;
    const unsigned bg = (unsigned)a.bg & 255u, flip = ((unsigned)a.bg ^ (unsigned)a.draw) & 255u;
    if (ch == 1 && (cols & 31) == 0) {
        // 16 pixels (half a plane word) -> one 16-byte store
        const int chunks = rows * cols / 16, cpr = cols / 16;
        for (int c = tid; c < chunks; c += BM_THREADS) {
            const int y = c / cpr, h = c - y * cpr;
            const unsigned bits = (anyp[y * S + (h >> 1)] >> ((h & 1) * 16)) & 0xffffu;
            uint4 v;
            unsigned *vw = reinterpret_cast<unsigned *>(&v);
            for (int q = 0; q < 4; q++) {
                const unsigned nib = (bits >> (4 * q)) & 15u;
                const unsigned ones = ((nib * 0x00204081u) & 0x01010101u) * 255u; // bit i -> byte i = 0xff
                vw[q] = (bg * 0x01010101u) ^ (ones & (flip * 0x01010101u));
            }
            reinterpret_cast<uint4 *>(dst)[c] = v;
        }
    } else if (ch == 4) {
        for (int p = tid; p < rows * cols; p += BM_THREADS) {
            const int y = p / cols, x = p - y * cols;
            const unsigned v = bg ^ (bm_bit(anyp, S, x, y) ? flip : 0u);
            reinterpret_cast<unsigned *>(dst)[p] = v * 0x00010101u | 0xff000000u; // lidar.py:150-152: opaque alpha
        }
    } else {
        for (size_t o = tid; o < img_bytes; o += BM_THREADS) {
            const int p = (int)(o / ch), y = p / cols, x = p - y * cols;
            dst[o] = (unsigned char)(bg ^ (bm_bit(anyp, S, x, y) ? flip : 0u)); // ch is 1 or 3 here
        }
    }
}

// ---- f1tenth_gym/examples/lidar.py:212-244: point-occupancy grid (one workgroup per scan)
struct OccArgs {
    const void *scans; int is_f64; long long stride; int n, num_beams;
    const double *cosv, *sinv; // [num_beams], angles np.linspace(-135, 135, n) * pi / 180
    double max_range, lo, hi;
    int grid;
    unsigned char *out;        // [n, grid, grid]
};

__global__ __launch_bounds__(BM_THREADS) void occupancy_kernel(OccArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    unsigned *bits = reinterpret_cast<unsigned *>(s_raw); // grid*grid bits
    const int img = blockIdx.x, tid = threadIdx.x, G = a.grid;
    if (img >= a.n) return;
    const int words = (G * G + 31) / 32;
    for (int i = tid; i < words; i += BM_THREADS) bits[i] = 0u;
    __syncthreads();
    for (int b = tid; b < a.num_beams; b += BM_THREADS) {
        const long long o = (long long)img * a.stride + b;
        const double r = a.is_f64 ? static_cast<const double *>(a.scans)[o] : (double)static_cast<const float *>(a.scans)[o];
        if (r >= a.max_range) continue;
        const double x = r * a.cosv[b], y = r * a.sinv[b];
        if (!(a.lo <= x && x <= a.hi && a.lo <= y && y <= a.hi)) continue;
        int i_row = (int)(((x - a.lo) / (a.hi - a.lo)) * (double)(G - 1));
        int i_col = (int)(((y - a.lo) / (a.hi - a.lo)) * (double)(G - 1));
        i_row = min(max(i_row, 0), G - 1);
        i_col = min(max(i_col, 0), G - 1);
        const int p = i_row * G + i_col;
        atomicOr(&bits[p >> 5], 1u << (p & 31));
    }
    __syncthreads();
    unsigned char *dst = a.out + (size_t)img * G * G;
    for (int p = tid; p < G * G; p += BM_THREADS) dst[p] = (unsigned char)((bits[p >> 5] >> (p & 31)) & 1u);
}

} // namespace f110
