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
// Segments: a lidar polygon's edges are a few pixels long, so an edge inside the image with at most BM_DIRECT items is
// drawn by its own thread straight from the points; the rest (long edges, clipLine cases, every ray) goes through
// 32-byte records whose items the threads share evenly (profiles/r05_bitmap.txt).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace f110 {

enum { BM_FILL = 0, BM_POLYGON = 1, BM_RAYS = 2 };
#ifndef F110_BM_THREADS
#define F110_BM_THREADS 512
#endif
constexpr int BM_THREADS = F110_BM_THREADS;
constexpr int BM_MAX_T = 2048;
constexpr int BM_PER_MAX = (BM_MAX_T + BM_THREADS - 1) / BM_THREADS;
constexpr int BM_XY_SHIFT = 16;
constexpr int BM_TL = 10;
#if defined(F110_BM_TIMELINE)
#define BM_STAMP(i) do { if (tid == 0) s_tl[i] = wall_clock64(); } while (0)
#else
#define BM_STAMP(i) do { } while (0)
#endif

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
    int qcap;                // segment records held in LDS at a time (bm_queue_cap)
    unsigned long long *tl;  // diagnostics (-DF110_BM_TIMELINE): [n][BM_TL] clock stamps of thread 0 at the stage boundaries
};

// One segment / polygon edge, ready to be walked item by item (32 bytes, two ds_read_b128):
//   items [0, L)      Bresenham pixels of the (clipped) segment, L = dmaj + 1 or 0
//   items [L, L + R)  FILL: crossings of the edge with image rows ya .. ya + R - 1, x in 48.16 fixed point
//                     RAYS: the 25 pixels of the 5x5 marker around the segment's end point
struct alignas(16) EdgeRec {
    unsigned xy0;        // x0 | y0 << 16: first pixel of the walk (inside the image)
    unsigned lin;        // dmaj | dmin << 13 | (sy < 0) << 26 | vert << 27 | has_line << 28
    unsigned rows;       // ya | R << 16
    int mark;            // RAYS: marker centre x | y << 16 (as int16)
    long long x, dx;     // crossing at row ya and its per-row increment
};

// Segments drawn straight from the points (no record): both ends inside the image and at most this many items.
#ifndef F110_BM_DIRECT
#define F110_BM_DIRECT 8
#endif
constexpr int BM_DIRECT = F110_BM_DIRECT;
#ifndef F110_BM_X
#define F110_BM_X 0   // experiments only (tools/build_variant.sh): 1 = no segments drawn, 2 = no parity pass, 4 = no points
#endif
#ifndef F110_BM_PRIO
#define F110_BM_PRIO 1
#endif
#ifndef F110_BM_MIN_SHARE
#define F110_BM_MIN_SHARE 1
#endif
constexpr int BM_MIN_SHARE = F110_BM_MIN_SHARE; // items per thread of the record path's walk, at least

// LDS of one image (byte offsets, each on a 16-byte boundary):
//   recs  EdgeRec[qcap]   records of the queued segments of one round (later the 2 KB grey-level table)
//   pts   int2[T]         the polygon's points
//   par   u32[rows*S] + carry bits[rows/32]   crossing parity plane (FILL only: no bytes in the other modes)
//   any   u32[rows*S]     the 1-bit image
//   queue u16[T]          queued segments: inside the image from the front, needing clipLine from the back
//   start int[qcap+1]     exclusive prefix of the round's item counts
//   beams u16[T]          the beam of every point (BitmapArgs::idx, staged once per workgroup)
//   cs    double2[T]      cos / sin of every point's drawing angle (staged once per workgroup)
//   stage u32[2][1 or 2][T64]  the ranges of the next two images as loaded (low words; for an fp64 scan also the high words), T64 = T rounded up to 64
struct BmLayout { unsigned recs, pts, par, any, queue, start, beams, cs, stage, bytes; };
__host__ __device__ inline BmLayout bm_layout(int T, int rows, int S, int qcap, int mode, int f64, bool ahead)
{
    BmLayout l;
    auto up = [](unsigned v) { return (v + 15u) & ~15u; };
    l.recs = 0;
    l.pts = up((unsigned)qcap * (unsigned)sizeof(EdgeRec) < 2048u ? 2048u : (unsigned)qcap * (unsigned)sizeof(EdgeRec));
    l.par = up(l.pts + (unsigned)T * 8u);
    l.any = up(l.par + (mode == BM_FILL ? ((unsigned)rows * S + (unsigned)((rows + 31) / 32)) * 4u : 0u));
    l.queue = up(l.any + (unsigned)rows * S * 4u);
    l.start = up(l.queue + (unsigned)T * 2u);
    l.beams = up(l.start + (unsigned)(qcap + 1) * 4u);
    l.cs = up(l.beams + (ahead ? (unsigned)T * 2u : 0u));
    l.stage = up(l.cs + (ahead ? (unsigned)T * 16u : 0u));
    l.bytes = up(l.stage + (ahead ? (f64 ? 4u : 2u) * (((unsigned)T + 63u) & ~63u) * 4u : 0u));
    return l;
}
// Two shapes of the kernel (bm_fetch_ahead):
//   FILL / POLYGON -- bound by the write and by the latency of a workgroup's stages: ranges fetched two images ahead into
//     LDS (beams / cs / stage above), 64 records per round (the direct pass leaves ~33 of a lidar polygon's 600 edges: long
//     ones and the ones that need clipLine), registers budgeted for 6 waves per SIMD (three workgroups per CU);
//   RAYS -- 50 000 ray and marker pixels per image, bound by VALU / LDS-atomic issue: as many waves as possible, i.e. no
//     staging buffers (<= 40 KB of LDS: four workgroups per CU, 8 waves per SIMD), every ray's record in one round, the
//     ranges loaded where they are used (their latency is a tenth of an image's drawing time).
// More queued segments than records are drawn in several rounds (drawing is OR / XOR into the planes: any order, any grouping).
//   (four-channel images, 256 KB each, are bound by the write alone: measured 4 % better in the second shape)
__host__ __device__ inline bool bm_fetch_ahead(int mode, int channels) { return mode != BM_RAYS && channels != 4; }
__host__ __device__ inline int bm_queue_cap(int T, int mode)
{
    return mode == BM_RAYS || T < 64 ? T : 64;
}
__host__ __device__ inline int bm_waves_per_eu(size_t lds_bytes) { return lds_bytes + 512 <= 40 * 1024 ? 8 : 6; }
__host__ __device__ inline size_t bitmap_lds_bytes(int T, int rows, int S, int mode, int channels, int f64)
{
    return bm_layout(T, rows, S, bm_queue_cap(T, mode), mode, f64, bm_fetch_ahead(mode, channels)).bytes;
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

// Line() + LineIterator::init (connectivity 8, leftToRight) and, for FILL, CollectPolyEdges (LINE_8, shift 0)
// of the segment p0 -> p1.  Both clip the same end points with clipLine, so it runs once.  Returns the item count.
__device__ inline int bm_edge_setup(int mode, int rows, int cols, int2 p0, int2 p1, EdgeRec &r)
{
    long long tx0 = p0.x, ty0 = p0.y, tx1 = p1.x, ty1 = p1.y;
    const bool outside = (unsigned)p0.x >= (unsigned)cols || (unsigned)p1.x >= (unsigned)cols ||
                         (unsigned)p0.y >= (unsigned)rows || (unsigned)p1.y >= (unsigned)rows;
    bool visible = true;
    if (outside) visible = bm_clip_line(cols, rows, tx0, ty0, tx1, ty1);
    int L = 0;
    r.xy0 = 0; r.lin = 0; r.rows = 0; r.mark = 0; r.x = 0; r.dx = 0;
    if (visible) {
        long long x1 = tx0, y1 = ty0;
        int dx = (int)(tx1 - tx0), dy = (int)(ty1 - ty0);
        if (dx < 0) { dx = -dx; dy = -dy; x1 = tx1; y1 = ty1; }      // leftToRight
        unsigned neg = 0;
        if (dy < 0) { dy = -dy; neg = 1; }
        const unsigned vert = dy > dx;
        const unsigned dmaj = vert ? dy : dx, dmin = vert ? dx : dy;
        r.xy0 = (unsigned)x1 | (unsigned)y1 << 16;
        r.lin = dmaj | dmin << 13 | neg << 26 | vert << 27 | 1u << 28;
        L = (int)dmaj + 1;
    }
    int R = 0;
    if (mode == BM_FILL && p0.y != p1.y) {
        const long long half = 1ll << (BM_XY_SHIFT - 1);
        long long c0x = (long long)p0.x << BM_XY_SHIFT, c1x = (long long)p1.x << BM_XY_SHIFT, c0y = p0.y, c1y = p1.y;
        if (outside) {   // "use clipped endpoints to create a more accurate PolyEdge"
            if (ty0 != ty1) { c0y = ty0; c1y = ty1; c0x = tx0 << BM_XY_SHIFT; c1x = tx1 << BM_XY_SHIFT; }
        } else {
            c0x += half; c1x += half;
        }
        // int64 quotient truncated toward zero; |numerator| < 2^49, so the rounded fp64 quotient truncates to the same integer
        const long long dx = (long long)((double)(c1x - c0x) / (double)(c1y - c0y));
        int y0, y1;
        long long x;
        if (p0.y < p1.y) { y0 = p0.y; y1 = p1.y; x = c0x + ((long long)p0.y - c0y) * dx; }
        else { y0 = p1.y; y1 = p0.y; x = c1x + ((long long)p1.y - c1y) * dx; }
        const int ya = y0 > 0 ? y0 : 0, yb = y1 < rows ? y1 : rows;
        if (yb > ya) {
            R = yb - ya;
            r.rows = (unsigned)ya | (unsigned)R << 16;
            r.x = x + (long long)(ya - y0) * dx;
            r.dx = dx;
        }
    } else if (mode == BM_RAYS) {
        // lidar.py:95: cv2.rectangle(p - 2, p + 2, filled); markers wholly outside the image draw nothing
        if (p1.x >= -2 && p1.x < cols + 2 && p1.y >= -2 && p1.y < rows + 2) {
            R = 25;
            r.mark = (int)((unsigned)(p1.x & 0xffff) | (unsigned)p1.y << 16);
        }
    }
    return L + R;
}

// (row and pitch fit 24 bits: one full-rate v_mad_u32_u24 instead of a quarter-rate 32-bit multiply per pixel)
// bm_edge_setup for a segment whose end points both lie inside the image (no clipLine, nothing past 32 bits but the
// record's two 64-bit fields): the same record, bit for bit.
__device__ inline int bm_edge_setup_inside(int mode, int2 p0, int2 p1, EdgeRec &r)
{
    int x1 = p0.x, y1 = p0.y, dx = p1.x - p0.x, dy = p1.y - p0.y;
    if (dx < 0) { dx = -dx; dy = -dy; x1 = p1.x; y1 = p1.y; }      // leftToRight
    unsigned neg = 0;
    if (dy < 0) { dy = -dy; neg = 1; }
    const unsigned vert = dy > dx;
    const unsigned dmaj = vert ? dy : dx, dmin = vert ? dx : dy;
    r.xy0 = (unsigned)x1 | (unsigned)y1 << 16;
    r.lin = dmaj | dmin << 13 | neg << 26 | vert << 27 | 1u << 28;
    r.rows = 0; r.mark = 0; r.x = 0; r.dx = 0;
    const int L = (int)dmaj + 1;
    int R = 0;
    if (mode == BM_FILL && p0.y != p1.y) {
        const int half = 1 << (BM_XY_SHIFT - 1);
        const int c0x = (p0.x << BM_XY_SHIFT) + half, c1x = (p1.x << BM_XY_SHIFT) + half; // x < 4096: fits 32 bits
        const int fdx = (int)((double)(c1x - c0x) / (double)(p1.y - p0.y)); // truncated toward zero, as the 64-bit form
        const int ya = p0.y < p1.y ? p0.y : p1.y;
        R = p0.y < p1.y ? p1.y - p0.y : p0.y - p1.y;
        r.rows = (unsigned)ya | (unsigned)R << 16;
        r.x = p0.y < p1.y ? c0x : c1x;
        r.dx = fdx;
    } else if (mode == BM_RAYS) {
        R = 25; // the marker around an end point inside the image always touches it
        r.mark = (int)((unsigned)(p1.x & 0xffff) | (unsigned)p1.y << 16);
    }
    return L + R;
}

// Workgroup barrier for data exchanged through LDS only.  __syncthreads() also waits for every global load and store of the
// wave (s_waitcnt vmcnt(0)); in this kernel that would drain the previous image's streamed-out pixels and the prefetched
// ranges at every stage boundary.  Global memory carries nothing between the threads here: inputs are read-only, the image
// is write-only.
__device__ inline void bm_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ inline void bm_set(unsigned *plane, int S, int x, int y) { atomicOr(&plane[__mul24(y, S) + (x >> 5)], 1u << (x & 31)); }

// block-wide exclusive scan of one int per thread; returns the exclusive prefix, the sum in `total`
__device__ inline int bm_block_scan(int v, int &total, int *s_wave)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) s_wave[wave] = incl;
    bm_barrier();
    int base = 0, tot = 0;
    for (int w = 0; w < BM_THREADS / 64; w++) {
        const int t = s_wave[w];
        if (w < wave) base += t;
        tot += t;
    }
    total = tot;
    return base + incl - v;
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

typedef unsigned bm_v4u __attribute__((ext_vector_type(4)));
#ifndef F110_BM_NT
#define F110_BM_NT 1
#endif
// One 16-byte store of the output image (written once, never read back by this kernel).  STREAM: the wave's
// lanes write consecutive 16-byte pieces (whole lines per instruction) -> non-temporal; lane-strided pieces
// are left to the L2 to merge (measured: non-temporal partial lines cost 2.6x on the 3-channel path).
template <bool STREAM>
__device__ inline void bm_store16(void *p, unsigned a, unsigned b, unsigned c, unsigned d)
{
    const bm_v4u v = {a, b, c, d};
    if (STREAM && F110_BM_NT) __builtin_nontemporal_store(v, reinterpret_cast<bm_v4u *>(p));
    else *reinterpret_cast<bm_v4u *>(p) = v;
}

// 4 pixel bits -> 4 bytes, each bg (bit 0) or draw (bit 1); cols = bg | draw << 8
__device__ inline unsigned bm_expand4(unsigned nib, unsigned cols2)
{
    const unsigned sel = (nib * 0x00204081u) & 0x01010101u; // bit i -> byte i (0 or 1): a v_perm selector
    return __builtin_amdgcn_perm(0u, cols2, sel);
}

// lidar.py:70-73: point k of image img = rint(center + (scaling_factor * data[k]) * {cos, sin}(angle k)).astype(int),
// data = scan[indices] (:63-64); the integers the reference hands to cv2.fillPoly / polylines / line.
__device__ inline double bm_range(const BitmapArgs &a, int img, int beam)
{
    const long long o = (long long)img * a.stride + beam;
    return a.is_f64 ? static_cast<const double *>(a.scans)[o] : (double)static_cast<const float *>(a.scans)[o];
}
// ---- the ranges of an image, fetched ahead of time straight into LDS (global_load_lds_dword: no register holds them)
// by the UPPER half of the workgroup's threads: issuer thread u = tid - BM_THREADS/2 fetches points u, u + BM_THREADS/2, ...
// Returns the number of loads THIS WAVE issued (wave-uniform): the caller waits with s_waitcnt vmcnt(that number) to know
// that everything issued BEFORE them -- the previous prefetch -- has landed.
// (Written as inline assembly, not __builtin_amdgcn_global_load_lds: the compiler would guard every later LDS read of the
// wave with s_waitcnt vmcnt(0) -- also the reads of the store loop, between the stores.  The waits are bm_wait_vm's.)
__device__ inline void bm_load_to_lds(const unsigned *src, const unsigned *lds_dst_wave /* the wave's lane 0 slot; uniform */)
{
    const unsigned m0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds_dst_wave); // low word of a flat LDS address = LDS offset
    unsigned keep; // (M0 is the compiler's own register: handed back as it was)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(m0) : "memory");
}
__device__ inline int bm_prefetch(const BitmapArgs &a, int img, const unsigned short *beams, unsigned *stage, int tid)
{
    constexpr int HALF = BM_THREADS / 2;
    const int T = a.T, T64 = (T + 63) & ~63;
    const int u = tid - HALF, k0 = u & ~63; // (k0: the wave's first point of a pass, wave-uniform)
    int issued = 0;
    if (u < 0) return 0;
    for (int j = 0; k0 + j * HALF < T; j++) {
        const int k = u + j * HALF, base = __builtin_amdgcn_readfirstlane(k0 + j * HALF);
        if (k < T) {
            const long long o = (long long)img * a.stride + beams[k];
            if (a.is_f64) {
                const unsigned *src = reinterpret_cast<const unsigned *>(static_cast<const double *>(a.scans) + o);
                bm_load_to_lds(src, stage + base);
                bm_load_to_lds(src + 1, stage + T64 + base);
            } else {
                bm_load_to_lds(static_cast<const unsigned *>(a.scans) + o, stage + base);
            }
        }
        issued += a.is_f64 ? 2 : 1;
    }
    return issued;
}
// s_waitcnt vmcnt(n) for a run-time n <= 16 (the counter's immediate)
__device__ inline void bm_wait_vm(int n)
{
    switch (n) {
#define BM_W(i) case i: asm volatile("s_waitcnt vmcnt(" #i ")" ::: "memory"); break;
    BM_W(1) BM_W(2) BM_W(3) BM_W(4) BM_W(5) BM_W(6) BM_W(7) BM_W(8) BM_W(9) BM_W(10) BM_W(11) BM_W(12) BM_W(13) BM_W(14) BM_W(15) BM_W(16)
#undef BM_W
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}
__device__ inline double bm_staged_range(const BitmapArgs &a, const unsigned *stage, int k)
{
    const int T64 = (a.T + 63) & ~63;
    return a.is_f64 ? __hiloint2double((int)stage[T64 + k], (int)stage[k]) : (double)__uint_as_float(stage[k]);
}
__device__ inline int2 bm_point_at(double scale, double r, double c, double s, int cx, int cy)
{
    const double d = scale * r;
    return make_int2((int)(long long)__builtin_rint((double)cx + d * c), (int)(long long)__builtin_rint((double)cy + d * s));
}
__device__ inline int2 bm_point(const BitmapArgs &a, int img, int k, int cx, int cy)
{
    return bm_point_at(a.scale, bm_range(a, img, a.idx[k]), a.cosv[k], a.sinv[k], cx, cy);
}

// function-level view of the point stage (f110_bitmap_points): out [n, T, 2] int32 (x, y)
#if defined(F110_UNIT_CONSUMERS)
static __global__ __launch_bounds__(256) void bitmap_points_kernel(BitmapArgs a, int *out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)a.n * a.T) return;
    const int img = (int)(i / a.T), k = (int)(i % a.T);
    const int2 p = bm_point(a, img, k, a.rows / 2, a.cols / 2);
    out[2 * i] = p.x;
    out[2 * i + 1] = p.y;
}
#endif

#if defined(F110_UNIT_CONSUMERS)
// The image workgroup g draws in round `it` of a launch of `grid` workgroups; a value >= n: none (and none in later rounds).
// (Rotating the rounds' slots, so that a workgroup sees every kind of image, changes nothing: the spread of the workgroups'
// lives -- 620 to 980 us in one launch -- follows their dispatch order, not their images; profiles/r05_bitmap.txt.)
__host__ __device__ inline int bm_image_of(int g, int it, int grid, int n)
{
    const long long i = (long long)it * grid + g;
    return i < n ? (int)i : n;
}

// WPE: waves per SIMD the registers are budgeted for; AHEAD: ranges fetched two images ahead (see bm_fetch_ahead)
template <int WPE, bool AHEAD>
static __global__ __launch_bounds__(BM_THREADS) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void bitmap_kernel(BitmapArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    __shared__ int s_wave[BM_THREADS / 64];
    __shared__ int s_nq[2]; // queued segments: [0] inside the image, [1] needing clipLine
    // the direct pass's crossing increments: floor((|dx| << 16) / |dy|) for |dx|, |dy| < 8 -- the quotient PolyEdge's
    // int64 division truncates toward zero is sign * this; a table look-up instead of an fp64 division per edge
    __shared__ int s_fdx[64];
    static_assert(BM_DIRECT <= 8, "s_fdx holds the increments of edges of at most 8 items");
    const int T = a.T, rows = a.rows, cols = a.cols, S = a.S, tid0 = threadIdx.x, mode = a.mode, qcap = a.qcap;
    const int lane = tid0 & 63;
    const BmLayout lay = bm_layout(T, rows, S, qcap, mode, a.is_f64, AHEAD);
    EdgeRec *recs = reinterpret_cast<EdgeRec *>(s_raw + lay.recs);
    int2 *pts = reinterpret_cast<int2 *>(s_raw + lay.pts);
    unsigned *parp = reinterpret_cast<unsigned *>(s_raw + lay.par);
    unsigned *carry = parp + rows * S;                   // one bit per row
    unsigned *anyp = reinterpret_cast<unsigned *>(s_raw + lay.any);
    unsigned short *queue = reinterpret_cast<unsigned short *>(s_raw + lay.queue);
    int *start = reinterpret_cast<int *>(s_raw + lay.start);
    unsigned short *beams = reinterpret_cast<unsigned short *>(s_raw + lay.beams);
    double2 *cs = reinterpret_cast<double2 *>(s_raw + lay.cs);
    unsigned *stage = reinterpret_cast<unsigned *>(s_raw + lay.stage);
    const int stage_words = (a.is_f64 ? 2 : 1) * ((T + 63) & ~63); // of one of the two buffers

#if defined(F110_BM_TIMELINE)
    __shared__ unsigned long long s_tl[BM_TL];
#endif
    const int cx = rows / 2, cy = cols / 2; // lidar.py:75: center = (dims[0]//2, dims[1]//2), used as (x, y)
    if (tid0 < 64) s_fdx[tid0] = (tid0 & 7) ? ((tid0 >> 3) << BM_XY_SHIFT) / (tid0 & 7) : 0;

    // A workgroup draws one image per round of the launch (bm_image_of; the launch holds as many workgroups as the GPU runs
    // at once).  What an image needs from memory -- its T ranges, an HBM gather -- is fetched TWO images ahead, into LDS:
    // with one image per workgroup 6 of a workgroup's 15 us went into waiting for these loads, and merely ISSUING them
    // stalls a wave for ~1.7 us while every CU's memory pipe is full of the other images' pixels.  So the loads are issued
    // by the upper half of the waves while the lower half runs the parity pass, and consumed an image later
    // (profiles/r05_bitmap.txt).
    // lidar.py:70-81: points = rint(center + (scaling_factor * data) * {cos, sin}(angles)).astype(int)
    if (AHEAD) {
#pragma unroll
        for (int q = 0; q < BM_PER_MAX; q++) {
            const int k = tid0 + q * BM_THREADS;
            if (k < T) {
                const int b = a.idx[k];
                const double2 c = make_double2(a.cosv[k], a.sinv[k]);
                beams[k] = (unsigned short)b;
                cs[k] = c;
                if (!(F110_BM_X & 4)) pts[k] = bm_point_at(a.scale, bm_range(a, blockIdx.x, b), c.x, c.y, cx, cy); // (round 0: image g)
            }
        }
        bm_barrier(); // (beams staged)
        const int img1 = bm_image_of(blockIdx.x, 1, gridDim.x, a.n);
        if (img1 < a.n && !(F110_BM_X & 4)) (void)bm_prefetch(a, img1, beams, stage + stage_words, tid0);
    }
    int it = 0;
    // (!AHEAD: one image per workgroup, the launch has a workgroup per image -- the loop is left after one pass, and the
    // compiler, seeing that, keeps nothing in registers around it.  The grid never exceeds the image count.)
    int img = blockIdx.x;
    do {
    // (the thread index re-enters every image through an opaque copy: per-thread addresses of the stages below are
    // then recomputed per image instead of being carried in registers around the whole loop -- the kernel has 64)
    int tid = tid0;
    asm volatile("" : "+v"(tid));
#if F110_BM_PRIO
    // The CU's arbiter serves the oldest wave first: of the three workgroups that share a CU for the whole launch, the one
    // dispatched first would draw its images a third faster than the last.  The user priority goes round instead.
    if (AHEAD) {
        const int turn = ((int)(blockIdx.x * 3u / gridDim.x) + it) % 3;
        if (turn == 0) __builtin_amdgcn_s_setprio(2); else if (turn == 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
    }
#endif
#if defined(F110_BM_TIMELINE)
    if (tid < BM_TL) s_tl[tid] = 0;
    bm_barrier();
#endif
    BM_STAMP(0);
    const int img_next = AHEAD ? bm_image_of(blockIdx.x, it + 1, gridDim.x, a.n) : a.n, img_next2 = AHEAD ? bm_image_of(blockIdx.x, it + 2, gridDim.x, a.n) : a.n;
    if (!AHEAD)
        for (int k = tid; k < ((F110_BM_X & 4) ? 0 : T); k += BM_THREADS) pts[k] = bm_point(a, img, k, cx, cy);
    // zero both planes (contiguous, each on a 16-byte boundary: 16 bytes per store)
    {
        uint4 *z4 = reinterpret_cast<uint4 *>(parp);
        const int n4 = (int)((lay.queue - lay.par) >> 4);
        for (int i = tid; i < n4; i += BM_THREADS) z4[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    if (tid < 2) s_nq[tid] = 0;
    bm_barrier();
    BM_STAMP(1); // planes zeroed (the points were made before the previous image's stores)
    if (a.draw_center && mode != BM_FILL && tid < 25) {
        // lidar.py:98-100: centre marker in the draw colour (FILL clears it after the fill, below)
        const int x = cx - 2 + tid % 5, y = cy - 2 + tid / 5;
        if ((unsigned)x < (unsigned)cols && (unsigned)y < (unsigned)rows) bm_set(anyp, S, x, y);
    }

    // ---- the segments: polygon edges pts[i-1] -> pts[i] (FILL, POLYGON) or rays centre -> pts[i].
    // Direct pass, one segment per thread: an edge of a lidar polygon joins two neighbouring beams' end points -- a handful
    // of pixels (90 % of them <= 4 items, profiles/r05_bitmap.txt).  An edge with both ends inside the image and at most
    // BM_DIRECT items is drawn here and now, from registers: Bresenham pixels by LineIterator's recurrence, crossings by
    // the 16.16 increment (32 bits suffice: x stays between the two end points).  Everything else -- long edges, edges that
    // need clipLine's 64-bit arithmetic, every ray -- is queued for the record path below, where threads share items, not segments.
    for (int i0 = 0; i0 < ((F110_BM_X & 1) ? 0 : T); i0 += BM_THREADS) {
        const int i = i0 + tid;
        const bool valid = i < T;
        int2 p0 = make_int2(0, 0), p1 = p0;
        if (valid) {
            p1 = pts[i];
            p0 = mode == BM_RAYS ? make_int2(cx, cy) : pts[i == 0 ? T - 1 : i - 1];
        }
        const bool outside = (unsigned)p0.x >= (unsigned)cols || (unsigned)p1.x >= (unsigned)cols ||
                             (unsigned)p0.y >= (unsigned)rows || (unsigned)p1.y >= (unsigned)rows;
        int x1 = p0.x, y1 = p0.y, dx = p1.x - p0.x, dy = p1.y - p0.y;
        if (dx < 0) { dx = -dx; dy = -dy; x1 = p1.x; y1 = p1.y; }      // leftToRight
        const int sy = dy < 0 ? -1 : 1, ady = dy < 0 ? -dy : dy;
        const bool vert = ady > dx;
        const int dmaj = vert ? ady : dx, dmin = vert ? dx : ady;
        const int R = mode == BM_FILL ? ady : 0;
        const bool direct = valid && !outside && mode != BM_RAYS && dmaj + 1 + R <= BM_DIRECT;
        // queue positions by wave-wide counts: one LDS atomic per wave and queue
        const bool q_in = valid && !outside && !direct, q_out = valid && outside;
        const unsigned long long m_in = __ballot(q_in), m_out = __ballot(q_out);
        if (m_in | m_out) {
            const unsigned long long below = (1ull << lane) - 1ull;
            int b_in = 0, b_out = 0;
            if (lane == 0) {
                if (m_in) b_in = atomicAdd(&s_nq[0], __popcll(m_in));
                if (m_out) b_out = atomicAdd(&s_nq[1], __popcll(m_out));
            }
            b_in = __shfl(b_in, 0, 64); b_out = __shfl(b_out, 0, 64);
            if (q_in) queue[b_in + __popcll(m_in & below)] = (unsigned short)i;
            if (q_out) queue[T - 1 - (b_out + __popcll(m_out & below))] = (unsigned short)i;
        }
        if (direct) {
            int m = 0, err = dmaj - 2 * dmin;
            for (int k = 0; k <= dmaj; k++) {
                const int x = vert ? x1 + m : x1 + k, y = vert ? y1 + sy * k : y1 + sy * m;
                bm_set(anyp, S, x, y);
                const int neg = err < 0;
                err += (neg ? 2 * dmaj : 0) - 2 * dmin;
                m += neg;
            }
            if (R > 0) {   // (the crossing items of bm_edge_setup_inside's record: x from the end point with the smaller y)
                const int half = 1 << (BM_XY_SHIFT - 1);
                const int fdx = sy * s_fdx[dx * 8 + ady]; // == (int)((double)((p1.x - p0.x) << 16) / (double)(p1.y - p0.y))
                int xr = ((p0.y < p1.y ? p0.x : p1.x) << BM_XY_SHIFT) + half;
                const int ya = p0.y < p1.y ? p0.y : p1.y;
                for (int j = 0; j < R; j++) {
                    const int X = xr >> BM_XY_SHIFT, y = ya + j;
                    xr += fdx;
                    atomicXor(&parp[__mul24(y, S) + (X >> 5)], 1u << (X & 31));
                    bm_set(anyp, S, X, y);
                }
            }
        }
    }
    bm_barrier();

    BM_STAMP(2); // direct pass done
    // ---- the ranges of the image after the next, into the buffer the previous image's were read from, issued by the
    // upper waves, which have nothing to do while the first waves set the records up; then wait for the loads issued
    // an image ago (everything but the ones just issued)
    if (AHEAD && tid >= BM_THREADS / 2) {
        const int issued = (img_next2 < a.n && !(F110_BM_X & 4)) ? bm_prefetch(a, img_next2, beams, stage + (it & 1) * stage_words, tid) : 0;
        bm_wait_vm(issued);
    }
    // ---- record path: rounds of at most qcap queued segments -- records, prefix of their item counts, item walk
    const int n_in = s_nq[0], nq = n_in + s_nq[1];
    for (int q0 = 0; q0 < nq; q0 += qcap) {
        const int nr = min(qcap, nq - q0);
        // records (the inside segments take the short 32-bit set-up, the ones that need clipLine's 64-bit arithmetic and
        // fp64 divisions the long one) and the exclusive prefix of their item counts
        int total;
        if (nr <= 64) {
            // (the usual case of FILL / POLYGON: a few dozen queued segments -- one wave sets them up and scans the counts
            // in its registers: one barrier, no block-wide scan)
            if (tid < 64) {
                int v = 0;
                if (tid < nr) {
                    const int j = q0 + tid;
                    const bool in = j < n_in;
                    const int i = in ? queue[j] : queue[T - 1 - (j - n_in)];
                    const int2 p1 = pts[i];
                    const int2 p0 = mode == BM_RAYS ? make_int2(cx, cy) : pts[i == 0 ? T - 1 : i - 1];
                    EdgeRec r;
                    if (in) v = bm_edge_setup_inside(mode, p0, p1, r);
                    else v = bm_edge_setup(mode, rows, cols, p0, p1, r);
                    recs[tid] = r;
                }
                int incl = v;
                for (int off = 1; off < 64; off <<= 1) {
                    const int t = __shfl_up(incl, off, 64);
                    if (lane >= off) incl += t;
                }
                if (tid < nr) start[tid] = incl - v;
                if (tid == 63) start[nr] = incl;
            }
            bm_barrier();
            BM_STAMP(3); // records + prefix
            total = start[nr];
        } else {
            // (separate loops, so that a wave runs the long form only if it holds such a segment)
            for (int pass = 0; pass < 2; pass++) {
                const int j0 = pass == 0 ? q0 : max(q0, n_in), j1 = pass == 0 ? min(q0 + nr, n_in) : q0 + nr;
                for (int j = j0 + tid; j < j1; j += BM_THREADS) {
                    const int i = pass == 0 ? queue[j] : queue[T - 1 - (j - n_in)];
                    const int2 p1 = pts[i];
                    const int2 p0 = mode == BM_RAYS ? make_int2(cx, cy) : pts[i == 0 ? T - 1 : i - 1];
                    EdgeRec r;
                    start[j - q0] = pass == 0 ? bm_edge_setup_inside(mode, p0, p1, r) : bm_edge_setup(mode, rows, cols, p0, p1, r);
                    recs[j - q0] = r;
                }
            }
            bm_barrier();
            BM_STAMP(3); // records
            // thread t owns records t*per .. t*per + per - 1
            const int per = (nr + BM_THREADS - 1) / BM_THREADS;
            int cnt[BM_PER_MAX], local = 0;
#pragma unroll
            for (int q = 0; q < BM_PER_MAX; q++) {
                const int i = tid * per + q;
                cnt[q] = (q < per && i < nr) ? start[i] : 0;
                local += cnt[q];
            }
            int base = bm_block_scan(local, total, s_wave);
#pragma unroll
            for (int q = 0; q < BM_PER_MAX; q++) {
                const int i = tid * per + q;
                if (q < per && i < nr) { start[i] = base; base += cnt[q]; }
            }
            if (tid == 0) start[nr] = total;
            bm_barrier();
        }

        BM_STAMP(4); // prefix
        // walk the items: every thread takes a contiguous share of all records' pixels / crossings / marker pixels
        // (at least BM_MIN_SHARE of them: finding its place costs a thread more than a few items do)
        {
            const int ipt = max((total + BM_THREADS - 1) / BM_THREADS, BM_MIN_SHARE);
            int j = tid * ipt;
            const int jend = min(j + ipt, total);
            if (j < jend) {
                int e = bm_find(start, nr, j);
                int k = j - start[e], n_items = start[e + 1] - start[e];
                EdgeRec r = recs[e];
                int L = (r.lin >> 28) ? (int)(r.lin & 0x1fffu) + 1 : 0;
                // Bresenham state after k major steps: minor offset m, error term err (LineIterator's recurrence in closed form)
                int dmaj = (int)(r.lin & 0x1fffu), dmin = (int)((r.lin >> 13) & 0x1fffu);
                int m = (k > 0 && k < L) ? (2 * dmin * k + dmaj - 1) / (2 * dmaj) : 0;
                int err = dmaj - 2 * dmin * (k + 1) + 2 * dmaj * m;
                long long xr = r.x + (long long)(k > L ? k - L : 0) * r.dx;
                for (; j < jend; j++, k++) {
                    while (k >= n_items) {   // next record with at least one item
                        e++;
                        n_items = start[e + 1] - start[e];
                        k = 0;
                        if (n_items > 0) {
                            r = recs[e];
                            L = (r.lin >> 28) ? (int)(r.lin & 0x1fffu) + 1 : 0;
                            dmaj = (int)(r.lin & 0x1fffu); dmin = (int)((r.lin >> 13) & 0x1fffu);
                            m = 0; err = dmaj - 2 * dmin;
                            xr = r.x;
                        }
                    }
                    if (k < L) {
                        const int x0 = (int)(r.xy0 & 0xffffu), y0 = (int)(r.xy0 >> 16);
                        const int sy = (r.lin >> 26) & 1u ? -1 : 1;
                        const bool vert = (r.lin >> 27) & 1u;
                        const int x = vert ? x0 + m : x0 + k, y = vert ? y0 + sy * k : y0 + sy * m;
                        bm_set(anyp, S, x, y);
                        const int neg = err < 0;
                        err += (neg ? 2 * dmaj : 0) - 2 * dmin;
                        m += neg;
                    } else if (mode == BM_FILL) {
                        const int y = (int)(r.rows & 0xffffu) + (k - L);
                        const long long X = xr >> BM_XY_SHIFT;
                        xr += r.dx;
                        if (X < 0) atomicXor(&carry[y >> 5], 1u << (y & 31));
                        else if (X < cols) {
                            atomicXor(&parp[__mul24(y, S) + (int)(X >> 5)], 1u << (X & 31));
                            bm_set(anyp, S, (int)X, y);
                        }
                    } else {
                        const int q = k - L;
                        const int x = (int)(short)(r.mark & 0xffff) - 2 + q % 5, y = (r.mark >> 16) - 2 + q / 5;
                        if ((unsigned)x < (unsigned)cols && (unsigned)y < (unsigned)rows) bm_set(anyp, S, x, y);
                    }
                }
            }
        }
        bm_barrier();
    }

    BM_STAMP(5); // walk
    const int ch = a.channels;
    const size_t img_bytes = (size_t)rows * cols * ch;
    unsigned char *dst = a.out + (size_t)img * img_bytes;
    const unsigned cols2 = ((unsigned)a.bg & 255u) | ((unsigned)a.draw & 255u) << 8;
    const bool use_tab = ch == 1 && (cols & 15) == 0; // (the layout reserves the table's 2 KB)
    uint2 *tab = reinterpret_cast<uint2 *>(recs); // (the records are dead from here on)
    if (use_tab && tid < 256) tab[tid] = make_uint2(bm_expand4((unsigned)tid & 15u, cols2), bm_expand4((unsigned)tid >> 4, cols2));
    if (mode == BM_FILL && !(F110_BM_X & 2)) {
        // ---- inside = crossing on the pixel, or an odd number of crossings strictly left of it; then the centre marker
        // (a row per thread; with rows for half of the threads only, two threads per row: the second takes the words from
        // S0 on, its carry = the row's carry bit ^ the parity of the crossings in the words before)
        const bool halves = 2 * rows <= BM_THREADS;
        const int S0 = halves ? (S + 1) / 2 : S;
        for (int yy = tid; yy < (halves ? 2 * rows : rows); yy += BM_THREADS) {
            const bool second = halves && yy >= rows;
            const int y = second ? yy - rows : yy;
            unsigned c = (carry[y >> 5] >> (y & 31)) & 1u;
            if (second) {
                unsigned x = 0;
                for (int w = 0; w < S0; w++) x ^= parp[y * S + w];
                c ^= (unsigned)__popc(x) & 1u;
            }
            const bool marker_row = a.draw_center && y >= cy - 2 && y <= cy + 2;
            for (int w = second ? S0 : 0; w < (second ? S : S0); w++) {
                unsigned p = parp[y * S + w];
                // (a word column in which no row of this wave has a crossing, a carry or the marker: left as it is)
                if (!__any((p | c) != 0u || marker_row)) continue;
                p ^= p << 1; p ^= p << 2; p ^= p << 4; p ^= p << 8; p ^= p << 16; // inclusive prefix parity
                unsigned v = anyp[y * S + w] | ((p << 1) ^ (0u - c));
                c ^= p >> 31;
                if (marker_row) {   // lidar.py:98-100: 5x5 box in the background colour
                    const int lo = max(cx - 2, 32 * w), hi = min(cx + 2, 32 * w + 31);
                    if (lo <= hi) v &= ~((0xffffffffu >> (31 - (hi - lo))) << (lo - 32 * w));
                }
                anyp[y * S + w] = v;
            }
        }
    }
    bm_barrier();

    BM_STAMP(6); // parity
    // ---- the next image's points (this image's are dead since the last record round; no store of this image is in
    // flight yet, so waiting for the ranges waits for nothing else)
    if (AHEAD && img_next < a.n) {
#pragma unroll
        for (int q = 0; q < BM_PER_MAX; q++) {
            const int k = tid + q * BM_THREADS;
            if (k < T && !(F110_BM_X & 4)) { const double2 c = cs[k]; pts[k] = bm_point_at(a.scale, bm_staged_range(a, stage + ((it + 1) & 1) * stage_words, k), c.x, c.y, cx, cy); }
        }
    }
    BM_STAMP(8); // next points
    // ---- stream the image out: grey levels and channels are expanded here (the only HBM write)
    // One channel: 8 pixel bits -> 8 grey bytes through a 256-entry table (2 KB, in the segment records' place: they are
    // dead after the walk), built by the first 256 threads.  Two ds_read_b64 per 16-byte store replace the twelve VALU
    // instructions of four bm_expand4 -- this kernel is bound by VALU issue (84 % busy), the LDS pipe has room.
    if (use_tab) {
        uint4 *dst4 = reinterpret_cast<uint4 *>(dst);
        const int chunks = rows * cols / 16, cpr = cols / 16;
        const int dq = BM_THREADS / cpr, dr = BM_THREADS - dq * cpr; // (y, h) advanced by carry: see below
        int y = tid / cpr, h = tid - y * cpr, yS = __mul24(y, S);
        const int dqS = __mul24(dq, S);
        if (dr == 0) {
            // (the threads of a workgroup cover whole rows, e.g. 256 columns: a thread stays in its column of chunks)
            const unsigned *src = anyp + yS + (h >> 1);
            const unsigned sh = (h & 1) * 16;
            for (int c = tid; c < chunks; c += BM_THREADS, src += dqS) {
                const unsigned bits = *src >> sh;
                const uint2 lo = tab[bits & 0xffu], hi = tab[(bits >> 8) & 0xffu];
                bm_store16<true>(dst4 + c, lo.x, lo.y, hi.x, hi.y);
            }
        } else
        for (int c = tid; c < chunks; c += BM_THREADS) {
            const unsigned bits = anyp[yS + (h >> 1)] >> ((h & 1) * 16);
            const uint2 lo = tab[bits & 0xffu], hi = tab[(bits >> 8) & 0xffu];
            bm_store16<true>(dst4 + c, lo.x, lo.y, hi.x, hi.y);
            h += dr; y += dq; yS += dqS;
            if (h >= cpr) { h -= cpr; y++; yS += S; }
        }
    } else if ((cols & 15) == 0) {
        uint4 *dst4 = reinterpret_cast<uint4 *>(dst);
        if (ch == 4) {
            // 4 pixels -> 16 bytes "v v v 255" each (lidar.py:150-152: opaque alpha), lanes contiguous
            const int quads = rows * cols / 4, qpr = cols / 4;
            // (y, h) of chunk c = tid + i * BM_THREADS advanced by carry, not divided out per chunk (a 32-bit division
            // is ~25 instructions, three of them quarter-rate multiplies: a tenth of this kernel)
            const int dq = BM_THREADS / qpr, dr = BM_THREADS - dq * qpr;
            int y = tid / qpr, h = tid - y * qpr, yS = __mul24(y, S);
            const int dqS = __mul24(dq, S);
            for (int c = tid; c < quads; c += BM_THREADS) {
                const unsigned e = bm_expand4((anyp[yS + (h >> 3)] >> ((h & 7) * 4)) & 15u, cols2);
                bm_store16<true>(dst4 + c, __builtin_amdgcn_perm(0u, e, 0x0d000000u), __builtin_amdgcn_perm(0u, e, 0x0d010101u),
                                 __builtin_amdgcn_perm(0u, e, 0x0d020202u), __builtin_amdgcn_perm(0u, e, 0x0d030303u));
                h += dr; y += dq; yS += dqS;
                if (h >= qpr) { h -= qpr; y++; yS += S; }
            }
        } else {
            // 16 pixels (half a plane word) per step: 16 or 48 output bytes as 16-byte stores
            const int chunks = rows * cols / 16, cpr = cols / 16;
            const int dq = BM_THREADS / cpr, dr = BM_THREADS - dq * cpr; // (see above)
            int y = tid / cpr, h = tid - y * cpr, yS = __mul24(y, S);
            const int dqS = __mul24(dq, S);
            for (int c = tid; c < chunks; c += BM_THREADS) {
                const unsigned bits = (anyp[yS + (h >> 1)] >> ((h & 1) * 16)) & 0xffffu;
                unsigned e4[4];
#pragma unroll
                for (int q = 0; q < 4; q++) e4[q] = bm_expand4((bits >> (4 * q)) & 15u, cols2);
                if (ch == 1) {
                    bm_store16<true>(dst4 + c, e4[0], e4[1], e4[2], e4[3]);
                } else {
                    unsigned w[12];
#pragma unroll
                    for (int q = 0; q < 4; q++) {   // v0 v0 v0 v1 | v1 v1 v2 v2 | v2 v3 v3 v3
                        w[3 * q + 0] = __builtin_amdgcn_perm(0u, e4[q], 0x01000000u);
                        w[3 * q + 1] = __builtin_amdgcn_perm(0u, e4[q], 0x02020101u);
                        w[3 * q + 2] = __builtin_amdgcn_perm(0u, e4[q], 0x03030302u);
                    }
#pragma unroll
                    for (int q = 0; q < 3; q++) bm_store16<false>(dst4 + 3 * c + q, w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
                }
                h += dr; y += dq; yS += dqS;
                if (h >= cpr) { h -= cpr; y++; yS += S; }
            }
        }
    } else {
        for (size_t o = tid; o < img_bytes; o += BM_THREADS) {
            const int p = (int)(o / ch), c = (int)(o - (size_t)p * ch), y = p / cols, x = p - y * cols;
            const unsigned bit = (anyp[y * S + (x >> 5)] >> (x & 31)) & 1u;
            dst[o] = c == 3 ? (unsigned char)255 : (unsigned char)(bit ? a.draw : a.bg);
        }
    }
#if defined(F110_BM_TIMELINE)
    bm_barrier();
    BM_STAMP(7); // stores issued
    if (tid == 0 && a.tl) for (int i = 0; i < BM_TL; i++) a.tl[(size_t)img * BM_TL + i] = s_tl[i];
#endif
    if (!AHEAD) break;
    bm_barrier(); // the planes, the records' place (grey-level table) and the points are the next image's from here
    img = img_next; it++;
    } while (img < a.n);
}
#endif

// ---- f1tenth_gym/examples/lidar.py:212-244: point-occupancy grid (one workgroup per scan)
struct OccArgs {
    const void *scans; int is_f64; long long stride; int n, num_beams;
    const double *cosv, *sinv; // [num_beams], angles np.linspace(-135, 135, n) * pi / 180
    double max_range, lo, hi;
    int grid;
    unsigned char *out;        // [n, grid, grid]
};

#if defined(F110_UNIT_CONSUMERS)
static __global__ __launch_bounds__(BM_THREADS) void occupancy_kernel(OccArgs a)
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
    if ((G * G) % 16 == 0) {
        for (int c = tid; c < G * G / 16; c += BM_THREADS) {
            const unsigned h = (bits[c >> 1] >> ((c & 1) * 16)) & 0xffffu;
            bm_store16<true>(reinterpret_cast<uint4 *>(dst) + c, bm_expand4(h & 15u, 0x0100u), bm_expand4((h >> 4) & 15u, 0x0100u),
                       bm_expand4((h >> 8) & 15u, 0x0100u), bm_expand4(h >> 12, 0x0100u));
        }
    } else {
        for (int p = tid; p < G * G; p += BM_THREADS) dst[p] = (unsigned char)((bits[p >> 5] >> (p & 31)) & 1u);
    }
}
#endif

} // namespace f110
