// f110_planner.h -- batched pure-pursuit planner: the caller on the other
// side of F110Env.step, reference examples/waypoint_follow.py:15-217, one wavefront per car.
// fp64, plain mul/add in the reference's order (its np.dot calls are BLAS-rounded, see DESIGN.md
// section 2); checked against oracle/planner.py in the tests.
#pragma once
#include "f110_device.h"

#pragma clang fp contract(off)

namespace f110 {

struct PlanArgs {
    const double *waypoints; // [M,3] x, y, speed
    int M;
    double lookahead, vgain, wheelbase, max_reacquire;
    const double *state;     // [n,7]
    int n;
    double *actions;         // [n,2] steer, speed
};

// waypoint_follow.py:49-129, one segment: returns true and (t) if the circle is hit
__device__ inline bool seg_circle(double sx, double sy, double ex, double ey, double px, double py, double radius,
                                  double &t1, double &t2)
{
    const double Vx = (ex + 1e-6) - sx, Vy = (ey + 1e-6) - sy; // end = trajectory[i+1,:] + 1e-6
    const double a = Vx * Vx + Vy * Vy;
    const double b = 2.0 * (Vx * (sx - px) + Vy * (sy - py));
    const double c = (sx * sx + sy * sy) + (px * px + py * py) - 2.0 * (sx * px + sy * py) - radius * radius;
    double disc = b * b - 4 * a * c;
    if (disc < 0) return false;
    disc = sqrt(disc);
    t1 = (-b - disc) / (2.0 * a);
    t2 = (-b + disc) / (2.0 * a);
    return true;
}

// One WAVEFRONT per car (round 2; round 1 ran one lane per car, every lane walking all M - 1 segments: 128 us for
// 65 536 cars on the 783-point example raceline).  The 64 lanes evaluate 64 consecutive segments at once, and whole
// 64-segment blocks are skipped when their bounding box is farther from the car than a distance already found --
// which is only possible because a wave serves ONE car (lanes with different cars would need different blocks).
// The results are the reference's: first arg-min of the sqrt'd distances (ties go to the lower index: blocks are
// visited in index order and replace on strict `<`; a skipped block lies strictly farther than the minimum), and the
// first segment in the reference's search order whose circle intersection qualifies (a ballot's lowest set bit).
constexpr int PP_WAVES = 16; // cars per workgroup (they share the LDS copy of the raceline)

// wave-wide min / max of a double with DPP moves (no LDS round trips): xor-1, xor-2 inside quads, mirror inside half rows
// and rows leave every lane with its 16-lane row's result; the four rows are combined through readlane.
template <int CTRL>
__device__ inline double dpp_mov_f64(double v)
{
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)u, (int)(unsigned)u, CTRL, 0xf, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)(u >> 32), (int)(unsigned)(u >> 32), CTRL, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

__device__ inline double readlane_f64(double v, int l)
{
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), l);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

template <bool MAX>
__device__ inline double wave_reduce_f64(double v)
{
    auto op = [](double a, double b) { return MAX ? fmax(a, b) : fmin(a, b); };
    v = op(v, dpp_mov_f64<0xB1>(v));  // quad_perm [1,0,3,2]
    v = op(v, dpp_mov_f64<0x4E>(v));  // quad_perm [2,3,0,1]
    v = op(v, dpp_mov_f64<0x141>(v)); // row_half_mirror
    v = op(v, dpp_mov_f64<0x140>(v)); // row_mirror
    return op(op(readlane_f64(v, 0), readlane_f64(v, 16)), op(readlane_f64(v, 32), readlane_f64(v, 48)));
}

__device__ inline double wave_min_f64(double v) { return wave_reduce_f64<false>(v); }
__device__ inline double wave_max_f64(double v) { return wave_reduce_f64<true>(v); }

__host__ __device__ inline size_t pure_pursuit_lds_bytes(int M)
{
    const int nblk = (M - 1 + 63) / 64;
    return ((size_t)M * 3 + (size_t)nblk * 4) * sizeof(double);
}

#if defined(F110_UNIT_CONSUMERS)
static __global__ __launch_bounds__(PP_WAVES * 64) void pure_pursuit_kernel(PlanArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double s_wp[]; // [M,3] waypoints, then [nblk,4] block boxes
    const int M = a.M, nseg = M - 1, nblk = (nseg + 63) >> 6;
    double *s_box = s_wp + (size_t)M * 3; // xmin, xmax, ymin, ymax of the points of segments 64b .. 64b+63
    for (int i = threadIdx.x; i < M * 3; i += blockDim.x) s_wp[i] = a.waypoints[i];
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    __shared__ int s_degenerate; // the raceline has a zero-length segment (see below)
    if (threadIdx.x == 0) s_degenerate = 0;
    __syncthreads();
    for (int b = wave; b < nblk; b += PP_WAVES) {
        const int i = 64 * b + lane;
        const bool ok = i < nseg;
        const double x0 = ok ? s_wp[3 * i] : 0, y0 = ok ? s_wp[3 * i + 1] : 0, x1 = ok ? s_wp[3 * i + 3] : 0, y1 = ok ? s_wp[3 * i + 4] : 0;
        const double inf = __builtin_inf();
        const double xl = wave_min_f64(ok ? fmin(x0, x1) : inf), xh = wave_max_f64(ok ? fmax(x0, x1) : -inf);
        const double yl = wave_min_f64(ok ? fmin(y0, y1) : inf), yh = wave_max_f64(ok ? fmax(y0, y1) : -inf);
        const bool deg = ok && ((x1 - x0) * (x1 - x0) + (y1 - y0) * (y1 - y0) == 0.0);
        if (lane == 0) { s_box[4 * b] = xl; s_box[4 * b + 1] = xh; s_box[4 * b + 2] = yl; s_box[4 * b + 3] = yh; }
        if (__builtin_amdgcn_ballot_w64(deg) != 0ull && lane == 0) s_degenerate = 1;
    }
    __syncthreads();
    __shared__ double s_res[PP_WAVES][4]; // per car of the block: target x, y, speed, 1.0 if there is a target
    const int car = blockIdx.x * PP_WAVES + wave;
    const int car_c = min(car, a.n - 1);    // (waves past the last car redo the last one: they must reach the barrier below)
    const double px = a.state[(size_t)car_c * 7], py = a.state[(size_t)car_c * 7 + 1];

    // distance of this lane's segment of block b (inf past the last segment) and its clipped parameter
    auto seg_dist = [&](int b, double &t) {
        const int i = 64 * b + lane;
        t = 0.0;
        if (i >= nseg) return __builtin_inf();
        const double x0 = s_wp[3 * i], y0 = s_wp[3 * i + 1];
        const double dx = s_wp[3 * i + 3] - x0, dy = s_wp[3 * i + 4] - y0;
        const double l2 = dx * dx + dy * dy;
        t = ((px - x0) * dx + (py - y0) * dy) / l2;
        t = t < 0.0 ? 0.0 : t;
        t = t > 1.0 ? 1.0 : t;
        const double qx = px - (x0 + t * dx), qy = py - (y0 + t * dy);
        return sqrt(qx * qx + qy * qy);
    };
    // lower bound of the distance from the car to anything in block b: its distance to the block's box
    auto box_dist = [&](int b) {
        const double ex = fmax(fmax(s_box[4 * b] - px, px - s_box[4 * b + 1]), 0.0);
        const double ey = fmax(fmax(s_box[4 * b + 2] - py, py - s_box[4 * b + 3]), 0.0);
        return sqrt(ex * ex + ey * ey);
    };

    // nearest_point_on_trajectory (:16-47).  Lane j holds the lower bound of block j (+64: a second word for long
    // racelines); an upper bound comes from the nearest segment of the block whose box is closest; then only the blocks
    // whose box is within that bound are evaluated, in index order.
    double lb[2];
    double lb_best = __builtin_inf();
    int b_star = 0;
    for (int w = 0; w < 2; w++) {
        const int b = 64 * w + lane;
        lb[w] = b < nblk ? box_dist(b) : __builtin_inf();
        if (64 * w < nblk) {
            const double m = wave_min_f64(lb[w]);
            if (m < lb_best) { lb_best = m; b_star = 64 * w + (int)__builtin_ctzll(__builtin_amdgcn_ballot_w64(lb[w] == m)); }
        }
    }
    double t_star;
    const double d_star = seg_dist(b_star, t_star);
    const double ub = wave_min_f64(d_star) + 1e-9; // nothing farther than this can be (or tie) the minimum
    double best = __builtin_inf(), best_t = 0;
    int best_i = 0;
    for (int w = 0; w < 2 && 64 * w < nblk; w++) {
        unsigned long long todo = __builtin_amdgcn_ballot_w64(!(lb[w] > ub)); // (a NaN pose keeps every block)
        if (64 * w + 64 > nblk) todo &= (~0ull) >> (64 * w + 64 - nblk);
        while (todo) {
            const int b = 64 * w + (int)__builtin_ctzll(todo);
            todo &= todo - 1;
            double t = t_star;
            const double d = b == b_star ? d_star : seg_dist(b, t); // (wave-uniform choice)
            const double m = wave_min_f64(d);
            if (m < best) { // strict: an equal distance in an earlier block keeps the lower index (np.argmin)
                const int l = (int)__builtin_ctzll(__builtin_amdgcn_ballot_w64(d == m)); // first lane = lowest segment index
                best = m; best_i = 64 * b + l; best_t = readlane_f64(t, l);
            }
        }
    }

    bool have = false;
    double lx = 0, ly = 0, lv = 0;
    if (s_degenerate) {
        // Two equal consecutive waypoints: t = 0/0 = NaN on that segment for EVERY pose, np.argmin returns it (the
        // first NaN), and its NaN distance fails both `<` tests of _get_current_waypoint (:189-204): plan() answers
        // (4.0, 0.0) whatever the pose.  (fmin-based reductions would drop the NaN and plan as if it were not there.)
    } else if (best < a.lookahead) {
        // first_point_on_trajectory_intersecting_circle(position, lookahead, wpts, i + t, wrap=True) (:49-129)
        const double targ = (double)best_i + best_t;
        const int start_i = (int)targ;
        const double start_t = fmod(targ, 1.0);
        int i2 = 0;
        bool found = false;
        for (int base = start_i; base < M - 1 && !found; base += 64) {
            const int i = base + lane;
            bool hit = false;
            double t1, t2;
            if (i < M - 1 && seg_circle(s_wp[3 * i], s_wp[3 * i + 1], s_wp[3 * i + 3], s_wp[3 * i + 4], px, py, a.lookahead, t1, t2)) {
                const bool h1 = t1 >= 0.0 && t1 <= 1.0, h2 = t2 >= 0.0 && t2 <= 1.0;
                hit = i == start_i ? ((h1 && t1 >= start_t) || (h2 && t2 >= start_t)) : (h1 || h2);
            }
            const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
            if (m) { found = true; i2 = base + (int)__builtin_ctzll(m); }
        }
        for (int base = -1; base < start_i && !found; base += 64) {
            const int i = base + lane;
            bool hit = false;
            double t1, t2;
            if (i < start_i) {
                const int i0 = i < 0 ? i + M : i, i1 = (i + 1) % M; // Python's % on a negative index
                if (seg_circle(s_wp[3 * i0], s_wp[3 * i0 + 1], s_wp[3 * i1], s_wp[3 * i1 + 1], px, py, a.lookahead, t1, t2))
                    hit = (t1 >= 0.0 && t1 <= 1.0) || (t2 >= 0.0 && t2 <= 1.0);
            }
            const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
            if (m) { found = true; i2 = base + (int)__builtin_ctzll(m); }
        }
        if (found) {
            const int j = i2 < 0 ? i2 + M : i2; // wpts[i2, :] with Python negative indexing
            have = true; lx = s_wp[3 * j]; ly = s_wp[3 * j + 1]; lv = s_wp[3 * best_i + 2];
        }
    } else if (best < a.max_reacquire) {
        have = true; lx = s_wp[3 * best_i]; ly = s_wp[3 * best_i + 1]; lv = s_wp[3 * best_i + 2];
    }
    if (lane == 0) { s_res[wave][0] = lx; s_res[wave][1] = ly; s_res[wave][2] = lv; s_res[wave][3] = have ? 1.0 : 0.0; }
    __syncthreads();
    // get_actuation (:131-144) for the block's cars, one LANE per car (the trigonometry would otherwise be issued
    // wave-wide for a single car)
    if (wave == 0 && lane < PP_WAVES) {
        const int c = blockIdx.x * PP_WAVES + lane;
        if (c < a.n) {
            double steer = 0.0, speed = 4.0; // plan(): lookahead_point is None -> (4.0, 0.0)
            if (s_res[lane][3] != 0.0) {
                const double cx = a.state[(size_t)c * 7], cy = a.state[(size_t)c * 7 + 1], theta = a.state[(size_t)c * 7 + 4];
                const double wy = sin(-theta) * (s_res[lane][0] - cx) + cos(-theta) * (s_res[lane][1] - cy);
                speed = s_res[lane][2];
                if (fabs(wy) < 1e-6) steer = 0.;
                else {
                    const double radius = 1 / (2.0 * wy / (a.lookahead * a.lookahead));
                    steer = atan(a.wheelbase / radius);
                }
                speed = a.vgain * speed;
            }
            a.actions[(size_t)c * 2] = steer;
            a.actions[(size_t)c * 2 + 1] = speed;
        }
    }
}
#endif

// ------------------------------------------------------------------ one LANE per car behind a grid of candidate lists (round 5)
// The wave-per-car kernel above prunes 64-segment blocks by their boxes, but it still spends a whole wavefront, a copy of the
// raceline in LDS and a dozen wave-wide reductions on every car: 66 us for 65 536 cars.  For a raceline that has been
// PREPARED (f110_pure_pursuit_prepare: once per raceline) the nearest segment is searched among a handful of candidates:
// a uniform grid over the raceline's surroundings holds, per cell, the ascending list of the segments that can be the
// nearest one for ANY pose in the cell.  With c the cell's centre, hd its half diagonal and D = min_j dist(c, seg_j): a
// segment with dist(c, seg) > D + 2 hd lies farther than the nearest one from every point of the cell (|dist(p, s) - dist(c, s)|
// <= |p - c| <= hd for every segment s), so the list holds dist(c, seg) <= D + 2 hd + 1e-6 (the margin is nine orders of
// magnitude above the rounding of the distances).  The arg-min itself is the reference's: the same expression per segment
// (:31-46), strict `<` in ascending index order = np.argmin's first minimum.  A cell with more candidates than the list
// holds (a pose equally far from a long stretch of the raceline), a pose outside the grid and a NaN pose take every
// segment, in the same lane: the same results, slower, and rare.  The look-ahead point (:49-129) is searched segment by
// segment from the nearest one, as the reference does.
constexpr int PG_CAP = 30;            // candidates a cell's list holds
constexpr unsigned PG_ALL = 255;      // count value: take every segment
struct PlanGrid {
    double x0, y0, inv_cell;          // cell (ix, iy) covers x0 + ix / inv_cell ...
    int gw, gh;
    const uint8_t *count;             // [gh * gw]
    const uint16_t *cand;             // [gh * gw][PG_CAP] ascending segment indices
    int degenerate;                   // the raceline has a zero-length segment: plan() answers (0, 4.0) for every pose
};

#if defined(F110_UNIT_CONSUMERS)
static __global__ __launch_bounds__(256) void pure_pursuit_grid_kernel(PlanArgs a, PlanGrid g)
{
    const int car = blockIdx.x * blockDim.x + threadIdx.x;
    if (car >= a.n) return;
    const double *__restrict__ wp = a.waypoints;
    const int M = a.M, nseg = M - 1;
    const double px = a.state[(size_t)car * 7], py = a.state[(size_t)car * 7 + 1], theta = a.state[(size_t)car * 7 + 4];
    double steer = 0.0, speed = 4.0; // plan(): lookahead_point is None -> (4.0, 0.0)
    bool have = false;
    double lx = 0, ly = 0, lv = 0;
    if (!g.degenerate) {
        // nearest_point_on_trajectory (:16-47) over the cell's candidates
        double best = __builtin_inf(), best_t = 0;
        int best_i = 0;
        auto consider = [&](int i) {
            const double x0 = wp[3 * i], y0 = wp[3 * i + 1];
            const double dx = wp[3 * i + 3] - x0, dy = wp[3 * i + 4] - y0;
            const double l2 = dx * dx + dy * dy;
            double t = ((px - x0) * dx + (py - y0) * dy) / l2;
            t = t < 0.0 ? 0.0 : t;
            t = t > 1.0 ? 1.0 : t;
            const double qx = px - (x0 + t * dx), qy = py - (y0 + t * dy);
            const double d = sqrt(qx * qx + qy * qy);
            if (d < best) { best = d; best_i = i; best_t = t; }
        };
        const double fx = floor((px - g.x0) * g.inv_cell), fy = floor((py - g.y0) * g.inv_cell);
        unsigned cnt = PG_ALL;
        size_t cell = 0;
        if (fx >= 0.0 && fx < (double)g.gw && fy >= 0.0 && fy < (double)g.gh) { // (false for a NaN pose)
            cell = (size_t)(int)fy * (size_t)g.gw + (size_t)(int)fx;
            cnt = g.count[cell];
        }
        if (cnt == PG_ALL) for (int i = 0; i < nseg; i++) consider(i);
        else {
            const uint16_t *lst = g.cand + cell * PG_CAP;
            for (unsigned k = 0; k < cnt; k++) consider((int)lst[k]);
        }
        if (best < a.lookahead) {
            // first_point_on_trajectory_intersecting_circle(position, lookahead, wpts, i + t, wrap=True) (:49-129)
            const double targ = (double)best_i + best_t;
            const int start_i = (int)targ;
            const double start_t = fmod(targ, 1.0);
            int i2 = 0;
            bool found = false;
            for (int i = start_i; i < M - 1 && !found; i++) {
                double t1, t2;
                if (seg_circle(wp[3 * i], wp[3 * i + 1], wp[3 * i + 3], wp[3 * i + 4], px, py, a.lookahead, t1, t2)) {
                    const bool h1 = t1 >= 0.0 && t1 <= 1.0, h2 = t2 >= 0.0 && t2 <= 1.0;
                    if (i == start_i ? ((h1 && t1 >= start_t) || (h2 && t2 >= start_t)) : (h1 || h2)) { found = true; i2 = i; }
                }
            }
            for (int i = -1; i < start_i && !found; i++) {
                const int i0 = i < 0 ? i + M : i, i1 = (i + 1) % M; // Python's % on a negative index
                double t1, t2;
                if (seg_circle(wp[3 * i0], wp[3 * i0 + 1], wp[3 * i1], wp[3 * i1 + 1], px, py, a.lookahead, t1, t2))
                    if ((t1 >= 0.0 && t1 <= 1.0) || (t2 >= 0.0 && t2 <= 1.0)) { found = true; i2 = i; }
            }
            if (found) {
                const int j = i2 < 0 ? i2 + M : i2; // wpts[i2, :] with Python negative indexing
                have = true; lx = wp[3 * j]; ly = wp[3 * j + 1]; lv = wp[3 * best_i + 2];
            }
        } else if (best < a.max_reacquire) {
            have = true; lx = wp[3 * best_i]; ly = wp[3 * best_i + 1]; lv = wp[3 * best_i + 2];
        }
    }
    if (have) { // get_actuation (:131-144)
        const double wy = sin(-theta) * (lx - px) + cos(-theta) * (ly - py);
        speed = lv;
        if (fabs(wy) < 1e-6) steer = 0.;
        else {
            const double radius = 1 / (2.0 * wy / (a.lookahead * a.lookahead));
            steer = atan(a.wheelbase / radius);
        }
        speed = a.vgain * speed;
    }
    a.actions[(size_t)car * 2] = steer;
    a.actions[(size_t)car * 2 + 1] = speed;
}
#endif

// ------------------------------------------------------------------ many tracks / racelines of any length
// The kernel above stages ONE raceline in LDS (<= 6 400 points).  The reference loads any CSV
// (examples/waypoint_follow.py:162) and every F110Env has its own planner and track; a shard that drives K random tracks
// (F110VecEnv.randomize_tracks) therefore needs a raceline PER CAR, and a long raceline must not be refused.  This
// form reads waypoints and block boxes from global memory (racelines are a few tens of KB: L1 / L2 hits), takes the
// raceline of every car from a slot table, and walks the block bounds in 64-block words without a register copy of
// them, so the number of points is unbounded.  Same results as pure_pursuit_kernel (the tests demand `==`).
struct TrackSet {
    const double *waypoints;  // [total,3] the K racelines back to back
    const int32_t *offsets;   // [K+1] first row of raceline k (offsets[K] = total)
    int K;
    double *boxes;            // workspace [(total >> 6) + K][5]: xmin, xmax, ymin, ymax of 64-segment block b of raceline k at
                              // row (offsets[k] >> 6) + k + b; [4] = 1.0 if the block holds a zero-length segment
};

__device__ inline size_t track_box_row(const TrackSet &t, int k) { return (size_t)(t.offsets[k] >> 6) + (size_t)k; }

// one wave per (raceline, 64-segment block): grid (ceil(max_blocks / 4), K), 256 threads
#if defined(F110_UNIT_CONSUMERS)
static __global__ __launch_bounds__(256) void track_boxes_kernel(TrackSet t)
{
    const int k = blockIdx.y;
    const int M = t.offsets[k + 1] - t.offsets[k], nseg = M - 1, nblk = (nseg + 63) >> 6;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= nblk) return;
    const double *wp = t.waypoints + (size_t)t.offsets[k] * 3;
    const int i = 64 * b + lane;
    const bool ok = i < nseg;
    const double x0 = ok ? wp[3 * i] : 0, y0 = ok ? wp[3 * i + 1] : 0, x1 = ok ? wp[3 * i + 3] : 0, y1 = ok ? wp[3 * i + 4] : 0;
    const double inf = __builtin_inf();
    const double xl = wave_min_f64(ok ? fmin(x0, x1) : inf), xh = wave_max_f64(ok ? fmax(x0, x1) : -inf);
    const double yl = wave_min_f64(ok ? fmin(y0, y1) : inf), yh = wave_max_f64(ok ? fmax(y0, y1) : -inf);
    const bool degenerate = ok && ((x1 - x0) * (x1 - x0) + (y1 - y0) * (y1 - y0) == 0.0);
    const bool any_deg = __builtin_amdgcn_ballot_w64(degenerate) != 0ull;
    if (lane == 0) {
        double *bx = t.boxes + (track_box_row(t, k) + (size_t)b) * 5;
        bx[0] = xl; bx[1] = xh; bx[2] = yl; bx[3] = yh; bx[4] = any_deg ? 1.0 : 0.0;
    }
}
#endif

struct PlanTracksArgs {
    TrackSet t;
    const int32_t *track_of_car; // [n] raceline of every car, or NULL (all on raceline 0)
    double lookahead, vgain, wheelbase, max_reacquire;
    const double *state;         // [n,7]
    int n;
    double *actions;             // [n,2]
};

constexpr int PPG_WAVES = 4;

#if defined(F110_UNIT_CONSUMERS)
static __global__ __launch_bounds__(PPG_WAVES * 64) void pure_pursuit_tracks_kernel(PlanTracksArgs a)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int car = blockIdx.x * PPG_WAVES + wave;
    if (car >= a.n) return;
    int k = a.track_of_car ? a.track_of_car[car] : 0;
    k = __builtin_amdgcn_readfirstlane(k);
    if (k < 0 || k >= a.t.K) k = 0; // (the host checks the table it is given; a device-side table is the caller's)
    const int M = a.t.offsets[k + 1] - a.t.offsets[k], nseg = M - 1, nblk = (nseg + 63) >> 6;
    const double *__restrict__ wp = a.t.waypoints + (size_t)a.t.offsets[k] * 3;
    // boxes == NULL (f110_pure_pursuit on a raceline too long for LDS, which has no workspace): no pruning, every
    // block is evaluated -- same results, more work
    const bool has_box = a.t.boxes != nullptr;
    const double *__restrict__ box = has_box ? a.t.boxes + track_box_row(a.t, k) * 5 : nullptr;
    const double px = a.state[(size_t)car * 7], py = a.state[(size_t)car * 7 + 1];
    bool degenerate = false;

    auto seg_dist = [&](int b, double &t) {
        const int i = 64 * b + lane;
        t = 0.0;
        if (i >= nseg) return __builtin_inf();
        const double x0 = wp[3 * i], y0 = wp[3 * i + 1];
        const double dx = wp[3 * i + 3] - x0, dy = wp[3 * i + 4] - y0;
        const double l2 = dx * dx + dy * dy;
        if (!has_box && l2 == 0.0) degenerate = true;
        t = ((px - x0) * dx + (py - y0) * dy) / l2;
        t = t < 0.0 ? 0.0 : t;
        t = t > 1.0 ? 1.0 : t;
        const double qx = px - (x0 + t * dx), qy = py - (y0 + t * dy);
        return sqrt(qx * qx + qy * qy);
    };
    auto box_dist = [&](int b) {
        if (!has_box) return 0.0;
        const double ex = fmax(fmax(box[5 * b] - px, px - box[5 * b + 1]), 0.0);
        const double ey = fmax(fmax(box[5 * b + 2] - py, py - box[5 * b + 3]), 0.0);
        return sqrt(ex * ex + ey * ey);
    };

    // nearest_point_on_trajectory (:16-47): pass 1 finds the block whose box is closest (lane j of word w = block
    // 64 w + j) and an upper bound from its nearest segment; pass 2 evaluates, in index order, the blocks whose box
    // is within that bound.  The bounds are recomputed per word instead of being kept: any number of blocks.
    const int nwords = (nblk + 63) >> 6;
    double lb_best = __builtin_inf();
    int b_star = 0;
    for (int w = 0; w < nwords; w++) {
        const int b = 64 * w + lane;
        const double lb = b < nblk ? box_dist(b) : __builtin_inf();
        degenerate = degenerate || (has_box && b < nblk && box[5 * b + 4] != 0.0);
        const double m = wave_min_f64(lb);
        if (m < lb_best) { lb_best = m; b_star = 64 * w + (int)__builtin_ctzll(__builtin_amdgcn_ballot_w64(lb == m)); }
    }
    double t_star;
    const double d_star = seg_dist(b_star, t_star);
    const double ub = wave_min_f64(d_star) + 1e-9;
    double best = __builtin_inf(), best_t = 0;
    int best_i = 0;
    for (int w = 0; w < nwords; w++) {
        const int bl = 64 * w + lane;
        const double lb = bl < nblk ? box_dist(bl) : __builtin_inf();
        unsigned long long todo = __builtin_amdgcn_ballot_w64(bl < nblk && !(lb > ub)); // (a NaN pose keeps every block)
        while (todo) {
            const int b = 64 * w + (int)__builtin_ctzll(todo);
            todo &= todo - 1;
            double t = t_star;
            const double d = b == b_star ? d_star : seg_dist(b, t);
            const double m = wave_min_f64(d);
            if (m < best) {
                const int l = (int)__builtin_ctzll(__builtin_amdgcn_ballot_w64(d == m));
                best = m; best_i = 64 * b + l; best_t = readlane_f64(t, l);
            }
        }
    }

    degenerate = __builtin_amdgcn_ballot_w64(degenerate) != 0ull; // (without boxes every segment has been looked at by now)
    bool have = false;
    double lx = 0, ly = 0, lv = 0;
    if (degenerate) {
        // a zero-length segment makes t = 0/0 = NaN there for EVERY pose; the reference's np.argmin returns that
        // segment, its NaN distance fails both `<` tests (:189-204) and plan() answers (4.0, 0.0)
    } else if (best < a.lookahead) {
        const double targ = (double)best_i + best_t;
        const int start_i = (int)targ;
        const double start_t = fmod(targ, 1.0);
        int i2 = 0;
        bool found = false;
        for (int base = start_i; base < M - 1 && !found; base += 64) {
            const int i = base + lane;
            bool hit = false;
            double t1, t2;
            if (i < M - 1 && seg_circle(wp[3 * i], wp[3 * i + 1], wp[3 * i + 3], wp[3 * i + 4], px, py, a.lookahead, t1, t2)) {
                const bool h1 = t1 >= 0.0 && t1 <= 1.0, h2 = t2 >= 0.0 && t2 <= 1.0;
                hit = i == start_i ? ((h1 && t1 >= start_t) || (h2 && t2 >= start_t)) : (h1 || h2);
            }
            const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
            if (m) { found = true; i2 = base + (int)__builtin_ctzll(m); }
        }
        for (int base = -1; base < start_i && !found; base += 64) {
            const int i = base + lane;
            bool hit = false;
            double t1, t2;
            if (i < start_i) {
                const int i0 = i < 0 ? i + M : i, i1 = (i + 1) % M;
                if (seg_circle(wp[3 * i0], wp[3 * i0 + 1], wp[3 * i1], wp[3 * i1 + 1], px, py, a.lookahead, t1, t2))
                    hit = (t1 >= 0.0 && t1 <= 1.0) || (t2 >= 0.0 && t2 <= 1.0);
            }
            const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
            if (m) { found = true; i2 = base + (int)__builtin_ctzll(m); }
        }
        if (found) {
            const int j = i2 < 0 ? i2 + M : i2;
            have = true; lx = wp[3 * j]; ly = wp[3 * j + 1]; lv = wp[3 * best_i + 2];
        }
    } else if (best < a.max_reacquire) {
        have = true; lx = wp[3 * best_i]; ly = wp[3 * best_i + 1]; lv = wp[3 * best_i + 2];
    }
    // get_actuation (:131-144): lane 0 of the car's wave
    if (lane == 0) {
        double steer = 0.0, speed = 4.0;
        if (have) {
            const double theta = a.state[(size_t)car * 7 + 4];
            const double wy = sin(-theta) * (lx - px) + cos(-theta) * (ly - py);
            speed = lv;
            if (fabs(wy) < 1e-6) steer = 0.;
            else {
                const double radius = 1 / (2.0 * wy / (a.lookahead * a.lookahead));
                steer = atan(a.wheelbase / radius);
            }
            speed = a.vgain * speed;
        }
        a.actions[(size_t)car * 2] = steer;
        a.actions[(size_t)car * 2 + 1] = speed;
    }
}
#endif

} // namespace f110
