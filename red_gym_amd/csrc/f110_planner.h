// f110_planner.h -- batched pure-pursuit planner: the caller on the other
// side of F110Env.step, reference examples/waypoint_follow.py:15-217, one lane per car.
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

__global__ __launch_bounds__(256) void pure_pursuit_kernel(PlanArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double s_wp[]; // [M,3]
    for (int i = threadIdx.x; i < a.M * 3; i += blockDim.x) s_wp[i] = a.waypoints[i];
    __syncthreads();
    const int car = blockIdx.x * blockDim.x + threadIdx.x;
    if (car >= a.n) return;
    const int M = a.M;
    const double px = a.state[(size_t)car * 7], py = a.state[(size_t)car * 7 + 1], theta = a.state[(size_t)car * 7 + 4];

    // nearest_point_on_trajectory (:16-47): first arg-min over the M-1 segments
    double best = __builtin_inf(), best_t = 0;
    int best_i = 0;
    for (int i = 0; i < M - 1; i++) {
        const double x0 = s_wp[3 * i], y0 = s_wp[3 * i + 1];
        const double dx = s_wp[3 * i + 3] - x0, dy = s_wp[3 * i + 4] - y0;
        const double l2 = dx * dx + dy * dy;
        double t = ((px - x0) * dx + (py - y0) * dy) / l2;
        t = t < 0.0 ? 0.0 : t;
        t = t > 1.0 ? 1.0 : t;
        const double qx = px - (x0 + t * dx), qy = py - (y0 + t * dy);
        const double dist = sqrt(qx * qx + qy * qy);
        if (dist < best) { best = dist; best_t = t; best_i = i; }
    }

    double steer = 0.0, speed = 4.0; // plan(): lookahead_point is None -> (4.0, 0.0)
    bool have = false;
    double lx = 0, ly = 0, lv = 0;
    if (best < a.lookahead) {
        // first_point_on_trajectory_intersecting_circle(position, lookahead, wpts, i + t, wrap=True)
        const double targ = (double)best_i + best_t;
        const int start_i = (int)targ;
        const double start_t = fmod(targ, 1.0);
        int i2 = 0;
        bool found = false;
        for (int i = start_i; i < M - 1 && !found; i++) {
            double t1, t2;
            if (!seg_circle(s_wp[3 * i], s_wp[3 * i + 1], s_wp[3 * i + 3], s_wp[3 * i + 4], px, py, a.lookahead, t1, t2)) continue;
            if (i == start_i) {
                if (t1 >= 0.0 && t1 <= 1.0 && t1 >= start_t) { found = true; i2 = i; }
                else if (t2 >= 0.0 && t2 <= 1.0 && t2 >= start_t) { found = true; i2 = i; }
            } else if (t1 >= 0.0 && t1 <= 1.0) { found = true; i2 = i; }
            else if (t2 >= 0.0 && t2 <= 1.0) { found = true; i2 = i; }
        }
        for (int i = -1; i < start_i && !found; i++) {
            const int i0 = i < 0 ? i + M : i, i1 = (i + 1) % M; // Python's % on a negative index
            double t1, t2;
            if (!seg_circle(s_wp[3 * i0], s_wp[3 * i0 + 1], s_wp[3 * i1], s_wp[3 * i1 + 1], px, py, a.lookahead, t1, t2)) continue;
            if (t1 >= 0.0 && t1 <= 1.0) { found = true; i2 = i; }
            else if (t2 >= 0.0 && t2 <= 1.0) { found = true; i2 = i; }
        }
        if (found) {
            const int j = i2 < 0 ? i2 + M : i2; // wpts[i2, :] with Python negative indexing
            have = true; lx = s_wp[3 * j]; ly = s_wp[3 * j + 1]; lv = s_wp[3 * best_i + 2];
        }
    } else if (best < a.max_reacquire) {
        have = true; lx = s_wp[3 * best_i]; ly = s_wp[3 * best_i + 1]; lv = s_wp[3 * best_i + 2];
    }
    if (have) {
        // get_actuation (:131-144)
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

} // namespace f110
