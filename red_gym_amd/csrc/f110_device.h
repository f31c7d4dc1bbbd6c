// f110_device.h -- device-side building blocks of the batched F1TENTH step for
// gfx950.  Every function states the reference lines it reproduces
// (paths relative to /root/reference/gym/f110_gym/envs/).
//
// Numerics contract: IEEE fp64, operations in the reference's order, no FMA
// contraction (file-level pragma + -ffp-contract=off), so that cell indices, LUT
// indices, iTTC / GJK / lap decisions are bit-exact against the CPU oracle on
// identical inputs.  sin/cos/tan/atan2 come from the device math library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

#define F110_PI 3.141592653589793

namespace f110 {

enum { P_MU, P_CSF, P_CSR, P_LF, P_LR, P_H, P_M, P_I, P_SMIN, P_SMAX, P_SVMIN, P_SVMAX,
       P_VSWITCH, P_AMAX, P_VMIN, P_VMAX, P_WIDTH, P_LENGTH, P_COUNT };

struct Params { double v[P_COUNT]; };

// ---------------------------------------------------------------- dynamics
// dynamic_models.py:30-60
__device__ inline double accl_constraints(double vel, double accl, double v_switch, double a_max,
                                          double v_min, double v_max)
{
    double pos_limit;
    if (vel > v_switch) pos_limit = a_max * v_switch / vel;
    else pos_limit = a_max;
    if ((vel <= v_min && accl <= 0) || (vel >= v_max && accl >= 0)) accl = 0.;
    else if (accl <= -a_max) accl = -a_max;
    else if (accl >= pos_limit) accl = pos_limit;
    return accl;
}

// dynamic_models.py:63-87
__device__ inline double steering_constraint(double steering_angle, double steering_velocity,
                                             double s_min, double s_max, double sv_min, double sv_max)
{
    if ((steering_angle <= s_min && steering_velocity <= 0) ||
        (steering_angle >= s_max && steering_velocity >= 0)) steering_velocity = 0.;
    else if (steering_velocity <= sv_min) steering_velocity = sv_min;
    else if (steering_velocity >= sv_max) steering_velocity = sv_max;
    return steering_velocity;
}

// dynamic_models.py:124-176 (with the kinematic branch :91-121 inlined)
__device__ inline void vehicle_dynamics_st(const double x[7], double sv_in, double accl_in,
                                           const Params &P, double f[7])
{
    const double *p = P.v;
    const double g = 9.81;
    const double mu = p[P_MU], C_Sf = p[P_CSF], C_Sr = p[P_CSR], lf = p[P_LF], lr = p[P_LR];
    const double h = p[P_H], m = p[P_M], I = p[P_I];
    double u0 = steering_constraint(x[2], sv_in, p[P_SMIN], p[P_SMAX], p[P_SVMIN], p[P_SVMAX]);
    double u1 = accl_constraints(x[3], accl_in, p[P_VSWITCH], p[P_AMAX], p[P_VMIN], p[P_VMAX]);
    if (fabs(x[3]) < 0.5) {
        double lwb = lf + lr;
        // vehicle_dynamics_ks re-applies the constraints to the constrained input (:113)
        double k0 = steering_constraint(x[2], u0, p[P_SMIN], p[P_SMAX], p[P_SVMIN], p[P_SVMAX]);
        double k1 = accl_constraints(x[3], u1, p[P_VSWITCH], p[P_AMAX], p[P_VMIN], p[P_VMAX]);
        double tx2 = tan(x[2]);
        double cx2 = cos(x[2]);
        double s4, c4;
        sincos(x[4], &s4, &c4); // one argument reduction and one pair of kernels for both (the library's sin and cos each evaluate the pair)
        f[0] = x[3] * c4;
        f[1] = x[3] * s4;
        f[2] = k0;
        f[3] = k1;
        f[4] = x[3] / lwb * tx2;
        f[5] = u1 / lwb * tx2 + x[3] / (lwb * (cx2 * cx2)) * u0;
        f[6] = 0;
    } else {
        double glr_m = g * lr - u1 * h;
        double glf_p = g * lf + u1 * h;
        double ang = x[6] + x[4];
        double sa, ca;
        sincos(ang, &sa, &ca);
        f[0] = x[3] * ca;
        f[1] = x[3] * sa;
        f[2] = u0;
        f[3] = u1;
        f[4] = x[5];
        f[5] = -mu * m / (x[3] * I * (lr + lf)) * ((lf * lf) * C_Sf * glr_m + (lr * lr) * C_Sr * glf_p) * x[5]
             + mu * m / (I * (lr + lf)) * (lr * C_Sr * glf_p - lf * C_Sf * glr_m) * x[6]
             + mu * m / (I * (lr + lf)) * lf * C_Sf * glr_m * x[2];
        f[6] = (mu / ((x[3] * x[3]) * (lr + lf)) * (C_Sr * glf_p * lr - C_Sf * glr_m * lf) - 1) * x[5]
             - mu / (x[3] * (lr + lf)) * (C_Sr * glf_p + C_Sf * glr_m) * x[6]
             + mu / (x[3] * (lr + lf)) * (C_Sf * glr_m) * x[2];
    }
}

// dynamic_models.py:179-221
__device__ inline void pid(double speed, double steer, double current_speed, double current_steer,
                           double max_sv, double max_a, double max_v, double min_v,
                           double &accl, double &sv)
{
    double steer_diff = steer - current_steer;
    if (fabs(steer_diff) > 1e-4) sv = (steer_diff / fabs(steer_diff)) * max_sv;
    else sv = 0.0;
    double vel_diff = speed - current_speed;
    double kp;
    if (current_speed > 0.) {
        if (vel_diff > 0) kp = 10.0 * max_a / max_v;
        else kp = 10.0 * max_a / (-min_v);
    } else {
        if (vel_diff > 0) kp = 2.0 * max_a / max_v;
        else kp = 2.0 * max_a / (-min_v);
    }
    accl = kp * vel_diff;
}

// base_classes.py:254-402: steering delay FIFO, pid, RK4 / Euler, yaw wrap.
__device__ inline void update_pose(double st[7], double sbuf[2], int &scnt, double raw_steer,
                                   double vel, const Params &P, double time_step, int integrator)
{
    double steer = 0.;
    if (scnt < 2) {
        steer = 0.;
        sbuf[1] = sbuf[0];
        sbuf[0] = raw_steer;
        scnt++;
    } else {
        steer = sbuf[1];
        sbuf[1] = sbuf[0];
        sbuf[0] = raw_steer;
    }
    double accl, sv;
    pid(vel, steer, st[3], st[2], P.v[P_SVMAX], P.v[P_AMAX], P.v[P_VMAX], P.v[P_VMIN], accl, sv);
    if (integrator == 1) {
        double k1[7], k2[7], k3[7], k4[7], tmp[7];
        vehicle_dynamics_st(st, sv, accl, P, k1);
#pragma unroll
        for (int i = 0; i < 7; i++) tmp[i] = st[i] + time_step * (k1[i] / 2);
        vehicle_dynamics_st(tmp, sv, accl, P, k2);
#pragma unroll
        for (int i = 0; i < 7; i++) tmp[i] = st[i] + time_step * (k2[i] / 2);
        vehicle_dynamics_st(tmp, sv, accl, P, k3);
#pragma unroll
        for (int i = 0; i < 7; i++) tmp[i] = st[i] + time_step * k3[i];
        vehicle_dynamics_st(tmp, sv, accl, P, k4);
        double w = time_step * (1. / 6);
#pragma unroll
        for (int i = 0; i < 7; i++) st[i] = st[i] + w * (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]);
    } else {
        double f[7];
        vehicle_dynamics_st(st, sv, accl, P, f);
#pragma unroll
        for (int i = 0; i < 7; i++) st[i] = st[i] + time_step * f[i];
    }
    if (st[4] > 2 * F110_PI) st[4] = st[4] - 2 * F110_PI;
    else if (st[4] < 0) st[4] = st[4] + 2 * F110_PI;
}

// ---------------------------------------------------------------- collision
// collision_models.py:219-260, vertex order rl, rr, fr, fl (:259)
__device__ inline void get_vertices(double x, double y, double th, double length, double width,
                                    double out[4][2])
{
    double c, s;
    sincos(th, &s, &c);
    const double hx[4] = {-length / 2, -length / 2, length / 2, length / 2};
    const double hy[4] = {width / 2, -width / 2, -width / 2, width / 2};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        out[k][0] = ((c * hx[k] + (-s) * hy[k]) + 0. * 0.) + x * 1.;
        out[k][1] = ((s * hx[k] + c * hy[k]) + 0. * 0.) + y * 1.;
    }
}

// collision_models.py:82-92 first arg-max of v . d
__device__ inline int furthest(const double v[4][2], double dx, double dy)
{
    int best = 0;
    double bv = v[0][0] * dx + v[0][1] * dy;
#pragma unroll
    for (int i = 1; i < 4; i++) {
        double t = v[i][0] * dx + v[i][1] * dy;
        if (t > bv) { bv = t; best = i; }
    }
    return best;
}

__device__ inline void vsel(const double v[4][2], int i, double &ox, double &oy)
{
    // register-friendly select (avoids runtime-indexed private arrays)
    ox = i == 0 ? v[0][0] : i == 1 ? v[1][0] : i == 2 ? v[2][0] : v[3][0];
    oy = i == 0 ? v[0][1] : i == 1 ? v[1][1] : i == 2 ? v[2][1] : v[3][1];
}

// collision_models.py:96-110
__device__ inline void support(const double v1[4][2], const double v2[4][2], double dx, double dy,
                               double &ax, double &ay)
{
    int i = furthest(v1, dx, dy);
    int j = furthest(v2, -dx, -dy);
    double ix, iy, jx, jy;
    vsel(v1, i, ix, iy);
    vsel(v2, j, jx, jy);
    ax = ix - jx;
    ay = iy - jy;
}

// collision_models.py:52-64  tripleProduct(a,b,c) = b*(a.c) - a*(b.c)
__device__ inline void triple(double ax, double ay, double bx, double by, double cx, double cy,
                              double &ox, double &oy)
{
    double ac = ax * cx + ay * cy;
    double bc = bx * cx + by * cy;
    ox = bx * ac - ax * bc;
    oy = by * ac - ay * bc;
}

// collision_models.py:114-182  GJK on two quads
__device__ inline bool gjk_collision(const double v1[4][2], const double v2[4][2])
{
    int index = 0;
    double s0x = 0, s0y = 0, s1x = 0, s1y = 0, s2x = 0, s2y = 0; // simplex rows
    double p1x = ((v1[0][0] + v1[1][0]) + v1[2][0]) + v1[3][0];
    double p1y = ((v1[0][1] + v1[1][1]) + v1[2][1]) + v1[3][1];
    double p2x = ((v2[0][0] + v2[1][0]) + v2[2][0]) + v2[3][0];
    double p2y = ((v2[0][1] + v2[1][1]) + v2[2][1]) + v2[3][1];
    // np.sum(axis=0) starts from 0: (((0+v0)+v1)+v2)+v3 == ((v0+v1)+v2)+v3
    p1x /= 4; p1y /= 4; p2x /= 4; p2y /= 4;
    double dx = p1x - p2x, dy = p1y - p2y;
    if (dx == 0 && dy == 0) dx = 1.0;
    double ax, ay;
    support(v1, v2, dx, dy, ax, ay);
    s0x = ax; s0y = ay;
    if (dx * ax + dy * ay <= 0) return false;
    dx = -ax; dy = -ay;
    int iter_count = 0;
    while (iter_count < 1000) {
        support(v1, v2, dx, dy, ax, ay);
        index += 1;
        if (index == 1) { s1x = ax; s1y = ay; } else { s2x = ax; s2y = ay; }
        if (dx * ax + dy * ay <= 0) return false;
        double aox = -ax, aoy = -ay;
        if (index < 2) {
            double abx = s0x - ax, aby = s0y - ay;
            triple(abx, aby, aox, aoy, abx, aby, dx, dy);
            if (sqrt(dx * dx + dy * dy) < 1e-10) {
                dx = aby;          // perpendicular(ab) :35-48
                dy = -1 * abx;
            }
            continue;
        }
        double abx = s1x - ax, aby = s1y - ay;
        double acx = s0x - ax, acy = s0y - ay;
        double px, py;
        triple(abx, aby, acx, acy, acx, acy, px, py); // acperp
        if (px * aox + py * aoy >= 0) {
            dx = px; dy = py;
        } else {
            double qx, qy;
            triple(acx, acy, abx, aby, abx, aby, qx, qy); // abperp
            if (qx * aox + qy * aoy < 0) return true;
            s0x = s1x; s0y = s1y;
            dx = qx; dy = qy;
        }
        s1x = s2x; s1y = s2y;
        index -= 1;
        iter_count += 1;
    }
    return false;
}

// ---------------------------------------------------------------- opponent ray cast
// laser_models.py:250-280 with v3 = (cos(beam_theta+pi/2), sin(beam_theta+pi/2)) precomputed
__device__ inline double get_range(double ox, double oy, double v3x, double v3y, double vax,
                                   double vay, double vbx, double vby)
{
    double v1x = ox - vax, v1y = oy - vay;
    double v2x = vbx - vax, v2y = vby - vay;
    double denom = v2x * v3x + v2y * v3y;
    double distance = __builtin_inf();
    if (fabs(denom) > 0.0) {
        // d1 = cross/denom >= 0, 0 <= d2 = dot/denom <= 1 (:271-274) decided without dividing:
        // the sign of an IEEE quotient is the sign product, and fl(q) <= 1 <=> q <= 1.
        const double cr = v2x * v1y - v2y * v1x;
        const double dt = v1x * v3x + v1y * v3y;
        const bool dpos = denom > 0.0;
        const bool d1_ok = (cr == 0.0) || ((cr > 0.0) == dpos);
        const bool d2_ge0 = (dt == 0.0) || ((dt > 0.0) == dpos);
        const bool d2_le1 = dpos ? (dt <= denom) : (dt >= denom);
        if (d1_ok && d2_ge0 && d2_le1) distance = cr / denom;
    } else {
        // are_collinear(o, va, vb) :233-247
        double bax = vax - ox, bay = vay - oy;
        double cax = ox - vbx, cay = oy - vby;
        if (fabs(bax * cay - bay * cax) < 1e-8) {
            double ebx = vbx - ox, eby = vby - oy;
            double da = sqrt(bax * bax + bay * bay);
            double db = sqrt(ebx * ebx + eby * eby);
            distance = da < db ? da : db;
        }
    }
    return distance;
}

// First-index arg-min of |scan_angles[i] - a| (np.argmin semantics) for the strictly
// increasing beam-angle table (base_classes.py:131-132): fl(scan_angles[i] - a) is monotone
// in i, so |.| falls while negative and rises once positive -- the minimum sits at the last
// entry <= a or its successor.  A linear estimate lands within a step or two of it; the two
// loops make the result independent of the estimate.  NaN -> nb (caller bails out).
__device__ inline int argmin_abs_diff_sorted(const double *__restrict__ scan_angles, int nb, double a)
{
    if (!(a == a)) return nb;
    const double sa0 = scan_angles[0];
    const double inv_incr = (double)(nb - 1) / (scan_angles[nb - 1] - sa0);
    double est = (a - sa0) * inv_incr;
    est = est < 0. ? 0. : (est > (double)(nb - 1) ? (double)(nb - 1) : est);
    int k = (int)est;
    while (k + 1 < nb && scan_angles[k + 1] <= a) k++;
    while (k > 0 && scan_angles[k] > a) k--;
    // candidates k and k+1 (first minimum wins)
    const double vk = fabs(scan_angles[k] - a);
    if (k + 1 < nb && fabs(scan_angles[k + 1] - a) < vk) return k + 1;
    return k;
}

// laser_models.py:283-315: lane v (mod 4) handles corner v, the span is the min/max of
// the four indices.
__device__ inline void blocked_view_indices(double px, double py, double pyaw, const double verts[4][2],
                                            const double *__restrict__ scan_angles, int nb, int lane,
                                            int &min_ind, int &max_ind)
{
    const double ex = cos(pyaw), ey = sin(pyaw);
    const double ego_ang = atan2(ey, ex);
    const int v = lane & 3;
    const double cx = v == 0 ? verts[0][0] : v == 1 ? verts[1][0] : v == 2 ? verts[2][0] : verts[3][0];
    const double cy = v == 0 ? verts[0][1] : v == 1 ? verts[1][1] : v == 2 ? verts[2][1] : verts[3][1];
    const double vx = cx - px, vy = cy - py;
    const double norm = sqrt(vx * vx + vy * vy);
    const double ux = vx / norm, uy = vy / norm;
    double angle = ego_ang - atan2(uy, ux);
    if (angle > F110_PI) angle = angle - 2 * F110_PI;
    else if (angle < -F110_PI) angle = angle + 2 * F110_PI;
    const int ind = argmin_abs_diff_sorted(scan_angles, nb, -angle);
    int lo = ind, hi = ind;
#pragma unroll
    for (int off = 1; off <= 2; off <<= 1) {
        const int o_lo = __shfl_xor(lo, off), o_hi = __shfl_xor(hi, off);
        lo = o_lo < lo ? o_lo : lo;
        hi = o_hi > hi ? o_hi : hi;
    }
    min_ind = lo;
    max_ind = hi;
}

// laser_models.py:319-346 on a wave: beams strided over lanes.  The scan lives in
// global memory as fp64 and/or fp32; (float)min(a, b) == min((float)a, (float)b)
// because rounding is monotonic, so the fp32 observation can be updated in place.
// The beam direction v3 = (cos, sin)(pose_yaw + scan_angle + pi/2) (:265) is formed by
// the angle-addition identity from one per-car sincos and a {cos, sin}(scan_angle) table
// instead of a per-beam fp64 sincos: same ~1e-16 accuracy class as the libm-vs-NumPy
// difference the tolerance already covers, at a tenth of the instructions (an opponent
// directly behind the car blocks the whole 2*pi scan, i.e. all 1080 beams).
__device__ inline void ray_cast_wave(double px, double py, double pyaw, const double verts[4][2],
                                     const double *__restrict__ scan_angles, const double2 *__restrict__ beam_cs,
                                     int nb, int lane, double *scan64, float *scan32, int *span_out)
{
    int min_ind, max_ind;
    blocked_view_indices(px, py, pyaw, verts, scan_angles, nb, lane, min_ind, max_ind);
    if (span_out && lane == 0) { span_out[0] = min_ind; span_out[1] = max_ind; }
    if (min_ind > nb - 1 || max_ind > nb - 1) return; // only reachable with NaN inputs
    const double A = pyaw + F110_PI / 2.;
    const double cA = cos(A), sA = sin(A);
    for (int i = min_ind + lane; i <= max_ind; i += 64) {
        const double2 cs = beam_cs[i];
        const double v3x = cA * cs.x - sA * cs.y, v3y = sA * cs.x + cA * cs.y;
        double best = __builtin_inf();
#pragma unroll
        for (int j = 0; j < 4; j++) {
            int jn = (j + 1) & 3;
            double r = get_range(px, py, v3x, v3y, verts[j][0], verts[j][1], verts[jn][0], verts[jn][1]);
            if (r < best) best = r;
        }
        if (scan64) { double cur = scan64[i]; if (best < cur) scan64[i] = best; }
        if (scan32) { float cur = scan32[i]; float b32 = (float)best; if (b32 < cur) scan32[i] = b32; }
    }
}

} // namespace f110
