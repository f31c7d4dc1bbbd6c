/*
 * f110_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Scalar fp64 CPU restatement of the reference's F110Env.step hot path
 * (WE-Autopilot/red_gym, gym/f110_gym/envs/*.py).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the
 * shipped path (red_gym_amd/) never does and fails loudly without its HIP
 * library.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * gym/f110_gym/envs/).  The operation order of each floating-point expression
 * follows the Python source (left-to-right, Python precedence); build with
 * -ffp-contract=off so no FMA is formed.
 *
 * Parity pin: checked against golden vectors produced by importing the
 * reference itself in the dev container (tests/golden/make_golden.py) and
 * against the reference's own known-answer tests (dynamic_models.py:255-279,
 * collision_models.py:306-324, unittest/legacy_scan.npz).  Quantities that the
 * reference computes through BLAS (`ndarray.dot`: GJK dot products,
 * get_vertices, get_range, _check_done's rotation) are rounding-unspecified
 * there (numpy's BLAS in the dev container fuses them into FMAs with a shape-
 * dependent pattern); this file uses plain mul/add and the tests allow 1e-12
 * on those floats while booleans / indices stay exact.  sin/cos/tan/atan2 come
 * from libm here (numpy ships its own SIMD kernels; Numba would call libm).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_PI 3.141592653589793 /* np.pi */

/* ------------------------------------------------------------------------- */
/* map + scanner configuration (laser_models.py:348-427)                      */
/* ------------------------------------------------------------------------- */
typedef struct {
    int height, width;      /* map_height, map_width  (laser_models.py:406-407) */
    double resolution;      /* :413 */
    double orig_x, orig_y;  /* :419-420 */
    double orig_c, orig_s;  /* :421-422 */
    const double *dt;       /* [height*width] row-major, = resolution*edt (:425) */
} orc_map;

typedef struct {
    int num_beams;               /* :362 */
    int theta_dis;               /* :365 */
    double fov;                  /* :363 */
    double eps;                  /* :364 */
    double max_range;            /* :366 */
    double theta_index_increment;/* :368 */
    const double *sines;         /* [theta_dis] :380 */
    const double *cosines;       /* [theta_dis] :381 */
} orc_scan_cfg;

/* laser_models.py:56-86  xy_2_rc */
static void orc_xy_2_rc(double x, double y, const orc_map *m, int *r, int *c)
{
    double x_trans = x - m->orig_x;
    double y_trans = y - m->orig_y;
    double x_rot = x_trans * m->orig_c + y_trans * m->orig_s;
    double y_rot = -x_trans * m->orig_s + y_trans * m->orig_c;
    if (x_rot < 0 || x_rot >= m->width * m->resolution ||
        y_rot < 0 || y_rot >= m->height * m->resolution) {
        *c = -1;
        *r = -1;
    } else {
        *c = (int)(x_rot / m->resolution);
        *r = (int)(y_rot / m->resolution);
    }
}

/* laser_models.py:89-104  distance_transform; dt[r, c] with Python negative
 * indexing: (-1,-1) reads dt[height-1, width-1]. */
static double orc_distance_transform(double x, double y, const orc_map *m)
{
    int r, c;
    orc_xy_2_rc(x, y, m, &r, &c);
    if (r < 0) r += m->height;
    if (c < 0) c += m->width;
    return m->dt[(size_t)r * (size_t)m->width + (size_t)c];
}

/* laser_models.py:107-146  trace_ray.  *lookups is incremented once per
 * distance-table read (instrumentation for SURVEY 8(d)'s byte model). */
static double orc_trace_ray(double x, double y, double theta_index,
                            const orc_scan_cfg *s, const orc_map *m,
                            int64_t *lookups)
{
    int theta_index_ = (int)theta_index;
    double sn = s->sines[theta_index_];
    double cs = s->cosines[theta_index_];
    double dist_to_nearest = orc_distance_transform(x, y, m);
    double total_dist = dist_to_nearest;
    int64_t n = 1;
    while (dist_to_nearest > s->eps && total_dist <= s->max_range) {
        x += dist_to_nearest * cs;
        y += dist_to_nearest * sn;
        dist_to_nearest = orc_distance_transform(x, y, m);
        total_dist += dist_to_nearest;
        n++;
    }
    if (total_dist > s->max_range)
        total_dist = s->max_range;
    if (lookups) *lookups += n;
    return total_dist;
}

/* laser_models.py:149-186  get_scan.  Optionally records the truncated LUT
 * index of every beam (beam_idx, may be NULL). */
void orc_get_scan(const double pose[3], const orc_scan_cfg *s, const orc_map *m,
                  double *scan, int32_t *beam_idx, int64_t *lookups)
{
    double theta_dis = (double)s->theta_dis;
    double theta_index = theta_dis * (pose[2] - s->fov / 2.) / (2. * ORC_PI);
    theta_index = fmod(theta_index, theta_dis);
    while (theta_index < 0)
        theta_index += theta_dis;
    for (int i = 0; i < s->num_beams; i++) {
        if (beam_idx) beam_idx[i] = (int32_t)theta_index;
        scan[i] = orc_trace_ray(pose[0], pose[1], theta_index, s, m, lookups);
        theta_index += s->theta_index_increment;
        while (theta_index >= theta_dis)
            theta_index -= theta_dis;
    }
}

/* batch helper for tests / cpu baseline: n poses -> [n, num_beams] */
void orc_scan_batch(const double *poses, int n, const orc_scan_cfg *s,
                    const orc_map *m, double *scans, int64_t *lookups_per_pose)
{
    for (int k = 0; k < n; k++) {
        int64_t l = 0;
        orc_get_scan(poses + 3 * k, s, m, scans + (size_t)k * s->num_beams, NULL, &l);
        if (lookups_per_pose) lookups_per_pose[k] = l;
    }
}

/* laser_models.py:189-217  check_ttc_jit (error_model='numpy': division by
 * zero yields inf/nan, no exception -- identical to C doubles). */
int orc_check_ttc(const double *scan, double vel, const double *cosines,
                  const double *side_distances, double ttc_thresh, int num_beams)
{
    int in_collision = 0;
    if (vel != 0.0) {
        for (int i = 0; i < num_beams; i++) {
            double proj_vel = vel * cosines[i];
            double ttc = (scan[i] - side_distances[i]) / proj_vel;
            if ((ttc < ttc_thresh) && (ttc >= 0.0)) {
                in_collision = 1;
                break;
            }
        }
    }
    return in_collision;
}

/* laser_models.py:220-230 */
static double orc_cross(const double v1[2], const double v2[2])
{
    return v1[0] * v2[1] - v1[1] * v2[0];
}

/* laser_models.py:233-247 */
static int orc_are_collinear(const double a[2], const double b[2], const double c[2])
{
    double tol = 1e-8;
    double ba[2] = {b[0] - a[0], b[1] - a[1]};
    double ca[2] = {a[0] - c[0], a[1] - c[1]};
    return fabs(orc_cross(ba, ca)) < tol;
}

/* laser_models.py:250-280  get_range */
double orc_get_range(const double pose[3], double beam_theta,
                     const double va[2], const double vb[2])
{
    double o[2] = {pose[0], pose[1]};
    double v1[2] = {o[0] - va[0], o[1] - va[1]};
    double v2[2] = {vb[0] - va[0], vb[1] - va[1]};
    double v3[2] = {cos(beam_theta + ORC_PI / 2.), sin(beam_theta + ORC_PI / 2.)};
    double denom = v2[0] * v3[0] + v2[1] * v3[1];
    double distance = INFINITY;
    if (fabs(denom) > 0.0) {
        double d1 = orc_cross(v2, v1) / denom;
        double d2 = (v1[0] * v3[0] + v1[1] * v3[1]) / denom;
        if (d1 >= 0.0 && d2 >= 0.0 && d2 <= 1.0)
            distance = d1;
    } else if (orc_are_collinear(o, va, vb)) {
        double ea[2] = {va[0] - o[0], va[1] - o[1]};
        double eb[2] = {vb[0] - o[0], vb[1] - o[1]};
        double da = sqrt(ea[0] * ea[0] + ea[1] * ea[1]);
        double db = sqrt(eb[0] * eb[0] + eb[1] * eb[1]);
        distance = da < db ? da : db; /* min(da, db) */
    }
    return distance;
}

/* first index of the minimum, as np.argmin */
static int orc_argmin_abs_diff(const double *scan_angles, int n, double a)
{
    int best = 0;
    double bv = fabs(scan_angles[0] - a);
    for (int i = 1; i < n; i++) {
        double v = fabs(scan_angles[i] - a);
        if (v < bv) { bv = v; best = i; }
    }
    return best;
}

/* laser_models.py:283-315  get_blocked_view_indices */
void orc_get_blocked_view_indices(const double pose[3], const double vertices[4][2],
                                  const double *scan_angles, int num_beams,
                                  int *min_ind, int *max_ind)
{
    double ex = cos(pose[2]), ey = sin(pose[2]);
    int lo = 0, hi = 0;
    for (int i = 0; i < 4; i++) {
        double vx = vertices[i][0] - pose[0];
        double vy = vertices[i][1] - pose[1];
        double norm = sqrt(vx * vx + vy * vy);
        double ux = vx / norm, uy = vy / norm;
        double angle = atan2(ey, ex) - atan2(uy, ux);
        if (angle > ORC_PI)
            angle = angle - 2 * ORC_PI;
        else if (angle < -ORC_PI)
            angle = angle + 2 * ORC_PI;
        int ind = orc_argmin_abs_diff(scan_angles, num_beams, -angle);
        if (i == 0) { lo = hi = ind; }
        else { if (ind < lo) lo = ind; if (ind > hi) hi = ind; }
    }
    *min_ind = lo;
    *max_ind = hi;
}

/* laser_models.py:319-346  ray_cast (modifies scan in place) */
void orc_ray_cast(const double pose[3], double *scan, const double *scan_angles,
                  int num_beams, const double vertices[4][2])
{
    double looped[5][2];
    memcpy(looped, vertices, sizeof(double) * 8);
    looped[4][0] = vertices[0][0];
    looped[4][1] = vertices[0][1];
    int min_ind, max_ind;
    orc_get_blocked_view_indices(pose, vertices, scan_angles, num_beams, &min_ind, &max_ind);
    for (int i = min_ind; i <= max_ind; i++) {
        for (int j = 0; j < 4; j++) {
            double scan_range = orc_get_range(pose, pose[2] + scan_angles[i], looped[j], looped[j + 1]);
            if (scan_range < scan[i])
                scan[i] = scan_range;
        }
    }
}

/* ------------------------------------------------------------------------- */
/* vehicle model (dynamic_models.py)                                          */
/* params order: mu C_Sf C_Sr lf lr h m I s_min s_max sv_min sv_max v_switch  */
/*               a_max v_min v_max width length   (f110_env.py:128)           */
/* ------------------------------------------------------------------------- */
enum { P_MU, P_CSF, P_CSR, P_LF, P_LR, P_H, P_M, P_I, P_SMIN, P_SMAX, P_SVMIN,
       P_SVMAX, P_VSWITCH, P_AMAX, P_VMIN, P_VMAX, P_WIDTH, P_LENGTH, P_COUNT };

/* dynamic_models.py:30-60 */
static double orc_accl_constraints(double vel, double accl, double v_switch,
                                   double a_max, double v_min, double v_max)
{
    double pos_limit;
    if (vel > v_switch)
        pos_limit = a_max * v_switch / vel;
    else
        pos_limit = a_max;
    if ((vel <= v_min && accl <= 0) || (vel >= v_max && accl >= 0))
        accl = 0.;
    else if (accl <= -a_max)
        accl = -a_max;
    else if (accl >= pos_limit)
        accl = pos_limit;
    return accl;
}

/* dynamic_models.py:63-87 */
static double orc_steering_constraint(double steering_angle, double steering_velocity,
                                      double s_min, double s_max, double sv_min, double sv_max)
{
    if ((steering_angle <= s_min && steering_velocity <= 0) ||
        (steering_angle >= s_max && steering_velocity >= 0))
        steering_velocity = 0.;
    else if (steering_velocity <= sv_min)
        steering_velocity = sv_min;
    else if (steering_velocity >= sv_max)
        steering_velocity = sv_max;
    return steering_velocity;
}

/* dynamic_models.py:91-121  vehicle_dynamics_ks (5 states) */
void orc_vehicle_dynamics_ks(const double x[5], const double u_init[2],
                             const double *p, double f[5])
{
    double lwb = p[P_LF] + p[P_LR];
    double u0 = orc_steering_constraint(x[2], u_init[0], p[P_SMIN], p[P_SMAX], p[P_SVMIN], p[P_SVMAX]);
    double u1 = orc_accl_constraints(x[3], u_init[1], p[P_VSWITCH], p[P_AMAX], p[P_VMIN], p[P_VMAX]);
    f[0] = x[3] * cos(x[4]);
    f[1] = x[3] * sin(x[4]);
    f[2] = u0;
    f[3] = u1;
    f[4] = x[3] / lwb * tan(x[2]);
}

/* dynamic_models.py:124-176  vehicle_dynamics_st (7 states) */
void orc_vehicle_dynamics_st(const double x[7], const double u_init[2],
                             const double *p, double f[7])
{
    const double g = 9.81;
    double mu = p[P_MU], C_Sf = p[P_CSF], C_Sr = p[P_CSR], lf = p[P_LF], lr = p[P_LR];
    double h = p[P_H], m = p[P_M], I = p[P_I];
    double u[2];
    u[0] = orc_steering_constraint(x[2], u_init[0], p[P_SMIN], p[P_SMAX], p[P_SVMIN], p[P_SVMAX]);
    u[1] = orc_accl_constraints(x[3], u_init[1], p[P_VSWITCH], p[P_AMAX], p[P_VMIN], p[P_VMAX]);

    if (fabs(x[3]) < 0.5) {
        double lwb = lf + lr;
        double f_ks[5];
        orc_vehicle_dynamics_ks(x, u, p, f_ks); /* constraints re-applied (:113) */
        double cx2 = cos(x[2]);
        f[0] = f_ks[0]; f[1] = f_ks[1]; f[2] = f_ks[2]; f[3] = f_ks[3]; f[4] = f_ks[4];
        f[5] = u[1] / lwb * tan(x[2]) + x[3] / (lwb * (cx2 * cx2)) * u[0];
        f[6] = 0;
    } else {
        double glr_m = g * lr - u[1] * h; /* (g*lr - u[1]*h) */
        double glf_p = g * lf + u[1] * h; /* (g*lf + u[1]*h) */
        f[0] = x[3] * cos(x[6] + x[4]);
        f[1] = x[3] * sin(x[6] + x[4]);
        f[2] = u[0];
        f[3] = u[1];
        f[4] = x[5];
        f[5] = -mu * m / (x[3] * I * (lr + lf)) * ((lf * lf) * C_Sf * glr_m + (lr * lr) * C_Sr * glf_p) * x[5]
             + mu * m / (I * (lr + lf)) * (lr * C_Sr * glf_p - lf * C_Sf * glr_m) * x[6]
             + mu * m / (I * (lr + lf)) * lf * C_Sf * glr_m * x[2];
        f[6] = (mu / ((x[3] * x[3]) * (lr + lf)) * (C_Sr * glf_p * lr - C_Sf * glr_m * lf) - 1) * x[5]
             - mu / (x[3] * (lr + lf)) * (C_Sr * glf_p + C_Sf * glr_m) * x[6]
             + mu / (x[3] * (lr + lf)) * (C_Sf * glr_m) * x[2];
    }
}

/* dynamic_models.py:179-221  pid */
void orc_pid(double speed, double steer, double current_speed, double current_steer,
             double max_sv, double max_a, double max_v, double min_v,
             double *accl_out, double *sv_out)
{
    double sv, accl, kp;
    double steer_diff = steer - current_steer;
    if (fabs(steer_diff) > 1e-4)
        sv = (steer_diff / fabs(steer_diff)) * max_sv;
    else
        sv = 0.0;
    double vel_diff = speed - current_speed;
    if (current_speed > 0.) {
        if (vel_diff > 0) {
            kp = 10.0 * max_a / max_v;
            accl = kp * vel_diff;
        } else {
            kp = 10.0 * max_a / (-min_v);
            accl = kp * vel_diff;
        }
    } else {
        if (vel_diff > 0) {
            kp = 2.0 * max_a / max_v;
            accl = kp * vel_diff;
        } else {
            kp = 2.0 * max_a / (-min_v);
            accl = kp * vel_diff;
        }
    }
    *accl_out = accl;
    *sv_out = sv;
}

/* ------------------------------------------------------------------------- */
/* RaceCar (base_classes.py:44-443)                                           */
/* ------------------------------------------------------------------------- */
typedef struct {
    double state[7];        /* [x, y, steer, v, yaw, yaw_rate, slip] :95-96 */
    double steer_buffer[2]; /* [0] = newest (np.append(raw, buf)), :106,:272 */
    int steer_count;
    int in_collision;
    int64_t noise_step;     /* draws taken from scan_rng since reset (:202,:405) */
} orc_car;

enum { ORC_RK4 = 1, ORC_EULER = 2 }; /* base_classes.py:40-42 */

/* base_classes.py:181-202 */
void orc_car_reset(orc_car *car, const double pose[3])
{
    memset(car, 0, sizeof(*car));
    car->state[0] = pose[0];
    car->state[1] = pose[1];
    car->state[4] = pose[2];
}

/* base_classes.py:254-402 (everything of update_pose before the scan) */
void orc_car_update_pose(orc_car *car, double raw_steer, double vel,
                         const double *p, double time_step, int integrator)
{
    double steer = 0.;
    if (car->steer_count < 2) {
        steer = 0.;
        car->steer_buffer[1] = car->steer_buffer[0];
        car->steer_buffer[0] = raw_steer;
        car->steer_count++;
    } else {
        steer = car->steer_buffer[1];
        car->steer_buffer[1] = car->steer_buffer[0];
        car->steer_buffer[0] = raw_steer;
    }

    double accl, sv;
    orc_pid(vel, steer, car->state[3], car->state[2], p[P_SVMAX], p[P_AMAX], p[P_VMAX], p[P_VMIN], &accl, &sv);
    double u[2] = {sv, accl};
    double *st = car->state;

    if (integrator == ORC_RK4) {
        double k1[7], k2[7], k3[7], k4[7], tmp[7];
        orc_vehicle_dynamics_st(st, u, p, k1);
        for (int i = 0; i < 7; i++) tmp[i] = st[i] + time_step * (k1[i] / 2);
        orc_vehicle_dynamics_st(tmp, u, p, k2);
        for (int i = 0; i < 7; i++) tmp[i] = st[i] + time_step * (k2[i] / 2);
        orc_vehicle_dynamics_st(tmp, u, p, k3);
        for (int i = 0; i < 7; i++) tmp[i] = st[i] + time_step * k3[i];
        orc_vehicle_dynamics_st(tmp, u, p, k4);
        double w = time_step * (1. / 6); /* self.time_step*(1/6) evaluated first (:371) */
        for (int i = 0; i < 7; i++)
            st[i] = st[i] + w * (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]);
    } else {
        double f[7];
        orc_vehicle_dynamics_st(st, u, p, f);
        for (int i = 0; i < 7; i++) st[i] = st[i] + time_step * f[i];
    }

    if (st[4] > 2 * ORC_PI)
        st[4] = st[4] - 2 * ORC_PI;
    else if (st[4] < 0)
        st[4] = st[4] + 2 * ORC_PI;
}

/* batch helper: n independent cars, one update_pose each (tests) */
void orc_update_pose_batch(orc_car *cars, int n, const double *actions /*[n,2]*/,
                           const double *p, double time_step, int integrator)
{
    for (int i = 0; i < n; i++)
        orc_car_update_pose(&cars[i], actions[2 * i], actions[2 * i + 1], p, time_step, integrator);
}

/* base_classes.py:123-156  per-beam tables */
void orc_beam_tables(int num_beams, double fov, double width, double lf, double lr,
                     double *scan_angles, double *cosines, double *side_distances)
{
    double scan_ang_incr = fov / (num_beams - 1); /* laser_models.py:367 */
    double dist_sides = width / 2.;
    double dist_fr = (lf + lr) / 2.;
    for (int i = 0; i < num_beams; i++) {
        double angle = -fov / 2. + i * scan_ang_incr;
        double to_side, to_fr;
        scan_angles[i] = angle;
        cosines[i] = cos(angle);
        if (angle > 0) {
            if (angle < ORC_PI / 2) {
                to_side = dist_sides / sin(angle);
                to_fr = dist_fr / cos(angle);
            } else {
                to_side = dist_sides / cos(angle - ORC_PI / 2.);
                to_fr = dist_fr / sin(angle - ORC_PI / 2.);
            }
        } else {
            if (angle > -ORC_PI / 2) {
                to_side = dist_sides / sin(-angle);
                to_fr = dist_fr / cos(-angle);
            } else {
                to_side = dist_sides / cos(-angle - ORC_PI / 2);
                to_fr = dist_fr / sin(-angle - ORC_PI / 2);
            }
        }
        side_distances[i] = to_side < to_fr ? to_side : to_fr; /* min(to_side, to_fr) */
    }
}

/* ------------------------------------------------------------------------- */
/* collision_models.py                                                         */
/* ------------------------------------------------------------------------- */
/* collision_models.py:219-260  get_vertices: rows of H = [[c,-s,0,x],[s,c,0,y],..]
 * times [+-L/2, +-W/2, 0, 1]; order rl, rr, fr, fl (:259). */
void orc_get_vertices(const double pose[3], double length, double width, double out[4][2])
{
    double x = pose[0], y = pose[1], th = pose[2];
    double c = cos(th), s = sin(th);
    double hx[4] = {-length / 2, -length / 2, length / 2, length / 2};
    double hy[4] = {width / 2, -width / 2, -width / 2, width / 2};
    for (int k = 0; k < 4; k++) {
        double px = ((c * hx[k] + (-s) * hy[k]) + 0. * 0.) + x * 1.;
        double py = ((s * hx[k] + c * hy[k]) + 0. * 0.) + y * 1.;
        double pw = 1.;
        out[k][0] = px / pw;
        out[k][1] = py / pw;
    }
}

/* :82-92 first arg-max of vertices . d */
static int orc_furthest(const double (*v)[2], int n, double dx, double dy)
{
    int best = 0;
    double bv = v[0][0] * dx + v[0][1] * dy;
    for (int i = 1; i < n; i++) {
        double t = v[i][0] * dx + v[i][1] * dy;
        if (t > bv) { bv = t; best = i; }
    }
    return best;
}

/* :96-110 */
static void orc_support(const double (*v1)[2], int n1, const double (*v2)[2], int n2,
                        const double d[2], double out[2])
{
    int i = orc_furthest(v1, n1, d[0], d[1]);
    int j = orc_furthest(v2, n2, -d[0], -d[1]);
    out[0] = v1[i][0] - v2[j][0];
    out[1] = v1[i][1] - v2[j][1];
}

/* :52-64  tripleProduct(a,b,c) = b*(a.c) - a*(b.c) */
static void orc_triple(const double a[2], const double b[2], const double c[2], double out[2])
{
    double ac = a[0] * c[0] + a[1] * c[1];
    double bc = b[0] * c[0] + b[1] * c[1];
    out[0] = b[0] * ac - a[0] * bc;
    out[1] = b[1] * ac - a[1] * bc;
}

/* :114-182  GJK */
int orc_collision(const double (*v1)[2], int n1, const double (*v2)[2], int n2)
{
    int index = 0;
    double simplex[3][2];
    double p1[2] = {0, 0}, p2[2] = {0, 0};
    /* avgPoint :68-78: np.sum(axis=0)/n (sequential row adds) */
    for (int i = 0; i < n1; i++) { p1[0] += v1[i][0]; p1[1] += v1[i][1]; }
    for (int i = 0; i < n2; i++) { p2[0] += v2[i][0]; p2[1] += v2[i][1]; }
    p1[0] /= n1; p1[1] /= n1; p2[0] /= n2; p2[1] /= n2;
    double d[2] = {p1[0] - p2[0], p1[1] - p2[1]};
    if (d[0] == 0 && d[1] == 0)
        d[0] = 1.0;
    double a[2];
    orc_support(v1, n1, v2, n2, d, a);
    simplex[index][0] = a[0]; simplex[index][1] = a[1];
    if (d[0] * a[0] + d[1] * a[1] <= 0)
        return 0;
    d[0] = -a[0]; d[1] = -a[1];
    int iter_count = 0;
    while (iter_count < 1000) {
        orc_support(v1, n1, v2, n2, d, a);
        index += 1;
        simplex[index][0] = a[0]; simplex[index][1] = a[1];
        if (d[0] * a[0] + d[1] * a[1] <= 0)
            return 0;
        double ao[2] = {-a[0], -a[1]};
        if (index < 2) {
            double ab[2] = {simplex[0][0] - a[0], simplex[0][1] - a[1]};
            orc_triple(ab, ao, ab, d);
            if (sqrt(d[0] * d[0] + d[1] * d[1]) < 1e-10) {
                /* perpendicular(ab) :35-48 */
                d[0] = ab[1];
                d[1] = -1 * ab[0];
            }
            continue;
        }
        double ab[2] = {simplex[1][0] - a[0], simplex[1][1] - a[1]};
        double ac[2] = {simplex[0][0] - a[0], simplex[0][1] - a[1]};
        double acperp[2];
        orc_triple(ab, ac, ac, acperp);
        if (acperp[0] * ao[0] + acperp[1] * ao[1] >= 0) {
            d[0] = acperp[0]; d[1] = acperp[1];
        } else {
            double abperp[2];
            orc_triple(ac, ab, ab, abperp);
            if (abperp[0] * ao[0] + abperp[1] * ao[1] < 0)
                return 1;
            simplex[0][0] = simplex[1][0]; simplex[0][1] = simplex[1][1];
            d[0] = abperp[0]; d[1] = abperp[1];
        }
        simplex[1][0] = simplex[2][0]; simplex[1][1] = simplex[2][1];
        index -= 1;
        iter_count += 1;
    }
    return 0;
}

/* :185-212  collision_multiple on quads */
void orc_collision_multiple(const double *vertices /*[n,4,2]*/, int n,
                            double *collisions, double *collision_idx)
{
    for (int i = 0; i < n; i++) { collisions[i] = 0.; collision_idx[i] = -1.; }
    for (int i = 0; i < n - 1; i++) {
        for (int j = i + 1; j < n; j++) {
            const double (*vi)[2] = (const double (*)[2])(vertices + 8 * i);
            const double (*vj)[2] = (const double (*)[2])(vertices + 8 * j);
            if (orc_collision(vi, 4, vj, 4)) {
                collisions[i] = 1.;
                collisions[j] = 1.;
                collision_idx[i] = j;
                collision_idx[j] = i;
            }
        }
    }
}

/* generic pair for the reference's KATs (collision_models.py:306-324) */
int orc_collision_quads(const double *a, const double *b)
{
    return orc_collision((const double (*)[2])a, 4, (const double (*)[2])b, 4);
}

/* ------------------------------------------------------------------------- */
/* Simulator (base_classes.py:445-623) + F110Env lap logic (f110_env.py)       */
/* ------------------------------------------------------------------------- */
#define ORC_MAX_AGENTS 32

typedef struct {
    int num_agents, ego_idx, integrator;
    double time_step;
    double params[P_COUNT];
    double ttc_thresh; /* base_classes.py:113 */
    orc_scan_cfg scan;
    orc_map map;
    const double *scan_angles, *cosines, *side_distances; /* [num_beams] */
    const double *noise;  /* [noise_steps, num_beams] or NULL = no noise */
    int64_t noise_steps;
    orc_car cars[ORC_MAX_AGENTS];
    double collisions[ORC_MAX_AGENTS];
    double collision_idx[ORC_MAX_AGENTS];
    /* F110Env members (f110_env.py:160-187) */
    double start_xs[ORC_MAX_AGENTS], start_ys[ORC_MAX_AGENTS], start_thetas[ORC_MAX_AGENTS];
    double start_rot[2][2];
    int near_starts[ORC_MAX_AGENTS];
    double toggle_list[ORC_MAX_AGENTS];
    double lap_times[ORC_MAX_AGENTS], lap_counts[ORC_MAX_AGENTS];
    double current_time;
    int64_t lookups; /* instrumentation: distance-table reads so far */
} orc_env;

size_t orc_env_sizeof(void) { return sizeof(orc_env); }

void orc_env_init(orc_env *e, int num_agents, int ego_idx, int integrator, double time_step,
                  const double *params18, const orc_scan_cfg *scan, const orc_map *map,
                  const double *scan_angles, const double *cosines, const double *side_distances,
                  const double *noise, int64_t noise_steps)
{
    memset(e, 0, sizeof(*e));
    e->num_agents = num_agents;
    e->ego_idx = ego_idx;
    e->integrator = integrator;
    e->time_step = time_step;
    memcpy(e->params, params18, sizeof(double) * P_COUNT);
    e->ttc_thresh = 0.005;
    e->scan = *scan;
    e->map = *map;
    e->scan_angles = scan_angles;
    e->cosines = cosines;
    e->side_distances = side_distances;
    e->noise = noise;
    e->noise_steps = noise_steps;
    for (int i = 0; i < num_agents; i++) { e->collision_idx[i] = -1.; e->near_starts[i] = 1; }
    e->start_rot[0][0] = 1; e->start_rot[1][1] = 1;
}

/* Simulator.step (base_classes.py:546-605).  scans: [num_agents, num_beams] out. */
void orc_sim_step(orc_env *e, const double *control_inputs /*[A,2]*/, double *scans)
{
    int A = e->num_agents, nb = e->scan.num_beams;
    double agent_poses[ORC_MAX_AGENTS][3];
    double verts[ORC_MAX_AGENTS][4][2];
    /* :561-567 update_pose + scan (+ noise, laser_models.py:450-452) */
    for (int i = 0; i < A; i++) {
        orc_car *car = &e->cars[i];
        orc_car_update_pose(car, control_inputs[2 * i], control_inputs[2 * i + 1],
                            e->params, e->time_step, e->integrator);
        double pose[3] = {car->state[0], car->state[1], car->state[4]};
        double *scan = scans + (size_t)i * nb;
        orc_get_scan(pose, &e->scan, &e->map, scan, NULL, &e->lookups);
        if (e->noise) {
            const double *nz = e->noise + (size_t)(car->noise_step % e->noise_steps) * nb;
            for (int k = 0; k < nb; k++) scan[k] += nz[k];
        }
        car->noise_step++;
        agent_poses[i][0] = pose[0]; agent_poses[i][1] = pose[1]; agent_poses[i][2] = pose[2];
    }
    /* :570 check_collision (:529-543) */
    for (int i = 0; i < A; i++)
        orc_get_vertices(agent_poses[i], e->params[P_LENGTH], e->params[P_WIDTH], verts[i]);
    orc_collision_multiple(&verts[0][0][0], A, e->collisions, e->collision_idx);
    /* :572-582 */
    for (int i = 0; i < A; i++) {
        orc_car *car = &e->cars[i];
        double *scan = scans + (size_t)i * nb;
        /* check_ttc :227-252 */
        int hit = orc_check_ttc(scan, car->state[3], e->cosines, e->side_distances, e->ttc_thresh, nb);
        if (hit) {
            for (int k = 3; k < 7; k++) car->state[k] = 0.;
        }
        car->in_collision = hit;
        /* ray_cast_agents :204-225: own *current* pose, opponents' snapshot poses */
        double pose[3] = {car->state[0], car->state[1], car->state[4]};
        for (int j = 0; j < A; j++) {
            if (j == i) continue;
            orc_ray_cast(pose, scan, e->scan_angles, nb, verts[j]);
        }
        if (car->in_collision)
            e->collisions[i] = 1.;
    }
}

/* F110Env._check_done (f110_env.py:202-244). Returns done. */
int orc_env_check_done(orc_env *e)
{
    int A = e->num_agents;
    double left_t = 2, right_t = 2;
    int all_done = 1;
    for (int i = 0; i < A; i++) {
        double px = e->cars[i].state[0] - e->start_xs[i];
        double py = e->cars[i].state[1] - e->start_ys[i];
        double dx = e->start_rot[0][0] * px + e->start_rot[0][1] * py;
        double temp_y = e->start_rot[1][0] * px + e->start_rot[1][1] * py;
        if (temp_y > left_t)
            temp_y -= left_t;
        else if (temp_y < -right_t)
            temp_y = -right_t - temp_y;
        else
            temp_y = 0;
        double dist2 = dx * dx + temp_y * temp_y;
        int closes = dist2 <= 0.1;
        if (closes && !e->near_starts[i]) {
            e->near_starts[i] = 1;
            e->toggle_list[i] += 1;
        } else if (!closes && e->near_starts[i]) {
            e->near_starts[i] = 0;
            e->toggle_list[i] += 1;
        }
        e->lap_counts[i] = floor(e->toggle_list[i] / 2); /* // 2 */
        if (e->toggle_list[i] < 4)
            e->lap_times[i] = e->current_time;
        if (!(e->toggle_list[i] >= 4)) all_done = 0;
    }
    return (e->collisions[e->ego_idx] != 0.) || all_done;
}

/* F110Env.step (f110_env.py:261-302) */
int orc_env_step(orc_env *e, const double *action, double *scans)
{
    orc_sim_step(e, action, scans);
    e->current_time = e->current_time + e->time_step;
    return orc_env_check_done(e);
}

/* F110Env.reset (f110_env.py:304-347) incl. the zero-action step */
int orc_env_reset(orc_env *e, const double *poses /*[A,3]*/, double *scans)
{
    int A = e->num_agents;
    e->current_time = 0.0;
    for (int i = 0; i < A; i++) {
        e->collisions[i] = 0.;
        e->near_starts[i] = 1;
        e->toggle_list[i] = 0.;
        e->start_xs[i] = poses[3 * i];
        e->start_ys[i] = poses[3 * i + 1];
        e->start_thetas[i] = poses[3 * i + 2];
    }
    double th = -e->start_thetas[e->ego_idx];
    e->start_rot[0][0] = cos(th); e->start_rot[0][1] = -sin(th);
    e->start_rot[1][0] = sin(th); e->start_rot[1][1] = cos(th);
    for (int i = 0; i < A; i++)
        orc_car_reset(&e->cars[i], poses + 3 * i); /* Simulator.reset :607-623 */
    double action[2 * ORC_MAX_AGENTS];
    memset(action, 0, sizeof(action));
    return orc_env_step(e, action, scans);
}

/* accessors for the ctypes wrapper */
void orc_env_get(const orc_env *e, double *states /*[A,7]*/, double *collisions, double *collision_idx,
                 double *lap_times, double *lap_counts, double *toggles, double *current_time,
                 int64_t *lookups)
{
    for (int i = 0; i < e->num_agents; i++) {
        memcpy(states + 7 * i, e->cars[i].state, sizeof(double) * 7);
        collisions[i] = e->collisions[i];
        collision_idx[i] = e->collision_idx[i];
        lap_times[i] = e->lap_times[i];
        lap_counts[i] = e->lap_counts[i];
        toggles[i] = e->toggle_list[i];
    }
    *current_time = e->current_time;
    if (lookups) *lookups = e->lookups;
}

void orc_env_set_state(orc_env *e, int agent, const double state[7], const double steer_buffer[2],
                       int steer_count, int64_t noise_step)
{
    memcpy(e->cars[agent].state, state, sizeof(double) * 7);
    e->cars[agent].steer_buffer[0] = steer_buffer[0];
    e->cars[agent].steer_buffer[1] = steer_buffer[1];
    e->cars[agent].steer_count = steer_count;
    e->cars[agent].noise_step = noise_step;
}

/* ------------------------------------------------------------------------- */
/* cpu_baseline leg of bench.py: B independent envs stepped with auto-reset,   */
/* optionally OpenMP over envs.  envs: contiguous array of orc_env.            */
/* ------------------------------------------------------------------------- */
void orc_batch_step(orc_env *envs, int B, const double *actions /*[B,A,2]*/,
                    const double *spawn /*[B,A,3]*/, uint8_t *pending_reset /*[B]*/,
                    double *scans /*[B,A,nb]*/, int threads)
{
    (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
#endif
    for (int b = 0; b < B; b++) {
        orc_env *e = &envs[b];
        int A = e->num_agents, nb = e->scan.num_beams;
        double *sc = scans + (size_t)b * A * nb;
        int done;
        if (pending_reset[b])
            done = orc_env_reset(e, spawn + (size_t)b * A * 3, sc);
        else
            done = orc_env_step(e, actions + (size_t)b * A * 2, sc);
        pending_reset[b] = (uint8_t)done;
    }
}
