/*
 * f110_oracle.c — CPU restatement of the f1tenth_gym env.step() hot path (plain C, float64).
 *
 * TEST INFRASTRUCTURE ONLY — see f110_oracle.h.  This file is the parity oracle for the
 * HIP kernels under f1tenth_gym_amd/csrc and the `cpu_baseline` ("port") of bench.py.
 * It is never linked into, imported by or called from the product path.
 *
 * Parity pinning: tests/test_oracle_golden.py checks every function here against
 *   (1) the reference's own known-answer vectors (dynamic_models.py:257-263,
 *       collision_models.py:313-324, unittest/legacy_scan.npz under its MSE<2 bar), and
 *   (2) golden vectors produced by importing the Python reference in the build container
 *       (oracle/refshim/gen_golden.py; fixtures in tests/golden/).
 *
 * References are gym/f110_gym/envs/<file>:<line> of f1tenth_gym v0.2.1.
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (oracle/Makefile).  The reference's numba
 * kernels are strict IEEE float64 without FMA contraction; so is this file.
 */
#include "f110_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ dynamic_models.py */

/* dynamic_models.py:29-60 */
double orc_accl_constraints(double vel, double accl, double v_switch, double a_max,
                            double v_min, double v_max)
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

/* dynamic_models.py:62-87 */
double orc_steering_constraint(double steering_angle, double steering_velocity, double s_min,
                               double s_max, double sv_min, double sv_max)
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

/* dynamic_models.py:90-121 */
void orc_vehicle_dynamics_ks(const double x[5], const double u_init[2], const double *p,
                             double f[5])
{
    double lwb = p[ORC_P_LF] + p[ORC_P_LR];
    double u0 = orc_steering_constraint(x[2], u_init[0], p[ORC_P_SMIN], p[ORC_P_SMAX],
                                        p[ORC_P_SVMIN], p[ORC_P_SVMAX]);
    double u1 = orc_accl_constraints(x[3], u_init[1], p[ORC_P_VSWITCH], p[ORC_P_AMAX],
                                     p[ORC_P_VMIN], p[ORC_P_VMAX]);
    f[0] = x[3] * cos(x[4]);
    f[1] = x[3] * sin(x[4]);
    f[2] = u0;
    f[3] = u1;
    f[4] = x[3] / lwb * tan(x[2]);
}

/* dynamic_models.py:123-176 */
void orc_vehicle_dynamics_st(const double x[7], const double u_init[2], const double *p,
                             double f[7])
{
    const double g = 9.81;
    double mu = p[ORC_P_MU], C_Sf = p[ORC_P_CSF], C_Sr = p[ORC_P_CSR];
    double lf = p[ORC_P_LF], lr = p[ORC_P_LR], h = p[ORC_P_H], m = p[ORC_P_M], I = p[ORC_P_I];
    double u[2];
    u[0] = orc_steering_constraint(x[2], u_init[0], p[ORC_P_SMIN], p[ORC_P_SMAX],
                                   p[ORC_P_SVMIN], p[ORC_P_SVMAX]);
    u[1] = orc_accl_constraints(x[3], u_init[1], p[ORC_P_VSWITCH], p[ORC_P_AMAX],
                                p[ORC_P_VMIN], p[ORC_P_VMAX]);

    if (fabs(x[3]) < 0.5) {
        /* :152-160 kinematic model for small velocities (constraints re-applied to the
         * already-constrained u inside vehicle_dynamics_ks, as the reference does) */
        double lwb = lf + lr;
        double f_ks[5];
        double cx2 = cos(x[2]);
        orc_vehicle_dynamics_ks(x, u, p, f_ks);
        f[0] = f_ks[0]; f[1] = f_ks[1]; f[2] = f_ks[2]; f[3] = f_ks[3]; f[4] = f_ks[4];
        f[5] = u[1] / lwb * tan(x[2]) + x[3] / (lwb * (cx2 * cx2)) * u[0];
        f[6] = 0.;
    } else {
        /* :164-174 */
        double glr_m = g * lr - u[1] * h; /* (g*lr - u[1]*h) */
        double glf_p = g * lf + u[1] * h; /* (g*lf + u[1]*h) */
        f[0] = x[3] * cos(x[6] + x[4]);
        f[1] = x[3] * sin(x[6] + x[4]);
        f[2] = u[0];
        f[3] = u[1];
        f[4] = x[5];
        f[5] = -mu * m / (x[3] * I * (lr + lf)) *
                   ((lf * lf) * C_Sf * glr_m + (lr * lr) * C_Sr * glf_p) * x[5]
               + mu * m / (I * (lr + lf)) * (lr * C_Sr * glf_p - lf * C_Sf * glr_m) * x[6]
               + mu * m / (I * (lr + lf)) * lf * C_Sf * glr_m * x[2];
        f[6] = (mu / ((x[3] * x[3]) * (lr + lf)) * (C_Sr * glf_p * lr - C_Sf * glr_m * lf) - 1) * x[5]
               - mu / (x[3] * (lr + lf)) * (C_Sr * glf_p + C_Sf * glr_m) * x[6]
               + mu / (x[3] * (lr + lf)) * (C_Sf * glr_m) * x[2];
    }
}

/* dynamic_models.py:178-221 */
void orc_pid(double speed, double steer, double current_speed, double current_steer,
             double max_sv, double max_a, double max_v, double min_v, double *accl, double *sv)
{
    double steer_diff = steer - current_steer;
    double kp, vel_diff;
    if (fabs(steer_diff) > 1e-4)
        *sv = (steer_diff / fabs(steer_diff)) * max_sv;
    else
        *sv = 0.0;

    vel_diff = speed - current_speed;
    if (current_speed > 0.) {
        if (vel_diff > 0)
            kp = 10.0 * max_a / max_v;
        else
            kp = 10.0 * max_a / (-min_v);
    } else {
        if (vel_diff > 0)
            kp = 2.0 * max_a / max_v;
        else
            kp = 2.0 * max_a / (-min_v);
    }
    *accl = kp * vel_diff;
}

/* ------------------------------------------------------------------ base_classes.py */

/* RaceCar.update_pose, base_classes.py:256-409 (everything before the scan call) */
void orc_update_pose(double state[7], double steer_buf[2], int *buf_count, double raw_steer,
                     double vel, const double *p, double time_step, int integrator,
                     double lidar_dist, double scan_pose[3])
{
    double steer = 0., accl, sv, u[2];
    int i;
    /* :271-278 steering delay buffer, newest first */
    if (*buf_count < 2) {
        steer = 0.;
        steer_buf[1] = steer_buf[0];
        steer_buf[0] = raw_steer;
        *buf_count += 1;
    } else {
        steer = steer_buf[1];
        steer_buf[1] = steer_buf[0];
        steer_buf[0] = raw_steer;
    }

    /* :282 */
    orc_pid(vel, steer, state[3], state[2], p[ORC_P_SVMAX], p[ORC_P_AMAX], p[ORC_P_VMAX],
            p[ORC_P_VMIN], &accl, &sv);
    u[0] = sv;
    u[1] = accl;

    if (integrator == ORC_INTEGRATOR_RK4) {
        /* :284-373 */
        double k1[7], k2[7], k3[7], k4[7], ks[7];
        double w = time_step * (1. / 6.);
        orc_vehicle_dynamics_st(state, u, p, k1);
        for (i = 0; i < 7; i++) ks[i] = state[i] + time_step * (k1[i] / 2);
        orc_vehicle_dynamics_st(ks, u, p, k2);
        for (i = 0; i < 7; i++) ks[i] = state[i] + time_step * (k2[i] / 2);
        orc_vehicle_dynamics_st(ks, u, p, k3);
        for (i = 0; i < 7; i++) ks[i] = state[i] + time_step * k3[i];
        orc_vehicle_dynamics_st(ks, u, p, k4);
        for (i = 0; i < 7; i++)
            state[i] = state[i] + w * (((k1[i] + 2 * k2[i]) + 2 * k3[i]) + k4[i]);
    } else {
        /* :375-395 */
        double f[7];
        orc_vehicle_dynamics_st(state, u, p, f);
        for (i = 0; i < 7; i++) state[i] = state[i] + time_step * f[i];
    }

    /* :400-404 bound yaw angle */
    if (state[4] > 2 * M_PI)
        state[4] = state[4] - 2 * M_PI;
    else if (state[4] < 0)
        state[4] = state[4] + 2 * M_PI;

    /* :407-409 */
    scan_pose[0] = state[0] + lidar_dist * cos(state[4]);
    scan_pose[1] = state[1] + lidar_dist * sin(state[4]);
    scan_pose[2] = state[4];
}

/* base_classes.py:125-158 */
void orc_build_beam_tables(int num_beams, double fov, double width, double lf, double lr,
                           double *scan_angles, double *cosines, double *side_distances)
{
    double scan_ang_incr = fov / (num_beams - 1); /* laser_models.py:367 */
    double dist_sides = width / 2.;
    double dist_fr = (lf + lr) / 2.;
    int i;
    for (i = 0; i < num_beams; i++) {
        double angle = -fov / 2. + i * scan_ang_incr;
        double to_side, to_fr;
        scan_angles[i] = angle;
        cosines[i] = cos(angle);
        if (angle > 0) {
            if (angle < M_PI / 2) {
                to_side = dist_sides / sin(angle);
                to_fr = dist_fr / cos(angle);
            } else {
                to_side = dist_sides / cos(angle - M_PI / 2.);
                to_fr = dist_fr / sin(angle - M_PI / 2.);
            }
        } else {
            if (angle > -M_PI / 2) {
                to_side = dist_sides / sin(-angle);
                to_fr = dist_fr / cos(-angle);
            } else {
                to_side = dist_sides / cos(-angle - M_PI / 2);
                to_fr = dist_fr / sin(-angle - M_PI / 2);
            }
        }
        side_distances[i] = to_side < to_fr ? to_side : to_fr; /* python min(a,b) */
    }
}

/* ------------------------------------------------------------------ laser_models.py */

/* laser_models.py:55-86 */
void orc_xy_2_rc(const orc_scan_cfg *c, double x, double y, int *r, int *col)
{
    double x_trans = x - c->orig_x;
    double y_trans = y - c->orig_y;
    double x_rot = x_trans * c->orig_c + y_trans * c->orig_s;
    double y_rot = -x_trans * c->orig_s + y_trans * c->orig_c;
    if (x_rot < 0 || x_rot >= c->width * c->resolution || y_rot < 0 ||
        y_rot >= c->height * c->resolution) {
        *col = -1;
        *r = -1;
    } else {
        *col = (int)(x_rot / c->resolution);
        *r = (int)(y_rot / c->resolution);
    }
}

/* laser_models.py:88-104; negative indices wrap like NumPy (dt[-1,-1]) */
static double orc_distance_transform(const orc_scan_cfg *c, double x, double y, int rc[2])
{
    int r, col;
    orc_xy_2_rc(c, x, y, &r, &col);
    rc[0] = r;
    rc[1] = col;
    if (r < 0) r += c->height;
    if (col < 0) col += c->width;
    return c->dt[(size_t)r * c->width + col];
}

/* laser_models.py:106-146 */
double orc_trace_ray(const orc_scan_cfg *c, double x, double y, double theta_index,
                     int hit_rc[2], int64_t *n_lookups)
{
    int theta_index_ = (int)theta_index;
    double s = c->sines[theta_index_];
    double co = c->cosines[theta_index_];
    int rc[2];
    int64_t n = 1;
    double dist_to_nearest = orc_distance_transform(c, x, y, rc);
    double total_dist = dist_to_nearest;

    while (dist_to_nearest > c->eps && total_dist <= c->max_range) {
        x += dist_to_nearest * co;
        y += dist_to_nearest * s;
        dist_to_nearest = orc_distance_transform(c, x, y, rc);
        total_dist += dist_to_nearest;
        n++;
    }
    if (total_dist > c->max_range) total_dist = c->max_range;
    if (hit_rc) {
        hit_rc[0] = rc[0];
        hit_rc[1] = rc[1];
    }
    if (n_lookups) *n_lookups += n;
    return total_dist;
}

/* laser_models.py:166-172 */
static double orc_theta_index_start(const orc_scan_cfg *c, double pose_theta)
{
    double theta_index = c->theta_dis * (pose_theta - c->fov / 2.) / (2. * M_PI);
    theta_index = fmod(theta_index, c->theta_dis);
    while (theta_index < 0) theta_index += c->theta_dis;
    return theta_index;
}

/* laser_models.py:148-186 */
void orc_get_scan(const orc_scan_cfg *c, const double pose[3], double *scan, int *hit_rc,
                  int64_t *n_lookups)
{
    double theta_index = orc_theta_index_start(c, pose[2]);
    int i;
    for (i = 0; i < c->num_beams; i++) {
        scan[i] = orc_trace_ray(c, pose[0], pose[1], theta_index, hit_rc ? hit_rc + 2 * i : 0,
                                n_lookups);
        theta_index += c->theta_index_increment;
        while (theta_index >= c->theta_dis) theta_index -= c->theta_dis;
    }
}

/* analysis aid (tools/debug/regroup_potential.py): table lookups of every beam of one scan */
void orc_get_scan_counts(const orc_scan_cfg *c, const double pose[3], int32_t *counts)
{
    double theta_index = orc_theta_index_start(c, pose[2]);
    int i;
    for (i = 0; i < c->num_beams; i++) {
        int64_t n = 0;
        (void)orc_trace_ray(c, pose[0], pose[1], theta_index, 0, &n);
        counts[i] = (int32_t)n;
        theta_index += c->theta_index_increment;
        while (theta_index >= c->theta_dis) theta_index -= c->theta_dis;
    }
}

void orc_beam_dir_indices(const orc_scan_cfg *c, double pose_theta, int *idx)
{
    double theta_index = orc_theta_index_start(c, pose_theta);
    int i;
    for (i = 0; i < c->num_beams; i++) {
        idx[i] = (int)theta_index;
        theta_index += c->theta_index_increment;
        while (theta_index >= c->theta_dis) theta_index -= c->theta_dis;
    }
}

/* laser_models.py:188-217 (error_model='numpy': x/0 -> inf/nan, no exception) */
int orc_check_ttc(const double *scan, int num_beams, double vel, const double *cosines,
                  const double *side_distances, double ttc_thresh)
{
    int i;
    if (vel != 0.0) {
        for (i = 0; i < num_beams; i++) {
            double proj_vel = vel * cosines[i];
            double ttc = (scan[i] - side_distances[i]) / proj_vel;
            if ((ttc < ttc_thresh) && (ttc >= 0.0)) return 1;
        }
    }
    return 0;
}

/* laser_models.py:219-230 */
static double orc_cross(const double v1[2], const double v2[2])
{
    return v1[0] * v2[1] - v1[1] * v2[0];
}

/* laser_models.py:232-247 */
static int orc_are_collinear(const double pt_a[2], const double pt_b[2], const double pt_c[2])
{
    const double tol = 1e-8;
    double ba[2] = {pt_b[0] - pt_a[0], pt_b[1] - pt_a[1]};
    double ca[2] = {pt_a[0] - pt_c[0], pt_a[1] - pt_c[1]};
    return fabs(orc_cross(ba, ca)) < tol;
}

/* laser_models.py:249-280 */
double orc_get_range(const double pose[3], double beam_theta, const double va[2],
                     const double vb[2])
{
    double o[2] = {pose[0], pose[1]};
    double v1[2] = {o[0] - va[0], o[1] - va[1]};
    double v2[2] = {vb[0] - va[0], vb[1] - va[1]};
    double v3[2] = {cos(beam_theta + M_PI / 2.), sin(beam_theta + M_PI / 2.)};
    double denom = v2[0] * v3[0] + v2[1] * v3[1];
    double distance = INFINITY;

    if (fabs(denom) > 0.0) {
        double d1 = orc_cross(v2, v1) / denom;
        double d2 = (v1[0] * v3[0] + v1[1] * v3[1]) / denom;
        if (d1 >= 0.0 && d2 >= 0.0 && d2 <= 1.0) distance = d1;
    } else if (orc_are_collinear(o, va, vb)) {
        double ax = va[0] - o[0], ay = va[1] - o[1];
        double bx = vb[0] - o[0], by = vb[1] - o[1];
        double da = sqrt(ax * ax + ay * ay);
        double db = sqrt(bx * bx + by * by);
        distance = da < db ? da : db; /* python min(da, db) */
    }
    return distance;
}

/* first index of the minimum of |scan_angles - a| (np.argmin, laser_models.py:310-313) */
static int orc_argmin_abs_diff(const double *scan_angles, int n, double a)
{
    int i, best = 0;
    double bv = fabs(scan_angles[0] - a);
    for (i = 1; i < n; i++) {
        double v = fabs(scan_angles[i] - a);
        if (v < bv) {
            bv = v;
            best = i;
        }
    }
    return best;
}

/* laser_models.py:282-315 */
void orc_get_blocked_view_indices(const double pose[3], const double vertices[8],
                                  const double *scan_angles, int num_beams, int *min_ind,
                                  int *max_ind)
{
    double ego_c = cos(pose[2]), ego_s = sin(pose[2]);
    int i, lo = 0, hi = 0;
    for (i = 0; i < 4; i++) {
        double vx = vertices[2 * i] - pose[0];
        double vy = vertices[2 * i + 1] - pose[1];
        double norm = sqrt(vx * vx + vy * vy);
        double ux = vx / norm, uy = vy / norm;
        double angle = atan2(ego_s, ego_c) - atan2(uy, ux);
        int ind;
        if (angle > M_PI)
            angle = angle - 2 * M_PI;
        else if (angle < -M_PI)
            angle = angle + 2 * M_PI;
        ind = orc_argmin_abs_diff(scan_angles, num_beams, -angle);
        if (i == 0) {
            lo = hi = ind;
        } else {
            if (ind < lo) lo = ind;
            if (ind > hi) hi = ind;
        }
    }
    *min_ind = lo;
    *max_ind = hi;
}

/* laser_models.py:318-346 */
void orc_ray_cast(const double pose[3], double *scan, const double *scan_angles,
                  int num_beams, const double vertices[8])
{
    double looped[10];
    int min_ind, max_ind, i, j;
    memcpy(looped, vertices, 8 * sizeof(double));
    looped[8] = vertices[0];
    looped[9] = vertices[1];
    orc_get_blocked_view_indices(pose, vertices, scan_angles, num_beams, &min_ind, &max_ind);
    for (i = min_ind; i < max_ind + 1; i++) {
        for (j = 0; j < 4; j++) {
            double scan_range =
                orc_get_range(pose, pose[2] + scan_angles[i], looped + 2 * j, looped + 2 * j + 2);
            if (scan_range < scan[i]) scan[i] = scan_range;
        }
    }
}

/* ---- exact EDT (laser_models.py:40-53 -> scipy.ndimage.distance_transform_edt) ----
 * scipy returns sqrt of the exact integer squared distance from every non-zero pixel to
 * the nearest zero pixel; restated here with Meijster, Roerdink & Hesselink's two-phase
 * integer algorithm (exact).  Images with no zero pixel at all are not meaningful maps. */
static int64_t orc_floordiv(int64_t a, int64_t b) /* b > 0 */
{
    int64_t q = a / b;
    if ((a % b != 0) && (a < 0)) q -= 1;
    return q;
}

void orc_edt_sq(const uint8_t *img, int height, int width, uint32_t *d2)
{
    const int64_t inf = (int64_t)height + width;
    int64_t *g = (int64_t *)malloc(sizeof(int64_t) * (size_t)height * width);
    int *s = (int *)malloc(sizeof(int) * width);
    int64_t *t = (int64_t *)malloc(sizeof(int64_t) * width);
    int x, y;
    /* phase 1: per column vertical distance */
    for (x = 0; x < width; x++) {
        g[x] = img[x] ? inf : 0;
        for (y = 1; y < height; y++)
            g[(size_t)y * width + x] = img[(size_t)y * width + x] ? 1 + g[(size_t)(y - 1) * width + x] : 0;
        for (y = height - 2; y >= 0; y--)
            if (g[(size_t)(y + 1) * width + x] < g[(size_t)y * width + x])
                g[(size_t)y * width + x] = 1 + g[(size_t)(y + 1) * width + x];
    }
    /* phase 2: per row lower envelope of parabolas */
#define ORC_F(xx, ii) (((int64_t)(xx) - (ii)) * ((int64_t)(xx) - (ii)) + grow[ii] * grow[ii])
    for (y = 0; y < height; y++) {
        const int64_t *grow = g + (size_t)y * width;
        int q = 0, u;
        s[0] = 0;
        t[0] = 0;
        for (u = 1; u < width; u++) {
            while (q >= 0 && ORC_F(t[q], s[q]) > ORC_F(t[q], u)) q--;
            if (q < 0) {
                q = 0;
                s[0] = u;
            } else {
                int64_t i = s[q];
                int64_t w = 1 + orc_floordiv((int64_t)u * u - i * i + grow[u] * grow[u] - grow[i] * grow[i],
                                             2 * ((int64_t)u - i));
                if (w < width) {
                    q++;
                    s[q] = u;
                    t[q] = w;
                }
            }
        }
        for (u = width - 1; u >= 0; u--) {
            int64_t v = ORC_F(u, s[q]);
            d2[(size_t)y * width + u] = (uint32_t)v;
            if (u == t[q]) q--;
        }
    }
#undef ORC_F
    free(g);
    free(s);
    free(t);
}

/* laser_models.py:398-404 (flip, threshold) + :425/:52 (resolution * edt) */
void orc_map_dt_from_image(const uint8_t *img_top_first, int height, int width,
                           double resolution, double *dt_out)
{
    uint8_t *bin = (uint8_t *)malloc((size_t)height * width);
    uint32_t *d2 = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)height * width);
    int r, c;
    for (r = 0; r < height; r++) {
        const uint8_t *src = img_top_first + (size_t)(height - 1 - r) * width; /* FLIP_TOP_BOTTOM */
        for (c = 0; c < width; c++) bin[(size_t)r * width + c] = src[c] > 128 ? 255 : 0;
    }
    orc_edt_sq(bin, height, width, d2);
    for (r = 0; r < height * width; r++) dt_out[r] = resolution * sqrt((double)d2[r]);
    free(bin);
    free(d2);
}

/* ------------------------------------------------------------------ collision_models.py */

/* collision_models.py:218-260; vertex order [rl, rr, fr, fl] (:259) */
void orc_get_vertices(const double pose[3], double length, double width, double vertices[8])
{
    double x = pose[0], y = pose[1], th = pose[2];
    double c = cos(th), s = sin(th);
    /* H.dot([bx, by, 0, 1]) rows 0,1: c*bx + (-s)*by + 0*0 + x*1 ; s*bx + c*by + 0*0 + y*1 */
    const double bx[4] = {-length / 2, -length / 2, length / 2, length / 2};
    const double by[4] = {width / 2, -width / 2, -width / 2, width / 2};
    int i;
    for (i = 0; i < 4; i++) {
        vertices[2 * i] = ((c * bx[i] + (-s) * by[i]) + 0. * 0.) + x * 1.;
        vertices[2 * i + 1] = ((s * bx[i] + c * by[i]) + 0. * 0.) + y * 1.;
    }
}

static double orc_dot2(const double a[2], const double b[2]) { return a[0] * b[0] + a[1] * b[1]; }

/* collision_models.py:51-64 : b*ac - a*bc */
static void orc_triple_product(const double a[2], const double b[2], const double c[2],
                               double out[2])
{
    double ac = orc_dot2(a, c);
    double bc = orc_dot2(b, c);
    out[0] = b[0] * ac - a[0] * bc;
    out[1] = b[1] * ac - a[1] * bc;
}

/* collision_models.py:81-92 np.argmax(vertices.dot(d)) — first maximum wins */
static int orc_index_of_furthest_point(const double v[8], const double d[2])
{
    int i, best = 0;
    double bv = v[0] * d[0] + v[1] * d[1];
    for (i = 1; i < 4; i++) {
        double val = v[2 * i] * d[0] + v[2 * i + 1] * d[1];
        if (val > bv) {
            bv = val;
            best = i;
        }
    }
    return best;
}

/* collision_models.py:95-110 */
static void orc_support(const double v1[8], const double v2[8], const double d[2], double out[2])
{
    double nd[2] = {-d[0], -d[1]};
    int i = orc_index_of_furthest_point(v1, d);
    int j = orc_index_of_furthest_point(v2, nd);
    out[0] = v1[2 * i] - v2[2 * j];
    out[1] = v1[2 * i + 1] - v2[2 * j + 1];
}

/* collision_models.py:113-182 (GJK) */
int orc_collision(const double v1[8], const double v2[8])
{
    int index = 0, iter_count = 0;
    double simplex[3][2];
    double p1[2], p2[2], d[2], a[2];

    /* avgPoint :67-78 : np.sum(axis=0)/4 */
    p1[0] = (((v1[0] + v1[2]) + v1[4]) + v1[6]) / 4;
    p1[1] = (((v1[1] + v1[3]) + v1[5]) + v1[7]) / 4;
    p2[0] = (((v2[0] + v2[2]) + v2[4]) + v2[6]) / 4;
    p2[1] = (((v2[1] + v2[3]) + v2[5]) + v2[7]) / 4;
    d[0] = p1[0] - p2[0];
    d[1] = p1[1] - p2[1];
    if (d[0] == 0 && d[1] == 0) d[0] = 1.0;

    orc_support(v1, v2, d, a);
    simplex[index][0] = a[0];
    simplex[index][1] = a[1];
    if (orc_dot2(d, a) <= 0) return 0;
    d[0] = -a[0];
    d[1] = -a[1];

    while (iter_count < 1e3) {
        double ao[2], ab[2], ac[2], acperp[2];
        orc_support(v1, v2, d, a);
        index += 1;
        simplex[index][0] = a[0];
        simplex[index][1] = a[1];
        if (orc_dot2(d, a) <= 0) return 0;
        ao[0] = -a[0];
        ao[1] = -a[1];

        if (index < 2) {
            ab[0] = simplex[0][0] - a[0];
            ab[1] = simplex[0][1] - a[1];
            orc_triple_product(ab, ao, ab, d);
            if (sqrt(d[0] * d[0] + d[1] * d[1]) < 1e-10) {
                /* perpendicular(ab) :34-48 */
                d[0] = ab[1];
                d[1] = -1 * ab[0];
            }
            continue;
        }
        ab[0] = simplex[1][0] - a[0];
        ab[1] = simplex[1][1] - a[1];
        ac[0] = simplex[0][0] - a[0];
        ac[1] = simplex[0][1] - a[1];
        orc_triple_product(ab, ac, ac, acperp);
        if (orc_dot2(acperp, ao) >= 0) {
            d[0] = acperp[0];
            d[1] = acperp[1];
        } else {
            double abperp[2];
            orc_triple_product(ac, ab, ab, abperp);
            if (orc_dot2(abperp, ao) < 0) return 1;
            simplex[0][0] = simplex[1][0];
            simplex[0][1] = simplex[1][1];
            d[0] = abperp[0];
            d[1] = abperp[1];
        }
        simplex[1][0] = simplex[2][0];
        simplex[1][1] = simplex[2][1];
        index -= 1;
        iter_count += 1;
    }
    return 0;
}

/* collision_models.py:184-212 */
void orc_collision_multiple(const double *vertices, int n, double *collisions,
                            double *collision_idx)
{
    int i, j;
    for (i = 0; i < n; i++) {
        collisions[i] = 0.;
        collision_idx[i] = -1.;
    }
    for (i = 0; i < n - 1; i++)
        for (j = i + 1; j < n; j++)
            if (orc_collision(vertices + 8 * i, vertices + 8 * j)) {
                collisions[i] = 1.;
                collisions[j] = 1.;
                collision_idx[i] = j;
                collision_idx[j] = i;
            }
}

/* ------------------------------------------------------------------ Simulator (batched) */

struct orc_sim {
    int E, A, N, B;
    double time_step, lidar_dist, ttc_thresh;
    int integrator;
    orc_scan_cfg scan;
    double *sines, *cosines_tab, *dt;
    double *params;     /* [A][18] per agent slot (RaceCar.params) */
    double sim_params[ORC_NPARAMS]; /* Simulator.params, used by check_collision :549 */
    double *scan_angles, *beam_cos, *side_distances; /* class-level tables :125-158 */
    double *noise;
    int noise_T;
    double *state, *steer_buf, *scans, *collisions, *collision_idx, *agent_poses;
    int32_t *buf_count, *in_collision, *step_count, *hit_rc;
    int64_t lookups;
};

orc_sim *orc_sim_create(int num_envs, int num_agents, int num_beams, double fov, double eps,
                        int theta_dis, double max_range, double time_step, int integrator,
                        double lidar_dist, double ttc_thresh, const double *params18)
{
    orc_sim *s = (orc_sim *)calloc(1, sizeof(orc_sim));
    int N = num_envs * num_agents, a, i;
    s->E = num_envs; s->A = num_agents; s->N = N; s->B = num_beams;
    s->time_step = time_step; s->integrator = integrator;
    s->lidar_dist = lidar_dist; s->ttc_thresh = ttc_thresh;
    s->scan.num_beams = num_beams; s->scan.fov = fov; s->scan.eps = eps;
    s->scan.theta_dis = theta_dis; s->scan.max_range = max_range;
    s->scan.angle_increment = fov / (num_beams - 1);                                   /* :367 */
    s->scan.theta_index_increment = theta_dis * s->scan.angle_increment / (2. * M_PI); /* :368 */
    s->sines = (double *)malloc(sizeof(double) * theta_dis);
    s->cosines_tab = (double *)malloc(sizeof(double) * theta_dis);
    /* laser_models.py:379-381: np.linspace(0, 2*pi, num=theta_dis) — endpoint inclusive.
     * Default tables use libm; tests overwrite them with the NumPy-computed ones. */
    for (i = 0; i < theta_dis; i++) {
        double th = (theta_dis > 1) ? i * ((2 * M_PI) / (theta_dis - 1)) : 0.0;
        s->sines[i] = sin(th);
        s->cosines_tab[i] = cos(th);
    }
    s->scan.sines = s->sines; s->scan.cosines = s->cosines_tab;
    s->params = (double *)malloc(sizeof(double) * ORC_NPARAMS * num_agents);
    for (a = 0; a < num_agents; a++) memcpy(s->params + a * ORC_NPARAMS, params18, sizeof(double) * ORC_NPARAMS);
    memcpy(s->sim_params, params18, sizeof(double) * ORC_NPARAMS);
    s->scan_angles = (double *)malloc(sizeof(double) * num_beams);
    s->beam_cos = (double *)malloc(sizeof(double) * num_beams);
    s->side_distances = (double *)malloc(sizeof(double) * num_beams);
    orc_build_beam_tables(num_beams, fov, params18[ORC_P_WIDTH], params18[ORC_P_LF],
                          params18[ORC_P_LR], s->scan_angles, s->beam_cos, s->side_distances);
    s->state = (double *)calloc((size_t)N * 7, sizeof(double));
    s->steer_buf = (double *)calloc((size_t)N * 2, sizeof(double));
    s->scans = (double *)calloc((size_t)N * num_beams, sizeof(double));
    s->collisions = (double *)calloc(N, sizeof(double));
    s->collision_idx = (double *)calloc(N, sizeof(double));
    s->agent_poses = (double *)calloc((size_t)N * 3, sizeof(double));
    s->buf_count = (int32_t *)calloc(N, sizeof(int32_t));
    s->in_collision = (int32_t *)calloc(N, sizeof(int32_t));
    s->step_count = (int32_t *)calloc(N, sizeof(int32_t));
    s->hit_rc = (int32_t *)calloc((size_t)N * num_beams * 2, sizeof(int32_t));
    for (i = 0; i < N; i++) s->collision_idx[i] = -1.;
    return s;
}

void orc_sim_destroy(orc_sim *s)
{
    if (!s) return;
    free(s->sines); free(s->cosines_tab); free(s->dt); free(s->params);
    free(s->scan_angles); free(s->beam_cos); free(s->side_distances); free(s->noise);
    free(s->state); free(s->steer_buf); free(s->scans); free(s->collisions);
    free(s->collision_idx); free(s->agent_poses); free(s->buf_count); free(s->in_collision);
    free(s->step_count); free(s->hit_rc);
    free(s);
}

void orc_sim_set_tables(orc_sim *s, const double *sines, const double *cosines)
{
    memcpy(s->sines, sines, sizeof(double) * s->scan.theta_dis);
    memcpy(s->cosines_tab, cosines, sizeof(double) * s->scan.theta_dis);
}

void orc_sim_set_map_dt(orc_sim *s, const double *dt, int height, int width, double resolution,
                        double orig_x, double orig_y, double orig_c, double orig_s)
{
    free(s->dt);
    s->dt = (double *)malloc(sizeof(double) * (size_t)height * width);
    memcpy(s->dt, dt, sizeof(double) * (size_t)height * width);
    s->scan.dt = s->dt; s->scan.height = height; s->scan.width = width;
    s->scan.resolution = resolution; s->scan.orig_x = orig_x; s->scan.orig_y = orig_y;
    s->scan.orig_c = orig_c; s->scan.orig_s = orig_s;
}

int orc_sim_set_params(orc_sim *s, int agent_idx, const double *params18)
{
    int a;
    if (agent_idx < 0) {
        for (a = 0; a < s->A; a++) memcpy(s->params + a * ORC_NPARAMS, params18, sizeof(double) * ORC_NPARAMS);
    } else if (agent_idx < s->A) {
        memcpy(s->params + agent_idx * ORC_NPARAMS, params18, sizeof(double) * ORC_NPARAMS);
    } else {
        return -1; /* IndexError, base_classes.py:534 */
    }
    return 0;
}

void orc_sim_set_noise(orc_sim *s, const double *noise, int T)
{
    free(s->noise);
    s->noise = 0;
    s->noise_T = 0;
    if (noise && T > 0) {
        s->noise = (double *)malloc(sizeof(double) * (size_t)T * s->B);
        memcpy(s->noise, noise, sizeof(double) * (size_t)T * s->B);
        s->noise_T = T;
    }
}

/* RaceCar.reset base_classes.py:183-204 */
void orc_sim_reset(orc_sim *s, const double *poses, const uint8_t *env_mask)
{
    int e, a;
    for (e = 0; e < s->E; e++) {
        if (env_mask && !env_mask[e]) continue;
        for (a = 0; a < s->A; a++) {
            int i = e * s->A + a;
            double *st = s->state + 7 * (size_t)i;
            memset(st, 0, 7 * sizeof(double));
            st[0] = poses[3 * i];
            st[1] = poses[3 * i + 1];
            st[4] = poses[3 * i + 2];
            s->steer_buf[2 * i] = s->steer_buf[2 * i + 1] = 0.;
            s->buf_count[i] = 0;
            s->in_collision[i] = 0;
            s->step_count[i] = 0; /* scan_rng re-seeded :204 -> noise row 0 next */
        }
    }
}

/* Simulator.step base_classes.py:553-612 for one env */
static void orc_sim_step_env(orc_sim *s, int e, const double *actions, int64_t *lookups)
{
    int A = s->A, B = s->B, a, j, k;
    double *verts = (double *)malloc(sizeof(double) * 8 * A);
    /* :568-574 integrate + map-only scan (+ noise) for every agent, snapshot poses */
    for (a = 0; a < A; a++) {
        int i = e * A + a;
        double *st = s->state + 7 * (size_t)i;
        double *scan = s->scans + (size_t)i * B;
        double scan_pose[3];
        int bc = s->buf_count[i];
        orc_update_pose(st, s->steer_buf + 2 * i, &bc, actions[2 * i], actions[2 * i + 1],
                        s->params + a * ORC_NPARAMS, s->time_step, s->integrator, s->lidar_dist,
                        scan_pose);
        s->buf_count[i] = bc;
        orc_get_scan(&s->scan, scan_pose, scan, s->hit_rc + (size_t)i * B * 2, lookups);
        if (s->noise) { /* laser_models.py:450-452 */
            const double *nz = s->noise + (size_t)(s->step_count[i] % s->noise_T) * B;
            for (k = 0; k < B; k++) scan[k] += nz[k];
        }
        s->step_count[i] += 1;
        s->agent_poses[3 * i] = st[0];
        s->agent_poses[3 * i + 1] = st[1];
        s->agent_poses[3 * i + 2] = st[4];
    }
    /* :577 / :536-550 GJK on post-integration poses, Simulator.params box */
    for (a = 0; a < A; a++)
        orc_get_vertices(s->agent_poses + 3 * (e * A + a), s->sim_params[ORC_P_LENGTH],
                         s->sim_params[ORC_P_WIDTH], verts + 8 * a);
    orc_collision_multiple(verts, A, s->collisions + e * A, s->collision_idx + e * A);
    /* :579-589 per agent in index order: iTTC on the noisy map-only scan, then opponents */
    for (a = 0; a < A; a++) {
        int i = e * A + a;
        double *st = s->state + 7 * (size_t)i;
        double *scan = s->scans + (size_t)i * B;
        double ego_pose[3];
        int hit = orc_check_ttc(scan, B, st[3], s->beam_cos, s->side_distances, s->ttc_thresh);
        if (hit) { /* :246-249 state[3:] = 0 */
            st[3] = 0.; st[4] = 0.; st[5] = 0.; st[6] = 0.;
        }
        s->in_collision[i] = hit;
        /* :206-227 ego pose is the live state (theta possibly just zeroed), opponents from
         * the :574 snapshot, box from the ego's own params */
        ego_pose[0] = st[0]; ego_pose[1] = st[1]; ego_pose[2] = st[4];
        for (j = 0; j < A; j++) {
            double ov[8];
            if (j == a) continue;
            orc_get_vertices(s->agent_poses + 3 * (e * A + j), s->params[a * ORC_NPARAMS + ORC_P_LENGTH],
                             s->params[a * ORC_NPARAMS + ORC_P_WIDTH], ov);
            orc_ray_cast(ego_pose, scan, s->scan_angles, B, ov);
        }
        if (hit) s->collisions[i] = 1.; /* :588-589 */
    }
    free(verts);
}

void orc_sim_step(orc_sim *s, const double *actions, int n_threads)
{
    int e;
    int64_t total = 0;
#ifdef _OPENMP
    if (n_threads > 1) {
#pragma omp parallel for num_threads(n_threads) schedule(dynamic, 16) reduction(+ : total)
        for (e = 0; e < s->E; e++) {
            int64_t lk = 0;
            orc_sim_step_env(s, e, actions, &lk);
            total += lk;
        }
        s->lookups += total;
        return;
    }
#endif
    (void)n_threads;
    for (e = 0; e < s->E; e++) orc_sim_step_env(s, e, actions, &total);
    s->lookups += total;
}

/* The CPU baseline's best shape (bench.py cpu_baseline): envs never interact, so every env is walked through ALL
 * `steps` steps by one thread — no barrier per step, the env's state stays in that core's cache — with the in-place
 * re-seat of an env whose ego (slot 0) collided done right here (what bench.py's Python loop did between steps:
 * orc_sim_reset with a mask).  actions = [n_sets][N][2], set t / steps_per_set is applied at step t (the bench's
 * pre-drawn action sets).  Identical, env by env, to `steps` calls of orc_sim_step + masked resets
 * (tests/test_oracle_golden.py).  Returns the number of re-seats. */
int64_t orc_sim_rollout(orc_sim *s, const double *actions, int n_sets, int steps_per_set, int steps,
                        const double *start_poses, int reseat_on_ego_collision, int n_threads)
{
    int e;
    int64_t total = 0, reseats = 0;
    const size_t set_stride = (size_t)2 * s->E * s->A;
    if (n_threads < 1) n_threads = 1;
#ifdef _OPENMP
#pragma omp parallel for num_threads(n_threads) schedule(dynamic, 1) reduction(+ : total, reseats)
#endif
    for (e = 0; e < s->E; e++) {
        int t, a;
        int64_t lk = 0;
        for (t = 0; t < steps; t++) {
            int set = t / steps_per_set;
            if (set >= n_sets) set = n_sets - 1;
            orc_sim_step_env(s, e, actions + set_stride * set, &lk);
            if (reseat_on_ego_collision && s->collisions[e * s->A] != 0.) {
                for (a = 0; a < s->A; a++) { /* orc_sim_reset for this env */
                    int i = e * s->A + a;
                    double *st = s->state + 7 * (size_t)i;
                    memset(st, 0, 7 * sizeof(double));
                    st[0] = start_poses[3 * i];
                    st[1] = start_poses[3 * i + 1];
                    st[4] = start_poses[3 * i + 2];
                    s->steer_buf[2 * i] = s->steer_buf[2 * i + 1] = 0.;
                    s->buf_count[i] = 0;
                    s->in_collision[i] = 0;
                    s->step_count[i] = 0;
                }
                reseats += 1;
            }
        }
        total += lk;
    }
    s->lookups += total;
    return reseats;
}

double *orc_sim_state(orc_sim *s) { return s->state; }
double *orc_sim_scans(orc_sim *s) { return s->scans; }
double *orc_sim_collisions(orc_sim *s) { return s->collisions; }
double *orc_sim_collision_idx(orc_sim *s) { return s->collision_idx; }
double *orc_sim_agent_poses(orc_sim *s) { return s->agent_poses; }
int32_t *orc_sim_in_collision(orc_sim *s) { return s->in_collision; }
int32_t *orc_sim_step_count(orc_sim *s) { return s->step_count; }
int32_t *orc_sim_hit_rc(orc_sim *s) { return s->hit_rc; }
int64_t orc_sim_lookups(orc_sim *s) { return s->lookups; }

/* ------------------------------------------------------------------ examples/waypoint_follow.py */

/* waypoint_follow.py:15-50.  Segment k joins waypoint k and k+1; the projection parameter is
 * clipped to [0,1]; the first segment with the smallest distance wins (np.argmin). */
int orc_nearest_on_trajectory(const double *wp, int M, double px, double py, double *dist,
                              double *t_out)
{
    int best = 0;
    double best_d = INFINITY, best_t = 0.0;
    for (int k = 0; k + 1 < M; ++k) {
        const double ax = wp[3 * k], ay = wp[3 * k + 1];
        const double dx = wp[3 * (k + 1)] - ax, dy = wp[3 * (k + 1) + 1] - ay;
        const double l2 = dx * dx + dy * dy;
        const double dot = (px - ax) * dx + (py - ay) * dy;
        double t = dot / l2;
        if (t < 0.0) t = 0.0;
        if (t > 1.0) t = 1.0;
        const double qx = ax + t * dx, qy = ay + t * dy;
        const double ex = px - qx, ey = py - qy;
        const double d = sqrt(ex * ex + ey * ey);
        if (d < best_d) {
            best_d = d;
            best_t = t;
            best = k;
        }
    }
    *dist = best_d;
    *t_out = best_t;
    return best;
}

/* one segment of :52-132: does the circle cut start -> end (+1e-6 on both end coordinates, :70)?
 * first = the segment the search starts in, where only parameters >= start_t count (:86-96) */
static int circle_hits_segment(double sx, double sy, double ex, double ey, double px, double py,
                               double radius, int first, double start_t)
{
    const double vx = (ex + 1e-6) - sx, vy = (ey + 1e-6) - sy;
    const double a = vx * vx + vy * vy;
    const double b = 2.0 * (vx * (sx - px) + vy * (sy - py));
    const double c = (sx * sx + sy * sy) + (px * px + py * py) - 2.0 * (sx * px + sy * py) - radius * radius;
    double disc = b * b - 4 * a * c;
    if (disc < 0) return 0;
    disc = sqrt(disc);
    const double t1 = (-b - disc) / (2.0 * a);
    const double t2 = (-b + disc) / (2.0 * a);
    if (first) {
        if (t1 >= 0.0 && t1 <= 1.0 && t1 >= start_t) return 1;
        if (t2 >= 0.0 && t2 <= 1.0 && t2 >= start_t) return 1;
        return 0;
    }
    return (t1 >= 0.0 && t1 <= 1.0) || (t2 >= 0.0 && t2 <= 1.0);
}

int orc_first_point_on_circle(const double *wp, int M, double px, double py, double radius,
                              double start)
{
    const int start_i = (int)start;
    const double start_t = fmod(start, 1.0);
    for (int i = start_i; i + 1 < M; ++i)
        if (circle_hits_segment(wp[3 * i], wp[3 * i + 1], wp[3 * (i + 1)], wp[3 * (i + 1) + 1], px, py, radius,
                                i == start_i, start_t))
            return i;
    /* wrap=True :108-130: segments -1 .. start_i-1 with Python's modulo indexing; the index that is
     * reported is the loop variable itself (-1 for the closing segment), and the caller then reads
     * wpts[-1] */
    for (int i = -1; i < start_i; ++i) {
        const int k0 = ((i % M) + M) % M, k1 = (((i + 1) % M) + M) % M;
        if (circle_hits_segment(wp[3 * k0], wp[3 * k0 + 1], wp[3 * k1], wp[3 * k1 + 1], px, py, radius, 0, 0.0))
            return i < 0 ? M + i : i;   /* as an index into the waypoint array */
    }
    return -1;
}

void orc_pure_pursuit_plan(const double *wp, int M, const double pose[3], double lookahead,
                           double vgain, double wheelbase, double max_reacquire,
                           double action[2])
{
    double dist, t;
    const int i = orc_nearest_on_trajectory(wp, M, pose[0], pose[1], &dist, &t);
    double gx, gy, speed;
    if (dist < lookahead) {
        const int i2 = orc_first_point_on_circle(wp, M, pose[0], pose[1], lookahead, (double)i + t);
        if (i2 < 0) { action[0] = 0.0; action[1] = 4.0; return; }     /* :213-214 */
        gx = wp[3 * i2];                                              /* the waypoint, not the cut (:193) */
        gy = wp[3 * i2 + 1];
        speed = wp[3 * i + 2];                                        /* speed of the NEAREST index (:195) */
    } else if (dist < max_reacquire) {
        gx = wp[3 * i];
        gy = wp[3 * i + 1];
        speed = wp[3 * i + 2];
    } else {
        action[0] = 0.0;
        action[1] = 4.0;
        return;
    }
    /* get_actuation :134-145 */
    const double wy = sin(-pose[2]) * (gx - pose[0]) + cos(-pose[2]) * (gy - pose[1]);
    double steer = 0.0;
    if (!(fabs(wy) < 1e-6)) {
        const double radius = 1 / (2.0 * wy / (lookahead * lookahead));
        steer = atan(wheelbase / radius);
    }
    action[0] = steer;
    action[1] = vgain * speed;
}
