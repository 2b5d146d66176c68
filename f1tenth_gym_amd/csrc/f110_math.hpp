// f110_math.hpp — scalar float64 building blocks of the env.step() hot path for gfx950.
//
// Every function is `__host__ __device__` so the very same code that the HIP kernels inline can
// be exercised on the build container's CPU by tests/host_harness (there is no GPU there);
// the product only ever runs the __device__ instantiations (f110_hip.hip).
//
// Bit-parity discipline: IEEE float64, the reference's operation order, and
// the translation unit is compiled with -ffp-contract=off so `a*b + c` never becomes an FMA.
// Reference citations are gym/f110_gym/envs/<file>:<line> of f1tenth_gym v0.2.1.
#pragma once

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#define F110_HD __host__ __device__ __forceinline__

namespace f110 {

constexpr double kPi = 3.14159265358979323846;
constexpr double kTwoPi = 2.0 * kPi;

// vehicle parameter slots, key order of f110_env.py:130
enum { P_MU = 0, P_CSF, P_CSR, P_LF, P_LR, P_H, P_M, P_I, P_SMIN, P_SMAX, P_SVMIN, P_SVMAX,
       P_VSWITCH, P_AMAX, P_VMIN, P_VMAX, P_WIDTH, P_LENGTH, NPARAMS };

struct VehicleParams {
    double v[NPARAMS];
};

// cos and sin of ONE angle.  On the device a single argument reduction serves both (ocml's sincos returns what its
// sin and its cos return — the ray-cast has relied on that since round 1); the host instantiation (tests/host_harness)
// makes the two libm calls the oracle makes.  A float64 sin or cos is ~250 dependent instructions of a kernel that is
// a dependent chain (k_integrate) or bound by its vector-ALU work (the finalize kernels): one call instead of two.
F110_HD void cos_sin(double a, double &c, double &s)
{
#if defined(__HIP_DEVICE_COMPILE__)
    sincos(a, &s, &c);
#else
    c = cos(a);
    s = sin(a);
#endif
}

// ------------------------------------------------------------------ dynamic_models.py
// accl_constraints :29-60
F110_HD double clamp_accel(double vel, double accl, const VehicleParams &p)
{
    const double a_max = p.v[P_AMAX];
    const double pos_limit = (vel > p.v[P_VSWITCH]) ? a_max * p.v[P_VSWITCH] / vel : a_max;
    if ((vel <= p.v[P_VMIN] && accl <= 0) || (vel >= p.v[P_VMAX] && accl >= 0)) return 0.;
    if (accl <= -a_max) return -a_max;
    if (accl >= pos_limit) return pos_limit;
    return accl;
}

// steering_constraint :62-87
F110_HD double clamp_steer_rate(double steer_angle, double sv, const VehicleParams &p)
{
    if ((steer_angle <= p.v[P_SMIN] && sv <= 0) || (steer_angle >= p.v[P_SMAX] && sv >= 0)) return 0.;
    if (sv <= p.v[P_SVMIN]) return p.v[P_SVMIN];
    if (sv >= p.v[P_SVMAX]) return p.v[P_SVMAX];
    return sv;
}

// vehicle_dynamics_ks :90-121 — x[0..4], raw inputs (sv, accl); f[0..4]
F110_HD void rhs_kinematic(const double *x, double sv_in, double accl_in, const VehicleParams &p, double *f)
{
    const double lwb = p.v[P_LF] + p.v[P_LR];
    const double u0 = clamp_steer_rate(x[2], sv_in, p);
    const double u1 = clamp_accel(x[3], accl_in, p);
    double c4, s4;
    cos_sin(x[4], c4, s4);
    f[0] = x[3] * c4;
    f[1] = x[3] * s4;
    f[2] = u0;
    f[3] = u1;
    f[4] = x[3] / lwb * tan(x[2]);
}

// Where the low-speed branch's tan(steer) and cos(steer) come from: computed in place (everywhere but k_integrate), or
// handed over by the kernel's second wave, which runs one stage ahead (LowTrigShared in f110_kernels.hpp).
struct LowTrigDirect {
    F110_HD void begin_stage(int /*stage*/) const {}   // (called by every lane at the top of every stage)
    F110_HD void operator()(int /*stage*/, double steer, double &tn, double &cd) const
    {
        tn = tan(steer);
        cd = cos(steer);
    }
};

// vehicle_dynamics_st :123-176 — x[0..6] = (x, y, steer, v, yaw, yaw_rate, slip)
template <class LowTrig>
F110_HD void rhs_single_track_with(const double *x, double sv_in, double accl_in, const VehicleParams &p, double *f, int stage,
                                   const LowTrig &low_trig)
{
    const double g = 9.81;
    const double u0 = clamp_steer_rate(x[2], sv_in, p);
    const double u1 = clamp_accel(x[3], accl_in, p);
    // :152-160 low-speed kinematic branch (the reference feeds the constrained inputs through
    // vehicle_dynamics_ks, which constrains them once more) or the single-track model proper.  Both start with
    // velocity * (cos, sin) of ONE angle — the heading, or heading + slip — so a wave with lanes on either
    // side (every wave of a batch that re-seats crashed cars: they restart at v = 0) evaluates ONE cos_sin for
    // both instead of one per branch: the same operations per lane, ~150 instructions off every RK4 stage of a
    // kernel that is a dependent chain.
    const bool low = fabs(x[3]) < 0.5;
    double ca, sa;
    cos_sin(low ? x[4] : x[6] + x[4], ca, sa);
    f[0] = x[3] * ca;
    f[1] = x[3] * sa;
    if (low) {
        const double lwb = p.v[P_LF] + p.v[P_LR];
        double tn, cd;
        low_trig(stage, x[2], tn, cd);
        f[2] = clamp_steer_rate(x[2], u0, p);   // vehicle_dynamics_ks :108-109 on the already constrained inputs
        f[3] = clamp_accel(x[3], u1, p);
        f[4] = x[3] / lwb * tn;
        f[5] = u1 / lwb * tn + x[3] / (lwb * (cd * cd)) * u0;
        f[6] = 0.;
        return;
    }
    const double mu = p.v[P_MU], csf = p.v[P_CSF], csr = p.v[P_CSR], lf = p.v[P_LF], lr = p.v[P_LR];
    const double h = p.v[P_H], m = p.v[P_M], iz = p.v[P_I];
    const double rear = g * lr - u1 * h;   // (g*lr - u[1]*h)
    const double front = g * lf + u1 * h;  // (g*lf + u[1]*h)
    const double wb = lr + lf;
    f[2] = u0;
    f[3] = u1;
    f[4] = x[5];
    // :169-171
    const double t1 = -mu * m / (x[3] * iz * wb) * ((lf * lf) * csf * rear + (lr * lr) * csr * front) * x[5];
    const double t2 = mu * m / (iz * wb) * (lr * csr * front - lf * csf * rear) * x[6];
    const double t3 = mu * m / (iz * wb) * lf * csf * rear * x[2];
    f[5] = t1 + t2 + t3;
    // :172-174
    const double s1 = (mu / ((x[3] * x[3]) * wb) * (csr * front * lr - csf * rear * lf) - 1) * x[5];
    const double s2 = mu / (x[3] * wb) * (csr * front + csf * rear) * x[6];
    const double s3 = mu / (x[3] * wb) * (csf * rear) * x[2];
    f[6] = s1 - s2 + s3;
}

F110_HD void rhs_single_track(const double *x, double sv_in, double accl_in, const VehicleParams &p, double *f)
{
    rhs_single_track_with(x, sv_in, accl_in, p, f, 0, LowTrigDirect());
}

// pid :178-221
F110_HD void speed_steer_controller(double speed, double steer, double cur_speed, double cur_steer,
                                    const VehicleParams &p, double &accl, double &sv)
{
    const double steer_diff = steer - cur_steer;
    sv = (fabs(steer_diff) > 1e-4) ? (steer_diff / fabs(steer_diff)) * p.v[P_SVMAX] : 0.0;
    const double vel_diff = speed - cur_speed;
    double kp;
    if (cur_speed > 0.)
        kp = (vel_diff > 0) ? 10.0 * p.v[P_AMAX] / p.v[P_VMAX] : 10.0 * p.v[P_AMAX] / (-p.v[P_VMIN]);
    else
        kp = (vel_diff > 0) ? 2.0 * p.v[P_AMAX] / p.v[P_VMAX] : 2.0 * p.v[P_AMAX] / (-p.v[P_VMIN]);
    accl = kp * vel_diff;
}

// ------------------------------------------------------------------ base_classes.py
// RaceCar.update_pose :256-409 minus the scan.  buf[0] = newest delayed steer command.
template <class LowTrig>
F110_HD void advance_vehicle_with(double *st, double &buf0, double &buf1, int &buf_cnt, double raw_steer,
                                  double speed_cmd, const VehicleParams &p, double dt, int integrator,
                                  double lidar_dist, double *scan_pose, const LowTrig &low_trig)
{
    // :271-278 two-step steering delay
    double steer = 0.;
    if (buf_cnt < 2) {
        buf_cnt += 1;
    } else {
        steer = buf1;
    }
    buf1 = buf0;
    buf0 = raw_steer;

    double accl, sv;
    speed_steer_controller(speed_cmd, steer, st[3], st[2], p, accl, sv);  // :282

    {
        // Integrator.RK4 :284-373 (k1..k4, state + dt/6 * (k1 + 2 k2 + 2 k3 + k4)) and Integrator.Euler
        // :375-395 as ONE loop over the stages, not unrolled: the right-hand side is the bulk of this
        // kernel's code, and a kernel that runs one wave per SIMD pays for every instruction byte it
        // has to fetch.  Same operations in the same order as the reference's straight-line code:
        // acc = ((k1 + 2*k2) + 2*k3) + 1*k4, stage inputs st + dt*(k/2), st + dt*(k/2), st + dt*k.
        const int stages = (integrator == 1) ? 4 : 1;
        double acc[7], tmp[7], kk[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) tmp[i] = st[i];
#pragma unroll 1
        for (int sidx = 0; sidx < stages; ++sidx) {
            low_trig.begin_stage(sidx);
            rhs_single_track_with(tmp, sv, accl, p, kk, sidx, low_trig);
            const double wgt = (sidx == 1 || sidx == 2) ? 2.0 : 1.0;
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                acc[i] = (sidx == 0) ? kk[i] : acc[i] + wgt * kk[i];
                const double h = (sidx < 2) ? kk[i] / 2 : kk[i];
                tmp[i] = st[i] + dt * h;
            }
        }
        const double w = (integrator == 1) ? dt * (1. / 6.) : dt;
#pragma unroll
        for (int i = 0; i < 7; ++i) st[i] = st[i] + w * acc[i];
    }
    // :400-404
    if (st[4] > kTwoPi)
        st[4] = st[4] - kTwoPi;
    else if (st[4] < 0)
        st[4] = st[4] + kTwoPi;
    // :407-409.  With the lidar on the reference point (lidar_dist == 0., the default of F110Env) the products are
    // +-0. and x + (+-0.) == x for every x except -0. — the two trig calls are skipped exactly then (finite heading,
    // no negative zero among the coordinates): the same bits, ~500 instructions off k_integrate's chain
    const bool on_axle = lidar_dist == 0.0 && fabs(st[4]) < 1e300 && !(st[0] == 0.0 && signbit(st[0])) && !(st[1] == 0.0 && signbit(st[1]));
    if (on_axle) {
        scan_pose[0] = st[0];
        scan_pose[1] = st[1];
    } else {
        double ch, sh;
        cos_sin(st[4], ch, sh);
        scan_pose[0] = st[0] + lidar_dist * ch;
        scan_pose[1] = st[1] + lidar_dist * sh;
    }
    scan_pose[2] = st[4];
}

F110_HD void advance_vehicle(double *st, double &buf0, double &buf1, int &buf_cnt, double raw_steer,
                             double speed_cmd, const VehicleParams &p, double dt, int integrator,
                             double lidar_dist, double *scan_pose)
{
    advance_vehicle_with(st, buf0, buf1, buf_cnt, raw_steer, speed_cmd, p, dt, integrator, lidar_dist, scan_pose, LowTrigDirect());
}

// The steering angle and the velocity through the RK4 stages, on their own: their derivatives — the constrained
// steering rate and acceleration, constrained a second time inside the low-speed branch (:152-160 -> :108-109) —
// depend on nothing else, so a second wave can walk (x[2], x[3]) ahead of the integration and have the low-speed
// branch's tan / cos ready (k_integrate).  Same operations in the same order as advance_vehicle_with's loop.
// emit(stage, tan, cos) is called by the lanes whose stage takes the low-speed branch, emit.end_stage(stage) by all.
template <class Emit>
F110_HD void low_speed_trig_ahead(double steer0, double vel0, double buf1, int buf_cnt, double speed_cmd, const VehicleParams &p,
                                  double dt, int integrator, const Emit &emit)
{
    const double steer = (buf_cnt < 2) ? 0. : buf1;   // :271-278 (the command that leaves the delay buffer this step)
    double accl, sv;
    speed_steer_controller(speed_cmd, steer, vel0, steer0, p, accl, sv);
    const int stages = (integrator == 1) ? 4 : 1;
    double x2 = steer0, x3 = vel0;
    for (int sidx = 0; sidx < stages; ++sidx) {
        const bool low = fabs(x3) < 0.5;
        if (low) emit(sidx, tan(x2), cos(x2));
        emit.end_stage(sidx);   // (every lane, every stage)
        const double u0 = clamp_steer_rate(x2, sv, p), u1 = clamp_accel(x3, accl, p);
        const double f2 = low ? clamp_steer_rate(x2, u0, p) : u0;
        const double f3 = low ? clamp_accel(x3, u1, p) : u1;
        const double h2 = (sidx < 2) ? f2 / 2 : f2, h3 = (sidx < 2) ? f3 / 2 : f3;
        x2 = steer0 + dt * h2;
        x3 = vel0 + dt * h3;
    }
}

// ---- the RK4 step taken apart by what depends on what (k_integrate_fan, round 4) --------------------------------
// advance_vehicle_with is one dependent chain of ~1900 instructions per agent: four stages, each a cos / sin of the
// heading, the low-speed branch's tan / cos of the steering angle or the single-track branch's three divisions by the
// velocity, and the stage update.  Its dataflow is much shallower than that:
//   * (steer, velocity) walk through the stages on their own (their derivatives are the constrained inputs);
//   * the low-speed branch's f4, f5 and the single-track branch's coefficients of (yaw rate, slip) in f5, f6 depend on
//     that walk only;
//   * given those, (yaw, yaw rate, slip) advance by a handful of multiply-adds per stage;
//   * the position derivatives — velocity x (cos, sin)(heading) — feed nothing but the position itself.
// So: one wave per (role, stage) computes the expensive piece it owns (fan_low, fan_dyn, fan_pos), the main wave
// chains the cheap recurrence (fan_main) and combines.  Every operation of advance_vehicle_with is executed once, on
// the same operands, in the same order within each expression: bit-identical (tests/test_host_math.py).
struct FanWalk {   // (steer, velocity) and their constrained rates at one stage
    double x2, x3, u0, u1;
    bool low;
};
// the pid of :282 — (accl, sv) — from what every role reads of the agent: raw state, delay buffer, speed command
F110_HD void fan_inputs(double steer0, double vel0, double buf1, int buf_cnt, double speed_cmd, const VehicleParams &p, double &accl, double &sv)
{
    const double steer = (buf_cnt < 2) ? 0. : buf1;   // :271-278 (the command that leaves the delay buffer this step)
    speed_steer_controller(speed_cmd, steer, vel0, steer0, p, accl, sv);
}
// the walk up to `stage` (0..3): same operations as low_speed_trig_ahead / advance_vehicle_with's loop for x[2], x[3]
F110_HD FanWalk fan_walk(double steer0, double vel0, double accl, double sv, const VehicleParams &p, double dt, int stage)
{
    FanWalk w;
    w.x2 = steer0;
    w.x3 = vel0;
    for (int sidx = 0;; ++sidx) {
        w.low = fabs(w.x3) < 0.5;
        w.u0 = clamp_steer_rate(w.x2, sv, p);
        w.u1 = clamp_accel(w.x3, accl, p);
        if (sidx == stage) return w;
        const double f2 = w.low ? clamp_steer_rate(w.x2, w.u0, p) : w.u0;
        const double f3 = w.low ? clamp_accel(w.x3, w.u1, p) : w.u1;
        const double h2 = (sidx < 2) ? f2 / 2 : f2, h3 = (sidx < 2) ? f3 / 2 : f3;
        w.x2 = steer0 + dt * h2;
        w.x3 = vel0 + dt * h3;
    }
}
// low-speed branch (:152-160): f[4], f[5] of rhs_single_track_with
F110_HD void fan_low(const FanWalk &w, const VehicleParams &p, double &f4, double &f5)
{
    const double lwb = p.v[P_LF] + p.v[P_LR];
    const double tn = tan(w.x2), cd = cos(w.x2);
    f4 = w.x3 / lwb * tn;
    f5 = w.u1 / lwb * tn + w.x3 / (lwb * (cd * cd)) * w.u0;
}
// single-track branch (:164-174): f5 = (k[0] * yaw_rate + k[1] * slip) + k[2], f6 = (k[3] * yaw_rate - k[4] * slip) + k[5]
F110_HD void fan_dyn(const FanWalk &w, const VehicleParams &p, double *k)
{
    const double g = 9.81;
    const double mu = p.v[P_MU], csf = p.v[P_CSF], csr = p.v[P_CSR], lf = p.v[P_LF], lr = p.v[P_LR];
    const double h = p.v[P_H], m = p.v[P_M], iz = p.v[P_I];
    const double rear = g * lr - w.u1 * h;
    const double front = g * lf + w.u1 * h;
    const double wb = lr + lf;
    k[0] = -mu * m / (w.x3 * iz * wb) * ((lf * lf) * csf * rear + (lr * lr) * csr * front);
    k[1] = mu * m / (iz * wb) * (lr * csr * front - lf * csf * rear);
    k[2] = mu * m / (iz * wb) * lf * csf * rear * w.x2;
    k[3] = mu / ((w.x3 * w.x3) * wb) * (csr * front * lr - csf * rear * lf) - 1;
    k[4] = mu / (w.x3 * wb) * (csr * front + csf * rear);
    k[5] = mu / (w.x3 * wb) * (csf * rear) * w.x2;
}
// position derivatives of one stage: velocity x (cos, sin)(angle), angle = heading (low speed) or heading + slip
F110_HD void fan_pos(double angle, double vel, double &f0, double &f1)
{
    double ca, sa;
    cos_sin(angle, ca, sa);
    f0 = vel * ca;
    f1 = vel * sa;
}
// The main chain.  take_low(stage, f4, f5) / take_dyn(stage, k6) fetch what fan_low / fan_dyn produced for the stages that
// took that branch; emit_angle(stage, angle, vel) hands the stage's position-derivative operands on.  st[2..6] are
// advanced to the end of the step here (st[0], st[1] by fan_combine_position); returns nothing else.
template <class TakeLow, class TakeDyn, class EmitAngle>
F110_HD void fan_main(double *st, double accl, double sv, const VehicleParams &p, double dt, const TakeLow &take_low, const TakeDyn &take_dyn,
                      const EmitAngle &emit_angle)
{
    double acc[5], tmp[5], kk[5];   // components 2..6
#pragma unroll
    for (int i = 0; i < 5; ++i) tmp[i] = st[2 + i];
#pragma unroll 1
    for (int sidx = 0; sidx < 4; ++sidx) {
        const double x2 = tmp[0], x3 = tmp[1], x4 = tmp[2], x5 = tmp[3], x6 = tmp[4];
        const double u0 = clamp_steer_rate(x2, sv, p);
        const double u1 = clamp_accel(x3, accl, p);
        const bool low = fabs(x3) < 0.5;
        emit_angle(sidx, low ? x4 : x6 + x4, x3);
        if (low) {
            kk[0] = clamp_steer_rate(x2, u0, p);
            kk[1] = clamp_accel(x3, u1, p);
            take_low(sidx, kk[2], kk[3]);
            kk[4] = 0.;
        } else {
            double k[6];
            take_dyn(sidx, k);
            kk[0] = u0;
            kk[1] = u1;
            kk[2] = x5;
            kk[3] = (k[0] * x5 + k[1] * x6) + k[2];
            kk[4] = (k[3] * x5 - k[4] * x6) + k[5];
        }
        const double wgt = (sidx == 1 || sidx == 2) ? 2.0 : 1.0;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            acc[i] = (sidx == 0) ? kk[i] : acc[i] + wgt * kk[i];
            const double h = (sidx < 2) ? kk[i] / 2 : kk[i];
            tmp[i] = st[2 + i] + dt * h;
        }
    }
    const double w = dt * (1. / 6.);
#pragma unroll
    for (int i = 0; i < 5; ++i) st[2 + i] = st[2 + i] + w * acc[i];
}
// x, y: the four stages' derivatives combined as advance_vehicle_with combines them
F110_HD double fan_combine(double x, double dt, double k1, double k2, double k3, double k4)
{
    double acc = k1;
    acc = acc + 2.0 * k2;
    acc = acc + 2.0 * k3;
    acc = acc + 1.0 * k4;
    return x + (dt * (1. / 6.)) * acc;
}
// what follows the integration in advance_vehicle_with: yaw wrap :400-404 and the lidar pose :407-409
F110_HD void fan_finish(double *st, double lidar_dist, double *scan_pose)
{
    if (st[4] > kTwoPi)
        st[4] = st[4] - kTwoPi;
    else if (st[4] < 0)
        st[4] = st[4] + kTwoPi;
    const bool on_axle = lidar_dist == 0.0 && fabs(st[4]) < 1e300 && !(st[0] == 0.0 && signbit(st[0])) && !(st[1] == 0.0 && signbit(st[1]));
    if (on_axle) {
        scan_pose[0] = st[0];
        scan_pose[1] = st[1];
    } else {
        double ch, sh;
        cos_sin(st[4], ch, sh);
        scan_pose[0] = st[0] + lidar_dist * ch;
        scan_pose[1] = st[1] + lidar_dist * sh;
    }
    scan_pose[2] = st[4];
}

// ------------------------------------------------------------------ laser_models.py
enum { LAYOUT_ROWMAJOR = 0, LAYOUT_PADDED = 3 };   // (1 = 4x4 tiles and 2 = byte codes + LUT: retired in round 5)

struct ScanConst {
    const double *table;   // the distance table dt[r][c] (PADDED: the interior of the padded copy, its row pitch)
    const double *table_rm;  // the same (a separate row-major original only while a map too large for PADDED is loaded)
    const double *reserved_pad_t; // (rounds 5-6: a tiled / row-pair copy of the PADDED table; retired — the field keeps the kernel-argument layout)
    const void *reserved_lut;
    const double2 *cs;     // (cos, sin) of linspace(0, 2pi, theta_dis), interleaved
    int32_t height, width, pad_tiles, theta_dis;
    int32_t num_beams, res_pow2, ident_rot, row_bytes;  // row_bytes = width * 8
    int32_t reserved_pad_t_row_bytes, pad1;
    double res, inv_res, orig_x, orig_y, orig_c, orig_s;
    double w_res, h_res;   // width*resolution, height*resolution (xy_2_rc :79)
    double oob_value;      // dt[-1,-1]: what an out-of-bounds sample reads (:80-81,:103)
    double eps, max_range, fov, theta_inc, dir_guard, inv_theta_dis;
    // PADDED: the row-major table surrounded by `pad_border` cells of oob_value on every side,
    // addressed in fixed point (see march_padded).  pad == nullptr: not available for this map.
    const double *pad;
    int32_t pad_border, pad_row_bytes, pad_width, pad_height;
    double pad_lo, pad_hi_x, pad_hi_y;  // lidar positions (padded cell units) whose rays stay inside
    double pad_cx, pad_cy;              // padded_position: u = x*axx + y*axy + cx  (IDENT: axx = inv_res, axy = 0)
    double pad_axx, pad_axy, pad_ayx, pad_ayy;
    int32_t pad_max_samples, pad_reserved;  // samples per ray the fixed-point error bound covers
};

// The PADDED fast path keeps a ray's position in padded-table cell units u (approximate on
// purpose: it differs from the reference's float64 position by less than kPadGuard / 4 cells,
// setup_padded) and reads the cell as the integer part of the fixed-point word that adding
// kFixBig = 1.5 * 2^36 leaves in the low mantissa bits (ulp = 2^-16 cells):
//   lo32(u + kFixBig) = round(u * 65536),  cell = word >> 16,  fraction = word & 0xffff.
// The addition rounds to nearest, so only a position within 2^-17 cells of a cell boundary can
// come out with fraction == 0; for those (1 sample in 30 000) the distance to the boundary is
// looked at in full precision: further than kPadGuard = 2^-27 cells -> the cell is floor(u), the
// same cell the reference's int(x_rot / resolution) and range tests (laser_models.py:79-84) pick,
// because everything that differs from the reference is smaller than kPadGuard / 4; closer -> the
// ray is re-marched with the reference's arithmetic (about one ray in 10^7).  Out-of-range
// samples need no test at all: the border cells hold the value the reference reads for them.
constexpr double kFixBig = 103079215104.0;  // 1.5 * 2^36
constexpr int kFixFracBits = 16;
constexpr int kPadSlack = 64;               // extra border cells so a lidar slightly outside the map stays fast
constexpr double kPadGuard = 7.450580596923828e-09;  // 2^-27 cells: closer to a cell boundary than this -> exact march

F110_HD uint32_t low_word(double v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__double2loint(v);
#else
    uint64_t b;
    __builtin_memcpy(&b, &v, 8);
    return (uint32_t)b;
#endif
}

// border width in cells for a map: rays sample at most max_range (+ rounding) away from the lidar
F110_HD int padded_border_cells(double max_range, double inv_res)
{
    return (int)ceil(max_range * inv_res) + 2 + kPadSlack;
}

// fills the PADDED constants of k (k.pad itself is set by the caller); false when the map is too
// large for 16-bit cell coordinates / 32-bit byte offsets and the exact path must be used
F110_HD bool setup_padded(ScanConst &k)
{
    const int b = padded_border_cells(k.max_range, k.inv_res);
    const long long wp = (long long)k.width + 2 * b, hp = (long long)k.height + 2 * b;
    k.pad = nullptr;
    k.pad_border = b;
    if (wp >= 65536 || hp >= 65536 || wp * 8 >= (1 << 24) || wp * hp * 8 >= (1ll << 32)) return false;
    k.pad_width = (int)wp;
    k.pad_height = (int)hp;
    k.pad_row_bytes = (int)(wp * 8);
    const double reach = ceil(k.max_range * k.inv_res) + 2.0;
    k.pad_lo = reach + 1.0;
    k.pad_hi_x = (double)wp - reach - 2.0;
    k.pad_hi_y = (double)hp - reach - 2.0;
    k.pad_axx = k.orig_c * k.inv_res;
    k.pad_axy = k.orig_s * k.inv_res;
    k.pad_ayx = -k.orig_s * k.inv_res;
    k.pad_ayy = k.orig_c * k.inv_res;
    k.pad_cx = (double)b - (k.orig_x * k.orig_c + k.orig_y * k.orig_s) * k.inv_res;
    k.pad_cy = (double)b - (k.orig_y * k.orig_c - k.orig_x * k.orig_s) * k.inv_res;
    // Error budget of march_padded against the reference's float64 positions, in cells.  Per
    // sample: the reference rounds x += d*c (<= 1 ulp of the largest coordinate), the fixed-point
    // side rounds one fma (<= 1/2 ulp of the padded width) and carries the 2^-52 relative error of
    // cu.  Once: the start position, the offsets, the rotation and the final division.
    const double xmax = fabs(k.orig_x) + fabs(k.orig_y) + ((double)k.width + (double)k.height) * k.res + 2.0 * k.max_range + 1.0;
    const double ulp_x = xmax * 2.220446049250313e-16, ulp_u = (double)(wp > hp ? wp : hp) * 2.220446049250313e-16;
    const double per_sample = ulp_x * k.inv_res + ulp_u + 4.0 * 2.220446049250313e-16 * k.max_range * k.inv_res;
    const double once = 8.0 * (ulp_x * k.inv_res + ulp_u);
    const double n_max = floor((0.25 * kPadGuard - once) / per_sample);
    if (!(n_max >= 256.0)) return false;  // coordinates too large for the bound to be useful
    k.pad_max_samples = n_max > 1e6 ? 1000000 : (int)n_max;
    return true;
}

// 24-bit multiply (v_mul_u32_u24 / v_mad_u32_u24 are full-rate; the 32-bit v_mul_lo_u32 is not).
// Rows, columns and row pitches of any realistic map are far below 2^24.
F110_HD uint32_t mul24(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul24(a, b);
#else
    return a * b;
#endif
}

// One table value.  (Rounds 1-4 also had a 4x4-tiled float64 layout and a 1-byte-code + exact-value-LUT layout behind this
// function — bit-identical, measured slower, retired in round 5: DESIGN_HISTORY.md.  `lut` is that layout's parameter.)
template <int LAYOUT>
F110_HD double table_fetch(const ScanConst &k, int r, int c, const double *lut)
{
    static_assert(LAYOUT == LAYOUT_ROWMAJOR || LAYOUT == LAYOUT_PADDED, "row-major or padded");
    (void)lut;
    // 32-bit BYTE offset from the (wave-uniform) table base: lets the compiler use the
    // scalar-base + 32-bit-VGPR-offset form of global_load (tables are < 4 GiB, checked on upload).
    // PADDED: k.table is the plain table here (the exact path)
    const uint32_t off = mul24((uint32_t)r, (uint32_t)k.row_bytes) + ((uint32_t)c << 3);
    return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(k.table) + off);
}

// int(v / resolution) of xy_2_rc :83-84 for v in [0, extent).  When the resolution is a power
// of two the reciprocal multiply is exact; otherwise the multiply decides unless the quotient
// lands within 1e-9 of a cell boundary, where the true IEEE division is evaluated.
template <bool POW2>
F110_HD int cell_index(double v, const ScanConst &k)
{
    const double q = v * k.inv_res;
    int ci = (int)q;
    if (!POW2) {
        const double fr = q - (double)ci;
        if (fr < 1e-9 || fr > 1.0 - 1e-9) ci = (int)(v / k.res);
    }
    return ci;
}

// xy_2_rc :55-86 + distance_transform :88-104.  rc = (-1,-1) when out of bounds.
template <int LAYOUT, bool POW2, bool IDENT>
F110_HD double sample_distance(const ScanConst &k, const double *lut, double x, double y, int &r, int &c)
{
    const double xt = x - k.orig_x;
    const double yt = y - k.orig_y;
    double xr, yr;
    if (IDENT) {  // origin yaw == 0: x_trans*1 + y_trans*0 == x_trans for finite values
        xr = xt;
        yr = yt;
    } else {
        xr = xt * k.orig_c + yt * k.orig_s;
        yr = -xt * k.orig_s + yt * k.orig_c;
    }
    // one combined predicate (bitwise &, no short-circuit) -> a single divergent region per
    // sample.  The reference's "x_rot < 0 or x_rot >= w*res or ..." (:79) is its complement
    // for every non-NaN position.
    const bool inside = (xr >= 0) & (xr < k.w_res) & (yr >= 0) & (yr < k.h_res);
    double d = k.oob_value;
    r = -1;
    c = -1;
    if (inside) {
        c = cell_index<POW2>(xr, k);
        r = cell_index<POW2>(yr, k);
        d = table_fetch<LAYOUT>(k, r, c, lut);
    }
    return d;
}

// trace_ray :133-146 — the marching loop, entered with the first sample (:129-130) already taken.
// Every ray of one scan starts at the same lidar position, so that first sample is shared by
// all beams of an agent: the step kernel receives it from k_integrate instead of gathering it
// once per ray.
template <int LAYOUT, bool POW2, bool IDENT>
F110_HD double march_from_first(const ScanConst &k, const double *lut, double x, double y, double c, double s,
                                double d, int &hit_r, int &hit_c, int &lookups)
{
    double total = d;
    int n = 1;
    while (d > k.eps && total <= k.max_range) {
        x += d * c;
        y += d * s;
        d = sample_distance<LAYOUT, POW2, IDENT>(k, lut, x, y, hit_r, hit_c);
        total += d;
        ++n;
    }
    lookups = n;
    return (total > k.max_range) ? k.max_range : total;
}

// trace_ray :106-146 (sphere tracing over the distance table)
template <int LAYOUT, bool POW2, bool IDENT>
F110_HD double march_ray(const ScanConst &k, const double *lut, double x, double y, double c, double s,
                         int &hit_r, int &hit_c, int &lookups)
{
    const double d = sample_distance<LAYOUT, POW2, IDENT>(k, lut, x, y, hit_r, hit_c);
    return march_from_first<LAYOUT, POW2, IDENT>(k, lut, x, y, c, s, d, hit_r, hit_c, lookups);
}

// ---- PADDED fast path ---------------------------------------------------------------------
// world position -> padded-table cell units (xy_2_rc :68-77 then / resolution, + border), with
// the rotation, the 1/resolution and the offsets folded into six constants (setup_padded)
template <bool IDENT>
F110_HD void padded_position(const ScanConst &k, double x, double y, double &ux, double &uy)
{
    if (IDENT) {
        ux = fma(x, k.inv_res, k.pad_cx);
        uy = fma(y, k.inv_res, k.pad_cy);
    } else {
        ux = fma(x, k.pad_axx, fma(y, k.pad_axy, k.pad_cx));
        uy = fma(y, k.pad_ayy, fma(x, k.pad_ayx, k.pad_cy));
    }
}

// ray direction -> padded cells advanced per metre of range
template <bool IDENT>
F110_HD void padded_rate(const ScanConst &k, double c, double s, double &cux, double &cuy)
{
    if (IDENT) {
        cux = c * k.inv_res;
        cuy = s * k.inv_res;
    } else {
        cux = fma(c, k.pad_axx, s * k.pad_axy);
        cuy = fma(s, k.pad_ayy, c * k.pad_ayx);
    }
}

// true when every sample of every ray from this lidar position lands inside the padded table
// (false for NaN): samples are taken at most max_range (+ rounding) from the lidar
F110_HD bool padded_start_ok(const ScanConst &k, double ux, double uy)
{
    return (ux >= k.pad_lo) & (ux <= k.pad_hi_x) & (uy >= k.pad_lo) & (uy <= k.pad_hi_y);
}

// trace_ray :133-146 on the padded table, entered with the first sample d taken.  (ux, uy) is
// the lidar in padded cell units and (cux, cuy) the cells advanced per metre, so sample n is at
// u_n = u_0 + (d_0 + ... + d_{n-1}) * cu: the same point as the reference's x_n of :140-141 up to
// rounding, |u_n - reference| < kPadGuard / 4 for n <= k.pad_max_samples (setup_padded).
// Returns false when the ray must be re-marched with the reference's arithmetic: a sample within
// kPadGuard of a cell boundary, or more samples than the error bound covers (about one ray in
// 10^7); the outputs are then meaningless.
// r/c: cell of the last sample in the reference's convention (-1,-1 out of bounds); untouched when
// the loop takes no sample.
// (Rounds 5-6 also marched a 4x4-tiled copy and a row-pair copy — 2 rows x 8 cells per 128-byte line — of this table behind this
// function: bit-identical, fewer distinct lines per 64-ray gather (11.4 -> 8.6 / 9.1), paid for in integer operations per sample
// (+6 / +2); tiles lost 2-3 % at every size, row pairs were neutral from 8192 agents up and -2.5 % below 2048 agents.  Retired in
// round 6 under the pre-registered stop rule: DESIGN.md §8, profiles/r05_tiled_table.txt, profiles/r06_rowpair*.txt.)
template <bool WANT_CELL>
F110_HD bool march_padded(const ScanConst &k, double ux, double uy, double cux, double cuy, double d, double &range,
                          int &hit_r, int &hit_c, int &lookups)
{
    double total = d;
    int n = 1;
    bool redo = false;
    const char *base = reinterpret_cast<const char *>(k.pad);
    while ((d > k.eps) & (total <= k.max_range) & !redo) {
        ux = fma(d, cux, ux);
        uy = fma(d, cuy, uy);
        const uint32_t wx = low_word(ux + kFixBig);
        const uint32_t wy = low_word(uy + kFixBig);
        uint32_t off = mul24(wy >> kFixFracBits, (uint32_t)k.pad_row_bytes) + ((wx >> kFixFracBits) << 3);
        if (WANT_CELL) {
            hit_c = (int)(wx >> kFixFracBits);
            hit_r = (int)(wy >> kFixFracBits);
        }
        if (((wx & 0xffffu) == 0u) | ((wy & 0xffffu) == 0u)) {
            // within 2^-17 of a cell boundary, where the word (rounded to nearest) may name the
            // cell above: take the floor, and give the ray up if it is closer than kPadGuard
            redo = (fabs(ux - rint(ux)) < kPadGuard) | (fabs(uy - rint(uy)) < kPadGuard);
            const int fc = (int)floor(ux), fr = (int)floor(uy);
            off = mul24((uint32_t)fr, (uint32_t)k.pad_row_bytes) + ((uint32_t)fc << 3);
            if (WANT_CELL) {
                hit_c = fc;
                hit_r = fr;
            }
        }
        d = *reinterpret_cast<const double *>(base + off);
        total += d;
        ++n;
    }
    if (WANT_CELL && n > 1) {
        hit_c -= k.pad_border;
        hit_r -= k.pad_border;
        if (hit_c < 0 || hit_c >= k.width || hit_r < 0 || hit_r >= k.height) {
            hit_r = -1;
            hit_c = -1;
        }
    }
    lookups = n;
    range = (total > k.max_range) ? k.max_range : total;
    return !(redo | (n > k.pad_max_samples));
}

// (Round 5 also had march_padded_spec here — the tail of a long ray two samples per memory round trip where the table value
// repeats, bit-identical — measured slower at every setting and retired in round 6: DESIGN.md §8, profiles/r05_spec_march.txt.)

// The exact march for the rays march_padded gives up on, written to keep the fast kernel's
// register footprint: its constants are fetched from the HBM copy of ScanConst when (if ever) it
// runs, and it uses the reference's own arithmetic (true divisions, width*resolution formed from
// the integers as laser_models.py:79 does) on the plain row-major table.
template <bool IDENT>
F110_HD double march_exact_cold(const ScanConst *kc_in, double x, double y, double c, double s, double d, int &hit_r,
                                int &hit_c, int &lookups)
{
    // volatile: every constant is re-read where it is used instead of living in a register for the
    // whole loop; this path runs for about one ray in 10^7, its footprint matters, its speed does not
    const volatile ScanConst *kc = kc_in;
    double total = d;
    int n = 1;
    while (d > kc->eps && total <= kc->max_range) {
        x += d * c;
        y += d * s;
        const double xt = x - kc->orig_x, yt = y - kc->orig_y;
        double xr = xt, yr = yt;
        if (!IDENT) {
            const double oc = kc->orig_c, os = kc->orig_s;
            xr = xt * oc + yt * os;
            yr = -xt * os + yt * oc;
        }
        const double res = kc->res;
        hit_r = -1;
        hit_c = -1;
        d = kc->oob_value;
        if ((xr >= 0) & (xr < (double)kc->width * res) & (yr >= 0) & (yr < (double)kc->height * res)) {
            hit_c = (int)(xr / res);
            hit_r = (int)(yr / res);
            const char *base = reinterpret_cast<const char *>(kc->table_rm);
            d = *reinterpret_cast<const double *>(base + (mul24((uint32_t)hit_r, (uint32_t)kc->row_bytes) + ((uint32_t)hit_c << 3)));
        }
        total += d;
        ++n;
    }
    lookups = n;
    const double max_range = kc->max_range;
    return (total > max_range) ? max_range : total;
}

// get_scan :166-172
F110_HD double scan_start_index(const ScanConst &k, double pose_theta)
{
    double ti = k.theta_dis * (pose_theta - k.fov / 2.) / (2. * kPi);
    // fmod(x, y) == x whenever |x| < |y| (exactly): with the yaw wrapped to [0, 2 pi] that is every call, and the
    // library fmod is a loop of a few hundred dependent instructions on k_integrate's chain
    if (!(fabs(ti) < (double)k.theta_dis)) ti = fmod(ti, (double)k.theta_dis);
    while (ti < 0) ti += k.theta_dis;
    return ti;
}

// int(theta_index) of beam i (:124 with the running index of :177-184).  The reference adds
// theta_index_increment i times in floating point; start + i*inc reproduces that to ~1e-10, so
// the truncation is decided in closed form unless the value is within dir_guard of an integer,
// in which case the sequential additions are replayed exactly.
F110_HD int beam_dir_index(const ScanConst &k, double start, int i)
{
    const double td = (double)k.theta_dis;
    // closed form (approximate on purpose: explicit FMAs, reciprocal instead of a division);
    // any value that lands within dir_guard of an integer — which includes the wrap points 0 and
    // theta_dis — is recomputed exactly below
    double t = fma((double)i, k.theta_inc, start);
    t = fma(-floor(t * k.inv_theta_dis), td, t);
    const double fr = t - floor(t);
    if (!(fabs(fr - 0.5) < 0.5 - k.dir_guard)) {
        double ti = start;
        for (int j = 0; j < i; ++j) {
            ti += k.theta_inc;
            while (ti >= td) ti -= td;
        }
        t = ti;
    }
    int idx = (int)t;
    // theta_index == theta_dis can only arise from start + theta_dis rounding (:172); the
    // reference would then index one past the table — clamp instead of reading out of bounds.
    return idx >= k.theta_dis ? k.theta_dis - 1 : idx;
}

// check_ttc_jit :188-217 — one beam's predicate
F110_HD bool ttc_beam_hit(double range, double side_distance, double vel, double beam_cos, double thresh)
{
    // ttc = (scan[i] - side_distances[i]) / (vel*cosines[i]);  hit <=> 0 <= ttc < thresh.
    // Decided without the float64 division unless |num| is within 1e-12 (relative) of
    // thresh*|den| — there, and for NaN / 0/0, the reference expression itself is evaluated.
    const double proj_vel = vel * beam_cos;
    const double num = range - side_distance;
    const double a = fabs(num), b = fabs(proj_vel);
    if (a < (thresh * (1.0 - 1e-12)) * b) return (num == 0.0) || ((num > 0.0) == (proj_vel > 0.0));
    if (a > (thresh * (1.0 + 1e-12)) * b) return false;
    const double ttc = num / proj_vel;
    return (ttc < thresh) && (ttc >= 0.0);
}

// get_range :249-280 with v3 = (cos, sin)(beam_theta + pi/2) supplied by the caller
F110_HD double edge_range(double ox, double oy, double v3x, double v3y, double vax, double vay,
                          double vbx, double vby)
{
    const double v1x = ox - vax, v1y = oy - vay;
    const double v2x = vbx - vax, v2y = vby - vay;
    const double denom = v2x * v3x + v2y * v3y;
    double distance = INFINITY;
    if (fabs(denom) > 0.0) {
        // (round 4 tried deciding hit / miss on the numerators and dividing only for an edge that is hit — bit-identical,
        // tests/test_host_math.py keeps the equivalence test — and lost at every size: 65 536 x 2 agents 95.7 -> 94.6 M
        // agent-steps/s, 16 cars per env 70.6 -> 64.2: the divisions pipeline across the four edges, the branches do not)
        const double d1 = (v2x * v1y - v2y * v1x) / denom;
        const double d2 = (v1x * v3x + v1y * v3y) / denom;
        if (d1 >= 0.0 && d2 >= 0.0 && d2 <= 1.0) distance = d1;
    } else {
        // are_collinear(o, va, vb) :232-247
        const double bax = vax - ox, bay = vay - oy;
        const double cax = ox - vbx, cay = oy - vby;
        if (fabs(bax * cay - bay * cax) < 1e-8) {
            const double da = sqrt(bax * bax + bay * bay);
            const double dbx = vbx - ox, dby = vby - oy;
            const double db = sqrt(dbx * dbx + dby * dby);
            distance = da < db ? da : db;
        }
    }
    return distance;
}

// first index minimising |scan_angles[i] - a| (np.argmin, :310-313).  scan_angles is strictly
// increasing (base_classes.py:133-134), so the minimum sits next to the closed-form estimate;
// a +-3 neighbourhood is scanned with the reference's own comparison.
F110_HD int nearest_beam(const double *scan_angles, int num_beams, double angle_inc, double a)
{
    double est = (a - scan_angles[0]) / angle_inc;
    int i0 = est < 0 ? 0 : (est > (double)(num_beams - 1) ? num_beams - 1 : (int)est);
    int lo = i0 - 3 < 0 ? 0 : i0 - 3;
    int hi = i0 + 3 > num_beams - 1 ? num_beams - 1 : i0 + 3;
    int best = lo;
    double bv = fabs(scan_angles[lo] - a);
    for (int i = lo + 1; i <= hi; ++i) {
        const double v = fabs(scan_angles[i] - a);
        if (v < bv) {
            bv = v;
            best = i;
        }
    }
    return best;
}

// np.argmin(np.abs(scan_angles - a)) as the reference evaluates it (:310-313): every entry, first minimum wins (a NaN entry
// wins over every number, as in NumPy).  For tables that are not the uniform ramp of base_classes.py:133-134 — the free
// functions ray_cast / check_ttc_jit accept any scan_angles array (unit entry points only: f110_raycast_batch).
F110_HD int nearest_beam_full(const double *scan_angles, int num_beams, double a)
{
    int best = 0;
    double bv = fabs(scan_angles[0] - a);
    for (int i = 1; i < num_beams; ++i) {
        const double v = fabs(scan_angles[i] - a);
        if (v < bv || (v != v && bv == bv)) {
            bv = v;
            best = i;
        }
    }
    return best;
}

// the angle get_blocked_view_indices looks up for one vertex (:296-309): -(heading - direction), wrapped once
F110_HD double vertex_view_angle(double ex, double ey, double etheta, double vx, double vy)
{
    const double dx = vx - ex, dy = vy - ey;
    const double norm = sqrt(dx * dx + dy * dy);
    const double ux = dx / norm, uy = dy / norm;
    double ce, se;
    cos_sin(etheta, ce, se);
    double angle = atan2(se, ce) - atan2(uy, ux);
    if (angle > kPi)
        angle = angle - 2 * kPi;
    else if (angle < -kPi)
        angle = angle + 2 * kPi;
    return -angle;
}

// one vertex's beam index of get_blocked_view_indices :282-315
// get_blocked_view_indices :296-311 for one vertex, with its two arc tangents supplied:
// head = atan2(sin(etheta), cos(etheta)), dir = atan2(uy, ux) of the normalised lidar -> vertex vector
F110_HD int vertex_beam_from_angles(double head, double dir, const double *scan_angles, int num_beams, double angle_inc)
{
    double angle = head - dir;
    if (angle > kPi)
        angle = angle - 2 * kPi;
    else if (angle < -kPi)
        angle = angle + 2 * kPi;
    return nearest_beam(scan_angles, num_beams, angle_inc, -angle);
}

F110_HD int vertex_beam_index(double ex, double ey, double etheta, double vx, double vy,
                              const double *scan_angles, int num_beams, double angle_inc)
{
    const double dx = vx - ex, dy = vy - ey;
    const double norm = sqrt(dx * dx + dy * dy);
    const double ux = dx / norm, uy = dy / norm;
    double ce, se;
    cos_sin(etheta, ce, se);
    return vertex_beam_from_angles(atan2(se, ce), atan2(uy, ux), scan_angles, num_beams, angle_inc);
}

// Conservative beam-index range whose rays can touch a disc (centre c, radius R) seen from the
// ego at (ex, ey, eth).  ray_cast (:338-345) evaluates every beam of [min_ind, max_ind] against
// the opponent's four edges, but a beam whose ray misses the box's circumscribed disc returns
// inf from all four get_range calls and leaves the scan untouched — skipping it is
// result-preserving.  (When the opponent straddles the rear +-pi direction the reference window
// degenerates to all B beams although none of them can hit: this cull removes that work.)
// Beams are sa[0] + b*inc to within rounding; the range is padded by 3 beams + 1e-6 rad.
// disc_beam_range with dist = |centre - lidar|, dir = atan2(dy, dx) and head = atan2(sin(eth), cos(eth)) supplied
F110_HD void disc_beam_range_from(double dist, double eth, double dir, double head, double R, const double *scan_angles, int num_beams,
                                  double angle_inc, int &cl, int &ch)
{
    // No cull when the lidar is inside / on the disc (or anything is NaN), and none when the
    // heading is astronomically large: the reference wraps yaw by a single 2*pi per step
    // (base_classes.py:400-404), so a diverged yaw rate leaves |yaw| ~ 1e15, where
    // pose[2] + scan_angles[i] (:343) is rounded to a grid coarser than the beam spacing and the
    // beams no longer point where their index says.  Below 1e6 that rounding is < 1e-10 rad.
    if (!(dist > R * 1.000001 + 1e-9) || !(fabs(eth) < 1e6)) {
        cl = 0;
        ch = num_beams - 1;
        return;
    }
    double phi = dir - head;
    phi -= kTwoPi * rint(phi / kTwoPi);  // (-pi, pi]
    const double psi = asin(R / dist) + 3.0 * angle_inc + 1e-6;
    const double sa0 = scan_angles[0];
    const double last = (double)(num_beams - 1);
    cl = num_beams;
    ch = -1;
#pragma unroll
    for (int k = -1; k <= 1; ++k) {
        double lo = ceil((phi + kTwoPi * k - psi - sa0) / angle_inc);
        double hi = floor((phi + kTwoPi * k + psi - sa0) / angle_inc);
        lo = lo < 0.0 ? 0.0 : lo;
        hi = hi > last ? last : hi;
        if (lo <= hi) {
            cl = (int)lo < cl ? (int)lo : cl;
            ch = (int)hi > ch ? (int)hi : ch;
        }
    }
}

F110_HD void disc_beam_range(double ex, double ey, double eth, double cx, double cy, double R,
                             const double *scan_angles, int num_beams, double angle_inc, int &cl, int &ch)
{
    const double dx = cx - ex, dy = cy - ey;
    const double dist = sqrt(dx * dx + dy * dy);
    if (!(dist > R * 1.000001 + 1e-9) || !(fabs(eth) < 1e6)) {   // before any trigonometry, as it always was
        cl = 0;
        ch = num_beams - 1;
        return;
    }
    double ce, se;
    cos_sin(eth, ce, se);
    disc_beam_range_from(dist, eth, atan2(dy, dx), atan2(se, ce), R, scan_angles, num_beams, angle_inc, cl, ch);
}

// get_blocked_view_indices :282-315 (min/max of the four vertex beam indices) intersected with
// the disc cull above.  Returns an empty range (hi < lo) when no beam of the window can hit.
F110_HD void opponent_beam_window(double ex, double ey, double eth, const double *v, double cx, double cy,
                                  double R, const double *scan_angles, int num_beams, double angle_inc,
                                  int &ref_lo, int &ref_hi, int &lo, int &hi)
{
    const int i0 = vertex_beam_index(ex, ey, eth, v[0], v[1], scan_angles, num_beams, angle_inc);
    const int i1 = vertex_beam_index(ex, ey, eth, v[2], v[3], scan_angles, num_beams, angle_inc);
    const int i2 = vertex_beam_index(ex, ey, eth, v[4], v[5], scan_angles, num_beams, angle_inc);
    const int i3 = vertex_beam_index(ex, ey, eth, v[6], v[7], scan_angles, num_beams, angle_inc);
    int a = i0 < i1 ? i0 : i1, b = i2 < i3 ? i2 : i3;
    ref_lo = a < b ? a : b;
    a = i0 > i1 ? i0 : i1;
    b = i2 > i3 ? i2 : i3;
    ref_hi = a > b ? a : b;
    int cl, ch;
    disc_beam_range(ex, ey, eth, cx, cy, R, scan_angles, num_beams, angle_inc, cl, ch);
    lo = ref_lo > cl ? ref_lo : cl;
    hi = ref_hi < ch ? ref_hi : ch;
}

// the four get_range calls of ray_cast's inner loop (:341-345) for one beam
F110_HD double box_range(double ex, double ey, double v3x, double v3y, const double *v, double r)
{
    double rr = edge_range(ex, ey, v3x, v3y, v[0], v[1], v[2], v[3]);
    if (rr < r) r = rr;
    rr = edge_range(ex, ey, v3x, v3y, v[2], v[3], v[4], v[5]);
    if (rr < r) r = rr;
    rr = edge_range(ex, ey, v3x, v3y, v[4], v[5], v[6], v[7]);
    if (rr < r) r = rr;
    rr = edge_range(ex, ey, v3x, v3y, v[6], v[7], v[0], v[1]);
    if (rr < r) r = rr;
    return r;
}

// ------------------------------------------------------------------ collision_models.py
// get_vertices :218-260 — order [rl, rr, fr, fl]; v[2*i], v[2*i+1]
F110_HD void box_vertices(double x, double y, double th, double length, double width, double *v)
{
    double c, s;
    cos_sin(th, c, s);
    const double hx = length / 2, hy = width / 2;
    const double bx[4] = {-hx, -hx, hx, hx};
    const double by[4] = {hy, -hy, -hy, hy};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = ((c * bx[i] + (-s) * by[i]) + 0.) + x;
        v[2 * i + 1] = ((s * bx[i] + c * by[i]) + 0.) + y;
    }
}

// the same with cos / sin of the heading supplied (computed once per agent by the caller)
F110_HD void box_vertices_cs(double x, double y, double c, double s, double length, double width, double *v)
{
    const double hx = length / 2, hy = width / 2;
    const double bx[4] = {-hx, -hx, hx, hx};
    const double by[4] = {hy, -hy, -hy, hy};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = ((c * bx[i] + (-s) * by[i]) + 0.) + x;
        v[2 * i + 1] = ((s * bx[i] + c * by[i]) + 0.) + y;
    }
}

F110_HD int furthest_vertex(const double *v, double dx, double dy)
{
    int best = 0;
    double bv = v[0] * dx + v[1] * dy;
#pragma unroll
    for (int i = 1; i < 4; ++i) {
        const double val = v[2 * i] * dx + v[2 * i + 1] * dy;
        if (val > bv) {
            bv = val;
            best = i;
        }
    }
    return best;
}

F110_HD void minkowski_support(const double *v1, const double *v2, double dx, double dy, double &ax, double &ay)
{
    const int i = furthest_vertex(v1, dx, dy);
    const int j = furthest_vertex(v2, -dx, -dy);
    ax = v1[2 * i] - v2[2 * j];
    ay = v1[2 * i + 1] - v2[2 * j + 1];
}

// tripleProduct(a, b, c) = b*(a.c) - a*(b.c)  :51-64
F110_HD void triple_product(double ax, double ay, double bx, double by, double cx, double cy, double &ox, double &oy)
{
    const double ac = ax * cx + ay * cy;
    const double bc = bx * cx + by * cy;
    ox = bx * ac - ax * bc;
    oy = by * ac - ay * bc;
}

// collision (GJK) :113-182 on two 4-vertex convex bodies
F110_HD bool gjk_overlap(const double *v1, const double *v2)
{
    // simplex slots kept in scalars (no runtime-indexed arrays -> no scratch on gfx950)
    double s0x, s0y, s1x = 0, s1y = 0;
    double dx = (((v1[0] + v1[2]) + v1[4]) + v1[6]) / 4 - (((v2[0] + v2[2]) + v2[4]) + v2[6]) / 4;
    double dy = (((v1[1] + v1[3]) + v1[5]) + v1[7]) / 4 - (((v2[1] + v2[3]) + v2[5]) + v2[7]) / 4;
    if (dx == 0 && dy == 0) dx = 1.0;
    double ax, ay;
    minkowski_support(v1, v2, dx, dy, ax, ay);
    s0x = ax;
    s0y = ay;
    if (dx * ax + dy * ay <= 0) return false;
    dx = -ax;
    dy = -ay;
    int index = 0;
    for (int iter = 0; iter < 1000;) {
        minkowski_support(v1, v2, dx, dy, ax, ay);
        index += 1;
        if (dx * ax + dy * ay <= 0) return false;
        const double aox = -ax, aoy = -ay;
        if (index < 2) {
            // line case: new point becomes simplex[1]
            s1x = ax;
            s1y = ay;
            const double abx = s0x - ax, aby = s0y - ay;
            triple_product(abx, aby, aox, aoy, abx, aby, dx, dy);
            if (sqrt(dx * dx + dy * dy) < 1e-10) {  // perpendicular(ab) :34-48
                dx = aby;
                dy = -1 * abx;
            }
            continue;
        }
        // triangle case: a = simplex[2], b = simplex[1], c = simplex[0]
        const double abx = s1x - ax, aby = s1y - ay;
        const double acx = s0x - ax, acy = s0y - ay;
        double px, py;
        triple_product(abx, aby, acx, acy, acx, acy, px, py);  // acperp
        if (px * aox + py * aoy >= 0) {
            dx = px;
            dy = py;
        } else {
            triple_product(acx, acy, abx, aby, abx, aby, px, py);  // abperp
            if (px * aox + py * aoy < 0) return true;
            s0x = s1x;
            s0y = s1y;
            dx = px;
            dy = py;
        }
        s1x = ax;
        s1y = ay;
        index -= 1;
        ++iter;
    }
    return false;
}

// ------------------------------------------------------------------ examples/waypoint_follow.py
// The reference's example policy (PurePursuitPlanner), so a closed loop can stay on the GPU.
// waypoints [M][3] = (x, y, speed).

// nearest_point_on_trajectory :15-50: projection on every segment, parameter clipped to [0,1],
// first segment with the smallest distance (np.argmin)
F110_HD int nearest_on_trajectory(const double *wp, int M, double px, double py, double &dist, double &t_best)
{
    int best = 0;
    dist = INFINITY;
    t_best = 0.0;
    for (int k = 0; k + 1 < M; ++k) {
        const double ax = wp[3 * k], ay = wp[3 * k + 1];
        const double dx = wp[3 * k + 3] - ax, dy = wp[3 * k + 4] - ay;
        double t = ((px - ax) * dx + (py - ay) * dy) / (dx * dx + dy * dy);
        t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
        const double rx = px - (ax + t * dx), ry = py - (ay + t * dy);
        const double d = sqrt(rx * rx + ry * ry);
        if (d < dist) {
            dist = d;
            t_best = t;
            best = k;
        }
    }
    return best;
}

// does the look-ahead circle cut the segment s -> e (+1e-6 on e, :70)?  In the segment the search
// starts in only parameters >= t_min count (:86-96); elsewhere t_min = 0.
F110_HD bool lookahead_cuts(double sx, double sy, double ex, double ey, double px, double py, double radius, double t_min)
{
    const double vx = (ex + 1e-6) - sx, vy = (ey + 1e-6) - sy;
    const double a = vx * vx + vy * vy;
    const double b = 2.0 * (vx * (sx - px) + vy * (sy - py));
    const double c = (sx * sx + sy * sy) + (px * px + py * py) - 2.0 * (sx * px + sy * py) - radius * radius;
    const double disc = b * b - 4 * a * c;
    if (disc < 0) return false;
    const double root = sqrt(disc);
    const double t1 = (-b - root) / (2.0 * a), t2 = (-b + root) / (2.0 * a);
    return (t1 >= 0.0 && t1 <= 1.0 && t1 >= t_min) || (t2 >= 0.0 && t2 <= 1.0 && t2 >= t_min);
}

// first_point_on_trajectory_intersecting_circle :52-132 (wrap=True) -> index of the waypoint the
// planner then steers to (the segment's first waypoint; M-1 for the closing segment, which the
// reference reaches as wpts[-1]), or -1
F110_HD int first_waypoint_on_circle(const double *wp, int M, double px, double py, double radius, double start)
{
    const int start_i = (int)start;
    const double start_t = fmod(start, 1.0);
    for (int i = start_i; i + 1 < M; ++i)
        if (lookahead_cuts(wp[3 * i], wp[3 * i + 1], wp[3 * i + 3], wp[3 * i + 4], px, py, radius, i == start_i ? start_t : 0.0))
            return i;
    for (int i = -1; i < start_i; ++i) {
        const int k0 = i < 0 ? M - 1 : i, k1 = i + 1;   // i in [-1, M-2]: Python's (i % M), ((i+1) % M)
        if (lookahead_cuts(wp[3 * k0], wp[3 * k0 + 1], wp[3 * k1], wp[3 * k1 + 1], px, py, radius, 0.0)) return k0;
    }
    return -1;
}

// PurePursuitPlanner.plan :203-217 with _get_current_waypoint :183-201 and get_actuation :134-145
F110_HD void pure_pursuit_plan(const double *wp, int M, double px, double py, double theta, double lookahead, double vgain,
                               double wheelbase, double max_reacquire, double &steer, double &speed)
{
    steer = 0.0;
    speed = 4.0;  // "no waypoint" action :213-214
    double dist, t;
    const int i = nearest_on_trajectory(wp, M, px, py, dist, t);
    int goal = i;
    if (dist < lookahead) {
        goal = first_waypoint_on_circle(wp, M, px, py, lookahead, (double)i + t);
        if (goal < 0) return;
    } else if (!(dist < max_reacquire)) {
        return;
    }
    // steer to the WAYPOINT `goal` (:193) at the speed of the NEAREST index (:195)
    const double wy = sin(-theta) * (wp[3 * goal] - px) + cos(-theta) * (wp[3 * goal + 1] - py);
    speed = vgain * wp[3 * i + 2];
    if (fabs(wy) < 1e-6) return;
    const double radius = 1 / (2.0 * wy / (lookahead * lookahead));
    steer = atan(wheelbase / radius);
}

}  // namespace f110
