// f110_hip.hip — gfx950 kernels + the C ABI (include/f110.h) of the batched F1TENTH hot path.
//
// One handle owns one MI355X and one HIP stream.  A step is four launches on that stream:
//   k_integrate   1 lane / agent   pid + steering delay + RK4|Euler + yaw wrap + lidar pose
//                                  (base_classes.py:256-409), SoA columns, coalesced
//   k_collide     1 lane / agent   get_vertices + GJK against the env's other agents
//                                  (collision_models.py:113-260, base_classes.py:536-550)
//   k_scan_rays   1 lane / ray, rays numbered agent*B + beam so a wave holds 64 consecutive
//                                  beams: sphere-trace the distance table (laser_models.py:106-186),
//                                  add the noise row (:450-452), evaluate the iTTC predicate
//                                  (:188-217) and write the range once, coalesced.  Kept free of
//                                  everything else so it runs at 8 waves/SIMD.
//   k_finalize    1 wave / agent   wall-hit state zeroing (base_classes.py:246-249), collision
//                                  flags (:588-589), opponent ray-cast on the beams each opponent
//                                  blocks (laser_models.py:282-346).
// Compiled with -ffp-contract=off: float64, reference operation order, no FMA contraction.
// There is no CPU fallback in this library.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/f110.h"
#include "f110_math.hpp"

using namespace f110;

// ============================================================================ device side

// What one ray needs to know about its agent, written by k_integrate: read through the scalar
// cache by k_scan_rays (a wave's 64 consecutive rays belong to at most two agents).
struct RayHdr {
    double x, y;        // lidar position (base_classes.py:407-408)
    double start;       // wrapped theta_index of beam 0 (laser_models.py:166-172)
    double vel;         // post-integration longitudinal velocity (iTTC)
    double d0;          // first table sample, shared by every beam of the scan (:129)
    int32_t noise_row;  // row of the noise table for this step, -1 = no noise
    int32_t hr0, hc0;   // cell of that first sample
    int32_t i0;         // table index of beam 0
    int32_t n_dirs;     // distinct table indices the scan's beams use (dedupe mode), else 0
    int32_t pad;
};
static_assert(sizeof(RayHdr) == 64, "RayHdr is read as four 16-byte scalar loads");

struct AgentArrays {
    int32_t n_agents_total;  // N
    int32_t agents_per_env;  // A
    double *state;           // [7][N]
    double *steer_buf;       // [2][N]
    int32_t *buf_cnt;        // [N]
    double *scan_pose;       // [3][N]  lidar pose after integration
    double *snap_pose;       // [3][N]  Simulator.agent_poses (:574)
    double *dir_start;       // [N]     wrapped theta_index of beam 0
    RayHdr *ray_hdr;         // [N]     per-agent constants of this step's scan
    double *scans;           // [N][B]
    double *collisions;      // [N]
    double *collision_idx;   // [N]
    int32_t *in_collision;   // [N]
    int32_t *step_count;     // [N]
    int32_t *opp_window;     // [N][A][4] beam range each opponent can occupy: {lo, hi} for the live
                             //           heading and {lo0, hi0} for heading 0 (after a wall hit)
    double *opp_verts;       // [N][A][8] the opponent's box drawn with the ego's length/width
    const double *params;    // [A][18]
    const double *noise;     // [noise_rows][B] or nullptr
    const double *scan_angles, *beam_cos, *side_dist;  // [B]
    int32_t noise_rows, integrator;
    double time_step, lidar_dist, ttc_thresh, angle_inc;
    double box_length, box_width;  // Simulator.params used by check_collision (:549)
};

__device__ __forceinline__ VehicleParams load_params(const double *p)
{
    VehicleParams vp;
#pragma unroll
    for (int i = 0; i < NPARAMS; ++i) vp.v[i] = p[i];
    return vp;
}

// ---- K1: integrate every agent one time step ------------------------------------------
__global__ void __launch_bounds__(256) k_integrate(AgentArrays a, ScanConst k, const double *__restrict__ actions)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = a.n_agents_total;
    if (i >= N) return;
    const VehicleParams vp = load_params(a.params + (size_t)(i % a.agents_per_env) * NPARAMS);
    double st[7];
#pragma unroll
    for (int c = 0; c < 7; ++c) st[c] = a.state[(size_t)c * N + i];
    double b0 = a.steer_buf[i], b1 = a.steer_buf[(size_t)N + i];
    int cnt = a.buf_cnt[i];
    const double2 act = reinterpret_cast<const double2 *>(actions)[i];
    double sp[3];
    advance_vehicle(st, b0, b1, cnt, act.x, act.y, vp, a.time_step, a.integrator, a.lidar_dist, sp);
#pragma unroll
    for (int c = 0; c < 7; ++c) a.state[(size_t)c * N + i] = st[c];
    a.steer_buf[i] = b0;
    a.steer_buf[(size_t)N + i] = b1;
    a.buf_cnt[i] = cnt;
    a.scan_pose[i] = sp[0];
    a.scan_pose[(size_t)N + i] = sp[1];
    a.scan_pose[2 * (size_t)N + i] = sp[2];
    a.snap_pose[i] = st[0];
    a.snap_pose[(size_t)N + i] = st[1];
    a.snap_pose[2 * (size_t)N + i] = st[4];
    const double start = scan_start_index(k, sp[2]);
    a.dir_start[i] = start;
    {
        // everything the ray kernel needs per agent, incl. the first table sample that all
        // beams share (trace_ray :129 evaluated at the lidar position; generic exact path)
        RayHdr hd;
        ScanConst kr = k;
        kr.table = k.table_rm;
        hd.x = sp[0];
        hd.y = sp[1];
        hd.start = start;
        hd.vel = st[3];
        hd.d0 = sample_distance<LAYOUT_ROWMAJOR, false, false>(kr, nullptr, sp[0], sp[1], hd.hr0, hd.hc0);
        int row = -1;
        if (a.noise_rows > 0) {
            row = a.step_count[i];
            if (row >= a.noise_rows) row %= a.noise_rows;
        }
        hd.noise_row = row;
        hd.i0 = beam_dir_index(k, start, 0);
        hd.n_dirs = 0;
        if (k.theta_inc < 1.0) {  // consecutive beams advance the table index by 0 or 1 (mod theta_dis)
            int span = beam_dir_index(k, start, k.num_beams - 1) - hd.i0;
            if (span < 0) span += k.theta_dis;
            hd.n_dirs = span + 1;
        }
        hd.pad = 0;
        a.ray_hdr[i] = hd;
    }
    a.in_collision[i] = 0;  // raised by k_scan_rays when any beam's iTTC is under the threshold
}

// ---- K1b: pairwise body collisions inside each env ------------------------------------
// collision_multiple visits pairs (i<j) ascending and overwrites collision_idx, so an agent's
// final index is the largest colliding partner; flags are symmetric.  Each lane evaluates the
// GJK of its pairs in the reference's (lower, higher) argument order.
__global__ void __launch_bounds__(256) k_collide(AgentArrays a, int32_t B)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = a.n_agents_total, A = a.agents_per_env;
    if (i >= N) return;
    const int env = i / A, me = i - env * A;
    double mine[8];
    const double mx = a.snap_pose[i], my = a.snap_pose[(size_t)N + i], mth = a.snap_pose[2 * (size_t)N + i];
    box_vertices(mx, my, mth, a.box_length, a.box_width, mine);
    // bodies whose centres are further apart than a box diagonal (+1 mm) cannot overlap
    const double reach = sqrt(a.box_length * a.box_length + a.box_width * a.box_width) + 1e-3;
    // RaceCar.ray_cast_agents draws the opponents with the EGO's length/width (:223)
    const double blen = a.params[(size_t)me * NPARAMS + P_LENGTH];
    const double bwid = a.params[(size_t)me * NPARAMS + P_WIDTH];
    const double disc_r = 0.5 * sqrt(blen * blen + bwid * bwid);
    bool hit = false;
    int partner = -1;
    for (int j = 0; j < A; ++j) {
        if (j == me) continue;
        const int o = env * A + j;
        const double ox = a.snap_pose[o], oy = a.snap_pose[(size_t)N + o], oth = a.snap_pose[2 * (size_t)N + o];
        const double dx = ox - mx, dy = oy - my;
        double other[8];
        if (dx * dx + dy * dy <= reach * reach) {
            box_vertices(ox, oy, oth, a.box_length, a.box_width, other);
            const bool c = (me < j) ? gjk_overlap(mine, other) : gjk_overlap(other, mine);
            if (c) {
                hit = true;
                partner = j;  // j ascending -> ends at the largest colliding index
            }
        }
        // beam window this opponent can occupy in my scan: once for my post-integration heading
        // (no wall hit) and once for heading 0 (RaceCar.check_ttc zeroes it on a wall hit, and the
        // ray-cast reads the live state, base_classes.py:225,246-249); k_finalize picks one.
        int ref_lo, ref_hi, lo, hi;
        box_vertices(ox, oy, oth, blen, bwid, other);
        int32_t *win = a.opp_window + ((size_t)i * A + j) * 4;
        opponent_beam_window(mx, my, mth, other, ox, oy, disc_r, a.scan_angles, B, a.angle_inc, ref_lo, ref_hi, lo, hi);
        win[0] = lo;
        win[1] = hi;
        opponent_beam_window(mx, my, 0.0, other, ox, oy, disc_r, a.scan_angles, B, a.angle_inc, ref_lo, ref_hi, lo, hi);
        win[2] = lo;
        win[3] = hi;
        double *ov = a.opp_verts + ((size_t)i * A + j) * 8;
#pragma unroll
        for (int c = 0; c < 8; ++c) ov[c] = other[c];
    }
    a.collisions[i] = hit ? 1.0 : 0.0;
    a.collision_idx[i] = (double)partner;
}

// ---- K2: ray march -------------------------------------------------------------------------
// Rays are numbered ray = pose*B + beam, one lane per ray, so the 64 lanes of a wave are
// consecutive beams of (at most two) poses: neighbouring beams touch neighbouring cells and
// have correlated lengths.  STEP=true is the env.step() form (SoA poses written by
// k_integrate, noise row, iTTC predicate); STEP=false is ScanSimulator2D.scan for the unit
// entry point (also reports terminating cells and lookup counts).
struct RayJob {
    uint32_t n_rays;          // poses * B
    uint32_t n_tasks;         // ceil(n_rays / 64): one task = 64 consecutive rays
    uint32_t tasks_per_wave;  // consecutive tasks each wave walks
    int32_t n_poses;
    uint32_t div_magic, div_shift;  // ray / B == umulhi(ray, magic) >> shift (0: plain division)
    int32_t xcd_remap;
    int32_t dir_mode, dir_stride;   // dedupe pass: rays are (agent, distinct direction), dir_stride per agent
    const double *dir_ranges;       // k_expand_beams: [n_poses][dir_stride] raw ranges of the dedupe pass
    const double *pose_x, *pose_y, *dir_start;  // [n_poses] (unit path)
    double *ranges;           // [n_poses][B]
    // STEP only
    const RayHdr *hdr;        // [n_poses] written by k_integrate
    const double *noise;      // [noise_rows][B] or nullptr
    const double *beam_cos, *side_dist;
    int32_t *wall_flag;       // [n_poses], zeroed by k_integrate
    double ttc_thresh;
    double ttc_side_max, ttc_k;  // r > ttc_side_max + ttc_k*|v| cannot satisfy the iTTC predicate
    // unit only
    int32_t *hit_rc;                 // [n_poses][B][2] or nullptr
    unsigned long long *lookups;     // [n_poses] or nullptr
};

__device__ __forceinline__ int uniform_i32(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double uniform_f64(double v)
{
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// Per-agent ray constants for the lane's agent p.  With >= 64 rays per agent the 64 rays of a wave
// belong to agent p0 (the first lane's) or p0+1, so two uniform 64-byte headers fetched through
// the scalar cache cover the wave and no vector-memory instruction is spent on them; uniform_*()
// pins each field to SGPRs so the compiler keeps two scalar loads + a per-lane select instead of
// one divergent vector load.  Fewer rays per agent: a wave may span more agents -> per-lane load.
struct LaneHdr {
    double x, y, start, vel, d0;
    int row, hr, hc, i0, n_dirs;
};

__device__ __forceinline__ LaneHdr load_lane_hdr(const RayHdr *hdr, uint32_t p, uint32_t n_poses, bool wide)
{
    LaneHdr o;
    if (wide) {
        typedef const __attribute__((address_space(4))) RayHdr *chdr_t;
        const uint32_t p0 = __builtin_amdgcn_readfirstlane(p);
        const uint32_t p1 = (p0 + 1u < n_poses) ? p0 + 1u : p0;
        const chdr_t h0 = (chdr_t)(hdr) + p0;
        const chdr_t h1 = (chdr_t)(hdr) + p1;
        const bool first = (p == p0);
        const double x0 = uniform_f64(h0->x), y0 = uniform_f64(h0->y), s0 = uniform_f64(h0->start);
        const double v0 = uniform_f64(h0->vel), d00 = uniform_f64(h0->d0);
        const int n0 = uniform_i32(h0->noise_row), r0 = uniform_i32(h0->hr0), c0 = uniform_i32(h0->hc0);
        const int i00 = uniform_i32(h0->i0), nd0 = uniform_i32(h0->n_dirs);
        const double x1 = uniform_f64(h1->x), y1 = uniform_f64(h1->y), s1 = uniform_f64(h1->start);
        const double v1 = uniform_f64(h1->vel), d01 = uniform_f64(h1->d0);
        const int n1 = uniform_i32(h1->noise_row), r1 = uniform_i32(h1->hr0), c1 = uniform_i32(h1->hc0);
        const int i01 = uniform_i32(h1->i0), nd1 = uniform_i32(h1->n_dirs);
        o.x = first ? x0 : x1;
        o.y = first ? y0 : y1;
        o.start = first ? s0 : s1;
        o.vel = first ? v0 : v1;
        o.d0 = first ? d00 : d01;
        o.row = first ? n0 : n1;
        o.hr = first ? r0 : r1;
        o.hc = first ? c0 : c1;
        o.i0 = first ? i00 : i01;
        o.n_dirs = first ? nd0 : nd1;
    } else {
        const RayHdr hd = hdr[p];
        o.x = hd.x; o.y = hd.y; o.start = hd.start; o.vel = hd.vel; o.d0 = hd.d0; o.row = hd.noise_row;
        o.hr = hd.hr0; o.hc = hd.hc0; o.i0 = hd.i0; o.n_dirs = hd.n_dirs;
    }
    return o;
}

// noise + iTTC + store for one beam (shared by k_scan_rays and k_expand_beams).
// check_ttc_jit is an any-over-beams: every hitting lane raises the agent's flag.
// r > max(side) + thresh*(1+1e-9)*max|cos|*|v| implies r - side_distances[b] >
// thresh*(1+1e-12)*|v*cosines[b]|, i.e. the "no hit" branch of ttc_beam_hit, so the per-beam
// tables are only read for the few beams that are that close.
struct RayJob;
__device__ __forceinline__ void finish_beam(const RayJob &j, uint32_t B, uint32_t p, int b, uint32_t ray, double r,
                                            int row, double vel);

template <int LAYOUT, bool POW2, bool IDENT, bool STEP>
__global__ void __launch_bounds__(256) k_scan_rays(RayJob j, ScanConst k)
{
    __shared__ double lut_lds[LAYOUT == LAYOUT_CODE8 ? 256 : 1];
    if (LAYOUT == LAYOUT_CODE8) {
        // stage the 2 KB value LUT once per workgroup; every wave then walks j.tasks_per_wave
        // consecutive 64-ray tasks so the fill is amortised
        for (int t = threadIdx.x; t < kLutEntries; t += blockDim.x) lut_lds[t] = k.lut[t];
        __syncthreads();
    }
    const uint32_t B = (STEP && j.dir_mode) ? (uint32_t)j.dir_stride : (uint32_t)k.num_beams;
    const uint32_t tpw = j.tasks_per_wave;
    const uint32_t lane = threadIdx.x & 63u;
    // Workgroup b is dispatched to XCD b % 8 (observed, MI355X_MICROARCH.md).  Re-map so that each
    // XCD walks one contiguous eighth of the rays: all beams of an agent, and agents that are
    // neighbours in the batch, then share one XCD's L2 instead of being spread over all eight.
    uint32_t blk = blockIdx.x;
    if (j.xcd_remap) {
        const uint32_t nb = gridDim.x, q = nb >> 3, rem = nb & 7u, x = blk & 7u, i = blk >> 3;
        blk = (x < rem ? x * (q + 1u) : rem * (q + 1u) + (x - rem) * q) + i;  // bijective for any nb
    }
    const uint32_t wave = (blk * blockDim.x + threadIdx.x) >> 6;
    for (uint32_t t = 0; t < tpw; ++t) {
        const uint32_t task = wave * tpw + t;
        if (task >= j.n_tasks) break;  // wave-uniform
        const uint32_t ray = task * 64u + lane;
        if (ray >= j.n_rays) break;
        const uint32_t p = j.div_magic ? (__umulhi(ray, j.div_magic) >> j.div_shift) : ray / B;
        const int b = (int)(ray - p * B);
        int hr, hc, nl;
        double r;
        if (STEP) {
            const LaneHdr hd = load_lane_hdr(j.hdr, p, (uint32_t)j.n_poses, B >= 64u);
            hr = hd.hr;
            hc = hd.hc;
            if (j.dir_mode) {
                // dedupe pass: "beam" b is the b-th distinct table direction of agent p's scan;
                // raw range only (noise / iTTC / beam expansion happen in k_expand_beams)
                if (b < hd.n_dirs) {
                    int didx = hd.i0 + b;
                    if (didx >= k.theta_dis) didx -= k.theta_dis;
                    const double2 cs = k.cs[didx];
                    j.ranges[ray] = march_from_first<LAYOUT, POW2, IDENT>(k, lut_lds, hd.x, hd.y, cs.x, cs.y, hd.d0, hr, hc, nl);
                }
                continue;
            }
            const double2 cs = k.cs[beam_dir_index(k, hd.start, b)];
            r = march_from_first<LAYOUT, POW2, IDENT>(k, lut_lds, hd.x, hd.y, cs.x, cs.y, hd.d0, hr, hc, nl);
            finish_beam(j, B, p, b, ray, r, hd.row, hd.vel);
            continue;
        } else {
            const double2 cs = k.cs[beam_dir_index(k, j.dir_start[p], b)];
            r = march_ray<LAYOUT, POW2, IDENT>(k, lut_lds, j.pose_x[p], j.pose_y[p], cs.x, cs.y, hr, hc, nl);
            if (j.hit_rc) {
                j.hit_rc[(size_t)ray * 2] = hr;
                j.hit_rc[(size_t)ray * 2 + 1] = hc;
            }
            if (j.lookups) atomicAdd(&j.lookups[p], (unsigned long long)nl);
        }
        j.ranges[ray] = r;
    }
}

__device__ __forceinline__ void finish_beam(const RayJob &j, uint32_t B, uint32_t p, int b, uint32_t ray, double r,
                                            int row, double vel)
{
    if (row >= 0) r += j.noise[(size_t)row * B + b];
    if (vel != 0.0 && !(r > j.ttc_side_max + j.ttc_k * fabs(vel)) &&
        ttc_beam_hit(r, j.side_dist[b], vel, j.beam_cos[b], j.ttc_thresh))
        j.wall_flag[p] = 1;
    j.ranges[ray] = r;
}

// ---- K2b: beam expansion of the dedupe pass ---------------------------------------------------
// More beams than table directions (BASELINE config 5: 4096 beams, theta_dis = 2000 -> 1497
// distinct directions per scan): beams that share a table index from the same origin are the same
// ray.  k_scan_rays (dir_mode) marches each distinct direction once; here every beam picks its
// direction's range, then gets its own noise sample and iTTC test.  Bit-identical to marching
// every beam, ~2.7x fewer table gathers.
__global__ void __launch_bounds__(256) k_expand_beams(RayJob j, ScanConst k)
{
    const uint32_t B = (uint32_t)k.num_beams;
    const uint32_t ray = blockIdx.x * blockDim.x + threadIdx.x;
    if (ray >= j.n_rays) return;
    const uint32_t p = j.div_magic ? (__umulhi(ray, j.div_magic) >> j.div_shift) : ray / B;
    const int b = (int)(ray - p * B);
    const LaneHdr hd = load_lane_hdr(j.hdr, p, (uint32_t)j.n_poses, B >= 64u);
    int s = beam_dir_index(k, hd.start, b) - hd.i0;
    if (s < 0) s += k.theta_dis;
    const double r = j.dir_ranges[(size_t)p * j.dir_stride + s];
    finish_beam(j, B, p, b, ray, r, hd.row, hd.vel);
}

// ---- K3: finalize ---------------------------------------------------------------------------
// One wave per agent.  RaceCar.check_ttc's side effects (:246-252), Simulator's collision OR
// (:588-589), then RaceCar.ray_cast_agents (:206-227): opponents from the :574 snapshot, ego
// pose = live state (heading already zeroed on a wall hit), box = the ego's own params.
__global__ void __launch_bounds__(64, 5) k_finalize(AgentArrays a, int32_t B)
{
    const int i = blockIdx.x, tid = threadIdx.x;
    const int N = a.n_agents_total, A = a.agents_per_env;
    const int wall = a.in_collision[i];
    const double ex = a.state[i], ey = a.state[(size_t)N + i];
    const double eth = wall ? 0.0 : a.state[4 * (size_t)N + i];
    if (tid == 0) {
        if (wall) {
            a.state[3 * (size_t)N + i] = 0.;
            a.state[4 * (size_t)N + i] = 0.;
            a.state[5 * (size_t)N + i] = 0.;
            a.state[6 * (size_t)N + i] = 0.;
            a.collisions[i] = 1.0;
        }
        a.step_count[i] += 1;
    }
    const int me = i % A;
    double *sc = a.scans + (size_t)i * B;
    for (int jj = 0; jj < A; ++jj) {
        if (jj == me) continue;
        const int32_t *win = a.opp_window + ((size_t)i * A + jj) * 4 + (wall ? 2 : 0);
        const int lo = win[0], hi = win[1];
        if (hi < lo) continue;  // nothing of this opponent can be hit
        const double *ov = a.opp_verts + ((size_t)i * A + jj) * 8;
        double v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = ov[c];
        for (int b = lo + tid; b <= hi; b += 64) {
            const double bt = eth + a.scan_angles[b];
            const double v3x = cos(bt + kPi / 2.), v3y = sin(bt + kPi / 2.);
            const double r0 = sc[b];
            const double r = box_range(ex, ey, v3x, v3y, v, r0);
            if (r < r0) sc[b] = r;
        }
        // the next opponent may touch the same beams: make this wave's stores visible to it
        __threadfence_block();
    }
}

// single-agent envs: no opponents, one lane per agent is enough
__global__ void __launch_bounds__(256) k_finalize_solo(AgentArrays a)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = a.n_agents_total;
    if (i >= N) return;
    const int wall = a.in_collision[i];
    if (wall) {
        a.state[3 * (size_t)N + i] = 0.;
        a.state[4 * (size_t)N + i] = 0.;
        a.state[5 * (size_t)N + i] = 0.;
        a.state[6 * (size_t)N + i] = 0.;
    }
    // collision_multiple on a single body returns zeros every step (collision_models.py:196-197);
    // k_collide is not launched for A = 1, so the flag is (re)written here
    a.collisions[i] = wall ? 1.0 : 0.0;
    a.step_count[i] += 1;
}

// unit-path helper: AoS poses [M][3] -> pose_x, pose_y, dir_start
__global__ void k_prepare_poses(ScanConst k, const double *__restrict__ poses, int m, double *__restrict__ px,
                                double *__restrict__ py, double *__restrict__ start)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    px[i] = poses[3 * i];
    py[i] = poses[3 * i + 1];
    start[i] = scan_start_index(k, poses[3 * i + 2]);
}

// ---- reset ------------------------------------------------------------------------------
__global__ void k_reset(AgentArrays a, const double *__restrict__ poses, const uint8_t *__restrict__ env_mask)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = a.n_agents_total;
    if (i >= N) return;
    if (env_mask && !env_mask[i / a.agents_per_env]) return;
#pragma unroll
    for (int c = 0; c < 7; ++c) a.state[(size_t)c * N + i] = 0.;
    a.state[i] = poses[3 * (size_t)i];
    a.state[(size_t)N + i] = poses[3 * (size_t)i + 1];
    a.state[4 * (size_t)N + i] = poses[3 * (size_t)i + 2];
    a.steer_buf[i] = 0.;
    a.steer_buf[(size_t)N + i] = 0.;
    a.buf_cnt[i] = 0;
    a.in_collision[i] = 0;
    a.step_count[i] = 0;
}

// in-place re-seat of finished environments (SURVEY §8d "mask reset"): an env whose ego agent
// has collisions != 0 is reset to its start poses, exactly as k_reset would with that env masked.
__global__ void k_reset_collided(AgentArrays a, const double *__restrict__ start_poses, int ego_idx,
                                 int32_t *__restrict__ n_reset)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = a.n_agents_total, A = a.agents_per_env;
    if (i >= N) return;
    const int env = i / A;
    if (a.collisions[env * A + ego_idx] == 0.0) return;
    // every lane of the env reads the ego flag before any lane of the env can have cleared it:
    // collisions[] is not written here (it keeps the step's value, as Simulator.collisions does)
#pragma unroll
    for (int c = 0; c < 7; ++c) a.state[(size_t)c * N + i] = 0.;
    a.state[i] = start_poses[3 * (size_t)i];
    a.state[(size_t)N + i] = start_poses[3 * (size_t)i + 1];
    a.state[4 * (size_t)N + i] = start_poses[3 * (size_t)i + 2];
    a.steer_buf[i] = 0.;
    a.steer_buf[(size_t)N + i] = 0.;
    a.buf_cnt[i] = 0;
    a.in_collision[i] = 0;
    a.step_count[i] = 0;
    if (n_reset && i - env * A == ego_idx) atomicAdd(n_reset, 1);
}

// ---- episode logic on the device ------------------------------------------------
// F110Env._check_done (f110_env.py:204-246): start/finish-zone toggles, lap counts and times, done
// = ego collided or every agent has 4 toggles — one lane per env, so an RL loop that keeps its
// policy on the GPU never has to read poses back to decide `done`.
struct EpisodeArrays {
    int32_t ego_idx, pad;
    double timestep;
    double *start_poses;   // [N][3]
    double *rot;           // [E][4] start_rot row-major (f110_env.py:331), computed by the host
    double *current_time;  // [E]
    uint8_t *near_start;   // [N]
    double *toggle;        // [N]
    double *lap_count;     // [N]
    double *lap_time;      // [N]
    uint8_t *done;         // [E]
    uint8_t *checkpoint;   // [N] toggle >= 4
};

__global__ void __launch_bounds__(256) k_episode(AgentArrays a, EpisodeArrays ep, int num_envs)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= num_envs) return;
    const int N = a.n_agents_total, A = a.agents_per_env;
    const double ct = ep.current_time[e] + ep.timestep;  // f110_env.py:295
    ep.current_time[e] = ct;
    const double r00 = ep.rot[4 * e], r01 = ep.rot[4 * e + 1], r10 = ep.rot[4 * e + 2], r11 = ep.rot[4 * e + 3];
    const double left_t = 2, right_t = 2;
    bool all4 = true;
    for (int s = 0; s < A; ++s) {
        const int i = e * A + s;
        const double px = a.state[i] - ep.start_poses[3 * (size_t)i];
        const double py = a.state[(size_t)N + i] - ep.start_poses[3 * (size_t)i + 1];
        const double dx = r00 * px + r01 * py;  // np.dot(start_rot, [px; py]) :223
        double ty = r10 * px + r11 * py;
        if (ty > left_t)
            ty -= left_t;
        else if (ty < -right_t)
            ty = -right_t - ty;
        else
            ty = 0;
        const double dist2 = dx * dx + ty * ty;
        const bool closes = dist2 <= 0.1;
        bool near = ep.near_start[i] != 0;
        double tog = ep.toggle[i];
        if (closes && !near) {
            near = true;
            tog += 1;
        } else if (!closes && near) {
            near = false;
            tog += 1;
        }
        ep.near_start[i] = near ? 1 : 0;
        ep.toggle[i] = tog;
        ep.lap_count[i] = floor(tog / 2);  // toggle_list // 2
        if (tog < 4) ep.lap_time[i] = ct;
        ep.checkpoint[i] = tog >= 4 ? 1 : 0;
        all4 = all4 && (tog >= 4);
    }
    ep.done[e] = (a.collisions[e * A + ep.ego_idx] != 0.0 || all4) ? 1 : 0;  // :244
}

// re-seat every env whose done flag is set (F110Env.reset :319-334 without its zero-action step)
__global__ void __launch_bounds__(256) k_episode_reset_done(AgentArrays a, EpisodeArrays ep, int32_t *__restrict__ n_reset)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = a.n_agents_total, A = a.agents_per_env;
    if (i >= N) return;
    const int e = i / A;
    if (!ep.done[e]) return;
#pragma unroll
    for (int c = 0; c < 7; ++c) a.state[(size_t)c * N + i] = 0.;
    a.state[i] = ep.start_poses[3 * (size_t)i];
    a.state[(size_t)N + i] = ep.start_poses[3 * (size_t)i + 1];
    a.state[4 * (size_t)N + i] = ep.start_poses[3 * (size_t)i + 2];
    a.steer_buf[i] = 0.;
    a.steer_buf[(size_t)N + i] = 0.;
    a.buf_cnt[i] = 0;
    a.in_collision[i] = 0;
    a.step_count[i] = 0;
    ep.near_start[i] = 1;
    ep.toggle[i] = 0.;
    if (i - e * A == 0) {
        ep.current_time[e] = 0.;
        if (n_reset) atomicAdd(n_reset, 1);
    }
}

// done[] is read by every lane of the env above and cleared here, in a separate launch
__global__ void k_episode_clear_done(EpisodeArrays ep, int num_envs)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < num_envs) ep.done[e] = 0;
}

__global__ void k_episode_reset(AgentArrays a, EpisodeArrays ep, const double *__restrict__ poses,
                                const double *__restrict__ rot, const uint8_t *__restrict__ env_mask)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = a.n_agents_total, A = a.agents_per_env;
    if (i >= N) return;
    const int e = i / A;
    if (env_mask && !env_mask[e]) return;
    ep.start_poses[3 * (size_t)i] = poses[3 * (size_t)i];
    ep.start_poses[3 * (size_t)i + 1] = poses[3 * (size_t)i + 1];
    ep.start_poses[3 * (size_t)i + 2] = poses[3 * (size_t)i + 2];
    ep.near_start[i] = 1;
    ep.toggle[i] = 0.;
    ep.checkpoint[i] = 0;
    if (i - e * A == 0) {
        ep.current_time[e] = 0.;
        ep.done[e] = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) ep.rot[4 * e + c] = rot[4 * e + c];
    }
}

// ---- unit kernels (one per reference function; parity tests) ------------------------------
__global__ void k_dir_index_unit(ScanConst k, const double *__restrict__ thetas, int m, int32_t *__restrict__ idx)
{
    const int p = blockIdx.x;
    const double start = scan_start_index(k, thetas[p]);
    for (int b = threadIdx.x; b < k.num_beams; b += blockDim.x) idx[(size_t)p * k.num_beams + b] = beam_dir_index(k, start, b);
}

__global__ void k_dynamics_unit(const double *__restrict__ x, const double *__restrict__ u, const double *__restrict__ params,
                                int m, double *__restrict__ f_st, double *__restrict__ f_ks)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const VehicleParams vp = load_params(params);
    double xs[7], f[7];
#pragma unroll
    for (int c = 0; c < 7; ++c) xs[c] = x[7 * (size_t)i + c];
    rhs_single_track(xs, u[2 * i], u[2 * i + 1], vp, f);
#pragma unroll
    for (int c = 0; c < 7; ++c) f_st[7 * (size_t)i + c] = f[c];
    rhs_kinematic(xs, u[2 * i], u[2 * i + 1], vp, f);
#pragma unroll
    for (int c = 0; c < 5; ++c) f_ks[5 * (size_t)i + c] = f[c];
}

__global__ void k_pid_unit(const double *__restrict__ in, const double *__restrict__ params, int m, double *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const VehicleParams vp = load_params(params);
    double accl, sv;
    speed_steer_controller(in[4 * i], in[4 * i + 1], in[4 * i + 2], in[4 * i + 3], vp, accl, sv);
    out[2 * i] = accl;
    out[2 * i + 1] = sv;
}

__global__ void k_update_pose_unit(const double *__restrict__ s0, const double *__restrict__ buf0, const int32_t *__restrict__ cnt0,
                                   const double *__restrict__ act, const double *__restrict__ params, double dt, int integ,
                                   double lidar_dist, int m, double *__restrict__ s1, double *__restrict__ buf1,
                                   int32_t *__restrict__ cnt1, double *__restrict__ spose)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const VehicleParams vp = load_params(params);
    double st[7], sp[3];
#pragma unroll
    for (int c = 0; c < 7; ++c) st[c] = s0[7 * (size_t)i + c];
    double b0 = buf0[2 * i], b1 = buf0[2 * i + 1];
    int cnt = cnt0[i];
    advance_vehicle(st, b0, b1, cnt, act[2 * i], act[2 * i + 1], vp, dt, integ, lidar_dist, sp);
#pragma unroll
    for (int c = 0; c < 7; ++c) s1[7 * (size_t)i + c] = st[c];
    buf1[2 * i] = b0;
    buf1[2 * i + 1] = b1;
    cnt1[i] = cnt;
    spose[3 * i] = sp[0];
    spose[3 * i + 1] = sp[1];
    spose[3 * i + 2] = sp[2];
}

__global__ void k_vertices_unit(const double *__restrict__ poses, double length, double width, int m, double *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    double v[8];
    box_vertices(poses[3 * i], poses[3 * i + 1], poses[3 * i + 2], length, width, v);
#pragma unroll
    for (int c = 0; c < 8; ++c) out[8 * (size_t)i + c] = v[c];
}

__global__ void k_gjk_unit(const double *__restrict__ va, const double *__restrict__ vb, int m, int32_t *__restrict__ flags)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    double a[8], b[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        a[c] = va[8 * (size_t)i + c];
        b[c] = vb[8 * (size_t)i + c];
    }
    flags[i] = gjk_overlap(a, b) ? 1 : 0;
}

// collision_multiple :184-212 — one lane per body, same last-writer rule as k_collide
__global__ void k_collision_multiple_unit(const double *__restrict__ verts, int groups, int n, double *__restrict__ col,
                                          double *__restrict__ idx)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= groups * n) return;
    const int g = t / n, me = t - g * n;
    double mine[8], other[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) mine[c] = verts[8 * (size_t)t + c];
    bool hit = false;
    int partner = -1;
    for (int j = 0; j < n; ++j) {
        if (j == me) continue;
#pragma unroll
        for (int c = 0; c < 8; ++c) other[c] = verts[8 * ((size_t)g * n + j) + c];
        const bool cc = (me < j) ? gjk_overlap(mine, other) : gjk_overlap(other, mine);
        if (cc) {
            hit = true;
            partner = j;
        }
    }
    col[t] = hit ? 1.0 : 0.0;
    idx[t] = (double)partner;
}

__global__ void k_ttc_unit(const double *__restrict__ scans, const double *__restrict__ vels, int m, int B,
                           const double *__restrict__ beam_cos, const double *__restrict__ side, double thresh,
                           int32_t *__restrict__ flags)
{
    const int p = blockIdx.x;
    const double vel = vels[p];
    int hit = 0;
    if (vel != 0.0)
        for (int b = threadIdx.x; b < B; b += blockDim.x)
            if (ttc_beam_hit(scans[(size_t)p * B + b], side[b], vel, beam_cos[b], thresh)) hit = 1;
    const int any = __syncthreads_or(hit);
    if (threadIdx.x == 0) flags[p] = any;
}

__global__ void k_raycast_unit(const double *__restrict__ ego, const double *__restrict__ verts, int m, int B,
                               const double *__restrict__ scan_angles, double angle_inc, double *__restrict__ scans,
                               int32_t *__restrict__ minmax)
{
    const int p = blockIdx.x, tid = threadIdx.x;
    const double ex = ego[3 * p], ey = ego[3 * p + 1], eth = ego[3 * p + 2];
    double v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = verts[8 * (size_t)p + c];
    // circumscribed disc of the quadrilateral: centroid + largest vertex distance
    const double cx = (((v[0] + v[2]) + v[4]) + v[6]) / 4, cy = (((v[1] + v[3]) + v[5]) + v[7]) / 4;
    double r2 = 0.0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const double dx = v[2 * c] - cx, dy = v[2 * c + 1] - cy;
        r2 = fmax(r2, dx * dx + dy * dy);
    }
    int ref_lo, ref_hi, lo, hi;
    opponent_beam_window(ex, ey, eth, v, cx, cy, sqrt(r2) * 1.000001, scan_angles, B, angle_inc, ref_lo, ref_hi, lo, hi);
    if (minmax && tid == 0) {
        minmax[2 * p] = ref_lo;
        minmax[2 * p + 1] = ref_hi;
    }
    double *sc = scans + (size_t)p * B;
    for (int b = lo + tid; b <= hi; b += blockDim.x) {
        const double bt = eth + scan_angles[b];
        const double r0 = sc[b];
        const double r = box_range(ex, ey, cos(bt + kPi / 2.), sin(bt + kPi / 2.), v, r0);
        if (r < r0) sc[b] = r;
    }
}

__global__ void k_get_range_unit(const double *__restrict__ in, int m, double *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const double *r = in + 8 * (size_t)i;
    const double bt = r[3];
    out[i] = edge_range(r[0], r[1], cos(bt + kPi / 2.), sin(bt + kPi / 2.), r[4], r[5], r[6], r[7]);
}

// ---- map pipeline: flip + threshold + exact EDT + dt = res*sqrt(d2) ------------------------
// laser_models.py:398-404
__global__ void k_flip_threshold(const uint8_t *__restrict__ img_top_first, int H, int W, uint8_t *__restrict__ bin)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)H * W) return;
    const int r = (int)(t / W), c = (int)(t - (size_t)r * W);
    bin[t] = img_top_first[(size_t)(H - 1 - r) * W + c] > 128 ? 1 : 0;
}

constexpr uint32_t kEdtInf = 0x00007FFFu;  // "no obstacle in this column": larger than any map side

// phase 1: per column, distance to the nearest obstacle cell in that column (lane = column)
__global__ void k_edt_columns(const uint8_t *__restrict__ bin, int H, int W, uint32_t *__restrict__ g)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= W) return;
    uint32_t run = kEdtInf;
    for (int y = 0; y < H; ++y) {
        run = bin[(size_t)y * W + x] ? (run >= kEdtInf ? kEdtInf : run + 1) : 0;
        g[(size_t)y * W + x] = run;
    }
    run = kEdtInf;
    for (int y = H - 1; y >= 0; --y) {
        const uint32_t cur = g[(size_t)y * W + x];
        run = (cur == 0) ? 0 : (run >= kEdtInf ? kEdtInf : run + 1);
        if (run < cur) g[(size_t)y * W + x] = run;
    }
}

// phase 2: per row, d2[u] = min_i (u-i)^2 + g[i]^2 — exact integer lower envelope by brute
// force; the row of g^2 is staged in LDS and every lane walks it (LDS broadcast reads).
__global__ void __launch_bounds__(256) k_edt_rows(const uint32_t *__restrict__ g, int H, int W, uint32_t *__restrict__ d2)
{
    extern __shared__ uint32_t g2[];
    const int y = blockIdx.x;
    for (int i = threadIdx.x; i < W; i += blockDim.x) {
        const uint32_t v = g[(size_t)y * W + i];
        g2[i] = v * v;  // <= 0x7FFF^2 < 2^30
    }
    __syncthreads();
    for (int u = threadIdx.x; u < W; u += blockDim.x) {
        uint32_t best = 0xFFFFFFFFu;
        for (int i = 0; i < W; ++i) {
            const int d = u - i;
            const uint32_t cand = (uint32_t)(d * d) + g2[i];
            best = cand < best ? cand : best;
        }
        d2[(size_t)y * W + u] = best;
    }
}

__global__ void k_dt_from_d2(const uint32_t *__restrict__ d2, size_t n, double res, double *__restrict__ dt)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) dt[t] = res * sqrt((double)d2[t]);  // laser_models.py:52
}

__global__ void k_retile(const double *__restrict__ rowmajor, int H, int W, int tiles_w, int tiles_h, double *__restrict__ tiled)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)tiles_w * tiles_h * 16;
    if (t >= total) return;
    const size_t tile = t >> 4;
    const int within = (int)(t & 15);
    const int r = (int)(tile / tiles_w) * 4 + (within >> 2);
    const int c = (int)(tile % tiles_w) * 4 + (within & 3);
    tiled[t] = (r < H && c < W) ? rowmajor[(size_t)r * W + c] : 0.0;
}

// CODE8 layout: code = rank of the cell's value among the 255 smallest distinct table values
// (binary search in the ascending LUT), 255 when it is not one of them; 16x8-cell tiles.
__global__ void k_build_codes(const double *__restrict__ rowmajor, int H, int W, int ctiles_w, int ctiles_h,
                              const double *__restrict__ lut, int n_lut, uint8_t *__restrict__ codes)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)ctiles_w * ctiles_h * 128;
    if (t >= total) return;
    const size_t tile = t >> 7;
    const int within = (int)(t & 127);
    const int r = (int)(tile / ctiles_w) * 8 + (within >> 4);
    const int c = (int)(tile % ctiles_w) * 16 + (within & 15);
    uint8_t code = 255;
    if (r < H && c < W) {
        const double v = rowmajor[(size_t)r * W + c];
        int lo = 0, hi = n_lut - 1;
        while (lo <= hi) {
            const int mid = (lo + hi) >> 1;
            const double m = lut[mid];
            if (m == v) {
                code = (uint8_t)mid;
                break;
            }
            if (m < v) lo = mid + 1; else hi = mid - 1;
        }
    }
    codes[t] = code;
}

__global__ void k_interleave_cs(const double *__restrict__ sines, const double *__restrict__ cosines, int n, double2 *__restrict__ cs)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) cs[i] = make_double2(cosines[i], sines[i]);
}

// ============================================================================ host side

struct f110_sim {
    f110_config cfg{};
    int N = 0;
    hipStream_t stream = nullptr;
    hipStream_t side_stream = nullptr;          // k_collide runs here, concurrently with k_scan_rays
    hipEvent_t ev_integrated = nullptr, ev_collided = nullptr;
    AgentArrays dev{};
    ScanConst k{};
    bool has_map = false;
    uint32_t step_magic = 0, step_shift = 0;  // ray -> agent division constants of the step launch
    int scan_tasks_per_wave = 1, num_cus = 256;  // consecutive 64-ray tasks per wave
    double ttc_side_max = INFINITY, ttc_cos_max = INFINITY;  // see f110_set_beam_tables
    uint8_t *d_codes = nullptr;
    double *d_dir_ranges = nullptr;  // dedupe pass output [N][dir_stride]
    int dir_stride = 0;              // > 0: dedupe enabled
    uint32_t dir_magic = 0, dir_shift = 0;
    double *d_lut = nullptr;
    int scan_block = 64;
    double *d_params = nullptr, *d_noise = nullptr, *d_scan_angles = nullptr, *d_beam_cos = nullptr, *d_side = nullptr;
    double *d_dt_row = nullptr, *d_dt_tiled = nullptr, *d_actions = nullptr, *d_poses = nullptr;
    double2 *d_cs = nullptr;
    uint8_t *d_mask = nullptr;
    ncclComm_t comm = nullptr;   // optional RCCL communicator for the observation gather
    int comm_ranks = 0;
    EpisodeArrays ep{};
    bool has_episode = false;
    double *d_rot_stage = nullptr;
    // timing
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    bool profiling = false;
    std::vector<hipEvent_t> prof_events;  // per step: before integrate, before scan, after scan, after finalize
    size_t prof_used = 0;
    char err[512] = {0};
};

static thread_local char g_err[512] = {0};

static int fail(f110_sim *h, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    snprintf(g_err, sizeof g_err, "%s", buf);
    if (h) snprintf(h->err, sizeof h->err, "%s", buf);
    return code;
}

#define HIPCHK(h, call)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess)                                                                        \
            return fail(h, F110_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

template <typename T>
static int dmalloc(f110_sim *h, T **p, size_t count)
{
    HIPCHK(h, hipMalloc(reinterpret_cast<void **>(p), count * sizeof(T) > 0 ? count * sizeof(T) : 8));
    return F110_OK;
}

#define TRY(expr)              \
    do {                       \
        int rc_ = (expr);      \
        if (rc_ != F110_OK) return rc_; \
    } while (0)

// RAII scratch for the unit entry points
struct Scratch {
    f110_sim *h;
    std::vector<void *> ptrs;
    explicit Scratch(f110_sim *hh) : h(hh) {}
    ~Scratch()
    {
        for (void *p : ptrs) (void)hipFree(p);
    }
    template <typename T>
    int up(const T *host, size_t count, T **dev)
    {
        TRY(dmalloc(h, dev, count));
        ptrs.push_back(*dev);
        if (host) HIPCHK(h, hipMemcpyAsync(*dev, host, count * sizeof(T), hipMemcpyHostToDevice, h->stream));
        return F110_OK;
    }
    template <typename T>
    int down(T *host, const T *dev, size_t count)
    {
        HIPCHK(h, hipMemcpyAsync(host, dev, count * sizeof(T), hipMemcpyDeviceToHost, h->stream));
        return F110_OK;
    }
};

static inline dim3 grid1d(size_t n, int block) { return dim3((unsigned)((n + block - 1) / block)); }

typedef void (*scan_rays_fn)(RayJob, ScanConst);
static int copy_col(f110_sim *h, double *dst, const double *src, size_t n);

// ray / B by multiply-high: find (magic, shift) with umulhi(x, magic) >> shift == x / B for every
// x < n (verified at every multiple of B and its predecessor, which is sufficient because both
// sides are monotone step functions of x).  Returns false when no 32-bit magic works.
static bool find_div_magic(uint32_t B, uint32_t n, uint32_t *magic, uint32_t *shift)
{
    if (B < 2) return false;
    for (uint32_t s = 0; s < 32; ++s) {
        const unsigned long long m = ((1ull << (32 + s)) + B - 1) / B;  // ceil(2^(32+s) / B)
        if (m >> 32) break;
        bool ok = true;
        for (unsigned long long x = B; ok && x <= (unsigned long long)n + B; x += B) {
            const unsigned long long xs[2] = {x - 1, x};
            for (unsigned long long v : xs) {
                if (v >= n) continue;
                if ((((v * m) >> 32) >> s) != v / B) ok = false;
            }
        }
        if (ok) {
            *magic = (uint32_t)m;
            *shift = s;
            return true;
        }
    }
    return false;
}

static void set_div_magic(RayJob &j, uint32_t B)
{
    uint32_t m = 0, s = 0;
    if (find_div_magic(B, j.n_rays, &m, &s)) {
        j.div_magic = m;
        j.div_shift = s;
    } else {
        j.div_magic = 0;
        j.div_shift = 0;
    }
}


template <bool STEP>
static scan_rays_fn pick_rays(const ScanConst &k, int layout)
{
#define SEL(L) (k.res_pow2 ? (k.ident_rot ? k_scan_rays<L, true, true, STEP> : k_scan_rays<L, true, false, STEP>) \
                           : (k.ident_rot ? k_scan_rays<L, false, true, STEP> : k_scan_rays<L, false, false, STEP>))
    if (layout == F110_MAP_CODE8) return SEL(LAYOUT_CODE8);
    return layout == F110_MAP_TILED_F64 ? SEL(LAYOUT_TILED) : SEL(LAYOUT_ROWMAJOR);
#undef SEL
}

static dim3 rays_grid(RayJob &j, int block, int tasks_per_wave)
{
    j.n_tasks = (j.n_rays + 63u) / 64u;
    j.tasks_per_wave = tasks_per_wave > 0 ? (uint32_t)tasks_per_wave : 1u;
    j.xcd_remap = 1;  // measured neutral on MI355X (the table's hot set is L2-resident either way)
    const uint32_t waves = (j.n_tasks + j.tasks_per_wave - 1) / j.tasks_per_wave;
    const uint32_t wpb = (uint32_t)block / 64u;
    return dim3((waves + wpb - 1) / wpb);
}

extern "C" {

const char *f110_last_error(const f110_sim *h) { return (h && h->err[0]) ? h->err : g_err; }
int f110_abi_version(void) { return F110_ABI_VERSION; }

int f110_device_count(int *count)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        if (count) *count = 0;
        return fail(nullptr, F110_ERR_HIP, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
    }
    if (count) *count = n;
    return F110_OK;
}

static void default_beam_tables(const f110_config &c, std::vector<double> &sa, std::vector<double> &co, std::vector<double> &sd)
{
    // base_classes.py:125-158 (the Python host normally overrides these with NumPy's values)
    const int B = c.num_beams;
    const double incr = c.fov / (B - 1);
    const double dist_sides = c.params[F110_P_WIDTH] / 2., dist_fr = (c.params[F110_P_LF] + c.params[F110_P_LR]) / 2.;
    sa.resize(B); co.resize(B); sd.resize(B);
    for (int i = 0; i < B; ++i) {
        const double angle = -c.fov / 2. + i * incr;
        double to_side, to_fr;
        sa[i] = angle;
        co[i] = std::cos(angle);
        if (angle > 0) {
            if (angle < kPi / 2) { to_side = dist_sides / std::sin(angle); to_fr = dist_fr / std::cos(angle); }
            else { to_side = dist_sides / std::cos(angle - kPi / 2.); to_fr = dist_fr / std::sin(angle - kPi / 2.); }
        } else {
            if (angle > -kPi / 2) { to_side = dist_sides / std::sin(-angle); to_fr = dist_fr / std::cos(-angle); }
            else { to_side = dist_sides / std::cos(-angle - kPi / 2); to_fr = dist_fr / std::sin(-angle - kPi / 2); }
        }
        sd[i] = to_side < to_fr ? to_side : to_fr;
    }
}

int f110_create(const f110_config *cfg, f110_sim **out)
{
    if (!cfg || !out) return fail(nullptr, F110_ERR_INVALID, "f110_create: null argument");
    *out = nullptr;
    if (cfg->abi_version != F110_ABI_VERSION) return fail(nullptr, F110_ERR_INVALID, "ABI version mismatch (%d vs %d)", cfg->abi_version, F110_ABI_VERSION);
    if (cfg->num_envs < 1 || cfg->num_agents < 1 || cfg->num_beams < 2 || cfg->theta_dis < 2)
        return fail(nullptr, F110_ERR_INVALID, "f110_create: num_envs/num_agents >= 1, num_beams/theta_dis >= 2 required");
    if (cfg->integrator != F110_INTEGRATOR_RK4 && cfg->integrator != F110_INTEGRATOR_EULER)
        return fail(nullptr, F110_ERR_INVALID, "Invalid Integrator Specified. Please choose RK4 or Euler");
    if (cfg->map_layout != F110_MAP_ROWMAJOR_F64 && cfg->map_layout != F110_MAP_TILED_F64 && cfg->map_layout != F110_MAP_CODE8)
        return fail(nullptr, F110_ERR_INVALID, "unknown map_layout %d", cfg->map_layout);
    if ((long long)cfg->num_envs * cfg->num_agents * (long long)cfg->num_beams > 0xFFFFFF00LL) return fail(nullptr, F110_ERR_INVALID, "num_envs*num_agents*num_beams must stay below 2^32");
    int ndev = 0;
    {
        hipError_t e = hipGetDeviceCount(&ndev);
        if (e != hipSuccess || ndev < 1)
            return fail(nullptr, F110_ERR_HIP, "no usable HIP device (%s); libf110_hip has no CPU fallback",
                        e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    }
    if (cfg->device_id < 0 || cfg->device_id >= ndev) return fail(nullptr, F110_ERR_INVALID, "device_id %d out of range (%d devices)", cfg->device_id, ndev);
    f110_sim *h = new (std::nothrow) f110_sim();
    if (!h) return fail(nullptr, F110_ERR_NOMEM, "out of host memory");
    h->cfg = *cfg;
    h->N = cfg->num_envs * cfg->num_agents;
    const int N = h->N, B = cfg->num_beams;
    h->scan_block = cfg->scan_block > 0 ? cfg->scan_block : 64;
    h->scan_tasks_per_wave = cfg->scan_tasks_per_wave > 0 ? cfg->scan_tasks_per_wave : 4;
    if (cfg->scan_block <= 0 && cfg->map_layout == F110_MAP_CODE8) h->scan_block = 256;
    if (h->scan_block % 64 != 0 || h->scan_block > 256) { delete h; return fail(nullptr, F110_ERR_INVALID, "scan_block must be 64, 128, 192 or 256"); }
#define CK(expr) do { int rc_ = (expr); if (rc_ != F110_OK) { snprintf(g_err, sizeof g_err, "%s", h->err); f110_destroy(h); return rc_; } } while (0)
#define CKH(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fail(nullptr, F110_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); f110_destroy(h); return F110_ERR_HIP; } } while (0)
    CKH(hipSetDevice(cfg->device_id));
    {
        hipDeviceProp_t prop;
        CKH(hipGetDeviceProperties(&prop, cfg->device_id));
        h->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    CKH(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    CKH(hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking));
    CKH(hipEventCreateWithFlags(&h->ev_integrated, hipEventDisableTiming));
    CKH(hipEventCreateWithFlags(&h->ev_collided, hipEventDisableTiming));
    CKH(hipEventCreate(&h->ev_begin));
    CKH(hipEventCreate(&h->ev_end));
    AgentArrays &d = h->dev;
    d.n_agents_total = N;
    d.agents_per_env = cfg->num_agents;
    CK(dmalloc(h, &d.state, (size_t)7 * N));
    CK(dmalloc(h, &d.steer_buf, (size_t)2 * N));
    CK(dmalloc(h, &d.buf_cnt, (size_t)N));
    CK(dmalloc(h, &d.scan_pose, (size_t)3 * N));
    CK(dmalloc(h, &d.snap_pose, (size_t)3 * N));
    CK(dmalloc(h, &d.dir_start, (size_t)N));
    CK(dmalloc(h, &d.ray_hdr, (size_t)N));
    CK(dmalloc(h, &d.scans, (size_t)N * B));
    CK(dmalloc(h, &d.collisions, (size_t)N));
    CK(dmalloc(h, &d.collision_idx, (size_t)N));
    CK(dmalloc(h, &d.in_collision, (size_t)N));
    CK(dmalloc(h, &d.step_count, (size_t)N));
    CK(dmalloc(h, &d.opp_window, (size_t)N * cfg->num_agents * 4));
    CK(dmalloc(h, &d.opp_verts, (size_t)N * cfg->num_agents * 8));
    CK(dmalloc(h, &h->d_params, (size_t)cfg->num_agents * NPARAMS));
    CK(dmalloc(h, &h->d_scan_angles, (size_t)B));
    CK(dmalloc(h, &h->d_beam_cos, (size_t)B));
    CK(dmalloc(h, &h->d_side, (size_t)B));
    CK(dmalloc(h, &h->d_cs, (size_t)cfg->theta_dis));
    CK(dmalloc(h, &h->d_actions, (size_t)2 * N));
    CK(dmalloc(h, &h->d_poses, (size_t)3 * N));
    CK(dmalloc(h, &h->d_mask, (size_t)cfg->num_envs));
    CKH(hipMemsetAsync(d.state, 0, sizeof(double) * 7 * N, h->stream));
    CKH(hipMemsetAsync(d.steer_buf, 0, sizeof(double) * 2 * N, h->stream));
    CKH(hipMemsetAsync(d.buf_cnt, 0, sizeof(int32_t) * N, h->stream));
    CKH(hipMemsetAsync(d.scan_pose, 0, sizeof(double) * 3 * N, h->stream));
    CKH(hipMemsetAsync(d.snap_pose, 0, sizeof(double) * 3 * N, h->stream));
    CKH(hipMemsetAsync(d.dir_start, 0, sizeof(double) * N, h->stream));
    CKH(hipMemsetAsync(d.scans, 0, sizeof(double) * (size_t)N * B, h->stream));
    CKH(hipMemsetAsync(d.collisions, 0, sizeof(double) * N, h->stream));
    CKH(hipMemsetAsync(d.in_collision, 0, sizeof(int32_t) * N, h->stream));
    CKH(hipMemsetAsync(d.step_count, 0, sizeof(int32_t) * N, h->stream));
    {
        std::vector<double> minus1(N, -1.0);  // collision_idx = -1 (base_classes.py:488)
        CKH(hipMemcpy(d.collision_idx, minus1.data(), sizeof(double) * N, hipMemcpyHostToDevice));
    }
    d.params = h->d_params;
    d.noise = nullptr;
    d.noise_rows = 0;
    d.scan_angles = h->d_scan_angles;
    d.beam_cos = h->d_beam_cos;
    d.side_dist = h->d_side;
    d.integrator = cfg->integrator;
    d.time_step = cfg->time_step;
    d.lidar_dist = cfg->lidar_dist;
    d.ttc_thresh = cfg->ttc_thresh;
    d.angle_inc = cfg->fov / (B - 1);  // laser_models.py:367
    d.box_length = cfg->params[F110_P_LENGTH];
    d.box_width = cfg->params[F110_P_WIDTH];
    // scan constants that do not depend on the map
    ScanConst &k = h->k;
    k.cs = h->d_cs;
    k.theta_dis = cfg->theta_dis;
    k.num_beams = B;
    k.eps = cfg->eps;
    k.max_range = cfg->max_range;
    k.fov = cfg->fov;
    k.theta_inc = cfg->theta_dis * d.angle_inc / (2. * kPi);  // :368
    k.inv_theta_dis = 1.0 / (double)cfg->theta_dis;
    {
        RayJob tmp{};
        tmp.n_rays = (uint32_t)N * (uint32_t)B;
        set_div_magic(tmp, (uint32_t)B);
        h->step_magic = tmp.div_magic;
        h->step_shift = tmp.div_shift;
    }
    {
        const double g = 64.0 * (double)B * 2.2737367544323206e-13;  // 64 * B * 2^-42
        k.dir_guard = g > 1e-8 ? g : 1e-8;
    }
    if (k.theta_inc < 1.0 && B >= 64) {
        // more beams than table directions: march each distinct direction once (k_expand_beams)
        int stride = (int)std::ceil((B - 1) * k.theta_inc) + 2;
        stride = std::min(stride, cfg->theta_dis);
        stride = (stride + 63) / 64 * 64;
        if (stride < B && (long long)N * stride < 0xFFFFFF00LL) {
            h->dir_stride = stride;
            CK(dmalloc(h, &h->d_dir_ranges, (size_t)N * stride));
            RayJob tmp{};
            tmp.n_rays = (uint32_t)N * (uint32_t)stride;
            set_div_magic(tmp, (uint32_t)stride);
            h->dir_magic = tmp.div_magic;
            h->dir_shift = tmp.div_shift;
        }
    }
    CK(f110_set_params(h, -1, cfg->params));
    {
        std::vector<double> s(cfg->theta_dis), c(cfg->theta_dis);
        for (int i = 0; i < cfg->theta_dis; ++i) {  // np.linspace(0, 2pi, theta_dis), laser_models.py:379
            const double th = i * (kTwoPi / (cfg->theta_dis - 1));
            s[i] = std::sin(th);
            c[i] = std::cos(th);
        }
        CK(f110_set_trig_tables(h, s.data(), c.data(), cfg->theta_dis));
        std::vector<double> sa, co, sd;
        default_beam_tables(*cfg, sa, co, sd);
        CK(f110_set_beam_tables(h, sa.data(), co.data(), sd.data(), B));
    }
    CKH(hipStreamSynchronize(h->stream));
#undef CK
#undef CKH
    *out = h;
    return F110_OK;
}

void f110_destroy(f110_sim *h)
{
    if (!h) return;
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->side_stream) (void)hipStreamSynchronize(h->side_stream);
    (void)f110_comm_destroy(h);
    AgentArrays &d = h->dev;
    void *ptrs[] = {d.opp_verts, d.ray_hdr, d.opp_window, d.state, d.steer_buf, d.buf_cnt, d.scan_pose, d.snap_pose, d.dir_start, d.scans, d.collisions,
                    d.collision_idx, d.in_collision, d.step_count, h->d_params, h->d_noise, h->d_scan_angles,
                    h->d_beam_cos, h->d_side, h->d_dt_row, h->d_dt_tiled, h->d_codes, h->d_dir_ranges, h->d_lut, h->d_actions, h->d_poses, h->d_cs, h->d_mask};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    {
        void *eptrs[] = {h->ep.start_poses, h->ep.rot, h->ep.current_time, h->ep.near_start, h->ep.toggle,
                         h->ep.lap_count, h->ep.lap_time, h->ep.done, h->ep.checkpoint, h->d_rot_stage};
        for (void *p : eptrs)
            if (p) (void)hipFree(p);
    }
    for (hipEvent_t e : h->prof_events) (void)hipEventDestroy(e);
    if (h->ev_integrated) (void)hipEventDestroy(h->ev_integrated);
    if (h->ev_collided) (void)hipEventDestroy(h->ev_collided);
    if (h->side_stream) (void)hipStreamDestroy(h->side_stream);
    if (h->ev_begin) (void)hipEventDestroy(h->ev_begin);
    if (h->ev_end) (void)hipEventDestroy(h->ev_end);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int f110_sync(f110_sim *h)
{
    if (!h) return fail(nullptr, F110_ERR_INVALID, "null handle");
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return F110_OK;
}

// ---- map ---------------------------------------------------------------------------------
static int finish_map(f110_sim *h, int H, int W, double res, double ox, double oy, double oc, double os)
{
    ScanConst &k = h->k;
    k.height = H;
    k.width = W;
    k.tiles_w = (W + 3) / 4;
    k.row_bytes = W * 8;
    k.res = res;
    k.inv_res = 1.0 / res;
    {
        int e = 0;
        const double mant = std::frexp(res, &e);
        k.res_pow2 = (mant == 0.5) ? 1 : 0;
    }
    k.orig_x = ox;
    k.orig_y = oy;
    k.orig_c = oc;
    k.orig_s = os;
    k.ident_rot = (oc == 1.0 && os == 0.0) ? 1 : 0;
    k.w_res = W * res;  // width * resolution, laser_models.py:79
    k.h_res = H * res;
    HIPCHK(h, hipMemcpyAsync(&k.oob_value, h->d_dt_row + ((size_t)H * W - 1), sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (h->cfg.map_layout == F110_MAP_TILED_F64) {
        const int tiles_h = (H + 3) / 4;
        if (h->d_dt_tiled) { (void)hipFree(h->d_dt_tiled); h->d_dt_tiled = nullptr; }
        const size_t total = (size_t)k.tiles_w * tiles_h * 16;
        TRY(dmalloc(h, &h->d_dt_tiled, total));
        hipLaunchKernelGGL(k_retile, grid1d(total, 256), dim3(256), 0, h->stream, h->d_dt_row, H, W, k.tiles_w, tiles_h, h->d_dt_tiled);
        HIPCHK(h, hipGetLastError());
        k.table = h->d_dt_tiled;
        k.table_rm = h->d_dt_row;
    } else {
        k.table = h->d_dt_row;
        k.table_rm = h->d_dt_row;
    }
    if (h->cfg.map_layout == F110_MAP_CODE8) {
        // the 255 smallest distinct table values (one-time host sort of the downloaded table)
        std::vector<double> vals((size_t)H * W);
        HIPCHK(h, hipMemcpyAsync(vals.data(), h->d_dt_row, vals.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        vals.erase(std::remove_if(vals.begin(), vals.end(), [](double v) { return v != v; }), vals.end());
        std::sort(vals.begin(), vals.end());
        vals.erase(std::unique(vals.begin(), vals.end()), vals.end());
        std::vector<double> lut(256, INFINITY);
        const int n_lut = (int)std::min<size_t>(vals.size(), (size_t)kLutEntries);
        for (int i = 0; i < n_lut; ++i) lut[i] = vals[i];
        const int ctw = (W + 15) / 16, cth = (H + 7) / 8;
        if (h->d_codes) { (void)hipFree(h->d_codes); h->d_codes = nullptr; }
        if (!h->d_lut) TRY(dmalloc(h, &h->d_lut, (size_t)256));
        const size_t total = (size_t)ctw * cth * 128;
        TRY(dmalloc(h, &h->d_codes, total));
        HIPCHK(h, hipMemcpyAsync(h->d_lut, lut.data(), 256 * sizeof(double), hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(k_build_codes, grid1d(total, 256), dim3(256), 0, h->stream, h->d_dt_row, H, W, ctw, cth, h->d_lut, n_lut, h->d_codes);
        HIPCHK(h, hipGetLastError());
        k.codes = h->d_codes;
        k.lut = h->d_lut;
        k.code_tile_row_bytes = ctw * 128;
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->has_map = true;
    return F110_OK;
}

int f110_set_map_image(f110_sim *h, const uint8_t *h_img, int32_t H, int32_t W, double res, double ox, double oy, double oyaw)
{
    if (!h || !h_img) return fail(h, F110_ERR_INVALID, "f110_set_map_image: null argument");
    if (H < 1 || W < 1 || H > 16384 || W > 16384 || !(res > 0)) return fail(h, F110_ERR_INVALID, "f110_set_map_image: bad shape %dx%d or resolution", H, W);
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    const size_t n = (size_t)H * W;
    uint8_t *d_img = nullptr, *d_bin = nullptr;
    uint32_t *d_g = nullptr, *d_d2 = nullptr;
    Scratch s(h);
    TRY(s.up(h_img, n, &d_img));
    TRY(s.up<uint8_t>(nullptr, n, &d_bin));
    TRY(s.up<uint32_t>(nullptr, n, &d_g));
    TRY(s.up<uint32_t>(nullptr, n, &d_d2));
    if (h->d_dt_row) { (void)hipFree(h->d_dt_row); h->d_dt_row = nullptr; }
    TRY(dmalloc(h, &h->d_dt_row, n));
    hipLaunchKernelGGL(k_flip_threshold, grid1d(n, 256), dim3(256), 0, h->stream, d_img, H, W, d_bin);
    hipLaunchKernelGGL(k_edt_columns, grid1d(W, 64), dim3(64), 0, h->stream, d_bin, H, W, d_g);
    hipLaunchKernelGGL(k_edt_rows, dim3(H), dim3(256), (size_t)W * sizeof(uint32_t), h->stream, d_g, H, W, d_d2);
    hipLaunchKernelGGL(k_dt_from_d2, grid1d(n, 256), dim3(256), 0, h->stream, d_d2, n, res, h->d_dt_row);
    HIPCHK(h, hipGetLastError());
    return finish_map(h, H, W, res, ox, oy, std::cos(oyaw), std::sin(oyaw));  // :421-422
}

int f110_set_map_dt(f110_sim *h, const double *h_dt, int32_t H, int32_t W, double res, double ox, double oy, double oc, double os)
{
    if (!h || !h_dt) return fail(h, F110_ERR_INVALID, "f110_set_map_dt: null argument");
    if (H < 1 || W < 1 || !(res > 0)) return fail(h, F110_ERR_INVALID, "f110_set_map_dt: bad shape or resolution");
    if ((unsigned long long)(H + 3) * (unsigned long long)(W + 3) * 8ull >= 0xFFFFFFFFull) return fail(h, F110_ERR_INVALID, "distance table must stay below 4 GiB");
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    const size_t n = (size_t)H * W;
    if (h->d_dt_row) { (void)hipFree(h->d_dt_row); h->d_dt_row = nullptr; }
    TRY(dmalloc(h, &h->d_dt_row, n));
    HIPCHK(h, hipMemcpyAsync(h->d_dt_row, h_dt, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    return finish_map(h, H, W, res, ox, oy, oc, os);
}

int f110_get_map_dt(f110_sim *h, double *out)
{
    if (!h || !out) return fail(h, F110_ERR_INVALID, "null argument");
    if (!h->has_map) return fail(h, F110_ERR_NO_MAP, "Map is not set for scan simulator.");
    HIPCHK(h, hipMemcpyAsync(out, h->d_dt_row, (size_t)h->k.height * h->k.width * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return F110_OK;
}

int f110_map_shape(f110_sim *h, int32_t *H, int32_t *W)
{
    if (!h) return fail(nullptr, F110_ERR_INVALID, "null handle");
    if (!h->has_map) return fail(h, F110_ERR_NO_MAP, "Map is not set for scan simulator.");
    if (H) *H = h->k.height;
    if (W) *W = h->k.width;
    return F110_OK;
}

int f110_set_trig_tables(f110_sim *h, const double *s, const double *c, int32_t n)
{
    if (!h || !s || !c) return fail(h, F110_ERR_INVALID, "null argument");
    if (n != h->cfg.theta_dis) return fail(h, F110_ERR_INVALID, "trig tables must have theta_dis=%d entries (got %d)", h->cfg.theta_dis, n);
    Scratch sc(h);
    double *ds = nullptr, *dc = nullptr;
    TRY(sc.up(s, (size_t)n, &ds));
    TRY(sc.up(c, (size_t)n, &dc));
    hipLaunchKernelGGL(k_interleave_cs, grid1d(n, 256), dim3(256), 0, h->stream, ds, dc, n, h->d_cs);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return F110_OK;
}

int f110_set_beam_tables(f110_sim *h, const double *sa, const double *co, const double *sd, int32_t B)
{
    if (!h || !sa || !co || !sd) return fail(h, F110_ERR_INVALID, "null argument");
    if (B != h->cfg.num_beams) return fail(h, F110_ERR_INVALID, "beam tables must have num_beams=%d entries (got %d)", h->cfg.num_beams, B);
    HIPCHK(h, hipMemcpyAsync(h->d_scan_angles, sa, sizeof(double) * B, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_beam_cos, co, sizeof(double) * B, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_side, sd, sizeof(double) * B, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    // beam spacing of THIS table (seed of the nearest-beam search in the opponent ray-cast)
    h->dev.angle_inc = (sa[B - 1] - sa[0]) / (B - 1);
    // iTTC early-out of k_scan_rays: with s = max|side|, c = max|cos|, a beam with
    // r > s + thresh*(1+1e-9)*c*|v| has (r - side_b) > thresh*(1+1e-12)*|v*cos_b| and cannot hit.
    {
        double smax = 0.0, cmax = 0.0;
        bool finite = true;
        for (int i = 0; i < B; ++i) {
            if (!(sd[i] == sd[i]) || !(co[i] == co[i])) finite = false;
            smax = std::max(smax, std::fabs(sd[i]));
            cmax = std::max(cmax, std::fabs(co[i]));
        }
        h->ttc_side_max = finite ? smax : INFINITY;
        h->ttc_cos_max = finite ? cmax : INFINITY;
    }
    return F110_OK;
}

int f110_set_params(f110_sim *h, int32_t agent_idx, const double *p)
{
    if (!h || !p) return fail(h, F110_ERR_INVALID, "null argument");
    const int A = h->cfg.num_agents;
    if (agent_idx >= A) return fail(h, F110_ERR_INVALID, "Index given is out of bounds for list of agents.");
    for (int a = 0; a < A; ++a)
        if (agent_idx < 0 || agent_idx == a)
            HIPCHK(h, hipMemcpyAsync(h->d_params + (size_t)a * NPARAMS, p, sizeof(double) * NPARAMS, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return F110_OK;
}

int f110_set_noise_table(f110_sim *h, const double *noise, int32_t rows, int32_t B)
{
    if (!h) return fail(nullptr, F110_ERR_INVALID, "null handle");
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->d_noise) { (void)hipFree(h->d_noise); h->d_noise = nullptr; }
    h->dev.noise = nullptr;
    h->dev.noise_rows = 0;
    if (!noise || rows <= 0) return F110_OK;
    if (B != h->cfg.num_beams) return fail(h, F110_ERR_INVALID, "noise table must have num_beams=%d columns (got %d)", h->cfg.num_beams, B);
    TRY(dmalloc(h, &h->d_noise, (size_t)rows * B));
    HIPCHK(h, hipMemcpyAsync(h->d_noise, noise, sizeof(double) * (size_t)rows * B, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->dev.noise = h->d_noise;
    h->dev.noise_rows = rows;
    return F110_OK;
}

// ---- reset / step ------------------------------------------------------------------------
int f110_reset_device(f110_sim *h, const double *d_poses, const uint8_t *d_env_mask)
{
    if (!h || !d_poses) return fail(h, F110_ERR_INVALID, "null argument");
    hipLaunchKernelGGL(k_reset, grid1d(h->N, 256), dim3(256), 0, h->stream, h->dev, d_poses, d_env_mask);
    HIPCHK(h, hipGetLastError());
    return F110_OK;
}

int f110_reset_collided_device(f110_sim *h, const double *d_start_poses, int32_t ego_idx, int32_t *d_count)
{
    if (!h || !d_start_poses) return fail(h, F110_ERR_INVALID, "null argument");
    if (ego_idx < 0 || ego_idx >= h->cfg.num_agents) return fail(h, F110_ERR_INVALID, "Index given is out of bounds for list of agents.");
    hipLaunchKernelGGL(k_reset_collided, grid1d(h->N, 256), dim3(256), 0, h->stream, h->dev, d_start_poses, ego_idx, d_count);
    HIPCHK(h, hipGetLastError());
    return F110_OK;
}

// ---- optional observation gather over RCCL / xGMI (BASELINE config 4) ---------------------------
// The step path has no collective: environments never interact and each rank's policy consumes
// its observations on the owning GPU.  For consumers that want every rank's scans on every GPU,
// this all-gathers them; RCCL is resolved at run time (dlopen) so the library has no link-time
// dependency on it and, inside a process that already loaded torch's RCCL, shares that copy.
namespace {
struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi *rccl_api()
{
    static RcclApi api;
    static bool tried = false;
    if (!tried) {
        tried = true;
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (api.lib) break;
        }
        if (api.lib) {
            api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(api.lib, "ncclGetUniqueId"));
            api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(api.lib, "ncclCommInitRank"));
            api.AllGather = reinterpret_cast<decltype(api.AllGather)>(dlsym(api.lib, "ncclAllGather"));
            api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(api.lib, "ncclCommDestroy"));
            api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(api.lib, "ncclGetErrorString"));
        }
    }
    const bool ok = api.lib && api.GetUniqueId && api.CommInitRank && api.AllGather && api.CommDestroy && api.GetErrorString;
    return ok ? &api : nullptr;
}
}  // namespace

int f110_comm_unique_id(void *out_id128)
{
    if (!out_id128) return fail(nullptr, F110_ERR_INVALID, "null argument");
    RcclApi *r = rccl_api();
    if (!r) return fail(nullptr, F110_ERR_STATE, "RCCL (librccl.so) could not be loaded");
    ncclUniqueId id;
    const ncclResult_t rc = r->GetUniqueId(&id);
    if (rc != ncclSuccess) return fail(nullptr, F110_ERR_HIP, "ncclGetUniqueId failed: %s", r->GetErrorString(rc));
    static_assert(sizeof(id) == F110_COMM_ID_BYTES, "ncclUniqueId size");
    memcpy(out_id128, &id, sizeof(id));
    return F110_OK;
}

int f110_comm_init(f110_sim *h, int32_t n_ranks, int32_t rank, const void *id128)
{
    if (!h || !id128 || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(h, F110_ERR_INVALID, "f110_comm_init: bad argument");
    RcclApi *r = rccl_api();
    if (!r) return fail(h, F110_ERR_STATE, "RCCL (librccl.so) could not be loaded");
    if (h->comm) return fail(h, F110_ERR_STATE, "communicator already initialised");
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    const ncclResult_t rc = r->CommInitRank(&h->comm, n_ranks, id, rank);
    if (rc != ncclSuccess) {
        h->comm = nullptr;
        return fail(h, F110_ERR_HIP, "ncclCommInitRank failed: %s", r->GetErrorString(rc));
    }
    h->comm_ranks = n_ranks;
    return F110_OK;
}

int f110_comm_all_gather_scans(f110_sim *h, void *d_recv)
{
    if (!h || !d_recv) return fail(h, F110_ERR_INVALID, "null argument");
    if (!h->comm) return fail(h, F110_ERR_STATE, "f110_comm_init has not been called");
    RcclApi *r = rccl_api();
    const size_t count = (size_t)h->N * h->cfg.num_beams;
    const ncclResult_t rc = r->AllGather(h->dev.scans, d_recv, count, ncclFloat64, h->comm, h->stream);
    if (rc != ncclSuccess) return fail(h, F110_ERR_HIP, "ncclAllGather failed: %s", r->GetErrorString(rc));
    return F110_OK;
}

int f110_comm_destroy(f110_sim *h)
{
    if (!h) return fail(nullptr, F110_ERR_INVALID, "null handle");
    if (h->comm) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        RcclApi *r = rccl_api();
        if (r) (void)r->CommDestroy(h->comm);
        h->comm = nullptr;
        h->comm_ranks = 0;
    }
    return F110_OK;
}

// ---- episode logic (f110_env.py:204-246,306-338) -----------------------------------------------
int f110_episode_init(f110_sim *h, int32_t ego_idx)
{
    if (!h) return fail(nullptr, F110_ERR_INVALID, "null handle");
    if (ego_idx < 0 || ego_idx >= h->cfg.num_agents) return fail(h, F110_ERR_INVALID, "Index given is out of bounds for list of agents.");
    const size_t N = (size_t)h->N, E = (size_t)h->cfg.num_envs;
    EpisodeArrays &ep = h->ep;
    if (!h->has_episode) {
        TRY(dmalloc(h, &ep.start_poses, 3 * N));
        TRY(dmalloc(h, &ep.rot, 4 * E));
        TRY(dmalloc(h, &ep.current_time, E));
        TRY(dmalloc(h, &ep.near_start, N));
        TRY(dmalloc(h, &ep.toggle, N));
        TRY(dmalloc(h, &ep.lap_count, N));
        TRY(dmalloc(h, &ep.lap_time, N));
        TRY(dmalloc(h, &ep.done, E));
        TRY(dmalloc(h, &ep.checkpoint, N));
        TRY(dmalloc(h, &h->d_rot_stage, 4 * E));
        HIPCHK(h, hipMemsetAsync(ep.start_poses, 0, 3 * N * sizeof(double), h->stream));
        HIPCHK(h, hipMemsetAsync(ep.rot, 0, 4 * E * sizeof(double), h->stream));
        HIPCHK(h, hipMemsetAsync(ep.current_time, 0, E * sizeof(double), h->stream));
        HIPCHK(h, hipMemsetAsync(ep.near_start, 1, N, h->stream));
        HIPCHK(h, hipMemsetAsync(ep.toggle, 0, N * sizeof(double), h->stream));
        HIPCHK(h, hipMemsetAsync(ep.lap_count, 0, N * sizeof(double), h->stream));
        HIPCHK(h, hipMemsetAsync(ep.lap_time, 0, N * sizeof(double), h->stream));
        HIPCHK(h, hipMemsetAsync(ep.done, 0, E, h->stream));
        HIPCHK(h, hipMemsetAsync(ep.checkpoint, 0, N, h->stream));
        h->has_episode = true;
    }
    ep.ego_idx = ego_idx;
    ep.timestep = h->cfg.time_step;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return F110_OK;
}

int f110_episode_reset(f110_sim *h, const double *poses, const double *rot, const uint8_t *env_mask)
{
    if (!h || !poses || !rot) return fail(h, F110_ERR_INVALID, "null argument");
    if (!h->has_episode) return fail(h, F110_ERR_STATE, "f110_episode_init has not been called");
    HIPCHK(h, hipMemcpyAsync(h->d_poses, poses, sizeof(double) * 3 * h->N, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_rot_stage, rot, sizeof(double) * 4 * h->cfg.num_envs, hipMemcpyHostToDevice, h->stream));
    if (env_mask) HIPCHK(h, hipMemcpyAsync(h->d_mask, env_mask, (size_t)h->cfg.num_envs, hipMemcpyHostToDevice, h->stream));
    const uint8_t *dm = env_mask ? h->d_mask : nullptr;
    hipLaunchKernelGGL(k_episode_reset, grid1d(h->N, 256), dim3(256), 0, h->stream, h->dev, h->ep, h->d_poses, h->d_rot_stage, dm);
    hipLaunchKernelGGL(k_reset, grid1d(h->N, 256), dim3(256), 0, h->stream, h->dev, h->d_poses, dm);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return F110_OK;
}

int f110_episode_step_device(f110_sim *h, const double *d_actions)
{
    if (!h) return fail(nullptr, F110_ERR_INVALID, "null handle");
    if (!h->has_episode) return fail(h, F110_ERR_STATE, "f110_episode_init has not been called");
    TRY(f110_step_device(h, d_actions));
    hipLaunchKernelGGL(k_episode, grid1d(h->cfg.num_envs, 256), dim3(256), 0, h->stream, h->dev, h->ep, h->cfg.num_envs);
    HIPCHK(h, hipGetLastError());
    return F110_OK;
}

int f110_episode_reset_done_device(f110_sim *h, int32_t *d_count)
{
    if (!h) return fail(nullptr, F110_ERR_INVALID, "null handle");
    if (!h->has_episode) return fail(h, F110_ERR_STATE, "f110_episode_init has not been called");
    hipLaunchKernelGGL(k_episode_reset_done, grid1d(h->N, 256), dim3(256), 0, h->stream, h->dev, h->ep, d_count);
    hipLaunchKernelGGL(k_episode_clear_done, grid1d(h->cfg.num_envs, 256), dim3(256), 0, h->stream, h->ep, h->cfg.num_envs);
    HIPCHK(h, hipGetLastError());
    return F110_OK;
}

int f110_episode_get(f110_sim *h, const f110_episode_host *o)
{
    if (!h || !o) return fail(h, F110_ERR_INVALID, "null argument");
    if (!h->has_episode) return fail(h, F110_ERR_STATE, "f110_episode_init has not been called");
    const size_t N = (size_t)h->N, E = (size_t)h->cfg.num_envs;
    const EpisodeArrays &ep = h->ep;
    TRY(copy_col(h, o->lap_times, ep.lap_time, N));
    TRY(copy_col(h, o->lap_counts, ep.lap_count, N));
    TRY(copy_col(h, o->toggles, ep.toggle, N));
    TRY(copy_col(h, o->current_time, ep.current_time, E));
    if (o->near_starts) HIPCHK(h, hipMemcpyAsync(o->near_starts, ep.near_start, N, hipMemcpyDeviceToHost, h->stream));
    if (o->done) HIPCHK(h, hipMemcpyAsync(o->done, ep.done, E, hipMemcpyDeviceToHost, h->stream));
    if (o->checkpoint_done) HIPCHK(h, hipMemcpyAsync(o->checkpoint_done, ep.checkpoint, N, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return F110_OK;
}

int f110_episode_device_views(f110_sim *h, f110_episode_views *v)
{
    if (!h || !v) return fail(h, F110_ERR_INVALID, "null argument");
    if (!h->has_episode) return fail(h, F110_ERR_STATE, "f110_episode_init has not been called");
    v->done = h->ep.done;
    v->checkpoint_done = h->ep.checkpoint;
    v->lap_times = h->ep.lap_time;
    v->lap_counts = h->ep.lap_count;
    v->toggles = h->ep.toggle;
    v->current_time = h->ep.current_time;
    return F110_OK;
}

int f110_reset(f110_sim *h, const double *poses, const uint8_t *env_mask)
{
    if (!h || !poses) return fail(h, F110_ERR_INVALID, "null argument");
    HIPCHK(h, hipMemcpyAsync(h->d_poses, poses, sizeof(double) * 3 * h->N, hipMemcpyHostToDevice, h->stream));
    if (env_mask) HIPCHK(h, hipMemcpyAsync(h->d_mask, env_mask, (size_t)h->cfg.num_envs, hipMemcpyHostToDevice, h->stream));
    TRY(f110_reset_device(h, h->d_poses, env_mask ? h->d_mask : nullptr));
    HIPCHK(h, hipStreamSynchronize(h->stream));  // host buffers are consumed on return
    return F110_OK;
}

static hipEvent_t prof_event(f110_sim *h)
{
    if (h->prof_used == h->prof_events.size()) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        h->prof_events.push_back(e);
    }
    return h->prof_events[h->prof_used++];
}

int f110_step_device(f110_sim *h, const double *d_actions)
{
    if (!h || !d_actions) return fail(h, F110_ERR_INVALID, "null argument");
    if (!h->has_map) return fail(h, F110_ERR_NO_MAP, "Map is not set for scan simulator.");
    const int N = h->N;
    const bool prof = h->profiling && h->prof_used + 4 <= 4 * 65536;
    hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr, e3 = nullptr;
    if (prof) {
        e0 = prof_event(h); e1 = prof_event(h); e2 = prof_event(h); e3 = prof_event(h);
        if (!e0 || !e1 || !e2 || !e3) return fail(h, F110_ERR_HIP, "hipEventCreate failed");
        HIPCHK(h, hipEventRecord(e0, h->stream));
    }
    hipLaunchKernelGGL(k_integrate, grid1d(N, 256), dim3(256), 0, h->stream, h->dev, h->k, d_actions);
    // k_collide only feeds k_finalize, k_scan_rays only needs k_integrate: run the two side by
    // side (second stream, event fork/join) so the pair test + window set-up hides under the scan
    const bool multi = h->cfg.num_agents > 1;
    if (multi) {
        HIPCHK(h, hipEventRecord(h->ev_integrated, h->stream));
        HIPCHK(h, hipStreamWaitEvent(h->side_stream, h->ev_integrated, 0));
        hipLaunchKernelGGL(k_collide, grid1d(N, 256), dim3(256), 0, h->side_stream, h->dev, h->k.num_beams);
        HIPCHK(h, hipEventRecord(h->ev_collided, h->side_stream));
    }
    if (prof) HIPCHK(h, hipEventRecord(e1, h->stream));
    {
        RayJob j{};
        j.n_rays = (uint32_t)N * (uint32_t)h->k.num_beams;
        j.n_poses = N;
        j.pose_x = h->dev.scan_pose;
        j.pose_y = h->dev.scan_pose + N;
        j.dir_start = h->dev.dir_start;
        j.ranges = h->dev.scans;
        j.hdr = h->dev.ray_hdr;
        j.noise = h->dev.noise;
        j.ttc_side_max = h->ttc_side_max;
        j.ttc_k = h->dev.ttc_thresh * (1.0 + 1e-9) * h->ttc_cos_max;
        j.beam_cos = h->dev.beam_cos;
        j.side_dist = h->dev.side_dist;
        j.wall_flag = h->dev.in_collision;
        j.ttc_thresh = h->dev.ttc_thresh;
        j.div_magic = h->step_magic;
        j.div_shift = h->step_shift;
        scan_rays_fn fn = pick_rays<true>(h->k, h->cfg.map_layout);
        if (h->dir_stride > 0) {
            RayJob jd = j;  // pass 1: one ray per (agent, distinct direction)
            jd.n_rays = (uint32_t)N * (uint32_t)h->dir_stride;
            jd.ranges = h->d_dir_ranges;
            jd.dir_mode = 1;
            jd.dir_stride = h->dir_stride;
            jd.div_magic = h->dir_magic;
            jd.div_shift = h->dir_shift;
            const dim3 gd = rays_grid(jd, h->scan_block, h->scan_tasks_per_wave);
            hipLaunchKernelGGL(fn, gd, dim3(h->scan_block), 0, h->stream, jd, h->k);
            j.dir_stride = h->dir_stride;  // pass 2: every beam picks its direction's range
            j.dir_ranges = h->d_dir_ranges;
            hipLaunchKernelGGL(k_expand_beams, grid1d(j.n_rays, 256), dim3(256), 0, h->stream, j, h->k);
        } else {
            const dim3 grid = rays_grid(j, h->scan_block, h->scan_tasks_per_wave);
            hipLaunchKernelGGL(fn, grid, dim3(h->scan_block), 0, h->stream, j, h->k);
        }
    }
    if (prof) HIPCHK(h, hipEventRecord(e2, h->stream));
    if (multi) HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_collided, 0));
    if (multi)
        hipLaunchKernelGGL(k_finalize, dim3(N), dim3(64), 0, h->stream, h->dev, h->k.num_beams);
    else
        hipLaunchKernelGGL(k_finalize_solo, grid1d(N, 256), dim3(256), 0, h->stream, h->dev);
    if (prof) HIPCHK(h, hipEventRecord(e3, h->stream));
    HIPCHK(h, hipGetLastError());
    return F110_OK;
}

int f110_step(f110_sim *h, const double *actions)
{
    if (!h || !actions) return fail(h, F110_ERR_INVALID, "null argument");
    if (!h->has_map) return fail(h, F110_ERR_NO_MAP, "Map is not set for scan simulator.");
    HIPCHK(h, hipMemcpyAsync(h->d_actions, actions, sizeof(double) * 2 * h->N, hipMemcpyHostToDevice, h->stream));
    TRY(f110_step_device(h, h->d_actions));
    HIPCHK(h, hipStreamSynchronize(h->stream));  // pageable host buffer: consumed on return
    return F110_OK;
}

// ---- read-back ---------------------------------------------------------------------------
static int copy_col(f110_sim *h, double *dst, const double *src, size_t n)
{
    if (dst) HIPCHK(h, hipMemcpyAsync(dst, src, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    return F110_OK;
}

int f110_get_obs(f110_sim *h, const f110_obs_host *o)
{
    if (!h || !o) return fail(h, F110_ERR_INVALID, "null argument");
    const size_t N = (size_t)h->N;
    const AgentArrays &d = h->dev;
    TRY(copy_col(h, o->scans, d.scans, N * h->cfg.num_beams));
    TRY(copy_col(h, o->poses_x, d.state, N));
    TRY(copy_col(h, o->poses_y, d.state + N, N));
    TRY(copy_col(h, o->poses_theta, d.state + 4 * N, N));
    TRY(copy_col(h, o->linear_vels_x, d.state + 3 * N, N));
    TRY(copy_col(h, o->ang_vels_z, d.state + 5 * N, N));
    TRY(copy_col(h, o->collisions, d.collisions, N));
    TRY(copy_col(h, o->collision_idx, d.collision_idx, N));
    std::vector<double> soa, poses;
    if (o->state) {
        soa.resize(7 * N);
        HIPCHK(h, hipMemcpyAsync(soa.data(), d.state, 7 * N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    }
    if (o->agent_poses) {
        poses.resize(3 * N);
        HIPCHK(h, hipMemcpyAsync(poses.data(), d.snap_pose, 3 * N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    }
    if (o->in_collision) HIPCHK(h, hipMemcpyAsync(o->in_collision, d.in_collision, N * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    if (o->step_count) HIPCHK(h, hipMemcpyAsync(o->step_count, d.step_count, N * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (o->state)
        for (size_t i = 0; i < N; ++i)
            for (int c = 0; c < 7; ++c) o->state[7 * i + c] = soa[(size_t)c * N + i];
    if (o->agent_poses)
        for (size_t i = 0; i < N; ++i)
            for (int c = 0; c < 3; ++c) o->agent_poses[3 * i + c] = poses[(size_t)c * N + i];
    return F110_OK;
}

int f110_set_state(f110_sim *h, const double *state7, const double *steer_buf, const int32_t *buf_count)
{
    if (!h || !state7) return fail(h, F110_ERR_INVALID, "null argument");
    const size_t N = (size_t)h->N;
    std::vector<double> soa(7 * N);
    for (size_t i = 0; i < N; ++i)
        for (int c = 0; c < 7; ++c) soa[(size_t)c * N + i] = state7[7 * i + c];
    HIPCHK(h, hipMemcpyAsync(h->dev.state, soa.data(), 7 * N * sizeof(double), hipMemcpyHostToDevice, h->stream));
    std::vector<double> sb;
    if (steer_buf) {
        sb.resize(2 * N);
        for (size_t i = 0; i < N; ++i) {
            sb[i] = steer_buf[2 * i];
            sb[N + i] = steer_buf[2 * i + 1];
        }
        HIPCHK(h, hipMemcpyAsync(h->dev.steer_buf, sb.data(), 2 * N * sizeof(double), hipMemcpyHostToDevice, h->stream));
    }
    if (buf_count) HIPCHK(h, hipMemcpyAsync(h->dev.buf_cnt, buf_count, N * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return F110_OK;
}

int f110_get_device_views(f110_sim *h, f110_device_views *v)
{
    if (!h || !v) return fail(h, F110_ERR_INVALID, "null argument");
    v->scans = h->dev.scans;
    v->state = h->dev.state;
    v->agent_poses = h->dev.snap_pose;
    v->collisions = h->dev.collisions;
    v->collision_idx = h->dev.collision_idx;
    v->in_collision = h->dev.in_collision;
    v->step_count = h->dev.step_count;
    v->stream = (void *)h->stream;
    return F110_OK;
}

int f110_device_alloc(f110_sim *h, size_t bytes, void **out)
{
    if (!h || !out) return fail(h, F110_ERR_INVALID, "null argument");
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    HIPCHK(h, hipMalloc(out, bytes ? bytes : 8));
    return F110_OK;
}

int f110_device_free(f110_sim *h, void *p)
{
    if (!h) return fail(nullptr, F110_ERR_INVALID, "null handle");
    if (p) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        HIPCHK(h, hipFree(p));
    }
    return F110_OK;
}

int f110_memcpy_h2d(f110_sim *h, void *dst, const void *src, size_t bytes)
{
    if (!h || !dst || !src) return fail(h, F110_ERR_INVALID, "null argument");
    HIPCHK(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return F110_OK;
}

int f110_memcpy_d2h(f110_sim *h, void *dst, const void *src, size_t bytes)
{
    if (!h || !dst || !src) return fail(h, F110_ERR_INVALID, "null argument");
    HIPCHK(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return F110_OK;
}

// ---- timing ------------------------------------------------------------------------------
int f110_timer_begin(f110_sim *h)
{
    if (!h) return fail(nullptr, F110_ERR_INVALID, "null handle");
    HIPCHK(h, hipEventRecord(h->ev_begin, h->stream));
    return F110_OK;
}

int f110_timer_end_ms(f110_sim *h, double *ms)
{
    if (!h || !ms) return fail(h, F110_ERR_INVALID, "null argument");
    HIPCHK(h, hipEventRecord(h->ev_end, h->stream));
    HIPCHK(h, hipEventSynchronize(h->ev_end));
    float f = 0.f;
    HIPCHK(h, hipEventElapsedTime(&f, h->ev_begin, h->ev_end));
    *ms = (double)f;
    return F110_OK;
}

int f110_profile_kernels(f110_sim *h, int32_t enable)
{
    if (!h) return fail(nullptr, F110_ERR_INVALID, "null handle");
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->profiling = enable != 0;
    h->prof_used = 0;
    return F110_OK;
}

int f110_profile_read(f110_sim *h, int32_t *n_launches, double *scan_ms, double *dyn_ms, double *fin_ms)
{
    if (!h) return fail(nullptr, F110_ERR_INVALID, "null handle");
    HIPCHK(h, hipStreamSynchronize(h->stream));
    double s = 0, dsum = 0, fsum = 0;
    int n = 0;
    for (size_t i = 0; i + 4 <= h->prof_used; i += 4) {
        float a = 0.f, b = 0.f, c = 0.f;
        HIPCHK(h, hipEventElapsedTime(&a, h->prof_events[i], h->prof_events[i + 1]));
        HIPCHK(h, hipEventElapsedTime(&b, h->prof_events[i + 1], h->prof_events[i + 2]));
        HIPCHK(h, hipEventElapsedTime(&c, h->prof_events[i + 2], h->prof_events[i + 3]));
        dsum += a;
        s += b;
        fsum += c;
        ++n;
    }
    if (fin_ms) *fin_ms = fsum;
    if (n_launches) *n_launches = n;
    if (scan_ms) *scan_ms = s;
    if (dyn_ms) *dyn_ms = dsum;
    return F110_OK;
}

// ---- unit entry points ---------------------------------------------------------------------
int f110_scan_batch(f110_sim *h, const double *poses, int32_t m, double *ranges, int32_t *hit_rc, int64_t *lookups)
{
    if (!h || !poses || !ranges || m < 0) return fail(h, F110_ERR_INVALID, "f110_scan_batch: bad argument");
    if (!h->has_map) return fail(h, F110_ERR_NO_MAP, "Map is not set for scan simulator.");
    if (m == 0) return F110_OK;
    const size_t B = (size_t)h->cfg.num_beams;
    Scratch s(h);
    double *dp = nullptr, *dr = nullptr;
    int32_t *dh = nullptr;
    unsigned long long *dl = nullptr;
    TRY(s.up(poses, (size_t)3 * m, &dp));
    TRY(s.up<double>(nullptr, (size_t)m * B, &dr));
    if (hit_rc) TRY(s.up<int32_t>(nullptr, (size_t)m * B * 2, &dh));
    if (lookups) {
        TRY(s.up<unsigned long long>(nullptr, (size_t)m, &dl));
        HIPCHK(h, hipMemsetAsync(dl, 0, sizeof(unsigned long long) * m, h->stream));
    }
    double *dpx = nullptr, *dpy = nullptr, *dst = nullptr;
    TRY(s.up<double>(nullptr, (size_t)m, &dpx));
    TRY(s.up<double>(nullptr, (size_t)m, &dpy));
    TRY(s.up<double>(nullptr, (size_t)m, &dst));
    hipLaunchKernelGGL(k_prepare_poses, grid1d(m, 128), dim3(128), 0, h->stream, h->k, dp, m, dpx, dpy, dst);
    RayJob j{};
    j.n_rays = (uint32_t)m * (uint32_t)B;
    j.n_poses = m;
    j.pose_x = dpx;
    j.pose_y = dpy;
    j.dir_start = dst;
    j.ranges = dr;
    j.hit_rc = dh;
    j.lookups = dl;
    set_div_magic(j, (uint32_t)B);
    scan_rays_fn fn = pick_rays<false>(h->k, h->cfg.map_layout);
    const dim3 grid = rays_grid(j, h->scan_block, h->scan_tasks_per_wave);
    hipLaunchKernelGGL(fn, grid, dim3(h->scan_block), 0, h->stream, j, h->k);
    HIPCHK(h, hipGetLastError());
    TRY(s.down(ranges, dr, (size_t)m * B));
    if (hit_rc) TRY(s.down(hit_rc, dh, (size_t)m * B * 2));
    if (lookups) TRY(s.down(reinterpret_cast<unsigned long long *>(lookups), dl, (size_t)m));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return F110_OK;
}

int f110_beam_dir_index_batch(f110_sim *h, const double *thetas, int32_t m, int32_t *idx)
{
    if (!h || !thetas || !idx || m < 0) return fail(h, F110_ERR_INVALID, "bad argument");
    if (m == 0) return F110_OK;
    Scratch s(h);
    double *dt = nullptr;
    int32_t *di = nullptr;
    TRY(s.up(thetas, (size_t)m, &dt));
    TRY(s.up<int32_t>(nullptr, (size_t)m * h->cfg.num_beams, &di));
    hipLaunchKernelGGL(k_dir_index_unit, dim3(m), dim3(128), 0, h->stream, h->k, dt, m, di);
    HIPCHK(h, hipGetLastError());
    TRY(s.down(idx, di, (size_t)m * h->cfg.num_beams));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return F110_OK;
}

int f110_dynamics_batch(f110_sim *h, const double *x, const double *u, const double *params, int32_t m, double *f_st, double *f_ks)
{
    if (!h || !x || !u || !params || !f_st || !f_ks || m < 0) return fail(h, F110_ERR_INVALID, "bad argument");
    if (m == 0) return F110_OK;
    Scratch s(h);
    double *dx, *du, *dp, *dfs, *dfk;
    TRY(s.up(x, (size_t)7 * m, &dx));
    TRY(s.up(u, (size_t)2 * m, &du));
    TRY(s.up(params, (size_t)NPARAMS, &dp));
    TRY(s.up<double>(nullptr, (size_t)7 * m, &dfs));
    TRY(s.up<double>(nullptr, (size_t)5 * m, &dfk));
    hipLaunchKernelGGL(k_dynamics_unit, grid1d(m, 128), dim3(128), 0, h->stream, dx, du, dp, m, dfs, dfk);
    HIPCHK(h, hipGetLastError());
    TRY(s.down(f_st, dfs, (size_t)7 * m));
    TRY(s.down(f_ks, dfk, (size_t)5 * m));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return F110_OK;
}

int f110_pid_batch(f110_sim *h, const double *in, const double *params, int32_t m, double *out)
{
    if (!h || !in || !params || !out || m < 0) return fail(h, F110_ERR_INVALID, "bad argument");
    if (m == 0) return F110_OK;
    Scratch s(h);
    double *di, *dp, *dout;
    TRY(s.up(in, (size_t)4 * m, &di));
    TRY(s.up(params, (size_t)NPARAMS, &dp));
    TRY(s.up<double>(nullptr, (size_t)2 * m, &dout));
    hipLaunchKernelGGL(k_pid_unit, grid1d(m, 128), dim3(128), 0, h->stream, di, dp, m, dout);
    HIPCHK(h, hipGetLastError());
    TRY(s.down(out, dout, (size_t)2 * m));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return F110_OK;
}

int f110_update_pose_batch(f110_sim *h, const double *s0, const double *b0, const int32_t *c0, const double *act,
                           const double *params, double dt, int32_t integ, double lidar_dist, int32_t m, double *s1,
                           double *b1, int32_t *c1, double *spose)
{
    if (!h || !s0 || !b0 || !c0 || !act || !params || !s1 || !b1 || !c1 || !spose || m < 0) return fail(h, F110_ERR_INVALID, "bad argument");
    if (integ != F110_INTEGRATOR_RK4 && integ != F110_INTEGRATOR_EULER) return fail(h, F110_ERR_INVALID, "Invalid Integrator Specified. Please choose RK4 or Euler");
    if (m == 0) return F110_OK;
    Scratch s(h);
    double *ds0, *db0, *dact, *dp, *ds1, *db1, *dsp;
    int32_t *dc0, *dc1;
    TRY(s.up(s0, (size_t)7 * m, &ds0));
    TRY(s.up(b0, (size_t)2 * m, &db0));
    TRY(s.up(c0, (size_t)m, &dc0));
    TRY(s.up(act, (size_t)2 * m, &dact));
    TRY(s.up(params, (size_t)NPARAMS, &dp));
    TRY(s.up<double>(nullptr, (size_t)7 * m, &ds1));
    TRY(s.up<double>(nullptr, (size_t)2 * m, &db1));
    TRY(s.up<int32_t>(nullptr, (size_t)m, &dc1));
    TRY(s.up<double>(nullptr, (size_t)3 * m, &dsp));
    hipLaunchKernelGGL(k_update_pose_unit, grid1d(m, 128), dim3(128), 0, h->stream, ds0, db0, dc0, dact, dp, dt, integ, lidar_dist, m, ds1, db1, dc1, dsp);
    HIPCHK(h, hipGetLastError());
    TRY(s.down(s1, ds1, (size_t)7 * m));
    TRY(s.down(b1, db1, (size_t)2 * m));
    TRY(s.down(c1, dc1, (size_t)m));
    TRY(s.down(spose, dsp, (size_t)3 * m));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return F110_OK;
}

int f110_get_vertices_batch(f110_sim *h, const double *poses, double length, double width, int32_t m, double *verts)
{
    if (!h || !poses || !verts || m < 0) return fail(h, F110_ERR_INVALID, "bad argument");
    if (m == 0) return F110_OK;
    Scratch s(h);
    double *dp, *dv;
    TRY(s.up(poses, (size_t)3 * m, &dp));
    TRY(s.up<double>(nullptr, (size_t)8 * m, &dv));
    hipLaunchKernelGGL(k_vertices_unit, grid1d(m, 128), dim3(128), 0, h->stream, dp, length, width, m, dv);
    HIPCHK(h, hipGetLastError());
    TRY(s.down(verts, dv, (size_t)8 * m));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return F110_OK;
}

int f110_gjk_batch(f110_sim *h, const double *va, const double *vb, int32_t m, int32_t *flags)
{
    if (!h || !va || !vb || !flags || m < 0) return fail(h, F110_ERR_INVALID, "bad argument");
    if (m == 0) return F110_OK;
    Scratch s(h);
    double *da, *db;
    int32_t *df;
    TRY(s.up(va, (size_t)8 * m, &da));
    TRY(s.up(vb, (size_t)8 * m, &db));
    TRY(s.up<int32_t>(nullptr, (size_t)m, &df));
    hipLaunchKernelGGL(k_gjk_unit, grid1d(m, 128), dim3(128), 0, h->stream, da, db, m, df);
    HIPCHK(h, hipGetLastError());
    TRY(s.down(flags, df, (size_t)m));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return F110_OK;
}

int f110_collision_multiple_batch(f110_sim *h, const double *verts, int32_t groups, int32_t n, double *col, double *idx)
{
    if (!h || !verts || !col || !idx || groups < 0 || n < 1) return fail(h, F110_ERR_INVALID, "bad argument");
    if (groups == 0) return F110_OK;
    const size_t t = (size_t)groups * n;
    Scratch s(h);
    double *dv, *dc, *di;
    TRY(s.up(verts, 8 * t, &dv));
    TRY(s.up<double>(nullptr, t, &dc));
    TRY(s.up<double>(nullptr, t, &di));
    hipLaunchKernelGGL(k_collision_multiple_unit, grid1d(t, 128), dim3(128), 0, h->stream, dv, groups, n, dc, di);
    HIPCHK(h, hipGetLastError());
    TRY(s.down(col, dc, t));
    TRY(s.down(idx, di, t));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return F110_OK;
}

int f110_ttc_batch(f110_sim *h, const double *scans, const double *vels, int32_t m, double thresh, int32_t *flags)
{
    if (!h || !scans || !vels || !flags || m < 0) return fail(h, F110_ERR_INVALID, "bad argument");
    if (m == 0) return F110_OK;
    const size_t B = (size_t)h->cfg.num_beams;
    Scratch s(h);
    double *ds, *dv;
    int32_t *df;
    TRY(s.up(scans, (size_t)m * B, &ds));
    TRY(s.up(vels, (size_t)m, &dv));
    TRY(s.up<int32_t>(nullptr, (size_t)m, &df));
    hipLaunchKernelGGL(k_ttc_unit, dim3(m), dim3(128), 0, h->stream, ds, dv, m, (int)B, h->d_beam_cos, h->d_side, thresh, df);
    HIPCHK(h, hipGetLastError());
    TRY(s.down(flags, df, (size_t)m));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return F110_OK;
}

int f110_raycast_batch(f110_sim *h, const double *ego, const double *verts, int32_t m, double *scans, int32_t *minmax)
{
    if (!h || !ego || !verts || !scans || m < 0) return fail(h, F110_ERR_INVALID, "bad argument");
    if (m == 0) return F110_OK;
    const size_t B = (size_t)h->cfg.num_beams;
    Scratch s(h);
    double *de, *dv, *ds;
    int32_t *dm = nullptr;
    TRY(s.up(ego, (size_t)3 * m, &de));
    TRY(s.up(verts, (size_t)8 * m, &dv));
    TRY(s.up(scans, (size_t)m * B, &ds));
    if (minmax) TRY(s.up<int32_t>(nullptr, (size_t)2 * m, &dm));
    hipLaunchKernelGGL(k_raycast_unit, dim3(m), dim3(128), 0, h->stream, de, dv, m, (int)B, h->d_scan_angles, h->dev.angle_inc, ds, dm);
    HIPCHK(h, hipGetLastError());
    TRY(s.down(scans, ds, (size_t)m * B));
    if (minmax) TRY(s.down(minmax, dm, (size_t)2 * m));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return F110_OK;
}

int f110_get_range_batch(f110_sim *h, const double *in, int32_t m, double *out)
{
    if (!h || !in || !out || m < 0) return fail(h, F110_ERR_INVALID, "bad argument");
    if (m == 0) return F110_OK;
    Scratch s(h);
    double *di, *dout;
    TRY(s.up(in, (size_t)8 * m, &di));
    TRY(s.up<double>(nullptr, (size_t)m, &dout));
    hipLaunchKernelGGL(k_get_range_unit, grid1d(m, 128), dim3(128), 0, h->stream, di, m, dout);
    HIPCHK(h, hipGetLastError());
    TRY(s.down(out, dout, (size_t)m));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return F110_OK;
}

int f110_edt_sq(f110_sim *h, const uint8_t *img, int32_t H, int32_t W, uint32_t *d2)
{
    if (!h || !img || !d2) return fail(h, F110_ERR_INVALID, "null argument");
    if (H < 1 || W < 1 || H > 16384 || W > 16384) return fail(h, F110_ERR_INVALID, "f110_edt_sq: bad shape %dx%d", H, W);
    const size_t n = (size_t)H * W;
    Scratch s(h);
    uint8_t *dimg;
    uint32_t *dg, *dd;
    TRY(s.up(img, n, &dimg));
    TRY(s.up<uint32_t>(nullptr, n, &dg));
    TRY(s.up<uint32_t>(nullptr, n, &dd));
    hipLaunchKernelGGL(k_edt_columns, grid1d(W, 64), dim3(64), 0, h->stream, dimg, H, W, dg);
    hipLaunchKernelGGL(k_edt_rows, dim3(H), dim3(256), (size_t)W * sizeof(uint32_t), h->stream, dg, H, W, dd);
    HIPCHK(h, hipGetLastError());
    TRY(s.down(d2, dd, n));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return F110_OK;
}

}  // extern "C"
