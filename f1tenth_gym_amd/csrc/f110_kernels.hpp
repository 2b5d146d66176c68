// f110_kernels.hpp — the gfx950 kernels of the batched F1TENTH env.step() hot path.
//
// Included by f110_hip.hip (which owns the handle, the launches and the C ABI).  One step is
//   k_integrate   1 lane / agent   pid + steering delay + RK4|Euler + yaw wrap + lidar pose
//                                  (base_classes.py:256-409) + the per-agent ray header
//   k_collide     1 lane / agent   get_vertices + GJK inside each env (collision_models.py:113-260,
//                                  base_classes.py:536-550) + opponent beam windows; side stream
//   k_scan_rays   1 lane / ray, a wave holds 64 consecutive beams: sphere-trace the distance table
//                                  (laser_models.py:106-186), noise row (:450-452), iTTC predicate
//                                  (:188-217), one coalesced store.  Lean on purpose: 8 waves/SIMD.
//   (k_expand_beams: second pass when beams outnumber table directions)
//   k_finalize    1 wave / agent   wall-hit side effects (base_classes.py:246-249), collision OR
//                                  (:588-589), opponent ray-cast on the culled window (:206-227)
//   (k_finalize_pair: two-agent envs — k_collide's work done at the top of k_finalize, no side stream)
//   (k_noise_cache / k_noise_rows: the scan noise, NumPy's PCG64 + ziggurat stream, f110_rng.hpp)
// plus reset / episode-logic kernels, the unit kernels behind the parity entry points, and the
// map pipeline (flip + threshold + exact EDT + dt = res*sqrt(d2)).
// Compiled with -ffp-contract=off: float64, reference operation order, no FMA contraction.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "f110_math.hpp"
#include "f110_rng.hpp"

using namespace f110;

// What one ray needs to know about its agent, written by k_integrate: read through the scalar
// cache by k_scan_rays (a wave's 64 consecutive rays belong to at most two agents).
struct RayHdr {
    double x, y;        // lidar position (base_classes.py:407-408)
    double start;       // wrapped theta_index of beam 0 (laser_models.py:166-172)
    double vel;         // post-integration longitudinal velocity (iTTC)
    double d0;          // first table sample, shared by every beam of the scan (:129)
    int32_t noise_row;  // row of the noise table / row cache for this step; -1 = no noise; -2 = this step's
                        // noise row was generated into the agent's own scans[] row (k_noise_rows)
    int32_t map_slot;   // which registered map this agent's env runs on (0 unless f110_set_env_maps)
    int32_t pad_hdr;
    int32_t i0;         // table index of beam 0
    int32_t n_dirs;     // distinct table indices the scan's beams use (dedupe mode), else 0
    int32_t fast;       // PADDED: every sample of every ray of this scan lands inside the padded table
};
static_assert(sizeof(RayHdr) == 64, "RayHdr is read as 16-byte scalar loads");

struct AgentArrays {
    int32_t n_agents_total;  // N
    int32_t agents_per_env;  // A
    int32_t agent_begin, agent_count;  // the agents this launch covers (an env-aligned group; all: 0, N)
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
    unsigned long long *path_stats;  // diagnostics or nullptr: rays of fast scans [0] / of exact scans [2]
    // a different track per env (f110_set_env_maps), else both nullptr
    const ScanConst *maps_full;      // [n_maps] every field of every registered map (exact paths)
    const int32_t *env_map;          // [num_envs] slot of each env
    // armed by f110_set_auto_reseat: k_finalize ends with the in-place re-seat of finished envs
    const double *reseat_poses;      // [N][3] or nullptr
    int32_t *reseat_count;           // device counter or nullptr
    int32_t reseat_ego, pad_reseat;
    uint32_t *sched_count_zero;  // longest-first scan order: this step's list counter, zeroed here (or nullptr)
    int32_t *opp_window;     // [N][A][4] beam range each opponent can occupy: {lo, hi} for the live
                             //           heading and {lo0, hi0} for heading 0 (after a wall hit)
    double *opp_verts;       // [N][A][8] the opponent's box drawn with the ego's length/width
    const double *params;    // [A][18] per agent slot, or [N][18] per agent (params_per_agent)
    int32_t params_per_agent, pad_params;
    const double *noise;     // [noise_rows][B] or nullptr: the uploaded table, or the device-generated row cache
    // device RNG (f110_set_noise_rng): 0 off / table, 1 one stream shared by every agent (the reference),
    // 2 a stream per agent.  Rows below noise_rows come from the cache in shared mode; later rows and
    // per-agent streams are generated into scans[] by k_noise_rows from the agent's carried state.
    int32_t noise_rng, pad_rng;
    U128 *rng_state;              // [N] state after the agent's last generated row
    const U128 *rng_seed;         // per-agent mode: [N][2] = {state at reset, inc}
    const U128 *rng_rowstate;     // shared mode: [noise_rows + 1] state at the start of every cached row
    U128 rng_inc;                 // shared mode: the stream's increment
    const double *scan_angles, *beam_cos, *side_dist;  // [B]
    int32_t noise_rows, integrator;
    double time_step, lidar_dist, ttc_thresh, angle_inc;
    double box_length, box_width;  // Simulator.params used by check_collision (:549)
    const struct FusedHost *fused_host;   // f110_step_host, A = 2: the finalize kernel ends with the episode logic + host block (or nullptr)
    unsigned long long fused_seq;         // ... and signals this sequence number (F110_STEP_SPIN_WAIT), 0 = no signal
};

__device__ __forceinline__ VehicleParams load_params(const double *p)
{
    VehicleParams vp;
#pragma unroll
    for (int i = 0; i < NPARAMS; ++i) vp.v[i] = p[i];
    return vp;
}

// ---- episode logic + host block (f110_episode_*, f110_step_host): the structs live here because the A = 2 finalize
// kernel can carry both as its epilogue (FusedHost) ------------------------------------------------------------------
// F110Env._check_done (f110_env.py:204-246): start/finish-zone toggles, lap counts and times, done
// = ego collided or every agent has 4 toggles.
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

// f110_step_host: what a host-driven loop reads after a step (see k_host_block).  Any pointer may be nullptr.
struct HostBlock {
    double *state;          // [7][N]
    double *collisions;     // [N]
    double *collision_idx;  // [N]
    double *agent_poses;    // [3][N]
    double *lap_time, *lap_count, *toggle;  // [N]
    double *current_time;   // [E]
    int32_t *in_collision;  // [N]
    uint8_t *near_start, *checkpoint;  // [N]
    uint8_t *done;          // [E]
    // completion word (F110_STEP_SPIN_WAIT): the workgroup that finishes last stores `seq` here, after every
    // workgroup's stores have been fenced at system scope — the host polls it instead of entering the runtime
    unsigned long long *seq_host;   // page-locked
    unsigned int *blocks_done;      // device counter, returns to 0
    unsigned long long seq;
    // small batches: the scans too are stored by the kernel (a DMA copy of 17 KB costs ~36 us of latency on this
    // runtime, a kernel's stores ~2); nullptr: the caller copies them (big batches: one DMA copy at link rate)
    double *scans;                  // [N][num_beams] page-locked, or nullptr
    int32_t num_beams, pad_beams;
};

// rows [first, first + count) of the device scans into the host block, by every thread of the workgroup
__device__ __forceinline__ void host_block_copy_scans(const HostBlock &hb, const double *__restrict__ scans, size_t first, size_t count)
{
    const size_t B = (size_t)hb.num_beams, n = count * B;
    const double *__restrict__ src = scans + first * B;
    double *__restrict__ dst = hb.scans + first * B;
    // four independent loads in flight per lane, then the four stores (a dependent load -> store per iteration would
    // walk the rows one memory round trip at a time)
    size_t q = threadIdx.x;
    for (; q + 3 * (size_t)blockDim.x < n; q += 4 * (size_t)blockDim.x) {
        const double v0 = src[q], v1 = src[q + blockDim.x], v2 = src[q + 2 * (size_t)blockDim.x], v3 = src[q + 3 * (size_t)blockDim.x];
        dst[q] = v0;
        dst[q + blockDim.x] = v1;
        dst[q + 2 * (size_t)blockDim.x] = v2;
        dst[q + 3 * (size_t)blockDim.x] = v3;
    }
    for (; q < n; q += blockDim.x) dst[q] = src[q];
}

// last statement of a workgroup of k_host_block: publish the block's host stores, count, and let the last one signal
__device__ __forceinline__ void host_block_signal(const HostBlock &hb)
{
    if (!hb.seq_host) return;
    __builtin_amdgcn_s_waitcnt(0); // every wave's stores into the host block have been acknowledged (a workgroup barrier alone does not wait for them) ...
    __syncthreads();               // ... in every wave of the workgroup
    if (threadIdx.x == 0) {
        __threadfence_system();    // ... and visible system-wide before the count
        const unsigned int prev = __hip_atomic_fetch_add(hb.blocks_done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == gridDim.x - 1) {
            __hip_atomic_store(hb.blocks_done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence_system();
            __hip_atomic_store(hb.seq_host, hb.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}


// f110_step_host with the pair kernel: episode logic + host block as k_finalize_pair_roles' epilogue (one launch and one
// drain less per step than k_host_block behind it).  Lives in device memory (the handle rewrites it when the caller's
// block changes); the completion word's sequence number travels by value in AgentArrays (it changes every step).
struct FusedHost {
    EpisodeArrays ep;
    HostBlock hb;
    int32_t episode, auto_reset;
};

// ---- K1b body: pairwise body collisions inside each env + opponent beam windows ---------------
// collision_multiple visits pairs (i<j) ascending and overwrites collision_idx, so an agent's
// final index is the largest colliding partner; flags are symmetric.  Each lane evaluates the
// GJK of its pairs in the reference's (lower, higher) argument order.
// AF = 0: any number of agents per env, partner poses through pose_of(j) from memory;
// AF = 2 / 4: exactly that many, loop unrolled so pose_of(j) can index registers.
template <int AF, typename PoseOf>
__device__ __forceinline__ void collide_agent(const AgentArrays &a, int32_t B, int i, double mx, double my, double mth, PoseOf pose_of)
{
    const int A = AF ? AF : a.agents_per_env;
    const int env = i / A, me = i - env * A;
    double mine[8];
    box_vertices(mx, my, mth, a.box_length, a.box_width, mine);
    // bodies whose centres are further apart than a box diagonal (+1 mm) cannot overlap
    const double reach = sqrt(a.box_length * a.box_length + a.box_width * a.box_width) + 1e-3;
    // RaceCar.ray_cast_agents draws the opponents with the EGO's length/width (:223)
    const size_t prow = (size_t)(a.params_per_agent ? i : me) * NPARAMS;
    const double blen = a.params[prow + P_LENGTH];
    const double bwid = a.params[prow + P_WIDTH];
    const double disc_r = 0.5 * sqrt(blen * blen + bwid * bwid);
    bool hit = false;
    int partner = -1;
#pragma unroll
    for (int j = 0; j < A; ++j) {
        if (j == me) continue;
        double ox, oy, oth;
        pose_of(j, ox, oy, oth);
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

// what k_integrate leaves behind for one agent: state, delay buffer, lidar pose, pose snapshot, the ray header
// everything the ray kernel needs per agent, incl. the first table sample that all beams share (trace_ray :129 evaluated at the
// lidar position; generic exact path)
__device__ __forceinline__ RayHdr make_ray_hdr(const AgentArrays &a, const ScanConst &k, int i, const double *st, const double *sp, double start)
{
    RayHdr hd;
    ScanConst kr = k;
    kr.table = k.table_rm;
    hd.x = sp[0];
    hd.y = sp[1];
    hd.start = start;
    hd.vel = st[3];
    // a different track per env (f110_set_env_maps): this agent's map constants come from the
    // registered slot instead of the kernel argument
    hd.map_slot = a.env_map ? a.env_map[i / a.agents_per_env] : 0;
    hd.pad_hdr = 0;
    const ScanConst *km = a.env_map ? a.maps_full + hd.map_slot : &kr;
    int r0, c0;
    hd.d0 = sample_distance<LAYOUT_ROWMAJOR, false, false>(*km, nullptr, sp[0], sp[1], r0, c0);
    int row = -1;
    if (a.noise_rng) {
        // k-th scan after reset adds the k-th B-sample draw of the stream (base_classes.py:204,
        // laser_models.py:450-452): from the row cache, or generated by k_noise_rows (-2)
        row = a.step_count[i];
        if (a.noise_rng == 2 || row >= a.noise_rows) row = -2;
    } else if (a.noise_rows > 0) {
        row = a.step_count[i];
        if (row >= a.noise_rows) row %= a.noise_rows;
    }
    hd.noise_row = row;
    hd.i0 = beam_dir_index(k, start, 0);
    hd.fast = 0;
    if (km->pad) {
        double ux, uy;
        padded_position<false>(*km, sp[0], sp[1], ux, uy);
        hd.fast = padded_start_ok(*km, ux, uy) ? 1 : 0;
    }
    hd.n_dirs = 0;
    if (k.theta_inc < 1.0) {  // consecutive beams advance the table index by 0 or 1 (mod theta_dis)
        int span = beam_dir_index(k, start, k.num_beams - 1) - hd.i0;
        if (span < 0) span += k.theta_dis;
        hd.n_dirs = span + 1;
    }
    return hd;
}

// what the integration leaves in HBM for one agent (every column but the ray header)
__device__ __forceinline__ void integrate_store_columns(const AgentArrays &a, int i, int N, const double *st, double b0, double b1, int cnt, const double *sp,
                                                        double start)
{
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
    a.dir_start[i] = start;
}

__device__ __forceinline__ void integrate_store(const AgentArrays &a, const ScanConst &k, int i, int N, const double *st, double b0, double b1,
                                                int cnt, const double *sp)
{
    const double start = scan_start_index(k, sp[2]);
    integrate_store_columns(a, i, N, st, b0, b1, cnt, sp, start);
    const RayHdr hd = make_ray_hdr(a, k, i, st, sp, start);
    if (a.path_stats) atomicAdd(&a.path_stats[hd.fast ? 0 : 2], (unsigned long long)k.num_beams);
    a.ray_hdr[i] = hd;
    a.in_collision[i] = 0;  // raised by k_scan_rays when any beam's iTTC is under the threshold
    if (a.sched_count_zero && i == a.agent_begin) {
        a.sched_count_zero[0] = 0u;   // this step's task list (TaskSched::count_w)
    }
}

// ---- K1: integrate every agent one time step ------------------------------------------
// AF = 0: integration only (k_collide follows on a side stream, hidden under the scan).
// AF = 2 / 4 (agents per env): the pair tests run in the same lane right after the integration,
// the partners' post-integration poses (the :574 snapshot) arriving by lane shuffle — the agents
// of an env are neighbouring lanes of one wave.  Used when the step runs as several env groups on
// their own streams: one launch and no event fork/join per group, and the longer dependent chain
// hides under the other groups' scans.
template <int AF>
__global__ void __launch_bounds__(AF ? 64 : 256) k_integrate(AgentArrays a, ScanConst k, const double *__restrict__ actions)
{
    const int i = a.agent_begin + (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int N = a.n_agents_total;
    if (i >= a.agent_begin + a.agent_count) return;
    const VehicleParams vp = load_params(a.params + (size_t)(a.params_per_agent ? i : i % a.agents_per_env) * NPARAMS);
    double st[7];
#pragma unroll
    for (int c = 0; c < 7; ++c) st[c] = a.state[(size_t)c * N + i];
    double b0 = a.steer_buf[i], b1 = a.steer_buf[(size_t)N + i];
    int cnt = a.buf_cnt[i];
    const double2 act = reinterpret_cast<const double2 *>(actions)[i];
    double sp[3];
    advance_vehicle(st, b0, b1, cnt, act.x, act.y, vp, a.time_step, a.integrator, a.lidar_dist, sp);
    integrate_store(a, k, i, N, st, b0, b1, cnt, sp);
    if constexpr (AF != 0) {
        // groups are env-aligned and AF divides 64: the AF lanes of an env are all here, in one wave.
        // Every lane takes part in every shuffle (a lane masked off would read as zero).
        const int lane = (int)(threadIdx.x & 63u), me = i % AF;
        double px[AF], py[AF], pth[AF];
#pragma unroll
        for (int jj = 0; jj < AF; ++jj) {
            px[jj] = __shfl(st[0], lane - me + jj);
            py[jj] = __shfl(st[1], lane - me + jj);
            pth[jj] = __shfl(st[4], lane - me + jj);
        }
        collide_agent<AF>(a, k.num_beams, i, st[0], st[1], st[4], [&](int jj, double &ox, double &oy, double &oth) {
            ox = px[jj];
            oy = py[jj];
            oth = pth[jj];
        });
    }
}

// ---- K1 in two waves (round 3; what the product launches when the pair tests are not fused in) ---------------
// k_integrate is one dependent float64 chain per wave (one wave per SIMD at every batch size: its time is its
// instruction count, ~12 cycles each).  The low-speed branch of the right-hand side — tan and cos of the steering
// angle, a quarter of a stage's instructions in a wave that has lanes on both sides of |v| = 0.5, which is nearly
// every wave of a batch that re-seats crashed cars — depends on (steer, v) only, and THEIR derivatives depend on
// nothing else (low_speed_trig_ahead).  So a second wave of the workgroup walks (steer, v) through the stages on its
// own, one stage ahead of the integration, and leaves tan / cos in LDS; the first wave picks them up behind one
// workgroup barrier per stage.  Same operations on the same values: bit-identical to k_integrate<0>.
struct LowTrigShared {
    double (*tn)[64];
    double (*cd)[64];
    int lane;
    __device__ __forceinline__ void begin_stage(int) const { __syncthreads(); }
    __device__ __forceinline__ void operator()(int stage, double, double &t, double &c) const
    {
        t = tn[stage][lane];
        c = cd[stage][lane];
    }
};
struct LowTrigEmit {
    double (*tn)[64];
    double (*cd)[64];
    int lane;
    __device__ __forceinline__ void operator()(int stage, double t, double c) const
    {
        tn[stage][lane] = t;
        cd[stage][lane] = c;
    }
    __device__ __forceinline__ void end_stage(int) const { __syncthreads(); }
};

__global__ void __launch_bounds__(128) k_integrate_duo(AgentArrays a, ScanConst k, const double *__restrict__ actions)
{
    __shared__ double s_tn[4][64], s_cd[4][64];
    const int lane = (int)(threadIdx.x & 63u);
    const bool ahead = threadIdx.x >= 64u;   // wave 1: the low-speed branch's trigonometry; wave 0: everything else
    const int N = a.n_agents_total;
    const int i_raw = a.agent_begin + (int)(blockIdx.x * 64u) + lane;
    const bool live = i_raw < a.agent_begin + a.agent_count;
    const int i = live ? i_raw : a.agent_begin;   // (lanes past the end shadow the first agent: every lane meets every barrier)
    const VehicleParams vp = load_params(a.params + (size_t)(a.params_per_agent ? i : i % a.agents_per_env) * NPARAMS);
    const double2 act = reinterpret_cast<const double2 *>(actions)[i];
    if (ahead) {
        low_speed_trig_ahead(a.state[(size_t)2 * N + i], a.state[(size_t)3 * N + i], a.steer_buf[(size_t)N + i], a.buf_cnt[i], act.y, vp,
                             a.time_step, a.integrator, LowTrigEmit{s_tn, s_cd, lane});
        return;
    }
    double st[7];
#pragma unroll
    for (int c = 0; c < 7; ++c) st[c] = a.state[(size_t)c * N + i];
    double b0 = a.steer_buf[i], b1 = a.steer_buf[(size_t)N + i];
    int cnt = a.buf_cnt[i];
    double sp[3];
    advance_vehicle_with(st, b0, b1, cnt, act.x, act.y, vp, a.time_step, a.integrator, a.lidar_dist, sp, LowTrigShared{s_tn, s_cd, lane});
    if (!live) return;
    integrate_store(a, k, i, N, st, b0, b1, cnt, sp);
}

// ---- K1 fanned out over thirteen waves (round 4; RK4) -------------------------------------------------------------
// k_integrate_duo still runs one ~1900-instruction chain per wave; at 4096 agents that chain IS the kernel's 9.4 us
// (one wave per SIMD: ~10 cycles per dependent instruction).  The step's dataflow is much shallower (f110_math.hpp,
// "the RK4 step taken apart"): a workgroup of 13 waves serves 64 agents —
//   wave 0        the main chain: loads, delay buffer, fan_main (a handful of multiply-adds per stage), combine, store;
//   waves 1-4     stage s's low-speed branch (tan / cos of the steering angle -> f4, f5), lanes that take it;
//   waves 5-8     stage s's single-track coefficients (the three divisions by the velocity), lanes that take it;
//   waves 9-12    stage s's position derivatives (cos / sin of the heading) once the main chain has the headings —
// three workgroup barriers in all.  Same operations on the same operands as k_integrate<0>: bit-identical
// (tests/test_host_math.py test_fan_integration_*, test_integrate_fan_is_invisible).
constexpr int kFanWaves = 13;
constexpr int kFanMaxAgents = 24576;   // measured (profiles/r04_fan_sweep.txt): pays up to 16 384 agents (-1 us per step), loses from 32 768
__global__ void __launch_bounds__(64 * kFanWaves) k_integrate_fan(AgentArrays a, ScanConst k, const double *__restrict__ actions)
{
    __shared__ double s_l4[4][64], s_l5[4][64], s_k[4][6][64], s_ang[4][64], s_vel[4][64], s_f0[4][64], s_f1[4][64];
    const int lane = (int)(threadIdx.x & 63u);
    const int role = (int)(threadIdx.x >> 6);
    const int N = a.n_agents_total;
    const int i_raw = a.agent_begin + (int)(blockIdx.x * 64u) + lane;
    const bool live = i_raw < a.agent_begin + a.agent_count;
    const int i = live ? i_raw : a.agent_begin;   // (lanes past the end shadow the first agent: every lane meets every barrier)
    const VehicleParams vp = load_params(a.params + (size_t)(a.params_per_agent ? i : i % a.agents_per_env) * NPARAMS);
    const double2 act = reinterpret_cast<const double2 *>(actions)[i];
    const double steer0 = a.state[(size_t)2 * N + i], vel0 = a.state[(size_t)3 * N + i];
    const double buf1_in = a.steer_buf[(size_t)N + i];
    const int cnt_in = a.buf_cnt[i];
    double accl, sv;
    fan_inputs(steer0, vel0, buf1_in, cnt_in, act.y, vp, accl, sv);
    double st[7];
    double b0 = 0., b1 = 0.;
    int cnt = 0;
    if (role == 0) {
#pragma unroll
        for (int c = 0; c < 7; ++c) st[c] = a.state[(size_t)c * N + i];
        b0 = a.steer_buf[i];
        b1 = buf1_in;
        cnt = cnt_in;
        if (cnt < 2) cnt += 1;   // :271-278 two-step steering delay
        b1 = b0;
        b0 = act.x;
    } else if (role <= 4) {
        const int s = role - 1;
        const FanWalk w = fan_walk(steer0, vel0, accl, sv, vp, a.time_step, s);
        if (w.low) {
            double f4, f5;
            fan_low(w, vp, f4, f5);
            s_l4[s][lane] = f4;
            s_l5[s][lane] = f5;
        }
    } else if (role <= 8) {
        const int s = role - 5;
        const FanWalk w = fan_walk(steer0, vel0, accl, sv, vp, a.time_step, s);
        if (!w.low) {
            double kk[6];
            fan_dyn(w, vp, kk);
#pragma unroll
            for (int c = 0; c < 6; ++c) s_k[s][c][lane] = kk[c];
        }
    }
    __syncthreads();
    const double x0 = st[0], y0 = st[1];
    if (role == 0) {
        fan_main(st, accl, sv, vp, a.time_step,
                 [&](int s, double &f4, double &f5) {
                     f4 = s_l4[s][lane];
                     f5 = s_l5[s][lane];
                 },
                 [&](int s, double *kk) {
#pragma unroll
                     for (int c = 0; c < 6; ++c) kk[c] = s_k[s][c][lane];
                 },
                 [&](int s, double ang, double v) {
                     s_ang[s][lane] = ang;
                     s_vel[s][lane] = v;
                 });
    }
    __syncthreads();
    if (role >= 9) {
        const int s = role - 9;
        double f0, f1;
        fan_pos(s_ang[s][lane], s_vel[s][lane], f0, f1);
        s_f0[s][lane] = f0;
        s_f1[s][lane] = f1;
    }
    __syncthreads();
    if (role != 0) return;
    st[0] = fan_combine(x0, a.time_step, s_f0[0][lane], s_f0[1][lane], s_f0[2][lane], s_f0[3][lane]);
    st[1] = fan_combine(y0, a.time_step, s_f1[0][lane], s_f1[1][lane], s_f1[2][lane], s_f1[3][lane]);
    double sp[3];
    fan_finish(st, a.lidar_dist, sp);
    if (!live) return;
    integrate_store(a, k, i, N, st, b0, b1, cnt, sp);
}

// ---- K1b: pairwise body collisions inside each env (separate launch, side stream) ------------
__global__ void __launch_bounds__(256) k_collide(AgentArrays a, int32_t B)
{
    const int i = a.agent_begin + (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int N = a.n_agents_total, A = a.agents_per_env;
    if (i >= a.agent_begin + a.agent_count) return;
    const int env = i / A;
    collide_agent<0>(a, B, i, a.snap_pose[i], a.snap_pose[(size_t)N + i], a.snap_pose[2 * (size_t)N + i],
                     [&](int jj, double &ox, double &oy, double &oth) {
                         const int o = env * A + jj;
                         ox = a.snap_pose[o];
                         oy = a.snap_pose[(size_t)N + o];
                         oth = a.snap_pose[2 * (size_t)N + o];
                     });
}

// ---- K2: ray march -------------------------------------------------------------------------
// Rays are numbered ray = pose*B + beam, one lane per ray, so the 64 lanes of a wave are
// consecutive beams of (almost always) one pose: neighbouring beams touch neighbouring cells and
// have correlated lengths.  STEP=true is the env.step() form (SoA poses written by
// k_integrate, noise row, iTTC predicate); STEP=false is ScanSimulator2D.scan for the unit
// entry point (also reports terminating cells and lookup counts).
// Longest-first dispatch of the step's scan tasks, for SMALL batches.  With few agents the scan kernel
// ends when its longest ray ends, and a 64-ray task whose longest ray takes 300 samples may
// only be STARTED in the last of the launch's eight-odd rounds of waves.  Ray lengths barely change from
// one step to the next and workgroups start in block order: a task whose longest ray exceeded `thr`
// lookups in the PREVIOUS step is put on a list, and the list is served by the first blocks of this step's
// launch.  flags_r[task] == epoch_r marks "on the list" (the normal blocks skip those tasks); this step's
// long tasks go to list_w / flags_w for the next step (stamped epoch_w; stale stamps never match again, so
// nothing is ever cleared).  Stale or missing entries only cost speed: every task is marched exactly once.

struct TaskSched {
    const uint32_t *flags_r;
    uint32_t *flags_w;
    const uint32_t *list_r;
    uint32_t *list_w;
    const uint32_t *count_r;
    uint32_t *count_w;   // zeroed by k_integrate of the same step
    uint32_t cap, thr;
    // Reserved words here and in RayJob keep the kernel-argument layout the scan kernels were tuned on: the retired ray pass /
    // two-pass / LDS-window fields used to sit there, and taking them out moved the hot fields across the boundaries of the
    // compiler's wide scalar loads — 4096 agents 46.3 -> 44.6 M agent-steps/s, back to 46.7 with the padding (round 5, A/B of
    // three builds in one session).  Asking for all arguments in ONE batch of scalar loads at the top of the kernel (nine
    // dependent round trips -> two) is no cure either: 46.7 -> 44.7 M at 4096 agents, no change at 2048 / 8192 / 16 384.
    const void *reserved_r[8];
    uint32_t reserved_rcap, reserved_rthr;
};

struct RayJob {
    uint32_t n_rays;          // poses * B
    uint32_t n_tasks;         // ceil(n_rays / 64): one task = 64 consecutive rays
    uint32_t tasks_per_wave;  // consecutive tasks each wave walks
    int32_t n_poses;
    uint32_t div_magic, div_shift;  // ray / B == umulhi(ray, magic) >> shift (0: plain division)
    int32_t pad_dir, dir_stride;    // k_scan_dirs_agent: distinct directions per agent, rounded up to whole 64-direction tasks
    const void *reserved_dir_ranges;
    uint32_t first_pose, reserved_spec; // k_scan_rays_agent: the launch covers agents first_pose .. (env group)
    unsigned long long *lookups_total;  // COUNT variants: table lookups of every marched ray, summed (or nullptr)
    // longest-first task order (k_scan_rays_agent<.., SCHED>): see TaskSched
    TaskSched sched;   // by value: the kernel argument segment is the one read a wave never waits long for
    uint32_t epoch_r, epoch_w, long_blocks, pad_blocks;
    // fusion-feasibility probe (experimental build): per-env count of finished scan tasks, reset by the last arriver
    uint32_t *env_done;
    uint32_t tasks_per_env, long_prio;
    uint32_t long_rev, pad_rev;   // 1: the long pass walks last step's list from its newest entry (the tasks that finished last)
    // timeline probe (experimental build, k_scan_rays_agent): [launch waves][8] = {begin, end (100 MHz clock), HW_ID | XCC_ID << 32,
    // lock-step samples marched | long pass << 32, task loop entered, first task's stamp + header arrived, its direction /
    // noise operands arrived (march begins), last march done} written by every wave that reaches the end of the kernel; nullptr = off
    unsigned long long *trace;
    // per-env maps: the order the scan walks the agents in — sorted by map slot, so that the XCD-contiguous
    // block order hands each XCD's L2 the agents of as few tracks as possible however the caller interleaved
    // them (nullptr: agent order)
    const uint32_t *order;
    const void *reserved_win[2];
    uint32_t reserved_win_pitch, pad_win;
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
    // diagnostics (unit form; the step form counts per scan in k_integrate), nullptr unless enabled:
    // rays marched {fixed-point only, given up and re-marched exactly, exact throughout}
    unsigned long long *path_stats;
    // HBM copy of the ScanConst the kernel also receives by value: the (very rare) exact re-march of
    // the PADDED layout reads its constants from here, so they cost the fast kernel no registers
    const ScanConst *k_cold;
};

__device__ __forceinline__ int uniform_i32(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double uniform_f64(double v)
{
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// Per-agent ray constants for the lane's agent p.  A wave holds 64 consecutive rays, so with >= 64
// rays per agent all its lanes belong to one agent except in the one wave per agent that
// straddles a boundary (1 in 17 at 1080 beams).  Uniform wave: the 64-byte header comes through
// the scalar cache (no vector-memory instruction, no per-lane select); uniform_*() pins the
// fields to SGPRs so the compiler does not turn them back into a divergent vector load.
// Straddling wave: plain per-lane loads.
struct LaneHdr {
    double x, y, start, vel, d0;
    int row, map_slot, i0, n_dirs, fast;
};

__device__ __forceinline__ LaneHdr load_lane_hdr(const RayHdr *hdr, uint32_t p)
{
    LaneHdr o;
    const uint32_t p0 = __builtin_amdgcn_readfirstlane(p);
    if (__ballot(p != p0) == 0ull) {  // wave-uniform branch
        typedef const __attribute__((address_space(4))) RayHdr *chdr_t;
        const chdr_t h0 = (chdr_t)(hdr) + p0;
        o.x = uniform_f64(h0->x);
        o.y = uniform_f64(h0->y);
        o.start = uniform_f64(h0->start);
        o.vel = uniform_f64(h0->vel);
        o.d0 = uniform_f64(h0->d0);
        o.row = uniform_i32(h0->noise_row);
        o.map_slot = uniform_i32(h0->map_slot);
        o.i0 = uniform_i32(h0->i0);
        o.n_dirs = uniform_i32(h0->n_dirs);
        o.fast = uniform_i32(h0->fast);
    } else {
        const RayHdr hd = hdr[p];
        o.x = hd.x; o.y = hd.y; o.start = hd.start; o.vel = hd.vel; o.d0 = hd.d0; o.row = hd.noise_row;
        o.map_slot = hd.map_slot; o.i0 = hd.i0; o.n_dirs = hd.n_dirs;
        o.fast = hd.fast;
    }
    return o;
}

// noise + iTTC + store for one beam (k_scan_rays, k_scan_dirs_agent).
// check_ttc_jit is an any-over-beams: every hitting lane raises the agent's flag.
// r > max(side) + thresh*(1+1e-9)*max|cos|*|v| implies r - side_distances[b] >
// thresh*(1+1e-12)*|v*cosines[b]|, i.e. the "no hit" branch of ttc_beam_hit, so the per-beam
// tables are only read for the few beams that are that close.
struct RayJob;
__device__ __forceinline__ void finish_beam(const RayJob &j, uint32_t B, uint32_t p, int b, uint32_t ray, double r,
                                            int row, double vel);
__device__ __forceinline__ void finish_beam_with(const RayJob &j, uint32_t p, int b, uint32_t ray, double r, double vel);

// all 64 lanes (reconverged after the task loop): sum the per-lane lookup counts, one atomic per wave
__device__ __forceinline__ void wave_add_lookups(unsigned long long *total, uint32_t mine)
{
    for (int off = 32; off; off >>= 1) mine += __shfl_xor(mine, off);
    if ((threadIdx.x & 63u) == 0u) atomicAdd(total, (unsigned long long)mine);
}

// One ray from its (shared) first sample d0 on.  path: 0 fixed-point march, 1 fixed-point march
// given up and re-marched exactly, 2 exact arithmetic throughout.
template <int LAYOUT, bool POW2, bool IDENT, bool WANT_CELL>
__device__ __forceinline__ double trace_from_first(const ScanConst &k, const ScanConst *k_cold, const double *lut, double x, double y,
                                                   bool fast, double c, double s, double d0, int &hr, int &hc, int &nl,
                                                   int &path)
{
    if (LAYOUT == LAYOUT_PADDED) {
        double r = 0.;
        bool exact = !fast;
        if (fast) {
            double ux, uy, cux, cuy;
            padded_position<IDENT>(k, x, y, ux, uy);
            padded_rate<IDENT>(k, c, s, cux, cuy);
            exact = !march_padded<WANT_CELL>(k, ux, uy, cux, cuy, d0, r, hr, hc, nl);
        }
        path = fast ? (exact ? 1 : 0) : 2;
        if (exact) r = march_exact_cold<IDENT>(k_cold, x, y, c, s, d0, hr, hc, nl);
        return r;
    }
    path = 2;
    return march_from_first<LAYOUT, POW2, IDENT>(k, lut, x, y, c, s, d0, hr, hc, nl);
}

template <int LAYOUT, bool POW2, bool IDENT, bool STEP>
__global__ void __launch_bounds__(256) k_scan_rays(RayJob j, ScanConst k)
{
    const double *lut_lds = nullptr;   // (the byte-code layout's value table: retired in round 5)
    const uint32_t B = (uint32_t)k.num_beams;
    const uint32_t tpw = j.tasks_per_wave;
    const uint32_t lane = threadIdx.x & 63u;
    // Workgroup b is dispatched to XCD b % 8 (observed, MI355X_MICROARCH.md).  Re-map so that each
    // XCD walks one contiguous eighth of the rays: all beams of an agent, and agents that are
    // neighbours in the batch, then share one XCD's L2 instead of being spread over all eight.
    uint32_t blk = blockIdx.x;
    {
        const uint32_t nb = gridDim.x, q = nb >> 3, rem = nb & 7u, x = blk & 7u, i = blk >> 3;
        blk = (x < rem ? x * (q + 1u) : rem * (q + 1u) + (x - rem) * q) + i;  // bijective for any nb
    }
    const uint32_t wave = (blk * blockDim.x + threadIdx.x) >> 6;
    uint32_t nl_acc = 0;   // STEP: table lookups of this lane's rays (summed per wave when counting is on)
    for (uint32_t t = 0; t < tpw; ++t) {
        const uint32_t task = wave * tpw + t;
        if (task >= j.n_tasks) break;  // wave-uniform
        const uint32_t ray = task * 64u + lane;
        if (ray >= j.n_rays) break;
        const uint32_t p = j.div_magic ? (__umulhi(ray, j.div_magic) >> j.div_shift) : ray / B;
        const int b = (int)(ray - p * B);
        int hr, hc, nl, path;
        double r;
        if (STEP) {
            const LaneHdr hd = load_lane_hdr(j.hdr, p);
            hr = -1;
            hc = -1;
            const double2 cs = k.cs[beam_dir_index(k, hd.start, b)];
            r = trace_from_first<LAYOUT, POW2, IDENT, false>(k, j.k_cold, lut_lds, hd.x, hd.y, hd.fast != 0, cs.x, cs.y, hd.d0, hr, hc, nl, path);
            nl_acc += (uint32_t)nl;
            finish_beam(j, B, p, b, ray, r, hd.row, hd.vel);
            continue;
        } else {
            const double2 cs = k.cs[beam_dir_index(k, j.dir_start[p], b)];
            const double x = j.pose_x[p], y = j.pose_y[p];
            const double d0 = sample_distance<LAYOUT, POW2, IDENT>(k, lut_lds, x, y, hr, hc);
            bool fast = false;
            if (LAYOUT == LAYOUT_PADDED) {
                double ux, uy;
                padded_position<IDENT>(k, x, y, ux, uy);
                fast = padded_start_ok(k, ux, uy);
            }
            r = trace_from_first<LAYOUT, POW2, IDENT, true>(k, j.k_cold, lut_lds, x, y, fast, cs.x, cs.y, d0, hr, hc, nl, path);
            if (j.path_stats) atomicAdd(&j.path_stats[path], 1ull);
            if (j.hit_rc) {
                j.hit_rc[(size_t)ray * 2] = hr;
                j.hit_rc[(size_t)ray * 2 + 1] = hc;
            }
            if (j.lookups) atomicAdd(&j.lookups[p], (unsigned long long)nl);
        }
        j.ranges[ray] = r;
    }
    if (STEP && j.lookups_total) wave_add_lookups(j.lookups_total, nl_acc);
}

// ---- K2a: the step's ray march, agent-aligned ----------------------------------------------------
// Same march as k_scan_rays<LAYOUT_PADDED, ..., STEP>, with every agent's beams laid out on whole
// 64-ray tasks (ceil(B/64) tasks per agent, the last one partly idle: 0.7 % of the lanes at 1080
// beams).  A wave then always belongs to ONE agent: the header is scalar in every wave (no per-lane
// header path, no per-lane ray -> (agent, beam) division, lidar position and first sample stay in
// SGPRs) — and the agent's map can be a per-env property (f110_set_env_maps): its constants, one
// 64-byte MapFast record, arrive through the scalar cache next to the header and the loop runs
// with scalar constants exactly as with one map.
struct MapFast {
    const double *pad;                       // the map's padded table
    double pad_cx, pad_cy;                   // padded_position constants (general, rotated form)
    double pad_axx, pad_axy, pad_ayx, pad_ayy;
    uint32_t pad_row_bytes;
    int32_t pad_max_samples;
};
static_assert(sizeof(MapFast) == 64, "MapFast is read as one 64-byte scalar load");

// the per-env map's constants through the scalar cache (one 64-byte record), pinned to SGPRs
__device__ __forceinline__ void load_map_fast(ScanConst &km, const MapFast *__restrict__ maps_fast, int slot)
{
    typedef const __attribute__((address_space(4))) MapFast *cmap_t;
    const cmap_t m0 = (cmap_t)(maps_fast) + slot;
    const uint64_t pa = (uint64_t)m0->pad;
    km.pad = (const double *)(((uint64_t)(uint32_t)uniform_i32((int)(pa >> 32)) << 32) | (uint32_t)uniform_i32((int)pa));
    km.pad_cx = uniform_f64(m0->pad_cx);
    km.pad_cy = uniform_f64(m0->pad_cy);
    km.pad_axx = uniform_f64(m0->pad_axx);
    km.pad_axy = uniform_f64(m0->pad_axy);
    km.pad_ayx = uniform_f64(m0->pad_ayx);
    km.pad_ayy = uniform_f64(m0->pad_ayy);
    km.pad_row_bytes = uniform_i32((int)m0->pad_row_bytes);
    km.pad_max_samples = uniform_i32(m0->pad_max_samples);
}

#ifndef F110_SCAN_WAVES_EXPR
#define F110_SCAN_WAVES_EXPR 8
#endif
template <bool PER_ENV_MAP, bool IDENT, bool COUNT, bool SCHED = false, bool ENVCNT = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(F110_SCAN_WAVES_EXPR))) k_scan_rays_agent(RayJob j, ScanConst k, const MapFast *__restrict__ maps_fast,
                                                          const ScanConst *__restrict__ maps_full, uint32_t tasks_per_agent)
{
    const uint32_t B = (uint32_t)k.num_beams;
    uint32_t tpw = j.tasks_per_wave;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t blk = blockIdx.x;
    typedef const __attribute__((address_space(4))) uint32_t *cu32_t;
    bool long_pass = false;
    uint32_t long_task = 0;
#ifdef F110_EXPERIMENTAL
    const unsigned long long trace_begin = j.trace ? wall_clock64() : 0ull;
    unsigned long long trace_loop = 0ull, trace_hdr = 0ull, trace_ops = 0ull, trace_marched = 0ull;
    uint32_t trace_samples = 0;
#endif
    if (SCHED) {
        if (blk < j.long_blocks) {   // the first blocks of the launch serve last step's long tasks, one per wave
            const TaskSched &sc = j.sched;
            const uint32_t wl = __builtin_amdgcn_readfirstlane((blk * blockDim.x + threadIdx.x) >> 6);
            uint32_t cnt = *(cu32_t)sc.count_r;
            cnt = cnt < sc.cap ? cnt : sc.cap;
            if (wl >= cnt) return;
            long_pass = true;
            long_task = ((cu32_t)sc.list_r)[j.long_rev ? cnt - 1u - wl : wl];
            tpw = 1u;
            if (j.long_prio) __builtin_amdgcn_s_setprio(3);   // (experiment: instruction-issue priority for the waves the launch waits for)
        } else {
            blk -= j.long_blocks;
        }
    }
    if (!long_pass) {
        const uint32_t nb = gridDim.x - (SCHED ? j.long_blocks : 0u), q = nb >> 3, rem = nb & 7u, x = blk & 7u, i = blk >> 3;
        blk = (x < rem ? x * (q + 1u) : rem * (q + 1u) + (x - rem) * q) + i;  // XCD-contiguous, as k_scan_rays
    }
    const uint32_t wave = (blk * blockDim.x + threadIdx.x) >> 6;
    uint32_t nl_acc = 0;
    // PER_ENV_MAP: the map record is (re)loaded only when the slot changes — the wave's consecutive tasks belong to
    // one or two agents, and the scan order groups the agents by slot (f110_set_env_maps)
    ScanConst km = k;
    const ScanConst *cold = j.k_cold;
    int cur_slot = -1;
#ifdef F110_EXPERIMENTAL
    if (j.trace) trace_loop = wall_clock64();
#endif
    for (uint32_t t = 0; t < tpw; ++t) {
        const uint32_t task = __builtin_amdgcn_readfirstlane(long_pass ? long_task : wave * tpw + t);
        if (task >= j.n_tasks) break;
        const uint32_t pl = task / tasks_per_agent;                 // scalar
        const uint32_t p = (PER_ENV_MAP && j.order) ? ((cu32_t)j.order)[pl] : j.first_pose + pl;
        // the task's two cold words — its stamp on last step's long list and its agent's header, both written by
        // other kernels on other XCDs — are requested together: one memory round trip, not two in a row (a small
        // batch is bound by its per-task latency chain: tools/debug/scan_timeline.py).  Measured and NOT done: keeping
        // the header in SGPRs while consecutive tasks belong to one agent (65 536 agents 92.1 -> 86.9 M agent-steps/s:
        // the loop-carried scalars cost more than the scalar-cache hit they save), and requesting the next task's
        // direction / noise operands before this task's march (92.1 -> 91.7).
        const uint32_t stamp = (SCHED && !long_pass) ? ((cu32_t)j.sched.flags_r)[task] : 0u;
        typedef const __attribute__((address_space(4))) RayHdr *chdr_t;
        const chdr_t h0 = (chdr_t)(j.hdr) + p;
        const double x = uniform_f64(h0->x), y = uniform_f64(h0->y), start = uniform_f64(h0->start);
        const double vel = uniform_f64(h0->vel), d0 = uniform_f64(h0->d0);
        const int row = uniform_i32(h0->noise_row), slot = uniform_i32(h0->map_slot), fast = uniform_i32(h0->fast);
#ifdef F110_EXPERIMENTAL
        if (j.trace && trace_hdr == 0ull) trace_hdr = wall_clock64();   // (reading the clock waits for the scalar loads above)
#endif
        if (SCHED && !long_pass && stamp == j.epoch_r) continue;   // served by the long pass
        const int b = (int)((task - pl * tasks_per_agent) * 64u + lane);
        if (b >= (int)B) continue;
        if (PER_ENV_MAP && slot != cur_slot) {   // wave-uniform
            cold = maps_full + slot;
            load_map_fast(km, maps_fast, slot);
            cur_slot = slot;
        }
        // the beam's noise sample is requested before the march so that its latency hides under it:
        // row of the table / row cache, or (row -2) the row k_noise_rows left in this agent's scans[]
        const double *nrow = row >= 0 ? j.noise + (size_t)row * B : j.ranges + (size_t)p * B;   // scalar
        const double nz = row != -1 ? nrow[b] : 0.0;
        double2 cs = k.cs[beam_dir_index(k, start, b)];
#ifdef F110_EXPERIMENTAL
        if (j.trace && trace_ops == 0ull) {
            asm volatile("" : "+v"(cs.x), "+v"(cs.y));   // (the direction has arrived)
            trace_ops = wall_clock64();
        }
#endif
        int hr = -1, hc = -1, nl = 0;
        double r = 0.;
        bool exact = fast == 0;
        if (fast) {
            double ux, uy, cux, cuy;
            padded_position<IDENT>(km, x, y, ux, uy);
            padded_rate<IDENT>(km, cs.x, cs.y, cux, cuy);
            exact = !march_padded<false>(km, ux, uy, cux, cuy, d0, r, hr, hc, nl);
        }
        if (exact) r = march_exact_cold<IDENT>(cold, x, y, cs.x, cs.y, d0, hr, hc, nl);
        if (COUNT) nl_acc += (uint32_t)nl;   // measurement variant only (bench.py's L-bar)
#ifdef F110_EXPERIMENTAL
        if (j.trace) {
            asm volatile("" : "+v"(r));
            trace_marched = wall_clock64();
        }
        if (j.trace) {   // the task's lock-step length = its longest ray
            int m = 0;   // max over the lanes that hold a beam, bit by bit (ten compares; a cheap probe distorts least)
            for (int bit = 1 << 10; bit; bit >>= 1)
                if (__ballot(nl >= (m | bit)) != 0ull) m |= bit;
            trace_samples += (uint32_t)m;
        }
#endif
        if (SCHED) {
            const TaskSched &sc = j.sched;
            if (__ballot(nl > (int)sc.thr) != 0ull && lane == 0u) {
                // lane 0 (beam task*64, always a valid beam) appends the task for the next step
                const uint32_t pos = atomicAdd(sc.count_w, 1u);
                if (pos < sc.cap) {
                    sc.list_w[pos] = task;
                    sc.flags_w[task] = j.epoch_w;
                }
            }
        }
        finish_beam_with(j, p, b, p * B + (uint32_t)b, row != -1 ? r + nz : r, vel);
        if (ENVCNT && lane == 0u) {
            // probe: what a per-env completion counter costs (one returning agent-scope atomic per task; the
            // arrival that completes the env resets the counter, as a fused finalize would before it runs)
            const uint32_t env = p / (j.tasks_per_env / tasks_per_agent);
            const uint32_t old = __hip_atomic_fetch_add(j.env_done + env, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old + 1u == j.tasks_per_env) __hip_atomic_store(j.env_done + env, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (COUNT) wave_add_lookups(j.lookups_total, nl_acc);
#ifdef F110_EXPERIMENTAL
    if (j.trace && lane == 0u) {
        uint32_t hw_id, xcc_id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
        unsigned long long *rec = j.trace + 8ull * ((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
        rec[0] = trace_begin;
        rec[1] = wall_clock64();
        rec[2] = (unsigned long long)hw_id | ((unsigned long long)xcc_id << 32);
        rec[3] = (unsigned long long)trace_samples | ((unsigned long long)(long_pass ? 1u : 0u) << 32);
        rec[4] = trace_loop;
        rec[5] = trace_hdr;
        rec[6] = trace_ops;
        rec[7] = trace_marched;
    }
#endif
}

// (Round 5 built k_scan_stream_agent here — "survivor compaction": a wave owns an agent's whole scan as a queue and re-fills finished
// lanes; 0.65-0.79 of the wave-level gathers, bit-identical — and measured it 7-20 % slower at every setting: the re-filled wave's rays
// share fewer lines, and the scan is bound by line traffic, not by instruction count.  Retired in round 6 under the pre-registered
// stop rule: DESIGN.md §8, profiles/r05_stream_scan.txt, r05_pmc_stream_vs_base.json.)

// iTTC + store for one beam whose noise sample has been added already
__device__ __forceinline__ void finish_beam_with(const RayJob &j, uint32_t p, int b, uint32_t ray, double r, double vel)
{
    if (vel != 0.0 && !(r > j.ttc_side_max + j.ttc_k * fabs(vel)) &&
        ttc_beam_hit(r, j.side_dist[b], vel, j.beam_cos[b], j.ttc_thresh))
        j.wall_flag[p] = 1;
    // (round 6 measured non-temporal range stores here for small batches — does the table stay in L2 when 35 MB of range writes per
    // step bypass it? — no change at any size, +-0.2 %: profiles/r06_nt_store.txt)
    j.ranges[ray] = r;
}

__device__ __forceinline__ void finish_beam(const RayJob &j, uint32_t B, uint32_t p, int b, uint32_t ray, double r,
                                            int row, double vel)
{
    if (row >= 0)
        r += j.noise[(size_t)row * B + b];
    else if (row == -2)
        r += j.ranges[ray];   // this step's noise row, left in the agent's scans[] by k_noise_rows
    finish_beam_with(j, p, b, ray, r, vel);
}

// ---- K2d: more beams than table directions, one pass -----------------------------------------------
// BASELINE config 5 (4096 beams, theta_dis = 2000 -> 1497 distinct directions per scan).  Agent-aligned
// like k_scan_rays_agent, but a task is 64 consecutive DISTINCT directions of one agent: each lane
// marches one direction once, then the wave writes the ~175 consecutive beams that use those 64
// directions in coalesced passes of 64 — beam b takes its range from lane rel(b) - s0 by shuffle and
// gets its own noise sample and iTTC test.  Replaces the (march to a [N][dir_stride] buffer,
// k_expand_beams) pair: one launch less and no 1.6 GB round trip of intermediate ranges per step.
// Bit-identical to marching every beam (beams that share a table index from one origin are one ray).
#ifndef F110_DIRS_WAVES_EXPR
#define F110_DIRS_WAVES_EXPR 8
#endif
template <bool PER_ENV_MAP, bool IDENT, bool COUNT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(F110_DIRS_WAVES_EXPR))) k_scan_dirs_agent(RayJob j, ScanConst k, const MapFast *__restrict__ maps_fast,
                                                          const ScanConst *__restrict__ maps_full, uint32_t tasks_per_agent)
{
    const uint32_t B = (uint32_t)k.num_beams;
    const uint32_t tpw = j.tasks_per_wave;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t blk = blockIdx.x;
    {
        const uint32_t nb = gridDim.x, q = nb >> 3, rem = nb & 7u, x = blk & 7u, i = blk >> 3;
        blk = (x < rem ? x * (q + 1u) : rem * (q + 1u) + (x - rem) * q) + i;  // XCD-contiguous, as k_scan_rays
    }
    const uint32_t wave = (blk * blockDim.x + threadIdx.x) >> 6;
    uint32_t nl_acc = 0;
    ScanConst km = k;   // PER_ENV_MAP: this agent's track (f110_set_env_maps), reloaded when the slot changes
    const ScanConst *cold = j.k_cold;
    int cur_slot = -1;
    for (uint32_t t = 0; t < tpw; ++t) {
        const uint32_t task = __builtin_amdgcn_readfirstlane(wave * tpw + t);
        if (task >= j.n_tasks) break;
        const uint32_t pl = task / tasks_per_agent;
        typedef const __attribute__((address_space(4))) uint32_t *cu32_t;
        const uint32_t p = (PER_ENV_MAP && j.order) ? ((cu32_t)j.order)[pl] : j.first_pose + pl;
        const int s0 = (int)((task - pl * tasks_per_agent) * 64u);
        typedef const __attribute__((address_space(4))) RayHdr *chdr_t;
        const chdr_t h0 = (chdr_t)(j.hdr) + p;
        const int n_dirs = uniform_i32(h0->n_dirs);
        if (s0 >= n_dirs) continue;   // wave-uniform: the padding tasks of dir_stride
        const double x = uniform_f64(h0->x), y = uniform_f64(h0->y), start = uniform_f64(h0->start);
        const double vel = uniform_f64(h0->vel), d0 = uniform_f64(h0->d0);
        const int row = uniform_i32(h0->noise_row), fast = uniform_i32(h0->fast), i0 = uniform_i32(h0->i0);
        if (PER_ENV_MAP) {
            const int slot = uniform_i32(h0->map_slot);
            if (slot != cur_slot) {   // wave-uniform
                cold = maps_full + slot;
                load_map_fast(km, maps_fast, slot);
                cur_slot = slot;
            }
        }
        // first beam whose table index is direction s0 (relative indices never decrease with the beam):
        // closed-form estimate, then the exact index function decides
        auto rel_of = [&](int b) {
            int rr = beam_dir_index(k, start, b) - i0;
            return rr < 0 ? rr + k.theta_dis : rr;
        };
        int b0 = 0;
        if (s0 > 0) {
            b0 = (int)ceil(((double)s0 - (start - floor(start))) / k.theta_inc) - 2;
            b0 = b0 < 0 ? 0 : (b0 >= (int)B ? (int)B - 1 : b0);
            b0 = __builtin_amdgcn_readfirstlane(b0);
            while (b0 > 0 && rel_of(b0 - 1) >= s0) --b0;
            while (b0 < (int)B && rel_of(b0) < s0) ++b0;
        }
        // Which beams this task writes, which lane's direction each of them takes and its noise sample are all known BEFORE the
        // march: the first kPre passes of 64 beams (~175 beams per task at 4096 beams on 2000 directions: three passes) ask for
        // their noise here, so that the loads fly under the march instead of forming a dependent load -> store round trip per
        // pass behind it (round 5: the kernel is a latency chain per task — 9 dependent gathers, then three such round trips)
        constexpr int kPre = 3;
        const double *nrow = row >= 0 ? j.noise + (size_t)row * B : j.ranges + (size_t)p * B;   // (row -2: this step's noise row, left in the agent's scans[] by k_noise_rows)
        bool pre_in[kPre];
        int pre_src[kPre];
        double pre_nz[kPre];
#pragma unroll
        for (int q = 0; q < kPre; ++q) {
            const int b = b0 + (int)lane + 64 * q;
            bool in = b < (int)B;
            int src = 0;
            if (in) {
                src = rel_of(b) - s0;
                in = src < 64;
            }
            pre_in[q] = in;
            pre_src[q] = src;
            pre_nz[q] = (in && row != -1) ? nrow[b] : 0.0;
        }
        double r_dir = 0.;
        if (s0 + (int)lane < n_dirs) {
            int didx = i0 + s0 + (int)lane;
            if (didx >= k.theta_dis) didx -= k.theta_dis;
            const double2 cs = k.cs[didx];
            int hr = -1, hc = -1, nl;
            bool exact = fast == 0;
            if (fast) {
                double ux, uy, cux, cuy;
                padded_position<IDENT>(km, x, y, ux, uy);
                padded_rate<IDENT>(km, cs.x, cs.y, cux, cuy);
                exact = !march_padded<false>(km, ux, uy, cux, cuy, d0, r_dir, hr, hc, nl);
            }
            if (exact) r_dir = march_exact_cold<IDENT>(cold, x, y, cs.x, cs.y, d0, hr, hc, nl);
            if (COUNT) nl_acc += (uint32_t)nl;
        }
        // the write passes (every lane takes part in every shuffle); the indices are monotone: the first pass with an idle lane is the last
        bool more = true;
#pragma unroll
        for (int q = 0; q < kPre; ++q) {
            if (more) {
                const int b = b0 + (int)lane + 64 * q;
                const double r = __shfl(r_dir, pre_in[q] ? pre_src[q] : 0);
                if (pre_in[q]) finish_beam_with(j, p, b, p * B + (uint32_t)b, row != -1 ? r + pre_nz[q] : r, vel);
                if (__ballot(!pre_in[q]) != 0ull) more = false;
            }
        }
        for (int b = b0 + (int)lane + 64 * kPre; more; b += 64) {   // (more beams per direction than kPre passes cover: load, then store)
            bool in = b < (int)B;
            int src = 0;
            if (in) {
                src = rel_of(b) - s0;
                in = src < 64;
            }
            const double r = __shfl(r_dir, in ? src : 0);
            if (in) finish_beam(j, B, p, b, p * B + (uint32_t)b, r, row, vel);
            if (__ballot(!in) != 0ull) more = false;
        }
    }
    if (COUNT) wave_add_lookups(j.lookups_total, nl_acc);
}


// ---- scan noise generated on the device --------------------------------------------
// rng.normal(0., std, num_beams) of laser_models.py:450-452 — NumPy's PCG64 + ziggurat, restated in
// f110_rng.hpp.  One wave produces one row of B samples: lane l evaluates the attempt that would
// start at draw l of the current 64-draw chunk (its generator state comes from the jump constants:
// A^(l+1) * s + G_(l+1) * inc), the few attempts that consume more than one draw decide which
// positions really start an attempt (zig_chain_starts), and the accepted values are packed in order.
struct NoiseGen {
    const uint64_t *zk;   // ziggurat tables (device copies)
    const double *zw, *zf;
    const U128 *ja, *jg;  // PcgJump (device copy): [65] each
    double scale;         // std_dev of the noise (loc = 0.)
};

// `state` (wave-uniform): stream position at the start of the row in, at its end out
__device__ __forceinline__ void noise_row_wave(const NoiseGen &g, U128 &state, const U128 inc, double *__restrict__ out, int B)
{
    const int lane = (int)(threadIdx.x & 63u);
    const ZigTables zt = {g.zk, g.zw, g.zf};
    const U128 aj = g.ja[lane + 1];
    const U128 gi = mul128(g.jg[lane + 1], inc);
    const uint64_t below = (1ull << lane) - 1ull;
    int produced = 0, skip = 0;
    U128 s = state;
    for (;;) {
        const U128 st = add128(mul128(aj, s), gi);
        const ZigAttempt z = zig_attempt(pcg_output(st), st, inc, zt);
        const uint64_t multi = __ballot(z.len > 1);
        int skip_out;
        const uint64_t starts = zig_chain_starts(multi, skip, [&](int p) { return __builtin_amdgcn_readlane(z.len, p); }, skip_out);
        const uint64_t em = starts & __ballot(z.emit);
        const int idx = produced + popc_u64(em & below);
        if (((em >> lane) & 1ull) && idx < B) out[idx] = 0.0 + g.scale * z.val;  // loc + scale * z (random_normal)
        const int cnt = popc_u64(em);
        if (produced + cnt >= B) {
            const int e = nth_set_bit(em, B - produced - 1);
            state = pcg_advance(s, inc, g.ja, g.jg, e + __builtin_amdgcn_readlane(z.len, e));
            return;
        }
        produced += cnt;
        skip = skip_out;
        s = pcg_advance(s, inc, g.ja, g.jg, 64);
    }
}

__device__ __forceinline__ U128 uniform_u128(U128 v)
{
    U128 o;
    o.hi = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v.hi >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)v.hi);
    o.lo = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v.lo >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)v.lo);
    return o;
}

// rows [row0, row1) of the shared stream into the row cache, one after the other (a row's start
// depends on how many draws the previous rows consumed); rowstate[r] = stream position at row r.
__global__ void __launch_bounds__(64) k_noise_cache(NoiseGen g, U128 inc, U128 *__restrict__ rowstate, double *__restrict__ cache, int row0,
                                                    int row1, int B)
{
    U128 state = uniform_u128(rowstate[row0]);
    for (int r = row0; r < row1; ++r) {
        noise_row_wave(g, state, inc, cache + (size_t)r * B, B);
        if ((threadIdx.x & 63u) == 0u) rowstate[r + 1] = state;
    }
}

// this step's noise row of every agent the row cache cannot serve (per-agent streams; episodes
// longer than the cache), generated into the agent's own scans[] row where the scan kernel picks it up
// (RayHdr::noise_row == -2).  A 256-thread workgroup first checks `agents_per_block` agents one lane each
// and lists the ones that need a row; its four waves then generate the listed rows, a wave per row.  In
// the shared-stream mode the list is almost always empty, and a launch of N / 256 workgroups that look
// and leave costs a couple of microseconds — the host cannot know the longest live episode without a
// round trip (re-seats happen on the device), so the launch is unconditional once the steps since the
// last full reset exceed the cache.
__global__ void __launch_bounds__(256) k_noise_rows(AgentArrays a, NoiseGen g, int B, int agents_per_block)
{
    __shared__ int list[256];
    __shared__ int n_listed;
    if (threadIdx.x == 0) n_listed = 0;
    __syncthreads();
    const int end = a.agent_begin + a.agent_count;
    const int i = a.agent_begin + (int)blockIdx.x * agents_per_block + (int)threadIdx.x;
    if ((int)threadIdx.x < agents_per_block && i < end && (a.noise_rng == 2 || a.step_count[i] >= a.noise_rows))
        list[atomicAdd(&n_listed, 1)] = i;
    __syncthreads();
    const int n = n_listed;
    for (int q = (int)(threadIdx.x >> 6); q < n; q += 4) {
        const int ag = __builtin_amdgcn_readfirstlane(list[q]);
        const int k = __builtin_amdgcn_readfirstlane(a.step_count[ag]);
        U128 state, inc;
        if (a.noise_rng == 2) {
            inc = uniform_u128(a.rng_seed[2 * (size_t)ag + 1]);
            state = uniform_u128(k == 0 ? a.rng_seed[2 * (size_t)ag] : a.rng_state[ag]);
        } else {
            inc = a.rng_inc;
            state = uniform_u128(k == a.noise_rows ? a.rng_rowstate[a.noise_rows] : a.rng_state[ag]);
        }
        noise_row_wave(g, state, inc, a.scans + (size_t)ag * B, B);
        if ((threadIdx.x & 63u) == 0u) a.rng_state[ag] = state;
    }
}

// In-place re-seat of one agent of a finished env (f110_reset_collided_device / auto re-seat): what
// k_reset does for a masked env.  collisions[] keeps the step's value, as Simulator.collisions does,
// and so does in_collision[] here (k_integrate clears it at the next step; other waves of this
// kernel may still be reading the ego's flag).
__device__ __forceinline__ void reseat_agent(const AgentArrays &a, int i, bool count)
{
    const int N = a.n_agents_total;
#pragma unroll
    for (int c = 0; c < 7; ++c) a.state[(size_t)c * N + i] = 0.;
    a.state[i] = a.reseat_poses[3 * (size_t)i];
    a.state[(size_t)N + i] = a.reseat_poses[3 * (size_t)i + 1];
    a.state[4 * (size_t)N + i] = a.reseat_poses[3 * (size_t)i + 2];
    a.steer_buf[i] = 0.;
    a.steer_buf[(size_t)N + i] = 0.;
    a.buf_cnt[i] = 0;
    a.step_count[i] = 0;
    if (count && a.reseat_count) atomicAdd(a.reseat_count, 1);
}

// ---- K3: finalize ---------------------------------------------------------------------------
// kFinalizeLanes lanes per agent.  A typical opponent window is ~36 beams and what one agent costs
// is a chain of memory round trips and float64 latencies, not throughput: with many agents, 16
// lanes each (four agents per wave) means a quarter of the waves for the same chains; with few
// agents the whole wave per agent (one pass over the window) is the shorter chain.
// RaceCar.check_ttc's side effects (:246-252), Simulator's collision OR (:588-589), then
// RaceCar.ray_cast_agents (:206-227): opponents from the :574 snapshot, ego pose = live state
// (heading already zeroed on a wall hit), box = the ego's own params.
template <int kFinalizeLanes>
__global__ void __launch_bounds__(256) k_finalize(AgentArrays a, int32_t B)
{
    constexpr int kFinalizeAgents = 256 / kFinalizeLanes;   // per 256-thread workgroup
    const int i = a.agent_begin + (int)(blockIdx.x * kFinalizeAgents + threadIdx.x / kFinalizeLanes), tid = threadIdx.x & (kFinalizeLanes - 1);
    const int N = a.n_agents_total, A = a.agents_per_env;
    if (i >= a.agent_begin + a.agent_count) return;
    // everything the wave may need is requested up front (one round trip), not behind the flag
    const int wall = a.in_collision[i];
    const double ex = a.state[i], ey = a.state[(size_t)N + i];
    const double th_live = a.state[4 * (size_t)N + i];
    const int me = i % A;
    const double eth = wall ? 0.0 : th_live;
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
    double *sc = a.scans + (size_t)i * B;
    for (int jj = 0; jj < A; ++jj) {
        if (jj == me) continue;
        const int4 w4 = *reinterpret_cast<const int4 *>(a.opp_window + ((size_t)i * A + jj) * 4);
        const double *ov = a.opp_verts + ((size_t)i * A + jj) * 8;
        double v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = ov[c];
        const int lo = wall ? w4.z : w4.x, hi = wall ? w4.w : w4.y;
        if (hi < lo) continue;  // nothing of this opponent can be hit
        for (int b = lo + tid; b <= hi; b += kFinalizeLanes) {
            const double bt = eth + a.scan_angles[b];
            const double r0 = sc[b];
            double v3x, v3y;
            sincos(bt + kPi / 2., &v3y, &v3x);  // one argument reduction for both (get_range :259-260)
            const double r = box_range(ex, ey, v3x, v3y, v, r0);
            if (r < r0) sc[b] = r;
        }
        // the next opponent may touch the same beams: make this wave's stores visible to it
        __threadfence_block();
    }
    if (a.reseat_poses && tid == 0) {
        // the ego's collisions value of this step = pair test (k_collide, complete) OR its wall flag
        // (k_scan_rays, complete); neither input is written by this kernel's re-seat
        const int ego = (i / A) * A + a.reseat_ego;
        if (a.collisions[ego] != 0.0 || a.in_collision[ego] != 0) reseat_agent(a, i, i == ego);
    }
}

// ---- f110_step_host's work as the pair kernel's epilogue (round 4) -----------------------------------------------
// What k_host_block does in a launch of its own, done by the agent threads of k_finalize_pair_roles (lanes t < AG of
// wave 0; the two agents of an env are lanes t, t ^ 1) once the kernel's own work is finished: the agent's columns into
// the caller's page-locked block, F110Env._check_done for the env, the in-place re-seat.  Same arithmetic as
// k_host_block / k_episode (tests: the VecEnv forms agree, the reference's 2-agent episodes).
__device__ __forceinline__ void pair_host_epilogue_with(const AgentArrays &a, const EpisodeArrays &ep, const HostBlock &hb, const int episode, const int auto_reset,
                                                        bool agent_thread, int i)
{
    const size_t N = (size_t)a.n_agents_total;
    double col = 0.;
    int running = 0;
    double ct = 0.;
    if (agent_thread) {
        const size_t iu = (size_t)i;
        const double x = a.state[iu], y = a.state[N + iu];
        if (hb.state) {
            hb.state[iu] = x;
            hb.state[N + iu] = y;
#pragma unroll
            for (int c = 2; c < 7; ++c) hb.state[c * N + iu] = a.state[c * N + iu];
        }
        col = a.collisions[iu];
        if (hb.collisions) hb.collisions[iu] = col;
        if (hb.collision_idx) hb.collision_idx[iu] = a.collision_idx[iu];
        if (hb.in_collision) hb.in_collision[iu] = a.in_collision[iu];
        if (hb.agent_poses) {
#pragma unroll
            for (int c = 0; c < 3; ++c) hb.agent_poses[c * N + iu] = a.snap_pose[c * N + iu];
        }
        if (episode) {
            const int e = i >> 1;
            ct = ep.current_time[e] + ep.timestep;  // f110_env.py:295 (written back by the env's even lane below)
            const double r00 = ep.rot[4 * e], r01 = ep.rot[4 * e + 1], r10 = ep.rot[4 * e + 2], r11 = ep.rot[4 * e + 3];
            const double left_t = 2, right_t = 2;
            const double px = x - ep.start_poses[3 * iu];
            const double py = y - ep.start_poses[3 * iu + 1];
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
            bool near = ep.near_start[iu] != 0;
            double tog = ep.toggle[iu];
            if (closes && !near) {
                near = true;
                tog += 1;
            } else if (!closes && near) {
                near = false;
                tog += 1;
            }
            const double laps = floor(tog / 2);  // toggle_list // 2
            ep.near_start[iu] = near ? 1 : 0;
            ep.toggle[iu] = tog;
            ep.lap_count[iu] = laps;
            double lt = ep.lap_time[iu];
            if (tog < 4) {
                lt = ct;
                ep.lap_time[iu] = ct;
                running = 1;
            }
            ep.checkpoint[iu] = tog >= 4 ? 1 : 0;
            if (hb.lap_time) hb.lap_time[iu] = lt;
            if (hb.lap_count) hb.lap_count[iu] = laps;
            if (hb.toggle) hb.toggle[iu] = tog;
            if (hb.near_start) hb.near_start[iu] = near ? 1 : 0;
            if (hb.checkpoint) hb.checkpoint[iu] = tog >= 4 ? 1 : 0;
        }
    }
    // the env's other agent is the neighbouring lane (every lane of the wave takes part in the shuffles)
    const double col_o = __shfl_xor(col, 1);
    const int run_o = __shfl_xor(running, 1);
    if (episode && agent_thread) {
        const size_t iu = (size_t)i;
        const int me = i & 1, e = i >> 1;
        const double ego_col = (me == ep.ego_idx) ? col : col_o;
        const bool done = ego_col != 0.0 || (running == 0 && run_o == 0);  // :244
        const bool reseat = auto_reset && done;
        if (me == 0) {
            if (hb.current_time) hb.current_time[e] = ct;
            if (hb.done) hb.done[e] = done ? 1 : 0;
            ep.done[e] = (done && !reseat) ? 1 : 0;
            ep.current_time[e] = reseat ? 0. : ct;
        }
        if (reseat) {   // F110Env.reset :319-334 without its zero-action step
#pragma unroll
            for (int c = 0; c < 7; ++c) a.state[c * N + iu] = 0.;
            a.state[iu] = ep.start_poses[3 * iu];
            a.state[N + iu] = ep.start_poses[3 * iu + 1];
            a.state[4 * N + iu] = ep.start_poses[3 * iu + 2];
            a.steer_buf[iu] = 0.;
            a.steer_buf[N + iu] = 0.;
            a.buf_cnt[iu] = 0;
            a.in_collision[iu] = 0;
            a.step_count[iu] = 0;
            ep.near_start[iu] = 1;
            ep.toggle[iu] = 0.;
        }
    }
}

// ... with the parameters where the launcher left them in device memory (k_finalize_pair_roles; k_step_tiny's general body)
__device__ __forceinline__ void pair_host_epilogue(const AgentArrays &a, bool agent_thread, int i)
{
    const FusedHost *fp = a.fused_host;
    const EpisodeArrays ep = fp->ep;
    const HostBlock hb = fp->hb;
    pair_host_epilogue_with(a, ep, hb, fp->episode, fp->auto_reset, agent_thread, i);
}

// ---- K3r: finalize for two-agent envs — pair test, opponent window and ray-cast in one kernel (rounds 2-3) ----------
// What k_collide prepares for k_finalize — the GJK flag of the env's one pair and the beam window the opponent can
// occupy — costs a second stream and an event fork/join per step (14 us); with two agents per env it sits at the top of
// the finalize kernel instead.  The workgroup's threads are dealt by ROLE: the 4 AG box corners (-> beam indices) fill
// waves 0-1, the AG disc culls sit in wave 2, the AG / 2 pair tests (ONE per env: both agents' calls are the same
// gjk_overlap(box(2e), box(2e + 1)) on the same numbers) in wave 3; each wave runs only its own piece, the pieces run side
// by side on the CU's four SIMDs, results meet in LDS.  Then the opponent windows of the workgroup's AG agents are
// flattened into one item list (a crashed pair sees windows of up to all beams; fixed lanes per agent serialise there).
// Same functions on the same operands as k_collide + k_finalize: bit-identical.  (The earlier forms — fixed lanes per
// agent, the prologue dealt by agent — were measured slower in round 3 and retired in round 5: DESIGN_HISTORY.md.)
// (Round 5 also ran this kernel as one-wave workgroups at 8 waves per SIMD, NT = 64, so that a second env block's finalize fits under
// the first block's scan: it hides, and costs what it hides — retired in round 6, DESIGN.md §8, profiles/r05_finalize_wave.txt.)
// The body is a device function of a 256-thread workgroup over the AG agents [first, first + AG) (round 6): k_finalize_pair_roles
// runs it once per workgroup; k_step_tiny — the whole step of a tiny batch as ONE launch — runs it in its last workgroup.
template <int AG, bool HOST>
__device__ __forceinline__ void finalize_pair_body(const AgentArrays &a, int32_t B, const int first, const int end)
{
    constexpr int NT = 256;
    static_assert((AG & (AG - 1)) == 0 && AG >= 2 && 4 * AG <= NT / 2, "AG is a power of two, 2..32");
    constexpr int R1 = 128, R2 = 192;   // first thread of the cull / the pair-test role
    static_assert(R2 + AG / 2 <= NT, "the three roles fit the workgroup");
    __shared__ double s_rec[AG][12];   // ex, ey, eth, the opponent's box (8), pad
    __shared__ int s_idx[AG][4], s_cl[AG], s_ch[AG], s_hit[AG / 2];
    __shared__ int s_lo[AG], s_cnt[AG], s_off[AG + 1];
    const int t = (int)threadIdx.x;
    const int N = a.n_agents_total;
    int role = -1, slot = 0, sub = 0;
    if (t < 4 * AG) {                        // waves 0-1: box corner `sub` of agent `slot`'s opponent -> beam index
        role = 0; slot = t >> 2; sub = t & 3;
    } else if (t >= R1 && t < R1 + AG) {   // wave 2: disc cull of agent `slot`
        role = 1; slot = t - R1;
    } else if (t >= R2 && t < R2 + AG / 2) {   // wave 3: the pair test of env (slot, slot + 1)
        role = 2; slot = 2 * (t - R2);
    }
    if (role >= 0 && first + slot < end) {
        const int i = first + slot, me = i & 1, o = i ^ 1;
        const double ex = a.state[i], ey = a.state[(size_t)N + i];
        const double th_live = a.state[4 * (size_t)N + i];   // == the :574 snapshot heading: nothing has zeroed it yet
        const double ox = a.snap_pose[o], oy = a.snap_pose[(size_t)N + o], oth = a.snap_pose[2 * (size_t)N + o];
        if (role == 2) {
            // collision_multiple on the env's one pair, boxes with the Simulator's length / width (:549); i is the
            // even agent: gjk_overlap(mine, other) here is gjk_overlap(other, mine) of the odd agent's call
            int hit = 0;
            const double reach = sqrt(a.box_length * a.box_length + a.box_width * a.box_width) + 1e-3;
            const double cdx = ox - ex, cdy = oy - ey;
            if (cdx * cdx + cdy * cdy <= reach * reach) {
                double mine[8], other[8];
                box_vertices(ex, ey, th_live, a.box_length, a.box_width, mine);
                box_vertices(ox, oy, oth, a.box_length, a.box_width, other);
                hit = gjk_overlap(mine, other) ? 1 : 0;
            }
            s_hit[slot >> 1] = hit;
        } else {
            const int wall = a.in_collision[i];
            const double eth = wall ? 0.0 : th_live;
            const size_t prow = (size_t)(a.params_per_agent ? i : me) * NPARAMS;
            const double blen = a.params[prow + P_LENGTH], bwid = a.params[prow + P_WIDTH];
            double v[8];   // the opponent drawn with MY length / width (RaceCar.ray_cast_agents :223)
            box_vertices(ox, oy, oth, blen, bwid, v);
            double ce_, se_;
        cos_sin(eth, ce_, se_);
        const double head = atan2(se_, ce_);
            const double px = role == 1 ? ox : (sub == 0 ? v[0] : (sub == 1 ? v[2] : (sub == 2 ? v[4] : v[6])));
            const double py = role == 1 ? oy : (sub == 0 ? v[1] : (sub == 1 ? v[3] : (sub == 2 ? v[5] : v[7])));
            const double dx = px - ex, dy = py - ey;
            const double norm = sqrt(dx * dx + dy * dy);
            const double qx = role == 0 ? dx / norm : dx, qy = role == 0 ? dy / norm : dy;
            const double dir = atan2(qy, qx);
            if (role == 0) {
                s_idx[slot][sub] = vertex_beam_from_angles(head, dir, a.scan_angles, B, a.angle_inc);
                if (sub == 0) {
                    s_rec[slot][0] = ex;
                    s_rec[slot][1] = ey;
                    s_rec[slot][2] = eth;
#pragma unroll
                    for (int c = 0; c < 8; ++c) s_rec[slot][3 + c] = v[c];
                }
            } else {
                int cl, ch;
                disc_beam_range_from(norm, eth, dir, head, 0.5 * sqrt(blen * blen + bwid * bwid), a.scan_angles, B, a.angle_inc, cl, ch);
                s_cl[slot] = cl;
                s_ch[slot] = ch;
            }
        }
    }
    __syncthreads();
    int my_hit = 0;
    const bool agent_thread = t < AG && first + t < end;
    if (t < AG) {
        int lo = 0, cnt = 0;
        if (agent_thread) {
            const int i = first + t, me = i & 1;
            const int i0 = s_idx[t][0], i1 = s_idx[t][1], i2 = s_idx[t][2], i3 = s_idx[t][3];
            const int cl = s_cl[t], ch = s_ch[t];
            my_hit = s_hit[t >> 1];
            int ref_lo = i0 < i1 ? i0 : i1, t2 = i2 < i3 ? i2 : i3;
            ref_lo = ref_lo < t2 ? ref_lo : t2;
            int ref_hi = i0 > i1 ? i0 : i1;
            t2 = i2 > i3 ? i2 : i3;
            ref_hi = ref_hi > t2 ? ref_hi : t2;
            lo = ref_lo > cl ? ref_lo : cl;
            const int hi = ref_hi < ch ? ref_hi : ch;
            cnt = hi >= lo ? hi - lo + 1 : 0;
            const int wall = a.in_collision[i];
            if (wall) {
                a.state[3 * (size_t)N + i] = 0.;
                a.state[4 * (size_t)N + i] = 0.;
                a.state[5 * (size_t)N + i] = 0.;
                a.state[6 * (size_t)N + i] = 0.;
            }
            a.collisions[i] = (my_hit || wall) ? 1.0 : 0.0;
            a.collision_idx[i] = my_hit ? (double)(1 - me) : -1.0;
            a.step_count[i] += 1;
        }
        s_lo[t] = lo;
        s_cnt[t] = cnt;
    }
    __syncthreads();
    if (t < 64) {   // exclusive scan of the AG window lengths (AG <= 32: one wave)
        int c = t < AG ? s_cnt[t] : 0;
#pragma unroll
        for (int d = 1; d < AG; d <<= 1) {
            const int up = __shfl_up(c, d);
            if (t >= d) c += up;
        }
        if (t < AG) s_off[t + 1] = c;
        if (t == 0) s_off[0] = 0;
    }
    __syncthreads();
    const int total = s_off[AG];
    for (int item = t; item < total; item += NT) {
        int ag = 0;   // the largest ag with s_off[ag] <= item (its window is not empty: item < s_off[ag + 1])
#pragma unroll
        for (int st = AG / 2; st; st >>= 1)
            if (s_off[ag + st] <= item) ag += st;
        const int b = s_lo[ag] + (item - s_off[ag]);
        const double bex = s_rec[ag][0], bey = s_rec[ag][1], beth = s_rec[ag][2];
        double bv[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) bv[c] = s_rec[ag][3 + c];
        double *sc = a.scans + (size_t)(first + ag) * B;
        const double bt = beth + a.scan_angles[b];
        const double r0 = sc[b];
        double v3x, v3y;
        sincos(bt + kPi / 2., &v3y, &v3x);
        const double r = box_range(bex, bey, v3x, v3y, bv, r0);
        if (r < r0) sc[b] = r;
    }
    if (HOST) {   // f110_step_host: the host block, the episode logic and its re-seat, right here (a.fused_host is set by the launcher)
        if (a.fused_host->hb.scans) {   // small batches: this workgroup's agents' scans, final once every item above is done
            __syncthreads();
            const int live_agents = (end - first) < AG ? (end - first) : AG;
            host_block_copy_scans(a.fused_host->hb, a.scans, (size_t)first, (size_t)(live_agents > 0 ? live_agents : 0));
        }
        if (t < 64) pair_host_epilogue(a, agent_thread, first + t);
    } else if (a.reseat_poses && agent_thread) {
        const int i = first + t, ego = (i & ~1) + a.reseat_ego;
        if (my_hit || a.in_collision[ego] != 0) reseat_agent(a, i, i == ego);
    }
}

// finalize_pair_body for ONE env of two agents that is the whole batch (k_step_tiny, N = 2): the same functions on the same operands,
// without the machinery for 32 agents per workgroup — the window lengths need no prefix scan (each half of the workgroup takes one
// agent's window), the agent's own columns are finished by the lane that runs its host epilogue (no barrier between the two), and the
// scans go into the host block EARLY: wave 1, which has no role, copies the 2 x B ranges while the role lanes do their trigonometry,
// and the beams the opponent shortens follow as they are found — instead of a 17 KB copy at the end of the chain.  (Storing them from
// the scan waves was tried first: their release then waits for the stores' acknowledgements over PCIe, +2.4 us where -2.6 were saved.)
// Three barriers instead of six on the step's critical path (profiles/r06_launch_latency.txt: the tail was 9.4 of the kernel's 32 us).
template <bool HOST>
__device__ __forceinline__ void finalize_duo_tiny(const AgentArrays &a, int32_t B, double *__restrict__ host_scans, const FusedHost &fh)
{
    __shared__ double d_rec[2][12];   // ex, ey, eth, the opponent's box (8), pad
    __shared__ int d_idx[2][4], d_cl[2], d_ch[2], d_hit;
    const int t = (int)threadIdx.x;
    constexpr int N = 2;
    int role = -1, slot = 0, sub = 0;
    if (t < 8) {                        // box corner `sub` of agent `slot`'s opponent -> beam index
        role = 0; slot = t >> 2; sub = t & 3;
    } else if (t >= 128 && t < 130) {   // wave 2: disc cull of agent `slot`
        role = 1; slot = t - 128;
    } else if (t == 192) {              // wave 3: the pair test
        role = 2;
    }
    if (HOST && host_scans && t >= 64 && t < 128) {
        const size_t n = (size_t)N * (size_t)B;
        size_t q = (size_t)(t - 64);
        for (; q + 192 < n; q += 256) {   // four independent loads in flight per lane, then the four stores
            const double v0 = a.scans[q], v1 = a.scans[q + 64], v2 = a.scans[q + 128], v3 = a.scans[q + 192];
            host_scans[q] = v0;
            host_scans[q + 64] = v1;
            host_scans[q + 128] = v2;
            host_scans[q + 192] = v3;
        }
        for (; q < n; q += 64) host_scans[q] = a.scans[q];
        __builtin_amdgcn_s_waitcnt(0);   // acknowledged before a window lane may store a shortened beam over one of them
    }
    if (role >= 0) {
        const int i = slot, me = i & 1, o = i ^ 1;
        const double ex = a.state[i], ey = a.state[(size_t)N + i];
        const double th_live = a.state[4 * (size_t)N + i];
        const double ox = a.snap_pose[o], oy = a.snap_pose[(size_t)N + o], oth = a.snap_pose[2 * (size_t)N + o];
        if (role == 2) {
            int hit = 0;
            const double reach = sqrt(a.box_length * a.box_length + a.box_width * a.box_width) + 1e-3;
            const double cdx = ox - ex, cdy = oy - ey;
            if (cdx * cdx + cdy * cdy <= reach * reach) {
                double mine[8], other[8];
                box_vertices(ex, ey, th_live, a.box_length, a.box_width, mine);
                box_vertices(ox, oy, oth, a.box_length, a.box_width, other);
                hit = gjk_overlap(mine, other) ? 1 : 0;
            }
            d_hit = hit;
        } else {
            const int wall = a.in_collision[i];
            const double eth = wall ? 0.0 : th_live;
            const size_t prow = (size_t)(a.params_per_agent ? i : me) * NPARAMS;
            const double blen = a.params[prow + P_LENGTH], bwid = a.params[prow + P_WIDTH];
            double v[8];
            box_vertices(ox, oy, oth, blen, bwid, v);
            double ce_, se_;
            cos_sin(eth, ce_, se_);
            const double head = atan2(se_, ce_);
            const double px = role == 1 ? ox : (sub == 0 ? v[0] : (sub == 1 ? v[2] : (sub == 2 ? v[4] : v[6])));
            const double py = role == 1 ? oy : (sub == 0 ? v[1] : (sub == 1 ? v[3] : (sub == 2 ? v[5] : v[7])));
            const double dx = px - ex, dy = py - ey;
            const double norm = sqrt(dx * dx + dy * dy);
            const double qx = role == 0 ? dx / norm : dx, qy = role == 0 ? dy / norm : dy;
            const double dir = atan2(qy, qx);
            if (role == 0) {
                d_idx[slot][sub] = vertex_beam_from_angles(head, dir, a.scan_angles, B, a.angle_inc);
                if (sub == 0) {
                    d_rec[slot][0] = ex;
                    d_rec[slot][1] = ey;
                    d_rec[slot][2] = eth;
#pragma unroll
                    for (int c = 0; c < 8; ++c) d_rec[slot][3 + c] = v[c];
                }
            } else {
                int cl, ch;
                disc_beam_range_from(norm, eth, dir, head, 0.5 * sqrt(blen * blen + bwid * bwid), a.scan_angles, B, a.angle_inc, cl, ch);
                d_cl[slot] = cl;
                d_ch[slot] = ch;
            }
        }
    }
    __syncthreads();
    // agent t's own columns, by lane t of wave 0 — the lane that runs its host epilogue below
    const bool agent_thread = t < N;
    const int my_hit = d_hit;
    if (agent_thread) {
        const int i = t, me = i & 1;
        const int wall = a.in_collision[i];
        if (wall) {
            a.state[3 * (size_t)N + i] = 0.;
            a.state[4 * (size_t)N + i] = 0.;
            a.state[5 * (size_t)N + i] = 0.;
            a.state[6 * (size_t)N + i] = 0.;
        }
        a.collisions[i] = (my_hit || wall) ? 1.0 : 0.0;
        a.collision_idx[i] = my_hit ? (double)(1 - me) : -1.0;
        a.step_count[i] += 1;
    }
    {   // agent `ag`'s opponent window, by its half of the workgroup
        const int ag = t >> 7, tt = t & 127;
        const int i0 = d_idx[ag][0], i1 = d_idx[ag][1], i2 = d_idx[ag][2], i3 = d_idx[ag][3];
        const int cl = d_cl[ag], ch = d_ch[ag];
        int ref_lo = i0 < i1 ? i0 : i1, t2 = i2 < i3 ? i2 : i3;
        ref_lo = ref_lo < t2 ? ref_lo : t2;
        int ref_hi = i0 > i1 ? i0 : i1;
        t2 = i2 > i3 ? i2 : i3;
        ref_hi = ref_hi > t2 ? ref_hi : t2;
        const int lo = ref_lo > cl ? ref_lo : cl;
        const int hi = ref_hi < ch ? ref_hi : ch;
        const int cnt = hi >= lo ? hi - lo + 1 : 0;
        const double bex = d_rec[ag][0], bey = d_rec[ag][1], beth = d_rec[ag][2];
        double bv[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) bv[c] = d_rec[ag][3 + c];
        double *sc = a.scans + (size_t)ag * B;
        for (int item = tt; item < cnt; item += 128) {
            const int b = lo + item;
            const double bt = beth + a.scan_angles[b];
            const double r0 = sc[b];
            double v3x, v3y;
            sincos(bt + kPi / 2., &v3y, &v3x);
            const double r = box_range(bex, bey, v3x, v3y, bv, r0);
            if (r < r0) {
                sc[b] = r;
                if (HOST && host_scans) host_scans[(size_t)ag * B + b] = r;
            }
        }
    }
    if (HOST) {   // (fh: the epilogue's parameters as kernel arguments — a.fused_host holds the same in device memory, a load behind every barrier here)
        if (fh.hb.scans && !host_scans) {   // (not copied early: the copy, once every window is done)
            __syncthreads();
            host_block_copy_scans(fh.hb, a.scans, 0, (size_t)N);
        }
        if (t < 64) pair_host_epilogue_with(a, fh.ep, fh.hb, fh.episode, fh.auto_reset, agent_thread, t);
    } else if (a.reseat_poses && agent_thread) {
        const int i = t, ego = a.reseat_ego;
        if (my_hit || a.in_collision[ego] != 0) reseat_agent(a, i, i == ego);
    }
}

// HOST (round 6): the f110_step_host epilogue is a TEMPLATE parameter, not a run-time branch — the instantiation f110_step_device
// launches does not carry the epilogue's registers; the HOST instantiation (small, host-synchronised batches) runs at 4 waves per SIMD.
#ifndef F110_FIN_WAVES
#define F110_FIN_WAVES 5   // 5 waves per SIMD = up to 96 VGPRs: the body needs 82 and keeps nothing in scratch (at 6 = 80 VGPRs it kept 80 bytes per lane
                           // there for the epilogue it never ran; 35.9 us per launch at 65 536 agents at 4, 5 and 6 waves alike: profiles/r06_finalize_waves.txt)
#endif
template <int AG, bool HOST = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(HOST ? 4 : F110_FIN_WAVES))) k_finalize_pair_roles(AgentArrays a, int32_t B)
{
    finalize_pair_body<AG, HOST>(a, B, a.agent_begin + (int)blockIdx.x * AG, a.agent_begin + a.agent_count);
    if (HOST && a.fused_seq) {   // F110_STEP_SPIN_WAIT: the completion word, by the last workgroup to get here
        HostBlock sig = a.fused_host->hb;
        sig.seq = a.fused_seq;
        host_block_signal(sig);
    }
}

// ---- K3m: the same for envs of MORE than two agents (round 3) ---------------------------------------------
// k_finalize_pair_roles generalised: a workgroup holds G whole envs of A agents.  What k_collide + k_finalize did
// with one lane per agent (a serial loop over the A - 1 opponents: box, GJK, two beam windows, 80 bytes of windows
// and boxes through HBM per ordered pair, a side stream with an event fork / join around the scan) becomes one
// kernel behind the scan:
//   agents    (round 4) what a record needs of its two agents, once per agent into LDS (MultiAgentLds);
//   records   every ORDERED pair (agent i, opponent o) of the workgroup's envs, R = G A (A - 1) <= MAXREC (64 for A <= 8,
//             256 with one env per workgroup for A <= 16): the four
//             corners of o's box (drawn with i's length / width, RaceCar.ray_cast_agents :223) -> beam indices on
//             threads 0-191 (four per record), the disc cull on threads 192-255 (one per record);
//   pairs     every UNORDERED pair (p < q) of an env, P = G A (A - 1) / 2: collision_multiple's GJK in the reference's
//             (lower, higher) argument order on threads 192-255 behind the culls — once, not once per side;
//   agents    flags (Simulator's collision OR :588-589; collision_idx = the LARGEST colliding partner, the reference's
//             last writer), check_ttc's side effects, step count, re-seat: thread a of the first G A;
//   windows   all records' beam windows flattened into one item list over the 256 threads, as in the pair kernel.  Two
//             opponents of one agent may cover the same beam; the reference takes them one after the other, each
//             keeping the smaller range (:206-227), i.e. the minimum — ranges are non-negative doubles, whose bit
//             patterns order like unsigned integers, so the items settle it with atomicMin on the pattern: no rounds,
//             no order, the same value.
// Same functions on the same operands as collide_agent / k_finalize: bit-identical (test_finalize_multi_*).
constexpr int kMaxRec = 64;      // A <= 8: a workgroup's record table
constexpr int kMaxRecBig = 256;  // 9 <= A <= 16: one env per workgroup, up to 240 records
// phase 1 of k_finalize_multi: 4 R corner tasks (an atan2, a square root, two divisions each) on the first three waves, R disc
// culls and then R / 2 pair tests on the fourth (round 4; round 3: corners on two waves, 7.5 rounds of them at 16 cars).
// What a task needs of its two AGENTS — ray-cast heading, atan2(sin, cos) of it, cos / sin of the opponent's snapshot
// heading — is computed once per agent in a phase of its own (MultiAgentLds), not 4 (A - 1) times (doubling probes at 16
// cars, round 4: corners 77 us, culls 17, pair tests 5, window items 135 of the kernel's 250).
constexpr int kMultiCornerThreads = 192;
struct MultiAgentLds {   // one per agent of the workgroup
    double ex, ey, eth, head;      // live position, ray-cast heading, atan2(sin(eth), cos(eth))
    double ox, oy, co, so;         // :574 snapshot position, cos / sin of the snapshot heading
    double blen, bwid, rad;        // the box this agent draws around its opponents (RaceCar.ray_cast_agents :223: its OWN length / width), half its diagonal
};
template <int MAXREC>
__global__ void __launch_bounds__(256) k_finalize_multi(AgentArrays a, int32_t B, int G)
{
    static_assert(MAXREC == 64 || MAXREC == 256, "record table sizes");
    __shared__ double s_rec[MAXREC][12];   // ex, ey, eth, the opponent's box (8), pad
    __shared__ int s_idx[MAXREC][4], s_cl[MAXREC], s_ch[MAXREC], s_hit[MAXREC];
    __shared__ int s_lo[MAXREC], s_off[MAXREC + 1], s_agent[MAXREC], s_ahit[MAXREC], s_wsum[4];
    constexpr int MAXAG = MAXREC == 64 ? 32 : 16;   // agents per workgroup: G A (A - 1) <= 64 -> G A <= 32; the 256-record table takes ONE env of <= 16
    __shared__ MultiAgentLds s_ag[MAXAG];
    const int t = (int)threadIdx.x;
    const int A = a.agents_per_env, N = a.n_agents_total;
    const int per_env = A * (A - 1), pairs_env = per_env / 2;
    const int first = a.agent_begin + (int)blockIdx.x * G * A, end = a.agent_begin + a.agent_count;
    int envs = (end - first) / A;
    envs = envs < G ? envs : G;                 // (whole envs only: ranges are env-aligned)
    const int R = envs * per_env, P = envs * pairs_env, AGN = envs * A;
    if (t < AGN) {   // everything that is per agent, once
        const int i = first + t;
        MultiAgentLds m;
        m.ex = a.state[i];
        m.ey = a.state[(size_t)N + i];
        const double th_live = a.state[4 * (size_t)N + i];   // == the :574 snapshot heading: nothing has zeroed it yet
        m.eth = a.in_collision[i] ? 0.0 : th_live;
        double ce_, se_;
        cos_sin(m.eth, ce_, se_);
        m.head = atan2(se_, ce_);
        m.ox = a.snap_pose[i];
        m.oy = a.snap_pose[(size_t)N + i];
        cos_sin(a.snap_pose[2 * (size_t)N + i], m.co, m.so);
        const size_t prow = (size_t)(a.params_per_agent ? i : t % A) * NPARAMS;
        m.blen = a.params[prow + P_LENGTH];
        m.bwid = a.params[prow + P_WIDTH];
        m.rad = 0.5 * sqrt(m.blen * m.blen + m.bwid * m.bwid);
        s_ag[t] = m;
    }
    __syncthreads();
    if (t < kMultiCornerThreads) {
        // corner `sub` of record `rec`'s opponent box -> beam index
        for (int q = t; q < 4 * R; q += kMultiCornerThreads) {
            const int rec = q >> 2, sub = q & 3;
            const int e = rec / per_env, w = rec - e * per_env, me = w / (A - 1), k = w - me * (A - 1), oj = k < me ? k : k + 1;
            const MultiAgentLds &mm = s_ag[e * A + me], &mo = s_ag[e * A + oj];
            const double ex = mm.ex, ey = mm.ey, eth = mm.eth, head = mm.head;
            double v[8];
            box_vertices_cs(mo.ox, mo.oy, mo.co, mo.so, mm.blen, mm.bwid, v);
            const double px = sub == 0 ? v[0] : (sub == 1 ? v[2] : (sub == 2 ? v[4] : v[6]));
            const double py = sub == 0 ? v[1] : (sub == 1 ? v[3] : (sub == 2 ? v[5] : v[7]));
            const double dx = px - ex, dy = py - ey;
            const double norm = sqrt(dx * dx + dy * dy);
            const double dir = atan2(dy / norm, dx / norm);
            s_idx[rec][sub] = vertex_beam_from_angles(head, dir, a.scan_angles, B, a.angle_inc);
            if (sub == 0) {
                s_rec[rec][0] = ex;
                s_rec[rec][1] = ey;
                s_rec[rec][2] = eth;
#pragma unroll
                for (int c = 0; c < 8; ++c) s_rec[rec][3 + c] = v[c];
                s_agent[rec] = e * A + me;
            }
        }
    } else {
        // the disc cull of record `rec`
        for (int rec = t - kMultiCornerThreads; rec < R; rec += 256 - kMultiCornerThreads) {
            const int e = rec / per_env, w = rec - e * per_env, me = w / (A - 1), k = w - me * (A - 1), oj = k < me ? k : k + 1;
            const MultiAgentLds &mm = s_ag[e * A + me], &mo = s_ag[e * A + oj];
            const double eth = mm.eth, head = mm.head;
            const double dx = mo.ox - mm.ex, dy = mo.oy - mm.ey;
            const double norm = sqrt(dx * dx + dy * dy);
            const double dir = atan2(dy, dx);
            int cl, ch;
            disc_beam_range_from(norm, eth, dir, head, mm.rad, a.scan_angles, B, a.angle_inc, cl, ch);
            s_cl[rec] = cl;
            s_ch[rec] = ch;
        }
        // collision_multiple's pair (p < q) of env e, boxes with the Simulator's length / width (:549)
        const double reach = sqrt(a.box_length * a.box_length + a.box_width * a.box_width) + 1e-3;
        for (int pr = t - kMultiCornerThreads; pr < P; pr += 256 - kMultiCornerThreads) {
            const int e = pr / pairs_env;
            int w = pr - e * pairs_env, p = 0;
            while (w >= A - 1 - p) { w -= A - 1 - p; ++p; }   // row p of the upper triangle holds A - 1 - p pairs
            const int q = p + 1 + w;
            const MultiAgentLds &mp = s_ag[e * A + p], &mq = s_ag[e * A + q];
            int hit = 0;
            const double cdx = mq.ox - mp.ox, cdy = mq.oy - mp.oy;
            if (cdx * cdx + cdy * cdy <= reach * reach) {
                double lower[8], higher[8];
                box_vertices_cs(mp.ox, mp.oy, mp.co, mp.so, a.box_length, a.box_width, lower);
                box_vertices_cs(mq.ox, mq.oy, mq.co, mq.so, a.box_length, a.box_width, higher);
                hit = gjk_overlap(lower, higher) ? 1 : 0;
            }
            s_hit[pr] = hit;
        }
    }
    __syncthreads();
    int my_hit = 0;
    const bool agent_thread = t < AGN;
    if (agent_thread) {
        const int e = t / A, me = t - e * A, i = first + t;
        int partner = -1;
        for (int j = 0; j < A; ++j) {   // ascending: ends at the largest colliding index
            if (j == me) continue;
            const int p = me < j ? me : j, q = me < j ? j : me;
            const int pr = e * pairs_env + p * (A - 1) - (p * (p - 1)) / 2 + (q - p - 1);   // rows 0..p-1 hold p (A - 1) - p (p - 1) / 2 pairs
            if (s_hit[pr]) partner = j;
        }
        my_hit = partner >= 0;
        const int wall = a.in_collision[i];
        if (wall) {
            a.state[3 * (size_t)N + i] = 0.;
            a.state[4 * (size_t)N + i] = 0.;
            a.state[5 * (size_t)N + i] = 0.;
            a.state[6 * (size_t)N + i] = 0.;
        }
        a.collisions[i] = (my_hit || wall) ? 1.0 : 0.0;
        a.collision_idx[i] = (double)partner;
        a.step_count[i] += 1;
        s_ahit[t] = my_hit;
    }
    const int wrec = MAXREC == 64 ? t - 64 : t;   // 64 records: wave 1 (beside the agents' wave); 256: every thread one record
    if (wrec >= 0 && wrec < MAXREC) {   // every record's window = corner hull clipped by the disc cull
        const int rec = wrec;
        int lo = 0, cnt = 0;
        if (rec < R) {
            const int i0 = s_idx[rec][0], i1 = s_idx[rec][1], i2 = s_idx[rec][2], i3 = s_idx[rec][3];
            int ref_lo = i0 < i1 ? i0 : i1, t2 = i2 < i3 ? i2 : i3;
            ref_lo = ref_lo < t2 ? ref_lo : t2;
            int ref_hi = i0 > i1 ? i0 : i1;
            t2 = i2 > i3 ? i2 : i3;
            ref_hi = ref_hi > t2 ? ref_hi : t2;
            const int cl = s_cl[rec], ch = s_ch[rec];
            lo = ref_lo > cl ? ref_lo : cl;
            const int hi = ref_hi < ch ? ref_hi : ch;
            cnt = hi >= lo ? hi - lo + 1 : 0;
        }
        s_lo[rec] = lo;
        int c = cnt;   // inclusive scan over the wave's 64 records
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(c, d);
            if ((rec & 63) >= d) c += up;
        }
        s_off[rec + 1] = c;
        if (rec == 0) s_off[0] = 0;
        if (MAXREC > 64 && (rec & 63) == 63) s_wsum[rec >> 6] = c;
    }
    __syncthreads();
    if (MAXREC > 64) {   // the later waves' records start behind the earlier waves' totals
        if (t >= 64 && t < MAXREC) {
            int base = 0;
            for (int w = 0; w < (t >> 6); ++w) base += s_wsum[w];
            s_off[t + 1] += base;
        }
        __syncthreads();
    }
    const int total = s_off[MAXREC];
    for (int item = t; item < total; item += 256) {
        int rec = 0;   // the largest rec with s_off[rec] <= item (its window is not empty: item < s_off[rec + 1])
#pragma unroll
        for (int st = MAXREC / 2; st; st >>= 1)
            if (s_off[rec + st] <= item) rec += st;
        const int b = s_lo[rec] + (item - s_off[rec]);
        const double bex = s_rec[rec][0], bey = s_rec[rec][1], beth = s_rec[rec][2];
        double bv[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) bv[c] = s_rec[rec][3 + c];
        double *sc = a.scans + (size_t)(first + s_agent[rec]) * B;
        const double bt = beth + a.scan_angles[b];
        const double r0 = sc[b];   // (possibly already lowered by another opponent's item: the minimum does not care)
        double v3x, v3y;
        sincos(bt + kPi / 2., &v3y, &v3x);
        const double r = box_range(bex, bey, v3x, v3y, bv, r0);
        if (r < r0) atomicMin(reinterpret_cast<unsigned long long *>(sc + b), (unsigned long long)__double_as_longlong(r));
    }
    if (a.reseat_poses && agent_thread) {
        const int e = t / A, i = first + t, ego_t = e * A + a.reseat_ego, ego = first + ego_t;
        if (s_ahit[ego_t] || a.in_collision[ego] != 0) reseat_agent(a, i, i == ego);
    }
}

// ---- K3t: envs of MORE than sixteen agents (round 4) — k_finalize_multi with the env's ordered pairs in TILES -------
// k_finalize_multi for any number of agents per env (17 ... 256; it also runs 3 ... 16, where k_finalize_multi's role-parallel
// form is faster: A = 8 75.7 against 79.3 M agent-steps/s, 16: equal): a workgroup holds G whole envs of A agents.  What k_collide + k_finalize did with one lane per agent (a serial loop over the A - 1 opponents: box, GJK, two
// beam windows, 80 bytes of windows and boxes through HBM per ordered pair, a side stream with an event fork / join
// around the scan) is one kernel behind the scan:
//   agents    once per agent, into LDS: the live pose and the heading the ray cast uses (0 after a wall hit, :246-249),
//             head = atan2(sin, cos) of it (get_blocked_view_indices :296-297), and cos / sin of the agent's :574
//             snapshot heading (what every box drawn around it needs) — per AGENT, not per ordered pair (round 4: the
//             record threads of round 3 recomputed them 4 (A - 1) times);
//   pairs     every UNORDERED pair (p < q) of an env: collision_multiple's GJK in the reference's (lower, higher)
//             argument order behind a conservative reject (centres further apart than the boxes' diagonal: the
//             prune north_star asks for; SURVEY a15: flag-preserving) — the largest colliding partner per agent by
//             atomicMax (collision_multiple's last writer, :199-210);
//   records   every ORDERED pair (agent i, opponent o), in tiles of at most MAXREC: first the disc cull (beams whose
//             ray can touch the disc around o's box), then — only for records the cull leaves a window — the four
//             corners of o's box (drawn with i's length / width, RaceCar.ray_cast_agents :223) -> beam indices, four
//             threads per record;
//   windows   a tile's beam windows flattened into one item list over the 256 threads.  An item whose map range is
//             already shorter than the distance to the nearest point of o's disc is finished (every edge distance is
//             at least that far: the minimum keeps the map range).  Two opponents of one agent may cover the same
//             beam; the reference takes them one after the other, each keeping the smaller range (:206-227), i.e. the
//             minimum — ranges are non-negative doubles, whose bit patterns order like unsigned integers, so the items
//             settle it with atomicMin on the pattern: no rounds, no order, the same value — also across tiles.
//   flags     Simulator's collision OR (:588-589), check_ttc's side effects, step count, re-seat: one thread per agent.
// Same functions on the same operands as collide_agent / k_finalize: bit-identical (test_finalize_multi_*,
// tests/golden/sim_rollout_multi.npz = the reference itself at A = 3, 4, 8).
constexpr int kMaxAgentsMulti = 256;   // agents per env this kernel takes (its per-agent LDS table)
// (MultiAgentLds: above, shared with k_finalize_multi)
// dynamic LDS of k_finalize_multi for `agents` agents per workgroup (host and device agree through this one function):
// the agent table, a tile's boxes / near distances, a tile's int tables, the scan offsets, the partners
__host__ __device__ inline size_t multi_lds_bytes(int agents, int maxrec)
{
    return sizeof(MultiAgentLds) * (size_t)agents + (size_t)maxrec * 9 * sizeof(double) + sizeof(int) * ((size_t)maxrec * 8 + (size_t)maxrec + 1 + 4 + (size_t)agents) + 16;
}

template <int MAXREC>
__global__ void __launch_bounds__(256) k_finalize_multi_tiled(AgentArrays a, int32_t B, int G)
{
    static_assert(MAXREC == 64 || MAXREC == 256, "record tile sizes");
    extern __shared__ __align__(16) unsigned char s_raw[];
    const int t = (int)threadIdx.x;
    const int A = a.agents_per_env, N = a.n_agents_total;
    const int per_env = A * (A - 1), pairs_env = per_env / 2;
    const int first = a.agent_begin + (int)blockIdx.x * G * A, end = a.agent_begin + a.agent_count;
    int envs = (end - first) / A;
    envs = envs < G ? envs : G;                 // (whole envs only: ranges are env-aligned)
    const int R = envs * per_env, P = envs * pairs_env, AGN = envs * A;
    // LDS carve-up (dynamic: sized by the launch for G * A agents, multi_lds_bytes)
    MultiAgentLds *s_ag = reinterpret_cast<MultiAgentLds *>(s_raw);
    double (*s_box)[8] = reinterpret_cast<double (*)[8]>(s_ag + G * A);   // a record's opponent box
    double *s_near = reinterpret_cast<double *>(s_box + MAXREC);          // a map range below this cannot be lowered by the record
    int (*s_idx)[4] = reinterpret_cast<int (*)[4]>(s_near + MAXREC);
    int *s_cl = reinterpret_cast<int *>(s_idx + MAXREC), *s_ch = s_cl + MAXREC, *s_lo = s_ch + MAXREC, *s_agent = s_lo + MAXREC;
    int *s_off = s_agent + MAXREC;      // [MAXREC + 1]
    int *s_wsum = s_off + MAXREC + 1;   // [4]
    int *s_partner = s_wsum + 4;        // [G * A] largest colliding partner (slot index) or -1

    // ---- agents: everything that is per agent, once
    for (int ag = t; ag < AGN; ag += 256) {
        const int i = first + ag;
        MultiAgentLds m;
        m.ex = a.state[i];
        m.ey = a.state[(size_t)N + i];
        const double th_live = a.state[4 * (size_t)N + i];   // == the :574 snapshot heading: nothing has zeroed it yet
        m.eth = a.in_collision[i] ? 0.0 : th_live;
        double ce_, se_;
        cos_sin(m.eth, ce_, se_);
        m.head = atan2(se_, ce_);
        m.ox = a.snap_pose[i];
        m.oy = a.snap_pose[(size_t)N + i];
        cos_sin(a.snap_pose[2 * (size_t)N + i], m.co, m.so);
        const size_t prow = (size_t)(a.params_per_agent ? i : ag % A) * NPARAMS;
        m.blen = a.params[prow + P_LENGTH];
        m.bwid = a.params[prow + P_WIDTH];
        m.rad = 0.5 * sqrt(m.blen * m.blen + m.bwid * m.bwid);
        s_ag[ag] = m;
        s_partner[ag] = -1;
    }
    __syncthreads();

    // ---- pairs: collision_multiple's pair (p < q) of the env, boxes with the Simulator's length / width (:549)
    {
        const double reach = sqrt(a.box_length * a.box_length + a.box_width * a.box_width) + 1e-3;
        for (int pr = t; pr < P; pr += 256) {
            const int e = pr / pairs_env;
            int w = pr - e * pairs_env, p = 0;
            while (w >= A - 1 - p) { w -= A - 1 - p; ++p; }   // row p of the upper triangle holds A - 1 - p pairs
            const int q = p + 1 + w;
            const MultiAgentLds &mp = s_ag[e * A + p], &mq = s_ag[e * A + q];
            const double cdx = mq.ox - mp.ox, cdy = mq.oy - mp.oy;
            if (cdx * cdx + cdy * cdy <= reach * reach) {
                double lower[8], higher[8];
                box_vertices_cs(mp.ox, mp.oy, mp.co, mp.so, a.box_length, a.box_width, lower);
                box_vertices_cs(mq.ox, mq.oy, mq.co, mq.so, a.box_length, a.box_width, higher);
                if (gjk_overlap(lower, higher)) {
                    atomicMax(&s_partner[e * A + p], q);
                    atomicMax(&s_partner[e * A + q], p);
                }
            }
        }
    }

    // ---- records, tile by tile
    for (int tile = 0; tile < R; tile += MAXREC) {
        const int RT = (R - tile) < MAXREC ? (R - tile) : MAXREC;   // records of this tile
        // the disc cull of record `rec` (and who it is)
        for (int rec = t; rec < RT; rec += 256) {
            const int gr = tile + rec;
            const int e = gr / per_env, w = gr - e * per_env, me = w / (A - 1), k = w - me * (A - 1), oj = k < me ? k : k + 1;
            const int ia = e * A + me;
            const MultiAgentLds &mm = s_ag[ia], &mo = s_ag[e * A + oj];
            const double blen = mm.blen, bwid = mm.bwid, rad = mm.rad;
            const double dx = mo.ox - mm.ex, dy = mo.oy - mm.ey;
            const double norm = sqrt(dx * dx + dy * dy);
            const double dir = atan2(dy, dx);
            int cl, ch;
            disc_beam_range_from(norm, mm.eth, dir, mm.head, rad, a.scan_angles, B, a.angle_inc, cl, ch);
            s_cl[rec] = cl;
            s_ch[rec] = ch;
            s_agent[rec] = ia;
            s_near[rec] = norm - rad * 1.000001 - 1e-9;   // no point of the box is nearer than this
            if (cl <= ch) box_vertices_cs(mo.ox, mo.oy, mo.co, mo.so, blen, bwid, s_box[rec]);
        }
        __syncthreads();
        // corner `sub` of record `rec`'s opponent box -> beam index (records with an empty cull are skipped)
        for (int q = t; q < 4 * RT; q += 256) {
            const int rec = q >> 2, sub = q & 3;
            if (s_cl[rec] > s_ch[rec]) continue;
            const MultiAgentLds &mm = s_ag[s_agent[rec]];
            const double px = s_box[rec][2 * sub], py = s_box[rec][2 * sub + 1];
            const double dx = px - mm.ex, dy = py - mm.ey;
            const double norm = sqrt(dx * dx + dy * dy);
            const double dir = atan2(dy / norm, dx / norm);
            s_idx[rec][sub] = vertex_beam_from_angles(mm.head, dir, a.scan_angles, B, a.angle_inc);
        }
        __syncthreads();
        if (t < MAXREC) {   // every record's window = corner hull clipped by the disc cull
            const int rec = t;
            int lo = 0, cnt = 0;
            if (rec < RT && s_cl[rec] <= s_ch[rec]) {
                const int i0 = s_idx[rec][0], i1 = s_idx[rec][1], i2 = s_idx[rec][2], i3 = s_idx[rec][3];
                int ref_lo = i0 < i1 ? i0 : i1, t2 = i2 < i3 ? i2 : i3;
                ref_lo = ref_lo < t2 ? ref_lo : t2;
                int ref_hi = i0 > i1 ? i0 : i1;
                t2 = i2 > i3 ? i2 : i3;
                ref_hi = ref_hi > t2 ? ref_hi : t2;
                const int cl = s_cl[rec], ch = s_ch[rec];
                lo = ref_lo > cl ? ref_lo : cl;
                const int hi = ref_hi < ch ? ref_hi : ch;
                cnt = hi >= lo ? hi - lo + 1 : 0;
            }
            s_lo[rec] = lo;
            int c = cnt;   // inclusive scan over the wave's 64 records
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int up = __shfl_up(c, d);
                if ((rec & 63) >= d) c += up;
            }
            s_off[rec + 1] = c;
            if (rec == 0) s_off[0] = 0;
            if (MAXREC > 64 && (rec & 63) == 63) s_wsum[rec >> 6] = c;
        }
        __syncthreads();
        if (MAXREC > 64) {   // the later waves' records start behind the earlier waves' totals
            if (t >= 64 && t < MAXREC) {
                int base = 0;
                for (int w = 0; w < (t >> 6); ++w) base += s_wsum[w];
                s_off[t + 1] += base;
            }
            __syncthreads();
        }
        const int total = s_off[MAXREC];
        for (int item = t; item < total; item += 256) {
            int rec = 0;   // the largest rec with s_off[rec] <= item (its window is not empty: item < s_off[rec + 1])
#pragma unroll
            for (int st = MAXREC / 2; st; st >>= 1)
                if (s_off[rec + st] <= item) rec += st;
            const int b = s_lo[rec] + (item - s_off[rec]);
            const int ia = s_agent[rec];
            double *sc = a.scans + (size_t)(first + ia) * B;
            const double r0 = sc[b];   // (possibly already lowered by another opponent's item: the minimum does not care)
            if (r0 < s_near[rec]) continue;   // the wall is nearer than any point of this opponent
            const MultiAgentLds &mm = s_ag[ia];
            double bv[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) bv[c] = s_box[rec][c];
            const double bt = mm.eth + a.scan_angles[b];
            double v3x, v3y;
            sincos(bt + kPi / 2., &v3y, &v3x);
            const double r = box_range(mm.ex, mm.ey, v3x, v3y, bv, r0);
            if (r < r0) atomicMin(reinterpret_cast<unsigned long long *>(sc + b), (unsigned long long)__double_as_longlong(r));
        }
        __syncthreads();   // the next tile overwrites the record tables
    }

    // ---- flags, check_ttc's side effects, step count, re-seat
    __syncthreads();
    for (int ag = t; ag < AGN; ag += 256) {
        const int i = first + ag;
        const int partner = s_partner[ag];
        const int wall = a.in_collision[i];
        if (wall) {
            a.state[3 * (size_t)N + i] = 0.;
            a.state[4 * (size_t)N + i] = 0.;
            a.state[5 * (size_t)N + i] = 0.;
            a.state[6 * (size_t)N + i] = 0.;
        }
        a.collisions[i] = (partner >= 0 || wall) ? 1.0 : 0.0;
        a.collision_idx[i] = (double)partner;
        a.step_count[i] += 1;
    }
    if (a.reseat_poses) {
        for (int ag = t; ag < AGN; ag += 256) {
            const int e = ag / A, i = first + ag, ego_t = e * A + a.reseat_ego, ego = first + ego_t;
            if (s_partner[ego_t] >= 0 || a.in_collision[ego] != 0) reseat_agent(a, i, i == ego);
        }
    }
}

// single-agent envs: no opponents, one lane per agent is enough
__device__ __forceinline__ void finalize_solo_agent(const AgentArrays &a, int i)
{
    const int N = a.n_agents_total;
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
    if (a.reseat_poses && wall) reseat_agent(a, i, true);
}

__global__ void __launch_bounds__(256) k_finalize_solo(AgentArrays a)
{
    const int i = a.agent_begin + (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= a.agent_begin + a.agent_count) return;
    finalize_solo_agent(a, i);
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
// (struct EpisodeArrays: defined next to AgentArrays, the finalize kernels take it too)

// (envs env0 .. env0 + num_envs - 1: the whole batch, or one env block on that block's stream)
__global__ void __launch_bounds__(256) k_episode(AgentArrays a, EpisodeArrays ep, int num_envs, int env0 = 0)
{
    const int e = env0 + (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (e >= env0 + num_envs) return;
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

// every per-agent / per-env scalar a host-side RL loop reads per step, packed into one block so that
// it crosses PCIe in one copy (f110_episode_step_host): 9 double columns [N], current_time [E], then
// the byte flags near_starts [N], checkpoint_done [N], done [E]
__global__ void __launch_bounds__(256) k_pack_episode(AgentArrays a, EpisodeArrays ep, int num_envs, double *__restrict__ cols, uint8_t *__restrict__ flags)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = a.n_agents_total;
    if (i < N) {
        cols[i] = a.state[i];
        cols[(size_t)N + i] = a.state[(size_t)N + i];
        cols[2 * (size_t)N + i] = a.state[4 * (size_t)N + i];
        cols[3 * (size_t)N + i] = a.state[3 * (size_t)N + i];
        cols[4 * (size_t)N + i] = a.state[5 * (size_t)N + i];
        cols[5 * (size_t)N + i] = a.collisions[i];
        cols[6 * (size_t)N + i] = ep.lap_time[i];
        cols[7 * (size_t)N + i] = ep.lap_count[i];
        cols[8 * (size_t)N + i] = ep.toggle[i];
        flags[i] = ep.near_start[i];
        flags[(size_t)N + i] = ep.checkpoint[i];
    }
    if (i < num_envs) {
        cols[9 * (size_t)N + i] = ep.current_time[i];
        flags[2 * (size_t)N + i] = ep.done[i];
    }
}

// f110_step_host: what a host-driven loop reads after a step, written by ONE kernel straight into the caller's
// page-locked block (the pointers are device-visible host memory: the stores cross PCIe as posted writes, no
// staging buffer and no copy command behind the kernel).  With the episode logic on (f110_episode_init) the
// same kernel is F110Env._check_done (k_episode's arithmetic, f110_env.py:204-246) and, with auto_reset, the
// in-place re-seat of finished envs (k_episode_reset_done) AFTER their terminal observation has been written.
// A workgroup owns whole envs (envs_per_block of them): phase 1 per agent (toggles + the agent's columns),
// phase 2 per env (done, current_time), phase 3 per agent (re-seat).  Any pointer may be nullptr.
// (struct HostBlock, host_block_signal: defined next to AgentArrays)
// the completion word of a launch whose LAST workgroup runs the host block by itself (k_step_tiny): no counting
__device__ __forceinline__ void host_block_signal_single(const HostBlock &hb)
{
    if (!hb.seq_host) return;
    __builtin_amdgcn_s_waitcnt(0); // every wave's stores into the host block have been acknowledged ...
    __syncthreads();               // ... in every wave of the workgroup
    if (threadIdx.x == 0) {
        __threadfence_system();    // ... and visible system-wide before the word
        __hip_atomic_store(hb.seq_host, hb.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// The body of k_host_block for the envs [e0, e0 + ne) of a 256-thread workgroup.  SINGLE: the workgroup is the only one that runs
// it in its launch (k_step_tiny's last workgroup) and signals by itself.
template <bool SINGLE>
__device__ __forceinline__ void host_block_body(const AgentArrays &a, const EpisodeArrays &ep, const HostBlock &hb, const int e0, const int ne, int episode,
                                                int auto_reset)
{
    __shared__ int s_running[256];    // per local env: agents with fewer than 4 toggles
    __shared__ uint8_t s_done[256];
    const int A = a.agents_per_env, tid = threadIdx.x;
    const size_t N = (size_t)a.n_agents_total;
    if (episode) {
        for (int le = tid; le < ne; le += 256) s_running[le] = 0;
        __syncthreads();
    }
    const int items = ne * A;
    for (int idx = tid; idx < items; idx += 256) {
        const int le = idx / A;
        const size_t i = (size_t)e0 * A + idx;
        const double x = a.state[i], y = a.state[N + i];
        if (hb.state) {
            hb.state[i] = x;
            hb.state[N + i] = y;
#pragma unroll
            for (int c = 2; c < 7; ++c) hb.state[c * N + i] = a.state[c * N + i];
        }
        if (hb.collisions) hb.collisions[i] = a.collisions[i];
        if (hb.collision_idx) hb.collision_idx[i] = a.collision_idx[i];
        if (hb.in_collision) hb.in_collision[i] = a.in_collision[i];
        if (hb.agent_poses) {
#pragma unroll
            for (int c = 0; c < 3; ++c) hb.agent_poses[c * N + i] = a.snap_pose[c * N + i];
        }
        if (episode) {
            const int e = e0 + le;
            const double ct = ep.current_time[e] + ep.timestep;  // f110_env.py:295 (written back in phase 2)
            const double r00 = ep.rot[4 * e], r01 = ep.rot[4 * e + 1], r10 = ep.rot[4 * e + 2], r11 = ep.rot[4 * e + 3];
            const double left_t = 2, right_t = 2;
            const double px = x - ep.start_poses[3 * i];
            const double py = y - ep.start_poses[3 * i + 1];
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
            const double laps = floor(tog / 2);  // toggle_list // 2
            ep.near_start[i] = near ? 1 : 0;
            ep.toggle[i] = tog;
            ep.lap_count[i] = laps;
            double lt = ep.lap_time[i];
            if (tog < 4) {
                lt = ct;
                ep.lap_time[i] = ct;
                atomicAdd(&s_running[le], 1);
            }
            ep.checkpoint[i] = tog >= 4 ? 1 : 0;
            if (hb.lap_time) hb.lap_time[i] = lt;
            if (hb.lap_count) hb.lap_count[i] = laps;
            if (hb.toggle) hb.toggle[i] = tog;
            if (hb.near_start) hb.near_start[i] = near ? 1 : 0;
            if (hb.checkpoint) hb.checkpoint[i] = tog >= 4 ? 1 : 0;
        }
    }
    if (hb.scans) host_block_copy_scans(hb, a.scans, (size_t)e0 * A, (size_t)items);
    if (!episode) {
        if (SINGLE) host_block_signal_single(hb);
        else host_block_signal(hb);
        return;
    }
    __syncthreads();
    for (int le = tid; le < ne; le += 256) {
        const int e = e0 + le;
        const bool done = a.collisions[(size_t)e * A + ep.ego_idx] != 0.0 || s_running[le] == 0;  // :244
        const double ct = ep.current_time[e] + ep.timestep;
        if (hb.current_time) hb.current_time[e] = ct;
        if (hb.done) hb.done[e] = done ? 1 : 0;
        s_done[le] = done ? 1 : 0;
        const bool reseat = auto_reset && done;
        ep.done[e] = (done && !reseat) ? 1 : 0;
        ep.current_time[e] = reseat ? 0. : ct;
    }
    if (SINGLE) host_block_signal_single(hb);   // the host block is complete here; the re-seat below touches device memory only
    else host_block_signal(hb);
    if (!auto_reset) return;
    __syncthreads();         // s_done
    for (int idx = tid; idx < items; idx += 256) {
        if (!s_done[idx / A]) continue;
        const size_t i = (size_t)e0 * A + idx;   // F110Env.reset :319-334 without its zero-action step
#pragma unroll
        for (int c = 0; c < 7; ++c) a.state[c * N + i] = 0.;
        a.state[i] = ep.start_poses[3 * i];
        a.state[N + i] = ep.start_poses[3 * i + 1];
        a.state[4 * N + i] = ep.start_poses[3 * i + 2];
        a.steer_buf[i] = 0.;
        a.steer_buf[N + i] = 0.;
        a.buf_cnt[i] = 0;
        a.in_collision[i] = 0;
        a.step_count[i] = 0;
        ep.near_start[i] = 1;
        ep.toggle[i] = 0.;
    }
}

__global__ void __launch_bounds__(256) k_host_block(AgentArrays a, EpisodeArrays ep, HostBlock hb, int num_envs, int envs_per_block,
                                                    int episode, int auto_reset)
{
    const int e0 = blockIdx.x * envs_per_block;
    host_block_body<false>(a, ep, hb, e0, min(envs_per_block, num_envs - e0), episode, auto_reset);   // ne >= 1: the grid is ceil(num_envs / envs_per_block)
}

// ---- K0: the WHOLE step of a tiny batch as one launch (round 6) ------------------------------------------------------------
// The reference's own shape — F110Env(num_agents = 2).step, one env — is launch- and sync-bound on the GPU: three kernels of 9 + 14 + 9
// us are reported after ~60 us (profiles/r06_f110env_breakdown.txt).  For a host-synchronised step of N <= kTinyMaxAgents agents, 1 or 2 per env, this
// kernel is the step: Simulator.step (base_classes.py:553-612) — update_pose for every agent, the scans, the iTTC test, the pair test,
// the opponent ray-cast — and, under f110_step_host, F110Env._check_done (f110_env.py:204-246) + the observation block in the caller's
// page-locked memory + the completion word the host polls.  Same device functions on the same operands as the three kernels: bit-identical
// (tests/test_gpu_round6.py).
//   scan phase    one wave per 64-beam task, four per workgroup.  A wave needs its agent's lidar pose and ray header, i.e. the agent's
//                 INTEGRATION: every lane computes it (advance_vehicle + make_ray_hdr: the same instructions one lane would run), so
//                 no kernel boundary and no HBM round trip separates integrate from scan.  Nothing live is overwritten meanwhile: the
//                 wave of an agent's FIRST task stores the new state / delay buffer / poses / header into SHADOW columns, the wall
//                 flags of the iTTC test go to a flag column of their own.
//   last block    workgroups count themselves done (one release per workgroup, acquire by the last, at agent scope); the one that arrives
//                 last runs the finalize body (finalize_duo_tiny / finalize_pair_body / finalize_solo_agent) and the host epilogue ON the
//                 shadow columns, stores the completion word, and only then copies the shadow columns over the live ones.
struct TinyCtl {
    unsigned int *done;      // workgroups finished this launch (left at 0 by the last one)
    int32_t *wall;           // [N] this step's iTTC flags (left at 0 by the last workgroup)
    double *state;           // shadow of AgentArrays: [7][N]
    double *steer_buf;       // [2][N]
    int32_t *buf_cnt;        // [N]
    double *scan_pose;       // [3][N]
    double *snap_pose;       // [3][N]
    double *dir_start;       // [N]
    RayHdr *ray_hdr;         // [N]
    uint32_t tasks_per_agent, pad_;
    unsigned long long *trace;   // lab timeline probe: [workgroups][16] stamps of the 100 MHz clock (nullptr = off)
    double *host_scans;          // N = 2: the caller's page-locked scans [2][B] (device view) — finalize_duo_tiny copies the ranges there while its
                                 // role lanes work, and the beams the opponent shortens as they are found (nullptr: no scans in the block)
    FusedHost fh;                // N = 2 under f110_step_host: the host epilogue's parameters (what a.fused_host points at), by value
    int act_inline;              // 1: the actions are `act` below (f110_step_host copied them into the launch: no read over PCIe inside the step)
    int skip;                    // lab (2 = N == 2 takes finalize_pair_body instead of finalize_duo_tiny: the A/B of the tail); 1 = the launch only reports start and completion (what does the dispatch itself cost?)
    double act[8];               // [kTinyMaxAgents][2] (steer, velocity)
    unsigned long long *start_word, start_seq;   // lab: a word in page-locked host memory the first workgroup stores start_seq to as it starts
};
static_assert(sizeof(TinyCtl::act) == 8 * sizeof(double), "TinyCtl::act holds kTinyMaxAgents actions");
constexpr int kTinyMaxAgents = 4;   // (measured: beyond a handful of agents the per-kernel form wins, f110_hip.hip tiny_applies)

template <bool PAIR, bool IDENT, bool HOST>
__global__ void __launch_bounds__(256) k_step_tiny(AgentArrays a, ScanConst k, RayJob j, const double *__restrict__ actions, TinyCtl ctl, EpisodeArrays ep,
                                                   HostBlock hb, int episode, int auto_reset)
{
    __shared__ int s_last;
    const int N = a.n_agents_total, B = k.num_beams;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t task = blockIdx.x * 4u + wave;
#ifdef F110_EXPERIMENTAL
#define TINY_STAMP(slot)                                                                                         \
    do {                                                                                                         \
        if (ctl.trace && threadIdx.x == 0) ctl.trace[16ull * blockIdx.x + (slot)] = wall_clock64();             \
    } while (0)
#else
#define TINY_STAMP(slot) do { } while (0)
#endif
    TINY_STAMP(0);   // the workgroup's first wave is running
#ifdef F110_EXPERIMENTAL
    if (ctl.start_word && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(ctl.start_word, ctl.start_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (ctl.skip == 1) {   // the dispatch alone: start word, completion word, nothing else
        if (ctl.start_word && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(ctl.start_word - 1, ctl.start_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
#endif
    AgentArrays sh = a;      // the step's view: what integration produces and the tail works on
    sh.state = ctl.state; sh.steer_buf = ctl.steer_buf; sh.buf_cnt = ctl.buf_cnt; sh.scan_pose = ctl.scan_pose;
    sh.snap_pose = ctl.snap_pose; sh.dir_start = ctl.dir_start; sh.ray_hdr = ctl.ray_hdr; sh.in_collision = ctl.wall;
    if (task < (uint32_t)N * ctl.tasks_per_agent) {
        const int i = (int)(task / ctl.tasks_per_agent);
        const uint32_t sub = task - (uint32_t)i * ctl.tasks_per_agent;
        // RaceCar.update_pose for agent i, by every lane of the wave (k_integrate's body)
        const VehicleParams vp = load_params(a.params + (size_t)(a.params_per_agent ? i : i % a.agents_per_env) * NPARAMS);
        double st[7];
#pragma unroll
        for (int c = 0; c < 7; ++c) st[c] = a.state[(size_t)c * N + i];
        double b0 = a.steer_buf[i], b1 = a.steer_buf[(size_t)N + i];
        int cnt = a.buf_cnt[i];
        double2 act;
        if (ctl.act_inline) {   // (i < 4: selects on kernel arguments)
            act.x = i == 0 ? ctl.act[0] : i == 1 ? ctl.act[2] : i == 2 ? ctl.act[4] : ctl.act[6];
            act.y = i == 0 ? ctl.act[1] : i == 1 ? ctl.act[3] : i == 2 ? ctl.act[5] : ctl.act[7];
        } else {
            act = reinterpret_cast<const double2 *>(actions)[i];
        }
        double sp[3];
        advance_vehicle(st, b0, b1, cnt, act.x, act.y, vp, a.time_step, a.integrator, a.lidar_dist, sp);
        const double start = scan_start_index(k, sp[2]);
        const RayHdr hd = make_ray_hdr(a, k, i, st, sp, start);
#ifdef F110_EXPERIMENTAL
        if (ctl.trace) {
            double probe = hd.d0;
            asm volatile("" : "+v"(probe));   // (the header's table sample has arrived)
        }
#endif
        TINY_STAMP(1);   // integration + ray header done (wave 0 of the workgroup)
        if (sub == 0u && lane == 0u) {   // the agent's new columns, into the shadow (the live ones are still being read)
            integrate_store_columns(sh, i, N, st, b0, b1, cnt, sp, start);
            sh.ray_hdr[i] = hd;
        }
        // get_scan for the task's 64 beams (k_scan_rays_agent's body)
        const int b = (int)(sub * 64u + lane);
        if (b < B) {
            const int row = hd.noise_row;
            const double nz = row >= 0 ? j.noise[(size_t)row * B + b] : 0.0;
            const double2 cs = k.cs[beam_dir_index(k, start, b)];
            int hr = -1, hc = -1, nl = 0;
            double r = 0.;
            bool exact = hd.fast == 0;
            if (hd.fast) {
                double ux, uy, cux, cuy;
                padded_position<IDENT>(k, hd.x, hd.y, ux, uy);
                padded_rate<IDENT>(k, cs.x, cs.y, cux, cuy);
                exact = !march_padded<false>(k, ux, uy, cux, cuy, hd.d0, r, hr, hc, nl);
            }
            if (exact) r = march_exact_cold<IDENT>(j.k_cold, hd.x, hd.y, cs.x, cs.y, hd.d0, hr, hc, nl);
            if (row != -1) r += nz;
            // check_ttc_jit's predicate for this beam (finish_beam_with), the flag into the step's own column
            if (hd.vel != 0.0 && !(r > j.ttc_side_max + j.ttc_k * fabs(hd.vel)) && ttc_beam_hit(r, j.side_dist[b], hd.vel, j.beam_cos[b], j.ttc_thresh))
                sh.in_collision[i] = 1;
            a.scans[(size_t)i * B + b] = r;
        }
    }
    TINY_STAMP(2);   // wave 0's beams marched and stored
    // ---- who is last?
    // Release, once per workgroup: every wave waits for its own stores to be in L2 (vmcnt 0), the barrier collects the waves, and ONE
    // lane writes the L2 back at agent scope before it counts the workgroup done (a fence by every wave is 34 L2 write-backs where 9 do).
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    TINY_STAMP(3);   // every wave of the workgroup has its stores in L2
    if (threadIdx.x == 0) {
        __threadfence();     // this workgroup's rows, flags and shadow columns are visible device-wide (other XCDs' L2s included)
        const unsigned int prev = __hip_atomic_fetch_add(ctl.done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        s_last = prev == gridDim.x - 1u ? 1 : 0;
        if (s_last) __hip_atomic_store(ctl.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();     // acquire: what the other workgroups released
    TINY_STAMP(4);   // the last workgroup knows it is the last
    // ---- the last workgroup: Simulator.step's tail ON the shadow columns (sh; its in_collision is the step's flag column), the host
    //      block and the completion word — and only then shadow -> live, which nobody waits for
    TINY_STAMP(5);
    if (PAIR) {
        if (N == 2 && ctl.skip != 2) {   // the reference's own shape: one env of two cars  (skip 2: lab A/B, the general body)
            finalize_duo_tiny<HOST>(sh, B, HOST ? ctl.host_scans : nullptr, ctl.fh);
            TINY_STAMP(6);
            if (HOST && a.fused_seq) {
                HostBlock sig = ctl.fh.hb;
                sig.seq = a.fused_seq;
                host_block_signal_single(sig);
            }
            TINY_STAMP(7);
        } else {
            constexpr int AG = 4;
            for (int first = 0; first < N; first += AG) {
                finalize_pair_body<AG, HOST>(sh, B, first, N);
                __syncthreads();   // the body's LDS is reused by the next group
            }
            TINY_STAMP(6);   // finalize + host block stores issued
            if (HOST && a.fused_seq) {
                HostBlock sig = a.fused_host->hb;
                sig.seq = a.fused_seq;
                host_block_signal_single(sig);
            }
            TINY_STAMP(7);   // completion word stored
        }
    } else {
        for (int i = (int)threadIdx.x; i < N; i += 256) finalize_solo_agent(sh, i);
        if (HOST) {
            __syncthreads();
            host_block_body<true>(sh, ep, hb, 0, N, episode, auto_reset);   // one agent per env: N envs, all in this workgroup (N <= 64)
        }
        TINY_STAMP(7);
    }
    __syncthreads();
    for (int i = (int)threadIdx.x; i < N; i += 256) {
#pragma unroll
        for (int c = 0; c < 7; ++c) a.state[(size_t)c * N + i] = sh.state[(size_t)c * N + i];
        a.steer_buf[i] = sh.steer_buf[i];
        a.steer_buf[(size_t)N + i] = sh.steer_buf[(size_t)N + i];
        a.buf_cnt[i] = sh.buf_cnt[i];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            a.scan_pose[(size_t)c * N + i] = sh.scan_pose[(size_t)c * N + i];
            a.snap_pose[(size_t)c * N + i] = sh.snap_pose[(size_t)c * N + i];
        }
        a.dir_start[i] = sh.dir_start[i];
        a.ray_hdr[i] = sh.ray_hdr[i];
        a.in_collision[i] = sh.in_collision[i];
        sh.in_collision[i] = 0;      // (the flag column is left clear for the next launch)
    }
#undef TINY_STAMP
}

// The scalar part of Simulator.step's observation (base_classes.py:594-610), packed for the RCCL
// observation gather: [7][N] = poses_x, poses_y, poses_theta, linear_vels_x, linear_vels_y (always 0.,
// :603), ang_vels_z, collisions.  Next to the scans it is what a consumer on another GPU needs.
constexpr int kObsScalars = 7;
__global__ void __launch_bounds__(256) k_pack_obs(AgentArrays a, double *__restrict__ cols)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t N = (size_t)a.n_agents_total;
    if (i >= a.n_agents_total) return;
    cols[i] = a.state[i];
    cols[N + i] = a.state[N + i];
    cols[2 * N + i] = a.state[4 * N + i];
    cols[3 * N + i] = a.state[3 * N + i];
    cols[4 * N + i] = 0.;
    cols[5 * N + i] = a.state[5 * N + i];
    cols[6 * N + i] = a.collisions[i];
}

// ---- do two streams make progress independently of each other? ------------------------------------------------------
// ROCm maps HIP streams onto a few hardware queues; two streams on ONE queue run their kernels one after the other, and
// env groups (f110_step_device) then cost time instead of saving it.  The mapping cannot be queried, but it can be
// observed: a kernel on stream a waits (bounded: `ticks` of the 100 MHz clock) for a flag that a kernel enqueued
// afterwards on stream b sets.  It sees the flag only if b's kernel ran while it was waiting.
__global__ void k_probe_wait(unsigned *flag, unsigned *seen_out, unsigned long long ticks)
{
    const unsigned long long t0 = wall_clock64();
    unsigned seen = 0;
    while ((seen = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u && wall_clock64() - t0 < ticks)
        __builtin_amdgcn_s_sleep(16);
    *seen_out = seen;
}
__global__ void k_probe_set(unsigned *flag) { __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// float32 transport of the observation gather (f110_comm_gather_obs, F110_GATHER_F32): round-to-nearest of every range
__global__ void __launch_bounds__(256) k_scans_to_f32(const double *__restrict__ src, float *__restrict__ dst, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (float)src[i];
}

// re-seat every env whose done flag is set (F110Env.reset :319-334 without its zero-action step)
// (agents i0 .. i0 + count - 1; count < 0: all)
__global__ void __launch_bounds__(256) k_episode_reset_done(AgentArrays a, EpisodeArrays ep, int32_t *__restrict__ n_reset, int i0 = 0, int count = -1)
{
    const int i = i0 + (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int N = a.n_agents_total, A = a.agents_per_env;
    if (i >= (count < 0 ? N : i0 + count)) return;
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
__global__ void k_episode_clear_done(EpisodeArrays ep, int num_envs, int env0 = 0)
{
    const int e = env0 + (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (e < env0 + num_envs) ep.done[e] = 0;
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
                               const double *__restrict__ scan_angles, double angle_inc, int uniform, double *__restrict__ scans,
                               int32_t *__restrict__ minmax)
{
    const int p = blockIdx.x, tid = threadIdx.x;
    const double ex = ego[3 * p], ey = ego[3 * p + 1], eth = ego[3 * p + 2];
    double v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = verts[8 * (size_t)p + c];
    int ref_lo, ref_hi, lo, hi;
    if (uniform) {
        // circumscribed disc of the quadrilateral: centroid + largest vertex distance
        const double cx = (((v[0] + v[2]) + v[4]) + v[6]) / 4, cy = (((v[1] + v[3]) + v[5]) + v[7]) / 4;
        double r2 = 0.0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const double dx = v[2 * c] - cx, dy = v[2 * c + 1] - cy;
            r2 = fmax(r2, dx * dx + dy * dy);
        }
        opponent_beam_window(ex, ey, eth, v, cx, cy, sqrt(r2) * 1.000001, scan_angles, B, angle_inc, ref_lo, ref_hi, lo, hi);
    } else {
        // a scan_angles table that is not the uniform ramp (the free function ray_cast takes any array, laser_models.py:318):
        // the reference's own full argmin per vertex (:310-313), no closed-form estimate, no disc cull — every beam of the window
        __shared__ int vidx[4];
        if (tid < 4) vidx[tid] = nearest_beam_full(scan_angles, B, vertex_view_angle(ex, ey, eth, v[2 * tid], v[2 * tid + 1]));
        __syncthreads();
        ref_lo = min(min(vidx[0], vidx[1]), min(vidx[2], vidx[3]));
        ref_hi = max(max(vidx[0], vidx[1]), max(vidx[2], vidx[3]));
        lo = ref_lo;
        hi = ref_hi;
    }
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

// ---- f110_helper_batch: the small functions `from f110_gym.envs import *` also exposes in the reference, one item per thread.
// Each case is the reference function's own expression order (the same device functions the step kernels inline).
__global__ void k_helper_unit(int op, const double *__restrict__ in, int m, int n, int in_w, int out_w, double *__restrict__ out, ScanConst k)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const double *a = in + (size_t)in_w * i;
    double *o = out + (size_t)out_w * i;
    switch (op) {
    case F110_OP_ACCL_CONSTRAINTS: {   // dynamic_models.py:29-60  (vel, accl, v_switch, a_max, v_min, v_max)
        VehicleParams p;
        p.v[P_VSWITCH] = a[2]; p.v[P_AMAX] = a[3]; p.v[P_VMIN] = a[4]; p.v[P_VMAX] = a[5];
        o[0] = clamp_accel(a[0], a[1], p);
        break;
    }
    case F110_OP_STEERING_CONSTRAINT: {   // :62-87  (steering_angle, steering_velocity, s_min, s_max, sv_min, sv_max)
        VehicleParams p;
        p.v[P_SMIN] = a[2]; p.v[P_SMAX] = a[3]; p.v[P_SVMIN] = a[4]; p.v[P_SVMAX] = a[5];
        o[0] = clamp_steer_rate(a[0], a[1], p);
        break;
    }
    case F110_OP_CROSS:   // laser_models.py:219-230
        o[0] = a[0] * a[3] - a[1] * a[2];
        break;
    case F110_OP_ARE_COLLINEAR: {   // :232-247  (pt_a, pt_b, pt_c)
        const double bax = a[2] - a[0], bay = a[3] - a[1];
        const double cax = a[0] - a[4], cay = a[1] - a[5];
        o[0] = fabs(bax * cay - bay * cax) < 1e-8 ? 1.0 : 0.0;
        break;
    }
    case F110_OP_PERPENDICULAR:   // collision_models.py:34-48
        o[0] = a[1];
        o[1] = -1 * a[0];
        break;
    case F110_OP_TRIPLE_PRODUCT:   // :51-64  (a, b, c)
        triple_product(a[0], a[1], a[2], a[3], a[4], a[5], o[0], o[1]);
        break;
    case F110_OP_AVG_POINT: {   // :67-78  np.sum(vertices, axis=0) / n: the rows are added in order
        double sx = a[0], sy = a[1];
        for (int q = 1; q < n; ++q) {
            sx += a[2 * q];
            sy += a[2 * q + 1];
        }
        o[0] = sx / n;
        o[1] = sy / n;
        break;
    }
    case F110_OP_FURTHEST_POINT: {   // :81-92  np.argmax(vertices.dot(d)): first maximum wins  (vertices [n][2], d)
        const double dx = a[2 * n], dy = a[2 * n + 1];
        int best = 0;
        double bv = a[0] * dx + a[1] * dy;
        for (int q = 1; q < n; ++q) {
            const double val = a[2 * q] * dx + a[2 * q + 1] * dy;
            if (val > bv) {
                bv = val;
                best = q;
            }
        }
        o[0] = (double)best;
        break;
    }
    case F110_OP_SUPPORT: {   // :95-110  (vertices1 [n][2], vertices2 [n][2], d)
        const double *v1 = a, *v2 = a + 2 * n;
        const double dx = a[4 * n], dy = a[4 * n + 1];
        int bi = 0, bj = 0;
        double vi = v1[0] * dx + v1[1] * dy, vj = v2[0] * -dx + v2[1] * -dy;
        for (int q = 1; q < n; ++q) {
            const double t1 = v1[2 * q] * dx + v1[2 * q + 1] * dy, t2 = v2[2 * q] * -dx + v2[2 * q + 1] * -dy;
            if (t1 > vi) {
                vi = t1;
                bi = q;
            }
            if (t2 > vj) {
                vj = t2;
                bj = q;
            }
        }
        o[0] = v1[2 * bi] - v2[2 * bj];
        o[1] = v1[2 * bi + 1] - v2[2 * bj + 1];
        break;
    }
    case F110_OP_GET_TRMTX: {   // :218-235  H [4][4] row-major
        double c, s;
        cos_sin(a[2], c, s);
        const double H[16] = {c, -s, 0., a[0], s, c, 0., a[1], 0., 0., 1., 0., 0., 0., 0., 1.};
        for (int q = 0; q < 16; ++q) o[q] = H[q];
        break;
    }
    case F110_OP_XY_2_RC: {   // laser_models.py:55-86  (x, y, orig_x, orig_y, orig_c, orig_s, height, width, resolution) -> (r, c)
        const double xt = a[0] - a[2], yt = a[1] - a[3];
        const double xr = xt * a[4] + yt * a[5];
        const double yr = -xt * a[5] + yt * a[4];
        const double res = a[8];
        double r = -1., c = -1.;
        if (!(xr < 0 || xr >= a[7] * res || yr < 0 || yr >= a[6] * res)) {
            c = (double)(int)(xr / res);
            r = (double)(int)(yr / res);
        }
        o[0] = r;
        o[1] = c;
        break;
    }
    case F110_OP_DISTANCE_TRANSFORM: {   // :88-104 on the handle's map  (x, y) -> dt[r, c]
        int r, c;
        o[0] = sample_distance<LAYOUT_ROWMAJOR, false, false>(k, nullptr, a[0], a[1], r, c);
        break;
    }
    case F110_OP_TRACE_RAY: {   // :106-146 on the handle's map and (cos, sin) table  (x, y, theta_index) -> range
        const int ti = (int)a[2];   // int(theta_index) :124; like NumPy, a negative index counts from the table's end
        const double2 cs = k.cs[ti < 0 ? ti + k.theta_dis : ti];
        int r, c, nl;
        o[0] = march_ray<LAYOUT_ROWMAJOR, false, false>(k, nullptr, a[0], a[1], cs.x, cs.y, r, c, nl);
        break;
    }
    default:
        break;
    }
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

// PADDED layout: the table inside a border of `b` cells that read dt[-1,-1], what the reference
// returns for any out-of-bounds sample (laser_models.py:80-81,103)
__global__ void k_build_padded(const double *__restrict__ rowmajor, int H, int W, int b, int Wp, int Hp, double *__restrict__ pad)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)Wp * Hp) return;
    const int r = (int)(i / Wp) - b, c = (int)(i % Wp) - b;
    const bool inside = (r >= 0) & (r < H) & (c >= 0) & (c < W);
    pad[i] = inside ? rowmajor[(size_t)r * W + c] : rowmajor[(size_t)H * W - 1];
}


// A reactive policy that CONSUMES the scans where the scan kernel left them (round 5; not a reference function — the stand-in
// for an RL policy in a device-resident loop: examples/rl_loop_device.py, bench.py's "scans consumed on device" leg).  One wave
// per agent: the B beams are cut into 64 sectors, sector s = beams [ceil(s B / 64), ceil((s + 1) B / 64)); each lane takes the
// mean range of its sector (beams added in ascending order).  Steer towards the centre of the sector with the largest mean
// among those within `sector_limit` rad of straight ahead (first maximum), speed from the shortest range of the eight middle
// sectors.  actions [n][2] = (steer, speed).
__global__ void __launch_bounds__(256) k_scan_policy(const double *__restrict__ scans, int B, double fov, int i0, int n, double steer_gain, double steer_max,
                                                     double sector_limit, double v_lo, double v_hi, double d_ref, double *__restrict__ actions)
{
    // the agent's row goes through LDS: 64 consecutive ranges per load instruction (4 lines), then every lane walks its own
    // sector there (reading the sectors straight from HBM touches 64 lines per instruction, 16 x the traffic)
    extern __shared__ double lds_rows[];   // [4 waves][B]
    const int a = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (a >= n) return;
    const int lane = (int)(threadIdx.x & 63u);
    const double *row = scans + (size_t)(i0 + a) * (size_t)B;
    double *mine = lds_rows + (size_t)(threadIdx.x >> 6) * (size_t)B;
    for (int b = lane; b < B; b += 64) mine[b] = row[b];
    __builtin_amdgcn_wave_barrier();   // (one wave: its LDS writes are in order; the barrier only pins the compiler)
    const int b0 = (lane * B + 63) / 64, b1 = ((lane + 1) * B + 63) / 64;
    double sum = 0., lo = INFINITY;
    for (int b = b0; b < b1; ++b) {
        const double r = mine[b];
        sum += r;
        lo = r < lo ? r : lo;
    }
    const double inc = fov / (double)(B - 1);
    const double centre = -fov / 2. + inc * (0.5 * (double)(b0 + b1 - 1));
    double best = (b1 > b0 && fabs(centre) <= sector_limit) ? sum / (double)(b1 - b0) : -INFINITY;
    int best_lane = lane;
    double front = (lane >= 28 && lane < 36) ? lo : INFINITY;
    for (int off = 32; off; off >>= 1) {
        const double ob = __shfl_xor(best, off);
        const int ol = __shfl_xor(best_lane, off);
        if (ob > best || (ob == best && ol < best_lane)) {
            best = ob;
            best_lane = ol;
        }
        const double of = __shfl_xor(front, off);
        front = of < front ? of : front;
    }
    const double best_centre = __shfl(centre, best_lane);
    if (lane == 0) {
        // no eligible sector (none within sector_limit, or every eligible mean is NaN): straight ahead, not the tie-break's lane 0
        double steer = best > -INFINITY ? steer_gain * best_centre : 0.0;
        steer = steer > steer_max ? steer_max : (steer < -steer_max ? -steer_max : steer);
        const double f = front / d_ref;
        actions[2 * (size_t)(i0 + a)] = steer;
        actions[2 * (size_t)(i0 + a) + 1] = v_lo + (v_hi - v_lo) * (f < 1. ? f : 1.);
    }
}

// examples/waypoint_follow.py: PurePursuitPlanner.plan, 16 lanes per pose (4 poses per wave).  The
// lanes of a group split the segments of the waypoint polyline: the nearest-point search is a
// per-lane first-minimum followed by a (distance, index) lexicographic min across the group — the
// same winner as np.argmin's first minimum — and the look-ahead search tests 16 segments at a
// time and takes the lowest hit.  Per-segment arithmetic is f110_math.hpp's, unchanged.
// actions [n][2] = (steer, speed).
constexpr int kPlanLanes = 16;

__global__ void __launch_bounds__(256) k_pure_pursuit(const double *__restrict__ wp, int M, const double *__restrict__ px_,
                                                      const double *__restrict__ py_, const double *__restrict__ pth_, int stride, int n,
                                                      double lookahead, double vgain, double wheelbase, double max_reacquire,
                                                      double *__restrict__ actions)
{
    const int gid = (blockIdx.x * blockDim.x + threadIdx.x) / kPlanLanes;   // pose
    const int sub = threadIdx.x & (kPlanLanes - 1);
    const int grp_shift = (threadIdx.x & 63) & ~(kPlanLanes - 1);            // first lane of my group in the wave
    const bool live = gid < n;
    const int g = live ? gid : n - 1;   // idle groups shadow the last pose so the wave stays convergent
    const double px = px_[(size_t)g * stride], py = py_[(size_t)g * stride], theta = pth_[(size_t)g * stride];
    // ---- nearest_point_on_trajectory :15-50
    double dist = INFINITY, tb = 0.0;
    int best = 0x7fffffff;
    for (int k = sub; k + 1 < M; k += kPlanLanes) {
        const double ax = wp[3 * k], ay = wp[3 * k + 1];
        const double dx = wp[3 * k + 3] - ax, dy = wp[3 * k + 4] - ay;
        double t = ((px - ax) * dx + (py - ay) * dy) / (dx * dx + dy * dy);
        t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
        const double rx = px - (ax + t * dx), ry = py - (ay + t * dy);
        const double d = sqrt(rx * rx + ry * ry);
        if (d < dist) {
            dist = d;
            tb = t;
            best = k;
        }
    }
#pragma unroll
    for (int m = 1; m < kPlanLanes; m <<= 1) {
        const double od = __shfl_xor(dist, m, kPlanLanes), ot = __shfl_xor(tb, m, kPlanLanes);
        const int ob = __shfl_xor(best, m, kPlanLanes);
        if (od < dist || (od == dist && ob < best)) {
            dist = od;
            tb = ot;
            best = ob;
        }
    }
    if (best == 0x7fffffff) best = 0;   // all distances NaN: np.argmin -> 0
    // ---- _get_current_waypoint :183-201
    int goal = best;
    bool have = true;
    if (dist < lookahead) {
        // first_point_on_trajectory_intersecting_circle :52-132, wrap=True
        const double start = (double)best + tb;
        const int start_i = (int)start;
        const double start_t = fmod(start, 1.0);
        goal = -1;
        for (int base = start_i; base + 1 < M && goal < 0; base += kPlanLanes) {
            const int i = base + sub;
            const bool hit = (i + 1 < M) &&
                             lookahead_cuts(wp[3 * i], wp[3 * i + 1], wp[3 * i + 3], wp[3 * i + 4], px, py, lookahead, i == start_i ? start_t : 0.0);
            const unsigned m16 = (unsigned)((__ballot(hit) >> grp_shift) & 0xffffull);
            if (m16) goal = base + __ffs(m16) - 1;
        }
        for (int base = -1; base < start_i && goal < 0; base += kPlanLanes) {
            const int i = base + sub;
            bool hit = false;
            if (i < start_i) {
                const int k0 = i < 0 ? M - 1 : i, k1 = i + 1;
                hit = lookahead_cuts(wp[3 * k0], wp[3 * k0 + 1], wp[3 * k1], wp[3 * k1 + 1], px, py, lookahead, 0.0);
            }
            const unsigned m16 = (unsigned)((__ballot(hit) >> grp_shift) & 0xffffull);
            if (m16) {
                const int i0 = base + __ffs(m16) - 1;
                goal = i0 < 0 ? M - 1 : i0;
            }
        }
        have = goal >= 0;
    } else if (!(dist < max_reacquire)) {
        have = false;
    }
    if (sub != 0 || !live) return;
    // ---- get_actuation :134-145, plan :203-217
    double steer = 0.0, speed = 4.0;
    if (have) {
        const double wy = sin(-theta) * (wp[3 * goal] - px) + cos(-theta) * (wp[3 * goal + 1] - py);
        speed = vgain * wp[3 * best + 2];
        if (!(fabs(wy) < 1e-6)) {
            const double radius = 1 / (2.0 * wy / (lookahead * lookahead));
            steer = atan(wheelbase / radius);
        }
    }
    reinterpret_cast<double2 *>(actions)[gid] = make_double2(steer, speed);
}

__global__ void k_interleave_cs(const double *__restrict__ sines, const double *__restrict__ cosines, int n, double2 *__restrict__ cs)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) cs[i] = make_double2(cosines[i], sines[i]);
}

