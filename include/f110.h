/*
 * f110.h — C ABI of libf110_hip.so: the MI355X (gfx950) batched F1TENTH env.step() hot path.
 *
 * The reference (f1tenth/f1tenth_gym v0.2.1) has no FFI: its operator boundary is the set of
 * Python call sites where RaceCar/Simulator call the @njit kernels.  Each entry point below
 * names the reference interface it replaces (paths relative to gym/f110_gym/envs/).  The
 * Python host package f1tenth_gym_amd binds exactly these symbols with ctypes
 * (f1tenth_gym_amd/_ffi.py); INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; every call returns 0 (F110_OK) or a negative code and
 *     leaves a message retrievable with f110_last_error() (handle may be NULL for create).
 *   - "h_" pointers are host memory owned by the caller, consumed before the call returns
 *     (or before f110_sync for *_async variants); "d_" pointers are device memory.
 *   - one handle = one GPU + one HIP stream; calls on one handle must be serialised by the
 *     caller, different handles may be driven from different threads / processes.
 *   - agents are indexed i = env * num_agents + agent  (N = num_envs * num_agents);
 *     all arithmetic is IEEE float64 in the reference's operation order (no FMA contraction).
 *   - there is NO CPU fallback: without a usable HIP device every compute call fails.
 */
#ifndef F110_H
#define F110_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define F110_ABI_VERSION 1

enum {
    F110_OK = 0,
    F110_ERR_INVALID = -1,   /* bad argument (ValueError / IndexError on the Python side) */
    F110_ERR_NO_MAP = -2,    /* scan/step before a map is set (laser_models.py:445-446) */
    F110_ERR_HIP = -3,       /* HIP runtime / device error */
    F110_ERR_STATE = -4,     /* call not valid in the handle's current state */
    F110_ERR_NOMEM = -5
};

/* vehicle parameter vector: key order of f110_env.py:130 */
enum {
    F110_P_MU = 0, F110_P_CSF, F110_P_CSR, F110_P_LF, F110_P_LR, F110_P_H, F110_P_M, F110_P_I,
    F110_P_SMIN, F110_P_SMAX, F110_P_SVMIN, F110_P_SVMAX, F110_P_VSWITCH, F110_P_AMAX,
    F110_P_VMIN, F110_P_VMAX, F110_P_WIDTH, F110_P_LENGTH, F110_NPARAMS
};

enum { F110_INTEGRATOR_RK4 = 1, F110_INTEGRATOR_EULER = 2 }; /* base_classes.py:40-42 */

/* distance-table layouts in HBM.  Values 1 (4x4-cell tiles), 2 (1-byte codes + exact value LUT in LDS) and
 * 4 (a 128x128-cell window of byte codes per agent in LDS) existed in the experimental build through round 4 — bit-identical,
 * measured slower (DESIGN_HISTORY.md) — and were retired in round 5: f110_create refuses them (F110_ERR_INVALID). */
enum {
    F110_MAP_ROWMAJOR_F64 = 0, /* dt[r][c] as the reference stores it */
    F110_MAP_TILED_F64 = 1,    /* retired */
    F110_MAP_CODE8 = 2,        /* retired */
    F110_MAP_PADDED_F64 = 3,   /* dt[r][c] inside a border of out-of-bounds cells (max_range wide), so
                                  the march loop needs no range test, with fixed-point cell addressing
                                  and an exact re-march for samples in the guard band (the fastest
                                  layout; maps too large for it run as F110_MAP_ROWMAJOR_F64) */
    F110_MAP_WINDOW_LDS = 4    /* retired */
};

/* Simulator(params, num_agents, seed, time_step, ego_idx, integrator, lidar_dist)
 * base_classes.py:465 + RaceCar(num_beams=1080, fov=4.7) :69 + ScanSimulator2D(eps, theta_dis,
 * max_range) laser_models.py:360 + ttc_thresh base_classes.py:115.  */
typedef struct f110_config {
    int32_t abi_version;   /* F110_ABI_VERSION */
    int32_t num_envs;      /* E independent environments (extension; reference has 1) */
    int32_t num_agents;    /* A agents per environment */
    int32_t num_beams;     /* B */
    int32_t theta_dis;
    int32_t integrator;
    int32_t device_id;     /* HIP device ordinal */
    int32_t map_layout;    /* F110_MAP_* */
    int32_t scan_block;    /* threads per scan workgroup (0 = default) */
    int32_t scan_tasks_per_wave; /* consecutive 64-ray tasks each wave walks (0 = default) */
    int32_t step_groups;   /* env blocks per step: 0 = automatic (f110_step_device calls that come back to back are submitted as
                              two halves of the envs on two streams, at the batch sizes where that pays; anything else in one
                              block), 1 = always one block, 2 = always two, > 2 = experimental build.  Results do not depend on it. */
    int32_t step_graph;    /* must be 0 (the step as one captured HIP graph: measured slower, retired in round 5; the field keeps the struct layout) */
    double fov, eps, max_range;
    double time_step, lidar_dist, ttc_thresh;
    double params[F110_NPARAMS]; /* initial vehicle params for every agent slot */
} f110_config;

typedef struct f110_sim f110_sim;

const char *f110_last_error(const f110_sim *h);
int f110_abi_version(void);
int f110_device_count(int *count);
/* "dddd:bb:dd.f" of HIP device `device` (hipDeviceGetPCIBusId): lets the launcher pin each rank's host
 * threads to the NUMA node its GPU hangs off (f1tenth_gym_amd/numa.py); out: >= 16 bytes */
int f110_device_pci_bus_id(int32_t device, char *out, int32_t len);
/* "csrc=<sha256 prefix of the kernel sources this library was built from>": profiles/ entries and
 * bench.py's roofline record carry the same hash, so a reader can tie a number to the code */
const char *f110_build_info(void);

/* 1 for libf110_hip_exp.so (-DF110_EXPERIMENTAL), 0 for the product library */
int f110_is_experimental(void);
/* The switchboard of the experimental build — every variant that was measured against the default and not
 * adopted, for the A/B tests and the profiles.  The product library refuses every key
 * (F110_ERR_STATE) and reads no environment variable; a step of the product has ONE dispatch per (agents per
 * env, beams) case.  Keys: scan_flat, collide_mode (0 side stream | 1 fused into k_integrate | 2 in line | 3 inside k_finalize),
 * task_order, task_thr, task_cap_div (list capacity = tasks / div), task_rev (walk the list from its newest entry), long_prio,
 * scan_occupancy, scan_env_counter (fusion probes), integrate_duo (-1|0|1: k_integrate in one wave or two per 64 agents),
 * integrate_fan (-1|0|1), group_split, step_tiny (0: tiny batches through the three kernels too — the A/B of k_step_tiny), scan_trace_hi / scan_trace_lo (the two halves of the device address of a caller-owned
 * [launch waves][8] uint64 buffer that every wave of the step's scan kernel stamps with its begin / end clock, CU and samples:
 * tools/debug/scan_timeline.py; 0 = off).  Retired with the code they switched (numbers in DESIGN.md section 8 / DESIGN_HISTORY.md) — round 5:
 * dedupe_two_pass, no_window, finalize_lanes / _flat / _roles, pair_always, step_graph, ray_pass / ray_thr / ray_waves; round 6 (the
 * pre-registered stop rule for march variants): scan_stream / stream_refill / stream_block / stream_grid (the lane-refill scan),
 * spec_from (the speculative tail march), finalize_wave (the A = 2 finalize as one-wave workgroups), pad_tiled (the step's march on a
 * 4x4-tiled or a row-pair copy of the PADDED table), scan_nt (non-temporal range stores): F110_ERR_INVALID. */
int f110_exp_set(f110_sim *h, const char *key, int32_t value);

int f110_create(const f110_config *cfg, f110_sim **out);
void f110_destroy(f110_sim *h);
int f110_sync(f110_sim *h);

/* ---- ScanSimulator2D.set_map  laser_models.py:383-427 ----
 * image: h_img [height][width] uint8, top row first as PIL decodes it; the library does the
 * FLIP_TOP_BOTTOM (:399), the <=128 threshold (:403-404), the exact EDT (:425,:40-53) and
 * dt = resolution*sqrt(d2) on the device.  origin = yaml 'origin' (x, y, yaw). */
int f110_set_map_image(f110_sim *h, const uint8_t *h_img, int32_t height, int32_t width,
                       double resolution, double origin_x, double origin_y, double origin_yaw);
/* same, from a caller-supplied distance table (row 0 = bottom of the picture). */
int f110_set_map_dt(f110_sim *h, const double *h_dt, int32_t height, int32_t width,
                    double resolution, double origin_x, double origin_y, double origin_c,
                    double origin_s);
int f110_get_map_dt(f110_sim *h, double *h_dt_out); /* row-major [height][width] */
int f110_map_shape(f110_sim *h, int32_t *height, int32_t *width);

/* sines/cosines = sin/cos(linspace(0, 2pi, theta_dis))  laser_models.py:379-381 (host-computed
 * so they are bit-identical to NumPy's). */
int f110_set_trig_tables(f110_sim *h, const double *h_sines, const double *h_cosines, int32_t n);
/* RaceCar class-level per-beam tables, base_classes.py:125-158 */
int f110_set_beam_tables(f110_sim *h, const double *h_scan_angles, const double *h_cosines,
                         const double *h_side_distances, int32_t num_beams);
/* Extension (domain randomisation): a different track per env.  The map of
 * f110_set_map_* is slot 0; f110_add_map_* registers further maps (same arguments, same exact-EDT
 * pipeline) and returns their slot; f110_set_env_maps assigns a slot to every env (h_env_map
 * [num_envs]; NULL = everybody back on slot 0).  Needs map_layout = F110_MAP_PADDED_F64 with every
 * map fitting it (any beam count).  Changing slot 0 or adding maps takes
 * effect at the next f110_set_env_maps.  Unit entry points (f110_scan_batch ...) keep using slot 0. */
int f110_add_map_image(f110_sim *h, const uint8_t *h_img, int32_t height, int32_t width,
                       double resolution, double origin_x, double origin_y, double origin_yaw,
                       int32_t *slot);
int f110_add_map_dt(f110_sim *h, const double *h_dt, int32_t height, int32_t width,
                    double resolution, double origin_x, double origin_y, double origin_c,
                    double origin_s, int32_t *slot);
int f110_set_env_maps(f110_sim *h, const int32_t *h_env_map);

/* Simulator.update_params base_classes.py:514-534 (agent_idx<0: all slots) */
int f110_set_params(f110_sim *h, int32_t agent_idx, const double *h_params18);
/* Extension (domain randomisation over vehicle dynamics): one parameter set per AGENT, h_params
 * [N][18] in the order of f110_config.params; NULL returns to the per-slot sets of f110_set_params
 * (which is refused while this is active).  As in the reference, the collision boxes keep the
 * constructor's length/width (base_classes.py:549) and the iTTC tables those of the first car. */
int f110_set_params_batch(f110_sim *h, const double *h_params);
/* scan noise, laser_models.py:450-452 with base_classes.py:204: row k is added to every
 * agent's scan on its k-th step after reset (rows wrap modulo n_rows).  n_rows=0: no noise. */
int f110_set_noise_table(f110_sim *h, const double *h_noise, int32_t n_rows, int32_t num_beams);
/* The same noise generated ON THE DEVICE: np.random.default_rng(seed).normal(0.,
 * std_dev, num_beams) per scan, laser_models.py:450-452 — NumPy's PCG64 stream through NumPy's
 * ziggurat, bit for bit — re-started for an agent whenever it is reset (base_classes.py:204).
 * Nothing is uploaded and memory stays flat however long the run is.
 *   h_state_inc, per_agent = 0: [4] = {state.hi, state.lo, inc.hi, inc.lo} of np.random.PCG64(seed)
 *       (f110_pcg64_seed computes them): one stream shared by every agent, as in the reference.  The
 *       first cache_rows rows (0: 16384) are generated once into a device row cache, extended on
 *       demand as episodes get longer; an agent whose episode outlives the cache continues from the
 *       stream position it carries.
 *   per_agent = 1: [N][4], a stream per agent (extension), always generated from the carried state.
 *   NULL: noise off.  Replaces any table of f110_set_noise_table, and vice versa. */
int f110_set_noise_rng(f110_sim *h, const uint64_t *h_state_inc, int32_t per_agent, double std_dev,
                       int32_t cache_rows);
/* np.random.PCG64(seed): SeedSequence(seed).generate_state(4, uint64) + pcg64_set_seed, host only;
 * seed < 2^64.  out4 = {state.hi, state.lo, inc.hi, inc.lo}. */
int f110_pcg64_seed(uint64_t seed, uint64_t *out4);
/* generate the shared stream's rows [0, rows) into the row cache now (instead of on demand) */
int f110_noise_prepare(f110_sim *h, int32_t rows);

/* Simulator.reset base_classes.py:614-630 / RaceCar.reset :183-204.
 * h_poses [N][3]; h_env_mask [num_envs] or NULL (all). */
int f110_reset(f110_sim *h, const double *h_poses, const uint8_t *h_env_mask);
int f110_reset_device(f110_sim *h, const double *d_poses, const uint8_t *d_env_mask);
/* in-place re-seat on the device, no host round trip: every env whose agent `ego_idx` has
 * collisions != 0 (the `done` condition of f110_env.py:244) is reset to d_start_poses [N][3];
 * *d_count (device int32, may be NULL) is incremented once per env reset. */
int f110_reset_collided_device(f110_sim *h, const double *d_start_poses, int32_t ego_idx,
                               int32_t *d_count);
/* The same re-seat folded into the end of every following f110_step / f110_step_device (one kernel
 * boundary less per step), until called again with d_start_poses = NULL.  State, delay buffer and
 * step counters end up exactly as after step + f110_reset_collided_device; in_collision keeps the
 * step's value (the separate call clears it). */
int f110_set_auto_reseat(f110_sim *h, const double *d_start_poses, int32_t ego_idx, int32_t *d_count);

/* ---- F110Env episode logic on the device: _check_done f110_env.py:204-246 (start/finish-zone
 * toggles, lap counts / times, done = ego collided or all agents have 4 toggles) and the state part
 * of reset() :319-334, so GPU-resident RL loops never read poses back to decide `done`.
 * f110_episode_reset also performs f110_reset for the masked envs.  h_rot [num_envs][4] is
 * start_rot (:331) row-major, computed by the host like the reference does. */
typedef struct f110_episode_host {
    double *lap_times, *lap_counts, *toggles; /* [N] */
    double *current_time;                     /* [num_envs] */
    uint8_t *near_starts;                     /* [N] */
    uint8_t *done;                            /* [num_envs] */
    uint8_t *checkpoint_done;                 /* [N] toggle_list >= 4 */
} f110_episode_host;
typedef struct f110_episode_views {
    uint8_t *done, *checkpoint_done;
    double *lap_times, *lap_counts, *toggles, *current_time;
} f110_episode_views;
int f110_episode_init(f110_sim *h, int32_t ego_idx);
int f110_episode_reset(f110_sim *h, const double *h_poses, const double *h_rot,
                       const uint8_t *h_env_mask);
/* (These two, like f110_pure_pursuit_device, keep a device-resident loop in its env blocks: behind a step that went out as two
 * env blocks — f110_config.step_groups — they run per block on the block's stream; an env's bookkeeping reads that env only.) */
int f110_episode_step_device(f110_sim *h, const double *d_actions); /* f110_step_device + _check_done */
int f110_episode_reset_done_device(f110_sim *h, int32_t *d_count);  /* re-seat envs whose done flag is set */
int f110_episode_get(f110_sim *h, const f110_episode_host *out);
/* One host round trip per step for host-side RL loops (F110VecEnv): h_actions [N][2] up, the step,
 * _check_done, every per-agent / per-env scalar of the observation down in ONE copy, then (auto_reset)
 * the in-place re-seat of the envs whose done flag is set.  h_packed (f110_episode_packed_bytes(h)
 * bytes; pinned memory from f110_host_alloc for a full-rate copy) receives
 *   double [9][N]  poses_x, poses_y, poses_theta, linear_vels_x, ang_vels_z, collisions,
 *                  lap_times, lap_counts, toggles
 *   double [E]     current_time
 *   uint8  [N] near_starts, [N] checkpoint_done, [E] done
 * — the observation of the step just taken (before any re-seat).  Scans stay in HBM. */
int f110_episode_step_host(f110_sim *h, const double *h_actions, int32_t auto_reset, void *h_packed);
size_t f110_episode_packed_bytes(const f110_sim *h);
int f110_host_alloc(f110_sim *h, size_t bytes, void **h_out);   /* page-locked host memory */
/* h may be NULL once the handle that allocated it is destroyed.  Must NOT run concurrently with a step (or any other call) on a
 * handle that uses the block: the call drains every live handle's stream and drops their cached views of the block without
 * taking a per-handle lock — the caller serialises it against those handles like any other call on them. */
int f110_host_free(f110_sim *h, void *h_ptr);
int f110_episode_device_views(f110_sim *h, f110_episode_views *out);

/* env.step() of a host-driven loop as ONE call (replaces, per step, F110Env.step -> Simulator.step's agent
 * loops + observation dict base_classes.py:553-612 and, once f110_episode_init was called, _check_done
 * f110_env.py:204-246 and the re-seat of reset() :319-334): actions up, the step, the episode logic, and the
 * requested observation columns written by one kernel straight into the caller's PAGE-LOCKED memory
 * (f110_host_alloc — required: the kernel stores into it in place; anything else is refused with
 * F110_ERR_INVALID), scans by one DMA copy behind it.  Any pointer may be NULL = not wanted.  The episode
 * columns and F110_STEP_AUTO_RESET need f110_episode_init (else F110_ERR_STATE).  The block holds the
 * observation of the step just taken; with F110_STEP_AUTO_RESET finished envs (done != 0) are re-seated
 * at their start poses AFTER it was written, and the device-side done flag is cleared. */
typedef struct f110_host_block {
    double *state;            /* [7][N] columns x, y, steer, v, yaw, yaw_rate, slip (poses_x = row 0, ...) */
    double *collisions;       /* [N] */
    double *collision_idx;    /* [N] */
    double *agent_poses;      /* [3][N] Simulator.agent_poses (:574 snapshot) */
    double *lap_times;        /* [N] */
    double *lap_counts;       /* [N] */
    double *toggles;          /* [N] */
    double *current_time;     /* [E] */
    int32_t *in_collision;    /* [N] */
    uint8_t *near_starts;     /* [N] */
    uint8_t *checkpoint_done; /* [N] */
    uint8_t *done;            /* [E] */
    double *scans;            /* [N][B] (any host memory; page-locked for a full-rate copy) */
} f110_host_block;
#define F110_STEP_AUTO_RESET 1      /* re-seat finished envs inside the call */
#define F110_STEP_NO_SYNC 2         /* return once everything is enqueued: f110_sync(h) completes the block
                                       (gym.vector's step_async / step_wait split) */
#define F110_STEP_ACTIONS_MAPPED 4  /* h_actions is f110_host_alloc memory: read in place, no staging copy */
#define F110_STEP_SPIN_WAIT 8       /* wait by polling a page-locked completion word the last workgroup stores,
                                       instead of a runtime synchronise (ignored with scans / NO_SYNC) */
#define F110_STEP_POLL 32           /* wait by polling hipStreamQuery (a busy core, ~1 us wake-up) instead of
                                       hipStreamSynchronize (coarse wake-up quanta beyond ~30 us of waiting); the poll is
                                       bounded: after 250 us the call sleeps in hipStreamSynchronize */
#define F110_STEP_NO_FUSE 16        /* A/B: always run the episode logic + host block as a kernel of their own (with 2
                                       agents per env they are otherwise the finalize kernel's epilogue) */
int f110_step_host(f110_sim *h, const double *h_actions /* [N][2] */, const f110_host_block *out, int32_t flags);
/* measurement aid: {calls, host microseconds spent enqueuing, host microseconds spent waiting} of the
 * f110_step_host calls since the last read (cleared by the read) */
int f110_step_host_stats(f110_sim *h, double *out3);

/* Simulator.step base_classes.py:553-612.  actions [N][2] = (steer, speed).
 * Asynchronous on the handle's stream; outputs are read with f110_get_* (which sync). */
int f110_step(f110_sim *h, const double *h_actions);
int f110_step_device(f110_sim *h, const double *d_actions);

/* observation / state read-back (device -> host).  Any pointer may be NULL. */
typedef struct f110_obs_host {
    double *scans;          /* [N][B]  obs['scans'] */
    double *poses_x;        /* [N] */
    double *poses_y;        /* [N] */
    double *poses_theta;    /* [N] */
    double *linear_vels_x;  /* [N] */
    double *ang_vels_z;     /* [N] */
    double *collisions;     /* [N]  GJK flag OR wall flag (base_classes.py:588-589) */
    double *collision_idx;  /* [N]  collision_models.py:184-212 */
    double *state;          /* [N][7] RaceCar.state */
    double *agent_poses;    /* [N][3] Simulator.agent_poses (:574 snapshot) */
    int32_t *in_collision;  /* [N]  RaceCar.in_collision */
    int32_t *step_count;    /* [N]  steps since reset */
} f110_obs_host;
int f110_get_obs(f110_sim *h, const f110_obs_host *out);
int f110_set_state(f110_sim *h, const double *h_state7 /* [N][7] */,
                   const double *h_steer_buf /* [N][2] newest first, or NULL */,
                   const int32_t *h_buf_count /* [N] or NULL */);

/* device-resident observation buffers (valid until f110_destroy; contents valid after the
 * step that produced them completes on the stream).  SoA columns of N doubles. */
typedef struct f110_device_views {
    double *scans;        /* [N][B] */
    double *state;        /* [7][N] columns x, y, steer, v, yaw, yaw_rate, slip */
    double *agent_poses;  /* [3][N] */
    double *collisions;   /* [N] */
    double *collision_idx;
    int32_t *in_collision;
    int32_t *step_count;
    void *stream;         /* hipStream_t */
} f110_device_views;
int f110_get_device_views(f110_sim *h, f110_device_views *out);
/* Hand f110_device_views.stream to EXTERNAL work (a torch / cupy stream wrapped around it, a user kernel) safely: everything the
 * handle has in flight — including the second env block of a two-block step, which runs on a stream of its own — is ordered in
 * front of what the caller enqueues on that stream next, and the following f110_*step* is submitted as ONE block on that stream,
 * i.e. behind the caller's work (its reads of the observation, its writes of the action buffer).  Call it once per iteration
 * between the handle's last call and the external work; it enqueues two event waits and never blocks the host.  Without it only
 * one-block steps (f110_config.step_groups = 1, or any handle that synchronises every step) are ordered against that stream. */
int f110_stream_fence(f110_sim *h);
/* free / total bytes of the handle's GPU (hipMemGetInfo) — lets a long run show that memory stays flat */
int f110_device_mem_info(f110_sim *h, size_t *free_bytes, size_t *total_bytes);
int f110_device_alloc(f110_sim *h, size_t bytes, void **d_out);
int f110_device_free(f110_sim *h, void *d_ptr);
int f110_memcpy_h2d(f110_sim *h, void *d_dst, const void *h_src, size_t bytes);
int f110_memcpy_d2h(f110_sim *h, void *h_dst, const void *d_src, size_t bytes);

/* ---- optional observation gather over RCCL / xGMI (BASELINE config 4; off the step path) ----
 * Environments never interact, so stepping needs no collective.  A consumer that wants every
 * rank's scans on every GPU creates one communicator (rank 0 makes the id, the launcher's control
 * plane broadcasts its 128 bytes) and calls f110_comm_all_gather_scans after a step: an
 * ncclAllGather of [N][B] float64 per rank, enqueued on the handle's stream.
 * d_recv: device buffer of n_ranks*N*B doubles. */
#define F110_COMM_ID_BYTES 128
int f110_comm_unique_id(void *out_id128);
int f110_comm_init(f110_sim *h, int32_t n_ranks, int32_t rank, const void *id128);
int f110_comm_all_gather_scans(f110_sim *h, void *d_recv);
/* The whole observation of Simulator.step (base_classes.py:594-610; SURVEY 8e: "scans [N_g,B] + 7
 * scalars/agent"): the scans as above AND the per-agent scalars, packed by a kernel behind the step as
 * double [F110_OBS_SCALARS][N] = poses_x, poses_y, poses_theta, linear_vels_x, linear_vels_y (always
 * 0., :603), ang_vels_z, collisions — two ncclAllGather calls inside ONE ncclGroupStart / End.
 * d_recv_scans: n_ranks*N*B doubles; d_recv_scalars: n_ranks*F110_OBS_SCALARS*N doubles (rank-major).
 * Honours f110_comm_set_overlap exactly like f110_comm_all_gather_scans (the scalar block is
 * double-buffered with the scans). */
#define F110_OBS_SCALARS 7
int f110_comm_all_gather_obs(f110_sim *h, void *d_recv_scans, void *d_recv_scalars);
/* The same gather with the two knobs SURVEY 8e prices: transport F110_GATHER_F32 sends the scans as float32 (a
 * conversion kernel in front of the collective; d_recv_scans then holds n_ranks*N*B FLOATS; the scalars stay
 * float64) — 142 MB instead of 283 MB per rank and step at 32 768 agents; root >= 0 gathers to that rank only
 * (grouped ncclSend / ncclRecv: every peer's block rides its one direct link to the root, the other ranks
 * receive nothing; their d_recv_scans may be NULL, and they pass a non-NULL d_recv_scalars — never written —
 * exactly when the root wants the scalar blocks), root = -1 is the all-gather above.  Honours
 * f110_comm_set_overlap. */
#define F110_GATHER_F64 0
#define F110_GATHER_F32 1
int f110_comm_gather_obs(f110_sim *h, void *d_recv_scans, void *d_recv_scalars, int32_t transport, int32_t root);
/* how the step is submitted (f110_config.step_groups): *groups = env blocks the handle can submit a step as (2 = two halves of
 * the envs on two streams that were OBSERVED to run next to each other when the handle was created; 1 = one block),
 * *probes = candidate streams that observation tried, *last = blocks the most recent f110_step_device was submitted as.
 * Any pointer may be NULL.  Bookkeeping for benchmarks and tests; no reference counterpart. */
int f110_step_groups(f110_sim *h, int32_t *groups, int32_t *probes, int32_t *last);
/* *launches = 1 when the most recent step ran as ONE kernel launch (round 6: a waiting f110_step_host of at most 4 agents with one or two
 * cars per env — the reference's own shape, F110Env(num_agents = 1 | 2) on one env — integrate, scan, finalize, the observation block and
 * the completion word in a single launch, k_step_tiny; results are the same bits), 0 = the per-kernel form.  Bookkeeping for
 * benchmarks and tests; no reference counterpart. */
int f110_step_launches(f110_sim *h, int32_t *launches);
/* size and rank of the communicator as RCCL itself reports them (ncclCommCount / ncclCommUserRank) */
int f110_comm_info(f110_sim *h, int32_t *n_ranks, int32_t *rank);
/* enable = 1: the gather OVERLAPS the following step.  The scans are double-buffered (a second
 * [N][B] buffer): f110_comm_all_gather_scans then runs on a stream of its own behind the step that
 * produced the current buffer, the next f110_step* fills the other buffer, and the step after that
 * waits for the gather before reusing the first.  While enabled, f110_device_views.scans alternates
 * between the two buffers (fetch it after each step); d_recv must not be reused by the caller before
 * the data has been consumed (alternate two receive buffers).  Any other call on the handle
 * (f110_sync, f110_get_obs, f110_memcpy_d2h, ...) first makes the main stream wait for outstanding
 * gathers. */
int f110_comm_set_overlap(f110_sim *h, int32_t enable);
int f110_comm_destroy(f110_sim *h);

/* HIP-event timing on the handle's stream (bench.py roofline leg).
 * f110_timer_begin/_end bracket a region; f110_profile_kernels(1) additionally brackets
 * every scan-kernel launch inside f110_step with its own event pair. */
int f110_timer_begin(f110_sim *h);
int f110_timer_end_ms(f110_sim *h, double *ms);
int f110_profile_kernels(f110_sim *h, int32_t enable);
int f110_profile_read(f110_sim *h, int32_t *n_launches, double *scan_ms_total,
                      double *dyn_ms_total, double *finalize_ms_total);

/* ---- unit entry points (one per reference kernel; used by the parity tests) ---- */
/* ScanSimulator2D.scan(pose, None)  laser_models.py:429-454 -> get_scan :148-186.
 * h_ranges [M][B]; h_hit_rc [M][B][2] (r,c) of the terminating sample or NULL;
 * h_lookups [M] table lookups per pose or NULL. */
int f110_scan_batch(f110_sim *h, const double *h_poses, int32_t m, double *h_ranges,
                    int32_t *h_hit_rc, int64_t *h_lookups);
/* examples/waypoint_follow.py:15-217 — PurePursuitPlanner.plan (nearest point on the waypoint
 * polyline, first look-ahead-circle cut with wrap-around, get_actuation), the reference's example
 * policy, so that a closed loop can stay on the GPU.  waypoints [M][3] = (x, y, speed) as the
 * planner reads them through conf.wpt_xind / wpt_yind / wpt_vind; actions [.][2] = (steer, speed),
 * the layout f110_step takes.  _batch: host poses [m][3]; _device: the live poses of all N agents
 * (observation poses_x / poses_y / poses_theta), device pointers, asynchronous on the handle's stream. */
int f110_pure_pursuit_batch(f110_sim *h, const double *h_waypoints, int32_t M, const double *h_poses,
                            int32_t m, double lookahead, double vgain, double wheelbase,
                            double max_reacquire, double *h_actions);
int f110_pure_pursuit_device(f110_sim *h, const double *d_waypoints, int32_t M, double lookahead,
                             double vgain, double wheelbase, double max_reacquire, double *d_actions);
/* A reactive policy that CONSUMES the step's scans on the device (NOT a reference function: the stand-in for an RL policy in
 * a device-resident loop — examples/rl_loop_device.py, bench.py's "scans consumed on device" leg).  Per agent: the num_beams
 * ranges in 64 sectors, steer = clamp(steer_gain * centre angle of the sector with the largest mean range among those within
 * sector_limit rad of straight ahead, +-steer_max), speed = v_lo + (v_hi - v_lo) * min(1, shortest range of the eight middle
 * sectors / d_ref).  Reads the observation of the step just taken (f110_get_device_views().scans), writes d_actions [N][2] =
 * (steer, speed) — the layout f110_step_device takes; asynchronous on the handle's stream (per env block behind a two-block step). */
int f110_scan_policy_device(f110_sim *h, double steer_gain, double steer_max, double sector_limit, double v_lo,
                            double v_hi, double d_ref, double *d_actions);
/* Diagnostics of the scan kernels (step and unit form): with enable = 1 every marched ray is counted
 * as {fixed-point march on the padded table, re-marched exactly after a guard-band sample, exact
 * because the lidar is off the padded table / the layout has no fast path}.  out3 (or NULL) receives
 * and clears the counters; enable = -1 leaves the switch as it is.  Off by default (atomics). */
int f110_scan_path_stats(f110_sim *h, int32_t enable, int64_t *out3);
/* vehicle_dynamics_st / vehicle_dynamics_ks  dynamic_models.py:90-176; x [M][7], u [M][2] */
int f110_dynamics_batch(f110_sim *h, const double *h_x, const double *h_u,
                        const double *h_params18, int32_t m, double *h_f_st, double *h_f_ks);
/* pid  dynamic_models.py:178-221; in [M][4] = (speed, steer, current_speed, current_steer) */
int f110_pid_batch(f110_sim *h, const double *h_in, const double *h_params18, int32_t m,
                   double *h_accl_sv /* [M][2] */);
/* RaceCar.update_pose  base_classes.py:256-409 (without the scan) */
int f110_update_pose_batch(f110_sim *h, const double *h_state0, const double *h_buf0,
                           const int32_t *h_cnt0, const double *h_actions,
                           const double *h_params18, double time_step, int32_t integrator,
                           double lidar_dist, int32_t m, double *h_state1, double *h_buf1,
                           int32_t *h_cnt1, double *h_scan_pose);
/* get_vertices collision_models.py:237-260; poses [M][3] -> [M][4][2] */
int f110_get_vertices_batch(f110_sim *h, const double *h_poses, double length, double width,
                            int32_t m, double *h_vertices);
/* collision (GJK) collision_models.py:113-182 on M pairs of [4][2] */
int f110_gjk_batch(f110_sim *h, const double *h_va, const double *h_vb, int32_t m,
                   int32_t *h_flags);
/* collision_multiple :184-212 on G groups of n bodies [G][n][4][2] */
int f110_collision_multiple_batch(f110_sim *h, const double *h_vertices, int32_t groups,
                                  int32_t n, double *h_collisions, double *h_collision_idx);
/* check_ttc_jit laser_models.py:188-217 on M scans [M][B] with the handle's beam tables */
int f110_ttc_batch(f110_sim *h, const double *h_scans, const double *h_vels, int32_t m,
                   double ttc_thresh, int32_t *h_flags);
/* ray_cast laser_models.py:318-346 (+ get_blocked_view_indices :282-315) on M cases:
 * ego pose [M][3], opponent vertices [M][4][2], scans in/out [M][B], window [M][2] or NULL */
int f110_raycast_batch(f110_sim *h, const double *h_ego, const double *h_vertices, int32_t m,
                       double *h_scans_inout, int32_t *h_min_max_ind);
/* get_range laser_models.py:249-280; in [M][8] = (pose3, beam_theta, va2, vb2) */
int f110_get_range_batch(f110_sim *h, const double *h_in, int32_t m, double *h_out);
/* exact squared EDT of a binary image (nonzero = free), laser_models.py:40-53 */
int f110_edt_sq(f110_sim *h, const uint8_t *h_img, int32_t height, int32_t width,
                uint32_t *h_d2);
/* get_dt laser_models.py:40-53: dt = resolution * scipy.ndimage.distance_transform_edt(bitmap) — h_bitmap [height][width]
 * uint8, nonzero = free space (what edt treats as foreground), as the array is (no flip, no threshold); exact EDT on the
 * device, h_dt [height][width] float64. */
int f110_dt_from_bitmap(f110_sim *h, const uint8_t *h_bitmap, int32_t height, int32_t width, double resolution, double *h_dt);
/* The remaining small functions that `from f110_gym.envs import *` exposes in the reference (envs/__init__.py:2-5), M items per
 * call, one thread each, in the reference's expression order.  h_in [M][in_width(op, n)], h_out [M][out_width(op)]; n = vertices
 * per body for the ops that take bodies (the reference calls them with 4), ignored otherwise.
 *   op                          reference                        in (doubles per item)                                   out
 *   F110_OP_ACCL_CONSTRAINTS    dynamic_models.py:29-60          vel, accl, v_switch, a_max, v_min, v_max                  accl
 *   F110_OP_STEERING_CONSTRAINT dynamic_models.py:62-87          steering_angle, steering_velocity, s_min, s_max, sv_min, sv_max   steering_velocity
 *   F110_OP_CROSS               laser_models.py:219-230          v1[2], v2[2]                                              cross product
 *   F110_OP_ARE_COLLINEAR       laser_models.py:232-247          pt_a[2], pt_b[2], pt_c[2]                                 0. / 1.
 *   F110_OP_PERPENDICULAR       collision_models.py:34-48        pt[2]                                                     [2]
 *   F110_OP_TRIPLE_PRODUCT      collision_models.py:51-64        a[2], b[2], c[2]                                          [2]
 *   F110_OP_AVG_POINT           collision_models.py:67-78        vertices[n][2]                                            [2]
 *   F110_OP_FURTHEST_POINT      collision_models.py:81-92        vertices[n][2], d[2]                                      index (as a double)
 *   F110_OP_SUPPORT             collision_models.py:95-110       vertices1[n][2], vertices2[n][2], d[2]                    [2]
 *   F110_OP_GET_TRMTX           collision_models.py:218-235      pose[3]                                                   H[4][4] row-major
 *   F110_OP_XY_2_RC             laser_models.py:55-86            x, y, orig_x, orig_y, orig_c, orig_s, height, width, resolution   r, c (as doubles; -1, -1 out of bounds)
 *   F110_OP_DISTANCE_TRANSFORM  laser_models.py:88-104           x, y  (the handle's map: f110_set_map_*)                  dt[r, c] (dt[-1, -1] out of bounds)
 *   F110_OP_TRACE_RAY           laser_models.py:106-146          x, y, theta_index  (the handle's map, trig tables, eps, max_range)   range
 * (get_scan = f110_scan_batch, get_range = f110_get_range_batch, get_blocked_view_indices = the window of f110_raycast_batch,
 * get_dt = f110_dt_from_bitmap, the rest of the star-exports have had entry points since round 1.) */
enum {
    F110_OP_ACCL_CONSTRAINTS = 1, F110_OP_STEERING_CONSTRAINT, F110_OP_CROSS, F110_OP_ARE_COLLINEAR, F110_OP_PERPENDICULAR,
    F110_OP_TRIPLE_PRODUCT, F110_OP_AVG_POINT, F110_OP_FURTHEST_POINT, F110_OP_SUPPORT, F110_OP_GET_TRMTX, F110_OP_XY_2_RC,
    F110_OP_DISTANCE_TRANSFORM, F110_OP_TRACE_RAY
};
int f110_helper_batch(f110_sim *h, int32_t op, const double *h_in, int32_t m, int32_t n, double *h_out);
/* rng.normal(0., std_dev, num_beams) drawn `rows` times in a row from the PCG64 state h_state_inc4
 * (numpy/random/src/distributions/distributions.c random_standard_normal; laser_models.py:450-452):
 * h_out [rows][num_beams]; h_state_out2 (or NULL) = {state.hi, state.lo} after the last draw */
int f110_noise_rows_batch(f110_sim *h, const uint64_t *h_state_inc4, double std_dev, int32_t rows,
                          int32_t num_beams, double *h_out, uint64_t *h_state_out2);
/* Measurement aid (bench.py's L-bar): with enable = 1 the step's scan kernels sum the table lookups
 * of every ray they march (the reference's dependent gathers, laser_models.py:129-143).  out2 (or
 * NULL) receives and clears {the sum, 0 (round 1-4: how many of them the retired LDS-window layout served)};
 * enable = -1 leaves the switch as it is.  Off by default. */
int f110_scan_lookup_count(f110_sim *h, int32_t enable, int64_t *out2);
/* table index int(theta_index) of every beam for M headings (get_scan :167-184) */
int f110_beam_dir_index_batch(f110_sim *h, const double *h_thetas, int32_t m, int32_t *h_idx);

#ifdef __cplusplus
}
#endif
#endif /* F110_H */
