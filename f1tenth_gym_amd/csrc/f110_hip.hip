// f110_hip.hip — host side of libf110_hip.so: the handle (one MI355X, one HIP stream + a side
// stream), device-memory ownership, kernel launches and the C ABI declared in include/f110.h.
// The kernels live in f110_kernels.hpp, the scalar float64 building blocks in f110_math.hpp.
// A step is k_integrate -> k_scan_rays_agent -> k_finalize_pair on one stream for two-agent envs,
// k_integrate -> { k_scan_rays || k_collide (side stream) } -> k_finalize otherwise.
// Compiled with -ffp-contract=off.  There is no CPU fallback in this library.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <set>
#include <string>
#include <vector>

#include "../../include/f110.h"
#include "f110_kernels.hpp"


// ============================================================================ host side

// Two builds of this file: libf110_hip.so (the product: ONE step dispatch per (agents per env, beams) case,
// the row-major and PADDED table layouts, no environment variables read anywhere) and, with
// -DF110_EXPERIMENTAL, libf110_hip_exp.so, which adds everything that was built, measured and not adopted
// (tiled / byte-code / LDS-window layouts, env groups, the HIP-graph step, other places
// for the pair tests, two-pass dedupe) plus f110_exp_set, the switchboard the A/B tests and profiles use.
#ifdef F110_EXPERIMENTAL
constexpr bool kExperimental = true;
#else
constexpr bool kExperimental = false;
#endif

// f110_exp_set keys (experimental build; the product keeps the defaults)
struct ExpSwitches {
    int scan_flat = 0;         // 1: the flat ray kernel instead of the agent-aligned one
    int long_prio = 0;         // 1: the longest-first pass's waves raise their issue priority (s_setprio 3)
    int scan_occupancy = 0;    // 4: run the step's scan kernel at 4 waves/SIMD (fusion feasibility A/B)
    int scan_env_counter = 0;  // 1: the scan kernel also counts finished tasks per env (fusion feasibility A/B)
    int integrate_duo = -1;    // 0: k_integrate<0> (one wave per 64 agents), 1: k_integrate_duo, -1 = default (duo)
    int group_split = 0;       // two env groups: percent of the envs in the first (0 = even)
    int integrate_fan = -1;    // 0 / 1: k_integrate_fan (thirteen waves per 64 agents, RK4) off / on at every size, -1 = default (small batches)
    int tiny_general_tail = 0;
    int tiny_start_probe = 0;  // k_step_tiny tells the host when its first workgroup starts (f110_step_host times launch -> start -> done)
    long probe_calls = 0;
    double probe_us[3] = {0, 0, 0};
    uint64_t tiny_trace = 0;   // device address of a caller-owned [workgroups][16] uint64 buffer: k_step_tiny's phase stamps (0 = off)
    uint64_t scan_trace = 0;   // device address of a caller-owned [waves][8] uint64 buffer: clock stamps, hardware id, samples of every scan wave (0 = off)
};

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
    bool scan_tasks_auto = false;                // chosen by batch size (f110_config.scan_tasks_per_wave = 0)
    double ttc_side_max = INFINITY, ttc_cos_max = INFINITY;  // see f110_set_beam_tables
    int dir_stride = 0;              // > 0: dedupe enabled
    int scan_block = 64;
    double *d_params_all = nullptr;   // [N][18] when f110_set_params_batch is active
    double *d_params = nullptr, *d_noise = nullptr, *d_scan_angles = nullptr, *d_beam_cos = nullptr, *d_side = nullptr;
    // a different track per env: slot 0 is the map of f110_set_map_*, further slots come from
    // f110_add_map_*; f110_set_env_maps assigns them and builds the device tables
    struct MapSlot {
        double *d_dt_row = nullptr, *d_dt_pad = nullptr;
        ScanConst k{};
    };
    std::vector<MapSlot> extra_maps;
    MapFast *d_maps_fast = nullptr;
    ScanConst *d_maps_full = nullptr;
    int32_t *d_env_map = nullptr;
    uint32_t *d_scan_order = nullptr;   // [N] agents sorted by map slot (nullptr while every env is on one slot)
    bool multi_map = false;
    ScanConst *d_k = nullptr;  // HBM copy of k (RayJob::k_cold), refreshed by cold_consts()
    ScanConst k_uploaded{};
    unsigned long long *d_path_stats = nullptr;  // [3], see f110_scan_path_stats
    bool path_stats_on = false;
    double *d_dt_row = nullptr, *d_dt_pad = nullptr, *d_actions = nullptr, *d_poses = nullptr;
    double2 *d_cs = nullptr;
    uint8_t *d_mask = nullptr;
    // env groups: the step of G > 1 independent env blocks runs on G streams of its own (no event
    // between the kernels of a group, no dependency between groups; see f110_step_device)
    int groups = 1;
    std::vector<hipStream_t> gstreams;   // two groups: the main stream and the side stream (borrowed); more: streams of their own
    std::vector<hipStream_t> gowned;
    std::vector<hipEvent_t> gevents;
    hipEvent_t ev_main = nullptr;
    bool groups_auto = true;    // cfg.step_groups == 0: two blocks only for steps that come back to back, at the sizes where it pays
    bool touched = true;        // something other than f110_step_device went through the handle since the last step
    int last_blocks = 0;        // env blocks the most recent f110_step_device was submitted as
    int group_probes = 0;       // candidates the two-group form probed for a stream that runs next to the main stream
    bool groups_busy = false;   // group streams hold work the main stream has not been joined with
    bool main_dirty = true;     // the main stream holds work the group streams have not waited for
    // device RNG for the scan noise (f110_set_noise_rng)
    NoiseGen noise_gen{};
    uint64_t *d_zig_k = nullptr;
    double *d_zig_w = nullptr, *d_zig_f = nullptr;
    U128 *d_jump = nullptr;          // [2][65]
    U128 *d_rng_state = nullptr, *d_rng_seed = nullptr, *d_rng_rowstate = nullptr;
    int noise_rows_ready = 0;        // rows of the row cache generated so far
    int noise_rows_alloc = 0;        // rows the row cache has memory for (grows on demand up to dev.noise_rows)
    // the next rows are generated AHEAD of need on a stream of their own (noise_start_ahead): a row is 12.8 us of one wave, the rows of a
    // stream come one after the other, and a step that waits for them pays that in full (measured, F110Env: +13 us on every step of a first episode)
    hipStream_t noise_stream = nullptr;
    hipEvent_t ev_noise = nullptr, ev_noise_src = nullptr;
    bool noise_ahead = false;        // rows [noise_rows_ready, noise_ahead_upto) are being generated into noise_ahead_block
    int noise_ahead_upto = 0, noise_ahead_alloc = 0;
    double *noise_ahead_block = nullptr;
    std::vector<double *> noise_retired;   // earlier, smaller blocks of the cache (steps in flight may read them): freed with the cache
    long long noise_ub = 0;          // upper bound of any agent's step_count (steps since the last full reset)
    unsigned long long *d_lookups = nullptr;  // f110_scan_lookup_count
    bool lookups_on = false;
    // the single-block step captured as a HIP graph (f110_config.step_graph): one submission per step
    int collide_mode = 0;        // where the pair tests run: 0 side stream, 1 fused into k_integrate, 2 in line, 3 inside k_finalize (A = 2)
    // longest-first order of the scan tasks (TaskSched, small batches): double-buffered flags / lists / counters
    bool task_order = false;
    uint32_t *d_tflags[2] = {nullptr, nullptr}, *d_tlist[2] = {nullptr, nullptr}, *d_tcount = nullptr;
    TaskSched tsched[2]{};           // [parity], passed to the scan by value
    bool sched_allocated = false;
    uint32_t task_cap_alloc = 0;
    uint32_t task_epoch = 2, task_cap = 0, task_thr = 96;   // epochs start above the flags' initial 0
    uint32_t task_cap_div = 32, task_rev = 0;                // list capacity = tasks / task_cap_div; 1 = newest list entries first
    ExpSwitches exp;
    uint32_t *d_env_done = nullptr;   // [num_envs] scan_env_counter probe
    ncclComm_t comm = nullptr;   // optional RCCL communicator for the observation gather
    bool beams_uniform = true;   // f110_set_beam_tables: scan_angles is (to a quarter of a spacing) the ramp base_classes.py:133-134 builds
    int comm_ranks = 0;
    // overlapped gather (f110_comm_set_overlap): scans are double-buffered, the all-gather of step t
    // runs on comm_stream while step t+1 fills the other buffer
    bool comm_overlap = false, comm_swap_next = false, comm_inflight = false;
    double *scan_bufs[2] = {nullptr, nullptr};
    double *obs_scal[2] = {nullptr, nullptr};   // [7][N] scalar observation block(s) of the observation gather
    float *scan_f32[2] = {nullptr, nullptr};    // [N][B] float32 copies of the scans (F110_GATHER_F32 transport)
    int comm_rank = -1;
    bool comm_send_scalars = false;
    int scans_cur = 0;
    bool gather_pending[2] = {false, false};
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_step_done = nullptr, ev_gather_done[2] = {nullptr, nullptr};
    EpisodeArrays ep{};
    bool has_episode = false;
    double *d_rot_stage = nullptr;
    void *d_packed = nullptr;   // f110_episode_step_host's packed block
    // f110_step_host: the caller's block as last validated, and its device-side pointers
    bool hb_valid = false;
    f110_host_block hb_host{};
    HostBlock hb_dev{};
    const double *hb_actions_host = nullptr, *hb_actions_dev = nullptr;
    bool hb_scans_by_kernel = false;             // the validated block's scans are stored by the kernel (small batch, page-locked)
    unsigned long long *hb_seq_host = nullptr;   // page-locked completion word (F110_STEP_SPIN_WAIT)
    unsigned int *hb_blocks_done = nullptr;
    unsigned long long hb_seq = 0;
    double hs_enqueue_us = 0, hs_wait_us = 0;    // f110_step_host_stats
    FusedHost *d_fused = nullptr;                // device copy of {episode arrays, host block, flags} for the pair kernel's epilogue
    FusedHost fused_host_copy{};                 // what d_fused holds
    bool fused_valid = false, fuse_request = false, fused_done = false;
    unsigned long long fuse_seq = 0;
    long long hs_calls = 0;
    // the whole step of a tiny batch as ONE launch (k_step_tiny): counters + shadow columns, allocated on first use
    TinyCtl tiny{};
    void *tiny_mem = nullptr;
    int tiny_off = 0;                 // lab A/B (f110_exp_set "step_tiny" = 0): the three-kernel form also for tiny batches
    int last_launches = 0;            // kernels the most recent step submitted its work as (f110_step_launches)
    const double *tiny_actions_host = nullptr;   // f110_step_host -> step_tiny: the caller's [N][2] actions (host memory), for this call
    bool tiny_request = false;        // f110_step_host: this step is one k_step_tiny launch
    bool tiny_host_request = false;   // ... with one agent per env: what k_host_block would be handed
    HostBlock tiny_hb{};
    int tiny_episode = 0, tiny_auto_reset = 0;
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

static int group_envs(const f110_sim *h);   // envs per env block (defined with f110_step_device)

// The group streams' outstanding work becomes a dependency of the main stream (stream-ordered, no
// host wait).  Every entry point except the step itself starts with it, so the env groups are an
// internal detail: anything enqueued or read through the handle sees completed steps.
static int join_groups(f110_sim *h)
{
    if (h->comm_inflight) {   // an overlapped gather: whatever follows on the main stream sees its result
        HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_gather_done[0], 0));
        HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_gather_done[1], 0));
        h->comm_inflight = false;
    }
    if (!h->groups_busy) return F110_OK;
    for (size_t g = 0; g < h->gstreams.size(); ++g) {
        if (h->gstreams[g] == h->stream) continue;
        HIPCHK(h, hipEventRecord(h->gevents[g], h->gstreams[g]));
        HIPCHK(h, hipStreamWaitEvent(h->stream, h->gevents[g], 0));
    }
    h->groups_busy = false;
    return F110_OK;
}

// first statement (after the null check) of every entry point that takes a handle: calls may come
// from any thread, so the handle's device is made current; group streams are joined; whatever the
// call enqueues on the main stream must be waited for by the next step's groups.
#define ENTER(h)                                          \
    do {                                                  \
        HIPCHK(h, hipSetDevice((h)->cfg.device_id));      \
        TRY(join_groups(h));                              \
        (h)->main_dirty = true;                           \
        (h)->touched = true;                              \
    } while (0)

static bool tiny_applies(const f110_sim *h);   // (below, with the step)
static void noise_release(f110_sim *h);        // (below, with the noise entry points)

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

static inline bool padded_family(int layout) { return layout == F110_MAP_PADDED_F64; }

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


// the device copy of h->k, brought up to date (stream-ordered) if any field changed
static const ScanConst *cold_consts(f110_sim *h)
{
    if (!h->d_k && hipMalloc(reinterpret_cast<void **>(&h->d_k), sizeof(ScanConst)) != hipSuccess) return nullptr;
    if (std::memcmp(&h->k, &h->k_uploaded, sizeof(ScanConst)) != 0) {
        std::memcpy(&h->k_uploaded, &h->k, sizeof(ScanConst));
        if (hipMemcpyAsync(h->d_k, &h->k_uploaded, sizeof(ScanConst), hipMemcpyHostToDevice, h->stream) != hipSuccess) return nullptr;
    }
    return h->d_k;
}

// The step's scan runs agent-aligned (k_scan_rays_agent) with the PADDED layout unless that would
// leave more than 3 % of the lanes idle (few beams); exp.scan_flat forces the flat kernel (A/B).
static bool agent_aligned(const f110_sim *h)
{
    if (h->exp.scan_flat || !padded_family(h->cfg.map_layout) || !h->k.pad || h->dir_stride > 0) return false;
    const int B = h->k.num_beams, lanes = (B + 63) / 64 * 64;
    return (lanes - B) * 100 <= 3 * B;
}

template <bool STEP>
static scan_rays_fn pick_rays(const ScanConst &k, int layout)
{
#define SEL(L) (k.res_pow2 ? (k.ident_rot ? k_scan_rays<L, true, true, STEP> : k_scan_rays<L, true, false, STEP>) \
                           : (k.ident_rot ? k_scan_rays<L, false, true, STEP> : k_scan_rays<L, false, false, STEP>))
    if (padded_family(layout) && k.pad) return SEL(LAYOUT_PADDED);
    return SEL(LAYOUT_ROWMAJOR);
#undef SEL
}

static dim3 rays_grid(RayJob &j, int block, int tasks_per_wave)
{
    j.n_tasks = (j.n_rays + 63u) / 64u;
    j.tasks_per_wave = tasks_per_wave > 0 ? (uint32_t)tasks_per_wave : 1u;
    const uint32_t waves = (j.n_tasks + j.tasks_per_wave - 1) / j.tasks_per_wave;
    const uint32_t wpb = (uint32_t)block / 64u;
    return dim3((waves + wpb - 1) / wpb);
}

// longest-first order of the step's scan tasks (TaskSched): buffers on first use, then on / off
static int task_order_setup(f110_sim *h, bool on)
{
    const size_t n_tasks = (size_t)h->N * (((size_t)h->cfg.num_beams + 63) / 64);
    if (!on || n_tasks >= 0x7fffffffu) {
        h->task_order = false;
        return F110_OK;
    }
    if (!h->sched_allocated) {
        h->task_cap_alloc = (uint32_t)std::max<size_t>(64, kExperimental ? n_tasks : n_tasks / h->task_cap_div);
        for (int q = 0; q < 2; ++q) {
            TRY(dmalloc(h, &h->d_tflags[q], n_tasks));
            TRY(dmalloc(h, &h->d_tlist[q], (size_t)h->task_cap_alloc));
            HIPCHK(h, hipMemsetAsync(h->d_tflags[q], 0, sizeof(uint32_t) * n_tasks, h->stream));
        }
        TRY(dmalloc(h, &h->d_tcount, 4));   // {task count of parity 0, of parity 1, spare, spare}
        HIPCHK(h, hipMemsetAsync(h->d_tcount, 0, 4 * sizeof(uint32_t), h->stream));
        h->sched_allocated = true;
    }
    h->task_cap = std::min<uint32_t>(h->task_cap_alloc, (uint32_t)std::max<size_t>(64, n_tasks / h->task_cap_div));
    for (int q = 0; q < 2; ++q)   // struct q is used at steps of parity q: it reads what parity q^1 wrote
        h->tsched[q] = TaskSched{h->d_tflags[q ^ 1], h->d_tflags[q], h->d_tlist[q ^ 1], h->d_tlist[q], h->d_tcount + (q ^ 1), h->d_tcount + q, h->task_cap, h->task_thr, {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}, 0u, 0u};
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->task_order = true;
    return F110_OK;
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

int f110_device_pci_bus_id(int32_t device, char *out, int32_t len)
{
    if (!out || len < 16) return fail(nullptr, F110_ERR_INVALID, "f110_device_pci_bus_id: buffer of at least 16 bytes required");
    const hipError_t e = hipDeviceGetPCIBusId(out, len, device);
    if (e != hipSuccess) return fail(nullptr, F110_ERR_HIP, "hipDeviceGetPCIBusId(%d) failed: %s", device, hipGetErrorString(e));
    return F110_OK;
}

#ifndef F110_SRC_HASH
#define F110_SRC_HASH "unknown"
#endif
const char *f110_build_info(void) { return "csrc=" F110_SRC_HASH; }

int f110_is_experimental(void) { return kExperimental ? 1 : 0; }

int f110_exp_set(f110_sim *h, const char *key, int32_t value)
{
    if (!h || !key) return fail(h, F110_ERR_INVALID, "null argument");
    if (!kExperimental) return fail(h, F110_ERR_STATE, "f110_exp_set(%s) is available in the experimental build only (libf110_hip_exp.so)", key);
    ENTER(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const std::string k(key);
    if (k == "scan_flat") h->exp.scan_flat = value;
    else if (k == "long_prio") h->exp.long_prio = value;
    else if (k == "scan_occupancy") h->exp.scan_occupancy = value;
    else if (k == "scan_env_counter") h->exp.scan_env_counter = value;
    else if (k == "integrate_duo") h->exp.integrate_duo = value;
    else if (k == "integrate_fan") h->exp.integrate_fan = value;
    else if (k == "group_split") h->exp.group_split = value;
    else if (k == "step_tiny") h->tiny_off = value == 0 ? 1 : 0;
    else if (k == "tiny_start_probe") {   // 1: k_step_tiny's first workgroup reports its start to the host; 0: print launch -> start -> word and stop
        if (!value && h->exp.tiny_start_probe && h->exp.probe_calls)
            std::fprintf(stderr, "[tiny_start_probe] %ld steps: call -> launched %.2f us, launched -> first wave seen %.2f us, first wave seen -> completion word seen %.2f us\n",
                         h->exp.probe_calls, h->exp.probe_us[0] / h->exp.probe_calls, h->exp.probe_us[1] / h->exp.probe_calls, h->exp.probe_us[2] / h->exp.probe_calls);
        h->exp.tiny_start_probe = value;
        h->exp.probe_calls = 0;
        h->exp.probe_us[0] = h->exp.probe_us[1] = h->exp.probe_us[2] = 0;
    }
    else if (k == "tiny_general_tail") h->exp.tiny_general_tail = value;   // 1: one env of two cars finishes in finalize_pair_body (the tail before finalize_duo_tiny)
    else if (k == "tiny_trace_hi") h->exp.tiny_trace = (h->exp.tiny_trace & 0xffffffffull) | ((uint64_t)(uint32_t)value << 32);
    else if (k == "tiny_trace_lo") h->exp.tiny_trace = (h->exp.tiny_trace & ~0xffffffffull) | (uint64_t)(uint32_t)value;
    else if (k == "scan_trace_hi") h->exp.scan_trace = (h->exp.scan_trace & 0xffffffffull) | ((uint64_t)(uint32_t)value << 32);
    else if (k == "scan_trace_lo") h->exp.scan_trace = (h->exp.scan_trace & ~0xffffffffull) | (uint64_t)(uint32_t)value;
    else if (k == "collide_mode") {
        if (value < 0 || value > 3) return fail(h, F110_ERR_INVALID, "collide_mode must be 0..3");
        h->collide_mode = value;
    }
    else if (k == "task_order") return task_order_setup(h, value != 0);
    else if (k == "task_cap_div" || k == "task_rev") {
        if (k == "task_rev") h->task_rev = value != 0;
        else h->task_cap_div = (uint32_t)std::max(1, value);
        if (h->task_order) return task_order_setup(h, true);
    } else if (k == "task_thr") {
        h->task_thr = (uint32_t)value;
        if (h->task_order) return task_order_setup(h, true);
    }
    else if (k == "pad_tiled" || k == "scan_nt" || k == "finalize_wave" || k == "spec_from" || k == "scan_stream" || k == "stream_refill" || k == "stream_block" ||
             k == "stream_grid")
        return fail(h, F110_ERR_INVALID, "f110_exp_set: '%s' was retired in round 6 with the variant it switched (DESIGN.md section 8)", key);
    else
        return fail(h, F110_ERR_INVALID, "f110_exp_set: unknown key '%s'", key);
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

// f110_kernels.hpp "do two streams make progress independently of each other?"; *yes is false whenever in doubt
static hipError_t streams_concurrent(hipStream_t a, hipStream_t b, bool *yes)
{
    *yes = false;
    unsigned *d = nullptr, hres[2] = {0u, 0u};
    hipError_t e = hipMalloc(&d, 2 * sizeof(unsigned));
    if (e != hipSuccess) return e;
    do {
        if ((e = hipMemsetAsync(d, 0, 2 * sizeof(unsigned), a)) != hipSuccess) break;
        if ((e = hipStreamSynchronize(a)) != hipSuccess) break;
        if ((e = hipStreamSynchronize(b)) != hipSuccess) break;
        hipLaunchKernelGGL(k_probe_wait, dim3(1), dim3(1), 0, a, d, d + 1, 30000ull);   // at most 300 us
        hipLaunchKernelGGL(k_probe_set, dim3(1), dim3(1), 0, b, d);
        if ((e = hipStreamSynchronize(a)) != hipSuccess) break;
        if ((e = hipStreamSynchronize(b)) != hipSuccess) break;
        if ((e = hipMemcpy(hres, d, sizeof hres, hipMemcpyDeviceToHost)) != hipSuccess) break;
        *yes = hres[1] != 0u;
    } while (0);
    (void)hipFree(d);
    return e;
}

// live handles and page-locked blocks: f110_step_host caches the device view of the host block it validated, keyed by
// host pointers — a block that is freed must drop every handle's cache of it, or a later allocation at the same addresses would
// be stepped into through stale device pointers
static std::mutex g_registry_mu;
static std::set<f110_sim *> g_handles;
static std::map<const char *, size_t> g_host_blocks;

int f110_create(const f110_config *cfg, f110_sim **out)
{
    if (!cfg || !out) return fail(nullptr, F110_ERR_INVALID, "f110_create: null argument");
    *out = nullptr;
    if (cfg->abi_version != F110_ABI_VERSION) return fail(nullptr, F110_ERR_INVALID, "ABI version mismatch (%d vs %d)", cfg->abi_version, F110_ABI_VERSION);
    if (cfg->num_envs < 1 || cfg->num_agents < 1 || cfg->num_beams < 2 || cfg->theta_dis < 2)
        return fail(nullptr, F110_ERR_INVALID, "f110_create: num_envs/num_agents >= 1, num_beams/theta_dis >= 2 required");
    if (cfg->integrator != F110_INTEGRATOR_RK4 && cfg->integrator != F110_INTEGRATOR_EULER)
        return fail(nullptr, F110_ERR_INVALID, "Invalid Integrator Specified. Please choose RK4 or Euler");
    if (cfg->map_layout == F110_MAP_TILED_F64 || cfg->map_layout == F110_MAP_CODE8 || cfg->map_layout == F110_MAP_WINDOW_LDS)
        return fail(nullptr, F110_ERR_INVALID, "map_layout %d (4x4 tiles / byte codes + LUT / LDS window) was measured slower in rounds 1-4 and retired in round 5: "
                    "use F110_MAP_PADDED_F64 (3) or F110_MAP_ROWMAJOR_F64 (0)", cfg->map_layout);
    if (cfg->map_layout != F110_MAP_ROWMAJOR_F64 && cfg->map_layout != F110_MAP_PADDED_F64)
        return fail(nullptr, F110_ERR_INVALID, "unknown map_layout %d", cfg->map_layout);
    if (cfg->step_graph != 0)
        return fail(nullptr, F110_ERR_INVALID, "step_graph (the step as one captured HIP graph) was measured slower (graph launch ~250 us on ROCm 7.2) and retired in round 5");
    if (!kExperimental && cfg->step_groups > 2)
        return fail(nullptr, F110_ERR_STATE, "step_groups > 2 is available in the experimental build only (libf110_hip_exp.so)");
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
    if (cfg->scan_tasks_per_wave > 0) {
        h->scan_tasks_per_wave = cfg->scan_tasks_per_wave;
    } else {
        // default: 3 consecutive tasks per wave once the batch fills every wave slot ~8 times over
        // (amortises the per-wave set-up); small batches are bound by their longest rays and want
        // the finer granularity (4096 agents: +5 %, 1024: +12 % with 1 task per wave).  Round 3, 3 / 4 / 6 / 8 / 17
        // tasks per wave: 65 536 agents 95.7 / 95.2 / 95.1 / 94.3 / 88.4 M agent-steps/s, 32 768: 87.6 / 86.2 / 85.9 / 83.3 / 77.9
        const size_t tasks = ((size_t)N * (size_t)B + 63) / 64, slots = (size_t)h->num_cus * 32;
        const size_t t = tasks / (slots * 8);
        h->scan_tasks_per_wave = t < 1 ? 1 : (t > 3 ? 3 : (int)t);
        h->scan_tasks_auto = true;
    }
    CKH(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    CKH(hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking));
    CKH(hipEventCreateWithFlags(&h->ev_integrated, hipEventDisableTiming));
    CKH(hipEventCreateWithFlags(&h->ev_collided, hipEventDisableTiming));
    CKH(hipEventCreate(&h->ev_begin));
    CKH(hipEventCreate(&h->ev_end));
    CKH(hipEventCreateWithFlags(&h->ev_main, hipEventDisableTiming));
    {
        // env groups: envs are independent of each other, so two halves of the batch need not run in lockstep.
        // Steps that the caller enqueues back to back (f110_step_device, nothing else through the handle in between)
        // are submitted as two env blocks on two streams: one block's latency-bound kernels, tails and the VALU-bound
        // finalize run under the other block's texture-bound scan, and the blocks drift apart over consecutive steps.
        // 0 = automatic (two blocks when steps come back to back AND the size is one where it pays: env_blocks_pay),
        // 1 = always one block, 2 = always two (a caller that synchronises every
        // step pays ~25 us of fork / join for it), > 2 = experimental build.
        int G = cfg->step_groups;
        h->groups_auto = G <= 0;
        if (G <= 0) G = 2;
        G = std::min(std::min(G, 16), cfg->num_envs);
        h->groups = G;
        // pair tests + opponent windows inside the finalize kernel: A = 2 (k_finalize_pair_roles) and every A up to
        // kMaxAgentsMulti = 256 (k_finalize_multi up to 16, k_finalize_multi_tiled above: an env's ordered pairs in tiles of a
        // workgroup's record table); above
        // that — no use case known — the round-1 form (k_collide on the side stream + k_finalize)
        h->collide_mode = (cfg->num_agents >= 2 && cfg->num_agents <= kMaxAgentsMulti) ? 3 : 0;
        // two groups: the main stream and ONE more stream that provably runs next to it.  Which hardware queue a stream
        // lands on depends on what else the process created before; a second stream on the main stream's queue turns
        // the gain (+2 .. +23 %) into a loss (-9 .. -50 %), so candidates are probed (k_probe_wait) — the side stream
        // first (unused by the in-kernel pair tests), then up to six fresh ones; with none, the step stays one block.
        const bool duo = G == 2 && (h->collide_mode == 3 || cfg->num_agents == 1);   // (the older forms launch k_collide on the side stream)
        if (h->groups_auto && !duo) G = h->groups = 1;
        if (duo) {
            hipStream_t second = nullptr;
            std::vector<hipStream_t> rejected;
            for (int attempt = 0; attempt < 7 && !second; ++attempt) {
                hipStream_t cand = h->side_stream;
                if (attempt > 0 && hipStreamCreateWithFlags(&cand, hipStreamNonBlocking) != hipSuccess) {
                    (void)hipGetLastError();   // no more streams to be had: the search ends, the step stays one block
                    break;
                }
                bool yes = false;
                const hipError_t ep = streams_concurrent(h->stream, cand, &yes);
                if (ep == hipSuccess && yes) {
                    second = cand;
                    if (attempt > 0) h->gowned.push_back(cand);
                } else if (attempt > 0) {
                    rejected.push_back(cand);   // kept until the search ends, so that the next candidate lands elsewhere
                }
                h->group_probes = attempt + 1;
                if (ep != hipSuccess) break;
            }
            for (hipStream_t r : rejected) (void)hipStreamDestroy(r);
            if (!second) G = h->groups = 1;
            else {
                hipEvent_t ge = nullptr;
                h->gstreams.push_back(h->stream);
                h->gevents.push_back(nullptr);
                h->gstreams.push_back(second);
                CKH(hipEventCreateWithFlags(&ge, hipEventDisableTiming));
                h->gevents.push_back(ge);
            }
        }
        for (int g = 0; g < G && G > 1 && !duo; ++g) {
            hipStream_t gs = nullptr;
            hipEvent_t ge = nullptr;
            CKH(hipStreamCreateWithFlags(&gs, hipStreamNonBlocking));
            h->gowned.push_back(gs);
            h->gstreams.push_back(gs);
            CKH(hipEventCreateWithFlags(&ge, hipEventDisableTiming));
            h->gevents.push_back(ge);
        }
    }
    AgentArrays &d = h->dev;
    d.n_agents_total = N;
    d.agents_per_env = cfg->num_agents;
    d.agent_begin = 0;
    d.agent_count = N;
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
    {
        // constants of the device noise generator (f110_rng.hpp): NumPy's ziggurat tables and the
        // PCG64 jump constants, 8 KB in all
        PcgJump jt;
        pcg_jump_table(jt);
        CK(dmalloc(h, &h->d_zig_k, 256));
        CK(dmalloc(h, &h->d_zig_w, 256));
        CK(dmalloc(h, &h->d_zig_f, 256));
        CK(dmalloc(h, &h->d_jump, 2 * 65));
        CK(dmalloc(h, &h->d_lookups, 2));
        CKH(hipMemcpy(h->d_zig_k, kZigK, sizeof kZigK, hipMemcpyHostToDevice));
        CKH(hipMemcpy(h->d_zig_w, kZigW, sizeof kZigW, hipMemcpyHostToDevice));
        CKH(hipMemcpy(h->d_zig_f, kZigF, sizeof kZigF, hipMemcpyHostToDevice));
        CKH(hipMemcpy(h->d_jump, jt.a, sizeof jt.a, hipMemcpyHostToDevice));
        CKH(hipMemcpy(h->d_jump + 65, jt.g, sizeof jt.g, hipMemcpyHostToDevice));
        CKH(hipMemsetAsync(h->d_lookups, 0, 2 * sizeof(unsigned long long), h->stream));
        h->noise_gen = NoiseGen{h->d_zig_k, h->d_zig_w, h->d_zig_f, h->d_jump, h->d_jump + 65, 0.0};
    }
    {
        // longest-first task order: for batches whose scan ends with its longest rays.  Window re-measured in round 3
        // with both kernels at 8 waves per SIMD (product kernel, list on / off): 512 agents 8.50 / 8.68 M agent-steps/s,
        // 1024: 15.4 / 15.2, 2048: 27.5 / 26.1, 4096: 43.7 / 39.9, 8192: 60.9 / 57.8, 12 288: 69.7 / 67.4,
        // 16 384: 75.2 / 73.6, 24 576: 81.0 / 83.0, 32 768: 82.6 / 86.3
        const size_t n_tasks = (size_t)N * (((size_t)B + 63) / 64);
#ifndef F110_TASK_ORDER_MAX_TASKS
#define F110_TASK_ORDER_MAX_TASKS 340000
#endif
        if (n_tasks >= 12000 && n_tasks < F110_TASK_ORDER_MAX_TASKS) CK(task_order_setup(h, true));
    }
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
        // more beams than table directions: march each distinct direction once (k_scan_dirs_agent)
        int stride = (int)std::ceil((B - 1) * k.theta_inc) + 2;
        // The direction kernel walks an agent's beams in order of their table index RELATIVE to beam 0's, which must not wrap:
        // a scan that spans the whole table — fov close to a full turn, e.g. 6.28 with 1000 directions: (B - 1) theta_inc =
        // 999.5, plus the start's fraction — ends on beam 0's directions again and those beams were left unwritten (found by
        // round 5's fuzz over constructor arguments).  Such scans march every beam (k_scan_rays_agent).
        const bool wraps = (B - 1) * k.theta_inc + 1.0 >= (double)cfg->theta_dis;
        stride = std::min(stride, cfg->theta_dis);
        stride = (stride + 63) / 64 * 64;
        if (!wraps && stride < B && (long long)N * stride < 0xFFFFFF00LL) {
            h->dir_stride = stride;
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
    {
        std::lock_guard<std::mutex> lk(g_registry_mu);
        g_handles.insert(h);
    }
    *out = h;
    return F110_OK;
}

void f110_destroy(f110_sim *h)
{
    if (!h) return;
    {   // first: no f110_host_free on another thread may look at this handle (or drain its streams) any more
        std::lock_guard<std::mutex> lk(g_registry_mu);
        g_handles.erase(h);
    }
    (void)hipSetDevice(h->cfg.device_id);
    for (hipStream_t gs : h->gstreams) (void)hipStreamSynchronize(gs);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->side_stream) (void)hipStreamSynchronize(h->side_stream);
    (void)f110_comm_destroy(h);
    h->comm_inflight = false;
    if (h->comm_stream) { (void)hipStreamSynchronize(h->comm_stream); (void)hipStreamDestroy(h->comm_stream); h->comm_stream = nullptr; }
    for (hipEvent_t e : {h->ev_step_done, h->ev_gather_done[0], h->ev_gather_done[1]})
        if (e) (void)hipEventDestroy(e);
    if (h->comm_overlap || h->scan_bufs[1]) {
        if (h->scan_bufs[0]) h->dev.scans = h->scan_bufs[0];   // the list below frees dev.scans
        if (h->scan_bufs[1]) (void)hipFree(h->scan_bufs[1]);
    }
    for (double *p : h->obs_scal)
        if (p) (void)hipFree(p);
    for (float *p : h->scan_f32)
        if (p) (void)hipFree(p);
    for (hipStream_t gs : h->gowned) (void)hipStreamDestroy(gs);
    for (hipEvent_t ge : h->gevents)
        if (ge) (void)hipEventDestroy(ge);
    if (h->ev_main) (void)hipEventDestroy(h->ev_main);
    noise_release(h);
    if (h->noise_stream) (void)hipStreamDestroy(h->noise_stream);
    if (h->ev_noise) (void)hipEventDestroy(h->ev_noise);
    if (h->ev_noise_src) (void)hipEventDestroy(h->ev_noise_src);
    {
        void *rp[] = {h->d_env_done, h->d_tflags[0], h->d_tflags[1], h->d_tlist[0], h->d_tlist[1], h->d_tcount, h->d_zig_k, h->d_zig_w, h->d_zig_f, h->d_jump, h->d_rng_state, h->d_rng_seed, h->d_rng_rowstate, h->d_lookups};
        for (void *p : rp)
            if (p) (void)hipFree(p);
    }
    AgentArrays &d = h->dev;
    void *ptrs[] = {d.opp_verts, d.ray_hdr, d.opp_window, d.state, d.steer_buf, d.buf_cnt, d.scan_pose, d.snap_pose, d.dir_start, d.scans, d.collisions,
                    d.collision_idx, d.in_collision, d.step_count, h->d_params, h->d_noise, h->d_scan_angles,
                    h->d_beam_cos, h->d_side, h->d_dt_row, h->d_dt_pad, h->d_actions, h->d_poses, h->d_cs, h->d_mask, h->d_path_stats, h->d_k, h->d_params_all};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    for (auto &ms : h->extra_maps) {
        if (ms.d_dt_row) (void)hipFree(ms.d_dt_row);
        if (ms.d_dt_pad) (void)hipFree(ms.d_dt_pad);
    }
    if (h->d_maps_fast) (void)hipFree(h->d_maps_fast);
    if (h->d_maps_full) (void)hipFree(h->d_maps_full);
    if (h->d_env_map) (void)hipFree(h->d_env_map);
    if (h->d_scan_order) (void)hipFree(h->d_scan_order);
    {
        void *eptrs[] = {h->ep.start_poses, h->ep.rot, h->ep.current_time, h->ep.near_start, h->ep.toggle,
                         h->ep.lap_count, h->ep.lap_time, h->ep.done, h->ep.checkpoint, h->d_rot_stage, h->d_packed};
        for (void *p : eptrs)
            if (p) (void)hipFree(p);
    }
    if (h->hb_seq_host) (void)hipHostFree(h->hb_seq_host);
    if (h->d_fused) (void)hipFree(h->d_fused);
    if (h->tiny_mem) (void)hipFree(h->tiny_mem);
    if (h->hb_blocks_done) (void)hipFree(h->hb_blocks_done);
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
    ENTER(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return F110_OK;
}

// ---- map ---------------------------------------------------------------------------------
// the per-map fields of ScanConst that do not involve device memory
static void fill_map_fields(ScanConst &k, int H, int W, double res, double ox, double oy, double oc, double os)
{
    k.height = H;
    k.width = W;
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
}

static int finish_map(f110_sim *h, int H, int W, double res, double ox, double oy, double oc, double os)
{
    ScanConst &k = h->k;
    h->multi_map = false;       // slot 0 changed: f110_set_env_maps has to be called again
    h->dev.maps_full = nullptr;
    h->dev.env_map = nullptr;
    fill_map_fields(k, H, W, res, ox, oy, oc, os);
    HIPCHK(h, hipMemcpyAsync(&k.oob_value, h->d_dt_row + ((size_t)H * W - 1), sizeof(double), hipMemcpyDeviceToHost, h->stream));
    k.table = h->d_dt_row;
    k.table_rm = h->d_dt_row;
    k.pad = nullptr;
    if (h->d_dt_pad) { (void)hipFree(h->d_dt_pad); h->d_dt_pad = nullptr; }
    if (padded_family(h->cfg.map_layout) && setup_padded(k)) {
        // the table again with a border of out-of-bounds cells wide enough for any ray of a lidar
        // that is on (or within kPadSlack cells of) the map; maps too large for 16-bit cell
        // coordinates keep k.pad == nullptr and run the plain row-major kernel
        const size_t total = (size_t)k.pad_width * k.pad_height;
        TRY(dmalloc(h, &h->d_dt_pad, total));
        hipLaunchKernelGGL(k_build_padded, grid1d(total, 256), dim3(256), 0, h->stream, h->d_dt_row, H, W, k.pad_border, k.pad_width, k.pad_height,
                           h->d_dt_pad);
        HIPCHK(h, hipGetLastError());
        k.pad = h->d_dt_pad;
        if (h->cfg.map_layout == F110_MAP_PADDED_F64) {
            // ONE table per map: the exact paths (k_integrate's first sample, the cold re-march, the unit
            // kernels) address dt[r][c] as table_rm + r * row_bytes + 8 c — point them at the interior of
            // the padded copy (its row pitch) and give the row-major original back (20.5 MB on example_map,
            // and the first sample now touches lines the march keeps hot)
            HIPCHK(h, hipStreamSynchronize(h->stream));
            k.table = k.table_rm = h->d_dt_pad + (size_t)k.pad_border * k.pad_width + k.pad_border;
            k.row_bytes = k.pad_row_bytes;
            (void)hipFree(h->d_dt_row);
            h->d_dt_row = nullptr;
        }
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->has_map = true;
    return F110_OK;
}

int f110_set_map_image(f110_sim *h, const uint8_t *h_img, int32_t H, int32_t W, double res, double ox, double oy, double oyaw)
{
    if (!h || !h_img) return fail(h, F110_ERR_INVALID, "f110_set_map_image: null argument");
    ENTER(h);
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
    ENTER(h);
    if (H < 1 || W < 1 || !(res > 0)) return fail(h, F110_ERR_INVALID, "f110_set_map_dt: bad shape or resolution");
    if ((unsigned long long)(H + 3) * (unsigned long long)(W + 3) * 8ull >= 0xFFFFFFFFull) return fail(h, F110_ERR_INVALID, "distance table must stay below 4 GiB");
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    const size_t n = (size_t)H * W;
    if (h->d_dt_row) { (void)hipFree(h->d_dt_row); h->d_dt_row = nullptr; }
    TRY(dmalloc(h, &h->d_dt_row, n));
    HIPCHK(h, hipMemcpyAsync(h->d_dt_row, h_dt, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    return finish_map(h, H, W, res, ox, oy, oc, os);
}

// ---- a different track per env ---------------------------------------------------------------------
// the exact-EDT pipeline of f110_set_map_image into a fresh row-major table
static int edt_table_from_image(f110_sim *h, const uint8_t *h_img, int H, int W, double res, double **d_dt_row)
{
    const size_t n = (size_t)H * W;
    uint8_t *d_img = nullptr, *d_bin = nullptr;
    uint32_t *d_g = nullptr, *d_d2 = nullptr;
    Scratch s(h);
    TRY(s.up(h_img, n, &d_img));
    TRY(s.up<uint8_t>(nullptr, n, &d_bin));
    TRY(s.up<uint32_t>(nullptr, n, &d_g));
    TRY(s.up<uint32_t>(nullptr, n, &d_d2));
    TRY(dmalloc(h, d_dt_row, n));
    hipLaunchKernelGGL(k_flip_threshold, grid1d(n, 256), dim3(256), 0, h->stream, d_img, H, W, d_bin);
    hipLaunchKernelGGL(k_edt_columns, grid1d(W, 64), dim3(64), 0, h->stream, d_bin, H, W, d_g);
    hipLaunchKernelGGL(k_edt_rows, dim3(H), dim3(256), (size_t)W * sizeof(uint32_t), h->stream, d_g, H, W, d_d2);
    hipLaunchKernelGGL(k_dt_from_d2, grid1d(n, 256), dim3(256), 0, h->stream, d_d2, n, res, *d_dt_row);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return F110_OK;
}

static int add_map_slot(f110_sim *h, double *d_dt_row, int H, int W, double res, double ox, double oy, double oc, double os, int32_t *slot)
{
    f110_sim::MapSlot ms;
    ms.d_dt_row = d_dt_row;
    ms.k = h->k;   // beam / trig / range constants are shared; the map fields follow
    fill_map_fields(ms.k, H, W, res, ox, oy, oc, os);
    ms.k.table = ms.k.table_rm = d_dt_row;
    HIPCHK(h, hipMemcpyAsync(&ms.k.oob_value, d_dt_row + ((size_t)H * W - 1), sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (!setup_padded(ms.k)) {
        (void)hipFree(d_dt_row);
        return fail(h, F110_ERR_INVALID, "f110_add_map: a per-env map must fit the padded layout (16-bit cell coordinates, < 4 GiB)");
    }
    const size_t total = (size_t)ms.k.pad_width * ms.k.pad_height;
    if (dmalloc(h, &ms.d_dt_pad, total) != F110_OK) {
        (void)hipFree(d_dt_row);
        return F110_ERR_NOMEM;
    }
    hipLaunchKernelGGL(k_build_padded, grid1d(total, 256), dim3(256), 0, h->stream, d_dt_row, H, W, ms.k.pad_border, ms.k.pad_width,
                       ms.k.pad_height, ms.d_dt_pad);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    ms.k.pad = ms.d_dt_pad;
    // one table per track (see finish_map): the exact paths read the padded copy's interior
    ms.k.table = ms.k.table_rm = ms.d_dt_pad + (size_t)ms.k.pad_border * ms.k.pad_width + ms.k.pad_border;
    ms.k.row_bytes = ms.k.pad_row_bytes;
    (void)hipFree(ms.d_dt_row);
    ms.d_dt_row = nullptr;
    h->extra_maps.push_back(ms);
    if (slot) *slot = (int32_t)h->extra_maps.size();   // slot 0 is the map of f110_set_map_*
    return F110_OK;
}

int f110_add_map_image(f110_sim *h, const uint8_t *h_img, int32_t H, int32_t W, double res, double ox, double oy, double oyaw, int32_t *slot)
{
    if (!h || !h_img) return fail(h, F110_ERR_INVALID, "f110_add_map_image: null argument");
    ENTER(h);
    if (H < 1 || W < 1 || H > 16384 || W > 16384 || !(res > 0)) return fail(h, F110_ERR_INVALID, "f110_add_map_image: bad shape %dx%d or resolution", H, W);
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    double *d_dt_row = nullptr;
    TRY(edt_table_from_image(h, h_img, H, W, res, &d_dt_row));
    return add_map_slot(h, d_dt_row, H, W, res, ox, oy, std::cos(oyaw), std::sin(oyaw), slot);
}

int f110_add_map_dt(f110_sim *h, const double *h_dt, int32_t H, int32_t W, double res, double ox, double oy, double oc, double os, int32_t *slot)
{
    if (!h || !h_dt) return fail(h, F110_ERR_INVALID, "f110_add_map_dt: null argument");
    ENTER(h);
    if (H < 1 || W < 1 || !(res > 0)) return fail(h, F110_ERR_INVALID, "f110_add_map_dt: bad shape or resolution");
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    double *d_dt_row = nullptr;
    TRY(dmalloc(h, &d_dt_row, (size_t)H * W));
    HIPCHK(h, hipMemcpyAsync(d_dt_row, h_dt, (size_t)H * W * sizeof(double), hipMemcpyHostToDevice, h->stream));
    return add_map_slot(h, d_dt_row, H, W, res, ox, oy, oc, os, slot);
}

int f110_set_env_maps(f110_sim *h, const int32_t *h_env_map)
{
    if (!h) return fail(nullptr, F110_ERR_INVALID, "null handle");
    ENTER(h);
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    if (!h_env_map) {   // back to one map for everybody
        h->multi_map = false;
        h->dev.maps_full = nullptr;
        h->dev.env_map = nullptr;
        return F110_OK;
    }
    if (!h->has_map) return fail(h, F110_ERR_NO_MAP, "Map is not set for scan simulator.");
    if (!padded_family(h->cfg.map_layout) || !h->k.pad)
        return fail(h, F110_ERR_STATE, "f110_set_env_maps needs map_layout = F110_MAP_PADDED_F64 and a slot-0 map that fits it");
    const int E = h->cfg.num_envs, M = 1 + (int)h->extra_maps.size();
    for (int e = 0; e < E; ++e)
        if (h_env_map[e] < 0 || h_env_map[e] >= M) return fail(h, F110_ERR_INVALID, "f110_set_env_maps: env %d -> slot %d, but %d maps are registered", e, h_env_map[e], M);
    std::vector<ScanConst> full(M);
    std::vector<MapFast> fast(M);
    for (int m = 0; m < M; ++m) {
        full[m] = m == 0 ? h->k : h->extra_maps[m - 1].k;
        // constants that may have changed since a slot was added (trig / beam tables) are global
        full[m].cs = h->k.cs;
        fast[m].pad = full[m].pad;
        fast[m].pad_cx = full[m].pad_cx;
        fast[m].pad_cy = full[m].pad_cy;
        fast[m].pad_axx = full[m].pad_axx;
        fast[m].pad_axy = full[m].pad_axy;
        fast[m].pad_ayx = full[m].pad_ayx;
        fast[m].pad_ayy = full[m].pad_ayy;
        fast[m].pad_row_bytes = (uint32_t)full[m].pad_row_bytes;
        fast[m].pad_max_samples = full[m].pad_max_samples;
    }
    if (h->d_maps_fast) { (void)hipFree(h->d_maps_fast); h->d_maps_fast = nullptr; }
    if (h->d_maps_full) { (void)hipFree(h->d_maps_full); h->d_maps_full = nullptr; }
    if (!h->d_env_map) HIPCHK(h, hipMalloc(reinterpret_cast<void **>(&h->d_env_map), sizeof(int32_t) * E));
    HIPCHK(h, hipMalloc(reinterpret_cast<void **>(&h->d_maps_fast), sizeof(MapFast) * M));
    HIPCHK(h, hipMalloc(reinterpret_cast<void **>(&h->d_maps_full), sizeof(ScanConst) * M));
    HIPCHK(h, hipMemcpyAsync(h->d_maps_fast, fast.data(), sizeof(MapFast) * M, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_maps_full, full.data(), sizeof(ScanConst) * M, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_env_map, h_env_map, sizeof(int32_t) * E, hipMemcpyHostToDevice, h->stream));
    {
        // the scan's agent order: by map slot (stable), so each XCD's contiguous share of the launch touches as few
        // tables as possible — interleaved assignments then cost what grouped ones do (measured, 8 tracks x 65 536
        // agents: interleaved 1.075 ms per step in agent order, grouped 0.738)
        const int A = h->cfg.num_agents;
        std::vector<uint32_t> order((size_t)h->N);
        for (int i = 0; i < h->N; ++i) order[i] = (uint32_t)i;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return h_env_map[a / A] < h_env_map[b / A]; });
        bool identity = true;
        for (int i = 0; i < h->N && identity; ++i) identity = order[i] == (uint32_t)i;
        if (h->d_scan_order) { (void)hipFree(h->d_scan_order); h->d_scan_order = nullptr; }
        if (!identity) {
            HIPCHK(h, hipMalloc(reinterpret_cast<void **>(&h->d_scan_order), sizeof(uint32_t) * h->N));
            HIPCHK(h, hipMemcpyAsync(h->d_scan_order, order.data(), sizeof(uint32_t) * h->N, hipMemcpyHostToDevice, h->stream));
        }
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->dev.maps_full = h->d_maps_full;
    h->dev.env_map = h->d_env_map;
    h->multi_map = true;
    return F110_OK;
}

int f110_get_map_dt(f110_sim *h, double *out)
{
    if (!h || !out) return fail(h, F110_ERR_INVALID, "null argument");
    ENTER(h);
    if (!h->has_map) return fail(h, F110_ERR_NO_MAP, "Map is not set for scan simulator.");
    const size_t row = (size_t)h->k.width * sizeof(double);
    if (h->d_dt_row)
        HIPCHK(h, hipMemcpyAsync(out, h->d_dt_row, (size_t)h->k.height * row, hipMemcpyDeviceToHost, h->stream));
    else   // the table lives inside its padded copy only
        HIPCHK(h, hipMemcpy2DAsync(out, row, h->k.table_rm, (size_t)h->k.row_bytes, row, (size_t)h->k.height, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return F110_OK;
}

int f110_map_shape(f110_sim *h, int32_t *H, int32_t *W)
{
    if (!h) return fail(nullptr, F110_ERR_INVALID, "null handle");
    ENTER(h);
    if (!h->has_map) return fail(h, F110_ERR_NO_MAP, "Map is not set for scan simulator.");
    if (H) *H = h->k.height;
    if (W) *W = h->k.width;
    return F110_OK;
}

int f110_set_trig_tables(f110_sim *h, const double *s, const double *c, int32_t n)
{
    if (!h || !s || !c) return fail(h, F110_ERR_INVALID, "null argument");
    ENTER(h);
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

static const char *const kBeamsMsg = "the beam tables (f110_set_beam_tables) are not the uniform ramp scan_angles[i] = scan_angles[0] + i*inc of base_classes.py:133-134: "
                                     "the step's opponent ray-cast assumes it; such tables are served by f110_raycast_batch / f110_ttc_batch only";

int f110_set_beam_tables(f110_sim *h, const double *sa, const double *co, const double *sd, int32_t B)
{
    if (!h || !sa || !co || !sd) return fail(h, F110_ERR_INVALID, "null argument");
    ENTER(h);
    if (B != h->cfg.num_beams) return fail(h, F110_ERR_INVALID, "beam tables must have num_beams=%d entries (got %d)", h->cfg.num_beams, B);
    HIPCHK(h, hipMemcpyAsync(h->d_scan_angles, sa, sizeof(double) * B, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_beam_cos, co, sizeof(double) * B, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_side, sd, sizeof(double) * B, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    // beam spacing of THIS table (seed of the nearest-beam search in the opponent ray-cast)
    h->dev.angle_inc = (sa[B - 1] - sa[0]) / (B - 1);
    // The step's opponent ray-cast finds a vertex's beam next to a closed-form estimate (nearest_beam: +-3 entries) and culls by a
    // disc — both assume the table is the increasing ramp sa[0] + i*inc that RaceCar builds (base_classes.py:133-134).  A table
    // within a quarter of a beam spacing of that ramp keeps the true argmin inside the searched neighbourhood; anything else
    // (the free functions ray_cast / check_ttc_jit accept ANY array) is served by the unit entry points only, with the
    // reference's full argmin (k_raycast_unit), and refused by the step (beams_must_be_uniform).
    {
        const double inc = h->dev.angle_inc;
        bool uni = B >= 2 && inc > 0.0 && inc == inc;
        for (int i = 0; uni && i < B; ++i) uni = std::fabs(sa[i] - (sa[0] + (double)i * inc)) <= 0.25 * inc;
        h->beams_uniform = uni;
    }
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

int f110_set_params_batch(f110_sim *h, const double *h_params)
{
    if (!h) return fail(nullptr, F110_ERR_INVALID, "null handle");
    ENTER(h);
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    if (!h_params) {   // back to one parameter set per agent slot
        h->dev.params = h->d_params;
        h->dev.params_per_agent = 0;
        return F110_OK;
    }
    const size_t n = (size_t)h->N * NPARAMS;
    if (!h->d_params_all) TRY(dmalloc(h, &h->d_params_all, n));
    HIPCHK(h, hipMemcpyAsync(h->d_params_all, h_params, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->dev.params = h->d_params_all;
    h->dev.params_per_agent = 1;
    return F110_OK;
}

int f110_set_params(f110_sim *h, int32_t agent_idx, const double *p)
{
    if (!h || !p) return fail(h, F110_ERR_INVALID, "null argument");
    ENTER(h);
    if (h->dev.params_per_agent) return fail(h, F110_ERR_STATE, "per-agent parameters are active (f110_set_params_batch); clear them first");
    const int A = h->cfg.num_agents;
    if (agent_idx >= A) return fail(h, F110_ERR_INVALID, "Index given is out of bounds for list of agents.");
    for (int a = 0; a < A; ++a)
        if (agent_idx < 0 || agent_idx == a)
            HIPCHK(h, hipMemcpyAsync(h->d_params + (size_t)a * NPARAMS, p, sizeof(double) * NPARAMS, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return F110_OK;
}

static int noise_cache_extend(f110_sim *h, int upto);
static int noise_adopt(f110_sim *h);

static void noise_release(f110_sim *h)
{
    if (h->noise_stream) (void)hipStreamSynchronize(h->noise_stream);
    if (h->noise_ahead && h->noise_ahead_block && h->noise_ahead_block != h->d_noise) (void)hipFree(h->noise_ahead_block);
    h->noise_ahead = false;
    h->noise_ahead_block = nullptr;
    for (double *p : h->noise_retired) (void)hipFree(p);
    h->noise_retired.clear();
    if (h->d_noise) { (void)hipFree(h->d_noise); h->d_noise = nullptr; }
    if (h->d_rng_state) { (void)hipFree(h->d_rng_state); h->d_rng_state = nullptr; }
    if (h->d_rng_seed) { (void)hipFree(h->d_rng_seed); h->d_rng_seed = nullptr; }
    if (h->d_rng_rowstate) { (void)hipFree(h->d_rng_rowstate); h->d_rng_rowstate = nullptr; }
    h->dev.noise = nullptr;
    h->dev.noise_rows = 0;
    h->dev.noise_rng = 0;
    h->dev.rng_state = nullptr;
    h->dev.rng_seed = nullptr;
    h->dev.rng_rowstate = nullptr;
    h->noise_rows_ready = 0;
    h->noise_rows_alloc = 0;
}

int f110_set_noise_table(f110_sim *h, const double *noise, int32_t rows, int32_t B)
{
    if (!h) return fail(nullptr, F110_ERR_INVALID, "null handle");
    ENTER(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    noise_release(h);
    if (!noise || rows <= 0) return F110_OK;
    if (B != h->cfg.num_beams) return fail(h, F110_ERR_INVALID, "noise table must have num_beams=%d columns (got %d)", h->cfg.num_beams, B);
    TRY(dmalloc(h, &h->d_noise, (size_t)rows * B));
    HIPCHK(h, hipMemcpyAsync(h->d_noise, noise, sizeof(double) * (size_t)rows * B, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->dev.noise = h->d_noise;
    h->dev.noise_rows = rows;
    return F110_OK;
}

int f110_pcg64_seed(uint64_t seed, uint64_t *out_state_inc)
{
    if (!out_state_inc) return fail(nullptr, F110_ERR_INVALID, "null argument");
    pcg64_seed_from_u64(seed, out_state_inc);
    return F110_OK;
}

int f110_set_noise_rng(f110_sim *h, const uint64_t *state_inc, int32_t per_agent, double std_dev, int32_t cache_rows)
{
    if (!h) return fail(nullptr, F110_ERR_INVALID, "null handle");
    ENTER(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    noise_release(h);
    if (!state_inc) return F110_OK;
    if (!(std_dev >= 0.0)) return fail(h, F110_ERR_INVALID, "noise std_dev must be >= 0");
    const int B = h->cfg.num_beams;
    const size_t N = (size_t)h->N;
    h->noise_gen.scale = std_dev;
    TRY(dmalloc(h, &h->d_rng_state, N));
    HIPCHK(h, hipMemsetAsync(h->d_rng_state, 0, sizeof(U128) * N, h->stream));
    h->dev.rng_state = h->d_rng_state;
    if (per_agent) {
        // h_state_inc [N][4]: two U128 {hi, lo} per agent — the layout of rng_seed
        TRY(dmalloc(h, &h->d_rng_seed, 2 * N));
        HIPCHK(h, hipMemcpy(h->d_rng_seed, state_inc, sizeof(U128) * 2 * N, hipMemcpyHostToDevice));
        h->dev.rng_seed = h->d_rng_seed;
        h->dev.noise_rng = 2;
        return F110_OK;
    }
    // capacity: 164 s of simulated time by default.  Rows are generated on demand, and the memory behind
    // them grows on demand too (noise_cache_extend): a short-lived Simulator pays for the rows it uses
    // (8.6 KB each at 1080 beams), not for the capacity (141 MB)
    int rows = cache_rows > 0 ? cache_rows : 16384;
    if ((size_t)rows * B * sizeof(double) > (size_t)1 << 31) rows = (int)(((size_t)1 << 31) / ((size_t)B * sizeof(double)));
    h->noise_rows_alloc = std::min(rows, 512);
    TRY(dmalloc(h, &h->d_noise, (size_t)h->noise_rows_alloc * B));
    TRY(dmalloc(h, &h->d_rng_rowstate, (size_t)rows + 1));
    const U128 st0 = {state_inc[0], state_inc[1]};
    HIPCHK(h, hipMemcpy(h->d_rng_rowstate, &st0, sizeof st0, hipMemcpyHostToDevice));
    h->dev.rng_inc = U128{state_inc[2], state_inc[3]};
    h->dev.rng_rowstate = h->d_rng_rowstate;
    h->dev.noise = h->d_noise;
    h->dev.noise_rows = rows;
    h->dev.noise_rng = 1;
    h->noise_rows_ready = 0;
    return noise_cache_extend(h, std::min(rows, 256));
}

int f110_noise_prepare(f110_sim *h, int32_t rows)
{
    if (!h) return fail(nullptr, F110_ERR_INVALID, "null handle");
    ENTER(h);
    if (h->dev.noise_rng != 1) return F110_OK;
    TRY(noise_adopt(h));
    return noise_cache_extend(h, rows);
}

int f110_noise_rows_batch(f110_sim *h, const uint64_t *state_inc, double std_dev, int32_t rows, int32_t num_beams, double *h_out,
                          uint64_t *h_state_out)
{
    if (!h || !state_inc || !h_out || rows < 1 || num_beams < 1) return fail(h, F110_ERR_INVALID, "f110_noise_rows_batch: bad argument");
    ENTER(h);
    Scratch s(h);
    U128 *d_rs = nullptr;
    double *d_out = nullptr;
    TRY(s.up<U128>(nullptr, (size_t)rows + 1, &d_rs));
    TRY(s.up<double>(nullptr, (size_t)rows * num_beams, &d_out));
    const U128 st0 = {state_inc[0], state_inc[1]}, inc = {state_inc[2], state_inc[3]};
    HIPCHK(h, hipMemcpyAsync(d_rs, &st0, sizeof st0, hipMemcpyHostToDevice, h->stream));
    NoiseGen g = h->noise_gen;
    g.scale = std_dev;
    hipLaunchKernelGGL(k_noise_cache, dim3(1), dim3(64), 0, h->stream, g, inc, d_rs, d_out, 0, rows, num_beams);
    HIPCHK(h, hipGetLastError());
    TRY(s.down(h_out, d_out, (size_t)rows * num_beams));
    U128 last{};
    HIPCHK(h, hipMemcpyAsync(&last, d_rs + rows, sizeof last, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h_state_out) {
        h_state_out[0] = last.hi;
        h_state_out[1] = last.lo;
    }
    return F110_OK;
}

int f110_scan_lookup_count(f110_sim *h, int32_t enable, int64_t *out_total)
{
    if (!h) return fail(nullptr, F110_ERR_INVALID, "null handle");
    ENTER(h);
    if (out_total) {
        unsigned long long v[2] = {0, 0};
        HIPCHK(h, hipMemcpyAsync(v, h->d_lookups, sizeof v, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipMemsetAsync(h->d_lookups, 0, sizeof v, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        out_total[0] = (int64_t)v[0];
        out_total[1] = (int64_t)v[1];
    }
    if (enable >= 0) h->lookups_on = enable != 0;
    return F110_OK;
}

// ---- reset / step ------------------------------------------------------------------------
int f110_reset_device(f110_sim *h, const double *d_poses, const uint8_t *d_env_mask)
{
    if (!h || !d_poses) return fail(h, F110_ERR_INVALID, "null argument");
    ENTER(h);
    if (!d_env_mask) h->noise_ub = 0;   // every agent's step_count is 0 again
    hipLaunchKernelGGL(k_reset, grid1d(h->N, 256), dim3(256), 0, h->stream, h->dev, d_poses, d_env_mask);
    HIPCHK(h, hipGetLastError());
    return F110_OK;
}

int f110_set_auto_reseat(f110_sim *h, const double *d_start_poses, int32_t ego_idx, int32_t *d_count)
{
    if (!h) return fail(nullptr, F110_ERR_INVALID, "null handle");
    ENTER(h);
    if (d_start_poses && (ego_idx < 0 || ego_idx >= h->cfg.num_agents))
        return fail(h, F110_ERR_INVALID, "Index given is out of bounds for list of agents.");
    h->dev.reseat_poses = d_start_poses;
    h->dev.reseat_ego = d_start_poses ? ego_idx : 0;
    h->dev.reseat_count = d_start_poses ? d_count : nullptr;
    return F110_OK;
}

int f110_reset_collided_device(f110_sim *h, const double *d_start_poses, int32_t ego_idx, int32_t *d_count)
{
    if (!h || !d_start_poses) return fail(h, F110_ERR_INVALID, "null argument");
    ENTER(h);
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
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;   // optional: gather to a root
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
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
            api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(dlsym(api.lib, "ncclGroupStart"));
            api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(dlsym(api.lib, "ncclGroupEnd"));
            api.CommCount = reinterpret_cast<decltype(api.CommCount)>(dlsym(api.lib, "ncclCommCount"));
            api.CommUserRank = reinterpret_cast<decltype(api.CommUserRank)>(dlsym(api.lib, "ncclCommUserRank"));
            api.Send = reinterpret_cast<decltype(api.Send)>(dlsym(api.lib, "ncclSend"));
            api.Recv = reinterpret_cast<decltype(api.Recv)>(dlsym(api.lib, "ncclRecv"));
        }
    }
    const bool ok = api.lib && api.GetUniqueId && api.CommInitRank && api.AllGather && api.CommDestroy && api.GetErrorString &&
                    api.GroupStart && api.GroupEnd && api.CommCount && api.CommUserRank;
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
    ENTER(h);
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
    h->comm_rank = rank;
    return F110_OK;
}

int f110_comm_set_overlap(f110_sim *h, int32_t enable)
{
    if (!h) return fail(nullptr, F110_ERR_INVALID, "null handle");
    ENTER(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (enable && !h->comm_overlap) {
        const size_t count = (size_t)h->N * h->cfg.num_beams;
        if (!h->scan_bufs[1]) {
            TRY(dmalloc(h, &h->scan_bufs[1], count));
            HIPCHK(h, hipMemsetAsync(h->scan_bufs[1], 0, sizeof(double) * count, h->stream));
            HIPCHK(h, hipStreamCreateWithFlags(&h->comm_stream, hipStreamNonBlocking));
            HIPCHK(h, hipEventCreateWithFlags(&h->ev_step_done, hipEventDisableTiming));
            HIPCHK(h, hipEventCreateWithFlags(&h->ev_gather_done[0], hipEventDisableTiming));
            HIPCHK(h, hipEventCreateWithFlags(&h->ev_gather_done[1], hipEventDisableTiming));
        }
        h->scan_bufs[0] = h->dev.scans;
        h->scans_cur = 0;
        h->gather_pending[0] = h->gather_pending[1] = false;
        h->comm_swap_next = false;
        h->comm_overlap = true;
    } else if (!enable && h->comm_overlap) {
        if (h->comm_stream) HIPCHK(h, hipStreamSynchronize(h->comm_stream));
        if (h->scans_cur != 0) {   // back on the buffer the handle owns by name
            HIPCHK(h, hipMemcpyAsync(h->scan_bufs[0], h->scan_bufs[1], sizeof(double) * (size_t)h->N * h->cfg.num_beams, hipMemcpyDeviceToDevice, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
            h->scans_cur = 0;
        }
        h->dev.scans = h->scan_bufs[0];
        h->comm_overlap = false;
    }
    return F110_OK;
}

// scans (+ the [7][N] scalar block when d_recv_scal != nullptr) of every rank to every rank.  The two
// ncclAllGather calls are one group (ncclGroupStart / End): RCCL fuses them into one launch per rank.
// transport F110_GATHER_F32: the scans cross the links as float32 (a conversion kernel in front of the collective; the
// receive buffer then holds floats) — half the bytes of the one leg that is link-bound.  root >= 0: only
// that rank receives (grouped ncclSend / ncclRecv: every peer's block rides its one direct xGMI link to the root; the
// other ranks receive nothing, 1 / n_ranks of the all-gather's receive volume per rank).
static int comm_gather(f110_sim *h, void *d_recv_scans, void *d_recv_scal, int transport = F110_GATHER_F64, int root = -1)
{
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    if (!h->comm) return fail(h, F110_ERR_STATE, "f110_comm_init has not been called");
    RcclApi *r = rccl_api();
    if (root >= h->comm_ranks) return fail(h, F110_ERR_INVALID, "f110_comm_gather_obs: root %d of %d ranks", root, h->comm_ranks);
    if (root >= 0 && (!r->Send || !r->Recv)) return fail(h, F110_ERR_STATE, "this RCCL has no ncclSend / ncclRecv: gather to a root is not available");
    const bool f32 = transport == F110_GATHER_F32;
    const bool i_receive = root < 0 || root == h->comm_rank;
    if (i_receive && !d_recv_scans) return fail(h, F110_ERR_INVALID, "null receive buffer on a receiving rank");
    const size_t N = (size_t)h->N, count = N * h->cfg.num_beams, scount = N * kObsScalars;
    const int cur = h->comm_overlap ? h->scans_cur : 0;
    const bool scal = d_recv_scal != nullptr || (root >= 0 && !i_receive && h->comm_send_scalars);
    if (scal && !h->obs_scal[cur]) {
        TRY(dmalloc(h, &h->obs_scal[cur], scount));   // (k_pack_obs writes every element before the gather reads it)
    }
    if (f32 && !h->scan_f32[cur]) TRY(dmalloc(h, &h->scan_f32[cur], count));
    auto gather_on = [&](hipStream_t st, const double *scans) -> int {
        const void *send = scans;
        const ncclDataType_t ty = f32 ? ncclFloat32 : ncclFloat64;
        const size_t esz = f32 ? sizeof(float) : sizeof(double);
        if (f32) {
            hipLaunchKernelGGL(k_scans_to_f32, grid1d(count, 256), dim3(256), 0, st, scans, h->scan_f32[cur], count);
            send = h->scan_f32[cur];
        }
        ncclResult_t rc = r->GroupStart();
        if (root < 0) {
            if (rc == ncclSuccess) rc = r->AllGather(send, d_recv_scans, count, ty, h->comm, st);
            if (rc == ncclSuccess && d_recv_scal) rc = r->AllGather(h->obs_scal[cur], d_recv_scal, scount, ncclFloat64, h->comm, st);
        } else {
            if (rc == ncclSuccess) rc = r->Send(send, count, ty, root, h->comm, st);
            if (rc == ncclSuccess && scal) rc = r->Send(h->obs_scal[cur], scount, ncclFloat64, root, h->comm, st);
            if (i_receive) {
                for (int p = 0; p < h->comm_ranks && rc == ncclSuccess; ++p) {
                    rc = r->Recv(static_cast<char *>(d_recv_scans) + (size_t)p * count * esz, count, ty, p, h->comm, st);
                    if (rc == ncclSuccess && d_recv_scal)
                        rc = r->Recv(static_cast<double *>(d_recv_scal) + (size_t)p * scount, scount, ncclFloat64, p, h->comm, st);
                }
            }
        }
        const ncclResult_t re = r->GroupEnd();
        if (rc == ncclSuccess) rc = re;
        if (rc != ncclSuccess) return fail(h, F110_ERR_HIP, "RCCL gather failed: %s", r->GetErrorString(rc));
        return F110_OK;
    };
    if (!h->comm_overlap) {
        ENTER(h);
        if (scal) hipLaunchKernelGGL(k_pack_obs, grid1d(N, 256), dim3(256), 0, h->stream, h->dev, h->obs_scal[0]);
        return gather_on(h->stream, h->dev.scans);
    }
    // overlapped: the gather waits for the step that produced this buffer and runs beside the next one,
    // which writes the other buffer; the step after that waits for this gather before reusing the buffer
    {
        const bool inflight = h->comm_inflight;   // join the groups, not earlier gathers (they stay asynchronous)
        h->comm_inflight = false;
        const int rj = join_groups(h);
        h->comm_inflight = inflight;
        if (rj != F110_OK) return rj;
    }
    // the scalar block is packed on the main stream, behind the step and before anything (a re-seat, the
    // next step) can change the state it reads; obs_scal[cur] was last read by the gather of two steps ago,
    // which the step that just ran has waited for
    if (scal) {
        hipLaunchKernelGGL(k_pack_obs, grid1d(N, 256), dim3(256), 0, h->stream, h->dev, h->obs_scal[cur]);
        // k_pack_obs READS state[] on the main stream: a next step that goes out as two env blocks must fork from it
        // (its second block's k_integrate would otherwise overwrite the state under the pack) — ADVICE r4
        h->main_dirty = true;
    }
    HIPCHK(h, hipEventRecord(h->ev_step_done, h->stream));
    HIPCHK(h, hipStreamWaitEvent(h->comm_stream, h->ev_step_done, 0));
    TRY(gather_on(h->comm_stream, h->scan_bufs[cur]));
    HIPCHK(h, hipEventRecord(h->ev_gather_done[cur], h->comm_stream));
    h->gather_pending[cur] = true;
    h->comm_swap_next = true;
    h->comm_inflight = true;
    return F110_OK;
}

int f110_comm_all_gather_scans(f110_sim *h, void *d_recv)
{
    if (!h || !d_recv) return fail(h, F110_ERR_INVALID, "null argument");
    return comm_gather(h, d_recv, nullptr);
}

int f110_comm_all_gather_obs(f110_sim *h, void *d_recv_scans, void *d_recv_scalars)
{
    if (!h || !d_recv_scans || !d_recv_scalars) return fail(h, F110_ERR_INVALID, "null argument");
    return comm_gather(h, d_recv_scans, d_recv_scalars);
}

int f110_comm_gather_obs(f110_sim *h, void *d_recv_scans, void *d_recv_scalars, int32_t transport, int32_t root)
{
    if (!h) return fail(h, F110_ERR_INVALID, "null handle");
    if (transport != F110_GATHER_F64 && transport != F110_GATHER_F32) return fail(h, F110_ERR_INVALID, "f110_comm_gather_obs: unknown transport %d", transport);
    // a rank that is not the root sends its scalar block exactly when the root asked for one: the ranks agree on that
    // through this flag's convention — a sender passes a non-null d_recv_scalars (never written) to say "with scalars"
    h->comm_send_scalars = d_recv_scalars != nullptr;
    return comm_gather(h, d_recv_scans, d_recv_scalars, transport, root);
}

int f110_comm_info(f110_sim *h, int32_t *n_ranks, int32_t *rank)
{
    if (!h) return fail(nullptr, F110_ERR_INVALID, "null handle");
    if (!h->comm) return fail(h, F110_ERR_STATE, "f110_comm_init has not been called");
    RcclApi *r = rccl_api();
    int n = 0, me = -1;
    ncclResult_t rc = r->CommCount(h->comm, &n);
    if (rc == ncclSuccess) rc = r->CommUserRank(h->comm, &me);
    if (rc != ncclSuccess) return fail(h, F110_ERR_HIP, "ncclCommCount / ncclCommUserRank failed: %s", r->GetErrorString(rc));
    if (n_ranks) *n_ranks = n;
    if (rank) *rank = me;
    return F110_OK;
}

int f110_comm_destroy(f110_sim *h)
{
    if (!h) return fail(nullptr, F110_ERR_INVALID, "null handle");
    ENTER(h);
    if (h->comm) {
        if (h->comm_stream) HIPCHK(h, hipStreamSynchronize(h->comm_stream));
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
    ENTER(h);
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
    ENTER(h);
    if (!h->has_episode) return fail(h, F110_ERR_STATE, "f110_episode_init has not been called");
    HIPCHK(h, hipMemcpyAsync(h->d_poses, poses, sizeof(double) * 3 * h->N, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_rot_stage, rot, sizeof(double) * 4 * h->cfg.num_envs, hipMemcpyHostToDevice, h->stream));
    if (env_mask) HIPCHK(h, hipMemcpyAsync(h->d_mask, env_mask, (size_t)h->cfg.num_envs, hipMemcpyHostToDevice, h->stream));
    const uint8_t *dm = env_mask ? h->d_mask : nullptr;
    if (!dm) h->noise_ub = 0;
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
    if (h->last_blocks == 2 && h->groups_busy) {
        // the step went out as two env blocks: each block's _check_done behind it on the block's own stream (an env's
        // lap bookkeeping reads that env's agents only), no join — the device-resident loop stays two independent halves
        const int per = group_envs(h), E = h->cfg.num_envs;
        for (int g = 0; g < 2; ++g) {
            const int e0 = g * per, e1 = std::min(E, e0 + per);
            if (e0 >= e1) break;
            hipLaunchKernelGGL(k_episode, grid1d(e1 - e0, 256), dim3(256), 0, h->gstreams[g], h->dev, h->ep, e1 - e0, e0);
        }
        HIPCHK(h, hipGetLastError());
        return F110_OK;
    }
    ENTER(h);   // _check_done reads every group's poses and flags
    hipLaunchKernelGGL(k_episode, grid1d(h->cfg.num_envs, 256), dim3(256), 0, h->stream, h->dev, h->ep, h->cfg.num_envs);
    HIPCHK(h, hipGetLastError());
    h->touched = false;   // the next step may split: what this call left on the main stream is forked from (main_dirty)
    return F110_OK;
}

int f110_episode_reset_done_device(f110_sim *h, int32_t *d_count)
{
    if (!h) return fail(nullptr, F110_ERR_INVALID, "null handle");
    if (!h->has_episode) return fail(h, F110_ERR_STATE, "f110_episode_init has not been called");
    if (h->last_blocks == 2 && h->groups_busy && !h->touched) {   // behind a two-block f110_episode_step_device: per block, no join
        HIPCHK(h, hipSetDevice(h->cfg.device_id));
        const int per = group_envs(h), E = h->cfg.num_envs, A = h->cfg.num_agents;
        for (int g = 0; g < 2; ++g) {
            const int e0 = g * per, e1 = std::min(E, e0 + per);
            if (e0 >= e1) break;
            hipLaunchKernelGGL(k_episode_reset_done, grid1d((e1 - e0) * A, 256), dim3(256), 0, h->gstreams[g], h->dev, h->ep, d_count, e0 * A, (e1 - e0) * A);
            hipLaunchKernelGGL(k_episode_clear_done, grid1d(e1 - e0, 256), dim3(256), 0, h->gstreams[g], h->ep, e1 - e0, e0);
        }
        HIPCHK(h, hipGetLastError());
        return F110_OK;
    }
    ENTER(h);
    hipLaunchKernelGGL(k_episode_reset_done, grid1d(h->N, 256), dim3(256), 0, h->stream, h->dev, h->ep, d_count);
    hipLaunchKernelGGL(k_episode_clear_done, grid1d(h->cfg.num_envs, 256), dim3(256), 0, h->stream, h->ep, h->cfg.num_envs);
    HIPCHK(h, hipGetLastError());
    h->touched = false;   // (as in f110_episode_step_device: the loop step / reset_done / step / ... may split from its second round on)
    return F110_OK;
}

size_t f110_episode_packed_bytes(const f110_sim *h)
{
    if (!h) return 0;
    const size_t N = (size_t)h->N, E = (size_t)h->cfg.num_envs;
    return (9 * N + E) * sizeof(double) + 2 * N + E;
}

int f110_host_alloc(f110_sim *h, size_t bytes, void **out)
{
    if (!h || !out) return fail(h, F110_ERR_INVALID, "null argument");
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    HIPCHK(h, hipHostMalloc(out, bytes > 0 ? bytes : 8, hipHostMallocDefault));
    std::lock_guard<std::mutex> lk(g_registry_mu);
    g_host_blocks[static_cast<const char *>(*out)] = bytes > 0 ? bytes : 8;
    return F110_OK;
}

int f110_host_free(f110_sim *h, void *p)
{
    if (!p) return F110_OK;
    if (h) {   // the handle's stream may still be copying from / into the block
        ENTER(h);
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    {
        // every live handle forgets what it cached about this block; a handle with an F110_STEP_NO_SYNC step still storing
        // into it is drained first (the caller may pass h = NULL: the block outlives the handle that allocated it)
        std::lock_guard<std::mutex> lk(g_registry_mu);
        const char *lo = static_cast<const char *>(p);
        const auto it = g_host_blocks.find(lo);
        const char *hi = lo + (it != g_host_blocks.end() ? it->second : 1);
        auto inside = [&](const void *q) { return q && static_cast<const char *>(q) >= lo && static_cast<const char *>(q) < hi; };
        for (f110_sim *o : g_handles) {
            if (!o->hb_valid && !o->fused_valid) continue;
            const f110_host_block &b = o->hb_host;
            const void *ptrs[] = {b.scans, b.state, b.agent_poses, b.collisions, b.collision_idx, b.in_collision, b.lap_times, b.lap_counts, b.toggles,
                                  b.current_time, b.near_starts, b.done, b.checkpoint_done, o->hb_actions_host};
            bool hit = false;
            for (const void *q : ptrs) hit = hit || inside(q);
            if (!hit) continue;
            if (o != h) {
                (void)hipSetDevice(o->cfg.device_id);
                (void)hipStreamSynchronize(o->stream);
            }
            o->hb_valid = false;
            o->fused_valid = false;
        }
        if (it != g_host_blocks.end()) g_host_blocks.erase(it);
    }
    HIPCHK(h, hipHostFree(p));
    return F110_OK;
}

int f110_episode_step_host(f110_sim *h, const double *h_actions, int32_t auto_reset, void *h_packed)
{
    if (!h || !h_actions || !h_packed) return fail(h, F110_ERR_INVALID, "null argument");
    if (!h->has_episode) return fail(h, F110_ERR_STATE, "f110_episode_init has not been called");
    if (!h->has_map) return fail(h, F110_ERR_NO_MAP, "Map is not set for scan simulator.");
    if (!h->beams_uniform) return fail(h, F110_ERR_STATE, kBeamsMsg);
    ENTER(h);
    const size_t N = (size_t)h->N, E = (size_t)h->cfg.num_envs, bytes = f110_episode_packed_bytes(h);
    if (!h->d_packed) HIPCHK(h, hipMalloc(&h->d_packed, bytes));
    HIPCHK(h, hipMemcpyAsync(h->d_actions, h_actions, sizeof(double) * 2 * N, hipMemcpyHostToDevice, h->stream));
    TRY(f110_step_device(h, h->d_actions));
    ENTER(h);
    hipLaunchKernelGGL(k_episode, grid1d(E, 256), dim3(256), 0, h->stream, h->dev, h->ep, (int)E);
    double *cols = reinterpret_cast<double *>(h->d_packed);
    uint8_t *flags = reinterpret_cast<uint8_t *>(cols + 9 * N + E);
    hipLaunchKernelGGL(k_pack_episode, grid1d(N > E ? N : E, 256), dim3(256), 0, h->stream, h->dev, h->ep, (int)E, cols, flags);
    HIPCHK(h, hipMemcpyAsync(h_packed, h->d_packed, bytes, hipMemcpyDeviceToHost, h->stream));
    if (auto_reset) {   // the packed block holds the terminal observation; the re-seat follows it
        hipLaunchKernelGGL(k_episode_reset_done, grid1d(N, 256), dim3(256), 0, h->stream, h->dev, h->ep, (int32_t *)nullptr);
        hipLaunchKernelGGL(k_episode_clear_done, grid1d(E, 256), dim3(256), 0, h->stream, h->ep, (int)E);
    }
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return F110_OK;
}

// One entry per env.step() of a host-driven loop (f110.h).  Every pointer of `out` and — with
// F110_STEP_ACTIONS_MAPPED — h_actions must be page-locked memory of f110_host_alloc: the kernels read / write
// it in place.  The (struct, actions) pair is validated once (hipHostGetDevicePointer) and remembered.
constexpr size_t kScansByKernelBytes = 8u << 20;   // scans up to this size ride in the host-block kernel instead of a DMA copy

static int map_host_ptr(f110_sim *h, const void *host, void **dev, const char *what)
{
    *dev = nullptr;
    if (!host) return F110_OK;
    if (hipHostGetDevicePointer(dev, const_cast<void *>(host), 0) != hipSuccess || !*dev) {
        (void)hipGetLastError();
        return fail(h, F110_ERR_INVALID, "f110_step_host: %s is not page-locked memory of f110_host_alloc", what);
    }
    return F110_OK;
}

int f110_step_host(f110_sim *h, const double *h_actions, const f110_host_block *out, int32_t flags)
{
    if (!h || !h_actions || !out) return fail(h, F110_ERR_INVALID, "null argument");
    if (!h->has_map) return fail(h, F110_ERR_NO_MAP, "Map is not set for scan simulator.");
    if (!h->beams_uniform) return fail(h, F110_ERR_STATE, kBeamsMsg);
    const bool episode = h->has_episode;
    if (!episode && (out->lap_times || out->lap_counts || out->toggles || out->current_time || out->near_starts ||
                     out->checkpoint_done || out->done || (flags & F110_STEP_AUTO_RESET)))
        return fail(h, F110_ERR_STATE, "f110_episode_init has not been called");
    ENTER(h);
    const size_t N = (size_t)h->N, E = (size_t)h->cfg.num_envs;
    const bool mapped_actions = (flags & F110_STEP_ACTIONS_MAPPED) != 0;
    if (!h->hb_valid || std::memcmp(&h->hb_host, out, sizeof *out) != 0 || h->hb_actions_host != (mapped_actions ? h_actions : nullptr)) {
        HostBlock d{};
        void *p = nullptr;
#define MAP_FIELD(dst, src, name)                     \
    do {                                              \
        TRY(map_host_ptr(h, out->src, &p, name));     \
        d.dst = reinterpret_cast<decltype(d.dst)>(p); \
    } while (0)
        MAP_FIELD(state, state, "state");
        MAP_FIELD(collisions, collisions, "collisions");
        MAP_FIELD(collision_idx, collision_idx, "collision_idx");
        MAP_FIELD(agent_poses, agent_poses, "agent_poses");
        MAP_FIELD(lap_time, lap_times, "lap_times");
        MAP_FIELD(lap_count, lap_counts, "lap_counts");
        MAP_FIELD(toggle, toggles, "toggles");
        MAP_FIELD(current_time, current_time, "current_time");
        MAP_FIELD(in_collision, in_collision, "in_collision");
        MAP_FIELD(near_start, near_starts, "near_starts");
        MAP_FIELD(checkpoint, checkpoint_done, "checkpoint_done");
        MAP_FIELD(done, done, "done");
#undef MAP_FIELD
        // scans of a small batch: stored by the kernel too (needs page-locked memory; else, and for big batches, the DMA copy)
        d.scans = nullptr;
        d.num_beams = h->cfg.num_beams;
        h->hb_scans_by_kernel = false;
        if (out->scans && N * (size_t)h->cfg.num_beams * sizeof(double) <= kScansByKernelBytes) {
            void *ps = nullptr;
            if (hipHostGetDevicePointer(&ps, out->scans, 0) == hipSuccess && ps) {
                d.scans = reinterpret_cast<double *>(ps);
                h->hb_scans_by_kernel = true;
            } else {
                (void)hipGetLastError();
            }
        }
        TRY(map_host_ptr(h, mapped_actions ? h_actions : nullptr, &p, "h_actions"));
        h->hb_actions_dev = reinterpret_cast<const double *>(p);
        h->hb_actions_host = mapped_actions ? h_actions : nullptr;
        h->hb_dev = d;
        h->hb_host = *out;
        h->hb_valid = true;
    }
    const auto t_in = std::chrono::steady_clock::now();
    // a tiny batch is one launch whose last workgroup writes the block (k_step_tiny): its completion is ONE word stored by one
    // workgroup, so the host always waits on that word (no runtime call on the way out)
    const bool tiny = tiny_applies(h) && !(flags & (F110_STEP_NO_FUSE | F110_STEP_NO_SYNC)) && !h->dev.reseat_poses;
    const bool spin = ((flags & F110_STEP_SPIN_WAIT) || tiny) && !(flags & F110_STEP_NO_SYNC) && (!out->scans || (h->hb_valid && h->hb_scans_by_kernel));
    if (spin && !h->hb_seq_host) {
        HIPCHK(h, hipHostMalloc(reinterpret_cast<void **>(&h->hb_seq_host), 64, hipHostMallocDefault));
        *h->hb_seq_host = 0;
        TRY(dmalloc(h, &h->hb_blocks_done, 1));
        HIPCHK(h, hipMemsetAsync(h->hb_blocks_done, 0, sizeof(unsigned int), h->stream));
    }
    const double *d_act = h->d_actions;
    if (mapped_actions)
        d_act = h->hb_actions_dev;   // k_integrate reads the [N][2] block over PCIe, once, coalesced
    else if (!tiny)                  // (k_step_tiny gets the actions as kernel arguments)
        HIPCHK(h, hipMemcpyAsync(h->d_actions, h_actions, sizeof(double) * 2 * N, hipMemcpyHostToDevice, h->stream));
    h->tiny_actions_host = tiny ? h_actions : nullptr;
    const int A = h->cfg.num_agents;
    HostBlock hbk = h->hb_dev;
    if (spin) {
        void *p = nullptr;
        TRY(map_host_ptr(h, h->hb_seq_host, &p, "completion word"));
        hbk.seq_host = reinterpret_cast<unsigned long long *>(p);
        hbk.blocks_done = h->hb_blocks_done;
        hbk.seq = ++h->hb_seq;
    }
    // A = 2: the pair kernel can carry the host block and the episode logic as its epilogue (one launch and one drain
    // less per step); its parameters live in device memory and are rewritten only when they change
    // (not with F110_STEP_SPIN_WAIT: the completion word needs a system-scope release per workgroup, and the pair kernel
    // has N / 32 of them with the scan's dirty lines still in L2 — measured 0.657 -> 0.767 ms at 32 768 envs)
    const bool want_fuse = A == 2 && !(flags & F110_STEP_NO_FUSE) && (!spin || tiny);
    h->tiny_request = tiny;
    h->tiny_host_request = tiny && A == 1;
    if (h->tiny_host_request) {
        h->tiny_hb = hbk;
        h->tiny_episode = episode ? 1 : 0;
        h->tiny_auto_reset = (flags & F110_STEP_AUTO_RESET) ? 1 : 0;
    }
    if (want_fuse) {
        FusedHost fh{};
        if (episode) fh.ep = h->ep;
        fh.hb = hbk;
        fh.hb.seq = 0;   // (the sequence number travels in the kernel arguments)
        fh.episode = episode ? 1 : 0;
        fh.auto_reset = (flags & F110_STEP_AUTO_RESET) ? 1 : 0;
        if (!h->d_fused) HIPCHK(h, hipMalloc(reinterpret_cast<void **>(&h->d_fused), sizeof(FusedHost)));
        if (!h->fused_valid || std::memcmp(&fh, &h->fused_host_copy, sizeof fh) != 0) {
            h->fused_host_copy = fh;
            HIPCHK(h, hipMemcpyAsync(h->d_fused, &h->fused_host_copy, sizeof fh, hipMemcpyHostToDevice, h->stream));
            h->fused_valid = true;
        }
    }
    h->fuse_request = want_fuse;
    h->fuse_seq = spin ? hbk.seq : 0;
    h->fused_done = false;
    const int rc_step = f110_step_device(h, d_act);
    h->fuse_request = false;
    h->tiny_request = h->tiny_host_request = false;
    h->tiny_actions_host = nullptr;
    if (rc_step != F110_OK) return rc_step;
    ENTER(h);
    if (!h->fused_done) {
        const int epb = A >= 256 ? 1 : 256 / A;
        hipLaunchKernelGGL(k_host_block, dim3((unsigned)((E + epb - 1) / epb)), dim3(256), 0, h->stream, h->dev, h->ep, hbk, (int)E, epb,
                           episode ? 1 : 0, (flags & F110_STEP_AUTO_RESET) ? 1 : 0);
    }
    // the scans are contiguous in HBM already: a DMA copy, behind the kernel (the re-seat leaves scans alone)
    if (out->scans && !h->hb_scans_by_kernel) HIPCHK(h, hipMemcpyAsync(out->scans, h->dev.scans, sizeof(double) * N * (size_t)h->cfg.num_beams, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipGetLastError());
    const auto t_enq = std::chrono::steady_clock::now();
    if (spin) {
        // poll the completion word the last workgroup stores (no runtime call on the way out); a kernel that
        // never signals (a device fault) is caught by the runtime after the spin budget
        const unsigned long long want = h->hb_seq;
        const volatile unsigned long long *w = h->hb_seq_host;
        bool seen = false;
        if (kExperimental && h->exp.tiny_start_probe && tiny && h->tiny.start_word) {
            for (long it = 0; it < 200000000L && __atomic_load_n(w + 1, __ATOMIC_ACQUIRE) != want; ++it) __builtin_ia32_pause();
            const auto t_start = std::chrono::steady_clock::now();
            for (long it = 0; it < 200000000L && __atomic_load_n(w, __ATOMIC_ACQUIRE) != want; ++it) __builtin_ia32_pause();
            const auto t_done = std::chrono::steady_clock::now();
            h->exp.probe_calls += 1;
            h->exp.probe_us[0] += std::chrono::duration<double, std::micro>(t_enq - t_in).count();
            h->exp.probe_us[1] += std::chrono::duration<double, std::micro>(t_start - t_enq).count();
            h->exp.probe_us[2] += std::chrono::duration<double, std::micro>(t_done - t_start).count();
        }
        for (long it = 0; it < 2000000000L; ++it) {
            if (__atomic_load_n(w, __ATOMIC_ACQUIRE) == want) {
                seen = true;
                break;
            }
            __builtin_ia32_pause();
            if ((it & 0xfffff) == 0xfffff &&
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t_enq).count() > 5.0) break;
        }
        if (!seen) HIPCHK(h, hipStreamSynchronize(h->stream));
    } else if (!(flags & F110_STEP_NO_SYNC)) {
        if (flags & F110_STEP_POLL) {
            // hipStreamSynchronize wakes up in coarse quanta once a wait has lasted ~30 us (measured: a 40 us step is
            // reported after 67 us); polling the stream costs a core for the step's duration and returns within ~1 us
            // ... so the poll is BOUNDED: a step that has not finished after kPollBudgetUs hands the core back and
            // sleeps in hipStreamSynchronize — the coarse wake-up is then a few per cent of a long wait, and ranks that share a
            // CPU quota (8 ranks on 16 cores on the bench box) do not spin against their own policy threads
            constexpr double kPollBudgetUs = 250.0;
            hipError_t q;
            uint32_t it = 0;
            while ((q = hipStreamQuery(h->stream)) == hipErrorNotReady) {
                __builtin_ia32_pause();
                if ((++it & 0x3fu) == 0u && std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_enq).count() > kPollBudgetUs) break;
            }
            if (q == hipErrorNotReady) HIPCHK(h, hipStreamSynchronize(h->stream));
            else if (q != hipSuccess) return fail(h, F110_ERR_HIP, "hipStreamQuery failed: %s", hipGetErrorString(q));
        } else {
            HIPCHK(h, hipStreamSynchronize(h->stream));
        }
    }
    const auto t_out = std::chrono::steady_clock::now();
    h->hs_enqueue_us += std::chrono::duration<double, std::micro>(t_enq - t_in).count();
    h->hs_wait_us += std::chrono::duration<double, std::micro>(t_out - t_enq).count();
    h->hs_calls += 1;
    return F110_OK;
}

int f110_step_host_stats(f110_sim *h, double *out3)
{
    if (!h || !out3) return fail(h, F110_ERR_INVALID, "null argument");
    out3[0] = (double)h->hs_calls;
    out3[1] = h->hs_enqueue_us;
    out3[2] = h->hs_wait_us;
    h->hs_calls = 0;
    h->hs_enqueue_us = h->hs_wait_us = 0;
    return F110_OK;
}

int f110_episode_get(f110_sim *h, const f110_episode_host *o)
{
    if (!h || !o) return fail(h, F110_ERR_INVALID, "null argument");
    ENTER(h);
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
    ENTER(h);
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
    ENTER(h);
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

// rows [noise_rows_ready, upto) of the shared stream into the row cache (stream-ordered on the main stream)
static int noise_cache_extend(f110_sim *h, int upto)
{
    const int cap = h->dev.noise_rows;
    if (upto > cap) upto = cap;
    if (upto <= h->noise_rows_ready) return F110_OK;
    if (upto > h->noise_rows_alloc) {
        // grow the cache's memory (doubling, capped): the rows generated so far move to the new block
        const size_t B = (size_t)h->cfg.num_beams;
        const int want = std::min(cap, std::max(2 * h->noise_rows_alloc, upto));
        double *bigger = nullptr;
        TRY(dmalloc(h, &bigger, (size_t)want * B));
        HIPCHK(h, hipMemcpyAsync(bigger, h->d_noise, sizeof(double) * (size_t)h->noise_rows_ready * B, hipMemcpyDeviceToDevice, h->stream));
        h->noise_retired.push_back(h->d_noise);   // earlier steps (env blocks on their own streams) may still read the old block
        h->d_noise = bigger;
        h->dev.noise = bigger;
        h->noise_rows_alloc = want;
    }
    hipLaunchKernelGGL(k_noise_cache, dim3(1), dim3(64), 0, h->stream, h->noise_gen, h->dev.rng_inc, h->d_rng_rowstate, h->d_noise, h->noise_rows_ready, upto,
                       h->cfg.num_beams);
    HIPCHK(h, hipGetLastError());
    h->noise_rows_ready = upto;
    h->main_dirty = true;
    return F110_OK;
}

// One step of the agents [begin, begin + count) (an env-aligned block) on stream `st`.
// Product build, one dispatch per (agents per env, beams) case:
//   A = 1                      k_integrate -> scan -> k_finalize_solo
//   A = 2                      k_integrate -> scan -> k_finalize_pair_roles    (pair test + window inside the last kernel)
//   A = 3 .. 16                k_integrate -> scan -> k_finalize_multi         (the same for every ordered pair of an env)
//   A = 17 .. 256              k_integrate -> scan -> k_finalize_multi_tiled   (the env's ordered pairs in tiles of 256 records)
//   A > 256                    k_integrate -> { scan || k_collide on the side stream } -> k_finalize
// and the scan kernel by table / beam count: k_scan_rays_agent (PADDED table; longest-first order for small
// batches), k_scan_dirs_agent (more beams than table directions), k_scan_rays (row-major table, few beams).
// The experimental build adds collide_mode 1 (pair tests fused into k_integrate) / 2 (k_collide in line), the
// other layouts' kernels, two-pass dedupe and the forced geometries of f110_exp_set.
// ev[0..3] (or nullptr): profiling events before integrate / before scan / after scan / after finalize.
enum ScanKind { SCAN_FLAT, SCAN_AGENT, SCAN_AGENT_SCHED, SCAN_DIRS };


static ScanKind pick_scan(const f110_sim *h, int begin, int count)
{
    const bool padded = padded_family(h->cfg.map_layout) && h->k.pad;
    if (h->dir_stride > 0) return padded ? SCAN_DIRS : SCAN_FLAT;   // (no PADDED table: every beam is marched)
    if (!(h->multi_map || agent_aligned(h))) return SCAN_FLAT;
    if (h->task_order && !h->multi_map && !h->lookups_on && begin == 0 && count == h->N && !h->exp.scan_env_counter) return SCAN_AGENT_SCHED;
    return SCAN_AGENT;
}

// The next doubling of the row cache, started on the noise stream when HALF of the rows there are have been used: generation (12.8 us a
// row) is faster than any step loop uses rows (a step of one env takes 30 us and more), so the rows are there when the episode reaches
// them and no step waits.  The rows so far are immutable; a cache that outgrows its memory continues in a new block (old rows copied
// on the same stream), which becomes THE cache at adoption — kernels get the block's address with every launch.
static int noise_start_ahead(f110_sim *h)
{
    const int cap = h->dev.noise_rows, ready = h->noise_rows_ready;
    const int upto = (int)std::min<long long>(cap, 2LL * ready);
    if (h->noise_ahead || upto <= ready) return F110_OK;
    if (!h->noise_stream) {
        HIPCHK(h, hipStreamCreateWithFlags(&h->noise_stream, hipStreamNonBlocking));
        HIPCHK(h, hipEventCreateWithFlags(&h->ev_noise, hipEventDisableTiming));
        HIPCHK(h, hipEventCreateWithFlags(&h->ev_noise_src, hipEventDisableTiming));
    }
    const size_t B = (size_t)h->cfg.num_beams;
    // behind whatever produced the rows so far (the main stream: f110_set_noise_rng, a step that could not wait; or this stream)
    HIPCHK(h, hipEventRecord(h->ev_noise_src, h->stream));
    HIPCHK(h, hipStreamWaitEvent(h->noise_stream, h->ev_noise_src, 0));
    double *block = h->d_noise;
    int alloc = h->noise_rows_alloc;
    if (upto > alloc) {
        alloc = std::min(cap, std::max(2 * alloc, upto));
        block = nullptr;
        TRY(dmalloc(h, &block, (size_t)alloc * B));
        HIPCHK(h, hipMemcpyAsync(block, h->d_noise, sizeof(double) * (size_t)ready * B, hipMemcpyDeviceToDevice, h->noise_stream));
    }
    hipLaunchKernelGGL(k_noise_cache, dim3(1), dim3(64), 0, h->noise_stream, h->noise_gen, h->dev.rng_inc, h->d_rng_rowstate, block, ready, upto, h->cfg.num_beams);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipEventRecord(h->ev_noise, h->noise_stream));
    h->noise_ahead = true;
    h->noise_ahead_upto = upto;
    h->noise_ahead_alloc = alloc;
    h->noise_ahead_block = block;
    return F110_OK;
}

// ... and taken over by the main stream (which waits for them, if they are not there yet)
static int noise_adopt(f110_sim *h)
{
    if (!h->noise_ahead) return F110_OK;
    HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_noise, 0));
    if (h->noise_ahead_block != h->d_noise) {
        h->noise_retired.push_back(h->d_noise);
        h->d_noise = h->noise_ahead_block;
        h->dev.noise = h->d_noise;
        h->noise_rows_alloc = h->noise_ahead_alloc;
    }
    h->noise_rows_ready = h->noise_ahead_upto;
    h->noise_ahead = false;
    h->noise_ahead_block = nullptr;
    h->main_dirty = true;
    return F110_OK;
}

// ---- the whole step of a tiny batch as ONE launch (k_step_tiny) ----------------------------------------------------------
// When: a HOST-SYNCHRONISED step (f110_step_host that waits: F110Env.step, F110VecEnv.step) of at most kTinyMaxAgents = 4 agents, one or
// two per env — the reference's own shape — with the agent-aligned scan on one PADDED map, this step's noise row in the table / row cache
// (or no noise), nothing that brackets or counts inside the step.  Measured where it was tried beyond that (profiles/r06_tiny_ab.txt, one
// launch against three kernels): F110Env with 2 cars 73.5 -> 69.5 us per step, with 1 car 69.3 -> 58.4 us; but 16 agents 79 -> 99 us and 64
// agents 86 -> 197 us (the last workgroup finalizes every agent by itself), and a device-resident loop of one env 53.1 -> 56.4 us (nothing
// waits between its steps, so the longer single kernel is all it sees).  Everything else takes the per-kernel form.
static bool tiny_applies(const f110_sim *h)
{
    const int A = h->cfg.num_agents;
    if (h->tiny_off || h->N > kTinyMaxAgents || (A != 1 && A != 2)) return false;
    if (A == 2 && h->collide_mode != 3) return false;
    if (h->multi_map || !agent_aligned(h) || h->lookups_on || h->path_stats_on || h->profiling) return false;
    if (h->dev.noise_rng && (h->dev.noise_rng == 2 || h->noise_ub >= (long long)h->dev.noise_rows)) return false;   // k_noise_rows would have to run first
    if (kExperimental && (h->exp.scan_env_counter || h->exp.scan_trace || h->exp.scan_occupancy)) return false;
    if (!h->groups_auto && h->groups > 1) return false;   // step_groups = 2 forced: the caller asked for env blocks
    return true;
}

static int tiny_setup(f110_sim *h)
{
    if (h->tiny_mem) return F110_OK;
    const size_t N = (size_t)h->N;
    // one allocation: [done + pad | wall N | buf_cnt N] int32 then the double columns, then the headers
    const size_t ints = 16 + 2 * N, dbls = (7 + 2 + 3 + 3 + 1) * N;
    const size_t bytes = ((ints * 4 + 63) / 64) * 64 + dbls * 8 + N * sizeof(RayHdr);
    HIPCHK(h, hipMalloc(&h->tiny_mem, bytes));
    HIPCHK(h, hipMemsetAsync(h->tiny_mem, 0, bytes, h->stream));
    char *base = static_cast<char *>(h->tiny_mem);
    TinyCtl &t = h->tiny;
    t.done = reinterpret_cast<unsigned int *>(base);
    t.wall = reinterpret_cast<int32_t *>(base) + 16;
    t.buf_cnt = t.wall + N;
    double *d = reinterpret_cast<double *>(base + ((ints * 4 + 63) / 64) * 64);
    t.state = d;            d += 7 * N;
    t.steer_buf = d;        d += 2 * N;
    t.scan_pose = d;        d += 3 * N;
    t.snap_pose = d;        d += 3 * N;
    t.dir_start = d;        d += N;
    t.ray_hdr = reinterpret_cast<RayHdr *>(d);
    t.tasks_per_agent = ((uint32_t)h->k.num_beams + 63u) / 64u;
    return F110_OK;
}

static int step_tiny(f110_sim *h, hipStream_t st, const double *d_actions)
{
    TRY(tiny_setup(h));
    const int N = h->N, A = h->cfg.num_agents;
    AgentArrays dev = h->dev;
    dev.agent_begin = 0;
    dev.agent_count = N;
    dev.sched_count_zero = nullptr;
    RayJob j{};
    j.noise = h->dev.noise;
    j.ttc_side_max = h->ttc_side_max;
    j.ttc_k = h->dev.ttc_thresh * (1.0 + 1e-9) * h->ttc_cos_max;
    j.beam_cos = h->dev.beam_cos;
    j.side_dist = h->dev.side_dist;
    j.ttc_thresh = h->dev.ttc_thresh;
    j.k_cold = cold_consts(h);
    if (!j.k_cold) return fail(h, F110_ERR_HIP, "f110_step_device: constant upload failed");
    h->tiny.tasks_per_agent = ((uint32_t)h->k.num_beams + 63u) / 64u;
    h->tiny.trace = kExperimental ? reinterpret_cast<unsigned long long *>(h->exp.tiny_trace) : nullptr;
    h->tiny.act_inline = 0;
    if (h->tiny_actions_host && N <= kTinyMaxAgents) {   // f110_step_host: the caller's actions travel with the launch
        std::memcpy(h->tiny.act, h->tiny_actions_host, sizeof(double) * 2 * (size_t)N);
        h->tiny.act_inline = 1;
    }
    h->tiny.start_word = nullptr;
    h->tiny.skip = kExperimental ? (h->exp.tiny_start_probe == 2 ? 1 : (h->exp.tiny_general_tail ? 2 : 0)) : 0;
    if (kExperimental && h->exp.tiny_start_probe && h->hb_seq_host && h->fuse_seq) {
        void *p = nullptr;
        TRY(map_host_ptr(h, h->hb_seq_host, &p, "completion word"));
        h->tiny.start_word = reinterpret_cast<unsigned long long *>(p) + 1;
        h->tiny.start_seq = h->hb_seq;
    }
    const dim3 grid(((unsigned)N * h->tiny.tasks_per_agent + 3u) / 4u), block(256);
    EpisodeArrays ep{};
    HostBlock hb{};
    int episode = 0, auto_reset = 0;
    bool host = false;
    if (A == 2) {
        if (h->fuse_request && !dev.reseat_poses) {   // f110_step_host: host block + episode logic as the last workgroup's epilogue
            dev.fused_host = h->d_fused;
            dev.fused_seq = h->fuse_seq;
            h->fused_done = true;
            host = true;
        }
    } else if (h->tiny_host_request && !dev.reseat_poses) {
        ep = h->ep;
        hb = h->tiny_hb;
        episode = h->tiny_episode;
        auto_reset = h->tiny_auto_reset;
        h->fused_done = true;
        host = true;
    }
    if (host && A == 2) h->tiny.fh = h->fused_host_copy;   // (what d_fused holds: f110_step_host keeps the two equal)
    // one env of two cars: finalize_duo_tiny puts the scans into the caller's block early (its idle wave copies, the window lanes follow)
    h->tiny.host_scans = (host && h->hb_valid && h->hb_scans_by_kernel && A == 2 && N == 2 && !(kExperimental && h->exp.tiny_general_tail)) ? h->hb_dev.scans : nullptr;
#define TINY(P, I, H_) hipLaunchKernelGGL((k_step_tiny<P, I, H_>), grid, block, 0, st, dev, h->k, j, d_actions, h->tiny, ep, hb, episode, auto_reset)
    const bool ident = h->k.ident_rot != 0;
    if (A == 2) {
        if (host) { if (ident) TINY(true, true, true); else TINY(true, false, true); }
        else { if (ident) TINY(true, true, false); else TINY(true, false, false); }
    } else {
        if (host) { if (ident) TINY(false, true, true); else TINY(false, false, true); }
        else { if (ident) TINY(false, true, false); else TINY(false, false, false); }
    }
#undef TINY
    h->last_launches = 1;
    return F110_OK;
}

static int step_range(f110_sim *h, hipStream_t st, int begin, int count, const double *d_actions, int collide_mode, hipEvent_t *ev)
{
    const int N = h->N, A = h->cfg.num_agents, B = h->k.num_beams;
    AgentArrays dev = h->dev;
    dev.agent_begin = begin;
    dev.agent_count = count;
    const bool multi = A > 1;
    // the scan kernel of this step, decided ONCE: the launch below and what k_integrate prepares for it
    // (the longest-first list counter) follow the same answer
    const ScanKind scan = pick_scan(h, begin, count);
    dev.sched_count_zero = scan == SCAN_AGENT_SCHED ? h->d_tcount + (h->task_epoch & 1u) : nullptr;
    if (dev.noise_rng && (dev.noise_rng == 2 || h->noise_ub >= (long long)dev.noise_rows)) {
        const int apb = dev.noise_rng == 2 ? 16 : 64;   // per-agent streams: every agent needs a row, keep the waves many
        hipLaunchKernelGGL(k_noise_rows, dim3((count + apb - 1) / apb), dim3(256), 0, st, dev, h->noise_gen, B, apb);
    }
    if (ev) HIPCHK(h, hipEventRecord(ev[0], st));
    bool fused_integrate = false;
#ifdef F110_EXPERIMENTAL
    if (multi && collide_mode == 1 && A == 2) {
        hipLaunchKernelGGL(k_integrate<2>, grid1d(count, 64), dim3(64), 0, st, dev, h->k, d_actions);
        fused_integrate = true;
    } else if (multi && collide_mode == 1 && A == 4) {
        hipLaunchKernelGGL(k_integrate<4>, grid1d(count, 64), dim3(64), 0, st, dev, h->k, d_actions);
        fused_integrate = true;
    }
#endif
    if (!fused_integrate) {
        bool duo = true;   // the integration in two waves per 64 agents (k_integrate_duo)
        // RK4: thirteen waves per 64 agents (k_integrate_fan) up to kFanMaxAgents agents — the kernel is a latency
        // chain there; big batches are throughput-bound and keep the two-wave form
        bool fan = h->dev.integrator == F110_INTEGRATOR_RK4 && count <= kFanMaxAgents;
#ifdef F110_EXPERIMENTAL
        if (h->exp.integrate_duo >= 0) duo = h->exp.integrate_duo != 0;
        if (h->exp.integrate_fan >= 0) fan = h->exp.integrate_fan != 0 && h->dev.integrator == F110_INTEGRATOR_RK4;
#endif
        if (fan) hipLaunchKernelGGL(k_integrate_fan, grid1d(count, 64), dim3(64 * kFanWaves), 0, st, dev, h->k, d_actions);
        else if (duo) hipLaunchKernelGGL(k_integrate_duo, grid1d(count, 64), dim3(128), 0, st, dev, h->k, d_actions);
#ifdef F110_EXPERIMENTAL
        else hipLaunchKernelGGL(k_integrate<0>, grid1d(count, 256), dim3(256), 0, st, dev, h->k, d_actions);
#endif
    }
    // A = 2: the pair test and the opponent window inside the finalize kernel (k_finalize_pair_flat): one stream, no
    // events.  (Round 2 kept the side-stream form for big batches stepped without the in-step re-seat — crashed cars
    // pile up, their windows grow to all beams, and fixed lanes per agent then serialise — the flattened window loop
    // balances that inside the workgroup: 65 536 parked cars 0.624 -> 0.579 ms, crashed cars piling up 0.899 -> 0.852.)
    const bool pair_in_finalize = multi && collide_mode == 3 && A == 2 && (begin % 2) == 0;
    const bool multi_in_finalize = multi && collide_mode == 3 && A > 2 && A <= kMaxAgentsMulti && (begin % A) == 0 && (count % A) == 0;
    const bool no_collide_launch = fused_integrate || pair_in_finalize || multi_in_finalize;
    const bool side_collide = multi && !no_collide_launch && !(kExperimental && collide_mode == 2);
    // k_collide only feeds k_finalize, k_scan_rays only needs k_integrate: run the two side by
    // side (second stream, event fork/join) so the pair test + window set-up hides under the scan
    if (multi && !no_collide_launch) {
        if (side_collide) {
            HIPCHK(h, hipEventRecord(h->ev_integrated, st));
            HIPCHK(h, hipStreamWaitEvent(h->side_stream, h->ev_integrated, 0));
            hipLaunchKernelGGL(k_collide, grid1d(count, 64), dim3(64), 0, h->side_stream, dev, B);
            HIPCHK(h, hipEventRecord(h->ev_collided, h->side_stream));
        } else {
            hipLaunchKernelGGL(k_collide, grid1d(count, 64), dim3(64), 0, st, dev, B);
        }
    }
    if (ev) HIPCHK(h, hipEventRecord(ev[1], st));
    {
        RayJob j{};
        j.n_rays = (uint32_t)N * (uint32_t)B;
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
        j.lookups_total = h->lookups_on ? h->d_lookups : nullptr;
        j.path_stats = h->path_stats_on ? h->d_path_stats : nullptr;   // (step form: only the ray pass counts here)
#ifdef F110_EXPERIMENTAL
        j.trace = reinterpret_cast<unsigned long long *>(h->exp.scan_trace);
#endif
        j.k_cold = cold_consts(h);
        if (!j.k_cold) return fail(h, F110_ERR_HIP, "f110_step_device: constant upload failed");
        j.order = (h->multi_map && begin == 0 && count == N) ? h->d_scan_order : nullptr;
        const bool cnt = j.lookups_total != nullptr;
        // agent-aligned launch geometry: whole 64-ray tasks per agent, so every wave belongs to one agent (and one map)
        auto agent_grid = [&](uint32_t tpa, dim3 &grid, uint32_t &wpb, int tpw) {
            (void)rays_grid(j, h->scan_block, tpw);
            j.n_tasks = (uint32_t)count * tpa;
            j.first_pose = (uint32_t)begin;
            wpb = (uint32_t)h->scan_block / 64u;
            const uint32_t waves = (j.n_tasks + j.tasks_per_wave - 1) / j.tasks_per_wave;
            grid = dim3((waves + wpb - 1) / wpb);
        };
        const dim3 block(h->scan_block);
        dim3 grid;
        uint32_t wpb = 1;
        switch (scan) {
        case SCAN_DIRS: {
            // more beams than table directions, PADDED table: march each distinct direction once and write
            // the beams that share it from the same wave (k_scan_dirs_agent)
            const uint32_t tpa = (uint32_t)h->dir_stride / 64u;
            // (a direction task ends with ~3 coalesced write passes: four per wave, where the beam kernel wants three —
            // BASELINE configs[4] 42.8 M agent-steps/s at 3, 45.2 M at 4)
            agent_grid(tpa, grid, wpb, h->scan_tasks_auto && h->scan_tasks_per_wave == 3 ? 4 : h->scan_tasks_per_wave);
#define DIRS_SCAN(PM, ID)                                                                                                                    \
    do {                                                                                                                                     \
        if (cnt) hipLaunchKernelGGL((k_scan_dirs_agent<PM, ID, true>), grid, block, 0, st, j, h->k, h->d_maps_fast, h->d_maps_full, tpa);     \
        else hipLaunchKernelGGL((k_scan_dirs_agent<PM, ID, false>), grid, block, 0, st, j, h->k, h->d_maps_fast, h->d_maps_full, tpa);        \
    } while (0)
            if (h->multi_map) DIRS_SCAN(true, false);
            else if (h->k.ident_rot) DIRS_SCAN(false, true);
            else DIRS_SCAN(false, false);
#undef DIRS_SCAN
            break;
        }
        case SCAN_AGENT_SCHED: {
            // longest-first: last step's long tasks are served by the first blocks of the launch
            const uint32_t tpa = ((uint32_t)B + 63u) / 64u;
            agent_grid(tpa, grid, wpb, h->scan_tasks_per_wave);
            const uint32_t parity = h->task_epoch & 1u;
            j.sched = h->tsched[parity];
            j.epoch_r = h->task_epoch - 1u;
            j.epoch_w = h->task_epoch;
            j.long_blocks = (h->task_cap + wpb - 1) / wpb;
            j.long_prio = (uint32_t)h->exp.long_prio;
            j.long_rev = h->task_rev;
            h->task_epoch += 1u;
            const dim3 sgrid(grid.x + j.long_blocks);
            size_t slds = 0;
#ifdef F110_EXPERIMENTAL
            // probe: fewer waves per SIMD (an LDS reservation per one-wave workgroup) so that last step's long tasks,
            // which start first, march at the idle chip's latency — does the shorter chain beat the lost throughput?
            if (h->exp.scan_occupancy > 0 && h->scan_block == 64) slds = (size_t)(160 * 1024 / (4 * h->exp.scan_occupancy)) & ~(size_t)255;
#endif
            if (h->k.ident_rot)
                hipLaunchKernelGGL((k_scan_rays_agent<false, true, false, true>), sgrid, block, slds, st, j, h->k, h->d_maps_fast, h->d_maps_full, tpa);
            else
                hipLaunchKernelGGL((k_scan_rays_agent<false, false, false, true>), sgrid, block, slds, st, j, h->k, h->d_maps_fast, h->d_maps_full, tpa);
            break;
        }
        case SCAN_AGENT: {
            const uint32_t tpa = ((uint32_t)B + 63u) / 64u;
            agent_grid(tpa, grid, wpb, h->scan_tasks_per_wave);
            size_t lds = 0;
#ifdef F110_EXPERIMENTAL
            // fusion-feasibility probes: the occupancy a kernel with k_finalize_pair's 118 VGPRs
            // would run at (LDS reservation per one-wave workgroup caps the CU at 16 waves), and the price of a
            // per-env completion counter
            if (h->exp.scan_occupancy > 0 && h->scan_block == 64) lds = (size_t)(160 * 1024 / (4 * h->exp.scan_occupancy)) & ~(size_t)255;
            if (h->exp.scan_env_counter && !h->multi_map && !cnt) {
                if (!h->d_env_done) {
                    TRY(dmalloc(h, &h->d_env_done, (size_t)h->cfg.num_envs));
                    HIPCHK(h, hipMemsetAsync(h->d_env_done, 0, sizeof(uint32_t) * h->cfg.num_envs, st));
                }
                j.env_done = h->d_env_done;
                j.tasks_per_env = tpa * (uint32_t)A;
                if (h->k.ident_rot)
                    hipLaunchKernelGGL((k_scan_rays_agent<false, true, false, false, true>), grid, block, lds, st, j, h->k, h->d_maps_fast, h->d_maps_full, tpa);
                else
                    hipLaunchKernelGGL((k_scan_rays_agent<false, false, false, false, true>), grid, block, lds, st, j, h->k, h->d_maps_fast, h->d_maps_full, tpa);
                break;
            }
#endif
#define AGENT_SCAN(PM, ID)                                                                                                           \
    do {                                                                                                                             \
        if (cnt)                                                                                                                     \
            hipLaunchKernelGGL((k_scan_rays_agent<PM, ID, true>), grid, block, lds, st, j, h->k, h->d_maps_fast, h->d_maps_full, tpa); \
        else                                                                                                                         \
            hipLaunchKernelGGL((k_scan_rays_agent<PM, ID, false>), grid, block, lds, st, j, h->k, h->d_maps_fast, h->d_maps_full, tpa); \
    } while (0)
            if (h->multi_map)
                AGENT_SCAN(true, false);
            else if (h->k.ident_rot)
                AGENT_SCAN(false, true);
            else
                AGENT_SCAN(false, false);
#undef AGENT_SCAN
            break;
        }
        default: {   // SCAN_FLAT: ray = agent * B + beam (row-major table, few beams, or — product build — beams
                     // beyond the table directions on a map the PADDED layout cannot hold)
            const dim3 fgrid = rays_grid(j, h->scan_block, h->scan_tasks_per_wave);
            hipLaunchKernelGGL(pick_rays<true>(h->k, h->cfg.map_layout), fgrid, block, 0, st, j, h->k);
            break;
        }
        }
    }
    if (ev) HIPCHK(h, hipEventRecord(ev[2], st));
    if (side_collide) HIPCHK(h, hipStreamWaitEvent(st, h->ev_collided, 0));
    if (multi) {
        // few lanes per agent pay off when opponent windows are short (~36 beams); cars that have
        // crashed into each other see windows of up to all beams, so the narrow forms are used only
        // when finished envs are re-seated inside the step (f110_set_auto_reseat)
        const bool narrow = h->dev.reseat_poses != nullptr;
        // (with the pair test inside the kernel a group also carries that prologue: 16 lanes from 8192 agents up)
        int lanes = narrow && N >= 131072 ? 8 : (narrow && N >= (pair_in_finalize ? 8192 : 32768) ? 16 : 64);
        if (pair_in_finalize) {
            // k_finalize_pair_roles: the window loop flattened over the workgroup, AG agents per 256 threads, the prologue dealt
            // by role.  More agents per workgroup = fewer prologue waves and a better-balanced item list; small batches want
            // the workgroups many (measured: 65 536 agents AG 32 / 16 / 4: 0.726 / 0.738 / 0.815 ms; 4096 agents AG 16 / 4:
            // 0.1047 / 0.1053).  (Rounds 2-3's forms — fixed lanes per agent, the prologue dealt by agent — were retired in round 5.)
            lanes = N >= 32768 ? 8 : (N >= 4096 ? 16 : 64);
            if (h->fuse_request && begin == 0 && count == N && !dev.reseat_poses) {   // f110_step_host: host block + episode logic as this kernel's epilogue
                dev.fused_host = h->d_fused;
                dev.fused_seq = h->fuse_seq;
                h->fused_done = true;
            }
            if (dev.fused_host) {   // the instantiation that carries the f110_step_host epilogue
                if (lanes <= 8) hipLaunchKernelGGL((k_finalize_pair_roles<32, true>), dim3((count + 31) / 32), dim3(256), 0, st, dev, B);
                else if (lanes == 16) hipLaunchKernelGGL((k_finalize_pair_roles<16, true>), dim3((count + 15) / 16), dim3(256), 0, st, dev, B);
                else hipLaunchKernelGGL((k_finalize_pair_roles<4, true>), dim3((count + 3) / 4), dim3(256), 0, st, dev, B);
            } else if (lanes <= 8) hipLaunchKernelGGL(k_finalize_pair_roles<32>, dim3((count + 31) / 32), dim3(256), 0, st, dev, B);
            else if (lanes == 16) hipLaunchKernelGGL(k_finalize_pair_roles<16>, dim3((count + 15) / 16), dim3(256), 0, st, dev, B);
            else hipLaunchKernelGGL(k_finalize_pair_roles<4>, dim3((count + 3) / 4), dim3(256), 0, st, dev, B);
        }
        else if (multi_in_finalize) {
            const int envs = count / A;   // whole envs per workgroup
            if (A * (A - 1) <= kMaxRec) {
                const int G = kMaxRec / (A * (A - 1));
                hipLaunchKernelGGL(k_finalize_multi<kMaxRec>, dim3((envs + G - 1) / G), dim3(256), 0, st, dev, B, G);
            } else if (A * (A - 1) <= kMaxRecBig) {
                hipLaunchKernelGGL(k_finalize_multi<kMaxRecBig>, dim3(envs), dim3(256), 0, st, dev, B, 1);
            } else {   // more than 16 agents: one env per workgroup, its A (A - 1) records in tiles of 256
                hipLaunchKernelGGL(k_finalize_multi_tiled<kMaxRecBig>, dim3(envs), dim3(256), multi_lds_bytes(A, kMaxRecBig), st, dev, B, 1);
            }
        }
        else if (lanes == 8)
            hipLaunchKernelGGL(k_finalize<8>, dim3((count + 31) / 32), dim3(256), 0, st, dev, B);
        else if (lanes == 16)
            hipLaunchKernelGGL(k_finalize<16>, dim3((count + 15) / 16), dim3(256), 0, st, dev, B);
        else
            hipLaunchKernelGGL(k_finalize<64>, dim3((count + 3) / 4), dim3(256), 0, st, dev, B);
    } else {
        hipLaunchKernelGGL(k_finalize_solo, grid1d(count, 256), dim3(256), 0, st, dev);
    }
    if (ev) HIPCHK(h, hipEventRecord(ev[3], st));
    return F110_OK;
}

// step_groups = 0: do two env blocks pay for N agents, A per env?  bench.py's steady-regime workload, one block -> two blocks
// (profiles/r04_groups_bench_sweep.txt, ms per step):  A = 2:  1024 +2.8 %, 2048 +0.6 %, 3072 -1.6 %, 4096 -1.3 %, 6144 +1.7 %,
// 8192 +5.8 %, 16 384 +9.5 %, 32 768 +3.7 %, 65 536 +1.7 %;  A = 1 / 4 / 8 at 16 384: +7.2 / +6.9 / +10.3 %, at 65 536: +5.7 / +4.1 / +4.4 %,
// at 4096: -0.9 / -0.1 / +3.8 %.  Around 3000 .. 5000 agents the step IS its longest ray's chain of dependent samples: two
// blocks have two such chains side by side and nothing to hide under them (a heavier finalize, A >= 8, changes that).
// A = 2 above 32 768 agents stays one block for +1.7 %: one launch per kernel and step keeps the headline's per-kernel
// accounting (rocprofv3 durations, PMC per dispatch) directly readable.
static bool env_blocks_pay(int N, int A)
{
    if (A == 2 && N > 32768) return false;
    if (A <= 4 && N >= 2560 && N <= 5120) return false;
    return true;
}

// how the env axis is cut into `groups` blocks: whole envs, whole 64-agent waves where possible
static int group_envs(const f110_sim *h)
{
    const int E = h->cfg.num_envs, G = h->groups;
    int per = (E + G - 1) / G;
#ifdef F110_EXPERIMENTAL
    if (G == 2 && h->exp.group_split > 0) per = std::max(1, (int)((long long)E * h->exp.group_split / 100));   // probe: uneven halves
#endif
    if (per >= 64) per = (per + 63) / 64 * 64;
    return per;
}

int f110_step_groups(f110_sim *h, int32_t *groups, int32_t *probes, int32_t *last)
{
    if (!h) return fail(h, F110_ERR_INVALID, "null argument");
    if (groups) *groups = h->groups;
    if (probes) *probes = h->group_probes;
    if (last) *last = h->last_blocks;
    return F110_OK;
}

int f110_step_launches(f110_sim *h, int32_t *launches)
{
    if (!h || !launches) return fail(h, F110_ERR_INVALID, "null argument");
    *launches = h->last_launches;
    return F110_OK;
}

int f110_step_device(f110_sim *h, const double *d_actions)
{
    if (!h || !d_actions) return fail(h, F110_ERR_INVALID, "null argument");
    if (!h->has_map) return fail(h, F110_ERR_NO_MAP, "Map is not set for scan simulator.");
    if (!h->beams_uniform) return fail(h, F110_ERR_STATE, kBeamsMsg);
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    const int N = h->N, A = h->cfg.num_agents;
    h->dev.path_stats = h->path_stats_on ? h->d_path_stats : nullptr;
    if (h->comm_overlap && h->comm_swap_next) {
        // the previous step's scans are being gathered: this step fills the other buffer, once the
        // gather that last read THAT buffer (two steps ago) is done
        h->scans_cur ^= 1;
        h->dev.scans = h->scan_bufs[h->scans_cur];
        if (h->gather_pending[h->scans_cur]) {
            const bool inflight = h->comm_inflight;
            h->comm_inflight = false;
            const int rj = join_groups(h);
            h->comm_inflight = inflight;
            if (rj != F110_OK) return rj;
            HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_gather_done[h->scans_cur], 0));
            h->gather_pending[h->scans_cur] = false;
            h->main_dirty = true;
        }
        h->comm_swap_next = false;
    }
    // shared noise stream: the row cache must reach the longest live episode (or its capacity)
    if (h->dev.noise_rng == 1 && h->noise_rows_ready < h->dev.noise_rows) {
        if (h->noise_ub >= h->noise_rows_ready) {
            TRY(join_groups(h));
            TRY(noise_adopt(h));   // the rows generated ahead
            if (h->noise_ub >= h->noise_rows_ready && h->noise_rows_ready < h->dev.noise_rows) {   // (not enough: in line, this step waits)
                const long long want = std::max<long long>(2LL * h->noise_rows_ready, h->noise_ub + 1);
                TRY(noise_cache_extend(h, (int)std::min<long long>(want, h->dev.noise_rows)));
            }
        }
        if (!h->noise_ahead && 2 * (h->noise_ub + 1) >= h->noise_rows_ready) TRY(noise_start_ahead(h));
    }
    const bool prof = h->profiling && h->prof_used + 4 <= 4 * 65536;
    // the env groups need the agent-aligned scan (a launch per agent range); per-kernel profiling
    // brackets the kernels of ONE stream, so a profiled step runs as one block on the main stream
    // (two groups borrow the side stream, which the older collide forms use themselves)
    // a tiny batch (the reference's own shape: one env of two cars) is ONE launch: integrate, scan, finalize and — under
    // f110_step_host — the observation block and the completion word, in k_step_tiny
    const bool tiny = h->tiny_request && !prof;   // (f110_step_host decided: tiny_applies)
    const bool grouped = !tiny && h->groups > 1 && !prof && (h->multi_map || agent_aligned(h)) && h->dir_stride == 0 &&
                         !(h->gstreams[0] == h->stream && h->collide_mode != 3 && A > 1) &&
                         (!h->groups_auto || (!h->touched && env_blocks_pay(N, A)));
    h->last_launches = 0;
    if (!grouped) {
        TRY(join_groups(h));
        h->main_dirty = true;
        hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
        if (prof) {
            for (int i = 0; i < 4; ++i)
                if (!(ev[i] = prof_event(h))) return fail(h, F110_ERR_HIP, "hipEventCreate failed");
        }
        const int cmode = h->collide_mode;
        if (tiny) {
            TRY(step_tiny(h, h->stream, d_actions));
        } else {
            TRY(step_range(h, h->stream, 0, N, d_actions, cmode, prof ? ev : nullptr));
        }
    } else {
        if (h->main_dirty) {
            HIPCHK(h, hipEventRecord(h->ev_main, h->stream));
            for (hipStream_t gs : h->gstreams)
                if (gs != h->stream) HIPCHK(h, hipStreamWaitEvent(gs, h->ev_main, 0));
            h->main_dirty = false;
        }
        const int per = group_envs(h), E = h->cfg.num_envs;
        const int mode = h->collide_mode == 3 ? 3 : ((A == 2 || A == 4) ? 1 : 2);
        for (int g = 0; g < h->groups; ++g) {
            const int e0 = g * per, e1 = std::min(E, e0 + per);
            if (e0 >= e1) break;
            TRY(step_range(h, h->gstreams[g], e0 * A, (e1 - e0) * A, d_actions, mode == 0 ? 2 : mode, nullptr));
        }
        h->groups_busy = true;
    }
    h->touched = false;
    h->last_blocks = grouped ? h->groups : 1;
    h->noise_ub += 1;
    HIPCHK(h, hipGetLastError());
    return F110_OK;
}

int f110_stream_fence(f110_sim *h)
{
    if (!h) return fail(nullptr, F110_ERR_INVALID, "null handle");
    ENTER(h);   // joins the env blocks into the main stream; main_dirty + touched: the next step is one block behind the caller's work
    return F110_OK;
}

int f110_step(f110_sim *h, const double *actions)
{
    if (!h || !actions) return fail(h, F110_ERR_INVALID, "null argument");
    if (!h->has_map) return fail(h, F110_ERR_NO_MAP, "Map is not set for scan simulator.");
    if (!h->beams_uniform) return fail(h, F110_ERR_STATE, kBeamsMsg);
    ENTER(h);   // the previous step (possibly still running on the group streams) reads d_actions
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
    ENTER(h);
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
    ENTER(h);
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
    ENTER(h);
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

int f110_device_mem_info(f110_sim *h, size_t *free_bytes, size_t *total_bytes)
{
    if (!h || !free_bytes || !total_bytes) return fail(h, F110_ERR_INVALID, "null argument");
    ENTER(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemGetInfo(free_bytes, total_bytes));
    return F110_OK;
}

int f110_device_alloc(f110_sim *h, size_t bytes, void **out)
{
    if (!h || !out) return fail(h, F110_ERR_INVALID, "null argument");
    ENTER(h);
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    HIPCHK(h, hipMalloc(out, bytes ? bytes : 8));
    return F110_OK;
}

int f110_device_free(f110_sim *h, void *p)
{
    if (!h) return fail(nullptr, F110_ERR_INVALID, "null handle");
    ENTER(h);
    if (p) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        HIPCHK(h, hipFree(p));
    }
    return F110_OK;
}

int f110_memcpy_h2d(f110_sim *h, void *dst, const void *src, size_t bytes)
{
    if (!h || !dst || !src) return fail(h, F110_ERR_INVALID, "null argument");
    ENTER(h);
    HIPCHK(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return F110_OK;
}

int f110_memcpy_d2h(f110_sim *h, void *dst, const void *src, size_t bytes)
{
    if (!h || !dst || !src) return fail(h, F110_ERR_INVALID, "null argument");
    ENTER(h);
    HIPCHK(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return F110_OK;
}

// ---- timing ------------------------------------------------------------------------------
int f110_timer_begin(f110_sim *h)
{
    if (!h) return fail(nullptr, F110_ERR_INVALID, "null handle");
    ENTER(h);
    HIPCHK(h, hipEventRecord(h->ev_begin, h->stream));
    return F110_OK;
}

int f110_timer_end_ms(f110_sim *h, double *ms)
{
    if (!h || !ms) return fail(h, F110_ERR_INVALID, "null argument");
    ENTER(h);
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
    ENTER(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->profiling = enable != 0;
    h->prof_used = 0;
    return F110_OK;
}

int f110_profile_read(f110_sim *h, int32_t *n_launches, double *scan_ms, double *dyn_ms, double *fin_ms)
{
    if (!h) return fail(nullptr, F110_ERR_INVALID, "null handle");
    ENTER(h);
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
    ENTER(h);
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
    j.path_stats = h->path_stats_on ? h->d_path_stats : nullptr;
    j.k_cold = cold_consts(h);
    if (!j.k_cold) return fail(h, F110_ERR_HIP, "f110_scan_batch: constant upload failed");
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

// ---- the reference's example policy (examples/waypoint_follow.py) -----------------------------
static int check_planner_args(f110_sim *h, const void *wp, int32_t M, double lookahead)
{
    if (!h || !wp) return fail(h, F110_ERR_INVALID, "pure pursuit: null argument");
    if (M < 2) return fail(h, F110_ERR_INVALID, "pure pursuit: need at least 2 waypoints (got %d)", M);
    if (!(lookahead > 0)) return fail(h, F110_ERR_INVALID, "pure pursuit: lookahead distance must be positive");
    return F110_OK;
}

int f110_pure_pursuit_batch(f110_sim *h, const double *h_waypoints, int32_t M, const double *h_poses, int32_t m, double lookahead,
                            double vgain, double wheelbase, double max_reacquire, double *h_actions)
{
    TRY(check_planner_args(h, h_waypoints, M, lookahead));
    if (!h_poses || !h_actions || m < 0) return fail(h, F110_ERR_INVALID, "pure pursuit: bad argument");
    if (m == 0) return F110_OK;
    ENTER(h);
    Scratch s(h);
    double *dw = nullptr, *dp = nullptr, *da = nullptr;
    TRY(s.up(h_waypoints, (size_t)3 * M, &dw));
    TRY(s.up(h_poses, (size_t)3 * m, &dp));
    TRY(s.up<double>(nullptr, (size_t)2 * m, &da));
    hipLaunchKernelGGL(k_pure_pursuit, grid1d((size_t)m * kPlanLanes, 256), dim3(256), 0, h->stream, dw, M, dp, dp + 1, dp + 2, 3, m, lookahead, vgain, wheelbase,
                       max_reacquire, da);
    HIPCHK(h, hipGetLastError());
    TRY(s.down(h_actions, da, (size_t)2 * m));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return F110_OK;
}

int f110_pure_pursuit_device(f110_sim *h, const double *d_waypoints, int32_t M, double lookahead, double vgain, double wheelbase,
                             double max_reacquire, double *d_actions)
{
    TRY(check_planner_args(h, d_waypoints, M, lookahead));
    if (!d_actions) return fail(h, F110_ERR_INVALID, "pure pursuit: null actions buffer");
    const int N = h->N;
    const double *st = h->dev.state;
    if (h->last_blocks == 2 && h->groups_busy && !h->touched) {
        // a closed device-side loop (plan, step, plan, ...) behind a two-block step: each block's agents are planned for on the
        // block's own stream (a pose in, an action out, per agent), so the two halves of the batch stay independent
        HIPCHK(h, hipSetDevice(h->cfg.device_id));
        const int per = group_envs(h), E = h->cfg.num_envs, A = h->cfg.num_agents;
        for (int g = 0; g < 2; ++g) {
            const int e0 = g * per, e1 = std::min(E, e0 + per);
            if (e0 >= e1) break;
            const size_t i0 = (size_t)e0 * A;
            const int n = (e1 - e0) * A;
            hipLaunchKernelGGL(k_pure_pursuit, grid1d((size_t)n * kPlanLanes, 256), dim3(256), 0, h->gstreams[g], d_waypoints, M, st + i0, st + N + i0,
                               st + 4 * (size_t)N + i0, 1, n, lookahead, vgain, wheelbase, max_reacquire, d_actions + 2 * i0);
        }
        HIPCHK(h, hipGetLastError());
        return F110_OK;
    }
    ENTER(h);
    hipLaunchKernelGGL(k_pure_pursuit, grid1d((size_t)N * kPlanLanes, 256), dim3(256), 0, h->stream, d_waypoints, M, st, st + N, st + 4 * (size_t)N, 1, N, lookahead,
                       vgain, wheelbase, max_reacquire, d_actions);
    HIPCHK(h, hipGetLastError());
    h->touched = false;   // (the next step may split: it forks from what this call left on the main stream)
    return F110_OK;
}

int f110_scan_policy_device(f110_sim *h, double steer_gain, double steer_max, double sector_limit, double v_lo, double v_hi, double d_ref, double *d_actions)
{
    if (!h || !d_actions) return fail(h, F110_ERR_INVALID, "scan policy: null argument");
    if (!(d_ref > 0.) || !(steer_max >= 0.)) return fail(h, F110_ERR_INVALID, "scan policy: d_ref must be > 0 and steer_max >= 0");
    if (h->cfg.num_beams > 2048) return fail(h, F110_ERR_INVALID, "scan policy: at most 2048 beams (the rows are staged in LDS)");
    if (h->cfg.num_beams < 64) return fail(h, F110_ERR_INVALID, "scan policy: needs at least 64 beams (one per sector)");
    if (!(sector_limit >= h->cfg.fov / 128.)) return fail(h, F110_ERR_INVALID, "scan policy: sector_limit must be at least fov/128 (half a sector), else no sector is eligible");
    const int N = h->N, B = h->cfg.num_beams;
    if (h->last_blocks == 2 && h->groups_busy && !h->touched) {
        // behind a two-block step: each block's agents on the block's own stream (an agent reads its own scan row only)
        HIPCHK(h, hipSetDevice(h->cfg.device_id));
        const int per = group_envs(h), E = h->cfg.num_envs, A = h->cfg.num_agents;
        for (int g = 0; g < 2; ++g) {
            const int e0 = g * per, e1 = std::min(E, e0 + per);
            if (e0 >= e1) break;
            const int n = (e1 - e0) * A;
            hipLaunchKernelGGL(k_scan_policy, grid1d((size_t)n * 64, 256), dim3(256), (size_t)4 * B * sizeof(double), h->gstreams[g], h->dev.scans, B, h->cfg.fov, e0 * A, n, steer_gain, steer_max,
                               sector_limit, v_lo, v_hi, d_ref, d_actions);
        }
        HIPCHK(h, hipGetLastError());
        return F110_OK;
    }
    ENTER(h);
    hipLaunchKernelGGL(k_scan_policy, grid1d((size_t)N * 64, 256), dim3(256), (size_t)4 * B * sizeof(double), h->stream, h->dev.scans, B, h->cfg.fov, 0, N, steer_gain, steer_max, sector_limit,
                       v_lo, v_hi, d_ref, d_actions);
    HIPCHK(h, hipGetLastError());
    h->touched = false;   // (as f110_pure_pursuit_device: the next step may split)
    return F110_OK;
}

int f110_scan_path_stats(f110_sim *h, int32_t enable, int64_t *out3)
{
    if (!h) return fail(nullptr, F110_ERR_INVALID, "null handle");
    ENTER(h);
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    if (!h->d_path_stats) {
        HIPCHK(h, hipMalloc(reinterpret_cast<void **>(&h->d_path_stats), 3 * sizeof(unsigned long long)));
        HIPCHK(h, hipMemsetAsync(h->d_path_stats, 0, 3 * sizeof(unsigned long long), h->stream));
    }
    if (out3) {
        HIPCHK(h, hipMemcpyAsync(out3, h->d_path_stats, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipMemsetAsync(h->d_path_stats, 0, 3 * sizeof(unsigned long long), h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    if (enable >= 0) h->path_stats_on = enable != 0;
    return F110_OK;
}

int f110_beam_dir_index_batch(f110_sim *h, const double *thetas, int32_t m, int32_t *idx)
{
    if (!h || !thetas || !idx || m < 0) return fail(h, F110_ERR_INVALID, "bad argument");
    ENTER(h);
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
    ENTER(h);
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
    ENTER(h);
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
    ENTER(h);
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
    ENTER(h);
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
    ENTER(h);
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
    ENTER(h);
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
    ENTER(h);
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
    ENTER(h);
    if (m == 0) return F110_OK;
    const size_t B = (size_t)h->cfg.num_beams;
    Scratch s(h);
    double *de, *dv, *ds;
    int32_t *dm = nullptr;
    TRY(s.up(ego, (size_t)3 * m, &de));
    TRY(s.up(verts, (size_t)8 * m, &dv));
    TRY(s.up(scans, (size_t)m * B, &ds));
    if (minmax) TRY(s.up<int32_t>(nullptr, (size_t)2 * m, &dm));
    hipLaunchKernelGGL(k_raycast_unit, dim3(m), dim3(128), 0, h->stream, de, dv, m, (int)B, h->d_scan_angles, h->dev.angle_inc, h->beams_uniform ? 1 : 0, ds, dm);
    HIPCHK(h, hipGetLastError());
    TRY(s.down(scans, ds, (size_t)m * B));
    if (minmax) TRY(s.down(minmax, dm, (size_t)2 * m));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return F110_OK;
}

int f110_get_range_batch(f110_sim *h, const double *in, int32_t m, double *out)
{
    if (!h || !in || !out || m < 0) return fail(h, F110_ERR_INVALID, "bad argument");
    ENTER(h);
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
    ENTER(h);
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

int f110_dt_from_bitmap(f110_sim *h, const uint8_t *img, int32_t H, int32_t W, double resolution, double *dt)
{
    if (!h || !img || !dt) return fail(h, F110_ERR_INVALID, "null argument");
    ENTER(h);
    if (H < 1 || W < 1 || H > 16384 || W > 16384) return fail(h, F110_ERR_INVALID, "f110_dt_from_bitmap: bad shape %dx%d", H, W);
    const size_t n = (size_t)H * W;
    Scratch s(h);
    uint8_t *dimg;
    uint32_t *dg, *dd;
    double *ddt;
    TRY(s.up(img, n, &dimg));
    TRY(s.up<uint32_t>(nullptr, n, &dg));
    TRY(s.up<uint32_t>(nullptr, n, &dd));
    TRY(s.up<double>(nullptr, n, &ddt));
    hipLaunchKernelGGL(k_edt_columns, grid1d(W, 64), dim3(64), 0, h->stream, dimg, H, W, dg);
    hipLaunchKernelGGL(k_edt_rows, dim3(H), dim3(256), (size_t)W * sizeof(uint32_t), h->stream, dg, H, W, dd);
    hipLaunchKernelGGL(k_dt_from_d2, grid1d(n, 256), dim3(256), 0, h->stream, dd, n, resolution, ddt);
    HIPCHK(h, hipGetLastError());
    TRY(s.down(dt, ddt, n));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return F110_OK;
}

int f110_helper_batch(f110_sim *h, int32_t op, const double *in, int32_t m, int32_t n, double *out)
{
    if (!h || !in || !out || m < 0) return fail(h, F110_ERR_INVALID, "f110_helper_batch: bad argument");
    ENTER(h);
    int in_w = 0, out_w = 0;
    bool bodies = false, needs_map = false;
    switch (op) {
    case F110_OP_ACCL_CONSTRAINTS: case F110_OP_STEERING_CONSTRAINT: in_w = 6; out_w = 1; break;
    case F110_OP_CROSS: in_w = 4; out_w = 1; break;
    case F110_OP_ARE_COLLINEAR: in_w = 6; out_w = 1; break;
    case F110_OP_PERPENDICULAR: in_w = 2; out_w = 2; break;
    case F110_OP_TRIPLE_PRODUCT: in_w = 6; out_w = 2; break;
    case F110_OP_AVG_POINT: bodies = true; in_w = 2 * n; out_w = 2; break;
    case F110_OP_FURTHEST_POINT: bodies = true; in_w = 2 * n + 2; out_w = 1; break;
    case F110_OP_SUPPORT: bodies = true; in_w = 4 * n + 2; out_w = 2; break;
    case F110_OP_GET_TRMTX: in_w = 3; out_w = 16; break;
    case F110_OP_XY_2_RC: in_w = 9; out_w = 2; break;
    case F110_OP_DISTANCE_TRANSFORM: needs_map = true; in_w = 2; out_w = 1; break;
    case F110_OP_TRACE_RAY: needs_map = true; in_w = 3; out_w = 1; break;
    default: return fail(h, F110_ERR_INVALID, "f110_helper_batch: unknown op %d", op);
    }
    if (bodies && (n < 1 || n > 4096)) return fail(h, F110_ERR_INVALID, "f110_helper_batch: a body needs 1..4096 vertices (got %d)", n);
    if (needs_map && !h->has_map) return fail(h, F110_ERR_NO_MAP, "Map is not set for scan simulator.");
    if (m == 0) return F110_OK;
    Scratch s(h);
    double *di, *dout;
    TRY(s.up(in, (size_t)in_w * m, &di));
    TRY(s.up<double>(nullptr, (size_t)out_w * m, &dout));
    hipLaunchKernelGGL(k_helper_unit, grid1d(m, 128), dim3(128), 0, h->stream, op, di, m, n, in_w, out_w, dout, h->k);
    HIPCHK(h, hipGetLastError());
    TRY(s.down(out, dout, (size_t)out_w * m));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return F110_OK;
}

}  // extern "C"
