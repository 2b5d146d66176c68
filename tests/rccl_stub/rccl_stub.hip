// TEST INFRASTRUCTURE — a stand-in for librccl.so.1 whose "ranks" are THREADS of one process on ONE device.
//
// RCCL refuses two ranks on one device, and no multi-GPU box has been available in any round, so everything in
// f110_comm_* that only happens at world size > 1 (the [ranks][N][B] receive layout, the offsets of gather-to-root,
// the double-buffered overlap choreography with a peer that is one step ahead or behind) had never executed.  This
// library implements the handful of entry points libf110_hip.so resolves with dlsym (f110_hip.hip rccl_api) with the
// semantics the real library has on a stream: a collective is enqueued on the caller's stream, reads the peers' send
// buffers when THEIR streams have reached the call, and a rank's later work on its stream starts only after every
// peer has finished reading its send buffer.  Data moves by device-to-device copies.  It tests OUR call pattern and
// buffer arithmetic; it says nothing about RCCL or xGMI.  Loaded by tests/test_gpu_round4.py through LD_LIBRARY_PATH.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <condition_variable>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {
struct Op {
    int kind;   // 0 all-gather, 1 send, 2 recv
    const void *send;
    void *recv;
    size_t bytes;
    int peer;
};
struct World {
    int n = 0, joined = 0;
    std::mutex m;
    std::condition_variable cv;
    long round = 0;
    int posted = 0, finished = 0;
    std::vector<std::vector<Op>> ops;          // per rank, this round
    std::vector<hipEvent_t> ready, done;       // per rank
};
std::mutex g_m;
std::map<std::string, World *> g_worlds;
std::atomic<unsigned> g_ids{1};
size_t esize(ncclDataType_t t) { return t == ncclFloat64 || t == ncclInt64 || t == ncclUint64 ? 8 : (t == ncclFloat32 || t == ncclInt32 || t == ncclUint32 ? 4 : 1); }
}  // namespace

struct ncclComm {
    World *w;
    int rank;
};
namespace {
thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;
thread_local ncclComm *t_comm = nullptr;
thread_local hipStream_t t_stream = nullptr;

// one group of operations, all ranks together
ncclResult_t exchange(ncclComm *c, hipStream_t st, std::vector<Op> &mine)
{
    World *w = c->w;
    const int n = w->n, me = c->rank;
    if (hipEventRecord(w->ready[me], st) != hipSuccess) return ncclUnhandledCudaError;   // my send buffers are valid from here on my stream
    long my_round;
    {
        std::unique_lock<std::mutex> lk(w->m);
        my_round = w->round;
        w->ops[me] = mine;
        if (++w->posted == n) w->cv.notify_all();
        w->cv.wait(lk, [&] { return w->posted == n || w->round != my_round; });
    }
    for (size_t k = 0; k < mine.size(); ++k) {
        const Op &o = mine[k];
        if (o.kind == 0) {   // all-gather: my receive block p = rank p's send buffer of ITS k-th operation
            for (int p = 0; p < n; ++p) {
                if (w->ops[p].size() <= k || w->ops[p][k].kind != 0 || w->ops[p][k].bytes != o.bytes) return ncclInvalidUsage;
                if (hipStreamWaitEvent(st, w->ready[p], 0) != hipSuccess) return ncclUnhandledCudaError;
                if (hipMemcpyAsync(static_cast<char *>(o.recv) + (size_t)p * o.bytes, w->ops[p][k].send, o.bytes, hipMemcpyDeviceToDevice, st) != hipSuccess)
                    return ncclUnhandledCudaError;
            }
        } else if (o.kind == 2) {   // recv from o.peer: its j-th send to me pairs with my j-th recv from it
            int j = 0;
            for (size_t q = 0; q < k; ++q)
                if (mine[q].kind == 2 && mine[q].peer == o.peer) ++j;
            const Op *src = nullptr;
            for (const Op &po : w->ops[o.peer])
                if (po.kind == 1 && po.peer == me && j-- == 0) { src = &po; break; }
            if (!src || src->bytes != o.bytes) return ncclInvalidUsage;
            if (hipStreamWaitEvent(st, w->ready[o.peer], 0) != hipSuccess) return ncclUnhandledCudaError;
            if (hipMemcpyAsync(o.recv, src->send, o.bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return ncclUnhandledCudaError;
        }
    }
    if (hipEventRecord(w->done[me], st) != hipSuccess) return ncclUnhandledCudaError;   // I have read everybody's send buffers
    {
        std::unique_lock<std::mutex> lk(w->m);
        if (++w->finished == n) {
            w->posted = w->finished = 0;
            w->round += 1;
            w->cv.notify_all();
        } else {
            w->cv.wait(lk, [&] { return w->round != my_round; });
        }
    }
    for (int p = 0; p < n; ++p)   // whatever I enqueue next may overwrite my send buffers: only after the peers have read them
        if (p != me && hipStreamWaitEvent(st, w->done[p], 0) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

ncclResult_t submit(ncclComm *c, hipStream_t st, const Op &o)
{
    if (!c) return ncclInvalidArgument;
    if (t_depth > 0) {
        if (t_comm && (t_comm != c || t_stream != st)) return ncclInvalidUsage;   // one communicator and stream per group (all this stub needs)
        t_comm = c;
        t_stream = st;
        t_ops.push_back(o);
        return ncclSuccess;
    }
    std::vector<Op> one{o};
    return exchange(c, st, one);
}
}  // namespace

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    std::memset(id, 0, sizeof *id);
    std::snprintf(id->internal, sizeof id->internal, "f110-rccl-stub-%u", g_ids.fetch_add(1));
    return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    World *w;
    {
        std::lock_guard<std::mutex> lk(g_m);
        World *&slot = g_worlds[std::string(id.internal, sizeof id.internal)];
        if (!slot) {
            slot = new World;
            slot->n = nranks;
            slot->ops.resize(nranks);
            slot->ready.resize(nranks);
            slot->done.resize(nranks);
            for (int p = 0; p < nranks; ++p) {
                if (hipEventCreateWithFlags(&slot->ready[p], hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
                if (hipEventCreateWithFlags(&slot->done[p], hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
            }
        }
        w = slot;
    }
    if (w->n != nranks) return ncclInvalidArgument;
    {
        std::unique_lock<std::mutex> lk(w->m);
        ++w->joined;
        w->cv.notify_all();
        w->cv.wait(lk, [&] { return w->joined >= w->n; });   // like the real call: returns when every rank has joined
    }
    *comm = new ncclComm{w, rank};
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    delete comm;   // (the world and its events live until the process ends: a test process)
    return ncclSuccess;
}
ncclResult_t ncclCommCount(const ncclComm_t comm, int *count)
{
    *count = comm->w->n;
    return ncclSuccess;
}
ncclResult_t ncclCommUserRank(const ncclComm_t comm, int *rank)
{
    *rank = comm->rank;
    return ncclSuccess;
}
const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : (r == ncclInvalidUsage ? "invalid usage (stub: mismatched operations between ranks)" : "stub error"); }
ncclResult_t ncclGroupStart()
{
    ++t_depth;
    return ncclSuccess;
}
ncclResult_t ncclGroupEnd()
{
    if (t_depth <= 0) return ncclInvalidUsage;
    if (--t_depth > 0) return ncclSuccess;
    ncclResult_t rc = ncclSuccess;
    if (t_comm) rc = exchange(t_comm, t_stream, t_ops);
    t_ops.clear();
    t_comm = nullptr;
    t_stream = nullptr;
    return rc;
}
ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream)
{
    return submit(comm, stream, Op{0, sendbuff, recvbuff, sendcount * esize(datatype), -1});
}
ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
    return submit(comm, stream, Op{1, sendbuff, nullptr, count * esize(datatype), peer});
}
ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
    return submit(comm, stream, Op{2, nullptr, recvbuff, count * esize(datatype), peer});
}
}
