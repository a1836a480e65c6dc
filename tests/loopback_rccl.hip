// tests/loopback_rccl.hip — TEST INFRASTRUCTURE, not part of the product: an in-process stand-in for librccl.
//
// RCCL takes one rank per device, so on a one-GPU box the library's C-level exchange (nbodysim_amd/csrc/nb_comm.cpp,
// the executor of the per-step schedule) can only ever run with ONE member through the real thing.  This file provides
// the twelve nccl* entry points nb_comm resolves (load_rccl) over plain device buffers of ONE process, several ranks per
// device allowed, so that nb_comm_create_all / nb_comm_step run with 2 and 4 members on one GPU and the buffer offsets,
// element counts and ncclGroup bracketing of run_schedule meet real data.  nb_comm loads it only when a test names it
// with nb_debug_comm_transport(); the exported symbol `nb_loopback_transport` tells nb_comm that ranks may share a device.
//
// Semantics kept: collectives are enqueued on the caller's streams and complete in stream order; the calls of all ranks
// of a communicator arrive inside one ncclGroupStart / ncclGroupEnd (the one-process-drives-all-ranks shape) or, for a
// communicator of one rank, on their own.  Sums are formed in RANK ORDER (rank 0 first), the association of the
// library's own in-process exchange (nb_exchange_accelerations / _allreduce), so results can be compared bit for bit.
// Stricter than RCCL on purpose: every collective ends with a barrier between the ranks' streams.
//
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -shared -fPIC -o tests/libnb_loopback_rccl.so tests/loopback_rccl.hip
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <vector>

extern "C" { int nb_loopback_transport = 1; }

namespace {

struct Group {
    std::vector<ncclComm *> members;     // by rank
    std::vector<void *> tmp;             // all-reduce scratch per rank
    size_t tmp_bytes = 0;
    int live = 0;
};

}  // namespace

struct ncclComm {
    std::shared_ptr<Group> g;
    int rank = 0, dev = 0;
    hipEvent_t ready = nullptr, mid = nullptr, done = nullptr;
};

namespace {

enum Kind { ALLGATHER, REDUCE_SCATTER, ALLREDUCE };
struct Pending { Kind kind; const void *send; void *recv; size_t count; ncclDataType_t ty; ncclComm *comm; hipStream_t st; };

thread_local int t_depth = 0;
thread_local std::vector<Pending> t_queue;
std::mutex g_mutex;
long g_calls = 0, g_fail_at = -1;        // fault injection: the g_fail_at-th collective call (0-based) fails

struct Ptrs { const void *p[64]; };

template <typename T>
__global__ void sum_ranks(Ptrs src, int n, size_t offset, size_t count, T *dst)
{
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    T a = static_cast<const T *>(src.p[0])[offset + k];
    for (int r = 1; r < n; ++r) a += static_cast<const T *>(src.p[r])[offset + k];      // rank order: ((s0 + s1) + s2) + ...
    dst[k] = a;
}

size_t esize(ncclDataType_t ty) { return ty == ncclDouble ? 8 : 4; }

#define HIPOK(call) do { if ((call) != hipSuccess) { (void)hipGetLastError(); return ncclUnhandledCudaError; } } while (0)

ncclResult_t launch_sum(const Ptrs &src, int n, size_t offset, size_t count, ncclDataType_t ty, void *dst, hipStream_t st)
{
    if (count == 0) return ncclSuccess;
    const unsigned grid = (unsigned)((count + 255) / 256);
    if (ty == ncclDouble) sum_ranks<double><<<grid, 256, 0, st>>>(src, n, offset, count, (double *)dst);
    else                  sum_ranks<float><<<grid, 256, 0, st>>>(src, n, offset, count, (float *)dst);
    HIPOK(hipGetLastError());
    return ncclSuccess;
}

// all ranks' streams wait for event `which` of every other rank (a barrier between the streams at this point of the schedule)
ncclResult_t fence(const std::vector<Pending *> &ops, hipEvent_t ncclComm::*which)
{
    for (Pending *p : ops) { HIPOK(hipSetDevice(p->comm->dev)); HIPOK(hipEventRecord(p->comm->*which, p->st)); }
    for (Pending *p : ops) {
        HIPOK(hipSetDevice(p->comm->dev));
        for (Pending *q : ops) if (q != p) HIPOK(hipStreamWaitEvent(p->st, q->comm->*which, 0));
    }
    return ncclSuccess;
}

ncclResult_t run_group(Group &g, std::vector<Pending *> &ops)
{
    const int n = (int)g.members.size();
    if ((int)ops.size() != n) return ncclInvalidUsage;                 // a rank is missing from the group: real RCCL would hang
    std::vector<Pending *> by_rank((size_t)n, nullptr);
    for (Pending *p : ops) {
        if (by_rank[(size_t)p->comm->rank]) return ncclInvalidUsage;   // two calls of one rank in one group
        by_rank[(size_t)p->comm->rank] = p;
    }
    const Pending &f = *by_rank[0];
    for (Pending *p : by_rank) if (p->kind != f.kind || p->count != f.count || p->ty != f.ty) return ncclInvalidArgument;
    const size_t es = esize(f.ty), bytes = f.count * es;
    ncclResult_t rc;
    if ((rc = fence(by_rank, &ncclComm::ready)) != ncclSuccess) return rc;           // every rank's input is complete
    Ptrs src;
    for (int r = 0; r < n; ++r) src.p[r] = by_rank[(size_t)r]->send;
    switch (f.kind) {
    case ALLGATHER:
        for (Pending *p : by_rank) {
            HIPOK(hipSetDevice(p->comm->dev));
            for (int s = 0; s < n; ++s) {
                char *dst = (char *)p->recv + (size_t)s * bytes;
                if ((const void *)dst != by_rank[(size_t)s]->send && bytes) HIPOK(hipMemcpyAsync(dst, by_rank[(size_t)s]->send, bytes, hipMemcpyDeviceToDevice, p->st));
            }
        }
        break;
    case REDUCE_SCATTER:
        for (Pending *p : by_rank) {
            HIPOK(hipSetDevice(p->comm->dev));
            if ((rc = launch_sum(src, n, (size_t)p->comm->rank * f.count, f.count, f.ty, p->recv, p->st)) != ncclSuccess) return rc;
        }
        break;
    case ALLREDUCE:
        if (g.tmp_bytes < bytes) {
            for (int r = 0; r < n; ++r) {
                HIPOK(hipSetDevice(g.members[(size_t)r]->dev));
                if (g.tmp[(size_t)r]) HIPOK(hipFree(g.tmp[(size_t)r]));
                g.tmp[(size_t)r] = nullptr;
                HIPOK(hipMalloc(&g.tmp[(size_t)r], bytes));
            }
            g.tmp_bytes = bytes;
        }
        for (Pending *p : by_rank) {
            HIPOK(hipSetDevice(p->comm->dev));
            if ((rc = launch_sum(src, n, 0, f.count, f.ty, g.tmp[(size_t)p->comm->rank], p->st)) != ncclSuccess) return rc;
        }
        if ((rc = fence(by_rank, &ncclComm::mid)) != ncclSuccess) return rc;         // in place: nobody overwrites an input that is still being read
        for (Pending *p : by_rank) {
            HIPOK(hipSetDevice(p->comm->dev));
            if (bytes) HIPOK(hipMemcpyAsync(p->recv, g.tmp[(size_t)p->comm->rank], bytes, hipMemcpyDeviceToDevice, p->st));
        }
        break;
    }
    return fence(by_rank, &ncclComm::done);
}

ncclResult_t flush()
{
    std::vector<Pending> q;
    q.swap(t_queue);
    std::lock_guard<std::mutex> lock(g_mutex);
    // the queue may hold several collectives per communicator (none of nb_comm's groups does): run them in issue order
    std::vector<bool> used(q.size(), false);
    for (size_t i = 0; i < q.size(); ++i) {
        if (used[i]) continue;
        Group *g = q[i].comm->g.get();
        std::vector<Pending *> ops;
        std::vector<bool> seen(g->members.size(), false);
        for (size_t j = i; j < q.size(); ++j) {
            if (used[j] || q[j].comm->g.get() != g || seen[(size_t)q[j].comm->rank]) continue;
            seen[(size_t)q[j].comm->rank] = true;
            used[j] = true;
            ops.push_back(&q[j]);
        }
        const ncclResult_t rc = run_group(*g, ops);
        if (rc != ncclSuccess) return rc;
    }
    return ncclSuccess;
}

ncclResult_t enqueue(Kind kind, const void *send, void *recv, size_t count, ncclDataType_t ty, ncclComm_t comm, hipStream_t st)
{
    if (!comm || !comm->g || (count && (!send || !recv)) || (ty != ncclFloat && ty != ncclDouble)) return ncclInvalidArgument;
    {
        std::lock_guard<std::mutex> lock(g_mutex);
        if (g_fail_at >= 0 && g_calls++ == g_fail_at) return ncclInternalError;
    }
    t_queue.push_back(Pending{kind, send, recv, count, ty, comm, st});
    return t_depth == 0 ? flush() : ncclSuccess;
}

ncclComm *make_member(const std::shared_ptr<Group> &g, int rank, int dev)
{
    ncclComm *c = new ncclComm;
    c->g = g; c->rank = rank; c->dev = dev;
    if (hipSetDevice(dev) != hipSuccess || hipEventCreateWithFlags(&c->ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->mid, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->done, hipEventDisableTiming) != hipSuccess) {
        delete c;
        return nullptr;
    }
    g->members[(size_t)rank] = c;
    g->live += 1;
    return c;
}

}  // namespace

extern "C" {

// test hook: the k-th collective call from now on (0-based) returns ncclInternalError; k < 0: never
void nb_loopback_fail_after(long k)
{
    std::lock_guard<std::mutex> lock(g_mutex);
    g_calls = 0;
    g_fail_at = k;
}

ncclResult_t ncclGetVersion(int *version) { if (!version) return ncclInvalidArgument; *version = 20000; return ncclSuccess; }   // "2.0.0": a stand-in

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    if (!id) return ncclInvalidArgument;
    memset(id->internal, 0, sizeof id->internal);
    snprintf(id->internal, sizeof id->internal, "nb-loopback-transport");
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId, int rank)
{
    if (!comm || nranks != 1 || rank != 0) return ncclInvalidUsage;    // one process: several ranks come through ncclCommInitAll
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return ncclUnhandledCudaError;
    auto g = std::make_shared<Group>();
    g->members.assign(1, nullptr); g->tmp.assign(1, nullptr);
    *comm = make_member(g, 0, dev);
    return *comm ? ncclSuccess : ncclUnhandledCudaError;
}

ncclResult_t ncclCommInitAll(ncclComm_t *comms, int ndev, const int *devlist)
{
    if (!comms || ndev < 1 || ndev > 64) return ncclInvalidArgument;
    auto g = std::make_shared<Group>();
    g->members.assign((size_t)ndev, nullptr); g->tmp.assign((size_t)ndev, nullptr);
    for (int r = 0; r < ndev; ++r) {
        comms[r] = make_member(g, r, devlist ? devlist[r] : r);
        if (!comms[r]) return ncclUnhandledCudaError;
    }
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    if (!comm) return ncclSuccess;
    std::lock_guard<std::mutex> lock(g_mutex);
    (void)hipSetDevice(comm->dev);
    for (hipEvent_t e : {comm->ready, comm->mid, comm->done}) if (e) (void)hipEventDestroy(e);
    Group &g = *comm->g;
    if (g.tmp[(size_t)comm->rank]) { (void)hipFree(g.tmp[(size_t)comm->rank]); g.tmp[(size_t)comm->rank] = nullptr; }
    g.members[(size_t)comm->rank] = nullptr;
    g.live -= 1;
    if (g.live == 0) g.tmp_bytes = 0;
    delete comm;
    return ncclSuccess;
}

ncclResult_t ncclCommAbort(ncclComm_t comm) { return ncclCommDestroy(comm); }

const char *ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "loopback transport: unhandled HIP error";
    case ncclInternalError: return "loopback transport: injected internal error";
    case ncclInvalidArgument: return "loopback transport: invalid argument (kind / count / type differ between the ranks of a group)";
    case ncclInvalidUsage: return "loopback transport: invalid usage (a rank missing from the group, or several ranks without ncclCommInitAll)";
    default: return "loopback transport: error";
    }
}

ncclResult_t ncclGroupStart() { t_depth += 1; return ncclSuccess; }

ncclResult_t ncclGroupEnd()
{
    if (t_depth <= 0) return ncclInvalidUsage;
    t_depth -= 1;
    return t_depth == 0 ? flush() : ncclSuccess;
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t sendcount, ncclDataType_t ty, ncclComm_t comm, hipStream_t st)
{
    return enqueue(ALLGATHER, send, recv, sendcount, ty, comm, st);
}

ncclResult_t ncclReduceScatter(const void *send, void *recv, size_t recvcount, ncclDataType_t ty, ncclRedOp_t op, ncclComm_t comm, hipStream_t st)
{
    if (op != ncclSum) return ncclInvalidArgument;
    return enqueue(REDUCE_SCATTER, send, recv, recvcount, ty, comm, st);
}

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t ty, ncclRedOp_t op, ncclComm_t comm, hipStream_t st)
{
    if (op != ncclSum) return ncclInvalidArgument;
    return enqueue(ALLREDUCE, send, recv, count, ty, comm, st);
}

}  // extern "C"
