// nb_comm.cpp — the C-level multi-GPU exchange: RCCL collectives on a communication stream per handle, ordered
// against the handle's compute stream by HIP events.  No host synchronisation inside the step loop.
//
// What it replaces: the reference's only fan-out, `std::async` over contiguous i-chunks inside Simulation::attract
// (Nbodysim/headers/Simulation.hpp:180-213).  north_star: "host code stays in C ... RCCL all-gather of positions
// over xGMI each step, overlapped with local-tile force compute on a second HIP stream" (SURVEY §7 step 7, §8e).
//
// Layering: this file uses ONLY the public C ABI of include/nbody.h for the handles (nb_step_begin / _mid / _finish,
// nb_stream, nb_pos_buffer, nb_acc_buffer, ...) plus HIP events and RCCL — it is the loop a C host would write
// itself, kept in the library so that every host gets the same, tested ordering.  The sequence
// of one step is DATA (nb_sched.h): the executor below interprets it.
//
// RCCL is resolved at first use with dlopen("librccl.so.1"): a process that already carries an RCCL (PyTorch
// bundles one under the same SONAME) keeps using that one, and hosts that never call nb_comm_* never load it.
// If it cannot be loaded nb_comm_create_* fails loudly (NB_ENODEVICE); there is no fallback transport here — the
// in-process exchanges (nb_exchange_*) are a different, explicit API.
#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "nb_internal.h"
#include "nb_sched.h"
#include "nbody.h"

using namespace nbk;

namespace {

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
};

Rccl g_rccl;
std::mutex g_rccl_mutex;

template <typename F> bool sym(void *lib, const char *name, F &fn)
{
    fn = reinterpret_cast<F>(dlsym(lib, name));
    return fn != nullptr;
}

int load_rccl()
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.lib) return NB_OK;
    void *lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!lib) return nb_fail(NB_ENODEVICE, "nb_comm: RCCL is not loadable (%s); the C-level exchange has no other transport", dlerror());
    Rccl r;
    r.lib = lib;
    const bool ok = sym(lib, "ncclGetUniqueId", r.GetUniqueId) && sym(lib, "ncclCommInitRank", r.CommInitRank) &&
                    sym(lib, "ncclCommInitAll", r.CommInitAll) && sym(lib, "ncclCommDestroy", r.CommDestroy) &&
                    sym(lib, "ncclAllGather", r.AllGather) && sym(lib, "ncclReduceScatter", r.ReduceScatter) &&
                    sym(lib, "ncclAllReduce", r.AllReduce) && sym(lib, "ncclGroupStart", r.GroupStart) &&
                    sym(lib, "ncclGroupEnd", r.GroupEnd) && sym(lib, "ncclGetErrorString", r.GetErrorString) &&
                    sym(lib, "ncclGetVersion", r.GetVersion);
    if (!ok) { dlclose(lib); return nb_fail(NB_ENODEVICE, "nb_comm: librccl lacks an expected nccl* symbol"); }
    g_rccl = r;
    return NB_OK;
}

#define HIPC(call)                                                                                                     \
    do {                                                                                                               \
        hipError_t e_ = (call);                                                                                        \
        if (e_ != hipSuccess) return nb_fail(NB_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define NCCLC(call)                                                                                                    \
    do {                                                                                                               \
        ncclResult_t r_ = (call);                                                                                      \
        if (r_ != ncclSuccess) return nb_fail(NB_EHIP, "%s failed: %s (%s:%d)", #call, g_rccl.GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

struct Member {
    nb_sim *sim = nullptr;
    int dev = 0;
    hipStream_t compute = nullptr, comm = nullptr;
    hipEvent_t ev[EV_COUNT] = {nullptr, nullptr, nullptr, nullptr};
    ncclComm_t nccl = nullptr;
};

}  // namespace

struct nb_comm {
    std::vector<Member> m;
    int protocol = NB_SHARD_NONE;
    int world = 1;
    int reals_per_element = 2, bytes_per_real = 4;
    uint64_t block_reals = 0, full_reals = 0;
    bool ag_pending = false;
    uint64_t steps = 0;
    std::vector<nb_comm_op> first, steady;      // schedule of the first step (no all-gather in flight) and of all later ones
};

namespace {

// common checks + per-member streams and events
int adopt(nb_comm *c, nb_sim *const *sims, int count, int world, const int *ranks)
{
    if (!sims || count < 1 || world < count) return nb_fail(NB_EINVAL, "nb_comm: bad handle list");
    const int proto = nb_shard_protocol(sims[0]);
    int rpe = 0, bpr = 0;
    if (nb_element_layout(sims[0], &rpe, &bpr)) return nb_last_error_code();
    const size_t n = nb_count(sims[0]);
    for (int k = 0; k < count; ++k) {
        nb_sim *s = sims[k];
        int r2 = 0, b2 = 0;
        if (!s || nb_shard_protocol(s) != proto || nb_count(s) != n || nb_element_layout(s, &r2, &b2) || r2 != rpe || b2 != bpr)
            return nb_fail(NB_EINVAL, "nb_comm: the handles must be the ranks of ONE sharded run (same n, precision, dims, protocol)");
        const size_t ic = nb_owned_count(s), ib = nb_owned_begin(s);
        int sw = 0;
        const int sr = nb_shard_rank(s, &sw);
        if (sw > 0 && proto != NB_SHARD_NONE && (sw != world || sr != ranks[k]))   // the pair split of the symmetric protocols is keyed on these
            return nb_fail(NB_EINVAL, "nb_comm: handle created as rank %d of %d joins the communicator as rank %d of %d", sr, sw, ranks[k], world);
        if (proto == NB_SHARD_ALLREDUCE || proto == NB_SHARD_NONE) {
            if (ic != n) return nb_fail(NB_EINVAL, "nb_comm: a replicated / unsharded handle owns all n particles");
            if (proto == NB_SHARD_NONE && world != 1) return nb_fail(NB_EINVAL, "nb_comm: an unsharded handle forms a communicator of one rank only");
        } else if (n % (size_t)world != 0 || ic != n / (size_t)world || ib != (size_t)ranks[k] * ic) {
            // equal blocks in rank order: what the in-place ncclAllGather (send = recv + rank * count) and the
            // reduce-scatter (equal receive counts) need
            return nb_fail(NB_EINVAL, "nb_comm: rank %d must own the block [rank * n/world, +n/world) (n = %zu, world = %d; has [%zu, +%zu))",
                           ranks[k], n, world, ib, ic);
        }
    }
    c->protocol = proto;
    c->world = world;
    c->reals_per_element = rpe;
    c->bytes_per_real = bpr;
    c->full_reals = (uint64_t)n * (uint64_t)rpe;
    c->block_reals = (uint64_t)nb_owned_count(sims[0]) * (uint64_t)rpe;
    if (proto == NB_SHARD_ALLREDUCE || proto == NB_SHARD_NONE) c->block_reals = c->full_reals / (uint64_t)world;
    c->m.resize((size_t)count);
    for (int k = 0; k < count; ++k) {
        Member &mb = c->m[(size_t)k];
        mb.sim = sims[k];
        mb.compute = (hipStream_t)nb_stream(sims[k]);
        mb.dev = nb_device(sims[k]);
        HIPC(hipSetDevice(mb.dev));
        HIPC(hipStreamCreateWithFlags(&mb.comm, hipStreamNonBlocking));
        for (int e = 0; e < EV_COUNT; ++e) HIPC(hipEventCreateWithFlags(&mb.ev[e], hipEventDisableTiming));
    }
    build_comm_schedule(proto, count, c->block_reals, c->full_reals, false, c->first);
    build_comm_schedule(proto, count, c->block_reals, c->full_reals, true, c->steady);
    return NB_OK;
}

void release(nb_comm *c)
{
    if (!c) return;
    for (Member &mb : c->m) {
        (void)hipSetDevice(mb.dev);
        if (mb.comm) (void)hipStreamSynchronize(mb.comm);
        if (mb.nccl && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(mb.nccl);
        for (hipEvent_t e : mb.ev) if (e) (void)hipEventDestroy(e);
        if (mb.comm) (void)hipStreamDestroy(mb.comm);
    }
    delete c;
}

int run_schedule(nb_comm *c, const std::vector<nb_comm_op> &ops, float dt)
{
    const ncclDataType_t ty = c->bytes_per_real == 8 ? ncclDouble : ncclFloat;
    const size_t esz = (size_t)c->reals_per_element * (size_t)c->bytes_per_real;
    int bound = -1;
    for (const nb_comm_op &o : ops) {
        if (o.kind == OP_GROUP_START) { NCCLC(g_rccl.GroupStart()); continue; }
        if (o.kind == OP_GROUP_END) { NCCLC(g_rccl.GroupEnd()); continue; }
        Member &mb = c->m[(size_t)o.handle];
        if (bound != mb.dev) { HIPC(hipSetDevice(mb.dev)); bound = mb.dev; }
        hipStream_t st = o.stream == ST_COMM ? mb.comm : mb.compute;
        int rc = NB_OK;
        switch (o.kind) {
        case OP_BEGIN:  rc = nb_step_begin(mb.sim, dt); break;
        case OP_MID:    rc = nb_step_mid(mb.sim); break;
        case OP_FINISH: rc = nb_step_finish(mb.sim); break;
        case OP_RECORD: HIPC(hipEventRecord(mb.ev[o.event], st)); break;
        case OP_WAIT:   HIPC(hipStreamWaitEvent(st, mb.ev[o.event], 0)); break;
        case OP_ALLGATHER: {
            // in place: the owned block already sits at its final position in the replica (send = recv + rank * count)
            char *replica = (char *)nb_pos_buffer(mb.sim, NB_POS_CURRENT);
            const char *mine = replica + nb_owned_begin(mb.sim) * esz;
            if (c->protocol == NB_SHARD_NONE) mine = replica;
            NCCLC(g_rccl.AllGather(mine, replica, (size_t)o.count, ty, mb.nccl, st));
            break;
        }
        case OP_REDUCE_SCATTER:
            NCCLC(g_rccl.ReduceScatter(nb_acc_buffer(mb.sim, 0), nb_acc_buffer(mb.sim, 1), (size_t)o.count, ty, ncclSum, mb.nccl, st));
            break;
        case OP_ALLREDUCE:
            NCCLC(g_rccl.AllReduce(nb_acc_buffer(mb.sim, 0), nb_acc_buffer(mb.sim, 0), (size_t)o.count, ty, ncclSum, mb.nccl, st));
            break;
        default:
            return nb_fail(NB_ESTATE, "nb_comm: unknown schedule op %d", o.kind);
        }
        if (rc) return rc;
    }
    return NB_OK;
}

}  // namespace

extern "C" int nb_comm_unique_id(void *id_out)
{
    static_assert(NB_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "nb_comm id = ncclUniqueId");
    if (!id_out) return nb_fail(NB_EINVAL, "nb_comm_unique_id: NULL argument");
    if (load_rccl()) return nb_last_error_code();
    ncclUniqueId id;
    NCCLC(g_rccl.GetUniqueId(&id));
    memcpy(id_out, id.internal, NB_COMM_ID_BYTES);
    return NB_OK;
}

extern "C" nb_comm *nb_comm_create_rank(nb_sim *s, const void *id, int rank, int world)
{
    nb_clear_error();
    if (!s || !id || world < 1 || rank < 0 || rank >= world) { nb_fail(NB_EINVAL, "nb_comm_create_rank: bad arguments"); return nullptr; }
    if (load_rccl()) return nullptr;
    nb_comm *c = new (std::nothrow) nb_comm;
    if (!c) { nb_fail(NB_ENOMEM, "nb_comm_create_rank: out of host memory"); return nullptr; }
    nb_sim *one[1] = {s};
    if (adopt(c, one, 1, world, &rank)) { release(c); return nullptr; }
    ncclUniqueId uid;
    memcpy(uid.internal, id, NB_COMM_ID_BYTES);
    auto init = [&]() -> int {
        HIPC(hipSetDevice(c->m[0].dev));
        NCCLC(g_rccl.CommInitRank(&c->m[0].nccl, world, uid, rank));
        return NB_OK;
    };
    if (init()) { release(c); return nullptr; }
    return c;
}

extern "C" nb_comm *nb_comm_create_all(nb_sim *const *sims, int count)
{
    nb_clear_error();
    if (!sims || count < 1 || count > 64) { nb_fail(NB_EINVAL, "nb_comm_create_all: 1..64 handles"); return nullptr; }
    if (load_rccl()) return nullptr;
    nb_comm *c = new (std::nothrow) nb_comm;
    if (!c) { nb_fail(NB_ENOMEM, "nb_comm_create_all: out of host memory"); return nullptr; }
    std::vector<int> ranks((size_t)count), devs((size_t)count);
    for (int k = 0; k < count; ++k) ranks[(size_t)k] = k;
    if (adopt(c, sims, count, count, ranks.data())) { release(c); return nullptr; }
    for (int k = 0; k < count; ++k) {
        devs[(size_t)k] = c->m[(size_t)k].dev;
        for (int j = 0; j < k; ++j)
            if (devs[(size_t)j] == devs[(size_t)k]) {
                nb_fail(NB_EINVAL, "nb_comm_create_all: handles %d and %d share device %d — RCCL takes one rank per device "
                                   "(several handles on one device exchange with nb_exchange_*)", j, k, devs[(size_t)k]);
                release(c);
                return nullptr;
            }
    }
    std::vector<ncclComm_t> comms((size_t)count, nullptr);
    auto init = [&]() -> int {
        NCCLC(g_rccl.CommInitAll(comms.data(), count, devs.data()));
        return NB_OK;
    };
    if (init()) { release(c); return nullptr; }
    for (int k = 0; k < count; ++k) c->m[(size_t)k].nccl = comms[(size_t)k];
    return c;
}

extern "C" int nb_comm_step(nb_comm *c, float dt, int nsteps)
{
    if (!c) return nb_fail(NB_EINVAL, "nb_comm_step: NULL communicator");
    if (nsteps < 0) return nb_fail(NB_EINVAL, "nb_comm_step: nsteps < 0");
    for (int k = 0; k < nsteps; ++k) {
        const int rc = run_schedule(c, c->ag_pending ? c->steady : c->first, dt);
        if (rc) return rc;
        c->ag_pending = c->protocol != NB_SHARD_ALLREDUCE;
        c->steps += 1;
    }
    return NB_OK;
}

// Compute streams wait for the collectives still in flight (stream-ordered; the host does not block), so that
// nb_sync / nb_energy / nb_wait on the handles see complete replicas.
extern "C" int nb_comm_flush(nb_comm *c)
{
    if (!c) return nb_fail(NB_EINVAL, "nb_comm_flush: NULL communicator");
    if (!c->ag_pending) return NB_OK;
    for (Member &mb : c->m) {
        HIPC(hipSetDevice(mb.dev));
        HIPC(hipStreamWaitEvent(mb.compute, mb.ev[EV_AG], 0));
    }
    c->ag_pending = false;
    return NB_OK;
}

extern "C" int nb_comm_wait(nb_comm *c)
{
    if (!c) return nb_fail(NB_EINVAL, "nb_comm_wait: NULL communicator");
    int rc = nb_comm_flush(c);
    if (rc) return rc;
    for (Member &mb : c->m) {
        HIPC(hipSetDevice(mb.dev));
        HIPC(hipStreamSynchronize(mb.comm));
        if ((rc = nb_wait(mb.sim))) return rc;
    }
    return NB_OK;
}

extern "C" void nb_comm_destroy(nb_comm *c)
{
    if (c) (void)nb_comm_wait(c);
    release(c);
}

extern "C" int nb_comm_info(const nb_comm *c, int *protocol, int *world, int *local_handles, int *rccl_version)
{
    if (!c) return nb_fail(NB_EINVAL, "nb_comm_info: NULL communicator");
    if (protocol) *protocol = c->protocol;
    if (world) *world = c->world;
    if (local_handles) *local_handles = (int)c->m.size();
    if (rccl_version) { int v = 0; if (g_rccl.GetVersion) (void)g_rccl.GetVersion(&v); *rccl_version = v; }
    return NB_OK;
}

// Host-only view of the schedule (no GPU, no RCCL): the operations of one step, in issue order.
extern "C" int nb_debug_comm_schedule(int protocol, int handles, uint64_t block_reals, uint64_t full_reals, int ag_pending,
                                      nb_comm_op *ops_out, size_t cap, size_t *count)
{
    if (handles < 1 || handles > 64 || !count) return nb_fail(NB_EINVAL, "nb_debug_comm_schedule: bad arguments");
    if (protocol < NB_SHARD_NONE || protocol > NB_SHARD_ALLREDUCE) return nb_fail(NB_EINVAL, "nb_debug_comm_schedule: bad protocol %d", protocol);
    std::vector<nb_comm_op> ops;
    build_comm_schedule(protocol, handles, block_reals, full_reals, ag_pending != 0, ops);
    *count = ops.size();
    if (ops_out) memcpy(ops_out, ops.data(), (ops.size() < cap ? ops.size() : cap) * sizeof(nb_comm_op));
    return NB_OK;
}
