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
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "nb_internal.h"
#include "nb_sched.h"
#include "nbody.h"
#include "nbody_debug.h"

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
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;      // optional: ends a communicator whose peers may never arrive
    bool loopback = false;                                // the test-only in-process transport (tests/loopback_rccl.hip): several ranks per device
};

Rccl g_rccl;
std::mutex g_rccl_mutex;
#ifdef NB_TEST_HOOKS
std::string g_transport_path;                             // nb_debug_comm_transport (test build only): load THIS library instead of librccl.so.1
#endif

template <typename F> bool sym(void *lib, const char *name, F &fn)
{
    fn = reinterpret_cast<F>(dlsym(lib, name));
    return fn != nullptr;
}

int load_rccl()
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.lib) return NB_OK;
    void *lib = nullptr;
#ifdef NB_TEST_HOOKS
    if (!g_transport_path.empty()) {
        lib = dlopen(g_transport_path.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!lib) return nb_fail(NB_ENODEVICE, "nb_comm: the transport named with nb_debug_comm_transport is not loadable (%s)", dlerror());
    } else
#endif
    {
        // The RCCL that sits on the SAME HIP / HSA runtime as this library: a process can hold two ROCm stacks (PyTorch bundles
        // its own libamdhip64 / libhsa-runtime64 / librccl under torch/lib), and a bare dlopen("librccl.so.1") returns whichever
        // copy was loaded first — if that is the copy of the OTHER stack, its runtime has never seen a device
        // ("no ROCm-capable device is detected" out of ncclCommInitAll; found when `import torch` came after this library's
        // first HIP call).  So: look next to the libamdhip64 this library's HIP calls resolve to, then fall back to the SONAME.
        Dl_info hip_at;
        if (dladdr((const void *)&hipGetDeviceCount, &hip_at) && hip_at.dli_fname) {
            std::string dir(hip_at.dli_fname);
            const size_t slash = dir.rfind('/');
            if (slash != std::string::npos) {
                dir.resize(slash + 1);
                lib = dlopen((dir + "librccl.so.1").c_str(), RTLD_NOW | RTLD_LOCAL);
                if (!lib) lib = dlopen((dir + "librccl.so").c_str(), RTLD_NOW | RTLD_LOCAL);
            }
        }
        if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    }
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
    (void)sym(lib, "ncclCommAbort", r.CommAbort);
    r.loopback = dlsym(lib, "nb_loopback_transport") != nullptr;
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

// per-step timing marks of one member (nb_comm_profile): events on its COMPUTE stream around every compute-stream
// operation of the schedule, so the phases are what the compute stream sees, waits for the collectives included
enum : int { PH_LOCAL = 0, PH_AG_WAIT = 1, PH_CROSS = 2, PH_REDUCE = 3, PH_FINISH = 4, PH_COUNT = 5 };
constexpr int MARKS_PER_STEP = 8;          // start + at most 6 compute-stream operations per step
constexpr size_t MARK_RING = 128;          // steps in flight with marks: the host blocks on the oldest when the ring is full
struct StepMarks { hipEvent_t ev[MARKS_PER_STEP]; int phase[MARKS_PER_STEP]; int used = 0; };

struct Member {
    nb_sim *sim = nullptr;
    int dev = 0;
    hipStream_t compute = nullptr, comm = nullptr;
    hipEvent_t ev[EV_COUNT] = {nullptr, nullptr, nullptr, nullptr};
    ncclComm_t nccl = nullptr;
    std::vector<StepMarks> ring;           // allocated by nb_comm_profile(on)
    size_t ring_head = 0, ring_count = 0;  // oldest un-harvested step, steps with marks in flight
    double phase_ms[PH_COUNT] = {0, 0, 0, 0, 0};
    uint64_t phase_steps = 0;
};

}  // namespace

struct nb_comm {
    std::vector<Member> m;
    int protocol = NB_SHARD_NONE;
    int world = 1;
    int reals_per_element = 2, bytes_per_real = 4;
    uint64_t block_reals = 0, full_reals = 0;
    bool ag_pending = false;
    bool failed = false;                        // a step failed half-way: the peers may wait for collectives this process never joined
    bool profile = false;
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
    // all-gather protocol: blocks of stride = ceil(n / world) particles, the last one shorter when world does not divide n;
    // the collective moves `stride` rows per rank whatever the block holds, so the replicas must have world * stride rows
    // (the library allocates them that way for a sharded handle; caller-owned ones say so in nb_params.pos_rows)
    const size_t stride = (n + (size_t)world - 1) / (size_t)world;
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
        } else if (proto == NB_SHARD_SYMMETRIC) {
            // equal blocks in rank order: what the in-place ncclAllGather (send = recv + rank * count) and the
            // reduce-scatter (equal receive counts) need
            if (n % (size_t)world != 0 || ic != n / (size_t)world || ib != (size_t)ranks[k] * ic)
                return nb_fail(NB_EINVAL, "nb_comm: rank %d must own the block [rank * n/world, +n/world) (n = %zu, world = %d; has [%zu, +%zu))",
                               ranks[k], n, world, ib, ic);
        } else {
            const size_t want_b = (size_t)ranks[k] * stride;
            if (want_b >= n || ib != want_b || ic != (n - want_b < stride ? n - want_b : stride))
                return nb_fail(NB_EINVAL, "nb_comm: rank %d must own the block [rank * ceil(n/world), +...) (n = %zu, world = %d, stride %zu; has [%zu, +%zu))",
                               ranks[k], n, world, stride, ib, ic);
            if (nb_pos_rows(s) < (size_t)world * stride)
                return nb_fail(NB_EINVAL, "nb_comm: world = %d does not divide n = %zu: the position replicas need %zu rows (equal counts per rank), "
                                          "this handle's hold %zu (caller-owned buffers: say so in nb_params.pos_rows)", world, n, (size_t)world * stride, nb_pos_rows(s));
        }
    }
    c->protocol = proto;
    c->world = world;
    c->reals_per_element = rpe;
    c->bytes_per_real = bpr;
    c->full_reals = (uint64_t)n * (uint64_t)rpe;
    c->block_reals = (uint64_t)(proto == NB_SHARD_SYMMETRIC ? n / (size_t)world : stride) * (uint64_t)rpe;
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

void free_marks(Member &mb)
{
    for (StepMarks &sm : mb.ring)
        for (hipEvent_t e : sm.ev) if (e) (void)hipEventDestroy(e);
    mb.ring.clear();
    mb.ring_head = mb.ring_count = 0;
}

void release(nb_comm *c)
{
    if (!c) return;
    for (Member &mb : c->m) {
        (void)hipSetDevice(mb.dev);
        if (c->failed) {
            // the peers may be waiting inside a collective this process never joined: synchronising the communication
            // stream or a plain ncclCommDestroy could block for ever — abort the communicator instead; a transport without
            // ncclCommAbort keeps its communicator (a deliberate LEAK, like the communication stream below: nbody.h says so)
            if (mb.nccl && g_rccl.CommAbort) (void)g_rccl.CommAbort(mb.nccl);
        } else {
            if (mb.comm) (void)hipStreamSynchronize(mb.comm);
            if (mb.nccl && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(mb.nccl);
        }
        free_marks(mb);
        for (hipEvent_t e : mb.ev) if (e) (void)hipEventDestroy(e);
        if (mb.comm && !c->failed) (void)hipStreamDestroy(mb.comm);
    }
    delete c;
}

// add the oldest marked step of a member to its phase sums (blocks until that step has run)
int harvest_one(Member &mb)
{
    StepMarks &sm = mb.ring[mb.ring_head];
    if (sm.used > 1) {
        HIPC(hipEventSynchronize(sm.ev[sm.used - 1]));
        for (int k = 1; k < sm.used; ++k) {
            float ms = 0.f;
            HIPC(hipEventElapsedTime(&ms, sm.ev[k - 1], sm.ev[k]));
            if (sm.phase[k] >= 0 && sm.phase[k] < PH_COUNT) mb.phase_ms[sm.phase[k]] += (double)ms;
        }
        mb.phase_steps += 1;
    }
    sm.used = 0;
    mb.ring_head = (mb.ring_head + 1) % mb.ring.size();
    mb.ring_count -= 1;
    return NB_OK;
}

int phase_of(const nb_comm_op &o)
{
    switch (o.kind) {
    case OP_BEGIN:  return PH_LOCAL;
    case OP_MID:    return PH_CROSS;
    case OP_FINISH: return PH_FINISH;
    case OP_WAIT:   return o.stream == ST_COMPUTE ? (o.event == EV_AG ? PH_AG_WAIT : o.event == EV_RED ? PH_REDUCE : -1) : -1;
    default:        return -1;
    }
}

int run_schedule_inner(nb_comm *c, const std::vector<nb_comm_op> &ops, float dt, bool &group_open)
{
    const ncclDataType_t ty = c->bytes_per_real == 8 ? ncclDouble : ncclFloat;
    const size_t esz = (size_t)c->reals_per_element * (size_t)c->bytes_per_real;
    int bound = -1;
    std::vector<StepMarks *> marks(c->m.size(), nullptr);
    if (c->profile) {
        for (size_t k = 0; k < c->m.size(); ++k) {
            Member &mb = c->m[k];
            if (mb.ring.empty()) continue;
            if (bound != mb.dev) { HIPC(hipSetDevice(mb.dev)); bound = mb.dev; }
            if (mb.ring_count == mb.ring.size()) { const int rc = harvest_one(mb); if (rc) return rc; }
            StepMarks *sm = &mb.ring[(mb.ring_head + mb.ring_count) % mb.ring.size()];
            mb.ring_count += 1;
            sm->used = 1; sm->phase[0] = -1;
            HIPC(hipEventRecord(sm->ev[0], mb.compute));
            marks[k] = sm;
        }
    }
    for (const nb_comm_op &o : ops) {
        if (o.kind == OP_GROUP_START) { NCCLC(g_rccl.GroupStart()); group_open = true; continue; }
        if (o.kind == OP_GROUP_END) { group_open = false; NCCLC(g_rccl.GroupEnd()); continue; }
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
        StepMarks *sm = marks[(size_t)o.handle];
        const int ph = phase_of(o);
        if (sm && ph >= 0 && sm->used < MARKS_PER_STEP) {
            sm->phase[sm->used] = ph;
            HIPC(hipEventRecord(sm->ev[sm->used], mb.compute));
            sm->used += 1;
        }
    }
    return NB_OK;
}

// One step.  If an operation fails half-way the communicator is marked FAILED (an open ncclGroup is closed first): the
// peers may already wait inside collectives this process will never issue, so nothing may block on the communication
// stream any more — nb_comm_destroy then aborts the RCCL communicator instead of synchronising (release()).
int run_schedule(nb_comm *c, const std::vector<nb_comm_op> &ops, float dt)
{
    bool group_open = false;
    const int rc = run_schedule_inner(c, ops, dt, group_open);
    if (rc) {
        if (group_open) (void)g_rccl.GroupEnd();
        c->failed = true;
    }
    return rc;
}

}  // namespace

#ifdef NB_TEST_HOOKS
// Name the library that provides the nccl* entry points (instead of librccl.so.1).  TEST BUILD ONLY (include/nbody_debug.h):
// tests/loopback_rccl.hip is an in-process transport over device buffers that lets several ranks share ONE device, so the
// executor above can be run with 2 and 4 members on a one-GPU box.  Must be called before the first nb_comm_* call that loads
// the transport.  The product library has no way to load anything but RCCL.
extern "C" int nb_debug_comm_transport(const char *path)
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.lib) return nb_fail(NB_ESTATE, "nb_debug_comm_transport: a transport is already loaded in this process");
    g_transport_path = path ? path : "";
    return NB_OK;
}
#endif

// NB_OK iff the transport can be loaded in this process (and its version); every rank of a one-process-per-GPU job
// checks this and the ranks AGREE on it before any of them enters the blocking ncclCommInitRank.
extern "C" int nb_comm_available(int *rccl_version)
{
    if (load_rccl()) return nb_last_error_code();
    if (rccl_version) { int v = 0; (void)g_rccl.GetVersion(&v); *rccl_version = v; }
    return NB_OK;
}

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
        for (int j = 0; j < k && !g_rccl.loopback; ++j)
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
    if (c->failed) return nb_fail(NB_ESTATE, "nb_comm_step: an earlier step failed half-way; destroy the communicator and the handles");
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
    if (c->failed) return nb_fail(NB_ESTATE, "nb_comm_flush: the communicator has failed");
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
    if (c && !c->failed) (void)nb_comm_wait(c);
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

// Per-phase timing inside the library's loop: HIP events on each handle's compute stream around every compute-stream
// operation of a step (MARK_RING steps deep; when the ring is full the host waits for the oldest marked step, so with
// profiling on it runs at most that far ahead).  Off by default: the loop then records no timing event at all.
extern "C" int nb_comm_profile(nb_comm *c, int on)
{
    if (!c) return nb_fail(NB_EINVAL, "nb_comm_profile: NULL communicator");
    for (Member &mb : c->m) {
        HIPC(hipSetDevice(mb.dev));
        while (mb.ring_count) { const int rc = harvest_one(mb); if (rc) return rc; }
        if (on && mb.ring.empty()) {
            mb.ring.resize(MARK_RING);
            for (StepMarks &sm : mb.ring)
                for (hipEvent_t &e : sm.ev) { e = nullptr; HIPC(hipEventCreate(&e)); }
        }
        if (!on) free_marks(mb);
    }
    c->profile = on != 0;
    return NB_OK;
}

extern "C" int nb_comm_phase_read(nb_comm *c, int handle, double *phase_ms, uint64_t *steps, int reset)
{
    if (!c || handle < 0 || (size_t)handle >= c->m.size()) return nb_fail(NB_EINVAL, "nb_comm_phase_read: bad arguments");
    Member &mb = c->m[(size_t)handle];
    HIPC(hipSetDevice(mb.dev));
    while (mb.ring_count) { const int rc = harvest_one(mb); if (rc) return rc; }
    if (phase_ms) for (int k = 0; k < PH_COUNT; ++k) phase_ms[k] = mb.phase_ms[k];
    if (steps) *steps = mb.phase_steps;
    if (reset) { for (double &v : mb.phase_ms) v = 0.0; mb.phase_steps = 0; }
    return NB_OK;
}

#ifdef NB_TEST_HOOKS
// Host-only view of the schedule (no GPU, no RCCL): the operations of one step, in issue order.  Test build only.
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
#endif
