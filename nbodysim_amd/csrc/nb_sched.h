// nb_sched.h — the per-step schedule of a sharded run as DATA (plain C++, no HIP, no RCCL).
//
// The reference's only fan-out is std::async over contiguous i-chunks inside one process
// (Nbodysim/headers/Simulation.hpp:180-213).  Here the fan-out is one handle per GPU, and one step of the sharded
// run is a fixed sequence of operations on two streams per handle — the handle's COMPUTE stream (force, kick,
// drift) and a COMMUNICATION stream (RCCL collectives) — ordered by events only: no host synchronisation inside
// the step loop.  build_comm_schedule() writes that sequence down once; nb_comm.cpp interprets it with the real
// HIP / RCCL calls, and the CPU tests (tests/test_comm_schedule.py, the ASan fuzz) check its order and counts.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <vector>

#include "nbody_debug.h"   /* nb_comm_op, NB_OP_*, NB_EV_*: the schedule is data (the types live beside the hook that exposes it) */

namespace nbk {

// nb_comm_op.kind (mirrored in include/nbody.h as NB_OP_*)
enum : int32_t {
    OP_BEGIN = 0,           // nb_step_begin(handle, dt)                        compute stream
    OP_MID = 1,             // nb_step_mid(handle)                              compute stream
    OP_FINISH = 2,          // nb_step_finish(handle)                           compute stream
    OP_RECORD = 3,          // hipEventRecord(event, stream)
    OP_WAIT = 4,            // hipStreamWaitEvent(stream, event)
    OP_ALLGATHER = 5,       // ncclAllGather of the owned position block, in place in the CURRENT replica   comm stream
    OP_REDUCE_SCATTER = 6,  // ncclReduceScatter(acc_full -> acc_owned, sum)                                comm stream
    OP_ALLREDUCE = 7,       // ncclAllReduce(acc_full, in place, sum)                                       comm stream
    OP_GROUP_START = 8,     // ncclGroupStart(): the collectives of all handles of ONE process are one group
    OP_GROUP_END = 9,
};
enum : int32_t { ST_COMPUTE = 0, ST_COMM = 1 };
// events of one handle
enum : int32_t {
    EV_POS = 0,   // compute: the owned block of the new CURRENT replica is written      -> comm may gather it
    EV_AG = 1,    // comm:    the all-gather into the CURRENT replica is complete        -> compute may read remote blocks
    EV_ACC = 2,   // compute: this handle's partial accelerations (acc_full) are complete -> comm may reduce them
    EV_RED = 3,   // comm:    the reduction is complete                                   -> compute may kick and drift
    EV_COUNT = 4
};

// One step of `handles` handles driven by this process (1 for one-process-per-GPU) in `protocol`
// (NB_SHARD_NONE / _ALLGATHER / _SYMMETRIC / _ALLREDUCE); elements are counted in reals:
//   block_reals = i_count x reals-per-element  (all-gather send count, reduce-scatter receive count)
//   full_reals  = n x reals-per-element         (all-reduce count)
// `ag_pending` = an all-gather of the previous step is in flight (false for the very first step of a run: the
// replicas start complete).  Appends to `ops`.
void build_comm_schedule(int protocol, int handles, uint64_t block_reals, uint64_t full_reals, bool ag_pending,
                         std::vector<nb_comm_op> &ops);

}  // namespace nbk
