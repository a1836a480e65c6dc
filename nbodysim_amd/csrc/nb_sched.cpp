// nb_sched.cpp — see nb_sched.h.  Host only.
#include "nb_sched.h"

namespace nbk {

namespace {
struct Emit {
    std::vector<nb_comm_op> &ops;
    int handles;
    void op(int32_t kind, int32_t h, int32_t stream, int32_t event, uint64_t count)
    {
        nb_comm_op o;
        o.kind = kind; o.handle = h; o.stream = stream; o.event = event; o.count = count;
        ops.push_back(o);
    }
    // the same call for every handle of this process
    void each(int32_t kind, int32_t stream = ST_COMPUTE, int32_t event = -1)
    {
        for (int h = 0; h < handles; ++h) op(kind, h, stream, event, 0);
    }
    // compute -> comm hand-over, the collective of all handles as one group, comm -> compute hand-back
    void collective(int32_t kind, uint64_t count, int32_t ready_ev, int32_t done_ev, bool wait_on_compute)
    {
        each(OP_RECORD, ST_COMPUTE, ready_ev);
        each(OP_WAIT, ST_COMM, ready_ev);
        if (handles > 1) op(OP_GROUP_START, -1, ST_COMM, -1, 0);
        for (int h = 0; h < handles; ++h) op(kind, h, ST_COMM, -1, count);
        if (handles > 1) op(OP_GROUP_END, -1, ST_COMM, -1, 0);
        each(OP_RECORD, ST_COMM, done_ev);
        if (wait_on_compute) each(OP_WAIT, ST_COMPUTE, done_ev);
    }
};
}  // namespace

void build_comm_schedule(int protocol, int handles, uint64_t block_reals, uint64_t full_reals, bool ag_pending,
                         std::vector<nb_comm_op> &ops)
{
    Emit e{ops, handles < 1 ? 1 : handles};
    switch (protocol) {
    case NB_SHARD_ALLREDUCE:
        // every handle holds all positions: its share of the pairs -> acc_full | all-reduce | kick, drift of all n
        e.each(OP_BEGIN);
        e.collective(OP_ALLREDUCE, full_reals, EV_ACC, EV_RED, true);
        e.each(OP_FINISH);
        break;
    case NB_SHARD_SYMMETRIC:
        // pairs inside the own block (overlaps the all-gather still in flight on the comm stream) | wait for it |
        // share of the cross-block pairs -> acc_full | reduce-scatter | kick, drift | all-gather of the new block
        e.each(OP_BEGIN);
        if (ag_pending) e.each(OP_WAIT, ST_COMPUTE, EV_AG);
        e.each(OP_MID);
        e.collective(OP_REDUCE_SCATTER, block_reals, EV_ACC, EV_RED, true);
        e.each(OP_FINISH);
        e.collective(OP_ALLGATHER, block_reals, EV_POS, EV_AG, false);
        break;
    case NB_SHARD_ALLGATHER:
    case NB_SHARD_NONE:
    default:
        // force from the own j-block (overlaps the all-gather in flight) | wait for it | force from the other blocks,
        // kick, drift | all-gather of the new block.  An unsharded handle (one rank) runs the same sequence: its
        // all-gather is RCCL's one-rank no-op.
        e.each(OP_BEGIN);
        if (ag_pending) e.each(OP_WAIT, ST_COMPUTE, EV_AG);
        e.each(OP_FINISH);
        e.collective(OP_ALLGATHER, block_reals, EV_POS, EV_AG, false);
        break;
    }
}

}  // namespace nbk
