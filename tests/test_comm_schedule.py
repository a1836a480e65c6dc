"""CPU: the schedule of the C-level RCCL exchange (nb_comm_step) as data — call order, streams, events and element
counts for world size 2, in both shapes (one process per GPU: 1 handle per process; one process driving both ranks:
2 handles, grouped collectives).  No GPU, no RCCL: `nb_debug_comm_schedule` returns exactly the list the executor
in nbodysim_amd/csrc/nb_comm.cpp interprets.

The checks are happens-before checks on the stream/event graph of three consecutive steps: every operation is
ordered after the previous one on its (handle, stream), and a WAIT is ordered after the latest RECORD of its event
issued before it.  What must be ordered (data hazards) and what must NOT be (the overlap north_star asks for:
all-gather of step k beside the local force of step k + 1) are both asserted."""
import numpy as np
import pytest

import hooks
from nbodysim_amd import _lib as L

N, WORLD, RPE = 65536, 2, 2
BLOCK, FULL = N // WORLD * RPE, N * RPE
COMPUTE, COMM = 0, 1


def three_steps(protocol, handles):
    first = hooks.comm_schedule(protocol, handles, BLOCK, FULL, False)
    steady = hooks.comm_schedule(protocol, handles, BLOCK, FULL, True)
    ops = []
    for k, sched in enumerate((first, steady, steady)):
        for o in sched:
            ops.append(dict(step=k, kind=int(o["kind"]), h=int(o["handle"]), stream=int(o["stream"]), ev=int(o["event"]), count=int(o["count"])))
    return ops


def happens_before(ops):
    """Reachability matrix of the issue-order graph (ops are few: dense boolean closure)."""
    m = len(ops)
    reach = np.zeros((m, m), bool)
    last_on = {}        # (h, stream) -> index of the last op there
    last_record = {}    # (h, event)  -> index of the last RECORD
    for i, o in enumerate(ops):
        if o["kind"] in (L.NB_OP_GROUP_START, L.NB_OP_GROUP_END):
            continue
        key = (o["h"], o["stream"])
        preds = []
        if key in last_on:
            preds.append(last_on[key])
        if o["kind"] == L.NB_OP_WAIT:
            assert (o["h"], o["ev"]) in last_record, f"WAIT on an event never recorded: {o}"
            preds.append(last_record[(o["h"], o["ev"])])
        for p in preds:
            reach[i] |= reach[p]
            reach[i, p] = True
        last_on[key] = i
        if o["kind"] == L.NB_OP_RECORD:
            last_record[(o["h"], o["ev"])] = i
    return reach


def find(ops, kind, h, step):
    hits = [i for i, o in enumerate(ops) if o["kind"] == kind and o["h"] == h and o["step"] == step]
    assert len(hits) == 1, (kind, h, step, hits)
    return hits[0]


def ordered(reach, a, b):
    return bool(reach[b, a])


@pytest.mark.parametrize("handles", [1, 2])
def test_streams_counts_and_grouping(handles):
    for protocol in (L.NB_SHARD_NONE, L.NB_SHARD_ALLGATHER, L.NB_SHARD_SYMMETRIC, L.NB_SHARD_ALLREDUCE):
        ops = three_steps(protocol, handles)
        depth = 0
        for o in ops:
            k = o["kind"]
            if k == L.NB_OP_GROUP_START:
                assert handles > 1 and depth == 0
                depth += 1
            elif k == L.NB_OP_GROUP_END:
                depth -= 1
                assert depth == 0
            elif k in (L.NB_OP_BEGIN, L.NB_OP_MID, L.NB_OP_FINISH):
                assert o["stream"] == COMPUTE and depth == 0
            elif k in (L.NB_OP_ALLGATHER, L.NB_OP_REDUCE_SCATTER, L.NB_OP_ALLREDUCE):
                assert o["stream"] == COMM                         # collectives never sit on the compute stream
                assert depth == (1 if handles > 1 else 0)            # one process, several ranks: one ncclGroup per collective
                assert o["count"] == (FULL if k == L.NB_OP_ALLREDUCE else BLOCK)
        assert depth == 0
        per_step = {kind: sum(1 for o in ops if o["kind"] == kind and o["step"] == 1 and o["h"] == 0)
                    for kind in (L.NB_OP_BEGIN, L.NB_OP_MID, L.NB_OP_FINISH, L.NB_OP_ALLGATHER, L.NB_OP_REDUCE_SCATTER, L.NB_OP_ALLREDUCE)}
        want = {L.NB_SHARD_SYMMETRIC: (1, 1, 1, 1, 1, 0), L.NB_SHARD_ALLREDUCE: (1, 0, 1, 0, 0, 1)}.get(protocol, (1, 0, 1, 1, 0, 0))
        assert tuple(per_step.values()) == want, (protocol, per_step)


@pytest.mark.parametrize("handles", [1, 2])
def test_allgather_protocol_order_and_overlap(handles):
    """north_star's protocol: the all-gather of step k follows its kick/drift, precedes the remote force of step
    k + 1, and is NOT ordered against the local force of step k + 1 (they overlap on two streams)."""
    ops = three_steps(L.NB_SHARD_ALLGATHER, handles)
    hb = happens_before(ops)
    for h in range(handles):
        for k in (0, 1):
            fin, ag = find(ops, L.NB_OP_FINISH, h, k), find(ops, L.NB_OP_ALLGATHER, h, k)
            beg1, fin1 = find(ops, L.NB_OP_BEGIN, h, k + 1), find(ops, L.NB_OP_FINISH, h, k + 1)
            assert ordered(hb, fin, ag) and ordered(hb, ag, fin1)
            assert not ordered(hb, ag, beg1) and not ordered(hb, beg1, ag)
    # first step of a run: nothing in flight, so nothing to wait for
    assert not any(o["kind"] == L.NB_OP_WAIT and o["stream"] == COMPUTE and o["step"] == 0 for o in ops)


@pytest.mark.parametrize("handles", [1, 2])
def test_symmetric_protocol_order_and_overlap(handles):
    ops = three_steps(L.NB_SHARD_SYMMETRIC, handles)
    hb = happens_before(ops)
    for h in range(handles):
        for k in (0, 1):
            mid, rs, fin, ag = (find(ops, kk, h, k) for kk in (L.NB_OP_MID, L.NB_OP_REDUCE_SCATTER, L.NB_OP_FINISH, L.NB_OP_ALLGATHER))
            beg1, mid1 = find(ops, L.NB_OP_BEGIN, h, k + 1), find(ops, L.NB_OP_MID, h, k + 1)
            assert ordered(hb, mid, rs) and ordered(hb, rs, fin) and ordered(hb, fin, ag) and ordered(hb, ag, mid1)
            assert ordered(hb, rs, mid1)                            # acc_full is rewritten only after it was reduced
            assert not ordered(hb, ag, beg1) and not ordered(hb, beg1, ag)   # local pairs beside the all-gather


@pytest.mark.parametrize("handles", [1, 2])
def test_allreduce_protocol_order(handles):
    ops = three_steps(L.NB_SHARD_ALLREDUCE, handles)
    hb = happens_before(ops)
    assert not any(o["kind"] in (L.NB_OP_ALLGATHER, L.NB_OP_REDUCE_SCATTER, L.NB_OP_MID) for o in ops)
    for h in range(handles):
        for k in (0, 1):
            beg, ar, fin = (find(ops, kk, h, k) for kk in (L.NB_OP_BEGIN, L.NB_OP_ALLREDUCE, L.NB_OP_FINISH))
            assert ordered(hb, beg, ar) and ordered(hb, ar, fin) and ordered(hb, fin, find(ops, L.NB_OP_BEGIN, h, k + 1))


def test_bad_arguments_are_refused():
    lib = hooks.lib()
    import ctypes as C
    cnt = C.c_size_t()
    assert lib.nb_debug_comm_schedule(9, 1, 1, 1, 0, None, 0, C.byref(cnt)) == L.NB_EINVAL
    assert lib.nb_debug_comm_schedule(L.NB_SHARD_SYMMETRIC, 0, 1, 1, 0, None, 0, C.byref(cnt)) == L.NB_EINVAL
    assert lib.nb_comm_step(None, 0.0, 1) == L.NB_EINVAL and lib.nb_comm_wait(None) == L.NB_EINVAL
    assert lib.nb_comm_create_all(None, 0) is None
