"""CPU: nothing in the multi-GPU start-up may wait without an end.

* a rank stuck inside a start-up candidate ends the job on EVERY rank, within the deadline, with a non-zero exit that names
  the candidate and lists the timings gathered so far (nbodysim_amd.dist.Watchdog / time_candidates; world 2 over gloo, fresh
  child processes);
* the number of steps — i.e. of collectives — a rank enqueues in bench.py's sustained stretch is a function of
  rank-independent inputs only (the round-3 hang: step counts sized from each rank's own clock);
* the RCCL id of a file-based launch carries the launch's nonce: a stale file of a crashed run is ignored, not consumed
  (nb_comm_id_publish / nb_comm_id_await).
"""
import ctypes as C
import importlib.util
import os
import socket
import subprocess
import sys
import threading
import time
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
TESTS = Path(__file__).resolve().parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def load_bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", ROOT / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    sys.modules["bench_under_test"] = mod
    spec.loader.exec_module(mod)
    return mod


# ---------------------------------------------------------------------------------------------------------------------
# 1. a stalled candidate
# ---------------------------------------------------------------------------------------------------------------------
def _tune_worker(rank, world, port, stall_rank, stall_in, deadline):
    """Child process: the start-up timing over gloo with stand-in candidates (two collectives each); `stall_rank` never
    comes back from candidate `stall_in`."""
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import datetime

    import torch
    import torch.distributed as dist

    from nbodysim_amd.dist import time_candidates

    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))

    def run_one(name, local):
        t0 = time.perf_counter()
        if name == stall_in and rank == stall_rank:
            time.sleep(3600)                           # this rank is stuck (a dead peer, a collective that never completes ...)
        t = torch.ones(4)
        dist.all_reduce(t)                             # the others wait for it HERE
        dist.barrier()
        # stand-in timings far apart (and the measured part capped), so that the verdict does not depend on how busy this CPU is
        return min(time.perf_counter() - t0, 0.05) + {"allgather": 3.0, "allreduce": 2.0, "symmetric": 1.0}[name]

    best, job = time_candidates(["allgather", "allreduce", "symmetric"], run_one, None, deadline, rank,
                                prefer=("symmetric", "allreduce", "allgather"), log=lambda m: print(m, file=sys.stderr, flush=True))
    print(f"rank {rank} chose {best} {sorted(job)}", flush=True)
    dist.destroy_process_group()


def _spawn_tune(stall_rank, stall_in, deadline):
    port = _free_port()
    code = ("import sys; sys.path.insert(0, {t!r}); import test_hang_guards as t; "
            "t._tune_worker({{rank}}, 2, {port}, {sr}, {si!r}, {dl})").format(t=str(TESTS), port=port, sr=stall_rank, si=stall_in, dl=deadline)
    return [subprocess.Popen([sys.executable, "-c", code.format(rank=r)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            for r in range(2)]


def test_a_rank_stuck_in_a_candidate_ends_every_rank_naming_it():
    t0 = time.time()
    procs = _spawn_tune(stall_rank=1, stall_in="allreduce", deadline=4.0)
    outs = [p.communicate(timeout=90) for p in procs]
    took = time.time() - t0
    for r, (p, (out, err)) in enumerate(zip(procs, outs)):
        assert p.returncode == 3, (r, p.returncode, err[-600:])                       # EXIT_DEADLINE on BOTH ranks: the stuck one and the waiting one
        assert f"rank {r}: deadline of 4 s expired while timing the start-up candidate 'allreduce'" in err, err[-600:]
        assert "so far: {'allgather':" in err                                       # the table gathered before the stall
        assert "chose" not in out
    assert "[tune] rank 0: allgather:" in outs[0][1]                                # each result flushed as it completes (rank 0)
    assert took < 60


def test_the_same_timing_completes_and_agrees_when_nobody_stalls():
    procs = _spawn_tune(stall_rank=-1, stall_in="", deadline=30.0)
    outs = [p.communicate(timeout=90) for p in procs]
    for r, (p, (out, err)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, err[-600:]
        assert f"rank {r} chose symmetric ['allgather', 'allreduce', 'symmetric']" in out


def test_watchdog_is_inert_inside_its_deadline_and_can_be_disabled():
    sys.path.insert(0, str(ROOT))
    from nbodysim_amd.dist import EXIT_DEADLINE, Watchdog
    fired = []
    with Watchdog(30.0, "nothing", exit_fn=fired.append):
        pass
    with Watchdog(0.0, "disabled", exit_fn=fired.append):
        time.sleep(0.05)
    assert fired == []
    with Watchdog(0.05, "a short nap", report=lambda: {"x": 1}, rank=5, exit_fn=fired.append):
        time.sleep(0.4)
    assert fired == [EXIT_DEADLINE]


# ---------------------------------------------------------------------------------------------------------------------
# 2. rank-independent collective counts (the round-3 hang class)
# ---------------------------------------------------------------------------------------------------------------------
def _count_worker(rank, world, port, out_dir):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    b = load_bench()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        calls = {"advance": 0, "steps": 0}

        def advance(k, dt):                            # every step of a sharded run is a collective on every rank
            calls["advance"] += 1
            for _ in range(k):
                t = torch.ones(1)
                dist.all_reduce(t)
                calls["steps"] += 1
            time.sleep(0.02 * (1 + 4 * rank))          # rank 1's clock runs much slower: its measured times differ

        local_hint = [40.0, 110.0][rank]               # ms per step as THIS rank measured it (skewed on purpose)
        out = b.sustained_rate(advance, lambda: None, 1e-3, local_hint, min_seconds=2.0, sampler_period=0.5, world=world)
        res = {"steps": calls["steps"], "advance_calls": calls["advance"], "reported": out["steps"],
               "from_local_hint": b.sustained_steps(local_hint, 2.0)}
        gathered = [None] * world
        dist.all_gather_object(gathered, res)
        if rank == 0:
            import json
            (Path(out_dir) / "counts.json").write_text(json.dumps(gathered))
    finally:
        dist.destroy_process_group()


def test_sustained_stretch_enqueues_the_same_number_of_collectives_on_every_rank(tmp_path):
    import json

    import torch.multiprocessing as mp
    mp.spawn(_count_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)     # a mismatch would deadlock -> spawn times out / fails
    r0, r1 = json.loads((tmp_path / "counts.json").read_text())
    assert r0["steps"] == r1["steps"] == r0["reported"] == r1["reported"]
    assert r0["advance_calls"] == r1["advance_calls"] == 1                      # one batch: no measured time feeds back into a count
    # ... and the guard is not vacuous: sized from each rank's OWN hint (the pre-fix logic) the counts differ
    assert r0["from_local_hint"] != r1["from_local_hint"]
    assert r0["steps"] == min(r0["from_local_hint"], r1["from_local_hint"])     # = the count of the MAX-reduced (slowest) hint


def test_single_rank_stretch_may_adapt_but_a_multi_rank_one_never_does():
    b = load_bench()
    n = {"calls": 0}

    def advance(k, dt):
        n["calls"] += 1
        time.sleep(0.01)
    out = b.sustained_rate(advance, lambda: None, 1e-3, 50.0, min_seconds=0.05, sampler_period=0.5, world=1)
    assert n["calls"] >= 2 and out["steps"] >= 40          # single GPU: further batches until the stretch is long enough
    n["calls"] = 0
    out = b.sustained_rate(advance, lambda: None, 1e-3, 50.0, min_seconds=0.05, sampler_period=0.5, world=4, reduce_max=lambda v, w: v)
    assert n["calls"] == 1 and out["steps"] == b.sustained_steps(50.0, 0.05)


# ---------------------------------------------------------------------------------------------------------------------
# 3. the id file of a one-process-per-GPU launch
# ---------------------------------------------------------------------------------------------------------------------
def test_id_file_handshake_ignores_a_stale_file(tmp_path):
    sys.path.insert(0, str(ROOT))
    import nbodysim_amd as nb
    from nbodysim_amd import _lib as L
    from nbodysim_amd.comm import id_await, id_publish

    nb.load()
    path = tmp_path / "rccl.id"
    old, new = bytes(range(128)), bytes(reversed(range(128)))
    id_publish(path, 1111, old)                                 # a launch that crashed before rank 0 could unlink its file
    assert id_await(path, 1111, 1000) == old
    t0 = time.time()
    with pytest.raises(L.NBodyError) as e:                      # the next launch (another nonce) must NOT consume it
        id_await(path, 2222, 300)
    assert e.value.code == L.NB_EIO and "belongs to another launch" in str(e.value) and 0.25 < time.time() - t0 < 5
    # rank 0 of the new launch publishes late: the waiting rank takes the NEW id
    threading.Timer(0.3, lambda: id_publish(path, 2222, new)).start()
    assert id_await(path, 2222, 5000) == new
    assert not list(tmp_path.glob("rccl.id.tmp*"))              # published atomically, nothing left behind
    # a pre-round-4 file (the bare 128 bytes) or a truncated one is never an id
    path.write_bytes(old)
    with pytest.raises(L.NBodyError):
        id_await(path, 0, 100)
    missing = tmp_path / "never.id"
    with pytest.raises(L.NBodyError) as e:
        id_await(missing, 7, 100)
    assert "belongs to another launch" not in str(e.value)
    lib = nb.load()
    assert lib.nb_comm_id_publish(None, 1, old) == L.NB_EINVAL and lib.nb_comm_id_await(str(path).encode(), 1, None, 10) == L.NB_EINVAL
