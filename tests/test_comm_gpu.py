"""GPU: the C-level RCCL exchange (nb_comm_*) on the one device a GPU box has.

RCCL takes one rank per device, so what can run here is a communicator of ONE rank — but through the real thing:
ncclCommInitAll / ncclCommInitRank, ncclAllGather / ncclReduceScatter / ncclAllReduce on the communication stream,
the event hand-overs, the library's schedule.  NB_FLAG_SHARD_SINGLE makes a handle run the SHARDED protocols with
one rank (every pair local, collectives degenerate to copies), so the whole split-step path is exercised, and the
result must be bit-identical to the same handle driven through the in-process exchange (nb_exchange_*), whose
multi-rank form is tested against the unsharded handle elsewhere (tests/test_headline_gpu.py, test_host_gpu.py).
World-size-2 call order and counts: tests/test_comm_schedule.py (CPU)."""
import ctypes as C
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT

import nbodysim_amd as nb
from nbodysim_amd import _lib as L
from nbodysim_amd.comm import Comm, unique_id

pytestmark = pytest.mark.gpu

EPS, DT, STEPS = 0.05, 1e-3, 5


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32 if a.dtype == np.float32 else np.uint64)


def _in_process(sim, steps):
    """The same single-rank handle through the in-process exchange: the reference the RCCL path must equal bit for bit."""
    lib = nb.load()
    arr = (C.c_void_p * 1)(sim._h)
    proto = sim.shard_protocol
    for _ in range(steps):
        sim.step_begin(DT)
        if proto == L.NB_SHARD_SYMMETRIC:
            sim.step_mid()
            L.check("nb_exchange_accelerations", lib.nb_exchange_accelerations(arr, 1))
        if proto == L.NB_SHARD_ALLREDUCE:
            L.check("nb_exchange_allreduce", lib.nb_exchange_allreduce(arr, 1))
        sim.step_finish()
        if proto != L.NB_SHARD_ALLREDUCE:
            L.check("nb_exchange_positions", lib.nb_exchange_positions(arr, 1))
    sim.wait()
    return sim.sync().copy()


@pytest.mark.parametrize("precision,dims,allreduce", [("fp32", 2, False), ("fp32", 2, True), ("fp64", 2, False), ("fp32", 3, False)])
def test_single_rank_sharded_protocols_through_rccl_equal_the_in_process_exchange(precision, dims, allreduce):
    n = 32768
    ic = nb.plummer_2d(n, 42) if dims == 2 else nb.plummer_3d(n, 42)
    kw = dict(eps=EPS, precision=precision, dims=dims, shard_rank=0, shard_world=1, shard_single=True, shard_allreduce=allreduce)
    with nb.Simulation(ic, **kw) as a:
        want_proto = L.NB_SHARD_ALLREDUCE if allreduce else L.NB_SHARD_SYMMETRIC
        assert a.shard_protocol == want_proto, a.describe()
        with pytest.raises(L.NBodyError):
            a.advance(1, DT)                                   # a sharded handle is not stepped with nb_step
        ref = _in_process(a, STEPS)
    with nb.Simulation(ic, **kw) as b:
        with Comm.all([b]) as comm:
            info = comm.info()
            assert info["protocol"] == want_proto and info["world"] == 1 and info["local_handles"] == 1 and info["rccl_version"] >= 20000
            comm.step(2, DT)
            comm.step(STEPS - 2, DT)
            comm.flush()                                       # compute stream waits for the last all-gather: nb_sync sees it
            got = b.sync().copy()
            assert b.frame == STEPS
            k, u = b.energy()
            (px, py, pz), _ = b.momentum()
            comm.wait()
    for f in ("pos", "vel", "acc"):
        assert np.array_equal(_bits(got[f]), _bits(ref[f])), f
    # and it is the physics of the plain handle (different launch split, same pairs): to rounding
    with nb.Simulation(ic, eps=EPS, precision=precision, dims=dims) as plain:
        plain.advance(STEPS, DT)
        p = plain.sync()
        k0, u0 = plain.energy()
        (qx, qy, qz), _ = plain.momentum()
    tol = 2e-6 if precision == "fp32" else 1e-7     # fp64 state comes back through the float Body record
    assert np.max(np.abs(got["pos"] - p["pos"])) <= tol * np.max(np.abs(p["pos"]))
    assert abs((k + u) - (k0 + u0)) <= (1e-5 if precision == "fp32" else 1e-10) * abs(k0 + u0)
    assert abs(px - qx) + abs(py - qy) + abs(pz - qz) < 1e-7


@pytest.mark.parametrize("n", [4096, 32768])
def test_unsharded_handle_in_a_one_rank_communicator_equals_nb_step(n):
    """protocol NONE: nb_comm_step = nb_step_begin + nb_step_finish + RCCL's one-rank all-gather of the positions;
    ncclCommInitRank form (unique id), as a one-process-per-GPU host would use it."""
    ic = nb.plummer_2d(n, 7)
    with nb.Simulation(ic, eps=EPS) as a:
        a.advance(STEPS, DT)
        want = a.sync().copy()
    with nb.Simulation(ic, eps=EPS) as b:
        uid = unique_id()
        assert len(uid) == L.NB_COMM_ID_BYTES and any(uid)
        with Comm.rank(b, uid, 0, 1) as comm:
            comm.step(STEPS, DT)
            comm.wait()
            got = b.sync().copy()
    for f in ("pos", "vel", "acc"):
        assert np.array_equal(_bits(got[f]), _bits(want[f])), f


def test_communicator_refuses_what_rccl_cannot_do():
    lib = nb.load()
    n = 16384
    ic = nb.plummer_2d(n, 3)
    sims = [nb.Simulation(ic, eps=EPS, i_begin=r * (n // 2), i_count=n // 2, shard_rank=r, shard_world=2) for r in range(2)]
    try:
        arr = (C.c_void_p * 2)(*[s._h for s in sims])
        assert lib.nb_comm_create_all(arr, 2) is None          # two ranks on one device
        assert L.last_error_code() == L.NB_EINVAL and "share device" in L.last_error()
        uid = unique_id()
        assert lib.nb_comm_create_rank(sims[1]._h, uid, 0, 2) is None   # rank 1's block presented as rank 0
        assert "rank" in L.last_error()
        with nb.Simulation(ic, eps=EPS) as plain:
            assert lib.nb_comm_create_rank(plain._h, uid, 0, 2) is None     # an unsharded handle in a 2-rank communicator
    finally:
        for s in sims:
            s.close()


def test_c_driver_rccl_single_rank_matches_plain_run(tmp_path):
    """`nbody_main -shards 1 -rccl`: plain C host, system RCCL (not PyTorch's), the library's step loop."""
    exe = ROOT / "build" / "nbody_main"
    if not exe.exists():
        subprocess.run(["make", "-C", str(ROOT / "nbodysim_amd" / "host")], check=True, capture_output=True)
    a, b = tmp_path / "rccl.nbd", tmp_path / "inproc.nbd"
    common = ["-n", "65536", "-s", "6", "-eps", "0.05"]
    r = subprocess.run([str(exe), *common, "-shards", "1", "-rccl", "-dump", str(a)], capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "exchange: RCCL" in r.stdout and "protocol=symmetric" in r.stdout and re.search(r"host enqueue [0-9.]+ us/step", r.stdout)
    r2 = subprocess.run([str(exe), *common, "-shards", "4", "-dump", str(b)], capture_output=True, text=True, timeout=240)
    assert r2.returncode == 0, r2.stdout + r2.stderr
    ba, fa, _ = nb.read_bodies(a)
    bb, fb, _ = nb.read_bodies(b)
    assert fa == fb == 6
    assert np.max(np.abs(ba["pos"] - bb["pos"])) <= 2e-6 * np.max(np.abs(bb["pos"]))


def test_c_driver_one_process_per_gpu_form(tmp_path):
    """`nbody_main -rank 0 -world 1 -idfile F`: the one-process-per-GPU launcher in plain C — RCCL id through a file,
    ncclCommInitRank, nb_comm_step — with the one rank this box allows; same trajectory as the in-process run."""
    exe = ROOT / "build" / "nbody_main"
    idf, a, b = tmp_path / "rccl.id", tmp_path / "mp.nbd", tmp_path / "inproc.nbd"
    common = ["-n", "65536", "-s", "4", "-eps", "0.05"]
    r = subprocess.run([str(exe), *common, "-rank", "0", "-world", "1", "-idfile", str(idf), "-dump", str(a)], capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "rank 0 of 1" in r.stdout and "protocol=symmetric" in r.stdout and not idf.exists()
    r2 = subprocess.run([str(exe), *common, "-shards", "2", "-dump", str(b)], capture_output=True, text=True, timeout=240)
    assert r2.returncode == 0, r2.stdout + r2.stderr
    ba, fa, _ = nb.read_bodies(str(a) + ".rank0")
    bb, fb, _ = nb.read_bodies(b)
    assert fa == 6 and fb == 4                       # the per-rank form runs 2 warm-up steps first
    r3 = subprocess.run([str(exe), *common, "-s", "6", "-shards", "2", "-dump", str(b)], capture_output=True, text=True, timeout=240)
    assert r3.returncode == 0, r3.stdout + r3.stderr
    bb, fb, _ = nb.read_bodies(b)
    assert fb == 6 and np.max(np.abs(ba["pos"] - bb["pos"])) <= 2e-6 * np.max(np.abs(bb["pos"]))
    bad = subprocess.run([str(exe), *common, "-rank", "1", "-world", "1", "-idfile", str(idf)], capture_output=True, text=True, timeout=60)
    assert bad.returncode != 0 and "0 <= R < P" in bad.stderr
    # a rank whose rank 0 never publishes THIS launch's id gives up at its deadline instead of waiting for ever; a stale
    # file of another launch (other nonce) does not satisfy it
    from nbodysim_amd.comm import id_publish
    id_publish(idf, 1, bytes(128))
    late = subprocess.run([str(exe), *common, "-rank", "1", "-world", "2", "-idfile", str(idf), "-nonce", "2", "-deadline", "2"],
                          capture_output=True, text=True, timeout=120)
    assert late.returncode != 0 and "belongs to another launch" in late.stderr


def test_c_loop_phases_add_up_to_the_step():
    """nb_comm_profile / nb_comm_phase_read: HIP events inside the library's loop, one real rank through RCCL.  The phases
    are consecutive intervals of the compute stream, so over a run that keeps the stream busy their sum is the step."""
    import time
    n, steps = 65536, 40
    ic = nb.plummer_2d(n, 42)
    for allreduce in (False, True):
        with nb.Simulation(ic, eps=EPS, shard_rank=0, shard_world=1, shard_single=True, shard_allreduce=allreduce) as s:
            with Comm.all([s]) as comm:
                comm.step(5, DT)
                comm.wait()
                comm.profile(True)
                t0 = time.perf_counter()
                comm.step(steps, DT)
                comm.wait()
                wall_ms = (time.perf_counter() - t0) / steps * 1e3
                ph = comm.phases(0)
                assert ph["steps"] == steps
                total = sum(ph[k] for k in L.NB_PH_NAMES)
                assert abs(total - wall_ms) <= 0.05 * wall_ms, (ph, wall_ms)
                if allreduce:
                    assert ph["cross"] == 0.0 and ph["ag_wait"] == 0.0 and ph["local"] > 0.8 * total      # one launch of all pairs, then the all-reduce
                else:
                    assert ph["local"] > 0.8 * total                   # one rank: every pair is local
                comm.profile(False)
                comm.step(3, DT)                                       # profiling off again: no marks, still steps
                comm.wait()
                assert comm.phases(0)["steps"] == 0 and s.frame == 5 + steps + 3


def test_one_rank_communicator_over_a_ragged_size(tmp_path):
    """n that is neither a multiple of the tile nor of anything else: `nbody_main -shards 1 -rccl` and, in-process, three
    ragged shards (ceil(n / 3) particles, a shorter last block) — same trajectory as the plain handle."""
    exe = ROOT / "build" / "nbody_main"
    a, b, c = tmp_path / "a.nbd", tmp_path / "b.nbd", tmp_path / "c.nbd"
    common = ["-n", "10007", "-eps", "0.05"]
    # the plain form runs 2 warm-up steps before its -s steps: 5 frames everywhere
    for extra, out in ((["-s", "5", "-shards", "1", "-rccl"], a), (["-s", "5", "-shards", "3", "-no-symmetry"], b), (["-s", "3"], c)):
        r = subprocess.run([str(exe), *common, *extra, "-dump", str(out)], capture_output=True, text=True, timeout=240)
        assert r.returncode == 0, r.stdout + r.stderr
    (ba, fa, _), (bb, fb, _), (bc, fc, _) = (nb.read_bodies(x) for x in (a, b, c))
    assert fa == fb == fc == 5
    assert np.max(np.abs(ba["pos"] - bc["pos"])) <= 2e-6 * np.max(np.abs(bc["pos"]))
    assert np.max(np.abs(bb["pos"] - bc["pos"])) <= 2e-6 * np.max(np.abs(bc["pos"]))
