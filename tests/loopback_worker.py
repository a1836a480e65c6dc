"""Child process of tests/test_comm_loopback_gpu.py (TEST INFRASTRUCTURE).

RCCL is loaded once per process, so the test-only loopback transport (tests/loopback_rccl.hip) needs a process of its own:
this script names it with nb_debug_comm_transport BEFORE anything loads a transport, then runs the library's C-level step
loop (nb_comm_create_all + nb_comm_step) with 2, 3 and 4 members on ONE GPU and compares every rank's state, bit for bit,
with the same handles driven through the library's in-process exchange (nb_exchange_*).  One JSON line per case on stdout.
"""
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
# this whole process runs on the -DNB_TEST_HOOKS build (tests/libnbody_hip_testhooks.so): the product library has no
# nb_debug_comm_transport and can load nothing but RCCL.  The binding takes the path from NBODY_HIP_LIB at its first load.
import hooks  # noqa: E402
os.environ["NBODY_HIP_LIB"] = str(hooks.HOOKS_PATH)
import nbodysim_amd as nb  # noqa: E402
from nbodysim_amd import _lib as L  # noqa: E402
from nbodysim_amd.comm import Comm  # noqa: E402

STUB = Path(__file__).resolve().parent / "libnb_loopback_rccl.so"
EPS, DT = 0.05, 1e-3


def make_handles(ic, world, allreduce=False, **kw):
    n = ic.shape[0]
    stride = -(-n // world)
    sims = []
    for r in range(world):
        lo, cnt = (0, n) if allreduce else (r * stride, min(stride, n - r * stride))
        sims.append(nb.Simulation(ic, eps=EPS, i_begin=lo, i_count=cnt, shard_rank=r, shard_world=world, shard_allreduce=allreduce, device=0, **kw))
    return sims


def in_process(sims, steps):
    lib = nb.load()
    arr = (C.c_void_p * len(sims))(*[s._h for s in sims])
    proto = sims[0].shard_protocol
    for _ in range(steps):
        for s in sims:
            s.step_begin(DT)
        if proto == L.NB_SHARD_SYMMETRIC:
            for s in sims:
                s.step_mid()
            L.check("nb_exchange_accelerations", lib.nb_exchange_accelerations(arr, len(sims)))
        if proto == L.NB_SHARD_ALLREDUCE:
            L.check("nb_exchange_allreduce", lib.nb_exchange_allreduce(arr, len(sims)))
        for s in sims:
            s.step_finish()
        if proto != L.NB_SHARD_ALLREDUCE:
            L.check("nb_exchange_positions", lib.nb_exchange_positions(arr, len(sims)))
    for s in sims:
        s.wait()
    return [s.sync().copy() for s in sims]


def via_comm(sims, steps, profile=False):
    with Comm.all(sims) as comm:
        info = comm.info()
        if profile:
            comm.profile(True)
        comm.step(2, DT)
        comm.step(steps - 2, DT)
        comm.flush()
        out = [s.sync().copy() for s in sims]
        comm.wait()
        phases = [comm.phases(k) for k in range(len(sims))] if profile else None
    return out, info, phases


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def case(name, n, world, steps=6, allreduce=False, dims=2, profile=False, **kw):
    ic = nb.plummer_2d(n, 42) if dims == 2 else nb.plummer_3d(n, 42)
    a = make_handles(ic, world, allreduce, dims=dims, **kw)
    proto = a[0].shard_protocol
    ref = in_process(a, steps)
    for s in a:
        s.close()
    b = make_handles(ic, world, allreduce, dims=dims, **kw)
    got, info, phases = via_comm(b, steps, profile)
    frames = [s.frame for s in b]
    late = [int(s.sym_info()["items_late"]) for s in b]
    for s in b:
        s.close()
    same = all(np.array_equal(bits(g[f]), bits(r[f])) for g, r in zip(got, ref) for f in ("pos", "vel", "acc"))
    worst = max(float(np.max(np.abs(g["pos"] - r["pos"]))) for g, r in zip(got, ref))
    out = {"case": name, "n": n, "world": world, "protocol": proto, "comm": info, "frames": frames, "bit_identical": bool(same),
           "max_abs_pos_diff": worst, "owned": [int(g.shape[0]) for g in got], "late_items": late}
    if phases:
        out["phases"] = phases
    print(json.dumps(out), flush=True)


def failure_case():
    """A collective fails half-way through a step (rank 1's reduce-scatter of the second step): the error surfaces, the
    communicator refuses further steps, and destroying it does not wait for peers that will never arrive."""
    stub = C.CDLL(str(STUB))
    stub.nb_loopback_fail_after.argtypes = [C.c_long]
    n, world = 16384, 2
    ic = nb.plummer_2d(n, 42)
    sims = make_handles(ic, world)
    lib = nb.load()
    arr = (C.c_void_p * world)(*[s._h for s in sims])
    h = lib.nb_comm_create_all(arr, world)
    assert h, L.last_error()
    # symmetric protocol, per step and rank: one reduce-scatter + one all-gather = 4 collective calls per step
    stub.nb_loopback_fail_after(4 + 1)
    rc = lib.nb_comm_step(h, DT, 3)
    err = L.last_error()
    rc2 = lib.nb_comm_step(h, DT, 1)
    rc3 = lib.nb_comm_wait(h)
    t0 = time.time()
    lib.nb_comm_destroy(h)
    took = time.time() - t0
    stub.nb_loopback_fail_after(-1)
    for s in sims:
        s.close()
    print(json.dumps({"case": "failure", "rc_step": rc, "error": err, "rc_next_step": rc2, "rc_wait": rc3, "destroy_seconds": took}), flush=True)


def main():
    lib = nb.load()
    lib.nb_debug_comm_transport.restype, lib.nb_debug_comm_transport.argtypes = hooks.DEBUG_PROTOTYPES["nb_debug_comm_transport"]
    L.check("nb_debug_comm_transport", lib.nb_debug_comm_transport(str(STUB).encode()))
    v = C.c_int()
    L.check("nb_comm_available", lib.nb_comm_available(C.byref(v)))
    print(json.dumps({"case": "transport", "version": v.value}), flush=True)
    case("symmetric fp32 world 2", 16384, 2, profile=True)
    case("symmetric fp32 world 4", 32768, 4)
    case("symmetric+late fp32 world 4", 32768, 4, sym_late_us=40.0)
    case("allreduce fp32 world 2", 16384, 2, allreduce=True)
    case("allreduce fp32 world 4", 32768, 4, allreduce=True)
    case("allgather fp32 world 2", 8192, 2, symmetry=False)
    case("allgather fp32 world 4", 8192, 4, symmetry=False)
    case("allgather ragged fp32 world 3", 10007, 3, symmetry=False)
    case("allgather ragged sequential quake world 3", 3001, 3, order="sequential", rsqrt="quake")
    case("symmetric fp64 world 2", 16384, 2, precision="fp64")
    case("allreduce fp64 world 4", 32768, 4, allreduce=True, precision="fp64")
    case("symmetric 3-D fp32 world 2", 16384, 2, dims=3)
    case("allgather 3-D fp32 world 4", 8192, 4, dims=3, symmetry=False)
    # the driver's node run at its largest: N = 262 144 over EIGHT ranks — the plan the library builds there (12 chunk pairs per item,
    # the late items ON by default from 8 ranks), the symmetric and the all-reduce protocol, 4 steps
    case("symmetric fp32 world 8 headline", 262144, 8, steps=4)
    case("allreduce fp32 world 8 headline", 262144, 8, steps=4, allreduce=True)
    failure_case()


if __name__ == "__main__":
    main()
