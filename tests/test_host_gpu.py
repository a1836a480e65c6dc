"""GPU: the compiled hosts over the C ABI — the plain-C driver, the C++ `Simulation`
adaptor driven with the reference's simulation_thread pattern (main.cpp:612-635) —
and the page-locked nb_sync path."""
import os
import re
import subprocess

import ctypes as C

import numpy as np
import pytest

from conftest import ROOT, flat_from_bodies

import nbodysim_amd as nb
from nbodysim_amd import _lib as L

pytestmark = pytest.mark.gpu


def _host_program(name):
    """build/<name>; built by __graft_entry__.build(), rebuilt here (plain gcc/g++ against the in-tree library) if absent."""
    exe = ROOT / "build" / name
    if not exe.exists():
        subprocess.run(["make", "-C", str(ROOT / "nbodysim_amd" / "host")], check=True, capture_output=True, timeout=300)
    assert exe.exists()
    return exe



def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_c_driver_runs_and_its_dump_matches_the_python_path(tmp_path):
    exe = _host_program("nbody_main")
    assert exe.exists(), "build() must produce build/nbody_main"
    dump = tmp_path / "c.nbd"
    r = subprocess.run([str(exe), "-n", "4096", "-s", "10", "-sync-every", "1", "-dump", str(dump)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"frame=(\d+)", r.stdout)
    assert m and int(m.group(1)) == 12          # 2 warm-up + 10 timed steps
    drift = float(re.search(r"drift=([-+0-9.e]+)", r.stdout).group(1))
    assert abs(drift) < 1e-3
    back, frame, p = nb.read_bodies(dump)
    assert frame == 12 and abs(p.eps - 0.01) < 1e-9
    with nb.Simulation(nb.plummer_2d(4096, 42), eps=0.01) as sim:
        sim.advance(12, 1e-3)
        want = flat_from_bodies(sim.sync())
    assert np.array_equal(bits(flat_from_bodies(back)), bits(want))   # same library, same bits


def test_cxx_adaptor_with_reference_caller_pattern():
    exe = _host_program("sim_thread_example")
    assert exe.exists()
    r = subprocess.run([str(exe), "2048"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"frame=(\d+) bodies=(\d+) body0=\(([-0-9.]+), ([-0-9.]+)\)", r.stdout)
    assert m and int(m.group(1)) == 10 and int(m.group(2)) == 2048
    from nbodysim_amd import simulation as S
    S.SIMULATION_DT = 1e-3
    try:
        with nb.Simulation(nb.plummer_2d(2048, 42), eps=0.05) as sim:
            for _ in range(10):
                sim.step()
            assert abs(sim.bodies["pos"][0, 0] - float(m.group(3))) < 2e-6
            assert abs(sim.bodies["pos"][0, 1] - float(m.group(4))) < 2e-6
    finally:
        S.SIMULATION_DT = 0.01


def test_sync_into_page_locked_host_memory_is_identical():
    """nb_sync into the three kinds of destination: a pageable numpy array (pipelined staging), a block from
    nb_host_alloc, and a caller-owned page-aligned range registered with nb_host_register (both DMA'd directly)."""
    import mmap
    lib = nb.load()
    n = 10000                                     # 640 000 bytes: not a whole number of pages -> register a rounded-up mapping
    ic = nb.plummer_2d(n, 3)
    with nb.Simulation(ic, eps=0.05) as sim:
        sim.advance(3, 1e-3)
        plain = sim.sync().tobytes()
        with nb.PinnedBodies(n) as pb:
            L.check("nb_sync", lib.nb_sync(sim._h, pb.array.ctypes.data))
            assert pb.array.tobytes() == plain
            pos = np.empty((n, 2), np.float32)
            L.check("nb_sync_positions", lib.nb_sync_positions(sim._h, pos.ctypes.data))
            assert np.array_equal(pos, pb.array["pos"])
        page = mmap.PAGESIZE
        size = -(-n * 64 // page) * page
        m = mmap.mmap(-1, size)                   # anonymous mapping: page-aligned, whole pages, owned by this test
        arr = np.frombuffer(m, dtype=L.BODY_DTYPE, count=n)
        L.check("nb_host_register", lib.nb_host_register(arr.ctypes.data, size))
        try:
            L.check("nb_sync", lib.nb_sync(sim._h, arr.ctypes.data))
            assert arr.tobytes() == plain
            assert lib.nb_host_register(arr.ctypes.data, size) == L.NB_ESTATE          # already registered
        finally:
            L.check("nb_host_unregister", lib.nb_host_unregister(arr.ctypes.data))
        del arr
        m.close()


def test_host_register_refuses_ranges_that_are_not_whole_pages():
    """a registration pins whole pages, so an unaligned heap array (whose first and last page
    also hold its neighbours' data) is refused, as is unregistering something that was never registered; the array
    is still a legal nb_sync / nb_upload / nb_snapshot_begin argument (staged)."""
    lib = nb.load()
    n = 40000
    ic = nb.plummer_2d(n, 3)
    out = nb.bodies_array(n)
    if out.ctypes.data % 4096 == 0:               # numpy happened to hand out an aligned block: shift by one record
        out = nb.bodies_array(n + 1)[1:]
    assert lib.nb_host_register(out.ctypes.data, out.nbytes) == L.NB_EINVAL
    assert "whole number" in L.last_error()
    assert lib.nb_host_unregister(out.ctypes.data) == L.NB_EINVAL
    assert lib.nb_host_free(out.ctypes.data) == L.NB_EINVAL
    with nb.Simulation(ic, eps=0.05) as sim:      # register-refused array right next to a pageable upload source: the r2 sequence
        sim.advance(2, 1e-3)
        want = sim.sync().copy()
        sim.upload(want)
        L.check("nb_sync", lib.nb_sync(sim._h, out.ctypes.data))
        for f in ("pos", "vel", "acc", "mass", "radius"):
            assert np.array_equal(out[f].view(np.uint32), want[f].view(np.uint32)), f


@pytest.mark.parametrize("n,protocol,late_us", [(8192, "allgather", None), (65536, "symmetric", None), (65536, "symmetric", "40"),
                                                (65536, "allreduce", None)])
def test_c_driver_with_in_process_shards_matches_single_handle(tmp_path, n, protocol, late_us):
    """`nbody_main -shards 4`: four sharded handles in one C process, exchanged with
    nb_exchange_positions / nb_exchange_accelerations (the multi-GPU-without-RCCL host);
    same trajectory as one handle, in both sharding protocols."""
    exe = _host_program("nbody_main")
    dump = tmp_path / "sh.nbd"
    extra = ["-late-us", late_us] if late_us else []   # hold local items back for the side stream (default from 8 ranks on)
    if protocol == "allreduce":
        extra = ["-allreduce"]                            # replicated integration, in-process all-reduce
    r = subprocess.run([str(exe), "-n", str(n), "-s", "6", "-shards", "4", "-eps", "0.05", "-dump", str(dump)] + extra,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert ("late=0" not in r.stdout) == bool(late_us) or protocol != "symmetric"
    assert "shards=4" in r.stdout and "frame=6" in r.stdout and f"protocol={protocol}" in r.stdout
    back, frame, _ = nb.read_bodies(dump)
    with nb.Simulation(nb.plummer_2d(n, 42), eps=0.05) as sim:
        sim.advance(6, 1e-3)
        want = sim.sync()
    rel = np.max(np.linalg.norm(back["pos"].astype(np.float64) - want["pos"], axis=1) / np.linalg.norm(want["pos"].astype(np.float64), axis=1))
    assert frame == 6 and rel < 2e-6


def test_cxx_default_constructed_simulation_is_the_reference_start():
    """`std::make_shared<Simulation>()` as in main.cpp:657: 25 000-body disc, eps = 1, dt = SIMULATION_DT = 0.01."""
    exe = _host_program("sim_thread_example")
    r = subprocess.run([str(exe), "reference", "3"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"frame=(\d+) bodies=(\d+) body1=\(([-0-9.e+]+), ([-0-9.e+]+)\) last=\(([-0-9.e+]+), ([-0-9.e+]+)\)", r.stdout)
    assert m and int(m.group(1)) == 3 and int(m.group(2)) == 25000, r.stdout
    with nb.Simulation(nb.default_ics(), eps=1.0, extras=3) as sim:
        sim.advance(3, 0.01)
        b = sim.sync()
    got = np.array([float(m.group(k)) for k in (3, 4, 5, 6)])
    want = np.concatenate([b["pos"][1], b["pos"][-1]]).astype(np.float64)
    assert np.allclose(got, want, rtol=1e-6, atol=0)


def test_c_driver_restart_applies_the_dump_header(tmp_path):
    """`nbody_main -load`: the run's parameters come from the dump header (eps, dt, rsqrt mode, sum order), so a
    restart continues the SAME run: 8 steps in one go == 3 steps, dump, load, 5 steps — bit for bit in the
    reference-arithmetic mode.  A forged body count in the header is refused before anything is allocated from it."""
    exe = _host_program("nbody_main")
    a, b, c = tmp_path / "a.nbd", tmp_path / "b.nbd", tmp_path / "c.nbd"
    common = ["-n", "3000", "-eps", "0.07", "-dt", "0.002", "-quake", "-sequential"]

    def run(*args):
        r = subprocess.run([str(exe), *args], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        return r.stdout

    run(*common, "-s", "6", "-dump", str(a))                     # 2 warm-up + 6 = frame 8
    run(*common, "-s", "1", "-dump", str(b))                     # frame 3
    out = run("-load", str(b), "-s", "3", "-dump", str(c))       # header supplies eps, dt, quake, sequential; 2 + 3 more steps
    assert "eps 0.07, dt 0.002, fp32, quake, sequential" in out and "rsqrt=quake sum=sequential" in out
    ba, fa, pa = nb.read_bodies(a)
    bc, fc, pc = nb.read_bodies(c)
    assert fa == fc == 8                                         # nb_params.first_frame: the counter continues across the restart
    assert pc.rsqrt_mode == L.NB_RSQRT_QUAKE and pc.sum_order == L.NB_SUM_SEQUENTIAL and abs(pc.dt - 0.002) < 1e-9
    for f in ("pos", "vel"):
        assert np.array_equal(ba[f].view(np.uint32), bc[f].view(np.uint32)), f
    raw = bytearray(b.read_bytes())
    raw[16:24] = (1 << 58).to_bytes(8, "little")
    bad = tmp_path / "bad.nbd"
    bad.write_bytes(bytes(raw))
    r = subprocess.run([str(exe), "-load", str(bad), "-s", "1"], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "out of range" in r.stderr


def test_pipelined_snapshot_is_the_state_at_begin_and_overlaps_later_steps():
    """nb_snapshot_begin / nb_snapshot_wait: the copy started after step k is frame k even though steps k+1, k+2 were
    enqueued behind it; pageable and page-locked destinations; one snapshot in flight; nb_sync drains it."""
    lib = nb.load()
    ic = nb.plummer_2d(40000, 3)
    with nb.Simulation(ic, eps=0.05) as sim:
        sim.advance(2, 1e-3)
        want = sim.sync().copy()                      # frame 2
        for pinned in (False, True):
            pb = nb.PinnedBodies(40000) if pinned else None
            out = pb.array if pinned else nb.bodies_array(40000)
            try:
                sim.upload(want)                       # back to frame 2's state (acc is carried in the records)
                sim.snapshot_begin(out)
                assert lib.nb_snapshot_begin(sim._h, out.ctypes.data) == L.NB_ESTATE        # one in flight
                sim.advance(2, 1e-3)                   # runs while the copy is in flight
                sim.snapshot_wait()
                for f in ("pos", "vel", "mass", "radius"):
                    assert np.array_equal(out[f].view(np.uint32), want[f].view(np.uint32)), (pinned, f)
                later = sim.sync()
                assert not np.array_equal(later["pos"], want["pos"])
                sim.snapshot_wait()                    # nothing pending: no-op
            finally:
                out = None
                if pb is not None:
                    pb.close()


def test_cxx_adaptor_overlapped_step_delivers_the_same_frames_one_call_late():
    """`Simulation::step_overlapped()` (double-buffered, pipelined D2H) against the reference's blocking pattern
    `step(); SHARED_BODIES = simulation->bodies`: the same frame, bit for bit, after one more call."""
    exe = _host_program("sim_thread_example")
    r = subprocess.run([str(exe), "overlap", "65536", "12"], capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"blocking=([0-9.]+) ms/frame overlapped=([0-9.]+) ms/frame differing_bodies=(\d+)", r.stdout)
    assert m and int(m.group(3)) == 0, r.stdout


def test_handles_and_communicators_release_their_device_memory():
    """Create / use / destroy a few dozen handles (plain, mass-scaled, single-rank sharded + nb_comm, snapshot, momentum,
    energy): the free device memory comes back (a leak of a slab, a sigma array, partial-sum buffers, streams or events
    would show as a steady loss)."""
    from nbodysim_amd.comm import Comm
    # free device memory through the HIP runtime the LIBRARY uses (already loaded: the same copy resolves here).  Not through
    # torch: PyTorch bundles a second ROCm stack, and when it is imported after this library's first HIP call its runtime
    # finds no device ("No HIP GPUs are available") — the order in which this file's tests run alone.
    hip = C.CDLL("libamdhip64.so")

    def free_bytes():
        assert hip.hipDeviceSynchronize() == 0
        free, total = C.c_size_t(), C.c_size_t()
        assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
        return free.value
    n = 65536
    ic = nb.plummer_2d(n, 1)
    gen = ic.copy()
    gen["mass"] = (np.random.default_rng(2).uniform(0.5, 1.5, n) / n).astype(np.float32)

    def cycle():
        with nb.Simulation(ic, eps=0.05) as s:
            s.advance(2, 1e-3); s.energy(); s.momentum(); s.sync()
            out = nb.bodies_array(n)
            s.snapshot_begin(out); s.advance(1, 1e-3); s.snapshot_wait()
        with nb.Simulation(gen, eps=0.05, mass_scaling=True) as s:
            s.advance(2, 1e-3); s.sync()
        with nb.Simulation(ic, eps=0.05, shard_rank=0, shard_world=1, shard_single=True) as s:
            with Comm.all([s]) as c:
                c.step(3, 1e-3); c.wait()
        with nb.PinnedBodies(n) as pb, nb.Simulation(ic, eps=0.05, precision="fp64") as s:
            s.advance(1, 1e-3)
            L.check("nb_sync", nb.load().nb_sync(s._h, pb.array.ctypes.data))

    cycle()                                               # first use: runtime pools, RCCL, code objects
    free0 = free_bytes()
    for _ in range(8):
        cycle()
    free1 = free_bytes()
    assert free0 - free1 < 64 << 20, f"device memory lost over 8 cycles: {(free0 - free1) / 2**20:.1f} MiB"


def test_caller_owned_buffers_filled_on_another_stream_right_before_nb_create():
    """VERDICT r5 next-round 3 — the round-5 stream race, pinned at the C boundary.  A plain host hands nb_create its own position
    replicas (nb_params.pos_buffers) and zero-fills them on ITS stream just before: the handle's stream is non-blocking, so nothing
    orders the library's upload against that fill, and a fill still queued could land AFTER the upload and wipe the positions.
    nb_create / nb_upload therefore wait for all previously enqueued device work when a buffer or the stream is the caller's
    (include/nbody.h, nb_params.pos_buffers).  Here the fill sits behind ~30 ms of queued memsets on a second non-blocking stream,
    so without the fence the upload would finish long before it.  Run once — the ordering is by construction, not by luck."""
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipFree.argtypes = [C.c_void_p]
    hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
    hip.hipStreamDestroy.argtypes = [C.c_void_p]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    hip.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    n = 65536
    ic = nb.plummer_2d(n, 9)
    nb.load()
    stream, scratch = C.c_void_p(), C.c_void_p()
    bufs = [C.c_void_p(), C.c_void_p()]
    big = 2 << 30
    assert hip.hipStreamCreateWithFlags(C.byref(stream), 1) == 0                # hipStreamNonBlocking
    assert hip.hipMalloc(C.byref(scratch), big) == 0
    for b in bufs:
        assert hip.hipMalloc(C.byref(b), n * 8) == 0

    def queue_fill():
        for _ in range(16):                                                     # tens of milliseconds of work ahead of the fill
            assert hip.hipMemsetAsync(scratch, 0, big, stream) == 0
        for b in bufs:
            assert hip.hipMemsetAsync(b, 0, n * 8, stream) == 0                 # the host's "initialise my buffers", still queued ...

    def device_rows(b):
        out = np.empty((n, 2), np.float32)
        assert hip.hipMemcpy(out.ctypes.data, b, n * 8, 2) == 0                 # hipMemcpyDeviceToHost (blocking: after everything)
        return out
    try:
        queue_fill()
        with nb.Simulation(ic, eps=0.05, pos_buffers=(bufs[0].value, bufs[1].value)) as sim:        # ... when nb_create uploads
            assert sim.pos_buffer(0) in (bufs[0].value, bufs[1].value)
            assert hip.hipStreamSynchronize(stream) == 0
            for b in bufs:
                assert np.array_equal(bits(device_rows(b)), bits(ic["pos"])), "the caller's queued fill landed after the upload"
            got = sim.sync()
            assert np.array_equal(bits(got["pos"]), bits(ic["pos"])) and np.array_equal(bits(got["vel"]), bits(ic["vel"]))
            # the same hazard at nb_upload
            moved = ic.copy()
            moved["pos"] += np.float32(0.25)
            queue_fill()
            sim.upload(moved)
            assert hip.hipStreamSynchronize(stream) == 0
            for b in bufs:
                assert np.array_equal(bits(device_rows(b)), bits(moved["pos"]))
            # and the handle computes with what was uploaded: against a library-owned handle of the same bodies, bit for bit
            sim.advance(2, 1e-3)
            mine = sim.sync().copy()
        with nb.Simulation(moved, eps=0.05) as ref:
            ref.advance(2, 1e-3)
            want = ref.sync()
        for f in ("pos", "vel", "acc"):
            assert np.array_equal(bits(mine[f]), bits(want[f])), f
    finally:
        hip.hipStreamSynchronize(stream)
        for b in bufs + [scratch]:
            hip.hipFree(b)
        hip.hipStreamDestroy(stream)
