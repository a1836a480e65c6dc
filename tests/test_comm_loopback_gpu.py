"""GPU: the library's C-level step loop (nb_comm_create_all + nb_comm_step, nbodysim_amd/csrc/nb_comm.cpp) EXECUTED with
2, 3 and 4 members on the one device a GPU box has.

RCCL takes one rank per device, so the real transport can only ever give this box a communicator of one rank
(tests/test_comm_gpu.py).  Here the executor runs over a test-only loopback transport (tests/loopback_rccl.hip: the nccl*
entry points over device buffers of one process, sums in rank order) named with nb_debug_comm_transport: the in-place
all-gather offset (send = replica + owned_begin), the reduce-scatter's acc_buffer(0) -> acc_buffer(1) counts, the
all-reduce, the ncclGroup bracketing and the event hand-overs meet real buffers, and every rank's trajectory must be
BIT-IDENTICAL to the same handles driven through the library's in-process exchange (nb_exchange_*), all three protocols,
fp32 / fp64 / 3-D, equal and ragged blocks.  The transport is loaded once per process, hence one child process for all cases."""
import json
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

STUB = ROOT / "tests" / "libnb_loopback_rccl.so"


@pytest.fixture(scope="module")
def results():
    if not STUB.exists():       # test infrastructure: normally built by __graft_entry__.build()
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", str(STUB),
                        str(ROOT / "tests" / "loopback_rccl.hip")], check=True, capture_output=True)
    r = subprocess.run([sys.executable, str(ROOT / "tests" / "loopback_worker.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = {}
    for line in r.stdout.splitlines():
        if line.startswith("{"):
            d = json.loads(line)
            out[d["case"]] = d
    return out


def test_the_loopback_transport_was_the_one_loaded(results):
    assert results["transport"]["version"] == 20000          # the stand-in's version, not RCCL's


@pytest.mark.parametrize("name,world,protocol", [
    ("symmetric fp32 world 2", 2, 2), ("symmetric fp32 world 4", 4, 2), ("symmetric+late fp32 world 4", 4, 2),
    ("allreduce fp32 world 2", 2, 3), ("allreduce fp32 world 4", 4, 3),
    ("allgather fp32 world 2", 2, 1), ("allgather fp32 world 4", 4, 1),
    ("allgather ragged fp32 world 3", 3, 1), ("allgather ragged sequential quake world 3", 3, 1),
    ("symmetric fp64 world 2", 2, 2), ("allreduce fp64 world 4", 4, 3),
    ("symmetric 3-D fp32 world 2", 2, 2), ("allgather 3-D fp32 world 4", 4, 1),
    ("symmetric fp32 world 8 headline", 8, 2), ("allreduce fp32 world 8 headline", 8, 3),
])
def test_c_loop_with_several_members_equals_the_in_process_exchange_bit_for_bit(results, name, world, protocol):
    d = results[name]
    assert d["world"] == world and d["protocol"] == protocol and d["comm"]["world"] == world and d["comm"]["local_handles"] == world
    assert d["frames"] == [4 if "headline" in name else 6] * world
    assert d["bit_identical"], d
    if name == "symmetric fp32 world 8 headline":
        assert all(k > 0 for k in d["late_items"]), d["late_items"]      # from 8 ranks the held-back local items are on by default
    n = d["n"]
    if protocol == 3:
        assert d["owned"] == [n] * world                     # replicated: every rank integrates everything
    else:
        stride = -(-n // world)
        assert d["owned"] == [min(stride, n - r * stride) for r in range(world)]      # ragged last block included


def test_c_loop_phase_marks_cover_every_step_of_every_member(results):
    ph = results["symmetric fp32 world 2"]["phases"]
    assert len(ph) == 2
    for p in ph:
        assert p["steps"] == 6
        assert p["local"] > 0 and p["cross"] > 0 and p["finish"] > 0 and p["reduce"] >= 0 and p["ag_wait"] >= 0
        assert sum(p[k] for k in ("local", "ag_wait", "cross", "reduce", "finish")) < 50.0      # ms per step: sane


def test_a_collective_failing_mid_step_marks_the_communicator_failed_and_destroy_does_not_wait(results):
    f = results["failure"]
    assert f["rc_step"] == -3 and "ReduceScatter" in f["error"] and "injected" in f["error"]      # NB_EHIP with RCCL's text
    assert f["rc_next_step"] == -7 and f["rc_wait"] == -7                                            # NB_ESTATE: no further steps, no waiting
    assert f["destroy_seconds"] < 5.0                                                                # ncclCommAbort path: no synchronisation with absent peers
