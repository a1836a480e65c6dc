"""GPU: dynamic work items of the symmetric launches (sym_item_index, nbodysim_amd/csrc/nb_kernels.hip.h) against item = workgroup
index (NB_FLAG_STATIC_ITEMS).

Past the first resident wave a workgroup draws the next item of the plan's list when it STARTS, so the XCDs of a part — which are
not equally fast (profiles/history/r04_xcd_speed.log) — end together.  Which workgroup runs which item cannot matter: every item writes
its own slab rows and the gather adds them in a fixed order.  So the bodies must be BIT-IDENTICAL to the static assignment, for
every kernel family that takes tickets (fp32 / fp64, 2-D / 3-D, both tile sizes, chunk pairs) and for the three launch kinds of a
sharded rank (local, cross, late items: one counter each).  Replaces the fan-out of `attract()` (Simulation.hpp:180-213)."""
import numpy as np
import pytest

import nbodysim_amd as nb
from nbodysim_amd import _lib as L

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _same(a, b):
    return all(np.array_equal(_bits(a[f]), _bits(b[f])) for f in ("pos", "vel", "acc"))


@pytest.mark.parametrize("n,dims,kw", [
    (16384, 2, dict()),                                              # wave-split tiles: 1 741 items on 1 024 slots
    (25000, 2, dict(uniform_mass=False, rsqrt="quake")),
    (65536, 2, dict()),                                              # classic tiles, chunk pairs: 3 483 items on 768 slots
    (65536, 2, dict(sym_chunks_per_item=2)),                         # many more items than slots: nearly all of them drawn
    (131072, 2, dict(uniform_mass=False)),
    (65536, 2, dict(precision="fp64")),
    (49152, 3, dict()),
    (32768, 3, dict(precision="fp64", uniform_mass=False)),
    (8192, 2, dict()),                                               # 272 items on 1 024 slots: one wave, nothing is drawn
    (4096, 2, dict()),                                               # one-sided kernel: no items, the flag is accepted and changes nothing
])
def test_dynamic_items_leave_the_same_bits_as_static_ones(n, dims, kw):
    ic = nb.plummer_2d(n, 5) if dims == 2 else nb.plummer_3d(n, 5)
    out = []
    for static in (True, False):
        with nb.Simulation(ic, eps=0.02, dims=dims, static_items=static, **kw) as s:
            for k in (1, 3, 21):                                     # several calls: the counters are monotonic over the handle's life
                s.advance(k, 1e-3)
            out.append((s.sync().copy(), s.energy(), s.frame))
    (a, ea, fa), (b, eb, fb) = out
    assert fa == fb == 25 and _same(a, b) and ea == eb


@pytest.mark.parametrize("allreduce", [False, True])
def test_the_launch_kinds_of_a_sharded_rank_draw_from_their_own_counters(allreduce):
    """One rank running the sharded protocols (NB_FLAG_SHARD_SINGLE): local items (side stream), cross items, late items — three
    launches per step, two of them side by side — or the replicated protocol's single launch."""
    n = 65536
    ic = nb.plummer_2d(n, 9)
    lib = nb.load()
    out = []
    for static in (True, False):
        with nb.Simulation(ic, eps=0.02, shard_rank=0, shard_world=1, shard_single=True, shard_allreduce=allreduce, sym_late_us=40.0,
                           sym_aux_stream=1, static_items=static) as s:
            arr = (L.C.c_void_p * 1)(s._h)
            for _ in range(12):
                s.step_begin(1e-3)
                if not allreduce:
                    s.step_mid()
                    L.check("nb_exchange_accelerations", lib.nb_exchange_accelerations(arr, 1))
                else:
                    L.check("nb_exchange_allreduce", lib.nb_exchange_allreduce(arr, 1))
                s.step_finish()
                if not allreduce:
                    L.check("nb_exchange_positions", lib.nb_exchange_positions(arr, 1))
            s.wait()
            out.append(s.sync().copy())
    assert _same(out[0], out[1])


def test_the_ticket_counters_wrap_around_2_to_the_32():
    """The counters are monotonic modulo 2^32 (no reset between launches): seeded 1 000 draws short of the wrap, a handle whose
    launches draw ~2 700 items each steps across it; same bits as static items."""
    n = 65536
    ic = nb.plummer_2d(n, 11)
    import hooks
    lib = hooks.lib()                                        # the -DNB_TEST_HOOKS build: the product has no nb_debug_ticket_seed
    with nb.Simulation(ic, eps=0.02, static_items=True) as s:      # the product library, static items
        s.advance(6, 1e-3)
        want = s.sync().copy()
    with nb.Simulation(ic, eps=0.02, library=lib) as s:
        assert s.sym_info()["items"] > 2000
        hooks.check("nb_debug_ticket_seed", lib.nb_debug_ticket_seed(s._h, 2 ** 32 - 1000))
        s.advance(6, 1e-3)
        got = s.sync().copy()
    assert _same(got, want)

