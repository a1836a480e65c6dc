"""GPU: one launch per step (sym_step_f32, nbodysim_amd/csrc/nb_kernels.hip.h; EXPERIMENTAL, opt-in NB_FLAG_ONE_LAUNCH_STEP,
DESIGN.md 4.8) against the default two launches per step.

The grid of a step is the plan's force items followed by the gather workgroups; a gather workgroup waits until every item that
contributes to its tile has arrived (per-tile counters), then sums the tile's slabs in sym_gather's order and association and
kicks and drifts its 64 particles.  Positions, velocities and accelerations must therefore be BIT-IDENTICAL to the same handle
stepped with force_sym_f32 + sym_gather — which is also the sharpest test of the hand-off: a slab row read before it was
complete, or a stale one, shows up as a differing bit.  Replaces what `Simulation::step()` does per call (Simulation.hpp:67-75)."""
import numpy as np
import pytest

import nbodysim_amd as nb

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _same(a, b):
    return all(np.array_equal(_bits(a[f]), _bits(b[f])) for f in ("pos", "vel", "acc"))


def _run(ic, batches, dt, one_launch, **kw):
    out = []
    with nb.Simulation(ic, one_launch=one_launch, **kw) as s:
        assert f"one_launch={int(one_launch)}" in s.describe(), s.describe()
        for k in batches:
            s.advance(k, dt)
            s.wait()
            out.append(s.sync().copy())
        frame, e = s.frame, s.energy()
    return out, frame, e


@pytest.mark.parametrize("n,kw", [
    (9216, dict(eps=0.05)),                                          # the smallest symmetric plan (18 wave-split tiles)
    (16384, dict(eps=0.05, uniform_mass=False)),
    (20001, dict(eps=0.01, rsqrt="quake")),                          # ragged last tile, the reference's rsqrt
    (32768, dict(eps=0.05, sym_chunks_per_item=4)),                  # many small items: more arrivals per tile
    (65536, dict(eps=0.01)),                                         # classic tiles (2048), chunk pairs
    (65536, dict(eps=0.01, sym_tile=512)),                           # wave-split tiles + chunk pairs
    (70001, dict(eps=0.01, uniform_mass=False)),                     # ragged, individual masses, chunk pairs
    (131072, dict(eps=0.01, sym_chunks_per_item=96)),                # coarse items spanning several tiles each
    (49152, dict(eps=0.05, mass_scaling=True, uniform_mass=False)),  # the opt-in mass-scaled body
    (262144, dict(eps=0.01)),                                        # the benchmark size: 4096 gather workgroups behind 8187 items
])
def test_one_launch_is_bit_identical_to_two_launches_per_step(n, kw):
    ic = nb.plummer_2d(n, 42)
    if kw.get("mass_scaling"):
        ic["mass"] *= np.random.default_rng(1).uniform(0.5, 2.0, n).astype(np.float32)
    batches, dt = ((1, 2, 9, 20) if n < 200000 else (1, 5)), 1e-3      # several calls: the counters carry over
    a, fa, ea = _run(ic, batches, dt, True, **kw)
    b, fb, eb = _run(ic, batches, dt, False, **kw)
    assert fa == fb == sum(batches)
    for k, (x, y) in enumerate(zip(a, b)):
        assert _same(x, y), (n, kw, f"after batch {k}")
    assert ea == eb


def test_the_reference_default_workload_in_one_launch_per_step():
    """Simulation()'s own start (25 000 bodies, 1e9 central mass, eps = 1, clamp + soft boundary in the fused kick)."""
    ic = nb.default_ics(25000)
    kw = dict(eps=1.0, extras=3)
    a, _, _ = _run(ic, (3, 300), 0.01, True, **kw)
    b, _, _ = _run(ic, (3, 300), 0.01, False, **kw)
    assert _same(a[0], b[0]) and _same(a[1], b[1])


def test_one_launch_steps_interleaved_with_the_other_entry_points():
    """Two-launch force evaluations (nb_accelerations), uploads and single steps between fused steps: the arrival counters
    count the fused launches only, the replicas flip with every step whoever made it."""
    n = 32768
    ic = nb.plummer_2d(n, 7)

    def drive(one_launch):
        with nb.Simulation(ic, eps=0.05, one_launch=one_launch) as s:
            s.advance(3, 1e-3)
            acc = s.accelerations()                    # force_sym + gather, no integration
            s.advance(1, 1e-3)
            s.step(1e-3)                               # nb_step(1) + nb_sync
            mid = s.bodies.copy()
            mid["vel"] *= np.float32(0.5)
            s.upload(mid)                              # host-side edit of all bodies
            s.advance(8, 1e-3)
            end = s.sync().copy()
            return acc, end, s.frame
    acc_o, end_o, fo = drive(True)
    acc_t, end_t, ft = drive(False)
    assert fo == ft == 13
    assert np.array_equal(_bits(acc_o), _bits(acc_t)) and _same(end_o, end_t)


def test_handles_that_cannot_step_in_one_launch_say_so():
    ic = nb.plummer_2d(32768, 1)
    for kw in (dict(precision="fp64", one_launch=True), dict(integrator="kdk", one_launch=True), dict(symmetry=False, one_launch=True),
               dict(pipeline=True, one_launch=True), dict()):
        with nb.Simulation(ic, eps=0.05, **kw) as s:
            assert "one_launch=0" in s.describe()
            s.advance(2, 1e-3)
            s.wait()
    with nb.Simulation(nb.plummer_3d(32768, 1), eps=0.05, dims=3, one_launch=True) as s:
        assert "one_launch=0" in s.describe()
