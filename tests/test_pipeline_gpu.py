"""GPU: the persistent step pipeline (sym_pipeline_f32, nbodysim_amd/csrc/nb_kernels.hip.h; EXPERIMENTAL, opt-in
NB_FLAG_PIPELINE: it is slower than two launches per step at every size measured) against the two-launch path.

`nb_step(dt, nsteps)` of a whole-system fp32 2-D handle is ONE persistent launch: resident workgroups draw (step, item)
tickets, per-tile counters stand where the launch boundaries were, the last arriver of a tile sums its slabs and kicks
and drifts it.  The sums keep sym_gather's order and association, so positions, velocities and accelerations must be
BIT-IDENTICAL to the same handle stepped with two launches per step (NB_FLAG_NO_PIPELINE) — which is also the sharpest
test of the hand-offs: a stale or early read of a slab row or a position shows up as a differing bit.  Replaces what
`Simulation::step()` does per call (Simulation.hpp:67-75) for a caller that asks for many steps at once."""
import numpy as np
import pytest

import nbodysim_amd as nb
from nbodysim_amd import _lib as L

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _same(a, b):
    return all(np.array_equal(_bits(a[f]), _bits(b[f])) for f in ("pos", "vel", "acc"))


def _run(ic, batches, dt, pipeline, **kw):
    out = []
    with nb.Simulation(ic, pipeline=pipeline, **kw) as s:
        assert f"pipeline={int(pipeline)}" in s.describe(), s.describe()
        for k in batches:
            s.advance(k, dt)
            s.wait()
            out.append(s.sync().copy())
        frame, e = s.frame, s.energy()
    return out, frame, e


@pytest.mark.parametrize("n,kw", [
    (16384, dict(eps=0.05)),                                         # wave-split tiles (512), equal masses
    (16384, dict(eps=0.05, uniform_mass=False)),
    (20001, dict(eps=0.01, rsqrt="quake")),                          # ragged last tile, the reference's rsqrt
    (32768, dict(eps=0.05, sym_chunks_per_item=4)),                  # many small items: more hand-offs per step
    (65536, dict(eps=0.01)),                                         # classic tiles (2048), chunk pairs
    (65536, dict(eps=0.01, sym_tile=512)),                           # wave-split tiles + chunk pairs
    (70001, dict(eps=0.01, uniform_mass=False)),                     # ragged, individual masses, chunk pairs
    (131072, dict(eps=0.01, sym_chunks_per_item=96)),                # forced coarse items
    (49152, dict(eps=0.05, mass_scaling=True, uniform_mass=False)),  # the opt-in mass-scaled body
])
def test_pipeline_is_bit_identical_to_two_launches_per_step(n, kw):
    ic = nb.plummer_2d(n, 42)
    if kw.get("mass_scaling"):
        ic["mass"] *= np.random.default_rng(1).uniform(0.5, 2.0, n).astype(np.float32)
    batches, dt = (1, 2, 9, 20), 1e-3                                 # several launches: the counters carry over
    a, fa, ea = _run(ic, batches, dt, True, **kw)
    b, fb, eb = _run(ic, batches, dt, False, **kw)
    assert fa == fb == sum(batches)
    for k, (x, y) in enumerate(zip(a, b)):
        assert _same(x, y), (n, kw, f"after batch {k}")
    assert ea == eb


def test_the_reference_default_workload_through_the_pipeline():
    """Simulation()'s own start (25 000 bodies, 1e9 central mass, eps = 1, clamp + soft boundary in the fused kick)."""
    ic = nb.default_ics(25000)
    kw = dict(eps=1.0, extras=3)
    a, _, _ = _run(ic, (3, 30), 0.01, True, **kw)
    b, _, _ = _run(ic, (3, 30), 0.01, False, **kw)
    assert _same(a[0], b[0]) and _same(a[1], b[1])


def test_pipeline_interleaved_with_the_other_entry_points():
    """Two-launch force evaluations (nb_accelerations), uploads and single steps between pipeline launches: the pipeline's
    counters count ITS steps only, the replicas flip with every step whoever made it."""
    n = 32768
    ic = nb.plummer_2d(n, 7)

    def drive(pipeline):
        with nb.Simulation(ic, eps=0.05, pipeline=pipeline) as s:
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
    acc_p, end_p, fp = drive(True)
    acc_t, end_t, ft = drive(False)
    assert fp == ft == 13
    assert np.array_equal(_bits(acc_p), _bits(acc_t)) and _same(end_p, end_t)


def test_pipeline_against_fp64_direct_at_the_reference_size(nbo):
    """...and it is the physics: 5 steps at N = 25 000 (Plummer, individual masses) within 1e-5 of the fp64 direct sum."""
    n, eps, dt, steps = 25000, 0.05, 1e-3, 5
    ic = nb.plummer_2d(n, 3)
    ic["mass"] *= np.random.default_rng(2).uniform(0.5, 2.0, n).astype(np.float32)
    with nb.Simulation(ic, eps=eps, pipeline=True) as s:
        assert "pipeline=1" in s.describe() and "tile=512" in s.describe()
        s.advance(steps, dt)
        got = s.sync().copy()
    st = nbo.step_f64(nbo.state_from_bodies(ic, np.float64), eps, dt, steps)
    pos = np.stack([st["x"], st["y"]], 1)
    vel = np.stack([st["vx"], st["vy"]], 1)
    assert np.max(np.linalg.norm(got["pos"] - pos, axis=1) / np.linalg.norm(pos, axis=1)) < 1e-5
    assert np.max(np.linalg.norm(got["vel"] - vel, axis=1)) / np.max(np.linalg.norm(vel, axis=1)) < 1e-5


def test_handles_that_cannot_pipeline_say_so():
    ic = nb.plummer_2d(32768, 1)
    for kw in (dict(precision="fp64", pipeline=True), dict(integrator="kdk", pipeline=True), dict(symmetry=False, pipeline=True), dict()):
        with nb.Simulation(ic, eps=0.05, **kw) as s:
            assert "pipeline=0" in s.describe()
            s.advance(2, 1e-3)
            s.wait()
    with nb.Simulation(nb.plummer_3d(32768, 1), eps=0.05, dims=3, pipeline=True) as s:
        assert "pipeline=0" in s.describe()
