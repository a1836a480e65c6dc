"""GPU: the 3-D build extension.  z lives in the padding float that follows y in the
reference's alignas(16) Vec2, sizeof(Body) stays 64.  The reference has no 3-D arithmetic, so the
oracle here is the fp64 restatement with a z term ("parity unpinned"); bars: <= 1e-5 relative on
positions / velocities, planar data reproduces the 2-D path, symmetric == one-sided."""
import os

import numpy as np
import pytest

from conftest import max_rel

import nbodysim_amd as nb

pytestmark = pytest.mark.gpu


def f32(x):
    return float(np.float32(x))


@pytest.mark.parametrize("n,steps", [(1000, 20), (5000, 5), (20000, 3), (70001, 1)])
@pytest.mark.parametrize("masses", ["uniform", "individual"])
def test_3d_matches_fp64_restatement(nbo, n, steps, masses):
    ic = nb.plummer_3d(n, 11).view(nb.BODY3_DTYPE)
    if masses == "individual":
        ic["mass"] = (np.random.default_rng(n).uniform(0.5, 1.5, n) / n).astype(np.float32)
    with nb.Simulation(ic, eps=0.03, dims=3) as sim:
        assert "3-D" in sim.describe() and f"symmetric={int(n >= 16384)}" in sim.describe()
        k0, u0 = sim.energy()
        sim.advance(steps, 1e-3)
        k1, u1 = sim.energy()
        got = sim.sync()
        assert sim.frame == steps
    st = nbo.state3_from_bodies(ic)
    e0 = sum(nbo.energy3(st, f32(0.03)))
    nbo.step3_f64(st, f32(0.03), f32(1e-3), steps)
    assert abs(k0 + u0 - e0) < 1e-12 * abs(e0)
    assert max_rel(got["pos"], np.stack([st["x"], st["y"], st["z"]], 1)) < 1e-5
    assert max_rel(got["vel"], np.stack([st["vx"], st["vy"], st["vz"]], 1)) < 1e-5
    a64 = np.stack([st["ax"], st["ay"], st["az"]], 1)
    assert np.max(np.abs(got["acc"] - a64)) < 1e-4 * np.max(np.abs(a64))
    assert abs((k1 + u1) - sum(nbo.energy3(st, f32(0.03)))) < 1e-5 * abs(e0)
    assert np.array_equal(got["mass"], ic["mass"])


def test_3d_with_planar_data_reproduces_the_2d_path():
    ic2 = nb.plummer_2d(20000, 4)
    with nb.Simulation(ic2, eps=0.05) as s2:
        s2.advance(4, 1e-3)
        b2 = s2.sync()
    with nb.Simulation(ic2, eps=0.05, dims=3) as s3:     # z = vz = 0 everywhere
        s3.advance(4, 1e-3)
        b3 = s3.sync()
    assert not b3["pos"][:, 2].any() and not b3["vel"][:, 2].any()
    assert max_rel(b3["pos"][:, :2], b2["pos"]) < 2e-6
    assert max_rel(b3["vel"][:, :2], b2["vel"]) < 2e-5


def test_3d_symmetric_equals_one_sided_and_momentum_is_conserved():
    ic = nb.plummer_3d(30000, 2).view(nb.BODY3_DTYPE)
    res = {}
    for tag, symm in (("sym", True), ("one_sided", False)):
        with nb.Simulation(ic, eps=0.02, dims=3, symmetry=symm) as sim:
            assert f"symmetric={int(symm)}" in sim.describe()
            res[tag] = sim.accelerations().astype(np.float64)
    scale = np.max(np.abs(res["one_sided"]))
    assert np.max(np.abs(res["sym"] - res["one_sided"])) < 2e-5 * scale
    m = ic["mass"].astype(np.float64)[:, None]
    assert np.abs((m * res["sym"]).sum(0)).max() < 1e-6 * np.abs(m * res["sym"]).sum(0).max()


def test_3d_dump_keeps_z_and_2d_dump_zeroes_padding(tmp_path):
    ic = nb.plummer_3d(3000, 9).view(nb.BODY3_DTYPE)
    with nb.Simulation(ic, eps=0.05, dims=3) as sim:
        sim.advance(2, 1e-3)
        a = sim.sync().copy()
        sim.dump(tmp_path / "s3.nbd")
    back, frame, p = nb.read_bodies(tmp_path / "s3.nbd")
    assert frame == 2 and p.dims == 3
    b3 = back.view(nb.BODY3_DTYPE)
    assert np.array_equal(b3["pos"], a["pos"]) and np.array_equal(b3["vel"], a["vel"]) and b3["pos"][:, 2].any()


def test_3d_rejects_unsupported_combinations():
    ic = nb.plummer_3d(1000, 1)
    for kw in (dict(precision="fp64"), dict(order="sequential"), dict(extras=1), dict(i_begin=0, i_count=500)):
        with pytest.raises(nb.NBodyError):
            nb.Simulation(ic, dims=3, **kw)


def test_3d_kdk_integrator_matches_numpy_leapfrog(nbo):
    ic = nb.plummer_3d(3000, 6).view(nb.BODY3_DTYPE)
    eps, dt, steps = f32(0.05), f32(2e-3), 4
    with nb.Simulation(ic, eps=eps, dims=3, integrator="kdk") as sim:
        sim.advance(steps, dt)
        got = sim.sync()
    st = nbo.state3_from_bodies(ic)
    ax, ay, az = (a.copy() for a in nbo.accel3_f64(st, eps))
    for _ in range(steps):
        st["vx"] += 0.5 * dt * ax; st["vy"] += 0.5 * dt * ay; st["vz"] += 0.5 * dt * az
        st["x"] += dt * st["vx"]; st["y"] += dt * st["vy"]; st["z"] += dt * st["vz"]
        ax, ay, az = (a.copy() for a in nbo.accel3_f64(st, eps))
        st["vx"] += 0.5 * dt * ax; st["vy"] += 0.5 * dt * ay; st["vz"] += 0.5 * dt * az
    assert max_rel(got["pos"], np.stack([st["x"], st["y"], st["z"]], 1)) < 1e-5
    assert max_rel(got["vel"], np.stack([st["vx"], st["vy"], st["vz"]], 1)) < 1e-5
