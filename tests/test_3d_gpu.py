"""GPU: the 3-D build extension.  z lives in the padding float that follows y in the
reference's alignas(16) Vec2, sizeof(Body) stays 64.  The reference has no 3-D arithmetic, so the
oracle here is the fp64 restatement with a z term ("parity unpinned"); bars: <= 1e-5 relative on
positions / velocities (fp32), total energy to 1e-10 (fp64), planar data reproduces the 2-D path,
symmetric == one-sided, sharded (both protocols, fp32 and fp64) == unsharded."""
import ctypes

import numpy as np
import pytest

from conftest import max_rel

import nbodysim_amd as nb
from nbodysim_amd import _lib as L

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
    # the chunk-pair sweep in 3-D (sym3_chunks2), equal and individual masses, even and odd (3) chunks per item, both rsqrt modes
    gen = ic.copy()
    gen["mass"] = (np.random.default_rng(4).uniform(0.5, 1.5, 30000) / 30000).astype(np.float32)
    for bodies in (ic, gen):
        for rsqrt in ("exact", "quake"):
            with nb.Simulation(bodies, eps=0.02, dims=3, rsqrt=rsqrt, sym_chunk_pairs=-1) as sim:
                single = sim.accelerations().astype(np.float64)
            for chunks in (0, 3):
                with nb.Simulation(bodies, eps=0.02, dims=3, rsqrt=rsqrt, sym_chunk_pairs=1, sym_chunks_per_item=chunks) as sim:
                    assert "chunk_pairs=1" in sim.describe()
                    pair = sim.accelerations().astype(np.float64)
                assert np.max(np.abs(pair - single)) < 2e-6 * np.max(np.abs(single)), (rsqrt, chunks)
    m = ic["mass"].astype(np.float64)[:, None]
    assert np.abs((m * res["sym"]).sum(0)).max() < 1e-6 * np.abs(m * res["sym"]).sum(0).max()
    # nb_momentum in 3-D (mass rides in pos.w on the device): fp32 and fp64 handles against the host sum, before and after steps
    for precision in ("fp32", "fp64"):
        with nb.Simulation(ic, eps=0.02, dims=3, precision=precision) as sim:
            (px, py, pz), lz = sim.momentum()
            want = (m * ic["vel"].astype(np.float64)).sum(0)
            lz0 = float((m[:, 0] * (ic["pos"][:, 0].astype(np.float64) * ic["vel"][:, 1] - ic["pos"][:, 1].astype(np.float64) * ic["vel"][:, 0])).sum())
            assert np.allclose([px, py, pz], want, rtol=0, atol=1e-12) and abs(lz - lz0) < 1e-12
            sim.advance(20, 1e-3)
            (qx, qy, qz), lz1 = sim.momentum()
            pscale = np.abs(m * ic["vel"]).sum(0).max()
            assert max(abs(qx - px), abs(qy - py), abs(qz - pz)) < 1e-5 * pscale and abs(lz1 - lz) < 1e-5 * abs(np.abs(m[:, 0] * np.linalg.norm(ic["pos"][:, :2], axis=1)
                                                                                                                   * np.linalg.norm(ic["vel"][:, :2], axis=1)).sum())


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
    """The reference defines its sequential order and iterate()'s extras in the plane only."""
    ic = nb.plummer_3d(1000, 1)
    for kw in (dict(order="sequential"), dict(extras=1), dict(precision="fp64", rsqrt="quake")):
        with pytest.raises(nb.NBodyError) as e:
            nb.Simulation(ic, dims=3, **kw)
        assert e.value.code == L.NB_EINVAL


@pytest.mark.parametrize("n,steps", [(3000, 6), (20000, 3), (40001, 2)])
@pytest.mark.parametrize("masses", ["uniform", "individual"])
def test_3d_fp64_matches_fp64_restatement(nbo, n, steps, masses):
    """force_sym3_f64 / force_tiled3_f64: the same trajectory as the CPU fp64 direct sum with a z term."""
    ic = nb.plummer_3d(n, 21).view(nb.BODY3_DTYPE)
    if masses == "individual":
        ic["mass"] = (np.random.default_rng(n).uniform(0.5, 1.5, n) / n).astype(np.float32)
    st = nbo.state3_from_bodies(ic)
    e0 = sum(nbo.energy3(st, f32(0.03)))
    nbo.step3_f64(st, f32(0.03), f32(1e-3), steps)
    e1 = sum(nbo.energy3(st, f32(0.03)))
    for symm in (True, False):
        with nb.Simulation(ic, eps=0.03, dims=3, precision="fp64", symmetry=symm) as sim:
            assert "fp64 3-D" in sim.describe() and f"symmetric={int(symm and n >= 16384)}" in sim.describe()
            k0, u0 = sim.energy()
            sim.advance(steps, 1e-3)
            k1, u1 = sim.energy()
            got = sim.sync()
        assert abs(k0 + u0 - e0) < 1e-12 * abs(e0)
        assert abs(k1 + u1 - e1) < 1e-10 * abs(e1), (symm, (k1 + u1 - e1) / e1)
        assert max_rel(got["pos"], np.stack([st["x"], st["y"], st["z"]], 1)) < 2e-7        # float output records
        assert max_rel(got["vel"], np.stack([st["vx"], st["vy"], st["vz"]], 1)) < 2e-7
        a64 = np.stack([st["ax"], st["ay"], st["az"]], 1)
        assert np.max(np.abs(got["acc"] - a64)) < 2e-7 * np.max(np.abs(a64))


def _run_sharded3(ic, parts, steps, dt, **kw):
    """P handles of a 3-D system on one GPU, driven from this process through the library's in-process exchanges."""
    lib = nb.load()
    n = ic.shape[0]
    blk = n // parts
    sims = [nb.Simulation(ic, dims=3, i_begin=r * blk, i_count=blk, shard_rank=r, shard_world=parts, **kw) for r in range(parts)]
    try:
        protos = {s.shard_protocol for s in sims}
        assert len(protos) == 1
        sym = protos == {L.NB_SHARD_SYMMETRIC}
        handles = (ctypes.c_void_p * parts)(*[s._h for s in sims])
        for _ in range(steps):
            for s in sims:
                s.step_begin(dt)
            if sym:
                for s in sims:
                    s.step_mid()
                L.check("nb_exchange_accelerations", lib.nb_exchange_accelerations(handles, parts))
            for s in sims:
                s.step_finish()
            L.check("nb_exchange_positions", lib.nb_exchange_positions(handles, parts))
        out = nb.bodies_array(n).view(nb.BODY3_DTYPE)
        k = u = 0.0
        for s in sims:
            out[s.i_begin:s.i_begin + s.i_count] = s.sync()
            kk, uu = s.energy()
            k, u = k + kk, u + uu
    finally:
        for s in sims:
            s.close()
    return out, k + u, sym


@pytest.mark.parametrize("precision", ["fp32", "fp64"])
def test_3d_replicated_allreduce_protocol(precision):
    """NB_SHARD_ALLREDUCE in 3-D: four in-process handles, each integrating all n after the all-reduce."""
    lib = nb.load()
    n, parts, steps = 65536, 4, 3
    ic = nb.plummer_3d(n, 9).view(nb.BODY3_DTYPE)
    with nb.Simulation(ic, dims=3, eps=0.02, precision=precision) as sim:
        sim.advance(steps, 1e-3)
        whole = sim.sync().copy()
    sims = [nb.Simulation(ic, dims=3, eps=0.02, precision=precision, shard_rank=r, shard_world=parts, shard_allreduce=True) for r in range(parts)]
    try:
        assert all(s.shard_protocol == L.NB_SHARD_ALLREDUCE for s in sims)
        handles = (ctypes.c_void_p * parts)(*[s._h for s in sims])
        for _ in range(steps):
            for s in sims:
                s.step_begin(1e-3)
            L.check("nb_exchange_allreduce", lib.nb_exchange_allreduce(handles, parts))
            for s in sims:
                s.step_finish()
        outs = [s.sync().copy() for s in sims]
    finally:
        for s in sims:
            s.close()
    for o in outs[1:]:
        assert np.array_equal(o["pos"].view(np.uint32), outs[0]["pos"].view(np.uint32))
    tol = (2e-6, 2e-5) if precision == "fp32" else (2e-7, 2e-7)
    assert max_rel(outs[0]["pos"], whole["pos"]) < tol[0] and max_rel(outs[0]["vel"], whole["vel"]) < tol[1]
    assert outs[0]["pos"][:, 2].any()


@pytest.mark.parametrize("precision", ["fp32", "fp64"])
@pytest.mark.parametrize("protocol,parts,late_us", [("symmetric", 2, -1.0), ("symmetric", 4, 40.0), ("allgather", 4, 0.0)])
def test_3d_sharded_handles_match_unsharded(precision, protocol, parts, late_us):
    n, steps = 65536, 3
    ic = nb.plummer_3d(n, 9).view(nb.BODY3_DTYPE)
    kw = dict(eps=0.02, precision=precision, symmetry=protocol == "symmetric", sym_late_us=late_us)
    with nb.Simulation(ic, dims=3, eps=0.02, precision=precision) as sim:
        sim.advance(steps, 1e-3)
        whole = sim.sync().copy()
        e_whole = sum(sim.energy())
    out, e, sym = _run_sharded3(ic, parts, steps, 1e-3, **kw)
    assert sym == (protocol == "symmetric")
    tol_p, tol_v = (2e-6, 2e-5) if precision == "fp32" else (2e-7, 2e-7)
    assert max_rel(out["pos"], whole["pos"]) < tol_p and max_rel(out["vel"], whole["vel"]) < tol_v
    assert np.max(np.abs(out["acc"] - whole["acc"])) < (2e-5 if precision == "fp32" else 2e-7) * np.max(np.abs(whole["acc"]))
    assert abs(e - e_whole) < (1e-5 if precision == "fp32" else 1e-10) * abs(e_whole)
    assert out["pos"][:, 2].any()


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


@pytest.mark.parametrize("precision", ["fp32", "fp64"])
def test_3d_positions_fast_path(precision):
    ic = nb.plummer_3d(5000, 2).view(nb.BODY3_DTYPE)
    with nb.Simulation(ic, eps=0.05, dims=3, precision=precision) as sim:
        sim.advance(3, 1e-3)
        full = sim.sync()["pos"].copy()
        xyz = sim.positions()
    assert xyz.shape == (5000, 3) and np.array_equal(xyz, full)
