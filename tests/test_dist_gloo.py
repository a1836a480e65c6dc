"""CPU, world_size 2 over gloo: the sharding plan and the position exchange of
nbodysim_amd.dist.  The force engine here is the ORACLE (test infrastructure):
what is under test is the host-side partition / all-gather layout that the GPU
path uses unchanged with the nccl (RCCL) backend."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


from conftest import free_port as _free_port  # noqa: E402  (below the ephemeral range: see there)


def _worker(rank, world, port, n, steps, out_dir):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "oracle"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    import torch
    import torch.distributed as dist

    import nbo
    from nbodysim_amd.dist import ShardPlan, exchange_positions

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        nbo.set_threads(2)
        flat = np.load(ROOT / "tests" / "golden" / "ic_plummer_1024.npy")[:n]
        plan = ShardPlan(n, world, rank)
        st = nbo.state_from_flat(flat)
        eps, dt = 0.05, 1e-3
        lo, hi = plan.i_begin, plan.i_end
        # double-buffered full replicas, as on the GPU
        pos = [torch.from_numpy(np.stack([st["x"], st["y"]], 1).copy()) for _ in range(2)]
        cur, pending = 0, None
        vx, vy = st["vx"][lo:hi].copy(), st["vy"][lo:hi].copy()
        for _ in range(steps):
            if pending is not None:
                pending.wait()
                pending = None
            full = pos[cur].numpy()
            s = {"x": np.ascontiguousarray(full[:, 0]), "y": np.ascontiguousarray(full[:, 1]), "m": st["m"]}
            ax, ay = nbo.accel_f32(s, eps, nbo.RSQRT_QUAKE, lo, hi)      # owned i only
            dt32 = np.float32(dt)
            vx = (vx + ax[lo:hi] * dt32).astype(np.float32)
            vy = (vy + ay[lo:hi] * dt32).astype(np.float32)
            nxt = pos[cur ^ 1]
            nxt[lo:hi, 0] = torch.from_numpy((full[lo:hi, 0] + vx * dt32).astype(np.float32))
            nxt[lo:hi, 1] = torch.from_numpy((full[lo:hi, 1] + vy * dt32).astype(np.float32))
            cur ^= 1
            pending = exchange_positions(pos[cur], plan, async_op=True)
        if pending is not None:
            pending.wait()
        np.save(Path(out_dir) / f"pos_rank{rank}.npy", pos[cur].numpy())
        np.save(Path(out_dir) / f"vel_rank{rank}.npy", np.stack([vx, vy], 1))
    finally:
        dist.destroy_process_group()


def test_shard_plan():
    from nbodysim_amd.dist import ShardPlan
    p = [ShardPlan(1024, 4, r) for r in range(4)]
    assert [q.i_begin for q in p] == [0, 256, 512, 768] and all(q.i_count == 256 for q in p)
    assert p[3].i_end == 1024 and p[1].block(2) == slice(512, 768)
    assert ShardPlan(1000, 3, 2).i_count == 332 and ShardPlan(1000, 3, 0).padded_n == 1002      # ragged: a shorter last block
    with pytest.raises(ValueError):
        ShardPlan(1024, 4, 4)
    assert ShardPlan(7, 1, 0).i_count == 7


@pytest.mark.parametrize("steps", [1, 5])
def test_two_rank_gloo_sharded_steps_match_unsharded_oracle(tmp_path, steps, gold, nbo):
    import torch.multiprocessing as mp
    n, world = 512, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, steps, str(tmp_path)), nprocs=world, join=True)
    st = nbo.state_from_flat(gold["ic_plummer_1024"][:n])
    nbo.step_f32(st, 0.05, 1e-3, steps, nbo.RSQRT_QUAKE)
    want_pos = np.stack([st["x"], st["y"]], 1)
    want_vel = np.stack([st["vx"], st["vy"]], 1)
    for r in range(world):
        got = np.load(tmp_path / f"pos_rank{r}.npy")
        # every rank ends with the same, complete replica, bit-identical to the unsharded run
        assert np.array_equal(got.view(np.uint32), want_pos.view(np.uint32)), r
        v = np.load(tmp_path / f"vel_rank{r}.npy")
        lo, hi = r * n // world, (r + 1) * n // world
        assert np.array_equal(v.view(np.uint32), want_vel[lo:hi].view(np.uint32))


def _sym_worker(rank, world, port, n, steps, out_dir):
    """The symmetric protocol's data flow on CPU: this rank's share of the UNORDERED pairs (the product planner's
    items, read through nb_debug_sym_plan — a host-only entry) evaluated with numpy in fp64, partial accelerations
    of all particles summed across ranks with dist.reduce_accelerations, owned block kicked and drifted, positions
    all-gathered in place with dist.exchange_positions."""
    import ctypes as C
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    import nbodysim_amd as nb
    from nbodysim_amd.dist import ShardPlan, exchange_positions, reduce_accelerations

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nbodysim_amd import _lib as L
        sys.path.insert(0, str(ROOT / "tests"))
        import hooks
        items, info = hooks.sym_plan(n, 256, rank, world)
        slab_r = np.zeros((info["slab_r_bytes"] // 8, 2))          # the travelling slab, laid out as the planner says
        ic = nb.plummer_2d(n, 11)
        m = ic["mass"].astype(np.float64)
        plan = ShardPlan(n, world, rank)
        lo, hi = plan.i_begin, plan.i_end
        eps2, dt = 0.05 ** 2, 1e-3
        pos = [torch.from_numpy(ic["pos"].astype(np.float64).copy()) for _ in range(2)]
        vel = ic["vel"].astype(np.float64)[lo:hi].copy()
        cur, pending = 0, None
        for _ in range(steps):
            if pending is not None:
                pending.wait()
                pending = None
            x = pos[cur].numpy()
            acc_full = np.zeros((n, 2))
            slab_r[:] = np.nan                                          # every element must be written before it is read
            for it in items:
                tile, c0, c, diag, r_base = int(it["tile"]), int(it["c0"]), int(it["cnt"]), int(it["diag"]), int(it["r_base"])
                sb = info["tile_particles"]
                S = slice(tile * sb, min((tile + 1) * sb, n))
                T = slice(c0 * 64, min((c0 + c) * 64, n))
                d = x[None, T, :] - x[S, None, :]                       # r = p_j - p_i (Quadtree.hpp:136)
                inv3 = (d[..., 0] ** 2 + d[..., 1] ** 2 + eps2) ** -1.5
                acc_full[S] += np.einsum("ij,ijk->ik", inv3 * m[None, T], d)
                if not diag:                                            # Newton's third law: the same pairs, other end
                    slab_r[r_base + T.start:r_base + T.stop] = -np.einsum("ij,ijk->jk", inv3 * m[S, None], d)
            for it in items[items["diag"] == 0]:                         # gather of the travelling partials
                T = slice(int(it["c0"]) * 64, min((int(it["c0"]) + int(it["cnt"])) * 64, n))
                acc_full[T] += slab_r[int(it["r_base"]) + T.start:int(it["r_base"]) + T.stop]
            assert np.isfinite(acc_full).all() and np.isfinite(slab_r).all()
            full_t, own_t = torch.from_numpy(acc_full), torch.zeros((hi - lo, 2), dtype=torch.float64)
            reduce_accelerations(full_t, own_t, plan)
            vel += own_t.numpy() * dt
            nxt = pos[cur ^ 1]
            nxt[lo:hi] = torch.from_numpy(x[lo:hi] + vel * dt)
            cur ^= 1
            pending = exchange_positions(pos[cur], plan, async_op=True)
        if pending is not None:
            pending.wait()
        np.save(Path(out_dir) / f"sympos_rank{rank}.npy", pos[cur].numpy())
        np.save(Path(out_dir) / f"symvel_rank{rank}.npy", vel)
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_symmetric_protocol_matches_unsharded_fp64(tmp_path, nbo):
    """world_size 2 over gloo on CPU, the reduce-scatter + all-gather protocol of the benchmark with the product's
    own pair split: identical (to fp64 rounding) to the unsharded fp64 direct sum."""
    import torch.multiprocessing as mp

    import nbodysim_amd as nb
    n, world, steps = 8192, 2, 2
    mp.spawn(_sym_worker, args=(world, _free_port(), n, steps, str(tmp_path)), nprocs=world, join=True)
    ic = nb.plummer_2d(n, 11)
    st = nbo.step_f64(nbo.state_from_bodies(ic, np.float64), 0.05, 1e-3, steps)
    want_pos, want_vel = np.stack([st["x"], st["y"]], 1), np.stack([st["vx"], st["vy"]], 1)
    for r in range(world):
        got = np.load(tmp_path / f"sympos_rank{r}.npy")
        assert np.max(np.abs(got - want_pos)) < 1e-12 * np.max(np.abs(want_pos))        # complete replica on every rank
        v = np.load(tmp_path / f"symvel_rank{r}.npy")
        blk = slice(r * n // world, (r + 1) * n // world)
        assert np.max(np.abs(v - want_vel[blk])) < 1e-11 * np.max(np.abs(want_vel))


def _agree_worker(rank, world, port, out_dir):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    from nbodysim_amd.dist import agree_on_fastest, ranks_agree

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = {}
        # same plan figures on both ranks / one rank built another plan (different chunks per item) / one failed to create
        res["same"] = ranks_agree([1, 2, 43, 123456789012, 4096, 128, 256, 0])
        res["diff"] = ranks_agree([1, 2, 43 + rank, 123456789012, 4096, 128, 256, 0])
        res["failed"] = ranks_agree([rank, 2 * rank, 0, 0, 0, 0, 0, 0])
        # rank 0 saw the symmetric protocol faster, rank 1 (whose reduce-scatter was exposed) much slower:
        # the job runs at the slowest rank's pace, so both must pick all-gather
        res["tune"] = agree_on_fastest({"symmetric": [1.00e-3, 1.60e-3][rank], "allgather": [1.35e-3, 1.30e-3][rank]})
        # near tie (within 1 %): the preferred (symmetric) protocol wins on both ranks
        res["tie"] = agree_on_fastest({"symmetric": [1.000e-3, 1.004e-3][rank], "allgather": [1.001e-3, 0.999e-3][rank]})
        # symmetric unavailable (system not eligible)
        res["only_ag"] = agree_on_fastest({"symmetric": float("inf"), "allgather": 2e-3})
        import pickle
        (Path(out_dir) / f"agree_{rank}.pkl").write_bytes(pickle.dumps(res))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_startup_agreement_logic(tmp_path):
    """The start-up decisions of a sharded run (nbodysim_amd.dist): every rank reaches the same verdict on
    (a) whether all ranks built the same pair split and (b) which exchange protocol the autotune keeps."""
    import pickle

    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_agree_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (pickle.loads((tmp_path / f"agree_{r}.pkl").read_bytes()) for r in range(world))
    assert r0 == r1                                                   # identical on every rank: nobody takes another branch
    assert r0["same"][0] is True and r0["same"][1] == r0["same"][2]
    assert r0["diff"][0] is False and r0["diff"][1][2] == 43 and r0["diff"][2][2] == 44
    assert r0["failed"][0] is False and r0["failed"][1][0] == 0      # min(created) == 0: some rank failed nb_create
    assert r0["tune"][0] == "allgather" and abs(r0["tune"][1]["symmetric"] - 1.6e-3) < 1e-12
    assert r0["tie"][0] == "symmetric"
    assert r0["only_ag"][0] == "allgather"


def _allreduce_worker(rank, world, port, n, steps, out_dir):
    """The replicated protocol's data flow on CPU: this rank's share of the unordered pairs (product planner) evaluated in
    numpy fp64 -> partial acceleration of ALL particles, dist.all_reduce in place, every rank kicks and drifts all n."""
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    import nbodysim_amd as nb
    from nbodysim_amd import _lib as L
    sys.path.insert(0, str(ROOT / "tests"))
    import hooks

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        items, info = hooks.sym_plan(n, 256, rank, world)
        ic = nb.plummer_2d(n, 11)
        m = ic["mass"].astype(np.float64)
        eps2, dt = 0.05 ** 2, 1e-3
        x = ic["pos"].astype(np.float64).copy()
        v = ic["vel"].astype(np.float64).copy()
        for _ in range(steps):
            acc = np.zeros((n, 2))
            for it in items:
                tile, c0, c, diag = int(it["tile"]), int(it["c0"]), int(it["cnt"]), int(it["diag"])
                sb = info["tile_particles"]
                S = slice(tile * sb, min((tile + 1) * sb, n))
                T = slice(c0 * 64, min((c0 + c) * 64, n))
                d = x[None, T, :] - x[S, None, :]
                inv3 = (d[..., 0] ** 2 + d[..., 1] ** 2 + eps2) ** -1.5
                acc[S] += np.einsum("ij,ijk->ik", inv3 * m[None, T], d)
                if not diag:
                    acc[T] -= np.einsum("ij,ijk->jk", inv3 * m[S, None], d)
            t = torch.from_numpy(acc)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)           # in place; the same bits on every rank
            v += t.numpy() * dt
            x += v * dt
        np.save(Path(out_dir) / f"arx_rank{rank}.npy", x)
        np.save(Path(out_dir) / f"arv_rank{rank}.npy", v)
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_allreduce_protocol_replicas_are_identical_and_match_unsharded(tmp_path, nbo):
    import torch.multiprocessing as mp

    import nbodysim_amd as nb
    n, world, steps = 8192, 2, 2
    mp.spawn(_allreduce_worker, args=(world, _free_port(), n, steps, str(tmp_path)), nprocs=world, join=True)
    st = nbo.step_f64(nbo.state_from_bodies(nb.plummer_2d(n, 11), np.float64), 0.05, 1e-3, steps)
    want_pos, want_vel = np.stack([st["x"], st["y"]], 1), np.stack([st["vx"], st["vy"]], 1)
    xs = [np.load(tmp_path / f"arx_rank{r}.npy") for r in range(world)]
    vs = [np.load(tmp_path / f"arv_rank{r}.npy") for r in range(world)]
    assert np.array_equal(xs[0], xs[1]) and np.array_equal(vs[0], vs[1])          # replicas: bit-identical
    assert np.max(np.abs(xs[0] - want_pos)) < 1e-12 * np.max(np.abs(want_pos))
    assert np.max(np.abs(vs[0] - want_vel)) < 1e-11 * np.max(np.abs(want_vel))


def test_shard_plan_blocks_cover_the_range_also_when_ragged():
    """ShardPlan: ceil(n / world) particles per rank, a shorter last block, equal collective counts over a padded replica."""
    from nbodysim_amd.dist import ShardPlan
    for n, world in ((262144, 8), (10007, 3), (20001, 2), (25000, 7), (15, 8)):
        plans = [ShardPlan(n, world, r) for r in range(world)]
        assert plans[0].i_begin == 0 and plans[-1].i_end == n
        assert all(a.i_end == b.i_begin for a, b in zip(plans, plans[1:]))
        assert all(1 <= p.i_count <= p.stride for p in plans) and sum(p.i_count for p in plans) == n
        assert plans[0].padded_n == world * plans[0].stride >= n and plans[0].ragged == (n % world != 0)
        assert [p.block(r) for p in plans[:1] for r in range(world)] == [slice(q.i_begin, q.i_end) for q in plans]
    with pytest.raises(ValueError):
        ShardPlan(7, 8, 0)                      # some rank would own nothing
    with pytest.raises(ValueError):
        ShardPlan(100, 4, 4)
