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


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


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
    with pytest.raises(ValueError):
        ShardPlan(1000, 3, 0)
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
