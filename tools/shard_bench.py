#!/usr/bin/env python3
"""tools/shard_bench.py — what one rank of a P-way strong-scaling run costs per step,
measured on ONE GPU without the exchange: a sharded handle (i_count = n/P) driven through
nb_step_begin / nb_step_finish.  Compares with the ideal (single-GPU step / P)."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import nbodysim_amd as nb  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
steps = 30
ic = nb.plummer_2d(n, 42)


def run(parts, rank):
    ic_count = n // parts
    kw = dict(i_begin=rank * ic_count, i_count=ic_count, shard_rank=rank, shard_world=parts) if parts > 1 else {}
    with nb.Simulation(ic, eps=0.01, **kw) as sim:
        def go(k):
            if parts == 1:
                sim.advance(k, 1e-3)
            else:
                for _ in range(k):
                    sim.step_begin(1e-3)
                    sim.step_mid()
                    sim.step_finish()
        go(3); sim.wait()
        sim.profile(True)
        t0 = time.perf_counter(); go(steps); sim.wait(); t = (time.perf_counter() - t0) / steps
        ms, cnt = sim.profile_read()
        return t * 1e3, ms / steps, sim.describe()


base, base_k, d = run(1, 0)
print(f"1 GPU : {base:.3f} ms/step (force kernels {base_k:.3f})  {d}")
for parts in (2, 4, 8):
    for rank in sorted({0, parts // 2, parts - 1}):
        t, k, d = run(parts, rank)
        print(f"{parts} ranks, rank {rank}: {t:.3f} ms/step (force kernels {k:.3f}) ideal {base / parts:.3f} -> efficiency {base / parts / t * 100:.1f}%  | {'|'.join(d.split('|')[3:5])}")
