#!/usr/bin/env python3
"""Per-rank cost of the symmetric sharded protocol on one GPU for several chunks-per-item."""
import os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import nbodysim_amd as nb
n, steps = 262144, 40
ic = nb.plummer_2d(n, 42)
for parts in (8, 4, 2, 1):
    for L in (0, 2, 4, 8, 16, 32, 64):
        rank = parts // 2
        kw = dict(i_begin=rank * (n // parts), i_count=n // parts, shard_rank=rank, shard_world=parts) if parts > 1 else {}
        with nb.Simulation(ic, eps=0.01, sym_chunks_per_item=L, **kw) as sim:
            def go(k):
                if parts == 1: sim.advance(k, 1e-3)
                else:
                    for _ in range(k): sim.step_begin(1e-3); sim.step_mid(); sim.step_finish()
            go(3); sim.wait(); sim.profile(True)
            t0 = time.perf_counter(); go(steps); sim.wait(); t = (time.perf_counter() - t0) / steps * 1e3
            ms, cnt = sim.profile_read()
            print(f"parts={parts} L={L:3d}: {t:.3f} ms/step kernel {ms/steps:.3f}  {sim.describe().split('|')[4]}", flush=True)
