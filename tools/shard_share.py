#!/usr/bin/env python3
"""One rank's share of a P-way symmetric split stepped on one GPU (no transport): ms/step for any N / precision.
   python tools/shard_share.py N P [fp32|fp64]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import nbodysim_amd as nb
n, parts = int(sys.argv[1]), int(sys.argv[2])
precision = sys.argv[3] if len(sys.argv) > 3 else "fp32"
steps = 10
ic = nb.plummer_2d(n, 42)
for rank in (0, parts // 2, parts - 1):
    with nb.Simulation(ic, eps=0.01, precision=precision, i_begin=rank * (n // parts), i_count=n // parts,
                       shard_rank=rank, shard_world=parts) as sim:
        def go(k):
            for _ in range(k): sim.step_begin(1e-3); sim.step_mid(); sim.step_finish()
        go(2); sim.wait()
        t0 = time.perf_counter(); go(steps); sim.wait(); t = (time.perf_counter() - t0) / steps * 1e3
        print(f"n={n} {precision} P={parts} rank {rank}: {t:.3f} ms/step  protocol={sim.shard_protocol}  {sim.describe().split('|')[4]}", flush=True)
