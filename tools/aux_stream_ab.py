#!/usr/bin/env python3
"""A/B on one GPU of running a rank's local items on a side stream (concurrent with its cross items): wall ms/step."""
import os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import nbodysim_amd as nb
steps, n = 60, 262144
ic = nb.plummer_2d(n, 42)
for parts in (2, 4, 8):
    for rep in range(3):
        for aux in (0, 1):
            rank = parts // 2
            with nb.Simulation(ic, eps=0.01, i_begin=rank * (n // parts), i_count=n // parts, shard_rank=rank, shard_world=parts,
                               sym_aux_stream=1 if aux else -1) as sim:
                def go(k):
                    for _ in range(k): sim.step_begin(1e-3); sim.step_mid(); sim.step_finish()
                go(5); sim.wait()
                t0 = time.perf_counter(); go(steps); sim.wait(); t = (time.perf_counter() - t0) / steps * 1e3
                print(f"n={n} parts={parts} side_stream={aux}: {t:.4f} ms/step", flush=True)
