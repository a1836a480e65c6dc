#!/usr/bin/env python3
"""A/B on one GPU of holding back nb_params.sym_late_us microseconds of local items for the side stream.  Without a
transport there is no reduce-scatter to hide, so this measures only what the split costs (wall ms/step)."""
import os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import nbodysim_amd as nb
steps, n = 60, 262144
ic = nb.plummer_2d(n, 42)
for precision in ("fp32", "fp64"):
    for parts in (2, 4, 8):
        for rep in range(2):
            for us in (0, 40):
                rank = parts // 2
                with nb.Simulation(ic, eps=0.01, precision=precision, i_begin=rank * (n // parts), i_count=n // parts,
                                   shard_rank=rank, shard_world=parts, sym_late_us=float(us) if us else -1.0) as sim:
                    def go(k):
                        for _ in range(k): sim.step_begin(1e-3); sim.step_mid(); sim.step_finish()
                    go(5); sim.wait()
                    t0 = time.perf_counter(); go(steps); sim.wait(); t = (time.perf_counter() - t0) / steps * 1e3
                    print(f"n={n} {precision} parts={parts} late_us={us:2d}: {t:.4f} ms/step {sim.describe().split('|')[4]}", flush=True)
