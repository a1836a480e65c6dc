#!/usr/bin/env python3
"""tools/shard_sweep.py — sweep launch geometry of one rank of an 8-way split on one GPU."""
import os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import nbodysim_amd as nb

n, parts, rank, steps = 262144, int(sys.argv[1]) if len(sys.argv) > 1 else 8, 4, 40
rank = parts // 2
ic = nb.plummer_2d(n, 42)
icount = n // parts
res = []
for P in ("4", "2", "1"):
    for js in (0, 56, 64, 112, 128, 224, 256, 448, 896):
        with nb.Simulation(ic, eps=0.01, i_begin=rank * icount, i_count=icount, j_slices=js, lanes_p=int(P), symmetry=False) as sim:
            for _ in range(3):
                sim.step_begin(1e-3); sim.step_finish()
            sim.wait(); sim.profile(True)
            t0 = time.perf_counter()
            for _ in range(steps):
                sim.step_begin(1e-3); sim.step_finish()
            sim.wait(); t = (time.perf_counter() - t0) / steps * 1e3
            ms, cnt = sim.profile_read()
            d = sim.describe().split("|")[3].strip()
        res.append((t, P, js, ms / steps, d))
        print(f"P={P} js={js:4d}: {t:.3f} ms/step  kernels {ms/steps:.3f}  {d}", flush=True)
print("best:", min(res))
