#!/usr/bin/env python3
"""tools/tail_sweep.py — guided-tail thresholds of the symmetric planner with the chunk-pair kernel (N = 262 144, one GPU):
ms per step for a few (a, b, c) = fractions of the launch's work from which items are cut into L/2, L/4, L/8 chunks."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import nbodysim_amd as nb  # noqa: E402

n, steps = 262144, 40
ic = nb.plummer_2d(n, 42)
cands = [("default 0.85/0.94/0.98", None), ("0.80/0.92/0.97", (0.80, 0.92, 0.97)), ("0.90/0.96/0.99", (0.90, 0.96, 0.99)),
         ("0.75/0.90/0.96", (0.75, 0.90, 0.96)), ("0.88/0.95/0.985", (0.88, 0.95, 0.985)), ("no guided tail", "off")]
for rnd in (1, 2):
    for name, tail in cands:
        kw = dict(guided_tail=False) if tail == "off" else (dict(sym_tail=tail) if tail else {})
        with nb.Simulation(ic, eps=0.01, **kw) as sim:
            sim.advance(6, 1e-3); sim.wait()
            sim.profile(True)
            t0 = time.perf_counter(); sim.advance(steps, 1e-3); sim.wait()
            wall = (time.perf_counter() - t0) / steps * 1e3
            ms, cnt = sim.profile_read()
            items = sim.sym_info()["items"]
        print(f"round {rnd} {name:24s} items={items:6d}  launch {ms / cnt:.3f} ms  step {wall:.3f} ms", flush=True)
