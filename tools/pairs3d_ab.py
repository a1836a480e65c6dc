#!/usr/bin/env python3
"""tools/pairs3d_ab.py — 3-D fp32 symmetric kernel with single chunks (-1) vs chunk pairs (+1), N = 262 144 and 65 536, equal and
individual masses; accelerations of both forms compared."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import nbodysim_amd as nb  # noqa: E402

for n, steps in ((262144, 30), (65536, 150)):
    ic = nb.plummer_3d(n, 42)
    for general in (False, True):
        acc = {}
        for rnd in (1, 2):
            for pairs in (-1, 1):
                with nb.Simulation(ic, eps=0.01, dims=3, sym_chunk_pairs=pairs, uniform_mass=not general) as sim:
                    if rnd == 1:
                        acc[pairs] = sim.accelerations().astype(np.float64)
                    sim.advance(5, 1e-3); sim.wait()
                    sim.profile(True)
                    t0 = time.perf_counter(); sim.advance(steps, 1e-3); sim.wait()
                    wall = (time.perf_counter() - t0) / steps * 1e3
                    ms, cnt = sim.profile_read()
                    d = sim.describe()
                print(f"n={n} {'individual' if general else 'equal'} masses round {rnd} chunk_pairs={pairs:+d}: launch {ms / cnt:.3f} ms  step {wall:.3f} ms  "
                      + d[d.index('symmetric='):d.index('slabs')], flush=True)
        a, b = acc[-1], acc[1]
        print(f"   max |a_pairs - a_single| / max|a| = {np.max(np.abs(a - b)) / np.max(np.abs(a)):.2e}", flush=True)
