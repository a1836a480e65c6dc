#!/usr/bin/env python3
"""Long runs of the hand-off heavy paths of round 4, compared bit for bit against their references:
  * the experimental step pipeline (one persistent launch per batch of steps) and the one-launch step (gather workgroups in the
    drain of the force launch) against two launches per step, with dynamic (default) and static work items;
  * (python tests/loopback_worker.py covers the C loop over the loopback transport: here only its step count is raised).
    python tools/r04_soak.py"""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import nbodysim_amd as nb

def bits(a): return np.ascontiguousarray(a).view(np.uint32)

for name, n, kw, dt, batches in (("p16384", 16384, dict(eps=0.05), 1e-3, [1000] * 6), ("ref25000", 25000, dict(eps=1.0, extras=3), 0.01, [700] * 4),
                                 ("p20001 quake general", 20001, dict(eps=0.01, rsqrt="quake", uniform_mass=False), 1e-3, [500] * 4),
                                 ("p65536", 65536, dict(eps=0.01), 1e-3, [300] * 4), ("p65536 tile 512", 65536, dict(eps=0.01, sym_tile=512), 1e-3, [300] * 3),
                                 ("p262144", 262144, dict(eps=0.01), 1e-3, [60] * 3)):
    ic = nb.default_ics(n) if name.startswith("ref") else nb.plummer_2d(n, 42)
    t0 = time.time()
    with nb.Simulation(ic, pipeline=True, **kw) as a, nb.Simulation(ic, pipeline=False, **kw) as b, nb.Simulation(ic, one_launch=True, **kw) as c, nb.Simulation(ic, static_items=True, **kw) as d:
        total, ok = 0, True
        for k in batches:
            a.advance(k, dt); b.advance(k, dt); c.advance(k, dt); d.advance(k, dt)
            x, y, z, w = a.sync(), b.sync(), c.sync(), d.sync()
            total += k
            same = all(np.array_equal(bits(x[f]), bits(y[f])) and np.array_equal(bits(z[f]), bits(y[f])) and np.array_equal(bits(w[f]), bits(y[f])) for f in ("pos", "vel", "acc"))
            ok = ok and same
            if not same:
                bad = int(np.sum(np.any(bits(x["pos"]) != bits(y["pos"]), axis=1)))
                print(f"   {name}: MISMATCH after {total} steps: {bad} bodies differ", flush=True)
                break
    print(f"{name}: {total} steps, pipeline == one launch per step == two launches per step (dynamic items) == two launches (static items), bit for bit: {ok}  ({time.time() - t0:.1f} s)", flush=True)
