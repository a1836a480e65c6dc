#!/usr/bin/env python3
"""Long runs of the default paths at the end of round 4 (dynamic items, small-system plans): finiteness, energy, momentum.
    python tools/r04_soak_default.py"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import nbodysim_amd as nb
for name, n, steps, kw, dt in (("p10000", 10000, 200000, dict(eps=0.05), 1e-3), ("ref25000", 25000, 60000, dict(eps=1.0, extras=3), 0.01), ("p65536", 65536, 15000, dict(eps=0.05), 1e-3), ("p262144 fp64", 262144, 300, dict(eps=0.05, precision="fp64"), 1e-3)):
    ic = nb.default_ics(n) if name.startswith("ref") else nb.plummer_2d(n, 42)
    t0 = time.time()
    with nb.Simulation(ic, **kw) as s:
        k0, u0 = s.energy()
        done = 0
        while done < steps:
            b = min(20000, steps - done); s.advance(b, dt); s.wait(); done += b
        k1, u1 = s.energy()
        (px, py, pz), lz = s.momentum()
        bodies = s.sync()
        ok = bool(np.all(np.isfinite(bodies["pos"])) and np.all(np.isfinite(bodies["vel"])))
    el = time.time() - t0
    print(f"{name}: {steps} steps in {el:.1f} s ({el/steps*1e6:.1f} us/step incl. energy), finite {ok}, energy {k0+u0:.6g} -> {k1+u1:.6g} (rel {(k1+u1-k0-u0)/abs(k0+u0):+.2e}), |p| {np.hypot(px,py):.2e}", flush=True)
