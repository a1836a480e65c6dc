#!/usr/bin/env python3
"""Symmetric wave-split kernel against the one-sided kernel below 16 384 bodies.  python tools/small_n_check.py"""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import nbodysim_amd as nb

def run(ic, steps, **kw):
    with nb.Simulation(ic, eps=0.01, **kw) as s:
        acc = s.accelerations()
        s.advance(50, 1e-3); s.wait()
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter(); s.advance(steps, 1e-3); s.wait(); best = min(best, (time.perf_counter() - t0) / steps)
        d = s.describe()
    return best * 1e6, acc, d

for n in (4096, 5000, 6144, 8192, 10000, 12288, 14336, 16384, 20000):
    for um in (True, False):
        ic = nb.plummer_2d(n, 42)
        one, a1, _ = run(ic, 500, symmetry=False, uniform_mass=um)
        res = [f"one-sided {one:7.1f} us"]
        for L in (0, 4, 8):
            sym, a2, d = run(ic, 500, uniform_mass=um, sym_chunks_per_item=L)
            err = float(np.max(np.abs(a2.astype(np.float64) - a1)) / np.max(np.abs(a1)))
            res.append(f"sym L={L}: {sym:7.1f} us ({(sym/one-1)*100:+.0f} %, |da| {err:.1e}, {'sym' if 'symmetric=1' in d else 'NOT-sym'})")
        print(f"n={n:6d} uniform={int(um)}  " + "  ".join(res), flush=True)
