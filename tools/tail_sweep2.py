#!/usr/bin/env python3
"""Guided-tail thresholds and chunks per item for the mid sizes (two launches per step).  python tools/tail_sweep2.py"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import nbodysim_amd as nb

def run(ic, kw, dt, steps, **tune):
    with nb.Simulation(ic, **kw, **tune) as s:
        s.advance(20, dt); s.wait()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); s.advance(steps, dt); s.wait(); best = min(best, (time.perf_counter() - t0) / steps)
        info = s.sym_info()
    return best * 1e6, info

for name, n, kw, dt, Ls in (("ref25000", 25000, dict(eps=1.0, extras=3), 0.01, (4, 8)), ("p32768", 32768, dict(eps=0.01), 1e-3, (4, 8)),
                            ("p65536", 65536, dict(eps=0.01), 1e-3, (4, 6, 8)), ("p131072", 131072, dict(eps=0.01), 1e-3, (12, 14, 16))):
    ic = nb.default_ics(n) if name.startswith("ref") else nb.plummer_2d(n, 42)
    steps = 300 if n <= 65536 else 80
    base, info = run(ic, kw, dt, steps)
    print(f"== {name}: default L={info['chunks_per_item']} items={info['items']} tile={info['tile_particles']}: {base:8.1f} us/step", flush=True)
    for L in Ls:
        for tail in ((0.85, 0.94, 0.98), (0.75, 0.9, 0.97), (0.65, 0.85, 0.95), (0.5, 0.8, 0.93), (0.4, 0.7, 0.9)):
            us, info = run(ic, kw, dt, steps, sym_chunks_per_item=L, sym_tail=tail)
            print(f"   L={L:3d} tail={tail}: items={info['items']:5d}  {us:8.1f} us/step ({(us/base-1)*100:+.1f} %)", flush=True)
