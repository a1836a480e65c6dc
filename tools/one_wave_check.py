#!/usr/bin/env python3
"""The plans of small systems (wave-split tiles: 4 chunks per item, no guided tail; nb_plan.cpp) against the rule they replace
(4 chunks per item with the early guided tail) and against the one-sided kernel: where does the symmetric path start to pay?
    python tools/one_wave_check.py"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import nbodysim_amd as nb


def run(ic, steps, **kw):
    with nb.Simulation(ic, eps=0.01, **kw) as s:
        s.advance(30, 1e-3); s.wait()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); s.advance(steps, 1e-3); s.wait(); best = min(best, (time.perf_counter() - t0) / steps)
        info, desc = s.sym_info(), s.describe()
    return best * 1e6, info, "symmetric=1" in desc


for rnd in range(2):
    for n in (4096, 5120, 5632, 6144, 6656, 7168, 8192, 9216, 10000, 12288, 14336, 16384, 18000, 20000, 22000, 25000, 28000, 32768, 36000, 40000, 45000):
        ic = nb.plummer_2d(n, 42)
        for general in (False, True):
            kw = dict(uniform_mass=not general)
            one, _, _ = run(ic, 400, symmetry=False, **kw)
            new, info, sym = run(ic, 400, sym_tile=512, **kw)            # sym_tile = 512 also lifts the size threshold of the symmetric path
            old, oinfo, _ = run(ic, 400, sym_tile=512, sym_chunks_per_item=4, sym_tail=(0.65, 0.85, 0.95), **kw)
            print(f"round {rnd + 1} n={n:6d} {'individual' if general else 'equal     '} masses | one-sided {one:7.1f} us | symmetric, plan as built: L={info['chunks_per_item']:2d} items={info['items']:5d} "
                  f"{new:7.1f} us ({(new / one - 1) * 100:+.0f} % vs one-sided) | L=4 early tail: items={oinfo['items']:5d} {old:7.1f} us ({(new / old - 1) * 100:+.0f} % new vs that) | sym={sym}", flush=True)
