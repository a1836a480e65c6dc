#!/usr/bin/env python3
"""The planner's own choices against round-3's (forced) at the small and mid sizes.  python tools/defaults_check.py"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import nbodysim_amd as nb

def run(ic, kw, dt, steps, **tune):
    with nb.Simulation(ic, **kw, **tune) as s:
        s.advance(20, dt); s.wait()
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter(); s.advance(steps, dt); s.wait(); best = min(best, (time.perf_counter() - t0) / steps)
        info = s.sym_info()
    return best * 1e6, info

old_L = {16384: 4, 25000: 4, 32768: 8, 49152: 4, 65536: 6, 98304: 10, 131072: 14, 262144: 44}
for rnd in (1, 2):
  for name, n, kw, dt in (("p16384", 16384, dict(eps=0.01), 1e-3), ("ref25000", 25000, dict(eps=1.0, extras=3), 0.01), ("p32768", 32768, dict(eps=0.01), 1e-3),
                        ("p65536", 65536, dict(eps=0.01), 1e-3), ("p98304", 98304, dict(eps=0.01), 1e-3), ("p131072", 131072, dict(eps=0.01), 1e-3), ("p262144", 262144, dict(eps=0.01), 1e-3)):
    ic = nb.default_ics(n) if name.startswith("ref") else nb.plummer_2d(n, 42)
    steps = max(20, min(300, int(300 * (65536.0 / n) ** 2)))
    new, info = run(ic, kw, dt, steps)
    old, info0 = run(ic, kw, dt, steps, sym_chunks_per_item=old_L[n], sym_tail=(0.85, 0.94, 0.98))
    print(f"round {rnd} {name:9s} new: L={info['chunks_per_item']:3d} items={info['items']:5d} {new:9.1f} us/step frac {14.0*n*n/new*1e6/157.3e12:.3f} | round-3 plan: L={info0['chunks_per_item']:3d} items={info0['items']:5d} {old:9.1f} us ({(new/old-1)*100:+.1f} %)", flush=True)
