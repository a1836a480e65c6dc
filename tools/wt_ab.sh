#!/bin/bash
# tools/wt_ab.sh — write-through (sc1) slab stores in force_sym_f32 against plain stores, two launches per step, same box, interleaved
for round in 1 2; do for v in plain wt; do
  echo "== round $round lib_$v"
  NBODY_HIP_LIB=$PWD/build/variants/lib_$v.so timeout -k 10 120 python - <<'PY'
import sys, time
sys.path.insert(0, '.')
import nbodysim_amd as nb
for name, n, kw, dt in (("ref25000", 25000, dict(eps=1.0, extras=3), 0.01), ("p65536", 65536, dict(eps=0.01), 1e-3), ("p262144", 262144, dict(eps=0.01), 1e-3)):
    ic = nb.default_ics(n) if name.startswith("ref") else nb.plummer_2d(n, 42)
    steps = max(20, int(300 * (65536.0 / n) ** 2)) if n > 65536 else 300
    with nb.Simulation(ic, pipeline=False, **kw) as s:
        s.advance(20, dt); s.wait()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); s.advance(steps, dt); s.wait(); best = min(best, (time.perf_counter() - t0) / steps)
    print(f"   {name:9s} {best*1e6:9.1f} us/step  frac {14.0*n*n/best/157.3e12:.3f}", flush=True)
PY
done; done
