#!/bin/bash
# tools/bench_variants.sh — the README table: bench.py on one box for the other sizes / precisions / kernels.  -> gpurun_out/bench_variants.log
set -u
out=gpurun_out/bench_variants.log
: > $out
run() {
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --no-sustained "$@" 2> /dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0]); r = d['roofline']; ds = d.get('device_state') or {}
print('$*'.ljust(42), f\"{d['ms_per_step']:9.3f} ms/step  {d['value']:.3e} pairs/s  frac {r['frac']:.4f}  executed {r['executed_frac'] or 0:.4f}  kernel {r['kernel']}  launch {r['avg_launch_ms']:.3f} ms  sclk {ds.get('sclk_mhz_mean') or 0:.0f} MHz  drift {d['energy']['rel_drift']:.1e}\")" >> $out
}
run --steps 30 --warmup 5
run --n 65536 --steps 200 --warmup 20
run --n 131072 --steps 60 --warmup 10
run --n 1048576 --steps 6 --warmup 2
run --steps 30 --warmup 5 --general-mass
run --steps 20 --warmup 4 --no-symmetry
run --steps 20 --warmup 4 --precision fp64
run --steps 20 --warmup 4 --dims 3
run --steps 12 --warmup 3 --dims 3 --precision fp64
cat $out
