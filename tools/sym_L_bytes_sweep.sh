#!/bin/bash
# tools/sym_L_bytes_sweep.sh — slab bytes against kernel time as the symmetric kernel's work items grow.
# An item of tile (2048 stationary particles) x L chunks (64 L travelling particles) evaluates W = 131072 L pairs and writes
# 8 (2048 + 64 L) bytes of partial sums once: total bytes ~ 8 N^2 / 2 (1 / (64 L) + 1 / 2048) — fewer, bigger items are the ONLY
# lever on the slab traffic of a plain-store (no atomics) scheme, and they cost parallelism.  Output: one line per L with the
# launch time (HIP events) and the plan's bytes -> gpurun_out/sym_L_bytes_sweep.log
set -u
mkdir -p gpurun_out
out=gpurun_out/sym_L_bytes_sweep.log
: > $out
for L in 0 24 43 64 96 128 192 256 384; do
  timeout -k 10 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-sustained --chunks-per-item $L 2> /dev/null | \
    python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0])
r = d['roofline']; import re
m = re.search(r'items=(\d+) chunks/item=(\d+)', d['config']['launch'])
print(f\"L={m.group(2):>4s} items={m.group(1):>6s}  launch {r['avg_launch_ms']:.3f} ms  step {d['ms_per_step']:.3f} ms  frac {r['frac']:.4f}  slab bytes/launch {r['traffic']/1e6:7.1f} MB = {r['traffic']/(36*d['config']['n']):5.1f} x algorithmic\")
" >> $out
done
cat $out
