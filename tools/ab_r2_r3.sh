#!/bin/bash
# tools/ab_r2_r3.sh — same-box A/B of the round-2 library (build/r2_tree, built from commit 23578c7) against the current one:
# bench.py of each tree, alternating, identical flags.  -> gpurun_out/ab_r2_r3.log
set -u
out=gpurun_out/ab_r2_r3.log
: > $out
for k in 1 2 3; do
  for tree in build/r2_tree .; do
    extra=""; [ "$tree" = "." ] && extra="--no-sustained"
    timeout -k 10 120 python $tree/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-secondary $extra 2> /dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0])
ds = d.get('device_state') or {}
print('$tree'.ljust(14), f\"step {d['ms_per_step']:.3f} ms  launch {d['roofline']['avg_launch_ms']:.3f} ms  frac {d['roofline']['frac']:.4f}  sclk {ds.get('sclk_mhz_mean')} MHz  power {ds.get('power_w_mean')} W\")
" >> $out
  done
done
cat $out
