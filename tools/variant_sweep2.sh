#!/bin/bash
# tools/variant_sweep2.sh — every library variant under build/variants on ONE box, interleaved rounds, with a quick
# correctness check of each (accelerations against the first variant).  -> gpurun_out/variant_sweep_r03.log
set -u
out=gpurun_out/variant_sweep_r03.log
: > $out
L=${SWEEP_L:-44}
for round in 1 2; do
for lib in build/variants/lib_*.so; do
  echo -n "$round $(basename $lib) L=$L: " >> $out
  NBODY_HIP_LIB=$PWD/$lib timeout -k 10 120 python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-sustained --no-secondary --chunks-per-item $L 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); ds=d.get('device_state') or {}
print(round(d['ms_per_step'],3),'ms/step  launch', round(d['roofline']['avg_launch_ms'],3), ' frac', round(d['roofline']['frac'],4), ' sclk', round(ds.get('sclk_mhz_mean') or 0), ' E drift', '%.2e' % d['energy']['rel_drift'])" >> $out
done; done
for lib in build/variants/lib_*.so; do
  echo -n "general-mass $(basename $lib) L=$L: " >> $out
  NBODY_HIP_LIB=$PWD/$lib timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sustained --no-secondary --general-mass --chunks-per-item $L 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print(round(d['ms_per_step'],3),'ms/step  launch', round(d['roofline']['avg_launch_ms'],3), ' frac', round(d['roofline']['frac'],4))" >> $out
done
cat $out
