#!/bin/bash
# tools/tq_small_sweep.sh — one vs two travelling particles per lane (build/variants/lib_tq1.so, lib_tq2_u2_evenL.so) across N,
# planner's own L (0) and forced even L; equal masses and (general) individual masses.  -> gpurun_out/tq_small_sweep.log
set -u
out=gpurun_out/tq_small_sweep.log
: > $out
run() { # n L lib extra
  NBODY_HIP_LIB=$PWD/build/variants/$3 timeout -k 10 120 python bench.py --n $1 --chunks-per-item $2 --steps $5 --warmup 10 --no-cpu-baseline --no-sustained --no-secondary $4 2>/dev/null | python -c "
import json,sys,re
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); m=re.search(r'items=(\d+) chunks/item=(\d+)', d['config']['launch'])
print('n=$1 $3 $4 L=%s items=%s: %.1f us/step  kernel %.1f us' % (m.group(2), m.group(1), d['ms_per_step']*1e3, d['roofline']['avg_launch_ms']*1e3))" >> $out
}
for n in 16384 25000 32768 65536 131072 262144; do
  steps=100; [ $n -ge 131072 ] && steps=30
  for L in 0 2 4 6; do
    [ $n -ge 131072 ] && [ $L -ne 0 ] && continue
    run $n $L lib_tq1.so "" $steps
    run $n $L lib_tq2_u2_evenL.so "" $steps
  done
  run $n 0 lib_tq1.so "--general-mass" $steps
  run $n 0 lib_tq2_u2_evenL.so "--general-mass" $steps
done
cat $out
