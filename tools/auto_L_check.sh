#!/bin/bash
# the planner's own chunks-per-item choice across N (compare with tools/sym_small_sweep.sh)
for n in 16384 25000 32768 49152 65536 98304 131072 262144; do
  echo -n "n=$n auto: "
  python bench.py --n $n --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step']*1000,1),'us/step kernel', round(d['roofline']['avg_launch_ms']*1000,1), d['config']['launch'].split('|')[-2], 'frac', round(d['roofline']['frac'],3), 'value %.3e' % d['value'])"
done
