#!/bin/bash
# chunks-per-item sweep of the symmetric kernel at small and medium N (L=0: the planner's own choice)
for n in 16384 25000 32768 65536 131072; do
 for L in 0 1 2 3 4 6 8 12 16; do
  echo -n "n=$n L=$L: "
  python bench.py --n $n --chunks-per-item $L --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step']*1000,1),'us/step kernel', round(d['roofline']['avg_launch_ms']*1000,1), d['config']['launch'].split('|')[-2])"
 done
done
