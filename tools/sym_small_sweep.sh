#!/bin/bash
for n in 16384 32768 65536 131072; do
 for L in 1 2 4 8 16 32; do
  echo -n "n=$n L=$L: "
  NB_SYM_L=$L python bench.py --n $n --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step']*1000,1),'us/step kernel', round(d['roofline']['avg_launch_ms']*1000,1), d['config']['launch'].split('|')[-2])"
 done
done
