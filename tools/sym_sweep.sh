#!/bin/bash
# sweep the chunks-per-item of the symmetric kernel on one box
for L in 4 8 16 32 64 128 256; do
  echo -n "L=$L: "
  python bench.py --chunks-per-item $L --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step'],3),'ms/step kernel', round(d['roofline']['avg_launch_ms'],3), d['config']['launch'].split('|')[-2])"
done
echo -n "one-sided: "; python bench.py --no-symmetry --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step'],3),'ms/step kernel', round(d['roofline']['avg_launch_ms'],3))"
