#!/bin/bash
# bench every library variant under build/variants on one box, two interleaved rounds
for round in 1 2; do
for lib in build/variants/lib_*.so; do
  echo -n "$round $(basename $lib): "
  NBODY_HIP_LIB=$PWD/$lib python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step'],3),'ms/step kernel', round(d['roofline']['avg_launch_ms'],3))"
done; done
