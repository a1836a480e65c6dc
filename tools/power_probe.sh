#!/bin/bash
# tools/power_probe.sh — sample clocks and power of GPU 0 while the benchmark's kernel runs back to back, to see what
# holds the chip below its 2.4 GHz peak clock under the packed-FMA load.  Ordinary user, read-only.
mkdir -p gpurun_out
( python bench.py --steps 1500 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/power_probe_bench.log 2>&1 ) &
BPID=$!
sleep 4
for i in $(seq 1 10); do
  echo "--- sample $i"
  rocm-smi -d 0 --showpower --showclocks --showperflevel --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|Perf|Temperature \(Sensor (edge|junction|hotspot)" | sed 's/^GPU\[0\]\s*: //'
  sleep 0.7
done
wait $BPID
tail -1 gpurun_out/power_probe_bench.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench during probe: ms/step', round(d['ms_per_step'],3), 'frac', round(d['roofline']['frac'],4))"
rocm-smi -d 0 --showmaxpower --showpowerprofile 2>/dev/null | grep -v "^=\|^$" | head -20
