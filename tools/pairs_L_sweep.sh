out=gpurun_out/pairs_L_sweep.log; : > $out
for round in 1 2; do for L in 28 36 42 48 56 64 84; do
  timeout -k 10 120 python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-secondary --no-sustained --chunks-per-item $L 2> /dev/null | python -c "
import sys, json, re
d = json.loads([l for l in sys.stdin if l.startswith('{')][0]); r = d['roofline']
m = re.search(r'items=(\d+) chunks/item=(\d+)', d['config']['launch'])
print(f\"round $round L={m.group(2):>3s} items={m.group(1):>6s} launch {r['avg_launch_ms']:.3f} ms step {d['ms_per_step']:.3f} ms frac {r['frac']:.4f} slab {r['traffic']/1e6:6.1f} MB\")" >> $out
done; done; cat $out
