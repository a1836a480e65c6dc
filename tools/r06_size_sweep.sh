#!/bin/bash
# tools/r06_size_sweep.sh — the README table's rows at the final tree: one bench line per size / variant (no CPU baseline, no secondaries)
set -u
mkdir -p gpurun_out/r06s
for n in 10000 16384 32768 65536 98304 131072 1048576; do
  timeout -k 10 200 python bench.py --nbodies $n --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-live-pmc > gpurun_out/r06s/n$n.log 2> gpurun_out/r06s/n$n.err || echo "n=$n failed"
done
timeout -k 10 200 python bench.py --dims 3 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-live-pmc > gpurun_out/r06s/d3.log 2> gpurun_out/r06s/d3.err || echo "3-D failed"
timeout -k 10 200 python bench.py --dims 3 --precision fp64 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-live-pmc > gpurun_out/r06s/d3_fp64.log 2> gpurun_out/r06s/d3_fp64.err || echo "3-D fp64 failed"
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r06s/*.log")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l); rf = d["roofline"]
            print(f"{f.split('/')[-1]:14s} n={d['config']['n']:8d} {d['dtype']} dims={d['config']['dims']} ms/step {d['ms_per_step']:.4f} (settled {d['sustained']['ms_per_step']:.4f})  pairs/s {d['value']:.3e}  frac {rf['frac']:.3f} settled {rf['frac_sustained']:.3f}  kernel {rf['kernel_instantiation'] or rf['kernel']}")
PY
