#!/usr/bin/env python3
"""One rank's share of a P-way symmetric split stepped on one GPU (no transport), for rocprofv3 --kernel-trace --stats:
   rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/shardprof -- python3 tools/shard_trace.py 8"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import nbodysim_amd as nb
parts = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n, steps = 262144, 40
ic = nb.plummer_2d(n, 42)
rank = parts // 2
with nb.Simulation(ic, eps=0.01, i_begin=rank * (n // parts), i_count=n // parts, shard_rank=rank, shard_world=parts) as sim:
    for _ in range(steps):
        sim.step_begin(1e-3); sim.step_mid(); sim.step_finish()
    sim.wait()
    print(sim.describe())
