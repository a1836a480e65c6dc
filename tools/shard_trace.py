#!/usr/bin/env python3
"""One rank's share of a P-way split stepped on one GPU (no transport), for rocprofv3 --kernel-trace --stats:
   rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/shardprof -- python3 tools/shard_trace.py 8 [symmetric|allreduce|allgather]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import nbodysim_amd as nb
parts = int(sys.argv[1]) if len(sys.argv) > 1 else 8
protocol = sys.argv[2] if len(sys.argv) > 2 else "symmetric"
n, steps = 262144, 40
ic = nb.plummer_2d(n, 42)
rank = parts // 2
own = dict(i_begin=0, i_count=n, shard_allreduce=True) if protocol == "allreduce" else dict(i_begin=rank * (n // parts), i_count=n // parts)
with nb.Simulation(ic, eps=0.01, shard_rank=rank, shard_world=parts, symmetry=protocol != "allgather", **own) as sim:
    for _ in range(steps):
        sim.step_begin(1e-3); sim.step_mid(); sim.step_finish()
    sim.wait()
    print(sim.describe())
