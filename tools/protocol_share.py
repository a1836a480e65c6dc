#!/usr/bin/env python3
"""tools/protocol_share.py — what one rank's COMPUTE costs per step in each exchange protocol, measured on ONE GPU
without any transport (the collectives are simply left out, so the physics is wrong but the launches are the real
ones): symmetric (reduce-scatter + all-gather), the same with the late items flipped, allreduce (replicated
integration) and allgather (one-sided kernels).  N = 262 144 by default; wall ms per step, no HIP events.
    python tools/protocol_share.py [N] [precision]"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import nbodysim_amd as nb  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
precision = sys.argv[2] if len(sys.argv) > 2 else "fp32"
steps = 60
ic = nb.plummer_2d(n, 42)
with nb.Simulation(ic, eps=0.01, precision=precision) as sim:
    sim.advance(5, 1e-3); sim.wait()
    t0 = time.perf_counter(); sim.advance(steps, 1e-3); sim.wait()
    base = (time.perf_counter() - t0) / steps * 1e3
print(f"n={n} {precision}: one GPU {base:.3f} ms/step", flush=True)
for parts in (2, 4, 8):
    rank, blk = parts // 2, n // parts
    rows = []
    for name, kw in (("symmetric", dict(sym_late_us=-1.0)), ("symmetric+late", dict(sym_late_us=40.0)),
                     ("allreduce", dict(shard_allreduce=True)), ("allgather", dict(symmetry=False))):
        own = dict(i_begin=0, i_count=n) if name == "allreduce" else dict(i_begin=rank * blk, i_count=blk)
        with nb.Simulation(ic, eps=0.01, precision=precision, shard_rank=rank, shard_world=parts, **own, **kw) as s:
            def go(k):
                for _ in range(k):
                    s.step_begin(1e-3); s.step_mid(); s.step_finish()
            go(5); s.wait()
            t0 = time.perf_counter(); go(steps); s.wait()
            t = (time.perf_counter() - t0) / steps * 1e3
        rows.append(f"{name} {t:.3f} ({base / parts / t * 100:.1f}%)")
    print(f"  {parts} ranks, rank {rank}: ideal {base / parts:.3f} | " + " | ".join(rows), flush=True)
