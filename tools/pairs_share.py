#!/usr/bin/env python3
"""tools/pairs_share.py — one rank's compute per step (collectives left out, one GPU) with the symmetric kernel sweeping
single chunks (sym_chunk_pairs = -1) or chunk pairs (+1), at 2 / 4 / 8 ranks and on one GPU.
    python tools/pairs_share.py [N] [precision]"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import nbodysim_amd as nb  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
steps = 60
ic = nb.plummer_2d(n, 42)
for general in (False, True):
    print(f"n={n} {'individual' if general else 'equal'} masses", flush=True)
    for pairs in (-1, 1):
        with nb.Simulation(ic, eps=0.01, sym_chunk_pairs=pairs, uniform_mass=not general) as sim:
            sim.advance(5, 1e-3); sim.wait()
            t0 = time.perf_counter(); sim.advance(steps, 1e-3); sim.wait()
            base = (time.perf_counter() - t0) / steps * 1e3
        row = [f"one GPU {base:.3f}"]
        for parts in (2, 4, 8):
            rank, blk = parts // 2, n // parts
            for name, kw in (("sym", dict(sym_late_us=-1.0, i_begin=rank * blk, i_count=blk)), ("allreduce", dict(shard_allreduce=True, i_begin=0, i_count=n))):
                with nb.Simulation(ic, eps=0.01, shard_rank=rank, shard_world=parts, sym_chunk_pairs=pairs, uniform_mass=not general, **kw) as s:
                    def go(k):
                        for _ in range(k):
                            s.step_begin(1e-3); s.step_mid(); s.step_finish()
                    go(5); s.wait()
                    t0 = time.perf_counter(); go(steps); s.wait()
                    t = (time.perf_counter() - t0) / steps * 1e3
                row.append(f"P={parts} {name} {t:.3f}")
        print(f"  chunk_pairs={pairs:+d}: " + " | ".join(row), flush=True)
