#!/usr/bin/env python3
"""One rank's compute per step at 2 / 4 / 8 ranks (collectives left out, one GPU): guided-tail thresholds and chunks per item.
    python tools/rank_tail_sweep.py [N]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import nbodysim_amd as nb

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
steps = 80
ic = nb.plummer_2d(n, 42)
with nb.Simulation(ic, eps=0.01) as sim:
    sim.advance(5, 1e-3); sim.wait()
    t0 = time.perf_counter(); sim.advance(40, 1e-3); sim.wait()
    base = (time.perf_counter() - t0) / 40 * 1e3
print(f"n={n} one GPU {base:.3f} ms/step", flush=True)
tails = (None, (0.75, 0.9, 0.97), (0.65, 0.85, 0.95), (0.5, 0.8, 0.93))
for parts in (2, 4, 8):
    rank, blk = parts // 2, n // parts
    for name, kw in (("sym", dict(sym_late_us=-1.0, i_begin=rank * blk, i_count=blk)), ("sym+late", dict(i_begin=rank * blk, i_count=blk, sym_late_us=40.0)),
                     ("allreduce", dict(shard_allreduce=True, i_begin=0, i_count=n))):
        for L in (0, 8, 12, 16):
            row = []
            for tail in tails:
                tk = dict(sym_tail=tail) if tail else {}
                best = 1e9
                info = None
                with nb.Simulation(ic, eps=0.01, shard_rank=rank, shard_world=parts, sym_chunks_per_item=L, **kw, **tk) as s:
                    def go(k):
                        for _ in range(k):
                            s.step_begin(1e-3); s.step_mid(); s.step_finish()
                    go(5); s.wait()
                    for _ in range(3):
                        t0 = time.perf_counter(); go(steps); s.wait(); best = min(best, (time.perf_counter() - t0) / steps * 1e3)
                    info = s.sym_info()
                row.append(f"{'default' if not tail else tail[0]}: {best:.4f} ({base / parts / best * 100:.1f} %)")
            print(f"  P={parts} {name:9s} L={info['chunks_per_item']:3d}{'*' if L == 0 else ' '} " + " | ".join(row), flush=True)
