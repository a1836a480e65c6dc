#!/usr/bin/env python3
"""A/B of the guided tail (finer items at the end of a launch) on one GPU: whole system and one rank's share.
guided_tail=False (NB_FLAG_NO_GUIDED_TAIL) is the uniform plan; sym_tail=(a, b, c) sets the work fractions where items shrink to L/2, L/4, L/8."""
import os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import nbodysim_amd as nb
steps = 40
VARIANTS = [("uniform", None), ("default", ""), ("0.5,0.75,0.9", "0.5,0.75,0.9"), ("0.6,0.85,0.95", "0.6,0.85,0.95"),
            ("0.85,0.94,0.98", "0.85,0.94,0.98"), ("0.9,0.9,0.9", "0.9,0.9,0.9"), ("0.8,0.8,0.95", "0.8,0.8,0.95")]
n = 262144
ic = nb.plummer_2d(n, 42)
for parts in (1, 2, 8):
    for rep in range(2):
        for name, tail in VARIANTS:
            rank = parts // 2
            kw = dict(i_begin=rank * (n // parts), i_count=n // parts, shard_rank=rank, shard_world=parts) if parts > 1 else {}
            if tail is None: kw["guided_tail"] = False
            elif tail: kw["sym_tail"] = tuple(float(x) for x in tail.split(","))
            with nb.Simulation(ic, eps=0.01, **kw) as sim:
                def go(k):
                    if parts == 1: sim.advance(k, 1e-3)
                    else:
                        for _ in range(k): sim.step_begin(1e-3); sim.step_mid(); sim.step_finish()
                go(3); sim.wait(); sim.profile(True)
                t0 = time.perf_counter(); go(steps); sim.wait(); t = (time.perf_counter() - t0) / steps * 1e3
                ms, cnt = sim.profile_read()
                print(f"n={n} parts={parts} tail={name:16s}: {t:.3f} ms/step kernel {ms/steps:.3f}  {sim.describe().split('|')[4]}", flush=True)
