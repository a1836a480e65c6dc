#!/usr/bin/env python3
"""Persistent step pipeline (one launch for nb_step's whole loop) against two launches per step, same handle parameters.

    python tools/pipe_ab.py [--cases ref25000,p16384,p32768,p65536,p131072,p262144] [--general] [--extra sym_tile=512,...]
"""
from __future__ import annotations

import argparse
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import nbodysim_amd as nb  # noqa: E402

PEAK = 157.3e12


def workload(name):
    if name.startswith("ref"):
        return nb.default_ics(int(name[3:])), dict(eps=1.0, extras=3), 0.01
    return nb.plummer_2d(int(name[1:]), 42), dict(eps=0.01), 1e-3


def run(ic, kw, dt, steps, reps, per_step=False, **tune):
    best = None
    with nb.Simulation(ic, **kw, **tune) as s:
        s.advance(max(10, steps // 4), dt)
        s.wait()
        for _ in range(reps):
            t0 = time.perf_counter()
            if per_step:                       # one launch per step: no waiting inside the launch, the gather fused into its end
                for _ in range(steps):
                    s.advance(1, dt)
            else:
                s.advance(steps, dt)
            s.wait()
            el = (time.perf_counter() - t0) / steps
            best = el if best is None else min(best, el)
        info, desc = s.sym_info(), s.describe()
    return best * 1e3, info, desc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="ref25000,p16384,p32768,p65536,p131072,p262144")
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--general", action="store_true")
    ap.add_argument("--extra", default="", help="comma-separated key=int tuning fields passed to both sides, e.g. sym_tile=512,sym_chunks_per_item=8")
    args = ap.parse_args()
    extra = {k: int(v) for k, v in (kv.split("=") for kv in args.extra.split(",") if kv)}
    for name in args.cases.split(","):
        ic, kw, dt = workload(name)
        n = ic.shape[0]
        if args.general:
            kw["uniform_mass"] = False
        steps = max(20, min(args.steps, int(args.steps * (65536.0 / n) ** 2)))
        frac = lambda ms: 14.0 * n * n / (ms * 1e-3) / PEAK
        two, info, _ = run(ic, kw, dt, steps, args.reps, pipeline=False, **extra)
        one, info1, desc = run(ic, kw, dt, steps, args.reps, pipeline=True, **extra)
        fused, _, _ = run(ic, kw, dt, steps, args.reps, per_step=True, pipeline=True, **extra)
        print(f"{name:10s} pipeline launched once per step (gather fused into the launch, no cross-step waits): {fused*1e3:9.1f} us/step ({(fused/two-1)*100:+.1f} %)", flush=True)
        print(f"{name:10s} n={n:7d} tile={info['tile_particles']:4d} L={info['chunks_per_item']:3d} items={info['items']:5d} steps={steps:3d} | "
              f"two launches {two*1e3:9.1f} us/step frac {frac(two):.3f} | pipeline {one*1e3:9.1f} us/step frac {frac(one):.3f} ({(one/two-1)*100:+.1f} %)"
              f" | {'pipeline=1' in desc}", flush=True)


if __name__ == "__main__":
    main()
