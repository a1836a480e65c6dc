#!/usr/bin/env python3
"""One launch per step (sym_step_f32: the gather + kick + drift workgroups ride in the drain of the force launch) against
two launches per step (force_sym_f32 + sym_gather), same handle parameters, interleaved rounds on one box; also checks
that the two forms leave bit-identical bodies.

    python tools/step_ab.py [--cases ref25000,p16384,...] [--general] [--rounds 2] [--extra sym_tile=512,...]
"""
from __future__ import annotations

import argparse
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import nbodysim_amd as nb  # noqa: E402

PEAK = 157.3e12


def workload(name):
    if name.startswith("ref"):
        return nb.default_ics(int(name[3:])), dict(eps=1.0, extras=3), 0.01
    return nb.plummer_2d(int(name[1:]), 42), dict(eps=0.01), 1e-3


def run(ic, kw, dt, steps, reps, **tune):
    best = None
    with nb.Simulation(ic, **kw, **tune) as s:
        s.advance(max(10, steps // 4), dt)
        s.wait()
        for _ in range(reps):
            t0 = time.perf_counter()
            s.advance(steps, dt)
            s.wait()
            el = (time.perf_counter() - t0) / steps
            best = el if best is None else min(best, el)
        info, desc = s.sym_info(), s.describe()
        end = s.sync().copy()
    return best * 1e3, info, desc, end


def same(a, b):
    return all(np.array_equal(np.ascontiguousarray(a[f]).view(np.uint32), np.ascontiguousarray(b[f]).view(np.uint32)) for f in ("pos", "vel", "acc"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="p9216,p16384,ref25000,p32768,p49152,p65536,p131072,p262144")
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--general", action="store_true")
    ap.add_argument("--what", default="one_launch", choices=("one_launch", "dynamic_items"), help="the switch under test: one launch per step against two (default), or dynamic against static work items")
    ap.add_argument("--extra", default="", help="comma-separated key=int tuning fields passed to both sides")
    args = ap.parse_args()
    extra = {k: int(v) for k, v in (kv.split("=") for kv in args.extra.split(",") if kv)}
    for rnd in range(args.rounds):
        for name in args.cases.split(","):
            ic, kw, dt = workload(name)
            n = ic.shape[0]
            if args.general:
                kw["uniform_mass"] = False
            steps = max(20, min(args.steps, int(args.steps * (65536.0 / n) ** 2)))
            frac = lambda ms: 14.0 * n * n / (ms * 1e-3) / PEAK
            if args.what == "dynamic_items":
                two, info, d2, e2 = run(ic, kw, dt, steps, args.reps, static_items=True, **extra)
                one, _, d1, e1 = run(ic, kw, dt, steps, args.reps, static_items=False, **extra)
                print(f"round {rnd + 1} {name:10s} n={n:7d} tile={info['tile_particles']:4d} L={info['chunks_per_item']:3d} items={info['items']:5d} steps={steps:3d} | "
                      f"static items {two*1e3:9.1f} us/step frac {frac(two):.3f} | dynamic items {one*1e3:9.1f} us/step frac {frac(one):.3f} "
                      f"({(one/two-1)*100:+.1f} %) | bit-identical {same(e1, e2)}", flush=True)
                continue
            two, info, d2, e2 = run(ic, kw, dt, steps, args.reps, one_launch=False, **extra)
            one, _, d1, e1 = run(ic, kw, dt, steps, args.reps, one_launch=True, **extra)
            assert "one_launch=0" in d2 and "one_launch=1" in d1, (d1, d2)
            print(f"round {rnd + 1} {name:10s} n={n:7d} tile={info['tile_particles']:4d} L={info['chunks_per_item']:3d} items={info['items']:5d} steps={steps:3d} | "
                  f"two launches {two*1e3:9.1f} us/step frac {frac(two):.3f} | one launch {one*1e3:9.1f} us/step frac {frac(one):.3f} "
                  f"({(one/two-1)*100:+.1f} %) | bit-identical {same(e1, e2)}", flush=True)


if __name__ == "__main__":
    main()
