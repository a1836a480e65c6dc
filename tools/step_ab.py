#!/usr/bin/env python3
"""A/B of ONE switch of `nbodysim_amd.Simulation` on one box: the same workloads with the switch off and on, in interleaved
rounds (the clock a part holds drifts within a call), with a bit comparison of the final bodies where the switch promises
the same bits.

    python tools/step_ab.py --what dynamic_items                     # dynamic against static work items (same bits)
    python tools/step_ab.py --what mass_scaling --general            # 12 + 2 body against 11 + 2 (different rounding: not the same bits)
    python tools/step_ab.py [--cases ref25000,p16384,...] [--rounds 2] [--extra sym_tile=512,...]
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


#: switch -> (label A, keywords A, label B, keywords B, do the two sides promise the same bits?)
SWITCHES = {
    "dynamic_items": ("static items", dict(static_items=True), "dynamic items", dict(static_items=False), True),
    "mass_scaling": ("12 + 2 body", dict(mass_scaling=False), "11 + 2 body (masses folded)", dict(mass_scaling=True), False),
    "guided_tail": ("uniform items", dict(guided_tail=False), "guided tail", dict(guided_tail=True), True),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="p9216,p16384,ref25000,p32768,p49152,p65536,p131072,p262144")
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--general", action="store_true")
    ap.add_argument("--what", default="dynamic_items", choices=sorted(SWITCHES), help="the switch under test")
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
            label_a, kw_a, label_b, kw_b, same_bits = SWITCHES[args.what]
            ta, info, _, ea = run(ic, kw, dt, steps, args.reps, **kw_a, **extra)
            tb, _, desc, eb = run(ic, kw, dt, steps, args.reps, **kw_b, **extra)
            print(f"round {rnd + 1} {name:10s} n={n:7d} tile={info['tile_particles']:4d} L={info['chunks_per_item']:3d} items={info['items']:5d} steps={steps:3d} | "
                  f"{label_a} {ta*1e3:9.1f} us/step frac {frac(ta):.3f} | {label_b} {tb*1e3:9.1f} us/step frac {frac(tb):.3f} "
                  f"({(tb/ta-1)*100:+.1f} %) | " + (f"bit-identical {same(ea, eb)}" if same_bits else "(different rounding by design)"), flush=True)

if __name__ == "__main__":
    main()
