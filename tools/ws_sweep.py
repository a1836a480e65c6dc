#!/usr/bin/env python3
"""Wave-split (512-particle tiles) against classic (2048) symmetric kernels across N, chunks per item and chunk pairs.

For each case: ms per step (wall, `steps` steps after a warm-up), mean force-launch time (HIP events), the algorithmic
roofline fraction (14 flop x N^2 / step time / 157.3 TF), and the accelerations' distance from the classic kernel's
(max |da| / global force scale) — the two kernels sum the same pairs in a different association.

    python tools/ws_sweep.py [--cases ref25000,p16384,p32768,p65536,p131072] [--steps 200]
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
        n = int(name[3:])
        return nb.default_ics(n), dict(eps=1.0, extras=3), 0.01
    n = int(name[1:])
    return nb.plummer_2d(n, 42), dict(eps=0.01), 1e-3


def run(ic, kw, dt, steps, **tune):
    with nb.Simulation(ic, **kw, **tune) as s:
        acc = s.accelerations()
        s.advance(10, dt)
        s.wait()
        s.profile(True)
        t0 = time.perf_counter()
        s.advance(steps, dt)
        s.wait()
        el = time.perf_counter() - t0
        ms, cnt = s.profile_read()
        info = s.sym_info()
        desc = s.describe()
    return el / steps * 1e3, ms / max(cnt, 1), info, acc, desc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="ref25000,p16384,p32768,p65536,p131072")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--Ls", default="0,4,8,12,16,24,32")
    ap.add_argument("--general", action="store_true", help="Plummer cases without the equal-mass specialisation")
    args = ap.parse_args()
    for name in args.cases.split(","):
        ic, kw, dt = workload(name)
        n = ic.shape[0]
        if args.general:
            kw["uniform_mass"] = False
        steps = max(20, min(args.steps, int(args.steps * (65536.0 / n) ** 2)))
        base_ms, base_k, info, acc0, desc = run(ic, kw, dt, steps, sym_tile=2048)
        scale = float(np.sqrt(np.mean(np.sum(acc0.astype(np.float64) ** 2, axis=1))))
        frac = lambda ms: 14.0 * n * n / (ms * 1e-3) / PEAK
        print(f"== {name} n={n} steps={steps} | {desc.split('|')[3].strip()}", flush=True)
        print(f"   classic tile=2048 L={info['chunks_per_item']:3d} items={info['items']:5d}  step {base_ms*1e3:8.1f} us  force {base_k*1e3:8.1f} us  frac {frac(base_ms):.3f}", flush=True)
        for pairs in (-1, 1):
            for L in [int(x) for x in args.Ls.split(",")]:
                if pairs == 1 and L and L % 8:
                    continue
                try:
                    ms, k, info, acc, _ = run(ic, kw, dt, steps, sym_tile=512, sym_chunks_per_item=L, sym_chunk_pairs=pairs)
                except nb.NBodyError as e:
                    print(f"   tile=512 L={L} pairs={pairs}: {e}")
                    continue
                err = float(np.max(np.linalg.norm(acc.astype(np.float64) - acc0, axis=1))) / scale
                print(f"   ws tile=512 pairs={pairs:2d} L={info['chunks_per_item']:3d}{'*' if L == 0 else ' '} items={info['items']:5d}  step {ms*1e3:8.1f} us  "
                      f"force {k*1e3:8.1f} us  frac {frac(ms):.3f}  ({(ms/base_ms-1)*100:+.1f} %)  |da|/scale {err:.2e}  slabs {info['slab_s_bytes']/2**20:.1f}+{info['slab_r_bytes']/2**20:.1f} MiB",
                      flush=True)


if __name__ == "__main__":
    main()
