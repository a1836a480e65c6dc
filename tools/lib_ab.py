#!/usr/bin/env python3
"""A previous build of the library against the current one, alternating, on one box: the same workloads stepped through both,
final bodies compared bit for bit (sha256 of pos | vel | acc) and timed.  For changes that must not alter a single sum.

    git worktree add /tmp/old <commit> && make -C /tmp/old/nbodysim_amd/csrc && cp /tmp/old/nbodysim_amd/libnbody_hip.so build/old_lib/
    python tools/lib_ab.py --old build/old_lib/libnbody_hip.so [--cases p9216,p16384,ref25000,...]

(the Python binding loads $NBODY_HIP_LIB when set — the LIBRARY reads no environment variables; each side runs in a child process)
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def child(cases, steps):
    import nbodysim_amd as nb
    out = {}
    for name in cases:
        if name.startswith("ref"):
            ic, kw, dt = nb.default_ics(int(name[3:])), dict(eps=1.0, extras=3), 0.01
        elif name.startswith("q"):                   # 3-D
            ic, kw, dt = nb.plummer_3d(int(name[1:]), 42), dict(eps=0.01, dims=3), 1e-3
        elif name.startswith("d"):                   # fp64
            ic, kw, dt = nb.plummer_2d(int(name[1:]), 42), dict(eps=0.01, precision="fp64"), 1e-3
        else:
            ic, kw, dt = nb.plummer_2d(int(name[1:]), 42), dict(eps=0.01), 1e-3
        for general in (False, True):
            k = dict(kw)
            if general:
                k["uniform_mass"] = False
            with nb.Simulation(ic, **k) as s:
                s.advance(30, dt)
                s.wait()
                t0 = time.perf_counter()
                s.advance(steps, dt)
                s.wait()
                el = (time.perf_counter() - t0) / steps
                b = s.sync()
                h = hashlib.sha256(b"".join(np.ascontiguousarray(b[f]).tobytes() for f in ("pos", "vel", "acc"))).hexdigest()
            out[f"{name}/{'individual' if general else 'equal'} masses"] = (h, el * 1e6)
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--old", default=str(ROOT / "build" / "old_lib" / "libnbody_hip.so"))
    ap.add_argument("--cases", default="p9216,p16384,ref25000,p32768,p65536,p131072")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--child", action="store_true")
    args = ap.parse_args()
    cases = args.cases.split(",")
    if args.child:
        child(cases, args.steps)
        return
    res = {}
    for tag, lib in (("old", args.old), ("new", None), ("old2", args.old), ("new2", None)):
        env = dict(os.environ)
        env.pop("NBODY_HIP_LIB", None)
        if lib:
            env["NBODY_HIP_LIB"] = str(Path(lib).resolve())
        r = subprocess.run([sys.executable, __file__, "--child", "--cases", args.cases, "--steps", str(args.steps)], env=env, capture_output=True, text=True, timeout=600)
        if r.returncode:
            print(tag, r.stderr[-2000:])
            sys.exit(1)
        res[tag] = json.loads(r.stdout.strip().splitlines()[-1])
    for k in res["old"]:
        o, n = res["old"][k][1] + res["old2"][k][1], res["new"][k][1] + res["new2"][k][1]
        print(f"{k:28s} same bits {res['old'][k][0] == res['new'][k][0]} | old {res['old'][k][1]:8.1f} {res['old2'][k][1]:8.1f} us/step | "
              f"new {res['new'][k][1]:8.1f} {res['new2'][k][1]:8.1f} us/step ({(n / o - 1) * 100:+.1f} %)")


if __name__ == "__main__":
    main()
