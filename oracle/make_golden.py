#!/usr/bin/env python3
"""oracle/make_golden.py — TEST INFRASTRUCTURE: regenerate tests/golden/*.

Runs ONLY in the build container (needs /root/reference to have been compiled
into oracle/_ref/libnbref.so by `make -C oracle ref`).  The outputs are data —
inputs and the reference's outputs — never reference source.

Fixtures (little-endian float32 .npy, columns x,y,vx,vy,ax,ay,mass,radius):
  ic_plummer_1024.npy        ICs: product generator nb_plummer_2d(1024, seed 42)
  ic_plummer_4096.npy        ICs: nb_plummer_2d(4096, seed 7)
  ref_direct_s{1,10,100}.npy REF-DIRECT states (reference Quadtree::acc leaf loop as a
                             direct sum + kick/drift), eps 0.05, dt 1e-3
  ref_step_s{1,10,100}.npy   REF-STEP states (production Simulation::step(), BH theta=1)
  ic_extras_512.npy, ref_step_extras_s{1,4}.npy   massless far-out fast bodies through the real step():
                             pins the velocity clamp and soft boundary of iterate() (Simulation.hpp:133-155)
  ref_direct_acc_{1024,4096}.npy   single REF-DIRECT force evaluation (n,2)
  ref_direct_acc_eps1_1024.npy     same with the reference's default eps = 1
  fast_inv_sqrt_x.npy / _y.npy     Quadtree::fast_inv_sqrt on a fixed 4096-point grid
  layout.json                sizeof/alignof/offsetof of the compiled Body / Vec2
  default_ics.json           digest + first/last bodies of Simulation()'s own 25 000-body ICs
  default_ics_first4096.npy  the innermost 4096 of those bodies (1e9 central mass, |v| up to 1426 > the 1000 clamp)
  manifest.json              build flags, compiler, sha256 of every fixture
"""
from __future__ import annotations

import hashlib
import json
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))

import nbo  # noqa: E402
from nbodysim_amd import _lib as L  # noqa: E402  (host-only generator; no GPU needed)

GOLD = ROOT / "tests" / "golden"
EPS, DT = 0.05, 1e-3


def flat_from_bodies(b: np.ndarray) -> np.ndarray:
    f = np.zeros((b.shape[0], 8), np.float32)
    f[:, 0:2], f[:, 2:4], f[:, 4:6] = b["pos"], b["vel"], b["acc"]
    f[:, 6], f[:, 7] = b["mass"], b["radius"]
    return f


def main() -> None:
    GOLD.mkdir(parents=True, exist_ok=True)
    ref = nbo.ref()
    out = {}

    ic1k = flat_from_bodies(L.plummer_2d(1024, 42))
    ic4k = flat_from_bodies(L.plummer_2d(4096, 7))
    out["ic_plummer_1024.npy"] = ic1k
    out["ic_plummer_4096.npy"] = ic4k

    for steps in (1, 10, 100):
        a = np.ascontiguousarray(ic1k.copy())
        ref.ref_direct_step(a, a.shape[0], EPS, DT, steps, 0)
        b = np.ascontiguousarray(ic1k.copy())
        ref.ref_direct_step(b, b.shape[0], EPS, DT, steps, 1)  # through Body::update (Body.hpp:34-38)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "iterate() lines and Body::update disagree"
        out[f"ref_direct_s{steps}.npy"] = a
        c = np.ascontiguousarray(ic1k.copy())
        frame = ref.ref_step(c, c.shape[0], EPS, DT, steps)
        assert frame == steps
        out[f"ref_step_s{steps}.npy"] = c
    # determinism of the reference build (two runs bit-identical)
    again = np.ascontiguousarray(ic1k.copy())
    ref.ref_direct_step(again, again.shape[0], EPS, DT, 100, 0)
    assert np.array_equal(again.view(np.uint32), out["ref_direct_s100.npy"].view(np.uint32))

    for name, ic, eps in (("ref_direct_acc_1024.npy", ic1k, EPS), ("ref_direct_acc_4096.npy", ic4k, EPS),
                          ("ref_direct_acc_eps1_1024.npy", ic1k, 1.0)):
        a = np.ascontiguousarray(ic.copy())
        ref.ref_direct_acc(a, a.shape[0], eps)
        out[name] = np.ascontiguousarray(a[:, 4:6])

    # The non-gravity parts of Simulation::iterate (velocity clamp :133-137, soft boundary :140-155)
    # pinned against the REAL step(): with every mass 0 the tree force is exactly 0 and radius 0 keeps
    # collide() inert, so step() = kick(0) + clamp + boundary + drift.
    rng = np.random.default_rng(20241223)
    ex = np.zeros((512, 8), np.float32)
    ang, rad = rng.uniform(0, 2 * np.pi, 512), rng.uniform(5e4, 1.4e5, 512)
    ex[:, 0], ex[:, 1] = rad * np.cos(ang), rad * np.sin(ang)
    ex[:, 2:4] = rng.normal(size=(512, 2)) * 900.0
    out["ic_extras_512.npy"] = ex.copy()
    for steps in (1, 4):
        c = np.ascontiguousarray(ex.copy())
        assert ref.ref_step(c, c.shape[0], 1.0, 0.01, steps) == steps
        assert not c[:, 4:6].any()          # accelerations are exactly zero
        out[f"ref_step_extras_s{steps}.npy"] = c

    x = np.concatenate([
        np.logspace(-12, 12, 2048).astype(np.float32),
        np.linspace(1.0, 4.0, 2040, endpoint=False).astype(np.float32),
        np.array([0.0, 1e-38, 1e-45, 3.4e38, 1.0, 2.0, 4.0, 0.0025], np.float32),
    ])
    y = np.zeros_like(x)
    ref.ref_fast_inv_sqrt(np.ascontiguousarray(x), y, x.shape[0])
    out["fast_inv_sqrt_x.npy"] = x
    out["fast_inv_sqrt_y.npy"] = y

    for name, arr in out.items():
        np.save(GOLD / name, arr)

    layout = nbo.ref_layout()
    (GOLD / "layout.json").write_text(json.dumps(layout, indent=1) + "\n")

    ics = np.zeros((25000, 8), np.float32)
    n = ref.ref_default_ics(ics, ics.shape[0])
    assert n == 25000
    defaults = {
        "n": int(n),
        "sha256_float32_le": hashlib.sha256(ics.tobytes()).hexdigest(),
        "body0": [float(v) for v in ics[0]],
        "body1": [float(v) for v in ics[1]],
        "body_last": [float(v) for v in ics[-1]],
        "total_mass": float(ics[:, 6].astype(np.float64).sum()),
        "note": "libstdc++-specific (std::uniform_real_distribution), SURVEY.md §8c",
    }
    (GOLD / "default_ics.json").write_text(json.dumps(defaults, indent=1) + "\n")
    # the innermost 4096 bodies of the reference's own demo ICs (sorted by radius, Simulation.hpp:585-589;
    # body 0 is the 1e9 central mass): data for running the reference's default workload through the GPU path
    np.save(GOLD / "default_ics_first4096.npy", np.ascontiguousarray(ics[:4096]))
    out["default_ics_first4096.npy"] = ics[:4096]

    gxx = subprocess.run(["g++", "--version"], capture_output=True, text=True).stdout.splitlines()[0]
    manifest = {
        "generator": "oracle/make_golden.py",
        "reference": "7IBBE77S/nbodysim @ /root/reference (headers included by path, unmodified)",
        "reference_build": "g++ -std=c++20 -O3 -msse4.1 -ffp-contract=off -include bit -include cstdint -I<ref>/Nbodysim/headers",
        "compiler": gxx,
        "params": {"eps": EPS, "dt": DT, "n": 1024, "seed": 42, "steps": [1, 10, 100]},
        "columns": ["x", "y", "vx", "vy", "ax", "ay", "mass", "radius"],
        "sha256": {name: hashlib.sha256((GOLD / name).read_bytes()).hexdigest() for name in sorted(out)},
    }
    (GOLD / "manifest.json").write_text(json.dumps(manifest, indent=1) + "\n")
    print("wrote", len(out) + 3, "fixtures to", GOLD)


if __name__ == "__main__":
    main()
