"""oracle/nbo.py — TEST INFRASTRUCTURE ONLY.

ctypes loader for the CPU restatement (``libnb_oracle.so``) and, where it has
been built in this container, the compiled reference shim
(``_ref/libnbref.so``).  Importers allowed: ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``.
The product package never imports this module.

The restatement is rebuilt with ``-march=native`` whenever the host CPU differs
from the one that built the travelling ``.so`` (the GPU box has another CPU
than the build container).
"""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB = HERE / "libnb_oracle.so"
REF_LIB = HERE / "_ref" / "libnbref.so"
STAMP = HERE / ".oracle_build_id"

RSQRT_EXACT, RSQRT_QUAKE = 0, 1

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


def _cpu_id() -> str:
    try:
        txt = Path("/proc/cpuinfo").read_text()
        flags = next((l for l in txt.splitlines() if l.startswith("flags")), "")
        model = next((l for l in txt.splitlines() if l.startswith("model name")), "")
    except OSError:
        flags = model = ""
    src = (HERE / "nb_oracle.c").read_bytes() + (HERE / "nb_oracle.h").read_bytes()
    return hashlib.sha256(flags.encode() + model.encode() + src).hexdigest()[:16]


def build(force: bool = False) -> Path:
    """(Re)build libnb_oracle.so for this host if needed."""
    want = _cpu_id()
    have = STAMP.read_text().strip() if STAMP.exists() else ""
    if force or not LIB.exists() or have != want:
        try:
            subprocess.run(["make", "-C", str(HERE), "-B", "libnb_oracle.so"], check=True, capture_output=True, text=True)
            STAMP.write_text(want)
        except (subprocess.CalledProcessError, FileNotFoundError) as e:  # keep a travelling .so if there is one
            if not LIB.exists():
                raise RuntimeError(f"cannot build the oracle: {getattr(e, 'stderr', e)}")
    return LIB


def host_threads() -> int:
    """CPU threads this process may really use: affinity mask capped by the cgroup CPU quota
    (the GPU box shows 256 logical CPUs but grants a share of them)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = Path(path).read_text().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    period = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
                    n = min(n, max(1, q // period))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        l = C.CDLL(str(LIB))
        l.nbo_fast_inv_sqrt.restype = C.c_float
        l.nbo_fast_inv_sqrt.argtypes = [C.c_float]
        l.nbo_set_threads.restype = C.c_int
        l.nbo_set_threads.argtypes = [C.c_int]
        l.nbo_get_threads.restype = C.c_int
        l.nbo_accel_f32.argtypes = [C.c_size_t, _f32p, _f32p, _f32p, C.c_float, C.c_int, C.c_size_t, C.c_size_t, _f32p, _f32p]
        l.nbo_accel_f64.argtypes = [C.c_size_t, _f64p, _f64p, _f64p, C.c_double, C.c_size_t, C.c_size_t, _f64p, _f64p]
        l.nbo_accel_f32_terms_acc64.argtypes = [C.c_size_t, _f32p, _f32p, _f32p, C.c_float, C.c_int, C.c_size_t, C.c_size_t, _f64p, _f64p]
        l.nbo_step_f32.argtypes = [C.c_size_t, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int]
        l.nbo_step_f64.argtypes = [C.c_size_t, _f64p, _f64p, _f64p, _f64p, _f64p, _f64p, _f64p, C.c_double, C.c_double, C.c_int]
        l.nbo_kick_drift_f32.argtypes = [C.c_size_t, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_float, C.c_int]
        l.nbo_energy_f64.argtypes = [C.c_size_t, _f64p, _f64p, _f64p, _f64p, _f64p, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        l.nbo_accel3_f64.argtypes = [C.c_size_t, _f64p, _f64p, _f64p, _f64p, C.c_double, _f64p, _f64p, _f64p]
        l.nbo_step3_f64.argtypes = [C.c_size_t] + [_f64p] * 10 + [C.c_double, C.c_double, C.c_int]
        l.nbo_energy3_f64.argtypes = [C.c_size_t] + [_f64p] * 7 + [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        l.nbo_set_threads(int(os.environ.get("NBO_THREADS", host_threads())))
        _lib = l
    return _lib


# ---------------------------------------------------------------------------
# State = dict of contiguous SoA arrays x, y, vx, vy, ax, ay, m (+ r carried)
# ---------------------------------------------------------------------------
FIELDS = ("x", "y", "vx", "vy", "ax", "ay", "m", "r")


def state_from_flat(flat: np.ndarray, dtype=np.float32) -> dict:
    """flat: (n, 8) array of x,y,vx,vy,ax,ay,m,r."""
    return {k: np.ascontiguousarray(flat[:, i], dtype=dtype) for i, k in enumerate(FIELDS)}


def state_to_flat(st: dict) -> np.ndarray:
    return np.stack([np.asarray(st[k]) for k in FIELDS], axis=1)


def state_from_bodies(bodies: np.ndarray, dtype=np.float32) -> dict:
    """bodies: structured array with pos, vel, acc, mass, radius (64-byte records)."""
    return {
        "x": np.ascontiguousarray(bodies["pos"][:, 0], dtype=dtype),
        "y": np.ascontiguousarray(bodies["pos"][:, 1], dtype=dtype),
        "vx": np.ascontiguousarray(bodies["vel"][:, 0], dtype=dtype),
        "vy": np.ascontiguousarray(bodies["vel"][:, 1], dtype=dtype),
        "ax": np.ascontiguousarray(bodies["acc"][:, 0], dtype=dtype),
        "ay": np.ascontiguousarray(bodies["acc"][:, 1], dtype=dtype),
        "m": np.ascontiguousarray(bodies["mass"], dtype=dtype),
        "r": np.ascontiguousarray(bodies["radius"], dtype=dtype),
    }


def fast_inv_sqrt(x: np.ndarray) -> np.ndarray:
    l = lib()
    return np.array([l.nbo_fast_inv_sqrt(float(v)) for v in np.asarray(x, dtype=np.float32)], dtype=np.float32)


def accel_f32(st: dict, eps: float, rsqrt: int, i_begin: int = 0, i_end: int | None = None):
    n = st["x"].shape[0]
    i_end = n if i_end is None else i_end
    ax = np.zeros(n, np.float32)
    ay = np.zeros(n, np.float32)
    eps2 = np.float32(eps) * np.float32(eps)
    lib().nbo_accel_f32(n, st["x"], st["y"], st["m"], eps2, rsqrt, i_begin, i_end, ax, ay)
    return ax, ay


def accel_f32_terms_acc64(st: dict, eps: float, rsqrt: int, i_begin: int = 0, i_end: int | None = None):
    """The reference's fp32 per-pair terms (Quadtree.hpp:136-143) summed in double: its arithmetic without the
    rounding of its single fp32 running sum."""
    n = st["x"].shape[0]
    i_end = n if i_end is None else i_end
    ax = np.zeros(n, np.float64)
    ay = np.zeros(n, np.float64)
    eps2 = np.float32(eps) * np.float32(eps)
    lib().nbo_accel_f32_terms_acc64(n, st["x"], st["y"], st["m"], eps2, rsqrt, i_begin, i_end, ax, ay)
    return ax, ay


def accel_f64(st: dict, eps: float, i_begin: int = 0, i_end: int | None = None):
    n = st["x"].shape[0]
    i_end = n if i_end is None else i_end
    ax = np.zeros(n, np.float64)
    ay = np.zeros(n, np.float64)
    lib().nbo_accel_f64(n, st["x"], st["y"], st["m"], float(eps) * float(eps), i_begin, i_end, ax, ay)
    return ax, ay


def step_f32(st: dict, eps: float, dt: float, nsteps: int, rsqrt: int, extras: int = 0) -> dict:
    """In place on a float32 state; returns it."""
    n = st["x"].shape[0]
    eps2 = np.float32(eps) * np.float32(eps)
    lib().nbo_step_f32(n, st["x"], st["y"], st["vx"], st["vy"], st["m"], st["ax"], st["ay"], eps2, dt, nsteps, rsqrt, extras)
    return st


def step_f64(st: dict, eps: float, dt: float, nsteps: int) -> dict:
    n = st["x"].shape[0]
    lib().nbo_step_f64(n, st["x"], st["y"], st["vx"], st["vy"], st["m"], st["ax"], st["ay"], float(eps) ** 2, dt, nsteps)
    return st


def energy(st: dict, eps: float):
    d = {k: np.ascontiguousarray(st[k], dtype=np.float64) for k in ("x", "y", "vx", "vy", "m")}
    k, u = C.c_double(), C.c_double()
    lib().nbo_energy_f64(d["x"].shape[0], d["x"], d["y"], d["vx"], d["vy"], d["m"], float(eps) ** 2, C.byref(k), C.byref(u))
    return k.value, u.value


# ---- 3-D extension (fp64 only; no reference to pin to) ---------------------------
def state3_from_bodies(bodies3: np.ndarray) -> dict:
    """bodies3: structured array viewed with 3-component pos / vel / acc (nbodysim_amd.BODY3_DTYPE)."""
    d = {}
    for k, f, c in (("x", "pos", 0), ("y", "pos", 1), ("z", "pos", 2), ("vx", "vel", 0), ("vy", "vel", 1), ("vz", "vel", 2)):
        d[k] = np.ascontiguousarray(bodies3[f][:, c], dtype=np.float64)
    d["m"] = np.ascontiguousarray(bodies3["mass"], dtype=np.float64)
    for k in ("ax", "ay", "az"):
        d[k] = np.zeros_like(d["x"])
    return d


def accel3_f64(st: dict, eps: float):
    n = st["x"].shape[0]
    lib().nbo_accel3_f64(n, st["x"], st["y"], st["z"], st["m"], float(eps) ** 2, st["ax"], st["ay"], st["az"])
    return st["ax"], st["ay"], st["az"]


def step3_f64(st: dict, eps: float, dt: float, nsteps: int) -> dict:
    n = st["x"].shape[0]
    lib().nbo_step3_f64(n, st["x"], st["y"], st["z"], st["vx"], st["vy"], st["vz"], st["m"], st["ax"], st["ay"], st["az"],
                        float(eps) ** 2, dt, nsteps)
    return st


def energy3(st: dict, eps: float):
    k, u = C.c_double(), C.c_double()
    lib().nbo_energy3_f64(st["x"].shape[0], st["x"], st["y"], st["z"], st["vx"], st["vy"], st["vz"], st["m"], float(eps) ** 2,
                          C.byref(k), C.byref(u))
    return k.value, u.value


def set_threads(n: int) -> int:
    """n <= 0 restores the default (host_threads())."""
    return lib().nbo_set_threads(n if n > 0 else int(os.environ.get("NBO_THREADS", host_threads())))


# ---------------------------------------------------------------------------
# compiled reference (only where oracle/_ref/libnbref.so exists)
# ---------------------------------------------------------------------------
_ref = None


def have_ref() -> bool:
    return REF_LIB.exists()


def ref() -> C.CDLL:
    global _ref
    if _ref is None:
        if not have_ref():
            raise RuntimeError("oracle/_ref/libnbref.so not built (needs /root/reference: `make -C oracle ref`)")
        l = C.CDLL(str(REF_LIB))
        l.ref_layout.argtypes = [C.POINTER(C.c_size_t)]
        l.ref_fast_inv_sqrt.argtypes = [_f32p, _f32p, C.c_size_t]
        l.ref_direct_acc.argtypes = [_f32p, C.c_size_t, C.c_float]
        l.ref_direct_step.argtypes = [_f32p, C.c_size_t, C.c_float, C.c_float, C.c_int, C.c_int]
        if hasattr(l, "ref_direct_acc_timed"):
            l.ref_direct_acc_timed.restype = C.c_double
            l.ref_direct_acc_timed.argtypes = [_f32p, C.c_size_t, C.c_float, C.c_size_t, C.c_size_t, C.c_int]
        l.ref_step.restype = C.c_size_t
        l.ref_step.argtypes = [_f32p, C.c_size_t, C.c_float, C.c_float, C.c_int]
        l.ref_default_ics.restype = C.c_size_t
        l.ref_default_ics.argtypes = [_f32p, C.c_size_t]
        _ref = l
    return _ref


def ref_layout() -> dict:
    out = (C.c_size_t * 8)()
    ref().ref_layout(out)
    keys = ["sizeof_Body", "alignof_Body", "off_pos", "off_vel", "off_acc", "off_mass", "off_radius", "sizeof_Vec2"]
    return {k: int(v) for k, v in zip(keys, out)}


def ref_direct_acc_timed(flat: np.ndarray, eps: float, i_begin: int, i_end: int, threads: int):
    """Seconds the compiled reference's own pairwise loop takes for i in [i_begin, i_end) against all bodies
    (std::async fan-out like Simulation::attract); also returns the flat records with acc filled for that slice."""
    out = np.ascontiguousarray(flat, dtype=np.float32).copy()
    secs = ref().ref_direct_acc_timed(out.reshape(-1), out.shape[0], float(eps), int(i_begin), int(i_end), int(threads))
    return float(secs), out
