#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path (BASELINE.json):
pair interactions/s of the direct O(N^2) softened-gravity + kick/drift step at
N = 262 144 (fp32, exact rsqrt, Plummer-2D synthetic data), 1/2/4/8 GPUs.

    python bench.py --gpus 1 --steps K --warmup W
    python bench.py --gpus N --steps K --warmup W          # typed as is: see "Launching" below
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Launching.  One process per GPU either way.  Under torch.distributed.run (WORLD_SIZE set) this process IS a rank.
Typed without a launcher, `--gpus N` (N > 1) makes this process the launcher: before it imports torch or touches a
GPU it starts `python -m torch.distributed.run ... bench.py <same arguments>` as a CHILD process (never an exec),
relays rank 0's one JSON line and the ranks' stderr, forwards SIGTERM / SIGINT, and returns the children's status.
Without torch.distributed.run (or with --one-process) ONE process drives N sharded handles, one per device, through
the library's own RCCL loop (nb_comm_create_all / nb_comm_step), with the same self-check and line.

A "step" = one force evaluation over all N^2 pairs (self pairs included, as the
kernel evaluates them) + kick + drift.  N is FIXED as GPUs are added (strong
scaling, the north-star target): each rank integrates N/gpus particles and the
ranks all-gather their (x,y) blocks every step (RCCL), overlapped with the
local-tile force.  Prints ONE JSON line on rank 0.

Extra objects in the line:
  roofline      fp32 vector-ALU roofline of the force kernel.  `achieved` / `frac`
                are ALGORITHMIC (the contract's definition): 14 flop per ORDERED
                pair x N^2 pairs per launch / mean launch duration
                measured with HIP events on the kernel's stream inside the timed
                region; peak 157.3 TFLOP/s (MI355X_MICROARCH.md).  The symmetric
                kernel evaluates every UNORDERED pair once, so the flops it really
                issues are fewer: `executed_tflops` / `executed_frac` count those
                (17 flop per unordered pair) and are the honest "fraction of the
                ALU peak" figure.  `frac_sustained` is the same fraction over the >= 2 s of steps run AFTER
                the timed region: the settled figure, the one to quote.
                `valu_busy` and `traffic` (HBM bytes of one launch of the dominant kernel) are COUNTER
                figures: by default this run collects them itself on this box, after everything that is
                timed — three separate `rocprofv3 --pmc` passes of this command in its shortest form as
                child processes (FETCH_SIZE x2 correction; MI355X_MICROARCH.md) — and says so (`pmc_status:
                "live"`; the committed record of profiles/hbm_traffic.json is carried beside them as
                `valu_busy_committed` / `traffic_committed`).  With --no-live-pmc, or where the profiler
                cannot run, the committed record is used if it is of exactly this kernel instantiation, N
                and work plan; otherwise the fields are null.  The work plan's own estimate (slab rows
                written + positions read) is `traffic_plan`, never in the counter's place.
  cpu_baseline  the oracle (C restatement of the reference's pairwise loop, the
                reference's own arithmetic) timed on this box's host cores over
                a bounded i-slice of the same workload.  Rank 0, --gpus 1 only.
  phases_ms     (N > 1) mean per-step time of each phase of the sharded step on the
                compute stream (local pairs | all-gather wait | cross pairs |
                reduce-scatter | kick+drift), so a scaling loss is attributable.
"""
from __future__ import annotations

import argparse
import json
import os
import signal
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

N_DEFAULT = 262_144
EPS, DT, SEED = 0.01, 1e-3, 42
FLOP_PER_PAIR = 14.0          # algorithmic flop per pair (2-D)
PEAK_FP32_TFLOPS = 157.3      # MI355X fp32 vector peak (MI355X_MICROARCH.md, chip table)
BYTES_PER_PARTICLE_STEP = 36  # read x,y,m,vx,vy + write x,y,vx,vy (fp32)


# flops the kernels really issue: per UNORDERED pair of the symmetric body (both directions) and per
# ORDERED pair of the one-sided body / of the symmetric kernel's diagonal items; [uniform-mass, general]
EXECUTED = {
    ("fp32", 2): {"sym": (17.0, 19.0), "one": (13.0, 14.0)},   # 2 pk_add 2 pk_fma 2 rsq 2 pk_mul (+2 pk_mul) 4|2 pk_fma per 2 pairs
    ("fp32", 3): {"sym": (24.0, 26.0), "one": (18.0, 19.0)},
    ("fp64", 2): {"sym": (24.0, 26.0), "one": (20.0, 21.0)},   # v_rsq_f64 + 6-op cube correction (rsqrt3_f64) = 10 flop
    ("fp64", 3): {"sym": (31.0, 33.0), "one": (25.0, 26.0)},
}


def host_cpu_info():
    """What "host cores of the GPU box" means for this process: logical CPUs, affinity mask, cgroup quota, model."""
    info = {"nproc": os.cpu_count()}
    try:
        info["affinity"] = len(os.sched_getaffinity(0))
    except AttributeError:
        info["affinity"] = None
    try:
        q = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        info["cgroup_cpus"] = None if q[0] == "max" else round(int(q[0]) / int(q[1]), 2)
    except (OSError, ValueError, IndexError):
        info["cgroup_cpus"] = None
    try:
        info["model"] = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except Exception:
        info["model"] = ""
    return info


class DeviceSampler:
    """Clock and socket power of the busiest visible GPU during the timed region, read from the amdgpu hwmon files
    (no privileges needed): the kernel is power-limited, so the step time of a given box follows
    the clock it can hold under its 1.4 kW cap — this records it next to the number it explains."""

    def __init__(self, period_s=0.02):
        import glob
        import threading
        self.files = [(f, f.replace("power1_input", "freq1_input"), f.replace("power1_input", "power1_cap"))
                      for f in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"))]
        self.period, self.samples, self._stop = period_s, [], threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True)

    @staticmethod
    def _read(path):
        try:
            return float(open(path).read())
        except (OSError, ValueError):
            return None

    def _run(self):
        while not self._stop.is_set():
            best = None
            for pw, fq, _ in self.files:
                w = self._read(pw)
                if w is not None and (best is None or w > best[0]):
                    best = (w, self._read(fq), self._read(pw.replace("power1_input", "temp2_input")), time.perf_counter())
            if best:
                self.samples.append(best)
            self._stop.wait(self.period)

    def start(self):
        if self.files:
            self._thread.start()

    def stop(self):
        self._stop.set()
        if self._thread.is_alive():
            self._thread.join(timeout=1.0)
        if not self.samples:
            return None
        pw = [s[0] * 1e-6 for s in self.samples]
        fq = [s[1] * 1e-6 for s in self.samples if s[1]]
        cap = max((self._read(c) or 0.0) for _, _, c in self.files) * 1e-6
        tj = [s[2] * 1e-3 for s in self.samples if len(s) > 2 and s[2]]
        k = max(1, len(self.samples) // 8)          # first / last eighth of the region: burst vs settled
        edge = lambda part: {"power_w": sum(x[0] for x in part) * 1e-6 / len(part),
                             "sclk_mhz": (sum(x[1] for x in part if x[1]) * 1e-6 / max(1, sum(1 for x in part if x[1]))) or None}
        out = {"samples": len(pw), "power_w_mean": sum(pw) / len(pw), "power_w_max": max(pw), "power_cap_w": cap or None,
               "sclk_mhz_mean": (sum(fq) / len(fq)) if fq else None, "sclk_mhz_min": min(fq) if fq else None,
               "junction_temp_c_max": max(tj) if tj else None,
               "at_start": edge(self.samples[:k]), "at_end": edge(self.samples[-k:]),
               "source": "amdgpu hwmon power1_input / freq1_input sampled during the region (rank 0's view; busiest visible card)"}
        # what limited the clock, from the samples themselves
        if cap and fq:
            near_cap = out["power_w_max"] >= 0.93 * cap
            out["clock_limit"] = ("socket power reached its cap during the region: power-limited" if near_cap else
                                  f"socket power peaked at {out['power_w_max']:.0f} W of a {cap:.0f} W cap with the clock at {out['sclk_mhz_mean']:.0f} MHz "
                                  "mean: the cap was not what held the clock on this box (these samples do not say what did)")
        return out


def sustained_steps(ms_per_step_hint, min_seconds=2.0):
    """Steps of the sustained stretch for a step time of `ms_per_step_hint`: a pure function — every rank must feed it the SAME hint."""
    return max(20, int(min_seconds * 1e3 / max(ms_per_step_hint, 1e-3)) + 1)


def reduce_max_over_ranks(value, world):
    """MAX of a host float over the ranks of the default process group (the identity for one rank)."""
    try:
        import torch.distributed as dist
    except ImportError:
        return float(value)
    if not (dist.is_available() and dist.is_initialized()):     # one rank, or one process driving every rank (no process group)
        return float(value)
    import torch
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def sustained_rate(advance, wait, dt, ms_per_step_hint, min_seconds=2.0, sampler_period=0.02, world=1, reduce_max=reduce_max_over_ranks):
    """The settled rate: at least `min_seconds` of back-to-back steps AFTER the driver's timed region (whose 20 steps
    sit before the power controller has settled), with the device clock / power of exactly that stretch.  Single GPU:
    if the first batch — sized from the timed region's rate — ends early, further batches sized from the rate measured
    so far follow until the stretch is long enough.
    SEVERAL RANKS: every step is a set of collectives, so every rank must enqueue the SAME number of steps.  Nothing
    rank-local may decide that number: the hint is MAX-reduced over the ranks HERE (whatever the caller passed), there
    is exactly one batch, and no measured time feeds back into a step count (round 3 lost a GPU call to exactly that:
    step counts sized from each rank's own clock -> mismatched collective counts -> deadlock)."""
    adaptive = world <= 1
    steps = sustained_steps(reduce_max(ms_per_step_hint, world), min_seconds)
    smp = DeviceSampler(period_s=sampler_period)
    smp.start()
    t0 = time.perf_counter()
    done = 0
    for _ in range(6):
        advance(steps, dt)
        wait()
        done += steps
        el = time.perf_counter() - t0
        if not adaptive or el >= min_seconds:
            break
        steps = max(20, int((min_seconds - el) * 1.1 * done / el) + 1)
    el = time.perf_counter() - t0
    return {"steps": done, "seconds": el, "ms_per_step": el / done * 1e3, "device_state": smp.stop()}


def frame_costs(make, ms_per_step_hint, dt):
    """What the reference's caller pays per FRAME through the drop-in (main.cpp:621-627: `step(); lock; copy bodies`), beside the
    resident step rate `value` is quoted on — host buffers included, so never part of `value` (the contract: PCIe-inclusive rates
    are noted, not reported).  `make()` -> a fresh handle of the benchmarked system.  Milliseconds per frame of
      step            nb_step(1) + blocking nb_sync into the caller's pageable array (Simulation::step(), Simulation.hpp:67-75)
      step_copy       the same + the caller's own copy of `bodies` (SHARED_BODIES = simulation->bodies)
      overlapped_copy nb_snapshot_begin / _wait: frame k's transfer behind step k + 1, one frame late (Simulation::step_overlapped)
      resident        nb_step(frames) + nb_wait: nothing leaves the device."""
    import ctypes
    frames = int(min(300, max(20, 400.0 / max(ms_per_step_hint, 1e-3))))
    out = {"frames": frames}

    def copy(dst, src):          # the caller's `SHARED_BODIES = simulation->bodies`: one memcpy of the records (numpy would copy field by field)
        ctypes.memmove(dst.ctypes.data, src.ctypes.data, src.nbytes)
    with make() as g:
        shared = g.bodies.copy()
        g.advance(3, dt)
        g.sync()
        t0 = time.perf_counter()
        for _ in range(frames):
            g.step(dt)
        out["step"] = (time.perf_counter() - t0) / frames * 1e3
        t0 = time.perf_counter()
        for _ in range(frames):
            g.step(dt)
            copy(shared, g.bodies)
        out["step_copy"] = (time.perf_counter() - t0) / frames * 1e3
        back, inflight = g.bodies.copy(), False
        t0 = time.perf_counter()
        for _ in range(frames):
            g.advance(1, dt)
            if inflight:
                g.snapshot_wait()
                copy(shared, back)
            g.snapshot_begin(back)
            inflight = True
        g.snapshot_wait()
        out["overlapped_copy"] = (time.perf_counter() - t0) / frames * 1e3
        g.wait()
        t0 = time.perf_counter()
        g.advance(frames, dt)
        g.wait()
        out["resident"] = (time.perf_counter() - t0) / frames * 1e3
        out["bytes_per_frame"] = int(g.bodies.nbytes)
    out["note"] = ("ms per frame of the reference caller's loop (main.cpp:621-627) through the C ABI, pageable host arrays; PCIe-inclusive, "
                   "never part of `value`; the C++ adaptor's own figures: build/sim_thread_example frames (INTEGRATION.md §2)")
    return out


def kernel_instantiation(desc, precision, dims, rsqrt):
    """The exact template instantiation of the dominant force kernel as rocprofv3 prints it, from nb_describe's fields —
    what a PMC record must name to be THIS run's kernel (None for the kernels no PMC set is kept for)."""
    f = dict(kv.split("=", 1) for kv in desc.replace("|", " ").split() if "=" in kv)
    if precision != "fp32" or dims != 2 or f.get("symmetric") != "1":
        return None
    mm = 0 if f.get("uniform_mass") == "1" else 2 if f.get("mass_scaled") == "1" else 1
    b = lambda v: "true" if v else "false"
    return f"nbk::force_sym_f32<{1 if rsqrt == 'quake' else 0}, {mm}, {b(f.get('chunk_pairs') == '1')}, {b(f.get('tile') == '512')}>"


def pmc_lookup(book, kernel_full, n, items):
    """The PMC record (profiles/hbm_traffic.json, written by tools/summarize_profile.py) of EXACTLY the kernel that ran:
    same template instantiation, same N, same number of work items (= same plan: chunks per item, tile, tail).  Returns
    (entry or None, status): "match", "stale: ..." (records exist for this N but for another kernel / plan: a kernel change
    or other tuning since the counters were taken), or "none"."""
    entries = (book or {}).get("entries") or []
    if kernel_full is None:
        return None, "none: no PMC set is kept for this kernel"
    same_n = [e for e in entries if e.get("n") == n]
    for e in same_n:
        if e.get("kernel") == kernel_full and e.get("grid_workgroups") in (None, 0, items):
            return e, "match"
    if same_n:
        have = ", ".join(f"{e.get('kernel')} x {e.get('grid_workgroups')} items @ {e.get('commit')}" for e in same_n)
        return None, f"stale: this run's kernel is {kernel_full} x {items} items; the PMC records for N = {n} are of: {have}"
    return None, "none"


def live_pmc(args, kernel_full, items, timeout_s=45.0):
    """The PMC figures of THIS run's dominant kernel, collected on THIS box after the timed region: three separate `rocprofv3 --pmc`
    passes (MI355X_MICROARCH.md, HBM / rocprofv3 section: FETCH_SIZE and WRITE_SIZE in separate passes, FETCH_SIZE doubled on gfx950)
    of this very command in its shortest form (3 steps, nothing but the step loop) as CHILD processes — `rocprofv3 ... -- python3
    bench.py ...`, the program itself after `--`, nothing exec'ed in place.  Returns (entry or None, status): the entry has the shape
    of a profiles/hbm_traffic.json record and is accepted only for exactly the kernel instantiation and item count that ran here.
    Never part of `value`; bounded by `timeout_s` per pass; any failure leaves the counter fields to the committed record or null."""
    import csv
    import glob
    import shutil
    import tempfile
    if kernel_full is None:
        return None, "none: no PMC collection for this kernel"
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not Path(rocprof).exists():
        return None, "none: rocprofv3 not found on this box"
    work = ["--nbodies", str(args.n), "--precision", args.precision, "--rsqrt", args.rsqrt, "--dims", str(args.dims), "--mass-scaling", args.mass_scaling]
    work += (["--general-mass"] if args.general_mass else []) + (["--no-symmetry"] if args.no_symmetry else [])
    work += ["--chunks-per-item", str(args.chunks_per_item)] if args.chunks_per_item else []
    child = [sys.executable, str(Path(__file__).resolve()), *work, "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-kernel-events",
             "--no-sustained", "--no-secondary", "--no-live-pmc"]
    tmp = tempfile.mkdtemp(prefix="nb_live_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    sets = (("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU"), ("GRBM_GUI_ACTIVE", "FETCH_SIZE"), ("WRITE_SIZE",))
    try:
        for k, counters in enumerate(sets):
            cmd = [rocprof, "--pmc", *counters, "--output-format", "csv", "-d", f"{tmp}/{k}", "--", *child]
            # its own session: on a time-out the whole group goes (the profiler AND the child it started), nothing is left on the GPU
            proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, start_new_session=True)
            try:
                out, _ = proc.communicate(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except (ProcessLookupError, PermissionError):
                    pass
                proc.wait()
                return None, f"live collection failed: pass {k} ({' '.join(counters)}) exceeded {timeout_s:.0f} s"
            if proc.returncode != 0:
                return None, f"live collection failed: pass {k} ({' '.join(counters)}) ended with status {proc.returncode}: {(out or '')[-160:]!r}"
        vals, grid = {}, None
        for f in glob.glob(f"{tmp}/*/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                if row["Kernel_Name"].split("(")[0].replace("void ", "") != kernel_full:
                    continue
                vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
                grid = int(row.get("Grid_Size") or 0) // 256 or grid
        need = ("SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE", "FETCH_SIZE", "WRITE_SIZE")
        if any(c not in vals for c in need):
            return None, f"live collection failed: no counter rows for {kernel_full} ({sorted(vals)} found)"
        if grid not in (None, items):
            return None, f"live collection: the profiled child ran {grid} work items, this run {items}"
        mean = {c: sum(v) / len(v) for c, v in vals.items()}
        read_raw, written = mean["FETCH_SIZE"] * 1024.0, mean["WRITE_SIZE"] * 1024.0
        entry = {"round": "live", "commit": None, "kernel": kernel_full, "n": args.n, "grid_workgroups": grid,
                 "valu_busy": 4.0 * mean["SQ_ACTIVE_INST_VALU"] / 1024.0 / (mean["GRBM_GUI_ACTIVE"] / 8.0),
                 "valu_cycles_per_inst": (4.0 * mean["SQ_ACTIVE_INST_VALU"] / mean["SQ_INSTS_VALU"]) if mean.get("SQ_INSTS_VALU") else None,
                 "force_kernel_hbm_bytes_per_launch": 2.0 * read_raw + written, "read_bytes_raw_FETCH_SIZE": read_raw,
                 "read_correction": "x2 (gfx950 FETCH_SIZE counts 64 B per 128-B request)", "write_bytes_WRITE_SIZE": written,
                 "launches_per_pass": min(len(v) for v in vals.values())}
        return entry, "live"
    except Exception as e:       # noqa: BLE001 - an optional measurement: never the reason a bench line is lost
        return None, f"live collection failed: {type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def cpu_baseline(ic, n, target_s=6.0):
    """The CPU figures beside the GPU number, on a bounded i-slice of the same workload (all n j-particles):
    * kind "reference" — the compiled reference's OWN pairwise loop (Quadtree::acc, Quadtree.hpp:113-155, driven as a
      direct sum through a single-leaf tree and fanned over std::async tasks like Simulation::attract) from
      oracle/_ref/libnbref.so, when that prebuilt checker travelled here;
    * "port" — the C restatement of that loop (bit-identical results; SoA, OpenMP, vectorised across i), always.
    The main object is the reference when available, with the port beside it; otherwise the port."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import nbo  # checker / baseline only

    st = nbo.state_from_bodies(ic)
    threads = nbo.set_threads(0)
    host = host_cpu_info()
    where = (f"OpenMP {threads} threads (nproc {host['nproc']}, affinity {host['affinity']}, cgroup quota {host['cgroup_cpus']} CPUs) "
             f"on {host['model']}")
    calib = min(n, 64 * threads)                 # a few i-blocks per thread
    t0 = time.perf_counter()
    nbo.accel_f32(st, EPS, nbo.RSQRT_QUAKE, 0, calib)
    t_cal = time.perf_counter() - t0
    rate = calib * n / max(t_cal, 1e-6)
    islice = int(min(n, max(calib, (rate * target_s / n) // (16 * threads) * (16 * threads))))
    t0 = time.perf_counter()
    ax, ay = nbo.accel_f32(st, EPS, nbo.RSQRT_QUAKE, 0, islice)
    t = time.perf_counter() - t0
    port = {
        "value": islice * n / t,
        "unit": "pair interactions/s",
        "cores": threads,
        "kind": "port",
        "host": host,
        "sample": f"reference pairwise arithmetic (Quake rsqrt, fp32, sequential j) for the first {islice} of {n} "
                  f"i-particles against all {n} j = {islice * n:.3e} pairs in {t:.2f} s; {where}",
    }
    try:
        if not nbo.have_ref() or not hasattr(nbo.ref(), "ref_direct_acc_timed"):
            return port
        flat = nbo.state_to_flat(st).astype(np.float32)
        secs, _ = nbo.ref_direct_acc_timed(flat, EPS, 0, 16 * threads, threads)
        rrate = 16 * threads * n / max(secs, 1e-6)
        rslice = int(min(n, max(16 * threads, (rrate * target_s / n) // threads * threads)))
        secs, out = nbo.ref_direct_acc_timed(flat, EPS, 0, rslice, threads)
        k = min(rslice, islice)
        same = bool(np.array_equal(out[:k, 4], ax[:k]) and np.array_equal(out[:k, 5], ay[:k]))
    except (OSError, RuntimeError, AttributeError):
        return port
    return {
        "value": rslice * n / secs,
        "unit": "pair interactions/s",
        "cores": threads,
        "kind": "reference",
        "host": host,
        "sample": f"the compiled reference's own Quadtree::acc leaf loop (64-byte AoS bodies, Quake rsqrt, fp32) for the first {rslice} of {n} "
                  f"i-particles against all {n} j = {rslice * n:.3e} pairs in {secs:.2f} s, {threads} std::async tasks on contiguous i-chunks as in "
                  f"Simulation::attract; " + where.replace("OpenMP ", "", 1),
        "bitwise_equal_to_port": same,
        "port": {k2: port[k2] for k2 in ("value", "unit", "cores", "sample")},
    }


EXIT_PARITY = 4      # exit status of every rank when the sharded trajectory fails its self-check


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)      # 0.37 s at one GPU, ~50 ms at eight: enough steps to average over
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--n", "--nbodies", dest="n", type=int, default=N_DEFAULT,
                    help="number of bodies (use --nbodies under torch.distributed.run, whose own parser finds --n ambiguous)")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "fp64"])
    ap.add_argument("--rsqrt", default="exact", choices=["exact", "quake"])
    ap.add_argument("--dims", type=int, default=2, choices=[2, 3], help="3 = the 3-D build extension (not the headline config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true", help="skip per-launch HIP events (roofline from wall time)")
    ap.add_argument("--backend", default=os.environ.get("NB_BENCH_BACKEND", "nccl"), choices=["nccl", "gloo"],
                    help="process-group backend; gloo + --share-gpu rehearses the multi-rank path on a one-GPU box")
    ap.add_argument("--protocol", default="tune", choices=["tune", "auto", "symmetric", "allreduce", "allgather"],
                    help="multi-GPU exchange: tune (default) times a few validated steps of north_star's all-gather protocol, of the replicated "
                         "all-reduce protocol and of the symmetric pair split (reduce-scatter + all-gather; with and without held-back late items) "
                         "before the timed region and keeps the fastest; the others force one")
    ap.add_argument("--driver", default="tune", choices=["tune", "torch", "c"],
                    help="multi-GPU step loop: c = the library's own RCCL loop (nb_comm_step: one foreign call for all steps; north_star: "
                         "'host code stays in C'); torch = collectives through torch.distributed between the library's split-step calls; "
                         "tune (default) = staged: the start-up timing runs over torch-driven candidates, the winner is measured in full and "
                         "becomes the fallback line, then the C loop runs the same protocol as ONE challenger under its own deadline and is "
                         "printed if it passed the self-check and is not more than 1 %% slower")
    ap.add_argument("--deadline", type=float, default=420.0,
                    help="seconds the whole multi-GPU run may take before the rank prints the phase it is in and exits with status 3 "
                         "— or, once the safe-first configuration has been measured, prints THAT line and exits 0 (a stuck collective "
                         "must not become a silent hang, and must fit inside the driver's own 600 s limit); 0 = none")
    ap.add_argument("--candidate-deadline", type=float, default=45.0,
                    help="seconds one start-up candidate (or forming the C-level communicator) may take before the same")
    ap.add_argument("--one-process", action="store_true",
                    help="N > 1 without a launcher: ONE process drives the N sharded handles (one per device) through the library's own RCCL "
                         "loop (nb_comm_create_all) instead of starting one child process per GPU; what `--gpus N` falls back to where "
                         "torch.distributed.run is missing.  With --share-gpu (rehearsal) the handles share GPU 0 and exchange in-process")
    ap.add_argument("--no-sustained", action="store_true", help="skip the >= 2 s settled-rate measurement after the timed region")
    ap.add_argument("--no-symmetry", action="store_true", help="single GPU: the one-sided LDS-tiled kernel (north_star's design)")
    ap.add_argument("--general-mass", action="store_true", help="disable the equal-mass specialisation of the kernels")
    ap.add_argument("--mass-scaling", default="off", choices=["off", "on", "measured"],
                    help="with --general-mass: fold the masses into the pair geometry — off (default: both per-pair mass multiplies, the 12 + 2 body), "
                         "on (NB_FLAG_MASS_SCALING), measured (NB_FLAG_MASS_SCALING_MEASURED: the library measures at upload whether that is harmless "
                         "for these bodies; unsharded handles)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the untimed general-mass secondary measurement")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="one GPU: do not collect this run's own PMC figures (three short `rocprofv3 --pmc` child runs of this command after the "
                         "timed region: roofline.valu_busy / traffic of THIS box); the committed record of profiles/hbm_traffic.json is then used")
    ap.add_argument("--chunks-per-item", type=int, default=0, help="symmetric kernel: force nb_params.sym_chunks_per_item (tuning sweeps)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: all ranks use GPU (LOCAL_RANK mod device_count)")
    ap.add_argument("--no-parity-check", action="store_true",
                    help="multi-GPU: skip the self-check of the sharded trajectory against one unsharded handle on rank 0")
    ap.add_argument("--rehearse-sharded", action="store_true",
                    help="ONE GPU, one rank: run the whole multi-rank flow anyway — process group (RCCL, one rank), sharded handles "
                         "(NB_FLAG_SHARD_SINGLE), safe-first measurement, start-up timing of every protocol under both step loops, "
                         "self-checks, the N > 1 line.  A rehearsal of the code a node run executes, not a measurement")
    ap.add_argument("--no-safe-first", action="store_true",
                    help="multi-GPU: do not measure the plain all-gather / torch-driven configuration before anything else is tried")
    return ap.parse_args(argv)


class UnshardedReference:
    """Rank 0's checker of a sharded run: ONE unsharded handle of the same system on this rank's GPU (the single-GPU
    product path, itself parity-tested against the oracle and the reference's goldens at this size:
    tests/test_headline_gpu.py).  ``rows(steps)`` = its [pos | vel] rows after ``steps`` steps from the initial
    conditions; the handle keeps stepping forward and results are cached by step count."""

    def __init__(self, make_handle):
        self.make_handle, self.h, self.frame, self.cache = make_handle, None, 0, {}

    def rows(self, steps: int, dt: float = DT):
        from nbodysim_amd.dist import state_rows
        if steps not in self.cache:
            if self.h is None or steps < self.frame:
                self.close()
                self.h, self.frame = self.make_handle(), 0
            self.h.advance(steps - self.frame, dt)
            self.frame = steps
            self.cache[steps] = state_rows(self.h.sync())
        return self.cache[steps]

    def close(self):
        if self.h is not None:
            self.h.close()
            self.h = None


def timed_region(sim, args, world, rank, barrier, device_sync, reference=None, phase=None, label="", check=True):
    """W untimed warm-up steps, then EXACTLY K steps between barrier + device synchronisation on both sides; the caller
    takes the MAX over ranks (`elapsed` is already that).  Several ranks: the sharded state is compared with the
    unsharded reference after the warm-up (a failure raises ParityError on every rank BEFORE anything is timed) and again
    after the timed steps (reported; the caller decides).  Returns the measurement as a dictionary."""
    from nbodysim_amd.dist import ParityError, compare_with_unsharded

    sharded = world > 1 or hasattr(sim, "plan")        # a DistributedSimulation — also the single-rank rehearsal of one
    inner = sim.sim if sharded else sim
    phase = phase if phase is not None else {}
    parity = None
    k0, u0 = sim.energy()
    phase["now"] = f"{label}warm-up"
    sim.advance(args.warmup, DT)
    sim.wait()
    if sharded and check:
        phase["now"] = f"{label}self-check after the warm-up"
        parity = compare_with_unsharded(sim.owned_rows(), sim.plan, (lambda: reference.rows(args.warmup)) if reference else None, args.warmup)
        if not parity["ok"]:
            raise ParityError(f"{label}{sim.protocol} protocol, {sim.driver} loop, after the warm-up", parity)
    if not args.no_kernel_events:
        inner.profile(True)
        if sharded:
            sim.profile_phases(True)
    phase["now"] = f"{label}timed region"
    barrier()
    device_sync()
    sampler = DeviceSampler(period_s=0.02 if not sharded else 0.1) if rank == 0 else None   # rare on the Python-driven sharded loop
    if sampler:
        sampler.start()
    t0 = time.perf_counter()
    sim.advance(args.steps, DT)
    sim.wait()
    device_sync()
    barrier()
    elapsed = time.perf_counter() - t0
    device_state = sampler.stop() if sampler else None
    force_ms, launches = (0.0, 0)
    if not args.no_kernel_events:
        force_ms, launches = inner.profile_read()
        inner.profile(False)
    phases = sim.phase_report() if (sharded and not args.no_kernel_events) else None
    if sharded and not args.no_kernel_events:
        sim.profile_phases(False)
    k1, u1 = sim.energy()
    phases_max = None
    if sharded:
        elapsed = reduce_max_over_ranks(elapsed, world)
        if phases is not None and getattr(sim, "world", 0) > 1:       # one process drives every rank: its report is already the slowest handle's
            phases_max = {k: v for k, v in phases.items() if isinstance(v, float)}
        elif phases is not None:   # slowest rank per phase, so an exposed collective on any rank shows
            import torch
            import torch.distributed as dist
            from nbodysim_amd.dist import _comm_device
            keys = [k for k, v in phases.items() if isinstance(v, float)]
            pt = torch.tensor([phases[k] for k in keys], dtype=torch.float64, device=_comm_device())
            dist.all_reduce(pt, op=dist.ReduceOp.MAX)
            phases_max = {k: float(v) for k, v in zip(keys, pt.tolist())}
        if check:
            phase["now"] = f"{label}self-check after the timed region"
            total = args.warmup + args.steps
            after = compare_with_unsharded(sim.owned_rows(), sim.plan, (lambda: reference.rows(total)) if reference else None, total)
            parity = {**parity, "after_timed_region": {k: after[k] for k in ("max_rel_pos", "max_rel_vel", "steps", "ok", "worst_particle", "worst_rank", "error")},
                      "ok": bool(parity["ok"] and after["ok"])}
    return {"elapsed": elapsed, "device_state": device_state, "force_ms": force_ms, "launches": launches, "phases": phases,
            "phases_max": phases_max, "energy": (k0, u0, k1, u1), "parity": parity}


def driver_reason(sim):
    """Which host runs the step loop of a sharded run, and why (from the start-up timing's record)."""
    t = getattr(sim, "tuning", None)
    drv = getattr(sim, "driver", None)
    if hasattr(sim, "transport"):           # nbodysim_amd.local_ranks.LocalRanksSimulation: no launcher, no process group
        return f"{drv}: ONE process drives every rank ({sim.transport} transport)"
    if not t:
        return f"{drv}: named on the command line (no start-up timing)" if drv else None
    ms, val, chosen = t.get("ms_per_step", {}), t.get("validation", {}), t.get("chosen", "")
    proto = chosen[2:] if chosen.startswith("c:") else chosen
    c_ms, t_ms = ms.get("c:" + proto), ms.get(proto)
    fmt = lambda v: "unavailable" if v is None else f"{v:.3f} ms/step"
    if not any(k.startswith("c:") for k in ms):
        return ("torch: the start-up timing ran over torch-driven candidates; the library's C loop challenges its winner afterwards where there is "
                "an RCCL process group (config.c_loop_challenger)")
    vs = (val.get("c:" + proto) or {}).get("vs_torch_loop")
    why = f"C loop {fmt(c_ms)} vs torch-driven {fmt(t_ms)} in the start-up timing of the '{proto}' protocol"
    if vs:
        why += f"; C trial against the torch trial: {vs}"
    if t.get("failed"):
        why += f"; skipped by agreement: {sorted(t['failed'])}"
    return f"{drv}: {why}"


def make_line(args, n, world, sim, m, sustained=None, secondary=None):
    """The ONE JSON line (rank 0) from a timed region's measurement `m` (timed_region) of `sim`."""
    sharded = world > 1 or hasattr(sim, "plan")
    inner = sim.sim if sharded else sim
    secondary = secondary or {}
    general, scaled, lds_tiled, auto = secondary.get("general"), secondary.get("scaled"), secondary.get("lds_tiled"), secondary.get("measured")
    elapsed, force_ms, launches = m["elapsed"], m["force_ms"], m["launches"]
    k0, u0, k1, u1 = m["energy"]
    flop_per_pair = FLOP_PER_PAIR if args.dims == 2 else 20.0   # 3-D: one more sub, fma, fma per pair side
    pairs_per_step = float(n) * float(n)
    value = pairs_per_step * args.steps / elapsed
    peak = PEAK_FP32_TFLOPS if args.precision == "fp32" else PEAK_FP32_TFLOPS / 2
    info = inner.sym_info()
    symmetric = bool(info["enabled"])
    um = "uniform_mass=1" in inner.describe()
    kernel = ("force_sym" if symmetric else "force_tiled") + ("3" if args.dims == 3 else "") + ("_f32" if args.precision == "fp32" else "_f64")
    # force launches per step on this rank: 1 (single GPU) or up to 3 (local + cross + late / local + remote ranges)
    owned = getattr(sim, "owned_per_rank", sim.plan.i_count) if sharded else n    # rank 0's block (ragged splits: ceil(n / world))
    pairs_this_rank = float(owned) * float(n) * args.steps             # its share of the ordered pairs, whatever the protocol
    if launches and force_ms > 0 and not sharded:
        kern_s = force_ms * 1e-3
        avg_launch_ms = force_ms / launches
    else:
        # sharded ranks run their force launches on two streams at once (their event intervals overlap), so the
        # per-rank figure is taken over the wall time of the step, collectives included
        kern_s = elapsed
        avg_launch_ms = force_ms / launches if launches else None
    achieved = flop_per_pair * pairs_this_rank / kern_s / 1e12
    # flops the kernel really issues per launch (whole system on one GPU): symmetric items evaluate each unordered
    # pair once for both particles, the diagonal items and the one-sided kernel every ordered pair
    ex = EXECUTED[(args.precision, args.dims)]
    if symmetric and not sharded:
        tile = float(info.get("tile_particles") or 2048)        # 2048 (classic) or 512 (wave-split kernels) stationary particles per item
        diag_units = info["tiles"] * tile / 64.0                 # every tile meets its own chunks one-sidedly
        sym_units = info["units_local"] + info["units_cross"] + info["units_late"] - diag_units
        exec_flop = (sym_units * ex["sym"][0 if um else 1] + diag_units * ex["one"][0 if um else 1]) * tile * 64.0
    elif not sharded:
        exec_flop = ex["one"][0 if um else 1] * pairs_per_step
    else:
        exec_flop = None
    executed = exec_flop * args.steps / kern_s / 1e12 if exec_flop else None
    # HBM bytes of one launch from the work plan: every item writes its stationary row and its travelling
    # partials once (plain stores, no re-reads); the positions (and masses) are read from HBM once, later reads hit L2
    esz = (8 if args.precision == "fp32" else 16) * (2 if args.dims == 3 else 1)
    traffic = float(info["slab_s_bytes"] + info["slab_r_bytes"] + n * esz) if (symmetric and not sharded) else None
    book = {}
    tfile = ROOT / "profiles" / "hbm_traffic.json"   # PMC-derived (tools/gpu_round.sh pmc + tools/summarize_profile.py)
    if tfile.exists() and not sharded:
        try:
            book = json.loads(tfile.read_text())
        except Exception:
            book = {}
    kernel_full = kernel_instantiation(inner.describe(), args.precision, args.dims, args.rsqrt) if not sharded else None
    pmc, pmc_status = pmc_lookup(book, kernel_full, n, info["items"])
    live, live_status = secondary.get("live_pmc") or (None, None)
    committed = pmc                                   # what the committed record says, kept beside a live figure as a cross-check
    if live is not None:
        pmc, pmc_status = live, "live"
    elif live_status:
        pmc_status = f"{pmc_status} ({live_status})"
    pmc_ok = pmc is not None
    pmc = pmc or {}
    traffic_pmc = pmc.get("force_kernel_hbm_bytes_per_launch") if pmc_ok else None
    # `traffic` is a COUNTER figure or nothing: with no PMC record of exactly this kernel, N and plan (pmc_status != "match") it
    # stays null and the plan-derived estimate is reported under its own name (traffic_plan) — never in the counter's place
    traffic_reported = traffic_pmc if traffic_pmc else None
    # the settled counterpart of `frac`: the same algorithmic flops over the mean launch duration of the >= 2 s stretch AFTER the timed
    # region (the driver's K steps sit in the power controller's burst window: 0.880 vs 0.867 settled in round 5)
    frac_sustained = None
    if sustained and not sharded:
        s_ms = sustained.get("avg_launch_ms") or sustained.get("ms_per_step")
        frac_sustained = flop_per_pair * pairs_per_step / (s_ms * 1e-3) / 1e12 / peak if s_ms else None
    if not sharded:
        workload = (f"N={n} {args.precision} direct O(N^2), one MI355X, kernel {kernel}: "
                    + (f"symmetric pair items ({info['items']} workgroups x {info['chunks_per_item']} chunks of 64, stationary particles in "
                       f"registers, travelling chunk rotated through the lanes)" if symmetric else "one-sided, j-particles through LDS tiles of 256"))
    else:
        workload = (f"N={n} {args.precision} direct O(N^2) sharded over {world} MI355X, "
                    + {"symmetric": "symmetric pair split: reduce-scatter of accelerations + all-gather of (x,y) per step",
                       "allreduce": "symmetric pair split, replicated integration: one all-reduce of the accelerations per step",
                       "allgather": "all-gather of (x,y) per step overlapped with the local-tile force"}[sim.protocol])
    frac_of = lambda ms: flop_per_pair * pairs_per_step / (ms * 1e-3) / 1e12 / peak
    mass_note = ("equal masses (the contract's Plummer data): both per-pair mass multiplies hoisted, 10 + 2 instructions per body" if um else
                 "individual masses: 12 + 2 instructions per body")
    line = {
        "metric": f"particle-pair interactions/sec at N={n:,} (direct O(N^2) softened gravity + kick/drift step)",
        "value": value,
        "unit": "pair interactions/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "steps_per_s": args.steps / elapsed,
        "particle_steps_per_s": n * args.steps / elapsed,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32" if args.precision == "fp32" else "f64",
        "data": "synthetic (3-D Plummer sphere projected to 2-D, mt19937 seed 42, equal masses, eps=0.01, dt=1e-3)",
        "config": {
            "workload": workload,
            "n": n, "eps": EPS, "dt": DT, "rsqrt": args.rsqrt, "sum_order": "tiled", "dims": args.dims,
            "parallelism": f"i-block x{world}" if sharded else "single GPU",
            "backend": args.backend if sharded else None,
            "protocol": getattr(sim, "protocol", None) if sharded else None,
            "protocol_tuning": getattr(sim, "tuning", None) if sharded else None,
            "driver": getattr(sim, "driver", None) if sharded else None,
            "driver_choice": driver_reason(sim) if sharded else None,
            "uniform_mass_specialisation": um,
            "launch": inner.describe(),
        },
        "roofline": {
            "bound": "valu",
            "achieved": achieved,
            "peak": peak,
            "unit": "TFLOP/s",
            "frac": achieved / peak,
            "frac_sustained": frac_sustained,
            "frac_sustained_kind": ("the settled figure — quote THIS one: same flops over the mean force-launch duration of the >= 2 s of steps run after the "
                                    "timed region (`sustained`); `frac` is the driver's K timed steps, which start before the power controller has settled"
                                    if frac_sustained is not None else None),
            "frac_kind": "algorithmic: 14 flop x N^2 ORDERED pairs / kernel time (the contract's definition); the kernel issues fewer "
                         "flops than that because it evaluates each unordered pair once - see executed_frac",
            "mass_model": mass_note,
            # the two headline fractions side by side: the contract's synthetic data has equal masses, the
            # reference's own bodies (Simulation.hpp:565-577) do not
            "frac_equal_masses": (achieved / peak) if (um and not sharded) else None,
            "frac_individual_masses": ((achieved / peak) if (not um and not sharded) else (frac_of(general["avg_launch_ms"]) if general else None)),
            "frac_individual_masses_default": ((achieved / peak) if (not um and not sharded and args.mass_scaling == "off") else
                                               (frac_of(general["avg_launch_ms"]) if general else None)),
            "executed_tflops": executed,
            "executed_frac": executed / peak if executed else None,
            "executed_flop_per_unordered_pair": ex["sym"][0 if um else 1] if symmetric else None,
            "valu_busy": pmc.get("valu_busy") if pmc_ok else None,
            "valu_busy_source": (("LIVE: separate rocprofv3 --pmc passes of this command (3 steps, child processes) on THIS box, run after the timed "
                                  "region of this very invocation") if pmc_status == "live" else
                                 ("profiles/hbm_traffic.json: a separate rocprofv3 --pmc run of this command on ANOTHER MI355X box, "
                                  "not a measurement of this run") if pmc_ok else None),
            "valu_busy_committed": (committed or {}).get("valu_busy") if pmc_status == "live" else None,
            "traffic_committed": (committed or {}).get("force_kernel_hbm_bytes_per_launch") if pmc_status == "live" else None,
            "valu_cycles_per_inst": pmc.get("valu_cycles_per_inst") if pmc_ok else None,
            "pmc_status": pmc_status,
            "pmc_commit": pmc.get("commit") if pmc_ok else None,
            "kernel_instantiation": kernel_full,
            "traffic": traffic_reported,
            "traffic_source": (("PMC, LIVE on this box in this run" if pmc_status == "live" else "PMC: profiles/hbm_traffic.json") +
                               " (separate rocprofv3 --pmc passes of this command on an MI355X: FETCH_SIZE x2 "
                               "+ WRITE_SIZE per launch of the force kernel)" if traffic_pmc else
                               "null: no PMC record of exactly this kernel instantiation, N and work plan is kept (pmc_status); the work plan's own "
                               "estimate is traffic_plan (stationary slab rows + travelling partials written once per launch + positions read once)"),
            "traffic_plan": traffic,
            "traffic_pmc": traffic_pmc,
            "flop_per_pair": flop_per_pair,
            "kernel": kernel,
            "avg_launch_ms": avg_launch_ms,
            "launches": launches,
            "general_mass": ({**general, "frac": frac_of(general["avg_launch_ms"]),
                              "note": "same kernel without the equal-mass specialisation (individual masses, 12 + 2 instructions per "
                                      "body): untimed secondary run"}
                             if general else None),
            "general_mass_scaled": ({**scaled, "frac": frac_of(scaled["avg_launch_ms"]),
                                     "note": "NB_FLAG_MASS_SCALING: masses folded into the pair geometry (11 + 2 per body) at the caller's request; off by "
                                             "default (DESIGN.md §4.1: the extra rounding is data-dependent)"}
                                    if scaled else None),
            "general_mass_measured": ({**auto, "frac": frac_of(auto["avg_launch_ms"]),
                                       "note": "NB_FLAG_MASS_SCALING_MEASURED (opt-in): the library compares the two bodies on the uploaded data (two force "
                                               "evaluations) and folds the masses only if the accelerations agree to 2e-6 of the force scale"}
                                      if auto else None),
            "one_sided_lds_tiled": ({**lds_tiled, "kernel": "force_tiled" + ("3" if args.dims == 3 else "") + ("_f32" if args.precision == "fp32" else "_f64"),
                                     "frac": frac_of(lds_tiled["avg_launch_ms"]),
                                     "note": "north_star's kernel design (every ordered pair, j-tiles of 256 in LDS) on the same workload: untimed secondary run"}
                                    if lds_tiled else None),
            "note": "fp32 vector-ALU bound (no dense contraction for MFMA; the f32 MFMA peak equals the vector peak, 157.3 TF); "
                    "algorithmic HBM bytes are 36 B per particle-step, ~1e5 flop/B: HBM is not the bound, the slab traffic is the "
                    "price of evaluating every pair once with plain stores (no atomics, bit-reproducible)",
            "algorithmic_hbm_gbps": BYTES_PER_PARTICLE_STEP * n * args.steps / elapsed / 1e9,
            "kernel_hbm_gbps": (traffic_reported / (avg_launch_ms * 1e-3) / 1e9) if (traffic_reported and avg_launch_ms) else None,
            "kernel_hbm_gbps_plan": (traffic / (avg_launch_ms * 1e-3) / 1e9) if (traffic and avg_launch_ms) else None,
            "arithmetic_intensity_flop_per_byte": flop_per_pair * float(n) / BYTES_PER_PARTICLE_STEP,       # algorithmic: 14 N flop over 36 B per particle-step
            "arithmetic_intensity_vs_traffic": (flop_per_pair * float(n) * float(n) / (traffic_reported or traffic)) if (traffic_reported or traffic) else None,
        },
        "device_state": m["device_state"],
        "sustained": ({**sustained, "value": float(n) * float(n) * sustained["steps"] / sustained["seconds"],
                       "frac": (flop_per_pair * float(n) * float(n) / world / (sustained["ms_per_step"] * 1e-3) / 1e12 / peak),
                       "note": "settled rate over >= 2 s of steps run AFTER the timed region (whole step, wall clock); the timed region "
                               "above is the driver's K steps and may sit in the power controller's burst window"}
                      if sustained else None),
        "energy": {"e0": k0 + u0, "e1": k1 + u1, "rel_drift": (k1 + u1 - k0 - u0) / (k0 + u0),
                   "steps": args.warmup + args.steps},
    }
    if sharded:
        line["phases_ms"] = {"rank0": m["phases"], "max_over_ranks": m["phases_max"],
                             "note": "per step, on the compute stream (waits included): local pairs | wait for the all-gather | "
                                     "cross pairs + slab gather | reduce-scatter | kick+drift; the all-gather itself runs on RCCL's stream"}
        line["parity_check"] = ({**m["parity"],
                                 "against": "one unsharded handle of the same system on rank 0's GPU (the single-GPU path), same steps; "
                                            "max over particles of |d pos| / |pos| and |d vel| / |vel|"}
                                if m.get("parity") else {"ok": None, "skipped": "--no-parity-check"})
    return line


class StopSignals:
    """SIGTERM / SIGINT -> ``on_stop(signum)``, called from a watcher THREAD the moment the signal arrives.  A Python-level
    signal handler only runs when the main thread next executes bytecode — never while it sits inside a collective or a
    device synchronisation, which is exactly where a run that is being stopped from outside sits.  The C-level handler,
    however, writes the signal's number to the wake-up descriptor at once (signal.set_wakeup_fd); the thread reads it there.
    Inactive outside the main thread (signal handlers cannot be installed there)."""

    def __init__(self, on_stop, signals=(signal.SIGTERM, signal.SIGINT)):
        self.on_stop, self.signals, self.active = on_stop, tuple(signals), False

    def __enter__(self):
        import socket
        import threading
        if threading.current_thread() is not threading.main_thread():
            return self
        self.r, self.w = socket.socketpair()
        self.w.setblocking(False)
        self.old_fd = signal.set_wakeup_fd(self.w.fileno(), warn_on_full_buffer=False)
        self.old = {sg: signal.signal(sg, lambda _s, _f: None) for sg in self.signals}
        self.active = True

        def watch():
            while True:
                try:
                    data = self.r.recv(16)
                except OSError:
                    return
                if not data:
                    return
                for b in data:
                    if b in self.signals:
                        self.on_stop(int(b))
        threading.Thread(target=watch, daemon=True).start()
        return self

    def __exit__(self, *exc):
        if self.active:
            self.active = False
            for sg, h in self.old.items():
                signal.signal(sg, h)
            signal.set_wakeup_fd(self.old_fd)
            for sk in (self.w, self.r):
                try:
                    sk.close()
                except OSError:
                    pass
        return False


def run_sharded(args, ic, n, world, rank, make_sim, make_reference, device_sync, barrier, emit=None) -> int:
    """The whole multi-rank run given the engine (`make_sim(protocol, driver)` -> a DistributedSimulation-shaped object,
    `make_reference()` -> rank 0's unsharded handle); returns the process's exit status.  bench.py's main() calls it with the
    GPU engine; tests/test_hang_guards.py drives it over gloo with a stand-in engine on the CPU.

    1. SAFE FIRST.  Unless a protocol was forced, the plainest configuration — north_star's all-gather protocol, collectives
       issued through torch.distributed — is created, checked against the unsharded handle, and timed (W + K steps, the full
       contract) before anything else runs.  Its line is kept.
    2. The start-up timing of the torch-driven candidates (each validated, a failing one skipped by agreement) and the timed region
       of the winner: a full, validated measurement, which replaces the safe-first line as the fallback if it is faster.
    2b. (--driver tune, RCCL) ONE challenger: the winning protocol under the library's own C loop (nb_comm_step), created, checked,
       timed and checked again like everything else.  It is printed if it passed and is not more than 1 % slower (north_star's host
       loop is the C one); either way the line carries both figures (config.c_loop_challenger / config.torch_driven).
    3. If step 2 cannot finish — a candidate hangs (its Watchdog expires), the winner fails its self-check, anything
       raises — the line of step 1 is printed instead, with `fallback` saying why, and the run ends with status 0: the first
       node this meets cannot lose the measurement to an optional faster path.  Only if NOTHING valid was measured does the
       run end non-zero (3 = deadline, 4 = self-check).  "Valid" = the self-check passed after the warm-up AND after the
       timed steps: a safe-first run that leaves the tolerance during the K timed steps is not kept as the fallback.
    4. A stop from outside (SIGTERM / SIGINT: a driver's time limit, a launcher tearing the job down) is one more way not to
       finish: rank 0 prints the safe-first line it holds, with `fallback.why` naming the signal, and every rank that knows a
       valid line exists ends with status 0 (StopSignals: it works while the main thread is inside a collective)."""
    import threading

    from nbodysim_amd.dist import ParityError, Watchdog

    emit = emit or (lambda text: print(text, flush=True))
    phase = {"now": "starting"}
    state = {"safe_line": None, "first_line": None, "printed": False}
    lock = threading.Lock()

    def print_once(line) -> bool:
        with lock:
            if state["printed"]:
                return False
            state["printed"] = True
        if rank == 0:
            emit(json.dumps(line))
        return True

    def last_resort(msg):
        """Watchdog expiry somewhere after the safe configuration was measured: print its line, end with status 0."""
        if state["safe_line"] is None:
            return None
        line = dict(state["safe_line"])
        line["fallback"] = {"used": True, "why": msg, "phase": phase["now"], "protocol_tuning": state.get("tuning")}
        print_once(line)
        return 0

    def on_stop(signum):
        """SIGTERM / SIGINT (watcher thread): the same last resort as an expired deadline, then the process ends HERE — the main
        thread may be inside a collective that will never complete."""
        name = signal.Signals(signum).name
        sys.stderr.write(f"[bench] rank {rank}: stopped by {name} (phase: {phase['now']})"
                         + ("; printing the safe-first line\n" if state["safe_line"] is not None else "; nothing valid has been measured yet\n"))
        sys.stderr.flush()
        code = last_resort(f"stopped by {name} before the run could finish")
        sys.stdout.flush()
        os._exit(0 if code == 0 else 128 + signum)

    Watchdog.last_resort = last_resort
    main_watchdog = Watchdog(args.deadline, "running bench.py", report=lambda: f"phase: {phase['now']}", rank=rank)
    main_watchdog.__enter__()
    stop_signals = StopSignals(on_stop)
    stop_signals.__enter__()
    reference = UnshardedReference(make_reference) if (rank == 0 and not args.no_parity_check) else None
    check = not args.no_parity_check
    sim = None
    status = 0
    try:
        safe_first = args.protocol == "tune" and not args.no_safe_first and not args.no_symmetry
        safe_m = None
        if safe_first:
            phase["now"] = "safe first: creating the all-gather / torch-driven configuration"
            try:
                sim = make_sim("allgather", "torch")
                state["safe_driver"] = "torch" if getattr(sim, "driver", "torch") == "torch" else sim.driver
                safe_m = timed_region(sim, args, world, rank, barrier, device_sync, reference, phase, "safe first: ", check)
                safe_line = make_line(args, n, world, sim, safe_m)          # every rank builds it (cheap); rank 0 would print it
                # kept only if it is VALID: timed_region raises for a failed check after the warm-up, a failure after the timed steps
                # is only recorded in parity["ok"] — such a line must not become the fallback (it would be printed with status 0)
                safe_valid = safe_m["parity"] is None or bool(safe_m["parity"]["ok"])
                if safe_valid:
                    state["safe_line"] = state["first_line"] = safe_line
                else:
                    state["safe_failed"] = (f"parity_check failed (allgather protocol, torch loop, after the timed region): "
                                            f"{safe_m['parity'].get('after_timed_region')}")
                if rank == 0:                                                # progress on stderr: what a failed node run's log shows first
                    sys.stderr.write(f"[bench] safe first (all-gather, torch-driven): {safe_line['ms_per_step']:.3f} ms/step, parity_check "
                                     f"pos {safe_m['parity']['max_rel_pos']:.2e} vel {safe_m['parity']['max_rel_vel']:.2e} ok={safe_m['parity']['ok']}; "
                                     + ("kept as the fallback line\n" if safe_valid else "NOT kept: it left the tolerance during the timed steps\n")
                                     if safe_m["parity"] else
                                     f"[bench] safe first (all-gather, torch-driven): {safe_line['ms_per_step']:.3f} ms/step (self-check off)\n")
                    if safe_valid:      # the measurement itself, on stderr, the moment it exists: even a SIGKILL leaves it in the log
                        sys.stderr.write("[bench] safe-first line (stdout gets ONE line at the end; this copy is for the log): "
                                         + json.dumps({k: safe_line[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step")}) + "\n")
                    sys.stderr.flush()
            except ParityError as e:
                # the plainest protocol is wrong on this node: nothing later can be trusted to be judged by the same check either,
                # but the faster candidates are still tried — each is validated on its own
                if rank == 0:
                    sys.stderr.write(f"[bench] {e}\n")
                state["safe_failed"] = str(e)
            finally:
                if sim is not None:
                    sim.close()
                    sim = None
        # STAGED since round 6: with --driver tune the start-up timing runs over the torch-driven candidates ONLY; the winner's full,
        # validated measurement then becomes the fallback line (it replaces the all-gather one if it is faster); only then does the
        # library's own C loop get its turn, as ONE challenger on the winning protocol (one extra RCCL communicator, not four).  A C loop
        # that hangs or fails on a node therefore costs its deadline and the C figure — never the torch-driven result.
        staged = args.driver == "tune"
        first_driver = "torch" if staged else args.driver
        phase["now"] = f"creating the sharded simulation (protocol {args.protocol}, driver {first_driver})"
        sim = make_sim("allgather" if args.no_symmetry else args.protocol, first_driver)
        state["tuning"] = getattr(sim, "tuning", None)        # what the start-up timing found, for the fallback line too
        if rank == 0:
            sys.stderr.write(f"[bench] running with protocol {getattr(sim, 'protocol', '?')}, step loop {getattr(sim, 'driver', '?')}: {driver_reason(sim)}\n")
            sys.stderr.flush()

        def measure(sim, label):
            m = timed_region(sim, args, world, rank, barrier, device_sync, reference, phase, label, check)
            sustained = None
            if not args.no_sustained:
                phase["now"] = f"{label}sustained stretch"
                barrier()
                sustained = sustained_rate(sim.advance, sim.wait, DT, m["elapsed"] / max(1, args.steps) * 1e3, sampler_period=0.1, world=world)
                barrier()
                sustained["seconds"] = reduce_max_over_ranks(sustained["seconds"], world)
                sustained["ms_per_step"] = sustained["seconds"] / sustained["steps"] * 1e3
            line = make_line(args, n, world, sim, m, sustained)
            if safe_m is not None:
                line["config"]["safe_first"] = {"protocol": "allgather", "driver": state.get("safe_driver", "torch"), "ms_per_step": safe_m["elapsed"] / args.steps * 1e3,
                                                "value": float(n) * float(n) * args.steps / safe_m["elapsed"],
                                                "parity_check": {k: safe_m["parity"][k] for k in ("max_rel_pos", "max_rel_vel", "ok")} if safe_m["parity"] else None,
                                                "note": "north_star's plain all-gather protocol, torch-driven: measured first (same W + K steps), kept as the "
                                                        "line to print if the faster candidates could not be measured"}
            elif state.get("safe_failed"):
                line["config"]["safe_first"] = {"failed": state["safe_failed"]}
            line["fallback"] = {"used": False}
            return m, line

        t_stage = time.perf_counter()
        m, line = measure(sim, "")
        t_stage = time.perf_counter() - t_stage
        if m["parity"] is not None and not m["parity"]["ok"]:
            state["wrong_line"] = line      # measured, but of a trajectory that left the tolerance during the timed steps
            raise ParityError(f"{sim.protocol} protocol, {sim.driver} loop, after the timed region", m["parity"]["after_timed_region"])
        # ---- the C-loop challenger (north_star: "host code stays in C") ------------------------------------------------------------
        if staged and getattr(sim, "c_loop_available", False):
            upgraded = state["safe_line"] is None or line["value"] >= state["safe_line"]["value"]
            if upgraded:
                state["safe_line"] = line                              # a full, validated measurement: the better thing to fall back to
            if rank == 0:
                sys.stderr.write(f"[bench] torch-driven {sim.protocol}: {line['ms_per_step']:.3f} ms/step, validated; "
                                 + ("now the fallback line" if upgraded else "the safe-first line stays the fallback (it was faster)")
                                 + ".  Trying the library's C loop on the same protocol\n")
                sys.stderr.flush()
            proto, extra = sim.protocol, dict(getattr(sim, "chosen_extra", None) or {})
            sim.close()
            sim = None
            challenger = {"protocol": proto, "driver": "c"}
            # its own deadline: forming one more RCCL communicator (bounded by --candidate-deadline inside the library's host too) plus
            # the same measurement the torch-driven loop just finished in t_stage seconds, with room; on expiry the fallback line goes out
            challenger_deadline = (args.candidate_deadline + 4.0 * t_stage + 20.0) if args.candidate_deadline > 0 else 0.0
            try:
                with Watchdog(challenger_deadline, f"running the C-loop challenger ({proto} protocol)", report=lambda: f"phase: {phase['now']}", rank=rank):
                    phase["now"] = f"C-loop challenger: creating the {proto} configuration under the library's own RCCL loop"
                    sim = make_sim(proto, "c", extra)
                    m3, line3 = measure(sim, "C-loop challenger: ")
                ok3 = m3["parity"] is None or bool(m3["parity"]["ok"])
                challenger.update({"ms_per_step": line3["ms_per_step"], "value": line3["value"],
                                   "parity_check": ({k: m3["parity"][k] for k in ("max_rel_pos", "max_rel_vel", "ok")} if m3["parity"] else None)})
                # near-ties (1 %) go to the C loop — north_star's host — which has passed the same self-check; it loses only where it is slower
                if ok3 and line3["value"] >= 0.99 * line["value"]:
                    torch_fig = {"protocol": line["config"]["protocol"], "driver": "torch", "ms_per_step": line["ms_per_step"], "value": line["value"]}
                    line = line3
                    line["config"]["torch_driven"] = torch_fig
                    line["config"]["protocol_tuning"] = state.get("tuning")       # the start-up timing that chose this protocol (torch-driven candidates)
                    challenger["won"] = True
                else:
                    challenger["won"] = False
                    challenger["why_not"] = "left the tolerance of the self-check" if not ok3 else "slower than the torch-driven loop by more than 1 %"
            except ParityError as e:           # the verdict is broadcast: every rank is here together
                challenger.update({"won": False, "why_not": str(e)})
                if rank == 0:
                    sys.stderr.write(f"[bench] C-loop challenger: {e}\n")
            except RuntimeError as e:          # "... could not be formed on every rank" / "... not the same sharded plan": raised by EVERY rank
                if "on every rank" not in str(e) and "did not build the same sharded plan" not in str(e):     # together (agreed inside the host)
                    raise
                challenger.update({"won": False, "why_not": str(e)})
                if rank == 0:
                    sys.stderr.write(f"[bench] C-loop challenger: {e}\n")
            line["config"]["c_loop_challenger"] = challenger
            line["config"]["driver_choice"] = (f"{line['config']['driver']}: C loop {challenger.get('ms_per_step', float('nan')):.3f} ms/step vs torch-driven "
                                               f"{(line['config'].get('torch_driven') or line)['ms_per_step']:.3f} ms/step on the '{proto}' protocol, both full "
                                               f"measurements with the self-check" + ("" if challenger.get("won") else f"; C loop not taken: {challenger.get('why_not')}"))
        # The line printed is the FASTEST of the full, validated measurements.  The start-up timing orders its candidates on a dozen steps
        # each; if the configuration it chose then measures slower over W + K steps than the plain all-gather one did (a mis-ordered
        # near-tie, a protocol that degrades over a longer run), the plain one's line goes out — and says what was tuned and lost.
        first = state.get("first_line")
        if first is not None and first["value"] > line["value"]:
            chosen = dict(first)
            chosen["config"] = dict(first["config"])
            chosen["config"]["tuned_but_slower"] = {"protocol": line["config"].get("protocol"), "driver": line["config"].get("driver"),
                                                    "ms_per_step": line["ms_per_step"], "value": line["value"],
                                                    "c_loop_challenger": line["config"].get("c_loop_challenger")}
            chosen["config"]["safe_first"] = line["config"].get("safe_first")
            chosen["config"]["protocol_tuning"] = state.get("tuning")
            chosen["fallback"] = {"used": False}
            if rank == 0:
                sys.stderr.write(f"[bench] the tuned configuration ({line['config'].get('protocol')}, {line['config'].get('driver')} loop) measured "
                                 f"{line['ms_per_step']:.3f} ms/step, the all-gather one {first['ms_per_step']:.3f}: printing the faster, validated line\n")
            line = chosen
        print_once(line)
    except ParityError as e:
        if rank == 0:
            sys.stderr.write(f"[bench] {e}\n")
        if state["safe_line"] is not None:
            line = dict(state["safe_line"])
            line["fallback"] = {"used": True, "why": str(e), "phase": phase["now"], "protocol_tuning": state.get("tuning")}
            print_once(line)
        else:
            if rank == 0:
                sys.stderr.write(json.dumps({"parity_check": e.result, "what": str(e)}) + "\n")
            if state.get("wrong_line") is not None:     # nothing valid to fall back to: the line goes out with parity_check.ok = false
                print_once(state["wrong_line"])         # and the run still ends non-zero
            status = EXIT_PARITY
    except Exception as e:      # noqa: BLE001
        if state["safe_line"] is None:
            raise
        sys.stderr.write(f"[bench] rank {rank}: {type(e).__name__}: {e} (phase: {phase['now']}); printing the safe-first line\n")
        line = dict(state["safe_line"])
        line["fallback"] = {"used": True, "why": f"{type(e).__name__}: {e}", "phase": phase["now"], "protocol_tuning": state.get("tuning")}
        print_once(line)
    finally:
        phase["now"] = "closing"
        try:
            if sim is not None:
                sim.close()
            if reference is not None:
                reference.close()
        finally:
            stop_signals.__exit__(None, None, None)
            main_watchdog.__exit__(None, None, None)
            Watchdog.last_resort = None
    return status


# ---------------------------------------------------------------------------------------------------------------------
# `python bench.py --gpus N` typed without a launcher (N > 1)
# ---------------------------------------------------------------------------------------------------------------------
def torchrun_available() -> bool:
    """Is torch.distributed.run there to start the ranks with?  Looked up ON DISK: the launching process imports neither
    torch nor anything else that could touch a GPU (find_spec of a top-level name does not import it)."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return False
    return bool(spec and spec.submodule_search_locations and
                any((Path(d) / "distributed" / "run.py").exists() for d in spec.submodule_search_locations))


def _free_port() -> int:
    """A TCP port for a rendezvous on 127.0.0.1, chosen BELOW the kernel's ephemeral range (ip_local_port_range, 32768-60999 here).
    The usual bind(0)-close-reuse pattern hands back an ephemeral port, and between the close and the rendezvous server's own bind
    the kernel may give that very port to an outgoing connection — including the waiting rank's own connect() retries, which can
    "self-connect" to a local port nobody listens on yet; the server then dies with EADDRINUSE (seen once in round 6 on a GPU box).
    Ports below the range are only ever taken by explicit binds: probing one and using it a second later is safe in practice."""
    import random
    import socket
    lo, hi = 20000, 32000
    try:
        first = int(open("/proc/sys/net/ipv4/ip_local_port_range").read().split()[0])
        hi = min(hi, first - 1) if first > lo + 1000 else hi
    except (OSError, ValueError, IndexError):
        pass
    rng = random.SystemRandom()
    for _ in range(200):
        port = rng.randrange(lo, hi)
        with socket.socket() as s:
            try:
                s.bind(("127.0.0.1", port))
            except OSError:
                continue
            return port
    with socket.socket() as s:              # nothing free down there (never seen): the old way
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _die_with_parent():
    """preexec of the launcher child (Linux): SIGTERM when the launching process dies — however it dies, a SIGKILL by the
    driver's own time limit included — so that no rank is left running on a GPU behind a dead bench.py."""
    try:
        import ctypes
        ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, signal.SIGTERM, 0, 0, 0)      # PR_SET_PDEATHSIG
    except Exception:   # noqa: BLE001 - a convenience, never a reason not to start
        pass


def ranks_command(argv, gpus, port, script=None, python=None):
    """The child command: the driver's own N > 1 form (one process per GPU over torch.distributed.run, rendezvous on
    127.0.0.1) with THIS invocation's arguments.  `--n X` is rewritten to `--nbodies X` (torch.distributed.run's parser
    finds a bare --n ambiguous)."""
    fwd = []
    for tok in argv:
        fwd.append("--nbodies" if tok == "--n" else ("--nbodies=" + tok[4:]) if tok.startswith("--n=") else tok)
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), str(script or Path(__file__).resolve()), *fwd]


LAST_LAUNCH = {"lines": 0, "stopped": False, "seconds": 0.0}      # what the last launch_ranks call saw (main() decides on a second attempt from it)


def launch_ranks(argv, gpus, deadline_s=0.0, out=None, command=None) -> int:
    """Start the N ranks as CHILD processes and wait for them; returns the status this process should end with.
    * the parent never imports torch or the library: nothing here touches a GPU, and nothing is exec'ed in place;
    * rank 0's stdout (the ONE JSON line) is relayed line by line as it arrives, stderr is inherited;
    * SIGTERM / SIGINT received here are forwarded to the launcher child, which passes them to the ranks — rank 0 then
      prints the safe-first line it already holds (run_sharded) — and the relay goes on until the children are gone;
    * `deadline_s` > 0: the children's own deadlines (--deadline) are the first line of defence; if the launcher child is
      still there 60 s later it is terminated (then killed) from here.
    Status: the launcher child's — except that a run which relayed a bench line and was stopped by a signal ends 0 (the
    line IS the result; the ranks that held no line died of the signal, which torch.distributed.run reports as a failure)."""
    out = out or sys.stdout
    t_start = time.time()
    LAST_LAUNCH.update({"lines": 0, "stopped": False, "seconds": 0.0})
    cmd = command or ranks_command(argv, gpus, _free_port())
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # RCCL shares device buffers through dmabuf IPC on this driver stack
    env.setdefault("OMP_NUM_THREADS", "1")                 # what torch.distributed.run would set itself (and warn about)
    sys.stderr.write(f"[bench] --gpus {gpus} without a launcher: starting {gpus} ranks as child processes: {' '.join(cmd)}\n")
    sys.stderr.flush()
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, text=True, bufsize=1, env=env, preexec_fn=_die_with_parent)
    stopped = {"by": None}

    def forward(signum, _frame):
        stopped["by"] = signum
        try:
            proc.send_signal(signum)
        except (ProcessLookupError, OSError):
            pass

    previous = {}
    try:
        for sg in (signal.SIGTERM, signal.SIGINT):
            previous[sg] = signal.signal(sg, forward)
    except ValueError:          # not the main thread (a test harness): no forwarding
        pass
    killer = None
    if deadline_s and deadline_s > 0:
        import threading

        def expire():
            sys.stderr.write(f"[bench] the ranks are still running {deadline_s + 60:.0f} s after the start: terminating them\n")
            sys.stderr.flush()
            forward(signal.SIGTERM, None)
            try:
                proc.wait(timeout=40)
            except subprocess.TimeoutExpired:
                proc.kill()
        killer = threading.Timer(deadline_s + 60.0, expire)
        killer.daemon = True
        killer.start()
    lines = 0
    try:
        while True:
            try:
                text = proc.stdout.readline()
            except InterruptedError:
                continue
            if not text:
                break
            if text.lstrip().startswith("{"):        # rank 0's line: the ONLY thing this process puts on stdout
                out.write(text)
                out.flush()
                lines += 1
            else:                                    # whatever else a rank or a library wrote to its stdout (gloo's connection notes ...)
                sys.stderr.write(text)
                sys.stderr.flush()
        rc = proc.wait()
    finally:
        if killer is not None:
            killer.cancel()
        for sg, h in previous.items():
            signal.signal(sg, h)
    LAST_LAUNCH.update({"lines": lines, "stopped": stopped["by"] is not None, "seconds": time.time() - t_start})
    if stopped["by"] is not None and lines:
        return 0
    return rc if rc >= 0 else 128 - rc


def run_one_process(args) -> int:
    """`--gpus N` in ONE process (no torch.distributed.run, or --one-process): N sharded handles, one per device, bound by the
    library's own RCCL communicator (nb_comm_create_all) and stepped by its C loop; the same flow as a launched run
    (run_sharded: safe-first all-gather measurement, self-checks against one unsharded handle, the N > 1 line).  No torch here."""
    import nbodysim_amd as nb
    from nbodysim_amd.local_ranks import LocalRanksSimulation

    world, n = args.gpus, args.n
    have = int(nb.load().nb_device_count())
    if have < 1:
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    if have < world and not args.share_gpu:
        raise SystemExit(f"--gpus {world} but this node shows {have} device(s) (--share-gpu rehearses the ranks on the devices there are)")
    devices = [r % have for r in range(world)]
    ic = nb.plummer_2d(n, SEED) if args.dims == 2 else nb.plummer_3d(n, SEED)
    scaling = {"off": False, "on": True, "measured": False}[args.mass_scaling]     # "measured" is for unsharded handles: the sharded run and its checker keep the same body
    physics = dict(eps=EPS, precision=args.precision, rsqrt=args.rsqrt, dims=args.dims, uniform_mass=not args.general_mass, mass_scaling=scaling)
    live = []

    def make_sim(protocol, _driver, _extra=None):
        sim = LocalRanksSimulation(ic, world, devices, protocol="allgather" if args.no_symmetry else protocol, tune_dt=DT,
                                   sym_chunks_per_item=args.chunks_per_item, **physics)
        live[:] = [sim]
        return sim

    def make_reference():
        return nb.Simulation(ic, device=devices[0], **physics)

    def device_sync():
        for sim in live:
            if sim.sims:
                sim.wait()

    args.backend = "rccl (one process)" if len(set(devices)) == world else "in-process exchange (shared GPU rehearsal)"
    return run_sharded(args, ic, n, world, 0, make_sim, make_reference, device_sync, lambda: None)


def main() -> None:
    args = parse_args()

    # RCCL shares device buffers between the ranks of a node through dmabuf IPC on this driver stack
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    # `python bench.py --gpus N` as typed (no launcher, N > 1).  Decided BEFORE torch, the library or any HIP call is
    # touched: the ranks are child processes of a parent that never sees a GPU — or, without torch.distributed.run, this
    # one process drives all N handles through the library's C loop.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.rehearse_sharded:
        if not args.one_process and torchrun_available():
            t_launch = time.time()
            rc = launch_ranks(sys.argv[1:], args.gpus, args.deadline)
            if rc != 0 and LAST_LAUNCH["lines"] == 0 and not LAST_LAUNCH["stopped"] and LAST_LAUNCH["seconds"] < 30.0:
                # the ranks were gone within seconds and never got as far as a line: a rendezvous that did not form (a port taken between
                # the probe and the bind) looks exactly like this; one more attempt, on another port, costs seconds
                sys.stderr.write(f"[bench] the ranks ended with status {rc} after {LAST_LAUNCH['seconds']:.0f} s without a line: one more attempt\n")
                sys.stderr.flush()
                rc = launch_ranks(sys.argv[1:], args.gpus, args.deadline)
            if rc != 0 and LAST_LAUNCH["lines"] == 0 and not LAST_LAUNCH["stopped"] and time.time() - t_launch < 450.0:
                # one process per GPU did not get as far as a line (a process group that would not form, IPC between the ranks refused ...):
                # the library's own loop needs neither — ONE process, peer access instead of IPC, ncclCommInitAll instead of a rendezvous.
                # This process has not touched a GPU yet; it does now (no exec involved).
                sys.stderr.write(f"[bench] the launched ranks ended with status {rc} and no line: falling back to ONE process driving all "
                                 f"{args.gpus} handles (nb_comm_create_all)\n")
                sys.stderr.flush()
                raise SystemExit(run_one_process(args))
            raise SystemExit(rc)
        if not args.one_process:
            sys.stderr.write("[bench] torch.distributed.run is not available: one process drives all ranks (nb_comm_create_all)\n")
        raise SystemExit(run_one_process(args))

    import torch

    import nbodysim_amd as nb

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--rehearse-sharded is the ONE-rank rehearsal of the multi-rank flow: use it with --gpus 1")
        args.gpus = world
    n = args.n
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    if args.share_gpu:
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)

    ic = nb.plummer_2d(n, SEED) if args.dims == 2 else nb.plummer_3d(n, SEED)   # every rank generates the same deterministic ICs
    scaling = {"off": False, "on": True, "measured": "measured"}[args.mass_scaling]

    rehearsal = args.rehearse_sharded and world == 1
    if world > 1 or rehearsal:
        import datetime

        if scaling == "measured":      # the upload-time measurement exists for unsharded handles only: a sharded run and its unsharded checker
            scaling = False            # must run the SAME pair body, so here it means the default, on both sides
            if rank == 0:
                sys.stderr.write("[bench] --mass-scaling measured applies to unsharded handles; this sharded run keeps both mass multiplies\n")

        import torch.distributed as dist

        from nbodysim_amd.dist import DistributedSimulation, Watchdog

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal and "MASTER_PORT" not in os.environ:        # plain `python bench.py --rehearse-sharded`: a process group of one
            os.environ["MASTER_PORT"] = str(_free_port())
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        # no wait without an end: the process group's own collective timeout (its watchdog aborts the process) and, around the
        # whole run, a wall-clock deadline that names the phase the rank was in (run_sharded)
        pg_timeout = datetime.timedelta(seconds=max(60.0, min(600.0, args.deadline or 600.0)))
        with Watchdog(min(args.deadline, 300.0) if args.deadline else 0.0, "forming the process group and its first collectives", rank=rank):
            if args.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=pg_timeout)
            else:
                dist.init_process_group("gloo", timeout=pg_timeout)
            # bring the communicator up (channels, RCCL kernels) before anything is tuned or timed, whatever --warmup is
            scratch = torch.zeros((world * 64, 2), dtype=torch.float32, device="cuda")
            dist.all_gather_into_tensor(scratch, scratch[rank * 64:(rank + 1) * 64])
            if args.backend == "nccl":
                dist.reduce_scatter_tensor(scratch[:64].clone(), scratch, op=dist.ReduceOp.SUM)
            dist.all_reduce(scratch, op=dist.ReduceOp.SUM)
            torch.cuda.synchronize()
        if args.no_symmetry and args.protocol not in ("tune", "allgather"):
            raise SystemExit("--no-symmetry with several ranks means the all-gather protocol")

        def make_sim(protocol, driver, extra=None):
            kw = dict(uniform_mass=not args.general_mass, dims=args.dims, sym_chunks_per_item=args.chunks_per_item, mass_scaling=scaling)
            kw.update(extra or {})                     # what the start-up timing chose beside the protocol (the late-item share)
            return DistributedSimulation(ic, eps=EPS, precision=args.precision, rsqrt=args.rsqrt, device_index=local_rank,
                                         protocol=protocol, tune_dt=DT, driver=driver if args.backend == "nccl" else "torch",
                                         deadline_s=args.candidate_deadline, verify=not args.no_parity_check,
                                         rehearse_single_rank=rehearsal, **kw)

        def make_reference():
            return nb.Simulation(ic, eps=EPS, precision=args.precision, rsqrt=args.rsqrt, device=local_rank, dims=args.dims,
                                 uniform_mass=not args.general_mass, mass_scaling=scaling)

        status = run_sharded(args, ic, n, world, rank, make_sim, make_reference, torch.cuda.synchronize, dist.barrier)
        sys.stdout.flush()
        # the line is out; tearing the process group down must not turn a finished run into a hang (a peer that left early)
        with Watchdog(60.0, "destroying the process group after the run", rank=rank, exit_fn=lambda _code: os._exit(status)):
            dist.destroy_process_group()
        if status:
            raise SystemExit(status)
        return

    # ---- one GPU ------------------------------------------------------------------------------------------------------
    sim = nb.Simulation(ic, eps=EPS, precision=args.precision, rsqrt=args.rsqrt, device=local_rank, dims=args.dims,
                        symmetry=not args.no_symmetry, uniform_mass=not args.general_mass, sym_chunks_per_item=args.chunks_per_item,
                        mass_scaling=scaling)
    m = timed_region(sim, args, 1, 0, lambda: None, torch.cuda.synchronize)

    # the settled rate, AFTER the timed region (never part of `value`)
    sustained = None
    if not args.no_sustained:
        events = not args.no_kernel_events       # per-launch events (nb_step bounds their number)
        if events:
            sim.profile(True)
        sustained = sustained_rate(sim.advance, sim.wait, DT, m["elapsed"] / max(1, args.steps) * 1e3, sampler_period=0.02, world=1)
        if events:
            sms, sl = sim.profile_read()
            sim.profile(False)
            if sl:
                sustained["avg_launch_ms"] = sms / sl

    # secondary figures, outside the timed region and not part of `value`
    secondary = {}
    if not args.no_secondary and not args.no_kernel_events:
        def second(warm=2, steps=max(4, args.steps // 2), **kw):
            kw.setdefault("symmetry", not args.no_symmetry)
            with nb.Simulation(ic, eps=EPS, precision=args.precision, rsqrt=args.rsqrt, device=local_rank, dims=args.dims, **kw) as g:
                g.advance(warm, DT)
                g.wait()
                g.profile(True)
                g.advance(steps, DT)
                g.wait()
                gms, gl = g.profile_read()
                desc = g.describe()
            check = [kv.split("=", 1)[1] for kv in desc.replace("|", " ").split() if kv.startswith("mass_scaling_check=")]
            return {"avg_launch_ms": gms / gl, "launches": gl, "mass_scaled": "mass_scaled=1" in desc,
                    "mass_scaling_check": float(check[0]) if check and float(check[0]) >= 0 else None,
                    "kernel_instantiation": kernel_instantiation(desc, args.precision, args.dims, args.rsqrt)}
        if not args.general_mass:
            # the same kernel without the equal-mass specialisation (12 + 2 instead of 10 + 2 instructions per body): what a system with
            # individual masses — the reference's own bodies, Simulation.hpp:565-577 — gets
            secondary["general"] = second(uniform_mass=False)                        # the default with individual masses: both per-pair mass multiplies (12 + 2)
            secondary["scaled"] = second(uniform_mass=False, mass_scaling=True)      # NB_FLAG_MASS_SCALING: masses folded into the pair geometry (11 + 2)
            secondary["measured"] = second(uniform_mass=False, mass_scaling="measured")   # opt-in: the library measures at upload which of the two this data gets
        if not args.no_symmetry:
            # north_star's literal kernel design — one-sided, j-particles staged through LDS tiles of 256 — on the same workload
            secondary["lds_tiled"] = second(steps=max(4, args.steps // 4), symmetry=False, uniform_mass=not args.general_mass)

    if not args.no_live_pmc and not args.no_secondary and not args.no_kernel_events:
        # this run's own counters, on this box: after everything that is timed, before the CPU baseline
        secondary["live_pmc"] = live_pmc(args, kernel_instantiation(sim.describe(), args.precision, args.dims, args.rsqrt), sim.sym_info()["items"])
    line = make_line(args, n, 1, sim, m, sustained, secondary)
    value = line["value"]
    if not args.no_secondary and args.dims == 2:
        line["frame_ms"] = frame_costs(lambda: nb.Simulation(ic, eps=EPS, precision=args.precision, rsqrt=args.rsqrt, device=local_rank, dims=args.dims,
                                                             symmetry=not args.no_symmetry, uniform_mass=not args.general_mass, mass_scaling=scaling),
                                       line["ms_per_step"], DT)
    if not args.no_cpu_baseline and args.dims == 2:
        line["cpu_baseline"] = cpu_baseline(ic, n)
        line["cpu_baseline"]["gpu_over_cpu"] = value / line["cpu_baseline"]["value"]
        if "port" in line["cpu_baseline"]:
            line["cpu_baseline"]["port"]["gpu_over_cpu"] = value / line["cpu_baseline"]["port"]["value"]
        # context only: the reference's PRODUCTION path is Barnes-Hut, not O(N^2)
        line["cpu_baseline"]["context"] = ("reference Simulation::step() (Barnes-Hut theta=1 + collide) ran at 2.31 steps/s at "
                                           "N=262144 in the survey container (8 vCPU Xeon), not on this box; never mixed into pair interactions/s")
    print(json.dumps(line), flush=True)
    sim.close()


if __name__ == "__main__":
    main()
