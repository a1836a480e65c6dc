"""Test infrastructure (CPU): a stand-in for ``nbodysim_amd.dist.DistributedSimulation`` whose force engine is the
ORACLE (oracle/nbo.py) and whose exchange is the real one (``exchange_positions`` over gloo).  It lets bench.py's
multi-rank flow — safe-first measurement, start-up timing with a failing / hanging / wrong candidate, the self-check
against an unsharded trajectory, the fallback line — run with two ranks in this container, where there is no GPU.
Nothing under nbodysim_amd/ or bench.py imports this; the product path has no CPU engine.

Faults are injected by name:  fault = (kind, where, rank)
    kind   "raise"    that rank raises after the candidate's collectives completed (a failed divergence check, a HIP
                      error surfacing at wait())
           "hang"     that rank never comes back from the candidate
           "corrupt"  that rank's block drifts away from the true trajectory (a mis-ordered exchange)
           "corrupt_late"  the same, but only from the third step on: the self-check after the warm-up (2 steps) passes,
                      the one after the timed steps does not
    Several faults at once: a tuple of such triples.
    where  the configuration it hits: "allgather/torch" (the safe-first one), "tune:<candidate>" (a trial of the start-up
           timing), "final" (the tuned configuration's own run) or "c-loop" (the C-loop challenger of bench.run_sharded's
           stage 2b; the stand-in offers one when NB_STANDIN_C_LOOP=1)
"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))

EPS, DT = 0.05, 1e-3


def load_flat(n):
    return np.load(ROOT / "tests" / "golden" / "ic_plummer_1024.npy")[:n].copy()


class StandinInner:
    """The few calls bench.make_line makes on the per-rank handle."""

    def profile(self, on=True):
        pass

    def profile_read(self, reset=True):
        return 0.0, 0

    def sym_info(self):
        return {"enabled": 0, "items": 0, "chunks_per_item": 0, "tiles": 0, "units_local": 0, "units_cross": 0, "units_late": 0,
                "slab_s_bytes": 0, "slab_r_bytes": 0, "tile_particles": 0}

    def describe(self):
        return "stand-in engine (oracle on the CPU) uniform_mass=1"


class Unsharded:
    """make_reference(): the whole system advanced by the oracle alone."""

    def __init__(self, flat):
        import nbo
        self.nbo, self.st = nbo, nbo.state_from_flat(flat)

    def advance(self, steps, dt):
        if steps:
            self.st = self.nbo.step_f32(self.st, EPS, dt, steps, self.nbo.RSQRT_QUAKE)

    def sync(self):
        out = np.zeros(len(self.st["x"]), dtype=[("pos", np.float32, 2), ("vel", np.float32, 2)])
        out["pos"] = np.stack([self.st["x"], self.st["y"]], 1)
        out["vel"] = np.stack([self.st["vx"], self.st["vy"]], 1)
        return out

    def close(self):
        pass


class StandinSharded:
    def __init__(self, flat, protocol, driver, fault=None, where=""):
        import nbo
        import torch
        import torch.distributed as dist

        from nbodysim_amd.dist import ShardPlan

        self.nbo, self.torch, self.dist = nbo, torch, dist
        nbo.set_threads(2)
        self.flat = flat
        self.plan = ShardPlan(flat.shape[0], dist.get_world_size(), dist.get_rank())
        self.fault, self.where = fault, where
        self.tuning = None
        self.sim = StandinInner()
        import os
        self.c_loop_available = os.environ.get("NB_STANDIN_C_LOOP") == "1"      # pretend the library's own loop can challenge (bench.run_sharded, stage 2b)
        self.chosen_extra = {}
        self._slow_final = os.environ.get("NB_STANDIN_SLOW_FINAL") == "1"
        if driver == "c":
            self.where = "c-loop"
        if protocol == "tune" or driver == "tune":
            protocol, driver = self._tune()
            self.where = "final"
        self.protocol, self.driver = ("allgather" if protocol == "auto" else protocol), driver
        self._reset()

    def _hit(self, kind):
        faults = self.fault if (self.fault and isinstance(self.fault[0], (tuple, list))) else ([self.fault] if self.fault else [])
        return any(f[0] == kind and f[1] == self.where and f[2] == self.plan.rank for f in faults)

    def _reset(self):
        st = self.nbo.state_from_flat(self.flat)
        p = self.plan
        self.m = st["m"]
        self.pos = [self.torch.from_numpy(np.stack([st["x"], st["y"]], 1).copy()) for _ in range(2)]
        self.cur, self.pending, self.steps_done = 0, None, 0
        self.vx, self.vy = st["vx"][p.i_begin:p.i_end].copy(), st["vy"][p.i_begin:p.i_end].copy()

    # -- the start-up timing: the REAL time_candidates over stand-in trials ---------------------------------------------
    def _tune(self):
        from nbodysim_amd.dist import after_collectives, compare_with_unsharded, time_candidates

        steps, names = 3, ["allgather", "allreduce", "symmetric"]
        ref = {}

        def unsharded():
            if "rows" not in ref:
                u = Unsharded(self.flat)
                u.advance(steps, DT)
                ref["rows"] = rows_of(u.sync())
            return ref["rows"]

        def run_one(name, local):
            if name == "symmetric":
                return float("inf")                       # "not eligible" on every rank
            trial = StandinSharded(self.flat, name, "torch", self.fault, "tune:" + name)
            trial.advance(steps, DT)
            trial.wait()
            self.dist.barrier()
            if trial._hit("hang"):
                time.sleep(3600)
            v = compare_with_unsharded(trial.owned_rows(), trial.plan, unsharded, steps)
            if trial._hit("raise"):
                raise after_collectives(RuntimeError("injected: this rank's trial of '%s' failed" % name))     # the trial's collectives are through
            return ({"allgather": 3.0, "allreduce": 1.0}[name]) if v["ok"] else float("inf")

        failed = {}
        import os
        best, job = time_candidates(names, run_one, None, float(os.environ.get("NB_STANDIN_CANDIDATE_DEADLINE", "6.0")), self.plan.rank, prefer=("symmetric", "allreduce", "allgather"),
                                    log=lambda m: print(m, file=sys.stderr, flush=True), failed=failed)
        self.tuning = {"chosen": best, "ms_per_step": {k: (v if np.isfinite(v) else None) for k, v in job.items()}, "failed": failed}
        return best, "torch"

    # -- DistributedSimulation's surface, as bench.py uses it ------------------------------------------------------------
    def advance(self, nsteps, dt=DT):
        p, nbo = self.plan, self.nbo
        lo, hi = p.i_begin, p.i_end
        dt32 = np.float32(dt)
        for _ in range(nsteps):
            if self.pending is not None:
                self.pending.wait()
                self.pending = None
            full = self.pos[self.cur].numpy()
            s = {"x": np.ascontiguousarray(full[:p.n, 0]), "y": np.ascontiguousarray(full[:p.n, 1]), "m": self.m}
            ax, ay = nbo.accel_f32(s, EPS, nbo.RSQRT_QUAKE, lo, hi)
            self.vx = (self.vx + ax[lo:hi] * dt32).astype(np.float32)
            self.vy = (self.vy + ay[lo:hi] * dt32).astype(np.float32)
            nxt = self.pos[self.cur ^ 1]
            self.steps_done += 1
            if self.where == "allgather/torch":
                time.sleep(0.02)       # the plain protocol is the slow one, as on hardware: which line is the better fallback is then not this CPU's noise
            if self.where == "final" and self._slow_final:
                time.sleep(0.08)       # ... unless a test wants the tuned configuration to lose its full measurement (NB_STANDIN_SLOW_FINAL=1)
            scale = np.float32(1.001) if (self._hit("corrupt") or (self._hit("corrupt_late") and self.steps_done > 2)) else np.float32(1.0)
            nxt[lo:hi, 0] = self.torch.from_numpy(((full[lo:hi, 0] + self.vx * dt32) * scale).astype(np.float32))
            nxt[lo:hi, 1] = self.torch.from_numpy(((full[lo:hi, 1] + self.vy * dt32) * scale).astype(np.float32))
            self.cur ^= 1
            self.pending = self._exchange()
        if self._hit("hang") and self.where in ("final", "c-loop"):
            time.sleep(3600)

    def _exchange(self):
        from nbodysim_amd.dist import exchange_positions
        return exchange_positions(self.pos[self.cur], self.plan, async_op=True)

    def wait(self):
        if self.pending is not None:
            self.pending.wait()
            self.pending = None

    def owned_rows(self):
        self.wait()
        p = self.plan
        full = self.pos[self.cur].numpy()
        return np.concatenate([full[p.i_begin:p.i_end].astype(np.float64), np.stack([self.vx, self.vy], 1).astype(np.float64)], axis=1)

    def energy(self):
        self.wait()
        p = self.plan
        k = 0.5 * float(np.sum(self.m[p.i_begin:p.i_end].astype(np.float64) * (self.vx.astype(np.float64) ** 2 + self.vy.astype(np.float64) ** 2)))
        t = self.torch.tensor([k, -1.0 / self.plan.world], dtype=self.torch.float64)     # the potential is not what these tests are about
        self.dist.all_reduce(t)
        return float(t[0]), float(t[1])

    def profile_phases(self, on=True):
        pass

    def phase_report(self):
        return {"local": 0.0, "ag_wait": 0.0, "remote_finish": 0.0, "stream_total": 0.0, "host_enqueue": 0.0, "steps": 0}

    def close(self):
        self.wait()


def rows_of(rec):
    return np.concatenate([rec["pos"].astype(np.float64), rec["vel"].astype(np.float64)], axis=1)


def bench_worker(rank, world, port, n, fault, extra_args=()):
    """Child process: bench.run_sharded over gloo with the stand-in engine; prints what run_sharded prints, exits with its status."""
    import importlib.util
    import os

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    import datetime

    import torch.distributed as dist

    spec = importlib.util.spec_from_file_location("bench_under_test", ROOT / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    sys.modules["bench_under_test"] = bench
    spec.loader.exec_module(bench)
    bench.DT = DT
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    args = bench.parse_args(["--gpus", str(world), "--steps", "3", "--warmup", "2", "--nbodies", str(n), "--backend", "gloo", "--no-sustained",
                             "--no-kernel-events", "--deadline", "60", *extra_args])
    flat = load_flat(n)
    status = bench.run_sharded(args, None, n, world, rank,
                               make_sim=lambda protocol, driver, extra=None: StandinSharded(flat, protocol, driver, fault,
                                                                                            "allgather/torch" if (protocol, driver) == ("allgather", "torch") else ""),
                               make_reference=lambda: Unsharded(flat), device_sync=lambda: None, barrier=dist.barrier)
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(status)          # the hang cases leave a rank asleep inside a collective: no orderly teardown
