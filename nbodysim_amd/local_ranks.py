"""ONE process drives every rank of a sharded run: one ``nb_sim`` handle per device, bound by the
library's own RCCL communicator (``nb_comm_create_all`` -> ncclCommInitAll) and stepped by its C loop (``nb_comm_step``:
force / kick / drift on each handle's compute stream, the collectives on a communication stream, HIP events between
the two, one foreign call for any number of steps).  No torch, no process group, no launcher: what ``bench.py --gpus N``
falls back to where ``torch.distributed.run`` is missing, and what a plain-C host does (``host/nbody_main.c -shards P
-rccl``).  Replaces the reference's only fan-out, ``std::async`` over i-chunks (``Simulation.hpp:180-213``).

Where the handles share a device (a rehearsal on a one-GPU box: RCCL takes one rank per device) the same split-step
calls are driven from here with the library's in-process exchange (``nb_exchange_*``: peer copies and ordered sums) — the
same kernels, the same pair split, no transport.

``LocalRanksSimulation`` has the surface ``bench.run_sharded`` uses of ``dist.DistributedSimulation`` (advance / wait /
owned_rows / energy / plan / sim / protocol / driver / phase_report), with every rank's block in this one process.
"""
from __future__ import annotations

import ctypes as C
import time
from typing import Optional, Sequence

import numpy as np

from . import _lib as L
from .comm import Comm
from .dist import ShardPlan, state_rows
from .simulation import Simulation

_PROTOCOL_NAMES = {L.NB_SHARD_ALLGATHER: "allgather", L.NB_SHARD_SYMMETRIC: "symmetric", L.NB_SHARD_ALLREDUCE: "allreduce"}


class LocalRanksSimulation:
    """``world`` sharded handles in this process.

    devices    HIP device ordinal of each rank (default 0 .. world-1).  All distinct: RCCL through ``Comm.all``
               (``transport == "rccl"``); otherwise the in-process exchange (``"in-process"``).
    protocol   "allgather" (north_star's: one-sided kernels, one all-gather per step), "symmetric", "allreduce", "auto"
               (the library's choice: symmetric where the system is eligible), or "tune": a few steps of each of the
               three on scratch handles, the fastest kept (``tuning`` records the timings).
    Other keyword arguments go to ``Simulation`` (eps, precision, rsqrt, dims, uniform_mass, mass_scaling ...).
    """

    def __init__(self, bodies: np.ndarray, world: int, devices: Optional[Sequence[int]] = None, protocol: str = "auto",
                 tune_steps: int = 8, tune_dt: float = 1e-3, **sim_kwargs):
        if world < 2:
            raise ValueError("LocalRanksSimulation is for two or more ranks; one rank is a plain Simulation")
        if protocol not in ("auto", "tune", "allgather", "symmetric", "allreduce"):
            raise ValueError(f"unknown protocol {protocol!r}")
        self.world = int(world)
        self.devices = list(devices) if devices is not None else list(range(self.world))
        if len(self.devices) != self.world:
            raise ValueError("one device ordinal per rank")
        self.transport = "rccl" if len(set(self.devices)) == self.world else "in-process"
        self.n = int(bodies.shape[0])
        self.plan = ShardPlan(self.n, 1, 0)          # as the self-check sees it: this process holds every block
        self.blocks = [ShardPlan(self.n, self.world, r) for r in range(self.world)]
        self.owned_per_rank = self.blocks[0].i_count
        self._args = dict(sim_kwargs)
        self._lib = L.load()
        self.sims: list = []
        self.comm: Optional[Comm] = None
        self.tuning: Optional[dict] = None
        self._host_enqueue_s, self._host_steps, self._phase_on = 0.0, 0, False
        if protocol == "tune":
            protocol = self._tune(bodies, tune_steps, tune_dt)
        self._create(bodies, protocol)

    # -- construction -------------------------------------------------------------------------------------------------
    def _create(self, bodies: np.ndarray, protocol: str) -> None:
        self.close()
        kw = dict(self._args)
        if protocol == "allgather":
            kw["symmetry"] = False
        replicated = protocol == "allreduce"
        try:
            for r, b in enumerate(self.blocks):
                self.sims.append(Simulation(bodies, device=self.devices[r], i_begin=0 if replicated else b.i_begin,
                                            i_count=self.n if replicated else b.i_count, shard_rank=r, shard_world=self.world,
                                            shard_allreduce=replicated, **kw))
            protos = {s.shard_protocol for s in self.sims}
            if len(protos) != 1:
                raise RuntimeError(f"the ranks did not build the same sharded plan (protocols {sorted(protos)})")
            self._proto = protos.pop()
            self.protocol = _PROTOCOL_NAMES[self._proto]
            if protocol in ("symmetric", "allreduce") and self.protocol != protocol:
                raise RuntimeError(f"protocol='{protocol}' requested but the system is not eligible "
                                   "(needs eps > 0, tiled sum, blocks of whole 2048-particle tiles, n/world >= 4096)")
            self._arr = (C.c_void_p * self.world)(*[s._h for s in self.sims])
            if self.transport == "rccl":
                self.comm = Comm.all(self.sims)
        except Exception:
            self.close()
            raise
        self.sim = self.sims[0]                       # rank 0's handle: plan figures, force-launch events, describe()
        self.symmetric = self._proto == L.NB_SHARD_SYMMETRIC
        self.replicated = self._proto == L.NB_SHARD_ALLREDUCE
        self.driver = ("c (one process, nb_comm_create_all)" if self.comm is not None else "in-process exchange (nb_exchange_*)")

    def _tune(self, bodies: np.ndarray, steps: int, dt: float) -> str:
        ms = {}
        for cand in ("allgather", "allreduce", "symmetric"):
            try:
                self._create(bodies, cand)
            except (RuntimeError, L.NBodyError) as e:
                if "not eligible" not in str(e):
                    raise
                ms[cand] = None
                continue
            self.advance(2, dt)
            self.wait()
            t0 = time.perf_counter()
            self.advance(steps, dt)
            self.wait()
            ms[cand] = (time.perf_counter() - t0) / steps * 1e3
            self.close()
        order = [k for k in ("symmetric", "allreduce", "allgather") if ms.get(k) is not None]
        best = order[0]
        for k in order[1:]:
            if ms[k] < 0.99 * ms[best]:
                best = k
        self.tuning = {"steps": steps, "ms_per_step": ms, "chosen": best, "failed": {},
                       "validation": {}, "validated_against": "the timed run's own self-check (no per-candidate validation in one-process mode)"}
        return best

    # -- stepping -----------------------------------------------------------------------------------------------------
    @property
    def frame(self) -> int:
        return self.sims[0].frame

    def _exchange(self, name: str) -> None:
        L.check(name, getattr(self._lib, name)(self._arr, self.world), self._lib)

    def advance(self, nsteps: int, dt: Optional[float] = None) -> None:
        t_host = time.perf_counter()
        if self.comm is not None:
            self.comm.step(nsteps, dt)
        else:
            for _ in range(nsteps):
                for s in self.sims:
                    s.step_begin(dt)
                if self.symmetric:
                    for s in self.sims:
                        s.step_mid()
                    self._exchange("nb_exchange_accelerations")
                elif self.replicated:
                    self._exchange("nb_exchange_allreduce")
                for s in self.sims:
                    s.step_finish()
                if not self.replicated:
                    self._exchange("nb_exchange_positions")
        self._host_enqueue_s += time.perf_counter() - t_host
        self._host_steps += nsteps

    def wait(self) -> None:
        if self.comm is not None:
            self.comm.wait()
        else:
            for s in self.sims:
                s.wait()

    # -- host views ---------------------------------------------------------------------------------------------------
    def sync(self) -> np.ndarray:
        """All n bodies, block by block (each handle copies back its own block; a replicated handle holds them all)."""
        self.wait()
        if self.comm is not None:
            self.comm.flush()
        if self.replicated:
            return self.sims[0].sync().copy()
        return np.concatenate([s.sync() for s in self.sims])

    def owned_rows(self) -> np.ndarray:
        return state_rows(self.sync())

    def replicas_identical(self) -> bool:
        if not self.replicated:
            return True
        first = self.sims[0].sync()
        return all(all(np.array_equal(first[f].view(np.uint32), s.sync()[f].view(np.uint32)) for f in ("pos", "vel")) for s in self.sims[1:])

    def energy(self) -> tuple:
        self.wait()
        if self.comm is not None:
            self.comm.flush()
        if self.replicated:
            return self.sims[0].energy()
        parts = [s.energy() for s in self.sims]
        return sum(k for k, _ in parts), sum(u for _, u in parts)

    # -- per-phase timing ---------------------------------------------------------------------------------------------
    def profile_phases(self, on: bool = True) -> None:
        self._phase_on = bool(on)
        self._host_enqueue_s, self._host_steps = 0.0, 0
        if self.comm is not None:
            self.comm.profile(on)

    def phase_report(self) -> dict:
        self.wait()
        out: dict = {}
        if self.comm is not None and self._phase_on:
            per = [self.comm.phases(k, reset=True) for k in range(self.world)]
            keys = [k for k in per[0] if k != "steps"]
            out = {k: max(p[k] for p in per) for k in keys}          # slowest handle per phase
            out["stream_total"] = sum(out.values())
            out["phase_steps"] = per[0]["steps"]
        out["host_enqueue"] = self._host_enqueue_s / max(1, self._host_steps) * 1e3
        out["steps"] = self._host_steps
        out["driver"] = self.driver
        return out

    def close(self) -> None:
        if self.comm is not None:
            self.comm.close()
            self.comm = None
        for s in self.sims:
            s.close()
        self.sims = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
