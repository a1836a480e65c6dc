"""Host-side mirror of the C-level RCCL exchange (``nb_comm_*`` in include/nbody.h).

The step loop of a sharded run lives in the library (``nb_comm_step``): force / kick / drift on each handle's
compute stream, ``ncclAllGather`` / ``ncclReduceScatter`` / ``ncclAllReduce`` on a communication stream, HIP events
between the two, no host synchronisation.  This module only forms the communicator and calls it — ``nb_comm_step(c,
dt, nsteps)`` is ONE foreign call for any number of steps, so the host cost per step is the C loop's, not Python's.

The reference has no distributed code; what this replaces is its in-process fan-out,
``std::async`` over i-chunks (``Simulation.hpp:180-213``).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

from . import _lib as L


def unique_id() -> bytes:
    """``nb_comm_unique_id``: the 128-byte RCCL id rank 0 creates and hands to the other ranks out of band."""
    buf = C.create_string_buffer(L.NB_COMM_ID_BYTES)
    L.check("nb_comm_unique_id", L.load().nb_comm_unique_id(buf))
    return buf.raw


def available() -> int:
    """``nb_comm_available``: the RCCL version the library's loop would use; raises ``NBodyError`` if the transport cannot
    be loaded in this process.  Ranks agree on this before any of them enters ``Comm.rank`` (ncclCommInitRank blocks)."""
    v = C.c_int()
    L.check("nb_comm_available", L.load().nb_comm_available(C.byref(v)))
    return v.value


def id_publish(path: str, nonce: int, uid: bytes) -> None:
    """Rank 0 of a launch without a process group: publish the id through a file, tagged with the launch's nonce."""
    L.check("nb_comm_id_publish", L.load().nb_comm_id_publish(str(path).encode(), nonce, uid))


def id_await(path: str, nonce: int, timeout_ms: int = 60000) -> bytes:
    """The other ranks: wait for THIS launch's id file (a file with another nonce is ignored); ``NBodyError`` (NB_EIO) on time-out."""
    buf = C.create_string_buffer(L.NB_COMM_ID_BYTES)
    L.check("nb_comm_id_await", L.load().nb_comm_id_await(str(path).encode(), nonce, buf, timeout_ms))
    return buf.raw


class Comm:
    """One ``nb_comm``.  ``Comm.all(sims)``: one process drives every rank (one handle per device);
    ``Comm.rank(sim, id, rank, world)``: one process per GPU."""

    def __init__(self, handle: int, sims: Sequence):
        self._lib = sims[0]._lib          # the build the handles came from (the product, or the tests' hooks build)
        self._h = handle
        self.sims = list(sims)            # keep the handles alive as long as the communicator

    @classmethod
    def all(cls, sims: Sequence) -> "Comm":
        lib = sims[0]._lib
        arr = (C.c_void_p * len(sims))(*[s._h for s in sims])
        h = lib.nb_comm_create_all(arr, len(sims))
        if not h:
            raise L.NBodyError("nb_comm_create_all", L.last_error_code(lib), L.last_error(lib))
        return cls(h, sims)

    @classmethod
    def rank(cls, sim, uid: bytes, rank: int, world: int) -> "Comm":
        if len(uid) != L.NB_COMM_ID_BYTES:
            raise ValueError("uid must be the NB_COMM_ID_BYTES bytes of nb_comm_unique_id")
        h = sim._lib.nb_comm_create_rank(sim._h, uid, rank, world)
        if not h:
            raise L.NBodyError("nb_comm_create_rank", L.last_error_code(sim._lib), L.last_error(sim._lib))
        return cls(h, [sim])

    def step(self, nsteps: int = 1, dt: Optional[float] = None) -> None:
        """Enqueue ``nsteps`` sharded steps; returns as soon as they are enqueued."""
        L.check("nb_comm_step", self._lib.nb_comm_step(self._h, 0.0 if dt is None else dt, nsteps), self._lib)

    def flush(self) -> None:
        L.check("nb_comm_flush", self._lib.nb_comm_flush(self._h), self._lib)

    def wait(self) -> None:
        L.check("nb_comm_wait", self._lib.nb_comm_wait(self._h), self._lib)

    def profile(self, on: bool = True) -> None:
        """Per-phase HIP events inside the library's loop (``nb_comm_profile``); off by default."""
        L.check("nb_comm_profile", self._lib.nb_comm_profile(self._h, int(on)), self._lib)

    def phases(self, handle: int = 0, reset: bool = True) -> dict:
        """``nb_comm_phase_read``: {phase: mean ms per step on the compute stream, ..., "steps": k} for local handle ``handle``."""
        ms = (C.c_double * L.NB_COMM_PHASES)()
        steps = C.c_uint64()
        L.check("nb_comm_phase_read", self._lib.nb_comm_phase_read(self._h, handle, ms, C.byref(steps), int(reset)), self._lib)
        k = max(1, int(steps.value))
        out = {name: ms[i] / k for i, name in enumerate(L.NB_PH_NAMES)}
        out["steps"] = int(steps.value)
        return out

    def info(self) -> dict:
        p, w, k, v = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        L.check("nb_comm_info", self._lib.nb_comm_info(self._h, C.byref(p), C.byref(w), C.byref(k), C.byref(v)), self._lib)
        return {"protocol": p.value, "world": w.value, "local_handles": k.value, "rccl_version": v.value}

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.nb_comm_destroy(self._h)
            self._h = None
            self.sims = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
