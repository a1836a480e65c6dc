"""One-process-per-GPU sharding of the hot path.

Each rank integrates one contiguous block of particles and holds a full-n replica
of the positions (double buffered, in torch tensors so the collectives write them
in place).  The library tells which exchange a handle needs (``nb_shard_protocol``):

NB_SHARD_ALLGATHER — north_star's protocol.  i-particles are independent: the only
exchange is the all-gather of the drifted (x, y) blocks, overlapped with the
local-tile force of the next step on the compute stream:

    step k:   [compute]  force(local j-block) ......... wait(AG k-1) force(remote) integrate
              [comm   ]  ... all-gather of step k-1's positions ...          \\-> all-gather k

NB_SHARD_SYMMETRIC (eps > 0, large n).  Every rank evaluates 1/world of the UNORDERED
pairs with the symmetric kernel — the pairs inside its own block plus an equal run of
the cross-block pairs — which yields a partial acceleration for every particle:

    [compute]  force_sym(pairs inside my block) | wait AG | force_sym(my cross-block run) -> acc_partial[n] | RS | kick, drift
    [comm   ]  ... all-gather(x,y) of the previous step ...                                  reduce-scatter(sum)   \\-> all-gather

NB_SHARD_ALLREDUCE (same eligibility).  The same pair split, but every rank then integrates ALL particles itself:

    [compute]  force_sym(all my items) -> acc_partial[n] | all-reduce(sum, in place) | kick, drift of all n

one collective per step and no position exchange (every rank holds the whole, bit-identical state); nothing overlaps
the all-reduce, but there is only that one latency to pay and one force launch instead of three.

~1.6x fewer VALU cycles per rank for one more latency-bound collective per step
(``reduce_scatter_tensor``; the all-gather stays hidden behind the local pairs).
Which of the two is faster on a given node depends on how exposed that second
collective is, so ``protocol="tune"`` (what ``bench.py`` uses for N > 1) times a few
steps of each after start-up, the ranks agree on the result by all-reduce, and the
faster one is kept — the plain all-gather path is the fallback by construction.
STATUS: the RCCL path with more than one rank has not run on hardware yet (no multi-GPU
box is reachable from the build container); it is rehearsed over gloo and through RCCL
with one rank (tests/test_dist_gpu.py, tests/test_dist_gloo.py).

Two hosts can run the step loop (``driver=``): this module (torch.distributed issues the
collectives between the library's split-step calls: 30-120 us of host time per step), or
the library itself (``driver="c"``: ``nb_comm_step``, nbodysim_amd/csrc/nb_comm.cpp — RCCL
collectives on a communication stream, event-ordered, one foreign call for any number of
steps: 31-40 us per step); ``driver="tune"`` lets both compete in the start-up timing.

The reference has no distributed code at all; this layer is new.
PyTorch is plumbing here (device buffers, streams, the process group); the
force/integrate work is the C ABI's.
"""
from __future__ import annotations

import os
import sys
import threading
import time
from dataclasses import dataclass
from typing import Callable, Dict, Optional, Sequence, Tuple

import numpy as np

from . import _lib as L

PROTOCOLS = ("auto", "symmetric", "allgather", "allreduce", "tune")
DRIVERS = ("torch", "c", "tune")
EXIT_DEADLINE = 3      # exit status of a rank whose Watchdog expired


class Watchdog:
    """A wall-clock deadline around a stretch of code that can block for ever inside a collective (a peer that died, a
    mismatched collective count, a communicator that never forms).  On expiry it prints what was running and the
    caller's report to stderr and ends the PROCESS with status ``EXIT_DEADLINE`` (``os._exit``: no re-exec — the process has
    touched the GPU — and no unwinding through the blocked call); the launcher then sees a non-zero exit naming the
    culprit instead of a silent hang.  The blocked calls (torch.distributed, HIP synchronisation, ctypes) all release the
    GIL, so the timer thread runs.  ``seconds`` <= 0 disables it."""

    #: last resort of the PROCESS, set by a host that already holds a valid result when it starts something optional
    #: (bench.py: the line of the plain all-gather configuration, measured before any faster candidate is tried):
    #: called on expiry with the message, after it went to stderr; an int it returns replaces EXIT_DEADLINE as the status.
    last_resort: Optional[Callable[[str], Optional[int]]] = None

    def __init__(self, seconds: float, what: str, report: Optional[Callable[[], object]] = None, rank: Optional[int] = None,
                 exit_fn: Callable[[int], None] = os._exit):
        self.seconds, self.what, self.report, self.rank, self.exit_fn = float(seconds), what, report, rank, exit_fn
        self._timer: Optional[threading.Timer] = None

    def _expire(self) -> None:
        code = EXIT_DEADLINE
        try:
            who = f"rank {self.rank}: " if self.rank is not None else ""
            msg = f"[nbodysim_amd watchdog] {who}deadline of {self.seconds:.0f} s expired while {self.what}"
            if self.report is not None:
                try:
                    msg += f" | so far: {self.report()}"
                except Exception as e:      # the report must never keep the process alive
                    msg += f" | (report failed: {e})"
            sys.stderr.write(msg + "\n")
            sys.stderr.flush()
            hook = Watchdog.last_resort
            if hook is not None:
                try:
                    alt = hook(msg)
                    if isinstance(alt, int):
                        code = alt
                except Exception as e:      # nor may the last resort
                    sys.stderr.write(f"[nbodysim_amd watchdog] last resort failed: {e}\n")
        finally:
            self.exit_fn(code)

    def __enter__(self):
        if self.seconds > 0:
            self._timer = threading.Timer(self.seconds, self._expire)
            self._timer.daemon = True
            self._timer.start()
        return self

    def __exit__(self, *exc):
        if self._timer is not None:
            self._timer.cancel()
            self._timer = None
        return False


@dataclass(frozen=True)
class ShardPlan:
    """Contiguous blocks of ``stride = ceil(n / world)`` particles; the last block is shorter when world does not
    divide n (ragged).  Collectives always move ``stride`` rows per rank (one all-gather, equal counts): the replicas
    are allocated with ``padded_n = world * stride`` rows and the rows past n are never read by the kernels."""

    n: int
    world: int
    rank: int

    def __post_init__(self):
        if self.world < 1 or not (0 <= self.rank < self.world):
            raise ValueError(f"bad rank/world {self.rank}/{self.world}")
        if self.stride * (self.world - 1) >= self.n:
            raise ValueError(f"n={self.n} is too small for {self.world} ranks: every rank must own at least one particle")

    @property
    def stride(self) -> int:
        return -(-self.n // self.world)

    @property
    def padded_n(self) -> int:
        return self.stride * self.world

    @property
    def ragged(self) -> bool:
        return self.padded_n != self.n

    @property
    def i_begin(self) -> int:
        return self.rank * self.stride

    @property
    def i_count(self) -> int:
        return min(self.stride, self.n - self.i_begin)

    @property
    def i_end(self) -> int:
        return self.i_begin + self.i_count

    def block(self, rank: int) -> slice:
        return slice(rank * self.stride, min((rank + 1) * self.stride, self.n))


def exchange_positions(full, plan: ShardPlan, group=None, async_op: bool = False, even_alone: bool = False):
    """All-gather the owned block of ``full`` (an (n, 2) torch tensor, any device)
    into every rank's ``full`` in place.  Returns the Work handle if async.  ``even_alone``: issue the collective with one
    rank too (the single-rank rehearsal of the sharded path: RCCL's one-rank all-gather)."""
    import torch.distributed as dist

    if plan.world == 1 and not even_alone:
        return None
    if full.shape[0] != plan.padded_n:
        raise ValueError(f"the replica must hold padded_n = {plan.padded_n} rows (n = {plan.n}, {plan.world} ranks), not {full.shape[0]}")
    own = full[plan.i_begin : plan.i_begin + plan.stride]     # equal counts; a ragged last block sends its padding rows too
    return dist.all_gather_into_tensor(full, own, group=group, async_op=async_op)


def reduce_accelerations(acc_full, acc_owned, plan: ShardPlan, group=None) -> None:
    """Sum the ranks' partial accelerations ``acc_full`` (n, 2); every rank keeps its own block in ``acc_owned``."""
    import torch.distributed as dist

    if dist.get_backend(group) == "nccl":
        dist.reduce_scatter_tensor(acc_owned, acc_full, op=dist.ReduceOp.SUM, group=group)
    else:   # gloo has no reduce-scatter: all-reduce, then keep the owned rows
        dist.all_reduce(acc_full, op=dist.ReduceOp.SUM, group=group)
        acc_owned.copy_(acc_full[plan.i_begin : plan.i_end])


# ---------------------------------------------------------------------------
# start-up agreement between the ranks (pure torch.distributed: testable over gloo on CPU)
# ---------------------------------------------------------------------------
def _comm_device(group=None):
    import torch
    import torch.distributed as dist

    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")


def ranks_agree(values: Sequence[int], group=None) -> Tuple[bool, list, list]:
    """True iff every rank passed the same integer vector.  One all-reduce (MAX over [v, -v]); returns
    (agree, per-entry minimum, per-entry maximum) — identical on every rank, so all of them take the same branch."""
    import torch
    import torch.distributed as dist

    v = [int(x) for x in values]
    t = torch.tensor(v + [-x for x in v], dtype=torch.int64, device=_comm_device(group))
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    k = len(v)
    hi = [int(x) for x in t[:k].tolist()]
    lo = [-int(x) for x in t[k:].tolist()]
    return lo == hi, lo, hi


def agree_on_fastest(local_seconds: Dict[str, float], group=None,
                     prefer: Sequence[str] = ("symmetric", "symmetric+late", "symmetric-late", "allreduce", "allgather")):
    """Every rank passes its own timing of each candidate (``inf`` = candidate unavailable); the job's time of a
    candidate is the MAX over ranks (the slowest rank sets the step rate).  Returns (winner, {name: job seconds}),
    identical on every rank; ties and near-ties (within 1 %) go to the earlier name in ``prefer``."""
    import torch
    import torch.distributed as dist

    names = sorted(local_seconds)
    t = torch.tensor([float(local_seconds[k]) for k in names], dtype=torch.float64, device=_comm_device(group))
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    job = {k: float(x) for k, x in zip(names, t.tolist())}
    order = [k for k in prefer if k in job] + [k for k in names if k not in prefer]
    best = order[0]
    for k in order[1:]:
        if job[k] < 0.99 * job[best]:
            best = k
    if not np.isfinite(job[best]):
        raise RuntimeError(f"no sharding protocol is available on every rank: {job}")
    return best, job


def after_collectives(err: BaseException) -> BaseException:
    """Mark ``err`` as raised at a point where this rank has issued every collective its peers expect of it (or where every rank
    raises together): ``time_candidates`` may then let the ranks agree on the failure instead of ending the job."""
    err.collectives_complete = True      # type: ignore[attr-defined]
    return err


def time_candidates(names: Sequence[str], run_one: Callable[[str, Dict[str, float]], float], group=None, deadline_s: float = 120.0,
                    rank: int = 0, prefer: Sequence[str] = (), log: Optional[Callable[[str], None]] = None,
                    cleanup: Optional[Callable[[str], None]] = None, failed: Optional[Dict[str, str]] = None):
    """The start-up timing as a pure driver (testable over gloo on the CPU): ``run_one(name, local_so_far)`` runs one
    candidate on this rank and returns its seconds per step (``inf`` = unavailable); candidates run in the order given —
    simplest first, so that the plainest protocol has a number before anything exotic is tried — each under its own
    ``Watchdog``: a rank stuck inside a candidate ends the job with a non-zero exit that NAMES the candidate and lists the
    timings gathered so far, on every rank that waits for it.
    A candidate that RAISES on any rank (and leaves its peers able to go on: see ``DistributedSimulation.step``, which keeps
    issuing a failed rank's collectives) is not the end of the run: after every candidate the ranks agree on whether it
    completed everywhere (one all-reduce); if not it counts as unavailable on ALL of them, ``cleanup(name)`` runs (errors
    ignored), ``failed[name]`` says why, and the next candidate is tried.  That agreement is itself a collective, so it is
    entered only for a failure that is KNOWN to have left this rank in step with its peers: ``run_one`` marks such an error with
    ``after_collectives(err)`` — raised once the trial's last collective was through, or raised by every rank together.  Anything
    else (a C-loop step that failed half-way through its schedule, an exception before the trial's barriers) is re-raised at
    once: this rank's next collective would pair with a DIFFERENT one of its peers', and the launcher tearing the job down now
    is the only safe outcome.  Returns (winner, {name: job seconds}) from ``agree_on_fastest`` — which raises only if nothing
    at all survived."""
    local: Dict[str, float] = {}
    failed = {} if failed is None else failed
    for name in names:
        with Watchdog(deadline_s, f"timing the start-up candidate '{name}'", report=lambda: dict(local), rank=rank):
            err: Optional[BaseException] = None
            try:
                value = float(run_one(name, local))
            except Exception as e:       # noqa: BLE001 - whatever it was, the other candidates still deserve their turn ...
                if not getattr(e, "collectives_complete", False):
                    raise                # ... unless this rank may be out of step with its peers: no further collective from here
                err, value = e, float("inf")
            _, lo, _ = ranks_agree([0 if err is not None else 1], group)
            if lo[0] == 0:
                value = float("inf")
                failed[name] = f"{type(err).__name__}: {err}" if err is not None else "failed on another rank"
                if cleanup is not None:
                    try:
                        cleanup(name)
                    except Exception:    # noqa: BLE001 - a half-built candidate may not close cleanly
                        pass
            local[name] = value
        if log is not None:
            v = local[name]
            log(f"[tune] rank {rank}: {name}: " + (f"{v * 1e3:.3f} ms/step" if np.isfinite(v) else f"unavailable{' (' + failed[name] + ')' if name in failed else ''}"))
    return agree_on_fastest(local, group, prefer=tuple(prefer) if prefer else tuple(names))


# ---------------------------------------------------------------------------
# the sharded trajectory against the unsharded one (north_star: within 1e-5 relative)
# ---------------------------------------------------------------------------
PARITY_TOL = 1e-5      # north_star / BASELINE.json: positions and velocities within 1e-5 relative


class ParityError(RuntimeError):
    """The sharded state is further than the tolerance from the unsharded one (``.result`` = the comparison)."""

    def __init__(self, what: str, result: dict):
        super().__init__(f"parity_check failed ({what}): max rel pos {result.get('max_rel_pos'):.3g}, vel {result.get('max_rel_vel'):.3g} "
                         f"after {result.get('steps')} steps (tolerance {result.get('tolerance'):g}); worst particle {result.get('worst_particle')} "
                         f"(block of rank {result.get('worst_rank')}){': ' + result['error'] if result.get('error') else ''}")
        self.result = result


def max_rel(a, b) -> Tuple[float, int]:
    """(max over particles of |a_i - b_i| / |b_i|, its index): the relative measure of SURVEY §8c (vector norms per
    particle; a particle at rest at the origin is measured absolutely)."""
    a = np.asarray(a, np.float64).reshape(len(a), -1)
    b = np.asarray(b, np.float64).reshape(len(b), -1)
    den = np.linalg.norm(b, axis=1)
    r = np.linalg.norm(a - b, axis=1) / np.where(den > 0, den, 1.0)
    r = np.where(np.isfinite(r), r, np.inf)                 # a NaN anywhere is a failure, not a pass
    k = int(np.argmax(r)) if len(r) else 0
    return (float(r[k]) if len(r) else 0.0), k


def state_rows(bodies: np.ndarray) -> np.ndarray:
    """Body records -> float64 rows [pos | vel] (4 columns, 6 for dims = 3): what the parity check compares."""
    return np.concatenate([np.asarray(bodies["pos"], np.float64), np.asarray(bodies["vel"], np.float64)], axis=1)


def gather_rows(owned: np.ndarray, plan: ShardPlan, group=None) -> np.ndarray:
    """The ranks' owned rows ((i_count, k) float64 each) as one (n, k) array on EVERY rank: one all-gather of equal
    (stride-row) counts, like the position exchange.  Works over RCCL (device tensors) and gloo (host tensors)."""
    import torch
    import torch.distributed as dist

    owned = np.ascontiguousarray(owned, dtype=np.float64)
    if owned.ndim != 2 or owned.shape[0] != plan.i_count:
        raise ValueError(f"rank {plan.rank} owns {plan.i_count} rows, got an array of shape {owned.shape}")
    if plan.world == 1 and not dist.is_initialized():
        return owned.copy()
    dev = _comm_device(group)                     # one rank of an initialised group still goes through the collective (rehearsals)
    mine = torch.zeros((plan.stride, owned.shape[1]), dtype=torch.float64)
    mine[: plan.i_count] = torch.from_numpy(owned)
    mine = mine.to(dev)
    full = torch.empty((plan.padded_n, owned.shape[1]), dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(full, mine, group=group)
    return full.cpu().numpy()[: plan.n].copy()


def compare_with_unsharded(owned: np.ndarray, plan: ShardPlan, reference: Optional[Callable[[], np.ndarray]], steps: int,
                           tol: float = PARITY_TOL, group=None, full_out: Optional[list] = None) -> dict:
    """The self-check of a sharded run: gather every rank's [pos | vel] rows, and on rank 0 compare them with
    ``reference()`` — the rows of the SAME system advanced the same ``steps`` by ONE unsharded handle (only rank 0 calls
    it).  The verdict is broadcast, so the dictionary is identical on every rank and all of them take the same branch:
    {"max_rel_pos", "max_rel_vel", "steps", "tolerance", "ok", "worst_particle", "worst_rank", "error"}.
    ``full_out``, if a list, receives the gathered (n, k) array (identical on every rank)."""
    import torch
    import torch.distributed as dist

    full = gather_rows(owned, plan, group)
    if full_out is not None:
        full_out.append(full)
    res = np.array([np.inf, np.inf, -1.0, 0.0, 0.0])        # pos, vel, worst particle, ok, reference failed
    err = ""
    if plan.rank == 0:
        try:
            ref = np.asarray(reference(), np.float64)
            if ref.shape != full.shape:
                raise ValueError(f"the unsharded reference has shape {ref.shape}, the gathered state {full.shape}")
            half = full.shape[1] // 2
            rp, kp = max_rel(full[:, :half], ref[:, :half])
            rv, kv = max_rel(full[:, half:], ref[:, half:])
            res[:] = [rp, rv, float(kp if rp >= rv else kv), 1.0 if (rp <= tol and rv <= tol) else 0.0, 0.0]
        except Exception as e:          # noqa: BLE001 - the other ranks wait in the broadcast below: always get there
            err = f"the unsharded reference failed on rank 0: {type(e).__name__}: {e}"
            res[4] = 1.0
    if plan.world > 1 or dist.is_initialized():
        t = torch.from_numpy(res).to(_comm_device(group))
        dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        res = t.cpu().numpy()
    worst = int(res[2])
    if res[4] and not err:
        err = "the unsharded reference failed on rank 0 (see its stderr)"
    return {"max_rel_pos": float(res[0]), "max_rel_vel": float(res[1]), "steps": int(steps), "tolerance": float(tol),
            "ok": bool(res[3] == 1.0), "worst_particle": worst if worst >= 0 else None,
            "worst_rank": (worst // plan.stride) if worst >= 0 else None, "error": err or None}


_AGREE_FIELDS = ("created", "protocol", "chunks_per_item", "cross_units_total", "local_units", "tiles", "cus", "late")


class DistributedSimulation:
    """Sharded ``Simulation``: one rank of a ``torch.distributed`` job, one GPU.

    protocol   "auto"       the library's choice (symmetric where eligible)
               "symmetric"  the symmetric pair split; error if the system is not eligible
               "allgather"  north_star's protocol: one-sided kernels, one all-gather per step
               "allreduce"  the symmetric pair split with replicated integration: one all-reduce per step
               "tune"       time ``tune_steps`` steps of each candidate on a scratch copy and keep the fastest (the ranks
                            agree by all-reduce): the symmetric split as the library would size it, the same with the
                            held-back "late" local items switched the other way (they hide the reduce-scatter; on by
                            default from 8 ranks), and the all-gather protocol
    deadline_s seconds a start-up candidate (and the formation of the C-level communicator) may take before the rank prints
               what it was doing and exits with status 3 (``Watchdog``); <= 0 disables
    driver     "torch"      this module issues the collectives through torch.distributed (RCCL via ProcessGroupNCCL)
                            between the library's split-step calls, one Python iteration per step
               "c"          the library's own loop (``nb_comm_step``, nbodysim_amd/csrc/nb_comm.cpp): RCCL collectives on
                            a communication stream, event-ordered, ONE foreign call for any number of steps; the RCCL
                            id is created by rank 0 and broadcast through the torch process group.  RCCL only.
               "tune"       both drivers are candidates of the start-up timing (of every protocol of protocol="tune", or
                            of the one protocol named).  A C-loop candidate may only win if its trial ended in the SAME
                            BITS as the torch-driven trial of the same protocol (all-gather protocol: required; the
                            reducing protocols: bit-identical or within 1e-6, RCCL being free to order a sum differently
                            in another communicator) — ``tuning["validation"]`` says which.
    rehearse_single_rank  with ONE rank: run the sharded protocols anyway (NB_FLAG_SHARD_SINGLE: every pair is local, the
               collectives are RCCL's one-rank forms), including the start-up timing — all of a node run's start-up code on a
               one-GPU box: both step loops as candidates of every protocol, the validation against the unsharded handle, the
               C-loop-against-torch-loop comparison (tests/test_dist_gpu.py).  Never faster than a plain handle.
    verify     (start-up timing only) every candidate's trial state is gathered and compared on rank 0 with ONE unsharded
               handle advanced the same steps; a candidate further than 1e-5 (north_star's tolerance) from it is
               disqualified like an unavailable one — a protocol that mis-orders an exchange on this node cannot win.
    Extra keyword arguments go to ``Simulation`` (``sym_late_us``, ``sym_chunks_per_item`` ... the tuning fields of
    ``nb_params``; they must be the same on every rank and are verified to be).
    """

    def __init__(self, bodies: np.ndarray, eps: float = 1.0, precision: str = "fp32", rsqrt: str = "exact",
                 order: str = "tiled", device_index: Optional[int] = None, group=None, j_slices: int = 0,
                 protocol: str = "auto", tune_steps: int = 12, tune_dt: float = 1e-3, driver: str = "torch",
                 deadline_s: float = 120.0, verify: bool = True, rehearse_single_rank: bool = False, **sim_kwargs):
        import torch
        import torch.distributed as dist

        if protocol not in PROTOCOLS:
            raise ValueError(f"protocol must be one of {PROTOCOLS}")
        if driver not in DRIVERS:
            raise ValueError(f"driver must be one of {DRIVERS}")
        self.dist = dist
        self.torch = torch
        self.group = group
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.plan = ShardPlan(int(bodies.shape[0]), world, rank)
        self._single = bool(rehearse_single_rank) and world == 1 and dist.is_initialized()
        self._multi = world > 1 or self._single            # the handle runs a sharded protocol
        if device_index is None:
            device_index = torch.cuda.current_device()
        self.device = torch.device("cuda", device_index)
        torch.cuda.set_device(self.device)
        self._dtype = torch.float64 if precision == "fp64" else torch.float32
        self._width = 4 if int(sim_kwargs.get("dims", 2)) == 3 else 2     # (x, y) or (x, y, z, m) per particle
        self._args = dict(eps=eps, precision=precision, rsqrt=rsqrt, order=order, j_slices=j_slices, **sim_kwargs)
        self._device_index = device_index
        self.stream = torch.cuda.Stream(self.device)
        self.sim = None
        self.pos: list = []
        self.acc_full = self.acc_owned = None
        self._pending = None
        self.tuning: Optional[dict] = None
        self._phase_on = False
        self._phase_events: list = []
        self._host_enqueue_s = 0.0
        self._host_steps = 0
        self.comm = None
        self.driver = "torch"
        self.deadline_s = float(deadline_s)     # per start-up candidate and for forming the C-level communicator; <= 0: none
        self.verify = bool(verify)
        self._broken: Optional[BaseException] = None   # first failure of a compute call of the torch-driven loop (see step())
        self._defer_errors = False                     # start-up trials: wait() keeps such a failure pending until the trial's collectives are through

        extra: dict = {}
        #: a host that stages its candidates (bench.run_sharded: torch-driven tuning first, then ONE C-loop challenger) asks here
        #: whether the library's own RCCL loop can be a challenger at all (RCCL process group), and what the timing chose beside
        #: the protocol (``chosen_extra``: the late-item share), so that the challenger runs the same configuration
        self.c_loop_available = bool(self._multi and dist.is_initialized() and dist.get_backend(group) == "nccl")
        if self._multi and (protocol == "tune" or driver == "tune"):
            protocol, extra, driver = self._tune(bodies, tune_steps, tune_dt, driver, None if protocol == "tune" else protocol)
        elif protocol == "tune":
            protocol = "auto"
        self.chosen_extra = dict(extra)
        if driver == "tune":
            driver = "torch"
        self._create(bodies, protocol, extra, driver)

    # -- construction ---------------------------------------------------------
    def _create(self, bodies: np.ndarray, protocol: str, extra: Optional[dict] = None, driver: str = "torch") -> None:
        """Allocate the replicas, create the handle, and verify that every rank got the same pair split.  A rank
        whose nb_create fails still reaches the collective, so the job fails on every rank instead of hanging."""
        from .simulation import Simulation

        torch, world, rank = self.torch, self.plan.world, self.plan.rank
        err: Optional[BaseException] = None
        self.sim = None
        self._broken = None
        try:
            # Full-n position replicas owned by torch so the collective can write them.  Allocated and zero-filled ON THE HANDLE'S
            # STREAM: the library uploads the bodies into them on that stream, and torch's own (default) stream is not ordered
            # against a non-blocking one — a zero-fill still queued there could land AFTER the upload and wipe the positions
            # (found by the start-up validation in round 5: with four ranks time-sharing one GPU every few creations lost its
            # positions that way).
            with torch.cuda.stream(self.stream):
                self.pos = [torch.zeros((self.plan.padded_n, self._width), dtype=self._dtype, device=self.device) for _ in range(2)]
                # buffers of the symmetric protocol (partial acceleration of all particles / summed owned block)
                self.acc_full = self.acc_owned = None
                acc_ptrs = None
                replicated = protocol == "allreduce" and self._multi
                if self._multi and protocol != "allgather":
                    self.acc_full = torch.zeros((self.plan.n, self._width), dtype=self._dtype, device=self.device)
                    if replicated:
                        acc_ptrs = (self.acc_full.data_ptr(), self.acc_full.data_ptr())
                    else:
                        self.acc_owned = torch.zeros((self.plan.i_count, self._width), dtype=self._dtype, device=self.device)
                        acc_ptrs = (self.acc_full.data_ptr(), self.acc_owned.data_ptr())
            self.stream.synchronize()
            kw = dict(self._args)
            kw.update(extra or {})
            if protocol == "allgather":
                kw["symmetry"] = False
            self.sim = Simulation(
                bodies, device=self._device_index,
                i_begin=0 if replicated else self.plan.i_begin, i_count=self.plan.n if replicated else self.plan.i_count,
                stream=self.stream.cuda_stream, pos_buffers=(self.pos[0].data_ptr(), self.pos[1].data_ptr()), pos_rows=self.plan.padded_n,
                shard_rank=rank, shard_world=world, acc_buffers=acc_ptrs, shard_allreduce=replicated, shard_single=self._single, **kw,
            )
        except (L.NBodyError, RuntimeError, MemoryError) as e:   # keep going to the collective below
            err = e
        self.symmetric = err is None and self.sim.shard_protocol == L.NB_SHARD_SYMMETRIC
        self.replicated = err is None and self.sim.shard_protocol == L.NB_SHARD_ALLREDUCE
        if self._multi:
            info = self.sim.sym_info() if err is None else {}
            vec = [
                0 if err is not None else 1,
                0 if err is not None else self.sim.shard_protocol,
                info.get("chunks_per_item", 0),
                info.get("cross_units_total", 0),
                info.get("units_local", 0) + info.get("units_late", 0),
                info.get("tiles", 0),
                info.get("cus", 0),
                1 if info.get("items_late", 0) else 0,
            ]
            ok, lo, hi = ranks_agree(vec, self.group)
            if not ok or lo[0] == 0:
                if self.sim is not None:
                    self.sim.close()
                    self.sim = None
                what = ", ".join(f"{k}: {a}..{b}" for k, a, b in zip(_AGREE_FIELDS, lo, hi) if a != b) or "nb_create failed on every rank"
                raise RuntimeError(f"rank {rank}: the ranks did not build the same sharded plan ({what})"
                                   + (f"; this rank: {err}" if err is not None else "")) from err
        elif err is not None:
            raise err
        if self._multi and ((protocol == "symmetric" and not self.symmetric) or (protocol == "allreduce" and not self.replicated)):
            self.sim.close()
            raise RuntimeError(f"protocol='{protocol}' requested but the system is not eligible "
                               "(needs eps > 0, tiled sum, blocks of whole 2048-particle tiles, n/world >= 4096)")
        self.protocol = "symmetric" if self.symmetric else "allreduce" if self.replicated else "allgather"
        self._cur = 0          # index into self.pos of the library's CURRENT replica
        self._pending = None   # Work of the all-gather filling the CURRENT replica
        assert self.sim.pos_buffer(0) == self.pos[0].data_ptr()
        self.driver = driver
        self.comm = None
        if driver == "c":                         # also with one rank: the same code path as on a node (RCCL's one-rank collectives)
            self._create_comm()

    def _create_comm(self) -> None:
        """The library's own RCCL communicator for this run (``nb_comm_create_rank``): rank 0's id goes to the other
        ranks through the torch process group; every rank reports success or failure before anyone proceeds."""
        from .comm import Comm, available, unique_id

        torch, dist, world, rank = self.torch, self.dist, self.plan.world, self.plan.rank
        dev = _comm_device(self.group)
        # 1. can EVERY rank load the transport?  Agreed before anyone enters ncclCommInitRank, which blocks until all
        #    ranks have arrived: a rank whose dlopen fails would otherwise leave the others waiting inside RCCL for ever.
        err: Optional[BaseException] = None
        try:
            available()
        except L.NBodyError as e:
            err = e
        ok, lo, _ = ranks_agree([0 if err is not None else 1], self.group)
        if not ok or lo[0] == 0:
            self.sim.close()
            self.sim = None
            raise RuntimeError(f"rank {rank}: the C-level RCCL communicator could not be formed on every rank (RCCL not loadable on some rank)"
                               + (f"; this rank: {err}" if err is not None else "")) from err
        uid = torch.zeros(L.NB_COMM_ID_BYTES, dtype=torch.uint8, device=dev)
        if rank == 0:
            try:
                uid = torch.frombuffer(bytearray(unique_id()), dtype=torch.uint8).to(dev)
            except L.NBodyError as e:           # still take part in the broadcast: an all-zero id tells the others
                err = e
        dist.broadcast(uid, src=0, group=self.group)
        raw = bytes(uid.cpu().numpy().tobytes())
        if err is None and any(raw):
            try:
                with Watchdog(self.deadline_s, "forming the C-level RCCL communicator (ncclCommInitRank blocks until every rank arrives)", rank=rank):
                    self.comm = Comm.rank(self.sim, raw, rank, world)
            except L.NBodyError as e:
                err = e
        elif err is None:
            err = RuntimeError("rank 0 could not create an RCCL id")
        ok, lo, _ = ranks_agree([0 if err is not None else 1], self.group)
        if not ok or lo[0] == 0:
            if self.comm is not None:
                self.comm.close()
                self.comm = None
            self.sim.close()
            self.sim = None
            raise RuntimeError(f"rank {rank}: the C-level RCCL communicator could not be formed on every rank"
                               + (f"; this rank: {err}" if err is not None else "")) from err

    #: arguments of ``Simulation`` that change the PHYSICS (everything else only shapes launches): what the unsharded
    #: reference handle of the self-check is created with
    _PHYSICS_ARGS = ("eps", "precision", "rsqrt", "order", "integrator", "extras", "dims", "uniform_mass", "mass_scaling")

    def reference_rows(self, bodies: np.ndarray, steps: int, dt: float) -> np.ndarray:
        """[pos | vel] rows of the SAME system advanced ``steps`` steps by ONE unsharded handle on this rank's device —
        what a sharded trajectory is checked against (``compare_with_unsharded``; rank 0 calls this)."""
        from .simulation import Simulation

        kw = {k: v for k, v in self._args.items() if k in self._PHYSICS_ARGS}
        with Simulation(bodies, device=self._device_index, **kw) as ref:
            ref.advance(steps, dt)
            return state_rows(ref.sync())

    def owned_rows(self) -> np.ndarray:
        """[pos | vel] rows of the owned block (float64), after waiting for the enqueued steps."""
        return state_rows(self.sync())

    def _tune(self, bodies: np.ndarray, steps: int, dt: float, driver: str = "torch", only: Optional[str] = None):
        """Time `steps` steps of each candidate on a scratch copy of the system (2 untimed steps first), wall clock
        between barriers, MAX over ranks; the handles are destroyed again, so the simulation proper starts from
        the caller's bodies at frame 0.  Returns (protocol, extra Simulation keyword arguments, driver).
        Every candidate must ALSO be right before it may be fast (``verify``): its trial state is compared with one
        unsharded handle's on rank 0 (1e-5), a replicated (all-reduce) candidate must end with bit-identical replicas on
        every rank, and a C-loop candidate must reproduce the torch-driven trial of its protocol (see ``driver``).  A
        candidate that raises on some rank is skipped by agreement (``time_candidates``), not fatal."""
        late_default_on = self.plan.world >= 8 and float(self._args.get("sym_late_us", 0.0)) == 0.0
        flipped = ("symmetric-late", {"sym_late_us": -1.0}) if late_default_on else ("symmetric+late", {"sym_late_us": 40.0})
        # simplest first: north_star's plain all-gather, then one collective per step (all-reduce), then the symmetric
        # split with its two collectives, then the same with the late items flipped
        base = {"allgather": ("allgather", {}), "allreduce": ("allreduce", {}), "symmetric": ("symmetric", {}),
                flipped[0]: ("symmetric", flipped[1])}
        if float(self._args.get("sym_late_us", 0.0)) != 0.0:      # the caller fixed the late share: nothing to flip
            del base[flipped[0]]
        if only is not None:                                      # one protocol named: only the drivers compete
            base = {"auto": (only, {})} if only == "auto" else {k: v for k, v in base.items() if v[0] == only}
        nccl = self.dist.get_backend(self.group) == "nccl"
        drivers = [d for d in (("torch", "c") if driver == "tune" else (driver,)) if d == "torch" or nccl]
        cands = {}
        for d in drivers:                                         # every torch-driven candidate before any C-loop one
            for name, (cand, extra) in base.items():
                cands[name if d == "torch" else "c:" + name] = (cand, extra, d)
        trial_steps = 2 + steps
        reference: Dict[str, np.ndarray] = {}                     # rank 0: the unsharded trial state, computed once
        states: Dict[str, np.ndarray] = {}                        # gathered trial state of every validated candidate
        validation: Dict[str, dict] = {}

        def unsharded() -> np.ndarray:
            if "rows" not in reference:
                reference["rows"] = self.reference_rows(bodies, trial_steps, dt)
            return reference["rows"]

        def run_one(name: str, local: Dict[str, float]) -> float:
            cand, extra, d = cands[name]
            pre = "c:" if d == "c" else ""
            if local.get(pre + "symmetric") == float("inf") and cand == "symmetric" and name != pre + "symmetric":
                return float("inf")                               # not eligible once, not eligible with the late items flipped
            try:
                self._create(bodies, cand, extra, d)
            except RuntimeError as e:                             # raised on EVERY rank (agreed inside _create)
                if "not eligible" not in str(e) and "communicator could not be formed" not in str(e):
                    raise after_collectives(e)
                return float("inf")
            # From here to the end of the trial EVERY rank walks the same sequence of collectives (steps, barriers, the replica check,
            # the validation gather), whatever happens to its own compute calls: a failed launch is kept pending (_defer_errors) and the
            # rank goes on issuing its collectives with garbage, so that no peer is left inside one; it is raised once the trial's last
            # collective is through, and time_candidates lets the ranks agree on it.
            self._defer_errors = True
            try:
                self.advance(2, dt)
                self.wait()
                self.dist.barrier(group=self.group)
                t0 = time.perf_counter()
                self.advance(steps, dt)
                self.wait()
                self.dist.barrier(group=self.group)
                per_step = (time.perf_counter() - t0) / steps
                v: dict = {}
                if self.replicated and not self.replicas_identical():
                    per_step, v["replicas"] = float("inf"), "diverged: this transport does not hand every rank the same sum"
                if self.verify:
                    got: list = []
                    try:                                              # a rank whose compute failed has no state worth reading (its handle
                        rows = self.owned_rows() if self._broken is None else None       # may sit inside a split step) — but it still takes
                    except L.NBodyError as e:                         # part in the gather below, with zeros
                        rows, self._broken = None, (self._broken or e)
                    if rows is None:
                        rows = np.zeros((self.plan.i_count, 2 * (3 if self._width == 4 else 2)))
                    v.update(compare_with_unsharded(rows, self.plan, unsharded, trial_steps, group=self.group, full_out=got))
                    if not v["ok"]:
                        per_step = float("inf")
                    elif d == "c" and name[2:] in states:             # the library's loop against the torch-driven loop, same protocol
                        same = bool(np.array_equal(got[0], states[name[2:]]))
                        half = got[0].shape[1] // 2
                        rel = 0.0 if same else max(max_rel(got[0][:, :half], states[name[2:]][:, :half])[0],
                                                   max_rel(got[0][:, half:], states[name[2:]][:, half:])[0])
                        reducing = self.symmetric or self.replicated
                        v["vs_torch_loop"] = ("bit-identical" if same else
                                              f"differs by {rel:.2e} (" + ("another order of the RCCL sum: accepted below 1e-6" if reducing and rel <= 1e-6
                                                                           else "REJECTED: the same exchange must give the same bits") + ")")
                        if not same and not (reducing and rel <= 1e-6):
                            per_step = float("inf")
                    elif d == "c":
                        v["vs_torch_loop"] = "no torch-driven trial of this protocol to compare with; judged on the unsharded check alone"
                    if np.isfinite(per_step):
                        states[name] = got[0]
            finally:
                self._defer_errors = False
            pending, self._broken = self._broken, None
            validation[name] = v
            if not np.isfinite(per_step) and self.plan.rank == 0:
                sys.stderr.write(f"[tune] rank 0: {name} DISQUALIFIED: " + ", ".join(f"{a}={b}" for a, b in v.items() if a in
                                 ("max_rel_pos", "max_rel_vel", "worst_particle", "worst_rank", "error", "vs_torch_loop", "replicas")) + "\n")
                sys.stderr.flush()
            self.close()
            if pending is not None:
                raise after_collectives(pending)      # deferred through the whole trial (_compute / _defer_errors): the peers are not stranded
            return per_step

        def cleanup(name: str) -> None:
            self._broken = None
            self.close()

        log = (lambda m: (sys.stderr.write(m + "\n"), sys.stderr.flush())) if self.plan.rank == 0 else None
        # near-ties (within 1 %) go to the earlier name: the protocol order below, and within a protocol the library's C loop before the
        # torch-driven one — a C-loop candidate that is still in the race HAS reproduced the torch-driven trial, and north_star wants
        # the host loop in C; it loses only where it is measurably slower
        order = ("symmetric", "symmetric+late", "symmetric-late", "allreduce", "allgather", "auto")
        prefer = tuple(x for k in order for x in ("c:" + k, k))
        failed: Dict[str, str] = {}
        best, job = time_candidates(list(cands), run_one, self.group, self.deadline_s, self.plan.rank,
                                    prefer=prefer, log=log, cleanup=cleanup, failed=failed)
        self.tuning = {"steps": steps, "ms_per_step": {k: (v * 1e3 if np.isfinite(v) else None) for k, v in job.items()}, "chosen": best,
                       "order": list(cands), "deadline_s_per_candidate": self.deadline_s,
                       "validation": {k: {a: b for a, b in v.items() if a in ("max_rel_pos", "max_rel_vel", "steps", "ok", "vs_torch_loop", "replicas", "error")}
                                      for k, v in validation.items()},
                       "failed": failed,
                       "validated_against": (f"one unsharded handle on rank 0, {trial_steps} steps, tolerance {PARITY_TOL:g}" if self.verify else None)}
        return cands[best]

    def replicas_identical(self) -> bool:
        """Replicated (all-reduce) protocol: every rank integrates all n particles from an all-reduced sum, so the
        replicas stay identical only if the transport hands every rank the SAME bits.  One all-reduce (MAX of
        [c, -c]) of a checksum of the current positions; True on every rank iff they all agree."""
        if self.plan.world == 1 or not self.replicated:
            return True
        self.wait()
        torch = self.torch
        with torch.cuda.stream(self.stream):
            c = self.pos[self._cur_index()].view(torch.int32).to(torch.int64).sum()
        self.stream.synchronize()
        ok, _, _ = ranks_agree([int(c.item())], self.group)
        return ok

    def _cur_index(self) -> int:
        return 0 if self.sim.pos_buffer(0) == self.pos[0].data_ptr() else 1

    # -- stepping ---------------------------------------------------------------
    @property
    def frame(self) -> int:
        return self.sim.frame

    def _reduce_accelerations(self) -> None:
        """Sum the ranks' partial accelerations; every rank keeps its own block."""
        reduce_accelerations(self.acc_full, self.acc_owned, self.plan, self.group)

    def _mark(self, marks: Optional[list]) -> None:
        if marks is not None:
            ev = self.torch.cuda.Event(enable_timing=True)
            ev.record(self.stream)
            marks.append(ev)

    def _compute(self, call, *args) -> None:
        """One compute call of the torch-driven loop.  After the first failure the rank stops computing but ``step`` KEEPS
        ISSUING ITS COLLECTIVES (same kinds, same counts): the peers of a rank whose launch failed are then not left
        inside a collective that never completes — they finish their steps (with this rank's garbage), and the failure
        surfaces on this rank at the next ``wait()``, where a host can let the ranks agree on it (``time_candidates``)."""
        if self._broken is not None:
            return
        try:
            call(*args)
        except L.NBodyError as e:
            self._broken = e

    def step(self, dt: Optional[float] = None) -> None:
        """One sharded step; only enqueues (no host sync)."""
        if self.comm is not None:
            t_host = time.perf_counter()
            self.comm.step(1, dt)
            self._host_enqueue_s += time.perf_counter() - t_host
            self._host_steps += 1
            return
        marks = [] if self._phase_on else None
        t_host = time.perf_counter()
        if self.replicated:
            with self.torch.cuda.stream(self.stream):
                self._mark(marks)
                self._compute(self.sim.step_begin, dt)   # all my pairs -> partial acceleration of every particle
                self._mark(marks)
                self.dist.all_reduce(self.acc_full, op=self.dist.ReduceOp.SUM, group=self.group)   # in place, same bits everywhere
                self._mark(marks)
                self._compute(self.sim.step_finish)      # every rank kicks and drifts all n
                self._mark(marks)
            self._host_enqueue_s += time.perf_counter() - t_host
            self._host_steps += 1
            if marks is not None:
                self._phase_events.append(marks)
            return
        with self.torch.cuda.stream(self.stream):
            self._mark(marks)
            self._compute(self.sim.step_begin, dt)   # pairs inside my own block / local j-block: overlaps the all-gather still in flight
            self._mark(marks)
            if self._pending is not None:
                self._pending.wait()         # compute stream waits: every rank's new positions are in the CURRENT replica
                self._pending = None
            self._mark(marks)
            if self.symmetric:
                self._compute(self.sim.step_mid)     # my run of the cross-block pairs -> partial acceleration of all n
                self._mark(marks)
                self._reduce_accelerations() # ordered after the force on this stream by the process group
                self._mark(marks)
            self._compute(self.sim.step_finish)      # (all-gather protocol: remote j-blocks,) kick, drift -> NEXT becomes CURRENT
            self._mark(marks)
            self._cur ^= 1
            self._pending = exchange_positions(self.pos[self._cur], self.plan, self.group, async_op=True, even_alone=self._single)
        self._host_enqueue_s += time.perf_counter() - t_host
        self._host_steps += 1
        if marks is not None:
            self._phase_events.append(marks)

    def advance(self, nsteps: int, dt: Optional[float] = None) -> None:
        if self.comm is not None:                 # the library's loop: one foreign call for all the steps
            t_host = time.perf_counter()
            self.comm.step(nsteps, dt)
            self._host_enqueue_s += time.perf_counter() - t_host
            self._host_steps += nsteps
            return
        for _ in range(nsteps):
            self.step(dt)

    def wait(self) -> None:
        if self.comm is not None:
            self.comm.wait()
            return
        with self.torch.cuda.stream(self.stream):
            if self._pending is not None:
                self._pending.wait()
                self._pending = None
        self.stream.synchronize()
        if self._broken is not None and not self._defer_errors:   # a compute call failed earlier; its collectives were still issued (_compute)
            err, self._broken = self._broken, None
            raise err

    # -- per-phase timing (HIP events on the compute stream) ----------------------
    def profile_phases(self, on: bool = True) -> None:
        self._phase_on = bool(on)
        self._phase_events = []
        self._host_enqueue_s, self._host_steps = 0.0, 0
        if self.comm is not None:
            self.comm.profile(on)

    def phase_report(self) -> dict:
        """Mean milliseconds per step of each phase AS SEEN BY THE COMPUTE STREAM (waiting included), plus the host's
        enqueue time per step.  symmetric: local | ag_wait | cross (+ gather of the slabs) | reduce_scatter | finish;
        all-gather: local | ag_wait | remote_finish.  When the local items run on the side stream
        (``local_on_side_stream``) their time shows up inside `cross`, which joins them."""
        self.wait()
        if self.comm is not None:                 # the library's loop times its own phases (nb_comm_profile / nb_comm_phase_read)
            ph = self.comm.phases(0, reset=True) if self._phase_on else {"steps": 0}
            if self.symmetric:
                out = {"local": ph.get("local", 0.0), "ag_wait": ph.get("ag_wait", 0.0), "cross": ph.get("cross", 0.0),
                       "reduce_scatter": ph.get("reduce", 0.0), "finish": ph.get("finish", 0.0)}
            elif self.replicated:
                out = {"force": ph.get("local", 0.0), "all_reduce": ph.get("reduce", 0.0), "finish": ph.get("finish", 0.0)}
            else:
                out = {"local": ph.get("local", 0.0), "ag_wait": ph.get("ag_wait", 0.0), "remote_finish": ph.get("finish", 0.0)}
            out["stream_total"] = sum(out.values())
            out["host_enqueue"] = self._host_enqueue_s / max(1, self._host_steps) * 1e3
            out["steps"] = self._host_steps
            out["phase_steps"] = ph["steps"]          # steps the phase events cover (0 unless profile_phases(True))
            out["driver"] = "c (nb_comm_step): HIP events inside the library's loop"
            return out
        names = (("local", "ag_wait", "cross", "reduce_scatter", "finish") if self.symmetric else
                 ("force", "all_reduce", "finish") if self.replicated else ("local", "ag_wait", "remote_finish"))
        tot = {k: 0.0 for k in names}
        for marks in self._phase_events:
            for k, (a, b) in zip(names, zip(marks[:-1], marks[1:])):
                tot[k] += a.elapsed_time(b)
        steps = max(1, len(self._phase_events))
        out = {k: v / steps for k, v in tot.items()}
        out["stream_total"] = sum(out.values())
        out["host_enqueue"] = self._host_enqueue_s / max(1, self._host_steps) * 1e3
        out["steps"] = len(self._phase_events)
        info, aux = self.sim.sym_info(), int(self._args.get("sym_aux_stream", 0))
        out["local_on_side_stream"] = bool(self.symmetric and (aux > 0 or (aux == 0 and info["items_local"] <= 4 * info["cus"])))
        return out

    # -- host views ---------------------------------------------------------------
    def sync(self) -> np.ndarray:
        """Owned block as Body records (each GPU copies back only its block; a replicated handle holds all n and
        returns its rank's block of them, so the result is the same in every protocol)."""
        self.wait()
        b = self.sim.sync()
        return b[self.plan.i_begin:self.plan.i_end] if self.replicated else b

    def gather_bodies(self) -> Optional[np.ndarray]:
        """All blocks on every rank (host-side object gather; for tests / dumps)."""
        mine = self.sync().copy()
        if self.plan.world == 1:
            return mine
        parts = [None] * self.plan.world
        self.dist.all_gather_object(parts, mine, group=self.group)
        return np.concatenate(parts)

    def energy(self) -> tuple:
        self.wait()
        if self.replicated and not self.replicas_identical():
            raise RuntimeError("the replicas of the all-reduce protocol have diverged: this transport does not deliver "
                               "bit-identical sums on every rank (use protocol='symmetric' or 'allgather')")
        k, u = self.sim.energy()
        if self.plan.world > 1 and not self.replicated:       # a replicated handle already holds the total
            t = self.torch.tensor([k, u], dtype=self.torch.float64, device=self.device)
            if self.dist.get_backend(self.group) != "nccl":
                t = t.cpu()
            self.dist.all_reduce(t, group=self.group)
            k, u = float(t[0]), float(t[1])
        return k, u

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def close(self) -> None:
        if self.sim is not None:
            try:
                self.wait()
            except L.NBodyError:                  # a failure already reported (or about to be, by the caller that closes)
                pass
            if self.comm is not None:
                self.comm.close()
                self.comm = None
            self.sim.close()
            self.sim = None
        self.pos = []
        self.acc_full = self.acc_owned = None
