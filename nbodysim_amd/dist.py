"""One-process-per-GPU sharding of the hot path.

Each rank integrates one contiguous block of particles and holds a full-n replica
of the positions (double buffered, in torch tensors so the collectives write them
in place).  The library tells which exchange a handle needs (``nb_shard_protocol``):

NB_SHARD_SYMMETRIC (eps > 0, large n — the benchmark case).  Every rank evaluates
1/world of the UNORDERED pairs with the symmetric kernel — the pairs inside its own
block plus an equal run of the cross-block pairs — which yields a partial
acceleration for every particle:

    [compute]  force_sym(pairs inside my block) | wait AG | force_sym(my cross-block run) -> acc_partial[n] | RS | kick, drift
    [comm   ]  ... all-gather(x,y) of the previous step ...                                  reduce-scatter(sum)   \\-> all-gather

two collectives per step (2 MiB each at N = 262 144, latency-bound), RCCL
``reduce_scatter_tensor`` / ``all_gather_into_tensor`` over xGMI; the all-gather is
hidden behind the local pairs (1/world of the rank's work).

NB_SHARD_ALLGATHER (everything else).  i-particles are independent: the only
exchange is the all-gather of the drifted (x, y) blocks, overlapped with the
local-tile force of the next step on the compute stream:

    step k:   [compute]  force(local j-block) ......... wait(AG k-1) force(remote) integrate
              [comm   ]  ... all-gather of step k-1's positions ...          \\-> all-gather k

The reference has no distributed code at all; this layer is new.
PyTorch is plumbing here (device buffers, streams, the process group); the
force/integrate work is the C ABI's.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _lib as L


@dataclass(frozen=True)
class ShardPlan:
    """Contiguous, equal blocks of the particle index range (n % world == 0)."""

    n: int
    world: int
    rank: int

    def __post_init__(self):
        if self.world < 1 or not (0 <= self.rank < self.world):
            raise ValueError(f"bad rank/world {self.rank}/{self.world}")
        if self.n % self.world != 0:
            raise ValueError(f"n={self.n} must be a multiple of the world size {self.world} "
                             "(equal blocks keep the all-gather a single collective)")

    @property
    def i_count(self) -> int:
        return self.n // self.world

    @property
    def i_begin(self) -> int:
        return self.rank * self.i_count

    @property
    def i_end(self) -> int:
        return self.i_begin + self.i_count

    def block(self, rank: int) -> slice:
        c = self.n // self.world
        return slice(rank * c, (rank + 1) * c)


def exchange_positions(full, plan: ShardPlan, group=None, async_op: bool = False):
    """All-gather the owned block of ``full`` (an (n, 2) torch tensor, any device)
    into every rank's ``full`` in place.  Returns the Work handle if async."""
    import torch.distributed as dist

    if plan.world == 1:
        return None
    own = full[plan.i_begin : plan.i_end]
    return dist.all_gather_into_tensor(full, own, group=group, async_op=async_op)


def reduce_accelerations(acc_full, acc_owned, plan: ShardPlan, group=None) -> None:
    """Sum the ranks' partial accelerations ``acc_full`` (n, 2); every rank keeps its own block in ``acc_owned``."""
    import torch.distributed as dist

    if dist.get_backend(group) == "nccl":
        dist.reduce_scatter_tensor(acc_owned, acc_full, op=dist.ReduceOp.SUM, group=group)
    else:   # gloo has no reduce-scatter: all-reduce, then keep the owned rows
        dist.all_reduce(acc_full, op=dist.ReduceOp.SUM, group=group)
        acc_owned.copy_(acc_full[plan.i_begin : plan.i_end])


class DistributedSimulation:
    """Sharded ``Simulation``: one rank of a ``torch.distributed`` job, one GPU."""

    def __init__(self, bodies: np.ndarray, eps: float = 1.0, precision: str = "fp32", rsqrt: str = "exact",
                 order: str = "tiled", device_index: Optional[int] = None, group=None, j_slices: int = 0):
        import torch
        import torch.distributed as dist

        from .simulation import Simulation

        self.dist = dist
        self.torch = torch
        self.group = group
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.plan = ShardPlan(int(bodies.shape[0]), world, rank)
        if device_index is None:
            device_index = torch.cuda.current_device()
        self.device = torch.device("cuda", device_index)
        torch.cuda.set_device(self.device)
        dtype = torch.float64 if precision == "fp64" else torch.float32
        # full-n position replicas owned by torch so the collective can write them
        self.pos = [torch.empty((self.plan.n, 2), dtype=dtype, device=self.device) for _ in range(2)]
        self.stream = torch.cuda.Stream(self.device)
        # buffers of the symmetric protocol (partial acceleration of all particles / summed owned block)
        self.acc_full = self.acc_owned = None
        acc_ptrs = None
        if world > 1:
            self.acc_full = torch.zeros((self.plan.n, 2), dtype=dtype, device=self.device)
            self.acc_owned = torch.zeros((self.plan.i_count, 2), dtype=dtype, device=self.device)
            acc_ptrs = (self.acc_full.data_ptr(), self.acc_owned.data_ptr())
        self.sim = Simulation(
            bodies, eps=eps, precision=precision, rsqrt=rsqrt, order=order, device=device_index, j_slices=j_slices,
            i_begin=self.plan.i_begin, i_count=self.plan.i_count, stream=self.stream.cuda_stream,
            pos_buffers=(self.pos[0].data_ptr(), self.pos[1].data_ptr()),
            shard_rank=rank, shard_world=world, acc_buffers=acc_ptrs,
        )
        self.symmetric = self.sim.shard_protocol == L.NB_SHARD_SYMMETRIC
        if world > 1:   # the protocol fixes which collectives a step issues: every rank must have chosen the same one
            t = torch.tensor([self.sim.shard_protocol, -self.sim.shard_protocol], dtype=torch.int64, device=self.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
            if int(t[0]) != -int(t[1]):
                raise RuntimeError("sharding protocol differs between ranks (different environment or device?)")
        self._cur = 0          # index into self.pos of the library's CURRENT replica
        self._pending = None   # Work of the all-gather filling the CURRENT replica
        assert self.sim.pos_buffer(0) == self.pos[0].data_ptr()

    @property
    def frame(self) -> int:
        return self.sim.frame

    def _reduce_accelerations(self) -> None:
        """Sum the ranks' partial accelerations; every rank keeps its own block."""
        reduce_accelerations(self.acc_full, self.acc_owned, self.plan, self.group)

    def step(self, dt: Optional[float] = None) -> None:
        """One sharded step; only enqueues (no host sync)."""
        if self.symmetric:
            with self.torch.cuda.stream(self.stream):
                self.sim.step_begin(dt)      # pairs inside my own block: overlaps the all-gather still in flight
                if self._pending is not None:
                    self._pending.wait()     # every rank's new positions are in the CURRENT replica
                    self._pending = None
                self.sim.step_mid()          # my run of the cross-block pairs -> partial acceleration of all n
                self._reduce_accelerations() # ordered after the force on this stream by the process group
                self.sim.step_finish()       # kick, drift of my block -> NEXT becomes CURRENT
                self._cur ^= 1
                self._pending = exchange_positions(self.pos[self._cur], self.plan, self.group, async_op=True)
            return
        with self.torch.cuda.stream(self.stream):
            self.sim.step_begin(dt)          # local j-block: overlaps the in-flight all-gather
            if self._pending is not None:
                self._pending.wait()         # compute stream waits for the remote blocks
                self._pending = None
            self.sim.step_finish()           # remote j-blocks, kick, drift -> NEXT becomes CURRENT
            self._cur ^= 1
            self._pending = exchange_positions(self.pos[self._cur], self.plan, self.group, async_op=True)

    def advance(self, nsteps: int, dt: Optional[float] = None) -> None:
        for _ in range(nsteps):
            self.step(dt)

    def wait(self) -> None:
        with self.torch.cuda.stream(self.stream):
            if self._pending is not None:
                self._pending.wait()
                self._pending = None
        self.stream.synchronize()

    def sync(self) -> np.ndarray:
        """Owned block as Body records (each GPU copies back only its block)."""
        self.wait()
        return self.sim.sync()

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
        k, u = self.sim.energy()
        if self.plan.world > 1:
            t = self.torch.tensor([k, u], dtype=self.torch.float64, device=self.device)
            self.dist.all_reduce(t, group=self.group)
            k, u = float(t[0]), float(t[1])
        return k, u

    def close(self) -> None:
        self.wait()
        self.sim.close()
