"""Host-side mirror of the reference's ``Simulation`` surface over the C ABI.

Reference (``Nbodysim/headers/Simulation.hpp:49-75``)::

    class Simulation { public: float dt; size_t frame; std::vector<Body> bodies;
                       Quadtree quadtree; Simulation(); void step(); };
    extern std::atomic<float> SIMULATION_DT;          // main.cpp:39, read once per step (:69)

Here ``Simulation(bodies, ...)`` takes the initial bodies instead of the
hard-coded ``uniform_disc(25000)``; ``step()`` advances one step with the
module-level ``SIMULATION_DT`` (or an explicit dt) and leaves ``bodies``
coherent, like the reference's ``step()`` does for its only caller
(``simulation_thread``, main.cpp:612-635).  ``advance(nsteps)`` is the
throughput form: it only enqueues work and does not refresh ``bodies``.

All compute happens in ``libnbody_hip.so`` on the GPU; nothing here computes
forces.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _lib as L

#: the reference's global time step (``std::atomic<float> SIMULATION_DT{0.01f}``, main.cpp:39)
SIMULATION_DT: float = 0.01


class Simulation:
    """One ``nb_sim`` handle plus the host ``bodies`` view."""

    def __init__(
        self,
        bodies: np.ndarray,
        eps: float = 1.0,           # Simulation.hpp:59 -> Quadtree(theta=1, epsilon=1)
        precision: str = "fp32",
        rsqrt: str = "exact",
        order: str = "tiled",
        integrator: str = "kick_drift",
        extras: int = 0,
        device: int = -1,
        j_slices: int = 0,
        i_begin: int = 0,
        i_count: int = 0,
        stream: Optional[int] = None,
        pos_buffers: Optional[tuple] = None,
        shard_rank: int = 0,
        shard_world: int = 0,
        acc_buffers: Optional[tuple] = None,
        dims: int = 2,
        symmetry: bool = True,
        uniform_mass: bool = True,
        guided_tail: bool = True,
        sym_chunks_per_item: int = 0,
        sym_aux_stream: int = 0,
        sym_late_us: float = 0.0,
        lanes_p: int = 0,
        sym_tail: Optional[tuple] = None,
        shard_allreduce: bool = False,
        first_frame: int = 0,
        shard_single: bool = False,
        mass_scaling=False,
        sym_chunk_pairs: int = 0,
        sym_tile: int = 0,
        pos_rows: int = 0,
        static_items: bool = False,
        library=None,
    ):
        """The last arguments (from ``uniform_mass`` on) are ``nb_params.flags`` and the launch-geometry tuning fields
        (0 / True = the library's automatic choice); the library reads no environment variables.  ``mass_scaling``: False / None
        (default) = both per-pair mass multiplies, True = fold the masses into the pair geometry wherever representable
        (include/nbody.h, NB_FLAG_MASS_SCALING), "measured" = the library measures at upload whether folding is harmless for
        these bodies and folds only then (NB_FLAG_MASS_SCALING_MEASURED).  ``library``: another
        build of the library bound with ``_lib.bind`` (the tests' -DNB_TEST_HOOKS build); default the product."""
        lib = library if library is not None else L.load()
        if bodies.dtype not in (L.BODY_DTYPE, L.BODY3_DTYPE):
            raise TypeError("bodies must be a numpy array of nbodysim_amd.BODY_DTYPE (64-byte Body records)")
        bodies = np.ascontiguousarray(bodies)
        view = L.BODY3_DTYPE if dims == 3 else L.BODY_DTYPE
        bodies = bodies.view(view)
        p = L.nb_params()
        lib.nb_params_default(C.byref(p))
        p.eps = eps
        p.dt = SIMULATION_DT
        p.precision = {"fp32": L.NB_FP32, "fp64": L.NB_FP64}[precision]
        p.rsqrt_mode = {"exact": L.NB_RSQRT_EXACT, "quake": L.NB_RSQRT_QUAKE}[rsqrt]
        p.sum_order = {"tiled": L.NB_SUM_TILED, "sequential": L.NB_SUM_SEQUENTIAL}[order]
        p.integrator = {"kick_drift": L.NB_INTEGRATOR_KICK_DRIFT, "kdk": L.NB_INTEGRATOR_KDK}[integrator]
        p.extras = extras
        p.device = device
        p.j_slices = j_slices
        p.i_begin = i_begin
        p.i_count = i_count
        if stream is not None:
            p.stream = stream
        if pos_buffers is not None:
            p.pos_buffers[0], p.pos_buffers[1] = pos_buffers
        p.dims = dims
        p.flags = ((0 if symmetry else L.NB_FLAG_NO_SYMMETRY) | (0 if uniform_mass else L.NB_FLAG_NO_UNIFORM_MASS)
                   | (0 if guided_tail else L.NB_FLAG_NO_GUIDED_TAIL) | (L.NB_FLAG_SHARD_ALLREDUCE if shard_allreduce else 0)
                   | (L.NB_FLAG_SHARD_SINGLE if shard_single else 0)
                   | (L.NB_FLAG_MASS_SCALING_MEASURED if mass_scaling == "measured" else L.NB_FLAG_MASS_SCALING if mass_scaling is True
                      else L.NB_FLAG_NO_MASS_SCALING if mass_scaling is False else 0)
                   | (L.NB_FLAG_STATIC_ITEMS if static_items else 0))
        p.sym_chunks_per_item, p.sym_aux_stream, p.sym_late_us, p.lanes_p = sym_chunks_per_item, sym_aux_stream, sym_late_us, lanes_p
        if sym_tail is not None:
            p.sym_tail[0], p.sym_tail[1], p.sym_tail[2] = sym_tail
        p.first_frame = first_frame
        p.sym_chunk_pairs = sym_chunk_pairs
        p.sym_tile = sym_tile
        p.pos_rows = pos_rows
        p.shard_rank, p.shard_world = shard_rank, shard_world
        if acc_buffers is not None:
            p.acc_buffers[0], p.acc_buffers[1] = acc_buffers
        self._lib = lib
        self._params = p
        self._h = lib.nb_create(bodies.ctypes.data, bodies.shape[0], C.byref(p))
        if not self._h:
            raise L.NBodyError("nb_create", L.last_error_code(lib), L.last_error(lib))
        self.n = int(lib.nb_count(self._h))
        self.i_begin = int(lib.nb_owned_begin(self._h))
        self.i_count = int(lib.nb_owned_count(self._h))
        #: host view of the owned block; refreshed by step() / sync()
        self.bodies = np.frombuffer(bodies[self.i_begin : self.i_begin + self.i_count].tobytes(), dtype=view).copy()

    # -- reference surface ---------------------------------------------------
    @property
    def frame(self) -> int:
        return int(self._lib.nb_frame(self._h))

    def step(self, dt: Optional[float] = None) -> None:
        """``Simulation::step()``: one step, then ``bodies`` is coherent."""
        L.check("nb_step", self._lib.nb_step(self._h, SIMULATION_DT if dt is None else dt, 1), self._lib)
        self.sync()

    # -- throughput / build-defined surface ------------------------------------
    def advance(self, nsteps: int, dt: Optional[float] = None) -> None:
        """Enqueue nsteps steps; does not wait and does not refresh ``bodies``."""
        L.check("nb_step", self._lib.nb_step(self._h, SIMULATION_DT if dt is None else dt, nsteps), self._lib)

    def wait(self) -> None:
        L.check("nb_wait", self._lib.nb_wait(self._h), self._lib)

    def sync(self) -> np.ndarray:
        L.check("nb_sync", self._lib.nb_sync(self._h, self.bodies.ctypes.data), self._lib)
        return self.bodies

    def snapshot_begin(self, out: Optional[np.ndarray] = None) -> np.ndarray:
        """Pipelined ``sync``: start copying the state as of the work enqueued so far into ``out`` (default:
        ``self.bodies``) and return at once; steps enqueued afterwards overlap the transfer.  Pair with ``snapshot_wait``."""
        out = self.bodies if out is None else out
        L.check("nb_snapshot_begin", self._lib.nb_snapshot_begin(self._h, out.ctypes.data), self._lib)
        return out

    def snapshot_wait(self) -> None:
        L.check("nb_snapshot_wait", self._lib.nb_snapshot_wait(self._h), self._lib)

    def positions(self) -> np.ndarray:
        out = np.empty((self.i_count, 3 if self._params.dims == 3 else 2), dtype=np.float32)
        L.check("nb_sync_positions", self._lib.nb_sync_positions(self._h, out.ctypes.data), self._lib)
        return out

    def upload(self, bodies: np.ndarray) -> None:
        if bodies.dtype != L.BODY_DTYPE or bodies.shape[0] != self.n:
            raise ValueError("upload needs the n bodies of the whole system")
        bodies = np.ascontiguousarray(bodies)
        L.check("nb_upload", self._lib.nb_upload(self._h, bodies.ctypes.data), self._lib)

    def accelerations(self) -> np.ndarray:
        """Evaluate a(x) at the current positions (``attract()`` as a direct sum)."""
        L.check("nb_accelerations", self._lib.nb_accelerations(self._h), self._lib)
        return self.sync()["acc"].copy()

    def energy(self) -> tuple:
        k, u = C.c_double(), C.c_double()
        L.check("nb_energy", self._lib.nb_energy(self._h, C.byref(k), C.byref(u)), self._lib)
        return k.value, u.value

    def momentum(self) -> tuple:
        """((px, py, pz), Lz): total linear momentum (``Body::momentum``, Body.hpp:103-106, summed; fp64 on the
        device) and angular momentum about the origin of the owned block."""
        p, lz = (C.c_double * 3)(), C.c_double()
        L.check("nb_momentum", self._lib.nb_momentum(self._h, p, C.byref(lz)), self._lib)
        return (p[0], p[1], p[2]), lz.value

    def dump(self, path: str) -> None:
        L.check("nb_dump", self._lib.nb_dump(self._h, str(path).encode()), self._lib)

    # split step for sharded handles
    def step_begin(self, dt: Optional[float] = None) -> None:
        L.check("nb_step_begin", self._lib.nb_step_begin(self._h, SIMULATION_DT if dt is None else dt), self._lib)

    def step_mid(self) -> None:
        L.check("nb_step_mid", self._lib.nb_step_mid(self._h), self._lib)

    def step_finish(self) -> None:
        L.check("nb_step_finish", self._lib.nb_step_finish(self._h), self._lib)

    def pos_buffer(self, which: int = L.NB_POS_CURRENT) -> int:
        return int(self._lib.nb_pos_buffer(self._h, which) or 0)

    @property
    def shard_protocol(self) -> int:
        return int(self._lib.nb_shard_protocol(self._h))

    def acc_buffer(self, which: int) -> int:
        return int(self._lib.nb_acc_buffer(self._h, which) or 0)

    @property
    def stream(self) -> int:
        return int(self._lib.nb_stream(self._h) or 0)

    def profile(self, on: bool = True) -> None:
        L.check("nb_profile_enable", self._lib.nb_profile_enable(self._h, int(on)), self._lib)

    def profile_read(self, reset: bool = True) -> tuple:
        ms, cnt = C.c_double(), C.c_uint64()
        L.check("nb_profile_read", self._lib.nb_profile_read(self._h, C.byref(ms), C.byref(cnt), int(reset)), self._lib)
        return ms.value, int(cnt.value)

    def sym_info(self) -> dict:
        """Figures of the symmetric work plan (``nb_sym_plan_info``): items, chunks per item, slab bytes ..."""
        info = L.nb_sym_info()
        info.struct_size = C.sizeof(L.nb_sym_info)
        L.check("nb_sym_plan_info", self._lib.nb_sym_plan_info(self._h, C.byref(info)), self._lib)
        return info.as_dict()

    def describe(self) -> str:
        buf = C.create_string_buffer(1024)
        L.check("nb_describe", self._lib.nb_describe(self._h, buf, len(buf)), self._lib)
        return buf.value.decode()

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.nb_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def write_bodies(path: str, bodies: np.ndarray, frame: int = 0, eps: float = 1.0, dt: float = SIMULATION_DT) -> None:
    p = L.default_params()
    p.eps, p.dt = eps, dt
    bodies = np.ascontiguousarray(bodies)
    L.check("nb_write_bodies", L.load().nb_write_bodies(str(path).encode(), bodies.ctypes.data, bodies.shape[0], frame, C.byref(p)))


def read_bodies(path: str):
    """Return (bodies, frame, params) of a dump written by nb_dump / nb_write_bodies."""
    lib = L.load()
    n, frame, p = C.c_size_t(), C.c_uint64(), L.nb_params()
    L.check("nb_read_header", lib.nb_read_header(str(path).encode(), C.byref(n), C.byref(frame), C.byref(p)))
    out = L.bodies_array(n.value)
    L.check("nb_read_bodies", lib.nb_read_bodies(str(path).encode(), out.ctypes.data, n.value))
    return out, int(frame.value), p
