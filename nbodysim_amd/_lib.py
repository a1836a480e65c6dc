"""ctypes binding of ``libnbody_hip.so`` (the C ABI declared in ``include/nbody.h``).

The binding is deliberately thin: it declares the structs and prototypes and
fails loudly.  There is no Python or CPU fallback for any compute entry point —
if the shared library is missing ``load()`` raises, and on a machine without a
HIP device ``nb_create`` returns NULL (``NB_ENODEVICE``), which the wrappers
turn into ``NBodyError``.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

PKG_DIR = Path(__file__).resolve().parent
LIB_PATH = PKG_DIR / "libnbody_hip.so"

NB_ABI_VERSION = 6

# enums (include/nbody.h)
NB_OK, NB_EINVAL, NB_ENODEVICE, NB_EHIP, NB_ENOMEM, NB_EIO, NB_EFORMAT, NB_ESTATE = 0, -1, -2, -3, -4, -5, -6, -7
NB_FP32, NB_FP64 = 0, 1
NB_RSQRT_EXACT, NB_RSQRT_QUAKE = 0, 1
NB_SUM_TILED, NB_SUM_SEQUENTIAL = 0, 1
NB_EXTRA_VCLAMP, NB_EXTRA_BOUNDARY = 1, 2
NB_INTEGRATOR_KICK_DRIFT, NB_INTEGRATOR_KDK = 0, 1
NB_POS_CURRENT, NB_POS_NEXT = 0, 1
NB_SHARD_NONE, NB_SHARD_ALLGATHER, NB_SHARD_SYMMETRIC, NB_SHARD_ALLREDUCE = 0, 1, 2, 3
NB_FLAG_NO_SYMMETRY, NB_FLAG_NO_UNIFORM_MASS, NB_FLAG_NO_GUIDED_TAIL, NB_FLAG_SHARD_ALLREDUCE, NB_FLAG_SHARD_SINGLE, NB_FLAG_MASS_SCALING = 1, 2, 4, 8, 16, 32
NB_FLAG_STATIC_ITEMS = 256
NB_FLAG_NO_MASS_SCALING = 512
NB_FLAG_MASS_SCALING_MEASURED = 1024

#: numpy view of the reference's 64-byte ``Body`` record (Body.hpp:6-13, Vec2.hpp:17-20)
BODY_DTYPE = np.dtype(
    {
        "names": ["pos", "vel", "acc", "mass", "radius"],
        "formats": [(np.float32, 2), (np.float32, 2), (np.float32, 2), np.float32, np.float32],
        "offsets": [0, 16, 32, 48, 52],
        "itemsize": 64,
    }
)


#: the same 64 bytes seen by a dims = 3 handle: z of pos / vel / acc in the first padding float
BODY3_DTYPE = np.dtype(
    {
        "names": ["pos", "vel", "acc", "mass", "radius"],
        "formats": [(np.float32, 3), (np.float32, 3), (np.float32, 3), np.float32, np.float32],
        "offsets": [0, 16, 32, 48, 52],
        "itemsize": 64,
    }
)


class NBodyError(RuntimeError):
    """Raised for every non-zero status / NULL handle coming out of the C ABI."""

    def __init__(self, where: str, code: int, text: str):
        super().__init__(f"{where}: [{code}] {text}")
        self.code = code


class nb_params(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("eps", C.c_float),
        ("dt", C.c_float),
        ("precision", C.c_int32),
        ("rsqrt_mode", C.c_int32),
        ("sum_order", C.c_int32),
        ("integrator", C.c_int32),
        ("extras", C.c_int32),
        ("device", C.c_int32),
        ("j_slices", C.c_int32),
        ("i_begin", C.c_uint64),
        ("i_count", C.c_uint64),
        ("stream", C.c_void_p),
        ("pos_buffers", C.c_void_p * 2),
        ("shard_rank", C.c_int32),
        ("shard_world", C.c_int32),
        ("acc_buffers", C.c_void_p * 2),
        ("dims", C.c_int32),
        ("flags", C.c_int32),
        ("sym_chunks_per_item", C.c_int32),
        ("sym_aux_stream", C.c_int32),
        ("sym_late_us", C.c_float),
        ("lanes_p", C.c_int32),
        ("sym_tail", C.c_float * 3),
        ("sym_chunk_pairs", C.c_int32),
        ("first_frame", C.c_uint64),
        ("sym_tile", C.c_int32),
        ("_reserved0", C.c_int32),
        ("pos_rows", C.c_uint64),
    ]


class nb_sym_info(C.Structure):
    """Figures of a handle's symmetric work plan (include/nbody.h)."""

    _fields_ = [
        ("struct_size", C.c_uint32),
        ("enabled", C.c_int32),
        ("chunks_per_item", C.c_uint32),
        ("items", C.c_uint32),
        ("items_local", C.c_uint32),
        ("items_cross", C.c_uint32),
        ("items_late", C.c_uint32),
        ("tiles", C.c_uint32),
        ("rows_s", C.c_uint32),
        ("segments", C.c_uint32),
        ("cus", C.c_uint32),
        ("units_local", C.c_uint64),
        ("units_cross", C.c_uint64),
        ("units_late", C.c_uint64),
        ("cross_units_total", C.c_uint64),
        ("slab_s_bytes", C.c_uint64),
        ("slab_r_bytes", C.c_uint64),
        ("coverage_entries", C.c_uint64),
        ("tile_particles", C.c_uint32),
        ("_reserved0", C.c_uint32),
    ]

    def as_dict(self) -> dict:
        return {k: int(getattr(self, k)) for k, _ in self._fields_ if k != "struct_size" and not k.startswith("_")}


#: numpy view of ``nb_sym_item`` (32 bytes)
SYM_ITEM_DTYPE = np.dtype([("tile", "<u4"), ("c0", "<u4"), ("cnt", "<u4"), ("s_row", "<u4"), ("r_base", "<i8"),
                           ("diag", "<u4"), ("group", "<u4")])
assert SYM_ITEM_DTYPE.itemsize == 32


#: numpy view of ``nb_comm_op`` (24 bytes): one operation of the sharded step's schedule
COMM_OP_DTYPE = np.dtype([("kind", "<i4"), ("handle", "<i4"), ("stream", "<i4"), ("event", "<i4"), ("count", "<u8")])
assert COMM_OP_DTYPE.itemsize == 24
NB_COMM_ID_BYTES = 128
(NB_OP_BEGIN, NB_OP_MID, NB_OP_FINISH, NB_OP_RECORD, NB_OP_WAIT, NB_OP_ALLGATHER, NB_OP_REDUCE_SCATTER, NB_OP_ALLREDUCE,
 NB_OP_GROUP_START, NB_OP_GROUP_END) = range(10)
NB_EV_POS, NB_EV_AG, NB_EV_ACC, NB_EV_RED = range(4)
NB_PH_NAMES = ("local", "ag_wait", "cross", "reduce", "finish")       # NB_PH_* of nb_comm_phase_read
NB_COMM_PHASES = len(NB_PH_NAMES)


#: every symbol include/nbody.h declares: name -> (restype, argtypes)
PROTOTYPES = {
    "nb_params_default": (None, [C.POINTER(nb_params)]),
    "nb_create": (C.c_void_p, [C.c_void_p, C.c_size_t, C.POINTER(nb_params)]),
    "nb_destroy": (None, [C.c_void_p]),
    "nb_step": (C.c_int, [C.c_void_p, C.c_float, C.c_int]),
    "nb_wait": (C.c_int, [C.c_void_p]),
    "nb_sync": (C.c_int, [C.c_void_p, C.c_void_p]),
    "nb_sync_positions": (C.c_int, [C.c_void_p, C.c_void_p]),
    "nb_snapshot_begin": (C.c_int, [C.c_void_p, C.c_void_p]),
    "nb_snapshot_wait": (C.c_int, [C.c_void_p]),
    "nb_host_register": (C.c_int, [C.c_void_p, C.c_size_t]),
    "nb_host_unregister": (C.c_int, [C.c_void_p]),
    "nb_host_alloc": (C.c_void_p, [C.c_size_t]),
    "nb_host_free": (C.c_int, [C.c_void_p]),
    "nb_upload": (C.c_int, [C.c_void_p, C.c_void_p]),
    "nb_accelerations": (C.c_int, [C.c_void_p]),
    "nb_energy": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "nb_momentum": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "nb_frame": (C.c_uint64, [C.c_void_p]),
    "nb_count": (C.c_size_t, [C.c_void_p]),
    "nb_owned_begin": (C.c_size_t, [C.c_void_p]),
    "nb_owned_count": (C.c_size_t, [C.c_void_p]),
    "nb_dump": (C.c_int, [C.c_void_p, C.c_char_p]),
    "nb_write_bodies": (C.c_int, [C.c_char_p, C.c_void_p, C.c_size_t, C.c_uint64, C.POINTER(nb_params)]),
    "nb_read_header": (C.c_int, [C.c_char_p, C.POINTER(C.c_size_t), C.POINTER(C.c_uint64), C.POINTER(nb_params)]),
    "nb_read_bodies": (C.c_int, [C.c_char_p, C.c_void_p, C.c_size_t]),
    "nb_step_begin": (C.c_int, [C.c_void_p, C.c_float]),
    "nb_step_mid": (C.c_int, [C.c_void_p]),
    "nb_step_finish": (C.c_int, [C.c_void_p]),
    "nb_pos_buffer": (C.c_void_p, [C.c_void_p, C.c_int]),
    "nb_stream": (C.c_void_p, [C.c_void_p]),
    "nb_pos_rows": (C.c_size_t, [C.c_void_p]),
    "nb_shard_protocol": (C.c_int, [C.c_void_p]),
    "nb_exchange_positions": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "nb_exchange_accelerations": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "nb_exchange_allreduce": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "nb_acc_buffer": (C.c_void_p, [C.c_void_p, C.c_int]),
    "nb_element_layout": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "nb_device": (C.c_int, [C.c_void_p]),
    "nb_shard_rank": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "nb_comm_unique_id": (C.c_int, [C.c_void_p]),
    "nb_comm_create_rank": (C.c_void_p, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "nb_comm_create_all": (C.c_void_p, [C.POINTER(C.c_void_p), C.c_int]),
    "nb_comm_step": (C.c_int, [C.c_void_p, C.c_float, C.c_int]),
    "nb_comm_flush": (C.c_int, [C.c_void_p]),
    "nb_comm_wait": (C.c_int, [C.c_void_p]),
    "nb_comm_destroy": (None, [C.c_void_p]),
    "nb_comm_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "nb_comm_available": (C.c_int, [C.POINTER(C.c_int)]),
    "nb_comm_profile": (C.c_int, [C.c_void_p, C.c_int]),
    "nb_comm_phase_read": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int]),
    "nb_comm_id_publish": (C.c_int, [C.c_char_p, C.c_uint64, C.c_void_p]),
    "nb_comm_id_await": (C.c_int, [C.c_char_p, C.c_uint64, C.c_void_p, C.c_int]),
    "nb_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "nb_profile_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int]),
    "nb_describe": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t]),
    "nb_plummer_2d": (C.c_int, [C.c_void_p, C.c_size_t, C.c_uint32]),
    "nb_plummer_3d": (C.c_int, [C.c_void_p, C.c_size_t, C.c_uint32]),
    "nb_default_ics": (C.c_int, [C.c_void_p, C.c_size_t]),
    "nb_sym_plan_info": (C.c_int, [C.c_void_p, C.POINTER(nb_sym_info)]),
    "nb_last_error_code": (C.c_int, []),
    "nb_device_count": (C.c_int, []),
    "nb_last_error": (C.c_char_p, []),
    "nb_abi_version": (C.c_int, []),
}

_lib = None


def bind(path, prototypes=None) -> C.CDLL:
    """Load a build of the library from ``path`` and declare ``prototypes`` (default: the product ABI) on it; raises if the
    file is missing, lacks a declared symbol, or was built for another ABI version.  ``load()`` is this for the product
    library; the tests bind their -DNB_TEST_HOOKS build (tests/hooks.py) the same way."""
    path = Path(path)
    if not path.exists():
        raise ImportError(
            f"{path} not found — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C nbodysim_amd/csrc` (there is no CPU fallback)"
        )
    lib = C.CDLL(str(path))
    for name, (restype, argtypes) in (PROTOTYPES if prototypes is None else prototypes).items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.nb_abi_version() != NB_ABI_VERSION:
        raise ImportError(f"{path}: ABI version {lib.nb_abi_version()} != binding {NB_ABI_VERSION}")
    return lib


def load() -> C.CDLL:
    """Load the product library; raise if it is missing or its ABI differs."""
    global _lib
    if _lib is None:
        _lib = bind(os.environ.get("NBODY_HIP_LIB", LIB_PATH))
    return _lib


def last_error(lib=None) -> str:
    return ((lib or load()).nb_last_error() or b"").decode("utf-8", "replace")


def last_error_code(lib=None) -> int:
    return int((lib or load()).nb_last_error_code())


def check(where: str, rc: int, lib=None) -> None:
    """Raise ``NBodyError`` for a non-zero status; ``lib`` = the build the call went to (its thread-local error text)."""
    if rc != NB_OK:
        raise NBodyError(where, rc, last_error(lib))


def default_params() -> nb_params:
    p = nb_params()
    load().nb_params_default(C.byref(p))
    return p


def bodies_array(n: int) -> np.ndarray:
    """Zero-initialised array of n 64-byte Body records (padding zero)."""
    raw = np.zeros(n * BODY_DTYPE.itemsize, dtype=np.uint8)
    return raw.view(BODY_DTYPE)


class PinnedBodies:
    """n Body records in page-locked memory owned by the library (``nb_host_alloc``): the destination that
    ``nb_sync`` / ``nb_snapshot_begin`` DMA into directly.  ``.array`` is the numpy view; ``close()`` (or the
    context manager) releases the block — the view must not be used afterwards."""

    def __init__(self, n: int):
        lib = load()
        self.nbytes = n * BODY_DTYPE.itemsize
        self._ptr = lib.nb_host_alloc(self.nbytes)
        if not self._ptr:
            raise NBodyError("nb_host_alloc", last_error_code(), last_error())
        buf = (C.c_uint8 * self.nbytes).from_address(self._ptr)
        self.array = np.frombuffer(buf, dtype=BODY_DTYPE)
        self.array[:] = np.zeros((), BODY_DTYPE)

    def close(self) -> None:
        if self._ptr:
            self.array = None
            check("nb_host_free", load().nb_host_free(self._ptr))
            self._ptr = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def plummer_3d(n: int, seed: int = 42) -> np.ndarray:
    """True 3-D Plummer sphere; view it with ``.view(BODY3_DTYPE)`` to reach z."""
    out = bodies_array(n)
    check("nb_plummer_3d", load().nb_plummer_3d(out.ctypes.data, n, seed))
    return out


def default_ics(n: int = 25000) -> np.ndarray:
    """The bodies the reference's ``Simulation()`` starts from (Simulation.hpp:58-65, uniform_disc :347-603)."""
    out = bodies_array(n)
    check("nb_default_ics", load().nb_default_ics(out.ctypes.data, n))
    return out


def plummer_2d(n: int, seed: int = 42) -> np.ndarray:
    """Synthetic workload of SURVEY §8d through the library's generator."""
    out = bodies_array(n)
    check("nb_plummer_2d", load().nb_plummer_2d(out.ctypes.data, n, seed))
    return out
