"""Test infrastructure: the -DNB_TEST_HOOKS build of the library (tests/libnbody_hip_testhooks.so, `make -C
nbodysim_amd/csrc hooks`, built by __graft_entry__.build()) and the host-side wrappers of its nb_debug_* functions
(include/nbody_debug.h).  The PRODUCT library exports none of these (tests/test_abi.py); tests that need a hook bind this
build next to it — everything that is not a hook still runs against nbodysim_amd/libnbody_hip.so."""
import ctypes as C
from pathlib import Path

import numpy as np

from nbodysim_amd import _lib as L

HOOKS_PATH = Path(__file__).resolve().parent / "libnbody_hip_testhooks.so"

DEBUG_PROTOTYPES = {
    "nb_debug_comm_transport": (C.c_int, [C.c_char_p]),
    "nb_debug_comm_schedule": (C.c_int, [C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "nb_debug_sym_plan": (C.c_int, [C.c_size_t, C.c_int, C.c_int, C.c_int, C.POINTER(L.nb_params), C.c_void_p, C.c_size_t, C.POINTER(L.nb_sym_info)]),
    "nb_debug_ticket_seed": (C.c_int, [C.c_void_p, C.c_uint32]),
    "nb_debug_fast_inv_sqrt": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
}

_hooks = None


def lib() -> C.CDLL:
    """The hooks build, bound with the product prototypes plus the debug ones."""
    global _hooks
    if _hooks is None:
        _hooks = L.bind(HOOKS_PATH, {**L.PROTOTYPES, **DEBUG_PROTOTYPES})
    return _hooks


def check(where: str, rc: int) -> None:
    L.check(where, rc, lib())


def sym_plan(n: int, cus: int = 256, rank: int = 0, world: int = 1, tuning: "L.nb_params | None" = None):
    """Host-only planner view: (items as a SYM_ITEM_DTYPE array, info dict).  No GPU needed."""
    h = lib()
    info = L.nb_sym_info()
    info.struct_size = C.sizeof(L.nb_sym_info)
    tp = C.byref(tuning) if tuning is not None else None
    check("nb_debug_sym_plan", h.nb_debug_sym_plan(n, cus, rank, world, tp, None, 0, C.byref(info)))
    items = np.zeros(info.items, L.SYM_ITEM_DTYPE)
    check("nb_debug_sym_plan", h.nb_debug_sym_plan(n, cus, rank, world, tp, items.ctypes.data, info.items, C.byref(info)))
    return items, info.as_dict()


def comm_schedule(protocol: int, handles: int, block_reals: int, full_reals: int, ag_pending: bool) -> np.ndarray:
    """Host-only: the operations ``nb_comm_step`` issues for one step (``nb_debug_comm_schedule``), as a COMM_OP_DTYPE array."""
    h = lib()
    cnt = C.c_size_t()
    check("nb_debug_comm_schedule", h.nb_debug_comm_schedule(protocol, handles, block_reals, full_reals, int(ag_pending), None, 0, C.byref(cnt)))
    ops = np.zeros(cnt.value, L.COMM_OP_DTYPE)
    check("nb_debug_comm_schedule", h.nb_debug_comm_schedule(protocol, handles, block_reals, full_reals, int(ag_pending),
                                                             ops.ctypes.data, cnt.value, C.byref(cnt)))
    return ops


def default_params() -> "L.nb_params":
    p = L.nb_params()
    lib().nb_params_default(C.byref(p))
    return p
