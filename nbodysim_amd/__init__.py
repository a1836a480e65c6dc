"""nbodysim_amd — MI355X (gfx950) drop-in for the reference's pairwise-gravity +
kick/drift path (7IBBE77S/nbodysim, Simulation::step()).

The product is ``libnbody_hip.so`` (C ABI in ``include/nbody.h``); this package
is the thin host-side mirror of the reference's ``Simulation`` interface used by
the tests and ``bench.py``.  Importing the package does not load the library;
the first use does, and raises if it has not been built.
"""
from ._lib import (  # noqa: F401
    BODY3_DTYPE,
    BODY_DTYPE,
    NBodyError,
    PinnedBodies,
    bodies_array,
    load,
    default_ics,
    plummer_2d,
    plummer_3d,
)
from .simulation import Simulation, read_bodies, write_bodies  # noqa: F401

__all__ = [
    "BODY3_DTYPE",
    "BODY_DTYPE",
    "NBodyError",
    "PinnedBodies",
    "Simulation",
    "bodies_array",
    "load",
    "default_ics",
    "plummer_2d",
    "plummer_3d",
    "read_bodies",
    "write_bodies",
]
