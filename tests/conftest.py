"""Shared fixtures.  `-m gpu` tests need an MI355X; everything else runs on CPU."""
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))

GOLD = ROOT / "tests" / "golden"


def free_port() -> int:
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


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    config.addinivalue_line("markers", "perf: orderings between measured durations (GPU box, -m perf); never part of the correctness suites")


@pytest.fixture(scope="session")
def gold():
    """Golden fixtures produced by the compiled reference (oracle/make_golden.py)."""
    return {p.stem: np.load(p) for p in GOLD.glob("*.npy")}


@pytest.fixture(scope="session")
def nbo():
    import nbo as _nbo
    _nbo.lib()
    return _nbo


def bodies_from_flat(flat):
    """(n,8) x,y,vx,vy,ax,ay,m,r -> 64-byte Body records."""
    from nbodysim_amd import bodies_array
    b = bodies_array(flat.shape[0])
    b["pos"], b["vel"], b["acc"] = flat[:, 0:2], flat[:, 2:4], flat[:, 4:6]
    b["mass"], b["radius"] = flat[:, 6], flat[:, 7]
    return b


def flat_from_bodies(b):
    f = np.zeros((b.shape[0], 8), np.float32)
    f[:, 0:2], f[:, 2:4], f[:, 4:6] = b["pos"], b["vel"], b["acc"]
    f[:, 6], f[:, 7] = b["mass"], b["radius"]
    return f


def max_rel(a, b):
    """max over particles of |a_i - b_i| / |b_i| (2-vector norms): the relative
    measure of SURVEY §8c, used for the north-star 1e-5 tolerance."""
    a = np.asarray(a, np.float64).reshape(len(a), -1)
    b = np.asarray(b, np.float64).reshape(len(b), -1)
    den = np.linalg.norm(b, axis=1)
    den = np.where(den > 0, den, 1.0)
    return float(np.max(np.linalg.norm(a - b, axis=1) / den))
