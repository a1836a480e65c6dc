"""CPU: the C-ABI library loads, exports every symbol include/nbody.h declares,
and its host-only entry points behave; compute entry points fail loudly without a GPU."""
import ctypes as C
import json
import re
import subprocess
from pathlib import Path

import numpy as np
import pytest

from conftest import ROOT, flat_from_bodies

import nbodysim_amd as nb
from nbodysim_amd import _lib as L

HEADER = (ROOT / "include" / "nbody.h").read_text()


def declared_symbols():
    # every `name(` at the start of a declaration that begins with nb_
    names = set(re.findall(r"\b(nb_[a-z0-9_]+)\s*\(", HEADER))
    return sorted(names)


def test_header_and_binding_agree():
    syms = declared_symbols()
    assert len(syms) >= 25
    assert set(syms) == set(L.PROTOTYPES), set(syms) ^ set(L.PROTOTYPES)


def dynamic_symbols(path):
    """Every DEFINED symbol of the dynamic table of `path` (nm -D): what a process that loads the library can bind."""
    out = subprocess.run(["nm", "-D", "--defined-only", str(path)], capture_output=True, text=True, check=True).stdout
    return {line.split()[-1] for line in out.splitlines() if line.strip()}


def test_library_exports_exactly_the_declared_c_abi():
    """The product: the dynamic symbol table is the C ABI of include/nbody.h and NOTHING else — no test hook, no kernel
    handle object, no planner internal, no C++ template instantiation (-fvisibility=hidden + nbodysim_amd/csrc/nbody.map)."""
    lib = C.CDLL(str(L.LIB_PATH))
    for name in declared_symbols():
        assert hasattr(lib, name), f"libnbody_hip.so lacks {name}"
    exported = dynamic_symbols(L.LIB_PATH)
    assert exported == set(declared_symbols()), exported ^ set(declared_symbols())
    assert not [s for s in exported if s.startswith("nb_debug")] and "nb_debug" not in HEADER


def test_test_hooks_live_in_their_own_header_and_only_in_the_test_build():
    """include/nbody_debug.h declares the nb_debug_* hooks; only the -DNB_TEST_HOOKS build (tests/libnbody_hip_testhooks.so)
    has them; apart from them the two builds export the same ABI."""
    import hooks
    debug_header = (ROOT / "include" / "nbody_debug.h").read_text()
    debug = set(re.findall(r"\b(nb_debug_[a-z0-9_]+)\s*\(", debug_header))
    assert debug == set(hooks.DEBUG_PROTOTYPES) and len(debug) == 5
    test_build = dynamic_symbols(hooks.HOOKS_PATH)
    assert test_build == set(declared_symbols()) | debug
    assert hooks.lib().nb_abi_version() == L.NB_ABI_VERSION
    integration = (ROOT / "INTEGRATION.md").read_text()
    assert "nb_debug_" not in integration                     # a maintainer binding the library never meets them


def test_library_contains_gfx950_code_object():
    blob = L.LIB_PATH.read_bytes()
    assert b"gfx950" in blob and b"force_tiled_f32" in blob


def test_abi_version_and_defaults():
    lib = nb.load()
    assert lib.nb_abi_version() == L.NB_ABI_VERSION
    p = L.default_params()
    assert p.struct_size == C.sizeof(L.nb_params)
    assert p.eps == 1.0 and abs(p.dt - 0.01) < 1e-9          # Simulation.hpp:59, main.cpp:39
    assert (p.precision, p.rsqrt_mode, p.sum_order, p.extras) == (L.NB_FP32, L.NB_RSQRT_EXACT, L.NB_SUM_TILED, 0)


def test_body_dtype_matches_reference_layout():
    import json
    lay = json.loads((ROOT / "tests" / "golden" / "layout.json").read_text())
    dt = L.BODY_DTYPE
    assert dt.itemsize == lay["sizeof_Body"] == 64
    assert dt.fields["pos"][1] == lay["off_pos"] and dt.fields["vel"][1] == lay["off_vel"]
    assert dt.fields["acc"][1] == lay["off_acc"] and dt.fields["mass"][1] == lay["off_mass"]
    assert dt.fields["radius"][1] == lay["off_radius"]


def test_header_compiles_as_c_and_cxx(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "nbody.h"\nint main(void){nb_params p; nb_params_default(&p); return sizeof(nb_body)==64?0:1;}\n')
    for cc, std in (("gcc", "-std=c11"), ("g++", "-std=c++17")):
        r = subprocess.run([cc, std, "-Wall", "-Werror", "-I", str(ROOT / "include"), "-c", "-x", "c" if cc == "gcc" else "c++",
                            str(src), "-o", str(tmp_path / "t.o")], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_plummer_generator_reproduces_fixture(gold):
    got = flat_from_bodies(nb.plummer_2d(1024, 42))
    assert np.array_equal(got.view(np.uint32), gold["ic_plummer_1024"].view(np.uint32))
    got = flat_from_bodies(nb.plummer_2d(4096, 7))
    assert np.array_equal(got.view(np.uint32), gold["ic_plummer_4096"].view(np.uint32))


def test_plummer_statistics():
    b = nb.plummer_2d(65536, 1)
    assert abs(b["mass"].astype(np.float64).sum() - 1.0) < 1e-6
    assert (b["radius"] == 0).all() and (b["acc"] == 0).all()
    r = np.linalg.norm(b["pos"].astype(np.float64), axis=1)
    # projected Plummer (a=1): half-mass cylinder radius R_h = 1; allow sampling noise
    assert abs(np.median(r) - 1.0) < 0.02
    assert r.max() <= 20.0 + 1e-4
    # 2-D velocity dispersion of a Plummer sphere: <v^2>_3D = 3 pi / 32 -> 2/3 of it in projection
    v2 = (b["vel"].astype(np.float64) ** 2).sum(1).mean()
    assert abs(v2 - (2.0 / 3.0) * 3.0 * np.pi / 32.0) < 0.01
    # padding bytes are zero
    raw = b.view(np.uint8).reshape(-1, 64)
    assert not raw[:, 8:16].any() and not raw[:, 24:32].any() and not raw[:, 40:48].any() and not raw[:, 56:64].any()


def test_dump_format_round_trip(tmp_path):
    b = nb.plummer_2d(1000, 3)
    b["acc"][:, 0] = 1.5
    path = tmp_path / "state.nbd"
    nb.write_bodies(path, b, frame=17, eps=0.05, dt=1e-3)
    raw = path.read_bytes()
    assert len(raw) == 64 + 1000 * 64 and raw[:8] == b"NBODYAMD"
    back, frame, p = nb.read_bodies(path)
    assert frame == 17 and abs(p.eps - 0.05) < 1e-9
    assert back.tobytes() == b.tobytes()


def test_dump_errors(tmp_path):
    lib = nb.load()
    bad = tmp_path / "bad.nbd"
    bad.write_bytes(b"NOTNBODY" + b"\0" * 56)
    n, fr = C.c_size_t(), C.c_uint64()
    assert lib.nb_read_header(str(bad).encode(), C.byref(n), C.byref(fr), None) == L.NB_EFORMAT
    assert b"magic" in lib.nb_last_error()
    assert lib.nb_read_header(str(tmp_path / "missing").encode(), C.byref(n), C.byref(fr), None) == L.NB_EIO
    short = tmp_path / "short.nbd"
    b = nb.plummer_2d(10, 1)
    nb.write_bodies(short, b)
    short.write_bytes(short.read_bytes()[:-100])
    out = nb.bodies_array(10)
    assert lib.nb_read_bodies(str(short).encode(), out.ctypes.data, 10) == L.NB_EFORMAT
    assert lib.nb_last_error_code() == L.NB_EFORMAT and b"bytes on disk" in lib.nb_last_error()
    whole = tmp_path / "whole.nbd"
    nb.write_bodies(whole, b)
    assert lib.nb_read_bodies(str(whole).encode(), out.ctypes.data, 11) == L.NB_EINVAL
    # a forged body count (n * 64 wraps a size_t) is refused by the header check: nothing is sized from it
    raw = bytearray(whole.read_bytes())
    raw[16:24] = (1 << 58).to_bytes(8, "little")
    forged = tmp_path / "forged.nbd"
    forged.write_bytes(bytes(raw))
    assert lib.nb_read_header(str(forged).encode(), C.byref(n), C.byref(fr), None) == L.NB_EFORMAT
    with pytest.raises(nb.NBodyError):
        nb.read_bodies(forged)


def test_create_argument_validation():
    lib = nb.load()
    b = nb.plummer_2d(16, 1)
    p = L.default_params()
    assert not lib.nb_create(None, 16, C.byref(p)) and b"no bodies" in lib.nb_last_error()
    assert lib.nb_last_error_code() == L.NB_EINVAL
    p.struct_size = 4
    assert not lib.nb_create(b.ctypes.data, 16, C.byref(p)) and b"struct_size" in lib.nb_last_error()
    p = L.default_params(); p.precision = 7
    assert not lib.nb_create(b.ctypes.data, 16, C.byref(p))
    p = L.default_params(); p.i_begin, p.i_count = 10, 10
    assert not lib.nb_create(b.ctypes.data, 16, C.byref(p)) and b"exceeds" in lib.nb_last_error()
    p = L.default_params(); p.precision, p.rsqrt_mode = L.NB_FP64, L.NB_RSQRT_QUAKE
    assert not lib.nb_create(b.ctypes.data, 16, C.byref(p))
    for field, bad in (("flags", 2048), ("flags", 64), ("flags", 128), ("extras", 8),     # 64 / 128: the two experimental step fusions of ABI 4, removed; 1024 is an ABI 6 bit
                        ("sym_tile", 1024), ("_reserved0", 1), ("sym_chunks_per_item", -1), ("sym_aux_stream", 2), ("lanes_p", 3), ("j_slices", -2)):
        p = L.default_params(); setattr(p, field, bad)
        assert not lib.nb_create(b.ctypes.data, 16, C.byref(p)) and lib.nb_last_error_code() == L.NB_EINVAL, field
    p = L.default_params(); p.sym_tail[0], p.sym_tail[1], p.sym_tail[2] = 0.9, 0.5, 0.95
    assert not lib.nb_create(b.ctypes.data, 16, C.byref(p)) and b"sym_tail" in lib.nb_last_error()
    with pytest.raises(TypeError):
        nb.Simulation(np.zeros(4, np.float32))


def test_no_cpu_fallback_without_device():
    """On a box without a GPU the product must refuse, not compute on the CPU."""
    lib = nb.load()
    if lib.nb_device_count() > 0:
        pytest.skip("a HIP device is visible here")
    with pytest.raises(nb.NBodyError) as e:
        nb.Simulation(nb.plummer_2d(64, 1))
    assert "no HIP device" in str(e.value)
    assert e.value.code == L.NB_ENODEVICE and lib.nb_last_error_code() == L.NB_ENODEVICE   # a NULL handle still says why


def test_host_memory_and_comm_entry_points_without_a_device():
    """The page-locked-memory and communicator entry points check their arguments first and fail loudly without a GPU
    (no silent pageable fallback behind nb_host_alloc, no other transport behind nb_comm_*)."""
    import mmap
    lib = nb.load()
    if lib.nb_device_count() > 0:
        pytest.skip("a GPU is visible: covered by the gpu suite")
    b = nb.bodies_array(1000)
    odd = b[1:]                                                  # 64 bytes into the array: not page-aligned
    assert lib.nb_host_register(odd.ctypes.data, odd.nbytes) == L.NB_EINVAL and b"whole number" in lib.nb_last_error()
    assert lib.nb_host_register(None, 4096) == L.NB_EINVAL
    assert lib.nb_host_unregister(b.ctypes.data) == L.NB_EINVAL and lib.nb_host_free(b.ctypes.data) == L.NB_EINVAL
    m = mmap.mmap(-1, 2 * mmap.PAGESIZE)
    addr = C.addressof(C.c_char.from_buffer(m))
    assert lib.nb_host_register(addr, 2 * mmap.PAGESIZE) == L.NB_EHIP            # whole pages, but no device to pin them for
    assert lib.nb_host_alloc(4096) is None and lib.nb_last_error_code() == L.NB_ENODEVICE
    assert lib.nb_host_free(None) == L.NB_OK
    p3 = (C.c_double * 3)()
    assert lib.nb_momentum(None, p3, None) == L.NB_EINVAL
    assert lib.nb_element_layout(None, None, None) == L.NB_EINVAL and lib.nb_device(None) == -1
    assert lib.nb_comm_create_rank(None, None, 0, 1) is None and lib.nb_last_error_code() == L.NB_EINVAL
    assert lib.nb_comm_flush(None) == L.NB_EINVAL and lib.nb_comm_info(None, None, None, None, None) == L.NB_EINVAL
    lib.nb_comm_destroy(None)                                    # no-op


def test_product_does_not_reference_oracle():
    """The product path may not import, link or call anything under oracle/."""
    for p in list((ROOT / "nbodysim_amd").rglob("*.py")) + list((ROOT / "nbodysim_amd" / "csrc").glob("*")) + \
            list((ROOT / "nbodysim_amd" / "host").glob("*")):
        if p.is_file() and p.suffix in {".py", ".c", ".h", ".hip", ".cpp", ".hpp", ""}:
            txt = p.read_text(errors="ignore")
            assert "nb_oracle" not in txt and "import nbo" not in txt and "libnbref" not in txt, p
    out = subprocess.run(["ldd", str(L.LIB_PATH)], capture_output=True, text=True).stdout
    assert "oracle" not in out


def test_plummer_3d_is_the_unprojected_2d_sample():
    """nb_plummer_3d keeps z; its (x, y) and (vx, vy) are exactly nb_plummer_2d's."""
    b2 = nb.plummer_2d(5000, 8)
    b3 = nb.plummer_3d(5000, 8).view(nb.BODY3_DTYPE)
    assert np.array_equal(b3["pos"][:, :2], b2["pos"]) and np.array_equal(b3["vel"][:, :2], b2["vel"])
    r = np.linalg.norm(b3["pos"].astype(np.float64), axis=1)
    assert abs(np.median(r) - 1.3048) < 0.03          # Plummer half-mass radius (a = 1): 1.3048
    assert r.max() <= 20.0 + 1e-4 and b3["pos"][:, 2].std() > 0.5
    v2 = (b3["vel"].astype(np.float64) ** 2).sum(1).mean()
    assert abs(v2 - 3.0 * np.pi / 32.0) < 0.01        # <v^2> = 3 pi / 32


def test_dump_header_records_dims(tmp_path):
    import ctypes as C
    b3 = nb.plummer_3d(100, 1)
    p = L.default_params(); p.dims = 3
    assert nb.load().nb_write_bodies(str(tmp_path / "d3").encode(), b3.ctypes.data, 100, 5, C.byref(p)) == 0
    back, frame, q = nb.read_bodies(tmp_path / "d3")
    assert q.dims == 3 and frame == 5 and back.tobytes() == b3.tobytes()
    p.dims = 2
    assert nb.load().nb_write_bodies(str(tmp_path / "d2").encode(), b3.ctypes.data, 100, 5, C.byref(p)) == 0
    back2, _, q2 = nb.read_bodies(tmp_path / "d2")
    assert q2.dims == 2 and not back2.view(nb.BODY3_DTYPE)["pos"][:, 2].any()    # planar dumps zero the padding


def test_default_ics_are_bit_identical_to_the_reference_constructor(gold):
    """nb_default_ics against the bodies the compiled reference's Simulation() holds (fixtures made by
    oracle/make_golden.py from the real constructor): all 25 000 by sha256, the innermost 4096 value by value."""
    import hashlib
    b = nb.default_ics(25000)
    flat = np.zeros((25000, 8), np.float32)
    flat[:, 0:2], flat[:, 2:4], flat[:, 4:6], flat[:, 6], flat[:, 7] = b["pos"], b["vel"], b["acc"], b["mass"], b["radius"]
    want = json.loads((ROOT / "tests" / "golden" / "default_ics.json").read_text())
    assert hashlib.sha256(flat.tobytes()).hexdigest() == want["sha256_float32_le"]
    assert np.array_equal(flat[:4096].view(np.uint32), gold["default_ics_first4096"].view(np.uint32))
    assert [float(v) for v in flat[1]] == want["body1"] and [float(v) for v in flat[-1]] == want["body_last"]
    assert not b.view(np.uint8).reshape(25000, 64)[:, [8, 15, 24, 31, 40, 47, 56, 63]].any()     # padding zeroed
    one = nb.default_ics(1)
    assert one["mass"][0] == 1e9 and one["radius"][0] == 200.0 and not one["pos"].any()
    assert nb.default_ics(0).shape == (0,)
