"""CPU: the host-only code of the library (work planner nb_plan.cpp, initial conditions and dump I/O nb_host.c)
under AddressSanitizer + UndefinedBehaviorSanitizer, driven by nbodysim_amd/csrc/nb_fuzz.cpp: randomised
(n, world, CUs, tuning) exact-once checks of the planner, dump round trips, forged and truncated dump headers.
GPU sanitizers are not available on the pool; this is the CPU build the task rules ask for."""
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
EXE = ROOT / "build" / "asan" / "nb_host_fuzz"


@pytest.fixture(scope="module")
def fuzz_exe():
    r = subprocess.run(["make", "-C", str(ROOT / "nbodysim_amd" / "csrc"), "asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert EXE.exists()
    return EXE


@pytest.mark.parametrize("seed", [1, 20261004])
def test_host_code_is_clean_under_asan_ubsan(fuzz_exe, tmp_path, seed):
    r = subprocess.run([str(fuzz_exe), "250", str(seed), str(tmp_path)], capture_output=True, text=True, timeout=300,
                       env={"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1", "PATH": "/usr/bin:/bin"})
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith("OK planner_cases=250") and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
