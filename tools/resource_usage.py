#!/usr/bin/env python3
"""Registers / LDS / occupancy of the library's kernels from `make -C nbodysim_amd/csrc asm` (build/asm/resource_usage.txt)."""
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
pat = sys.argv[1] if len(sys.argv) > 1 else "force_sym"
txt = (ROOT / "build" / "asm" / "resource_usage.txt").read_text()
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split("\n")[0].strip()
    try:
        name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip() or name
    except OSError:
        pass
    if pat not in name:
        continue
    g = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]
    scratch, occ, lds = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")
    print(f"{name.split('(')[0][:70]:70s} VGPR {g('VGPRs'):>3} SGPR {g('SGPRs'):>3} scratch {scratch} waves/SIMD {occ} LDS {lds}")
