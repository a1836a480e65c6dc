#!/usr/bin/env python3
"""tools/summarize_profile.py — turn the rocprofv3 outputs merged back under gpurun_out/
into the small, committed summaries under profiles/.

    python tools/summarize_profile.py r01 [N of the profiled command] [traffic|notraffic] [directory of the --pmc passes]

Reads  gpurun_out/prof/**/_kernel_stats.csv     (rocprofv3 --kernel-trace --stats)
       gpurun_out/pmc/*/**/_counter_collection.csv  (separate --pmc passes)
Writes profiles/<round>_kernel_stats.csv, profiles/<round>_pmc_summary.json and
       an entry of profiles/hbm_traffic.json (read by bench.py for roofline.traffic): one per (exact kernel
       instantiation, N), with the commit id the counters were taken at.
HBM bytes follow MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are in KiB-units of
1024 B... (rocprofv3 reports them in kilobytes); on gfx950 FETCH_SIZE under-reports wide
coalesced reads by 2x, so reads are doubled; WRITE_SIZE is taken as is.
"""
import collections
import csv
import glob
import json
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
n_run = int(sys.argv[2]) if len(sys.argv) > 2 else 262144          # N of the profiled bench.py command
write_traffic = (sys.argv[3] if len(sys.argv) > 3 else "traffic") == "traffic"   # "notraffic": leave profiles/hbm_traffic.json alone
pmc_dir = sys.argv[4] if len(sys.argv) > 4 else "gpurun_out/pmc"                 # where the --pmc passes of this workload were written
out = ROOT / "profiles"
out.mkdir(exist_ok=True)

stats = sorted(glob.glob(str(ROOT / "gpurun_out/prof/**/*_kernel_stats.csv"), recursive=True))
if stats:
    shutil.copy(stats[-1], out / f"{tag}_kernel_stats.csv")
    print("kernel stats ->", out / f"{tag}_kernel_stats.csv")

summary = {}
meta = {}
for f in sorted(glob.glob(str(ROOT / pmc_dir / "*/**/*counter_collection.csv"), recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta.setdefault(k, {x: r[x] for x in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "SGPR_Count", "Scratch_Size") if x in r})
    for k, d in agg.items():
        for c, v in d.items():
            summary.setdefault(k, {})[c] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}

derived = {}
for k, d in summary.items():
    g = lambda c: d.get(c, {}).get("mean_per_launch")
    x = {}
    if g("GRBM_GUI_ACTIVE"):
        x["gpu_cycles_per_xcd"] = g("GRBM_GUI_ACTIVE") / 8
    if g("SQ_INSTS_VALU") and g("SQ_ACTIVE_INST_VALU"):
        x["valu_cycles_per_inst"] = 4 * g("SQ_ACTIVE_INST_VALU") / g("SQ_INSTS_VALU")
        if g("GRBM_GUI_ACTIVE"):
            x["valu_busy_frac"] = 4 * g("SQ_ACTIVE_INST_VALU") / 1024 / (g("GRBM_GUI_ACTIVE") / 8)
    if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None and g("TCC_HIT_sum") + g("TCC_MISS_sum") > 0:
        x["l2_hit_rate"] = g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))
    if g("FETCH_SIZE") is not None:
        x["hbm_read_bytes_raw"] = g("FETCH_SIZE") * 1024
        x["hbm_read_bytes_corrected"] = 2 * g("FETCH_SIZE") * 1024
    if g("WRITE_SIZE") is not None:
        x["hbm_write_bytes"] = g("WRITE_SIZE") * 1024
    derived[k] = x

(out / f"{tag}_pmc_summary.json").write_text(json.dumps({"counters": summary, "derived": derived, "dispatch": meta}, indent=1) + "\n")
print("pmc summary ->", out / f"{tag}_pmc_summary.json")
# the dominant force kernel of the run: the symmetric kernel when it ran, else the one-sided tiled kernel.
# profiles/hbm_traffic.json holds ONE ENTRY PER (exact kernel instantiation, N): bench.py reports a PMC figure only for the
# instantiation and size that really ran, and says which commit the counters were taken at.
import subprocess
try:
    commit = subprocess.run(["git", "-C", str(ROOT), "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    dirty = bool(subprocess.run(["git", "-C", str(ROOT), "status", "--porcelain", "--", "nbodysim_amd/csrc"], capture_output=True, text=True).stdout.strip())
except OSError:
    commit, dirty = None, False
force_keys = ([k for k in derived if "force_sym_f32" in k] or [k for k in derived if "force_tiled_f32" in k]) if write_traffic else []
# a run may hold several force kernels (the upload-time mass-scaling check launches the other body once): the run's kernel is the one
# with the most launches
launches_of = lambda k: max((c.get("launches", 0) for c in summary.get(k, {}).values()), default=0)
force_keys.sort(key=launches_of, reverse=True)
tfile = out / "hbm_traffic.json"
try:
    book = json.loads(tfile.read_text())
    if "entries" not in book:
        book = {"entries": []}          # the pre-round-4 single-record form carried neither the instantiation's WS flag nor a commit
except (OSError, ValueError):
    book = {"entries": []}
for k, x in derived.items():
    if k in force_keys[:1] and "hbm_read_bytes_corrected" in x and "hbm_write_bytes" in x:
        t = {"round": tag, "commit": commit, "kernel_sources_dirty": dirty, "kernel": k, "n": n_run,
             "grid_workgroups": int(meta.get(k, {}).get("Grid_Size", 0) or 0) // 256,
             "valu_busy": x.get("valu_busy_frac"), "l2_hit_rate": x.get("l2_hit_rate"),
             "force_kernel_hbm_bytes_per_launch": x["hbm_read_bytes_corrected"] + x["hbm_write_bytes"],
             "read_bytes_raw_FETCH_SIZE": x["hbm_read_bytes_raw"], "read_correction": "x2 (gfx950 FETCH_SIZE counts 64 B per 128-B request)",
             "write_bytes_WRITE_SIZE": x["hbm_write_bytes"]}
        book["entries"] = [e for e in book["entries"] if not (e.get("kernel") == k and e.get("n") == n_run)] + [t]
        tfile.write_text(json.dumps(book, indent=1) + "\n")
        print("hbm traffic ->", t)
