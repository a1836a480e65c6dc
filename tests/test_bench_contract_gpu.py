"""GPU: bench.py prints exactly one JSON line with the fields the driver and the judge read."""
import json
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
            "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"]


def run_bench(*extra):
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), *extra], capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_line_contract():
    d = run_bench("--n", "32768", "--steps", "4", "--warmup", "1")
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "strong" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and "synthetic" in d["data"] and "workload" in d["config"] and "model" not in d["config"]
    assert d["unit"] == "pair interactions/s"
    assert abs(d["value"] - 32768.0 ** 2 * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) < 1e-6 * d["value"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf
    assert rf["unit"] == "TFLOP/s" and rf["peak"] == 157.3 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert rf["launches"] == 4 and rf["avg_launch_ms"] > 0          # HIP events around every force launch of the timed region
    # the whole roofline truth: the symmetric kernel issues fewer flops than the algorithmic count it delivers
    assert rf["kernel"] == "force_sym_f32" and rf["kernel"] in d["config"]["workload"] and "LDS tile" not in d["config"]["workload"]
    assert 0 < rf["executed_frac"] < rf["frac"] and abs(rf["executed_frac"] - rf["executed_tflops"] / rf["peak"]) < 1e-9
    assert abs(rf["executed_frac"] / rf["frac"] - 17.0 / 28.0) < 0.04             # 17 flop per unordered pair vs 2 x 14 algorithmic (+ the one-sided diagonal items)
    # counter figures are COUNTER figures or null, never plan-derived (the plan's estimate has its own name).  No committed PMC record exists
    # for N = 32 768, so what the line carries is this run's own LIVE collection (three short rocprofv3 --pmc child runs after the timed
    # region) — or, where the profiler could not run, nulls and a status that says why
    assert rf["traffic_plan"] > 36 * 32768 and rf["kernel_hbm_gbps_plan"] > 0
    if rf["pmc_status"] == "live":
        assert 0.3 < rf["valu_busy"] < 1.0 and "LIVE" in rf["valu_busy_source"] and rf["valu_busy_committed"] is None
        assert rf["traffic"] == rf["traffic_pmc"] and 0.5 * rf["traffic_plan"] < rf["traffic"] < 3.0 * rf["traffic_plan"] and "LIVE" in rf["traffic_source"]
        assert rf["kernel_hbm_gbps"] > 0 and 3.5 < rf["valu_cycles_per_inst"] < 8.0
    else:
        assert rf["pmc_status"].startswith("none") and ("live collection" in rf["pmc_status"] or "rocprofv3 not found" in rf["pmc_status"]), rf["pmc_status"]
        assert rf["traffic"] is None and rf["traffic_pmc"] is None and rf["kernel_hbm_gbps"] is None and rf["valu_busy"] is None
    # the settled fraction next to the burst one: from the force launches of the >= 2 s stretch after the timed region
    assert 0 < rf["frac_sustained"] < 1.2 and "settled" in rf["frac_sustained_kind"]
    assert abs(rf["frac_sustained"] - 14.0 * 32768.0 ** 2 / (d["sustained"]["avg_launch_ms"] * 1e-3) / 1e12 / 157.3) < 1e-9
    # PMC provenance: the line names the exact instantiation that ran; counter figures appear only with a record of exactly it
    assert rf["kernel_instantiation"] == "nbk::force_sym_f32<0, 0, false, true>" and rf["pmc_commit"] is None
    assert abs(rf["arithmetic_intensity_flop_per_byte"] - 14.0 * 32768.0 / 36.0) < 1e-6
    assert abs(rf["arithmetic_intensity_vs_traffic"] - 14.0 * 32768.0 ** 2 / (rf["traffic"] or rf["traffic_plan"])) < 1e-6 * rf["arithmetic_intensity_vs_traffic"]
    # what the reference's caller pays per frame through the drop-in: PCIe-inclusive, beside `value`, never in it
    fm = d["frame_ms"]
    assert {"step", "step_copy", "overlapped_copy", "resident", "frames", "bytes_per_frame"} <= set(fm) and fm["bytes_per_frame"] == 64 * 32768
    assert all(fm[k] > 0 for k in ("step", "step_copy", "overlapped_copy", "resident"))
    g = rf["general_mass"]
    # individual masses: 12 + 2 ops per body instead of 10 + 2.  WHICH kernel ran is asserted (its template instantiation: MM_GENERAL = 1,
    # mass-scaled = 2), never how long it took: durations are recorded fields; their ordering is a `-m perf` matter (tests/test_perf_order.py)
    assert g["kernel_instantiation"] == "nbk::force_sym_f32<0, 1, false, true>" and g["avg_launch_ms"] > 0 and g["frac"] > 0
    assert rf["frac_equal_masses"] == rf["frac"] and abs(rf["frac_individual_masses"] - g["frac"]) < 1e-12
    assert {"nproc", "affinity", "cgroup_cpus", "model"} <= set(d["cpu_baseline"]["host"])
    t = rf["one_sided_lds_tiled"]
    assert t["kernel"] == "force_tiled_f32" and t["kernel_instantiation"] is None and t["avg_launch_ms"] > 0 and t["frac"] > 0
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0
    if cb["kind"] == "reference":     # the prebuilt oracle/_ref/libnbref.so travelled here: the reference's own loop, port beside it
        assert cb["bitwise_equal_to_port"] is True and cb["port"]["value"] > 0 and cb["value"] > 0
    assert abs(d["energy"]["rel_drift"]) < 1e-3
    # burst vs settled: >= 2 s of steps AFTER the timed region, with their own clock / power samples; never part of `value`
    su = d["sustained"]
    assert su["steps"] >= 20 and su["seconds"] >= 1.9 and su["ms_per_step"] > 0 and 0 < su["frac"] < 1.2
    assert abs(su["value"] - 32768.0 ** 2 * su["steps"] / su["seconds"]) < 1e-6 * su["value"]
    sc = rf["general_mass_scaled"]
    assert sc["mass_scaled"] is True and g["mass_scaled"] is False and sc["avg_launch_ms"] > 0
    assert sc["kernel_instantiation"] == "nbk::force_sym_f32<0, 2, false, true>"
    # folding is the caller's decision since ABI 6: the default with individual masses is the 12 + 2 body; the MEASURED rule is opt-in
    # (scaled vs unscaled accelerations of these bodies, 2e-6 of the force scale).  At the headline size the light equal masses pass
    # (4e-7: tests/test_headline_gpu.py); at this small N with eps = 0.01 the closest pairs dominate a body's force and the verdict may
    # go either way — what is asserted is that the kernel that ran follows the figure
    au = rf["general_mass_measured"]
    assert au["mass_scaling_check"] is not None and au["mass_scaling_check"] >= 0
    assert au["mass_scaled"] is (au["mass_scaling_check"] <= 2e-6)
    assert au["kernel_instantiation"] == (sc if au["mass_scaled"] else g)["kernel_instantiation"]
    assert g["mass_scaling_check"] is None and abs(rf["frac_individual_masses_default"] - g["frac"]) < 1e-12
    if d["device_state"]:                       # hwmon files readable on this box
        assert {"at_start", "at_end", "sclk_mhz_mean", "power_w_mean"} <= set(d["device_state"])


def test_bench_fp64_and_3d_variants_run():
    d = run_bench("--n", "16384", "--steps", "2", "--warmup", "1", "--precision", "fp64", "--no-cpu-baseline", "--no-live-pmc")
    assert d["dtype"] == "f64" and d["roofline"]["peak"] == 157.3 / 2 and "cpu_baseline" not in d
    d = run_bench("--n", "16384", "--steps", "2", "--warmup", "1", "--dims", "3", "--no-cpu-baseline", "--no-live-pmc")
    assert d["config"]["dims"] == 3 and d["roofline"]["flop_per_pair"] == 20.0
    d = run_bench("--n", "16384", "--steps", "2", "--warmup", "1", "--no-symmetry", "--no-cpu-baseline", "--no-live-pmc")
    assert d["roofline"]["kernel"] == "force_tiled_f32" and "LDS tiles of 256" in d["config"]["workload"]
    assert abs(d["roofline"]["executed_frac"] / d["roofline"]["frac"] - 13.0 / 14.0) < 1e-6


def test_bench_line_of_a_two_rank_rehearsal():
    """The N > 1 path of bench.py end to end, as the driver launches it (torch.distributed.run, one process per rank),
    rehearsed with two ranks over gloo on this one GPU (RCCL refuses two ranks per device): the autotune's verdict,
    the per-phase report and the contract fields are all there.  Not a scaling measurement."""
    from conftest import free_port
    port = free_port()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2", "--nbodies", "32768", "--steps", "4", "--warmup", "1",
                        "--backend", "gloo", "--share-gpu"], capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                                  # rank 0 prints the one line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["steps"] == 4 and "cpu_baseline" not in d
    assert d["config"]["protocol"] in ("symmetric", "allreduce", "allgather") and d["config"]["backend"] == "gloo"
    t = d["config"]["protocol_tuning"]
    assert t["chosen"].startswith(d["config"]["protocol"]) and set(t["ms_per_step"]) == {"symmetric", "symmetric+late", "allreduce", "allgather"}
    # self-validating: the sharded trajectory against ONE unsharded handle on rank 0, before and after the timed steps; every start-up
    # candidate validated the same way; the plain all-gather configuration measured first and kept as the fallback line
    pc = d["parity_check"]
    assert pc["ok"] is True and pc["steps"] == 1 and pc["max_rel_pos"] < 1e-5 and pc["max_rel_vel"] < 1e-5
    assert pc["after_timed_region"]["ok"] is True and pc["after_timed_region"]["steps"] == 5
    assert t["failed"] == {} and all(v["ok"] for v in t["validation"].values()) and set(t["validation"]) == set(t["ms_per_step"])
    assert d["fallback"] == {"used": False} and d["config"]["safe_first"]["parity_check"]["ok"] is True and d["config"]["safe_first"]["ms_per_step"] > 0
    ph = d["phases_ms"]
    assert ph["rank0"]["steps"] == 4 and ph["max_over_ranks"]["stream_total"] > 0
    assert abs(d["value"] - 32768.0 ** 2 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert abs(d["energy"]["rel_drift"]) < 1e-3


def test_bench_single_rank_rehearsal_of_the_node_flow_through_rccl():
    """`bench.py --rehearse-sharded`: the WHOLE multi-rank flow with the one rank a one-GPU box allows, through RCCL itself (a
    process group of one, NB_FLAG_SHARD_SINGLE handles): safe-first measurement, start-up timing of every protocol under the
    torch-driven loop, the winner's full measurement, then the library's C loop as ONE challenger on the winning protocol (its own
    RCCL communicator; staged since round 6), self-checks before and after the timed steps of each, and the N > 1 form of the line."""
    d = run_bench("--rehearse-sharded", "--n", "32768", "--steps", "4", "--warmup", "1", "--no-sustained")
    assert d["n_gpus"] == 1 and d["config"]["backend"] == "nccl" and d["config"]["parallelism"] == "i-block x1" and "cpu_baseline" not in d
    t = d["config"]["protocol_tuning"]
    timed = {k for k, v in t["ms_per_step"].items() if v is not None}
    assert {"allgather", "allreduce", "symmetric"} <= timed and not any(k.startswith("c:") for k in t["ms_per_step"]) and t["failed"] == {}
    assert all(t["validation"][k]["ok"] for k in timed)
    ch = d["config"]["c_loop_challenger"]
    assert ch["driver"] == "c" and ch["protocol"] == d["config"]["protocol"] and ch["parity_check"]["ok"] is True and ch["ms_per_step"] > 0
    assert d["config"]["driver"] == ("c" if ch["won"] else "torch") and d["config"]["driver_choice"].startswith(d["config"]["driver"] + ":")
    assert "C loop" in d["config"]["driver_choice"] and "torch-driven" in d["config"]["driver_choice"]
    if ch["won"]:
        assert d["config"]["torch_driven"]["ms_per_step"] > 0 and d["value"] >= 0.99 * d["config"]["torch_driven"]["value"]
    pc = d["parity_check"]
    assert pc["ok"] is True and pc["max_rel_pos"] < 1e-5 and pc["after_timed_region"]["ok"] is True and pc["after_timed_region"]["steps"] == 5
    sf = d["config"]["safe_first"]
    assert sf["protocol"] == "allgather" and sf["driver"] == "torch" and sf["parity_check"]["ok"] is True
    assert d["fallback"] == {"used": False} and d["phases_ms"]["rank0"]["steps"] == 4
    assert abs(d["value"] - 32768.0 ** 2 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]


def test_bench_gpus_2_as_typed_starts_its_own_ranks():
    """`python3 bench.py --gpus 2 ...` with NO launcher — the form a driver's scaling run may well use.  The
    process starts the two ranks as children (torch.distributed.run), relays rank 0's line and returns their status.  Two ranks over
    gloo on this one GPU (RCCL refuses two ranks per device): a rehearsal of the launch path, not a scaling measurement."""
    d = run_bench("--gpus", "2", "--backend", "gloo", "--share-gpu", "--steps", "4", "--warmup", "1", "--nbodies", "32768", "--no-sustained")
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 1 and d["scaling"] == "strong" and "cpu_baseline" not in d
    assert d["parity_check"]["ok"] is True and d["parity_check"]["after_timed_region"]["ok"] is True
    assert d["fallback"] == {"used": False} and d["config"]["safe_first"]["parity_check"]["ok"] is True
    assert d["config"]["backend"] == "gloo" and d["config"]["protocol"] in ("symmetric", "allreduce", "allgather")
    assert abs(d["value"] - 32768.0 ** 2 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]


def test_bench_gpus_2_in_one_process_through_the_librarys_own_loop():
    """The fallback where torch.distributed.run is missing (forced here with --one-process): ONE process, two sharded handles, no
    torch, no process group.  On a node the handles sit on two devices and the library's RCCL loop (nb_comm_create_all / nb_comm_step)
    runs them; on this one GPU they share the device and the library's in-process exchange stands in for the transport — same
    kernels, same pair split, same self-check against one unsharded handle, same line."""
    d = run_bench("--gpus", "2", "--one-process", "--share-gpu", "--steps", "4", "--warmup", "1", "--nbodies", "32768", "--no-sustained")
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["scaling"] == "strong"
    assert d["parity_check"]["ok"] is True and d["parity_check"]["max_rel_pos"] < 1e-5 and d["parity_check"]["after_timed_region"]["steps"] == 5
    assert d["fallback"] == {"used": False} and d["config"]["safe_first"]["parity_check"]["ok"] is True
    assert d["config"]["safe_first"]["protocol"] == "allgather" and "in-process" in d["config"]["driver"] and "in-process" in d["config"]["backend"]
    t = d["config"]["protocol_tuning"]
    assert t["chosen"] == d["config"]["protocol"] and set(t["ms_per_step"]) == {"allgather", "allreduce", "symmetric"}
    assert abs(d["value"] - 32768.0 ** 2 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert abs(d["energy"]["rel_drift"]) < 1e-3


def test_launched_ranks_that_cannot_form_a_group_fall_back_to_one_process():
    """The whole last-resort chain of `bench.py --gpus N` typed without a launcher, end to end: on this one GPU the two launched ranks
    cannot form an RCCL process group (RCCL: "Duplicate GPU detected") and end without a line; the launcher tries once more on
    another port, then ONE process drives both handles itself (here over the in-process exchange: the handles share the GPU) — and
    the run ends with a validated line and status 0.  On a node the same chain covers a process group that will not form."""
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--share-gpu", "--steps", "4", "--warmup", "1", "--nbodies", "32768",
                        "--no-sustained", "--deadline", "120"], capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and r.stdout.strip() == lines[0]                     # nothing but the line on stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["parity_check"]["ok"] is True and d["fallback"] == {"used": False} and "in-process" in d["config"]["backend"]
    assert r.stderr.count("starting 2 ranks as child processes") == 2 and "one more attempt" in r.stderr
    assert "falling back to ONE process driving all 2 handles" in r.stderr
