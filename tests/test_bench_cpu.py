"""CPU: the host-side helpers of bench.py (no GPU): the executed-flop table behind `roofline.executed_frac`, the host
description printed with `cpu_baseline`, and the hwmon sampler degrading to nothing where there is no amdgpu."""
import importlib.util
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def load_bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", ROOT / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    sys.modules["bench_under_test"] = mod
    spec.loader.exec_module(mod)
    return mod


def test_executed_flop_table_matches_the_instruction_counts_of_design_md():
    b = load_bench()
    # fp32 2-D symmetric body = 2 stationary x 1 travelling, both directions:
    # 2 pk_add (4 flop) + 2 pk_fma (8) + 2 rsq (2) + 2 pk_mul (4) + 4 pk_fma (16) = 34 flop per 2 unordered pairs
    assert b.EXECUTED[("fp32", 2)]["sym"][0] == 34 / 2 and b.EXECUTED[("fp32", 2)]["sym"][1] == (34 + 4) / 2
    # one-sided body per 2 ordered pairs: 2 pk_add + 2 pk_fma + 2 rsq + 2 pk_mul (+1 pk_mul for the masses) + 2 pk_fma
    assert b.EXECUTED[("fp32", 2)]["one"] == ((4 + 8 + 2 + 4 + 8) / 2, (4 + 8 + 2 + 6 + 8) / 2)
    # fp64 2-D symmetric body = 1 x 1: 2 add + 2 fma (4) + rsq + cube correction (3 mul + 3 fma = 9) + 4 fma (8)
    assert b.EXECUTED[("fp64", 2)]["sym"][0] == 2 + 4 + 1 + 9 + 8
    # one-sided fp64 body per ordered pair: 2 add + 2 fma + 10 (rsq + cube correction) + mass mul + 2 fma
    assert b.EXECUTED[("fp64", 2)]["one"][1] == 2 + 4 + 10 + 1 + 4
    # the third component adds one sub, one fma for r^2 and one fma per accumulator end
    for prec in ("fp32", "fp64"):
        assert b.EXECUTED[(prec, 3)]["sym"][0] - b.EXECUTED[(prec, 2)]["sym"][0] == 1 + 2 + 4
    # algorithmic (14 per ORDERED pair) always exceeds executed per ordered pair for the symmetric kernels
    assert all(v["sym"][1] / 2 < (14 if dims == 2 else 20) for (prec, dims), v in b.EXECUTED.items())
    assert b.FLOP_PER_PAIR == 14.0 and b.PEAK_FP32_TFLOPS == 157.3 and b.BYTES_PER_PARTICLE_STEP == 36


def test_host_description_and_sampler_without_a_gpu():
    b = load_bench()
    h = b.host_cpu_info()
    assert {"nproc", "affinity", "cgroup_cpus", "model"} <= set(h) and h["nproc"] >= 1
    s = b.DeviceSampler(period_s=0.001)
    s.start()
    out = s.stop()
    assert out is None or {"power_w_mean", "sclk_mhz_mean", "samples"} <= set(out)


def test_pmc_figures_are_reported_only_for_the_kernel_that_ran():
    """roofline.valu_busy / traffic_pmc come from profiles/hbm_traffic.json: a record is used only if it names the exact
    template instantiation, N and item count of this run; records of another kernel or plan at the same N read "stale"."""
    b = load_bench()
    desc = ("n=262144 owned=[0,+262144) fp32 rsqrt=exact sum=tiled | force: block=256 | uniform_mass=1 mass_scaled=0 | "
            "symmetric=1 pipeline=0 tile=2048 chunk_pairs=1 items=8187 chunks/item=44 late=0 slabs=1.0+1.0 MiB | CUs=256")
    k = b.kernel_instantiation(desc, "fp32", 2, "exact")
    assert k == "nbk::force_sym_f32<0, 0, true, false>"
    assert b.kernel_instantiation(desc.replace("uniform_mass=1", "uniform_mass=0"), "fp32", 2, "quake") == "nbk::force_sym_f32<1, 1, true, false>"
    assert b.kernel_instantiation(desc.replace("tile=2048", "tile=512").replace("chunk_pairs=1", "chunk_pairs=0"), "fp32", 2, "exact") == "nbk::force_sym_f32<0, 0, false, true>"
    assert b.kernel_instantiation(desc, "fp64", 2, "exact") is None and b.kernel_instantiation(desc.replace("symmetric=1", "symmetric=0"), "fp32", 2, "exact") is None
    book = {"entries": [{"kernel": k, "n": 262144, "grid_workgroups": 8187, "commit": "abc", "valu_busy": 0.95, "force_kernel_hbm_bytes_per_launch": 3.0e8},
                        {"kernel": "nbk::force_sym_f32<0, 1, false, true>", "n": 25000, "grid_workgroups": 2450, "commit": "abc"}]}
    e, st = b.pmc_lookup(book, k, 262144, 8187)
    assert st == "match" and e["valu_busy"] == 0.95
    e, st = b.pmc_lookup(book, k, 262144, 5629)                               # same kernel, another plan (--chunks-per-item)
    assert e is None and st.startswith("stale") and "5629 items" in st
    e, st = b.pmc_lookup(book, "nbk::force_sym_f32<0, 1, true, false>", 262144, 8187)     # --general-mass: another instantiation
    assert e is None and st.startswith("stale") and "abc" in st
    assert b.pmc_lookup(book, k, 65536, 3428) == (None, "none")
    assert b.pmc_lookup({}, k, 262144, 8187) == (None, "none") and b.pmc_lookup(book, None, 262144, 8187)[0] is None
    # the single-record file of rounds 1-3 (no entries list) is never matched: it names no full instantiation and no commit
    assert b.pmc_lookup({"kernel": "nbk::force_sym_f32<0, 0, true>", "n": 262144}, k, 262144, 8187) == (None, "none")


def test_the_line_says_which_host_loop_won_and_why():
    """config.driver_choice: derived from the start-up timing's record."""
    b = load_bench()

    class Sim:
        pass
    s = Sim()
    s.driver = "c"
    s.tuning = {"ms_per_step": {"allgather": 1.6, "symmetric": 1.05, "c:symmetric": 0.98, "c:allgather": None},
                "validation": {"c:symmetric": {"vs_torch_loop": "bit-identical"}}, "chosen": "c:symmetric", "failed": {"c:allreduce": "x"}}
    why = b.driver_reason(s)
    assert why.startswith("c: C loop 0.980 ms/step vs torch-driven 1.050 ms/step") and "bit-identical" in why and "c:allreduce" in why
    s.driver, s.tuning = "torch", {"ms_per_step": {"allgather": 1.6, "symmetric": 1.05}, "validation": {}, "chosen": "symmetric", "failed": {}}
    assert b.driver_reason(s).startswith("torch: the start-up timing ran over torch-driven candidates")
    s.tuning = None
    assert "named on the command line" in b.driver_reason(s)
    assert b.parse_args([]).driver == "tune" and b.parse_args([]).protocol == "tune"        # the defaults of a node run


def test_unsharded_reference_steps_forward_caches_and_restarts():
    """bench.UnshardedReference: rank 0's checker advances ONE handle forward, caches the rows per step count, and starts a fresh
    handle only when asked for an earlier state than it holds."""
    import numpy as np
    b = load_bench()
    made = []

    class Handle:
        def __init__(self):
            self.frame, self.closed = 0, False
            made.append(self)

        def advance(self, k, dt):
            assert k >= 0
            self.frame += k

        def sync(self):
            out = np.zeros(3, dtype=[("pos", np.float32, 2), ("vel", np.float32, 2)])
            out["pos"][:, 0] = self.frame
            return out

        def close(self):
            self.closed = True
    ref = b.UnshardedReference(Handle)
    assert ref.rows(3)[0, 0] == 3 and len(made) == 1
    assert ref.rows(8)[0, 0] == 8 and len(made) == 1 and made[0].frame == 8          # stepped forward, same handle
    assert ref.rows(3)[0, 0] == 3 and len(made) == 1                                  # cached
    assert ref.rows(5)[0, 0] == 5 and len(made) == 2 and made[0].closed               # earlier than held and not cached: a fresh handle
    assert ref.rows(5).shape == (3, 4)
    ref.close()
    assert made[1].closed


# ---------------------------------------------------------------------------------------------------------------------
# `python bench.py --gpus N` typed without a launcher
# ---------------------------------------------------------------------------------------------------------------------
def test_the_ranks_command_is_the_drivers_own_multi_gpu_form():
    b = load_bench()
    assert b.torchrun_available()                                             # looked up on disk: the launching process imports no torch
    cmd = b.ranks_command(["--gpus", "4", "--steps", "20", "--n", "65536", "--n=4096", "--warmup", "5"], 4, 29517, script="/x/bench.py", python="py")
    assert cmd == ["py", "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=4", "--master-addr", "127.0.0.1", "--master-port", "29517",
                   "/x/bench.py", "--gpus", "4", "--steps", "20", "--nbodies", "65536", "--nbodies=4096", "--warmup", "5"]
    a = b.parse_args([])
    assert a.deadline <= 420.0 and a.candidate_deadline <= 45.0 and a.one_process is False     # inside the driver's 600 s limit


def test_launcher_relays_the_line_and_returns_the_childrens_status():
    import io
    b = load_bench()
    out = io.StringIO()
    rc = b.launch_ranks([], 2, out=out, command=[sys.executable, "-c", "import sys; print('[Gloo] a library note on stdout'); print('{\"value\": 1}'); sys.stderr.write('rank noise\\n')"])
    assert rc == 0 and out.getvalue() == '{"value": 1}\n'            # only the line reaches stdout; anything else a rank printed there goes to stderr
    out = io.StringIO()
    assert b.launch_ranks([], 2, out=out, command=[sys.executable, "-c", "import sys; sys.exit(4)"]) == 4 and out.getvalue() == ""
    out = io.StringIO()          # a child killed by a signal is a failure, never a silent 0
    assert b.launch_ranks([], 2, out=out, command=[sys.executable, "-c", "import os, signal; os.kill(os.getpid(), signal.SIGKILL)"]) == 128 + 9


def test_launcher_forwards_a_stop_and_keeps_relaying_until_the_children_are_gone(tmp_path):
    """SIGTERM at the launching process goes on to the children; what they print while stopping (rank 0: the safe-first line) is
    still relayed, and a run that delivered a line ends 0 even though the launcher child reports the signal as a failure."""
    import signal
    import subprocess
    import time
    child = tmp_path / "child.py"
    child.write_text("import signal, sys, time\n"
                     "def stop(s, f):\n"
                     "    print('{\"fallback\": {\"used\": true}}', flush=True)\n"
                     "    sys.exit(1)\n"
                     "signal.signal(signal.SIGTERM, stop)\n"
                     "sys.stderr.write('child ready\\n'); sys.stderr.flush()\n"
                     "time.sleep(60)\n")
    code = ("import importlib.util, sys\n"
            f"spec = importlib.util.spec_from_file_location('b', {str(ROOT / 'bench.py')!r}); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)\n"
            f"sys.exit(b.launch_ranks([], 2, command=[sys.executable, {str(child)!r}]))\n")
    p = subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    t0 = time.time()
    seen = ""
    while "child ready" not in seen and time.time() - t0 < 30:
        seen += p.stderr.readline()
    assert "child ready" in seen
    p.send_signal(signal.SIGTERM)
    out, err = p.communicate(timeout=30)
    assert p.returncode == 0, err
    assert out == '{"fallback": {"used": true}}\n'
    assert "torch" not in code and time.time() - t0 < 30


def test_typed_without_a_launcher_the_ranks_start_as_children_even_here():
    """No GPU in this container: the two ranks start (torch.distributed.run, one process each), each refuses loudly — the
    product has no CPU path — and the launcher returns their failure.  What matters here: `--gpus 2` no longer ends in
    'must be launched with torch.distributed.run'."""
    import subprocess
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--n", "4096"],
                       capture_output=True, text=True, timeout=300, cwd=str(ROOT))
    assert "starting 2 ranks as child processes" in r.stderr and "--nproc-per-node=2" in r.stderr and "--nbodies 4096" in r.stderr
    assert r.stderr.count("starting 2 ranks as child processes") == 2 and "one more attempt" in r.stderr      # ranks gone in seconds without a line: tried once more
    assert "must be launched with" not in r.stderr
    import torch
    if not torch.cuda.is_available():
        # (the launcher tears the other rank down as soon as the first one fails: one or two refusals reach stderr)
        assert r.returncode != 0 and r.stderr.count("bench.py needs a GPU") >= 1 and r.stdout.strip() == ""
        # ... and the last resort was tried too: ONE process driving all handles (which, here, finds no GPU either)
        assert "falling back to ONE process driving all 2 handles" in r.stderr and r.stderr.rstrip().endswith("the hot path has no CPU fallback")
