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
