"""`-m perf`: orderings between MEASURED durations.  Not part of the correctness suites: `-m gpu -x` must only go red on
defects, and which of two kernels is faster on a given box in a 4-step run depends on the clock ramp.
Run on a GPU box with `python -m pytest tests -m perf`; skipped where there is no GPU."""
import pytest

from test_bench_contract_gpu import run_bench

pytestmark = pytest.mark.perf


def _gpu():
    import torch
    return torch.cuda.is_available()


@pytest.mark.skipif(not _gpu(), reason="orderings of measured GPU durations need a GPU")
def test_orderings_of_the_bench_line_at_a_size_where_they_are_stable():
    d = run_bench("--n", "131072", "--steps", "20", "--warmup", "5")
    rf = d["roofline"]
    g, sc, t = rf["general_mass"], rf["general_mass_scaled"], rf["one_sided_lds_tiled"]
    assert g["avg_launch_ms"] > rf["avg_launch_ms"] and g["frac"] < rf["frac"]             # 12 + 2 vs 10 + 2 instructions per body
    assert rf["avg_launch_ms"] < sc["avg_launch_ms"] < g["avg_launch_ms"] * 1.02         # 11 + 2 sits between them
    assert t["avg_launch_ms"] > rf["avg_launch_ms"] and 0.3 < t["frac"] < rf["frac"]       # every ordered pair vs every unordered pair
    cb = d["cpu_baseline"]
    if cb["kind"] == "reference":
        assert cb["port"]["value"] > cb["value"]                                             # SoA + vectorised port vs the reference's AoS loop
