"""bench.py's helpers and the shape of the bench line it emits (checked on the committed line of the last GPU run)."""
import importlib.util
import json
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
spec = importlib.util.spec_from_file_location("bench", ROOT / "bench.py")
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def test_algorithmic_bytes_formula():
    # SURVEY.md §8(d) with p=1, n=2, c=126, E=10, r=3 at L=150, K=24
    ctr = dict(reads=1, probes=1, node_visits=2, bases_compared=126, class_sizes=10, result_sizes=3)
    want = 38 + 4 + 1 * (8 + 6 + 12) + 2 * 21 + 2 * 126 / 8 + 8 * 2 + 4 * 10 + 12 + 4 * 3
    assert abs(bench.algorithmic_bytes_per_read(ctr, 150, 24) - want) < 1e-9


def test_usable_cpus_is_sane():
    import os
    n = bench.usable_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)


import pytest


@pytest.mark.parametrize("name", ["r01_pool_bench.json", "r02_bench_config3.json", "r02_bench_config5.json", "r03_bench_config3.json", "r03_bench_config5.json"])
def test_committed_bench_line_has_the_contract_fields(name):
    line = json.loads((ROOT / "profiles" / name).read_text())
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["vs_baseline"] is None and line["scaling"] == "weak" and line["higher_is_better"] is True and "workload" in line["config"]
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["peak"] == 8000.0
    assert abs(r["achieved"] - r["algorithmic_bytes_per_read"] * r["reads_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert r["traffic"] is None or r["traffic"] > r["algorithmic_bytes_per_read"] * r["reads_per_launch"]
    c = line["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["unit"] == line["unit"] and "sample" in c
    assert abs(line["value"] - line["config"]["reads_per_step_per_gpu"] * line["n_gpus"] / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]
    assert r["kernel_ms"] <= line["ms_per_step"] * 1.001   # the dominant kernel fits inside the step it belongs to
    if name.startswith("r03"):   # round 3: the map kernel is timed by the library, the step holds more kernels; config 5 rides in the default line
        assert r["kernel"] == "pa_map_pool_kernel" and r["kernel_ms"] <= r["step_device_ms"] <= line["ms_per_step"] * 1.001 and "rccl_ranks" in line
        assert c["cpus_visible"] >= c["cores"]
        if "config3" in name:
            c5 = line["config5"]
            assert c5["value"] > 0 and c5["parity_sample"]["bit_exact_vs_oracle"] is True and 0 < c5["roofline"]["frac"] < 1
            assert r["traffic"] is not None and line["e2e_reads_per_s"] > 0
    if name.startswith("r02"):   # round 2: the host-to-host leg (config 3 only) and the index set-up times ride along as extra keys
        assert "index_build_s" in line["config"] and "kernel_ms_steps" in r and len(r["kernel_ms_steps"]) == line["steps"]
        if "config3" in name:
            assert line["e2e_reads_per_s"] > 0 and 0 < line["e2e_pcie_frac"] < 1
