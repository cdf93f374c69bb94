"""CPU checks of the measurement contract: the committed bench line carries every field the driver reads, and
bench.py refuses to run without a GPU instead of timing anything on the CPU (the product has no CPU fallback)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def test_committed_bench_line_has_the_contract_fields():
    with open(os.path.join(ROOT, "profiles", "r01_lattice_bench.json")) as f:
        line = json.load(f)
    for key in CONTRACT:
        assert key in line, key
    assert line["n_gpus"] == 1 and line["higher_is_better"] is True and line["vs_baseline"] is None
    assert line["dtype"] == "f64" and line["data"] == "synthetic" and "workload" in line["config"]
    assert abs(line["value"] - 100000 * line["steps"] / (line["ms_per_step"] * 1e-3 * line["steps"])) < 1e-6 * line["value"]
    roof = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof, key
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12
    assert abs(roof["achieved"] - roof["algorithmic_bytes_per_launch"] / (roof["avg_launch_us"] * 1e-6) / 1e9) < 1e-6 * roof["achieved"]
    cpu = line["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cpu, key
    assert cpu["kind"] == "port" and cpu["cores"] == 1 and cpu["unit"] == line["unit"]


def test_bench_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is visible")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode != 0
    assert r.stdout.strip() == ""                            # no JSON line, nothing measured
    assert "GPU" in r.stderr or "device" in r.stderr.lower()
