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
    with open(os.path.join(ROOT, "profiles", "r02_bench_config2.json")) as f:
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
    # round-2 harness: the measured limiter next to the contract's bound, this box's copy-kernel ceiling, per-workload
    # traffic (a number recorded for this workload or null), median of >= 5 repeats, how the timed steps were launched
    assert roof["bound_measured"] and 1000.0 < roof["hbm_copy_GBps"] < 8000.0 and roof["kernels_per_step"] <= 2
    assert roof["traffic"] is None or roof["traffic"] > 0
    assert line["repeats"] >= 5 and len(line["repeats_ms_per_step"]) == line["repeats"]
    assert abs(line["ms_per_step"] - sorted(line["repeats_ms_per_step"])[len(line["repeats_ms_per_step"]) // 2]) < 1e-12
    assert isinstance(line["graph_replay"], bool) and line["steps_from_graphs"] + line["steps_launched_singly"] == line["steps"]
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12
    assert abs(roof["achieved"] - roof["algorithmic_bytes_per_launch"] / (roof["avg_launch_us"] * 1e-6) / 1e9) < 1e-6 * roof["achieved"]
    cpu = line["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cpu, key
    assert cpu["kind"] == "port" and cpu["cores"] == 1 and cpu["unit"] == line["unit"]


def test_committed_resident_loop_bench_line():
    """The default bench line since the resident loop exists (csrc/tile_loop.hpp): the timed steps run inside one launch; the
    roofline object prices that launch (per-step algorithmic bytes x its steps over its duration, timed by events on the
    dispatch), traffic is the recorded per-step counter figure scaled to the launch."""
    for name, steps in (("r02_loop_bench_config2.json", 2000), ("r02_loop_bench_config2_driver_form.json", 20)):
        with open(os.path.join(ROOT, "profiles", name)) as f:
            line = json.load(f)
        for key in CONTRACT[:-1]:
            assert key in line, key
        assert line["steps"] == steps and line["n_gpus"] == 1 and line["dtype"] == "f64"
        assert line["resident_loop"] == "used" and line["steps_in_resident_loop"] in (steps, steps - 1)
        assert line["steps_from_graphs"] + line["steps_launched_singly"] + line["steps_in_resident_loop"] == steps
        assert abs(line["value"] - 100000 / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]
        roof = line["roofline"]
        assert roof["kernel"] == "tile_loop" and roof["steps_per_launch"] == line["steps_in_resident_loop"]
        assert abs(roof["achieved"] - roof["algorithmic_bytes_per_launch"] / (roof["avg_launch_us"] * 1e-6) / 1e9) < 1e-6 * roof["achieved"]
        assert abs(roof["frac"] - roof["achieved"] / 8000.0) < 1e-12
        per_step = roof["per_kernel"]["tile_step"]["algorithmic_bytes_per_launch"]
        assert abs(roof["algorithmic_bytes_per_launch"] - per_step * roof["steps_per_launch"]) < 1e-6 * roof["algorithmic_bytes_per_launch"]
        # the launch cannot be shorter than the steps it takes inside the timed region allow
        assert roof["avg_launch_us"] <= line["ms_per_step"] * 1e3 * steps


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


def test_gpus_flag_spawns_one_rank_per_gpu_without_touching_the_gpu_in_the_parent():
    """`python bench.py --gpus N` (no launcher): the parent starts N rank processes with the torchrun environment and relays
    rank 0's line; it never imports the C-ABI binding (a process that initialised the GPU must not be re-exec'd, and the
    parent needs no GPU at all)."""
    import io
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    before = set(sys.modules)
    spec.loader.exec_module(bench)
    made = []

    class FakeProc:
        def __init__(self, cmd, env=None, stdout=None, text=None):
            self.cmd, self.env, self.returncode = cmd, env, 0
            self.stdout = io.StringIO('{"n_gpus": 3}\n') if env["RANK"] == "0" else None
            made.append(self)

        def poll(self):
            return 0

        def wait(self, timeout=None):
            return 0

    rc = bench.spawn_ranks(3, ["--gpus", "3", "--steps", "4"], popen=FakeProc, port=23456)
    assert rc == 0 and len(made) == 3
    for r, p in enumerate(made):
        assert p.env["RANK"] == p.env["LOCAL_RANK"] == str(r) and p.env["WORLD_SIZE"] == "3"
        assert p.env["MASTER_ADDR"] == "127.0.0.1" and p.env["MASTER_PORT"] == "23456"
        assert p.cmd[0] == sys.executable and p.cmd[1].endswith("bench.py") and p.cmd[2:] == ["--gpus", "3", "--steps", "4"]
    assert not any("capi" in m for m in set(sys.modules) - before), "the launcher parent must not load the GPU binding"
    # a launcher that disagrees with --gpus is an error, not a silent single-rank run
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=dict(os.environ, WORLD_SIZE="2", RANK="0"),
                       capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert r.returncode != 0 and "disagrees" in r.stderr
