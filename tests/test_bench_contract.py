"""bench.py prints ONE JSON line with the keys the driver reads (task contract)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline"]


def test_bench_keys_are_in_the_source():
    """cheap CPU-side guard: every contract key is emitted by the JSON literal in bench.py"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    body = src[src.index("out = {"):src.index("print(json.dumps(out))")]
    code = "\n".join(l.split("#")[0] if '"#' not in l else l for l in body.splitlines())      # drop comments
    for k in REQUIRED:
        assert '"%s"' % k in code, k


@pytest.mark.gpu
def test_bench_prints_the_contract_line():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "64"],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in REQUIRED:
        assert k in d, k
    assert d["metric"] == json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    assert d["unit"] == "frames/s" and d["value"] > 0 and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"], k
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 3          # ORB || LSD || planes, src/Frame.cc:210-215
    assert d["cpu_baseline_all_cores"]["cores"] == os.cpu_count()
    assert set(d["latency_ms"]) == {"B1", "B32"} and d["pcie_inclusive_frames_per_s"] > 0
    assert len(d["kernel_roofline"]) >= 11 and "lsd_pre" in d["kernel_roofline"] and all("frac" in v for v in d["kernel_roofline"].values())
    assert d["config"]["distinct_frames"] == 64 and d["config"]["scene_mix"] == {"lowtex": 16, "std": 48}


@pytest.mark.gpu
def test_bench_stream_mode_line():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "stream", "--steps", "24", "--warmup", "2", "--no-extras"],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["unit"] == "frames/s" and d["value"] > 0 and d["steps"] == 24 and d["config"]["mode"] == "stream"
    assert d["latency_ms"]["pipelined_p50"] > 0 and d["cpu_baseline"]["cores"] == 3
    assert d["config"]["mean_point_matches"] > 100


def test_gpus_flag_must_match_world_size():
    """--gpus N inside a torchrun environment of another size is an error (before anything touches the GPU)"""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True, timeout=120, env=env)
    assert p.returncode == 2 and "WORLD_SIZE" in p.stderr
