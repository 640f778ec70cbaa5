"""The driver's contract for bench.py, checked on the GPU with a short run: one JSON line on stdout, the keys and
types the contract names, `roofline` and `cpu_baseline` objects, the dominant kernel's figures consistent with each
other, and the skipped record when more GPUs are asked for than the box has."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args, timeout=600):
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + list(args), capture_output=True, text=True,
                       timeout=timeout, cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-500:]
    return json.loads(lines[0])


def test_bench_line_keeps_the_driver_contract():
    d = run_bench("--gpus", "1", "--steps", "8", "--warmup", "2", "--no-extra", "--no-cold", "--no-traffic")
    assert d["metric"].startswith("image-pairs/sec fwd+bwd @1024") and d["unit"] == "image-pairs/s"
    assert d["n_gpus"] == 1 and d["steps"] == 8 and d["warmup"] == 2
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 4 * 1e3 / d["ms_per_step"]) <= 0.01 * d["value"]          # value = pairs / time
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["achieved"] - r["algorithmic_bytes"] / (r["avg_us"] * 1e-6) / 1e9) <= 0.01 * r["achieved"]
    assert r["kernel"] in r["per_kernel"] and len(r["per_kernel"]) == 14
    assert 0.2 < r["frac"] < 1.0                                                        # a kernel, not a typo
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert d["steady_state"]["ms_per_step"] > 0
    # the timed K steps and the steady-state pass measure the same step: within 15 % of each other
    assert abs(d["ms_per_step"] - d["steady_state"]["ms_per_step"]) <= 0.15 * d["ms_per_step"]
    # what ran before the timed region is in the line, and the old-order figure of rounds 1-3 beside the headline
    assert d["pre_timed"]["probe_order"] == "before" and d["pre_timed"]["graph_replays_before_warmup"] == 80
    f = d["first_steps_after_idle"]
    assert f["steps"] == 8 and f["warmup"] == 2 and f["value"] > 0
    assert abs(f["value"] - 4 * 1e3 / f["ms_per_step"]) <= 0.01 * f["value"]


def test_bench_prints_a_skipped_record_when_the_box_has_fewer_gpus():
    import torch
    if torch.cuda.device_count() >= 8:
        pytest.skip("this box has 8 GPUs")
    d = run_bench("--gpus", "8", "--steps", "2", "--warmup", "1")
    assert "skipped" in d and d["n_gpus"] == 8
