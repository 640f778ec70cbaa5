"""bench.py pieces that run without a GPU: the launch order the in-step trace is decoded with, and
the CPU baseline's report form."""
import importlib.util
import os

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench_module", os.path.join(REPO, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def test_step_order_is_the_order_direction_issues_its_launches():
    order = bench.step_order(4)
    assert order == ["corr_fwd_L0", "warp_fwd_L1", "corr_fwd_L1", "warp_fwd_L2", "corr_fwd_L2", "warp_fwd_L3",
                     "corr_fwd_L3", "corr_bwd_L3", "warp_bwd_L3", "corr_bwd_L2", "warp_bwd_L2", "corr_bwd_L1",
                     "warp_bwd_L1", "corr_bwd_L0"]
    assert len(set(order)) == 14


def test_algorithmic_bytes_match_the_survey_formulas():
    # L3 at B = 1 fp32: 19.005 / 27.394 MB
    f, b = bench.corr_bytes(32, 1, 128, 256)
    assert (f, b) == (19005440, 27394048)
    wf, wb = bench.warp_bytes(32, 1, 128, 256)
    assert wf + wb == (2 * 32 + 2 + 3 * 32 + 4) * 128 * 256 * 4


def test_cpu_baseline_reports_min_and_median_per_shape():
    r = bench.cpu_baseline([(8, 8, 16), (4, 16, 32)], budget_s=5.0)
    assert r["kind"] == "port" and r["unit"] == "image-pairs/s" and r["value"] > 0
    assert r["cores"] >= 1 and r["host_cores"] == os.cpu_count()
    assert set(r["per_shape_ms"]) == {"config1_1x64x64x128", "L0", "L1"}
    for row in r["per_shape_ms"].values():
        for what in ("fwd", "fwd_bwd"):
            assert 0 < row[what + "_ms_min"] <= row[what + "_ms_median"]
            assert row[what + "_iters"] >= 3
    pair_s = 2e-3 * sum(r["per_shape_ms"]["L%d" % l]["fwd_bwd_ms_median"] for l in range(2))
    assert np.isclose(r["value"], 1.0 / pair_s, rtol=1e-3)


def test_in_step_pass_refuses_to_nest_profilers():
    """under an outer rocprofv3 the bench must not start its own profiler child, and the
    child's environment never inherits tool / preload variables."""
    assert not bench.profiler_in_environment({"PATH": "/usr/bin", "LD_PRELOAD": ""})
    assert bench.profiler_in_environment({"ROCP_TOOL_LIBRARIES": "/opt/rocm/lib/librocprofiler-sdk-tool.so"})
    assert bench.profiler_in_environment({"LD_PRELOAD": "/opt/rocm/lib/librocprofiler-sdk-tool.so.0"})
    assert bench.profiler_in_environment({"ROCPROF_OUTPUT_PATH": "/tmp/x"})
    for name in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROF_OUTPUT_PATH", "HSA_TOOLS_LIB", "ROCPROFILER_LOG_LEVEL"):
        assert bench._is_profiler_variable(name), name
    assert not bench._is_profiler_variable("HSA_ENABLE_IPC_MODE_LEGACY")
    os.environ["ROCP_TOOL_LIBRARIES"] = "x.so"
    try:
        class A:
            pairs, width, height, dtype, flow, no_mfma = 4, 1024, 512, "f32", "smooth", False
        times, why = bench.in_step_times(A(), 4)
        assert times is None and "profiler" in why
    finally:
        del os.environ["ROCP_TOOL_LIBRARIES"]


def test_live_traffic_refuses_to_nest_profilers_too():
    os.environ["ROCP_TOOL_LIBRARIES"] = "x.so"
    try:
        class A:
            pairs, width, height, dtype, flow, no_mfma = 4, 1024, 512, "f32", "smooth", False
        res, why = bench.live_traffic(A(), 4)
        assert res is None and "profiler" in why
    finally:
        del os.environ["ROCP_TOOL_LIBRARIES"]


def _run_bench(extra_env, *flags, timeout=420):
    import subprocess
    import sys
    env = dict(os.environ, **extra_env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--device", "cpu", "--arch", "w18", "--pairs", "1",
                           "--width", "128", "--height", "64", "--steps", "2", "--warmup", "1"] + list(flags),
                          env=env, capture_output=True, text=True, timeout=timeout)


def test_the_n_gt_1_route_of_bench_runs_end_to_end_under_gloo_world_2():
    """`bench.py --gpus N` at N > 1 had never executed anywhere.  On this CPU box the REAL route runs with
    gloo in place of RCCL: spawn_ranks -> torch.distributed.run (two children) -> rendezvous on 127.0.0.1 -> wrap_ddp
    around the host model (W18, stock-PyTorch fallback ops: the product has no CPU path) -> warm-up + timed DDP steps
    (barrier on both sides, max over ranks) -> the no_sync single-GPU pass -> rank 0 prints ONE JSON line whose top
    level carries single_gpu_step and scaling_efficiency."""
    import json
    r = _run_bench({}, "--gpus", "2")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "single_gpu_step", "scaling_efficiency"):
        assert key in d, key
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["step"] == "model" and d["config"]["collective_backend"] == "gloo" and d["config"]["device"] == "cpu"
    assert "DistributedDataParallel" in d["config"]["sharding"] and "value_times" in d["config"]
    one = d["single_gpu_step"]["pairs_per_s_per_gpu"]
    assert one > 0 and np.isclose(d["scaling_efficiency"], d["value"] / (2 * one), rtol=2e-3)
    assert np.isclose(d["value"], 2 * 1 * 2 / (d["ms_per_step"] * 2e-3), rtol=2e-2)      # pairs x ranks x steps / time


def test_a_rank_that_raises_takes_the_job_down_instead_of_hanging_its_peers():
    """rank 1 raises inside the second timed step while rank 0 is in DDP's gradient all-reduce; the launcher
    must come back with a non-zero exit code (well inside the timeout), with no JSON line."""
    r = _run_bench({"CERB_BENCH_FAIL_AT": "1:1"}, "--gpus", "2", timeout=240)
    assert r.returncode != 0
    assert "injected failure on rank 1" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_cpu_dry_run_refuses_the_op_only_step():
    r = _run_bench({}, "--gpus", "1", "--step", "ops", timeout=120)
    assert r.returncode != 0 and "no CPU implementation" in (r.stderr + r.stdout)


def test_cpu_baseline_also_reports_the_all_cores_figure_as_specified():
    r = bench.cpu_baseline([(8, 8, 16), (4, 16, 32)], budget_s=5.0)
    assert r["value_all_cores"] > 0 and r["cores_all"] == os.cpu_count()
    assert set(r["all_cores"]["per_shape_ms"]) == {"L0", "L1"}
    pair_s = 2e-3 * sum(r["all_cores"]["per_shape_ms"]["L%d" % l]["fwd_bwd_ms_median"] for l in range(2))
    assert np.isclose(r["value_all_cores"], 1.0 / pair_s, rtol=1e-3)
