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
