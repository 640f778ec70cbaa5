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
