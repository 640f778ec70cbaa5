"""The torch-free C++ inference host (cerberusnet_amd/runtime: SURVEY.md 8(f)-4, the MI355X counterpart of
the reference's TensorRT plugins runtime/cerberus_net/trt_plugins/correlation.cu:94-166 and
grid_sampler.cu:238-271).  CPU: it builds, links the C ABI and fails loudly without a device.  GPU: raw
buffers in, raw buffers out, against the goldens captured from the reference's Python ops and the oracle."""
import json
import os
import subprocess

import numpy as np
import pytest

from cerberusnet_amd import _lib
from cerberusnet_amd.build import RUNTIME_BIN, build_runtime
from cerberusnet_amd.synth import hash_uniform
from conftest import rel_err


def run(args, **kw):
    return subprocess.run([RUNTIME_BIN] + args, capture_output=True, text=True, timeout=300, **kw)


def test_runtime_builds_links_the_c_abi_and_has_no_cpu_path(tmp_path):
    build_runtime()
    assert os.path.exists(RUNTIME_BIN)
    assert int(run(["--abi"]).stdout) == _lib.ABI_VERSION              # linked against THIS library
    ldd = subprocess.run(["ldd", RUNTIME_BIN], capture_output=True, text=True).stdout
    assert "libcerberus_hip.so" in ldd and "libtorch" not in ldd and "libc10" not in ldd   # torch-free
    import torch
    if not torch.cuda.is_available():
        r = run(["--dir", str(tmp_path), "--levels", "8,4,4"])
        assert r.returncode == 1 and "no HIP device" in r.stderr
    assert run(["--levels", "x"]).returncode != 0
    assert run(["--dtype"]).returncode == 2


@pytest.mark.gpu
@pytest.mark.parametrize("graph", [True, False])
def test_runtime_pyramid_from_raw_buffers_against_goldens_and_oracle(tmp_path, golden, graph):
    """Two levels: level 0 = the reference golden corr_b (1,32,16,24) (CorrelationTorch output, d = 4), level 1
    = warp -> correlation on hash inputs against the oracle chain (flow_warp_ref -> corr_forward_ref), both
    with the LeakyReLU(0.1) of pwcnet_sfd.py:182 fused into the store."""
    import torch
    import oracle
    g = golden("corr_b")
    shapes = [(32, 16, 24), (16, 20, 36)]
    g["x1"].astype(np.float32).tofile(tmp_path / "f1_0.bin")
    g["x2"].astype(np.float32).tofile(tmp_path / "f2_0.bin")
    f1 = hash_uniform((1, 16, 20, 36), 501)
    f2 = hash_uniform((1, 16, 20, 36), 502)
    fl = hash_uniform((1, 2, 20, 36), 503, -5.0, 5.0)
    f1.tofile(tmp_path / "f1_1.bin"); f2.tofile(tmp_path / "f2_1.bin"); fl.tofile(tmp_path / "flow_1.bin")
    r = run(["--dir", str(tmp_path), "--levels", ";".join("%d,%d,%d" % s for s in shapes), "--reps", "5"] +
            ([] if graph else ["--no-graph"]))
    assert r.returncode == 0, r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["launch"] == ("hipGraph" if graph else "eager") and line["us_per_frame_pair_direction"] > 0
    leaky = lambda a: np.where(a > 0, a, 0.1 * a)
    cost0 = np.fromfile(tmp_path / "cost_0.bin", np.float32).reshape(1, 81, 16, 24)
    assert rel_err(cost0, leaky(g["out64"])) < 1e-5
    warped = np.fromfile(tmp_path / "warped_1.bin", np.float32).reshape(1, 16, 20, 36)
    wref = oracle.flow_warp_ref(torch.from_numpy(f2), torch.from_numpy(fl)).numpy()
    assert rel_err(warped, wref) < 1e-5
    cost1 = np.fromfile(tmp_path / "cost_1.bin", np.float32).reshape(1, 81, 20, 36)
    assert rel_err(cost1, leaky(oracle.corr_forward_ref(f1, wref, 4, 1, 4, 1, 1))) < 1e-5


@pytest.mark.gpu
def test_runtime_fp16_and_batched_pyramid(tmp_path):
    import torch
    import oracle
    B, C, H, W = 2, 32, 24, 40
    f1 = hash_uniform((B, C, H, W), 511).astype(np.float16)
    f2 = hash_uniform((B, C, H, W), 512).astype(np.float16)
    f1.tofile(tmp_path / "f1_0.bin"); f2.tofile(tmp_path / "f2_0.bin")
    r = run(["--dir", str(tmp_path), "--levels", "%d,%d,%d" % (C, H, W), "--batch", str(B), "--dtype", "f16",
             "--slope", "1.0"])
    assert r.returncode == 0, r.stderr
    cost = np.fromfile(tmp_path / "cost_0.bin", np.float16).reshape(B, 81, H, W).astype(np.float64)
    ref = oracle.corr_forward_ref(f1.astype(np.float64), f2.astype(np.float64), 4, 1, 4, 1, 1)
    assert rel_err(cost, ref) < 2e-3
