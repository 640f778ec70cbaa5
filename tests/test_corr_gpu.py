"""GPU parity tests for the correlation HIP path (through torch.ops -> C ABI).

Tolerance: fp32 outputs must agree with the oracle to <= 1e-5 relative, measured
as max|a-ref| / max|ref| (BASELINE.json north_star; element-wise relative error
is ill-conditioned for near-zero correlation values)."""
import numpy as np
import pytest
import torch

import cerberusnet_amd as ca
from cerberusnet_amd import _lib
from cerberusnet_amd.synth import hash_uniform, W32_PYRAMID_1024x512
from conftest import rel_err
import oracle

pytestmark = pytest.mark.gpu
TOL = 1e-5
DEV = "cuda:0"


def experiments_build():
    """True when libcerberus_hip.so was built with -DCERB_EXPERIMENTS: the measured-and-rejected
    kernel variants (forward 1, 2, 8; backward 2, 6, 7, 9, 10) exist only in such test builds."""
    import ctypes
    v = ctypes.c_int(0)
    _lib.check(_lib.get().cerberus_get_option(b"experiments_build", ctypes.byref(v)), "get_option")
    return bool(v.value)


FWD_EXPERIMENTS, BWD_EXPERIMENTS = {1, 2, 8}, {2, 6, 7, 9, 10}
try:        # decided at collection time (no GPU needed: the library is only loaded, nothing is launched)
    EXPERIMENTS = experiments_build()
except Exception:   # library not built: the GPU tests fail loudly on their own
    EXPERIMENTS = False


def variants(all_of_them, experimental):
    """The kernel variants this build holds: the measured-and-rejected ones exist in the -DCERB_EXPERIMENTS library only
    (lib/libcerberus_hip_experiments.so); tests/test_experiments_gpu.py runs this file against it in a child process, so
    the product run collects no test it would have to skip."""
    return [v for v in all_of_them if EXPERIMENTS or v not in experimental]


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return t if dtype is None else t.to(dtype)


def run_fwd(x1, x2, p):
    return torch.ops.cerberus.correlation(dev(x1), dev(x2), *p, 1).cpu().numpy()


def run_bwd(x1, x2, go, p):
    g1, g2 = torch.ops.cerberus.correlation_backward(dev(x1), dev(x2), dev(go), *p, 1)
    return g1.cpu().numpy(), g2.cpu().numpy()


@pytest.fixture(params=[0, 1], ids=["tuned", "generic"])
def force_generic(request):
    _lib.set_option("corr_force_generic", request.param)
    yield request.param
    _lib.set_option("corr_force_generic", 0)


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_golden_vectors(golden, tag, force_generic):
    g = golden("corr_" + tag)
    d = int(g["d"])
    p = (d, 1, d, 1, 1)
    assert rel_err(run_fwd(g["x1"], g["x2"], p), g["out64"]) < TOL
    g1, g2 = run_bwd(g["x1"], g["x2"], g["gout"], p)
    assert rel_err(g1, g["g1_64"]) < TOL
    assert rel_err(g2, g["g2_64"]) < TOL


# ragged / edge shapes at the configuration every model uses (pad=d=4,k=1,s=1)
D4_SHAPES = [(1, 1, 1, 1), (1, 3, 5, 7), (2, 7, 9, 33), (1, 32, 16, 24), (3, 16, 13, 64),
             (1, 64, 20, 36), (2, 33, 8, 130), (1, 5, 70, 9), (1, 256, 16, 32),
             (2, 128, 32, 64), (1, 48, 30, 62), (1, 32, 17, 258)]


@pytest.mark.parametrize("shape", D4_SHAPES)
def test_d4_against_c_oracle(shape, force_generic):
    B, C, H, W = shape
    x1 = hash_uniform(shape, 100 + C)
    x2 = hash_uniform(shape, 200 + H)
    go = hash_uniform((B, 81, H, W), 300 + W)
    p = (4, 1, 4, 1, 1)
    assert rel_err(run_fwd(x1, x2, p), oracle.corr_forward_ref(x1, x2, *p)) < TOL
    r1, r2 = oracle.corr_backward_ref(x1, x2, go, *p)
    g1, g2 = run_bwd(x1, x2, go, p)
    assert rel_err(g1, r1) < TOL
    assert rel_err(g2, r2) < TOL


GENERAL = [(4, 1, 4, 1, 1), (3, 3, 4, 1, 2), (5, 3, 4, 1, 1), (2, 1, 4, 1, 1), (4, 1, 10, 1, 1),
           (6, 1, 6, 1, 3), (3, 3, 20, 1, 2), (0, 1, 0, 1, 1), (2, 1, 2, 1, 1), (1, 1, 1, 1, 1),
           (4, 3, 3, 1, 1)]


@pytest.mark.parametrize("p", GENERAL)
def test_general_parameters_against_c_oracle(p):
    pad, k, d, s1, s2 = p
    need = 2 * ((k - 1) // 2 + d) - 2 * pad
    B, C, H, W = 2, 6, max(9, need + 6), max(11, need + 9)
    x1 = hash_uniform((B, C, H, W), 1)
    x2 = hash_uniform((B, C, H, W), 2)
    ref = oracle.corr_forward_ref(x1, x2, *p)
    assert rel_err(run_fwd(x1, x2, p), ref) < TOL
    go = hash_uniform(ref.shape, 3)
    r1, r2 = oracle.corr_backward_ref(x1, x2, go, *p)
    g1, g2 = run_bwd(x1, x2, go, p)
    assert rel_err(g1, r1) < TOL
    assert rel_err(g2, r2) < TOL


def test_reference_main_benchmark_point_at_its_own_size():
    """The reference's only in-tree exercise of the op besides the models: ``correlation.py:82-113``
    (``__main__``): pad 4 < d 10, k 1, strides 1 (output shrinks: Q5) on (16, 128|256, 128|256,
    64|128) tensors.  One batch item of its smallest draw, forward and both gradients, against
    the C restatement of the CUDA kernels (441 displacements x 128 channels)."""
    p = (4, 1, 10, 1, 1)
    shape = (1, 128, 128, 64)
    x1, x2 = hash_uniform(shape, 21), hash_uniform(shape, 22)
    ref = oracle.corr_forward_ref(x1, x2, *p)
    assert ref.shape == (1, 441, 116, 52)
    out = run_fwd(x1, x2, p)
    assert _lib.last_kernel(0) == "corr_fwd_generic"
    assert rel_err(out, ref) < TOL
    go = hash_uniform(ref.shape, 23)
    r1, r2 = oracle.corr_backward_ref(x1, x2, go, *p)
    g1, g2 = run_bwd(x1, x2, go, p)
    assert rel_err(g1, r1) < TOL
    assert rel_err(g2, r2) < TOL


def test_forward_stride1_gt_1_and_backward_rejects_it():
    x1 = hash_uniform((1, 4, 10, 12), 5)
    x2 = hash_uniform((1, 4, 10, 12), 6)
    for p in [(4, 1, 4, 2, 1), (4, 1, 4, 3, 2), (3, 3, 4, 2, 2)]:
        ref = oracle.corr_forward_ref(x1, x2, *p)
        out = run_fwd(x1, x2, p)
        assert out.shape == ref.shape and rel_err(out, ref) < TOL
    with pytest.raises(RuntimeError, match="stride1"):
        torch.ops.cerberus.correlation_backward(dev(x1), dev(x2), dev(np.zeros((1, 81, 5, 6),
                                                np.float32)), 4, 1, 4, 2, 1, 1)


@pytest.mark.parametrize("lvl", [0, 1, 2, 3])
def test_fullsize_levels_against_reference_checksums(golden, lvl, force_generic):
    """BASELINE config-3 level shapes (B=1), inputs from the portable generator,
    compared with the checksums/samples the REFERENCE produced in-container."""
    g = golden("fullsize")
    C, H, W = W32_PYRAMID_1024x512[lvl]
    shp = (1, C, H, W)
    x1, x2, go = hash_uniform(shp, 0), hash_uniform(shp, 1), hash_uniform((1, 81, H, W), 2)
    p = (4, 1, 4, 1, 1)
    out = run_fwd(x1, x2, p)
    g1, g2 = run_bwd(x1, x2, go, p)
    name = "L%d" % lvl
    for key, arr in (("out", out), ("g1", g1), ("g2", g2)):
        amax = float(g["%s_%s_absmax" % (name, key)])
        got = arr.reshape(-1)[g["%s_%s_idx" % (name, key)]]
        assert np.abs(got - g["%s_%s_val" % (name, key)]).max() <= TOL * amax
        a64 = arr.astype(np.float64)
        assert abs(a64.sum() - g["%s_%s_sum" % (name, key)]) <= 1e-6 * amax * np.sqrt(a64.size)
        assert abs((a64 * a64).sum() - g["%s_%s_sumsq" % (name, key)]) <= \
            1e-5 * g["%s_%s_sumsq" % (name, key)]


def test_config1_forward_against_reference_checksums(golden):
    g = golden("fullsize")
    shp = (1, 64, 64, 128)
    out = run_fwd(hash_uniform(shp, 0), hash_uniform(shp, 1), (4, 1, 4, 1, 1))
    amax = float(g["cfg1_out_absmax"])
    assert np.abs(out.reshape(-1)[g["cfg1_out_idx"]] - g["cfg1_out_val"]).max() <= TOL * amax


@pytest.mark.parametrize("lvl", [0, 1, 2, 3])
def test_fullsize_batch4_properties(lvl):
    """B=4 (config 4 per-GPU batch): size-independent properties.
    (1) batch items are independent: item n of the B=4 call is equal (to fp32 rounding) to
        the B=1 call on item n; (2) bilinearity: corr(a*x1, x2) == a*corr(x1, x2)
        exactly for a power of two; (3) adjoint identity ties backward to forward:
        <corr(x1,x2), gO> == <x1, g1> == <x2, g2>."""
    C, H, W = W32_PYRAMID_1024x512[lvl]
    B = 4
    x1 = dev(hash_uniform((B, C, H, W), 7))
    x2 = dev(hash_uniform((B, C, H, W), 8))
    go = dev(hash_uniform((B, 81, H, W), 9))
    p = (4, 1, 4, 1, 1, 1)
    out = torch.ops.cerberus.correlation(x1, x2, *p)
    g1, g2 = torch.ops.cerberus.correlation_backward(x1, x2, go, *p)
    for n in (0, 3):
        # the B=1 call may dispatch to a different tile / channel-split variant
        # (different summation order), so compare to rounding, not bit-for-bit
        o1 = torch.ops.cerberus.correlation(x1[n:n + 1], x2[n:n + 1], *p)[0]
        assert rel_err(o1.cpu().numpy(), out[n].cpu().numpy()) < 1e-6
        h1, h2 = torch.ops.cerberus.correlation_backward(x1[n:n + 1], x2[n:n + 1],
                                                         go[n:n + 1], *p)
        assert rel_err(h1[0].cpu().numpy(), g1[n].cpu().numpy()) < 1e-6
        assert rel_err(h2[0].cpu().numpy(), g2[n].cpu().numpy()) < 1e-6
    assert torch.equal(torch.ops.cerberus.correlation(x1 * 4.0, x2, *p), out * 4.0)
    lhs = (out.double() * go.double()).sum().item()
    # fp32 results summed over ~1e7 terms: scale the tolerance by the sum of magnitudes
    scale = (out.double() * go.double()).abs().sum().item()
    assert abs(lhs - (x1.double() * g1.double()).sum().item()) <= 1e-6 * scale
    assert abs(lhs - (x2.double() * g2.double()).sum().item()) <= 1e-6 * scale
    # swapping the roles of the two maps mirrors the displacement axis
    swapped = torch.ops.cerberus.correlation(x2, x1, *p)
    chk = out[:, 40]  # zero displacement channel is symmetric
    assert torch.allclose(swapped[:, 40], chk, rtol=0, atol=1e-6)


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-12), (torch.float16, 2e-3),
                                       (torch.bfloat16, 1.6e-2)])
def test_other_dtypes(dtype, tol, force_generic):
    """fp64 bit-for-bit class accuracy; fp16/bf16 are storage formats with fp32
    accumulation (Q6): compare with the fp64 oracle run on the rounded inputs;
    the tolerance is the output rounding of the storage type."""
    shp = (2, 24, 10, 20)
    x1 = torch.from_numpy(hash_uniform(shp, 1)).to(dtype)
    x2 = torch.from_numpy(hash_uniform(shp, 2)).to(dtype)
    go = torch.from_numpy(hash_uniform((2, 81, 10, 20), 3)).to(dtype)
    p = (4, 1, 4, 1, 1)
    ref = oracle.corr_forward_ref(x1.double().numpy(), x2.double().numpy(), *p)
    r1, r2 = oracle.corr_backward_ref(x1.double().numpy(), x2.double().numpy(),
                                      go.double().numpy(), *p)
    out = torch.ops.cerberus.correlation(x1.to(DEV), x2.to(DEV), *p, 1)
    assert out.dtype == dtype
    g1, g2 = torch.ops.cerberus.correlation_backward(x1.to(DEV), x2.to(DEV), go.to(DEV), *p, 1)
    assert rel_err(out.double().cpu().numpy(), ref) < tol
    assert rel_err(g1.double().cpu().numpy(), r1) < tol
    assert rel_err(g2.double().cpu().numpy(), r2) < tol


def test_module_train_and_eval_paths_and_inplace_leaky():
    """Correlation nn.Module: training -> CorrelationFunction (autograd), eval ->
    raw op (correlation.py:72-80); the caller's in-place leaky_relu on the output
    (pwcnet_sfd.py:182) must stay legal; gradients equal torch autograd through
    the reference-semantics CorrelationTorch."""
    shp = (2, 16, 12, 20)
    a = dev(hash_uniform(shp, 11)).requires_grad_(True)
    b = dev(hash_uniform(shp, 12)).requires_grad_(True)
    corr = ca.Correlation(pad_size=4, kernel_size=1, max_displacement=4, stride1=1, stride2=1,
                          corr_multiply=1).to(DEV)
    corr.train()
    out = corr(a, b)
    torch.nn.functional.leaky_relu(out, 0.1, inplace=True)
    out.square().sum().backward()
    a2 = a.detach().clone().requires_grad_(True)
    b2 = b.detach().clone().requires_grad_(True)
    ref = torch.nn.functional.leaky_relu(ca.CorrelationTorch(4)(a2, b2), 0.1)
    ref.square().sum().backward()
    assert rel_err(out.detach().cpu().numpy(), ref.detach().cpu().numpy()) < TOL
    assert rel_err(a.grad.cpu().numpy(), a2.grad.cpu().numpy()) < TOL
    assert rel_err(b.grad.cpu().numpy(), b2.grad.cpu().numpy()) < TOL
    corr.eval()
    with torch.no_grad():
        ev = corr(a, b)
    assert rel_err(ev.cpu().numpy(), ca.CorrelationTorch(4)(a2, b2).detach().cpu().numpy()) < TOL


def test_fused_leaky_epilogue_matches_two_step():
    shp = (2, 16, 12, 20)
    a = dev(hash_uniform(shp, 13)).requires_grad_(True)
    b = dev(hash_uniform(shp, 14)).requires_grad_(True)
    p = (4, 1, 4, 1, 1, 1)
    fused = torch.ops.cerberus.correlation_leaky(a, b, *p, 0.1)
    two = torch.nn.functional.leaky_relu(torch.ops.cerberus.correlation(a, b, *p), 0.1)
    assert torch.equal(fused, two)
    w = dev(hash_uniform(fused.shape, 15))
    ga, gb = torch.autograd.grad((fused * w).sum(), (a, b))
    ha, hb = torch.autograd.grad((two * w).sum(), (a, b))
    assert torch.equal(ga, ha) and torch.equal(gb, hb)


def test_non_contiguous_and_empty_inputs():
    base1 = dev(hash_uniform((2, 8, 10, 24), 21))
    base2 = dev(hash_uniform((2, 8, 10, 24), 22))
    v1, v2 = base1[:, :, :, ::2], base2.transpose(2, 3)[:, :, :12, :].transpose(2, 3)
    assert not v1.is_contiguous()
    p = (4, 1, 4, 1, 1, 1)
    out = torch.ops.cerberus.correlation(v1, v2[:, :, :, :12], *p)
    ref = oracle.corr_forward_ref(v1.cpu().numpy(), v2[:, :, :, :12].cpu().numpy(), 4, 1, 4, 1, 1)
    assert rel_err(out.cpu().numpy(), ref) < TOL
    e = torch.empty(0, 8, 10, 12, device=DEV)
    assert torch.ops.cerberus.correlation(e, e, *p).shape == (0, 81, 10, 12)
    g1, g2 = torch.ops.cerberus.correlation_backward(e, e, torch.empty(0, 81, 10, 12, device=DEV), *p)
    assert g1.shape == e.shape and g2.shape == e.shape


def test_error_behaviour():
    a = dev(hash_uniform((1, 4, 8, 8), 1))
    p = (4, 1, 4, 1, 1, 1)
    with pytest.raises(RuntimeError, match="shapes differ"):
        torch.ops.cerberus.correlation(a, a[:, :, :, :4], *p)
    with pytest.raises(RuntimeError, match="dtypes differ"):
        torch.ops.cerberus.correlation(a, a.half(), *p)
    with pytest.raises(RuntimeError, match="gradOutput shape"):
        torch.ops.cerberus.correlation_backward(a, a, a, *p)
    with pytest.raises(RuntimeError):  # empty output geometry
        torch.ops.cerberus.correlation(a, a, 0, 1, 10, 1, 1, 1)


def test_current_stream_is_honoured_and_graph_capturable():
    a = dev(hash_uniform((1, 32, 16, 24), 31))
    b = dev(hash_uniform((1, 32, 16, 24), 32))
    p = (4, 1, 4, 1, 1, 1)
    want = torch.ops.cerberus.correlation(a, b, *p)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        got = torch.ops.cerberus.correlation(a, b, *p)
    side.synchronize()
    assert torch.equal(got, want)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        cap = torch.ops.cerberus.correlation(a, b, *p)
        cg1, cg2 = torch.ops.cerberus.correlation_backward(a, b, cap, *p)
    cap.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(cap, want)
    e1, e2 = torch.ops.cerberus.correlation_backward(a, b, want, *p)
    assert torch.equal(cg1, e1) and torch.equal(cg2, e2)


def test_native_library_is_loaded():
    import os
    maps = open("/proc/%d/maps" % os.getpid()).read()
    assert "libcerberus_hip.so" in maps


@pytest.mark.parametrize("variant", variants([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13], FWD_EXPERIMENTS))
@pytest.mark.parametrize("shape", [(2, 32, 13, 72), (1, 16, 9, 64), (1, 48, 20, 33),
                                   (2, 64, 5, 16), (1, 16, 3, 130), (2, 7, 11, 132)])
def test_every_tuned_forward_variant(variant, shape):
    """Force each tile / channel-split variant of the tuned forward (vector and
    scalar staging paths, ragged tiles) against the C oracle."""
    B, C, H, W = shape
    x1, x2 = hash_uniform(shape, 41), hash_uniform(shape, 42)
    ref = oracle.corr_forward_ref(x1, x2, 4, 1, 4, 1, 1)
    _lib.set_option("corr_fwd_variant", variant)
    try:
        out = run_fwd(x1, x2, (4, 1, 4, 1, 1))
        name = _lib.last_kernel(0)
    finally:
        _lib.set_option("corr_fwd_variant", 0)
    assert name.startswith("corr_fwd_d4"), name
    assert rel_err(out, ref) < TOL


if EXPERIMENTS:   # (variant 17 lives in the experiments library only: not even collected in the product run)
    @pytest.mark.parametrize("shape", [(2, 32, 13, 72), (1, 16, 9, 64), (3, 64, 21, 132), (2, 8, 4, 32), (1, 40, 37, 256),
                                       (4, 32, 128, 256)])
    def test_pipelined_persistent_forward_against_the_oracle(shape):
        """Round 5: the persistent, cross-item pipelined forward (corr_fwd_pipe.hip, variant 17: 4 x 32 tiles, two channel
        halves per wavefront met by v_permlane32_swap, every workgroup walking several tiles behind one run-ahead loader):
        ragged tiles, one and several items per workgroup, batch borders inside a workgroup's walk; the benched level
        against the default kernel (itself pinned against the oracle at that size)."""
        B, C, H, W = shape
        x1, x2 = hash_uniform(shape, 43), hash_uniform(shape, 44)
        if B * H * W > 64 * 1024:
            ref = run_fwd(x1, x2, (4, 1, 4, 1, 1))
        else:
            ref = oracle.corr_forward_ref(x1, x2, 4, 1, 4, 1, 1)
        for grid in (0, 3, 1 << 20):                  # default walk, three workgroups for everything, one tile each
            _lib.set_option("corr_fwd_variant", 17)
            _lib.set_option("corr_bwd_cslice", grid)
            try:
                out = run_fwd(x1, x2, (4, 1, 4, 1, 1))
                name = _lib.last_kernel(0)
            finally:
                _lib.set_option("corr_fwd_variant", 0)
                _lib.set_option("corr_bwd_cslice", 0)
            assert name == "corr_fwd_d4_pipe_4x32_s2", name
            assert rel_err(out, ref) < TOL, grid


@pytest.mark.parametrize("variant", variants([0, 1, 2, 3, 4], BWD_EXPERIMENTS))
@pytest.mark.parametrize("cslice", [0, 2, 4, 8, 1000])
@pytest.mark.parametrize("shape", [(2, 12, 13, 72), (1, 10, 20, 32), (1, 7, 18, 33),
                                   (1, 16, 40, 28), (2, 5, 9, 130), (2, 9, 21, 136)])
def test_tuned_backward_tiles_and_channel_slices(variant, cslice, shape):
    B, C, H, W = shape
    x1, x2 = hash_uniform(shape, 43), hash_uniform(shape, 44)
    go = hash_uniform((B, 81, H, W), 45)
    r1, r2 = oracle.corr_backward_ref(x1, x2, go, 4, 1, 4, 1, 1)
    _lib.set_option("corr_bwd_cslice", cslice)
    _lib.set_option("corr_bwd_variant", variant)
    try:
        g1, g2 = run_bwd(x1, x2, go, (4, 1, 4, 1, 1))
        name = _lib.last_kernel(1)
    finally:
        _lib.set_option("corr_bwd_cslice", 0)
        _lib.set_option("corr_bwd_variant", 0)
    assert name.startswith("corr_bwd_d4"), name
    assert rel_err(g1, r1) < TOL
    assert rel_err(g2, r2) < TOL


@pytest.mark.parametrize("staged,dma,shape", [(3, 10, (2, 32, 13, 72)), (4, 11, (1, 48, 20, 136)),
                                              (5, 12, (2, 64, 5, 48)), (6, 13, (1, 96, 9, 40))])
def test_channel_group_dma_forward_is_bit_identical_to_the_staged_variant(staged, dma, shape):
    """Variants 3..6 (2/4/8/16 lane groups, register-staged) and 10..13 (the same tiles fed by
    the loader wavefront) sum in the same order: identical bits."""
    x1 = torch.from_numpy(hash_uniform(shape, 56)).to(DEV)
    x2 = torch.from_numpy(hash_uniform(shape, 57)).to(DEV)
    outs, names = [], []
    for v in (staged, dma):
        _lib.set_option("corr_fwd_variant", v)
        try:
            outs.append(torch.ops.cerberus.correlation(x1, x2, 4, 1, 4, 1, 1, 1))
            names.append(_lib.last_kernel(0))
        finally:
            _lib.set_option("corr_fwd_variant", 0)
    assert "dma" not in names[0] and "dma" in names[1], names
    assert torch.equal(outs[0], outs[1])
    ref = oracle.corr_forward_ref(x1.cpu().numpy(), x2.cpu().numpy(), 4, 1, 4, 1, 1)
    assert rel_err(outs[1].cpu().numpy(), ref) < TOL


@pytest.mark.parametrize("shape", [(2, 32, 24, 128), (1, 6, 8, 64), (3, 17, 37, 196), (2, 24, 21, 32),
                                   (1, 7, 16, 28)])
def test_dma_kernels_are_bit_identical_to_the_register_staged_ones(shape):
    """The LDS-DMA kernels change the data movement only (same tile, same lane mapping, same
    summation order): their results must equal the register-staged kernels bit for bit."""
    B, C, H, W = shape
    x1 = torch.from_numpy(hash_uniform(shape, 46)).to(DEV)
    x2 = torch.from_numpy(hash_uniform(shape, 47)).to(DEV)
    go = torch.from_numpy(hash_uniform((B, 81, H, W), 48)).to(DEV)
    res = {}
    for tag, fv, bv in (("staged", 7, 1), ("dma", 9, 4)):
        _lib.set_option("corr_fwd_variant", fv)
        _lib.set_option("corr_bwd_variant", bv)
        try:
            out = torch.ops.cerberus.correlation(x1, x2, 4, 1, 4, 1, 1, 1)
            fname = _lib.last_kernel(0)
            g1, g2 = torch.ops.cerberus.correlation_backward(x1, x2, go, 4, 1, 4, 1, 1, 1)
            bname = _lib.last_kernel(1)
        finally:
            _lib.set_option("corr_fwd_variant", 0)
            _lib.set_option("corr_bwd_variant", 0)
        res[tag] = (out, g1, g2, fname, bname)
    assert "dma" in res["dma"][3] and "dma" in res["dma"][4], res["dma"][3:]
    assert "dma" not in res["staged"][3] and "dma" not in res["staged"][4], res["staged"][3:]
    for a, b in zip(res["staged"][:3], res["dma"][:3]):
        assert torch.equal(a, b)


def test_headline_shapes_use_the_tuned_kernels():
    for C, H, W in W32_PYRAMID_1024x512:
        a = torch.zeros(1, C, H, W, device=DEV)
        out = torch.ops.cerberus.correlation(a, a, 4, 1, 4, 1, 1, 1)
        assert _lib.last_kernel(0).startswith("corr_fwd_d4"), _lib.last_kernel(0)
        torch.ops.cerberus.correlation_backward(a, a, out, 4, 1, 4, 1, 1, 1)
        assert _lib.last_kernel(1).startswith("corr_bwd_d4"), _lib.last_kernel(1)


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 2e-3), (torch.bfloat16, 1.6e-2)])
@pytest.mark.parametrize("lvl", [0, 3])
def test_half_storage_takes_the_tuned_kernels(dtype, tol, lvl):
    """AMP configuration (BASELINE config 5 uses fp16): 16-bit storage runs on the tuned
    kernels (fp32 accumulation), checked against the fp64 oracle on the rounded inputs."""
    C, H, W = W32_PYRAMID_1024x512[lvl]
    H, W = min(H, 32), min(W, 64)          # keep the CPU oracle fast
    shp = (1, C, H, W)
    x1 = torch.from_numpy(hash_uniform(shp, 51)).to(dtype)
    x2 = torch.from_numpy(hash_uniform(shp, 52)).to(dtype)
    go = torch.from_numpy(hash_uniform((1, 81, H, W), 53)).to(dtype)
    p = (4, 1, 4, 1, 1)
    out = torch.ops.cerberus.correlation(x1.to(DEV), x2.to(DEV), *p, 1)
    assert _lib.last_kernel(0).startswith("corr_fwd_d4"), _lib.last_kernel(0)
    g1, g2 = torch.ops.cerberus.correlation_backward(x1.to(DEV), x2.to(DEV), go.to(DEV), *p, 1)
    assert _lib.last_kernel(1).startswith("corr_bwd_d4"), _lib.last_kernel(1)
    assert out.dtype == dtype and g1.dtype == dtype
    ref = oracle.corr_forward_ref(x1.double().numpy(), x2.double().numpy(), *p)
    r1, r2 = oracle.corr_backward_ref(x1.double().numpy(), x2.double().numpy(),
                                      go.double().numpy(), *p)
    assert rel_err(out.double().cpu().numpy(), ref) < tol
    assert rel_err(g1.double().cpu().numpy(), r1) < tol
    assert rel_err(g2.double().cpu().numpy(), r2) < tol
    # odd width -> no 8-byte groups -> generic kernels, same numbers
    xo1, xo2 = x1[..., :W - 1].contiguous().to(DEV), x2[..., :W - 1].contiguous().to(DEV)
    outo = torch.ops.cerberus.correlation(xo1, xo2, *p, 1)
    assert _lib.last_kernel(0) == "corr_fwd_generic"
    refo = oracle.corr_forward_ref(xo1.double().cpu().numpy(), xo2.double().cpu().numpy(), *p)
    assert rel_err(outo.double().cpu().numpy(), refo) < tol


CONFIG5_PYRAMID_2048x1024 = [(256, 32, 64), (128, 64, 128), (64, 128, 256), (32, 256, 512)]


@pytest.mark.parametrize("lvl", [0, 1, 2, 3])
def test_config5_fp16_level_shapes_at_full_size(lvl):
    """BASELINE config 5 (AMP fp16, 2048x1024 -> W32 levels) at its full level sizes, B = 1.
    The CPU oracle is too slow there, so: (i) the fp16 result must equal the fp32 kernels run
    on the same fp16-rounded inputs up to the fp16 rounding of the output (both accumulate in
    fp32), (ii) the adjoint identity <corr(x1,x2), gO> pairing with the gradients, and (iii) a
    full-size window of the output against the C oracle."""
    C, H, W = CONFIG5_PYRAMID_2048x1024[lvl]
    shp = (1, C, H, W)
    x1 = torch.from_numpy(hash_uniform(shp, 61)).half().to(DEV)
    x2 = torch.from_numpy(hash_uniform(shp, 62)).half().to(DEV)
    go = torch.from_numpy(hash_uniform((1, 81, H, W), 63)).half().to(DEV)
    p = (4, 1, 4, 1, 1, 1)
    out = torch.ops.cerberus.correlation(x1, x2, *p)
    assert _lib.last_kernel(0).startswith("corr_fwd_d4"), _lib.last_kernel(0)
    g1, g2 = torch.ops.cerberus.correlation_backward(x1, x2, go, *p)
    assert _lib.last_kernel(1).startswith("corr_bwd_d4"), _lib.last_kernel(1)
    o32 = torch.ops.cerberus.correlation(x1.float(), x2.float(), *p)
    h1, h2 = torch.ops.cerberus.correlation_backward(x1.float(), x2.float(), go.float(), *p)
    for a, b in ((out, o32), (g1, h1), (g2, h2)):
        scale = float(b.abs().max())
        assert float((a.float() - b).abs().max()) <= 1.5e-3 * scale   # fp16 output rounding
    # (ii) <corr(x1,x2), gO> = <x1, g1> = <x2, g2> (bilinear form), in fp64 on the fp32 results
    lhs = float((o32.double() * go.double()).sum())
    r1 = float((x1.double() * h1.double()).sum())
    r2 = float((x2.double() * h2.double()).sum())
    mag = float((o32.double() * go.double()).abs().sum())
    assert abs(lhs - r1) <= 1e-6 * mag and abs(lhs - r2) <= 1e-6 * mag
    # (iii) a corner window (includes the zero padding) against the C oracle
    hh, ww = min(H, 24), min(W, 40)
    ref = oracle.corr_forward_ref(x1[..., :hh + 4, :ww + 4].double().cpu().numpy(),
                                  x2[..., :hh + 4, :ww + 4].double().cpu().numpy(), 4, 1, 4, 1, 1)
    got = o32[..., :hh, :ww].double().cpu().numpy()
    assert rel_err(got, ref[..., :hh, :ww]) < TOL


@pytest.mark.parametrize("C", [1, 2, 3, 5])
@pytest.mark.parametrize("hw", [(8, 64), (5, 32), (17, 132)])
def test_dma_pipelines_with_fewer_chunks_than_ring_buffers(C, hw):
    """One or two channel chunks only: the LDS ring never fills, the loader / issue logic
    runs entirely in its prologue and tail paths.  Against the C oracle, forward + backward."""
    H, W = hw
    shape = (2, C, H, W)
    x1, x2 = hash_uniform(shape, 81), hash_uniform(shape, 82)
    go = hash_uniform((2, 81, H, W), 83)
    out = run_fwd(x1, x2, (4, 1, 4, 1, 1))
    assert "dma" in _lib.last_kernel(0), _lib.last_kernel(0)
    g1, g2 = run_bwd(x1, x2, go, (4, 1, 4, 1, 1))
    assert any(k in _lib.last_kernel(1) for k in ("dma", "g3", "rows", "strip")), _lib.last_kernel(1)
    ref = oracle.corr_forward_ref(x1, x2, 4, 1, 4, 1, 1)
    r1, r2 = oracle.corr_backward_ref(x1, x2, go, 4, 1, 4, 1, 1)
    assert rel_err(out, ref) < TOL
    assert rel_err(g1, r1) < TOL and rel_err(g2, r2) < TOL


def test_seeded_random_shape_sweep_tuned_vs_generic():
    """60 seeded random shapes (odd channel counts, ragged tiles, tiny and wide maps, W % 4 == 0
    and not): every tuned dispatch (LDS-DMA, register-staged, three-group) against the generic
    kernels, which the other tests pin to the oracle.  Forward and both gradients."""
    import numpy as np
    rng = np.random.RandomState(20240607)
    kernels = set()
    for trial in range(60):
        B = int(rng.randint(1, 4))
        C = int(rng.choice([1, 2, 3, 5, 8, 12, 17, 32, 48, 64, 96]))
        H = int(rng.randint(1, 41))
        W = int(rng.choice([4, 8, 12, 20, 28, 32, 36, 64, 68, 100, 128, 132, 200])) + int(rng.randint(0, 2)) * int(rng.randint(0, 4))
        shape = (B, C, H, W)
        x1 = torch.from_numpy(hash_uniform(shape, 900 + trial)).to(DEV)
        x2 = torch.from_numpy(hash_uniform(shape, 1900 + trial)).to(DEV)
        go = torch.from_numpy(hash_uniform((B, 81, H, W), 2900 + trial)).to(DEV)
        p = (4, 1, 4, 1, 1, 1)
        out = torch.ops.cerberus.correlation(x1, x2, *p)
        kernels.add(_lib.last_kernel(0))
        g1, g2 = torch.ops.cerberus.correlation_backward(x1, x2, go, *p)
        kernels.add(_lib.last_kernel(1))
        _lib.set_option("corr_force_generic", 1)
        try:
            ref = torch.ops.cerberus.correlation(x1, x2, *p)
            r1, r2 = torch.ops.cerberus.correlation_backward(x1, x2, go, *p)
        finally:
            _lib.set_option("corr_force_generic", 0)
        for a, b, what in ((out, ref, "out"), (g1, r1, "g1"), (g2, r2, "g2")):
            scale = float(b.abs().max()) or 1.0
            err = float((a - b).abs().max()) / scale
            assert err < TOL, (shape, what, err, sorted(kernels))
    assert any("dma" in k for k in kernels) and any("generic" not in k for k in kernels), kernels


@pytest.mark.parametrize("variant", variants([6, 7, 8, 9, 10], BWD_EXPERIMENTS))
@pytest.mark.parametrize("shape", [(2, 32, 13, 72), (1, 10, 20, 64), (1, 7, 18, 36), (2, 40, 9, 132),
                                   (1, 64, 4, 64), (3, 33, 37, 196), (1, 1, 1, 68), (2, 12, 8, 256)])
def test_row_streaming_backward_against_the_oracle(shape, variant):
    """Variant 6 (accumulators in registers, the nine vertical displacements streamed through
    an LDS ring by LDS-DMA, all channels of a workgroup at once): ragged tiles in both
    directions, channel ranges that do not fill a workgroup (C not a multiple of 32) and more
    than one range, the shifted gradOutput slots of the second gradient at every border."""
    B, C, H, W = shape
    x1, x2 = hash_uniform(shape, 143), hash_uniform(shape, 144)
    go = hash_uniform((B, 81, H, W), 145)
    r1, r2 = oracle.corr_backward_ref(x1, x2, go, 4, 1, 4, 1, 1)
    _lib.set_option("corr_bwd_variant", variant)
    try:
        g1, g2 = run_bwd(x1, x2, go, (4, 1, 4, 1, 1))
        name = _lib.last_kernel(1)
        g1b, g2b = run_bwd(x1, x2, go, (4, 1, 4, 1, 1))
    finally:
        _lib.set_option("corr_bwd_variant", 0)
    assert name.startswith("corr_bwd_d4_rows_4x64") or name == "corr_bwd_d4_col_4x64", name
    assert rel_err(g1, r1) < TOL
    assert rel_err(g2, r2) < TOL
    assert np.array_equal(g1, g1b) and np.array_equal(g2, g2b)


def test_row_streaming_backward_keeps_nonfinite_gradients_local():
    """Out-of-image gradOutput slots read exact zeros (buffer bounds), so a NaN / Inf in
    gradOutput reaches exactly the gradient elements the reference's sums touch."""
    shape = (1, 8, 12, 64)
    x1, x2 = hash_uniform(shape, 151), hash_uniform(shape, 152)
    go = hash_uniform((1, 81, 12, 64), 153)
    go[0, 40, 5, 0] = np.nan       # centre displacement, left border pixel
    go[0, 3, 0, 63] = np.inf       # top-right corner
    r1, r2 = oracle.corr_backward_ref(x1, x2, go, 4, 1, 4, 1, 1)
    _lib.set_option("corr_bwd_variant", 8)     # the row-streaming kernel the dispatcher uses (medium maps)
    try:
        g1, g2 = run_bwd(x1, x2, go, (4, 1, 4, 1, 1))
        assert _lib.last_kernel(1).startswith("corr_bwd_d4_rows_4x64"), _lib.last_kernel(1)
    finally:
        _lib.set_option("corr_bwd_variant", 0)
    assert np.array_equal(np.isfinite(g1), np.isfinite(r1))
    assert np.array_equal(np.isfinite(g2), np.isfinite(r2))
    ok1, ok2 = np.isfinite(r1), np.isfinite(r2)
    assert rel_err(np.where(ok1, g1, 0), np.where(ok1, r1, 0)) < TOL
    assert rel_err(np.where(ok2, g2, 0), np.where(ok2, r2, 0)) < TOL


@pytest.mark.parametrize("shape,total,off", [((2, 32, 13, 72), 115, 0), ((1, 16, 9, 64), 100, 7),
                                             ((2, 7, 11, 132), 83, 2), ((4, 32, 128, 256), 115, 0),
                                             ((1, 10, 20, 33), 90, 4)])
def test_concat_buffer_output_against_the_oracle(shape, total, off):
    """the cost volume (fused LeakyReLU) lands in channels [off, off+81) of a
    wider buffer (batch stride = total*H*W); every other channel of the buffer stays
    untouched.  Vector, scalar (W % 4 != 0) and unaligned-offset paths, against the C oracle."""
    B, C, H, W = shape
    x1, x2 = hash_uniform(shape, 161), hash_uniform(shape, 162)
    ref = oracle.corr_forward_ref(x1, x2, 4, 1, 4, 1, 1)
    ref = np.where(ref > 0, ref, ref * np.float32(0.1))
    buf = torch.full((B, total, H, W), 7.5, device=DEV)
    torch.ops.cerberus.correlation_leaky_into(buf, dev(x1), dev(x2), off, 4, 1, 4, 1, 1, 1, 0.1)
    got = buf.cpu().numpy()
    assert rel_err(got[:, off:off + 81], ref) < TOL
    rest = np.delete(got, np.s_[off:off + 81], axis=1)
    assert np.all(rest == 7.5)
    with pytest.raises(RuntimeError, match="cannot hold"):
        torch.ops.cerberus.correlation_leaky_into(buf, dev(x1), dev(x2), total - 80, 4, 1, 4, 1, 1, 1, 0.1)
    with pytest.raises(RuntimeError, match="cannot hold"):
        torch.ops.cerberus.correlation_leaky_into(buf[:, :, :, :-1], dev(x1), dev(x2), 0, 4, 1, 4, 1, 1, 1, 0.1)


def test_concat_function_equals_cat_of_the_separate_ops():
    """CostVolumeConcat (forward values and every gradient) against
    torch.cat([correlation_leaky(...), a, b]) built from the separate ops."""
    from cerberusnet_amd.correlation_package.correlation import cost_volume_concat
    shape = (2, 16, 24, 64)
    mk = lambda s, seed: dev(hash_uniform(s, seed)).requires_grad_(True)
    x1, x2, a, b = mk(shape, 171), mk(shape, 172), mk((2, 32, 24, 64), 173), mk((2, 2, 24, 64), 174)
    w = dev(hash_uniform((2, 81 + 34, 24, 64), 175))
    hyper = (4, 1, 4, 1, 1, 1)
    fused = cost_volume_concat(x1, x2, [a, b], hyper, 0.1)
    gf = torch.autograd.grad((fused * w).sum(), (x1, x2, a, b))
    sep = torch.cat([torch.ops.cerberus.correlation_leaky(x1, x2, *hyper, 0.1), a, b], dim=1)
    gs = torch.autograd.grad((sep * w).sum(), (x1, x2, a, b))
    assert torch.equal(fused, sep)
    for u, v in zip(gf, gs):
        assert torch.equal(u, v)


@pytest.mark.parametrize("dtype,ulp", [(torch.float16, 2.0 ** -10), (torch.bfloat16, 2.0 ** -7)])
def test_mfma_backward_equals_the_valu_backward_up_to_output_rounding(dtype, ulp):
    """16-bit storage: the backward runs on the matrix cores (corr_mfma.hip: band matrix of
    gradOutput x window rows, v_mfma_f32_16x16x32).  fp16 / bf16 products are exact in fp32 and
    both kernels accumulate in fp32, so they may differ only by summation order before the one
    rounding to the storage type: at most one unit of the output's last place (relative to the
    tensor's magnitude).  Shapes: ragged rows and tiles, W a multiple of 4 but not of 64,
    channel counts that are not multiples of 16 / 32, several batch items, a map narrower than
    a tile, and the padding on every border."""
    p = (4, 1, 4, 1, 1, 1)
    for k, shp in enumerate([(2, 32, 9, 68), (1, 24, 17, 132), (3, 7, 5, 12), (1, 64, 33, 64),
                             (1, 256, 16, 32), (2, 40, 4, 200)]):
        B, C, H, W = shp
        x1 = torch.from_numpy(hash_uniform(shp, 910 + k)).to(dtype).to(DEV)
        x2 = torch.from_numpy(hash_uniform(shp, 920 + k)).to(dtype).to(DEV)
        go = torch.from_numpy(hash_uniform((B, 81, H, W), 930 + k)).to(dtype).to(DEV)
        g1, g2 = torch.ops.cerberus.correlation_backward(x1, x2, go, *p)
        assert _lib.last_kernel(1) == "corr_bwd_d4_mfma_seg_4x64", _lib.last_kernel(1)
        _lib.set_option("corr_bwd_variant", 1)
        try:
            v1, v2 = torch.ops.cerberus.correlation_backward(x1, x2, go, *p)
            assert "mfma" not in _lib.last_kernel(1)
        finally:
            _lib.set_option("corr_bwd_variant", 0)
        for a, b in ((g1, v1), (g2, v2)):
            scale = float(b.float().abs().max())
            assert float((a.float() - b.float()).abs().max()) <= ulp * scale, shp
        if H * W * C <= 40000:
            r1, r2 = oracle.corr_backward_ref(x1.double().cpu().numpy(), x2.double().cpu().numpy(),
                                              go.double().cpu().numpy(), 4, 1, 4, 1, 1)
            tol = 2e-3 if dtype == torch.float16 else 1.6e-2
            assert rel_err(g1.double().cpu().numpy(), r1) < tol
            assert rel_err(g2.double().cpu().numpy(), r2) < tol


@pytest.mark.parametrize("dtype,ulp", [(torch.float16, 2.0 ** -10), (torch.bfloat16, 2.0 ** -7)])
def test_mfma_forward_equals_the_valu_forward_up_to_output_rounding(dtype, ulp):
    """16-bit storage, C <= 128: the forward runs on the matrix cores (pixel x window-column
    products per 32 channels, the wanted diagonals extracted through a per-wave LDS tile).  Same
    exact products, fp32 sums, one rounding: at most one unit of the output's last place from
    the VALU kernel.  Covers the three tile shapes (C <= 32: 4 x 64, C <= 64: 4 x 32, C <= 128:
    4 x 16 -- round 5), ragged tiles, odd channel counts, LeakyReLU fused, the strided
    (concat-buffer) output, and NaN / Inf reaching exactly the same outputs; small shapes also
    against the oracle."""
    p = (4, 1, 4, 1, 1, 1)
    for k, shp in enumerate([(2, 32, 9, 68), (1, 24, 17, 132), (3, 7, 5, 12), (1, 64, 33, 64),
                             (2, 40, 6, 200), (1, 33, 4, 36), (2, 128, 9, 36), (1, 65, 17, 68),
                             (3, 100, 5, 12), (1, 96, 33, 132), (4, 128, 32, 64)]):
        B, C, H, W = shp
        a1 = hash_uniform(shp, 940 + k)
        a2 = hash_uniform(shp, 950 + k)
        a1[0, 0, H // 2, W // 3] = np.inf
        a2[-1, C - 1, 0, W - 1] = np.nan
        x1 = torch.from_numpy(a1).to(dtype).to(DEV)
        x2 = torch.from_numpy(a2).to(dtype).to(DEV)
        _lib.set_option("corr_fwd_variant", 0 if C > 16 else 14)   # auto skips it for C <= 16
        out = torch.ops.cerberus.correlation_leaky(x1, x2, *p, 0.1)
        assert _lib.last_kernel(0).startswith("corr_fwd_d4_mfma"), _lib.last_kernel(0)
        buf = torch.zeros(B, 81 + 5, H, W, dtype=dtype, device=DEV)
        torch.ops.cerberus.correlation_leaky_into(buf, x1, x2, 3, *p, 0.1)
        _lib.set_option("corr_fwd_variant", 7)
        try:
            ref = torch.ops.cerberus.correlation_leaky(x1, x2, *p, 0.1)
            assert "mfma" not in _lib.last_kernel(0)
        finally:
            _lib.set_option("corr_fwd_variant", 0)
        assert torch.equal(torch.isnan(out), torch.isnan(ref)), shp
        assert torch.equal(torch.isinf(out), torch.isinf(ref)), shp
        fin = torch.isfinite(ref)
        scale = float(ref[fin].float().abs().max())
        assert float((out[fin].float() - ref[fin].float()).abs().max()) <= ulp * scale, shp
        assert torch.equal(buf[:, 3:84].view(torch.int16), out.view(torch.int16))
        assert float(buf[:, :3].abs().max()) == 0.0 and float(buf[:, 84:].abs().max()) == 0.0
        if H * W * C <= 45000:
            c1, c2 = np.where(np.isfinite(a1), a1, 0.0), np.where(np.isfinite(a2), a2, 0.0)
            y1 = torch.from_numpy(c1.astype(np.float32)).to(dtype)
            y2 = torch.from_numpy(c2.astype(np.float32)).to(dtype)
            got = torch.ops.cerberus.correlation(y1.to(DEV), y2.to(DEV), *p)
            assert _lib.last_kernel(0).startswith("corr_fwd_d4_mfma") or C <= 16
            want = oracle.corr_forward_ref(y1.double().numpy(), y2.double().numpy(), 4, 1, 4, 1, 1)
            assert rel_err(got.double().cpu().numpy(), want) < (2e-3 if dtype == torch.float16 else 1.6e-2), shp


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_mfma_backward_keeps_nonfinite_values_local(dtype):
    """The band matrix of the matrix-core backward holds explicit zeros, and 0 x Inf = NaN inside
    an MFMA: left alone, an Inf / NaN of the source would spread over its 16-pixel segment.  A row
    with a non-finite accumulator is therefore recomputed tap by tap; NaN and Inf must land on
    exactly the elements the vector kernel gives them (same masks), finite values must agree."""
    p = (4, 1, 4, 1, 1, 1)
    shp = (2, 40, 13, 132)
    B, C, H, W = shp
    a1, a2 = hash_uniform(shp, 971), hash_uniform(shp, 972)
    g = hash_uniform((B, 81, H, W), 973)
    a1[0, 3, 5, 17] = np.inf
    a1[1, 39, 12, 131] = np.nan
    a2[0, 0, 0, 0] = -np.inf
    a2[1, 20, 6, 64] = np.nan
    a2[0, 33, 9, 70] = np.inf
    g[0, 40, 2, 100] = np.nan
    g[1, 0, 12, 3] = np.inf
    x1 = torch.from_numpy(a1).to(dtype).to(DEV)
    x2 = torch.from_numpy(a2).to(dtype).to(DEV)
    go = torch.from_numpy(g).to(dtype).to(DEV)
    g1, g2 = torch.ops.cerberus.correlation_backward(x1, x2, go, *p)
    assert _lib.last_kernel(1) == "corr_bwd_d4_mfma_seg_4x64"
    _lib.set_option("corr_bwd_variant", 1)
    try:
        v1, v2 = torch.ops.cerberus.correlation_backward(x1, x2, go, *p)
    finally:
        _lib.set_option("corr_bwd_variant", 0)
    ulp = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
    for a, b in ((g1, v1), (g2, v2)):
        assert torch.equal(torch.isnan(a), torch.isnan(b))
        assert torch.equal(torch.isinf(a), torch.isinf(b))
        assert int(torch.isnan(b).sum()) > 0 and int(torch.isfinite(b).sum()) > b.numel() // 2
        fin = torch.isfinite(b)
        assert torch.equal(torch.sign(a[torch.isinf(b)]), torch.sign(b[torch.isinf(b)]))
        scale = float(b[fin].float().abs().max())
        assert float((a[fin].float() - b[fin].float()).abs().max()) <= ulp * scale


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_mfma_forward_column_walk_equals_the_register_staged_form_bit_for_bit(dtype):
    """Round 6: the matrix-core forward for 16 < C <= 64, W % 8 == 0 keeps its tiles in LDS as they lie in memory (LDS-DMA,
    no registers), takes the MFMA operands with the transposing read ds_read_b64_tr_b16 and walks down a column of 4 x 32
    tiles with the window rows in a ring, the next tile's copies in flight behind the current tile's MFMAs and stores
    (`corr_fwd_d4_mfma_walk_4x32`, `..._c64` with 64 channel slots); variant 26 is the same without the walk (4 x 64 tiles,
    C <= 32, `..._tr_4x64`), variant 20 the register-staged form of rounds 4-5.  Same operands, same MFMAs, same rounding: identical bits -- every border, ragged
    tiles, one to many tiles per walk, channel counts below 32, the fused LeakyReLU in both forms (0 < slope <= 1 is a
    max, anything else a select), the strided output, NaN / Inf in the inputs."""
    p = (4, 1, 4, 1, 1, 1)
    shapes = [(2, 32, 8, 64), (1, 20, 13, 72), (3, 24, 5, 136), (2, 32, 4, 8), (1, 17, 9, 200), (2, 32, 30, 64),
              (1, 32, 64, 320), (2, 32, 37, 96), (1, 32, 100, 32), (1, 28, 23, 40), (4, 32, 128, 256), (1, 32, 1, 16),
              (2, 64, 9, 72), (1, 40, 13, 40), (3, 33, 5, 136), (1, 64, 37, 96), (4, 64, 64, 128), (1, 48, 70, 32)]
    for k, shp in enumerate(shapes):
        B, C, H, W = shp
        a1, a2 = hash_uniform(shp, 2610 + k), hash_uniform(shp, 2630 + k)
        if k % 3 == 1:
            a1[0, 0, H // 2, W // 3] = np.inf
            a2[-1, C - 1, 0, W - 1] = np.nan
            a2[0, 1, H - 1, 0] = -np.inf
        x1, x2 = torch.from_numpy(a1).to(dtype).to(DEV), torch.from_numpy(a2).to(dtype).to(DEV)
        for slope in ((0.1, 1.0, 0.0, 2.5, -0.5) if k < 4 else (0.1,)):
            outs = {}
            forms = (((0, "corr_fwd_d4_mfma_walk_4x32"), (26, "corr_fwd_d4_mfma_tr_4x64"), (20, "corr_fwd_d4_mfma_4x64")) if C <= 32 else
                     ((0, "corr_fwd_d4_mfma_walk_4x32_c64"), (20, "corr_fwd_d4_mfma_4x32")))
            for v, name in forms:
                _lib.set_option("corr_fwd_variant", v)
                try:
                    outs[v] = torch.ops.cerberus.correlation_leaky(x1, x2, *p, slope)
                    assert _lib.last_kernel(0) == name, (_lib.last_kernel(0), shp)
                finally:
                    _lib.set_option("corr_fwd_variant", 0)
            ref = outs[20]
            for v in [f[0] for f in forms if f[0] != 20]:
                assert torch.equal(torch.isnan(outs[v]), torch.isnan(ref)), (shp, slope, v)
                keep = ~torch.isnan(ref)
                assert torch.equal(outs[v][keep].view(torch.int16), ref[keep].view(torch.int16)), (shp, slope, v)
        buf = torch.zeros(B, 81 + 5, H, W, dtype=dtype, device=DEV)
        torch.ops.cerberus.correlation_leaky_into(buf, x1, x2, 3, *p, 0.1)
        assert _lib.last_kernel(0) == forms[0][1]
        plain = torch.ops.cerberus.correlation_leaky(x1, x2, *p, 0.1)
        keep = ~torch.isnan(plain)
        assert torch.equal(buf[:, 3:84][keep].view(torch.int16), plain[keep].view(torch.int16)), shp
        assert float(buf[:, :3].abs().max()) == 0.0 and float(buf[:, 84:].abs().max()) == 0.0
    # tiles per walk forced (the option the backward's walk uses): 1, 3 and more tiles than the column has
    x1 = torch.from_numpy(hash_uniform((2, 32, 45, 96), 2660)).to(dtype).to(DEV)
    x2 = torch.from_numpy(hash_uniform((2, 32, 45, 96), 2661)).to(dtype).to(DEV)
    ref = torch.ops.cerberus.correlation(x1, x2, *p)
    for nwalk in (1, 3, 5, 64):
        _lib.set_option("corr_bwd_cslice", nwalk)
        try:
            got = torch.ops.cerberus.correlation(x1, x2, *p)
        finally:
            _lib.set_option("corr_bwd_cslice", 0)
        assert torch.equal(got.view(torch.int16), ref.view(torch.int16)), nwalk


def test_mfma_forward_column_walk_soak_with_a_busy_second_stream():
    """The column walk synchronises by hand (counted `s_waitcnt vmcnt`, raw barriers, an LDS-DMA ring the compiler is kept out
    of): 400 random shapes, three launches each, half of them while a second stream keeps the memory system busy, some with
    forced tiles per walk -- every result bit for bit the register-staged form's.  (`tools/soak_corr16.py` is the long
    version: 1.6 M launches without a mismatch in round 6.)"""
    p = (4, 1, 4, 1, 1, 1)
    rng = np.random.default_rng(20261004)
    side = torch.cuda.Stream()
    noise_a = torch.randn(16 << 20, device=DEV)
    noise_b = torch.empty_like(noise_a)
    for it in range(400):
        B, C, H = int(rng.integers(1, 5)), int(rng.integers(17, 65)), int(rng.integers(1, 97))
        W = 8 * int(rng.integers(1, 41))
        dt = torch.float16 if it % 2 else torch.bfloat16
        x1 = torch.randn(B, C, H, W, device=DEV).to(dt)
        x2 = torch.randn(B, C, H, W, device=DEV).to(dt)
        slope = float(rng.choice([0.1, 1.0, 0.0, 2.0]))
        _lib.set_option("corr_fwd_variant", 20)
        try:
            ref = torch.ops.cerberus.correlation_leaky(x1, x2, *p, slope)
        finally:
            _lib.set_option("corr_fwd_variant", 0)
        _lib.set_option("corr_bwd_cslice", int(rng.choice([0, 0, 1, 2, 3, 7])))
        try:
            if it % 2:
                with torch.cuda.stream(side):
                    noise_b.copy_(noise_a)
            outs = [torch.ops.cerberus.correlation_leaky(x1, x2, *p, slope) for _ in range(3)]
            assert "walk" in _lib.last_kernel(0)
        finally:
            _lib.set_option("corr_bwd_cslice", 0)
        torch.cuda.synchronize()
        for o in outs:
            assert torch.equal(o.view(torch.int16), ref.view(torch.int16)), (B, C, H, W, dt, slope)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_mfma_backward_segment_per_wave_equals_row_per_wave_bit_for_bit(dtype):
    """Round 5: the matrix-core backward with a wave per 16-pixel segment (window rows kept in registers across the
    displacement steps, gradOutput fetched as dwords by lane pairs that swap halves, the odd shifts of side 1 as
    halfwords) runs the same MFMAs on the same operands in the same order as the row-per-wave kernel it replaces
    (variant 11): identical bits on finite data -- ragged tiles, every border, several walks, channel slices."""
    p = (4, 1, 4, 1, 1, 1)
    for k, shp in enumerate([(2, 32, 9, 68), (1, 24, 17, 132), (3, 7, 5, 12), (1, 64, 33, 64), (1, 256, 16, 32),
                             (2, 40, 4, 200), (2, 40, 13, 132), (4, 32, 128, 256), (1, 33, 70, 260)]):
        B, C, H, W = shp
        x1 = torch.from_numpy(hash_uniform(shp, 1910 + k)).to(dtype).to(DEV)
        x2 = torch.from_numpy(hash_uniform(shp, 1920 + k)).to(dtype).to(DEV)
        go = torch.from_numpy(hash_uniform((B, 81, H, W), 1930 + k)).to(dtype).to(DEV)
        g1, g2 = torch.ops.cerberus.correlation_backward(x1, x2, go, *p)
        assert _lib.last_kernel(1) == "corr_bwd_d4_mfma_seg_4x64", _lib.last_kernel(1)
        _lib.set_option("corr_bwd_variant", 11)
        try:
            r1, r2 = torch.ops.cerberus.correlation_backward(x1, x2, go, *p)
            assert _lib.last_kernel(1) == "corr_bwd_d4_mfma_4x64", _lib.last_kernel(1)
        finally:
            _lib.set_option("corr_bwd_variant", 0)
        assert torch.equal(g1.view(torch.int16), r1.view(torch.int16)), shp
        assert torch.equal(g2.view(torch.int16), r2.view(torch.int16)), shp


@pytest.mark.parametrize("dtype,ulp", [(torch.float16, 2.0 ** -10), (torch.bfloat16, 2.0 ** -7)])
def test_seeded_random_shape_sweep_16bit_tuned_vs_generic(dtype, ulp):
    """40 seeded random shapes in 16-bit storage: whatever the dispatcher picks (matrix-core
    forward for 16 < C <= 128, vector forward otherwise, matrix-core backward walking 1..n tiles,
    generic kernels when W % 4 != 0) against the generic 16-bit kernels (fp32 accumulation, same
    exact products), forward and both gradients: at most two units of the output's last place."""
    rng = np.random.RandomState(20241003)
    kernels = set()
    p = (4, 1, 4, 1, 1, 1)
    for trial in range(40):
        B = int(rng.randint(1, 4))
        C = int(rng.choice([1, 3, 8, 16, 17, 24, 32, 33, 48, 64, 65, 96, 130]))
        H = int(rng.randint(1, 70))
        W = int(rng.choice([4, 8, 12, 20, 32, 36, 64, 68, 100, 128, 132, 200, 260])) + int(rng.randint(0, 2)) * int(rng.randint(0, 4))
        shape = (B, C, H, W)
        x1 = torch.from_numpy(hash_uniform(shape, 5900 + trial)).to(dtype).to(DEV)
        x2 = torch.from_numpy(hash_uniform(shape, 6900 + trial)).to(dtype).to(DEV)
        go = torch.from_numpy(hash_uniform((B, 81, H, W), 7900 + trial)).to(dtype).to(DEV)
        out = torch.ops.cerberus.correlation(x1, x2, *p)
        kernels.add(_lib.last_kernel(0))
        g1, g2 = torch.ops.cerberus.correlation_backward(x1, x2, go, *p)
        kernels.add(_lib.last_kernel(1))
        _lib.set_option("corr_force_generic", 1)
        try:
            ref = torch.ops.cerberus.correlation(x1, x2, *p)
            r1, r2 = torch.ops.cerberus.correlation_backward(x1, x2, go, *p)
        finally:
            _lib.set_option("corr_force_generic", 0)
        for a, b, what in ((out, ref, "out"), (g1, r1, "g1"), (g2, r2, "g2")):
            scale = float(b.float().abs().max()) or 1.0
            err = float((a.float() - b.float()).abs().max()) / scale
            assert err <= 2 * ulp, (shape, what, err, sorted(kernels))
    assert any("mfma" in k and "fwd" in k for k in kernels) and any("mfma" in k and "bwd" in k for k in kernels), kernels


# ---- round 3: the strip backward (corr_strip.hip) ------------------------------------------
STRIP_SHAPES = [(2, 32, 16, 256), (1, 64, 6, 256), (5, 32, 10, 256), (1, 32, 12, 128), (2, 64, 8, 128),
                (2, 32, 8, 64), (1, 16, 16, 64), (1, 96, 24, 64)]


@pytest.mark.parametrize("shape", STRIP_SHAPES)
def test_strip_backward_against_the_oracle(shape):
    """corr_bwd_d4_strip_kernel (whole image rows per wavefront, neighbours by DPP, gradOutput
    streamed through LDS-DMA, cyclic step order): every supported width -- 256 (wave_shr), 128
    (wave_shr + edge select), 64 (row_shr) --, rows at both image borders in one workgroup,
    several batch items (every rotation of the step order), against the C oracle; two launches
    are bit-identical."""
    B, C, H, W = shape
    x1, x2 = hash_uniform(shape, 401), hash_uniform(shape, 402)
    go = hash_uniform((B, 81, H, W), 403)
    r1, r2 = oracle.corr_backward_ref(x1, x2, go, 4, 1, 4, 1, 1)
    _lib.set_option("corr_bwd_variant", 12)
    try:
        g1, g2 = run_bwd(x1, x2, go, (4, 1, 4, 1, 1))
        name = _lib.last_kernel(1)
        g1b, g2b = run_bwd(x1, x2, go, (4, 1, 4, 1, 1))
    finally:
        _lib.set_option("corr_bwd_variant", 0)
    assert name.startswith("corr_bwd_d4_strip_w%d" % W), name
    assert rel_err(g1, r1) < TOL
    assert rel_err(g2, r2) < TOL
    assert np.array_equal(g1, g1b) and np.array_equal(g2, g2b)


@pytest.mark.parametrize("shape", [(2, 32, 16, 256), (1, 64, 8, 256), (3, 32, 12, 256), (4, 32, 128, 256)])
def test_strip_backward_with_two_row_pairs_per_workgroup(shape):
    """Round 6, VERDICT r5 #4: StripCfg NRP = 2 (option corr_bwd_cslice = 16): sixteen waves = two row pairs x eight channel
    groups, the pairs two steps apart in the cyclic schedule, one gradOutput ring for both -- against the oracle (the
    benched level against the default strip kernel, itself pinned there) and bit-identical to the one-pair form (the same
    per-lane arithmetic in the same order)."""
    B, C, H, W = shape
    x1, x2 = hash_uniform(shape, 451), hash_uniform(shape, 452)
    go = hash_uniform((B, 81, H, W), 453)
    _lib.set_option("corr_bwd_variant", 12)
    try:
        g1d, g2d = run_bwd(x1, x2, go, (4, 1, 4, 1, 1))
        _lib.set_option("corr_bwd_cslice", 16)
        g1, g2 = run_bwd(x1, x2, go, (4, 1, 4, 1, 1))
        name = _lib.last_kernel(1)
    finally:
        _lib.set_option("corr_bwd_variant", 0)
        _lib.set_option("corr_bwd_cslice", 0)
    assert name == "corr_bwd_d4_strip_w256_2pairs", name
    if B * H * W <= 64 * 1024:
        r1, r2 = oracle.corr_backward_ref(x1, x2, go, 4, 1, 4, 1, 1)
        assert rel_err(g1, r1) < TOL and rel_err(g2, r2) < TOL
    assert np.array_equal(g1, g1d) and np.array_equal(g2, g2d)


RAGGED_STRIP_SHAPES = [(2, 32, 14, 224), (1, 64, 7, 152), (2, 32, 11, 112), (1, 40, 9, 76), (2, 24, 13, 56), (1, 16, 5, 28),
                       (1, 33, 6, 252), (3, 7, 3, 4), (1, 36, 17, 132), (2, 48, 10, 68), (1, 20, 8, 64), (1, 30, 6, 256),
                       (1, 32, 7, 128)]


@pytest.mark.parametrize("shape", RAGGED_STRIP_SHAPES)
def test_strip_backward_on_ragged_shapes_against_the_oracle(shape):
    """Round 6: the strip backward on any width that is a multiple of 4 up to 256 (the lanes past a row's
    end load nothing: their zeros are the zero padding the last strip's DPP neighbour shift picks up; the second
    gradient's shifted gradOutput rows are zeroed at the TRUE row end), heights that are not a multiple of the
    workgroup's rows and channel counts that are not a multiple of its channels (waves without channels idle, stores
    masked) -- the level shapes of 896 x 448 and 1216 x 352 frames among them -- against the C oracle, incl. a NaN at a
    row's last pixel that the next row must not see."""
    B, C, H, W = shape
    x1, x2 = hash_uniform(shape, 421), hash_uniform(shape, 422)
    go = hash_uniform((B, 81, H, W), 423)
    x1[0, 0, H // 2, W - 1] = np.inf
    go[0, 44, H - 1, W - 1] = np.nan
    r1, r2 = oracle.corr_backward_ref(x1, x2, go, 4, 1, 4, 1, 1)
    _lib.set_option("corr_bwd_variant", 12)
    try:
        g1, g2 = run_bwd(x1, x2, go, (4, 1, 4, 1, 1))
        name = _lib.last_kernel(1)
        g1b, g2b = run_bwd(x1, x2, go, (4, 1, 4, 1, 1))
    finally:
        _lib.set_option("corr_bwd_variant", 0)
    assert name.startswith("corr_bwd_d4_strip_rag"), name
    assert np.array_equal(np.isnan(g1), np.isnan(r1)) and np.array_equal(np.isinf(g1), np.isinf(r1))
    assert np.array_equal(np.isnan(g2), np.isnan(r2)) and np.array_equal(np.isinf(g2), np.isinf(r2))
    ok1, ok2 = np.isfinite(r1), np.isfinite(r2)
    assert rel_err(np.where(ok1, g1, 0), np.where(ok1, r1, 0)) < TOL
    assert rel_err(np.where(ok2, g2, 0), np.where(ok2, r2, 0)) < TOL
    assert np.array_equal(g1, g1b, equal_nan=True) and np.array_equal(g2, g2b, equal_nan=True)


def test_strip_backward_is_the_default_where_the_wide_level_fills_the_chip():
    shape = (4, 32, 128, 256)
    x1, x2 = dev(hash_uniform(shape, 1)), dev(hash_uniform(shape, 2))
    go = dev(hash_uniform((4, 81, 128, 256), 3))
    torch.ops.cerberus.correlation_backward(x1, x2, go, 4, 1, 4, 1, 1, 1)
    assert _lib.last_kernel(1) == "corr_bwd_d4_strip_w256", _lib.last_kernel(1)
    torch.ops.cerberus.correlation_backward(x1[:1], x2[:1], go[:1], 4, 1, 4, 1, 1, 1)
    assert not _lib.last_kernel(1).startswith("corr_bwd_d4_strip"), _lib.last_kernel(1)
    # the 128-wide level of the benched pyramid at 4 pairs (2 image rows per wavefront); the 64-wide one (4 rows)
    # once it has more workgroups than the coarse-level kernel likes (16 pairs)
    for (B, C, H, W), want in (((4, 64, 64, 128), "corr_bwd_d4_strip_w128"), ((16, 128, 32, 64), "corr_bwd_d4_strip_w64")):
        a, b = dev(hash_uniform((B, C, H, W), 1)), dev(hash_uniform((B, C, H, W), 2))
        g = dev(hash_uniform((B, 81, H, W), 3))
        torch.ops.cerberus.correlation_backward(a, b, g, 4, 1, 4, 1, 1, 1)
        assert _lib.last_kernel(1) == want, _lib.last_kernel(1)
        torch.ops.cerberus.correlation_backward(a[:1], b[:1], g[:1], 4, 1, 4, 1, 1, 1)
        assert not _lib.last_kernel(1).startswith("corr_bwd_d4_strip"), _lib.last_kernel(1)


@pytest.mark.parametrize("W", [256, 128, 64])
def test_strip_backward_keeps_nonfinite_values_local(W):
    """The zeros a DPP shift fills in at a row's ends are the reference's zero padding, rows
    outside the image are read as zeros through the buffer bounds, and the shifted gradOutput
    rows of the second gradient get their out-of-row taps zeroed in LDS: NaN / Inf in gradOutput
    or in the feature maps reach exactly the elements the reference's sums touch (which include
    0 x Inf = NaN against the zero padding in the first gradient, correlation_cuda_kernel.cu:150-165)."""
    H = 16 if W == 64 else 12         # a 64-wide workgroup owns 8 rows
    shape = (1, 32, H, W)
    x1, x2 = hash_uniform(shape, 411), hash_uniform(shape, 412)
    go = hash_uniform((1, 81, H, W), 413)
    go[0, 40, 5, 0] = np.nan          # centre displacement, left border pixel
    go[0, 3, 0, W - 1] = np.inf       # top-right corner, tap above the image
    go[0, 77, H - 1, W - 3] = -np.inf  # bottom row, near the right end
    go[0, 9, 6, W // 2] = np.nan
    x1[0, 3, 4, W - 1] = np.inf       # last pixel of a row: the next row's first strip must not see it
    x2[0, 17, 7, 0] = -np.inf
    x2[0, 5, 0, W // 3] = np.nan
    r1, r2 = oracle.corr_backward_ref(x1, x2, go, 4, 1, 4, 1, 1)
    _lib.set_option("corr_bwd_variant", 12)
    try:
        g1, g2 = run_bwd(x1, x2, go, (4, 1, 4, 1, 1))
        assert _lib.last_kernel(1).startswith("corr_bwd_d4_strip"), _lib.last_kernel(1)
    finally:
        _lib.set_option("corr_bwd_variant", 0)
    assert np.array_equal(np.isnan(g1), np.isnan(r1)) and np.array_equal(np.isinf(g1), np.isinf(r1))
    assert np.array_equal(np.isnan(g2), np.isnan(r2)) and np.array_equal(np.isinf(g2), np.isinf(r2))
    ok1, ok2 = np.isfinite(r1), np.isfinite(r2)
    assert ok1.sum() > ok1.size // 2 and ok2.sum() > ok2.size // 2
    assert rel_err(np.where(ok1, g1, 0), np.where(ok1, r1, 0)) < TOL
    assert rel_err(np.where(ok2, g2, 0), np.where(ok2, r2, 0)) < TOL


# ---- round 3: the coarse-level kernels (corr_coarse.hip) -------------------------------------
COARSE_SHAPES = [(2, 128, 7, 32), (1, 64, 5, 64), (3, 128, 3, 64), (2, 128, 8, 16), (1, 256, 20, 32), (4, 64, 33, 64),
                 (1, 128, 1, 32)]


@pytest.mark.parametrize("shape", COARSE_SHAPES)
def test_coarse_level_kernels_against_the_oracle(shape):
    """corr_fwd_d4_coarse_kernel / corr_bwd_d4_coarse_kernel (W = 16 / 32 / 64: channel groups interleaved in
    the 16-lane DPP rows, lane-swap reduction, gradOutput rows by wave-private LDS-DMA with the second
    gradient's shift on the global side): forced on maps shorter and taller than the 9-row window, forward
    with the fused LeakyReLU into a wider buffer, against the C oracle; two launches are bit-identical."""
    B, C, H, W = shape
    x1, x2 = hash_uniform(shape, 421), hash_uniform(shape, 422)
    go = hash_uniform((B, 81, H, W), 423)
    ref = oracle.corr_forward_ref(x1, x2, 4, 1, 4, 1, 1)
    r1, r2 = oracle.corr_backward_ref(x1, x2, go, 4, 1, 4, 1, 1)
    _lib.set_option("corr_fwd_variant", 15)
    _lib.set_option("corr_bwd_variant", 14)
    try:
        out = run_fwd(x1, x2, (4, 1, 4, 1, 1))
        fname = _lib.last_kernel(0)
        buf = torch.full((B, 90, H, W), 7.5, device=DEV)
        torch.ops.cerberus.correlation_leaky_into(buf, dev(x1), dev(x2), 5, 4, 1, 4, 1, 1, 1, 0.1)
        assert _lib.last_kernel(0) == fname
        g1, g2 = run_bwd(x1, x2, go, (4, 1, 4, 1, 1))
        bname = _lib.last_kernel(1)
        g1b, g2b = run_bwd(x1, x2, go, (4, 1, 4, 1, 1))
        outb = run_fwd(x1, x2, (4, 1, 4, 1, 1))
    finally:
        _lib.set_option("corr_fwd_variant", 0)
        _lib.set_option("corr_bwd_variant", 0)
    assert fname == "corr_fwd_d4_coarse_%d" % W and bname == "corr_bwd_d4_coarse_%d" % W, (fname, bname)
    assert rel_err(out, ref) < TOL
    got = buf.cpu().numpy()
    assert rel_err(got[:, 5:86], np.where(ref > 0, ref, ref * np.float32(0.1))) < TOL
    assert np.all(np.delete(got, np.s_[5:86], axis=1) == 7.5)
    assert rel_err(g1, r1) < TOL and rel_err(g2, r2) < TOL
    assert np.array_equal(g1, g1b) and np.array_equal(g2, g2b) and np.array_equal(out, outb)


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 2e-3), (torch.bfloat16, 1.6e-2)])
@pytest.mark.parametrize("shape", [(2, 128, 7, 32), (1, 128, 12, 64), (2, 128, 8, 16), (4, 256, 32, 64)])
def test_coarse_level_forward_with_16bit_storage_against_the_oracle(dtype, tol, shape):
    """fp16 / bf16 storage through the coarse-level forward (loads widened on use, fp32 arithmetic, one rounding on
    store), fused LeakyReLU into a wider 16-bit buffer: against the fp64 oracle on the rounded inputs, as the default
    dispatch picks it (config 5's coarsest level is the last shape; since round 5 the matrix-core forward takes
    C <= 128) and forced."""
    B, C, H, W = shape
    x1 = torch.from_numpy(hash_uniform(shape, 441)).to(dtype)
    x2 = torch.from_numpy(hash_uniform(shape, 442)).to(dtype)
    p = (4, 1, 4, 1, 1)
    ref = oracle.corr_forward_ref(x1.double().numpy(), x2.double().numpy(), *p)
    out = torch.ops.cerberus.correlation(x1.to(DEV), x2.to(DEV), *p, 1)
    assert _lib.last_kernel(0) == ("corr_fwd_d4_mfma_4x16" if C <= 128 else "corr_fwd_d4_coarse_%d" % W), _lib.last_kernel(0)
    assert out.dtype == dtype
    assert rel_err(out.double().cpu().numpy(), ref) < tol
    buf = torch.full((B, 90, H, W), 7.5, device=DEV, dtype=dtype)
    _lib.set_option("corr_fwd_variant", 15)
    try:
        torch.ops.cerberus.correlation_leaky_into(buf, x1.to(DEV), x2.to(DEV), 5, *p, 1, 0.1)
        assert _lib.last_kernel(0) == "corr_fwd_d4_coarse_%d" % W, _lib.last_kernel(0)
    finally:
        _lib.set_option("corr_fwd_variant", 0)
    got = buf.double().cpu().numpy()
    assert rel_err(got[:, 5:86], np.where(ref > 0, ref, ref * 0.1)) < tol
    assert np.all(np.delete(got, np.s_[5:86], axis=1) == 7.5)
    # the vector kernels it replaces agree to the output rounding
    _lib.set_option("corr_fwd_variant", 16)
    try:
        old = torch.ops.cerberus.correlation(x1.to(DEV), x2.to(DEV), *p, 1)
        assert "coarse" not in _lib.last_kernel(0)
    finally:
        _lib.set_option("corr_fwd_variant", 0)
    assert rel_err(out.double().cpu().numpy(), old.double().cpu().numpy()) < tol


@pytest.mark.parametrize("dtype,tol", [(torch.float32, TOL), (torch.float16, 2e-3), (torch.bfloat16, 1.6e-2)])
@pytest.mark.parametrize("shape", [(2, 128, 7, 28), (1, 256, 5, 12), (1, 128, 9, 56), (2, 64, 6, 44), (1, 128, 3, 20),
                                   (1, 256, 4, 60), (4, 256, 14, 28), (1, 128, 2, 4), (2, 128, 5, 36),
                                   (4, 256, 11, 38), (1, 256, 5, 62), (2, 128, 6, 6), (1, 128, 13, 34), (2, 256, 3, 18), (1, 128, 4, 10)])
def test_coarse_forward_on_ragged_widths_against_the_oracle(shape, dtype, tol):
    """Round 6: the coarse-level forward on any EVEN width up to 64 -- the strips past the row's end load
    nothing (their zeros are the padding the last strip's neighbour shift reads) and store nothing; a width that is 2 mod 4
    (the 38-wide coarsest level of 1216 x 352 frames) ends in HALF a strip whose load brings the next row's first two
    pixels along: zeroed where the strip is the shifted operand, stored as two 8-byte halves --, fp32 and 16-bit storage,
    incl. an Inf at a row's last pixel (and, behind it, at the next row's first) and the LeakyReLU epilogue."""
    B, C, H, W = shape
    x1, x2 = hash_uniform(shape, 431), hash_uniform(shape, 432)
    if dtype == torch.float32:
        x2[0, 0, H // 2, W - 1] = np.inf
        if H > 3:
            x2[0, 1, 1, 0] = np.nan         # the first pixel of a row: the half strip in front of it must not see it
    a, b = dev(x1).to(dtype), dev(x2).to(dtype)
    ref = oracle.corr_forward_ref(a.float().cpu().numpy(), b.float().cpu().numpy(), 4, 1, 4, 1, 1)
    _lib.set_option("corr_fwd_variant", 15)
    try:
        out = torch.ops.cerberus.correlation(a, b, 4, 1, 4, 1, 1, 1)
        name = _lib.last_kernel(0)
        lk = torch.ops.cerberus.correlation_leaky(a, b, 4, 1, 4, 1, 1, 1, 0.1)
    finally:
        _lib.set_option("corr_fwd_variant", 0)
    assert name.startswith("corr_fwd_d4_coarse_rag" if W % 4 == 0 else "corr_fwd_d4_coarse_half"), name
    o = out.float().cpu().numpy()
    assert np.array_equal(np.isfinite(o), np.isfinite(ref))
    ok = np.isfinite(ref)
    assert rel_err(np.where(ok, o, 0), np.where(ok, ref, 0)) < tol
    lref = np.where(ref > 0, ref, ref * np.float32(0.1))
    assert rel_err(np.where(ok, lk.float().cpu().numpy(), 0), np.where(ok, lref, 0)) < tol


@pytest.mark.parametrize("shape", [(2, 128, 7, 28), (1, 256, 5, 12), (1, 128, 9, 56), (2, 64, 6, 44), (1, 128, 3, 20),
                                   (1, 256, 4, 60), (4, 256, 14, 28), (1, 128, 2, 4), (2, 128, 5, 36), (4, 128, 28, 56),
                                   (4, 256, 11, 38), (1, 256, 5, 62), (2, 128, 6, 6), (1, 128, 13, 34), (2, 256, 3, 18), (1, 128, 4, 10)])
def test_coarse_backward_on_ragged_widths_against_the_oracle(shape):
    """The coarse-level backward on the same widths: dead strips neither load nor store, the second gradient's shifted
    gradOutput rows are zeroed at the TRUE row end (also when that lies inside the half strip of a width that is 2 mod 4);
    NaN / Inf stay where the reference's sums put them."""
    B, C, H, W = shape
    x1, x2 = hash_uniform(shape, 441), hash_uniform(shape, 442)
    go = hash_uniform((B, 81, H, W), 443)
    x1[0, 1, H // 2, W - 1] = np.inf
    go[0, 44, H - 1, W - 1] = np.nan
    go[0, 8, 0, 0] = -np.inf
    if H > 3:
        x2[0, 2, 2, 0] = np.inf             # the first pixel of a row, behind the previous row's half strip
    r1, r2 = oracle.corr_backward_ref(x1, x2, go, 4, 1, 4, 1, 1)
    _lib.set_option("corr_bwd_variant", 14)
    try:
        g1, g2 = run_bwd(x1, x2, go, (4, 1, 4, 1, 1))
        name = _lib.last_kernel(1)
    finally:
        _lib.set_option("corr_bwd_variant", 0)
    assert name.startswith("corr_bwd_d4_coarse_rag" if W % 4 == 0 else "corr_bwd_d4_coarse_half"), name
    assert np.array_equal(np.isnan(g1), np.isnan(r1)) and np.array_equal(np.isinf(g1), np.isinf(r1))
    assert np.array_equal(np.isnan(g2), np.isnan(r2)) and np.array_equal(np.isinf(g2), np.isinf(r2))
    ok1, ok2 = np.isfinite(r1), np.isfinite(r2)
    assert rel_err(np.where(ok1, g1, 0), np.where(ok1, r1, 0)) < TOL
    assert rel_err(np.where(ok2, g2, 0), np.where(ok2, r2, 0)) < TOL


def test_coarse_level_kernels_are_the_default_on_the_coarse_levels_of_the_benched_pyramid():
    for (B, C, H, W), fw, bw in (((4, 256, 16, 32), "corr_fwd_d4_coarse_32", "corr_bwd_d4_coarse_32"),
                                 ((4, 128, 32, 64), "corr_fwd_d4_coarse_64", "corr_bwd_d4_coarse_64"),
                                 ((1, 128, 32, 64), "corr_fwd_d4_coarse_64", "corr_bwd_d4_coarse_64")):
        a, b = dev(hash_uniform((B, C, H, W), 1)), dev(hash_uniform((B, C, H, W), 2))
        g = dev(hash_uniform((B, 81, H, W), 3))
        torch.ops.cerberus.correlation(a, b, 4, 1, 4, 1, 1, 1)
        torch.ops.cerberus.correlation_backward(a, b, g, 4, 1, 4, 1, 1, 1)
        assert (_lib.last_kernel(0), _lib.last_kernel(1)) == (fw, bw), (_lib.last_kernel(0), _lib.last_kernel(1))
    # many pairs per call: the tile / strip kernels again
    a, b = dev(hash_uniform((16, 128, 32, 64), 1)), dev(hash_uniform((16, 128, 32, 64), 2))
    torch.ops.cerberus.correlation(a, b, 4, 1, 4, 1, 1, 1)
    assert "coarse" not in _lib.last_kernel(0), _lib.last_kernel(0)
    # channel counts the lane layout does not divide: the tile kernels, silently
    a, b = dev(hash_uniform((1, 24, 8, 32), 1)), dev(hash_uniform((1, 24, 8, 32), 2))
    torch.ops.cerberus.correlation(a, b, 4, 1, 4, 1, 1, 1)
    assert "coarse" not in _lib.last_kernel(0), _lib.last_kernel(0)


@pytest.mark.parametrize("W", [32, 64])
def test_coarse_level_kernels_keep_nonfinite_values_local(W):
    """Zero padding by DPP fill (columns), by a skipped row with the reference's x * 0 (rows) and by the LDS patch
    of the shifted gradOutput rows: NaN / Inf reach exactly the elements the reference's sums touch, forward and
    both gradients (0 x Inf = NaN against the padding included, correlation_cuda_kernel.cu:60-75, 150-165)."""
    H = 12
    shape = (1, 128, H, W)
    x1, x2 = hash_uniform(shape, 431), hash_uniform(shape, 432)
    go = hash_uniform((1, 81, H, W), 433)
    go[0, 40, 5, 0] = np.nan
    go[0, 3, 0, W - 1] = np.inf        # top-right corner, tap above the image
    go[0, 77, H - 1, W - 3] = -np.inf  # bottom row
    go[0, 9, 6, W // 2] = np.nan
    x1[0, 3, 4, W - 1] = np.inf        # last pixel of a row: the next row's first strip must not see it
    x1[0, 100, 0, 2] = np.nan          # top row: displacement rows above the image
    x2[0, 17, 7, 0] = -np.inf
    x2[0, 5, 0, W // 3] = np.nan
    ref = oracle.corr_forward_ref(x1, x2, 4, 1, 4, 1, 1)
    r1, r2 = oracle.corr_backward_ref(x1, x2, go, 4, 1, 4, 1, 1)
    _lib.set_option("corr_fwd_variant", 15)
    _lib.set_option("corr_bwd_variant", 14)
    try:
        out = run_fwd(x1, x2, (4, 1, 4, 1, 1))
        g1, g2 = run_bwd(x1, x2, go, (4, 1, 4, 1, 1))
        assert "coarse" in _lib.last_kernel(0) and "coarse" in _lib.last_kernel(1)
    finally:
        _lib.set_option("corr_fwd_variant", 0)
        _lib.set_option("corr_bwd_variant", 0)
    for got, want in ((out, ref), (g1, r1), (g2, r2)):
        assert np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(np.isinf(got), np.isinf(want))
        ok = np.isfinite(want)
        assert ok.sum() > ok.size // 2
        assert rel_err(np.where(ok, got, 0), np.where(ok, want, 0)) < TOL


def test_seeded_random_shape_sweep_coarse_level_kernels_vs_generic():
    """40 seeded random coarse-level shapes (W = 16 / 32 / 64, many channels, 1..40 rows, 1..5 pairs) through the DEFAULT
    dispatch against the generic kernels (which the oracle tests pin): forward and both gradients; the coarse kernels
    must have been among the kernels chosen."""
    rng = np.random.RandomState(20261003)
    kernels = set()
    for trial in range(40):
        B = int(rng.randint(1, 6))
        W = int(rng.choice([16, 32, 64]))
        C = int(rng.choice([64, 128, 192, 256, 320, 96, 48]))
        H = int(rng.randint(1, 41))
        shape = (B, C, H, W)
        x1, x2 = dev(hash_uniform(shape, 3900 + trial)), dev(hash_uniform(shape, 4900 + trial))
        go = dev(hash_uniform((B, 81, H, W), 5900 + trial))
        p = (4, 1, 4, 1, 1, 1)
        out = torch.ops.cerberus.correlation(x1, x2, *p)
        kernels.add(_lib.last_kernel(0))
        g1, g2 = torch.ops.cerberus.correlation_backward(x1, x2, go, *p)
        kernels.add(_lib.last_kernel(1))
        _lib.set_option("corr_force_generic", 1)
        try:
            ref = torch.ops.cerberus.correlation(x1, x2, *p)
            r1, r2 = torch.ops.cerberus.correlation_backward(x1, x2, go, *p)
        finally:
            _lib.set_option("corr_force_generic", 0)
        for a, b, what in ((out, ref, "out"), (g1, r1, "g1"), (g2, r2, "g2")):
            scale = float(b.abs().max()) or 1.0
            assert float((a - b).abs().max()) / scale < TOL, (shape, what, sorted(kernels))
    for want in ("corr_fwd_d4_coarse_16", "corr_fwd_d4_coarse_32", "corr_fwd_d4_coarse_64",
                 "corr_bwd_d4_coarse_16", "corr_bwd_d4_coarse_32", "corr_bwd_d4_coarse_64"):
        assert want in kernels, (want, sorted(kernels))


# ---- round 3: the benched configurations against the oracle at FULL size ---------------------
@pytest.mark.parametrize("lvl", [0, 1, 2, 3])
def test_config3_batch4_levels_against_the_oracle_at_full_size(lvl):
    """The benched workload itself (BASELINE configs 3 / 4: W32 levels of 1024x512, 4 pairs per
    GPU, fp32): forward and both gradients of the DEFAULT kernels, full tensors, against the C
    restatement of the reference's CUDA kernels (seconds per level on the host)."""
    C, H, W = W32_PYRAMID_1024x512[lvl]
    shp = (4, C, H, W)
    x1, x2, go = hash_uniform(shp, 31), hash_uniform(shp, 32), hash_uniform((4, 81, H, W), 33)
    p = (4, 1, 4, 1, 1)
    out = run_fwd(x1, x2, p)
    g1, g2 = run_bwd(x1, x2, go, p)
    assert _lib.last_kernel(0).startswith("corr_fwd_d4") and _lib.last_kernel(1).startswith("corr_bwd_d4")
    assert rel_err(out, oracle.corr_forward_ref(x1, x2, *p)) < TOL
    r1, r2 = oracle.corr_backward_ref(x1, x2, go, *p)
    assert rel_err(g1, r1) < TOL
    assert rel_err(g2, r2) < TOL


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 2e-3), (torch.bfloat16, 1.6e-2)])
@pytest.mark.parametrize("lvl", [0, 1, 2, 3])
def test_config5_16bit_levels_against_the_oracle_at_full_size(lvl, dtype, tol):
    """BASELINE config 5 (AMP, 2048x1024 -> W32 levels) at full level size: the 16-bit kernels
    (matrix cores where the dispatcher takes them) against the fp64 oracle run on the rounded
    inputs, full tensors; the tolerance is the output rounding of the storage type."""
    C, H, W = CONFIG5_PYRAMID_2048x1024[lvl]
    shp = (1, C, H, W)
    x1 = torch.from_numpy(hash_uniform(shp, 61)).to(dtype)
    x2 = torch.from_numpy(hash_uniform(shp, 62)).to(dtype)
    go = torch.from_numpy(hash_uniform((1, 81, H, W), 63)).to(dtype)
    p = (4, 1, 4, 1, 1)
    out = torch.ops.cerberus.correlation(x1.to(DEV), x2.to(DEV), *p, 1)
    fwd_name = _lib.last_kernel(0)
    g1, g2 = torch.ops.cerberus.correlation_backward(x1.to(DEV), x2.to(DEV), go.to(DEV), *p, 1)
    bwd_name = _lib.last_kernel(1)
    assert "mfma" in bwd_name, bwd_name
    if 16 < C <= 128:
        assert "mfma" in fwd_name, fwd_name
    a1, a2, ag = x1.double().numpy(), x2.double().numpy(), go.double().numpy()
    assert rel_err(out.double().cpu().numpy(), oracle.corr_forward_ref(a1, a2, *p)) < tol
    r1, r2 = oracle.corr_backward_ref(a1, a2, ag, *p)
    assert rel_err(g1.double().cpu().numpy(), r1) < tol
    assert rel_err(g2.double().cpu().numpy(), r2) < tol


def _oracle_per_item(fn, *arrays):
    """The single-threaded C oracle on every batch item in its own host thread (ctypes releases
    the GIL; batch items never interact, correlation_cuda_kernel.cu:35,105,181)."""
    from concurrent.futures import ThreadPoolExecutor
    B = arrays[0].shape[0]
    with ThreadPoolExecutor(max_workers=B) as ex:
        parts = list(ex.map(lambda b: fn(*[a[b:b + 1] for a in arrays]), range(B)))
    if isinstance(parts[0], tuple):
        return tuple(np.concatenate([p[i] for p in parts], 0) for i in range(len(parts[0])))
    return np.concatenate(parts, 0)


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 2e-3), (torch.bfloat16, 1.6e-2)])
@pytest.mark.parametrize("lvl", [0, 1, 2, 3])
def test_config5_16bit_levels_at_the_benched_batch_against_the_oracle(lvl, dtype, tol):
    """the dispatcher switches kernels on the batch (coarse / strip workgroup-count
    thresholds), so the kernels `bench.py --dtype f16 --width 2048 --height 1024` times at 4 pairs are
    not all the ones the B = 1 test above runs.  Here: the benched call itself -- 4 pairs, default
    dispatch, full tensors -- against the fp64 oracle on the rounded inputs, every item; and the kernel
    the dispatcher took must be the one bench.py's Workload dispatches for the same level (same shapes,
    same dtype, same options: read back through cerberus_last_kernel from both)."""
    C, H, W = CONFIG5_PYRAMID_2048x1024[lvl]
    shp = (4, C, H, W)
    x1 = torch.from_numpy(hash_uniform(shp, 71)).to(dtype)
    x2 = torch.from_numpy(hash_uniform(shp, 72)).to(dtype)
    go = torch.from_numpy(hash_uniform((4, 81, H, W), 73)).to(dtype)
    p = (4, 1, 4, 1, 1)
    out = torch.ops.cerberus.correlation(x1.to(DEV), x2.to(DEV), *p, 1)
    fwd_name = _lib.last_kernel(0)
    g1, g2 = torch.ops.cerberus.correlation_backward(x1.to(DEV), x2.to(DEV), go.to(DEV), *p, 1)
    bwd_name = _lib.last_kernel(1)
    # what the benchmark's own workload object dispatches at this level (pairs = 4)
    import bench
    wl = bench.Workload(4, 2048, 1024, torch.device(DEV), "smooth", False, 1, dtype)
    assert tuple(wl.levels[lvl]) == (C, H, W)
    t = wl.dirs[0][lvl]
    torch.ops.cerberus.correlation(t["f1"], t["f2"], *bench.CORR_P)
    assert _lib.last_kernel(0) == fwd_name, (_lib.last_kernel(0), fwd_name)
    torch.ops.cerberus.correlation_backward(t["f1"], t["f2"], t["gout"], *bench.CORR_P)
    assert _lib.last_kernel(1) == bwd_name, (_lib.last_kernel(1), bwd_name)
    del wl, t
    a1, a2, ag = x1.double().numpy(), x2.double().numpy(), go.double().numpy()
    ref = _oracle_per_item(lambda a, b: oracle.corr_forward_ref(a, b, *p), a1, a2)
    r1, r2 = _oracle_per_item(lambda a, b, g: oracle.corr_backward_ref(a, b, g, *p), a1, a2, ag)
    o, h1, h2 = (v.double().cpu().numpy() for v in (out, g1, g2))
    for b in range(4):
        assert rel_err(o[b], ref[b]) < tol, (b, fwd_name)
        assert rel_err(h1[b], r1[b]) < tol, (b, bwd_name)
        assert rel_err(h2[b], r2[b]) < tol, (b, bwd_name)


def test_concat_buffer_written_in_place_before_backward_raises():
    """CostVolumeConcat saves the concatenation buffer for backward like any output; an
    in-place write to it (which would flip LeakyReLU derivative signs silently) trips autograd's
    version check, and the correlation backward is skipped when neither input needs a gradient."""
    from cerberusnet_amd.correlation_package.correlation import cost_volume_concat
    shp = (1, 8, 10, 20)
    a = dev(hash_uniform(shp, 1)).requires_grad_(True)
    b = dev(hash_uniform(shp, 2)).requires_grad_(True)
    other = dev(hash_uniform((1, 3, 10, 20), 3)).requires_grad_(True)
    hyper = (4, 1, 4, 1, 1, 1)
    buf = cost_volume_concat(a, b, [other], hyper, 0.1)
    buf.mul_(2.0)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        buf.sum().backward()
    buf = cost_volume_concat(a.detach(), b.detach(), [other], hyper, 0.1)
    before = _lib.last_kernel(1)
    _lib.set_option("corr_bwd_variant", 0)
    buf.sum().backward()                       # only `other` needs a gradient
    assert torch.equal(other.grad, torch.ones_like(other))
    assert _lib.last_kernel(1) == before       # no correlation backward was launched


# ---- round 6: f2, the warp fused into the correlation forward (warp_corr.hip) ---------------------------------------
@pytest.mark.parametrize("dtype,tol", [(torch.float32, TOL), (torch.float16, 2e-3), (torch.bfloat16, 1.6e-2)])
@pytest.mark.parametrize("shape,amp", [((2, 32, 16, 24), 3.0), ((1, 12, 9, 37), 0.7), ((2, 7, 20, 70), 9.0), ((1, 64, 33, 64), 40.0),
                                       ((1, 3, 5, 4), 2.0)])
@pytest.mark.parametrize("pad", ["border", "zeros"])
def test_fused_warp_correlation_equals_the_two_stand_alone_ops(shape, amp, pad, dtype, tol):
    """cerberus::warp_correlation_leaky against the oracle's composition (flow_warp_ref -> corr_forward_ref ->
    LeakyReLU) and against this package's own two kernels; ragged tiles, flows from sub-pixel to larger than the image, both
    padding modes, an fp32 flow beside 16-bit features."""
    B, C, H, W = shape
    x1, x2 = hash_uniform(shape, 461), hash_uniform(shape, 462)
    flo = hash_uniform((B, 2, H, W), 463, -amp, amp)
    a, b, f = dev(x1).to(dtype), dev(x2).to(dtype), dev(flo)
    pm = {"zeros": 0, "border": 1}[pad]
    fused = torch.ops.cerberus.warp_correlation_leaky(a, b, f, pm, 0.1)
    assert _lib.last_kernel(0) == "warp_corr_fwd_8x32"
    warped = torch.ops.cerberus.flow_warp(b, f, pm, 0)
    two = torch.ops.cerberus.correlation_leaky(a, warped, 4, 1, 4, 1, 1, 1, 0.1)
    assert rel_err(fused.float().cpu().numpy(), two.float().cpu().numpy()) < tol
    if dtype == torch.float32:
        w_ref = oracle.flow_warp_ref(torch.from_numpy(x2), torch.from_numpy(flo), pad).numpy()
        ref = oracle.corr_forward_ref(x1, w_ref, 4, 1, 4, 1, 1)
        ref = np.where(ref > 0, ref, ref * np.float32(0.1))
        assert rel_err(fused.cpu().numpy(), ref) < TOL


def test_fused_warp_correlation_backward_recomputes_the_warp():
    """The training path of f2 saves neither the warped features nor the warp context: its backward recomputes the warp
    and runs the tuned backward kernels -- gradients w.r.t. both feature maps and the flow equal those of the unfused chain
    bit for bit (the same kernels on the same values)."""
    shape = (2, 32, 24, 64)
    x1 = dev(hash_uniform(shape, 471)).requires_grad_(True)
    x2 = dev(hash_uniform(shape, 472)).requires_grad_(True)
    fl = dev(hash_uniform((2, 2, 24, 64), 473, -3.0, 3.0)).requires_grad_(True)
    go = dev(hash_uniform((2, 81, 24, 64), 474))
    out = torch.ops.cerberus.warp_correlation_leaky(x1, x2, fl, 1, 0.1)
    g1, g2, gf = torch.autograd.grad(out, (x1, x2, fl), go)
    from cerberusnet_amd.loss_functions.UnFlowLoss import flow_warp
    ref = torch.ops.cerberus.correlation_leaky(x1, flow_warp(x2, fl), 4, 1, 4, 1, 1, 1, 0.1)
    r1, r2, rf = torch.autograd.grad(ref, (x1, x2, fl), go)
    assert rel_err(out.detach().cpu().numpy(), ref.detach().cpu().numpy()) < TOL
    # (the LeakyReLU mask comes from the stored result's sign: identical wherever the two forwards agree in sign, i.e.
    # everywhere but at values within rounding of zero)
    for a, b in ((g1, r1), (g2, r2), (gf, rf)):
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 1e-4


@pytest.mark.parametrize("shape", [(4, 128, 32, 64), (1, 256, 16, 32), (2, 64, 24, 40)])
def test_fused_warp_correlation_on_small_maps_splits_the_channel_sum(shape):
    """Fewer than 256 tiles: several workgroups per tile sum channel slices into an fp32 volume (float atomics) and a second
    launch finishes it; with no workspace the same call takes the one-launch form.  Both against the two stand-alone ops."""
    import ctypes
    B, C, H, W = shape
    a, b = dev(hash_uniform(shape, 481)), dev(hash_uniform(shape, 482))
    f = dev(hash_uniform((B, 2, H, W), 483, -2.0, 2.0))
    lib = _lib.get()
    assert lib.cerberus_warp_correlation_workspace_bytes(B, C, H, W) == B * 81 * H * W * 4
    assert lib.cerberus_warp_correlation_workspace_bytes(4, 32, 128, 256) == 0
    two = torch.ops.cerberus.correlation_leaky(a, torch.ops.cerberus.flow_warp(b, f, 1, 0), 4, 1, 4, 1, 1, 1, 0.1)
    fused = torch.ops.cerberus.warp_correlation_leaky(a, b, f, 1, 0.1)
    assert rel_err(fused.cpu().numpy(), two.cpu().numpy()) < TOL
    out = torch.empty_like(two)
    rc = lib.cerberus_warp_correlation_forward(a.data_ptr(), b.data_ptr(), f.data_ptr(), out.data_ptr(), None, 0, B, C, H, W, 1,
                                               ctypes.c_float(0.1), 0, 0, 0, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert rc == 0 and rel_err(out.cpu().numpy(), two.cpu().numpy()) < TOL
