"""Caller row on the GPU: PWCNetHead on the HIP ops vs goldens from the REFERENCE head and vs
the same head on the pure-torch backend (same MIOpen convolutions, so any difference is the
hot-path ops)."""
import numpy as np
import pytest
import torch

from cerberusnet_amd.nnet_models import PWCNetHead
from cerberusnet_amd.synth import fill_parameters
from conftest import rel_err, l2_err
from test_pwchead_cpu import CHANS, build, pyramids

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("est,tag", [("FlowEstimatorLite", "lite"), ("FlowEstimatorDense", "dense")])
@pytest.mark.parametrize("fuse", ["concat", "leaky", "none"])
def test_hip_head_matches_reference_goldens(golden, est, tag, fuse):
    """fuse: 'concat' = the correlation kernel writes LeakyReLU(corr) straight into the
    estimator's concatenation buffer (the default), 'leaky' = fused LeakyReLU
    + torch.cat, 'none' = the reference's three passes.  All against the REFERENCE head."""
    g = golden("pwchead_" + tag)
    head = build(est, fuse_leaky=fuse != "none", fuse_concat=fuse == "concat",
                 fuse_upsample=fuse != "none").to(DEV)
    p1, p2 = pyramids(g, DEV)
    flows = head((None, p1), (None, p2))
    for i, f in enumerate(flows):
        # GPU convolutions (MIOpen) differ from the CPU ones in summation order: 1e-4
        assert rel_err(f.detach().cpu().numpy(), g["flow_%d" % i]) < 1e-4
    loss = sum((f * f).mean() for f in flows)
    assert abs(loss.item() - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    grads = torch.autograd.grad(loss, p1 + p2 + list(head.parameters()))
    for l in range(4):
        assert l2_err(grads[l].cpu().numpy(), g["g_im1_%d" % l]) < 2e-3
        assert l2_err(grads[4 + l].cpu().numpy(), g["g_im2_%d" % l]) < 2e-3
    norms = np.array([float(x.double().norm()) for x in grads[8:]])
    assert np.allclose(norms, g["param_grad_norms"], rtol=2e-3, atol=1e-9)


@pytest.mark.parametrize("est,tag", [("FlowEstimatorLite", "lite")])
def test_hip_head_with_the_fused_warp_correlation_matches_reference_goldens(golden, est, tag):
    """f2 inside the head (fuse_warp=True: warp + correlation + LeakyReLU as one forward kernel per level, the backward
    recomputing the warp) against the REFERENCE head's goldens: flows, loss, input and parameter gradients."""
    g = golden("pwchead_" + tag)
    head = build(est, fuse_warp=True).to(DEV)
    p1, p2 = pyramids(g, DEV)
    flows = head((None, p1), (None, p2))
    for i, f in enumerate(flows):
        assert rel_err(f.detach().cpu().numpy(), g["flow_%d" % i]) < 1e-4
    loss = sum((f * f).mean() for f in flows)
    assert abs(loss.item() - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    grads = torch.autograd.grad(loss, p1 + p2 + list(head.parameters()))
    for l in range(4):
        assert l2_err(grads[l].cpu().numpy(), g["g_im1_%d" % l]) < 2e-3
        assert l2_err(grads[4 + l].cpu().numpy(), g["g_im2_%d" % l]) < 2e-3
    norms = np.array([float(x.double().norm()) for x in grads[8:]])
    assert np.allclose(norms, g["param_grad_norms"], rtol=2e-3, atol=1e-9)


def test_hip_and_torch_backends_agree_on_device(golden):
    g = golden("pwchead_lite")
    res = {}
    for backend in ("hip", "torch"):
        head = build("FlowEstimatorLite", correlation_backend=backend).to(DEV)
        p1, p2 = pyramids(g, DEV)
        flows = head((None, p1), (None, p2))
        loss = sum((f * f).mean() for f in flows)
        grads = torch.autograd.grad(loss, p1 + p2 + list(head.parameters()))
        res[backend] = ([f.detach().cpu().numpy() for f in flows],
                        [x.cpu().numpy() for x in grads])
    for a, b in zip(res["hip"][0], res["torch"][0]):
        assert rel_err(a, b) < 2e-5
    for a, b in zip(res["hip"][1], res["torch"][1]):
        assert l2_err(a, b) < 2e-3


def test_both_directions_training_step_runs_and_is_finite():
    head = build("FlowEstimatorLite").to(DEV)
    opt = torch.optim.Adam(head.parameters(), lr=1e-4, betas=(0.9, 0.99), weight_decay=1e-6)
    feats1 = [torch.randn(2, c, 8 * 2 ** l, 16 * 2 ** l, device=DEV) for l, c in enumerate(reversed(CHANS))]
    feats2 = [torch.randn(2, c, 8 * 2 ** l, 16 * 2 ** l, device=DEV) for l, c in enumerate(reversed(CHANS))]
    for _ in range(2):
        opt.zero_grad()
        fw = head((None, feats1), (None, feats2))
        bw = head((None, feats2), (None, feats1))
        loss = sum(f.abs().mean() for f in fw + bw)
        loss.backward()
        opt.step()
    assert torch.isfinite(loss)
    assert all(torch.isfinite(p.grad).all() for p in head.parameters())


def test_graphed_flow_step_equals_eager():
    """the whole bidirectional head step captured into one hipGraph must give
    the eager step's loss, flows and gradients, for the captured inputs and for new ones.

    The comparison is BIT-EXACT and runs with MIOpen's deterministic algorithms.  Measured on
    MI355X (tools/diag_graph_step.py, round 2): with the default (atomic) convolution
    algorithms two EAGER runs of the same step already differ -- flows by 2e-7 and, because a
    last-bit change of a flow value can flip a floor() in the warp's bilinear taps, the finest
    level's input gradient by up to 1.7e-3 in l2 (0.5 % of its elements) -- which is what made
    the round-1 form of this test (l2 < 1e-3, default algorithms) flaky.  With deterministic
    algorithms eager == eager == graph replay, to the last bit, for every tensor: every kernel
    of this package is bit-reproducible, so that is the honest bar.

    The loss reduces in two block-level stages (rows of 4096, then <= 512 values).  A single
    ``f.abs().mean()`` over the 524288 elements of a full-resolution flow is one of PyTorch's
    multi-block reductions (staging buffer + semaphores); replayed from a hipGraph with eager
    work interleaved between replays, THAT scalar came back wrong (off by a constant, often
    about half) while every flow and every gradient of the same replay stayed bit-exact --
    measured with tools/diag_graph_order.py ('blockloss' vs default), independent of this
    package's kernels.  GraphedFlowStep's docstring carries the caveat."""
    from cerberusnet_amd.graphs import GraphedFlowStep
    det = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    try:
        torch.manual_seed(3)
        head = build("FlowEstimatorLite").to(DEV)
        shapes = [(2, c, 8 * 2 ** l, 16 * 2 ** l) for l, c in enumerate(reversed(CHANS))]
        mk = lambda: [torch.randn(s, device=DEV) for s in shapes]
        loss_fn = lambda flows: sum(f.abs().reshape(-1, 4096).mean(1).mean() for f in flows)

        def eager(p1, p2):
            for p in head.parameters():
                p.grad = None
            a = [t.clone().requires_grad_(True) for t in p1]
            b = [t.clone().requires_grad_(True) for t in p2]
            fw = head((None, a), (None, b))
            bw = head((None, b), (None, a))
            loss = loss_fn(list(fw) + list(bw))
            loss.backward()
            return (loss.detach().clone(), [f.detach().clone() for f in fw],
                    [p.grad.detach().clone() for p in head.parameters()],
                    [t.grad.detach().clone() for t in a] + [t.grad.detach().clone() for t in b])

        p1, p2 = mk(), mk()
        step = GraphedFlowStep(head, loss_fn, p1, p2, input_grads=True)
        for trial in range(3):
            if trial:
                p1, p2 = mk(), mk()
            loss, fw, _ = step(p1, p2)
            g_loss, g_fw = loss.detach().clone(), [f.detach().clone() for f in fw]
            g_par = [p.grad.detach().clone() for p in head.parameters()]
            g_in = [g.detach().clone() for g in step.input_gradients()[0] + step.input_gradients()[1]]
            e_loss, e_fw, e_par, e_in = eager(p1, p2)
            assert torch.equal(g_loss, e_loss)
            for a, b in zip(g_fw + g_par + g_in, e_fw + e_par + e_in):
                assert torch.equal(a, b)
        with pytest.raises(RuntimeError, match="shape"):
            step([t[:1] for t in p1], p2)
    finally:
        torch.backends.cudnn.deterministic = det


def test_forward_both_equals_the_two_separate_calls(golden):
    """both flow directions stacked along the batch axis in one pass of the head
    (``forward_both``) against the two calls ``cerberus.py:131,135`` makes -- flows of both
    directions, and the parameter gradients of a loss over both -- and against the flows the
    REFERENCE head produced (golden fixture) for the 1 -> 2 direction."""
    det = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    try:
        g = golden("pwchead_lite")
        head = build("FlowEstimatorLite").to(DEV)
        p1, p2 = pyramids(g, DEV)
        for p in head.parameters():
            p.grad = None
        fw, bw = head.forward_both((None, p1), (None, p2))
        sum((f * f).mean() for f in fw + bw).backward()
        grads_both = [p.grad.clone() for p in head.parameters()]
        for p in head.parameters():
            p.grad = None
        fw2 = head((None, p1), (None, p2))
        bw2 = head((None, p2), (None, p1))
        sum((f * f).mean() for f in list(fw2) + list(bw2)).backward()
        for a, b in zip(fw + bw, list(fw2) + list(bw2)):
            assert a.shape == b.shape
            assert rel_err(a.detach().cpu().numpy(), b.detach().cpu().numpy()) < 1e-5
        for a, p in zip(grads_both, head.parameters()):
            assert l2_err(a.cpu().numpy(), p.grad.cpu().numpy()) < 2e-3
        for i, f in enumerate(fw):
            assert rel_err(f.detach().cpu().numpy(), g["flow_%d" % i]) < 1e-4
    finally:
        torch.backends.cudnn.deterministic = det


def _graphed_plain_mean_setup():
    from cerberusnet_amd.graphs import GraphedFlowStep
    torch.manual_seed(5)
    head = build("FlowEstimatorLite").to(DEV)
    shapes = [(2, c, 8 * 2 ** l, 16 * 2 ** l) for l, c in enumerate(reversed(CHANS))]
    mk = lambda: [torch.randn(s, device=DEV) for s in shapes]
    loss_fn = lambda flows: sum(f.abs().mean() for f in flows)      # the natural (UnFlow-style) loss
    p1, p2 = mk(), mk()
    return head, mk, loss_fn, GraphedFlowStep(head, loss_fn, p1, p2)


def _eager_loss(head, loss_fn, p1, p2):
    with torch.no_grad():
        fw = head((None, p1), (None, p2))
        bw = head((None, p2), (None, p1))
        return loss_fn(list(fw) + list(bw)).clone()


def test_graphed_flow_step_returns_the_right_loss_for_a_plain_mean():
    """with the natural loss -- whole-tensor means, PyTorch's multi-block
    reduction -- the scalar a replay computed came back wrong once eager work ran between
    replays.  GraphedFlowStep now evaluates the loss eagerly on the flows the replay wrote: the
    returned value must equal the eager step's for new inputs, with eager steps interleaved."""
    det = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    try:
        head, mk, loss_fn, step = _graphed_plain_mean_setup()
        for _ in range(4):
            p1, p2 = mk(), mk()
            loss, _, _ = step(p1, p2)
            got = loss.detach().clone()
            want = _eager_loss(head, loss_fn, p1, p2)     # also the interleaved eager work
            assert torch.equal(got, want), (float(got), float(want))
    finally:
        torch.backends.cudnn.deterministic = det


def test_graphed_flow_step_captured_loss_is_exact_with_block_level_reductions():
    """Round 4: the defect is ATen's multi-block reduction inside the replayed graph, not the capture
    (tools/diag_graph_loss.py).  With the loss built from block-level reductions (graph_safe_mean) the scalar
    the GRAPH computed is the eager step's, bit for bit up to the summation order of the two forms, for new
    inputs and with eager head passes interleaved -- and trust_captured_loss=True returns it without the
    eager re-evaluation."""
    from cerberusnet_amd.graphs import GraphedFlowStep, graph_safe_mean
    det = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    try:
        torch.manual_seed(5)
        head = build("FlowEstimatorLite").to(DEV)
        shapes = [(2, c, 8 * 2 ** l, 16 * 2 ** l) for l, c in enumerate(reversed(CHANS))]
        mk = lambda: [torch.randn(s, device=DEV) for s in shapes]
        loss_fn = lambda flows: sum(graph_safe_mean(f.abs()) for f in flows)
        step = GraphedFlowStep(head, loss_fn, mk(), mk(), trust_captured_loss=True)
        for _ in range(5):
            p1, p2 = mk(), mk()
            loss, _, _ = step(p1, p2)
            assert loss is step.captured_loss
            got = loss.detach().clone()
            want = _eager_loss(head, loss_fn, p1, p2)     # also the interleaved eager work
            assert torch.equal(got, want), (float(got), float(want))
            plain = _eager_loss(head, lambda flows: sum(f.abs().mean() for f in flows), p1, p2)
            assert abs(float(got) - float(plain)) <= 1e-5 * abs(float(plain))
    finally:
        torch.backends.cudnn.deterministic = det


@pytest.mark.xfail(strict=False, reason="the loss scalar computed INSIDE the replayed graph comes back wrong "
                   "after interleaved eager work -- also with the head on stock PyTorch ops only "
                   "(tools/diag_graph_order.py torch), so not this package's kernels; GraphedFlowStep does not return it")
def test_graphed_flow_step_captured_loss_scalar_known_defect():
    """Keeps the defect visible: `captured_loss` is the scalar the graph itself reduced."""
    det = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    try:
        head, mk, loss_fn, step = _graphed_plain_mean_setup()
        for _ in range(4):
            p1, p2 = mk(), mk()
            step(p1, p2)
            got = step.captured_loss.detach().clone()
            want = _eager_loss(head, loss_fn, p1, p2)
            assert torch.equal(got, want), (float(got), float(want))
    finally:
        torch.backends.cudnn.deterministic = det


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_head_under_autocast_runs_the_16bit_kernels(golden, dt):
    """AMP (BASELINE config 5): under torch.autocast the convolutions hand 16-bit features to the
    hot-path ops, which must run in that dtype (CorrelationFunction has custom_fwd without
    cast_inputs, correlation.py:30; the warp follows its image) and stay close to the fp32 run:
    flows within a few percent (16-bit feature storage), finite gradients for every input
    and parameter, 16-bit tiled warp backward (no global-atomic CAS loop)."""
    from cerberusnet_amd import _lib
    g = golden("pwchead_lite")
    head = build("FlowEstimatorLite").to(DEV)
    p1, p2 = pyramids(g, DEV)
    ref = [f.detach() for f in head((None, p1), (None, p2))]
    seen = []
    with torch.autocast("cuda", dtype=dt):
        h1 = [t.to(dt) for t in p1]
        h2 = [t.to(dt) for t in p2]
        for t in h1 + h2:
            t.retain_grad()
        flows = head((None, h1), (None, h2))
        seen.append(_lib.last_kernel(0))
        loss = sum((f.float() ** 2).mean() for f in flows)
    loss.backward()
    # the 16-bit correlation takes the register-staged tuned kernels (the LDS-DMA ones are fp32)
    assert "corr_fwd_d4" in seen[0] and "dma" not in seen[0], seen
    tol = 0.05 if dt == torch.float16 else 0.25
    for f, r in zip(flows, ref):
        assert rel_err(f.detach().float().cpu().numpy(), r.cpu().numpy()) < tol
    for t in h1 + h2:
        assert t.grad is not None and t.grad.dtype == dt and torch.isfinite(t.grad.float()).all()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in head.parameters())


def test_graphed_inference_equals_eager_eval():
    """The captured forward-only (eval, inference_mode) head replays the eager eval forward
    bit for bit, for the captured inputs and for new ones; wrong shapes are rejected."""
    from cerberusnet_amd.graphs import GraphedFlowInference
    det = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    try:
        torch.manual_seed(5)
        head = build("FlowEstimatorLite").to(DEV).eval()
        shapes = [(1, c, 8 * 2 ** l, 16 * 2 ** l) for l, c in enumerate(reversed(CHANS))]
        mk = lambda: [torch.randn(s, device=DEV) for s in shapes]
        p1, p2 = mk(), mk()
        run = GraphedFlowInference(head, p1, p2, bidirectional=True)
        for trial in range(3):
            if trial:
                p1, p2 = mk(), mk()
            fw, bw = run(p1, p2)
            got = [f.clone() for f in fw + bw]
            with torch.inference_mode():
                want = list(head((None, p1), (None, p2))) + list(head((None, p2), (None, p1)))
            for a, b in zip(got, want):
                assert torch.equal(a, b)
        with pytest.raises(RuntimeError, match="shape"):
            run([t[:, :1] for t in p1], p2)
    finally:
        torch.backends.cudnn.deterministic = det


def test_onnx_export_carries_the_reference_custom_nodes(golden, tmp_path):
    """SURVEY 8(f)-4 (export half): the head exports with the reference's custom node types --
    ``cerberus::correlation`` with its six integer attributes and ``torch::grid_sampler``
    (onnx_export.py:18-28) -- one correlation per pyramid level, one sampler per warped level.
    (No onnx package in this image: the serialized graph is searched for the node and attribute names.)"""
    from cerberusnet_amd.utilities.onnx_export import export_flow_head
    g = golden("pwchead_lite")
    head = build("FlowEstimatorLite", fuse_concat=False, fuse_upsample=False).to(DEV)
    p1, p2 = pyramids(g, DEV)
    path = str(tmp_path / "head.onnx")
    try:
        export_flow_head(head, [t.detach() for t in p1], [t.detach() for t in p2], path)
    except (ImportError, ModuleNotFoundError) as exc:      # the legacy exporter wants the onnx package on some builds
        pytest.skip("torch.onnx.export unavailable here: %r" % (exc,))
    blob = open(path, "rb").read()
    assert blob.count(b"correlation") >= 4 and b"cerberus" in blob
    assert blob.count(b"grid_sampler") >= 3
    for attr in (b"pad_size", b"kernel_size", b"max_displacement", b"stride1", b"stride2", b"corr_multiply",
                 b"interpolation_mode", b"padding_mode", b"align_corners"):
        assert attr in blob, attr
    with pytest.raises(ValueError, match="fuse_concat"):
        export_flow_head(build("FlowEstimatorLite").to(DEV), p1, p2, path)
