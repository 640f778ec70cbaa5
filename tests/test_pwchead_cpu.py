"""The caller row: this repository's PWCNetHead against goldens captured
from the REFERENCE PWCNetHead (tools/gen_golden.py, in-container, CorrelationTorch backend).
CPU: explicit 'torch' backend (wiring only).  The HIP backend is checked in test_pwchead_gpu.py."""
import numpy as np
import pytest
import torch

from cerberusnet_amd.nnet_models import PWCNetHead
from cerberusnet_amd.synth import fill_parameters
from conftest import rel_err, l2_err

CHANS = [8, 12, 16, 24]
KW = dict(correlation_args=dict(pad_size=4, kernel_size=1, max_displacement=4, stride1=1,
                                stride2=1, corr_multiply=1),
          context_network=dict(type="ContextNetwork", args={}))


def build(est, **extra):
    head = PWCNetHead(CHANS, upsample=True, flow_est_network=dict(type=est, args={}),
                      **{"1x1_conv_out": 32}, **KW, **extra)
    fill_parameters(head, 1000)
    return head.train()


def pyramids(g, device="cpu"):
    p1 = [torch.from_numpy(g["im1_%d" % l]).to(device).requires_grad_(True) for l in range(4)]
    p2 = [torch.from_numpy(g["im2_%d" % l]).to(device).requires_grad_(True) for l in range(4)]
    return p1, p2


@pytest.mark.parametrize("est,tag", [("FlowEstimatorLite", "lite"), ("FlowEstimatorDense", "dense")])
def test_state_dict_matches_reference_and_outputs_match_goldens(golden, est, tag):
    g = golden("pwchead_" + tag)
    head = build(est, correlation_backend="torch")
    sd = head.state_dict()
    assert list(sd.keys()) == list(g["keys"])            # reference checkpoints load unchanged
    assert [str(tuple(v.shape)) for v in sd.values()] == list(g["shapes"])
    p1, p2 = pyramids(g)
    flows = head((None, p1), (None, p2))
    assert len(flows) == 4
    for i, f in enumerate(flows):
        assert f.shape == g["flow_%d" % i].shape
        assert rel_err(f.detach().numpy(), g["flow_%d" % i]) < 1e-5
    loss = sum((f * f).mean() for f in flows)
    assert abs(loss.item() - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
    grads = torch.autograd.grad(loss, p1 + p2 + list(head.parameters()))
    for l in range(4):
        assert l2_err(grads[l].numpy(), g["g_im1_%d" % l]) < 1e-3
        assert l2_err(grads[4 + l].numpy(), g["g_im2_%d" % l]) < 1e-3
    norms = np.array([float(x.double().norm()) for x in grads[8:]])
    assert np.allclose(norms, g["param_grad_norms"], rtol=1e-3, atol=1e-9)


def test_default_backend_has_no_cpu_path():
    g_head = build("FlowEstimatorLite")  # correlation_backend defaults to "hip"
    feats = [torch.randn(1, c, 4 * 2 ** l, 6 * 2 ** l) for l, c in enumerate(reversed(CHANS))]
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        g_head((None, feats), (None, feats))


def test_constructor_defaults_follow_reference():
    head = PWCNetHead(CHANS)  # no kwargs: dense estimator, d=4 correlation, 32-wide 1x1
    assert type(head.flow_estimator).__name__ == "FlowEstimatorDense"
    assert head.corr.max_displacement == 4 and head.corr.pad_size == 4
    assert head.conv_1x1[0][0].in_channels == 24 and head.conv_1x1[3][0].in_channels == 8
    with pytest.raises(NotImplementedError):
        PWCNetHead(CHANS, flow_est_network=dict(type="Nope"))


def test_forward_both_equals_the_two_separate_calls_on_cpu(golden):
    """forward_both (both directions stacked along the batch axis) == the two calls of
    cerberus.py:131,135, on the explicit torch backend (CPU wiring test)."""
    g = golden("pwchead_lite")
    head = build("FlowEstimatorLite", correlation_backend="torch").eval()
    p1, p2 = pyramids(g)
    with torch.no_grad():
        fw, bw = head.forward_both((None, p1), (None, p2))
        fw2, bw2 = head((None, p1), (None, p2)), head((None, p2), (None, p1))
    for a, b in zip(fw + bw, list(fw2) + list(bw2)):
        assert a.shape == b.shape
        assert rel_err(a.numpy(), b.numpy()) < 1e-5


def test_graph_safe_reductions_equal_the_plain_ones():
    """graphs.graph_safe_sum / graph_safe_mean (block-level reductions for losses computed inside a replayed
    hipGraph): values and gradients of sum / mean, ragged sizes included."""
    from cerberusnet_amd.graphs import graph_safe_mean, graph_safe_sum
    for n in (1, 17, 4096, 4097, 3 * 4096 + 5, 2 * 2 * 64 * 128):
        x = torch.randn(n, dtype=torch.float64, requires_grad=True)
        y = x.detach().clone().requires_grad_(True)
        a, b = graph_safe_sum(x * x), (y * y).sum()
        assert torch.allclose(a, b, rtol=1e-12)
        a.backward(); b.backward()
        assert torch.allclose(x.grad, y.grad, rtol=1e-12)
        assert torch.allclose(graph_safe_mean(x.detach()), y.detach().mean(), rtol=1e-12)
    # more than block ** 2 elements: every stage must still be a row-wise reduction of <= block columns
    x = torch.randn(16 * 16 * 16 + 3, dtype=torch.float64)
    assert torch.allclose(graph_safe_sum(x, block=16), x.sum(), rtol=1e-12)
    seen = []
    real = torch.Tensor.sum
    def spy(self, *a, **k):
        seen.append((tuple(self.shape), a))
        return real(self, *a, **k)
    torch.Tensor.sum = spy
    try:
        graph_safe_sum(x, block=16)
    finally:
        torch.Tensor.sum = real
    assert all(shape[-1] <= 16 for shape, _ in seen), seen
