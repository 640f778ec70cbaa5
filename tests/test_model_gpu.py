"""The host model on the MI355X with the HIP backend of the hot-path ops (correlation_leaky_into,
flow_warp, flow_upsample, area_resize) against the goldens captured from the REFERENCE CerberusBase +
unFlowLoss on CPU (tools/gen_golden_model.py).  The convolutions are MIOpen's: tolerance 1e-3 of the
output range (fp32 convolution algorithms differ between MIOpen and the CPU reference)."""
import numpy as np
import pytest
import torch

from cerberusnet_amd import _lib
from cerberusnet_amd.loss_functions import unFlowLoss
from cerberusnet_amd.nnet_models import CerberusBase, cerberus_flow_config
from cerberusnet_amd.nnet_models.hrnetv2 import W32
from cerberusnet_amd.synth import fill_parameters, hash_uniform
from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def build(**extra):
    model = CerberusBase(**cerberus_flow_config(W32, **extra)).to(DEV)
    fill_parameters(model.backbone, 400)
    fill_parameters(model.flow, 500)
    return model.train()


def test_model_step_on_the_hip_ops_matches_the_reference_goldens(golden):
    g = golden("cerberus_w32")
    model = build()
    l_img = torch.from_numpy(hash_uniform((2, 3, 64, 128), 401, -2.0, 2.0)).to(DEV)
    l_seq = torch.from_numpy(hash_uniform((2, 3, 64, 128), 402, -2.0, 2.0)).to(DEV)
    out = model(l_img=l_img, l_seq=l_seq, consistency=True)
    assert _lib.last_kernel(0).startswith("corr_fwd"), _lib.last_kernel(0)
    for name in ("flow", "flow_b"):
        for i, f in enumerate(out[name]):
            assert rel_err(f.detach().cpu().numpy(), g["%s_%d" % (name, i)]) < 1e-3, (name, i)
    loss = unFlowLoss(weights={"l1": 0.15, "ssim": 0.85}, consistency=True)(out, {"l_img": l_img, "l_seq": l_seq})
    assert abs(loss.item() - float(g["loss"])) <= 1e-3 * abs(float(g["loss"]))
    grads = torch.autograd.grad(loss, list(model.parameters()))
    norms = np.array([float(x.double().norm()) for x in grads])
    ref = g["param_grad_norms"]
    big = ref > 1e-3 * ref.max()
    assert np.allclose(norms[big], ref[big], rtol=2e-2)
    assert "libcerberus_hip.so" in open("/proc/self/maps").read()


def test_hip_and_torch_backends_of_the_model_agree_on_the_gpu():
    """Same weights, same frames, the op pair swapped for the reference's own fallback ops on the same
    device: isolates the HIP kernels' contribution from MIOpen's."""
    a, b = build(), build(correlation_backend="torch")
    l_img = torch.from_numpy(hash_uniform((1, 3, 128, 256), 411, -2.0, 2.0)).to(DEV)
    l_seq = torch.from_numpy(hash_uniform((1, 3, 128, 256), 412, -2.0, 2.0)).to(DEV)
    with torch.no_grad():
        fa = a(l_img=l_img, l_seq=l_seq)["flow"]
        fb = b(l_img=l_img, l_seq=l_seq)["flow"]
    for x, y in zip(fa, fb):
        assert rel_err(x.cpu().numpy(), y.cpu().numpy()) < 1e-4


def test_config2_w18_backbone_bf16_plumbing():
    """BASELINE config 2 (HRNetV2-W18 + segmentation head, 512x256, bf16): it contains no hot-path op and its
    OCR head is outside this package's scope -- what IS here, the W18 backbone, runs forward + backward under bf16
    autocast at the config's frame size and returns the four scales the heads consume, finite."""
    from cerberusnet_amd.nnet_models import HighResolutionNet, hrnet_config
    from cerberusnet_amd.nnet_models.hrnetv2 import W18
    net = HighResolutionNet(**hrnet_config(W18)).to(DEV).train()
    fill_parameters(net, 300)
    x = torch.from_numpy(hash_uniform((2, 3, 256, 512), 421, -2.0, 2.0)).to(DEV)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        feats, pyr = net(x)
    assert feats.shape == (2, sum(W18), 64, 128) and feats.dtype in (torch.bfloat16, torch.float32)   # (autocast: the residual sums promote)
    assert [tuple(p.shape[1:]) for p in pyr] == [(144, 8, 16), (72, 16, 32), (36, 32, 64), (18, 64, 128)]
    feats.float().square().mean().backward()
    assert all(torch.isfinite(p.grad).all() for p in net.parameters() if p.grad is not None)
    assert sum(p.grad is not None for p in net.parameters()) > 300
