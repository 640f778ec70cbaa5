"""The host model (HRNetV2 backbone + PWC flow head + photometric loss; SURVEY.md section 7 step 9,
8(d) end-to-end timing) against goldens captured from the REFERENCE's own classes in the build
container (tools/gen_golden_model.py: hrnetv2.py:265-417, cerberus.py:88-146, UnFlowLoss.py:189-322).
CPU, explicit 'torch' backend of the two hot-path ops (wiring only; the HIP backend runs in
test_model_gpu.py), plus the world-2 gloo DDP equivalence of the whole model."""
import copy
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cerberusnet_amd import distributed as cdist
from cerberusnet_amd.loss_functions import unFlowLoss
from cerberusnet_amd.nnet_models import CerberusBase, HighResolutionNet, cerberus_flow_config, hrnet_config
from cerberusnet_amd.nnet_models.hrnetv2 import W18, W32
from cerberusnet_amd.synth import fill_parameters, hash_uniform
from conftest import rel_err


def sampled_ok(g, name, t, tol=1e-5):
    a = t.detach().numpy()
    assert list(a.shape) == list(g[name + "_shape"])
    scale = float(g[name + "_absmax"])
    assert np.abs(a.reshape(-1)[g[name + "_idx"]] - g[name + "_val"]).max() <= tol * scale
    d = a.astype(np.float64)
    assert abs((d * d).sum() - float(g[name + "_sumsq"])) <= 1e-4 * float(g[name + "_sumsq"])
    assert abs(np.abs(d).max() - scale) <= tol * scale


@pytest.mark.parametrize("tag,widths,nparam", [("w18", W18, 9562260), ("w32", W32, 29305536)])
def test_backbone_keys_shapes_and_forward_match_the_reference(golden, tag, widths, nparam):
    g = golden("hrnet_" + tag)
    net = HighResolutionNet(**hrnet_config(widths))
    sd = net.state_dict()
    assert list(sd.keys()) == list(g["keys"])                       # reference checkpoints load unchanged
    assert [str(tuple(v.shape)) for v in sd.values()] == list(g["shapes"])
    assert [k for k, _ in net.named_parameters()] == list(g["param_names"])
    assert sum(p.numel() for p in net.parameters()) == nparam == int(g["n_params"])
    assert sum(isinstance(m, torch.nn.BatchNorm2d) for m in net.modules()) == 305
    fill_parameters(net, 300)
    net.train()
    feats, pyr = net(torch.from_numpy(hash_uniform((2, 3, 64, 128), 301, -2.0, 2.0)))
    sampled_ok(g, "feats", feats)
    assert [p.shape[1] for p in pyr] == list(reversed(widths))      # low resolution first (hrnetv2.py:417)
    for i, p in enumerate(pyr):
        sampled_ok(g, "pyr%d" % i, p)
    sampled_ok(g, "bn1_running_mean", net.bn1.running_mean)
    lean = HighResolutionNet(concat_features=False, **hrnet_config(widths))
    lean.load_state_dict(sd)
    assert lean.train()(torch.zeros(1, 3, 32, 64))[0] is None


def build_model(**extra):
    cfg = cerberus_flow_config(W32, correlation_backend="torch")
    cfg.update(extra)
    model = CerberusBase(**cfg)
    fill_parameters(model.backbone, 400)
    fill_parameters(model.flow, 500)
    return model.train()


def frames(n=2):
    return (torch.from_numpy(hash_uniform((2, 3, 64, 128), 401, -2.0, 2.0))[:n],
            torch.from_numpy(hash_uniform((2, 3, 64, 128), 402, -2.0, 2.0))[:n])


def test_cerberus_flow_model_and_loss_match_the_reference(golden):
    g = golden("cerberus_w32")
    model = build_model()
    sd = model.state_dict()
    assert list(sd.keys()) == list(g["keys"])                       # the reference's backbone.* / flow.* entries
    assert [str(tuple(v.shape)) for v in sd.values()] == list(g["shapes"])
    assert sum(p.numel() for p in model.parameters()) == int(g["n_params_backbone_flow"])
    l_img, l_seq = frames()
    out = model(l_img=l_img, l_seq=l_seq, consistency=True)
    assert set(out) == {"flow", "flow_b"}
    for name in ("flow", "flow_b"):
        assert len(out[name]) == 4
        for i, f in enumerate(out[name]):
            assert rel_err(f.detach().numpy(), g["%s_%d" % (name, i)]) < 1e-5, (name, i)
    loss = unFlowLoss(weights={"l1": 0.15, "ssim": 0.85}, consistency=True, backend="torch")(
        out, {"l_img": l_img, "l_seq": l_seq})
    assert abs(loss.item() - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    grads = torch.autograd.grad(loss, list(model.parameters()))
    norms = np.array([float(x.double().norm()) for x in grads])
    assert np.allclose(norms, g["param_grad_norms"], rtol=2e-3, atol=1e-9)
    assert all(float(n) > 0 for n in norms)                         # every parameter receives a gradient


def test_out_of_scope_heads_raise_instead_of_being_dropped():
    cfg = cerberus_flow_config(W32)
    with pytest.raises(NotImplementedError, match="outside this package's scope"):
        CerberusBase(segmentation_config={"type": "OCRNetHead", "cfg": {}}, **cfg)
    with pytest.raises(NotImplementedError):
        unFlowLoss(weights={"ternary": 1.0})


# ---- world-2 gloo: DDP gradient of the whole model == whole-batch gradient -----------------------
def _loss(model, l_img, l_seq):
    out = model(l_img=l_img, l_seq=l_seq, consistency=True)
    return unFlowLoss(backend="torch")(out, {"l_img": l_img, "l_seq": l_seq})


def _worker(rank, world, port, path):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(3)
    assert cdist.init_from_env("gloo") == (rank, world)
    model = build_model().eval()            # BatchNorm in eval: per-replica statistics are not what is tested
    ddp = cdist.wrap_ddp(model)
    l_img, l_seq = frames()
    mine = cdist.shard_pairs(2, rank, world)
    _loss(ddp, l_img[mine], l_seq[mine]).backward()
    if rank == 0:
        torch.save([p.grad.clone() for p in model.parameters()], path)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_ddp_gradient_of_the_model_equals_single_process(tmp_path):
    path = str(tmp_path / "g.pt")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, path), nprocs=2, join=True)
    got = torch.load(path)
    torch.set_num_threads(6)
    model = build_model().eval()
    l_img, l_seq = frames()
    # the DDP mean of per-rank losses (one pair each) == the loss averaged over the two pairs
    (0.5 * (_loss(model, l_img[:1], l_seq[:1]) + _loss(model, l_img[1:], l_seq[1:]))).backward()
    worst = 0.0
    for a, p in zip(got, model.parameters()):
        worst = max(worst, float((a - p.grad).norm() / (p.grad.norm() + 1e-12)))
    assert worst < 1e-4, worst
