#!/usr/bin/env python3
"""Goldens for the HOST MODEL (HRNetV2 backbone + flow head + photometric loss), captured by running
the REFERENCE's own classes on CPU in the build container.  Writes

  tests/golden/hrnet_w18.npz, hrnet_w32.npz : ``HighResolutionNet`` state_dict keys / shapes and, for
       hash-filled weights and a hash-generated frame, sampled elements + checksums of the five outputs
  tests/golden/cerberus_w32.npz : the reference ``CerberusBase`` (W32 + OCR + DepthHeadV1 + PWCNetHead,
       Correlation routed to the reference's CorrelationTorch) on a 64x128 frame pair: the keys of its
       ``backbone.`` / ``flow.`` entries, the eight returned flow tensors, the reference ``unFlowLoss``
       value on them and the norms of d loss / d parameter

Nothing of the reference is copied: modules are imported from /root/reference, only numbers are saved.
"""
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
import gen_golden as gg  # noqa: E402

REF, OUT = gg.REF, gg.OUT


def import_reference_cerberus():
    corr_mod, unflow = gg.import_reference()
    gg.import_reference_pwc_head(corr_mod)
    stub = types.ModuleType("nnet_training.nnet_models.detr_sfd")
    stub.DetrSegmHead = type("DetrSegmHead", (torch.nn.Module,), {})
    sys.modules["nnet_training.nnet_models.detr_sfd"] = stub
    for name in ("hrnetv2", "ocr_utils", "ocrnet", "ocrnet_sfd", "aspp", "deeplab_panoptic", "cerberus"):
        gg._load("nnet_training.nnet_models." + name, os.path.join(REF, "nnet_models", name + ".py"))
    return sys.modules["nnet_training.nnet_models.cerberus"], sys.modules["nnet_training.nnet_models.hrnetv2"], unflow


def stats(rec, name, t):
    a = t.detach().numpy()
    idx, val = gg.sampled(a)
    d = a.astype(np.float64)
    rec[name + "_shape"] = np.array(a.shape)
    rec[name + "_sum"], rec[name + "_sumsq"], rec[name + "_absmax"] = d.sum(), (d * d).sum(), np.abs(d).max()
    rec[name + "_idx"], rec[name + "_val"] = idx, val


def main():
    import copy
    from cerberusnet_amd.nnet_models.hrnetv2 import hrnet_config, W18, W32
    from cerberusnet_amd.nnet_models.cerberus import cerberus_flow_config
    from cerberusnet_amd.synth import fill_parameters, hash_uniform
    cerb, hr, unflow = import_reference_cerberus()
    torch.set_num_threads(8)

    for tag, widths in (("w18", W18), ("w32", W32)):
        net = hr.HighResolutionNet(**copy.deepcopy(hrnet_config(widths)))
        fill_parameters(net, 300)
        net.train()
        x = torch.from_numpy(hash_uniform((2, 3, 64, 128), 301, -2.0, 2.0))
        feats, pyr = net(x)
        rec = {"keys": np.array(list(net.state_dict().keys())),
               "shapes": np.array([str(tuple(v.shape)) for v in net.state_dict().values()]),
               "param_names": np.array([k for k, _ in net.named_parameters()]),
               "n_params": np.int64(sum(p.numel() for p in net.parameters()))}
        stats(rec, "feats", feats)
        for i, p in enumerate(pyr):
            stats(rec, "pyr%d" % i, p)
        # running statistics after one training-mode forward (BatchNorm momentum 0.1)
        stats(rec, "bn1_running_mean", net.bn1.running_mean)
        np.savez_compressed(os.path.join(OUT, "hrnet_%s.npz" % tag), **rec)

    # ---- the reference CerberusBase, all three heads, on a small frame pair ----
    cfg = cerberus_flow_config(W32)
    model = cerb.CerberusBase(
        name="golden", backbone_config=copy.deepcopy(cfg["backbone_config"]),
        segmentation_config={"type": "OCRNetHead", "cfg": {"mid_channels": 64, "key_channels": 32, "classes": 19}},
        depth_config={"type": "DepthHeadV1", "cfg": {"inter_ch": [32, 16]}},
        flow_config=copy.deepcopy(cfg["flow_config"]))
    fill_parameters(model.backbone, 400)
    fill_parameters(model.flow, 500)
    model.train()
    l_img = torch.from_numpy(hash_uniform((2, 3, 64, 128), 401, -2.0, 2.0))
    l_seq = torch.from_numpy(hash_uniform((2, 3, 64, 128), 402, -2.0, 2.0))
    out = model(l_img=l_img, l_seq=l_seq, consistency=True)
    sd = model.state_dict()
    keep = [k for k in sd if k.startswith(("backbone.", "flow."))]
    rec = {"keys": np.array(keep), "shapes": np.array([str(tuple(sd[k].shape)) for k in keep]),
           "n_params_backbone_flow": np.int64(sum(p.numel() for n, p in model.named_parameters()
                                                  if n.startswith(("backbone.", "flow."))))}
    for name in ("flow", "flow_b"):
        for i, f in enumerate(out[name]):
            rec["%s_%d" % (name, i)] = f.detach().numpy()
    loss_mod = unflow.unFlowLoss(weights={"l1": 0.15, "ssim": 0.85}, consistency=True)
    loss = loss_mod({"flow": out["flow"], "flow_b": out["flow_b"]}, {"l_img": l_img, "l_seq": l_seq})
    params = [p for n, p in model.named_parameters() if n.startswith(("backbone.", "flow."))]
    grads = torch.autograd.grad(loss, params)
    rec["loss"] = np.float64(loss.item())
    rec["param_grad_norms"] = np.array([float(g.double().norm()) for g in grads])
    np.savez_compressed(os.path.join(OUT, "cerberus_w32.npz"), **rec)
    for f in ("hrnet_w18.npz", "hrnet_w32.npz", "cerberus_w32.npz"):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
