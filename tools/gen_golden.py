#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own Python on CPU.

Runs only in the build container (needs /root/reference); the fixtures it
writes are committed and are the only thing that travels to the GPU box.
Nothing from the reference is copied: its modules are imported from where they
lie (SURVEY.md Appendix A recipe) and only inputs / outputs are saved.

  * corr_*.npz  : CorrelationTorch(d)(x1, x2) and torch-autograd gradients for a
                  fixed gradOutput    (correlation_package/correlation.py:4-21)
  * warp_*.npz  : flow_warp(image, flow, pad) and autograd gradients w.r.t.
                  image and flow      (loss_functions/UnFlowLoss.py:83-94)
  * warp_q2.npz : the zero-flow-is-not-identity vector
  * fullsize.npz: checksums + 64 sampled elements of the reference output for
                  the config-1 tensor and the four config-3 pyramid levels,
                  inputs from cerberusnet_amd.synth.hash_uniform
"""
import importlib.util
import os
import sys
import types
import warnings

import numpy as np
import torch

REF = "/root/reference/nnet_training"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)


def _load(modname, path):
    spec = importlib.util.spec_from_file_location(modname, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


def import_reference():
    """Appendix A: bare namespace packages + load-by-path, load_library no-op."""
    for name, sub in (("nnet_training", ""),
                      ("nnet_training.loss_functions", "loss_functions"),
                      ("nnet_training.correlation_package", "correlation_package")):
        pkg = types.ModuleType(name)
        pkg.__path__ = [os.path.join(REF, sub)]
        sys.modules[name] = pkg
    _load("nnet_training.loss_functions.loss_functions",
          os.path.join(REF, "loss_functions", "loss_functions.py"))
    unflow = _load("nnet_training.loss_functions.UnFlowLoss",
                   os.path.join(REF, "loss_functions", "UnFlowLoss.py"))
    real_load = torch.ops.load_library
    torch.ops.load_library = lambda *_a, **_k: None
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            corr = _load("nnet_training.correlation_package.correlation",
                         os.path.join(REF, "correlation_package",
                                      "correlation.py"))
    finally:
        torch.ops.load_library = real_load
    return corr, unflow


def import_reference_pwc_head(corr_mod):
    """PWCNetHead from the reference tree (Appendix A step 3), with Correlation.forward routed
    to the reference's own CorrelationTorch (torch.ops.cerberus does not exist on CPU)."""
    pkg = types.ModuleType("nnet_training.nnet_models")
    pkg.__path__ = [os.path.join(REF, "nnet_models")]
    sys.modules["nnet_training.nnet_models"] = pkg
    for name in ("pwcnet_modules", "nnet_ops", "fast_scnn"):
        _load("nnet_training.nnet_models." + name, os.path.join(REF, "nnet_models", name + ".py"))

    def torch_forward(self, a, b):
        assert self.pad_size == self.max_displacement and self.kernel_size == 1
        assert self.stride1 == 1 and self.stride2 == 1
        return corr_mod.CorrelationTorch(self.max_displacement)(a, b)
    corr_mod.Correlation.forward = torch_forward
    mod = _load("nnet_training.nnet_models.pwcnet_sfd",
                os.path.join(REF, "nnet_models", "pwcnet_sfd.py"))
    return mod.PWCNetHead


def sampled(arr, n=64):
    flat = arr.reshape(-1)
    idx = (np.arange(n, dtype=np.int64) * 2654435761 + 12345) % flat.size
    return idx, flat[idx].copy()


def main():
    from cerberusnet_amd.synth import hash_uniform, W32_PYRAMID_1024x512
    os.makedirs(OUT, exist_ok=True)
    corr_mod, unflow = import_reference()
    torch.set_num_threads(8)

    # ---------------- correlation small goldens ----------------
    for tag, (B, C, H, W, d) in {"a": (2, 5, 6, 7, 2), "b": (1, 32, 16, 24, 4),
                                 "c": (2, 64, 9, 11, 4),
                                 "d": (1, 3, 5, 4, 1)}.items():
        x1 = torch.from_numpy(hash_uniform((B, C, H, W), 10 + ord(tag)))
        x2 = torch.from_numpy(hash_uniform((B, C, H, W), 20 + ord(tag)))
        go = torch.from_numpy(hash_uniform((B, (2 * d + 1) ** 2, H, W),
                                           30 + ord(tag)))
        x1.requires_grad_(True)
        x2.requires_grad_(True)
        out = corr_mod.CorrelationTorch(d)(x1, x2)
        g1, g2 = torch.autograd.grad(out, (x1, x2), go)
        # fp64 run of the same reference code (tighter anchor for the oracle)
        x1d = x1.detach().double().requires_grad_(True)
        x2d = x2.detach().double().requires_grad_(True)
        outd = corr_mod.CorrelationTorch(d)(x1d, x2d)
        g1d, g2d = torch.autograd.grad(outd, (x1d, x2d), go.double())
        np.savez_compressed(
            os.path.join(OUT, "corr_%s.npz" % tag), d=np.int64(d),
            x1=x1.detach().numpy(), x2=x2.detach().numpy(), gout=go.numpy(),
            out=out.detach().numpy(), g1=g1.numpy(), g2=g2.numpy(),
            out64=outd.detach().numpy(), g1_64=g1d.numpy(), g2_64=g2d.numpy())

    # ---------------- flow_warp small goldens ----------------
    for tag, (B, C, H, W) in {"a": (2, 3, 7, 9), "b": (1, 16, 12, 20)}.items():
        img = torch.from_numpy(hash_uniform((B, C, H, W), 40 + ord(tag)))
        # flows ~ +-8 px: plenty of out-of-range sample points
        flo = torch.from_numpy(hash_uniform((B, 2, H, W), 50 + ord(tag),
                                            -8.0, 8.0))
        go = torch.from_numpy(hash_uniform((B, C, H, W), 60 + ord(tag)))
        rec = dict(image=img.numpy(), flow=flo.numpy(), gout=go.numpy())
        for pad in ("border", "zeros"):
            i = img.clone().requires_grad_(True)
            f = flo.clone().requires_grad_(True)
            out = unflow.flow_warp(i, f, pad=pad)
            gi, gf = torch.autograd.grad(out, (i, f), go)
            rec["out_" + pad] = out.detach().numpy()
            rec["gimage_" + pad] = gi.numpy()
            rec["gflow_" + pad] = gf.numpy()
            rec["nearest_" + pad] = unflow.flow_warp(
                img, flo, pad=pad, mode="nearest").numpy()
        np.savez_compressed(os.path.join(OUT, "warp_%s.npz" % tag), **rec)

    # Q2: zero flow is not the identity
    ramp = torch.arange(24, dtype=torch.float32).view(1, 1, 4, 6)
    q2 = unflow.flow_warp(ramp, torch.zeros(1, 2, 4, 6))
    np.savez_compressed(os.path.join(OUT, "warp_q2.npz"), image=ramp.numpy(),
                        out=q2.numpy())

    # ---------------- PWCNetHead (the caller of both ops) ----------------
    from cerberusnet_amd.synth import fill_parameters
    Head = import_reference_pwc_head(corr_mod)
    chans = [8, 12, 16, 24]                      # HRNet order: high resolution first
    sizes = [(4, 6), (8, 12), (16, 24), (32, 48)]  # pyramid order: low resolution first
    for est in ("FlowEstimatorLite", "FlowEstimatorDense"):
        head = Head(chans, upsample=True,
                    correlation_args=dict(pad_size=4, kernel_size=1, max_displacement=4,
                                          stride1=1, stride2=1, corr_multiply=1),
                    flow_est_network=dict(type=est, args={}),
                    context_network=dict(type="ContextNetwork", args={}),
                    **{"1x1_conv_out": 32})
        fill_parameters(head, 1000)
        head.train()
        rec = {"keys": np.array(list(head.state_dict().keys())),
               "shapes": np.array([str(tuple(v.shape)) for v in head.state_dict().values()])}
        pyr1, pyr2 = [], []
        for lvl, ((h, w), c) in enumerate(zip(sizes, reversed(chans))):
            a = hash_uniform((1, c, h, w), 70 + lvl)
            b = hash_uniform((1, c, h, w), 80 + lvl)
            rec["im1_%d" % lvl], rec["im2_%d" % lvl] = a, b
            pyr1.append(torch.from_numpy(a).requires_grad_(True))
            pyr2.append(torch.from_numpy(b).requires_grad_(True))
        flows = head((None, pyr1), (None, pyr2))
        loss = sum((f * f).mean() for f in flows)
        grads = torch.autograd.grad(loss, pyr1 + pyr2 + list(head.parameters()))
        for i, f in enumerate(flows):
            rec["flow_%d" % i] = f.detach().numpy()
        rec["loss"] = np.float64(loss.item())
        for lvl in range(4):
            rec["g_im1_%d" % lvl] = grads[lvl].numpy()
            rec["g_im2_%d" % lvl] = grads[4 + lvl].numpy()
        rec["param_grad_norms"] = np.array([float(g.double().norm()) for g in grads[8:]])
        np.savez_compressed(os.path.join(OUT, "pwchead_%s.npz" % est[13:].lower()), **rec)

    # ---------------- full-size checksums ----------------
    rec = {}
    shapes = {"cfg1": (1, 64, 64, 128)}
    for lvl, (C, H, W) in enumerate(W32_PYRAMID_1024x512):
        shapes["L%d" % lvl] = (1, C, H, W)
    for name, (B, C, H, W) in shapes.items():
        x1 = torch.from_numpy(hash_uniform((B, C, H, W), 0)).requires_grad_(True)
        x2 = torch.from_numpy(hash_uniform((B, C, H, W), 1)).requires_grad_(True)
        go = torch.from_numpy(hash_uniform((B, 81, H, W), 2))
        out = corr_mod.CorrelationTorch(4)(x1, x2)
        g1, g2 = torch.autograd.grad(out, (x1, x2), go)
        for key, t in (("out", out.detach()), ("g1", g1), ("g2", g2)):
            a = t.numpy().astype(np.float64)
            idx, val = sampled(t.numpy())
            rec["%s_%s_sum" % (name, key)] = a.sum()
            rec["%s_%s_sumsq" % (name, key)] = (a * a).sum()
            rec["%s_%s_absmax" % (name, key)] = np.abs(a).max()
            rec["%s_%s_idx" % (name, key)] = idx
            rec["%s_%s_val" % (name, key)] = val
        rec["%s_shape" % name] = np.array([B, C, H, W])
    np.savez_compressed(os.path.join(OUT, "fullsize.npz"), **rec)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
