"""INTEGRATION.md section 2, executed: the REFERENCE's own, unchanged `PWCNetHead` (nnet_models/pwcnet_sfd.py:121-203)
imported from /root/reference with the two names it binds -- `nnet_training.correlation_package.correlation` and
`flow_warp` of `nnet_training.loss_functions.UnFlowLoss` -- resolved to this package.  There is no GPU here and the
product has no CPU path, so what can be shown is the wiring: the reference head constructs this package's
`Correlation` from its `correlation_args`, its forward reaches `torch.ops.cerberus.correlation` through
`CorrelationFunction`, and the op answers with its own "no CPU implementation" error (not an AttributeError, not a
silent fallback).  Build container only: skipped where /root/reference does not exist (the GPU box)."""
import importlib.util
import os
import sys
import types

import pytest
import torch

REF = "/root/reference/nnet_training"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree exists in the build container only")


def _load(modname, path):
    spec = importlib.util.spec_from_file_location(modname, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


def test_the_reference_head_runs_on_this_packages_ops_through_the_two_patched_names():
    import cerberusnet_amd
    from cerberusnet_amd.correlation_package import correlation as our_corr
    saved = {k: v for k, v in sys.modules.items() if k.startswith("nnet_training")}
    try:
        for name, sub in (("nnet_training", ""), ("nnet_training.loss_functions", "loss_functions"),
                          ("nnet_training.correlation_package", "correlation_package"),
                          ("nnet_training.nnet_models", "nnet_models")):
            pkg = types.ModuleType(name)
            pkg.__path__ = [os.path.join(REF, sub)]
            sys.modules[name] = pkg
        # patch 1: the correlation module IS this package's
        sys.modules["nnet_training.correlation_package.correlation"] = our_corr
        # patch 2: flow_warp of the reference's UnFlowLoss module is this package's
        _load("nnet_training.loss_functions.loss_functions", os.path.join(REF, "loss_functions", "loss_functions.py"))
        unflow = _load("nnet_training.loss_functions.UnFlowLoss", os.path.join(REF, "loss_functions", "UnFlowLoss.py"))
        unflow.flow_warp = cerberusnet_amd.flow_warp
        for name in ("pwcnet_modules", "nnet_ops", "fast_scnn"):
            _load("nnet_training.nnet_models." + name, os.path.join(REF, "nnet_models", name + ".py"))
        ref = _load("nnet_training.nnet_models.pwcnet_sfd", os.path.join(REF, "nnet_models", "pwcnet_sfd.py"))
        assert ref.Correlation is our_corr.Correlation and ref.flow_warp is cerberusnet_amd.flow_warp
        head = ref.PWCNetHead([8, 12, 16, 24], upsample=True,
                              correlation_args=dict(pad_size=4, kernel_size=1, max_displacement=4, stride1=1, stride2=1,
                                                    corr_multiply=1),
                              flow_est_network=dict(type="FlowEstimatorLite", args={}),
                              context_network=dict(type="ContextNetwork", args={}), **{"1x1_conv_out": 32})
        assert isinstance(head.corr, our_corr.Correlation) and len(head.corr.state_dict()) == 0
        assert (head.corr.pad_size, head.corr.max_displacement, head.corr.kernel_size) == (4, 4, 1)
        pyr = [torch.randn(1, c, 4 * 2 ** l, 6 * 2 ** l) for l, c in enumerate([24, 16, 12, 8])]
        for training in (True, False):                      # train -> CorrelationFunction.apply, eval -> the raw op
            head.train(training)
            with pytest.raises(RuntimeError, match="no CPU implementation"):
                head((None, pyr), (None, pyr))
    finally:
        for k in [k for k in sys.modules if k.startswith("nnet_training")]:
            del sys.modules[k]
        sys.modules.update(saved)
