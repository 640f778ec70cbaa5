"""Diagnose GraphedFlowStep vs eager (round-1 red test): how far apart are (a) two eager runs,
(b) graph replay and eager, per tensor, for a non-smooth and a smooth loss.  GPU only."""
import gc
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from cerberusnet_amd.graphs import GraphedFlowStep  # noqa: E402
from test_pwchead_cpu import CHANS, build  # noqa: E402

DEV = "cuda:0"


def l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / max(float(b.norm()), 1e-300))


def mx(a, b):
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-300))


def frac(a, b, tol=1e-5):
    """fraction of elements further apart than tol * max|b|"""
    return float(((a - b).abs() > tol * b.abs().max()).double().mean())


def run(loss_name):
    torch.manual_seed(3)
    head = build("FlowEstimatorLite").to(DEV)
    shapes = [(2, c, 8 * 2 ** l, 16 * 2 ** l) for l, c in enumerate(reversed(CHANS))]
    mk = lambda: [torch.randn(s, device=DEV) for s in shapes]
    if loss_name == "abs":
        loss_fn = lambda flows: sum(f.abs().mean() for f in flows)
    else:
        loss_fn = lambda flows: sum((f * f).mean() for f in flows)

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
                [t.grad.detach().clone() for t in a])

    p1, p2 = mk(), mk()
    e1 = eager(p1, p2)
    e2 = eager(p1, p2)
    print("== loss", loss_name)
    print("eager vs eager: loss", float(e1[0]), float(e2[0]),
          "flows bit-identical:", [bool((a == b).all()) for a, b in zip(e1[1], e2[1])])
    print("  param grads  max l2:", max(l2(a, b) for a, b in zip(e1[2], e2[2])))
    print("  input grads  l2:", [l2(a, b) for a, b in zip(e1[3], e2[3])],
          "frac>1e-5:", [frac(a, b) for a, b in zip(e1[3], e2[3])])
    gc.collect()
    step = GraphedFlowStep(head, loss_fn, p1, p2, input_grads=True)
    for trial in range(3):
        if trial:
            p1, p2 = mk(), mk()
        loss, fw, _ = step(p1, p2)
        torch.cuda.synchronize()
        g = (loss.detach().clone(), [f.detach().clone() for f in fw],
             [p.grad.detach().clone() for p in head.parameters()],
             [x.detach().clone() for x in step.input_gradients()[0]])
        e = eager(p1, p2)
        print("trial", trial, "graph vs eager: loss", float(g[0]), float(e[0]),
              "flows bit-identical:", [bool((a == b).all()) for a, b in zip(g[1], e[1])],
              "flow max:", [mx(a, b) for a, b in zip(g[1], e[1])])
        print("  param grads  max l2:", max(l2(a, b) for a, b in zip(g[2], e[2])))
        print("  input grads  l2:", [l2(a, b) for a, b in zip(g[3], e[3])],
              "max:", [mx(a, b) for a, b in zip(g[3], e[3])],
              "frac>1e-5:", [frac(a, b) for a, b in zip(g[3], e[3])])
        # replay twice: is the graph itself reproducible?
        loss2, _, _ = step(p1, p2)
        torch.cuda.synchronize()
        g2 = [x.detach().clone() for x in step.input_gradients()[0]]
        print("  graph vs graph input grads l2:", [l2(a, b) for a, b in zip(g2, g[3])])


if __name__ == "__main__":
    for det in (False, True):
        torch.backends.cudnn.deterministic = det
        print("#### cudnn.deterministic =", det)
        for name in ("abs", "sq"):
            run(name)
