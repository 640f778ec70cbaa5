"""Repro of the test order: construct the graph FIRST (nothing eager before), one replay per
trial, eager after it.  Prints which flow level diverges, for a few head configurations."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from cerberusnet_amd.graphs import GraphedFlowStep
from test_pwchead_cpu import CHANS, build
DEV = "cuda:0"
torch.backends.cudnn.deterministic = True
cfg = sys.argv[1] if len(sys.argv) > 1 else "default"
kw = {"default": {}, "nocat": {"fuse_concat": False}, "noleaky": {"fuse_leaky": False, "fuse_concat": False},
      "torch": {"correlation_backend": "torch"}, "noupsample": {"fuse_upsample": False},
      "plain": {"fuse_upsample": False, "fuse_leaky": False, "fuse_concat": False}}[cfg]
if cfg == "torch":
    # the reference's mesh_grid builds its mesh on the CPU and copies it over (UnFlowLoss.py:11-20): not
    # capturable.  For this diagnosis only: the same mesh built on the device.
    import cerberusnet_amd.nnet_models.pwcnet_sfd as _m
    def _mesh_on_device(B, H, W):
        ys, xs = torch.meshgrid(torch.arange(H, device=DEV, dtype=torch.float32),
                                torch.arange(W, device=DEV, dtype=torch.float32), indexing="ij")
        return torch.stack([xs, ys], 0).unsqueeze(0).repeat(B, 1, 1, 1)
    _m.mesh_grid = _mesh_on_device
torch.manual_seed(3)
head = build("FlowEstimatorLite", **kw).to(DEV)
shapes = [(2, c, 8 * 2 ** l, 16 * 2 ** l) for l, c in enumerate(reversed(CHANS))]
mk = lambda: [torch.randn(s, device=DEV) for s in shapes]
if "blockloss" in sys.argv:
    # block-level reductions only (rows of 4096, then <= 512 values): no multi-block "global"
    # reduction with a semaphore buffer inside the captured graph
    loss_fn = lambda flows: sum(f.abs().reshape(-1, 4096).mean(1).mean() for f in flows)
else:
    loss_fn = lambda flows: sum(f.abs().mean() for f in flows)

def eager(p1, p2):
    for p in head.parameters():
        p.grad = None
    a = [t.clone().requires_grad_(True) for t in p1]
    b = [t.clone().requires_grad_(True) for t in p2]
    fw = head((None, a), (None, b))
    bw = head((None, b), (None, a))
    loss = loss_fn(list(fw) + list(bw))
    loss.backward()
    _keep = [p.grad.detach().clone() for p in head.parameters()] + [t.grad.detach().clone() for t in a + b]
    return loss.detach().clone(), [f.detach().clone() for f in fw], [f.detach().clone() for f in bw]

p1, p2 = mk(), mk()
step = GraphedFlowStep(head, loss_fn, p1, p2, input_grads=True)
for trial in range(4):
    if trial:
        p1, p2 = mk(), mk()
    loss, fw, bw = step(p1, p2)
    if "sync" in sys.argv:
        torch.cuda.synchronize()
    g = (step.captured_loss.clone(), [f.clone() for f in fw], [f.clone() for f in bw])   # the scalar the GRAPH reduced
    if "noeager" in sys.argv and trial == 0:
        e = g
    else:
        e = eager(p1, p2)
    print(cfg, "trial", trial, "loss", float(g[0]), float(e[0]),
          "fw equal", [bool(torch.equal(x, y)) for x, y in zip(g[1], e[1])],
          "bw equal", [bool(torch.equal(x, y)) for x, y in zip(g[2], e[2])])
