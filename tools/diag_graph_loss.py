"""Round 4: where does the captured loss scalar of GraphedFlowStep go wrong?
Same protocol as tools/diag_graph_order.py (graph first, one replay per trial, an EAGER step after each), but the
captured step also keeps every per-flow term of the loss alive as a graph output, in three forms:
  means[i]   = flows[i].abs().mean()                 (ATen's multi-block reduction: scratch buffer + semaphores + memset)
  rows[i]    = flows[i].abs().reshape(-1, 4096).mean(1)   (one block per row: no scratch)
  loss       = sum(means)                            (python sum: every partial sum is freed inside the capture)
  loss_stack = torch.stack(means).sum()              (no freed intermediates)
and prints which of them differ from the eager values of the same replay's flows.  argv: [torch] [samestream]"""
import os, sys, warnings
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_pwchead_cpu import CHANS, build
DEV = "cuda:0"
torch.backends.cudnn.deterministic = True
kw = {"correlation_backend": "torch"} if "torch" in sys.argv else {}
if "torch" in sys.argv:
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
static1, static2 = mk(), mk()
params = [p for p in head.parameters() if p.requires_grad]

def step():
    for p in params:
        p.grad = None
    fw = head((None, static1), (None, static2))
    bw = head((None, static2), (None, static1))
    flows = list(fw) + list(bw)
    means = [f.abs().mean() for f in flows]
    rows = [f.abs().reshape(-1, 4096).mean(1) if f.numel() % 4096 == 0 else f.abs().reshape(1, -1).mean(1) for f in flows]
    loss = sum(means)
    loss_stack = torch.stack(means).sum()
    loss.backward()
    return dict(flows=flows, means=means, rows=rows, loss=loss.detach(), loss_stack=loss_stack.detach())

side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with warnings.catch_warnings(record=True) as wlist:
    warnings.simplefilter("always")
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    import gc; gc.collect()
    g = torch.cuda.CUDAGraph()
    if "samestream" in sys.argv:
        with torch.cuda.graph(g, stream=side):
            out = step()
    else:
        with torch.cuda.graph(g):
            out = step()
print("warnings during warm-up + capture:", sorted({str(w.message)[:70] for w in wlist}))
# addresses: the graph's outputs against each other
ptrs = [("mean%d" % i, t) for i, t in enumerate(out["means"])] + [("loss", out["loss"]), ("loss_stack", out["loss_stack"])]
print("graph-pool addresses of the scalars:", [(n, hex(t.data_ptr())) for n, t in ptrs])

def eager_terms(flows):
    with torch.no_grad():
        means = [f.abs().mean() for f in flows]
        return means, sum(means), torch.stack(means).sum()

for trial in range(5):
    for dst, src in zip(static1 + static2, mk() + mk()):
        dst.copy_(src)
    g.replay()
    torch.cuda.synchronize()
    snap = {k: ([t.clone() for t in v] if isinstance(v, list) else v.clone()) for k, v in out.items()}
    means_e, loss_e, stack_e = eager_terms(snap["flows"])
    bad_means = [i for i, (a, b) in enumerate(zip(snap["means"], means_e)) if not torch.equal(a, b)]
    rows_ok = all(torch.allclose(r.mean(), m, rtol=1e-5) for r, m in zip(snap["rows"], means_e))
    print("trial %d: captured loss %.6f  eager-on-the-same-flows %.6f | sum(means) of the captured terms %.6f | loss_stack %.6f (%s) | wrong means %s | row means consistent %s"
          % (trial, float(snap["loss"]), float(loss_e), float(sum(snap["means"])), float(snap["loss_stack"]),
             "ok" if torch.equal(snap["loss_stack"], stack_e) else "WRONG", bad_means, rows_ok))
    # the interleaved EAGER work (what triggers the defect)
    if "onlyreduce" in sys.argv:      # eager multi-block reductions of same-sized tensors, nothing else
        for f in snap["flows"]:
            torch.randn_like(f).abs().mean()
        torch.cuda.synchronize()
        continue
    if "onlyalloc" in sys.argv:       # eager allocations + fills of the scratch sizes, no reductions
        junk = [torch.full((n,), 7.0e9, device=DEV) for n in (64, 128, 256, 512, 1024, 4096, 16384, 65536)]
        torch.cuda.synchronize()
        del junk
        continue
    if "noeager" in sys.argv:
        continue
    a = [t.clone().requires_grad_(True) for t in mk()]
    b = [t.clone().requires_grad_(True) for t in mk()]
    fw = head((None, a), (None, b)); bw = head((None, b), (None, a))
    l = sum(f.abs().mean() for f in list(fw) + list(bw)); l.backward()
    torch.cuda.synchronize()
