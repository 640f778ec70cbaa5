#!/usr/bin/env python3
"""GPU: package-free repro attempt for the wrong replayed loss scalar of GraphedFlowStep.
Nothing of cerberusnet_amd is imported.  A multi-block reduction (x.abs().mean() over 2^19+ elements,
like the loss over a full-resolution flow) is captured into a hipGraph in several ways, eager GPU work
is interleaved between replays, and the replayed scalar is compared with the eager one."""
import json
import torch

dev = "cuda:0"
torch.manual_seed(0)


def eager_noise():
    a = torch.randn(1024, 1024, device=dev)
    b = (a @ a).abs().mean()          # another multi-block reduction + allocator traffic
    c = torch.randn(3, 1 << 20, device=dev).abs().mean()
    return float(b) + float(c)


def trial(name, build):
    static_x = torch.randn(2, 2, 512, 1024, device=dev)        # 2 M elements: the final flows of 2 pairs
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            build(static_x)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = build(static_x)
    bad = 0
    worst = 0.0
    for i in range(20):
        static_x.copy_(torch.randn_like(static_x))
        g.replay()
        got = float(out if out.dim() == 0 else out.sum())
        want = float(build(static_x) if out.dim() == 0 else build(static_x).sum())
        eager_noise()
        if got != want:
            bad += 1
            worst = max(worst, abs(got - want) / max(abs(want), 1e-30))
    print(json.dumps(dict(case=name, mismatches=bad, of=20, worst_rel=worst)), flush=True)


trial("mean", lambda x: x.abs().mean())
trial("sum_of_means", lambda x: sum(t.abs().mean() for t in (x, x * 2, x[:, :1], x[:1])))
w = torch.randn(2, 2, 3, 3, device=dev, requires_grad=True)


def with_backward(x):
    w.grad = None
    y = torch.nn.functional.conv2d(x, w, padding=1)
    loss = y.abs().mean() + x.abs().mean()
    loss.backward()
    return loss.detach()


trial("conv_mean_backward", with_backward)
