#!/usr/bin/env python3
"""Flow-head step (both directions, loss, backward) at the config-3 pyramid: eager vs one
hipGraph (cerberusnet_amd.graphs.GraphedFlowStep), HIP ops vs stock-PyTorch ops.
usage: bench_head.py [pairs=1,4] [estimator=FlowEstimatorLite]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cerberusnet_amd.graphs import GraphedFlowInference, GraphedFlowStep
from cerberusnet_amd.nnet_models import PWCNetHead
from cerberusnet_amd.synth import W32_PYRAMID_1024x512, fill_parameters, hash_uniform

pairs_list = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1,4").split(",")]
est = sys.argv[2] if len(sys.argv) > 2 else "FlowEstimatorLite"
chans = [c for c, _, _ in reversed(W32_PYRAMID_1024x512)]          # high-res first, as HRNet lists them
corr_args = dict(pad_size=4, kernel_size=1, max_displacement=4, stride1=1, stride2=1, corr_multiply=1)
loss_fn = lambda flows: sum(f.abs().mean() for f in flows)


def timed(fn, n):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for backend in ("hip", "torch"):
    head = PWCNetHead(chans, upsample=True, correlation_args=corr_args,
                      flow_est_network=dict(type=est, args={}),
                      context_network=dict(type="ContextNetwork", args={}),
                      correlation_backend=backend, **{"1x1_conv_out": 32}).cuda().train()
    fill_parameters(head, 7)
    for B in pairs_list:
        pyr = lambda seed: [torch.from_numpy(hash_uniform((B, C, H, W), seed + i)).cuda()
                            for i, (C, H, W) in enumerate(W32_PYRAMID_1024x512)]
        p1, p2 = pyr(10), pyr(20)

        def eager():
            for p in head.parameters():
                p.grad = None
            fw = head((None, p1), (None, p2))
            bw = head((None, p2), (None, p1))
            loss_fn(list(fw) + list(bw)).backward()

        t_eager = timed(eager, 10)
        row = dict(backend=backend, estimator=est, pairs=B, eager_ms=round(t_eager * 1e3, 3))
        if backend == "hip":
            step = GraphedFlowStep(head, loss_fn, p1, p2)
            t_graph = timed(lambda: step.graph.replay(), 20)
            row.update(graph_ms=round(t_graph * 1e3, 3), pairs_per_s_graph=round(B / t_graph, 1))
        row["pairs_per_s_eager"] = round(B / t_eager, 1)
        if backend == "hip":
            # inference (eval, forward only, one direction): eager vs the captured forward
            head.eval()
            def infer():
                with torch.inference_mode():
                    head((None, p1), (None, p2))
            t_inf = timed(infer, 20)
            run = GraphedFlowInference(head, p1, p2)
            t_ginf = timed(lambda: run.graph.replay(), 50)
            row.update(infer_eager_ms=round(t_inf * 1e3, 3), infer_graph_ms=round(t_ginf * 1e3, 3))
            head.train()
        print(json.dumps(row), flush=True)
