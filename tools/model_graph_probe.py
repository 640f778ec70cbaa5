#!/usr/bin/env python3
"""GPU: is the host model's training step launch-bound?  Eager step vs the same step captured into one hipGraph
(zero_grad, HRNetV2-W32 on both frames, PWC head both directions, unFlowLoss, backward, Adam capturable)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cerberusnet_amd.loss_functions import unFlowLoss
from cerberusnet_amd.nnet_models import CerberusBase, cerberus_flow_config
from cerberusnet_amd.synth import fill_parameters, hash_uniform
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
model = CerberusBase(**cerberus_flow_config()).to(dev).train()
fill_parameters(model.backbone, 400); fill_parameters(model.flow, 500)
l_img = torch.from_numpy(hash_uniform((B, 3, 512, 1024), 11, -2.0, 2.0)).to(dev)
l_seq = torch.from_numpy(hash_uniform((B, 3, 512, 1024), 12, -2.0, 2.0)).to(dev)
loss_fn = unFlowLoss()
opt = torch.optim.Adam(model.parameters(), lr=1e-4, betas=(0.9, 0.99), weight_decay=1e-6, capturable=True)
def step():
    opt.zero_grad(set_to_none=True)
    out = model(l_img=l_img, l_seq=l_seq, consistency=True)
    loss = loss_fn(out, {"l_img": l_img, "l_seq": l_seq})
    loss.backward()
    opt.step()
    return loss
def timeit(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(4): step()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
print("eager: %.1f ms/step (B = %d)" % (timeit(step, 8), B), flush=True)
import gc; gc.collect()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    loss = step()
for _ in range(3): g.replay()
print("hipGraph replay: %.1f ms/step" % timeit(g.replay, 8), flush=True)
