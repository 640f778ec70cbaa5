"""N>1 on real GPUs (RCCL, world_size 2): the same two checks as tests/test_ddp_cpu.py, with the
head on the HIP ops and the exchange on a side stream.  Skips on a 1-GPU box (the driver's
8-GPU node, or any 2-GPU lease, runs it)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cerberusnet_amd import distributed as cdist
from cerberusnet_amd.synth import hash_uniform

pytestmark = pytest.mark.gpu


def _need_two_gpus():
    # device_count() does not initialise the GPU in this (parent) process
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs (this box has %d)" % torch.cuda.device_count())


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    assert cdist.init_from_env("nccl", dev) == (rank, world)
    # 1. the stand-alone bucketed exchange, overlapped with a kernel on the main stream
    ex = cdist.GradientExchange(1_000_000, dev, bucket_mb=1.0)
    ex.flat.copy_(torch.from_numpy(hash_uniform((1_000_000,), 700 + rank)))
    ex.start()
    busy = torch.ones(1 << 20, device=dev).mul_(2.0)      # main stream keeps working
    ex.finish()
    torch.cuda.synchronize()
    # 2. DDP around the flow head on the HIP ops: averaged gradient == whole-batch gradient
    from test_ddp_cpu import PAIRS, batch, step_loss
    from cerberusnet_amd.nnet_models import PWCNetHead
    from cerberusnet_amd.synth import fill_parameters
    from test_ddp_cpu import CHANS
    head = PWCNetHead(CHANS, flow_est_network=dict(type="FlowEstimatorLite", args={}))
    fill_parameters(head, 2000)
    head = cdist.wrap_ddp(head.train().to(dev), dev)
    p1, p2 = batch(cdist.shard_pairs(PAIRS, rank, world))
    step_loss(head, [t.to(dev) for t in p1], [t.to(dev) for t in p2]).backward()
    grads = [p.grad.detach().cpu() for p in head.parameters()]
    if rank == 0:
        torch.save({"flat": ex.flat.cpu(), "grads": grads, "busy": float(busy[0])}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_world2_exchange_and_ddp_gradients(tmp_path):
    _need_two_gpus()
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    want = (torch.from_numpy(hash_uniform((1_000_000,), 700)) +
            torch.from_numpy(hash_uniform((1_000_000,), 701))) / 2
    assert torch.allclose(got["flat"], want, rtol=0, atol=1e-7)
    assert got["busy"] == 2.0
    # single-process reference on the CPU 'torch' backend of the same head (whole batch)
    from test_ddp_cpu import build, batch, step_loss
    head = build()
    p1, p2 = batch(list(range(4)))
    step_loss(head, p1, p2).backward()
    for g, p in zip(got["grads"], head.parameters()):
        den = float(p.grad.norm()) or 1.0
        assert float((g - p.grad).norm()) / den < 2e-3


def _one_rank_worker(rank, port, out):
    os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)      # RCCL, one rank
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from cerberusnet_amd.loss_functions import unFlowLoss
    from cerberusnet_amd.nnet_models import CerberusBase, cerberus_flow_config
    from cerberusnet_amd.synth import fill_parameters
    torch.backends.cudnn.deterministic = True

    def build():
        m = CerberusBase(**cerberus_flow_config()).to(dev).train()
        fill_parameters(m.backbone, 400)
        fill_parameters(m.flow, 500)
        return m
    l_img = torch.from_numpy(hash_uniform((2, 3, 128, 256), 431, -2.0, 2.0)).to(dev)
    l_seq = torch.from_numpy(hash_uniform((2, 3, 128, 256), 432, -2.0, 2.0)).to(dev)
    loss_fn = unFlowLoss()
    res = {}
    for tag, wrap in (("plain", False), ("ddp", True)):
        model = build()
        net = cdist.wrap_ddp(model, dev, force=True) if wrap else model
        if wrap:
            assert isinstance(net, torch.nn.parallel.DistributedDataParallel)
        for _ in range(2):                                   # two steps: static_graph settles in the first
            for p in model.parameters():
                p.grad = None
            out_ = net(l_img=l_img, l_seq=l_seq, consistency=True)
            loss_fn(out_, {"l_img": l_img, "l_seq": l_seq}).backward()
        res[tag] = [p.grad.detach().cpu() for p in model.parameters()]
        res[tag + "_keys"] = list(cdist.rank0_state_dict(net).keys())
    # the stand-alone exchange on the same communicator
    ex = cdist.GradientExchange(300_000, dev, bucket_mb=0.5)
    ex.flat.copy_(torch.from_numpy(hash_uniform((300_000,), 433)))
    ex.start(); ex.finish(); torch.cuda.synchronize()
    res["flat"] = ex.flat.cpu()
    torch.save(res, out)
    dist.destroy_process_group()


def test_rccl_one_rank_group_runs_the_model_under_ddp(tmp_path):
    """What a 1-GPU box CAN verify of the N > 1 path: a real RCCL communicator (one rank), the host model under
    DistributedDataParallel (bucket hooks, gradient_as_bucket_view, static_graph with the head called twice) on the
    HIP ops, gradients equal to the unwrapped model's, checkpoints without the `module.` prefix, and the bucketed
    exchange (an average over one rank = identity)."""
    out = str(tmp_path / "one.pt")
    mp.spawn(_one_rank_worker, args=(_free_port(), out), nprocs=1, join=True)
    got = torch.load(out)
    assert got["plain_keys"] == got["ddp_keys"] and not any(k.startswith("module.") for k in got["ddp_keys"])
    worst = max(float((a - b).norm() / (b.norm() + 1e-12)) for a, b in zip(got["ddp"], got["plain"]))
    assert worst < 1e-4, worst
    assert torch.allclose(got["flat"], torch.from_numpy(hash_uniform((300_000,), 433)), rtol=0, atol=1e-7)
