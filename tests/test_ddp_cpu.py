"""N>1 wiring on CPU (gloo, world_size 2): sharding image pairs over ranks + DDP gradient
averaging of the flow head equals the single-process gradient on the whole batch
(SURVEY.md 8e equivalence test).  Uses the explicit 'torch' backend of the head: the HIP ops
have no CPU path, and this test is about the distributed wiring, not the kernels."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cerberusnet_amd import distributed as cdist
from cerberusnet_amd.nnet_models import PWCNetHead
from cerberusnet_amd.synth import fill_parameters, hash_uniform

CHANS = [8, 12, 16, 24]
SIZES = [(4, 6), (8, 12), (16, 24), (32, 48)]
PAIRS = 4


def build():
    head = PWCNetHead(CHANS, flow_est_network=dict(type="FlowEstimatorLite", args={}),
                      correlation_backend="torch")
    fill_parameters(head, 2000)
    return head.train()


def batch(indices):
    p1 = [torch.from_numpy(hash_uniform((PAIRS, c, h, w), 300 + l))[indices]
          for l, ((h, w), c) in enumerate(zip(SIZES, reversed(CHANS)))]
    p2 = [torch.from_numpy(hash_uniform((PAIRS, c, h, w), 400 + l))[indices]
          for l, ((h, w), c) in enumerate(zip(SIZES, reversed(CHANS)))]
    return p1, p2


def step_loss(head, p1, p2):
    """Both flow directions through the same head (cerberus.py:131,135), mean over pairs."""
    fw = head((None, p1), (None, p2))
    bw = head((None, p2), (None, p1))
    return sum((f * f).mean() for f in fw) + sum((f * f).mean() for f in bw)


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    assert cdist.init_from_env("gloo") == (rank, world)
    head = cdist.wrap_ddp(build())
    mine = cdist.shard_pairs(PAIRS, rank, world)
    p1, p2 = batch(mine)
    step_loss(head, p1, p2).backward()
    grads = [p.grad.clone() for p in head.parameters()]
    # every rank holds the same averaged gradient
    for g in grads:
        ref = g.clone()
        dist.broadcast(ref, src=0)
        assert torch.equal(ref, g)
    if rank == 0:
        torch.save({"grads": grads, "keys": list(cdist.rank0_state_dict(head).keys())}, out)
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_ddp_gradient_equals_single_process(tmp_path):
    out = str(tmp_path / "ddp.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    torch.set_num_threads(4)
    head = build()
    p1, p2 = batch(list(range(PAIRS)))
    step_loss(head, p1, p2).backward()
    want = [p.grad for p in head.parameters()]
    assert got["keys"] == list(head.state_dict().keys())  # no 'module.' prefix in checkpoints
    assert len(got["grads"]) == len(want)
    for a, b in zip(got["grads"], want):
        scale = float(b.abs().max()) + 1e-12
        assert float((a - b).abs().max()) <= 2e-5 * scale


def test_shard_pairs_partitions_the_batch():
    for world in (1, 2, 3, 8):
        seen = sorted(i for r in range(world) for i in cdist.shard_pairs(11, r, world))
        assert seen == list(range(11))
    assert cdist.init_from_env() == (0, 1)  # no env -> single process, no group


def _exchange_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    assert cdist.init_from_env("gloo") == (rank, world)
    ex = cdist.GradientExchange(10_000, torch.device("cpu"), bucket_mb=0.01)
    assert len(ex.buckets) == 4 and ex.nbytes == 40_000          # 2621 floats per bucket
    ex.flat.copy_(torch.from_numpy(hash_uniform((10_000,), 900 + rank)))
    ex.start()
    ex.finish()
    if rank == 0:
        torch.save(ex.flat.clone(), out)
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_exchange_averages_buckets_over_ranks(tmp_path):
    """The stand-alone bucketed exchange bench.py puts in its timed loop (gloo, world 2): every
    bucket ends up holding the mean over ranks."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "flat.pt")
    mp.spawn(_exchange_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    want = (torch.from_numpy(hash_uniform((10_000,), 900)) +
            torch.from_numpy(hash_uniform((10_000,), 901))) / 2
    assert torch.equal(got, want)
    assert cdist.GradientExchange.bus_bandwidth(100, 1.0, 8) == 175.0
