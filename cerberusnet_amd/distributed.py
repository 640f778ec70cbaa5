"""Data-parallel wiring for the flow head (the reference has none:
``utilities/model_trainer.py:57`` is single-GPU).

One process per GPU, ``torch.distributed`` backend ``"nccl"`` (= RCCL on ROCm) over the
node's xGMI mesh; ``"gloo"`` for the CPU wiring tests.  The correlation / warp ops have no
parameters and never mix batch items, so sharding image pairs over ranks needs no
collective in the op itself; the one exchange per step is the gradient all-reduce of the
head's convolution weights, which DDP buckets and overlaps with backward.

xGMI is a point-to-point mesh (7 links x ~153 GB/s per GPU): a ring all-reduce is bound by
one link, so buckets are kept large (few, big collectives) rather than NVSwitch-small.
"""
import os
from typing import Optional

import torch
import torch.distributed as dist
from torch.nn.parallel import DistributedDataParallel


def init_from_env(backend: Optional[str] = None, device: Optional[torch.device] = None):
    """Rendezvous from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun sets them).
    Returns (rank, world_size).  No-op when WORLD_SIZE is 1 or unset."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world == 1:
        return 0, 1
    if not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kwargs = {}
        if backend == "nccl" and device is not None:
            kwargs["device_id"] = device
        dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
    return rank, world


def shard_pairs(n_pairs: int, rank: int, world: int):
    """Indices of the image pairs rank `rank` owns (DistributedSampler-style striding)."""
    return list(range(rank, n_pairs, world))


def wrap_ddp(module: torch.nn.Module, device: Optional[torch.device] = None,
             bucket_cap_mb: int = 64) -> torch.nn.Module:
    """DDP around a flow head.  The head is called twice per step (1->2 and 2->1,
    cerberus.py:131,135): its parameters are used twice in one autograd graph, every
    parameter always receives a gradient, and the graph is the same every step ->
    ``static_graph=True``, ``find_unused_parameters=False``."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return module
    ids = [device.index] if device is not None and device.type == "cuda" else None
    return DistributedDataParallel(module, device_ids=ids, bucket_cap_mb=bucket_cap_mb,
                                   gradient_as_bucket_view=True, static_graph=True,
                                   find_unused_parameters=False)


def rank0_state_dict(module: torch.nn.Module):
    """Checkpoint contents without the ``module.`` prefix DDP adds, so files stay loadable
    by the reference (model_trainer.py:106-119 saves ``model_state_dict``)."""
    inner = module.module if isinstance(module, DistributedDataParallel) else module
    return inner.state_dict()
