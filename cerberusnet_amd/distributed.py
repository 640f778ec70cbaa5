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
             bucket_cap_mb: int = 64, force: bool = False, sync_bn: bool = False) -> torch.nn.Module:
    """DDP around a flow head.  The head is called twice per step (1->2 and 2->1,
    cerberus.py:131,135): its parameters are used twice in one autograd graph, every
    parameter always receives a gradient, and the graph is the same every step ->
    ``static_graph=True``, ``find_unused_parameters=False``.
    ``force=True`` wraps even in a one-rank group: the RCCL communicator, DDP's bucket hooks and the
    static-graph bookkeeping then run for real on a single GPU (the self-test a 1-GPU box can do).
    ``sync_bn=True`` converts the module's BatchNorm layers to ``SyncBatchNorm`` first (the reference's own note,
    ``ocr_utils.py:13-17``: "if you want to do multi-gpu training, this needs a synchronised version"); off by default:
    at 4 pairs per GPU per-replica statistics are the cheaper choice (305 more latency-bound collectives per pass
    otherwise)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
        return module
    if sync_bn:
        module = torch.nn.SyncBatchNorm.convert_sync_batchnorm(module)
    ids = [device.index] if device is not None and device.type == "cuda" else None
    return DistributedDataParallel(module, device_ids=ids, bucket_cap_mb=bucket_cap_mb,
                                   gradient_as_bucket_view=True, static_graph=True,
                                   find_unused_parameters=False)


def rank0_state_dict(module: torch.nn.Module):
    """Checkpoint contents without the ``module.`` prefix DDP adds, so files stay loadable
    by the reference (model_trainer.py:106-119 saves ``model_state_dict``)."""
    inner = module.module if isinstance(module, DistributedDataParallel) else module
    return inner.state_dict()


class GradientExchange:
    """The per-step gradient exchange of data-parallel training as a stand-alone piece: a
    flat gradient buffer, cut into a few large buckets, all-reduced (mean) over the ranks on a
    SIDE stream so that it overlaps whatever the main stream runs (the next kernels of the
    step), joined back before the optimizer would read it.

    This is what ``DistributedDataParallel`` does for a module's parameters (``wrap_ddp``);
    callers that own their step -- a hipGraph replay of the hot path, ``bench.py`` -- use this
    class to put the same traffic on RCCL / xGMI.  The reference has no multi-GPU training
    (``utilities/model_trainer.py:57``); the wiring this stands for is the one the north star
    names (``training_executor.py:26-56`` model construction, ``model_trainer.py:187-226``
    ``loss.backward(); optimizer.step()``).

    Buckets are large on purpose: xGMI is a point-to-point mesh, a ring all-reduce is bound by
    one ~153 GB/s link, so few big collectives beat many NVSwitch-sized ones.
    """

    def __init__(self, numel: int, device: torch.device, bucket_mb: float = 64.0,
                 dtype: torch.dtype = torch.float32):
        self.flat = torch.zeros(numel, dtype=dtype, device=device)
        per = max(1, int(bucket_mb * 2 ** 20) // self.flat.element_size())
        self.buckets = list(self.flat.split(per))
        self.on_gpu = device.type == "cuda"
        self.stream = torch.cuda.Stream(device) if self.on_gpu else None
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        backend = dist.get_backend() if dist.is_initialized() else ""
        # RCCL averages in the collective; gloo has no AVG: sum, then divide
        self._avg = backend == "nccl"

    @property
    def nbytes(self) -> int:
        return self.flat.numel() * self.flat.element_size()

    def start(self):
        """Enqueue the all-reduce of every bucket.  On a GPU it runs on the side stream, behind
        what the main stream has enqueued SO FAR (the producer of the gradients) and beside
        what it enqueues next."""
        if self.world == 1:
            return
        if self.on_gpu:
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                for b in self.buckets:
                    dist.all_reduce(b, op=dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM)
                    if not self._avg:
                        b.div_(self.world)
        else:
            for b in self.buckets:
                dist.all_reduce(b, op=dist.ReduceOp.SUM)
                b.div_(self.world)

    def finish(self):
        """The main stream waits for the exchange (the optimizer step would come next)."""
        if self.world > 1 and self.on_gpu:
            torch.cuda.current_stream().wait_stream(self.stream)

    @staticmethod
    def bus_bandwidth(nbytes: int, seconds: float, world: int) -> float:
        """Ring all-reduce bus bandwidth in bytes/s: algorithmic bytes/s x 2(n-1)/n."""
        return nbytes / seconds * 2.0 * (world - 1) / world if world > 1 else 0.0
