"""One process per GPU; RCCL (torch.distributed backend "nccl") over xGMI for the only
exchange the path has: the sum of per-shard partial reductions (``w = sum_shards Z^T Z p``
per CG iteration; ``acc``, ``Z^T y``, ``y^T y`` once per preconditioner pass).  The reference
is single-device (docs/FAQ.rst:12-15); datapoints are independent and every consumer is a
``for chunk: acc += f(chunk)`` reduction (fitting_toolkit/cg_tools.py:189-191,
preconditioners/rand_nys_constructors.py:115-119), so rows are sharded contiguously and the
replicated CG state stays identical on every rank without further communication.

On a CPU-only box the same code runs over the ``gloo`` backend (tests/test_dist_cpu.py).
"""
import os

import torch
import torch.distributed as dist


class Comm:
    """Rank / world-size holder with a sum all-reduce that is a no-op for one rank."""

    def __init__(self, rank=0, world_size=1, group=None, through_backend=False):
        self.rank, self.world_size, self.group = rank, world_size, group
        # one-rank rehearsal of the collective path (XGPR_DIST_FORCE=1): the calls go to the backend even though
        # there is nobody to exchange with, so that communicator set-up and stream ordering are exercised on one GPU
        self.through_backend = through_backend or world_size > 1

    @property
    def is_distributed(self):
        return self.world_size > 1

    def all_reduce_(self, tensor):
        """In-place sum over ranks (RCCL ring / tree over xGMI on GPUs)."""
        if self.through_backend:
            dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=self.group)
        return tensor

    def all_reduce_max_(self, tensor):
        """In-place maximum over ranks (dataset statistics only)."""
        if self.through_backend:
            dist.all_reduce(tensor, op=dist.ReduceOp.MAX, group=self.group)
        return tensor

    def barrier(self):
        if self.through_backend:
            dist.barrier(group=self.group)

    def shard_bounds(self, n):
        """Contiguous row range [lo, hi) of this rank for n datapoints (remainder spread
        over the first ranks)."""
        base, rem = divmod(n, self.world_size)
        lo = self.rank * base + min(self.rank, rem)
        return lo, lo + base + (1 if self.rank < rem else 0)


SINGLE = Comm()


def init_from_env(device_type="cuda"):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (as set by
    ``python -m torch.distributed.run``) and bind this process to its GPU."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal aids for a one-GPU box (several ranks sharing device 0 over gloo, which stages CUDA tensors
    # through the host): XGPR_LOCAL_DEVICE pins the device index, XGPR_DIST_BACKEND picks the backend.
    # Production runs set neither: one GPU per rank, RCCL.
    if "XGPR_LOCAL_DEVICE" in os.environ:
        local = int(os.environ["XGPR_LOCAL_DEVICE"])
    if device_type == "cuda":
        torch.cuda.set_device(local)
    force = os.environ.get("XGPR_DIST_FORCE", "") == "1"
    if world == 1 and not force:
        return Comm()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if not dist.is_initialized():
        backend = os.environ.get("XGPR_DIST_BACKEND", "nccl" if device_type == "cuda" else "gloo")
        kwargs = {}
        if device_type == "cuda" and backend == "nccl":
            kwargs["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
    return Comm(rank, world, through_backend=force)
