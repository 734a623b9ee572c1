"""One process per GPU; RCCL (torch.distributed backend "nccl") over xGMI for the only
exchange the path has: the sum of per-shard partial reductions (``w = sum_shards Z^T Z p``
per CG iteration; ``acc``, ``Z^T y``, ``y^T y`` once per preconditioner pass).  The reference
is single-device (docs/FAQ.rst:12-15); datapoints are independent and every consumer is a
``for chunk: acc += f(chunk)`` reduction (fitting_toolkit/cg_tools.py:189-191,
preconditioners/rand_nys_constructors.py:115-119), so rows are sharded contiguously and the
replicated CG state stays identical on every rank without further communication.

On a CPU-only box the same code runs over the ``gloo`` backend (tests/test_dist_cpu.py).

Where the sum runs.  ``torch.distributed.all_reduce`` issues the RCCL kernel on ProcessGroupNCCL's own stream and
chains it to the compute stream with an event on each side.  For the float64 sums of this path (64 KiB per CG
iteration: latency-bound) ``Comm`` instead owns a second RCCL communicator, created through the C ABI
(``xgpr_rccl_comm_init``; the 128-byte id travels through torch.distributed's own channel) and calls
``xgpr_allreduce_sum_f64``: ``ncclAllReduce`` enqueued on the CURRENT stream, directly behind the kernel that wrote
the partial sum.  The direct path is OPT-IN (``XGPR_RCCL_DIRECT=1``): until a run with two or more ranks on real
hardware has confirmed it (bench.py reports ``direct_equals_torch_allreduce`` and the final loss when it is on), every
collective stays on torch.distributed.  When requested it is still dropped, on all ranks together, if its set-up or its
self test -- a sum of ones -- fails, and the communicator is destroyed (``Comm.close``, also at interpreter exit).
"""
import atexit
import ctypes
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
        self._rccl = None                 # ncclComm_t of the direct path (enable_direct_rccl)
        self._rccl_lib = None             # the C-ABI library it was created through

    def _direct_requested(self):
        return (self.through_backend and os.environ.get("XGPR_RCCL_DIRECT", "0") == "1"
                and dist.get_backend(self.group) == "nccl")

    @staticmethod
    def _load_rccl_entry_points():
        """The C-ABI library with RCCL resolved (xgpr_rccl_load), or raises."""
        from xgpr_amd import _lib          # (absolute: bench.py --dist-check loads this file by path)
        lib = _lib.load()
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        _lib.check(lib.xgpr_rccl_load((path if os.path.exists(path) else "librccl.so").encode()))
        return lib

    def enable_direct_rccl(self, device):
        """Create this rank's communicator for the on-stream all-reduce (module docstring).  Collective: every rank
        calls it.  Returns True when the direct path is in use."""
        if not self._direct_requested():
            return False
        def all_agree(ok):               # every rank keeps or drops the direct path together
            flag = torch.tensor([ok], dtype=torch.int32, device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
            return int(flag.item()) == 1

        # Every torch.distributed collective below is reached by every rank whatever happened locally (a rank that
        # failed a local step still takes part, with "not ok"), and no rank issues a collective on the NEW communicator
        # before all ranks have agreed that every one of them holds it -- a one-sided failure cannot leave the others waiting.
        lib = None
        try:
            lib = self._load_rccl_entry_points()
        except (RuntimeError, OSError, AttributeError, ImportError):
            lib = None
        if not all_agree(1 if lib is not None else 0):
            return False
        self._rccl_lib = lib
        box = [None]
        if self.rank == 0:
            try:
                ident = ctypes.create_string_buffer(128)
                if lib.xgpr_rccl_unique_id(ctypes.cast(ident, ctypes.c_void_p)) == 0:
                    box = [ident.raw]
            except (RuntimeError, OSError, AttributeError):
                box = [None]
        dist.broadcast_object_list(box, src=0, group=self.group)
        if box[0] is None:
            return False
        ok = 1
        try:
            ident = ctypes.create_string_buffer(box[0], 128)
            handle = ctypes.c_void_p()
            if lib.xgpr_rccl_comm_init(ctypes.cast(ctypes.byref(handle), ctypes.c_void_p), self.world_size,
                                       ctypes.cast(ident, ctypes.c_void_p), self.rank) != 0:
                ok = 0
            else:
                self._rccl = handle
        except (RuntimeError, OSError, AttributeError):
            ok = 0
        if not all_agree(ok):            # somebody has no communicator: nobody uses theirs
            self.close()
            return False
        ok = 1
        try:
            probe = torch.ones(8, dtype=torch.float64, device=device)
            self._direct_sum(probe)
            if probe.is_cuda:
                torch.cuda.synchronize(device)
            if not bool((probe == float(self.world_size)).all()):
                ok = 0
        except (RuntimeError, OSError, AttributeError):
            ok = 0
        if not all_agree(ok):
            self.close()
        else:
            atexit.register(self._abandon)
        return self._rccl is not None

    def close(self):
        """Destroy the direct communicator (if any); the sums go back to torch.distributed.  Local, idempotent.
        The owner calls this at an orderly point: every rank alive, the stream the last all-reduce was queued on
        synchronised (done here for CUDA), BEFORE torch.distributed.destroy_process_group()."""
        handle, self._rccl = self._rccl, None
        if handle is not None:
            try:
                if torch.cuda.is_available():
                    torch.cuda.synchronize()
                self._rccl_lib.xgpr_rccl_comm_destroy(handle)
            except (RuntimeError, OSError, AttributeError):
                pass

    def _abandon(self):
        """Interpreter exit without close(): the handle is dropped, NOT destroyed -- ncclCommDestroy at that point runs
        after destroy_process_group(), possibly after a peer has died, with work of unknown state on the stream, and can
        turn an error exit into a hang; the process is going away and takes the communicator with it."""
        self._rccl = None

    def _direct_sum(self, tensor):
        rc = self._rccl_lib.xgpr_allreduce_sum_f64(self._rccl, ctypes.c_void_p(tensor.data_ptr()), tensor.numel(),
                                                   ctypes.c_void_p(torch.cuda.current_stream(tensor.device).cuda_stream))
        if rc != 0:
            raise RuntimeError("xgpr_allreduce_sum_f64 failed: " + self._rccl_lib.xgpr_last_error().decode())

    @property
    def direct_rccl(self):
        return self._rccl is not None

    @property
    def is_distributed(self):
        return self.world_size > 1

    def all_reduce_(self, tensor):
        """In-place sum over ranks (RCCL ring / tree over xGMI on GPUs)."""
        if self.through_backend:
            if self._rccl is not None and tensor.dtype == torch.float64 and tensor.is_cuda and tensor.is_contiguous():
                self._direct_sum(tensor)
            else:
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


class _stdout_to_stderr:
    """RCCL prints a version banner on the C-level stdout when the first communicator of a process is created; the
    benchmark's contract is ONE JSON line on stdout.  While communicators are set up, file descriptor 1 points at
    stderr, and libc's buffer is flushed before it is put back."""

    def __enter__(self):
        import sys
        sys.stdout.flush()
        self._libc = ctypes.CDLL(None)
        self._libc.fflush(None)
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        self._libc.fflush(None)
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


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
    with _stdout_to_stderr():
        if not dist.is_initialized():
            backend = os.environ.get("XGPR_DIST_BACKEND", "nccl" if device_type == "cuda" else "gloo")
            kwargs = {}
            if device_type == "cuda" and backend == "nccl":
                kwargs["device_id"] = torch.device("cuda", local)
            dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
        comm = Comm(rank, world, through_backend=force)
        if device_type == "cuda":
            # the first collective creates torch's communicator (if init_process_group has not), enable_direct_rccl
            # this package's own
            warm = torch.zeros(1, dtype=torch.float64, device=torch.device("cuda", local))
            dist.all_reduce(warm)
            torch.cuda.synchronize()
            comm.enable_direct_rccl(torch.device("cuda", local))
    return comm
