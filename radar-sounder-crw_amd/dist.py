"""Data-parallel plumbing for the training step: independent radargram sequences are sharded over
ranks (one process per GPU), and the only exchange is ONE all-reduce (mean) of the flat encoder
gradient per step -- RCCL over xGMI when the backend is "nccl" (SURVEY.md section 8(e)).

The reference instead wraps the encoder in ``torch.nn.DataParallel`` (scripts/train.py:45-47) and
runs the whole walk on GPU 0; that pattern is not reproduced.
"""
import os

import torch
import torch.distributed as dist

# which collective the last `FlatGradBucket.all_reduce_mean` issued ("nccl:avg", "gloo:sum/div" or None) -- diagnostics / tests
LAST_COLLECTIVE = None


def init_from_env(backend=None):
    """-> (rank, world_size, local_rank).  Initialises torch.distributed when launched by
    torchrun / torch.distributed.run (WORLD_SIZE > 1); single process otherwise."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal of the N > 1 control flow on a ONE-GPU box (tests / tools only): every rank on device 0, exchange over gloo
    # (RCCL refuses two ranks on one device).  CRW_DIST_REHEARSAL=1 python -m torch.distributed.run --nproc-per-node 2 bench.py ...
    if os.environ.get("CRW_DIST_REHEARSAL"):
        # never silently: leaked into a real multi-GPU launch this would pile every rank onto device 0 over gloo
        if world > 1 and torch.cuda.is_available() and torch.cuda.device_count() >= world and os.environ["CRW_DIST_REHEARSAL"] != "force":
            raise RuntimeError(f"CRW_DIST_REHEARSAL is set but this node has {torch.cuda.device_count()} GPUs for {world} ranks: "
                               "unset it for a real run (or set CRW_DIST_REHEARSAL=force)")
        if rank == 0:
            print("[dist] CRW_DIST_REHEARSAL: every rank on GPU 0, exchange over gloo (control-flow rehearsal, not a measurement)",
                  flush=True)
        backend, local = "gloo", 0
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_indices(n_items, rank, world):
    """rank r takes items r, r+world, ... (equal shard sizes: the tail that does not divide is dropped
    so that mean-of-rank-means equals the global mean)."""
    per = n_items // world
    return [rank + i * world for i in range(per)]


class FlatGradBucket:
    """All parameters' gradients live in one flat fp32 buffer, so the data-parallel exchange is a single
    collective on one contiguous 1.05 MB (CNN) / 19.9 MB (Resnet) message instead of one per tensor.

    ``lazy=False``: ``p.grad`` are permanent views of the buffer; ``zero()`` clears it and autograd ADDS into
    the views (one small add kernel per parameter and step).
    ``lazy=True`` (bench.py / train.py): ``zero()`` drops the gradients (``p.grad = None``), autograd then hands
    over its freshly computed tensors without an add, and ``all_reduce_mean()`` gathers them into the buffer
    with ONE concatenation kernel, reduces, and re-binds ``p.grad`` to views of the reduced buffer for the
    optimizer -- 1 kernel per step instead of a fill plus one add per parameter."""

    def __init__(self, params, lazy=False):
        self.params = [p for p in params if p.requires_grad]
        self.lazy = lazy
        n = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(n, dtype=ref.dtype, device=ref.device)
        self._views = []
        o = 0
        for p in self.params:
            self._views.append(self.flat[o:o + p.numel()].view_as(p))
            o += p.numel()
        self._bind()

    def _bind(self):
        for p, v in zip(self.params, self._views):
            p.grad = v

    def zero(self):
        if self.lazy:
            for p in self.params:
                p.grad = None
        else:
            self.flat.zero_()

    def all_reduce_mean(self):
        if self.lazy:
            grads = [(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in self.params]
            if any(g.data_ptr() != v.data_ptr() for g, v in zip(grads, self._views)):  # not already bound
                torch.cat(grads, out=self.flat)
            self._bind()
        global LAST_COLLECTIVE
        LAST_COLLECTIVE = None
        if dist.is_initialized():  # also with one rank: the collective then is the identity, and is exercised
            if dist.get_backend() == "nccl":
                dist.all_reduce(self.flat, op=dist.ReduceOp.AVG)
                LAST_COLLECTIVE = "nccl:avg"
            else:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
                self.flat.div_(dist.get_world_size())
                LAST_COLLECTIVE = "gloo:sum/div"
        return self.flat
