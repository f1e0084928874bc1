"""Data parallelism over the GPUs of one node: one process per GPU, ``torch.distributed`` with the
``nccl`` backend (= RCCL over xGMI on ROCm); ``gloo`` on CPU for the tests.

Crystals are independent units (no edge crosses graphs), so the batch shards with NO data-path
collective; the only exchanges per step are
  1. phonon loss: all-reduce of the two SSE scalars before backward — the reference's loss is ONE
     rmse over all B*51 elements (`main_phDOS.py:109-114`), which does not decompose over shards;
  2. one all-reduce (sum) of the flat fp32 gradient buffer.
Shards must be padded to the GLOBAL ``n_max``: the reference attends over zero-padded atoms, so
``Nmax`` changes the numerics.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import torch
import torch.distributed as td

from .batch import CrystalBatch, collate


def shard_bounds(n_edges: Sequence[int], world: int) -> List[Tuple[int, int]]:
    """Contiguous shards of the crystal list, balanced by edge count (the dominant cost)."""
    n = len(n_edges)
    if world <= 0:
        raise ValueError("world must be positive")
    total = float(sum(n_edges))
    bounds, start, acc = [], 0, 0.0
    for r in range(world):
        remaining_ranks = world - r
        if r == world - 1:
            end = n
        else:
            target = (total - acc) / remaining_ranks
            end, s = start, 0.0
            # leave at least one crystal for each remaining rank when possible
            max_end = n - (remaining_ranks - 1) if n >= world else n
            while end < max_end and (s + n_edges[end] / 2.0 <= target or end == start):
                s += n_edges[end]
                end += 1
            acc += s
        bounds.append((start, end))
        start = end
    return bounds


def shard_batch(crystals: Sequence[Dict[str, object]], world: int, rank: int, sort_edges: bool = True) -> CrystalBatch:
    """This rank's shard of a global batch, padded to the global ``n_max``."""
    n_max = max(int(c["x"].shape[0]) for c in crystals)
    lo, hi = shard_bounds([int(c["edge_index"].shape[1]) for c in crystals], world)[rank]
    if hi <= lo:
        raise ValueError(f"rank {rank} of {world} got an empty shard ({len(crystals)} crystals)")
    g = collate(crystals[lo:hi], sort_edges=sort_edges, n_max=n_max)
    g.n_global = len(crystals)          # train.Trainer reads it instead of running a count collective every step
    return g


class _Done:
    """Handle of a host-staged collective: the sum is done, its copy back to the device is queued on ``stream``."""

    def __init__(self, stream=None):
        self.stream = stream

    def wait(self):
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)
        return True


class DataParallel:
    """The two collectives of a data-parallel step (see module doc).

    Production backend is ``nccl`` (= RCCL over xGMI): collectives run on device buffers, stream-ordered.  With a
    ``gloo`` group (the CPU tests, and the 2-ranks-on-one-GPU harness of tests/test_dp_gpu.py — RCCL refuses two ranks
    on one device) device tensors are staged through the host: same sums, synchronous."""

    def __init__(self, group=None):
        if not td.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.world = td.get_world_size(group)
        self.rank = td.get_rank(group)
        self.staged = td.get_backend(group) != "nccl"

    def _sum(self, t: torch.Tensor) -> None:
        if self.staged and t.is_cuda:
            h = t.detach().cpu()                   # (synchronises with the stream that produced t)
            td.all_reduce(h, op=td.ReduceOp.SUM, group=self.group)
            t.copy_(h)
        else:
            td.all_reduce(t, op=td.ReduceOp.SUM, group=self.group)

    def all_reduce_sse(self, sse: torch.Tensor) -> None:
        """In-place sum over ranks of the phonon loss' two SSE scalars (no host synchronisation; the
        global element count is static: crystals in the un-sharded batch x 51)."""
        self._sum(sse)

    def global_count(self, local: int) -> int:
        """Sum of a per-rank count.  Blocking + host read: callers that know the global batch size pass it instead
        (``dist.shard_batch`` records it on the batch, ``Trainer.step(g, n_global)``)."""
        t = torch.tensor([float(local)], dtype=torch.float64, device="cpu" if self.staged else "cuda")
        td.all_reduce(t, op=td.ReduceOp.SUM, group=self.group)
        return int(round(float(t[0])))

    def min_max(self, value: int) -> Tuple[int, int]:
        """(min, max) over ranks of a per-rank integer.  Blocking + host read: for one-time consistency checks, never per step."""
        t = torch.tensor([float(value), -float(value)], dtype=torch.float64, device="cpu" if self.staged else "cuda")
        td.all_reduce(t, op=td.ReduceOp.MIN, group=self.group)
        return int(round(float(t[0]))), int(round(-float(t[1])))

    def all_reduce_grads(self, flat_grad: torch.Tensor) -> None:
        self._sum(flat_grad)

    def all_reduce_grads_async(self, flat_grad: torch.Tensor):
        """Start the sum of a (contiguous slice of the) flat gradient buffer; ordered after the work already queued on
        the CURRENT stream.  Returns the handle; ``.wait()`` makes the then-current stream wait for the result."""
        if self.staged and flat_grad.is_cuda:
            self._sum(flat_grad)
            return _Done(torch.cuda.current_stream())
        return td.all_reduce(flat_grad, op=td.ReduceOp.SUM, group=self.group, async_op=True)
