"""Gradient exchange of the data-parallel step: one process per GPU, torch.distributed (RCCL on
ROCm, gloo in the CPU tests).

The reference hides this in ``MirroredStrategy``: every ``optimizer.apply_gradients`` all-reduces
(SUM -- the losses are already divided by the GLOBAL batch, gan_losses.py:10,16) one gradient set
and ``strategy.reduce(SUM)`` folds 7 scalars (training_step.py:104-134,233-235).  Here each of the
three sets is ONE flat buffer, all-reduced asynchronously as soon as its backward pass has
produced it, so the collective overlaps the next backward pass; waited for right before the
corresponding Adam update.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist


class GradExchange:
    def __init__(self, process_group=None):
        self.pg = process_group
        up = dist.is_available() and dist.is_initialized()
        # TBG_FORCE_EXCHANGE=1: run the whole exchange path (bucketed D backward, split-graph capture, one collective per
        # gradient slice) at world size 1 as well.  A 1-rank SUM all-reduce is the identity, so the step's results must be
        # bit-identical to the non-distributed step -- which is how a single-GPU box can execute ncclAllReduce between the
        # replays of HIP graphs that share a pool before the first multi-GPU run does (tests/test_distributed_gpu.py).
        forced = up and os.environ.get("TBG_FORCE_EXCHANGE", "0") == "1"
        self.active = up and (dist.get_world_size(process_group) > 1 or forced)
        self._pending: List = []
        self.muted = False  # measurement aid (bench.py dist_record): skip the gradient collectives, keep every launch

    def world_size(self) -> int:
        return dist.get_world_size(self.pg) if self.active else 1

    def start(self, flat_grad: torch.Tensor):
        """Asynchronous SUM all-reduce of one flat gradient buffer; returns a handle (or None)."""
        if not self.active or self.muted:
            return None
        return dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)

    @staticmethod
    def finish(handle) -> None:
        if handle is not None:
            handle.wait()

    def reduce_now(self, bufs: Sequence[torch.Tensor]) -> None:
        if self.active and not self.muted:
            for b in bufs:
                dist.all_reduce(b, op=dist.ReduceOp.SUM, group=self.pg)

    def reduce_scalars(self, scalars: Sequence[torch.Tensor]) -> List[torch.Tensor]:
        """strategy.reduce(SUM) of the loss scalars as one collective."""
        if not self.active:
            return list(scalars)
        packed = torch.stack([s.reshape(()) for s in scalars])
        dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=self.pg)
        return list(packed.unbind(0))
