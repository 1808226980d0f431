"""Data-parallel training of NlosPose: one process per GPU, gradients averaged with
bucketed all-reduce over RCCL/xGMI (torch.distributed backend "nccl" on ROCm).

The reference is single-GPU (a commented-out nn.DataParallel at train.py:111); the path
shards by SAMPLE only: every stage is per-sample except BatchNorm batch
statistics (kept per device, as on the reference's single device) and the batch-global
Dice term, which `all_reduce_dice_terms` makes exact across ranks with one 3-scalar
all-reduce.

Design for xGMI (point-to-point links, ring collectives are per-link bound): few large
buckets in reverse registration order -- posenet layer4 + head hold ~80 % of the 353 MB
and finish first in backward -- each all-reduced asynchronously on RCCL's own stream as
soon as its last gradient has been accumulated, so the transfer hides under the rest of
backward.  Gradients live as views into the flat bucket buffers: no pack/unpack copies.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


class GradBucketReducer:
    def __init__(self, module: torch.nn.Module, bucket_mb: float = 64.0, group: Optional[dist.ProcessGroup] = None,
                 broadcast_from: int = 0, force_collectives: bool = False):
        self.module = module
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # force_collectives: issue the (trivial) collectives even at world size 1, to rehearse the code path
        self._active = self.world > 1 or (force_collectives and dist.is_initialized())
        params = [p for p in module.parameters() if p.requires_grad]
        if self._active:
            with torch.no_grad():
                for t in list(module.parameters()) + list(module.buffers()):
                    dist.broadcast(t.data, src=broadcast_from, group=group)
        # buckets in reverse registration order ~ the order autograd produces gradients
        cap = int(bucket_mb * (1 << 20))
        self.buckets: List[List[torch.nn.Parameter]] = [[]]
        size = 0
        for p in reversed(params):
            nbytes = p.numel() * p.element_size()
            if self.buckets[-1] and size + nbytes > cap:
                self.buckets.append([])
                size = 0
            self.buckets[-1].append(p)
            size += nbytes
        self.flat: List[torch.Tensor] = []
        self._bucket_of = {}
        for bi, bucket in enumerate(self.buckets):
            n = sum(p.numel() for p in bucket)
            flat = torch.zeros(n, dtype=bucket[0].dtype, device=bucket[0].device)
            o = 0
            for p in bucket:
                p.grad = flat[o:o + p.numel()].view_as(p)
                o += p.numel()
                self._bucket_of[p] = bi
            self.flat.append(flat)
        self._pending = [0] * len(self.buckets)
        self._handles = []
        self._hooks = []
        # gradients are accumulated into the bucket views on the main stream as soon as autograd gets them: the
        # opt-in side-stream weight gradient (hip_ops.set_wgrad_async) would be read before it is written
        from . import hip_ops
        hip_ops.set_wgrad_async(False)
        if self._active:
            for p in params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self.begin_step()

    # -- per-step protocol: zero_grad() ; forward ; backward ; finish() ; optimizer.step()
    def zero_grad(self) -> None:
        for f in self.flat:
            f.zero_()
        self.begin_step()

    def begin_step(self) -> None:
        self._pending = [len(b) for b in self.buckets]
        self._handles = []

    def _on_grad(self, p: torch.nn.Parameter) -> None:
        bi = self._bucket_of[p]
        self._pending[bi] -= 1
        if self._pending[bi] == 0:
            self._launch(bi)

    def _launch(self, bi: int) -> None:
        flat = self.flat[bi]
        flat.div_(self.world)
        self._handles.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self) -> None:
        """Wait for every bucket (buckets whose parameters got no gradient this step are
        reduced here so that all ranks issue the same collectives)."""
        if self._active:
            for bi, left in enumerate(self._pending):
                if left > 0:
                    self._pending[bi] = 0
                    self._launch(bi)
            for h in self._handles:
                h.wait()
        self._handles = []

    def remove_hooks(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks = []


def all_reduce_dice_terms(inter: torch.Tensor, psum: torch.Tensor, tsum: torch.Tensor, group=None):
    """Make the batch-global Dice score (utils/criterion.py:358-368) exact under data
    parallelism: sums of sigma(x)t, sigma(x) and t over ALL ranks' samples."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        v = torch.stack([inter, psum, tsum])
        dist.all_reduce(v, group=group)
        return v[0], v[1], v[2]
    return inter, psum, tsum
