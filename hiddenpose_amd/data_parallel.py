"""Data-parallel training of NlosPose: one process per GPU, gradients averaged with
bucketed collectives over RCCL/xGMI (torch.distributed backend "nccl" on ROCm).

The reference is single-GPU (a commented-out nn.DataParallel at train.py:111); the path
shards by SAMPLE only: every stage is per-sample except BatchNorm batch
statistics (kept per device, as on the reference's single device) and the batch-global
Dice term (utils/criterion.py:358-368), which `criterion.BCEDiceLoss(global_batch=True)`
makes exact across ranks with one 3-scalar all-reduce inside the loss (forward) and no
collective at all in its backward (hip_ops._BceDiceGlobal).

Design for xGMI (point-to-point links, ring collectives are per-link bound): few large
buckets in reverse registration order -- posenet layer4 + head hold ~80 % of the 353 MB
and finish first in backward -- each reduced on a dedicated communication stream as soon
as its last gradient has been accumulated, so the transfer hides under the rest of
backward.  Gradients live as views into the flat bucket buffers: no pack/unpack copies.

Three exchange algorithms (`algo`), to be A/B-ed on the first 8-GPU run:
  "all_reduce"  one RCCL all-reduce per bucket (RCCL picks ring/tree);
  "rs_ag"       reduce_scatter_tensor + all_gather_into_tensor per bucket;
  "a2a"         direct reduce-scatter: all_to_all_single puts shard j of every rank on rank j over
                its own xGMI link (all 7 links busy, 1/8 of the bucket per link instead of the
                ring's 7/8 through one), local sum, then all_gather_into_tensor.
`wire_dtype=torch.bfloat16` sends bf16 on the wire (BASELINE configs[2]: 176 MB instead of
353 MB per step); the buckets themselves and the sum handed to Adam stay fp32.

Side-stream weight gradients (hip_ops.set_wgrad_async): those gradients are accumulated into
the bucket views on the side stream and reported through hip_ops._side_grad_listeners; a
bucket's collective is queued behind BOTH streams, so the overlap mode stays on under DP.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist

ALGOS = ("all_reduce", "rs_ag", "a2a")


class GradBucketReducer:
    def __init__(self, module: torch.nn.Module, bucket_mb: float = 64.0, group: Optional[dist.ProcessGroup] = None,
                 broadcast_from: int = 0, force_collectives: bool = False, algo: str = "all_reduce",
                 wire_dtype: Optional[torch.dtype] = None):
        if algo not in ALGOS:
            raise ValueError(f"algo must be one of {ALGOS}, got {algo!r}")
        self.module = module
        self.group = group
        self.algo = algo
        self.wire_dtype = wire_dtype
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
            n_pad = -(-n // self.world) * self.world          # rs_ag / a2a split a bucket into `world` equal shards
            flat = torch.zeros(n_pad, dtype=bucket[0].dtype, device=bucket[0].device)
            o = 0
            for p in bucket:
                p.grad = flat[o:o + p.numel()].view_as(p)
                o += p.numel()
                self._bucket_of[p] = bi
            self.flat.append(flat)
        # persistent staging per bucket (allocated on first use, on the communication stream, and kept): the wire image
        # of the bucket, the owned shard and -- for "a2a" -- the receive buffer.  Nothing is allocated per step.
        self._wire: List[Optional[torch.Tensor]] = [None] * len(self.buckets)
        self._shard: List[Optional[torch.Tensor]] = [None] * len(self.buckets)
        self._recv: List[Optional[torch.Tensor]] = [None] * len(self.buckets)
        self._pending = [0] * len(self.buckets)
        self._arrived = set()
        self._hooks = []
        self._cuda = self.flat[0].is_cuda
        self._comm_stream = torch.cuda.Stream(self.flat[0].device) if (self._cuda and self._active) else None
        self._side = None
        # Optional instrumentation (enable_timing): per bucket, a pair of events on the communication stream around its
        # exchange; per step, a pair on the training stream around finish()'s wait -- the communication time that backward did
        # NOT hide ("exposed").  Off by default: nothing is recorded.
        self._timing = False
        self._t_bucket = [[] for _ in self.buckets]
        self._t_exposed = []
        self._t_host = [[] for _ in self.buckets]     # CPU tensors (gloo): host seconds per exchange
        if self._active:
            for p in params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
            from . import hip_ops

            hip_ops._side_grad_listeners.append(self._on_side_grad)
        self.begin_step()

    # -- per-step protocol: zero_grad() ; forward ; backward ; finish() ; optimizer.step()
    def zero_grad(self) -> None:
        for f in self.flat:
            f.zero_()
        self.begin_step()

    def begin_step(self) -> None:
        self._pending = [len(b) for b in self.buckets]
        self._arrived = set()
        self._side = None

    def _on_grad(self, p: torch.nn.Parameter) -> None:
        if p in self._arrived or p not in self._bucket_of:
            return
        self._arrived.add(p)
        bi = self._bucket_of[p]
        self._pending[bi] -= 1
        if self._pending[bi] == 0:
            self._launch(bi)

    def _on_side_grad(self, p: torch.nn.Parameter, side_stream) -> None:
        """A weight gradient was accumulated into its bucket view on hip_ops' side stream."""
        self._side = side_stream
        self._on_grad(p)

    def _exchange(self, bi: int) -> None:
        """Average bucket `bi` over the group, in place; runs on the current (communication) stream.  The staging
        buffers are only ever touched on that stream, bucket by bucket in launch order, so they can be reused every step."""
        flat = self.flat[bi]
        flat.div_(self.world)
        buf = flat
        if self.wire_dtype is not None and self.wire_dtype != flat.dtype:
            if self._wire[bi] is None:
                self._wire[bi] = torch.empty(flat.numel(), dtype=self.wire_dtype, device=flat.device)
            buf = self._wire[bi]
            buf.copy_(flat)
        if self.algo == "all_reduce":
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
        else:
            w = self.world
            if self._shard[bi] is None:
                self._shard[bi] = torch.empty(buf.numel() // w, dtype=buf.dtype, device=buf.device)
            shard = self._shard[bi]
            if self.algo == "rs_ag":
                dist.reduce_scatter_tensor(shard, buf, op=dist.ReduceOp.SUM, group=self.group)
            else:
                if self._recv[bi] is None:
                    self._recv[bi] = torch.empty_like(buf)
                recv = self._recv[bi]
                dist.all_to_all_single(recv, buf, group=self.group)
                torch.sum(recv.view(w, -1), dim=0, out=shard)
            dist.all_gather_into_tensor(buf, shard, group=self.group)
        if buf is not flat:
            flat.copy_(buf)

    def _launch(self, bi: int) -> None:
        flat = self.flat[bi]
        if self._comm_stream is None:
            if self._timing:
                import time
                t0 = time.perf_counter()
                self._exchange(bi)
                self._t_host[bi].append(time.perf_counter() - t0)
            else:
                self._exchange(bi)
            return
        cs = self._comm_stream
        cs.wait_stream(torch.cuda.current_stream(flat.device))
        if self._side is not None:
            cs.wait_stream(self._side)
        with torch.cuda.stream(cs):
            if self._timing:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(cs)
                self._exchange(bi)
                b.record(cs)
                self._t_bucket[bi].append((a, b))
            else:
                self._exchange(bi)

    # -- instrumentation
    def enable_timing(self, on: bool = True) -> None:
        self._timing = bool(on) and self._active
        self._t_bucket = [[] for _ in self.buckets]
        self._t_exposed = []
        self._t_host = [[] for _ in self.buckets]

    def read_timing(self) -> dict:
        """Means over the steps since enable_timing(): per-bucket exchange time on the communication stream (cast to the wire
        dtype + collective(s) + cast back), the bucket sizes, and the exposed communication per step (how long the training
        stream waited in finish()).  Synchronises the device."""
        if self._cuda:
            torch.cuda.synchronize(self.flat[0].device)
            per_bucket = [round(sum(a.elapsed_time(b) for a, b in ev) / len(ev), 4) if ev else None for ev in self._t_bucket]
            exposed = [a.elapsed_time(b) for a, b in self._t_exposed]
        else:
            per_bucket = [round(1e3 * sum(v) / len(v), 4) if v else None for v in self._t_host]
            exposed = [1e3 * v for v in self._t_exposed]
        wire_b = 2 if self.wire_dtype == torch.bfloat16 else self.flat[0].element_size()
        return {"algo": self.algo, "wire_dtype": str(self.wire_dtype or self.flat[0].dtype).replace("torch.", ""),
                "bucket_mb": [round(f.numel() * wire_b / 2**20, 2) for f in self.flat],
                "bucket_exchange_ms": per_bucket,
                "exchange_ms_per_step": round(sum(v for v in per_bucket if v is not None), 4),
                "exposed_ms_per_step": round(sum(exposed) / len(exposed), 4) if exposed else None,
                "steps_timed": len(exposed)}

    def finish(self) -> None:
        """Wait for every bucket (buckets whose parameters got no gradient this step are
        reduced here so that all ranks issue the same collectives)."""
        if self._active:
            for bi, left in enumerate(self._pending):
                if left > 0:
                    self._pending[bi] = 0
                    self._launch(bi)
            if self._comm_stream is not None:
                main = torch.cuda.current_stream(self.flat[0].device)
                if self._timing:
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(main)
                    main.wait_stream(self._comm_stream)
                    b.record(main)
                    self._t_exposed.append((a, b))
                else:
                    main.wait_stream(self._comm_stream)
            elif self._timing:
                self._t_exposed.append(0.0)      # synchronous collectives (CPU tensors): all of it is exposed, see bucket times

    def remove_hooks(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks = []
        from . import hip_ops

        if self._on_side_grad in hip_ops._side_grad_listeners:
            hip_ops._side_grad_listeners.remove(self._on_side_grad)


def all_reduce_dice_terms(inter: torch.Tensor, psum: torch.Tensor, tsum: torch.Tensor, group=None):
    """Sums of sigma(x)t, sigma(x) and t over ALL ranks' samples (the batch-global Dice score of
    utils/criterion.py:358-368).  Plain (non-differentiable) sums: hip_ops._BceDiceGlobal, which is what the training
    path uses, derives its gradient from the global sums analytically and needs no collective in backward."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        v = torch.stack([inter, psum, tsum])
        dist.all_reduce(v, group=group)
        return v[0], v[1], v[2]
    return inter, psum, tsum
