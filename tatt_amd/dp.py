"""Data-parallel plumbing (device-agnostic torch code; the collectives run on RCCL over xGMI on the GPU box and on
gloo in the CPU tests).

Replaces torch.nn.DataParallel of the reference (interfaces/base.py:386-396): one process per GPU, each with a
full replica; per-replica BatchNorm statistics and query-GRU (same semantics as DataParallel replicas);
gradients are exchanged with sum all-reduces of contiguous BUCKETS of one flat fp32 buffer (7,608,334 parameters +
alignment padding = 30.4 MB for TATT), the 1/world factor is folded into the optimiser kernel, THEN the global-norm clip and
Adam run identically on every rank.  Parameters that never receive a gradient (14 tensors of the reference)
contribute zeros on all ranks.

Buckets follow the order in which gradients complete during the backward pass: trunk + up-sampler
(block8 ... block2) first, then the TP interpreter (its 4.7 M-parameter query GRU is the bulk of the bytes), block1 + STN head
last.  `GradCuts` lets the backward run in as many stages, so that bucket k travels over xGMI while stage k+1 computes.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


class FlatParams:
    """Re-homes parameters into one flat buffer (bucket after bucket) and owns a second flat buffer for their gradients.
    Every parameter starts on a 64-byte boundary (ALIGN floats): the kernels move weights with 16-byte vector loads, and a
    weight that follows an odd-sized tensor (a 37-wide bias, a single PReLU slope) would otherwise lose its aligned fast paths.
    The padding elements are zero in every buffer, so norms, Adam and the all-reduces are unaffected.

    `buckets`: [(name, [parameters])] (default: one bucket with `model.parameters()`); `ranges[k]` = (start, end) of bucket k."""

    ALIGN = 16

    def __init__(self, model: Optional[torch.nn.Module] = None, buckets: Optional[Sequence[Tuple[str, list]]] = None):
        if buckets is None:
            buckets = [("all", [p for p in model.parameters()])]
        self.bucket_names = [n for n, _ in buckets]
        self.bucket_params = [list(ps) for _, ps in buckets]
        self.params = [p for ps in self.bucket_params for p in ps]
        assert len({id(p) for p in self.params}) == len(self.params), "a parameter appears in two buckets"
        dev = self.params[0].device
        a = self.ALIGN
        self.n = sum((p.numel() + a - 1) // a * a for p in self.params)
        self.p = torch.zeros(self.n, device=dev, dtype=torch.float32)
        self.g = torch.zeros(self.n, device=dev, dtype=torch.float32)
        self.offsets = {}
        self.ranges: List[Tuple[int, int]] = []
        off = 0
        with torch.no_grad():
            for ps in self.bucket_params:
                start = off
                for p in ps:
                    k = p.numel()
                    self.p[off:off + k].copy_(p.reshape(-1))
                    p.data = self.p[off:off + k].view_as(p)
                    p._tatt_in_flat = True                   # (raw-pointer kernels update it: its packed filter layouts are refreshed, ops.PACKED)
                    p.grad = None
                    self.offsets[id(p)] = (off, k)
                    off += (k + a - 1) // a * a
                self.ranges.append((start, off))

    def zero_grad(self):
        self.g.zero_()

    def grad_view(self, p):
        off, k = self.offsets[id(p)]
        return self.g[off:off + k].view_as(p)

    def span(self, params) -> Tuple[int, int]:
        """(start, end) of the smallest flat range covering `params`; asserts it covers nothing else (contiguity)."""
        ids = {id(p) for p in params}
        offs = [self.offsets[i] for i in ids]
        start = min(o for o, _ in offs)
        end = max((o + k + self.ALIGN - 1) // self.ALIGN * self.ALIGN for o, k in offs)
        inside = [p for p in self.params if start <= self.offsets[id(p)][0] < end]
        assert {id(p) for p in inside} == ids, "parameter group is not contiguous in the flat buffer"
        return start, end

    def gather_grads(self, k: int):
        """Bucket k of the flat gradient buffer <- the parameters' fresh `.grad` tensors (zeros where a parameter got none):
        one fill + one multi-tensor copy instead of one accumulate kernel per parameter."""
        if self.g.is_cuda:
            # one launch of the library (descriptor table in the kernel arguments); the alignment padding between parameters is zero
            # since the buffer was allocated and is never written
            from . import ops
            ent = []
            for p in self.bucket_params[k]:
                off, n = self.offsets[id(p)]
                g = p.grad
                if g is not None and (g.dtype != torch.float32 or not g.is_contiguous()):
                    g = g.float().contiguous()
                ent.append((g, off, n))
            ops.gather_grads(self.g, ent)
            return
        s, e = self.ranges[k]
        self.g[s:e].zero_()
        have = [p for p in self.bucket_params[k] if p.grad is not None]
        if have:
            torch._foreach_copy_([self.grad_view(p) for p in have], [p.grad for p in have])


class GradCuts:
    """Runs ONE backward pass in stages.  A forward that supports it calls `cut(stage, t)` on the tensors that connect its
    parts: the part downstream continues on a detached copy, so `loss.backward()` stops there (stage "trunk"); `run(stage)`
    then back-propagates the gradient collected on the copies into the part upstream.  Several copies of one tensor (block1's
    output feeds the TP interpreter and the trunk) are summed.  Numerically identical to the single-pass backward: the same
    kernels run in the same order, only the place where the engine is re-entered changes."""

    def __init__(self):
        self._stages = {}

    def reset(self):
        self._stages = {}

    def cut(self, stage: str, t: torch.Tensor) -> torch.Tensor:
        d = t.detach().requires_grad_(True)
        self._stages.setdefault(stage, {}).setdefault(id(t), (t, []))[1].append(d)
        return d

    def run(self, stage: str):
        outs, grads = [], []
        for orig, copies in self._stages.pop(stage, {}).values():
            gs = [d.grad for d in reversed(copies) if d.grad is not None]   # later consumers first, like the single-pass engine
            g = None
            if gs:
                if len(gs) >= 2 and gs[0].is_cuda and len(gs) <= 8 and gs[0].dtype == torch.float32:
                    from . import ops
                    g = ops.add_n(gs)                # one launch, same left-to-right order as the chain of adds below
                else:
                    g = gs[0]
                    for t in gs[1:]:
                        g = g + t
            if g is not None and orig.requires_grad:
                outs.append(orig)
                grads.append(g)
        if outs:
            torch.autograd.backward(outs, grads)


def broadcast_model(flat: FlatParams, model: torch.nn.Module, group=None, src: int = 0, extra: Sequence[torch.Tensor] = ()):
    """Ranks start from rank `src`'s weights and buffers (DataParallel re-replicates every forward instead).  `extra`: tensors
    outside `model.parameters()/buffers()` that must agree as well (a frozen teacher's weights)."""
    dist.broadcast(flat.p, src, group=group)
    for b in list(model.buffers()) + list(extra):
        if b.is_contiguous():
            dist.broadcast(b, src, group=group)
        else:                                        # e.g. tps.inverse_kernel: torch.inverse hands back column-major strides
            t = b.contiguous()
            dist.broadcast(t, src, group=group)
            b.copy_(t)
    if flat.p.is_cuda:
        # the weights changed behind torch's version counters (writes through the flat buffer): packed filter copies are stale
        from . import ops
        ops.PACKED.refresh(flat.p.device)


def allreduce_bucket(flat: FlatParams, k: int, group=None, async_op: bool = False):
    """Sum all-reduce of bucket k of the flat gradient buffer.  With `async_op` the collective is enqueued behind the work
    already issued on the current stream (RCCL runs it on its own stream) and the returned handle's `wait()` makes the current
    stream wait for it -- the later backward stages overlap the transfer."""
    s, e = flat.ranges[k]
    return dist.all_reduce(flat.g[s:e], op=dist.ReduceOp.SUM, group=group, async_op=async_op)


def allreduce_grads(flat: FlatParams, group=None):
    """Sum all-reduce of the whole flat gradient buffer (all buckets, one after the other)."""
    for k in range(len(flat.ranges)):
        allreduce_bucket(flat, k, group)


def rank_seed(base_seed: int, rank: int) -> int:
    """Per-rank data seed of the synthetic benchmark: rank r draws Generator(seed=r)."""
    return base_seed + rank


def rank_dropout_seed(base: int, rank: int) -> int:
    """Dropout seed word of rank `rank`: replicas draw independent masks, like DataParallel replicas of the reference (each
    replica's nn.Dropout consumes its own device's generator)."""
    z = (base ^ (0x9E3779B97F4A7C15 * (rank + 1))) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return (z ^ (z >> 31)) & 0x7FFFFFFFFFFFFFFF
