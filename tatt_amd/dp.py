"""Data-parallel plumbing (device-agnostic torch code; the collectives run on RCCL over xGMI on the GPU box and on
gloo in the CPU tests).

Replaces torch.nn.DataParallel of the reference (interfaces/base.py:386-396): one process per GPU, each with a
full replica; per-replica BatchNorm statistics and query-GRU (same semantics as DataParallel replicas);
gradients are exchanged with ONE sum all-reduce of a flat fp32 buffer (7,608,334 parameters + alignment padding = 30.4 MB for
TATT), the
1/world factor is folded into the optimiser kernel, THEN the global-norm clip and Adam run identically on every rank.
Parameters that never receive a gradient (14 tensors of the reference) contribute zeros on all ranks.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class FlatParams:
    """Re-homes a module's parameters into one flat buffer and gives every parameter a `.grad` view into a second one.
    Every parameter starts on a 64-byte boundary (ALIGN floats): the kernels move weights with 16-byte vector loads, and a
    weight that follows an odd-sized tensor (a 37-wide bias, a single PReLU slope) would otherwise lose its aligned fast paths.
    The padding elements are zero in every buffer, so norms, Adam and the all-reduce are unaffected."""

    ALIGN = 16

    def __init__(self, model: torch.nn.Module):
        self.params = [p for p in model.parameters()]
        dev = self.params[0].device
        a = self.ALIGN
        self.n = sum((p.numel() + a - 1) // a * a for p in self.params)
        self.p = torch.zeros(self.n, device=dev, dtype=torch.float32)
        self.g = torch.zeros(self.n, device=dev, dtype=torch.float32)
        self.offsets = {}
        off = 0
        with torch.no_grad():
            for p in self.params:
                k = p.numel()
                self.p[off:off + k].copy_(p.reshape(-1))
                p.data = self.p[off:off + k].view_as(p)
                p.grad = self.g[off:off + k].view_as(p)
                self.offsets[id(p)] = (off, k)
                off += (k + a - 1) // a * a

    def zero_grad(self):
        self.g.zero_()

    def grad_view(self, p):
        off, k = self.offsets[id(p)]
        return self.g[off:off + k].view_as(p)


def broadcast_model(flat: FlatParams, model: torch.nn.Module, group=None, src: int = 0):
    """Ranks start from rank `src`'s weights and buffers (DataParallel re-replicates every forward instead)."""
    dist.broadcast(flat.p, src, group=group)
    for b in model.buffers():
        dist.broadcast(b, src, group=group)


def allreduce_grads(flat: FlatParams, group=None):
    """Sum all-reduce of the flat gradient buffer (one collective per step)."""
    dist.all_reduce(flat.g, op=dist.ReduceOp.SUM, group=group)


def rank_seed(base_seed: int, rank: int) -> int:
    """Per-rank data seed of the synthetic benchmark: rank r draws Generator(seed=r)."""
    return base_seed + rank
