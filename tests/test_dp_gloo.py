"""CPU, world_size 2 over gloo: tatt_amd.train.Trainer's data-parallel step -- flat bucketed buffers, broadcast from rank 0, the
backward run in stages (tatt_amd.dp.GradCuts) with one asynchronous sum all-reduce per bucket, 1/world folded into Adam, clip +
Adam identically on every rank -- reproduces the single-process result on the concatenated batch: the semantics of the
reference's DataParallel (gradients summed over replicas of a mean loss, interfaces/base.py:386-396).

The generator itself only runs on a GPU, so the model here is a small torch module that implements the same Trainer protocol
(`grad_buckets` / `set_grad_cuts`, three backward stages like TSRN_TL_TRANS: trunk, "tp", "first"); the device kernels behind
the optimiser are injected (tests.util.TorchStepKernels).  Everything else -- the code under test -- is the product's."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import tatt_oracle as O


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class Toy(torch.nn.Module):
    """first -> (tp branch, trunk) -> out, cut like tatt_amd.tsrn._GeneratorBase._trunk_forward."""

    def __init__(self):
        super().__init__()
        self.first = torch.nn.Sequential(torch.nn.Conv2d(4, 8, 3, padding=1), torch.nn.PReLU())
        self.tp = torch.nn.Conv2d(8, 8, 1)
        self.trunk = torch.nn.Conv2d(16, 4, 3, padding=1)
        self.unused = torch.nn.Parameter(torch.ones(5))          # a parameter that never gets a gradient
        self._cuts = None

    def set_grad_cuts(self, cuts):
        self._cuts = cuts

    def grad_buckets(self):
        return [("trunk", list(self.trunk.parameters())), ("tp", list(self.tp.parameters()) + [self.unused]),
                ("first", list(self.first.parameters()))]

    def forward(self, x):
        c = self._cuts if self.training else None
        b1 = self.first(x)
        t = torch.tanh(self.tp(c.cut("first", b1) if c else b1))
        if c:
            b1, t = c.cut("first", b1), c.cut("tp", t)
        return self.trunk(torch.cat([b1, t], 1))


def _toy(seed):
    torch.manual_seed(seed)
    return Toy()


def _loss(sr, hr):
    return ((sr - hr) ** 2).mean() * 100


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tatt_amd.dp import rank_seed, FlatParams
    from tatt_amd.train import Trainer
    from tests.util import TorchStepKernels
    model = _toy(3 if rank == 0 else 100 + rank).train()          # ranks start from DIFFERENT weights, rank 0's win
    tr = Trainer(model, process_group=dist.group.WORLD, kernels=TorchStepKernels(), loss_fn=_loss)
    assert tr.stages == ["trunk", "tp", "first"] and tr.cuts is not None and len(tr.flat.ranges) == 3
    g = torch.Generator().manual_seed(rank_seed(0, rank))
    x, y = torch.rand(3, 4, 8, 8, generator=g), torch.rand(3, 4, 8, 8, generator=g)
    losses = [float(tr.step(x, None, y)) for _ in range(2)]
    # buckets go on the wire in completion order, each exactly once, the first ones BEFORE the last pass has run:
    # (last pass of the group, first bucket, last bucket)
    assert tr.reduce_log == [(1, 0, 0), (2, 1, 2)], tr.reduce_log
    # parameters start on 64-byte boundaries inside the flat buffers; the padding stays zero through the updates
    used = torch.zeros(tr.flat.n, dtype=torch.bool)
    for prm in tr.flat.params:
        off, k = tr.flat.offsets[id(prm)]
        assert off % FlatParams.ALIGN == 0
        used[off:off + k] = True
    assert float(tr.flat.p[~used].abs().max()) == 0.0 and float(tr.flat.g[~used].abs().max()) == 0.0
    # numpy payloads are pickled by value (torch tensors would travel as shared-memory handles that die with this process)
    q.put((rank, {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}, float(tr.last_grad_norm), losses,
           x.numpy(), y.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_trainer_equals_single_process_on_full_batch():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, sd0, n0, l0, x0, y0), (_, sd1, n1, l1, x1, y1) = [
        (r, {k: torch.from_numpy(v) for k, v in sd.items()}, n, l, torch.from_numpy(x), torch.from_numpy(y)) for r, sd, n, l, x, y in res]
    for k in sd0:
        assert torch.equal(sd0[k], sd1[k]), k                     # ranks stay in lock-step
    assert n0 == n1
    # single process, plain autograd, on the concatenated batch: mean loss over 2x the samples = average of the per-rank gradients
    ref = _toy(3).train()
    x, y = torch.cat([x0, x1]), torch.cat([y0, y1])
    names = [k for k, _ in ref.named_parameters()]
    sd = {k: v.detach().clone() for k, v in ref.named_parameters()}
    state = {}
    for step in (1, 2):
        ref.load_state_dict(sd, strict=False)
        ref.zero_grad()
        _loss(ref(x), y).backward()
        grads = {k: p.grad for k, p in ref.named_parameters() if p.grad is not None}
        clipped, total = O.clip_grad_norm(grads, 0.25)
        for k, gk in clipped.items():
            m, v = state.get(k, (torch.zeros_like(gk), torch.zeros_like(gk)))
            sd[k], m, v = O.adam_step(sd[k], gk, m, v, step)
            state[k] = (m, v)
    assert abs(float(total) - n0) < 1e-4 * n0                     # norm of the rank-averaged gradient at the second step
    for k in names:
        assert float((sd[k] - sd0[k]).abs().max()) < 2e-5, k
    assert torch.equal(sd0["unused"], torch.ones(5))              # the grad-less parameter is untouched


def test_grad_cuts_match_single_pass_backward():
    """The staged backward is the single-pass backward: same gradients bit for bit (same kernels, same order)."""
    from tatt_amd.dp import GradCuts
    a, b = _toy(5).train(), _toy(5).train()
    x, y = torch.rand(2, 4, 8, 8), torch.rand(2, 4, 8, 8)
    _loss(a(x), y).backward()
    cuts = GradCuts()
    b.set_grad_cuts(cuts)
    _loss(b(x), y).backward()
    assert b.first[0].weight.grad is None and b.tp.weight.grad is None and b.trunk.weight.grad is not None
    cuts.run("tp")
    assert b.tp.weight.grad is not None and b.first[0].weight.grad is None
    cuts.run("first")
    for (k, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        assert (pa.grad is None) == (pb.grad is None), k
        if pa.grad is not None:
            assert float((pa.grad - pb.grad).abs().max()) < 1e-6, k


def test_rank_dropout_seeds_differ():
    from tatt_amd.dp import rank_dropout_seed
    s = {rank_dropout_seed(0x1234ABCD5678EF01, r) for r in range(8)}
    assert len(s) == 8 and all(0 <= v < 2 ** 63 for v in s)
