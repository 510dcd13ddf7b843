"""CPU, world_size 2 over gloo: the data-parallel plumbing (flat buffers, broadcast, ONE sum all-reduce, then
clip + Adam identically on every rank) reproduces the single-process result on the concatenated batch --
the semantics of the reference's DataParallel (gradients summed over replicas of a mean loss)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import tatt_oracle as O


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _tiny():
    torch.manual_seed(3)
    return torch.nn.Sequential(torch.nn.Conv2d(4, 8, 3, padding=1), torch.nn.PReLU(), torch.nn.Conv2d(8, 4, 3, padding=1))


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tatt_amd.dp import FlatParams, broadcast_model, allreduce_grads, rank_seed
    torch.manual_seed(100 + rank)                      # ranks start from DIFFERENT weights ...
    model = torch.nn.Sequential(torch.nn.Conv2d(4, 8, 3, padding=1), torch.nn.PReLU(), torch.nn.Conv2d(8, 4, 3, padding=1))
    unused = torch.nn.Parameter(torch.ones(5))         # a parameter that never gets a gradient
    model.register_parameter("unused", unused)
    flat = FlatParams(model)
    if rank == 0:
        ref = _tiny()
        with torch.no_grad():
            for p, r in zip(list(model.parameters())[1:], ref.parameters()):   # 'unused' is registered first
                p.copy_(r)
    broadcast_model(flat, model)                       # ... and are made identical to rank 0
    g = torch.Generator().manual_seed(rank_seed(0, rank))
    x, y = torch.rand(3, 4, 8, 8, generator=g), torch.rand(3, 4, 8, 8, generator=g)
    flat.zero_grad()
    loss = ((model(x) - y) ** 2).mean() * 100
    loss.backward()
    allreduce_grads(flat)
    gavg = flat.g / world
    grads = {"g": gavg}
    clipped, total = O.clip_grad_norm(grads, 0.25)
    p1, _, _ = O.adam_step(flat.p, clipped["g"], torch.zeros_like(flat.p), torch.zeros_like(flat.p), 1)
    # parameters start on 64-byte boundaries inside the flat buffers; the padding stays zero through the update
    used = torch.zeros(flat.n, dtype=torch.bool)
    per_param = []
    for prm in flat.params:
        off, k = flat.offsets[id(prm)]
        assert off % FlatParams.ALIGN == 0
        used[off:off + k] = True
        per_param.append(p1[off:off + k].clone())
    assert float(p1[~used].abs().max()) == 0.0 and float(flat.g[~used].abs().max()) == 0.0
    q.put((rank, torch.cat(per_param), float(total), x, y))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_dp_equals_single_process_on_full_batch():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, w0, n0, x0, y0), (_, w1, n1, x1, y1) = res
    assert torch.equal(w0, w1) and n0 == n1                      # ranks stay in lock-step
    # single process on the concatenated batch: mean loss over 2x the samples = average of the per-rank grads
    ref = _tiny()
    x, y = torch.cat([x0, x1]), torch.cat([y0, y1])
    loss = ((ref(x) - y) ** 2).mean() * 100
    loss.backward()
    flat_p = torch.cat([torch.ones(5)] + [p.detach().reshape(-1) for p in ref.parameters()])
    flat_g = torch.cat([torch.zeros(5)] + [p.grad.reshape(-1) for p in ref.parameters()])
    clipped, total = O.clip_grad_norm({"g": flat_g}, 0.25)
    p1, _, _ = O.adam_step(flat_p, clipped["g"], torch.zeros_like(flat_p), torch.zeros_like(flat_p), 1)
    assert abs(float(total) - n0) < 1e-4 * n0
    assert float((p1 - w0).abs().max()) < 1e-5
    assert torch.equal(w0[:5], torch.ones(5))                    # the grad-less parameter is untouched
