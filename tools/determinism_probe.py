#!/usr/bin/env python3
"""GPU box: is a training step bit-reproducible?  Fresh model, same seeds, same data, N repetitions per configuration; prints the
losses of the first two steps and a checksum of the weights after them.  Toggles isolate the concurrent pieces."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tatt_amd  # noqa: E402
from tatt_amd import functional as Fh, ops  # noqa: E402
from tatt_amd.train import Trainer  # noqa: E402
from oracle.fixtures import randomize_state_dict, make_inputs  # noqa: E402

dev = torch.device("cuda:0")
STD = dict(scale_factor=2, width=128, height=32, STN=True, mask=True, srb_nums=5, hidden_units=32)


def run(**kw):
    torch.manual_seed(1234)
    m = tatt_amd.TSRN_TL_TRANS(**STD)
    m.load_state_dict(randomize_state_dict(m.state_dict()))
    m = m.to(dev).train()
    Fh.set_seed(dev, 99)
    tr = Trainer(m, dropout_seed=5, **kw)
    out = []
    for i in range(3):
        x, tp, hr = make_inputs(4, seed=40 + i)
        out.append(float(tr.step(x.to(dev), tp.to(dev), hr.to(dev))))
    torch.cuda.synchronize()
    return out + [float(tr.flat_p.double().sum()), float(tr.flat_m.double().abs().sum())]


import socket  # noqa: E402
import torch.distributed as dist  # noqa: E402
sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
PG = dist.group.WORLD
orig_fork = Fh.FWD_FORK.run
for name, setup, kw in [
        ("data parallel (1 rank), eager", lambda: None, dict(process_group=PG)),
        ("data parallel (1 rank), hipGraph", lambda: None, dict(process_group=PG, use_graph=True, warmup_eager=1)),
        ("single GPU, hipGraph", lambda: None, dict(use_graph=True, warmup_eager=1)),
        ("default (2 lanes, forward fork, conv9 mfma)", lambda: None, {}),
        ("forward fork off", lambda: setattr(Fh.FWD_FORK, "run", lambda ref, fn: fn()), {}),
        ("one stream", lambda: setattr(Fh.FWD_FORK, "run", orig_fork), dict(side_stream=False)),
        ("no deferral", lambda: None, dict(defer_param_grads=False)),
        ("exact-fp32 3x3 convolutions and GRU projections, one stream",
         lambda: (setattr(ops, "CONV3_SB", False), setattr(Fh, "TOKGEMM_SB", False)), dict(side_stream=False))]:
    setup()
    rs = [run(**kw) for _ in range(4)]
    same = all(r == rs[0] for r in rs)
    print("%-48s %s" % (name, "bit-identical x4" if same else "DIFFERS"))
    for r in rs if not same else rs[:1]:
        print("     loss1 %.9f loss2 %.9f loss3 %.9f sum(p) %.9f sum|m| %.9f" % tuple(r))
dist.destroy_process_group()
