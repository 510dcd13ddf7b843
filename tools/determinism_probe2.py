#!/usr/bin/env python3
"""GPU box: replays the sequence of tests/test_model_gpu.py::test_single_rank_process_group_runs_the_staged_step and prints the
per-step losses of each run (bitwise) -- hunting a 1e-6 difference seen once between two data-parallel runs."""
import os
import socket
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tatt_amd  # noqa: E402
from tatt_amd import functional as Fh  # noqa: E402
from tatt_amd.dp import rank_dropout_seed  # noqa: E402
from tatt_amd.train import Trainer  # noqa: E402
from oracle.fixtures import randomize_state_dict, make_inputs  # noqa: E402

dev = torch.device("cuda:0")
STD = dict(scale_factor=2, width=128, height=32, STN=True, mask=True, srb_nums=5, hidden_units=32)


def run(nsteps, **kw):
    torch.manual_seed(1234)
    m = tatt_amd.TSRN_TL_TRANS(**STD)
    m.load_state_dict(randomize_state_dict(m.state_dict()))
    m = m.to(dev).train()
    Fh.set_seed(dev, 99)
    tr = Trainer(m, **kw)
    out = []
    for i in range(nsteps):
        x, tp, hr = make_inputs(4, seed=40 + i)
        out.append(tr.step(x.to(dev), tp.to(dev), hr.to(dev)))
    torch.cuda.synchronize()
    return ["%.9f" % float(v) for v in out], int(Fh.seed_tensor(dev))


sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
PG = dist.group.WORLD
base = 77
for rep in range(2):
    print("l0 single GPU graph ", run(5, use_graph=True, dropout_seed=rank_dropout_seed(base, 0)))
    print("l1 DP graph         ", run(5, use_graph=True, process_group=PG, dropout_seed=base))
    print("l2 DP eager         ", run(3, use_graph=False, process_group=PG, dropout_seed=base))
    print("l3 DP eager again   ", run(3, use_graph=False, process_group=PG, dropout_seed=base))
    print("l4 single GPU eager ", run(3, use_graph=False, dropout_seed=rank_dropout_seed(base, 0)))
dist.destroy_process_group()
