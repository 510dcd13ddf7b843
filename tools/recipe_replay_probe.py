#!/usr/bin/env python3
"""GPU box: the TSSIM recipe (two generator forwards per step) run eagerly and as a replayed hipGraph from identical state: per-step losses
and final weights must be bit-identical (the second half of tests/test_losses_gpu.py::test_tssim_recipe_trainer_step, without the oracle).
usage: python tools/recipe_replay_probe.py [mod.attr=val ...]"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.getcwd())
for kv in sys.argv[1:]:
    path, val = kv.split("=", 1)
    mod, attr = path.rsplit(".", 1)
    setattr(importlib.import_module(mod), attr, int(val))
import tatt_amd  # noqa: E402
from oracle.fixtures import randomize_state_dict, make_inputs  # noqa: E402
from tatt_amd.train import Trainer, TssimRecipe  # noqa: E402

dev = torch.device("cuda:0")
kw = dict(scale_factor=2, width=128, height=32, STN=False, mask=True, srb_nums=5, hidden_units=32)
x, tp, hr = (t.to(dev) for t in make_inputs(3, seed=21))


def run(use_graph, recipe=True):
    torch.manual_seed(1234)
    m = tatt_amd.TSRN_TL_TRANS(**kw)
    m.load_state_dict(randomize_state_dict(m.state_dict()))
    m = m.to(dev).train()
    m.infoGen.dropout_on = False
    t = Trainer(m, use_graph=use_graph, warmup_eager=2, recipe=TssimRecipe(5.0, seed=9) if recipe else None)
    ls = [float(t.step(x, tp, hr)) for _ in range(6)]
    torch.cuda.synchronize()
    return ls, t.flat_p.clone()


for recipe in (True, False):
    le, pe = run(False, recipe)
    lg, pg = run(True, recipe)
    print("recipe" if recipe else "plain ", "eager", ["%.5f" % v for v in le])
    print("recipe" if recipe else "plain ", "graph", ["%.5f" % v for v in lg], "weights", "equal" if torch.equal(pe, pg) else "DIFFER", flush=True)
