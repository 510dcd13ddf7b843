#!/usr/bin/env python3
"""GPU box: capture one training step as a hipGraph in debug mode and dump its DOT file (node = kernel, edges = dependencies):
which edges does the two-stream capture really contain?   python tools/graph_dot_dump.py gpurun_out/step_graph.dot"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tatt_amd  # noqa: E402,F401
from tatt_amd.train import Trainer  # noqa: E402
from bench import make_batch, make_model  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/step_graph.dot"
dev = torch.device("cuda:0")
torch.manual_seed(1234)
model = make_model("tatt").to(dev).train()
tr = Trainer(model, use_graph=False)
x, tp, hr = make_batch(48, 0, dev)
for _ in range(2):
    tr.step(x, tp, hr)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
g.enable_debug_mode()
with torch.cuda.graph(g):
    for k in range(len(tr.stages)):
        tr._stage(k, x, tp, hr)
    tr._optim()
g.debug_dump(out)
print("dumped", out, os.path.getsize(out))
