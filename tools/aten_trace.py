#!/usr/bin/env python3
"""GPU box: which ATen (non-tatt_amd) kernels does one eager training step launch, and from where?  Prints, per ATen op that
launched a device kernel, the call count and the innermost tatt_amd / autograd frames."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tatt_amd  # noqa: E402
from tatt_amd.train import Trainer  # noqa: E402
from bench import make_batch, make_model  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(1234)
model = make_model("tatt").to(dev).train()
tr = Trainer(model, use_graph=False)
x, tp, hr = make_batch(48, 0, dev)
for _ in range(2):
    tr.step(x, tp, hr)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    tr.step(x, tp, hr)
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CPU and ev.name.startswith("aten::") and ev.cuda_time_total > 0 \
            and not any(c.name.startswith("aten::") and c.cuda_time_total > 0 for c in ev.cpu_children):
        st = [s for s in (ev.stack or []) if "tatt_amd" in s or "bench" in s]
        shapes = str(ev.input_shapes)[:60]
        cnt[(ev.name, st[0].split("/")[-1] if st else "<autograd engine>", shapes)] += 1
for (name, where, shapes), n in sorted(cnt.items(), key=lambda kv: -kv[1])[:70]:
    print("%4d  %-28s %-60s %s" % (n, name, where[:60], shapes))
