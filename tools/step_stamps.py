"""GPU box: the timeline of ONE replayed training step as the GPU executes it: one-thread stamp kernels (tatt_stamp: the 100 MHz wall
clock) are captured into the step's hipGraph at the points the code marks with functional.stamp(); every replay re-writes them.
Prints, for the mean of the last replays, each stamp's time relative to the step's first stamp.  Each stamp is a launch (~3-5 us on
its stream), so absolute times are slightly longer than the unstamped step; the ORDER and the gaps are what this is for.
usage: python tools/step_stamps.py [hook=value ...]"""
import sys, os, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tatt_amd
from tatt_amd import functional as Fh
from tatt_amd.train import Trainer
import bench

for a in sys.argv[1:]:
    path, val = a.split("=")
    mod, attr = path.rsplit(".", 1)
    cur = getattr(importlib.import_module(mod), attr)
    setattr(importlib.import_module(mod), attr, val if isinstance(cur, str) else type(cur)(int(val)))

dev = torch.device("cuda:0")
x, tp, hr = bench.make_batch(48, 0, dev)
torch.manual_seed(0)
m = tatt_amd.TSRN_TL_TRANS(scale_factor=2, width=128, height=32, STN=True, mask=True, srb_nums=5, hidden_units=32).to(dev).train()
tr = Trainer(m, use_graph=True, warmup_eager=2)
for _ in range(2):                       # the Trainer's eager warm-up steps, without stamps
    tr.step(x, tp, hr)
Fh.STAMPS = []
tr.step(x, tp, hr)                       # the capturing step: stamps become graph nodes
stamps = list(Fh.STAMPS)
assert stamps and tr._graphs is not None, "the stamped step was not the capturing one"
Fh.STAMPS = None
acc = None
N = 10
for _ in range(3):
    tr.step(x, tp, hr)
for _ in range(N):
    tr.step(x, tp, hr)
    torch.cuda.synchronize()
    v = torch.stack([t for _, t in stamps]).reshape(-1).cpu().double()
    acc = v if acc is None else acc + v
acc /= N
t0 = float(acc.min())
rows = sorted(zip([float(a) for a in acc], [n for n, _ in stamps]))
for t, name in rows:
    print("%9.1f us  %s" % ((t - t0) / 100.0, name))
Fh.sync_check()
