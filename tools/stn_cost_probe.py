"""GPU box: what the STN head (+ TPS sampler) costs the training step, and how that interacts with the persistent query-GRU launches.
Step time (hipGraph replay, B = 48, 16x64, dropout on) for STN on / off x query-GRU chains off / bwd / fwd+bwd.  The STN-off model is
the upper bound of what consolidating the STN head's ~50 + ~60 small launches can give."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tatt_amd
from tatt_amd import functional as Fh
from tatt_amd.train import Trainer
import bench

dev = torch.device("cuda:0")
x, tp, hr = bench.make_batch(48, 0, dev)


def run(stn, fwd, bwd, fused=True, steps=30):
    import tatt_amd.tsrn as T
    T.STN_FUSED = fused
    Fh.QGRU_CHAIN_FWD, Fh.QGRU_CHAIN_BWD = fwd, bwd
    torch.manual_seed(0)
    m = tatt_amd.TSRN_TL_TRANS(scale_factor=2, width=128, height=32, STN=stn, mask=True, srb_nums=5, hidden_units=32).to(dev).train()
    tr = Trainer(m, use_graph=True, warmup_eager=2)
    for _ in range(6):
        tr.step(x, tp, hr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step(x, tp, hr)
    torch.cuda.synchronize()
    Fh.sync_check()
    return (time.perf_counter() - t0) / steps * 1e3


for rep in range(2):
    for stn, fused in ((True, False), (True, True), (False, True)):
        for fwd, bwd in ((False, False), (False, True), (True, False), (True, True)):
            print("STN %-5s fused head %-5s chain fwd %-5s bwd %-5s  %.3f ms/step" % (stn, fused, fwd, bwd, run(stn, fwd, bwd, fused)),
                  flush=True)
