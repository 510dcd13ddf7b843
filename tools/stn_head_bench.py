"""GPU box: the STN head alone (forward, forward + backward), operator chain vs fused launches, replayed as hipGraphs (no host cost
between the launches: what the training step sees).  Under rocprofv3 --kernel-trace --stats the per-kernel durations of both variants."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tatt_amd.tsrn as T
from tatt_amd import functional as Fh

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 48
torch.manual_seed(0)
stn = T.STNHead(4, 20, "none").to(dev).train()
with torch.no_grad():
    stn.stn_fc2.weight.normal_(0, 0.05)
x = torch.rand(B, 4, 16, 64, device=dev)
dctrl = torch.randn(B, 20, 2, device=dev)


def fwd():
    with torch.no_grad():
        return T._stn_forward(x, stn, False)


def fwd_bwd():
    for p in stn.parameters():
        p.grad = None
    c = T._stn_forward(x, stn, False)
    c.backward(dctrl)


def timeit_graph(name, fn, iters=30):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
    torch.cuda.synchronize()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print("%-34s %8.1f us" % (name, e0.elapsed_time(e1) / iters * 1e3), flush=True)


for fused in (False, True):
    T.STN_FUSED = fused
    tag = "fused" if fused else "chain"
    timeit_graph("stn head forward, %s" % tag, fwd)
    timeit_graph("stn head forward + backward, %s" % tag, fwd_bwd)
Fh.sync_check()
