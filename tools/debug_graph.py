"""GPU-box diagnostic: staged hipGraph capture of the TATT step (which stage breaks capture?)."""
import faulthandler
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.enable()
import torch
import tatt_amd
from tatt_amd import ops, functional as Fh
from tatt_amd.train import Trainer, image_loss

dev = torch.device("cuda:0")
stage = sys.argv[1] if len(sys.argv) > 1 else "all"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4


def say(*a):
    print(*a, flush=True)


def capture(fn, warm=2):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(warm):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    return g, out


if stage in ("kernel", "all"):
    x = torch.randn(1000, 64, device=dev)
    w = torch.randn(64, 64, device=dev)
    g, y = capture(lambda: ops.linear_fwd(x, w))
    say("kernel capture ok", float((y - x @ w.t()).abs().max()))

torch.manual_seed(0)
m = tatt_amd.TSRN_TL_TRANS(scale_factor=2, width=128, height=32, STN=True).to(dev)
x = torch.rand(B, 4, 16, 64, device=dev)
tp = torch.softmax(torch.randn(B, 37, 1, 26, device=dev), 1)
hr = torch.rand(B, 4, 32, 128, device=dev)

if stage in ("evalfwd", "all"):
    m.eval()
    with torch.no_grad():
        g, out = capture(lambda: m(x, tp)[0])
    say("eval forward capture ok", float(out.abs().sum()))

if stage in ("trainfwd", "all"):
    m.train()
    with torch.no_grad():
        g, out = capture(lambda: m(x, tp)[0])
    say("train forward (no grad) capture ok", float(out.abs().sum()))

if stage in ("fwdbwd", "all"):
    m.train()

    def fb():
        for p in m.parameters():
            p.grad = None
        sr, _ = m(x, tp)
        loss = image_loss(sr, hr).mean() * 100
        loss.backward()
        m.block = None
        return loss.detach()
    g, out = capture(fb)
    say("fwd+bwd capture ok", float(out))

if stage in ("trainer", "all"):
    m.train()
    tr = Trainer(m, use_graph=True, warmup_eager=2)
    for i in range(6):
        l = tr.step(x, tp, hr)
        torch.cuda.synchronize()
        say("trainer step", i, float(l))
    import time
    t0 = time.time()
    for i in range(10):
        tr.step(x, tp, hr)
    torch.cuda.synchronize()
    say("graph replay ms/step", (time.time() - t0) * 100)
