#!/usr/bin/env python3
"""GPU box: repeat the kernels / the forward pass that run concurrently with something (LDS double buffers, the forward fork of the
query GRU) many times on identical inputs and compare every result bit for bit with the first one."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tatt_amd  # noqa: E402
from tatt_amd import functional as Fh, ops  # noqa: E402
from oracle.fixtures import randomize_state_dict, make_inputs  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
B = 48


def repeat(name, fn, n):
    ref = fn().clone()
    bad = 0
    for _ in range(n):
        out = fn()
        if not torch.equal(out, ref):
            bad += 1
    torch.cuda.synchronize()
    print("%-44s %4d repetitions, %d differ" % (name, n, bad), flush=True)


x = torch.randn(B, 32, 128, 64, generator=g).to(dev)
w = (torch.randn(4, 64, 9, 9, generator=g) * 0.02).to(dev)
b = torch.randn(4, generator=g).to(dev)
repeat("conv9 64->4 mfma, HR", lambda: ops.conv2d_forward(x, w, b), 300)
xl = torch.randn(B, 16, 64, 64, generator=g).to(dev)
w1 = (torch.randn(64, 4, 9, 9, generator=g) * 0.02).to(dev)
repeat("conv9 dgrad mfma, LR", lambda: ops.conv2d_dgrad(xl, w1), 300)
x4 = torch.randn(B, 16, 64, 4, generator=g).to(dev)
b64 = torch.randn(64, generator=g).to(dev)
repeat("conv9 4->64 (block1 forward), LR", lambda: ops.conv2d_forward(x4, w1, b64), 300)
dy4 = torch.randn(B, 32, 128, 4, generator=g).to(dev)
repeat("conv9 4->64 (output conv data gradient), HR", lambda: ops.conv2d_dgrad(dy4, w), 200)
repeat("conv9 weight gradient 64 x 4, HR", lambda: ops.conv_wgrad(x, dy4, 4, 9, 9), 100)
repeat("conv9 weight gradient 4 x 64, LR", lambda: ops.conv_wgrad(x4, xl, 64, 9, 9), 100)
qa = [torch.randn(B * 64, 1536, generator=g).to(dev) for _ in range(2)]
qb = [torch.randn(B * 64, 512, generator=g).to(dev) for _ in range(2)]
repeat("query GRU dW_hh, both directions", lambda: torch.cat([t.reshape(-1) for t in ops.qgru_wgrad_sb(qa[0], qa[1], qb[0], qb[1])]), 100)
w3 = (torch.randn(64, 64, 3, 3, generator=g) * 0.05).to(dev)
b3 = torch.randn(64, generator=g).to(dev)
repeat("conv3 ws16 64->64", lambda: ops.conv2d_forward(xl, w3, b3), 300)
w4 = (torch.randn(256, 64, 3, 3, generator=g) * 0.05).to(dev)
repeat("conv3 ws16 64->256", lambda: ops.conv2d_forward(xl, w4, None), 100)
repeat("conv3 ws16 dgrad", lambda: ops.conv2d_dgrad(xl, w3), 200)
repeat("conv3 wgrad (batched reduce off)", lambda: ops.conv_wgrad(xl, xl, 64, 3, 3), 100)

# second-generation TP layer (csrc/tplayer2.hip): the backward's four waves share the weight gradients through their LDS images (16
# barriers per round), every launch re-reads the forward's relu bits
r3 = lambda *sh: (torch.randn(*sh, generator=g) * 0.3).to(dev)
tx, tq, tK, tV, tup = r3(B, 1024, 64), r3(B, 1024, 64), r3(B, 26, 64), r3(B, 26, 64), r3(B, 1024, 64)
tlp = (r3(192, 64), r3(192), r3(64, 64), r3(64), r3(64, 64), r3(64), r3(64, 64), r3(64), r3(64) + 1, r3(64), r3(64) + 1, r3(64))
tln = (r3(64) + 1, r3(64))
tsd = Fh.seed_tensor(dev)
tpk = ops.tplayer2_prep(tlp, tK, tV)
t2f = lambda: ops.tplayer2_fwd(tx, tq, tpk, tlp, tln, 0.5, 1, 0.1, 0.1, 0.1, tsd, 10, 1e-5, False, True, 26)
thm = t2f()[3]
repeat("tplayer2 fwd (fin, wavg, relu bits)", lambda: torch.cat([t2f()[1].reshape(-1), t2f()[2].reshape(-1), t2f()[3].float()]), 100)


def t2b():
    dx, dq, kv, fl, pp, G2 = ops.tplayer2_bwd(tx, tq, tpk, tlp, tln, 0.5, 1, 0.1, 0.1, 0.1, tsd, 10, 1e-5, None, tup, None, None, True, 26, hmask=thm)
    dK, dV = ops.tplayer2_reduce_kv(kv, fl, B, 1024, 26)
    return torch.cat([dx.reshape(-1), dq.reshape(-1), pp.reshape(-1), dK.reshape(-1), dV.reshape(-1)])


repeat("tplayer2 bwd (dx, dqpos, records, dK, dV)", t2b, 200)

torch.manual_seed(1234)
m = tatt_amd.TSRN_TL_TRANS(scale_factor=2, width=128, height=32, STN=True, mask=True, srb_nums=5, hidden_units=32)
m.load_state_dict(randomize_state_dict(m.state_dict()))
m = m.to(dev).train()
xi, tp, hr = (t.to(dev) for t in make_inputs(8, seed=3))
# something else keeps the GPU busy on a third stream while the forward (with its fork) runs
busy = torch.cuda.Stream()
junk = torch.randn(4096, 4096, device=dev)


def fwd():
    Fh.set_seed(dev, 7)
    Fh.FWD_FORK.enabled = True
    try:
        with torch.cuda.stream(busy):
            for _ in range(3):
                torch.mm(junk, junk)
        with torch.no_grad():
            sr, mid = m(xi, tp)
    finally:
        Fh.FWD_FORK.enabled = False
    return torch.cat([sr.reshape(-1), mid["trans_feat"].reshape(-1)])


repeat("train-mode forward with the query-GRU fork", fwd, 150)

# ---- round 4: the new kernels (second-generation BiGRU recurrences, fragment stream, split-bf16 3x3 weight gradient) -----------------
M = B * 16 * 64
gi = torch.randn(M, 192, generator=g).to(dev)
whh, bhh = (torch.randn(96, 32, generator=g) * 0.2).to(dev), (torch.randn(96, generator=g) * 0.1).to(dev)
dout = torch.randn(M, 64, generator=g).to(dev)
xa, xb_ = torch.randn(M, 64, generator=g).to(dev), torch.randn(M, 64, generator=g).to(dev)
for vert in (True, False):
    geom = ops.seq_geom(B, 16, 64, vert)
    nm = "vertical" if vert else "horizontal"
    repeat("gru32 forward, " + nm, lambda: torch.cat([t.reshape(-1) for t in ops.gru32_fwd(gi, whh, bhh, whh, bhh, geom, save=True)]), 100)
    out_, gates_ = ops.gru32_fwd(gi, whh, bhh, whh, bhh, geom, save=True)
    repeat("gru32 backward + fragments, " + nm,
           lambda: torch.cat([t.reshape(-1) for t in ops.gru32_bwd_frag(gates_, out_, dout, whh, whh, geom)]), 100)
    _, frag_ = ops.gru32_bwd_frag(gates_, out_, dout, whh, whh, geom)

    def wg():
        dWp, dWh, dbp, dbh = (torch.empty(192, 128, device=dev), torch.empty(192, 32, device=dev), torch.empty(192, device=dev),
                              torch.empty(192, device=dev))
        ops.gru_wgrad_frag(frag_, xa, xb_, geom, dWp, dWh, dbp, dbh)
        return torch.cat([dWp.reshape(-1), dWh.reshape(-1), dbp, dbh])
    repeat("gru weight gradients from fragments, " + nm, wg, 100)
repeat("conv3 wgrad split-bf16 (+ bias)", lambda: torch.cat([t.reshape(-1) for t in ops.conv_wgrad(xl, xl, 64, 3, 3, want_db=True)]), 100)

# ---- the whole staged two-lane step, replayed from identical state: loss and every weight bit for bit --------------------------------
from tatt_amd.train import Trainer  # noqa: E402


def train_run(nsteps, use_graph):
    torch.manual_seed(1234)
    mm = tatt_amd.TSRN_TL_TRANS(scale_factor=2, width=128, height=32, STN=True, mask=True, srb_nums=5, hidden_units=32)
    mm.load_state_dict(randomize_state_dict(mm.state_dict()))
    mm = mm.to(dev).train()
    Fh.set_seed(dev, 99)
    tr = Trainer(mm, use_graph=use_graph, warmup_eager=2)
    x8, tp8, hr8 = (t.to(dev) for t in make_inputs(48, seed=5))
    ls = [float(tr.step(x8, tp8, hr8)) for _ in range(nsteps)]
    torch.cuda.synchronize()
    return ls, tr.flat_p.clone()


ref_l, ref_p = train_run(6, True)
bad = 0
for rep in range(6):
    l, pp = train_run(6, True)
    bad += int(l != ref_l or not torch.equal(pp, ref_p))
print("%-44s %4d repetitions, %d differ" % ("6 training steps (graph, 2 lanes, B = 48)", 6, bad), flush=True)
l, pp = train_run(6, False)
print("%-44s eager vs graph: losses %s, weights %s" % ("", "equal" if l == ref_l else "DIFFER", "equal" if torch.equal(pp, ref_p) else "DIFFER"),
      flush=True)
