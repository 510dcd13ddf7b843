#!/usr/bin/env python3
"""GPU box: where does the TP interpreter's backward lose precision?  HIP (fp32) and the oracle in fp32 (CPU) are both compared
with the oracle in fp64 on the same inputs, operator group by operator group: relative l2 errors of outputs and gradients."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tatt_amd  # noqa: E402
from tatt_amd import functional as Fh  # noqa: E402
from tatt_amd.tsrn import _tp_interpreter  # noqa: E402
from oracle import tatt_oracle as O  # noqa: E402
from oracle.fixtures import randomize_state_dict  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(1234)
m = tatt_amd.TSRN_TL_TRANS(scale_factor=2, width=128, height=32, STN=False, mask=True, srb_nums=5, hidden_units=32)
m.load_state_dict(randomize_state_dict(m.state_dict()))
sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
m = m.to(dev).train()
m.infoGen.dropout_on = False
B = 4
g = torch.Generator().manual_seed(1)
feat = torch.randn(B, 64, 16, 64, generator=g) * 0.5
tp = torch.softmax(torch.randn(B, 37, 1, 26, generator=g), 1)
w_out = torch.randn(B, 64, 16, 64, generator=g)


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def oracle_run(dtype):
    leaves = {k: v.detach().clone().to(dtype).requires_grad_(True) for k, v in sd.items() if k.startswith("infoGen") and O.is_param(k)}
    full = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
    full.update(leaves)
    f = feat.clone().to(dtype).requires_grad_(True)
    t = tp.clone().to(dtype).requires_grad_(True)
    tp_map, wts = O.tp_interpreter(f, t, full, "infoGen", False)
    (tp_map * w_out.to(dtype)).sum().backward()
    return tp_map, wts, f.grad, t.grad, {k: v.grad for k, v in leaves.items()}


o64 = oracle_run(torch.float64)
o32 = oracle_run(torch.float32)
fh = feat.permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_(True)
th = tp.to(dev).requires_grad_(True)
for p in m.parameters():
    p.grad = None
tp_map, wts = _tp_interpreter(fh, th, m.infoGen, True)
(tp_map * w_out.permute(0, 2, 3, 1).contiguous().to(dev)).sum().backward()
print("%-70s %10s %10s" % ("tensor", "hip-vs-64", "cpu32-vs-64"))
print("%-70s %10.2e %10.2e" % ("tp_map", rel(tp_map.permute(0, 3, 1, 2), o64[0]), rel(o32[0], o64[0])))
print("%-70s %10.2e %10.2e" % ("pr_weights", rel(wts, o64[1]), rel(o32[1], o64[1])))
print("%-70s %10.2e %10.2e" % ("d feat", rel(fh.grad.permute(0, 3, 1, 2), o64[2]), rel(o32[2], o64[2])))
print("%-70s %10.2e %10.2e" % ("d text prior", rel(th.grad, o64[3]), rel(o32[3], o64[3])))
params = dict(m.named_parameters())
for k, g64 in o64[4].items():
    if g64 is None:
        continue
    print("%-70s %10.2e %10.2e" % (k, rel(params[k].grad, g64), rel(o32[4][k], g64)))

# ---- single operators, fp64 yardstick ------------------------------------------------------------------------------------------
print("\nsingle operators (forward / input gradients), relative l2 error vs fp64")
M, C = 4096, 64
x = torch.randn(M, C, generator=g)
r = torch.randn(M, C, generator=g)
gam, bet = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
wy = torch.randn(M, C, generator=g)


def ln_ref(dt):
    a, b2, gg, bb = (t.clone().to(dt).requires_grad_(True) for t in (x, r, gam, bet))
    y = O.layer_norm(a + b2, gg, bb)
    (y * wy.to(dt)).sum().backward()
    return y, a.grad, gg.grad, bb.grad


class _LN(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.weight, self.bias, self.eps = torch.nn.Parameter(gam.clone().to(dev)), torch.nn.Parameter(bet.clone().to(dev)), 1e-5


ln = _LN()
xa, xb = x.to(dev).requires_grad_(True), r.to(dev).requires_grad_(True)
y = Fh.layer_norm(xa, xb, ln)
(y * wy.to(dev)).sum().backward()
r64, r32 = ln_ref(torch.float64), ln_ref(torch.float32)
for name, h, i in (("layer_norm y", y, 0), ("layer_norm dx", xa.grad, 1), ("layer_norm dgamma", ln.weight.grad, 2), ("layer_norm dbeta", ln.bias.grad, 3)):
    print("%-70s %10.2e %10.2e" % (name, rel(h, r64[i]), rel(r32[i], r64[i])))

# linear
W, bb = torch.randn(64, 64, generator=g) * 0.2, torch.randn(64, generator=g) * 0.1


def lin_ref(dt):
    a, w_, b_ = (t.clone().to(dt).requires_grad_(True) for t in (x, W, bb))
    y = torch.relu(a @ w_.t() + b_)
    (y * wy.to(dt)).sum().backward()
    return y, a.grad, w_.grad, b_.grad


xa = x.to(dev).requires_grad_(True)
Wd, bd = W.to(dev).requires_grad_(True), bb.to(dev).requires_grad_(True)
y = Fh.linear(xa, Wd, bd, act=1)
(y * wy.to(dev)).sum().backward()
r64, r32 = lin_ref(torch.float64), lin_ref(torch.float32)
for name, h, i in (("linear+relu y", y, 0), ("linear dx", xa.grad, 1), ("linear dW", Wd.grad, 2), ("linear db", bd.grad, 3)):
    print("%-70s %10.2e %10.2e" % (name, rel(h, r64[i]), rel(r32[i], r64[i])))

# attention core + projections (decoder geometry: 1024 queries, 26 keys)
pre = "infoGen.transformer.decoder.layers.0.multihead_attn"
q_in = torch.randn(B, 1024, 64, generator=g)
k_in = torch.randn(B, 26, 64, generator=g)
v_in = torch.randn(B, 26, 64, generator=g)
wo = torch.randn(B, 1024, 64, generator=g)
ww = torch.randn(B, 1024, 26, generator=g)


def mha_ref(dt):
    a, k_, v_ = (t.clone().to(dt).requires_grad_(True) for t in (q_in, k_in, v_in))
    s2 = {k: v.to(dt) for k, v in sd.items() if k.startswith(pre)}
    out, wt = O.mha(a, k_, v_, s2, pre, 4)
    ((out * wo.to(dt)).sum() + (wt * ww.to(dt)).sum()).backward()
    return out, wt, a.grad, k_.grad, v_.grad


qa, ka, va = (t.to(dev).requires_grad_(True) for t in (q_in, k_in, v_in))
mh = m.infoGen.transformer.decoder.layers[0].multihead_attn
out, wt = Fh.multihead_attention(qa, ka, va, mh, False, 7)
((out * wo.to(dev)).sum() + (wt * ww.to(dev)).sum()).backward()
r64, r32 = mha_ref(torch.float64), mha_ref(torch.float32)
for name, h, i in (("mha out", out, 0), ("mha weights", wt, 1), ("mha dq_in", qa.grad, 2), ("mha dk_in", ka.grad, 3), ("mha dv_in", va.grad, 4)):
    print("%-70s %10.2e %10.2e" % (name, rel(h, r64[i]), rel(r32[i], r64[i])))

# query GRU
def q_ref(dt):
    s2 = {k: (v.clone().to(dt).requires_grad_(True) if k.startswith("infoGen.transformer.gru_encoding") or k == "infoGen.init_factor.weight" else v)
          for k, v in sd.items()}
    q = O.query_embedding(s2, "infoGen", B, 16, 64)
    (q * wo.to(dt)).sum().backward()
    return q, s2["infoGen.init_factor.weight"].grad, s2["infoGen.transformer.gru_encoding.weight_hh_l0_reverse"].grad, s2["infoGen.transformer.gru_encoding.weight_ih_l0"].grad


for p in m.parameters():
    p.grad = None
q = Fh.query_embedding(m.infoGen.init_factor.weight, m.infoGen.transformer.gru_encoding, B, 16, 64).reshape(B, 1024, 64)
(q * wo.to(dev)).sum().backward()
r64, r32 = q_ref(torch.float64), q_ref(torch.float32)
for name, h, i in (("query gru q", q, 0), ("query gru d init_factor", m.infoGen.init_factor.weight.grad, 1),
                   ("query gru d weight_hh_reverse", m.infoGen.transformer.gru_encoding.weight_hh_l0_reverse.grad, 2),
                   ("query gru d weight_ih", m.infoGen.transformer.gru_encoding.weight_ih_l0.grad, 3)):
    print("%-70s %10.2e %10.2e" % (name, rel(h, r64[i]), rel(r32[i], r64[i])))

# ---- the STN front end at the tatt_train_b4 geometry: how far do the control points / the rectified image move? ------------------
print("\nSTN head + TPS sampler (train mode, B = 4), relative l2 / max-abs error vs fp64")
import numpy as np  # noqa: E402
from tatt_amd.tsrn import _stn_forward, _tps_forward  # noqa: E402
z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "tatt_train_b4.npz"))
torch.manual_seed(1234)
m2 = tatt_amd.TSRN_TL_TRANS(scale_factor=2, width=128, height=32, STN=True, mask=True, srb_nums=5, hidden_units=32)
m2.load_state_dict(randomize_state_dict(m2.state_dict()))
sd2 = {k: v.detach().clone() for k, v in m2.state_dict().items()}
m2 = m2.to(dev).train()
xin = torch.from_numpy(z["x"])


def stn_ref(dt):
    s2 = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in sd2.items()}
    ctrl = O.stn_head(xin.to(dt), s2, "stn_head", True, {})
    xr, src = O.tps_transform(xin.to(dt), ctrl, s2, "tps")
    return ctrl, src, xr


with torch.no_grad():
    ctrl = _stn_forward(xin.to(dev), m2.stn_head)
    xr, src = _tps_forward(xin.to(dev), ctrl, m2.tps)
r64, r32 = stn_ref(torch.float64), stn_ref(torch.float32)


def mx(a, b):
    return float((a.detach().double().cpu() - b.detach().double().cpu()).abs().max())


for name, h, i in (("control points", ctrl, 0), ("source coordinates", src, 1), ("rectified image", xr.permute(0, 3, 1, 2), 2)):
    print("%-40s rel %10.2e %10.2e   max-abs %10.2e %10.2e" % (name, rel(h, r64[i]), rel(r32[i], r64[i]), mx(h, r64[i]), mx(r32[i], r64[i])))
# with the SAME (fp64-rounded) control points: the sampler alone
ctrl64 = r64[0].float()
with torch.no_grad():
    xr2, src2 = _tps_forward(xin.to(dev), ctrl64.to(dev), m2.tps)
s2 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd2.items()}
xr64, src64 = O.tps_transform(xin.double(), ctrl64.double(), s2, "tps")
s3 = dict(sd2)
xr32, src32 = O.tps_transform(xin, ctrl64, s3, "tps")
print("%-40s rel %10.2e %10.2e   max-abs %10.2e %10.2e" % ("sampler only: source coordinates", rel(src2, src64), rel(src32, src64), mx(src2, src64), mx(src32, src64)))
print("%-40s rel %10.2e %10.2e   max-abs %10.2e %10.2e" % ("sampler only: rectified image", rel(xr2.permute(0, 3, 1, 2), xr64), rel(xr32, xr64),
                                                                mx(xr2.permute(0, 3, 1, 2), xr64), mx(xr32, xr64)))
