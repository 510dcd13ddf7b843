"""GPU box: 12 training steps at lr 1e-5 (the drift test of tests/test_model_gpu.py) with exact fp32 products against the split-bf16 default,\nwith the 9x9 families switched back to fp32 one by one: which family moves the loss trajectory how far."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import tatt_amd
from tatt_amd import ops
import test_model_gpu as T
dev = torch.device("cuda:0")
def run(n, lr):
    return T._run_steps(dev, n, 8, False, use_graph=False, lr=lr)
tatt_amd.set_arithmetic("fp32")
ref = run(12, 1e-5)
tatt_amd.set_arithmetic("split_bf16")
for name, c4, wg, c9 in (("all sb", 1, 1, 1), ("no c4", 0, 1, 1), ("no wgrad", 1, 0, 1), ("64->4 only", 0, 0, 1), ("no conv9 sb", 0, 0, 0)):
    ops.CONV9_SB_C4, ops.CONV9_SB_WGRAD, ops.CONV9_SB = bool(c4), bool(wg), bool(c9)
    out = run(12, 1e-5)
    rels = [abs(a - b) / abs(a) for a, b in zip(ref[0], out[0])]
    dp = float((ref[1]["p"] - out[1]["p"]).abs().max()); rp = float((ref[1]["p"] - out[1]["p"]).norm() / ref[1]["p"].norm())
    print("%-12s loss rel max %.3e (per step: %s) weights max %.3e l2 %.3e" % (name, max(rels), " ".join("%.1e" % r for r in rels), dp, rp), flush=True)
