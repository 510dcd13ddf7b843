"""GPU-box diagnostic: cycle breakdown inside the production 3x3 conv kernel (s_memtime instrumentation)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tatt_amd import ops
dev = torch.device("cuda:0")
B = 48
for Cin, Cout in ((64, 64), (64, 256), (256, 64)):
    x = torch.randn(B, 16, 64, Cin, device=dev)
    w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
    wt = ops.repack_weight(w, 2)
    y = torch.empty(B, 16, 64, Cout, device=dev)
    prof = torch.zeros(256 * 4 * 6, dtype=torch.int64, device=dev)
    run = lambda: ops.call("tatt_conv3_c64_fwd_t", ops.P(x), ops.P(wt), None, ops.P(y), B, 16, 64, Cin, Cout, 0, 0.0, ops.stream())
    for _ in range(3):
        run()
    ops.call("tatt_conv3_set_prof", ops.P(prof))
    run()
    torch.cuda.synchronize()
    ops.call("tatt_conv3_set_prof", None)
    p = prof.reshape(256, 4, 6).double().cpu()
    names = ["issue-loads", "mfma-block", "publish", "barrier", "epilogue", "total"]
    items = (B * 16 * (Cout // 64)) * (Cin // 64) / 256.0
    print("Cin=%d Cout=%d: %.1f work items (x9 taps) per work-group" % (Cin, Cout, items))
    for i, nme in enumerate(names):
        v = p[:, :, i]
        print("  %-12s mean %9.0f cycles/wave  (per tap %7.1f)   min %9.0f max %9.0f" % (nme, v.mean(), v.mean() / (items * 9), v.min(), v.max()))
