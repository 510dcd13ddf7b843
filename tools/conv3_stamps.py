"""s_memtime stamps of conv3_c64_sb3_kernel (B = 48, 16x64, 64 -> 64 channels): build csrc/conv3.hip with S2_STAMP 1 first.
Output of the round-6 session: profiles/r06_conv3_sb2_stamps.txt."""
import sys, os, ctypes
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from tatt_amd import ops
from tatt_amd._lib import LIB
dev = torch.device("cuda:0")
LIB.load()
LIB.tatt_conv3_sb_generation(3)        # the stamps live in the one-wave-per-SIMD kernel (build with S2_STAMP 1 in csrc/conv3.hip)
B = 48
x = torch.randn(B, 16, 64, 64, device=dev); w = torch.randn(64, 64, 3, 3, device=dev) * 0.05; b = torch.randn(64, device=dev)
y = torch.empty_like(x)
wsb = ops.repack_weight(w, LIB.tatt_conv3_sb_packing(B, 16, 64, 64, 64, 0, 0))
def run():
    ops.call("tatt_conv3_c64_fwd_sb", ops.P(x), 64, 0, ops.P(wsb), ops.P(b), ops.P(y), B, 16, 64, 64, 0, 0.0, None, None, 0, None, ops.stream())
import numpy as np
def timeit(n=100):
    """GPU time per launch: n launches captured into one hipGraph (no host launch gaps), replayed 5 times"""
    for _ in range(3): run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            for _ in range(n): run()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * n) * 1e3
names = {1: "no MFMA", 2: "no halo loads", 4: "no filter loads", 8: "no stores", 16: "exit at once", 32: "exit after the first staging"}
print("graph-timed: %.2f us" % timeit())
buf = torch.zeros(256 * 8 * 32, dtype=torch.int64, device=dev)
LIB.tatt_conv3_debug_stamps(ctypes.c_void_p(buf.data_ptr()))
run(); torch.cuda.synchronize()
LIB.tatt_conv3_debug_stamps(ctypes.c_void_p(0))
s = buf.cpu().view(256, 8, 32)[:, :4]
rel = (s - s[:, :, 0:1]).numpy()
np.set_printoptions(linewidth=250)
print("cols: start staged0 | k0_beg k0_end k0_bar | k1_beg k1_end k1_bar | k2_beg k2_end k2_bar | epi_beg epi_end")
for wg in (0, 100):
    print("WG %d waves:" % wg); print(rel[wg, :, :13])
print("median over WGs, per wave:"); print(np.median(rel[:, :, :13], axis=0).astype(int))
wc = (s[:, :, 31] - s[:, :, 30]).numpy().astype(np.float64)
life = rel[:, :, 12].astype(np.float64)
print("wall clock per wave: median %.2f us;  s_memtime ticks median %.0f  ->  %.3f GHz" % (np.median(wc) * 0.01, np.median(life), np.median(life) / (np.median(wc) * 10.0)))
w0 = s[:, :, 30].numpy().astype(np.float64); w1 = s[:, :, 31].numpy().astype(np.float64)
print("first wave start -> last wave end: %.2f us; starts spread %.2f us; ends spread %.2f us" % ((w1.max() - w0.min()) * 0.01, (w0.max() - w0.min()) * 0.01, (w1.max() - w1.min()) * 0.01))
