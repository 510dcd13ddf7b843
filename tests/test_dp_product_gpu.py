"""Data parallelism of the PRODUCT generator with world > 1 (BASELINE.json configs[2] at the scale one GPU allows): two processes, one
MI355X, TSRN_TL_TRANS (STN on, dropout off), 2 LR images per rank, one Trainer step each with a real process group between them.
Reference semantics (interfaces/base.py:386-396 nn.DataParallel + super_resolution.py:1072-1085): every replica normalises with its own
batch statistics and runs its own batch-axis query GRU; the gradients are averaged; every replica then clips and applies Adam
identically.  Expected values come from the CPU oracle: the gradient of EACH shard computed separately (a single-process B = 4 step is
the wrong reference: BatchNorm and the query GRU see the shard, not the global batch), averaged, clipped, stepped."""
import os
import socket
import subprocess
import sys

import pytest
import torch

from oracle import tatt_oracle as O
from oracle.fixtures import randomize_state_dict, make_inputs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(backend, tmp):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), DP_BACKEND=backend,
                   DP_OUT=str(tmp), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_worker.py")], env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    ok = True
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
            ok = False
        outs.append(o)
        ok = ok and p.returncode == 0
    return ok, outs


def test_world2_product_step_matches_per_shard_oracle(dev, tmp_path):
    # RCCL first (does it accept two ranks on one device?); the answer is recorded, gloo on device tensors is the fallback
    ok, outs = _launch("nccl", tmp_path)
    backend = "nccl"
    if not ok:
        print("nccl with two ranks on one device: refused --", outs[0][-300:].replace("\n", " | "))
        for f in tmp_path.glob("rank*.pt"):
            f.unlink()
        ok, outs = _launch("gloo", tmp_path)
        backend = "gloo"
    assert ok, "\n".join(o[-2000:] for o in outs)
    print("world-2 step ran on backend:", backend)
    r0, r1 = (torch.load(tmp_path / ("rank%d.pt" % r), weights_only=False) for r in range(2))
    assert r0["world"] == 2 and r1["world"] == 2 and r0["backend"] == backend
    # ---- start-up: rank 0's weights everywhere, packed filters rebuilt from them
    import tatt_amd
    STD = dict(scale_factor=2, width=128, height=32, STN=True, mask=True, srb_nums=5, hidden_units=32)
    torch.manual_seed(1234)
    ref = tatt_amd.TSRN_TL_TRANS(**STD)
    sd0 = randomize_state_dict(ref.state_dict(), seed=0)
    sd0 = {k: (v * 1.001 if O.is_param(k) else v) for k, v in sd0.items()}      # (the worker's second broadcast: rank 0 scaled its weights)
    for k, v in sd0.items():
        assert torch.equal(r0["sd_start"][k], v), k
        assert torch.equal(r1["sd_start"][k], v), k                  # rank 1 started from other weights: the broadcast replaced them
    assert not r0["w_changed_by_broadcast"] and r1["w_changed_by_broadcast"]
    for r in (r0, r1):       # the cached packed filter (same buffer, same torch version counter) was rebuilt from the broadcast weights
        assert r["packed_same_buffer"] and r["packed_changed"] and r["packed_fresh"]
    # ---- the protocol: three all-reduces in bucket-completion order, same on both ranks
    assert r0["stages"] == ["trunk", "srb4", "srb3", "srb2", "srb1", "srb0", "tp", "first", "stn"]
    assert r0["reduce_log"] == [(6, 0, 5), (7, 6, 6), (8, 7, 8)] and r1["reduce_log"] == r0["reduce_log"]
    # ---- the arithmetic: per-shard oracle gradients, averaged, clipped, Adam
    x, tp, hr = make_inputs(4, seed=60)
    sd0 = {k: v.clone() for k, v in sd0.items()}
    losses, grads, stats = [], [], []
    for r in range(2):
        sl = slice(2 * r, 2 * r + 2)
        loss, g, new_sd, _, _, _ = O.train_step(sd0, x[sl], tp[sl], hr[sl], tatt=True, stn=True)
        losses.append(float(loss))
        grads.append(g)
        stats.append(new_sd)
    assert abs(r0["loss"] - losses[0]) < 1e-4 * abs(losses[0]) and abs(r1["loss"] - losses[1]) < 1e-4 * abs(losses[1])
    avg = {k: (grads[0][k] + grads[1][k]) * 0.5 for k in grads[0] if grads[0][k] is not None}
    clipped, total = O.clip_grad_norm(avg)
    assert abs(r0["grad_norm"] - float(total)) < 2e-3 * float(total) and abs(r1["grad_norm"] - r0["grad_norm"]) < 1e-6 * r0["grad_norm"]
    import numpy as np
    noise = set(np.load(os.path.join(ROOT, "tests", "golden", "tatt_train_b4.npz"))["noise_keys"].tolist())
    bad = []
    for k, g in clipped.items():
        p1, m1, v1 = O.adam_step(sd0[k], g, torch.zeros_like(g), torch.zeros_like(g), 1, 1e-3)
        for r, res in enumerate((r0, r1)):
            if k in noise:
                continue
            dm = float((res["m"][k] - m1).abs().max()) / (float(m1.abs().max()) + 1e-12)
            dp = float((res["sd_end"][k].float() - p1).abs().mean())
            if dm > 2e-2 or dp > 2e-4:
                bad.append((r, k, dm, dp))
    assert not bad, bad[:8]
    # both ranks end with the same weights; BatchNorm statistics are per replica (different shards -> different running stats)
    for k in r0["sd_end"]:
        if "running_" in k or k.endswith("num_batches_tracked"):
            continue
        assert torch.equal(r0["sd_end"][k], r1["sd_end"][k]), k
    k = "block2.bn1.running_mean"
    assert float((r0["sd_end"][k] - stats[0][k]).abs().max()) < 1e-5 and float((r1["sd_end"][k] - stats[1][k]).abs().max()) < 1e-5
    assert float((r0["sd_end"][k] - r1["sd_end"][k]).abs().max()) > 1e-6


# ---------------------------------------------------------------------------------------------------------------------------------
# no node with more than one GPU has ever run this tree, and an all-reduce of a one-rank RCCL group holds no
# CUs.  The nearest stand-in for a live collective's channel kernels is tatt_cu_holder (work-groups of 512 threads that sit resident):
# here they are launched on a stream of their own right where every bucket's all-reduce goes on the wire, while the data-parallel
# step (staged backward, bucketed all-reduce, the launches that synchronise 256 work-groups in flight -- query-GRU chains, STN head)
# runs beside them.
_DP_HOLDER_WORKER = r'''
import os, sys, torch
sys.path.insert(0, os.environ["TATT_ROOT"])
import tatt_amd
from tatt_amd import ops, functional as Fh
from tatt_amd.train import Trainer
from oracle.fixtures import randomize_state_dict, make_inputs
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
STD = dict(scale_factor=2, width=128, height=32, STN=True, mask=True, srb_nums=5, hidden_units=32)
B, NSTEP = 48, 3
out = {}
for groups in (0, 16, 32, 64):
    torch.manual_seed(1234)
    m = tatt_amd.TSRN_TL_TRANS(**STD)
    m.load_state_dict(randomize_state_dict(m.state_dict()))
    m = m.to(dev).train()
    m.infoGen.dropout_on = False
    Fh.set_seed(dev, 99)
    tr = Trainer(m, use_graph=False, process_group=torch.distributed.group.WORLD)
    side = torch.cuda.Stream()
    sink = torch.zeros(4, dtype=torch.int32, device=dev)
    reduce0 = tr._reduce

    def reduce_with_holders(lo, hi, after_pass, groups=groups):
        if groups:                                   # resident for ~0.4 ms from the moment the bucket goes on the wire
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                ops.call("tatt_cu_holder", groups, 40000, 65536, ops.P(sink), ops.stream())
        reduce0(lo, hi, after_pass)
    tr._reduce = reduce_with_holders
    losses = []
    for i in range(NSTEP):
        x, tp, hr = make_inputs(B, seed=70 + i)
        losses.append(float(tr.step(x.to(dev), tp.to(dev), hr.to(dev))))
    torch.cuda.synchronize()
    tatt_amd.sync_check()
    out[groups] = {"losses": losses, "p": tr.flat_p.detach().cpu().clone(), "reduces": list(tr.reduce_log)}
torch.save(out, os.environ["DP_OUT"])
torch.distributed.destroy_process_group()
'''


def test_dp_step_beside_resident_cu_holders(dev, tmp_path):
    """Three data-parallel steps (B = 48, one-rank RCCL group: same code path as N ranks -- staged backward, three bucketed all-reduces) with
    16 / 32 / 64 CU-holder work-groups (64 KB of LDS each) made resident at every all-reduce: no bounded wait may expire
    (tatt_amd.sync_check) and losses and weights must equal the run without holders bit for bit."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    outf = tmp_path / "holders.pt"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TATT_ROOT=ROOT, DP_OUT=str(outf), HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", _DP_HOLDER_WORKER], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=900)
    assert p.returncode == 0, p.stdout[-3000:]
    res = torch.load(outf, weights_only=False)
    base = res[0]
    assert len(base["reduces"]) == 3
    for groups in (16, 32, 64):
        assert res[groups]["reduces"] == base["reduces"]
        assert res[groups]["losses"] == base["losses"], (groups, res[groups]["losses"], base["losses"])
        assert torch.equal(res[groups]["p"], base["p"]), groups
