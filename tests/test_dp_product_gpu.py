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
