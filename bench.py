#!/usr/bin/env python3
"""Headline benchmark: LR images/s of a full TATT training step (forward + ImageLoss + backward + clip 0.25 + Adam,
dropout ON, STN ON) on synthetic 16x64 -> 32x128 batches, B = 48 per GPU, fp32 (BASELINE.json configs[1]/[2]).

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  `value` is the whole-job aggregate (weak scaling: 48 images per GPU per step).
Extra objects: `roofline` -- the step's DOMINANT kernel, chosen from the committed per-step kernel table of the last rocprofv3 trace
(profiles/step_kernel_table.json) and timed live with HIP events on the launch stream after the timed region; `frac` = ALGORITHMIC
FLOPs (or bytes) / time / the peak of the unit the kernel actually issues on (the bf16 matrix cores for the split-bf16 kernels: the two
extra products per fp32 product are reported separately as `mfma_pipe_util`); `roofline_hbm` / `roofline_mfma` -- the same for the
heaviest kernel on the other roof; `roofline_attn` -- the attention path; `cpu_baseline` -- the CPU oracle timed on this host's cores
(N=1 only, a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0       # same guide: bf16 MFMA, dense (the sparse figure is never used)
PEAK_HBM_GBPS = 8000.0               # same guide: HBM3E spec peak (6.3 TB/s measured achievable)
# SURVEY.md 8d (FlopCounterMode on the reference graph): forward + backward FLOPs per LR image
TILES = {"std": dict(H=16, W=64, batch=48, flop_per_image=7.613e9, stn=True, loss_key="tatt_b48_16x64"),           # configs[1]/[2]
         "large": dict(H=32, W=128, batch=16, flop_per_image=36.66e9, stn=False, loss_key="tatt_b16_32x128")}     # configs[4]
PMC_FILE = os.path.join(ROOT, "profiles", "conv3_ws_pmc.json")      # HBM traffic per kernel (separate rocprofv3 --pmc passes)
TABLE_FILE = os.path.join(ROOT, "profiles", "step_kernel_table.json")   # per-kernel time of ONE replayed step (tools/prof_timeline.py --json)
LOSS_FILE = os.path.join(ROOT, "tests", "golden", "bench_losses.json")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)       # >= 10 warm-up + >= 50 timed steps
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--sustain", type=int, default=450, help="steps replayed AFTER the timed region for `sustained_ms_per_step` (0: skip)")
    ap.add_argument("--no-exact-fp32", action="store_true", help="skip the second Trainer under set_arithmetic('fp32') (`exact_fp32`)")
    ap.add_argument("--tile", default="std", choices=sorted(TILES), help="std: 16x64 LR (headline); large: 32x128 LR, B=16, STN off")
    ap.add_argument("--batch", type=int, default=None, help="LR images per GPU per step (default: 48 std / 16 large)")
    ap.add_argument("--dp-selftest", action="store_true",
                    help="N=1 only: run the data-parallel step (RCCL process group of one rank, staged backward, bucketed all-reduce)")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--tssim", action="store_true",
                    help="the shipped recipe (train_TATT.sh --tssim_loss --rotate_train=5): rotation + second forward + TRI_SSIM")
    ap.add_argument("--no-defer", action="store_true", help="A/B: weight-gradient kernels inline in the backward (no staging)")
    ap.add_argument("--no-side-stream", action="store_true", help="A/B: deferred weight-gradient kernels on the main stream")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=48)
    ap.add_argument("--cpu-baseline-only", action="store_true", help="(internal) child process of the cpu_baseline leg")
    ap.add_argument("--arch", default="tatt", choices=["tatt", "tsrn", "tbsrn", "tatt_tpg"])
    return ap.parse_args()


def make_batch(B, rank, dev, H=16, W=64):
    g = torch.Generator().manual_seed(rank)                         # rank r draws data seed r
    x = torch.rand(B, 4, H, W, generator=g)
    x[:, 3] = (x[:, 3] > 0.5).float()                               # binarised mask channel (dataset/dataset.py:1312-1317)
    hr = torch.rand(B, 4, 2 * H, 2 * W, generator=g)
    tp = torch.softmax(torch.randn(B, 37, 1, 26, generator=g), 1)
    return x.to(dev), tp.to(dev), hr.to(dev)


def first_step_loss(model, x, tp, hr):
    """`ImageLoss(sr, hr).mean() * 100` of the first training step with every nn.Dropout in eval mode, on a COPY of the model (the
    timed model's BatchNorm statistics and seed word stay untouched).  The reference's value for bench.py's own model and rank-0
    batch is stored in tests/golden/bench_losses.json (tools/gen_golden.py, case_bench_losses)."""
    import copy
    from tatt_amd.train import image_loss_mean
    from tatt_amd import functional as Fh
    m = copy.deepcopy(model).train()
    m.infoGen.dropout_on = False
    seed = Fh.seed_tensor(x.device).clone()
    with torch.no_grad():
        sr, _ = m(x, tp)
        loss = float(image_loss_mean(sr, hr, scale=100.0))
    Fh.seed_tensor(x.device).copy_(seed)
    return loss


def dominant_kernel_traffic(key, kernel):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/conv3_ws_pmc.json: 2 x FETCH_SIZE -- the
    gfx950 correction for 16-byte coalesced reads, MI355X_MICROARCH.md -- + WRITE_SIZE), or None if that kernel / shape was not
    profiled (counters cannot be collected from inside a timed run: separate rocprofv3 --pmc passes, tools/pmc_collect.sh)."""
    try:
        return json.load(open(PMC_FILE))["kernels"][kernel][key]["traffic_bytes"]
    except (OSError, KeyError, ValueError):
        return None


ROTATE_BYTES = 320e6                 # buffer sets rotated per timed kernel: more than the 256 MB Infinity Cache in total


def _nsets(bytes_per_launch):
    """How many independent buffer sets a kernel timer rotates through so that a replayed launch reads HBM, not the Infinity Cache."""
    return max(3, int(ROTATE_BYTES // max(bytes_per_launch, 1)) + 1)


def _timed_ms(run, n=50, warm=5):
    """Average duration of one launch with HIP events on the stream it launches on (torch's current stream).  `run` is a callable or a
    LIST of callables on independent buffer sets, cycled through launch by launch (`_nsets`)."""
    runs = run if isinstance(run, (list, tuple)) else [run]
    for i in range(max(warm, len(runs))):
        runs[i % len(runs)]()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(n):
        runs[i % len(runs)]()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def _k_conv3_sb(dev, B, H, W):
    """3x3 convolution 64 -> 64 channels on B x H x W pixels, forward / data-gradient launches of the step (conv3_c64_sb4_kernel:
    split-bf16 operands on the bf16 matrix cores, round 6; conv3_c64_ws16_kernel with tatt_amd.ops.CONV3_SB = False: exact fp32 MFMA)."""
    from tatt_amd import ops
    w = torch.randn(64, 64, 3, 3, device=dev) * 0.05
    b = torch.zeros(64, device=dev)
    px = B * H * W
    nbytes = 2 * px * 64 * 4 + 9 * 64 * 64 * 4
    wl = ops.repack_weight(w, ops.LIB.tatt_conv3_sb_packing(B, H, W, 64, 64, 0, 0) if ops.CONV3_SB else ops._WS_FWD_MODE)
    runs = []
    for _ in range(_nsets(nbytes)):
        x, y = torch.randn(B, H, W, 64, device=dev), torch.empty(B, H, W, 64, device=dev)
        if ops.CONV3_SB:
            runs.append(lambda x=x, y=y: ops.call("tatt_conv3_c64_fwd_sb", ops.P(x), 64, 0, ops.P(wl), ops.P(b), ops.P(y), B, H, W, 64, 0, 0.0,
                                                  None, None, 0, None, ops.stream()))
        else:
            runs.append(lambda x=x, y=y: ops.call(ops._WS_ENTRY, ops.P(x), ops.P(wl), ops.P(b), ops.P(y), B, H, W, 64, 0, 0.0, ops.stream()))
    return dict(ms=_timed_ms(runs), flops=2.0 * px * 576 * 64, bytes=nbytes, split=bool(ops.CONV3_SB), bound="mfma", sets=len(runs),
                what="3x3 conv, 64->64 ch, %d x%dx%d px; fp32 in/out" % (B, H, W))


def _k_conv3_wgrad(dev, B, H, W):
    """Weight gradient of that convolution: dW[tap][ci][co] = sum over pixels (tatt_conv3_c64_wgrad_partial[_sb]); the partial slabs of
    the G persistent groups are part of its algorithmic output."""
    from tatt_amd import ops
    px = B * H * W
    G = min(B * H * (W // 64), ops.CONV3_WGRAD_GROUPS)
    nbytes = 2 * px * 64 * 4 + G * 36864 * 4
    name = "tatt_conv3_c64_wgrad_partial_sb" if ops.CONV3_WGRAD_SB else "tatt_conv3_c64_wgrad_partial"
    runs = []
    for _ in range(_nsets(nbytes)):
        x, dy = torch.randn(B, H, W, 64, device=dev), torch.randn(B, H, W, 64, device=dev)
        part = torch.empty(G * 36864 + G * 64, device=dev)
        runs.append(lambda x=x, dy=dy, part=part: ops.call(name, ops.P(x), ops.P(dy), ops.P(part), ops.P(part[G * 36864:]), B, H, W, 64, 64, G,
                                                           ops.stream()))
    return dict(ms=_timed_ms(runs), flops=2.0 * px * 576 * 64, bytes=nbytes, split=bool(ops.CONV3_WGRAD_SB), bound="mfma", sets=len(runs),
                what="3x3 weight gradient, 64x64 ch, %d x%dx%d px, %d partial slabs" % (B, H, W, G))


def _gru_setup(dev, B, H, W, vertical):
    from tatt_amd import ops
    M = B * H * W
    gi, whh, bhh = torch.randn(M, 192, device=dev), torch.randn(96, 32, device=dev) * 0.2, torch.randn(96, device=dev) * 0.1
    geom = ops.seq_geom(B, H, W, vertical)
    out, gates = ops.gru32_fwd(gi, whh, bhh, whh, bhh, geom, save=True)
    return M, whh, geom, out, gates, torch.randn(M, 64, device=dev)


def _k_gru32_bwd(dev, B, H, W):
    """BPTT of one bidirectional GRU (hidden 32) of a GruBlock, both scan directions of the image averaged (tatt_gru32_bwd2: reads the
    saved gates, h, dout = 1.5 KB per token; writes dgi + the weight-gradient pass's operand fragments = 2 KB per token)."""
    from tatt_amd import ops
    ms, nsets = [], 0
    for vertical in (True, False):
        sets = []
        M, whh, geom, out, gates, dout = _gru_setup(dev, B, H, W, vertical)
        frag = ops.gru_frag_ok(geom)
        per_tok = (256 + 64 + 64 + 192 + 320) * 4 if frag else (256 + 64 + 64 + 192 + 192 + 64) * 4
        sets.append((gates, out, dout))
        for _ in range(_nsets(M * per_tok) - 1):
            sets.append((gates.clone(), out.clone(), dout.clone()))
        fn = ops.gru32_bwd_frag if frag else ops.gru32_bwd
        ms.append(_timed_ms([lambda s=s_: fn(s[0], s[1], s[2], whh, whh, geom) for s_ in sets], 30, 3))
        nsets = len(sets)
        del sets
    return dict(ms=sum(ms) / 2, flops=2.0 * M * 2 * 96 * 32, bytes=M * per_tok, split=False, bound="hbm", sets=nsets,
                what="BiGRU(32) backward recurrence over %d tokens (vertical %.1f us, horizontal %.1f us)" % (M, ms[0] * 1e3, ms[1] * 1e3))


def _k_gru_wgrad(dev, B, H, W):
    """Weight gradients of one GruBlock from the recurrence's operand fragments (tatt_gru_wgrad_frag; K = 128: the concatenated input)."""
    from tatt_amd import ops
    M, whh, geom, out, gates, dout = _gru_setup(dev, B, H, W, True)
    _, frag = ops.gru32_bwd_frag(gates, out, dout, whh, whh, geom)
    G = max(1, min(M // 32, ops.GRU_WGRAD_FRAG_GROUPS, 256))
    nbytes = M * (320 + 128) * 4 + G * 192 * 162 * 4
    runs = []
    for i in range(_nsets(nbytes)):
        fr = frag if i == 0 else frag.clone()
        x, xb = torch.randn(M, 64, device=dev), torch.randn(M, 64, device=dev)
        ws1, ws2 = torch.empty(G * 192 * 129, device=dev), torch.empty(G * 192 * 33, device=dev)
        runs.append(lambda fr=fr, x=x, xb=xb, ws1=ws1, ws2=ws2: ops.call("tatt_gru_wgrad_frag", ops.P(fr), ops.P(x), ops.P(xb), ops.P(ws1),
                                                                         ops.P(ws2), *geom, G, ops.stream()))
    return dict(ms=_timed_ms(runs, 30, 3), flops=2.0 * M * 192 * 160, bytes=nbytes, split=True, bound="hbm", sets=len(runs),
                what="GruBlock weight gradients (192 x 128 + 2 x 96 x 32) over %d tokens, %d partial slabs" % (M, G))


# kernels bench.py can time live, by the name they carry in the per-step table (prefix match)
TIMEABLE = [("conv3_c64_sb4_kernel", _k_conv3_sb), ("conv3_c64_sb3_kernel", _k_conv3_sb), ("conv3_c64_sb_kernel", _k_conv3_sb),
            ("conv3_c64_ws16_kernel", _k_conv3_sb), ("conv3_c64_wgrad_sb2_kernel", _k_conv3_wgrad), ("conv3_c64_wgrad_sb_kernel", _k_conv3_wgrad),
            ("conv3_c64_wgrad_kernel", _k_conv3_wgrad), ("gru32_bwd2_kernel", _k_gru32_bwd), ("gru32_bwd_kernel", _k_gru32_bwd),
            ("gru_wgrad_frag_kernel", _k_gru_wgrad)]


def step_table():
    try:
        return json.load(open(TABLE_FILE))
    except (OSError, ValueError):
        return None


def pick_kernels(table):
    """-> [(table name, timer, per-step us, launches)] of the timeable kernels, heaviest first.  Without a table: the 3x3 convolution."""
    if not table:
        return [("conv3_c64_sb4_kernel", _k_conv3_sb, None, None)]
    agg = {}
    for name, e in table["kernels"].items():
        for key, fn in TIMEABLE:
            if name.startswith(key):
                a = agg.setdefault(key, [key, fn, 0.0, 0])
                a[2] += e["us"]
                a[3] += e["launches"]
                break
    return sorted((tuple(v) for v in agg.values()), key=lambda v: -v[2]) or [("conv3_c64_sb4_kernel", _k_conv3_sb, None, None)]


def roofline_block(name, k, per_step_us, launches, shape_key):
    """The bench line's roofline object for one timed kernel.  achieved = ALGORITHMIC flops (bytes) per launch / average launch time;
    peak = the unit the kernel issues on: the bf16 matrix cores for the split-bf16 kernels (dense 2.5 PFLOP/s), fp32 MFMA otherwise."""
    sec = k["ms"] * 1e-3
    tflops, gbps = k["flops"] / sec / 1e12, k["bytes"] / sec / 1e9
    peak_f = PEAK_BF16_MFMA_TFLOPS if k["split"] else PEAK_FP32_MFMA_TFLOPS
    r = {"bound": k["bound"], "kernel": "%s (%s%s)" % (name, k["what"], "; split-bf16 hi+lo operands, 3 bf16 MFMA products per fp32 product, fp32 "
                                                        "accumulation" if k["split"] else ""),
         "kernel_ms": round(k["ms"], 4), "flops_per_launch": k["flops"], "algorithmic_bytes": k["bytes"],
         "traffic": dominant_kernel_traffic(shape_key, name)}
    if k["bound"] == "mfma":
        r.update(achieved=round(tflops, 2), peak=peak_f, unit="TFLOP/s", frac=round(tflops / peak_f, 4),
                 hbm_gbps=round(gbps, 1), hbm_frac=round(gbps / PEAK_HBM_GBPS, 4))
    else:
        r.update(achieved=round(gbps, 1), peak=PEAK_HBM_GBPS, unit="GB/s", frac=round(gbps / PEAK_HBM_GBPS, 4),
                 tflops=round(tflops, 2), mfma_frac=round(tflops / peak_f, 4))
    if k["split"]:
        r["mfma_pipe_util"] = round(3.0 * tflops / PEAK_BF16_MFMA_TFLOPS, 4)     # issue slots of the bf16 pipe taken (three products per fp32 product)
        r["fp32_equivalent_tflops"] = round(tflops, 2)
    r["timing"] = "HIP events on the launch stream, %d independent buffer sets rotated launch by launch (> 256 MB in total: HBM, not the " \
                  "Infinity Cache)" % k.get("sets", 1)
    if per_step_us is not None and launches:
        # the same kernel as the step runs it (beside the other lane): algorithmic work / its average launch time in the committed trace
        in_ms = per_step_us / launches * 1e-3
        in_frac = (k["flops"] / (in_ms * 1e-3) / 1e12 / peak_f) if k["bound"] == "mfma" else (k["bytes"] / (in_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS)
        r["in_step"] = {"us_per_step": round(per_step_us, 1), "launches_per_step": launches, "us_per_launch": round(in_ms * 1e3, 2),
                        "source": "profiles/step_kernel_table.json (one replayed step of the last rocprofv3 kernel trace; launches of other "
                                  "shapes of the same kernel are averaged in)"}
        r["in_step_frac"] = round(in_frac, 4)
    return r


def time_tp_layer(dev, B, L, S=26):
    """The attention path: one fused TP-interpreter decoder layer (last layer: final norms + attention weights out), forward and backward,
    dropout on, at the benchmark's token count, as the training step runs it: csrc/tplayer2.hip -- operand packing (tplayer2_prep_kernel,
    timed with the forward), tplayer2_fwd_kernel (exact fp32 MFMA: the relu decisions must stay inside fp32 round-off of the reference),
    tplayer2_bwd_kernel (every product -- projections, 26-key attention, weight / key / value gradients -- as split-bf16 on the bf16 matrix
    cores) -- or, with tatt_amd.set_arithmetic("fp32"), the first generation tplayer_kernel (csrc/tplayer.hip, fp32 MFMA 16x16x4,
    vector-ALU softmax).  -> dict for the bench line (`roofline_attn`)."""
    from tatt_amd import ops, functional as Fh
    g = torch.Generator().manual_seed(0)
    r = lambda *s: (torch.randn(*s, generator=g) * 0.3).to(dev)
    lp = (r(192, 64), r(192), r(64, 64), r(64), r(64, 64), r(64), r(64, 64), r(64), r(64) + 1, r(64), r(64) + 1, r(64))
    lnF = (r(64) + 1, r(64))
    seed = Fh.seed_tensor(dev)
    gen2 = bool(ops.TPLAYER_BWD2 and ops.tplayer2_geom(B, L, S)[0])
    tok = B * L
    b_fwd = tok * (64 * 4 * 3 + S * 4)                   # x, qpos in; the layer's output map + the (L, S) attention weights out
    b_bwd = tok * 64 * 4 * 5                             # x, qpos, upstream gradient in; dx, dqpos out
    fw, bw = [], []
    for _ in range(_nsets(b_bwd)):
        x, qpos, K, V, up = r(B, L, 64), r(B, L, 64), r(B, S, 64), r(B, S, 64), r(B, L, 64)
        if gen2:
            pk = ops.tplayer2_prep(lp, K, V)
            f = lambda x=x, qpos=qpos, pk=pk: ops.tplayer2_fwd(x, qpos, pk, lp, lnF, 0.5, 1, 0.1, 0.1, 0.1, seed, 10, 1e-5, False, True, S)
            hm = f()[3]
            fw.append(lambda K=K, V=V, f=f: (ops.tplayer2_prep(lp, K, V), f()))         # the operand packing belongs to the layer's forward
            bw.append(lambda x=x, qpos=qpos, pk=pk, up=up, hm=hm: ops.tplayer2_bwd(x, qpos, pk, lp, lnF, 0.5, 1, 0.1, 0.1, 0.1, seed, 10, 1e-5,
                                                                                   None, up, None, None, True, S, hmask=hm))
        else:
            fw.append(lambda x=x, qpos=qpos, K=K, V=V: ops.tplayer_fwd(x, qpos, K, V, lp, lnF, 0.5, 1, 0.1, 0.1, 0.1, seed, 10, 1e-5, False, True))
            bw.append(lambda x=x, qpos=qpos, K=K, V=V, up=up: ops.tplayer_bwd(x, qpos, K, V, lp, lnF, 0.5, 1, 0.1, 0.1, 0.1, seed, 10, 1e-5, None, up,
                                                                              None, None, True))
    tf = _timed_ms(fw, 30, 3)
    tb = _timed_ms(bw, 30, 3)
    f_fwd = tok * (4 * 2 * 64 * 64 + (2 * 2 * S * 64 if gen2 else 0))     # the four projections (+ QK^T and PV where they run on the matrix cores)
    f_bwd = tok * (12 * 2 * 64 * 64 + (8 if gen2 else 2) * 2 * S * 64)     # recompute + data gradients + weight gradients (+ attention fwd/bwd) + dK / dV
    ach = (f_fwd + f_bwd) / ((tf + tb) * 1e-3) / 1e12
    peak = PEAK_BF16_MFMA_TFLOPS if gen2 else PEAK_FP32_MFMA_TFLOPS
    pipe = "bf16 MFMA, split operands (3 products per fp32 product)" if gen2 else "fp32 MFMA"
    out = {"kernel": "%s (fused cross-attention + LayerNorm + FFN layer, %d x %d query tokens, %d keys)" % (
               "tplayer2_prep_kernel + tplayer2_fwd_kernel + tplayer2_bwd_kernel" if gen2 else "tplayer_kernel<fwd> + tplayer_kernel<bwd>", B, L, S),
           "bound": "mfma", "unit": "TFLOP/s", "achieved": round(ach, 2), "fwd_ms": round(tf, 4), "bwd_ms": round(tb, 4),
           # the forward's products are exact fp32 in both generations (v_mfma_f32_16x16x4_f32)
           "fwd": {"tflops": round(f_fwd / (tf * 1e-3) / 1e12, 2), "peak": PEAK_FP32_MFMA_TFLOPS,
                   "frac": round(f_fwd / (tf * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4), "pipe": "fp32 MFMA"},
           "bwd": {"tflops": round(f_bwd / (tb * 1e-3) / 1e12, 2), "peak": peak, "frac": round(f_bwd / (tb * 1e-3) / 1e12 / peak, 4), "pipe": pipe},
           # continuity with rounds 3-4: fp32-equivalent MFMA work of forward + backward against the fp32 MFMA peak
           "peak": PEAK_FP32_MFMA_TFLOPS, "mfma_util": round(ach / PEAK_FP32_MFMA_TFLOPS, 4),
           "hbm_gbps": round((b_fwd + b_bwd) / ((tf + tb) * 1e-3) / 1e9, 1),
           "hbm_frac": round((b_fwd + b_bwd) / ((tf + tb) * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4),
           "timing": "HIP events, %d buffer sets rotated" % len(fw),
           "launches_per_step": "2 decoder layers + 1 encoder layer, forward and backward"}
    if gen2:
        out["bwd"]["mfma_pipe_util"] = round(3.0 * f_bwd / (tb * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4)
    return out


def time_tbsrn_attention(dev, B, P=1024, h=4, d=32):
    """The dominant kernels of the TBSRN step (configs[3]): the score-free self-attention of one FeatureEnhancer (csrc/sattn2.hip: split
    bf16 on v_mfma_f32_32x32x16_bf16; csrc/sattn.hip under set_arithmetic("fp32")): forward (QK^T and PV: 4 B h P^2 d FLOPs) and backward
    (D; dK, dV; dQ: seven P x P x d products = 14 B h P^2 d), dropout on, keep bits handed from the forward to the backward as the model
    does.  -> (ms forward, ms backward, FLOPs, algorithmic HBM bytes = what each of the four launches must read and write: Q K V -> O; dO O -> D;
    Q K V dO -> dK dV; Q K V dO -> dQ = 17 token tensors, and the keep bits once out, twice in)."""
    from tatt_amd import ops, functional as Fh
    E = h * d
    Q, K, V, dO = (torch.randn(B, P, E, device=dev) for _ in range(4))
    O, lse, ws = torch.empty_like(Q), torch.empty(B, h, P, device=dev), torch.empty(B, h, P, device=dev)
    dQ, dK, dV = torch.empty_like(Q), torch.empty_like(Q), torch.empty_like(Q)
    seed = Fh.seed_tensor(dev)
    sc = d ** -0.5
    bits = torch.empty(B * h * P * (P // 32), device=dev, dtype=torch.int32)
    Fh._sattn_select()
    fwd = lambda: ops.call("tatt_sattn_fwd_bits", ops.P(Q), ops.P(K), ops.P(V), ops.P(O), ops.P(lse), ops.P(bits), B, P, h, sc, 0.1,
                           ops.P(seed), 100, ops.stream())
    bwd = lambda: ops.call("tatt_sattn_bwd_bits", ops.P(Q), ops.P(K), ops.P(V), ops.P(O), ops.P(lse), ops.P(dO), ops.P(bits), ops.P(dQ),
                           ops.P(dK), ops.P(dV), ops.P(ws), B, P, h, sc, 0.1, ops.P(seed), 100, ops.stream())
    tf, tb = _timed_ms(fwd, 10, 2), _timed_ms(bwd, 10, 2)
    return tf, tb, 4.0 * B * h * P * P * d, 14.0 * B * h * P * P * d, 17.0 * B * P * E * 4 + 3.0 * B * h * P * P / 8


def executed_flop_per_image(tile, B):
    """FLOPs the build executes per LR image: the reference graph's count minus the query GRU's input projection, which the build
    hoists out of the recurrence (identical maths: the input is the same at every step).  As written the
    reference spends 2 * W * (64 H) * (3 * 32 H) * 2 directions per image in the forward (x3 with the backward); hoisted it runs once
    per step, i.e. 1/B of that per image."""
    H, W = tile["H"], tile["W"]
    proj = 3.0 * 2.0 * W * (64 * H) * (96 * H) * 2
    return tile["flop_per_image"] - proj * (1.0 - 1.0 / B)


def make_model(arch, tile="std"):
    import tatt_amd
    t = TILES[tile]
    kw = dict(scale_factor=2, width=2 * t["W"], height=2 * t["H"], STN=t["stn"], mask=True, srb_nums=5, hidden_units=32)
    if arch == "tbsrn":
        return tatt_amd.TBSRN(input_channel=4, **kw)
    if arch == "tatt_tpg":                     # the SR generator trained together with its CRNN student prior generator
        from tatt_amd.train import TextPriorSR
        return TextPriorSR(tatt_amd.TSRN_TL_TRANS(**kw), tatt_amd.CRNN(32, 1, 37, 256), teacher=tatt_amd.CRNN(32, 1, 37, 256).eval())
    return (tatt_amd.TSRN_TL_TRANS if arch == "tatt" else tatt_amd.TSRN)(**kw)


def usable_cores():
    """Cores this process may really use: affinity mask and the cgroup CPU quota (a container on a 200-core host with a
    quota of a few cores would otherwise oversubscribe OpenMP by 50x and crawl)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 32))


def cpu_baseline_worker(arch, B, tile="std"):
    """Runs in a CHILD process (no HIP context, bounded by the parent's timeout): the CPU oracle (validated against the
    reference, tests/golden/REPORT.txt) timed on this host -- full training steps (fwd + loss + bwd + clip + Adam, dropout
    on) on the same synthetic workload.  A B=8 step is timed first; the B=`--cpu-batch` step only runs if it is predicted
    to finish in about half a minute."""
    from oracle import tatt_oracle as O
    cores = usable_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(1234)
    sd = make_model(arch, tile).state_dict()
    t = TILES[tile]
    g = torch.Generator().manual_seed(0)
    x, hr = torch.rand(B, 4, t["H"], t["W"], generator=g), torch.rand(B, 4, 2 * t["H"], 2 * t["W"], generator=g)
    tp = torch.softmax(torch.randn(B, 37, 1, 26, generator=g), 1) if arch == "tatt" else None
    kw = dict(tatt=arch == "tatt", stn=t["stn"], drop_on=True, tbsrn=arch == "tbsrn")

    def run(b):
        t0 = time.time()
        O.train_step(sd, x[:b], None if tp is None else tp[:b], hr[:b], **kw)
        return time.time() - t0
    run(2)                                                          # warm-up
    b = min(8, B)
    dt = run(b)
    if b < B and dt * B / b < 40.0:
        b, dt = B, run(B)
    nrep, tot = 1, dt
    while tot < 12.0 and nrep < 16:                                # about 10-30 s of CPU work in total
        tot += run(b)
        nrep += 1
    return {"value": round(b * nrep / tot, 3), "unit": "LR images/s", "cores": cores, "kind": "port",
            "sample": "%d full train step(s) (fwd+loss+bwd+clip+Adam, dropout on) of the CPU oracle at B=%d, fp32, %d threads, "
                      "%.1f s in total; the reference itself measured 3.6 img/s on 8 vCPU (BASELINE.md)" % (nrep, b, cores, tot)}


def cpu_baseline(arch, B, tile="std", timeout=240):
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--arch", arch, "--cpu-batch", str(B), "--tile", tile]
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:                       # never let the reported baseline take the benchmark down
        return {"value": None, "unit": "LR images/s", "cores": usable_cores(), "kind": "port",
                "sample": "CPU oracle leg failed or exceeded %d s: %s" % (timeout, type(e).__name__)}


def main():
    a = parse()
    if a.cpu_baseline_only:
        print(json.dumps(cpu_baseline_worker(a.arch, a.cpu_batch, a.tile)))
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (a.gpus, a.gpus))
    tile = TILES[a.tile]
    if a.batch is None:
        a.batch = tile["batch"]
    if a.tile != "std" and a.arch != "tatt":
        raise SystemExit("--tile large is the TATT configuration (BASELINE.json configs[4])")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    pg = None
    if world > 1 or a.dp_selftest:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # nccl == RCCL on ROCm
        pg = torch.distributed.group.WORLD

    import tatt_amd
    from tatt_amd.train import Trainer
    from __graft_entry__ import build
    build()

    torch.manual_seed(1234)
    model = make_model(a.arch, a.tile).to(dev).train()
    use_graph = (not a.no_graph) and a.warmup >= 3
    x, tp, hr = make_batch(a.batch, rank, dev, tile["H"], tile["W"])
    # known-answer check of the workload before anything is timed: this model, this batch, dropout off -> the reference's loss
    kat = None
    if a.arch == "tatt" and rank == 0 and a.batch == tile["batch"]:
        want = json.load(open(LOSS_FILE))[tile["loss_key"]]
        got = first_step_loss(model, x, tp, hr)
        assert abs(got - want) < 2e-4 * want, "first-step loss %.6f differs from the reference's %.6f" % (got, want)
        kat = {"first_step_loss_dropout_off": round(got, 6), "reference": round(want, 6)}
    recipe = None
    if a.tssim:
        from tatt_amd.train import TssimRecipe
        recipe = TssimRecipe(5.0, seed=rank)
    tr = Trainer(model, use_graph=use_graph, warmup_eager=2, process_group=pg, recipe=recipe, defer_param_grads=not a.no_defer, side_stream=not a.no_side_stream)
    if a.arch != "tatt":
        tp = None                      # tsrn / tbsrn take no prior; tatt_tpg computes it from the LR image with the CRNN student

    def barrier():
        if pg is not None:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    graph_ok = use_graph                        # a failed hipGraph capture raises: the line below never reports eager as graph
    for _ in range(a.warmup):
        tr.step(x, tp, hr)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = tr.step(x, tp, hr)
    t_issue = time.perf_counter() - t0             # host time to ISSUE the steps (graph launches); the GPU may still be running
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t)
    loss_v = float(loss)
    assert loss_v == loss_v, "loss is NaN"
    from tatt_amd import functional as _Fh
    _Fh.sync_check()                                 # launches that synchronise their work-groups in flight bound every spin: an expired one voids the run
    # after the timed region: a long replay (>= 2 s of GPU time: clocks and thermals settle, the driver's busy sampler sees it) ...
    coll = None
    if pg is not None:
        # collectives profiled OUTSIDE the timed region, over a bounded number of steps: an event pair around every wait for a collective
        # (the time the step's stream sat blocked behind it) and around every pass group
        tr.profile_collectives = True
        for _ in range(min(20, a.steps)):
            tr.step(x, tp, hr)
        barrier()
        coll = tr.collective_report()
        tr.profile_collectives = False
    sustained = None
    if a.sustain > 0:
        barrier()
        t1 = time.perf_counter()
        for _ in range(a.sustain):
            tr.step(x, tp, hr)
        barrier()
        sustained = (time.perf_counter() - t1) / a.sustain * 1e3
        _Fh.sync_check()
    # ... and the same step with exact fp32 products everywhere (the reference's arithmetic: model/tsrn.py:877,885,1071 are plain
    # nn.Conv2d / nn.GRU): a second model + Trainer built under set_arithmetic("fp32") -- captured graphs keep their kernels
    exact = None
    if world == 1 and pg is None and not a.no_exact_fp32 and a.arch in ("tatt", "tsrn") and tatt_amd.get_arithmetic() == "split_bf16":
        tatt_amd.set_arithmetic("fp32")
        try:
            torch.manual_seed(1234)
            m2 = make_model(a.arch, a.tile).to(dev).train()
            rec2 = None
            if a.tssim:
                from tatt_amd.train import TssimRecipe
                rec2 = TssimRecipe(5.0, seed=rank)
            tr2 = Trainer(m2, use_graph=use_graph, warmup_eager=2, recipe=rec2, defer_param_grads=not a.no_defer, side_stream=not a.no_side_stream)
            for _ in range(a.warmup):
                tr2.step(x, tp, hr)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(a.steps):
                l2 = tr2.step(x, tp, hr)
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t1
            assert float(l2) == float(l2)
            _Fh.sync_check()
            exact = {"ms_per_step": round(dt2 / a.steps * 1e3, 4), "value": round(a.batch * a.steps / dt2, 2), "unit": "LR images/s",
                     "steps": a.steps, "warmup": a.warmup,
                     "arithmetic": "tatt_amd.set_arithmetic('fp32'): every product of the step exact fp32 (v_mfma_f32_* / VALU), same model, batch, "
                                   "optimiser and launch mode as the headline"}
            del tr2, m2
        finally:
            tatt_amd.set_arithmetic("split_bf16")

    if rank == 0:
        ms = dt / a.steps * 1e3
        ips = a.batch * world * a.steps / dt
        from tatt_amd import ops as _ops
        shape_key = "B%d_%dx%d" % (a.batch, tile["H"], tile["W"])
        roof_other = None
        if a.arch == "tbsrn":
            tf, tb, ff, fb, kbytes = time_tbsrn_attention(dev, a.batch, tile["H"] * tile["W"])
            kms, kflops = tf + tb, ff + fb
            ach = kflops / (kms * 1e-3) / 1e12
            from tatt_amd import functional as _Fh
            sb = _Fh.SATTN_SB
            peak = PEAK_BF16_MFMA_TFLOPS if sb else PEAK_FP32_MFMA_TFLOPS
            roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4), "traffic": dominant_kernel_traffic(shape_key, "sattn2_fwd_bwd" if sb else "sattn_fwd_bwd"),
                    "algorithmic_bytes": kbytes,
                    "hbm_gbps": round(kbytes / (kms * 1e-3) / 1e9, 1),
                    "kernel": ("sattn2_fwd_kernel<2> + sattn_prep_kernel + sattn2_bwd_kv_kernel<2> + sattn2_bwd_q_kernel<2>: score-free self-attention "
                               "of one FeatureEnhancer (B = %d, P = %d, 4 heads x 32; split bf16 on v_mfma_f32_32x32x16_bf16 -- three matrix "
                               "products per fp32 product: mfma_pipe_util = 3 x frac; online softmax, dropout on with keep bits); 5 per step. "
                               "`achieved` counts fp32-equivalent FLOPs.  The kernels are bound by VALU + MFMA issue together (the softmax / "
                               "dropout / operand-split arithmetic is ~4/5 of the issue cycles; the two pipes do not overlap on gfx950: "
                               "profiles/r06_valu_mfma_coissue.txt), not by the matrix peak" if sb else
                               "sattn_fwd_kernel + sattn_bwd_kv_kernel + sattn_bwd_q_kernel: score-free self-attention of one FeatureEnhancer "
                               "(B = %d, P = %d, 4 heads x 32; fp32 MFMA 16x16x4, online softmax, dropout on); 5 per step") % (
                                   a.batch, tile["H"] * tile["W"]),
                    "mfma_pipe_util": round((3.0 if sb else 1.0) * ach / peak, 4),
                    "kernel_ms": round(kms, 4), "fwd_ms": round(tf, 4), "bwd_ms": round(tb, 4), "flops_per_launch": kflops,
                    "fwd_tflops": round(ff / (tf * 1e-3) / 1e12, 2), "bwd_tflops": round(fb / (tb * 1e-3) / 1e12, 2)}
        else:
            cands = pick_kernels(step_table() if (a.arch == "tatt" and a.tile == "std") else None)
            name, fn, us, nl = cands[0]
            k0 = fn(dev, a.batch, tile["H"], tile["W"])
            roof = roofline_block(name, k0, us, nl, shape_key)
            for name2, fn2, us2, nl2 in cands[1:]:           # the heaviest kernel on the OTHER roof
                k2 = fn2(dev, a.batch, tile["H"], tile["W"])
                if k2["bound"] != k0["bound"]:
                    roof_other = ("roofline_" + k2["bound"], roofline_block(name2, k2, us2, nl2, shape_key))
                    break
        attn = time_tp_layer(dev, a.batch, tile["H"] * tile["W"]) if a.arch in ("tatt", "tatt_tpg") else None
        out = {
            "metric": "LR images/s (train fwd+bwd+clip+Adam) at %dx%d->%dx%d" % (tile["H"], tile["W"], 2 * tile["H"], 2 * tile["W"]),
            "value": round(ips, 2), "unit": "LR images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms, 4), "sustained_ms_per_step": None if sustained is None else round(sustained, 4),
            "sustained_steps": a.sustain if sustained is not None else 0, "exact_fp32": exact,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "fp32" + (" (split-bf16 MFMA operands in conv3 / conv3-wgrad / conv9 / tokgemm / gru-wgrad%s)"
                               % (" / query-GRU recurrence and dW_hh / TP-layer backward" if _Fh.QGRU_CHAIN_SB and a.arch in ("tatt", "tatt_tpg") and a.tile == "std"
                                  else (" / self-attention / FeatureEnhancer projections" if a.arch == "tbsrn" and _Fh.SATTN_SB else ""))
                               if _ops.CONV3_SB else ""),
            "data": "synthetic",
            "config": {"workload": "TATT (TSRN_TL_TRANS, STN %s, dropout on) train step, batch %d/GPU, %dx%d LR -> %dx%d SR, "
                                   "ImageLoss + clip 0.25 + Adam(1e-3,(0.5,0.999))" % (
                                       "on" if tile["stn"] else "off", a.batch, tile["H"], tile["W"], 2 * tile["H"], 2 * tile["W"])
                       + (" + rotate_train 5 deg + second forward + TRI_SSIM (train_TATT.sh recipe)" if a.tssim else "")
                       if a.arch == "tatt" else "%s train step, batch %d/GPU" % (a.arch.upper(), a.batch),
                       "global_batch": a.batch * world, "parallelism": "dp%d" % world + (" (self-test: RCCL group of one rank)" if a.dp_selftest else ""),
                       "launch": ("hipGraph replay" if graph_ok else "eager") + ("" if a.no_defer else ", staged backward" + ("" if a.no_side_stream else " on 2 streams")), "final_loss": round(loss_v, 5),
                       "host_issue_ms_per_step": round(t_issue / a.steps * 1e3, 3), "known_answer": kat,
                       "arithmetic": "fp32 storage, accumulation and results throughout (the reference's arithmetic)"
                                     + ("; the 3x3 convolutions, the 9x9 convolutions at the image end, the GruBlock input projections, the GruBlock "
                                        "weight gradients, the backward of the TP-interpreter layers and the recurrent products and recurrent "
                                        "weight gradient of the query GRU evaluate "
                                        "every fp32 product as three bf16 matrix-core products of hi/lo operand halves (2^-16 relative, "
                                        "measured 1e-6 on SR: profiles/r03_split_bf16_probe.txt; query GRU 5e-6 of the fp32 result: "
                                        "tests/test_kernels_gpu.py)" if _ops.CONV3_SB else ""),
                       # algorithmic = the reference graph's FLOP count; executed = minus the query-GRU input projection
                       # the build hoists out of the recurrence
                       "whole_step_tflops": ({"algorithmic": round(ips * tile["flop_per_image"] / 1e12, 2),
                                              "executed": round(ips * executed_flop_per_image(tile, a.batch) / 1e12, 2),
                                              "note": "mixed pipes (about a third of the FLOPs run as split-bf16 on the bf16 matrix cores, "
                                                      "the rest on fp32 MFMA / VALU): quoted against no single peak"}
                                             if a.arch == "tatt" else None)},
            "roofline": roof,
        }
        if roof_other is not None:
            out[roof_other[0]] = roof_other[1]
        if attn is not None:
            out["roofline_attn"] = attn
        if coll is not None:
            out["collectives"] = coll
        if world == 1 and not a.no_cpu_baseline and a.arch != "tatt_tpg":
            out["cpu_baseline"] = cpu_baseline(a.arch, a.cpu_batch if a.tile == "std" else min(a.cpu_batch, a.batch), a.tile)
        line = json.dumps(out)
    if pg is not None:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()      # (RCCL prints its version banner when the communicator goes away)
    if rank == 0:
        sys.stdout.flush()
        print(line, flush=True)                          # the ONE JSON line: last thing on stdout
    if pg is not None:
        # RCCL 2.26 prints a version banner to stdout from a destructor at interpreter teardown -- after the JSON line.  Everything
        # is flushed and the process group is gone: leave without running the teardown handlers.
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
