"""Differentiable operators of the TATT hot path: each `torch.autograd.Function` runs hand-written HIP
kernels (tatt_amd.ops -> libtatt_hip.so) in forward AND backward.  torch.autograd is only the tape.

Feature maps are (B, H, W, C) contiguous tensors ("NHWC"); token matrices are their (B*H*W, C) views.
"""
from __future__ import annotations

import math
from typing import Optional

import weakref

import torch
from torch.autograd import Function

from . import ops
from .ops import ACT_NONE, ACT_RELU, ACT_MISH, ACT_TANH

_SEEDS = {}
_CUR_SEED = {}


def seed_tensor(device) -> torch.Tensor:
    """Device-resident dropout seed word of `device` (hipGraph-replay safe: kernels read it from memory)."""
    key = str(device)
    if key not in _SEEDS:
        _SEEDS[key] = torch.tensor([0x1234ABCD5678EF01 & 0x7FFFFFFFFFFFFFFF], dtype=torch.int64, device=device)
    return _SEEDS[key]


def set_seed(device, value: int):
    """Re-seed the dropout stream of `device` (e.g. base_seed mixed with the data-parallel rank)."""
    seed_tensor(device).fill_(int(value) & 0x7FFFFFFFFFFFFFFF)
    _CUR_SEED.pop(str(device), None)


def next_dropout_step(device):
    """Advance the seed word without starting a forward (operator-level tests)."""
    ops.bump_seed(seed_tensor(device))
    _CUR_SEED.pop(str(device), None)


def begin_training_forward(device) -> torch.Tensor:
    """Called by a generator at the start of every TRAINING forward: advances the device's seed word and returns a private
    snapshot of it.  Every dropout site of this forward -- and of its backward, whenever that runs, even after further
    forwards -- reads the snapshot, so masks differ from call to call without any help from the training loop and the
    backward regenerates exactly the masks of its own forward (one extra 1-thread launch; replays correctly in a hipGraph)."""
    g = seed_tensor(device)
    snap = torch.empty_like(g)
    ops.bump_seed(g, snap)
    _CUR_SEED[str(device)] = snap
    return snap


def current_seed(device) -> torch.Tensor:
    """Seed word the dropout sites read: the snapshot of the forward in flight, else the device's seed word."""
    k = str(device)
    return _CUR_SEED[k] if k in _CUR_SEED else seed_tensor(device)


# --------------------------------------------------------------------------------------------------
# parameter gradients off the critical path
# --------------------------------------------------------------------------------------------------
class _DeferredParamGrads:
    """Weight / bias gradients (and their split-K reductions) and the whole backward of the query GRU feed nothing but the
    optimiser: a third of a training step's kernel time that the activation-gradient chain never waits for.  With `enabled`
    (the Trainer switches it on around a backward stage) an operator's backward does NOT compute them: it hands a closure to
    `submit`, returns None to autograd for those inputs, and the Trainer `flush`es the closures at the end of the stage, which
    assigns the results to `param.grad` directly.  The activation-gradient chain then runs without the small reduction kernels
    between its links (4 % of the step), and a data-parallel bucket is complete -- and on the wire -- one stage earlier.

    Operands are kept referenced until `release()`."""

    def __init__(self):
        self.enabled = False
        self.stage = 0                   # index of the backward stage whose main lane is running (set by the Trainer)
        self.due_of = {}                 # id(parameter) -> index of the stage (= gradient bucket) its gradient belongs to
        self.immediate = frozenset()     # id(parameter): run the closure AT ONCE on `side_stream`, beside the main lane that produced it
        self.side_stream = None          # the Trainer's second stream while a stage's main lane runs with two lanes
        self._pending = []
        self._keep = []

    @staticmethod
    def _run(fn):
        r = fn()
        if hasattr(r, "send"):                       # generator closure: nothing to wait for, run it through
            try:
                while True:
                    next(r)
            except StopIteration as stop:
                r = stop.value
        return r

    @staticmethod
    def _assign(params, grads):
        for p, g in zip(params, grads):
            if p is not None and g is not None:
                if p.grad is None:
                    p.grad = g
                elif g.is_cuda and g.dtype == torch.float32 and p.grad.shape == g.shape:
                    p.grad = ops.add_n([p.grad, g])          # (a second forward of the same step: the shipped recipe)
                else:
                    p.grad = p.grad + g

    def submit(self, params, fn, *keep, lag=0):
        """params: tuple of leaf tensors (or None); fn() -> tuple of their gradients (or None), same order (or a generator that
        yields once between its split-K GEMMs and their consumer and returns the tuple).  The closure runs with the side lane of
        the current stage, or of a later one: `lag` stages later, and not before the stage whose gradient bucket holds its
        parameters (`due_of`, from the model's grad_buckets(): the query GRU's 47 dependent launches and the 9x9 output
        convolution's 0.4 ms weight gradient are filed under later buckets, where the main lane beside them has room)."""
        if not self.enabled or any(p is not None and not p.is_leaf for p in params):
            return self._run(fn)
        if (self.side_stream is not None and lag == 0 and any(p is not None and id(p) in self.immediate for p in params)
                and all(p is None or self.due_of.get(id(p), 0) <= self.stage for p in params)):
            # lag 0: run AT ONCE on the side lane of the producing stage's OWN pass, behind what that lane already holds (the deepest
            # STN convolutions' weight gradients in the last pass, whose own work otherwise all follows its main lane).  Same stream
            # as every side lane, so the gather of the bucket and the per-pass join order it; operands stay referenced until release().
            # (A THIRD stream for this was measured, round 5: the graph executor then queues the main lane's kernels behind the side
            # lane's -- 0.17 ms slower, profiles/r05_tail_lane_ab.txt.)
            main = torch.cuda.current_stream(self.side_stream.device)
            self.side_stream.wait_stream(main)
            with torch.cuda.stream(self.side_stream):
                self._assign(params, self._run(fn))
            self._keep.append(keep)
            return (None,) * len(params)
        due = self.stage + lag
        for p in params:                 # a parameter filed under a LATER bucket: its kernels run with that stage's side lane
            if p is not None:
                due = max(due, self.due_of.get(id(p), 0))
        self._pending.append((due, params, fn))
        self._keep.append(keep)
        return (None,) * len(params)

    def flush(self, upto=None):
        """Run the pending closures due at stage <= `upto` (all of them if None) in submission order on the CURRENT stream;
        results become / are added to `.grad`.  While they run, the split-K reductions behind their weight-gradient GEMMs are only
        registered and then summed by one launch per 36 (ops.reduce_defer); a closure that consumes such a result itself is a
        GENERATOR: it yields once after issuing its GEMMs and is resumed after the batched reduction."""
        pending = [e for e in self._pending if upto is None or e[0] <= upto]
        if not pending:
            return
        self._pending = [e for e in self._pending if not (upto is None or e[0] <= upto)]
        results = []
        ops.reduce_defer(True)
        try:
            for _, params, fn in pending:
                r = fn()
                if hasattr(r, "send"):                # generator: run up to its yield
                    next(r)
                results.append((params, r))
        finally:
            ops.reduce_defer(False)                   # one launch per 36 registered reductions
        for params, r in results:
            if hasattr(r, "send"):
                try:
                    next(r)
                    raise RuntimeError("a deferred gradient closure may yield only once")
                except StopIteration as stop:
                    r = stop.value
            self._assign(params, r)

    def release(self):
        assert not self._pending, "deferred parameter gradients were never flushed"
        self._keep.clear()


# One scheduler per PROCESS, on purpose: torch runs every backward node of a device in its autograd worker thread while the forward
# and the Trainer run in the caller's thread, so the state of the step in flight (pending closures, stage, dirty streams) must be
# visible across threads -- a thread-local would be empty exactly where it is read.  A process therefore drives ONE training step at
# a time (several trainers take turns: every step leaves the state clean); one process per GPU is the deployment model anyway.
SIDE = _DeferredParamGrads()


class _ForwardFork:
    """A short parallel branch in the FORWARD pass: work that depends on parameters only (the query GRU: 48 dependent launches)
    is issued on a second stream behind an event and joined right before its first consumer, so that it runs beside the STN head,
    the first convolution and the text encoder.  Inside the step's hipGraph this is a parallel branch (the executor overlaps short
    branches, tools/graph_sched_probe.py).  The branch reads parameters only and its outputs are first read after the join, so no
    tensor crosses the streams unordered.  Enabled by the Trainer together with its second stream."""

    def __init__(self):
        self.enabled = False
        self._streams = {}
        self._dirty = set()

    def run(self, ref, fn):
        if not (self.enabled and ref.is_cuda):
            return fn()
        k = str(ref.device)
        if k not in self._streams:
            self._streams[k] = torch.cuda.Stream(device=ref.device)
        side, main = self._streams[k], torch.cuda.current_stream(ref.device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            out = fn()
        self._dirty.add(k)
        return out

    def join(self, device):
        k = str(device)
        if k in self._dirty:
            torch.cuda.current_stream(device).wait_stream(self._streams[k])
            self._dirty.discard(k)


FWD_FORK = _ForwardFork()
# a second, independent branch (own stream): the student recogniser's pass of TextPriorSR, beside the STN head and the first convolution;
# joined where the generator first reads the text prior (tsrn._trunk_forward)
FWD_FORK_B = _ForwardFork()


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


# --------------------------------------------------------------------------------------------------
class Conv2dFn(Function):
    """nn.Conv2d (stride 1, 'same') on a (B,H,W,Cin)-indexed tensor (any strides) -> (B,H,W,Cout) contiguous,
    optional fused output activation (none / tanh)."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, any_width=False):
        Cout, Cin, KH, KW = weight.shape
        y = ops.conv2d_forward(x, weight, bias, act, any_width)
        ctx.save_for_backward(x, weight, y if act != ACT_NONE else None)
        ctx.act = act
        ctx.any_width = any_width
        ctx.has_bias = bias is not None
        ctx.leaves = (weight, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        Cout, Cin, KH, KW = weight.shape
        dy = _c(dy)
        if ctx.act != ACT_NONE:
            dy = ops.act_bwd(y, dy, ctx.act, True)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.conv2d_dgrad(dy, weight, ctx.any_width)
        want_dw, want_db = ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        aw = ctx.any_width

        def param_grads():
            if want_dw and want_db:
                return ops.conv_wgrad(x, dy, Cout, KH, KW, want_db=True, any_width=aw)
            dw = ops.conv_wgrad(x, dy, Cout, KH, KW, any_width=aw) if want_dw else None
            db = ops.colsum(dy.reshape(-1, Cout)) if want_db else None
            return dw, db
        wleaf, bleaf = ctx.leaves
        if want_dw and want_db and SIDE.enabled and (SIDE.due_of.get(id(wleaf)) != SIDE.due_of.get(id(bleaf))
                                                     or (id(wleaf) in SIDE.immediate) != (id(bleaf) in SIDE.immediate)):
            # weight and bias filed under different stages (tsrn.OUTCONV_*): two closures, each with its own stage's side lane
            (dw,) = SIDE.submit((wleaf,), lambda: (ops.conv_wgrad(x, dy, Cout, KH, KW, any_width=aw),), x, dy)
            (db,) = SIDE.submit((bleaf,), lambda: (ops.colsum(dy.reshape(-1, Cout)),), dy)
            return dx, dw, db, None, None
        dw, db = SIDE.submit(ctx.leaves, param_grads, x, dy)
        return dx, dw, db, None, None


def conv2d(x, weight, bias, act=ACT_NONE, any_width=False):
    """any_width: see ops.conv2d_forward (the CRNN's maps)"""
    return Conv2dFn.apply(x, weight, bias, act, any_width)


# --------------------------------------------------------------------------------------------------
class BatchNormActFn(Function):
    """nn.BatchNorm (train: batch statistics + running-stat update; eval: running stats) + activation."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training, momentum, eps, act):
        C = x.shape[-1]
        x2 = x.reshape(-1, C)
        if training:
            mean, rstd = ops.bn_stats(x2, eps, momentum, running_mean, running_var)
        else:
            mean, rstd = running_mean, ops.bn_rstd(running_var, eps)
        y = ops.bn_apply(x2, mean, rstd, gamma, beta, act).reshape(x.shape)
        ctx.save_for_backward(x, gamma, beta, mean, rstd)
        ctx.act, ctx.training = act, training
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, mean, rstd = ctx.saved_tensors
        C = x.shape[-1]
        dx, dg, db = ops.bn_bwd(x.reshape(-1, C), _c(dy).reshape(-1, C), mean, rstd, gamma, beta, ctx.act, ctx.training)
        return dx.reshape(x.shape), dg, db, None, None, None, None, None, None


def batch_norm_act(x, bn, act=ACT_NONE, count=True):
    """bn: an nn.BatchNorm{1,2}d used as parameter/buffer holder.  count=False: the caller advances `num_batches_tracked` itself
    (the generators bump the counters of all their BatchNorms with one launch, see tatt_amd.tsrn._GeneratorBase)."""
    training = bn.training
    if count and training and bn.track_running_stats:
        bn.num_batches_tracked += 1
    return BatchNormActFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, bn.momentum, bn.eps,
                                act)


# --------------------------------------------------------------------------------------------------
class ConvBnFn(Function):
    """3x3 convolution 64 -> 64 with BatchNorm (train mode) folded in on both sides (reference RecurrentResidualBlock:
    conv1 -> bn1 -> mish -> conv2 -> bn2, model/tsrn.py:877-886,896-903; block7, :609-614):

      * input side (`in_*` given): x is the PRE-BatchNorm output of the producing convolution; in_act(x * in_scale + in_shift) is
        applied while the kernel stages its halo, so the normalised + activated map is never written to HBM;
      * output side: the per-channel sum / sum of squares of y leave the convolution's epilogue as per-work-group partials and one
        small launch turns them into mean / rstd / running statistics and the folded (scale, shift) of THIS layer's BatchNorm.

    Returns y (raw convolution output) and mean, rstd, scale, shift of its BatchNorm (not differentiable: whoever applies the
    BatchNorm -- the next ConvBnFn through `in_*`, or BatchNormApplyFn -- back-propagates through the statistics).
    conv | stats1 | stats2 | apply | conv | ...  ->  conv(+stats) | finish | conv(+apply, +stats) | finish."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, running_mean, running_var, momentum, eps,
                in_mean, in_rstd, in_gamma, in_beta, in_scale, in_shift, in_act):
        B, H, W, C = x.shape
        y, part, G = ops.conv3_bn_forward(x, weight, bias, in_scale, in_shift, in_act, True)
        mean, rstd, scale, shift = ops.bn_stats_finish(part, G, 64, B * H * W, eps, momentum, gamma, beta, running_mean, running_var)
        folded = in_scale is not None
        ctx.save_for_backward(x, weight, in_mean, in_rstd, in_gamma, in_beta)
        ctx.folded, ctx.in_act, ctx.has_bias = folded, in_act, bias is not None
        ctx.leaves = (weight, bias)
        ctx.mark_non_differentiable(mean, rstd, scale, shift)
        ctx.set_materialize_grads(False)          # (otherwise autograd zero-fills a gradient for each of the four statistics: 4 launches)
        return y, mean, rstd, scale, shift

    @staticmethod
    def backward(ctx, dy, *_):
        x, weight, in_mean, in_rstd, in_gamma, in_beta = ctx.saved_tensors
        if dy is None:
            return (None,) * 16
        dy = _c(dy)
        x2 = x.reshape(-1, 64)
        dx = dgi = dbi = None
        need_in = ctx.folded and (ctx.needs_input_grad[0] or ctx.needs_input_grad[11] or ctx.needs_input_grad[12])
        if ctx.needs_input_grad[0] or need_in:
            dr = ops.conv2d_dgrad(dy, weight)
            if ctx.folded:
                dx, dgi, dbi = ops.bn_bwd(x2, dr.reshape(-1, 64), in_mean, in_rstd, in_gamma, in_beta, ctx.in_act, True)
                dx = dx.reshape(x.shape)
            else:
                dx = dr
        want_dw, want_db = ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]

        def param_grads():
            # the convolution's actual input (normalised + activated) is rebuilt here, off the critical path
            xin = x if not ctx.folded else ops.bn_apply(x2, in_mean, in_rstd, in_gamma, in_beta, ctx.in_act).reshape(x.shape)
            if want_dw and want_db:
                return ops.conv_wgrad(xin, dy, 64, 3, 3, want_db=True)
            dw = ops.conv_wgrad(xin, dy, 64, 3, 3) if want_dw else None
            db = ops.colsum(dy.reshape(-1, 64)) if want_db else None
            return dw, db
        # everything the deferred closure reads must outlive this node (autograd frees saved tensors when the node is done, and a block
        # freed on the main stream is reused there at once, beside the side lane that still reads it)
        dw, db = SIDE.submit(ctx.leaves, param_grads, x, dy, in_mean, in_rstd, in_gamma, in_beta, weight)
        return (dx, dw, db) + (None,) * 8 + (dgi, dbi, None, None, None)


class BatchNormApplyFn(Function):
    """y = act((x - mean) * rstd * gamma + beta) for statistics computed elsewhere (ConvBnFn); the backward is the full train-mode
    BatchNorm backward (the dependence of mean / rstd on x included)."""

    @staticmethod
    def forward(ctx, x, mean, rstd, gamma, beta, act):
        C = x.shape[-1]
        y = ops.bn_apply(x.reshape(-1, C), mean, rstd, gamma, beta, act).reshape(x.shape)
        ctx.save_for_backward(x, mean, rstd, gamma, beta)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd, gamma, beta = ctx.saved_tensors
        C = x.shape[-1]
        dx, dg, db = ops.bn_bwd(x.reshape(-1, C), _c(dy).reshape(-1, C), mean, rstd, gamma, beta, ctx.act, True)
        return dx.reshape(x.shape), None, None, dg, db, None


def conv_bn(x, conv, bn, prev=None):
    """conv: nn.Conv2d(64, 64, 3, padding=1), bn: its nn.BatchNorm2d (train mode), both parameter holders.
    prev = (y_prev's ConvBnFn statistics tuple, bn_prev, act): fold bn_prev + act of the producing layer into this convolution.
    -> (y, stats) with stats = (mean, rstd, scale, shift) of `bn` over y."""
    if prev is None:
        ins = (None,) * 6 + (ACT_NONE,)
    else:
        (pm, pr, ps, psh), pbn, pact = prev
        ins = (pm, pr, pbn.weight, pbn.bias, ps, psh, pact)
    out = ConvBnFn.apply(x, conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps, *ins)
    return out[0], tuple(out[1:])


def bn_apply_stats(y, stats, bn, act=ACT_NONE):
    return BatchNormApplyFn.apply(y, stats[0], stats[1], bn.weight, bn.bias, act)


SRB_BWD_FUSED = False       # True -> SrbTrunkFn (BatchNorm backward folded into the data-gradient convolutions).  Built, parity green, and SLOWER on the
                            # GPU: 5.49 vs 5.34 ms per step (profiles/r04_e_ab_srb_fused.txt) -- the split-bf16 convolution is bound by instruction
                            # issue, so the activation derivative + partial sums in its epilogue (~160 VALU per tile) and the second staged map cost
                            # as much there as the 12 us bandwidth-bound stage-1 / apply launches they replace.  Off; kept as the measured alternative


class SrbTrunkFn(Function):
    """conv1 -> bn1 -> mish -> conv2 -> bn2 of a RecurrentResidualBlock (reference model/tsrn.py:877-886, 896-903) as ONE operator, so
    that the BACKWARD can fold the two BatchNorm backwards into the data-gradient convolutions the way the forward folds the
    BatchNorms into the convolutions:

      forward   conv1(+stats) | finish | conv2(bn1 + mish on the way in, +stats) | finish | apply bn2             (as conv_bn / bn_apply_stats)
      backward  partials(bn2) | finish | dgrad conv2 (bn2 backward on the way in; mish' and bn1's partials in the epilogue)
                | finish | dgrad conv1 (bn1 backward on the way in)                                                 5 launches, was 8

    dx = gamma rstd (du - mean(du) - xhat mean(du xhat)) is affine per channel in (du, x) once the two means are known, so the consumer
    applies it while staging its halo (tatt_conv3_c64_dgrad_bn_sb).  The materialised dy maps are needed by the weight gradients
    only: they are rebuilt on the side lane (tatt_bn_bwd_affine) with the rest of the parameter-gradient work."""

    @staticmethod
    def forward(ctx, x, w1, b1, g1, be1, rm1, rv1, mom1, eps1, w2, b2, g2, be2, rm2, rv2, mom2, eps2):
        B, H, W, C = x.shape
        M = B * H * W
        y1, part1, G1 = ops.conv3_bn_forward(x, w1, b1, None, None, ACT_NONE, True)
        mean1, rstd1, sc1, sh1 = ops.bn_stats_finish(part1, G1, 64, M, eps1, mom1, g1, be1, rm1, rv1)
        y2, part2, G2 = ops.conv3_bn_forward(y1, w2, b2, sc1, sh1, ACT_MISH, True)
        mean2, rstd2, _, _ = ops.bn_stats_finish(part2, G2, 64, M, eps2, mom2, g2, be2, rm2, rv2)
        r = ops.bn_apply(y2.reshape(-1, 64), mean2, rstd2, g2, be2, ACT_NONE).reshape(x.shape)
        ctx.save_for_backward(x, y1, y2, w1, w2, g1, be1, g2, be2, mean1, rstd1, mean2, rstd2)
        ctx.has_b = (b1 is not None, b2 is not None)
        ctx.leaves = (w1, b1, w2, b2)
        return r

    @staticmethod
    def backward(ctx, d):
        x, y1, y2, w1, w2, g1, be1, g2, be2, mean1, rstd1, mean2, rstd2 = ctx.saved_tensors
        B, H, W, C = x.shape
        M = B * H * W
        d = _c(d)
        y1f, y2f = y1.reshape(M, 64), y2.reshape(M, 64)
        part2, G2 = ops.bn_bwd_partials(y2f, d.reshape(M, 64), mean2, rstd2, g2, be2, ACT_NONE)
        dg2, dbe2, coef2 = ops.bn_bwd_finish(part2, G2, 64, M, mean2, rstd2, g2)
        du1, part1, G1 = ops.conv3_dgrad_bn(d, w2, y2, coef2, ep=(y1, mean1, rstd1, g1, be1, ACT_MISH))
        dg1, dbe1, coef1 = ops.bn_bwd_finish(part1, G1, 64, M, mean1, rstd1, g1)
        dx = None
        if ctx.needs_input_grad[0]:
            dx, _, _ = ops.conv3_dgrad_bn(du1, w1, y1, coef1)
        want_b1, want_b2 = ctx.has_b

        def param_grads():
            # the weight gradients want the materialised maps: dy2, dy1 (BatchNorm backward applied) and conv2's actual input
            dy2 = ops.bn_bwd_affine(y2f, d.reshape(M, 64), coef2).reshape(x.shape)
            a1 = ops.bn_apply(y1f, mean1, rstd1, g1, be1, ACT_MISH).reshape(x.shape)
            r2 = ops.conv_wgrad(a1, dy2, 64, 3, 3, want_db=want_b2)
            dy1 = ops.bn_bwd_affine(y1f, du1.reshape(M, 64), coef1).reshape(x.shape)
            r1 = ops.conv_wgrad(x, dy1, 64, 3, 3, want_db=want_b1)
            dw2, db2 = r2 if want_b2 else (r2, None)
            dw1, db1 = r1 if want_b1 else (r1, None)
            return dw1, db1, dw2, db2
        dw1, db1, dw2, db2 = SIDE.submit(ctx.leaves, param_grads, x, y1, y2, d, du1, coef1, coef2, mean1, rstd1, g1, be1, w1, w2)
        return (dx, dw1, db1, dg1, dbe1, None, None, None, None, dw2, db2, dg2, dbe2, None, None, None, None)


def srb_trunk(x, blk):
    """blk: a RecurrentResidualBlock parameter holder (conv1, bn1, conv2, bn2); train mode, 64-channel NHWC map (conv3_bn_fusable)."""
    c1, n1, c2, n2 = blk.conv1, blk.bn1, blk.conv2, blk.bn2
    return SrbTrunkFn.apply(x, c1.weight, c1.bias, n1.weight, n1.bias, n1.running_mean, n1.running_var, n1.momentum, n1.eps,
                            c2.weight, c2.bias, n2.weight, n2.bias, n2.running_mean, n2.running_var, n2.momentum, n2.eps)


# --------------------------------------------------------------------------------------------------
class _PackedLinears:
    """Split-bf16 operands (tatt_tokgemm_pack) of the nn.Linear weights of the forward in flight, keyed by the weight's storage: a
    generator packs all of its token projections with one or two launches at the start of its forward (`linear_prepack`: forward
    operand W (N, K) and data-gradient operand W^T) instead of two tiny launches in front of every GEMM; LinearFn / QKVProjFn look
    their weight up and carry the data-gradient operand to the backward."""

    def __init__(self):
        self.table = {}


_PKL = _PackedLinears()
_TOKGEMM_NK = {(192, 128), (192, 64), (128, 192), (128, 128), (128, 64), (64, 192), (64, 64), (64, 128)}      # tatt_tokgemm_sb_ex


def linear_prepack(linears):
    """linears: nn.Linear parameter holders on one device whose (out, in) and (in, out) are shapes of tatt_tokgemm_sb_ex."""
    import ctypes
    _PKL.table = {}
    if not TOKGEMM_SB:
        return
    pk_ptrs, pk_dims, outs = [], [], {}
    for l in linears:
        W = l.weight
        N, K = W.shape
        if not W.is_cuda or (N, K) not in _TOKGEMM_NK or (K, N) not in _TOKGEMM_NK or W.data_ptr() in outs:
            continue
        Wfk, Wbk = ops.new(W, N * K), ops.new(W, N * K)
        pk_ptrs += [W.data_ptr(), Wfk.data_ptr(), W.data_ptr(), Wbk.data_ptr()]
        pk_dims += [N, K, K, 0, K, N, K, 1]
        outs[W.data_ptr()] = (Wfk, Wbk, W._version)
    if pk_ptrs:
        ops.call("tatt_tokgemm_pack_batch", (ctypes.c_void_p * len(pk_ptrs))(*pk_ptrs), (ctypes.c_int * len(pk_dims))(*pk_dims),
                 len(pk_ptrs) // 2, ops.stream())
    _PKL.table = outs


def linear_prepack_done():
    _PKL.table = {}


def _packed_linear(weight, M):
    """(forward operand, data-gradient operand) of a prepacked weight this forward may use on M tokens, else None"""
    e = _PKL.table.get(weight.data_ptr()) if TOKGEMM_SB else None
    if e is None or e[2] != weight._version or M % 64 or M < 64:
        return None
    return e[0], e[1]


def _tokgemm_ex(X, Wpk, bias, N, K, act=ACT_NONE, out=None, accum=False):
    ops._check_dev(X)
    M = X.shape[0]
    Y = ops.new(X, M, N) if out is None else out
    ops.call("tatt_tokgemm_sb_ex", ops.P(X), None, K, ops.P(Wpk), ops.P(bias), ops.P(Y), None, N, M, N, K, int(act), int(bool(accum)),
             ops.stream())
    return Y


def _linear_wgrad(dy2, x2, dw, db):
    """dw (N, K) = dy2^T x2, db = column sums of dy2 (or None): the split-bf16 token pass when it applies, else the fp32 GEMM"""
    if TOK_WGRAD_SB and ops.tok_wgrad_takes(dy2, x2) and dw.is_contiguous():
        ops.tok_wgrad_sb(dy2, x2, dw, db)
    else:
        ops.linear_bwd_weight(dy2, x2, out=dw, out_ld=dw.shape[1], rowsum=db)


TOK_WGRAD_SB = True         # tatt_amd.set_arithmetic: False -> fp32 GEMMs for the weight gradients of prepacked linears


class LinearFn(Function):
    """y = act(alpha*(x @ W^T + b)); optional second input concatenated along the feature axis.  A weight the forward in flight has
    prepacked (linear_prepack) runs on the bf16 matrix cores with split operands (csrc/tokgemm.hip), forward and data gradient."""

    @staticmethod
    def forward(ctx, x, xb, weight, bias, act, alpha):
        K1 = x.shape[-1]
        x2 = x.reshape(-1, K1)
        xb2 = xb.reshape(-1, xb.shape[-1]) if xb is not None else None
        pk = _packed_linear(weight, x2.shape[0]) if (xb is None and alpha == 1.0 and act in (ACT_NONE, ACT_RELU)) else None
        if pk is not None:
            y = _tokgemm_ex(x2, pk[0], bias, weight.shape[0], K1, act)
        else:
            y = ops.linear_fwd(x2, weight, bias, act=act, alpha=alpha, x2b=xb2)
        ctx.save_for_backward(x, xb, weight, y if act != ACT_NONE else None)
        ctx.act, ctx.alpha, ctx.has_bias = act, alpha, bias is not None
        ctx.leaves = (weight, bias)
        ctx.wbk = pk[1] if pk is not None else None
        return y.reshape(*x.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x, xb, weight, y = ctx.saved_tensors
        N, K = weight.shape
        K1 = x.shape[-1]
        dy2 = _c(dy).reshape(-1, N)
        if ctx.act != ACT_NONE:
            dy2 = ops.act_bwd(y, dy2, ctx.act, True)
        x2 = x.reshape(-1, K1)
        dx = dxb = dw = db = None
        if ctx.needs_input_grad[0]:
            if ctx.wbk is not None:
                dx = _tokgemm_ex(dy2, ctx.wbk, None, K, N).reshape(x.shape)
            else:
                dx = ops.linear_bwd_input(dy2, weight, col0=0, ncols=K1, alpha=ctx.alpha).reshape(x.shape)
        if xb is not None and ctx.needs_input_grad[1]:
            dxb = ops.linear_bwd_input(dy2, weight, col0=K1, ncols=K - K1, alpha=ctx.alpha).reshape(xb.shape)
        want_db = ctx.has_bias and ctx.needs_input_grad[3]
        want_dw = ctx.needs_input_grad[2]

        def param_grads():
            dw = db = None
            if want_dw:
                dw = ops.new(dy2, N, K)
                db = ops.new(dy2, N) if want_db else None             # bias gradient rides along with the weight-gradient GEMM
                if ctx.wbk is not None and xb is None:
                    _linear_wgrad(dy2, x2, dw, db)
                else:
                    ops.linear_bwd_weight(dy2, x2, alpha=ctx.alpha, out=dw, out_ld=K, rowsum=db)
                if xb is not None:
                    ops.linear_bwd_weight(dy2, xb.reshape(-1, K - K1), alpha=ctx.alpha, out=dw.reshape(-1)[K1:], out_ld=K)
            elif want_db:
                db = ops.colsum(dy2, scale=ctx.alpha)
            return dw, db
        dw, db = SIDE.submit(ctx.leaves, param_grads, dy2, x, xb)
        return dx, dxb, dw, db, None, None


class QKVProjFn(Function):
    """The three projections of one token matrix in front of an attention (reference MultiHeadedAttention.forward,
    model/tbsrn.py:119-128: linears[0..2] applied to the same x) as one operator: q, k, v = x Wq^T + bq, ...; backward
    dx = dq Wq + dk Wk + dv Wv accumulated by the GEMM epilogues (tatt_tokgemm_sb_ex accum) instead of three maps and two additions.
    Needs prepacked weights (linear_prepack)."""

    @staticmethod
    def forward(ctx, x, wq, bq, wk, bk, wv, bv):
        K = x.shape[-1]
        x2 = x.reshape(-1, K)
        pks = [_packed_linear(w, x2.shape[0]) for w in (wq, wk, wv)]
        N = wq.shape[0]
        outs = [_tokgemm_ex(x2, pk[0], b, N, K) for pk, b in zip(pks, (bq, bk, bv))]
        ctx.save_for_backward(x, wq, wk, wv)
        ctx.wbk = [pk[1] for pk in pks]
        ctx.has_b = [b is not None for b in (bq, bk, bv)]
        ctx.leaves = (wq, bq, wk, bk, wv, bv)
        return tuple(o.reshape(*x.shape[:-1], N) for o in outs)

    @staticmethod
    def backward(ctx, dq, dk, dv):
        x, wq, wk, wv = ctx.saved_tensors
        N, K = wq.shape
        x2 = x.reshape(-1, K)
        ds = [_c(d).reshape(-1, N) for d in (dq, dk, dv)]
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _tokgemm_ex(ds[0], ctx.wbk[0], None, K, N)
            _tokgemm_ex(ds[1], ctx.wbk[1], None, K, N, out=dx, accum=True)
            _tokgemm_ex(ds[2], ctx.wbk[2], None, K, N, out=dx, accum=True)
            dx = dx.reshape(x.shape)
        has_b = ctx.has_b

        def param_grads():
            res = []
            for d, hb in zip(ds, has_b):
                dw = ops.new(d, N, K)
                db = ops.new(d, N) if hb else None
                _linear_wgrad(d, x2, dw, db)
                res += [dw, db]
            return tuple(res)
        g = SIDE.submit(ctx.leaves, param_grads, x, *ds)
        return (dx,) + tuple(g)


class FeedForwardFn(Function):
    """w_2(Dropout_p(relu(w_1 x))) (reference PositionwiseFeedForward, model/tbsrn.py:154-164) as one operator on prepacked weights
    (linear_prepack): the dropout rides in the epilogue of the first GEMM, and the backward of dropout and relu in the epilogue of the
    second GEMM's data gradient (gated by the saved F = Dropout(relu(.)): F > 0 exactly where the unit was active and kept) -- two
    GEMMs forward, two backward, no element-wise pass.  Masks = tatt_dropout's for the same seed word, site and flat index."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, pdrop, site):
        K = x.shape[-1]
        x2 = x.reshape(-1, K)
        M = x2.shape[0]
        pk1, pk2 = _packed_linear(w1, M), _packed_linear(w2, M)
        Nf, No = w1.shape[0], w2.shape[0]
        seed = current_seed(x.device) if pdrop > 0.0 else None
        f = ops.new(x2, M, Nf)
        ops.call("tatt_tokgemm_sb_ffn", ops.P(x2), ops.P(pk1[0]), ops.P(b1), ops.P(f), M, Nf, K, ACT_RELU, float(pdrop), ops.P(seed), int(site),
                 None, 1.0, ops.stream())
        y = _tokgemm_ex(f, pk2[0], b2, No, Nf)
        ctx.save_for_backward(x, f, w1, w2)
        ctx.wbk = (pk1[1], pk2[1])
        ctx.pdrop = float(pdrop)
        ctx.has_b = (b1 is not None, b2 is not None)
        ctx.leaves = (w1, b1, w2, b2)
        return y.reshape(*x.shape[:-1], No)

    @staticmethod
    def backward(ctx, dy):
        x, f, w1, w2 = ctx.saved_tensors
        Nf, K = w1.shape
        No = w2.shape[0]
        x2 = x.reshape(-1, K)
        M = x2.shape[0]
        dy2 = _c(dy).reshape(-1, No)
        dpre = ops.new(dy2, M, Nf)                                   # gradient in front of the relu
        ops.call("tatt_tokgemm_sb_ffn", ops.P(dy2), ops.P(ctx.wbk[1]), None, ops.P(dpre), M, Nf, No, ACT_NONE, 0.0, None, 0, ops.P(f),
                 1.0 / (1.0 - ctx.pdrop), ops.stream())
        dx = _tokgemm_ex(dpre, ctx.wbk[0], None, K, Nf).reshape(x.shape) if ctx.needs_input_grad[0] else None
        hb1, hb2 = ctx.has_b

        def param_grads():
            dw2, db2 = ops.new(dy2, No, Nf), (ops.new(dy2, No) if hb2 else None)
            _linear_wgrad(dy2, f, dw2, db2)
            dw1, db1 = ops.new(dy2, Nf, K), (ops.new(dy2, Nf) if hb1 else None)
            _linear_wgrad(dpre, x2, dw1, db1)
            return dw1, db1, dw2, db2
        g = SIDE.submit(ctx.leaves, param_grads, x, f, dy2, dpre)
        return (dx,) + tuple(g) + (None, None)


class FeedForwardLnFn(Function):
    """LayerNorm(x + w_2(Dropout_p(relu(w_1 x)))) -- the feed-forward sub-layer of the TBSRN FeatureEnhancer with its residual LayerNorm
    (reference model/tbsrn.py:87-90) as ONE operator.  The forward is FeedForwardFn's two GEMMs and the LayerNorm launch; what the
    composition buys is the backward: x has two consumers (the feed-forward and the residual), and as separate operators autograd adds
    their two gradients with an element-wise launch (25 MB read twice, written once, per block).  Here the LayerNorm backward writes the
    residual's gradient and w_1's data-gradient GEMM ACCUMULATES into it (tatt_tokgemm_sb_ex accum).  Prepacked weights (linear_prepack)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, gamma, beta, eps, mode, pdrop, site):
        K = x.shape[-1]
        x2 = x.reshape(-1, K)
        M = x2.shape[0]
        pk1, pk2 = _packed_linear(w1, M), _packed_linear(w2, M)
        Nf, No = w1.shape[0], w2.shape[0]
        seed = current_seed(x.device) if pdrop > 0.0 else None
        f = ops.new(x2, M, Nf)
        ops.call("tatt_tokgemm_sb_ffn", ops.P(x2), ops.P(pk1[0]), ops.P(b1), ops.P(f), M, Nf, K, ACT_RELU, float(pdrop), ops.P(seed), int(site),
                 None, 1.0, ops.stream())
        y = _tokgemm_ex(f, pk2[0], b2, No, Nf)
        out, stats = ops.ln_fwd(x2, y, gamma, beta, eps, mode)
        ctx.save_for_backward(x, f, y, stats, w1, w2, gamma)
        ctx.wbk = (pk1[1], pk2[1])
        ctx.pdrop, ctx.eps, ctx.mode = float(pdrop), eps, mode
        ctx.has_b = (b1 is not None, b2 is not None)
        ctx.leaves = (w1, b1, w2, b2)
        return out.reshape(x.shape)

    @staticmethod
    def backward(ctx, dout):
        x, f, y, stats, w1, w2, gamma = ctx.saved_tensors
        Nf, K = w1.shape
        No = w2.shape[0]
        x2 = x.reshape(-1, K)
        M = x2.shape[0]
        # d(x + y): the residual's gradient and the feed-forward output's gradient are the same tensor
        dxy, _, dg, db = ops.ln_bwd(x2, y, _c(dout).reshape(-1, K), stats, gamma, ctx.eps, ctx.mode)
        dpre = ops.new(dxy, M, Nf)                                   # gradient in front of the relu
        ops.call("tatt_tokgemm_sb_ffn", ops.P(dxy), ops.P(ctx.wbk[1]), None, ops.P(dpre), M, Nf, No, ACT_NONE, 0.0, None, 0, ops.P(f),
                 1.0 / (1.0 - ctx.pdrop), ops.stream())
        hb1, hb2 = ctx.has_b

        def param_grads():
            dw2, db2 = ops.new(dyc, No, Nf), (ops.new(dyc, No) if hb2 else None)
            _linear_wgrad(dyc, f, dw2, db2)
            dw1, db1 = ops.new(dyc, Nf, K), (ops.new(dyc, Nf) if hb1 else None)
            _linear_wgrad(dpre, x2, dw1, db1)
            return dw1, db1, dw2, db2
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.new(dxy, M, K)                                  # = dxy (residual) + dpre W1; dxy stays what w_2's weight gradient reads
            ops.call("tatt_tokgemm_sb_add", ops.P(dpre), ops.P(ctx.wbk[0]), None, ops.P(dxy), ops.P(dx), M, K, Nf, ops.stream())
            dx = dx.reshape(x.shape)
        dyc = dxy
        g = SIDE.submit(ctx.leaves, param_grads, x, f, dyc, dpre)
        return (dx,) + tuple(g) + (dg, db, None, None, None, None)


def feed_forward(x, w_1, w_2, pdrop, training, site):
    """w_1, w_2: nn.Linear holders; dropout (site) between them in training"""
    x = _c(x)
    M = x.numel() // x.shape[-1]
    p = float(pdrop) if training else 0.0
    if _packed_linear(w_1.weight, M) is not None and _packed_linear(w_2.weight, M) is not None and FFN_FUSED:
        return FeedForwardFn.apply(x, w_1.weight, w_1.bias, w_2.weight, w_2.bias, p, site)
    f = linear(x, w_1.weight, w_1.bias, act=ACT_RELU)
    f = dropout(f, pdrop, training, site)
    return linear(f, w_2.weight, w_2.bias)


def feed_forward_ln(x, w_1, w_2, gamma, beta, eps, mode, pdrop, training, site):
    """LayerNorm_mode(x + w_2(Dropout(relu(w_1 x)))): one operator when the weights are prepacked, else the operator chain"""
    x = _c(x)
    M = x.numel() // x.shape[-1]
    p = float(pdrop) if training else 0.0
    if (FFN_LN_FUSED and FFN_FUSED and _packed_linear(w_1.weight, M) is not None and _packed_linear(w_2.weight, M) is not None
            and w_2.weight.shape[0] == x.shape[-1] and x.shape[-1] <= 128 and w_1.weight.shape[0] <= 128):
        return FeedForwardLnFn.apply(x, w_1.weight, w_1.bias, w_2.weight, w_2.bias, gamma, beta, eps, mode, p, site)
    f = feed_forward(x, w_1, w_2, pdrop, training, site)
    return LayerNormFn.apply(x, f, gamma, beta, eps, mode, 0.0, 0)


FFN_LN_FUSED = True         # test / A-B hook: False -> feed-forward and LayerNorm as two operators (autograd adds x's two gradients)
FFN_FUSED = True            # test / A-B hook: False -> linear + dropout + linear as separate operators


def qkv_projection(x, lq, lk, lv):
    """lq, lk, lv: nn.Linear holders applied to the same (B, P, K) token matrix"""
    x = _c(x)
    M = x.numel() // x.shape[-1]
    if all(_packed_linear(l.weight, M) is not None for l in (lq, lk, lv)):
        return QKVProjFn.apply(x, lq.weight, lq.bias, lk.weight, lk.bias, lv.weight, lv.bias)
    return tuple(linear(x, l.weight, l.bias) for l in (lq, lk, lv))


def linear(x, weight, bias=None, act=ACT_NONE, alpha=1.0, xb=None):
    return LinearFn.apply(_c(x), None if xb is None else _c(xb), weight, bias, act, alpha)


# --------------------------------------------------------------------------------------------------
class BiGRU32Fn(Function):
    """Bidirectional GRU (hidden 32) over the rows or columns of a (B,H,W,64) token grid."""

    @staticmethod
    def forward(ctx, x, wih_f, whh_f, bih_f, bhh_f, wih_r, whh_r, bih_r, bhh_r, vertical):
        B, H, W, C = x.shape
        x2 = x.reshape(-1, C)
        M = x2.shape[0]
        gi = ops.new(x, M, 192)
        ops.linear_fwd(x2, wih_f, bih_f, out=gi[:, :96])
        ops.linear_fwd(x2, wih_r, bih_r, out=gi[:, 96:])
        geom = ops.seq_geom(B, H, W, vertical)
        out, gates = ops.gru32_fwd(gi, whh_f, bhh_f, whh_r, bhh_r, geom, save=any(ctx.needs_input_grad))
        ctx.save_for_backward(x, gates, out, wih_f, whh_f, wih_r, whh_r)
        ctx.geom = geom
        return out.reshape(B, H, W, 64)

    @staticmethod
    def backward(ctx, dout):
        x, gates, out, wih_f, whh_f, wih_r, whh_r = ctx.saved_tensors
        C = x.shape[-1]
        x2 = x.reshape(-1, C)
        dgi, dgh, hprev = ops.gru32_bwd(gates, out, _c(dout).reshape(-1, 64), whh_f, whh_r, ctx.geom)
        dx = ops.linear_bwd_input(dgi[:, :96], wih_f)
        ops.linear_bwd_input(dgi[:, 96:], wih_r, out=dx, beta=1.0)
        g = []
        for d in range(2):
            gi_d, gh_d = dgi[:, 96 * d:96 * (d + 1)], dgh[:, 96 * d:96 * (d + 1)]
            g.append((ops.linear_bwd_weight(gi_d, x2), ops.linear_bwd_weight(gh_d, hprev[:, 32 * d:32 * (d + 1)]),
                      ops.colsum(gi_d), ops.colsum(gh_d)))
        (dwih_f, dwhh_f, dbih_f, dbhh_f), (dwih_r, dwhh_r, dbih_r, dbhh_r) = g
        return dx.reshape(x.shape), dwih_f, dwhh_f, dbih_f, dbhh_f, dwih_r, dwhh_r, dbih_r, dbhh_r, None


def bigru32(x, gru, vertical):
    """gru: an nn.GRU(64, 32, bidirectional=True) used as parameter holder."""
    return BiGRU32Fn.apply(_c(x), gru.weight_ih_l0, gru.weight_hh_l0, gru.bias_ih_l0, gru.bias_hh_l0,
                           gru.weight_ih_l0_reverse, gru.weight_hh_l0_reverse, gru.bias_ih_l0_reverse,
                           gru.bias_hh_l0_reverse, vertical)


# --------------------------------------------------------------------------------------------------
class _Precomposed:
    """Composed GruBlock projections (W_ih W_c, W_ih b_c + b_ih) of the forward in flight, keyed by the block's conv weight: a
    generator composes all of its blocks with ONE launch at the start of its forward (`gru_precompose`) instead of one tiny launch
    in front of every block's GEMM (10 dependent launches per TATT forward); GruBlockFn.forward pops its entry."""

    def __init__(self):
        self.table = {}


_PRE = _Precomposed()


def gru_precompose(blocks):
    """blocks: GruBlock parameter holders (conv1 = 1x1 conv, gru = nn.GRU(64, 32, bidirectional)) on one device."""
    import ctypes
    blocks = [b for b in blocks if b.conv1.weight.is_cuda]
    _PRE.table = {}
    if not blocks:
        return
    ptrs, Ks, outs = [], [], {}
    pk_ptrs, pk_dims = [], []
    for b in blocks:
        g, Wc = b.gru, b.conv1.weight
        K = Wc.numel() // Wc.shape[0]
        Wp, bp = ops.new(Wc, 192, K), ops.new(Wc, 192)
        ptrs += [t.data_ptr() for t in (g.weight_ih_l0, g.weight_ih_l0_reverse, g.bias_ih_l0, g.bias_ih_l0_reverse, Wc, b.conv1.bias, Wp, bp)]
        Ks.append(K)
        Wfk = Wbk = None
        if TOKGEMM_SB and K in (64, 128):            # split-bf16 operands of the projection (forward) and of its data gradient
            Wfk, Wbk = ops.new(Wc, 192 * K), ops.new(Wc, 192 * K)
            pk_ptrs += [Wp.data_ptr(), Wfk.data_ptr(), Wp.data_ptr(), Wbk.data_ptr()]
            pk_dims += [192, K, K, 0, K, 192, K, 1]
        outs[Wc.data_ptr()] = (Wp, bp, Wc._version, Wfk, Wbk)
    ops.call("tatt_gru_compose_batch", (ctypes.c_void_p * len(ptrs))(*ptrs), (ctypes.c_int * len(Ks))(*Ks), len(Ks), ops.stream())
    if pk_ptrs:
        ops.call("tatt_tokgemm_pack_batch", (ctypes.c_void_p * len(pk_ptrs))(*pk_ptrs), (ctypes.c_int * len(pk_dims))(*pk_dims),
                 len(pk_ptrs) // 2, ops.stream())
    _PRE.table = outs


def gru_precompose_done():
    """End of the generator's forward: drop whatever `gru_precompose` composed and no block popped (the table must not keep
    composed / packed weights of one forward alive until the next)."""
    _PRE.table = {}


TOKGEMM_SB = True           # test / A-B hook: False -> the exact-fp32 MFMA GEMMs for the GRU input projections
GRU_WGRAD_SB = True         # test / A-B hook: False -> three fp32-MFMA weight-gradient GEMMs per GruBlock instead of the fused pass
GRU_WGRAD_FRAG = True       # test / A-B hook: False -> tatt_gru_wgrad_sb from dgi / dgh / hprev (round 3) instead of the fragment stream


def _tokgemm(X1, X2, Wpk, bias, N, K, N1=None):
    """[X1 | X2] (M, K) @ W^T (+ bias) through tatt_tokgemm_sb -> (Y1 (M, N1), Y2 (M, N - N1) or None)"""
    M, K1 = X1.shape
    N1 = N if N1 is None else N1
    Y1 = ops.new(X1, M, N1)
    Y2 = ops.new(X1, M, N - N1) if N1 < N else None
    ops.call("tatt_tokgemm_sb", ops.P(X1), ops.P(X2), K1, ops.P(Wpk), ops.P(bias), ops.P(Y1), ops.P(Y2), N1, M, N, K, ops.stream())
    return Y1, Y2


class GruBlockFn(Function):
    """reference GruBlock (model/tsrn.py:1067-1084) as ONE operator: 1x1 conv (optionally over cat[x, xb]) -> BiGRU(64 -> 2x32)
    along image columns (vertical) or rows.

    The 1x1 conv has no non-linearity before the GRU's input projection, so the two linear maps are composed on the fly:
        gi = W_ih (W_c x + b_c) + b_ih = (W_ih W_c) x + (W_ih b_c + b_ih)
    (a 192 x K matrix product of a few kFLOP) and the token matrix is streamed ONCE per direction of the pass instead of three
    times: forward = 1 GEMM + recurrence; backward = recurrence + 1 GEMM for dx (+1 for the concatenated half) + 2 split-K
    GEMMs for the weight gradients; the gradients of W_ih, W_c, b_c follow from the composed ones by tiny products."""

    @staticmethod
    def forward(ctx, x, xb, conv_w, conv_b, wih_f, whh_f, bih_f, bhh_f, wih_r, whh_r, bih_r, bhh_r, vertical):
        B, H, W, K1 = x.shape
        Wc = conv_w.reshape(conv_w.shape[0], -1)                 # (64, K)
        K = Wc.shape[1]
        x2 = x.reshape(-1, K1)
        xb2 = xb.reshape(-1, K - K1) if xb is not None else None
        pre = _PRE.table.pop(conv_w.data_ptr(), None)           # composed at the start of this forward (gru_precompose)?
        Wfk = Wbk = None
        if pre is not None and pre[2] == conv_w._version and pre[0].device == x.device:
            Wp, bp, Wfk, Wbk = pre[0], pre[1], pre[3], pre[4]
        else:
            Wp = ops.new(x, 192, K)                              # composed projection  [W_ih_f; W_ih_r] @ W_c
            bp = ops.new(x, 192)
            ops.call("tatt_gru_compose", ops.P(wih_f), ops.P(wih_r), ops.P(bih_f), ops.P(bih_r), ops.P(Wc), ops.P(conv_b),
                     ops.P(Wp), ops.P(bp), K, ops.stream())
        use_tg = TOKGEMM_SB and x2.shape[0] % 64 == 0 and K in (64, 128) and K1 % 4 == 0
        if use_tg and Wfk is None:
            Wfk, Wbk = ops.new(x, 192 * K), ops.new(x, 192 * K)
            ops.call("tatt_tokgemm_pack", ops.P(Wp), ops.P(Wfk), 192, K, K, 0, ops.stream())
            ops.call("tatt_tokgemm_pack", ops.P(Wp), ops.P(Wbk), K, 192, K, 1, ops.stream())
        if use_tg:
            gi, _ = _tokgemm(x2, xb2, Wfk, bp, 192, K)
        else:
            gi = ops.linear_fwd(x2, Wp, bp, x2b=xb2)
            Wbk = None
        geom = ops.seq_geom(B, H, W, vertical)
        out, gates = ops.gru32_fwd(gi, whh_f, bhh_f, whh_r, bhh_r, geom, save=any(ctx.needs_input_grad))
        ctx.save_for_backward(x, xb, Wc, Wp, gates, out, wih_f, whh_f, wih_r, whh_r, conv_b, Wbk)
        ctx.geom = geom
        ctx.wshape = conv_w.shape
        ctx.leaves = (conv_w, conv_b, wih_f, whh_f, bih_f, bhh_f, wih_r, whh_r, bih_r, bhh_r)
        return out.reshape(B, H, W, 64)

    @staticmethod
    def backward(ctx, dout):
        x, xb, Wc, Wp, gates, out, wih_f, whh_f, wih_r, whh_r, conv_b, Wbk = ctx.saved_tensors
        K1 = x.shape[-1]
        K = Wc.shape[1]
        x2 = x.reshape(-1, K1)
        xb2 = xb.reshape(-1, K - K1) if xb is not None else None
        # round 4: the recurrence leaves the weight-gradient pass's operands in MFMA fragment order (no dgh / hprev round trip)
        use_frag = (GRU_WGRAD_FRAG and GRU_WGRAD_SB and K in (64, 128) and K1 == 64 and ops.gru_frag_ok(ctx.geom)
                    and x2.is_contiguous() and (xb2 is None or xb2.is_contiguous()))
        dgh = hprev = frag = None
        if use_frag:
            dgi, frag = ops.gru32_bwd_frag(gates, out, _c(dout).reshape(-1, 64), whh_f, whh_r, ctx.geom)
        else:
            dgi, dgh, hprev = ops.gru32_bwd(gates, out, _c(dout).reshape(-1, 64), whh_f, whh_r, ctx.geom)
        dxb = None
        if Wbk is not None and ctx.needs_input_grad[0] and (xb is None or ctx.needs_input_grad[1]):
            dx, dxb = _tokgemm(dgi, None, Wbk, None, K, 192, K1)      # dx | dxb = dgi Wp on the bf16 matrix cores (split operands)
            dx = dx.reshape(x.shape)
            dxb = dxb.reshape(xb.shape) if xb is not None else None
        elif xb is not None and ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and K == 2 * K1:
            dx, dxb = ops.linear_bwd_input_halves(dgi, Wp)          # both halves of the concatenated input: one launch
            dx, dxb = dx.reshape(x.shape), dxb.reshape(xb.shape)
        else:
            dx = ops.linear_bwd_input(dgi, Wp, col0=0, ncols=K1).reshape(x.shape) if ctx.needs_input_grad[0] else None
            if xb is not None and ctx.needs_input_grad[1]:
                dxb = ops.linear_bwd_input(dgi, Wp, col0=K1, ncols=K - K1).reshape(xb.shape)
        geom = ctx.geom

        def param_grads():
            # weight-gradient GEMMs over the tokens; the bias gradients (column sums of dgi / dgh) ride along as a virtual ones column
            dbp, dbhh = ops.new(dgi, 192), ops.new(dgi, 192)
            dWp = ops.new(dgi, 192, K)
            if use_frag:
                dWhh = ops.new(dgi, 192, 32)                          # compact: [forward; reverse]
                ops.gru_wgrad_frag(frag, x2, xb2, geom, dWp, dWhh, dbp, dbhh)
            elif GRU_WGRAD_SB and K in (64, 128) and K1 == 64 and ops.gru_wgrad_fusable(dgi, dgh, x2, xb2, hprev):
                dWhh = ops.new(dgi, 192, 64)                          # one pass over the tokens for all four results (split-bf16 MFMA)
                ops.gru_wgrad_sb(dgi, dgh, x2, xb2, hprev, dWp, dWhh, dbp, dbhh)
            else:
                ops.linear_bwd_weight(dgi, x2, out=dWp, out_ld=K, rowsum=dbp)
                if xb is not None:
                    ops.linear_bwd_weight(dgi, xb2, out=dWp.reshape(-1)[K1:], out_ld=K)
                dWhh = ops.linear_bwd_weight(dgh, hprev, rowsum=dbhh)     # (192, 64): diagonal blocks are the two directions
            yield                                                     # (batched split-K reduction of the three GEMMs above)
            dwih_f, dwih_r = ops.new(dgi, 96, 64), ops.new(dgi, 96, 64)
            dwhh_f, dwhh_r = ops.new(dgi, 96, 32), ops.new(dgi, 96, 32)
            dWc, dbc = ops.new(dgi, 64, K), ops.new(dgi, 64)
            ops.call("tatt_gru_tail_c" if use_frag else "tatt_gru_tail", ops.P(dWp), ops.P(dbp), ops.P(Wc), ops.P(conv_b),
                     ops.P(wih_f), ops.P(wih_r), ops.P(dwih_f), ops.P(dwih_r), ops.P(dWc), ops.P(dbc), K, ops.P(dWhh),
                     ops.P(dwhh_f), ops.P(dwhh_r), ops.stream())
            return (dWc.reshape(ctx.wshape), dbc, dwih_f, dwhh_f, dbp[:96], dbhh[:96], dwih_r, dwhh_r, dbp[96:], dbhh[96:])
        return (dx, dxb) + tuple(SIDE.submit(ctx.leaves, param_grads, dgi, dgh, hprev, frag, x, xb, Wp, Wc)) + (None,)


def gru_block(x, blk, vertical, xb=None):
    """blk: a GruBlock parameter holder (conv1 = 1x1 nn.Conv2d, gru = nn.GRU(64, 32, bidirectional))."""
    g = blk.gru
    return GruBlockFn.apply(_c(x), None if xb is None else _c(xb), blk.conv1.weight, blk.conv1.bias,
                            g.weight_ih_l0, g.weight_hh_l0, g.bias_ih_l0, g.bias_hh_l0,
                            g.weight_ih_l0_reverse, g.weight_hh_l0_reverse, g.bias_ih_l0_reverse, g.bias_hh_l0_reverse,
                            vertical)


# --------------------------------------------------------------------------------------------------
class AddFn(Function):
    @staticmethod
    def forward(ctx, a, b):
        return ops.axpby(a, b, 1.0, 1.0)

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


def add(a, b):
    return AddFn.apply(_c(a), _c(b))


class Fork2Fn(Function):
    """x -> (x, x) for a tensor with two consumers: the two gradients are summed by ONE launch of the library (ops.add_n) instead of the
    autograd engine's own accumulation kernel (a + b either way: bit-identical)."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, ga, gb):
        if ga is None or gb is None:
            return ga if gb is None else gb
        if ga.is_cuda and ga.dtype == torch.float32 and ga.shape == gb.shape:
            return ops.add_n([_c(ga), _c(gb)])
        return ga + gb


def fork2(x):
    return Fork2Fn.apply(x) if (x.requires_grad and torch.is_grad_enabled()) else (x, x)


class AddRowBcastFn(Function):
    """a (B,L,C) + b (L,C) broadcast over B; b carries no gradient (positional-encoding buffer)."""

    @staticmethod
    def forward(ctx, a, b):
        return ops.add_rowbcast(a, b, b.shape[0])

    @staticmethod
    def backward(ctx, dy):
        return dy, None


class ScaleFn(Function):
    @staticmethod
    def forward(ctx, a, s):
        ctx.s = s
        return ops.axpby(a, None, s, 0.0)

    @staticmethod
    def backward(ctx, dy):
        return ops.axpby(_c(dy), None, ctx.s, 0.0), None


class MeanOf2Fn(Function):
    @staticmethod
    def forward(ctx, a, b):
        return ops.axpby(a, b, 0.5, 0.5)

    @staticmethod
    def backward(ctx, dy):
        h = ops.axpby(_c(dy), None, 0.5, 0.0)
        return h, h


class ActFn(Function):
    """y = act(x) for activations whose derivative is a function of the output (relu, tanh)."""

    @staticmethod
    def forward(ctx, x, act):
        y = ops.act_fwd(x, act)
        ctx.save_for_backward(y)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return ops.act_bwd(y, _c(dy), ctx.act, True), None


class PReLUFn(Function):
    @staticmethod
    def forward(ctx, x, alpha):
        ctx.save_for_backward(x, alpha)
        return ops.prelu_fwd(x, alpha)

    @staticmethod
    def backward(ctx, dy):
        x, alpha = ctx.saved_tensors
        dx, da = ops.prelu_bwd(x, _c(dy), alpha)
        return dx, da.reshape(alpha.shape)


def prelu(x, alpha):
    return PReLUFn.apply(_c(x), alpha)


class PixelShuffleActFn(Function):
    @staticmethod
    def forward(ctx, x, act):
        ctx.save_for_backward(x)
        ctx.act = act
        return ops.pixel_shuffle_fwd(x, act)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.pixel_shuffle_bwd(x, _c(dy), ctx.act), None


class MaxPoolFn(Function):
    """nn.MaxPool2d((kh,kw), stride (sh,sw), padding (ph,pw)) on an NHWC map; stride defaults to the kernel."""

    @staticmethod
    def forward(ctx, x, kh, kw, sh, sw, ph, pw):
        ctx.save_for_backward(x)
        ctx.k = (kh, kw, sh, sw, ph, pw)
        return ops.maxpool_fwd(x, *ctx.k)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return (ops.maxpool_bwd(x, _c(dy), *ctx.k),) + (None,) * 6


def max_pool(x, kh, kw, sh=None, sw=None, ph=0, pw=0):
    return MaxPoolFn.apply(x, kh, kw, sh or kh, sw or kw, ph, pw)


class Permute4dFn(Function):
    """Materialised permutation of a 4-D tensor (layout change), backward = inverse permutation."""

    @staticmethod
    def forward(ctx, x, perm):
        ctx.perm = perm
        return ops.to_contiguous(x.permute(*perm))

    @staticmethod
    def backward(ctx, dy):
        inv = [0] * 4
        for i, p in enumerate(ctx.perm):
            inv[p] = i
        return ops.to_contiguous(_c(dy).permute(*inv)), None


class LayerNormFn(Function):
    """LayerNorm(a + b) over the last axis (b optional); with p > 0: LayerNorm(a + Dropout_p(b)), the dropout (site `site` of this
    forward's seed) applied while b is read -- same mask as `dropout(b, p, True, site)`."""

    @staticmethod
    def forward(ctx, a, b, gamma, beta, eps, mode, p, site):
        C = a.shape[-1]
        ctx.p, ctx.site = float(p), site
        ctx.seed = current_seed(a.device) if p > 0.0 else None
        y, stats = ops.ln_fwd(a.reshape(-1, C), None if b is None else b.reshape(-1, C), gamma, beta, eps, mode, ctx.p, ctx.seed, site)
        ctx.save_for_backward(a, b, stats, gamma)
        ctx.eps, ctx.mode = eps, mode
        return y.reshape(a.shape)

    @staticmethod
    def backward(ctx, dy):
        a, b, stats, gamma = ctx.saved_tensors
        C = a.shape[-1]
        dx, db2, dg, db = ops.ln_bwd(a.reshape(-1, C), None if b is None else b.reshape(-1, C), _c(dy).reshape(-1, C), stats,
                                     gamma, ctx.eps, ctx.mode, ctx.p, ctx.seed, ctx.site)
        return dx.reshape(a.shape), (db2.reshape(a.shape) if b is not None else None), dg, db, None, None, None, None


def layer_norm(a, b, ln, p=0.0, training=False, site=0):
    """LayerNorm(a + b); with training and p > 0: LayerNorm(a + Dropout_p(b)) in one kernel."""
    p = p if (training and b is not None) else 0.0
    return LayerNormFn.apply(_c(a), None if b is None else _c(b), ln.weight, ln.bias, ln.eps, 0, p, site)


class DropoutFn(Function):
    @staticmethod
    def forward(ctx, x, p, site):
        ctx.p, ctx.site = p, site
        ctx.seed = current_seed(x.device)
        return ops.dropout(x, p, ctx.seed, site)

    @staticmethod
    def backward(ctx, dy):
        return ops.dropout(_c(dy), ctx.p, ctx.seed, ctx.site), None, None


def dropout(x, p, training, site):
    if not training or p <= 0.0:
        return x
    return DropoutFn.apply(_c(x), p, site)


class AttnCoreFn(Function):
    @staticmethod
    def forward(ctx, Q, K, V, pdrop, site):
        seed = current_seed(Q.device)
        c, w = ops.attn_fwd(Q, K, V, pdrop, seed, site, True)
        ctx.save_for_backward(Q, K, V)
        ctx.pdrop, ctx.site, ctx.seed = pdrop, site, seed
        ctx.set_materialize_grads(False)
        return c, w

    @staticmethod
    def backward(ctx, dc, dw):
        Q, K, V = ctx.saved_tensors
        if dc is None:
            dc = torch.zeros_like(Q)
        dQ, dK, dV = ops.attn_bwd(Q, K, V, _c(dc), None if dw is None else _c(dw), ctx.pdrop, ctx.seed, ctx.site)
        return dQ, dK, dV, None, None


class MhaInProjFn(Function):
    """The packed input projection of nn.MultiheadAttention: Q = (q_in W_q^T + b_q) / sqrt(d), K = k_in W_k^T + b_k,
    V = v_in W_v^T + b_v with in_proj_weight = [W_q; W_k; W_v] (3E, E).  One operator so that the gradient of the PACKED
    parameter is written in place by the three weight-gradient GEMMs (slicing the parameter in autograd costs a zero-fill,
    a block copy and an accumulate per slice and per step)."""

    @staticmethod
    def forward(ctx, q_in, k_in, v_in, w, b, qscale):
        E = w.shape[1]
        xs = (q_in, k_in, v_in)
        outs = []
        for i, x in enumerate(xs):
            y = ops.linear_fwd(x.reshape(-1, E), w[i * E:(i + 1) * E], b[i * E:(i + 1) * E], alpha=qscale if i == 0 else 1.0)
            outs.append(y.reshape(*x.shape[:-1], E))
        ctx.save_for_backward(q_in, k_in, v_in, w)
        ctx.qscale = qscale
        ctx.leaves = (w, b)
        return tuple(outs)

    @staticmethod
    def backward(ctx, dQ, dK, dV):
        q_in, k_in, v_in, w = ctx.saved_tensors
        E = w.shape[1]
        dxs, dys = [], []
        for i, (dy, x) in enumerate(((dQ, q_in), (dK, k_in), (dV, v_in))):
            a = ctx.qscale if i == 0 else 1.0
            dy2 = _c(dy).reshape(-1, E)
            dys.append(dy2)
            dxs.append(ops.linear_bwd_input(dy2, w[i * E:(i + 1) * E], alpha=a).reshape(x.shape) if ctx.needs_input_grad[i] else None)

        def param_grads():
            dw, db = ops.new(w, 3 * E, E), ops.new(w, 3 * E)
            for i, (dy2, x) in enumerate(zip(dys, (q_in, k_in, v_in))):
                a = ctx.qscale if i == 0 else 1.0
                ops.linear_bwd_weight(dy2, x.reshape(-1, E), alpha=a, out=dw[i * E:(i + 1) * E], out_ld=E, rowsum=db[i * E:(i + 1) * E])
            return dw, db
        dw, db = SIDE.submit(ctx.leaves, param_grads, dys, q_in, k_in, v_in)
        return dxs[0], dxs[1], dxs[2], dw, db, None


def multihead_attention(q_in, k_in, v_in, mha, training, site):
    """nn.MultiheadAttention forward (batch-major tensors (B,L,E)); mha = parameter holder.
    Returns (output (B,L,E), head-averaged attention weights (B,L,S))."""
    E = mha.embed_dim
    d = E // mha.num_heads
    Q, K, V = MhaInProjFn.apply(_c(q_in), _c(k_in), _c(v_in), mha.in_proj_weight, mha.in_proj_bias, 1.0 / math.sqrt(d))
    pdrop = float(mha.dropout) if training else 0.0
    ctx_, wts = AttnCoreFn.apply(Q, K, V, pdrop, site)
    out = linear(ctx_, mha.out_proj.weight, mha.out_proj.bias)
    return out, wts


# --------------------------------------------------------------------------------------------------
class TPStackCfg:
    """Static description of a fused stack (see TPStackFn): dropout sites per layer, probabilities, final norm, outputs."""

    def __init__(self, sites, p_attn, p_res, p_ffn, fin, need_wavg, shared_mem, eps):
        self.sites, self.p_attn, self.p_res, self.p_ffn = tuple(sites), float(p_attn), float(p_res), float(p_ffn)
        self.fin, self.need_wavg, self.shared_mem, self.eps = bool(fin), bool(need_wavg), bool(shared_mem), float(eps)
        self.n = len(self.sites)
        assert 1 <= self.n <= 2 and not (self.shared_mem and self.n != 1)


class TPStackFn(Function):
    """One or two transformer layers of the TP interpreter over one memory, each layer ONE kernel (csrc/tplayer.hip):

        layer:  x <- LN_B(x1 + Drop(FFN(x1))),  x1 = LN_A(x + Drop(MHA(q = x + qpos, k = mem + pos, v = mem)))
        out  =  mean over the layers of LN_F(layer output)   (cfg.fin: reference TransformerDecoder.forward,
                model/transformer_v2.py:380-390, averaged by TPInterpreter, model/tsrn.py:218)   or the last layer's output

    = reference TransformerDecoderLayer_TP.forward_post (:806-833) for the decoder, TransformerEncoderLayer.forward_post (:470-484)
    for the encoder (x is mem: cfg.shared_mem).  The K / V rows of each layer's packed in-projection are applied here (two small
    GEMMs per layer over the S <= 32 memory tokens).  Nothing but the layer inputs is saved: the backward kernel recomputes.
    params: per layer (in_proj_weight, in_proj_bias, out_proj.weight, out_proj.bias, linear1.weight, linear1.bias, linear2.weight,
    linear2.bias, norm_A.weight, norm_A.bias, norm_B.weight, norm_B.bias), then (norm_F.weight, norm_F.bias) with cfg.fin."""

    @staticmethod
    def forward(ctx, x, qpos, mem, pos, cfg, *params):
        n = cfg.n
        lps = [params[12 * l:12 * l + 12] for l in range(n)]
        lnF = tuple(params[12 * n:12 * n + 2]) if cfg.fin else None
        B, L, E = x.shape
        S = mem.shape[1]
        drop = cfg.p_attn > 0.0 or cfg.p_res > 0.0 or cfg.p_ffn > 0.0
        seed = current_seed(x.device) if drop else None
        xs, Ks, Vs, hms, packs = [x], [], [], [], []
        out = wavg = None
        # training (a backward will follow): the second generation of the layer (csrc/tplayer2.hip) where it takes the geometry -- its
        # forward leaves the FFN's relu bits and the packed operands for its backward
        gen2 = any(ctx.needs_input_grad) and ops.TPLAYER_BWD2 and ops.tplayer2_geom(B, L, S)[0] == 1
        if gen2:
            # ONE launch: mem + pos, every layer's key / value projection, the operand packing (K and V never reach memory)
            kin, pks = ops.tplayer2_kvprep(_c(mem), _c(pos), lps)
        else:
            kin = ops.axpby(mem, pos, 1.0, 1.0) if pos.dim() == 3 else ops.add_rowbcast(mem, pos, pos.shape[0])
        kin2, mem2 = kin.reshape(-1, E), mem.reshape(-1, E)
        for l, lp in enumerate(lps):
            in_w, in_b = lp[0], lp[1]
            K = V = None
            if not gen2:
                K = ops.linear_fwd(kin2, in_w[E:2 * E], in_b[E:2 * E]).reshape(B, S, E)
                V = ops.linear_fwd(mem2, in_w[2 * E:], in_b[2 * E:]).reshape(B, S, E)
            last = l == n - 1
            fin_here = cfg.fin and last
            if gen2:
                pk = pks[l]
                xout, fin, w, hm = ops.tplayer2_fwd(xs[l], qpos, pk, lp, lnF if fin_here else None, 1.0 / n, int(n == 2), cfg.p_attn,
                                                    cfg.p_res, cfg.p_ffn, seed, cfg.sites[l], cfg.eps, not fin_here,
                                                    cfg.need_wavg and last, S)
            else:
                pk = hm = None
                xout, fin, w = ops.tplayer_fwd(xs[l], qpos, K, V, lp, lnF if fin_here else None, 1.0 / n, int(n == 2), cfg.p_attn,
                                               cfg.p_res, cfg.p_ffn, seed, cfg.sites[l], cfg.eps, not fin_here,
                                               cfg.need_wavg and last)
            hms.append(hm)
            packs.append(pk)
            if last:
                out, wavg = (fin if fin_here else xout), w
            else:
                xs.append(xout)
            Ks.append(K)
            Vs.append(V)
        ctx.save_for_backward(qpos, mem, kin, *xs, *Ks, *Vs, *params)
        ctx.cfg, ctx.seed, ctx.hms, ctx.packs = cfg, seed, hms, packs
        ctx.set_materialize_grads(False)          # an unused `wavg` must not cost a zero-filled (B,L,S) gradient
        return out, wavg

    @staticmethod
    def backward(ctx, dout, dwavg):
        cfg = ctx.cfg
        n = cfg.n
        sv = ctx.saved_tensors
        qpos, mem, kin = sv[0], sv[1], sv[2]
        xs, Ks, Vs = sv[3:3 + n], sv[3 + n:3 + 2 * n], sv[3 + 2 * n:3 + 3 * n]
        params = sv[3 + 3 * n:]
        lps = [params[12 * l:12 * l + 12] for l in range(n)]
        lnF = tuple(params[12 * n:12 * n + 2]) if cfg.fin else None
        B, L, E = xs[0].shape
        S = mem.shape[1]
        kin2, mem2 = kin.reshape(-1, E), mem.reshape(-1, E)
        up = _c(dout) if dout is not None else torch.zeros_like(xs[0])
        dwavg = _c(dwavg) if dwavg is not None else None
        want_dq = ctx.needs_input_grad[1]
        want_dmem = ctx.needs_input_grad[2] or (cfg.shared_mem and ctx.needs_input_grad[0])
        dmem2 = dq = None
        pgrads = [None] * len(params)
        for l in range(n - 1, -1, -1):
            lp = lps[l]
            last = l == n - 1
            fin_here = cfg.fin and last
            # second generation (split-bf16, csrc/tplayer2.hip) where it takes the geometry, else the exact-fp32 first generation
            gen2 = ctx.packs[l] is not None
            dxo, dfi, dwa = None if fin_here else up, up if fin_here else None, dwavg if last else None
            G2 = 0
            if gen2:
                dx, dq, kvpart, kvflags, ppart, G2 = ops.tplayer2_bwd(xs[l], qpos, ctx.packs[l], lp, lnF if fin_here else None, 1.0 / n,
                                                                      int(n == 2), cfg.p_attn, cfg.p_res, cfg.p_ffn, ctx.seed, cfg.sites[l],
                                                                      cfg.eps, dxo, dfi, dwa, dq, want_dq, S, hmask=ctx.hms[l])
                dK, dV = ops.tplayer2_reduce_kv(kvpart, kvflags, B, L, S)
            else:
                dx, dq, kvpart, ppart = ops.tplayer_bwd(xs[l], qpos, Ks[l], Vs[l], lp, lnF if fin_here else None, 1.0 / n, int(n == 2),
                                                        cfg.p_attn, cfg.p_res, cfg.p_ffn, ctx.seed, cfg.sites[l], cfg.eps, dxo, dfi, dwa,
                                                        dq, want_dq)
                dK, dV = ops.tplayer_reduce_kv(kvpart, B, L, S)
            dK2, dV2 = dK.reshape(-1, E), dV.reshape(-1, E)
            in_w = lp[0]
            if want_dmem:
                if dmem2 is None:
                    if cfg.shared_mem:                       # x is the memory itself (encoder): its gradients add up in dx
                        dmem2 = dx.reshape(-1, E)
                        ops.linear_bwd_input(dK2, in_w[E:2 * E], out=dmem2, beta=1.0)
                    else:
                        dmem2 = ops.linear_bwd_input(dK2, in_w[E:2 * E])
                else:
                    ops.linear_bwd_input(dK2, in_w[E:2 * E], out=dmem2, beta=1.0)
                ops.linear_bwd_input(dV2, in_w[2 * E:], out=dmem2, beta=1.0)

            def param_grads(lp=lp, ppart=ppart, dK2=dK2, dV2=dV2, fin_here=fin_here, G2=G2):
                g = [ops.new(dK2, *t.shape) for t in lp]
                gF = [ops.new(dK2, *t.shape) for t in lnF] if fin_here else [None, None]
                dsts = [g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7], g[8], g[9], g[10], g[11], gF[0], gF[1]]
                if G2:
                    ops.tplayer_reduce_params_g(ppart, G2, dsts)
                else:
                    ops.tplayer_reduce_params(ppart, B, L, dsts)
                ops.linear_bwd_weight(dK2, kin2, out=g[0][E:2 * E], out_ld=E, rowsum=g[1][E:2 * E])
                ops.linear_bwd_weight(dV2, mem2, out=g[0][2 * E:], out_ld=E, rowsum=g[1][2 * E:])
                return tuple(g) + (tuple(gF) if fin_here else ())
            leaves = tuple(lp) + (tuple(lnF) if fin_here else ())
            got = SIDE.submit(leaves, param_grads, ppart, dK2, dV2, kin2, mem2)
            pgrads[12 * l:12 * l + 12] = got[:12]
            if fin_here:
                pgrads[12 * n:12 * n + 2] = got[12:14]
            up = dx
        dmem = None
        if want_dmem and not cfg.shared_mem:
            dmem = dmem2.reshape(mem.shape)
        return (up, dq if want_dq else None, dmem, None, None) + tuple(pgrads)


# --------------------------------------------------------------------------------------------------
# Query GRU recurrences as persistent launches (tatt_qgru_fwd_chain / tatt_qgru_bwd_chain) instead of one launch per time step.  Every spin in them is bounded
# by the wall clock; qgru_chain_check() reads the error words of the most recent launches (synchronises: call it outside a capture).
QGRU_CHAIN_FWD = True
QGRU_CHAIN_BWD = True
# their recurrent products on the bf16 matrix cores with split operands (hi hi + hi lo + lo hi, fp32 accumulation) instead of fp32 MFMA:
# the fp32 form takes a third of the chip's fp32 matrix throughput while the chain runs and slows the lane beside it
QGRU_CHAIN_SB = True
QGRU_WGRAD_SB = True            # the recurrent weight gradient (dgh^T h_prev over all steps) in split bf16 too (tatt_qgru_wgrad_sb)
QGRU_CHAIN_SYNC = []


# ---- residency and the sticky error word of the launches that synchronise their work-groups in flight --------------------------------
_CAPACITY = {}           # device index -> (query-GRU chain capacities x4, STN map launches, STN fully connected launches)
_STICKY = {}             # device index -> the device's sticky error word (int32 x 4; word 0 is registered with the library)


def _has_gpu():
    return torch.cuda.is_available()


def sync_capacity(device=None):
    """Work-groups of each in-flight-synchronising kernel that fit on the device at once (tatt_qgru_chain_capacity, tatt_stn_capacity:
    occupancy per CU x the CUs this process sees), asked once per device.  Those launches are only correct with their whole grid
    resident: on a partitioned GPU (fewer CUs reported) the callers below take the per-step / operator-chain paths instead.  A CU mask
    (HSA_CU_MASK) or CUs held by co-running kernels are NOT seen here: the bounded waits + sticky error word cover those."""
    idx = None if device is None else torch.device(device).index
    if idx is None:
        idx = torch.cuda.current_device()
    if idx not in _CAPACITY:
        import ctypes
        q, s = (ctypes.c_int * 4)(), (ctypes.c_int * 2)()
        with torch.cuda.device(idx):
            ops.call("tatt_qgru_chain_capacity", q)
            ops.call("tatt_stn_capacity", s)
        _CAPACITY[idx] = (q[0], q[1], q[2], q[3], s[0], s[1])
    return _CAPACITY[idx]


def sticky_word(device):
    """The device's sticky error word: allocated (zero) and registered with the library on first use -- outside any capture, so that no
    replayed fill ever resets it (the Trainer asks for it when it is built).  A wait that expires in any query-GRU chain or STN-head
    launch ORs a code into it; nothing but a new process clears it."""
    idx = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    if idx not in _STICKY:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("tatt_amd: the sticky error word must be allocated outside a graph capture (build the Trainer first)")
        w = torch.zeros(4, dtype=torch.int32, device=torch.device("cuda", idx))
        with torch.cuda.device(idx):
            ops.call("tatt_set_sticky", ops.P(w))
        _STICKY[idx] = w
    return _STICKY[idx]


def _qgru_chain_takes(W, HID, bwd=False, device=None):
    """Geometry of the persistent chains AND their whole grid resident on `device` (default: the current one).  The capacity is occupancy
    x the CUs the device reports: it follows partition modes (CPX / a smaller part) but NOT a CU mask (HSA_CU_MASK) nor CUs held by
    co-running kernels -- there the bounded spin + sticky error word (sync_check) is the protection, not this gate."""
    if not (HID == 512 and W % 16 == 0 and (W // 16) * 2 * (HID // 16) <= 256):
        return False
    if not _has_gpu():                           # (host-only callers: the geometry predicate alone)
        return True
    cap = sync_capacity(device)
    return (W // 16) * 2 * (HID // 16) <= cap[(2 if bwd else 0) + (0 if QGRU_CHAIN_SB else 1)]


def _qgru_chain_sync(ref):
    if not torch.cuda.is_current_stream_capturing():
        sticky_word(ref.device)                  # (registered before the first launch that could raise it)
    sync = torch.empty(1024, device=ref.device, dtype=torch.int32)
    QGRU_CHAIN_SYNC.append(sync)
    del QGRU_CHAIN_SYNC[:-8]
    return sync


def qgru_chain_check():
    for s in QGRU_CHAIN_SYNC:
        if int(s[1023].item()) != 0:
            raise RuntimeError("tatt_amd: a persistent query-GRU launch gave up waiting for its neighbours (work-groups not co-resident?)")


class QueryGruFn(Function):
    """Query positional embedding: init_factor (H*W, C) -> BiGRU(C*H -> C*H/2 per direction) whose TIME axis is
    the sample axis (reference quirk) -> (B, H, W, C)."""

    @staticmethod
    def forward(ctx, emb, wih0, whh0, bih0, bhh0, wih1, whh1, bih1, bhh1, B, H, W):
        # parameters only: may run as a parallel branch of the forward pass -- the CALLER joins (FWD_FORK.join) before the first use
        return FWD_FORK.run(emb, lambda: QueryGruFn._forward(ctx, emb, wih0, whh0, bih0, bhh0, wih1, whh1, bih1, bhh1, B, H, W))

    @staticmethod
    def _forward(ctx, emb, wih0, whh0, bih0, bhh0, wih1, whh1, bih1, bhh1, B, H, W):
        ctx.leaves = (emb, wih0, whh0, bih0, bhh0, wih1, whh1, bih1, bhh1)
        C = emb.shape[1]
        HID = whh0.shape[1]
        IN = wih0.shape[1]
        assert IN == H * C and 2 * HID == H * C
        dev = emb
        # x[w, h*C + c] = emb[h*W + w, c]
        stamp("qgru fwd: branch start", emb)
        x = ops.new(dev, W, IN)
        ops.copy4d(emb, x, (1, W, H, C), (0, C, W * C, 1), (0, IN, C, 1))
        # h of both directions with ONE zero time slot: in front of the forward direction's sequence, behind the reverse one's -- so
        # that "h_prev of every step" is one contiguous (B*W, HID) matrix per direction (the W_hh gradient GEMM then covers all rows
        # and its row sums are the hidden-bias gradient: no separate column-sum pass)
        hbuf = ops.new(dev, 2, B + 1, W, HID)
        ops.zero_f32(hbuf[0, 0])
        ops.zero_f32(hbuf[1, B])
        hseq = (hbuf[0, 1:], hbuf[1, :B])
        gsave = ops.new(dev, 2, B, 4, W, HID)
        q = ops.new(dev, B, H, W, C)
        Hh = H // 2
        chain = QGRU_CHAIN_FWD and _qgru_chain_takes(W, HID, device=dev.device) and IN % 1024 == 0
        if chain:
            # ONE persistent launch: the input projection of each tile, the B time steps of both directions, h written in both layouts
            xch = ops.new(dev, 2, B + 1, W, HID) if QGRU_CHAIN_SB else None      # h in matrix-core operand form (split-bf16 recurrence)
            ops.call("tatt_qgru_fwd_chain", None, None, ops.P(whh0), ops.P(whh1), ops.P(bhh0), ops.P(bhh1), ops.P(hbuf[0]),
                     ops.P(hbuf[1]), ops.P(gsave[0]), ops.P(gsave[1]), ops.P(_qgru_chain_sync(dev)), B, W, HID, 0, B, ops.P(x),
                     ops.P(wih0), ops.P(wih1), ops.P(bih0), ops.P(bih1), IN, ops.P(q), C,
                     ops.P(xch[0]) if xch is not None else None, ops.P(xch[1]) if xch is not None else None, ops.stream())
        else:
            gi = [ops.linear_fwd(x, wih0, bih0), ops.linear_fwd(x, wih1, bih1)]       # (W, 3*HID) each
            for s in range(B):
                t0, t1 = s, B - 1 - s
                hp0 = hseq[0][t0 - 1] if s > 0 else None
                hp1 = hseq[1][t1 + 1] if s > 0 else None
                ops.call("tatt_qgru_fwd_step", ops.P(gi[0]), ops.P(gi[1]), ops.P(whh0), ops.P(whh1), ops.P(bhh0),
                         ops.P(bhh1), ops.P(hp0), ops.P(hp1), ops.P(hseq[0][t0]), ops.P(hseq[1][t1]),
                         ops.P(gsave[0, t0]), ops.P(gsave[1, t1]), W, HID, ops.stream())
            # q[n, h, w, c] = hseq[d][n][w][(h % (H/2))*C + c], d = h // (H/2)
            for d in range(2):
                ops.copy4d(hseq[d], q[:, d * Hh:], (B, W, Hh, C), (W * HID, HID, C, 1), (H * W * C, C, W * C, 1))
        # W_hh^T of both directions for the per-step backward kernels (their B operand wants the 3*HID axis contiguous): parameters
        # only, so the two transposes ride on this forked branch.  The persistent backward launch reads W_hh as it is stored.
        whhT = None
        if any(ctx.needs_input_grad[:9]) and not (QGRU_CHAIN_BWD and _qgru_chain_takes(W, HID, True, device=dev.device) and B > 1):
            whhT = ops.new(dev, 2, HID, 3 * HID)
            for d, whh in enumerate((whh0, whh1)):                 # (3*HID, HID) -> (HID, 3*HID)
                ops.copy4d(whh, whhT[d], (1, 1, HID, 3 * HID), (0, 0, 1, HID), (0, 0, 3 * HID, 1))
        stamp("qgru fwd: branch end", emb)
        ctx.save_for_backward(emb, x, wih0, whh0, wih1, whh1, hbuf, gsave, whhT)
        ctx.dims = (B, H, W, C, HID, IN)
        return q

    @staticmethod
    def backward(ctx, dq):
        # every output of this backward is a parameter gradient (the embedding depends on parameters only): all of it is deferrable
        dq = _c(dq)
        saved = ctx.saved_tensors
        return tuple(SIDE.submit(ctx.leaves, lambda: QueryGruFn._backward(ctx, saved, dq), dq, saved)) + (None, None, None)

    @staticmethod
    def _backward(ctx, saved, dq):
        emb, x, wih0, whh0, wih1, whh1, hbuf, gsave, whhT = saved
        B, H, W, C, HID, IN = ctx.dims
        hseq = (hbuf[0, 1:], hbuf[1, :B])
        dev = emb
        Hh = H // 2
        stamp("qgru bwd: start", emb)
        dhseq = ops.new(dev, 2, B, W, HID)
        for d in range(2):
            ops.copy4d(dq[:, d * Hh:], dhseq[d], (B, W, Hh, C), (H * W * C, C, W * C, 1), (W * HID, HID, C, 1))
        dgh = ops.new(dev, 2, B, W, 3 * HID)
        dgi_acc = ops.new(dev, 2, W, 3 * HID)
        dhc = ops.new(dev, 2, W, HID)

        def prev_h(t0, t1):
            return (hseq[0][t0 - 1] if t0 > 0 else None), (hseq[1][t1 + 1] if t1 < B - 1 else None)

        # backward sweep: step s visits time B-1-s in the forward direction and time s in the reverse direction.  (Issuing the first
        # steps one pass early, as a closure of their own beside block1's backward, was measured: no gain -- profiles/r04_m_ab.txt.)
        hp0, hp1 = prev_h(B - 1, 0)
        ops.call("tatt_qgru_bwd_gates", ops.P(dhseq[0, B - 1]), ops.P(dhseq[1, 0]), ops.P(gsave[0, B - 1]),
                 ops.P(gsave[1, 0]), ops.P(hp0), ops.P(hp1), ops.P(dhc[0]), ops.P(dhc[1]), ops.P(dgi_acc[0]),
                 ops.P(dgi_acc[1]), ops.P(dgh[0, B - 1]), ops.P(dgh[1, 0]), W, HID, 1, ops.stream())
        chain = (QGRU_CHAIN_BWD or whhT is None) and _qgru_chain_takes(W, HID, True, device=dev.device) and B > 1
        if whhT is None and not chain and B > 1:
            raise RuntimeError("tatt_amd: QGRU_CHAIN_BWD was switched off between a forward and its backward")
        if chain:
            # the remaining B-1 steps as ONE persistent launch (work-groups exchange dgh through write-through stores and flag words)
            sync = _qgru_chain_sync(dev)
            wt = (whhT[0], whhT[1], 1) if whhT is not None else (whh0, whh1, 0)
            xch = ops.new(dev, 2, B, W, 3 * HID) if QGRU_CHAIN_SB else None      # dgh in matrix-core operand form
            ops.call("tatt_qgru_bwd_chain", ops.P(dgh[0]), ops.P(dgh[1]), ops.P(wt[0]), ops.P(wt[1]), ops.P(dhseq[0]),
                     ops.P(dhseq[1]), ops.P(gsave[0]), ops.P(gsave[1]), ops.P(hbuf[0]), ops.P(hbuf[1]), ops.P(dhc[0]),
                     ops.P(dhc[1]), ops.P(dgi_acc[0]), ops.P(dgi_acc[1]), ops.P(sync), B, W, HID, 0, B - 1, wt[2],
                     ops.P(xch[0]) if xch is not None else None, ops.P(xch[1]) if xch is not None else None, ops.stream())
        for s in range(0 if chain else B - 1):
            c0, c1 = B - 1 - s, s                # current step's time indices
            n0, n1 = c0 - 1, c1 + 1              # next step's
            hp0, hp1 = prev_h(n0, n1)
            ops.call("tatt_qgru_bwd_fused", ops.P(dgh[0, c0]), ops.P(dgh[1, c1]), ops.P(whhT[0]), ops.P(whhT[1]),
                     ops.P(dhseq[0, n0]), ops.P(dhseq[1, n1]), ops.P(gsave[0, n0]), ops.P(gsave[1, n1]), ops.P(hp0),
                     ops.P(hp1), ops.P(dhc[0]), ops.P(dhc[1]), ops.P(dgi_acc[0]), ops.P(dgi_acc[1]),
                     ops.P(dgh[0, n0]), ops.P(dgh[1, n1]), W, HID, ops.stream())
        stamp("qgru bwd: recurrence done", emb)
        grads = []
        dx = ops.new(dev, W, IN)
        # dW_hh = sum_t dgh_t^T h_{t-1} over ALL steps (h_prev of the first step is the zero slot); db_hh = its row sums
        g2 = [dgh[d].reshape(B * W, 3 * HID) for d in range(2)]
        hprev_all = [(hbuf[0, :B] if d == 0 else hbuf[1, 1:]).reshape(B * W, HID) for d in range(2)]
        hh = None
        if QGRU_WGRAD_SB and ops.qgru_wgrad_takes(g2[0], hprev_all[0]):
            # both directions in ONE split-bf16 launch (2 x 4.8 GFLOP: 70 us each on the fp32 pipe, the longest kernels of this lane)
            hh = ops.qgru_wgrad_sb(g2[0], g2[1], hprev_all[0], hprev_all[1])
        for d, (wih, whh) in enumerate(((wih0, whh0), (wih1, whh1))):
            if hh is not None:
                dwhh, dbhh = hh[2 * d], hh[2 * d + 1]
            else:
                dbhh = ops.new(dev, 3 * HID)
                dwhh = ops.linear_bwd_weight(g2[d], hprev_all[d], rowsum=dbhh)
            dbih = ops.new(dev, 3 * HID)
            dwih = ops.linear_bwd_weight(dgi_acc[d], x, rowsum=dbih)     # the bias gradient rides along (row sums of dgi_acc^T)
            ops.linear_bwd_input(dgi_acc[d], wih, out=dx, beta=0.0 if d == 0 else 1.0)
            grads.append((dwih, dwhh, dbih, dbhh))
        # dx is consumed right here, but its GEMMs split the contraction and, inside SIDE.flush, their reductions are only registered:
        # the accumulating second one (beta = 1) already forces everything pending through -- say so explicitly (a no-op then)
        ops.reduce_flush()
        demb = torch.empty_like(emb)
        ops.copy4d(dx, demb, (1, W, H, C), (0, IN, C, 1), (0, C, W * C, 1))
        stamp("qgru bwd: end", emb)
        (a0, b0, c0, d0), (a1, b1, c1, d1) = grads
        return demb, a0, b0, c0, d0, a1, b1, c1, d1


def query_embedding(emb, gru, B, H, W):
    return QueryGruFn.apply(emb, gru.weight_ih_l0, gru.weight_hh_l0, gru.bias_ih_l0, gru.bias_hh_l0,
                            gru.weight_ih_l0_reverse, gru.weight_hh_l0_reverse, gru.bias_ih_l0_reverse,
                            gru.bias_hh_l0_reverse, B, H, W)


# --------------------------------------------------------------------------------------------------
# Timeline inside a replayed hipGraph (tools/step_stamps.py): with STAMPS a list, stamp(name) launches a one-thread kernel that writes
# the 100 MHz wall clock when the current stream reaches that point; the stamps are graph nodes like any other launch and are
# re-written by every replay.  rocprofv3 serialises graph nodes; this shows the overlap as it happens.
# --------------------------------------------------------------------------------------------------
STAMPS = None


def stamp(name, ref=None):
    if STAMPS is None:
        return
    dev = ref.device if ref is not None else torch.device("cuda", torch.cuda.current_device())
    t = torch.zeros(1, device=dev, dtype=torch.int64)
    STAMPS.append((name, t))
    ops.call("tatt_stamp", ops.P(t), ops.stream())


# --------------------------------------------------------------------------------------------------
# STN head as a few launches (csrc/stnhead.hip): one Function for the whole head.  Convolutions (forward, data and weight gradients) go
# through the ordinary entry points; everything between them is one launch per layer and direction, and the fully connected end is one
# launch each way.  53 -> 17 launches forward, 62 -> 23 backward, at both exposed ends of the training step.
# --------------------------------------------------------------------------------------------------
STN_FOLD_SPLITS = True   # test / A-B hook: False -> every split convolution of the head is followed by its own tatt_splitk_reduce launch
STN_FOLD_MAX = 9         # ... only up to this many partial maps: the consumer's few work-groups add them one dependent load after the other
STN_SYNC = []            # every sync buffer handed to those launches (sync_check reads their error words)


def _stn_sync(holder, site, ref):
    """the 256-word sync buffer of one call site: zeroed once, reused by every later launch of that site"""
    table = holder.__dict__.setdefault("_tatt_sync", {})
    key = (site, ref.device)
    if key not in table:
        if not torch.cuda.is_current_stream_capturing():
            sticky_word(ref.device)
        table[key] = torch.zeros(256, device=ref.device, dtype=torch.int32)
    buf = table[key]
    if not any(r() is buf for r in STN_SYNC):        # registered on LOOKUP: a deep-copied / unpickled module brings its buffers unregistered
        STN_SYNC[:] = [r for r in STN_SYNC if r() is not None]      # (the buffers belong to their module and go with it)
        STN_SYNC.append(weakref.ref(buf))
    return buf


def sync_check():
    """Raise if a launch that synchronises its work-groups in flight gave up waiting (every such spin is bounded by the wall clock).
    Synchronises the device: call it outside a capture."""
    for idx, w in _STICKY.items():               # the sticky word first: it also remembers launches whose own words are gone
        code = int(w[0].item())
        if code:
            raise RuntimeError("tatt_amd: a launch that synchronises its work-groups in flight gave up waiting on cuda:%d (%s): its "
                               "results -- and everything computed from them since -- are invalid (work-groups not co-resident?)" % (
                                   idx, " + ".join(n for b, n in ((1, "query-GRU chain"), (2, "STN head")) if code & b)))
    qgru_chain_check()
    for r in STN_SYNC:
        s = r()
        if s is not None and int(s[255].item()) != 0:
            raise RuntimeError("tatt_amd: an STN-head launch gave up waiting for its neighbours (work-groups not co-resident?)")


def stn_head_fusable(x, stn, B):
    bns = [stn.stn_convnet[i][1] for i in (0, 2, 4, 6, 8, 10)] + [stn.stn_fc1[1]]
    if not all(bn.training and bn.momentum is not None and bn.affine and bn.track_running_stats for bn in bns):
        return False
    H, W = x.shape[1], x.shape[2]
    if not (B <= 64 and H == 16 and W % 32 == 0 and (W // 32) * 256 == stn.stn_fc1[0].in_features == 512
            and stn.stn_fc2.out_features % 4 == 0 and stn.stn_fc2.out_features <= 64):
        return False
    if not (_has_gpu() and x.is_cuda):           # (host-only callers: the geometry predicate alone)
        return True
    cap = sync_capacity(x.device)                # the launches need up to 128 / 32 work-groups resident at once
    return cap[4] >= 128 and cap[5] >= 32


class StnHeadFn(Function):
    """STNHead.forward in training mode (model/stn_head.py:92-106) on the NHWC-indexed view x (B,H,W,Cin): control points (B, NO).
    args: x, then (conv.weight, conv.bias, bn.weight, bn.bias) x 6, fc1.weight, fc1.bias, bn1.weight, bn1.bias, fc2.weight,
    fc2.bias, then the STNHead module (running statistics, sync buffers)."""
    POOLS = ((2, 2), (2, 2), (2, 2), (2, 2), (1, 2), (1, 1))

    @staticmethod
    def forward(ctx, x, *args):
        stn = args[-1]
        pr = args[:-1]
        B = x.shape[0]
        saved, geoms = [], []
        a = x
        for L in range(6):
            w, b, ga, be = pr[4 * L:4 * L + 4]
            bn = stn.stn_convnet[2 * L][1]
            C = w.shape[0]
            fold = STN_FOLD_SPLITS and a.is_contiguous() and 1 < ops.conv_split(a, C, 3, 3) <= STN_FOLD_MAX
            if fold:
                # the deep layers' convolutions split their contraction over the CUs: the partial maps are summed (+ bias) by the
                # BatchNorm launch as it loads them -- one launch less per link of this dependent chain
                xparts, S = ops.conv_partials(a, ops.repack_weight(w, 0), C, 3, 3)
                xc = ops.new(a, B, a.shape[1], a.shape[2], C)
            else:
                xc = ops.conv2d_forward(a, w, b, ACT_NONE)
            _, H, W, C = xc.shape
            ph, pw = StnHeadFn.POOLS[L]
            A = ops.new(xc, B, H // ph, W // pw, C)
            mean, rstd = ops.new(xc, C), ops.new(xc, C)
            part = ops.new(xc, 128 * 3 * C, dtype=torch.float64)
            if fold:
                ops.call("tatt_stn_bn_pool_fwd_parts", ops.P(xparts), S, ops.P(b), ops.P(xc), ops.P(A), ops.P(ga), ops.P(be),
                         ops.P(mean), ops.P(rstd), ops.P(bn.running_mean), ops.P(bn.running_var), ops.P(part),
                         ops.P(_stn_sync(stn, "f%d" % L, xc)), B, H, W, C, ph, pw, float(bn.eps), float(bn.momentum), ops.stream())
            else:
                ops.call("tatt_stn_bn_pool_fwd", ops.P(xc), ops.P(A), ops.P(ga), ops.P(be), ops.P(mean), ops.P(rstd),
                         ops.P(bn.running_mean), ops.P(bn.running_var), ops.P(part), ops.P(_stn_sync(stn, "f%d" % L, xc)), B, H, W,
                         C, ph, pw, float(bn.eps), float(bn.momentum), ops.stream())
            saved += [a, xc, mean, rstd]
            geoms.append((H, W, C, ph, pw))
            a = A
        w1, b1, g1, be1, w2, b2 = pr[24:30]
        bn1 = stn.stn_fc1[1]
        NO = w2.shape[0]
        U, S = ops.new(a, B, 512), ops.new(a, B, 512)
        mean1, rstd1 = ops.new(a, 512), ops.new(a, 512)
        ctrl = ops.new(a, B, NO)
        part = ops.new(a, 32 * B * NO)
        ops.call("tatt_stn_fc_fwd", ops.P(a), ops.P(w1), ops.P(b1), ops.P(g1), ops.P(be1), ops.P(bn1.running_mean),
                 ops.P(bn1.running_var), ops.P(w2), ops.P(b2), ops.P(U), ops.P(mean1), ops.P(rstd1), ops.P(S), ops.P(ctrl), ops.P(part),
                 ops.P(_stn_sync(stn, "ffc", a)), B, NO, float(bn1.eps), float(bn1.momentum), ops.stream())
        ctx.save_for_backward(*saved, a, U, S, mean1, rstd1, *pr)
        ctx.geoms, ctx.stn, ctx.NO = geoms, stn, NO
        return ctrl

    @staticmethod
    def backward(ctx, dctrl):
        t = ctx.saved_tensors
        saved, (a6, U, S, mean1, rstd1), pr = t[:24], t[24:29], t[29:]
        stn, NO = ctx.stn, ctx.NO
        B = a6.shape[0]
        dctrl = _c(dctrl)
        stamp("stn bwd: start", dctrl)
        w1, b1, g1, be1, w2, b2 = pr[24:30]
        dW2, db2 = torch.empty_like(w2), torch.empty_like(b2)
        dg1, dbe1, db1, dW1 = torch.empty_like(g1), torch.empty_like(be1), torch.empty_like(b1), torch.empty_like(w1)
        dU, dA = ops.new(a6, B, 512), torch.empty_like(a6)
        ops.call("tatt_stn_fc_bwd", ops.P(dctrl), ops.P(w2), ops.P(S), ops.P(U), ops.P(mean1), ops.P(rstd1), ops.P(g1), ops.P(w1),
                 ops.P(a6), ops.P(dW2), ops.P(db2), ops.P(dg1), ops.P(dbe1), ops.P(dW1), ops.P(db1), ops.P(dU), ops.P(dA),
                 ops.P(_stn_sync(stn, "bfc", a6)), B, NO, ops.stream())
        grads = [None] * 24
        stamp("stn bwd: fc done", dctrl)
        S = 1                                        # dA: S partial maps (a data-gradient convolution's split contraction, unsummed)
        for L in range(5, -1, -1):
            a_in, xc, mean, rstd = saved[4 * L:4 * L + 4]
            w, b, ga, be = pr[4 * L:4 * L + 4]
            H, W, C, ph, pw = ctx.geoms[L]
            dX = torch.empty_like(xc)
            dga, dbe, dbias = torch.empty_like(ga), torch.empty_like(be), torch.empty_like(b)
            part = ops.new(xc, 128 * 3 * C, dtype=torch.float64)
            ops.call("tatt_stn_bn_pool_bwd_parts", ops.P(xc), ops.P(dA), S, ops.P(ga), ops.P(be), ops.P(mean), ops.P(rstd), ops.P(dX),
                     ops.P(dga), ops.P(dbe), ops.P(dbias), ops.P(part), ops.P(_stn_sync(stn, "b%d" % L, xc)), B, H, W, C, ph, pw,
                     ops.stream())
            stamp("stn bwd: layer %d BatchNorm done" % (L + 1), dctrl)
            if L > 0:
                if STN_FOLD_SPLITS and 1 < ops.conv_split(dX, w.shape[1], 3, 3) <= STN_FOLD_MAX:
                    dA, S = ops.conv_partials(dX, ops.repack_weight(w, 1), w.shape[1], 3, 3)     # summed by the next layer's launch
                else:
                    dA, S = ops.conv2d_dgrad(dX, w), 1
                stamp("stn bwd: layer %d data gradient done" % (L + 1), dctrl)
            Cout = w.shape[0]
            (dw,) = SIDE.submit((w,), (lambda a_in=a_in, dX=dX, Cout=Cout: (ops.conv_wgrad(a_in, dX, Cout, 3, 3),)), a_in, dX)
            grads[4 * L:4 * L + 4] = [dw, dbias, dga, dbe]
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.conv2d_dgrad(dX, pr[0])
        stamp("stn bwd: activation chain done", dctrl)
        return (dx,) + tuple(grads) + (dW1, db1, dg1, dbe1, dW2, db2, None)


def stn_head(x, stn):
    ps = []
    for L in range(6):
        conv, bn = stn.stn_convnet[2 * L][0], stn.stn_convnet[2 * L][1]
        ps += [conv.weight, conv.bias, bn.weight, bn.bias]
    fc1, bn1 = stn.stn_fc1[0], stn.stn_fc1[1]
    ps += [fc1.weight, fc1.bias, bn1.weight, bn1.bias, stn.stn_fc2.weight, stn.stn_fc2.bias]
    return StnHeadFn.apply(x, *ps, stn)


# --------------------------------------------------------------------------------------------------
class TpsGridFn(Function):
    @staticmethod
    def forward(ctx, ctrl, inv, pad, repr_):
        # torch.inverse hands the kernel inverse back with column-major strides; the kernels index it row-major.  (The matrix is
        # symmetric up to round-off, so reading it transposed "works" -- with 7e-6 of error in the sampling grid.)
        inv, pad, repr_ = _c(inv), _c(pad), _c(repr_)
        ctx.save_for_backward(inv, repr_)
        ctx.N = ctrl.shape[1]
        return ops.tps_grid_fwd(ctrl, inv, pad, repr_)

    @staticmethod
    def backward(ctx, dsrc):
        inv, repr_ = ctx.saved_tensors
        return ops.tps_grid_bwd(_c(dsrc), inv, repr_, ctx.N), None, None, None


class GridSampleFn(Function):
    """x (B,C,H,W) any strides (no gradient), src (B,H*W,2) in [0,1] coordinates -> (B,H,W,C)."""

    @staticmethod
    def forward(ctx, x, src):
        # the kernels take the OUTPUT size from the input image (the reference's TPS output size equals its input size)
        assert src.dim() == 3 and src.shape[0] == x.shape[0] and src.shape[1] == x.shape[2] * x.shape[3] and src.shape[2] == 2, \
            "sampling grid (B, H*W, 2) does not match the image (B, C, H, W): %s vs %s" % (tuple(src.shape), tuple(x.shape))
        ctx.save_for_backward(x, src)
        return ops.grid_sample_fwd(x, src)

    @staticmethod
    def backward(ctx, dout):
        x, src = ctx.saved_tensors
        if ctx.needs_input_grad[0]:
            raise RuntimeError("tatt_amd: gradient w.r.t. the LR image through the TPS sampler is not on the path")
        return None, ops.grid_sample_bwd(x, src, _c(dout))


# --------------------------------------------------------------------------------------------------
# TBSRN variant (reference model/tbsrn.py): full self-attention over the H*W positions
# --------------------------------------------------------------------------------------------------
class SelfAttnCoreFn(Function):
    """softmax(Q K^T / sqrt(d_k)) -> dropout -> @ V for h heads of d_k = E/h (reference `attention`, model/tbsrn.py:130-151).
    Q, K, V: (B, P, E) already projected.  The (B, h, P, P) probabilities are materialised (288 GB of HBM: B=48, P=1024 is
    0.8 GB per block) and every product is a batched fp32-MFMA GEMM (tatt_gemm); the row softmax (+dropout) is tatt_softmax_rows_*."""

    @staticmethod
    def forward(ctx, Q, K, V, h, pdrop, site):
        B, Pn, E = Q.shape
        d = E // h
        scale = 1.0 / math.sqrt(d)
        seed = current_seed(Q.device)
        S = ops.new(Q, B, h, Pn, Pn)
        for hh in range(h):
            ops.gemm(Q.reshape(-1)[hh * d:], E, 1, K.reshape(-1)[hh * d:], 1, E, S.reshape(-1)[hh * Pn * Pn:], Pn, 1, Pn, Pn, d,
                     Z=B, bsA=Pn * E, bsB=Pn * E, bsC=h * Pn * Pn, alpha=scale)
        Pd = ops.new(Q, B, h, Pn, Pn) if pdrop > 0.0 else None
        ops.call("tatt_softmax_rows_fwd", ops.P(S), ops.P(Pd), B * h * Pn, Pn, pdrop, ops.P(seed), site, ops.stream())
        Pm = Pd if Pd is not None else S
        O = torch.empty_like(Q)
        for hh in range(h):
            ops.gemm(Pm.reshape(-1)[hh * Pn * Pn:], Pn, 1, V.reshape(-1)[hh * d:], E, 1, O.reshape(-1)[hh * d:], E, 1, Pn, d, Pn,
                     Z=B, bsA=h * Pn * Pn, bsB=Pn * E, bsC=Pn * E)
        # S now holds the (un-dropped) probabilities.  The dropped copy is kept too (0.8 GB per block at B = 48, P = 1024 -- HBM is
        # 288 GB) rather than regenerated in the backward: one full read+write pass less per block.
        ctx.save_for_backward(Q, K, V, S, Pd)
        ctx.cfg = (h, pdrop, site, seed, scale)
        return O

    @staticmethod
    def backward(ctx, dO):
        Q, K, V, Pp, Pd = ctx.saved_tensors
        h, pdrop, site, seed, scale = ctx.cfg
        B, Pn, E = Q.shape
        d = E // h
        dO = _c(dO)
        if Pd is None:
            Pd = Pp
        dQ, dK, dV = torch.empty_like(Q), torch.empty_like(K), torch.empty_like(V)
        dP = ops.new(Q, B, h, Pn, Pn)
        for hh in range(h):
            o, so = hh * d, hh * Pn * Pn
            # dV_h = Pd^T dO_h
            ops.gemm(Pd.reshape(-1)[so:], 1, Pn, dO.reshape(-1)[o:], E, 1, dV.reshape(-1)[o:], E, 1, Pn, d, Pn,
                     Z=B, bsA=h * Pn * Pn, bsB=Pn * E, bsC=Pn * E)
            # dPd = dO_h V_h^T
            ops.gemm(dO.reshape(-1)[o:], E, 1, V.reshape(-1)[o:], 1, E, dP.reshape(-1)[so:], Pn, 1, Pn, Pn, d,
                     Z=B, bsA=Pn * E, bsB=Pn * E, bsC=h * Pn * Pn)
        ops.call("tatt_softmax_rows_bwd", ops.P(Pp), ops.P(dP), B * h * Pn, Pn, pdrop, ops.P(seed), site, ops.stream())
        for hh in range(h):
            o, so = hh * d, hh * Pn * Pn
            # dQ_h = scale * dS K_h ;  dK_h = scale * dS^T Q_h
            ops.gemm(dP.reshape(-1)[so:], Pn, 1, K.reshape(-1)[o:], E, 1, dQ.reshape(-1)[o:], E, 1, Pn, d, Pn,
                     Z=B, bsA=h * Pn * Pn, bsB=Pn * E, bsC=Pn * E, alpha=scale)
            ops.gemm(dP.reshape(-1)[so:], 1, Pn, Q.reshape(-1)[o:], E, 1, dK.reshape(-1)[o:], E, 1, Pn, d, Pn,
                     Z=B, bsA=h * Pn * Pn, bsB=Pn * E, bsC=Pn * E, alpha=scale)
        return dQ, dK, dV, None, None, None


def _sattn_select():
    want = 2 if SATTN_SB else 1
    if ops.LIB.tatt_sattn_generation(0) != want:
        ops.LIB.tatt_sattn_generation(want)


class SelfAttnFlashFn(Function):
    """The same attention as SelfAttnCoreFn without the (B, h, P, P) tensors: csrc/sattn2.hip (split bf16; csrc/sattn.hip = the exact
    fp32 kernels) -- online softmax forward; backward recomputes the probabilities tile by tile from Q, K and the saved per-query
    log-sum-exp.  With dropout the forward leaves the keep decisions behind as bits (B h P^2 / 8 bytes) so that the two backward kernels
    do not recompute the counter hash (SATTN_KEEP_BITS).  d_k = 32, P a multiple of 64."""

    @staticmethod
    def forward(ctx, Q, K, V, h, pdrop, site):
        ops._check_dev(Q)
        B, Pn, E = Q.shape
        scale = 1.0 / math.sqrt(E // h)
        seed = current_seed(Q.device) if pdrop > 0.0 else None
        O, lse = torch.empty_like(Q), ops.new(Q, B, h, Pn)
        _sattn_select()
        bits = torch.empty(B * h * Pn * (Pn // 32), device=Q.device, dtype=torch.int32) if (pdrop > 0.0 and SATTN_KEEP_BITS and Pn % 32 == 0) else None
        ops.call("tatt_sattn_fwd_bits", ops.P(Q), ops.P(K), ops.P(V), ops.P(O), ops.P(lse), ops.P(bits), B, Pn, h, scale, float(pdrop),
                 ops.P(seed), int(site), ops.stream())
        ctx.save_for_backward(Q, K, V, O, lse)
        ctx.cfg = (h, float(pdrop), int(site), seed, scale, bits)
        return O

    @staticmethod
    def backward(ctx, dO):
        Q, K, V, O, lse = ctx.saved_tensors
        h, pdrop, site, seed, scale, bits = ctx.cfg
        B, Pn, E = Q.shape
        dO = _c(dO)
        dQ, dK, dV = torch.empty_like(Q), torch.empty_like(K), torch.empty_like(V)
        ws = ops.new(Q, B, h, Pn)
        _sattn_select()
        ops.call("tatt_sattn_bwd_bits", ops.P(Q), ops.P(K), ops.P(V), ops.P(O), ops.P(lse), ops.P(dO), ops.P(bits), ops.P(dQ), ops.P(dK),
                 ops.P(dV), ops.P(ws), B, Pn, h, scale, pdrop, ops.P(seed), site, ops.stream())
        return dQ, dK, dV, None, None, None


class AttnLnFn(Function):
    """LayerNorm(x + W_o attention(W_q x, W_k x, W_v x)) -- the attention sub-layer of the TBSRN FeatureEnhancer with its residual
    LayerNorm (reference model/tbsrn.py:80-86, 119-151) as ONE operator: QKVProjFn, SelfAttnFlashFn, the output projection and
    LayerNormFn back to back, with the same launches.  As with FeedForwardLnFn the composition is for the backward: the gradient the
    residual carries is the addend of the first of the three data-gradient GEMMs into x (tatt_tokgemm_sb_add) instead of an element-wise
    launch by autograd.  Prepacked weights (linear_prepack); d_k = 32, a token count the score-free kernels take."""

    @staticmethod
    def forward(ctx, x, wq, bq, wk, bk, wv, bv, wo, bo, gamma, beta, eps, mode, h, pdrop, site):
        ops._check_dev(x)
        B, Pn, E = x.shape
        x2 = x.reshape(-1, E)
        M = x2.shape[0]
        pks = [_packed_linear(w, M) for w in (wq, wk, wv, wo)]
        q, k, v = (_tokgemm_ex(x2, pk[0], b, E, E).reshape(B, Pn, E) for pk, b in zip(pks[:3], (bq, bk, bv)))
        scale = 1.0 / math.sqrt(E // h)
        seed = current_seed(x.device) if pdrop > 0.0 else None
        O, lse = torch.empty_like(q), ops.new(q, B, h, Pn)
        _sattn_select()
        bits = torch.empty(B * h * Pn * (Pn // 32), device=x.device, dtype=torch.int32) if (pdrop > 0.0 and SATTN_KEEP_BITS and Pn % 32 == 0) else None
        ops.call("tatt_sattn_fwd_bits", ops.P(q), ops.P(k), ops.P(v), ops.P(O), ops.P(lse), ops.P(bits), B, Pn, h, scale, float(pdrop),
                 ops.P(seed), int(site), ops.stream())
        a = _tokgemm_ex(O.reshape(-1, E), pks[3][0], bo, E, E)
        out, stats = ops.ln_fwd(x2, a, gamma, beta, eps, mode)
        ctx.save_for_backward(x, q, k, v, O, lse, a, stats, wq, wk, wv, wo, gamma)
        ctx.wbk = [pk[1] for pk in pks]
        ctx.cfg = (h, float(pdrop), int(site), seed, scale, bits, eps, mode)
        ctx.has_b = [b is not None for b in (bq, bk, bv, bo)]
        ctx.leaves = (wq, bq, wk, bk, wv, bv, wo, bo)
        return out.reshape(x.shape)

    @staticmethod
    def backward(ctx, dout):
        x, q, k, v, O, lse, a, stats, wq, wk, wv, wo, gamma = ctx.saved_tensors
        h, pdrop, site, seed, scale, bits, eps, mode = ctx.cfg
        B, Pn, E = x.shape
        x2, O2 = x.reshape(-1, E), O.reshape(-1, E)
        M = x2.shape[0]
        dxa, _, dg, db = ops.ln_bwd(x2, a, _c(dout).reshape(-1, E), stats, gamma, eps, mode)      # d(x + a)
        dO = _tokgemm_ex(dxa, ctx.wbk[3], None, E, E).reshape(B, Pn, E)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        ws = ops.new(q, B, h, Pn)
        _sattn_select()
        ops.call("tatt_sattn_bwd_bits", ops.P(q), ops.P(k), ops.P(v), ops.P(O), ops.P(lse), ops.P(dO), ops.P(bits), ops.P(dq), ops.P(dk),
                 ops.P(dv), ops.P(ws), B, Pn, h, scale, pdrop, ops.P(seed), site, ops.stream())
        dq2, dk2, dv2 = dq.reshape(-1, E), dk.reshape(-1, E), dv.reshape(-1, E)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.new(dxa, M, E)                                  # = dxa (residual) + dq Wq + dk Wk + dv Wv; dxa stays for W_o's gradient
            ops.call("tatt_tokgemm_sb_add", ops.P(dq2), ops.P(ctx.wbk[0]), None, ops.P(dxa), ops.P(dx), M, E, E, ops.stream())
            _tokgemm_ex(dk2, ctx.wbk[1], None, E, E, out=dx, accum=True)
            _tokgemm_ex(dv2, ctx.wbk[2], None, E, E, out=dx, accum=True)
            dx = dx.reshape(x.shape)
        has_b = ctx.has_b

        def param_grads():
            res = []
            for d, src, hb in ((dq2, x2, has_b[0]), (dk2, x2, has_b[1]), (dv2, x2, has_b[2]), (dxa, O2, has_b[3])):
                dw = ops.new(d, E, E)
                dbias = ops.new(d, E) if hb else None
                _linear_wgrad(d, src, dw, dbias)
                res += [dw, dbias]
            return tuple(res)
        g = SIDE.submit(ctx.leaves, param_grads, x, O, dq, dk, dv, dxa)
        return (dx,) + tuple(g) + (dg, db, None, None, None, None, None)


def attention_ln(x, mh, gamma, beta, eps, mode, pdrop, site):
    """LayerNorm_mode(x + linears[3](attention(linears[0..2](x)))) of a MultiHeadedAttention holder `mh`: one operator when the weights
    are prepacked and the score-free kernels take the geometry, else the operator chain"""
    x = _c(x)
    B, Pn, E = x.shape
    lq, lk, lv, lo = mh.linears[0], mh.linears[1], mh.linears[2], mh.linears[3]
    if (ATTN_LN_FUSED and SATTN_FLASH and E == 32 * mh.h and E <= 128 and Pn % 64 == 0
            and all(_packed_linear(l.weight, B * Pn) is not None and tuple(l.weight.shape) == (E, E) for l in (lq, lk, lv, lo))):
        return AttnLnFn.apply(x, lq.weight, lq.bias, lk.weight, lk.bias, lv.weight, lv.bias, lo.weight, lo.bias, gamma, beta, eps, mode,
                              mh.h, float(pdrop), site)
    q, k, v = qkv_projection(x, lq, lk, lv)
    a = self_attention(q, k, v, mh.h, pdrop, site)
    a = linear(a, lo.weight, lo.bias)
    return LayerNormFn.apply(x, a, gamma, beta, eps, mode, 0.0, 0)


ATTN_LN_FUSED = True        # test / A-B hook: False -> projections, attention, output projection and LayerNorm as separate operators


def self_attention(Q, K, V, h, pdrop, site):
    """Multi-head self-attention core over (B, P, E) projected tensors: the score-free kernels when they apply, else the
    materialised path (any head width / ragged P)."""
    B, Pn, E = Q.shape
    if E == 32 * h and Pn % 64 == 0 and SATTN_FLASH:
        return SelfAttnFlashFn.apply(_c(Q), _c(K), _c(V), h, pdrop, site)
    return SelfAttnCoreFn.apply(Q, K, V, h, pdrop, site)


SATTN_FLASH = True          # test / A-B hook: False -> materialised scores (SelfAttnCoreFn)
SATTN_SB = True             # tatt_amd.set_arithmetic: False -> the exact-fp32 kernels of csrc/sattn.hip
SATTN_KEEP_BITS = True      # test / A-B hook: False -> all three attention kernels recompute the dropout masks


class CatPEFn(Function):
    """tokens (B,P,C) ++ positional table (P,Cp) broadcast over the batch -> (B,P,C+Cp)  (reference model/tbsrn.py:84-87)."""

    @staticmethod
    def forward(ctx, x, pe):
        B, Pn, C = x.shape
        Cp = pe.shape[1]
        out = ops.new(x, B, Pn, C + Cp)
        ops.copy4d(x, out, (1, B, Pn, C), (0, Pn * C, C, 1), (0, Pn * (C + Cp), C + Cp, 1))
        ops.copy4d(pe, out.reshape(-1)[C:], (1, B, Pn, Cp), (0, 0, Cp, 1), (0, Pn * (C + Cp), C + Cp, 1))
        ctx.C = C
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = _c(dout)
        B, Pn, Ct = dout.shape
        C = ctx.C
        dx = ops.new(dout, B, Pn, C)
        ops.copy4d(dout, dx, (1, B, Pn, C), (0, Pn * Ct, Ct, 1), (0, Pn * C, C, 1))
        return dx, None


class ImageLossFn(Function):
    """reference ImageLoss(gradient=True, loss_weight=[w0, w1]) (loss/image_loss.py:19-34): per-sample loss (B,) when `scale` is
    None, else the scalar scale * mean_b (the training loop's `.mean() * 100`) -- one forward and one backward kernel."""

    @staticmethod
    def forward(ctx, sr, hr, w0, w1, scale):
        ops._check_dev(sr)
        ops._check_dev(hr)
        B, C, H, W = sr.shape
        assert hr.shape == sr.shape
        per = ops.new(sr, B)
        mean = ops.new(sr, 1) if scale is not None else None
        ops.call("tatt_image_loss_fwd", ops.P(sr), *sr.stride(), ops.P(hr), *hr.stride(), ops.P(per), ops.P(mean),
                 0.0 if scale is None else float(scale), B, C, H, W, w0, w1, ops.stream())
        ctx.save_for_backward(sr, hr)
        ctx.cfg = (w0, w1, scale)
        return per if scale is None else mean.reshape(())

    @staticmethod
    def backward(ctx, g):
        sr, hr = ctx.saved_tensors
        w0, w1, scale = ctx.cfg
        B, C, H, W = sr.shape
        dsr = torch.empty_like(sr)                       # keeps sr's (channels-last) strides
        assert dsr.stride() == sr.stride()
        g = _c(g)
        ops.call("tatt_image_loss_bwd", ops.P(sr), *sr.stride(), ops.P(hr), *hr.stride(), ops.P(g) if scale is None else None,
                 None if scale is None else ops.P(g), 0.0 if scale is None else float(scale), ops.P(dsr), B, C, H, W, w0, w1,
                 ops.stream())
        return dsr, None, None, None, None


# --------------------------------------------------------------------------------------------------
# CRNN text-prior generator pieces
# --------------------------------------------------------------------------------------------------
class Conv2x2ValidFn(Function):
    """nn.Conv2d(Cin, Cout, 2, stride 1, padding 0) on an NHWC map (reference model/crnn/crnn.py:34-47, conv6): evaluated by the
    implicit-GEMM kernel on the full input grid (its padding for a 2x2 filter is 0), the last row / column are dropped."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        B, H, W, Cin = x.shape
        Cout = weight.shape[0]
        full = ops.conv_fwd(x, ops.repack_weight(weight, 0), bias, Cout, 2, 2)
        y = ops.new(x, B, H - 1, W - 1, Cout)
        ops.copy4d(full, y, (B, H - 1, W - 1, Cout), full.stride(), y.stride())
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        B, H, W, Cin = x.shape
        Cout = weight.shape[0]
        dy = _c(dy)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            # dx[h,w] = sum_{kh,kw} dy[h-kh, w-kw] W[kh,kw]: a forward pass of the flipped filter over dy shifted by (+1,+1)
            sh = torch.zeros(B, H, W, Cout, device=x.device)
            ops.copy4d(dy, sh[:, 1:, 1:], dy.shape, dy.stride(), sh.stride())
            dx = ops.conv_fwd(sh, ops.repack_weight(weight, 1), None, Cin, 2, 2)
        if ctx.needs_input_grad[1]:
            full = torch.zeros(B, H, W, Cout, device=x.device)
            ops.copy4d(dy, full, dy.shape, dy.stride(), full.stride())
            dw = ops.conv_wgrad(x, full, Cout, 2, 2)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = ops.colsum(dy.reshape(-1, Cout))
        return dx, dw, db


class BiLSTMFn(Function):
    """nn.LSTM(I, H, bidirectional=True) on a time-major sequence x (T, Bt, I) -> (T, Bt, 2H), h0 = c0 = 0
    (reference model/crnn/crnn.py:10,20).  Input projections of all steps and both directions: two GEMMs into one (T*Bt, 8H)
    buffer; recurrence: one fused launch per time step (tatt_lstm_fwd_step / tatt_lstm_bwd_step); weight gradients: GEMMs over
    the saved sequences."""

    @staticmethod
    def forward(ctx, x, wih_f, whh_f, bih_f, bhh_f, wih_r, whh_r, bih_r, bhh_r):
        T, Bt, I = x.shape
        H = whh_f.shape[1]
        x2 = x.reshape(T * Bt, I)
        gi = ops.new(x, T * Bt, 8 * H)
        ops.linear_fwd(x2, wih_f, bih_f, out=gi[:, :4 * H])
        ops.linear_fwd(x2, wih_r, bih_r, out=gi[:, 4 * H:])
        out = ops.new(x, T, Bt, 2 * H)
        cseq = ops.new(x, 2, T, Bt, H)
        gsave = ops.new(x, 2, T, Bt, 4, H)
        for s in range(T):
            ops.call("tatt_lstm_fwd_step", ops.P(gi), ops.P(whh_f), ops.P(whh_r), ops.P(bhh_f), ops.P(bhh_r), ops.P(out),
                     ops.P(cseq), ops.P(gsave), T, Bt, H, s, ops.stream())
        ctx.save_for_backward(x, wih_f, whh_f, wih_r, whh_r, out, cseq, gsave)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, wih_f, whh_f, wih_r, whh_r, out, cseq, gsave = ctx.saved_tensors
        T, Bt, I = x.shape
        H = whh_f.shape[1]
        dout = _c(dout)
        whhT = [ops.to_contiguous(w.t().reshape(1, 1, H, 4 * H)).reshape(H, 4 * H) for w in (whh_f, whh_r)]
        dgates = ops.new(x, 2, T, Bt, 4 * H)
        dcc = ops.new(x, 2, Bt, H)
        for s in range(T - 1, -1, -1):
            ops.call("tatt_lstm_bwd_step", ops.P(dout), ops.P(whhT[0]), ops.P(whhT[1]), ops.P(cseq), ops.P(gsave), ops.P(dgates),
                     ops.P(dcc), T, Bt, H, s, ops.stream())
        x2 = x.reshape(T * Bt, I)
        grads = []
        dx = None
        for d, (wih, whh) in enumerate(((wih_f, whh_f), (wih_r, whh_r))):
            dg = dgates[d].reshape(T * Bt, 4 * H)
            db = ops.new(x, 4 * H)
            dwih = ops.linear_bwd_weight(dg, x2, rowsum=db)
            # h_{t-1} of direction d at time t is out[t-1] (forward) / out[t+1] (reverse): contiguous time slices
            hp = out[:T - 1, :, :H] if d == 0 else out[1:, :, H:]
            dgs = dgates[0, 1:] if d == 0 else dgates[1, :T - 1]
            dwhh = ops.linear_bwd_weight(dgs.reshape(-1, 4 * H), _rows(hp, H))
            if ctx.needs_input_grad[0]:
                if dx is None:
                    dx = ops.linear_bwd_input(dg, wih)
                else:
                    ops.linear_bwd_input(dg, wih, out=dx, beta=1.0)
            grads.append((dwih, dwhh, db))
        (dwih_f, dwhh_f, db_f), (dwih_r, dwhh_r, db_r) = grads
        return (None if dx is None else dx.reshape(T, Bt, I), dwih_f, dwhh_f, db_f, db_f, dwih_r, dwhh_r, db_r, db_r)


def _rows(hp, H):
    """(T', Bt, H) slice of the (T, Bt, 2H) output -> 2-D (T'*Bt, H) view with row pitch 2H (no copy)."""
    Tn, Bt, _ = hp.shape
    return hp.as_strided((Tn * Bt, H), (hp.stride(1), 1), hp.storage_offset())


def bilstm(x, rnn):
    """rnn: an nn.LSTM(I, H, bidirectional=True) used as parameter holder."""
    return BiLSTMFn.apply(_c(x), rnn.weight_ih_l0, rnn.weight_hh_l0, rnn.bias_ih_l0, rnn.bias_hh_l0,
                          rnn.weight_ih_l0_reverse, rnn.weight_hh_l0_reverse, rnn.bias_ih_l0_reverse, rnn.bias_hh_l0_reverse)


class SoftmaxRowsFn(Function):
    """softmax over the last axis of a 2-D tensor (rows x L, L <= 4096): the class softmax that turns recogniser logits into the
    text prior (reference interfaces/super_resolution.py:796)."""

    @staticmethod
    def forward(ctx, x):
        rows, L = x.shape
        p = x.clone()
        ops.call("tatt_softmax_rows_fwd", ops.P(p), None, rows, L, 0.0, ops.P(seed_tensor(x.device)), 0, ops.stream())
        ctx.save_for_backward(p)
        return p

    @staticmethod
    def backward(ctx, dp):
        (p,) = ctx.saved_tensors
        rows, L = p.shape
        ds = _c(dp).clone()
        ops.call("tatt_softmax_rows_bwd", ops.P(p), ops.P(ds), rows, L, 0.0, ops.P(seed_tensor(p.device)), 0, ops.stream())
        return ds


class SemanticLossFn(Function):
    """reference SemanticLoss.forward (loss/semantic_loss.py:21-38): L1 + KL between the student prior `pred` and the (detached)
    teacher prior `gt`; scalar."""

    @staticmethod
    def forward(ctx, pred, gt):
        ops._check_dev(pred)
        ops._check_dev(gt)
        pred, gt = _c(pred), _c(gt)
        assert pred.shape == gt.shape
        out = ops.new(pred, 1)
        ops.call("tatt_semantic_loss_fwd", ops.P(pred), ops.P(gt), pred.numel(), ops.P(out), ops.stream())
        ctx.save_for_backward(pred, gt)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        pred, gt = ctx.saved_tensors
        d = torch.empty_like(pred)
        ops.call("tatt_semantic_loss_bwd", ops.P(pred), ops.P(gt), ops.P(_c(g)), pred.numel(), ops.P(d), ops.stream())
        return d, None


# --------------------------------------------------------------------------------------------------
# SSIM / TRI_SSIM and the rotation augmentation of the shipped recipe
# --------------------------------------------------------------------------------------------------
class SsimFn(Function):
    """Per-sample mean of the SSIM map over (C,H,W): reference _ssim (utils/ssim_psnr.py:76-97) for two images, _tri_ssim
    (:99-129) for three; 11x11 Gaussian window, zero padding.  One forward kernel; backward = coefficient maps + one filter pass."""

    @staticmethod
    def forward(ctx, x1, x2, x3):
        ops._check_dev(x1)
        ops._check_dev(x2)
        B, C, H, W = x1.shape
        assert x2.shape == x1.shape and (x3 is None or x3.shape == x1.shape)
        out = ops.new(x1, B)
        tiles = ops.cdiv(H, 16) * ops.cdiv(W, 64)
        ws = ops.new(x1, B * C * tiles, dtype=torch.float64)
        s3 = x3.stride() if x3 is not None else (0, 0, 0, 0)
        ops.call("tatt_ssim_fwd", ops.P(x1), *x1.stride(), ops.P(x2), *x2.stride(), ops.P(x3), *s3, B, C, H, W, ops.P(out), ops.P(ws),
                 ops.stream())
        ctx.save_for_backward(x1, x2, x3)
        return out

    @staticmethod
    def backward(ctx, g):
        x1, x2, x3 = ctx.saved_tensors
        B, C, H, W = x1.shape
        need = ctx.needs_input_grad
        maps = ops.new(x1, B * C * 5 * H * W)
        dx = [ops.new(x1, B, C, H, W) if (need[i] and (i < 2 or x3 is not None)) else None for i in range(3)]
        s3 = x3.stride() if x3 is not None else (0, 0, 0, 0)
        ops.call("tatt_ssim_bwd", ops.P(x1), *x1.stride(), ops.P(x2), *x2.stride(), ops.P(x3), *s3, B, C, H, W, ops.P(_c(g)), ops.P(maps),
                 ops.P(dx[0]), ops.P(dx[1]), ops.P(dx[2]), ops.stream())
        return dx[0], dx[1], dx[2]


class AffineSampleFn(Function):
    """F.grid_sample(x, F.affine_grid(theta, x.shape)) -- bilinear, zeros padding, align_corners=False -- with the gradient
    w.r.t. the IMAGE (theta carries none): the rotation of the training recipe is applied to a network output
    (interfaces/super_resolution.py:910-914)."""

    @staticmethod
    def forward(ctx, x, theta):
        ops._check_dev(x)
        B, C, H, W = x.shape
        theta = _c(theta.reshape(B, 6).to(x.device, torch.float32))
        out = ops.new(x, B, C, H, W)
        ops.call("tatt_affine_sample_fwd", ops.P(x), *x.stride(), ops.P(theta), ops.P(out), B, C, H, W, ops.stream())
        ctx.save_for_backward(theta)
        ctx.shape = (B, C, H, W)
        return out

    @staticmethod
    def backward(ctx, dout):
        (theta,) = ctx.saved_tensors
        B, C, H, W = ctx.shape
        dimg = ops.new(dout, B, C, H, W)
        ops.call("tatt_affine_sample_bwd", ops.P(theta), ops.P(_c(dout)), ops.P(dimg), B, C, H, W, ops.stream())
        return dimg, None
