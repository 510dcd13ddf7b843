"""Training-step harness for the TATT hot path: loss, global-norm clip, Adam, data parallelism.

Mirrors the reference's hot loop (interfaces/super_resolution.py:873-894,1072-1085):
    sr = model(lr, text_prior); loss = ImageLoss(sr, hr).mean()*100; zero_grad; backward;
    clip_grad_norm_(0.25); Adam(1e-3, betas=(0.5, 0.999)).step()
and replaces torch.nn.DataParallel (interfaces/base.py:386-396) by one process per GPU with an RCCL
all-reduce of ONE flat gradient buffer over xGMI (`torch.distributed`, backend "nccl" = RCCL on ROCm).

* Parameters, gradients and Adam moments live in flat fp32 buffers (7.6 M elements = 30.4 MB each); the module's
  nn.Parameters are views into the flat parameter buffer; the HIP backward kernels produce fresh gradient tensors which one
  multi-tensor copy per bucket gathers into the flat gradient buffer (`.grad` is NOT a view of it).
* Clip + Adam are two HIP kernels (tatt_l2norm, tatt_adam_step) whose step-varying scalars live in device
  memory, so a whole step (forward, loss, backward, optimiser) can be captured once as a hipGraph and replayed.
* ImageLoss (loss/image_loss.py) is one fused forward and one fused backward HIP kernel (tatt_image_loss_*) instead of
  ~80 element-wise torch kernels on the (B,4,2H,2W) images.
"""
from __future__ import annotations

import os
from typing import Optional

import torch

from . import functional as Fh
from . import ops
from .dp import FlatParams, GradCuts, broadcast_model, rank_dropout_seed


def image_loss(sr, hr, weights=(1.0, 1e-4)):
    """ImageLoss(gradient=True, loss_weight=[1, 1e-4]).forward (reference loss/image_loss.py:19-34): per-sample loss (B,)."""
    return Fh.ImageLossFn.apply(sr, hr, float(weights[0]), float(weights[1]), None)


def image_loss_mean(sr, hr, weights=(1.0, 1e-4), scale=100.0):
    """`ImageLoss(sr, hr).mean() * scale` of the training loop (interfaces/super_resolution.py:889-894) as one scalar."""
    return Fh.ImageLossFn.apply(sr, hr, float(weights[0]), float(weights[1]), float(scale))


def semantic_loss(pred, gt):
    """SemanticLoss()(pred, gt) of the reference (loss/semantic_loss.py:21-38): prior distillation, student vs teacher."""
    return Fh.SemanticLossFn.apply(pred, gt.detach())


def calculate_psnr(img1, img2):
    """calculate_psnr of the reference (utils/ssim_psnr.py:9-15): device scalar, images in [0, 1], first 3 channels."""
    ops._check_dev(img1)
    ops._check_dev(img2)
    B, C, H, W = img1.shape
    out = ops.new(img1, 1)
    ops.call("tatt_psnr", ops.P(img1), *img1.stride(), ops.P(img2), *img2.stride(), ops.P(out), B, C, H, W, ops.stream())
    return out.reshape(())


class TextPriorSR(torch.nn.Module):
    """The generator together with its trainable text-prior generator, as the reference's loop composes them: lr image ->
    parse_crnn_data -> CRNN student -> softmax prior -> SR(x, prior).  With a frozen `teacher` recogniser the step also carries the
    prior-distillation term of the reference, `sem_loss(student prior on LR, teacher prior on HR) * 100`
    (super_resolution.py:767,879): `Trainer` adds `extra_loss(hr)` to the image loss.

    `detach_prior` selects which of the reference's two compositions is built:
      * True (default) -- `--arch tatt` and the rest of ABLATION_SET (interfaces/super_resolution.py:59-61, 770-914): the SR generator
        receives `label_vecs_final.detach()` (:873); the student learns from the distillation term alone; the second forward of the
        TSSIM recipe (:911) gets the SAME detached prior (`forward(x, reuse_prior=True)`: the student is not run again);
      * False -- `--arch tsrn_tl` / `tsrn_tl_wmask` (:729-768): `model(images_lr, label_vecs_final)`, gradients of the SR loss reach the
        recogniser through the prior."""

    def __init__(self, sr, tpg, teacher=None, in_width=100, detach_prior=True):
        super().__init__()
        self.sr, self.tpg, self.in_width = sr, tpg, in_width
        self.detach_prior = bool(detach_prior)
        if teacher is not None:                              # the reference calls aster.eval() and never optimises it
            teacher.eval()
            teacher.requires_grad_(False)
        object.__setattr__(self, "_teacher", teacher)        # frozen: deliberately NOT a registered sub-module / parameter owner
        self._student_probs = None
        self._prior_cache = None                             # detached prior of the last full forward (recipe's second forward)
        self._gt_ahead = None                                # (stream, teacher prior) of begin_teacher
        self._teacher_streams = {}

    def _apply(self, fn, *args, **kwargs):                 # .to(device) moves the (unregistered) teacher too
        super()._apply(fn, *args, **kwargs)
        if self._teacher is not None:
            self._teacher._apply(fn, *args, **kwargs)
        return self

    @property
    def block(self):
        return self.sr.block

    @block.setter
    def block(self, v):
        self.sr.block = v

    # -- Trainer protocol: the SR generator's stages, then one more for the recogniser --------------------------------------
    def grad_buckets(self, dp=False):
        """The student receives gradient from the distillation loss (first backward stage) and -- with detach_prior=False -- from the
        SR generator's text encoder (stage "tp"), so its own backward is a last stage, "tpg", fed by their sum (forward() cuts there)."""
        b = [(n, list(ps)) for n, ps in self.sr.grad_buckets(dp)]
        b.append(("tpg", list(self.tpg.parameters())))
        return b

    def set_grad_cuts(self, cuts):
        self.sr.set_grad_cuts(cuts)

    def clip_groups(self, clip):
        """The reference clips each model of `model_list` by its own norm (`for model in model_list: clip_grad_norm_(...)`,
        interfaces/super_resolution.py:1082-1083) and leaves the recogniser's gradients unclipped."""
        return [(list(self.sr.parameters()), clip), (list(self.tpg.parameters()), 0.0)]

    def frozen_tensors(self):
        if self._teacher is None:
            return []
        return list(self._teacher.parameters()) + list(self._teacher.buffers())

    def _probs(self, net, img):
        from .crnn import parse_crnn_data
        logits = net(parse_crnn_data(img[:, :3], self.in_width))                 # (T, B, 37)
        T, B, C = logits.shape
        return Fh.SoftmaxRowsFn.apply(logits.reshape(T * B, C)).reshape(T, B, C)

    def forward(self, x, reuse_prior=False):
        if reuse_prior:                  # interfaces/super_resolution.py:911: the first forward's prior, detached; no second student pass
            assert self._prior_cache is not None, "reuse_prior needs a full forward first"
            return self.sr(x, self._prior_cache)
        # the student's pass needs only the LR image: with STUDENT_FORK it is a parallel branch (stream of its own) beside the generator's
        # STN head and first convolution, joined where the generator first reads the prior (tsrn._trunk_forward); its backward (stage
        # "tpg") runs on that stream too -- the autograd engine orders it against the streams that feed and follow it.  Off by default.
        if STUDENT_FORK:
            probs = Fh.FWD_FORK_B.run(x, lambda: self._probs(self.tpg, x))
        else:
            probs = self._probs(self.tpg, x)
        cuts = getattr(self.sr, "_grad_cuts", None) if self.training else None
        # staged backward: every consumer of the prior continues on its own detached copy; stage "tpg" adds their gradients up
        self._student_probs = cuts.cut("tpg", probs) if cuts else probs
        if self.detach_prior:
            p_sr = probs.detach()
        else:
            p_sr = cuts.cut("tpg", probs) if cuts else probs
        prior = p_sr.permute(1, 0, 2).unsqueeze(1).permute(0, 3, 1, 2)            # (B, 37, 1, T)
        self._prior_cache = prior.detach()
        out = self.sr(x, prior)
        Fh.FWD_FORK_B.join(x.device)                        # (a generator that never read the prior: join before anything else does)
        return out

    def begin_teacher(self, hr):
        """Trainer hook, called BEFORE forward(): the frozen teacher's pass over the HR batch depends on nothing the step computes, so
        with the Trainer's second lane on it is issued at once on a stream of its own and runs beside the generator's forward (some 80
        small, latency-bound launches: 26 LSTM steps x 2 layers and the chunked convolutions); extra_loss() joins.  Inside the step's
        hipGraph this is one more parallel branch.  Without the second lane (or on the CPU) extra_loss runs the teacher itself."""
        self._gt_ahead = None
        if self._teacher is None or not Fh.FWD_FORK.enabled or not hr.is_cuda or not TEACHER_AHEAD:
            return
        k = str(hr.device)
        if k not in self._teacher_streams:
            self._teacher_streams[k] = torch.cuda.Stream(device=hr.device)
        st = self._teacher_streams[k]
        st.wait_stream(torch.cuda.current_stream(hr.device))
        with torch.cuda.stream(st), torch.no_grad():
            gt = self._probs(self._teacher, hr)
        self._gt_ahead = (st, gt)

    def extra_loss(self, hr):
        """Distillation term; None without a teacher.  Call after forward()."""
        if self._teacher is None:
            return None
        assert not self._teacher.training, "the teacher recogniser must stay in eval mode (reference: aster.eval())"
        if self._gt_ahead is not None:
            st, gt = self._gt_ahead
            self._gt_ahead = None
            torch.cuda.current_stream(hr.device).wait_stream(st)
        else:
            with torch.no_grad():
                gt = self._probs(self._teacher, hr)
        loss = semantic_loss(self._student_probs, gt) * 100.0
        self._student_probs = None
        return loss


STUDENT_FORK = False        # True: the student's pass as a parallel branch of the forward (FWD_FORK_B).  Measured, round 6, same call,
                            # tatt_tpg at B = 48: 8.04 ms in line / 8.15-8.66 ms (bimodal) forked -- a third concurrent branch makes the
                            # graph executor queue the lanes behind each other (profiles/r05_tail_lane_ab.txt); kept as a tested hook
TEACHER_AHEAD = True        # test / A-B hook: False -> the teacher's pass runs inside extra_loss, after the generator's forward


class TssimRecipe:
    """The shipped training recipe (train_TATT.sh: --tssim_loss --rotate_train=5), reference interfaces/super_resolution.py:637-654
    and :873-914: every step draws a rotation angle in [-rotate_train, rotate_train] degrees and an aspect jitter per sample;

        x_rot, hr_rot = rotate(x), rotate(hr);   x_ret = rotate(x_rot, -angle)
        sr = G(x_rot, prior);                    sr_ret = G(x_ret, prior)
        loss = ImageLoss(sr, hr_rot).mean() * 100 + (1 - TRI_SSIM(rotate(sr_ret), sr, hr_rot).mean()) * 10

    Two generator forwards per step (their parameter gradients add up), four image resamplings, one TRI_SSIM.  The angles live in
    device tensors that `new_step` refreshes from the host RNG, so the step stays replayable as a hipGraph."""

    def __init__(self, rotate_train=5.0, seed=0, off_range=0.2):
        import numpy as np
        self.rotate_train, self.off_range = float(rotate_train), off_range
        self.rng = np.random.RandomState(seed)
        self.theta_pos = self.theta_neg = None
        self.last = None

    def draw(self, B):
        """(arcs, rand_offs) as the reference draws them (np.random.rand(B) * 2r - r degrees; np.random.rand(B))."""
        import math
        arcs = torch.tensor((self.rng.rand(B) * self.rotate_train * 2 - self.rotate_train) / 180.0 * math.pi).float()
        offs = torch.tensor(self.rng.rand(B)).float()
        return arcs, offs

    def new_step(self, x, arcs=None, offs=None):
        from .losses import rotation_theta
        B, _, H, W = x.shape
        if arcs is None:
            arcs, offs = self.draw(B)
        self.last = (arcs, offs)
        pos, neg = rotation_theta(arcs, offs, H, W, self.off_range), rotation_theta(-arcs, offs, H, W, self.off_range)
        if self.theta_pos is None:
            self.theta_pos, self.theta_neg = pos.to(x.device), neg.to(x.device)
        else:
            self.theta_pos.copy_(pos)
            self.theta_neg.copy_(neg)

    def loss(self, model, x, tp, hr):
        from .losses import TRI_SSIM
        rot = Fh.AffineSampleFn.apply
        with torch.no_grad():
            x_rot, hr_rot = rot(x, self.theta_pos), rot(hr, self.theta_pos)
            x_ret = rot(x_rot, self.theta_neg)
        # generators that take no external prior (TSRN, TBSRN, TextPriorSR: its student recogniser reads the rotated LR image, as the
        # reference's loop does, interfaces/super_resolution.py:786-815) are called with the image alone
        fwd = (lambda im: model(im, tp)) if tp is not None else model
        # a generator with its own student recogniser: ONE student pass per step, its prior detached and shared by both forwards (:873,911)
        fwd_ret = (lambda im: model(im, reuse_prior=True)) if (tp is None and hasattr(model, "tpg")) else fwd
        if hasattr(model, "begin_teacher"):
            model.begin_teacher(hr_rot)                      # the frozen teacher's pass over hr_rot, beside the first generator forward
        out = fwd(x_rot)
        sr = out[0] if isinstance(out, tuple) else out
        # the distillation term belongs to the FIRST forward (student prior on x_rot vs teacher prior on hr_rot, :767,879); it has to be
        # taken before the second forward overwrites the model's cached student prior
        extra = model.extra_loss(hr_rot) if hasattr(model, "extra_loss") else None
        out = fwd_ret(x_ret)
        sr_ret = out[0] if isinstance(out, tuple) else out
        l_img = image_loss_mean(sr, hr_rot, scale=100.0)
        l_tssim = (1.0 - TRI_SSIM()(rot(sr_ret, self.theta_pos), sr, hr_rot)) * 10.0
        loss = l_img + l_tssim
        return loss if extra is None else loss + extra


class HipStepKernels:
    """The optimiser side of a step on the GPU: global-norm clip + Adam on flat buffers (tatt_l2norm, tatt_adam_step)."""

    def l2norm(self, g, out, ws):
        ops.call("tatt_l2norm", ops.P(g), g.numel(), ops.P(out), ops.P(ws), ops.stream())

    def adam(self, p, g, m, v, lr, b1, b2, eps, gnorm, max_norm, gscale, step):
        ops.call("tatt_adam_step", ops.P(p), ops.P(g), ops.P(m), ops.P(v), p.numel(), lr, b1, b2, eps, ops.P(gnorm), max_norm,
                 gscale, ops.P(step), ops.stream())

    def inc(self, counter):
        ops.inc_i64(counter)

    def guard(self):
        """In front of the optimiser: trap if a launch that synchronises its work-groups in flight gave up waiting (tatt_sync_guard)."""
        ops.call("tatt_sync_guard", ops.stream())

    def after_update(self, device):
        """The weights moved behind torch's back (raw-pointer kernel): rebuild the cached packed filter layouts, one launch."""
        ops.PACKED.refresh(device)


def _default_loss(sr, hr):
    return image_loss_mean(sr, hr, scale=100.0)


class Trainer:
    """One training step per `step()` call: forward, ImageLoss.mean()*100, backward, [all-reduce], clip, Adam.

    The backward runs in the stages the model announces (`grad_buckets` / `set_grad_cuts`: up-sampler + block7, the five SRBs one
    by one, the TP interpreter, block1 + TPS sampler, the STN head).  Each stage has two lanes:
      * main lane: the activation-gradient chain of the stage; weight / bias gradients and the query GRU's backward -- a third of
        the step's kernel time that nothing on that chain waits for -- are only REGISTERED while it runs (`defer_param_grads`,
        tatt_amd.functional.SIDE);
      * side lane: the registered kernels that are due (their split-K reductions batched into one launch) -- those of this stage
        and those the model filed under this stage's bucket although an earlier stage produced them -- then the gather of the
        stage's gradients into its bucket of the flat buffer.
    Pass k of a step runs main lane k beside side lane k-1: with `side_stream` the side lane is a short branch forked onto a
    second HIP stream and joined at the end of the pass; the last stage's side lane follows its main lane on the main stream.  (Data parallel) bucket k-1 is complete after pass k and its asynchronous
    sum all-reduce over RCCL travels while pass k+1 computes.
    `use_graph`: after `warmup_eager` eager steps the step is captured and replayed: ONE hipGraph for the whole step on a single
    GPU (the forks and joins become parallel branches), one per pass + one for the optimiser under data parallelism (the
    collectives are launched between them).
    `process_group`: data parallelism, one process per GPU: rank 0's weights/buffers are broadcast at start, every rank seeds its
    dropout stream differently (DataParallel replicas draw independent masks), 1/world is folded into the Adam kernel, every rank
    then clips and updates identically.
    `kernels` / `loss_fn`: the device kernels behind the optimiser and the loss (default: the HIP ones; the gloo CPU test of
    this orchestration injects torch stand-ins -- there is no CPU path in the product).

    What the second stream buys on ROCm 7 was measured (profiles/README.md round 2, tools/graph_sched_probe.py): parallel branches
    INSIDE one hipGraph overlap well when they are short or their kernels long; separate graph launches (or eager launches) on
    two streams execute one after the other.  Hence short per-stage branches inside the step's graph."""

    def __init__(self, model, lr=1e-3, betas=(0.5, 0.999), eps=1e-8, clip=0.25, use_graph=False, warmup_eager=2,
                 process_group=None, broadcast_init=True, defer_param_grads=True, side_stream=True, kernels=None, loss_fn=None,
                 dropout_seed=None, recipe=None):
        self.model = model
        self.lr, self.betas, self.eps, self.clip = lr, betas, eps, clip
        self.pg = process_group
        # a process group switches the data-parallel path on -- also with a single rank (self-test of the path on one GPU)
        self.dp = process_group is not None
        self.world = torch.distributed.get_world_size(process_group) if self.dp else 1
        self.rank = torch.distributed.get_rank(process_group) if self.dp else 0
        self.kernels = kernels if kernels is not None else HipStepKernels()
        self.loss_fn = loss_fn if loss_fn is not None else _default_loss
        self.recipe = recipe                             # e.g. TssimRecipe: computes the step's loss itself (several forwards)
        self.profile_collectives = False                 # True: time every wait for a collective (collective_report)
        self._coll_events = []
        self._group_events = []                          # (group of passes, start, end) when profiling: GPU time of each group
        dev = next(model.parameters()).device
        self.dev = dev
        self.cuda = dev.type == "cuda"
        self.defer = bool(defer_param_grads)
        self.two_lanes = self.defer and bool(side_stream) and self.cuda
        staged = (self.dp or self.defer) and hasattr(model, "grad_buckets") and hasattr(model, "set_grad_cuts")
        buckets = None
        if hasattr(model, "grad_buckets"):
            import inspect
            takes_dp = "dp" in inspect.signature(model.grad_buckets).parameters
            buckets = model.grad_buckets(dp=self.dp) if takes_dp else model.grad_buckets()
        if buckets is not None and not staged:           # one stage: one bucket, same parameter ORDER as the staged layout
            buckets = [("all", [p for _, ps in buckets for p in ps])]
        self.flat = FlatParams(model, buckets)
        self.stages = self.flat.bucket_names             # stage k fills bucket k; stage 0 is the backward from the loss
        # id(parameter) -> its bucket: a deferred weight-gradient closure runs with the side lane of that stage (or of the stage
        # that produced it, whichever is later), so that the bucket is complete when that side lane gathers it
        self._due = {id(p): k for k, (_, ps) in enumerate(buckets or []) for p in ps}
        self._immediate = frozenset(id(p) for p in model.immediate_grad_params()) if hasattr(model, "immediate_grad_params") else frozenset()
        # installed on the model only while a step's forward runs (_main_lane): a forward made outside the Trainer -- a plain
        # loss.backward() loop, a gradient check -- sees an ordinary, uncut graph
        self.cuts = GradCuts() if staged and len(self.stages) > 1 else None
        self.params = self.flat.params
        self.n = self.flat.n
        self.flat_p, self.flat_g = self.flat.p, self.flat.g
        self.flat_m = torch.zeros(self.n, device=dev)
        self.flat_v = torch.zeros(self.n, device=dev)
        groups = model.clip_groups(clip) if hasattr(model, "clip_groups") else [(self.params, clip)]
        self.groups = [self.flat.span(ps) + (float(c),) for ps, c in groups]          # (start, end, max_norm)
        assert sorted(self.groups)[0][0] == 0 and sum(e - s for s, e, _ in self.groups) == self.n
        if self.dp and broadcast_init:
            extra = model.frozen_tensors() if hasattr(model, "frozen_tensors") else ()
            broadcast_model(self.flat, model, process_group, extra=extra)
        if self.cuda:
            base = 0x1234ABCD5678EF01 if dropout_seed is None else int(dropout_seed)
            if self.dp or dropout_seed is not None:
                Fh.set_seed(dev, rank_dropout_seed(base, self.rank) if self.dp else base)
        self.side = torch.cuda.Stream(device=dev) if self.two_lanes else None
        if self.cuda and isinstance(self.kernels, HipStepKernels):
            Fh.sticky_word(dev)                      # allocated and registered now: never inside the capture of a step
        self._merge_last = len(self.stages) >= 2         # (also without a second stream: one pass structure everywhere)
        self._npass = len(self.stages) + (0 if self._merge_last else 1)
        self.gnorms = [torch.zeros(1, device=dev) for _ in self.groups]
        self.gnorm = self.gnorms[0]
        self.step_count = torch.zeros(1, dtype=torch.int64, device=dev)
        self.norm_ws = torch.empty(1024, dtype=torch.float64, device=dev)
        self.use_graph = use_graph and self.cuda
        self.warmup_eager = warmup_eager
        self._graphs = None
        self._one = None
        self._static = None
        self._nsteps = 0
        self._works = []
        self.last_loss = None
        # Data parallel: the passes run in at most three groups, each followed by ONE all-reduce of the (contiguous) buckets it
        # completed -- [passes 0 .. n-3] -> the trunk / residual-block buckets, [pass n-2] -> the bucket of stage n-3 (TATT: the TP
        # interpreter with the query GRU, 19 MB), [pass n-1] -> the last two.  With `use_graph` a group is one hipGraph: four graph
        # launches and three collectives per step instead of ten and ten.
        n = self._npass
        groups = [list(range(0, n - 2)), [n - 2], [n - 1]] if n >= 3 else [[k] for k in range(n)]
        self._groups = [g for g in groups if g]
        self.reduce_log = []                         # [(last pass of the group, first bucket, last bucket)] of the latest step

    # -- pieces ----------------------------------------------------------------------------------
    def _main_lane(self, k, x=None, tp=None, hr=None):
        """Stage 0: forward + loss + backward from the loss; stage k > 0: the part of the backward `stages[k]` names."""
        Fh.SIDE.enabled = self.defer
        Fh.FWD_FORK.enabled = Fh.FWD_FORK_B.enabled = self.two_lanes
        Fh.SIDE.stage = k
        Fh.SIDE.due_of = self._due
        Fh.SIDE.immediate = self._immediate
        Fh.SIDE.side_stream = self.side if self.two_lanes and len(self.stages) > 1 else None
        try:
            if k == 0:
                for p in self.params:
                    p.grad = None
                if self.cuts is not None:
                    self.cuts.reset()
                    self.model.set_grad_cuts(self.cuts)
                if self.recipe is not None:
                    loss = self.recipe.loss(self.model, x, tp, hr)
                else:
                    if hasattr(self.model, "begin_teacher"):
                        self.model.begin_teacher(hr)
                    out = self.model(x, tp) if tp is not None else self.model(x)
                    sr = out[0] if isinstance(out, tuple) else out
                    loss = self.loss_fn(sr, hr)
                    extra = self.model.extra_loss(hr) if hasattr(self.model, "extra_loss") else None
                    if extra is not None:
                        loss = loss + extra
                if self._one is None or self._one.device != loss.device or self._one.dtype != loss.dtype:
                    self._one = torch.ones_like(loss)            # (allocated once: the engine would fill a fresh one per step)
                loss.backward(self._one)
                self.last_loss = loss.detach()
            else:
                self.cuts.run(self.stages[k])
        finally:
            Fh.SIDE.enabled = False
            Fh.SIDE.side_stream = None
            Fh.FWD_FORK.enabled = Fh.FWD_FORK_B.enabled = False
            if k == 0 and self.cuts is not None:
                self.model.set_grad_cuts(None)
        if k == len(self.stages) - 1 and hasattr(self.model, "block"):
            self.model.block = None                  # do not keep the autograd graph of this step alive

    def _side_lane(self, k):
        """The deferred parameter-gradient kernels due by stage k (everything left at the last stage), then bucket k of the flat
        gradient buffer."""
        Fh.SIDE.flush(None if k == len(self.stages) - 1 else k)
        self.flat.gather_grads(k)

    def _pass(self, k, x=None, tp=None, hr=None):
        """Pass k of a step = the main lane of stage k (k < stages) beside the side lane of stage k - 1 (k >= 1): with
        `two_lanes` the side lane is forked onto the second stream at the start of the pass and joined at its end -- a short
        parallel branch, which the hipGraph executor does overlap (tools/graph_sched_probe.py); otherwise it simply runs first."""
        nst = len(self.stages)
        main = torch.cuda.current_stream(self.dev) if self.cuda else None
        # The LAST stage's own parameter-gradient kernels follow its main lane on the main stream, beside the side lane of the
        # stage before it (the query GRU's 47-launch chain is still running there) -- there is no pass of its own for them.
        merge_last = self._merge_last
        if merge_last and k == nst:
            return
        name = self.stages[k] if k < nst and isinstance(self.stages[k], str) else str(k)
        Fh.stamp("pass %s: start" % name)
        if k >= 1:
            if self.two_lanes:
                self.side.wait_stream(main)
                with torch.cuda.stream(self.side):
                    self._side_lane(k - 1)
                    Fh.stamp("pass %s: side lane done" % name)
            else:
                self._side_lane(k - 1)
        if k < nst:
            self._main_lane(k, x, tp, hr)
            Fh.stamp("pass %s: main lane done" % name)
        joined = False
        if merge_last and k == nst - 1:
            Fh.SIDE.flush(None)
            if k >= 1 and self.two_lanes:
                main.wait_stream(self.side)      # (gradients of this stage's bucket that ran AT ONCE on the side lane: SIDE.immediate)
                joined = True
            self.flat.gather_grads(k)
            Fh.stamp("pass %s: merged side work done" % name)
        if k >= 1 and self.two_lanes and not joined:
            main.wait_stream(self.side)          # (without this per-pass join the side lanes form one long branch, which the
                                                 #  hipGraph executor does not overlap with the main lane: 8.98 ms instead of 7.86)

    def _optim(self):
        if self.cuda and hasattr(self.kernels, "guard"):
            self.kernels.guard()                     # (a one-thread launch: invalid gradients never reach the weights silently)
        if hasattr(self.kernels, "inc"):
            self.kernels.inc(self.step_count)
        else:
            self.step_count += 1
        b1, b2 = self.betas
        for (s, e, max_norm), gn in zip(self.groups, self.gnorms):
            if max_norm > 0.0:
                self.kernels.l2norm(self.flat_g[s:e], gn, self.norm_ws)
            self.kernels.adam(self.flat_p[s:e], self.flat_g[s:e], self.flat_m[s:e], self.flat_v[s:e], self.lr, b1, b2, self.eps,
                              gn, max_norm, 1.0 / self.world, self.step_count)
        if hasattr(self.kernels, "after_update"):
            self.kernels.after_update(self.dev)

    def _reduce(self, lo, hi, after_pass):
        """Data parallel: ONE asynchronous sum all-reduce of buckets lo .. hi (adjacent in the flat buffer), ordered behind the
        work already issued on the current stream (RCCL runs it on its own stream: it overlaps the passes launched next)."""
        if self.dp and hi >= lo:
            self.reduce_log.append((after_pass, lo, hi))
            s, e = self.flat.ranges[lo][0], self.flat.ranges[hi][1]
            self._works.append(torch.distributed.all_reduce(self.flat.g[s:e], op=torch.distributed.ReduceOp.SUM, group=self.pg,
                                                            async_op=True))

    def _buckets_done_by(self, k):
        """Index of the last bucket that is complete once pass k has run (-1: none): pass k gathers bucket k - 1; the last pass
        gathers its own bucket too."""
        nst = len(self.stages)
        return nst - 1 if (self._merge_last and k >= nst - 1) else min(k - 1, nst - 1)

    def _wait_reduces(self):
        """The current stream waits for the collectives (device-side waits; the host does not block on a GPU).  With
        `profile_collectives` an event pair brackets every wait: their distance is the time the step's stream sat blocked behind that
        collective -- its EXPOSED time (0 when it finished under the passes launched after it)."""
        prof = self.profile_collectives and self.cuda
        for i, w in enumerate(self._works):
            if prof:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                w.wait()
                e1.record()
                self._coll_events.append((i, e0, e1))
                del self._coll_events[:-self.PROFILE_KEEP]       # bounded: a long profiled run keeps the latest steps only
            else:
                w.wait()
        self._works = []

    PROFILE_KEEP = 256                               # event pairs kept while `profile_collectives` is on (each step adds a handful)

    def collective_report(self):
        """What the data-parallel step sent and what it cost, for the bench line: the world size as the process group's backend reports it,
        the all-reduces of one step (bytes, the pass they were issued after) and -- if profiled -- the average exposed time of each.
        Profiling costs an event pair per collective and per pass group inside the step: profile a bounded number of steps OUTSIDE a
        timed region (bench.py does); the report resets the buffers."""
        if not self.dp:
            return None
        rep = {"backend": torch.distributed.get_backend(self.pg), "world_size": torch.distributed.get_world_size(self.pg),
               "per_step": [{"after_pass": ap, "buckets": [self.flat.bucket_names[k] for k in range(lo, hi + 1)],
                             "bytes": 4 * (self.flat.ranges[hi][1] - self.flat.ranges[lo][0])} for ap, lo, hi in self.reduce_log]}
        if self._coll_events:
            torch.cuda.synchronize(self.dev)
            n = max(i for i, _, _ in self._coll_events) + 1
            tot, cnt = [0.0] * n, [0] * n
            for i, e0, e1 in self._coll_events:
                tot[i] += e0.elapsed_time(e1)
                cnt[i] += 1
            for i in range(min(n, len(rep["per_step"]))):
                rep["per_step"][i]["exposed_ms"] = round(tot[i] / max(cnt[i], 1), 4)
            rep["exposed_ms_per_step"] = round(sum(t / max(c, 1) for t, c in zip(tot, cnt)), 4)
            rep["steps_profiled"] = max(cnt)
        if self._group_events:
            torch.cuda.synchronize(self.dev)
            n = max(i for i, _, _ in self._group_events) + 1
            tot, cnt = [0.0] * n, [0] * n
            for i, e0, e1 in self._group_events:
                tot[i] += e0.elapsed_time(e1)
                cnt[i] += 1
            rep["pass_groups"] = [{"passes": [self.stages[k] if k < len(self.stages) else "side(%s)" % self.stages[-1] for k in g],
                                   "gpu_ms": round(tot[i] / max(cnt[i], 1), 4)} for i, g in enumerate(self._groups)]
        self._coll_events, self._group_events = [], []
        return rep

    @property
    def last_grad_norm(self):
        """||g||_2 of the (rank-averaged) gradient of the first clip group (the SR generator) before clipping."""
        return self.gnorm / self.world

    # -- public ----------------------------------------------------------------------------------
    def step(self, x, tp, hr):
        """x (B,4,H,W), tp (B,37,1,26) or None (TSRN), hr (B,4,2H,2W), all on this rank's GPU.  Returns the loss of this step
        (a fresh device scalar, no host sync)."""
        self._nsteps += 1
        nst = len(self.stages)
        if self.recipe is not None:
            self.recipe.new_step(x)
        graphs = None
        if self.use_graph and self._nsteps > self.warmup_eager:
            if self._graphs is None:
                self._capture(x, tp, hr)
            graphs = self._graphs
            sx, stp, shr = self._static
            sx.copy_(x)
            shr.copy_(hr)
            if stp is not None:
                stp.copy_(tp)
        if graphs is not None and "step" in graphs:
            graphs["step"].replay()                  # single GPU: the whole step is one graph
            return self.last_loss.clone()            # (the captured tensor is overwritten by the next replay)
        self.reduce_log = []
        sent = -1                                    # last bucket already on the wire
        prof = self.profile_collectives and self.cuda
        for gi, group in enumerate(self._groups):
            if prof:
                g0 = torch.cuda.Event(enable_timing=True)
                g0.record()
            if graphs is None:
                for k in group:
                    self._pass(k, x, tp, hr)
            else:
                graphs["pass"][gi].replay()
            if prof:
                g1 = torch.cuda.Event(enable_timing=True)
                g1.record()
                self._group_events.append((gi, g0, g1))
                del self._group_events[:-self.PROFILE_KEEP]
            done = self._buckets_done_by(group[-1])
            self._reduce(sent + 1, done, group[-1])  # what this group completed: on the wire while the next group computes
            sent = max(sent, done)
        self._wait_reduces()
        if graphs is None:
            Fh.SIDE.release()
            self._optim()
            return self.last_loss
        graphs["optim"].replay()
        return self.last_loss.clone()

    def _capture(self, x, tp, hr):
        """Capturing executes nothing: step() replays right away, so the step that captured is a real step."""
        self._static = (x.clone(), None if tp is None else tp.clone(), hr.clone())
        sx, stp, shr = self._static
        torch.cuda.synchronize()
        nst = len(self.stages)
        if not self.dp:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for k in range(self._npass):
                    self._pass(k, sx, stp, shr)
                Fh.SIDE.release()
                self._optim()
            self._graphs = {"step": g}
            return
        # Data parallel: one graph per pass + one for the optimiser, all in one memory pool (activations saved by stage 0 are read
        # by the later stages); the collectives are launched between the replays.  The RCCL watchdog thread polls events while we
        # capture: only THIS thread's calls are policed (thread_local).
        pool = torch.cuda.graph_pool_handle()
        kw = dict(pool=pool, capture_error_mode="thread_local")
        graphs = {"pass": []}
        for group in self._groups:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, **kw):
                for k in group:
                    self._pass(k, sx, stp, shr)
            graphs["pass"].append(g)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, **kw):
            self._optim()
        graphs["optim"] = g
        Fh.SIDE.release()
        self._graphs = graphs
