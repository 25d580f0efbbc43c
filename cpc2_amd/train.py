"""The caller side of the hot path: factories, the optimisation step and data-parallel glue.

Mirrors (semantics, names) /root/reference/cpc/train.py:27-59 (getCriterion), :72-187
(trainStep / valStep), :472-484 (Adam over criterion + model parameters), :523-527 (DDP wrap) and
/root/reference/cpc/feature_loader.py:202-235 (getEncoder / getAR).  Differences, all deliberate:
  * device-agnostic tensors come from the caller (the reference hard-codes .cuda());
  * parameters and gradients live in ONE flat fp32 buffer each, so the optimiser is one fused
    HIP Adam launch and data parallelism is one RCCL all-reduce per step (instead of DDP buckets);
  * per-step losses stay on the device; the host reads them every `loggingStep` steps only.
"""
import time

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from ._lib import check, ptr, stream_ptr
from .criterion import CPCUnsupersivedCriterion, NoneCriterion, carry_join, first_frames, first_windows, split_windows
from .model import CPCAR, CPCEncoder, CPCModel, join_tail


# --------------------------------------------------------------------------- factories
def getEncoder(args):
    """feature_loader.py:202-212 (CPCEncoder branch)."""
    if getattr(args, "encoder_type", "cpc") in ("mfcc", "lfb"):
        raise NotImplementedError("only the raw-waveform CPCEncoder is on the MI355X hot path")
    return CPCEncoder(args.hiddenEncoder, args.normMode)


def getAR(args):
    """feature_loader.py:215-235: every branch (transformer, the bidirectional GRU of cpc_mode='bert', no_ar, CPCAR)."""
    if args.arMode == "transformer":
        from .transformers import buildTransformerAR
        arNet = buildTransformerAR(args.hiddenEncoder, args.hiddenGar, args.nLevelsGRU,
                                   args.sizeWindow // 160, args.abspos)
        args.hiddenGar = args.hiddenEncoder
        return arNet
    if getattr(args, "cpc_mode", None) == "bert":
        from .model import BiDIRARTangled
        return BiDIRARTangled(args.hiddenEncoder, args.hiddenGar, args.nLevelsGRU)
    if args.arMode == "no_ar":
        from .model import NoAr
        return NoAr()
    return CPCAR(args.hiddenEncoder, args.hiddenGar, args.samplingType == "sequential", args.nLevelsGRU,
                 mode=args.arMode, reverse=getattr(args, "cpc_mode", None) == "reverse")


def getCriterion(args, downsampling, nSpeakers=0, nPhones=0):
    """train.py:27-48 (unsupervised branch)."""
    if getattr(args, "supervised", False):
        raise NotImplementedError("supervised criteria are not on the MI355X hot path")
    if getattr(args, "cpc_mode", None) == "none":
        return NoneCriterion()
    sizeInputSeq = args.sizeWindow // downsampling
    return CPCUnsupersivedCriterion(args.nPredicts, args.hiddenGar, args.hiddenEncoder, args.negativeSamplingExt,
                                    mode=getattr(args, "cpc_mode", None), rnnMode=args.rnnMode,
                                    dropout=getattr(args, "dropout", False), nSpeakers=nSpeakers,
                                    sizeInputSeq=sizeInputSeq,
                                    multihead_rnn=getattr(args, "multihead_rnn", False),
                                    transformer_pruning=getattr(args, "transformer_pruning", 0),
                                    n_skipped=getattr(args, "n_skipped", 0),
                                    growth_rate=getattr(args, "growth_rate", None),
                                    inflection_point_x=getattr(args, "inflection_point_x", None))


# --------------------------------------------------------------------------- flat parameters + fused Adam
class FlatAdam(torch.optim.Optimizer):
    """torch.optim.Adam(params, lr, betas, eps) of train.py:477-479 on one flat buffer.

    Re-homes every parameter into `self.flat` (views, same values) and every .grad into
    `self.flat_grad`, then `step()` is a single cpc_adam_step launch.  State-dict interop with
    torch.optim.Adam is kept through `state_dict()` / `load_state_dict()` (per-parameter
    exp_avg / exp_avg_sq / step).  A torch.optim.Optimizer with ONE param group, so the reference's
    learning-rate schedulers (train.py:501-520: StepLR, the LambdaLR ramp, SchedulerCombiner) attach
    to it unchanged; the learning rate is read from param_groups[0]["lr"] at every step."""

    def __init__(self, params, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, direct_grads=True):
        self.params = [p for p in params]
        self.direct_grads = direct_grads
        if not self.params:
            raise ValueError("FlatAdam got an empty parameter list")
        dev = self.params[0].device
        _lib.require_gpu(*self.params)
        super(FlatAdam, self).__init__(self.params, dict(lr=lr, betas=betas, eps=eps))
        self.lr, self.betas, self.eps = lr, betas, eps
        self.step_count = 0
        total = sum(p.numel() for p in self.params)
        self.flat = torch.empty(total, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        self.offsets = []
        off = 0
        for p in self.params:
            n = p.numel()
            self.flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + n].view(p.shape)
            if direct_grads:
                # the fused backward kernels write each gradient straight into flat_grad and autograd adopts
                # that view as .grad (no per-parameter add kernels); see _lib.grad_buffers
                p._cpc_flat = (self.flat_grad, off)
                p.grad = None
            else:
                p.grad = self.flat_grad[off:off + n].view(p.shape)
            self.offsets.append(off)
            off += n

    def zero_grad(self, set_to_none=False):
        self.flat_grad.zero_()
        if self.direct_grads:
            for p in self.params:
                p.grad = None
            return
        for p, off in zip(self.params, self.offsets):      # re-attach if someone replaced .grad
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * off:
                p.grad = self.flat_grad[off:off + p.numel()].view(p.shape)

    def _gather_stray_grads(self, ranges=None):
        """direct mode: a gradient that autograd did NOT adopt from flat_grad (accumulated or produced by a
        torch op) lives in its own tensor: copy it home before the fused update.  ranges: only the parameters inside
        these [lo, hi) pieces of the flat buffer (the others have been all-reduced already)."""
        base = self.flat_grad.data_ptr()
        for p, off in zip(self.params, self.offsets):
            if ranges is not None and not any(lo <= off < hi for lo, hi in ranges):
                continue
            if p.grad is not None and p.grad.data_ptr() != base + 4 * off:
                self.flat_grad[off:off + p.numel()].copy_(p.grad.reshape(-1))

    def step(self, closure=None, grad_scale=1.0):
        if closure is not None:
            raise NotImplementedError("FlatAdam.step does not re-evaluate a closure")
        if self.flat.is_cuda:
            join_tail(self.flat.device)          # (a deferred recurrent backward's weight gradients: normally joined at the end of backward)
        if self.direct_grads:
            self._gather_stray_grads()
        self.step_count += 1
        lr = self.param_groups[0]["lr"]
        check(_lib.load().cpc_adam_step(ptr(self.flat), ptr(self.flat_grad), ptr(self.exp_avg), ptr(self.exp_avg_sq),
                                        self.flat.numel(), self.step_count, lr, self.betas[0], self.betas[1], self.eps,
                                        grad_scale, stream_ptr(self.flat.device)), "adam_step")

    def state_dict(self):
        state = {}
        for i, (p, off) in enumerate(zip(self.params, self.offsets)):
            n = p.numel()
            state[i] = {"step": torch.tensor(float(self.step_count)),
                        "exp_avg": self.exp_avg[off:off + n].view(p.shape).clone(),
                        "exp_avg_sq": self.exp_avg_sq[off:off + n].view(p.shape).clone()}
        return {"state": state, "param_groups": [{"lr": self.param_groups[0]["lr"], "betas": self.betas,
                                                  "eps": self.eps, "params": list(range(len(self.params)))}]}

    def load_state_dict(self, sd):
        for i, (p, off) in enumerate(zip(self.params, self.offsets)):
            st = sd["state"].get(i)
            if st is None:
                continue
            n = p.numel()
            self.exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
            self.step_count = int(st["step"])
        self.param_groups[0]["lr"] = sd["param_groups"][0]["lr"]


def buildOptimizer(cpcModel, cpcCriterion, lr=2e-4, beta1=0.9, beta2=0.999, epsilon=1e-8):
    """train.py:472-479: criterion parameters first, then model parameters."""
    g_params = list(cpcCriterion.parameters()) + list(cpcModel.parameters())
    return FlatAdam(g_params, lr=lr, betas=(beta1, beta2), eps=epsilon)


# --------------------------------------------------------------------------- learning-rate schedule
def ramp_scheduling_function(n_epoch_ramp, epoch, square_ramp=False):
    """cpc/utils/misc.py:77-83: linear (or squared) warm-up factor over the first n_epoch_ramp epochs."""
    if epoch >= n_epoch_ramp:
        return 1
    frac = (epoch + 1) / n_epoch_ramp
    return frac ** 2 if square_ramp else frac


class SchedulerCombiner:
    """cpc/utils/misc.py:85-122: schedulers activated one after the other.  step() advances every scheduler from the
    one before the current activation boundary on, LAST first -- so while the ramp is active the step schedule's
    epoch counter already runs, and the ramp (stepped last) has the final word on the learning rate."""

    def __init__(self, scheduler_list, activation_step, curr_step=0):
        if len(scheduler_list) != len(activation_step):
            raise ValueError("The number of scheduler must be the same as the number of activation step")
        if activation_step[0] > curr_step:
            raise ValueError("The first activation step cannot be higher than the current step.")
        self.scheduler_list = scheduler_list
        self.activation_step = list(activation_step)
        self.curr_step = curr_step

    def step(self):
        import bisect
        self.curr_step += 1
        first = bisect.bisect_left(self.activation_step, self.curr_step) - 1
        for i in reversed(range(first, len(self.scheduler_list))):
            self.scheduler_list[i].step()

    def __str__(self):
        return "SchedulerCombiner \n(\n" + "".join(f"({i}) {sc} \n" for i, sc in enumerate(self.scheduler_list)) + ")\n"


def buildScheduler(optimizer, schedulerStep=-1, schedulerRamp=None, epochs_done=0):
    """train.py:501-520: StepLR(gamma 0.5) every schedulerStep epochs, a linear ramp over the first schedulerRamp
    epochs, both (combined), or None; then fast-forwarded over the epochs a resumed run has behind it."""
    scheduler = None
    if schedulerStep > 0:
        scheduler = torch.optim.lr_scheduler.StepLR(optimizer, schedulerStep, gamma=0.5)
    if schedulerRamp is not None:
        n_epoch = schedulerRamp
        ramp = torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda epoch: ramp_scheduling_function(n_epoch, epoch),
                                                 last_epoch=-1)
        scheduler = ramp if scheduler is None else SchedulerCombiner([ramp, scheduler], [0, schedulerRamp])
    if scheduler is not None:
        for _ in range(epochs_done):
            scheduler.step()
    return scheduler


# --------------------------------------------------------------------------- data parallel
class DataParallelContext:
    """One process per GPU (train.py:291-295, 523-527 with --distributed).  Replaces the two DDP wrappers by: one
    broadcast of the flat parameter buffer from rank 0 at start, and all-reduces (SUM) of the flat gradient buffer whose
    1/world_size is folded into the Adam kernel.

    Overlap (the reference gets it from DDP's buckets, train.py:523-527): with `early_params` -- the parameters whose
    gradients are complete before the encoder's backward starts: the criterion's and the context network's -- their slices
    of the flat gradient are reduced ASYNCHRONOUSLY from an autograd hook on the encoder's output (`attach`, called by
    cpcStep), i.e. while the encoder's backward kernels run; `reduce_and_step` then reduces the encoder's slice, waits
    for the early ones and updates.  Without early_params (or overlap=False): one blocking all-reduce of the whole buffer.
    `DataParallelContext.for_modules(optimizer, cpcModel, cpcCriterion)` takes the early parameters from the modules.

    ONE backward pass per optimisation step: the hook reduces the early slices at the end of the first backward pass
    after `attach`; a second backward pass before `reduce_and_step` (gradient accumulation) would add onto sums.  `attach`
    raises if it is called again before `reduce_and_step` has consumed a fired hook.

    `trace`: None, or a callable(kind, lo, hi, view) invoked on the compute stream right before every all-reduce
    ("pre": view = flat_grad[lo:hi] as this rank computed it) and after the last one of a step ("post": the whole
    buffer of sums) -- the equivalence tests clone what they see there (tests/dp_job.py)."""

    def __init__(self, optimizer, early_params=None, overlap=True, trace=None, timing=False):
        self.opt = optimizer
        self.trace = trace
        self.timing = timing                    # bench.py: events around the part of a step the exchange holds the stream for
        self._spans = []
        self._coop_seen = None
        self._helper = None                     # the stream the early all-reduces are issued from (attach)
        self.active = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size() if self.active else 1
        self.early, self.late, self._pending, self._fired = [], [], [], False
        if self.active:
            dist.broadcast(self.opt.flat, src=0)
            if early_params and overlap and hasattr(optimizer, "offsets"):
                self._split(early_params)

    @classmethod
    def for_modules(cls, optimizer, cpcModel, cpcCriterion, overlap=True, trace=None):
        """The context the training loop uses: the criterion's and the context network's parameters reduce early."""
        inner = getattr(cpcModel, "module", cpcModel)
        crit = getattr(cpcCriterion, "module", cpcCriterion)
        early = list(crit.parameters()) + (list(inner.gAR.parameters()) if hasattr(inner, "gAR") else [])
        return cls(optimizer, early_params=early, overlap=overlap, trace=trace)

    def _split(self, early_params):
        """[lo, hi) ranges of the flat buffer: early (merged runs of the early parameters) and the rest."""
        ids = {id(p) for p in early_params}
        total = self.opt.flat.numel()
        marks = sorted((off, off + p.numel()) for p, off in zip(self.opt.params, self.opt.offsets) if id(p) in ids)
        for lo, hi in marks:
            if self.early and self.early[-1][1] == lo:
                self.early[-1] = (self.early[-1][0], hi)
            else:
                self.early.append((lo, hi))
        pos = 0
        for lo, hi in self.early:
            if lo > pos:
                self.late.append((pos, lo))
            pos = hi
        if pos < total:
            self.late.append((pos, total))

    def _all_reduce(self, lo, hi, async_op=False):
        """SUM over the ranks of flat_grad[lo:hi], in place, enqueued behind everything the current stream holds."""
        view = self.opt.flat_grad[lo:hi]
        if self.trace is not None:
            self.trace("pre", lo, hi, view)
        return dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=async_op)

    # ---- what the exchange costs a step (bench.py's `comm.exposed_ms_per_step`)
    def timing_reset(self):
        self._spans = []

    def timing_read(self):
        """Mean milliseconds per reduce_and_step() during which the compute stream was held by the exchange (from the start of
        the late all-reduce to the end of the waits for the early ones), or None without a process group / timing."""
        if not (self.active and self.timing and self._spans):
            return None
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in self._spans) / len(self._spans)

    def _span(self):
        if not (self.timing and self.opt.flat_grad.is_cuda):
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    @staticmethod
    def _coop_launches():
        """Cooperative recurrent launches of this process so far (cpc2_hip.h, cpc_coop_launches)."""
        return int(_lib.load().cpc_coop_launches())

    def attach(self, encoder_output):
        """Register the hook that starts the early reductions when the gradient of `encoder_output` is ready (everything
        downstream of the encoder has then written its parameter gradients)."""
        encoder_output = getattr(encoder_output, "_cpc_join_source", encoder_output)   # (criterion.py, grad_join)
        if not (self.active and self.early and encoder_output.requires_grad):
            return
        if self._fired:
            raise RuntimeError("DataParallelContext: a backward pass has already reduced the early gradient slices of this "
                               "step; call reduce_and_step() before the next forward pass (one backward pass per step)")
        # Cooperative recurrent kernels (GRU / LSTM at hidden 256 / 512) need EVERY workgroup resident; a collective's kernels
        # share the device.  The schedule keeps the two apart by stream order: the early all-reduce is issued behind the recurrent
        # backward (it waits for everything the compute stream holds at that point), and reduce_and_step() makes the compute
        # stream wait for every collective before the next step's forward pass.  The first half is asserted here: if the forward
        # pass of this step launched cooperative kernels, a recurrent backward (cooperative or not: the two fit a CU separately)
        # must have been issued before the hook fires.
        on_gpu = self.opt.flat_grad.is_cuda
        now = self._coop_launches() if on_gpu else 0
        fwd_coop = now - (self._coop_seen if self._coop_seen is not None else now)
        bwd_calls_at_attach = int(_lib.load().cpc_recurrent_backward_calls()) if on_gpu else 0

        def start(_grad):
            if not self._fired:                 # once per backward pass
                self._fired = True
                if on_gpu and fwd_coop > 0 and int(_lib.load().cpc_recurrent_backward_calls()) == bwd_calls_at_attach:
                    raise RuntimeError("DataParallelContext: the early all-reduce would be issued BEFORE the recurrent backward of "
                                       "this step (a collective's kernels beside a cooperative kernel that needs every workgroup "
                                       "resident): attach() must be given the encoder output, upstream of the context network")
                if getattr(self.opt, "direct_grads", False):
                    self.opt._gather_stray_grads(self.early)     # a gradient autograd did not write in place (accumulation)
                if not on_gpu:
                    self._pending = [self._all_reduce(lo, hi, async_op=True) for lo, hi in self.early]
                    return
                # The early slices are complete once the compute stream has reached this point AND the library's side stream has
                # finished the context network's deferred weight gradients.  Only the EXCHANGE has to wait for the latter: it is issued
                # from a helper stream that is ordered behind both, and the compute stream goes straight on into the encoder's
                # backward (round 4 joined the side stream into the compute stream here: ~0.1 ms of the encoder's backward waiting
                # for products nothing in it reads).  The deferred calls' buffers stay on the books until the join at the end of the
                # backward pass / in reduce_and_step().
                device = self.opt.flat_grad.device
                if self._helper is None:
                    # (a stream of the library's making: on a hardware queue that is neither the compute stream's nor the side
                    #  stream's -- a pool stream that happened to share the compute stream's queue would park the wait for the side
                    #  stream IN FRONT of the encoder's backward; cpc2_hip.h, cpc_stream_create_apart)
                    import ctypes
                    raw = ctypes.c_void_p()
                    avoid = (ctypes.c_void_p * 1)(torch.cuda.current_stream(device).cuda_stream)
                    check(_lib.load().cpc_stream_create_apart(avoid, 1, ctypes.byref(raw)), "stream_create_apart")
                    self._helper = torch.cuda.ExternalStream(raw.value, device=device)
                self._helper.wait_stream(torch.cuda.current_stream(device))
                with torch.cuda.stream(self._helper):
                    check(_lib.load().cpc_side_tail_wait(stream_ptr(device)), "side_tail_wait")
                    self._pending = [self._all_reduce(lo, hi, async_op=True) for lo, hi in self.early]
        encoder_output.register_hook(start)

    def reduce_and_step(self):
        if self.opt.flat_grad.is_cuda:
            join_tail(self.opt.flat_grad.device)                 # (parameter gradients still on the library's side stream)
        if self.active and self.opt.flat_grad.is_cuda:
            self._coop_seen = self._coop_launches()              # (the next step's forward launches count from here)
        if self.active:
            if self._fired and self._pending:
                if getattr(self.opt, "direct_grads", False):
                    self.opt._gather_stray_grads(self.late)      # (the early slices hold SUMS by now)
                t0 = self._span()
                for lo, hi in self.late:
                    self._all_reduce(lo, hi)
                with _lib.host_wait("collective_wait"):
                    for work in self._pending:
                        work.wait()
                if t0 is not None:
                    self._spans.append((t0, self._span()))
                if self.trace is not None:
                    self.trace("post", 0, self.opt.flat_grad.numel(), self.opt.flat_grad)
                # the optimiser must not take a stray early gradient for the reduced one again
                stray = getattr(self.opt, "direct_grads", False)
                if stray:
                    self.opt.direct_grads = False
                try:
                    self.opt.step(grad_scale=1.0 / self.world)
                finally:
                    if stray:
                        self.opt.direct_grads = True
                self._pending, self._fired = [], False
                return
            if getattr(self.opt, "direct_grads", False):
                self.opt._gather_stray_grads()
            t0 = self._span()
            self._all_reduce(0, self.opt.flat_grad.numel())
            if t0 is not None:
                self._spans.append((t0, self._span()))
            if self.trace is not None:
                self.trace("post", 0, self.opt.flat_grad.numel(), self.opt.flat_grad)
            self._pending, self._fired = [], False
        self.opt.step(grad_scale=1.0 / self.world)


# --------------------------------------------------------------------------- the step
def _context_windows_only(cpcModel):
    """May cpcStep run the context network on the b context windows alone?  train.py:99-103 keeps `c_feature[:b]` and
    `encoded_data[b:]` of the model's 2b-window outputs: the context of the future half is sliced away, its gradient is identically
    zero and so are its terms in every parameter gradient.  Every context network of this package is per window (no statistics
    across windows), so running it on the first b windows gives the same c_feature[:b], the same gradients and the same update.
    Only for the bare CPCModel (a wrapper's forward() is the plug-in boundary and stays the 2b-window call), without span masking
    (its numpy draws cover all 2b rows) and without a state carried across calls (keepHidden stores the hidden state of all
    2b windows)."""
    import os
    from .model import BiDIRAR, BiDIRARTangled, NoAr
    from .transformers import StaticPositionEmbedding, TransformerLayer
    if os.environ.get("CPC_STRICT_2B") or type(cpcModel) is not CPCModel or cpcModel.mask_prob > 0.0:
        return False
    if not hasattr(cpcModel.gEncoder, "forward_channel_last"):
        return False
    # this form calls gEncoder.forward_channel_last and gAR directly: forward (pre-)hooks registered on the model or its encoder
    # (feature taps, profilers) would silently stop firing -- with hooks the reference's own call is kept
    for m in (cpcModel, cpcModel.gEncoder):
        if m._forward_hooks or m._forward_pre_hooks:
            return False
    ar = cpcModel.gAR
    if isinstance(ar, CPCAR):
        return not ar.keepHidden and ar.hidden is None
    if isinstance(ar, (NoAr, BiDIRAR, BiDIRARTangled)):
        return True
    return isinstance(ar, torch.nn.Sequential) and all(isinstance(m, (StaticPositionEmbedding, TransformerLayer)) for m in ar)


def _guard_cooperative_kernels(cpcModel, cpcCriterion):
    """The cooperative recurrent kernels (GRU / LSTM at hidden 256 / 512) need every workgroup resident, i.e. nothing else on the
    CUs while they run.  DataParallelContext keeps its collectives away from them by stream order; the reference's arrangement --
    DistributedDataParallel (or DataParallel over several devices) around model / criterion, train.py:523-532 -- does not: DDP
    all-reduces the criterion's bucket (an RCCL kernel on the same CUs) while the recurrent backward runs.  Handed such a wrapper
    on an RCCL process group, the process switches to the streaming recurrent kernels for good (cpc_coop_set_policy)."""
    wrappers = (torch.nn.parallel.DistributedDataParallel, torch.nn.DataParallel)
    if not (isinstance(cpcModel, wrappers) or isinstance(cpcCriterion, wrappers)):
        return
    multi_dp = any(isinstance(m, torch.nn.DataParallel) and len(m.device_ids or ()) > 1 for m in (cpcModel, cpcCriterion))
    on_rccl = dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl"
    if (multi_dp or on_rccl) and _lib.load().cpc_coop_set_policy(-1) == 0:
        _lib.load().cpc_coop_set_policy(1)


def _context_frames_only(cpcModel, cpcCriterion, frames):
    """May the context network stop after the W = frames - nPredicts steps whose output the criterion reads?  criterion.py:296 keeps
    `cFeature[:, :windowSize]`; a context network that is CAUSAL in time (the recurrent ones, forward direction) and keeps no state
    across calls computes those W frames without the nPredicts behind them, and their gradient is identically zero -- the same
    argument as for the context windows, on the time axis.  Returns W, or 0."""
    import os
    if os.environ.get("CPC_STRICT_FRAMES") or type(cpcCriterion) is not CPCUnsupersivedCriterion or cpcCriterion.mode == "reverse":
        return 0
    ar = cpcModel.gAR
    if not isinstance(ar, CPCAR) or ar.reverse or ar.keepHidden or ar.hidden is not None:
        return 0
    w = frames - cpcCriterion.nPredicts
    return w if w >= 1 else 0


def cpcStep(past, future, label, cpcModel, cpcCriterion, signal_quality=None, dedup=False, dp=None, strict=False):
    """train.py:95-108: model on cat([past, future]); context from the past half, targets from the
    future half; returns (totLoss, allLosses [1,K], allAcc [1,K]).

    Default: the encoder runs on the 2b windows, the context network on the b windows whose context the reference keeps
    (_context_windows_only: identical outputs, gradients and updates).  strict=True: the reference's own dataflow, the
    context network on all 2b windows through CPCModel.forward.

    dedup=True: when `future` IS `past` (no augmentation: dataset.py:308-321 yields the same window twice) the
    two halves of the reference's 2b-window batch are identical and every op up to the criterion is per
    window, so one pass over b windows gives bit-identical c_feature / encoded_data at half the work."""
    b = past.size(0)
    if past.is_cuda:
        _guard_cooperative_kernels(cpcModel, cpcCriterion)
    if dedup and (future is past or (future.data_ptr() == past.data_ptr() and future.shape == past.shape)):
        with _ar_scope(cpcModel):
            c_feature, encoded_data, label = cpcModel(past, label)
        if dp is not None:
            dp.attach(encoded_data)
        with _defer_scope(cpcCriterion, encoded_data):
            allLosses, allAcc = cpcCriterion(c_feature, encoded_data, label, signal_quality)
        return sum_losses(allLosses), allLosses, allAcc
    if not strict and _context_windows_only(cpcModel):
        with _ar_scope(cpcModel):
            # (train.py:99's cat([past, future]) as two pointers: only the first layer reads the waveform)
            encoded_full = cpcModel.gEncoder.forward_channel_last(past, future)      # [2b, T, H]
            frames = encoded_full.size(1)
            w = _context_frames_only(cpcModel, cpcCriterion, frames)
            if w:
                # ... and only over the W frames criterion.py:296 keeps of it (a causal recurrent network: _context_frames_only)
                context_in, encoded_data = split_windows(encoded_full, b, tag=("split_w", w))
                c_feature = cpcModel.gAR(first_frames(context_in, w, frames))         # [b, W, H]
                c_feature._cpc_frames_of = frames
            else:
                context_in, encoded_data = split_windows(encoded_full, b)
                c_feature = cpcModel.gAR(context_in)                                  # [b, T, H]: train.py:102's c_feature[:b]
        if dp is not None:
            dp.attach(encoded_full)
        with _defer_scope(cpcCriterion, encoded_data):
            allLosses, allAcc = cpcCriterion(c_feature, encoded_data, label, signal_quality)
        return sum_losses(allLosses), allLosses, allAcc
    combined = torch.cat([past, future], dim=0)
    # (train.py:100,105 concatenate the labels too and take the first half back: CPCModel hands `label` through untouched, so the
    #  round trip -- two small kernels per step -- is skipped for it; any other model gets the reference's tensors)
    passthrough = isinstance(getattr(cpcModel, "module", cpcModel), CPCModel)
    with _ar_scope(cpcModel):
        c_feature, encoded_full, label2 = cpcModel(combined, label if passthrough else torch.cat([label, label]))
    label = label2 if passthrough else label2[:b]
    if dp is not None:
        dp.attach(encoded_full)             # data parallel: the criterion / context gradients are reduced under the encoder's backward
    c_feature = first_windows(c_feature, b)
    encoded_data = carry_join(encoded_full[b:, :, :], encoded_full, b)
    with _defer_scope(cpcCriterion, encoded_full):
        allLosses, allAcc = cpcCriterion(c_feature, encoded_data, label, signal_quality)
    return sum_losses(allLosses), allLosses, allAcc


# ---- totLoss = allLosses.sum(); totLoss.backward() without the four one-element kernels autograd puts between the criterion's
# forward and backward kernels (ones_like of the root, the sum's expand, the copy that makes it contiguous: ~30 us of a 5 ms step)
_ONES = {}


def _ones(device, n):
    key = (str(device), n)
    t = _ONES.get(key)
    if t is None:
        t = _ONES[key] = torch.ones(n, dtype=torch.float32, device=device)
    return t


class _SumLosses(torch.autograd.Function):
    """x.sum() whose backward hands out a cached tensor of ones when the incoming gradient is backward()'s cached 1.0."""

    @staticmethod
    def forward(ctx, x):
        ctx.shape = x.shape
        return x.sum()

    @staticmethod
    def backward(ctx, g):
        n = 1
        for d in ctx.shape:
            n *= d
        if g.data_ptr() == _ones(g.device, 1).data_ptr():
            return _ones(g.device, n).view(ctx.shape)
        return g.expand(ctx.shape)


def sum_losses(allLosses):
    """train.py:106 (`totLoss = allLosses.sum()`)."""
    if allLosses.is_cuda and allLosses.dtype == torch.float32 and allLosses.requires_grad:
        return _SumLosses.apply(allLosses)
    return allLosses.sum()


def backward(totLoss):
    """train.py:109 (`totLoss.backward()`), seeded with a cached 1.0 instead of a fresh ones_like."""
    if totLoss.is_cuda and totLoss.dtype == torch.float32 and totLoss.dim() == 0:
        totLoss.backward(gradient=_ones(totLoss.device, 1).view(()))
    else:
        totLoss.backward()


def _ar_scope(cpcModel):
    """The context network's and the encoder's deferred parameter gradients (model.py, deferred_weight_gradients) are cpcStep's to
    allow for the BARE model only: the gradients are read by the optimiser after the backward pass (and by DataParallelContext's early
    all-reduce, which joins first) -- a DistributedDataParallel / DataParallel wrapper would read them as they are accumulated."""
    import contextlib
    stack = contextlib.ExitStack()
    if isinstance(cpcModel, CPCModel):
        for m in cpcModel.modules():            # CPCEncoder, CPCAR, TransformerLayer
            if hasattr(m, "deferred_weight_gradients"):
                stack.enter_context(m.deferred_weight_gradients())
    return stack


def _defer_scope(cpcCriterion, encoded_full):
    """The criterion's deferred backward (criterion.py, deferred_backward) is this function's to allow: the encoder output
    never leaves cpcStep, so the criterion call is its only consumer.  Only for the bare criterion module -- wrapped the way
    the reference wraps it (DistributedDataParallel, train.py:526; DataParallel, :531) the wrapper's reducer reads the
    predictors' weight gradients as soon as autograd has accumulated them, before a side stream could have written them."""
    import contextlib
    if isinstance(cpcCriterion, CPCUnsupersivedCriterion):
        return cpcCriterion.deferred_backward(encoded_full)
    return contextlib.nullcontext()


def _check_async(device):
    """Raise if a cooperative recurrent kernel timed out since the last check (it then produced NaN): cpc2_hip.h,
    cpc_async_error_check.  Called where the loop synchronises with the device anyway."""
    from . import _lib
    _lib.check(_lib.load().cpc_async_error_check(_lib.stream_ptr(torch.device(device))), "async error check")


def trainStep(dataLoader, cpcModel, cpcCriterion, optimizer, scheduler, loggingStep, dp=None, device=None):
    """train.py:72-142.  `dataLoader` yields (sequence [b,2,1,L], label [b][, signal_quality])."""
    cpcModel.train()
    cpcCriterion.train()
    device = device or next(cpcModel.parameters()).device
    dp = dp or DataParallelContext.for_modules(optimizer, cpcModel, cpcCriterion)      # (run() builds it once and passes it)
    start_time = time.perf_counter()
    n_examples, it = 0, 0
    sum_loss = sum_acc = None
    logs = {}
    for step, full_data in enumerate(dataLoader):
        sequence, label, *signal_quality = full_data
        sequence = sequence.to(device, non_blocking=True)
        label = label.to(device, non_blocking=True)
        signal_quality = signal_quality[0].to(device, non_blocking=True) if len(signal_quality) else None
        past, future = sequence[:, 0, ...], sequence[:, 1, ...]
        n_examples += past.size(0)
        totLoss, allLosses, allAcc = cpcStep(past, future, label, cpcModel, cpcCriterion, signal_quality, dp=dp)
        backward(totLoss)
        dp.reduce_and_step()
        optimizer.zero_grad()
        if allLosses.nelement() > 0:
            it += 1
            ls, ac = allLosses.detach().mean(dim=0), allAcc.mean(dim=0)
            sum_loss = ls if sum_loss is None else sum_loss + ls
            sum_acc = ac if sum_acc is None else sum_acc + ac
            if (step + 1) % loggingStep == 0:
                elapsed = time.perf_counter() - start_time
                print(f"Update {step + 1}\nelapsed: {elapsed:.1f} s")
                print(f"{1000.0 * elapsed / loggingStep:.1f} ms per batch, {1000.0 * elapsed / n_examples:.1f} ms / example")
                print("locLoss_train", (sum_loss / it).cpu().numpy())
                _check_async(device)                      # the copy above synchronised anyway
                start_time, n_examples = time.perf_counter(), 0
    if scheduler is not None:
        scheduler.step()
    _check_async(device)
    if it > 0:
        logs["locLoss_train"] = (sum_loss / it).cpu().numpy()
        logs["locAcc_train"] = (sum_acc / it).cpu().numpy()
    logs["iter"] = it
    return logs


def valStep(dataLoader, cpcModel, cpcCriterion, device=None):
    """train.py:145-187."""
    cpcCriterion.eval()
    cpcModel.eval()
    device = device or next(cpcModel.parameters()).device
    it = 0
    sum_loss = sum_acc = None
    for full_data in dataLoader:
        sequence, label, *_ = full_data
        sequence, label = sequence.to(device, non_blocking=True), label.to(device, non_blocking=True)
        past, future = sequence[:, 0, ...], sequence[:, 1, ...]
        with torch.no_grad():
            _, allLosses, allAcc = cpcStep(past, future, label, cpcModel, cpcCriterion, None)
        it += 1
        ls, ac = allLosses.mean(dim=0), allAcc.mean(dim=0)
        sum_loss = ls if sum_loss is None else sum_loss + ls
        sum_acc = ac if sum_acc is None else sum_acc + ac
    logs = {"iter": it}
    _check_async(device)
    if it > 0:
        logs["locLoss_val"] = (sum_loss / it).cpu().numpy()
        logs["locAcc_val"] = (sum_acc / it).cpu().numpy()
    return logs


def run(trainDataset, valDataset, batchSize, samplingMode, cpcModel, cpcCriterion, nEpoch, pathCheckpoint, optimizer,
        scheduler, logs, no_artefacts=False, batchSizePerGPU=None, dp=None, is_master=True):
    """The epoch loop of train.py:190-255: per epoch a fresh train loader (random offsets) and a sequential
    validation loader, trainStep, valStep, the logs dict extended key by key, and every logs["saveStep"] epochs (and
    after the last one) a checkpoint `{pathCheckpoint}_{epoch}.pt` in the reference layout plus `_logs.json`.
    Resumes from len(logs["epoch"]).  Like the reference, `best` holds the model state of the last epoch whose
    validation accuracy beat bestAcc -- which is never raised from 0 (train.py:207,237-238)."""
    import json
    from . import feature_loader as fl
    # built ONCE (its constructor broadcasts rank 0's parameters): the criterion's and the context network's gradient
    # slices are reduced under the encoder's backward, as bench.py measures it
    dp = dp or DataParallelContext.for_modules(optimizer, cpcModel, cpcCriterion)
    print(f"Running {nEpoch} epochs")
    best_acc, best_state = 0, None
    t_start = time.perf_counter()
    # Everything alive by now (modules, optimiser state, the dataset's buffers) lives as long as the run: moved out of the cyclic
    # collector's way, so that its occasional full pass does not stop the training thread for longer than its lead over the device
    # (70-165 ms about once per 15-25 thousand steps with torch's objects in the walk: profiles/r06_long_soak.txt).  Objects made
    # from here on are collected as before.
    import gc
    gc.collect()
    gc.freeze()
    for epoch in range(len(logs["epoch"]), nEpoch):
        print(f"Starting epoch {epoch}")
        train_loader = trainDataset.getDataLoader(batchSize, samplingMode, True, numWorkers=0, remove_artefacts=no_artefacts,
                                                  batch_size_per_gpu=batchSizePerGPU)
        val_loader = valDataset.getDataLoader(batchSize, "sequential", False, numWorkers=0)
        print("Training dataset %d batches, Validation dataset %d batches, batch size %d"
              % (len(train_loader), len(val_loader), batchSize))
        epoch_logs = trainStep(train_loader, cpcModel, cpcCriterion, optimizer, scheduler, logs["logging_step"], dp=dp)
        epoch_logs.update(valStep(val_loader, cpcModel, cpcCriterion))
        print(f"Ran {epoch + 1} epochs in {time.perf_counter() - t_start:.2f} seconds")
        if "locAcc_val" in epoch_logs and float(epoch_logs["locAcc_val"].mean()) > best_acc:
            best_state = {k: v.detach().clone() for k, v in fl.get_module(cpcModel).state_dict().items()}
        for key, value in epoch_logs.items():
            logs.setdefault(key, [None] * epoch).append(value.tolist() if isinstance(value, np.ndarray) else value)
        logs["epoch"].append(epoch)
        if pathCheckpoint is not None and is_master and (epoch % logs["saveStep"] == 0 or epoch == nEpoch - 1):
            fl.save_checkpoint(fl.get_module(cpcModel).state_dict(), fl.get_module(cpcCriterion).state_dict(),
                               optimizer.state_dict(), best_state, f"{pathCheckpoint}_{epoch}.pt")
            with open(pathCheckpoint + "_logs.json", "w") as fh:
                json.dump(logs, fh, indent=2)
    return logs


def init_distributed_mode(backend="nccl"):
    """distributed_mode.py:75-142 reduced to the torch.distributed.run environment: RANK, LOCAL_RANK,
    WORLD_SIZE, MASTER_ADDR/PORT; one process per GPU; backend 'nccl' is RCCL on ROCm."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
    if world > 1 and not dist.is_initialized():
        dist.init_process_group(init_method="env://", backend=backend, world_size=world, rank=rank)
    return rank, local_rank, world
