"""Drop-in counterpart of the reference's unsupervised CPC criterion.

Mirrors /root/reference/cpc/criterion/criterion.py: PredictionNetwork (:97-173, linear branch
:144-150), BaseCriterion (:176-183), NoneCriterion (:185-191), CPCUnsupersivedCriterion (:193-363)
-- same constructor/forward signatures, attribute names and state-dict keys
(`wPrediction.predictors.{k}.weight`).  The K candidate tensors of sampleClean (:237-286) are never
materialised: the fused HIP kernel gathers negatives on the fly from the index stream.

Negative indices are drawn on the HOST with the same MT19937 stream torch's CPU generator would
produce (the reference's CPU path); by default the criterion consumes -- and advances -- torch's
global CPU generator, so `torch.manual_seed(s)` gives bit-identical indices to the reference.
"""
import ctypes
import os

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from ._lib import check, f32c, grad_buffers, ptr, require_gpu, scratch, stream_ptr


# --------------------------------------------------------------------------- index sampler
def _sleep_until(event):
    """Wait for a device event WITHOUT spinning on a core: this is where the training thread stands still while it is its two steps
    ahead of the device (2.4-2.8 ms of every 4.8 ms step), and hipEventSynchronize -- blocking-sync flag or not -- burns that time
    as CPU time (bench.py: host.thread_cpu_ms_per_step 4.1 against 1.4 of work); eight ranks share one host."""
    import time
    while not event.query():
        time.sleep(0.0001)


class NegativeSampler:
    """Host MT19937 sampler (cpc_negidx_sample_host) + pinned staging ring for the H2D copy."""

    RING = int(os.environ.get("CPC_SAMPLER_RING", "4"))

    def __init__(self):
        self._lib = _lib.load()
        self._h = ctypes.c_void_p(self._lib.cpc_mt_create(5489))
        if not self._h:
            raise MemoryError("cpc_mt_create failed")
        self.follow_torch = True      # consume torch's global CPU generator (reference semantics)
        self._ring, self._dev_ring, self._ext_ring, self._events, self._slot = {}, {}, {}, {}, 0
        self._prefetched = None
        self._prefetch_state = None   # generator state in front of the draw that is in flight (see sample())
        self._torch_seen = None       # follow_torch: torch's generator state as this sampler last left it
        self._prefetch_skip = 0       # words the draw in flight dropped first (consumed by a smaller call from the draw before it)
        self._ahead_shape = None      # (n, device, shape) the draws ahead are made for: the largest call seen
        self._ahead_misses = 0        # calls in a row that were not the one drawn for
        self._last = None             # (key, slot) of the previous device-side call: its buffers' release event is recorded by the next one
        # draw the next call's words during this step, on a worker thread with a stream of its own.  On by default since round 6: a
        # draw ahead that turns out not to fit (another size, a host-side call, torch's generator used by someone else) is undone,
        # so the index sequence is the reference's either way (tests: test_prefetch_*); False = everything at call time
        self.prefetch = True
        self.alias_ring = False       # opt-in with prefetch: sample() returns one of RING persistent buffers instead of a copy of it

    def __del__(self):
        try:
            if self._h:
                self._lib.cpc_mt_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def seed(self, seed):
        """Private stream seeded like torch.manual_seed(seed); stops following the global generator."""
        check(self._lib.cpc_mt_seed(self._h, ctypes.c_uint32(int(seed) & 0xFFFFFFFF)), "mt_seed")
        self.follow_torch = False
        self._prefetched = None       # (cpc_mt_seed waited for any draw in flight; its words are discarded)

    # torch CPU generator legacy state: u64 seed, i32 left, i32 seeded, u64 next, u64 mt[624], ...
    def _pull_torch_state(self):
        st = torch.get_rng_state().numpy()
        left = int(st[8:12].view(np.int32)[0])
        nxt = int(st[16:24].view(np.uint64)[0])
        mt = np.ascontiguousarray(st[24:24 + 624 * 8].view(np.uint64).astype(np.uint32))
        check(self._lib.cpc_mt_set_state(self._h, mt.ctypes.data_as(ctypes.c_void_p), left, nxt), "mt_set_state")
        return st

    def _push_torch_state(self, st):
        mt = np.empty(624, dtype=np.uint32)
        left, nxt = ctypes.c_int(0), ctypes.c_int(0)
        check(self._lib.cpc_mt_get_state(self._h, mt.ctypes.data_as(ctypes.c_void_p), ctypes.byref(left),
                                         ctypes.byref(nxt)), "mt_get_state")
        st = st.copy()
        st[8:12].view(np.int32)[0] = left.value
        st[16:24].view(np.uint64)[0] = nxt.value
        st[24:24 + 624 * 8].view(np.uint64)[:] = mt.astype(np.uint64)
        torch.set_rng_state(torch.from_numpy(st))
        self._torch_seen = st[8:24 + 624 * 8].copy()      # what torch's generator holds after this sampler's draw

    def _torch_unchanged(self):
        """Has nobody drawn from torch's global CPU generator since this sampler left it (a draw ahead is only valid then)?"""
        if self._torch_seen is None:
            return False
        return np.array_equal(torch.get_rng_state().numpy()[8:24 + 624 * 8], self._torch_seen)

    def _cancel_prefetch(self):
        """Undo a draw ahead that will not be used: the private generator goes back to where it stood before it (with follow_torch
        the next draw starts from torch's state anyway)."""
        if self._prefetched is not None:
            check(self._lib.cpc_negidx_wait(self._h), "negidx_wait")
            mt, left, nxt = self._prefetch_state
            check(self._lib.cpc_mt_set_state(self._h, mt.ctypes.data_as(ctypes.c_void_p), left, nxt), "mt_set_state")
            if self._prefetch_skip:            # (the draw had first put the generator behind words an earlier call consumed)
                skipped = torch.empty(self._prefetch_skip, dtype=torch.int32)
                check(self._lib.cpc_mt_draw_host(self._h, ptr(skipped), self._prefetch_skip), "mt_draw_host")
            self._prefetched = None

    def sample_host(self, batch, seq_len, window, n_neg, out=None, want_parts=False, time_major=False):
        """int32 extIdx on the host (criterion.py:247-266): [batch, n_neg, window] in the reference's
        order, or the same values as [batch, window, n_neg] with time_major (the kernels' layout)."""
        n = batch * n_neg * window
        self._cancel_prefetch()
        if out is None:
            out = torch.empty(n, dtype=torch.int32)
        bidx = torch.empty(n, dtype=torch.int64) if want_parts else None
        sidx = torch.empty(n, dtype=torch.int64) if want_parts else None
        st = self._pull_torch_state() if self.follow_torch else None
        check(self._lib.cpc_negidx_sample_host(self._h, batch, seq_len, window, n_neg, int(time_major), ptr(out),
                                               ptr(bidx), ptr(sidx)), "negidx_sample_host")
        if st is not None:
            self._push_torch_state(st)
        return (out, bidx, sidx) if want_parts else out

    def _rings(self, key, n, device):
        if key not in self._ring:
            self._ring[key] = [torch.empty(2 * n, dtype=torch.int32).pin_memory() for _ in range(self.RING)]
            self._dev_ring[key] = [torch.empty(2 * n, dtype=torch.int32, device=device) for _ in range(self.RING)]
            self._ext_ring[key] = [torch.empty(n, dtype=torch.int32, device=device) for _ in range(self.RING)]
            self._events[key] = [None] * self.RING
        return self._ring[key], self._dev_ring[key], self._events[key]

    def sample(self, batch, seq_len, window, n_neg, device, time_major=True):
        """Device int32 extIdx.  time_major (the fused kernels' layout): the host only draws the raw MT19937
        words into a pinned buffer (with `prefetch`: on a worker thread, one call ahead), the device does the
        integer arithmetic (cpc_negidx_expand).  Otherwise: full host path, reference order.

        A draw ahead is made for the LARGEST call seen so far (the full batch).  The stream of words is one sequence whatever it is
        cut into -- a call of n negatives consumes its first 2 n -- so what follows a draw ahead of 2 n' words is one of:
          * the call it was made for: indices already expanded by the worker, nothing on the caller's stream but a wait;
          * a SMALLER call (the same-speaker sampler ends every speaker with a partial batch): its 2 n words are a prefix of the
            words already on the device; one expansion kernel, and the generator is put behind those 2 n words by the worker in
            front of its next draw (private stream only);
          * anything else (a larger call, a host-side call, torch's generator used in between with follow_torch): the draw is undone
            and the call draws for itself.
        Either way the index sequence is the reference's (tests: test_prefetch_*)."""
        n = batch * n_neg * window
        if not time_major:
            host = self.sample_host(batch, seq_len, window, n_neg)
            return host.to(device)
        device = torch.device(device)
        key = (n, str(device))
        ring, dev_ring, events = self._rings(key, n, device)
        # The buffers of the PREVIOUS call's slot are free again once everything enqueued since -- that call's expansion and the
        # criterion kernels that read its index tensor -- has run: marked here, one call later, on the stream those kernels are on.
        if self._last is not None and self._last[0] in self._events:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(device))
            self._events[self._last[0]][self._last[1]] = ev
        shape = (batch, seq_len, window, n_neg)
        ahead = self._prefetched                        # (key, slot, shape) of the draw in flight, or None
        if ahead is not None and self.follow_torch and not self._torch_unchanged():
            self._cancel_prefetch()                     # someone else drew from torch's generator meanwhile: ITS state decides
            ahead = None
        fixup = None                                    # (state, words consumed): where the generator has to be put before the next draw
        used = None                                     # (key, slot) whose buffers this call's kernels read
        if ahead is not None and ahead[0] == key:
            # the call the draw was made for: drawn, uploaded AND expanded (on the worker's own stream) while the GPU was busy
            # (waits for the worker's HOST part -- the draw, the enqueue of copy + expansion on its stream -- and orders the
            #  training stream behind the event recorded there: the host is not held up by the device)
            akey, aslot, ashape = ahead
            with _lib.host_wait("sampler_worker"):
                check(self._lib.cpc_negidx_wait_on(self._h, _lib.stream_ptr(device)), "negidx_wait_on")
            if ashape == shape:
                # (a COPY of the ring buffer, 3.8 MB at the benchmark shape: the worker rewrites the buffer RING - 1 calls later, and
                #  a caller may keep the indices -- saved for a backward that runs late, logged, compared -- for longer than that;
                #  `alias_ring = True` hands out the buffer itself to a caller that consumes it before the next RING - 1 samples)
                ext = self._ext_ring[akey][aslot] if self.alias_ring else self._ext_ring[akey][aslot].clone()
            else:                              # (the same number of words for another shape: the words are right, the expansion is not)
                ext = torch.empty(n, dtype=torch.int32, device=device)
                check(self._lib.cpc_negidx_expand(ptr(self._dev_ring[akey][aslot]), ptr(ext), batch, seq_len, window, n_neg, _lib.stream_ptr(device)),
                      "negidx_expand")
            used = (akey, aslot)
            self._ahead_misses = 0
            if self.follow_torch:              # torch's generator moves on by what the reference's two randint calls consume
                self._push_torch_state(torch.get_rng_state().numpy())
        elif (ahead is not None and not self.follow_torch and n < ahead[0][0] and ahead[0][1] == key[1]
              and self._prefetch_skip + 2 * n <= 2 * ahead[0][0] and self._ahead_misses < 8):
            # a smaller call: its words are the first 2 n of those already on the device
            akey, aslot, _ashape = ahead
            with _lib.host_wait("sampler_worker"):
                check(self._lib.cpc_negidx_wait_on(self._h, _lib.stream_ptr(device)), "negidx_wait_on")
            ext = torch.empty(n, dtype=torch.int32, device=device)
            check(self._lib.cpc_negidx_expand(ptr(self._dev_ring[akey][aslot]), ptr(ext), batch, seq_len, window, n_neg, _lib.stream_ptr(device)),
                  "negidx_expand")
            used = (akey, aslot)
            # (behind the words consumed since the state the draw in flight started from: a run of smaller calls adds up -- and is
            #  bounded above: past one draw's worth, or after eight calls in a row that were not the one drawn for, the draws ahead
            #  follow the calls' new size instead)
            fixup = (self._prefetch_state, self._prefetch_skip + 2 * n)
            self._ahead_misses += 1
        else:
            # words drawn ahead for a call that cannot use them: the generator goes back to where it stood before that draw, so
            # that this call consumes exactly the words the reference's two torch.randint calls would (criterion.py:247-256)
            self._cancel_prefetch()
            slot = self._slot % self.RING
            self._slot += 1
            host = ring[slot]
            if events[slot] is not None:
                with _lib.host_wait("sampler_buffer_event"):
                    _sleep_until(events[slot])     # the kernel that last read this slot's buffers has finished
            st = self._pull_torch_state() if self.follow_torch else None
            check(self._lib.cpc_mt_draw_host(self._h, ptr(host), 2 * n), "mt_draw_host")
            if st is not None:
                self._push_torch_state(st)
            raw = dev_ring[slot]
            raw.copy_(host, non_blocking=True)
            ext = torch.empty(n, dtype=torch.int32, device=device)
            check(self._lib.cpc_negidx_expand(ptr(raw), ptr(ext), batch, seq_len, window, n_neg, _lib.stream_ptr(device)),
                  "negidx_expand")
            used = (key, slot)
        self._last = used
        self._prefetched = None
        if self.prefetch:
            # draw the NEXT call's words now, on the library's worker thread, for the largest call seen so far
            if self._ahead_shape is None or self._ahead_shape[1] != str(device) or n > self._ahead_shape[0] or self._ahead_misses >= 8:
                self._ahead_shape = (n, str(device), shape)
                self._ahead_misses = 0
            an, _adev, ashape = self._ahead_shape
            akey = (an, str(device))
            aring, adev_ring, aevents = self._rings(akey, an, device)
            aslot = self._slot % self.RING
            self._slot += 1
            if aevents[aslot] is not None:
                with _lib.host_wait("sampler_buffer_event"):
                    _sleep_until(aevents[aslot])
            dev_index = device.index if device.index is not None else torch.cuda.current_device()
            ab, at, aw, ann = ashape
            if fixup is not None:
                (mt, left, nxt), consumed = fixup
                # (the state this draw will start from, for a rewind by the NEXT call, is not known on the host: it is the state
                #  behind the consumed words, which the worker produces -- a next call that cannot use the draw rewinds to the old
                #  state and draws those words again)
                self._prefetch_state = (mt, left, nxt)
                self._prefetch_skip = consumed
                check(self._lib.cpc_mt_redraw_expand_device_async(self._h, mt.ctypes.data_as(ctypes.c_void_p), left, nxt, consumed, ptr(aring[aslot]),
                                                                  ptr(adev_ring[aslot]), ptr(self._ext_ring[akey][aslot]), dev_index, ab, at, aw, ann,
                                                                  _lib.stream_ptr(device)), "mt_redraw_expand_device_async")
            else:
                # (where the generator stands before the draw ahead: a next call that cannot use it rewinds to it)
                mt = np.empty(624, dtype=np.uint32)
                left, nxt = ctypes.c_int(0), ctypes.c_int(0)
                check(self._lib.cpc_mt_get_state(self._h, mt.ctypes.data_as(ctypes.c_void_p), ctypes.byref(left), ctypes.byref(nxt)), "mt_get_state")
                self._prefetch_state = (mt, left.value, nxt.value)
                self._prefetch_skip = 0
                check(self._lib.cpc_mt_draw_expand_device_async(self._h, ptr(aring[aslot]), ptr(adev_ring[aslot]), ptr(self._ext_ring[akey][aslot]),
                                                                dev_index, ab, at, aw, ann, _lib.stream_ptr(device)), "mt_draw_expand_device_async")
            self._prefetched = (akey, aslot, ashape)
        elif fixup is not None:
            # (prefetch was switched off in between: put the generator behind the consumed words here)
            (mt, left, nxt), consumed = fixup
            check(self._lib.cpc_mt_set_state(self._h, mt.ctypes.data_as(ctypes.c_void_p), left, nxt), "mt_set_state")
            scratch_words = torch.empty(consumed, dtype=torch.int32)
            check(self._lib.cpc_mt_draw_host(self._h, ptr(scratch_words), consumed), "mt_draw_host")
        return ext


# --------------------------------------------------------------------------- fused InfoNCE
def _packed_view(tensors):
    """[K, *shape] view over K equally shaped tensors that sit back to back in memory (the predictors of a
    FlatAdam-managed criterion do), else None."""
    first = tensors[0]
    step = first.numel() * first.element_size()
    need = first.storage_offset() * first.element_size() + len(tensors) * step
    if first.untyped_storage().nbytes() < need:          # adjacent by accident, not one allocation
        return None
    for i, t in enumerate(tensors):
        if not t.is_contiguous() or t.shape != first.shape or t.data_ptr() != first.data_ptr() + i * step:
            return None
    k = len(tensors)
    return torch.as_strided(first, (k,) + tuple(first.shape), (first.numel(),) + tuple(first.stride()))


# ---- the deferred backward (cpc2_hip.h, cpc_infonce_backward_deferred) --------------------------------------------------------
# The criterion's dz and predictor weight gradients are produced on a stream of the library's while the context network's
# backward runs; whoever consumes them has to sit behind join_deferred().  Nothing but the caller can promise that, so the
# deferral is an EXPLICIT opt-in of the caller's, never something a tensor attribute switches on by travelling through
# wrappers: `with criterion.deferred_backward(encoded_full): criterion(c, z, label)` (cpcStep does it) states that
#   * `encoded_full` came out of grad_join() (CPCModel.forward applies it BEFORE the context network, so autograd runs its
#     backward after the context network's and before anything that reads the summed gradient of the encoder output),
#   * the criterion call inside the scope is the ONLY consumer of that tensor (a second loss term on it, or a tensor hook,
#     would make autograd read dz before the join), and
#   * nothing reads the predictors' weight gradients before the backward pass has ended.
# The scope is refused (the backward then runs at once, on the caller's stream) when the criterion object is not the bare
# module (DistributedDataParallel / DataParallel around it: DDP's reducer copies a weight gradient into its bucket the moment
# autograd accumulates it, i.e. before the side stream has written it) or when a predictor weight carries a backward or
# post-accumulate hook.  A callback at the end of the backward pass joins whatever is still pending (an encoder without
# gradient: nobody reads dz, the optimiser reads the weight gradients).
_deferred = {}          # device index -> tensors the side stream still reads / writes (kept alive until the join)


def join_deferred(device):
    """Make the current stream of `device` wait for a pending deferred criterion backward (no-op when none is)."""
    device = torch.device(device)
    if device.type != "cuda":
        return
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if _deferred.pop(idx, None) is not None:
        check(_lib.load().cpc_infonce_join(stream_ptr(device)), "infonce_join")


class _GradJoin(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        join_deferred(g.device)
        return g


def grad_join(encoded):
    """Identity on the encoder output; its backward is where a deferred criterion backward is joined.  Returns the tensor the
    criterion should be given (marked `_cpc_join`; `_cpc_join_source` = the tensor whose gradient is the complete one)."""
    if not (encoded.is_cuda and encoded.requires_grad and torch.is_grad_enabled()):
        return encoded
    out = _GradJoin.apply(encoded)
    out._cpc_join = True
    out._cpc_join_source = encoded
    return out


def carry_join(view, parent, start):
    """`view` = parent[start:start + n] (windows) of a grad_join() output.  The criterion then differentiates with respect to
    `parent` itself (its dz lands in rows start.. of a gradient of parent's size), so that no slicing node of autograd's --
    which would read dz BEFORE the context network's backward, i.e. before the join -- sits between the two."""
    if getattr(parent, "_cpc_join", False) and parent.is_contiguous() and view.is_contiguous() and \
            view.data_ptr() == parent.data_ptr() + start * parent.stride(0) * parent.element_size():
        view._cpc_join = True
        view._cpc_join_parent = (parent, int(start))
    return view


# ---- gradients that are zero by construction, without materialising the zeros every step ----------------------------------------
# train.py:102-103 slices the model's outputs (context of the first b windows, targets of the last b): autograd turns each slice's
# gradient back into the full tensor with a zero fill plus a copy (16.8 MB + 8.4 MB per slice at the benchmark shape, two small
# launches each).  The halves that are zero are zero EVERY step, so they live in buffers that are allocated and zeroed once per
# (device, shape, slot): the criterion's backward writes dc / dz straight into the other half and hands the whole buffer on.
# Nothing ever writes the zero half (autograd only reads gradients it does not own: this cache holds a reference), and a buffer is
# reused only after the backward pass that read it has been enqueued on the same stream.
from ._tail import _cached_grad_buffer, grad_home, grad_home_view  # noqa: E402


class _FirstWindows(torch.autograd.Function):
    """x[:n] (windows) whose backward does not fill and copy: the gradient of the slice is written by its producer straight into
    the first n windows of a cached buffer whose other windows stay zero (see above); any other gradient is copied into it."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.full_shape, ctx.n = tuple(x.shape), n
        out = x[:n]
        return out.view_as(out)

    @staticmethod
    def backward(ctx, g):
        buf = _cached_grad_buffer(ctx.full_shape, g.device, ("context", ctx.n))
        if not (g.data_ptr() == buf.data_ptr() and g.is_contiguous() and g.dtype == torch.float32):
            buf[:ctx.n].copy_(g)
        return buf, None


def first_windows(x, n):
    """x[:n] for the context features in cpcStep (train.py:102); marks the result so that the criterion's backward writes its
    gradient where _FirstWindows.backward expects it."""
    if not (x.is_cuda and x.requires_grad and torch.is_grad_enabled() and x.is_contiguous() and x.dtype == torch.float32):
        return x[:n]
    out = _FirstWindows.apply(x, n)
    out._cpc_first_of = (tuple(x.shape), n)
    return out


class _SplitWindows(torch.autograd.Function):
    """(x[:n], x[n:]) along the window axis whose backward neither fills nor adds: both gradients are written by their producers
    straight into the two parts of ONE cached buffer (grad_home), which is handed on whole.  The deferred criterion backward is
    joined here -- autograd runs this node when BOTH gradients are due, i.e. after the context network's backward."""

    @staticmethod
    def forward(ctx, x, n, tag):
        ctx.full_shape, ctx.n, ctx.tag = tuple(x.shape), n, tag
        ctx.set_materialize_grads(False)
        first, rest = x[:n], x[n:]
        return first.view_as(first), rest.view_as(rest)

    @staticmethod
    def backward(ctx, g_first, g_rest):
        n = ctx.n
        device = (g_first if g_first is not None else g_rest).device
        join_deferred(device)
        buf = _cached_grad_buffer(ctx.full_shape, device, (ctx.tag, n))
        row = buf.stride(0) * buf.element_size()
        for g, part, at in ((g_first, buf[:n], 0), (g_rest, buf[n:], n)):
            if g is None:
                part.zero_()
            elif not (g.data_ptr() == buf.data_ptr() + at * row and g.is_contiguous() and g.dtype == torch.float32):
                part.copy_(g)
        return buf, None, None


def split_windows(x, n, tag="split"):
    """cpcStep's split of the encoder output [2b, T, H] into the context network's windows x[:n] and the criterion's target windows
    x[n:] (train.py:99-103 keeps exactly these: the context of the first half, the encoded data of the second).  Each part is marked
    with its home in the gradient buffer (`_cpc_grad_home`, honoured by the recurrent / transformer / criterion backward) and the
    second with `_cpc_join`: whoever consumes the gradient of `x` sits behind the join of a deferred criterion backward."""
    if not (x.is_cuda and x.requires_grad and torch.is_grad_enabled() and x.is_contiguous() and x.dtype == torch.float32):
        return x[:n], x[n:]
    first, rest = _SplitWindows.apply(x, n, tag)
    first._cpc_grad_home = (tuple(x.shape), (tag, n), 0)
    rest._cpc_grad_home = (tuple(x.shape), (tag, n), n)
    rest._cpc_join = True
    return first, rest


class _FirstFrames(torch.autograd.Function):
    """x[:, :w] (the frames the criterion uses) as a contiguous tensor; the gradient goes back into x's home in the gradient buffer
    (grad_home), whose frames w.. nobody writes: they stay zero from the buffer's allocation on (its slot is this form's own)."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.home, ctx.shape, ctx.w = grad_home(x), tuple(x.shape), w
        return x[:, :w].contiguous()

    @staticmethod
    def backward(ctx, g):
        if ctx.home is not None:
            full_shape, slot, start = ctx.home
            buf = _cached_grad_buffer(full_shape, g.device, slot)[start:start + ctx.shape[0]]
        else:
            buf = torch.zeros(ctx.shape, dtype=torch.float32, device=g.device)
        buf[:, :ctx.w].copy_(g)
        return buf, None


def first_frames(x, w, total):
    """x[:, :w] of a `total`-frame sequence for a causal context network whose later frames nobody uses (criterion.py:296 keeps
    cFeature[:, :windowSize]); the result is marked `_cpc_frames_of = total` so that the criterion knows it was handed the slice."""
    if x.is_cuda and x.requires_grad and torch.is_grad_enabled() and x.dtype == torch.float32 and x.is_contiguous():
        out = _FirstFrames.apply(x, w)
    else:
        out = x[:, :w].contiguous()
    out._cpc_frames_of = total
    return out


class _DeferScope:
    def __init__(self, criterion, encoded_full):
        self.criterion, self.tensor = criterion, encoded_full

    def __enter__(self):
        ok = getattr(self.tensor, "_cpc_join", False) and not hasattr(self.tensor, "_cpc_join_parent")
        self.criterion._defer_scope = self.tensor if ok else None
        return self

    def __exit__(self, *exc):
        self.criterion._defer_scope = None
        return False


class _InfoNCEFn(torch.autograd.Function):
    """inputs: c, z, ext_idx, weights, n_neg, defer, then the K predictor weights [dim_enc, dim_ar].  defer: None, or
    (start, n): the criterion's targets are windows start .. start + n of z and the backward is the deferred one."""

    @staticmethod
    def forward(ctx, c, z, ext_idx, weights, n_neg, defer, *wk):
        require_gpu(c, z, ext_idx, *wk)
        lib = _lib.load()
        ctx.c_first_of = getattr(c, "_cpc_first_of", None) if (c.is_contiguous() and c.dtype == torch.float32) else None
        c = f32c(c)
        ctx.z_full_shape = None
        ctx.z_home = grad_home(z)              # (split_windows: dz is written where the split's backward looks for it)
        if defer is not None:
            ctx.z_full_shape, ctx.z_start = tuple(z.shape), defer[0]
            z = z[defer[0]:defer[0] + defer[1]]
        z = f32c(z)
        wpred = _packed_view([w.detach() for w in wk])
        if wpred is None:
            wpred = torch.stack([f32c(w.detach()) for w in wk], dim=0)
        b, tc, dim_ar = c.shape
        k, dim_enc, _ = wpred.shape
        t = z.shape[1]
        # c holds the t frames of the sequence, or only the W = t - k the criterion uses (criterion.py:296: cFeature[:, :windowSize])
        if z.shape != (b, t, dim_enc) or wpred.shape[2] != dim_ar or tc not in (t, t - k):
            raise ValueError(f"shape mismatch c={tuple(c.shape)} z={tuple(z.shape)} W={tuple(wpred.shape)}")
        ctx.cw = tc != t
        if ext_idx.dtype != torch.int32 or ext_idx.numel() != b * n_neg * (t - k):
            raise ValueError("ext_idx must be int32 [b, W, n_neg]")
        w = f32c(weights) if weights is not None else None
        nsaved = lib.cpc_infonce_saved_bytes(b, t, k, dim_ar, dim_enc, n_neg)
        nscratch = lib.cpc_infonce_scratch_bytes(b, t, k, dim_ar, dim_enc, n_neg)
        if nsaved == 0:
            check(-1, "infonce shape query")
        losses = torch.empty(k, dtype=torch.float32, device=c.device)
        acc = torch.empty(k, dtype=torch.float32, device=c.device)
        saved = torch.empty(nsaved, dtype=torch.uint8, device=c.device)
        sc = scratch(nscratch, c.device)
        fwd = lib.cpc_infonce_forward_cw if ctx.cw else lib.cpc_infonce_forward
        check(fwd(ptr(c), ptr(z), ptr(wpred), ptr(ext_idx), ptr(w), ptr(losses), ptr(acc),
                  ptr(saved), ptr(sc), b, t, k, dim_ar, dim_enc, n_neg, stream_ptr(c.device)), "infonce_forward")
        ctx.save_for_backward(c, z, wpred, ext_idx, w, saved)
        ctx.param_refs = wk
        ctx.dims = (b, t, k, dim_ar, dim_enc, n_neg)
        ctx.mark_non_differentiable(acc)
        ctx.set_materialize_grads(False)       # (no zeros tensor -- a launch -- for the gradient of `acc`, which nobody has)
        return losses, acc

    @staticmethod
    def backward(ctx, dlosses, _dacc):
        lib = _lib.load()
        c, z, wpred, ext_idx, w, saved = ctx.saved_tensors
        b, t, k, dim_ar, dim_enc, n_neg = ctx.dims
        if dlosses is None:                    # (materialize_grads is off: a loss nobody differentiated)
            dlosses = torch.zeros(k, dtype=torch.float32, device=c.device)
        dlosses = f32c(dlosses)
        if ctx.c_first_of is not None:       # (first_windows: dc goes where the slice's backward looks for it, the rest stays zero)
            dc = _cached_grad_buffer(ctx.c_first_of[0], c.device, ("context", ctx.c_first_of[1]))[:c.shape[0]]
        else:
            dc = torch.empty_like(c)
        defer = ctx.z_full_shape is not None
        if defer and ctx.z_full_shape != tuple(z.shape):
            # (windows outside start .. start + n get no gradient from the criterion: zero once, see _cached_grad_buffer)
            dz_out = _cached_grad_buffer(ctx.z_full_shape, z.device, ("targets", ctx.z_start, z.shape[0]))
            dz = dz_out[ctx.z_start:ctx.z_start + z.shape[0]]
        else:
            dz_out = dz = grad_home_view(ctx.z_home, z)
        gw = grad_buffers(ctx.param_refs)
        dw = _packed_view(gw)                 # contiguous in the flat gradient buffer -> written in place
        direct = dw is not None
        if not direct:
            dw = torch.empty_like(wpred)
        nscratch = lib.cpc_infonce_scratch_bytes(b, t, k, dim_ar, dim_enc, n_neg)
        # (the weight gradients may only be late when nothing reads them before the end of the backward pass: written in place
        #  into the flat gradient buffer.  A private buffer is added to .grad by autograd as soon as this function returns.)
        if defer and direct:
            join_deferred(c.device)            # (one pending backward per device)
            sc = scratch(nscratch, c.device, tag="infonce_deferred")
            if ctx.cw:
                check(lib.cpc_infonce_backward_cw(ptr(c), ptr(z), ptr(wpred), ptr(ext_idx), ptr(w), ptr(dlosses), ptr(saved),
                                                  ptr(sc), ptr(dc), ptr(dz), ptr(dw), b, t, k, dim_ar, dim_enc, n_neg, 1,
                                                  stream_ptr(c.device)), "infonce_backward_cw")
            else:
                check(lib.cpc_infonce_backward_deferred(ptr(c), ptr(z), ptr(wpred), ptr(ext_idx), ptr(w), ptr(dlosses), ptr(saved),
                                                        ptr(sc), ptr(dc), ptr(dz), ptr(dw), b, t, k, dim_ar, dim_enc, n_neg,
                                                        stream_ptr(c.device)), "infonce_backward_deferred")
            idx = c.device.index if c.device.index is not None else torch.cuda.current_device()
            _deferred[idx] = (c, z, wpred, ext_idx, w, dlosses, saved, sc, dz_out, dw)
            device = c.device
            torch.autograd.Variable._execution_engine.queue_callback(lambda: join_deferred(device))
        else:
            sc = scratch(nscratch, c.device)
            if ctx.cw:
                check(lib.cpc_infonce_backward_cw(ptr(c), ptr(z), ptr(wpred), ptr(ext_idx), ptr(w), ptr(dlosses), ptr(saved),
                                                  ptr(sc), ptr(dc), ptr(dz), ptr(dw), b, t, k, dim_ar, dim_enc, n_neg, 0,
                                                  stream_ptr(c.device)), "infonce_backward_cw")
            else:
                check(lib.cpc_infonce_backward(ptr(c), ptr(z), ptr(wpred), ptr(ext_idx), ptr(w), ptr(dlosses), ptr(saved),
                                               ptr(sc), ptr(dc), ptr(dz), ptr(dw), b, t, k, dim_ar, dim_enc, n_neg,
                                               stream_ptr(c.device)), "infonce_backward")
        if not direct:
            gw = list(dw.unbind(0))
        return (dc, dz_out, None, None, None, None) + tuple(gw)


class _InfoNCEPredFn(torch.autograd.Function):
    """Same criterion, predictions supplied by predictor modules: inputs z, ext_idx, weights, n_neg, then the K
    prediction tensors [b, W, dim_enc]."""

    @staticmethod
    def forward(ctx, z, ext_idx, weights, n_neg, *preds):
        require_gpu(z, ext_idx, *preds)
        lib = _lib.load()
        z = f32c(z)
        preds = tuple(f32c(p) for p in preds)
        b, t, dim_enc = z.shape
        k = len(preds)
        if any(p.shape != (b, t - k, dim_enc) for p in preds):
            raise ValueError(f"predictions must be [b, W, dim_enc] = {(b, t - k, dim_enc)}")
        if ext_idx.dtype != torch.int32 or ext_idx.numel() != b * n_neg * (t - k):
            raise ValueError("ext_idx must be int32 [b, W, n_neg]")
        w = f32c(weights) if weights is not None else None
        nsaved = lib.cpc_infonce_saved_bytes(b, t, k, dim_enc, dim_enc, n_neg)
        nscratch = lib.cpc_infonce_scratch_bytes(b, t, k, dim_enc, dim_enc, n_neg)
        if nsaved == 0:
            check(-1, "infonce shape query")
        losses = torch.empty(k, dtype=torch.float32, device=z.device)
        acc = torch.empty(k, dtype=torch.float32, device=z.device)
        saved = torch.empty(nsaved, dtype=torch.uint8, device=z.device)
        sc = scratch(nscratch, z.device)
        check(lib.cpc_infonce_forward_pred(_lib.ptr_array(preds), ptr(z), ptr(ext_idx), ptr(w), ptr(losses), ptr(acc),
                                           ptr(saved), ptr(sc), b, t, k, dim_enc, n_neg, stream_ptr(z.device)),
              "infonce_forward_pred")
        ctx.save_for_backward(z, ext_idx, w, saved, *preds)
        ctx.dims = (b, t, k, dim_enc, n_neg)
        ctx.mark_non_differentiable(acc)
        ctx.set_materialize_grads(False)
        return losses, acc

    @staticmethod
    def backward(ctx, dlosses, _dacc):
        lib = _lib.load()
        z, ext_idx, w, saved, *preds = ctx.saved_tensors
        b, t, k, dim_enc, n_neg = ctx.dims
        if dlosses is None:
            dlosses = torch.zeros(k, dtype=torch.float32, device=z.device)
        dlosses = f32c(dlosses)
        dz = torch.empty_like(z)
        dpreds = [torch.empty_like(p) for p in preds]
        sc = scratch(lib.cpc_infonce_scratch_bytes(b, t, k, dim_enc, dim_enc, n_neg), z.device)
        check(lib.cpc_infonce_backward_pred(_lib.ptr_array(preds), ptr(z), ptr(ext_idx), ptr(w), ptr(dlosses), ptr(saved),
                                            ptr(sc), _lib.ptr_array(dpreds), ptr(dz), b, t, k, dim_enc, n_neg,
                                            stream_ptr(z.device)), "infonce_backward_pred")
        return (dz, None, None, None) + tuple(dpreds)


class _LinearFn(torch.autograd.Function):
    """y = x W^T on the library's GEMMs (x [..., K], W [N, K]): the predictions of the linear predictors as tensors, for the
    paths that need them outside the fused kernel (predictor dropout)."""

    @staticmethod
    def forward(ctx, x, weight):
        require_gpu(x, weight)
        lib = _lib.load()
        x2 = f32c(x).reshape(-1, x.shape[-1])
        w = f32c(weight.detach())
        m, kdim = x2.shape
        n = w.shape[0]
        y = torch.empty(m, n, dtype=torch.float32, device=x.device)
        check(lib.cpc_gemm_nt(ptr(x2), kdim, ptr(w), kdim, ptr(y), n, None, m, n, kdim, stream_ptr(x.device)), "gemm_nt")
        ctx.save_for_backward(x2, w)
        ctx.xshape = x.shape
        return y.view(*x.shape[:-1], n)

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x2, w = ctx.saved_tensors
        m, kdim = x2.shape
        n = w.shape[0]
        dy2 = f32c(dy).reshape(m, n)
        wt = w.t().contiguous()                                     # [K, N]: the NT form's B operand of dx = dy W
        dx = torch.empty(m, kdim, dtype=torch.float32, device=dy.device)
        check(lib.cpc_gemm_nt(ptr(dy2), n, ptr(wt), n, ptr(dx), kdim, None, m, kdim, n, stream_ptr(dy.device)), "gemm_nt")
        dw = torch.empty(n, kdim, dtype=torch.float32, device=dy.device)
        nb = lib.cpc_gemm_tn_scratch_bytes(n, kdim, m)
        sc = scratch(nb, dy.device)
        check(lib.cpc_gemm_tn(ptr(dy2), n, ptr(x2), kdim, ptr(dw), kdim, n, kdim, m, ptr(sc), nb, stream_ptr(dy.device)), "gemm_tn")
        return dx.view(ctx.xshape), dw


# --------------------------------------------------------------------------- modules
class PredictionNetwork(nn.Module):
    """criterion.py:97-173: K predictors under `predictors` (same keys / init as the reference):
    linear (the `else` branch :144-150, nn.Linear(dimOutputAR, dimOutputEncoder, bias=False)), one-layer
    transformers (rnnMode='transformer' :136-143, the fork's default; keys `predictors.{k}.0.*`), or the recurrent
    ones (rnnMode='LSTM' / 'RNN' :115-123)."""

    def __init__(self, nPredicts, dimOutputAR, dimOutputEncoder, rnnMode=None, dropout=False,
                 sizeInputSeq=116, transformer_pruning=0):
        super(PredictionNetwork, self).__init__()
        if rnnMode in ("ffd", "conv4", "conv8", "conv12"):
            raise NotImplementedError(
                f"rnnMode={rnnMode!r}: only 'linear', 'transformer', 'LSTM' and 'RNN' predictors have an MI355X kernel path")
        self.predictors = nn.ModuleList()
        self.RESIDUAL_STD = 0.01
        self.dimOutputAR = dimOutputAR
        # criterion.py:113,168-169: nn.Dropout(0.5) on every prediction in training mode.  The mask comes from torch's
        # generator of the prediction's device (as in the reference's own GPU run: not reproducible against a CPU run)
        self.dropout = nn.Dropout(p=0.5) if dropout else None
        self.rnnMode = rnnMode
        if rnnMode == 'transformer':
            from .transformers import buildTransformerAR
            for _ in range(nPredicts):
                self.predictors.append(buildTransformerAR(dimOutputEncoder, dimOutputAR, nLayers=1,
                                                          sizeSeq=sizeInputSeq, abspos=False))
            return
        if rnnMode in ('LSTM', 'RNN'):                   # criterion.py:115-123
            from .model import LSTMPredictor, RNNPredictor
            for _ in range(nPredicts):
                self.predictors.append(LSTMPredictor(dimOutputAR, dimOutputEncoder, batch_first=True) if rnnMode == 'LSTM'
                                       else RNNPredictor(dimOutputAR, dimOutputEncoder))
            return
        for _ in range(nPredicts):
            self.predictors.append(nn.Linear(dimOutputAR, dimOutputEncoder, bias=False))
            if dimOutputEncoder > dimOutputAR:
                residual = dimOutputEncoder - dimOutputAR
                self.predictors[-1].weight.data.copy_(torch.cat([
                    torch.randn(dimOutputAR, dimOutputAR),
                    self.RESIDUAL_STD * torch.randn(residual, dimOutputAR)], dim=0))

    def packed_weight(self):
        """[K, dimOutputEncoder, dimOutputAR]; autograd splits the gradient back per predictor."""
        return torch.stack([p.weight for p in self.predictors], dim=0)


class MultiHeadPredictionNetwork(nn.Module):
    """criterion.py:44-94 with rnnMode='transformer' (--multihead_rnn): ONE transformer whose feed-forward net emits
    nPredicts residual branches (`predictor.0.*`, a MultiClassifierTransformerHead); prediction k is
    predictor(c)[:, :, k]."""

    def __init__(self, nPredicts, dimOutputAR, dimOutputEncoder, rnnMode='transformer_multi', dropout=False,
                 sizeInputSeq=116, transformer_pruning=0):
        super(MultiHeadPredictionNetwork, self).__init__()
        if rnnMode != 'transformer':
            if rnnMode == 'transformer_adaptive_span':
                raise NotImplementedError("rnnMode='transformer_adaptive_span' is not on the MI355X hot path")
            raise ValueError(f"unknown mode {rnnMode}")
        from .transformers import buildMultHeadTransformerAR
        self.dimOutputAR = dimOutputAR
        self.dropout = nn.Dropout(p=0.5) if dropout else None      # criterion.py:63,86-87
        self.nPredicts = nPredicts
        self.rnnMode = 'transformer_multi'
        self.predictor = buildMultHeadTransformerAR(dimOutputEncoder, dimOutputAR, nLayers=1, sizeSeq=sizeInputSeq,
                                                    abspos=False, nHeads=nPredicts)


class BaseCriterion(nn.Module):

    def warmUp(self):
        return False

    def update(self):
        return


class NoneCriterion(BaseCriterion):
    def __init__(self):
        super(NoneCriterion, self).__init__()

    def forward(self, cFeature, encodedData, label):
        return torch.zeros(1, 1, device=cFeature.device), torch.zeros(1, 1, device=cFeature.device)


class CPCUnsupersivedCriterion(BaseCriterion):
    """criterion.py:193-363."""

    def __init__(self,
                 nPredicts,             # Number of steps
                 dimOutputAR,           # Dimension of G_ar
                 dimOutputEncoder,      # Dimension of the convolutional net
                 negativeSamplingExt,   # Number of negative samples to draw
                 mode=None,
                 rnnMode=False,
                 dropout=False,
                 nSpeakers=0,
                 sizeInputSeq=116,
                 multihead_rnn=False,
                 transformer_pruning=0,
                 n_skipped=0,
                 growth_rate=None,
                 inflection_point_x=None):
        super(CPCUnsupersivedCriterion, self).__init__()
        if multihead_rnn:                                # criterion.py:213-218
            self.wPrediction = MultiHeadPredictionNetwork(nPredicts, dimOutputAR, dimOutputEncoder, rnnMode=rnnMode,
                                                          dropout=dropout, sizeInputSeq=sizeInputSeq - nPredicts,
                                                          transformer_pruning=transformer_pruning)
        else:
            self.wPrediction = PredictionNetwork(nPredicts, dimOutputAR, dimOutputEncoder, rnnMode=rnnMode,
                                                 dropout=dropout, sizeInputSeq=sizeInputSeq - nPredicts)
        self.nSkipped = n_skipped
        self.nPredicts = nPredicts
        self.negativeSamplingExt = negativeSamplingExt
        self.growth_rate = growth_rate
        self.inflection_point_x = inflection_point_x
        self.weighting_function = lambda x: 0.00001 + 1 / (1 + torch.exp(-growth_rate * (x - inflection_point_x)))
        if mode not in [None, "reverse"]:
            raise ValueError("Invalid mode")
        self.mode = mode
        self.sampler = NegativeSampler()
        self._defer_scope = None       # set by deferred_backward() for the duration of the caller's scope

    def seed(self, seed):
        """Use a private negative-index stream (e.g. one per data-parallel rank)."""
        self.sampler.seed(seed)

    def _may_defer(self):
        """No predictor weight with a hook that would read its gradient before the end of the backward pass."""
        for p in self.wPrediction.parameters():
            if getattr(p, "_backward_hooks", None) or getattr(p, "_post_accumulate_grad_hooks", None):
                return False
        return True

    def deferred_backward(self, encoded_full):
        """Context manager: the criterion call(s) inside may run the deferred backward with respect to `encoded_full`, a
        grad_join() output that nothing else consumes (see the comment above join_deferred)."""
        return _DeferScope(self, encoded_full)

    def _prepare(self, cFeature, encodedData):
        if self.mode == "reverse":                      # criterion.py:292-294
            encodedData = torch.flip(encodedData, [1])
            cFeature = torch.flip(cFeature, [1])
        return cFeature, encodedData

    def sampleIndices(self, batchSize, nNegativeExt, windowSize, device, time_major=True):
        """Device int32 extIdx (criterion.py:247-266): [batchSize, windowSize, negativeSamplingExt] when
        time_major (the kernels' layout), else the reference's [batchSize, negativeSamplingExt, windowSize]."""
        return self.sampler.sample(batchSize, nNegativeExt, windowSize, self.negativeSamplingExt, device,
                                   time_major=time_major)

    def _predictions(self, cW):
        """The K prediction tensors [b, W, dim_enc] of the predictor modules on cW = c[:, :W] (criterion.py:161-169), or
        None when the predictors are linear and no dropout is active: the fused kernel then computes them itself."""
        net = self.wPrediction
        drop = net.dropout if (net.dropout is not None and self.training) else None
        if net.rnnMode == 'transformer_multi':
            preds = list(torch.unbind(net.predictor(cW), dim=2))            # criterion.py:85 prediction[:, :, k]
        elif net.rnnMode in ('transformer', 'LSTM', 'RNN'):
            preds = [predictor(cW) for predictor in net.predictors]
            preds = [p[0] if isinstance(p, tuple) else p for p in preds]   # criterion.py:164-165
        elif drop is not None:
            preds = [_LinearFn.apply(cW, predictor.weight) for predictor in net.predictors]
        else:
            return None
        if drop is not None:
            preds = [drop(p) for p in preds]                                # criterion.py:168-169
        return preds

    def sampleClean(self, encodedData, windowSize):
        """criterion.py:237-286: the K candidate tensors [b, 1 + negativeSamplingExt, windowSize, dimEncoded] (positive first)
        and the all-zero labels.  The negative indices are the reference's, bit for bit (same generator stream, same
        order); the rows are gathered on the device.  The training step never calls this -- its kernel gathers on the
        fly -- it is the reference's inspection API."""
        batchSize, nNegativeExt, dimEncoded = encodedData.size()
        extIdx = self.sampler.sample(batchSize, nNegativeExt, windowSize, self.negativeSamplingExt, encodedData.device,
                                     time_major=False).long()
        rows = encodedData.contiguous().view(-1, dimEncoded)
        negExt = rows.index_select(0, extIdx).view(batchSize, self.negativeSamplingExt, windowSize, dimEncoded)
        labelLoss = torch.zeros(batchSize * windowSize, dtype=torch.long, device=encodedData.device)
        outputs = []
        for k in range(1, self.nPredicts + 1):
            posSeq = encodedData[:, k:k + windowSize].reshape(batchSize, 1, windowSize, dimEncoded)
            outputs.append(torch.cat((posSeq, negExt), dim=1))
        return outputs, labelLoss

    def _logits(self, cFeature, encodedData, extIdx, n_neg):
        """float [b, W, K, 1 + n_neg] straight from the fused forward kernel (no loss, no graph)."""
        lib = _lib.load()
        with torch.no_grad():
            windowSize = cFeature.size(1) - self.nPredicts
            preds = self._predictions(cFeature[:, :windowSize].contiguous())
            z = f32c(encodedData)
            b, t, dim_enc = z.shape
            k = self.nPredicts
            losses = torch.empty(k, dtype=torch.float32, device=z.device)
            acc = torch.empty(k, dtype=torch.float32, device=z.device)
            dim_ar = dim_enc if preds is not None else cFeature.size(2)
            nsaved = lib.cpc_infonce_saved_bytes(b, t, k, dim_ar, dim_enc, n_neg)
            if nsaved == 0:
                check(-1, "infonce shape query")
            saved = torch.empty(nsaved, dtype=torch.uint8, device=z.device)
            sc = scratch(lib.cpc_infonce_scratch_bytes(b, t, k, dim_ar, dim_enc, n_neg), z.device)
            if preds is None:
                c = f32c(cFeature)
                wpred = torch.stack([f32c(p.weight.detach()) for p in self.wPrediction.predictors], dim=0)
                check(lib.cpc_infonce_forward(ptr(c), ptr(z), ptr(wpred), ptr(extIdx), None, ptr(losses), ptr(acc), ptr(saved),
                                              ptr(sc), b, t, k, dim_ar, dim_enc, n_neg, stream_ptr(z.device)), "infonce_forward")
            else:
                preds = [f32c(p) for p in preds]
                check(lib.cpc_infonce_forward_pred(_lib.ptr_array(preds), ptr(z), ptr(extIdx), None, ptr(losses), ptr(acc),
                                                   ptr(saved), ptr(sc), b, t, k, dim_enc, n_neg, stream_ptr(z.device)),
                      "infonce_forward_pred")
            off = lib.cpc_infonce_logits_offset(b, t, k, dim_ar, dim_enc, n_neg)
            n = b * windowSize * k * (n_neg + 1)
            slot_order = saved[off:off + 4 * n].view(torch.float32).view(b, windowSize, k, n_neg + 1)
            # the kernel leaves a (b, t)'s negatives in the order it visited them (sorted by z-row block): back to the caller's
            poff = lib.cpc_infonce_perm_offset(b, t, k, dim_ar, dim_enc, n_neg)
            perm = saved[poff:poff + 2 * b * windowSize * n_neg].view(torch.int16).view(b, windowSize, 1, n_neg).long() & 0xFFFF
            out = torch.empty_like(slot_order)
            out[..., :1] = slot_order[..., :1]
            out[..., 1:].scatter_(3, perm.expand(b, windowSize, k, n_neg), slot_order[..., 1:])
            return out

    def getPrediction(self, cFeature, encodedData, label):
        """criterion.py:291-302: the K score tensors [b, 1 + negativeSamplingExt, W] (candidate 0 = the positive; score =
        mean over channels of prediction * candidate) and the all-zero labels, computed by the fused forward kernel --
        the candidate tensors are not built.  Detached: training goes through forward()."""
        require_gpu(cFeature, encodedData)
        cFeature, encodedData = self._prepare(cFeature, encodedData)
        batchSize, seqSize, _ = cFeature.size()
        windowSize = seqSize - self.nPredicts
        extIdx = self.sampleIndices(batchSize, seqSize, windowSize, cFeature.device)
        logits = self._logits(cFeature, encodedData, extIdx, self.negativeSamplingExt)
        labelLoss = torch.zeros(batchSize * windowSize, dtype=torch.long, device=cFeature.device)
        return [logits[:, :, k].transpose(1, 2).contiguous() for k in range(self.nPredicts)], labelLoss

    def getCosineDistances(self, cFeature, encodedData):
        """criterion.py:304-327: the positive candidate's score alone, K tensors [b, 1, W].  No negative is drawn (the
        generator stream is left untouched): the kernel runs with one dummy negative per position, whose score is dropped."""
        require_gpu(cFeature, encodedData)
        cFeature, encodedData = self._prepare(cFeature, encodedData)
        batchSize, seqSize, _ = cFeature.size()
        windowSize = seqSize - self.nPredicts
        dummy = torch.zeros(batchSize * windowSize, dtype=torch.int32, device=cFeature.device)
        logits = self._logits(cFeature, encodedData, dummy, 1)
        return [logits[:, :, k, :1].transpose(1, 2).contiguous() for k in range(self.nPredicts)]

    def forward(self, cFeature, encodedData, label, signal_quality=None):
        batchSize, seqSize, _ = cFeature.size()
        windowSize = seqSize - self.nPredicts
        if getattr(cFeature, "_cpc_frames_of", None) == encodedData.size(1) and seqSize == encodedData.size(1) - self.nPredicts \
                and self.mode != "reverse":
            # the caller (cpcStep) handed over cFeature[:, :windowSize] of a causal context network -- what :296 slices out itself
            seqSize, windowSize = encodedData.size(1), seqSize
        # deferred backward: only inside an explicit deferred_backward() scope of the caller's whose tensor this encodedData is
        # (or is a window slice of: carry_join), and not in reverse mode (the flip is an autograd node of its own that would
        # read dz at once)
        defer = None
        scope = self._defer_scope
        if scope is not None and self.mode != "reverse" and not os.environ.get("CPC_NCE_NO_DEFER") \
                and encodedData.dtype == torch.float32 and encodedData.is_contiguous() and getattr(encodedData, "_cpc_join", False):
            zFull, start = getattr(encodedData, "_cpc_join_parent", (encodedData, 0))
            if zFull is scope and self._may_defer():
                defer = (start, batchSize)
        cFeature, encodedData = self._prepare(cFeature, encodedData)
        if signal_quality is not None:                  # criterion.py:334-338
            quality_weighting = self.weighting_function(signal_quality.mean(dim=1))
            quality_weighting = quality_weighting.unsqueeze(1).repeat(1, windowSize).contiguous().view(-1).float()
        else:
            quality_weighting = None                    # ones (criterion.py:340)
        extIdx = self.sampleIndices(batchSize, seqSize, windowSize, cFeature.device)
        preds = self._predictions(cFeature[:, :windowSize].contiguous()) if self._needs_modules() else None
        if preds is not None:
            losses, acc = _InfoNCEPredFn.apply(encodedData, extIdx, quality_weighting, self.negativeSamplingExt, *preds)
        else:
            losses, acc = _InfoNCEFn.apply(cFeature, zFull if defer is not None else encodedData, extIdx, quality_weighting,
                                           self.negativeSamplingExt, defer, *[p.weight for p in self.wPrediction.predictors])
        if self.nSkipped:                              # (a slice of nothing would still cost its backward a zero-fill and a copy)
            losses, acc = losses[self.nSkipped:], acc[self.nSkipped:]
        return losses.view(1, -1), acc.view(1, -1)

    def _needs_modules(self):
        net = self.wPrediction
        return net.rnnMode in ('transformer_multi', 'transformer', 'LSTM', 'RNN') or (net.dropout is not None and self.training)
