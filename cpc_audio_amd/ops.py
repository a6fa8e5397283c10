"""torch.autograd.Function wrappers over the C ABI (include/cpc_hip.h).

Each Function is one node of the autograd graph for a whole stage of the hot path
(encoder / autoregressor / criterion), so the Python overhead per train step is three
forward and three backward calls.  All device memory (outputs, saved activations,
scratch) comes from torch's caching allocator; kernels are enqueued on torch's current
stream of the input's device.  There is no CPU implementation: CPU tensors raise.
"""
import ctypes
import os
from contextlib import nullcontext as _nullcontext

import torch

from . import _lib
from ._lib import ptr as _p

_HID = 256


def _require_cuda(t, what):
    if not t.is_cuda:
        raise RuntimeError(f"cpc_audio_amd.{what}: expected a tensor on an AMD GPU (got {t.device}); "
                           "this package has no CPU path")
    if t.dtype != torch.float32:
        raise TypeError(f"cpc_audio_amd.{what}: fp32 expected, got {t.dtype}")


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptrs(ts):
    return (ctypes.c_void_p * len(ts))(*[_p(t) for t in ts])


_layout_cache = {}


def check_device_errors(clear=True):
    """Raise if a kernel of this library flagged an error on the current device since the last check (include/cpc_hip.h,
    cpc_device_error_flags): a persistent-recurrence workgroup that gave up polling, a negative index out of range.
    Synchronises with the device -- the train loops call it where they already synchronise (logging, epoch end)."""
    mask = _lib.get().cpc_device_error_flags(1 if clear else 0)
    if mask < 0:
        raise _lib.CpcHipError("cpc_device_error_flags could not read the device flags")
    what = []
    if mask & 1:
        what.append("a workgroup of the persistent GRU recurrence timed out waiting for its neighbours "
                    "(outputs contain NaN from that step on)")
    if mask & 2:
        what.append("cpc_nce_prepare received negative-sample indices outside [0,B) x [0,S) (they were clamped)")
    if mask & 4:
        what.append("a workgroup of the N-split conv forward timed out waiting for its partner's ChannelNorm statistics "
                    "(its rows contain NaN)")
    if what:
        raise _lib.CpcHipError("device-side error: " + "; ".join(what))

# ---- stream-level overlap inside one train step ---------------------------------------------------------------------
# The dz half of the criterion's backward (per-destination gather-GEMM + one dense GEMM) is not on the way to dc, and
# the network that consumes dc (the persistent GRU backward: 128 workgroups, latency-bound) leaves most of the chip
# idle; the weight-gradient GEMMs hang off the dx chain.  A train loop that wants them on side streams owns a
# StepContext and runs forward + backward inside ``with ctx:``.  All overlap state (events, launches held back)
# lives on that object -- only the side streams are per process, see _side_streams -- so two loops on two threads / devices (nn.DataParallel-style
# replicas) do not see each other.  The Functions pick the context up in forward (on the caller's
# thread) and carry it to backward on their autograd ctx: autograd runs backward on its own worker thread.
# Without an active context every Function is single-stream.
import threading

_tls = threading.local()


MAX_STREAM_CANDIDATES = 12


def streams_overlap(a, b):
    """True if a kernel on stream ``b`` runs while stream ``a`` is busy, i.e. the two do not share a hardware queue
    (cpc_streams_overlap: a ~3 ms blocking probe)."""
    lib = _lib.get()
    out = ctypes.c_int(0)
    lib.check(lib.cpc_streams_overlap(ctypes.c_void_p(a.cuda_stream), ctypes.c_void_p(b.cuda_stream), ctypes.byref(out)),
              "streams_overlap")
    return bool(out.value)


def pick_concurrent_stream(device, priority, beside):
    """A pooled stream that really executes beside every stream in ``beside``.  The HIP runtime maps streams onto four hardware
    queues per priority level in creation order (measured with this probe: torch's pool streams 1..6 pair up as (1,6) (2,5) (3,4)
    in a fresh process; after init_process_group("nccl"), which creates six streams of its own, the default stream shares a queue
    with pool stream 4), and two streams on one queue run in submission order whatever their events say.  Draw from torch's
    pool until the probe says the candidate overlaps with all of ``beside`` (at most MAX_STREAM_CANDIDATES draws, ~3 ms per
    probe, once per process); if none does, the last one is used and a warning names the cost.  During a stream capture nothing
    can be probed: plain draw.  CPC_STREAM_PROBE=0 switches the probe off."""
    st = torch.cuda.Stream(device=device, priority=priority)
    if os.environ.get("CPC_STREAM_PROBE", "1") == "0" or torch.cuda.is_current_stream_capturing():
        return st
    with torch.cuda.device(device):
        for n in range(MAX_STREAM_CANDIDATES):
            if all(streams_overlap(o, st) for o in beside):
                return st
            if n + 1 < MAX_STREAM_CANDIDATES:
                st = torch.cuda.Stream(device=device, priority=priority)
    import warnings
    warnings.warn(f"no stream found that runs beside the step's other streams after {MAX_STREAM_CANDIDATES} draws: the overlapped "
                  "parts of the train step will run in submission order (slower, not wrong)")
    return st


# The side streams themselves are per process and device, shared by every StepContext: a second train loop in the process (a
# second Trainer, the bench's bf16 pass) has nothing to gain from more of them (four hardware queues serve them all).  Two loops on two threads then
# enqueue on the same side streams: each orders its own work with its own events, the streams only add FIFO order between them.
# What two threads must NOT share is the MAIN stream: the C side's pooled events are keyed by it (csrc/capi.hip, stream_events),
# and a record of one thread between another's record and wait re-targets that wait -- with the shared side streams behind
# it, possibly in a circle.  Every step is therefore issued under a claim on (device, current stream), see issuing_step.
_side_streams = {}
_side_streams_lock = threading.Lock()
_issuing = {}                       # (device index, stream handle) -> [thread ident, depth] while a step is being issued
_issuing_lock = threading.Lock()


class issuing_step:
    """``with issuing_step(device):`` around the host-side issue of one train step (StepContext.__enter__ .. __exit__, the
    composite step's calls).  A second thread that starts issuing on the same device AND the same current stream meanwhile
    gets a RuntimeError instead of a step whose stream order is undefined: concurrent train loops on one device (threaded
    data-parallel replicas) each need a stream of their own -- ``with torch.cuda.stream(torch.cuda.Stream()):`` around the loop
    (tests/test_gpu_modules.py::test_two_trainers_on_two_threads_...).  The same thread may nest."""

    def __init__(self, device=None):
        self.device, self.key = device, None

    def __enter__(self):
        if not torch.cuda.is_available():
            return self
        dev = torch.cuda.current_device() if self.device is None else torch.device(self.device).index
        if dev is None:
            dev = torch.cuda.current_device()
        key = (dev, torch.cuda.current_stream(dev).cuda_stream)
        me = threading.get_ident()
        with _issuing_lock:
            held = _issuing.get(key)
            if held is not None and held[0] != me:
                raise RuntimeError(f"cpc_audio_amd: two threads are issuing train steps on the same stream of cuda:{dev}; give "
                                   "every concurrent train loop its own stream (with torch.cuda.stream(torch.cuda.Stream()): ...)")
            if held is None:
                _issuing[key] = [me, 1]
            else:
                held[1] += 1
        self.key = key
        return self

    def __exit__(self, *exc):
        if self.key is not None:
            with _issuing_lock:
                held = _issuing.get(self.key)
                if held is not None:
                    held[1] -= 1
                    if held[1] <= 0:
                        del _issuing[self.key]
            self.key = None
        return False


def _new_side_stream(key, device, which):
    # All side streams at normal priority.  Round 4 ran the weight-gradient stream at high priority for a while -- its last
    # kernel, layer 1's weight gradient, is the tail of the step, and with priority its workgroups are dispatched ahead of
    # conv0's backward beside it: 2.865-2.873 vs 2.880-2.884 ms per step sustained on one box, no difference on another -- but
    # a high-priority stream gets a hardware queue of its own, and that queue must not be the process's FIFTH.  ROCclr's log
    # (AMD_LOG_LEVEL=3) of a run that had touched the GPU before init_process_group("nccl"): the default stream creates queue
    # 1, RCCL's streams queues 2-4 (the normal-priority pool is full at four), the side streams share those, the
    # high-priority stream creates queue 5 -- after which every kernel of the main stream on queue 1 runs 20-40 us longer,
    # alone or not, and the step takes 4.6 ms instead of 2.9 (the same with the roles mirrored, with all side streams at high
    # priority, or with GPU_MAX_HW_QUEUES=8; with the priority dropped, or GPU_MAX_HW_QUEUES <= 3, 2.89 ms in every creation
    # order).  A graph capture's warm-up stream, a second train loop or a caller's own stream would be a fifth queue just the
    # same; with normal-priority streams only, the process stays within the four queues of the normal pool whatever else it
    # creates.  CPC_SIDE_PRIORITY="2"-style lists of `which` opt back in (A/B runs).
    env = os.environ.get("CPC_SIDE_PRIORITY")
    hi = env is not None and str(which) in env.split(",")
    beside = [torch.cuda.current_stream(device)] + [v for k, v in _side_streams.items() if k[0] == key[0]]
    return pick_concurrent_stream(device, -1 if hi else 0, beside)


def _process_side_stream(key, device, which):
    with _side_streams_lock:
        st = _side_streams.get(key)
        if st is None:
            st = _side_streams[key] = _new_side_stream(key, device, which)
        return st


class StepContext:
    """Overlap state of one train loop.  ``overlap``: the criterion's dz path and head gradient on a side stream;
    ``wgrad_stream`` (with overlap): the encoder's / recurrence's weight-gradient GEMMs on their own stream."""

    def __init__(self, overlap=True, wgrad_stream=True):
        self.overlap = bool(overlap)
        self.wgrad_stream = bool(wgrad_stream)
        self._streams = {}
        self.side_events = []       # work the next backward op depends on (dz)
        self.late_events = []       # work only the optimiser reads (the prediction heads' weight gradient)
        self.wgrad_events = []      # the same on the weight-gradient stream (the recurrence's weight / bias gradients)
        self.deferred = []          # side-stream launches held back until the AR backward is in flight (it needs whole
        #                             CUs: its 768-thread workgroups cannot squeeze in beside a chip full of gather blocks)
        self.pre_encoder_backward = []   # callables(ctx) run when the encoder's backward starts: every other gradient
        #                             of the step is final (or queued on the side stream) by then -- FlatGradAllReduce.begin
        self.prepared = None        # what the criterion queued on the side stream at the start of the step
        #                             (CPCUnsupersivedCriterion.prepare_step): negative draws, index preparation, GEMM bounds

    def side_stream(self, device, which=0):
        """which = 0: the criterion's stream (negative draws, dz path, head gradient, early all-reduce bucket);
        1: forward-only preparation of the recurrence's backward (it must not queue in front of the negative draws);
        2: weight gradients."""
        key = (torch.device(device).index, which)
        st = self._streams.get(key)
        if st is None:
            st = self._streams[key] = _process_side_stream(key, device, which)
        return st

    def reserve(self, device):
        """Create the three side streams now.  Hardware queues are handed out in creation order and a process should not need
        more than four (side_stream above): streams created before init_process_group get queues of
        their own, RCCL's then share the fourth."""
        with torch.cuda.device(device):
            return [self.side_stream(device, which) for which in range(3)]

    def abandon(self):
        """After an exception inside an overlapped step: forget the launches still held back and the events not yet
        waited for (what was already enqueued simply completes), so that the next step does not start from stale state."""
        del self.deferred[:]
        del self.side_events[:]
        del self.late_events[:]
        del self.wgrad_events[:]
        self.prepared = None

    def launch_deferred(self):
        while self.deferred:
            self.deferred.pop(0)()

    def wait(self, final=True, wgrad=None):
        """Make the current stream wait for everything launched (or still held) for the side streams.  ``final=False``
        (used between the backward ops) leaves out what only the optimiser reads; the owner calls this with
        final=True after backward() and before touching any ``.grad``.  ``wgrad`` (default: same as ``final``): also
        wait for the weight-gradient stream."""
        self.launch_deferred()
        cur = torch.cuda.current_stream()
        while self.side_events:
            cur.wait_event(self.side_events.pop())
        while final and self.late_events:
            cur.wait_event(self.late_events.pop())
        while (final if wgrad is None else wgrad) and self.wgrad_events:
            cur.wait_event(self.wgrad_events.pop())

    def __enter__(self):
        claim = issuing_step()
        claim.__enter__()                                 # (raises before anything of this context is touched)
        try:
            # where the step starts on the caller's stream: side-stream work that depends on nothing of the step (the negative
            # draws) forks from HERE -- early enough to run beside the encoder, and an explicit fork, which a stream capture
            # (train.Trainer(graph=True)) needs to see the side stream's work as part of the step
            begin = None
            if self.overlap and torch.cuda.is_available():
                begin = torch.cuda.Event()
                begin.record()
        except BaseException as e:
            # __exit__ does not run when __enter__ raises: give the (device, stream) claim back here, or every other thread
            # is refused on this stream until the process ends
            claim.__exit__(type(e), e, e.__traceback__)
            raise
        self._claim, self.begin = claim, begin
        self._prev = getattr(_tls, "ctx", None)
        _tls.ctx = self
        return self

    def __exit__(self, et, ev, tb):
        _tls.ctx = self._prev
        self._claim.__exit__(et, ev, tb)
        if et is not None:
            self.abandon()
        return False


def current():
    """The StepContext active on this thread (None: single-stream)."""
    return getattr(_tls, "ctx", None)


def _overlap(step):
    return step is not None and step.overlap


def _wait(step, **kw):
    if step is not None:
        step.wait(**kw)


_VIEW_NODES = ("PermuteBackward0", "TransposeBackward0", "ViewBackward0", "UnsafeViewBackward0", "AliasBackward0")
_OWN_WAITING_NODES = ("GruFunctionBackward", "TransformerLayerFunctionBackward")


def dz_may_be_deferred(c, z):
    """Whether InfoNCEFunction.backward may hand autograd a dz that is filled later, on the side stream.  True only
    when this package can prove nobody reads it earlier: z reaches EncoderFunction through pure views (its backward
    waits first), and the other consumer of z -- the network that made c -- ends in one of this package's Functions
    (their backward waits before returning dx, which autograd then adds to dz).  Anything else -- criterion mode
    'reverse' puts a torch.flip between the criterion and the encoder, a foreign autoregressor -- gets dz on the
    current stream right away."""
    fn = z.grad_fn
    while fn is not None and type(fn).__name__ in _VIEW_NODES:
        fn = fn.next_functions[0][0]
    if fn is not None and type(fn).__name__ != "EncoderFunctionBackward":
        return False
    fn = c.grad_fn
    while fn is not None and type(fn).__name__ in _VIEW_NODES + ("FlipBackward0", "SliceBackward0"):
        fn = fn.next_functions[0][0]
    return fn is None or type(fn).__name__ in _OWN_WAITING_NODES

# Parity tests set KEEP_DEBUG = True to look at the encoder's saved activations (the ReLU
# masks of the device path, see oracle/cpc_oracle._ReluTieAware).  Never used by the product.
KEEP_DEBUG = False
debug_last = {}


def _layout(kind, fn, n, *args):
    key = (kind,) + args
    v = _layout_cache.get(key)
    if v is None:
        sizes = (ctypes.c_long * n)()
        _lib.Bound.check(fn(*args, sizes), kind)
        v = tuple(sizes)
        _layout_cache[key] = v
    return v


def saved_encoder_activations(saved, B, L):
    """fp32 copies (B, L_i, 256) of the four intermediate activations y0..y3 an EncoderFunction forward left in ``saved``
    (layers 0 and 1 are kept as two fp16 pieces per element in the default mode).  Parity tests only."""
    lib = _lib.get()
    with torch.cuda.device(saved.device) if saved.is_cuda else _nullcontext():
        sizes = _layout("encoder_layout", lib.cpc_encoder_layout, 22, B, L)
        out = []
        for i in range(4):
            y = torch.empty(B, sizes[3 + i], _HID, device=saved.device, dtype=torch.float32)
            lib.check(lib.cpc_encoder_saved_activation(_p(saved), i, _p(y), B, L, _stream() if saved.is_cuda else None),
                      "encoder_saved_activation")
            out.append(y)
    return out


class EncoderFunction(torch.autograd.Function):
    """wave (B,1,L) + the 20 encoder parameters (state-dict order) -> z (B, L/160, 256)."""

    @staticmethod
    def forward(ctx, wave, *params):
        _require_cuda(wave, "EncoderFunction")
        lib = _lib.get()
        B, ch, L = wave.shape
        if ch != 1:
            raise ValueError("CPCEncoder expects (B,1,L) waveforms")
        wave = wave.contiguous()
        params = [p.detach().contiguous() for p in params]
        with torch.cuda.device(wave.device):
            sizes = _layout("encoder_layout", lib.cpc_encoder_layout, 22, B, L)
            saved = torch.empty(sizes[0], device=wave.device, dtype=torch.float32)
            scratch = torch.empty(max(1, sizes[1]), device=wave.device, dtype=torch.float32)
            z = torch.empty(B, sizes[7], _HID, device=wave.device, dtype=torch.float32)
            lib.check(lib.cpc_encoder_forward(_p(wave), _ptrs(params), _p(saved), _p(scratch), _p(z), B, L,
                                              _stream()), "encoder_forward")
        ctx.save_for_backward(wave, saved, z, *params)
        ctx.dims = (B, L, sizes[2])
        ctx.step = current()
        if KEEP_DEBUG:
            debug_last["encoder"] = (saved, sizes, z)
        return z

    @staticmethod
    def backward(ctx, dz):
        lib = _lib.get()
        # dz may carry the criterion's side-stream part.  The weight-gradient stream is not waited for here: the call
        # below runs its own GEMMs there and joins it before it returns.
        step = ctx.step
        _wait(step, final=False)
        if step is not None:
            for hook in step.pre_encoder_backward:
                hook(step)
        wave, saved, z, *params = ctx.saved_tensors
        B, L, nscr = ctx.dims
        dz = dz.contiguous()
        with torch.cuda.device(wave.device):
            scratch = torch.empty(nscr, device=wave.device, dtype=torch.float32)
            grads = [torch.empty_like(p) for p in params]
            if _overlap(step) and step.wgrad_stream:
                # the weight-gradient GEMMs beside the dx chain (its norm backwards stream, conv0's backward is
                # VALU-bound: both leave the matrix pipes idle); the call joins the two streams before it returns
                lib.check(lib.cpc_encoder_backward_streams(_p(wave), _ptrs(params), _p(saved), _p(z), _p(dz), _p(scratch),
                                                           _ptrs(grads), B, L, _stream(),
                                                           step.side_stream(wave.device, 2).cuda_stream), "encoder_backward")
            else:
                lib.check(lib.cpc_encoder_backward(_p(wave), _ptrs(params), _p(saved), _p(z), _p(dz), _p(scratch),
                                                   _ptrs(grads), B, L, _stream()), "encoder_backward")
        _wait(step)                            # the last backward op: everything on the side stream is due now
        return (None, *grads)


class GruFunction(torch.autograd.Function):
    """x (B,S,256), h0 (nl,B,256) or None, 4*nl GRU parameters -> y (B,S,256), hN (nl,B,256)."""

    @staticmethod
    def forward(ctx, x, h0, *params):
        _require_cuda(x, "GruFunction")
        lib = _lib.get()
        B, S, D = x.shape
        nl = len(params) // 4
        leaves = params                           # the tensors autograd knows (normally the module's Parameters)
        if D != _HID or params[1].shape != (3 * _HID, _HID):
            raise NotImplementedError("the HIP GRU is built for dimEncoded == dimOutput == 256")
        x = x.contiguous()
        params = [p.detach().contiguous() for p in params]
        h0c = None if h0 is None else h0.detach().contiguous()
        with torch.cuda.device(x.device):
            sizes = _layout("gru_layout", lib.cpc_gru_layout, 3, B, S, nl)
            saved = torch.empty(sizes[0], device=x.device, dtype=torch.float32)
            scratch = torch.empty(sizes[1], device=x.device, dtype=torch.float32)
            y = torch.empty(B, S, _HID, device=x.device, dtype=torch.float32)
            hN = torch.empty(nl, B, _HID, device=x.device, dtype=torch.float32)
            # train loops (an overlapping StepContext): everything of the two-layer backward that depends on the forward only is
            # prepared now -- the gate-derivative coefficients by the forward recurrence itself (its gate threads hold their
            # inputs in registers), the hand-over buffers and the transposed weights on a side stream beside the criterion's
            # forward -- instead of between the criterion's backward and the recurrence
            coef = None
            step = ctx.step = current()
            if _overlap(step) and nl == 2 and any(ctx.needs_input_grad):
                ncoef = lib.cpc_gru_coef_floats(B, S, nl)
                if ncoef > 0:
                    coef = torch.empty(ncoef, device=x.device, dtype=torch.float32)
                    # the side stream may touch this block only once the current stream has passed this point: the allocator
                    # hands out memory whose previous user may still be queued on the current stream
                    allocated = torch.cuda.Event()
                    allocated.record()
            lib.check(lib.cpc_gru_forward_coef(_p(x), _p(h0c), _ptrs(params), _p(saved), _p(scratch), _p(y), _p(hN), _p(coef),
                                               B, S, nl, _stream()), "gru_forward")
            if coef is not None:
                main, side = torch.cuda.current_stream(), step.side_stream(x.device, 1)
                side.wait_event(allocated)                       # (the work itself depends on the parameters only)
                lib.check(lib.cpc_gru_backward_coef(_p(h0c), _ptrs(params), _p(saved), _p(y), _p(coef), 1, B, S, nl,
                                                    side.cuda_stream), "gru_backward_coef")
                for t in (coef, *params):
                    t.record_stream(side)
                ctx.coef_ready = torch.cuda.Event()
                ctx.coef_ready.record(side)
        ctx.coef = coef
        ctx.leaves = list(leaves) if all(isinstance(q, torch.Tensor) and q.is_leaf for q in leaves) else None
        ctx.save_for_backward(x, saved, y, *params)
        ctx.h0 = h0c
        ctx.dims = (B, S, nl, sizes[2])
        ctx.mark_non_differentiable(hN)
        ctx.set_materialize_grads(False)       # no zero-filled gradient for hN on every step
        return y, hN

    @staticmethod
    def backward(ctx, dy, _dhN):
        lib = _lib.get()
        x, saved, y, *params = ctx.saved_tensors
        B, S, nl, nscr = ctx.dims
        dy = torch.zeros_like(y) if dy is None else dy.contiguous()
        with torch.cuda.device(x.device):
            scratch = torch.empty(nscr, device=x.device, dtype=torch.float32)
            dx = torch.empty_like(x)
            grads = [torch.empty_like(p) for p in params]
            if ctx.coef is not None:
                torch.cuda.current_stream().wait_event(ctx.coef_ready)
            step = ctx.step
            split = _overlap(step) and step.wgrad_stream and nl == 2 and ctx.leaves is not None
            if split:
                # dx on this stream; the weight / bias gradients -- read by nobody before the optimiser -- on the
                # weight-gradient stream, beside the start of the encoder's backward.  They are added to .grad there
                # (autograd gets None for them: it would accumulate on this stream, before they exist).
                wst = step.side_stream(x.device, 2)
                lib.check(lib.cpc_gru_backward_streams(_p(x), _p(ctx.h0), _ptrs(params), _p(saved), _p(y), _p(dy),
                                                       _p(ctx.coef), _p(scratch), _p(dx), _ptrs(grads), B, S, nl,
                                                       _stream(), wst.cuda_stream), "gru_backward")
                with torch.cuda.stream(wst), torch.no_grad():
                    for leaf, g in zip(ctx.leaves, grads):
                        if not leaf.requires_grad:
                            continue
                        if leaf.grad is None:
                            leaf.grad = g.view_as(leaf)
                        else:
                            leaf.grad.add_(g.view_as(leaf))
                            leaf.grad.record_stream(wst)
                for t in [scratch, saved, x, y, dy] + grads + ([] if ctx.coef is None else [ctx.coef]):
                    t.record_stream(wst)
                ev = torch.cuda.Event()
                ev.record(wst)
                step.wgrad_events.append(ev)
                grads = [None] * len(grads)
            else:
                lib.check(lib.cpc_gru_backward_with_coef(_p(x), _p(ctx.h0), _ptrs(params), _p(saved), _p(y), _p(dy),
                                                         _p(ctx.coef), _p(scratch), _p(dx), _ptrs(grads), B, S, nl,
                                                         _stream()), "gru_backward")
        ctx.coef = None                # its hand-over buffers are consumed: a second backward through this node recomputes
        _wait(ctx.step, final=False)   # starts the criterion's deferred dz path beside the recurrence just launched, and makes
        #                         this stream wait for it: autograd adds dx to that dz next
        return (dx, None, *grads)


def candidate_destinations(ext, B, S, K):
    """(perm, row_ptr) for cpc_nce_backward: the B*W*(N+K) candidate slots sorted (stably) by the
    row of z.view(B*S,256) their gradient lands on.  ext: (B,W,N) int32 negative rows; the K
    positives of window (b,t) land on rows b*S + t + k, k = 1..K (criterion.py:210-215)."""
    W = ext.shape[1]
    dev = ext.device
    b = torch.arange(B, device=dev, dtype=torch.int32).view(B, 1, 1)
    t = torch.arange(W, device=dev, dtype=torch.int32).view(1, W, 1)
    k = torch.arange(1, K + 1, device=dev, dtype=torch.int32).view(1, 1, K)
    dest = torch.cat([ext, b * S + t + k], dim=2).reshape(-1)
    sorted_dest, perm = torch.sort(dest, stable=True)
    bounds = torch.arange(B * S + 1, device=dev, dtype=torch.int32)
    row_ptr = torch.searchsorted(sorted_dest, bounds)
    return perm.to(torch.int32), row_ptr.to(torch.int32)


class head_group:
    """``with head_group(lib, (k0, k_total)):`` -- the cpc_nce_* calls inside work on heads k0 .. of a criterion with k_total
    prediction steps (cpc_nce_head_group: the score tiles hold 16 heads, more are walked in groups).  The setting belongs to the
    calling thread, so every function that makes such calls -- autograd runs backward on a thread of its own -- brackets its own."""

    def __init__(self, lib, group):
        self.lib, self.group = lib, group

    def __enter__(self):
        if self.group is not None:
            self.lib.check(self.lib.cpc_nce_head_group(int(self.group[0]), int(self.group[1])), "nce_head_group")

    def __exit__(self, *exc):
        if self.group is not None:
            self.lib.cpc_nce_head_group(0, 0)
        return False


def prepare_negatives(batchIdx, seqIdx, B, S, K, N, group=None):
    """(ext (B,W,Np), perm, row_ptr) int32 from the two int64 draws of sampleClean -- cpc_nce_prepare.  Np = N rounded up to the
    kernels' 16-wide candidate tile (cpc_nce_padded_negatives): the padding entries are masked by position in the scoring
    kernels, so every call that takes these lists is also told N.  ``group`` = (k0, k_total): the lists of heads k0 .. k0+K-1 of
    a criterion with k_total > 16 prediction steps (head_group)."""
    lib = _lib.get()
    W = S - (K if group is None else group[1])
    Np = int(lib.cpc_nce_padded_negatives(N))
    dev = batchIdx.device
    batchIdx, seqIdx = batchIdx.contiguous(), seqIdx.contiguous()
    if batchIdx.dtype != torch.int64 or seqIdx.dtype != torch.int64 or batchIdx.numel() != B * N * W:
        raise ValueError("prepare_negatives: expected two int64 tensors of B*N*W draws")
    with torch.cuda.device(dev):
        ext = torch.empty(B, W, Np, device=dev, dtype=torch.int32)
        perm = torch.empty(B * W * (Np + K), device=dev, dtype=torch.int32)
        row_ptr = torch.empty(B * S + 1, device=dev, dtype=torch.int32)
        work = torch.empty(B * W * (Np + K) + 2 * B * S + 2, device=dev, dtype=torch.int32)
        with head_group(lib, group):
            lib.check(lib.cpc_nce_prepare(_p(batchIdx), _p(seqIdx), _p(ext), _p(perm), _p(row_ptr), _p(work), B, S, K, N,
                                          _stream()), "nce_prepare")
    return ext, perm, row_ptr


class InfoNCEFunction(torch.autograd.Function):
    """c, z (B,S,256), wall (K*256,256), ext (B,W,N) int32, perm, row_ptr -> losses (K), acc (K).
    ``heads``: optionally the K leaf parameters (256,256) that ``wall`` is the row-wise concatenation of.  With
    an overlapping StepContext their gradient is then formed on the side stream as well and accumulated into ``.grad`` directly
    (bit-identical values; ``wall`` itself receives no gradient), which takes that GEMM off the path to the encoder."""

    @staticmethod
    def forward(ctx, c, z, wall, ext, perm, row_ptr, heads=None, defer_dz=False, saved=None, n_valid=None, group=None):
        """saved: optionally the workspace of this call with the GEMM operand bounds already in it (nce_bounds_into).
        n_valid: negatives per window as drawn when ext's rows are padded to the 16-wide tile (prepare_negatives).
        group: (k0, k_total) when ``wall`` holds heads k0 .. of a criterion with k_total > 16 prediction steps (head_group)."""
        _require_cuda(c, "InfoNCEFunction")
        lib = _lib.get()
        B, S, H = c.shape
        K = wall.shape[0] // _HID
        W, N = ext.shape[1], ext.shape[2]
        if n_valid is not None:
            if int(lib.cpc_nce_padded_negatives(int(n_valid))) != N:
                raise ValueError("InfoNCEFunction: ext is not padded for n_valid negatives")
            N = int(n_valid)
        if H != _HID or z.shape != (B, S, _HID) or W != S - (K if group is None else group[1]) or ext.dtype != torch.int32:
            raise ValueError("InfoNCEFunction: inconsistent shapes")
        c, z, wall, ext = c.contiguous(), z.contiguous(), wall.detach().contiguous(), ext.contiguous()
        with torch.cuda.device(c.device), head_group(lib, group):
            sizes = _layout(f"nce_layout{group or ''}", lib.cpc_nce_layout, 6, B, S, K, N)
            fwd = lib.cpc_nce_forward_prepared
            if saved is None or saved.numel() != sizes[0] or saved.device != c.device:
                saved = torch.empty(sizes[0], device=c.device, dtype=torch.float32)
                fwd = lib.cpc_nce_forward
            scratch = torch.empty(sizes[1], device=c.device, dtype=torch.float32)
            losses = torch.empty(K, device=c.device, dtype=torch.float32)
            acc = torch.empty(K, device=c.device, dtype=torch.float32)
            lib.check(fwd(_p(c), _p(z), _p(wall), _p(ext), _p(saved), _p(scratch), _p(losses), _p(acc), B, S, K, N,
                          _stream()), "nce_forward")
        ctx.save_for_backward(c, z, wall, ext, saved, perm, row_ptr)
        ctx.dims = (B, S, K, N, sizes[2])
        ctx.group = group
        ctx.set_materialize_grads(False)       # no zero-filled gradient for the accuracies
        ctx.heads = list(heads) if heads is not None else None
        ctx.step = current()
        ctx.defer_dz = bool(defer_dz)          # the caller's proof that nobody reads dz early (dz_may_be_deferred)
        if ctx.heads is not None and (len(ctx.heads) != K or any(h.shape != (_HID, _HID) for h in ctx.heads)):
            raise ValueError("InfoNCEFunction: heads must be the K (256,256) weights stacked in wall")
        ctx.mark_non_differentiable(acc)
        return losses, acc

    @staticmethod
    def backward(ctx, gloss, _gacc):
        lib = _lib.get()
        c, z, wall, ext, saved, perm, row_ptr = ctx.saved_tensors
        B, S, K, N, nscr = ctx.dims
        gloss = torch.zeros(K, device=c.device) if gloss is None else gloss.contiguous()
        with torch.cuda.device(c.device):
            scratch = torch.empty(nscr, device=c.device, dtype=torch.float32)
            dc, dz, dwall = torch.empty_like(c), torch.empty_like(z), torch.empty_like(wall)
            step = ctx.step
            if _overlap(step) and ctx.group is None:
                main, side = torch.cuda.current_stream(), step.side_stream(c.device)
                ready = torch.cuda.Event()
                heads = ctx.heads if ctx.heads is not None and any(h.requires_grad for h in ctx.heads) else None
                # dc (and dwall, unless the leaf weights are known) now, on this stream; dz = NULL leaves the dz path out
                lib.check(lib.cpc_nce_backward_streams(_p(c), _p(z), _p(wall), _p(ext), _p(perm), _p(row_ptr), _p(saved),
                                                       _p(gloss), _p(scratch), _p(dc), None,
                                                       None if heads else _p(dwall), B, S, K, N,
                                                       main.cuda_stream, main.cuda_stream), "nce_backward")
                ready.record(main)

                def dz_path():            # ... dz later, on the side stream, once the AR backward has been launched
                    side.wait_event(ready)
                    lib.check(lib.cpc_nce_backward_dz(_p(c), _p(wall), _p(perm), _p(row_ptr), _p(saved), _p(scratch), _p(dz),
                                                      B, S, K, N, side.cuda_stream), "nce_backward_dz")
                    for t in (dz, scratch, c, wall, perm, row_ptr, saved):
                        t.record_stream(side)                     # the allocator must not recycle them early
                    ev = torch.cuda.Event()
                    ev.record(side)
                    step.side_events.append(ev)
                if ctx.defer_dz:
                    step.deferred.append(dz_path)
                else:
                    # somebody this package does not know may read dz as soon as this backward returns (criterion mode
                    # 'reverse': a torch.flip; a foreign autoregressor): it is formed now, on this stream
                    lib.check(lib.cpc_nce_backward_dz(_p(c), _p(wall), _p(perm), _p(row_ptr), _p(saved), _p(scratch), _p(dz),
                                                      B, S, K, N, main.cuda_stream), "nce_backward_dz")
                    ready.record(main)         # the head gradient on the side stream reads the same scratch
                if heads:
                    dheads = dwall
                    def dwall_path():     # ... and the head-weight gradient after it: only the optimiser reads it
                        side.wait_event(ready)
                        with torch.cuda.stream(side):
                            lib.check(lib.cpc_nce_backward_dwall(_p(c), _p(saved), _p(scratch), _p(dheads), B, S, K, N,
                                                                 side.cuda_stream), "nce_backward_dwall")
                            with torch.no_grad():
                                for k, h in enumerate(heads):
                                    if not h.requires_grad:
                                        continue
                                    g = dheads[k * _HID:(k + 1) * _HID]
                                    if h.grad is None:
                                        h.grad = g
                                    else:
                                        h.grad.add_(g)
                                        h.grad.record_stream(side)
                        for t in (c, scratch, dheads, saved):
                            t.record_stream(side)
                        ev = torch.cuda.Event()
                        ev.record(side)
                        step.late_events.append(ev)
                    step.deferred.append(dwall_path)
                    dwall = None
            else:
                with head_group(lib, ctx.group):
                    lib.check(lib.cpc_nce_backward(_p(c), _p(z), _p(wall), _p(ext), _p(perm), _p(row_ptr), _p(saved),
                                                   _p(gloss), _p(scratch), _p(dc), _p(dz), _p(dwall), B, S, K, N,
                                                   _stream()), "nce_backward")
        return dc, dz, dwall, None, None, None, None, None, None, None, None


def nce_bounds_into(wall, c_bound, B, S, K, N):
    """A workspace for InfoNCEFunction(saved=...) with the operand bounds of the criterion's GEMMs already in it
    (cpc_nce_bounds on the current stream): max|wall| reduced now -- the weights do not change inside a step -- and |c|
    bounded a priori by ``c_bound``."""
    lib = _lib.get()
    with torch.cuda.device(wall.device):
        sizes = _layout("nce_layout", lib.cpc_nce_layout, 6, B, S, K, N)
        saved = torch.empty(sizes[0], device=wall.device, dtype=torch.float32)
        lib.check(lib.cpc_nce_bounds(None, float(c_bound), _p(wall.detach()), _p(saved), B, S, K, N, _stream()), "nce_bounds")
    return saved


class InfoNCEScoresFunction(torch.autograd.Function):
    """pred (B,W,K*256) from any prediction network, z (B,S,256), ext, perm, row_ptr -> losses (K), acc (K)."""

    @staticmethod
    def forward(ctx, pred, z, ext, perm, row_ptr, n_valid=None, group=None):
        _require_cuda(pred, "InfoNCEScoresFunction")
        lib = _lib.get()
        B, S, H = z.shape
        W, N = ext.shape[1], ext.shape[2]
        if n_valid is not None:                       # (ext rows padded to the 16-wide tile: prepare_negatives)
            if int(lib.cpc_nce_padded_negatives(int(n_valid))) != N:
                raise ValueError("InfoNCEScoresFunction: ext is not padded for n_valid negatives")
            N = int(n_valid)
        K = S - W if group is None else pred.shape[2] // _HID        # (group = (k0, k_total): pred holds heads k0 .. k0+K-1)
        if (H != _HID or pred.shape != (B, W, K * _HID) or ext.dtype != torch.int32
                or (group is not None and W != S - group[1])):
            raise ValueError("InfoNCEScoresFunction: inconsistent shapes")
        pred, z, ext = pred.contiguous(), z.contiguous(), ext.contiguous()
        with torch.cuda.device(z.device), head_group(lib, group):
            sizes = _layout(f"nce_layout{group or ''}", lib.cpc_nce_layout, 6, B, S, K, N)
            saved = torch.empty(sizes[0], device=z.device, dtype=torch.float32)
            scratch = torch.empty(sizes[1], device=z.device, dtype=torch.float32)
            losses = torch.empty(K, device=z.device, dtype=torch.float32)
            acc = torch.empty(K, device=z.device, dtype=torch.float32)
            lib.check(lib.cpc_nce_scores_forward(_p(pred), _p(z), _p(ext), _p(saved), _p(scratch), _p(losses), _p(acc),
                                                 B, S, K, N, _stream()), "nce_scores_forward")
        ctx.save_for_backward(pred, z, ext, saved, perm, row_ptr)
        ctx.dims = (B, S, K, N, sizes[2])
        ctx.group = group
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(acc)
        return losses, acc

    @staticmethod
    def backward(ctx, gloss, _gacc):
        lib = _lib.get()
        pred, z, ext, saved, perm, row_ptr = ctx.saved_tensors
        B, S, K, N, nscr = ctx.dims
        gloss = torch.zeros(K, device=z.device) if gloss is None else gloss.contiguous()
        with torch.cuda.device(z.device), head_group(lib, ctx.group):
            scratch = torch.empty(nscr, device=z.device, dtype=torch.float32)
            dpred, dz = torch.empty_like(pred), torch.empty_like(z)
            lib.check(lib.cpc_nce_scores_backward(_p(pred), _p(z), _p(ext), _p(perm), _p(row_ptr), _p(saved), _p(gloss),
                                                  _p(scratch), _p(dpred), _p(dz), B, S, K, N, _stream()),
                      "nce_scores_backward")
        return dpred, dz, None, None, None, None, None


class TransformerGroupFunction(torch.autograd.Function):
    """x (B,S,256), dropout probability, seed, G + the 13 parameter kinds of G TransformerLayers, each stacked (G, ...)
    (Krelpos possibly None) -> (B,S,G*256), layer g at columns g*256..: the K transformer predictors of the criterion on the
    same input, every kernel launched once for all of them (cpc_transformer_group_{forward,backward}).  Layer g's dropout
    masks are those of a single-layer call with seed + g."""

    @staticmethod
    def forward(ctx, x, drop_p, seed, G, *params):
        _require_cuda(x, "TransformerGroupFunction")
        lib = _lib.get()
        B, S, D = x.shape
        if D != _HID:
            raise NotImplementedError("the HIP transformer layer is built for d_model == 256")
        if S > 512:
            raise NotImplementedError("the HIP attention kernels hold sequences of at most 512 steps")
        if S > 128 and (float(drop_p) > 0 or (torch.is_grad_enabled() and (x.requires_grad or any(
                p is not None and p.requires_grad for p in params)))):
            # (callers -- transformers.TransformerLayer -- route such calls to the torch-op forward)
            raise NotImplementedError("beyond 128 steps the HIP transformer layer is forward-only (inference, no dropout)")
        x = x.contiguous()
        params = [None if p is None else p.detach().contiguous() for p in params]
        if any(p is not None and p.shape[0] != G for p in params):
            raise ValueError("TransformerGroupFunction: every parameter kind must be stacked (G, ...)")
        if params[4] is not None and tuple(params[4].shape[1:]) != (32, S):
            raise ValueError(f"Krelpos is {tuple(params[4].shape[1:])}; the layers were built for sequences of "
                             f"{params[4].shape[2]} steps, got {S} (cpc/transformers.py:22-24)")
        with torch.cuda.device(x.device):
            sizes = _layout("transformer_layout", lib.cpc_transformer_layout, 8, B, S)
            saved = torch.empty(G * sizes[0], device=x.device, dtype=torch.float32)
            scratch = torch.empty(G * sizes[1], device=x.device, dtype=torch.float32)
            out = torch.empty(B, S, G * _HID, device=x.device, dtype=torch.float32)
            lib.check(lib.cpc_transformer_group_forward(_p(x), _ptrs(params), _p(saved), _p(scratch), _p(out), B, S, G,
                                                        float(drop_p), int(seed), _stream()), "transformer_group_forward")
        ctx.drop = (float(drop_p), int(seed))
        if KEEP_DEBUG:                      # one entry per layer, as G single-layer calls would leave
            for g in range(G):
                debug_last.setdefault("transformer", []).append((saved[g * sizes[0]:(g + 1) * sizes[0]], sizes))
        ctx.has_rel = params[4] is not None
        ctx.step = current()
        ctx.save_for_backward(x, saved, *[p for p in params if p is not None])
        ctx.dims = (B, S, G, sizes[2])
        return out

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.get()
        x, saved, *ps = ctx.saved_tensors
        params = ps if ctx.has_rel else ps[:4] + [None] + ps[4:]
        B, S, G, nscr = ctx.dims
        dy = dy.contiguous()
        with torch.cuda.device(x.device):
            scratch = torch.empty(G * nscr, device=x.device, dtype=torch.float32)
            dx = torch.empty_like(x)
            grads = [None if p is None else torch.empty_like(p) for p in params]
            lib.check(lib.cpc_transformer_group_backward(_p(x), _ptrs(params), _p(saved), _p(dy), _p(scratch), _p(dx),
                                                         _ptrs(grads), B, S, G, ctx.drop[0], ctx.drop[1], _stream()),
                      "transformer_group_backward")
        _wait(ctx.step, final=False)
        return (dx, None, None, None, *grads)


class TransformerLayerFunction(torch.autograd.Function):
    """x (B,S,256), dropout probability, seed + the 13 layer parameters (state-dict order, Krelpos possibly None) -> (B,S,256).
    One TransformerLayer of cpc/transformers.py:103-111 through cpc_transformer_layer_{forward,backward}_dropout."""

    @staticmethod
    def forward(ctx, x, drop_p, seed, *params):
        """drop_p, seed: dropout probability of this call (0: none) and the 64-bit seed its masks derive from."""
        _require_cuda(x, "TransformerLayerFunction")
        lib = _lib.get()
        B, S, D = x.shape
        if D != _HID:
            raise NotImplementedError("the HIP transformer layer is built for d_model == 256")
        if S > 512:
            raise NotImplementedError("the HIP attention kernels hold sequences of at most 512 steps")
        if S > 128 and (float(drop_p) > 0 or (torch.is_grad_enabled() and (x.requires_grad or any(
                p is not None and p.requires_grad for p in params)))):
            # (callers -- transformers.TransformerLayer -- route such calls to the torch-op forward)
            raise NotImplementedError("beyond 128 steps the HIP transformer layer is forward-only (inference, no dropout)")
        x = x.contiguous()
        params = [None if p is None else p.detach().contiguous() for p in params]
        if params[4] is not None and tuple(params[4].shape) != (32, S):
            raise ValueError(f"Krelpos is {tuple(params[4].shape)}; the layer was built for sequences of "
                             f"{params[4].shape[1]} steps, got {S} (cpc/transformers.py:22-24)")
        with torch.cuda.device(x.device):
            sizes = _layout("transformer_layout", lib.cpc_transformer_layout, 8, B, S)
            saved = torch.empty(sizes[0], device=x.device, dtype=torch.float32)
            scratch = torch.empty(sizes[1], device=x.device, dtype=torch.float32)
            out = torch.empty(B, S, _HID, device=x.device, dtype=torch.float32)
            lib.check(lib.cpc_transformer_layer_forward_dropout(_p(x), _ptrs(params), _p(saved), _p(scratch), _p(out), B, S,
                                                                float(drop_p), int(seed), _stream()),
                      "transformer_layer_forward")
        ctx.drop = (float(drop_p), int(seed))
        if KEEP_DEBUG:                      # parity tests read the FFN's ReLU mask (oracle _ReluTieAware)
            debug_last.setdefault("transformer", []).append((saved, sizes))
        ctx.has_rel = params[4] is not None
        ctx.step = current()
        ctx.save_for_backward(x, saved, *[p for p in params if p is not None])
        ctx.dims = (B, S, sizes[2])
        return out

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.get()
        x, saved, *ps = ctx.saved_tensors
        params = ps if ctx.has_rel else ps[:4] + [None] + ps[4:]
        B, S, nscr = ctx.dims
        dy = dy.contiguous()
        with torch.cuda.device(x.device):
            scratch = torch.empty(nscr, device=x.device, dtype=torch.float32)
            dx = torch.empty_like(x)
            grads = [None if p is None else torch.empty_like(p) for p in params]
            lib.check(lib.cpc_transformer_layer_backward_dropout(_p(x), _ptrs(params), _p(saved), _p(dy), _p(scratch), _p(dx),
                                                                 _ptrs(grads), B, S, ctx.drop[0], ctx.drop[1], _stream()),
                      "transformer_layer_backward")
        _wait(ctx.step, final=False)
        return (dx, None, None, *grads)
