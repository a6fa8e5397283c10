"""Minimal train-step harness with the semantics of the reference's cpc/train.py:trainStep
(:78-99): forward, ``allLosses.sum().backward()``, gradient SUM-all-reduce across ranks (in
place of nn.DataParallel's reduce-add), Adam step, zero_grad.  Builders mirror
cpc/feature_loader.py:124-153 and cpc/train.py:24-51 for the north-star configuration
(--arMode GRU --nLevelsGRU 2 --rnnMode linear)."""
import torch

from .criterion import CPCUnsupersivedCriterion
from .dist import FlatGradAllReduce
from .model import CPCAR, CPCEncoder, CPCModel
from .optim import Adam


def build_model(hiddenEncoder=256, hiddenGar=256, nLevelsGRU=2, keepHidden=False, reverse=False, arMode="GRU",
                sizeWindow=20480, abspos=False, transformerDropout=0.1):
    """cpc/feature_loader.py:124-153 (getEncoder / getAR) + cpc/train.py:311.  arMode 'GRU' (north star) or
    'transformer' (BASELINE.json config 4: buildTransformerAR(hiddenEncoder, 1, sizeWindow // 160, abspos))."""
    enc = CPCEncoder(hiddenEncoder, "layerNorm")
    if arMode == "transformer":
        from .transformers import buildTransformerAR
        ar = buildTransformerAR(hiddenEncoder, 1, sizeWindow // 160, abspos, dropout=transformerDropout)
    else:
        ar = CPCAR(hiddenEncoder, hiddenGar, keepHidden, nLevelsGRU, mode=arMode, reverse=reverse)
    return CPCModel(enc, ar)


def build_criterion(nPredicts=12, hiddenGar=256, hiddenEncoder=256, negativeSamplingExt=128,
                    sizeWindow=20480, downsampling=160, mode=None, rnnMode="linear", transformerDropout=0.1, dropout=False):
    return CPCUnsupersivedCriterion(nPredicts, hiddenGar, hiddenEncoder, negativeSamplingExt, mode=mode,
                                    rnnMode=rnnMode, dropout=dropout, sizeInputSeq=sizeWindow // downsampling,
                                    transformerDropout=transformerDropout)


def load_flat_params(model, criterion, params):
    """Load a dict keyed like the reference's state dicts (gEncoder.*, gAR.*, wPrediction.*)."""
    model.load_state_dict({k: v for k, v in params.items() if not k.startswith("wPrediction")}, strict=True)
    criterion.load_state_dict({k: v for k, v in params.items() if k.startswith("wPrediction")}, strict=True)


class CompositeStep:
    """Forward + backward of the north-star configuration through ONE C call (cpc_train_step, csrc/train_step.hip) on the four
    streams of an ops.StepContext, with the gradients written straight into the flat buffer of a dist.FlatGradAllReduce
    (``p.grad`` become views of it) -- what train.Trainer and harness.train_epoch run wherever ``ok()`` says the configuration
    is the one the composite covers; both fall back to the autograd-driven path otherwise.  ``forward_backward`` leaves every
    gradient final on the current stream (in a process group: with the early bucket's all-reduce started; the caller's
    ``allreduce()`` finishes the exchange) and returns (losses (1, K), accuracies (1, K)) as fresh tensors."""

    def __init__(self, model, criterion, ctx, allreduce):
        self.model, self.criterion, self.ctx, self.allreduce = model, criterion, ctx, allreduce
        self.tables = None                # pointer tables / workspace of the last batch shape
        self.mid_wait = None              # data parallel: how a stream waits for conv2..4's weight gradients of the last step

    # ---- cross-step pipelining of a single-rank loop (``forward_backward(open_tail=True)`` + ``finish(optimizer)``) -------------
    # The step's last kernel is layer 1's weight gradient on the weight-gradient stream: it runs ~0.13 ms past the end of the
    # main stream's chain, and behind it came Adam, 40-60 us of weight re-layout and conv0 -- which needs none of
    # conv1.weight.  With an open tail the main stream does not wait for that kernel: ``finish`` updates conv1.weight on the
    # weight-gradient stream behind it and everything else on the main stream (optim.Adam.step_split), prepares the next step's
    # weight layouts right there (cpc_train_step_tail) and the next ``forward_backward`` starts with conv0 under the tail; its
    # layer 1 waits for conv1's update.  Same kernels on the same values: bit-identical results.  Between an open-tailed step and
    # the next one conv1.weight and its optimiser state belong to the weight-gradient stream -- ``join()`` before anything else
    # reads or writes parameters on the current stream (the train loops of this package do; a parameter changed in place by
    # torch in between is detected by its version counter and costs one preparation at the head of the step, as before).
    def join(self):
        """The current stream waits for the tail of the last open-tailed step (no host synchronisation)."""
        f = self.tables
        if f is None:
            return
        # whoever joins is about to touch parameters outside the composite -- possibly through raw pointers (optim.Adam's
        # launches do not bump torch's version counters): the layouts the tail prepared are not trusted afterwards, the next
        # composite step prepares its own (one ~45 us preparation per join, i.e. per epoch / validation pass)
        f["ready"] = None
        if f.get("tail"):
            # forward_backward(open_tail=True) whose finish() never ran (an exception in between): close the tail
            from . import _lib
            dev = f["ws"].device
            with torch.cuda.device(dev):
                _lib.get().check(_lib.get().cpc_train_step_wait(f["main"], 1, torch.cuda.current_stream(dev).cuda_stream),
                                 "train_step_wait")
            f["tail"] = False
        if f.get("open"):
            from . import _lib
            dev = f["ws"].device
            with torch.cuda.device(dev):
                _lib.get().check(_lib.get().cpc_train_step_wait(f["main"], 2, torch.cuda.current_stream(dev).cuda_stream),
                                 "train_step_wait")
            f["open"] = False

    def finish(self, optimizer):
        """After ``forward_backward(open_tail=True)`` (and instead of ``optimizer.step()`` when it returns True): the split
        update and the next step's weight preparation.  False: the tail has been closed (the current stream has waited for layer
        1's weight gradient) and the caller runs ``optimizer.step()`` as usual -- the first step of a run (the optimiser's
        one-launch path arms itself there), a foreign optimiser."""
        f = self.tables
        if f is None or not f.get("tail"):
            return False
        from . import _lib
        from .optim import Adam
        lib = _lib.get()
        f["tail"] = False
        dev = f["ws"].device
        with torch.cuda.device(dev):
            main = torch.cuda.current_stream(dev)
            prep, wst = self.ctx.side_stream(dev, 1), self.ctx.side_stream(dev, 2)
            enc = self.model.gEncoder
            conv0 = [enc.conv0.weight, enc.conv0.bias, enc.batchNorm0.weight, enc.batchNorm0.bias]
            c0 = {id(p) for p in conv0}
            rest = [p for p in f["plist"] if id(p) not in c0 and p is not enc.conv1.weight]
            ok = main.cuda_stream == f["main"] and isinstance(optimizer, Adam)
            if ok:
                # the gradients of "the rest" -- the recurrence's and conv2..4's weights (weight-gradient stream), the heads' (side
                # stream), the bias / norm gradients of layers 1..4 (already on the preparation stream) -- were final long before
                # the chain's last kernels: their update and their layouts run on the preparation stream beside those
                for which in (0, 3):
                    lib.check(lib.cpc_train_step_wait(f["main"], which, prep.cuda_stream), "train_step_wait")
                ok = optimizer.step_split([([enc.conv1.weight], wst), (rest, prep)])      # conv0's four tensors: current stream
            if not ok:
                lib.check(lib.cpc_train_step_wait(f["main"], 1, main.cuda_stream), "train_step_wait")
                return False
            B, L, K, N = f["key"][:4]
            f["parity"] ^= 1
            lib.check(lib.cpc_train_step_tail(f["params"], _lib.ptr(f["ws"]), B, L, K, N, f["parity"], main.cuda_stream,
                                              prep.cuda_stream, wst.cuda_stream), "train_step_tail")
            f["open"] = True
            f["ready"] = tuple(p._version for p in f["plist"])
        return True

    def ok(self, batchData, negatives=None):
        from . import ops
        from .model import CPCAR, CPCEncoder, CPCModel
        m, cr = self.model, self.criterion
        if not (torch.is_tensor(batchData) and batchData.is_cuda and batchData.dtype == torch.float32
                and batchData.dim() == 3 and batchData.shape[1] == 1 and torch.is_grad_enabled() and not ops.KEEP_DEBUG):
            return False
        # exactly this package's classes, without hooks or parametrizations: the composite never calls a module's __call__ /
        # forward, so a subclass override, a forward / backward hook or a parametrized weight would be skipped silently --
        # such setups take the autograd path
        if not (type(m) is CPCModel and type(m.gEncoder) is CPCEncoder and type(m.gAR) is CPCAR
                and type(cr) is CPCUnsupersivedCriterion):
            return False
        mods = self.__dict__.get("_mods")
        if mods is None or mods[0] is not m or mods[1] is not cr:       # (the module tree of a CPCModel / criterion is fixed)
            mods = self._mods = (m, cr, list(m.modules()) + list(cr.modules()))
        for mod in mods[2]:
            if (mod._forward_hooks or mod._forward_pre_hooks or mod._backward_hooks or mod._backward_pre_hooks
                    or getattr(mod, "parametrizations", None)):
                return False
        ar = m.gAR
        if not (m.gEncoder.hip and ar.hip):                # (options served by torch ops: model.CPCEncoder / CPCAR)
            return False
        if (ar.reverse or ar.baseNet.num_layers != 2 or cr.mode is not None or cr.wPrediction.scores_apart
                or cr.nPredicts > 16):                       # (more heads: walked in groups by the criterion module)
            return False
        if negatives is not None and not all(torch.is_tensor(t) and t.is_cuda and t.dtype == torch.int64 for t in negatives):
            return False
        ps = self.allreduce.params
        own = {id(p) for p in list(cr.parameters()) + list(m.parameters())}
        if not (len(ps) == 20 + 8 + cr.nPredicts and all(id(p) in own for p in ps)
                and all(p.requires_grad and p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in ps)):
            return False
        # the K heads' gradients must be one contiguous block of the flat buffer, in head order (a caller-supplied
        # FlatGradAllReduce may order them otherwise: autograd path instead of an exception from _tables)
        pos = {id(p): i for i, p in enumerate(ps)}
        heads = [pos.get(id(h.weight)) for h in cr.wPrediction.predictors]
        return None not in heads and all(b == a + 1 for a, b in zip(heads, heads[1:]))

    def _tables(self, batchData):
        """Pointer tables and workspace of cpc_train_step for this batch shape; rebuilt when a parameter got new storage."""
        import ctypes
        from . import _lib
        lib = _lib.get()
        m, cr = self.model, self.criterion
        with torch.no_grad():
            wall = cr.wPrediction.stacked_weight()                    # (re-)establishes the K heads as views of one buffer
        heads = [p.weight for p in cr.wPrediction.predictors]
        plist = m.gEncoder._flat_params() + m.gAR._flat_params()
        B, _, L = batchData.shape
        K, N = cr.nPredicts, cr.negativeSamplingExt
        key = (B, L, K, N, batchData.device, lib.cpc_get_mfma_mode(), wall.data_ptr()) + tuple(p.data_ptr() for p in plist)
        f = self.tables
        if f is not None and f["key"] == key:
            return f
        self.join()                                                    # (the old workspace may still be in use by an open tail)
        views = self.allreduce._views(plist + heads)                   # the flat gradient buffer: gradients are written in place
        hv = views[len(plist):]
        step = hv[0].numel() * hv[0].element_size()
        if any(v.data_ptr() != hv[0].data_ptr() + k * step for k, v in enumerate(hv)):
            raise RuntimeError("CompositeStep: the prediction heads' gradients are not contiguous in the flat buffer")
        for p in plist:
            if not p.is_contiguous():
                raise RuntimeError("CompositeStep: non-contiguous parameter")
        arr = ctypes.c_void_p * 29
        sizes = (ctypes.c_long * 8)()
        with torch.cuda.device(batchData.device):
            lib.check(lib.cpc_train_step_layout(B, L, K, N, sizes), "train_step_layout")
            ws = torch.empty(sizes[0], device=batchData.device, dtype=torch.float32)
        f = self.tables = {
            "key": key, "params": arr(*([p.data_ptr() for p in plist] + [wall.data_ptr()])),
            "grads": arr(*([v.data_ptr() for v in views[:len(plist)]] + [hv[0].data_ptr()])),
            "plist": plist + heads, "views": views, "ws": ws, "S": int(sizes[3]), "sizes": tuple(sizes),
            "ones": torch.ones(K, device=batchData.device), "hN": torch.empty(2, B, 256, device=batchData.device),
            "parity": 0, "open": False, "ready": None, "tail": False, "main": None}
        return f

    def forward_backward(self, batchData, negatives=None, prefetch=False, open_tail=False):
        from . import _lib
        lib = _lib.get()
        dev = batchData.device
        batchData = batchData.contiguous()
        f = self._tables(batchData)
        m, cr, ctx = self.model, self.criterion, self.ctx
        ar = m.gAR
        B, _, L = batchData.shape
        K, N, S = cr.nPredicts, cr.negativeSamplingExt, f["S"]
        from .ops import issuing_step
        with torch.cuda.device(dev), issuing_step(dev):
            main = torch.cuda.current_stream(dev)
            side, prep, wst = ctx.side_stream(dev, 0), ctx.side_stream(dev, 1), ctx.side_stream(dev, 2)
            pkey = (B, L, K, N)
            if negatives is None and f.get("prefetched") == pkey:
                bidx = sidx = None                       # drawn and prepared at the end of the previous step (below)
                f["prefetched"] = None
            elif negatives is None:
                # the two draws of sampleClean (criterion.py:181-189) on the side stream, forked from the start of the step:
                # torch's generator, in the reference's order
                begin = torch.cuda.Event()
                begin.record(main)
                side.wait_event(begin)
                with torch.cuda.stream(side):
                    bidx, sidx = cr.drawNegatives(B, S, S - K, dev)
            else:
                bidx, sidx = negatives[0].contiguous(), negatives[1].contiguous()
                if bidx.numel() != B * N * (S - K) or sidx.numel() != bidx.numel():
                    raise ValueError("negatives must be two int64 tensors of B*N*W draws")
                bidx.record_stream(side)
                sidx.record_stream(side)
            h0 = ar.hidden
            bounded = h0 is None or h0 is getattr(ar, "_own_hidden", None)
            if h0 is not None:
                h0 = h0.contiguous()
                if h0.shape != (2, B, 256) or not h0.is_cuda or h0.dtype != torch.float32:
                    raise ValueError("CPCAR.hidden does not match the batch")
            out = torch.empty(2, K, device=dev, dtype=torch.float32)
            hN = torch.empty(2, B, 256, device=dev, dtype=torch.float32) if ar.keepHidden else f["hN"]
            for p, v in zip(f["plist"], f["views"]):
                if p.grad is not v:
                    p.grad = v                                  # gradients live in the flat buffer, overwritten by every step
            P = _lib.ptr
            # the weight layouts the previous step's tail prepared are valid if nothing touched a parameter since (torch's
            # version counters; this package's kernels do not count) and the step runs on the stream they were ordered for
            ready = (f["ready"] is not None and f["main"] == main.cuda_stream
                     and f["ready"] == tuple(p._version for p in f["plist"]))
            if f["open"] and not ready:
                self.join()
            f["ready"] = None
            f["open"] = False                              # (ready: the call itself waits, in front of layer 1)
            f["main"] = main.cuda_stream
            flags = (8 if ready else 0) | (16 if f["parity"] else 0)
            tail = bool(open_tail) and not self.allreduce._active() and not torch.cuda.is_current_stream_capturing()
            f["tail"] = False

            def call(phases):
                lib.check(lib.cpc_train_step(P(batchData), P(bidx), P(sidx), P(h0), 1.0 if bounded else 0.0, f["params"],
                                             f["grads"], P(f["ones"]), P(f["ws"]), out[0].data_ptr(), out[1].data_ptr(), P(hN),
                                             B, L, K, N, phases, main.cuda_stream, side.cuda_stream, prep.cuda_stream,
                                             wst.cuda_stream), "train_step")
            try:
                if self.allreduce._active():
                    # data parallel: the heads' and the recurrence's gradients leave for the other ranks while the encoder's
                    # backward runs (dist.FlatGradAllReduce.begin: on the side stream, behind the heads' gradient there and
                    # behind the recurrence's on the weight-gradient stream)
                    call(1 | flags)
                    ev = torch.cuda.Event()
                    ev.record(wst)
                    ctx.wgrad_events.append(ev)
                    self.allreduce.begin(ctx)
                    del ctx.wgrad_events[:]
                    call(2 | (flags & 16))
                    # conv2..4's weight gradients were final on the weight-gradient stream long before the call's last kernel:
                    # their bucket goes out beside the rest of the backward (dist.FlatGradAllReduce.__call__(mid_wait=...))
                    self.mid_wait = lambda stream, m=main.cuda_stream: lib.check(
                        lib.cpc_train_step_wait(m, 0, stream.cuda_stream), "train_step_wait")
                else:
                    self.mid_wait = None
                    call(3 | flags | (4 if tail else 0))
                    f["tail"] = tail
            except BaseException:
                del ctx.wgrad_events[:]
                self.allreduce.abort()
                raise
            if ar.keepHidden:
                ar.hidden = ar._own_hidden = hN                 # cpc/model.py:194-198 (a fresh tensor per step: nothing aliases it)
            if negatives is None and prefetch:
                # the next step's draws + their index lists now, on the side stream behind this step's last reader of the lists
                with torch.cuda.stream(side):
                    nb, ns = cr.drawNegatives(B, S, S - K, dev)
                lib.check(lib.cpc_train_step_prefetch(P(nb), P(ns), P(f["ws"]), B, L, K, N, side.cuda_stream), "train_step_prefetch")
                f["prefetched"] = pkey
        return out[0:1], out[1:2]


class Trainer:
    """``graph=True`` (single GPU): the whole step -- forward, backward on all streams, Adam, zero_grad -- is captured
    ONCE as a HIP graph and replayed with one launch per step.  The step is ~70 kernel launches on four streams issued from
    Python (3-5 ms of host time per step, measured, against 3.9 ms of GPU time): on a slow or busy host the eager loop is
    host-bound, the replayed graph is not.  Static shapes only (a new batch shape re-captures); ``negatives`` supplied by
    the caller, a learning-rate change or world_size > 1 fall back to the eager step."""

    AUTO_PROBE_STEPS = 6      # graph="auto": eager steps timed (host enqueue time vs GPU time) before deciding

    def __init__(self, model, criterion, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, graph=False, check_errors_every=256,
                 fused=True, prefetch_negatives=False, pipeline_tail=False):
        """check_errors_every: every that many steps the device-side error flags are read (ops.check_device_errors: a
        recurrence workgroup that gave up polling, a negative index out of range) and turned into an exception -- one device
        synchronisation per that many steps; 0 leaves the check to the caller.
        graph: False (eager), True (HIP graph replay), or "auto": the first steps run eagerly and are timed; the graph is
        used only if the host needs more than 85 % of the GPU's step time to issue a step.  (Measured on MI355X boxes: the
        replayed graph costs 0.24-0.28 ms of host time per step (the composite step issued eagerly: 0.5 ms) and runs 6 % longer on
        the GPU than the eagerly issued streams (3.10 vs 2.93 ms, round 4) -- a win only where the host cannot keep up, see step.)
        fused (default True): where the configuration is the north-star one -- CPCEncoder + 2-layer GRU CPCAR + linear
        heads, every parameter trainable -- forward and backward of a step are issued by ONE C call (cpc_train_step, csrc/
        train_step.hip: the same entry points in the same order on the same four streams, bit-identical results) into a
        persistent workspace, with the gradients written straight into the flat all-reduce buffer; anything else (transformer
        AR / predictors, criterion mode 'reverse', frozen parameters) runs the autograd path below."""
        self.model, self.criterion = model, criterion
        self.fused = bool(fused)
        # pipeline_tail (composite step, one rank, eager launches): leave the tail of every step open -- the next step's conv0 runs
        # under this step's last weight gradient, conv1.weight is updated on the weight-gradient stream (CompositeStep.finish).
        # Off by default because of what it asks of the caller: between two steps conv1.weight belongs to that stream, so
        # anything else that touches parameters, gradients or optimiser state on the current stream calls ``join()`` first
        # (state_dict(), validation, a checkpoint -- harness.train_epoch and bench.py do).  CPC_PIPELINE_TAIL=1 / 0 overrides.
        import os as _os
        env_pt = _os.environ.get("CPC_PIPELINE_TAIL")
        self.pipeline_tail = bool(pipeline_tail) if env_pt is None else env_pt == "1"
        # composite step only, off by default: draw the NEXT step's negatives and prepare their index lists at the end of a step
        # (beside its tail: layer 1's weight gradient alone on the matrix pipes) instead of behind conv0 of the next one.  The
        # draws come from torch's generator in the same order either way -- the A/B's losses are equal to the last digit --; only
        # code that draws from the same generator BETWEEN two train steps (a validation pass) would see them one step early.
        # Measured neutral (2.880 vs 2.883 ms per step sustained, three alternations: what the preparation costs conv1 / conv2
        # at the start of a step it costs the weight gradient at the end), so the reference's draw-inside-the-step order stays.
        # CPC_PREFETCH_NEGATIVES=1 / 0 overrides (A/B runs).
        import os
        env = os.environ.get("CPC_PREFETCH_NEGATIVES")
        self.prefetch_negatives = bool(prefetch_negatives) if env is None else env == "1"
        params = list(criterion.parameters()) + list(model.parameters())      # train.py:332
        self.optimizer = Adam(params, lr=lr, betas=betas, eps=eps)      # train.py:335-337; one launch per step on the GPU
        enc = {id(p) for p in model.gEncoder.parameters()} if hasattr(model, "gEncoder") else set()
        # three buckets: heads + recurrence (final when the encoder's backward starts), the weight gradients of conv2..4 (final
        # on the weight-gradient stream ~0.5 ms before the backward ends; composite step only), the rest -- conv0, conv1 and the
        # 256-element bias / norm gradients, 0.53 M values = 2.1 MB: the one collective a step exposes
        mid = [getattr(model.gEncoder, f"conv{i}").weight for i in (2, 3, 4)] if enc and hasattr(model.gEncoder, "conv4") else None
        self.allreduce = FlatGradAllReduce(params, early=[p for p in params if id(p) not in enc] if enc else None, mid=mid)
        from . import ops
        self.ctx = ops.StepContext(overlap=True)
        if params and params[0].is_cuda and not torch.cuda.is_current_stream_capturing():
            # the side streams exist from here on: a Trainer built before init_process_group("nccl") keeps hardware queues of
            # its own (the same step took 2.9 ms with them created before RCCL's streams, 4.6 after, when
            # the process had touched the GPU before the group was set up)
            self.ctx.reserve(params[0].device)
        self.ctx.pre_encoder_backward.append(self.allreduce.begin)
        self._composite = CompositeStep(model, criterion, self.ctx, self.allreduce)
        self.graph = True if graph is True else ("auto" if graph == "auto" else False)
        self._probe = []                  # graph="auto": (host seconds, start event, end event) of the first eager steps
        self._captured = None             # (key, CUDAGraph, static input, static label, static outputs)
        self._capturing = False           # inside capture(): no host-side throttle, no completion events (see _eager_step)
        self._done_events = []
        self.check_errors_every = int(check_errors_every)
        self._steps = 0

    def _ones_like(self, t):
        o = getattr(self, "_ones", None)
        if o is None or o.shape != t.shape or o.device != t.device or o.dtype != t.dtype:
            o = self._ones = torch.ones_like(t)
        return o

    # Eager steps in flight.  The host issues a step in ~1.5 ms, the GPU runs it in 3.4: unchecked, the host runs ahead until
    # the stream's queue pushes back, every step in flight holds its own workspaces (blocks used on a side stream return
    # to torch's allocator only when that stream has passed them), and the allocator keeps asking the driver for more --
    # measured at B = 128: 27 hipMalloc calls in 10 steps after warm-up, steps of 6 ms with stalls to 9-17 ms, reserved
    # memory still growing.  Two steps ahead keep the GPU fed and bound both.
    MAX_IN_FLIGHT = 2

    # ---- the composite step (cpc_train_step): CompositeStep below ----------------------------------------------------------
    @property
    def _fused(self):                      # (tests look at it: the pointer tables / workspace of the composite, None until used)
        return self._composite.tables

    def _fused_ok(self, batchData, negatives):
        return self.fused and self._composite.ok(batchData, negatives)

    def _fused_step(self, batchData, label, negatives=None):
        throttle = not self._capturing
        if throttle:
            done = self._done_events
            if len(done) >= self.MAX_IN_FLIGHT:
                import time
                t0 = time.perf_counter()
                done.pop(0).synchronize()
                self.wait_seconds = getattr(self, "wait_seconds", 0.0) + (time.perf_counter() - t0)
        from .ops import issuing_step
        with torch.cuda.device(batchData.device), issuing_step(batchData.device):
            try:
                losses, acc = self._composite.forward_backward(batchData, negatives,
                                                               prefetch=self.prefetch_negatives and not self._capturing,
                                                               open_tail=self.pipeline_tail and not self._capturing)
                self.allreduce(mid_wait=self._composite.mid_wait)
                if not self._composite.finish(self.optimizer):
                    self.optimizer.step()
            except BaseException:
                # a step that dies between forward_backward(open_tail=True) and finish() leaves layer 1's weight gradient on
                # its stream: the current stream takes it back before anybody's except / finally reads parameters
                if not self._capturing:
                    self._composite.join()
                raise
            self.optimizer.zero_grad()
            if throttle:
                ev = torch.cuda.Event()
                ev.record()
                self._done_events.append(ev)
        return losses, acc

    def join(self):
        """The current stream waits for the open tail of the last step (``pipeline_tail``): call it before reading or writing
        parameters, gradients or optimiser state outside ``step()``.  No host synchronisation; a no-op otherwise."""
        self._composite.join()

    def _eager_step(self, batchData, label, negatives=None):
        if self._fused_ok(batchData, negatives):
            return self._fused_step(batchData, label, negatives)
        self._composite.join()
        # the overlap state (side streams, events, launches held back) lives on this Trainer's StepContext: two Trainers
        # on two threads / devices do not share any
        # (not while a capture is being prepared or recorded: an event recorded into a capturing stream belongs to the
        # graph and must never be synchronised with from the host)
        throttle = batchData.is_cuda and not self._capturing
        if throttle:
            done = self._done_events
            if len(done) >= self.MAX_IN_FLIGHT:
                import time
                t0 = time.perf_counter()
                done.pop(0).synchronize()
                self.wait_seconds = getattr(self, "wait_seconds", 0.0) + (time.perf_counter() - t0)   # not host WORK
        try:
            with self.ctx as step:
                self._prepare_criterion(step, batchData, negatives)
                c_feature, encoded_data, label = self.model(batchData, label)
                allLosses, allAcc = self.criterion(c_feature, encoded_data, label, negatives=negatives)
                # allLosses.sum().backward() (train.py:85-87) without the sum / fill / expand kernels: d sum / d loss_k = 1
                torch.autograd.backward([allLosses], [self._ones_like(allLosses)])
                step.wait()
        except BaseException:
            self.allreduce.abort()
            raise
        self.allreduce()
        self.optimizer.step()
        self.optimizer.zero_grad()
        if throttle:
            ev = torch.cuda.Event()
            ev.record()
            self._done_events.append(ev)
        return allLosses.detach(), allAcc.detach()

    def _prepare_criterion(self, step, batchData, negatives):
        from .harness import prepare_criterion
        prepare_criterion(step, self.model, self.criterion, batchData, negatives)

    def _graph_key(self, batchData):
        lrs = tuple(float(g["lr"]) for g in self.optimizer.param_groups)
        from . import _lib
        # ... and the library's arithmetic / storage mode: its kernels are frozen into the graph
        return (tuple(batchData.shape), batchData.device, self.model.training, lrs, _lib.get().cpc_get_mfma_mode())

    def _graph_safe(self):
        """What a replayed graph cannot express: a recurrent state carried from step to step on the Python side
        (CPCAR.keepHidden swaps a tensor per step) and per-call host random numbers (the transformer layers' dropout seeds)."""
        ar = getattr(self.model, "gAR", None)
        if getattr(ar, "keepHidden", False):
            return False
        wp = getattr(self.criterion, "wPrediction", None)
        if getattr(wp, "dropout", None) is not None and wp.training:      # torch draws the masks per call
            return False
        from .transformers import TransformerLayer
        for mod in list(self.model.modules()) + list(self.criterion.modules()):
            if isinstance(mod, TransformerLayer) and mod.training and mod.dropout_p > 0:
                return False
        return True

    def capture(self, batchData, label):
        """Capture the step for batches shaped like ``batchData`` (step() does it on demand).  The two warm-up steps torch's
        capture recipe needs run on a snapshot: parameters, optimiser moments and the step count are restored afterwards,
        so the first call to step() performs exactly one update, like every other."""
        self._composite.join()
        self._ones_like(torch.empty(1, len(self.criterion.wPrediction.predictors), device=batchData.device))
        self.optimizer.device_step_counter(True)
        params = [p for g in self.optimizer.param_groups for p in g["params"]]
        with torch.no_grad():
            snap = [(p.detach().clone(), self.optimizer.state[p]["exp_avg"].clone(),
                     self.optimizer.state[p]["exp_avg_sq"].clone()) for p in params]
            step0 = self.optimizer._device_step.clone()
        static_in = batchData.clone()
        static_label = None if label is None else label.clone()
        side = torch.cuda.Stream(device=batchData.device)
        side.wait_stream(torch.cuda.current_stream())
        self._capturing = True
        try:
            try:
                with torch.cuda.stream(side):                 # warm-up on a side stream, as torch's capture recipe asks
                    for _ in range(2):
                        self._eager_step(static_in, static_label)
            finally:
                # whatever the warm-up did -- completed, or raised half-way -- its updates are taken back, so that a caller
                # who falls back to the eager step does not apply it on top of them
                torch.cuda.current_stream().wait_stream(side)
                with torch.no_grad():
                    for p, (w, m, v) in zip(params, snap):
                        p.copy_(w)
                        self.optimizer.state[p]["exp_avg"].copy_(m)
                        self.optimizer.state[p]["exp_avg_sq"].copy_(v)
                    self.optimizer._device_step.copy_(step0)
                    self.optimizer.zero_grad()
                torch.cuda.synchronize(batchData.device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="relaxed"):
                out = self._eager_step(static_in, static_label)
        finally:
            self._capturing = False
            del self._done_events[:]                          # eager steps before the capture have completed (synchronize above)
        self._captured = (self._graph_key(batchData), g, static_in, static_label, out)

    def step(self, batchData, label, negatives=None):
        """One optimiser step; returns (losses (1,K), accuracies (1,K)) of this step as fresh tensors."""
        out = self._step(batchData, label, negatives)
        self._steps += 1
        if self.check_errors_every > 0 and self._steps % self.check_errors_every == 0 and batchData.is_cuda:
            from . import ops
            with torch.cuda.device(batchData.device):
                ops.check_device_errors()
        return out

    def _step(self, batchData, label, negatives=None):
        if self.graph == "auto":
            ok = (negatives is None and batchData.is_cuda and not self.allreduce._active() and torch.is_grad_enabled()
                  and self._graph_safe())
            if not ok:
                return self._eager_step(batchData, label, negatives)
            import time
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            t0, w0 = time.perf_counter(), getattr(self, "wait_seconds", 0.0)
            out = self._eager_step(batchData, label)
            host = time.perf_counter() - t0 - (getattr(self, "wait_seconds", 0.0) - w0)     # issuing, not waiting for step n-2
            e1.record()
            self._probe.append((host, e0, e1))
            if len(self._probe) >= self.AUTO_PROBE_STEPS:
                e1.synchronize()
                steady = self._probe[2:]                    # the first calls pay allocator / lazy-initialisation costs
                host_ms = 1e3 * sum(h for h, _, _ in steady) / len(steady)
                gpu_ms = sum(a.elapsed_time(b) for _, a, b in steady) / len(steady)
                # Replay costs the host one launch but the GPU 6 % more than the eagerly issued streams (round 4: 3.10 vs 2.93 ms,
                # DESIGN.md 4.9), and the composite step needs 0.5 ms of host time where round 3's eager step needed 1.7-2.4:
                # the graph is taken only when the host could not keep up otherwise.
                self.graph = host_ms > 0.85 * gpu_ms
                self.launch_decision = {"host_ms_per_step": round(host_ms, 3), "gpu_ms_per_step": round(gpu_ms, 3),
                                        "graph": self.graph}
                self._probe = []
            return out
        use_graph = (self.graph and negatives is None and batchData.is_cuda and not self.allreduce._active()
                     and torch.is_grad_enabled() and self._graph_safe())
        if not use_graph:
            if self._captured is not None:                 # leave capturable mode consistently
                self.optimizer.device_step_counter(False)
                self._captured = None
            return self._eager_step(batchData, label, negatives)
        if self._captured is None or self._captured[0] != self._graph_key(batchData):
            try:
                self.capture(batchData, label)
            except Exception as e:                          # a runtime that cannot capture this step: stay eager, say so once
                import warnings
                warnings.warn(f"cpc_audio_amd.Trainer: HIP graph capture failed ({e!r}); running the step eagerly")
                self.graph = False
                self._captured = None
                self.optimizer.device_step_counter(False)
                return self._eager_step(batchData, label, negatives)
        _, g, static_in, static_label, out = self._captured
        if batchData.data_ptr() != static_in.data_ptr():
            static_in.copy_(batchData)
        if static_label is not None and label is not None and label.data_ptr() != static_label.data_ptr():
            static_label.copy_(label)
        g.replay()
        return out[0].clone(), out[1].clone()      # the graph's static outputs are overwritten by the next replay
