"""Data-parallel gradient exchange: one process per GPU, a flat RCCL SUM all-reduce per step in two buckets.

The reference wraps model and criterion in torch.nn.DataParallel (cpc/train.py:372-375):
every step it broadcasts all parameters, gathers c and z to GPU 0, re-scatters them into
the criterion and reduce-adds gradients to GPU 0.  The path shards naturally over
sequences (negatives are drawn inside each replica's sub-batch, criterion.py:176-184), so
here each rank runs the whole step on its own sub-batch and the only collective is a SUM
all-reduce of the 2,893,056 gradient values (11.57 MB) over xGMI.  SUM, not mean: the
reference sums the per-replica losses (train.py:85, ``allLosses.sum()`` over the gathered
(nGPU, K) tensor), see SURVEY.md T8.

Buckets of one persistent flat buffer (two, or three with ``mid``: see FlatGradAllReduce), because the backward pass ends with the encoder (about a third of the
step) and everything else -- prediction heads and auto-regressive network, 55 % of the values -- is final before it
starts: ``begin()`` (hooked to the start of the encoder's backward by the package's train loops, ops.
pre_encoder_backward) sends that ``early`` bucket off on the side stream while the encoder's backward runs, and
the call after backward() sends the rest, waits for both and writes the sums back (``abort()`` after a failed step).  Every rank issues the two
collectives in the same order; without ``begin()`` the call reduces the whole buffer at once.
"""
import torch
import torch.distributed as dist


class _HostStagedWork:
    """Work handle of a GPU bucket reduced through the host (a gloo group: RCCL cannot put two ranks on one device, and a
    gloo-only cluster has no device collectives).  It walks the stream dependencies of the RCCL path instead of replacing them
    with a blocking ``t.cpu()``: the device-to-host copy into a pinned buffer is queued on the ISSUING stream (non-blocking, an
    event behind it), a single worker thread -- one per FlatGradAllReduce, so that every rank issues its collectives in the same
    order -- waits for that event, runs the gloo all-reduce on the pinned buffer and queues the copy back on a copy stream with
    an event behind it; ``wait()`` makes the CURRENT stream wait for that event (the host blocks only until the copy back has
    been queued, as ``Work.wait()`` of a process group does until the collective has been enqueued)."""

    def __init__(self, future):
        self._future = future

    def wait(self):
        done = self._future.result()
        torch.cuda.current_stream(done[1]).wait_event(done[0])
        return True


class _IssuedOnStream:
    """Handle of a collective issued as a synchronous op on the stream that was current: ordering is that stream's (and the event
    recorded behind it), there is nothing else to wait for."""

    def wait(self):
        return True


class FlatGradAllReduce:
    """Flattens the gradients of ``params`` into one persistent buffer and all-reduces it (SUM).
    ``early``: the subset of ``params`` whose gradients are complete when ``begin()`` is called (heads + auto-regressive
    network: final when the encoder's backward starts).  ``mid``: a subset that becomes final during the encoder's backward on a
    stream of its own -- the weight gradients of conv layers 2..4, done on the weight-gradient stream ~0.5 ms before the
    backward ends; the caller of ``__call__`` says how a stream waits for them (``mid_wait``).  What is left -- conv0, conv1
    and the 256-element bias / norm gradients, 0.53 M values -- is the one collective a step exposes."""

    def __init__(self, params, early=None, mid=None, group=None):
        params = [p for p in params if p.requires_grad]
        ids = {id(p) for p in (early or [])}
        mids = {id(p) for p in (mid or [])} - ids
        self.early = [p for p in params if id(p) in ids]
        self.mid = [p for p in params if id(p) in mids]
        self.late = [p for p in params if id(p) not in ids and id(p) not in mids]
        self.params = self.early + self.mid + self.late      # buffer order: early bucket first, then mid, then late
        self.n_early = sum(p.numel() for p in self.early)
        self.n_mid = sum(p.numel() for p in self.mid)
        self.group = group
        self.numel = sum(p.numel() for p in self.params)
        self.buf = None
        self.views = None
        self._pending = None                                  # work handle of the early bucket
        self._pending_event = None
        self._stage = {}                                      # host-staged path: (data_ptr, numel) of a bucket -> pinned buffer
        self._worker = None                                   # ... its one worker thread (ordered collectives)
        self._copy_stream = None
        self._side = None                                     # the stream begin() issued the early bucket on
        self.single_rank_too = False                          # tests: run the collectives in a 1-rank group as well

    def _active(self):
        return (dist.is_available() and dist.is_initialized()
                and (dist.get_world_size(self.group) > 1 or self.single_rank_too))

    def _reduce(self, t, async_op=False):
        """SUM all-reduce of ``t`` in place.  RCCL cannot put two ranks on one device and a gloo-only cluster has no device
        collectives: a GPU tensor in a gloo group is staged through the host by a worker thread (_HostStagedWork: copies
        queued on the issuing stream, nothing blocks the host); the buckets, their order and what waits for what stay as
        on the RCCL path."""
        if t.is_cuda and dist.get_backend(self.group) == "gloo":
            work = self._reduce_host_staged(t)
            if async_op:
                return work
            work.wait()
            return None
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)

    def _reduce_host_staged(self, t):
        from concurrent.futures import ThreadPoolExecutor
        if self._worker is None:
            self._worker = ThreadPoolExecutor(max_workers=1, thread_name_prefix="cpc-gloo-stage")
            self._copy_stream = torch.cuda.Stream(device=t.device)
        key = (t.data_ptr(), t.numel())
        host = self._stage.get(key)
        if host is None:
            host = self._stage[key] = torch.empty(t.numel(), dtype=t.dtype, pin_memory=True)
        cur = torch.cuda.current_stream(t.device)
        host.copy_(t.view(-1), non_blocking=True)              # behind everything the issuing stream holds for this bucket
        staged = torch.cuda.Event()
        staged.record(cur)
        group, copy_stream, dev = self.group, self._copy_stream, t.device

        def run():
            with torch.cuda.device(dev):
                staged.synchronize()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
                with torch.cuda.stream(copy_stream):
                    t.view(-1).copy_(host, non_blocking=True)
                    back = torch.cuda.Event()
                    back.record(copy_stream)
            return back, dev

        return _HostStagedWork(self._worker.submit(run))

    def _views(self, params):
        ref = self.params[0]
        if self.buf is None or self.buf.device != ref.device:
            self.buf = torch.empty(self.numel, device=ref.device, dtype=ref.dtype)
            self.views, off = {}, 0
            for p in self.params:
                n = p.numel()
                self.views[id(p)] = self.buf[off:off + n].view_as(p)
                off += n
        return [self.views[id(p)] for p in params]

    def _pack(self, params):
        views = self._views(params)
        # (a gradient that already IS its slice of the buffer -- train.Trainer's composite step writes them there -- needs no copy)
        have = [(v, p.grad) for v, p in zip(views, params) if p.grad is not None and p.grad.data_ptr() != v.data_ptr()]
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        for v, p in zip(views, params):
            if p.grad is None:
                v.zero_()

    def begin(self, step=None):
        """Start reducing the ``early`` bucket.  ``step``: the ops.StepContext of the train loop (this is hooked to its
        ``pre_encoder_backward``).  On a GPU the packing and the collective are ordered after everything the context's
        side stream holds -- the heads' gradient is formed there -- and after the current stream's work up to now; the
        current stream does not wait for them."""
        if not self._active() or not self.early or self._pending is not None:
            return
        ref = self.early[0]
        bucket = None
        if ref.is_cuda and step is not None:
            main, side = torch.cuda.current_stream(ref.device), step.side_stream(ref.device)
            ev = torch.cuda.Event()
            ev.record(main)
            side.wait_event(ev)
            for e in step.wgrad_events:              # the recurrence's gradients are formed on the weight-gradient stream
                side.wait_event(e)
            with torch.cuda.stream(side):
                self._pack(self.early)
                bucket = self.buf[:self.n_early]
                # On RCCL the collective is issued as a SYNCHRONOUS op of the side stream: torch.distributed then launches it on
                # the current stream itself (the side stream: nothing of the main stream waits for it) instead of on the process
                # group's internal stream behind an event wait.  That internal stream shares a hardware queue with whichever of
                # the step's streams the runtime mapped there, and its wait -- queued at begin(), released only when the heads'
                # gradient is done -- then holds back every packet queued behind it (head-of-line blocking: measured +0.3 ms per
                # step with the default 4 hardware queues, nothing with 6).  The host-staged gloo path stays asynchronous.
                direct = dist.get_backend(self.group) != "gloo"
                work = self._reduce(bucket, async_op=not direct)
                self._pending = _IssuedOnStream() if direct else work
                done = torch.cuda.Event()
                done.record(side)
            self._pending_event = done                       # what the current stream waits for in __call__
            self._side = side
            for p in self.early:
                if p.grad is not None:
                    p.grad.record_stream(side)
        else:
            self._pack(self.early)
            bucket = self.buf[:self.n_early]
            self._pending = self._reduce(bucket, async_op=True)
            self._pending_event = None
            self._side = None

    def abort(self):
        """A step raised after begin(): let the early bucket's collective finish (every rank issued it; dropping the
        handle would leave the next step waiting on a stale one and copying last step's sums into .grad) and forget it."""
        pending, self._pending = self._pending, None
        event, self._pending_event = self._pending_event, None
        if pending is not None:
            try:
                pending.wait()
                if event is not None:                          # ... and the next step's writers of the flat buffer come behind it
                    torch.cuda.current_stream().wait_event(event)
            except Exception:                                  # the group may be the thing that failed
                pass

    def __call__(self, mid_wait=None):
        """Finish the exchange: every gradient is final on the current stream when this is called.
        ``mid_wait`` (GPU, after ``begin()``): ``mid_wait(stream)`` makes ``stream`` wait for the ``mid`` parameters'
        gradients -- they became final earlier, on another stream; the mid bucket is then issued on the side stream behind the
        early one (both run beside the rest of the backward) and only the late bucket on the current stream.  Without it mid
        and late -- contiguous in the buffer -- go out as one collective.  Every rank must make the same choice."""
        if not self._active():
            return
        if self._pending is None:                             # begin() was not called: one collective
            self._pack(self.params)
            self._reduce(self.buf)
        else:
            cur_ev = []
            lo = self.n_early
            if self.mid and mid_wait is not None and self._side is not None:
                side = self._side
                mid_wait(side)
                with torch.cuda.stream(side):                 # (FIFO behind the early bucket: one communicator, one order)
                    self._pack(self.mid)
                    direct = dist.get_backend(self.group) != "gloo"
                    work = self._reduce(self.buf[lo:lo + self.n_mid], async_op=not direct)
                    if not direct:
                        work.wait()                           # host-staged: the side stream waits for the copy back
                    done = torch.cuda.Event()
                    done.record(side)
                cur_ev.append(done)
                for p in self.mid:
                    if p.grad is not None:
                        p.grad.record_stream(side)
                lo += self.n_mid
            # the collectives of one communicator in one order on the device as well: the current stream takes up the early
            # (and mid) bucket's completion BEFORE it issues the last one -- they finished long ago, the wait costs nothing
            self._pending.wait()
            if self._pending_event is not None:
                torch.cuda.current_stream().wait_event(self._pending_event)
            for e in cur_ev:
                torch.cuda.current_stream().wait_event(e)
            rest = self.params[len(self.early) + (len(self.mid) if lo > self.n_early else 0):]
            if rest:
                self._pack(rest)
                self._reduce(self.buf[lo:])
            self._pending = self._pending_event = None
        views = self._views(self.params)
        have = [(p.grad, v) for p, v in zip(self.params, views) if p.grad is not None and p.grad.data_ptr() != v.data_ptr()]
        if have:
            torch._foreach_copy_([g for g, _ in have], [v for _, v in have])
        for p, v in zip(self.params, views):
            if p.grad is None:
                p.grad = v.clone()
