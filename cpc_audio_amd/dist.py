"""Data-parallel gradient exchange: one process per GPU, ONE flat RCCL all-reduce per step.

The reference wraps model and criterion in torch.nn.DataParallel (cpc/train.py:372-375):
every step it broadcasts all parameters, gathers c and z to GPU 0, re-scatters them into
the criterion and reduce-adds gradients to GPU 0.  The path shards naturally over
sequences (negatives are drawn inside each replica's sub-batch, criterion.py:176-184), so
here each rank runs the whole step on its own sub-batch and the only collective is a SUM
all-reduce of the 2,893,056 gradient values (11.57 MB) over xGMI.  SUM, not mean: the
reference sums the per-replica losses (train.py:85, ``allLosses.sum()`` over the gathered
(nGPU, K) tensor), see SURVEY.md T8.
"""
import torch
import torch.distributed as dist


class FlatGradAllReduce:
    """Flattens the gradients of ``params`` into one persistent buffer and all-reduces it."""

    def __init__(self, params, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.numel = sum(p.numel() for p in self.params)
        self.buf = None

    def __call__(self):
        if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(self.group) == 1:
            return
        ref = self.params[0]
        if self.buf is None or self.buf.device != ref.device:
            self.buf = torch.empty(self.numel, device=ref.device, dtype=ref.dtype)
        views, off = [], 0
        for p in self.params:
            n = p.numel()
            views.append(self.buf[off:off + n].view_as(p))
            off += n
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in self.params]
        torch._foreach_copy_(views, grads)
        dist.all_reduce(self.buf, op=dist.ReduceOp.SUM, group=self.group)
        for p, v in zip(self.params, views):
            if p.grad is None:
                p.grad = v.clone()
            else:
                p.grad.copy_(v)
