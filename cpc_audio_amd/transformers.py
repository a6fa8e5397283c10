"""Drop-in replacement for the reference's cpc/transformers.py (BASELINE.json config 4).

Same class names, constructor signatures, parameter / buffer names (state-dict keys ``multihead.Wo.weight``,
``multihead.Att.Krelpos``, ``multihead.Att.z``, ``multihead.Att.mask``, ``ln_multihead.*``, ``ffnetwork.lin1.*`` ...)
and the same ``buildTransformerAR`` factory (cpc/transformers.py:130-139), so checkpoints load both ways.  A
``TransformerLayer`` of the default geometry runs as ONE fused HIP layer (csrc/transformer.hip); the sub-modules hold its
parameters and, called on their own (or inside a layer the fused kernels do not cover: more than 128 steps, other widths),
compute the reference's forward with torch ops.

Dropout: the reference hard-codes p = 0.1 inside TransformerLayer (cpc/transformers.py:18,93,100).  The HIP layer applies
it in training mode inside its kernels -- Philox4x32-10 keep-masks derived from a 64-bit seed the layer draws per call and
regenerated (not stored) in the backward pass; ``cpc_dropout_keep_mask`` returns the masks of a seed so that the oracle can be
run with the very same ones (tests/test_gpu_transformer.py) -- and ignores it in eval mode.  ``dropout=0`` is the setting with
a parity that does not depend on the random stream.
"""
import math

import torch
import torch.nn as nn

from .ops import TransformerLayerFunction


class ScaledDotProductAttention(nn.Module):
    """cpc/transformers.py:10-49: holds Krelpos (dk, sizeSeq) and the reference's ``z`` / ``mask`` buffers."""

    def __init__(self, sizeSeq, dk, dropout, relpos=False):
        super().__init__()
        self.dropout_p = float(dropout)
        self.relpos = relpos
        self.sizeSeq = sizeSeq
        if relpos:
            self.Krelpos = nn.Parameter(torch.Tensor(dk, sizeSeq))
            self.initmat_(self.Krelpos)
            self.register_buffer("z", torch.zeros(1, sizeSeq, 1))
        mask = torch.tril(torch.ones(sizeSeq, sizeSeq), diagonal=0)
        mask = 1 - mask
        mask[mask == 1] = -float("inf")
        self.register_buffer("mask", mask.unsqueeze(0))

    def initmat_(self, mat, dim=0):
        stdv = 1.0 / math.sqrt(mat.size(dim))
        mat.data.uniform_(-stdv, stdv)

    def forward(self, Q, K, V):
        """Q, K, V (N, S, dk) -> (N, S, dk), cpc/transformers.py:37-49, as torch ops: the stand-alone form of the sub-module
        (a TransformerLayer of the default geometry never calls it: its attention runs inside the fused HIP layer).  The
        reference's relative-position term -- a zero column glued to Q.Krelpos and a reshape -- is the skew
        ``score[i, j] += q_i . Krelpos[:, sizeSeq - 1 - (i - j)]`` for j <= i, written out as an index here."""
        S = Q.size(1)
        if S > self.sizeSeq:
            raise ValueError(f"sequence of {S} steps in an attention built for sizeSeq = {self.sizeSeq}")
        score = torch.bmm(Q, K.transpose(1, 2))
        if self.relpos:
            rel = torch.matmul(Q, self.Krelpos)                               # (N, S, sizeSeq): column c <-> distance sizeSeq-1-c
            steps = torch.arange(S, device=Q.device)
            col = (self.sizeSeq - 1 - (steps.view(S, 1) - steps.view(1, S))).clamp(0, self.sizeSeq - 1)
            score = score + torch.gather(rel, 2, col.unsqueeze(0).expand(Q.size(0), S, S))   # (j > i: masked below)
        A = torch.softmax(score / math.sqrt(K.size(-1)) + self.mask[:, :S, :S], dim=2)
        A = nn.functional.dropout(A, self.dropout_p, self.training)
        return torch.bmm(A, V)


class MultiHeadAttention(nn.Module):
    """cpc/transformers.py:52-85."""

    def __init__(self, sizeSeq, dropout, dmodel, nheads, abspos):
        super().__init__()
        self.Wo = nn.Linear(dmodel, dmodel, bias=False)
        self.Wk = nn.Linear(dmodel, dmodel, bias=False)
        self.Wq = nn.Linear(dmodel, dmodel, bias=False)
        self.Wv = nn.Linear(dmodel, dmodel, bias=False)
        self.nheads = nheads
        self.dk = dmodel // nheads
        self.Att = ScaledDotProductAttention(sizeSeq, self.dk, dropout, not abspos)

    def _heads(self, x):                          # (B, S, h dk) -> (B h, S, dk)
        B, S = x.size(0), x.size(1)
        return x.view(B, S, self.nheads, self.dk).transpose(1, 2).reshape(B * self.nheads, S, self.dk)

    def forward(self, Q, K, V):
        """cpc/transformers.py:79-85 as torch ops (stand-alone use; see ScaledDotProductAttention.forward)."""
        B, S = Q.size(0), Q.size(1)
        y = self.Att(self._heads(self.Wq(Q)), self._heads(self.Wk(K)), self._heads(self.Wv(V)))
        return self.Wo(y.view(B, self.nheads, S, self.dk).transpose(1, 2).reshape(B, S, self.nheads * self.dk))


class FFNetwork(nn.Module):
    """cpc/transformers.py:88-100."""

    def __init__(self, din, dout, dff, dropout):
        super().__init__()
        self.lin1 = nn.Linear(din, dff, bias=True)
        self.lin2 = nn.Linear(dff, dout, bias=True)
        self.relu = nn.ReLU()
        self.dropout_p = float(dropout)

    def forward(self, x):
        """cpc/transformers.py:99-100 as torch ops (stand-alone use)."""
        return self.lin2(nn.functional.dropout(self.relu(self.lin1(x)), self.dropout_p, self.training))


class TransformerLayer(nn.Module):
    """cpc/transformers.py:103-111."""

    def __init__(self, sizeSeq=32, dmodel=512, dff=2048, dropout=0.1, nheads=8, abspos=False):
        super().__init__()
        # The fused HIP layer (csrc/transformer.hip) is built for what BASELINE config 4 uses -- dmodel 256, dff 2048, 8 heads,
        # sequences of at most 128 steps (the 20480-sample training window), forward and backward -- and, forward only (inference:
        # no gradient, no dropout), for layers built for up to 512 steps: the 400 frames of a 64000-sample feature-extraction
        # chunk (cpc/feature_loader.py:247-266).  Anything else -- training at more than 128 steps, another width, more than 512
        # steps -- runs through the sub-modules' torch-op forwards below: correct on any device, differentiable, not the hot path.
        self.fused = dmodel == 256 and dff == 2048 and nheads == 8 and sizeSeq <= 512
        self.fused_train = self.fused and sizeSeq <= 128
        self.sizeSeq = sizeSeq
        self.dropout_p = float(dropout)
        self.multihead = MultiHeadAttention(sizeSeq, dropout, dmodel, nheads, abspos)
        self.ln_multihead = nn.LayerNorm(dmodel)
        self.ffnetwork = FFNetwork(dmodel, dmodel, dff, dropout)
        self.ln_ffnetwork = nn.LayerNorm(dmodel)

    def _needs_autograd(self, x):
        return torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))

    def forward(self, x):
        # beyond 128 steps the kernels are forward-only: a call that must be differentiable, or drops out, composes torch ops
        # (... as does a sequence of another length than the long layer was built for: the torch-op forward slices Krelpos / the mask)
        long_train = self.fused and not self.fused_train and x.is_cuda and (
            self._needs_autograd(x) or (self.training and self.dropout_p > 0) or x.size(1) != self.sizeSeq)
        if not (self.fused and x.is_cuda) or long_train:
            if self.fused_train:
                raise RuntimeError("TransformerLayer: the fused HIP layer needs CUDA tensors (there is no CPU fallback for the hot path)")
            y = self.ln_multihead(x + self.multihead(x, x, x))                 # cpc/transformers.py:109-111
            return self.ln_ffnetwork(y + self.ffnetwork(y))
        # nn.Dropout semantics (transformers.py:18,93): active in training mode only.  The masks of a call derive from one
        # 64-bit seed drawn from torch's CPU generator (reproducible under torch.manual_seed, no device synchronisation).
        p, seed = 0.0, 0
        if self.training and self.dropout_p > 0:
            p = self.dropout_p
            seed = int(torch.empty((), dtype=torch.int64).random_().item()) & 0xFFFFFFFFFFFFFFFF
        m, f = self.multihead, self.ffnetwork
        krel = m.Att.Krelpos if m.Att.relpos else None
        return TransformerLayerFunction.apply(x, p, seed, m.Wo.weight, m.Wk.weight, m.Wq.weight, m.Wv.weight, krel,
                                              self.ln_multihead.weight, self.ln_multihead.bias, f.lin1.weight,
                                              f.lin1.bias, f.lin2.weight, f.lin2.bias, self.ln_ffnetwork.weight,
                                              self.ln_ffnetwork.bias)


class StaticPositionEmbedding(nn.Module):
    """cpc/transformers.py:113-127: x + pe, pe[t, i] = sin(t * w_i) for even i, cos(t * w_i) for odd i, with
    w_i = 10000^(-2 * (i // 2) / dmodel); the table is a (non-trainable) buffer named ``pe`` as in the reference."""

    def __init__(self, seqlen, dmodel):
        super().__init__()
        index = torch.arange(dmodel)
        freq = torch.exp(-math.log(10000) * (2 * (index // 2).float() / dmodel))
        angle = torch.arange(0., seqlen).unsqueeze(1) * freq.unsqueeze(0)
        table = torch.where((index % 2 == 0).unsqueeze(0), torch.sin(angle), torch.cos(angle))
        self.register_buffer("pe", table.unsqueeze(0))

    def forward(self, x):
        return x + self.pe[:, :x.size(1), :]


def buildTransformerAR(dimEncoded, nLayers, sizeSeq, abspos, dropout=0.1):
    """cpc/transformers.py:130-139 (``dropout`` is an addition: the reference's layers always use 0.1)."""
    layerSequence = []
    if abspos:
        layerSequence += [StaticPositionEmbedding(sizeSeq, dimEncoded)]
    layerSequence += [TransformerLayer(sizeSeq=sizeSeq, dmodel=dimEncoded, abspos=abspos, dropout=dropout)
                      for _ in range(nLayers)]
    return nn.Sequential(*layerSequence)
