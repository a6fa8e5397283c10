"""Drop-in replacements for the hot-path classes of the reference's cpc/criterion/criterion.py.

Same class names (including the reference's spelling ``CPCUnsupersivedCriterion``,
exported at cpc/criterion/__init__.py:5-6), constructor signatures and state-dict keys
(``wPrediction.predictors.{k}.weight``).  forward/backward of the criterion run in the
fused InfoNCE kernels of libcpc_hip.so; negatives are never materialised.
"""
import math

import torch
import torch.nn as nn

from .ops import InfoNCEFunction, InfoNCEScoresFunction, prepare_negatives

_HEAD_TILE = 16          # prediction heads per call of the score kernels (a wavefront's MFMA tile; ops.head_group walks more)


class _Stacked(torch.autograd.Function):
    """The buffer K equally shaped parameters are views of (one behind the other), as a differentiable function of them:
    its gradient is handed to them block by block."""

    @staticmethod
    def forward(ctx, flat, *parts):
        ctx.shape = tuple(parts[0].shape)
        ctx.set_materialize_grads(False)
        return flat.view_as(flat)

    @staticmethod
    def backward(ctx, g):
        n = len(ctx.needs_input_grad) - 1
        if g is None:
            return (None,) * (n + 1)
        g = g.reshape(n, *ctx.shape)
        return (None,) + tuple(g[k] if ctx.needs_input_grad[k + 1] else None for k in range(n))


def stacked_parameters(owner, key, params, shape=None):
    """``params``: K equally shaped parameters.  -> one tensor of shape ``shape`` (default (K, *p.shape)) holding them one
    behind the other, tied to them for autograd, WITHOUT a per-step ``torch.cat``: the parameters are kept as views of one
    buffer (cached on ``owner`` under ``key``; re-established whenever somebody gave them new storage, e.g. ``.to(device)``;
    in-place updates -- the optimiser, ``load_state_dict`` -- keep it)."""
    cache = owner.__dict__.setdefault("_stacked_cache", {})
    flat = cache.get(key)
    p0 = params[0]
    step = p0.numel() * p0.element_size()
    if (flat is None or flat.device != p0.device or flat.dtype != p0.dtype
            or any(p.data_ptr() != flat.data_ptr() + k * step for k, p in enumerate(params))):
        with torch.no_grad():
            flat = torch.stack([p.detach() for p in params]).contiguous()
            for k, p in enumerate(params):
                p.data = flat[k]
        cache[key] = flat
    out = flat if shape is None else flat.view(shape)
    return _Stacked.apply(out, *params)


# the 13 parameters of a TransformerLayer in the order of the C ABI (cpc_transformer_layer_forward)
_LAYER_PARAMS = (lambda l: l.multihead.Wo.weight, lambda l: l.multihead.Wk.weight, lambda l: l.multihead.Wq.weight,
                 lambda l: l.multihead.Wv.weight, lambda l: l.multihead.Att.Krelpos if l.multihead.Att.relpos else None,
                 lambda l: l.ln_multihead.weight, lambda l: l.ln_multihead.bias, lambda l: l.ffnetwork.lin1.weight,
                 lambda l: l.ffnetwork.lin1.bias, lambda l: l.ffnetwork.lin2.weight, lambda l: l.ffnetwork.lin2.bias,
                 lambda l: l.ln_ffnetwork.weight, lambda l: l.ln_ffnetwork.bias)


class _Equalized(nn.Module):
    """cpc/criterion/custom_layers.py:45-78 as the criterion uses it (``equalized=True``): the wrapped layer (parameter keys
    ``module.weight`` / ``module.bias``) starts from N(0, 1) weights and a zero bias, and its output is multiplied at run time by
    He's constant sqrt(2 / fan_in) (custom_layers.py:33-42), kept in the plain attribute ``weight`` as in the reference."""

    def __init__(self, module):
        super().__init__()
        self.module = module
        with torch.no_grad():
            module.weight.normal_(0, 1)
            if module.bias is not None:
                module.bias.zero_()
        self.equalized = True
        self.weight = math.sqrt(2.0 / module.weight[0].numel())

    def forward(self, x):
        return self.module(x) * self.weight


class FFNetwork(nn.Module):
    """cpc/criterion/criterion.py:11-20 (``--rnnMode ffd``): lin2(drop(relu(lin1 x))) on equalized linear layers."""

    def __init__(self, din, dout, dff, dropout):
        super().__init__()
        self.lin1 = _Equalized(nn.Linear(din, dff, bias=True))
        self.lin2 = _Equalized(nn.Linear(dff, dout, bias=True))
        self.relu = nn.ReLU()
        self.drop = nn.Dropout(dropout)

    def forward(self, x):
        return self.lin2(self.drop(self.relu(self.lin1(x))))


class ShiftedConv(nn.Module):
    """cpc/criterion/criterion.py:23-41 (``--rnnMode conv4 / conv8 / conv12``): a causal convolution over the time axis of a
    (N, S, C) sequence -- kernelSize - 1 zero frames in front, an equalized Conv1d, back to (N, S, C)."""

    def __init__(self, dimOutputAR, dimOutputEncoder, kernelSize):
        super().__init__()
        self.module = _Equalized(nn.Conv1d(dimOutputAR, dimOutputEncoder, kernelSize, padding=0))
        self.kernelSize = kernelSize

    def forward(self, x):
        y = nn.functional.pad(x.transpose(1, 2), (self.kernelSize - 1, 0))
        return self.module(y).transpose(1, 2)


_TORCH_PREDICTORS = {
    # criterion.py:63-81.  nn.RNN is built WITHOUT batch_first there (:64-65) and so walks the batch axis of the (B, W, C)
    # context as its time axis; kept, since a checkpoint trained that way means exactly that
    "RNN": lambda a, e: nn.RNN(a, e),
    "LSTM": lambda a, e: nn.LSTM(a, e, batch_first=True),
    "ffd": lambda a, e: FFNetwork(a, e, e, 0),
    "conv4": lambda a, e: ShiftedConv(a, e, 4),
    "conv8": lambda a, e: ShiftedConv(a, e, 8),
    "conv12": lambda a, e: ShiftedConv(a, e, 12),
}


class PredictionNetwork(nn.Module):
    """cpc/criterion/criterion.py:44-118: K prediction networks.  ``--rnnMode linear`` (the ``else`` branch at
    :89-95, north-star configuration): K bias-free nn.Linear(dimOutputAR, dimOutputEncoder) fused into the
    criterion kernels; ``--rnnMode transformer`` (:82-88, BASELINE.json config 4): K one-layer transformers
    ``buildTransformerAR(dimOutputEncoder, 1, sizeInputSeq, False)`` on the HIP transformer layer
    (``transformerDropout`` is an addition -- the reference's layers always use 0.1, which has no parity).  The reference's
    other choices (:63-81: ``RNN``, ``LSTM``, ``ffd``, ``conv4/8/12``; none in a BASELINE config) are torch modules with the
    reference's parameter names whose predictions the HIP score kernels take as a tensor (``scores_apart``)."""

    def __init__(self, nPredicts, dimOutputAR, dimOutputEncoder, rnnMode=None, dropout=False,
                 sizeInputSeq=116, transformerDropout=0.1):
        super().__init__()
        if dimOutputAR != 256 or dimOutputEncoder != 256:
            raise NotImplementedError("the HIP criterion is built for hiddenGar == hiddenEncoder == 256")
        # (any number of prediction steps: the score tiles hold 16 heads per wavefront, a larger criterion is walked in groups of
        # 16 -- CPCUnsupersivedCriterion._forward_in_head_groups)
        self.predictors = nn.ModuleList()
        self.RESIDUAL_STD = 0.01
        self.dimOutputAR = dimOutputAR
        # criterion.py:59,113-114: nn.Dropout(p=0.5) on every head's prediction (``--dropout``).  While it is active (training
        # mode) the predictions are formed as a tensor of their own -- one library GEMM for the K linear heads, or the transformer
        # predictors -- dropped out by torch and scored by the same kernels the transformer predictors use (scores_apart below);
        # in eval mode it is the identity and the linear heads stay fused in the criterion kernels.
        self.dropout = nn.Dropout(p=0.5) if dropout else None
        self.rnnMode = rnnMode
        for _ in range(nPredicts):
            if rnnMode == "transformer":
                from .transformers import buildTransformerAR
                self.predictors.append(buildTransformerAR(dimOutputEncoder, 1, sizeInputSeq, False,
                                                          dropout=transformerDropout))
            elif rnnMode in _TORCH_PREDICTORS:
                self.predictors.append(_TORCH_PREDICTORS[rnnMode](dimOutputAR, dimOutputEncoder))
            else:
                self.predictors.append(nn.Linear(dimOutputAR, dimOutputEncoder, bias=False))

    @property
    def scores_apart(self):
        """True where the predictions exist as a tensor between the prediction networks and the scores (transformer predictors;
        any predictor with the reference's dropout active): InfoNCEScoresFunction instead of the fused InfoNCEFunction."""
        return (self.rnnMode == "transformer" or self.rnnMode in _TORCH_PREDICTORS
                or (self.dropout is not None and self.training))

    group_predictors = True      # False: the transformer predictors run head by head (the path a mixed set falls back to; tests)

    def predictions(self, c):
        """c (B,W,256) -> (B,W,K*256): head k at columns k*256.. (the layout the score kernels read).  K one-layer transformer
        predictors (the only kind buildTransformerAR(.., 1, .., False) builds) run in lock-step, one launch per kernel for all
        of them (ops.TransformerGroupFunction); the K linear heads as one GEMM on the stacked weight; anything else head by head.
        The reference's dropout (criterion.py:113-114: per head, independent elementwise masks) is one dropout of the whole tensor."""
        pred = self._predictions(c)
        return pred if self.dropout is None else self.dropout(pred)

    def _predictions(self, c):
        from .transformers import TransformerLayer
        layers = [p[0] if isinstance(p, nn.Sequential) and len(p) == 1 else None for p in self.predictors]
        if (self.group_predictors and self.rnnMode == "transformer" and len(layers) > 1
                and all(isinstance(l, TransformerLayer) and l.fused for l in layers)
                and len({(l.dropout_p, l.training, l.multihead.Att.relpos) for l in layers}) == 1 and c.is_cuda):
            from .ops import TransformerGroupFunction
            l0 = layers[0]
            p, seed = 0.0, 0
            if l0.training and l0.dropout_p > 0:
                p = l0.dropout_p
                seed = int(torch.empty((), dtype=torch.int64).random_().item()) & 0xFFFFFFFFFFFFFFFF
            kinds = []
            for i, get in enumerate(_LAYER_PARAMS):
                ps = [get(l) for l in layers]
                kinds.append(None if ps[0] is None else stacked_parameters(self, f"layer{i}", ps))
            return TransformerGroupFunction.apply(c, p, seed, len(layers), *kinds)
        if all(isinstance(p, nn.Linear) and p.bias is None for p in self.predictors):
            return torch.nn.functional.linear(c, self.stacked_weight())
        out = [p(c) for p in self.predictors]
        return torch.cat([o[0] if isinstance(o, tuple) else o for o in out], dim=2)     # (recurrent cells return (y, state))

    def stacked_weight(self):
        """(K*256, 256): the K head weights stacked along the output dimension (stacked_parameters: no per-step cat)."""
        ws = [p.weight for p in self.predictors]
        return stacked_parameters(self, "heads", ws, (len(ws) * ws[0].shape[0], ws[0].shape[1]))

    def forward(self, c, candidates):
        """Reference API (criterion.py:97-118) on materialised candidates; kept for callers
        that use PredictionNetwork directly.  The criterion below does not go through it."""
        assert len(candidates) == len(self.predictors)
        out = []
        for k in range(len(self.predictors)):
            locC = self.predictors[k](c)
            if isinstance(locC, tuple):
                locC = locC[0]
            if self.dropout is not None:
                locC = self.dropout(locC)
            locC = locC.view(locC.size(0), 1, locC.size(1), locC.size(2))
            out.append((locC * candidates[k]).mean(dim=3))
        return out


class BaseCriterion(nn.Module):
    """cpc/criterion/criterion.py:121-127."""

    def warmUp(self):
        return False

    def update(self):
        return


class CPCUnsupersivedCriterion(BaseCriterion):
    """cpc/criterion/criterion.py:139-257."""

    def __init__(self,
                 nPredicts,             # Number of steps
                 dimOutputAR,           # Dimension of G_ar
                 dimOutputEncoder,      # Dimension of the convolutional net
                 negativeSamplingExt,   # Number of negative samples to draw
                 mode=None,
                 rnnMode=False,
                 dropout=False,
                 speakerEmbedding=0,
                 nSpeakers=0,
                 sizeInputSeq=128,
                 transformerDropout=0.1):
        super().__init__()
        if speakerEmbedding > 0:
            raise NotImplementedError("speakerEmbedding is deprecated in the reference "
                                      "(cpc_default_config.py:69-71) and not implemented here")
        self.speakerEmb = None
        self.wPrediction = PredictionNetwork(nPredicts, dimOutputAR, dimOutputEncoder, rnnMode=rnnMode,
                                             dropout=dropout, sizeInputSeq=sizeInputSeq - nPredicts,
                                             transformerDropout=transformerDropout)
        self.nPredicts = nPredicts
        self.negativeSamplingExt = negativeSamplingExt
        # (any number: the kernels walk candidates in 16-wide tiles and mask the padding of the last one, ops.prepare_negatives)
        self.lossCriterion = nn.CrossEntropyLoss()   # kept for API parity; the fused kernel computes it
        if mode not in [None, "reverse"]:
            raise ValueError("Invalid mode")
        self.mode = mode

    def drawNegatives(self, batchSize, seqSize, windowSize, device):
        """The two draws of sampleClean in the reference's order (criterion.py:181-189):
        batchIdx in [0,B) first, then seqIdx in [1,S), both flat in (b,n,t) order."""
        n = self.negativeSamplingExt * windowSize * batchSize
        batchIdx = torch.randint(low=0, high=batchSize, size=(n,), device=device)
        seqIdx = torch.randint(low=1, high=seqSize, size=(n,), device=device)
        return batchIdx, seqIdx

    def negativeRows(self, batchIdx, seqIdx, batchSize, seqSize, windowSize):
        """criterion.py:191-199: row = ((seqIdx + t) mod S) + batchIdx*S, returned as the (B, W, N) int32
        layout the score kernel reads.  Torch reference of what cpc_nce_prepare computes on the device
        (used by tests; forward() calls the kernel)."""
        N = self.negativeSamplingExt
        t = torch.arange(windowSize, device=seqIdx.device).view(1, 1, windowSize)
        s = torch.remainder(seqIdx.view(batchSize, N, windowSize) + t, seqSize)
        ext = s + batchIdx.view(batchSize, N, windowSize) * seqSize
        return ext.permute(0, 2, 1).contiguous().to(torch.int32)

    def prepare_step(self, step, batchSize, seqSize, device, c_bound=None, negatives=None):
        """Everything of this criterion's step that depends on nothing the GPU is still to compute, queued on the step's side
        stream from the START of the step (``step``: the ops.StepContext the train loop has just entered) so that it runs beside
        the encoder: the negative draws with their index preparation (or the preparation of the caller's ``negatives``), and --
        linear heads, ``c_bound`` given: the context network bounds its output a priori, |c| <= c_bound (a GRU: 1) -- the
        operand bounds of the prediction GEMMs, whose reduction otherwise sits on the critical path between the context
        network and the first GEMM.  forward() picks the result up when the shapes match and falls back to doing it in line
        otherwise; calling this is optional."""
        from . import ops
        if (step is None or not step.overlap or torch.device(device).type != "cuda" or self.mode == "reverse"
                or self.nPredicts > _HEAD_TILE):
            return
        K, N = self.nPredicts, self.negativeSamplingExt
        W = seqSize - K
        main, side = torch.cuda.current_stream(device), step.side_stream(device)
        if step.begin is not None:
            side.wait_event(step.begin)
        given = None if negatives is None else id(negatives[0])
        with torch.cuda.stream(side):
            if negatives is None:
                negatives = self.drawNegatives(batchSize, seqSize, W, device)
            ext, perm, row_ptr = prepare_negatives(negatives[0], negatives[1], batchSize, seqSize, K, N)
            saved = None
            if c_bound is not None and not self.wPrediction.scores_apart:
                saved = ops.nce_bounds_into(self.wPrediction.stacked_weight(), c_bound, batchSize, seqSize, K, N)
            ready = torch.cuda.Event()
            ready.record(side)
        step.prepared = {"key": (batchSize, seqSize, K, N, torch.device(device)), "index": (ext, perm, row_ptr),
                         "saved": saved, "ready": ready, "given": given, "c_bound": c_bound}

    def forward(self, cFeature, encodedData, label, negatives=None):
        """-> (losses (1,K), acc (1,K)) as criterion.py:256-257.  ``negatives`` optionally
        supplies (batchIdx, seqIdx) instead of drawing them (parity tests)."""
        if self.mode == "reverse":
            encodedData = torch.flip(encodedData, [1])
            cFeature = torch.flip(cFeature, [1])
        batchSize, seqSize, _ = cFeature.size()
        windowSize = seqSize - self.nPredicts
        from . import ops
        if self.nPredicts > _HEAD_TILE:
            return self._forward_in_head_groups(cFeature, encodedData, negatives)
        step = ops.current()
        prepared, saved = (step.prepared if step is not None else None), None
        if step is not None:
            step.prepared = None
        if (prepared is not None and self.mode != "reverse"
                and prepared["given"] == (None if negatives is None else id(negatives[0])) and prepared["key"] == (batchSize, seqSize, self.nPredicts, self.negativeSamplingExt, cFeature.device)):
            # queued at the start of the step (prepare_step): long finished by now
            torch.cuda.current_stream().wait_event(prepared["ready"])
            ext, perm, row_ptr = prepared["index"]
            saved = prepared["saved"]
            # the workspace carries GEMM operand scales for |c| <= c_bound: taken only if THIS cFeature is the tensor the bounded
            # network returned (CPCAR.forward tags it); a transformed c, or one from a state assigned from outside, has its
            # bounds reduced in line instead (fp16 pieces overflow silently ~8x above the bound)
            tag = getattr(cFeature, "_cpc_abs_bound", None)
            if saved is not None and (tag is None or prepared["c_bound"] is None or tag > prepared["c_bound"]):
                saved = None
            for t in (ext, perm, row_ptr) + (() if saved is None else (saved,)):
                t.record_stream(torch.cuda.current_stream())
        elif negatives is None and step is not None and step.overlap and cFeature.is_cuda:
            # the draws and their index preparation depend on nothing the GPU is still computing (encoder, AR): issued
            # on the side stream they run beside the latency-bound recurrence instead of after it
            main, side = torch.cuda.current_stream(), step.side_stream(cFeature.device)
            # (holding them back until the recurrence starts was measured twice: 4.187 vs 4.162 ms/step in round 1, 3.426 vs
            # 3.373 with the paced polls of round 2 -- they disturb its hand-over more than they cost beside the first conv
            # layers, where the host-side lead puts them)
            if step.begin is not None:
                side.wait_event(step.begin)           # fork from the start of the step (ops.StepContext.__enter__)
            with torch.cuda.stream(side):
                negatives = self.drawNegatives(batchSize, seqSize, windowSize, cFeature.device)
                ext, perm, row_ptr = prepare_negatives(negatives[0], negatives[1], batchSize, seqSize, self.nPredicts,
                                                       self.negativeSamplingExt)
            main.wait_stream(side)
            for t in (ext, perm, row_ptr):
                t.record_stream(main)
        else:
            if negatives is None:
                negatives = self.drawNegatives(batchSize, seqSize, windowSize, cFeature.device)
            ext, perm, row_ptr = prepare_negatives(negatives[0], negatives[1], batchSize, seqSize, self.nPredicts,
                                                   self.negativeSamplingExt)
        if self.wPrediction.scores_apart:
            pred = self.wPrediction.predictions(cFeature[:, :windowSize].contiguous())
            losses, acc = InfoNCEScoresFunction.apply(pred, encodedData, ext, perm, row_ptr, self.negativeSamplingExt)
        else:
            heads = [p.weight for p in self.wPrediction.predictors]
            # dz may be filled late (on the side stream) only if nobody outside this package can read it first: not in
            # mode 'reverse' (the flip above sits between the criterion and the encoder), not behind a foreign network
            defer = step is not None and step.overlap and ops.dz_may_be_deferred(cFeature, encodedData)
            losses, acc = InfoNCEFunction.apply(cFeature, encodedData, self.wPrediction.stacked_weight(), ext, perm,
                                                row_ptr, heads, defer, saved, self.negativeSamplingExt)
        return losses.view(1, -1), acc.view(1, -1)

    def _forward_in_head_groups(self, cFeature, encodedData, negatives):
        """nPredicts > 16 (no BASELINE config; the reference takes any): the same kernels on 16 heads at a time
        (ops.head_group -- every group sees the W = S - nPredicts windows of the whole criterion and the same negatives, head
        k's positive is z[t + k + 1]); the per-head losses and accuracies are independent, autograd adds the groups' dc / dz.
        Single stream, no overlap: not the tuned path."""
        B, S, _ = cFeature.size()
        K, N = self.nPredicts, self.negativeSamplingExt
        W = S - K
        if negatives is None:
            negatives = self.drawNegatives(B, S, W, cFeature.device)
        wp = self.wPrediction
        pred = wp.predictions(cFeature[:, :W].contiguous()) if wp.scores_apart else None
        wall = None if wp.scores_apart else wp.stacked_weight()
        losses, accs = [], []
        for k0 in range(0, K, _HEAD_TILE):
            kg = min(_HEAD_TILE, K - k0)
            ext, perm, row_ptr = prepare_negatives(negatives[0], negatives[1], B, S, kg, N, group=(k0, K))
            if wp.scores_apart:
                l, a = InfoNCEScoresFunction.apply(pred[:, :, k0 * 256:(k0 + kg) * 256], encodedData, ext, perm, row_ptr, N,
                                                   (k0, K))
            else:
                l, a = InfoNCEFunction.apply(cFeature, encodedData, wall[k0 * 256:(k0 + kg) * 256], ext, perm, row_ptr, None,
                                             False, None, N, (k0, K))
            losses.append(l)
            accs.append(a)
        return torch.cat(losses).view(1, -1), torch.cat(accs).view(1, -1)
