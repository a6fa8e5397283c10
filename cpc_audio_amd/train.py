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
                    sizeWindow=20480, downsampling=160, mode=None, rnnMode="linear", transformerDropout=0.1):
    return CPCUnsupersivedCriterion(nPredicts, hiddenGar, hiddenEncoder, negativeSamplingExt, mode=mode,
                                    rnnMode=rnnMode, dropout=False, sizeInputSeq=sizeWindow // downsampling,
                                    transformerDropout=transformerDropout)


def load_flat_params(model, criterion, params):
    """Load a dict keyed like the reference's state dicts (gEncoder.*, gAR.*, wPrediction.*)."""
    model.load_state_dict({k: v for k, v in params.items() if not k.startswith("wPrediction")}, strict=True)
    criterion.load_state_dict({k: v for k, v in params.items() if k.startswith("wPrediction")}, strict=True)


class Trainer:
    def __init__(self, model, criterion, lr=2e-4, betas=(0.9, 0.999), eps=1e-8):
        self.model, self.criterion = model, criterion
        params = list(criterion.parameters()) + list(model.parameters())      # train.py:332
        self.optimizer = Adam(params, lr=lr, betas=betas, eps=eps)      # train.py:335-337; one launch per step on the GPU
        enc = {id(p) for p in model.gEncoder.parameters()} if hasattr(model, "gEncoder") else set()
        self.allreduce = FlatGradAllReduce(params, early=[p for p in params if id(p) not in enc] if enc else None)
        from . import ops
        self.ctx = ops.StepContext(overlap=True)
        self.ctx.pre_encoder_backward.append(self.allreduce.begin)

    def _ones_like(self, t):
        o = getattr(self, "_ones", None)
        if o is None or o.shape != t.shape or o.device != t.device or o.dtype != t.dtype:
            o = self._ones = torch.ones_like(t)
        return o

    def step(self, batchData, label, negatives=None):
        # the overlap state (side streams, events, launches held back) lives on this Trainer's StepContext: two Trainers
        # on two threads / devices do not share any
        try:
            with self.ctx as step:
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
        return allLosses.detach(), allAcc.detach()
