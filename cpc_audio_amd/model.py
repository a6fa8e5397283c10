"""Drop-in replacements for the hot-path modules of the reference's cpc/model.py.

Same class names, constructor signatures, attributes and state-dict keys
(gEncoder.conv{i}.{weight,bias}, gEncoder.batchNorm{i}.{weight,bias} of shape (1,C,1),
gAR.baseNet.{weight,bias}_{ih,hh}_l{n}), so reference checkpoints load and the
reference's cpc/train.py loop can drive these modules unchanged.
forward/backward run in the HIP kernels of libcpc_hip.so.
"""
import torch
import torch.nn as nn

from .ops import EncoderFunction, GruFunction


class ChannelNorm(nn.Module):
    """cpc/model.py:25-58.  Inside CPCEncoder the normalisation is fused into the conv
    kernels; this module owns the affine parameters (checkpoint keys) and, when called on
    its own, evaluates the same formula with torch ops (not on the train-step hot path)."""

    def __init__(self, numFeatures, epsilon=1e-05, affine=True):
        super().__init__()
        self.epsilon, self.affine, self.p = epsilon, affine, 0
        shape = (1, numFeatures, 1)                      # broadcast over (batch, channel, time): the checkpoint layout
        self.weight = nn.Parameter(torch.ones(shape)) if affine else None
        self.bias = nn.Parameter(torch.zeros(shape)) if affine else None

    def reset_parameters(self):
        if self.affine:
            with torch.no_grad():
                self.weight.fill_(1.0)
                self.bias.zero_()

    def forward(self, x):
        var, mean = torch.var_mean(x, dim=1, keepdim=True)          # unbiased variance over the channels, as the reference
        xhat = (x - mean) * torch.rsqrt(var + self.epsilon)
        return torch.addcmul(self.bias, xhat, self.weight) if self.affine else xhat


class IDModule(nn.Module):
    """cpc/model.py:17-22 (normMode 'ID')."""

    def __init__(self, *args, **kwargs):
        super().__init__()

    def forward(self, x):
        return x


class CPCEncoder(nn.Module):
    """cpc/model.py:61-105: five strided Conv1d + norm + ReLU, downsampling 160.

    conv{i} / batchNorm{i} carry the reference's names, shapes and default initialisation.  The configuration every BASELINE
    config and the reference's defaults use -- 256 channels, normMode 'layerNorm' (ChannelNorm) -- runs cpc_encoder_forward
    (HIP, ``self.hip``); the reference's other options (``batchNorm`` / ``instanceNorm`` / ``ID``, other widths; cpc/model.py:
    73-80) are served by the modules' own torch ops: correct on any device, differentiable, not the hot path."""

    def __init__(self, sizeHidden=512, normMode="layerNorm"):
        super().__init__()
        validModes = ["batchNorm", "instanceNorm", "ID", "layerNorm"]
        if normMode not in validModes:
            raise ValueError(f"Norm mode must be in {validModes}")
        self.hip = normMode == "layerNorm" and sizeHidden == 256
        if normMode == "instanceNorm":
            def normLayer(c): return nn.InstanceNorm1d(c, affine=True)
        elif normMode == "ID":
            normLayer = IDModule
        elif normMode == "layerNorm":
            normLayer = ChannelNorm
        else:
            normLayer = nn.BatchNorm1d
        self.dimEncoded = sizeHidden
        self.conv0 = nn.Conv1d(1, sizeHidden, 10, stride=5, padding=3)
        self.batchNorm0 = normLayer(sizeHidden)
        self.conv1 = nn.Conv1d(sizeHidden, sizeHidden, 8, stride=4, padding=2)
        self.batchNorm1 = normLayer(sizeHidden)
        self.conv2 = nn.Conv1d(sizeHidden, sizeHidden, 4, stride=2, padding=1)
        self.batchNorm2 = normLayer(sizeHidden)
        self.conv3 = nn.Conv1d(sizeHidden, sizeHidden, 4, stride=2, padding=1)
        self.batchNorm3 = normLayer(sizeHidden)
        self.conv4 = nn.Conv1d(sizeHidden, sizeHidden, 4, stride=2, padding=1)
        self.batchNorm4 = normLayer(sizeHidden)
        self.DOWNSAMPLING = 160

    def getDimOutput(self):
        return self.conv4.out_channels

    def _flat_params(self):
        out = []
        for i in range(5):
            conv, norm = getattr(self, f"conv{i}"), getattr(self, f"batchNorm{i}")
            out += [conv.weight, conv.bias, norm.weight, norm.bias]
        return out

    def forward(self, x):
        """(B,1,L) -> (B,C,L/160), as the reference.  The kernels work channels-last, so the
        result is a (B,C,S) VIEW of a contiguous (B,S,C) tensor; CPCModel's permute(0,2,1)
        (model.py:287) therefore yields a contiguous (B,S,C) z at no cost."""
        if not self.hip:
            for i in range(5):                                 # cpc/model.py:99-105 with the modules' own torch ops
                x = torch.relu(getattr(self, f"batchNorm{i}")(getattr(self, f"conv{i}")(x)))
            return x
        z = EncoderFunction.apply(x, *self._flat_params())
        return z.permute(0, 2, 1)


class CPCAR(nn.Module):
    """cpc/model.py:155-204.  For the GRU of the north-star configuration (256 -> 256) baseNet is a torch.nn.GRU used as the
    parameter container (same keys / gate layout) and forward runs cpc_gru_forward (HIP, ``self.hip``).  The reference's other
    autoregressors -- ``LSTM`` (its argparse default, cpc_default_config.py:74), ``RNN``, other widths -- run through baseNet's
    own torch forward: same semantics incl. the carried hidden state, any device, not the hot path."""

    def __init__(self, dimEncoded, dimOutput, keepHidden, nLevelsGRU, mode="GRU", reverse=False):
        super().__init__()
        self.RESIDUAL_STD = 0.1
        self.hip = mode not in ("LSTM", "RNN") and dimEncoded == 256 and dimOutput == 256
        cell = nn.LSTM if mode == "LSTM" else (nn.RNN if mode == "RNN" else nn.GRU)
        self.baseNet = cell(dimEncoded, dimOutput, num_layers=nLevelsGRU, batch_first=True)
        self.hidden = None
        self.keepHidden = keepHidden
        self.reverse = reverse

    def getDimOutput(self):
        return self.baseNet.hidden_size

    def _flat_params(self):
        out = []
        for l in range(self.baseNet.num_layers):
            out += [getattr(self.baseNet, f"{n}_l{l}") for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
        return out

    def forward(self, x):
        """(B, S, 256) -> (B, S, 256).  In reverse mode the sequence is processed back to front and handed back in its
        original order (cpc/model.py:185-204); the final hidden state is kept for the next call when keepHidden is set."""
        flip = (lambda t: torch.flip(t, [1])) if self.reverse else (lambda t: t)
        if not self.hip:                                       # cpc/model.py:185-204 with baseNet's own torch forward
            y, h = self.baseNet(flip(x), self.hidden)
            if self.keepHidden:
                self.hidden = tuple(t.detach() for t in h) if isinstance(h, tuple) else h.detach()
            return flip(y)
        # |h_t| <= 1 holds when the recurrence starts from zero or from one of its OWN final states (a convex combination of
        # tanh outputs and the previous state); a state assigned from outside carries no such bound
        bounded = self.hidden is None or self.hidden is getattr(self, "_own_hidden", None)
        y, h_last = GruFunction.apply(flip(x), self.hidden, *self._flat_params())
        if self.keepHidden:
            self.hidden = self._own_hidden = h_last.detach()
        out = flip(y)
        if bounded:
            out._cpc_abs_bound = 1.0      # read by CPCUnsupersivedCriterion.forward: the a-priori operand bound of its fp16-piece
            #                               GEMMs applies to THIS tensor only (anything derived from it loses the tag)
        return out


class CPCModel(nn.Module):
    """cpc/model.py:276-289."""

    def __init__(self, encoder, AR):
        super().__init__()
        self.gEncoder = encoder
        self.gAR = AR

    def forward(self, batchData, label):
        encodedData = self.gEncoder(batchData).permute(0, 2, 1)
        cFeature = self.gAR(encodedData)
        return cFeature, encodedData, label
