"""CPU restatement of the reference's transformer layer (cpc/transformers.py) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product path (cpc_audio_amd/) never does.  Pinned against the imported reference by
oracle/make_golden_transformer.py (outputs <= 2e-6, gradients <= 1e-5 relative), which also writes the
fixtures tests/golden/transformer_*.npz.

The layer is used by the reference in two places of BASELINE.json config 4:
  * as the auto-regressive network, ``--arMode transformer``: buildTransformerAR(hiddenEncoder, 1,
    sizeWindow // 160, abspos)  (cpc/feature_loader.py:138-141), i.e. ONE layer, sequence 128, d_model 256;
  * as the K prediction networks, ``--rnnMode transformer``: K x buildTransformerAR(dimOutputEncoder, 1,
    sizeInputSeq, False) with sizeInputSeq = 128 - K (cpc/criterion/criterion.py:82-88, :162-164).

Restated semantics (dropout = 0 / eval: the reference hard-codes dropout 0.1, transformers.py:93, which has
no deterministic counterpart):

  q, k, v = x Wq^T, x Wk^T, x Wv^T                     (bias-free, transformers.py:60-63, 82-84)
  split into h = 8 heads of d_k = d_model / 8          (:73-79)
  score[i, j] = (q_i . k_j + q_i . P[:, S-1-(i-j)]) / sqrt(d_k)   for j <= i, -inf for j > i
        -- the "z trick" (:40-47: prepend a zero column to Q.P, view as (S+1, S), drop the first row) puts
           entry Q.P[i, S-1-(i-j)] at position (i, j) of the lower triangle; the upper triangle is masked
           (:27-31), so P = Krelpos (d_k, S) is indexed by the distance i - j and shared by all heads (:22-25, :66)
  A = softmax_j(score);  o_i = sum_j A[i, j] v_j       (:48-49)
  y  = LayerNorm(x + concat_heads(o) Wo^T)             (:85, :109)
  out = LayerNorm(y + lin2(relu(lin1(y))))             (:97-100, :110), d_ff = 2048
With abspos=True the relative term is absent and sin/cos position embeddings are added to the input once
(:113-127); the reference's CPC configs use abspos=False.
"""
import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

N_HEADS = 8
D_FF = 2048


def layer_param_shapes(d_model: int = 256, size_seq: int = 128, abspos: bool = False, prefix: str = ""):
    """State-dict keys/shapes of one reference TransformerLayer (parameters only, in module order)."""
    dk = d_model // N_HEADS
    s = {}
    for n in ("Wo", "Wk", "Wq", "Wv"):
        s[f"{prefix}multihead.{n}.weight"] = (d_model, d_model)
    if not abspos:
        s[f"{prefix}multihead.Att.Krelpos"] = (dk, size_seq)
    s[f"{prefix}ln_multihead.weight"] = (d_model,)
    s[f"{prefix}ln_multihead.bias"] = (d_model,)
    s[f"{prefix}ffnetwork.lin1.weight"] = (D_FF, d_model)
    s[f"{prefix}ffnetwork.lin1.bias"] = (D_FF,)
    s[f"{prefix}ffnetwork.lin2.weight"] = (d_model, D_FF)
    s[f"{prefix}ffnetwork.lin2.bias"] = (d_model,)
    s[f"{prefix}ln_ffnetwork.weight"] = (d_model,)
    s[f"{prefix}ln_ffnetwork.bias"] = (d_model,)
    return s


def make_layer_params(seed: int = 0, d_model: int = 256, size_seq: int = 128, abspos: bool = False,
                      prefix: str = "") -> Dict[str, Tensor]:
    """Deterministic parameters: N(0,1)/sqrt(fan_in) matrices, 0.1 N(0,1) biases, LayerNorm 1 + 0.1 N / 0.1 N,
    Krelpos N(0,1)/sqrt(d_k)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, shp in layer_param_shapes(d_model, size_seq, abspos, prefix).items():
        r = torch.randn(shp, generator=g)
        if k.endswith("Krelpos"):
            out[k] = r / math.sqrt(shp[0])
        elif k.endswith(".weight") and len(shp) == 2:
            out[k] = r / math.sqrt(shp[1])
        elif "ln_" in k and k.endswith(".weight"):
            out[k] = 1.0 + 0.1 * r
        else:
            out[k] = 0.1 * r
    return out


def static_position_embedding(seqlen: int, d_model: int) -> Tensor:
    """transformers.py:113-124."""
    pos = torch.arange(0., seqlen).unsqueeze(1).repeat(1, d_model)
    dim = torch.arange(0., d_model).unsqueeze(0).repeat(seqlen, 1)
    pos = pos * torch.exp(-math.log(10000) * (2 * (dim // 2) / d_model))
    pos[:, 0::2] = torch.sin(pos[:, 0::2])
    pos[:, 1::2] = torch.cos(pos[:, 1::2])
    return pos


def attention_probabilities(q: Tensor, k: Tensor, krelpos: Optional[Tensor]) -> Tensor:
    """q, k: (N, S, d_k) -> A (N, S, S), causal, with the relative-position term indexed by distance."""
    n, s, dk = q.shape
    score = torch.bmm(q, k.transpose(1, 2))
    if krelpos is not None:
        assert krelpos.shape == (dk, s), "the layer is built for one fixed sequence length (transformers.py:22)"
        qp = q.matmul(krelpos)                                   # (N, S, S): column c <-> distance S-1-c
        i = torch.arange(s).view(s, 1)
        j = torch.arange(s).view(1, s)
        col = (s - 1 - (i - j)).clamp(0, s - 1)                  # upper triangle: any valid column (masked below)
        score = score + torch.gather(qp, 2, col.expand(n, s, s))
    score = score / math.sqrt(dk)
    future = torch.triu(torch.ones(s, s, dtype=torch.bool), diagonal=1)
    score = score.masked_fill(future, float("-inf"))
    return torch.softmax(score, dim=2)


def layer_forward(p: Dict[str, Tensor], x: Tensor, prefix: str = "", collect: Optional[dict] = None,
                  relu_override: Optional[Tensor] = None, attn_keep: Optional[Tensor] = None,
                  ffn_keep: Optional[Tensor] = None) -> Tensor:
    """One TransformerLayer, x (B, S, d_model) -> (B, S, d_model).
    ``relu_override``: optional bool (B, S, d_ff) = [hidden_device > 0]; used for the derivative of the
    feed-forward ReLU at numerically tied pre-activations only (|x| < 1e-5, cpc_oracle._ReluTieAware).
    ``attn_keep`` (B*heads, S, S) / ``ffn_keep`` (B, S, d_ff): the training-mode dropout of transformers.py:18,50 and
    :93,100 with EXPLICIT masks (entries 0 or 1 / (1 - p), what nn.Dropout multiplies by), so that a device run can be
    reproduced with the masks it drew."""
    b, s, d = x.shape
    h, dk = N_HEADS, d // N_HEADS

    def heads(t):
        return t.view(b, s, h, dk).transpose(1, 2).reshape(b * h, s, dk)

    q = heads(x @ p[f"{prefix}multihead.Wq.weight"].t())
    k = heads(x @ p[f"{prefix}multihead.Wk.weight"].t())
    v = heads(x @ p[f"{prefix}multihead.Wv.weight"].t())
    a = attention_probabilities(q, k, p.get(f"{prefix}multihead.Att.Krelpos"))
    if attn_keep is not None:
        a = a * attn_keep                                         # self.drop(A), transformers.py:50
    o = torch.bmm(a, v).view(b, h, s, dk).transpose(1, 2).reshape(b, s, d)
    att = o @ p[f"{prefix}multihead.Wo.weight"].t()
    y = F.layer_norm(x + att, (d,), p[f"{prefix}ln_multihead.weight"], p[f"{prefix}ln_multihead.bias"], 1e-5)
    pre = y @ p[f"{prefix}ffnetwork.lin1.weight"].t() + p[f"{prefix}ffnetwork.lin1.bias"]
    if relu_override is not None:
        from .cpc_oracle import _ReluTieAware
        hid = _ReluTieAware.apply(pre, relu_override, 1e-5)
    else:
        hid = torch.relu(pre)
    if ffn_keep is not None:
        hid = hid * ffn_keep                                      # self.drop(self.relu(...)), transformers.py:100
    ff = hid @ p[f"{prefix}ffnetwork.lin2.weight"].t() + p[f"{prefix}ffnetwork.lin2.bias"]
    out = F.layer_norm(y + ff, (d,), p[f"{prefix}ln_ffnetwork.weight"], p[f"{prefix}ln_ffnetwork.bias"], 1e-5)
    if collect is not None:
        collect.update(a=a, o=o, y=y, hid=hid)
    return out


def ar_forward(p: Dict[str, Tensor], z: Tensor, n_layers: int = 1, abspos: bool = False, prefix: str = "",
               relu_override=None) -> Tensor:
    """buildTransformerAR(...) as an nn.Sequential (transformers.py:130-139): optional position embedding at
    index 0, then the layers; state-dict keys are '<index>.<layer key>'."""
    x = z
    first = 0
    if abspos:
        x = x + static_position_embedding(z.size(1), z.size(2)).unsqueeze(0)
        first = 1
    for i in range(n_layers):
        x = layer_forward(p, x, prefix=f"{prefix}{first + i}.",
                          relu_override=None if relu_override is None else relu_override[i])
    return x
