"""Host logic of the harness (no GPU): schedulers with the reference's known answers
(cpc/utils/unit_tests.py:20-61), synthetic loader, checkpoint discovery."""
import json
import os

import pytest
import torch

from cpc_audio_amd import harness as H


def _opt():
    module = torch.nn.Linear(1, 1)
    return torch.optim.SGD(list(module.parameters()), lr=1)


def test_ramp_scheduler_known_answers():
    opt = _opt()
    sch = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda e: H.ramp_scheduling_function(3, e))
    opt.step()
    assert opt.param_groups[0]["lr"] == pytest.approx(1 / 3)
    sch.step()
    assert opt.param_groups[0]["lr"] == pytest.approx(2 / 3)
    sch.step()
    assert opt.param_groups[0]["lr"] == 1
    for _ in range(12):
        sch.step()
        assert opt.param_groups[0]["lr"] == 1


def test_combined_ramp_and_step_known_answers():
    opt = _opt()
    step = torch.optim.lr_scheduler.StepLR(opt, 6, gamma=0.5)
    ramp = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda e: H.ramp_scheduling_function(3, e))
    sch = H.SchedulerCombiner([ramp, step], [0, 3])
    opt.step()
    assert opt.param_groups[0]["lr"] == pytest.approx(1 / 3)
    sch.step()
    assert opt.param_groups[0]["lr"] == pytest.approx(2 / 3)
    sch.step()
    assert opt.param_groups[0]["lr"] == 1
    sch.step()
    for _ in range(3):
        assert opt.param_groups[0]["lr"] == 1
        sch.step()
    assert opt.param_groups[0]["lr"] == 0.5
    with pytest.raises(ValueError):
        H.SchedulerCombiner([ramp], [0, 3])
    with pytest.raises(ValueError):
        H.SchedulerCombiner([ramp, step], [2, 3])


def test_synthetic_loader_shapes_and_determinism():
    a = list(H.SyntheticLoader(3, 4, 640, seed=7))
    b = list(H.SyntheticLoader(3, 4, 640, seed=7))
    assert len(a) == 3 and a[0][0].shape == (4, 1, 640) and a[0][1].shape == (4,)
    assert a[0][1].dtype == torch.long and float(a[0][0].abs().max()) <= 1.0
    assert all(torch.equal(x[0], y[0]) for x, y in zip(a, b))
    assert not torch.equal(a[0][0], a[1][0])


def test_checkpoint_discovery_and_roundtrip(tmp_path):
    from cpc_audio_amd.train import build_criterion, build_model
    d = str(tmp_path)
    assert H.get_checkpoint_data(os.path.join(d, "missing")) is None
    assert H.get_checkpoint_data(d) is None
    model, crit = build_model(), build_criterion()
    opt = torch.optim.Adam(list(crit.parameters()) + list(model.parameters()), lr=2e-4)
    for ep in (0, 5, 10):
        H.save_checkpoint(model.state_dict(), crit.state_dict(), opt.state_dict(), None,
                          os.path.join(d, f"checkpoint_{ep}.pt"))
    H.save_logs({"epoch": [0, 1], "locLoss_train": [torch.zeros(12).numpy()]}, os.path.join(d, "checkpoint_logs.json"))
    with open(os.path.join(d, "checkpoint_args.json"), "w") as f:
        json.dump({"hiddenEncoder": 256, "arMode": "GRU"}, f)
    path, logs, args = H.get_checkpoint_data(d)
    assert path.endswith("checkpoint_10.pt") and logs["epoch"] == [0, 1] and args.arMode == "GRU"
    model2, crit2 = build_model(), build_criterion()
    state = H.load_checkpoint(path, model2, crit2)
    assert set(state.keys()) == {"gEncoder", "cpcCriterion", "optimizer", "best"}
    for (k, v), (k2, v2) in zip(model.state_dict().items(), model2.state_dict().items()):
        assert k == k2 and torch.equal(v, v2)


class _PoolMaker(torch.nn.Module):
    """A stand-in feature maker for the chunking logic: frame f of a chunk = (mean, max) of its 160 samples + the chunk's own
    first sample (so a feature knows which chunk produced it); optionally stateful like a keepHidden autoregressor (adds the
    number of chunks seen so far)."""

    def __init__(self, stateful=False):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(1))
        self.stateful, self.calls, self.batch_sizes = stateful, 0, []
        if stateful:
            self.featureMaker = type("M", (), {"gAR": type("A", (), {"keepHidden": True})()})()

    def getDownsamplingFactor(self):
        return 160

    def forward(self, data):
        x, _ = data
        self.batch_sizes.append(x.shape[0])
        fr = x.shape[2] // 160
        v = x[:, 0, :fr * 160].reshape(x.shape[0], fr, 160)
        out = torch.stack([v.mean(2), v.amax(2), x[:, 0, :1] * (1.0 + torch.arange(fr))], dim=2)
        if self.stateful:
            out = out + self.calls
            self.calls += 1
        return out


def _reference_chunking(maker, seq, strict, max_size_seq, seq_norm):
    """The loop of cpc/feature_loader.py:228-269, restated as the test's oracle (one call per chunk)."""
    n, start, out = seq.size(1), 0, []
    norm = (lambda f: (f - f.mean(1, keepdim=True)) / torch.sqrt(f.var(1, keepdim=True) + 1e-8)) if seq_norm else (lambda f: f)
    while start < n:
        if strict and start + max_size_seq > n:
            break
        out.append(norm(maker((seq[:, start:min(n, start + max_size_seq)].view(1, 1, -1), None))))
        start += max_size_seq
    if strict and start < n:
        f = norm(maker((seq[:, -max_size_seq:].view(1, 1, -1), None)))
        out.append(f[:, -((n - start) // 160):])
    return torch.cat(out, dim=1)


def test_build_feature_batches_equal_chunks_and_cuts_like_the_reference():
    torch.manual_seed(0)
    for n in (6400 * 3 + 1777, 6400 * 4, 6400 + 159, 3000):
        seq = torch.randn(1, n)
        for strict in (False, True):
            for seq_norm in (False, True):
                ref = _reference_chunking(_PoolMaker(), seq, strict, 6400, seq_norm)
                mk = _PoolMaker()
                got = H.build_feature(mk, seq, strict=strict, max_size_seq=6400, seq_norm=seq_norm)
                assert got.shape == ref.shape, (n, strict, got.shape, ref.shape)
                assert torch.allclose(got, ref, atol=2e-5 if seq_norm else 1e-6), (n, strict, seq_norm)
                assert mk.batch_sizes[0] == max(n // 6400, 1)          # all whole chunks in one call
                # a stateful autoregressor sees the chunks one after the other, in file order
                ref_s = _reference_chunking(_PoolMaker(stateful=True), seq, strict, 6400, seq_norm)
                mk_s = _PoolMaker(stateful=True)
                got_s = H.build_feature(mk_s, seq, strict=strict, max_size_seq=6400, seq_norm=seq_norm)
                assert torch.allclose(got_s, ref_s, atol=2e-5 if seq_norm else 1e-6) and set(mk_s.batch_sizes) == {1}
    plan = H.chunk_plan(64000 * 2 + 12345, 64000, True, 160)
    assert plan == [(0, 64000, None), (64000, 128000, None), (76345, 140345, 77)]
    assert H.chunk_plan(1000, 64000, True, 160) == [(0, 1000, 6)]


def test_transformer_submodules_run_stand_alone_and_long_layers_match_the_oracle():
    """cpc/transformers.py's classes are drop-in one by one: ScaledDotProductAttention / MultiHeadAttention / FFNetwork compute
    the reference's forward with torch ops when called on their own, and a TransformerLayer the fused HIP kernels do not cover
    (sizeSeq = 400: a 64000-sample feature-extraction chunk, cpc/feature_loader.py:247-266) composes them -- checked against the
    oracle's restatement (oracle/transformer_oracle.py, pinned to the imported reference), values and gradients, with and
    without the relative-position term, also on a sequence shorter than the layer was built for."""
    import torch
    from cpc_audio_amd.transformers import TransformerLayer, buildTransformerAR
    from oracle import transformer_oracle as T
    torch.manual_seed(0)
    for abspos, S_built, S in ((False, 400, 400), (True, 160, 160), (False, 144, 130)):
        ar = buildTransformerAR(256, 1, S_built, abspos, dropout=0.0)
        layer = ar[-1]
        assert isinstance(layer, TransformerLayer) and not layer.fused_train      # (on CUDA, no-grad: HIP forward up to 512 steps)
        x = torch.randn(2, S, 256, requires_grad=True)
        y = ar(x)
        p = {k: v.detach() for k, v in ar.state_dict().items()}
        xr = x.detach().clone().requires_grad_(True)
        if S == S_built:
            yr = T.ar_forward(p, xr, 1, abspos)
            assert (y - yr).abs().max().item() < 1e-5
            g = torch.randn_like(y)
            (y * g).sum().backward()
            (yr * g).sum().backward()
            assert (x.grad - xr.grad).abs().max().item() < 1e-5 * max(1.0, xr.grad.abs().max().item())
        else:
            # a shorter sequence in a longer layer: the distance-indexed relative term and the sliced mask are those of a layer
            # built for the short length whose Krelpos is the LAST S columns of this one (distance d lives in column sizeSeq-1-d)
            q = {k: v for k, v in p.items()}
            q["0.multihead.Att.Krelpos"] = p["0.multihead.Att.Krelpos"][:, S_built - S:]
            yr = T.ar_forward(q, xr, 1, abspos)
            assert (y - yr).abs().max().item() < 1e-5
    # the sub-modules on their own
    layer = TransformerLayer(sizeSeq=24, dmodel=64, dff=128, dropout=0.0, nheads=4)
    x = torch.randn(3, 24, 64)
    att = layer.multihead(x, x, x)
    assert att.shape == (3, 24, 64) and torch.isfinite(att).all()
    assert layer.ffnetwork(x).shape == (3, 24, 64)
    q = torch.randn(6, 24, 16)
    a = layer.multihead.Att(q, q, q)
    # causal: step 0 attends to itself only
    assert torch.allclose(a[:, 0], q[:, 0], atol=1e-6)


def test_reference_options_outside_the_hip_geometry_run_on_the_modules_own_torch_ops():
    """cpc/model.py:73-80 (normMode batchNorm / instanceNorm / ID, any width) and :168-176 (arMode LSTM -- the reference's argparse
    default -- / RNN): constructed with the reference's signatures, the reference's state-dict keys, and served by torch ops on
    any device (``.hip`` is False; the HIP kernels cover the 256-channel ChannelNorm encoder and the 256 -> 256 GRU).  The
    layerNorm encoder at another width is checked against the oracle; the LSTM carries its (h, c) state like the reference."""
    import torch
    from cpc_audio_amd.model import CPCAR, CPCEncoder, CPCModel
    from oracle import cpc_oracle as O
    torch.manual_seed(0)
    wave = O.make_waveform(2, 3200, seed=3)
    for mode in ("batchNorm", "instanceNorm", "ID", "layerNorm"):
        enc = CPCEncoder(64, mode)
        assert not enc.hip and enc.getDimOutput() == 64 and enc.DOWNSAMPLING == 160
        y = enc(wave)
        assert y.shape == (2, 64, 20) and torch.isfinite(y).all() and (y >= 0).all()
        keys = set(enc.state_dict())
        assert {f"conv{i}.weight" for i in range(5)} <= keys and {f"conv{i}.bias" for i in range(5)} <= keys
        if mode == "layerNorm":
            assert enc.state_dict()["batchNorm3.weight"].shape == (1, 64, 1)
            p = {f"gEncoder.{k}": v for k, v in enc.state_dict().items()}
            assert (y - O.encoder_forward(p, wave)).abs().max().item() < 1e-5
    assert CPCEncoder(256, "layerNorm").hip and not CPCEncoder(256, "batchNorm").hip and not CPCEncoder(512).hip
    x = torch.randn(2, 20, 64)
    for mode, cell in (("LSTM", torch.nn.LSTM), ("RNN", torch.nn.RNN), ("GRU", torch.nn.GRU)):
        ar = CPCAR(64, 32, True, 2, mode=mode, reverse=(mode == "RNN"))
        assert not ar.hip and isinstance(ar.baseNet, cell) and ar.getDimOutput() == 32
        c1 = ar(x)
        assert c1.shape == (2, 20, 32)
        h = ar.hidden
        assert (isinstance(h, tuple) and len(h) == 2 and not h[0].requires_grad) if mode == "LSTM" else not h.requires_grad
        c2 = ar(x)                                          # starts from the carried state: differs from the first call
        assert not torch.allclose(c1, c2)
        ref = cell(64, 32, num_layers=2, batch_first=True)
        ref.load_state_dict(ar.baseNet.state_dict())
        xin = torch.flip(x, [1]) if mode == "RNN" else x
        want, _ = ref(xin)
        assert torch.allclose(c1, torch.flip(want, [1]) if mode == "RNN" else want, atol=1e-6)
    assert CPCAR(256, 256, False, 2).hip and not CPCAR(256, 256, False, 2, mode="LSTM").hip
    c, z, _ = CPCModel(CPCEncoder(64, "ID"), CPCAR(64, 64, False, 1, mode="LSTM"))(wave, None)
    assert c.shape == z.shape == (2, 20, 64)


def test_issue_guard_refuses_a_second_thread_on_the_same_stream(monkeypatch):
    """ops.issuing_step (the claim on (device, current stream) a step is issued under), with torch.cuda stubbed out: the same
    thread may nest, another thread on the same stream is refused, another stream is fine, and the claim is gone afterwards.
    (tests/test_gpu_modules.py has the same on real streams.)"""
    import threading
    import types
    import torch
    from cpc_audio_amd import ops
    stream_of = {}                                           # thread ident -> fake stream handle

    def current_stream(dev=None):
        return types.SimpleNamespace(cuda_stream=stream_of.get(threading.get_ident(), 7))
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    monkeypatch.setattr(torch.cuda, "current_stream", current_stream)
    seen = {}

    def other(key, own_stream):
        if own_stream:
            stream_of[threading.get_ident()] = 11
        try:
            with ops.issuing_step(0):
                seen[key] = "ok"
        except RuntimeError as e:
            seen[key] = str(e)

    with ops.issuing_step(0):
        with ops.issuing_step(0):                            # nesting on one thread
            for key, own in (("same", False), ("own", True)):
                t = threading.Thread(target=other, args=(key, own))
                t.start()
                t.join()
        assert ops._issuing                                  # the outer claim is still held
    assert "its own stream" in seen["same"] and seen["own"] == "ok", seen
    assert not ops._issuing
    t = threading.Thread(target=other, args=("after", False))
    t.start()
    t.join()
    assert seen["after"] == "ok" and not ops._issuing
