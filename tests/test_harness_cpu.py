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
