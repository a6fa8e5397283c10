"""Train / validation / checkpoint harness around the MI355X hot path.

Mirrors the semantics of the reference's loop so the HIP modules are usable end to end without the
CUDA-only ``cpc/train.py`` script:

  * ``train_epoch`` / ``val_epoch``     cpc/train.py:64-119, :122-155  (loss-sum backward, per-head loss and
                                         accuracy averaged over the steps; log keys ``locLoss_train``,
                                         ``locAcc_train``, ``locLoss_val``, ``locAcc_val``, ``iter``)
  * ``run``                             cpc/train.py:158-222  (epoch loop, ``logs["epoch"]``, best-accuracy
                                         snapshot, ``{path}_{epoch}.pt`` + ``{path}_logs.json``)
  * ``save_checkpoint`` / ``get_checkpoint_data`` / ``load_checkpoint``
                                         cpc/feature_loader.py:100-121, :201-209.  Unlike the reference at this
                                         tag ``run`` also WRITES ``checkpoint_args.json``, so
                                         resuming actually works.
  * ``ramp_scheduling_function`` / ``SchedulerCombiner``   cpc/utils/misc.py:77-121
  * ``FeatureModule`` / ``build_feature`` cpc/feature_loader.py:15-38, :221-269 (chunked inference with
                                         carried GRU state), taking a waveform tensor instead of a file path
                                         (torchaudio is not part of this environment).

Differences by design: logs are accumulated ON DEVICE and synchronised once per ``logging_step`` instead of
a device->host copy every step (train.py:98-99); multi-GPU is one process per GPU with one SUM all-reduce
(``dist.FlatGradAllReduce``) instead of ``nn.DataParallel``.
"""
import argparse
import json
import os
import time
from bisect import bisect_left
from copy import deepcopy

import numpy as np
import torch

from . import ops

from .dist import FlatGradAllReduce


# --------------------------------------------------------------------------- schedulers (misc.py:77-121)
def ramp_scheduling_function(n_epoch_ramp, epoch):
    if epoch >= n_epoch_ramp:
        return 1
    return (epoch + 1) / n_epoch_ramp


class SchedulerCombiner:
    """Applies a list of learning-rate schedulers sequentially (cpc/utils/misc.py:84-121)."""

    def __init__(self, scheduler_list, activation_step, curr_step=0):
        if len(scheduler_list) != len(activation_step):
            raise ValueError("The number of scheduler must be the same as the number of activation step")
        if activation_step[0] > curr_step:
            raise ValueError("The first activation step cannot be higher than the current step.")
        self.scheduler_list = scheduler_list
        self.activation_step = deepcopy(activation_step)
        self.curr_step = curr_step

    def step(self):
        self.curr_step += 1
        index = bisect_left(self.activation_step, self.curr_step) - 1
        for i in reversed(range(index, len(self.scheduler_list))):
            self.scheduler_list[i].step()


def build_scheduler(optimizer, scheduler_step=-1, scheduler_ramp=None):
    """cpc/train.py:345-366: StepLR(gamma=0.5) optionally preceded by a linear ramp."""
    scheduler = None
    if scheduler_step > 0:
        scheduler = torch.optim.lr_scheduler.StepLR(optimizer, scheduler_step, gamma=0.5)
    if scheduler_ramp is not None:
        n_epoch = scheduler_ramp
        ramp = torch.optim.lr_scheduler.LambdaLR(
            optimizer, lr_lambda=lambda epoch: ramp_scheduling_function(n_epoch, epoch))
        scheduler = ramp if scheduler is None else SchedulerCombiner([ramp, scheduler], [0, scheduler_ramp])
    return scheduler


# --------------------------------------------------------------------------- synthetic data
class SyntheticLoader:
    """Yields ``n_batches`` of (wave (B,1,L) fp32, label (B,) int64): white noise 0.1*N(0,1) clamped to
    [-1,1] -- the shape cpc/dataset.py:185-202 produces."""

    def __init__(self, n_batches, batch_size, size_window=20480, seed=1234, device="cpu"):
        self.n_batches, self.batch_size, self.size_window = n_batches, batch_size, size_window
        self.seed, self.device = seed, device

    def __len__(self):
        return self.n_batches

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed)
        for _ in range(self.n_batches):
            wave = (0.1 * torch.randn(self.batch_size, 1, self.size_window, generator=g)).clamp_(-1, 1)
            yield wave.to(self.device), torch.zeros(self.batch_size, dtype=torch.long, device=self.device)


# --------------------------------------------------------------------------- epoch loops
def _to_device(t, device):
    return t if t.device == device else t.to(device, non_blocking=True)


def train_epoch(loader, model, criterion, optimizer, scheduler=None, logging_step=1000, allreduce=None,
                verbose=False):
    """One pass over ``loader``; returns {"locLoss_train", "locAcc_train", "iter"} (numpy, averaged)."""
    model.train()
    criterion.train()
    device = next(model.parameters()).device
    sum_loss = sum_acc = None
    n_iter, t0, n_ex = 0, time.perf_counter(), 0
    ones = None
    step_ctx = ops.StepContext(overlap=True)
    if allreduce is not None:
        step_ctx.pre_encoder_backward.append(allreduce.begin)
    in_flight = []                                        # at most two steps ahead of the GPU (train.Trainer.MAX_IN_FLIGHT)
    for step, (batch, label) in enumerate(loader):
        batch, label = _to_device(batch, device), _to_device(label, device)
        n_ex += batch.size(0)
        if device.type == "cuda" and len(in_flight) >= 2:
            in_flight.pop(0).synchronize()
        try:
            with step_ctx as sc:                          # side streams for the dz path / weight gradients (ops.StepContext)
                c_feature, encoded, label = model(batch, label)
                all_losses, all_acc = criterion(c_feature, encoded, label)
                if ones is None or ones.shape != all_losses.shape or ones.device != all_losses.device:
                    ones = torch.ones_like(all_losses)
                torch.autograd.backward([all_losses], [ones])  # = all_losses.sum().backward() (train.py:85-87), 3 kernels less
                sc.wait()
        except BaseException:
            if allreduce is not None:
                allreduce.abort()
            raise
        if allreduce is not None:
            allreduce()
        optimizer.step()
        optimizer.zero_grad()
        if device.type == "cuda":
            in_flight.append(torch.cuda.Event())
            in_flight[-1].record()
        with torch.no_grad():                             # accumulate on device, no per-step sync
            l, a = all_losses.detach().mean(dim=0), all_acc.mean(dim=0)
            sum_loss = l if sum_loss is None else sum_loss + l
            sum_acc = a if sum_acc is None else sum_acc + a
        n_iter += 1
        if verbose and (step + 1) % logging_step == 0:
            el = time.perf_counter() - t0
            print(f"Update {step + 1}: {1000.0 * el / logging_step:.1f} ms per batch, "
                  f"{1000.0 * el / n_ex:.2f} ms / example, loss {float((sum_loss / n_iter).mean()):.4f}")
            t0, n_ex = time.perf_counter(), 0
    if scheduler is not None:
        scheduler.step()
    if n_iter == 0:
        return {"iter": 0}
    if device.type == "cuda":
        with torch.cuda.device(device):
            ops.check_device_errors()       # the averages below synchronise anyway: the place to look at the device flags
    return {"locLoss_train": (sum_loss / n_iter).cpu().numpy(), "locAcc_train": (sum_acc / n_iter).cpu().numpy(),
            "iter": n_iter}


def val_epoch(loader, model, criterion):
    """cpc/train.py:122-155: forward only, under no_grad."""
    model.eval()
    criterion.eval()
    device = next(model.parameters()).device
    sum_loss = sum_acc = None
    n_iter = 0
    for batch, label in loader:
        batch, label = _to_device(batch, device), _to_device(label, device)
        with torch.no_grad():
            c_feature, encoded, label = model(batch, label)
            all_losses, all_acc = criterion(c_feature, encoded, label)
            l, a = all_losses.mean(dim=0), all_acc.mean(dim=0)
            sum_loss = l if sum_loss is None else sum_loss + l
            sum_acc = a if sum_acc is None else sum_acc + a
        n_iter += 1
    if n_iter == 0:
        return {"iter": 0}
    return {"locLoss_val": (sum_loss / n_iter).cpu().numpy(), "locAcc_val": (sum_acc / n_iter).cpu().numpy(),
            "iter": n_iter}


# --------------------------------------------------------------------------- checkpoints
def save_checkpoint(model_state, criterion_state, optimizer_state, best_state, path_checkpoint):
    """cpc/feature_loader.py:201-209: same dict keys."""
    torch.save({"gEncoder": model_state, "cpcCriterion": criterion_state, "optimizer": optimizer_state,
                "best": best_state}, path_checkpoint)


def save_logs(data, path_logs):
    def default(o):
        if hasattr(o, "tolist"):
            return o.tolist()
        raise TypeError(f"not JSON serializable: {type(o)}")
    with open(path_logs, "w") as f:
        json.dump(data, f, indent=2, default=default)


def get_checkpoint_data(path_dir):
    """cpc/feature_loader.py:100-121: newest ``checkpoint_N.pt`` + logs + saved args (or None)."""
    if not os.path.isdir(path_dir):
        return None
    cps = [x for x in os.listdir(path_dir)
           if os.path.splitext(x)[1] == ".pt" and os.path.splitext(x[11:])[0].isdigit()]
    if not cps:
        return None
    cps.sort(key=lambda x: int(os.path.splitext(x[11:])[0]))
    data = os.path.join(path_dir, cps[-1])
    with open(os.path.join(path_dir, "checkpoint_logs.json")) as f:
        logs = json.load(f)
    args_path = os.path.join(path_dir, "checkpoint_args.json")
    args = None
    if os.path.exists(args_path):
        with open(args_path) as f:
            args = argparse.Namespace(**json.load(f))
    return os.path.abspath(data), logs, args


def load_checkpoint(path, model, criterion=None, optimizer=None):
    state = torch.load(path, map_location="cpu")
    model.load_state_dict(state["gEncoder"], strict=False)       # feature_loader.py:180
    if criterion is not None and state.get("cpcCriterion") is not None:
        criterion.load_state_dict(state["cpcCriterion"])
    if optimizer is not None and state.get("optimizer") is not None:
        optimizer.load_state_dict(state["optimizer"])
    return state


def run(train_loader_fn, val_loader_fn, model, criterion, n_epoch, path_checkpoint, optimizer, scheduler=None,
        logs=None, args=None, save_step=5, logging_step=1000, verbose=True):
    """Epoch loop of cpc/train.py:158-222.  ``*_loader_fn()`` return a fresh iterable per epoch.
    ``path_checkpoint`` is the prefix ``<dir>/checkpoint`` (files ``<prefix>_<epoch>.pt``,
    ``<prefix>_logs.json``, ``<dir>/checkpoint_args.json``)."""
    logs = {"epoch": [], "iter": [], "saveStep": save_step, "logging_step": logging_step} if logs is None else logs
    logs.setdefault("epoch", [])
    start_epoch = len(logs["epoch"])
    best_acc, best_state = 0.0, None
    enc = {id(p) for p in model.gEncoder.parameters()} if hasattr(model, "gEncoder") else set()
    every = list(criterion.parameters()) + list(model.parameters())
    allreduce = FlatGradAllReduce(every, early=[p for p in every if id(p) not in enc] if enc else None)
    if path_checkpoint is not None and args is not None:
        os.makedirs(os.path.dirname(path_checkpoint) or ".", exist_ok=True)
        with open(os.path.join(os.path.dirname(path_checkpoint) or ".", "checkpoint_args.json"), "w") as f:
            json.dump(vars(args) if isinstance(args, argparse.Namespace) else dict(args), f, indent=2)
    t0 = time.time()
    for epoch in range(start_epoch, n_epoch):
        loc_train = train_epoch(train_loader_fn(), model, criterion, optimizer, scheduler, logging_step,
                                allreduce, verbose)
        loc_val = val_epoch(val_loader_fn(), model, criterion)
        if verbose:
            print(f"Ran {epoch + 1} epochs in {time.time() - t0:.2f} seconds")
        if "locAcc_val" in loc_val:
            acc = float(loc_val["locAcc_val"].mean())
            if acc > best_acc:                            # (the reference never updates bestAcc)
                best_acc = acc
                best_state = deepcopy({k: v.detach().cpu() for k, v in model.state_dict().items()})
        for key, value in dict(loc_train, **loc_val).items():
            if key not in logs:
                logs[key] = [None for _ in range(epoch)]
            logs[key].append(value.tolist() if isinstance(value, np.ndarray) else value)
        logs["epoch"].append(epoch)
        if path_checkpoint is not None and (epoch % logs.get("saveStep", save_step) == 0 or epoch == n_epoch - 1):
            save_checkpoint(model.state_dict(), criterion.state_dict(), optimizer.state_dict(), best_state,
                            f"{path_checkpoint}_{epoch}.pt")
            save_logs(logs, path_checkpoint + "_logs.json")
    return logs


# --------------------------------------------------------------------------- inference (feature_loader.py)
class FeatureModule(torch.nn.Module):
    """cpc/feature_loader.py:15-38: returns the context features c (or the encoder output z when
    ``get_encoded``) of a (wave, label) pair."""

    def __init__(self, featureMaker, get_encoded, collapse=False):
        super().__init__()
        self.get_encoded = get_encoded
        self.featureMaker = featureMaker
        self.collapse = collapse

    def getDownsamplingFactor(self):
        return self.featureMaker.gEncoder.DOWNSAMPLING

    def forward(self, data):
        batchAudio, label = data
        device = next(self.featureMaker.parameters()).device
        cFeature, encoded, _ = self.featureMaker(batchAudio.to(device), label)
        if self.get_encoded:
            cFeature = encoded
        if self.collapse:
            cFeature = cFeature.contiguous().view(-1, cFeature.size(2))
        return cFeature


def seq_normalization(out):
    """cpc/feature_loader.py:221-225."""
    mean = out.mean(dim=1, keepdim=True)
    var = out.var(dim=1, keepdim=True)
    return (out - mean) / torch.sqrt(var + 1e-08)


def build_feature(feature_maker, seq, strict=False, max_size_seq=64000, seq_norm=False):
    """cpc/feature_loader.py:228-269 on an in-memory waveform ``seq`` of shape (1, n_samples):
    64000-sample chunks, optional strict tail handling and per-chunk time normalisation; with
    ``gAR.keepHidden`` the GRU state is carried across chunks (cpc/eval/ABX.py:170).  Returns
    (1, n_frames, feature_dim) on the CPU."""
    size_seq = seq.size(1)
    start, out = 0, []
    while start < size_seq:
        if strict and start + max_size_seq > size_seq:
            break
        end = min(size_seq, start + max_size_seq)
        sub = seq[:, start:end].reshape(1, 1, -1)
        with torch.no_grad():
            feats = feature_maker((sub, None))
            if seq_norm:
                feats = seq_normalization(feats)
        out.append(feats.detach().cpu())
        start += max_size_seq
    if strict and start < size_seq:
        sub = seq[:, -max_size_seq:].reshape(1, 1, -1)
        with torch.no_grad():
            feats = feature_maker((sub, None))
            if seq_norm:
                feats = seq_normalization(feats)
        delta = (size_seq - start) // feature_maker.getDownsamplingFactor()
        out.append(feats[:, -delta:].detach().cpu())
    return torch.cat(out, dim=1)
