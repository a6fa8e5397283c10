"""Pin the oracle against the imported reference and write the golden fixtures.

TEST INFRASTRUCTURE ONLY.  Run in the build container (where /root/reference is
mounted):

    python oracle/make_golden.py

For every case it
  1. builds the reference modules (cpc/model.py, cpc/criterion/criterion.py) with
     the deterministic parameter recipe of oracle/cpc_oracle.make_params,
  2. runs the reference forward + ``allLosses.sum().backward()`` (train.py:83-87)
     on the seeded white-noise batch, with torch.manual_seed(idx_seed) set right
     before the criterion so that the two torch.randint draws of sampleClean can
     be replayed and captured,
  3. runs the oracle on the same weights / input / captured indices and ASSERTS
     equality (outputs <= 2e-6, losses <= 1e-5, every parameter gradient to 1e-5
     relative),
  4. writes tests/golden/<case>.npz holding small slices + checksums of the
     REFERENCE's results (data only -- no reference code).

The fixtures are what tests/test_oracle_golden.py checks the oracle against on
machines where the reference is absent (the GPU box).
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import cpc_oracle as O          # noqa: E402
from oracle import ref_import               # noqa: E402

GOLDEN_DIR = os.path.join(os.path.dirname(HERE), "tests", "golden")

# name, batch, wave seed, param seed, head_scale, idx seed, full (store slices) or light
CASES = [
    ("b2_init", 2, 1234, 0, 1.0, 77, True),
    ("b2_hot", 2, 1235, 1, 256.0, 78, True),    # logits O(1): loss leaves ln(129) (T9)
    ("b8_cfg1", 8, 1234, 0, 1.0, 79, False),    # BASELINE.json configs[0]
]


def probe(t: torch.Tensor) -> float:
    """Permutation-sensitive checksum: sum_i t_i * cos(0.37 i + 0.1) in float64."""
    f = t.detach().double().reshape(-1)
    i = torch.arange(f.numel(), dtype=torch.float64)
    return float((f * torch.cos(0.37 * i + 0.1)).sum())


def checksums(t: torch.Tensor):
    f = t.detach().double()
    return [float(f.sum()), float(f.norm()), probe(t)]


def run_reference(params, wave, idx_seed):
    model, crit = ref_import.build_reference(params)
    model.train()
    crit.train()
    acts = []
    hooks = [getattr(model.gEncoder, f"batchNorm{i}").register_forward_hook(
        lambda m, i_, o, acts=acts: acts.append(torch.relu(o).detach())) for i in range(5)]
    c, z, _ = model(wave, torch.zeros(wave.shape[0], dtype=torch.long))
    for h in hooks:
        h.remove()
    z.retain_grad()
    c.retain_grad()
    torch.manual_seed(idx_seed)
    # the reference's own logits: what PredictionNetwork.forward (criterion.py:97-118) returns -- K tensors (B, 1+N, W)
    logits = []
    lh = crit.wPrediction.register_forward_hook(lambda m, i_, o: logits.extend(t.detach().clone() for t in o))
    losses, acc = crit(c, z, None)
    lh.remove()
    losses.sum().backward()
    grads = {}
    for k, v in model.state_dict(keep_vars=True).items():
        grads[k] = v.grad
    for k, v in crit.state_dict(keep_vars=True).items():
        grads[k] = v.grad
    return dict(c=c.detach(), z=z.detach().contiguous(), losses=losses.detach(), acc=acc.detach(),
                grads=grads, dz=z.grad.contiguous(), dc=c.grad, acts=acts, logits=logits)


def main():
    assert ref_import.reference_available(), "needs /root/reference"
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    torch.set_num_threads(8)
    meta = {"torch": torch.__version__, "cases": {}}
    for name, B, wseed, pseed, hscale, iseed, full in CASES:
        params = O.make_params(seed=pseed, head_scale=hscale)
        wave = O.make_waveform(B, 20480, seed=wseed)
        ref = run_reference(params, wave, iseed)
        S, K, N = 128, 12, 128
        W = S - K
        torch.manual_seed(iseed)
        bidx, sidx = O.draw_negative_indices(B, S, W, N)
        ora = O.train_step(params, wave, bidx, sidx)

        # ---- pin: oracle == reference ----
        dz_max = (ora["z"] - ref["z"]).abs().max().item()
        dc_max = (ora["c"] - ref["c"]).abs().max().item()
        dl_max = (ora["losses"] - ref["losses"]).abs().max().item()
        assert dz_max <= 2e-6 and dc_max <= 2e-6, (dz_max, dc_max)
        assert dl_max <= 1e-5, dl_max   # fp32 round-off of a different summation order
        assert torch.equal(ora["acc"], ref["acc"]) or (ora["acc"] - ref["acc"]).abs().max() < 2.0 / (W * B)
        worst = 0.0
        for k, g in ref["grads"].items():
            rel = ((ora["grads"][k] - g).norm() / (g.norm() + 1e-30)).item()
            worst = max(worst, rel)
            assert rel <= 1e-5, (k, rel)
        print(f"[{name}] oracle==reference: |dz|={dz_max:.2e} |dc|={dc_max:.2e} "
              f"|dloss|={dl_max:.2e} worst grad rel={worst:.2e}  losses[0]={ref['losses'][0,0]:.6f}")

        # ---- fixture (reference results, data only) ----
        fx = {
            "losses": ref["losses"].numpy(), "acc": ref["acc"].numpy(),
            "batch_idx": bidx.numpy().astype(np.uint8), "seq_idx": sidx.numpy().astype(np.uint8),
            "z_sums": np.array(checksums(ref["z"])), "c_sums": np.array(checksums(ref["c"])),
            "dz_sums": np.array(checksums(ref["dz"])), "dc_sums": np.array(checksums(ref["dc"])),
            "act_sums": np.array([checksums(a) for a in ref["acts"]]),
            "grad_names": np.array(list(ref["grads"].keys())),
            "grad_sums": np.array([checksums(g) for g in ref["grads"].values()]),
        }
        if full:
            fx["z_slice"] = ref["z"][:, ::16, :].numpy()
            fx["c_slice"] = ref["c"][:, ::16, :].numpy()
            fx["dz_slice"] = ref["dz"][:, ::16, ::4].numpy()
            # logits of heads k=1 and k=12 at 4 time steps: the REFERENCE's own (forward hook on wPrediction in
            # run_reference); the oracle's restatement of them on the reference's c / z is asserted equal first
            ext = O.negative_rows(bidx, sidx, B, S, W, N)
            lo = O.criterion_logits(params, ref["c"], ref["z"], ext)
            lg = ref["logits"]
            assert len(lg) == K and tuple(lg[0].shape) == tuple(lo[0].shape), (len(lg), lg[0].shape, lo[0].shape)
            dlg = max((a - b).abs().max().item() for a, b in zip(lg, lo))
            assert dlg <= 1e-5 * max(1.0, max(a.abs().max().item() for a in lg)), dlg
            print(f"[{name}] oracle logits == reference logits: max|d| = {dlg:.2e}")
            fx["logits_k1"] = lg[0][:, :, [0, 37, 80, 115]].numpy()
            fx["logits_k12"] = lg[11][:, :, [0, 37, 80, 115]].numpy()
            g = ref["grads"]
            fx["g_conv1_w_slice"] = g["gEncoder.conv1.weight"][::32, ::32, :].numpy()
            fx["g_conv0_w"] = g["gEncoder.conv0.weight"].numpy()
            fx["g_whh0_slice"] = g["gAR.baseNet.weight_hh_l0"][::48, ::16].numpy()
            fx["g_head5_slice"] = g["wPrediction.predictors.5.weight"][::16, ::16].numpy()
        path = os.path.join(GOLDEN_DIR, f"{name}.npz")
        np.savez_compressed(path, **fx)
        meta["cases"][name] = dict(batch=B, wave_seed=wseed, param_seed=pseed, head_scale=hscale,
                                   idx_seed=iseed, full=full, bytes=os.path.getsize(path))
    with open(os.path.join(GOLDEN_DIR, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print(json.dumps(meta, indent=1))


if __name__ == "__main__":
    main()
