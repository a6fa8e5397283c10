"""Socket power and shader clock while layer 1's forward DMA kernel runs back to back, with zero and with ReLU'd random
operands (is the kernel running against the power budget?).  Samples `rocm-smi` from a thread.
usage (GPU): python tools/power_probe.py [seconds=4]"""
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpc_audio_amd import _lib           # noqa: E402
from cpc_audio_amd._lib import ptr as P  # noqa: E402


def sample(stop, out):
    while not stop.is_set():
        r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True)
        out.append(r.stdout.strip().replace("\n", " | "))
        time.sleep(0.5)


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
    dev = torch.device("cuda:0")
    lib = _lib.get()
    st = torch.cuda.current_stream().cuda_stream
    B, Lout, k, s, p = 64, 1024, 8, 4, 2
    Lin = (Lout - 1) * s + k - 2 * p
    bias = torch.randn(256, device=dev) * 0.1
    nw, nb = torch.ones(256, device=dev), torch.zeros(256, device=dev)
    bound = torch.tensor([4.0], device=dev)
    zeros = torch.zeros(32, device=dev)
    lib.cpc_set_mfma_mode(3)
    yh, xh = torch.empty(B, Lout, 256, device=dev), torch.empty(B, Lout, 256, device=dev)
    rs = torch.empty(B * Lout, device=dev)
    for data in ("idle", "zero", "relu"):
        x = torch.randn(B, Lin, 256, device=dev).clamp_(-4, 4).relu_()
        w = torch.randn(256, 256, k, device=dev) / (16.0 * k ** 0.5)
        if data == "zero":
            x.zero_(); w.zero_()
        xh2 = torch.empty(B, Lin, 256, device=dev)
        lib.check(lib.cpc_h2_encode(P(x), P(xh2), B * Lin, P(bound), st))
        wq = torch.empty(256 * k * 256 + 64, device=dev)
        lib.check(lib.cpc_conv_weight_relayout_h2(P(w), P(wq), k, st))
        torch.cuda.synchronize()
        stop, out = threading.Event(), []
        th = threading.Thread(target=sample, args=(stop, out))
        th.start()
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < secs:
            if data != "idle":
                for _ in range(200):
                    lib.cpc_conv_gemm_forward_h2(P(xh2), P(wq), P(bias), P(nw), P(nb), P(yh), P(xh), P(rs), P(bound), P(bound),
                                                 P(zeros), B, Lin, k, s, p, 256, st)
                torch.cuda.synchronize()
                n += 200
            else:
                time.sleep(0.2)
        dt = time.perf_counter() - t0
        stop.set(); th.join()
        print(f"== {data}: {n} launches in {dt:.2f} s = {1e6 * dt / max(n, 1):.1f} us per launch")
        for o in out[1:6]:
            print("   ", o[:300])


def encoder_loop():
    """The same sampling while the encoder's forward + backward (one stream) and the whole train step run back to back."""
    import ctypes
    from cpc_audio_amd.train import Trainer, build_criterion, build_model
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model, crit = build_model().to(dev), build_criterion().to(dev)
    tr = Trainer(model, crit)
    wave = (0.1 * torch.randn(64, 1, 20480)).clamp_(-1, 1).to(dev)
    label = torch.zeros(64, dtype=torch.long, device=dev)
    for _ in range(5):
        tr.step(wave, label)
    torch.cuda.synchronize()
    stop, out = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, out))
    th.start()
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < secs:
        for _ in range(20):
            tr.step(wave, label)
        torch.cuda.synchronize()
        n += 20
    dt = time.perf_counter() - t0
    stop.set(); th.join()
    print(f"== train step: {n} steps in {dt:.2f} s = {1e3 * dt / n:.3f} ms per step")
    for o in out[1:7]:
        print("   ", o[:300])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "step":
        encoder_loop()
    else:
        main()
