"""Train-step throughput of the MI355X-native CPC hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]

One "step" = one full optimiser step of the north-star configuration (5-layer conv encoder,
2-layer GRU, K=12 linear InfoNCE heads, 128 negatives): forward + ``allLosses.sum().backward()``
+ gradient SUM all-reduce (N > 1) + Adam, on a synthetic white-noise batch of B x 1 x 20480
fp32 per GPU (1.28 s @ 16 kHz per sequence) that is already resident in HBM.  For N > 1 launch with
``python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...``
(one process per GPU, RCCL).  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

AUDIO_S_PER_SEQ = 20480 / 16000.0
F32_MFMA_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak
BF16_MFMA_PEAK_TFLOPS = 2500.0    # dense bf16 MFMA peak (same guide)
X3_PRODUCTS = 3                   # fp16 MFMAs issued per fp32 product in the split-fp16 conv tiles (gemm_tile.h; bf16 split: 6)
HBM_PEAK_GBPS = 8000.0
# HBM bytes per launch at B = 64 from the PMC counters (profiles/r1_pmc_roofline_kernels.md)
# conv1 (tap-fastest K walk, profiles/r1_pmc_counters_v19.csv): 430.8 MB fetched + 134.5 MB written, against 405 MB
# algorithmic; the plain K walk fetched 671.4 MB (805.9 MB in total).
PMC_TRAFFIC_B64 = {"conv1_fwd": 565.3e6, "conv0_fwd": 276.5e6}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="sequences per GPU (BASELINE.json configs[1]: 64)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def hip_event_time(fn, iters, warm=3):
    """Average duration of fn() in ms, hip events on the stream fn launches on (torch's current)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def roofline_probe(B, dev):
    """Roofline of the dominant kernel, conv_fwd_kernel<128,2> on layer 1 (k8 s4, 256->256): the
    implicit-GEMM + ChannelNorm + ReLU kernel.  Algorithmic work per 1.28 s window (SURVEY.md
    section 8d): 536,870,912 MAC = 1.0737 GFLOP; one launch processes B windows."""
    from cpc_audio_amd import _lib
    from cpc_audio_amd._lib import ptr as P
    lib = _lib.get()
    Lin, k, s, p = 4096, 8, 4, 2
    Lout = 1024
    w = torch.randn(256, 256, k, device=dev) / 45.0
    wp = torch.empty(256 * k * 256 * 3 // 2, device=dev)
    bias, nw, nb = torch.randn(256, device=dev) * 0.1, torch.ones(256, device=dev), torch.zeros(256, device=dev)
    y = torch.empty(B, Lout, 256, device=dev)
    xh = torch.empty_like(y)
    rs = torch.empty(B * Lout, device=dev)
    xamax = torch.zeros(1, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    lib.check(lib.cpc_conv_weight_relayout(P(w), P(wp), k, st))

    # the HBM-bound layer (conv0 + norm + ReLU), which precedes layer 1 in the step and produces its input
    L = 20480
    wave = torch.randn(B, L, device=dev) * 0.1
    w0 = torch.randn(256, 10, device=dev) * 0.3
    L0 = 4096
    y0 = torch.empty(B, L0, 256, device=dev)
    m0, r0 = torch.empty(B * L0, device=dev), torch.empty(B * L0, device=dev)

    def f0():
        lib.check(lib.cpc_conv0_forward(P(wave), P(w0), P(bias), P(nw), P(nb), P(y0), P(m0), P(r0), B, L, st))

    def f():
        lib.check(lib.cpc_conv_gemm_forward(P(y0), P(wp), P(bias), P(nw), P(nb), P(y), P(xh), P(rs), P(xamax), B, Lin, k, s, p, st))

    # Timed on the activations conv0 actually produces for the bench waveform (MFMA power, and with it the
    # sustained clock, depends on the operand values: dense random inputs run ~15 % slower), 20 launches between
    # one pair of hip events on the launching stream.  rocprofv3 shows 0.35-0.36 ms for the same kernel inside the
    # train step (profiles/README.md); event pairs around every single launch add ~50 us of marker overhead.
    f0()
    lib.check(lib.cpc_absmax(P(y0), y0.numel(), P(xamax), st))     # operand bound for the fp16-split mode
    ms = hip_event_time(f, iters=20)
    flops = 2.0 * 536870912 * B
    ach = flops / (ms * 1e-3) / 1e12
    # The kernel runs on the fp16 matrix pipe (same dense peak as bf16: 2.5 PFLOP/s) with operands split into two
    # fp16 pieces: every algorithmic fp32 FLOP costs 3 fp16 MFMA FLOPs, so the roof for ALGORITHMIC FLOP/s is
    # 2500 / 3 = 833 TFLOP/s.
    peak = BF16_MFMA_PEAK_TFLOPS / X3_PRODUCTS
    roof = {"bound": "mfma",
            "kernel": "conv_fwd_kernel<128,2> (encoder layer 1: implicit GEMM on the fp16 pipe with scaled 2-piece split "
                      "fp32 operands (hh+hl+lh), fp32 accumulate, + ChannelNorm + ReLU)",
            "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(ach / peak, 4),
            "traffic": PMC_TRAFFIC_B64["conv1_fwd"] if B == 64 else None,
            "ms_per_launch": round(ms, 4), "flop_per_launch": flops,
            "f16_mfma_TFLOPs": round(ach * X3_PRODUCTS, 1), "f16_mfma_peak": BF16_MFMA_PEAK_TFLOPS,
            "vs_f32_mfma_peak": round(ach / F32_MFMA_PEAK_TFLOPS, 4)}
    # conv0: algorithmic bytes = waveform read + one activation write (+ mean/rstd)
    ms0 = hip_event_time(f0, iters=20)
    byts = B * (L * 4 + L0 * 256 * 4 + 2 * L0 * 4)
    g = byts / (ms0 * 1e-3) / 1e9
    hbm = {"bound": "hbm", "kernel": "conv0_fwd_kernel (conv0 + ChannelNorm + ReLU)", "achieved": round(g, 1),
           "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(g / HBM_PEAK_GBPS, 4),
           "traffic": PMC_TRAFFIC_B64["conv0_fwd"] if B == 64 else None,
           "ms_per_launch": round(ms0, 4), "bytes_per_launch": byts}
    return roof, hbm


def scoring_probe(B, dev):
    """The contrastive score matrix (north star: 'MFMA only for the dense z_{t+k}.W_k.c_t score matrix'):
    cpc_nce_scores_forward = nce_fwd_kernel (one wavefront per window: P[16x256] . Cand^T[256 x (N+K)] on
    v_mfma_f32_16x16x4_f32 with the candidate rows gathered from L2, online log-softmax) + its three small reduction
    launches.  Algorithmic work: 2 * 256 * K * (N + 1) FLOP per window (the K = 12 real heads, N negatives + 1
    positive each); peak = the exact-f32 MFMA rate the kernel issues at."""
    import ctypes
    from cpc_audio_amd import _lib
    from cpc_audio_amd._lib import ptr as P
    lib = _lib.get()
    S, K, N = 128, 12, 128
    W = S - K
    sizes = (ctypes.c_long * 6)()
    lib.check(lib.cpc_nce_layout(B, S, K, N, sizes))
    pred = torch.randn(B, W, K * 256, device=dev)
    z = torch.randn(B, S, 256, device=dev)
    ext = torch.randint(0, B * S, (B, W, N), device=dev, dtype=torch.int32)
    saved = torch.empty(sizes[0], device=dev)
    scratch = torch.empty(sizes[1], device=dev)
    losses, acc = torch.empty(K, device=dev), torch.empty(K, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def f():
        lib.check(lib.cpc_nce_scores_forward(P(pred), P(z), P(ext), P(saved), P(scratch), P(losses), P(acc),
                                             B, S, K, N, st))

    ms = hip_event_time(f, iters=20)
    flops = 2.0 * 256 * K * (N + 1) * B * W
    ach = flops / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": "nce_fwd_kernel (score matrix on exact-f32 MFMAs, gathered candidate rows, online "
                                       "log-softmax) + 3 reduction launches (cpc_nce_scores_forward)",
            "achieved": round(ach, 2), "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach / F32_MFMA_PEAK_TFLOPS, 4), "traffic": None, "ms_per_launch": round(ms, 4),
            "flop_per_launch": flops}


def cpu_baseline():
    """The oracle (CPU port of the reference path) timed on this host: forward + backward + Adam at
    B = 8 (BASELINE.json configs[0]), bounded to ~10-30 s of CPU work."""
    from oracle import cpc_oracle as O
    n = os.cpu_count() or 1
    threads = min(n, 64)
    torch.set_num_threads(threads)
    p = O.make_params(seed=0)
    tr = O.CpuTrainer(p)
    B = 8
    wave = O.make_waveform(B, 20480, seed=1234)
    tr.step(wave)                       # warm-up
    t0 = time.perf_counter()
    steps = 0
    while True:
        tr.step(wave)
        steps += 1
        el = time.perf_counter() - t0
        if el > 12.0 or steps >= 8:
            break
    v = steps * B * AUDIO_S_PER_SEQ / el
    return {"value": round(v, 2), "unit": "audio-s/s", "cores": threads, "kind": "port",
            "sample": f"{steps} train steps (fwd+bwd+Adam) of B=8x20480 fp32 on {threads} host threads, "
                      f"{el:.1f} s, oracle/cpc_oracle.CpuTrainer"}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an AMD GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # nccl == RCCL on ROCm
    if a.gpus != world and rank == 0:
        print(f"warning: --gpus {a.gpus} but WORLD_SIZE={world}", file=sys.stderr)

    from cpc_audio_amd.train import Trainer, build_criterion, build_model

    B = a.batch
    torch.manual_seed(0)                       # random-init weights (the reference's default initialisers),
    model, crit = build_model().to(dev), build_criterion().to(dev)   # identical on every rank
    trainer = Trainer(model, crit)
    g = torch.Generator().manual_seed(1234 + rank)
    wave = (0.1 * torch.randn(B, 1, 20480, generator=g)).clamp_(-1, 1).to(dev)
    label = torch.zeros(B, dtype=torch.long, device=dev)
    torch.manual_seed(99 + rank)

    for _ in range(a.warmup):
        trainer.step(wave, label)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        losses, _ = trainer.step(wave, label)
    sync()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    loss0 = float(losses.mean().item())

    out = None
    if rank == 0:
        value = world * B * AUDIO_S_PER_SEQ * a.steps / el
        out = {
            "metric": "audio-seconds/sec (train step)", "value": round(value, 1), "unit": "audio-s/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1000.0 * el / a.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (conv GEMMs and the GRU forward recurrence: fp32 operands scaled and split into 2 fp16 pieces, "
                     "3 fp16 MFMAs per product; GRU backward recurrence and InfoNCE scores: f32 MFMAs; other GEMMs: 3 bf16 "
                     "pieces, 6 bf16 MFMAs per product; fp32 accumulate; fp32-level accuracy)",
            "data": "synthetic white noise 0.1*N(0,1) clamped to [-1,1], resident in HBM; random-init weights",
            "config": {"workload": "default CPC train step (conv encoder + 2-layer GRU + K=12 InfoNCE, 128 negatives), "
                                   f"{world}x{B}x20480 fp32 (BASELINE.json configs[1] at fp32)",
                       "per_gpu_batch": B, "global_batch": world * B, "window": 20480, "parallelism": f"dp{world}",
                       "optimizer": "Adam(lr=2e-4)", "loss_mean_over_heads": round(loss0, 5)},
        }
        if world == 1:
            try:
                roof, hbm = roofline_probe(B, dev)
                out["roofline"] = roof
                out["roofline_hbm_layer"] = hbm
                out["roofline_scoring"] = scoring_probe(B, dev)
            except Exception as e:       # never lose the bench line over the probe
                out["roofline"] = {"error": repr(e)}
            if not a.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
