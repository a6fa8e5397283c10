"""s_memtime stamps of two ping-pong iterations (k-steps 40, 41) of one workgroup of layer 1's forward DMA kernel, from a
library built with -DCPC_DMA_TIMING:
    hipcc <build.py flags> -DCPC_DMA_TIMING -c cpc_audio_amd/csrc/conv_dma.hip -o /tmp/conv_dma_t.o, linked with the other
    objects into tools/_bin/libcpc_timing.so (tools/build_timing_lib.sh); run with CPC_HIP_LIB=tools/_bin/libcpc_timing.so.
Stamps per iteration: 0 top of the load slot, 1 reads + DMA issued, 2 lgkmcnt/vmcnt waits done, 3 barrier passed (multiply
slot starts), 4 the 24 MFMAs issued, 5 barrier passed."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpc_audio_amd import _lib           # noqa: E402
from cpc_audio_amd._lib import ptr as P  # noqa: E402


def main():
    pipe = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    dev = torch.device("cuda:0")
    lib = _lib.get()
    raw = ctypes.CDLL(os.environ["CPC_HIP_LIB"])
    st = torch.cuda.current_stream().cuda_stream
    torch.manual_seed(0)
    B, Lout, k, s, p = 64, 1024, 8, 4, 2
    Lin = (Lout - 1) * s + k - 2 * p
    bias = torch.randn(256, device=dev) * 0.1
    nw, nb = torch.ones(256, device=dev), torch.zeros(256, device=dev)
    bound = torch.tensor([4.0], device=dev)
    zeros = torch.zeros(32, device=dev)
    x = torch.randn(B, Lin, 256, device=dev).clamp_(-4, 4).relu_()
    xh2 = torch.empty(B, Lin, 256, device=dev)
    lib.cpc_set_mfma_mode(3)
    lib.check(lib.cpc_h2_encode(P(x), P(xh2), B * Lin, P(bound), st))
    w = torch.randn(256, 256, k, device=dev) / (16.0 * k ** 0.5)
    wq = torch.empty(256 * k * 256 + 64, device=dev)
    lib.check(lib.cpc_conv_weight_relayout_h2(P(w), P(wq), k, st))
    yh = torch.empty(B, Lout, 256, device=dev)
    xh = torch.empty(B, Lout, 256, device=dev)
    rs = torch.empty(B * Lout, device=dev)
    lib.cpc_set_dma_pipeline(pipe)
    for _ in range(5):
        lib.check(lib.cpc_conv_gemm_forward_h2(P(xh2), P(wq), P(bias), P(nw), P(nb), P(yh), P(xh), P(rs), P(bound),
                                               P(bound), P(zeros), B, Lin, k, s, p, 256, st))
    torch.cuda.synchronize()
    host = (ctypes.c_ulonglong * 96)()
    assert raw.cpc_debug_dma_stamps(host) == 0
    t00 = min(host[w_ * 12] for w_ in range(8))
    if pipe == 1:
        print("(generic two-stage loop: stamps = top | vmcnt wait | barrier | DMA issue of the next stage | LDS reads + MFMAs | -)")
    print("wave | iteration 40: t0 reads+dma waits barrier mfma barrier | iteration 41 ...   (clocks, relative to the first stamp)")
    for w_ in range(8):
        v = [host[w_ * 12 + i] - t00 for i in range(12)]
        d = [v[0]] + [v[i] - v[i - 1] for i in range(1, 12)]
        print(f"{w_}  | start {v[0]:6d} | " + " ".join(f"{x_:5d}" for x_ in d[1:6]) + f" | gap {d[6]:5d} | " + " ".join(f"{x_:5d}" for x_ in d[7:12]))


if __name__ == "__main__":
    main()
