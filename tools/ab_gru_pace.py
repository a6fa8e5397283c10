"""A/B of the poll pacing of the persistent recurrence (cpc_set_gru_poll_pacing: -1 default steering, -(16 up + clean) other
steering constants, >= 0 pinned), the recurrence calls alone.  usage: python tools/ab_gru_pace.py [B]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpc_audio_amd import _lib
from cpc_audio_amd._lib import ptr as P
from tools.bench_gru import timeit

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S = 128
dev = torch.device("cuda:0")
lib = _lib.get()
torch.manual_seed(0)
shapes = [(768, 256), (768, 256), (768,), (768,)] * 2
plist = [(torch.randn(s, device=dev) / 16.0) for s in shapes]
x = torch.randn(B, S, 256, device=dev); dy = torch.randn(B, S, 256, device=dev)
sizes = (ctypes.c_long * 3)(); lib.check(lib.cpc_gru_layout(B, S, 2, sizes))
saved = torch.empty(sizes[0], device=dev); fscr = torch.empty(sizes[1], device=dev); bscr = torch.empty(sizes[2], device=dev)
y = torch.empty(B, S, 256, device=dev); hN = torch.empty(2, B, 256, device=dev); dx = torch.empty(B, S, 256, device=dev)
grads = [torch.empty_like(t) for t in plist]
parr = (ctypes.c_void_p * 8)(*[P(t) for t in plist]); garr = (ctypes.c_void_p * 8)(*[P(t) for t in grads])
st = torch.cuda.current_stream().cuda_stream
fwd = lambda: lib.check(lib.cpc_gru_forward(P(x), None, parr, P(saved), P(fscr), P(y), P(hN), B, S, 2, st))
bwd = lambda: lib.check(lib.cpc_gru_backward(P(x), None, parr, P(saved), P(y), P(dy), P(bscr), P(dx), garr, B, S, 2, st))
enc = lambda up, clean: -(16 * up + clean)
for rnd in range(2):
    for name, v in (("default(4,4)", -1), ("(2,4)", enc(2, 4)), ("(2,2)", enc(2, 2)), ("(4,2)", enc(4, 2)), ("(1,1)", enc(1, 1)), ("(8,4)", enc(8, 4)),
                    ("(4,8)", enc(4, 8)), ("(3,1)", enc(3, 1))):
        lib.check(lib.cpc_set_gru_poll_pacing(v, v))
        f, b = timeit(fwd, 30), timeit(bwd, 30)
        print(f"round {rnd} pace {name:13s}: forward call {f * 1e3:7.1f} us  backward call {b * 1e3:7.1f} us  flags {lib.cpc_device_error_flags(1)}", flush=True)
lib.cpc_set_gru_poll_pacing(-1, -1)
