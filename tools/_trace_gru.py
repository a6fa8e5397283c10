
import ctypes, os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpc_audio_amd import _lib
from cpc_audio_amd._lib import ptr as P
lib = _lib.get()
raw = ctypes.CDLL(_lib.LIB_PATH) if hasattr(_lib, "LIB_PATH") else lib._lib
B, S = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 128
dev = torch.device("cuda:0")
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
def run():
    lib.check(lib.cpc_gru_forward(P(x), None, parr, P(saved), P(fscr), P(y), P(hN), B, S, 2, st))
    lib.check(lib.cpc_gru_backward(P(x), None, parr, P(saved), P(y), P(dy), P(bscr), P(dx), garr, B, S, 2, st))
for _ in range(3): run()
torch.cuda.synchronize()
tr = torch.zeros(4 * 2 * S * 8, dtype=torch.int64, device=dev)
f = raw.cpc_debug_set_trace; f.argtypes = [ctypes.c_void_p]; f.restype = ctypes.c_int
assert f(tr.data_ptr()) == 0
run(); torch.cuda.synchronize()
assert f(None) == 0
T = tr.cpu().view(4, 2, S, 8).double()
names = ["fwd L0", "fwd L1", "bwd L0", "bwd L1"]
for k in range(4):
    for grp in range(2):
        a = T[k, grp]
        sl = a[20:110]
        if sl[:, 0].abs().sum() == 0: continue
        poll = (sl[:, 1] - sl[:, 0]).mean(); mf = (sl[:, 2] - sl[:, 1]).mean(); bar = (sl[:, 3] - sl[:, 2]).mean()
        step = (sl[1:, 0] - sl[:-1, 0]).abs().mean()
        line = f"{names[k]} waves {'0-3' if grp == 0 else '4-7'}: step {step:.0f} cyc  poll {poll:.0f}  mfma+lds {mf:.0f}  barrier-wait {bar:.0f}  polls/step {(sl[-1,6]-sl[0,6]).abs()/(len(sl)-1):.2f}"
        if grp == 0:
            g = (sl[:, 5] - sl[:, 4]).mean(); lag = (sl[:, 4] - sl[:, 3]).mean()
            line += f"  | gate: after-barrier lag {lag:.0f}  math+xstore {g:.0f}"
        print(line)
