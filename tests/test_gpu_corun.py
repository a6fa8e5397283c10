"""Co-residency regression tests (tools/probe_corun.py).

Round 1 found conv0_bwd_kernel returning wrong partial sums when -- and only when -- a 16-bit-MFMA GEMM kernel shared the
chip with it and the compiler had formed packed fp32 arithmetic (v_pk_fma_f32) in it; it is built without SLP
vectorisation since.  These tests pin that state for every kernel of the library that contains packed fp32 or polls
other workgroups' results and that the train step runs beside the 16-bit-MFMA GEMMs on another stream:

    conv0_bwd_kernel, norm_bwd_kernel, gru_bwd_coef_kernel, the recurrence (persistent and per-step, forward and backward)

each run alone, then again while conv_wgrad_kernel<2> / conv_dgrad_kernel<128,.,2> run on a second stream, compared
bit for bit; plus a short version of tools/stress_overlap.py (whole overlapped train steps against the single-stream
step).  A different compiler, flag set or CPC_HIP_LIB override that brings the corruption back fails here."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    return torch.device("cuda:0")


class _Corunner:
    """conv1's weight gradient and data gradient (fp16-split MFMA kernels) at B = 16, launched on their own stream."""

    def __init__(self, dev):
        from cpc_audio_amd import _lib
        from cpc_audio_amd._lib import ptr as P
        self.lib, self.P = _lib.get(), P
        B, Lin, k, s, p = 16, 4096, 8, 4, 2
        Lout = 1024
        g = torch.Generator(device="cpu").manual_seed(0)
        self.dims = (B, Lin, k, s, p)
        self.dx = (torch.randn(B, Lout, 256, generator=g) * 1e-3).to(dev)
        self.x = torch.relu(torch.randn(B, Lin, 256, generator=g)).to(dev)
        self.w = (torch.randn(256, 256, k, generator=g) / 45).to(dev)
        self.wd = torch.empty(256 * k * 256 * 3 // 2 + 64, device=dev)
        self.dprev = torch.empty(B, Lin, 256, device=dev)
        self.amax = torch.zeros(4, device=dev)
        self.xam = torch.zeros(1, device=dev)
        self.lib.check(self.lib.cpc_absmax(P(self.dx), self.dx.numel(), P(self.amax), None))
        self.lib.check(self.lib.cpc_absmax(P(self.x), self.x.numel(), P(self.xam), None))
        M, K = B * Lout, k * 256
        tiles = 2 * (K // 128)
        S = -(-768 // tiles)
        rows = max(-(-(-(-M // S)) // 32) * 32, 256)
        self.S, self.rows = -(-M // rows), rows
        self.part = torch.empty(self.S * 256 * K, device=dev)
        self.dW = torch.empty_like(self.w)
        self.stream = torch.cuda.Stream()
        torch.cuda.synchronize()

    def wgrad(self):
        B, Lin, k, s, p = self.dims
        P = self.P
        self.lib.check(self.lib.cpc_conv_layer_wgrad(P(self.dx), P(self.x), P(self.part), P(self.dW), P(self.amax), P(self.xam),
                                                     B, Lin, k, s, p, self.S, self.rows, self.stream.cuda_stream))

    def dgrad(self):
        B, Lin, k, s, p = self.dims
        P = self.P
        self.lib.check(self.lib.cpc_conv_layer_dgrad(P(self.dx), P(self.w), P(self.wd), 0, None, None, None, None, P(self.dprev),
                                                     None, None, None, P(self.amax), None, B, Lin, k, s, p,
                                                     self.stream.cuda_stream))

    def launch(self, kind, times):
        for _ in range(times):
            (self.wgrad if kind == "wgrad" else self.dgrad)()


def _check_beside(victim, outputs, corun, rounds=3, lead=2, tail=3):
    """victim(stream) fills ``outputs``; solo reference first, then ``rounds`` runs with the co-runner's launches queued on
    its own stream before (so that it is on the chip when the victim starts) and after the victim's launch."""
    s1 = torch.cuda.Stream()
    torch.cuda.synchronize()
    victim(s1)
    torch.cuda.synchronize()
    ref = [t.clone() for t in outputs]
    victim(s1)
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(ref, outputs)), "victim is not reproducible on its own"
    for kind in ("wgrad", "dgrad"):
        for it in range(rounds):
            for t in outputs:
                t.fill_(float("nan"))
            torch.cuda.synchronize()
            corun.launch(kind, lead)
            victim(s1)
            corun.launch(kind, tail)
            torch.cuda.synchronize()
            bad = [i for i, (a, b) in enumerate(zip(ref, outputs)) if not torch.equal(a, b)]
            assert not bad, f"beside conv_{kind}: outputs {bad} differ from the solo run (round {it})"


def test_conv0_backward_is_bit_exact_beside_the_fp16_gemm_kernels():
    dev = _dev()
    co = _Corunner(dev)
    lib, P = co.lib, co.P
    B, L = 16, 20480
    g = torch.Generator(device="cpu").manual_seed(1)
    wave = (0.1 * torch.randn(B, L, generator=g)).clamp_(-1, 1).to(dev)
    w0 = (torch.randn(256, 10, generator=g) * 0.3).to(dev)
    b0 = (torch.randn(256, generator=g) * 0.1).to(dev)
    nw, nb = torch.ones(256, device=dev), torch.zeros(256, device=dev)
    y0 = torch.empty(B, 4096, 256, device=dev)
    m0, r0 = torch.empty(B * 4096, device=dev), torch.empty(B * 4096, device=dev)
    lib.check(lib.cpc_conv0_forward(P(wave), P(w0), P(b0), P(nw), P(nb), P(y0), P(m0), P(r0), B, L, None))
    dy0 = (torch.randn(B, 4096, 256, generator=g) * 1e-3).to(dev)
    scr = torch.empty(lib.cpc_conv0_backward_scratch_floats(B, L), device=dev)
    grads = [torch.empty(256, 10, device=dev)] + [torch.empty(256, device=dev) for _ in range(3)]

    def victim(st):
        lib.check(lib.cpc_conv0_backward(P(wave), P(w0), P(b0), P(nw), P(nb), P(m0), P(r0), P(dy0), P(scr), P(grads[0]),
                                         P(grads[1]), P(grads[2]), P(grads[3]), B, L, st.cuda_stream))
    _check_beside(victim, grads, co)


def test_conv0_forward_is_bit_exact_beside_the_fp16_gemm_kernels():
    """conv0's forward never runs beside a GEMM inside ONE train step, but it does as soon as two train loops share a device
    (test_two_trainers_on_two_threads...): both of its forms (fp32 output, fp16-piece output)."""
    dev = _dev()
    co = _Corunner(dev)
    lib, P = co.lib, co.P
    B, L = 16, 20480
    g = torch.Generator(device="cpu").manual_seed(2)
    wave = (0.1 * torch.randn(B, L, generator=g)).clamp_(-1, 1).to(dev)
    w0 = (torch.randn(256, 10, generator=g) * 0.3).to(dev)
    b0 = (torch.randn(256, generator=g) * 0.1).to(dev)
    nw, nb = torch.ones(256, device=dev), torch.zeros(256, device=dev)
    bound = (15.968719 * nw.abs().max() + nb.abs().max()).view(1).clone()
    for h2 in (False, True):
        y0 = torch.empty(B, 4096, 256, device=dev)
        m0, r0 = torch.empty(B * 4096, device=dev), torch.empty(B * 4096, device=dev)

        def victim(st):
            lib.check(lib.cpc_conv0_forward_h2(P(wave), P(w0), P(b0), P(nw), P(nb), P(y0), P(m0), P(r0),
                                               P(bound) if h2 else None, B, L, st.cuda_stream))
        outs = [y0, m0, r0]
        bits = lambda: [t.view(torch.int32).clone() for t in outs]          # H2 storage: compare bit patterns, not floats
        s1 = torch.cuda.Stream()
        torch.cuda.synchronize()
        victim(s1)
        torch.cuda.synchronize()
        ref = bits()
        for kind in ("wgrad", "dgrad"):
            for it in range(3):
                for t in outs:
                    t.zero_()
                torch.cuda.synchronize()
                co.launch(kind, 2)
                victim(s1)
                co.launch(kind, 3)
                torch.cuda.synchronize()
                bad = [i for i, (a, b_) in enumerate(zip(ref, bits())) if not torch.equal(a, b_)]
                assert not bad, f"h2={h2}, beside conv_{kind}: outputs {bad} differ from the solo run (round {it})"


def test_norm_backward_is_bit_exact_beside_the_fp16_gemm_kernels():
    dev = _dev()
    co = _Corunner(dev)
    lib, P = co.lib, co.P
    M = 16 * 1024
    g = torch.Generator(device="cpu").manual_seed(2)
    dy = (torch.randn(M, 256, generator=g) * 1e-3).to(dev)
    xhat = torch.randn(M, 256, generator=g).to(dev)
    y = torch.relu(xhat + 0.1)
    rstd = (torch.rand(M, generator=g) + 0.5).to(dev)
    nw = (1 + 0.1 * torch.randn(256, generator=g)).to(dev)
    dx = torch.empty(M, 256, device=dev)
    nblk = -(-M // 32)
    colpart = torch.empty(nblk * 3 * 256, device=dev)
    tmp = torch.empty(128 * 3 * 256 + 64, device=dev)
    small3 = torch.empty(3 * 256, device=dev)
    amax = torch.zeros(1, device=dev)

    def victim(st):
        with torch.cuda.stream(st):
            amax.zero_()                                   # on the victim's stream: the kernel accumulates into it
        lib.check(lib.cpc_norm_backward(P(dy), P(xhat), P(y), P(rstd), P(nw), P(dx), P(colpart), P(tmp), P(small3),
                                        P(amax), M, st.cuda_stream))
    _check_beside(victim, [dx, small3, amax], co)


@pytest.mark.parametrize("gru_mode", [2, 1])
def test_recurrence_is_bit_exact_beside_the_fp16_gemm_kernels(gru_mode):
    """cpc_gru_forward / cpc_gru_backward_coef / cpc_gru_backward at B = 64.  gru_mode 2 (default): the persistent kernels --
    workgroups poll each other's results; a co-runner changes when they become resident, never what they compute.  gru_mode 1:
    the launch-per-step two-layer kernels.  Between them these are the kernels that contained packed (until round 4: build.PACKED_FP32_ALLOWED is empty now)
    fp32 arithmetic (explicit f32x4 operations on MFMA accumulators): gru2_persist_fwd_h2_kernel, gru2_persist_bwd_kernel,
    gru2_bwd_kernel."""
    dev = _dev()
    co = _Corunner(dev)
    lib, P = co.lib, co.P
    from cpc_audio_amd import _lib as L
    lib.check(lib.cpc_set_gru_mode(gru_mode))
    try:
        _recurrence_beside(dev, co, lib, P)
    finally:
        lib.check(lib.cpc_set_gru_mode(L.DEFAULT_GRU_MODE))


def _recurrence_beside(dev, co, lib, P):
    B, S, nl = 64, 128, 2
    g = torch.Generator(device="cpu").manual_seed(3)
    params = []
    for _ in range(nl):
        params += [(torch.randn(768, 256, generator=g) / 16).to(dev), (torch.randn(768, 256, generator=g) / 16).to(dev),
                   (torch.randn(768, generator=g) * 0.1).to(dev), (torch.randn(768, generator=g) * 0.1).to(dev)]
    pp = (ctypes.c_void_p * len(params))(*[P(t) for t in params])
    sizes = (ctypes.c_long * 3)()
    lib.check(lib.cpc_gru_layout(B, S, nl, sizes))
    x = torch.randn(B, S, 256, generator=g).to(dev)
    dy = (torch.randn(B, S, 256, generator=g) * 1e-2).to(dev)
    saved = torch.empty(sizes[0], device=dev)
    fscr, bscr = torch.empty(sizes[1], device=dev), torch.empty(sizes[2], device=dev)
    y, hN = torch.empty(B, S, 256, device=dev), torch.empty(nl, B, 256, device=dev)
    dx = torch.empty(B, S, 256, device=dev)
    grads = [torch.empty_like(t) for t in params]
    gp = (ctypes.c_void_p * len(grads))(*[P(t) for t in grads])

    def victim(st):
        lib.check(lib.cpc_gru_forward(P(x), None, pp, P(saved), P(fscr), P(y), P(hN), B, S, nl, st.cuda_stream))
        lib.check(lib.cpc_gru_backward(P(x), None, pp, P(saved), P(y), P(dy), P(bscr), P(dx), gp, B, S, nl, st.cuda_stream))
    _check_beside(victim, [y, hN, dx] + grads, co, rounds=2)
    assert torch.isfinite(y).all() and torch.isfinite(dx).all()


def test_overlapped_train_steps_stay_bit_identical_to_the_single_stream_step():
    """tools/stress_overlap.py in small: 12 train steps at B = 64 with every stream overlap on."""
    dev = _dev()
    from cpc_audio_amd import ops
    from cpc_audio_amd.train import build_criterion, build_model
    B = 64
    torch.manual_seed(0)
    model, crit = build_model().to(dev), build_criterion().to(dev)
    params = list(model.parameters()) + list(crit.parameters())
    g = torch.Generator(device="cpu").manual_seed(1)
    wave = (0.1 * torch.randn(B, 1, 20480, generator=g)).clamp_(-1, 1).to(dev)
    label = torch.zeros(B, dtype=torch.long, device=dev)
    negs = (torch.randint(0, B, (B * 128 * 116,), generator=g).to(dev),
            torch.randint(1, 128, (B * 128 * 116,), generator=g).to(dev))

    def step(overlap):
        for q in params:
            q.grad = None
        with ops.StepContext(overlap=overlap) as sc:
            c, z, _ = model(wave, label)
            losses, _ = crit(c, z, None, negatives=negs)
            torch.autograd.backward([losses], [torch.ones_like(losses)])
            sc.wait()
        torch.cuda.synchronize()
        return [q.grad.clone() for q in params], losses.detach().clone()

    ref, lref = step(False)
    for i in range(12):
        cur, l = step(True)
        assert torch.equal(l, lref), i
        diff = [k for k, (a, b) in enumerate(zip(ref, cur)) if not torch.equal(a, b)]
        assert not diff, (i, diff)


@pytest.mark.parametrize("B,burst", [(16, 12), (64, 60)])
def test_train_steps_are_bit_identical_beside_foreign_fp16_gemm_kernels(B, burst):
    """Every kernel of the step at once: whole overlapped train steps (B = 16: the small tiles; B = 64: the benchmark's kernels)
    while ANOTHER stream keeps 16-bit-MFMA GEMM kernels on the chip (what a second train loop, or another process, on the same
    device does), against the quiet run."""
    dev = _dev()
    co = _Corunner(dev)
    from cpc_audio_amd import ops
    from cpc_audio_amd.train import build_criterion, build_model
    torch.manual_seed(0)
    model, crit = build_model().to(dev), build_criterion().to(dev)
    params = list(model.parameters()) + list(crit.parameters())
    g = torch.Generator(device="cpu").manual_seed(3)
    wave = (0.1 * torch.randn(B, 1, 20480, generator=g)).clamp_(-1, 1).to(dev)
    negs = (torch.randint(0, B, (B * 128 * 116,), generator=g).to(dev),
            torch.randint(1, 128, (B * 128 * 116,), generator=g).to(dev))

    def step(noise):
        for q in params:
            q.grad = None
        torch.cuda.synchronize()
        if noise:
            co.launch(noise, burst)                                # ~ the duration of the step, on its own stream
        with ops.StepContext(overlap=True) as sc:
            c, z, _ = model(wave, None)
            losses, _ = crit(c, z, None, negatives=negs)
            torch.autograd.backward([losses], [torch.ones_like(losses)])
            sc.wait()
        torch.cuda.synchronize()
        return [q.grad.clone() for q in params], losses.detach().clone()

    ref, lref = step(None)
    for i in range(6):
        cur, l = step("wgrad" if i % 2 == 0 else "dgrad")
        assert torch.equal(l, lref), i
        diff = [k for k, (a, b) in enumerate(zip(ref, cur)) if not torch.equal(a, b)]
        assert not diff, (i, diff)
