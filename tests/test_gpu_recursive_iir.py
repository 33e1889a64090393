"""GPU parity of the exact recursive IIR backends (`IIRFilter(backend="lfilter" | "ssm")`):
the parallel-scan kernel `gfx_biquad_cascade_f32` against the reference's own section loop / state-space code
run on stand-ins for torchaudio.lfilter and torchlpc.sample_wise_lpc (tests/golden/make_golden.py:g12), plus
size-independent properties at the headline length."""
import pytest
import torch

from conftest import assert_close, assert_parity

pytestmark = pytest.mark.gpu
G = "g12_recursive_iir"


@pytest.mark.parametrize("backend", ["lfilter", "ssm"])
@pytest.mark.parametrize("K", [1, 3])
@pytest.mark.parametrize("C,Cf", [(2, 1), (1, 2), (2, 2)])
def test_recursive_backends_match_the_reference(golden, backend, K, C, Cf):
    from grafx_amd.processors import IIRFilter

    g = golden(G)
    tag = f"K{K}_C{C}_F{Cf}"
    m = IIRFilter(order=2, backend=backend, flashfftconv=False)
    with torch.no_grad():
        y = m(g[f"x_{tag}"].cuda(), g[f"Bs_{tag}"].cuda(), g[f"As_{tag}"].cuda())
    # both sides are fp32 recursions with different operation orders; the float64 run of the reference arbitrates
    assert_parity(y.cpu(), g[f"y_{backend}_{tag}"], g[f"y64_{backend}_{tag}"], 1e-5, f"{backend} {tag}")


def test_ssm_and_lfilter_agree_for_one_section_and_differ_for_more(golden):
    g = golden(G)
    assert_close(g["y_ssm_K1_C2_F1"], g["y_lfilter_K1_C2_F1"], 1e-5, "reference K=1")
    d = (g["y_ssm_K3_C2_F1"] - g["y_lfilter_K3_C2_F1"]).abs().max() / g["y_lfilter_K3_C2_F1"].abs().max()
    assert d > 0.1  # upstream's "ssm" feeds the original input to every section's recursion (core/iir.py:226-246)


def test_recursive_cascade_matches_float64_direct_form_across_tile_boundaries():
    """L = 5000 (4 full tiles + a ragged one), strided buffer views, high-Q poles near the unit circle."""
    from grafx_amd import ops

    torch.manual_seed(0)
    R, K, L = 3, 6, 5000
    x = torch.randn(R, 2, L)
    radius = torch.tensor([0.5, 0.9, 0.99, 0.999, 0.9995, 0.8]).expand(R, 1, K)
    theta = torch.rand(R, 1, K) * 3.0 + 0.05
    As = torch.stack([torch.ones_like(radius), -2 * radius * torch.cos(theta), radius.square()], -1)
    Bs = torch.stack([torch.ones_like(radius), -1.6 * torch.cos(theta), 0.64 * torch.ones_like(radius)], -1)
    y64 = x.double()
    for k in range(K):  # plain direct-form recursion in float64
        b, a = Bs[:, 0, k].double(), As[:, 0, k].double()
        out = torch.zeros_like(y64)
        w1 = torch.zeros(R, 2, dtype=torch.float64)
        w2 = torch.zeros(R, 2, dtype=torch.float64)
        for n in range(L):
            w = y64[..., n] - a[:, 1:2] * w1 - a[:, 2:3] * w2
            out[..., n] = b[:, 0:1] * w + b[:, 1:2] * w1 + b[:, 2:3] * w2
            w2, w1 = w1, w
        y64 = out
    buf = torch.zeros(1, 8, 2, L, device="cuda")
    buf[0, 2:5] = x.cuda()
    ops.biquad_cascade(buf.narrow(1, 2, 3), Bs.cuda(), As.cuda(), out=buf.narrow(1, 5, 3))
    assert_close(buf[0, 5:8].cpu(), y64.float(), 2e-5, "cascade vs float64 direct form")
    assert torch.equal(buf[0, :2], torch.zeros_like(buf[0, :2]))


@pytest.mark.parametrize("C,Cf,odd", [(2, 2, False), (2, 1, True), (1, 2, False), (1, 1, True)])
def test_recursive_cascade_with_sixteen_lanes_per_pair_equals_the_whole_wave_form(C, Cf, odd):
    """From 8192 pairs of row-channels on the cascade runs with sixteen lanes per pair (four pairs a wave, DPP row scans);
    below, with the whole wave per pair (rows' totals carried across by row_bcast).  The same rows through both forms --
    all at once, and in slices small enough for the other kernel -- against a float64 direct-form recursion of every row:
    both are within rounding of it (random numerators make a few rows ill-conditioned in ANY fp32 evaluation: those are
    compared form against form), row by row the sixteen-lane form is as close as the whole-wave form; odd row counts
    leave a lane group / a pair's second half without work, L = 300 a ragged last tile in both tile sizes."""
    from grafx_amd import ops

    torch.manual_seed(C + 2 * Cf)
    Cout = max(C, Cf)
    R = (16384 // Cout) + (3 if odd else 0)        # >= 8192 pairs
    K, L = 3, 300 if odd else 868
    x = torch.randn(R, C, L, device="cuda")
    rad = 0.5 + 0.45 * torch.rand(R, Cf, K, device="cuda")
    th = 3.0 * torch.rand(R, Cf, K, device="cuda")
    As = torch.stack([torch.ones_like(rad), -2 * rad * torch.cos(th), rad * rad], -1)
    Bs = torch.randn(R, Cf, K, 3, device="cuda")
    y = ops.biquad_cascade(x, Bs, As)
    step = 2048 // Cout
    parts = torch.cat([ops.biquad_cascade(x[i : i + step], Bs[i : i + step], As[i : i + step]) for i in range(0, R, step)])
    ref = x.double().expand(R, Cout, L)
    for k in range(K):
        b, a = Bs[:, :, k].double().expand(R, Cout, 3), As[:, :, k].double().expand(R, Cout, 3)
        out = torch.zeros_like(ref)
        w1 = torch.zeros(R, Cout, dtype=torch.float64, device="cuda")
        w2 = torch.zeros_like(w1)
        for n in range(L):
            w = ref[..., n] - a[..., 1] * w1 - a[..., 2] * w2
            out[..., n] = b[..., 0] * w + b[..., 1] * w1 + b[..., 2] * w2
            w2, w1 = w1, w
        ref = out
    scale = ref.abs().amax(dim=(1, 2), keepdim=True)
    e16 = ((y.double() - ref).abs() / scale).amax(dim=(1, 2))
    e64 = ((parts.double() - ref).abs() / scale).amax(dim=(1, 2))
    assert e16.median().item() < 2e-6 and e64.median().item() < 2e-6
    assert (e16 <= 3 * e64 + 2e-6).all(), f"worst row: {e16.max().item():.2e} against {e64[e16.argmax()].item():.2e}"
    assert (e16 < 1e-5).float().mean().item() > 0.99


def test_recursive_cascade_at_headline_length_is_linear_and_matches_the_fsm_limit():
    """L = 131072: superposition holds to rounding, and for a well-damped equaliser the FSM backend (whose only
    error is time aliasing of the truncated impulse response) agrees with the exact recursion."""
    import grafx_amd.processors as P
    from grafx_amd import ops

    torch.manual_seed(1)
    R, L = 4, 131072
    eq_exact = P.ParametricEqualizer(num_filters=6, backend="lfilter", flashfftconv=False).cuda()
    eq_fsm = P.ParametricEqualizer(num_filters=6, backend="fsm", flashfftconv=False, fsm_fir_len=4001).cuda()
    p = {k: 0.3 * torch.randn(R, 1, 6, device="cuda") for k in ("w0", "q_inv", "log_gain")}
    x1, x2 = torch.randn(R, 2, L, device="cuda"), torch.randn(R, 2, L, device="cuda")
    with torch.no_grad():
        y1, y2, y12 = eq_exact(x1, **p), eq_exact(x2, **p), eq_exact(x1 + 2 * x2, **p)
        assert_close(y12.cpu(), (y1 + 2 * y2).cpu(), 1e-5, "superposition")
        assert_close(eq_fsm(x1, **p).cpu(), y1.cpu(), 1e-3, "FSM vs exact recursion")


def test_recursive_backend_refuses_bad_orders():
    from grafx_amd.processors import IIRFilter

    with pytest.raises(NotImplementedError):
        IIRFilter(order=4, backend="ssm", flashfftconv=False)
    with pytest.raises(NotImplementedError):
        IIRFilter(order=3, backend="lfilter", flashfftconv=False)
    with pytest.raises(NotImplementedError):
        IIRFilter(order=1, backend="ssm", flashfftconv=False)      # upstream's ssm asserts order 2 (core/iir.py:226)


def test_first_order_sections_on_the_recursive_backend():
    """IIRFilter(order=1, backend="lfilter") (reference core/iir.py:154-183 with two coefficients per section): first-order
    sections ride on the biquad kernel with a zero third coefficient -- against the direct-form recursion in float64,
    forward and gradients."""
    from grafx_amd.processors import IIRFilter

    torch.manual_seed(1)
    R, K, L = 4, 3, 3000
    x = torch.randn(R, 2, L)
    pole = torch.tensor([0.5, -0.9, 0.995]).expand(R, 1, K)
    As = torch.stack([torch.ones_like(pole), -pole], -1)
    Bs = torch.stack([torch.ones_like(pole) * 0.7, 0.3 * torch.ones_like(pole)], -1) + 0.05 * torch.randn(R, 1, K, 2)

    def direct(xd, B, A):
        y = xd
        for k in range(K):
            b, a = B[:, 0, k], A[:, 0, k]
            out, w1 = [], torch.zeros(R, 2, dtype=xd.dtype)
            for n in range(L):      # a0 y[n] = b0 x[n] + b1 x[n-1] - a1 y[n-1] (lfilter normalises by a0, as the kernel does)
                w = (y[..., n] - a[:, 1:2] * w1) / a[:, 0:1]
                out.append(b[:, 0:1] * w + b[:, 1:2] * w1)
                w1 = w
            y = torch.stack(out, -1)
        return y

    m = IIRFilter(order=1, backend="lfilter", flashfftconv=False).cuda()
    with torch.no_grad():
        y = m(x.cuda(), Bs.cuda(), As.cuda())
    assert_close(y.cpu(), direct(x.double(), Bs.double(), As.double()).float(), 2e-5, "first-order cascade")
    Bg, Ag = Bs.cuda().requires_grad_(), As.cuda().requires_grad_()
    w = torch.randn(R, 2, L)
    gB, gA = torch.autograd.grad((m(x.cuda(), Bg, Ag) * w.cuda()).sum(), (Bg, Ag))
    B64, A64 = Bs.double().requires_grad_(), As.double().requires_grad_()
    rB, rA = torch.autograd.grad((direct(x.double(), B64, A64) * w.double()).sum(), (B64, A64))
    assert gB.shape == Bs.shape and gA.shape == As.shape
    assert_close(gB.cpu(), rB.float(), 1e-4, "first-order cascade grad Bs")
    assert_close(gA.cpu(), rA.float(), 1e-4, "first-order cascade grad As")


def test_graphic_equalizer_on_the_exact_backend_with_31_sections():
    """The largest cascade upstream builds (third-octave GEQ, 31 bands; sr = 48 kHz keeps the 20.16 kHz band) on the
    recursive kernel (4 waves x 31 sections of scan constants in LDS) against a float64 direct-form recursion."""
    import grafx_amd.processors as P
    from grafx_amd.processors.core.geq import GraphicEqualizerBiquad

    torch.manual_seed(4)
    L = 700
    x = torch.randn(2, 2, L)
    lg = 0.4 * torch.randn(2, 1, 31)
    m = P.GraphicEqualizer(scale="third_octave", sr=48000, backend="lfilter", flashfftconv=False).cuda()
    assert m.geq.num_bands == 31
    with torch.no_grad():
        y = m(x.cuda(), log_gains=lg.cuda()).cpu()
    # The 20 Hz bands put poles at radius 0.9994 and angle 0.0026 rad, where a direct-form biquad is ill-conditioned
    # in fp32: with identical (fp32-rounded, a0-normalised) coefficients a plain sequential fp32 recursion is itself
    # ~4e-4 away from the float64 one.  The kernel must be as close to float64 as that, not closer than fp32 allows.
    Bs, As = GraphicEqualizerBiquad(scale="third_octave", sr=48000)(lg)
    Bn, An = Bs / As[..., :1], As / As[..., :1]

    def direct_form(dtype):
        sig = x.to(dtype)
        for k in range(31):
            b, a = Bn[:, 0, k].to(dtype), An[:, 0, k].to(dtype)
            out = torch.zeros_like(sig)
            w1 = torch.zeros(2, 2, dtype=dtype)
            w2 = torch.zeros(2, 2, dtype=dtype)
            for n in range(L):
                w = sig[..., n] - a[:, 1:2] * w1 - a[:, 2:3] * w2
                out[..., n] = b[:, 0:1] * w + b[:, 1:2] * w1 + b[:, 2:3] * w2
                w2, w1 = w1, w
            sig = out
        return sig

    ref64, ref32 = direct_form(torch.float64), direct_form(torch.float32)
    fp32_noise = (ref32.double() - ref64).abs().max() / ref64.abs().max()
    ours = (y.double() - ref64).abs().max() / ref64.abs().max()
    assert ours <= max(2 * fp32_noise, 5e-5), f"kernel {ours:.2e} vs sequential fp32 {fp32_noise:.2e} (both against float64)"


@pytest.mark.parametrize("backend,K", [("lfilter", 1), ("lfilter", 3), ("ssm", 1)])
@pytest.mark.parametrize("C,Cf", [(2, 1), (1, 2), (2, 2)])
def test_recursive_backends_are_differentiable(backend, K, C, Cf):
    """Gradients of IIRFilter(backend="lfilter" | "ssm") (BiquadCascadeFn: the native recursion run backwards in time)
    against torch autograd of a float64 direct-form recursion written as a Python loop (upstream differentiates the
    same recursion through torchaudio.lfilter, core/iir.py:154-183)."""
    from grafx_amd.processors import IIRFilter

    torch.manual_seed(10 * K + C + Cf)
    R, L = 3, 400
    x = torch.randn(R, C, L)
    radius = 0.5 + 0.45 * torch.rand(R, Cf, K)
    theta = torch.rand(R, Cf, K) * 2.8 + 0.1
    As = torch.stack([1.0 + 0.2 * torch.rand(R, Cf, K), -2 * radius * torch.cos(theta), radius.square()], -1)
    Bs = torch.randn(R, Cf, K, 3)
    wgt = torch.randn(R, max(C, Cf), L)

    def ref(x, Bs, As):   # float64 direct form II transposed, channel broadcast like the reference
        Co = max(C, Cf)
        y = x.expand(R, Co, L)
        for k in range(K):
            b, a = Bs[:, :, k].expand(R, Co, 3), As[:, :, k].expand(R, Co, 3)
            s1 = torch.zeros(R, Co, dtype=x.dtype)
            s2 = torch.zeros(R, Co, dtype=x.dtype)
            out = []
            for n in range(L):
                xn = y[..., n]
                yn = (b[..., 0] * xn + s1) / a[..., 0]
                s1 = b[..., 1] * xn - a[..., 1] * yn + s2
                s2 = b[..., 2] * xn - a[..., 2] * yn
                out.append(yn)
            y = torch.stack(out, -1)
        return y

    x64, B64, A64 = (t.double().requires_grad_(True) for t in (x, Bs, As))
    (ref(x64, B64, A64) * wgt.double()).sum().backward()
    xg, Bg, Ag = (t.cuda().requires_grad_(True) for t in (x, Bs, As))
    y = IIRFilter(order=2, backend=backend, flashfftconv=False)(xg, Bg, Ag)
    (y * wgt.cuda()).sum().backward()
    for name, got, want in (("x", xg.grad, x64.grad), ("Bs", Bg.grad, B64.grad), ("As", Ag.grad, A64.grad)):
        assert got is not None and got.shape == want.shape, name
        err = (got.cpu().double() - want).abs().max() / want.abs().max()
        assert err <= 2e-4, f"{backend} K={K} C={C}/{Cf} grad {name}: {err:.2e}"


@pytest.mark.parametrize("K,C,Cf", [(2, 2, 1), (3, 1, 2), (4, 2, 2)])
def test_ssm_quirk_with_several_sections_trains(K, C, Cf):
    """Upstream's "ssm" with K > 1 (core/iir.py:186-260: every section's recursion driven by the ORIGINAL input) used to be
    forward-only here; spelled out as K parallel single-section recursions folded with the direct gains it has gradients.
    Forward: the values of the kernel's ssm_quirk path; gradients: torch autograd of the same formula as a float64 loop."""
    from grafx_amd.processors import IIRFilter

    torch.manual_seed(K + C + 2 * Cf)
    R, L = 3, 300
    x = torch.randn(R, C, L)
    radius = 0.4 + 0.5 * torch.rand(R, Cf, K)
    theta = torch.rand(R, Cf, K) * 2.8 + 0.1
    As = torch.stack([1.0 + 0.2 * torch.rand(R, Cf, K), -2 * radius * torch.cos(theta), radius.square()], -1)
    Bs = torch.randn(R, Cf, K, 3)
    wgt = torch.randn(R, max(C, Cf), L)

    def ref(x, Bs, As):
        Co = max(C, Cf)
        x0 = x.expand(R, Co, L)
        y = x0
        for k in range(K):
            b, a = Bs[:, :, k].expand(R, Co, 3), As[:, :, k].expand(R, Co, 3)
            b0 = b[..., 0] / a[..., 0]
            a1, a2 = a[..., 1] / a[..., 0], a[..., 2] / a[..., 0]
            c1, c2 = b[..., 1] / a[..., 0] - b0 * a1, b[..., 2] / a[..., 0] - b0 * a2
            w1 = torch.zeros(R, Co, dtype=x.dtype)
            w2 = torch.zeros(R, Co, dtype=x.dtype)
            out = []
            for n in range(L):      # d[n] = (c1 z^-1 + c2 z^-2) / A applied to the ORIGINAL input
                w = x0[..., n] - a1 * w1 - a2 * w2
                out.append(b0 * y[..., n] + c1 * w1 + c2 * w2)
                w2, w1 = w1, w
            y = torch.stack(out, -1)
        return y

    m = IIRFilter(order=2, backend="ssm", flashfftconv=False)
    with torch.no_grad():
        y_kernel = m(x.cuda(), Bs.cuda(), As.cuda()).cpu()
    x64, B64, A64 = (t.double().requires_grad_(True) for t in (x, Bs, As))
    y64 = ref(x64, B64, A64)
    assert (y_kernel.double() - y64.detach()).abs().max() <= 2e-5 * y64.abs().max()
    (y64 * wgt.double()).sum().backward()
    xg, Bg, Ag = (t.cuda().requires_grad_(True) for t in (x, Bs, As))
    y = m(xg, Bg, Ag)
    assert (y.detach().cpu().double() - y64.detach()).abs().max() <= 2e-5 * y64.abs().max()
    (y * wgt.cuda()).sum().backward()
    for name, got, want in (("x", xg.grad, x64.grad), ("Bs", Bg.grad, B64.grad), ("As", Ag.grad, A64.grad)):
        assert got is not None and got.shape == want.shape, name
        err = (got.cpu().double() - want).abs().max() / want.abs().max()
        assert err <= 2e-4, f"ssm quirk K={K} C={C}/{Cf} grad {name}: {err:.2e}"
