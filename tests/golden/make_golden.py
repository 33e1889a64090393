#!/usr/bin/env python3
"""Generate golden vectors by running the upstream reference itself.

Runs ONLY in the build container (needs /root/reference and the four
third-party stand-ins in tests/golden/_shims).  Writes small .npz / .json
fixtures next to this file; those are data (inputs + expected outputs), the
reference's source never leaves /root/reference.

    python tests/golden/make_golden.py

Groups follow SURVEY.md §8c (G1..G9).
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "_shims"))
sys.path.insert(1, "/root/reference/src")

import numpy as np  # noqa: E402
import torch  # noqa: E402

import grafx  # noqa: E402  (the upstream reference)
from grafx.data import GRAFX, NodeConfigs, convert_to_tensor  # noqa: E402
from grafx.processors import (  # noqa: E402
    BiquadFilter,
    Compressor,
    NoiseGate,
    ParametricEqualizer,
    STFTMaskedNoiseReverb,
    StereoGain,
)
from grafx.processors.core.convolution import convolve  # noqa: E402
from grafx.processors.core.envelope import Ballistics, TruncatedOnePoleIIRFilter  # noqa: E402
from grafx.processors.core.iir import IIRFilter  # noqa: E402
from grafx.render import prepare_render, render_grafx, reorder_for_fast_render  # noqa: E402
from grafx.utils import create_empty_parameters  # noqa: E402

assert grafx.__file__.startswith("/root/reference"), grafx.__file__


def npy(t):
    return t.detach().cpu().numpy()


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: (npy(v) if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrays.items()})
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def grads(y, params, seed):
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(y.shape, generator=g)
    gs = torch.autograd.grad((y * w).sum(), list(params.values()))
    return w, {f"grad_{k}": v for k, v in zip(params.keys(), gs)}


# ----------------------------------------------------------------------------- G1 convolve
def g1():
    out = {}
    torch.manual_seed(1)
    for (L, N) in [(1024, 128), (1025, 128), (1024, 127), (1000, 301)]:
        xs = torch.randn(3, 2, L)
        hs = torch.randn(3, 2, N) / N**0.5
        out[f"x_L{L}_N{N}"], out[f"h_L{L}_N{N}"] = xs, hs
        for (C, Cf) in [(1, 1), (2, 1), (1, 2), (2, 2)]:
            x, h = xs[:, :C], hs[:, :Cf]  # channel-sliced views of the stored pair
            tag = f"L{L}_N{N}_C{C}_Cf{Cf}"
            for mode in ("causal", "zerophase"):
                out[f"y_{mode}_{tag}"] = convolve(x, h, mode=mode)
    save("g1_convolve", **out)


# ----------------------------------------------------------------------------- G2 IIRFilter fsm
def g2():
    out = {}
    torch.manual_seed(2)
    for N in (256, 257, 500):
        for K in (1, 6):
            flt = IIRFilter(order=2, backend="fsm", flashfftconv=False, fsm_fir_len=N)
            Bs = torch.randn(3, 2, K, 3) * 0.3 + torch.tensor([1.0, 0, 0])
            r = torch.rand(3, 2, K) * 0.9
            th = torch.rand(3, 2, K) * 3.1
            As = torch.stack([torch.ones_like(r), -2 * r * torch.cos(th), r * r], -1)
            x = torch.randn(3, 2, 1023)
            tag = f"N{N}_K{K}"
            resp = IIRFilter.iir_fsm(Bs, As, delays=flt.delays).prod(-2)
            fir = torch.fft.irfft(resp, dim=-1, n=N)
            out[f"Bs_{tag}"], out[f"As_{tag}"], out[f"x_{tag}"] = Bs, As, x
            out[f"fir_{tag}"], out[f"y_{tag}"] = fir, flt(x, Bs, As)
    save("g2_iir_fsm", **out)


# ----------------------------------------------------------------------------- G3 PEQ
def g3():
    out = {}
    for ch in ("mono", "stereo", "midside"):
        for N in (256, 257):
            for std in (0.01, 1.0):
                torch.manual_seed(3)
                peq = ParametricEqualizer(num_filters=6, processor_channel=ch, flashfftconv=False, fsm_fir_len=N)
                shp = peq.parameter_size()["w0"]
                p = {k: (std * torch.randn(3, *shp)).requires_grad_() for k in ("w0", "q_inv", "log_gain")}
                x = torch.randn(3, 2, 1023)
                y = peq(x, **p)
                tag = f"{ch}_N{N}_std{std}"
                w, gr = grads(y, p, 30)
                out[f"x_{tag}"], out[f"y_{tag}"], out[f"w_{tag}"] = x, y, w
                for k, v in p.items():
                    out[f"{k}_{tag}"] = v
                for k, v in gr.items():
                    out[f"{k}_{tag}"] = v
    save("g3_peq", **out)


# ----------------------------------------------------------------------------- G4 biquad / gain
def g4():
    out = {}
    torch.manual_seed(4)
    for K in (1, 4):
        for N in (256, 257):
            for normalized in (False, True):
                bq = BiquadFilter(num_filters=K, normalized=normalized, flashfftconv=False, fsm_fir_len=N)
                p = {"Bs": 0.3 * torch.randn(3, K, 3), "A1_pre": torch.randn(3, K), "A2_pre": torch.randn(3, K)}
                if normalized:
                    p["A0"] = 1 + 0.2 * torch.randn(3, K)
                x = torch.randn(3, 2, 1023)
                tag = f"K{K}_N{N}_norm{int(normalized)}"
                out[f"x_{tag}"], out[f"y_{tag}"] = x, bq(x, **p)
                for k, v in p.items():
                    out[f"{k}_{tag}"] = v
    x, lg = torch.randn(3, 2, 777), torch.randn(3, 2)
    out["gain_x"], out["gain_log_gain"], out["gain_y"] = x, lg, StereoGain()(x, lg)
    save("g4_biquad_gain", **out)


# ----------------------------------------------------------------------------- G5 reverb
def g5():
    out = {}
    for ir_len in (3000, 3001):
        for ch in ("pseudo_midside", "midside", "stereo"):
            torch.manual_seed(5)
            rv = STFTMaskedNoiseReverb(ir_len=ir_len, processor_channel=ch, flashfftconv=False)
            p = {k: torch.randn(2, 2, 193).requires_grad_() for k in ("init_log_magnitude", "delta_log_magnitude")}
            x = torch.randn(2, 2, 2048)
            y = rv(x, **p)
            tag = f"ir{ir_len}_{ch}"
            if ch == "pseudo_midside":
                out[f"ir_{tag}"] = rv.compute_ir(p["init_log_magnitude"], p["delta_log_magnitude"])
                out[f"noise_stft_re_{tag}"] = rv.noise_stft.real[0, :, :8, :8]
                out[f"noise_stft_im_{tag}"] = rv.noise_stft.imag[0, :, :8, :8]
                out[f"noise_stft_abs_sum_{tag}"] = rv.noise_stft.abs().double().sum()
                w, gr = grads(y, p, 50)
                out[f"w_{tag}"] = w
                for k, v in gr.items():
                    out[f"{k}_{tag}"] = v
            out[f"x_{tag}"], out[f"y_{tag}"] = x, y
            for k, v in p.items():
                out[f"{k}_{tag}"] = v
    torch.manual_seed(55)
    rv = STFTMaskedNoiseReverb(ir_len=3001, gain_envelope=True, flashfftconv=False)
    p = {k: torch.randn(2, *s) for k, s in rv.parameter_size().items()}
    x = torch.randn(2, 2, 2048)
    out["x_genv"], out["y_genv"] = x, rv(x, **p)
    for k, v in p.items():
        out[f"{k}_genv"] = v
    save("g5_reverb", **out)


# ----------------------------------------------------------------------------- G6 dynamics
def g6():
    out = {}
    for cls in (Compressor, NoiseGate):
        for knee in ("hard", "quadratic", "exponential"):
            for sm, iir_len in (("iir", 512), ("iir", 511), (None, 0), ("ballistics", 0)):
                torch.manual_seed(6)
                x = torch.randn(3, 2, 1280) * torch.logspace(-2, 0, 3)[:, None, None]
                out["x_shared"] = x
                kw = dict(energy_smoother=sm, knee=knee, flashfftconv=False)
                if sm == "iir":
                    kw["iir_len"] = iir_len
                m = cls(**kw)
                p = {k: torch.randn(3, s).requires_grad_() for k, s in m.parameter_size().items()}
                y = m(x, **p)
                tag = f"{cls.__name__}_{knee}_{sm}_{iir_len}"
                out[f"y_{tag}"] = y
                for k, v in p.items():
                    out[f"{k}_{tag}"] = v
                if sm == "iir" and iir_len == 511 and knee == "quadratic":
                    w, gr = grads(y, p, 60)
                    out[f"w_{tag}"] = w
                    for k, v in gr.items():
                        out[f"{k}_{tag}"] = v
    # gain smoothers
    for gs, in_log in (("iir", False), ("iir", True), ("ballistics", False)):
        torch.manual_seed(66)
        m = Compressor(energy_smoother="iir", gain_smoother=gs, gain_smooth_in_log=in_log, iir_len=511, flashfftconv=False)
        p = {k: torch.randn(3, s) for k, s in m.parameter_size().items()}
        x = torch.randn(3, 2, 2048)
        tag = f"gs_{gs}_{int(in_log)}"
        out[f"x_{tag}"], out[f"y_{tag}"] = x, m(x, **p)
        for k, v in p.items():
            out[f"{k}_{tag}"] = v
    save("g6_dynamics", **out)


# ----------------------------------------------------------------------------- G7 / G9 smoothers
def g7_g9():
    out = {}
    torch.manual_seed(7)
    z = torch.tensor([[-3.0], [0.0], [2.5], [6.0], [14.0]])
    for n in (512, 511):
        f = TruncatedOnePoleIIRFilter(iir_len=n, flashfftconv=False)
        u = torch.rand(5, 2048)
        out[f"onepole_h_{n}"] = f.compute_impulse(z)
        out[f"onepole_u_{n}"], out[f"onepole_y_{n}"] = u, f(u, z)
    out["onepole_z"] = z
    u, zb = torch.rand(4, 1500) * 2, torch.randn(4, 2) * 2
    out["ball_u"], out["ball_z"], out["ball_y"] = u, zb, Ballistics()(u, zb)  # provisional: recalled torchcomp semantics
    save("g7_g9_smoothers", **out)


# ----------------------------------------------------------------------------- G8 routing + renders
def access_json(a):
    idx = a.idx.tolist() if isinstance(a.idx, torch.Tensor) else (list(a.idx) if a.idx is not None else None)
    return {"method": a.method, "idx": idx}


def render_data_json(rd):
    return {
        "method": rd.method, "num_nodes": int(rd.num_nodes), "max_order": int(rd.max_order), "siso_only": bool(rd.siso_only),
        "iter_list": [{
            "node_type": it.node_type,
            "source_reads": [access_json(a) for a in it.source_reads],
            "aggregations": [access_json(a) for a in it.aggregations],
            "parameter_read": access_json(it.parameter_read),
            "dest_write": access_json(it.dest_write),
        } for it in rd.iter_list],
    }


def build_console(n_ch=32, n_bus=4):
    G = GRAFX(config=NodeConfigs(["eq", "compressor", "reverb"]))
    out_id = G.add("out")
    buses = [G.add("mix") for _ in range(n_bus)]
    send = G.add("mix")
    for ch in range(n_ch):
        _, last = G.add_serial_chain(["in", "eq", "compressor"])
        G.connect(last, buses[ch // (n_ch // n_bus)])
        G.connect(last, send)
    for b in buses:
        e = G.add("eq")
        c = G.add("compressor")
        G.connect(b, e)
        G.connect(e, c)
        G.connect(c, out_id)
    r = G.add("reverb")
    G.connect(send, r)
    G.connect(r, out_id)
    return G


def tensor_json(G_t):
    return {"node_types": G_t.node_types.tolist(), "edge_indices": G_t.edge_indices.tolist(),
            "rendering_orders": G_t.rendering_orders.tolist(), "type_sequence": list(G_t.type_sequence)}


def g8():
    meta, arrays = {}, {}
    # cfg-1 plumbing graph: in -> gain -> biquad -> out
    torch.manual_seed(0)
    G = GRAFX(config=NodeConfigs(["gain", "biquad"]))
    _, last = G.add_serial_chain(["in", "gain", "biquad"])
    G.connect(last, G.add("out"))
    procs = {"gain": StereoGain(), "biquad": BiquadFilter(num_filters=1, flashfftconv=False, fsm_fir_len=257)}
    for method in ("beam", "greedy", "one-by-one"):
        G_t = reorder_for_fast_render(convert_to_tensor(G), method=method)
        rd = prepare_render(G_t)
        meta[f"cfg1_{method}"] = {"tensor": tensor_json(G_t), "render": render_data_json(rd)}
    G_t = reorder_for_fast_render(convert_to_tensor(G), method="beam")
    rd = prepare_render(G_t)
    params = create_empty_parameters(procs, G, std=1e-2)
    x = torch.randn(1, 2, 2048)
    y, _, buf = render_grafx(procs, x, params, rd)
    arrays["cfg1_x"], arrays["cfg1_y"], arrays["cfg1_buf"] = x, y, buf
    for t, d in params.items():
        for k, v in d.items():
            arrays[f"cfg1_p_{t}_{k}"] = v

    # console graphs: full 32-channel (routing only + tiny render) and an 8-channel/2-bus variant
    for name, (n_ch, n_bus) in {"console32": (32, 4), "console8": (8, 2)}.items():
        G = build_console(n_ch, n_bus)
        for method in ("beam", "greedy", "one-by-one"):
            G_t = reorder_for_fast_render(convert_to_tensor(G), method=method)
            rd = prepare_render(G_t)
            meta[f"{name}_{method}"] = {"tensor": tensor_json(G_t), "render": render_data_json(rd)}
        meta[f"{name}_raw"] = {"node_types": convert_to_tensor(G).node_types.tolist(),
                               "edge_indices": convert_to_tensor(G).edge_indices.tolist()}
    torch.manual_seed(8)
    G = build_console(8, 2)
    procs = {
        "eq": ParametricEqualizer(num_filters=6, flashfftconv=False, fsm_fir_len=257),
        "compressor": Compressor(energy_smoother="iir", iir_len=255, flashfftconv=False),
        "reverb": STFTMaskedNoiseReverb(ir_len=1501, flashfftconv=False),
    }
    G_t = reorder_for_fast_render(convert_to_tensor(G), method="beam")
    rd = prepare_render(G_t)
    params = create_empty_parameters(procs, G, std=0.3)
    x = torch.randn(2, 8, 2, 1024)
    y, _, buf = render_grafx(procs, x, params, rd)
    arrays["console8_x"], arrays["console8_y"] = x, y
    arrays["console8_buf_last8"] = buf[:, -8:]
    for t, d in params.items():
        for k, v in d.items():
            arrays[f"console8_p_{t}_{k}"] = v
    with open(os.path.join(HERE, "g8_routing.json"), "w") as f:
        json.dump(meta, f, separators=(",", ":"))
    print("g8_routing.json", os.path.getsize(os.path.join(HERE, "g8_routing.json")) // 1024, "KiB")
    save("g8_render", **arrays)


# ----------------------------------------------------------------------------- G10 "next row" processors
def g10():
    import grafx.processors as P

    out = {}
    torch.manual_seed(10)
    x = torch.randn(3, 2, 1023)
    out["x"] = x
    for name in ("LowPassFilter", "HighPassFilter", "BandPassFilter", "BandRejectFilter", "AllPassFilter"):
        for N in (256, 257):
            m = getattr(P, name)(flashfftconv=False, fsm_fir_len=N)
            p = {k: torch.randn(3, 1) for k in m.parameter_size()}
            out[f"y_{name}_N{N}"] = m(x, **p)
            for k, v in p.items():
                out[f"{k}_{name}_N{N}"] = v
    for name in ("PeakingFilter", "LowShelf", "HighShelf"):
        m = getattr(P, name)(num_filters=2, flashfftconv=False, fsm_fir_len=257)
        p = {k: 0.5 * torch.randn(3, 2) for k in m.parameter_size()}
        out[f"y_{name}"] = m(x, **p)
        for k, v in p.items():
            out[f"{k}_{name}"] = v
    m = P.StateVariableFilter(num_filters=2, flashfftconv=False, fsm_fir_len=257)
    p = {k: 0.5 * torch.randn(3, 2) for k in m.parameter_size()}
    out["y_svf"] = m(x, **p)
    for k, v in p.items():
        out[f"{k}_svf"] = v
    for L in (1024, 1023):
        m = P.ZeroPhaseFIREqualizer(num_magnitude_bins=128)
        xl = torch.randn(3, 2, L)
        lm = 0.5 * torch.randn(3, 128)
        out[f"x_zpfir_L{L}"], out[f"lm_zpfir_L{L}"], out[f"y_zpfir_L{L}"] = xl, lm, m(xl, log_magnitude=lm)
    for cls, kw in ((P.ApproxCompressor, "iir_len"), (P.ApproxNoiseGate, "freq_sample_n")):
        m = cls(**{kw: 255}, flashfftconv=False)
        p = {k: torch.randn(3, 1) for k in m.parameter_size()}
        out[f"y_{cls.__name__}"] = m(x, **p)
        for k, v in p.items():
            out[f"{k}_{cls.__name__}"] = v
    # containers around simple processors
    gain, lp = P.StereoGain(), P.LowPassFilter(flashfftconv=False, fsm_fir_len=257)
    pg, pl = {"log_gain": torch.randn(3, 2)}, {"w0": torch.randn(3, 1), "q_inv": torch.randn(3, 1)}
    w = torch.rand(3, 1)
    out["dw_w"], out["dw_lg"] = w, pg["log_gain"]
    out["y_drywet"] = P.DryWet(gain)(x, drywet_weight=w, **pg)
    out["lp_w0"], out["lp_q_inv"] = pl["w0"], pl["q_inv"]
    y, _ = P.SerialChain({"g": gain, "lp": lp})(x, g=pg, lp=pl)
    out["y_serial"] = y
    pw = torch.randn(3, 2)
    out["pm_w"] = pw
    y, _ = P.ParallelMix({"g": gain, "lp": lp})(x, parallel_weights=pw, g=pg, lp=pl)
    out["y_parallel"] = y
    y, inter = P.GainStagingRegularization(gain)(x, **pg)
    out["y_gsr"], out["gsr_reg"] = y, inter["gain_reg"]
    sg = torch.randn(3, 1)
    out["side_lg"], out["y_side"] = sg, P.SideGainImager()(x, sg)
    mid, side = P.StereoToMidSide()(x)
    out["ms_mid"], out["ms_side"], out["y_ms2lr"] = mid, side, P.MidSideToStereo()(mid, side)
    save("g10_next_rows", **out)


def g11():
    """Second batch of "next" rows: graphic / filterbank equalisers, pole-zero filter, memoryless distortions,
    multitap delay, noise-shaping reverb, envelope followers."""
    import grafx.processors as P
    from grafx.processors.core.delay import SurrogateDelay
    from grafx.processors.core.fft_filterbank import TriangularFilterBank
    from grafx.processors.dynamics import BallisticsEnvelopeFollower, IIREnvelopeFollower

    out = {}
    torch.manual_seed(11)
    np.random.seed(11)  # FilteredNoiseShapingReverb draws its noise from numpy's global generator at construction
    x = torch.randn(3, 2, 2048)
    out["x"] = x
    for scale in ("bark", "third_octave"):
        for ch in ("mono", "stereo", "midside"):
            m = P.GraphicEqualizer(processor_channel=ch, scale=scale, sr=44100, flashfftconv=False, fsm_fir_len=1025)
            (k, shp), = m.parameter_size().items()
            lg = 0.5 * torch.randn(3, *shp)
            lg[0, 0, :3] = 0.0  # exercise the |log gain| < 1e-3 branch
            out[f"geq_lg_{scale}_{ch}"], out[f"geq_y_{scale}_{ch}"] = lg, m(x, log_gains=lg)
    for scale in ("bark_traunmuller", "bark_schroeder", "bark_wang", "mel_htk", "mel_slaney", "linear", "log"):
        fb = TriangularFilterBank(num_frequency_bins=128, num_filters=20, scale=scale, f_min=40, f_max=16000, sr=44100)
        out[f"fbank_{scale}"] = fb.filterbank
    fb = TriangularFilterBank(num_frequency_bins=128, num_filters=20, scale="mel_htk", f_min=40, f_max=16000, sr=44100,
                              low_half_triangle=False)
    out["fbank_mel_htk_nolow"] = fb.filterbank
    e = torch.rand(3, 128)
    out["fbank_e"], out["fbank_analysis"] = e, fb(e, mode="analysis")
    for tag, kw in (("plain", dict()), ("fb", dict(use_filterbank=True, filterbank_kwargs=dict(
            num_filters=24, scale="bark_traunmuller", f_min=40, f_max=16000, sr=44100)))):
        for ch in ("mono", "stereo", "midside"):
            m = P.NewZeroPhaseFIREqualizer(num_frequency_bins=128, processor_channel=ch, **kw)
            (k, shp), = m.parameter_size().items()
            lm = 0.5 * torch.randn(3, *shp)
            out[f"nzp_lm_{tag}_{ch}"], out[f"nzp_y_{tag}_{ch}"] = lm, m(x, log_magnitude=lm)
    # PoleZeroFilter: the reference passes 3-D coefficients to IIRFilter, which only broadcasts for one row
    m = P.PoleZeroFilter(num_filters=3, flashfftconv=False, fsm_fir_len=513)
    pz = dict(log_gain=0.3 * torch.randn(1, 1), poles=torch.randn(1, 3, 2), zeros=torch.randn(1, 3, 2))
    out["pz_x"] = x[:1]
    for k, v in pz.items():
        out[f"pz_{k}"] = v
    out["pz_y"] = m(x[:1], **pz)
    # memoryless distortions
    cfgs = {
        "tanh_a": (P.TanhDistortion, dict(pre_post_gain=True, inverse_post_gain=True, remove_dc=False, use_bias=False)),
        "tanh_b": (P.TanhDistortion, dict(pre_post_gain=True, inverse_post_gain=False, remove_dc=True, use_bias=True)),
        "tanh_c": (P.TanhDistortion, dict(pre_post_gain=False, inverse_post_gain=False, remove_dc=False, use_bias=True)),
        "pw_a": (P.PiecewiseTanhDistortion, dict(pre_post_gain=True, inverse_post_gain=True, remove_dc=False)),
        "pw_b": (P.PiecewiseTanhDistortion, dict(pre_post_gain=True, inverse_post_gain=False, remove_dc=True)),
        "pow_a": (P.PowerDistortion, dict(max_order=10, pre_gain=True, remove_dc=False, use_tanh=False)),
        "pow_b": (P.PowerDistortion, dict(max_order=6, pre_gain=False, remove_dc=True, use_tanh=True)),
        "cheb_a": (P.ChebyshevDistortion, dict(max_order=10, pre_gain=True, remove_dc=False, use_tanh=False)),
        "cheb_b": (P.ChebyshevDistortion, dict(max_order=6, pre_gain=False, remove_dc=True, use_tanh=True)),
    }
    xs = 0.7 * x
    out["nl_x"] = xs
    for tag, (cls, kw) in cfgs.items():
        m = cls(**kw)
        ps = {}
        for k, shp in m.parameter_size().items():
            ps[k] = 0.5 * torch.randn(3, shp)
            out[f"nl_{tag}_{k}"] = ps[k]
        out[f"nl_{tag}_y"] = m(xs, **ps)
    # surrogate delay + multitap delay
    for st in (True, False):
        d = SurrogateDelay(N=256, straight_through=st)
        z = torch.view_as_complex(torch.randn(5, 2))
        irs, loss = d(z)
        out[f"sd_z_{int(st)}"], out[f"sd_ir_{int(st)}"], out[f"sd_loss_{int(st)}"] = torch.view_as_real(z), irs, loss
    for tag, kw in (("zp", dict(zp_filter_per_tap=True, zp_filter_bins=8)), ("nozp", dict(zp_filter_per_tap=False))):
        for ch in ("stereo", "mono"):
            m = P.MultitapDelay(segment_len=101, num_segments=5, num_delay_per_segment=2, processor_channel=ch,
                                flashfftconv=False, pre_delay=7 if ch == "stereo" else 0, **kw)
            ps = {k: torch.randn(3, *shp) for k, shp in m.parameter_size().items()}
            y, reg = m(x, **ps)
            for k, v in ps.items():
                out[f"mtd_{tag}_{ch}_{k}"] = v
            out[f"mtd_{tag}_{ch}_y"], out[f"mtd_{tag}_{ch}_reg"] = y, reg["radii_reg"]
            out[f"mtd_{tag}_{ch}_ir"] = m.get_ir(ps["delay_z"], ps.get("log_fir_magnitude"))[0]
    # noise-shaping reverb (fixed noise so that the forward is deterministic given the buffer)
    for ch in ("midside", "stereo", "mono"):
        for fade in (False, True):
            m = P.FilteredNoiseShapingReverb(ir_len=1501, num_bands=4, processor_channel=ch, f_min=100, f_max=8000,
                                             scale="log", sr=30000, noise_randomness="fixed", use_fade_in=fade,
                                             flashfftconv=False)
            ps = {k: torch.randn(3, *shp) for k, shp in m.parameter_size().items()}
            tag = f"{ch}_{int(fade)}"
            out[f"fnr_noise_{tag}"] = m.filtered_noise
            for k, v in ps.items():
                out[f"fnr_{tag}_{k}"] = v
            out[f"fnr_{tag}_y"] = m(x, **ps)
    # the Linkwitz-Riley band split itself (deterministic input)
    from grafx.processors.core.noise import apply_linkwitz_riley
    sig = np.random.RandomState(5).rand(2, 4000) * 2 - 1
    out["lr_in"] = torch.from_numpy(sig)
    for zp in (True, False):
        out[f"lr_out_{int(zp)}"] = torch.from_numpy(apply_linkwitz_riley(sig, num_bands=4, f_min=100, f_max=8000,
                                                                         scale="log", sr=30000, zerophase=zp, order=2))
    # envelope followers
    for det in ("energy", "amplitude"):
        m = IIREnvelopeFollower(detect_with=det, iir_len=255, flashfftconv=False)
        za = torch.randn(3, 1)
        out[f"envf_iir_{det}_z"], out[f"envf_iir_{det}_y"] = za, m(x, za)
        m = BallisticsEnvelopeFollower(detect_with=det)
        zb = torch.randn(3, 2)
        out[f"envf_bal_{det}_z"], out[f"envf_bal_{det}_y"] = zb, m(x, zb)
    save("g11_next_rows2", **out)


def g12():
    """Exact recursive IIR backends.  torchaudio / torchlpc are not installed; the stand-ins in _shims implement
    their documented recursions (scipy.signal.lfilter per row; y_t = x_t - sum_k A[t,k] y_{t-k-1}), everything else
    (section loop, channel broadcasting, the state-space decomposition of "ssm") is the reference's own code."""
    out = {}
    torch.manual_seed(12)
    for K in (1, 3):
        for (C, Cf) in ((2, 1), (1, 2), (2, 2)):
            tag = f"K{K}_C{C}_F{Cf}"
            x = torch.randn(2, C, 3000)
            a1 = 2 * torch.tanh(0.7 * torch.randn(2, Cf, K))
            a2 = ((2 - a1.abs()) * torch.tanh(0.7 * torch.randn(2, Cf, K) + 0.5) + a1.abs()) / 2
            a0 = 1 + 0.3 * torch.rand(2, Cf, K)
            As = torch.stack([a0, a0 * a1, a0 * a2], -1)
            Bs = torch.randn(2, Cf, K, 3)
            out[f"x_{tag}"], out[f"Bs_{tag}"], out[f"As_{tag}"] = x, Bs, As
            for backend in ("lfilter", "ssm"):
                m = IIRFilter(order=2, backend=backend, flashfftconv=False)
                out[f"y_{backend}_{tag}"] = m(x, Bs, As)
                out[f"y64_{backend}_{tag}"] = m(x.double(), Bs.double(), As.double()).float()  # f64 result, stored rounded
    save("g12_recursive_iir", **out)


def g13(n_draws=48):
    """Randomised sweep over the "next" rows: class, constructor options, row count, length and parameter scale drawn at
    random; inputs, parameters, the reference's fp32 output and its float64 output (stored rounded) per draw.  The spec
    of every draw (class name + kwargs) travels as JSON so that the GPU test can rebuild the module."""
    import random

    import grafx.processors as P

    rng = random.Random(13)
    torch.manual_seed(13)
    out, specs = {}, []
    iir = dict(flashfftconv=False)
    menu = [
        ("LowPassFilter", lambda: dict(fsm_fir_len=rng.choice([64, 255, 256, 513]), **iir)),
        ("HighPassFilter", lambda: dict(fsm_fir_len=rng.choice([64, 255, 256, 513]), **iir)),
        ("BandPassFilter", lambda: dict(fsm_fir_len=rng.choice([64, 255, 256, 513]), **iir)),
        ("BandRejectFilter", lambda: dict(fsm_fir_len=rng.choice([64, 255, 256, 513]), **iir)),
        ("AllPassFilter", lambda: dict(fsm_fir_len=rng.choice([64, 255, 256, 513]), **iir)),
        ("PeakingFilter", lambda: dict(num_filters=rng.choice([1, 3]), fsm_fir_len=rng.choice([255, 513]), **iir)),
        ("LowShelf", lambda: dict(num_filters=rng.choice([1, 2]), fsm_fir_len=rng.choice([255, 513]), **iir)),
        ("HighShelf", lambda: dict(num_filters=rng.choice([1, 2]), fsm_fir_len=rng.choice([255, 513]), **iir)),
        ("StateVariableFilter", lambda: dict(num_filters=rng.choice([1, 2, 4]), fsm_fir_len=rng.choice([255, 513]), **iir)),
        ("BiquadFilter", lambda: dict(num_filters=rng.choice([1, 3]), normalized=rng.random() < 0.5,
                                      fsm_fir_len=rng.choice([255, 256, 513]), **iir)),
        ("ParametricEqualizer", lambda: dict(num_filters=rng.choice([2, 5]), processor_channel=rng.choice(["mono", "stereo", "midside"]),
                                             use_shelving_filters=rng.random() < 0.5, fsm_fir_len=rng.choice([255, 256, 513]), **iir)),
        ("GraphicEqualizer", lambda: dict(processor_channel=rng.choice(["mono", "stereo"]), scale=rng.choice(["bark", "third_octave"]),
                                          fsm_fir_len=1025, **iir)),
        ("ZeroPhaseFIREqualizer", lambda: dict(num_magnitude_bins=rng.choice([64, 257]))),
        ("NewZeroPhaseFIREqualizer", lambda: dict(num_frequency_bins=rng.choice([64, 256]),
                                                  processor_channel=rng.choice(["mono", "stereo", "midside"]))),
        ("TanhDistortion", lambda: dict(pre_post_gain=rng.random() < 0.5, inverse_post_gain=rng.random() < 0.5,
                                        remove_dc=rng.random() < 0.5, use_bias=rng.random() < 0.5)),
        ("PowerDistortion", lambda: dict(max_order=rng.choice([3, 8]), pre_gain=rng.random() < 0.5,
                                         remove_dc=rng.random() < 0.5, use_tanh=rng.random() < 0.5)),
        ("ChebyshevDistortion", lambda: dict(max_order=rng.choice([3, 8]), pre_gain=rng.random() < 0.5,
                                             remove_dc=rng.random() < 0.5, use_tanh=rng.random() < 0.5)),
        ("SideGainImager", lambda: dict()),
        ("Compressor", lambda: dict(energy_smoother=rng.choice(["iir", None]), knee=rng.choice(["hard", "quadratic", "exponential"]),
                                    iir_len=rng.choice([63, 255]), flashfftconv=False)),
        ("NoiseGate", lambda: dict(energy_smoother=rng.choice(["iir", None]), knee=rng.choice(["hard", "quadratic", "exponential"]),
                                   iir_len=rng.choice([63, 255]), flashfftconv=False)),
    ]
    i = 0
    while i < n_draws:
        name, make_kwargs = menu[i % len(menu)] if i < len(menu) else rng.choice(menu)
        kwargs = make_kwargs()
        try:
            m = getattr(P, name)(**kwargs)
        except TypeError:
            # constructor option not offered by this class upstream: draw again without it
            kwargs = {k: v for k, v in kwargs.items() if k in ("flashfftconv", "fsm_fir_len", "num_filters")}
            m = getattr(P, name)(**kwargs)
        R, L = rng.choice([1, 2, 3]), rng.choice([500, 1023, 1024, 1025])
        std = rng.choice([0.1, 0.5, 1.0])
        x = torch.randn(R, 2, L)
        sizes = m.parameter_size()
        p = {k: std * torch.randn(R, *((shp,) if isinstance(shp, int) else tuple(shp))) for k, shp in sizes.items()}
        if "log_threshold" in p:
            p["log_threshold"] = p["log_threshold"] - 2
        try:
            y = m(x, **p)
            y64 = m.double()(x.double(), **{k: v.double() for k, v in p.items()})
        except Exception as e:  # a combination the reference itself rejects: record nothing, draw again
            print(f"  skipped {name} {kwargs}: {type(e).__name__}")
            menu = [(n, f) for n, f in menu if n != name] if i < len(menu) else menu
            i += 1 if i < len(menu) else 0
            continue
        y = y[0] if isinstance(y, tuple) else y
        y64 = y64[0] if isinstance(y64, tuple) else y64
        tag = f"c{len(specs):02d}"
        out[f"{tag}_x"], out[f"{tag}_y"], out[f"{tag}_y64"] = x, y, y64.float()
        for k, v in p.items():
            out[f"{tag}_p_{k}"] = v
        specs.append({"tag": tag, "cls": name, "kwargs": kwargs, "params": list(p)})
        i += 1
    out["specs"] = np.array(json.dumps(specs))
    save("g13_random_next_rows", **out)


def g14(n_graphs=24):
    """Random DAGs through the reference's graph API -> tensors -> the three schedulers -> RenderData, plus the
    reference's render of each (beam schedule, StereoGain / BiquadFilter nodes, batch 2, 256 samples).  The build
    recipe of every graph (add / connect calls in order) travels with the expected data."""
    import random

    rng = random.Random(14)
    torch.manual_seed(14)
    meta, arrays = [], {}
    procs = {"gain": StereoGain(), "biquad": BiquadFilter(num_filters=2, flashfftconv=False, fsm_fir_len=65)}
    for gi in range(n_graphs):
        G = GRAFX(config=NodeConfigs(["gain", "biquad"]))
        recipe, nodes = [], []
        n_src = rng.randint(1, 4)
        for _ in range(n_src):
            nodes.append(G.add("in"))
            recipe.append(["add", "in"])
        for _ in range(rng.randint(2, 9)):
            kind = rng.choice(["gain", "biquad", "gain", "biquad", "mix"])
            v = G.add(kind)
            recipe.append(["add", kind])
            for s in rng.sample(nodes, 1 if kind != "mix" else min(len(nodes), rng.randint(2, 4))):
                G.connect(s, v)
                recipe.append(["connect", int(s), int(v)])
            nodes.append(v)
        out = G.add("out")
        recipe.append(["add", "out"])
        for s in rng.sample(nodes[n_src:], min(rng.randint(1, 3), len(nodes) - n_src)):
            G.connect(s, out)
            recipe.append(["connect", int(s), int(out)])
        entry = {"recipe": recipe, "n_src": n_src, "schedules": {}}
        for method in ("beam", "greedy", "one-by-one"):
            G_t = reorder_for_fast_render(convert_to_tensor(G), method=method)
            entry["schedules"][method] = {"tensor": tensor_json(G_t), "render": render_data_json(prepare_render(G_t))}
        rd = prepare_render(reorder_for_fast_render(convert_to_tensor(G), method="beam"))
        params = create_empty_parameters(procs, G, std=0.3)
        x = torch.randn(2, n_src, 2, 256)
        y, _, buf = render_grafx(procs, x, params, rd)
        arrays[f"g{gi:02d}_x"], arrays[f"g{gi:02d}_y"], arrays[f"g{gi:02d}_buf"] = x, y, buf
        for t, d in params.items():
            for k, v in d.items():
                arrays[f"g{gi:02d}_p_{t}_{k}"] = v
        entry["params"] = {t: list(d) for t, d in params.items()}
        meta.append(entry)
    with open(os.path.join(HERE, "g14_random_graphs.json"), "w") as f:
        json.dump(meta, f, separators=(",", ":"))
    print("g14_random_graphs.json", os.path.getsize(os.path.join(HERE, "g14_random_graphs.json")) // 1024, "KiB")
    save("g14_random_graphs", **arrays)


class TrimmedGain(torch.nn.Module):
    """A user-defined processor with a per-type parameter and a *common* parameter (render_grafx's
    common_parameters: one row per node of the graph, whatever its type): y = x * exp(log_gain) * trim."""

    def forward(self, input_signals, log_gain, trim):
        return input_signals * torch.exp(log_gain)[..., None] * trim[..., None]

    def parameter_size(self):
        return {"log_gain": 2}


def g15(n_graphs=8):
    """render_grafx with common_parameters (reference render/graph.py:72-75, 132-141), 3-D and 4-D inputs."""
    import random

    rng = random.Random(15)
    torch.manual_seed(15)
    meta, arrays = [], {}
    procs = {"gain": TrimmedGain(), "trim": TrimmedGain()}
    for gi in range(n_graphs):
        G = GRAFX(config=NodeConfigs(["gain", "trim"]))
        recipe, nodes = [], []
        n_src = rng.randint(1, 3)
        for _ in range(n_src):
            nodes.append(G.add("in"))
            recipe.append(["add", "in"])
        for _ in range(rng.randint(2, 7)):
            kind = rng.choice(["gain", "trim", "mix"])
            v = G.add(kind)
            recipe.append(["add", kind])
            for s in rng.sample(nodes, 1 if kind != "mix" else min(len(nodes), rng.randint(2, 3))):
                G.connect(s, v)
                recipe.append(["connect", int(s), int(v)])
            nodes.append(v)
        out = G.add("out")
        recipe.append(["add", "out"])
        for s in rng.sample(nodes[n_src:], min(2, len(nodes) - n_src)):
            G.connect(s, out)
            recipe.append(["connect", int(s), int(out)])
        G_t = reorder_for_fast_render(convert_to_tensor(G), method="beam")
        rd = prepare_render(G_t)
        params = create_empty_parameters(procs, G, std=0.3)
        common = {"trim": 1.0 + 0.2 * torch.randn(int(rd.num_nodes), 1)}
        batched = gi % 2 == 0
        x = torch.randn(2, n_src, 2, 128) if batched else torch.randn(n_src, 2, 128)
        y, _, buf = render_grafx(procs, x, params, rd, common_parameters=common)
        tag = f"g{gi:02d}"
        arrays[f"{tag}_x"], arrays[f"{tag}_y"], arrays[f"{tag}_buf"], arrays[f"{tag}_trim"] = x, y, buf, common["trim"]
        for t, d in params.items():
            for k, v in d.items():
                arrays[f"{tag}_p_{t}_{k}"] = v
        meta.append({"recipe": recipe, "params": {t: list(d) for t, d in params.items()}})
    with open(os.path.join(HERE, "g15_common_parameters.json"), "w") as f:
        json.dump(meta, f, separators=(",", ":"))
    save("g15_common_parameters", **arrays)


def g16(n_graphs=10):
    """Training through render_grafx: random DAGs of the headline processor types, a loss on the output (and, for odd
    draws, on an intermediate node of the returned buffer), the reference's autograd gradients of every parameter and
    of the input.  Parameters that receive no gradient upstream are recorded as such (key list "nograd")."""
    import random

    rng = random.Random(16)
    torch.manual_seed(16)
    meta, arrays = [], {}
    for gi in range(n_graphs):
        L = rng.choice([1024, 2047])
        procs = {"eq": ParametricEqualizer(num_filters=4, flashfftconv=False, fsm_fir_len=257),
                 "compressor": Compressor(energy_smoother="iir", iir_len=255, flashfftconv=False),
                 "reverb": STFTMaskedNoiseReverb(ir_len=1501, flashfftconv=False)}
        G = GRAFX(config=NodeConfigs(["eq", "compressor", "reverb"]))
        recipe, nodes = [], []
        n_src = 2
        for _ in range(n_src):
            nodes.append(G.add("in"))
            recipe.append(["add", "in"])
        for _ in range(rng.randint(3, 5)):
            kind = rng.choice(["eq", "compressor", "eq", "compressor", "reverb", "mix"])
            v = G.add(kind)
            recipe.append(["add", kind])
            for s in rng.sample(nodes, 1 if kind != "mix" else min(len(nodes), rng.randint(2, 4))):
                G.connect(s, v)
                recipe.append(["connect", int(s), int(v)])
            nodes.append(v)
        out = G.add("out")
        recipe.append(["add", "out"])
        for s in rng.sample(nodes[n_src:], min(3, len(nodes) - n_src)):
            G.connect(s, out)
            recipe.append(["connect", int(s), int(out)])
        G_t = reorder_for_fast_render(convert_to_tensor(G), method="beam")
        rd = prepare_render(G_t)
        params = create_empty_parameters(procs, G, std=0.3)
        x = torch.randn(2, n_src, 2, L, requires_grad=True)
        y, _, buf = render_grafx(procs, x, params, rd, input_signal_grad=True)
        w = torch.linspace(0.5, 1.5, L)
        loss = (y * w).square().mean()
        if gi % 2 == 1:
            loss = loss + 0.3 * (buf[:, buf.shape[1] // 2] * w).abs().mean()
        loss.backward()
        tag = f"g{gi:02d}"
        arrays[f"{tag}_x"], arrays[f"{tag}_gx"], arrays[f"{tag}_y"] = x.detach(), x.grad, y.detach()
        nograd, plist = [], {}
        for t, d in params.items():
            plist[t] = list(d)
            for k, v in d.items():
                arrays[f"{tag}_p_{t}_{k}"] = v.detach()
                if v.grad is None:
                    nograd.append([t, k])
                else:
                    arrays[f"{tag}_g_{t}_{k}"] = v.grad
        meta.append({"recipe": recipe, "L": L, "params": plist, "nograd": nograd, "buffer_loss": gi % 2 == 1})
    with open(os.path.join(HERE, "g16_render_gradients.json"), "w") as f:
        json.dump(meta, f, separators=(",", ":"))
    save("g16_render_gradients", **arrays)


def g17():
    """Upstream's DEFAULT constructor arguments on the reference's PyTorch-CPU path: `flashfftconv=True` warns
    ("FlashFFTConv is not available. Using native convolution instead.", core/convolution.py:47-51) and takes
    `_native_forward` (convolution.py:82-83), i.e. convolve() with its odd-length aliasing.  Outputs and parameter
    gradients at an even audio length (L + N - 1 odd for every default tap count: aliased) and an odd one (plain)."""
    import warnings

    from grafx.processors.core.convolution import FIRConvolution

    out = {}
    for L in (4096, 4095):
        torch.manual_seed(17)
        x = torch.randn(2, 2, L)
        out[f"x_L{L}"] = x
        cases = {}
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            cases["peq"] = (ParametricEqualizer(num_filters=6),
                            {k: (0.3 * torch.randn(2, 1, 6)).requires_grad_() for k in ("w0", "q_inv", "log_gain")})
            cases["compressor"] = (Compressor(),
                                   {"log_threshold": (torch.randn(2, 1) - 2).requires_grad_(),
                                    "log_ratio": torch.randn(2, 1).requires_grad_(),
                                    "log_knee": torch.randn(2, 1).requires_grad_(),
                                    "z_alpha_pre": (torch.randn(2, 1) + 2).requires_grad_()})
            cases["reverb"] = (STFTMaskedNoiseReverb(),
                               {k: torch.randn(2, 2, 193).requires_grad_()
                                for k in ("init_log_magnitude", "delta_log_magnitude")})
            r, th = torch.rand(2, 2, 3) * 0.9, torch.rand(2, 2, 3) * 3.1
            cases["iir"] = (IIRFilter(),
                            {"Bs": (torch.randn(2, 2, 3, 3) * 0.3 + torch.tensor([1.0, 0, 0])).requires_grad_(),
                             "As": torch.stack([torch.ones_like(r), -2 * r * torch.cos(th), r * r], -1).requires_grad_()})
            cases["fir"] = (FIRConvolution(), {"fir": (torch.randn(2, 1, 512) / 16).requires_grad_()})
        msgs = sorted({str(w.message) for w in rec})
        assert msgs == ["FlashFFTConv is not available. Using native convolution instead."], msgs
        out["warning"] = msgs[0]
        for name, (m, p) in cases.items():
            assert set(p) == set(m.parameter_size()) if hasattr(m, "parameter_size") else True
            y = m(x, **p)
            tag = f"{name}_L{L}"
            w, gr = grads(y, p, 170)
            out[f"y_{tag}"] = y
            out[f"w_L{L}"] = w          # the same draw for every case (same shape, same seed)
            for k, v in p.items():
                out[f"{k}_{tag}"] = v
            for k, v in gr.items():
                out[f"{k}_{tag}"] = v
    save("g17_default_arguments", **out)


def g18():
    """g17 for the rest of the processor zoo: every class that takes `flashfftconv` -- left at upstream's default (True ->
    warning + native convolve() on the reference's CPU path) together with the default tap counts -- at an even audio
    length, where every one of them aliases.  Outputs only."""
    import warnings

    import grafx.processors as P
    out, specs = {}, []
    torch.manual_seed(18)
    L = 4096
    x = torch.randn(2, 2, L)
    out["x"] = x
    cases = [
        ("NoiseGate", lambda: P.NoiseGate(), None), ("ApproxCompressor", lambda: P.ApproxCompressor(), None),
        ("ApproxNoiseGate", lambda: P.ApproxNoiseGate(), None),
        ("LowPassFilter", lambda: P.LowPassFilter(), None), ("HighPassFilter", lambda: P.HighPassFilter(), None),
        ("BandPassFilter", lambda: P.BandPassFilter(), None), ("BandRejectFilter", lambda: P.BandRejectFilter(), None),
        ("AllPassFilter", lambda: P.AllPassFilter(), None),
        ("PeakingFilter", lambda: P.PeakingFilter(num_filters=2), 0.5), ("LowShelf", lambda: P.LowShelf(num_filters=2), 0.5),
        ("HighShelf", lambda: P.HighShelf(num_filters=2), 0.5),
        ("StateVariableFilter", lambda: P.StateVariableFilter(num_filters=2), 0.5),
        ("BiquadFilter", lambda: P.BiquadFilter(num_filters=2), 0.3),
        ("GraphicEqualizer", lambda: P.GraphicEqualizer(), 0.5),
    ]   # (IIREnvelopeFollower.parameter_size() raises upstream: dynamics.py:762 asks its smoother for one)
    for name, make, std in cases:
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            m = make()
        assert any("FlashFFTConv is not available" in str(w.message) for w in rec), name
        ps = {}
        for k, shp in m.parameter_size().items():
            shp = (shp,) if isinstance(shp, int) else tuple(shp)
            ps[k] = (std or 1.0) * torch.randn(2, *shp)
        y = m(x, **ps)
        y = y[0] if isinstance(y, tuple) else y
        out[f"y_{name}"] = y
        for k, v in ps.items():
            out[f"{k}_{name}"] = v
        specs.append({"cls": name, "params": list(ps)})
    out["specs"] = json.dumps(specs)
    save("g18_default_arguments_zoo", **out)


if __name__ == "__main__":
    torch.set_num_threads(4)
    only = sys.argv[1:]
    for fn in (g1, g2, g3, g4, g5, g6, g7_g9, g8, g10, g11, g12, g13, g14, g15, g16, g17, g18):
        if not only or fn.__name__ in only:
            fn()
