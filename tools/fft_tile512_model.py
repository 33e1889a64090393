"""numpy model of the 512-thread LDS FFT tile (16 points per thread) — index math only.

M = 8192 = 32 x 16 x 16; the radix-32 pass is split over lane pairs (t, t^1).
Roles of thread t:   pass 1: column b = t>>1, half h = t&1      (points a = a' + 16h)
                     pass 2: k1 = t>>4, d = t&15
                     pass 3: LDS row t holds butterfly j(t); mirror butterfly 512-j sits in lane t^1
Run: python tools/fft_tile512_model.py
"""
import numpy as np

M, T, F = 8192, 512, 16384


def W(n, k):
    return np.exp(-2j * np.pi * (np.asarray(k) % n) / n)


def brev4(v):
    return int(f"{v:04b}"[::-1], 2)


def j_of(t):
    if t == 0:
        return 0
    if t == 1:
        return 256
    return t >> 1 if t % 2 == 0 else 512 - (t >> 1)


def rho(j):
    if j == 0:
        return 0
    if j == 256:
        return 1
    return 2 * j if j < 256 else 2 * (512 - j) + 1


def forward(z):
    """per-thread simulation; returns regs[t][r] = Z[j(t) + 512*brev4(r)]"""
    S1 = np.zeros((32, 256), complex)
    for t in range(T):
        b, h = t >> 1, t & 1
        p = np.array([z[256 * (a + 16 * h) + b] for a in range(16)])
        q = np.array([z[256 * (a + 16 * (1 - h)) + b] for a in range(16)])  # lane t^1
        e = p + q if h == 0 else (q - p) * W(32, np.arange(16))
        E = np.fft.fft(e)  # E[m] -> k1 = 2m + h
        for m in range(16):
            S1[2 * m + h, b] = E[m] * W(M, b * (2 * m + h))
    S2 = np.zeros((512, 16), complex)
    for t in range(T):
        k1, d = t >> 4, t & 15
        u = np.fft.fft(np.array([S1[k1, 16 * c + d] for c in range(16)]))
        for k2 in range(16):
            S2[rho(k1 + 32 * k2), d] = u[k2] * W(256, d * k2)
    regs = np.zeros((T, 16), complex)
    for t in range(T):
        w = np.fft.fft(S2[t])
        for k3 in range(16):
            regs[t, brev4(k3)] = w[k3]
    return regs


def mirror(regs):
    """Q[t][r] = Z[M - k] for k = bin of regs[t][r]  (lane t^1 register 15-r; lanes 0,1 are self-mirrored)"""
    Q = np.zeros_like(regs)
    for t in range(T):
        for r in range(16):
            if t == 0:
                Q[t, r] = regs[0, brev4((16 - brev4(r)) & 15)]
            elif t == 1:
                Q[t, r] = regs[1, 15 - r]
            else:
                Q[t, r] = regs[t ^ 1, 15 - r]
    return Q


def inverse(regs):
    S2 = np.zeros((512, 16), complex)
    for t in range(T):
        P = np.array([regs[t, brev4(k3)] for k3 in range(16)])
        S2[t] = np.fft.ifft(P) * 16
    S1 = np.zeros((32, 256), complex)
    for t in range(T):
        k1, d = t >> 4, t & 15
        u = np.array([S2[rho(k1 + 32 * k2), d] * np.conj(W(256, d * k2)) for k2 in range(16)])
        v = np.fft.ifft(u) * 16
        for c in range(16):
            S1[k1, 16 * c + d] = v[c]
    z = np.zeros(M, complex)
    own = np.zeros((T, 16), complex)
    for t in range(T):
        b, h = t >> 1, t & 1
        g = np.array([S1[2 * m + h, b] * np.conj(W(M, b * (2 * m + h))) for m in range(16)])
        o = np.fft.ifft(g) * 16
        if h == 1:
            o = o * np.conj(W(32, np.arange(16)))
        own[t] = o
    for t in range(T):
        b, h = t >> 1, t & 1
        q = own[t ^ 1]
        res = own[t] + q if h == 0 else q - own[t]
        for a in range(16):
            z[256 * (a + 16 * h) + b] = res[a]
    return z


def bins(t):
    return np.array([j_of(t) + 512 * brev4(r) for r in range(16)])


def filter_alpha_beta(h):
    hz = np.zeros(F)
    hz[: len(h)] = h
    regs = forward(hz[0::2] + 1j * hz[1::2])
    Q = mirror(regs)
    al = np.zeros_like(regs)
    be = np.zeros_like(regs)
    for t in range(T):
        A, Bc = regs[t], np.conj(Q[t])
        He, Ho = (A + Bc) / 2, -1j * (A - Bc) / 2
        Wk = W(M, bins(t))
        al[t] = (2 * He + 1j * (1 - Wk) * Ho) / (2 * M)
        be[t] = 1j * (1 + Wk) * Ho / (2 * M)
    return al, be


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    z = rng.standard_normal(M) + 1j * rng.standard_normal(M)
    regs = forward(z)
    ref = np.fft.fft(z)
    for t in range(T):
        assert np.allclose(regs[t], ref[bins(t)]), ("fwd", t)
    assert sorted(np.concatenate([bins(t) for t in range(T)]).tolist()) == list(range(M))
    Q = mirror(regs)
    for t in range(T):
        assert np.allclose(Q[t], ref[(M - bins(t)) % M]), ("mirror", t)
    assert np.allclose(inverse(regs), z * M), "inv"
    assert sorted(rho(j) for j in range(512)) == list(range(512))
    N = 4001
    h = rng.standard_normal(N) / 60
    x = rng.standard_normal(F)
    al, be = filter_alpha_beta(h)
    X = forward(x[0::2] + 1j * x[1::2])
    Y = al * X + be * np.conj(mirror(X))
    zz = inverse(Y)
    y = np.empty(F)
    y[0::2], y[1::2] = zz.real, zz.imag
    full = np.convolve(x, h)
    circ = full[:F].copy()
    circ[: N - 1] += full[F:]
    print("circular conv max err", np.abs(y - circ).max())
    assert np.allclose(y, circ, atol=1e-9)
    print("tile512 model OK")
