"""numpy model of the LDS FFT tile used by csrc/fftconv (index math only).

M = 8192 complex points = 32 x 16 x 16, 256 threads.  Real signals are packed
two samples per complex point (polyphase); the filter is stored per thread as
(He, Ho) pairs in exactly the register layout the conv kernel consumes.
Run: python tools/fft_tile_model.py
"""
import numpy as np

M, T, F = 8192, 256, 16384


def W(n, k):
    return np.exp(-2j * np.pi * (k % n) / n)


def fwd(z):
    """natural-order z[M] -> X[k1][k2][k3] with k = k1 + 32 k2 + 512 k3 (DIF 32,16,16)."""
    u = np.fft.fft(z.reshape(32, 256), axis=0)  # [k1][b]
    u = u * W(M, np.outer(np.arange(32), np.arange(256)))
    s = u.reshape(32, 16, 16)  # [k1][c][d]
    u2 = np.fft.fft(s, axis=1)  # [k1][k2][d]
    u2 = u2 * W(256, np.arange(16)[None, :, None] * np.arange(16)[None, None, :])
    return np.fft.fft(u2, axis=2)  # [k1][k2][k3]


def inv(X):
    """mirror of fwd (unnormalised inverse): X[k1][k2][k3] -> natural-order z*M."""
    u2 = np.fft.ifft(X, axis=2) * 16  # [k1][k2][d]
    u2 = u2 * np.conj(W(256, np.arange(16)[None, :, None] * np.arange(16)[None, None, :]))
    s = np.fft.ifft(u2, axis=1) * 16  # [k1][c][d]
    u = s.reshape(32, 256) * np.conj(W(M, np.outer(np.arange(32), np.arange(256))))
    return (np.fft.ifft(u, axis=0) * 32).reshape(M)


def thread_butterflies(t):
    """j = k1 + 32 k2 in [0,512): thread t owns j and its negation 512-j; thread 0 owns the two self-negating j."""
    return (0, 256) if t == 0 else (t, 512 - t)


def pairs_of_thread(t):
    """list of (slot, (bf, k3), (bf', k3')) with k' = M - k (mod M); self pairs have both equal."""
    jA, jB = thread_butterflies(t)
    out = []
    if t == 0:
        out.append((0, (0, 0), (0, 0)))       # k = 0      (self)
        out.append((8, (0, 8), (0, 8)))       # k = M/2    (self)
        for k3 in range(1, 8):
            out.append((k3, (0, k3), (0, 16 - k3)))
        for k3 in range(8):
            out.append((9 + k3, (1, k3), (1, 15 - k3)))
    else:
        for k3 in range(16):
            out.append((k3, (0, k3), (1, 15 - k3)))
    return out


def k_of(t, bf, k3):
    return thread_butterflies(t)[bf] + 512 * k3


def filter_slots(h):
    """(He, Ho) per (slot, thread) for a real filter partition h (len <= F/2), scaled by 1/(4M) in total."""
    hz = np.zeros(F)
    hz[: len(h)] = h
    Zh = fwd(hz[0::2] + 1j * hz[1::2])
    He = np.zeros((17, T), complex)
    Ho = np.zeros((17, T), complex)
    for t in range(T):
        for slot, (bfa, ka3), (bfb, kb3) in pairs_of_thread(t):
            ja, jb = thread_butterflies(t)[bfa], thread_butterflies(t)[bfb]
            A = Zh[ja % 32, ja // 32, ka3]
            B = np.conj(Zh[jb % 32, jb // 32, kb3])
            # (A+B)/2 is He; 1/M for the unnormalised inverse; 1/2 because the conv side skips it in Xe/Xo
            He[slot, t] = (A + B) / (4 * M)
            Ho[slot, t] = -1j * (A - B) / (4 * M)
    return He, Ho


def conv_tile(x_tile, He, Ho):
    """circular conv of a real tile (len F) with the stored filter; returns real len F."""
    Z = fwd(x_tile[0::2] + 1j * x_tile[1::2])
    Y = np.zeros_like(Z)
    for t in range(T):
        for slot, (bfa, ka3), (bfb, kb3) in pairs_of_thread(t):
            ja, jb = thread_butterflies(t)[bfa], thread_butterflies(t)[bfb]
            ia, ib = (ja % 32, ja // 32, ka3), (jb % 32, jb // 32, kb3)
            k = k_of(t, bfa, ka3)
            A, B = Z[ia], np.conj(Z[ib])
            Xe, Xo = (A + B), -1j * (A - B)           # the 1/2 factors live in He/Ho
            Wk = W(M, k)
            he, ho = He[slot, t], Ho[slot, t]
            Ye = he * Xe + Wk * ho * Xo
            Yo = ho * Xe + he * Xo
            Y[ia] = Ye + 1j * Yo
            if ia != ib:
                Y[ib] = np.conj(Ye - 1j * Yo)
    zz = inv(Y)
    y = np.empty(F)
    y[0::2], y[1::2] = zz.real, zz.imag
    return y


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    z = rng.standard_normal(M) + 1j * rng.standard_normal(M)
    X = fwd(z)
    ref = np.fft.fft(z)
    k1, k2, k3 = np.meshgrid(np.arange(32), np.arange(16), np.arange(16), indexing="ij")
    assert np.allclose(X, ref[k1 + 32 * k2 + 512 * k3]), "fwd"
    assert np.allclose(inv(X), z * M), "inv"
    # every (bf,k3) of every thread appears exactly once
    seen = np.zeros((512, 16), int)
    for t in range(T):
        for slot, (bfa, ka3), (bfb, kb3) in pairs_of_thread(t):
            ja, jb = thread_butterflies(t)[bfa], thread_butterflies(t)[bfb]
            assert (k_of(t, bfa, ka3) + k_of(t, bfb, kb3)) % M == 0
            seen[ja, ka3] += 1
            if (ja, ka3) != (jb, kb3):
                seen[jb, kb3] += 1
    assert (seen == 1).all(), "pair cover"
    N = 4001
    h = rng.standard_normal(N) / 60
    x = rng.standard_normal(F)
    He, Ho = filter_slots(h)
    y = conv_tile(x, He, Ho)
    full = np.convolve(x, h)
    circ = full[:F].copy()
    circ[: N - 1] += full[F:]
    print("circular conv max err", np.abs(y - circ).max())
    assert np.allclose(y, circ, atol=1e-9)
    print("model OK")
