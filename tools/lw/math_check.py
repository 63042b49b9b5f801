"""Numerical check of the long-window path's algebra in float64 numpy: odd-frequency four-step
FFT (N = R x M), conj-reversed upper rows, the in-lane 2x2 CMAC tables, and the inverse.  Run: python tools/lw/math_check.py"""
import numpy as np

rng = np.random.default_rng(1)


def w(n, e):            # omega_n^e = exp(-2 pi i e / n)
    return np.exp(-2j * np.pi * np.asarray(e, dtype=np.float64) / n)


def check(R, M, taps, n_ch):
    N = R * M
    n_pairs = (n_ch + 1) // 2
    x = rng.uniform(-0.5, 0.5, (n_ch, N))
    hL = rng.standard_normal((n_ch, taps)) * np.exp(-np.arange(taps) / (taps / 6))
    hR = rng.standard_normal((n_ch, taps)) * np.exp(-np.arange(taps) / (taps / 6))
    # truth: linear convolution, valid for n >= taps-1 (window = whole signal, no wrap)
    yl = sum(np.convolve(x[c], hL[c])[:N] for c in range(n_ch))
    yr = sum(np.convolve(x[c], hR[c])[:N] for c in range(n_ch))

    n = np.arange(N)
    mod = w(2 * N, n)                       # omega_N^{n/2}

    def oddfft(v):                          # X[k] = sum v[n] omega_N^{n (k + 1/2)}
        return np.fft.fft(v * mod)

    def pad(h):
        o = np.zeros(N); o[:taps] = h; return o

    t = np.arange(M)
    j = np.arange(R)
    k1 = np.arange(R)
    tau = w(2 * N, np.outer(2 * k1 + 1, t))          # [k1][t] omega_N^{t (k1 + 1/2)}
    # ---- tables ----
    T = np.zeros((R // 2, n_pairs, M, 4), dtype=complex)
    for p in range(n_pairs):
        a, b = 2 * p, 2 * p + 1
        hb_l = pad(hL[b]) if b < n_ch else np.zeros(N)
        hb_r = pad(hR[b]) if b < n_ch else np.zeros(N)
        zl = oddfft(pad(hL[a]) + 1j * hb_l)
        zr = oddfft(pad(hR[a]) + 1j * hb_r)
        k = np.arange(N)
        A = (np.conj(zl[N - 1 - k]) + 1j * np.conj(zr[N - 1 - k])) / (2 * N)
        B = (zl + 1j * zr) / (2 * N)
        for ra in range(R // 2):
            kk = ra + R * np.arange(M)
            kp = N - 1 - kk
            T[ra, p, :, 0] = A[kk]; T[ra, p, :, 1] = B[kk]
            T[ra, p, :, 2] = np.conj(A[kp]); T[ra, p, :, 3] = np.conj(B[kp])
    # ---- kernel A: per pair, rows S[k1][t] ----
    S = np.zeros((n_pairs, R, M), dtype=complex)
    for p in range(n_pairs):
        z = x[2 * p] + 1j * (x[2 * p + 1] if 2 * p + 1 < n_ch else 0)
        zz = z.reshape(R, M)                              # [j][t]
        D = np.einsum('jt,kj->kt', zz, w(2 * R, np.outer(2 * k1 + 1, j)))     # odd DFT over j
        u = D * tau
        S[p, :R // 2] = u[:R // 2]
        S[p, R // 2:] = np.conj(u[R // 2:]) * w(M, t)[None, :]
        # cross-check against the definition
        X = oddfft(z)
        for kk1 in (0, 1, R // 2, R - 1):
            assert np.allclose(np.fft.fft(u[kk1]), X[kk1 + R * np.arange(M)], atol=1e-8 * N)
        # upper rows carry the partner's twiddle: S[R-1-k1] = conj(D[R-1-k1]) tau[k1]
        for kk1 in range(R // 2):
            assert np.allclose(S[p, R - 1 - kk1], np.conj(D[R - 1 - kk1]) * tau[kk1], atol=1e-9 * R)
        if 2 * p + 1 >= n_ch:                              # real pair: upper stored row == lower row
            for kk1 in range(R // 2):
                assert np.allclose(S[p, R - 1 - kk1], S[p, kk1], atol=1e-9 * R)
    # ---- kernel B ----
    Wrows = np.zeros((R // 2, 2, M), dtype=complex)       # s1, s2 per row pair
    for ra in range(R // 2):
        rb = R - 1 - ra
        W1 = np.zeros(M, dtype=complex); W2 = np.zeros(M, dtype=complex)
        for p in range(n_pairs):
            Z1 = np.fft.fft(S[p, ra]); V = np.fft.fft(S[p, rb])
            W1 += Z1 * T[ra, p, :, 0] + V * T[ra, p, :, 1]
            W2 += V * T[ra, p, :, 2] + Z1 * T[ra, p, :, 3]
        Wrows[ra, 0] = np.fft.ifft(W1) * M
        Wrows[ra, 1] = np.fft.ifft(W2) * M
    # ---- kernel C ----
    g = np.zeros((R, M), dtype=complex)
    for ra in range(R // 2):
        g[ra] = np.conj(tau[ra]) * Wrows[ra, 0]
        g[R - 1 - ra] = np.conj(np.conj(tau[ra]) * Wrows[ra, 1])
    y = np.einsum('kt,jk->jt', g, np.conj(w(2 * R, np.outer(j, 2 * k1 + 1)))).reshape(N)
    v = slice(taps - 1, N)
    err = max(np.abs(y.real[v] - yl[v]).max(), np.abs(y.imag[v] - yr[v]).max()) / max(np.abs(yl).max(), np.abs(yr).max())
    print(f"R={R} M={M} taps={taps} C={n_ch}: max rel err {err:.2e}")
    assert err < 1e-10
    # ---- kernel A's two-level odd DFT: j = j1 + 8 j2, k1 = Ra kb + ka ----
    Ra = R // 8
    z = (x[0] + 1j * x[1 % n_ch]).reshape(R, M)[:, 5]     # one column t = 5
    D_ref = w(2 * R, np.outer(2 * k1 + 1, j)) @ z
    D2 = np.zeros(R, dtype=complex)
    for j1 in range(8):
        xs = z[j1 + 8 * np.arange(Ra)]                    # j2
        E = w(2 * Ra, np.outer(2 * np.arange(Ra) + 1, np.arange(Ra))) @ xs       # odd DFT-Ra over j2 -> ka
        E = E * w(2 * R, j1 * (2 * np.arange(Ra) + 1))    # omega_R^{j1 (ka + 1/2)}
        for kb in range(8):
            D2[Ra * kb + np.arange(Ra)] += E * w(8, j1 * kb)
    assert np.allclose(D2, D_ref, atol=1e-10 * R)
    # ---- kernel C's two-level inverse: k1 = Ra kb + ka, j = j1 + 8 j2 ----
    gcol = g[:, 7]
    y_ref = np.conj(w(2 * R, np.outer(j, 2 * k1 + 1))) @ gcol
    y2 = np.zeros(R, dtype=complex)
    for j1 in range(8):
        F = np.zeros(Ra, dtype=complex)
        for ka in range(Ra):
            F[ka] = sum(gcol[Ra * kb + ka] * np.conj(w(8, j1 * kb)) for kb in range(8))
        F = F * np.conj(w(2 * R, j1 * (2 * np.arange(Ra) + 1)))
        y2[j1 + 8 * np.arange(Ra)] = np.conj(w(2 * Ra, np.outer(np.arange(Ra), 2 * np.arange(Ra) + 1))) @ F
    assert np.allclose(y2, y_ref, atol=1e-10 * R)


check(16, 64, 100, 7)
check(32, 32, 200, 8)
check(128, 16, 300, 3)
check(16, 128, 50, 1)
print("ok")
