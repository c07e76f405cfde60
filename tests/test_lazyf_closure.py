"""The packed Smith-Waterman kernel (k_sw_byte1p) replaces the reference's lazy-F loop for one-segment queries
(alignment/StripedSmithWaterman.cpp:418-448) by its closed form, a decayed max-scan.  This checks the identity the kernel
relies on: the literal loop -- including its early exit -- leaves exactly
    H[k] = max(H0[k], max_{d>=1} (H0[k-d] - gapOpen - (d-1) * gapExtend))
on random, sparse, monotone and single-peak columns for several gap settings (u8 saturating arithmetic as in the AVX2 code)."""
import numpy as np


def subs(a, b):
    return np.maximum(a - b, 0)


def lazy_loop(h0, g_open, g_ext):
    h, n = h0.copy(), len(h0)
    f = np.zeros(n, dtype=np.int64)
    f[1:] = subs(h0, g_open)[:-1]                    # vF = shiftl(H - gapOpen)
    steps = 0
    while np.any(subs(f, subs(h, g_open)) != 0):     # movemask(vF > vH - gapOpen) != 0
        h = np.maximum(h, f)
        nf = subs(f, g_ext)
        f = np.zeros(n, dtype=np.int64)
        f[1:] = nf[:-1]
        steps += 1
        assert steps <= n
    return h, steps


def closure(h0, g_open, g_ext):
    # the kernel's scan: S[k] = max_d (A[k-d] - d*gapExtend), A = H0 - gapOpen; H = max(H0, S shifted by one lane)
    a = subs(h0, g_open)
    s = a.copy()
    for k in range(1, len(a)):
        s[k] = max(s[k], s[k - 1] - g_ext)
    g = np.zeros_like(s)
    g[1:] = s[:-1]
    return np.maximum(h0, g)


def test_lazy_f_loop_equals_decayed_max_scan():
    rng = np.random.default_rng(20240917)
    gaps = [(16, 2), (11, 1), (16, 1), (5, 2), (3, 3), (20, 1), (1, 1)]
    longest = 0
    for trial in range(40000):
        g_open, g_ext = gaps[trial % len(gaps)]
        mode = trial % 5
        if mode == 0:
            h0 = rng.integers(0, 256, 32)
        elif mode == 1:
            h0 = rng.integers(0, 60, 32)
        elif mode == 2:
            h0 = rng.integers(0, 20, 32)
            h0[rng.integers(0, 32)] = rng.integers(30, 256)
        elif mode == 3:
            h0 = np.sort(rng.integers(0, 256, 32))
            if trial & 1:
                h0 = h0[::-1].copy()
        else:
            h0 = rng.integers(0, 256, 32) * (rng.random(32) < 0.3)
        h0 = h0.astype(np.int64)
        got, steps = lazy_loop(h0, g_open, g_ext)
        longest = max(longest, steps)
        assert np.array_equal(got, closure(h0, g_open, g_ext)), (g_open, g_ext, h0.tolist())
    assert longest >= 20          # the inputs do exercise long propagation chains
