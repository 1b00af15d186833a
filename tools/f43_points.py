"""Interpolation points for Winograd F(4x4,3x3) in fp32 (CPU study behind rerevst-code_amd/csrc/conv_f43.h).

The error of fp32 F(4x4,3x3) is dominated by the accumulation over the input channels IN THE TRANSFORM DOMAIN (the
stage table below), whose values exceed the outputs by the norms of the transforms.  Points balanced around 1 shrink those
norms: 0, +-3/4, +-3/2, inf cut the error 2.4x against the textbook 0, +-1, +-2, inf, with dyadic (exact) matrix entries.
Also checks the kernel's op sequences (the twelve-op input transform and the output transform) against B^T and A^T.

    python tools/f43_points.py > profiles/r04_f43_points.txt
"""
import numpy as np


def cook_toom(points, m=4, r=3):
    """A^T [m x n], G [n x r], B^T [n x n] of F(m, r) for the finite `points` plus the point at infinity (n = m + r - 1)."""
    n = m + r - 1
    a = [float(p) for p in points]
    AT = np.zeros((m, n)); G = np.zeros((n, r))
    for j, aj in enumerate(a):
        N = np.prod([aj - al for l, al in enumerate(a) if l != j])
        for i in range(m): AT[i, j] = aj ** i
        for k in range(r): G[j, k] = aj ** k / N
    AT[m - 1, n - 1] = 1.0; G[n - 1, r - 1] = 1.0
    # B^T from the identity A^T diag(G e_k) B^T = (shift by k), k = 0..r-1
    L = np.vstack([AT @ np.diag(G[:, k]) for k in range(r)])
    R = np.vstack([np.eye(m, n, k) for k in range(r)])
    BT = np.linalg.lstsq(L, R, rcond=None)[0]
    assert np.abs(L @ BT - R).max() < 1e-9
    return AT, G, BT


def error(points, stages="VMA", C=256, n=200, seed=5):
    """Mean over n random 6x6 patches (post-ReLU-like inputs, C channels) of max|y - direct| / rms(direct); `stages`: which of
    V (input transform), M (channel accumulation), A (output transform) run in float32 (the rest in float64)."""
    AT, G, BT = cook_toom(points)
    rng = np.random.default_rng(seed); errs = []
    f32, f64 = np.float32, np.float64
    for t in range(n):
        d = np.maximum(rng.normal(0.3, 1, (C, 6, 6)), 0).astype(f32); g = (rng.normal(0, 1, (C, 3, 3)) * 0.03).astype(f32)
        U = np.einsum('ia,cab,jb->cij', G, g.astype(f64), G).astype(f32).astype(f64)          # weights: transformed in double, rounded once
        tV = f32 if 'V' in stages else f64
        V = np.einsum('ia,cab->cib', BT.astype(tV), d.astype(tV)).astype(tV); V = np.einsum('cib,jb->cij', V, BT.astype(tV)).astype(tV)
        P = U * V.astype(f32).astype(f64)
        if 'M' in stages:
            M = np.zeros((6, 6), f32); Pf = P.astype(f32)
            for c in range(C): M = (M + Pf[c]).astype(f32)
            M = M.astype(f64)
        else:
            M = P.sum(0)
        tA = f32 if 'A' in stages else f64
        Y = (AT.astype(tA) @ M.astype(tA)).astype(tA); Y = (Y @ AT.T.astype(tA)).astype(tA)
        direct = np.array([[(d[:, i:i + 3, j:j + 3].astype(f64) * g).sum() for j in range(4)] for i in range(4)])
        errs.append(np.abs(Y - direct).max() / np.sqrt((direct ** 2).mean()))
    return float(np.mean(errs))


def kernel_sequences_match(a=0.75, b=1.5):
    """The op sequences of conv_f43.h reproduce B^T d and A^T m for the points 0, +-a, +-b, inf."""
    AT, G, BT = cook_toom([0, a, -a, b, -b])
    A2, B2, P, S = a * a, b * b, a * a * b * b, a * a + b * b
    rng = np.random.default_rng(0)
    d = rng.normal(size=6); d0, d1, d2, d3, d4, d5 = d
    A = d4 - B2 * d2; B = d3 - B2 * d1; C = d4 - A2 * d2; F = d3 - A2 * d1
    g = P * d0 + d4; t0 = g - S * d2; g = P * d1 + d5; t5 = g - S * d3
    t = np.array([t0, A + a * B, A - a * B, C + b * F, C - b * F, t5])
    assert np.allclose(t, BT @ d, atol=1e-12), (t, BT @ d)
    m = rng.normal(size=6); m0, m1, m2, m3, m4, m5 = m
    s1, e1, s2, e2 = m1 + m2, m1 - m2, m3 + m4, m3 - m4
    y = np.array([m0 + s1 + s2, a * e1 + b * e2, A2 * s1 + B2 * s2, a ** 3 * e1 + b ** 3 * e2 + m5])
    assert np.allclose(y, AT @ m, atol=1e-12)
    return AT, G, BT


if __name__ == "__main__":
    np.set_printoptions(precision=6, suppress=True, linewidth=150)
    print("# tools/f43_points.py — fp32 error of Winograd F(4x4,3x3) by interpolation points (mean over 200 random 6x6 patches, 256 channels,")
    print("# post-ReLU-like inputs, of max|y - direct| / rms(direct); weights transformed in double and rounded once)")
    for pts in ([0, 1, -1, 2, -2], [0, .5, -.5, 1, -1], [0, 1, -1, .5, -2], [0, 2 ** -.5, -2 ** -.5, 2 ** .5, -2 ** .5], [0, .75, -.75, 1.5, -1.5],
                [0, .6875, -.6875, 1.375, -1.375], [0, .625, -.625, 1.25, -1.25]):
        AT, G, BT = cook_toom(pts)
        print("points %-46s all fp32 %.2e | only V %.2e | only M (channel accumulation) %.2e | only A %.2e | none (operand storage) %.2e | max|B^T| %.2f max|A^T| %.2f"
              % (pts, error(pts), error(pts, "V"), error(pts, "M"), error(pts, "A"), error(pts, ""), np.abs(BT).max(), np.abs(AT).max()), flush=True)
    AT, G, BT = kernel_sequences_match()
    print("# shipped: 0, +-3/4, +-3/2, inf (all entries dyadic).  B^T =\n", BT, "\n# A^T =\n", AT, "\n# G =\n", G)
    print("# the kernel's twelve-op input transform and its output transform reproduce B^T d and A^T m: checked")
