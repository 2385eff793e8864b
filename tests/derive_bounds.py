#!/usr/bin/env python3
"""Rigorous first-order rounding-error weights of the fp32 fast-tier transforms, derived mechanically.

The fused kernels decide per coefficient (forward) / per row (inverse) whether the fp32 result can be trusted by
comparing its distance to a rounding boundary with an a-priori bound on |fp32 value - float64 reference|.  Those
bounds are sums over the inputs, u * sum_i W[o][i] |x_i|, with weights W that depend only on the straight-line
program.  This script runs such a program SYMBOLICALLY: every intermediate carries

    coef[i]   its exact value as a linear form in the inputs,
    err[i]    a bound on its accumulated rounding error, in units of u = 2^-24, per unit of |x_i|,
    exact     whether it is an integer-valued quantity small enough that adds and subtractions of it are exact in
              fp32 (8-bit samples and their sums: the PIXEL promise),

with the standard model fl(a op b) = (a op b)(1 + d), |d| <= u:

    add / sub      err = err_a + err_b (+ |coef| unless both operands are `exact`)
    mul by k       err = |k| err_a + |k coef_a| [rounding of the constant, unless k is a power of two]
                                   + |k coef_a| [rounding of the product, unless k is a power of two]
    fma a*k + b    err = |k| err_a + err_b + |k coef_a| [constant] + |coef| [the one rounding of the result]

(first order in u; the kernels add a 2^-10 relative margin for the second-order terms).  A two-pass (separable)
transform composes as  W2D[(k,l)][(i,j)] = |C2[k][i]| W1[l][j] + W2[k][i] |C1[l][j]|.

`python tests/derive_bounds.py` prints the tables that csrc/jpegx_math.h carries (jpegx_aan_fwd_F, jpegx_aan_inv_w and
the scale factors folded into the quantiser tables); tests/test_host_properties.py checks that the header still
matches what this derivation gives, and tests/test_emul.py checks the bounds against observed errors on adversarial
blocks.
"""
import math

import numpy as np


class V:
    """one intermediate of the straight-line program"""
    __slots__ = ("coef", "err", "exact")

    def __init__(self, coef, err, exact):
        self.coef, self.err, self.exact = coef, err, exact


def inputs(n, exact):
    return [V(np.eye(n)[i], np.zeros(n), exact) for i in range(n)]


def add(a, b, sign=1.0):
    coef = a.coef + sign * b.coef
    ex = a.exact and b.exact
    return V(coef, a.err + b.err + (0.0 if ex else np.abs(coef)), ex)


def sub(a, b):
    return add(a, b, -1.0)


def _pow2(k):
    m, _ = math.frexp(abs(k))
    return m == 0.5


def mul(a, k):
    coef = a.coef * k
    if _pow2(k):
        return V(coef, abs(k) * a.err, False)
    return V(coef, abs(k) * a.err + 2.0 * np.abs(coef), False)


def fma(a, k, b):
    coef = a.coef * k + b.coef
    const = 0.0 if _pow2(k) else np.abs(a.coef * k)
    return V(coef, abs(k) * a.err + b.err + const + np.abs(coef), False)


C = [math.cos(m * math.pi / 16) for m in range(8)]      # C[m] = cos(m pi / 16)


# ---- the forward 8-point transform of the kernels: csrc/jpegx_math.h jpegx_dct8_aan_f32 -------------------------
def dct8_aan(x):
    """Arai-Agui-Nakajima flow graph with the multiply-adds fused (30 operations); output k is g[k] times the
    un-normalised DCT-II coefficient of transforms.py:4-11."""
    t0, t7 = add(x[0], x[7]), sub(x[0], x[7])
    t1, t6 = add(x[1], x[6]), sub(x[1], x[6])
    t2, t5 = add(x[2], x[5]), sub(x[2], x[5])
    t3, t4 = add(x[3], x[4]), sub(x[3], x[4])
    t10, t13 = add(t0, t3), sub(t0, t3)
    t11, t12 = add(t1, t2), sub(t1, t2)
    y = [None] * 8
    y[0], y[4] = add(t10, t11), sub(t10, t11)
    s = add(t12, t13)
    y[2], y[6] = fma(s, C[4], t13), fma(s, -C[4], t13)
    a10, a11, a12 = add(t4, t5), add(t5, t6), add(t6, t7)
    z5 = mul(sub(a10, a12), C[6])
    z2 = fma(a10, C[2] - C[6], z5)
    z4 = fma(a12, C[2] + C[6], z5)
    z11, z13 = fma(a11, C[4], t7), fma(a11, -C[4], t7)
    y[5], y[3] = add(z13, z2), sub(z13, z2)
    y[1], y[7] = add(z11, z4), sub(z11, z4)
    return y


def dct8_plain(x):
    """the round-1/2 even/odd form (jpegx_dct8_f32), for cross-checking the hand-derived rounding counts"""
    s = [add(x[i], x[7 - i]) for i in range(4)]
    d = [sub(x[i], x[7 - i]) for i in range(4)]
    e0, e1, e2, e3 = add(s[0], s[3]), add(s[1], s[2]), sub(s[0], s[3]), sub(s[1], s[2])
    y = [None] * 8
    y[0] = add(e0, e1)
    y[4] = mul(sub(e0, e1), C[4])
    y[2] = fma(e3, C[6], mul(e2, C[2]))
    y[6] = fma(e3, -C[2], mul(e2, C[6]))
    y[1] = fma(d[3], C[7], fma(d[2], C[5], fma(d[1], C[3], mul(d[0], C[1]))))
    y[3] = fma(d[3], -C[5], fma(d[2], -C[1], fma(d[1], -C[7], mul(d[0], C[3]))))
    y[5] = fma(d[3], C[3], fma(d[2], C[7], fma(d[1], -C[1], mul(d[0], C[5]))))
    y[7] = fma(d[3], -C[1], fma(d[2], C[3], fma(d[1], -C[5], mul(d[0], C[7]))))
    return y


def true_dct():
    return np.array([[math.cos(math.pi / 8 * (n + 0.5) * k) for n in range(8)] for k in range(8)])


def forward_tables(flow=dct8_aan):
    """(g, F_pixel, F_generic): g[k] = scale of output k; F[k*8+l] = the bound's factor for coefficient (k, l):
    |t32 - t64| <= F u S |rq'| with S = sum |x|, rq' the fp32 multiplier the kernel uses (the quantiser's reciprocal
    with the scales folded in): max over samples of the 2-D weight, plus max |coef| for the rounding of rq' itself."""
    T = true_dct()
    out = {}
    for pixel in (True, False):
        row = flow(inputs(8, pixel))                      # pass 1 works on the samples
        g = np.array([np.max(np.abs(row[k].coef)) / np.max(np.abs(T[k])) for k in range(8)])
        for k in range(8):
            assert np.allclose(row[k].coef, g[k] * T[k], rtol=0, atol=1e-12), k
        F = np.zeros(64)
        for l in range(8):
            # pass 2 works on the pass-1 outputs of column l: exact integers only for l = 0 with pixel input
            col = flow(inputs(8, pixel and row[l].exact))
            for k in range(8):
                w2d = np.abs(col[k].coef)[:, None] * row[l].err[None, :] + col[k].err[:, None] * np.abs(row[l].coef)[None, :]
                c2d = np.abs(col[k].coef)[:, None] * np.abs(row[l].coef)[None, :]
                F[k * 8 + l] = w2d.max() + c2d.max()
        out[pixel] = F
    return g, out[True], out[False]


# ---- the inverse 8-point transform of the kernels: jpegx_idct8_aan_f32 --------------------------------------------
def idct8_aan(X):
    """AAN inverse flow graph with fused multiply-adds; input k must be pre-scaled by h[k] (folded into the
    dequantisation multiplier), output n = the sample x_n of transforms.py:40-44."""
    t10, t11 = add(X[0], X[4]), sub(X[0], X[4])
    t13 = add(X[2], X[6])
    t12 = fma(sub(X[2], X[6]), 2 * C[4], mul(t13, -1.0))
    e0, e3 = add(t10, t13), sub(t10, t13)
    e1, e2 = add(t11, t12), sub(t11, t12)
    z13, z10 = add(X[5], X[3]), sub(X[5], X[3])
    z11, z12 = add(X[1], X[7]), sub(X[1], X[7])
    o7 = add(z11, z13)
    p11 = mul(sub(z11, z13), 2 * C[4])
    z5 = mul(add(z10, z12), 2 * C[2])
    p10 = fma(z12, -2 * (C[2] - C[6]), z5)
    p12 = fma(z10, -2 * (C[2] + C[6]), z5)
    o6 = sub(p12, o7)
    o5 = sub(p11, o6)
    o4 = sub(p10, o5)
    x = [None] * 8
    x[0], x[7] = add(e0, o7), sub(e0, o7)
    x[1], x[6] = add(e1, o6), sub(e1, o6)
    x[2], x[5] = add(e2, o5), sub(e2, o5)
    x[3], x[4] = add(e3, o4), sub(e3, o4)
    return x


def idct8_plain(X):
    """the even/odd inverse of the kernels (jpegx_idct8_f32): x = X0 / 8 + 1/4 sum_k C[k][n] X_k"""
    q = 0.25
    p0, p4 = mul(X[0], 0.125), mul(X[4], q * C[4])
    g0, g1 = add(p0, p4), sub(p0, p4)
    h0 = fma(X[6], q * C[6], mul(X[2], q * C[2]))
    h1 = fma(X[6], -q * C[2], mul(X[2], q * C[6]))
    E0, E1, E2, E3 = add(g0, h0), add(g1, h1), sub(g1, h1), sub(g0, h0)
    O0 = fma(X[7], q * C[7], fma(X[5], q * C[5], fma(X[3], q * C[3], mul(X[1], q * C[1]))))
    O1 = fma(X[7], -q * C[5], fma(X[5], -q * C[1], fma(X[3], -q * C[7], mul(X[1], q * C[3]))))
    O2 = fma(X[7], q * C[3], fma(X[5], q * C[7], fma(X[3], -q * C[1], mul(X[1], q * C[5]))))
    O3 = fma(X[7], -q * C[1], fma(X[5], q * C[3], fma(X[3], -q * C[5], mul(X[1], q * C[7]))))
    x = [None] * 8
    x[0], x[7] = add(E0, O0), sub(E0, O0)
    x[1], x[6] = add(E1, O1), sub(E1, O1)
    x[2], x[5] = add(E2, O2), sub(E2, O2)
    x[3], x[4] = add(E3, O3), sub(E3, O3)
    return x


def inverse_plain_weights():
    """w[k*8+l]: the weight of |d_kl| (the dequantised coefficient) in the bound u sum_kl w |d_kl| on |x32 - x64| for
    the even/odd inverse, columns (index k) then rows (index l), without the dequantisation's own roundings (those
    add dq s_k s_l, s_0 = 1/8, s_k = 1/4: jpegx_inv_weight)."""
    M = true_idct()
    one = idct8_plain(inputs(8, False))
    A = np.array([one[n].coef for n in range(8)])
    assert np.allclose(A, M, rtol=0, atol=1e-12)
    E = np.array([one[n].err for n in range(8)])
    W = np.zeros(64)
    for k in range(8):
        for l in range(8):
            w2d = np.abs(A[:, l])[None, :] * E[:, k][:, None] + E[:, l][None, :] * np.abs(A[:, k])[:, None]
            W[k * 8 + l] = w2d.max()
    return W


def true_idct():
    """x = Cn^T (Dinv X) = X0 / 8 + 1/4 sum_{k>=1} C[k][n] X_k (transforms.py:40-44)"""
    T = true_dct()
    M = np.zeros((8, 8))                                  # M[n][k]
    for n in range(8):
        for k in range(8):
            M[n][k] = (0.125 if k == 0 else 0.25) * T[k][n]
    return M


def inverse_tables():
    """(h, w0, w2): h[k] = pre-scale of input k; w[k*8+l] = the weight of |v_kl| (the PRE-SCALED dequantised
    coefficient the kernel holds) in the bound sum_kl w |v_kl| u on |x32 - x64|, for 0 and 2 roundings in the
    dequantisation multiply (dq: the fp32 rounding of the multiplier and of the product)."""
    M = true_idct()
    one = idct8_aan(inputs(8, False))
    A = np.array([one[n].coef for n in range(8)])         # A[n][k]: what the flow graph computes per unit of input k
    h = np.array([np.max(np.abs(M[:, k])) / np.max(np.abs(A[:, k])) for k in range(8)])
    for k in range(8):
        assert np.allclose(A[:, k] * h[k], M[:, k], rtol=0, atol=1e-12), k
    E = np.array([one[n].err for n in range(8)])          # E[n][k]
    w = {}
    for dq in (0, 2):
        W = np.zeros(64)
        for k in range(8):                                # column pass index k, then row pass index l
            for l in range(8):
                # sample (i, j): |A[i][k]| E[j][l] (pass-1 error through pass 2) ... pass 1 = columns (index k), pass 2 = rows (index l)
                w2d = np.abs(A[:, l])[None, :] * E[:, k][:, None] + E[:, l][None, :] * np.abs(A[:, k])[:, None]
                c2d = np.abs(A[:, k])[:, None] * np.abs(A[:, l])[None, :]
                W[k * 8 + l] = w2d.max() + dq * c2d.max()
        w[dq] = W
    return h, w[0], w[2]


QTABLE = [16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
          18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99]


def quantise_levels(F, g, nlevels=15):
    """The kernels keep E * level in a register per LEVEL, not per coefficient: round every F up to one of `nlevels`
    values chosen (dynamic programme over the sorted F) to lose as little as possible, weighted by how much a
    coefficient's bound matters under the JPEG table (1 / (g_k g_l q)).  Returns (levels, index per coefficient)."""
    gg = np.outer(g, g).ravel()
    wt = 1.0 / (gg * np.array(QTABLE, dtype=float))
    order = np.argsort(F)
    f, w = F[order], wt[order]
    n = len(f)
    # cost[a][b] = loss when items a..b-1 all take level f[b-1]
    INF = 1e300
    best = [[INF] * (nlevels + 1) for _ in range(n + 1)]
    back = [[0] * (nlevels + 1) for _ in range(n + 1)]
    best[0][0] = 0.0
    cw = np.concatenate([[0.0], np.cumsum(w)])
    cwf = np.concatenate([[0.0], np.cumsum(w * f)])
    for b in range(1, n + 1):
        for m in range(1, nlevels + 1):
            for a in range(m - 1, b):
                if best[a][m - 1] >= INF:
                    continue
                loss = f[b - 1] * (cw[b] - cw[a]) - (cwf[b] - cwf[a])
                c = best[a][m - 1] + loss
                if c < best[b][m]:
                    best[b][m], back[b][m] = c, a
    m = min(range(1, nlevels + 1), key=lambda mm: best[n][mm])
    cuts, b = [], n
    while b > 0:
        cuts.append(b)
        b, m = back[b][m], m - 1
    cuts = sorted(cuts)
    levels = [float(f[c - 1]) for c in cuts]
    idx = np.zeros(64, dtype=int)
    for pos, item in enumerate(order):
        idx[item] = next(i for i, c in enumerate(cuts) if pos < c)
    return levels, idx


def main():
    g, Fp, Fg = forward_tables()
    for name, F in (("pixel", Fp), ("generic", Fg)):
        levels, idx = quantise_levels(F, g)
        up = [math.ceil(v * 64 - 1e-7) / 64 for v in levels]          # rounded up to multiples of 1/64: exact in fp32
        print("levels %s:" % name, ", ".join("%.6ff" % v for v in up))
        print("level index %s:" % name, ", ".join(str(int(i)) for i in idx))
    print("forward scale g[k] (AAN output k = g[k] * DCT coefficient k):")
    print("  ", ", ".join("%.17g" % v for v in g))
    print("F, pixel input (ceil):", [int(math.ceil(v - 1e-9)) for v in Fp])
    print("F, generic input (ceil):", [int(math.ceil(v - 1e-9)) for v in Fg])
    _, Pp, Pg = forward_tables(dct8_plain)
    print("plain even/odd form, pixel (exact):", [round(float(v), 2) for v in Pp])
    wp = inverse_plain_weights()
    print("inverse (even/odd form) weights, rounded up to multiples of 2^-12:")
    print("  ", ", ".join("%.10ff" % (math.ceil(v * 4096 - 1e-9) / 4096) for v in wp))
    h, w0, w2 = inverse_tables()
    print("inverse pre-scale h[k]:")
    print("  ", ", ".join("%.17g" % v for v in h))
    print("inverse weights, dq = 0:", [round(float(v), 4) for v in w0])
    print("inverse weights, dq = 2:", [round(float(v), 4) for v in w2])


if __name__ == "__main__":
    main()
