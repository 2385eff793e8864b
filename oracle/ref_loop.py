"""Faithful Python/NumPy per-block loop of steps 4-6 -- TEST INFRASTRUCTURE ONLY.

This is the "what the reference costs on a CPU" leg of ``bench.py``'s ``cpu_baseline``
(kind "port"): it keeps the reference's per-block call structure so that its speed is
representative of the reference's own pure-Python pipeline --

  * step 4: per block, 8 + 8 ``ndarray.dot`` 8x8 . 8 mat-vecs (transforms.py:46-75),
    after a ``split_into_blocks``-style copy of the plane (util.py:68-89, base.py:58-72);
  * step 5: per block ``np.round(block * (1.0 / q))`` (quantizers.py:47-49);
  * step 6: per block a 64-element Python gather (pipeline/zigzag_order.py:11-14);
  * three separate passes over the plane, float64 throughout.

Its floating-point order is whatever the host's BLAS does, so it is NOT the checker
(``jpegx_oracle.c`` is); tests only require it to agree with the checker after rounding
on tie-free data.
"""
import numpy as np

from . import tables as _tables


def _blocks(a, n):
    """(H/n, W/n, n, n) copy of the plane, then row-major enumeration (y outer, x inner)."""
    h, w = a.shape[0] // n, a.shape[1] // n
    tiles = a.reshape(h, n, w, n).swapaxes(1, 2).copy()
    for y in range(h):
        for x in range(w):
            yield tiles[y, x], y, x


def _rows_through(mat, c):
    res = np.zeros(mat.shape)
    for i in range(mat.shape[0]):
        res[i] = c.dot(mat[i])
    return res


def dct_pass(plane, c):
    out = np.zeros(plane.shape, dtype=float)
    for blk, y, x in _blocks(plane, 8):
        m = _rows_through(blk, c)
        out[8 * y:8 * y + 8, 8 * x:8 * x + 8] = _rows_through(m.T, c).T
    return out


def quant_pass(coeffs, q):
    out = np.zeros(coeffs.shape, dtype=coeffs.dtype)
    for blk, y, x in _blocks(coeffs, 8):
        out[8 * y:8 * y + 8, 8 * x:8 * x + 8] = np.round(blk * (1.0 / q))
    return out


def zigzag_pass(quant, order):
    pairs = [(int(n) >> 3, int(n) & 7) for n in order]
    out = np.zeros((quant.shape[0] // 8, quant.shape[1] // 8, 64), dtype=quant.dtype)
    for blk, y, x in _blocks(quant, 8):
        out[y, x] = np.array([blk[i, j] for i, j in pairs])
    return out


def forward_qtable(plane):
    """Steps 4+5+6 with the JPEG table on a float64 plane -> (H/8, W/8, 64) float64."""
    t = _tables()
    a = np.asarray(plane, dtype=np.float64)
    return zigzag_pass(quant_pass(dct_pass(a, t["dct_matrix"]), t["qtable"]), t["zigzag8"])
