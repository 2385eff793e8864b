"""DCT matrices and the ``DCT`` class with the reference's interface (reference: transforms.py).

Size 8 -- the codec's hot path -- runs on the GPU through libjpegx's exact float64 kernels and
is bit-identical to the reference's float64 results; the 8x8 constant tables are the
reference's own values (tests/golden/tables.npz -> include/jpegx_tables.inc).  Other sizes are
outside the accelerated path and are evaluated with plain NumPy products.
"""
import numpy as np


def dct_matrix(size):
    """Un-normalised DCT-II matrix C[k, n] = cos(pi/N (n + 1/2) k) (transforms.py:4-11)."""
    n = np.arange(size, dtype=np.float64)
    out = np.empty((size, size), dtype=np.float64)
    for k in range(size):
        # same expression order as the reference so the doubles agree to the last bit
        out[k, :] = np.cos(np.array([np.pi / size * (m + 0.5) * k for m in n]))
    return out


def dct_matrix_normalized(size):
    """Rows of dct_matrix scaled to unit length (transforms.py:14-20)."""
    w = dct_matrix(size)
    for k in range(size):
        w[k] /= np.linalg.norm(w[k])
    return w


def normalization_matrix(size):
    """diag(1 / ||row_k||) (transforms.py:23-26)."""
    return np.diag(1.0 / np.linalg.norm(dct_matrix(size), axis=1))


class DCT:
    """1-D / 2-D DCT pair (transforms.py:29-75): forward ``C x``, inverse ``Cn^T (D^-1 x)``."""

    def __init__(self, size):
        self._size = size
        self._dct_matrix = dct_matrix(size)
        self._dct_normalized = dct_matrix_normalized(size)
        self._normalization_matrix = normalization_matrix(size)

    def transform_1d(self, x):
        assert x.ndim == 1
        return self._dct_matrix.dot(x)

    def transform_1d_inverse(self, x):
        assert x.ndim == 1
        return self._dct_normalized.transpose().dot(self._normalization_matrix.dot(x))

    def transform_2d(self, a):
        """Rows first, then columns: C A C^T (transforms.py:46-58)."""
        assert a.ndim == 2
        assert a.shape[0] == a.shape[1]
        if self._gpu_block(a):
            import jpegx
            return jpegx.dct8x8_f64(np.asarray(a, dtype=np.float64))
        rows = self._each_row(np.asarray(a), self.transform_1d)
        return self._each_row(rows.T, self.transform_1d).T

    def transform_2d_inverse(self, a):
        """Columns first, then rows (transforms.py:60-69)."""
        assert a.ndim == 2
        assert a.shape[0] == a.shape[1]
        if self._gpu_block(a):
            import jpegx
            return jpegx.idct8x8_f64(np.asarray(a, dtype=np.float64), do_round=False)
        cols = self._each_row(np.asarray(a).T, self.transform_1d_inverse).T
        return self._each_row(cols, self.transform_1d_inverse)

    def _gpu_block(self, a):
        return self._size == 8 and a.shape == (8, 8) and not np.iscomplexobj(a)

    @staticmethod
    def _each_row(matrix, transformation):
        out = np.zeros(matrix.shape)
        for i, row in enumerate(matrix):
            out[i] = transformation(row)
        return out
