"""Step 6: zigzag scan of every block (reference: pipeline/zigzag_order.py).

``Zigzag`` builds the scan order for any block size (for 8 it is the standard JPEG order).
``ZigzagOrder`` turns an (H, W) plane into (H/n, W/n, n*n); for dct_size 8 the permutation is
done by the GPU (jpegx_zigzag / jpegx_unzigzag), otherwise by one NumPy gather.
"""
import numpy as np

from util import BadArrayShapeError
from .base import AlgorithmStep


class Zigzag:
    def __init__(self, block_size):
        self._size = block_size
        self._indices = None

    @property
    def zigzag_indices(self):
        """[(i, j), ...]: anti-diagonals d = i + j in turn; even d runs bottom-left -> top-right,
        odd d the other way (pipeline/zigzag_order.py:27-43,55-79)."""
        if self._indices is None:
            n = self._size
            order = []
            for d in range(2 * n - 1):
                cells = [(i, d - i) for i in range(max(0, d - n + 1), min(d, n - 1) + 1)]
                order.extend(cells if d % 2 else cells[::-1])
            self._indices = order
        return self._indices

    def flat_indices(self):
        return np.array([i * self._size + j for i, j in self.zigzag_indices], dtype=np.intp)

    def zigzag_order(self, block):
        self._validate_block(block)
        return np.asarray(block).reshape(-1)[self.flat_indices()]

    def restore(self, zigzag_array):
        self._validate_zigzag(zigzag_array)
        block = np.zeros(self._size * self._size, dtype=zigzag_array.dtype)
        block[self.flat_indices()] = zigzag_array
        return block.reshape(self._size, self._size)

    def _validate_block(self, a):
        if not (a.ndim == 2 and a.shape[0] == a.shape[1] == self._size):
            raise BadArrayShapeError(a.shape)

    def _validate_zigzag(self, zigzag_array):
        if not (zigzag_array.ndim == 1 and zigzag_array.shape[0] == self._size ** 2):
            raise BadArrayShapeError(zigzag_array.shape)


def _gpu_ok(n, array):
    return n == 8 and array.size > 0 and array.dtype.itemsize in (2, 4, 8, 16) and array.dtype.kind in "fiuc"


class ZigzagOrder(AlgorithmStep):
    step_index = 6

    def execute(self, array):
        n = self._config.dct_size
        array = np.asarray(array)
        hb, wb = array.shape[0] // n, array.shape[1] // n
        if _gpu_ok(n, array) and array.shape == (hb * 8, wb * 8):
            import jpegx
            return jpegx.zigzag(array)
        tiles = array[:hb * n, :wb * n].reshape(hb, n, wb, n).swapaxes(1, 2).reshape(hb, wb, n * n)
        return tiles[:, :, Zigzag(n).flat_indices()]

    def invert(self, array):
        n = self._config.dct_size
        array = np.asarray(array)
        hb, wb = array.shape[0], array.shape[1]
        if _gpu_ok(n, array) and array.ndim == 3 and array.shape[2] == 64:
            import jpegx
            return jpegx.unzigzag(array)
        z = Zigzag(n)
        if array.ndim != 3 or array.shape[2] != n * n:
            raise BadArrayShapeError(array.shape)
        tiles = np.zeros((hb, wb, n * n), dtype=array.dtype)
        tiles[:, :, z.flat_indices()] = array
        return tiles.reshape(hb, wb, n, n).swapaxes(1, 2).reshape(hb * n, wb * n)
