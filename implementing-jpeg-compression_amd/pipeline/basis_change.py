"""Step 4: per-block change of basis (reference: pipeline/basis_change.py).

transform 'DCT' with dct_size 8 -- the hot path -- runs on the GPU (libjpegx float64 kernels,
bit-identical to the reference's float64 output).  Other DCT sizes and the 'DFT' option are
outside the accelerated path and are evaluated on the host.
"""
import numpy as np

from transforms import DCT
from .base import AlgorithmStep


def _tiles(a, n):
    return a.reshape(a.shape[0] // n, n, a.shape[1] // n, n).swapaxes(1, 2)


def _untile(t):
    hb, wb, n, _ = t.shape
    return t.swapaxes(1, 2).reshape(hb * n, wb * n)


class BasisChange(AlgorithmStep):
    step_index = 4

    def _on_gpu(self, array):
        return (self._config.transform == "DCT" and self._config.dct_size == 8
                and array.ndim == 2 and array.shape[0] % 8 == 0 and array.shape[1] % 8 == 0
                and array.size > 0 and not np.iscomplexobj(array))

    def execute(self, array):
        transform, n = self._config.transform, self._config.dct_size
        array = np.asarray(array)
        if self._on_gpu(array):
            import jpegx
            return jpegx.dct8x8_f64(array.astype(np.float64))
        if transform == "DCT":
            res = np.zeros(array.shape, dtype=float)
            self.apply_blockwise(array, DCT(n).transform_2d, n, res)
        elif transform == "DFT":
            res = _untile(np.fft.fft2(_tiles(array, n), axes=(2, 3)))
        else:  # the reference falls through to an unbound local (basis_change.py:15-26)
            raise UnboundLocalError("local variable 'res' referenced before assignment")
        return res

    def invert(self, array):
        transform, n = self._config.transform, self._config.dct_size
        array = np.asarray(array)
        if self._on_gpu(array):
            import jpegx
            return jpegx.idct8x8_f64(array.astype(np.float64), do_round=True).astype(int)
        res = np.zeros(array.shape, dtype=float)
        if transform == "DCT":
            self.apply_blockwise(array, DCT(n).transform_2d_inverse, n, res)
        elif transform == "DFT":
            # like the reference, the imaginary part is dropped when stored into a float array
            res[:] = _untile(np.fft.ifft2(_tiles(array, n), axes=(2, 3))).real
        return np.array(np.round(res), dtype=int)
