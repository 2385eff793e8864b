"""Step 5: per-block quantisation (reference: pipeline/quantization.py).

With dct_size 8 and real data the whole plane is quantised by one exact float64 kernel launch
(jpegx_quantize_f64 / jpegx_restore_f64); otherwise the configured quantiser object is applied
block by block on the host, as in the reference.
"""
import numpy as np

from .base import AlgorithmStep

_MODES = ("none", "discard", "divide", "qtable")


class Quantization(AlgorithmStep):
    step_index = 5

    def _gpu_args(self, array):
        """(mode, param) when the plane can go through libjpegx (stock quantiser objects only), else None."""
        cfg = self._config
        if (cfg.dct_size != 8 or array.ndim != 2 or array.size == 0 or array.shape[0] % 8 or array.shape[1] % 8
                or array.dtype.kind not in "fiu" or cfg.quantization.name not in _MODES):
            return None
        return cfg.quantization.gpu_mode()

    def _run(self, array, gpu_fn, attr):
        array = np.asarray(array)
        args = self._gpu_args(array)
        if args is not None:
            import jpegx
            return getattr(jpegx, gpu_fn)(array.astype(np.float64), *args).astype(array.dtype)
        res = np.zeros(array.shape, dtype=array.dtype)
        self.apply_blockwise(array, getattr(self._config.quantization.quantizer, attr), self._config.dct_size, res)
        return res

    def execute(self, array):
        return self._run(array, "quantize_f64", "quantize")

    def invert(self, array):
        return self._run(array, "restore_f64", "restore")
