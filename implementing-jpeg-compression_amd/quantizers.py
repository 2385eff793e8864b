"""The four quantisers with the reference's interface (reference: quantizers.py).

``quantize`` / ``restore`` accept whatever the reference's accept.  Real-valued 2-D arrays whose
sides are multiples of 8 (single 8x8 blocks or whole planes) are processed by libjpegx's exact
float64 kernels, which tile the per-block rule over the array exactly like
pipeline.quantization.Quantization does; any other shape or dtype (the reference's own unit
tests use 2x2, 3x3, 1-D and complex inputs) is element-wise host arithmetic with the same
formulae.
"""
import numpy as np


def _gpu_eligible(a):
    return (isinstance(a, np.ndarray) and a.ndim == 2 and a.size > 0 and a.shape[0] % 8 == 0
            and a.shape[1] % 8 == 0 and a.dtype.kind == "f")


def _on_gpu(fn_name, a, mode, param):
    import jpegx
    return getattr(jpegx, fn_name)(a.astype(np.float64), mode, param)


class RoundingQuantizer:
    """round(a), identity restore (quantizers.py:4-9)."""

    def quantize(self, a):
        a = np.asarray(a)
        if type(self) is RoundingQuantizer and _gpu_eligible(a):
            return _on_gpu("quantize_f64", a, "none", 0.0)
        return np.round(a)

    def restore(self, a):
        return a


class DiscardingQuantizer(RoundingQuantizer):
    """Round, then keep only the top-left keep x keep coefficients (quantizers.py:12-20)."""

    def __init__(self, keep=2):
        self.keep = keep

    def quantize(self, a):
        a = np.asarray(a)
        if type(self) is DiscardingQuantizer and a.shape == (8, 8) and _gpu_eligible(a) and \
                isinstance(self.keep, (int, np.integer)) and self.keep >= 0:
            return _on_gpu("quantize_f64", a, "discard", float(self.keep))
        out = np.round(a)
        out[self.keep:] = 0
        out[:, self.keep:] = 0
        return out


class DivisionQuantizer(RoundingQuantizer):
    """round(a / divisor), restore a * divisor (quantizers.py:23-31)."""

    def __init__(self, divisor=40):
        self.divisor = divisor

    def quantize(self, a):
        a = np.asarray(a)
        if type(self) is DivisionQuantizer and _gpu_eligible(a) and self.divisor != 0:
            return _on_gpu("quantize_f64", a, "divide", float(self.divisor))
        return np.round(a / float(self.divisor))

    def restore(self, a):
        return a * self.divisor


# the JPEG Annex-K luminance table (reference: quantizers.py:35-42), kept flat here
_LUMINANCE = (16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55,
              14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
              18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92,
              49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99)


class JpegQuantizationTable(RoundingQuantizer):
    """Standard luminance table: round(a * (1.0 / q)), restore round(a * q) (quantizers.py:34-53)."""
    table = [list(_LUMINANCE[r * 8:(r + 1) * 8]) for r in range(8)]  # nested list, as callers expect

    def __init__(self):
        self._qtable = np.array(self.table)

    def _stock(self):
        """The device kernels hold the standard table: an edited one stays on the host."""
        return type(self) is JpegQuantizationTable and self._qtable.shape == (8, 8) and \
            np.array_equal(self._qtable.ravel(), _LUMINANCE)

    def quantize(self, a):
        a = np.asarray(a)
        if a.shape == (8, 8) and _gpu_eligible(a) and self._stock():
            return _on_gpu("quantize_f64", a, "qtable", 0.0)
        return np.round(a * (1.0 / self._qtable))

    def restore(self, a):
        a = np.asarray(a)
        if a.shape == (8, 8) and _gpu_eligible(a) and self._stock():
            return _on_gpu("restore_f64", a, "qtable", 0.0)
        return np.round(a * self._qtable)
