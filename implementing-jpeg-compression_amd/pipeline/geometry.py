"""Steps 0-3: the stages that only change the band's geometry or range, ahead of the transform
(behaviour of the reference's pipeline/padding.py, subsampling.py, dct_padding.py, normalization.py).

  0 Padding        edge-replicate to a multiple of block_size (the same object back when block_size is 1)
  1 SubSampling    mean over block_size x block_size tiles; up-sampling by replication on the way back
  2 DCTPadding     edge-replicate the sub-sampled band to a multiple of dct_size
  3 Normalization  nothing on the way in (there is NO -128 level shift); clamp to [0, 255] on the way back

The fused device paths fold 1 (any block_size), the clamp of 3 and the replication of 1 into the transform
kernels (jpegx_forward_fused_pooled / _u8, jpegx_mean_pool_f64, jpegx_inverse_fused_u8_inflated); these classes
are the stand-alone steps of the plugin API.  The four module names of the reference re-export them.
"""
import numpy as np

from util import inflate, pad_array, padded_size, split_into_blocks, undo_pad_array
from .base import AlgorithmStep


def band_geometry(config):
    """Sizes a band goes through: (rows, cols) as configured, after step 0, after step 1, after step 2."""
    bs, n = config.block_size, config.dct_size
    original = (config.height, config.width)
    padded = tuple(padded_size(v, bs) for v in original)
    pooled = tuple(v // bs for v in padded)
    blocked = tuple(padded_size(v, n) for v in pooled)
    return original, padded, pooled, blocked


def _grown(before, after):
    return after[0] - before[0], after[1] - before[1]


class Padding(AlgorithmStep):
    step_index = 0

    def execute(self, array):
        return array if self._config.block_size == 1 else pad_array(array, self._config.block_size)

    def invert(self, array):
        original, padded, _, _ = band_geometry(self._config)
        return undo_pad_array(array, _grown(original, padded))


class SubSampling(AlgorithmStep):
    step_index = 1

    def execute(self, array):
        tiles = split_into_blocks(array, self._config.block_size)
        return np.mean(tiles, axis=(2, 3))

    def invert(self, array):
        return inflate(array, self._config.block_size)


class DCTPadding(AlgorithmStep):
    step_index = 2

    def execute(self, array):
        return pad_array(array, self._config.dct_size)

    def invert(self, array):
        _, _, pooled, blocked = band_geometry(self._config)
        return undo_pad_array(array, _grown(pooled, blocked))


class Normalization(AlgorithmStep):
    step_index = 3

    def execute(self, array):
        return array

    def invert(self, array):
        np.clip(array, 0, 255, out=array)        # in place, like the reference's element loop
        return array
