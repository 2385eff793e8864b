"""Step 2: pad the sub-sampled band to a multiple of dct_size (reference: pipeline/dct_padding.py)."""
from util import pad_array, padded_size, undo_pad_array
from .base import AlgorithmStep


class DCTPadding(AlgorithmStep):
    step_index = 2

    def execute(self, array):
        return pad_array(array, self._config.dct_size)

    def invert(self, array):
        cfg = self._config
        sub_w = padded_size(cfg.width, cfg.block_size) // cfg.block_size
        sub_h = padded_size(cfg.height, cfg.block_size) // cfg.block_size
        return undo_pad_array(array, (padded_size(sub_h, cfg.dct_size) - sub_h,
                                      padded_size(sub_w, cfg.dct_size) - sub_w))
