"""Step 0: pad the band to a multiple of the sub-sampling block (reference: pipeline/padding.py)."""
from util import pad_array, undo_pad_array
from .base import AlgorithmStep


class Padding(AlgorithmStep):
    step_index = 0

    def execute(self, array):
        bs = self._config.block_size
        if bs == 1:
            return array          # the very same object, as in the reference (padding.py:9-10)
        return pad_array(array, bs)

    def invert(self, array):
        extra_rows, extra_cols = self.calculate_padding(self._config.block_size)
        return undo_pad_array(array, (extra_rows, extra_cols))
