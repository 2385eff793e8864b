"""Step 0: pad the band to a multiple of the sub-sampling block (reference: pipeline/padding.py)."""
from util import pad_array, undo_pad_array
from .base import AlgorithmStep


class Padding(AlgorithmStep):
    step_index = 0

    def execute(self, array):
        # the reference hands the input object back untouched when block_size is 1 (padding.py:9-10)
        return array if self._config.block_size == 1 else pad_array(array, self._config.block_size)

    def invert(self, array):
        return undo_pad_array(array, self.calculate_padding(self._config.block_size))
