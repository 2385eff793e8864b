"""Step 1: mean-pool sub-sampling (reference: pipeline/subsampling.py).

The fused GPU path folds this stage into the forward kernel for block_size 1, 2 and 4
(jpegx_forward_fused_pooled); this class is the stand-alone step.
"""
import numpy as np

from util import inflate, split_into_blocks
from .base import AlgorithmStep


class SubSampling(AlgorithmStep):
    step_index = 1

    def execute(self, array):
        return np.mean(split_into_blocks(array, self._config.block_size), axis=(2, 3))

    def invert(self, array):
        return inflate(array, self._config.block_size)
