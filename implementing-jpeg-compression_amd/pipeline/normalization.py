"""Step 3: identity on the way in (NO -128 level shift), clamp to [0, 255] on the way out
(reference: pipeline/normalization.py).  The clamp is also available fused into the inverse
kernel (JPEGX_F_CLAMP_U8)."""
import numpy as np

from .base import AlgorithmStep


class Normalization(AlgorithmStep):
    step_index = 3

    def execute(self, array):
        return array

    def invert(self, array):
        # in place, like the reference's element loop (normalization.py:10-14)
        np.clip(array, 0, 255, out=array)
        return array
