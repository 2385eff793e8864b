"""Step 7: per-block run-length coding of the zigzag stream (reference:
pipeline/run_length_encoding.py).  Host-side entropy stage, outside the GPU hot path
(SURVEY.md 8(f)-2): every block ends with EOB, zero chains are 15 long."""
import numpy as np

from util import RunLengthCode, padded_size
from .base import AlgorithmStep


class RunLengthBlock:
    def __init__(self, block_size):
        self._size = block_size

    def non_zeros(self, a):
        for i in np.flatnonzero(a):
            yield a[i], int(i)

    def encode(self, zigzag_array):
        values = np.zeros(zigzag_array.shape, dtype=int)
        values[:] = np.round(zigzag_array)
        codes = []
        last = -1
        for value, index in self.non_zeros(values):
            codes.extend(RunLengthCode.encode(index - last - 1, value))
            last = index
        codes.append(RunLengthCode.EOB())
        return codes

    def decode(self, rle_block):
        out = []
        for code in rle_block:
            if code.is_EOB():
                out.extend([0] * (self._size - len(out)))
                break
            out.extend(code.decode())
        return np.array(out)


class RunLengthEncoding(AlgorithmStep):
    step_index = 7

    def execute(self, array):
        coder = RunLengthBlock(block_size=array.shape[2])
        codes = []
        for i in range(array.shape[0]):
            for j in range(array.shape[1]):
                codes.extend(coder.encode(array[i, j]))
        return [c.as_tuple() for c in codes]

    def invert(self, tuples_list):
        n2 = self._config.dct_size ** 2
        coder = RunLengthBlock(block_size=n2)
        flat = []
        for block in self._rle_blocks(tuples_list):
            flat.extend(coder.decode(block))
        return np.array(flat).reshape((self._height_in_blocks(), self._width_in_blocks(), n2))

    def _blocks_along(self, extent):
        cfg = self._config
        sub = padded_size(extent, cfg.block_size) // cfg.block_size
        return padded_size(sub, cfg.dct_size) // cfg.dct_size

    def _height_in_blocks(self):
        return self._blocks_along(self._config.height)

    def _width_in_blocks(self):
        return self._blocks_along(self._config.width)

    def _rle_blocks(self, tuples_list):
        block = []
        for t in tuples_list:
            code = RunLengthCode(*t)
            block.append(code)
            if code.is_EOB():
                yield block
                block = []
