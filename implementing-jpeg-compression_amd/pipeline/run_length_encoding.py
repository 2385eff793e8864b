"""Step 7: run-length coding of the zigzag stream, whole plane at a time with NumPy.

Behaviour follows the reference's pipeline/run_length_encoding.py (its per-block Python loops are
replaced by array operations): per block, every non-zero value becomes ``(zeros before it, bit
size, value)``, runs longer than fifteen are split off as ``(15, 0, 0)`` chain codes that stand for
FIFTEEN zeros, and the block ends with the end-of-block pair ``(0, 0)``.  The device form of this
stage is csrc/jpegx_entropy.hip; this module is the host-side step class the plugin API exposes.
"""
import numpy as np

from util import BadRleCodeError, RunLengthCode, padded_size
from .base import AlgorithmStep

CHAIN = RunLengthCode.max_run_length        # zeros covered by one (15, 0, 0) code


def _integers(values):
    """What the reference stores in its integer work array: np.round, then the real part as int."""
    a = np.round(np.asarray(values))
    if np.iscomplexobj(a):
        a = a.real
    return a.astype(np.int64)


def _bit_sizes(amplitudes):
    """ceil(log2(|a| + 1)) + 1 for integer a != 0, i.e. bit_length(|a|) plus the sign bit."""
    rest = np.abs(amplitudes)
    size = np.ones(rest.shape, dtype=np.int64)
    while rest.any():
        size += rest > 0
        rest = rest >> 1
    return size


def _plane_codes(blocks):
    """Run-length codes of an (nblocks, n) integer array as flat arrays (run, size, amplitude) in
    stream order plus a mask telling which entries are end-of-block markers."""
    nblocks = blocks.shape[0]
    blk, idx = np.nonzero(blocks)                       # row-major: block by block, ascending index
    amp = blocks[blk, idx]
    first = np.ones(blk.shape, dtype=bool)              # first non-zero of its block
    first[1:] = blk[1:] != blk[:-1]
    prev = np.where(first, -1, np.concatenate(([0], idx[:-1])))
    chains, run = np.divmod(idx - prev - 1, CHAIN)
    size = _bit_sizes(amp)
    if size.size and size.max() > 15:
        k = int(np.argmax(size))
        raise BadRleCodeError("({}, {}, {})".format(int(run[k]), int(size[k]), int(amp[k])))
    per_value = chains + 1                              # a value's chain codes and the value itself
    per_block = np.bincount(blk, weights=per_value, minlength=nblocks).astype(np.int64) + 1   # + end marker
    block_end = np.cumsum(per_block) - 1
    total = int(per_block.sum())
    out_run = np.full(total, CHAIN, dtype=np.int64)     # whatever is not written below is a chain code
    out_size = np.zeros(total, dtype=np.int64)
    out_amp = np.zeros(total, dtype=np.int64)
    # a value sits behind: the entries of all earlier blocks (markers included), the entries of the
    # earlier values of its own block, and its own chain codes
    pos = np.cumsum(per_value) - 1 + blk
    out_run[pos], out_size[pos], out_amp[pos] = run, size, amp
    is_end = np.zeros(total, dtype=bool)
    is_end[block_end] = True
    out_run[block_end] = 0
    return out_run, out_size, out_amp, is_end


class RunLengthBlock:
    """One block at a time (the unit the reference's tests exercise)."""

    def __init__(self, block_size):
        self._size = block_size

    def encode(self, zigzag_array):
        run, size, amp, is_end = _plane_codes(_integers(zigzag_array).reshape(1, -1))
        return [RunLengthCode.EOB() if e else RunLengthCode(r, s, a)
                for r, s, a, e in zip(run.tolist(), size.tolist(), amp.tolist(), is_end.tolist())]

    def decode(self, rle_block):
        values = []
        for code in rle_block:
            if code.is_EOB():
                values.extend([0] * (self._size - len(values)))
                break
            values.extend(code.decode())
        return np.array(values)


class RunLengthEncoding(AlgorithmStep):
    step_index = 7

    def execute(self, array):
        array = np.asarray(array)
        blocks = _integers(array).reshape(array.shape[0] * array.shape[1], array.shape[2])
        run, size, amp, is_end = _plane_codes(blocks)
        return [(0, 0) if e else (r, s, a)
                for r, s, a, e in zip(run.tolist(), size.tolist(), amp.tolist(), is_end.tolist())]

    def invert(self, tuples_list):
        n = self._config.dct_size ** 2
        hb, wb = self._height_in_blocks(), self._width_in_blocks()
        codes = [t if len(t) == 3 else (t[0], t[1], 0) for t in tuples_list]
        if not codes:
            return np.array([]).reshape((hb, wb, n))
        run, size, amp = (np.array(col) for col in zip(*codes))
        bad = ((size == 0) & (amp != 0)) | (run < 0) | (run > 15) | (size < 0) | (size > 15) | \
              ((run > 0) & (run < 15) & (size == 0) & (amp == 0))
        if bad.any():
            k = int(np.argmax(bad))
            raise BadRleCodeError("({}, {}, {})".format(run[k], size[k], amp[k]))
        is_end = (run == 0) & (size == 0)
        is_chain = (run == CHAIN) & (size == 0)
        covers = np.where(is_end, 0, np.where(is_chain, CHAIN, run + 1))      # coefficients a code stands for
        nblocks = int(is_end.sum())                                            # codes behind the last marker are dropped
        block = np.cumsum(is_end) - is_end                                     # block a code belongs to
        reach = np.cumsum(covers)
        base = np.concatenate(([0], reach[np.flatnonzero(is_end)]))            # coefficients emitted before each block
        offset = reach - base[np.minimum(block, nblocks)]                      # 1-based end offset inside the block
        inside = block < nblocks
        if (offset[inside] > n).any():
            raise ValueError("a run-length block holds more than %d coefficients" % n)
        keep = inside & ~is_end & ~is_chain
        flat = np.zeros(nblocks * n, dtype=amp.dtype)
        flat[block[keep] * n + offset[keep] - 1] = amp[keep]
        return flat.reshape((hb, wb, n))

    def _blocks_along(self, extent):
        cfg = self._config
        sub = padded_size(extent, cfg.block_size) // cfg.block_size
        return padded_size(sub, cfg.dct_size) // cfg.dct_size

    def _height_in_blocks(self):
        return self._blocks_along(self._config.height)

    def _width_in_blocks(self):
        return self._blocks_along(self._config.width)
