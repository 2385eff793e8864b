"""Step 8: the run-length codes as a byte stream (behaviour of the reference's
pipeline/rle_byte_stream.py + util.RunLengthCode.as_bitsring, done with NumPy bit arrays).

Layout per code: 4-bit run, 4-bit size, then -- unless the code is a ``(15, 0, 0)`` zero chain -- a sign
bit ('1' = strictly positive) followed by ``|amplitude|`` in binary without leading zeros.  The end
of a block is one zero byte and the stream is padded with zero bits to the next byte boundary after
it, so every block starts on a byte.  The device form of this stage is csrc/jpegx_entropy.hip.
"""
import numpy as np

from util import BadRleCodeError
from .base import AlgorithmStep

_ROOM = 40          # bits of room behind the header byte in the scratch bit matrix


def _validated(tuples_list):
    """(run, size, amplitude) columns of the code list; the same rejections as util.RunLengthCode."""
    codes = [t if len(t) == 3 else (t[0], t[1], 0) for t in tuples_list]
    for r, s, a in codes:
        if (s == 0 and a != 0) or not 0 <= r <= 15 or not 0 <= s <= 15 or (0 < r < 15 and s == 0 and a == 0):
            raise BadRleCodeError("({}, {}, {})".format(r, s, a))
    if not codes:
        return (np.zeros(0, dtype=np.int64),) * 3
    run, size, amp = (np.array(col) for col in zip(*codes))
    if np.iscomplexobj(amp):
        amp = amp.real
    return run.astype(np.int64), size.astype(np.int64), np.round(amp).astype(np.int64)


class RleBytestream(AlgorithmStep):
    step_index = 8

    def execute(self, tuples_list):
        run, size, amp = _validated(tuples_list)
        if run.size == 0:
            return b""
        is_end = (run == 0) & (size == 0)
        has_amp = ~is_end & ~((run == 15) & (size == 0))
        mag = np.abs(amp)
        digits = np.ones(mag.shape, dtype=np.int64)     # binary digits of |amplitude|, at least one
        rest = mag >> 1
        while rest.any():
            digits += rest > 0
            rest = rest >> 1
        if digits.max() + 1 > _ROOM:
            raise BadRleCodeError("amplitude too wide: {}".format(int(mag.max())))
        nbits = 8 + np.where(has_amp, 1 + digits, 0)
        # one row of bits per code: header byte, sign, magnitude (MSB first), zeros behind
        word = ((run << 4) | size) << _ROOM
        word |= np.where(has_amp, (((amp > 0).astype(np.int64) << digits) | mag) << (_ROOM - 1 - digits), 0)
        rows = np.unpackbits(word.astype(">u8").view(np.uint8).reshape(-1, 8), axis=1)[:, 64 - 8 - _ROOM:]
        # where each code starts: running total of bits, rounded up to a byte behind each end marker
        block = np.cumsum(is_end) - is_end
        bits_of_block = np.bincount(block, weights=nbits).astype(np.int64)
        padded = (bits_of_block + 7) // 8 * 8
        start = (np.cumsum(padded) - padded)[block] + (np.cumsum(nbits) - nbits) - (np.cumsum(bits_of_block) - bits_of_block)[block]
        stream = np.zeros(int(padded.sum()), dtype=np.uint8)
        col = np.arange(rows.shape[1])
        live = col[None, :] < nbits[:, None]
        stream[(start[:, None] + col[None, :])[live]] = rows[live]
        return np.packbits(stream).tobytes()

    def invert(self, bytestream):
        data = bytes(bytestream)
        out = []
        window, have, pos = 0, 0, 0                 # upcoming bits as an int, how many of them, next byte to load

        def take(n):
            """(value, bits there were): a read that runs past the end returns the bits that are left -- the reference's
            bitarray slices come back short (rle_byte_stream.py:20-24) -- and int('', base=2) of none at all raises."""
            nonlocal window, have, pos
            while have < n and pos < len(data):
                window = (window << 8) | data[pos]
                pos += 1
                have += 8
            got = min(n, have)
            have -= got
            value = window >> have
            window &= (1 << have) - 1
            return value, got

        while pos < len(data) or have > 0:
            (run, got_run), (size, got_size) = take(4), take(4)
            if got_run == 0 or got_size == 0:
                raise ValueError("the stream ends inside a code's header")
            if run == 0 and size == 0:
                window, have = 0, 0                 # drop the padding: the next block starts on a byte
                out.append((0, 0))
            elif run == 15 and size == 0:
                out.append((15, 0, 0))
            else:
                bits, got = take(size)
                if got <= 1:                        # the reference fails here too: int('', base=2) (no amplitude bits)
                    raise ValueError("run-length code ({}, {}) has no amplitude bits".format(run, size))
                magnitude = bits & ((1 << (got - 1)) - 1)
                out.append((run, size, magnitude if bits >> (got - 1) else -magnitude))
        return out
