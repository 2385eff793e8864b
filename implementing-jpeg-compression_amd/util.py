"""Array / bit utilities with the reference's names and behaviour (reference: util.py).

Host-side plumbing only: nothing here is on the GPU hot path (the kernels replace the block
splitting by index arithmetic).  The implementations are vectorised NumPy;
the reference's 3rd-party ``bitarray`` dependency is replaced by the small :class:`Bits` below.
"""
import math

import numpy as np


class BadArrayShapeError(Exception):
    """util.py:92-93"""


class EmptyArrayError(Exception):
    """util.py:96-97"""


class BadRleCodeError(Exception):
    """util.py:228-229"""


def inflate(a, factor):
    """Undo sub-sampling by pixel replication (util.py:6-14)."""
    return np.repeat(np.repeat(a, factor, axis=0), factor, axis=1)


def padded_size(size, factor):
    """Smallest multiple of ``factor`` that is >= size (util.py:100-101)."""
    return math.ceil(float(size) / factor) * factor


def pad_array(a, block_size):
    """Edge-replicate columns, then rows, up to a multiple of ``block_size`` (util.py:17-41)."""
    a = np.asarray(a)
    if a.ndim != 2:
        raise BadArrayShapeError()
    if a.shape[0] == 0 or a.shape[1] == 0:
        raise EmptyArrayError()
    extra_rows = padded_size(a.shape[0], block_size) - a.shape[0]
    extra_cols = padded_size(a.shape[1], block_size) - a.shape[1]
    if extra_rows == 0 and extra_cols == 0:
        return a
    return np.pad(a, ((0, extra_rows), (0, extra_cols)), mode="edge")


def undo_pad_array(a, padding):
    """Drop ``padding = (rows, cols)`` from the bottom / right (util.py:44-47)."""
    return a[:a.shape[0] - padding[0], :a.shape[1] - padding[1]]


def calculate_padding(a, factor):
    """(rows, cols) that pad_array would add (util.py:104-107)."""
    return (padded_size(a.shape[0], factor) - a.shape[0],
            padded_size(a.shape[1], factor) - a.shape[1])


def split_into_blocks(a, block_size):
    """(H/b, W/b, b, b) array of blocks, block [y, x] = a[y*b:(y+1)*b, x*b:(x+1)*b] (util.py:68-89)."""
    a = pad_array(a, block_size)
    hb, wb = a.shape[0] // block_size, a.shape[1] // block_size
    return a.reshape(hb, block_size, wb, block_size).swapaxes(1, 2).copy()


def band_to_array(band):
    """PIL band -> (height, width) integer array (util.py:110-112).  8-bit bands stay uint8 (the reference
    builds an int64 array from a Python list; same values, an eighth of the bytes to check and upload)."""
    a = np.asarray(band)
    if a.dtype != np.uint8:
        a = a.astype(np.int64)
    return a.reshape((band.height, band.width))


class Bits:
    """Minimal MSB-first bit string standing in for ``bitarray.bitarray`` (absent here).

    Only the operations the reference uses on bitarrays: construction from a '01' string,
    append/extend, ``+``, ``len``, slicing, ``to01``, ``tobytes`` (zero padded), ``frombytes``.
    """
    __slots__ = ("_s",)

    def __init__(self, init=None):
        if init is None:
            self._s = ""
        elif isinstance(init, Bits):
            self._s = init._s
        else:
            s = str(init)
            if s.strip("01"):
                raise ValueError("Bits accepts only '0'/'1' characters")
            self._s = s

    def append(self, bit):
        self._s += "1" if bit else "0"

    def extend(self, other):
        self._s += other._s if isinstance(other, Bits) else Bits(other)._s

    def __add__(self, other):
        return Bits(self._s + Bits(other)._s)

    def __len__(self):
        return len(self._s)

    def __getitem__(self, key):
        r = self._s[key]
        return Bits(r) if isinstance(key, slice) else r == "1"

    def __eq__(self, other):
        return isinstance(other, Bits) and self._s == other._s

    def __repr__(self):
        return "Bits('%s')" % self._s

    def to01(self):
        return self._s

    def tobytes(self):
        if not self._s:
            return b""
        pad = (-len(self._s)) % 8
        return int(self._s + "0" * pad, 2).to_bytes((len(self._s) + pad) // 8, "big")

    def frombytes(self, data):
        if data:
            self._s += bin(int.from_bytes(data, "big"))[2:].zfill(8 * len(data))


# the reference spells it ``bitarray``; expose the stand-in under that name too
bitarray = Bits


class BitEncoder:
    """Integer -> bit strings (util.py:115-131)."""

    def encode_unsigned(self, x):
        return Bits(self._to_bitstring(x))

    def encode_signed(self, x):
        # sign bit first: '1' for strictly positive, '0' otherwise (util.py:121-123)
        return Bits(("1" if x > 0 else "0") + self._to_bitstring(x))

    def pad_bitstring(self, bits, size=4):
        short = size - len(bits)
        return Bits("0" * short) + bits if short > 0 else bits

    def _to_bitstring(self, x):
        return bin(abs(x))[2:]


class RunLengthCode:
    """(run of zeros, bit size, amplitude) triple of the entropy stage (util.py:134-225).

    Quirks kept on purpose: a zero chain code stands for FIFTEEN zeros (not 16), and
    ``size = ceil(log2(|amplitude| + 1)) + 1`` includes the sign bit.
    """
    max_run_length = 15

    def __init__(self, run_length, size, amplitude=0):
        text = "({}, {}, {})".format(run_length, size, amplitude)
        bad = (
            (size == 0 and amplitude != 0)
            or not 0 <= run_length <= 15
            or not 0 <= size <= 15
            or (0 < run_length < 15 and size == 0 and amplitude == 0)
        )
        if bad:
            raise BadRleCodeError(text)
        self.run_length = run_length
        self.size = size
        self.amplitude = amplitude

    @staticmethod
    def EOB():
        return RunLengthCode(0, 0, 0)

    @staticmethod
    def all_zeros():
        return RunLengthCode(15, 0, 0)

    @staticmethod
    def encode(run_length, amplitude):
        chains, rest = divmod(run_length, RunLengthCode.max_run_length)
        bit_size = math.ceil(math.log2(abs(amplitude) + 1)) + 1
        return [RunLengthCode.all_zeros() for _ in range(chains)] + [RunLengthCode(rest, bit_size, amplitude)]

    def decode(self):
        if self.is_zeros_chain():
            return [0] * self.max_run_length
        return [0] * self.run_length + [self.amplitude]

    def is_zeros_chain(self):
        return self.run_length == self.max_run_length and self.size == 0 and self.amplitude == 0

    def is_EOB(self):
        return self.run_length == 0 and self.size == 0

    def as_tuple(self):
        if self.is_EOB():
            return 0, 0
        amplitude = self.amplitude if np.iscomplex(self.amplitude) else int(round(self.amplitude))
        return self.run_length, self.size, amplitude

    def as_bitsring(self):  # (sic) the reference's spelling, util.py:203
        if self.is_EOB():
            return Bits("0" * 8)
        enc = BitEncoder()
        out = Bits()
        out.extend(enc.pad_bitstring(enc.encode_unsigned(self.run_length)))
        out.extend(enc.pad_bitstring(enc.encode_unsigned(self.size)))
        if not self.is_zeros_chain():
            out.extend(enc.encode_signed(self.amplitude))
        return out

    as_bitstring = as_bitsring

    def __eq__(self, other):
        return (self.run_length, self.size, self.amplitude) == (other.run_length, other.size, other.amplitude)

    def __repr__(self):
        return "({}, {}, {})".format(self.run_length, self.size, self.amplitude)
