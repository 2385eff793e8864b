"""Step 8: bit packing of the run-length codes (reference: pipeline/rle_byte_stream.py).

4-bit run, 4-bit size, then a sign bit ('1' = positive) and the magnitude bits; EOB is a zero
byte and the stream is padded to a byte boundary after every EOB.  Uses util.Bits in place of
the reference's 3rd-party bitarray."""
from util import Bits, RunLengthCode
from .base import AlgorithmStep


class BitDecoder:
    def __init__(self, array):
        self._array = array
        self._pos = 0

    def read(self, n):
        chunk = self._array[self._pos:self._pos + n]
        self._pos += n
        return chunk

    def read_quad(self):
        return self.read(4)

    def decode_unsigned(self, n):
        return int(self.read(n).to01(), base=2)

    def decode_signed(self, n):
        text = self.read(n).to01()
        magnitude = int(text[1:], base=2)
        return magnitude if text[0] == "1" else -magnitude

    def skip_padding(self):
        self._pos += (-self._pos) % 8

    def is_end(self):
        return self._pos >= len(self._array)


class RleBytestream(AlgorithmStep):
    step_index = 8

    def execute(self, tuples_list):
        pieces = []
        nbits = 0
        for t in tuples_list:
            code = RunLengthCode(*t)
            text = code.as_bitsring().to01()
            pieces.append(text)
            nbits += len(text)
            if code.is_EOB():
                pad = (-nbits) % 8
                pieces.append("0" * pad)
                nbits += pad
        return Bits("".join(pieces)).tobytes()

    def invert(self, bytestream):
        bits = Bits()
        bits.frombytes(bytestream)
        return [code.as_tuple() for code in self._codes(bits)]

    def _pad_bitarray(self, a):
        while len(a) % 8 > 0:
            a.append(False)

    def _codes(self, bits):
        decoder = BitDecoder(bits)
        while not decoder.is_end():
            run_len = decoder.decode_unsigned(4)
            size = decoder.decode_unsigned(4)
            if run_len == 0 and size == 0:
                decoder.skip_padding()
                yield RunLengthCode.EOB()
            elif run_len == 15 and size == 0:
                yield RunLengthCode(15, 0, 0)
            else:
                yield RunLengthCode(run_len, size, decoder.decode_signed(size))
