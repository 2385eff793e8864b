"""Container format of the codec (reference: file_format.py).

Layout, little endian: <H header length, <H width, <H height, <H block_size, <H dct_size,
3 ASCII bytes transform, <H json length, quantiser JSON; then for Y, Cb, Cr: <L length + bytes.
"""
import struct

import pipeline

_SHORT = struct.Struct("<H")
_LONG = struct.Struct("<L")


def pack_integer(value):
    return _SHORT.pack(value)


def unpack_integer(bytestream):
    return _SHORT.unpack(bytestream)[0]


def pack_long(value):
    return _LONG.pack(value)


def unpack_long(bytestream):
    return _LONG.unpack(bytestream)[0]


def pack_string(s):
    return bytes(s, encoding="ascii")


def unpack_string(bytestream):
    return bytestream.decode()


class Reader:
    """Sequential reader over a bytes object (file_format.py:5-19)."""

    def __init__(self, seq):
        self._seq = seq
        self._index = 0

    def read(self, n):
        chunk = self._seq[self._index:self._index + n]
        self._index += n
        return chunk

    def read_short(self):
        return self.read(2)

    def read_long(self):
        return self.read(4)


def create_header(config):
    quant_json = config.quantization.to_json()
    fields = [config.width, config.height, config.block_size, config.dct_size]
    body = b"".join(pack_integer(v) for v in fields) + pack_string(config.transform) + \
        pack_integer(len(quant_json)) + pack_string(quant_json)
    return pack_integer(2 + 13 + len(quant_json)) + body


def get_header(bytestream):
    r = Reader(bytestream)
    unpack_integer(r.read_short())                       # header length (unused here)
    width, height, block_size, dct_size = (unpack_integer(r.read_short()) for _ in range(4))
    transform = unpack_string(r.read(3))
    quant = pipeline.QuantizationMethod.from_json(unpack_string(r.read(unpack_integer(r.read_short()))))
    return pipeline.Configuration(width=width, height=height, block_size=block_size, dct_size=dct_size,
                                  transform=transform, quantization=quant)


def generate_data(config, compressed_data):
    out = create_header(config)
    for band in (compressed_data.y, compressed_data.cb, compressed_data.cr):
        out += pack_long(len(band)) + band
    return out


def read_data(bytestream):
    config = get_header(bytestream)
    r = Reader(bytestream)
    r.read(unpack_integer(r.read_short()) - 2)           # skip the rest of the header
    bands = [r.read(unpack_long(r.read_long())) for _ in range(3)]
    return config, pipeline.CompressedData(*bands)
