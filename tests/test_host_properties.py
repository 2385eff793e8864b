"""Property tests (hypothesis) for the host-side mirror: entropy stage round trips, the C++ decoder
against the Python encoder, bit-string stand-in, container, padding arithmetic.  CPU only."""
import numpy as np
from hypothesis import given, settings, strategies as st

import file_format
import util
from pipeline import CompressedData, Configuration, QuantizationMethod
from pipeline.rle_byte_stream import RleBytestream
from pipeline.run_length_encoding import RunLengthBlock, RunLengthEncoding

amps = st.integers(min_value=-16383, max_value=16383)
sparse_block = st.lists(st.one_of(st.just(0), st.just(0), st.just(0), amps), min_size=64, max_size=64)


@settings(max_examples=150, deadline=None)
@given(sparse_block)
def test_rle_block_round_trip(values):
    a = np.array(values)
    blk = RunLengthBlock(64)
    codes = blk.encode(a)
    assert blk.decode(codes).tolist() == a.tolist()
    assert codes[-1].is_EOB() and all(0 <= c.run_length <= 15 and 0 <= c.size <= 15 for c in codes)


@settings(max_examples=60, deadline=None)
@given(st.lists(sparse_block, min_size=1, max_size=6))
def test_bytestream_round_trip_and_cxx_decoder(blocks):
    import jpegx
    import oracle
    zz = np.array(blocks, dtype=np.int16).reshape(1, len(blocks), 64)
    tuples = RunLengthEncoding(None).execute(zz.astype(float))
    blob = RleBytestream(None).execute(tuples)
    assert RleBytestream(None).invert(blob) == tuples
    assert blob == oracle.rle_bytestream(zz)                       # oracle restatement agrees
    back = jpegx.entropy_decode(blob, len(blocks)).reshape(zz.shape)
    assert np.array_equal(back, zz)                                # libjpegx host decoder inverts it
    cfg = Configuration(width=8 * len(blocks), height=8, block_size=1)
    assert np.array_equal(RunLengthEncoding(cfg).invert(tuples), zz)


@settings(max_examples=200, deadline=None)
@given(st.text(alphabet="01", max_size=70))
def test_bits_bytes_round_trip(text):
    b = util.Bits(text)
    data = b.tobytes()
    assert len(data) == (len(text) + 7) // 8
    c = util.Bits()
    c.frombytes(data)
    assert c.to01() == text + "0" * ((-len(text)) % 8)
    assert (util.Bits(text[:3]) + util.Bits(text[3:])).to01() == text


@settings(max_examples=100, deadline=None)
@given(st.integers(1, 65535), st.integers(1, 65535), st.integers(1, 64), st.sampled_from([1, 2, 8, 16]),
       st.sampled_from(["DCT", "DFT"]), st.binary(max_size=40), st.binary(max_size=40), st.binary(max_size=40))
def test_container_round_trip(w, h, bs, dct, transform, y, cb, cr):
    q = QuantizationMethod("divide", divisor=7) if dct != 8 else QuantizationMethod("qtable")
    cfg = Configuration(width=w, height=h, block_size=bs, dct_size=dct, transform=transform, quantization=q)
    blob = file_format.generate_data(cfg, CompressedData(y, cb, cr))
    cfg2, data = file_format.read_data(blob)
    assert (cfg2.width, cfg2.height, cfg2.block_size, cfg2.dct_size, cfg2.transform) == (w, h, bs, dct, transform)
    assert cfg2.quantization.name == q.name and cfg2.quantization.params == q.params
    assert (data.y, data.cb, data.cr) == (y, cb, cr)


@settings(max_examples=100, deadline=None)
@given(st.integers(1, 40), st.integers(1, 40), st.integers(1, 9))
def test_padding_and_blocks_shapes(h, w, b):
    a = np.arange(h * w).reshape(h, w)
    p = util.pad_array(a, b)
    assert p.shape == (util.padded_size(h, b), util.padded_size(w, b))
    assert np.array_equal(p[:h, :w], a) and np.all(p[h:, :w] == a[-1:, :]) and np.all(p[:, w:] == p[:, w - 1:w])
    assert util.undo_pad_array(p, util.calculate_padding(a, b)).shape == a.shape
    blocks = util.split_into_blocks(a, b)
    assert blocks.shape == (p.shape[0] // b, p.shape[1] // b, b, b)
    assert np.array_equal(blocks[0, 0], p[:b, :b])
