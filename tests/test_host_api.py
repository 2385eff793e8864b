"""Host-side mirror of the reference's API: the known answers of the reference's own unit tests
(/root/reference/tests/*.py) restated against our modules.  Everything here is host logic that
needs no GPU (sizes other than the accelerated 8x8 path, entropy stage, container, registry)."""
import numpy as np
import pytest

import file_format
import pipeline
import quantizers
import transforms
import util
from pipeline import Configuration, CompressedData, QuantizationMethod
from pipeline.base import AlgorithmStep, MissingStepIndexError, step_classes
from pipeline.rle_byte_stream import RleBytestream
from pipeline.run_length_encoding import RunLengthBlock, RunLengthEncoding
from pipeline.subsampling import SubSampling
from pipeline.zigzag_order import Zigzag, ZigzagOrder
from util import BadArrayShapeError, BadRleCodeError, EmptyArrayError, RunLengthCode


# ---- step registry (pipeline/base.py:4-31) ---------------------------------------------------
def test_registry_is_sorted_and_complete():
    assert [c.step_index for c in step_classes] == list(range(9))
    assert [c.__name__ for c in step_classes] == [
        "Padding", "SubSampling", "DCTPadding", "Normalization", "BasisChange", "Quantization",
        "ZigzagOrder", "RunLengthEncoding", "RleBytestream"]


def test_new_step_registers_and_missing_index_raises():
    before = list(step_classes)
    try:
        class Extra(AlgorithmStep):
            step_index = 2.5
        assert Extra in step_classes and step_classes.index(Extra) == 3
        with pytest.raises(MissingStepIndexError):
            class NoIndex(AlgorithmStep):
                pass
    finally:
        step_classes[:] = before


# ---- util (tests/util_tests.py, tests/padding_tests.py) --------------------------------------
def test_split_into_blocks_order_and_padding():
    blocks = util.split_into_blocks(np.arange(16).reshape(4, 4), block_size=2)
    assert blocks.shape == (2, 2, 2, 2)
    assert [blocks[y, x].ravel().tolist() for y in range(2) for x in range(2)] == [
        [0, 1, 4, 5], [2, 3, 6, 7], [8, 9, 12, 13], [10, 11, 14, 15]]
    small = util.split_into_blocks(np.array([[20], [10]]), block_size=3)
    assert small.shape == (1, 1, 3, 3) and small[0, 0].tolist() == [[20, 20, 20], [10, 10, 10], [10, 10, 10]]
    assert util.split_into_blocks(np.array([[3 - 2j]]), 1)[0, 0].ravel().tolist() == [3 - 2j]


def test_split_into_blocks_rejects_bad_input():
    with pytest.raises(BadArrayShapeError):
        util.split_into_blocks(np.array([32, 31]), block_size=2)
    with pytest.raises(BadArrayShapeError):
        util.split_into_blocks(np.array([[[32]]]), block_size=2)
    with pytest.raises(EmptyArrayError):
        util.split_into_blocks(np.array([[]]), block_size=3)


def test_pad_array_and_padded_size():
    assert util.pad_array(np.array([[20], [10]]), 3).tolist() == [[20, 20, 20], [10, 10, 10], [10, 10, 10]]
    same = np.array([[20, 3], [10, 9]])
    assert util.pad_array(same, 2).tolist() == same.tolist()
    assert [util.padded_size(n, 3) for n in (3, 4, 5, 6, 7)] == [3, 6, 6, 6, 9]
    assert util.inflate(np.array([[1, 2]]), 2).tolist() == [[1, 1, 2, 2], [1, 1, 2, 2]]


# ---- sub-sampling (tests/subsample_tests.py) --------------------------------------------------
def test_subsampling_means():
    a = np.array([[1, 2, 2, 1], [3, 2, 8, 1], [0, 0, 2, 2], [0, 4, 2, 2]])
    cfg = Configuration(width=123, height=854, block_size=2, dct_size=2)
    assert SubSampling(cfg).execute(a).tolist() == [[2, 3], [1, 2]]
    cfg = Configuration(width=123, height=854, block_size=4, dct_size=2)
    assert SubSampling(cfg).execute(a).tolist() == [[2]]


# ---- quantisers (tests/quantization_tests.py) -------------------------------------------------
def test_quantizers_known_answers():
    q = quantizers.RoundingQuantizer()
    assert np.allclose(q.quantize(np.array([[3.4, 8.0], [0, 0.6]])), [[3, 8], [0, 1]])
    assert np.allclose(q.quantize(np.array([[1.7j, 3j], [0j, 0.6 + 1j]])), [[2j, 3j], [0j, 1 + 1j]])
    d = quantizers.DiscardingQuantizer(2).quantize(np.arange(9).reshape(3, 3))
    assert d.tolist() == [[0, 1, 0], [3, 4, 0], [0, 0, 0]]
    m = quantizers.DivisionQuantizer(40)
    assert m.quantize(np.array([80, 24, 169])).tolist() == [2, 1, 4]
    assert m.restore(np.array([2, 1, 4])).tolist() == [80, 40, 160]
    assert quantizers.JpegQuantizationTable.table[0][:4] == [16, 11, 10, 16]


def test_quantization_method_and_configuration_errors():
    assert QuantizationMethod("divide", divisor=93).params == {"divisor": 93}
    rt = QuantizationMethod.from_json(QuantizationMethod("discard", keep=3).to_json())
    assert rt.name == "discard" and rt.quantizer.keep == 3
    with pytest.raises(pipeline.BadQuantizationError):
        QuantizationMethod("bogus")
    with pytest.raises(pipeline.BadQuantizationError):
        QuantizationMethod("divide", nonsense=1)
    with pytest.raises(pipeline.BadQuantizationError):
        Configuration(width=8, height=8, dct_size=4, quantization=QuantizationMethod("qtable"))
    assert Configuration(width=8, height=8).quantization.name == "none"
    assert Configuration(width=8, height=8).block_size == 2


# ---- DCT class for sizes off the accelerated path (tests/basis_change_tests.py) ---------------
def test_dct_round_trips_other_sizes():
    a = np.round(255 * np.cos(np.arange(100.0)))
    d = transforms.DCT(100)
    assert np.allclose(a, d.transform_1d_inverse(d.transform_1d(a)), rtol=0.01)
    b = np.array([[1, 2], [3, 4]])
    d2 = transforms.DCT(2)
    assert np.allclose(b, d2.transform_2d_inverse(d2.transform_2d(b)), rtol=0.01)


def test_dct_tables_match_the_reference(tables):
    assert np.array_equal(transforms.dct_matrix(8), tables["dct_matrix"])
    assert np.array_equal(transforms.dct_matrix_normalized(8), tables["dct_normalized"])
    assert np.array_equal(np.diag(transforms.normalization_matrix(8)), tables["norm_diag"])


# ---- zigzag (tests/zigzag_tests.py) -----------------------------------------------------------
def test_zigzag_known_orders(tables):
    assert Zigzag(4).zigzag_order(np.arange(16).reshape(4, 4)).tolist() == \
        [0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15]
    assert Zigzag(3).zigzag_order(np.arange(9).reshape(3, 3)).tolist() == [0, 1, 3, 6, 4, 2, 5, 7, 8]
    assert Zigzag(8).flat_indices().tolist() == tables["zigzag8"].tolist()
    a = np.arange(16).reshape(4, 4)
    assert Zigzag(4).restore(Zigzag(4).zigzag_order(a)).tolist() == a.tolist()


def test_zigzag_malformed_arrays():
    with pytest.raises(BadArrayShapeError):
        Zigzag(3).zigzag_order(np.arange(12).reshape(3, 4))
    with pytest.raises(BadArrayShapeError):
        Zigzag(3).zigzag_order(np.arange(12))
    with pytest.raises(BadArrayShapeError):
        Zigzag(3).zigzag_order(np.arange(16).reshape(4, 4))
    with pytest.raises(BadArrayShapeError):
        Zigzag(4).restore(np.arange(16).reshape(4, 4))
    with pytest.raises(BadArrayShapeError):
        Zigzag(4).restore(np.arange(23))


def test_zigzag_step_small_blocks():
    cfg = Configuration(width=4, height=4, block_size=1, dct_size=2)
    res = ZigzagOrder(cfg).execute(np.arange(16).reshape(4, 4))
    assert res.shape == (2, 2, 4)
    assert res.tolist() == [[[0, 1, 4, 5], [2, 3, 6, 7]], [[8, 9, 12, 13], [10, 11, 14, 15]]]
    cfg = Configuration(width=8, height=4, block_size=1, dct_size=2)
    for a in (np.arange(32).reshape(4, 8), np.arange(32).reshape(4, 8) * 2j):
        step = ZigzagOrder(cfg)
        assert step.invert(step.execute(a)).tolist() == a.tolist()


# ---- run-length stage (tests/RLE_tests.py) ----------------------------------------------------
def test_rle_block_known_answers():
    a = np.array([-15, 0, 0, 0, 3, 2, 0, 0, 0, 0, 120, 0, 0, 0, 0])
    blk = RunLengthBlock(block_size=a.shape[0])
    res = blk.encode(a)
    assert res[:4] == [RunLengthCode(0, 5, -15), RunLengthCode(3, 3, 3), RunLengthCode(0, 3, 2), RunLengthCode(4, 8, 120)]
    assert res[4].is_EOB()
    assert blk.decode(res).tolist() == a.tolist()
    b = np.array([0, 2] + [0] * 32 + [5] + [0] * 5)
    res = RunLengthBlock(b.shape[0]).encode(b)
    assert res[:4] == [RunLengthCode(1, 3, 2), RunLengthCode(15, 0, 0), RunLengthCode(15, 0, 0), RunLengthCode(2, 4, 5)]
    assert RunLengthBlock(b.shape[0]).decode(res).tolist() == b.tolist()
    z = RunLengthBlock(9).encode(np.zeros(9))
    assert len(z) == 1 and z[0] == RunLengthCode.EOB()


def test_rle_step_known_answers():
    a = np.zeros((3, 1, 9))
    a[0, 0] = [21, 3, 0, 0, 0, 0, 2, 0, 0]
    a[1, 0] = [0, 0, 0, 15, 0, 0, 0, 0, 9]
    assert RunLengthEncoding(config=None).execute(a) == [
        (0, 6, 21), (0, 3, 3), (4, 3, 2), (0, 0), (3, 5, 15), (4, 5, 9), (0, 0), (0, 0)]
    cfg = Configuration(width=3, height=9, block_size=1, dct_size=3)
    rle = RunLengthEncoding(config=cfg)
    assert rle.invert(rle.execute(a)).tolist() == a.tolist()


def test_bytestream_known_bit_strings():
    s = RleBytestream(config=None)
    bits = util.Bits()
    bits.frombytes(s.execute([(4, 3, 2), (0, 0)]))
    assert bits.to01() == "0100" + "0011" + "110" + "0" * 13
    bits = util.Bits()
    bits.frombytes(s.execute([(15, 0, 0), (0, 0)]))
    assert bits.to01() == "1111" + "0000" + "0" * 8
    for x in ([(15, 0, 0), (15, 0, 0), (0, 2, 1), (0, 0)],
              [(1, 2, -1), (0, 3, -2), (8, 3, -3), (8, 5, -15), (0, 0)],
              [(14, 4, 7), (0, 0)],
              [(14, 4, 7), (0, 0), (0, 0), (15, 0, 0), (0, 2, 1), (0, 0)]):
        assert s.invert(s.execute(x)) == x


@pytest.mark.parametrize("bad", [(15, 0, 1), (15, 0, -10), (16, 3, 3), (-1, 3, 3), (10, 16, 0), (4, -1, 0),
                                 (40, -18, 0), (12, 0, 0)])
def test_bytestream_rejects_bad_codes(bad):
    with pytest.raises(BadRleCodeError):
        RleBytestream(config=None).execute([bad, (0, 0)])


def test_bits_stand_in():
    b = util.Bits("101")
    b.append(True)
    b.extend(util.Bits("0000"))
    assert len(b) == 8 and b.to01() == "10110000" and b.tobytes() == bytes([0xB0])
    assert (util.Bits("1") + util.Bits("01")).to01() == "101"
    assert b[0:4].to01() == "1011"
    assert util.BitEncoder().encode_signed(-5).to01() == "0101"
    assert util.BitEncoder().pad_bitstring(util.BitEncoder().encode_unsigned(3)).to01() == "0011"


# ---- container (tests/file_format_tests.py) ---------------------------------------------------
def test_header_and_data_round_trip():
    cfg = Configuration(width=320, height=400, block_size=4, dct_size=8, transform="DFT",
                        quantization=QuantizationMethod("qtable"))
    res = file_format.get_header(file_format.create_header(cfg))
    assert (res.width, res.height, res.block_size, res.dct_size, res.transform, res.quantization.name) == \
        (320, 400, 4, 8, "DFT", "qtable")
    cfg = Configuration(width=320, height=400, block_size=44, dct_size=16, transform="DCT",
                        quantization=QuantizationMethod("divide", divisor=93))
    blob = file_format.generate_data(cfg, CompressedData(y=bytes([4, 8, 15, 16, 23, 42]), cb=bytes([1, 2, 3, 4, 5]),
                                                         cr=bytes([10])))
    read_cfg, data = file_format.read_data(blob)
    assert read_cfg.dct_size == 16 and read_cfg.quantization.params == {"divisor": 93}
    assert (data.y, data.cb, data.cr) == (bytes([4, 8, 15, 16, 23, 42]), bytes([1, 2, 3, 4, 5]), bytes([10]))
    # exact byte layout of the header (file_format.py:67-83)
    hdr = file_format.create_header(cfg)
    assert hdr[:2] == (2 + 13 + len(cfg.quantization.to_json())).to_bytes(2, "little")
    assert hdr[2:10] == b"".join(v.to_bytes(2, "little") for v in (320, 400, 44, 16)) and hdr[10:13] == b"DCT"


# ---- pipeline on configurations that stay on the host (tests/integration_tests.py:50-66) ------
def test_pipeline_with_1pixel_blocks_is_lossless():
    original = np.arange(64).reshape(8, 8)
    cfg = Configuration(width=8, height=8, block_size=1, dct_size=1)
    restored = pipeline.decompress_band(pipeline.compress_band(original, cfg), cfg)
    assert np.allclose(original, restored, rtol=0.000001)


def test_pipeline_range_with_small_dct():
    original = np.array([[220, 255, 123, 205], [255, 255, 112, 10], [15, 51, 83, 221], [239, 73, 62, 22]])
    cfg = Configuration(width=4, height=4, block_size=1, dct_size=2, quantization=QuantizationMethod("divide", divisor=129))
    restored = pipeline.decompress_band(pipeline.compress_band(original, cfg), cfg)
    assert np.all(restored < 256) and np.all(restored > -1)


def test_cli_flag_surface():
    import compress
    args = compress.build_parser().parse_args(["in.png", "out.bin"])
    assert (args.block_size, args.dct_size, args.transform, args.quantization, args.qkeep, args.qdivisor) == \
        (4, 8, "DCT", "qtable", 2, 40)
    assert compress.quantization_from_args(args).name == "qtable"
    args = compress.build_parser().parse_args(["a", "b", "--quantization", "divide", "--qdivisor", "12"])
    assert compress.quantization_from_args(args).quantizer.divisor == 12
    args = compress.build_parser().parse_args(["a", "b", "--quantization", "none"])
    assert compress.quantization_from_args(args) is None


# ---- libjpegx host-side entropy decoder (C++, no GPU needed) ----------------------------------
def test_host_entropy_decoder_inverts_the_python_encoder(golden):
    import jpegx
    c = golden("noise64")
    for suffix in ("qtable", "none", "divide40", "discard2"):
        zz = c["zz_" + suffix]
        blob = RleBytestream(None).execute(RunLengthEncoding(None).execute(zz.astype(float)))
        back = jpegx.entropy_decode(blob, zz.shape[0] * zz.shape[1]).reshape(zz.shape)
        assert np.array_equal(back, zz)
        cfg = Configuration(width=64, height=64, block_size=1)
        ref = RunLengthEncoding(cfg).invert(RleBytestream(cfg).invert(blob))
        assert np.array_equal(ref, zz)
    for bad in (b"", b"\x12", b"\xf0\xf0\xf0\xf0\xf0\x00"):
        with pytest.raises(jpegx.JpegxError):
            jpegx.entropy_decode(bad, 1)


# ---- when the three hot steps may be fused (plugin contract, pipeline/base.py:23-31) -----------
def test_hot_steps_fuse_only_when_adjacent_and_stock():
    from pipeline import _hot_run, _HOT_STEPS, _BUILTIN_STEPS
    from pipeline.basis_change import BasisChange
    from pipeline.quantization import Quantization
    assert _hot_run(list(_BUILTIN_STEPS)) == 4

    class Between:           # stands for a user step registered with 4 < step_index < 5
        step_index = 4.5

    class MyQuantization(Quantization):
        step_index = 5
    steps = list(_BUILTIN_STEPS)
    assert _hot_run(steps[:5] + [Between] + steps[5:]) is None
    assert _hot_run(steps[:5] + [MyQuantization] + steps[6:]) is None          # a subclass is not the built-in step
    assert _hot_run([BasisChange] + list(_HOT_STEPS)) == 1
    assert _hot_run(list(_HOT_STEPS[:2])) is None
    # the registry itself is untouched by these local classes only because they do not derive from AlgorithmStep
    # through the metaclass... MyQuantization does: take it out again
    pipeline.step_classes.remove(MyQuantization)
    assert pipeline._stock_registry()


def test_gpu_mode_accepts_stock_quantiser_objects_only():
    import quantizers
    assert QuantizationMethod("qtable").gpu_mode() == ("qtable", 0.0)
    assert QuantizationMethod("none").gpu_mode() == ("none", 0.0)
    assert QuantizationMethod("divide", divisor=40).gpu_mode() == ("divide", 40.0)
    assert QuantizationMethod("discard", keep=3).gpu_mode() == ("discard", 3.0)
    assert QuantizationMethod("divide", divisor=0).gpu_mode() is None
    m = QuantizationMethod("qtable")
    m.quantizer._qtable = m.quantizer._qtable * 2                 # an edited table is a different quantiser
    assert m.gpu_mode() is None
    m = QuantizationMethod("qtable")
    m.quantizer.table = [[1] * 8] * 8
    assert m.gpu_mode() is None

    class Mine(quantizers.DivisionQuantizer):
        def quantize(self, a):
            return np.floor(a / self.divisor)
    m = QuantizationMethod("divide", divisor=7)
    m.quantizer = Mine(7)
    assert m.gpu_mode() is None
    cfg = Configuration(width=8, height=8, block_size=1, quantization=m)
    assert not pipeline._accelerated(cfg)                          # everything then runs step by step on the host objects
