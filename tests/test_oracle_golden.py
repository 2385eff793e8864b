"""The oracle (oracle/jpegx_oracle.c) against every golden vector produced by the unmodified
reference, and against the reference's own known-answer tests for this path.  CPU only."""
import os

import numpy as np
import pytest

import oracle
from conftest import CASES, MODES


def test_tables_match_reference(tables):
    t = oracle.tables()
    for key in ("dct_matrix", "dct_normalized", "norm_diag", "qtable", "zigzag8"):
        assert np.array_equal(t[key], tables[key]), key
    # the standard JPEG zigzag order
    assert t["zigzag8"][:10].tolist() == [0, 1, 8, 16, 9, 2, 3, 10, 17, 24]


@pytest.mark.parametrize("case", CASES)
def test_forward_stages_bit_identical(golden, case):
    c = golden(case)
    dct = oracle.dct_plane(c["pre"])
    assert np.array_equal(dct, c["dct"])                          # float64, bit for bit
    for suffix, mode, param in MODES:
        q = oracle.quant_plane(dct, mode, param)
        assert np.array_equal(q, c["q_" + suffix])
        assert np.array_equal(oracle.zigzag_plane(q), c["zz_" + suffix])
        fused = oracle.forward_f32(c["pre"].astype(np.float32), mode, param)
        assert np.array_equal(fused, c["zz_" + suffix])


@pytest.mark.parametrize("case", CASES)
def test_inverse_stages_bit_identical(golden, case):
    c = golden(case)
    for suffix, mode, param in MODES:
        r = oracle.restore_plane(oracle.unzigzag_plane(c["zz_" + suffix]), mode, param)
        assert np.array_equal(r, c["restore_" + suffix])
        assert np.array_equal(oracle.idct_plane(r), c["idct_" + suffix])
        assert np.array_equal(oracle.inverse_i16(c["zz_" + suffix], mode, param), c["idct_" + suffix])


def test_tie_stress_blocks_are_really_ties(golden):
    """The tie fixture holds exact DC ties and (4,4) ties; the reference's answers are pinned."""
    c = golden("ties128")
    kinds = np.load(__import__("os").path.join(__import__("conftest").GOLDEN, "ties_kinds.npy"))
    dc = c["dct"][0::8, 0::8]
    assert np.all((dc[(kinds == 1) | (kinds == 3)] % 16) == 8)
    c44 = c["dct"][4::8, 4::8][(kinds == 2) | (kinds == 3)]
    assert np.all(np.abs(np.abs(c44 / 68.0 % 1.0) - 0.5) < 1e-9)


def test_mean_pool_matches_reference(golden):
    c = golden("pooled128")
    assert np.array_equal(oracle.mean_pool(c["input"].astype(np.float64), 2), c["pre"])


def test_reference_known_answers_quantizers():
    """/root/reference/tests/quantization_tests.py:10-54 restated on 8x8-tiled planes."""
    a = np.zeros((8, 8))
    a[0, :3] = [80, 24, 169]
    assert oracle.quant_plane(a, "divide", 40)[0, :3].tolist() == [2, 1, 4]
    assert oracle.restore_plane(oracle.quant_plane(a, "divide", 40), "divide", 40)[0, :3].tolist() == [80, 40, 160]
    b = np.zeros((8, 8))
    b[:2, :2] = [[3.4, 8.0], [0, 0.6]]
    assert oracle.quant_plane(b, "none")[:2, :2].tolist() == [[3, 8], [0, 1]]
    d = np.arange(64, dtype=float).reshape(8, 8)
    q = oracle.quant_plane(d, "discard", 2)
    assert q[:2, :2].tolist() == [[0, 1], [8, 9]] and q[2:].sum() == 0 and q[:, 2:].sum() == 0


def test_reference_dct_round_trip_property():
    """/root/reference/tests/basis_change_tests.py:32-38: inverse(forward(arange(64))) == input."""
    a = np.arange(64, dtype=float).reshape(8, 8)
    back = oracle.idct_plane(oracle.dct_plane(a), rounded=False)
    assert np.allclose(a, back, rtol=0.01)
    assert np.abs(back - a).max() < 1e-10


def test_python_loop_restatement_agrees(golden):
    from oracle import ref_loop
    c = golden("smooth64")
    assert np.array_equal(ref_loop.forward_qtable(c["pre"]), c["zz_qtable"])


def test_bad_shapes_are_rejected():
    with pytest.raises(ValueError):
        oracle.dct_plane(np.zeros((12, 8)))


# ---- entropy stage (steps 7-8): pinned by the reference's own known answers -------------------
def test_rle_tuples_known_answers():
    """/root/reference/tests/RLE_tests.py:16-64."""
    assert oracle.rle_block_tuples([-15, 0, 0, 0, 3, 2, 0, 0, 0, 0, 120, 0, 0, 0, 0]) == \
        [(0, 5, -15), (3, 3, 3), (0, 3, 2), (4, 8, 120), (0, 0)]
    assert oracle.rle_block_tuples([0, 2] + [0] * 32 + [5] + [0] * 5) == \
        [(1, 3, 2), (15, 0, 0), (15, 0, 0), (2, 4, 5), (0, 0)]
    assert oracle.rle_block_tuples([0] * 9) == [(0, 0)]
    # step level, tests/RLE_tests.py:66-86
    blocks = [[21, 3, 0, 0, 0, 0, 2, 0, 0], [0, 0, 0, 15, 0, 0, 0, 0, 9], [0] * 9]
    got = [t for b in blocks for t in oracle.rle_block_tuples(b)]
    assert got == [(0, 6, 21), (0, 3, 3), (4, 3, 2), (0, 0), (3, 5, 15), (4, 5, 9), (0, 0), (0, 0)]


def test_rle_bytestream_known_bit_strings():
    """/root/reference/tests/RLE_tests.py:98-122: '0100 0011 110' + padding, '1111 0000' + EOB."""
    def bits(b):
        return "".join(format(x, "08b") for x in b)
    assert bits(oracle.rle_bytestream(np.array([[0, 0, 0, 0, 2, 0, 0, 0, 0]]))) == "0100" + "0011" + "110" + "0" * 13
    # a value after 15 zeros: (15,0,0) then (0, size, amp): '11110000' + '0000' '0010' '11' + pad + EOB
    z = np.zeros((1, 20), np.int16)
    z[0, 15] = 1
    assert bits(oracle.rle_bytestream(z)).startswith("11110000" + "0000" + "0010" + "11")
    assert bits(oracle.rle_bytestream(np.zeros((2, 9), np.int16))) == "0" * 16
    # negative amplitudes: sign bit 0 (tests/RLE_tests.py:132-139 round-trip codes)
    z = np.zeros((1, 4), np.int16)
    z[0, 1] = -1
    assert bits(oracle.rle_bytestream(z)).startswith("0001" + "0010" + "01")


def test_rle_bytestream_matches_the_host_mirror(golden):
    """The pure-Python mirror of steps 7-8 (product host code) and the oracle agree on real streams."""
    from pipeline.rle_byte_stream import RleBytestream
    from pipeline.run_length_encoding import RunLengthEncoding
    c = golden("smooth64")
    for suffix, _, _ in MODES:
        zz = c["zz_" + suffix]
        want = RleBytestream(None).execute(RunLengthEncoding(None).execute(zz.astype(float)))
        blob, sizes = oracle.rle_bytestream(zz, want_block_bytes=True)
        assert blob == want and int(sizes.sum()) == len(blob)


# ---- step 7 and the container pinned by outputs of the reference itself (make_golden.py) -------
def reference_tuples(rows):
    """(K, 3) int32 fixture rows -> the reference's tuple list ((0, 0, 0) rows are the (0, 0) end markers)."""
    return [(0, 0) if (r == 0 and s == 0) else (r, s, a) for r, s, a in np.asarray(rows).tolist()]


def block_bytes_from_tuples(tuples):
    """Bytes of every block's code string implied by the reference's step-7 tuples and its bit layout
    (rle_byte_stream.py:48-59, util.py:203-221): 8 header bits per code, plus `size` amplitude bits unless the
    code is a zero chain or the end marker; each block is padded to a byte."""
    sizes, bits = [], 0
    for t in tuples:
        bits += 8
        if len(t) == 3 and not (t[0] == 15 and t[1] == 0):
            bits += t[1]
        if len(t) == 2:
            sizes.append((bits + 7) // 8)
            bits = 0
    return np.array(sizes, dtype=np.uint32)


def test_rle_block_fixture_from_the_reference():
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "rle_blocks.npz"))
    for i in range(int(fx["n"])):
        want = reference_tuples(fx["codes_%d" % i])
        assert oracle.rle_block_tuples(fx["block_%d" % i]) == want, i


@pytest.mark.parametrize("case", CASES)
def test_rle_tuples_of_every_golden_stream(golden, case):
    """oracle.rle_block_tuples block by block == RunLengthEncoding.execute of the reference on the same
    zigzag stream; the byte stream's per-block sizes and total follow from those tuples."""
    c = golden(case)
    for suffix, _, _ in MODES:
        zz = c["zz_" + suffix]
        want = reference_tuples(c["rle_" + suffix])
        got = [t for blk in zz.reshape(-1, 64) for t in oracle.rle_block_tuples(blk)]
        assert got == want, (case, suffix)
        blob, sizes = oracle.rle_bytestream(zz, want_block_bytes=True)
        expect = block_bytes_from_tuples(want)
        assert np.array_equal(sizes, expect) and len(blob) == int(expect.sum())


def test_host_rle_step_reproduces_the_reference_tuples(golden):
    from pipeline import Configuration
    from pipeline.rle_byte_stream import RleBytestream
    from pipeline.run_length_encoding import RunLengthEncoding
    for case in CASES:
        c = golden(case)
        h, w = c["pre"].shape
        cfg = Configuration(width=w, height=h, block_size=1, dct_size=8)
        for suffix, _, _ in MODES:
            zz = c["zz_" + suffix]
            want = reference_tuples(c["rle_" + suffix])
            step = RunLengthEncoding(cfg)
            assert step.execute(zz.astype(np.float64)) == want
            assert np.array_equal(step.invert(want), zz)
            blob = RleBytestream(cfg).execute(want)
            assert blob == oracle.rle_bytestream(zz) and RleBytestream(cfg).invert(blob) == want


def test_container_bytes_match_the_reference():
    """file_format.create_header / generate_data / read_data against the header and file bytes the reference
    wrote for seven configurations (tests/golden/container.npz)."""
    import json
    import file_format
    from pipeline import CompressedData, Configuration, QuantizationMethod
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "container.npz"))
    for i in range(int(fx["n"])):
        d = json.loads(fx["config_%d" % i].tobytes().decode())
        q = QuantizationMethod(d["q"][0], **d["q"][1]) if d["q"] is not None else None
        cfg = Configuration(width=d["width"], height=d["height"], block_size=d["block_size"], dct_size=d["dct_size"],
                            transform=d["transform"], quantization=q)
        assert file_format.create_header(cfg) == fx["header_%d" % i].tobytes(), d
        parts = [fx["%s_%d" % (n, i)].tobytes() for n in ("y", "cb", "cr")]
        assert file_format.generate_data(cfg, CompressedData(*parts)) == fx["file_%d" % i].tobytes(), d
        back, data = file_format.read_data(fx["file_%d" % i].tobytes())
        assert (back.width, back.height, back.block_size, back.dct_size, back.transform) == \
            (d["width"], d["height"], d["block_size"], d["dct_size"], d["transform"])
        assert back.quantization.name == (d["q"][0] if d["q"] else "none")
        assert [data.y, data.cb, data.cr] == parts


def test_block_size_3_case_from_the_reference(golden):
    """SubSampling with block_size 3 gives k/9 -- not fp32 numbers; the oracle's float64 chain reproduces the
    reference's arrays for it (tests/golden/case_pooled3x72.npz)."""
    c = golden("pooled3x72")
    pre = oracle.mean_pool(c["input"].astype(np.float64), 3)
    assert np.array_equal(pre, c["pre"]) and not np.array_equal(pre.astype(np.float32).astype(np.float64), pre)
    dct = oracle.dct_plane(pre)
    assert np.array_equal(dct, c["dct"])
    for suffix, mode, param in MODES:
        assert np.array_equal(oracle.zigzag_plane(oracle.quant_plane(dct, mode, param)), c["zz_" + suffix].astype(np.float64))


# ---- the way back (steps 8 and 7 inverted): the oracle's decoder pinned ---------------------------------------------
def _from_bits(s):
    s = s.replace(" ", "")
    assert len(s) % 8 == 0
    return bytes(int(s[i:i + 8], 2) for i in range(0, len(s), 8))


def test_rle_stream_decoder_known_answers():
    """The bit strings the reference's tests expect from RleBytestream.execute (tests/RLE_tests.py:98-122), read
    backwards, and the round trips that file asserts for RleBytestream.invert (:124-139, :182-196) -- with the encoding
    direction pinned above, invert(execute(x)) == x pins the decoder on those tuple lists."""
    assert oracle.rle_stream_tuples(_from_bits("0100 0011 110" + "0" * 13)) == [(4, 3, 2), (0, 0)]
    assert oracle.rle_stream_tuples(_from_bits("1111 0000" + "0" * 8)) == [(15, 0, 0), (0, 0)]
    trips = [
        ([(15, 0, 0), (15, 0, 0), (0, 2, 1), (0, 0)], 31),                                   # :124-130
        ([(1, 2, -1), (0, 3, -2), (8, 3, -3), (8, 5, -15), (0, 0)], 21),                    # :132-138
        ([(14, 4, 7), (0, 0)], 15),                                                         # :182-188
        ([(14, 4, 7), (0, 0), (0, 0), (15, 0, 0), (0, 2, 1), (0, 0)], 16),                  # :190-196
    ]
    for x, n in trips:
        nblocks = sum(1 for t in x if len(t) == 2)
        z = oracle.rle_tuples_decode(x, nblocks, n)                    # RunLengthEncoding.invert: the values
        assert [t for blk in z for t in oracle.rle_block_tuples(blk)] == x      # ... whose tuples are x again
        blob = oracle.rle_bytestream(z.reshape(nblocks, 1, n))
        assert oracle.rle_stream_tuples(blob) == x
        assert np.array_equal(oracle.rle_decode(blob, nblocks, n), z)


@pytest.mark.parametrize("case", CASES)
def test_rle_decoder_on_every_golden_stream(golden, case):
    """RunLengthEncoding.invert of the reference's own step-7 tuples is the golden zigzag array (the reference ran that
    direction when the fixtures were made: `izz_*` continues from it); the oracle's decoder gives the same from the
    tuples and from the bytes."""
    c = golden(case)
    for suffix, _, _ in MODES:
        zz = c["zz_" + suffix].reshape(-1, 64)
        want = reference_tuples(c["rle_" + suffix])
        assert np.array_equal(oracle.rle_tuples_decode(want, len(zz)), zz), (case, suffix)
        blob = oracle.rle_bytestream(zz.reshape(len(zz), 1, 64))
        assert oracle.rle_stream_tuples(blob) == want
        assert np.array_equal(oracle.rle_decode(blob, len(zz)), zz)


def test_rle_decoder_fails_where_the_reference_does():
    """Error behaviour read off the reference's code (it cannot be run: bitarray): int('', 2) for a code of size 1 and
    for the headers (1..14, 0); reshape for a wrong number of values.  And what it lets pass: codes behind the last end
    marker (an unfinished block is never yielded, run_length_encoding.py:91-97), damaged padding bits (stepped over
    unread, rle_byte_stream.py:26-28), a block longer than 64 values as long as the TOTAL fits (run_length_encoding.py:
    37-38: `[0] * negative` is empty)."""
    z = np.zeros((3, 64), np.int16)
    z[0, 3], z[1, 0], z[2, 63] = 5, -7, 1
    good = oracle.rle_bytestream(z.reshape(3, 1, 64))
    assert np.array_equal(oracle.rle_decode(good, 3), z)
    with pytest.raises(oracle.RleStreamError):
        oracle.rle_decode(good, 2)                                        # a block too many for the plane
    with pytest.raises(oracle.RleStreamError):
        oracle.rle_decode(good, 4)
    with pytest.raises(oracle.RleStreamError):
        oracle.rle_decode(good + b"\x00", 3)                              # one more (empty) block behind the plane
    with pytest.raises(oracle.RleStreamError):
        oracle.rle_decode(_from_bits("0000 0001 1" + "0" * 7) + good, 4)  # size 1: sign bit only
    with pytest.raises(oracle.RleStreamError):
        oracle.rle_decode(_from_bits("0011 0000") + good, 3)              # (3, 0)
    # the last byte cut off: two bits of the last end marker are left, and a read at the end of the stream returns the
    # bits there are (run '0000', size '00'): still an end marker.  One byte more and the amplitude has no bits: int('', 2)
    assert np.array_equal(oracle.rle_decode(good[:-1], 3), z)
    with pytest.raises(oracle.RleStreamError):
        oracle.rle_decode(good[:-2], 3)
    assert np.array_equal(oracle.rle_decode(good + b"\xf0", 3), z)        # a zero chain behind the last end marker: dropped
    first = oracle.rle_bytestream(z[:1].reshape(1, 1, 64))
    bits = "".join(format(b, "08b") for b in first)
    assert bits.endswith("0" * 9)                                         # the block ends '...' + end marker + padding
    tampered = _from_bits(bits[:-1] + "1") + good[len(first):]
    assert np.array_equal(oracle.rle_decode(tampered, 3), z)              # a padding bit set: nobody reads it
    # one block of 128 values (two zero chains short of it, then a value) passes for two blocks of 64
    long_block = [(15, 0, 0)] * 8 + [(7, 2, 1), (0, 0)]
    got = oracle.rle_tuples_decode(long_block, 2)
    assert got.shape == (2, 64) and got[1, 63] == 1 and np.count_nonzero(got) == 1


def test_host_mirror_decoder_against_the_oracle_on_damaged_streams():
    """The product's pure-Python steps 8/7 backwards (pipeline/rle_byte_stream.py, run_length_encoding.py of the package)
    against the oracle's restatement of the reference on random bytes and damaged well-formed streams: whatever the
    package accepts the reference accepts too, with the same values; the package is stricter only on streams no encoder
    writes (codes behind the last end marker, blocks of more than 64 values)."""
    from pipeline import Configuration
    from pipeline.rle_byte_stream import RleBytestream
    from pipeline.run_length_encoding import RunLengthEncoding
    rng = np.random.default_rng(21)
    seen = {"equal": 0, "both refuse": 0, "package stricter": 0}
    for trial in range(300):
        nb = int(rng.integers(1, 40))
        cfg = Configuration(width=8 * nb, height=8, block_size=1, dct_size=8)
        kind = trial % 4
        zz = (rng.integers(-300, 300, (nb, 64)) * (rng.random((nb, 64)) < rng.random())).astype(np.int16)
        dmg = bytearray(oracle.rle_bytestream(zz.reshape(nb, 1, 64)))
        if kind == 0:
            dmg = bytearray(rng.integers(0, 256, int(rng.integers(1, 600)), dtype=np.uint8).tobytes())
        elif kind == 1:
            for _ in range(int(rng.integers(0, 3))):
                dmg[int(rng.integers(0, len(dmg)))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 2:
            dmg = dmg[:int(rng.integers(1, len(dmg) + 1))]
        else:
            i = int(rng.integers(0, len(dmg) + 1))
            dmg[i:i] = rng.integers(0, 256, int(rng.integers(1, 5)), dtype=np.uint8).tobytes()
        blob = bytes(dmg)
        try:
            want = oracle.rle_decode(blob, nb)
        except oracle.RleStreamError:
            want = None
        try:
            got = np.asarray(RunLengthEncoding(cfg).invert(RleBytestream(cfg).invert(blob))).reshape(nb, 64)
        except Exception:
            got = None
        assert not (got is not None and want is None), trial
        if got is not None:
            assert np.array_equal(got, want), trial
            seen["equal"] += 1
        else:
            seen["both refuse" if want is None else "package stricter"] += 1
    assert seen["equal"] >= 30 and seen["both refuse"] >= 100 and seen["package stricter"] <= 15, seen


def test_native_host_parser_against_the_oracle_on_damaged_streams():
    """jpegx_host_entropy_decode (C++, no device involved: what decompress_band falls back to and what the device
    decoder is fuzzed against in test_gpu_entropy.py) with the oracle's restatement of the reference as the judge: what
    it accepts the reference accepts, with the same values.  It refuses more than the reference does, on streams no
    encoder writes: it wants every header and amplitude whole (the reference reads short at the end of the stream),
    nothing behind the plane's last block (the reference drops codes that no end marker follows) and at most 64
    values per block (the reference only checks the total)."""
    import jpegx
    z = np.zeros((3, 64), np.int16)
    z[0, 3], z[1, 0], z[2, 63] = 5, -7, 1
    good = oracle.rle_bytestream(z.reshape(3, 1, 64))
    assert np.array_equal(jpegx.entropy_decode(good, 3), z)
    first = oracle.rle_bytestream(z[:1].reshape(1, 1, 64))
    bits = "".join(format(b, "08b") for b in first)
    tampered = _from_bits(bits[:-1] + "1") + good[len(first):]
    assert np.array_equal(jpegx.entropy_decode(tampered, 3), z)          # padding bits are stepped over unread, as in the reference
    long_block = oracle.rle_bytestream(oracle.rle_tuples_decode([(15, 0, 0)] * 8 + [(7, 2, 1), (0, 0)], 1, 128).reshape(1, 1, 128))
    for lenient, n in ((good + b"\xf0", 3), (good[:-1], 3), (long_block, 2)):
        oracle.rle_decode(lenient, n)                                    # the reference lets these pass ...
        with pytest.raises(jpegx.JpegxError):
            jpegx.entropy_decode(lenient, n)                             # ... the native parser does not
    rng = np.random.default_rng(5)
    seen = {"equal": 0, "both refuse": 0, "parser stricter": 0}
    for trial in range(400):
        nb = int(rng.integers(1, 120))
        kind = trial % 4
        zz = (rng.integers(-300, 300, (nb, 64)) * (rng.random((nb, 64)) < rng.random())).astype(np.int16)
        dmg = bytearray(oracle.rle_bytestream(zz.reshape(nb, 1, 64)))
        if kind == 0:
            dmg = bytearray(rng.integers(0, 256, int(rng.integers(1, 2048)), dtype=np.uint8).tobytes())
        elif kind == 1:
            for _ in range(int(rng.integers(0, 3))):
                dmg[int(rng.integers(0, len(dmg)))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 2:
            dmg = dmg[:max(1, len(dmg) - int(rng.integers(0, 4)))]
        else:
            i = int(rng.integers(0, len(dmg) + 1))
            dmg[i:i] = rng.integers(0, 256, int(rng.integers(1, 5)), dtype=np.uint8).tobytes()
        blob = bytes(dmg)
        try:
            want = oracle.rle_decode(blob, nb)
        except oracle.RleStreamError:
            want = None
        try:
            got = jpegx.entropy_decode(blob, nb)
        except jpegx.JpegxError:
            got = None
        assert not (got is not None and want is None), trial
        if got is not None:
            assert np.array_equal(got, want), trial
            seen["equal"] += 1
        else:
            seen["both refuse" if want is None else "parser stricter"] += 1
    assert seen["equal"] >= 40 and seen["both refuse"] >= 100 and seen["parser stricter"] >= 1, seen
