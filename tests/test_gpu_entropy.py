"""GPU: the entropy stage (steps 7+8) on the device against the oracle's restatement, which is
pinned by the reference's own known answers (tests/test_oracle_golden.py)."""
import numpy as np
import pytest

import oracle
from conftest import CASES, MODES

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("suffix,mode,param", MODES)
def test_entropy_encode_matches_oracle_on_reference_streams(gpu, golden, case, suffix, mode, param):
    zz = golden(case)["zz_" + suffix]
    assert gpu.entropy_encode(zz) == oracle.rle_bytestream(zz)


def test_entropy_encode_known_answers(gpu):
    """/root/reference/tests/RLE_tests.py:98-122 restated on 64-value blocks."""
    def bits(b):
        return "".join(format(x, "08b") for x in b)
    z = np.zeros((1, 64), np.int16)
    z[0, 4] = 2
    assert bits(gpu.entropy_encode(z)) == "0100" + "0011" + "110" + "0" * 13
    z = np.zeros((1, 64), np.int16)
    z[0, 15] = 1
    z[0, 63] = -3                                    # run 47 = 3 chains + 2
    want = "11110000" + "0000" "0010" "11" + "11110000" * 3 + "0010" "0011" "011" + "0" * 8
    got = bits(gpu.entropy_encode(z))
    assert got.startswith(want) and len(got) % 8 == 0 and set(got[len(want):]) <= {"0"}
    assert gpu.entropy_encode(np.zeros((3, 64), np.int16)) == b"\x00\x00\x00"


@pytest.mark.parametrize("kind", ["noise", "smooth"])
def test_entropy_encode_large_and_ragged(gpu, kind):
    a = gpu.synth.generate_plane(kind, 1024, 1024, seed=13)
    for mode, param in (("qtable", 0.0), ("none", 0.0), ("divide", 7.0)):
        zz = oracle.forward_f32(a, mode, param)
        assert gpu.entropy_encode(zz) == oracle.rle_bytestream(zz)
    rng = np.random.default_rng(2)
    for nblocks in (1, 63, 64, 65, 1000):
        z = (rng.integers(-40, 41, (nblocks, 64)) * (rng.random((nblocks, 64)) < 0.2)).astype(np.int16)
        z[0, :] = rng.integers(-16383, 16384, 64)     # dense block with maximal amplitudes
        assert gpu.entropy_encode(z) == oracle.rle_bytestream(z)


def test_entropy_rejects_amplitudes_beyond_15_bits(gpu):
    z = np.zeros((2, 64), np.int16)
    z[1, 3] = 16384
    with pytest.raises(gpu.JpegxError, match="15 bits"):
        gpu.entropy_encode(z)


def test_compress_plane_all_on_device(gpu):
    """Steps 1-8 with only the final bytes leaving the GPU == oracle forward + oracle entropy stage."""
    a = gpu.synth.generate_plane("smooth", 256, 512, seed=6)
    assert gpu.compress_plane(a, 1, "qtable") == oracle.rle_bytestream(oracle.forward_f32(a, "qtable"))
    pooled = oracle.mean_pool(a, 2).astype(np.float32)
    assert gpu.compress_plane(a, 2, "qtable") == oracle.rle_bytestream(oracle.forward_f32(pooled, "qtable"))


@pytest.mark.parametrize("kind,bs,mode,param", [("noise", 1, "qtable", 0.0), ("smooth", 1, "qtable", 0.0), ("smooth", 2, "qtable", 0.0),
                                               ("noise", 4, "divide", 3.0), ("smooth", 1, "none", 0.0), ("smooth", 1, "discard", 4.0)])
def test_forward_kernel_sizes_its_own_blocks(gpu, kind, bs, mode, param):
    """compress_band's forward kernel counts the entropy stage's bits per block from its registers (two coefficients per
    instruction, chain codes decided by the wave): same coefficients as the plain kernel, same sizes as the sizes pass
    over the stream -- which test_device_block_sizes_follow_from_the_reference_tuples pins to the reference's step-7
    tuples -- and the scan behind it writes total and error flag without the workspace having been cleared."""
    a = gpu.synth.generate_plane(kind, 256 * bs, 512 * bs, seed=11).astype(np.uint8)
    zz, sizes, total, rc = gpu.forward_u8_block_sizes(a, bs, mode, param)
    pooled = oracle.mean_pool(a.astype(np.float32), bs).astype(np.float32) if bs > 1 else a.astype(np.float32)
    want = oracle.forward_f32(pooled, mode, param)
    assert np.array_equal(zz, want)
    ref_sizes = gpu.entropy_block_sizes(want)
    assert np.array_equal(sizes, ref_sizes)
    assert rc == 0 and total == int(ref_sizes.sum()) == len(oracle.rle_bytestream(want))
    if kind == "smooth" and mode in ("qtable", "discard"):
        assert (np.abs(want.reshape(-1, 64)) > 0).sum(axis=1).min() < 20         # long zero runs: chain codes were counted


def test_band_of_more_than_one_scan_chunk(gpu):
    """4096 x 4112: 263 168 blocks = 4112 waves, one more than a scan chunk holds -- the two-level scan behind the forward
    kernel's sizes (flag bit masked out of the waves' totals) instead of the single-launch form."""
    a = gpu.synth.generate_plane("noise", 4096, 4112, seed=3).astype(np.uint8)
    assert gpu.compress_plane(a, 1, "qtable") == oracle.rle_bytestream(oracle.forward_f32(a.astype(np.float32), "qtable"))


@pytest.mark.parametrize("case", CASES)
def test_device_block_sizes_follow_from_the_reference_tuples(gpu, golden, case):
    """k_rle_sizes against the reference's own step-7 output (tests/golden rle_*): the bytes of every block
    follow from its tuples and the bit layout; the device stream has exactly that total length."""
    from test_oracle_golden import block_bytes_from_tuples, reference_tuples
    c = golden(case)
    for suffix, _, _ in MODES:
        zz = c["zz_" + suffix]
        expect = block_bytes_from_tuples(reference_tuples(c["rle_" + suffix]))
        assert np.array_equal(gpu.entropy_block_sizes(zz), expect), (case, suffix)
        assert len(gpu.entropy_encode(zz)) == int(expect.sum())


def test_emit_after_a_flagged_sizes_pass_writes_nothing(gpu):
    """A C-ABI caller that ignores jpegx_entropy_total's error must not get an out-of-bounds emit: blocks with
    16-bit amplitudes can exceed the emit kernel's staging area, so the kernel refuses to run."""
    import ctypes
    L = gpu.lib()
    z = np.full((130, 64), 32767, np.int16)              # every coefficient needs 16 bits
    dzz, dws = gpu.DeviceBuffer(z.nbytes), gpu.DeviceBuffer(L.jpegx_entropy_workspace_bytes(130))
    dout = gpu.DeviceBuffer(130 * 256)
    dzz.upload(z)
    gpu.check(L.jpegx_memset(dout.ptr, 0xAB, 130 * 256, None))
    gpu.check(L.jpegx_entropy_sizes(dzz.ptr, 130, dws.ptr, None))
    total = ctypes.c_ulonglong(0)
    assert L.jpegx_entropy_total(dws.ptr, ctypes.byref(total), None) == -1
    assert b"15 bits" in L.jpegx_last_error()
    gpu.check(L.jpegx_entropy_emit(dzz.ptr, 130, dws.ptr, dout.ptr, None))      # enqueues, writes nothing
    gpu.check(L.jpegx_device_synchronize())
    assert np.all(dout.download((130 * 256,), np.uint8) == 0xAB)


@pytest.mark.parametrize("case", CASES)
def test_device_entropy_decoder_matches_the_host_parser(gpu, golden, case):
    """The parallel decoder (block starts recovered from the zero byte that ends every block, way marks, four lanes per
    block) returns exactly what the oracle's decoder (the reference's steps 8 and 7 backwards, restated; pinned in
    test_oracle_golden.py) and the sequential host parser return, on every golden stream."""
    c = golden(case)
    for suffix, _, _ in MODES:
        zz = c["zz_" + suffix]
        blob = oracle.rle_bytestream(zz)
        n = zz.shape[0] * zz.shape[1]
        want = oracle.rle_decode(blob, n)
        assert np.array_equal(want.reshape(zz.shape), zz)
        assert np.array_equal(gpu.entropy_decode_gpu(blob, n), want), (case, suffix)
        assert np.array_equal(gpu.entropy_decode(blob, n), want)


@pytest.mark.parametrize("kind", ["noise", "smooth"])
def test_device_entropy_decoder_large_and_adversarial(gpu, kind):
    a = gpu.synth.generate_plane(kind, 2048, 2048, seed=17)
    for mode, param in (("qtable", 0.0), ("none", 0.0), ("divide", 7.0)):
        zz = oracle.forward_f32(a, mode, param)
        blob = oracle.rle_bytestream(zz)
        assert np.array_equal(gpu.entropy_decode_gpu(blob, zz.shape[0] * zz.shape[1]).reshape(zz.shape), zz), mode
    rng = np.random.default_rng(5)
    for nblocks in (1, 2, 63, 64, 65, 1000, 4097):
        z = (rng.integers(-40, 41, (nblocks, 64)) * (rng.random((nblocks, 64)) < 0.2)).astype(np.int16)
        z[0, :] = rng.integers(-16383, 16384, 64)               # dense block, maximal amplitudes
        z[nblocks // 2, :] = 0                                   # an all-zero block: the single byte 0x00
        # amplitudes whose bits contain whole zero bytes: false block-start candidates inside a block
        z[-1, ::2] = rng.choice(np.array([-16384 + 256, 8192, 4096, 256, -512], dtype=np.int16), 32)
        blob = oracle.rle_bytestream(z)
        assert blob.count(b"\x00") > nblocks or nblocks < 3
        assert np.array_equal(gpu.entropy_decode_gpu(blob, nblocks), z), nblocks
    assert np.array_equal(gpu.entropy_decode_gpu(b"\x00" * 300, 300), np.zeros((300, 64), np.int16))


def test_flat_regions_in_a_busy_stream_take_the_second_try_not_the_whole_stream_scheme(gpu):
    """Segments are sized from the stream's AVERAGE block length; a stretch of three-byte blocks (a flat region: only the
    DC coefficient) inside a stream of long blocks holds several hundred block starts per segment -- more than its tables
    do.  The decoder then goes again with 256-byte segments before it falls back to the whole-stream scheme (and its
    host round trip); single-byte blocks (all coefficients zero) overflow those too."""
    rng = np.random.default_rng(21)
    busy = rng.integers(-300, 300, (3000, 64)).astype(np.int16)
    busy[:, 0] = rng.integers(1, 1000, 3000)                    # like every block of non-negative samples that holds anything: DC > 0
    flat = np.zeros((2500, 64), np.int16)
    flat[:, 0] = rng.integers(100, 1000, 2500)
    z = np.concatenate([busy[:1500], flat, busy[1500:]])
    blob = oracle.rle_bytestream(z)
    assert np.array_equal(gpu.entropy_decode_gpu(blob, z.shape[0]), z)
    assert gpu.last_decode_level() == 1
    assert np.array_equal(gpu.entropy_decode_gpu(oracle.rle_bytestream(busy), busy.shape[0]), busy)
    assert gpu.last_decode_level() == 0
    black = np.concatenate([busy[:1500], np.zeros((4000, 64), np.int16), busy[1500:]])
    assert np.array_equal(gpu.entropy_decode_gpu(oracle.rle_bytestream(black), black.shape[0]), black)
    assert gpu.last_decode_level() == 2


def test_first_try_drops_candidates_that_cannot_start_a_block_and_the_second_try_catches_what_that_misses(gpu):
    """The decoder's first try keeps, behind the first 192 bytes of a segment, only candidates whose first byte is below
    0x10: a block of non-negative samples that holds anything has a non-zero DC, so its first header's run nibble is
    zero (an empty block is the byte 0x00) -- most false candidates (a zero byte inside a block's bits) fail that, which
    halves the candidates of a busy stream.  A stream with blocks that start otherwise (DC 0 beside non-zero AC: the
    reference writes them for very dark content) lands on a dropped position, takes the second try (all candidates) and
    decodes to the same coefficients; the working set then leaves the filter off for its next calls (such content pays
    once in a while, not every time)."""
    rng = np.random.default_rng(77)
    usual = rng.integers(-200, 200, (6000, 64)).astype(np.int16)
    usual[:, 0] = rng.integers(1, 1020, 6000)
    dark = usual.copy()
    dark[::50, 0] = 0                                            # DC 0, AC non-zero: first header (run >= 1, size)
    gpu.lib().jpegx_host_pool_release()                          # fresh working sets: the pause counters start at zero
    blob_u, blob_d = oracle.rle_bytestream(usual), oracle.rle_bytestream(dark)
    assert np.array_equal(gpu.entropy_decode_gpu(blob_u, len(usual)), usual) and gpu.last_decode_level() == 0
    assert np.array_equal(gpu.entropy_decode_gpu(blob_d, len(dark)), dark) and gpu.last_decode_level() == 1
    for _ in range(3):                                           # the filter pauses: first try again, with every candidate
        assert np.array_equal(gpu.entropy_decode_gpu(blob_d, len(dark)), dark) and gpu.last_decode_level() == 0
    assert np.array_equal(gpu.entropy_decode_gpu(blob_u, len(usual)), usual) and gpu.last_decode_level() == 0
    # damaged streams are still refused (one try later than without the filter), never accepted
    bad = bytearray(blob_u)
    bad[len(bad) // 2] ^= 0x40
    try:
        got = gpu.entropy_decode_gpu(bytes(bad), len(usual))
    except gpu.JpegxError:
        got = None
    try:
        want = oracle.rle_decode(bytes(bad), len(usual))
    except oracle.RleStreamError:
        want = None
    assert got is None or (want is not None and np.array_equal(got, want))


def test_device_decoder_on_caller_buffers(gpu, golden):
    """jpegx_entropy_decode / _status: the decoder on device pointers the caller owns (stream in, int16 zigzag blocks out,
    a workspace sized by jpegx_entropy_decode_workspace_bytes) -- what an integrator uses who keeps the streams on the
    device, e.g. behind the RCCL gather.  Same coefficients as the oracle's decoder; the second try takes what the first
    hands on; malformed streams are refused."""
    rng = np.random.default_rng(8)
    streams = [golden("noise64")["zz_qtable"].reshape(-1, 64), golden("smooth64")["zz_none"].reshape(-1, 64)]
    z = rng.integers(-150, 150, (20000, 64)).astype(np.int16)
    z[:, 0] = rng.integers(1, 900, len(z))
    streams.append(z)
    dark = z[:3000].copy()
    dark[::40, 0] = 0                                           # blocks the first try's filter drops: level 1 takes the stream
    streams.append(dark)
    L = gpu.lib()
    for i, zz in enumerate(streams):
        blob = oracle.rle_bytestream(zz.reshape(len(zz), 1, 64))
        want = oracle.rle_decode(blob, len(zz))
        d_bytes = gpu.DeviceBuffer(len(blob) + 16)
        d_bytes.upload(np.frombuffer(blob + bytes(16), dtype=np.uint8))
        ws = gpu.DeviceBuffer(int(L.jpegx_entropy_decode_workspace_bytes(len(blob), len(zz))))
        d_zz = gpu.DeviceBuffer(len(zz) * 128)
        for rep in range(2):                                    # the same workspace again: every call clears its state itself
            level = gpu.entropy_decode_device(d_bytes.ptr, len(blob), len(zz), ws.ptr, d_zz.ptr)
            assert level == (1 if i == 3 else 0), (i, level)
            assert np.array_equal(d_zz.download((len(zz), 64), np.int16), want), i
        bad = bytearray(blob + bytes(16))
        bad[len(blob) // 2] ^= 0x10
        d_bytes.upload(np.frombuffer(bytes(bad), dtype=np.uint8))
        try:
            gpu.entropy_decode_device(d_bytes.ptr, len(blob), len(zz), ws.ptr, d_zz.ptr)
            got = d_zz.download((len(zz), 64), np.int16)
        except gpu.JpegxError:
            got = None
        try:
            ref = oracle.rle_decode(bytes(bad[:len(blob)]), len(zz))
        except oracle.RleStreamError:
            ref = None
        assert got is None or (ref is not None and np.array_equal(got, ref)), i
    with pytest.raises(gpu.JpegxError):
        gpu.check(L.jpegx_entropy_decode(d_bytes.ptr, 10, 1, ws.ptr, d_zz.ptr, 2, None), "level 2")


def test_decoder_state_stays_clean_across_streams_of_different_lengths(gpu):
    """The segmented decoder keeps its status blocks and exit words zero from call to call by itself (no memset launch per
    call) and its other arrays move with the segment count: a short stream's arrays must not end up where a later,
    longer stream keeps its exit words (they did once: the state has an allocation of its own since).  Streams of
    1 ... 4097 blocks, twice round, refused streams in between, everything on the same pooled working set."""
    rng = np.random.default_rng(12)
    streams = []
    for nblocks in (1, 2, 63, 64, 65, 1000, 4097, 3, 5, 200):
        z = (rng.integers(-40, 41, (nblocks, 64)) * (rng.random((nblocks, 64)) < 0.2)).astype(np.int16)
        z[0, :] = rng.integers(-16383, 16384, 64)
        streams.append((z, oracle.rle_bytestream(z)))
    for rep in range(2):
        for z, blob in streams:
            assert np.array_equal(gpu.entropy_decode_gpu(blob, z.shape[0]), z), (rep, z.shape[0])
            if z.shape[0] in (65, 200):                    # a refused stream leaves the state clean as well
                with pytest.raises(gpu.JpegxError):
                    gpu.entropy_decode_gpu(blob[:-1], z.shape[0])
                with pytest.raises(gpu.JpegxError):
                    gpu.entropy_decode_gpu(blob, z.shape[0] + 1)


def test_device_entropy_decoder_rejects_what_the_host_parser_rejects(gpu):
    z = np.zeros((4, 64), np.int16)
    z[:, 3] = 5
    good = oracle.rle_bytestream(z)
    assert np.array_equal(gpu.entropy_decode_gpu(good, 4), z)
    for bad, n in ((good[:-1], 4), (good, 5), (b"\x12", 1), (b"\xf0\xf0\xf0\xf0\xf0\x00", 1), (b"\x30\x00", 1),
                   (good[:3] + b"\xff" + good[4:], 4)):
        with pytest.raises(gpu.JpegxError):
            gpu.entropy_decode_gpu(bad, n)
        with pytest.raises(gpu.JpegxError):
            gpu.entropy_decode(bad, n)


def test_device_and_host_decoders_agree_on_damaged_streams(gpu):
    """Seeded fuzz of the two decoders (parallel on the device, sequential C++ on the host) on random bytes and on
    well-formed streams with flipped bits, truncations and inserted bytes: the device never accepts what the host
    parser rejects, what both accept decodes to the same coefficients, and no input faults or hangs the device.
    The device is stricter in exactly one respect: it finds block starts behind 0x00 bytes, so a block whose
    zero padding after the end marker has been damaged is refused there (decompress_band then takes the host
    parser, which skips padding bits unread) -- the encoder never writes such padding (rle_byte_stream.py:55-56)."""
    rng = np.random.default_rng(5)
    seen = {"equal": 0, "both refuse": 0, "device stricter": 0}
    for trial in range(240):
        nb = int(rng.integers(1, 200))
        kind = trial % 4
        if kind == 0:
            blob = rng.integers(0, 256, int(rng.integers(1, 4096)), dtype=np.uint8).tobytes()
        else:
            zz = (rng.integers(-300, 300, (nb, 64)) * (rng.random((nb, 64)) < rng.random())).astype(np.int16)
            dmg = bytearray(oracle.rle_bytestream(zz.reshape(nb, 1, 64)))
            if kind == 1:
                for _ in range(int(rng.integers(1, 4))):
                    dmg[int(rng.integers(0, len(dmg)))] ^= 1 << int(rng.integers(0, 8))
            elif kind == 2:
                dmg = dmg[:int(rng.integers(1, len(dmg) + 1))]
            else:
                i = int(rng.integers(0, len(dmg)))
                dmg[i:i] = rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8).tobytes()
            blob = bytes(dmg)
        for n in ((nb, max(1, nb - 1)) if trial % 7 == 0 else (nb,)):
            try:
                host = gpu.entropy_decode(blob, n)
            except gpu.JpegxError:
                host = None
            # the judge: the oracle's restatement of the reference's decoder.  What the host parser (hence the device)
            # accepts, the reference accepts with the same values; the reverse does not hold -- the reference reads
            # short at the end of the stream and drops codes behind the last end marker (test_oracle_golden.py)
            try:
                ref = oracle.rle_decode(blob, n)
            except oracle.RleStreamError:
                ref = None
            assert not (host is not None and ref is None), (trial, n)
            if host is not None:
                assert np.array_equal(host, ref), (trial, n)
            seen["reference accepts, host parser refuses"] = seen.get("reference accepts, host parser refuses", 0) + \
                (ref is not None and host is None)
            try:
                dev = gpu.entropy_decode_gpu(blob, n)
            except gpu.JpegxError:
                dev = None
            assert not (dev is not None and host is None), (trial, n)
            if dev is not None:
                assert np.array_equal(dev, host), (trial, n)
                seen["equal"] += 1
            else:
                seen["both refuse" if host is None else "device stricter"] += 1
    assert seen["equal"] >= 10 and seen["both refuse"] >= 100 and seen["device stricter"] <= 5, seen
    assert seen["reference accepts, host parser refuses"] <= 12, seen


def test_segmented_and_general_decoders_agree(gpu, monkeypatch):
    """The two device schemes for finding block starts -- by segments in LDS (five launches, no host round trip) and by
    pointer jumping over the whole stream (the fallback) -- give the same coefficients as the host parser, on streams
    that exercise the segment logic: blocks straddling segment boundaries, segments whose entries disagree, a last
    segment holding only the tail of the last block, all-zero planes (every byte a block: more candidates than a
    segment's tables hold, so the segmented scheme hands the stream back), sparse and dense content."""
    rng = np.random.default_rng(12)
    streams = []
    for nblocks, density, amp in ((5000, 0.9, 300), (20000, 0.2, 40), (3000, 0.02, 4), (700, 1.0, 16383), (4099, 0.5, 2000)):
        z = (rng.integers(-amp, amp + 1, (nblocks, 64)) * (rng.random((nblocks, 64)) < density)).astype(np.int16)
        z[::7, ::3] = rng.choice(np.array([256, -512, 4096, 8192, -16384 + 256, 1, -1], dtype=np.int16), z[::7, ::3].shape)
        streams.append(z)
    streams.append(np.zeros((9000, 64), np.int16))                      # 9000 bytes of 0x00
    streams.append(np.zeros((1, 64), np.int16))
    for z in streams:
        blob = oracle.rle_bytestream(z.reshape(len(z), 1, 64))
        host = gpu.entropy_decode(blob, len(z))
        assert np.array_equal(host, z)
        monkeypatch.delenv("JPEGX_DECODE_GENERAL", raising=False)
        assert np.array_equal(gpu.entropy_decode_gpu(blob, len(z)), z), len(z)
        monkeypatch.setenv("JPEGX_DECODE_GENERAL", "1")
        assert np.array_equal(gpu.entropy_decode_gpu(blob, len(z)), z), len(z)
        monkeypatch.delenv("JPEGX_DECODE_GENERAL", raising=False)
        # cut the stream so that the last segment holds a few bytes only (sizes around multiples of 4096)
        sizes = np.cumsum(oracle.rle_bytestream(z.reshape(len(z), 1, 64), want_block_bytes=True)[1]) if len(z) > 1 else None
        if sizes is not None and sizes[-1] > 9000:
            for target in (4096, 8192):
                i = int(np.searchsorted(sizes, target + 2))
                if 0 < i < len(z):
                    assert np.array_equal(gpu.entropy_decode_gpu(blob[:int(sizes[i])], i + 1), z[:i + 1]), (len(z), target)


def test_decoders_refuse_sign_only_codes_and_trailing_bytes(gpu):
    """Two kinds of damaged stream the reference itself rejects: a code of size 1 -- a sign bit with
    no amplitude bits, int('', 2) in rle_byte_stream.py:35-42 -- and bytes or whole blocks behind the last block of
    the plane (the reference parses them and then fails in its reshape, run_length_encoding.py:77-79).  Device
    decoder, C++ host parser and the Python step class all refuse them."""
    from pipeline import Configuration, QuantizationMethod
    from pipeline.rle_byte_stream import RleBytestream
    z = np.zeros((4, 64), np.int16)
    z[:, 3] = 5
    good = oracle.rle_bytestream(z)
    assert np.array_equal(gpu.entropy_decode_gpu(good, 4), z) and np.array_equal(gpu.entropy_decode(good, 4), z)
    sign_only = bytes([0x01, 0x80, 0x00])                 # run 0, size 1, sign bit '1', then the end marker
    for bad, n in ((sign_only, 1), (good + b"\x00", 4), (good + good[:len(good) // 4], 4), (good, 3), (good + b"\x07", 4)):
        with pytest.raises(gpu.JpegxError):
            gpu.entropy_decode_gpu(bad, n)
        with pytest.raises(gpu.JpegxError):
            gpu.entropy_decode(bad, n)
    with pytest.raises(ValueError):
        RleBytestream(Configuration(8, 8, 1, quantization=QuantizationMethod("none"))).invert(sign_only)


@pytest.mark.parametrize("bs", [1, 2, 3, 4, 5])
def test_decompress_plane_all_on_device(gpu, bs):
    """bytes -> samples without the coefficients ever visiting the host == host parse + fused inverse."""
    a = gpu.synth.generate_plane("noise", 256 * bs, 512 * bs, seed=8)
    blob = gpu.compress_plane(a.astype(np.uint8), bs, "qtable")
    zz = gpu.entropy_decode(blob, 32 * 64).reshape(32, 64, 64)
    want = gpu.inverse_fused_u8(zz, "qtable", inflate=bs)
    assert np.array_equal(gpu.decompress_plane(blob, 256, 512, bs, "qtable"), want)
