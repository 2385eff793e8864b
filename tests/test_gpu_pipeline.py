"""GPU: the reference-compatible step classes and the codec facade, driven exactly like the
reference's own tests drive them (tests/integration_tests.py, tests/zigzag_tests.py ...), with
the reference's outputs (golden vectors) as the expectation."""
import numpy as np
import pytest

import oracle
import pipeline
import transforms
from conftest import CASES
from pipeline import Configuration, QuantizationMethod, compress_band, decompress_band
from pipeline.basis_change import BasisChange
from pipeline.quantization import Quantization
from pipeline.rle_byte_stream import RleBytestream
from pipeline.run_length_encoding import RunLengthEncoding
from pipeline.zigzag_order import ZigzagOrder

pytestmark = pytest.mark.gpu

METHODS = [("qtable", lambda: QuantizationMethod("qtable")), ("none", lambda: QuantizationMethod("none")),
           ("divide40", lambda: QuantizationMethod("divide", divisor=40)),
           ("discard2", lambda: QuantizationMethod("discard", keep=2))]


def config_for(c, method):
    h, w = c["input"].shape
    return Configuration(width=w, height=h, block_size=int(c["block_size"]), dct_size=8, transform="DCT",
                         quantization=method)


@pytest.mark.parametrize("case", CASES)
def test_step_classes_reproduce_the_reference_arrays(gpu, golden, case):
    c = golden(case)
    for suffix, mk in METHODS:
        cfg = config_for(c, mk())
        dct = BasisChange(cfg).execute(c["pre"])
        assert dct.dtype == np.float64 and np.array_equal(dct, c["dct"])
        q = Quantization(cfg).execute(dct)
        assert np.array_equal(q, c["q_" + suffix].astype(np.float64))
        zz = ZigzagOrder(cfg).execute(q)
        assert zz.shape == c["zz_" + suffix].shape and np.array_equal(zz, c["zz_" + suffix].astype(np.float64))
        back = ZigzagOrder(cfg).invert(zz)
        assert np.array_equal(back, q)
        rest = Quantization(cfg).invert(back)
        assert np.array_equal(rest, c["restore_" + suffix].astype(np.float64))
        rec = BasisChange(cfg).invert(rest)
        assert rec.dtype.kind == "i" and np.array_equal(rec, c["idct_" + suffix])


@pytest.mark.parametrize("case", CASES)
def test_compress_band_and_back_match_the_reference(gpu, golden, case):
    """compress_band's hidden hot-path output (recovered by undoing the entropy stage) and
    decompress_band's final band equal the reference's, for all four quantisers."""
    c = golden(case)
    band = c["input"].astype(np.int64)
    for suffix, mk in METHODS:
        cfg = config_for(c, mk())
        blob = compress_band(band, cfg)
        assert isinstance(blob, bytes)
        zz = RunLengthEncoding(cfg).invert(RleBytestream(cfg).invert(blob))
        assert np.array_equal(zz, c["zz_" + suffix])
        restored = decompress_band(blob, cfg)
        assert np.array_equal(restored, c["band_" + suffix])


def test_reference_integration_cases(gpu):
    """/root/reference/tests/integration_tests.py:11-48 with its tolerances."""
    original = np.arange(128).reshape(8, 16)
    cfg = Configuration(width=16, height=8, block_size=3)          # means k/9: float64 exact-kernel path
    assert np.allclose(original, decompress_band(compress_band(original, cfg), cfg), rtol=1)
    cfg = Configuration(width=16, height=8, block_size=3, transform="DFT")
    assert np.allclose(original, decompress_band(compress_band(original, cfg), cfg), rtol=1)
    small = np.arange(6).reshape(2, 3)
    cfg = Configuration(width=3, height=2, block_size=1)
    assert np.allclose(small, decompress_band(compress_band(small, cfg), cfg), rtol=0.000001)


def test_non_fp32_representable_input_takes_the_exact_path(gpu):
    """block_size 3 produces thirds: the facade must not round them to fp32."""
    import oracle
    rng = np.random.default_rng(5)
    band = rng.integers(0, 256, (48, 72))
    cfg = Configuration(width=72, height=48, block_size=3, quantization=QuantizationMethod("qtable"))
    pre = band.reshape(16, 3, 24, 3).mean(axis=(1, 3))
    want = oracle.zigzag_plane(oracle.quant_plane(oracle.dct_plane(pre), "qtable"))
    blob = compress_band(band, cfg)
    zz = RunLengthEncoding(cfg).invert(RleBytestream(cfg).invert(blob))
    assert np.array_equal(zz, want)


def test_dct_class_on_the_gpu(gpu, tables):
    d = transforms.DCT(8)
    a = np.arange(64).reshape(8, 8)
    y = d.transform_2d(a)
    import oracle
    assert np.array_equal(y, oracle.dct_plane(a))
    assert np.allclose(a, d.transform_2d_inverse(y), rtol=0.01)


def test_jpeg_facade_round_trip_through_pil(gpu):
    from PIL import Image
    from jpegx import synth
    h, w = 40, 56
    rgb = np.dstack([synth.generate_plane("smooth", h, w, seed=s, dtype=np.uint8) for s in (1, 2, 3)])
    im = Image.fromarray(rgb, mode="RGB").convert("YCbCr")
    cfg = Configuration(width=w, height=h, block_size=2, dct_size=8, quantization=QuantizationMethod("qtable"))
    blob = pipeline.Jpeg(cfg).compress(im)
    out = pipeline.Jpeg.decompress(blob)
    assert out.size == (w, h) and out.mode == "YCbCr"
    err = np.abs(np.asarray(out, dtype=np.int64)[..., 0] - np.asarray(im, dtype=np.int64)[..., 0])
    assert err.mean() < 12


def test_cli_compress_decompress_files(gpu, tmp_path):
    """compress.py / decompress.py entry points on real files (reference: compress.py:6-20, decompress.py:5-10)."""
    from PIL import Image
    import compress
    import decompress
    import file_format
    from jpegx import synth
    h, w = 72, 104                                       # not multiples of 16: exercises both paddings
    rgb = np.dstack([synth.generate_plane("smooth", h, w, seed=s, dtype=np.uint8) for s in (4, 5, 6)])
    src, packed, back = tmp_path / "in.png", tmp_path / "out.bin", tmp_path / "back.png"
    Image.fromarray(rgb, mode="RGB").save(src)
    for bs, q in ((2, QuantizationMethod("qtable")), (1, QuantizationMethod("divide", divisor=10)), (4, None)):
        compress.compress(str(src), str(packed), block_size=bs, dct_size=8, transform="DCT", quantization=q)
        cfg, data = file_format.read_data(packed.read_bytes())
        assert (cfg.width, cfg.height, cfg.block_size, cfg.dct_size) == (w, h, bs, 8)
        decompress.decompress(str(packed), str(back))
        out = np.asarray(Image.open(back).convert("RGB"), dtype=np.int64)
        assert out.shape == rgb.shape
        assert np.abs(out - rgb.astype(np.int64)).mean() < (6 if bs == 1 else 16)


def test_decompress_band_fast_path_equals_generic_path(gpu, golden):
    """The fused back end (C++ entropy parse + one launch) and the step-by-step walk agree."""
    from pipeline.base import step_classes
    c = golden("pooled128")
    band = c["input"].astype(np.int64)
    cfg = config_for(c, QuantizationMethod("qtable"))
    blob = compress_band(band, cfg)
    fast = decompress_band(blob, cfg)
    a = blob
    for cls in reversed(step_classes):                   # the reference's plain loop
        a = cls(cfg).invert(a)
    assert np.array_equal(fast, a) and np.array_equal(fast, c["band_qtable"])


def test_stream_with_damaged_padding_takes_the_host_parser_and_gives_its_result(gpu, golden):
    """The device decoder refuses a stream whose zero padding behind an end marker has been tampered with (it finds
    block starts behind 0x00 bytes); the sequential parser -- like the reference's -- never reads padding bits.
    decompress_band must then return what the step-by-step walk over the damaged stream returns."""
    import jpegx
    from pipeline.base import step_classes
    c = golden("pooled128")
    cfg = config_for(c, QuantizationMethod("qtable"))
    blob = bytearray(compress_band(c["input"].astype(np.int64), cfg))
    nblocks = (c["input"].shape[0] // cfg.block_size // 8) * (c["input"].shape[1] // cfg.block_size // 8)
    damaged = None
    for i in range(1, len(blob) - 1):                      # a 0x00 byte that ends a block early in the stream: set its last bit
        if blob[i] == 0:
            trial = bytes(blob[:i]) + b"\x01" + bytes(blob[i + 1:])
            try:
                jpegx.entropy_decode(trial, nblocks)
            except jpegx.JpegxError:
                continue
            try:
                jpegx.entropy_decode_gpu(trial, nblocks)
            except jpegx.JpegxError:
                damaged = trial
                break
    assert damaged is not None, "no padding bit found to tamper with"
    a = damaged
    for cls in reversed(step_classes):
        a = cls(cfg).invert(a)
    assert np.array_equal(decompress_band(damaged, cfg), a)
    assert np.array_equal(pipeline.decompress_band_u8(damaged, cfg), a.astype(np.uint8))


def _step_by_step(band, cfg, classes):
    a = band
    for cls in classes:
        a = cls(cfg).execute(a)
    return a


def test_user_step_between_the_hot_steps_runs_at_its_place(gpu, golden):
    """Plugin contract (pipeline/base.py:23-31): a step registered with 4 < step_index < 5 sees the DCT
    coefficients, not the zigzagged integers -- the three hot steps are then NOT fused.  Compared with
    executing every registered class one by one; the registry is restored afterwards."""
    from pipeline.base import AlgorithmStep, step_classes
    c = golden("noise64")
    band = c["input"].astype(np.int64)
    cfg = config_for(c, QuantizationMethod("qtable"))
    stock = list(step_classes)
    try:
        class HalveHighRows(AlgorithmStep):
            step_index = 4.5

            def execute(self, array):
                out = np.array(array, dtype=np.float64)
                out[4::8, :] *= 0.5
                return out

            def invert(self, array):
                out = np.array(array, dtype=np.float64)
                out[4::8, :] *= 2.0
                return out
        assert not pipeline._stock_registry() and pipeline._hot_run(list(step_classes)) is None
        want = _step_by_step(band, cfg, list(step_classes))
        got = compress_band(band, cfg)
        assert got == want
        assert got != _step_by_step(band, cfg, stock)                         # the user step really took part
        back = decompress_band(got, cfg)
        a = got
        for cls in reversed(list(step_classes)):
            a = cls(cfg).invert(a)
        assert np.array_equal(back, a)
    finally:
        step_classes[:] = stock
    assert pipeline._stock_registry()


def test_edited_quantisation_table_is_honoured(gpu, golden):
    """A QuantizationMethod whose quantiser object is not the stock one (here: a doubled table) must not be
    mapped onto the built-in 'qtable' kernel mode: the result follows the object, as in the reference."""
    c = golden("smooth64")
    band = c["input"].astype(np.int64)
    m = QuantizationMethod("qtable")
    m.quantizer._qtable = m.quantizer._qtable * 2
    cfg = config_for(c, m)
    blob = compress_band(band, cfg)
    zz = RunLengthEncoding(cfg).invert(RleBytestream(cfg).invert(blob))
    dct = BasisChange(cfg).execute(c["pre"])
    want = np.round(dct.reshape(8, 8, 8, 8).transpose(0, 2, 1, 3) * (1.0 / m.quantizer._qtable)).transpose(0, 2, 1, 3).reshape(64, 64)
    assert np.array_equal(ZigzagOrder(cfg).invert(zz), want)
    assert not np.array_equal(zz, c["zz_qtable"])


def test_amplitudes_beyond_15_bits_raise_the_reference_exception(gpu):
    """divide with a tiny divisor on an 8-bit band: DC / 0.01 needs more than 15 bits.  The reference raises
    util.BadRleCodeError from step 7; so do the fused path (uint8 and fp32 input) and the step-by-step path."""
    import util
    band = np.full((16, 32), 255, dtype=np.int64)
    for bs in (1, 2):
        cfg = Configuration(width=32, height=16, block_size=bs, quantization=QuantizationMethod("divide", divisor=0.01))
        with pytest.raises(util.BadRleCodeError):
            compress_band(band, cfg)
    cfg = Configuration(width=32, height=16, block_size=1, quantization=QuantizationMethod("divide", divisor=0.01))
    with pytest.raises(util.BadRleCodeError):
        _step_by_step(band, cfg, pipeline._BUILTIN_STEPS)
    # the raw kernels saturate instead of wrapping around
    zz = gpu.forward_fused(band.astype(np.float32), "divide", 0.01)
    assert zz[0, 0, 0] == 32767
    with pytest.raises(gpu.JpegxError):
        gpu.forward_fused_u8(band.astype(np.uint8), 1, "divide", 0.01)
    with pytest.raises(util.BadRleCodeError):
        decompress_band(b"\x30\x00", cfg)                    # run 3 with size 0: not a legal code


def test_block_size_3_runs_on_the_device_and_matches_the_reference(gpu, golden):
    """block_size 3 (means k/9 are not fp32 numbers): SubSampling on the device in float64 (jpegx_mean_pool_f64),
    the all-float64 fused forward (jpegx_forward_fused_f64) and the device entropy stage -- no host pooling --
    against the reference's arrays for the same band (case_pooled3x72.npz)."""
    c = golden("pooled3x72")
    band = c["input"].astype(np.int64)
    assert np.array_equal(gpu.mean_pool_f64(band.astype(np.uint8), 3), c["pre"])
    assert np.array_equal(gpu.mean_pool_f64(band.astype(np.float32), 3), c["pre"])
    for suffix, mk in METHODS:
        assert np.array_equal(gpu.forward_fused_f64(c["pre"], *mk().gpu_mode()), c["zz_" + suffix])
        cfg = config_for(c, mk())
        for b in (band, band.astype(np.uint8)):
            blob = compress_band(b, cfg)
            assert isinstance(blob, bytes)
            zz = RunLengthEncoding(cfg).invert(RleBytestream(cfg).invert(blob))
            assert np.array_equal(zz, c["zz_" + suffix])
            assert np.array_equal(decompress_band(blob, cfg), c["band_" + suffix])
    # the native path really took it (not the host-pooling fallback)
    assert gpu.compress_plane_native(band, 3, "qtable", 0.0) == compress_band(band, config_for(c, QuantizationMethod("qtable")))
    # a float64 plane straight into steps 4-6 (BasisChange input that is not exact in fp32)
    assert np.array_equal(pipeline._hot_forward(c["pre"], config_for(c, QuantizationMethod("qtable"))), c["zz_qtable"].astype(np.float64))


def test_block_size_3_decompresses_in_one_native_call(gpu, golden):
    """SubSampling.invert for a block_size the kernel does not know at compile time: the replication factor is a
    run-time argument of the fused inverse (jpegx_inverse_fused_u8_inflated), so decompress_band with block_size 3
    is one native job -- bytes up, device decoder, inverse + clamp + 3 x 3 replication, samples down -- and equals
    the reference's band (case_pooled3x72.npz band_*)."""
    c = golden("pooled3x72")
    for suffix, mk in METHODS:
        cfg = config_for(c, mk())
        mode, param = mk().gpu_mode()
        zz = c["zz_" + suffix]
        blob = oracle.rle_bytestream(zz)
        want = c["band_" + suffix]
        hb, wb = zz.shape[:2]
        assert np.array_equal(gpu.decompress_plane(blob, hb * 8, wb * 8, 3, mode, param), want.astype(np.uint8)), suffix
        assert np.array_equal(gpu.inverse_fused_u8(zz, mode, param, inflate=3), want.astype(np.uint8))
        assert np.array_equal(pipeline.decompress_band_u8(blob, cfg), want.astype(np.uint8))
        assert np.array_equal(decompress_band(blob, cfg), want)
    # other factors against plain replication of the un-inflated result
    zz = c["zz_qtable"]
    base = gpu.inverse_fused_u8(zz, "qtable", 0.0, inflate=1)
    for rep in (5, 6, 7, 9, 16, 31):
        assert np.array_equal(gpu.inverse_fused_u8(zz, "qtable", 0.0, inflate=rep), np.repeat(np.repeat(base, rep, axis=0), rep, axis=1)), rep


@pytest.mark.parametrize("bs", [1, 2, 3])
def test_whole_image_job_equals_three_band_jobs(gpu, bs):
    """Jpeg.compress / Jpeg.decompress through ONE native job per picture (jpegx_host_compress_image /
    _decompress_image: bands on two alternating streams, the bytes of band k copied into their own bytes object
    while band k + 1 is computed, np.dstack done on the device): same container bytes as three compress_band calls
    (pipeline/__init__.py:102-110), same picture as three decompress_band calls."""
    from PIL import Image
    import file_format
    h, w = 96 * bs, 160 * bs
    rgb = np.stack([gpu.synth.generate_plane("smooth", h, w, seed=s) for s in (1, 2, 3)], axis=-1).astype(np.uint8)
    im = Image.fromarray(rgb, mode="RGB").convert("YCbCr")
    for mk in (lambda: QuantizationMethod("qtable"), lambda: QuantizationMethod("divide", divisor=7)):
        cfg = Configuration(width=w, height=h, block_size=bs, quantization=mk())
        per_band = [compress_band(np.asarray(b), cfg) for b in im.split()]
        blobs = gpu.compress_image_native([np.ascontiguousarray(np.asarray(b)) for b in im.split()], bs, *cfg.quantization.gpu_mode())
        assert blobs == per_band
        # the same from the interleaved pixels in one piece (what Jpeg.compress hands over since round 3: no image.split())
        assert gpu.compress_image_packed(np.asarray(im), bs, *cfg.quantization.gpu_mode()) == per_band
        four = np.concatenate([np.asarray(im), 255 - np.asarray(im)[..., :1]], axis=-1)
        assert gpu.compress_image_packed(np.ascontiguousarray(four), bs, *cfg.quantization.gpu_mode())[:3] == per_band
        data = pipeline.Jpeg(cfg).compress(im)
        assert data == file_format.generate_data(cfg, pipeline.CompressedData(*per_band))
        back = pipeline.Jpeg.decompress(data)
        want = np.dstack([pipeline.decompress_band_u8(b, cfg) for b in per_band])
        assert np.array_equal(np.asarray(back), want)
        planar = gpu.decompress_image_native(per_band, h // bs, w // bs, bs, *cfg.quantization.gpu_mode(), h, w, interleave=False)
        assert np.array_equal(planar, np.moveaxis(want, 2, 0))
    # a ragged picture (padding to the block size, cropping on the way back) and int64 bands
    cfg = Configuration(width=150, height=90, block_size=1, quantization=QuantizationMethod("qtable"))
    small = im.crop((0, 0, 150, 90))
    data = pipeline.Jpeg(cfg).compress(small)
    bands = [compress_band(np.asarray(b).astype(np.int64), cfg) for b in small.split()]
    assert data == file_format.generate_data(cfg, pipeline.CompressedData(*bands))
    assert np.array_equal(np.asarray(pipeline.Jpeg.decompress(data)), np.dstack([pipeline.decompress_band_u8(b, cfg) for b in bands]))


def test_open_compress_job_refuses_a_second_pooled_call_on_its_thread(gpu):
    """jpegx_host_compress_begin keeps the device's pool until _finish / _abort.  A second pooled entry on the SAME
    thread -- another _begin, a host-pointer convenience, a decompress, a pool release -- used to lock the pool's
    non-recursive mutex again and hang; now it returns JPEGX_E_INVALID, and _finish / _abort find the thread's own
    job whatever the thread's current device is."""
    import ctypes
    L = gpu.lib()
    band = gpu.synth.generate_plane("smooth", 64, 128, seed=3).astype(np.uint8)
    n = ctypes.c_size_t(0)
    gpu.check(L.jpegx_host_compress_begin(band.ctypes.data, 1, 64, 128, 128, 1, gpu.Q_QTABLE, 0.0, ctypes.byref(n)), "begin")
    try:
        assert L.jpegx_host_compress_begin(band.ctypes.data, 1, 64, 128, 128, 1, gpu.Q_QTABLE, 0.0, ctypes.byref(n)) == -1
        assert b"open on this thread" in L.jpegx_last_error()
        with pytest.raises(gpu.JpegxError):
            gpu.forward_fused(band.astype(np.float32), "qtable")
        with pytest.raises(gpu.JpegxError):
            gpu.decompress_plane(b"\x00" * 128, 64, 128, 1, "qtable")
        assert L.jpegx_host_pool_release() == -1
    finally:
        out = ctypes.create_string_buffer(max(1, n.value))
        gpu.check(L.jpegx_host_compress_finish(out), "finish")
    assert out.raw[:n.value] == gpu.compress_plane(band, 1, "qtable")
    assert L.jpegx_host_compress_finish(out) == -1          # nothing open any more
    assert L.jpegx_host_compress_abort() == 0


def test_saturated_int16_is_not_mistaken_for_a_coefficient(gpu):
    """_hot_forward's all-float64 kernel saturates to int16: a coefficient at or below -32768 must send the band
    down the exact step-by-step road (np.abs of int16 -32768 wraps to -32768 and used to let it through, ADVICE
    round 2).  A float64 plane whose samples are not fp32 numbers and whose first block has DC = -38421."""
    pre = np.full((8, 16), 1.0 / 3.0)
    pre[:, :8] = -600.0 - 1.0 / 3.0
    cfg = Configuration(width=16, height=8, block_size=1, quantization=QuantizationMethod("none"))
    want = oracle.zigzag_plane(oracle.quant_plane(oracle.dct_plane(pre), "none"))
    assert want[0, 0, 0] == -38421.0
    assert gpu.forward_fused_f64(pre, "none")[0, 0, 0] == -32768          # what the saturating kernel hands back
    assert np.array_equal(pipeline._hot_forward(pre, cfg), want)


def test_explicit_device_forms_of_the_entries(gpu):
    """jpegx_<name>_on(device, ...): the same work with the device named in the call; the thread's
    current device is left as it was, a device that does not exist is JPEGX_E_NODEVICE."""
    import ctypes
    L = gpu.lib()
    a = gpu.synth.generate_plane("noise", 64, 128, seed=2).astype(np.uint8)
    din, dzz = ctypes.c_void_p(), ctypes.c_void_p()
    gpu.check(L.jpegx_malloc_on(0, ctypes.byref(din), a.nbytes), "malloc_on")
    gpu.check(L.jpegx_malloc_on(0, ctypes.byref(dzz), a.size * 2), "malloc_on")
    try:
        gpu.check(L.jpegx_memcpy_h2d(din, a.ctypes.data, a.nbytes, None), "h2d")
        gpu.check(L.jpegx_forward_fused_u8_on(0, din, 64, 128, 128, 1, gpu.Q_QTABLE, 0.0, 0, dzz, None), "forward_fused_u8_on")
        zz = np.empty((8, 16, 64), np.int16)
        gpu.check(L.jpegx_memcpy_d2h(zz.ctypes.data, dzz, zz.nbytes, None), "d2h")
        gpu.check(L.jpegx_device_synchronize(), "sync")
        assert np.array_equal(zz, oracle.forward_f32(a.astype(np.float32), "qtable"))
        out = np.empty((64, 128), np.uint8)
        blob = gpu.compress_plane(a, 1, "qtable")
        buf = np.frombuffer(blob, np.uint8)
        gpu.check(L.jpegx_host_decompress_plane_on(0, buf.ctypes.data, buf.size, 64, 128, 1, gpu.Q_QTABLE, 0.0, out.ctypes.data, 128), "decompress_on")
        assert np.array_equal(out, gpu.decompress_plane(blob, 64, 128, 1, "qtable"))
        assert L.jpegx_forward_fused_u8_on(63, din, 64, 128, 128, 1, gpu.Q_QTABLE, 0.0, 0, dzz, None) == -3
        dev = ctypes.c_int(-1)
        gpu.check(L.jpegx_get_device(ctypes.byref(dev)), "get_device")
        assert dev.value == 0
    finally:
        L.jpegx_free_on(0, din)
        L.jpegx_free_on(0, dzz)


@pytest.mark.parametrize("dtype", [np.int64, np.int32])
def test_wide_integer_bands_are_narrowed_in_strips_and_uploaded_as_they_come(gpu, dtype):
    """What util.band_to_array hands compress_band is an int64 array (util.py:110-112).  The native job narrows it to bytes
    on host threads strip by strip, each strip on its way to the device while the next is narrowed (large bands only:
    this one takes that path, the golden 64 x 64 cases do not): same bytes as the uint8 band, for heights that do not
    divide into the strips and threads evenly; a sample outside 0..255 anywhere -- the last row included -- is not an
    8-bit band, and the job hands it to the step-by-step path like the reference would run it."""
    for h, w in ((2048, 2048), (1040, 1552), (4096, 272)):
        band8 = gpu.synth.generate_plane("smooth", h, w, seed=h + w, dtype=np.int64).astype(np.uint8)
        want = gpu.compress_plane_native(band8, 1, "qtable", 0.0)
        wide = band8.astype(dtype)
        assert gpu.compress_plane_native(wide, 1, "qtable", 0.0) == want, (h, w)
        assert gpu.compress_plane_native(wide, 2, "qtable", 0.0) == gpu.compress_plane_native(band8, 2, "qtable", 0.0)
        for y, x, v in ((h - 1, w - 1, 256), (0, 0, -1), (h // 2 + 3, 5, 1 << 20)):
            bad = wide.copy()
            bad[y, x] = v
            assert gpu.compress_plane_native(bad, 1, "qtable", 0.0) is None, (h, w, y, x)
        assert gpu.compress_plane_native(wide, 1, "qtable", 0.0) == want                    # and the pool is fine afterwards
    cfg = Configuration(width=2048, height=2048, block_size=1, dct_size=8, quantization=QuantizationMethod("qtable"))
    band8 = gpu.synth.generate_plane("noise", 2048, 2048, seed=3, dtype=np.int64).astype(np.uint8)
    assert compress_band(band8.astype(np.int64), cfg) == compress_band(band8, cfg)


def test_concurrent_callers_share_a_device(gpu, golden):
    """Several host threads in the band calls at once (ctypes drops the GIL): a device has four job contexts, every
    context its own streams, buffers and decoder state, so the calls overlap instead of queueing behind one lock -- and
    every one of them returns what it returns alone."""
    import threading
    cfg = Configuration(width=1024, height=1024, block_size=1, dct_size=8, quantization=QuantizationMethod("qtable"))
    cfg2 = Configuration(width=1024, height=1024, block_size=2, dct_size=8, quantization=QuantizationMethod("divide", divisor=9))
    bands = [gpu.synth.generate_plane(k, 1024, 1024, seed=s, dtype=np.int64).astype(np.uint8) for k in ("noise", "smooth") for s in (1, 2, 3)]
    want = [(compress_band(b, cfg), compress_band(b, cfg2)) for b in bands]
    back = [(pipeline.decompress_band_u8(w[0], cfg), pipeline.decompress_band_u8(w[1], cfg2)) for w in want]
    errors = []

    def worker(tid):
        try:
            for rep in range(12):
                i = (tid + rep) % len(bands)
                if compress_band(bands[i], cfg) != want[i][0] or compress_band(bands[i].astype(np.int64), cfg2) != want[i][1]:
                    errors.append(("compress", tid, rep))
                if not np.array_equal(pipeline.decompress_band_u8(want[i][0], cfg), back[i][0]) or \
                        not np.array_equal(np.asarray(decompress_band(want[i][1], cfg2)), back[i][1]):
                    errors.append(("decompress", tid, rep))
        except Exception as exc:                                            # noqa: BLE001 -- reported below
            errors.append((type(exc).__name__, str(exc)[:200], tid))
    threads = [threading.Thread(target=worker, args=(t,)) for t in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:5]
