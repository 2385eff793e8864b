"""GPU parity (MI355X): every kernel of libjpegx.so, through the C ABI, against the oracle and
against the golden vectors produced by the unmodified reference (tests/golden/make_golden.py)."""
import ctypes

import numpy as np
import pytest

import oracle
from conftest import CASES, MODES

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("suffix,mode,param", MODES)
def test_forward_fused_matches_reference_golden(gpu, golden, case, suffix, mode, param):
    """Steps 4+5+6 fused: quantised + zigzagged integers bit-exact vs the reference's output."""
    c = golden(case)
    pre = c["pre"].astype(np.float32)
    assert np.array_equal(pre.astype(np.float64), c["pre"])  # inputs are exact in fp32
    got = gpu.forward_fused(pre, mode, param)
    assert got.dtype == np.int16 and got.shape == c["zz_" + suffix].shape
    assert np.array_equal(got, c["zz_" + suffix])
    # the generic (no pixel promise) variant must agree too
    got2 = gpu.forward_fused(pre, mode, param, pixel_input=False)
    assert np.array_equal(got2, c["zz_" + suffix])


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("suffix,mode,param", MODES)
def test_inverse_fused_matches_reference_golden(gpu, golden, case, suffix, mode, param):
    """Steps 6+5+4 inverted, fused: rounded samples bit-exact vs BasisChange.invert of the reference."""
    c = golden(case)
    zz = c["zz_" + suffix]
    want = c["idct_" + suffix]
    got = gpu.inverse_fused(zz, mode, param, out="f32")
    assert np.array_equal(got.astype(np.int64), want)
    got16 = gpu.inverse_fused(zz, mode, param, out="i16")
    assert np.array_equal(got16.astype(np.int64), want)
    gotu8 = gpu.inverse_fused(zz, mode, param, out="u8")
    assert np.array_equal(gotu8.astype(np.int64), np.clip(want, 0, 255))
    gotc = gpu.inverse_fused(zz, mode, param, out="f32", clamp=True)
    assert np.array_equal(gotc.astype(np.int64), np.clip(want, 0, 255))


@pytest.mark.parametrize("case", CASES)
def test_exact_f64_stage_kernels_bit_identical(gpu, golden, case):
    """The float64 stage kernels reproduce the reference's float64 arrays bit for bit."""
    c = golden(case)
    dct = gpu.dct8x8_f64(c["pre"])
    assert np.array_equal(dct, c["dct"])
    for suffix, mode, param in MODES:
        q = gpu.quantize_f64(dct, mode, param)
        assert np.array_equal(q, c["q_" + suffix].astype(np.float64))
        z = gpu.zigzag(q)
        assert np.array_equal(z, c["zz_" + suffix].astype(np.float64))
        back = gpu.unzigzag(z)
        assert np.array_equal(back, q)
        r = gpu.restore_f64(back, mode, param)
        assert np.array_equal(r, c["restore_" + suffix].astype(np.float64))
        i = gpu.idct8x8_f64(r, do_round=True)
        assert np.array_equal(i, c["idct_" + suffix].astype(np.float64))
        fl = gpu.idct8x8_f64(r, do_round=False)
        assert np.array_equal(fl, oracle.idct_plane(r, rounded=False))


@pytest.mark.parametrize("case", CASES)
def test_dct_f32_within_1e4_of_reference(gpu, golden, case):
    """Unfused fp32 DCT: |c_gpu - c_ref| <= 1e-4 * max(1, max_block |c_ref|) (north_star tolerance)."""
    c = golden(case)
    ref = c["dct"]
    got = gpu.dct8x8_f32(c["pre"].astype(np.float32)).astype(np.float64)
    h, w = ref.shape
    blockmax = np.abs(ref).reshape(h // 8, 8, w // 8, 8).max(axis=(1, 3))
    tol = 1e-4 * np.maximum(1.0, blockmax)
    err = np.abs(got - ref).reshape(h // 8, 8, w // 8, 8).max(axis=(1, 3))
    assert np.all(err <= tol), float((err / tol).max())
    # and the fp32 inverse brings the coefficients back to the samples
    back = gpu.idct8x8_f32(got.astype(np.float32))
    assert np.allclose(back, c["pre"], atol=2e-3)


def test_zigzag_element_sizes(gpu):
    rng = np.random.default_rng(3)
    for dt in (np.int16, np.float32, np.float64, np.complex128):
        a = rng.integers(-1000, 1000, (16, 24)).astype(dt)
        z = gpu.zigzag(a)
        assert np.array_equal(z, oracle.zigzag_plane(a.real).astype(dt) if dt != np.complex128
                              else oracle.zigzag_plane(a.real).astype(dt))
        assert np.array_equal(gpu.unzigzag(z), a)


@pytest.mark.parametrize("kind", ["noise", "smooth"])
def test_generator_matches_host_twin(gpu, kind):
    h, w = 64, 256
    buf = gpu.DeviceBuffer(h * w * 4)
    gpu.generate_plane_device(buf.ptr, h, w, kind, seed=5, plane=3, row0=16)
    got = buf.download((h, w), np.float32)
    want = gpu.synth.generate_plane(kind, h, w, seed=5, plane=3, row0=16)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("kind,size", [("noise", 1024), ("smooth", 1024), ("noise", 4096)])
@pytest.mark.parametrize("suffix,mode,param", MODES + [("divide7", "divide", 7.0)])
def test_forward_fused_vs_oracle_large(gpu, kind, size, suffix, mode, param):
    """Config-2-sized planes against the C oracle on identical synthetic input (bit-exact)."""
    if size == 4096 and mode != "qtable":
        pytest.skip("full-size plane is checked for the headline quantiser only")
    a = gpu.synth.generate_plane(kind, size, size, seed=11)
    got = gpu.forward_fused(a, mode, param)
    want = oracle.forward_f32(a, mode, param)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("kind", ["noise", "smooth"])
@pytest.mark.parametrize("suffix,mode,param", MODES)
def test_inverse_fused_vs_oracle_large(gpu, kind, suffix, mode, param):
    a = gpu.synth.generate_plane(kind, 1024, 1024, seed=12)
    zz = oracle.forward_f32(a, mode, param)
    got = gpu.inverse_fused(zz, mode, param, out="i16")
    want = oracle.inverse_i16(zz, mode, param)
    assert np.array_equal(got.astype(np.int32), want)


def test_pooled_forward_matches_reference_golden(gpu, golden):
    """Fused 2x2 mean prologue (config 3 chroma path) == SubSampling + steps 4-6 of the reference."""
    c = golden("pooled128")
    raw = c["input"].astype(np.float32)
    for suffix, mode, param in MODES:
        got = gpu.forward_fused_pooled(raw, 2, mode, param)
        assert np.array_equal(got, c["zz_" + suffix])
    # 4x4 pooling against the oracle
    a = gpu.synth.generate_plane("noise", 256, 256, seed=21)
    pooled = oracle.mean_pool(a, 4)
    got = gpu.forward_fused_pooled(a, 4, "qtable")
    assert np.array_equal(got, oracle.forward_f32(pooled.astype(np.float32), "qtable"))


def test_ragged_block_counts(gpu):
    """Block counts that are not multiples of 64 exercise the partial last wave."""
    for h, w in [(8, 8), (8, 40), (24, 72), (16, 520), (72, 8)]:
        a = gpu.synth.generate_plane("noise", h, w, seed=h * 131 + w)
        assert np.array_equal(gpu.forward_fused(a, "qtable"), oracle.forward_f32(a, "qtable"))
        zz = oracle.forward_f32(a, "qtable")
        assert np.array_equal(gpu.inverse_fused(zz, "qtable", out="i16").astype(np.int32), oracle.inverse_i16(zz, "qtable"))


def test_round_trip_psnr(gpu):
    """Config 4: forward then inverse on the GPU; PSNR vs input matches the oracle's round trip."""
    a = gpu.synth.generate_plane("smooth", 1024, 1024, seed=4)
    zz = gpu.forward_fused(a, "qtable")
    rec = gpu.inverse_fused(zz, "qtable", out="u8").astype(np.float64)
    mse = np.mean((rec - a) ** 2)
    psnr = 10 * np.log10(255.0 ** 2 / mse)
    ref = np.clip(oracle.inverse_i16(oracle.forward_f32(a, "qtable"), "qtable"), 0, 255)
    psnr_ref = 10 * np.log10(255.0 ** 2 / np.mean((ref - a) ** 2))
    assert abs(psnr - psnr_ref) < 1e-9
    assert psnr > 30.0


def test_bad_arguments_raise(gpu):
    a = np.zeros((12, 16), dtype=np.float32)
    with pytest.raises(gpu.JpegxError):
        gpu.forward_fused(a, "qtable")          # height not a multiple of 8
    with pytest.raises(gpu.JpegxError):
        gpu.forward_fused(np.zeros((8, 8), np.float32), "divide", 0.0)
    with pytest.raises(gpu.JpegxError):
        gpu.forward_fused(np.zeros((8, 8), np.float32), "bogus")


@pytest.mark.parametrize("flags", [0x100, 0x200, 0x300, 0x800, 0x900, 0x4000, 0x4004, 0x4100, 0x8000])
@pytest.mark.parametrize("kind", ["noise", "smooth"])
def test_forward_tuning_variants_are_bit_identical(gpu, kind, flags):
    """Cache-policy and LDS-strip variants of the fused forward kernel give the same integers."""
    a = gpu.synth.generate_plane(kind, 256, 1024, seed=31)          # W/8 = 128: strip variant eligible
    want = oracle.forward_f32(a, "qtable")
    for pixel in (True, False):
        assert np.array_equal(gpu.forward_fused(a, "qtable", pixel_input=pixel, flags_extra=flags), want)
    for mode, param in (("none", 0.0), ("divide", 7.0), ("discard", 3.0)):
        assert np.array_equal(gpu.forward_fused(a, mode, param, flags_extra=flags), oracle.forward_f32(a, mode, param))
    ties = np.tile(np.load(__import__("os").path.join(__import__("conftest").GOLDEN, "case_ties128.npz"))["pre"], (1, 4))
    assert np.array_equal(gpu.forward_fused(ties.astype(np.float32), "qtable", flags_extra=flags),
                          oracle.forward_f32(ties.astype(np.float32), "qtable"))


@pytest.mark.parametrize("shape", [(8, 8), (8, 24), (16, 72), (64, 520), (24, 4104), (1032, 1000)])
def test_float64_kernel_eight_lanes_per_block(gpu, shape):
    """k_forward_fused_f64x8 (what fine quantisers get: 'none', small divisors, wide 'discard' windows), forced for
    every quantiser: block counts that are not multiples of 8 or 16, pitch > W, signed and fractional samples --
    equal to the oracle and to the lane-per-block float64 kernel it replaces."""
    h, w = shape
    rng = np.random.default_rng(h * 10007 + w)
    planes = [gpu.synth.generate_plane("noise", h, w, seed=5), rng.normal(0, 500, (h, w)).astype(np.float32),
              (rng.integers(0, 4096, (h, w)) / 16.0).astype(np.float32)]
    for a in planes:
        for mode, param in (("none", 0.0), ("qtable", 0.0), ("divide", 1.5), ("divide", -0.75), ("discard", 5.0), ("discard", 0.0),
                            ("divide", 2.0), ("divide", -4.0), ("divide", 8.0), ("divide", 2.0 ** 60)):
            want = oracle.forward_f32(a, mode, param)
            got = gpu.forward_fused(a, mode, param, flags_extra=gpu.F_TUNE_F64_KERNEL)
            assert np.array_equal(got, want), (shape, mode, param)
            assert np.array_equal(gpu.forward_fused(a, mode, param, flags_extra=gpu.F_TUNE_F64_KERNEL | gpu.F_TUNE_F64_LANE_PER_BLOCK), want)
    # amplitudes beyond int16 saturate (the reference keeps float64 there and fails later, in its entropy stage)
    a = planes[0]
    for d in (2.0 ** -40, 0.01):
        sat = gpu.forward_fused(a, "divide", d, flags_extra=gpu.F_TUNE_F64_KERNEL)
        assert np.array_equal(sat, gpu.forward_fused(a, "divide", d, flags_extra=gpu.F_TUNE_F64_KERNEL | gpu.F_TUNE_F64_LANE_PER_BLOCK))
        exact = oracle.zigzag_plane(np.rint(oracle.dct_plane(a.astype(np.float64)) / d))
        assert np.array_equal(sat, np.clip(exact, -32768, 32767).astype(np.int16))
    # the same kernel on float64 planes (means of block_size 3: multiples of 1/9 are not exact in fp32)
    thirds = oracle.mean_pool(rng.integers(0, 256, (3 * h, 3 * w)).astype(np.float64), 3)
    for mode, param in (("qtable", 0.0), ("none", 0.0), ("divide", 2.5), ("discard", 3.0)):
        want = oracle.zigzag_plane(oracle.quant_plane(oracle.dct_plane(thirds), mode, param)).astype(np.int16)
        assert np.array_equal(gpu.forward_fused_f64(thirds, mode, param), want), (shape, mode)
        assert np.array_equal(gpu.forward_fused_f64(thirds, mode, param, flags_extra=gpu.F_TUNE_F64_LANE_PER_BLOCK), want), (shape, mode)
    # a padded device plane: pitch > W, the padding holds garbage that must not be read into the result
    pitch = w + 24
    wide = rng.integers(0, 256, (h, pitch)).astype(np.float32)
    din, dout = gpu.DeviceBuffer(wide.nbytes), gpu.DeviceBuffer((h // 8) * (w // 8) * 128)
    din.upload(wide)
    gpu.forward_fused_device(din.ptr, h, w, dout.ptr, "none", 0.0, gpu.F_PIXEL_INPUT | gpu.F_TUNE_F64_KERNEL, pitch=pitch)
    gpu.check(gpu.lib().jpegx_device_synchronize())
    assert np.array_equal(dout.download((h // 8, w // 8, 64), np.int16), oracle.forward_f32(wide[:, :w], "none"))


@pytest.mark.parametrize("shape", [(8, 8), (64, 520), (24, 4104), (1032, 1000)])
def test_xcd_contiguous_block_order_is_only_an_order(gpu, shape):
    """JPEGX_F_TUNE_XCD_CONTIG (the order launches of 2^22 blocks and more use by default: the XCDs take turns in
    runs of strips) on block counts that are not multiples of a run: forward and inverse results are unchanged."""
    h, w = shape
    a = gpu.synth.generate_plane("noise", h, w, seed=31)
    want = oracle.forward_f32(a, "qtable")
    dzz, dout = gpu.DeviceBuffer(want.nbytes), gpu.DeviceBuffer(h * w * 4)
    dzz.upload(want)
    for run in (0, 1, 3, 31):                         # default run (2^7 strips), 2, 8, one run per XCD
        flags = gpu.F_TUNE_XCD_CONTIG | gpu.F_TUNE_XCD_RUN(run)
        assert np.array_equal(gpu.forward_fused(a, "qtable", flags_extra=flags), want), run
        for ot, dt in ((gpu.OUT_F32, np.float32), (gpu.OUT_I16, np.int16)):
            gpu.inverse_fused_device(dzz.ptr, h, w, dout.ptr, "qtable", 0.0, flags, out_type=ot)
            gpu.check(gpu.lib().jpegx_device_synchronize())
            assert np.array_equal(dout.download((h, w), dt).astype(np.int32), oracle.inverse_i16(want, "qtable")), run


@pytest.mark.parametrize("kind", ["noise", "smooth"])
def test_column_wise_exact_tier_is_bit_exact(gpu, golden, kind):
    """k_forward_fused_strip_cols (the default once many blocks are expected to need the float64 tier, forced
    here for every quantiser): identical to the oracle on planes, on the tie-stress fixture, with and without the
    pixel promise, on ragged block counts and on blocks where several columns are flagged at once."""
    F = gpu.F_TUNE_COLUMN_UNITS
    a = gpu.synth.generate_plane(kind, 1024, 1024, seed=19)
    for mode, param in (("qtable", 0.0), ("divide", 7.0), ("divide", 3.0), ("divide", 1.5), ("none", 0.0), ("discard", 4.0),
                        ("divide", -2.5), ("divide", 40.0)):
        want = oracle.forward_f32(a, mode, param)
        for pixel in (True, False):
            got = gpu.forward_fused(a, mode, param, pixel_input=pixel, flags_extra=F | 0x8000)     # 0x8000: never the f64 kernel
            assert np.array_equal(got, want), (mode, param, pixel)
    ties = golden("ties128")
    assert np.array_equal(gpu.forward_fused(ties["pre"].astype(np.float32), "qtable", flags_extra=F), ties["zz_qtable"])
    assert np.array_equal(gpu.forward_fused(ties["pre"].astype(np.float32), "none", flags_extra=F | 0x8000), ties["zz_none"])
    rng = np.random.default_rng(3)
    for h, w in ((8, 8), (24, 520), (64, 4104)):
        b = (rng.integers(0, 1024, (h, w)) / 4.0).astype(np.float32)
        assert np.array_equal(gpu.forward_fused(b, "divide", 2.0, flags_extra=F | 0x8000), oracle.forward_f32(b, "divide", 2.0)), (h, w)
    # and it is what fine quantisers get by default
    assert np.array_equal(gpu.forward_fused(a, "divide", 7.0), oracle.forward_f32(a, "divide", 7.0))


def test_device_api_with_pitch_stream_and_events(gpu):
    """Device-pointer entry points: padded rows (pitch > W), a non-default stream, HIP events."""
    import ctypes
    L = gpu.lib()
    h, w, pitch = 64, 520, 544                      # W/8 = 65 blocks per row: ragged strips too
    a = gpu.synth.generate_plane("noise", h, w, seed=77)
    padded = np.full((h, pitch), -1.0, dtype=np.float32)
    padded[:, :w] = a
    st = ctypes.c_void_p()
    gpu.check(L.jpegx_stream_create(ctypes.byref(st)))
    din, dzz, dback = gpu.DeviceBuffer(padded.nbytes), gpu.DeviceBuffer(h * w * 2), gpu.DeviceBuffer(h * pitch * 4)
    din.upload(padded, stream=st.value)
    e0, e1 = gpu.Event(), gpu.Event()
    e0.record(st.value)
    gpu.forward_fused_device(din.ptr, h, w, dzz.ptr, "qtable", 0.0, gpu.F_PIXEL_INPUT, pitch=pitch, stream=st.value)
    gpu.check(L.jpegx_memset(dback.ptr, 0, h * pitch * 4, st.value))
    gpu.inverse_fused_device(dzz.ptr, h, w, dback.ptr, "qtable", 0.0, 0, out_type=gpu.OUT_F32, out_pitch=pitch,
                             stream=st.value)
    e1.record(st.value)
    e1.synchronize()
    assert e0.elapsed_ms(e1) > 0.0
    zz = dzz.download((h // 8, w // 8, 64), np.int16, stream=st.value)
    want = oracle.forward_f32(a, "qtable")
    assert np.array_equal(zz, want)
    back = dback.download((h, pitch), np.float32, stream=st.value)
    assert np.array_equal(back[:, :w].astype(np.int32), oracle.inverse_i16(want, "qtable"))
    assert np.all(back[:, w:] == 0)                 # the padding columns are never written
    gpu.check(L.jpegx_stream_destroy(st.value))
    # misaligned pitch is rejected, not mis-executed
    with pytest.raises(gpu.JpegxError):
        gpu.forward_fused_device(din.ptr, h, w, dzz.ptr, "qtable", 0.0, gpu.F_PIXEL_INPUT, pitch=pitch + 1)


def test_exact_tier_census_counters(gpu):
    """jpegx_set_debug_counters reports how many blocks took the float64 tier (3-4 % on noise)."""
    import ctypes
    L = gpu.lib()
    h = w = 1024
    din, dzz, cnt = gpu.DeviceBuffer(h * w * 4), gpu.DeviceBuffer(h * w * 2), gpu.DeviceBuffer(16)
    gpu.generate_plane_device(din.ptr, h, w, "noise", seed=1)
    gpu.check(L.jpegx_memset(cnt.ptr, 0, 16, None))
    gpu.check(L.jpegx_set_debug_counters(cnt.ptr))
    try:
        gpu.forward_fused_device(din.ptr, h, w, dzz.ptr, "qtable", 0.0, gpu.F_PIXEL_INPUT)
        gpu.check(L.jpegx_device_synchronize())
    finally:
        gpu.check(L.jpegx_set_debug_counters(None))
    flagged, total = cnt.download((2,), np.uint64)
    assert total == (h // 8) * (w // 8)
    assert 0.01 < flagged / total < 0.08


@pytest.mark.parametrize("bs", [1, 2, 4])
def test_inverse_u8_with_fused_inflate(gpu, bs):
    """Clamp + SubSampling.invert (util.inflate) fused into the inverse: equals np.repeat of the oracle."""
    for h, w in ((64, 512), (24, 40)):
        a = gpu.synth.generate_plane("noise", h, w, seed=9 + bs)
        zz = oracle.forward_f32(a, "qtable")
        want = np.clip(oracle.inverse_i16(zz, "qtable"), 0, 255).astype(np.uint8)
        want = np.repeat(np.repeat(want, bs, axis=0), bs, axis=1)
        got = gpu.inverse_fused_u8(zz, "qtable", inflate=bs)
        assert got.shape == want.shape and np.array_equal(got, want)


def test_config3_full_size_ycbcr420(gpu):
    """BASELINE config 3 at full size: 8192x8192 Y + two 2x2-pooled chroma planes, bit-exact vs the oracle."""
    n = 8192
    y = gpu.synth.generate_plane("smooth", n, n, seed=0, plane=0)
    assert np.array_equal(gpu.forward_fused(y, "qtable"), oracle.forward_f32(y, "qtable"))
    del y
    for plane in (1, 2):
        c = gpu.synth.generate_plane("smooth", n, n, seed=0, plane=plane)
        pooled = c.reshape(n // 2, 2, n // 2, 2).sum(axis=(1, 3), dtype=np.float32) * np.float32(0.25)
        assert np.array_equal(gpu.forward_fused_pooled(c, 2, "qtable"), oracle.forward_f32(pooled, "qtable"))


def test_batched_launch_equals_per_plane_launches(gpu):
    """Size-independent property used at config-5 scale: a stack of planes processed as ONE tall plane
    gives exactly the concatenation of the per-plane streams (blocks never interact)."""
    import hashlib
    n, planes = 1024, 6
    stack = np.concatenate([gpu.synth.generate_plane("noise", n, n, seed=2, plane=p) for p in range(planes)], axis=0)
    whole = gpu.forward_fused(stack, "qtable")
    digests = []
    for p in range(planes):
        one = gpu.forward_fused(stack[p * n:(p + 1) * n], "qtable")
        assert np.array_equal(one, whole[p * (n // 8):(p + 1) * (n // 8)])
        digests.append(hashlib.sha256(one.tobytes()).digest())
    assert hashlib.sha256(b"".join(digests)).hexdigest() == hashlib.sha256(
        b"".join(hashlib.sha256(whole[p * (n // 8):(p + 1) * (n // 8)].tobytes()).digest() for p in range(planes))).hexdigest()
    # and the round trip of the stack equals the oracle's for one plane
    rec = gpu.inverse_fused(whole, "qtable", out="u8")
    want = np.clip(oracle.inverse_i16(oracle.forward_f32(stack[:n], "qtable"), "qtable"), 0, 255)
    assert np.array_equal(rec[:n], want)


def test_full_scale_batch_in_xcd_private_order_equals_per_plane_launches(gpu):
    """16 planes 4096x4096 stacked into one launch -- 2^22 blocks, the size from which the XCD-private block order
    is the default -- against the same planes transformed one by one (natural order), forward and inverse: a
    size-independent property (blocks never interact) checked at the scale the bench runs at."""
    n, planes = 4096, 16
    L = gpu.lib()
    src, zz, one = gpu.DeviceBuffer(planes * n * n * 4), gpu.DeviceBuffer(planes * n * n * 2), gpu.DeviceBuffer(n * n * 2)
    rec, rec1 = gpu.DeviceBuffer(planes * n * n), gpu.DeviceBuffer(n * n)
    for p in range(planes):
        gpu.generate_plane_device(src.ptr + p * n * n * 4, n, n, "noise", seed=3, plane=p)
    gpu.forward_fused_device(src.ptr, n * planes, n, zz.ptr, "qtable", 0.0, gpu.F_PIXEL_INPUT)
    gpu.check(L.jpegx_inverse_fused(zz.ptr, n * planes, n, 3, 0.0, 0, rec.ptr, n, gpu.OUT_U8, None))
    gpu.check(L.jpegx_device_synchronize())
    for p in (0, 7, 15):
        gpu.forward_fused_device(src.ptr + p * n * n * 4, n, n, one.ptr, "qtable", 0.0, gpu.F_PIXEL_INPUT | gpu.F_TUNE_NO_XCD_CONTIG)
        gpu.check(L.jpegx_inverse_fused(one.ptr, n, n, 3, 0.0, gpu.F_TUNE_NO_XCD_CONTIG, rec1.ptr, n, gpu.OUT_U8, None))
        gpu.check(L.jpegx_device_synchronize())
        assert np.array_equal(zz.download((n * n,), np.int16, offset=p * n * n * 2), one.download((n * n,), np.int16)), p
        assert np.array_equal(rec.download((n * n,), np.uint8, offset=p * n * n), rec1.download((n * n,), np.uint8)), p
    # and plane 0 of the batch against the oracle
    want = oracle.forward_f32(gpu.synth.generate_plane("noise", 512, n, seed=3, plane=0), "qtable")
    assert np.array_equal(zz.download(want.shape, np.int16), want)


def test_random_shapes_modes_and_values_against_oracle(gpu):
    """Seeded fuzz: random plane shapes, quantisers, parameters and value ranges (incl. negative and
    fractional fp32 samples, which take the generic non-pixel variant)."""
    rng = np.random.default_rng(20261004)
    for trial in range(40):
        h, w = 8 * int(rng.integers(1, 20)), 8 * int(rng.integers(1, 90))
        kind = trial % 4
        if kind == 0:
            a = rng.integers(0, 256, (h, w)).astype(np.float32)
        elif kind == 1:
            a = (rng.integers(0, 1024, (h, w)) / 4.0).astype(np.float32)          # quarter steps
        elif kind == 2:
            a = rng.normal(0, 300, (h, w)).astype(np.float32)                     # signed, fractional
        else:
            a = (rng.integers(-2000, 2000, (h, w)) * 0.5).astype(np.float32)
        mode, param = [("qtable", 0.0), ("none", 0.0), ("divide", float(rng.integers(1, 200))),
                       ("discard", float(rng.integers(0, 10)))][int(rng.integers(0, 4))]
        want = oracle.forward_f32(a, mode, param)
        got = gpu.forward_fused(a, mode, param)
        assert np.array_equal(got, want), (trial, h, w, mode, param)
        if np.abs(want).max() < 16384:
            assert gpu.entropy_encode(want) == oracle.rle_bytestream(want)
        back = gpu.inverse_fused(want, mode, param, out="f32")
        assert np.array_equal(back.astype(np.int64), oracle.inverse_i16(want, mode, param)), (trial, mode, param)


@pytest.mark.parametrize("flags", [0, 0x40000, 0x4000, 0x4004, 0x800, 0x200])
def test_non_finite_and_huge_samples_stay_inside_their_block(gpu, flags):
    """NaN, +-Inf and 1e30 in a float plane (the reference would fail on them at its integer conversion): every
    kernel variant returns, and the blocks that do not contain such a sample are what they are without them."""
    clean = gpu.synth.generate_plane("noise", 64, 256, seed=12).astype(np.float32)
    dirty = clean.copy()
    bad = {3: np.nan, 5: np.inf, 7: -np.inf, 9: 1e30, 11: -1e30, 40: 3e38, 77: np.nan}
    for blk, val in bad.items():
        by, bx = divmod(blk, 256 // 8)
        dirty[by * 8 + (blk % 5), bx * 8 + (blk % 7)] = val
    keep = np.ones((64 // 8) * (256 // 8), dtype=bool)
    keep[list(bad)] = False
    for mode, param in (("qtable", 0.0), ("none", 0.0), ("divide", 3.0)):
        want = oracle.forward_f32(clean, mode, param).reshape(-1, 64)
        got = gpu.forward_fused(dirty, mode, param, pixel_input=False, flags_extra=flags).reshape(-1, 64)
        assert np.array_equal(got[keep], want[keep]), (mode, flags)
    got64 = gpu.forward_fused_f64(dirty.astype(np.float64), "qtable").reshape(-1, 64)
    assert np.array_equal(got64[keep], oracle.forward_f32(clean, "qtable").reshape(-1, 64)[keep])


def test_hot_entries_with_an_explicit_device_index(gpu):
    """jpegx_forward_fused_on / jpegx_inverse_fused_on: same results as on the current device, the thread's current
    device is unchanged afterwards, a device that does not exist is an error and not a fault."""
    L = gpu.lib()
    a = gpu.synth.generate_plane("noise", 64, 128, seed=2)
    want = oracle.forward_f32(a, "qtable")
    din, dzz, dout = gpu.DeviceBuffer(a.nbytes), gpu.DeviceBuffer(want.nbytes), gpu.DeviceBuffer(a.nbytes)
    din.upload(a)
    cur = ctypes.c_int(-1)
    gpu.check(L.jpegx_get_device(ctypes.byref(cur)))
    gpu.check(L.jpegx_forward_fused_on(cur.value, din.ptr, 64, 128, 128, gpu.mode_of("qtable"), 0.0, gpu.F_PIXEL_INPUT, dzz.ptr, None))
    gpu.check(L.jpegx_inverse_fused_on(cur.value, dzz.ptr, 64, 128, gpu.mode_of("qtable"), 0.0, 0, dout.ptr, 128, gpu.OUT_F32, None))
    gpu.check(L.jpegx_device_synchronize())
    assert np.array_equal(dzz.download(want.shape, np.int16), want)
    assert np.array_equal(dout.download(a.shape, np.float32).astype(np.int32), oracle.inverse_i16(want, "qtable"))
    after = ctypes.c_int(-1)
    gpu.check(L.jpegx_get_device(ctypes.byref(after)))
    assert after.value == cur.value
    assert L.jpegx_forward_fused_on(4096, din.ptr, 64, 128, 128, gpu.mode_of("qtable"), 0.0, 0, dzz.ptr, None) != 0
    gpu.check(L.jpegx_get_device(ctypes.byref(after)))
    assert after.value == cur.value
    # a refused HIP call is reported once, by the entry that made it: it must not linger as the thread's "last
    # error" and fail the next, perfectly good launch (every launch is followed by a hipGetLastError check)
    assert L.jpegx_set_device(4096) != 0
    assert np.array_equal(gpu.forward_fused(a, "qtable"), want)
    gpu.check(L.jpegx_forward_fused(din.ptr, 64, 128, 128, gpu.mode_of("qtable"), 0.0, gpu.F_PIXEL_INPUT, dzz.ptr, None))
    gpu.check(L.jpegx_device_synchronize())


def test_native_rccl_gather_single_rank(gpu):
    """jpegx_comm_* on a 1-rank communicator: the gather is a send+recv to self through RCCL.  (More
    ranks need more GPUs; the driver's multi-GPU run and tests/test_multigpu_cpu.py cover the sharding.)"""
    from jpegx.multigpu import NativeComm
    a = gpu.synth.generate_plane("smooth", 64, 512, seed=3)
    zz = oracle.forward_f32(a, "qtable")
    src, dst = gpu.DeviceBuffer(zz.nbytes), gpu.DeviceBuffer(zz.nbytes)
    src.upload(zz)
    gpu.check(gpu.lib().jpegx_memset(dst.ptr, 0, zz.nbytes, None))
    comm = NativeComm(1, 0, lambda ident: ident)
    try:
        comm.gather_bytes(src.ptr, zz.nbytes, dst.ptr, [zz.nbytes], root=0)
        gpu.check(gpu.lib().jpegx_device_synchronize())
        assert np.array_equal(dst.download(zz.shape, np.int16), zz)
    finally:
        comm.close()


def test_overlapped_transform_and_gather_loopback(gpu):
    """The multi-GPU driver step (jpegx.multigpu.transform_and_gather) on one GPU: chunked transform on
    one stream, per-chunk events, send/recv of every chunk on a second stream through a 1-rank RCCL
    communicator; what lands in the root buffer is the oracle's stream of every plane, and RCCL
    reports the communicator size it was created with."""
    from jpegx import multigpu
    n, planes, chunk = 256, 5, 2
    L = gpu.lib()
    src, out, root = gpu.DeviceBuffer(planes * n * n * 4), gpu.DeviceBuffer(planes * n * n * 2), gpu.DeviceBuffer(planes * n * n * 2)
    for p in range(planes):
        gpu.generate_plane_device(src.ptr + p * n * n * 4, n, n, "noise", seed=9, plane=p)
    gpu.check(L.jpegx_memset(root.ptr, 0, planes * n * n * 2, None))
    gpu.check(L.jpegx_device_synchronize())
    s1, s2 = ctypes.c_void_p(), ctypes.c_void_p()
    gpu.check(L.jpegx_stream_create(ctypes.byref(s1)))
    gpu.check(L.jpegx_stream_create(ctypes.byref(s2)))
    plan = multigpu.GatherPlan(planes, 1, n * n * 2, chunk)
    events = [gpu.Event() for _ in range(plan.rounds)]
    comm = multigpu.NativeComm(1, 0, lambda ident: ident)
    try:
        assert comm.count() == 1
        multigpu.transform_and_gather(comm, plan, src.ptr, out.ptr, root.ptr, n, "qtable", 0.0, gpu.F_PIXEL_INPUT,
                                      s1.value, s2.value, events, root=0, loopback=True)
        gpu.check(L.jpegx_stream_synchronize(s1.value))
        gpu.check(L.jpegx_stream_synchronize(s2.value))
        got = root.download((planes, n // 8, n // 8, 64), np.int16)
        for p in range(planes):
            assert np.array_equal(got[p], oracle.forward_f32(gpu.synth.generate_plane("noise", n, n, seed=9, plane=p), "qtable"))
    finally:
        comm.close()
        gpu.check(L.jpegx_stream_destroy(s1.value))
        gpu.check(L.jpegx_stream_destroy(s2.value))


def test_gather_bookkeeping_of_a_three_rank_world_on_one_gpu(gpu):
    """transform_and_gather's N > 1 branch (what each rank sends in which round, where the root lands it, the
    root's own planes written in place) with the ranks played one after another on this GPU and a recording
    stand-in for the communicator that performs the matched transfers as device copies afterwards: the root's
    buffer must be the un-sharded batch's stream.  Uneven shards (7 planes over 3 ranks), chunk of 2."""
    from jpegx import multigpu
    n, total, world, chunk = 256, 7, 3, 2
    L = gpu.lib()
    plane_in, plane_out = n * n * 4, n * n * 2
    plan = multigpu.GatherPlan(total, world, plane_out, chunk)
    root_buf = gpu.DeviceBuffer(total * plane_out)
    gpu.check(L.jpegx_memset(root_buf.ptr, 0xEE, total * plane_out, None))
    calls = {}                                           # (rank, round) -> recorded gather_bytes arguments

    class Recorder:
        def __init__(self, rank):
            self.rank, self.nranks, self.round = rank, world, 0

        def gather_bytes(self, send_ptr, send_bytes, recv_ptr=None, recv_bytes=None, recv_offsets=None, root=0, stream=None):
            calls[(self.rank, self.round)] = (send_ptr, int(send_bytes), recv_ptr, list(recv_bytes), list(recv_offsets), root)
            self.round += 1

    keep = []
    for rank in range(world):
        lo, hi = plan.spans[rank]
        src = gpu.DeviceBuffer(max(1, hi - lo) * plane_in)
        for p in range(lo, hi):
            gpu.generate_plane_device(src.ptr + (p - lo) * plane_in, n, n, "noise", seed=4, plane=p)
        if rank == 0:
            stream_ptr = root_buf.ptr + lo * plane_out               # the root's planes are written in place
        else:
            own = gpu.DeviceBuffer(max(1, hi - lo) * plane_out)
            keep.append(own)
            stream_ptr = own.ptr
        keep.append(src)
        events = [gpu.Event() for _ in range(plan.rounds)]
        multigpu.transform_and_gather(Recorder(rank), plan, src.ptr, stream_ptr, root_buf.ptr if rank == 0 else None, n,
                                      "qtable", 0.0, gpu.F_PIXEL_INPUT, None, None, events, root=0)
    gpu.check(L.jpegx_device_synchronize())
    for k in range(plan.rounds):                          # what RCCL would do with each grouped round
        _, root_send, recv_ptr, sizes, offs, _ = calls[(0, k)]
        assert root_send == 0 and sizes[0] == 0           # the root ships nothing to itself
        for r in range(1, world):
            send_ptr, send_bytes = calls[(r, k)][:2]
            assert send_bytes == sizes[r]                 # sender and root agree on the size of the message
            if send_bytes:
                gpu.check(L.jpegx_memcpy_d2d(recv_ptr + offs[r], send_ptr, send_bytes, None))
    gpu.check(L.jpegx_device_synchronize())
    got = root_buf.download((total, n // 8, n // 8, 64), np.int16)
    for p in range(total):
        assert np.array_equal(got[p], oracle.forward_f32(gpu.synth.generate_plane("noise", n, n, seed=4, plane=p), "qtable")), p


@pytest.mark.parametrize("kind", ["noise", "smooth"])
def test_planes_in_one_launch_match_the_oracle(gpu, kind):
    """jpegx_forward_fused_planes (configs[2] layout, reduced size): a bs = 1 plane, two 2x2-pooled planes and
    a 4x4-pooled ragged plane in ONE grid give each plane's oracle stream; descriptor order is free."""
    n = 512
    specs = [(n, n, 1), (n // 2, n // 2, 2), (n // 2, n // 2, 2), (40, 72, 4)]     # (H, W after pooling, bs)
    bufs, outs, want, descs = [], [], [], []
    for i, (h, w, bs) in enumerate(specs):
        full = gpu.synth.generate_plane(kind, h * bs, w * bs, seed=21, plane=i)
        b, o = gpu.DeviceBuffer(full.nbytes), gpu.DeviceBuffer(h * w * 2)
        b.upload(full)
        bufs.append(b)
        outs.append(o)
        pooled = full if bs == 1 else oracle.mean_pool(full.astype(np.float64), bs)
        want.append(oracle.forward_f32(pooled, "qtable"))
        descs.append((b.ptr, h, w, w * bs, bs, o.ptr))
    for order in ([0, 1, 2, 3], [3, 1, 0, 2]):
        for o, (h, w, bs) in zip(outs, specs):
            gpu.check(gpu.lib().jpegx_memset(o.ptr, 0xFF, h * w * 2, None))
        gpu.forward_fused_planes_device([descs[i] for i in order], "qtable", 0.0, gpu.F_PIXEL_INPUT)
        gpu.check(gpu.lib().jpegx_device_synchronize())
        for o, w_, (h, w, bs) in zip(outs, want, specs):
            assert np.array_equal(o.download((h // 8, w // 8, 64), np.int16), w_), (order, h, w, bs)
    with pytest.raises(gpu.JpegxError):
        gpu.forward_fused_planes_device([descs[0]] * 9)
    with pytest.raises(gpu.JpegxError):
        gpu.forward_fused_planes_device([(bufs[0].ptr, n, n, n, 3, outs[0].ptr)])


@pytest.mark.parametrize("kind", ["noise", "smooth"])
def test_inverse_fused_full_size_vs_oracle(gpu, kind):
    """configs[3] at full size: the fused inverse of a whole 4096x4096 plane (f32, clamped, and int16)
    equals the oracle's reconstruction sample for sample."""
    n = 4096
    a = gpu.synth.generate_plane(kind, n, n, seed=13)
    zz = gpu.forward_fused(a, "qtable")
    want = oracle.inverse_i16(zz, "qtable")
    assert np.array_equal(gpu.inverse_fused(zz, "qtable", out="i16").astype(np.int32), want)
    got = gpu.inverse_fused(zz, "qtable", out="f32", clamp=True)
    assert np.array_equal(got, np.clip(want, 0, 255).astype(np.float32))


def test_unusual_quantiser_parameters(gpu):
    """Negative / fractional divisors (np.round(a / float(d)) accepts any float), keep = 0 and keep >= 8."""
    a = gpu.synth.generate_plane("noise", 128, 256, seed=41)
    for mode, param in (("divide", -3.0), ("divide", 2.5), ("divide", -64.0), ("divide", 1.0), ("divide", 1e4),
                        ("discard", 0.0), ("discard", 8.0), ("discard", 11.0), ("discard", 1.0)):
        want = oracle.forward_f32(a, mode, param)
        for pixel in (True, False):
            assert np.array_equal(gpu.forward_fused(a, mode, param, pixel_input=pixel), want), (mode, param, pixel)
        assert np.array_equal(gpu.forward_fused(a, mode, param, flags_extra=0x800), want), (mode, param)
        assert np.array_equal(gpu.inverse_fused(want, mode, param, out="f32").astype(np.int64),
                              oracle.inverse_i16(want, mode, param)), (mode, param)


def test_pooled_and_inflated_with_padded_pitches(gpu):
    """Device entry points with pitch > width for the pooled forward and the inflated u8 inverse."""
    L = gpu.lib()
    h, w, bs = 32, 72, 2
    raw = gpu.synth.generate_plane("noise", h * bs, w * bs, seed=8)
    pitch = w * bs + 16
    padded = np.zeros((h * bs, pitch), np.float32)
    padded[:, :w * bs] = raw
    din, dzz = gpu.DeviceBuffer(padded.nbytes), gpu.DeviceBuffer(h * w * 2)
    din.upload(padded)
    gpu.forward_fused_device(din.ptr, h, w, dzz.ptr, "qtable", 0.0, gpu.F_PIXEL_INPUT, pitch=pitch, pool=bs)
    zz = dzz.download((h // 8, w // 8, 64), np.int16)
    pooled = oracle.mean_pool(raw, bs).astype(np.float32)
    want = oracle.forward_f32(pooled, "qtable")
    assert np.array_equal(zz, want)
    opitch = w * bs + 48
    dout = gpu.DeviceBuffer(h * bs * opitch)
    gpu.check(L.jpegx_memset(dout.ptr, 7, h * bs * opitch, None))
    gpu.check(L.jpegx_inverse_fused_u8_inflated(dzz.ptr, h, w, 3, 0.0, 0, bs, dout.ptr, opitch, None))
    got = dout.download((h * bs, opitch), np.uint8)
    rec = np.clip(oracle.inverse_i16(want, "qtable"), 0, 255).astype(np.uint8)
    assert np.array_equal(got[:, :w * bs], np.repeat(np.repeat(rec, bs, 0), bs, 1))
    assert np.all(got[:, w * bs:] == 7)


@pytest.mark.parametrize("bs", [1, 2, 4])
def test_forward_u8_input(gpu, golden, bs):
    """uint8 planes (with the 2x2 mean fused for bs=2): same integers as the fp32 path and the oracle."""
    for h, w in ((64, 512), (24, 48), (8, 16), (136, 1040)):
        raw = gpu.synth.generate_plane("noise", h * bs, w * bs, seed=3 * bs + h, dtype=np.uint8)
        pre = oracle.mean_pool(raw, bs).astype(np.float32) if bs > 1 else raw.astype(np.float32)
        for mode, param in (("qtable", 0.0), ("none", 0.0), ("divide", 7.0), ("discard", 3.0)):
            want = oracle.forward_f32(pre, mode, param)
            assert np.array_equal(gpu.forward_fused_u8(raw, bs, mode, param), want), (h, w, mode)
            assert np.array_equal(gpu.forward_fused_u8(raw, bs, mode, param, flags_extra=0x100), want)
    if bs < 4:
        c = golden("ties128" if bs == 1 else "pooled128")
        assert np.array_equal(gpu.forward_fused_u8(c["input"].astype(np.uint8), bs, "qtable"), c["zz_qtable"])
        assert gpu.compress_plane(c["input"].astype(np.uint8), bs, "qtable") == oracle.rle_bytestream(c["zz_qtable"])
    if bs == 1:
        with pytest.raises(gpu.JpegxError):
            gpu.forward_fused_u8(np.zeros((8, 8), np.uint8), 1)          # W % 16 != 0
