"""Codec facade and configuration with the reference's names (reference: pipeline/__init__.py).

``compress_band`` / ``decompress_band`` walk the registered steps like the reference, except
that for the accelerated configuration (transform 'DCT', dct_size 8) the three hot steps --
BasisChange, Quantization, ZigzagOrder -- are replaced by ONE fused GPU kernel launch
(libjpegx ``jpegx_forward_fused`` / ``jpegx_inverse_fused``) whose integer output is bit-exact
with the step-by-step float64 pipeline.  There is no CPU fallback for the accelerated
configuration: without libjpegx.so or a GPU the call raises ``jpegx.JpegxError``.
"""
import json

import numpy as np

import file_format
from quantizers import DiscardingQuantizer, DivisionQuantizer, JpegQuantizationTable, RoundingQuantizer
from util import band_to_array
from . import (basis_change, dct_padding, normalization, padding, quantization,  # noqa: F401  (registration)
               rle_byte_stream, run_length_encoding, subsampling, zigzag_order)
from .base import step_classes


class BadQuantizationError(Exception):
    pass


class QuantizationMethod:
    """Named quantiser + its keyword parameters (pipeline/__init__.py:13-47)."""
    name_to_quantizer = {
        "none": RoundingQuantizer,
        "discard": DiscardingQuantizer,
        "divide": DivisionQuantizer,
        "qtable": JpegQuantizationTable,
    }

    def __init__(self, name, **kwargs):
        self.name = name
        self.params = kwargs
        self.quantizer = self._get_quantizer()

    def _get_quantizer(self):
        error_msg = "name {}, params {}".format(self.name, self.params)
        factory = self.name_to_quantizer.get(self.name)
        if factory is None:
            raise BadQuantizationError(error_msg)
        try:
            return factory(**self.params)
        except Exception:
            raise BadQuantizationError(error_msg)

    def to_json(self):
        d = dict(self.params)
        d["quantization_scheme_name"] = self.name
        return json.dumps(d)

    @staticmethod
    def from_json(s):
        d = json.loads(s)
        name = d.pop("quantization_scheme_name")
        return QuantizationMethod(name, **d)

    def gpu_mode(self):
        """(mode name, scalar parameter) understood by libjpegx, or None if not expressible.  Only the STOCK
        quantiser objects qualify: a replaced or subclassed quantiser, or an edited luminance table, is a
        different function and goes through the object's own quantize/restore like in the reference."""
        q = self.quantizer
        if type(q) is not self.name_to_quantizer.get(self.name):
            return None
        if self.name == "discard":
            return ("discard", float(q.keep)) if isinstance(q.keep, (int, np.integer)) and q.keep >= 0 else None
        if self.name == "divide":
            try:
                d = float(q.divisor)
            except (TypeError, ValueError):
                return None
            return ("divide", d) if d != 0 and np.isfinite(d) else None
        if self.name == "qtable":
            stock = np.array(JpegQuantizationTable.table)
            same = np.array_equal(np.asarray(q.table), stock) and np.array_equal(getattr(q, "_qtable", stock), stock)
            return ("qtable", 0.0) if same else None
        return ("none", 0.0) if self.name == "none" else None


class Configuration:
    """pipeline/__init__.py:50-64"""

    def __init__(self, width, height, block_size=2, dct_size=8, transform="DCT", quantization=None):
        self.width = width
        self.height = height
        self.block_size = block_size
        self.dct_size = dct_size
        self.transform = transform
        if quantization is None:
            quantization = QuantizationMethod("none")
        elif quantization.name == "qtable" and dct_size != 8:
            raise BadQuantizationError()
        self.quantization = quantization


def _accelerated(config):
    return config.transform == "DCT" and config.dct_size == 8 and config.quantization.gpu_mode() is not None


def _hot_forward(pre, config):
    """Steps 4+5+6 on the plane that leaves step 3: one fused launch when the samples are exact in
    fp32 (always the case for 8-bit data with block_size 1, 2, 4, ...), else the exact float64 kernels."""
    import jpegx
    mode, param = config.quantization.gpu_mode()
    pre = np.asarray(pre)
    as32 = pre.astype(np.float32)
    if np.array_equal(as32.astype(np.float64), pre.astype(np.float64)):
        return jpegx.forward_fused(as32, mode, param).astype(np.float64)
    pre = pre.astype(np.float64)
    if pre.shape[1] % 2 == 0:
        zz = jpegx.forward_fused_f64(pre, mode, param)          # one all-float64 launch
        if zz.min() > -32768 and zz.max() < 32767:              # int16 saturation would hide larger values (np.abs wraps at -32768)
            return zz.astype(np.float64)
    coeffs = jpegx.quantize_f64(jpegx.dct8x8_f64(pre), mode, param)
    return jpegx.zigzag(coeffs)


def _hot_inverse(zz, config):
    """Steps 6+5+4 inverted: fused launch for int16-range coefficients, float64 kernels otherwise."""
    import jpegx
    mode, param = config.quantization.gpu_mode()
    zz = np.asarray(zz)
    if zz.size and np.abs(zz).max() <= 32767 and np.array_equal(zz, np.rint(zz)) and \
            (mode != "divide" or abs(param) * 32767 < 2 ** 24):
        return jpegx.inverse_fused(zz.astype(np.int16), mode, param, out="f32").astype(int)
    plane = jpegx.restore_f64(jpegx.unzigzag(zz.astype(np.float64)), mode, param)
    return jpegx.idct8x8_f64(plane, do_round=True).astype(int)


_HOT_STEPS = (basis_change.BasisChange, quantization.Quantization, zigzag_order.ZigzagOrder)
_BUILTIN_STEPS = (padding.Padding, subsampling.SubSampling, dct_padding.DCTPadding, normalization.Normalization,
                  basis_change.BasisChange, quantization.Quantization, zigzag_order.ZigzagOrder,
                  run_length_encoding.RunLengthEncoding, rle_byte_stream.RleBytestream)


def _stock_registry():
    """True while nobody has registered extra steps: only then may whole groups of steps be fused."""
    return len(step_classes) == len(_BUILTIN_STEPS) and all(a is b for a, b in zip(step_classes, _BUILTIN_STEPS))


def _hot_run(classes):
    """Index at which the three built-in hot steps stand directly one after another in ``classes`` (only
    then may steps 4+5+6 be one launch: a user step registered between them, or a subclass standing in for
    one of them, must run at its own place in the order, like in the reference), else None."""
    for i in range(len(classes) - 2):
        if all(a is b for a, b in zip(classes[i:i + 3], _HOT_STEPS)):
            return i
    return None


def _bad_rle(exc):
    """libjpegx reports the reference's BadRleCodeError condition (amplitude beyond 15 bits, illegal code in a
    stream) in its error text; callers of the pipeline API get the reference's exception type."""
    import util
    if "BadRleCodeError" in str(exc):
        return util.BadRleCodeError(str(exc))
    if "ValueError" in str(exc):                    # what the reference raises for these streams (int('', 2) / reshape)
        return ValueError(str(exc))
    return exc


def _front_end_fused(band, config, with_entropy):
    """Steps 0-6 (or 0-8 when with_entropy) on the GPU when the band needs no DCT padding: Padding
    on the host (edge replication to a multiple of block_size), then SubSampling + BasisChange +
    Quantization + ZigzagOrder inside jpegx_forward_fused_pooled, then optionally
    RunLengthEncoding + RleBytestream (jpegx_entropy_*) with the coefficients never leaving the
    device.  Returns None when not applicable."""
    bs = config.block_size
    band = np.asarray(band)
    if not 1 <= bs <= 255 or band.ndim != 2 or band.size == 0 or band.dtype.kind not in "ui":
        return None
    import jpegx
    mode, param = config.quantization.gpu_mode()
    padded = band if bs == 1 else padding.Padding(config).execute(band)
    if with_entropy and not (padded.shape[0] % (8 * bs) or padded.shape[1] % (8 * bs)):
        # the common case in one native call: range check, upload, steps 1+4..8, bytes back
        blob = jpegx.compress_plane_native(np.ascontiguousarray(padded), bs, mode, param)
        if blob is not None:
            return blob
    if bs not in (1, 2, 4) or (band.dtype != np.uint8 and (band.min() < 0 or band.max() > 255)):
        return None
    if padded.shape[0] % (8 * bs) or padded.shape[1] % (8 * bs):
        # DCT padding is needed: it replicates POOLED edge samples (dct_padding.py:8-9), so pool on
        # the host first (steps 1-3), then the device does steps 4-8 on the padded plane
        pre = padded
        for cls in (subsampling.SubSampling, dct_padding.DCTPadding, normalization.Normalization):
            pre = cls(config).execute(pre)
        if not np.array_equal(pre.astype(np.float32).astype(np.float64), pre):
            return None
        plane = pre.astype(np.uint8) if np.array_equal(pre, np.rint(pre)) else pre.astype(np.float32)
        if with_entropy:
            return jpegx.compress_plane(plane, 1, mode, param)
        return jpegx.forward_fused(plane.astype(np.float32), mode, param, pixel_input=True).astype(np.float64)
    if with_entropy:
        return jpegx.compress_plane(padded, bs, mode, param)      # uint8 upload when the shape allows
    return jpegx.forward_fused_pooled(padded.astype(np.float32), bs, mode, param, pixel_input=True).astype(np.float64)


def _back_end_fused(zz, config):
    """Steps 6-0 inverted in one launch: un-zigzag, dequantise, IDCT, round, clamp to [0, 255]
    (Normalization.invert), replicate block_size x block_size (SubSampling.invert) on the GPU,
    then the two crops (DCTPadding.invert, Padding.invert) as one slice."""
    bs = config.block_size
    zz = np.asarray(zz)
    mode, param = config.quantization.gpu_mode()
    if not 1 <= bs <= 255 or zz.ndim != 3 or zz.shape[2] != 64 or zz.size == 0:
        return None
    if mode == "divide" and abs(param) * 32767 >= 2 ** 24:
        return None
    if zz.dtype != np.int16:                  # the C++ entropy decoder hands over int16; anything else is checked
        if np.abs(zz).max() > 32767 or not np.array_equal(zz, np.rint(zz)):
            return None
        zz = zz.astype(np.int16)
    import jpegx
    full = jpegx.inverse_fused_u8(zz, mode, param, inflate=bs)
    return full[:config.height, :config.width].astype(int)


def compress_band(a, config):
    """Run every registered step forward (pipeline/__init__.py:71-76)."""
    import jpegx
    fused = _accelerated(config)
    todo = list(step_classes)
    try:
        if fused and _stock_registry():
            blob = _front_end_fused(a, config, with_entropy=True)
            if blob is not None:
                return blob                               # all nine steps on the device
        at = _hot_run(todo) if fused else None
        for k, cls in enumerate(todo):
            if at is not None and at < k <= at + 2:
                continue                                   # folded into the fused launch at `at`
            if at is not None and k == at:
                a = _hot_forward(a, config)
                continue
            a = cls(config).execute(a)
        return a
    except jpegx.JpegxError as exc:
        raise _bad_rle(exc)


def decompress_band_u8(compression_result, config):
    """decompress_band for callers that want the displayable uint8 samples (what Jpeg.decompress turns the
    band into anyway, pipeline/__init__.py:119-122): skips the int64 array the reference's API returns --
    for a 4096x4096 band that conversion alone costs more than the whole device pipeline."""
    import jpegx
    a = compression_result
    if _accelerated(config) and _stock_registry() and isinstance(a, (bytes, bytearray)) and len(a) \
            and 1 <= config.block_size <= 255:
        mode, param = config.quantization.gpu_mode()
        if not (mode == "divide" and abs(param) * 32767 >= 2 ** 24):
            rle = run_length_encoding.RunLengthEncoding(config)
            hb, wb = rle._height_in_blocks(), rle._width_in_blocks()
            try:
                return jpegx.decompress_plane(a, hb * 8, wb * 8, config.block_size, mode, param)[:config.height, :config.width]
            except jpegx.JpegxError:
                pass
    return decompress_band(compression_result, config).astype(np.uint8)


def decompress_band(compression_result, config):
    """Run every registered step backwards (pipeline/__init__.py:79-88)."""
    import jpegx
    a = compression_result
    fused = _accelerated(config)
    todo = list(reversed(step_classes))
    try:
        if fused and _stock_registry():
            if isinstance(a, (bytes, bytearray)):
                rle = run_length_encoding.RunLengthEncoding(config)
                hb, wb = rle._height_in_blocks(), rle._width_in_blocks()
                mode, param = config.quantization.gpu_mode()
                if 1 <= config.block_size <= 255 and len(a) and not (mode == "divide" and abs(param) * 32767 >= 2 ** 24):
                    # all nine steps inverted on the device, entropy decoding included; only the samples come back
                    try:
                        band = jpegx.decompress_plane_i64(a, hb * 8, wb * 8, config.block_size, mode, param,
                                                          config.height, config.width)
                        return band if band.dtype == np.dtype(int) else band.astype(int)
                    except jpegx.JpegxError:
                        pass        # not a well-formed stream: the host parser below says exactly what is wrong
                # entropy stage inverted on the host by libjpegx's C++ parser (steps 8, 7)
                a = jpegx.entropy_decode(a, hb * wb).reshape(hb, wb, 64)
            else:
                for cls in todo[:2]:
                    a = cls(config).invert(a)
            band = _back_end_fused(a, config)
            if band is not None:
                return band
            todo = todo[2:]
        at = _hot_run(list(reversed(todo))) if fused else None      # position counted in forward order
        at = None if at is None else len(todo) - 3 - at             # -> index of ZigzagOrder in the reversed list
        for k, cls in enumerate(todo):
            if at is not None and at < k <= at + 2:
                continue
            if at is not None and k == at:
                a = _hot_inverse(a, config)
                continue
            a = cls(config).invert(a)
        return a
    except jpegx.JpegxError as exc:
        raise _bad_rle(exc)


class CompressedData:
    def __init__(self, y, cb, cr):
        self.y = y
        self.cb = cb
        self.cr = cr


class Jpeg:
    """PIL image <-> container bytes (pipeline/__init__.py:98-124)."""

    def __init__(self, config):
        self.config = config

    def compress(self, image):
        whole = _compress_pixels(image, self.config)         # the picture's pixels as ONE array, the bands made on the device
        if whole is not None:
            return whole
        arrays = [band_to_array(band) for band in image.split()]
        whole = _compress_image(arrays, self.config)        # the finished container, written band by band from the device
        if whole is not None:
            return whole
        bands = [compress_band(a, self.config) for a in arrays]
        return file_format.generate_data(self.config, CompressedData(*bands))

    @staticmethod
    def decompress(bytestream):
        from PIL import Image
        config, data = file_format.read_data(bytestream)
        size = (config.height, config.width)
        packed = _decompress_image((data.y, data.cb, data.cr), config)
        if packed is None:
            packed = np.dstack([decompress_band_u8(b, config).reshape(size) for b in (data.y, data.cb, data.cr)])
        return Image.fromarray(packed, mode="YCbCr")


def _compress_pixels(image, config):
    """Jpeg.compress without `image.split()`: for a 4096 x 4096 picture PIL needs 25 ms to split the bands and hand each
    over as an array, 14 ms to hand over the interleaved pixels in one piece (np.asarray(image)) -- and the native job
    behind either takes 1.5 ms.  Multi-band 8-bit pictures whose size needs no padding take this road; the bytes are
    those of the per-band road.  None otherwise."""
    import jpegx
    if not (_accelerated(config) and _stock_registry()):
        return None
    bs = config.block_size
    try:
        bands = image.getbands()
    except Exception:
        return None
    if not 1 <= bs <= 255 or len(bands) != 3 or image.mode not in ("YCbCr", "RGB", "LAB", "HSV"):      # CompressedData holds three bands
        return None
    if image.height % (8 * bs) or image.width % (8 * bs) or (config.height, config.width) != (image.height, image.width):
        return None
    pixels = np.asarray(image)
    if pixels.dtype != np.uint8 or pixels.ndim != 3:
        return None
    mode, param = config.quantization.gpu_mode()
    try:
        return jpegx.compress_image_packed(np.ascontiguousarray(pixels), bs, mode, param, prefix=file_format.create_header(config))
    except jpegx.JpegxError as exc:
        raise _bad_rle(exc)


def _compress_image(arrays, config):
    """The three bands of one picture through ONE native job (jpegx_host_compress_image) that writes the finished
    container: the bytes of file_format.generate_data over three compress_band calls (pipeline/__init__.py:102-110,
    file_format.py:86-93), with the bands alternating between two streams and no concatenation on the host.
    None when the configuration or the bands do not take that road."""
    import jpegx
    if not (_accelerated(config) and _stock_registry()):
        return None
    bs = config.block_size
    if not 1 <= bs <= 255 or any(a.ndim != 2 or a.size == 0 or a.dtype.kind not in "ui" for a in arrays):
        return None
    mode, param = config.quantization.gpu_mode()
    padded = [np.ascontiguousarray(a if bs == 1 else padding.Padding(config).execute(a)) for a in arrays]
    if any(p.shape[0] % (8 * bs) or p.shape[1] % (8 * bs) for p in padded):
        return None                                         # DCT padding needed: the per-band road pools on the host first
    try:
        return jpegx.compress_image_native(padded, bs, mode, param, prefix=file_format.create_header(config))
    except jpegx.JpegxError as exc:
        raise _bad_rle(exc)


def _decompress_image(blobs, config):
    """Jpeg.decompress's three decompress_band calls + np.dstack (pipeline/__init__.py:112-124) as ONE native job;
    (height, width, 3) uint8, or None when this road does not apply or the device decoder refuses a stream (the
    per-band road then names the fault)."""
    import jpegx
    if not (_accelerated(config) and _stock_registry() and 1 <= config.block_size <= 255):
        return None
    if any(not isinstance(b, (bytes, bytearray)) or not len(b) for b in blobs):
        return None
    mode, param = config.quantization.gpu_mode()
    if mode == "divide" and abs(param) * 32767 >= 2 ** 24:
        return None
    rle = run_length_encoding.RunLengthEncoding(config)
    hb, wb = rle._height_in_blocks(), rle._width_in_blocks()
    try:
        return jpegx.decompress_image_native(blobs, hb * 8, wb * 8, config.block_size, mode, param, config.height, config.width)
    except jpegx.JpegxError:
        return None
