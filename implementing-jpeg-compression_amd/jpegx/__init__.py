"""ctypes binding of libjpegx.so (C ABI: include/jpegx.h) -- the only road from the Python
host code to the gfx950 kernels.  No PyTorch, no fallback: if the shared library is missing
or no GPU is usable every entry point raises :class:`JpegxError`.

Layers in this module
  * ``lib()``            the raw ``ctypes.CDLL`` with argtypes set for every symbol of jpegx.h;
  * ``Device*`` helpers  thin RAII wrappers for device buffers, streams and events;
  * host conveniences    NumPy in / NumPy out wrappers of the ``jpegx_host_*`` entry points,
                         used by the reference-compatible step classes in ``pipeline/``.
"""
import ctypes
import os

import numpy as np

from . import synth  # noqa: F401  (re-export: jpegx.synth.generate_plane)

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# JPEGX_LIB_PATH: load another build of the same library (A/B measurements of compiler flags)
LIB_PATH = os.environ.get("JPEGX_LIB_PATH") or os.path.join(_PKG_ROOT, "libjpegx.so")

Q_NONE, Q_DISCARD, Q_DIVIDE, Q_QTABLE = 0, 1, 2, 3
MODE_BY_NAME = {"none": Q_NONE, "discard": Q_DISCARD, "divide": Q_DIVIDE, "qtable": Q_QTABLE}
F_PIXEL_INPUT = 1
F_CLAMP_U8 = 2
F_TUNE_F64_LANE_PER_BLOCK = 0x4
F_TUNE_DIRECT_STORE = 0x8
F_TUNE_F64_KERNEL = 0x4000
F_TUNE_NO_F64_KERNEL = 0x8000
F_TUNE_NO_NT = 0x100
F_TUNE_NO_STRIP = 0x200
F_TUNE_SKIP_EXACT = 0x400
F_TUNE_WAVE_PER_BLOCK = 0x800
F_TUNE_XCD_CONTIG = 0x10000
F_TUNE_NO_XCD_CONTIG = 0x20000
F_TUNE_COLUMN_UNITS = 0x40000
F_TUNE_NO_COLUMN_UNITS = 0x80000


def F_TUNE_XCD_RUN(logr):
    """JPEGX_F_TUNE_XCD_RUN: run length 2^logr strips of the XCD-private order (31 = one run per XCD)."""
    return (int(logr) & 31) << 20


OUT_F32, OUT_I16, OUT_U8 = 0, 1, 2
_OUT_DTYPES = {OUT_F32: np.float32, OUT_I16: np.int16, OUT_U8: np.uint8}
_OUT_BY_NAME = {"f32": OUT_F32, "i16": OUT_I16, "u8": OUT_U8}


class JpegxError(RuntimeError):
    """A libjpegx call failed (or the library / GPU is missing)."""


_lib = None

_c = ctypes
_vp, _int, _dbl, _uint, _sz, _pd = _c.c_void_p, _c.c_int, _c.c_double, _c.c_uint, _c.c_size_t, _c.c_ssize_t
_u32 = _c.c_uint32

# name -> argtypes; every symbol declared in include/jpegx.h (tests/test_abi.py checks the set)
SIGNATURES = {
    "jpegx_init": [_int],
    "jpegx_shutdown": [],
    "jpegx_version": [],
    "jpegx_device_count": [_c.POINTER(_int)],
    "jpegx_set_device": [_int],
    "jpegx_get_device": [_c.POINTER(_int)],
    "jpegx_device_name": [_int, _c.c_char_p, _sz],
    "jpegx_device_synchronize": [],
    "jpegx_malloc": [_c.POINTER(_vp), _sz],
    "jpegx_free": [_vp],
    "jpegx_memset": [_vp, _int, _sz, _vp],
    "jpegx_memcpy_h2d": [_vp, _vp, _sz, _vp],
    "jpegx_memcpy_d2h": [_vp, _vp, _sz, _vp],
    "jpegx_memcpy_d2d": [_vp, _vp, _sz, _vp],
    "jpegx_stream_create": [_c.POINTER(_vp)],
    "jpegx_stream_destroy": [_vp],
    "jpegx_stream_synchronize": [_vp],
    "jpegx_event_create": [_c.POINTER(_vp)],
    "jpegx_event_destroy": [_vp],
    "jpegx_event_record": [_vp, _vp],
    "jpegx_event_synchronize": [_vp],
    "jpegx_stream_wait_event": [_vp, _vp],
    "jpegx_event_elapsed_ms": [_vp, _vp, _c.POINTER(_c.c_float)],
    "jpegx_generate_plane": [_vp, _int, _int, _pd, _int, _u32, _u32, _int, _vp],
    "jpegx_forward_fused": [_vp, _int, _int, _pd, _int, _dbl, _uint, _vp, _vp],
    "jpegx_forward_fused_on": [_int, _vp, _int, _int, _pd, _int, _dbl, _uint, _vp, _vp],
    "jpegx_forward_fused_pooled": [_vp, _int, _int, _pd, _int, _int, _dbl, _uint, _vp, _vp],
    "jpegx_forward_fused_u8": [_vp, _int, _int, _pd, _int, _int, _dbl, _uint, _vp, _vp],
    "jpegx_forward_fused_planes": [_vp, _int, _int, _dbl, _uint, _vp],
    "jpegx_forward_fused_f64": [_vp, _int, _int, _pd, _int, _dbl, _uint, _vp, _vp],
    "jpegx_mean_pool_f64": [_vp, _int, _int, _int, _pd, _int, _vp, _pd, _vp],
    "jpegx_inverse_fused": [_vp, _int, _int, _int, _dbl, _uint, _vp, _pd, _int, _vp],
    "jpegx_inverse_fused_on": [_int, _vp, _int, _int, _int, _dbl, _uint, _vp, _pd, _int, _vp],
    "jpegx_inverse_fused_u8_inflated": [_vp, _int, _int, _int, _dbl, _uint, _int, _vp, _pd, _vp],
    "jpegx_dct8x8_f32": [_vp, _int, _int, _pd, _vp, _pd, _vp],
    "jpegx_idct8x8_f32": [_vp, _int, _int, _pd, _vp, _pd, _vp],
    "jpegx_dct8x8_f64": [_vp, _int, _int, _pd, _vp, _pd, _vp],
    "jpegx_idct8x8_f64": [_vp, _int, _int, _pd, _vp, _pd, _int, _vp],
    "jpegx_quantize_f64": [_vp, _int, _int, _pd, _int, _dbl, _vp, _pd, _vp],
    "jpegx_restore_f64": [_vp, _int, _int, _pd, _int, _dbl, _vp, _pd, _vp],
    "jpegx_zigzag": [_vp, _int, _int, _pd, _int, _vp, _vp],
    "jpegx_unzigzag": [_vp, _int, _int, _int, _vp, _pd, _vp],
    "jpegx_host_forward_fused": [_vp, _int, _int, _pd, _int, _dbl, _uint, _vp],
    "jpegx_host_forward_fused_f64": [_vp, _int, _int, _int, _dbl, _uint, _vp],
    "jpegx_host_inverse_fused": [_vp, _int, _int, _int, _dbl, _uint, _vp, _pd, _int],
    "jpegx_host_inverse_fused_u8_inflated": [_vp, _int, _int, _int, _dbl, _uint, _int, _vp, _pd],
    "jpegx_host_dct8x8_f64": [_vp, _int, _int, _vp],
    "jpegx_host_idct8x8_f64": [_vp, _int, _int, _vp, _int],
    "jpegx_host_quantize_f64": [_vp, _int, _int, _int, _dbl, _vp],
    "jpegx_host_restore_f64": [_vp, _int, _int, _int, _dbl, _vp],
    "jpegx_host_zigzag": [_vp, _int, _int, _int, _vp],
    "jpegx_host_unzigzag": [_vp, _int, _int, _int, _vp],
    "jpegx_host_dct8x8_f32": [_vp, _int, _int, _vp],
    "jpegx_host_idct8x8_f32": [_vp, _int, _int, _vp],
    "jpegx_set_debug_counters": [_vp],
    "jpegx_host_compress_begin": [_vp, _int, _int, _int, _pd, _int, _int, _dbl, _c.POINTER(_sz)],
    "jpegx_host_compress_finish": [_vp],
    "jpegx_host_compress_abort": [],
    "jpegx_host_pool_release": [],
    "jpegx_host_decompress_plane": [_vp, _sz, _int, _int, _int, _int, _dbl, _vp, _pd],
    "jpegx_host_decompress_plane_i64": [_vp, _sz, _int, _int, _int, _int, _dbl, _vp, _int, _int],
    "jpegx_host_entropy_decode_gpu": [_vp, _sz, _c.c_longlong, _vp],
    "jpegx_host_compress_image": [_vp, _int, _int, _int, _int, _pd, _int, _int, _dbl, _vp, _sz, _int, _vp, _vp, _c.POINTER(_sz)],
    "jpegx_host_decompress_image": [_vp, _c.POINTER(_sz), _int, _int, _int, _int, _int, _dbl, _vp, _pd, _int, _int, _int],
    "jpegx_host_compress_image_packed": [_vp, _int, _int, _int, _pd, _int, _int, _dbl, _vp, _sz, _int, _vp, _vp, _c.POINTER(_sz)],
    "jpegx_interleave_u8": [_vp, _int, _int, _int, _pd, _vp, _pd, _vp],
    "jpegx_deinterleave_u8": [_vp, _pd, _int, _int, _int, _vp, _pd, _vp],
    "jpegx_entropy_workspace_bytes": [_c.c_longlong],
    "jpegx_entropy_sizes": [_vp, _c.c_longlong, _vp, _vp],
    "jpegx_entropy_total": [_vp, _c.POINTER(_c.c_ulonglong), _vp],
    "jpegx_entropy_block_sizes": [_vp, _c.c_longlong, _vp, _vp],
    "jpegx_entropy_emit": [_vp, _c.c_longlong, _vp, _vp, _vp],
    "jpegx_host_entropy_encode": [_vp, _c.c_longlong, _vp, _sz, _c.POINTER(_sz)],
    "jpegx_host_entropy_decode": [_vp, _sz, _c.c_longlong, _vp],
    "jpegx_entropy_decode_workspace_bytes": [_sz, _c.c_longlong],
    "jpegx_entropy_decode": [_vp, _sz, _c.c_longlong, _vp, _vp, _int, _vp],
    "jpegx_entropy_decode_status": [_vp, _vp],
    "jpegx_comm_available": [],
    "jpegx_comm_unique_id": [_vp],
    "jpegx_comm_create": [_c.POINTER(_vp), _int, _int, _vp],
    "jpegx_comm_create_deadline": [_c.POINTER(_vp), _int, _int, _vp, _dbl],
    "jpegx_comm_destroy": [_vp],
    "jpegx_comm_abort": [_vp],
    "jpegx_comm_count": [_vp, _c.POINTER(_int)],
    "jpegx_comm_gather_bytes": [_vp, _vp, _sz, _vp, _c.POINTER(_sz), _c.POINTER(_sz), _int, _vp],
}
# explicit-device forms (csrc/jpegx_on.cpp): the plain signature behind a leading device index
for _name in ("jpegx_malloc", "jpegx_free", "jpegx_stream_create", "jpegx_generate_plane", "jpegx_forward_fused_pooled",
              "jpegx_forward_fused_u8", "jpegx_forward_fused_f64", "jpegx_forward_fused_planes", "jpegx_mean_pool_f64",
              "jpegx_inverse_fused_u8_inflated", "jpegx_entropy_sizes", "jpegx_entropy_total", "jpegx_entropy_block_sizes",
              "jpegx_entropy_emit", "jpegx_entropy_decode", "jpegx_entropy_decode_status", "jpegx_host_compress_begin", "jpegx_host_compress_image", "jpegx_host_compress_image_packed", "jpegx_host_decompress_plane",
              "jpegx_host_decompress_plane_i64", "jpegx_host_decompress_image", "jpegx_host_entropy_decode_gpu",
              "jpegx_host_pool_release", "jpegx_comm_create_deadline"):
    SIGNATURES[_name + "_on"] = [_int] + SIGNATURES[_name]
RESTYPES = {"jpegx_entropy_workspace_bytes": _sz, "jpegx_entropy_decode_workspace_bytes": _sz}   # everything else returns int


def lib():
    """Load libjpegx.so once; raise JpegxError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise JpegxError(
                "libjpegx.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C implementing-jpeg-compression_amd/csrc` (there is no CPU fallback)"
                % LIB_PATH)
        try:
            L = ctypes.CDLL(LIB_PATH)
        except OSError as exc:  # missing libamdhip64 etc.
            raise JpegxError("cannot load %s: %s" % (LIB_PATH, exc))
        L.jpegx_last_error.restype = ctypes.c_char_p
        L.jpegx_last_error.argtypes = []
        for name, argtypes in SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = argtypes
            fn.restype = RESTYPES.get(name, ctypes.c_int)
        _lib = L
    return _lib


def check(rc, what="jpegx call"):
    if rc != 0:
        raise JpegxError("%s failed (%d): %s" % (what, rc, lib().jpegx_last_error().decode("utf-8", "replace")))


def device_count():
    n = ctypes.c_int(0)
    rc = lib().jpegx_device_count(ctypes.byref(n))
    return n.value if rc == 0 else 0


def require_device():
    if device_count() < 1:
        raise JpegxError("no usable HIP device: " + lib().jpegx_last_error().decode("utf-8", "replace"))


def device_name(device=0):
    buf = ctypes.create_string_buffer(256)
    check(lib().jpegx_device_name(device, buf, 256), "jpegx_device_name")
    return buf.value.decode()


def mode_of(mode):
    if isinstance(mode, str):
        try:
            return MODE_BY_NAME[mode]
        except KeyError:
            raise JpegxError("unknown quantiser %r" % (mode,))
    return int(mode)


# ---------------------------------------------------------------------------------------------
# device-side helpers (bench.py, tests, multi-GPU driver)
# ---------------------------------------------------------------------------------------------
class DeviceBuffer:
    """hipMalloc'ed span owned by this object."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        p = ctypes.c_void_p()
        check(lib().jpegx_malloc(ctypes.byref(p), self.nbytes), "jpegx_malloc")
        self.ptr = p.value

    def free(self):
        if self.ptr:
            lib().jpegx_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def upload(self, array, stream=None, offset=0):
        a = np.ascontiguousarray(array)
        assert offset + a.nbytes <= self.nbytes
        check(lib().jpegx_memcpy_h2d(self.ptr + offset, a.ctypes.data, a.nbytes, stream), "jpegx_memcpy_h2d")
        check(lib().jpegx_stream_synchronize(stream), "jpegx_stream_synchronize")

    def download(self, shape, dtype, stream=None, offset=0):
        out = np.empty(shape, dtype=dtype)
        assert offset + out.nbytes <= self.nbytes
        check(lib().jpegx_memcpy_d2h(out.ctypes.data, self.ptr + offset, out.nbytes, stream), "jpegx_memcpy_d2h")
        check(lib().jpegx_stream_synchronize(stream), "jpegx_stream_synchronize")
        return out


class Event:
    def __init__(self):
        p = ctypes.c_void_p()
        check(lib().jpegx_event_create(ctypes.byref(p)), "jpegx_event_create")
        self.handle = p.value

    def record(self, stream=None):
        check(lib().jpegx_event_record(self.handle, stream), "jpegx_event_record")

    def synchronize(self):
        check(lib().jpegx_event_synchronize(self.handle), "jpegx_event_synchronize")

    def elapsed_ms(self, later):
        ms = ctypes.c_float()
        check(lib().jpegx_event_elapsed_ms(self.handle, later.handle, ctypes.byref(ms)), "jpegx_event_elapsed_ms")
        return ms.value

    def __del__(self):
        try:
            if self.handle:
                lib().jpegx_event_destroy(self.handle)
        except Exception:
            pass


def generate_plane_device(ptr, height, width, kind, seed=0, plane=0, row0=0, pitch=None, stream=None):
    k = synth._KINDS[kind] if isinstance(kind, str) else int(kind)
    check(lib().jpegx_generate_plane(ptr, height, width, pitch or width, k, seed, plane, row0, stream),
          "jpegx_generate_plane")


def forward_fused_device(in_ptr, height, width, out_ptr, mode="qtable", param=0.0, flags=F_PIXEL_INPUT,
                         pitch=None, pool=1, stream=None):
    """Enqueue steps 4+5+6 on device pointers (fp32 plane -> int16 zigzag stream)."""
    check(lib().jpegx_forward_fused_pooled(in_ptr, height, width, pitch or width * pool, pool, mode_of(mode),
                                           float(param), flags, out_ptr, stream), "jpegx_forward_fused")


class PlaneDesc(ctypes.Structure):
    """struct jpegx_plane_desc (include/jpegx.h)."""
    _fields_ = [("d_in", _vp), ("d_out", _vp), ("H", _int), ("W", _int), ("pitch", _pd), ("bs", _int)]


def forward_fused_planes_device(planes, mode="qtable", param=0.0, flags=F_PIXEL_INPUT, stream=None):
    """Enqueue steps (1+)4+5+6 for several planes as ONE launch.  ``planes``: iterable of
    (in_ptr, height, width, pitch_elems, block_size, out_ptr) with height/width AFTER pooling."""
    planes = list(planes)
    arr = (PlaneDesc * len(planes))()
    for d, (in_ptr, h, w, pitch, bs, out_ptr) in zip(arr, planes):
        d.d_in, d.d_out, d.H, d.W, d.pitch, d.bs = in_ptr, out_ptr, h, w, pitch, bs
    check(lib().jpegx_forward_fused_planes(arr, len(planes), mode_of(mode), float(param), flags, stream),
          "jpegx_forward_fused_planes")


def forward_fused_u8_device(in_ptr, height, width, out_ptr, mode="qtable", param=0.0, flags=0, pitch=None,
                            pool=1, stream=None):
    """Enqueue steps (1+)4+5+6 on a uint8 device plane of (height*pool, width*pool) bytes."""
    check(lib().jpegx_forward_fused_u8(in_ptr, height, width, pitch or width * pool, pool, mode_of(mode),
                                       float(param), flags, out_ptr, stream), "jpegx_forward_fused_u8")


def u8_path_ok(width_out, pool, pitch_bytes, mode="qtable", param=0.0):
    """What the uint8 kernels accept: block_size 1 (W % 16 == 0), 2 or 4, 16-byte aligned rows, and a
    quantiser whose multiplier is at most 2 (they pack to int16 without saturating)."""
    if mode_of(mode) == Q_DIVIDE and abs(float(param)) < 0.5:
        return False
    return pool in (1, 2, 4) and pitch_bytes % 16 == 0 and (pool > 1 or width_out % 16 == 0)


def inverse_fused_device(in_ptr, height, width, out_ptr, mode="qtable", param=0.0, flags=0, out_type=OUT_F32,
                         out_pitch=None, stream=None):
    check(lib().jpegx_inverse_fused(in_ptr, height, width, mode_of(mode), float(param), flags, out_ptr,
                                    out_pitch or width, out_type, stream), "jpegx_inverse_fused")


# ---------------------------------------------------------------------------------------------
# NumPy-in / NumPy-out conveniences (synchronous; H2D + kernel + D2H inside the library)
# ---------------------------------------------------------------------------------------------
def _plane(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    if a.ndim != 2:
        raise JpegxError("expected a 2-D plane, got shape %r" % (a.shape,))
    return a


def is_pixel_like(a):
    """True when every sample is a non-negative multiple of 2^-8 below 2^9 (JPEGX_F_PIXEL_INPUT)."""
    a = np.asarray(a)
    if a.size == 0:
        return False
    if a.dtype.kind in "ui":
        return bool(a.min() >= 0 and a.max() < 512)
    s = a * 256.0
    return bool(np.all(a >= 0) and np.all(a < 512) and np.all(s == np.rint(s)))


def forward_fused(plane, mode="qtable", param=0.0, pixel_input=None, flags_extra=0):
    """fp32 plane (H, W) -> int16 (H/8, W/8, 64): BasisChange+Quantization+ZigzagOrder.execute."""
    a = _plane(plane, np.float32)
    h, w = a.shape
    if pixel_input is None:
        pixel_input = is_pixel_like(a)
    out = np.empty((h // 8, w // 8, 64), dtype=np.int16)
    check(lib().jpegx_host_forward_fused(a.ctypes.data, h, w, w, mode_of(mode), float(param),
                                         (F_PIXEL_INPUT if pixel_input else 0) | flags_extra, out.ctypes.data),
          "jpegx_host_forward_fused")
    return out


def mean_pool_f64(plane, block_size):
    """SubSampling.execute on the device for any block_size: uint8 / integer-valued fp32 plane -> float64 means."""
    a = np.ascontiguousarray(plane)
    if a.dtype != np.uint8:
        a = a.astype(np.float32)
    bs = int(block_size)
    hh, ww = a.shape
    if hh % bs or ww % bs:
        raise JpegxError("plane must be a multiple of block_size in both dimensions")
    h, w = hh // bs, ww // bs
    din, dout = DeviceBuffer(a.nbytes), DeviceBuffer(h * w * 8)
    try:
        din.upload(a)
        check(lib().jpegx_mean_pool_f64(din.ptr, a.dtype.itemsize, h, w, ww, bs, dout.ptr, w, None), "jpegx_mean_pool_f64")
        return dout.download((h, w), np.float64)
    finally:
        din.free()
        dout.free()


def forward_fused_f64(plane, mode="qtable", param=0.0, flags_extra=0):
    """float64 plane (H, W) -> int16 (H/8, W/8, 64): steps 4+5+6 entirely in float64 in the reference's
    operation order (for samples that are not exact in fp32)."""
    a = _plane(plane, np.float64)
    h, w = a.shape
    out = np.empty((h // 8, w // 8, 64), dtype=np.int16)
    check(lib().jpegx_host_forward_fused_f64(a.ctypes.data, h, w, mode_of(mode), float(param), int(flags_extra), out.ctypes.data),
          "jpegx_host_forward_fused_f64")
    return out


def forward_fused_pooled(plane, block_size, mode="qtable", param=0.0, pixel_input=None):
    """Like forward_fused with the SubSampling mean-pool (block_size in 1,2,4) fused in."""
    a = _plane(plane, np.float32)
    hh, ww = a.shape
    bs = int(block_size)
    if hh % (8 * bs) or ww % (8 * bs):
        raise JpegxError("pooled plane must be a multiple of 8*block_size in both dimensions")
    h, w = hh // bs, ww // bs
    if pixel_input is None:
        pixel_input = is_pixel_like(a)
    out = np.empty((h // 8, w // 8, 64), dtype=np.int16)
    din, dout = DeviceBuffer(a.nbytes), DeviceBuffer(out.nbytes)
    try:
        din.upload(a)
        forward_fused_device(din.ptr, h, w, dout.ptr, mode, param, F_PIXEL_INPUT if pixel_input else 0,
                             pitch=ww, pool=bs)
        return dout.download(out.shape, np.int16)
    finally:
        din.free()
        dout.free()


def forward_fused_u8(plane, block_size=1, mode="qtable", param=0.0, flags_extra=0):
    """uint8 plane (H*bs, W*bs) -> int16 (H/8, W/8, 64): SubSampling (bs=2) + steps 4-6, 4x less PCIe."""
    a = _plane(plane, np.uint8)
    bs = int(block_size)
    hh, ww = a.shape
    if hh % (8 * bs) or ww % (8 * bs):
        raise JpegxError("plane must be a multiple of 8*block_size in both dimensions")
    h, w = hh // bs, ww // bs
    out = np.empty((h // 8, w // 8, 64), dtype=np.int16)
    din, dout = DeviceBuffer(a.nbytes), DeviceBuffer(out.nbytes)
    try:
        din.upload(a)
        forward_fused_u8_device(din.ptr, h, w, dout.ptr, mode, param, flags_extra, pitch=ww, pool=bs)
        return dout.download(out.shape, np.int16)
    finally:
        din.free()
        dout.free()


def inverse_fused(zz, mode="qtable", param=0.0, out="f32", clamp=False):
    """int16 (H/8, W/8, 64) -> (H, W) samples: ZigzagOrder+Quantization+BasisChange.invert (rounded)."""
    z = np.ascontiguousarray(zz, dtype=np.int16)
    if z.ndim != 3 or z.shape[2] != 64:
        raise JpegxError("expected a (H/8, W/8, 64) coefficient stream, got %r" % (z.shape,))
    h, w = z.shape[0] * 8, z.shape[1] * 8
    ot = _OUT_BY_NAME[out] if isinstance(out, str) else int(out)
    res = np.empty((h, w), dtype=_OUT_DTYPES[ot])
    check(lib().jpegx_host_inverse_fused(z.ctypes.data, h, w, mode_of(mode), float(param),
                                         F_CLAMP_U8 if clamp else 0, res.ctypes.data, w, ot),
          "jpegx_host_inverse_fused")
    return res


def inverse_fused_u8(zz, mode="qtable", param=0.0, inflate=1):
    """int16 (H/8, W/8, 64) -> uint8 (H*inflate, W*inflate): fused inverse + clamp + SubSampling.invert."""
    z = np.ascontiguousarray(zz, dtype=np.int16)
    if z.ndim != 3 or z.shape[2] != 64:
        raise JpegxError("expected a (H/8, W/8, 64) coefficient stream, got %r" % (z.shape,))
    h, w = z.shape[0] * 8, z.shape[1] * 8
    bs = int(inflate)
    pitch = (w * bs + 15) // 16 * 16                         # the kernel wants 16-byte aligned rows
    out = np.empty((h * bs, pitch), dtype=np.uint8)
    check(lib().jpegx_host_inverse_fused_u8_inflated(z.ctypes.data, h, w, mode_of(mode), float(param), 0, bs,
                                                     out.ctypes.data, pitch), "jpegx_host_inverse_fused_u8_inflated")
    return out[:, :w * bs]


def dct8x8_f64(a):
    """BasisChange.execute (DCT, dct_size 8), bit-identical float64."""
    a = _plane(a, np.float64)
    out = np.empty_like(a)
    check(lib().jpegx_host_dct8x8_f64(a.ctypes.data, a.shape[0], a.shape[1], out.ctypes.data), "jpegx_host_dct8x8_f64")
    return out


def idct8x8_f64(a, do_round=True):
    """BasisChange.invert (DCT, dct_size 8); do_round applies the np.round of basis_change.py:43."""
    a = _plane(a, np.float64)
    out = np.empty_like(a)
    check(lib().jpegx_host_idct8x8_f64(a.ctypes.data, a.shape[0], a.shape[1], out.ctypes.data, 1 if do_round else 0),
          "jpegx_host_idct8x8_f64")
    return out


def quantize_f64(a, mode, param=0.0):
    a = _plane(a, np.float64)
    out = np.empty_like(a)
    check(lib().jpegx_host_quantize_f64(a.ctypes.data, a.shape[0], a.shape[1], mode_of(mode), float(param),
                                        out.ctypes.data), "jpegx_host_quantize_f64")
    return out


def restore_f64(a, mode, param=0.0):
    a = _plane(a, np.float64)
    out = np.empty_like(a)
    check(lib().jpegx_host_restore_f64(a.ctypes.data, a.shape[0], a.shape[1], mode_of(mode), float(param),
                                       out.ctypes.data), "jpegx_host_restore_f64")
    return out


def zigzag(a):
    """ZigzagOrder.execute for dct_size 8 on any 2/4/8/16-byte dtype -> (H/8, W/8, 64)."""
    a = np.ascontiguousarray(a)
    if a.ndim != 2:
        raise JpegxError("expected a 2-D plane")
    h, w = a.shape
    out = np.empty((h // 8, w // 8, 64), dtype=a.dtype)
    check(lib().jpegx_host_zigzag(a.ctypes.data, h, w, a.dtype.itemsize, out.ctypes.data), "jpegx_host_zigzag")
    return out


def unzigzag(z):
    """ZigzagOrder.invert for dct_size 8."""
    z = np.ascontiguousarray(z)
    if z.ndim != 3 or z.shape[2] != 64:
        raise JpegxError("expected a (H/8, W/8, 64) array")
    h, w = z.shape[0] * 8, z.shape[1] * 8
    out = np.empty((h, w), dtype=z.dtype)
    check(lib().jpegx_host_unzigzag(z.ctypes.data, h, w, z.dtype.itemsize, out.ctypes.data), "jpegx_host_unzigzag")
    return out


def dct8x8_f32(a):
    a = _plane(a, np.float32)
    out = np.empty_like(a)
    check(lib().jpegx_host_dct8x8_f32(a.ctypes.data, a.shape[0], a.shape[1], out.ctypes.data), "jpegx_host_dct8x8_f32")
    return out


def idct8x8_f32(a):
    a = _plane(a, np.float32)
    out = np.empty_like(a)
    check(lib().jpegx_host_idct8x8_f32(a.ctypes.data, a.shape[0], a.shape[1], out.ctypes.data), "jpegx_host_idct8x8_f32")
    return out


def entropy_encode(zz):
    """int16 (..., 64) zigzag stream -> bytes: RunLengthEncoding.execute + RleBytestream.execute on the GPU."""
    z = np.ascontiguousarray(zz, dtype=np.int16)
    if z.ndim < 1 or z.shape[-1] != 64 or z.size == 0:
        raise JpegxError("expected a (..., 64) coefficient stream, got %r" % (z.shape,))
    nblocks = z.size // 64
    n = ctypes.c_size_t(0)
    check(lib().jpegx_host_entropy_encode(z.ctypes.data, nblocks, None, 0, ctypes.byref(n)), "jpegx_host_entropy_encode")
    out = np.empty(max(1, n.value), dtype=np.uint8)
    check(lib().jpegx_host_entropy_encode(z.ctypes.data, nblocks, out.ctypes.data, out.size, ctypes.byref(n)),
          "jpegx_host_entropy_encode")
    return out[:n.value].tobytes()


def entropy_block_sizes(zz):
    """Bytes of every block's code string as the device sizes pass computes them (uint32 per block)."""
    z = np.ascontiguousarray(zz, dtype=np.int16)
    if z.ndim < 1 or z.shape[-1] != 64 or z.size == 0:
        raise JpegxError("expected a (..., 64) coefficient stream, got %r" % (z.shape,))
    nblocks = z.size // 64
    L = lib()
    dzz, dws = DeviceBuffer(z.nbytes), DeviceBuffer(L.jpegx_entropy_workspace_bytes(nblocks))
    try:
        dzz.upload(z)
        check(L.jpegx_entropy_sizes(dzz.ptr, nblocks, dws.ptr, None), "jpegx_entropy_sizes")
        out = np.empty(nblocks, dtype=np.uint32)
        check(L.jpegx_entropy_block_sizes(dws.ptr, nblocks, out.ctypes.data, None), "jpegx_entropy_block_sizes")
        return out
    finally:
        dzz.free()
        dws.free()


def last_decode_level():
    """Which scheme took the last stream the device decoder saw on this thread: 0 segments sized from the stream's average,
    1 the smallest segments (second try), 2 the whole-stream scheme (test hook, not part of include/jpegx.h)."""
    L = lib()
    L.jpegx_internal_last_decode_level.restype = ctypes.c_int
    return int(L.jpegx_internal_last_decode_level())


def forward_u8_block_sizes(plane, block_size=1, mode="qtable", param=0.0):
    """What compress_band's first two launches leave behind for a uint8 plane: the coefficient stream, the bytes of every
    block's code string as the forward kernel itself counts them, the stream's total and the error flag of the scan
    (the library's internal entries; test and measurement hook, not part of include/jpegx.h)."""
    a = np.ascontiguousarray(plane, dtype=np.uint8)
    hh, ww = a.shape
    if block_size not in (1, 2, 4) or hh % (8 * block_size) or ww % (8 * block_size):
        raise JpegxError("uint8 plane of multiples of 8 * block_size expected")
    H, W = hh // block_size, ww // block_size
    nblocks = (H // 8) * (W // 8)
    L = lib()
    L.jpegx_internal_entropy_views.argtypes = [ctypes.c_void_p, ctypes.c_longlong] + [ctypes.POINTER(ctypes.c_void_p)] * 3
    L.jpegx_internal_entropy_views.restype = None
    L.jpegx_internal_forward_u8_sized.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_ssize_t, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                                  ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    L.jpegx_internal_forward_u8_sized.restype = ctypes.c_int
    L.jpegx_internal_entropy_scan.argtypes = [ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p]
    L.jpegx_internal_entropy_scan.restype = ctypes.c_int
    din, dzz, dws = DeviceBuffer(a.nbytes), DeviceBuffer(nblocks * 128), DeviceBuffer(L.jpegx_entropy_workspace_bytes(nblocks))
    try:
        din.upload(a)
        bb, wb, hb = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        L.jpegx_internal_entropy_views(dws.ptr, nblocks, ctypes.byref(bb), ctypes.byref(wb), ctypes.byref(hb))
        check(L.jpegx_internal_forward_u8_sized(din.ptr, H, W, ww, block_size, mode_of(mode), float(param), 0, dzz.ptr, bb, wb, hb, None), "forward_u8_sized")
        check(L.jpegx_internal_entropy_scan(nblocks, dws.ptr, None), "entropy_scan")
        tot = ctypes.c_ulonglong(0)
        rc = L.jpegx_entropy_total(dws.ptr, ctypes.byref(tot), None)
        sizes = np.empty(nblocks, dtype=np.uint32)
        check(L.jpegx_entropy_block_sizes(dws.ptr, nblocks, sizes.ctypes.data, None), "jpegx_entropy_block_sizes")
        return dzz.download((H // 8, W // 8, 64), np.int16), sizes, int(tot.value), rc
    finally:
        din.free()
        dzz.free()
        dws.free()


_ELEM_OF = {np.dtype(np.uint8): 1, np.dtype(np.int32): 4, np.dtype(np.int64): 8}
_pyapi = ctypes.pythonapi
_pyapi.PyBytes_FromStringAndSize.restype = ctypes.py_object
_pyapi.PyBytes_FromStringAndSize.argtypes = [ctypes.c_void_p, ctypes.c_ssize_t]
_pyapi.PyBytes_AsString.restype = ctypes.c_void_p
_pyapi.PyBytes_AsString.argtypes = [ctypes.py_object]


def compress_plane_native(plane, block_size=1, mode="qtable", param=0.0):
    """compress_plane through libjpegx's native host pipeline (jpegx_host_compress_begin / _finish): pooled
    device buffers and stream, range check + narrowing of int32 / int64 bands in native threads, and the
    byte stream copied from the device straight into the returned ``bytes`` object.  Returns None when the
    plane is not an 8-bit band in a layout that path takes (the caller then uses compress_plane)."""
    src = plane if isinstance(plane, np.ndarray) else np.asarray(plane)
    bs = int(block_size)
    elem = _ELEM_OF.get(src.dtype)
    if elem is None or src.ndim != 2 or not src.flags.c_contiguous:
        return None
    hh, ww = src.shape
    if hh == 0 or hh % (8 * bs) or ww % (8 * bs) or not 1 <= bs <= 255:
        return None
    if bs in (1, 2, 4) and (ww % 16 or not u8_path_ok(ww // bs, bs, ww, mode, param)):
        return None
    L = lib()
    n = ctypes.c_size_t(0)
    rc = L.jpegx_host_compress_begin(src.ctypes.data, elem, hh // bs, ww // bs, ww, bs, mode_of(mode), float(param),
                                     ctypes.byref(n))
    if rc == -4:                                        # JPEGX_E_UNSUPPORTED: not an 8-bit band after all
        return None
    check(rc, "jpegx_host_compress_begin")
    try:
        blob = _pyapi.PyBytes_FromStringAndSize(None, n.value)      # uninitialised bytes, filled below
        rc = L.jpegx_host_compress_finish(_pyapi.PyBytes_AsString(blob))
    except BaseException:
        L.jpegx_host_compress_abort()
        raise
    check(rc, "jpegx_host_compress_finish")
    return blob


ALLOC_FN = ctypes.CFUNCTYPE(ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t)   # jpegx_alloc_fn


def compress_image_native(planes, block_size=1, mode="qtable", param=0.0, prefix=None):
    """Steps 1-8 for all bands of one picture in ONE native job (jpegx_host_compress_image): the bands share one lock
    of the device's pool and alternate between two streams; once all byte counts are known libjpegx asks (through a
    callback) for ONE ``bytes`` object and copies every band from the device straight to its place in it.
    ``planes``: 2-D arrays of one shape and dtype (uint8 / int32 / int64), already padded to a multiple of
    8 * block_size.  With ``prefix`` (the container header) the result is the finished file -- prefix, then every
    band behind its '<L' byte count (file_format.generate_data); without it the list of the bands' byte strings.
    None when the bands are not 8-bit planes in a layout this path takes."""
    arrs = [p if isinstance(p, np.ndarray) else np.asarray(p) for p in planes]
    bs = int(block_size)
    if not arrs or len(arrs) > 4:
        return None
    a0 = arrs[0]
    elem = _ELEM_OF.get(a0.dtype)
    if elem is None or a0.ndim != 2:
        return None
    if any(a.shape != a0.shape or a.dtype != a0.dtype or not a.flags.c_contiguous for a in arrs):
        return None
    hh, ww = a0.shape
    if hh == 0 or hh % (8 * bs) or ww % (8 * bs) or not 1 <= bs <= 255:
        return None
    if bs in (1, 2, 4) and (ww % 16 or not u8_path_ok(ww // bs, bs, ww, mode, param)):
        return None
    L = lib()
    box = []

    def alloc(_user, nbytes):
        blob = _pyapi.PyBytes_FromStringAndSize(None, nbytes)       # uninitialised bytes, filled by the device copies
        box.append(blob)
        return _pyapi.PyBytes_AsString(blob)
    cb = ALLOC_FN(alloc)
    ptrs = (ctypes.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    sizes = (ctypes.c_size_t * len(arrs))()
    head = bytes(prefix) if prefix is not None else b""
    rc = L.jpegx_host_compress_image(ptrs, len(arrs), elem, hh // bs, ww // bs, ww, bs, mode_of(mode), float(param),
                                     head, len(head), 1 if prefix is not None else 0, cb, None, sizes)
    if rc == -4:                                        # JPEGX_E_UNSUPPORTED: not 8-bit bands after all
        return None
    check(rc, "jpegx_host_compress_image")
    whole = box[0]
    if prefix is not None:
        return whole
    out, at = [], 0
    for n in sizes:
        out.append(whole[at:at + n])
        at += n
    return out


def compress_image_packed(pixels, block_size=1, mode="qtable", param=0.0, prefix=None):
    """compress_image_native from pixel-interleaved samples: ``pixels`` is the (rows, cols, nbands) uint8 array that
    np.asarray(image) gives for a multi-band PIL image -- half the host time of image.split() plus an array per band; the
    planes are made on the device (jpegx_host_compress_image_packed).  rows and cols must be multiples of 8 * block_size
    (no padding step in between).  Same bytes as compress_image_native; None when this road does not apply."""
    a = pixels if isinstance(pixels, np.ndarray) else np.asarray(pixels)
    bs = int(block_size)
    if a.ndim != 3 or a.dtype != np.uint8 or not 1 <= a.shape[2] <= 4 or not a.flags.c_contiguous:
        return None
    hh, ww, nb = a.shape
    if hh == 0 or hh % (8 * bs) or ww % (8 * bs) or not 1 <= bs <= 255 or hh > 65535:
        return None
    if bs in (1, 2, 4) and (ww % 16 or not u8_path_ok(ww // bs, bs, ww, mode, param)):
        return None
    L = lib()
    box = []

    def alloc(_user, nbytes):
        blob = _pyapi.PyBytes_FromStringAndSize(None, nbytes)       # uninitialised bytes, filled by the device copies
        box.append(blob)
        return _pyapi.PyBytes_AsString(blob)
    cb = ALLOC_FN(alloc)
    sizes = (ctypes.c_size_t * nb)()
    head = bytes(prefix) if prefix is not None else b""
    check(L.jpegx_host_compress_image_packed(a.ctypes.data, nb, hh // bs, ww // bs, ww * nb, bs, mode_of(mode), float(param),
                                             head, len(head), 1 if prefix is not None else 0, cb, None, sizes),
          "jpegx_host_compress_image_packed")
    whole = box[0]
    if prefix is not None:
        return whole
    out, at = [], 0
    for n in sizes:
        out.append(whole[at:at + n])
        at += n
    return out


def decompress_image_native(blobs, height, width, block_size, mode, param, rows, cols, interleave=True):
    """All bands of one picture back in ONE native job (jpegx_host_decompress_image).  (height, width): a band after
    pooling (multiples of 8); the result is cropped to (rows, cols): uint8 (rows, cols, nbands) when ``interleave``
    (what PIL's Image.fromarray takes), else (nbands, rows, cols)."""
    bufs = [np.frombuffer(bytes(b), dtype=np.uint8) for b in blobs]
    n = len(bufs)
    ptrs = (ctypes.c_void_p * n)(*[b.ctypes.data if b.size else None for b in bufs])
    sizes = (ctypes.c_size_t * n)(*[b.size for b in bufs])
    out = np.empty((rows, cols, n) if interleave else (n, rows, cols), dtype=np.uint8)
    check(lib().jpegx_host_decompress_image(ptrs, sizes, n, int(height), int(width), int(block_size), mode_of(mode), float(param),
                                            out.ctypes.data, cols * n if interleave else cols, int(rows), int(cols), 1 if interleave else 0),
          "jpegx_host_decompress_image")
    return out


def compress_plane(plane, block_size=1, mode="qtable", param=0.0):
    """Steps 1-8 of the codec for one (already padded) plane with everything on the device:
    fused mean-pool + DCT + quantise + zigzag, then the entropy stage; only the final bytes come back."""
    src = np.asarray(plane)
    bs = int(block_size)
    if src.ndim != 2:
        raise JpegxError("expected a 2-D plane, got shape %r" % (src.shape,))
    hh, ww = src.shape
    if hh % (8 * bs) or ww % (8 * bs):
        raise JpegxError("plane must be a multiple of 8*block_size in both dimensions")
    blob = compress_plane_native(src, bs, mode, param)
    if blob is not None:
        return blob
    h, w = hh // bs, ww // bs
    nblocks = (h // 8) * (w // 8)
    as_u8 = (src.dtype == np.uint8 or (src.dtype.kind in "ui" and src.size and src.min() >= 0 and src.max() <= 255)) \
        and u8_path_ok(w, bs, ww, mode, param)
    a = np.ascontiguousarray(src, dtype=np.uint8 if as_u8 else np.float32)
    L = lib()
    din, dzz = DeviceBuffer(a.nbytes), DeviceBuffer(h * w * 2)
    dws = DeviceBuffer(L.jpegx_entropy_workspace_bytes(nblocks))
    dout = None
    try:
        din.upload(a)
        if as_u8:
            forward_fused_u8_device(din.ptr, h, w, dzz.ptr, mode, param, 0, pitch=ww, pool=bs)
        else:
            forward_fused_device(din.ptr, h, w, dzz.ptr, mode, param, F_PIXEL_INPUT if is_pixel_like(a) else 0,
                                 pitch=ww, pool=bs)
        check(L.jpegx_entropy_sizes(dzz.ptr, nblocks, dws.ptr, None), "jpegx_entropy_sizes")
        total = ctypes.c_ulonglong(0)
        check(L.jpegx_entropy_total(dws.ptr, ctypes.byref(total), None), "jpegx_entropy_total")
        dout = DeviceBuffer(max(1, total.value))
        check(L.jpegx_entropy_emit(dzz.ptr, nblocks, dws.ptr, dout.ptr, None), "jpegx_entropy_emit")
        return dout.download((total.value,), np.uint8).tobytes()
    finally:
        for b in (din, dzz, dws, dout):
            if b is not None:
                b.free()


def entropy_decode_gpu(blob, nblocks):
    """bytes -> int16 (nblocks, 64) with the parallel decoder on the device (jpegx_entropy_decode.hip)."""
    buf = np.frombuffer(bytes(blob), dtype=np.uint8)
    out = np.empty((int(nblocks), 64), dtype=np.int16)
    check(lib().jpegx_host_entropy_decode_gpu(buf.ctypes.data if buf.size else None, buf.size, int(nblocks),
                                              out.ctypes.data), "jpegx_host_entropy_decode_gpu")
    return out


def decompress_plane(blob, height, width, block_size=1, mode="qtable", param=0.0):
    """Steps 8-1 inverted for one plane, everything on the device: entropy decoding, un-zigzag, dequantise, IDCT,
    round, clamp, replicate block_size x block_size.  (height, width) = the plane AFTER pooling, multiples of 8.
    Returns uint8 (height * block_size, width * block_size)."""
    buf = np.frombuffer(bytes(blob), dtype=np.uint8)
    bs = int(block_size)
    pitch = (width * bs + 15) // 16 * 16
    out = np.empty((height * bs, pitch), dtype=np.uint8)
    check(lib().jpegx_host_decompress_plane(buf.ctypes.data if buf.size else None, buf.size, int(height), int(width), bs,
                                            mode_of(mode), float(param), out.ctypes.data, pitch), "jpegx_host_decompress_plane")
    return out[:, :width * bs]


def decompress_plane_i64(blob, height, width, block_size, mode, param, rows, cols):
    """decompress_plane as the (rows, cols) int64 band the reference's decompress_band returns."""
    buf = np.frombuffer(bytes(blob), dtype=np.uint8)
    out = np.empty((int(rows), int(cols)), dtype=np.int64)
    check(lib().jpegx_host_decompress_plane_i64(buf.ctypes.data if buf.size else None, buf.size, int(height), int(width),
                                                int(block_size), mode_of(mode), float(param), out.ctypes.data, int(rows), int(cols)),
          "jpegx_host_decompress_plane_i64")
    return out


def entropy_decode_device(d_bytes, nbytes, nblocks, d_workspace, d_zz, stream=None):
    """The device decoder on the caller's device buffers (jpegx_entropy_decode / _status): first try, then -- if that
    could not take the stream -- the second; returns the level that took it.  d_bytes: the stream with 16 zero bytes
    behind it; d_workspace: jpegx_entropy_decode_workspace_bytes(nbytes, nblocks) bytes, 256-byte aligned.  Raises
    JpegxError for a malformed stream, and for the (degenerate) streams only the whole-stream scheme takes."""
    L = lib()
    for level in (0, 1):
        check(L.jpegx_entropy_decode(d_bytes, int(nbytes), int(nblocks), d_workspace, d_zz, level, stream), "jpegx_entropy_decode")
        rc = L.jpegx_entropy_decode_status(d_workspace, stream)
        if rc == 0:
            return level
        if rc != 1:
            check(rc, "jpegx_entropy_decode_status")
    raise JpegxError("jpegx_entropy_decode: neither level took the stream (more block starts than a segment's tables hold)")


def entropy_decode(blob, nblocks):
    """bytes -> int16 (nblocks, 64): RleBytestream.invert + RunLengthEncoding.invert (host, sequential)."""
    buf = np.frombuffer(bytes(blob), dtype=np.uint8)
    out = np.empty((int(nblocks), 64), dtype=np.int16)
    check(lib().jpegx_host_entropy_decode(buf.ctypes.data if buf.size else None, buf.size, int(nblocks),
                                          out.ctypes.data), "jpegx_host_entropy_decode")
    return out
