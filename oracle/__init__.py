"""CPU oracle for the per-block transform path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package; the product path never does.

Two restatements live here:
  * ``jpegx_oracle.c`` (loaded through ctypes below): defined floating-point order,
    pinned bit for bit against the imported reference by tests/golden/*.npz.  This is
    the checker.
  * ``ref_loop.py``: a faithful Python/NumPy per-block loop with the reference's call
    structure (16 ``ndarray.dot`` calls, reciprocal-multiply round, 64-element gather
    per block) -- used only to time "what the reference costs" on the GPU box's host.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libjpegx_oracle.so")

Q_NONE, Q_DISCARD, Q_DIVIDE, Q_QTABLE = 0, 1, 2, 3
MODE_BY_NAME = {"none": Q_NONE, "discard": Q_DISCARD, "divide": Q_DIVIDE, "qtable": Q_QTABLE}

_lib = None


def build(force=False):
    """Compile jpegx_oracle.c with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_SO) or \
            os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "jpegx_oracle.c")):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        dp = ctypes.POINTER(ctypes.c_double)
        c_int, c_dbl, c_pd = ctypes.c_int, ctypes.c_double, ctypes.c_ssize_t
        L.jo_dct_plane.argtypes = [dp, c_int, c_int, c_pd, dp]
        L.jo_idct_plane.argtypes = [dp, c_int, c_int, dp]
        L.jo_quant_plane.argtypes = [dp, c_int, c_int, c_int, c_dbl, dp]
        L.jo_restore_plane.argtypes = [dp, c_int, c_int, c_int, c_dbl, dp]
        L.jo_zigzag_plane.argtypes = [dp, c_int, c_int, dp]
        L.jo_unzigzag_plane.argtypes = [dp, c_int, c_int, dp]
        L.jo_forward_f32.argtypes = [ctypes.POINTER(ctypes.c_float), c_int, c_int, c_pd, c_int, c_dbl,
                                     ctypes.POINTER(ctypes.c_int16), dp]
        L.jo_forward_f32_mt.argtypes = [ctypes.POINTER(ctypes.c_float), c_int, c_int, c_pd, c_int, c_dbl,
                                        ctypes.POINTER(ctypes.c_int16), c_int]
        L.jo_inverse_i16.argtypes = [ctypes.POINTER(ctypes.c_int16), c_int, c_int, c_int, c_dbl,
                                     ctypes.POINTER(ctypes.c_int32), dp]
        L.jo_mean_pool.argtypes = [dp, c_int, c_int, c_int, dp]
        L.jo_rle_block_tuples.argtypes = [ctypes.POINTER(ctypes.c_int16), c_int, ctypes.POINTER(c_int)]
        L.jo_rle_bytestream.argtypes = [ctypes.POINTER(ctypes.c_int16), ctypes.c_longlong, c_int,
                                        ctypes.POINTER(ctypes.c_uint8), ctypes.c_longlong,
                                        ctypes.POINTER(ctypes.c_uint32)]
        L.jo_rle_bytestream.restype = ctypes.c_longlong
        L.jo_rle_stream_tuples.argtypes = [ctypes.POINTER(ctypes.c_uint8), ctypes.c_longlong, ctypes.POINTER(c_int), ctypes.c_longlong]
        L.jo_rle_stream_tuples.restype = ctypes.c_longlong
        L.jo_rle_tuples_decode.argtypes = [ctypes.POINTER(c_int), ctypes.c_longlong, ctypes.c_longlong, c_int,
                                           ctypes.POINTER(ctypes.c_int32)]
        L.jo_rle_tuples_decode.restype = ctypes.c_longlong
        for n in ("jo_table_dct_matrix", "jo_table_dct_normalized", "jo_table_norm_diag"):
            getattr(L, n).restype = dp
        for n in ("jo_table_qtable", "jo_table_zigzag"):
            getattr(L, n).restype = ctypes.POINTER(ctypes.c_int)
        if L.jo_selfcheck() != 0:
            raise RuntimeError("oracle build has no fused fma(); rebuild with -mfma")
        _lib = L
    return _lib


def _mode(mode):
    return MODE_BY_NAME[mode] if isinstance(mode, str) else int(mode)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


def _check(rc, what):
    if rc != 0:
        raise ValueError("oracle.%s failed with code %d" % (what, rc))


def tables():
    L = lib()
    return {
        "dct_matrix": np.ctypeslib.as_array(L.jo_table_dct_matrix(), (8, 8)).copy(),
        "dct_normalized": np.ctypeslib.as_array(L.jo_table_dct_normalized(), (8, 8)).copy(),
        "norm_diag": np.ctypeslib.as_array(L.jo_table_norm_diag(), (8,)).copy(),
        "qtable": np.ctypeslib.as_array(L.jo_table_qtable(), (8, 8)).copy(),
        "zigzag8": np.ctypeslib.as_array(L.jo_table_zigzag(), (64,)).copy(),
    }


def dct_plane(a):
    """BasisChange.execute (pipeline/basis_change.py:11-18), dct_size 8."""
    a = _f64(a)
    out = np.empty_like(a)
    _check(lib().jo_dct_plane(_p(a, ctypes.c_double), a.shape[0], a.shape[1], a.shape[1],
                              _p(out, ctypes.c_double)), "dct_plane")
    return out


def idct_plane(a, rounded=True):
    """BasisChange.invert (pipeline/basis_change.py:28-43); rounded=False gives the floats."""
    a = _f64(a)
    out = np.empty_like(a)
    _check(lib().jo_idct_plane(_p(a, ctypes.c_double), a.shape[0], a.shape[1],
                               _p(out, ctypes.c_double)), "idct_plane")
    return np.rint(out).astype(np.int64) if rounded else out


def quant_plane(a, mode, param=0.0):
    """Quantization.execute (pipeline/quantization.py:8-18)."""
    a = _f64(a)
    out = np.empty_like(a)
    _check(lib().jo_quant_plane(_p(a, ctypes.c_double), a.shape[0], a.shape[1], _mode(mode),
                                float(param), _p(out, ctypes.c_double)), "quant_plane")
    return out


def restore_plane(a, mode, param=0.0):
    """Quantization.invert (pipeline/quantization.py:20-30)."""
    a = _f64(a)
    out = np.empty_like(a)
    _check(lib().jo_restore_plane(_p(a, ctypes.c_double), a.shape[0], a.shape[1], _mode(mode),
                                  float(param), _p(out, ctypes.c_double)), "restore_plane")
    return out


def zigzag_plane(a):
    """ZigzagOrder.execute (pipeline/zigzag_order.py:85-99) -> (H/8, W/8, 64)."""
    a = _f64(a)
    h, w = a.shape
    out = np.empty((h // 8, w // 8, 64), dtype=np.float64)
    _check(lib().jo_zigzag_plane(_p(a, ctypes.c_double), h, w, _p(out, ctypes.c_double)), "zigzag_plane")
    return out


def unzigzag_plane(z):
    """ZigzagOrder.invert (pipeline/zigzag_order.py:101-119)."""
    z = _f64(z)
    h, w = z.shape[0] * 8, z.shape[1] * 8
    out = np.empty((h, w), dtype=np.float64)
    _check(lib().jo_unzigzag_plane(_p(z, ctypes.c_double), h, w, _p(out, ctypes.c_double)), "unzigzag_plane")
    return out


def forward_f32(plane, mode, param=0.0, want_dct=False):
    """Steps 4+5+6 on an fp32 plane -> int16 (H/8, W/8, 64) [, float64 (H, W) coefficients]."""
    a = np.ascontiguousarray(plane, dtype=np.float32)
    h, w = a.shape
    zz = np.empty((h // 8, w // 8, 64), dtype=np.int16)
    dct = np.empty((h, w), dtype=np.float64) if want_dct else None
    _check(lib().jo_forward_f32(_p(a, ctypes.c_float), h, w, w, _mode(mode), float(param),
                                _p(zz, ctypes.c_int16),
                                _p(dct, ctypes.c_double) if want_dct else None), "forward_f32")
    return (zz, dct) if want_dct else zz


def forward_f32_mt(plane, mode, param=0.0, threads=1):
    """forward_f32 with the block rows spread over OpenMP threads (identical output)."""
    a = np.ascontiguousarray(plane, dtype=np.float32)
    h, w = a.shape
    zz = np.empty((h // 8, w // 8, 64), dtype=np.int16)
    _check(lib().jo_forward_f32_mt(_p(a, ctypes.c_float), h, w, w, _mode(mode), float(param),
                                   _p(zz, ctypes.c_int16), int(threads)), "forward_f32_mt")
    return zz


def inverse_i16(zz, mode, param=0.0, want_float=False):
    """Steps 6+5+4 inverted: int16 (H/8, W/8, 64) -> int32 (H, W) rounded, unclamped."""
    z = np.ascontiguousarray(zz, dtype=np.int16)
    h, w = z.shape[0] * 8, z.shape[1] * 8
    out = np.empty((h, w), dtype=np.int32)
    fl = np.empty((h, w), dtype=np.float64) if want_float else None
    _check(lib().jo_inverse_i16(_p(z, ctypes.c_int16), h, w, _mode(mode), float(param),
                                _p(out, ctypes.c_int32),
                                _p(fl, ctypes.c_double) if want_float else None), "inverse_i16")
    return (out, fl) if want_float else out


def mean_pool(a, bs):
    """SubSampling.execute (pipeline/subsampling.py:9-11) for sizes that divide evenly."""
    a = _f64(a)
    h, w = a.shape
    out = np.empty((h // bs, w // bs), dtype=np.float64)
    _check(lib().jo_mean_pool(_p(a, ctypes.c_double), h, w, int(bs), _p(out, ctypes.c_double)), "mean_pool")
    return out


def rle_block_tuples(values):
    """RunLengthBlock.encode + as_tuple for one block (pipeline/run_length_encoding.py:14-32); EOB -> (0, 0)."""
    z = np.ascontiguousarray(values, dtype=np.int16).reshape(-1)
    buf = np.zeros(3 * (2 * z.size + 2), dtype=np.int32)
    k = lib().jo_rle_block_tuples(_p(z, ctypes.c_int16), z.size, _p(buf, ctypes.c_int))
    if k < 0:
        raise ValueError("amplitude needs more than 15 bits")
    out = []
    for r, s_, a in buf[:3 * k].reshape(k, 3).tolist():
        out.append((0, 0) if (r == 0 and s_ == 0) else (r, s_, a))
    return out


def rle_bytestream(zz, want_block_bytes=False):
    """Steps 7+8 (RunLengthEncoding.execute + RleBytestream.execute) on an integer (..., n) zigzag stream."""
    z = np.ascontiguousarray(zz, dtype=np.int16)
    n = z.shape[-1]
    nblocks = z.size // n
    sizes = np.zeros(nblocks, dtype=np.uint32)
    total = lib().jo_rle_bytestream(_p(z, ctypes.c_int16), nblocks, n, None, 0, _p(sizes, ctypes.c_uint32))
    if total < 0:
        raise ValueError("amplitude needs more than 15 bits")
    out = np.zeros(total, dtype=np.uint8)
    got = lib().jo_rle_bytestream(_p(z, ctypes.c_int16), nblocks, n, _p(out, ctypes.c_uint8), total, None)
    assert got == total
    return (out.tobytes(), sizes) if want_block_bytes else out.tobytes()


class RleStreamError(ValueError):
    """What the reference raises on a damaged stream: ValueError (int('', 2), reshape) or BadRleCodeError."""


def _triples(tuples):
    flat = np.zeros((len(tuples), 3), dtype=np.int32)
    for i, t in enumerate(tuples):
        flat[i, :len(t)] = t
    return flat


def rle_stream_tuples(blob):
    """RleBytestream.invert (pipeline/rle_byte_stream.py:61-88): the stream's tuples, end markers as (0, 0)."""
    b = np.frombuffer(bytes(blob), dtype=np.uint8)
    cap = b.size + 1                                   # a tuple takes at least a byte, the first may sit in a partial read
    buf = np.zeros((cap, 3), dtype=np.int32)
    k = lib().jo_rle_stream_tuples(_p(b, ctypes.c_uint8) if b.size else None, b.size, _p(buf, ctypes.c_int), cap)
    if k < 0:
        raise RleStreamError("bad code in the stream (%d)" % k)
    return [(0, 0) if (r == 0 and s_ == 0) else (r, s_, a) for r, s_, a in buf[:k].tolist()]


def rle_tuples_decode(tuples, nblocks, n=64):
    """RunLengthEncoding.invert (pipeline/run_length_encoding.py:66-79) -> (nblocks, n) int32."""
    flat = np.ascontiguousarray(_triples(tuples))
    out = np.zeros((nblocks, n), dtype=np.int32)
    got = lib().jo_rle_tuples_decode(_p(flat, ctypes.c_int), len(flat), nblocks, n, _p(out, ctypes.c_int32))
    if got < 0:
        raise RleStreamError("the blocks' values do not fill a (%d, %d) array" % (nblocks, n))
    return out


def rle_decode(blob, nblocks, n=64):
    """Steps 8 + 7 backwards on a byte stream -> (nblocks, n) int32; RleStreamError where the reference raises."""
    b = np.frombuffer(bytes(blob), dtype=np.uint8)
    cap = b.size + 1
    buf = np.zeros((cap, 3), dtype=np.int32)
    k = lib().jo_rle_stream_tuples(_p(b, ctypes.c_uint8) if b.size else None, b.size, _p(buf, ctypes.c_int), cap)
    if k < 0:
        raise RleStreamError("bad code in the stream (%d)" % k)
    out = np.zeros((nblocks, n), dtype=np.int32)
    got = lib().jo_rle_tuples_decode(_p(buf, ctypes.c_int), k, nblocks, n, _p(out, ctypes.c_int32))
    if got < 0:
        raise RleStreamError("the blocks' values do not fill a (%d, %d) array" % (nblocks, n))
    return out
