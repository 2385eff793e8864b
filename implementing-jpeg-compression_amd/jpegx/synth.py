"""Integer-only synthetic plane generators.

The same arithmetic is implemented on the device by ``jpegx_generate_plane``
(csrc/jpegx_stage.hip) so that host and GPU produce identical bits without any
host->device copy.  Values are integers in 0..255 ("noise") or 0..206
("smooth"), exactly representable in fp32.

There is no reference counterpart: the reference reads its planes from PIL
(``/root/reference/util.py:110-112``); these generators stand in for image data.
"""
import numpy as np

KIND_NOISE = 0
KIND_SMOOTH = 1
_KINDS = {"noise": KIND_NOISE, "smooth": KIND_SMOOTH}

_M32 = np.uint32(0xFFFFFFFF)


def hash32(x):
    """lowbias32 integer mixer on uint32 arrays (wrap-around arithmetic)."""
    x = np.asarray(x, dtype=np.uint32).copy()
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x7FEB352D)
    x ^= x >> np.uint32(15)
    x *= np.uint32(0x846CA68B)
    x ^= x >> np.uint32(16)
    return x


def plane_seed(seed, plane):
    """Per-plane seed: hash32(seed + plane * 0x9E3779B9) (mod 2^32)."""
    v = (int(seed) + int(plane) * 0x9E3779B9) & 0xFFFFFFFF
    return int(hash32(np.array([v], dtype=np.uint32))[0])


def generate_plane(kind, height, width, seed=0, plane=0, row0=0, dtype=np.float32):
    """Rows ``row0 .. row0+height-1`` of synthetic plane ``plane``.

    kind: "noise" (uniform 0..255, worst case for rounding ties) or "smooth"
    (sawtooth ramps + checkerboard edges + 4 bits of noise, 0..206).
    The pixel index hashed is ``y * width + x`` with the *absolute* row ``y``.
    """
    k = _KINDS[kind] if isinstance(kind, str) else int(kind)
    ps = np.uint32(plane_seed(seed, plane))
    with np.errstate(over="ignore"):
        y = (np.arange(height, dtype=np.uint32) + np.uint32(row0))[:, None]
        x = np.arange(width, dtype=np.uint32)[None, :]
        h = hash32((y * np.uint32(width) + x) ^ ps)
        if k == KIND_NOISE:
            v = h >> np.uint32(24)
        elif k == KIND_SMOOTH:
            ramp = ((np.uint32(3) * x + np.uint32(5) * y) >> np.uint32(4)) & np.uint32(127)
            chk = np.uint32(64) * (((x >> np.uint32(6)) ^ (y >> np.uint32(6))) & np.uint32(1))
            v = ramp + chk + (h & np.uint32(15))
        else:
            raise ValueError("unknown synthetic plane kind %r" % (kind,))
    return v.astype(dtype)
