#!/usr/bin/env python3
"""Why is the 4096^2 round trip (forward, inverse, forward, inverse ...) slower than its two halves timed alone?
Times F only, I only, F I alternating, F F I I, and F I with the inverse reading a stream the forward did not just write."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "implementing-jpeg-compression_amd"))
sys.path.insert(0, ROOT)
import jpegx
import bench

kind = sys.argv[1] if len(sys.argv) > 1 else "noise"
planes = int(sys.argv[2]) if len(sys.argv) > 2 else 16
n = 4096
H = n * planes
jpegx.require_device()
src, rec, zz, zz2 = (jpegx.DeviceBuffer(H * n * 4), jpegx.DeviceBuffer(H * n * 4), jpegx.DeviceBuffer(H * n * 2), jpegx.DeviceBuffer(H * n * 2))
for p in range(planes):
    jpegx.generate_plane_device(src.ptr + p * n * n * 4, n, n, kind, seed=0, plane=p)
F = lambda out=zz: jpegx.forward_fused_device(src.ptr, H, n, out.ptr, "qtable", 0.0, jpegx.F_PIXEL_INPUT)
I = lambda inp=zz: jpegx.inverse_fused_device(inp.ptr, H, n, rec.ptr, "qtable", 0.0, jpegx.F_CLAMP_U8, out_type=jpegx.OUT_F32)
F(zz2)
blocks = (H // 8) * (n // 8)
cases = {
    "F": (lambda: F(), 1, 384), "I": (lambda: I(), 1, 384),
    "F I": (lambda: (F(), I()), 1, 768), "F F I I": (lambda: (F(), F(), I(), I()), 2, 768),
    "F I(other stream)": (lambda: (F(), I(zz2)), 1, 768),
    "I F": (lambda: (I(), F()), 1, 768),
}
for rep in range(3):
    for name, (fn, div, bpb) in cases.items():
        ms = bench._timed_launches(jpegx, fn, 20) / div
        print("%-20s %.4f ms  %.1f GB/s  frac %.4f" % (name, ms, blocks * bpb / ms / 1e6, blocks * bpb / ms / 1e6 / 8000.0), flush=True)
