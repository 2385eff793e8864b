#!/usr/bin/env python3
"""Decode one 4096 x 4096 noise and one smooth band with a -DJPEGX_DECODE_STATS build (JPEGX_LIB_PATH) and keep the last
call's per-segment time stamps: JPEGX_DECODE_STATS=<file> python microbench/decode_trace_run.py -> <file>.noise / .smooth"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "implementing-jpeg-compression_amd"))
import jpegx
jpegx.require_device()
n = 4096
src = jpegx.DeviceBuffer(n * n * 4)
zz = jpegx.DeviceBuffer(n * n * 2)
for kind in ("noise", "smooth"):
    jpegx.generate_plane_device(src.ptr, n, n, kind, seed=0, plane=0)
    jpegx.forward_fused_device(src.ptr, n, n, zz.ptr, "qtable", 0.0, jpegx.F_PIXEL_INPUT)
    plane = zz.download((n * n // 64, 64), np.int16)
    blob = jpegx.entropy_encode(plane)
    for _ in range(3):
        back = jpegx.entropy_decode_gpu(blob, n * n // 64)
    assert np.array_equal(back, plane)
    os.rename(os.environ["JPEGX_DECODE_STATS"], os.environ["JPEGX_DECODE_STATS"] + "." + kind)
    print(kind, len(blob), "bytes")
