#!/usr/bin/env python3
"""Device decoder on noise bands whose streams sit at the edge of one round of k_seg_starts waves:
rocprofv3 --kernel-trace -- python3 microbench/decode_plan_probe.py  (JPEGX_DECODE_SEG forces a segment size)"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "implementing-jpeg-compression_amd"))
import jpegx
jpegx.require_device()
for h, w in ((4096, 4096), (4096, 4352), (4096, 4608)):
    src = jpegx.DeviceBuffer(h * w * 4)
    zz = jpegx.DeviceBuffer(h * w * 2)
    jpegx.generate_plane_device(src.ptr, h, w, "noise", seed=0, plane=0)
    jpegx.forward_fused_device(src.ptr, h, w, zz.ptr, "qtable", 0.0, jpegx.F_PIXEL_INPUT)
    plane = zz.download((h * w // 64, 64), np.int16)
    blob = jpegx.entropy_encode(plane)
    for _ in range(8):
        back = jpegx.entropy_decode_gpu(blob, h * w // 64)
    assert np.array_equal(back, plane)
    print(h, w, len(blob), "bytes", flush=True)
    src.free(); zz.free()
