#!/usr/bin/env python3
"""The entropy stage on the device, both ways, for the record (profiles/rNN_entropy_stage.*): encode (k_rle_sizes + scans +
k_rle_emit on a device-resident stream, HIP events) and decode (segment-local block starts, scan, block decode from LDS;
through jpegx_host_entropy_decode_gpu, so the wall time includes the copies -- the kernels' own times come
from `rocprofv3 --kernel-trace --stats -- python3 microbench/entropy_stage.py`).  4096 x 4096 planes, JPEG table.
Algorithmic bytes: encode reads the 128-byte block twice (sizes, emit) and writes the code bytes; decode reads the code
bytes and writes 128 bytes per block."""
import ctypes
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "implementing-jpeg-compression_amd"))
import jpegx  # noqa: E402


def main():
    jpegx.require_device()
    L = jpegx.lib()
    n, planes = 4096, 8
    nblk1 = (n // 8) ** 2
    out = {}
    for kind in ("noise", "smooth"):
        src, zz = jpegx.DeviceBuffer(planes * n * n * 4), jpegx.DeviceBuffer(planes * n * n * 2)
        for p in range(planes):
            jpegx.generate_plane_device(src.ptr + p * n * n * 4, n, n, kind, seed=0, plane=p)
        jpegx.forward_fused_device(src.ptr, n * planes, n, zz.ptr, "qtable", 0.0, jpegx.F_PIXEL_INPUT)
        nblk = nblk1 * planes
        ws = jpegx.DeviceBuffer(int(L.jpegx_entropy_workspace_bytes(nblk)))
        jpegx.check(L.jpegx_entropy_sizes(zz.ptr, nblk, ws.ptr, None), "sizes")
        tot = ctypes.c_ulonglong(0)
        jpegx.check(L.jpegx_entropy_total(ws.ptr, ctypes.byref(tot), None), "total")
        coded = jpegx.DeviceBuffer(max(16, tot.value))

        def encode():
            jpegx.check(L.jpegx_entropy_sizes(zz.ptr, nblk, ws.ptr, None), "sizes")
            jpegx.check(L.jpegx_entropy_emit(zz.ptr, nblk, ws.ptr, coded.ptr, None), "emit")
        e0, e1 = jpegx.Event(), jpegx.Event()
        for _ in range(5):
            encode()
        jpegx.check(L.jpegx_device_synchronize(), "sync")
        e0.record()
        for _ in range(20):
            encode()
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_ms(e1) / 20
        enc_bytes = nblk * 256 + tot.value
        out[kind] = {"encode": {"planes": planes, "blocks": nblk, "coded_bytes": int(tot.value), "bytes_per_block_coded": round(tot.value / nblk, 2),
                                "ms": round(ms, 4), "Mblocks_per_s": round(nblk / ms / 1e3, 1), "algorithmic_GBps": round(enc_bytes / ms / 1e6, 1)}}
        # decode: one plane's code bytes through the host entry (H2D + kernels + D2H)
        one = jpegx.DeviceBuffer(n * n * 2)
        jpegx.check(L.jpegx_memcpy_d2d(one.ptr, zz.ptr, n * n * 2, None), "d2d")
        jpegx.check(L.jpegx_device_synchronize(), "sync")
        plane_zz = one.download((nblk1, 64), np.int16)
        blob = jpegx.entropy_encode(plane_zz)
        back = jpegx.entropy_decode_gpu(blob, nblk1)
        assert np.array_equal(back, plane_zz)
        ts = []
        for _ in range(12):
            t0 = time.perf_counter()
            jpegx.entropy_decode_gpu(blob, nblk1)
            ts.append((time.perf_counter() - t0) * 1e3)
        out[kind]["decode_host_entry"] = {"blocks": nblk1, "coded_bytes": len(blob), "wall_ms_median": round(sorted(ts)[len(ts) // 2], 3),
                                          "note": "H2D of the code bytes + 3 kernels (k_seg_starts, k_seg_scan, k_dec_blocks_lds) + D2H of 32 MiB"}
        for b in (src, zz, ws, coded, one):
            b.free()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
