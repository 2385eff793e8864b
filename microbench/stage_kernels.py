#!/usr/bin/env python3
"""Device-time of the per-stage kernels behind the stand-alone step classes (one 4096x4096 plane)."""
import os
import statistics
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "implementing-jpeg-compression_amd"))
import jpegx  # noqa: E402


def timed(fn, rounds=5, iters=5):
    L = jpegx.lib()
    e0, e1 = jpegx.Event(), jpegx.Event()
    for _ in range(3):
        fn()
    jpegx.check(L.jpegx_device_synchronize())
    ts = []
    for _ in range(rounds):
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_ms(e1) / iters)
    return statistics.median(ts)


def main():
    jpegx.require_device()
    L = jpegx.lib()
    n = 4096
    nblk = (n // 8) ** 2
    f32a, f32b = jpegx.DeviceBuffer(n * n * 4), jpegx.DeviceBuffer(n * n * 4)
    f64a, f64b = jpegx.DeviceBuffer(n * n * 8), jpegx.DeviceBuffer(n * n * 8)
    jpegx.generate_plane_device(f32a.ptr, n, n, "smooth")
    jpegx.check(L.jpegx_memset(f64a.ptr, 0, n * n * 8, None))
    rows = [
        ("dct8x8_f32", lambda: jpegx.check(L.jpegx_dct8x8_f32(f32a.ptr, n, n, n, f32b.ptr, n, None)), 512),
        ("idct8x8_f32", lambda: jpegx.check(L.jpegx_idct8x8_f32(f32a.ptr, n, n, n, f32b.ptr, n, None)), 512),
        ("dct8x8_f64", lambda: jpegx.check(L.jpegx_dct8x8_f64(f64a.ptr, n, n, n, f64b.ptr, n, None)), 1024),
        ("idct8x8_f64", lambda: jpegx.check(L.jpegx_idct8x8_f64(f64a.ptr, n, n, n, f64b.ptr, n, 1, None)), 1024),
        ("quantize_f64", lambda: jpegx.check(L.jpegx_quantize_f64(f64a.ptr, n, n, n, 3, 0.0, f64b.ptr, n, None)), 1024),
        ("restore_f64", lambda: jpegx.check(L.jpegx_restore_f64(f64a.ptr, n, n, n, 3, 0.0, f64b.ptr, n, None)), 1024),
        ("zigzag f64", lambda: jpegx.check(L.jpegx_zigzag(f64a.ptr, n, n, n, 8, f64b.ptr, None)), 1024),
        ("unzigzag f64", lambda: jpegx.check(L.jpegx_unzigzag(f64a.ptr, n, n, 8, f64b.ptr, n, None)), 1024),
        ("zigzag i16", lambda: jpegx.check(L.jpegx_zigzag(f64a.ptr, n, n, n, 2, f64b.ptr, None)), 256),
    ]
    for name, fn, bpb in rows:
        ms = timed(fn)
        print("%-14s %.4f ms  %8.1f Mblocks/s  %7.1f GB/s (%d B/block)" % (name, ms, nblk / ms / 1e3, nblk * bpb / ms / 1e6, bpb))


if __name__ == "__main__":
    main()
