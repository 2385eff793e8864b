#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configs on one GPU (they are parity-test cases, not the
bench line; numbers go to profiles/ for the record).

  c3: 8192x8192 YCbCr 4:2:0 -- Y fused forward + Cb, Cr with the fused 2x2 mean prologue
  c4: round trip of 4096x4096 planes -- fused forward then fused inverse (f32 out), PSNR vs input
  inv: fused inverse alone (f32 / i16 / u8 output)
Every figure: median of interleaved rounds, HIP events around back-to-back steps; bytes are the
algorithmic bytes of SURVEY.md 8(d).
"""
import argparse
import json
import os
import statistics
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "implementing-jpeg-compression_amd"))
import jpegx  # noqa: E402

PIX = jpegx.F_PIXEL_INPUT


def timed(fn, rounds=7, iters=10):
    L = jpegx.lib()
    e0, e1 = jpegx.Event(), jpegx.Event()
    for _ in range(40):
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


def timed_wall(fn, rounds=7, iters=20):
    """Host wall clock around `iters` enqueues + one device synchronize (for multi-stream steps)."""
    import time
    L = jpegx.lib()
    for _ in range(40):
        fn()
    jpegx.check(L.jpegx_device_synchronize())
    ts = []
    for _ in range(rounds):
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        jpegx.check(L.jpegx_device_synchronize())
        ts.append((time.perf_counter() - t0) * 1e3 / iters)
    return statistics.median(ts)


def c3(kind, layout="separate"):
    """layout: "separate" = three launches, one per plane (Cb, Cr 4096 waves each: a single generation of
    workgroups on 256 CUs); "stacked" = Cb and Cr stored one above the other and transformed by ONE launch;
    "streams" = stacked chroma on a second HIP stream, concurrent with the Y launch (wall-clock timed)."""
    n = 8192
    ybuf = jpegx.DeviceBuffer(n * n * 4)
    chroma = jpegx.DeviceBuffer(2 * n * n * 4)             # Cb rows, then Cr rows
    jpegx.generate_plane_device(ybuf.ptr, n, n, kind, seed=0, plane=0)
    jpegx.generate_plane_device(chroma.ptr, n, n, kind, seed=0, plane=1)
    jpegx.generate_plane_device(chroma.ptr + n * n * 4, n, n, kind, seed=0, plane=2)
    zy, zc = jpegx.DeviceBuffer(n * n * 2), jpegx.DeviceBuffer(n * n)

    streams = [None, None]
    if layout == "streams":
        import ctypes
        for i in range(2):
            h = ctypes.c_void_p()
            jpegx.check(jpegx.lib().jpegx_stream_create(ctypes.byref(h)))
            streams[i] = h.value

    def step():
        jpegx.forward_fused_device(ybuf.ptr, n, n, zy.ptr, "qtable", 0.0, PIX, stream=streams[0])
        if layout == "streams":
            jpegx.forward_fused_device(chroma.ptr, n, n // 2, zc.ptr, "qtable", 0.0, PIX, pool=2, stream=streams[1])
        elif layout == "stacked":
            jpegx.forward_fused_device(chroma.ptr, n, n // 2, zc.ptr, "qtable", 0.0, PIX, pool=2)
        else:
            jpegx.forward_fused_device(chroma.ptr, n // 2, n // 2, zc.ptr, "qtable", 0.0, PIX, pool=2)
            jpegx.forward_fused_device(chroma.ptr + n * n * 4, n // 2, n // 2, zc.ptr + n * n // 2, "qtable", 0.0, PIX, pool=2)
    ms = timed_wall(step) if layout == "streams" else timed(step)
    blocks = (n // 8) ** 2 + 2 * (n // 16) ** 2
    nbytes = (n // 8) ** 2 * 384 + 2 * (n // 16) ** 2 * 1152
    return {"config": "c3 8192x8192 YCbCr 4:2:0 forward (Y + 2x pooled chroma, %s chroma launches)" % layout,
            "kind": kind, "ms": round(ms, 4),
            "Mblocks_per_s": round(blocks / ms / 1e3, 1), "GBps": round(nbytes / ms / 1e6, 1),
            "frac_of_8TBps": round(nbytes / ms / 1e6 / 8000, 4), "blocks": blocks, "bytes": nbytes}


def c4(kind, planes=16):
    import oracle
    n = 4096
    H = n * planes
    src, rec = jpegx.DeviceBuffer(H * n * 4), jpegx.DeviceBuffer(H * n * 4)
    zz = jpegx.DeviceBuffer(H * n * 2)
    for p in range(planes):
        jpegx.generate_plane_device(src.ptr + p * n * n * 4, n, n, kind, seed=0, plane=p)

    def step():
        jpegx.forward_fused_device(src.ptr, H, n, zz.ptr, "qtable", 0.0, PIX)
        jpegx.inverse_fused_device(zz.ptr, H, n, rec.ptr, "qtable", 0.0, jpegx.F_CLAMP_U8, out_type=jpegx.OUT_F32)
    ms = timed(step)
    a = src.download((n, n), np.float32).astype(np.float64)
    b = rec.download((n, n), np.float32).astype(np.float64)
    psnr = 10 * np.log10(255.0 ** 2 / np.mean((a - b) ** 2))
    sub = a[:512, :512].astype(np.float32)
    ref = np.clip(oracle.inverse_i16(oracle.forward_f32(sub, "qtable"), "qtable"), 0, 255)
    same = bool(np.array_equal(ref, b[:512, :512]))
    blocks = (H // 8) * (n // 8)
    return {"config": "c4 round trip 4096x4096 x%d: fused forward + fused inverse (f32, clamped)" % planes, "kind": kind,
            "ms": round(ms, 4), "Mblocks_per_s": round(blocks / ms / 1e3, 1), "GBps": round(blocks * 768 / ms / 1e6, 1),
            "frac_of_8TBps": round(blocks * 768 / ms / 1e6 / 8000, 4), "psnr_dB_plane0": round(float(psnr), 3),
            "reconstruction_equals_oracle_512x512": same}


def inv(kind, out, planes=16):
    n = 4096
    H = n * planes
    src, zz = jpegx.DeviceBuffer(H * n * 4), jpegx.DeviceBuffer(H * n * 2)
    for p in range(planes):
        jpegx.generate_plane_device(src.ptr + p * n * n * 4, n, n, kind, seed=0, plane=p)
    jpegx.forward_fused_device(src.ptr, H, n, zz.ptr, "qtable", 0.0, PIX)
    ot = {"f32": jpegx.OUT_F32, "i16": jpegx.OUT_I16, "u8": jpegx.OUT_U8}[out]
    ms = timed(lambda: jpegx.inverse_fused_device(zz.ptr, H, n, src.ptr, "qtable", 0.0, 0, out_type=ot))
    bpb = 128 + {"f32": 256, "i16": 128, "u8": 64}[out]
    blocks = (H // 8) * (n // 8)
    return {"config": "inverse fused, %s output" % out, "kind": kind, "ms": round(ms, 4),
            "Mblocks_per_s": round(blocks / ms / 1e3, 1), "GBps": round(blocks * bpb / ms / 1e6, 1),
            "frac_of_8TBps": round(blocks * bpb / ms / 1e6 / 8000, 4), "bytes_per_block": bpb}


def entropy(kind, planes=16):
    """Steps 7+8 on the device: sizes + scan, then emit; bytes = 2 reads of the stream + the output."""
    import ctypes
    L = jpegx.lib()
    n = 4096
    H = n * planes
    src, zz = jpegx.DeviceBuffer(H * n * 4), jpegx.DeviceBuffer(H * n * 2)
    for p in range(planes):
        jpegx.generate_plane_device(src.ptr + p * n * n * 4, n, n, kind, seed=0, plane=p)
    jpegx.forward_fused_device(src.ptr, H, n, zz.ptr, "qtable", 0.0, PIX)
    nblocks = (H // 8) * (n // 8)
    ws = jpegx.DeviceBuffer(L.jpegx_entropy_workspace_bytes(nblocks))
    jpegx.check(L.jpegx_entropy_sizes(zz.ptr, nblocks, ws.ptr, None))
    total = ctypes.c_ulonglong(0)
    jpegx.check(L.jpegx_entropy_total(ws.ptr, ctypes.byref(total), None))
    out = jpegx.DeviceBuffer(total.value)
    ms_sizes = timed(lambda: jpegx.check(L.jpegx_entropy_sizes(zz.ptr, nblocks, ws.ptr, None)))
    ms_emit = timed(lambda: jpegx.check(L.jpegx_entropy_emit(zz.ptr, nblocks, ws.ptr, out.ptr, None)))
    nbytes = 2 * nblocks * 128 + total.value
    ms = ms_sizes + ms_emit
    return {"config": "entropy stage (run-length + bit packing) on %d planes 4096x4096" % planes, "kind": kind,
            "ms_sizes_scan": round(ms_sizes, 4), "ms_emit": round(ms_emit, 4), "Mblocks_per_s": round(nblocks / ms / 1e3, 1),
            "compressed_bytes_per_block": round(total.value / nblocks, 2), "GBps": round(nbytes / ms / 1e6, 1),
            "frac_of_8TBps": round(nbytes / ms / 1e6 / 8000, 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="*", default=["c3", "c4", "inv", "entropy"])
    a = ap.parse_args()
    jpegx.require_device()
    for kind in ("smooth", "noise"):
        if "c3" in a.what:
            print(json.dumps(c3(kind, "separate")), flush=True)
            print(json.dumps(c3(kind, "stacked")), flush=True)
            print(json.dumps(c3(kind, "streams")), flush=True)
        if "c4" in a.what:
            print(json.dumps(c4(kind)), flush=True)
        if "inv" in a.what:
            for out in ("f32", "i16", "u8"):
                print(json.dumps(inv(kind, out)), flush=True)
        if "entropy" in a.what:
            print(json.dumps(entropy(kind)), flush=True)


if __name__ == "__main__":
    main()
