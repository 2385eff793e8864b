#!/usr/bin/env python3
"""Interleaved A/B timing of fused-kernel variants in ONE process (guide rule 24).

usage: python microbench/ab_forward.py [--kind noise] [--planes 16] [--rounds 7] name=flagshex ...
e.g.   python microbench/ab_forward.py nt=0x1 plain=0x101
Each variant is timed as `--iters` back-to-back launches between two HIP events; rounds are
interleaved across variants; prints median / min ms per launch and GB/s (384 B per block).
"""
import argparse
import os
import statistics
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "implementing-jpeg-compression_amd"))
import jpegx  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="+")
    ap.add_argument("--kind", default="noise")
    ap.add_argument("--planes", type=int, default=16)
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--warm-ms", type=float, default=30.0, help="device time of untimed launches before the rounds: a burst of ten "
                    "0.15 ms launches from idle clocks reads 5-10 %% low (round 3: 28.0 vs 30.9 Gblocks/s for the uint8 forward)")
    ap.add_argument("--direction", default="forward", choices=["forward", "inverse"])
    ap.add_argument("--out-type", default="f32")
    ap.add_argument("--pool", type=int, default=1, help="forward: fused mean-pool factor (input is pool x larger)")
    ap.add_argument("--u8", action="store_true", help="forward: uint8 input planes (jpegx_forward_fused_u8)")
    ap.add_argument("--mode", default="qtable", choices=["qtable", "none", "divide", "discard"])
    ap.add_argument("--param", type=float, default=0.0)
    a = ap.parse_args()
    jpegx.require_device()
    L = jpegx.lib()
    size, planes = a.size, a.planes
    H, W = size * planes, size
    nblk = (H // 8) * (W // 8)
    pool = a.pool
    plane_buf = jpegx.DeviceBuffer(H * W * 4 * pool * pool)
    zz_buf = jpegx.DeviceBuffer(H * W * 2)
    for p in range(planes):
        jpegx.generate_plane_device(plane_buf.ptr + p * size * size * 4 * pool * pool, size * pool, size * pool,
                                    a.kind, seed=0, plane=p)
    u8_buf = None
    if a.u8:
        import numpy as np
        u8_buf = jpegx.DeviceBuffer(H * W * pool * pool)
        rows = 512                                      # convert on the host in slabs (setup only)
        for y0 in range(0, H * pool, rows):
            slab = plane_buf.download((rows, W * pool), np.float32, offset=y0 * W * pool * 4).astype(np.uint8)
            u8_buf.upload(slab, offset=y0 * W * pool)
    jpegx.forward_fused_device(plane_buf.ptr, H, W, zz_buf.ptr, a.mode, a.param, jpegx.F_PIXEL_INPUT, pool=pool)
    jpegx.check(L.jpegx_device_synchronize())
    variants = [(v.split("=")[0], int(v.split("=")[1], 0)) for v in a.variants]
    ot = {"f32": 0, "i16": 1, "u8": 2}[a.out_type]

    def launch(flags):
        if a.direction == "forward" and a.u8:
            jpegx.forward_fused_u8_device(u8_buf.ptr, H, W, zz_buf.ptr, a.mode, a.param, flags & ~1, pool=pool)
        elif a.direction == "forward":
            jpegx.forward_fused_device(plane_buf.ptr, H, W, zz_buf.ptr, a.mode, a.param, flags, pool=pool)
        else:
            jpegx.inverse_fused_device(zz_buf.ptr, H, W, plane_buf.ptr, a.mode, a.param, flags, out_type=ot)

    times = {n: [] for n, _ in variants}
    e0, e1 = jpegx.Event(), jpegx.Event()
    for n, f in variants:
        launch(f)
    jpegx.check(L.jpegx_device_synchronize())
    spent = 0.0
    while spent < a.warm_ms:                              # clocks up before anything is timed
        e0.record()
        for n, f in variants:
            for _ in range(a.iters):
                launch(f)
        e1.record()
        e1.synchronize()
        spent += e0.elapsed_ms(e1)
    for _ in range(a.rounds):
        for n, f in variants:
            e0.record()
            for _ in range(a.iters):
                launch(f)
            e1.record()
            e1.synchronize()
            times[n].append(e0.elapsed_ms(e1) / a.iters)
    for n, f in variants:
        med, mn = statistics.median(times[n]), min(times[n])
        if a.direction == "forward":
            bpb = (64 if a.u8 else 256) * pool * pool + 128
        else:
            bpb = 128 + {"f32": 256, "i16": 128, "u8": 64}[a.out_type]
        print("%-12s flags=0x%03x  median %.4f ms  min %.4f ms  %.1f GB/s (median, %d B/block)  %.1f Mblocks/s"
              % (n, f, med, mn, bpb * nblk / med / 1e6, bpb, nblk / med / 1e3))


if __name__ == "__main__":
    main()
