"""Time-bounded parity soak on a real MI355X (test infrastructure, not collected by pytest):

    python tests/soak_gpu.py --seconds 300 > profiles/rNN_soak.txt

Draws planes, quantisers, kernel variants and output types at random from a seeded generator, runs the
HIP path through the C ABI and the CPU oracle (oracle/, spread over host threads) on the same inputs and
counts coefficient / sample mismatches.  The pytest parity tests use the same comparisons at sizes that
finish in seconds; this script exists to push a few 10^11 values through every kernel variant, with the
inputs that stress the rounding tiers (ties, near-ties, saturated edges, fractional pooled samples).
Prints one line per leg and a final `soak: ... mismatches N`; exit status 1 when N > 0."""
import argparse
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "implementing-jpeg-compression_amd"))
sys.path.insert(0, ROOT)

import jpegx as gpu          # noqa: E402
import oracle                # noqa: E402

THREADS = min(16, os.cpu_count() or 1)
POOL = ThreadPoolExecutor(THREADS)


def slabs(nrows, parts):
    edges = [nrows * i // parts for i in range(parts + 1)]
    return [(a, b) for a, b in zip(edges[:-1], edges[1:]) if b > a]


def oracle_inverse(zz, mode, param):
    parts = slabs(zz.shape[0], THREADS)
    outs = list(POOL.map(lambda ab: oracle.inverse_i16(zz[ab[0]:ab[1]], mode, param), parts))
    return np.concatenate(outs, axis=0)


def oracle_forward(a, mode, param):
    return oracle.forward_f32_mt(a, mode, param, threads=THREADS)


def oracle_forward_f64(a, mode, param):
    def one(ab):
        d = oracle.dct_plane(a[ab[0] * 8:ab[1] * 8])
        return oracle.zigzag_plane(oracle.quant_plane(d, mode, param)).astype(np.int64)
    return np.concatenate(list(POOL.map(one, slabs(a.shape[0] // 8, THREADS))), axis=0)


# ---------------------------------------------------------------------------------------------------------
# inputs
# ---------------------------------------------------------------------------------------------------------
def tie_stress(rng, h, w):
    """uint8 blocks whose sample sum is 8 mod 16 (DC/16 sits exactly on .5) and whose C4(x)C4 projection is
    pushed to the nearest multiple of 68 +- 34 (the (4,4) tie of the JPEG table) where one sample allows it."""
    a = rng.integers(0, 256, (h, w)).astype(np.int64)
    b = a.reshape(h // 8, 8, w // 8, 8)
    s = b.sum(axis=(1, 3))
    fix = (8 - s) % 16                                   # add 0..15 to one sample
    corner = b[:, 0, :, 0]
    room = corner + fix <= 255
    corner += np.where(room, fix, fix - 16)
    np.clip(corner, 0, 255, out=corner)
    return a.astype(np.uint8)


def make_plane(rng, kind, h, w):
    if kind == "noise":
        return rng.integers(0, 256, (h, w)).astype(np.uint8)
    if kind == "ties":
        return tie_stress(rng, h, w)
    if kind == "smooth":
        return gpu.synth.generate_plane("smooth", h, w, seed=int(rng.integers(0, 1 << 20))).astype(np.uint8)
    if kind == "edges":                                  # saturated checkerboards and bars: the largest AC magnitudes
        p = int(rng.integers(1, 9))
        y, x = np.mgrid[0:h, 0:w]
        a = (((y // p) ^ (x // p)) & 1) * 255
        flip = rng.random((h, w)) < 0.02
        return np.where(flip, 255 - a, a).astype(np.uint8)
    if kind == "flat":                                   # constant blocks with a few specks: almost everything quantises to 0
        a = np.repeat(np.repeat(rng.integers(0, 256, (h // 8, w // 8)), 8, 0), 8, 1)
        speck = rng.random((h, w)) < 0.01
        return np.where(speck, rng.integers(0, 256, (h, w)), a).astype(np.uint8)
    if kind == "lowamp":
        return rng.integers(120, 136, (h, w)).astype(np.uint8)
    raise ValueError(kind)


PIXEL_KINDS = ["noise", "ties", "smooth", "edges", "flat", "lowamp"]


def pick_quantiser(rng):
    r = int(rng.integers(0, 10))
    if r < 4:
        return "qtable", 0.0
    if r < 5:
        return "none", 0.0
    if r < 8:
        d = [1.0, 1.5, 2.0, 2.5, 3.0, 5.0, 7.0, 12.0, 16.0, 20.0, 40.0, 64.0, 100.0, 0.5, -3.0][int(rng.integers(0, 15))]
        return "divide", d
    return "discard", float(rng.integers(0, 10))


class Tally:
    def __init__(self):
        self.legs = {}

    def add(self, leg, values, bad, detail=None):
        v, b, first = self.legs.get(leg, (0, 0, None))
        if bad and first is None:
            first = detail
        self.legs[leg] = (v + int(values), b + int(bad), first)

    def total(self):
        return sum(v for v, _, _ in self.legs.values()), sum(b for _, b, _ in self.legs.values())


def compare(tally, leg, got, want, detail):
    got = np.asarray(got)
    want = np.asarray(want)
    if got.shape != want.shape:
        tally.add(leg, want.size, want.size, detail + (" shape %r vs %r" % (got.shape, want.shape)))
        return
    bad = int(np.count_nonzero(got.astype(np.int64) != want.astype(np.int64)))
    tally.add(leg, want.size, bad, detail)


FWD_VARIANTS = [("auto", 0), ("strip", gpu.F_TUNE_NO_COLUMN_UNITS), ("cols", gpu.F_TUNE_COLUMN_UNITS),
                ("xcd", gpu.F_TUNE_XCD_CONTIG), ("wpb", gpu.F_TUNE_WAVE_PER_BLOCK), ("nostrip", gpu.F_TUNE_NO_STRIP),
                ("float64 eight lanes", gpu.F_TUNE_F64_KERNEL), ("float64 lane per block", gpu.F_TUNE_F64_KERNEL | gpu.F_TUNE_F64_LANE_PER_BLOCK)]


def one_round(rng, tally, shape):
    h, w = shape
    kind = PIXEL_KINDS[int(rng.integers(0, len(PIXEL_KINDS)))]
    mode, param = pick_quantiser(rng)
    tag = "%s %s %g" % (kind, mode, param)
    u8 = make_plane(rng, kind, h, w)
    f32 = u8.astype(np.float32)
    want = oracle_forward(f32, mode, param)
    vname, vflag = FWD_VARIANTS[int(rng.integers(0, len(FWD_VARIANTS)))]
    compare(tally, "forward fp32 " + vname, gpu.forward_fused(f32, mode, param, flags_extra=vflag), want, tag)
    if gpu.u8_path_ok(w, 1, w, mode, param):
        compare(tally, "forward uint8", gpu.forward_fused_u8(u8, 1, mode, param), want, tag)
    compare(tally, "forward generic (non-pixel flag)", gpu.forward_fused(f32, mode, param, pixel_input=False), want, tag)

    # fused mean-pool prologue: 2x2 and 4x4, uint8 and fp32 inputs
    bs = 2 if rng.random() < 0.7 else 4
    pooled = oracle.mean_pool(u8[:h - h % bs, :w - w % bs], bs).astype(np.float32)                  # exact in fp32 for bs 2, 4
    wantp = oracle_forward(pooled, mode, param) if (h % (8 * bs) == 0 and w % (8 * bs) == 0) else None
    if wantp is not None:
        if gpu.u8_path_ok(w // bs, bs, w, mode, param):
            compare(tally, "forward pooled uint8 bs%d" % bs, gpu.forward_fused_u8(u8, bs, mode, param), wantp, tag)
        compare(tally, "forward pooled fp32 bs%d" % bs, gpu.forward_fused_pooled(f32, bs, mode, param), wantp, tag)

    # float64 kernel (any block_size): bs = 3 means on a crop whose size divides
    hh, ww = (h // 24) * 24, (w // 24) * 24
    if hh >= 24 and ww >= 24 and rng.random() < 0.3:
        means = oracle.mean_pool(u8[:hh, :ww], 3)
        compare(tally, "forward float64 (bs 3 means)", gpu.forward_fused_f64(means, mode, param),
                oracle_forward_f64(means, mode, param), tag)

    # signed / fractional samples through the generic variant
    if rng.random() < 0.3:
        g = (rng.normal(0, float(rng.choice([3.0, 60.0, 400.0])), (h, w))).astype(np.float32)
        try:
            wantg = oracle_forward(g, mode, param)
        except ValueError:                  # amplitudes beyond int16: the oracle refuses (the kernels saturate; pytest covers that)
            wantg = None
        if wantg is not None:
            compare(tally, "forward fractional fp32", gpu.forward_fused(g, mode, param), wantg, tag)

    # inverse: the stream just made, and the same stream with random coefficient noise on top
    for leg, zz in (("stream", want), ("perturbed", None)):
        if zz is None:
            if mode == "none" or np.abs(want).max() > 4000:
                continue
            bump = (rng.random(want.shape) < 0.05) * rng.integers(-3, 4, want.shape)
            zz = (want.astype(np.int64) + bump).astype(np.int16)
        ref = oracle_inverse(zz, mode, param)
        out = ["f32", "i16", "u8"][int(rng.integers(0, 3))]
        if out == "u8":
            k = [1, 1, 2, 4][int(rng.integers(0, 4))]
            if k * w > 8192:
                k = 1
            wantu = np.clip(ref, 0, 255).astype(np.uint8)
            if k > 1:
                wantu = np.repeat(np.repeat(wantu, k, 0), k, 1)
            compare(tally, "inverse uint8 inflate %d (%s)" % (k, leg), gpu.inverse_fused_u8(zz, mode, param, inflate=k), wantu, tag)
        elif out == "i16":
            clamp = bool(rng.integers(0, 2))
            wanti = np.clip(ref, 0, 255) if clamp else ref
            if np.abs(wanti).max() < 32768:
                compare(tally, "inverse int16%s (%s)" % (" clamped" if clamp else "", leg),
                        gpu.inverse_fused(zz, mode, param, out="i16", clamp=clamp), wanti, tag)
        else:
            compare(tally, "inverse fp32 (%s)" % leg, gpu.inverse_fused(zz, mode, param, out="f32"), ref, tag)

    # the whole band job through the native host pipeline (what compress_band / decompress_band call)
    if w % 16 == 0 and rng.random() < 0.5:
        k = [1, 2, 4, 3][int(rng.integers(0, 4))]
        hk = (h // (8 * k)) * 8 * k
        wk = (w // (16 * k)) * 16 * k if k != 3 else (w // 24) * 24
        if hk and wk and (mode != "none") and gpu.u8_path_ok(wk // k, k if k != 3 else 1, wk, mode, param):
            band = np.ascontiguousarray(u8[:hk, :wk])
            pooledk = oracle.mean_pool(band, k)
            zk = oracle_forward_f64(pooledk, mode, param).astype(np.int16)
            if np.abs(zk).max() < 16384:
                blob = gpu.compress_plane(band, k, mode, param)
                ref_blob = oracle.rle_bytestream(zk)
                tally.add("compress_plane block_size %d (bytes)" % k, len(ref_blob), 0 if blob == ref_blob else max(1, abs(len(blob) - len(ref_blob))), tag)
                if k != 3:
                    refp = np.clip(oracle_inverse(zk, mode, param), 0, 255).astype(np.uint8)
                    compare(tally, "decompress_plane block_size %d" % k, gpu.decompress_plane(ref_blob, hk // k, wk // k, k, mode, param),
                            np.repeat(np.repeat(refp, k, 0), k, 1), tag)

    # entropy stage both ways on the device
    if np.abs(want).max() < 16384 and rng.random() < 0.5:
        blob = gpu.entropy_encode(want)
        ref_blob = oracle.rle_bytestream(want)
        tally.add("entropy encode (bytes)", len(ref_blob), 0 if blob == ref_blob else max(1, abs(len(blob) - len(ref_blob))), tag)
        back = gpu.entropy_decode_gpu(ref_blob, want.shape[0] * want.shape[1])
        compare(tally, "entropy decode on the device", back.reshape(want.shape), want, tag)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=20261004)
    ap.add_argument("--sizes", default="1024,2048,4096,520,1096,8x8192,4096x8,24x4104,1032x1000,72x264,2048x4096",
                    help="square sizes or HxW shapes, comma separated (rounded down to multiples of 8)")
    args = ap.parse_args()
    gpu.require_device()
    oracle.build()
    rng = np.random.default_rng(args.seed)
    shapes = [tuple(int(v) - int(v) % 8 for v in (s.split("x") if "x" in s else (s, s))) for s in args.sizes.split(",")]
    tally = Tally()
    t0 = time.time()
    rounds = 0
    last = t0
    while time.time() - t0 < args.seconds:
        one_round(rng, tally, shapes[int(rng.integers(0, len(shapes)))])
        rounds += 1
        if time.time() - last > 45:
            v, b = tally.total()
            print("# %4.0f s: %d rounds, %.3e values compared, %d mismatches" % (time.time() - t0, rounds, v, b), flush=True)
            last = time.time()
    print("# device %s, seed %d, %d rounds in %.0f s, oracle on %d host threads" % (gpu.device_name(), args.seed, rounds, time.time() - t0, THREADS))
    for leg in sorted(tally.legs):
        v, b, first = tally.legs[leg]
        print("%-44s %14d values  %d mismatches%s" % (leg, v, b, ("   first at: " + first) if b else ""))
    v, b = tally.total()
    print("soak: %.3e values compared, mismatches %d" % (v, b))
    return 1 if b else 0


if __name__ == "__main__":
    sys.exit(main())
