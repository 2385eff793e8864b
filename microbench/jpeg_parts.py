#!/usr/bin/env python3
"""Where Jpeg.compress / Jpeg.decompress spend their time on a 4096 x 4096 picture (PIL beside the native job)."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "implementing-jpeg-compression_amd"))
import jpegx, pipeline, file_format
from PIL import Image
jpegx.require_device()
size = 4096
def t(f, n=7):
    ts = []
    r = None
    for _ in range(n):
        r = None
        t0 = time.perf_counter(); r = f(); ts.append((time.perf_counter() - t0) * 1e3)
    return "%.1f (median %.1f)" % (min(ts), sorted(ts)[len(ts) // 2]), r
for kind in ("smooth", "noise"):
    px = np.stack([jpegx.synth.generate_plane(kind, size, size, seed=s, dtype=np.int64).astype(np.uint8) for s in (1, 2, 3)], axis=-1)
    im = Image.fromarray(px, mode="YCbCr")
    cfg = pipeline.Configuration(width=size, height=size, block_size=1, dct_size=8, quantization=pipeline.QuantizationMethod("qtable"))
    head = file_format.create_header(cfg)
    mode, param = cfg.quantization.gpu_mode()
    print(kind)
    s1, arr = t(lambda: np.asarray(im)); print("  np.asarray(image)                ", s1)
    s2, _ = t(lambda: [np.asarray(b) for b in im.split()]); print("  split + asarray per band         ", s2)
    s3, data = t(lambda: jpegx.compress_image_packed(arr, 1, mode, param, prefix=head)); print("  compress_image_packed(pixels)    ", s3)
    planes = [np.ascontiguousarray(px[..., k]) for k in range(3)]
    s4, _ = t(lambda: jpegx.compress_image_native(planes, 1, mode, param, prefix=head)); print("  compress_image_native(planes)    ", s4)
    s5, _ = t(lambda: pipeline.Jpeg(cfg).compress(im)); print("  Jpeg.compress(image)             ", s5)
    s6, _ = t(lambda: pipeline.Jpeg.decompress(data)); print("  Jpeg.decompress(container)       ", s6)
    s7, cd = t(lambda: file_format.read_data(data)); print("  file_format.read_data            ", s7)
    config, d = cd
    s8, packed = t(lambda: pipeline._decompress_image((d.y, d.cb, d.cr), config)); print("  _decompress_image (native job)   ", s8)
    s9, _ = t(lambda: Image.fromarray(packed, mode="YCbCr")); print("  Image.fromarray(packed)          ", s9)
