O=gpurun_out/r3d; mkdir -p $O
JPEGX_TRACE=1 python - > $O/trace.txt 2>&1 <<'PY'
import sys, time, numpy as np
sys.path.insert(0, "implementing-jpeg-compression_amd")
import jpegx
for kind in ("noise",):
    bands = [jpegx.synth.generate_plane(kind, 4096, 4096, seed=s, dtype=np.int64).astype(np.uint8) for s in (1, 2, 3)]
    for it in range(4):
        t0 = time.perf_counter()
        blobs = jpegx.compress_image_native(bands, 1, "qtable", 0.0)
        print("iteration", it, "%.2f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
        sys.stderr.flush()
PY
tail -60 $O/trace.txt
