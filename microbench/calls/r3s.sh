O=gpurun_out/r3s; mkdir -p $O
export TMPDIR=/tmp
JPEGX_DECODE_STATS=$O/trace.bin JPEGX_LIB_PATH=microbench/_ab/libjpegx_stats.so timeout -k 10 120 python - <<'PY'
import sys, os, numpy as np
sys.path.insert(0, "implementing-jpeg-compression_amd")
import jpegx
jpegx.require_device()
n = 4096
src = jpegx.DeviceBuffer(n*n*4); zz = jpegx.DeviceBuffer(n*n*2)
for kind in ("noise", "smooth"):
    jpegx.generate_plane_device(src.ptr, n, n, kind, seed=0, plane=0)
    jpegx.forward_fused_device(src.ptr, n, n, zz.ptr, "qtable", 0.0, jpegx.F_PIXEL_INPUT)
    plane = zz.download((n*n//64, 64), np.int16)
    blob = jpegx.entropy_encode(plane)
    for _ in range(3):
        back = jpegx.entropy_decode_gpu(blob, n*n//64)
    assert np.array_equal(back, plane)
    os.rename(os.environ["JPEGX_DECODE_STATS"], os.environ["JPEGX_DECODE_STATS"] + "." + kind)
    print(kind, len(blob))
PY
python microbench/decode_trace.py $O/trace.bin.noise 7160
python microbench/decode_trace.py $O/trace.bin.smooth 7221
rm -f $O/trace.bin.*
