O=gpurun_out/r3t; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_entropy.py -x -q > $O/pytest_entropy.log 2>&1; echo "pytest entropy rc=$?"; tail -5 $O/pytest_entropy.log
JPEGX_DECODE_STATS=$O/trace.bin JPEGX_LIB_PATH=microbench/_ab/libjpegx_stats.so timeout -k 10 120 python microbench/decode_trace_run.py
python microbench/decode_trace.py $O/trace.bin.noise
python microbench/decode_trace.py $O/trace.bin.smooth
rm -f $O/trace.bin.*
