O=gpurun_out/r5a; mkdir -p $O
export TMPDIR=/tmp
JPEGX_DECODE_STATS=$O/trace.bin JPEGX_LIB_PATH=microbench/_ab/libjpegx_stats.so timeout -k 10 120 python microbench/decode_trace_run.py
python microbench/decode_trace.py $O/trace.bin.noise > $O/noise.txt
python microbench/decode_trace.py $O/trace.bin.smooth > $O/smooth.txt
rm -f $O/trace.bin.*
