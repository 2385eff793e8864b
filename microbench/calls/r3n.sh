O=gpurun_out/r3q; mkdir -p $O
export TMPDIR=/tmp
JPEGX_DECODE_STATS=1 JPEGX_LIB_PATH=microbench/_ab/libjpegx_stats.so timeout -k 10 120 python microbench/entropy_stage.py > $O/stats.json 2> $O/stats.err; echo rc=$?
sort $O/stats.err | uniq -c | sort -rn | head -20
