O=gpurun_out/r3r; mkdir -p $O
export TMPDIR=/tmp
for v in stats nowait; do
JPEGX_DECODE_STATS=1 JPEGX_LIB_PATH=microbench/_ab/libjpegx_$v.so timeout -k 10 120 python microbench/entropy_stage.py > $O/$v.json 2> $O/$v.err; echo $v rc=$?
grep "jpegx decode" $O/$v.err | tail -3; tail -2 $O/$v.err
done
cd /tmp
JPEGX_LIB_PATH=$GRAFT_REPO_ROOT/microbench/_ab/libjpegx_nowait.so rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt -- python3 $GRAFT_REPO_ROOT/microbench/entropy_stage.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
grep -h "k_seg_decode" $O/kt/*/*kernel_stats.csv | head
