O=$GRAFT_REPO_ROOT/gpurun_out/r6e; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/kt_def -- python3 $GRAFT_REPO_ROOT/microbench/host_api.py > $O/def.txt 2> $O/def.err || exit 1
export JPEGX_LIB_PATH=$GRAFT_REPO_ROOT/microbench/_ab/libjpegx_emit9k.so
rocprofv3 --kernel-trace --output-format csv -d $O/kt_emit9k -- python3 $GRAFT_REPO_ROOT/microbench/host_api.py > $O/emit9k.txt 2> $O/emit9k.err || exit 1
