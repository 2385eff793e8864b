O=$GRAFT_REPO_ROOT/gpurun_out/r5j; mkdir -p $O
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for i in 1 2; do
  python bench.py --no-cpu-baseline > $O/def$i.json 2>> $O/err.txt || exit 1
  JPEGX_LIB_PATH=microbench/_ab/libjpegx_nosdwa.so python bench.py --no-cpu-baseline > $O/nosdwa$i.json 2>> $O/err.txt || exit 1
done
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/kt_def -- python3 $GRAFT_REPO_ROOT/microbench/entropy_stage.py > $O/es_def.json 2> $O/es_def.err || exit 1
export JPEGX_LIB_PATH=$GRAFT_REPO_ROOT/microbench/_ab/libjpegx_nosdwa.so
rocprofv3 --kernel-trace --output-format csv -d $O/kt_nosdwa -- python3 $GRAFT_REPO_ROOT/microbench/entropy_stage.py > $O/es_nosdwa.json 2> $O/es_nosdwa.err || exit 1
