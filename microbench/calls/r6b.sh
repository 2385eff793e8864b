O=$GRAFT_REPO_ROOT/gpurun_out/r6b; mkdir -p $O
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest.log; [ $rc = 0 ] || exit 1
cd /tmp
for S in default 3584 3328; do
  unset JPEGX_DECODE_SEG
  case $S in default) ;; *) export JPEGX_DECODE_SEG=$S;; esac
  rocprofv3 --kernel-trace --output-format csv -d $O/kt_$S -- python3 $GRAFT_REPO_ROOT/microbench/entropy_stage.py > $O/es_$S.json 2> $O/es_$S.err || exit 1
done
