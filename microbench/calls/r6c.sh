O=$GRAFT_REPO_ROOT/gpurun_out/r6c; mkdir -p $O
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k entropy > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log; [ $rc = 0 ] || exit 1
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/kt_default -- python3 $GRAFT_REPO_ROOT/microbench/entropy_stage.py > $O/es_default.json 2> $O/es_default.err || exit 1
