O=$GRAFT_REPO_ROOT/gpurun_out/r5k; mkdir -p $O
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/kt_def -- python3 $GRAFT_REPO_ROOT/microbench/entropy_stage.py > $O/es_def.json 2> $O/es_def.err || exit 1
