O=$GRAFT_REPO_ROOT/gpurun_out/r5l; mkdir -p $O
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -15 $O/pytest.log; [ $rc = 0 ] || exit 1
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/kt_def -- python3 $GRAFT_REPO_ROOT/microbench/entropy_stage.py > $O/es_def.json 2> $O/es_def.err || exit 1
cd $GRAFT_REPO_ROOT
python microbench/host_api.py > $O/host_api.txt 2>&1
