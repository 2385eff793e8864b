O=$GRAFT_REPO_ROOT/gpurun_out/r6j; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log; [ $rc = 0 ] || exit 1
python microbench/host_api.py > $O/host_api.txt 2>&1; echo "rc=$?"
