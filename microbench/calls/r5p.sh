O=$GRAFT_REPO_ROOT/gpurun_out/r5p; mkdir -p $O
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -6 $O/pytest.log; [ $rc = 0 ] || exit 1
timeout -k 10 300 python microbench/host_threads.py > $O/host_threads.txt 2>&1; echo "rc=$?"
