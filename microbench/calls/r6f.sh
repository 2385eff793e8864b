O=$GRAFT_REPO_ROOT/gpurun_out/r6f; mkdir -p $O
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -8 $O/pytest.log; [ $rc = 0 ] || exit 1
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
