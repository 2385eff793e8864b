O=$GRAFT_REPO_ROOT/gpurun_out/r6i; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log; [ $rc = 0 ] || exit 1
timeout -k 10 500 python tests/soak_gpu.py --seconds 300 --seed 777 > $O/soak_300.txt 2>&1; echo "soak rc=$?"; tail -2 $O/soak_300.txt
