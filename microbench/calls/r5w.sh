O=$GRAFT_REPO_ROOT/gpurun_out/r5w; mkdir -p $O
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "image or pipeline or Jpeg or jpeg" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/pytest.log; [ $rc = 0 ] || exit 1
python microbench/host_api.py --image > $O/host_api_image.txt 2>&1; echo "rc=$?"
