O=gpurun_out/r3l; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_entropy.py -x -q > $O/pytest_entropy.log 2>&1; echo "pytest entropy rc=$?"; tail -15 $O/pytest_entropy.log
