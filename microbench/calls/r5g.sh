O=gpurun_out/r5g; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "rc=$?"
