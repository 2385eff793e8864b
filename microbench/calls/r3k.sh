O=gpurun_out/r3k; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cat $O/bench.json
