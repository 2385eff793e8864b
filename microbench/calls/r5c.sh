O=gpurun_out/r5c; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
python microbench/host_api.py > $O/host_api.txt 2>&1 && python microbench/host_api.py --image > $O/host_api_image.txt 2>&1 && python bench.py > $O/bench.json 2> $O/bench.err; echo "rc=$?"
