O=gpurun_out/r3j; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
python microbench/host_api.py > $O/host_api.txt 2>&1; cat $O/host_api.txt
timeout -k 10 600 bash profiles/collect_forward.sh $O > $O/collect.log 2>&1; echo "collect rc=$?"; tail -c 1500 $O/p_kt.json
