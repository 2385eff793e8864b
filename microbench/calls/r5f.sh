O=gpurun_out/r5f; mkdir -p $O
for T in 8 16 32; do JPEGX_WIDEN_THREADS=$T python microbench/host_api.py > $O/host_api_w$T.txt 2>&1 || exit 1; done
python -m pytest tests -m gpu -x -q -k "pipeline or host" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
