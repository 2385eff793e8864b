set -o pipefail
O=gpurun_out/r3c; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -15 $O/pytest.log
P=microbench/_ab/libjpegx_prev.so
bash microbench/ab_libs.sh "u8=0x0 u8_skipx=0x400 --u8 --planes 16 --rounds 5" $P > $O/ab_u8.txt 2>&1
python microbench/ab_forward.py u8=0x0 u8_skipx=0x400 --u8 --planes 16 --rounds 5 --kind smooth > $O/ab_u8_smooth.txt 2>&1
python microbench/host_api.py > $O/host_api.txt 2>&1; echo "host_api rc=$?"
python microbench/host_api.py --image > $O/host_api_image.txt 2>&1; echo "host_api image rc=$?"
tail -3 $O/ab_u8.txt; cat $O/ab_u8_smooth.txt $O/host_api_image.txt
