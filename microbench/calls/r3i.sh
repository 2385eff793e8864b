O=gpurun_out/r3i; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for rep in 1 2 3; do
python microbench/ab_forward.py u8=0x0 u8_skipx=0x400 --u8 --planes 16 --rounds 5 | sed "s/^/noise rep$rep /"
python microbench/ab_forward.py u8=0x0 u8_skipx=0x400 --u8 --planes 16 --rounds 5 --kind smooth | sed "s/^/smooth rep$rep /"
done > $O/ab_u8.txt 2>&1
cat $O/ab_u8.txt
