O=gpurun_out/r3h; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
P=microbench/_ab/libjpegx_prev.so
bash microbench/ab_libs.sh "u8=0x0 u8_skipx=0x400 --u8 --planes 16 --rounds 5" $P > $O/ab_u8.txt 2>&1
bash microbench/ab_libs.sh "u8=0x0 u8_skipx=0x400 --u8 --planes 16 --rounds 5 --kind smooth" $P > $O/ab_u8_smooth.txt 2>&1
bash microbench/ab_libs.sh "nt=0x1 --planes 64 --rounds 5" $P > $O/ab_f32.txt 2>&1
bash microbench/ab_libs.sh "pooled=0x1 --pool 2 --planes 4 --rounds 5" $P > $O/ab_pool2.txt 2>&1
bash microbench/ab_libs.sh "u8=0x0 --u8 --pool 2 --planes 8 --rounds 5" $P > $O/ab_u8_pool2.txt 2>&1
bash microbench/ab_libs.sh "u8=0x0 --u8 --pool 4 --planes 2 --rounds 5" $P > $O/ab_u8_pool4.txt 2>&1
bash microbench/ab_libs.sh "cols=0x40001 --mode divide --param 7 --planes 16 --rounds 5" $P > $O/ab_cols.txt 2>&1
bash microbench/ab_libs.sh "inv=0x0 skipx=0x400 --direction inverse --out-type u8 --planes 16 --rounds 5" $P > $O/ab_inv_u8.txt 2>&1
timeout -k 10 200 python tests/soak_gpu.py --seconds 90 --seed 31 > $O/soak_90.txt 2>&1; echo "soak rc=$?"; tail -4 $O/soak_90.txt
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-600 $O/bench.json
