set -o pipefail
O=gpurun_out/r3b; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -3 $O/pytest.log
microbench/valu_rate > $O/valu_rate.txt 2>&1; echo "valu_rate rc=$?"
P=microbench/_ab/libjpegx_prev.so
bash microbench/ab_libs.sh "u8=0x0 u8_skipx=0x400 direct=0x8 --u8 --planes 16 --rounds 5" $P > $O/ab_u8.txt 2>&1
bash microbench/ab_libs.sh "u8=0x0 u8_skipx=0x400 --u8 --planes 16 --rounds 5 --kind smooth" $P > $O/ab_u8_smooth.txt 2>&1
bash microbench/ab_libs.sh "inv=0x0 skipx=0x400 --direction inverse --out-type u8 --planes 16 --rounds 5" $P > $O/ab_inv_u8.txt 2>&1
bash microbench/ab_libs.sh "inv=0x0 skipx=0x400 --direction inverse --out-type u8 --planes 16 --rounds 5 --kind smooth" $P > $O/ab_inv_u8_smooth.txt 2>&1
bash microbench/ab_libs.sh "inv=0x0 skipx=0x400 --direction inverse --out-type i16 --planes 16 --rounds 5" $P > $O/ab_inv_i16.txt 2>&1
bash microbench/ab_libs.sh "inv=0x0 skipx=0x400 --direction inverse --out-type f32 --planes 16 --rounds 5" $P > $O/ab_inv_f32.txt 2>&1
bash microbench/ab_libs.sh "nt=0x1 skipx=0x401 --planes 64 --rounds 5" $P > $O/ab_f32.txt 2>&1
bash microbench/ab_libs.sh "pooled=0x1 --pool 2 --planes 4 --rounds 5" $P > $O/ab_pool2.txt 2>&1
bash microbench/ab_libs.sh "u8=0x0 --u8 --pool 2 --planes 8 --rounds 5" $P > $O/ab_u8_pool2.txt 2>&1
bash microbench/ab_libs.sh "u8=0x0 --u8 --pool 4 --planes 2 --rounds 5" $P > $O/ab_u8_pool4.txt 2>&1
bash microbench/ab_libs.sh "cols=0x40001 --mode divide --param 7 --planes 16 --rounds 5" $P > $O/ab_cols.txt 2>&1
R=$O/pmc; mkdir -p $R
for P3 in "A:SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "C:GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_IFETCH SQ_INST_LEVEL_VMEM"; do
  N=${P3%%:*}; C=${P3#*:}
  rocprofv3 --pmc $C --output-format csv -d $R/fwd_u8_$N -- python3 microbench/ab_forward.py v=0x0 --u8 --planes 16 --rounds 3 --iters 5 > $R/fwd_u8_$N.out 2>&1
  rocprofv3 --pmc $C --output-format csv -d $R/inv_u8_$N -- python3 microbench/ab_forward.py v=0x0 --direction inverse --out-type u8 --planes 16 --rounds 3 --iters 5 > $R/inv_u8_$N.out 2>&1
  rocprofv3 --pmc $C --output-format csv -d $R/inv_u8_skipx_$N -- python3 microbench/ab_forward.py v=0x400 --direction inverse --out-type u8 --planes 16 --rounds 3 --iters 5 > $R/inv_u8_skipx_$N.out 2>&1
done
tail -4 $O/ab_u8.txt $O/ab_inv_u8.txt
