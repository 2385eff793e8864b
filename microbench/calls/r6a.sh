O=$GRAFT_REPO_ROOT/gpurun_out/r6a; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for S in 3328 3584 3840; do
  export JPEGX_DECODE_SEG=$S
  rocprofv3 --kernel-trace --output-format csv -d $O/kt_$S -- python3 $GRAFT_REPO_ROOT/microbench/entropy_stage.py > $O/es_$S.json 2> $O/es_$S.err || exit 1
done
unset JPEGX_DECODE_SEG
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python tests/soak_gpu.py --seconds 360 --seed 123 > $O/soak_360.txt 2>&1; echo "soak rc=$?"; tail -3 $O/soak_360.txt
