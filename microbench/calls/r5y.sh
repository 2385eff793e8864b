O=$GRAFT_REPO_ROOT/gpurun_out/r5y; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for S in nofilter default 2048 3072 4096; do
  unset JPEGX_DECODE_SEG JPEGX_DECODE_NOFILTER
  case $S in nofilter) export JPEGX_DECODE_NOFILTER=1;; default) ;; *) export JPEGX_DECODE_SEG=$S;; esac
  rocprofv3 --kernel-trace --output-format csv -d $O/kt_$S -- python3 $GRAFT_REPO_ROOT/microbench/entropy_stage.py > $O/es_$S.json 2> $O/es_$S.err || exit 1
done
