O=$GRAFT_REPO_ROOT/gpurun_out/r6h; mkdir -p $O
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "entropy or pipeline" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log; [ $rc = 0 ] || exit 1
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/kt_plan -- python3 $GRAFT_REPO_ROOT/microbench/decode_plan_probe.py > $O/plan.txt 2> $O/plan.err || exit 1
export JPEGX_DECODE_SEG=3840
rocprofv3 --kernel-trace --output-format csv -d $O/kt_3840 -- python3 $GRAFT_REPO_ROOT/microbench/decode_plan_probe.py > $O/f3840.txt 2> $O/f3840.err || exit 1
