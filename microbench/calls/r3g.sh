O=gpurun_out/r3g; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "entropy or decompress or image or pipeline" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
rm -rf $O/kt_seg
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_seg -- python3 microbench/entropy_stage.py > $O/entropy_seg.json 2> $O/entropy_seg.err; echo "seg rc=$?"
