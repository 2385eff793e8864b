O=gpurun_out/r3f; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log
rm -rf $O/kt_seg $O/kt_gen
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_seg -- python3 microbench/entropy_stage.py > $O/entropy_seg.json 2> $O/entropy_seg.err; echo "seg rc=$?"
JPEGX_DECODE_GENERAL=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_gen -- python3 microbench/entropy_stage.py > $O/entropy_gen.json 2> $O/entropy_gen.err; echo "gen rc=$?"
cat $O/entropy_seg.json; echo; cat $O/entropy_gen.json; echo
for d in kt_seg kt_gen; do echo "== $d"; f=$(ls $O/$d/*/*kernel_stats.csv | head -1); cat $f | cut -d, -f1-4 | head -24; done
python microbench/host_api.py > $O/host_api.txt 2>&1; cat $O/host_api.txt
