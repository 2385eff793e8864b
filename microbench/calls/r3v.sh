O=gpurun_out/r3w; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_entropy.py -x -q > $O/pytest_entropy.log 2>&1; echo "pytest entropy rc=$?"; tail -5 $O/pytest_entropy.log
JPEGX_DECODE_STATS=$O/trace.bin JPEGX_LIB_PATH=microbench/_ab/libjpegx_stats.so timeout -k 10 120 python microbench/decode_trace_run.py
python microbench/decode_trace.py $O/trace.bin.noise | grep -v "^  seg"
python microbench/decode_trace.py $O/trace.bin.smooth | grep -v "^  seg"
rm -f $O/trace.bin.*
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt -- python3 $GRAFT_REPO_ROOT/microbench/entropy_stage.py > $GRAFT_REPO_ROOT/$O/entropy_stage.json 2> $GRAFT_REPO_ROOT/$O/entropy_stage.err; echo "rc=$?"
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob,collections,statistics
f=glob.glob('gpurun_out/r3w/kt/**/*kernel_trace.csv',recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].replace('(anonymous namespace)::','').split('(')[0][-34:]
    d[(k,r.get('Grid_Size_X') or r.get('Grid_Size'))].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(d.items()):
    if 'seg' in k[0] or 'dec' in k[0] or 'fill' in k[0]: print(k, len(v), 'median %.1f min %.1f us'%(statistics.median(v),min(v)))
PY
