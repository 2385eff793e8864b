O=gpurun_out/r3p; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt -- python3 $GRAFT_REPO_ROOT/microbench/entropy_stage.py > $GRAFT_REPO_ROOT/$O/entropy_stage.json 2> $GRAFT_REPO_ROOT/$O/entropy_stage.err; echo "rc=$?"
cd $GRAFT_REPO_ROOT
cat $O/entropy_stage.json
python - <<'PY'
import csv,glob,collections,statistics
f=glob.glob('gpurun_out/r3p/kt/**/*kernel_trace.csv',recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].split('(')[0][-40:]
    d[(k,r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size'))].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(d.items()):
    print(k, len(v), 'median %.1f min %.1f us'%(statistics.median(v),min(v)))
PY
timeout -k 10 400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
