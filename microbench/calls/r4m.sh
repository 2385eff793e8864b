O=gpurun_out/r4m; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt -- python3 $GRAFT_REPO_ROOT/microbench/host_api.py > $GRAFT_REPO_ROOT/$O/host_api.txt 2> $GRAFT_REPO_ROOT/$O/err.txt; echo "rc=$?"
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob,collections,statistics
f=glob.glob('gpurun_out/r4m/kt/**/*kernel_trace.csv',recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].replace('(anonymous namespace)::','').split('(')[0][-40:]
    d[(k,r.get('Grid_Size_X') or r.get('Grid_Size'))].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(d.items()):
    print(k, len(v), 'median %.1f min %.1f us'%(statistics.median(v),min(v)))
PY
grep -v "^fresh" $O/host_api.txt | head -12
