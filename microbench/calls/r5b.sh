O=$GRAFT_REPO_ROOT/gpurun_out/r5b; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for S in 2048 2816 3072 3328 3584; do
  export JPEGX_DECODE_SEG=$S
  rocprofv3 --kernel-trace --output-format csv -d $O/kt$S -- python3 $GRAFT_REPO_ROOT/microbench/entropy_stage.py > $O/es$S.json 2> $O/es$S.err || exit 1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob,collections,statistics
for S in (2048,2816,3072,3328,3584):
    f=glob.glob('gpurun_out/r5b/kt%d/**/*kernel_trace.csv'%S,recursive=True)[0]
    d=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].replace('(anonymous namespace)::','').split('(')[0][-34:]
        d[(k,r.get('Grid_Size_X') or r.get('Grid_Size'))].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
    for k,v in sorted(d.items()):
        if 'seg' in k[0] or 'dec_' in k[0]: print(S, k, len(v), 'median %.1f min %.1f us'%(statistics.median(v),min(v)))
PY
