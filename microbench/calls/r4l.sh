O=gpurun_out/r4l; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for sg in 1280 1536 1792 2048 2560 3072 4096; do
JPEGX_DECODE_SEG=$sg rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt$sg -- python3 $GRAFT_REPO_ROOT/microbench/entropy_stage.py > /dev/null 2> $GRAFT_REPO_ROOT/$O/err$sg.txt; echo "seg $sg rc=$?"
python3 - <<PY
import csv,glob,collections,statistics
f=glob.glob('$GRAFT_REPO_ROOT/$O/kt$sg/**/*kernel_trace.csv',recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].replace('(anonymous namespace)::','').split('(')[0][-34:]
    d[(k,r.get('Grid_Size_X') or r.get('Grid_Size'))].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(d.items()):
    if 'seg_starts' in k[0] or 'dec_blocks' in k[0] or 'k_dec_parse' in k[0]: print('  ',k, len(v), 'median %.1f min %.1f us'%(statistics.median(v),min(v)))
PY
done
