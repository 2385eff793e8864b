O=$GRAFT_REPO_ROOT/gpurun_out/r5m; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python tests/soak_gpu.py --seconds 840 --seed 91 > $O/soak_840.txt 2>&1; echo "soak rc=$?"; tail -5 $O/soak_840.txt
