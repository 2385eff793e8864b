O=$GRAFT_REPO_ROOT/gpurun_out/r6d; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python tests/soak_gpu.py --seconds 600 --seed 4242 > $O/soak_600.txt 2>&1; echo "soak rc=$?"; tail -3 $O/soak_600.txt
