O=gpurun_out/r4p; mkdir -p $O
timeout -k 10 1150 python tests/soak_gpu.py --seconds 1080 --seed 77 > $O/soak_1080.txt 2>&1; echo "soak rc=$?"; tail -3 $O/soak_1080.txt
