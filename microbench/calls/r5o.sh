O=gpurun_out/r5o; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --share-device --planes-total 64 --steps 5 --warmup 2 --no-cpu-baseline > $O/reh_torch2.json 2> $O/reh_torch2.err; echo "reh_torch2 rc=$?"
timeout -k 10 300 python bench.py --gpus 4 --share-device --planes-total 64 --steps 5 --warmup 2 --no-cpu-baseline > $O/reh_own4.json 2> $O/reh_own4.err; echo "reh_own4 rc=$?"
