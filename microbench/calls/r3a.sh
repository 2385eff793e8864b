set -o pipefail
O=gpurun_out/r3a; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -3 $O/pytest.log
microbench/valu_rate > $O/valu_rate.txt 2>&1; echo "valu_rate rc=$?"
rocprofv3 -L > $O/counters_list.txt 2>&1; echo "counters list rc=$?"
ls /dev/shm > $O/shm_before.txt 2>&1
bash profiles/collect_pmc_r03.sh $O/pmc > $O/pmc.log 2>&1; echo "pmc rc=$?"
ls -la /dev/shm > $O/shm_after_pmc.txt 2>&1
# rehearsals right after the profiler passes (the condition under which round 2's rehearsal hung)
timeout -k 10 400 python bench.py --gpus 2 --share-device --planes-total 64 --steps 5 --warmup 2 --no-cpu-baseline > $O/reh_own2.json 2> $O/reh_own2.err; echo "reh_own2 rc=$?"
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --share-device --planes-total 64 --steps 5 --warmup 2 --no-cpu-baseline > $O/reh_torch2.json 2> $O/reh_torch2.err; echo "reh_torch2 rc=$?"
# the same with the process-shared rocm_smi mutex (what round 2 ran with), for the record
RSMI_MUTEX_THREAD_ONLY=0 timeout -k 10 400 python bench.py --gpus 2 --share-device --planes-total 64 --steps 5 --warmup 2 --no-cpu-baseline --comm-timeout 45 > $O/reh_own2_sharedmutex.json 2> $O/reh_own2_sharedmutex.err; echo "reh_own2_sharedmutex rc=$?"
bash microbench/ab_libs.sh "u8=0x0 u8_skipx=0x400 --u8 --planes 16 --rounds 5" microbench/_ab/libjpegx_wpe5.so > $O/ab_u8_wpe5.txt 2>&1; echo "ab rc=$?"
python microbench/ab_forward.py inv=0x0 skipx=0x400 --direction inverse --out-type u8 --planes 16 --rounds 5 > $O/ab_inv_u8.txt 2>&1
python microbench/host_api.py > $O/host_api.txt 2>&1; echo "host_api rc=$?"
tail -5 $O/valu_rate.txt
