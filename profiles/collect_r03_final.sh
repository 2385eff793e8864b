#!/bin/bash
# Round-3 evidence, final kernels: run on the GPU box from the repo root, one part per gpurun call
#   gpurun --timeout 1100 -- 'bash profiles/collect_r03_final.sh A'      (traces: bench, entropy stage, host API)
#   gpurun --timeout 1100 -- 'bash profiles/collect_r03_final.sh B'      (counter passes, never together with a trace)
# then locally: python profiles/summarize.py r03_forward gpurun_out/fin/p k_forward_fused_strip
#               python profiles/summarize_pmc.py r03 gpurun_out/fin/pmc
#               python profiles/summarize_trace.py r03_entropy_stage_kernels gpurun_out/fin/entropy_kt  (and the others)
R=gpurun_out/fin; mkdir -p $R
export TMPDIR=/tmp
case "$1" in
A)
    bash profiles/collect_forward.sh $R > $R/collect_forward.log 2>&1; echo "collect_forward rc=$?"
    rm -rf $R/entropy_kt $R/hostapi_kt
    rocprofv3 --kernel-trace --stats --output-format csv -d $R/entropy_kt -- python3 microbench/entropy_stage.py > $R/entropy_stage.json 2> $R/entropy_stage.err; echo "entropy rc=$?"
    rocprofv3 --kernel-trace --stats --output-format csv -d $R/hostapi_kt -- python3 microbench/host_api.py > $R/host_api_under_trace.txt 2> $R/host_api_under_trace.err; echo "hostapi trace rc=$?"
    python3 microbench/host_api.py > $R/host_api.txt 2> $R/host_api.err; echo "hostapi rc=$?"
    python3 microbench/host_api.py --image > $R/host_api_image.txt 2> $R/host_api_image.err; echo "hostapi image rc=$?"
    python3 bench.py > $R/bench_n1.json 2> $R/bench_n1.err; echo "bench rc=$?"; tail -c 400 $R/bench_n1.json
    ;;
B)
    bash profiles/collect_pmc_r03.sh $R/pmc > $R/collect_pmc.log 2>&1; echo "pmc rc=$?"; ls $R/pmc | head -40
    ;;
esac
