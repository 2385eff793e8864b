#!/bin/bash
# The three rocprofv3 passes behind profiles/rNN_forward_*: run on the GPU box from the repo root
# (gpurun -- 'bash profiles/collect_forward.sh'), then locally, once gpurun_out/ has been merged back:
#   python profiles/summarize.py rNN_forward gpurun_out/p k_forward_fused_strip
# Counters are collected in passes of their own (never together with a trace); the program stands directly after `--`.
set -e
R=${1:-gpurun_out}; mkdir -p $R
export TMPDIR=/tmp
rm -rf $R/p_kt $R/p_fetch $R/p_write
rocprofv3 --kernel-trace --stats --output-format csv -d $R/p_kt -- python3 bench.py --no-cpu-baseline > $R/p_kt.json 2> $R/p_kt.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/p_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-verify > $R/p_fetch.json 2> $R/p_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/p_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-verify > $R/p_write.json 2> $R/p_write.err
tail -c 600 $R/p_kt.json
