#!/bin/bash
# Round-3 counter passes for the kernels the reference-API path launches: the uint8
# forward kernel (compress_band), the uint8 inverse (decompress_band) and the device entropy decoder.
# Run on the GPU box from the repo root: gpurun -- 'bash profiles/collect_pmc_r03.sh'
# Counters are collected in passes of their own (never together with a trace), at most 8 SQ counters per pass;
# the program stands directly after `--`.  Condense afterwards with: python profiles/summarize_pmc.py r03 gpurun_out/pmc
R=${1:-gpurun_out/pmc}
export TMPDIR=/tmp
mkdir -p $R
PASS_A="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
PASS_B="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA"
PASS_C="GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_IFETCH"
run_pass() {   # name counters... -- program args...
    local name=$1; shift
    local ctr=$1; shift
    rm -rf $R/$name
    rocprofv3 --pmc $ctr --output-format csv -d $R/$name -- "$@" > $R/$name.out 2> $R/$name.err || echo "pass $name failed (see $R/$name.err)"
}
for P in A B C; do
    eval CTR=\$PASS_$P
    run_pass fwd_u8_$P "$CTR" python3 microbench/ab_forward.py v=0x0 --u8 --planes 16 --rounds 3 --iters 5
    run_pass fwd_u8_skipx_$P "$CTR" python3 microbench/ab_forward.py v=0x400 --u8 --planes 16 --rounds 3 --iters 5
    run_pass inv_u8_$P "$CTR" python3 microbench/ab_forward.py v=0x0 --direction inverse --out-type u8 --planes 16 --rounds 3 --iters 5
    run_pass fwd_f32_$P "$CTR" python3 microbench/ab_forward.py v=0x1 --planes 16 --rounds 3 --iters 5
    run_pass entropy_$P "$CTR" python3 microbench/entropy_stage.py
    run_pass band_jobs_$P "$CTR" python3 microbench/host_api.py      # what compress_band / decompress_band launch for single bands: the sized forward, k_rle_emit2, the decoder
done
ls $R
