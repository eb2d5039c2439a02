#!/bin/bash
# SQ counters of the low-rank chain kernels (run on the GPU box from the repo root) -> gpurun_out/lr_pmc.txt
export TMPDIR=/tmp
out=gpurun_out/lr_pmc; mkdir -p $out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS -d $out/a -- python3 scripts/probe/lowrank_check.py 131072 > $out/a.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE -d $out/b -- python3 scripts/probe/lowrank_check.py 131072 > $out/b.log 2>&1
for d in a b; do f=$(find $out/$d -name "*.db" | head -1); [ -n "$f" ] && cp $f $out/$d.db; rm -rf $out/$d; done
python3 scripts/pmc_dump.py $out/a.db $out/b.db | grep -i "lr_\|gf_chain_bwd_kernel\|gf_chain_kernel" > gpurun_out/lr_pmc.txt
tail -3 $out/a.log; rm -f $out/*.db
