#!/bin/bash
# SQ counters of the C3 training step's kernels (run on the GPU box from the repo root) -> gpurun_out/train_pmc.txt
export TMPDIR=/tmp
out=gpurun_out/train_pmc; mkdir -p $out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -d $out/a -- python3 bench.py --train --no-cpu-baseline --steps 5 --warmup 2 > $out/a.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_ATOMIC_RETURN SQ_WAVES -d $out/b -- python3 bench.py --train --no-cpu-baseline --steps 5 --warmup 2 > $out/b.log 2>&1
for d in a b; do f=$(find $out/$d -name "*.db" | head -1); [ -n "$f" ] && cp $f $out/$d.db; rm -rf $out/$d; done
python3 scripts/pmc_dump.py $out/a.db $out/b.db | grep -i "gf_chain_bwd_kernel<float, 4, true\|cond_gf_split_bwd\|gfb_chain_inv" > gpurun_out/train_pmc.txt
rm -f $out/*.db; tail -2 $out/b.log
