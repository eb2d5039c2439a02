#!/bin/bash
# vector / LDS instruction mix of the broadcast g chain's adjoint (C2 training step): two PMC passes, per-kernel dump
export TMPDIR=/tmp
G=gpurun_out/pmc_c2_train; rm -rf $G; mkdir -p $G
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES -d $G/a -- python3 bench.py --no-pmc --no-sweep --no-cpu-baseline --workload c2 --train --steps 4 --warmup 2 > $G/a.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_INT32 -d $G/b -- python3 bench.py --no-pmc --no-sweep --no-cpu-baseline --workload c2 --train --steps 4 --warmup 2 > $G/b.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVES -d $G/c -- python3 bench.py --no-pmc --no-sweep --no-cpu-baseline --workload c2 --train --steps 4 --warmup 2 > $G/c.log 2>&1
for d in a b c; do f=$(find $G/$d -name "*.db" | head -1); [ -n "$f" ] && cp $f $G/$d.db; rm -rf $G/$d; done
python3 scripts/pmc_dump.py $G/a.db $G/b.db $G/c.db > $G/pmc.txt 2>&1
rm -f $G/*.db
grep -c . $G/pmc.txt
