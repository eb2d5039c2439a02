#!/bin/bash
# additional round-6 evidence: kernel tables of the sampling direction (C3, C5, c3b), vector / matrix / LDS counters of the C3 training step's kernels
export TMPDIR=/tmp
G=gpurun_out
stats() { out=$1; shift; rm -rf $out; mkdir -p $out
  rocprofv3 --kernel-trace --stats -d $out/stats -- python3 bench.py --no-pmc --no-sweep --no-cpu-baseline "$@" > $out/stats.log 2>&1
  f=$(find $out/stats -name "*.db" | head -1); [ -n "$f" ] && python3 scripts/rocprof_summary.py $f > $out/kernel_stats.md 2>&1; rm -rf $out/stats; }
for wl in c3 c5 c3b; do stats $G/prof_r06_${wl}_sample --workload $wl --scaling weak --direction sample; done
P=$G/pmc_c3_train; rm -rf $P; mkdir -p $P
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES -d $P/a -- python3 bench.py --no-pmc --no-sweep --no-cpu-baseline --workload c3 --scaling weak --train --steps 4 --warmup 2 > $P/a.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d $P/b -- python3 bench.py --no-pmc --no-sweep --no-cpu-baseline --workload c3 --scaling weak --train --steps 4 --warmup 2 > $P/b.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_ANY SQ_WAVES -d $P/c -- python3 bench.py --no-pmc --no-sweep --no-cpu-baseline --workload c3 --scaling weak --train --steps 4 --warmup 2 > $P/c.log 2>&1
for d in a b c; do f=$(find $P/$d -name "*.db" | head -1); [ -n "$f" ] && cp $f $P/$d.db; rm -rf $P/$d; done
python3 scripts/pmc_dump.py $P/a.db $P/b.db $P/c.db > $P/pmc.txt 2>&1; rm -f $P/*.db
ls $G | grep -E "sample|pmc_c3" | tr "\n" " "
