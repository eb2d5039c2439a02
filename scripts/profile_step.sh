#!/bin/bash
# rocprofv3 passes of the benchmark step (run on the GPU box from the repo root): kernel stats, HBM traffic, SQ counters -> gpurun_out/prof_$1
# usage: scripts/profile_step.sh <tag> [bench.py arguments of the child, e.g. --workload c5]
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
# the stats pass runs the DEFAULT bench command (the default number of timed steps, clocks ramped) so that its per-kernel averages are the ones the bench line reports
rocprofv3 --kernel-trace --stats -d $out/stats -- python3 bench.py --no-pmc --no-sweep "$@" > $out/stats.log 2>&1     # (--no-sweep: only full-size launches, so the averages are the timed step's)
rocprofv3 --pmc FETCH_SIZE -d $out/fetch -- python3 bench.py --pmc-child "$@" > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $out/write -- python3 bench.py --pmc-child "$@" > $out/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_LDS -d $out/sq1 -- python3 bench.py --pmc-child "$@" > $out/sq1.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_BUSY_CYCLES -d $out/sq2 -- python3 bench.py --pmc-child "$@" > $out/sq2.log 2>&1
for d in stats fetch write sq1 sq2; do f=$(find $out/$d -name "*.db" | head -1); [ -n "$f" ] && cp $f $out/$d.db; rm -rf $out/$d; done
python3 scripts/rocprof_summary.py $out/stats.db > $out/kernel_stats.md 2>&1
python3 scripts/pmc_dump.py $out/fetch.db $out/write.db $out/sq1.db $out/sq2.db > $out/pmc.txt 2>&1
rm -f $out/stats.db                                 # the kernel trace is the bulk; its summary is kernel_stats.md (gpurun merges <= 64 MiB)
ls -la $out
