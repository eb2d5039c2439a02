#!/bin/bash
# rocprofv3 kernel statistics of the training step (run on the GPU box from the repo root) -> gpurun_out/prof_train_<workload>/
# usage: scripts/profile_train.sh c3|c5 [rows] [pmc]   (pmc: also SQ / HBM counter passes -> pmc.txt)
wl=$1; rows=${2:-262144}
out=gpurun_out/prof_train_$wl
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
python3 scripts/bench_train.py --workload $wl --rows $rows > $out/bench_train.txt 2>&1
rocprofv3 --kernel-trace --stats -d $out/stats -- python3 scripts/bench_train.py --workload $wl --rows $rows > $out/stats.log 2>&1
f=$(find $out/stats -name "*.db" | head -1); [ -n "$f" ] && cp $f $out/stats.db; rm -rf $out/stats
python3 scripts/rocprof_summary.py $out/stats.db > $out/kernel_stats.md 2>&1
rm -f $out/stats.db                                 # 19 MB of trace: the summary is what travels back (gpurun merges <= 64 MiB)
if [ "$3" == "pmc" ]; then
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_LDS -d $out/sq1 -- python3 scripts/bench_train.py --workload $wl --rows $rows --pmc-child > $out/sq1.log 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_BUSY_CYCLES -d $out/sq2 -- python3 scripts/bench_train.py --workload $wl --rows $rows --pmc-child > $out/sq2.log 2>&1
  rocprofv3 --pmc FETCH_SIZE -d $out/fetch -- python3 scripts/bench_train.py --workload $wl --rows $rows --pmc-child > $out/fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $out/write -- python3 scripts/bench_train.py --workload $wl --rows $rows --pmc-child > $out/write.log 2>&1
  for d in sq1 sq2 fetch write; do f=$(find $out/$d -name "*.db" | head -1); [ -n "$f" ] && cp $f $out/$d.db; rm -rf $out/$d; done
  python3 scripts/pmc_dump.py $out/sq1.db $out/sq2.db $out/fetch.db $out/write.db > $out/pmc.txt 2>&1
  rm -f $out/sq1.db $out/sq2.db $out/fetch.db $out/write.db
fi
grep -v "Warn\|amdgpu.ids\|args.workload" $out/bench_train.txt
head -40 $out/kernel_stats.md
