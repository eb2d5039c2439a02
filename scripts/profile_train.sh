#!/bin/bash
# rocprofv3 kernel statistics of the training step (run on the GPU box from the repo root) -> gpurun_out/prof_train_<workload>/
# usage: scripts/profile_train.sh c3|c5 [rows]
wl=$1; rows=${2:-262144}
out=gpurun_out/prof_train_$wl
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
python3 scripts/bench_train.py --workload $wl --rows $rows > $out/bench_train.txt 2>&1
rocprofv3 --kernel-trace --stats -d $out/stats -- python3 scripts/bench_train.py --workload $wl --rows $rows > $out/stats.log 2>&1
f=$(find $out/stats -name "*.db" | head -1); [ -n "$f" ] && cp $f $out/stats.db; rm -rf $out/stats
python3 scripts/rocprof_summary.py $out/stats.db > $out/kernel_stats.md 2>&1
rm -f $out/stats.db                                 # 19 MB of trace: the summary is what travels back (gpurun merges <= 64 MiB)
grep -v "Warn\|amdgpu.ids\|args.workload" $out/bench_train.txt
head -40 $out/kernel_stats.md
