#!/bin/bash
# second half of round 4: rocprofv3 kernel statistics + bench lines of the training and sampling steps (run on the GPU box from the repo root)
export TMPDIR=/tmp
for wl in c3 c5; do
  for mode in train sample; do
    flag="--train"; [ $mode == sample ] && flag="--direction sample"
    out=gpurun_out/prof_r04_${wl}_$mode; rm -rf $out; mkdir -p $out
    python3 bench.py --no-pmc --workload $wl --scaling weak $flag > gpurun_out/bench_r04_${wl}_$mode.json 2> $out/bench.err
    rocprofv3 --kernel-trace --stats -d $out/stats -- python3 bench.py --no-pmc --no-cpu-baseline --workload $wl --scaling weak $flag > $out/stats.log 2>&1
    f=$(find $out/stats -name "*.db" | head -1); [ -n "$f" ] && python3 scripts/rocprof_summary.py $f > $out/kernel_stats.md 2>&1; rm -rf $out/stats
  done
done
python3 scripts/probe/lowrank_check.py 131072 > gpurun_out/lowrank_check_r04.txt 2>&1
ls -la gpurun_out | tail -12
