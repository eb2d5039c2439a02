#!/bin/bash
# every rocprofv3 pass behind profiles/r03_*: run on the GPU box from the repo root (gpurun), results under gpurun_out/
export TMPDIR=/tmp
bash scripts/profile_step.sh r03_c3 > gpurun_out/prof_r03_c3.log 2>&1
bash scripts/profile_step.sh r03_c5 --workload c5 > gpurun_out/prof_r03_c5.log 2>&1
for wl in c3 c5; do
  out=gpurun_out/prof_r03_sample_$wl; rm -rf $out; mkdir -p $out
  python3 bench.py --no-pmc --workload $wl --direction sample > $out/bench.json 2> $out/bench.err
  rocprofv3 --kernel-trace --stats -d $out/stats -- python3 bench.py --no-pmc --workload $wl --direction sample > $out/stats.log 2>&1
  f=$(find $out/stats -name "*.db" | head -1); [ -n "$f" ] && python3 scripts/rocprof_summary.py $f > $out/kernel_stats.md 2>&1; rm -rf $out/stats
done
bash scripts/profile_train.sh c3 262144 pmc > gpurun_out/prof_train_c3.log 2>&1
bash scripts/profile_train.sh c5 131072 pmc > gpurun_out/prof_train_c5.log 2>&1
for wl in c3 c5; do python3 bench.py --no-pmc --workload $wl --train > gpurun_out/bench_r03_${wl}_train.json 2> gpurun_out/bench_r03_${wl}_train.err; done
python3 bench.py > gpurun_out/bench_r03_default.json 2> gpurun_out/bench_r03_default.err
python3 bench.py --workload c5 > gpurun_out/bench_r03_c5.json 2> gpurun_out/bench_r03_c5.err
ls -la gpurun_out | tail -30
