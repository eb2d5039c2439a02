#!/bin/bash
# every rocprofv3 pass and bench line behind profiles/r04_*: run on the GPU box from the repo root (gpurun), results under gpurun_out/
export TMPDIR=/tmp
bash scripts/profile_step.sh r04_c3 > gpurun_out/prof_r04_c3.log 2>&1
bash scripts/profile_step.sh r04_c5 --workload c5 --scaling weak > gpurun_out/prof_r04_c5.log 2>&1
for wl in c2 c4; do
  out=gpurun_out/prof_r04_$wl; rm -rf $out; mkdir -p $out
  rocprofv3 --kernel-trace --stats -d $out/stats -- python3 bench.py --no-pmc --no-sweep --no-cpu-baseline --workload $wl > $out/stats.log 2>&1
  f=$(find $out/stats -name "*.db" | head -1); [ -n "$f" ] && python3 scripts/rocprof_summary.py $f > $out/kernel_stats.md 2>&1; rm -rf $out/stats
done
python3 bench.py > gpurun_out/bench_r04_default.json 2> gpurun_out/bench_r04_default.err
python3 bench.py --workload c5 --scaling weak > gpurun_out/bench_r04_c5.json 2> gpurun_out/bench_r04_c5.err
python3 bench.py --workload c2 --no-sweep > gpurun_out/bench_r04_c2.json 2> gpurun_out/bench_r04_c2.err
python3 bench.py --workload c4 --no-sweep > gpurun_out/bench_r04_c4.json 2> gpurun_out/bench_r04_c4.err
for wl in c3 c5; do
  python3 bench.py --no-pmc --workload $wl --scaling weak --train > gpurun_out/bench_r04_${wl}_train.json 2> gpurun_out/bench_r04_${wl}_train.err
  python3 bench.py --no-pmc --workload $wl --scaling weak --direction sample > gpurun_out/bench_r04_${wl}_sample.json 2> gpurun_out/bench_r04_${wl}_sample.err
done
python3 scripts/rows_sweep.py c3_e4s2e4 f32 12 20 > gpurun_out/rows_sweep_r04_c3.txt 2>&1
python3 scripts/rows_sweep.py c1_e2_gg f64 12 12 > gpurun_out/rows_sweep_r04_c1.txt 2>&1
ls -la gpurun_out | tail -30
