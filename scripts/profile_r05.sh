#!/bin/bash
# every rocprofv3 pass and bench line behind profiles/r05_*: run on the GPU box from the repo root (gpurun), results under gpurun_out/
export TMPDIR=/tmp
bash scripts/profile_step.sh r05_c3 > gpurun_out/prof_r05_c3.log 2>&1
bash scripts/profile_step.sh r05_c5 --workload c5 --scaling weak > gpurun_out/prof_r05_c5.log 2>&1
# the same step on ONE stream: kernels do not overlap, so rocprofv3's average and the line's mean_launch_ms are the same quantity (with three streams
# a launch's duration is stretched by its neighbours, and differently under the profiler)
out=gpurun_out/prof_r05_c3_one_stream; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/stats -- python3 bench.py --no-pmc --no-sweep --no-cpu-baseline --pipeline-depth 1 > $out/stats.log 2>&1
f=$(find $out/stats -name "*.db" | head -1); [ -n "$f" ] && python3 scripts/rocprof_summary.py $f > $out/kernel_stats.md 2>&1; rm -rf $out/stats
for wl in c2 c4; do
  out=gpurun_out/prof_r05_$wl; rm -rf $out; mkdir -p $out
  rocprofv3 --kernel-trace --stats -d $out/stats -- python3 bench.py --no-pmc --no-sweep --no-cpu-baseline --workload $wl > $out/stats.log 2>&1
  f=$(find $out/stats -name "*.db" | head -1); [ -n "$f" ] && python3 scripts/rocprof_summary.py $f > $out/kernel_stats.md 2>&1; rm -rf $out/stats
done
# the float64 step of C3: float64 vector instructions by class, busy cycles, clock (two passes: at most 8 counters each)
out=gpurun_out/prof_r05_c3_f64; rm -rf $out; mkdir -p $out
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES -d $out/a -- python3 bench.py --pmc-child --pmc-dtype f64 > $out/a.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_LDS -d $out/b -- python3 bench.py --pmc-child --pmc-dtype f64 > $out/b.log 2>&1
rocprofv3 --kernel-trace --stats -d $out/t -- python3 bench.py --pmc-child --pmc-dtype f64 > $out/t.log 2>&1
for d in a b t; do f=$(find $out/$d -name "*.db" | head -1); [ -n "$f" ] && cp $f $out/$d.db; rm -rf $out/$d; done
python3 scripts/pmc_dump.py $out/a.db $out/b.db > $out/pmc.txt 2>&1
python3 scripts/rocprof_summary.py $out/t.db > $out/kernel_stats.md 2>&1; rm -f $out/t.db
python3 bench.py > gpurun_out/bench_r05_default.json 2> gpurun_out/bench_r05_default.err
python3 bench.py --pipeline-depth 1 --no-cpu-baseline --no-pmc > gpurun_out/bench_r05_one_stream.json 2> gpurun_out/bench_r05_one_stream.err
python3 bench.py --workload c5 --scaling weak > gpurun_out/bench_r05_c5.json 2> gpurun_out/bench_r05_c5.err
python3 bench.py --workload c2 --no-sweep > gpurun_out/bench_r05_c2.json 2> gpurun_out/bench_r05_c2.err
python3 bench.py --workload c4 --no-sweep > gpurun_out/bench_r05_c4.json 2> gpurun_out/bench_r05_c4.err
# the training steps' kernel tables (per-kernel durations of the same command the bench lines below time)
for wl in c3 c5; do
  out=gpurun_out/prof_r05_${wl}_train; rm -rf $out; mkdir -p $out
  rocprofv3 --kernel-trace --stats -d $out/stats -- python3 bench.py --no-pmc --no-sweep --no-cpu-baseline --workload $wl --scaling weak --train > $out/stats.log 2>&1
  f=$(find $out/stats -name "*.db" | head -1); [ -n "$f" ] && python3 scripts/rocprof_summary.py $f > $out/kernel_stats.md 2>&1; rm -rf $out/stats
done
for wl in c2 c4; do
  python3 bench.py --no-pmc --no-cpu-baseline --no-sweep --workload $wl --train > gpurun_out/bench_r05_${wl}_train.json 2> gpurun_out/bench_r05_${wl}_train.err
  python3 bench.py --no-pmc --no-cpu-baseline --no-sweep --workload $wl --direction sample > gpurun_out/bench_r05_${wl}_sample.json 2> gpurun_out/bench_r05_${wl}_sample.err
done
for wl in c3 c5; do
  python3 bench.py --no-pmc --workload $wl --scaling weak --train > gpurun_out/bench_r05_${wl}_train.json 2> gpurun_out/bench_r05_${wl}_train.err
  python3 bench.py --no-pmc --workload $wl --scaling weak --direction sample > gpurun_out/bench_r05_${wl}_sample.json 2> gpurun_out/bench_r05_${wl}_sample.err
done
ls -la gpurun_out | tail -30
