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
# the float64 C3 step (the secondary leg of the default bench line): kernel trace of 30 evaluations of 2^20 rows, and the int8-slice MLP's counters
out=gpurun_out/prof_r03_c3_f64; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/stats -- python3 scripts/probe/stress.py c3_e4s2e4 f64 30 > $out/log.txt 2>&1
f=$(find $out/stats -name "*.db" | head -1); [ -n "$f" ] && python3 scripts/rocprof_summary.py $f > $out/kernel_stats.md 2>&1; rm -rf $out/stats
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d $out/sq1 -- python3 scripts/probe/stress.py c3_e4s2e4 f64 3 > $out/sq1.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $out/fetch -- python3 scripts/probe/stress.py c3_e4s2e4 f64 3 > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $out/write -- python3 scripts/probe/stress.py c3_e4s2e4 f64 3 > $out/write.log 2>&1
for d in sq1 fetch write; do f=$(find $out/$d -name "*.db" | head -1); [ -n "$f" ] && cp $f $out/$d.db; rm -rf $out/$d; done
python3 scripts/pmc_dump.py $out/sq1.db $out/fetch.db $out/write.db 2>&1 | grep "mlp2_i8\|gf_chain_kernel<double\|gfb_chain_inv_kernel<double" > $out/pmc.txt
rm -f $out/*.db
python3 scripts/bench_configs.py > gpurun_out/bench_configs_r03.txt 2>&1
ls -la gpurun_out | tail -30
