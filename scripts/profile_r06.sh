#!/bin/bash
# every rocprofv3 pass and bench line behind profiles/r06_*: run on the GPU box from the repo root (gpurun), results under gpurun_out/
# (scripts/collect_r06.py turns them into the committed summaries).  PMC passes and kernel traces are separate runs (gpurun refuses the mix).
export TMPDIR=/tmp
G=gpurun_out
stats() {   # $1 = out dir, rest = bench arguments: rocprofv3 --kernel-trace --stats of that command -> kernel_stats.md
  out=$1; shift; rm -rf $out; mkdir -p $out
  rocprofv3 --kernel-trace --stats -d $out/stats -- python3 bench.py --no-pmc --no-sweep --no-cpu-baseline "$@" > $out/stats.log 2>&1
  f=$(find $out/stats -name "*.db" | head -1); [ -n "$f" ] && python3 scripts/rocprof_summary.py $f > $out/kernel_stats.md 2>&1; rm -rf $out/stats
}
issue() {   # $1 = out dir, $2 = f32|f64, rest = --pmc-child arguments: vector instructions by class, busy cycles, clock + kernel durations
  out=$1; dt=$2; shift; shift; rm -rf $out; mkdir -p $out
  if [ $dt = f64 ]; then
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES -d $out/a -- python3 bench.py --pmc-child --pmc-dtype f64 "$@" > $out/a.log 2>&1
  else
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES -d $out/a -- python3 bench.py --pmc-child --pmc-dtype f32 "$@" > $out/a.log 2>&1
  fi
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_LDS SQ_INSTS_VALU_INT32 -d $out/b -- python3 bench.py --pmc-child --pmc-dtype $dt "$@" > $out/b.log 2>&1
  rocprofv3 --kernel-trace --stats -d $out/t -- python3 bench.py --pmc-child --pmc-dtype $dt "$@" > $out/t.log 2>&1
  for d in a b t; do f=$(find $out/$d -name "*.db" | head -1); [ -n "$f" ] && cp $f $out/$d.db; rm -rf $out/$d; done
  python3 scripts/pmc_dump.py $out/a.db $out/b.db > $out/pmc.txt 2>&1
  python3 scripts/rocprof_summary.py $out/t.db > $out/kernel_stats.md 2>&1; rm -f $out/t.db
}
PART=${1:-all}      # 1: kernel stats + traffic, 2: vector-issue profiles, 3: the log-prob bench lines, 4: training / sampling / the fixture scan
if [ $PART = all ] || [ $PART = 1 ]; then
# ---- per-kernel durations of the default command (three streams), the same step on one stream, the other configurations
stats $G/prof_r06_c3
stats $G/prof_r06_c3_one_stream --pipeline-depth 1
for wl in c2 c4 c3b c5; do stats $G/prof_r06_$wl --workload $wl; done
# ---- HBM traffic of the default command (FETCH_SIZE and WRITE_SIZE cannot share a pass)
for c in FETCH_SIZE WRITE_SIZE; do
  out=$G/prof_r06_c3_traffic_$c; rm -rf $out; mkdir -p $out
  rocprofv3 --pmc $c -d $out/p -- python3 bench.py --pmc-child > $out/p.log 2>&1
  f=$(find $out/p -name "*.db" | head -1); [ -n "$f" ] && cp $f $out/p.db; rm -rf $out/p
done
fi
if [ $PART = all ] || [ $PART = 2 ]; then
# ---- the vector-issue profiles (instructions per row by class, clock): float32 of C3 / C2 / C4 / C3b, float64 of C3
issue $G/prof_r06_issue_c3_f32 f32
issue $G/prof_r06_issue_c2_f32 f32 --workload c2
issue $G/prof_r06_issue_c4_f32 f32 --workload c4
issue $G/prof_r06_issue_c3b_f32 f32 --workload c3b
issue $G/prof_r06_issue_c3_f64 f64
fi
if [ $PART = all ] || [ $PART = 3 ]; then
# ---- the bench lines
python3 bench.py > $G/bench_r06_default.json 2> $G/bench_r06_default.err
python3 bench.py --pipeline-depth 1 --no-cpu-baseline --no-pmc > $G/bench_r06_one_stream.json 2> $G/bench_r06_one_stream.err
python3 bench.py --workload c5 --scaling weak > $G/bench_r06_c5.json 2> $G/bench_r06_c5.err
for wl in c2 c4 c3b; do python3 bench.py --workload $wl --no-sweep > $G/bench_r06_$wl.json 2> $G/bench_r06_$wl.err; done
fi
if [ $PART = all ] || [ $PART = 4 ]; then
# ---- the training steps' kernel tables and lines, sampling lines
for wl in c3 c3b c5; do stats $G/prof_r06_${wl}_train --workload $wl --scaling weak --train; done
stats $G/prof_r06_c4_train --workload c4 --train
for wl in c2 c4; do
  python3 bench.py --no-pmc --no-cpu-baseline --no-sweep --workload $wl --train > $G/bench_r06_${wl}_train.json 2> $G/bench_r06_${wl}_train.err
  python3 bench.py --no-pmc --no-cpu-baseline --no-sweep --workload $wl --direction sample > $G/bench_r06_${wl}_sample.json 2> $G/bench_r06_${wl}_sample.err
done
for wl in c3 c3b c5; do
  python3 bench.py --no-pmc --workload $wl --scaling weak --train > $G/bench_r06_${wl}_train.json 2> $G/bench_r06_${wl}_train.err
  python3 bench.py --no-pmc --workload $wl --scaling weak --direction sample > $G/bench_r06_${wl}_sample.json 2> $G/bench_r06_${wl}_sample.err
done
# ---- every golden fixture's forward / sampling / training step (the table DESIGN section 8 quotes)
python3 scripts/probe/scan_fixtures.py 65536 "" f64 > $G/scan_r06_f64.txt 2>&1
python3 scripts/probe/scan_fixtures.py 65536 "" f32 > $G/scan_r06_f32.txt 2>&1
fi
ls -la $G | tail -40
